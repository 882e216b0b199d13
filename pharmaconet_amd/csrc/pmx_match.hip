// pmx_match.hip - the fused matcher: one wavefront scores one ligand, start to finish, out of LDS (gfx950, wave64).
//
// For a ligand the wave (1) builds the self / pair score tables of match_utils.py in its slice of LDS, (2) derives the
// search bounds from them and (3) walks ClusterMatchTree.dfs_run (tree.py:55-104) over them - nothing but the packed
// record is read from HBM and nothing but the score is written. Ligands are binned by the LDS bytes their tables need
// (bin_kernel); every bin is one launch of match_kernel with that much LDS per wave, whose wavefronts pull ligands off
// the bin's list with an atomic cursor until it is empty.
//
// Table phase (graph_match.py:222-279, match_utils.py:9-122). A table entry (ligand cluster i, model cluster a, ligand
// cluster j, model cluster b) is the sum over ligand node pairs (u in i, v in j) of a sum over model node pairs
// (m in a, n in b) compatible with (u, v) of a Gaussian in the u-v distance. Which (m, n) take part depends only on the
// PATTERN of u and v: pattern(u) = (candidate clusters of u's ligand cluster, type mask of u). So the wave groups the
// ligand's nodes by pattern and, for every ordered pair of patterns, puts the node pairs (u, v) x conformers on its
// lanes: all lanes then run the SAME loops over model clusters and model nodes - the edge parameters are wave-uniform
// (one LDS broadcast read per term), nothing diverges, and each lane ends up with complete per-entry sums and pass
// counts for its (u, v, conformer), which it adds into the entry accumulators in LDS. Rows are shared by the model
// clusters that contain them (a model node sits in several clusters, density_map.py:131-177): a row's partial sum over
// the nodes n of cluster b is computed once and added to every row-side cluster that contains m - the loop structure
// of the reference's Numba kernel (match_utils_numba.py:70-83).
//
// Tree phase. Control is wave-uniform (one DFS per wave, its stack in lane-indexed registers); the data-parallel part
// is candidates x conformers: lane (s, c) evaluates candidate s of the frame's level for conformer c against the
// matched ancestors. Exactness of the bound test and of handing subtrees to other waves: DESIGN.md section 3.
#include "pmx_device.h"

#pragma clang fp contract(off)

namespace pmx {

constexpr int kXC = 8;            // row-side model clusters accumulated per pass over the rows
constexpr int kPairBuf = 128;     // node pairs buffered before they are cut into batches
constexpr int kNumBins = 10;      // LDS size classes
constexpr uint32_t kBinBig = kNumBins; // tables that fit no class: kept in HBM (match_kernel<G, false>)

// Per-wave LDS header of the fused matcher (fixed part; the tables follow).
struct MatchCtx {
    uint64_t cand[PMX_MAX_LEVELS];     // candidate model clusters of each level (graph_match.py:124-137)
    uint32_t rowbase[PMX_MAX_LEVELS];  // first pair entry of level i
    uint16_t ksum[PMX_MAX_LEVELS + 4]; // exclusive prefix sums of k
    uint8_t lstart[PMX_MAX_LEVELS];    // node range of each level's ligand cluster
    uint8_t lend[PMX_MAX_LEVELS];
    uint8_t lk[PMX_MAX_LEVELS];        // candidates per level
    uint8_t pad0[4];
    uint8_t nodelevel[64];             // level of each ligand node (0xff: its cluster has no candidate / is beyond the cap)
    uint8_t nodetm[64];                // type mask of each ligand node
    uint16_t pairbuf[kPairBuf];        // u | v << 8
    uint32_t pad1[8];
};
static_assert(sizeof(MatchCtx) == 768, "MatchCtx layout");

// LDS bytes one ligand needs: header, P [T][G] f32, S [ksumtot][G] f32, then the larger of the table phase's fail
// counts (u16 [T][G]) and the tree phase's bounds + path totals (2 x f64 [nl + 1][G]).
template <int G>
__host__ __device__ inline uint32_t match_bytes(uint32_t T, uint32_t ksumtot, uint32_t nl) {
    const uint32_t p = (uint32_t)round16(uint64_t(T) * G * 4), s = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
    const uint32_t f = (uint32_t)round16(uint64_t(T) * G * 2), w = 2u * (nl + 1) * G * 8;
    return (uint32_t)sizeof(MatchCtx) + p + s + (f > w ? f : w);
}

// LDS bytes of the staged model: folded edge table + cluster node sets + type -> node sets.
__host__ __device__ inline uint32_t model_lds_bytes(int Nm) { return (uint32_t)round16(uint64_t(Nm) * Nm * 16) + 64 * 8 + 128 * 8; }

struct BinInfo {          // device-resident, written by bin_kernel, read by the match kernels
    uint32_t count[kNumBins + 1];
    uint32_t cursor[kNumBins + 1];
    uint32_t cap[kNumBins];   // LDS bytes per wave of each class
    uint32_t max_need;        // largest table of the call (bytes)
    uint32_t big_slot_cursor;
};

struct MatchParams {
    DevModel M;
    const float4 *wtab; // [Nm * Nm] {mean, s, T, w_m w_n / std}: the call's weights folded in (match_utils.py:65)
    const float *wsum;  // [K * 128] sum of the weights of cluster a's nodes compatible with type mask t
    DevLibrary lib;
    uint64_t first;      // library index of the call's first ligand
    const uint32_t *list; // this class's ligands (indices into the call's range)
    BinInfo *bins;
    uint32_t bin;        // class index
    uint32_t wave_bytes; // LDS bytes per wave (class cap); 0 for the HBM class
    uint8_t *arena;      // HBM class: table memory, cut into equal slots
    uint64_t arena_bytes;
    float *scores;
    unsigned long long *stats; // [0] tree steps [1] pairs batches [2] lane-terms / 64 (diagnostics)
    uint32_t flags;      // 4: no bound test
};

// -------------------------------------------------------------------------------------------- small helpers
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
    return (uint64_t)(uint32_t)uni((int)(uint32_t)x) | ((uint64_t)(uint32_t)uni((int)(uint32_t)(x >> 32)) << 32);
}
__device__ __forceinline__ int lane_get(int v, int idx) { return __builtin_amdgcn_readlane(v, idx); }
__device__ __forceinline__ void lane_set(int &v, int idx, int val) { v = ((int)(threadIdx.x & 63) == idx) ? val : v; }

__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off);
    hi = __shfl_xor(hi, off);
    return __hiloint2double(hi, lo);
}

// One Gaussian term (match_utils.py:55-68): e = {mean, s, T, w / std}
template <bool PASS>
__device__ __forceinline__ void gterm(const float d, const float4 e, float &acc, unsigned &np) {
    const float t = __builtin_fabsf(d - e.x);
    const float q = t * e.y;
    acc = __builtin_fmaf(e.w, __builtin_amdgcn_exp2f(-(q * q)), acc);
    if (PASS) np += (t <= e.z) ? 1u : 0u;
}

// The terms of one row m against the model nodes of column set B (ascending n), all wave-uniform; B non-empty.
template <bool PASS>
__device__ __forceinline__ void row_terms(const float4 *row, uint64_t B, const float d, float &acc, unsigned &np) {
    for (;;) {
        const int n0 = __ffsll((unsigned long long)B) - 1;
        B &= B - 1;
        if (!B) {
            gterm<PASS>(d, row[n0], acc, np);
            return;
        }
        const int n1 = __ffsll((unsigned long long)B) - 1;
        B &= B - 1;
        if (!B) {
            const float4 e0 = row[n0], e1 = row[n1];
            gterm<PASS>(d, e0, acc, np);
            gterm<PASS>(d, e1, acc, np);
            return;
        }
        const int n2 = __ffsll((unsigned long long)B) - 1;
        B &= B - 1;
        if (!B) {
            const float4 e0 = row[n0], e1 = row[n1], e2 = row[n2];
            gterm<PASS>(d, e0, acc, np);
            gterm<PASS>(d, e1, acc, np);
            gterm<PASS>(d, e2, acc, np);
            return;
        }
        const int n3 = __ffsll((unsigned long long)B) - 1;
        B &= B - 1;
        const float4 e0 = row[n0], e1 = row[n1], e2 = row[n2], e3 = row[n3];
        gterm<PASS>(d, e0, acc, np);
        gterm<PASS>(d, e1, acc, np);
        gterm<PASS>(d, e2, acc, np);
        gterm<PASS>(d, e3, acc, np);
        if (!B) return;
    }
}

// ------------------------------------------------------------------------------------------------ bin_kernel
// One thread per ligand: levels and table sizes (graph_match.py:124-137, :87-88) -> LDS size class; ligands without a
// tree get their score here (0: graph_match.py:95-99; NaN: record outside the structural limits).
template <int G>
__global__ void bin_kernel(DevLibrary lib, const uint64_t *tclus, uint64_t first, uint32_t count, BinInfo *bins, uint32_t *lists,
                           int32_t *status, float *scores) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Record r = parse_record(lib.data + lib.offsets[first + i]);
    if (!record_supported(r)) {
        if (status) status[i] = PMX_LIGAND_UNSUPPORTED;
        scores[i] = __builtin_nanf("");
        return;
    }
    if (status) status[i] = PMX_LIGAND_OK;
    const Levels L = scan_levels(r, tclus, [](int, int, int, uint64_t, uint32_t) {});
    if (L.nl == 0) {
        scores[i] = 0.f;
        return;
    }
    const uint32_t need = match_bytes<G>(L.T, L.ksumtot, (uint32_t)L.nl);
    uint32_t b = 0;
    while (b < (uint32_t)kNumBins && need > bins->cap[b]) ++b;
    const uint32_t pos = atomicAdd(&bins->count[b], 1u);
    lists[(size_t)b * count + pos] = i;
    atomicMax(&bins->max_need, need);
}

__global__ void bins_init_kernel(BinInfo *bins, const uint32_t *caps, unsigned long long *stats) {
    const int t = threadIdx.x;
    if (t <= kNumBins) {
        bins->count[t] = 0;
        bins->cursor[t] = 0;
    }
    if (t < kNumBins) bins->cap[t] = caps[t];
    if (t == 0) {
        bins->max_need = 0;
        bins->big_slot_cursor = 0;
    }
    if (t < 16) stats[t] = 0;
}

// The call's weights folded into the edge table, and the weight sums the reference normalises with.
__global__ void fold_weights_kernel(DevModel M, Weights W, float4 *wtab, float *wsum) {
    const int Nm = M.Nm, K = M.K;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Nm * Nm) {
        const int m = t / Nm, n = t - m * Nm;
        float4 e = M.edge[t];
        e.w = (W.w[M.node_type[m]] * W.w[M.node_type[n]]) / e.w; // weights / stds (match_utils.py:65)
        wtab[t] = e;
    }
    if (t < K * 128) {
        const int a = t >> 7, tm = t & 127;
        const uint64_t nodes = M.cnodes[a] & M.tnodes[tm];
        float s = 0.f;
        for (int m = 0; m < Nm; ++m)
            if ((nodes >> m) & 1) s = s + W.w[M.node_type[m]];
        wsum[t] = s;
    }
}

// --------------------------------------------------------------------------------------------- the matcher
template <int G, bool LDS_TABLES>
struct Matcher {
    static constexpr int NP = 64 / G; // node pairs (table phase) / candidates (tree phase) per pass of the wave
    const MatchParams &p;
    const float4 *tab;      // LDS: folded edge table [Nm][Nm]
    const uint64_t *cnodes; // LDS [64]
    const uint64_t *tnodes; // LDS [128]
    MatchCtx &X;
    float *Pt;              // [T][G]
    float *St;              // [ksumtot][G]
    uint32_t *Ft;           // table phase: fail counts, two u16 per word
    double *Rt;             // tree phase: bounds [(nl + 1)][G]   (overlays Ft)
    double *Tt;             // tree phase: path totals by match count [(nl + 1)][G]
    const int lane, s, c;
    Record r;
    int C, cc, nl;
    bool lane_live;
    uint32_t T, ksumtot;
    int Nm;
    unsigned long long n_steps = 0, n_batches = 0, n_terms = 0;

    __device__ Matcher(const MatchParams &p_, const float4 *tab_, const uint64_t *cn_, const uint64_t *tn_, unsigned char *ctx)
        : p(p_), tab(tab_), cnodes(cn_), tnodes(tn_), X(*reinterpret_cast<MatchCtx *>(ctx)), lane(threadIdx.x & 63), s((threadIdx.x & 63) / G),
          c((threadIdx.x & 63) % G) {}

    __device__ __forceinline__ void lds_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    __device__ __forceinline__ float dist(int u, int v) const {
        const float *pu = r.xyz + (size_t)(u * 3) * C + cc, *pv = r.xyz + (size_t)(v * 3) * C + cc;
        return norm3(pu[0] - pv[0], pu[C] - pv[C], pu[2 * C] - pv[2 * C]);
    }

    // ---- levels, node tables, cleared accumulators. Returns false when the ligand has no tree.
    __device__ bool setup(uint32_t li, unsigned char *tables) {
        r = parse_record(p.lib.data + p.lib.offsets[p.first + li]);
        C = r.C;
        cc = c < C ? c : C - 1;
        lane_live = c < C;
        Nm = p.M.Nm;
        const Levels L = scan_levels(r, p.M.tclus, [&](int lev, int start, int end, uint64_t cand, uint32_t k) {
            X.cand[lev] = cand;
            X.lstart[lev] = (uint8_t)start;
            X.lend[lev] = (uint8_t)end;
            X.lk[lev] = (uint8_t)k;
        });
        nl = uni(L.nl);
        T = (uint32_t)uni((int)L.T);
        ksumtot = (uint32_t)uni((int)L.ksumtot);
        if (nl == 0) return false;
        {
            uint32_t ks = 0, rb = 0;
            for (int i = 0; i < nl; ++i) {
                const uint32_t k = X.lk[i];
                X.ksum[i] = (uint16_t)ks;
                X.rowbase[i] = rb;
                ks += k;
                rb += k * (ksumtot - ks);
            }
            X.ksum[nl] = (uint16_t)ks;
        }
        X.nodelevel[lane] = 0xff;
        X.nodetm[lane] = lane < r.n ? r.typemask[lane] : (uint8_t)0;
        lds_sync();
        {
            int mine = 0xff;
            for (int lev = 0; lev < nl; ++lev)
                if (lane >= (int)X.lstart[lev] && lane < (int)X.lend[lev]) mine = lev;
            X.nodelevel[lane] = (uint8_t)mine;
        }
        const uint32_t p_bytes = (uint32_t)round16(uint64_t(T) * G * 4), s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
        const uint32_t f_bytes = (uint32_t)round16(uint64_t(T) * G * 2);
        Pt = reinterpret_cast<float *>(tables);
        St = reinterpret_cast<float *>(tables + p_bytes);
        Ft = reinterpret_cast<uint32_t *>(tables + p_bytes + s_bytes);
        Rt = reinterpret_cast<double *>(tables + p_bytes + s_bytes);
        Tt = Rt + (size_t)(nl + 1) * G;
        uint32_t *z = reinterpret_cast<uint32_t *>(tables);
        const uint32_t words = (p_bytes + s_bytes + f_bytes) / 4;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0u;
        lds_sync();
        return true;
    }

    __device__ __forceinline__ void add_entry(uint32_t idx, float val, bool fail) {
        const uint32_t w = idx * G + (uint32_t)c;
        atomicAdd(&Pt[w], val);
        if (fail) atomicAdd(&Ft[w >> 1], 1u << (16 * (w & 1)));
    }

    // ---- one batch of node pairs (u in group 1, v in group 2, level(u) < level(v)) against every entry (a, b)
    __device__ void pair_batch(int b0, int cnt, uint64_t cand1, int t1, uint64_t cand2, int t2) {
        const bool act = s < cnt;
        const int pr = X.pairbuf[b0 + (act ? s : 0)];
        const int u = pr & 255, v = pr >> 8;
        const int i = X.nodelevel[u], j = X.nodelevel[v];
        const float d = dist(u, v);
        const int kI = __popcll(cand1), kJ = __popcll(cand2);
        const uint32_t ebase = X.rowbase[i] + (uint32_t)kI * ((uint32_t)X.ksum[j] - (uint32_t)X.ksum[i + 1]);
        const uint64_t T1 = uni64(tnodes[t1]), T2 = uni64(tnodes[t2]);
        const bool on = act && lane_live;
        ++n_batches;
        uint64_t left = cand1;
        for (int x0 = 0; x0 < kI; x0 += kXC) {
            // the row-side clusters of this pass
            uint64_t cn[kXC];
            int na[kXC];
            float wa[kXC];
            uint64_t rows = 0;
            int kc = 0;
#pragma unroll
            for (int x = 0; x < kXC; ++x) {
                cn[x] = 0;
                na[x] = 0;
                wa[x] = 0.f;
                if (left) {
                    const int a = __ffsll((unsigned long long)left) - 1;
                    left &= left - 1;
                    cn[x] = uni64(cnodes[a]) & T1;
                    na[x] = __popcll(cn[x]);
                    wa[x] = p.wsum[a * 128 + t1];
                    rows |= cn[x];
                    kc = x + 1;
                }
            }
            if (!rows) continue;
            int y = 0;
            for (uint64_t bm = cand2; bm; bm &= bm - 1, ++y) {
                const int b = __ffsll((unsigned long long)bm) - 1;
                const uint64_t B = uni64(cnodes[b]) & T2;
                if (!B) continue;
                const int nb = __popcll(B);
                const float wb = p.wsum[b * 128 + t2];
                float acc[kXC];
                unsigned np[kXC];
#pragma unroll
                for (int x = 0; x < kXC; ++x) {
                    acc[x] = 0.f;
                    np[x] = 0u;
                }
                for (uint64_t mm = rows; mm; mm &= mm - 1) {
                    const int m = __ffsll((unsigned long long)mm) - 1;
                    float racc = 0.f;
                    unsigned rnp = 0u;
                    row_terms<true>(tab + m * Nm, B, d, racc, rnp);
                    n_terms += (unsigned)nb;
#pragma unroll
                    for (int x = 0; x < kXC; ++x) {
                        if ((cn[x] >> m) & 1) {
                            acc[x] = acc[x] + racc;
                            np[x] += rnp;
                        }
                    }
                }
#pragma unroll
                for (int x = 0; x < kXC; ++x) {
                    if (x < kc && na[x]) {
                        const int mn = na[x] * nb; // num_match (match_utils.py:34)
                        float val = acc[x] / (float)mn;
                        if (wa[x] * wb == 0.f) val = __builtin_nanf(""); // 1 / weights_sum (match_utils.py:51-52)
                        if (on) add_entry(ebase + (uint32_t)((x0 + x) * kJ + y), val, 2 * (int)np[x] < mn); // :61
                    }
                }
            }
        }
    }

    // ---- one batch of node pairs of the same ligand cluster (u < v): the diagonal entries (a, a) -> self table
    __device__ void self_batch(int b0, int cnt, uint64_t cand1, int t1, int t2) {
        const bool act = s < cnt;
        const int pr = X.pairbuf[b0 + (act ? s : 0)];
        const int u = pr & 255, v = pr >> 8;
        const int i = X.nodelevel[u];
        const float d = dist(u, v);
        const uint32_t sbase = X.ksum[i];
        const uint64_t T1 = uni64(tnodes[t1]), T2 = uni64(tnodes[t2]);
        const bool on = act && lane_live;
        ++n_batches;
        int x = 0;
        for (uint64_t am = cand1; am; am &= am - 1, ++x) {
            const int a = __ffsll((unsigned long long)am) - 1;
            const uint64_t ca = uni64(cnodes[a]);
            const uint64_t A = ca & T1, B = ca & T2;
            if (!A || !B) continue;
            float acc = 0.f;
            unsigned dummy = 0u;
            for (uint64_t mm = A; mm; mm &= mm - 1) {
                const int m = __ffsll((unsigned long long)mm) - 1;
                float racc = 0.f;
                row_terms<false>(tab + m * Nm, B, d, racc, dummy);
                acc = acc + racc;
            }
            const int mn = __popcll(A) * __popcll(B);
            n_terms += (unsigned)mn;
            float val = acc / (float)mn;
            if (p.wsum[a * 128 + t1] * p.wsum[a * 128 + t2] == 0.f) val = __builtin_nanf("");
            if (on) atomicAdd(&St[(sbase + (uint32_t)x) * G + (uint32_t)c], val);
        }
    }

    // cut the buffered pairs into batches; keep the incomplete last one unless `all`
    template <typename F>
    __device__ __forceinline__ void run_batches(int &npairs, bool all, F &&batch) {
        lds_sync();
        int done = 0;
        while (npairs - done >= NP || (all && npairs > done)) {
            const int cnt = (npairs - done) < NP ? (npairs - done) : NP;
            batch(done, cnt);
            done += cnt;
        }
        const int rest = npairs - done;
        if (done && rest) {
            const int tmp = lane < rest ? (int)X.pairbuf[done + lane] : 0;
            lds_sync();
            if (lane < rest) X.pairbuf[lane] = (uint16_t)tmp;
            lds_sync();
        }
        npairs = rest;
    }

    __device__ void build_tables() {
        const int mylevel = X.nodelevel[lane];
        const bool valid = mylevel != 0xff;
        const uint64_t mycand = valid ? X.cand[mylevel] : 0ull;
        const int candlo = (int)(uint32_t)mycand, candhi = (int)(uint32_t)(mycand >> 32);
        const int mytm = X.nodetm[lane];
        const unsigned long long validmask = __ballot(valid);
        const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull; // lanes below this one
        unsigned long long rem1 = validmask;
        while (rem1) {
            const int r1 = __ffsll(rem1) - 1;
            const int c1lo = lane_get(candlo, r1), c1hi = lane_get(candhi, r1), t1 = lane_get(mytm, r1);
            const unsigned long long g1 = __ballot(valid && candlo == c1lo && candhi == c1hi && mytm == t1);
            rem1 &= ~g1;
            const uint64_t cand1 = (uint64_t)(uint32_t)c1lo | ((uint64_t)(uint32_t)c1hi << 32);
            uint64_t nodes1 = 0;
            for (uint64_t cm = cand1; cm; cm &= cm - 1) nodes1 |= uni64(cnodes[__ffsll((unsigned long long)cm) - 1]);
            if (!(nodes1 & uni64(tnodes[t1]))) continue; // no model node is compatible with these ligand nodes (graph_match.py:148-155)
            unsigned long long rem2 = validmask;
            while (rem2) {
                const int r2 = __ffsll(rem2) - 1;
                const int c2lo = lane_get(candlo, r2), c2hi = lane_get(candhi, r2), t2 = lane_get(mytm, r2);
                const unsigned long long g2 = __ballot(valid && candlo == c2lo && candhi == c2hi && mytm == t2);
                rem2 &= ~g2;
                const uint64_t cand2 = (uint64_t)(uint32_t)c2lo | ((uint64_t)(uint32_t)c2hi << 32);
                uint64_t nodes2 = 0;
                for (uint64_t cm = cand2; cm; cm &= cm - 1) nodes2 |= uni64(cnodes[__ffsll((unsigned long long)cm) - 1]);
                if (!(nodes2 & uni64(tnodes[t2]))) continue;
                // pairs across ligand clusters: u in g1, v in g2 in a later cluster (match_utils.py:26-31 via graph_match.py:233-279)
                int npairs = 0;
                for (unsigned long long um = g1; um; um &= um - 1) {
                    const int u = __ffsll(um) - 1;
                    const int endu = uni((int)X.lend[lane_get(mylevel, u)]);
                    const unsigned long long vm = endu >= 64 ? 0ull : (g2 & (~0ull << endu));
                    if (!vm) continue;
                    if (npairs + 64 > kPairBuf) run_batches(npairs, false, [&](int b0, int cnt) { pair_batch(b0, cnt, cand1, t1, cand2, t2); });
                    if ((vm >> lane) & 1) X.pairbuf[npairs + __popcll(vm & lt)] = (uint16_t)(u | (lane << 8));
                    npairs += __popcll(vm);
                }
                if (npairs) run_batches(npairs, true, [&](int b0, int cnt) { pair_batch(b0, cnt, cand1, t1, cand2, t2); });
                // pairs inside one ligand cluster, u < v (match_utils.py:87, itertools.combinations)
                if (cand1 == cand2) {
                    npairs = 0;
                    for (unsigned long long um = g1; um; um &= um - 1) {
                        const int u = __ffsll(um) - 1;
                        const int endu = uni((int)X.lend[lane_get(mylevel, u)]);
                        const unsigned long long inlevel = (endu >= 64 ? ~0ull : ((1ull << endu) - 1ull)) & (u >= 63 ? 0ull : (~0ull << (u + 1)));
                        const unsigned long long vm = g2 & inlevel;
                        if (!vm) continue;
                        if (npairs + 64 > kPairBuf) run_batches(npairs, false, [&](int b0, int cnt) { self_batch(b0, cnt, cand1, t1, t2); });
                        if ((vm >> lane) & 1) X.pairbuf[npairs + __popcll(vm & lt)] = (uint16_t)(u | (lane << 8));
                        npairs += __popcll(vm);
                    }
                    if (npairs) run_batches(npairs, true, [&](int b0, int cnt) { self_batch(b0, cnt, cand1, t1, t2); });
                }
            }
        }
        lds_sync();
        finish_tables();
    }

    // ---- cluster-distance prefilter (graph_match.py:263-268) and the fail rule (match_utils.py:71-74) turn the
    // accumulators into the pair table: -1 where the entry is invalid.
    __device__ void finish_tables() {
        for (int i = 0; i < nl; ++i) {
            const int si = uni((int)X.lstart[i]), ei = uni((int)X.lend[i]), ki = uni((int)X.lk[i]);
            Pos ctr_i;
            float size_i;
            cluster_center_size(r.xyz, C, si, ei, cc, ctr_i, size_i);
            const uint64_t cand_i = uni64(X.cand[i]);
            for (int j = i + 1; j < nl; ++j) {
                const int sj = uni((int)X.lstart[j]), ej = uni((int)X.lend[j]), kj = uni((int)X.lk[j]);
                Pos ctr_j;
                float size_j;
                cluster_center_size(r.xyz, C, sj, ej, cc, ctr_j, size_j);
                const float ldist = norm3(ctr_i.x - ctr_j.x, ctr_i.y - ctr_j.y, ctr_i.z - ctr_j.z); // graph_match.py:240
                const float lsize = size_i + size_j;                                                 // :241
                const uint64_t cand_j = uni64(X.cand[j]);
                const uint32_t base = (uint32_t)uni((int)(X.rowbase[i] + (uint32_t)ki * ((uint32_t)X.ksum[j] - (uint32_t)X.ksum[i + 1])));
                const int E = ki * kj;
                const float inv_kj = 1.0f / (float)kj;
                for (int e0 = 0; e0 < E; e0 += NP) {
                    const int e = e0 + s;
                    const bool on = e < E;
                    bool near = false;
                    int L1 = 0, L2 = 0;
                    if (on) {
                        const int x = (int)(((float)e + 0.5f) * inv_kj), y = e - x * kj;
                        // the x-th / y-th candidate cluster
                        uint64_t am = cand_i, bm = cand_j;
                        for (int q = 0; q < x; ++q) am &= am - 1;
                        for (int q = 0; q < y; ++q) bm &= bm - 1;
                        const int a = __ffsll((unsigned long long)am) - 1, b = __ffsll((unsigned long long)bm) - 1;
                        const float2 mp = p.M.cpair[a * p.M.K + b];
                        near = lane_live && !((fabsf(ldist - mp.x) - lsize) > mp.y);
                        for (int u = si; u < ei; ++u) L1 += (cnodes[a] & tnodes[X.nodetm[u]]) ? 1 : 0; // graph_match.py:164-171
                        for (int v = sj; v < ej; ++v) L2 += (cnodes[b] & tnodes[X.nodetm[v]]) ? 1 : 0;
                    }
                    const unsigned long long bal = __ballot(near);
                    const unsigned long long slot = (G == 64) ? bal : ((bal >> (s * G)) & ((1ull << G) - 1ull));
                    if (on && lane_live) {
                        const uint32_t w = (base + (uint32_t)e) * G + (uint32_t)c;
                        const float score = Pt[w];
                        const int fails = (int)((Ft[w >> 1] >> (16 * (w & 1))) & 0xffffu);
                        float value = -1.f;
                        if (slot) value = (2 * fails <= L1 * L2) ? score : -1.f; // match_utils.py:22,71-74
                        Pt[w] = value;
                    }
                }
            }
        }
        lds_sync();
    }

    // ---- search bounds: R[f][c] = the most levels f.. can add to conformer c's total
    __device__ void build_bounds() {
        double suffix = 0.0;
        if (s == 0) Rt[(size_t)nl * G + c] = 0.0;
        for (int l = nl - 1; l >= 0; --l) {
            const int kl = uni((int)X.lk[l]), ksl = uni((int)X.ksum[l]);
            double u = 0.0;
            for (int b = s; b < kl; b += NP) {
                double v = (double)St[(size_t)(ksl + b) * G + c];
                for (int j = 0; j < l; ++j) {
                    const int kj = uni((int)X.lk[j]);
                    const uint32_t e0 = (uint32_t)uni((int)(X.rowbase[j] + (uint32_t)kj * (uint32_t)(ksl - (int)X.ksum[j + 1]))) + (uint32_t)b;
                    float m = 0.f;
                    for (int a = 0; a < kj; ++a) {
                        const float pv = Pt[(size_t)(e0 + (uint32_t)a * (uint32_t)kl) * G + c];
                        m = pv > m ? pv : m;
                    }
                    v += (double)m;
                }
                u = v > u ? v : u;
            }
#pragma unroll
            for (int dd = G; dd < 64; dd <<= 1) {
                const double o = shfl_xor_f64(u, dd);
                u = o > u ? o : u;
            }
            suffix += u;
            if (p.flags & 4u) suffix = __builtin_inf();
            if (s == 0) Rt[(size_t)l * G + c] = suffix;
        }
        lds_sync();
    }

    // ---- tree search: ClusterMatchTree.dfs_run (tree.py:55-104) with wave-uniform control.
    // Frame f = the tree node whose children are the candidates of level f. Lane-indexed registers hold the stack:
    // fr_* at lane f, mt_* (matched ancestors) at lane q, al_* (conformers alive) at lane = number of matches.
    static constexpr int F_MATCHED = 1 << 8, F_ANY = 1 << 9, F_SKIP = 1 << 10;

    __device__ float walk() {
        int fr_lo = 0, fr_hi = 0, fr_info = 0; // todo mask (64 bit), {mx:8, flags, nm << 16}
        int mt_base = 0, mt_ka = 0;            // rowbase[j] - k_j * ksum[j + 1], k_j | a << 8
        int al_lo = 0, al_hi = 0;
        double best = 0.0; // graph_match.py:104
        constexpr double kBoundSlack = 1.0 + 1e-9;
        const unsigned long long allc = (C >= 64) ? ~0ull : ((1ull << C) - 1ull);
        const int lane_off = lane; // float offset of (candidate s, conformer c) inside a batch of NP candidates

        // conformer-validity and (for leaf levels) totals of the candidates of level F for the current path
        auto eval_level = [&](int F, int nm, unsigned long long alive, bool leaves, const double ptot) -> unsigned long long {
            const int kF = uni((int)X.lk[F]), ksF = uni((int)X.ksum[F]);
            const int ebv = mt_base + (mt_ka & 255) * ksF + ((mt_ka >> 8) & 255) * kF; // lane q: ancestor q's row for level F
            unsigned long long E = 0;
            for (int b0 = 0; b0 < kF; b0 += NP) {
                const bool on = b0 + s < kF;
                bool ok = on && ((alive >> c) & 1);
                const int off = (on ? b0 * G + lane_off : c);
                double pair = 0.0;
                int q = 0;
                for (; q + 4 <= nm; q += 4) {
                    float v[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) v[w] = Pt[lane_get(ebv, q + w) * G + off];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        ok = ok && (v[w] > 0.f); // tree.py:81
                        pair += (double)v[w];
                    }
                }
                for (; q < nm; ++q) {
                    const float v0 = Pt[lane_get(ebv, q) * G + off];
                    ok = ok && (v0 > 0.f);
                    pair += (double)v0;
                }
                if (leaves) { // per-conformer maximum over leaves (graph_match.py:105-108)
                    const double t = ptot + (double)St[ksF * G + off] + pair; // tree.py:38-41
                    if (ok && t > best) best = t;
                }
                const unsigned long long bal = __ballot(ok);
                unsigned long long part;
                if (G == 1) {
                    part = bal;
                } else if (G == 64) {
                    part = bal ? 1ull : 0ull;
                } else {
                    const bool any = lane < NP && ((bal >> (lane * G)) & ((1ull << G) - 1ull)) != 0;
                    part = __ballot(any);
                }
                E |= part << b0;
            }
            return E;
        };
        auto pool_best = [&]() {
#pragma unroll
            for (int dd = G; dd < 64; dd <<= 1) {
                const double o = shfl_xor_f64(best, dd);
                best = o > best ? o : best;
            }
        };
        // a frame whose children are leaves is finished inside its evaluation; returns its return value (tree.py:102)
        auto leaf_frame = [&](int F, int nm, unsigned long long alive, bool matched) -> int {
            const double ptot = Tt[(size_t)nm * G + c];
            const double before = best;
            const unsigned long long E = eval_level(F, nm, alive, true, ptot);
            const int mx = E ? 1 : 0;
            if (!E || nm + mx < 5) { // the skip leaf carries this node's totals (tree.py:98-101, :42-43)
                if (((alive >> c) & 1) && ptot > best) best = ptot;
            }
            if (NP > 1 && __ballot(best > before)) pool_best();
            return mx + (matched ? 1 : 0);
        };

        // root
        lane_set(al_lo, 0, (int)(uint32_t)allc);
        lane_set(al_hi, 0, (int)(uint32_t)(allc >> 32));
        if (s == 0) Tt[c] = 0.0;
        lds_sync();
        int f = 0;
        if (nl == 1) {
            (void)leaf_frame(0, 0, allc, false);
        } else {
            const unsigned long long E0 = eval_level(0, 0, allc, false, 0.0);
            lane_set(fr_lo, 0, (int)(uint32_t)E0);
            lane_set(fr_hi, 0, (int)(uint32_t)(E0 >> 32));
            lane_set(fr_info, 0, E0 ? F_ANY : 0);
            for (;;) {
                ++n_steps;
                int info = lane_get(fr_info, f);
                const unsigned long long todo = (unsigned long long)(uint32_t)lane_get(fr_lo, f) | ((unsigned long long)(uint32_t)lane_get(fr_hi, f) << 32);
                const int nm = (info >> 16) & 255;
                const unsigned long long alive =
                    (unsigned long long)(uint32_t)lane_get(al_lo, nm) | ((unsigned long long)(uint32_t)lane_get(al_hi, nm) << 32);
                if (todo) { // next existing candidate child (tree.py:94-97)
                    const int b = __ffsll(todo) - 1;
                    const unsigned long long left = todo & (todo - 1);
                    lane_set(fr_lo, f, (int)(uint32_t)left);
                    lane_set(fr_hi, f, (int)(uint32_t)(left >> 32));
                    // total and conformer mask of the child: parent + self + accumulated pair (tree.py:38-41, :78-82)
                    const int kf = uni((int)X.lk[f]), ksf = uni((int)X.ksum[f]);
                    const int ebv = mt_base + (mt_ka & 255) * ksf + ((mt_ka >> 8) & 255) * kf;
                    const int off = b * G + c;
                    bool ok = (alive >> c) & 1;
                    double pair = 0.0;
                    int q = 0;
                    for (; q + 4 <= nm; q += 4) {
                        float v[4];
#pragma unroll
                        for (int w = 0; w < 4; ++w) v[w] = Pt[lane_get(ebv, q + w) * G + off];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            ok = ok && (v[w] > 0.f);
                            pair += (double)v[w];
                        }
                    }
                    for (; q < nm; ++q) {
                        const float v0 = Pt[lane_get(ebv, q) * G + off];
                        ok = ok && (v0 > 0.f);
                        pair += (double)v0;
                    }
                    const double t = Tt[(size_t)nm * G + c] + (double)St[ksf * G + off] + pair;
                    const unsigned long long bal = __ballot(ok);
                    const unsigned long long cmask = (G == 64) ? bal : (bal & ((1ull << G) - 1ull));
                    if (nm >= 4) { // the child holds >= 5 matches: dropping its subtree cannot change a skip decision
                        const double rb = Rt[(size_t)(f + 1) * G + c];
                        if (!__ballot(ok && (t + rb) * kBoundSlack > best)) { // no leaf below can exceed the maxima found so far
                            const int mx = info & 255;
                            if (mx < 1) lane_set(fr_info, f, (info & ~255) | 1);
                            continue;
                        }
                    }
                    // descend
                    if (s == 0) Tt[(size_t)(nm + 1) * G + c] = t;
                    lane_set(al_lo, nm + 1, (int)(uint32_t)cmask);
                    lane_set(al_hi, nm + 1, (int)(uint32_t)(cmask >> 32));
                    lane_set(mt_base, nm, uni((int)X.rowbase[f] - kf * (int)X.ksum[f + 1]));
                    lane_set(mt_ka, nm, kf | (b << 8));
                    lds_sync();
                    const int F = f + 1;
                    if (F == nl - 1) {
                        const int ret = leaf_frame(F, nm + 1, cmask, true);
                        const int mx = info & 255;
                        if (ret > mx) lane_set(fr_info, f, (info & ~255) | ret);
                    } else {
                        const unsigned long long E = eval_level(F, nm + 1, cmask, false, 0.0);
                        lane_set(fr_lo, F, (int)(uint32_t)E);
                        lane_set(fr_hi, F, (int)(uint32_t)(E >> 32));
                        lane_set(fr_info, F, ((nm + 1) << 16) | F_MATCHED | (E ? F_ANY : 0));
                        f = F;
                    }
                    continue;
                }
                const int mx = info & 255;
                if (!(info & F_SKIP) && (!(info & F_ANY) || nm + mx < 5)) { // skip child (tree.py:98-101)
                    info |= F_SKIP;
                    lane_set(fr_info, f, info);
                    const int F = f + 1;
                    if (F == nl - 1) {
                        const int ret = leaf_frame(F, nm, alive, false);
                        if (ret > mx) lane_set(fr_info, f, (info & ~255) | ret);
                    } else {
                        const unsigned long long E = eval_level(F, nm, alive, false, 0.0);
                        lane_set(fr_lo, F, (int)(uint32_t)E);
                        lane_set(fr_hi, F, (int)(uint32_t)(E >> 32));
                        lane_set(fr_info, F, (nm << 16) | (E ? F_ANY : 0));
                        f = F;
                    }
                    continue;
                }
                // all children done: return max_num_matches + matched (tree.py:102)
                const int ret = mx + ((info & F_MATCHED) ? 1 : 0);
                if (f == 0) break;
                --f;
                const int pinfo = lane_get(fr_info, f);
                if (ret > (pinfo & 255)) lane_set(fr_info, f, (pinfo & ~255) | ret);
            }
        }
        pool_best();
        // mean over conformers (graph_match.py:109); lanes of dead conformers hold 0
        double sum = lane_live ? best : 0.0;
#pragma unroll
        for (int dd = 1; dd < G; dd <<= 1) sum += shfl_xor_f64(sum, dd);
        return (float)(sum / (double)C);
    }
};

// One block = W wavefronts sharing the staged model; every wavefront pulls ligands of this size class until none are left.
template <int G, bool LDS_TABLES>
__global__ __launch_bounds__(1024) void match_kernel(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Nm = p.M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + round16(uint64_t(Nm) * Nm * 16));
    uint64_t *tnodes = cnodes + 64;
    for (int i = threadIdx.x; i < Nm * Nm; i += blockDim.x) tab[i] = p.wtab[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = p.M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = p.M.tnodes[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *ctx = reinterpret_cast<unsigned char *>(tnodes + 128) + (size_t)wave * (LDS_TABLES ? p.wave_bytes : (uint32_t)sizeof(MatchCtx));
    unsigned char *tables = ctx + sizeof(MatchCtx);
    if (!LDS_TABLES) { // this wave's slot of the HBM arena, sized for the call's largest table
        const uint32_t need = (uint32_t)round16(uni((int)p.bins->max_need));
        if (need == 0) return;
        const uint64_t nslots = p.arena_bytes / need;
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(&p.bins->big_slot_cursor, 1u);
        slot = (uint32_t)uni((int)slot);
        if (slot >= nslots) return;
        tables = p.arena + (size_t)slot * need;
    }
    const uint32_t count = (uint32_t)uni((int)p.bins->count[p.bin]);
    Matcher<G, LDS_TABLES> mt(p, tab, cnodes, tnodes, ctx);
    for (;;) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&p.bins->cursor[p.bin], 1u);
        pos = (uint32_t)uni((int)pos);
        if (pos >= count) break;
        const uint32_t li = (uint32_t)uni((int)p.list[pos]);
        if (!mt.setup(li, tables)) continue;
        mt.build_tables();
        mt.build_bounds();
        const float score = mt.walk();
        if (lane == 0) p.scores[li] = score;
    }
    if (lane == 0) {
        atomicAdd(&p.stats[0], mt.n_steps);
        atomicAdd(&p.stats[1], mt.n_batches);
        atomicAdd(&p.stats[2], mt.n_terms);
    }
}

} // namespace pmx

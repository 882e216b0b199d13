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

constexpr int kPairBuf = 128;     // node pairs buffered before they are cut into batches
constexpr int kListSlots = 16;    // column lists staged in LDS per column group (more candidate clusters: read from HBM)
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
    uint32_t meta[8];                  // coop: nl, T, ksumtot written by the building wave for the others
    uint4 lists[2 * kListSlots];       // column offset lists (DevModel::olist) of the current column group's clusters
};
static_assert(sizeof(MatchCtx) == 768 + 32 * kListSlots, "MatchCtx layout");

// LDS bytes one ligand needs: header, P [T][G] f32, S [ksumtot][G] f32, then the larger of the table phase's scratch
// (fail counts u16 [T][G] + cluster geometry float4 [nl][G]) and the tree phase's (bounds and path totals, 2 x f64
// [nl + 1][G], and the lookahead sums f64 [ksumtot][G]).
template <int G>
__host__ __device__ inline uint32_t match_bytes(uint32_t T, uint32_t ksumtot, uint32_t nl) {
    const uint32_t p = (uint32_t)round16(uint64_t(T) * G * 4), s = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
    const uint32_t f = (uint32_t)round16(uint64_t(T) * G * 2) + nl * G * 16, w = (2u * (nl + 1) + ksumtot) * G * 8;
    return (uint32_t)sizeof(MatchCtx) + p + s + (f > w ? f : w);
}

// LDS bytes of the staged model: folded edge table, cluster node sets, type -> node sets, cluster-pair prefilter table.
__host__ __device__ inline uint32_t model_lds_bytes(int Nm, int K) {
    return (uint32_t)round16(uint64_t(Nm) * Nm * 16) + 64 * 8 + 128 * 8 + (uint32_t)round16(uint64_t(K) * K * 8);
}

struct BinInfo {          // device-resident, written by bin_kernel, read by the match kernels
    uint32_t count[kNumBins + 1];   // ligands scored by one wavefront each (match_kernel)
    uint32_t cursor[kNumBins + 1];
    uint32_t hcount[kNumBins + 1];  // ligands scored by a whole block (coop_kernel): large tables, and trees that ran over budget
    uint32_t hcursor[kNumBins + 1];
    uint32_t cap[kNumBins];   // LDS bytes per ligand of each class
    uint32_t max_need;        // largest table of the call (bytes)
    uint32_t coop_from;       // classes >= this go to the cooperative kernel straight away
};

// A subtree handed from the wave that walks the top of a tree to the block's other waves: the path of matches down to its
// root (a node with >= 5 matches, so that its return value is settled: DESIGN.md section 3), followed by double tot[G].
struct RootHeader {
    uint8_t f;      // level of the root's own match; its frame is f + 1
    uint8_t nm;     // matches on the path, root included
    uint8_t pad[6];
    uint64_t mask;  // conformers alive at the root
    uint8_t path[2 * PMX_MAX_LEVELS]; // (level, candidate) of every match on the path
    uint8_t pad2[8];
};
static_assert(sizeof(RootHeader) == 64, "RootHeader layout");
template <int G>
__host__ __device__ constexpr uint32_t root_bytes() {
    return sizeof(RootHeader) + G * 8;
}

struct CoopShared { // per block, in LDS
    unsigned long long best[64]; // per-conformer maxima pooled over the block's waves (bit patterns of non-negative doubles)
    uint32_t item;      // the block's current list position
    uint32_t nroots;
    uint32_t next_root;
    uint32_t has_tree;
    uint32_t overflow;  // roots that did not fit the buffer were walked in place
    uint32_t pad[3];
};

struct MatchParams {
    DevModel M;
    const float4 *wtab; // [Nm * Nm] {mean, s, T, w_m w_n / std}: the call's weights folded in (match_utils.py:65)
    uint64_t nzw;       // model nodes whose type has a non-zero weight (a node pair list without any gives 1 / 0: match_utils.py:51)
    DevLibrary lib;
    uint64_t first;      // library index of the call's first ligand
    const uint32_t *list; // this class's ligands (indices into the call's range)
    BinInfo *bins;
    uint32_t bin;        // class index
    uint32_t wave_bytes; // LDS bytes per ligand (class cap); 0 for the HBM class
    uint8_t *arena;      // HBM class: table memory, cut into equal slots
    uint64_t arena_bytes;
    uint32_t *hlist;     // this class's cooperative list (bins->hcount)
    uint8_t *roots;      // coop_kernel: subtree roots, one segment per block
    uint32_t roots_cap;  // roots per block
    uint32_t budget;     // match_kernel: tree steps after which a ligand is passed on to coop_kernel
    uint32_t pool_bytes; // coop_kernel: LDS behind the tables for the helper waves' walk state
    const uint64_t *taboff; // tables_kernel_v3: arena offsets of the chunk's table blocks (scan_kernel)
    uint8_t *out_arena;
    double *seed;        // experiment (PMX_SEED_BEST): per-conformer maxima of a previous pass [count][G]; mode 1 = record, 2 = seed
    int seed_mode;
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

// Where the folded edge table is read from: an LDS copy (broadcast reads) or, SMEM = true, HBM through the scalar cache
// (every address is wave-uniform, so a term's parameters arrive in SGPRs and the block needs no LDS for the model).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) *const_f4v_ptr;
template <bool SMEM>
__device__ __forceinline__ float4 edge_at(const float4 *tab, int byte_offset) {
    if (SMEM) {
        const f4v e = reinterpret_cast<const_f4v_ptr>(reinterpret_cast<unsigned long long>(tab))[byte_offset >> 4];
        return make_float4(e.x, e.y, e.z, e.w);
    }
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(tab) + byte_offset);
}

// One Gaussian term (match_utils.py:55-68): e = {mean, s, T, w / std}
template <bool PASS>
__device__ __forceinline__ void gterm(const float d, const float4 e, float &acc, unsigned &np) {
    const float t = __builtin_fabsf(d - e.x);
    const float q = t * e.y;
    acc = __builtin_fmaf(e.w, __builtin_amdgcn_exp2f(-(q * q)), acc);
    if (PASS) np += (t <= e.z) ? 1u : 0u;
}

// The terms of one row (byte offset `row` into the edge table) against the model nodes of column set B (ascending n), all
// wave-uniform; B non-empty. Fallback for column sets of more than 12 nodes.
template <bool PASS, bool SMEM>
__device__ __forceinline__ void row_terms(const float4 *tab, int row, uint64_t B, const float d, float &acc, unsigned &np) {
    for (; B; B &= B - 1) gterm<PASS>(d, edge_at<SMEM>(tab, row + 16 * (__ffsll((unsigned long long)B) - 1)), acc, np);
}

// ------------------------------------------------------------------------------------------------ bin_kernel
// One thread per ligand: levels and table sizes (graph_match.py:124-137, :87-88) -> LDS size class; ligands without a
// tree get their score here (0: graph_match.py:95-99; NaN: record outside the structural limits).
template <int G>
__global__ void bin_kernel(DevLibrary lib, const uint64_t *tclus, uint64_t first, uint32_t count, BinInfo *bins, uint32_t *lists,
                           uint32_t *hlists, int32_t *status, float *scores) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Record r = parse_record(lib.data + lib.offsets[first + i]);
    if (!record_supported(r)) {
        if (status) status[i] = PMX_LIGAND_UNSUPPORTED;
        if (scores) scores[i] = __builtin_nanf("");
        return;
    }
    if (status) status[i] = PMX_LIGAND_OK;
    const Levels L = scan_levels(r, tclus, [](int, int, int, uint64_t, uint32_t) {});
    if (L.nl == 0) {
        if (scores) scores[i] = 0.f;
        return;
    }
    const uint32_t need = match_bytes<G>(L.T, L.ksumtot, (uint32_t)L.nl);
    uint32_t b = 0;
    while (b < (uint32_t)kNumBins && need > bins->cap[b]) ++b;
    if (b >= bins->coop_from) {
        const uint32_t pos = atomicAdd(&bins->hcount[b], 1u);
        hlists[(size_t)b * count + pos] = i;
    } else {
        const uint32_t pos = atomicAdd(&bins->count[b], 1u);
        lists[(size_t)b * count + pos] = i;
    }
    atomicMax(&bins->max_need, need);
}

__global__ void bins_init_kernel(BinInfo *bins, const uint32_t *caps, uint32_t coop_from, unsigned long long *stats) {
    const int t = threadIdx.x;
    if (t <= kNumBins) {
        bins->count[t] = 0;
        bins->cursor[t] = 0;
        bins->hcount[t] = 0;
        bins->hcursor[t] = 0;
    }
    if (t < kNumBins) bins->cap[t] = caps[t];
    if (t == 0) {
        bins->max_need = 0;
        bins->coop_from = coop_from;
    }
    for (int i = t; i < 128; i += 64) stats[i] = 0;
}

// The call's weights folded into the edge table, and the weight sums the reference normalises with.
__global__ void fold_weights_kernel(DevModel M, Weights W, float4 *wtab) {
    const int Nm = M.Nm;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Nm * Nm) {
        const int m = t / Nm, n = t - m * Nm;
        float4 e = M.edge[t];
        e.w = (W.w[M.node_type[m]] * W.w[M.node_type[n]]) / e.w; // weights / stds (match_utils.py:65)
        wtab[t] = e;
    }
}

// --------------------------------------------------------------------------------------------- the matcher
template <int G, bool LDS_TABLES, bool SMEM = false>
struct Matcher {
    static constexpr int NP = 64 / G; // node pairs (table phase) / candidates (tree phase) per pass of the wave
    const MatchParams &p;
    const float4 *tab;      // LDS: folded edge table [Nm][Nm]
    const uint64_t *cnodes; // LDS [64]
    const uint64_t *tnodes; // LDS [128]
    const float2 *cpair;    // LDS [K * K]
    int ctm;                // lane a: type mask of model cluster a (graph_match.py:130-134)
    MatchCtx &X;
    float *Pt;              // [T][G]
    float *St;              // [ksumtot][G]
    uint32_t *Ft;           // table phase: fail counts, two u16 per word
    float4 *Gt;             // table phase: {center, size} of each level's ligand cluster per conformer [nl][G]
    double *Rt;             // tree phase: bounds [(nl + 1)][G]   (overlays Ft)
    double *Tt;             // tree phase: path totals by match count [(nl + 1)][G]
    double *La;             // tree phase: lookahead sums [ksumtot][G] (follows Tt)
    const int lane, s, c;
    Record r;
    int C, cc, nl;
    bool lane_live;
    uint32_t T, ksumtot;
    int Nm;
    uint16_t *pairbuf;      // this wave's pair buffer and column lists (its own copies when several waves build one ligand)
    uint4 *lists;
    uint32_t cur_li = 0;
    int bw = 0, bn = 1;     // this wave takes the table entries (x, y) with (x * kJ + y) % bn == bw
    unsigned long long n_steps = 0, n_batches = 0, n_terms = 0, n_top = 0, cyc_tab = 0, cyc_walk = 0, cyc_setup = 0, cyc_batch = 0, cyc_finish = 0, cyc_bounds = 0;

    __device__ Matcher(const MatchParams &p_, const float4 *tab_, const uint64_t *cn_, const uint64_t *tn_, const float2 *cp_, unsigned char *ctx)
        : p(p_), tab(tab_), cnodes(cn_), tnodes(tn_), cpair(cp_), X(*reinterpret_cast<MatchCtx *>(ctx)), lane(threadIdx.x & 63),
          s((threadIdx.x & 63) / G), c((threadIdx.x & 63) % G) {
        // cluster a has type t <=> a is a candidate of the single-type mask 1 << t
        int m = 0;
        for (int t = 0; t < PMX_NUM_TYPES; ++t) m |= (int)((p_.M.tclus[1 << t] >> lane) & 1ull) << t;
        ctm = lane < p_.M.K ? m : 0;
        pairbuf = X.pairbuf;
        lists = X.lists;
    }

    __device__ __forceinline__ void lds_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    __device__ __forceinline__ float dist(int u, int v) const {
        const float *pu = r.xyz + (size_t)(u * 3) * C + cc, *pv = r.xyz + (size_t)(v * 3) * C + cc;
        return norm3(pu[0] - pv[0], pu[C] - pv[C], pu[2 * C] - pv[2 * C]);
    }

    // ---- levels, node tables, cleared accumulators. Returns false when the ligand has no tree.
    // The record's type masks and cluster ends are fetched with one coalesced load into lane-indexed registers; the
    // level scan (graph_match.py:124-137, :87-88) then runs on registers.
    __device__ bool setup(uint32_t li, unsigned char *tables) {
        r = parse_record(p.lib.data + p.lib.offsets[p.first + li]);
        cur_li = li;
        r.n = uni(r.n);
        r.C = uni(r.C);
        r.ncl = uni(r.ncl);
        C = r.C;
        cc = c < C ? c : C - 1;
        lane_live = c < C;
        Nm = p.M.Nm;
        const int mytm = lane < r.n ? (int)r.typemask[lane] : 0;
        const int myend = lane < r.ncl ? (int)r.cluster_end[lane] : 0;
        int mylevel = 0xff;
        {
            uint32_t sumk = 0, sumk2 = 0;
            int start = 0, lev = 0;
            for (int ci = 0; ci < r.ncl && lev < PMX_MAX_LEVELS; ++ci) {
                const int end = lane_get(myend, ci);
                const unsigned long long members = __ballot(lane >= start && lane < end);
                int lmask = 0;
                for (unsigned long long mm = members; mm; mm &= mm - 1) lmask |= lane_get(mytm, __ffsll(mm) - 1);
                const uint64_t cand = __ballot((ctm & lmask) != 0);
                if (cand) {
                    const uint32_t k = (uint32_t)__popcll(cand);
                    if (lane == 0) {
                        X.cand[lev] = cand;
                        X.lstart[lev] = (uint8_t)start;
                        X.lend[lev] = (uint8_t)end;
                        X.lk[lev] = (uint8_t)k;
                        X.ksum[lev] = (uint16_t)sumk;
                    }
                    if ((members >> lane) & 1) mylevel = lev;
                    sumk += k;
                    sumk2 += k * k;
                    ++lev;
                }
                start = end;
            }
            nl = lev;
            ksumtot = sumk;
            T = (sumk * sumk - sumk2) / 2; // sum_{i<j} k_i k_j
            if (nl == 0) return false;
            if (lane == 0) X.ksum[nl] = (uint16_t)sumk;
        }
        X.nodelevel[lane] = (uint8_t)mylevel;
        X.nodetm[lane] = (uint8_t)mytm;
        lds_sync();
        if (lane < nl) { // first pair entry of each level: sum over earlier levels of k_i * (ksumtot - ksum[i + 1])
            uint32_t rb = 0;
            for (int i = 0; i < lane; ++i) rb += (uint32_t)X.lk[i] * (ksumtot - (uint32_t)X.ksum[i + 1]);
            X.rowbase[lane] = rb;
        }
        const uint32_t p_bytes = (uint32_t)round16(uint64_t(T) * G * 4), s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
        const uint32_t f_bytes = (uint32_t)round16(uint64_t(T) * G * 2);
        Pt = reinterpret_cast<float *>(tables);
        St = reinterpret_cast<float *>(tables + p_bytes);
        Ft = reinterpret_cast<uint32_t *>(tables + p_bytes + s_bytes);
        Gt = reinterpret_cast<float4 *>(tables + p_bytes + s_bytes + f_bytes);
        Rt = reinterpret_cast<double *>(tables + p_bytes + s_bytes);
        Tt = Rt + (size_t)(nl + 1) * G;
        La = Tt + (size_t)(nl + 1) * G;
        uint32_t *z = reinterpret_cast<uint32_t *>(tables);
        const uint32_t words = (p_bytes + s_bytes + f_bytes) / 4;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0u;
        lds_sync();
        // cluster geometry of every level (ligand.py:458-473), NP levels at a time
        for (int l0 = 0; l0 < nl; l0 += NP) {
            const int l = l0 + s;
            if (l < nl) {
                Pos ctr;
                float size;
                cluster_center_size(r.xyz, C, X.lstart[l], X.lend[l], cc, ctr, size);
                Gt[(size_t)l * G + c] = make_float4(ctr.x, ctr.y, ctr.z, size);
            }
        }
        lds_sync();
        return true;
    }

    // The finished tables as one block of the chunk's arena, in the layout the tree kernels read (pmx_device.h, TabHeader).
    __device__ void write_out(uint8_t *blk) {
        TabHeader *H = reinterpret_cast<TabHeader *>(blk);
        if (lane == 0) {
            H->nl = (uint32_t)nl;
            H->T = T;
            H->ksumtot = ksumtot;
            H->pad = 0;
        }
        if (lane < nl) {
            H->k[lane] = X.lk[lane];
            H->rowbase[lane] = X.rowbase[lane];
        }
        if (lane <= nl) H->ksum[lane] = X.ksum[lane];
        const uint32_t v_bytes = (uint32_t)round16(uint64_t(T) * sizeof(vmask_t<G>));
        const uint32_t s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4), p_bytes = (uint32_t)round16(uint64_t(T) * G * 4);
        vmask_t<G> *Vo = reinterpret_cast<vmask_t<G> *>(blk + sizeof(TabHeader));
        for (uint32_t e = lane; e < T; e += 64) { // conformer-validity mask of each pair entry (tree.py:81)
            unsigned long long m = 0;
            for (int q = 0; q < C; ++q) m |= (unsigned long long)(Pt[e * G + q] > 0.f) << q;
            Vo[e] = (vmask_t<G>)m;
        }
        uint4 *So = reinterpret_cast<uint4 *>(blk + sizeof(TabHeader) + v_bytes);
        const uint4 *Si = reinterpret_cast<const uint4 *>(St);
        for (uint32_t i = lane; i < s_bytes / 16; i += 64) So[i] = Si[i];
        uint4 *Po = reinterpret_cast<uint4 *>(blk + sizeof(TabHeader) + v_bytes + s_bytes);
        const uint4 *Pi = reinterpret_cast<const uint4 *>(Pt);
        for (uint32_t i = lane; i < p_bytes / 16; i += 64) Po[i] = Pi[i];
        double *Ro = reinterpret_cast<double *>(blk + sizeof(TabHeader) + v_bytes + s_bytes + p_bytes);
        for (int i = lane; i < (nl + 1) * G; i += 64) Ro[i] = Rt[i];
    }

    // The same by a team of waves: wave `tw` of `tn` copies every tn-th chunk of V and P; wave 0 also the header, S and R.
    __device__ void write_out_team(uint8_t *blk, int tw, int tn) {
        const uint32_t v_bytes = (uint32_t)round16(uint64_t(T) * sizeof(vmask_t<G>));
        const uint32_t s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4), p_bytes = (uint32_t)round16(uint64_t(T) * G * 4);
        vmask_t<G> *Vo = reinterpret_cast<vmask_t<G> *>(blk + sizeof(TabHeader));
        for (uint32_t e = tw * 64 + lane; e < T; e += 64 * tn) {
            unsigned long long m = 0;
            for (int q = 0; q < C; ++q) m |= (unsigned long long)(Pt[e * G + q] > 0.f) << q;
            Vo[e] = (vmask_t<G>)m;
        }
        uint4 *Po = reinterpret_cast<uint4 *>(blk + sizeof(TabHeader) + v_bytes + s_bytes);
        const uint4 *Pi = reinterpret_cast<const uint4 *>(Pt);
        for (uint32_t i = tw * 64 + lane; i < p_bytes / 16; i += 64 * tn) Po[i] = Pi[i];
        if (tw != 0) return;
        TabHeader *H = reinterpret_cast<TabHeader *>(blk);
        if (lane == 0) {
            H->nl = (uint32_t)nl;
            H->T = T;
            H->ksumtot = ksumtot;
            H->pad = 0;
        }
        if (lane < nl) {
            H->k[lane] = X.lk[lane];
            H->rowbase[lane] = X.rowbase[lane];
        }
        if (lane <= nl) H->ksum[lane] = X.ksum[lane];
        uint4 *So = reinterpret_cast<uint4 *>(blk + sizeof(TabHeader) + v_bytes);
        const uint4 *Si = reinterpret_cast<const uint4 *>(St);
        for (uint32_t i = lane; i < s_bytes / 16; i += 64) So[i] = Si[i];
        double *Ro = reinterpret_cast<double *>(blk + sizeof(TabHeader) + v_bytes + s_bytes + p_bytes);
        for (int i = lane; i < (nl + 1) * G; i += 64) Ro[i] = Rt[i];
    }

    // coop_kernel: a wave that did not build the tables takes their geometry from the wave that did
    __device__ void attach(uint32_t li, unsigned char *tables, double *own_tt) {
        r = parse_record(p.lib.data + p.lib.offsets[p.first + li]);
        r.n = uni(r.n);
        r.C = uni(r.C);
        r.ncl = uni(r.ncl);
        C = r.C;
        cc = c < C ? c : C - 1;
        lane_live = c < C;
        Nm = p.M.Nm;
        nl = uni((int)X.meta[0]);
        T = (uint32_t)uni((int)X.meta[1]);
        ksumtot = (uint32_t)uni((int)X.meta[2]);
        const uint32_t p_bytes = (uint32_t)round16(uint64_t(T) * G * 4), s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
        Pt = reinterpret_cast<float *>(tables);
        St = reinterpret_cast<float *>(tables + p_bytes);
        Ft = reinterpret_cast<uint32_t *>(tables + p_bytes + s_bytes);
        Gt = reinterpret_cast<float4 *>(tables + p_bytes + s_bytes + (uint32_t)round16(uint64_t(T) * G * 2));
        Rt = reinterpret_cast<double *>(tables + p_bytes + s_bytes);
        Tt = own_tt;
        La = Tt + (size_t)(nl + 1) * G;
    }

    __device__ __forceinline__ void add_entry(uint32_t idx, float val, bool fail) {
        const uint32_t w = idx * G + (uint32_t)c;
        atomicAdd(&Pt[w], val);
        if (fail) atomicAdd(&Ft[w >> 1], 1u << (16 * (w & 1)));
    }

    // The terms of rows A x columns B (both ascending) for this lane's distance, B given as the model's precomputed
    // list of byte offsets into a row of the staged edge table (DevModel::olist): everything is wave-uniform, the
    // column count is a compile-time constant, so a row is a straight line of NB broadcast reads and NB terms.
    template <int NB, bool PASS>
    __device__ __forceinline__ void rows_fixed(uint64_t A, const uint4 l0, const uint4 l1, const float d, float &acc, unsigned &np) const {
        const uint32_t w[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        int off[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) off[t] = (int)(((t + 1) & 1) ? (w[(t + 1) >> 1] >> 16) : (w[(t + 1) >> 1] & 0xffffu));
        const int rowbytes = Nm * 16;
        for (uint64_t mm = A; mm; mm &= mm - 1) {
            const int row = (__ffsll((unsigned long long)mm) - 1) * rowbytes;
#pragma unroll
            for (int t0 = 0; t0 < NB; t0 += 4) {
                float4 e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t0 + q < NB) e[q] = edge_at<SMEM>(tab, row + off[t0 + q]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t0 + q < NB) gterm<PASS>(d, e[q], acc, np);
            }
        }
    }
    // `slot` = position of the column cluster among its level's candidates (its list is staged in LDS when < kListSlots)
    template <bool PASS>
    __device__ __forceinline__ void rows_any(uint64_t A, uint64_t B, int slot, int list, const float d, float &acc, unsigned &np) const {
        const uint4 l0 = slot < kListSlots ? lists[2 * slot] : p.M.olist[2 * list];
        const uint4 l1 = slot < kListSlots ? lists[2 * slot + 1] : p.M.olist[2 * list + 1];
        const uint4 u0 = make_uint4((uint32_t)uni((int)l0.x), (uint32_t)uni((int)l0.y), (uint32_t)uni((int)l0.z), (uint32_t)uni((int)l0.w));
        const uint4 u1 = make_uint4((uint32_t)uni((int)l1.x), (uint32_t)uni((int)l1.y), (uint32_t)uni((int)l1.z), (uint32_t)uni((int)l1.w));
        switch ((int)(u0.x & 0xffffu)) {
        case 1: rows_fixed<1, PASS>(A, u0, u1, d, acc, np); break;
        case 2: rows_fixed<2, PASS>(A, u0, u1, d, acc, np); break;
        case 3: rows_fixed<3, PASS>(A, u0, u1, d, acc, np); break;
        case 4: rows_fixed<4, PASS>(A, u0, u1, d, acc, np); break;
        case 5: rows_fixed<5, PASS>(A, u0, u1, d, acc, np); break;
        case 6: rows_fixed<6, PASS>(A, u0, u1, d, acc, np); break;
        case 7: rows_fixed<7, PASS>(A, u0, u1, d, acc, np); break;
        case 8: rows_fixed<8, PASS>(A, u0, u1, d, acc, np); break;
        case 9: rows_fixed<9, PASS>(A, u0, u1, d, acc, np); break;
        case 10: rows_fixed<10, PASS>(A, u0, u1, d, acc, np); break;
        case 11: rows_fixed<11, PASS>(A, u0, u1, d, acc, np); break;
        case 12: rows_fixed<12, PASS>(A, u0, u1, d, acc, np); break;
        default: // more than 12 compatible nodes in one model cluster: walk the masks
            for (uint64_t mm = A; mm; mm &= mm - 1) row_terms<PASS, SMEM>(tab, (__ffsll((unsigned long long)mm) - 1) * Nm * 16, B, d, acc, np);
        }
    }

    // ---- one batch of node pairs (u in group 1, v in group 2, level(u) < level(v)) against every entry (a, b):
    // scoring_matching_pair's inner loops (match_utils.py:26-69) for this lane's (u, v, conformer)
    __device__ void pair_batch(int b0, int cnt, float d, uint64_t cand1, int t1, uint64_t cand2, int t2) {
        const bool act = s < cnt;
        const int pr = pairbuf[b0 + (act ? s : 0)];
        const int u = pr & 255, v = pr >> 8;
        const int i = X.nodelevel[u], j = X.nodelevel[v];
        const int kI = __popcll(cand1), kJ = __popcll(cand2);
        const uint32_t ebase = X.rowbase[i] + (uint32_t)kI * ((uint32_t)X.ksum[j] - (uint32_t)X.ksum[i + 1]);
        const uint64_t T1 = uni64(tnodes[t1]), T2 = uni64(tnodes[t2]);
        const bool on = act && lane_live;
        ++n_batches;
        int x = 0;
        for (uint64_t am = cand1; am; am &= am - 1, ++x) {
            const int a = __ffsll((unsigned long long)am) - 1;
            const uint64_t A = uni64(cnodes[a]) & T1;
            if (!A) continue;
            const int na = __popcll(A);
            int y = 0;
            for (uint64_t bm = cand2; bm; bm &= bm - 1, ++y) {
                const int b = __ffsll((unsigned long long)bm) - 1;
                const uint64_t B = uni64(cnodes[b]) & T2;
                if (!B) continue;
                if (bn > 1 && (x * kJ + y) % bn != bw) continue; // another wave of the block builds this entry
                float acc = 0.f;
                unsigned np = 0u;
                rows_any<true>(A, B, y, b * 128 + t2, d, acc, np);
                const int mn = na * __popcll(B); // num_match (match_utils.py:34)
                n_terms += (unsigned)mn;
                float val = acc / (float)mn;
                if (!(A & p.nzw) || !(B & p.nzw)) val = __builtin_nanf(""); // weights_sum = 0: 1 / weights_sum (match_utils.py:50-52)
                if (on) add_entry(ebase + (uint32_t)(x * kJ + y), val, 2 * (int)np < mn); // :61
            }
        }
    }

    // ---- one batch of node pairs of the same ligand cluster (u < v): the diagonal entries (a, a) -> self table
    // (scoring_matching_self, match_utils.py:87-120)
    __device__ void self_batch(int b0, int cnt, float d, uint64_t cand1, int t1, int t2) {
        const bool act = s < cnt;
        const int pr = pairbuf[b0 + (act ? s : 0)];
        const int u = pr & 255;
        const int i = X.nodelevel[u];
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
            if (bn > 1 && x % bn != bw) continue;
            float acc = 0.f;
            unsigned dummy = 0u;
            rows_any<false>(A, B, x, a * 128 + t2, d, acc, dummy);
            const int mn = __popcll(A) * __popcll(B);
            n_terms += (unsigned)mn;
            float val = acc / (float)mn;
            if (!(A & p.nzw) || !(B & p.nzw)) val = __builtin_nanf("");
            if (on) atomicAdd(&St[(sbase + (uint32_t)x) * G + (uint32_t)c], val);
        }
    }

    // distance of the pair this lane takes in the batch starting at b0 (garbage but harmless for lanes beyond the batch)
    __device__ __forceinline__ float batch_dist(int b0, int npairs) const {
        const int idx = b0 + s < npairs ? b0 + s : b0;
        const int pr = pairbuf[idx];
        return dist(pr & 255, pr >> 8);
    }
    template <typename F>
    __device__ __forceinline__ void run_batches(int &npairs, bool all, F &&batch) {
        lds_sync();
        int done = 0;
        float dnext = batch_dist(0, npairs); // the next batch's coordinates are fetched while this one is computed
        while (npairs - done >= NP || (all && npairs > done)) {
            const int cnt = (npairs - done) < NP ? (npairs - done) : NP;
            const float d = dnext;
            if (done + cnt < npairs) dnext = batch_dist(done + cnt, npairs);
            const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
            batch(done, cnt, d);
            cyc_batch += __builtin_amdgcn_s_memtime() - tb0;
            done += cnt;
        }
        const int rest = npairs - done;
        if (done && rest) {
            const int tmp = lane < rest ? (int)pairbuf[done + lane] : 0;
            lds_sync();
            if (lane < rest) pairbuf[lane] = (uint16_t)tmp;
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
                { // stage the column lists of this group's candidate clusters
                    const int y = lane >> 1;
                    uint64_t bm = cand2;
                    for (int q = 0; q < y && bm; ++q) bm &= bm - 1;
                    if (y < kListSlots && bm) lists[lane] = p.M.olist[2 * ((__ffsll((unsigned long long)bm) - 1) * 128 + t2) + (lane & 1)];
                    lds_sync();
                }
                // pairs across ligand clusters: u in g1, v in g2 in a later cluster (match_utils.py:26-31 via graph_match.py:233-279)
                int npairs = 0;
                for (unsigned long long um = g1; um; um &= um - 1) {
                    const int u = __ffsll(um) - 1;
                    const int endu = uni((int)X.lend[lane_get(mylevel, u)]);
                    const unsigned long long vm = endu >= 64 ? 0ull : (g2 & (~0ull << endu));
                    if (!vm) continue;
                    if (npairs + 64 > kPairBuf) run_batches(npairs, false, [&](int b0, int cnt, float d) { pair_batch(b0, cnt, d, cand1, t1, cand2, t2); });
                    if ((vm >> lane) & 1) pairbuf[npairs + __popcll(vm & lt)] = (uint16_t)(u | (lane << 8));
                    npairs += __popcll(vm);
                }
                if (npairs) run_batches(npairs, true, [&](int b0, int cnt, float d) { pair_batch(b0, cnt, d, cand1, t1, cand2, t2); });
                // pairs inside one ligand cluster, u < v (match_utils.py:87, itertools.combinations)
                if (cand1 == cand2) {
                    npairs = 0;
                    for (unsigned long long um = g1; um; um &= um - 1) {
                        const int u = __ffsll(um) - 1;
                        const int endu = uni((int)X.lend[lane_get(mylevel, u)]);
                        const unsigned long long inlevel = (endu >= 64 ? ~0ull : ((1ull << endu) - 1ull)) & (u >= 63 ? 0ull : (~0ull << (u + 1)));
                        const unsigned long long vm = g2 & inlevel;
                        if (!vm) continue;
                        if (npairs + 64 > kPairBuf) run_batches(npairs, false, [&](int b0, int cnt, float d) { self_batch(b0, cnt, d, cand1, t1, t2); });
                        if ((vm >> lane) & 1) pairbuf[npairs + __popcll(vm & lt)] = (uint16_t)(u | (lane << 8));
                        npairs += __popcll(vm);
                    }
                    if (npairs) run_batches(npairs, true, [&](int b0, int cnt, float d) { self_batch(b0, cnt, d, cand1, t1, t2); });
                }
            }
        }
        lds_sync();
        if (bn == 1) {
            const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
            finish_tables();
            cyc_finish += __builtin_amdgcn_s_memtime() - tf0;
        }
    }

    // ---- cluster-distance prefilter (graph_match.py:263-268) and the fail rule (match_utils.py:71-74) turn the
    // accumulators into the pair table: -1 where the entry is invalid.
    __device__ void finish_tables() {
        for (int i = 0; i < nl; ++i) {
            const int si = uni((int)X.lstart[i]), ei = uni((int)X.lend[i]), ki = uni((int)X.lk[i]);
            const float4 gi = Gt[(size_t)i * G + c];
            const uint64_t cand_i = uni64(X.cand[i]);
            for (int j = i + 1; j < nl; ++j) {
                if (bn > 1 && (i * PMX_MAX_LEVELS + j) % bn != bw) continue; // another wave of the team finishes this level pair
                const int sj = uni((int)X.lstart[j]), ej = uni((int)X.lend[j]), kj = uni((int)X.lk[j]);
                const float4 gj = Gt[(size_t)j * G + c];
                const float ldist = norm3(gi.x - gj.x, gi.y - gj.y, gi.z - gj.z); // graph_match.py:240
                const float lsize = gi.w + gj.w;                                   // :241
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
                        const float2 mp = cpair[a * p.M.K + b];
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
    // fr_* at lane f, mt_* (matched ancestors) at lane q, al_* (conformers alive) at lane = number of matches, lv_* (level
    // tables) at lane l. A candidate (f, b) of a frame whose node has ancestors Q needs sum_{q in Q} P[q -> (f, b)]
    // (tree.py:78-82). The ancestors of frame f are those of its parent frame plus, if the node is a match, that match;
    // the parent's part is the same for every child of the parent, so each frame computes its LOOKAHEAD onto the next
    // level once - La[level][candidate][conformer], NaN where a pair entry is invalid - and a step is a handful of reads:
    // lookahead + the new match's row, for the child itself and for the candidates of the level below it.
    static constexpr int F_MATCHED = 1 << 8, F_ANY = 1 << 9, F_SKIP = 1 << 10, F_LA = 1 << 11;
    static constexpr double kBoundSlack = 1.0 + 1e-9; // covers the float64 rounding of the sums the bound is compared with

    struct Walk {
        int fr_lo = 0, fr_hi = 0, fr_info = 0; // todo mask (64 bit), {mx:8, flags, nm << 16}
        int mt_base = 0, mt_ka = 0;            // rowbase[j] - k_j * ksum[j + 1], k_j | a << 8 | j << 16
        int al_lo = 0, al_hi = 0;
        int lv_k = 0, lv_ks = 0, lv_row = 0;
        double best = 0.0;                     // graph_match.py:104
    };

    __device__ __forceinline__ unsigned long long any_per_candidate(bool ok) const {
        const unsigned long long bal = __ballot(ok);
        if (G == 1) return bal;
        if (G == 64) return bal ? 1ull : 0ull;
        const bool any = lane < NP && ((bal >> (lane * G)) & ((1ull << G) - 1ull)) != 0;
        return __ballot(any);
    }
    __device__ __forceinline__ void pool_best(Walk &w) const {
#pragma unroll
        for (int dd = G; dd < 64; dd <<= 1) {
            const double o = shfl_xor_f64(w.best, dd);
            w.best = o > w.best ? o : w.best;
        }
    }

    // Lookahead of a frame with nm ancestors (mt_* lanes [0, nm)) onto level F: float64 sums in ancestor order.
    __device__ void lookahead(const Walk &w, int F, int nm) {
        const int kF = lane_get(w.lv_k, F), ksF = lane_get(w.lv_ks, F);
        const int ebv = (w.mt_base + (w.mt_ka & 255) * ksF + ((w.mt_ka >> 8) & 255) * kF) * G; // lane q: ancestor q's row for level F
        for (int b0 = 0; b0 < kF; b0 += NP) {
            const bool on = b0 + s < kF;
            const int off = on ? b0 * G + lane : c;
            bool ok = true;
            double pair = 0.0;
            int q = 0;
            for (; q + 4 <= nm; q += 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = Pt[lane_get(ebv, q + u) + off];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ok = ok && (v[u] > 0.f); // tree.py:81
                    pair += (double)v[u];
                }
            }
            for (; q < nm; ++q) {
                const float v0 = Pt[lane_get(ebv, q) + off];
                ok = ok && (v0 > 0.f);
                pair += (double)v0;
            }
            if (on) La[ksF * G + off] = ok ? pair : __builtin_nan("");
        }
    }

    // Enter frame F: a node with nm matches, conformer mask `alive` and totals t, whose ancestors are those the lookahead
    // of level F was computed for plus, if has_row, the match whose pair-table row for level F starts at entry prow.
    // Candidates that exist -> E; those worth entering -> todo (all of E while nm < 4; with nm >= 4 those whose subtree can
    // still raise a maximum - the others count as children that returned 1). A frame whose children are leaves is
    // finished here: returns true and its return value (tree.py:102). Written without branches around the loads: the
    // common case (at most NP candidates) is one straight line.
    __device__ __forceinline__ bool enter(Walk &w, const int F, const int nm, const unsigned long long alive, const double t, const int prow,
                                          const bool has_row, const bool matched, int &ret) {
        const int kF = lane_get(w.lv_k, F), ksF = lane_get(w.lv_ks, F);
        const bool leaves = F == nl - 1;
        const double before = w.best;
        const double rb = Rt[(size_t)(F + 1) * G + c]; // row nl of R is 0
        const bool mine = (alive >> c) & 1;
        unsigned long long E = 0, L = 0;
        for (int b0 = 0; b0 < kF; b0 += NP) {
            const bool on = b0 + s < kF;
            const int off = on ? b0 * G + lane : c;
            const double la = La[ksF * G + off];
            const float pn = Pt[prow * G + off];
            const float self = St[ksF * G + off];
            const float pnv = has_row ? pn : 1.f;
            const bool ok = on && mine && (la == la) && (pnv > 0.f);
            const double tc = t + (double)self + (la + (double)(has_row ? pn : 0.f)); // tree.py:38-41
            E |= any_per_candidate(ok) << b0;
            if (leaves) { // per-conformer maximum over leaves (graph_match.py:105-108)
                w.best = (ok && tc > w.best) ? tc : w.best;
            } else {
                L |= any_per_candidate(ok && (tc + rb) * kBoundSlack > w.best) << b0;
            }
        }
        if (leaves) {
            const int mx = E ? 1 : 0;
            if (!E || nm + mx < 5) w.best = (mine && t > w.best) ? t : w.best; // the skip leaf carries this node's totals (tree.py:98-101, :42-43)
            if (NP > 1 && __ballot(w.best > before)) pool_best(w);
            ret = mx + (matched ? 1 : 0);
            return true;
        }
        if (nm < 4) L = E;
        lane_set(w.fr_lo, F, (int)(uint32_t)L);
        lane_set(w.fr_hi, F, (int)(uint32_t)(L >> 32));
        // candidates dropped by the bound test count as children that returned 1 (their subtrees hold >= 5 matches)
        lane_set(w.fr_info, F, (nm << 16) | (matched ? F_MATCHED : 0) | (E ? F_ANY : 0) | ((E & ~L) ? 1 : 0));
        return false;
    }

    enum { WALK_PLAIN = 0, WALK_COLLECT = 1 };

    __device__ __forceinline__ void walk_init(Walk &w) const {
        w.lv_k = lane < nl ? (int)X.lk[lane] : 0;
        w.lv_ks = lane <= nl ? (int)X.ksum[lane] : 0;
        w.lv_row = lane < nl ? (int)X.rowbase[lane] : 0;
    }

    // The DFS below frame f0 (already entered). PLAIN: returns false when `budget` steps did not suffice (the walk is
    // abandoned). COLLECT: children that would hold 5 matches are not entered but handed to sink(level, cand, mask, total)
    // (if it returns false - no room - they are walked in place).
    template <int MODE, typename Sink>
    __device__ bool walk_loop(Walk &w, const int f0, unsigned long long budget, Sink &&sink) {
        int f = f0;
        for (;;) {
            ++n_steps;
            if (MODE == WALK_PLAIN && budget-- == 0) return false;
            const int info = lane_get(w.fr_info, f);
            const unsigned long long todo = (unsigned long long)(uint32_t)lane_get(w.fr_lo, f) | ((unsigned long long)(uint32_t)lane_get(w.fr_hi, f) << 32);
            const int nm = (info >> 16) & 255;
            const int mx = info & 255;
            const unsigned long long alive =
                (unsigned long long)(uint32_t)lane_get(w.al_lo, nm) | ((unsigned long long)(uint32_t)lane_get(w.al_hi, nm) << 32);
            const int F = f + 1;
            const bool skip = !todo && !(info & F_SKIP) && (!(info & F_ANY) || nm + mx < 5); // tree.py:98-101
            if (!todo && !skip) { // all children done: return max_num_matches + matched (tree.py:102)
                const int r1 = mx + ((info & F_MATCHED) ? 1 : 0);
                if (f == f0) return true;
                --f;
                const int pinfo = lane_get(w.fr_info, f);
                if (r1 > (pinfo & 255)) lane_set(w.fr_info, f, (pinfo & ~255) | r1);
                continue;
            }
            if (!(info & F_LA)) { // this frame's lookahead onto level F, once
                lookahead(w, F, nm);
                lds_sync();
            }
            int newinfo = info | F_LA | (skip ? F_SKIP : 0);
            if (skip) { // skip child: same matches, same conformers, same totals
                const double t = Tt[(size_t)nm * G + c];
                int r1 = 0;
                if (enter(w, F, nm, alive, t, 0, false, false, r1)) {
                    if (r1 > mx) newinfo = (newinfo & ~255) | r1;
                    lane_set(w.fr_info, f, newinfo);
                } else {
                    lane_set(w.fr_info, f, newinfo);
                    f = F;
                }
                continue;
            }
            // next candidate child (tree.py:94-97)
            const int b = __ffsll(todo) - 1;
            const unsigned long long left = todo & (todo - 1);
            lane_set(w.fr_lo, f, (int)(uint32_t)left);
            lane_set(w.fr_hi, f, (int)(uint32_t)(left >> 32));
            const int kf = lane_get(w.lv_k, f), ksf = lane_get(w.lv_ks, f), kF = lane_get(w.lv_k, F), ksF = lane_get(w.lv_ks, F);
            const int rowf = lane_get(w.lv_row, f);
            // the child's total and conformer mask: parent + self + accumulated pair (tree.py:38-41, :78-82); all lanes of a
            // conformer compute it redundantly: the parent frame's lookahead onto level f, plus - when this frame's node is
            // itself a match - that match's row
            const bool fm = (info & F_MATCHED) != 0;
            const int qm = nm > 0 ? nm - 1 : 0;
            const int ka = lane_get(w.mt_ka, qm);
            const int eb = fm ? lane_get(w.mt_base, qm) + (ka & 255) * ksf + ((ka >> 8) & 255) * kf : 0;
            const int offf = (ksf + b) * G + c;
            const double la = La[offf];
            const float pm = Pt[(eb + b) * G + c];
            const double tpar = Tt[(size_t)nm * G + c];
            const float selff = St[offf];
            const double rbf = Rt[(size_t)F * G + c];
            const bool okf = ((alive >> c) & 1) && (la == la) && ((fm ? pm : 1.f) > 0.f);
            const double t = tpar + (double)selff + (la + (double)(fm ? pm : 0.f));
            const unsigned long long bal = __ballot(okf);
            const unsigned long long cmask = (G == 64) ? bal : (bal & ((1ull << G) - 1ull));
            if (nm >= 4) { // the child holds >= 5 matches: dropping its subtree cannot change a skip decision
                bool gone = !__ballot(okf && (t + rbf) * kBoundSlack > w.best); // the maxima may have grown since the frame was entered
                if (MODE == WALK_COLLECT && !gone && nm == 4) gone = sink(f, b, cmask, t, w); // given away: it returns at least 1
                if (gone) {
                    if (mx < 1) newinfo = (newinfo & ~255) | 1;
                    lane_set(w.fr_info, f, newinfo);
                    continue;
                }
            }
            // descend
            if (s == 0) Tt[(size_t)(nm + 1) * G + c] = t;
            lane_set(w.al_lo, nm + 1, (int)(uint32_t)cmask);
            lane_set(w.al_hi, nm + 1, (int)(uint32_t)(cmask >> 32));
            lane_set(w.mt_base, nm, rowf - kf * ksF); // ksum[f + 1] = ksum[F]
            lane_set(w.mt_ka, nm, kf | (b << 8) | (f << 16));
            int r1 = 0;
            if (enter(w, F, nm + 1, cmask, t, rowf + b * kF, true, true, r1)) {
                if (r1 > mx) newinfo = (newinfo & ~255) | r1;
                lane_set(w.fr_info, f, newinfo);
            } else {
                lane_set(w.fr_info, f, newinfo);
                f = F;
            }
        }
    }

    // mean over conformers of the per-conformer maxima (graph_match.py:109); `best` must be pooled over the slots
    __device__ __forceinline__ float mean_score(double best) const {
        double sum = lane_live ? best : 0.0;
#pragma unroll
        for (int dd = 1; dd < G; dd <<= 1) sum += shfl_xor_f64(sum, dd);
        return (float)(sum / (double)C);
    }

    // The whole tree with one wave. Returns false (score untouched) when `budget` steps did not suffice.
    __device__ bool walk(unsigned long long budget, float &score) {
        Walk w;
        const unsigned long long allc = (C >= 64) ? ~0ull : ((1ull << C) - 1ull);
        walk_init(w);
        lane_set(w.al_lo, 0, (int)(uint32_t)allc);
        lane_set(w.al_hi, 0, (int)(uint32_t)(allc >> 32));
        if (s == 0) Tt[c] = 0.0;
        lookahead(w, 0, 0); // no ancestors: zeros
        lds_sync();
        if (p.seed_mode == 2) w.best = p.seed[(size_t)cur_li * G + c] * (1.0 - 1e-7);
        int ret = 0;
        if (!enter(w, 0, 0, allc, 0.0, 0, false, false, ret)) {
            if (!walk_loop<WALK_PLAIN>(w, 0, budget, [](int, int, unsigned long long, double, Walk &) { return false; })) return false;
        }
        pool_best(w);
        if (p.seed_mode == 1 && s == 0) p.seed[(size_t)cur_li * G + c] = w.best;
        score = mean_score(w.best);
        return true;
    }

    // ---- cooperative walk (coop_kernel): wave 0 walks the top of the tree and collects the roots of the subtrees with 5
    // matches; then every wave of the block takes roots off a counter. `Tw` = this wave's own path totals.
    __device__ void coop_top(CoopShared &S, uint8_t *roots, uint32_t roots_cap) {
        Walk w;
        const unsigned long long allc = (C >= 64) ? ~0ull : ((1ull << C) - 1ull);
        walk_init(w);
        lane_set(w.al_lo, 0, (int)(uint32_t)allc);
        lane_set(w.al_hi, 0, (int)(uint32_t)(allc >> 32));
        if (s == 0) Tt[c] = 0.0;
        lookahead(w, 0, 0);
        lds_sync();
        uint32_t nroots = 0;
        int ret = 0;
        if (!enter(w, 0, 0, allc, 0.0, 0, false, false, ret)) {
            (void)walk_loop<WALK_COLLECT>(w, 0, 0, [&](int f, int b, unsigned long long cmask, double t, Walk &ww) -> bool {
                if (nroots >= roots_cap) {
                    S.overflow = 1;
                    return false;
                }
                RootHeader *rh = reinterpret_cast<RootHeader *>(roots + (size_t)nroots * root_bytes<G>());
                // lane q < 4 holds ancestor q in mt_ka (k | cand << 8 | level << 16)
                if (lane < 4) {
                    rh->path[2 * lane] = (uint8_t)(ww.mt_ka >> 16);
                    rh->path[2 * lane + 1] = (uint8_t)(ww.mt_ka >> 8);
                }
                if (lane == 0) {
                    rh->f = (uint8_t)f;
                    rh->nm = 5;
                    rh->mask = cmask;
                    rh->path[8] = (uint8_t)f;
                    rh->path[9] = (uint8_t)b;
                }
                if (s == 0) reinterpret_cast<double *>(rh + 1)[c] = t;
                ++nroots;
                return true;
            });
        }
        pool_best(w);
        if (s == 0 && w.best > 0.0) atomicMax(&S.best[c], (unsigned long long)__double_as_longlong(w.best));
        if (lane == 0) S.nroots = nroots;
    }

    __device__ void coop_subtrees(CoopShared &S, const uint8_t *roots) {
        Walk w;
        walk_init(w);
        const uint32_t nroots = (uint32_t)uni((int)S.nroots);
        for (;;) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(&S.next_root, 1u);
            i = (uint32_t)uni((int)i);
            if (i >= nroots) break;
            const RootHeader *rh = reinterpret_cast<const RootHeader *>(roots + (size_t)i * root_bytes<G>());
            const int nm0 = uni((int)rh->nm), f0 = uni((int)rh->f) + 1;
            const unsigned long long mask = uni64(rh->mask);
            const double t = reinterpret_cast<const double *>(rh + 1)[c];
            // the maxima found by the whole block so far
            w.best = __longlong_as_double((long long)S.best[c]);
            if (lane < nm0) {
                const int j = rh->path[2 * lane], a = rh->path[2 * lane + 1];
                const int kj = X.lk[j];
                w.mt_base = (int)X.rowbase[j] - kj * (int)X.ksum[j + 1];
                w.mt_ka = kj | (a << 8) | (j << 16);
            }
            lane_set(w.al_lo, nm0, (int)(uint32_t)mask);
            lane_set(w.al_hi, nm0, (int)(uint32_t)(mask >> 32));
            // a subtree that can no longer raise any maximum is not walked
            const double rb = Rt[(size_t)f0 * G + c];
            if (!__ballot(((mask >> c) & 1) && (t + rb) * kBoundSlack > w.best)) continue;
            if (s == 0) Tt[(size_t)nm0 * G + c] = t;
            // the root's frame: lookahead of its parent (the first nm0 - 1 matches) plus the root's own row
            lookahead(w, f0, nm0 - 1);
            lds_sync();
            const double before = w.best;
            int ret = 0;
            const int jr = uni((int)rh->path[2 * (nm0 - 1)]), ar = uni((int)rh->path[2 * (nm0 - 1) + 1]);
            const int prow = (int)X.rowbase[jr] + ar * (int)X.lk[f0]; // rowbase[j] + a * k[j + 1], j + 1 = f0
            if (!enter(w, f0, nm0, mask, t, uni(prow), true, true, ret))
                (void)walk_loop<WALK_PLAIN>(w, f0, ~0ull, [](int, int, unsigned long long, double, Walk &) { return false; });
            pool_best(w);
            if (s == 0 && w.best > before) atomicMax(&S.best[c], (unsigned long long)__double_as_longlong(w.best));
        }
    }
};

// One block = W wavefronts sharing the staged model; every wavefront pulls ligands of this size class until none are left.
// A tree that needs more than p.budget steps is passed on to coop_kernel.
template <int G, bool LDS_TABLES>
__global__ __launch_bounds__(1024) void match_kernel(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Nm = p.M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + round16(uint64_t(Nm) * Nm * 16));
    uint64_t *tnodes = cnodes + 64;
    float2 *cpair = reinterpret_cast<float2 *>(tnodes + 128);
    const int K = p.M.K;
    for (int i = threadIdx.x; i < Nm * Nm; i += blockDim.x) tab[i] = p.wtab[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = p.M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = p.M.tnodes[i];
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) cpair[i] = p.M.cpair[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *ctx = smem + model_lds_bytes(Nm, K) + (size_t)wave * p.wave_bytes;
    unsigned char *tables = ctx + sizeof(MatchCtx);
    const uint32_t count = (uint32_t)uni((int)p.bins->count[p.bin]);
    Matcher<G, LDS_TABLES> mt(p, tab, cnodes, tnodes, cpair, ctx);
    for (;;) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&p.bins->cursor[p.bin], 1u);
        pos = (uint32_t)uni((int)pos);
        if (pos >= count) break;
        const uint32_t li = (uint32_t)uni((int)p.list[pos]);
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        const bool has_tree = mt.setup(li, tables);
        mt.cyc_setup += __builtin_amdgcn_s_memtime() - ts0;
        if (!has_tree) continue;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        mt.build_tables();
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        mt.build_bounds();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        mt.cyc_bounds += t1 - tb;
        const unsigned long long steps0 = mt.n_steps, top0 = mt.n_top;
        float score = 0.f;
        const bool done = mt.walk(p.budget, score);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        mt.cyc_tab += t1 - t0;
        mt.cyc_walk += t2 - t1;
        if (lane == 0) {
            if (done) {
                p.scores[li] = score;
            } else { // over budget: the block-cooperative kernel takes it
                const uint32_t hp = atomicAdd(&p.bins->hcount[p.bin], 1u);
                p.hlist[hp] = li;
            }
            const unsigned long long st = mt.n_steps - steps0;
            const int bucket = st ? min(11, (63 - __clzll((long long)st)) / 2) : 0; // steps in [4^b, 4^(b+1))
            atomicAdd(&p.stats[4 + bucket], 1ull);
            atomicAdd(&p.stats[16 + bucket], st);
            atomicAdd(&p.stats[32 + bucket], mt.n_top - top0);
            atomicMax(&p.stats[3], st);
        }
    }
    if (lane == 0) {
        atomicAdd(&p.stats[0], mt.n_steps);
        atomicAdd(&p.stats[1], mt.n_batches);
        atomicAdd(&p.stats[2], mt.n_terms);
        atomicAdd(&p.stats[80 + 4 * p.bin], mt.n_terms);
        atomicAdd(&p.stats[81 + 4 * p.bin], mt.n_steps);
        atomicAdd(&p.stats[82 + 4 * p.bin], mt.n_batches);
        atomicAdd(&p.stats[83 + 4 * p.bin], mt.cyc_batch);
        atomicAdd(&p.stats[70], mt.cyc_setup);
        atomicAdd(&p.stats[71], mt.cyc_batch);
        atomicAdd(&p.stats[72], mt.cyc_finish);
        atomicAdd(&p.stats[73], mt.cyc_bounds);
        atomicAdd(&p.stats[44], mt.cyc_tab);
        atomicAdd(&p.stats[45], mt.cyc_walk);
        atomicAdd(&p.stats[46 + 2 * p.bin], mt.cyc_tab);
        atomicAdd(&p.stats[47 + 2 * p.bin], mt.cyc_walk);
    }
}

// The table phase of the chunk pipeline (pmx_api.hip): every wavefront pulls ligands of its LDS size class, builds their
// tables and search bounds in LDS and writes them to the chunk's arena for the tree kernels.
template <int G>
__global__ __launch_bounds__(1024) void tables_kernel_v3(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Nm = p.M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + round16(uint64_t(Nm) * Nm * 16));
    uint64_t *tnodes = cnodes + 64;
    float2 *cpair = reinterpret_cast<float2 *>(tnodes + 128);
    const int K = p.M.K;
    for (int i = threadIdx.x; i < Nm * Nm; i += blockDim.x) tab[i] = p.wtab[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = p.M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = p.M.tnodes[i];
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) cpair[i] = p.M.cpair[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *ctx = smem + model_lds_bytes(Nm, K) + (size_t)wave * p.wave_bytes;
    unsigned char *tables = ctx + sizeof(MatchCtx);
    const uint32_t count = (uint32_t)uni((int)p.bins->count[p.bin]);
    Matcher<G, true> mt(p, tab, cnodes, tnodes, cpair, ctx);
    for (;;) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&p.bins->cursor[p.bin], 1u);
        pos = (uint32_t)uni((int)pos);
        if (pos >= count) break;
        const uint32_t li = (uint32_t)uni((int)p.list[pos]);
        if (!mt.setup(li, tables)) continue;
        mt.build_tables();
        mt.build_bounds();
        mt.write_out(p.out_arena + p.taboff[li]);
    }
}

// tables_kernel_v4: the same table builder with the occupancy of a kernel that owns no model copy. A block is a TEAM of
// waves on one ligand at a time: the edge table is read through the scalar cache (SMEM = true), LDS holds only the small
// cluster tables and the ligand's accumulators; every wave builds the entries (x, y) of its residue class and finishes
// the level pairs of its residue class (disjoint entries: the bits do not depend on the team size), wave 0 computes the
// bounds, all waves write the block of the chunk's arena. Many small blocks per CU: the serial steps of one team are
// covered by the others.
template <int G>
__global__ __launch_bounds__(1024) void tables_kernel_v4(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = p.M.K, Nm = p.M.Nm;
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem);
    uint64_t *tnodes = cnodes + 64;
    float2 *cpair = reinterpret_cast<float2 *>(tnodes + 128);
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = p.M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = p.M.tnodes[i];
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) cpair[i] = p.M.cpair[i];
    const uint32_t small_bytes = 64 * 8 + 128 * 8 + (uint32_t)round16(uint64_t(K) * K * 8);
    uint32_t *item = reinterpret_cast<uint32_t *>(smem + small_bytes); // [0] list position, [1] has_tree
    unsigned char *ctx = smem + small_bytes + 16;
    unsigned char *tables = ctx + sizeof(MatchCtx);
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform for the compiler too: what a wave does
                                                                             // under `if (wave == 0)` stays in scalar registers
    unsigned char *helper = ctx + p.wave_bytes + (size_t)(wave > 0 ? wave - 1 : 0) * (2 * kPairBuf + 32 * kListSlots);
    const uint32_t count = p.bins->count[p.bin];
    Matcher<G, true, true> mt(p, p.wtab, cnodes, tnodes, cpair, ctx);
    mt.bn = nwaves;
    mt.bw = wave;
    if (wave != 0) {
        mt.pairbuf = reinterpret_cast<uint16_t *>(helper);
        mt.lists = reinterpret_cast<uint4 *>(helper + 2 * kPairBuf);
    }
    (void)Nm;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) item[0] = atomicAdd(&p.bins->cursor[p.bin], 1u);
        __syncthreads();
        const uint32_t pos = item[0];
        if (pos >= count) break;
        const uint32_t li = p.list[pos];
        if (wave == 0) {
            const bool has_tree = mt.setup(li, tables);
            if (lane == 0) {
                mt.X.meta[0] = (uint32_t)mt.nl;
                mt.X.meta[1] = mt.T;
                mt.X.meta[2] = mt.ksumtot;
                item[1] = has_tree ? 1u : 0u;
            }
        }
        __syncthreads();
        if (!item[1]) continue;
        if (wave != 0) mt.attach(li, tables, nullptr);
        mt.build_tables();
        __syncthreads();
        mt.finish_tables();
        __syncthreads();
        if (wave == 0) mt.build_bounds();
        __syncthreads();
        mt.write_out_team(p.out_arena + p.taboff[li], wave, nwaves);
    }
}

// Block-cooperative form for ligands whose tables are large (few fit a CU) or whose tree is heavy: ONE ligand per block
// at a time, its tables in the block's LDS (or, LDS_TABLES = false, in the block's slot of the HBM arena). Wave 0 builds the
// tables and walks the top of the tree, collecting the roots of the subtrees with 5 matches (>= 95 % of a heavy tree's
// nodes lie below them); then all waves take roots off a counter. Per-wave LDS: path totals only.
template <int G, bool LDS_TABLES>
__global__ __launch_bounds__(1024) void coop_kernel(const MatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Nm = p.M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + round16(uint64_t(Nm) * Nm * 16));
    uint64_t *tnodes = cnodes + 64;
    float2 *cpair = reinterpret_cast<float2 *>(tnodes + 128);
    const int K = p.M.K;
    for (int i = threadIdx.x; i < Nm * Nm; i += blockDim.x) tab[i] = p.wtab[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = p.M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = p.M.tnodes[i];
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) cpair[i] = p.M.cpair[i];
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform for the compiler too: what a wave does
                                                                             // under `if (wave == 0)` stays in scalar registers
    CoopShared &S = *reinterpret_cast<CoopShared *>(smem + model_lds_bytes(Nm, K));
    unsigned char *ctx = reinterpret_cast<unsigned char *>(&S + 1);
    const uint32_t table_bytes = LDS_TABLES ? p.wave_bytes - (uint32_t)sizeof(MatchCtx) : 0u;
    unsigned char *tables = ctx + sizeof(MatchCtx);
    // helpers' path totals + lookahead sums come out of a pool behind the tables (wave 0 uses the tables' own scratch)
    double *pool = reinterpret_cast<double *>(ctx + sizeof(MatchCtx) + table_bytes);
    const uint32_t pool_doubles = p.pool_bytes / 8;
    if (!LDS_TABLES) { // this block's slot of the HBM arena, sized for the call's largest table
        const uint64_t need = round16((uint64_t)p.bins->max_need);
        if (need == 0 || (uint64_t)(blockIdx.x + 1) * need > p.arena_bytes) return; // blocks without a slot do not take part
        tables = p.arena + (size_t)blockIdx.x * need;
    }
    uint8_t *roots = p.roots + (size_t)blockIdx.x * p.roots_cap * root_bytes<G>();
    const uint32_t count = p.bins->hcount[p.bin];
    Matcher<G, LDS_TABLES> mt(p, tab, cnodes, tnodes, cpair, ctx);
    for (;;) {
        __syncthreads(); // also orders the previous ligand's last reads before the next one's writes
        if (threadIdx.x == 0) {
            S.item = atomicAdd(&p.bins->hcursor[p.bin], 1u);
            S.nroots = 0;
            S.next_root = 0;
            S.has_tree = 0;
        }
        if (threadIdx.x < 64) S.best[threadIdx.x] = 0ull;
        __syncthreads();
        const uint32_t pos = S.item;
        if (pos >= count) break;
        const uint32_t li = p.hlist[pos];
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        // helper waves keep their pair buffer / column lists (build) and later their walk state in a pool behind the tables
        const uint32_t helper_bytes = nwaves > 1 ? p.pool_bytes / (uint32_t)(nwaves - 1) : 0u;
        unsigned char *own = reinterpret_cast<unsigned char *>(pool) + (size_t)(wave > 0 ? wave - 1 : 0) * helper_bytes;
        if (wave == 0) {
            const bool has_tree = mt.setup(li, tables);
            if (lane == 0) {
                mt.X.meta[0] = (uint32_t)mt.nl;
                mt.X.meta[1] = mt.T;
                mt.X.meta[2] = mt.ksumtot;
                S.has_tree = has_tree ? 1u : 0u;
                if (!has_tree) p.scores[li] = 0.f;
            }
        }
        __syncthreads();
        if (!S.has_tree) continue;
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        // every wave builds its share of the table entries (disjoint entries: the sums do not depend on the number of waves)
        mt.bn = nwaves;
        mt.bw = wave;
        if (wave != 0) {
            mt.attach(li, tables, reinterpret_cast<double *>(own));
            mt.pairbuf = reinterpret_cast<uint16_t *>(own);
            mt.lists = reinterpret_cast<uint4 *>(own + 2 * kPairBuf);
        }
        mt.build_tables();
        __syncthreads();
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
        unsigned long long c3 = c2;
        mt.finish_tables(); // level pairs are divided among the waves like the entries were
        __syncthreads();
        if (wave == 0) {
            mt.build_bounds();
            c3 = __builtin_amdgcn_s_memtime();
            mt.coop_top(S, roots, p.roots_cap);
            __threadfence_block();
        }
        __syncthreads();
        const unsigned long long c4 = __builtin_amdgcn_s_memtime();
        bool takes_part = true;
        if (wave != 0) {
            const uint32_t need = (mt.X.meta[0] + 1 + mt.X.meta[2]) * G; // doubles: (nl + 1 + ksumtot) * G
            takes_part = (uint64_t)wave * need <= pool_doubles;
            if (takes_part) mt.attach(li, tables, pool + (size_t)(wave - 1) * need);
        }
        if (takes_part) mt.coop_subtrees(S, roots);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long c5 = __builtin_amdgcn_s_memtime();
            atomicAdd(&p.stats[100], c1 - c0);
            atomicAdd(&p.stats[101], c2 - c1);
            atomicAdd(&p.stats[102], c3 - c2);
            atomicAdd(&p.stats[103], c4 - c3);
            atomicAdd(&p.stats[104], c5 - c4);
            atomicAdd(&p.stats[105], (unsigned long long)S.nroots);
            atomicAdd(&p.stats[106], (unsigned long long)S.overflow);
            atomicAdd(&p.stats[107], 1ull);
            atomicMax(&p.stats[108], (unsigned long long)S.nroots);
            S.overflow = 0;
        }
        if (wave == 0) {
            const double best = __longlong_as_double((long long)S.best[lane % G]);
            const float score = mt.mean_score(best);
            if (lane == 0) p.scores[li] = score;
        }
        (void)nwaves;
    }
    if (lane == 0) {
        atomicAdd(&p.stats[0], mt.n_steps);
        atomicAdd(&p.stats[1], mt.n_batches);
        atomicAdd(&p.stats[2], mt.n_terms);
        atomicAdd(&p.stats[74], mt.n_steps);
    }
}

} // namespace pmx

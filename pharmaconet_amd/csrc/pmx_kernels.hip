// pmx_kernels.hip - the screening hot path on gfx950 (CDNA4, wave64).
//
// Execution model. A ligand's conformers live in a *conformer group*: G = 2^ceil(log2(max conformers))
// adjacent lanes of a wavefront, lane c of the group holding conformer c (8 lanes at the 8-conformer
// shape of BASELINE.json, 64 at 64 conformers). Everything that is the same for all conformers of a
// ligand (tree control, candidate sets, table indices) is computed redundantly by the group's lanes,
// so control flow is uniform inside a group and diverges only between groups. Conformers are coupled
// exactly where the reference couples them (the joint `any conformer valid` and
// `num_matches + max_num_matches < 5` tests of tree.py:83-84,98), through __ballot.
//
// Kernels per chunk of ligands (table phase on a side stream, tree phase on the caller's stream, ordered by
// events; see pmx_api.hip):
//   clear_kernel       zeroes the chunk's counters and accumulators
//   sizes_kernel       one thread per ligand: candidate sets -> number of tree levels, table size
//   scan_kernel        table sizes -> offsets in the scratch arena
//   tables_kernel_v2   one wavefront per ligand: the self / pair score tables of match_utils.py
//   bounds_kernel      per level, the most it can still add to a total (lets the tree search drop subtrees)
//   tree_kernel<G,0>   one wavefront (= block) per ligand: the DFS of tree.py over those tables, shared
//                      by the wave's 64 / G groups; per-conformer maximum over leaves, mean -> score
//   tree_kernel<G,1>   the same walker on subtrees queued by over-budget trees, in rounds
//   finalize_kernel    scores of ligands whose tree was split
//
// Floating point: -ffp-contract=off (see build flags). Distances and the 2-sigma test reproduce the
// reference's float32 results bit for bit; Gaussian sums agree to float32 rounding.
#include "pmx_device.h"

#pragma clang fp contract(off)

namespace pmx {

// ---------------------------------------------------------------------------------------- levels
// Cluster candidates and tree levels (graph_match.py:124-137, :87-88). Clusters arrive sorted by
// priority_fn; a cluster is kept if some model cluster shares a type with it; at most 20 are kept.
struct Levels {
    int nl;
    uint32_t T, ksumtot;
};

template <typename F>
__device__ inline Levels scan_levels(const Record &r, const uint64_t *tclus, F &&emit) {
    Levels L{0, 0, 0};
    uint32_t sumk = 0, sumk2 = 0; // T = sum_{i<j} k_i k_j = (sumk^2 - sumk2) / 2
    int start = 0;
    for (int ci = 0; ci < r.ncl && L.nl < PMX_MAX_LEVELS; ++ci) {
        int end = r.cluster_end[ci];
        unsigned lmask = 0;
        for (int u = start; u < end; ++u) lmask |= r.typemask[u];
        uint64_t cand = tclus[lmask & 127u];
        if (cand) {
            uint32_t k = (uint32_t)__popcll(cand);
            emit(L.nl, start, end, cand, k);
            sumk += k;
            sumk2 += k * k;
            L.nl++;
        }
        start = end;
    }
    L.ksumtot = sumk;
    L.T = (sumk * sumk - sumk2) / 2;
    return L;
}

template <int G>
__device__ inline uint32_t table_units(const Levels &L) {
    if (L.nl == 0) return 0;
    return (uint32_t)(round16(table_bytes<G>(L.T, L.ksumtot, (uint32_t)L.nl)) / 16);
}

// ----------------------------------------------------------------------------------- sizes_kernel
// Wave-wide reductions in front of statistics atomics: one atomic per wavefront instead of one per lane on the same
// address (device-scope atomics on one address serialise at some 20 ns each on this part; inactive lanes contribute 0).
__device__ inline unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ inline unsigned long long wave_max(unsigned long long v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return v;
}

template <int G>
__global__ void sizes_kernel(DevLibrary lib, const uint64_t *tclus, uint64_t first, uint32_t count, uint32_t *units,
                             int32_t *status, uint32_t *meta /* [0] = max levels */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nl = 0;
    if (i < count) {
        Record r = parse_record(lib.data + lib.offsets[first + i]);
        if (!record_supported(r)) {
            units[i] = 0;
            status[i] = PMX_LIGAND_UNSUPPORTED;
        } else {
            Levels L = scan_levels(r, tclus, [](int, int, int, uint64_t, uint32_t) {});
            units[i] = table_units<G>(L);
            status[i] = PMX_LIGAND_OK;
            nl = (uint32_t)L.nl;
        }
    }
    nl = (uint32_t)wave_max(nl);
    if ((threadIdx.x & 63) == 0 && nl > 0) atomicMax(&meta[0], nl);
}

// Exclusive scan of `units` (16-byte units) into byte offsets; one block of 1024 threads.
__global__ void scan_kernel(const uint32_t *units, uint32_t count, uint64_t *taboff, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (count + 1023) / 1024;
    const uint32_t lo = min(t * per, count), hi = min(lo + per, count);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += units[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint32_t i = lo; i < hi; ++i) {
        taboff[i] = run * 16;
        run += units[i];
    }
    if (t == 1023) {
        taboff[count] = part[1023] * 16;
        *total = part[1023] * 16;
    }
}

// Cross-lane hand-off through LDS inside one wavefront: DS operations of a wave execute in order, but
// accesses made through generic pointers are FLAT instructions, which travel another path and may pass
// (or be passed by) DS operations. Drain both counters before another lane's data is consumed.
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

#ifndef PMX_V2_BATCH
#define PMX_V2_BATCH 3 // columns per batch of the term loops: 3 measured best (1 fills more slots but gives up the read batching)
#endif
// --------------------------------------------------------------------------- table kernels: helpers

struct Pos {
    float x, y, z;
};

__device__ inline Pos load_pos(const float *xyz, int C, int u, int c) {
    // unsigned 32-bit element offsets from the record's (wave-uniform) base: one add per address, no 64-bit lane arithmetic
    const uint32_t o = (uint32_t)__mul24(u * 3, C) + (uint32_t)c, uc = (uint32_t)C;
    return Pos{xyz[o], xyz[o + uc], xyz[o + 2u * uc]};
}

// np.linalg.norm of a float32 3-vector: sqrt((x*x + y*y) + z*z), every step rounded (ligand.py:349-351).
__device__ inline float norm3(float dx, float dy, float dz) {
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

// LigandNodeCluster.center / .size for one conformer (ligand.py:458-473).
__device__ inline void cluster_center_size(const float *xyz, int C, int start, int end, int c, Pos &center, float &size) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int u = start; u < end; ++u) {
        Pos p = load_pos(xyz, C, u, c);
        sx = sx + p.x;
        sy = sy + p.y;
        sz = sz + p.z;
    }
    float cnt = (float)(end - start);
    center = Pos{sx / cnt, sy / cnt, sz / cnt};
    float mx = 0.f;
    for (int u = start; u < end; ++u) {
        Pos p = load_pos(xyz, C, u, c);
        float r = norm3(p.x - center.x, p.y - center.y, p.z - center.z);
        mx = (u == start || r > mx) ? r : mx;
    }
    size = mx;
}

// One (ligand node, ligand node) term of match_utils.py:26-69: the sum over compatible model node
// pairs (m in A, n in B) of w_m w_n / std * exp(-z^2 / 2), and the count of pairs within 2 sigma.
// The columns of B are decoded once (twelve at a time) and reused for every row of A, so the inner loop is
// batches of three independent LDS reads + the arithmetic; the Gaussian is one v_exp_f32 (results below 2^-126
// flush to zero, far under float32 resolution of the sums). Order: m ascending, n ascending, as the
// reference (clusters of more than 12 compatible nodes are summed in column blocks of 12).
__device__ inline void node_pair(const float4 *tab, int Ns, uint64_t A, uint64_t B, float d, float &acc, int &npass) {
    constexpr int W = 3, NCOL = 12; // batch width and columns decoded per pass: cluster sizes of 3, 6, 9 nodes waste nothing
    while (B) {
        int col[NCOL];
        int nc = 0;
#pragma unroll
        for (int y = 0; y < NCOL; ++y) {
            col[y] = 0;
            if (B) {
                col[y] = __ffsll((unsigned long long)B) - 1;
                B &= B - 1;
                nc = y + 1;
            }
        }
        // weights of the absent columns are zero: fma(0, x, acc) leaves acc unchanged, so the reads of a batch
        // can be issued together and nothing branches per term
        for (uint64_t am = A; am; am &= am - 1) {
            const float4 *row = tab + (__ffsll((unsigned long long)am) - 1) * Ns;
#pragma unroll
            for (int y0 = 0; y0 < NCOL; y0 += W) {
                if (y0 < nc) {
                    float4 e[W];
#pragma unroll
                    for (int y = 0; y < W; ++y) e[y] = row[col[y0 + y]];
#pragma unroll
                    for (int y = 0; y < W; ++y) {
                        const bool on = y0 + y < nc;
                        const float t = fabsf(d - e[y].x);
                        const float q = t * e[y].y;
                        acc = __builtin_fmaf(on ? e[y].w : 0.f, __builtin_amdgcn_exp2f(-(q * q)), acc);
                        npass += (on && t <= e[y].z) ? 1 : 0;
                    }
                }
            }
        }
    }
}

// The same term with the two node sets given as the model's precomputed lists (DevModel::clist: byte 0 = count,
// bytes 1..12 = node numbers ascending, padded with Nm): no mask walking. Only for sets of at most 12 nodes
// (count != 0xff). `tab` has Ns = Nm + 1 columns; column Nm is neutral ({0, 0, -1, 0}: contributes exactly 0 to
// the sum and never counts as a pass), so the padding of the last batch needs no per-term masking.
__device__ inline void node_pair_lists(const float4 *tab, int Ns, const uint4 la, const uint4 lb, float d, float &acc, int &npass) {
    constexpr int W = PMX_V2_BATCH, NCOL = 12;
    const int na = (int)(la.x & 255u), nc = (int)(lb.x & 255u);
    int col[NCOL]; // absent columns hold the neutral column (weight 0, threshold -1): no masking per term
    col[0] = (lb.x >> 8) & 255u, col[1] = (lb.x >> 16) & 255u, col[2] = lb.x >> 24;
    col[3] = lb.y & 255u, col[4] = (lb.y >> 8) & 255u, col[5] = (lb.y >> 16) & 255u, col[6] = lb.y >> 24;
    col[7] = lb.z & 255u, col[8] = (lb.z >> 8) & 255u, col[9] = (lb.z >> 16) & 255u, col[10] = lb.z >> 24;
    col[11] = lb.w & 255u;
    // rows: a 96-bit shift register over bytes 1..12 of la
    uint32_t s0 = (la.x >> 8) | (la.y << 24), s1 = (la.y >> 8) | (la.z << 24), s2 = (la.z >> 8) | (la.w << 24);
    for (int r = 0; r < na; ++r) {
        const float4 *row = tab + (int)(s0 & 255u) * Ns;
        s0 = (s0 >> 8) | (s1 << 24);
        s1 = (s1 >> 8) | (s2 << 24);
        s2 >>= 8;
#pragma unroll
        for (int y0 = 0; y0 < NCOL; y0 += W) {
            if (y0 < nc) {
                float4 e[W];
#pragma unroll
                for (int y = 0; y < W; ++y) e[y] = row[col[y0 + y]];
#pragma unroll
                for (int y = 0; y < W; ++y) {
                    const float t = fabsf(d - e[y].x);
                    const float q = t * e[y].y;
                    acc = __builtin_fmaf(e[y].w, __builtin_amdgcn_exp2f(-(q * q)), acc);
                    npass += (t <= e[y].z) ? 1 : 0;
                }
            }
        }
    }
}

// One (ligand node, ligand node) term against model clusters a and b; returns |A| * |B| (0: nothing compatible).
__device__ inline int cluster_node_pair(const DevModel &M, const float4 *tab, const uint64_t *cnodes, const uint64_t *tnodes, int a, int b,
                                        unsigned tmu, unsigned tmv, const uint4 la /* M.clist[a * 128 + tmu] */,
                                        const uint4 lb /* M.clist[b * 128 + tmv] */, float d, float &acc, int &npass) {
    const int na = (int)(la.x & 255u), nb = (int)(lb.x & 255u);
    if (na == 0 || nb == 0) return 0;
    if (na != 255 && nb != 255) {
        // Columns come in batches of three, so the longer list makes the better columns (9 rows x 1 column costs 27 term
        // slots, 1 row x 9 columns costs 9). The staged table is symmetric when the model's edges are (they are distances),
        // so swapping the lists changes the order of the sum only.
        const bool swap = M.symmetric && na * ((nb + 2) / 3) > nb * ((na + 2) / 3);
        const uint4 rows = swap ? lb : la, cols = swap ? la : lb;
        node_pair_lists(tab, M.Nm + 1, rows, cols, d, acc, npass);
        return na * nb;
    }
    const uint64_t A = cnodes[a] & tnodes[tmu], B = cnodes[b] & tnodes[tmv];
    node_pair(tab, M.Nm + 1, A, B, d, acc, npass);
    return __popcll(A) * __popcll(B);
}

// ------------------------------------------------------------------------------- tables_kernel_v2
// The self / pair score tables of match_utils.py, organised for lane utilisation: ONE wavefront per ligand. Within a pair of
// ligand clusters (i, j) the work is the flat list of items (entry (a, b), ligand node u, ligand node v);
// the wave's 64 / G lane groups ("slots") take consecutive items, lane c of a slot its conformer c, and
// add the item's likelihood and fail flag into per-entry LDS accumulators (ds_add). Consecutive items
// share (a, b) and so have the same number of model node pairs: the slots of a wave stay balanced, and
// all lanes walk the same ligand, so there is no divergence between ligands.
constexpr int kTabEntryChunk = 32; // entries accumulated in LDS at a time

struct WaveLevels { // per wave, in LDS
    uint64_t cand[PMX_MAX_LEVELS];
    uint8_t start[PMX_MAX_LEVELS];
    uint8_t end[PMX_MAX_LEVELS];
    uint8_t k[PMX_MAX_LEVELS];
};
static_assert(sizeof(WaveLevels) % 16 == 0, "WaveLevels alignment");

// candidate slot -> model cluster id, [PMX_MAX_LEVELS][stride]: a level has at most K candidates, so the stride follows the
// model (the wave's LDS footprint decides how many waves a CU holds: 3.5 KB instead of 6.9 KB at K = 11)
__host__ __device__ constexpr uint32_t tables_v2_cand_stride(int K) { return (uint32_t)((K + 3) & ~3); }

template <int G>
__host__ __device__ constexpr uint32_t tables_v2_wave_bytes(int K) {
    // levels, candidate lists, accumulators, near-entry list, the ligand's node type masks
    return sizeof(WaveLevels) + PMX_MAX_LEVELS * tables_v2_cand_stride(K) + kTabEntryChunk * G * 8 + kTabEntryChunk * 4 + PMX_MAX_LIGAND_NODES;
}

#ifndef PMX_V2_MINWAVES
#define PMX_V2_MINWAVES 6
#endif

template <int G, bool ZW /* some type weight is 0: node pairs whose weights sum to 0 score NaN (match_utils.py:50-52) */>
__global__ __launch_bounds__(512, PMX_V2_MINWAVES) void tables_kernel_v2(DevModel M, DevLibrary lib, Weights W, uint64_t first, uint32_t count,
                                                        const int32_t *status, const uint64_t *taboff, uint8_t *arena,
                                                        const uint32_t *list, const uint32_t *list_count, uint32_t *cursor) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int GPW = 64 / G; // slots per wave
    const uint32_t todo = list ? *list_count : count; // ligands of this launch (all, or the listed ones)
    if ((uint64_t)blockIdx.x * (blockDim.x / 64) >= todo) return; // more blocks than work
    const int Nm = M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    const int Ns = Nm + 1; // row stride: one neutral column after the model's nodes (see node_pair_lists)
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + (size_t)Nm * Ns * sizeof(float4));
    uint64_t *tnodes = cnodes + 64;
    unsigned char *wave_base = reinterpret_cast<unsigned char *>(tnodes + 128) + (size_t)(threadIdx.x >> 6) * tables_v2_wave_bytes<G>(M.K);
    WaveLevels &WL = *reinterpret_cast<WaveLevels *>(wave_base);
    const int cs = (int)tables_v2_cand_stride(M.K);
    uint8_t *candlist = wave_base + sizeof(WaveLevels); // [PMX_MAX_LEVELS][cs]
    float *acc_score = reinterpret_cast<float *>(candlist + PMX_MAX_LEVELS * cs); // [kTabEntryChunk][G]
    unsigned *acc_fail = reinterpret_cast<unsigned *>(acc_score + kTabEntryChunk * G);
    uint32_t *near_list = acc_fail + kTabEntryChunk * G;                   // [kTabEntryChunk] entry | a << 8 | b << 16
    uint8_t *tm = reinterpret_cast<uint8_t *>(near_list + kTabEntryChunk); // [PMX_MAX_LIGAND_NODES]

    for (int i = threadIdx.x; i < Nm * Ns; i += blockDim.x) {
        const int m = i / Ns, n = i - m * Ns;
        float4 e = make_float4(0.f, 0.f, -1.f, 0.f);
        if (n < Nm) {
            e = M.edge[m * Nm + n];
            e.w = (W.w[M.node_type[m]] * W.w[M.node_type[n]]) / e.w; // weights / stds (match_utils.py:65)
        }
        tab[i] = e;
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = M.tnodes[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int s = lane / G, c = lane % G;
    // model nodes whose type has a non-zero weight: a node-pair term whose weights sum to 0 is 0 * (1 / 0) = NaN in the
    // reference (match_utils.py:50-52,69)
    const unsigned long long nzw = __ballot(lane < Nm && W.w[M.node_type[lane < Nm ? lane : 0]] != 0.f);
    constexpr bool some_zero = ZW;
    // The blocks are persistent: a wave that has finished a ligand fetches the next one from `cursor`, so the
    // block's staged model table serves as many ligands as it takes and no wave idles behind a slower neighbour
    // (ligand work spans more than 10x; with one ligand per wave a block lived as long as its slowest wave).
    for (;;) {
    uint32_t next = 0;
    if (lane == 0) next = atomicAdd(cursor, 1u);
    next = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    if (next >= todo) break;
    const uint64_t gid = list ? list[next] : next; // ligand of this wave
    if (gid >= count) continue;
    if (status[gid] != PMX_LIGAND_OK) continue;
    const uint64_t off = taboff[gid];
    if (taboff[gid + 1] == off) continue;

    const Record r = parse_record(lib.data + lib.offsets[first + gid]);
    const int C = r.C;
    const int cc = c < C ? c : C - 1;
    const bool lane_live = c < C;
    const unsigned long long slot_mask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (s * G));
    const Levels L = scan_levels(r, M.tclus, [&](int lev, int start, int end, uint64_t cand, uint32_t k) {
        WL.cand[lev] = cand;
        WL.start[lev] = (uint8_t)start;
        WL.end[lev] = (uint8_t)end;
        WL.k[lev] = (uint8_t)k;
    });
    const int nl = L.nl;
    for (int lev = 0; lev < nl; ++lev) { // candidate slot -> cluster id
        uint64_t cand = WL.cand[lev];
        for (int q = 0; cand; cand &= cand - 1, ++q) candlist[lev * cs + (q)] = (uint8_t)(__ffsll((unsigned long long)cand) - 1);
    }
    for (int i = lane; i < kTabEntryChunk * G; i += 64) {
        acc_score[i] = 0.f;
        acc_fail[i] = 0u;
    }
    if (lane < r.n) tm[lane] = r.typemask[lane]; // looked up per item and per entry: one LDS read instead of a global load

    uint8_t *blk = arena + off;
    TabHeader *H = reinterpret_cast<TabHeader *>(blk);
    vmask_t<G> *Vt = reinterpret_cast<vmask_t<G> *>(blk + sizeof(TabHeader));
    float *St = reinterpret_cast<float *>(blk + sizeof(TabHeader) + round16(uint64_t(L.T) * sizeof(vmask_t<G>)));
    float *Pt = St + round16(uint64_t(L.ksumtot) * G * 4) / 4;
    H->nl = (uint32_t)nl;
    H->T = L.T;
    H->ksumtot = L.ksumtot;
    H->pad = 0;
    {
        uint32_t ks = 0, rb = 0;
        for (int i = 0; i < nl; ++i) {
            const uint32_t k = WL.k[i];
            H->k[i] = (uint8_t)k;
            H->ksum[i] = (uint16_t)ks;
            H->rowbase[i] = rb;
            ks += k;
            rb += k * (L.ksumtot - ks);
        }
        H->ksum[nl] = (uint16_t)ks;
    }
    wave_lds_sync();

    // the record base is the same for the whole wave: say so, and coordinate loads take it as a scalar base + 32-bit lane offset
    const uint64_t xyz_bits = reinterpret_cast<uint64_t>(r.xyz);
    const float *xyz = reinterpret_cast<const float *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xyz_bits >> 32)) << 32) |
                                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xyz_bits));
    uint32_t self_base = 0, pair_base = 0;
    for (int i = 0; i < nl; ++i) {
        const int si = WL.start[i], ni = WL.end[i] - si, ki = WL.k[i];
        Pos ctr_i;
        float size_i;
        cluster_center_size(xyz, C, si, si + ni, cc, ctr_i, size_i);

        // ---- self table S[i][a] (match_utils.py:77-122): items (a, u < v)
        for (int e0 = 0; e0 < ki; e0 += kTabEntryChunk) {
            const int ecur = min(kTabEntryChunk, ki - e0);
            if (ni > 1) {
                const int per = ni * ni; // walk all (u, v), keep u < v
                const float inv_per = 1.0f / (float)per, inv_ni = 1.0f / (float)ni;
                const int total = ecur * per;
                for (int t = s; t < total; t += GPW) {
                    const int e = (int)(((float)t + 0.5f) * inv_per), rr = t - __mul24(e, per);
                    const int u = (int)(((float)rr + 0.5f) * inv_ni), v = rr - __mul24(u, ni);
                    if (u >= v) continue;
                    const int a = candlist[i * cs + (e0 + e)];
                    const Pos pu = load_pos(xyz, C, si + u, cc), pv = load_pos(xyz, C, si + v, cc);
                    const unsigned tmu = tm[si + u], tmv = tm[si + v];
                    const uint4 la = M.clist[(uint32_t)(a * 128) + tmu], lb = M.clist[(uint32_t)(a * 128) + tmv];
                    float d = norm3(pu.x - pv.x, pu.y - pv.y, pu.z - pv.z);
                    asm volatile("" : "+v"(d)); // the distance is computed while the list loads are in flight (see the pair items)
                    float acc = 0.f;
                    int np = 0;
                    const int mn = cluster_node_pair(M, tab, cnodes, tnodes, a, a, tmu, tmv, la, lb, d, acc, np);
                    if (!mn) continue;
                    const bool zw = some_zero && (!(cnodes[a] & tnodes[tmu] & nzw) || !(cnodes[a] & tnodes[tmv] & nzw));
                    atomicAdd(&acc_score[e * G + c], zw ? __builtin_nanf("") : acc / (float)mn);
                }
            }
            wave_lds_sync();
            for (int e = s; e < ecur; e += GPW) {
                St[(size_t)(self_base + e0 + e) * G + c] = acc_score[e * G + c];
                acc_score[e * G + c] = 0.f;
            }
            wave_lds_sync();
        }
        self_base += ki;

        for (int j = i + 1; j < nl; ++j) {
            const int sj = WL.start[j], nj = WL.end[j] - sj, kj = WL.k[j];
            Pos ctr_j;
            float size_j;
            cluster_center_size(xyz, C, sj, sj + nj, cc, ctr_j, size_j);
            const float ldist = norm3(ctr_i.x - ctr_j.x, ctr_i.y - ctr_j.y, ctr_i.z - ctr_j.z); // graph_match.py:240
            const float lsize = size_i + size_j;                                                 // :241
            const int E = ki * kj, per = ni * nj;
            const float inv_per = 1.0f / (float)per, inv_nj = 1.0f / (float)nj, inv_kj = 1.0f / (float)kj;
            for (int e0 = 0; e0 < E; e0 += kTabEntryChunk) {
                const int ecur = min(kTabEntryChunk, E - e0);
                // cluster-distance prefilter (graph_match.py:263-268) once per entry: keep the entries for which
                // some conformer passes, as a compact list, so that items are only made for those
                int n_near = 0;
                for (int e1 = 0; e1 < ecur; e1 += GPW) {
                    const int e = e1 + s;
                    bool near = false;
                    uint32_t packed = 0; // the entry with its two model clusters: what an item needs, in one LDS word
                    if (e < ecur) {
                        const int ea = (int)(((float)(e0 + e) + 0.5f) * inv_kj), eb = (e0 + e) - __mul24(ea, kj);
                        const int a = candlist[i * cs + (ea)], b = candlist[j * cs + (eb)];
                        const float2 mp = M.cpair[a * M.K + b];
                        near = lane_live && !((fabsf(ldist - mp.x) - lsize) > mp.y);
                        packed = (uint32_t)e | ((uint32_t)a << 8) | ((uint32_t)b << 16);
                    }
                    const unsigned long long bal = __ballot(near);
                    // one bit per slot: does any conformer of that slot's entry pass
                    for (int q = 0; q < GPW; ++q) {
                        const unsigned long long m = (G == 64) ? bal : ((bal >> (q * G)) & ((1ull << G) - 1ull));
                        if (m && e1 + q < ecur) {
                            if (s == q) near_list[n_near] = packed;
                            ++n_near;
                        }
                    }
                }
                wave_lds_sync();
                const int total = n_near * per;
                for (int t = s; t < total; t += GPW) {
                    // (all products are far below 2^24: full-rate 24-bit multiplies)
                    const int en = (int)(((float)t + 0.5f) * inv_per), rr = t - __mul24(en, per);
                    const int u = (int)(((float)rr + 0.5f) * inv_nj), v = rr - __mul24(u, nj);
                    const uint32_t packed = near_list[en];
                    const int e = (int)(packed & 255u), a = (int)((packed >> 8) & 255u), b = (int)(packed >> 16);
                    // Latency order: the six coordinate loads go out first (they only need u and v), then the two node
                    // lists (which need the LDS lookups of a, b and the type masks); the distance is finished while the
                    // lists are in flight. Left to itself the compiler sinks the coordinate loads below the test on the
                    // lists, which makes three dependent memory round trips per item out of two.
                    const Pos pu = load_pos(xyz, C, si + u, cc), pv = load_pos(xyz, C, sj + v, cc);
                    const unsigned tmu = tm[si + u], tmv = tm[sj + v];
                    const uint4 la = M.clist[(uint32_t)(a * 128) + tmu], lb = M.clist[(uint32_t)(b * 128) + tmv];
                    float d = norm3(pu.x - pv.x, pu.y - pv.y, pu.z - pv.z);
                    asm volatile("" : "+v"(d));
                    float acc = 0.f;
                    int np = 0;
                    const int mn = cluster_node_pair(M, tab, cnodes, tnodes, a, b, tmu, tmv, la, lb, d, acc, np);
                    if (!mn) continue;
                    const bool zw = some_zero && (!(cnodes[a] & tnodes[tmu] & nzw) || !(cnodes[b] & tnodes[tmv] & nzw));
                    atomicAdd(&acc_score[e * G + c], zw ? __builtin_nanf("") : acc / (float)mn);
                    if (2 * np < mn) atomicAdd(&acc_fail[e * G + c], 1u); // num_pass < num_match * 0.5 (match_utils.py:61)
                }
                wave_lds_sync();
                // finish the entries of this chunk: slots take entries, lane c its conformer
                for (int e1 = 0; e1 < ecur; e1 += GPW) {
                    const int e = e1 + s;
                    const bool on = e < ecur;
                    float value = -1.f;
                    bool valid = false;
                    if (on) {
                        const int ea = (int)(((float)(e0 + e) + 0.5f) * inv_kj), eb = (e0 + e) - ea * kj;
                        const int a = candlist[i * cs + (ea)], b = candlist[j * cs + (eb)];
                        const float2 mp = M.cpair[a * M.K + b];
                        const bool near = lane_live && !((fabsf(ldist - mp.x) - lsize) > mp.y);
                        const bool near_any = (__ballot(near) & slot_mask) != 0;
                        int L1 = 0, L2 = 0; // ligand nodes with a compatible model node (graph_match.py:164-171)
                        for (int u = 0; u < ni; ++u) L1 += (cnodes[a] & tnodes[tm[si + u]]) ? 1 : 0;
                        for (int v = 0; v < nj; ++v) L2 += (cnodes[b] & tnodes[tm[sj + v]]) ? 1 : 0;
                        const float score = acc_score[e * G + c];
                        const int fails = (int)acc_fail[e * G + c];
                        acc_score[e * G + c] = 0.f;
                        acc_fail[e * G + c] = 0u;
                        if (near_any) {
                            valid = lane_live && (2 * fails <= L1 * L2) && (score > 0.f); // match_utils.py:71-74, tree.py:81
                            value = (2 * fails <= L1 * L2) ? score : -1.f;
                        }
                    }
                    const unsigned long long vbal = __ballot(valid);
                    if (on) {
                        const uint32_t idx = pair_base + (uint32_t)(e0 + e);
                        Pt[(size_t)idx * G + c] = value;
                        Vt[idx] = (vmask_t<G>)((G == 64) ? vbal : ((vbal >> (s * G)) & ((1ull << G) - 1ull)));
                    }
                }
                wave_lds_sync();
            }
            pair_base += (uint32_t)E;
        }
    }
    } // next ligand
}

// ---------------------------------------------------------------------------------- bounds_kernel
// Upper bounds for the tree search. The reported score only needs the per-conformer MAXIMUM over leaves
// (graph_match.py:103-109), so a subtree whose best possible leaf cannot exceed the maximum found so far
// may be dropped without changing the result, provided the walker's return-value bookkeeping does not
// need it - which holds for subtrees rooted at nodes with >= 5 matches (see tree_kernel).
// A leaf total is sum of S over the matched (level, candidate) picks + sum of P over pairs of picks.
// For conformer c, level l can add at most
//   U[l][c] = max(0, max_b ( S[l][b][c] + sum_{j < l} max(0, max_a P[(j, a), (l, b)][c]) ))
// whatever is picked on the other levels, so R[f][c] = sum_{l >= f} U[l][c] bounds everything levels
// f.. add to a node's total. One wavefront per ligand: lane c of group g takes candidates b = g, g + 64/G, ...
template <int G>
__global__ __launch_bounds__(256) void bounds_kernel(uint32_t count, const int32_t *status, const uint64_t *taboff, uint8_t *arena,
                                                      int no_bounds /* debug: write +inf, so that nothing is ever dropped */,
                                                      const uint32_t *list, const uint32_t *list_count) {
    constexpr int GPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, c = lane % G;
    uint32_t li = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (list) {
        if (li >= *list_count) return;
        li = list[li];
    }
    if (li >= count) return;
    if (status[li] != PMX_LIGAND_OK) return;
    const uint64_t o0 = taboff[li], o1 = taboff[li + 1];
    if (o1 == o0) return;
    uint8_t *blk = arena + o0;
    const TabHeader *H = reinterpret_cast<const TabHeader *>(blk);
    const int nl = (int)H->nl;
    const uint32_t T = H->T, ksumtot = H->ksumtot;
    const uint32_t v_bytes = (uint32_t)round16(uint64_t(T) * sizeof(vmask_t<G>));
    const uint32_t s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
    const uint32_t p_bytes = (uint32_t)round16(uint64_t(T) * G * 4);
    const unsigned char *tab = blk + sizeof(TabHeader);
    const float *St = reinterpret_cast<const float *>(tab + v_bytes);
    const float *Pt = reinterpret_cast<const float *>(tab + v_bytes + s_bytes);
    double *Rt = reinterpret_cast<double *>(blk + sizeof(TabHeader) + v_bytes + s_bytes + p_bytes);
    double suffix = 0.0;
    if (no_bounds) {
        for (int l = g; l <= nl; l += GPW) Rt[(size_t)l * G + c] = __builtin_inf();
        return;
    }
    if (g == 0) Rt[(size_t)nl * G + c] = 0.0;
    for (int l = nl - 1; l >= 0; --l) {
        const int kl = H->k[l], ksl = H->ksum[l];
        double u = 0.0;
        for (int b = g; b < kl; b += GPW) {
            double v = (double)St[(size_t)(ksl + b) * G + c];
            for (int j = 0; j < l; ++j) {
                const int kj = H->k[j];
                const uint32_t e0 = H->rowbase[j] + (uint32_t)kj * (uint32_t)(ksl - (int)H->ksum[j + 1]) + (uint32_t)b;
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
        for (int d = G; d < 64; d <<= 1) {
            const double o = __shfl_xor(u, d);
            u = o > u ? o : u;
        }
        suffix += u;
        if (g == 0) Rt[(size_t)l * G + c] = suffix;
    }
}

// ------------------------------------------------------------------------------------ tree_kernel
// Iterative form of ClusterMatchTree.dfs_run (tree.py:55-104). A frame f describes the tree node at
// level f - 1 (frame 0 = root): `todo` = existing candidate children of level f not yet explored,
// `mx` = max_num_matches so far, flags = {matched, any candidate child existed, skip child handled,
// expanded}. The matched ancestors of the current path are kept as a list (match q: table row offset,
// k_j, chosen candidate a_j, level j), and the conformer mask / float64 totals are indexed by the
// number of matches (a skip child shares its parent's). Entering a frame evaluates every candidate
// (f, b) against the matched ancestors at once (lanes of the group take different b):
//   mask(b)  = mask(parent) & AND_q V[entry(q, f, b)]                  (tree.py:78-84)
// and descending into an existing candidate adds
//   total(b) = total(parent) + S[f][b] + sum_q P[entry(q, f, b)]       (tree.py:38-41)
//
// One wavefront per ligand. The ligand's tables (V, S, P, R; ~11 KB at BASELINE shapes) stay in the arena and
// are read through L2 with 32-bit offsets from a wave-uniform base; the wave's 64 / G conformer groups walk
// different subtrees of the SAME tree, sharing its tables in cache.
//
// Work splitting. Trees are heavy-tailed (median ~1e3 nodes, tail > 1e7), and a tree walked by one
// group is a serial chain. The only coupling between sibling subtrees is the skip rule
// `num_matches + max_num_matches < 5` (tree.py:98). Since `num_matches(A) + max_num_matches(A)` is the
// largest match count of any leaf below A's candidate children, the rule only asks whether a node
// with >= 5 matches exists there. Hence a subtree rooted at a node Y with num_matches(Y) >= 5 can be
// cut out: every ancestor's decision is already settled by Y's existence (returning 1 for Y gives
// each ancestor A at least 5 - num_matches(A)), decisions inside the subtree depend on candidate
// existence only, and leaves only feed a per-conformer maximum. So a frame with >= 4 matches may give
// away unexplored candidate children:
//   * inside the wave: whenever a group is idle, busy groups push one child each onto a small LDS
//     stack and idle groups pop (ballot-ranked, no atomics);
//   * across waves: a job that exceeds its iteration budget appends all its open children to a global
//     task queue; tasks are run by the same kernel (TASKS = true) in rounds, splitting again when over
//     budget; per-conformer maxima of split ligands are combined with atomicMax in `bestbuf`
//     (non-negative doubles order as uint64);
// and it may DROP a child whose subtree cannot raise any conformer's maximum (bound test against the
// suffix bounds R of bounds_kernel): the maxima, hence the score, and by the argument above every skip
// decision stay what they were. Frames with < 4 matches need their children's real return values; their
// children are handed over inside the wave only, with a join (the frame counts the children it has out in
// `.x` and waits for the walkers' reports, see `rep`).
struct TaskHeader { // 64 bytes, followed by double tot[G]
    uint32_t lig;   // ligand index inside the chunk
    uint8_t f0;     // frame of the subtree's root
    uint8_t nm;     // matches on the path, root included
    uint8_t jg;     // in-wave hand-over from a frame with < 4 matches: group and frame that need the root's
    uint8_t jf;     // return value (tree.py:98); jg = 0xff otherwise
    uint64_t mask;  // conformer mask of the root
    uint8_t path[2 * PMX_MAX_LEVELS]; // (level, candidate) of every match on the path
    uint8_t pad2[8];
};
static_assert(sizeof(TaskHeader) == 64, "TaskHeader layout");

template <int G>
__host__ __device__ constexpr uint32_t task_bytes() {
    return sizeof(TaskHeader) + G * 8;
}

constexpr int kStatShards = 256;
constexpr int kQueueShards = 64; // = the wave size: a task wave finds its record with one scan over the shards' ranges

struct TreeParams {
    const uint8_t *arena;
    const uint64_t *taboff;
    const int32_t *status;
    DevLibrary lib;
    uint64_t first;      // library index of the chunk's first ligand
    uint32_t count;      // jobs: ligands (TASKS = false) or the tasks of this round (TASKS = true)
    // The task queue is kQueueShards independent queues (shard s owns records [s * qcap, (s + 1) * qcap)); a wave appends to
    // shard blockIdx % kQueueShards. One tail for the whole GPU was a single address taking ~10^6 returning atomics per
    // chunk: the exporting waves queued on it (12 % of the first-tier kernel with the appends already pooled per wave).
    uint32_t *qtail;     // [kQueueShards] tails
    uint32_t *qflag;     // set when a shard is full (the walker then keeps the subtree)
    uint8_t *queue;
    uint32_t qcap;       // records per shard
    uint32_t shard_lo[kQueueShards], shard_hi[kQueueShards]; // TASKS: this round's records of each shard
    unsigned long long *bestbuf; // [chunk][G]
    uint8_t *deferred;           // [chunk] ligand was split: score comes from bestbuf
    int depth_cap;
    int K;               // model clusters (bound on candidates per level)
    uint32_t step_cap;   // children / returns handled by one walker step at most
    uint32_t share_levels; // in-wave sharing hands over only subtrees with at least this many levels below their root
    uint32_t min_levels; // in export mode only subtrees with at least this many levels below their root are queued
    uint32_t flags;      // debug: 1 = no in-wave sharing, 2 = no global donation (4 = no bound test: see bounds_kernel)
    uint32_t budget;     // wave iterations after which a job donates its open subtrees to the queue
    unsigned long long *nsteps; // [kStatShards][4] {DFS steps, wave iterations, longest whole-ligand job, longest task}: diagnostics,
                                // sharded by block so that a million waves do not queue on three addresses
    uint32_t *dbg;       // [0] = error flag (iteration cap hit), then 8 words per group
    unsigned long long max_iters; // safety cap on wave iterations per job (a tree walk is finite; never spin forever)
    float *scores;
};

// LDS bytes of one conformer group's tree state for stacks that hold `depth` levels of a model with K clusters.
template <int G>
__host__ __device__ inline uint32_t tree_group_bytes(int depth, int K) {
    const uint32_t todo_bytes = (uint32_t)round16((uint64_t)(depth + 1) * 8);                      // unexplored existing candidates per frame
    const uint32_t cm_bytes = (uint32_t)round16((uint64_t)(depth + 1) * K * sizeof(vmask_t<G>));  // conformer masks of a frame's candidates
    const uint32_t msk_bytes = (uint32_t)round16((uint64_t)(depth + 1) * sizeof(vmask_t<G>));     // conformer masks by match count
    const uint32_t frm_bytes = (uint32_t)round16((uint64_t)(depth + 1) * 4);                      // frames {-, mx, flags, nm}
    const uint32_t mat_bytes = (uint32_t)round16((uint64_t)depth * 8);                            // matched ancestors
    const uint32_t eb_bytes = (uint32_t)round16((uint64_t)depth * 4);                             // their pair-table rows for the current frame
    return todo_bytes + cm_bytes + msk_bytes + frm_bytes + mat_bytes + eb_bytes;
}

// k[32], ksum[24], rowbase[20] of the job's ligand; one report word per group (LDS per wave decides how many waves a CU holds)
template <int G>
__host__ __device__ constexpr uint32_t tree_shared_hdr() {
    return 32 + 48 + 80 + (uint32_t)round16(4u * (64 / G));
}

template <int G>
__host__ __device__ constexpr uint32_t tree_local_stack_entries() {
    return 64 / G; // one entry per conformer group
}

// LDS bytes of one wave of tree_kernel.
template <int G>
__host__ __device__ inline uint32_t tree_wave_bytes(int depth, int K) {
    return tree_shared_hdr<G>() + tree_local_stack_entries<G>() * task_bytes<G>() +
           (64 / G) * tree_group_bytes<G>(depth, K);
}

// Pair-table index of (matched ancestor q, candidate 0 of level f): + b gives candidate b.
__device__ inline int entry_base(const int2 mq, int ksf, int kf) {
    return mq.x + (mq.y & 255) * ksf + ((mq.y >> 8) & 255) * kf;
}

// log2 of the bytes of one pair entry in the P table (G floats)
template <int G>
__host__ __device__ constexpr int p_entry_shift() {
    return G == 1 ? 2 : G == 2 ? 3 : G == 4 ? 4 : G == 8 ? 5 : G == 16 ? 6 : G == 32 ? 7 : 8;
}

// The same sum with the ancestors' entry bases of the current frame taken from the group's LDS cache
// (eb[q] = entry_base(mat[q], ksf, kf)): one shift-add per term, 32-bit offsets from the wave-uniform
// table base. bc = (b << p_entry_shift) + 4 * c.
template <int G>
__device__ inline double pair_sum_eb(const unsigned char *Pt, const int *eb, int nm, uint32_t bc) {
    constexpr int SH = p_entry_shift<G>();
    double pair = 0.0;
    int q = 0;
    for (; q + 8 <= nm; q += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float *>(Pt + (((uint32_t)eb[q + u] << SH) + bc));
#pragma unroll
        for (int u = 0; u < 8; ++u) pair += (double)v[u];
    }
    for (; q + 4 <= nm; q += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float *>(Pt + (((uint32_t)eb[q + u] << SH) + bc));
#pragma unroll
        for (int u = 0; u < 4; ++u) pair += (double)v[u];
    }
    for (; q < nm; ++q) pair += (double)*reinterpret_cast<const float *>(Pt + (((uint32_t)eb[q] << SH) + bc));
    return pair;
}

// sum_q P[entry(q, f, b)][c] in ancestor order (tree.py:78-82), loads issued four at a time
template <int G>
__device__ inline double pair_sum(const float *Pt, const int2 *mat, int nm, int ksf, int kf, int b, int c) {
    double pair = 0.0;
    int q = 0;
    for (; q + 8 <= nm; q += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Pt[(size_t)(entry_base(mat[q + u], ksf, kf) + b) * G + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) pair += (double)v[u];
    }
    for (; q + 4 <= nm; q += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = Pt[(size_t)(entry_base(mat[q + u], ksf, kf) + b) * G + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) pair += (double)v[u];
    }
    for (; q < nm; ++q) pair += (double)Pt[(size_t)(entry_base(mat[q], ksf, kf) + b) * G + c];
    return pair;
}

// Walks one job (a whole ligand tree, or a subtree task) with all conformer groups of the wave.
// The job's tables are read from the arena (L2); staging them in LDS was measured and costs more in
// occupancy than it saves in latency.
template <int G, bool TASKS>
#ifdef PMX_GUARDS
#define PMX_GUARD(code)                                                         \
    do {                                                                        \
        if (++guard > 50000000u) {                                              \
            p.dbg[0] = 2;                                                       \
            p.dbg[3] = (code);                                                  \
            return;                                                             \
        }                                                                       \
    } while (0)
#else
#define PMX_GUARD(code) (void)guard
#endif
__device__ __forceinline__ void run_job(const TreeParams &p, unsigned char *smem, const uint32_t li, const TaskHeader *task,
                        const uint8_t *blk) {
    using vm_t = vmask_t<G>;
    constexpr int GPW = 64 / G;
    constexpr int LCAP = (int)tree_local_stack_entries<G>();
    constexpr unsigned F_MATCHED = 1, F_ANY = 2, F_SKIP = 4, F_EXPANDED = 8;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, c = lane % G;
    const int D = p.depth_cap, K = p.K;
    const uint32_t budget = p.budget, qcap = p.qcap, flags = p.flags, min_levels = p.min_levels, share_levels = p.share_levels;
    const uint32_t step_cap = p.step_cap;
    const unsigned long long max_iters = p.max_iters;
    uint32_t *const qflag = p.qflag;
    uint8_t *const queue = p.queue;
    constexpr uint32_t kNoSlot = 0xffffffffu;
    const unsigned long long below = (g == 0) ? 0ull : ((1ull << (g * G)) - 1ull); // lanes of lower groups

    // ---- LDS carve
    unsigned char *shared = smem;
    uint8_t *hk = shared;                                               // k[32]
    uint16_t *hksum = reinterpret_cast<uint16_t *>(shared + 32);        // [24]
    uint32_t *hrow = reinterpret_cast<uint32_t *>(shared + 32 + 48);    // [20]
    uint32_t *rep = reinterpret_cast<uint32_t *>(shared + 32 + 48 + 80); // [GPW] return values of handed-over subtrees
    unsigned char *lstk = shared + tree_shared_hdr<G>();                    // local task stack
    const uint32_t todo_bytes = (uint32_t)round16((uint64_t)(D + 1) * 8);
    const uint32_t cm_bytes = (uint32_t)round16((uint64_t)(D + 1) * K * sizeof(vm_t));
    const uint32_t msk_bytes = (uint32_t)round16((uint64_t)(D + 1) * sizeof(vm_t));
    const uint32_t frm_bytes = (uint32_t)round16((uint64_t)(D + 1) * 4);
    unsigned char *base = lstk + LCAP * task_bytes<G>() + (size_t)g * tree_group_bytes<G>(D, K);
    // float64 totals by match count live in the wave's private (scratch) memory, not in LDS: they are read
    // and written next to global table loads anyway, and LDS per wave decides how many waves a CU holds
#ifndef PMX_TOT_WINDOW
#define PMX_TOT_WINDOW 0 // totals of the first PMX_TOT_WINDOW match counts live in LDS (0: all in private memory)
#endif
    constexpr int TW = PMX_TOT_WINDOW;
    double tot_hi[TW >= PMX_MAX_LEVELS + 1 ? 1 : PMX_MAX_LEVELS + 1 - TW];
    double *totw = reinterpret_cast<double *>(smem + tree_wave_bytes<G>(D, K)) + lane; // [TW][64]
    auto tget = [&](int nmq) -> double { return (TW > 0 && nmq < TW) ? totw[nmq * 64] : tot_hi[TW >= PMX_MAX_LEVELS + 1 ? 0 : nmq - TW]; };
    auto tset = [&](int nmq, double v) {
        if (TW > 0 && nmq < TW) totw[nmq * 64] = v;
        else tot_hi[TW >= PMX_MAX_LEVELS + 1 ? 0 : nmq - TW] = v;
    };
    uint64_t *todo = reinterpret_cast<uint64_t *>(base);                // [D + 1]
    vm_t *cm = reinterpret_cast<vm_t *>(base + todo_bytes);             // [D + 1][K]
    vm_t *msk = reinterpret_cast<vm_t *>(base + todo_bytes + cm_bytes); // [D + 1]
    uchar4 *frm = reinterpret_cast<uchar4 *>(base + todo_bytes + cm_bytes + msk_bytes); // [D + 1] {-, mx, flags, nm}
    int2 *mat = reinterpret_cast<int2 *>(base + todo_bytes + cm_bytes + msk_bytes + frm_bytes); // [D] {R, k_j | a_j << 8 | j << 16}
    // [D] entry_base(mat[q], level ebf) of the matched ancestors for the frame the group is working in
    int *eb = reinterpret_cast<int *>(__builtin_assume_aligned(
        base + todo_bytes + cm_bytes + msk_bytes + frm_bytes + (uint32_t)round16((uint64_t)D * 8), 16));
    int ebf = -1;
    int jg = -1, jf = 0, root_ret = 0; // who waits for this walker's root (see the joins below)

    uint32_t guard = 0;
#ifdef PMX_PROF // build with PMX_CXXFLAGS=-DPMX_PROF: where a wave's time goes (s_memtime) and what it does
    unsigned long long pc_share = 0, pc_expand = 0, pc_adv = 0, pc_all = 0, pt0 = 0, pt1 = 0;
    unsigned long long px_a = 0, px_b = 0, px_c = 0, px_d = 0, px_r = 0, px_n = 0, px_t = 0;
    unsigned pn_leaf = 0, pn_exp = 0, pn_desc = 0, pn_pl = 0, pn_vl = 0, pn_fill = 0, pn_ret = 0, pn_skip = 0, pn_exported = 0, pn_pruned = 0;
#define PROF(x) x
#else
#define PROF(x)
#endif
    // ---- the job's tables
    const TabHeader *H = reinterpret_cast<const TabHeader *>(blk);
    const int nl = __builtin_amdgcn_readfirstlane((int)H->nl);
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)H->T);
    const uint32_t ksumtot = (uint32_t)__builtin_amdgcn_readfirstlane((int)H->ksumtot);
    const uint32_t v_bytes = (uint32_t)round16(uint64_t(T) * sizeof(vm_t));
    const uint32_t s_bytes = (uint32_t)round16(uint64_t(ksumtot) * G * 4);
    const uint32_t p_bytes = (uint32_t)round16(uint64_t(T) * G * 4);
    const unsigned char *tab = blk + sizeof(TabHeader);
    const float *St = reinterpret_cast<const float *>(tab + v_bytes);
    const float *Pt = reinterpret_cast<const float *>(tab + v_bytes + s_bytes);
    const unsigned char *Pb = reinterpret_cast<const unsigned char *>(Pt); // wave-uniform: one job per wave
    const unsigned char *Vb = tab; // validity masks
    const unsigned char *Sb = reinterpret_cast<const unsigned char *>(St);
    const unsigned char *Rb = tab + v_bytes + s_bytes + p_bytes; // R[f][c]: what levels f.. can still add (bounds_kernel)
    constexpr int RSH = p_entry_shift<G>() + 1;                  // log2 bytes of one row of R (G doubles)
    constexpr double kBoundSlack = 1.0 + 1e-9; // covers the float64 rounding of the sums the bound is compared with
    constexpr int PSH = p_entry_shift<G>();
    constexpr int VSH = sizeof(vm_t) == 1 ? 0 : sizeof(vm_t) == 2 ? 1 : sizeof(vm_t) == 4 ? 2 : 3;
    for (int i = lane; i <= nl; i += 64) {
        PMX_GUARD(2);
        hksum[i] = H->ksum[i];
        if (i < nl) {
            hk[i] = H->k[i];
            hrow[i] = H->rowbase[i];
        }
    }
    if (lane < GPW) rep[lane] = 0;
    wave_lds_sync();

    // ---- walker state
    bool busy = false, exported = false, export_mode = false;
    int f = -1, f0 = 0, sfr = 1 << 20, sp = 0, C = 1;
    uint32_t iters = 0;
    unsigned long long nsteps = 0, total_iters = 0;
    double best = 0.0; // graph_match.py:104
    if (TASKS) best = __longlong_as_double((long long)p.bestbuf[(size_t)li * G + c]); // maxima of the ligand's finished walkers

    // can the subtree below a child (frame fr + 1, conformer mask m, total t) still raise a conformer's maximum?
    auto bound_of = [&](int fr) -> double { // R[fr + 1][c]: what the levels below a child of frame fr can still add
        return *reinterpret_cast<const double *>(Rb + (((uint32_t)(fr + 1) << RSH) + 8u * (uint32_t)c));
    };
    auto may_improve_r = [&](vm_t m, double t, double r) -> bool {
        const unsigned long long bal = __ballot(((m >> c) & 1) && (t + r) * kBoundSlack > best);
        return ((G == 64) ? bal : ((bal >> (g * G)) & ((1ull << G) - 1ull))) != 0;
    };
    auto may_improve = [&](int fr, vm_t m, double t) -> bool { return may_improve_r(m, t, bound_of(fr)); };
    // start a walker on the subtree described by a task record (global queue or local stack)
    auto adopt = [&](const TaskHeader *th) {
        const int nm0 = th->nm;
        f0 = f = th->f0;
        sfr = f;
        for (int q = c; q < nm0; q += G) {
            const int j = th->path[2 * q], a = th->path[2 * q + 1];
            const int kj = hk[j];
            mat[q] = make_int2((int)hrow[j] - kj * (int)hksum[j + 1], kj | (a << 8) | (j << 16));
        }
        tset(nm0, reinterpret_cast<const double *>(th + 1)[c]);
        msk[nm0] = (vm_t)th->mask;
        frm[f] = make_uchar4(0, 0, F_MATCHED, (unsigned char)nm0);
        ebf = -1;
        jg = th->jg == 0xff ? -1 : (int)th->jg;
        jf = th->jf;
        // a subtree with >= 5 matches at its root that can no longer raise any maximum (the maxima may have grown
        // since it was handed over) is not walked at all
        busy = nm0 < 5 || may_improve(f - 1, (vm_t)th->mask, tget(nm0));
    };
    // describe candidate b of frame fr (conformer mask m) as a task record
    auto describe = [&](TaskHeader *th, int fr, int nmr, int b, vm_t m, double t, bool joined) {
        th->lig = li;
        th->jg = joined ? (uint8_t)g : (uint8_t)0xff;
        th->jf = (uint8_t)fr;
        th->f0 = (uint8_t)(fr + 1);
        th->nm = (uint8_t)(nmr + 1);
        th->mask = (uint64_t)m;
        for (int q = 0; q < nmr; ++q) {
            const int y = mat[q].y;
            th->path[2 * q] = (uint8_t)(y >> 16);
            th->path[2 * q + 1] = (uint8_t)(y >> 8);
        }
        th->path[2 * nmr] = (uint8_t)fr;
        th->path[2 * nmr + 1] = (uint8_t)b;
        reinterpret_cast<double *>(th + 1)[c] = t;
    };
    // total of candidate b of frame fr: parent + self + accumulated pair (tree.py:38-41); any frame, slow path
    auto child_total = [&](int fr, int nmr, int b) -> double {
        const int kr = hk[fr], ksr = hksum[fr];
        return tget(nmr) + (double)St[(size_t)(ksr + b) * G + c] + pair_sum<G>(Pt, mat, nmr, ksr, kr, b, c);
    };
    // a frame may give children away once it has >= 4 matches (their subtrees hold >= 5, see above)
    // One queue slot for every group that is exporting a subtree at this point of the program: the groups that got here
    // together share one atomic on the queue tail (it is a single address for the whole GPU, ~10^6 exports per chunk)
    // The shard a wave appends to: its own (blockIdx mod kQueueShards) for 64 wave iterations at a time, then the next one -
    // a wave's tasks stay together (consecutive task waves work on one ligand's tables) and a monster tree that exports for
    // tens of thousands of iterations fills all shards, not one. (Derived from the iteration count, which is live anyway:
    // the walker has no register to spare for a cursor.)
    auto my_shard = [&]() -> uint32_t { return (blockIdx.x + (uint32_t)(total_iters >> 6)) & (kQueueShards - 1); };
    auto reserve_slot = [&]() -> uint32_t { // record index in the whole queue, or kNoSlot when the shard is full
        const unsigned long long heads = __ballot(c == 0); // lane 0 of every group present
        const int leader = __ffsll(heads) - 1;
        const uint32_t sh = my_shard();
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(p.qtail + sh, (uint32_t)__popcll(heads));
        base = __shfl(base, leader);
        const uint32_t slot = base + (uint32_t)__popcll(heads & below);
        return slot < qcap ? sh * qcap + slot : kNoSlot;
    };
    auto donatable = [&](int fr) -> bool {
        const uchar4 Fr = frm[fr];
        return Fr.w >= 4 && (Fr.z & F_EXPANDED) && todo[fr] != 0;
    };

    // Enter frame fr: evaluate every candidate of level fr against the matched ancestors (tree.py:78-84);
    // lane c takes candidates c, c + G, ...; mask(b) = mask(parent) & AND_q V[entry(q, fr, b)].
    // pair-table rows of the matched ancestors against level fr, one ancestor per lane of the group
    auto fill_eb = [&](int fr, int nmr) {
        const int kr = hk[fr], ksr = hksum[fr];
        for (int q = c; q < nmr; q += G) eb[q] = entry_base(mat[q], ksr, kr);
        ebf = fr;
        wave_lds_sync();
    };
    auto expand = [&](int fr) {
        PROF(const unsigned long long e0 = __builtin_amdgcn_s_memtime());
        uchar4 Fr = frm[fr];
        const int nmr = Fr.w;
        const int kr = hk[fr];
        PROF(++pn_exp; pn_vl += (unsigned)(nmr * ((kr + G - 1) / G)));
        uint64_t E = 0;
        const vm_t pm = msk[nmr];
        fill_eb(fr, nmr);
        for (int b0 = 0; b0 < kr; b0 += G) {
            const int b = b0 + c;
            const bool on = b < kr;
            const int bb = on ? b : 0;
            vm_t m = on ? pm : (vm_t)0;
            const uint32_t bv = (uint32_t)bb << VSH;
            auto vld = [&](int q) -> vm_t { return *reinterpret_cast<const vm_t *>(Vb + (((uint32_t)eb[q] << VSH) + bv)); };
            int q = 0;
            for (; q + 8 <= nmr; q += 8) {
                vm_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = vld(q + u);
#pragma unroll
                for (int u = 0; u < 8; ++u) m &= v[u];
            }
            for (; q + 4 <= nmr; q += 4) {
                const vm_t v0 = vld(q), v1 = vld(q + 1), v2 = vld(q + 2), v3 = vld(q + 3);
                m &= (vm_t)(v0 & v1 & v2 & v3);
            }
            for (; q < nmr; ++q) m &= vld(q);
            if (on) cm[fr * K + b] = m;
            const unsigned long long bal = __ballot(on && m != 0);
            E |= ((G == 64) ? bal : ((bal >> (g * G)) & ((1ull << G) - 1ull))) << b0;
        }
        wave_lds_sync();
        if (fr + 1 == nl) {
            // The children are leaves: finish the whole frame here. Each existing candidate feeds the
            // per-conformer maximum (graph_match.py:105-108) and returns 1; the skip leaf (tree.py:98-101,
            // :42-43) carries this node's totals and returns 0; then return to the parent (tree.py:102).
            const double tp = tget(nmr);
            const int ksr = hksum[fr];
            uint64_t left = E;
            while (left) {
                const int b = __ffsll((unsigned long long)left) - 1;
                left &= left - 1;
                const uint32_t bc = ((uint32_t)b << PSH) + 4u * (uint32_t)c;
                const float self = *reinterpret_cast<const float *>(Sb + (((uint32_t)ksr << PSH) + bc));
                const double t = tp + (double)self + pair_sum_eb<G>(Pb, eb, nmr, bc);
                if (((cm[fr * K + b] >> c) & 1) && t > best) best = t;
                PROF(++pn_leaf; pn_pl += (unsigned)nmr);
            }
            const int mx = E ? 1 : 0;
            if (!E || nmr + mx < 5) {
                if (((pm >> c) & 1) && tp > best) best = tp;
            }
            const unsigned char ret = (unsigned char)(mx + ((Fr.z & F_MATCHED) ? 1 : 0));
            f = fr - 1;
            if (f >= f0) {
                uchar4 Pf = frm[f];
                Pf.y = Pf.y > ret ? Pf.y : ret;
                frm[f] = Pf;
            } else {
                root_ret = ret;
            }
            PROF(++pn_ret);
        } else {
            todo[fr] = E;
            Fr.z |= F_EXPANDED | (E ? F_ANY : 0);
            frm[fr] = Fr;
        }
        PROF(pc_expand += __builtin_amdgcn_s_memtime() - e0);
    };

    if (g == 0) {
        if (TASKS) {
            adopt(task);
        } else {
            C = parse_record(p.lib.data + p.lib.offsets[p.first + li]).C;
            f0 = f = 0; // root frame
            sfr = 0;
            tset(0, 0.0);
            msk[0] = (vm_t)((C >= 64) ? ~0ull : ((1ull << C) - 1ull));
            frm[0] = make_uchar4(0, 0, 0, 0);
            busy = true;
        }
    }
    wave_lds_sync();

    for (;;) {
        PMX_GUARD(12);
        PROF(pt0 = __builtin_amdgcn_s_memtime());
        const unsigned long long busy_bal = __ballot(busy && c == 0);
        sp = __builtin_amdgcn_readfirstlane(sp);
        if (!busy_bal && sp == 0) break;
        const int n_busy = __popcll(busy_bal), n_idle = GPW - n_busy;
        nsteps += (unsigned)n_busy;
        if (++total_iters > max_iters) { // cannot happen for a finite tree; report instead of spinning
            if (c == 0 && g < 8) { // the host decodes 8 groups; meta has room for no more
                uint32_t *d = p.dbg + 16 + g * 8;
                d[0] = li; d[1] = busy; d[2] = (uint32_t)f; d[3] = (uint32_t)f0; d[4] = (uint32_t)sp; d[5] = (uint32_t)sfr;
                d[6] = busy && f >= 0 ? *reinterpret_cast<uint32_t *>(&frm[f]) : 0u; d[7] = busy && f >= 0 ? (uint32_t)todo[f] : 0u;
                p.dbg[0] = 1; p.dbg[1] = (uint32_t)nl; p.dbg[2] = (uint32_t)n_busy;
            }
            break;
        }

        // ---- in-wave work sharing
        if (GPW > 1 && n_idle > 0 && !(flags & 1)) {
            const int need = n_idle - sp;
            if (need > 0 && n_busy > 0) {
                bool can = false;
                if (busy) {
                    // skip frames that have no child left to give away
                    while (sfr <= f && sfr < nl) {
                        PMX_GUARD(5);
                        if ((frm[sfr].z & F_EXPANDED) && todo[sfr] == 0) ++sfr;
                        else break;
                    }
                    // children too close to the leaves are cheaper to walk than to hand over
                    can = sfr <= f && sfr < nl && nl - (sfr + 1) >= (int)share_levels && (frm[sfr].z & F_EXPANDED) && todo[sfr] != 0;
                }
                const unsigned long long don_bal = __ballot(can && c == 0);
                const int rank = __popcll(don_bal & below);
                const int room = LCAP - sp;
                const int take = min(min(need, room), (int)__popcll(don_bal));
                if (can && rank < take) {
                    uchar4 Fr = frm[sfr];
                    const uint64_t left = todo[sfr];
                    const int b = __ffsll((unsigned long long)left) - 1;
                    todo[sfr] = left & (left - 1);
                    // With >= 4 matches here the child's return value is settled (at least 1, see above). Below that
                    // the skip rule (tree.py:98) needs the real value: the frame counts the children it has out
                    // (.x) and waits for their reports before it decides.
                    const bool joined = Fr.w < 4;
                    describe(reinterpret_cast<TaskHeader *>(lstk + (size_t)(sp + rank) * task_bytes<G>()), sfr, Fr.w, b, cm[sfr * K + b],
                             child_total(sfr, Fr.w, b), joined);
                    if (joined) Fr.x = (unsigned char)(Fr.x + 1);
                    else Fr.y = Fr.y > 1 ? Fr.y : 1;
                    frm[sfr] = Fr;
                }
                sp = __builtin_amdgcn_readfirstlane(sp + take);
                wave_lds_sync();
            }
            if (sp > 0) {
                // the groups walk parts of one tree: pool the maxima found so far (tightens the bound test)
#pragma unroll
                for (int d = G; d < 64; d <<= 1) {
                    const double o = __shfl_xor(best, d);
                    best = o > best ? o : best;
                }
                const unsigned long long idle_bal = __ballot(!busy && c == 0);
                const int rank = __popcll(idle_bal & below);
                const int npop = min(sp, n_idle);
                if (!busy && rank < npop) adopt(reinterpret_cast<const TaskHeader *>(lstk + (size_t)(sp - 1 - rank) * task_bytes<G>()));
                sp = __builtin_amdgcn_readfirstlane(sp - npop);
                wave_lds_sync();
            }
        }

        // ---- over budget: hand every open subtree (and the local stack) to the global queue
        if (++iters > budget && !(flags & 2)) {
            iters = 0;
            bool gave = false;
            if (busy) {
                for (int fr = max(f0, sfr); fr <= f && fr < nl; ++fr) {
                    PMX_GUARD(6);
                    if (!donatable(fr)) continue;
                    uchar4 Fr = frm[fr];
                    uint64_t left = todo[fr];
                    while (left) {
                        PMX_GUARD(7);
                        const uint32_t slot = reserve_slot();
                        if (slot == kNoSlot) { // queue full: the walker keeps the rest
                            if (c == 0) *qflag = 1;
                            break;
                        }
                        const int b = __ffsll((unsigned long long)left) - 1;
                        describe(reinterpret_cast<TaskHeader *>(queue + (size_t)slot * task_bytes<G>()), fr, Fr.w, b, cm[fr * K + b],
                                 child_total(fr, Fr.w, b), false);
                        left &= left - 1;
                        gave = true;
                        Fr.y = Fr.y > 1 ? Fr.y : 1;
                    }
                    todo[fr] = left;
                    frm[fr] = Fr;
                }
            }
            // local stack -> global queue (group e copies entry e, 16 bytes per lane and pass)
            for (int e0 = 0; e0 < sp; e0 += GPW) {
                PMX_GUARD(8);
                const int e = e0 + g;
                uint32_t slot = 0;
                const uint32_t sh = my_shard();
                if (e < sp && c == 0) slot = atomicAdd(p.qtail + sh, 1u);
                slot = __shfl(slot, g * G);
                if (e < sp) {
                    if (slot < qcap) {
                        slot += sh * qcap;
                        const uint4 *src = reinterpret_cast<const uint4 *>(lstk + (size_t)e * task_bytes<G>());
                        uint4 *dst = reinterpret_cast<uint4 *>(queue + (size_t)slot * task_bytes<G>());
                        for (uint32_t w = c; w < task_bytes<G>() / 16; w += G) dst[w] = src[w];
                        gave = true;
                    } else if (c == 0) {
                        *qflag = 1; // cannot happen in practice: entry is lost only if the queue is full
                    }
                }
            }
            if (__ballot(gave)) exported = true;
            // From now on children that may be given away (>= 5 matches) go straight to the queue when the
            // walker reaches them, so this wave only finishes the part of the tree it cannot split.
            export_mode = true;
            // entries that did not fit stay on the local stack only if the queue was full; keep them
            if (!__builtin_amdgcn_readfirstlane((int)*qflag)) sp = 0;
        }

        PROF(pt1 = __builtin_amdgcn_s_memtime(); pc_share += pt1 - pt0);
        // ---- one DFS step per busy group: advance to (and through) the next frame expansion.
        // Leaf children are consumed inside their parent's step; a step ends when a new frame has been
        // entered and its candidates evaluated, or when the group's subtree is finished.
        const bool was_busy = busy;
        if (busy) {
            // a step ends in one frame expansion; all groups of the wave do theirs together, below
            bool need_exp = f < nl && !(frm[f].z & F_EXPANDED); // first step of a root (whole tree or adopted subtree)
            if (!need_exp) {
                for (uint32_t inner = 0; inner < step_cap; ++inner) { // optional bound on the work of one step (PMX_STEP_CAP)
                    PMX_GUARD(13);
                    PROF(px_t = __builtin_amdgcn_s_memtime(); ++px_n);
                    uchar4 F = frm[f];
                    const int nm = F.w;
                    const bool matched = F.z & F_MATCHED;
                    if (f == nl) { // a subtree root that is itself a leaf (tree.py:103-104)
                        const double t = tget(nm);
                        if (((msk[nm] >> c) & 1) && t > best) best = t;
                        const unsigned char ret = matched ? 1 : 0;
                        --f;
                        if (f < f0) {
                            root_ret = ret;
                            break;
                        }
                        uchar4 Pf = frm[f];
                        Pf.y = Pf.y > ret ? Pf.y : ret;
                        frm[f] = Pf;
                        continue;
                    }
                    const uint64_t left = todo[f];
                    if (left) { // next existing candidate child (tree.py:94-97)
                        const int kf = hk[f], ksf = hksum[f];
                        const int b = __ffsll((unsigned long long)left) - 1;
                        todo[f] = left & (left - 1);
                        const vm_t m = cm[f * K + b];
                        if (ebf != f) {
                            fill_eb(f, nm); // back from a deeper frame
                            PROF(++pn_fill);
                        }
                        PROF(pn_pl += (unsigned)nm);
                        PROF({ const unsigned long long z = __builtin_amdgcn_s_memtime(); px_a += z - px_t; px_t = z; });
                        // parent + self + accumulated pair (tree.py:38-41)
                        const uint32_t bc = ((uint32_t)b << PSH) + 4u * (uint32_t)c;
                        // the bound row goes out with the table loads (one memory round trip per child, not two); frames
                        // f < nl only, so row f + 1 exists
                        const double rbound = bound_of(f);
                        const float self = *reinterpret_cast<const float *>(Sb + (((uint32_t)ksf << PSH) + bc));
                        const double t = tget(nm) + (double)self + pair_sum_eb<G>(Pb, eb, nm, bc);
                        PROF({ asm volatile("" :: "v"(t)); const unsigned long long z = __builtin_amdgcn_s_memtime(); px_b += z - px_t; px_t = z; });
                        if (nm >= 4) { // the child holds >= 5 matches: its subtree is a pure enumeration of leaves
                            if (!may_improve_r(m, t, rbound)) { // no leaf below can exceed the maxima found so far
                                F.y = F.y > 1 ? F.y : 1; // the dropped child returns at least 1
                                frm[f] = F;
                                PROF(++pn_pruned);
                                PROF({ const unsigned long long z = __builtin_amdgcn_s_memtime(); px_c += z - px_t; px_t = z; });
                                continue;
                            }
                            if (export_mode && nl - (f + 1) >= (int)min_levels) {
                                const uint32_t slot = reserve_slot();
                                if (slot != kNoSlot) {
                                    describe(reinterpret_cast<TaskHeader *>(queue + (size_t)slot * task_bytes<G>()), f, nm, b, m, t, false);
                                    F.y = F.y > 1 ? F.y : 1; // the child given away returns at least 1
                                    frm[f] = F;
                                    exported = true;
                                    PROF(++pn_exported);
                                    continue;
                                }
                                if (c == 0) *qflag = 1; // queue full: walk it here
                            }
                        }
                        tset(nm + 1, t);
                        msk[nm + 1] = m;
                        // entry(this match, level f', b') = rowbase[f] + k_f * (ksum[f'] - ksum[f + 1]) + b * k_f' + b'
                        mat[nm] = make_int2((int)hrow[f] - kf * (int)hksum[f + 1], kf | (b << 8) | (f << 16));
                        ++f;
                        frm[f] = make_uchar4(0, 0, F_MATCHED, (unsigned char)(nm + 1));
                        if (f < sfr) sfr = f;
                        PROF(++pn_desc);
                        PROF({ const unsigned long long z = __builtin_amdgcn_s_memtime(); px_d += z - px_t; px_t = z; });
                        need_exp = true;
                        break;
                    }
                    if (F.x) break; // children walked by other groups have not reported yet: wait for their return values
                    if (!(F.z & F_SKIP) && (!(F.z & F_ANY) || (nm + F.y) < 5)) { // skip child (tree.py:98-101)
                        F.z |= F_SKIP;
                        frm[f] = F;
                        ++f;
                        frm[f] = make_uchar4(0, 0, 0, (unsigned char)nm);
                        if (f < sfr) sfr = f;
                        PROF(++pn_skip);
                        need_exp = true;
                        break;
                    }
                    // all children done: return max_num_matches + matched (tree.py:102)
                    const unsigned char ret = (unsigned char)(F.y + (matched ? 1 : 0));
                    PROF(++pn_ret);
                    --f;
                    if (f < f0) {
                        root_ret = ret;
                        break;
                    }
                    uchar4 Pf = frm[f];
                    Pf.y = Pf.y > ret ? Pf.y : ret;
                    frm[f] = Pf;
                    PROF({ const unsigned long long z = __builtin_amdgcn_s_memtime(); px_r += z - px_t; px_t = z; });
                }
            }
            if (need_exp) expand(f);
            if (f < f0) busy = false;
        }
        // ---- joins: a walker whose root was handed over by a frame with < 4 matches reports the root's return
        // value; the frame's group takes it into max_num_matches (tree.py:96-97) and stops waiting for it
        if (GPW > 1) {
            const bool post = was_busy && !busy && jg >= 0;
            if (__ballot(post)) {
                if (post && c == 0) rep[g] = 0x80000000u | (uint32_t)jg | ((uint32_t)jf << 8) | ((uint32_t)root_ret << 16);
                wave_lds_sync();
                for (int m = 0; m < GPW; ++m) {
                    const uint32_t r = rep[m];
                    if ((r >> 31) && (int)(r & 255u) == g) {
                        const int fr = (int)((r >> 8) & 255u);
                        const unsigned char ret = (unsigned char)((r >> 16) & 255u);
                        uchar4 Fr = frm[fr];
                        Fr.x = (unsigned char)(Fr.x - 1);
                        Fr.y = Fr.y > ret ? Fr.y : ret;
                        frm[fr] = Fr;
                    }
                }
                wave_lds_sync();
                if (post) {
                    if (c == 0) rep[g] = 0;
                    jg = -1;
                }
                wave_lds_sync();
            }
        }
        PROF(const unsigned long long pt2 = __builtin_amdgcn_s_memtime(); pc_adv += pt2 - pt1; pc_all += pt2 - pt0);
    }
#ifdef PMX_PROF
    {
        unsigned long long *pr = reinterpret_cast<unsigned long long *>(p.dbg + 96) + (TASKS ? 16 : 0);
        if (lane == 0) {
            atomicAdd(pr + 0, pc_all); atomicAdd(pr + 1, pc_share); atomicAdd(pr + 2, pc_expand); atomicAdd(pr + 3, pc_adv);
            atomicAdd(pr + 4, total_iters);
            unsigned long long *px = reinterpret_cast<unsigned long long *>(p.dbg + 160) + (TASKS ? 8 : 0);
            atomicAdd(px + 0, px_a); atomicAdd(px + 1, px_b); atomicAdd(px + 2, px_c); atomicAdd(px + 3, px_d); atomicAdd(px + 4, px_r); atomicAdd(px + 5, px_n);
        }
        if (c == 0) {
            atomicAdd(pr + 5, (unsigned long long)pn_leaf); atomicAdd(pr + 6, (unsigned long long)pn_exp);
            atomicAdd(pr + 7, (unsigned long long)pn_desc); atomicAdd(pr + 8, (unsigned long long)pn_pl);
            atomicAdd(pr + 9, (unsigned long long)pn_vl); atomicAdd(pr + 10, (unsigned long long)pn_fill);
            atomicAdd(pr + 11, (unsigned long long)pn_ret); atomicAdd(pr + 12, (unsigned long long)pn_skip);
            atomicAdd(pr + 13, (unsigned long long)pn_exported); atomicAdd(pr + 14, (unsigned long long)pn_pruned);
        }
    }
#endif

    // ---- combine the groups' per-conformer maxima; lanes of group 0 end up with the wave's maxima
#pragma unroll
    for (int d = G; d < 64; d <<= 1) {
        const double o = __shfl_xor(best, d);
        best = o > best ? o : best;
    }
    if (lane == 0) {
        unsigned long long *st = p.nsteps + 4 * (blockIdx.x & (kStatShards - 1));
        atomicAdd(st, nsteps);
        atomicAdd(st + 1, total_iters);
        atomicMax(st + (TASKS ? 3 : 2), total_iters); // longest job of the launch (tail diagnostics)
    }
    exported = __ballot(exported) != 0;
    if (TASKS || exported) { // split ligand: combine across waves, score comes from finalize_kernel
        if (g == 0 && best > 0.0) atomicMax(&p.bestbuf[(size_t)li * G + c], (unsigned long long)__double_as_longlong(best));
        if (!TASKS && lane == 0) p.deferred[li] = 1;
    } else { // mean over conformers (graph_match.py:109); idle lanes hold 0
        C = __shfl(C, 0);
        double s = best;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) s += __shfl_xor(s, d);
        if (lane == 0) p.scores[li] = (float)(s / (double)C);
    }
}

#undef PMX_GUARD
#undef PROF

// One block (= one wavefront) per job, no persistent fetch loop: the hardware dispatcher hands the next
// block to whichever CU frees a slot, which is the dynamic load balancing a work counter would give,
// and the kernel stays a straight line of wave-uniform branches around the walker.
template <int G, bool TASKS>
#define PMX_TREE_WAVES_DEFAULT 6
#ifndef PMX_TASK_WAVES
#define PMX_TASK_WAVES 4 // the task kernel: <= 128 VGPRs, nothing spilled - its short jobs gain more from that than from a fifth and sixth wave (98 vs 119 ms per pass; 3 is the same, 5 in between)
#endif
#ifndef PMX_TREE_WAVES
#define PMX_TREE_WAVES PMX_TREE_WAVES_DEFAULT // waves per SIMD the register budget is set for: 6 -> <= 80 VGPRs, measured best (5 and 8 are slower)
#endif
__global__ __launch_bounds__(64, TASKS ? PMX_TASK_WAVES : PMX_TREE_WAVES) void tree_kernel(const TreeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t nx = blockIdx.x;
    if (nx >= p.count) return;
    const TaskHeader *task = nullptr;
    uint32_t li = nx;
    int status = PMX_LIGAND_OK;
    if (TASKS) {
        // block nx -> (shard, record): inclusive scan of the shards' record counts over the lanes
        const uint32_t lo_l = p.shard_lo[lane], cnt_l = p.shard_hi[lane] - lo_l;
        uint32_t inc = cnt_l;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        const int sh = __popcll(__ballot(nx >= inc)); // shards wholly before block nx (inc is non-decreasing)
        const uint32_t before = sh ? (uint32_t)__shfl(inc, sh - 1) : 0u;
        const uint32_t rec = (uint32_t)__shfl(lo_l, sh) + (nx - before);
        const uint64_t tbits = reinterpret_cast<uint64_t>(p.queue + ((size_t)sh * p.qcap + rec) * task_bytes<G>());
        task = reinterpret_cast<const TaskHeader *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tbits >> 32)) << 32) |
                                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tbits)); // wave-uniform
        li = __builtin_amdgcn_readfirstlane(task->lig);
    } else {
        status = __builtin_amdgcn_readfirstlane(p.status[li]);
    }
    // table offsets in 16-byte units fit 32 bits (arena < 64 GB)
    const uint32_t u0 = __builtin_amdgcn_readfirstlane((uint32_t)(p.taboff[li] >> 4));
    const uint32_t u1 = __builtin_amdgcn_readfirstlane((uint32_t)(p.taboff[li + 1] >> 4));
    const uint32_t bytes = (u1 - u0) * 16u;
    const uint8_t *blk = p.arena + (size_t)u0 * 16;
    if (status != PMX_LIGAND_OK) {
        if (lane == 0) p.scores[li] = __builtin_nanf("");
    } else if (bytes == 0) { // no ligand cluster has a candidate (graph_match.py:95-99)
        if (!TASKS && lane == 0) p.scores[li] = 0.f;
    } else {
        run_job<G, TASKS>(p, smem, li, task, blk);
    }
}

// Scores of ligands that were split into tasks: mean over conformers of the combined maxima.
template <int G>
__global__ void finalize_kernel(DevLibrary lib, uint64_t first, uint32_t count, const uint8_t *deferred,
                                const unsigned long long *bestbuf, float *scores) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || !deferred[i]) return;
    const int C = parse_record(lib.data + lib.offsets[first + i]).C;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += __longlong_as_double((long long)bestbuf[(size_t)i * G + c]);
    scores[i] = (float)(s / (double)C);
}

// Zeroes the per-chunk device state. A kernel (not hipMemsetAsync) so that everything in the scoring
// stream is ordered kernel -> kernel: memsets queued between two dependent kernels were observed to let
// the second kernel start before the first had finished on ROCm 7.2 / gfx950.
__global__ void clear_kernel(uint32_t *meta, uint32_t meta_words, unsigned long long *bestbuf, uint64_t best_words,
                             uint8_t *deferred, uint32_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < meta_words) meta[i] = 0;
    if (i < best_words) bestbuf[i] = 0;
    if (i < n) deferred[i] = 0;
}

__global__ void set_word_kernel(uint32_t *word, uint32_t value) { *word = value; }

// -------------------------------------------------------------------------------- library stats
// Also validates every record (a truncated or corrupt library must not make the scoring kernels read out of bounds): the
// header-implied size has to fit the record's byte range, cluster ends have to be monotonic and <= n_nodes, type masks
// <= 127. A record that fails is neutralised in the device copy (header zeroed -> PMX_LIGAND_UNSUPPORTED, score NaN) and
// counted; offsets that are not multiples of 16 or run backwards make the upload fail (out[5]).
__global__ void library_stats_kernel(DevLibrary lib, uint8_t *data_rw, uint64_t nbytes,
                                     unsigned long long *out /* [0] conformers [1] maxn [2] maxC [3] maxcl [4] unsupported [5] bad offsets [6] corrupt */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long conf = 0, maxn = 0, maxc = 0, maxcl = 0, unsupported = 0, bad_offsets = 0, corrupt = 0;
    if (i < lib.n) {
        const uint64_t o0 = lib.offsets[i], o1 = lib.offsets[i + 1];
        if ((o0 & 15) || o1 < o0 + 8 || o1 > nbytes) {
            bad_offsets = 1;
        } else {
            Record r = parse_record(lib.data + o0);
            const uint64_t need = ((8ull + (uint64_t)r.n + (uint64_t)r.ncl + 3ull) & ~3ull) + 12ull * (uint64_t)r.n * (uint64_t)r.C;
            bool ok = need <= o1 - o0;
            if (ok && record_supported(r)) {
                int prev = 0;
                for (int q = 0; q < r.ncl && ok; ++q) {
                    const int e = r.cluster_end[q];
                    ok = e >= prev && e <= r.n;
                    prev = e;
                }
                for (int u = 0; u < r.n && ok; ++u) ok = r.typemask[u] < 128;
            }
            if (!ok) { // neutralise: 0 nodes, 0 conformers, 0 clusters
                *reinterpret_cast<uint64_t *>(data_rw + o0) = 0ull;
                corrupt = 1;
                unsupported = 1;
            } else {
                conf = (unsigned long long)r.C;
                maxn = (unsigned long long)r.n;
                maxc = (unsigned long long)r.C;
                maxcl = (unsigned long long)r.ncl;
                if (!record_supported(r)) unsupported = 1;
            }
        }
    }
    // one set of atomics per wavefront, not per ligand
    conf = wave_sum(conf);
    maxn = wave_max(maxn);
    maxc = wave_max(maxc);
    maxcl = wave_max(maxcl);
    unsupported = wave_sum(unsupported);
    bad_offsets = wave_sum(bad_offsets);
    corrupt = wave_sum(corrupt);
    if ((threadIdx.x & 63) == 0) {
        if (conf) atomicAdd(&out[0], conf);
        if (maxn) atomicMax(&out[1], maxn);
        if (maxc) atomicMax(&out[2], maxc);
        if (maxcl) atomicMax(&out[3], maxcl);
        if (unsupported) atomicAdd(&out[4], unsupported);
        if (bad_offsets) atomicAdd(&out[5], bad_offsets);
        if (corrupt) atomicAdd(&out[6], corrupt);
    }
}

} // namespace pmx

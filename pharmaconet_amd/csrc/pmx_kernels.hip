// pmx_kernels.hip - the screening hot path on gfx950 (CDNA4, wave64).
//
// Execution model. A ligand is owned by a *conformer group*: G = 2^ceil(log2(max conformers)) adjacent
// lanes of a wavefront, lane c of the group holding conformer c. A wave64 therefore carries 64 / G
// ligands (8 at the 8-conformer shape of BASELINE.json, 1 at 64 conformers). Everything that is the
// same for all conformers of a ligand (tree control, candidate sets, table indices) is computed
// redundantly by the group's lanes, so control flow is uniform inside a group and diverges only
// between groups. Conformers are coupled exactly where the reference couples them (the joint
// `any conformer valid` and `num_matches + max_num_matches < 5` tests of tree.py:83-84,98).
//
// Three kernels per chunk of ligands:
//   sizes_kernel   one thread per ligand: candidate sets -> number of tree levels, table size
//   tables_kernel  conformer groups: the self / pair score tables of match_utils.py -> scratch arena
//   tree_kernel    conformer groups, persistent with dynamic ligand fetch: the DFS of tree.py over
//                  those tables, per-conformer maximum over leaves, mean -> score
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
    return (uint32_t)(table_bytes<G>(L.T, L.ksumtot) / 16);
}

// ----------------------------------------------------------------------------------- sizes_kernel
template <int G>
__global__ void sizes_kernel(DevLibrary lib, const uint64_t *tclus, uint64_t first, uint32_t count, uint32_t *units,
                             int32_t *status, uint32_t *meta /* [0] = max levels */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Record r = parse_record(lib.data + lib.offsets[first + i]);
    if (!record_supported(r)) {
        units[i] = 0;
        status[i] = PMX_LIGAND_UNSUPPORTED;
        return;
    }
    Levels L = scan_levels(r, tclus, [](int, int, int, uint64_t, uint32_t) {});
    units[i] = table_units<G>(L);
    status[i] = PMX_LIGAND_OK;
    if (L.nl > 0) atomicMax(&meta[0], (uint32_t)L.nl);
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

// ---------------------------------------------------------------------------------- tables_kernel
struct GroupLevels { // per conformer group, in LDS
    uint64_t cand[PMX_MAX_LEVELS];
    uint8_t start[PMX_MAX_LEVELS];
    uint8_t end[PMX_MAX_LEVELS];
    uint8_t k[PMX_MAX_LEVELS];
    uint8_t pad[4];
};
static_assert(sizeof(GroupLevels) % 16 == 0, "GroupLevels alignment");

struct Pos {
    float x, y, z;
};

__device__ inline Pos load_pos(const float *xyz, int C, int u, int c) {
    const float *p = xyz + (size_t)(u * 3) * C + c;
    return Pos{p[0], p[C], p[2 * C]};
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
__device__ inline void node_pair(const float4 *tab, int Nm, uint64_t A, uint64_t B, float d, float &acc, int &npass) {
    for (uint64_t am = A; am; am &= am - 1) {
        const float4 *row = tab + (__ffsll((unsigned long long)am) - 1) * Nm;
        for (uint64_t bn = B; bn; bn &= bn - 1) {
            const float4 e = row[__ffsll((unsigned long long)bn) - 1];
            const float t = fabsf(d - e.x);
            const float q = t * e.y;
            acc = __builtin_fmaf(e.w, exp2f(-(q * q)), acc);
            npass += (t <= e.z) ? 1 : 0;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void tables_kernel(DevModel M, DevLibrary lib, Weights W, uint64_t first, uint32_t count,
                                                     const int32_t *status, const uint64_t *taboff, uint8_t *arena) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int GPW = 64 / G; // groups per wave
    const int Nm = M.Nm;
    float4 *tab = reinterpret_cast<float4 *>(smem);
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(smem + (size_t)Nm * Nm * sizeof(float4));
    uint64_t *tnodes = cnodes + 64;
    GroupLevels *glev = reinterpret_cast<GroupLevels *>(tnodes + 128);

    // stage the model: edge table with the call's weights folded in (weights / stds, match_utils.py:65)
    for (int i = threadIdx.x; i < Nm * Nm; i += blockDim.x) {
        float4 e = M.edge[i];
        const float wm = W.w[M.node_type[i / Nm]], wn = W.w[M.node_type[i % Nm]];
        e.w = (wm * wn) / e.w;
        tab[i] = e;
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) cnodes[i] = M.cnodes[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tnodes[i] = M.tnodes[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g = lane / G, c = lane % G;
    const int gib = (threadIdx.x >> 6) * GPW + g;
    const uint64_t gid = (uint64_t)blockIdx.x * (blockDim.x / 64) * GPW + gib;
    if (gid >= count) return;
    if (status[gid] != PMX_LIGAND_OK) return;
    const uint64_t off = taboff[gid];
    if (taboff[gid + 1] == off) return; // no levels: score 0 (graph_match.py:95-99)

    const Record r = parse_record(lib.data + lib.offsets[first + gid]);
    const int C = r.C;
    const int cc = c < C ? c : C - 1; // idle lanes of the group shadow the last conformer
    const bool lane_live = c < C;
    GroupLevels &GL = glev[gib];
    const Levels L = scan_levels(r, M.tclus, [&](int lev, int start, int end, uint64_t cand, uint32_t k) {
        GL.cand[lev] = cand;
        GL.start[lev] = (uint8_t)start;
        GL.end[lev] = (uint8_t)end;
        GL.k[lev] = (uint8_t)k;
    });
    const int nl = L.nl;

    uint8_t *blk = arena + off;
    TabHeader *H = reinterpret_cast<TabHeader *>(blk);
    vmask_t<G> *Vt = reinterpret_cast<vmask_t<G> *>(blk + sizeof(TabHeader));
    float *St = reinterpret_cast<float *>(blk + sizeof(TabHeader) + round16(uint64_t(L.T) * sizeof(vmask_t<G>)));
    float *Pt = St + round16(uint64_t(L.ksumtot) * G * 4) / 4;

    // header (every lane of the group writes the same bytes)
    H->nl = (uint32_t)nl;
    H->T = L.T;
    H->ksumtot = L.ksumtot;
    H->pad = 0;
    {
        uint32_t ks = 0, rb = 0;
        for (int i = 0; i < nl; ++i) {
            const uint32_t k = GL.k[i];
            H->k[i] = (uint8_t)k;
            H->ksum[i] = (uint16_t)ks;
            H->rowbase[i] = rb;
            ks += k;
            rb += k * (L.ksumtot - ks); // k_i * sum_{j > i} k_j
        }
        H->ksum[nl] = (uint16_t)ks;
    }

    const float *xyz = r.xyz;
    const uint8_t *tm = r.typemask;
    uint32_t self_idx = 0, pair_idx = 0;
    for (int i = 0; i < nl; ++i) {
        const int si = GL.start[i], ei = GL.end[i];
        const uint64_t candi = GL.cand[i];
        Pos ctr_i;
        float size_i;
        cluster_center_size(xyz, C, si, ei, cc, ctr_i, size_i);

        // self table S[i][a] (match_utils.py:77-122): node pairs u < v inside the cluster, no pass logic
        for (uint64_t ca = candi; ca; ca &= ca - 1) {
            const uint64_t nodes_a = cnodes[__ffsll((unsigned long long)ca) - 1];
            float score = 0.f;
            for (int u = si; u < ei; ++u) {
                const uint64_t A = nodes_a & tnodes[tm[u]];
                if (!A) continue;
                const Pos pu = load_pos(xyz, C, u, cc);
                for (int v = u + 1; v < ei; ++v) {
                    const uint64_t B = nodes_a & tnodes[tm[v]];
                    if (!B) continue;
                    const Pos pv = load_pos(xyz, C, v, cc);
                    const float d = norm3(pu.x - pv.x, pu.y - pv.y, pu.z - pv.z);
                    float acc = 0.f;
                    int np = 0;
                    node_pair(tab, Nm, A, B, d, acc, np);
                    score = score + acc / (float)(__popcll(A) * __popcll(B));
                }
            }
            St[(size_t)self_idx * G + c] = score;
            ++self_idx;
        }

        for (int j = i + 1; j < nl; ++j) {
            const int sj = GL.start[j], ej = GL.end[j];
            const uint64_t candj = GL.cand[j];
            Pos ctr_j;
            float size_j;
            cluster_center_size(xyz, C, sj, ej, cc, ctr_j, size_j);
            const float ldist = norm3(ctr_i.x - ctr_j.x, ctr_i.y - ctr_j.y, ctr_i.z - ctr_j.z); // graph_match.py:240
            const float lsize = size_i + size_j;                                                 // :241
            for (uint64_t ca = candi; ca; ca &= ca - 1) {
                const int a = __ffsll((unsigned long long)ca) - 1;
                const uint64_t nodes_a = cnodes[a];
                for (uint64_t cb = candj; cb; cb &= cb - 1) {
                    const int b = __ffsll((unsigned long long)cb) - 1;
                    const uint64_t nodes_b = cnodes[b];
                    // cluster-distance prefilter, graph_match.py:263-268: skip when no conformer can match
                    const float2 mp = M.cpair[a * M.K + b];
                    const bool near = lane_live && !((fabsf(ldist - mp.x) - lsize) > mp.y);
                    const unsigned long long near_bal = __ballot(near);
                    const unsigned long long grp = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (g * G));
                    float value = -1.f;
                    bool valid = false;
                    if (near_bal & grp) {
                        // match_utils.py:9-74
                        float score = 0.f;
                        int fails = 0, L1 = 0, L2 = 0;
                        for (int v = sj; v < ej; ++v) L2 += (nodes_b & tnodes[tm[v]]) ? 1 : 0;
                        for (int u = si; u < ei; ++u) {
                            const uint64_t A = nodes_a & tnodes[tm[u]];
                            if (!A) continue;
                            ++L1;
                            const Pos pu = load_pos(xyz, C, u, cc);
                            for (int v = sj; v < ej; ++v) {
                                const uint64_t B = nodes_b & tnodes[tm[v]];
                                if (!B) continue;
                                const Pos pv = load_pos(xyz, C, v, cc);
                                const float d = norm3(pu.x - pv.x, pu.y - pv.y, pu.z - pv.z);
                                float acc = 0.f;
                                int np = 0;
                                node_pair(tab, Nm, A, B, d, acc, np);
                                const int mn = __popcll(A) * __popcll(B);
                                score = score + acc / (float)mn;
                                fails += (2 * np < mn) ? 1 : 0; // num_pass < num_match * 0.5   (:61)
                            }
                        }
                        valid = lane_live && (2 * fails <= L1 * L2) && (score > 0.f); // :71-74, tree.py:81
                        value = (2 * fails <= L1 * L2) ? score : -1.f;
                    }
                    const unsigned long long vbal = __ballot(valid);
                    Pt[(size_t)pair_idx * G + c] = value;
                    Vt[pair_idx] = (vmask_t<G>)((G == 64) ? vbal : ((vbal >> (g * G)) & ((1ull << G) - 1ull)));
                    ++pair_idx;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ tree_kernel
// Iterative form of ClusterMatchTree.dfs_run (tree.py:55-104). A frame f describes the tree node at
// level f - 1 (frame 0 = root): `cur` = next candidate of level f to try, `mx` = max_num_matches so
// far, flags = {matched, any candidate child existed, skip child handled}. The matched ancestors of
// the current path are kept as a list (match q: table row offset, k_j, chosen candidate a_j, level j),
// and the conformer mask / float64 totals are indexed by the number of matches (a skip child shares
// its parent's). A candidate (f, b) is evaluated against the matched ancestors when it is reached:
//   mask  = mask(parent) & AND_q V[entry(q, f, b)]                     (tree.py:78-84)
//   total = total(parent) + S[f][b] + sum_q P[entry(q, f, b)]          (tree.py:38-41)
//
// Work splitting. Trees are heavy-tailed (median ~1e3 nodes, tail > 1e7), and a tree walked by one
// group is a serial chain. The only coupling between sibling subtrees is the skip rule
// `num_matches + max_num_matches < 5` (tree.py:98). Since `num_matches(A) + max_num_matches(A)` is the
// largest match count of any leaf below A's candidate children, the rule only asks whether a node
// with >= 5 matches exists there. Hence a subtree rooted at a node Y with num_matches(Y) >= 5 can be
// cut out: every ancestor's decision is already settled by Y's existence (returning 1 for Y gives
// each ancestor A at least 5 - num_matches(A)), decisions inside the subtree depend on candidate
// existence only, and leaves only feed a per-conformer maximum. A walker that exceeds its step budget
// therefore stops descending into such nodes and appends them to a task queue; tasks are walked by
// the same code (TASKS = true) in rounds, splitting again when over budget, and per-conformer maxima
// of split ligands are combined with atomicMax in `bestbuf` (non-negative doubles order as uint64).
struct TaskHeader { // 64 bytes, followed by double tot[G]
    uint32_t lig;   // ligand index inside the chunk
    uint8_t f0;     // frame of the subtree's root
    uint8_t nm;     // matches on the path, root included
    uint8_t pad[2];
    uint64_t mask;  // conformer mask of the root
    uint8_t path[2 * PMX_MAX_LEVELS]; // (level, candidate) of every match on the path
    uint8_t pad2[8];
};
static_assert(sizeof(TaskHeader) == 64, "TaskHeader layout");

template <int G>
__host__ __device__ constexpr uint32_t task_bytes() {
    return sizeof(TaskHeader) + G * 8;
}

struct TreeParams {
    const uint8_t *arena;
    const uint64_t *taboff;
    const int32_t *status;
    DevLibrary lib;
    uint64_t first;      // library index of the chunk's first ligand
    uint32_t count;      // work items: ligands (TASKS = false) or tasks [task_lo, task_lo + count)
    uint32_t task_lo;
    uint32_t *counter;   // dynamic fetch counter
    uint32_t *qtail;     // task queue tail; qtail[1] = overflow flag
    uint8_t *queue;
    uint32_t qcap;
    unsigned long long *bestbuf; // [chunk][G]
    uint8_t *deferred;           // [chunk] ligand was split: score comes from bestbuf
    int depth_cap;
    int K;               // model clusters (bound on candidates per level)
    uint32_t budget;     // steps after which a walker donates its unexplored subtrees to the queue
    unsigned long long *nsteps; // total DFS steps (diagnostics)
    float *scores;
};

// LDS bytes of one conformer group's tree state for stacks that hold `depth` levels of a model with K clusters.
template <int G>
__host__ __device__ inline uint32_t tree_group_bytes(int depth, int K) {
    const uint32_t tot_bytes = (uint32_t)(depth + 1) * G * 8;                                      // float64 totals [depth + 1][G]
    const uint32_t todo_bytes = (uint32_t)(depth + 1) * 8;                                         // unexplored existing candidates per frame
    const uint32_t cm_bytes = (uint32_t)round16((uint64_t)(depth + 1) * K * sizeof(vmask_t<G>));  // conformer masks of a frame's candidates
    const uint32_t msk_bytes = (uint32_t)round16((uint64_t)(depth + 1) * sizeof(vmask_t<G>));     // conformer masks by match count
    const uint32_t frm_bytes = (uint32_t)round16((uint64_t)(depth + 1) * 4);                      // frames {-, mx, flags, nm}
    const uint32_t mat_bytes = (uint32_t)round16((uint64_t)depth * 8);                            // matched ancestors
    return tot_bytes + todo_bytes + cm_bytes + msk_bytes + frm_bytes + mat_bytes + 32 + 48 + 80;   // + k[32], ksum[24], rowbase[20]
}

// Pair-table index of (matched ancestor q, candidate 0 of level f): + b gives candidate b.
__device__ inline int entry_base(const int2 mq, int ksf, int kf) {
    return mq.x + (mq.y & 255) * ksf + ((mq.y >> 8) & 255) * kf;
}

// sum_q P[entry(q, f, b)][c] in ancestor order (tree.py:78-82), loads issued four at a time
template <int G>
__device__ inline double pair_sum(const float *Pt, const int2 *mat, int nm, int ksf, int kf, int b, int c) {
    double pair = 0.0;
    int q = 0;
    for (; q + 4 <= nm; q += 4) {
        const int i0 = entry_base(mat[q], ksf, kf) + b, i1 = entry_base(mat[q + 1], ksf, kf) + b;
        const int i2 = entry_base(mat[q + 2], ksf, kf) + b, i3 = entry_base(mat[q + 3], ksf, kf) + b;
        const float p0 = Pt[(size_t)i0 * G + c], p1 = Pt[(size_t)i1 * G + c];
        const float p2 = Pt[(size_t)i2 * G + c], p3 = Pt[(size_t)i3 * G + c];
        pair += (double)p0;
        pair += (double)p1;
        pair += (double)p2;
        pair += (double)p3;
    }
    for (; q < nm; ++q) pair += (double)Pt[(size_t)(entry_base(mat[q], ksf, kf) + b) * G + c];
    return pair;
}

template <int G, bool TASKS>
__global__ __launch_bounds__(64) void tree_kernel(TreeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using vm_t = vmask_t<G>;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, c = lane % G;
    const int D = p.depth_cap; // levels the stacks can hold
    const int K = p.K;

    const uint32_t tot_bytes = (uint32_t)(D + 1) * G * 8;
    const uint32_t todo_bytes = (uint32_t)(D + 1) * 8;
    const uint32_t cm_bytes = (uint32_t)round16((uint64_t)(D + 1) * K * sizeof(vm_t));
    const uint32_t msk_bytes = (uint32_t)round16((uint64_t)(D + 1) * sizeof(vm_t));
    const uint32_t frm_bytes = (uint32_t)round16((uint64_t)(D + 1) * 4);
    const uint32_t mat_bytes = (uint32_t)round16((uint64_t)D * 8);
    unsigned char *base = smem + (size_t)g * tree_group_bytes<G>(D, K);
    double *tot = reinterpret_cast<double *>(base);                         // [D + 1][G]
    uint64_t *todo = reinterpret_cast<uint64_t *>(base + tot_bytes);        // [D + 1]
    vm_t *cm = reinterpret_cast<vm_t *>(base + tot_bytes + todo_bytes);     // [D + 1][K]
    vm_t *msk = reinterpret_cast<vm_t *>(base + tot_bytes + todo_bytes + cm_bytes); // [D + 1]
    uchar4 *frm = reinterpret_cast<uchar4 *>(base + tot_bytes + todo_bytes + cm_bytes + msk_bytes); // [D + 1] {-, mx, flags, nm}
    int2 *mat = reinterpret_cast<int2 *>(base + tot_bytes + todo_bytes + cm_bytes + msk_bytes + frm_bytes); // [D] {R, k_j | a_j << 8 | j << 16}
    uint8_t *hk = base + tot_bytes + todo_bytes + cm_bytes + msk_bytes + frm_bytes + mat_bytes; // k[32]
    uint16_t *hksum = reinterpret_cast<uint16_t *>(hk + 32);                // [24]
    uint32_t *hrow = reinterpret_cast<uint32_t *>(hk + 32 + 48);            // [20]
    constexpr unsigned F_MATCHED = 1, F_ANY = 2, F_SKIP = 4, F_EXPANDED = 8;

    // One flat loop: each group is either between work items (f < f0: finish the previous one, fetch
    // the next) or inside a tree (one DFS step per iteration), so groups of a wave advance independently.
    bool running = true, have = false, exported = false;
    int f = -1, f0 = 0, nl = 0, C = 1;
    uint32_t li = 0, steps = 0, steps_total = 0;
    double best = 0.0;
    const vm_t *Vt = nullptr;
    const float *St = nullptr, *Pt = nullptr;

    // hand the subtree of candidate b of frame fr (conformer mask m) to the task queue
    auto donate = [&](int fr, int nmr, int b, vm_t m) -> bool {
        uint32_t slot = 0;
        if (c == 0) slot = atomicAdd(p.qtail, 1u);
        slot = __shfl(slot, g * G);
        if (slot >= p.qcap) {
            if (c == 0) p.qtail[1] = 1;
            return false;
        }
        const int kr = hk[fr], ksr = hksum[fr];
        const double pair = pair_sum<G>(Pt, mat, nmr, ksr, kr, b, c);
        TaskHeader *th = reinterpret_cast<TaskHeader *>(p.queue + (size_t)slot * task_bytes<G>());
        th->lig = li;
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
        reinterpret_cast<double *>(th + 1)[c] = tot[nmr * G + c] + (double)St[(size_t)(ksr + b) * G + c] + pair;
        return true;
    };

    while (running) {
        if (f < f0) {
            if (have) {
                if (c == 0) atomicAdd(p.nsteps, (unsigned long long)steps_total + steps);
                if (TASKS || exported) { // split ligand: combine per-conformer maxima across walkers
                    if (best > 0.0) atomicMax(&p.bestbuf[(size_t)li * G + c], (unsigned long long)__double_as_longlong(best));
                    if (!TASKS && c == 0) p.deferred[li] = 1;
                } else { // mean over conformers (graph_match.py:109); idle lanes hold 0
                    double s = best;
#pragma unroll
                    for (int d = 1; d < G; d <<= 1) s += __shfl_xor(s, d);
                    if (c == 0) p.scores[li] = (float)(s / (double)C);
                }
                have = false;
            }
            uint32_t nx = 0;
            if (c == 0) nx = atomicAdd(p.counter, 1u); // dynamic fetch of the next work item
            nx = __shfl(nx, g * G);
            f = -1;
            f0 = 0;
            if (nx >= p.count) {
                running = false;
                continue;
            }
            const TaskHeader *task = nullptr;
            if (TASKS) {
                task = reinterpret_cast<const TaskHeader *>(p.queue + (size_t)(p.task_lo + nx) * task_bytes<G>());
                li = task->lig;
            } else {
                li = nx;
                if (p.status[li] != PMX_LIGAND_OK) {
                    if (c == 0) p.scores[li] = __builtin_nanf("");
                    continue;
                }
            }
            const uint64_t off = p.taboff[li];
            if (!TASKS && p.taboff[li + 1] == off) { // no ligand cluster has a candidate (graph_match.py:95-99)
                if (c == 0) p.scores[li] = 0.f;
                continue;
            }
            const uint8_t *blk = p.arena + off;
            const TabHeader *H = reinterpret_cast<const TabHeader *>(blk);
            nl = (int)H->nl;
            const uint32_t T = H->T, ksumtot = H->ksumtot;
            Vt = reinterpret_cast<const vm_t *>(blk + sizeof(TabHeader));
            St = reinterpret_cast<const float *>(blk + sizeof(TabHeader) + round16(uint64_t(T) * sizeof(vm_t)));
            Pt = St + round16(uint64_t(ksumtot) * G * 4) / 4;
            for (int i = c; i <= nl; i += G) { // the group's lanes share the header copy
                hksum[i] = H->ksum[i];
                if (i < nl) {
                    hk[i] = H->k[i];
                    hrow[i] = H->rowbase[i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            best = 0.0; // graph_match.py:104
            have = true;
            exported = false;
            steps = 0;
            steps_total = 0;
            if (TASKS) {
                const int nm0 = task->nm;
                f0 = f = task->f0;
                for (int q = c; q < nm0; q += G) {
                    const int j = task->path[2 * q], a = task->path[2 * q + 1];
                    const int kj = H->k[j];
                    mat[q] = make_int2((int)H->rowbase[j] - kj * (int)H->ksum[j + 1], kj | (a << 8) | (j << 16));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                tot[nm0 * G + c] = reinterpret_cast<const double *>(task + 1)[c];
                msk[nm0] = (vm_t)task->mask;
                frm[f] = make_uchar4(0, 0, F_MATCHED, (unsigned char)nm0);
            } else {
                C = parse_record(p.lib.data + p.lib.offsets[p.first + li]).C;
                f0 = f = 0; // root frame
                tot[c] = 0.0;
                msk[0] = (vm_t)((C >= 64) ? ~0ull : ((1ull << C) - 1ull));
                frm[0] = make_uchar4(0, 0, 0, 0);
            }
            continue;
        }
        ++steps;
        if (steps > p.budget) {
            // Over budget: donate every unexplored candidate subtree on the stack (shallow frames first -
            // they are the large ones) to the task queue, then carry on with what is left under a fresh
            // budget. Only frames with >= 4 matches may donate (their children have >= 5, see above).
            steps_total += steps;
            steps = 0;
            for (int fr = f0; fr <= f && fr < nl; ++fr) {
                uchar4 Fr = frm[fr];
                const int nmr = Fr.w;
                if (nmr < 4 || !(Fr.z & F_EXPANDED)) continue;
                uint64_t left = todo[fr];
                while (left) {
                    const int b = __ffsll((unsigned long long)left) - 1;
                    if (!donate(fr, nmr, b, cm[fr * K + b])) break;
                    left &= left - 1;
                    exported = true;
                    Fr.y = Fr.y > 1 ? Fr.y : 1; // the donated child returns at least 1
                }
                todo[fr] = left;
                frm[fr] = Fr;
            }
        }
        uchar4 F = frm[f];
        const int nm = F.w;
        const bool matched = F.z & F_MATCHED;
        if (f == nl) { // leaf (tree.py:103-104): per-conformer maximum (graph_match.py:105-108)
            const double t = tot[nm * G + c];
            if (((msk[nm] >> c) & 1) && t > best) best = t;
            --f;
            if (f >= f0) {
                uchar4 Pf = frm[f];
                const unsigned char ret = matched ? 1 : 0;
                Pf.y = Pf.y > ret ? Pf.y : ret;
                frm[f] = Pf;
            }
            continue;
        }
        const int kf = hk[f], ksf = hksum[f];
        if (!(F.z & F_EXPANDED)) {
            // Evaluate every candidate of level f against the matched ancestors at once: lane c takes
            // candidates c, c + G, ...; mask(b) = mask(parent) & AND_q V[entry(q, f, b)]  (tree.py:78-84).
            uint64_t E = 0;
            const vm_t pm = msk[nm];
            for (int b0 = 0; b0 < kf; b0 += G) {
                const int b = b0 + c;
                const bool on = b < kf;
                const int bb = on ? b : 0;
                vm_t m = on ? pm : (vm_t)0;
                int q = 0;
                for (; q + 4 <= nm; q += 4) {
                    const vm_t v0 = Vt[entry_base(mat[q], ksf, kf) + bb], v1 = Vt[entry_base(mat[q + 1], ksf, kf) + bb];
                    const vm_t v2 = Vt[entry_base(mat[q + 2], ksf, kf) + bb], v3 = Vt[entry_base(mat[q + 3], ksf, kf) + bb];
                    m &= (vm_t)(v0 & v1 & v2 & v3);
                }
                for (; q < nm; ++q) m &= Vt[entry_base(mat[q], ksf, kf) + bb];
                if (on) cm[f * K + b] = m;
                const unsigned long long bal = __ballot(on && m != 0);
                E |= ((G == 64) ? bal : ((bal >> (g * G)) & ((1ull << G) - 1ull))) << b0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            todo[f] = E;
            F.z |= F_EXPANDED | (E ? F_ANY : 0);
            frm[f] = F;
            continue;
        }
        const uint64_t left = todo[f];
        if (left) { // descend into the next existing candidate child (tree.py:94-97)
            const int b = __ffsll((unsigned long long)left) - 1;
            todo[f] = left & (left - 1);
            // parent + self + accumulated pair (tree.py:38-41)
            const double t = tot[nm * G + c] + (double)St[(size_t)(ksf + b) * G + c] + pair_sum<G>(Pt, mat, nm, ksf, kf, b, c);
            tot[(nm + 1) * G + c] = t;
            msk[nm + 1] = cm[f * K + b];
            // entry(this match, level f', b') = rowbase[f] + k_f * (ksum[f'] - ksum[f + 1]) + b * k_f' + b'
            mat[nm] = make_int2((int)hrow[f] - kf * (int)hksum[f + 1], kf | (b << 8) | (f << 16));
            ++f;
            frm[f] = make_uchar4(0, 0, F_MATCHED, (unsigned char)(nm + 1));
            continue;
        }
        if (!(F.z & F_SKIP)) { // skip child (tree.py:98-101)
            F.z |= F_SKIP;
            frm[f] = F;
            if (!(F.z & F_ANY) || (nm + F.y) < 5) {
                ++f;
                frm[f] = make_uchar4(0, 0, 0, (unsigned char)nm);
                continue;
            }
        }
        // all children done: return max_num_matches + matched (tree.py:102)
        const unsigned char ret = (unsigned char)(F.y + (matched ? 1 : 0));
        --f;
        if (f >= f0) {
            uchar4 Pf = frm[f];
            Pf.y = Pf.y > ret ? Pf.y : ret;
            frm[f] = Pf;
        }
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

// -------------------------------------------------------------------------------- library stats
__global__ void library_stats_kernel(DevLibrary lib, unsigned long long *out /* [0] conformers [1] maxn [2] maxC [3] maxcl [4] unsupported */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lib.n) return;
    Record r = parse_record(lib.data + lib.offsets[i]);
    atomicAdd(&out[0], (unsigned long long)r.C);
    atomicMax(&out[1], (unsigned long long)r.n);
    atomicMax(&out[2], (unsigned long long)r.C);
    atomicMax(&out[3], (unsigned long long)r.ncl);
    if (!record_supported(r)) atomicAdd(&out[4], 1ull);
}

} // namespace pmx

// pmx_screen.hip - the screening hot path on gfx950 (CDNA4, wave64): one wavefront scores one ligand from its packed
// record to its score.
//
// Reference path: PharmacophoreModel._scoring -> GraphMatcher.run() (src/pmnet/scoring/graph_match.py:63-279,
// scoring/match_utils.py:9-122, scoring/tree.py:15-104); citations below are relative to /root/reference/src/pmnet.
//
// Lanes. A wavefront is 64 / G *slots* of G lanes; lane c of a slot is conformer c (G = 2^ceil(log2(max conformers)):
// 8 at BASELINE.json's 8-conformer shape). What a slot stands for changes with the phase:
//   * table phase: a slot owns one table entry - (ligand cluster i, model cluster a) for the self table S, ((i, a), (j, b))
//     for the pair table P - and walks that entry's ligand node pairs (u, v) in a wave-uniform loop. The sum over the
//     compatible model node pairs of one (u, v),
//         F(d) = 1 / (|A||B|) * sum_{m in A, n in B} w_m w_n / std_mn * exp(-((d - mean_mn) / std_mn)^2 / 2),
//     depends on the ligand only through the scalar d = |x_u - x_v|: it is a property of the model (and the call's type
//     weights). fn_build_kernel tabulates every such F once per (model, weights) as piecewise quintic Hermite cells (from
//     F, F', F'' evaluated in float64 at the knots; measured deviation from the exact sum < 3e-9 of the function's peak at
//     h <= std_min / 5), so that an item costs one 32-byte gather and five FMAs instead of |A||B| (27 on average, up to
//     169) Gaussian terms. The discrete part of match_utils.py - `num_pass < num_match * 0.5` (:61) - is NOT approximated:
//     the set of distances where at least half of the model node pairs lie within 2 sigma is a union of float intervals
//     computed exactly on the host when the model is created (pmx_api.hip, fn_windows) and stored with the cells; cells
//     where that set is not one interval are flagged and counted term by term on the device. Cells whose polynomial is not
//     accurate relative to the function's own (tail) value are flagged too, and evaluated term by term in self entries
//     (FnCell, exact_value). The node distances of the cluster pair in work are staged once in LDS (build_tables).
//   * tree phase: ONE depth-first walker per wavefront with wave-uniform control (scalar registers, scalar branches);
//     a slot evaluates one candidate child of the current frame, so that a frame's children - their conformer
//     masks, float64 totals and bound tests - are one pass of independent loads. The walker's stack lives in lane-indexed
//     registers (v_readlane / v_writelane), the float64 path totals in 1.3 KB of LDS. Children are tested against a
//     per-candidate bound (build_bounds) and, before the walker enters one, against the bound its actual path gives
//     (path_bound): 60 frames per ligand on the bench library where the reference's search has 11 300 nodes.
//
// Memory. The score tables of a ligand (S, P, search bounds R / W / OB; 15 KB on average) are written to a per-wavefront slice of
// global memory and read back by the same wavefront: they stay in the CU's L1 / the XCD's L2 and are overwritten by the
// wave's next ligand - there is no per-chunk table arena, no size pass, no host read. Ligands whose tables exceed the
// slice, and trees that run over their budget, move to a bump-allocated arena: over-budget walkers append the open
// subtrees with >= 5 matches to a task queue (exactness argument: see walk()), which launches of task_kernel drain in rounds.
// Everything is ordered on the caller's stream.
#include <type_traits>

// The kernels are compiled twice. libpmx's own (namespace pmx, pmx_api.hip) know two PMX_TREE_FLAGS switches - no budget (2) and tables
// alone (16384) - and read every other bit as zero, so the walker and the table loops carry none of the validation switches (1.5 % of the
// pass; as a template parameter in one translation unit the two sets of kernels cost each other registers). pmx_screen_debug.hip compiles
// this file again as namespace pmx_dbg with every switch live; a call with any other bit set launches those.
#ifndef PMX_NS
#define PMX_NS pmx
#endif
#define PMX_SCORES_F64 (1u << 30) // ScreenParams::flags: `scores` is a double array (pmx_score_f64); set by the host, not by PMX_TREE_FLAGS
#define PMX_PRODUCT_FLAGS (2u | 16384u)
#ifdef PMX_DEBUG_KERNELS
#define PMX_WFLAGS(p) ((p).flags)
#else
#define PMX_WFLAGS(p) ((p).flags & PMX_PRODUCT_FLAGS)
#endif

#include "pmx_device.h"

#pragma clang fp contract(off)

namespace PMX_NS {
using namespace pmx; // (pmx_device.h)

// One cell of a tabulated pair function: value(t) = c0 + t (c1 + t (c2 + t (c3 + t (c4 + t c5)))), t in [0, 1) the position
// inside the cell; the item passes the 2-sigma majority test of match_utils.py:56-61 iff lo <= d <= hi (lo = NaN: the pass
// set is not an interval inside this cell - count the terms).
// The lowest mantissa bit of c[5] is a flag: the polynomial is not accurate *relative to the function's own value* in this
// cell (the far tails of the Gaussians, and the last cell, which stands for every distance beyond the grid). Entries of the
// self table have no majority test (match_utils.py:77-122), so a self entry can consist of tail values only and be a
// ligand's whole score: the self loop evaluates the terms of a flagged cell one by one, in the reference's float32
// operations (exact_value). A pair entry that counts at all holds items that passed the 2-sigma majority test - values near
// the functions' peaks, next to which a tail value's error is below float32 rounding - and ignores the flag.
struct FnCell {
    float c[6];
    float lo, hi;
};
static_assert(sizeof(FnCell) == 32, "FnCell layout");

struct FnTable {
    const FnCell *cells; // two planes of float4[functions][ncell]: {c0, c1, c2, c3} | {c4, c5, lo, hi} (plane B = plane A + plane16 float4s):
                         // the eight conformers of a slot read neighbouring cells, and 16 bytes per cell and plane keep them in one cache line
    uint32_t plane16;
    uint32_t NS;         // node subsets (0 = empty)
    uint32_t ncell;
    float inv_h;
    uint32_t tri;        // the functions of a symmetric model are stored once per unordered subset pair
};

// Header of one ligand's tables, in a wave's slice or in the arena:
//   [RecHeader][best u64[G]][S float[ksumtot][G]][P float[T][G]][R double[nl + 1][G]][W double[ksumtot][G]][V mask[T]]
//   [OB float[nl][ksumtot][G]][LV u8[ksumtot]][DP u8[ksumtot]]                    (where per-candidate bounds exist; at 32 / 64
//   lanes OB is the one row BF float[ksumtot][G], see ob_rows())
// OB[f][x] for a candidate x = (l, b') of a level l > f: S[l][b'] + sum_{f < j < l} max(0, max_a P[(j, a), (l, b')]), rounded up - what
// (l, b') can add to a leaf total apart from its pair entries with the matches on the path down to level f (path_bound()).
// LV[x] = the level of candidate x.
// DP[x] = the longest chain of candidates of ascending levels that starts with x and in which every candidate has an entry
// with some conformer > 0 against the one before it (V != 0): no path through x holds more matches from x on (probe()).
// V[e] = the conformers c with P[e][c] > 0 (one bit per conformer, max(G, 8) / 8 bytes per entry): what decides which
// children of a tree node exist (tree.py:78-84), read with the lanes spread over candidates.
// Pair entry of (i, a) with a candidate x = ksum[j] + b of a deeper level j: rowbase[i] + a * nd_i + (x - ksum[i + 1]), nd_i = ksumtot -
// ksum[i + 1] the candidates below level i: the entries of (i, a) with ALL deeper candidates are one contiguous run, so the row of a
// match on the path against any deeper candidate x is (a number fixed per match) + x - what the walker's passes and path_bound() read.
struct RecHeader {
    uint32_t lig; // ligand index relative to the call's `first`
    uint32_t nl, T, ksumtot;
    uint32_t bytes; // of the whole record
    uint32_t C;
    uint32_t pad[2];
    uint8_t k[PMX_MAX_LEVELS];
    uint8_t pad2[12];
    uint16_t ksum[PMX_MAX_LEVELS + 4];
    uint32_t rowbase[PMX_MAX_LEVELS];
    uint8_t pad3[64];
};
static_assert(sizeof(RecHeader) == 256, "RecHeader layout");

template <int G>
__host__ __device__ constexpr uint32_t rec_s_off() {
    return sizeof(RecHeader) + G * 8;
}
template <int G>
__host__ __device__ inline uint32_t rec_p_off(uint32_t ksumtot) {
    return rec_s_off<G>() + (uint32_t)round16((uint64_t)ksumtot * G * 4);
}
template <int G>
__host__ __device__ inline uint32_t rec_r_off(uint32_t ksumtot, uint32_t T) {
    return rec_p_off<G>(ksumtot) + (uint32_t)round16((uint64_t)T * G * 4);
}
template <int G>
__host__ __device__ constexpr uint32_t vmask_bytes() {
    return G < 8 ? 1u : (uint32_t)G / 8u;
}
template <int G>
__host__ __device__ inline uint32_t rec_w_off(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return rec_r_off<G>(ksumtot, T) + (nl + 1u) * G * 8u;
}
// (per-candidate bounds exist where a pass holds >= 4 candidates: 1 .. 16 conformer lanes; see build_bounds)
template <int G>
__host__ __device__ constexpr bool cand_bounds() {
    return 64 / G >= 4;
}
// Upper bounds need no precision (round 6): W is float32 rounded up instead of float64, the OB rows bfloat16 rounded up instead of float32 -
// 12 KB of a bench ligand's 25 KB of tables were these two. (PMX_SLIM_BOUNDS=0 builds the round-5 layout for A/B runs.)
#ifndef PMX_SLIM_BOUNDS
#define PMX_SLIM_BOUNDS 0 // ([MI355X] W as float32: tables 43.5 -> 43.0 ms, the walk 2 ms slower - 100.8 against 99.7 ms per pass, twice; not kept)
#endif
#ifndef PMX_SLIM_OB
#define PMX_SLIM_OB 1 // ([MI355X] OB as bfloat16 rounded up: 99.7 -> 98.3 ms per pass, 200.6 -> 184.9 KB per ligand across the L2 <-> fabric boundary, frames 60.5 -> 60.7)
#endif
#ifndef PMX_SLIM_PA
#define PMX_SLIM_PA 0 // path_bound()'s pair sums as bfloat16 rounded up
#endif
constexpr bool kSlimPA = PMX_SLIM_PA != 0;
typedef std::conditional<kSlimPA, uint16_t, float>::type PaElt;
constexpr bool kSlimBounds = PMX_SLIM_BOUNDS != 0; // W as float32
constexpr bool kSlimOB = PMX_SLIM_OB != 0;         // OB as bfloat16
template <int G>
__host__ __device__ constexpr uint32_t ob_elt_bytes() {
    return (kSlimOB && cand_bounds<G>()) ? 2u : 4u;
}
// bytes of the W region: a float (double before round 6) per candidate and conformer - and room for the cluster centres build_tables() parks there
// (float2[nl][G]) until build_bounds() writes W
template <int G>
__host__ __device__ inline uint32_t rec_w_bytes(uint32_t ksumtot, uint32_t nl) {
    if (!cand_bounds<G>()) return 0u;
    if (!kSlimBounds) return ksumtot * G * 8u;
    const uint32_t w = ksumtot * G * 4u, park = nl * G * 8u;
    return (uint32_t)round16((uint64_t)(w > park ? w : park));
}
template <int G>
__host__ __device__ inline uint32_t rec_v_off(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return rec_w_off<G>(ksumtot, T, nl) + rec_w_bytes<G>(ksumtot, nl);
}
template <int G>
__host__ __device__ inline uint32_t rec_ob_off(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return rec_v_off<G>(ksumtot, T, nl) + (uint32_t)round16((uint64_t)T * vmask_bytes<G>());
}
// rows of the OB table: one per level where per-candidate bounds exist; ONE otherwise (32 / 64 conformer lanes) - BF[x] = base(x)
// rounded up, what candidate x can add to a leaf total at most whatever is matched above it (path_bound_wide())
template <int G>
__host__ __device__ constexpr uint32_t ob_rows(uint32_t nl) {
    return cand_bounds<G>() ? nl : 1u;
}
template <int G>
__host__ __device__ inline uint32_t rec_ci_off(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return rec_ob_off<G>(ksumtot, T, nl) + (uint32_t)round16((uint64_t)ob_rows<G>(nl) * ksumtot * G * ob_elt_bytes<G>());
}
template <int G>
__host__ __device__ inline uint64_t rec_bytes(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return (uint64_t)rec_s_off<G>() + round16((uint64_t)ksumtot * G * 4) + round16((uint64_t)T * G * 4) + (uint64_t)(nl + 1) * G * 8 +
           (uint64_t)rec_w_bytes<G>(ksumtot, nl) + round16((uint64_t)T * vmask_bytes<G>()) +
           round16((uint64_t)ob_rows<G>(nl) * ksumtot * G * ob_elt_bytes<G>()) + 2 * round16((uint64_t)ksumtot);
}
// (DP u8[ksumtot] follows LV: rec_ci_off + round16(ksumtot))
template <int G>
__host__ __device__ inline uint32_t rec_dp_off(uint32_t ksumtot, uint32_t T, uint32_t nl) {
    return rec_ci_off<G>(ksumtot, T, nl) + (uint32_t)round16((uint64_t)ksumtot);
}

// A subtree handed to the task queue: its root has >= 5 matches (see walk()).
struct TaskRec { // 64 bytes, followed by double tot[G]
    uint32_t rec16; // arena offset of the ligand's record, in 16-byte units
    uint8_t f0;     // frame of the subtree's root
    uint8_t nm;     // matches on the path, root included
    uint16_t pad;
    uint64_t mask;                    // conformer mask of the root
    uint8_t path[2 * PMX_MAX_LEVELS]; // (level, candidate) of every match on the path
    uint32_t pad2[2];
};
static_assert(sizeof(TaskRec) == 64, "TaskRec layout");
template <int G>
__host__ __device__ constexpr uint32_t task_rec_bytes() {
    return sizeof(TaskRec) + G * 8;
}

constexpr int kShards = 64; // task queue shards (= the wave size: a task wave finds its record with one scan over the shards)
constexpr int kStatWords = 26;
constexpr int kScreenStatShards = 64;

// Device-side control block of one call (zeroed by ctl_clear_kernel at the start of every super-chunk).
// The task queue is kShards independent queues (shard s owns records [s * qcap, (s + 1) * qcap)): a device-scope atomic
// on one address is a serial resource on this multi-XCD part, and exports come by the million.
struct Ctl {
    uint32_t cursor[4];   // ligand cursors of the launches of a super-chunk: [0] slice pass, [1] large-slice pass, [2] arena pass
    uint32_t ovf_count;   // ligands whose tables do not fit a slice
    uint32_t carry_count; // ligands whose tables do not fit a large slice either
    uint32_t heavy_count; // records in the arena that finalize has to score
    uint32_t pad0;
    uint32_t retry_count[2]; // ligands of the arena pass that found the arena full (retried with the arena to themselves)
    uint32_t pad00[2];
    unsigned long long arena_top; // bump allocator (bytes)
    uint32_t qflag;               // a queue shard was full (the walker then keeps the subtree: exact, only slower)
    uint32_t pad1;
    uint32_t q_res[kShards];      // records reserved
    uint32_t round_lo[kShards], round_hi[kShards]; // the records of the current round (round_kernel)
    uint32_t round_total, task_cursor;
    uint32_t pad[2];
    uint32_t xcd_cursor[8][16];   // task cursors of the round, one per group of 8 shards (one 64-byte line each)
    uint32_t round_inc[kShards];  // records of the round in shards 0 .. s (task number -> shard)
    unsigned long long stats[kScreenStatShards][kStatWords]; // sharded: [0] frames [1] passes [2] walks over budget [3] items [4] exact-count cells [5] longest walk [6] tasks [7] slice overflows [8..12] phase ticks [13] self items evaluated term by term
};

// With 32 or 64 conformer lanes the float64 path totals (21 rows of G) are 5 / 11 KB: kept in LDS they cap the CU at 8 wavefronts.
// There they live in global memory (one buffer per wavefront, L1 / L2 resident), and the children cache - a frame of those
// shapes never has all its candidates in one pass - has no LDS at all.
constexpr uint32_t kTotBufBytes = 16384;
template <int G>
__host__ __device__ constexpr bool totals_in_lds() {
    return G < 32;
}
struct ScreenParams {
    DevModel M;
    FnTable F;
    DevLibrary lib;
    const uint16_t *sidtab;    // [K * 128] node subset of (model cluster, ligand type mask); 0 = empty
    const uint32_t *sub_off;   // [NS + 1] the model nodes of node subset s: sub_nodes[sub_off[s] .. sub_off[s + 1]), ascending (0 = the empty subset)
    const uint8_t *sub_nodes;
    Weights W;                 // for the exact-term debug path
    uint64_t first;            // library index of the call's first ligand
    uint32_t lo, hi;           // ligands [lo, hi) of the call (relative to first) are this super-chunk
    Ctl *ctl;
    uint8_t *totbuf;           // [waves][kTotBufBytes]: the path totals of the 32 / 64-lane shapes (LDS at fewer lanes)
    uint8_t *pabuf;            // [waves][pa_bytes]: path_bound()'s pair sums of the matches on the path, float[matches][ksumtot][G]
    uint32_t pa_bytes;
    uint8_t *slices;           // [waves][slice_bytes]
    uint32_t slice_bytes;
    uint8_t *arena;
    unsigned long long arena_bytes;
    uint32_t *ovf_list, *carry_list, *heavy_list; // ligand indices / arena offsets (16-byte units)
    uint32_t list_cap;
    uint8_t *queue;
    uint32_t qcap;             // records per shard
    uint32_t budget;           // passes after which a walker starts handing subtrees to the queue
    uint32_t min_levels;       // only subtrees with at least this many levels below their root are queued
    uint32_t flags;            // 2: never queue, 4: no bound test, 8: exact Gaussian terms instead of the tabulated functions, 32768: no chain lengths (probe()), 65536: no dead-entry test (build_tables), 131072: no path-aware test at 32 / 64 lanes (path_bound_wide())
    uint32_t max_nodes;        // of the library (sizes the LDS node tables)
    uint32_t last_round;       // task_kernel: never queue (walk every subtree to its end)
    uint32_t bound_cost; // per-candidate bounds are built when their cost estimate stays below this (build_bounds)
    uint32_t dead_min_entries; // the dead-entry test (build_tables) runs for level pairs with at least this many entries
    float *scores;             // float[count]; double[count] when flags & PMX_SCORES_F64 (put_score())
    int32_t *status;
    int mode;                  // 0: slice pass over [lo, hi); 1: large-slice pass over ovf_list; 2: arena pass over carry_list; 3: arena pass over retry_in
    const uint32_t *retry_in;  // mode 3: the ligands an earlier arena pass had no room for (count: ctl->retry_count[retry_slot ^ 1])
    uint32_t *retry_out;       // modes 2, 3: where such ligands go (count: ctl->retry_count[retry_slot]); nullptr: they are reported as too large
    uint32_t retry_slot;
};


// ------------------------------------------------------------------------------------------------ helpers
// Instruction injection (analysis builds only: -DPMX_INJECT_VALU_ITEM=n, -DPMX_INJECT_VALU_WALK=n, -DPMX_INJECT_SALU_WALK=n): n extra instructions of one
// kind per table item batch / per trip of the walker's loop. The slope of the pass time against n says which issue port a phase is bound by.
template <int N>
__device__ __forceinline__ void inject_valu() {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_nop");
}
template <int N>
__device__ __forceinline__ void inject_salu() {
    int x = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(x) : : "scc");
}
#ifndef PMX_INJECT_VALU_ITEM
#define PMX_INJECT_VALU_ITEM 0
#endif
#ifndef PMX_INJECT_VALU_WALK
#define PMX_INJECT_VALU_WALK 0
#endif
#ifndef PMX_INJECT_SALU_WALK
#define PMX_INJECT_SALU_WALK 0
#endif

// A ligand's score: the float32 of the float64 mean the reference returns (graph_match.py:109), or that float64 itself (pmx_score_f64).
__device__ __forceinline__ void put_score(const ScreenParams &p, uint32_t li, double v) {
    // ([MI355X] A/B: the kernels always writing the float64 and a conversion kernel per chunk for pmx_score: 99.8 ms against 98.7-98.9 for this branch)
    if (p.flags & PMX_SCORES_F64) reinterpret_cast<double *>(p.scores)[li] = v;
    else p.scores[li] = (float)v;
}

__device__ inline int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// lane `lane` (wave-uniform) of v := value (this clang has no v_writelane builtin; a compare + select does it)
__device__ inline int wl(int v, int lane, int value) { return (int)(threadIdx.x & 63) == lane ? value : v; }
// The lane id as a value the optimiser cannot see through: address arithmetic derived from it stays inside the loop that uses it
// (hoisted out of the persistent loops it was kept live - spilled - for the whole kernel).
__device__ inline int lane_id() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)uni((int)(v >> 32)) << 32) | (uint64_t)(uint32_t)uni((int)(uint32_t)v);
}
template <typename T>
__device__ inline T *uniptr(T *p) {
    return reinterpret_cast<T *>(uni64(reinterpret_cast<uint64_t>(p)));
}
// Largest value of the wavefront, in every lane: butterfly inside the rows of 16 lanes (DPP), then the four rows.
__device__ inline float wave_max_f32(float v) {
    int x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false))); // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false))); // quad_perm [2,3,0,1]
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false))); // row_ror:4
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false))); // row_ror:8
    x = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Largest value over the lanes that stand for the same conformer (lane % G) in all 64 / G slots, in every lane: rotations
// inside the rows of 16 lanes (DPP), then the rows by the lane swaps of gfx950 (v_permlane16_swap / v_permlane32_swap) -
// no trip through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to).
template <int G>
__device__ __forceinline__ float slot_max_f32(float v) {
    if (G <= 1) { const int x = __float_as_int(v); v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false))); } // row_ror:1
    if (G <= 2) { const int x = __float_as_int(v); v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x122, 0xf, 0xf, false))); } // row_ror:2
    if (G <= 4) { const int x = __float_as_int(v); v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false))); } // row_ror:4
    if (G <= 8) { const int x = __float_as_int(v); v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false))); } // row_ror:8
    if (G <= 16) {
        const unsigned x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); // {rows 0 0 2 2, rows 1 1 3 3}
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    if (G <= 32) {
        const unsigned x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); // {lower half twice, upper half twice}
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
__device__ inline void wave_sync() { // LDS / global hand-over between the lanes of one wavefront
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
// Hand-over through LDS only (does not wait for outstanding global stores)
__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The smallest float32 that is not below x (NaN stays NaN).
__device__ inline float float_up(double x) {
    const float f = (float)x;
    if (!((double)f < x)) return f;
    const uint32_t b = __float_as_uint(f);
    return __uint_as_float(f > 0.f ? b + 1u : (f < 0.f ? b - 1u : 1u));
}
// The smallest bfloat16 that is not below f, as its 16 bits (NaN stays NaN; +inf beyond the largest finite one - an upper bound either way).
__device__ inline uint16_t bf16_up(float f) {
    const uint32_t b = __float_as_uint(f);
    if (f != f) return (uint16_t)0x7fc0u;
    const uint32_t hi = b >> 16;
    if ((b & 0xffffu) == 0u || (b >> 31)) return (uint16_t)hi; // exact, or negative: dropping low bits moves a negative value up
    return (uint16_t)(hi + 1u);
}
__device__ inline float bf16_value(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ inline float norm3f(float dx, float dy, float dz) { // np.linalg.norm of a float32 3-vector (ligand.py:349-351)
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

// ------------------------------------------------------------------------------------ fn_build_kernel
// Tabulates F_(sa, sb)(d) for every pair of node subsets on the grid x_i = i * h: quintic Hermite cells from F, F', F''
// at the two ends of a cell, evaluated in float64. `win` holds the exact pass windows of every cell (host, model-only).
// A subset pair with a zero weight sum scores NaN in the reference (0 * (1 / 0), match_utils.py:50-52,69): NaN cells.
__global__ void fn_build_kernel(DevModel M, Weights W, const uint32_t *sub_off, const uint8_t *sub_nodes, uint32_t NS, uint32_t ncell, float h,
                                const float2 *win, FnCell *cells, double rel_tol, double max_exponent) {
    const uint32_t fid = blockIdx.x;
    uint32_t sa, sb;
    if (M.symmetric) { // triangular: fid = sa (sa + 1) / 2 + sb, sb <= sa
        sa = (uint32_t)((sqrtf(8.f * (float)fid + 1.f) - 1.f) * 0.5f);
        while ((sa + 1) * (sa + 2) / 2 <= fid) ++sa;
        while (sa * (sa + 1) / 2 > fid) --sa;
        sb = fid - sa * (sa + 1) / 2;
    } else {
        sa = fid / NS, sb = fid - sa * NS;
    }
    const uint8_t *A = sub_nodes + sub_off[sa], *B = sub_nodes + sub_off[sb];
    const int nA = (int)(sub_off[sa + 1] - sub_off[sa]), nB = (int)(sub_off[sb + 1] - sub_off[sb]);
    const int Nm = M.Nm;
    bool a_nz = false, b_nz = false;
    for (int i = 0; i < nA; ++i) a_nz = a_nz || W.w[M.node_type[A[i]]] != 0.f;
    for (int i = 0; i < nB; ++i) b_nz = b_nz || W.w[M.node_type[B[i]]] != 0.f;
    const bool empty = nA == 0 || nB == 0;
    const bool nanfn = !empty && (!a_nz || !b_nz);
    const double inv_mn = empty ? 0.0 : 1.0 / (double)(nA * nB);
    for (uint32_t i = threadIdx.x; i < ncell; i += blockDim.x) {
        double f[2], d1[2], d2[2];
        for (int e = 0; e < 2; ++e) {
            const double x = (double)(i + e) * (double)h;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            if (!empty && !nanfn) {
                for (int ia = 0; ia < nA; ++ia) {
                    const int m = A[ia];
                    for (int ib = 0; ib < nB; ++ib) {
                        const int n = B[ib];
                        const float4 eg = M.edge[m * Nm + n]; // {mean, s, T, std}
                        const float wprod = W.w[M.node_type[m]] * W.w[M.node_type[n]];
                        const double coef = (double)(wprod / eg.w); // weights / stds in float32 (match_utils.py:65)
                        const double sd = (double)eg.w, z = (x - (double)eg.x) / sd;
                        const double g = exp(-0.5 * z * z);
                        s0 += coef * g;
                        s1 += coef * g * (-z / sd);
                        s2 += coef * g * ((z * z - 1.0) / (sd * sd));
                    }
                }
            }
            f[e] = s0 * inv_mn;
            d1[e] = s1 * inv_mn * (double)h;
            d2[e] = s2 * inv_mn * (double)h * (double)h;
        }
        const double df = f[1] - f[0];
        FnCell c;
        c.c[0] = (float)f[0];
        c.c[1] = (float)d1[0];
        c.c[2] = (float)(0.5 * d2[0]);
        c.c[3] = (float)(10.0 * df - 6.0 * d1[0] - 4.0 * d1[1] - 1.5 * d2[0] + 0.5 * d2[1]);
        c.c[4] = (float)(-15.0 * df + 8.0 * d1[0] + 7.0 * d1[1] + 1.5 * d2[0] - d2[1]);
        c.c[5] = (float)(6.0 * df - 3.0 * d1[0] - 3.0 * d1[1] - 0.5 * d2[0] + 0.5 * d2[1]);
        // worst deviation of the float32 polynomial from the function, relative to the function, at eight points inside the cell
        bool rough = i + 1 == ncell; // (the last cell is also where every distance beyond the grid lands)
        if (!empty && !nanfn && !rough) {
            for (int k = 0; k < 8 && !rough; ++k) {
                const double t = ((double)k + 0.5) * 0.125, x = ((double)i + t) * (double)h;
                double s0 = 0.0, e0 = 0.0;
                for (int ia = 0; ia < nA; ++ia) {
                    const int m = A[ia];
                    for (int ib = 0; ib < nB; ++ib) {
                        const int n = B[ib];
                        const float4 eg = M.edge[m * Nm + n];
                        const float wprod = W.w[M.node_type[m]] * W.w[M.node_type[n]];
                        const double z = (x - (double)eg.x) / (double)eg.w;
                        const double g = (double)(wprod / eg.w) * exp(-0.5 * z * z);
                        s0 += g;
                        e0 += g * (0.5 * z * z);
                    }
                }
                const double fx = s0 * inv_mn;
                const double px = (double)c.c[0] + t * ((double)c.c[1] + t * ((double)c.c[2] + t * ((double)c.c[3] + t * ((double)c.c[4] + t * (double)c.c[5]))));
                // ... and where the function is down to exp(-max_exponent) of its terms' peaks: the reference computes z and z^2 in
                // float32, which moves exp(-z^2 / 2) by up to 1.8e-7 z^2 / 2 of its value - rounding a smooth table cannot follow
                rough = fabs(px - fx) > rel_tol * fx || e0 > max_exponent * s0;
            }
        }
        c.c[5] = __uint_as_float((__float_as_uint(c.c[5]) & ~1u) | (rough ? 1u : 0u));
        if (nanfn) c.c[0] = __builtin_nanf("");
        const float2 w = win[(size_t)fid * ncell + i];
        c.lo = w.x;
        c.hi = w.y;
        float4 *planes = reinterpret_cast<float4 *>(cells);
        planes[(size_t)fid * ncell + i] = make_float4(c.c[0], c.c[1], c.c[2], c.c[3]);
        planes[(size_t)gridDim.x * ncell + (size_t)fid * ncell + i] = make_float4(c.c[4], c.c[5], c.lo, c.hi);
    }
}

// ------------------------------------------------------------------------------------------- LDS of a wave
// Frames nl - 3 .. nl - 2 - kTcLevels keep their children's totals in LDS: when the walker comes back to such a frame the
// remaining candidates are taken from there instead of being evaluated again (a third of all passes were re-evaluations).
#ifndef PMX_TC_LEVELS
#define PMX_TC_LEVELS 4
#endif
constexpr int kTcLevels = PMX_TC_LEVELS;
static_assert(kTcLevels >= 1 && kTcLevels <= 8, "cache slot number is three bits of Walk::hk");
template <int G>
struct WaveShape {
    uint32_t kp;     // candidates per level, padded
    uint32_t nc_cap; // node-candidate entries
    uint32_t off_cand, off_lcnt, off_nc, off_tot, off_pool, off_stat, off_task, off_tch, off_tc, off_cb, off_ub, bytes;
};
template <int G>
__host__ __device__ inline WaveShape<G> wave_shape(int K, int max_nodes) {
    WaveShape<G> w;
    w.kp = (uint32_t)((K + 3) & ~3);
    w.nc_cap = w.kp * (uint32_t)((max_nodes + 3) & ~3);
    uint32_t o = 512; // fixed part: type masks, level arrays
    w.off_cand = o;
    o += PMX_MAX_LEVELS * w.kp;
    w.off_lcnt = o;
    o += PMX_MAX_LEVELS * w.kp;
    o = (o + 15u) & ~15u;
    w.off_nc = o;
    o += w.nc_cap * 2;
    o = (o + 15u) & ~15u;
    w.off_tot = o;
    if (totals_in_lds<G>()) o += (PMX_MAX_LEVELS + 1) * G * 8;
    w.off_pool = o;
    o += G * 8;
    w.off_stat = o; // the wave's statistics (kept out of the registers)
    o += 208; // sizeof(WaveStats)
    w.off_task = o; // subtree record of the root of the ligand in work
    o += task_rec_bytes<G>();
    o = (o + 15u) & ~15u;
    w.off_tch = o; // totals of a frame's children (fused last two levels)
    // path_bound() keeps the tested child's totals in the first G entries and nothing else of this block: its level maxima live behind them
    // (the fused block, which fills all 64 entries, and path_bound() never run inside one another) - the room that saves is a fourth cached level
    {
        uint32_t blk = 64 * 8;
        if (cand_bounds<G>()) blk = blk > (uint32_t)(G * 8 + PMX_MAX_LEVELS * G * 4) ? blk : (uint32_t)(G * 8 + PMX_MAX_LEVELS * G * 4);
        o += blk;
    }
    w.off_tc = o; // the children's totals of the kTcLevels deepest unfused frames + their validity ballots
    if (totals_in_lds<G>()) o += kTcLevels * (64 * 8 + 8);
    w.off_cb = o; // candidates of a filtered frame that are still to visit, one 64-bit set per level (32 / 64 conformer lanes)
    if (64 / G <= 2) o += PMX_MAX_LEVELS * 8;
    w.off_ub = w.off_tch + G * 8; // path_bound(): the most a level can add, per conformer (inside the block of the children's totals, see above)
    w.bytes = o;
    return w;
}
// fixed part (512 bytes): tm[64] | lstart[20] lend[20] lk[20] pad[4] | ksum u16[24] | ncoff u16[24] | rowbase u32[20] | cand bits u64[20] | ksumtot, T | path staging u16[20]
constexpr uint32_t kOffTm = 0, kOffStart = 64, kOffEnd = 84, kOffK = 104, kOffKsum = 128, kOffNcoff = 176, kOffRow = 224, kOffBits = 304, kOffPath = 472;
static_assert(kOffBits + 8 * PMX_MAX_LEVELS + 8 <= kOffPath && kOffPath + 2 * PMX_MAX_LEVELS <= 512, "fixed LDS part");

// ------------------------------------------------------------------------------------------------- walker
// Iterative form of ClusterMatchTree.dfs_run (tree.py:55-104) with wave-uniform control. Frame f is the tree node whose
// children are the candidates of level f (frame 0 = root). State of the current frame in scalars: nm = matches on the
// path, mask = conformers still valid (tree.py:78-84), nb = next candidate to look at, mx = max_num_matches so far
// (tree.py:96-97), flags = {this node is a match, a candidate child existed, skip child done}.
//
// One *pass* evaluates the next 64 / G candidates b of the frame at once (slot s <-> candidate nb + s, lane c <-> conformer):
//   valid(b, c) = mask(c) and P[q -> (f, b)][c] > 0 for every matched ancestor q          (tree.py:78-84)
//   total(b, c) = (total(parent, c) + S[f][b][c]) + sum_q P[q -> (f, b)][c]   in float64   (tree.py:38-41)
// and the walker descends into the first candidate that exists (some conformer valid). After the return the remaining
// candidates are evaluated again from nb on - nothing is cached per frame, which is what keeps the state in registers.
// A frame at the last level is finished inside its pass: every existing candidate is a leaf that feeds the per-conformer
// maximum (graph_match.py:103-109), then the skip leaf (tree.py:98-101).
//
// Exactness of pruning and splitting. `num_matches(A) + max_num_matches(A)` (tree.py:98) is the largest match count of a leaf
// below A's candidate children, so the skip rule only asks whether a node with >= 5 matches exists there. For a child Y
// of a frame with >= 4 matches (Y holds >= 5): every ancestor's skip decision is settled by Y's existence, decisions
// inside Y's subtree depend on candidate existence only (nm + mx < 5 is never true there), and its leaves only feed a
// per-conformer maximum. So Y may be (i) dropped when no leaf below it can exceed the maxima found so far - leaf totals
// are bounded by total(Y) + W[Y] (build_bounds) - and (ii) walked by another wavefront (task queue); both count as
// "returned >= 1" for the parent. A child Y with fewer than 5 matches whose bound fails feeds no maximum either; what its
// parent's decision (nm + mx < 5) needs from it is whether a node with >= 5 matches exists below it - probe() - and nothing
// once mx has reached 5 - nm through a sibling. The order in which children are visited changes neither maxima nor
// existence. Scores and every skip decision stay what the reference computes.
struct WaveStats { // lives in LDS, updated by lane 0
    unsigned long long frames, passes, over, items, exact, longest, tasks, overflow;
    unsigned long long cyc_scan, cyc_tables, cyc_bounds, cyc_walk, exactv, npath, pad[2]; // s_memtime ticks per phase | self items evaluated term by term
    unsigned long long dbg[8]; // instrumented builds (-DPMX_COUNTERS): see walk()
    unsigned long long dead, pad3; // pair entries the dead-entry test of build_tables settled without computing them
};
static_assert(sizeof(WaveStats) == 208, "WaveStats layout");

template <int G>
struct Walk {
    // tables of the job
    const unsigned char *Sb, *Pb, *Rb, *Wb, *Vb, *OBb; // (CI follows OB)
    int nl;
    uint32_t ksumtot;
    bool path_on = false; // the job's tables fit the wave's path-sum buffer: path_bound() may be used
    int hk, hks, hrow; // lane l: k[l], ksum[l], rowbase[l]
    // path: lane q holds match q
    int matB = 0, matKA = 0; // entry(match, x) - x = rowbase[j] + a * nd_j - ksum[j + 1] | a << 8 | j << 16
    // stack: lane f holds frame f
    int stA = 0, stB = 0, stC = 0; // mask lo, mask hi, nb | mx << 8 | flags << 16 | nm << 24
    double best = 0.0, flushed = 0.0;
    uint32_t frames = 0, passes = 0, npath = 0, ndrop = 0; // (npath / ndrop: path_bound() calls, children it dropped)
    // current frame (its state is in lane f of the stack like every other frame's; a walk can be interrupted and resumed, see kOverBudget)
    int f = 0, f0 = 0;
};
// entry((f, b) -> x) - x for a match (f, b): what lane q of Walk::matB holds for match q
template <int G>
__device__ __forceinline__ int match_base(const Walk<G> &w, int f, int b) {
    const int k1 = rl(w.hks, f + 1);
    return rl(w.hrow, f) + b * ((int)w.ksumtot - k1) - k1;
}
constexpr int kOverBudget = -1;
template <int G>
__host__ __device__ constexpr uint64_t group_mask() {
    return G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
}

// per-level facts in bits 8.. of Walk::hk: the last level | its parent level with the children's leaves fused into its pass |
// a level whose children's totals are cached (slot number in bits 12..14)
constexpr int kLvLeaf = 256, kLvFuse = 512, kLvCache = 1024;
constexpr unsigned kMatched = 1, kAny = 2, kSkipped = 4, kCached = 8, kFused = 16, kFiltered = 32, kPath = 64; // kPath: the path sums of this frame's matches are in the wave's buffer
constexpr double kBoundSlack = 1.0 + 1e-9;
#ifndef PMX_PATH_MIN_LEVELS
#define PMX_PATH_MIN_LEVELS 3
#endif
constexpr int kPathMinLevels = PMX_PATH_MIN_LEVELS;
#ifndef PMX_PATH_WINDOWS
#define PMX_PATH_WINDOWS 3
#endif
constexpr int kPathWindows = PMX_PATH_WINDOWS; // windows of 64 / G candidates whose loads go out together in path_bound() // path_bound() is asked where the frame's level and at least this many - 1 more lie below // covers the float64 rounding of the sums the bound is compared with


// Row loops of the walker: `load(q)` for q = 0 .. n - 1 go out kRowBatch at a time and `use(value)` takes them in order; what is
// left at the end goes out as ONE batch too. (A remainder loop that loads one row, waits, uses it and loads the next costs a
// memory round trip per row: with 7 matched ancestors - the average of a table pass - that was four round trips instead of two.)
#ifndef PMX_ROW_BATCH
#define PMX_ROW_BATCH 8
#endif
constexpr int kRowBatch = PMX_ROW_BATCH;
template <int N, typename Load, typename Use>
__device__ __forceinline__ void rows_batch(int q, Load &&load, Use &&use) {
    decltype(load(0)) v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = load(q + u);
#pragma unroll
    for (int u = 0; u < N; ++u) use(v[u]);
}
template <typename Load, typename Use>
__device__ __forceinline__ void for_rows(int n, Load &&load, Use &&use) {
    int q = 0;
    for (; q + kRowBatch <= n; q += kRowBatch) rows_batch<kRowBatch>(q, load, use);
    static_assert(kRowBatch == 4 || kRowBatch == 8, "remainder cases");
    if (kRowBatch == 8 && n - q >= 4) {
        switch (n - q) {
        case 7: rows_batch<7>(q, load, use); break;
        case 6: rows_batch<6>(q, load, use); break;
        case 5: rows_batch<5>(q, load, use); break;
        default: rows_batch<4>(q, load, use); break;
        }
        return;
    }
    switch (n - q) {
    case 3: rows_batch<3>(q, load, use); break;
    case 2: rows_batch<2>(q, load, use); break;
    case 1: rows_batch<1>(q, load, use); break;
    default: break;
    }
}

// Can the child (frame f, candidate `cand`, conformer mask `cmask`) of the current frame, which holds nm matches, still
// reach 5 matches - i.e. does the reference's tree hold a node with >= 5 matches below it? The same depth-first search on
// validity alone (no totals), stopped at the first such node; it follows the skip rule of tree.py:98, under which a node
// with >= 5 matches is reached whenever a valid assignment with >= 5 matches exists (see walk()). Uses the stack and path
// lanes above the current frame, which the walker re-writes when it descends itself.
template <int G>
__device__ __forceinline__ bool probe(Walk<G> &w, int f, int nm, int cand, uint64_t cmask, uint32_t &passes) {
    const int lane = lane_id();
    const int nl = w.nl;
    const unsigned char *Vb = w.Vb;
    if (nm + 1 >= 5) return true;
    // DP[x]: no chain of pairwise compatible candidates that starts with x holds more than DP[x] of them (chain_lengths()), so a
    // node with 5 matches lies below a path of nm matches through x only if DP[x] >= 5 - nm. The child itself first, then every
    // candidate the search would try: what they rule out is not there to find.
    const unsigned char *DP = w.OBb + (round16((uint64_t)ob_rows<G>((uint32_t)nl) * w.ksumtot * G * ob_elt_bytes<G>()) + (size_t)round16((uint64_t)w.ksumtot));
    if (uni((int)DP[rl(w.hks, f) + cand]) < 5 - nm) return false;
    // enter the child
    const int fbase = f;
    w.matB = wl(w.matB, nm, match_base(w, f, cand));
    w.matKA = wl(w.matKA, nm, (cand << 8) | (f << 16));
    ++f;
    ++nm;
    uint64_t mask = cmask;
    int nb = 0, mx = 0;
    unsigned flags = kMatched;
    for (;;) {
        int ret;
        if (f == nl) { // below the last level: a leaf
            ret = (flags & kMatched) ? 1 : 0;
        } else {
            const int kf = rl(w.hk, f) & 255, ksf = rl(w.hks, f);
            bool descended = false;
            if (nb < kf) {
                const int ebv = w.matB + ksf;
                // every candidate of the level at once, lane l <-> candidate l: which exist as children - some conformer of
                // the frame has every pair entry > 0 - is one AND of V masks per matched ancestor (no table row is read)
                constexpr uint32_t VB = vmask_bytes<G>();
                bool in = lane >= nb && lane < kf;
                const uint32_t lo_ = (uint32_t)(lane < kf ? lane : 0) * VB;
                const int reach = DP[(uint32_t)ksf + (lane < kf ? (uint32_t)lane : 0u)];
                auto vload = [&](int q) -> unsigned long long {
                    const unsigned char *ve = Vb + (uint32_t)rl(ebv, q) * VB + lo_;
                    if (G <= 8) return *ve;
                    else if (G == 16) return *reinterpret_cast<const uint16_t *>(ve);
                    else if (G == 32) return *reinterpret_cast<const uint32_t *>(ve);
                    else return *reinterpret_cast<const unsigned long long *>(ve);
                };
                unsigned long long m = mask;
                for_rows(nm, vload, [&](unsigned long long v) { m &= v; });
                in = in && reach >= 5 - nm;
                const unsigned long long ex = __ballot(in && m != 0ull);
                ++passes;
                if (!ex) {
                    nb = kf;
                } else {
                    flags |= kAny;
                    if (nm + 1 >= 5) return true; // a node with 5 matches
                    const int bsel = __ffsll(ex) - 1;
                    nb = bsel + 1;
                    w.stA = wl(w.stA, f, (int)(uint32_t)mask);
                    if (G > 32) w.stB = wl(w.stB, f, (int)(uint32_t)(mask >> 32));
                    w.stC = wl(w.stC, f, nb | (mx << 8) | ((int)flags << 16) | (nm << 24));
                    w.matB = wl(w.matB, nm, match_base(w, f, bsel));
                    w.matKA = wl(w.matKA, nm, (bsel << 8) | (f << 16));
                    mask = (uint64_t)(uint32_t)rl((int)(uint32_t)m, bsel);
                    if (G > 32) mask |= (uint64_t)(uint32_t)rl((int)(uint32_t)(m >> 32), bsel) << 32;
                    ++f;
                    ++nm;
                    flags = kMatched;
                    nb = 0;
                    mx = 0;
                    descended = true;
                }
            }
            if (descended) continue;
            if (!(flags & kSkipped) && (!(flags & kAny) || nm + mx < 5)) { // skip child (tree.py:98-101)
                flags |= kSkipped;
                w.stA = wl(w.stA, f, (int)(uint32_t)mask);
                if (G > 32) w.stB = wl(w.stB, f, (int)(uint32_t)(mask >> 32));
                w.stC = wl(w.stC, f, nb | (mx << 8) | ((int)flags << 16) | (nm << 24));
                ++f;
                flags = 0;
                nb = 0;
                mx = 0;
                continue;
            }
            ret = mx + ((flags & kMatched) ? 1 : 0);
        }
        --f;
        if (f <= fbase) return false; // the child's subtree is exhausted: no node with 5 matches
        const int sc = rl(w.stC, f);
        mask = (uint64_t)(uint32_t)rl(w.stA, f);
        if (G > 32) mask |= (uint64_t)(uint32_t)rl(w.stB, f) << 32;
        nb = sc & 255;
        mx = (sc >> 8) & 255;
        flags = (unsigned)(sc >> 16) & 255u;
        nm = (sc >> 24) & 255;
        mx = mx > ret ? mx : ret;
    }
}

// Path-aware bound (round 4). W[(f, b)] bounds what the levels below f can add under a child Y = (f, b) with every level
// above f at its *maximum* pair entry; with seven matches on the path that is far from what they do add. Here the deeper
// candidates are priced with the pair entries of the matches actually on the path: for a candidate x = (l, b') of a level l > f
//     v(x)[c] = OB[f][x][c] + sum_{q on the path, Y included} P[q -> x][c]        (left out unless every such entry is > 0)
// (OB: x's self entry + the maxima of the levels between f and l, build_bounds), a level adds at most max(0, max_x v(x)), and
// the subtree below Y at most the sum of that over the levels l > f: no leaf below Y exceeds total(Y) + that. Nothing else
// changes - a child that fails is dropped exactly as one that fails the W test (see walk(): "exactness"). The pair sums of
// the path are kept per match count in a buffer of the wave (pa[matches][candidate][conformer], float32: an upper bound needs
// no more; the sums are of non-negative terms, so rounding to nearest loses at most 2^-24 per addition, which the final
// factor covers) and extended by Y's entries here - they are the sums of Y's own frame when the walker goes there.
// On the bench library the walker enters 4 times fewer frames with it (tests/bound_study: 272 -> 69 per ligand), 9-13 times
// fewer on the fixture pockets.
__device__ __forceinline__ float pa_get(const PaElt *row, size_t i) {
    if constexpr (kSlimPA) return bf16_value(row[i]);
    else return row[i];
}
__device__ __forceinline__ void pa_put(PaElt *row, size_t i, float v) { // (-inf stays -inf; sums are rounded up: an upper bound)
    if constexpr (kSlimPA) row[i] = bf16_up(v);
    else row[i] = v;
}
template <int G>
__device__ __forceinline__ bool path_bound(const Walk<G> &w, const ScreenParams &p, PaElt *pa, float *ub, const double *tch, const unsigned long long *pool,
                                           int f, int nm, int bsel, uint64_t cmask) {
    constexpr int SLOTS = 64 / G;
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    const int nl = w.nl;
    const uint32_t ksumtot = w.ksumtot;
    const uint32_t x0 = (uint32_t)rl(w.hks, f + 1); // first candidate of the levels below f
    const float *Pf = reinterpret_cast<const float *>(w.Pb) + (long)match_base<G>(w, f, bsel) * G; // Y's entries: Pf[x * G + c] (the base may be negative, base + x is not)
    const unsigned char *OB = w.OBb + (size_t)f * ksumtot * G * ob_elt_bytes<G>();
    const unsigned char *LV = w.OBb + round16((uint64_t)nl * ksumtot * G * ob_elt_bytes<G>());
    const PaElt *pin = pa + (size_t)nm * ksumtot * G;
    PaElt *pout = pa + (size_t)(nm + 1) * ksumtot * G;
    for (int i = lane; i < (nl - f - 1) * G; i += 64) ub[(f + 1) * G + i] = 0.f;
    lds_sync();
    // kPathWindows windows of SLOTS candidates per trip, everything of a window in one round of loads (the entries of Y with the deeper
    // candidates are one contiguous run: no lookup in front of the pair rows)
    for (uint32_t x = x0; x < ksumtot; x += kPathWindows * SLOTS) {
        uint32_t xx[kPathWindows], lv[kPathWindows];
        float ob[kPathWindows], have[kPathWindows], pv[kPathWindows];
        bool on[kPathWindows];
#pragma unroll
        for (int u = 0; u < kPathWindows; ++u) {
            on[u] = x + (uint32_t)(u * SLOTS + s) < ksumtot;
            xx[u] = on[u] ? x + (uint32_t)(u * SLOTS + s) : x0;
            lv[u] = LV[xx[u]];
            if (ob_elt_bytes<G>() == 2) ob[u] = bf16_value(reinterpret_cast<const uint16_t *>(OB)[(size_t)xx[u] * G + c]);
            else ob[u] = reinterpret_cast<const float *>(OB)[(size_t)xx[u] * G + c];
            have[u] = nm ? pa_get(pin, (size_t)xx[u] * G + c) : 0.f;
            pv[u] = Pf[(size_t)xx[u] * G + c];
        }
#pragma unroll
        for (int u = 0; u < kPathWindows; ++u) {
            const float sum = pv[u] > 0.f ? have[u] + pv[u] : -__builtin_inff(); // (-inf stays -inf: a candidate out for this conformer stays out)
            if (on[u]) {
                pa_put(pout, (size_t)xx[u] * G + c, sum);
                const float v = fmaxf(sum + ob[u], 0.f); // (a NaN self entry - zero weights - can raise no maximum: 0)
                atomicMax(reinterpret_cast<unsigned int *>(ub) + lv[u] * G + (uint32_t)c, __float_as_uint(v));
            }
        }
    }
    lds_sync();
    float below = 0.f;
    for (int l = f + 1; l < nl; ++l) below = below + ub[l * G + c];
    const double bound = (double)below * (1.0 + 4e-6);
    const double pooled = __longlong_as_double((long long)pool[c]);
    const double bp = pooled > w.best ? pooled : w.best;
    return __ballot(((cmask >> c) & 1ull) && (tch[c] + bound) * kBoundSlack > bp) != 0ull;
}

// The path-aware test where a pass holds one or two candidates (32 / 64 conformer lanes). There the test above would move a row of
// every deeper candidate per evaluation (250-300 candidates x 64 conformers of the stress model: a quarter of a megabyte). But under
// a path of five matches hardly any deeper candidate is still compatible with ALL of them (a pair of candidates is compatible in
// a fifth of the cases on that model), and which ones are is in the V masks: with the lanes spread over the deeper candidates, one
// AND of masks per match on the path - the child Y = (f, bsel) included - lists them, 64 candidates per trip and 8 bytes per
// candidate and match. Every candidate x left is priced at BF[x] = base(x) rounded up - its self entry plus, for EVERY level above
// its own, the largest pair entry any candidate of that level has with it (build_bounds): no leaf adds more for x whatever is matched
// above it - for the conformers its mask still holds; a level adds at most the largest of its candidates, the subtree below Y at
// most the sum over the levels. Candidate numbers ascend with the level, so the level maxima are a running maximum: no LDS. A child
// that fails is dropped like one that fails the level-bound test (it holds >= 5 matches: nothing else is asked of it).
// tests/bound_study (model_stress64, 16 ligands x 64 conformers): the level bound in index order enters 10 036 frames per ligand,
// this test under >= 5 matches 1 187 with 1 344 evaluations (with the pair entries of the path and the OB table as above: 757).
template <int G>
__device__ __forceinline__ bool path_bound_wide(const Walk<G> &w, const double t /* the child's total, in the lanes of its slot: */, const bool sel,
                                                const unsigned long long *pool, int f, int nm, int bsel, uint64_t cmask) {
    static_assert(G >= 32, "lanes over candidates, conformer masks of 32 / 64 bits");
    constexpr uint32_t VB = vmask_bytes<G>();
    const int lane = lane_id();
    const int c = lane % G;
    const uint32_t ksumtot = w.ksumtot;
    const uint32_t x0 = (uint32_t)rl(w.hks, f + 1); // first candidate of the levels below f
    const float *BF = reinterpret_cast<const float *>(w.OBb);
    const unsigned char *LV = w.OBb + round16((uint64_t)ksumtot * G * 4u);
    const unsigned char *Vy = w.Vb + (long)match_base<G>(w, f, bsel) * (long)VB; // Y's masks: Vy + x * VB (the base may be negative, base + x is not)
    float below = 0.f, cur = 0.f;
    int cur_lv = -1;
    for (uint32_t xb = x0; xb < ksumtot; xb += 64u) {
        const uint32_t x = xb + (uint32_t)lane;
        const bool in = x < ksumtot;
        const uint32_t xo = (in ? x : x0) * VB;
        auto vload = [&](int q) -> unsigned long long {
            const unsigned char *ve = w.Vb + (long)rl(w.matB, q) * (long)VB + xo;
            if (G == 32) return *reinterpret_cast<const uint32_t *>(ve);
            else return *reinterpret_cast<const unsigned long long *>(ve);
        };
        unsigned long long m = cmask;
        if (G == 32) m &= *reinterpret_cast<const uint32_t *>(Vy + xo);
        else m &= *reinterpret_cast<const unsigned long long *>(Vy + xo);
        const int lvl = LV[in ? x : x0];
        for_rows(nm, vload, [&](unsigned long long v) { m &= v; });
        unsigned long long ex = __ballot(in && m != 0ull);
        while (ex) { // the candidates still compatible with the whole path, in ascending order
            const int xl = __ffsll(ex) - 1;
            ex &= ex - 1ull;
            uint64_t mm = (uint64_t)(uint32_t)rl((int)(uint32_t)m, xl);
            if (G > 32) mm |= (uint64_t)(uint32_t)rl((int)(uint32_t)(m >> 32), xl) << 32;
            const int lv = rl(lvl, xl);
            const float bf = BF[(size_t)(xb + (uint32_t)xl) * G + c];
            if (lv != cur_lv) {
                below = below + cur;
                cur = 0.f;
                cur_lv = lv;
            }
            const float v = ((mm >> c) & 1ull) ? bf : 0.f;
            cur = fmaxf(cur, v); // (a NaN base - zero weights - raises no maximum)
        }
    }
    below = below + cur;
    const double bound = (double)below * (1.0 + 4e-6);
    const double pooled = __longlong_as_double((long long)pool[c]);
    const double bp = pooled > w.best ? pooled : w.best;
    return __ballot(sel && ((cmask >> c) & 1ull) && (t + bound) * kBoundSlack > bp) != 0ull;
}

// The path sums of the wave's buffer for a job that starts with matches on its path (a queued subtree): the rows of match
// after match, as path_bound() would have left them.
template <int G>
__device__ __forceinline__ void path_sums_of_root(const Walk<G> &w, PaElt *pa, int nm0) {
    constexpr int SLOTS = 64 / G;
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    const uint32_t ksumtot = w.ksumtot;
    for (int q = 0; q < nm0; ++q) {
        const uint32_t jq = ((uint32_t)rl(w.matKA, q) >> 16) & 255u;
        const uint32_t x0 = (uint32_t)rl(w.hks, (int)jq + 1);
        const float *Pq = reinterpret_cast<const float *>(w.Pb) + (long)rl(w.matB, q) * G;
        const PaElt *pin = pa + (size_t)q * ksumtot * G;
        PaElt *pout = pa + (size_t)(q + 1) * ksumtot * G;
        for (uint32_t x = x0; x < ksumtot; x += SLOTS) {
            const bool on = x + (uint32_t)s < ksumtot;
            const uint32_t xx = on ? x + (uint32_t)s : x0;
            const float pv = Pq[(size_t)xx * G + c];
            const float have = q ? pa_get(pin, (size_t)xx * G + c) : 0.f;
            if (on) pa_put(pout, (size_t)xx * G + c, pv > 0.f ? have + pv : -__builtin_inff());
        }
        wave_sync(); // (the next match reads what this one wrote)
    }
}

template <int G>
__device__ __forceinline__ int walk(Walk<G> &w, const ScreenParams &p, double *tot, unsigned long long *pool, uint16_t *pathbuf, double *tch, double *tc,
                                    unsigned long long *cbl, PaElt *pa, float *ub, uint32_t rec16 /* arena record of the job (exports refer to it) */, bool export_mode,
                                    unsigned long long budget, uint32_t wave_id, WaveStats *stat) {
    constexpr int SLOTS = 64 / G;
    constexpr int PSH = G == 1 ? 2 : G == 2 ? 3 : G == 4 ? 4 : G == 8 ? 5 : G == 16 ? 6 : G == 32 ? 7 : 8; // log2 bytes of an entry
    constexpr uint64_t GM = group_mask<G>();
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    const uint32_t lane_off = (uint32_t)lane * 4u; // (s * G + c) floats: candidate nb + s, conformer c
    const int nl = w.nl;
    const unsigned char *Sb = w.Sb, *Pb = w.Pb, *Wb = w.Wb, *Vb = w.Vb;
    const int bound_from = (PMX_WFLAGS(p) & 4) ? 255 : 4; // matches on the path from which children are bound-tested
    const bool no_filter = (PMX_WFLAGS(p) & 128) != 0;

    const uint32_t budget32 = (export_mode || budget > 0xfffffff0ull) ? 0xffffffffu : (uint32_t)budget; // (a walk of 2^32 passes does not end in this life)
    const int f0 = w.f0;
    // The only scalar carried from one iteration to the next is the frame number: every frame's state - the current one's
    // too - lives in lane f of stA / stB / stC and is read at the top of an iteration and written back at its end. (With the
    // current frame in scalars of its own, a third of the walker's instructions were copies between registers where the many
    // paths of the loop meet.) One iteration = one pass over the frame's next candidates, or the end of the frame.
    int f = w.f;
    int ret = 0;
    // The walkers of a split ligand (its subtrees, and the walk that queued them) trade maxima through the ligand's record
    // while they run, not only when they end: one returning atomic maximum per conformer every kShareEvery passes gives this
    // wave's maxima to the others and theirs to this wave's bound test. (Maxima of leaves of the same tree: exact.)
#ifndef PMX_SHARE_EVERY
#define PMX_SHARE_EVERY 16
#endif
    constexpr uint32_t kShareEvery = PMX_SHARE_EVERY;
    uint32_t next_share = w.passes + kShareEvery;
#ifdef PMX_COUNTERS
    uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // fused passes | fused children | cached passes | leaf passes | other passes from the tables | descents | ancestors over table passes | shares
#define PMX_COUNT(i, n) do { if (PMX_COUNTERS == 1 && (i) < 6) dbg[i] += (uint32_t)(n); } while (0)
    auto flush_dbg = [&]() {
        if (lane == 0)
            for (int i = 0; i < 6; ++i) stat->dbg[i] += dbg[i];
    };
#else
#define PMX_COUNT(i, n)
    auto flush_dbg = [&]() {};
#endif
    for (;;) {
        inject_valu<PMX_INJECT_VALU_WALK>();
        inject_salu<PMX_INJECT_SALU_WALK>();
        if (w.passes > budget32) { // over budget: the caller moves the job's tables to the arena and resumes in export mode
            w.f = f;
            flush_dbg();
            return kOverBudget;
        }
        if (rec16 != 0u && w.passes >= next_share && !(PMX_WFLAGS(p) & 8192)) {
            next_share = w.passes + kShareEvery;
            PMX_COUNT(7, 1);
            if (s == 0) {
                unsigned long long *gb = reinterpret_cast<unsigned long long *>(p.arena + (size_t)rec16 * 16 + sizeof(RecHeader));
                const unsigned long long mine = pool[c];
                const unsigned long long theirs = mine ? atomicMax(&gb[c], mine) : __hip_atomic_load(&gb[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (theirs > mine) pool[c] = theirs;
            }
            lds_sync();
        }
        const int sc = rl(w.stC, f);
        int nb = sc & 255, mx = (sc >> 8) & 255;
        unsigned flags = (unsigned)(sc >> 16) & 255u;
        const int nm = (sc >> 24) & 255;
        uint64_t mask = (uint64_t)(uint32_t)rl(w.stA, f);
        if (G > 32) mask |= (uint64_t)(uint32_t)rl(w.stB, f) << 32;
        // what is fixed per level was worked out once per job (prepare_walk): bits 8.. of the level's entry
        const int hv = rl(w.hk, f);
        const int kf = hv & 255, ksf = rl(w.hks, f);
        const bool leaf_level = (hv & kLvLeaf) != 0;
        // ordered frames: of the candidates of a pass (a window of the frame's candidates) the walker visits the surviving child
        // with the largest bound first; rem = the slots of the window not visited yet, in lane f of stB
        constexpr bool ORD = G >= 2 && G <= 32;
        const bool ordered0 = ORD && !leaf_level && !(hv & kLvFuse) && !(PMX_WFLAGS(p) & 2048);
        uint32_t rem = 0xffffffffu;
        if (ORD) rem = (uint32_t)rl(w.stB, f);
        if (nb < kf) {
            // ---------------------------------------------------------------- one pass over candidates nb .. nb + SLOTS - 1
            const double tparent = tot[nm * G + c];
            // the bound row and the pooled maxima go out with the table loads (one memory round trip per pass, not two); frames
            // f < nl only, so row f + 1 exists
            const bool bounded = nm >= bound_from && !leaf_level;
            double pooled = 0.0;
            if (bounded || ordered0) pooled = __longlong_as_double((long long)pool[c]);
            // pair-table rows of the matched ancestors against level f: lane q
            const int ebv = w.matB + ksf;
            // A frame with more candidates than slots is *filtered* first: which candidates exist as children - some conformer of
            // the frame has every pair entry > 0 - is read off the V masks with the lanes spread over candidates, and the passes
            // then take the existing candidates only, SLOTS at a time (most candidates do not exist: without this a frame of a
            // large model, or of a 64-conformer library with one slot per pass, spends its passes on them).
            bool filt = false;
            unsigned long long cb = 0, cb_rest = 0;
            int bvec = nb + s; // the candidate of this lane's slot
            if constexpr (SLOTS <= 2) { // (with 8 slots - the 8-conformer shape - the filter's own pass costs more than it saves: measured)
                filt = kf > SLOTS && nm > 0 && !no_filter;
                if (filt) {
                    if (!(flags & kFiltered)) {
                        constexpr uint32_t VB = vmask_bytes<G>();
                        const bool in = lane < kf;
                        const uint32_t lo_ = (uint32_t)(in ? lane : 0) * VB;
                        unsigned long long m = mask;
                        auto vload = [&](int q) -> unsigned long long {
                            const unsigned char *ve = Vb + (uint32_t)rl(ebv, q) * VB + lo_;
                            if (G <= 8) return *ve;
                            else if (G == 16) return *reinterpret_cast<const uint16_t *>(ve);
                            else if (G == 32) return *reinterpret_cast<const uint32_t *>(ve);
                            else return *reinterpret_cast<const unsigned long long *>(ve);
                        };
                        for_rows(nm, vload, [&](unsigned long long v) { m &= v; });
                        cb = __ballot(in && m != 0ull);
                        flags |= kFiltered;
                        ++w.passes;
                        if (cb == 0ull) { // no child exists: the frame's candidates are done
                            w.stC = wl(w.stC, f, 255 | (mx << 8) | ((int)flags << 16) | (nm << 24));
                            continue;
                        }
                    } else {
                        cb = uni64(cbl[f]);
                    }
                    unsigned long long x = cb;
                    bvec = 255;
#pragma unroll
                    for (int ss = 0; ss < SLOTS; ++ss) {
                        const int bb = x ? __ffsll(x) - 1 : 255;
                        x &= x - 1ull;
                        bvec = s == ss ? bb : bvec;
                    }
                    cb_rest = x;
                }
            }
            const bool ordered = ordered0 && !filt;
            const bool on = bvec < kf;
            const int b_first = filt ? (cb ? __ffsll(cb) - 1 : 0) : nb; // a candidate idle slots may read (in bounds)
            // the candidate's bound goes out with the table loads (one memory round trip per pass, not two)
            double rbound = 0.0;
            if (bounded || (ordered && cand_bounds<G>())) { // (the bound also orders the children of frames it cannot drop yet)
                if constexpr (cand_bounds<G>()) {
                    if (kSlimBounds) rbound = (double)*reinterpret_cast<const float *>(Wb + (((uint32_t)(ksf + (on ? bvec : b_first))) << PSH) + 4u * (uint32_t)c);
                    else rbound = *reinterpret_cast<const double *>(Wb + (((uint32_t)(ksf + (on ? bvec : b_first))) << (PSH + 1)) + 8u * (uint32_t)c);
                }
                else rbound = *reinterpret_cast<const double *>(w.Rb + (((uint32_t)(f + 1) << (PSH + 1)) + 8u * (uint32_t)c));
            }
            double t;
            bool valid;
            // cache slot of this frame: the kTcLevels frames above the fused one, one pass wide
            const int tci = (hv >> 12) & 7;
            const bool cacheable = (hv & kLvCache) != 0;
            if (cacheable && (flags & kCached)) { // back from a child: the remaining candidates as evaluated on the way in
                const unsigned long long vb0 = *reinterpret_cast<const unsigned long long *>(tc + kTcLevels * 64 + tci);
                const int src = lane + nb * G;
                t = tc[tci * 64 + (on ? src : lane)];
                valid = on && ((uni64(vb0) >> src) & 1ull);
                PMX_COUNT(2, 1);
            } else {
                PMX_COUNT(leaf_level ? 3 : 4, 1);
                PMX_COUNT(6, nm);
                const uint32_t bo = ((uint32_t)(on ? bvec : b_first) << PSH) + (uint32_t)c * 4u; // idle slots read an existing candidate
                const float self = *reinterpret_cast<const float *>(Sb + (((uint32_t)ksf << PSH) + bo));
                float lo = 1.f; // smallest pair entry: the candidate is valid for this conformer iff every entry is > 0 (tree.py:81)
                double sum = 0.0;
                for_rows(
                    nm, [&](int q) { return *reinterpret_cast<const float *>(Pb + (((uint32_t)rl(ebv, q) << PSH) + bo)); },
                    [&](float v) {
                        lo = fminf(lo, v);
                        sum += (double)v;
                    });
                // (v_min_f32 skips a NaN entry - a zero-weight pair, match_utils.py:50-52 - but the sum does not: NaN is not > 0)
                valid = on && ((mask >> c) & 1ull) && lo > 0.f && sum == sum;
                t = (tparent + (double)self) + sum; // parent + self + accumulated pair (tree.py:38-41)
            }
            if (ordered) valid = valid && ((rem >> s) & 1u);
            const unsigned long long vb = __ballot(valid);
            if (cacheable && !(flags & kCached)) { // first pass of the frame (a cached frame has one window)
                tc[tci * 64 + lane] = t;
                if (lane == 0) *reinterpret_cast<unsigned long long *>(tc + kTcLevels * 64 + tci) = vb;
                flags |= kCached;
            }
            ++w.passes;
            if (vb) flags |= kAny;
            unsigned long long ab = vb;
#if defined(PMX_COUNTERS) && PMX_COUNTERS == 2
            const unsigned long long dbg_vb = vb;
#endif
            // Children with fewer than 5 matches are bound-tested too where the frame is ordered: nothing below a child that
            // fails can raise a maximum, so all the frame still needs from it is whether it reaches 5 matches (tree.py:98) -
            // nothing at all once another child has (max_num_matches is a maximum), else what probe() answers.
            const bool shallow = ordered && cand_bounds<G>() && nm < 4 && bound_from != 255 && !(PMX_WFLAGS(p) & 4096);
            if ((bounded || shallow) && vb) { // drop the children that cannot raise a maximum
                const double bp = pooled > w.best ? pooled : w.best;
                ab = __ballot(valid && (t + rbound) * kBoundSlack > bp);
            }
#if defined(PMX_COUNTERS) && PMX_COUNTERS == 2
            { // what the bound test does: [0] passes without an existing child [1] passes whose existing children are all dropped [2] existing children [3] survivors [4] passes under >= 5 matches [5] survivors under >= 5 matches [6] existing under >= 5 [7] passes back from a child (cached)
                auto slots = [&](unsigned long long b) { int n = 0; for (int ss = 0; ss < SLOTS; ++ss) n += ((b >> (ss * G)) & GM) ? 1 : 0; return n; };
                dbg[0] += dbg_vb == 0;
                dbg[1] += dbg_vb != 0 && ab == 0;
                dbg[2] += slots(dbg_vb);
                dbg[3] += slots(ab);
                dbg[4] += nm >= 4;
                dbg[5] += nm >= 4 ? slots(ab) : 0;
                dbg[6] += nm >= 4 ? slots(dbg_vb) : 0;
                dbg[7] += (cacheable && (flags & kCached) && !(hv & 0)) ? 0 : 0;
            }
#endif
            int probe_slot = -1; // a child of this pass whose reach is probed (one call site)
            bool probe_rem_done = false;
            bool handled = false, pending = false;
            if (shallow && ab != vb) {
                const bool slot_vb = ((vb >> (s * G)) & GM) != 0, slot_ab = ((ab >> (s * G)) & GM) != 0;
                const unsigned long long dropped = __ballot(c == 0 && slot_vb && !slot_ab);
                if (dropped) {
                    if (mx >= 5 - nm) { // a sibling reached 5 matches already: the dropped children change nothing
                        rem &= ~(uint32_t)__ballot(lane < SLOTS && ((dropped >> ((lane * G) & 63)) & 1ull));
                    } else if (!ab) { // nothing left to walk: the frame has to know
                        probe_slot = (__ffsll(dropped) - 1) / G;
                        handled = true;
                    } else { // the survivors first: one of them reaching 5 matches saves the probes
                        pending = true;
                    }
                }
            }
            if (handled) {
            } else if (leaf_level) {
                if (valid && t > w.best) w.best = t; // graph_match.py:105-108
                nb += SLOTS;
                cb = cb_rest;
            } else if (hv & kLvFuse) {
                // The children of this frame are frames of the last level, whose children are leaves: finish all of them here.
                // Lane (s', c) takes leaf candidate s' of level f + 1; what a leaf's total and validity owe to the path above
                // this frame is computed once, then every surviving child b of this pass adds its own pair entry:
                //   total(b, b') = (total(b) + S[f + 1][b']) + (sum_q P[q -> (f + 1, b')] + P[(f, b) -> (f + 1, b')])   (tree.py:38-41)
                // in the reference's order (the child is the deepest ancestor, so its entry comes last).
                if (ab) {
                    PMX_COUNT(0, 1);
                    const int f1 = f + 1, k1 = rl(w.hk, f1) & 255, ks1 = rl(w.hks, f1);
                    tch[lane] = t; // the children's totals, read back per child by every slot
                    const int ebv1 = w.matB + ks1;
                    const bool on1 = s < k1;
                    const uint32_t bo1 = on1 ? lane_off : (uint32_t)c * 4u;
                    const float self1 = *reinterpret_cast<const float *>(Sb + (((uint32_t)ks1 << PSH) + bo1));
                    bool base_valid = on1;
                    double base_sum = 0.0;
                    for_rows(
                        nm, [&](int q) { return *reinterpret_cast<const float *>(Pb + (((uint32_t)rl(ebv1, q)) << PSH) + bo1); },
                        [&](float v) { // added in order
                            base_valid = base_valid & (v > 0.f);
                            base_sum += (double)v;
                        });
                    // entry((f, b) -> (f + 1, b')) = rowbase[f] + b * k1 + b'
                    const uint32_t row_f = (uint32_t)rl(w.hrow, f), nd_f = w.ksumtot - (uint32_t)ks1; // entry((f, b) -> (f + 1, b')) = rowbase[f] + b * nd_f + b'
                    lds_sync();
                    unsigned long long left = ab;
                    while (left) {
                        const int sb = (__ffsll(left) - 1) / G;
                        left &= ~(GM << (sb * G));
                        const uint64_t cm = (vb >> (sb * G)) & GM;
                        const double tb = tch[sb * G + c];
                        const float pfb = *reinterpret_cast<const float *>(Pb + ((row_f + (uint32_t)rl(bvec, sb * G) * nd_f) << PSH) + bo1);
                        const bool v1 = base_valid && pfb > 0.f && ((cm >> c) & 1ull);
                        const double t1 = (tb + (double)self1) + (base_sum + (double)pfb);
                        const bool any1 = __ballot(v1) != 0;
                        if (v1 && t1 > w.best) w.best = t1;                                             // leaves (graph_match.py:105-108)
                        if ((!any1 || nm < 3) && ((cm >> c) & 1ull) && tb > w.best) w.best = tb;        // the child's skip leaf (tree.py:98-101)
                        const int r1 = 1 + (any1 ? 1 : 0);
                        mx = mx > r1 ? mx : r1;
                        ++w.frames;
                        PMX_COUNT(1, 1);
                    }
                    w.passes += 1;
                }
                if (vb) mx = mx > 1 ? mx : 1; // (children dropped by the bound test return at least 1)
                nb += SLOTS;
                cb = cb_rest;
                flags |= kFused;
            } else if (ab) {
                bool keep = true; // the walker descends itself
                if (export_mode && nl - (f + 1) >= (int)p.min_levels) {
                    // Over budget: hand the surviving children of this pass to the task queue - one reservation, one record
                    // per slot. Children with >= 5 matches count as "returned >= 1" (see above). Below that the frame needs
                    // to know whether a child reaches 5 matches (tree.py:98), which probe() answers: only the first
                    // surviving child is handed over then, and this frame's max_num_matches is raised to 5 - nm if
                    // the child can get there (what it returns beyond that changes no decision anywhere).
                    const bool deep = nm >= 4;
                    const int first_ss = (__ffsll(ab) - 1) / G;
                    bool slot_alive = ((ab >> (s * G)) & GM) != 0;
                    if (!deep) slot_alive = slot_alive && s == first_ss;
                    const unsigned long long heads = __ballot(slot_alive && c == 0);
                    const uint32_t n = (uint32_t)__popcll(heads);
                    // all subtrees of a ligand go to one shard, and the task wavefronts of one XCD drain one group of
                    // shards (task_kernel): the walkers that share a ligand's tables run side by side under one L2
                    const uint32_t sh = (PMX_WFLAGS(p) & 256) ? ((wave_id + (uint32_t)(w.passes >> 4)) & (kShards - 1)) : ((rec16 * 2654435761u) >> 26);
                    static_assert(kShards == 64, "shard hash");
                    // one atomic add reserves the records (no retry loop: the walkers of one ligand export to one shard at
                    // the same time); a reservation that crosses the end of the shard fills its part below the end
                    // with empty subtrees of this ligand (no conformer: prepare_walk drops them)
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&p.ctl->q_res[sh], n);
                    base = (uint32_t)uni((int)base);
                    if (base + n > p.qcap) {
                        for (uint32_t i = base + (uint32_t)lane; i < p.qcap; i += 64u) {
                            uint32_t *nr = reinterpret_cast<uint32_t *>(p.queue + ((size_t)sh * p.qcap + i) * task_rec_bytes<G>());
                            for (uint32_t wd = 0; wd < task_rec_bytes<G>() / 4; ++wd) nr[wd] = 0u;
                            nr[0] = rec16;
                            nr[1] = (uint32_t)(f + 1) | (5u << 8); // f0, nm
                        }
                        base = 0xffffffffu;
                    }
                    if (base != 0xffffffffu) {
                        if (lane < nm) pathbuf[lane] = (uint16_t)(((w.matKA >> 16) & 255) | (((w.matKA >> 8) & 255) << 8));
                        lds_sync();
                        if (slot_alive) {
                            const uint32_t rank = (uint32_t)__popcll(heads & ((1ull << (s * G)) - 1ull));
                            unsigned char *tr = p.queue + ((size_t)sh * p.qcap + base + rank) * task_rec_bytes<G>();
                            TaskRec *th = reinterpret_cast<TaskRec *>(tr);
                            if (c == 0) {
                                th->rec16 = rec16;
                                th->f0 = (uint8_t)(f + 1);
                                th->nm = (uint8_t)(nm + 1);
                                th->pad = 0;
                                th->mask = (vb >> (s * G)) & GM;
                            }
                            const uint32_t mine = (uint32_t)f | ((uint32_t)bvec << 8); // this slot's own match, entry nm
                            for (int wd = c; wd < PMX_MAX_LEVELS / 2; wd += G) { // two path entries per 32-bit word
                                const int q0 = 2 * wd, q1 = 2 * wd + 1;
                                const uint32_t e0 = q0 < nm ? pathbuf[q0] : (q0 == nm ? mine : 0u);
                                const uint32_t e1 = q1 < nm ? pathbuf[q1] : (q1 == nm ? mine : 0u);
                                reinterpret_cast<uint32_t *>(th->path)[wd] = e0 | (e1 << 16);
                            }
                            reinterpret_cast<double *>(tr + sizeof(TaskRec))[c] = t;
                        }
                        if (lane == 0) stat->overflow += n; // (records written to the queue)
                        if (deep) {
                            mx = mx > 1 ? mx : 1; // children given away (or dropped) return at least 1
                            nb += SLOTS;
                            rem = 0xffffffffu;
                            cb = cb_rest;
                        } else {
                            probe_slot = first_ss;
                        }
                        keep = false;
                    } else if (lane == 0) {
                        p.ctl->qflag = 1; // shard full: walk the subtree here
                    }
                }
                if (keep) {
                    // descend into the first surviving child (tree.py:94-97)
                    int ss;
                    bool go = true;           // (false: the chosen child fails the path-aware bound test and is dropped)
                    unsigned child_path = 0u; // kPath if the child's path sums are in the wave's buffer
                    if (ordered) {
                      // A child that fails the path-aware test is dropped and the next survivor of the same pass is tried at once: nothing the
                      // next trip round the loop would work out again (the pass from the cache or the tables, its bound test) has changed.
                      const bool path_test = cand_bounds<G>() && (flags & kPath) && nl - f >= kPathMinLevels && !(PMX_WFLAGS(p) & 1024);
                      for (;;) {
                        const bool alive = (ab >> lane) & 1ull;
                        const float key = alive ? fmaxf((float)(t + rbound), 0.f) : -1.f; // (a NaN total orders as 0)
                        const float top = wave_max_f32(key);
                        ss = (__ffsll(__ballot(alive && key == top)) - 1) / G;
                        if (path_test) {
                            // the child with the largest W bound, against the bound its actual path gives (path_bound())
                            if (s == ss) tch[c] = t;
                            lds_sync();
#ifdef PMX_WALK_TICKS // instrumented builds: s_memtime ticks inside path_bound() and probe() in WaveStats::dbg[0], [1]
                            const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
                            go = path_bound<G>(w, p, pa, ub, tch, pool, f, nm, rl(bvec, ss * G), (vb >> (ss * G)) & GM);
#ifdef PMX_WALK_TICKS
                            if (lane == 0) stat->dbg[0] += __builtin_amdgcn_s_memtime() - tk0;
#endif
                            child_path = kPath;
                            ++w.npath;
                            if (!go) ++w.ndrop;
                        }
                        if (vb != ab || !go) mx = mx > 1 ? mx : 1; // (an existing child - visited, dropped or probed - returns at least 1)
                        rem &= ~(1u << ss);
                        if (!(ab & ~(GM << (ss * G))) && !pending) { // no other survivor: the window ends with this child
                            nb += SLOTS;
                            rem = 0xffffffffu;
                        }
                        if (!go && nm < 4 && mx < 5 - nm) { // the frame still has to know whether the dropped child reaches 5 matches
                            probe_slot = ss;
                            probe_rem_done = true;
                        }
                        if (go || probe_slot >= 0 || !(ab & ~(GM << (ss * G)))) break;
                        ab &= ~(GM << (ss * G)); // (the dropped child leaves the survivors; the trip this saves counts as a pass)
                        ++w.passes;
                      }
                    } else {
                        ss = (__ffsll(ab) - 1) / G;
                        const unsigned long long before = ss == 0 ? 0ull : (vb & ((1ull << (ss * G)) - 1ull));
                        if (before) mx = mx > 1 ? mx : 1; // existing children dropped by the bound test return at least 1
                        if constexpr (G >= 32) {
                            const bool shallow_w = nm < 4 && bound_from != 255 && !(PMX_WFLAGS(p) & 4096) && !(PMX_WFLAGS(p) & 262144);
                            if ((bounded || shallow_w) && !(PMX_WFLAGS(p) & 131072)) { // the path-aware test of these shapes: children that passed the level bound, and children with fewer than 5 matches
                                go = path_bound_wide<G>(w, t, s == ss, pool, f, nm, rl(bvec, ss * G), (vb >> (ss * G)) & GM);
                                ++w.npath;
                                if (!go) {
                                    ++w.ndrop;
                                    const int bdrop = rl(bvec, ss * G);
                                    if (nm < 4 && mx < 5 - nm) { // the frame still has to know whether the dropped child reaches 5 matches (tree.py:98)
                                        probe_slot = ss;         // (the probe's own bookkeeping moves nb / cb past the child)
                                    } else {
                                        mx = mx > 1 ? mx : 1;
                                        nb = bdrop + 1;
                                        cb &= ~((2ull << bdrop) - 1ull);
                                    }
                                }
                            }
                        }
                    }
                    if (go) {
                    const int bsel = rl(bvec, ss * G);
                    if (!ordered) nb = bsel + 1;
                    if (filt) { // what is left of the frame's candidates (the slots below ss were dropped)
                        cb &= ~((2ull << bsel) - 1ull);
                        if (lane == 0) cbl[f] = cb;
                        nb = cb ? 0 : 255;
                    }
                    const uint64_t cmask = (vb >> (ss * G)) & GM;
                    if (s == ss) tot[(nm + 1) * G + c] = t;
                    // this frame's state, then the child's: lane f + 1
                    w.stC = wl(wl(w.stC, f, nb | (mx << 8) | ((int)flags << 16) | (nm << 24)), f + 1, ((int)(kMatched | child_path) << 16) | ((nm + 1) << 24));
                    w.stA = wl(w.stA, f + 1, (int)(uint32_t)cmask);
                    if (G > 32) w.stB = wl(w.stB, f + 1, (int)(uint32_t)(cmask >> 32));
                    if (ORD) w.stB = wl(wl(w.stB, f, (int)rem), f + 1, -1);
                    w.matB = wl(w.matB, nm, match_base(w, f, bsel));
                    w.matKA = wl(w.matKA, nm, (bsel << 8) | (f << 16));
                    ++f;
                    ++w.frames;
                    PMX_COUNT(5, 1);
                    if (totals_in_lds<G>()) lds_sync(); // the child's total is read by all slots
                    else wave_sync();
                    continue;
                    }
                }
            } else { // every existing child of this pass was dropped (or none existed)
                if (vb) mx = mx > 1 ? mx : 1;
                nb += SLOTS;
                rem = 0xffffffffu;
                cb = cb_rest;
            }
            if (probe_slot >= 0) { // (handed over, or dropped by the bound test: either way the walker does not go there)
                uint32_t pp = 0;
                const int bp_ = rl(bvec, probe_slot * G);
#ifdef PMX_WALK_TICKS
                const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
#endif
                const bool reach = probe<G>(w, f, nm, bp_, (vb >> (probe_slot * G)) & GM, pp);
#ifdef PMX_WALK_TICKS
                if (lane == 0) stat->dbg[1] += __builtin_amdgcn_s_memtime() - tk1;
#endif
                if (lane == 0) {
                    stat->pad[0] += pp;
                    stat->pad[1] += 1;
                    stat->passes += pp;
                }
                mx = mx > 1 ? mx : 1;
                if (reach) mx = mx > 5 - nm ? mx : 5 - nm;
                if (ordered) {
                    if (!probe_rem_done) rem &= ~(1u << probe_slot);
                } else {
                    nb = bp_ + 1;
                }
                cb &= ~((2ull << bp_) - 1ull);
            }
            if (filt) {
                if (lane == 0) cbl[f] = cb;
                nb = cb ? 0 : 255;
            }
            if (nb < kf) { // more candidates: another pass
                w.stC = wl(w.stC, f, nb | (mx << 8) | ((int)flags << 16) | (nm << 24));
                if (ORD) w.stB = wl(w.stB, f, (int)rem);
                continue;
            }
        }
        // -------------------------------------------------------------------- the candidates of this frame are done
        if (leaf_level) {
            mx = (flags & kAny) ? 1 : 0;
            if (!(flags & kAny) || nm + mx < 5) { // skip leaf (tree.py:98-101, :42-43): this node's totals
                const double tparent = tot[nm * G + c];
                if (((mask >> c) & 1ull) && tparent > w.best) w.best = tparent;
            }
        }
        if (leaf_level || (flags & kFused)) {
            // publish improved maxima to the other slots (the bound test reads them)
            const bool up = w.best > w.flushed;
            if (__ballot(up)) {
                if (up) {
                    atomicMax(&pool[c], (unsigned long long)__double_as_longlong(w.best));
                    w.flushed = w.best;
                }
            }
        }
        if (!leaf_level && !(flags & kSkipped) && (!(flags & kAny) || nm + mx < 5)) { // skip child (tree.py:98-101)
            flags |= kSkipped;
            w.stC = wl(wl(w.stC, f, nb | (mx << 8) | ((int)flags << 16) | (nm << 24)), f + 1, ((int)(flags & kPath) << 16) | (nm << 24)); // (same matches: same path sums)
            w.stA = wl(w.stA, f + 1, (int)(uint32_t)mask);
            if (G > 32) w.stB = wl(w.stB, f + 1, (int)(uint32_t)(mask >> 32));
            if (ORD) w.stB = wl(w.stB, f + 1, -1);
            ++f;
            ++w.frames;
            continue;
        }
        // return max_num_matches + matched (tree.py:102) to the parent frame - and straight through every ancestor that has
        // nothing left to do: its candidates are done and it needs no skip child (it was entered for one of its children, so a
        // child existed: the skip child is due only while num_matches + max_num_matches < 5, tree.py:98). A third of the
        // walker's iterations used to be such returns, each a full trip round the loop.
        ret = mx + ((flags & kMatched) ? 1 : 0);
        bool out = false;
        for (;;) {
            --f;
            if (f < f0) {
                out = true;
                break;
            }
            int pc = rl(w.stC, f);
            int pmx = (pc >> 8) & 255;
            if (ret > pmx) {
                pmx = ret;
                pc = (pc & ~0xff00) | (ret << 8);
                w.stC = wl(w.stC, f, pc);
            }
            if ((pc & 255) < (rl(w.hk, f) & 255)) break;                                        // candidates left
            const int pfl = (pc >> 16) & 255, pnm = (pc >> 24) & 255;
            if (!(pfl & (int)kSkipped) && pnm + pmx < 5) break;                                  // its skip child is due
            ret = pmx + ((pfl & (int)kMatched) ? 1 : 0);
        }
        if (out) break;
    }
    flush_dbg();
    return ret;
}
#undef PMX_COUNT


// ------------------------------------------------------------------------------------------ table phase
// Round 6: instruction injection (inject_valu / inject_salu above) showed that every instruction of a table item costs its full issue price -
// 16 / 32 v_nops per batch of two items: tables alone 43.5 -> 45.3 / 47.7 ms - so the item is on a diet: the function index of a symmetric model
// (every model the reference can make) without the general form and its 64-bit multiply-add, the cell number kept as the integer it is, 24-bit
// multiplies where an entry is decoded. PMX_ITEM_DIET=0 builds the round-5 item for A/B runs.
#ifndef PMX_ITEM_DIET
#define PMX_ITEM_DIET 1
#endif
// TRI: the caller has established that the table is triangular (FnTable::tri, wave-uniform)
template <bool TRI>
__device__ __forceinline__ uint32_t fn_index_t(const FnTable &F, uint32_t sidu, uint32_t sidv) {
    if (TRI) {
        const uint32_t hi = max(sidu, sidv), lo = min(sidu, sidv);
        return ((__umul24(hi, hi) + hi) >> 1) + lo; // (subset ids are 16 bits)
    }
    return __umul24(sidu, F.NS) + sidv;
}
__device__ __forceinline__ uint32_t fn_index(const FnTable &F, uint32_t sidu, uint32_t sidv) {
    // (both forms and a bit select on the wave-uniform `tri`: a branch here is a branch per table item)
    const uint32_t hi = max(sidu, sidv), lo = min(sidu, sidv);
    const uint32_t t = (__umul24(hi, hi + 1u) >> 1) + lo, f = __umul24(sidu, F.NS) + sidv; // (subset ids are 16 bits)
    const uint32_t m = 0u - F.tri;
    return (t & m) | (f & ~m);
}

// One (ligand node, ligand node) item term by term, in the float32 operations of the reference (match_utils.py:50-69 and
// :108-120; same order as oracle/pmx_oracle.c node_pair_term): weights_sum by float32 additions, z = (d - mean) / std with
// an IEEE division, exp(-0.5 z^2) to float32 accuracy, the likelihood added up in the order of itertools.product, then
// likelihood * (1 / weights_sum) * (weights_sum / num_match). A subset pair whose weights sum to 0 gives NaN like the
// reference (x * inf * 0). np = the terms within 2 sigma (:56-60). A, B non-empty.
__device__ __forceinline__ float exact_value(const ScreenParams &p, uint32_t sidu, uint32_t sidv, float d, int &np, int &mn) {
    const uint8_t *A = p.sub_nodes + p.sub_off[sidu], *B = p.sub_nodes + p.sub_off[sidv];
    const int nA = (int)(p.sub_off[sidu + 1] - p.sub_off[sidu]), nB = (int)(p.sub_off[sidv + 1] - p.sub_off[sidv]);
    float weights_sum = 0.f;
    for (int ia = 0; ia < nA; ++ia) {
        const float wa = p.W.w[p.M.node_type[A[ia]]];
        for (int ib = 0; ib < nB; ++ib) weights_sum = weights_sum + wa * p.W.w[p.M.node_type[B[ib]]];
    }
    mn = nA * nB;
    const float normalize_coeff = 1.0f / weights_sum, score_coeff = weights_sum / (float)mn;
    float likelihood = 0.f;
    np = 0;
    for (int ia = 0; ia < nA; ++ia) {
        const int m = A[ia];
        const float wa = p.W.w[p.M.node_type[m]];
        for (int ib = 0; ib < nB; ++ib) {
            const int n = B[ib];
            const float4 e = p.M.edge[m * p.M.Nm + n]; // {mean, s, T, std}
            const float t = d - e.x, z = t / e.w;
            np += fabsf(t) <= e.z ? 1 : 0; // == abs(z) < 2 (pmx_device.h)
            const float wos = (wa * p.W.w[p.M.node_type[n]]) / e.w;
            likelihood = likelihood + wos * expf(-0.5f * (z * z));
        }
    }
    return likelihood * normalize_coeff * score_coeff;
}
// The terms of the subset pair within 2 sigma at distance d against half of their number (match_utils.py:56-61): does the item fail?
__device__ __forceinline__ bool majority_fails(const ScreenParams &p, uint32_t sidu, uint32_t sidv, float d) {
    const uint8_t *A = p.sub_nodes + p.sub_off[sidu], *B = p.sub_nodes + p.sub_off[sidv];
    const int nA = (int)(p.sub_off[sidu + 1] - p.sub_off[sidu]), nB = (int)(p.sub_off[sidv + 1] - p.sub_off[sidv]);
    int np = 0;
    for (int ia = 0; ia < nA; ++ia)
        for (int ib = 0; ib < nB; ++ib) {
            const float4 e = p.M.edge[(int)A[ia] * p.M.Nm + (int)B[ib]];
            np += fabsf(d - e.x) <= e.z ? 1 : 0;
        }
    return 2 * np < nA * nB;
}

// One (ligand node, ligand node) item of match_utils.py:26-69 for the subset pair (sidu, sidv) at distance d: the tabulated
// sum (already divided by |A||B|) and whether the item fails the majority test of :56-61. SELF: the item belongs to a self
// entry, where a cell flagged as rough (FnCell) is evaluated term by term. EXACT (PMX_TREE_FLAGS & 8): every item is.
template <bool EXACT, bool SELF>
__device__ __forceinline__ void item(const ScreenParams &p, uint32_t sidu, uint32_t sidv, float d, float &acc, int &fails,
                                     uint32_t &n_exact, uint32_t &n_exactv) {
    if (!EXACT) {
        const float x = d * p.F.inv_h; // exact: inv_h is a power of two
        const int ci = min((int)x, (int)p.F.ncell - 1);
        const float t = fminf(x - (float)ci, 1.0f);
        const float4 *cell = reinterpret_cast<const float4 *>(p.F.cells) + (fn_index(p.F, sidu, sidv) * p.F.ncell + (uint32_t)ci);
        const float4 a = cell[0], b = cell[p.F.plane16];
        float v = __builtin_fmaf(t, b.y, b.x);
        v = __builtin_fmaf(t, v, a.w);
        v = __builtin_fmaf(t, v, a.z);
        v = __builtin_fmaf(t, v, a.y);
        v = __builtin_fmaf(t, v, a.x);
        if (SELF) {
            if (__builtin_expect((__float_as_uint(b.y) & 1u) != 0u && sidu != 0u && sidv != 0u, 0)) {
                int np, mn;
                v = exact_value(p, sidu, sidv, d, np, mn);
                ++n_exactv;
            }
            acc = acc + v;
            return; // (no majority test on self entries)
        }
        acc = acc + v;
        if (__builtin_expect(b.z != b.z, 0)) { // the pass set is not one interval inside this cell: count the terms
            fails += majority_fails(p, sidu, sidv, d) ? 1 : 0;
            ++n_exact;
        } else {
            fails += (d >= b.z && d <= b.w) ? 0 : 1;
        }
        return;
    }
    // debug / validation (flags & 8): every item term by term
    if (sidu == 0u || sidv == 0u) return; // (0 = the empty subset)
    int np, mn;
    acc = acc + exact_value(p, sidu, sidv, d, np, mn);
    fails += 2 * np < mn ? 1 : 0;
}

// The same item in two steps, so that the loads of several items are in flight together: address + loads, then value + test.
// (What an item holds while its cell is on the way is what limits how many can be: the cell, the distance, the two subset ids
// in one word; the position inside the cell is worked out again from the distance.)
struct ItemLoad {
    float4 a, b;
    float d, cell; // the distance and the number of its cell (as a float: the position inside the cell is d / h - cell)
    uint32_t sids; // sidu | sidv << 16
};
// the cell of a distance: min(floor(d / h), ncell - 1)
__device__ __forceinline__ float cell_of(const ScreenParams &p, float d) {
    return (float)min((int)(d * p.F.inv_h), (int)p.F.ncell - 1); // (d * inv_h is exact: inv_h is a power of two)
}
// (the diet's form: the function index by the caller's knowledge of the table's shape, the cell number computed once)
template <bool TRI>
__device__ __forceinline__ ItemLoad item_load_t(const ScreenParams &p, uint32_t sidu, uint32_t sidv, float d) {
    ItemLoad L;
    L.d = d;
    const int ci = min((int)(d * p.F.inv_h), (int)p.F.ncell - 1);
    L.cell = (float)ci;
    L.sids = sidu | (sidv << 16);
    const uint32_t off = (__umul24(fn_index_t<TRI>(p.F, sidu, sidv), p.F.ncell) + (uint32_t)ci) << 4;
    const unsigned char *pa = reinterpret_cast<const unsigned char *>(p.F.cells);
    const unsigned char *pb = pa + (size_t)p.F.plane16 * 16u;
    L.a = *reinterpret_cast<const float4 *>(pa + off);
    L.b = *reinterpret_cast<const float4 *>(pb + off);
    return L;
}
__device__ __forceinline__ ItemLoad item_load(const ScreenParams &p, uint32_t sidu, uint32_t sidv, float d, float cell) {
    ItemLoad L;
    L.d = d;
    L.cell = cell;
    L.sids = sidu | (sidv << 16);
    // (functions x cells < 2^27 - the table is addressed with 32 bits - and a function has hundreds of cells: 24-bit factors)
    const uint32_t off = (__umul24(fn_index(p.F, sidu, sidv), p.F.ncell) + (uint32_t)(int)cell) << 4;
    const unsigned char *pa = reinterpret_cast<const unsigned char *>(p.F.cells);
    const unsigned char *pb = pa + (size_t)p.F.plane16 * 16u; // (both planes: uniform base + 32-bit lane offset)
    L.a = *reinterpret_cast<const float4 *>(pa + off);
    L.b = *reinterpret_cast<const float4 *>(pb + off);
    return L;
}
// TAILS: the call's type weights differ by more than PMX_TAILS_RATIO (pmx_api.hip) - a pair item honours the rough-cell flag like a
// self item. With the reference's default weights (8 : 1 at most) an entry that counts is made of items near their functions' peaks, next
// to which the error of a tail value is below float32 rounding; with `--cation 100 --hydrophobic 0.1` (screening.py:54-62) an entry can
// be a handful of passing Hydrophobic items beside one failing Cation x Cation item five sigma out whose function is 10^6 times theirs -
// and the tail IS the entry (tests/test_gpu_pair_tails.py).
template <bool TAILS>
__device__ __forceinline__ void item_finish(const ScreenParams &p, const ItemLoad &L, float &acc, int &fails, uint32_t &n_exact, uint32_t &n_exactv) {
    if (TAILS) {
        const uint32_t su = L.sids & 0xffffu, sv = L.sids >> 16;
        if (__builtin_expect((__float_as_uint(L.b.y) & 1u) != 0u && su != 0u && sv != 0u, 0)) {
            int np, mn;
            acc = acc + exact_value(p, su, sv, L.d, np, mn);
            fails += 2 * np < mn ? 1 : 0; // match_utils.py:56-61
            ++n_exactv;
            return;
        }
    }
    const float t = fminf(__builtin_fmaf(L.d, p.F.inv_h, -L.cell), 1.0f); // (= d / h - cell exactly: the product is exact)
    float v = __builtin_fmaf(t, L.b.y, L.b.x);
    v = __builtin_fmaf(t, v, L.a.w);
    v = __builtin_fmaf(t, v, L.a.z);
    v = __builtin_fmaf(t, v, L.a.y);
    v = __builtin_fmaf(t, v, L.a.x);
    acc = acc + v;
#if PMX_ITEM_DIET
    // lo <= d <= hi as "d is the median of (d, lo, hi)" (every window has lo <= hi; pmx_api.hip fn_windows): one compare, no mask arithmetic. A cell whose
    // pass set is not one interval (lo = NaN; 0.8 items per ligand) is put right behind one wave-wide test instead of an exec-mask detour per item.
    const bool fail = __builtin_amdgcn_fmed3f(L.d, L.b.z, L.b.w) != L.d;
    fails += fail ? 1 : 0;
    if (__builtin_expect(__ballot(L.b.z != L.b.z) != 0ull, 0)) {
        if (L.b.z != L.b.z) { // count the terms
            fails += (majority_fails(p, L.sids & 0xffffu, L.sids >> 16, L.d) ? 1 : 0) - (fail ? 1 : 0);
            ++n_exact;
        }
    }
#else
    if (__builtin_expect(L.b.z != L.b.z, 0)) { // the pass set is not one interval inside this cell: count the terms
        fails += majority_fails(p, L.sids & 0xffffu, L.sids >> 16, L.d) ? 1 : 0;
        ++n_exact;
    } else {
        fails += (L.d >= L.b.z && L.d <= L.b.w) ? 0 : 1;
    }
#endif
}

#ifndef PMX_ITEM_BATCH
#define PMX_ITEM_BATCH 2 // items whose loads are in flight together (4 costs 30 spilled registers at 80)
#endif
struct LevelInfo {
    int nl;
    uint32_t ksumtot, T;
};

// Cluster candidates and tree levels (graph_match.py:124-137, :87-88) of the record, into the wave's LDS: clusters arrive
// sorted by priority_fn; a cluster is kept if some model cluster shares a type with it; at most 20 are kept. Then the
// node-candidate table nc[level][candidate][node] = node subset of the model cluster compatible with the ligand node
// (graph_match.py:145-155) and the counts L of ligand nodes with a non-empty subset (graph_match.py:164-171).
template <int G>
__device__ __forceinline__ LevelInfo scan_ligand(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const Record &r) {
    const int lane = lane_id();
    uint8_t *tm = lds + kOffTm, *lstart = lds + kOffStart, *lend = lds + kOffEnd, *lk = lds + kOffK;
    uint16_t *ksum = reinterpret_cast<uint16_t *>(lds + kOffKsum), *ncoff = reinterpret_cast<uint16_t *>(lds + kOffNcoff);
    uint32_t *rowbase = reinterpret_cast<uint32_t *>(lds + kOffRow);
    uint32_t *scal = reinterpret_cast<uint32_t *>(lds + kOffBits + 8 * PMX_MAX_LEVELS); // ksumtot, T
    uint8_t *cand = lds + ws.off_cand, *lcnt = lds + ws.off_lcnt;
    uint16_t *nc = reinterpret_cast<uint16_t *>(lds + ws.off_nc);
    if (lane < r.n) tm[lane] = r.typemask[lane];
    wave_sync();
    int cs = 0, ce = 0;
    uint64_t cb0 = 0, cb1 = 0; // candidate clusters of the ligand cluster (PMX_MAX_MODEL_CLUSTERS bits)
    if (lane < r.ncl) {
        cs = lane ? r.cluster_end[lane - 1] : 0;
        ce = r.cluster_end[lane];
        unsigned lm = 0;
        for (int u = cs; u < ce; ++u) lm |= tm[u];
        cb0 = p.M.tclus[2u * (lm & 127u)];
        cb1 = p.M.tclus[2u * (lm & 127u) + 1u];
    }
    const bool has = (cb0 | cb1) != 0ull;
    const int kc = (int)__popcll(cb0) + (int)__popcll(cb1);
    const unsigned long long bal = __ballot(has);
    const int lev = __popcll(bal & ((1ull << lane) - 1ull));
    const int nl = min((int)__popcll(bal), PMX_MAX_LEVELS);
    // (a level's candidates are a 64-bit set in the walker: a ligand cluster with more - only a model of more than 64 clusters has
    // that many of one type - makes the ligand unsupported)
    if (__ballot(has && lev < PMX_MAX_LEVELS && kc > PMX_MAX_LEVEL_CANDIDATES) != 0ull) {
        LevelInfo bad;
        bad.nl = -1, bad.ksumtot = 0, bad.T = 0;
        return bad;
    }
    if (has && lev < PMX_MAX_LEVELS) {
        lstart[lev] = (uint8_t)cs;
        lend[lev] = (uint8_t)ce;
        lk[lev] = (uint8_t)kc;
        int q = 0;
        for (uint64_t x = cb0; x; x &= x - 1, ++q) cand[lev * ws.kp + q] = (uint8_t)(__ffsll((unsigned long long)x) - 1);
        for (uint64_t x = cb1; x; x &= x - 1, ++q) cand[lev * ws.kp + q] = (uint8_t)(64 + __ffsll((unsigned long long)x) - 1);
    }
    wave_sync();
    if (lane == 0) {
        uint32_t ks = 0, no = 0;
        for (int l = 0; l < nl; ++l) {
            ksum[l] = (uint16_t)ks;
            ncoff[l] = (uint16_t)no;
            ks += lk[l];
            no += (uint32_t)lk[l] * (uint32_t)(lend[l] - lstart[l]);
        }
        ksum[nl] = (uint16_t)ks;
        uint32_t rb = 0, run = 0;
        for (int l = 0; l < nl; ++l) {
            rowbase[l] = rb;
            run += lk[l];
            rb += (uint32_t)lk[l] * (ks - run);
        }
        scal[0] = ks;
        scal[1] = rb;
    }
    wave_sync();
    for (int l = 0; l < nl; ++l) {
        const int s0 = uni(lstart[l]), n = uni(lend[l]) - s0, k = uni(lk[l]), base = uni(ncoff[l]);
        const float inv_n = 1.0f / (float)n;
        for (int idx = lane; idx < k * n; idx += 64) {
            const int q = (int)(((float)idx + 0.5f) * inv_n), u = idx - q * n;
            nc[base + idx] = p.sidtab[(uint32_t)cand[l * ws.kp + q] * 128u + tm[s0 + u]];
        }
    }
    wave_sync();
    for (int l = 0; l < nl; ++l) {
        const int n = uni(lend[l]) - uni(lstart[l]), k = uni(lk[l]), base = uni(ncoff[l]);
        if (lane < k) {
            int cnt = 0;
            for (int u = 0; u < n; ++u) cnt += nc[base + lane * n + u] != 0 ? 1 : 0;
            lcnt[l * ws.kp + lane] = (uint8_t)cnt;
        }
    }
    wave_sync();
    LevelInfo L;
    L.nl = nl;
    L.ksumtot = (uint32_t)uni((int)scal[0]);
    L.T = (uint32_t)uni((int)scal[1]);
    return L;
}

struct Pos3 {
    float x, y, z;
};

// LigandNodeCluster.center / .size for one conformer (ligand.py:458-473).
// (a pointer into device memory, said so: a generic pointer costs flat loads, which also wait on the LDS counter, and 64-bit
// address arithmetic per load)
typedef const __attribute__((address_space(1))) float *GlobalFloats;
__device__ __forceinline__ void center_size(GlobalFloats xyz, int C, int start, int end, int cc, Pos3 &center, float &size) {
    // (four nodes' coordinates per trip, added in node order as before: a load per coordinate, each waited for, made this two memory
    // round trips per node of the cluster)
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int u0 = start; u0 < end; u0 += 4) {
        float x[4], y[4], z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = (uint32_t)(min(u0 + k, end - 1) * 3 * C + cc);
            x[k] = xyz[o], y[k] = xyz[o + C], z[k] = xyz[o + 2 * C];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = u0 + k < end;
            sx = in ? sx + x[k] : sx;
            sy = in ? sy + y[k] : sy;
            sz = in ? sz + z[k] : sz;
        }
    }
    const float cnt = (float)(end - start);
    center = Pos3{sx / cnt, sy / cnt, sz / cnt};
    float mx = 0.f;
    for (int u0 = start; u0 < end; u0 += 4) {
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = (uint32_t)(min(u0 + k, end - 1) * 3 * C + cc);
            r[k] = norm3f(xyz[o] - center.x, xyz[o + C] - center.y, xyz[o + 2 * C] - center.z);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = (u0 + k == start || (u0 + k < end && r[k] > mx)) ? r[k] : mx;
    }
    size = mx;
}

// The self / pair score tables of match_utils.py for the ligand whose levels are in LDS, into `rec`.
template <int G, bool EXACT, bool TAILS>
__device__ __forceinline__ void build_tables(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const Record &r, const LevelInfo &L,
                                             unsigned char *rec, uint32_t &n_items, uint32_t &n_exact, uint32_t &n_exactv, uint32_t &n_dead) {
    constexpr int SLOTS = 64 / G;
    constexpr uint64_t GM = group_mask<G>();
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    const int C = r.C, cc = c < C ? c : C - 1;
    const uint8_t *lstart = lds + kOffStart, *lend = lds + kOffEnd, *lk = lds + kOffK;
    const uint16_t *ksum = reinterpret_cast<const uint16_t *>(lds + kOffKsum), *ncoff = reinterpret_cast<const uint16_t *>(lds + kOffNcoff);
    const uint8_t *cand = lds + ws.off_cand, *lcnt = lds + ws.off_lcnt;
    const uint16_t *nc = reinterpret_cast<const uint16_t *>(lds + ws.off_nc);
    const uint32_t *rowbase_l = reinterpret_cast<const uint32_t *>(lds + kOffRow);
    GlobalFloats xyz = (GlobalFloats)uniptr(r.xyz);
    float *St = reinterpret_cast<float *>(rec + rec_s_off<G>());
    float *Pt = reinterpret_cast<float *>(rec + rec_p_off<G>(L.ksumtot));
    unsigned char *Vt = rec + rec_v_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    const int nl = L.nl, K = p.M.K;
    // Node distances of the cluster (pair) in work, once, in LDS: every table entry of the pair - k_i k_j of them - reads the
    // distances of the same node pairs, and a distance from coordinates is six loads, each of which occupies the L1 for four
    // cycles whether or not its lanes share an address. The table phase was bound by exactly that (rocprofv3: 0.86 L1 accesses
    // per cycle and CU, 67 % of its wave cycles waiting on memory). The walker's LDS (children cache, level maxima) is idle
    // in this phase and holds 82 node pairs at 8 lanes; larger pairs - and the 32 / 64-lane shapes, which keep nothing
    // there - compute from the coordinates as before. (The cell of the distance staged with it - the same for every entry
    // too - halves what fits and costs more than it saves: measured.)
    constexpr uint32_t kPfBytes = 2u * G * 4u; // (the cluster distance and size sum of the level pair, below, come first)
    // At 32 / 64 lanes the wave's buffer of path totals in global memory (idle until the walk) takes their place: one coalesced
    // load per item instead of six and the square root.
    constexpr bool kStageLds = totals_in_lds<G>();
    float *dl = kStageLds ? reinterpret_cast<float *>(lds + ws.off_tch + kPfBytes) : reinterpret_cast<float *>(p.totbuf + (size_t)blockIdx.x * kTotBufBytes);
    const int dcap = kStageLds ? (int)((ws.bytes - ws.off_tch - kPfBytes) / (uint32_t)(G * 4)) : (int)(kTotBufBytes / (uint32_t)(G * 4));
#ifdef PMX_TABLE_TICKS // instrumented builds: s_memtime ticks of the parts of this phase in WaveStats::dbg - [0] self tables [1] centres of a level pair [2] its node distances [3] prefilter and the rows of failing entries [4] items [5] chain lengths (build_bounds)
    unsigned long long tick_ = __builtin_amdgcn_s_memtime();
#define PMX_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) reinterpret_cast<WaveStats *>(lds + ws.off_stat)->dbg[i] += t_ - tick_; tick_ = t_; } while (0)
#else
#define PMX_TICK(i)
#endif
    auto node_distance = [&](int a0, int u, int b0, int v) {
        const uint32_t ou = (uint32_t)((a0 + u) * 3 * C + cc), ov = (uint32_t)((b0 + v) * 3 * C + cc);
        return norm3f(xyz[ou] - xyz[ov], xyz[ou + C] - xyz[ov + C], xyz[ou + 2 * C] - xyz[ov + 2 * C]);
    };
    auto stage_distances = [&](int a0, int na, int b0, int nb) { // dl[(u * nb + v) * G + c] = |x_(a0 + u) - x_(b0 + v)|
        lds_sync();                                              // (readers of the last pair's distances are done)
        const float inv_nb = 1.0f / (float)nb;
        for (int pr = s; pr < na * nb; pr += 2 * SLOTS) { // (two node pairs per trip: twelve coordinate loads in flight instead of six)
            const int pr2 = pr + SLOTS < na * nb ? pr + SLOTS : pr;
            const int u = (int)(((float)pr + 0.5f) * inv_nb), v = pr - u * nb;
            const int u2 = (int)(((float)pr2 + 0.5f) * inv_nb), v2 = pr2 - u2 * nb;
            const float d1 = node_distance(a0, u, b0, v), d2 = node_distance(a0, u2, b0, v2);
            dl[pr * G + c] = d1;
            dl[pr2 * G + c] = d2;
        }
        if (kStageLds) lds_sync();
        else wave_sync();
    };
    // Centre and size of every level's ligand cluster (ligand.py:458-473), once, with a slot per level: the pair loop below needs
    // them for every pair of levels and used to work them out again per pair (nl (nl - 1) / 2 + nl times instead of nl: 6 % of
    // the bench pass). They wait in the record's R / W regions, which build_bounds() fills only after this phase.
    constexpr bool kCentersStaged = cand_bounds<G>(); // (the W region exists; single-node clusters - the 32 / 64-lane stress model - gain nothing)
    float2 *cxy = reinterpret_cast<float2 *>(rec + rec_r_off<G>(L.ksumtot, L.T));
    float2 *czs = reinterpret_cast<float2 *>(rec + rec_w_off<G>(L.ksumtot, L.T, (uint32_t)L.nl));
    if (kCentersStaged) {
        for (int l = s; l < nl; l += SLOTS) {
            Pos3 ctr;
            float size;
            center_size(xyz, C, (int)lstart[l], (int)lend[l], cc, ctr, size);
            cxy[l * G + c] = make_float2(ctr.x, ctr.y);
            czs[l * G + c] = make_float2(ctr.z, size);
        }
        wave_sync();
    }
    for (int i = 0; i < nl; ++i) {
        const int si = uni(lstart[i]), ni = uni(lend[i]) - si, ki = uni(lk[i]), nci = uni(ncoff[i]), ksi = uni(ksum[i]);
        const uint32_t row_i = (uint32_t)uni((int)rowbase_l[i]), nd_i = L.ksumtot - (uint32_t)uni(ksum[i + 1]);
        // ---- self table S[i][a] (match_utils.py:77-122): node pairs u < v of the cluster
#ifndef PMX_SELF_STAGE_MIN
#define PMX_SELF_STAGE_MIN 4 // (a cluster of two or three nodes has one or three self items: a staging trip - 16 node pairs wide, a round trip through LDS - costs more instructions than computing them in place)
#endif
#ifndef PMX_CUT
#define PMX_CUT 0 // analysis builds (tools/build_variant.py + PMX_TREE_FLAGS=16384): 1 no self items, 2 no bounds pass, 4 no pair items - the instruction budget of a section is what its absence takes out of SQ_INSTS_*; scores are meaningless
#endif
        const bool self_staged = !(PMX_CUT & 1) && ni >= PMX_SELF_STAGE_MIN && ni * ni <= dcap;
        if (self_staged) stage_distances(si, ni, si, ni);
        for (int q0 = 0; q0 < ki; q0 += SLOTS) {
            const int q = q0 + s;
            const bool on = q < ki;
            const int row = nci + (on ? q : 0) * ni;
            float acc = 0.f;
            int fails = 0;
            for (int u = 0; u + 1 < ((PMX_CUT & 1) ? 0 : ni); ++u) {
                const uint32_t sidu = nc[row + u];
                for (int v = u + 1; v < ni; ++v) {
                    const float d = self_staged ? dl[(u * ni + v) * G + c] : node_distance(si, u, si, v);
                    item<EXACT, true>(p, sidu, nc[row + v], d, acc, fails, n_exact, n_exactv);
                    ++n_items;
                }
            }
            if (on) St[(size_t)(ksi + q) * G + c] = acc;
        }
        PMX_TICK(0);
        Pos3 ctr_i;
        float size_i;
        if (kCentersStaged) {
            const float2 a = cxy[i * G + c], b = czs[i * G + c];
            ctr_i = Pos3{a.x, a.y, b.x};
            size_i = b.y;
        } else {
            center_size(xyz, C, si, si + ni, cc, ctr_i, size_i);
        }
        for (int j = i + 1; j < nl; ++j) {
            const int sj = uni(lstart[j]), nj = uni(lend[j]) - sj, kj = uni(lk[j]), ncj = uni(ncoff[j]);
            Pos3 ctr_j;
            float size_j;
            if (kCentersStaged) {
                const float2 a = cxy[j * G + c], b = czs[j * G + c];
                ctr_j = Pos3{a.x, a.y, b.x};
                size_j = b.y;
            } else {
                center_size(xyz, C, sj, sj + nj, cc, ctr_j, size_j);
            }
            const float ldist = norm3f(ctr_i.x - ctr_j.x, ctr_i.y - ctr_j.y, ctr_i.z - ctr_j.z); // graph_match.py:240
            const float lsize = size_i + size_j;                                                  // :241
            const int E = ki * kj;
            const float inv_kj = 1.0f / (float)kj;
            const uint32_t off_j = (uint32_t)(uni(ksum[j]) - uni(ksum[i + 1])); // (j's candidates inside the run of (i, a)'s entries)
            // Which entries pass the cluster-distance prefilter (graph_match.py:263-268: an entry is computed if some conformer
            // passes) is settled first, 64 entries at a time with the lanes spread over *entries* - for a model of 30-40 clusters
            // most of the k_i k_j entries of a level pair fail, and walking them eight at a time was most of the table phase.
            // Failing entries get their -1 row and empty mask right there; the passing ones are listed and computed eight at a time.
            float *pf = reinterpret_cast<float *>(lds + ws.off_tch); // [G] cluster distance | [G] size sum, per conformer
            uint8_t *plist = lds + ws.off_task;                      // passing entries of the chunk (the root record is written later)
            if (s == 0) {
                pf[c] = ldist;
                pf[G + c] = lsize;
            }
            const bool staged = ni * nj <= dcap;
            const bool dead_test = staged && ni <= 64 && nj <= 64 && !(PMX_WFLAGS(p) & 65536u);
            PMX_TICK(1);
            if (staged) stage_distances(si, ni, sj, nj);
            else lds_sync();
            PMX_TICK(2);
            for (int eb = 0; eb < E; eb += 64) {
                unsigned long long pbal;
                {
                    const int e = eb + lane;
                    const bool in = e < E;
                    const int ee = in ? e : eb;
                    const int sa = (int)(((float)ee + 0.5f) * inv_kj), sb = ee - sa * kj;
                    const float2 mp = p.M.cpair[cand[i * ws.kp + sa] * K + cand[j * ws.kp + sb]];
                    bool pass = false;
                    {   // (eight conformers per trip - lanes past C hold copies of conformer C - 1, which an OR does not mind: the reads of a trip are two wide LDS loads)
                        constexpr int KP = G < 8 ? G : 8;
                        for (int k0 = 0; k0 < C; k0 += KP) {
#pragma unroll
                            for (int kk = 0; kk < KP; ++kk) pass = pass || !((fabsf(pf[k0 + kk] - mp.x) - pf[G + k0 + kk]) > mp.y);
                        }
                    }
                    pass = pass && in;
                    // Dead entries. An entry that passes the prefilter is still -1 for every conformer when more than half of
                    // its counted node pairs fail the 2-sigma majority test (match_utils.py:55-61, :71-74) - for a pocket of 20-40
                    // clusters that is every second entry and every second item. A node pair whose distance lies outside the
                    // hull of the windows of ALL model node pairs of the two clusters (DevModel::cwin, exact float ends) passes no
                    // term, whatever the node subsets: it certainly fails. Counting those - two compares on a staged distance,
                    // no function cell - gives a lower bound cf on an entry's fails per conformer, and 2 cf > L1 L2 for every
                    // conformer settles the entry: its row is -1 and its mask empty, exactly what the items would have given.
                    // (Lanes over entries like the prefilter: C x pairs trips per 64 entries against pairs x 64 / SLOTS item trips.)
                    if (dead_test && E >= (int)p.dead_min_entries) {
                        const float2 w = p.M.cwin[cand[i * ws.kp + sa] * K + cand[j * ws.kp + sb]];
                        unsigned long long mu = 0ull, mv = 0ull; // nodes with a non-empty subset under the candidate (graph_match.py:164-171)
                        if (pass) {
                            const int rowa = nci + sa * ni, rowb = ncj + sb * nj;
                            for (int u = 0; u < ni; ++u) mu |= (unsigned long long)(nc[rowa + u] != 0 ? 1 : 0) << u;
                            for (int v = 0; v < nj; ++v) mv |= (unsigned long long)(nc[rowb + v] != 0 ? 1 : 0) << v;
                        }
                        const int L1L2 = (int)__popcll(mu) * (int)__popcll(mv);
                        bool dead = pass && L1L2 > 0;
                        constexpr int KB = G < 8 ? G : 8; // conformers per trip: their distances of a node pair are one batch of loads
                        for (int k0 = 0; k0 < C; k0 += KB) {
                            if (__ballot(dead) == 0ull) break;
                            int cf[KB];
#pragma unroll
                            for (int kk = 0; kk < KB; ++kk) cf[kk] = 0;
                            for (int u = 0; u < ni; ++u) {
                                const int bu = (int)(mu >> u) & 1;
                                for (int v = 0; v < nj; ++v) {
                                    const int on_uv = bu & (int)(mv >> v);
                                    const float *dp = dl + (u * nj + v) * G + k0; // (lanes of a slot past C hold copies of conformer C - 1)
#pragma unroll
                                    for (int kk = 0; kk < KB; ++kk) {
                                        const float d = dp[kk];
                                        cf[kk] += (on_uv & ((d < w.x || d > w.y) ? 1 : 0));
                                    }
                                }
                            }
#pragma unroll
                            for (int kk = 0; kk < KB; ++kk) dead = dead && 2 * cf[kk] > L1L2;
                        }
                        n_dead += (uint32_t)__popcll(__ballot(dead));
                        pass = pass && !dead;
                    }
                    pbal = __ballot(pass);
                    if (in && !pass) {
                        float *row = Pt + (size_t)(row_i + (uint32_t)sa * nd_i + off_j + (uint32_t)sb) * G;
                        if (G >= 4) {
#pragma unroll
                            for (int g = 0; g < G; g += 4) *reinterpret_cast<float4 *>(row + g) = make_float4(-1.f, -1.f, -1.f, -1.f);
                        } else {
                            for (int g = 0; g < G; ++g) row[g] = -1.f;
                        }
                        unsigned char *ve = Vt + (size_t)(row_i + (uint32_t)sa * nd_i + off_j + (uint32_t)sb) * vmask_bytes<G>();
                        for (uint32_t g = 0; g < vmask_bytes<G>(); ++g) ve[g] = 0;
                    }
                    if (pass) {
                        const uint32_t lo32 = (uint32_t)pbal, hi32 = (uint32_t)(pbal >> 32);
                        plist[__builtin_amdgcn_mbcnt_hi(hi32, __builtin_amdgcn_mbcnt_lo(lo32, 0u))] = (uint8_t)lane;
                    }
                }
                lds_sync();
                PMX_TICK(3);
                const int npass = (int)__popcll(pbal);
                // what is written for a finished entry: match_utils.py:71-74 (-1 unless num_fails <= L1 * L2 / 2), the row and its V mask
                auto finish_entry = [&](int e, bool on, float acc, int fails) {
                    const int sa = (int)(((float)e + 0.5f) * inv_kj), sb = PMX_ITEM_DIET ? e - __mul24(sa, kj) : e - sa * kj;
                    const int L1 = lcnt[i * ws.kp + sa], L2 = lcnt[j * ws.kp + sb]; // graph_match.py:164-171
                    const float value = 2 * fails <= L1 * L2 ? acc : -1.f;
                    const uint32_t pe = PMX_ITEM_DIET ? row_i + __umul24((uint32_t)sa, nd_i) + off_j + (uint32_t)sb
                                                      : row_i + (uint32_t)sa * nd_i + off_j + (uint32_t)sb; // entry((i, sa) -> (j, sb)); sa < 64, nd_i <= 20 x 64
                    if (on) Pt[(size_t)pe * G + c] = value;
                    const unsigned long long pos = __ballot(on && value > 0.f);
                    if (on && c == 0) {
                        const unsigned long long m = (pos >> (s * G)) & GM;
                        unsigned char *ve = Vt + (size_t)pe * vmask_bytes<G>();
                        if (G <= 8) *ve = (unsigned char)m;
                        else if (G == 16) *reinterpret_cast<uint16_t *>(ve) = (uint16_t)m;
                        else if (G == 32) *reinterpret_cast<uint32_t *>(ve) = (uint32_t)m;
                        else *reinterpret_cast<unsigned long long *>(ve) = m;
                    }
                };
                if (EXACT) {
                    for (int p0 = 0; p0 < npass; p0 += SLOTS) {
                        const bool on = p0 + s < npass;
                        const int e = eb + (int)plist[on ? p0 + s : p0];
                        const int sa = (int)(((float)e + 0.5f) * inv_kj), sb = e - sa * kj;
                        float acc = 0.f;
                        int fails = 0;
                        const int rowa = nci + sa * ni, rowb = ncj + sb * nj;
                        for (int u = 0; u < ni; ++u) {
                            const uint32_t sidu = nc[rowa + u];
                            for (int v = 0; v < nj; ++v) {
                                const float d = staged ? dl[(u * nj + v) * G + c] : node_distance(si, u, sj, v);
                                item<EXACT, false>(p, sidu, nc[rowb + v], d, acc, fails, n_exact, n_exactv);
                            }
                        }
                        finish_entry(e, on, acc, fails);
                    }
                    n_items += (uint32_t)(((npass + SLOTS - 1) / SLOTS) * ni * nj);
                } else if constexpr (SLOTS == 1) {
                    // 64 conformer lanes: the wavefront works on ONE entry at a time, so nothing forces it through the node pairs that
                    // count for nothing - a node whose subset under the candidate is empty (graph_match.py:148-155: no model node of its
                    // types in the cluster) adds 0 and never fails, and 45 % of the stress model's items are such pairs. (With 8 slots
                    // the slots walk in step, and a pair that is empty for one entry is not for its neighbours.) The items of the chunk
                    // are one list as below - entries in turn, of each its counted pairs in the reference's order, PMX_ITEM_BATCH cells on
                    // the way at a time across entry boundaries - over L1 x L2 pairs per entry instead of all of them.
                    constexpr int IB = PMX_ITEM_BATCH;
                    int total = 0, npass2 = 0;
                    {
                        const bool inl = lane < npass;
                        const int el = plist[inl ? lane : 0];
                        const int e = eb + el;
                        const int sa = (int)(((float)e + 0.5f) * inv_kj), sb = e - sa * kj;
                        int cnt = inl ? (int)lcnt[i * ws.kp + sa] * (int)lcnt[j * ws.kp + sb] : 0;
                        if (inl && cnt == 0) { // no counted pair: the sum of nothing, no fails (match_utils.py:71-74): 0 for every conformer, empty mask
                            const uint32_t pe = row_i + (uint32_t)sa * nd_i + off_j + (uint32_t)sb;
                            float *row = Pt + (size_t)pe * G;
#pragma unroll
                            for (int g = 0; g < G; g += 4) *reinterpret_cast<float4 *>(row + g) = make_float4(0.f, 0.f, 0.f, 0.f);
                            unsigned char *ve = Vt + (size_t)pe * vmask_bytes<G>();
                            for (uint32_t g = 0; g < vmask_bytes<G>(); ++g) ve[g] = 0;
                        }
                        const unsigned long long hb = __ballot(inl && cnt > 0);
                        lds_sync(); // (every lane has read its entry: the list is compacted in place)
                        if (inl && cnt > 0) plist[__builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u))] = (uint8_t)el;
                        npass2 = (int)__popcll(hb);
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) cnt += __shfl_xor(cnt, d);
                        total = uni(cnt);
                    }
                    lds_sync();
                    int le = -1, lu = 0, rowa = 0, rowb = 0;
                    unsigned long long mur = 0ull, mvr = 0ull, mv_full = 0ull; // nodes of the two clusters still to come for the entry being loaded
                    auto counted = [&](int k) { // L1 x L2 of the k-th listed entry
                        const int e = eb + uni((int)plist[k]);
                        const int sa = (int)(((float)e + 0.5f) * inv_kj), sb = e - sa * kj;
                        return uni((int)lcnt[i * ws.kp + sa]) * uni((int)lcnt[j * ws.kp + sb]);
                    };
                    int fk = 0, fin_left = npass2 > 0 ? counted(0) : 0;
                    float acc = 0.f;
                    int fails = 0;
                    for (int t0 = 0; t0 < total; t0 += IB) {
                        ItemLoad Lq[IB];
#pragma unroll
                        for (int q = 0; q < IB; ++q) {
                            const bool in = t0 + q < total; // (past the end: the empty subset pair)
                            int lv = 0;
                            if (in) {
                                if (mvr == 0ull) {
                                    mur &= mur - 1ull; // the next node of the first cluster (0 stays 0)
                                    if (mur == 0ull) { // the next entry
                                        ++le;
                                        const int e = eb + uni((int)plist[le]);
                                        const int sa = (int)(((float)e + 0.5f) * inv_kj), sb = e - sa * kj;
                                        rowa = nci + sa * ni, rowb = ncj + sb * nj;
                                        mur = __ballot(lane < ni && nc[rowa + (lane < ni ? lane : 0)] != 0);
                                        mv_full = __ballot(lane < nj && nc[rowb + (lane < nj ? lane : 0)] != 0);
                                    }
                                    lu = __ffsll((unsigned long long)mur) - 1;
                                    mvr = mv_full;
                                }
                                lv = __ffsll((unsigned long long)mvr) - 1;
                                mvr &= mvr - 1ull;
                            }
                            const int uu = in ? lu : 0, vv = in ? lv : 0;
                            const float d = staged ? dl[(uu * nj + vv) * G + c] : node_distance(si, uu, sj, vv);
                            Lq[q] = item_load(p, in ? (uint32_t)nc[rowa + uu] : 0u, in ? (uint32_t)nc[rowb + vv] : 0u, d, cell_of(p, d));
                        }
#pragma unroll
                        for (int q = 0; q < IB; ++q) {
                            if (t0 + q < total) {
                                item_finish<TAILS>(p, Lq[q], acc, fails, n_exact, n_exactv);
                                if (--fin_left == 0) {
                                    finish_entry(eb + uni((int)plist[fk]), true, acc, fails);
                                    acc = 0.f, fails = 0;
                                    if (++fk < npass2) fin_left = counted(fk);
                                }
                            }
                        }
                    }
                    n_items += (uint32_t)total;
                } else {
                    // The (entry, node pair) items of the chunk as ONE list per slot - slot s takes the passing entries s, s + SLOTS, ...
                    // and every entry its node pairs (u, v) in the reference's order (u outer) - walked PMX_ITEM_BATCH items at a
                    // time: coordinates, distances and cell loads of a batch go out together, whichever entries they belong to,
                    // then the values are added in order and an entry is written when its last pair is in. (With a batch per
                    // entry, entries of one or three node pairs - single-node clusters: most of a large model's - spent half
                    // of every batch on padding, and at 32 / 64 conformer lanes, one entry per pass, nothing overlapped at all.)
                    constexpr int IB = PMX_ITEM_BATCH;
                    const int npair = ni * nj;
                    const int nround = (npass + SLOTS - 1) / SLOTS;
                    const int total = nround * npair;
                    // (The loop is instantiated for staged / computed distances: what is fixed per level pair is decided once, not per
                    // item, and the list's end is tested per batch, not per item. [MI355X] tables alone 57.4 -> 56.7 ms per 1 M ligands.)
                    auto run_items = [&](auto staged_tag, auto tri_tag) {
                        constexpr bool STG = decltype(staged_tag)::value;
                        constexpr bool TRI = decltype(tri_tag)::value; // (false: nothing is known, the general index)
                        int lk = 0, lu = 0, lv = 0, lpos = 0; // next item to load: entry round, node pair, its number
                        int fk = 0, fr = 0;                   // next item to finish: entry round, pair number
                        int rowa = 0, rowb = 0;               // node-candidate rows of this slot's entry of round lk
                        auto slot_entry = [&](int k, bool &on) {
                            on = k * SLOTS + s < npass;
                            return eb + (int)plist[on ? k * SLOTS + s : k * SLOTS];
                        };
                        auto decode = [&](int k) {
                            bool on;
                            const int e = slot_entry(k, on);
                            const int sa = (int)(((float)e + 0.5f) * inv_kj);
                            if (PMX_ITEM_DIET) { // (entries, candidates and nodes are far below 2^23: 24-bit multiplies are full rate, 32-bit ones a quarter)
                                const int sb = e - __mul24(sa, kj);
                                rowa = nci + __mul24(sa, ni), rowb = ncj + __mul24(sb, nj);
                            } else {
                                const int sb = e - sa * kj;
                                rowa = nci + sa * ni, rowb = ncj + sb * nj;
                            }
                        };
                        decode(0);
                        float acc = 0.f;
                        int fails = 0;
                        uint32_t sidu_cur = nc[rowa]; // (diet: the first node's subset is read when the node changes, not per item)
                        auto load_next = [&]() {
                            const float d = STG ? dl[lpos * G + c] : node_distance(si, lu, sj, lv);
                            const uint32_t sidu = PMX_ITEM_DIET ? sidu_cur : (uint32_t)nc[rowa + lu];
                            const ItemLoad L = (PMX_ITEM_DIET && TRI) ? item_load_t<true>(p, sidu, (uint32_t)nc[rowb + lv], d)
                                                                      : item_load(p, sidu, (uint32_t)nc[rowb + lv], d, cell_of(p, d));
                            ++lpos;
                            if (++lv == nj) {
                                lv = 0;
                                if (++lu == ni) {
                                    lu = 0, lpos = 0;
                                    if (++lk < nround) decode(lk);
                                }
                                if (PMX_ITEM_DIET) sidu_cur = nc[rowa + lu];
                            }
                            return L;
                        };
                        auto finish_next = [&](const ItemLoad &L) {
                            item_finish<TAILS>(p, L, acc, fails, n_exact, n_exactv);
                            if (++fr == npair) {
                                bool on;
                                const int e = slot_entry(fk, on);
                                finish_entry(e, on, acc, fails);
                                fr = 0, ++fk;
                                acc = 0.f, fails = 0;
                            }
                        };
                        int t0 = 0;
                        for (; t0 + IB <= total; t0 += IB) { // whole batches: no test of the list's end inside
                            inject_valu<PMX_INJECT_VALU_ITEM>();
                            ItemLoad L[IB];
#pragma unroll
                            for (int q = 0; q < IB; ++q) L[q] = load_next();
#pragma unroll
                            for (int q = 0; q < IB; ++q) finish_next(L[q]);
                        }
                        for (; t0 < total; ++t0) finish_next(load_next()); // what is left of the list, one at a time
                    };
                    // (the triangular index where the distances are staged - nearly every item of nearly every model; one more copy of the loop)
#if PMX_ITEM_DIET
                    // (two copies of the loop, not three: a table that is not triangular - a model whose edge matrix is not symmetric, which the
                    // reference cannot make - takes the general loop, which computes its distances; the kernel is 63 KB beside a 64 KB instruction cache)
                    if (PMX_CUT & 4) {
                    } else if (staged && p.F.tri) run_items(std::true_type{}, std::true_type{});
                    else run_items(std::false_type{}, std::false_type{});
#else
                    if (staged) run_items(std::true_type{}, std::false_type{});
                    else run_items(std::false_type{}, std::false_type{});
#endif
                    n_items += (uint32_t)total;
#ifdef PMX_TABLE_FILL // instrumented builds: [1] wave-iterations of the pair items, [5] slot-items of them that belong to an entry
                    if (lane == 0) {
                        reinterpret_cast<WaveStats *>(lds + ws.off_stat)->dbg[1] += (unsigned long long)total;
                        reinterpret_cast<WaveStats *>(lds + ws.off_stat)->dbg[5] += (unsigned long long)(npass * npair);
                    }
#endif
                }
                lds_sync(); // (the list is rewritten by the next chunk)
                PMX_TICK(4);
            }
        }
    }
}

// DP[x] of the record (see its layout): levels from the last one up, the entries of a level's candidates with all deeper
// candidates - one contiguous run of V masks - read with the lanes spread over entries, the maxima taken in LDS (the walker's
// children cache is idle here). A ligand with more candidates than that holds gets 255 everywhere: nothing is ruled out.
template <int G>
__device__ __forceinline__ void chain_lengths(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const LevelInfo &L, unsigned char *rec) {
    const int lane = lane_id();
    const uint8_t *lk = lds + kOffK;
    const uint16_t *ksum = reinterpret_cast<const uint16_t *>(lds + kOffKsum);
    const uint32_t *rowbase = reinterpret_cast<const uint32_t *>(lds + kOffRow);
    const unsigned char *Vt = rec + rec_v_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    unsigned char *DPt = rec + rec_dp_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    // (32 / 64 conformer lanes keep next to nothing in LDS: there the lengths are worked out in the wave's buffer of path totals in
    // global memory, idle until the walk)
    constexpr bool kInLds = totals_in_lds<G>();
    uint32_t *dpl = kInLds ? reinterpret_cast<uint32_t *>(lds + ws.off_tch) : reinterpret_cast<uint32_t *>(p.totbuf + (size_t)blockIdx.x * kTotBufBytes);
    const uint32_t cap = kInLds ? (ws.bytes - ws.off_tch) / 4u : kTotBufBytes / 4u;
    if (L.ksumtot > cap || (PMX_WFLAGS(p) & (4u | 32768u))) {
        for (uint32_t x = (uint32_t)lane; x < L.ksumtot; x += 64u) DPt[x] = 255;
        return;
    }
    auto sync = [&]() {
        if (kInLds) lds_sync();
        else wave_sync();
    };
    sync();
    for (uint32_t x = (uint32_t)lane; x < L.ksumtot; x += 64u) dpl[x] = 1u;
    constexpr uint32_t VB = vmask_bytes<G>();
    for (int j = L.nl - 2; j >= 0; --j) {
        sync(); // (the deeper levels' lengths are final)
        const uint32_t kj = (uint32_t)uni(lk[j]), ksj = (uint32_t)uni((int)ksum[j]), ks1 = (uint32_t)uni((int)ksum[j + 1]);
        const uint32_t nd = L.ksumtot - ks1, row = (uint32_t)uni((int)rowbase[j]);
        const float inv_nd = 1.0f / (float)nd;
        for (uint32_t e0 = (uint32_t)lane; e0 < kj * nd; e0 += 256u) { // (four masks per lane and trip: their loads in flight together)
            bool v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t e = min(e0 + 64u * u, kj * nd - 1u);
                const unsigned char *ve = Vt + (size_t)(row + e) * VB;
                if (VB == 1) v[u] = *ve != 0;
                else if (VB == 2) v[u] = *reinterpret_cast<const uint16_t *>(ve) != 0;
                else if (VB == 4) v[u] = *reinterpret_cast<const uint32_t *>(ve) != 0u;
                else v[u] = *reinterpret_cast<const unsigned long long *>(ve) != 0ull;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t e = e0 + 64u * u;
                if (e < kj * nd && v[u]) {
                    const uint32_t a = (uint32_t)(((float)e + 0.5f) * inv_nd), xo = e - a * nd;
                    atomicMax(&dpl[ksj + a], dpl[ks1 + xo] + 1u);
                }
            }
        }
    }
    sync();
    for (uint32_t x = (uint32_t)lane; x < L.ksumtot; x += 64u) DPt[x] = (unsigned char)dpl[x];
    sync();
}

// Upper bounds for the tree search: level l can add at most
//   U[l][c] = max(0, max_b (S[l][b][c] + sum_{j < l} max(0, max_a P[(j, a), (l, b)][c])))
// to a conformer's total whatever is picked on the other levels, so R[f][c] = sum_{l >= f} U[l][c] bounds everything the
// levels f.. add. (The reported score only needs the per-conformer maximum over leaves, graph_match.py:103-109.)
template <int G>
__device__ __forceinline__ void build_bounds(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const LevelInfo &L, unsigned char *rec) {
    constexpr int SLOTS = 64 / G;
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    const uint8_t *lk = lds + kOffK;
    const uint16_t *ksum = reinterpret_cast<const uint16_t *>(lds + kOffKsum);
    const uint32_t *rowbase = reinterpret_cast<const uint32_t *>(lds + kOffRow);
    const float *St = reinterpret_cast<const float *>(rec + rec_s_off<G>());
    const float *Pt = reinterpret_cast<const float *>(rec + rec_p_off<G>(L.ksumtot));
    double *Rt = reinterpret_cast<double *>(rec + rec_r_off<G>(L.ksumtot, L.T));
    unsigned char *Wraw = rec + rec_w_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    unsigned char *OBraw = rec + rec_ob_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    // W[i] (a bound: rounded up when it is kept as a float32) and OB[i] (bfloat16 rounded up where per-candidate bounds exist)
    auto w_put = [&](size_t i, double v) {
        if (kSlimBounds) reinterpret_cast<float *>(Wraw)[i] = float_up(v);
        else reinterpret_cast<double *>(Wraw)[i] = v;
    };
    auto w_get = [&](size_t i) -> double { return kSlimBounds ? (double)reinterpret_cast<const float *>(Wraw)[i] : reinterpret_cast<const double *>(Wraw)[i]; };
    auto ob_put = [&](size_t i, double v) {
        if (ob_elt_bytes<G>() == 2) reinterpret_cast<uint16_t *>(OBraw)[i] = bf16_up(float_up(v));
        else reinterpret_cast<float *>(OBraw)[i] = float_up(v);
    };
    unsigned char *LVt = rec + rec_ci_off<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    const int nl = L.nl;
    // Round 6: the level maxima the base pass works out - MP[j][x] = max(0, max_a P[(j, a), x]) for a candidate x of a deeper level - are kept (in the
    // wave's path-sum buffer, idle until the walk) for the W pass below, which used to work every one of them out again per window of level j's
    // candidates: a loop over the level's candidates and a cross-slot maximum per deeper candidate ([MI355X] 5.4 of the table phase's 41.4 ms).
#ifndef PMX_W_FROM_MAXIMA
#define PMX_W_FROM_MAXIMA 1
#endif
    float *MP = reinterpret_cast<float *>(p.pabuf + (size_t)blockIdx.x * p.pa_bytes);
    const bool mp_ok = PMX_W_FROM_MAXIMA && cand_bounds<G>() && p.pabuf != nullptr && (uint64_t)nl * L.ksumtot * G * 4u <= (uint64_t)p.pa_bytes;
#ifdef PMX_TABLE_TICKS
    unsigned long long tick_ = __builtin_amdgcn_s_memtime();
#endif
    if (PMX_CUT & 2) return;
    chain_lengths<G>(p, lds, ws, L, rec);
    PMX_TICK(5);
    if (PMX_WFLAGS(p) & 4) { // debug: nothing is ever dropped
        for (int l = s; l <= nl; l += SLOTS) Rt[(size_t)l * G + c] = __builtin_inf();
        if (cand_bounds<G>())
            for (uint32_t e = s; e < L.ksumtot; e += SLOTS) w_put((size_t)e * G + c, __builtin_inf());
        return;
    }
    double suffix = 0.0;
    if (s == 0) Rt[(size_t)nl * G + c] = 0.0;
    for (int l = nl - 1; l >= 0; --l) {
        const int kl = uni(lk[l]), ksl = uni(ksum[l]);
        double u = 0.0;
        for (int b = s; b < kl; b += SLOTS) {
            // the levels above l from the nearest one up: what has been added when level j is reached is what (l, b) can add
            // apart from its pair entries with levels <= j - OB[j][(l, b)], path_bound()'s table
            double v = (double)St[(size_t)(ksl + b) * G + c];
            // (two levels above l per trip, eight entries of each in flight: a maximum does not mind the last candidate being read again where
            // fewer are left. One load at a time, each waited for, this loop was a memory round trip per candidate of every level above.)
            for (int j = l - 1; j >= 0; j -= 2) {
                const int j2 = j - 1; // (-1: level j is the last one)
                const int kj = uni(lk[j]), kj2 = j2 >= 0 ? uni(lk[j2]) : 0;
                const uint32_t nd_j = L.ksumtot - (uint32_t)uni((int)ksum[j + 1]), nd_j2 = L.ksumtot - (uint32_t)uni((int)ksum[j2 + 1]);
                const uint32_t e0 = (uint32_t)uni((int)rowbase[j]) + (uint32_t)(ksl - uni((int)ksum[j + 1])) + (uint32_t)b; // entry((j, 0) -> (l, b))
                const uint32_t e02 = j2 >= 0 ? (uint32_t)uni((int)rowbase[j2]) + (uint32_t)(ksl - uni((int)ksum[j2 + 1])) + (uint32_t)b : e0;
                float m = 0.f, m2 = 0.f;
                for (int a0 = 0; a0 < max(kj, kj2); a0 += 8) {
                    float pv[8], pw[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        pv[u] = Pt[(size_t)(e0 + (uint32_t)min(a0 + u, kj - 1) * nd_j) * G + c];
                        pw[u] = j2 >= 0 ? Pt[(size_t)(e02 + (uint32_t)min(a0 + u, kj2 - 1) * nd_j2) * G + c] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        m = pv[u] > m ? pv[u] : m;
                        m2 = pw[u] > m2 ? pw[u] : m2;
                    }
                }
                if (cand_bounds<G>()) ob_put(((size_t)j * L.ksumtot + (size_t)(ksl + b)) * G + c, v);
                if (mp_ok) MP[((size_t)j * L.ksumtot + (size_t)(ksl + b)) * G + c] = m;
                v += (double)m;
                if (j2 >= 0) {
                    if (cand_bounds<G>()) ob_put(((size_t)j2 * L.ksumtot + (size_t)(ksl + b)) * G + c, v);
                    if (mp_ok) MP[((size_t)j2 * L.ksumtot + (size_t)(ksl + b)) * G + c] = m2;
                    v += (double)m2;
                }
            }
            if (cand_bounds<G>()) w_put((size_t)(ksl + b) * G + c, v); // base(l, b), replaced by the candidate's own bound below
            else ob_put((size_t)(ksl + b) * G + c, v);                 // BF: base(l, b) for path_bound_wide()
            if (c == 0) LVt[ksl + b] = (unsigned char)l;
            u = v > u ? v : u;
        }
#pragma unroll
        for (int d = G; d < 64; d <<= 1) {
            const double o = __shfl_xor(u, d);
            u = o > u ? o : u;
        }
        suffix += u;
        if (s == 0) Rt[(size_t)l * G + c] = suffix;
    }
    // W[(f, b)][c]: what the levels below f can add under a path whose newest match is (f, b) - as U, but with (f, b)'s own
    // pair entries instead of level f's maxima, over the candidates compatible with (f, b) only. Levels in ascending order:
    // the entries of the levels l > f still hold base(l, .).
    if (!cand_bounds<G>()) return; // one or two candidates per pass (32 / 64 conformers): the walker uses R
    wave_sync();
    // (the work below grows with windows^2 per level: with very many candidates it would cost more than the walk saves, and
    // every candidate gets its level's bound instead)
    uint32_t cost = 0;
    for (int f = 0; f < nl; ++f) {
        const uint32_t wf = ((uint32_t)uni(lk[f]) + SLOTS - 1) / SLOTS;
        cost += wf * wf * (L.ksumtot - (uint32_t)uni((int)ksum[f + 1]));
    }
    if ((PMX_WFLAGS(p) & 512) || cost > p.bound_cost) {
        for (int f = 0; f < nl; ++f) {
            const int kf = uni(lk[f]), ksf = uni(ksum[f]);
            const double r = Rt[(size_t)(f + 1) * G + c];
            wave_sync();
            for (int b = s; b < kf; b += SLOTS) w_put((size_t)(ksf + b) * G + c, r);
        }
        return;
    }
    if (mp_ok) {
        // Slot s <-> candidate b = b0 + s of level f, alone with its own entries: for every deeper candidate x = (l, b1) its base with level f's
        // maximum taken out and (f, b)'s own entry put in - three loads and two additions - the largest per level, the levels added up. The same
        // numbers in the same order as the loop this replaces: the same W to the last bit.
        for (int f = 0; f < nl; ++f) {
            const int kf = uni(lk[f]), ksf = uni(ksum[f]);
            const uint32_t x0 = (uint32_t)uni((int)ksum[f + 1]), nd_f = L.ksumtot - x0;
            const float *MPf = MP + (size_t)f * L.ksumtot * G;
            for (int b0 = 0; b0 < kf; b0 += SLOTS) {
                const int b = b0 + s;
                const float *Pb_ = Pt + ((size_t)(uint32_t)uni((int)rowbase[f]) + (size_t)(uint32_t)min(b, kf - 1) * nd_f) * G; // entry((f, b) -> x) = rowbase[f] + b nd_f + (x - x0)
                double acc = 0.0;
                for (int l = f + 1; l < nl; ++l) {
                    const int kl = uni(lk[l]), ksl = uni(ksum[l]);
                    double u = 0.0;
                    constexpr int B1 = 4; // (deeper candidates per trip, their loads in flight together; a maximum does not mind the last one being taken again)
                    for (int b10 = 0; b10 < kl; b10 += B1) {
                        double base[B1];
                        float mp[B1], pv[B1];
#pragma unroll
                        for (int q = 0; q < B1; ++q) {
                            const uint32_t x = (uint32_t)(ksl + min(b10 + q, kl - 1));
                            base[q] = w_get((size_t)x * G + c);
                            mp[q] = MPf[(size_t)x * G + c];
                            pv[q] = Pb_[(size_t)(x - x0) * G + c];
                        }
#pragma unroll
                        for (int q = 0; q < B1; ++q) {
                            const double val = (base[q] - (double)mp[q]) + (double)pv[q];
                            u = (pv[q] > 0.f && val > u) ? val : u;
                        }
                    }
                    acc += u;
                }
                if (b < kf) w_put((size_t)(ksf + b) * G + c, acc * (1.0 + 1e-12));
            }
        }
        wave_sync();
        return;
    }
    // (no room for the maxima: every candidate gets its level's bound)
    for (int f = 0; f < nl; ++f) {
        const int kf = uni(lk[f]), ksf = uni(ksum[f]);
        const double r = Rt[(size_t)(f + 1) * G + c];
        wave_sync();
        for (int b = s; b < kf; b += SLOTS) w_put((size_t)(ksf + b) * G + c, r);
    }
}

// ------------------------------------------------------------------------------------------ kernels
// Bump allocation in the arena by lane 0; returns the byte offset (never 0: offset 0 means "not in the arena") or ~0ull.
__device__ inline unsigned long long arena_alloc(const ScreenParams &p, uint32_t bytes) {
    unsigned long long off = 0;
    if ((threadIdx.x & 63) == 0) off = atomicAdd(&p.ctl->arena_top, (unsigned long long)((bytes + 255u) & ~255u)) + 256ull;
    off = uni64(off);
    return off + bytes <= p.arena_bytes ? off : ~0ull;
}

#ifndef PMX_SCREEN_WAVES
#define PMX_SCREEN_WAVES 6 // waves per SIMD the register budget is set for (<= 80 VGPRs: nothing spilled to memory)
#endif


// A job of a wavefront is a subtree record: one taken from the queue (the ligand's tables are in the arena), or the root of
// a ligand whose tables this wave has just built (record in the wave's LDS, tables in its slice or in the arena).

// Ligand -> job: levels, tables, bounds, and the root's subtree record in LDS. Returns the ligand's record (slice or arena), or
// nullptr when the ligand is finished without a tree search (unsupported record, no candidates, tables too large for this pass).
template <int G, bool EXACT, bool TAILS>
__device__ __forceinline__ unsigned char *prepare_ligand(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const uint32_t li, const uint32_t wave_id,
                                                         WaveStats *stat) {
    const int lane = lane_id();
    const int c = lane % G;
    Record r = parse_record(uniptr(p.lib.data + p.lib.offsets[p.first + li]));
    r.n = uni(r.n), r.C = uni(r.C), r.ncl = uni(r.ncl); // the record is the same for the whole wave: say so
    r.typemask = uniptr(r.typemask), r.cluster_end = uniptr(r.cluster_end), r.xyz = uniptr(r.xyz);
    if (p.mode == 0) {
        if (!record_supported(r)) {
            if (lane == 0) {
                put_score(p, li, __builtin_nan(""));
                if (p.status) p.status[li] = PMX_LIGAND_UNSUPPORTED;
            }
            return nullptr;
        }
        if (lane == 0 && p.status) p.status[li] = PMX_LIGAND_OK;
    }
    const unsigned long long t_a = __builtin_amdgcn_s_memtime();
    const LevelInfo L = scan_ligand<G>(p, lds, ws, r);
    if (L.nl < 0) { // a ligand cluster with more than PMX_MAX_LEVEL_CANDIDATES candidate clusters
        if (lane == 0) {
            put_score(p, li, __builtin_nan(""));
            if (p.status) p.status[li] = PMX_LIGAND_UNSUPPORTED;
        }
        return nullptr;
    }
    if (L.nl == 0) { // no ligand cluster has a candidate (graph_match.py:95-99)
        if (lane == 0) put_score(p, li, 0.0);
        return nullptr;
    }
    const uint64_t bytes64 = rec_bytes<G>(L.ksumtot, L.T, (uint32_t)L.nl);
    unsigned char *rec = p.slices + (size_t)wave_id * p.slice_bytes;
    uint32_t rec16 = 0;
    if (p.mode < 2) {
        if (bytes64 > p.slice_bytes) { // tables do not fit the slice: a later pass with larger slices (or the arena) takes this ligand
            if (lane == 0) {
                if (p.mode == 0) {
                    const uint32_t o = atomicAdd(&p.ctl->ovf_count, 1u);
                    if (o < p.list_cap) p.ovf_list[o] = li;
                    atomicAdd(&p.ctl->stats[wave_id & (kScreenStatShards - 1)][7], 1ull);
                } else {
                    const uint32_t o = atomicAdd(&p.ctl->carry_count, 1u);
                    if (o < p.list_cap) p.carry_list[o] = li;
                }
            }
            return nullptr;
        }
    } else {
        const bool fits = bytes64 < (1ull << 31) && bytes64 + 256ull <= p.arena_bytes;
        const unsigned long long off = fits ? arena_alloc(p, (uint32_t)bytes64) : ~0ull;
        if (off == ~0ull) {
            // No room. Tables larger than the whole arena are reported; otherwise the arena is full of other ligands' tables (it
            // is a bump allocator that empties between passes), which says nothing about this ligand: it is listed and taken
            // again by a later arena pass that starts empty, so that a score does not depend on what else is in the batch.
            if (lane == 0) {
                if (fits && p.retry_out) {
                    const uint32_t o = atomicAdd(&p.ctl->retry_count[p.retry_slot], 1u);
                    if (o < p.list_cap) p.retry_out[o] = li;
                } else {
                    put_score(p, li, __builtin_nan(""));
                    if (p.status) p.status[li] = PMX_LIGAND_TOO_LARGE;
                }
            }
            return nullptr;
        }
        rec = p.arena + off;
        rec16 = (uint32_t)(off >> 4);
    }
    // ---- header
    RecHeader *H = reinterpret_cast<RecHeader *>(rec);
    const uint8_t *lk = lds + kOffK;
    const uint16_t *ksum = reinterpret_cast<const uint16_t *>(lds + kOffKsum);
    const uint32_t *rowbase = reinterpret_cast<const uint32_t *>(lds + kOffRow);
    if (lane == 0) {
        H->lig = li;
        H->nl = (uint32_t)L.nl;
        H->T = L.T;
        H->ksumtot = L.ksumtot;
        H->bytes = (uint32_t)bytes64;
        H->C = (uint32_t)r.C;
        H->pad[0] = 0; // not (yet) registered for finalize_kernel
    }
    if (lane < L.nl) {
        H->k[lane] = lk[lane];
        H->rowbase[lane] = rowbase[lane];
    }
    if (lane <= L.nl) H->ksum[lane] = ksum[lane];
    if (lane < G) reinterpret_cast<unsigned long long *>(rec + sizeof(RecHeader))[lane] = 0ull;
    uint32_t n_items = 0, n_exact = 0, n_exactv = 0, n_dead = 0;
    const unsigned long long t_b = __builtin_amdgcn_s_memtime();
    build_tables<G, EXACT, TAILS>(p, lds, ws, r, L, rec, n_items, n_exact, n_exactv, n_dead);
    wave_sync();
    const unsigned long long t_c = __builtin_amdgcn_s_memtime();
    build_bounds<G>(p, lds, ws, L, rec);
    // ---- the root as a subtree record (in LDS): frame 0, no matches, every conformer, totals 0
    {
        unsigned char *tr = lds + ws.off_task;
        TaskRec *th = reinterpret_cast<TaskRec *>(tr);
        if (lane == 0) {
            th->rec16 = rec16;
            th->f0 = 0;
            th->nm = 0;
            th->pad = 0;
            th->mask = (r.C >= 64) ? ~0ull : ((1ull << r.C) - 1ull);
        }
        if (lane < G) reinterpret_cast<double *>(tr + sizeof(TaskRec))[c] = 0.0;
    }
    wave_sync();
    const unsigned long long t_d = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        stat->cyc_scan += t_b - t_a, stat->cyc_tables += t_c - t_b, stat->cyc_bounds += t_d - t_c;
        stat->items += n_items;
        stat->dead += n_dead;
    }
    if (n_exact) atomicAdd(&stat->exact, (unsigned long long)n_exact);
    if (n_exactv) atomicAdd(&stat->exactv, (unsigned long long)n_exactv);
    return rec;
}

// Subtree record -> walker state. Returns false when the subtree can no longer raise any maximum (the maxima may have grown
// since it was queued) and is dropped unwalked.
template <int G>
__device__ __forceinline__ bool prepare_walk(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, const unsigned char *tr, unsigned char *rec, Walk<G> &w) {
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    double *tot = totals_in_lds<G>() ? reinterpret_cast<double *>(lds + ws.off_tot) : reinterpret_cast<double *>(p.totbuf + (size_t)blockIdx.x * kTotBufBytes);
    unsigned long long *pool = reinterpret_cast<unsigned long long *>(lds + ws.off_pool);
    const TaskRec *th = reinterpret_cast<const TaskRec *>(tr);
    const RecHeader *H = reinterpret_cast<const RecHeader *>(rec);
    const int nl = uni((int)H->nl);
    const uint32_t ksumtot = (uint32_t)uni((int)H->ksumtot), T = (uint32_t)uni((int)H->T);
    w.Sb = rec + rec_s_off<G>();
    w.Pb = rec + rec_p_off<G>(ksumtot);
    w.Rb = rec + rec_r_off<G>(ksumtot, T);
    w.Wb = rec + rec_w_off<G>(ksumtot, T, (uint32_t)nl);
    w.Vb = rec + rec_v_off<G>(ksumtot, T, (uint32_t)nl);
    w.OBb = rec + rec_ob_off<G>(ksumtot, T, (uint32_t)nl);
    w.nl = nl;
    w.ksumtot = ksumtot;
    // path_bound() keeps a row of pair sums per candidate and match count in the wave's buffer: used when they fit
    // (and the table word X holds a pair entry number in 20 bits)
    w.path_on = cand_bounds<G>() && !(PMX_WFLAGS(p) & (4u | 1024u)) && (uint64_t)(nl + 1) * ksumtot * G * 4u <= (uint64_t)p.pa_bytes && T < (1u << 20);
    {
        const int kl = lane < nl ? (int)H->k[lane] : 0, knext = lane + 1 < nl ? (int)H->k[lane + 1] : 0;
        const int tci = nl - 3 - lane;
        int kind = lane == nl - 1 ? kLvLeaf : 0;
        if (lane == nl - 2 && knext <= 64 / G && !(PMX_WFLAGS(p) & 32)) kind |= kLvFuse;
        if (totals_in_lds<G>() && tci >= 0 && tci < kTcLevels && kl <= 64 / G && !(PMX_WFLAGS(p) & 64)) kind |= kLvCache | (tci << 12);
        w.hk = kl | kind;
    }
    w.hks = lane <= nl ? (int)H->ksum[lane] : 0;
    w.hrow = lane < nl ? (int)H->rowbase[lane] : 0;
    const int nm0 = uni((int)th->nm), f0 = uni((int)th->f0);
    if (lane < nm0) {
        const int j = th->path[2 * lane], a = th->path[2 * lane + 1];
        w.matB = (int)H->rowbase[j] + a * ((int)ksumtot - (int)H->ksum[j + 1]) - (int)H->ksum[j + 1];
        w.matKA = (a << 8) | (j << 16);
    }
    const unsigned long long *gbest = reinterpret_cast<const unsigned long long *>(rec + sizeof(RecHeader));
    if (s == 0) {
        tot[nm0 * G + c] = reinterpret_cast<const double *>(tr + sizeof(TaskRec))[c];
        pool[c] = __hip_atomic_load(&gbest[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // maxima of the ligand's finished walkers
    }
    w.f = w.f0 = f0;
    const uint64_t mask0 = uni64(th->mask);
    w.stA = wl(w.stA, f0, (int)(uint32_t)mask0);
    if (G > 32) w.stB = wl(w.stB, f0, (int)(uint32_t)(mask0 >> 32));
    else w.stB = -1;
    w.stC = wl(w.stC, f0, (((nm0 ? (int)kMatched : 0) | (w.path_on ? (int)kPath : 0)) << 16) | (nm0 << 24));
    wave_sync();
    if (!(PMX_WFLAGS(p) & 4) && f0 < nl && nm0 >= 5) {
        const double r = *reinterpret_cast<const double *>(w.Rb + ((size_t)f0 * G + c) * 8);
        const double t = tot[nm0 * G + c];
        if (__ballot(((mask0 >> c) & 1ull) && (t + r) * kBoundSlack > __longlong_as_double((long long)pool[c])) == 0) return false;
    }
    if (cand_bounds<G>() && w.path_on && nm0 > 0) // a queued subtree: the pair sums of the matches it starts with
        path_sums_of_root<G>(w, reinterpret_cast<PaElt *>(p.pabuf + (size_t)blockIdx.x * p.pa_bytes), nm0);
    return true;
}

// The tree search of a prepared job and what follows it: the maxima go to the score (a ligand walked by this wave alone) or to
// the ligand's record in the arena (a split ligand; finalize_kernel takes the mean). The walk is interrupted once, when it
// runs over its budget: a ligand's tables then move to the arena (queued subtrees refer to them) and the walk resumes handing
// subtrees with >= 5 matches to the queue.
template <int G>
__device__ __forceinline__ void run_job(const ScreenParams &p, unsigned char *lds, const WaveShape<G> &ws, Walk<G> &w, unsigned char *rec, uint32_t rec16,
                                        const bool is_task, const uint32_t wave_id, WaveStats *stat) {
    const int lane = lane_id();
    const int s = lane / G, c = lane % G;
    double *tot = totals_in_lds<G>() ? reinterpret_cast<double *>(lds + ws.off_tot) : reinterpret_cast<double *>(p.totbuf + (size_t)blockIdx.x * kTotBufBytes);
    unsigned long long *pool = reinterpret_cast<unsigned long long *>(lds + ws.off_pool);
    uint16_t *pathbuf = reinterpret_cast<uint16_t *>(lds + kOffPath);
    double *tch = reinterpret_cast<double *>(lds + ws.off_tch);
    double *tc = reinterpret_cast<double *>(lds + ws.off_tc);
    unsigned long long *cbl = reinterpret_cast<unsigned long long *>(lds + ws.off_cb);
    PaElt *pa = reinterpret_cast<PaElt *>(p.pabuf + (size_t)blockIdx.x * p.pa_bytes);
    float *ub = reinterpret_cast<float *>(lds + ws.off_ub);
    const unsigned long long t_d = __builtin_amdgcn_s_memtime();
    unsigned long long budget = ((PMX_WFLAGS(p) & 2) || p.last_round) ? ~0ull : (unsigned long long)p.budget;
    bool export_mode = false, split = is_task;
    for (;;) {
        const int rc = walk<G>(w, p, tot, pool, pathbuf, tch, tc, cbl, pa, ub, rec16, export_mode, budget, wave_id, stat);
        if (rc != kOverBudget) break;
        if (lane == 0) ++stat->over;
        budget = ~0ull;
        RecHeader *H = reinterpret_cast<RecHeader *>(rec);
        if (rec16 == 0) { // tables in the wave's slice: move them to the arena
            const uint32_t bytes = (uint32_t)uni((int)H->bytes);
            const unsigned long long off = arena_alloc(p, bytes);
            if (off != ~0ull) {
                const uint4 *src = reinterpret_cast<const uint4 *>(rec);
                uint4 *dst = reinterpret_cast<uint4 *>(p.arena + off);
                const uint32_t n16 = (bytes + 15u) / 16u;
                for (uint32_t i = lane; i < n16; i += 64) dst[i] = src[i];
                rec16 = (uint32_t)(off >> 4);
                rec = p.arena + off;
                // (the walker keeps reading the slice copy through w.Sb / Pb / Rb: same bytes)
            }
        }
        if (rec16 != 0) { // (arena full otherwise: the wave walks the tree alone - exact, only slower)
            export_mode = true;
            H = reinterpret_cast<RecHeader *>(rec);
            if (!is_task && !split) { // first time: finalize_kernel has to score this ligand
                if (lane == 0) {
                    const uint32_t o = atomicAdd(&p.ctl->heavy_count, 1u);
                    if (o < p.list_cap) p.heavy_list[o] = rec16;
                }
            }
            split = true; // (the arena copy is read by later kernels: nothing to fence)
        }
    }
    if (lane == 0) {
        stat->cyc_walk += __builtin_amdgcn_s_memtime() - t_d;
        stat->frames += w.frames;
        stat->passes += w.passes;
        stat->npath += w.npath;
        stat->dbg[7] += w.ndrop; // (the last word of the instrumented builds' counters is the product's: children dropped by path_bound())
        stat->longest = w.passes > stat->longest ? w.passes : stat->longest;
    }
    // ---- per-conformer maxima over the slots -> score
    if (w.best > 0.0) atomicMax(&pool[c], (unsigned long long)__double_as_longlong(w.best));
    wave_sync();
    const unsigned long long bbits = pool[c];
    const RecHeader *H = reinterpret_cast<const RecHeader *>(rec);
    if (split) { // every walker of a split ligand adds its maxima, finalize_kernel takes the mean
        unsigned long long *gbest = reinterpret_cast<unsigned long long *>(rec + sizeof(RecHeader));
        if (s == 0 && bbits != 0ull) atomicMax(gbest + c, bbits);
    } else { // mean over conformers (graph_match.py:109); lanes beyond C hold 0
        const int C = uni((int)H->C);
        double sum = (s == 0 && c < C) ? __longlong_as_double((long long)bbits) : 0.0;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) put_score(p, (uint32_t)uni((int)H->lig), sum / (double)C);
    }
    wave_sync();
}


__device__ inline void flush_wave_stats(const ScreenParams &p, const WaveStats *stat, uint32_t wave_id, unsigned long long alive) {
    unsigned long long *st = p.ctl->stats[wave_id & (kScreenStatShards - 1)];
    atomicAdd(st + 0, stat->frames);
    atomicAdd(st + 1, stat->passes);
    atomicAdd(st + 2, stat->over);
    atomicAdd(st + 3, stat->items);
    atomicAdd(st + 4, stat->exact);
    atomicMax(st + 5, stat->longest);
    atomicAdd(st + 6, stat->tasks);
    atomicAdd(st + 14, stat->overflow);
    atomicAdd(st + 15, stat->pad[0]);
    atomicAdd(st + 7, stat->pad[1] << 32);
    atomicAdd(st + 8, stat->cyc_scan);
    atomicAdd(st + 9, stat->cyc_tables);
    atomicAdd(st + 10, stat->cyc_bounds);
    atomicAdd(st + 11, stat->cyc_walk);
    atomicAdd(st + 12, alive);
    atomicAdd(st + 13, stat->exactv);
    atomicAdd(st + 22, stat->npath);
    atomicAdd(st + 23, stat->dbg[7]);
    atomicAdd(st + 24, stat->dead);
#if defined(PMX_COUNTERS) || defined(PMX_TABLE_TICKS) || defined(PMX_TABLE_FILL) || defined(PMX_WALK_TICKS)
    for (int i = 0; i < 6; ++i) atomicAdd(st + 16 + i, stat->dbg[i]);
#endif
}

// Persistent wavefronts (one per block) over the ligands of a pass: build a ligand's tables in the wave's slice (modes 0 / 1)
// or in the arena (mode 2), walk its tree, write its score. A tree that runs over its budget hands its open subtrees to the
// task queue, which task_kernel drains afterwards. (One kernel for ligands and queued subtrees together was built: the two
// bodies in one loop cost 80-300 spilled registers, inside the walker's pass loop; apart they need none.)
template <int G, bool EXACT, bool TAILS>
__global__ __launch_bounds__(64, PMX_SCREEN_WAVES) void ligand_kernel(const ScreenParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane0 = lane_id();
    const uint32_t wave_id = blockIdx.x;
    const WaveShape<G> ws = wave_shape<G>(p.M.K, (int)p.max_nodes);
    const uint32_t todo = p.mode == 0 ? p.hi - p.lo
                          : min(p.mode == 1 ? p.ctl->ovf_count : (p.mode == 2 ? p.ctl->carry_count : p.ctl->retry_count[p.retry_slot ^ 1u]), p.list_cap);
    const uint32_t *list = p.mode == 1 ? p.ovf_list : (p.mode == 2 ? p.carry_list : p.retry_in);
    constexpr uint32_t kBatch = 1; // ligands claimed per atomic on the cursor
    WaveStats *stat = reinterpret_cast<WaveStats *>(lds + ws.off_stat);
    if (lane0 < (int)(sizeof(WaveStats) / 8)) reinterpret_cast<unsigned long long *>(stat)[lane0] = 0ull;
    wave_sync();
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    uint32_t lig_next = 0, lig_end = 0;
    for (;;) {
        const int lane = lane_id();
        if (lig_next == lig_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&p.ctl->cursor[p.mode], kBatch);
            base = (uint32_t)uni((int)base);
            lig_next = min(base, todo);
            lig_end = min(base + kBatch, todo);
            if (lig_next == lig_end) break;
        }
        const uint32_t next = lig_next++;
        const uint32_t li = p.mode == 0 ? p.lo + next : (uint32_t)uni((int)list[next]);
        unsigned char *rec = prepare_ligand<G, EXACT, TAILS>(p, lds, ws, li, wave_id, stat);
        if (!rec) continue;
        const unsigned char *root = lds + ws.off_task;
        const uint32_t rec16 = (uint32_t)uni((int)reinterpret_cast<const TaskRec *>(root)->rec16);
        Walk<G> w;
        if (!(PMX_WFLAGS(p) & 16384) && prepare_walk<G>(p, lds, ws, root, rec, w)) run_job<G>(p, lds, ws, w, rec, rec16, false, wave_id, stat);
    }
    wave_sync();
    if (lane0 == 0) flush_wave_stats(p, stat, wave_id, __builtin_amdgcn_s_memtime() - t_start);
}

// Snapshot of the task queue between rounds: the records reserved since the last round are this round's subtrees. (Rounds are
// separate launches on purpose: what one wave queues has to be visible to waves on other XCDs, whose L2 is not coherent
// with the writer's inside a kernel - a queue drained by the kernel that fills it needs an L2 write-back per hand-over,
// measured at 15x the walk time.)
__global__ void round_kernel(Ctl *ctl, uint32_t qcap) {
    const int lane = threadIdx.x & 63;
    const uint32_t lo = ctl->round_hi[lane], hi = min(ctl->q_res[lane], qcap);
    uint32_t n = hi - lo, inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) n += __shfl_xor(n, d);
    ctl->round_lo[lane] = lo;
    ctl->round_hi[lane] = hi;
    ctl->round_inc[lane] = inc;
    if (lane == 0) {
        ctl->round_total = n;
        ctl->task_cursor = 0;
        for (int x = 0; x < 8; ++x) ctl->xcd_cursor[x][0] = 0;
    }
}

// Persistent wavefronts over the round's subtrees; a subtree that runs over its budget queues its own open subtrees for the
// next round (the last round's budget is unlimited).
#ifndef PMX_TASK_WAVES
#define PMX_TASK_WAVES 6
#endif
template <int G>
__global__ __launch_bounds__(64, PMX_TASK_WAVES) void task_kernel(const ScreenParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane0 = lane_id();
    const uint32_t wave_id = blockIdx.x;
    const uint32_t total = p.ctl->round_total;
    if (total == 0) return;
    const WaveShape<G> ws = wave_shape<G>(p.M.K, (int)p.max_nodes);
    WaveStats *stat = reinterpret_cast<WaveStats *>(lds + ws.off_stat);
    if (lane0 < (int)(sizeof(WaveStats) / 8)) reinterpret_cast<unsigned long long *>(stat)[lane0] = 0ull;
    wave_sync();
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    // Task number -> (shard, record) through the inclusive counts round_kernel left. Shards 8x .. 8x + 7 (a contiguous range
    // of task numbers) belong to XCD x - block b runs on XCD b % 8 on this part, which only matters for speed: a wavefront
    // takes from its own group until it is empty, then from the next ones. (Nothing of this stays in registers over a walk.)
    uint32_t skip = 0;
    for (;;) {
        const int lane = lane_id();
        const uint32_t inc = p.ctl->round_inc[lane];
        uint32_t nx = 0xffffffffu;
        while (skip < 8) {
            const int xx = (int)((blockIdx.x + skip) & 7u);
            const uint32_t st = xx ? (uint32_t)rl((int)inc, 8 * xx - 1) : 0u, en = (uint32_t)rl((int)inc, 8 * xx + 7);
            if (st < en) {
                uint32_t cur = 0;
                if (lane == 0) cur = atomicAdd(&p.ctl->xcd_cursor[xx][0], 1u);
                cur = (uint32_t)uni((int)cur);
                if (cur < en - st) {
                    nx = st + cur;
                    break;
                }
            }
            ++skip;
        }
        if (nx == 0xffffffffu) break;
        const int sh = __popcll(__ballot(nx >= inc)); // shards wholly before task nx (inc is non-decreasing)
        const uint32_t before = sh ? (uint32_t)rl((int)inc, sh - 1) : 0u;
        const uint32_t recno = (uint32_t)uni((int)p.ctl->round_lo[sh]) + (nx - before);
        const unsigned char *tr = p.queue + ((size_t)sh * p.qcap + recno) * task_rec_bytes<G>();
        if (lane == 0) ++stat->tasks;
        unsigned char *rec = p.arena + (size_t)(uint32_t)uni((int)reinterpret_cast<const TaskRec *>(tr)->rec16) * 16;
        Walk<G> w;
        if (prepare_walk<G>(p, lds, ws, tr, rec, w)) run_job<G>(p, lds, ws, w, rec, (uint32_t)((rec - p.arena) >> 4), true, wave_id, stat);
    }
    wave_sync();
    if (lane0 == 0) flush_wave_stats(p, stat, wave_id, __builtin_amdgcn_s_memtime() - t_start);
}

// Scores of the ligands whose tree was split: mean over conformers of the combined maxima (graph_match.py:109).
template <int G>
__global__ void finalize_kernel(const ScreenParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(p.ctl->heavy_count, p.list_cap)) return;
    const unsigned char *rec = p.arena + (size_t)p.heavy_list[i] * 16;
    const RecHeader *H = reinterpret_cast<const RecHeader *>(rec);
    const unsigned long long *best = reinterpret_cast<const unsigned long long *>(rec + sizeof(RecHeader));
    const int C = (int)H->C;
    double sum = 0.0;
    for (int c = 0; c < C; ++c) sum += __longlong_as_double((long long)best[c]);
    put_score(p, H->lig, sum / (double)C);
}

// In front of an arena pass over the ligands the last one had no room for: every subtree of the super-chunk is done and
// finalize has run, so the arena and the task queue start empty again.
__global__ void retry_prep_kernel(Ctl *ctl, uint32_t slot_out) {
    const int lane = threadIdx.x & 63;
    ctl->q_res[lane] = 0;
    ctl->round_lo[lane] = 0;
    ctl->round_hi[lane] = 0;
    if (lane == 0) {
        ctl->arena_top = 0;
        ctl->heavy_count = 0;
        ctl->cursor[3] = 0;
        ctl->retry_count[slot_out] = 0;
    }
}

// Start of a super-chunk: cursors, lists, arena and queue are empty again (the statistics survive unless asked).
__global__ void ctl_clear_kernel(Ctl *ctl, int clear_stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t head_words = offsetof(Ctl, stats) / 4, all_words = sizeof(Ctl) / 4;
    uint32_t *w = reinterpret_cast<uint32_t *>(ctl);
    if (i < head_words) {
        if (i != offsetof(Ctl, qflag) / 4) w[i] = 0;
        else if (clear_stats) w[i] = 0;
    } else if (i < all_words && clear_stats) {
        w[i] = 0;
    }
}

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

} // namespace PMX_NS

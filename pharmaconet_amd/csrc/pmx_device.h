// pmx_device.h - device-side layouts shared by the kernels of libpmx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pmx.h"

namespace pmx {

// Device copy of one pharmacophore model. `edge[m * Nm + n]` = {mean, s, T, std}:
//   s = sqrt(0.5 * log2(e)) / std     so that exp(-0.5 z^2) = exp2(-(|d - mean| * s)^2)
//   T = the largest float with  fl(T / std) < 2  (fl = float32 division, round to nearest even),
//       so `|d - mean| <= T` is bit-for-bit the reference's `abs((d - mean) / std) < 2.0`
//       (src/pmnet/scoring/match_utils.py:55-57) without a division in the inner loop.
struct DevModel {
    int32_t Nm, K;
    int32_t symmetric;        // edge[m][n] == edge[n][m] bit for bit (distances are; checked when the model is created)
    int32_t pad_;
    const float4 *edge;       // [Nm * Nm]
    const uint8_t *node_type; // [64]
    const uint64_t *cnodes;   // [64]   model cluster -> node set
    const uint64_t *tnodes;   // [128]  ligand node type mask -> model nodes of any of those types
    const uint64_t *tclus;    // [128]  ligand cluster type mask -> model clusters sharing a type (graph_match.py:130-134)
    const float2 *cpair;      // [K * K] {float32(|center_a - center_b|), float32(size_a + size_b)}  (graph_match.py:263-265)
    // [K * 128] nodes of cluster a compatible with ligand type mask t, as a list: byte 0 = count (0xff: more than
    // 12, use the masks), bytes 1..12 = node numbers ascending (the reference's order, graph_match.py:148-150)
    const uint4 *clist;
    // [K * 128][2] the same lists as 16-bit byte offsets into a row of the staged edge table (node * 16): word 0 = count
    // (0xffff: more than 12), words 1..12 = offsets ascending
    const uint4 *olist;
};

struct DevLibrary {
    uint64_t n;
    const uint64_t *offsets;
    const uint8_t *data;
};

struct Weights {
    float w[PMX_NUM_TYPES];
};

// Header of one ligand's pair-score table block in the scratch arena (all offsets derive from it).
//   V : vmask_t [T]           conformer-validity mask of each pair entry (bit c <=> P[.][c] > 0, tree.py:81)
//   S : float   [ksumtot][G]  self table  (match_utils.py:77-122)
//   P : float   [T][G]        pair table  (match_utils.py:9-74), -1 where invalid
//   R : double  [nl + 1][G]   R[f][c] = upper bound on what levels f.. can still add to conformer c's total
//                             (bounds_kernel; lets the walker drop subtrees that cannot raise the maximum)
// Pair entry (i, a, j, b), i < j: rowbase[i] + k[i] * (ksum[j] - ksum[i + 1]) + a * k[j] + b.
struct TabHeader {
    uint32_t nl;      // number of tree levels (ligand clusters kept, <= 20)
    uint32_t T;       // number of pair entries
    uint32_t ksumtot; // number of self entries
    uint32_t pad;
    uint8_t k[32];         // candidates per level
    uint16_t ksum[24];     // exclusive prefix sums of k
    uint32_t rowbase[20];
};
static_assert(sizeof(TabHeader) == 176, "TabHeader layout");

template <int G>
using vmask_t = std::conditional_t<(G <= 8), uint8_t,
                                   std::conditional_t<(G <= 16), uint16_t, std::conditional_t<(G <= 32), uint32_t, uint64_t>>>;

__host__ __device__ inline uint64_t round16(uint64_t x) { return (x + 15) & ~uint64_t(15); }

template <int G>
__host__ __device__ inline uint64_t table_bytes(uint32_t T, uint32_t ksumtot, uint32_t nl) {
    return sizeof(TabHeader) + round16(uint64_t(T) * sizeof(vmask_t<G>)) + round16(uint64_t(ksumtot) * G * 4) +
           round16(uint64_t(T) * G * 4) + uint64_t(nl + 1) * G * 8;
}

// A ligand record of the packed library (pharmaconet_amd/library.py).
struct Record {
    int n, C, ncl;
    const uint8_t *typemask;
    const uint8_t *cluster_end;
    const float *xyz; // [n][3][C]
};

__device__ inline Record parse_record(const uint8_t *rec) {
    Record r;
    const uint16_t *h = reinterpret_cast<const uint16_t *>(rec);
    r.n = h[0];
    r.C = h[1];
    r.ncl = h[2];
    r.typemask = rec + 8;
    r.cluster_end = rec + 8 + r.n;
    uint32_t off = (8u + uint32_t(r.n) + uint32_t(r.ncl) + 3u) & ~3u;
    r.xyz = reinterpret_cast<const float *>(rec + off);
    return r;
}

__device__ inline bool record_supported(const Record &r) {
    return r.C >= 1 && r.C <= PMX_MAX_CONFORMERS && r.n <= PMX_MAX_LIGAND_NODES && r.ncl <= PMX_MAX_LIGAND_CLUSTERS;
}

} // namespace pmx

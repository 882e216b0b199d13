// pmx_device.h - device-side layouts shared by the kernels of libpmx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    const uint8_t *node_type; // [PMX_MAX_MODEL_NODES]
    const uint64_t *tclus;    // [128][2] ligand cluster type mask -> model clusters sharing a type (graph_match.py:130-134), 128 bits
    const float2 *cpair;      // [K * K] {float32(|center_a - center_b|), float32(size_a + size_b)}  (graph_match.py:263-265)
    const float2 *cwin;       // [K * K] {lo, hi}: the hull of the 2-sigma pass windows of every node pair (m in a, n in b) - a ligand
                              // node pair at a distance outside it fails the majority test of match_utils.py:55-61 against every
                              // pair of node subsets of the two clusters (exact float ends, as `T` above; {+inf, -inf}: no window)
};

struct DevLibrary {
    uint64_t n;
    const uint64_t *offsets;
    const uint8_t *data;
};

struct Weights {
    float w[PMX_NUM_TYPES];
};

__host__ __device__ inline uint64_t round16(uint64_t x) { return (x + 15) & ~uint64_t(15); }

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

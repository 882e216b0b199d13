// pmx_topk.hip - the ranking step of screening.py:70 on the device: the k best scores in descending
// order, ties in ascending index order (Python's sort is stable and the input is in library order).
// A stable descending radix sort of (score, index) pairs (hipCUB) followed by a k-element copy; this is
// not on the hot path (one pass over 4 bytes per ligand against the kilobytes the scoring reads).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>

#include "pmx.h"

int pmx_topk_fail(int code, const char *msg); // defined in pmx_api.hip

namespace {

__global__ void topk_prepare(const float *scores, const uint64_t *index, uint64_t n, uint64_t base, float *keys, uint64_t *vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = scores[i];
    keys[i] = (s != s) ? -INFINITY : s; // NaN (unsupported ligand) ranks last
    vals[i] = index ? index[i] : base + i;
}

__global__ void topk_emit(const float *keys, const uint64_t *vals, uint64_t n, int k, float *out_scores, uint64_t *out_index) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    if ((uint64_t)i < n) {
        out_scores[i] = keys[i];
        out_index[i] = vals[i];
    } else {
        out_scores[i] = -INFINITY;
        out_index[i] = UINT64_MAX;
    }
}

} // namespace

#define TK_CHECK(expr)                                                        \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            if (buf) (void)hipFree(buf);                                      \
            return pmx_topk_fail(PMX_ERR_HIP, hipGetErrorString(e_));         \
        }                                                                     \
    } while (0)

extern "C" int pmx_topk(const float *scores_dev, const uint64_t *index_dev, uint64_t n, uint64_t base_index, int k,
                        float *out_scores_dev, uint64_t *out_index_dev, int device, void *stream_) {
    unsigned char *buf = nullptr;
    if (k < 0 || (!scores_dev && n) || (k && (!out_scores_dev || !out_index_dev))) return pmx_topk_fail(PMX_ERR_INVALID, "bad top-k argument");
    if (n > (uint64_t)INT32_MAX) return pmx_topk_fail(PMX_ERR_INVALID, "top-k over more than 2^31 - 1 scores: shard the library");
    if (k == 0) return PMX_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TK_CHECK(hipSetDevice(device));
    if (n == 0) {
        topk_emit<<<dim3((k + 255) / 256), dim3(256), 0, stream>>>(nullptr, nullptr, 0, k, out_scores_dev, out_index_dev);
        TK_CHECK(hipGetLastError());
        return PMX_OK;
    }
    size_t temp_bytes = 0;
    TK_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_bytes, (const float *)nullptr, (float *)nullptr,
                                                          (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)n, 0, 32, stream));
    const size_t kb = ((n * 4 + 255) / 256) * 256, vb = ((n * 8 + 255) / 256) * 256;
    TK_CHECK(hipMalloc((void **)&buf, 2 * kb + 2 * vb + temp_bytes + 256));
    float *keys_in = reinterpret_cast<float *>(buf), *keys_out = reinterpret_cast<float *>(buf + kb);
    uint64_t *vals_in = reinterpret_cast<uint64_t *>(buf + 2 * kb), *vals_out = reinterpret_cast<uint64_t *>(buf + 2 * kb + vb);
    void *temp = buf + 2 * kb + 2 * vb;
    topk_prepare<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(scores_dev, index_dev, n, base_index, keys_in, vals_in);
    TK_CHECK(hipGetLastError());
    TK_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 32, stream));
    topk_emit<<<dim3((k + 255) / 256), dim3(256), 0, stream>>>(keys_out, vals_out, n, k, out_scores_dev, out_index_dev);
    TK_CHECK(hipGetLastError());
    TK_CHECK(hipStreamSynchronize(stream));
    (void)hipFree(buf);
    return PMX_OK;
}

// pmx_topk.hip - the ranking step of screening.py:70 on the device: the k best scores in descending
// order, ties in ascending index order (Python's sort is stable and the input is in library order).
// A stable descending radix sort of (score, index) pairs (rocPRIM's device radix sort through its hipCUB front end)
// followed by a k-element copy, in a cached workspace: stream-ordered, no allocation or synchronisation per call.
// Not on the hot path (one pass over 4 bytes per ligand against the kilobytes the scoring reads). Also here: the RCCL
// exchange of per-rank top-k lists (pmx_topk_allgather).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

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
        if (e_ != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

// Sort workspace per (device, stream): grown on demand, never freed, so that a call is stream-ordered work only - no
// allocation, no synchronisation (the previous owner of the buffer is earlier work on the same stream).
namespace {
struct TopkWs {
    unsigned char *buf = nullptr;
    size_t bytes = 0;
};
std::map<std::pair<int, hipStream_t>, TopkWs> g_topk_ws;
std::mutex g_topk_mu;
} // namespace

extern "C" int pmx_topk(const float *scores_dev, const uint64_t *index_dev, uint64_t n, uint64_t base_index, int k,
                        float *out_scores_dev, uint64_t *out_index_dev, int device, void *stream_) {
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
    const size_t need = 2 * kb + 2 * vb + temp_bytes + 256;
    TopkWs *ws;
    {
        std::lock_guard<std::mutex> lock(g_topk_mu);
        ws = &g_topk_ws[std::make_pair(device, stream)];
    }
    if (ws->bytes < need) { // growth is the only time the call waits for the stream
        if (ws->buf) {
            TK_CHECK(hipStreamSynchronize(stream));
            (void)hipFree(ws->buf);
            ws->buf = nullptr;
            ws->bytes = 0;
        }
        TK_CHECK(hipMalloc((void **)&ws->buf, need + need / 4));
        ws->bytes = need + need / 4;
    }
    unsigned char *buf = ws->buf;
    float *keys_in = reinterpret_cast<float *>(buf), *keys_out = reinterpret_cast<float *>(buf + kb);
    uint64_t *vals_in = reinterpret_cast<uint64_t *>(buf + 2 * kb), *vals_out = reinterpret_cast<uint64_t *>(buf + 2 * kb + vb);
    void *temp = buf + 2 * kb + 2 * vb;
    topk_prepare<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(scores_dev, index_dev, n, base_index, keys_in, vals_in);
    TK_CHECK(hipGetLastError());
    TK_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 32, stream));
    topk_emit<<<dim3((k + 255) / 256), dim3(256), 0, stream>>>(keys_out, vals_out, n, k, out_scores_dev, out_index_dev);
    TK_CHECK(hipGetLastError());
    return PMX_OK;
}

int pmx_topk_release(int device) {
    std::lock_guard<std::mutex> lock(g_topk_mu);
    for (auto it = g_topk_ws.begin(); it != g_topk_ws.end();) {
        if (it->first.first == device) {
            if (it->second.buf) (void)hipFree(it->second.buf);
            it = g_topk_ws.erase(it);
        } else {
            ++it;
        }
    }
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------ RCCL exchange
// The one data-path collective of a sharded screen (screening.py:66-70 done by ranks instead of a process pool): every
// rank's k best (score, global index) pairs are all-gathered over RCCL (xGMI inside a node) and merged identically on
// every rank with the ranking rule of screening.py:70.
struct pmx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    float *gs = nullptr;     // [nranks * kcap]
    uint64_t *gi = nullptr;
    int kcap = 0;
};

extern "C" int pmx_comm_unique_id(char id_out[PMX_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= PMX_COMM_ID_BYTES, "PMX_COMM_ID_BYTES too small");
    if (!id_out) return pmx_topk_fail(PMX_ERR_INVALID, "null id");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return pmx_topk_fail(PMX_ERR_HIP, "ncclGetUniqueId failed");
    std::memset(id_out, 0, PMX_COMM_ID_BYTES);
    std::memcpy(id_out, &id, sizeof(id));
    return PMX_OK;
}

extern "C" int pmx_comm_create(const char id[PMX_COMM_ID_BYTES], int rank, int nranks, int device, pmx_comm **out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return pmx_topk_fail(PMX_ERR_INVALID, "bad communicator argument");
    TK_CHECK(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    pmx_comm *c = new pmx_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return pmx_topk_fail(PMX_ERR_HIP, ncclGetErrorString(r));
    }
    *out = c;
    return PMX_OK;
}

extern "C" int pmx_comm_destroy(pmx_comm *c) {
    if (!c) return PMX_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->gs) (void)hipFree(c->gs);
    if (c->gi) (void)hipFree(c->gi);
    delete c;
    return PMX_OK;
}

extern "C" int pmx_topk_allgather(pmx_comm *c, const float *scores_k_dev, const uint64_t *index_k_dev, int k, float *out_scores_dev,
                                  uint64_t *out_index_dev, void *stream_) {
    if (!c || k < 0 || (k && (!scores_k_dev || !index_k_dev || !out_scores_dev || !out_index_dev))) return pmx_topk_fail(PMX_ERR_INVALID, "bad argument");
    if (k == 0) return PMX_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TK_CHECK(hipSetDevice(c->device));
    if (c->kcap < k) {
        if (c->gs) {
            TK_CHECK(hipStreamSynchronize(stream));
            (void)hipFree(c->gs);
            (void)hipFree(c->gi);
            c->gs = nullptr;
            c->gi = nullptr;
        }
        TK_CHECK(hipMalloc((void **)&c->gs, (size_t)c->nranks * k * sizeof(float)));
        TK_CHECK(hipMalloc((void **)&c->gi, (size_t)c->nranks * k * sizeof(uint64_t)));
        c->kcap = k;
    }
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclAllGather(scores_k_dev, c->gs, (size_t)k, ncclFloat32, c->comm, stream);
    if (r == ncclSuccess) r = ncclAllGather(index_k_dev, c->gi, (size_t)k, ncclUint64, c->comm, stream);
    if (r == ncclSuccess) r = ncclGroupEnd();
    if (r != ncclSuccess) return pmx_topk_fail(PMX_ERR_HIP, ncclGetErrorString(r));
    // ranks hold contiguous ascending shards, so position order in the gathered array is global index order: the
    // stable descending sort keeps ties in ascending index order (padding: -inf / UINT64_MAX sorts last)
    return pmx_topk(c->gs, c->gi, (uint64_t)c->nranks * (uint64_t)k, 0, k, out_scores_dev, out_index_dev, c->device, stream_);
}

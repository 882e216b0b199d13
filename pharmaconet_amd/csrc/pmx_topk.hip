// pmx_topk.hip - the ranking step of screening.py:70 on the device: the k best scores in descending
// order, ties in ascending index order (Python's sort is stable and the input is in library order).
//
// A radix *select*, not a sort of all n: every score maps to a 32-bit rank key (smaller = better; NaN - an unsupported
// ligand - after every real score, padding after that), and the k-th smallest (key, position) pair is found digit by digit:
// three histogram passes over the keys (11 + 11 + 10 bits) give the threshold key, three more over the positions of the
// elements that tie on it give the last position taken; one pass then copies exactly the k winners, and one workgroup
// sorts those k. Seven reads of 4 bytes per ligand and a sort of k pairs instead of a radix sort of n pairs; no library,
// no host synchronisation: the digits are chosen by single-workgroup kernels on the device. Also here: the RCCL exchange
// of per-rank top-k lists (pmx_topk_allgather).
#include <hip/hip_runtime.h>
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

constexpr int kBins = 2048;
constexpr int kMaxK = 1 << 16;

struct SelState {
    uint32_t prefix, mask; // digits chosen so far (of the key, then of the position)
    uint32_t need;         // rank of the wanted element among those that match the prefix (1-based)
    uint32_t T, P;         // results: threshold key, last position taken among the elements with key == T
    uint32_t count;        // winners copied
    uint32_t pad[2];
};

// Rank key of a score: ascending key = descending score. Real scores (incl. -inf) < NaN (0xfffffffe) < padding (0xffffffff).
__device__ inline uint32_t rank_key(float s, bool padding) {
    if (padding) return 0xffffffffu;
    if (s != s) return 0xfffffffeu;
    uint32_t u = __float_as_uint(s);
    u = (u >> 31) ? ~u : (u | 0x80000000u); // ascending with the float
    return ~u;
}

__global__ void sel_init(SelState *st, uint32_t *hist, uint32_t k, int take_all) {
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) {
        st->prefix = 0, st->mask = 0, st->need = k, st->count = 0;
        st->T = take_all ? 0xffffffffu : 0u;
        st->P = 0xffffffffu;
    }
}

// mode 0: histogram of digit (key >> shift) over the elements whose key matches the prefix; mode 1: the same over the positions
// of the elements with key == T.
__global__ void sel_hist(const float *scores, const uint64_t *index, uint32_t n, const SelState *st, uint32_t *hist, int shift, int bits, int mode) {
    __shared__ uint32_t h[kBins];
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t prefix = st->prefix, mask = st->mask, T = st->T, dm = (1u << bits) - 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = rank_key(scores[i], index && index[i] == UINT64_MAX);
        const uint32_t v = mode == 0 ? key : (uint32_t)i;
        if (mode == 1 && key != T) continue;
        if ((v & mask) != prefix) continue;
        atomicAdd(&h[(v >> shift) & dm], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// One workgroup: the digit whose bin holds the element of rank `need`; the histogram is cleared for the next pass.
__global__ void sel_scan(SelState *st, uint32_t *hist, int shift, int bits, int last_of_key, int last_of_pos) {
    __shared__ uint32_t part[kBins];
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) part[i] = i < nb ? hist[i] : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t need = st->need, run = 0;
        int b = nb - 1;
        for (int i = 0; i < nb; ++i) {
            if (run + part[i] >= need) {
                b = i;
                break;
            }
            run += part[i];
        }
        st->need = need - run;
        st->prefix |= (uint32_t)b << shift;
        st->mask |= ((1u << bits) - 1u) << shift;
        if (last_of_key) { // the key is complete: now the positions of the elements that tie on it
            st->T = st->prefix;
            st->prefix = 0, st->mask = 0;
        }
        if (last_of_pos) st->P = st->prefix;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) hist[i] = 0;
}

__global__ void sel_compact(const float *scores, const uint64_t *index, uint32_t n, SelState *st, uint2 *cand, uint32_t cap) {
    const uint32_t T = st->T, P = st->P;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = rank_key(scores[i], index && index[i] == UINT64_MAX);
        if (key < T || (key == T && (uint32_t)i <= P)) {
            const uint32_t slot = atomicAdd(&st->count, 1u);
            if (slot < cap) cand[slot] = make_uint2(key, (uint32_t)i);
        }
    }
}

// One workgroup: bitonic sort of the winners by (key, position), then the output (scores as given - a NaN stays a NaN -
// and the global indices); positions past the input are padding.
__global__ void sel_sort_emit(const float *scores, const uint64_t *index, uint64_t base, const SelState *st, uint2 *cand, uint32_t m /* pow2 >= k */,
                              int k, float *out_scores, uint64_t *out_index) {
    const uint32_t cnt = min(st->count, m);
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x)
        if (i >= cnt) cand[i] = make_uint2(0xffffffffu, 0xffffffffu);
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < m / 2; t += blockDim.x) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint2 a = cand[lo], b = cand[hi];
                const unsigned long long ka = ((unsigned long long)a.x << 32) | a.y, kb = ((unsigned long long)b.x << 32) | b.y;
                if ((ka > kb) == up) {
                    cand[lo] = b;
                    cand[hi] = a;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        if ((uint32_t)i < cnt) {
            const uint32_t pos = cand[i].y;
            out_scores[i] = scores[pos];
            out_index[i] = index ? index[pos] : base + pos;
        } else {
            out_scores[i] = -INFINITY;
            out_index[i] = UINT64_MAX;
        }
    }
}

} // namespace

#define TK_CHECK(expr)                                                        \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

// Workspace per (device, stream), of fixed size (it depends on the largest k only), allocated once under the lock; the
// lock is held while a call enqueues, so two host threads ranking on one stream are serialised like the stream itself.
namespace {
struct TopkWs {
    unsigned char *buf = nullptr;
};
constexpr size_t kWsBytes = 256 + kBins * 4 + (size_t)kMaxK * 8;
std::map<std::pair<int, hipStream_t>, TopkWs> g_topk_ws;
std::mutex g_topk_mu;
} // namespace

extern "C" int pmx_topk(const float *scores_dev, const uint64_t *index_dev, uint64_t n, uint64_t base_index, int k,
                        float *out_scores_dev, uint64_t *out_index_dev, int device, void *stream_) {
    if (k < 0 || (!scores_dev && n) || (k && (!out_scores_dev || !out_index_dev))) return pmx_topk_fail(PMX_ERR_INVALID, "bad top-k argument");
    if (n > (uint64_t)INT32_MAX) return pmx_topk_fail(PMX_ERR_INVALID, "top-k over more than 2^31 - 1 scores: shard the library");
    if (k > kMaxK) return pmx_topk_fail(PMX_ERR_INVALID, "top-k: k above 65536");
    if (k == 0) return PMX_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TK_CHECK(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_topk_mu);
    TopkWs &ws = g_topk_ws[std::make_pair(device, stream)];
    if (!ws.buf) TK_CHECK(hipMalloc((void **)&ws.buf, kWsBytes));
    SelState *st = reinterpret_cast<SelState *>(ws.buf);
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws.buf + 256);
    uint2 *cand = reinterpret_cast<uint2 *>(ws.buf + 256 + kBins * 4);
    uint32_t m = 2;
    while (m < (uint32_t)k) m <<= 1;
    const uint32_t n32 = (uint32_t)n;
    const bool take_all = n <= (uint64_t)k;
    sel_init<<<dim3(1), dim3(256), 0, stream>>>(st, hist, (uint32_t)k, take_all ? 1 : 0);
    const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (n + 1023) / 1024 + 1);
    if (!take_all) {
        const int kshift[3] = {21, 10, 0}, kbits[3] = {11, 11, 10};
        for (int d = 0; d < 3; ++d) {
            sel_hist<<<dim3(blocks), dim3(256), 0, stream>>>(scores_dev, index_dev, n32, st, hist, kshift[d], kbits[d], 0);
            sel_scan<<<dim3(1), dim3(256), 0, stream>>>(st, hist, kshift[d], kbits[d], d == 2, 0);
        }
        const int pshift[3] = {21, 10, 0}, pbits[3] = {10, 11, 10}; // positions are below 2^31
        for (int d = 0; d < 3; ++d) {
            sel_hist<<<dim3(blocks), dim3(256), 0, stream>>>(scores_dev, index_dev, n32, st, hist, pshift[d], pbits[d], 1);
            sel_scan<<<dim3(1), dim3(256), 0, stream>>>(st, hist, pshift[d], pbits[d], 0, d == 2);
        }
    }
    if (n) sel_compact<<<dim3(blocks), dim3(256), 0, stream>>>(scores_dev, index_dev, n32, st, cand, m);
    sel_sort_emit<<<dim3(1), dim3(1024), 0, stream>>>(scores_dev, index_dev, base_index, st, cand, m, k, out_scores_dev, out_index_dev);
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

// What RCCL itself says about the communicator (not what the caller passed in): ranks, this rank, its device.
extern "C" int pmx_comm_info(pmx_comm *c, int *rank_out, int *nranks_out, int *device_out) {
    if (!c || !c->comm) return pmx_topk_fail(PMX_ERR_INVALID, "null communicator");
    int n = 0, r = 0, d = 0;
    ncclResult_t e = ncclCommCount(c->comm, &n);
    if (e == ncclSuccess) e = ncclCommUserRank(c->comm, &r);
    if (e == ncclSuccess) e = ncclCommCuDevice(c->comm, &d);
    if (e != ncclSuccess) return pmx_topk_fail(PMX_ERR_HIP, ncclGetErrorString(e));
    if (rank_out) *rank_out = r;
    if (nranks_out) *nranks_out = n;
    if (device_out) *device_out = d;
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
    // ranks hold contiguous ascending shards, so position order in the gathered array is global index order: ties on the score
    // go by position = by global index; NaN entries (unsupported ligands, real indices) rank after every real score, and
    // padding entries (index UINT64_MAX, from ranks with fewer than k ligands) after those
    return pmx_topk(c->gs, c->gi, (uint64_t)c->nranks * (uint64_t)k, 0, k, out_scores_dev, out_index_dev, c->device, stream_);
}

// pmx_density.hip - voxel components of hotspot density maps on gfx950: the heavy half of the model-side graph build
// (SURVEY.md section 8, row f3).
//
// Reference: DensityMapGraph.__extract_pharmacophores (src/pmnet/utils/density_map.py:78-110) - the 26-connected components of
// `mask > 0` on a 64^3 grid per hotspot, each found by a breadth-first search in Python: a component's voxels are listed in the
// order the search discovers them (the queue is the member list itself; a voxel's 26 neighbours are tried in
// itertools.product((-1, 0, 1), repeat=3) order), and that order is the order of the float64 sums of the density-weighted
// centroid (:211-213). The seed of a search is `set.pop()` on a CPython set of voxel tuples: which voxel that is follows from the
// interpreter's hash table and stays on the host (pharmaconet_amd/model_builder.py), at C speed. The device does the two things
// that cost the reference a Python loop over 26 neighbours per voxel:
//   * dm_label_kernel: which component a voxel belongs to - minimum-label propagation over the active voxels of a map with
//     pointer jumping, one workgroup per map, until a sweep changes nothing (label = the smallest linear index of the
//     component). The host then knows the component of a popped seed and removes all its voxels from the set at once.
//   * dm_order_kernel: the breadth-first discovery order of a component from its seed, level by level, one workgroup per
//     component: every voxel of the frontier (in queue order) claims its undiscovered neighbours with atomicMin(queue position x
//     32 + neighbour number) - the sequential search gives a voxel to the EARLIEST queue entry that sees it, and among the
//     children of one entry the neighbour order decides - then a prefix sum over the claims per frontier voxel places the
//     next level. The member list that comes out is, element for element, the reference's `cluster` list.
// Component labels are independent of the search order; the order kernel's output is checked against the host search on the
// fixture maps and on random blobs (tests/test_model_builder.py, -m gpu) and the model state it leads to against the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <vector>

#include "pmx.h"

int pmx_topk_fail(int code, const char *msg); // sets the thread's error text; defined in pmx_api.hip
#include <cstdarg>
#include <cstdio>
static int fail(int code, const char *fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return pmx_topk_fail(code, buf);
}

#define DM_HIPCHECK(call)                                                                                        \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) return fail(PMX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));          \
    } while (0)

namespace {

// Reads / writes that other wavefronts of the workgroup must see although they may sit behind the CU's vector L1: device scope.
__device__ inline int32_t ld(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kLabelThreads = 1024;
constexpr int kOrderThreads = 256;

// labels[v] = smallest linear index of v's 26-connected component of maps > 0 (-1 where maps <= 0). One workgroup per map.
// `active` [n_maps][V]: scratch list of the map's voxels with maps > 0.
__global__ __launch_bounds__(kLabelThreads) void dm_label_kernel(const float *maps, int S, int32_t *labels, int32_t *active) {
    const int V = S * S * S;
    const float *mk = maps + (size_t)blockIdx.x * V;
    int32_t *lab = labels + (size_t)blockIdx.x * V;
    int32_t *act = active + (size_t)blockIdx.x * V;
    __shared__ int n_active, changed;
    if (threadIdx.x == 0) n_active = 0;
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += kLabelThreads) {
        const bool on = mk[v] > 0.f;
        st(lab + v, on ? v : -1);
        if (on) act[atomicAdd(&n_active, 1)] = v;
    }
    __syncthreads();
    const int n = n_active;
    for (;;) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kLabelThreads) {
            const int v = act[k];
            const int x = v / (S * S), y = (v / S) % S, z = v % S;
            const int l = ld(lab + v);
            int m = l;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        const int qx = x + dx, qy = y + dy, qz = z + dz;
                        if ((unsigned)qx >= (unsigned)S || (unsigned)qy >= (unsigned)S || (unsigned)qz >= (unsigned)S) continue;
                        const int ln = ld(lab + (qx * S + qy) * S + qz);
                        if (ln >= 0 && ln < m) m = ln;
                    }
            for (int hop = 0; hop < 8; ++hop) { // pointer jumping: the label of my label's voxel is in my component too
                const int r = ld(lab + m);
                if (r >= m) break;
                m = r;
            }
            if (m < l) {
                atomicMin(lab + v, m);
                changed = 1;
            }
        }
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
    }
}

// Exclusive prefix sum of a[0 .. m) in place by one workgroup; *total = the sum. (chunk per thread, then the chunk sums by thread 0)
__device__ inline void block_exclusive_scan(int32_t *a, int m, int *sums, int *total) {
    const int t = threadIdx.x, ch = (m + kOrderThreads - 1) / kOrderThreads;
    const int lo = min(t * ch, m), hi = min(lo + ch, m);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    sums[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < kOrderThreads; ++i) {
            const int x = sums[i];
            sums[i] = run;
            run += x;
        }
        *total = run;
    }
    __syncthreads();
    int run = sums[t];
    for (int i = lo; i < hi; ++i) {
        const int x = a[i];
        a[i] = run;
        run += x;
    }
    __syncthreads();
}

// members[comp_off[c] + r] = linear index of the r-th voxel the reference's breadth-first search from comp_seed[c] discovers
// (density_map.py:93-109). claim [n_maps][V] starts at 0x7f7f7f7f everywhere; counts [total voxels] is scratch. One workgroup
// per component.
__global__ __launch_bounds__(kOrderThreads) void dm_order_kernel(const float *maps, int S, const int32_t *comp_map, const int32_t *comp_seed,
                                                                 const int32_t *comp_off, int32_t *claim, int32_t *members, int32_t *counts) {
    const int V = S * S * S, c = blockIdx.x;
    const float *mk = maps + (size_t)comp_map[c] * V;
    int32_t *cl = claim + (size_t)comp_map[c] * V;
    const int off = comp_off[c], cap = comp_off[c + 1] - off;
    int32_t *mem = members + off, *cnt = counts + off;
    __shared__ int sums[kOrderThreads];
    __shared__ int total;
    if (threadIdx.x == 0) {
        st(mem + 0, comp_seed[c]);
        st(cl + comp_seed[c], -1); // discovered
    }
    __syncthreads();
    int ls = 0, le = 1;
    while (ls < le) {
        const int m = le - ls;
        // 1. every frontier voxel bids for its undiscovered neighbours: the earliest queue position wins, then the neighbour number
        for (int i = threadIdx.x; i < m; i += kOrderThreads) {
            const int p = ld(mem + ls + i);
            const int x = p / (S * S), y = (p / S) % S, z = p % S;
            int o = 0;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        if (dx == 0 && dy == 0 && dz == 0) continue;
                        const int key = i * 32 + o;
                        ++o;
                        const int qx = x + dx, qy = y + dy, qz = z + dz;
                        if ((unsigned)qx >= (unsigned)S || (unsigned)qy >= (unsigned)S || (unsigned)qz >= (unsigned)S) continue;
                        const int q = (qx * S + qy) * S + qz;
                        if (mk[q] > 0.f) atomicMin(cl + q, key); // (a discovered voxel holds -1 and stays so)
                    }
        }
        __syncthreads();
        // 2. how many each frontier voxel won
        for (int i = threadIdx.x; i < m; i += kOrderThreads) {
            const int p = ld(mem + ls + i);
            const int x = p / (S * S), y = (p / S) % S, z = p % S;
            int o = 0, k = 0;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        if (dx == 0 && dy == 0 && dz == 0) continue;
                        const int key = i * 32 + o;
                        ++o;
                        const int qx = x + dx, qy = y + dy, qz = z + dz;
                        if ((unsigned)qx >= (unsigned)S || (unsigned)qy >= (unsigned)S || (unsigned)qz >= (unsigned)S) continue;
                        const int q = (qx * S + qy) * S + qz;
                        if (mk[q] > 0.f && ld(cl + q) == key) ++k;
                    }
            cnt[ls + i] = k;
        }
        __syncthreads();
        // 3. where each one's children start in the next level
        block_exclusive_scan(cnt + ls, m, sums, &total);
        const int added = total;
        if (le + added > cap) break; // (cannot happen when the component sizes come from the labels; never write past the slice)
        // 4. the next level, in queue order
        for (int i = threadIdx.x; i < m; i += kOrderThreads) {
            const int p = ld(mem + ls + i);
            const int x = p / (S * S), y = (p / S) % S, z = p % S;
            int o = 0, run = le + cnt[ls + i];
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        if (dx == 0 && dy == 0 && dz == 0) continue;
                        const int key = i * 32 + o;
                        ++o;
                        const int qx = x + dx, qy = y + dy, qz = z + dz;
                        if ((unsigned)qx >= (unsigned)S || (unsigned)qy >= (unsigned)S || (unsigned)qz >= (unsigned)S) continue;
                        const int q = (qx * S + qy) * S + qz;
                        if (mk[q] > 0.f && ld(cl + q) == key) {
                            st(mem + run, q);
                            ++run;
                            st(cl + q, -1);
                        }
                    }
        }
        __syncthreads();
        ls = le;
        le += added;
    }
}

} // namespace

struct pmx_density {
    int device = 0;
    int32_t n_maps = 0, size = 0;
    float *maps = nullptr;     // device [n_maps][V]
    int32_t *labels = nullptr; // device [n_maps][V]
    int32_t *work = nullptr;   // device [n_maps][V]: active list of the label kernel, then the claims of the order kernel
};

// Uploads `n_maps` density maps (float32 [size][size][size] each, C order: mask[x][y][z]) and labels their components.
// Stands in for the `np.where(mask > 0)` + search of density_map.py:91-110, for all hotspots of a pocket at once.
extern "C" int pmx_density_create(const float *maps_host, int32_t n_maps, int32_t size, int device, pmx_density **out) {
    if (!maps_host || !out || n_maps < 1 || size < 1 || size > 128) return fail(PMX_ERR_INVALID, "pmx_density_create: bad argument");
    DM_HIPCHECK(hipSetDevice(device));
    const size_t V = (size_t)size * size * size, n = V * (size_t)n_maps;
    if (n >= (1ull << 31)) return fail(PMX_ERR_INVALID, "pmx_density_create: %d maps of %d^3 voxels are too many for one call", n_maps, size);
    pmx_density *d = new pmx_density();
    d->device = device, d->n_maps = n_maps, d->size = size;
    if (hipMalloc((void **)&d->maps, n * 4) != hipSuccess || hipMalloc((void **)&d->labels, n * 4) != hipSuccess ||
        hipMalloc((void **)&d->work, n * 4) != hipSuccess) {
        if (d->maps) (void)hipFree(d->maps);
        if (d->labels) (void)hipFree(d->labels);
        if (d->work) (void)hipFree(d->work);
        delete d;
        return fail(PMX_ERR_OOM, "pmx_density_create: out of device memory");
    }
    hipError_t e = hipMemcpy(d->maps, maps_host, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        dm_label_kernel<<<dim3((unsigned)n_maps), dim3(kLabelThreads)>>>(d->maps, size, d->labels, d->work);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(d->maps), (void)hipFree(d->labels), (void)hipFree(d->work);
        delete d;
        return fail(PMX_ERR_HIP, "pmx_density_create: %s", hipGetErrorString(e));
    }
    *out = d;
    return PMX_OK;
}

// labels_out (host, [n_maps * size^3]): the smallest linear index ((x * size + y) * size + z) of each voxel's component, -1 outside.
extern "C" int pmx_density_labels(pmx_density *d, int32_t *labels_out) {
    if (!d || !labels_out) return fail(PMX_ERR_INVALID, "null argument");
    DM_HIPCHECK(hipSetDevice(d->device));
    const size_t n = (size_t)d->size * d->size * d->size * (size_t)d->n_maps;
    DM_HIPCHECK(hipMemcpy(labels_out, d->labels, n * 4, hipMemcpyDeviceToHost));
    return PMX_OK;
}

// The breadth-first member lists of `n_components` components: component c lies in map comp_map[c], is searched from the voxel
// comp_seed[c] (linear index) and has comp_offset[c + 1] - comp_offset[c] voxels; members_out[comp_offset[c] + r] = its r-th
// discovered voxel (linear index), r = 0 being the seed. Every component of a map may be asked for once per call.
extern "C" int pmx_density_order(pmx_density *d, int32_t n_components, const int32_t *comp_map, const int32_t *comp_seed,
                                 const int32_t *comp_offset, int32_t *members_out) {
    if (!d || n_components < 0 || (n_components > 0 && (!comp_map || !comp_seed || !comp_offset || !members_out)))
        return fail(PMX_ERR_INVALID, "pmx_density_order: bad argument");
    if (n_components == 0) return PMX_OK;
    DM_HIPCHECK(hipSetDevice(d->device));
    const size_t V = (size_t)d->size * d->size * d->size;
    const int64_t total = comp_offset[n_components];
    if (comp_offset[0] != 0 || total < 0 || (uint64_t)total > V * (uint64_t)d->n_maps) return fail(PMX_ERR_INVALID, "pmx_density_order: bad offsets");
    for (int32_t c = 0; c < n_components; ++c) {
        if (comp_map[c] < 0 || comp_map[c] >= d->n_maps || comp_seed[c] < 0 || (size_t)comp_seed[c] >= V || comp_offset[c + 1] <= comp_offset[c])
            return fail(PMX_ERR_INVALID, "pmx_density_order: component %d is malformed", c);
    }
    int32_t *dev = nullptr; // [3 * n_components + 1] descriptors | [total] members | [total] counts
    const size_t desc = (size_t)3 * n_components + 1;
    DM_HIPCHECK(hipMalloc((void **)&dev, (desc + 2 * (size_t)total) * 4));
    std::vector<int32_t> host(desc);
    std::memcpy(host.data(), comp_map, (size_t)n_components * 4);
    std::memcpy(host.data() + n_components, comp_seed, (size_t)n_components * 4);
    std::memcpy(host.data() + 2 * (size_t)n_components, comp_offset, ((size_t)n_components + 1) * 4);
    hipError_t e = hipMemcpy(dev, host.data(), desc * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d->work, 0x7f, V * (size_t)d->n_maps * 4); // claims: "nobody yet" = 0x7f7f7f7f
    if (e == hipSuccess) e = hipMemset(dev + desc, 0xff, (size_t)total * 4);      // members not reached stay -1
    if (e == hipSuccess) {
        dm_order_kernel<<<dim3((unsigned)n_components), dim3(kOrderThreads)>>>(d->maps, d->size, dev, dev + n_components, dev + 2 * (size_t)n_components, d->work,
                                                                              dev + desc, dev + desc + total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(members_out, dev + desc, (size_t)total * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(PMX_ERR_HIP, "pmx_density_order: %s", hipGetErrorString(e));
    return PMX_OK;
}

extern "C" int pmx_density_destroy(pmx_density *d) {
    if (!d) return PMX_OK;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->maps);
    (void)hipFree(d->labels);
    (void)hipFree(d->work);
    delete d;
    return PMX_OK;
}

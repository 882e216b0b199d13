// pmx_api.hip - host side of libpmx.so: the C ABI of include/pmx.h over the kernels of pmx_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <cstddef>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include <time.h>

#include "pmx.h"
#include "pmx_kernels.hip"
#include "pmx_screen.hip"

using namespace pmx;

// ------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static thread_local pmx_score_stats g_stats = {};
static int g_profiling = 0;
struct ScreenWs;
static thread_local ScreenWs *g_last_screen = nullptr;
static thread_local int g_last_device = 0;
static int screen_stats(pmx_score_stats *out);

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHECK(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? PMX_ERR_OOM : PMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                            \
    } while (0)

extern "C" const char *pmx_last_error(void) { return g_err; }
extern "C" int pmx_version(void) { return 100; }
extern "C" int pmx_set_profiling(int enabled) {
    g_profiling = enabled;
    return PMX_OK;
}
extern "C" int pmx_score_stats_get(pmx_score_stats *out) {
    if (!out) return fail(PMX_ERR_INVALID, "null stats");
    *out = g_stats;
    if (g_last_screen) return screen_stats(out);
    return PMX_OK;
}

// -------------------------------------------------------------------------------------- model
struct FnEntry { // tabulated pair functions for one set of type weights
    Weights W;
    FnCell *cells = nullptr;
    hipEvent_t ready = nullptr; // recorded after fn_build_kernel on the stream that built it
    uint64_t stamp = 0;
};

struct pmx_model {
    int device;
    DevModel dm;
    void *blob;
    uint8_t node_type[PMX_MAX_MODEL_NODES]; // host copy
    // node subsets and tabulated pair functions (pmx_screen.hip)
    uint32_t NS = 0, ncell = 0;
    float h = 0.f;
    uint16_t *sidtab = nullptr;  // device [K * 128]
    uint64_t *subnodes = nullptr; // device [NS]
    float2 *win = nullptr;        // device [NS * NS * ncell] exact pass windows
    uint64_t n_complex_cells = 0;
    std::mutex fn_mu;
    std::vector<FnEntry> fn;
    uint64_t fn_stamp = 0;
};

// Largest float T with fl(T / std) < 2 under round-to-nearest-even float32 division: the quotient
// rounds below 2 exactly when T / std < 2 - 2^-24, and std * (2 - 2^-24) is exact in double.
static float pass_threshold(float std) {
    const double bound = (double)std * (2.0 - std::ldexp(1.0, -24));
    float t = (float)bound;
    if ((double)t >= bound) t = std::nextafterf(t, -INFINITY);
    return t;
}


// ---------------------------------------------------------------------------- pair functions: subsets and pass windows
// The floats d >= 0 with |fl(d - mean)| <= T, i.e. abs((d - mean) / std) < 2 in the reference's float32 arithmetic
// (match_utils.py:55-57; T = pass_threshold(std)). fl(d - mean) is monotonic in d, so the set is an interval of floats;
// its ends are found by bisection on the bit patterns (non-negative floats order like their bits).
static bool edge_window(float mean, float T, float &lo, float &hi) {
    auto f32 = [](uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; };
    auto ge = [&](uint32_t b) { volatile float x = f32(b) - mean; return x >= -T; };
    auto le = [&](uint32_t b) { volatile float x = f32(b) - mean; return x <= T; };
    const uint32_t top = 0x7f7fffffu;
    if (!le(0u) || !ge(top)) return false;
    uint32_t a = 0, b = top; // smallest b with ge
    if (ge(0u)) b = 0;
    else {
        while (b - a > 1) {
            const uint32_t m = a + (b - a) / 2;
            if (ge(m)) b = m; else a = m;
        }
    }
    const uint32_t lo_b = b;
    a = 0, b = top; // largest a with le
    if (le(top)) a = top;
    else {
        while (b - a > 1) {
            const uint32_t m = a + (b - a) / 2;
            if (le(m)) a = m; else b = m;
        }
    }
    const uint32_t hi_b = a;
    if (lo_b > hi_b) return false;
    lo = f32(lo_b);
    hi = f32(hi_b);
    return true;
}

static int build_pair_functions(pmx_model *m, const pmx_model_desc *d, const std::vector<uint64_t> &cnodes, const uint64_t *tnodes) {
    const int Nm = m->dm.Nm, K = m->dm.K;
    // node subsets: (model cluster, ligand type mask) -> the cluster's nodes of those types (graph_match.py:148-150)
    std::vector<uint64_t> subs(1, 0ull);
    std::vector<uint16_t> sidtab((size_t)std::max(K, 1) * 128, 0);
    for (int a = 0; a < K; ++a)
        for (int mask = 0; mask < 128; ++mask) {
            const uint64_t nodes = cnodes[a] & tnodes[mask];
            if (!nodes) continue;
            size_t id = 1;
            for (; id < subs.size(); ++id)
                if (subs[id] == nodes) break;
            if (id == subs.size()) subs.push_back(nodes);
            sidtab[(size_t)a * 128 + mask] = (uint16_t)id;
        }
    const uint32_t NS = (uint32_t)subs.size();
    // grid: h = the largest power of two <= std_min / 5 (quintic Hermite error < 3e-8 of the peak, measured), range to mean + 7.5 std
    float std_min = 1e30f, dmax = 1.f;
    for (int i = 0; i < Nm * Nm; ++i) {
        std_min = std::min(std_min, d->edge_std[i]);
        dmax = std::max(dmax, d->edge_mean[i] + 7.5f * d->edge_std[i]);
    }
    if (Nm == 0) std_min = 1.f;
    float h = 0.5f;
    while (h > std_min / 5.f && h > 1.f / 64.f) h *= 0.5f;
    const uint32_t ncell = (uint32_t)std::ceil((double)dmax / (double)h) + 1;
    if (ncell > 16384 || (uint64_t)NS * NS * ncell * sizeof(FnCell) > (2ull << 30))
        return fail(PMX_ERR_INVALID, "pair-function tables of this model would take %llu cells x %u x %u subsets", (unsigned long long)ncell, NS, NS);
    // exact pass window of every model edge
    std::vector<float> wlo((size_t)Nm * Nm), whi((size_t)Nm * Nm);
    std::vector<uint8_t> wok((size_t)Nm * Nm);
    for (int i = 0; i < Nm * Nm; ++i) wok[i] = edge_window(d->edge_mean[i], pass_threshold(d->edge_std[i]), wlo[i], whi[i]) ? 1 : 0;
    const float INF = INFINITY;
    std::vector<float2> win((size_t)NS * NS * ncell);
    uint64_t n_complex = 0;
    std::vector<std::pair<float, int>> ev;
    std::vector<std::pair<float, float>> pass;
    for (uint32_t sa = 0; sa < NS; ++sa)
        for (uint32_t sb = 0; sb < NS; ++sb) {
            float2 *out = win.data() + ((size_t)sa * NS + sb) * ncell;
            const uint64_t A = subs[sa], B = subs[sb];
            if (!A || !B) { // no item: never a fail
                for (uint32_t i = 0; i < ncell; ++i) out[i] = make_float2(-INF, INF);
                continue;
            }
            const int mn = __builtin_popcountll(A) * __builtin_popcountll(B);
            ev.clear();
            for (uint64_t am = A; am; am &= am - 1)
                for (uint64_t bm = B; bm; bm &= bm - 1) {
                    const int e = __builtin_ctzll(am) * Nm + __builtin_ctzll(bm);
                    if (!wok[e]) continue;
                    ev.emplace_back(wlo[e], +1);
                    ev.emplace_back(std::nextafterf(whi[e], INF), -1); // first float after the window
                }
            std::sort(ev.begin(), ev.end());
            pass.clear();
            int cnt = 0;
            bool in = false;
            float start = 0.f;
            for (size_t i = 0; i < ev.size();) {
                const float x = ev[i].first;
                for (; i < ev.size() && ev[i].first == x; ++i) cnt += ev[i].second;
                const bool ok = 2 * cnt >= mn; // num_pass >= num_match * 0.5 (match_utils.py:61)
                if (ok && !in) { in = true; start = x; }
                else if (!ok && in) { in = false; pass.emplace_back(start, std::nextafterf(x, -INF)); }
            }
            if (in) pass.emplace_back(start, INF);
            for (uint32_t i = 0; i < ncell; ++i) {
                const float x0 = (float)i * h, x1 = (i + 1 == ncell) ? INF : (float)(i + 1) * h;
                int hits = 0;
                float2 w = make_float2(INF, -INF); // never passes
                for (const auto &pr : pass)
                    if (pr.first < x1 && pr.second >= x0) {
                        ++hits;
                        w = make_float2(pr.first, pr.second);
                    }
                if (hits > 1) {
                    w = make_float2(NAN, NAN);
                    ++n_complex;
                }
                out[i] = w;
            }
        }
    HIPCHECK(hipMalloc((void **)&m->sidtab, sidtab.size() * 2));
    HIPCHECK(hipMalloc((void **)&m->subnodes, (size_t)NS * 8));
    HIPCHECK(hipMalloc((void **)&m->win, win.size() * sizeof(float2)));
    HIPCHECK(hipMemcpy(m->sidtab, sidtab.data(), sidtab.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m->subnodes, subs.data(), (size_t)NS * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m->win, win.data(), win.size() * sizeof(float2), hipMemcpyHostToDevice));
    m->NS = NS;
    m->ncell = ncell;
    m->h = h;
    m->n_complex_cells = n_complex;
    return PMX_OK;
}

// The tabulated functions for the call's weights: built on `stream` the first time, kept for the last four weight sets.
static int pair_functions(pmx_model *m, const Weights &W, hipStream_t stream, FnTable *out) {
    std::lock_guard<std::mutex> lock(m->fn_mu);
    FnEntry *hit = nullptr;
    for (FnEntry &e : m->fn)
        if (std::memcmp(&e.W, &W, sizeof(W)) == 0) hit = &e;
    if (!hit) {
        if (m->fn.size() < 4) {
            m->fn.emplace_back();
            hit = &m->fn.back();
            HIPCHECK(hipMalloc((void **)&hit->cells, (size_t)m->NS * m->NS * m->ncell * sizeof(FnCell)));
            HIPCHECK(hipEventCreateWithFlags(&hit->ready, hipEventDisableTiming));
        } else { // recycle the least recently used entry once nothing queued still reads it
            hit = &m->fn[0];
            for (FnEntry &e : m->fn)
                if (e.stamp < hit->stamp) hit = &e;
            HIPCHECK(hipDeviceSynchronize());
        }
        hit->W = W;
        fn_build_kernel<<<dim3(m->NS * m->NS), dim3(128), 0, stream>>>(m->dm, W, m->subnodes, m->NS, m->ncell, m->h, m->win, hit->cells);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(hit->ready, stream));
    } else {
        HIPCHECK(hipStreamWaitEvent(stream, hit->ready, 0));
    }
    hit->stamp = ++m->fn_stamp;
    out->cells = hit->cells;
    out->NS = m->NS;
    out->ncell = m->ncell;
    out->inv_h = 1.0f / m->h;
    out->pad = 0;
    return PMX_OK;
}

extern "C" int pmx_model_create(const pmx_model_desc *d, int device, pmx_model **out) {
    if (!d || !out) return fail(PMX_ERR_INVALID, "null argument");
    const int Nm = d->n_nodes, K = d->n_clusters;
    if (Nm < 0 || Nm > PMX_MAX_MODEL_NODES) return fail(PMX_ERR_INVALID, "model has %d nodes (max %d)", Nm, PMX_MAX_MODEL_NODES);
    if (K < 0 || K > PMX_MAX_MODEL_CLUSTERS) return fail(PMX_ERR_INVALID, "model has %d clusters (max %d)", K, PMX_MAX_MODEL_CLUSTERS);
    for (int i = 0; i < Nm; ++i)
        if (d->node_type[i] >= PMX_NUM_TYPES) return fail(PMX_ERR_INVALID, "node %d has type id %d", i, d->node_type[i]);
    HIPCHECK(hipSetDevice(device));

    const size_t n_edge = (size_t)Nm * Nm, n_pair = (size_t)K * K;
    const size_t off_edge = 0;
    const size_t off_type = off_edge + round16(n_edge * sizeof(float4));
    const size_t off_cnodes = off_type + 64;
    const size_t off_tnodes = off_cnodes + 64 * 8;
    const size_t off_tclus = off_tnodes + 128 * 8;
    const size_t off_cpair = off_tclus + 128 * 8;
    const size_t off_clist = off_cpair + round16(n_pair * sizeof(float2));
    const size_t off_olist = off_clist + (size_t)std::max(K, 1) * 128 * 16;
    const size_t total = off_olist + (size_t)std::max(K, 1) * 128 * 32 + 16;
    std::vector<unsigned char> host(total, 0);
    float4 *edge = reinterpret_cast<float4 *>(host.data() + off_edge);
    uint8_t *ntype = host.data() + off_type;
    uint64_t *cnodes = reinterpret_cast<uint64_t *>(host.data() + off_cnodes);
    uint64_t *tnodes = reinterpret_cast<uint64_t *>(host.data() + off_tnodes);
    uint64_t *tclus = reinterpret_cast<uint64_t *>(host.data() + off_tclus);
    float2 *cpair = reinterpret_cast<float2 *>(host.data() + off_cpair);

    const double s_const = std::sqrt(0.5 * 1.4426950408889634074); // sqrt(0.5 * log2(e))
    for (size_t i = 0; i < n_edge; ++i) {
        const float mean = d->edge_mean[i], sd = d->edge_std[i];
        if (!(sd > 0.f)) return fail(PMX_ERR_INVALID, "edge %zu has distance_std %g", i, (double)sd);
        edge[i] = make_float4(mean, (float)(s_const / (double)sd), pass_threshold(sd), sd);
    }
    uint64_t type_nodes[PMX_NUM_TYPES] = {0};
    for (int m = 0; m < Nm; ++m) {
        ntype[m] = d->node_type[m];
        type_nodes[d->node_type[m]] |= 1ull << m;
    }
    for (int a = 0; a < K; ++a) cnodes[a] = d->cluster_nodes[a];
    for (int mask = 0; mask < 128; ++mask) {
        uint64_t nodes = 0, clus = 0;
        for (int t = 0; t < PMX_NUM_TYPES; ++t)
            if (mask >> t & 1) nodes |= type_nodes[t];
        for (int a = 0; a < K; ++a)
            if (d->cluster_typemask[a] & mask) clus |= 1ull << a;
        tnodes[mask] = nodes;
        tclus[mask] = clus;
    }
    for (int a = 0; a < K; ++a)
        for (int mask = 0; mask < 128; ++mask) {
            unsigned char *e = host.data() + off_clist + ((size_t)a * 128 + mask) * 16;
            const uint64_t nodes = cnodes[a] & tnodes[mask];
            const int cnt = __builtin_popcountll(nodes);
            if (cnt > 12) {
                e[0] = 0xff;
                continue;
            }
            e[0] = (unsigned char)cnt;
            for (int q = 1; q <= 12; ++q) e[q] = (unsigned char)Nm; // padding: the neutral column of the staged table
            int q = 1;
            for (int m = 0; m < Nm; ++m)
                if (nodes >> m & 1) e[q++] = (unsigned char)m;
        }
    for (int a = 0; a < K; ++a)
        for (int mask = 0; mask < 128; ++mask) {
            uint16_t *e = reinterpret_cast<uint16_t *>(host.data() + off_olist + ((size_t)a * 128 + mask) * 32);
            const uint64_t nodes = cnodes[a] & tnodes[mask];
            const int cnt = __builtin_popcountll(nodes);
            e[0] = cnt > 12 ? (uint16_t)0xffff : (uint16_t)cnt;
            int q = 1;
            for (int m = 0; m < Nm && cnt <= 12; ++m)
                if (nodes >> m & 1) e[q++] = (uint16_t)(m * 16);
        }
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            const double *ca = d->cluster_center + 3 * a, *cb = d->cluster_center + 3 * b;
            const double dist = std::sqrt((ca[0] - cb[0]) * (ca[0] - cb[0]) + (ca[1] - cb[1]) * (ca[1] - cb[1]) +
                                          (ca[2] - cb[2]) * (ca[2] - cb[2])); // graph_match.py:263-264
            cpair[a * K + b] = make_float2((float)dist, (float)(d->cluster_size[a] + d->cluster_size[b])); // :265
        }

    void *blob = nullptr;
    HIPCHECK(hipMalloc(&blob, total));
    hipError_t e = hipMemcpy(blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return fail(PMX_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    }
    pmx_model *m = new pmx_model();
    m->device = device;
    m->blob = blob;
    std::memcpy(m->node_type, ntype, sizeof(m->node_type));
    unsigned char *b8 = static_cast<unsigned char *>(blob);
    m->dm.Nm = Nm;
    m->dm.K = K;
    m->dm.symmetric = 1;
    for (int a = 0; a < Nm && m->dm.symmetric; ++a)
        for (int b = 0; b < a; ++b)
            if (std::memcmp(&d->edge_mean[a * Nm + b], &d->edge_mean[b * Nm + a], 4) || std::memcmp(&d->edge_std[a * Nm + b], &d->edge_std[b * Nm + a], 4)) {
                m->dm.symmetric = 0;
                break;
            }
    m->dm.pad_ = 0;
    m->dm.edge = reinterpret_cast<const float4 *>(b8 + off_edge);
    m->dm.node_type = b8 + off_type;
    m->dm.cnodes = reinterpret_cast<const uint64_t *>(b8 + off_cnodes);
    m->dm.tnodes = reinterpret_cast<const uint64_t *>(b8 + off_tnodes);
    m->dm.tclus = reinterpret_cast<const uint64_t *>(b8 + off_tclus);
    m->dm.cpair = reinterpret_cast<const float2 *>(b8 + off_cpair);
    m->dm.clist = reinterpret_cast<const uint4 *>(b8 + off_clist);
    m->dm.olist = reinterpret_cast<const uint4 *>(b8 + off_olist);
    {
        std::vector<uint64_t> cn(cnodes, cnodes + 64);
        const int rc = build_pair_functions(m, d, cn, tnodes);
        if (rc != PMX_OK) {
            pmx_model_destroy(m);
            return rc;
        }
    }
    *out = m;
    return PMX_OK;
}

extern "C" int pmx_model_destroy(pmx_model *m) {
    if (!m) return PMX_OK;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->blob);
    if (m->sidtab) (void)hipFree(m->sidtab);
    if (m->subnodes) (void)hipFree(m->subnodes);
    if (m->win) (void)hipFree(m->win);
    for (FnEntry &e : m->fn) {
        if (e.cells) (void)hipFree(e.cells);
        if (e.ready) (void)hipEventDestroy(e.ready);
    }
    delete m;
    return PMX_OK;
}

// ------------------------------------------------------------------------------------ library
struct pmx_library {
    int device;
    DevLibrary dl;
    uint64_t *offsets;
    uint8_t *data;
    pmx_library_info info;
};

extern "C" int pmx_library_upload(const pmx_library_view *v, int device, pmx_library **out) {
    if (!v || !out) return fail(PMX_ERR_INVALID, "null argument");
    if (!v->offsets) return fail(PMX_ERR_INVALID, "null offsets");
    HIPCHECK(hipSetDevice(device));
    const uint64_t n = v->n_ligands;
    const hipMemcpyKind kind = v->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    uint64_t nbytes = 0;
    if (v->on_device) {
        HIPCHECK(hipMemcpy(&nbytes, v->offsets + n, 8, hipMemcpyDeviceToHost));
    } else {
        nbytes = v->offsets[n];
    }
    pmx_library *lib = new pmx_library();
    lib->device = device;
    lib->offsets = nullptr;
    lib->data = nullptr;
    hipError_t e = hipMalloc((void **)&lib->offsets, (n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&lib->data, std::max<uint64_t>(nbytes, 16));
    if (e == hipSuccess) e = hipMemcpy(lib->offsets, v->offsets, (n + 1) * 8, kind);
    if (e == hipSuccess && nbytes) e = hipMemcpy(lib->data, v->data, nbytes, kind);
    unsigned long long *stats_dev = nullptr;
    unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMalloc((void **)&stats_dev, sizeof(stats));
    if (e == hipSuccess) e = hipMemset(stats_dev, 0, sizeof(stats));
    lib->dl.n = n;
    lib->dl.offsets = lib->offsets;
    lib->dl.data = lib->data;
    if (e == hipSuccess && n) {
        library_stats_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(lib->dl, lib->data, nbytes, stats_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(stats, stats_dev, sizeof(stats), hipMemcpyDeviceToHost);
    if (stats_dev) (void)hipFree(stats_dev);
    if (e != hipSuccess) {
        if (lib->offsets) (void)hipFree(lib->offsets);
        if (lib->data) (void)hipFree(lib->data);
        delete lib;
        return fail(e == hipErrorOutOfMemory ? PMX_ERR_OOM : PMX_ERR_HIP, "library upload failed: %s", hipGetErrorString(e));
    }
    if (stats[5]) { // offsets are validated on the device, for host and device views alike
        (void)hipFree(lib->offsets);
        (void)hipFree(lib->data);
        delete lib;
        return fail(PMX_ERR_INVALID, "%llu record offsets are not 16-byte aligned, run backwards or point past the data", stats[5]);
    }
    lib->info.n_ligands = n;
    lib->info.n_bytes = nbytes;
    lib->info.total_conformers = stats[0];
    lib->info.max_nodes = (int32_t)stats[1];
    lib->info.max_conformers = (int32_t)stats[2];
    lib->info.max_clusters = (int32_t)stats[3];
    lib->info.n_unsupported = (int32_t)stats[4];
    *out = lib;
    return PMX_OK;
}

extern "C" int pmx_library_info_get(const pmx_library *lib, pmx_library_info *info) {
    if (!lib || !info) return fail(PMX_ERR_INVALID, "null argument");
    *info = lib->info;
    return PMX_OK;
}

extern "C" int pmx_library_destroy(pmx_library *lib) {
    if (!lib) return PMX_OK;
    (void)hipSetDevice(lib->device);
    (void)hipFree(lib->offsets);
    (void)hipFree(lib->data);
    delete lib;
    return PMX_OK;
}

// ---------------------------------------------------------------------------------- workspace
// Chunks are software-pipelined over two buffer slots: while the tree kernels of chunk k run on the caller's
// stream, the table kernels of chunk k + 1 run on an internal side stream. The two phases bind differently
// (tables: VALU / LDS; tree search: memory latency), so their wavefronts share the CUs well.
constexpr size_t kMetaBytes = 1024 + (size_t)kStatShards * 32 + (size_t)kQueueShards * 4; // counters + the tree kernels' sharded statistics + the task queue's tails
constexpr size_t kTailWord = 256 + (size_t)kStatShards * 8;                               // first tail, in 32-bit words

struct Slot {
    uint32_t *units = nullptr;
    int32_t *status = nullptr;
    uint64_t *taboff = nullptr;
    uint8_t *arena = nullptr;
    size_t arena_cap = 0;
    uint32_t *meta = nullptr;      // device: [0] max levels, [1] fetch counter, [2..3] table bytes (u64), [5] queue overflow flag, [14] / [15] ligand cursors of the table kernel, [32..] debug and profiling words; then kStatShards x 4 u64 of tree statistics and the kQueueShards tails of the task queue
    uint32_t *meta_host = nullptr; // pinned mirror
    unsigned long long *bestbuf = nullptr; // [chunk_cap][64] per-conformer maxima of split ligands
    uint8_t *deferred = nullptr;           // [chunk_cap]
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // profiling: sizes | tables | tree start | tier 1 | tasks
    hipEvent_t tables_done = nullptr; // side stream: tables + bounds of the chunk in this slot are written
    hipEvent_t walk_done = nullptr;   // caller's stream: the tree kernels no longer need this slot
    // per-chunk values carried from the table phase to the tree phase
    uint32_t n = 0, max_levels = 0;
    uint64_t lig0 = 0, table_total = 0;
    bool walked = false;
};

struct Workspace {
    uint32_t chunk_cap = 0;
    Slot slot[2];
    uint8_t *queue = nullptr; // task queue of the tree kernels (one chunk walks at a time)
    size_t queue_bytes = 0;
    int num_cu = 0;
    hipStream_t side = nullptr;
    hipStream_t own = nullptr;   // the tree-phase stream of a secondary pipeline (the first one uses the caller's)
    hipEvent_t entry = nullptr;
    hipEvent_t done = nullptr;   // a secondary pipeline has finished its range
    uint32_t lds_attr_set = 0; // bit log2(G): the kernels of that conformer-group width may use all of the CU's LDS on this device
};
static std::map<std::pair<int, int>, Workspace> g_ws; // (device, pipeline)
static std::mutex g_mu;

static uint32_t chunk_size() { // read per call: tests vary it to cut the library differently
    const char *s = std::getenv("PMX_CHUNK");
    const long x = s ? std::atol(s) : 0;
    return (uint32_t)(x > 0 ? std::min<long>(x, 1 << 22) : 262144);
}

static long env_long(const char *name, long dflt) {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    return std::atol(s);
}

static int ensure_workspace(int device, int pipeline, Workspace **out) {
    Workspace &w = g_ws[std::make_pair(device, pipeline)];
    const uint32_t cap = chunk_size();
    if (w.chunk_cap < cap) {
        for (Slot &sl : w.slot) {
            if (sl.units) (void)hipFree(sl.units);
            if (sl.status) (void)hipFree(sl.status);
            if (sl.taboff) (void)hipFree(sl.taboff);
            if (sl.bestbuf) (void)hipFree(sl.bestbuf);
            if (sl.deferred) (void)hipFree(sl.deferred);
            HIPCHECK(hipMalloc((void **)&sl.units, (size_t)cap * 4));
            HIPCHECK(hipMalloc((void **)&sl.status, (size_t)cap * 4));
            HIPCHECK(hipMalloc((void **)&sl.taboff, ((size_t)cap + 1) * 8));
            HIPCHECK(hipMalloc((void **)&sl.bestbuf, (size_t)cap * 64 * 8));
            HIPCHECK(hipMalloc((void **)&sl.deferred, (size_t)cap));
        }
        w.chunk_cap = cap;
    }
    if (!w.side) {
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, device));
        w.num_cu = prop.multiProcessorCount;
        HIPCHECK(hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&w.own, hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&w.entry, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        for (Slot &sl : w.slot) {
            HIPCHECK(hipMalloc((void **)&sl.meta, kMetaBytes));
            HIPCHECK(hipHostMalloc((void **)&sl.meta_host, kMetaBytes));
            for (auto &ev : sl.ev) HIPCHECK(hipEventCreate(&ev));
            HIPCHECK(hipEventCreateWithFlags(&sl.tables_done, hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&sl.walk_done, hipEventDisableTiming));
        }
    }
    if (!w.queue) {
        w.queue_bytes = (size_t)std::max<long>(1, env_long("PMX_TASKQ_MB", 2048)) << 20;
        HIPCHECK(hipMalloc((void **)&w.queue, w.queue_bytes));
    }
    *out = &w;
    return PMX_OK;
}

static constexpr size_t kLdsPerCu = 160 * 1024;

static bool trace_on() {
    static int v = -1;
    if (v < 0) v = std::getenv("PMX_TRACE") ? 1 : 0;
    return v == 1;
}
static double trace_ms() { // milliseconds since the first trace line of the process
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define TRACE(...)                                       \
    do {                                                 \
        if (trace_on()) {                                \
            fprintf(stderr, "[pmx %10.3f] ", trace_ms()); \
            fprintf(stderr, __VA_ARGS__);                \
            fprintf(stderr, "\n");                       \
            fflush(stderr);                              \
        }                                                \
    } while (0)

static constexpr int kRetrySmaller = -100; // internal: the chunk's tables exceed the arena limit

// Table phase of one chunk on stream `q`: sizes -> scan -> (one small device-to-host read) -> pair-score
// tables -> search bounds.
template <int G>
static int table_phase(const pmx_model *model, const pmx_library *lib, const Weights &W, uint64_t lig0, uint32_t n,
                       int32_t *status, Slot &sl, hipStream_t q, int ws_num_cu) {
    const int Nm = model->dm.Nm;
    bool zero_weight = false; // a type with weight 0 among the model's nodes
    for (int m = 0; m < Nm; ++m) zero_weight = zero_weight || W.w[model->node_type[m]] == 0.f;
    if (sl.walked) HIPCHECK(hipStreamWaitEvent(q, sl.walk_done, 0)); // the slot's previous chunk has been walked
    sl.n = n;
    sl.lig0 = lig0;
    {
        const uint64_t words = std::max<uint64_t>((uint64_t)n * G, kMetaBytes / 4);
        clear_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, q>>>(sl.meta, kMetaBytes / 4, sl.bestbuf, (uint64_t)n * G, sl.deferred, n);
    }
    if (g_profiling) HIPCHECK(hipEventRecord(sl.ev[0], q));
    sizes_kernel<G><<<dim3((n + 255) / 256), dim3(256), 0, q>>>(lib->dl, model->dm.tclus, lig0, n, sl.units, status, sl.meta);
    scan_kernel<<<dim3(1), dim3(1024), 0, q>>>(sl.units, n, sl.taboff, reinterpret_cast<uint64_t *>(sl.meta + 2));
    HIPCHECK(hipGetLastError());
    TRACE("chunk at %llu: sizes+scan launched, n=%u", (unsigned long long)lig0, n);
    HIPCHECK(hipMemcpyAsync(sl.meta_host, sl.meta, 1024, hipMemcpyDeviceToHost, q));
    HIPCHECK(hipStreamSynchronize(q));
    sl.max_levels = sl.meta_host[0];
    std::memcpy(&sl.table_total, sl.meta_host + 2, 8);
    TRACE("max_levels=%u table bytes=%llu", sl.max_levels, (unsigned long long)sl.table_total);
    // tree_kernel addresses a ligand's tables with 32 bits of 16-byte units: a chunk's tables must stay below 64 GB
    // (and below PMX_ARENA_MAX_MB); a chunk that would need more is cut smaller by the caller (kRetrySmaller)
    const uint64_t arena_limit = std::min<uint64_t>((1ull << 36) - (1ull << 22), (uint64_t)std::max<long>(64, env_long("PMX_ARENA_MAX_MB", 40960)) << 20);
    if (sl.table_total + sl.table_total / 4 + (1u << 20) > arena_limit) return kRetrySmaller;
    if (sl.table_total > sl.arena_cap) {
        if (sl.arena) (void)hipFree(sl.arena);
        sl.arena = nullptr;
        sl.arena_cap = 0;
        const size_t want = (size_t)(sl.table_total + sl.table_total / 4 + (1u << 20));
        HIPCHECK(hipMalloc((void **)&sl.arena, want));
        sl.arena_cap = want;
    }
    if (g_profiling) HIPCHECK(hipEventRecord(sl.ev[1], q));
    if (sl.table_total > 0) {
        // <= 8 ligands (waves) per block share one staged model table; the 160 KB of LDS always hold at least one
        const size_t model_lds = (size_t)Nm * (Nm + 1) * sizeof(float4) + 64 * 8 + 128 * 8; // edge table + one neutral column
        if (model_lds + tables_v2_wave_bytes<G>(model->dm.K) + 1024 > kLdsPerCu) return fail(PMX_ERR_INVALID, "model tables do not fit LDS");
        const int v2_waves = (int)std::min<size_t>((size_t)env_long("PMX_V2_WAVES", 8), (kLdsPerCu - 1024 - model_lds) / tables_v2_wave_bytes<G>(model->dm.K));
        size_t lds2 = model_lds + (size_t)v2_waves * tables_v2_wave_bytes<G>(model->dm.K);
        lds2 = std::max<size_t>(lds2, (size_t)env_long("PMX_V2_LDS_MIN", 0));
        // persistent blocks: as many as the CUs hold at once (PMX_V2_BLOCKS per CU), fed from the cursor in meta[14]
        const unsigned v2_blocks = (unsigned)std::min<uint64_t>((n + v2_waves - 1) / v2_waves,
                                                                (uint64_t)ws_num_cu * (uint64_t)std::max<long>(1, env_long("PMX_V2_BLOCKS", 4)));
        if (zero_weight)
            tables_kernel_v2<G, true><<<dim3(v2_blocks), dim3(64 * v2_waves), lds2, q>>>(
                model->dm, lib->dl, W, lig0, n, status, sl.taboff, sl.arena, nullptr, nullptr, sl.meta + 14);
        else
            tables_kernel_v2<G, false><<<dim3(v2_blocks), dim3(64 * v2_waves), lds2, q>>>(
                model->dm, lib->dl, W, lig0, n, status, sl.taboff, sl.arena, nullptr, nullptr, sl.meta + 14);
        bounds_kernel<G><<<dim3((n + 3) / 4), dim3(256), 0, q>>>(n, status, sl.taboff, sl.arena,
                                                                 (int)(env_long("PMX_TREE_FLAGS", 0) & 4), nullptr, nullptr);
        HIPCHECK(hipGetLastError());
    }
    if (g_profiling) HIPCHECK(hipEventRecord(sl.ev[2], q));
    HIPCHECK(hipEventRecord(sl.tables_done, q));
    return PMX_OK;
}

// Tree phase of the chunk in `sl` on the caller's stream: one wavefront per ligand, then rounds over the task
// queue (one small device-to-host read per round), then the scores of split ligands.
template <int G>
static int tree_phase(const pmx_model *model, const pmx_library *lib, int32_t *status, float *scores, Slot &sl, Workspace &ws,
                      hipStream_t stream) {
    const uint32_t n = sl.n;
    const uint64_t lig0 = sl.lig0;
    HIPCHECK(hipStreamWaitEvent(stream, sl.tables_done, 0));
    if (g_profiling) HIPCHECK(hipEventRecord(sl.ev[3], stream));
    const int depth = std::max<int>(1, (int)sl.max_levels);
    const int Kc = std::max(1, model->dm.K);
#ifndef PMX_TOT_WINDOW
#define PMX_TOT_WINDOW 0
#endif
    const size_t lds = tree_wave_bytes<G>(depth, Kc) + (size_t)std::min<int>(PMX_TOT_WINDOW, depth + 1) * 64 * 8 +
                       (size_t)std::max<long>(0, env_long("PMX_LDS_PAD", 0));
    if (lds > kLdsPerCu) return fail(PMX_ERR_INVALID, "tree state of %zu bytes does not fit LDS", lds);
    TreeParams tp;
    tp.arena = sl.arena;
    tp.taboff = sl.taboff;
    tp.status = status;
    tp.lib = lib->dl;
    tp.first = lig0;
    tp.count = n;
    tp.qtail = sl.meta + kTailWord;
    tp.qflag = sl.meta + 5;
    tp.queue = ws.queue;
    tp.qcap = (uint32_t)std::min<size_t>(ws.queue_bytes / task_bytes<G>() / kQueueShards, 0x7fffffffu / kQueueShards);
    for (int sh = 0; sh < kQueueShards; ++sh) tp.shard_lo[sh] = tp.shard_hi[sh] = 0;
    tp.bestbuf = sl.bestbuf;
    tp.deferred = sl.deferred;
    tp.depth_cap = depth;
    tp.K = Kc;
    tp.budget = (uint32_t)std::max<long>(64, env_long("PMX_BUDGET", 1024));
    tp.scores = scores;
    tp.step_cap = (uint32_t)std::max<long>(1, env_long("PMX_STEP_CAP", 1 << 20));
    tp.share_levels = (uint32_t)std::max<long>(0, env_long("PMX_SHARE_LEVELS", 1));
    tp.min_levels = (uint32_t)std::max<long>(0, env_long("PMX_MIN_LEVELS", 4));
    tp.flags = (uint32_t)env_long("PMX_TREE_FLAGS", 0);
    tp.nsteps = reinterpret_cast<unsigned long long *>(sl.meta + 256); // kStatShards x 4 words after the 1 KB of counters
    tp.dbg = sl.meta + 32;
    tp.max_iters = (unsigned long long)std::max<long>(1, env_long("PMX_MAXITERS", 1L << 31));
    TRACE("tree kernel: grid=%u lds=%zu depth=%d", n, lds, depth);
    tree_kernel<G, false><<<dim3(n), dim3(64), lds, stream>>>(tp);
    HIPCHECK(hipGetLastError());
    if (g_profiling) HIPCHECK(hipEventRecord(sl.ev[4], stream));
    // rounds over the task queue: walkers that ran over budget appended subtrees
    bool any_task = false;
    for (;;) {
        HIPCHECK(hipMemcpyAsync(sl.meta_host, sl.meta, kMetaBytes, hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        const uint32_t *mh = sl.meta_host;
        unsigned long long tstat[4] = {0, 0, 0, 0}; // the tree statistics, summed over their shards
        for (int sh = 0; sh < kStatShards; ++sh) {
            unsigned long long v[4];
            std::memcpy(v, mh + 256 + 8 * sh, 32);
            tstat[0] += v[0];
            tstat[1] += v[1];
            tstat[2] = std::max(tstat[2], v[2]);
            tstat[3] = std::max(tstat[3], v[3]);
        }
        if (!any_task && g_stats.n_rounds == 0) {
            g_stats.n_steps_first += tstat[0];
        }
        // this round: what every shard of the queue has received since the last one
        uint64_t round_tasks = 0;
        for (int sh = 0; sh < kQueueShards; ++sh) {
            tp.shard_lo[sh] = tp.shard_hi[sh];
            tp.shard_hi[sh] = std::min<uint32_t>(mh[kTailWord + sh], tp.qcap);
            round_tasks += tp.shard_hi[sh] - tp.shard_lo[sh];
        }
        TRACE("round: %llu tasks, steps=%llu iters=%llu", (unsigned long long)round_tasks, tstat[0], tstat[1]);
        if (mh[5]) g_stats.queue_overflow = 1;
        if (mh[32] == 2) return fail(PMX_ERR_INVALID, "tree kernel watchdog fired at location %u", mh[35]);
        if (mh[32]) {
            char buf[400];
            int o = snprintf(buf, sizeof(buf), "tree walk hit the iteration cap (nl=%u busy=%u):", mh[33], mh[34]);
            for (int gq = 0; gq < 8 && o < 380; ++gq) {
                const uint32_t *d = mh + 32 + 16 + gq * 8;
                o += snprintf(buf + o, sizeof(buf) - o, " [g%d li=%u busy=%u f=%d f0=%d sp=%d sfr=%d frm=%08x todo=%x]", gq, d[0], d[1], (int)d[2], (int)d[3], (int)d[4], (int)d[5], d[6], d[7]);
            }
            return fail(PMX_ERR_INVALID, "%s", buf);
        }
        if (round_tasks == 0) {
            g_stats.n_steps += tstat[0];
            g_stats.n_iters += tstat[1];
            g_stats.max_iters_ligand = std::max<uint64_t>(g_stats.max_iters_ligand, tstat[2]);
            g_stats.max_iters_task = std::max<uint64_t>(g_stats.max_iters_task, tstat[3]);
            unsigned long long ns = 0;
            (void)ns;
#ifdef PMX_PROF
            {
                fprintf(stderr, "PMXPROF");
                for (int i = 0; i < 48; ++i) {
                    std::memcpy(&ns, mh + 128 + 2 * i, 8);
                    fprintf(stderr, " %llu", ns);
                }
                fprintf(stderr, "\n");
            }
#endif
            break;
        }
        tp.count = (uint32_t)round_tasks;
        tree_kernel<G, true><<<dim3(tp.count), dim3(64), lds, stream>>>(tp);
        HIPCHECK(hipGetLastError());
        g_stats.n_tasks += tp.count;
        g_stats.n_rounds += 1;
        any_task = true;
    }
    if (any_task) {
        finalize_kernel<G><<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(lib->dl, lig0, n, sl.deferred, sl.bestbuf, scores);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipEventRecord(sl.walk_done, stream));
    sl.walked = true;
    if (g_profiling) {
        HIPCHECK(hipEventRecord(sl.ev[5], stream));
        HIPCHECK(hipEventSynchronize(sl.ev[5]));
        float a = 0, b = 0, c = 0, d = 0;
        HIPCHECK(hipEventElapsedTime(&a, sl.ev[0], sl.ev[1]));
        HIPCHECK(hipEventElapsedTime(&b, sl.ev[1], sl.ev[2]));
        HIPCHECK(hipEventElapsedTime(&c, sl.ev[3], sl.ev[4]));
        HIPCHECK(hipEventElapsedTime(&d, sl.ev[4], sl.ev[5]));
        g_stats.ms_sizes += a;
        g_stats.ms_tables += b;
        g_stats.ms_tree += c;
        g_stats.ms_tasks += d;
        g_stats.ms_total += a + b + c + d;
    }
    g_stats.table_bytes += sl.table_total;
    g_stats.n_chunks += 1;
    return PMX_OK;
}

// Work items = (model, chunk), model-major: the table phase of item i + 1 overlaps the tree phase of item i, also across
// the models of pmx_score_multi. scores_dev is [n_models][count]; the status (a property of the ligand record) is
// reported once, by the first model.
template <int G>
static int score_chunks(const pmx_model *const *models, int n_models, const pmx_library *lib, const Weights &W, uint64_t first,
                        uint64_t count, uint64_t model_stride, float *scores_dev, int32_t *status_dev, hipStream_t stream,
                        Workspace &ws) {
    // equal chunks: the range is cut into the fewest chunks of at most chunk_size() ligands, all of the same size
    uint32_t cap_max = std::min<uint32_t>(chunk_size(), ws.chunk_cap);
    const uint32_t attr_bit = 1u << __builtin_ctz((unsigned)G);
    if (!(ws.lds_attr_set & attr_bit)) { // once per device (the workspace is per device) and group width
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tables_kernel_v2<G, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tables_kernel_v2<G, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tree_kernel<G, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tree_kernel<G, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu));
        ws.lds_attr_set |= attr_bit;
    }
    // PMX_OVERLAP=0 runs both phases on the caller's stream (no concurrency between chunks)
    const bool overlap = env_long("PMX_OVERLAP", 1) != 0;
    hipStream_t side = overlap ? ws.side : stream;
    if (overlap) { // the side stream starts after whatever the caller queued before this call
        HIPCHECK(hipEventRecord(ws.entry, stream));
        HIPCHECK(hipStreamWaitEvent(side, ws.entry, 0));
    }
    for (;;) {
    const uint64_t want_chunks = (count + cap_max - 1) / cap_max;
    const uint32_t cap = (uint32_t)((count + want_chunks - 1) / std::max<uint64_t>(want_chunks, 1));
    const uint64_t n_chunks = (count + cap - 1) / cap, n_items = n_chunks * (uint64_t)n_models;
    auto model_of = [&](uint64_t it) { return models[it / n_chunks]; };
    auto chunk_n = [&](uint64_t it) { return (uint32_t)std::min<uint64_t>(cap, count - (it % n_chunks) * cap); };
    auto chunk_first = [&](uint64_t it) { return first + (it % n_chunks) * cap; };
    auto chunk_status = [&](uint64_t it) { return (status_dev && it < n_chunks) ? status_dev + it * cap : ws.slot[it & 1].status; };
    auto chunk_scores = [&](uint64_t it) { return scores_dev + (it / n_chunks) * model_stride + (it % n_chunks) * cap; };
    int rc = table_phase<G>(model_of(0), lib, W, chunk_first(0), chunk_n(0), chunk_status(0), ws.slot[0], side, ws.num_cu);
    for (uint64_t it = 0; it < n_items && rc == PMX_OK; ++it) {
        if (it + 1 < n_items)
            rc = table_phase<G>(model_of(it + 1), lib, W, chunk_first(it + 1), chunk_n(it + 1), chunk_status(it + 1),
                                ws.slot[(it + 1) & 1], side, ws.num_cu);
        if (rc == PMX_OK) rc = tree_phase<G>(model_of(it), lib, chunk_status(it), chunk_scores(it), ws.slot[it & 1], ws, stream);
    }
    if (rc != PMX_OK) { // leave no work in flight that still references the slots
        (void)hipStreamSynchronize(side);
        (void)hipStreamSynchronize(stream);
    }
    if (rc != kRetrySmaller) return rc;
    // a chunk's tables did not fit the arena: start this range again with chunks of half the size (scoring is idempotent)
    if (cap_max <= 1024) return fail(PMX_ERR_OOM, "the score tables of 1024 ligands exceed the arena limit (PMX_ARENA_MAX_MB)");
    cap_max = std::max<uint32_t>(1024, cap / 2);
    for (Slot &sl : ws.slot) sl.walked = false;
    TRACE("chunk tables exceed the arena limit: retrying with chunks of at most %u ligands", cap_max);
    }
}



// ------------------------------------------------------------------------------------ the screening engine (pmx_screen.hip)
// Everything a call does is enqueued on the caller's stream: no device-to-host read, no host thread, no lock held while
// kernels run. Per super-chunk of <= PMX_SUPER ligands: clear the control block; ligand_kernel over the range (tables in
// per-wave slices); ligand_kernel over the ligands whose tables need the arena; a fixed number of task rounds (each a
// snapshot of the queue + one persistent launch that exits at once when the round is empty; the last round never
// queues); finalize; then the same once more for ligands the arena had no room for (normally none).
constexpr uint32_t kParamSlots = 64; // launches in flight on one stream never come near this
struct ScreenWs {
    Ctl *ctl = nullptr;
    ScreenParams *params = nullptr;
    uint8_t *slices = nullptr;
    size_t slices_bytes = 0;
    uint8_t *big = nullptr;
    size_t big_bytes = 0;
    uint32_t epoch = 0;
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
    uint8_t *queue = nullptr;
    size_t queue_bytes = 0;
    uint32_t *lists = nullptr; // ovf | carry | heavy
    uint32_t list_cap = 0;
    int num_cu = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // profiling: start | ligand passes done | task rounds done
    bool ev_valid = false;
    hipStream_t last_stream = nullptr;
    std::mutex mu; // held while a call enqueues (the workspace belongs to one call at a time, in stream order)
};
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<ScreenWs>> g_screen; // (device, stream)

static int ensure_screen(int device, hipStream_t stream, ScreenWs **out) {
    ScreenWs *w;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto &slot = g_screen[std::make_pair(device, stream)];
        if (!slot) slot.reset(new ScreenWs());
        w = slot.get();
    }
    *out = w;
    return PMX_OK;
}

template <typename T>
static int grow(T **ptr, size_t *have, size_t want, hipStream_t stream) {
    if (*have >= want) return PMX_OK;
    if (*ptr) {
        HIPCHECK(hipStreamSynchronize(stream)); // queued work may still use the old buffer (only when a buffer grows)
        (void)hipFree(*ptr);
        *ptr = nullptr;
        *have = 0;
    }
    HIPCHECK(hipMalloc((void **)ptr, want));
    *have = want;
    return PMX_OK;
}

template <int G>
static int score_screen(const pmx_model *model, const pmx_library *lib, const Weights &W, uint64_t first, uint64_t count, float *scores_dev,
                        int32_t *status_dev, hipStream_t stream, ScreenWs &ws, bool first_model) {
    if (count > 0xfffffff0ull) return fail(PMX_ERR_INVALID, "more than 2^32 ligands in one call");
    if (!ws.ctl) {
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, lib->device));
        ws.num_cu = prop.multiProcessorCount;
        HIPCHECK(hipMalloc((void **)&ws.ctl, sizeof(Ctl)));
        HIPCHECK(hipMalloc((void **)&ws.params, sizeof(ScreenParams) * kParamSlots));
        for (auto &e : ws.ev) HIPCHECK(hipEventCreate(&e));
    }
    ScreenParams p;
    p.M = model->dm;
    int rc = pair_functions(const_cast<pmx_model *>(model), W, stream, &p.F);
    if (rc) return rc;
    p.lib = lib->dl;
    p.sidtab = model->sidtab;
    p.subnodes = model->subnodes;
    p.W = W;
    p.first = first;
    p.ctl = ws.ctl;
    p.flags = (uint32_t)env_long("PMX_TREE_FLAGS", 0);
    p.max_nodes = (uint32_t)std::max(4, std::min(lib->info.max_nodes, PMX_MAX_LIGAND_NODES));
    const WaveShape<G> shape = wave_shape<G>(model->dm.K, (int)p.max_nodes);
    const uint32_t waves_per_cu = (uint32_t)std::max<long>(1, std::min<long>({(long)(kLdsPerCu / shape.bytes), 4L * PMX_SCREEN_WAVES, env_long("PMX_WAVES_PER_CU", 32)}));
    const uint32_t grid = (uint32_t)ws.num_cu * waves_per_cu;
    const uint32_t slice_bytes = (uint32_t)std::max<long>(4, env_long("PMX_SLICE_KB", 48)) * 1024u;
    rc = grow(&ws.slices, &ws.slices_bytes, (size_t)grid * slice_bytes, stream);
    if (rc) return rc;
    // large slices for the ligands whose tables exceed a slice: as large as a table of this model and library can get, at most
    // PMX_BIG_SLICE_MB each, PMX_BIG_TOTAL_MB together (what is larger still goes to the arena)
    uint32_t big_bytes, big_grid;
    {
        const uint64_t nlmax = (uint64_t)std::min<int>(PMX_MAX_LEVELS, std::max(1, lib->info.max_clusters));
        const uint64_t K = (uint64_t)std::max(1, model->dm.K);
        const uint64_t worst = rec_bytes<G>((uint32_t)(nlmax * K), (uint32_t)(nlmax * (nlmax - 1) / 2 * K * K), (uint32_t)nlmax);
        const uint64_t cap = (uint64_t)std::max<long>(1, env_long("PMX_BIG_SLICE_MB", 32)) << 20;
        big_bytes = (uint32_t)std::max<uint64_t>(slice_bytes, (std::min(worst, cap) + 4095) & ~4095ull);
        const uint64_t total = (uint64_t)std::max<long>(64, env_long("PMX_BIG_TOTAL_MB", 1024)) << 20;
        big_grid = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(grid, total / big_bytes));
    }
    rc = grow(&ws.big, &ws.big_bytes, (size_t)big_grid * big_bytes, stream);
    if (rc) return rc;
    rc = grow(&ws.arena, &ws.arena_bytes, (size_t)std::max<long>(16, env_long("PMX_ARENA_MB", 6144)) << 20, stream);
    if (rc) return rc;
    rc = grow(&ws.queue, &ws.queue_bytes, (size_t)std::max<long>(1, env_long("PMX_TASKQ_MB", 2048)) << 20, stream);
    if (rc) return rc;
    // super-chunk: the arena holds the tables of the ligands whose tree is split, until the chunk's subtrees are done
    const uint32_t super = (uint32_t)std::max<long>(1024, std::min<long>(env_long("PMX_SUPER", (1L << 20) * 8 / std::max(G, 8)), 1 << 24));
    {
        size_t have = (size_t)ws.list_cap * 12;
        rc = grow(&ws.lists, &have, (size_t)super * 12, stream);
        if (rc) return rc;
        ws.list_cap = super;
    }
    p.arena = ws.arena;
    p.arena_bytes = std::min<unsigned long long>(ws.arena_bytes, (1ull << 36) - 4096);
    p.ovf_list = ws.lists;
    p.carry_list = ws.lists + super;
    p.heavy_list = ws.lists + 2 * (size_t)super;
    p.list_cap = super;
    p.queue = ws.queue;
    p.qcap = (uint32_t)std::min<size_t>(ws.queue_bytes / task_rec_bytes<G>() / kShards, 0x3fffffffu / kShards);
    p.budget = (uint32_t)std::max<long>(16, env_long("PMX_BUDGET", 1024));
    p.min_levels = (uint32_t)std::max<long>(0, env_long("PMX_MIN_LEVELS", 3));
    p.max_passes = (unsigned long long)std::max<long>(1, env_long("PMX_MAXITERS", 1L << 40));
    p.scores = scores_dev;
    p.status = status_dev;
    const int rounds = (int)std::max<long>(1, env_long("PMX_ROUNDS", 8));
    p.last_round = 0;
    p.pad_ = 0;
    const bool exact = (p.flags & 8) != 0;
    const size_t lds = shape.bytes;
    if (g_profiling && first_model) HIPCHECK(hipEventRecord(ws.ev[0], stream));
    auto launch = [&](int mode, uint32_t blocks) {
        p.mode = mode;
        if (exact) ligand_kernel<G, true><<<dim3(blocks), dim3(64), lds, stream>>>(p);
        else ligand_kernel<G, false><<<dim3(blocks), dim3(64), lds, stream>>>(p);
    };
    for (uint64_t lo = 0; lo < count; lo += super) {
        p.lo = (uint32_t)lo;
        p.hi = (uint32_t)std::min<uint64_t>(count, lo + super);
        ctl_clear_kernel<<<dim3((sizeof(Ctl) / 4 + 255) / 256), dim3(256), 0, stream>>>(ws.ctl, (lo == 0 && first_model) ? 1 : 0);
        // every ligand whose tables fit a slice
        p.slices = ws.slices;
        p.slice_bytes = slice_bytes;
        launch(0, grid);
        // the others with large slices (fewer wavefronts)
        p.slices = ws.big;
        p.slice_bytes = big_bytes;
        launch(1, big_grid);
        // and what exceeds those from the arena
        launch(2, big_grid);
        // the subtrees the over-budget walkers queued, and the ones those queue in turn: a fixed number of rounds, each a snapshot of
        // the queue and one persistent launch (an empty round exits at once); the last round walks everything to its end
        for (int r = 0; r < rounds; ++r) {
            p.last_round = r + 1 == rounds ? 1u : 0u;
            round_kernel<<<dim3(1), dim3(64), 0, stream>>>(ws.ctl, p.qcap);
            task_kernel<G><<<dim3(std::min<uint32_t>(grid, (uint32_t)ws.num_cu * 4u * PMX_TASK_WAVES)), dim3(64), lds, stream>>>(p);
        }
        finalize_kernel<G><<<dim3((super + 255) / 256), dim3(256), 0, stream>>>(p);
    }
    HIPCHECK(hipGetLastError());
    if (g_profiling) HIPCHECK(hipEventRecord(ws.ev[1], stream));
    ws.ev_valid = g_profiling != 0;
    ws.last_stream = stream;
    return PMX_OK;
}

// Statistics of the last call on this thread's workspace: synchronises the stream the call ran on.
static int screen_stats(pmx_score_stats *out) {
    ScreenWs *w = g_last_screen;
    if (!w || !w->ctl) return PMX_OK;
    HIPCHECK(hipSetDevice(g_last_device));
    HIPCHECK(hipStreamSynchronize(w->last_stream));
    std::vector<unsigned char> host(sizeof(Ctl));
    HIPCHECK(hipMemcpy(host.data(), w->ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    const Ctl *c = reinterpret_cast<const Ctl *>(host.data());
    unsigned long long st[kStatWords] = {0};
    for (int sh = 0; sh < kScreenStatShards; ++sh)
        for (int i = 0; i < kStatWords; ++i) st[i] = (i == 5) ? std::max(st[i], c->stats[sh][i]) : st[i] + c->stats[sh][i];
    *out = pmx_score_stats{};
    out->n_steps = st[0];
    out->n_iters = st[1];
    out->n_heavy = st[2];
    out->n_items = st[3];
    out->n_exact_cells = st[4];
    out->max_iters_ligand = st[5];
    out->n_tasks = st[6];
    out->n_overflow = st[7] & 0xffffffffull;
    out->n_steps_first = st[14]; // records written to the queue
    out->queue_overflow = c->qflag;
    if (trace_on())
        fprintf(stderr, "[pmx] wave ticks: scan %llu tables %llu bounds %llu walk %llu | alive %llu idle %llu | arena top %llu heavy %u ovf %u carry %u | probes %llu probe passes %llu exported %llu\n", st[8], st[9], st[10], st[11], st[12], st[13],
                (unsigned long long)c->arena_top, c->heavy_count, c->ovf_count, c->carry_count, st[7] >> 32, st[15], st[14]);
    if (w->ev_valid) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, w->ev[0], w->ev[1]));
        out->ms_total = ms;
    }
    if (c->err) return fail(PMX_ERR_INVALID, "tree walk hit the iteration cap (PMX_MAXITERS)");
    return PMX_OK;
}

static int next_pow2(int x) {
    int g = 1;
    while (g < x) g <<= 1;
    return g;
}

extern "C" int pmx_score_multi(const pmx_model *const *models, int n_models, const pmx_library *lib,
                               const float weights[PMX_NUM_TYPES], uint64_t first, uint64_t count, float *scores_dev,
                               int32_t *status_dev, void *stream_) {
    if (!models || n_models < 0 || !lib || !weights || (!scores_dev && count && n_models)) return fail(PMX_ERR_INVALID, "null argument");
    for (int i = 0; i < n_models; ++i) {
        if (!models[i]) return fail(PMX_ERR_INVALID, "null model");
        if (models[i]->device != lib->device) return fail(PMX_ERR_INVALID, "model and library live on different devices");
    }
    if (first > lib->info.n_ligands || count > lib->info.n_ligands - first) return fail(PMX_ERR_INVALID, "ligand range out of bounds");
    if (env_long("PMX_ENGINE", 3) == 3) {
        g_stats = pmx_score_stats{};
        g_last_screen = nullptr;
        if (count == 0 || n_models == 0) return PMX_OK;
        HIPCHECK(hipSetDevice(lib->device));
        Weights W;
        for (int t = 0; t < PMX_NUM_TYPES; ++t) W.w[t] = weights[t];
        hipStream_t stream = static_cast<hipStream_t>(stream_);
        const int G = next_pow2(std::max(1, std::min(lib->info.max_conformers, PMX_MAX_CONFORMERS)));
        ScreenWs *ws = nullptr;
        int rc = ensure_screen(lib->device, stream, &ws);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(ws->mu);
        for (int m = 0; m < n_models && rc == PMX_OK; ++m) {
            float *sc = scores_dev + (size_t)m * count;
            int32_t *st = m == 0 ? status_dev : nullptr;
            switch (G) {
            case 1: rc = score_screen<1>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            case 2: rc = score_screen<2>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            case 4: rc = score_screen<4>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            case 8: rc = score_screen<8>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            case 16: rc = score_screen<16>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            case 32: rc = score_screen<32>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            default: rc = score_screen<64>(models[m], lib, W, first, count, sc, st, stream, *ws, m == 0); break;
            }
        }
        g_last_screen = ws;
        g_last_device = lib->device;
        return rc;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    g_stats = pmx_score_stats{};
    if (count == 0 || n_models == 0) return PMX_OK;
    HIPCHECK(hipSetDevice(lib->device));
    Weights W;
    for (int t = 0; t < PMX_NUM_TYPES; ++t) W.w[t] = weights[t];
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int G = next_pow2(std::max(1, std::min(lib->info.max_conformers, PMX_MAX_CONFORMERS)));
    auto run = [&](uint64_t lo, uint64_t n, hipStream_t q, Workspace &ws) -> int {
        float *sc = scores_dev + lo;
        int32_t *st = status_dev ? status_dev + lo : nullptr;
        switch (G) {
        case 1: return score_chunks<1>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        case 2: return score_chunks<2>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        case 4: return score_chunks<4>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        case 8: return score_chunks<8>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        case 16: return score_chunks<16>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        case 32: return score_chunks<32>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        default: return score_chunks<64>(models, n_models, lib, W, first + lo, n, count, sc, st, q, ws);
        }
    };
    Workspace *ws0 = nullptr;
    int rc = ensure_workspace(lib->device, 0, &ws0);
    if (rc) return rc;
    // Several independent chunk pipelines over equal parts of the range, each driven by its own host thread: the
    // kernels of one fill the gaps of the others (kernel tails, host round trips, the phase the others are not in).
    const uint64_t min_part = std::min<uint32_t>(chunk_size(), 65536);
    const int P = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max<long>(1, env_long("PMX_PIPELINES", 3)), 8, count / min_part}));
    if (P == 1) return run(0, count, stream, *ws0);
    struct Part {
        Workspace *ws = nullptr;
        uint64_t lo = 0, n = 0;
        int rc = PMX_OK;
        std::string err;
        pmx_score_stats st = {};
    };
    std::vector<Part> parts((size_t)P);
    for (int i = 0; i < P; ++i) {
        parts[i].lo = count * (uint64_t)i / (uint64_t)P;
        parts[i].n = count * (uint64_t)(i + 1) / (uint64_t)P - parts[i].lo;
        rc = ensure_workspace(lib->device, i, &parts[i].ws);
        if (rc) return rc;
        if (i > 0) HIPCHECK(hipEventRecord(parts[i].ws->entry, stream)); // every pipeline starts after the caller's queued work
    }
    std::vector<std::thread> threads;
    for (int i = 1; i < P; ++i)
        threads.emplace_back([&, i] {
            Part &pt = parts[i];
            if (hipSetDevice(lib->device) != hipSuccess || hipStreamWaitEvent(pt.ws->own, pt.ws->entry, 0) != hipSuccess) {
                pt.rc = PMX_ERR_HIP;
                pt.err = "pipeline thread: device setup failed";
                return;
            }
            g_stats = pmx_score_stats{};
            pt.rc = run(pt.lo, pt.n, pt.ws->own, *pt.ws);
            if (pt.rc == PMX_OK && hipEventRecord(pt.ws->done, pt.ws->own) != hipSuccess) pt.rc = PMX_ERR_HIP;
            if (pt.rc != PMX_OK) pt.err = g_err;
            pt.st = g_stats;
        });
    rc = run(parts[0].lo, parts[0].n, stream, *ws0);
    for (auto &t : threads) t.join();
    bool failed = rc != PMX_OK;
    for (int i = 1; i < P; ++i) failed = failed || parts[i].rc != PMX_OK;
    if (failed) { // nothing may still be writing scores_dev when the caller sees the error
        for (int i = 0; i < P; ++i) {
            (void)hipStreamSynchronize(parts[i].ws->own);
            (void)hipStreamSynchronize(parts[i].ws->side);
        }
        (void)hipStreamSynchronize(stream);
    }
    if (rc != PMX_OK) return rc;
    for (int i = 1; i < P; ++i)
        if (parts[i].rc != PMX_OK) return fail(parts[i].rc, "%s", parts[i].err.c_str());
    for (int i = 1; i < P; ++i) {
        HIPCHECK(hipStreamWaitEvent(stream, parts[i].ws->done, 0)); // the caller's stream sees every part
        const pmx_score_stats &st1 = parts[i].st;
        g_stats.ms_sizes += st1.ms_sizes;
        g_stats.ms_tables += st1.ms_tables;
        g_stats.ms_tree += st1.ms_tree;
        g_stats.ms_tasks += st1.ms_tasks;
        g_stats.ms_total += st1.ms_total;
        g_stats.table_bytes += st1.table_bytes;
        g_stats.n_chunks += st1.n_chunks;
        g_stats.n_tasks += st1.n_tasks;
        g_stats.n_rounds += st1.n_rounds;
        g_stats.queue_overflow |= st1.queue_overflow;
        g_stats.n_steps += st1.n_steps;
        g_stats.n_iters += st1.n_iters;
        g_stats.max_iters_ligand = std::max(g_stats.max_iters_ligand, st1.max_iters_ligand);
        g_stats.max_iters_task = std::max(g_stats.max_iters_task, st1.max_iters_task);
        g_stats.n_steps_first += st1.n_steps_first;
    }
    return PMX_OK;
}

extern "C" int pmx_score(const pmx_model *model, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
                         uint64_t count, float *scores_dev, int32_t *status_dev, void *stream) {
    if (!model) return fail(PMX_ERR_INVALID, "null argument");
    return pmx_score_multi(&model, 1, lib, weights, first, count, scores_dev, status_dev, stream);
}


// Frees the cached scoring workspaces of `device` (table arenas, task queues, class lists, fused-engine buffers): they are
// grown on demand and kept between calls, which is what a screening loop wants and what a long-lived host program that
// is done screening does not.
int pmx_topk_release(int device);
extern "C" int pmx_release_workspaces(int device) {
    HIPCHECK(hipSetDevice(device));
    HIPCHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto it = g_ws.begin(); it != g_ws.end();) {
        if (it->first.first != device) {
            ++it;
            continue;
        }
        Workspace &w = it->second;
        for (Slot &sl : w.slot) {
            for (void *q : {(void *)sl.units, (void *)sl.status, (void *)sl.taboff, (void *)sl.arena, (void *)sl.meta, (void *)sl.bestbuf, (void *)sl.deferred})
                if (q) (void)hipFree(q);
            if (sl.meta_host) (void)hipHostFree(sl.meta_host);
            for (auto &ev : sl.ev)
                if (ev) (void)hipEventDestroy(ev);
            if (sl.tables_done) (void)hipEventDestroy(sl.tables_done);
            if (sl.walk_done) (void)hipEventDestroy(sl.walk_done);
        }
        if (w.queue) (void)hipFree(w.queue);
        if (w.side) (void)hipStreamDestroy(w.side);
        if (w.own) (void)hipStreamDestroy(w.own);
        if (w.entry) (void)hipEventDestroy(w.entry);
        if (w.done) (void)hipEventDestroy(w.done);
        it = g_ws.erase(it);
    }
    return pmx_topk_release(device);
}

// error hook for pmx_topk.hip (keeps the thread-local message in one translation unit)
int pmx_topk_fail(int code, const char *msg) { return fail(code, "%s", msg); }

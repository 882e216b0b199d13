// pmx_api.hip - host side of libpmx.so: the C ABI of include/pmx.h over the kernels of pmx_screen.hip.
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
#include "pmx_screen.hip"
#include "pmx_debug.h"

using namespace pmx;

// ------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static thread_local pmx_score_stats g_stats = {};
static int g_profiling = 0;
struct ScreenWs;
static thread_local std::shared_ptr<ScreenWs> g_last_screen;
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
    uint32_t NS = 0, NF = 0, ncell = 0;
    float h = 0.f;
    uint16_t *sidtab = nullptr;  // device [K * 128]
    uint32_t *sub_off = nullptr;  // device [NS + 1]: the nodes of subset s are sub_nodes[sub_off[s] .. sub_off[s + 1]), ascending
    uint8_t *sub_nodes = nullptr; // device
    float2 *win = nullptr;        // device [NS * NS * ncell] exact pass windows
    uint64_t n_complex_cells = 0;
    std::mutex fn_mu;
    std::vector<FnEntry> fn;
    uint64_t fn_stamp = 0;
};

// A set of model nodes (PMX_MAX_MODEL_NODES bits).
struct NodeSet {
    static constexpr int W = PMX_MAX_MODEL_NODES / 64;
    uint64_t w[W] = {};
    bool any() const { for (int i = 0; i < W; ++i) if (w[i]) return true; return false; }
    int count() const { int c = 0; for (int i = 0; i < W; ++i) c += __builtin_popcountll(w[i]); return c; }
    void set(int m) { w[m >> 6] |= 1ull << (m & 63); }
    bool operator==(const NodeSet &o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
    NodeSet operator&(const NodeSet &o) const { NodeSet r; for (int i = 0; i < W; ++i) r.w[i] = w[i] & o.w[i]; return r; }
    std::vector<int> list() const { // ascending
        std::vector<int> v;
        for (int i = 0; i < W; ++i)
            for (uint64_t x = w[i]; x; x &= x - 1) v.push_back(i * 64 + __builtin_ctzll(x));
        return v;
    }
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

static int build_pair_functions(pmx_model *m, const pmx_model_desc *d, const std::vector<NodeSet> &cnodes, const NodeSet *tnodes) {
    const int Nm = m->dm.Nm, K = m->dm.K;
    // node subsets: (model cluster, ligand type mask) -> the cluster's nodes of those types (graph_match.py:148-150)
    std::vector<NodeSet> subs(1);
    std::vector<uint16_t> sidtab((size_t)std::max(K, 1) * 128, 0);
    for (int a = 0; a < K; ++a)
        for (int mask = 0; mask < 128; ++mask) {
            const NodeSet nodes = cnodes[a] & tnodes[mask];
            if (!nodes.any()) continue;
            size_t id = 1;
            for (; id < subs.size(); ++id)
                if (subs[id] == nodes) break;
            if (id == subs.size()) subs.push_back(nodes);
            sidtab[(size_t)a * 128 + mask] = (uint16_t)id;
        }
    const uint32_t NS = (uint32_t)subs.size();
    if (NS > 65535u) return fail(PMX_ERR_INVALID, "model has %u distinct node subsets (max 65535)", NS);
    std::vector<std::vector<int>> sublist(NS);
    std::vector<uint32_t> sub_off(NS + 1, 0);
    std::vector<uint8_t> sub_nodes;
    for (uint32_t s = 0; s < NS; ++s) {
        sublist[s] = subs[s].list();
        sub_off[s] = (uint32_t)sub_nodes.size();
        for (int x : sublist[s]) sub_nodes.push_back((uint8_t)x);
    }
    sub_off[NS] = (uint32_t)sub_nodes.size();
    if (sub_nodes.empty()) sub_nodes.push_back(0);
    // grid: h = the largest power of two <= std_min / 4 (quintic Hermite error < 6e-8 of the peak, measured), range to mean + 7 std
    float std_min = 1e30f, dmax = 1.f;
    for (int i = 0; i < Nm * Nm; ++i) {
        std_min = std::min(std_min, d->edge_std[i]);
        dmax = std::max(dmax, d->edge_mean[i] + 7.0f * d->edge_std[i]);
    }
    if (Nm == 0) std_min = 1.f;
    float h = 0.5f;
    while (h > std_min / 4.f && h > 1.f / 64.f) h *= 0.5f;
    const uint32_t ncell = (uint32_t)std::ceil((double)dmax / (double)h) + 1;
    if (ncell > 16384 || (uint64_t)NS * NS * ncell * sizeof(FnCell) >= (4ull << 30)) // (the kernels address the table with 32-bit byte offsets)
        return fail(PMX_ERR_INVALID, "pair-function tables of this model would take %llu cells x %u x %u subsets", (unsigned long long)ncell, NS, NS);
    // exact pass window of every model edge
    std::vector<float> wlo((size_t)Nm * Nm), whi((size_t)Nm * Nm);
    std::vector<uint8_t> wok((size_t)Nm * Nm);
    for (int i = 0; i < Nm * Nm; ++i) wok[i] = edge_window(d->edge_mean[i], pass_threshold(d->edge_std[i]), wlo[i], whi[i]) ? 1 : 0;
    const float INF = INFINITY;
    // a symmetric model (edge[m][n] == edge[n][m]: distances are) has F_(A,B) == F_(B,A): the pair (lo, hi) is stored once
    const bool tri = m->dm.symmetric != 0;
    const uint32_t NF = tri ? NS * (NS + 1) / 2 : NS * NS;
    std::vector<float2> win((size_t)NF * ncell);
    uint64_t n_complex = 0;
    std::vector<std::pair<float, int>> ev;
    std::vector<std::pair<float, float>> pass;
    for (uint32_t sa = 0; sa < NS; ++sa)
        for (uint32_t sb = 0; sb < NS; ++sb) {
            if (tri && sb > sa) continue;
            float2 *out = win.data() + (size_t)(tri ? sa * (sa + 1) / 2 + sb : sa * NS + sb) * ncell;
            const std::vector<int> &A = sublist[sa], &B = sublist[sb];
            if (A.empty() || B.empty()) { // no item: never a fail
                for (uint32_t i = 0; i < ncell; ++i) out[i] = make_float2(-INF, INF);
                continue;
            }
            const int mn = (int)(A.size() * B.size());
            ev.clear();
            for (const int am : A)
                for (const int bm : B) {
                    const int e = am * Nm + bm;
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
                float2 w = make_float2(INF, INF); // never passes (no distance is >= INF; written so that lo <= hi holds for every window: the device's median-of-three test)
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
    HIPCHECK(hipMalloc((void **)&m->sub_off, (size_t)(NS + 1) * 4));
    HIPCHECK(hipMalloc((void **)&m->sub_nodes, sub_nodes.size()));
    HIPCHECK(hipMalloc((void **)&m->win, win.size() * sizeof(float2)));
    HIPCHECK(hipMemcpy(m->sidtab, sidtab.data(), sidtab.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m->sub_off, sub_off.data(), (size_t)(NS + 1) * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m->sub_nodes, sub_nodes.data(), sub_nodes.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(m->win, win.data(), win.size() * sizeof(float2), hipMemcpyHostToDevice));
    m->NS = NS;
    m->NF = NF;
    m->ncell = ncell;
    m->h = h;
    m->n_complex_cells = n_complex;
    return PMX_OK;
}

// Cells whose polynomial deviates from the function by more than this, relative to the function, are flagged (FnCell) and
// evaluated term by term where a self entry meets them. 2e-7 is just above what the float32 coefficients themselves cost
// (each rounded to 6e-8 of its value) and leaves room for the reference's own float32 rounding of z and z^2 (about
// 1e-7 z^2 / 2 relative) inside the 2e-6 the parity tests allow; on the bench library 1-3 self items per ligand are flagged.
static double fn_rel_tol() {
    const char *s = std::getenv("PMX_FN_RELTOL");
    const double v = (s && *s) ? std::atof(s) : 2e-7;
    return v > 0.0 ? v : 2e-7;
}

// ... and cells where the terms that make up the function are beyond exp(-8) of their peaks (weighted mean of z^2 / 2 above 8,
// |z| > 4): there the reference's own float32 rounding of z and z^2 reaches 1.4e-6 of the value, and only the same operations
// in the same order reproduce it. ([MI355X] round 6: 6 -> 8. A flagged lane makes its whole wavefront walk the term-by-term loop: 6.8 -> 3.1 flagged self values
// per ligand on the bench library, 92.6 -> 91.0 ms per pass, every score of the 10^6 ligands the same bits; the tail sweeps of tests/test_gpu_tails.py and
// test_gpu_pair_tails.py hold at their 2e-6 - 1.8e-6 against the term-by-term engine - and fail narrowly, 1.87e-6, at 9.)
static double fn_max_exponent() {
    const char *s = std::getenv("PMX_FN_MAXEXP");
    const double v = (s && *s) ? std::atof(s) : 8.0;
    return v > 0.0 ? v : 8.0;
}

// The tabulated functions for the call's weights: built on `stream` the first time, kept for the last four weight sets.
static int pair_functions(pmx_model *m, const Weights &W, hipStream_t stream, FnTable *out) {
    std::lock_guard<std::mutex> lock(m->fn_mu);
    FnEntry *hit = nullptr;
    for (FnEntry &e : m->fn)
        if (std::memcmp(&e.W, &W, sizeof(W)) == 0) hit = &e;
    if (!hit) {
        if (m->fn.size() < 4) {
            FnEntry fresh; // (enters the cache only once its buffer and event exist)
            HIPCHECK(hipMalloc((void **)&fresh.cells, (size_t)m->NF * m->ncell * sizeof(FnCell)));
            const hipError_t ee = hipEventCreateWithFlags(&fresh.ready, hipEventDisableTiming);
            if (ee != hipSuccess) {
                (void)hipFree(fresh.cells);
                return fail(PMX_ERR_HIP, "hipEventCreateWithFlags failed: %s", hipGetErrorString(ee));
            }
            m->fn.push_back(fresh);
            hit = &m->fn.back();
        } else { // recycle the least recently used entry once nothing queued still reads it
            hit = &m->fn[0];
            for (FnEntry &e : m->fn)
                if (e.stamp < hit->stamp) hit = &e;
            HIPCHECK(hipDeviceSynchronize());
        }
        hit->W = W;
        fn_build_kernel<<<dim3(m->NF), dim3(128), 0, stream>>>(m->dm, W, m->sub_off, m->sub_nodes, m->NS, m->ncell, m->h, m->win, hit->cells, fn_rel_tol(), fn_max_exponent());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(hit->ready, stream));
    } else {
        HIPCHECK(hipStreamWaitEvent(stream, hit->ready, 0));
    }
    hit->stamp = ++m->fn_stamp;
    out->cells = hit->cells;
    out->plane16 = m->NF * m->ncell;
    out->NS = m->NS;
    out->tri = m->dm.symmetric != 0 ? 1u : 0u;
    out->ncell = m->ncell;
    out->inv_h = 1.0f / m->h;
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
    const size_t off_tclus = off_type + PMX_MAX_MODEL_NODES;
    const size_t off_cpair = off_tclus + 128 * 16;
    const size_t off_cwin = off_cpair + round16(n_pair * sizeof(float2));
    const size_t total = off_cwin + round16(n_pair * sizeof(float2)) + 16;
    std::vector<unsigned char> host(total, 0);
    float4 *edge = reinterpret_cast<float4 *>(host.data() + off_edge);
    uint8_t *ntype = host.data() + off_type;
    uint64_t *tclus = reinterpret_cast<uint64_t *>(host.data() + off_tclus); // [128][2]
    std::vector<NodeSet> cnodes((size_t)std::max(K, 1));
    NodeSet tnodes[128];
    float2 *cpair = reinterpret_cast<float2 *>(host.data() + off_cpair);
    float2 *cwin = reinterpret_cast<float2 *>(host.data() + off_cwin);

    const double s_const = std::sqrt(0.5 * 1.4426950408889634074); // sqrt(0.5 * log2(e))
    for (size_t i = 0; i < n_edge; ++i) {
        const float mean = d->edge_mean[i], sd = d->edge_std[i];
        if (!(sd > 0.f)) return fail(PMX_ERR_INVALID, "edge %zu has distance_std %g", i, (double)sd);
        edge[i] = make_float4(mean, (float)(s_const / (double)sd), pass_threshold(sd), sd);
    }
    NodeSet type_nodes[PMX_NUM_TYPES];
    for (int m = 0; m < Nm; ++m) {
        ntype[m] = d->node_type[m];
        type_nodes[d->node_type[m]].set(m);
    }
    const int NW = std::max(1, (Nm + 63) / 64); // words per cluster in cluster_nodes
    for (int a = 0; a < K; ++a)
        for (int m = 0; m < Nm; ++m)
            if (d->cluster_nodes[(size_t)a * NW + (m >> 6)] >> (m & 63) & 1) cnodes[a].set(m);
    for (int mask = 0; mask < 128; ++mask) {
        NodeSet nodes;
        uint64_t clus[2] = {0, 0};
        for (int t = 0; t < PMX_NUM_TYPES; ++t)
            if (mask >> t & 1)
                for (int i = 0; i < NodeSet::W; ++i) nodes.w[i] |= type_nodes[t].w[i];
        for (int a = 0; a < K; ++a)
            if (d->cluster_typemask[a] & mask) clus[a >> 6] |= 1ull << (a & 63);
        tnodes[mask] = nodes;
        tclus[2 * mask] = clus[0];
        tclus[2 * mask + 1] = clus[1];
    }
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            const double *ca = d->cluster_center + 3 * a, *cb = d->cluster_center + 3 * b;
            const double dist = std::sqrt((ca[0] - cb[0]) * (ca[0] - cb[0]) + (ca[1] - cb[1]) * (ca[1] - cb[1]) +
                                          (ca[2] - cb[2]) * (ca[2] - cb[2])); // graph_match.py:263-264
            cpair[a * K + b] = make_float2((float)dist, (float)(d->cluster_size[a] + d->cluster_size[b])); // :265
        }
    // hull of the exact 2-sigma windows of a cluster pair's node pairs (the dead-entry test of build_tables)
    {
        std::vector<float> wlo(n_edge), whi(n_edge);
        std::vector<uint8_t> wok(n_edge);
        for (size_t i = 0; i < n_edge; ++i) wok[i] = edge_window(d->edge_mean[i], pass_threshold(d->edge_std[i]), wlo[i], whi[i]) ? 1 : 0;
        for (int a = 0; a < K; ++a)
            for (int b = 0; b < K; ++b) {
                float lo = INFINITY, hi = -INFINITY;
                for (const int am : cnodes[a].list())
                    for (const int bm : cnodes[b].list()) {
                        const size_t e = (size_t)am * Nm + bm;
                        if (!wok[e]) continue;
                        lo = std::min(lo, wlo[e]);
                        hi = std::max(hi, whi[e]);
                    }
                cwin[a * K + b] = make_float2(lo, hi);
            }
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
    m->dm.tclus = reinterpret_cast<const uint64_t *>(b8 + off_tclus);
    m->dm.cpair = reinterpret_cast<const float2 *>(b8 + off_cpair);
    m->dm.cwin = reinterpret_cast<const float2 *>(b8 + off_cwin);
    {
        const int rc = build_pair_functions(m, d, cnodes, tnodes);
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
    if (m->sub_off) (void)hipFree(m->sub_off);
    if (m->sub_nodes) (void)hipFree(m->sub_nodes);
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
    bool owns; // false: the caller's device buffers, adopted as they are (pmx_library_view.on_device == 2)
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
    const bool adopt = v->on_device == 2;
    if (adopt && !v->data) return fail(PMX_ERR_INVALID, "null data");
    pmx_library *lib = new pmx_library();
    lib->device = device;
    lib->offsets = nullptr;
    lib->data = nullptr;
    lib->owns = !adopt;
    hipError_t e = hipSuccess;
    if (adopt) { // the buffers stay the caller's: no allocation, no copy
        lib->offsets = const_cast<uint64_t *>(v->offsets);
        lib->data = const_cast<uint8_t *>(v->data);
    } else {
        e = hipMalloc((void **)&lib->offsets, (n + 1) * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&lib->data, std::max<uint64_t>(nbytes, 16));
        if (e == hipSuccess) e = hipMemcpy(lib->offsets, v->offsets, (n + 1) * 8, kind);
        if (e == hipSuccess && nbytes) e = hipMemcpy(lib->data, v->data, nbytes, kind);
    }
    // The counters live in one small buffer per device that is never freed: hipFree waits for every stream of the device, and an adopted
    // library is made while other streams copy and score ([MI355X] the pipeline's scoring calls each started when the NEXT chunk's copy had ended).
    static std::mutex stats_mu;
    static unsigned long long *stats_of_device[64] = {};
    unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    lib->dl.n = n;
    lib->dl.offsets = lib->offsets;
    lib->dl.data = lib->data;
    if (e == hipSuccess && (device < 0 || device >= 64)) e = hipErrorInvalidDevice;
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(stats_mu);
        unsigned long long *&stats_dev = stats_of_device[device];
        if (!stats_dev) e = hipMalloc((void **)&stats_dev, sizeof(stats));
        if (e == hipSuccess) e = hipMemset(stats_dev, 0, sizeof(stats));
        if (e == hipSuccess && n) {
            library_stats_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(lib->dl, lib->data, nbytes, stats_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(stats, stats_dev, sizeof(stats), hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) {
        if (lib->owns && lib->offsets) (void)hipFree(lib->offsets);
        if (lib->owns && lib->data) (void)hipFree(lib->data);
        delete lib;
        return fail(e == hipErrorOutOfMemory ? PMX_ERR_OOM : PMX_ERR_HIP, "library upload failed: %s", hipGetErrorString(e));
    }
    if (stats[5]) { // offsets are validated on the device, for host and device views alike
        if (lib->owns) (void)hipFree(lib->offsets), (void)hipFree(lib->data);
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
    if (lib->owns) (void)hipFree(lib->offsets), (void)hipFree(lib->data);
    delete lib;
    return PMX_OK;
}

// ---------------------------------------------------------------------------------- helpers
static std::mutex g_mu;

static long env_long(const char *name, long dflt) {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    return std::atol(s);
}

static constexpr size_t kLdsPerCu = 160 * 1024;

static bool trace_on() {
    static int v = -1;
    if (v < 0) v = std::getenv("PMX_TRACE") ? 1 : 0;
    return v == 1;
}

// ------------------------------------------------------------------------------------ the screening engine (pmx_screen.hip)
// A call is cut into chunks of <= PMX_SUPER ligands (per pocket); a chunk is: clear the control block; ligand_kernel over the
// range (tables in per-wave slices); ligand_kernel over the ligands whose tables need larger slices / the arena; a fixed
// number of task rounds (each a snapshot of the queue + one persistent launch that exits at once when the round is empty; the
// last round never queues); finalize. The ligand kernels of all chunks go out on the caller's stream; the task rounds of a
// chunk go out on a side stream of the workspace, behind an event of the chunk's ligand kernels, and so run *next to* the
// ligand kernels of the following chunk (of the same pocket or the next one): two sets of control block / arena / queue /
// lists alternate, a set is reused when its chunk's rounds are done (event), and the caller's stream waits for the side
// stream at the end - everything stays ordered on the caller's stream, nothing is read back, no host thread. The two kernels
// share the wave slots of a CU (PMX_LIG_SHARE): the walkers of the task rounds are bound by scalar issue, the table phase
// of the ligand kernel by vector issue and memory, so side by side they fill what the other leaves idle.
struct ChunkSet {
    Ctl *ctl = nullptr;
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
    uint8_t *queue = nullptr;
    size_t queue_bytes = 0;
    uint32_t *lists = nullptr; // ovf | carry | heavy
    size_t lists_bytes = 0;
    hipEvent_t lig_done = nullptr, tasks_done = nullptr;
    bool pending = false; // tasks_done was recorded by the call in progress
    size_t arena_shrunk_to = 0; // the arena size that was accepted when memory was short (0: never shrunk)
};
struct ScreenWs {
    ChunkSet set[2];
    uint8_t *slices = nullptr;
    size_t slices_bytes = 0;
    uint8_t *big = nullptr;
    size_t big_bytes = 0;
    uint8_t *totbuf = nullptr;
    size_t totbuf_bytes = 0;
    uint8_t *pabuf = nullptr; // path_bound()'s pair sums: ligand kernel's wavefronts | task kernel's
    size_t pabuf_bytes = 0;
    int num_cu = 0;
    hipStream_t side = nullptr; // the task rounds' stream
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // profiling: call start | last chunk: ligand kernels start, done | end | last chunk: rounds start, done
    uint64_t ligands_last = 0;
    bool ev_valid = false;
    bool ctl_used[2] = {false, false};
    hipStream_t last_stream = nullptr;
    std::mutex mu; // held while a call enqueues (the workspace belongs to one call at a time, in stream order)
    void free_buffers() {
        for (ChunkSet &c : set) {
            for (void *q : {(void *)c.ctl, (void *)c.arena, (void *)c.queue, (void *)c.lists})
                if (q) (void)hipFree(q);
            for (hipEvent_t e : {c.lig_done, c.tasks_done})
                if (e) (void)hipEventDestroy(e);
            c = ChunkSet{};
        }
        for (void *q : {(void *)slices, (void *)big, (void *)totbuf, (void *)pabuf})
            if (q) (void)hipFree(q);
        slices = big = totbuf = pabuf = nullptr;
        slices_bytes = big_bytes = totbuf_bytes = pabuf_bytes = 0;
        for (auto &e : ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (side) (void)hipStreamDestroy(side);
        side = nullptr;
        ev_valid = false;
        ctl_used[0] = ctl_used[1] = false;
    }
    uint64_t stamp = 0;    // last use (ensure_screen): the least recently used workspace of a device goes first
    bool released = false; // pmx_release_workspaces (or the cap on workspaces per device) took the buffers: a caller that was waiting on `mu` asks for a new workspace
};
// Workspaces are shared: the map, a call in progress and the thread that asks for the last call's statistics each hold a
// reference, so pmx_release_workspaces can take a workspace out of the map and free its buffers while none of them is left
// with a dangling pointer (the object itself goes with its last reference).
static std::map<std::pair<int, hipStream_t>, std::shared_ptr<ScreenWs>> g_screen; // (device, stream)

static uint64_t g_screen_stamp = 0;

// The workspace of (device, stream), made on first use. At most PMX_MAX_WORKSPACES (default 4) are kept per device: a host
// program that scores on short-lived streams would otherwise leave some 40 GB behind per stream it ever used (the key is the raw
// stream handle; nothing tells libpmx that a stream is gone). When one more is needed the least recently used idle one is
// taken out of the map and freed - after a device synchronisation, its stream may no longer exist - and a caller that was
// waiting for it finds it `released` and asks again.
static std::shared_ptr<ScreenWs> ensure_screen(int device, hipStream_t stream) {
    std::shared_ptr<ScreenWs> out, victim;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        const auto key = std::make_pair(device, stream);
        auto it = g_screen.find(key);
        if (it != g_screen.end() && it->second) {
            it->second->stamp = ++g_screen_stamp;
            return it->second;
        }
        const long cap = std::max<long>(1, env_long("PMX_MAX_WORKSPACES", 4));
        long have = 0;
        for (const auto &kv : g_screen) have += kv.first.first == device && kv.second ? 1 : 0;
        if (have >= cap) {
            auto lru = g_screen.end();
            for (auto jt = g_screen.begin(); jt != g_screen.end(); ++jt) {
                if (jt->first.first != device || !jt->second) continue;
                if (lru != g_screen.end() && jt->second->stamp >= lru->second->stamp) continue;
                if (!jt->second->mu.try_lock()) continue; // a call is enqueuing on it
                if (lru != g_screen.end()) lru->second->mu.unlock();
                lru = jt;
            }
            if (lru != g_screen.end()) {
                victim = std::move(lru->second); // (its mutex is held)
                g_screen.erase(lru);
            }
        }
        out = std::make_shared<ScreenWs>();
        out->stamp = ++g_screen_stamp;
        g_screen[key] = out;
    }
    if (victim) {
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        victim->free_buffers();
        victim->released = true;
        victim->mu.unlock();
    }
    return out;
}

// (a buffer only ever grows; queued work may still use the old one, on the caller's stream or on the side stream)
template <typename T>
static int grow(T **ptr, size_t *have, size_t want, hipStream_t stream, hipStream_t side, size_t min_bytes = 0) {
    if (*have >= want) return PMX_OK;
    if (*ptr) {
        HIPCHECK(hipStreamSynchronize(stream));
        if (side) HIPCHECK(hipStreamSynchronize(side));
        (void)hipFree(*ptr);
        *ptr = nullptr;
        *have = 0;
    }
    // min_bytes: the buffer is a cache (the table arena) - a smaller one is slower, never wrong: halve on out-of-memory
    for (;;) {
        const hipError_t e = hipMalloc((void **)ptr, want);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || min_bytes == 0 || want / 2 < min_bytes)
            return fail(e == hipErrorOutOfMemory ? PMX_ERR_OOM : PMX_ERR_HIP, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
        want /= 2;
    }
    *have = want;
    return PMX_OK;
}

// What one pocket of a call needs: kernel parameters, launch shapes, chunk size.
struct PocketPlan {
    ScreenParams p;
    size_t lds = 0;
    uint32_t waves_per_cu = 0, task_waves_per_cu = 0;
    uint32_t slice_bytes = 0, big_bytes = 0, big_grid = 0;
    uint64_t worst_bytes = 0;
    uint32_t super = 0;
    uint32_t pa_bytes = 0; // path_bound()'s buffer per wavefront
};

template <int G>
static int score_screen(const pmx_model *const *models, int n_models, const pmx_library *lib, const Weights &W, uint64_t first, uint64_t count,
                        void *scores_dev, bool scores_f64, int32_t *status_dev, hipStream_t stream, ScreenWs &ws) {
    if (count > 0xfffffff0ull) return fail(PMX_ERR_INVALID, "more than 2^32 ligands in one call");
    if (!ws.num_cu) { // (num_cu is set last: a workspace whose events, stream or control blocks could not be made stays uninitialised)
        auto init = [&]() -> int {
            hipDeviceProp_t prop;
            HIPCHECK(hipGetDeviceProperties(&prop, lib->device));
            for (auto &e : ws.ev) HIPCHECK(hipEventCreate(&e));
            HIPCHECK(hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking));
            for (ChunkSet &c : ws.set) {
                HIPCHECK(hipMalloc((void **)&c.ctl, sizeof(Ctl)));
                HIPCHECK(hipEventCreateWithFlags(&c.lig_done, hipEventDisableTiming));
                HIPCHECK(hipEventCreateWithFlags(&c.tasks_done, hipEventDisableTiming));
            }
            ws.num_cu = prop.multiProcessorCount;
            return PMX_OK;
        };
        const int irc = init();
        if (irc) {
            ws.free_buffers();
            return irc;
        }
    }
    const uint32_t flags = (uint32_t)env_long("PMX_TREE_FLAGS", 0);
    const uint32_t max_nodes = (uint32_t)std::max(4, std::min(lib->info.max_nodes, PMX_MAX_LIGAND_NODES));
    // [MI355X] round 6, passes a walk may take before it splits. On the bench library (92 passes per ligand, 7 % of the walks over 384) 384 / 384
    // and 768 / 384 are the same 99.1 ms; on SURVEY 8d-2's own library (800 passes per ligand) 384 sends 57-74 % of the ligands to the arena and
    // the queue (714 ms with a 64 GB arena, queue full), 768 a third of them (539 ms). A queued subtree keeps the smaller budget: the rounds' tail.
    const uint32_t lig_budget = (uint32_t)std::max<long>(16, env_long("PMX_BUDGET", 768));
    const uint32_t task_budget = (uint32_t)std::max<long>(16, env_long("PMX_TASK_BUDGET", std::min<long>(lig_budget, 384))); // a queued subtree's own budget
    const int rounds = (int)std::max<long>(1, env_long("PMX_ROUNDS", 12));
    const int task_decay_from = (int)std::max<long>(1, env_long("PMX_TASK_DECAY_FROM", 99));
    const uint32_t task_budget_min = (uint32_t)std::max<long>(8, env_long("PMX_TASK_BUDGET_MIN", 48));
    const bool exact = (flags & 8) != 0;
    // any validation switch: the kernels of pmx_screen_debug.hip (libpmx's own read those bits as zero, pmx_screen.hip PMX_WFLAGS)
    const bool debug_kernels = (flags & ~PMX_PRODUCT_FLAGS) != 0;
    bool debug_ok = true;
    // Type weights further apart than PMX_TAILS_RATIO (default 8 = the ratio of the reference's own defaults, graph_match.py:32-40: any override that spreads the weights further takes the exact tails): pair items
    // evaluate rough cells term by term like self items do (item_finish<TAILS>, pmx_screen.hip) - slower, and only then.
    // PMX_PAIR_TAILS = 0 / 1 forces it off / on.
    bool tails = false;
    {
        float wmin = INFINITY, wmax = 0.f;
        bool present[PMX_NUM_TYPES] = {};
        for (int m = 0; m < n_models; ++m)
            for (int i = 0; i < models[m]->dm.Nm; ++i) present[models[m]->node_type[i]] = true; // (only the types the pockets hold can meet in an entry)
        for (int t = 0; t < PMX_NUM_TYPES; ++t) {
            const float a = std::fabs(W.w[t]);
            if (present[t] && a > 0.f && std::isfinite(a)) wmin = std::min(wmin, a), wmax = std::max(wmax, a);
        }
        const char *rs = std::getenv("PMX_TAILS_RATIO");
        const double ratio = (rs && *rs) ? std::atof(rs) : 8.0;
        tails = wmax > 0.f && (double)wmax > ratio * (double)wmin;
        const long force = env_long("PMX_PAIR_TAILS", -1);
        if (force == 0) tails = false;
        if (force > 0) tails = true;
    }
    // arena passes over the ligands an arena pass had no room for, each with the arena to itself (a pass with an empty list exits at
    // once): a ligand is reported PMX_LIGAND_TOO_LARGE when its tables exceed the whole arena - or when the arena-class ligands of a
    // chunk need more than 1 + PMX_ARENA_RETRIES arenas
    const int arena_retries = (int)std::max<long>(1, env_long("PMX_ARENA_RETRIES", 4));

    // ---- per pocket: parameters and launch shapes
    std::vector<PocketPlan> plan((size_t)n_models);
    size_t slices_need = 0, big_need = 0, totbuf_need = 0, pabuf_need = 0;
    uint32_t super_max = 0;
    bool retry_possible = false;
    for (int m = 0; m < n_models; ++m) {
        const pmx_model *model = models[m];
        PocketPlan &pl = plan[(size_t)m];
        ScreenParams &p = pl.p;
        p = ScreenParams{};
        p.M = model->dm;
        int rc = pair_functions(const_cast<pmx_model *>(model), W, stream, &p.F);
        if (rc) return rc;
        p.lib = lib->dl;
        p.sidtab = model->sidtab;
        p.sub_off = model->sub_off;
        p.sub_nodes = model->sub_nodes;
        p.W = W;
        p.first = first;
        p.flags = flags | (scores_f64 ? PMX_SCORES_F64 : 0u);
        p.max_nodes = max_nodes;
        p.budget = lig_budget;
        p.min_levels = (uint32_t)std::max<long>(0, env_long("PMX_MIN_LEVELS", 3));
        p.bound_cost = (uint32_t)std::max<long>(0, env_long("PMX_BOUND_COST", 8192));
        p.dead_min_entries = (uint32_t)std::max<long>(1, env_long("PMX_DEAD_MIN_ENTRIES", 32));
        p.scores = scores_f64 ? reinterpret_cast<float *>(static_cast<double *>(scores_dev) + (size_t)m * count) : static_cast<float *>(scores_dev) + (size_t)m * count;
        p.status = m == 0 ? status_dev : nullptr;
        const WaveShape<G> shape = wave_shape<G>(model->dm.K, (int)max_nodes);
        pl.lds = shape.bytes;
        pl.waves_per_cu = (uint32_t)std::max<long>(2, std::min<long>({(long)(kLdsPerCu / shape.bytes), 4L * PMX_SCREEN_WAVES, env_long("PMX_WAVES_PER_CU", 32)}));
        // the task kernel is the walker alone: it may be built for more waves per SIMD than the ligand kernel (PMX_TASK_WAVES)
        pl.task_waves_per_cu = (uint32_t)std::max<long>(2, std::min<long>({(long)(kLdsPerCu / shape.bytes), 4L * PMX_TASK_WAVES, env_long("PMX_TASK_WAVES_PER_CU", 32)}));
        const uint32_t grid = (uint32_t)ws.num_cu * std::max(pl.waves_per_cu, pl.task_waves_per_cu); // (sizes the per-wavefront buffers of both kernels)
        // per-wavefront slice: 112 KB at 8 conformer lanes (every ligand of the bench library fits, path_bound()'s table included), scaled with the lanes
        // (table bytes grow with the square of the model's cluster count: the 11-cluster 6OIM-like model is the reference point)
        const long k_scale = std::max(1L, std::min(16L, ((long)model->dm.K * model->dm.K + 60) / 121));
        pl.slice_bytes = (uint32_t)std::max<long>(4, env_long("PMX_SLICE_KB", (cand_bounds<G>() ? 112L : 80L) * std::max(1, G / 8) * k_scale)) * 1024u;
        slices_need = std::max(slices_need, (size_t)grid * pl.slice_bytes);
        // large slices for the ligands whose tables exceed a slice: as large as a table of this model and library can get, at most
        // PMX_BIG_SLICE_MB each, PMX_BIG_TOTAL_MB together (what is larger still goes to the arena)
        {
            const uint64_t nlmax = (uint64_t)std::min<int>(PMX_MAX_LEVELS, std::max(1, lib->info.max_clusters));
            const uint64_t K = (uint64_t)std::max(1, std::min(model->dm.K, PMX_MAX_LEVEL_CANDIDATES)); // (candidates per level)
            const uint64_t worst = rec_bytes<G>((uint32_t)(nlmax * K), (uint32_t)(nlmax * (nlmax - 1) / 2 * K * K), (uint32_t)nlmax);
            const uint64_t cap = (uint64_t)std::max<long>(1, env_long("PMX_BIG_SLICE_MB", G >= 32 ? 4 : 32)) << 20;
            pl.worst_bytes = worst;
            pl.big_bytes = (uint32_t)std::max<uint64_t>(pl.slice_bytes, (std::min(worst, cap) + 4095) & ~4095ull);
            const uint64_t total = (uint64_t)std::max<long>(64, env_long("PMX_BIG_TOTAL_MB", G >= 32 ? 16384 : 4096)) << 20;
            pl.big_grid = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(grid, total / pl.big_bytes));
        }
        big_need = std::max(big_need, (size_t)pl.big_grid * pl.big_bytes);
        if (cand_bounds<G>()) { // float[matches <= levels][candidates of all levels][G], at most 1 MB per wavefront (larger jobs do without)
            const uint64_t nlmax = (uint64_t)std::min<int>(PMX_MAX_LEVELS, std::max(1, lib->info.max_clusters));
            const uint64_t need = (nlmax + 1) * nlmax * (uint64_t)std::max(1, std::min(model->dm.K, PMX_MAX_LEVEL_CANDIDATES)) * G * 4;
            pl.pa_bytes = (uint32_t)std::min<uint64_t>((need + 255) & ~255ull, (uint64_t)std::max<long>(1, env_long("PMX_PATH_KB", 1024)) << 10);
            pabuf_need = std::max(pabuf_need, (size_t)grid * pl.pa_bytes * 2u); // ligand kernel | task kernel
        }
        retry_possible = retry_possible || pl.worst_bytes > pl.big_bytes;
        if (!totals_in_lds<G>()) totbuf_need = (size_t)ws.num_cu * 4u * std::max(PMX_SCREEN_WAVES, PMX_TASK_WAVES) * kTotBufBytes * 2u; // ligand kernel | task kernel
        // chunk: what the arena has to hold at a time are the tables of the chunk's split trees. (Cutting a pocket's pass into more
        // chunks than the arena asks for does not pay: every chunk ends in a dozen rounds with a tail each - 1 M ligands in 8
        // chunks: 258 ms back to back, 234 ms with the rounds beside the next chunk's ligand kernel, 207 ms in one chunk.)
        const long super_dflt = std::max(16384L, (1L << 20) * 8 / std::max(G, 8) / k_scale);
        pl.super = (uint32_t)std::max<long>(1024, std::min<long>(env_long("PMX_SUPER", super_dflt), 1 << 24));
        super_max = std::max(super_max, pl.super);
    }
    int rc = grow(&ws.slices, &ws.slices_bytes, slices_need, stream, ws.side);
    if (rc) return rc;
    rc = grow(&ws.big, &ws.big_bytes, big_need, stream, ws.side);
    if (rc) return rc;
    if (totbuf_need) {
        rc = grow(&ws.totbuf, &ws.totbuf_bytes, totbuf_need, stream, ws.side);
        if (rc) return rc;
    }
    if (pabuf_need) {
        rc = grow(&ws.pabuf, &ws.pabuf_bytes, pabuf_need, stream, ws.side);
        if (rc) return rc;
    }
    uint64_t n_chunks = 0;
    for (const PocketPlan &pl : plan) n_chunks += (count + pl.super - 1) / pl.super;
    // The rounds go to the side stream when there is something to run them next to. (Not when an arena pass may have to be
    // retried: the retry's ligand kernels use the large slices, as the next chunk's do.)
    // [MI355X] history of this switch. With the round-3 search (task rounds = 47 % of a pass, scalar-bound) the rounds beside the
    // next chunk's ligand kernel gained 3.6 % on the 12.5 M-ligand shard (2.39 s against 2.47 s) and lost wherever chunks were
    // small (16 pockets: 24.5 s against 22.2 s). With the path-aware bound the rounds are 22 % of a pass and mostly tail: the
    // same shard takes 2.15 s with them beside the next ligand kernel (which then has half the wave slots) and 1.66 s back to
    // back. So PMX_OVERLAP = 0 (default): everything on the caller's stream; 1: beside each other within a pocket whose chunks
    // hold >= 2^19 ligands; 2: always, across pockets as well.
    const long overlap_mode = env_long("PMX_OVERLAP", 0);
    uint32_t super_min = ~0u;
    for (const PocketPlan &pl : plan) super_min = std::min(super_min, pl.super);
    const bool overlap = n_chunks >= 2 && !retry_possible && overlap_mode != 0 && (overlap_mode >= 2 || super_min >= (1u << 19));
    const bool overlap_pockets = overlap_mode >= 2;
    hipStream_t side = overlap ? ws.side : stream;
    // Arena, task queue and lists: one set - a second one only where the rounds of a chunk run beside the next chunk's ligand kernel
    // (PMX_OVERLAP, off by default): in order on one stream a chunk is through with them when the next one starts.
    for (int ci = 0; ci < (overlap ? 2 : 1); ++ci) {
        ChunkSet &c = ws.set[ci];
        // (PMX_ARENA_MB / PMX_TASKQ_MB are per set; a smaller arena than asked for is slower - more trees walked by one
        // wavefront alone - never wrong, so it shrinks when memory is short: several streams each keep a workspace. A size that was
        // accepted after shrinking stands until the workspace is released: asking for the full size again on every call would
        // synchronise, free and fail again each time)
        // (32 / 64 conformer lanes: records of megabytes - 16 GB held the split trees of a 16 384-ligand chunk of the stress configuration
        // to within 3 %, and every tree past the end is walked by one wavefront alone)
        // ([MI355X] round 6: 64 GB at up to 16 lanes. SURVEY 8d-2's library puts 29 GB of split trees' tables into the arena per 1 M-ligand chunk at the old
        // budget, 17 GB at the new one; with a 16 GB arena 200 000 trees found it full and were walked by one wavefront each - the longest for 1.2 M
        // passes, the pass 3.4 s instead of 0.54 s. The part has 288 GB.)
        size_t arena_want = (size_t)std::max<long>(1, env_long("PMX_ARENA_MB", G >= 32 ? 32768 : 65536)) << 20;
        if (!c.arena && !std::getenv("PMX_ARENA_MB")) { // first allocation, no explicit size: at most a third of what the device has free
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 3 < arena_want) {
                arena_want = std::max<size_t>((size_t)1 << 30, (free_b / 3) & ~(((size_t)1 << 20) - 1));
                c.arena_shrunk_to = arena_want;
            } else {
                (void)hipGetLastError();
            }
        }
        const size_t arena_min = std::min<size_t>((size_t)1 << 30, arena_want);
        if (c.arena_shrunk_to) arena_want = std::min(arena_want, std::max(c.arena_shrunk_to, arena_min));
        const size_t asked = arena_want;
        int rc2 = grow(&c.arena, &c.arena_bytes, arena_want, stream, ws.side, arena_min);
        if (rc2) return rc2;
        if (c.arena_bytes < asked) c.arena_shrunk_to = c.arena_bytes;
        rc2 = grow(&c.queue, &c.queue_bytes, (size_t)std::max<long>(1, env_long("PMX_TASKQ_MB", (G >= 32 ? 1024L : 2048L) * std::max(1, G / 8))) << 20, stream, ws.side);
        if (rc2) return rc2;
        rc2 = grow(&c.lists, &c.lists_bytes, (size_t)super_max * 12, stream, ws.side);
        if (rc2) return rc2;
    }
    const double lig_share = std::min(0.9, std::max(0.1, std::atof(std::getenv("PMX_LIG_SHARE") ? std::getenv("PMX_LIG_SHARE") : "0.5")));

    if (g_profiling) HIPCHECK(hipEventRecord(ws.ev[0], stream));
    ws.ctl_used[0] = ws.ctl_used[1] = false;
    ws.set[0].pending = ws.set[1].pending = false;
    uint64_t seq = 0;
    for (int m = 0; m < n_models; ++m) {
        PocketPlan &pl = plan[(size_t)m];
        ScreenParams p = pl.p;
        const uint32_t super = pl.super;
        const uint32_t full = pl.waves_per_cu;
        const uint32_t lig_waves = std::min(full - 1, std::max(1u, (uint32_t)(full * lig_share + 0.5)));
        for (uint64_t lo = 0; lo < count; lo += super, ++seq) {
            const bool last_of_call = seq + 1 == n_chunks;
            // a chunk that has the device to itself: the call's first / last, or (by default) a pocket's first / last
            const bool first_chunk = seq == 0 || (!overlap_pockets && lo == 0), last_chunk = last_of_call || (!overlap_pockets && lo + super >= count);
            const int ci = overlap ? (int)(seq & 1) : 0;
            ChunkSet &c = ws.set[ci];
            p.lo = (uint32_t)lo;
            p.hi = (uint32_t)std::min<uint64_t>(count, lo + super);
            p.ctl = c.ctl;
            p.arena = c.arena;
            p.arena_bytes = std::min<unsigned long long>(c.arena_bytes, (1ull << 36) - 4096);
            p.ovf_list = c.lists;
            p.carry_list = c.lists + super;
            p.heavy_list = c.lists + 2 * (size_t)super;
            p.list_cap = super;
            p.queue = c.queue;
            p.qcap = (uint32_t)std::min<size_t>(c.queue_bytes / task_rec_bytes<G>() / kShards, 0x3fffffffu / kShards);
            p.totbuf = ws.totbuf;
            p.pabuf = ws.pabuf;
            p.pa_bytes = pl.pa_bytes;
            // this set's last chunk (two chunks ago) has to be through its rounds
            if (overlap && c.pending) HIPCHECK(hipStreamWaitEvent(stream, c.tasks_done, 0));
            if (overlap && !overlap_pockets && lo == 0 && seq > 0 && ws.set[(seq - 1) & 1].pending) // (the rounds of the pocket before)
                HIPCHECK(hipStreamWaitEvent(stream, ws.set[(seq - 1) & 1].tasks_done, 0));
            if (g_profiling && last_of_call) HIPCHECK(hipEventRecord(ws.ev[1], stream));
            ws.ligands_last = p.hi - p.lo;
            ctl_clear_kernel<<<dim3((sizeof(Ctl) / 4 + 255) / 256), dim3(256), 0, stream>>>(c.ctl, ws.ctl_used[ci] ? 0 : 1);
            ws.ctl_used[ci] = true;
            // the first chunk's ligand kernel has the device to itself, the others share it with the rounds of the chunk before
            const uint32_t lig_grid = (uint32_t)ws.num_cu * ((overlap && !first_chunk) ? lig_waves : full);
            const uint32_t task_grid = (uint32_t)ws.num_cu * ((overlap && !last_chunk) ? std::min<uint32_t>(pl.task_waves_per_cu, full - lig_waves) : pl.task_waves_per_cu);
            auto launch = [&](int mode, uint32_t blocks, hipStream_t on) {
                p.mode = mode;
                if (debug_kernels) debug_ok &= pmx_debug::launch_ligand(G, exact, tails, blocks, pl.lds, on, &p, sizeof p);
                else if (tails) ligand_kernel<G, false, true><<<dim3(blocks), dim3(64), pl.lds, on>>>(p);
                else ligand_kernel<G, false, false><<<dim3(blocks), dim3(64), pl.lds, on>>>(p);
            };
            p.last_round = 0;
            p.budget = lig_budget;
            // every ligand whose tables fit a slice
            p.slices = ws.slices;
            p.slice_bytes = pl.slice_bytes;
            launch(0, lig_grid, stream);
            // the others with large slices (fewer wavefronts)
            p.slices = ws.big;
            p.slice_bytes = pl.big_bytes;
            launch(1, std::min(pl.big_grid, lig_grid), stream);
            // and what exceeds those from the arena; ligands that find it full (of the tables of over-budget trees, or of each
            // other) are listed in the storage of the overflow list, which is done with
            p.retry_out = c.lists;
            p.retry_slot = 0;
            launch(2, std::min(pl.big_grid, lig_grid), stream);
            if (g_profiling && last_of_call) HIPCHECK(hipEventRecord(ws.ev[2], stream));
            if (overlap) {
                HIPCHECK(hipEventRecord(c.lig_done, stream));
                HIPCHECK(hipStreamWaitEvent(side, c.lig_done, 0));
            }
            if (g_profiling && last_of_call) HIPCHECK(hipEventRecord(ws.ev[4], side));
            // the subtrees the over-budget walkers queued, and the ones those queue in turn: a fixed number of rounds, each a snapshot of
            // the queue and one persistent launch (an empty round exits at once); the last round walks everything to its end
            if (!totals_in_lds<G>()) p.totbuf = ws.totbuf + ws.totbuf_bytes / 2;
            if (ws.pabuf) p.pabuf = ws.pabuf + ws.pabuf_bytes / 2;
            auto rounds_and_finalize = [&]() {
                for (int r = 0; r < rounds; ++r) {
                    // Late rounds hold few subtrees (fewer than wavefronts): what they cost is their longest walk, i.e. the budget. Halving
                    // it from round PMX_TASK_DECAY_FROM on spreads a deep subtree over the idle wavefronts sooner.
                    p.budget = r < task_decay_from ? task_budget : std::max<uint32_t>(task_budget_min, task_budget >> std::min(r - task_decay_from + 1, 16));
                    p.last_round = r + 1 == rounds ? 1u : 0u;
                    round_kernel<<<dim3(1), dim3(64), 0, side>>>(c.ctl, p.qcap);
                    if (debug_kernels) debug_ok &= pmx_debug::launch_task(G, task_grid, pl.lds, side, &p, sizeof p);
                    else task_kernel<G><<<dim3(task_grid), dim3(64), pl.lds, side>>>(p);
                }
                finalize_kernel<G><<<dim3((super + 255) / 256), dim3(256), 0, side>>>(p);
                p.last_round = 0;
                p.budget = lig_budget;
            };
            rounds_and_finalize();
            // Ligands the arena pass had no room for, with the arena to themselves (only models and libraries whose largest tables
            // exceed a large slice ever get here; `side` is the caller's stream then): PMX_ARENA_RETRIES times, the last time reporting what
            // still does not fit.
            if (pl.worst_bytes > pl.big_bytes) {
                for (int t = 0; t < arena_retries; ++t) {
                    retry_prep_kernel<<<dim3(1), dim3(64), 0, side>>>(c.ctl, (uint32_t)(t + 1) & 1u);
                    p.retry_in = (t & 1) == 0 ? c.lists : c.lists + super;
                    p.retry_out = t + 1 == arena_retries ? nullptr : ((t & 1) == 0 ? c.lists + super : c.lists);
                    p.retry_slot = (uint32_t)(t + 1) & 1u;
                    p.totbuf = ws.totbuf;
                    p.pabuf = ws.pabuf;
                    launch(3, std::min(pl.big_grid, lig_grid), side);
                    if (!totals_in_lds<G>()) p.totbuf = ws.totbuf + ws.totbuf_bytes / 2;
                    if (ws.pabuf) p.pabuf = ws.pabuf + ws.pabuf_bytes / 2;
                    rounds_and_finalize();
                }
                p.retry_in = nullptr;
            }
            if (g_profiling && last_of_call) HIPCHECK(hipEventRecord(ws.ev[5], side));
            if (overlap) {
                HIPCHECK(hipEventRecord(c.tasks_done, side));
                c.pending = true;
            }
        }
    }
    HIPCHECK(hipGetLastError());
    if (!debug_ok) return fail(PMX_ERR_INVALID, "the validation kernels (pmx_screen_debug.hip) do not match this build's parameter block");
    if (overlap) // the call ends on the caller's stream
        for (ChunkSet &c : ws.set)
            if (c.pending) HIPCHECK(hipStreamWaitEvent(stream, c.tasks_done, 0));
    if (g_profiling) HIPCHECK(hipEventRecord(ws.ev[3], stream));
    ws.ev_valid = g_profiling != 0;
    ws.last_stream = stream;
    return PMX_OK;
}

// Statistics of the last call on this thread's workspace: synchronises the stream the call ran on.
static int screen_stats(pmx_score_stats *out) {
    const std::shared_ptr<ScreenWs> w = g_last_screen;
    if (!w) return PMX_OK;
    std::lock_guard<std::mutex> lock(w->mu);
    if (w->released || !w->num_cu) return PMX_OK; // (the workspace was released after the call: its counters went with it)
    HIPCHECK(hipSetDevice(g_last_device));
    HIPCHECK(hipStreamSynchronize(w->last_stream));
    unsigned long long st[kStatWords] = {0};
    std::vector<unsigned char> host(sizeof(Ctl));
    *out = pmx_score_stats{};
    for (int ci = 0; ci < 2; ++ci) {
        if (!w->ctl_used[ci]) continue;
        HIPCHECK(hipMemcpy(host.data(), w->set[ci].ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        const Ctl *c = reinterpret_cast<const Ctl *>(host.data());
        for (int sh = 0; sh < kScreenStatShards; ++sh)
            for (int i = 0; i < kStatWords; ++i) st[i] = (i == 5) ? std::max(st[i], c->stats[sh][i]) : st[i] + c->stats[sh][i];
        out->queue_overflow |= c->qflag;
        out->arena_bytes = std::max<uint64_t>(out->arena_bytes, c->arena_top);
        out->arena_capacity = std::max<uint64_t>(out->arena_capacity, w->set[ci].arena_bytes);
    }
    out->n_frames = st[0];
    out->n_passes = st[1];
    out->n_heavy = st[2];
    out->n_items = st[3];
    out->n_exact_cells = st[4];
    out->max_passes = st[5];
    out->n_tasks = st[6];
    out->n_slice_overflow = st[7] & 0xffffffffull;
    out->n_probes = st[7] >> 32;
    out->n_probe_passes = st[15];
    out->n_exported = st[14];
    out->n_exact_values = st[13];
    for (int i = 0; i < 6; ++i) out->dbg[i] = st[16 + i];
    out->n_path_bounds = st[22];
    out->n_path_drops = st[23];
    out->n_dead_entries = st[24];
    out->ticks_scan = st[8], out->ticks_tables = st[9], out->ticks_bounds = st[10], out->ticks_walk = st[11], out->ticks_alive = st[12];
    out->ligands_last = w->ligands_last;
    if (w->ev_valid) {
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, w->ev[0], w->ev[3]));
        out->ms_total = ms;
        HIPCHECK(hipEventElapsedTime(&ms, w->ev[1], w->ev[2]));
        out->ms_ligand = ms;
        HIPCHECK(hipEventElapsedTime(&ms, w->ev[4], w->ev[5]));
        out->ms_tasks = ms;
    }
    return PMX_OK;
}

static int next_pow2(int x) {
    int g = 1;
    while (g < x) g <<= 1;
    return g;
}

static int score_any(const pmx_model *const *models, int n_models, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
                     uint64_t count, void *scores_dev, bool scores_f64, int32_t *status_dev, void *stream_) {
    if (!models || n_models < 0 || !lib || !weights || (!scores_dev && count && n_models)) return fail(PMX_ERR_INVALID, "null argument");
    for (int i = 0; i < n_models; ++i) {
        if (!models[i]) return fail(PMX_ERR_INVALID, "null model");
        if (models[i]->device != lib->device) return fail(PMX_ERR_INVALID, "model and library live on different devices");
    }
    if (first > lib->info.n_ligands || count > lib->info.n_ligands - first) return fail(PMX_ERR_INVALID, "ligand range out of bounds");
    g_stats = pmx_score_stats{};
    g_last_screen.reset();
    if (count == 0 || n_models == 0) return PMX_OK;
    HIPCHECK(hipSetDevice(lib->device));
    Weights W;
    for (int t = 0; t < PMX_NUM_TYPES; ++t) W.w[t] = weights[t];
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int G = next_pow2(std::max(1, std::min(lib->info.max_conformers, PMX_MAX_CONFORMERS)));
    int rc = PMX_OK;
    std::shared_ptr<ScreenWs> ws;
    std::unique_lock<std::mutex> lock;
    for (;;) { // one call at a time enqueues on a (device, stream) workspace
        ws = ensure_screen(lib->device, stream);
        lock = std::unique_lock<std::mutex>(ws->mu);
        if (!ws->released) break;
        lock.unlock(); // released while this call waited: the map holds a fresh one (or will make one)
    }
    switch (G) {
    case 1: rc = score_screen<1>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    case 2: rc = score_screen<2>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    case 4: rc = score_screen<4>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    case 8: rc = score_screen<8>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    case 16: rc = score_screen<16>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    case 32: rc = score_screen<32>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    default: rc = score_screen<64>(models, n_models, lib, W, first, count, scores_dev, scores_f64, status_dev, stream, *ws); break;
    }
    g_last_screen = ws;
    g_last_device = lib->device;
    return rc;
}

extern "C" int pmx_score_multi(const pmx_model *const *models, int n_models, const pmx_library *lib,
                               const float weights[PMX_NUM_TYPES], uint64_t first, uint64_t count, float *scores_dev,
                               int32_t *status_dev, void *stream) {
    return score_any(models, n_models, lib, weights, first, count, scores_dev, false, status_dev, stream);
}

extern "C" int pmx_score(const pmx_model *model, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
                         uint64_t count, float *scores_dev, int32_t *status_dev, void *stream) {
    if (!model) return fail(PMX_ERR_INVALID, "null argument");
    return score_any(&model, 1, lib, weights, first, count, scores_dev, false, status_dev, stream);
}

// The same with the score as the float64 the reference returns (graph_match.py:109: `float(np.mean(...))` of float64 maxima).
extern "C" int pmx_score_multi_f64(const pmx_model *const *models, int n_models, const pmx_library *lib,
                                   const float weights[PMX_NUM_TYPES], uint64_t first, uint64_t count, double *scores_dev,
                                   int32_t *status_dev, void *stream) {
    return score_any(models, n_models, lib, weights, first, count, scores_dev, true, status_dev, stream);
}

extern "C" int pmx_score_f64(const pmx_model *model, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
                             uint64_t count, double *scores_dev, int32_t *status_dev, void *stream) {
    if (!model) return fail(PMX_ERR_INVALID, "null argument");
    return score_any(&model, 1, lib, weights, first, count, scores_dev, true, status_dev, stream);
}


// Frees the cached scoring workspaces of `device` (table arenas, task queues, class lists, fused-engine buffers): they are
// grown on demand and kept between calls, which is what a screening loop wants and what a long-lived host program that
// is done screening does not.
int pmx_topk_release(int device);
extern "C" int pmx_release_workspaces(int device) {
    HIPCHECK(hipSetDevice(device));
    HIPCHECK(hipDeviceSynchronize());
    std::vector<std::shared_ptr<ScreenWs>> taken;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (auto it = g_screen.begin(); it != g_screen.end();) {
            if (it->first.first != device) {
                ++it;
                continue;
            }
            taken.push_back(std::move(it->second));
            it = g_screen.erase(it);
        }
    }
    for (auto &w : taken) {
        std::lock_guard<std::mutex> wl(w->mu); // a call that is enqueuing on this workspace finishes first
        HIPCHECK(hipDeviceSynchronize());       // ... and what it enqueued
        w->free_buffers();
        w->released = true;
    }
    return pmx_topk_release(device);
}

// error hook for pmx_topk.hip (keeps the thread-local message in one translation unit)
int pmx_topk_fail(int code, const char *msg) { return fail(code, "%s", msg); }

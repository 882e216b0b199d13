// valu_calib.hip - calibration microbenchmarks for the rooflines quoted in bench.py / DESIGN.md (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/calib/valu_calib.hip -o tools/calib/valu_calib
//   ./tools/calib/valu_calib > profiles/r2_valu_calibration.json
//
// Measures, on all CUs with W waves per SIMD:
//   fma        independent v_fma_f32 chains                   -> plain VALU lane-operations per second
//   pk_fma     independent v_pk_fma_f32 chains (2 floats/lane) -> packed rate
//   exp        independent v_exp_f32                           -> transcendental rate
//   term_reg   the Gaussian term of match_utils.py:55-68 with its edge parameters in registers
//              (sub, mul, mul, exp, fma, cmp, addc = 7 VALU)  -> terms per second ceiling
//   term_smem  the same term with {mean, s, T, w} fetched by scalar loads (uniform address, 16 B per term)
//   term_lds   the same with a wave-uniform ds_read_b128 per term (LDS broadcast)
//   term_lds8  per-lane ds_read_b128, reused for 8 conformers held in registers
// Every figure is "lane-terms per second" (64 per wave-instruction) or lane-ops per second.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

constexpr int ITERS = 4096;

__global__ void k_fma(float *out, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k_pkfma(float *out, float a, float b) {
    float2v x[16];
    const float2v av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = float2v{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_exp(float *out, float a) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = -(float)(threadIdx.x & 7) * 0.01f - i * a;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.0f; // exp + one plain op
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// exp alone is not measurable without a consumer; k_exp does (exp, sub) pairs, k_sub does the sub alone:
__global__ void k_sub(float *out, float a) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = -(float)(threadIdx.x & 7) * 0.01f - i * a;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = x[i] * a - 1.0f; // mul, sub (contraction off)
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#pragma clang fp contract(off)
__device__ __forceinline__ void term(float d, float mean, float s, float T, float w, float &acc, unsigned &np) {
    const float t = __builtin_fabsf(d - mean);
    const float q = t * s;
    acc = __builtin_fmaf(w, __builtin_amdgcn_exp2f(-(q * q)), acc);
    np += (t <= T) ? 1u : 0u;
}

// edge parameters in registers: 8 terms per pass, parameters perturbed per pass so nothing folds
__global__ void k_term_reg(float *out, const float4 *tab, int n_terms) {
    float d = 3.0f + 0.01f * (float)threadIdx.x;
    float4 e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tab[i];
    float acc = 0.f;
    unsigned np = 0;
    for (int it = 0; it < n_terms / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) term(d, e[i].x, e[i].y, e[i].z, e[i].w, acc, np);
        d += 1e-4f; // one extra plain op per 8 terms: keeps the terms from being hoisted out of the loop
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)np;
}

// scalar loads: a uniform walk over a table of `n_tab` entries, 16 B per term
__global__ void k_term_smem(float *out, const float4 *__restrict__ tab, int n_tab, int n_terms) {
    const float d = 3.0f + 0.01f * (float)threadIdx.x;
    float acc = 0.f;
    unsigned np = 0;
    int pos = (blockIdx.x * 37) % n_tab;
    for (int it = 0; it < n_terms / 12; ++it) {
        const float4 *seg = reinterpret_cast<const float4 *>(__builtin_assume_aligned(tab + pos, 16));
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float4 e = seg[i];
            term(d, e.x, e.y, e.z, e.w, acc, np);
        }
        pos += 12;
        if (pos + 12 > n_tab) pos = 0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)np;
}

// LDS broadcast: every lane reads the same 16 bytes
__global__ void k_term_lds(float *out, const float4 *__restrict__ tab, int n_tab, int n_terms) {
    extern __shared__ float4 lt[];
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
    const float d = 3.0f + 0.01f * (float)threadIdx.x;
    float acc = 0.f;
    unsigned np = 0;
    int pos = ((threadIdx.x >> 6) * 37) % n_tab;
    for (int it = 0; it < n_terms / 12; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float4 e = lt[pos + i];
            term(d, e.x, e.y, e.z, e.w, acc, np);
        }
        pos += 12;
        if (pos + 12 > n_tab) pos = 0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)np;
}

// per-lane LDS read reused for 8 conformers in registers (lanes = columns)
__global__ void k_term_lds8(float *out, const float4 *__restrict__ tab, int n_tab, int n_terms) {
    extern __shared__ float4 lt[];
    for (int i = threadIdx.x; i < n_tab; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
    float d[8], acc[8];
    unsigned np[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        d[c] = 3.0f + 0.01f * (float)threadIdx.x + 0.1f * c;
        acc[c] = 0.f;
        np[c] = 0;
    }
    int col = threadIdx.x & 31;
    for (int it = 0; it < n_terms / 8; ++it) {
        const float4 e = lt[(it * 40 + col) % n_tab];
#pragma unroll
        for (int c = 0; c < 8; ++c) term(d[c], e.x, e.y, e.z, e.w, acc[c], np[c]);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c] + (float)np[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F &&launch, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate / 1e6;
    const int n_tab = 3072; // 48 KB of edge entries, the size of a 6OIM-like row-segment table
    std::vector<float4> host(n_tab + 16);
    for (int i = 0; i < n_tab + 16; ++i)
        host[i] = make_float4(2.0f + 0.001f * (i % 997), 0.5f + 0.0001f * (i % 89), 2.5f + 0.001f * (i % 13), 0.3f + 0.001f * (i % 7));
    float4 *tab;
    float *out;
    CHECK(hipMalloc(&tab, host.size() * sizeof(float4)));
    CHECK(hipMemcpy(tab, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)cus * 64 * 2048 * sizeof(float)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_term_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_term_lds8), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));

    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f,\n", prop.name, cus, clock_ghz);
    printf("  \"note\": \"lane-ops/s = 64 x wave instructions/s; spec_simd32 = CUs x 4 SIMD x 32 lanes x clock, spec_simd16 = half of it\",\n");
    printf("  \"spec_simd32_lane_ops\": %.4e,\n  \"results\": [\n", cus * 4.0 * 32.0 * clock_ghz * 1e9);
    bool first = true;
    auto emit = [&](const char *name, int wps, double ms, double lane_items, const char *unit) {
        printf("%s    {\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"rate\": %.4e, \"unit\": \"%s\"}", first ? "" : ",\n", name, wps, ms,
               lane_items / (ms * 1e-3), unit);
        first = false;
        fflush(stdout);
    };
    for (int wps : {1, 2, 4, 8}) {
        const int block = 256;                 // 4 waves: one per SIMD
        const int grid = cus * wps;            // wps blocks per CU
        const double lanes = (double)grid * block;
        double ms;
        ms = time_ms([&] { k_fma<<<grid, block>>>(out, 1.0001f, 0.5f); });
        emit("fma", wps, ms, lanes * ITERS * 16, "lane-ops/s (v_fma_f32)");
        ms = time_ms([&] { k_pkfma<<<grid, block>>>(out, 1.0001f, 0.5f); });
        emit("pk_fma", wps, ms, lanes * ITERS * 16, "lane-instructions/s (v_pk_fma_f32, 2 fma each)");
        ms = time_ms([&] { k_exp<<<grid, block>>>(out, 0.001f); });
        emit("exp_plus_sub", wps, ms, lanes * ITERS * 16, "lane-(exp,sub) pairs/s");
        ms = time_ms([&] { k_sub<<<grid, block>>>(out, 0.999f); });
        emit("mul_plus_sub", wps, ms, lanes * ITERS * 16, "lane-(mul,sub) pairs/s");
        const int n_terms = 12 * 8 * 512;
        ms = time_ms([&] { k_term_reg<<<grid, block>>>(out, tab, n_terms); });
        emit("term_reg", wps, ms, lanes * n_terms, "lane-terms/s");
        ms = time_ms([&] { k_term_smem<<<grid, block>>>(out, tab, n_tab, n_terms); });
        emit("term_smem", wps, ms, lanes * n_terms, "lane-terms/s");
        ms = time_ms([&] { k_term_lds<<<grid, block, n_tab * sizeof(float4)>>>(out, tab, n_tab, n_terms); });
        emit("term_lds_broadcast", wps, ms, lanes * n_terms, "lane-terms/s");
        ms = time_ms([&] { k_term_lds8<<<grid, block, n_tab * sizeof(float4)>>>(out, tab, n_tab, n_terms); });
        emit("term_lds_per_lane_x8", wps, ms, lanes * n_terms, "lane-terms/s");
    }
    printf("\n  ]\n}\n");
    return 0;
}

// Microbenchmark (not part of the product): which evaluation of alpha = min(0.99, o * exp(power)) and of T / (1 - alpha)
// reproduces the HOST's bits (glibc expf, IEEE division -- what the CPU restatement computes), and what each costs in
// VALU issue time at 8 waves per SIMD.
//   fast    : o * v_exp_f32(power * log2 e)                           (round 2's blend kernels)
//   hilo    : v_exp_f32(ph) * (1 + c), ph = fl(power * log2 e), c = power - ph ln 2 in two fma steps
//   ocml    : HIP's expf (range reduction + v_exp_f32 + ldexp)
//   ref     : csrc/exact_math.h exp_ref -- glibc's algorithm in the double pipe (Horner form)
//   glibc   : the same, operation for operation as glibc associates it
//   division: v_rcp_f32 product / residual-corrected (div_ref) / with a Newton step on the reciprocal first / IEEE sequence
// build: hipcc --offload-arch=gfx950 -O3 -o exp_variants exp_variants.hip ; run: ./exp_variants
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../diff-gaussian-rasterization_amd/csrc/exact_math.h"
using namespace dgr;

__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float exp_hilo(float x) {
    const float ph = x * 1.4426950408889634f;
    float c = __builtin_fmaf(-ph, 0.693147182464599609375f, x);
    c = __builtin_fmaf(-ph, -1.90465429995776804525e-09f, c);
    const float r = __builtin_amdgcn_exp2f(ph);
    return __builtin_fmaf(r, c, r);
}
__device__ __forceinline__ float exp_glibc(float x, const uint64_t* tab) {
#pragma clang fp contract(off)
    constexpr double INVLN2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = INVLN2N * (double)x;
    const double kd0 = z + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd0);
    const double kd = kd0 - SHIFT;
    const double r = z - kd;
    uint64_t t = tab[ki & 31u];
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    const double zz = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    return (float)(y * s);
}
__device__ __forceinline__ float div_newton(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    const float y = __builtin_fmaf(e, y0, y0);
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
template <int V>
__device__ __forceinline__ float exp_v(float x, const uint64_t* tab) {
    return V == 0 ? exp_fast(x) : V == 1 ? exp_hilo(x) : V == 2 ? expf(x) : V == 3 ? exp_ref(x, tab) : exp_glibc(x, tab);
}
template <int V>
__device__ __forceinline__ float div_v(float a, float b) {
    float inv;
    return V == 0 ? a * __builtin_amdgcn_rcpf(b) : V == 1 ? div_ref(a, b, inv) : V == 2 ? a / b : div_newton(a, b);
}

template <int V>
__global__ void __launch_bounds__(256) acc_exp(const float* x, const float* o, float* out, int n) {
    __shared__ uint64_t tab[32];
    exp_ref_table_fill(tab, threadIdx.x);
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = fminf(0.99f, o[i] * exp_v<V>(x[i], tab));
}
template <int V>
__global__ void __launch_bounds__(256) acc_div(const float* a, const float* b, float* out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = div_v<V>(a[i], b[i]);
}

constexpr int ITER = 2048;
// issue cost: 4 independent evaluations per iteration, arguments kept in range by a cheap dependent update
template <int V>
__global__ void __launch_bounds__(256, 8) time_exp(float* out, float seed) {
    __shared__ uint64_t tab[32];
    exp_ref_table_fill(tab, threadIdx.x);
    __syncthreads();
    float x0 = -seed - 1e-3f * threadIdx.x, x1 = x0 - 0.1f, x2 = x0 - 0.2f, x3 = x0 - 0.3f, acc = 0.f;
    for (int i = 0; i < ITER; i++) {
        const float e0 = exp_v<V>(x0, tab), e1 = exp_v<V>(x1, tab), e2 = exp_v<V>(x2, tab), e3 = exp_v<V>(x3, tab);
        acc += (e0 + e1) + (e2 + e3);
        x0 = __builtin_fmaf(e0, -1e-3f, x0 * 0.999f); x1 = __builtin_fmaf(e1, -1e-3f, x1 * 0.999f);
        x2 = __builtin_fmaf(e2, -1e-3f, x2 * 0.999f); x3 = __builtin_fmaf(e3, -1e-3f, x3 * 0.999f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int V>
__global__ void __launch_bounds__(256, 8) time_div(float* out, float seed) {
    float t0 = seed, t1 = seed * 0.9f, t2 = seed * 0.8f, t3 = seed * 0.7f;
    const float b = 0.97f + 1e-4f * threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        t0 = div_v<V>(t0, b) * 0.97f; t1 = div_v<V>(t1, b) * 0.97f; t2 = div_v<V>(t2, b) * 0.97f; t3 = div_v<V>(t3, b) * 0.97f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = (t0 + t1) + (t2 + t3);
}
// the baseline of the timing loops (the argument updates alone)
__global__ void __launch_bounds__(256, 8) time_exp_base(float* out, float seed) {
    float x0 = -seed - 1e-3f * threadIdx.x, x1 = x0 - 0.1f, x2 = x0 - 0.2f, x3 = x0 - 0.3f, acc = 0.f;
    for (int i = 0; i < ITER; i++) {
        const float e0 = x0 * 0.5f, e1 = x1 * 0.5f, e2 = x2 * 0.5f, e3 = x3 * 0.5f;
        acc += (e0 + e1) + (e2 + e3);
        x0 = __builtin_fmaf(e0, -1e-3f, x0 * 0.999f); x1 = __builtin_fmaf(e1, -1e-3f, x1 * 0.999f);
        x2 = __builtin_fmaf(e2, -1e-3f, x2 * 0.999f); x3 = __builtin_fmaf(e3, -1e-3f, x3 * 0.999f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K, typename... A>
static float timed(K k, int blocks, A... a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a...);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

static void report(const char* name, const std::vector<float>& got, const std::vector<float>& ref) {
    size_t bad = 0; long maxulp = 0;
    for (size_t i = 0; i < ref.size(); i++) {
        int32_t a, b; memcpy(&a, &got[i], 4); memcpy(&b, &ref[i], 4);
        if (a != b) { bad++; long d = labs((long)a - (long)b); if (d > maxulp) maxulp = d; }
    }
    printf("  %-8s differs from the host on %10zu of %zu (%.3e), largest difference %ld ulp\n", name, bad, ref.size(), (double)bad / ref.size(), maxulp);
}

int main() {
    const int n = 1 << 26;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> ux(-2.9f, 0.0f), uo(0.06f, 1.0f), ut(1e-4f, 1.0f), ua(15.f / 255.f, 0.99f);
    std::vector<float> x(n), o(n), ref(n), got(n), ta(n), tb(n);
    for (int i = 0; i < n; i++) { x[i] = ux(rng); o[i] = uo(rng); ref[i] = fminf(0.99f, o[i] * expf(x[i])); }
    float *dx, *dout, *dother;
    hipMalloc(&dx, n * 4); hipMalloc(&dother, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dother, o.data(), n * 4, hipMemcpyHostToDevice);
    printf("alpha = min(0.99, o * exp(power)), power in [-2.9, 0], o in [0.06, 1], %d samples\n", n);
#define ACC(V, NAME) hipLaunchKernelGGL(acc_exp<V>, dim3(2048), dim3(256), 0, 0, dx, dother, dout, n); \
    hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost); report(NAME, got, ref);
    ACC(0, "fast") ACC(1, "hilo") ACC(2, "ocml") ACC(3, "ref") ACC(4, "glibc")
    // exp alone (o = 1)
    for (int i = 0; i < n; i++) { o[i] = 1.0f; ref[i] = fminf(0.99f, expf(x[i])); }
    hipMemcpy(dother, o.data(), n * 4, hipMemcpyHostToDevice);
    printf("exp(power) alone (o = 1)\n");
    ACC(0, "fast") ACC(1, "hilo") ACC(2, "ocml") ACC(3, "ref") ACC(4, "glibc")
    // division T / (1 - alpha)
    for (int i = 0; i < n; i++) { ta[i] = ut(rng); tb[i] = 1.0f - ua(rng); ref[i] = ta[i] / tb[i]; }
    hipMemcpy(dx, ta.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dother, tb.data(), n * 4, hipMemcpyHostToDevice);
    printf("T / (1 - alpha), T in [1e-4, 1], alpha in [15/255, 0.99]\n");
#define ACD(V, NAME) hipLaunchKernelGGL(acc_div<V>, dim3(2048), dim3(256), 0, 0, dx, dother, dout, n); \
    hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost); report(NAME, got, ref);
    ACD(0, "rcp") ACD(1, "div_ref") ACD(3, "newton") ACD(2, "ieee")

    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int blocks = 256 * 8;
    const double per = 1e-3 * clk_khz * 1e3 / (8.0 * ITER * 4);  // cycles per evaluation per SIMD at 8 waves per SIMD
    const float base = timed(time_exp_base, blocks, dout, 1.0f);
    printf("issue cost at 8 waves per SIMD, %d MHz (cycles per wave-evaluation per SIMD, loop baseline %.1f subtracted)\n", clk_khz / 1000, base * per);
    printf("  exp fast %.1f  hilo %.1f  ocml %.1f  ref %.1f  glibc %.1f\n", (timed(time_exp<0>, blocks, dout, 1.0f) - base) * per,
           (timed(time_exp<1>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<2>, blocks, dout, 1.0f) - base) * per,
           (timed(time_exp<3>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<4>, blocks, dout, 1.0f) - base) * per);
    printf("  div (incl. one v_mul) rcp %.1f  div_ref %.1f  newton %.1f  ieee %.1f\n", timed(time_div<0>, blocks, dout, 0.5f) * per,
           timed(time_div<1>, blocks, dout, 0.5f) * per, timed(time_div<3>, blocks, dout, 0.5f) * per, timed(time_div<2>, blocks, dout, 0.5f) * per);
    return 0;
}

// Microbenchmark (not part of the product): an fp32-only expf for the blend kernels' alpha -- range reduction by
// k = rint(x log2 e), r = x - k ln2 (two fused steps), a degree-6 polynomial with c0 = c1 = 1, c2 = 1/2 (inline constants),
// ldexp -- against exact_math.h's double-pipe form (exp_glibc): bits against the same operation sequence on the host,
// and issue cost at 8 waves per SIMD with the constants in VGPRs / where the compiler puts them.  Also the issue rate of
// the instruction classes the backward pair loop could trade its 4.2-cycle selects for.
// build: hipcc --offload-arch=gfx950 -O3 -o exp_poly32 exp_poly32.hip ; run: ./exp_poly32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../diff-gaussian-rasterization_amd/csrc/exact_math.h"
using namespace dgr;

// host restatement of the same sequence (what oracle/dgr_oracle.cpp evaluates)
static float exp_p32_host(float x) {
    x = fmaxf(x, -104.0f);
    const float kf = rintf(x * 0x1.715476p+0f);
    float r = fmaf(kf, -0x1.62e430p-1f, x);
    r = fmaf(kf, 0x1.05c610p-29f, r);
    float p = fmaf(0x1.6b6e26p-10f, r, 0x1.122faep-7f);
    p = fmaf(p, r, 0x1.555688p-5f);
    p = fmaf(p, r, 0x1.5554a4p-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(fmaf(p, r, 1.0f), (int)kf);
}

struct K32 { float log2e, nln2hi, nln2lo, c3, c4, c5, c6, lo; };
__device__ __forceinline__ float vconst(float v) { float r; asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; }
template <bool VG>
__device__ __forceinline__ K32 k32() {
    K32 k = {0x1.715476p+0f, -0x1.62e430p-1f, 0x1.05c610p-29f, 0x1.5554a4p-3f, 0x1.555688p-5f, 0x1.122faep-7f, 0x1.6b6e26p-10f, -104.0f};
    if (VG) { k.log2e = vconst(k.log2e); k.nln2hi = vconst(k.nln2hi); k.nln2lo = vconst(k.nln2lo); k.c3 = vconst(k.c3); k.c4 = vconst(k.c4);
              k.c5 = vconst(k.c5); k.c6 = vconst(k.c6); k.lo = vconst(k.lo); }
    return k;
}
// form A: v_rndne + v_cvt_i32 + v_ldexp (the first candidate; exact_math.h: exp_p32 is form B below with literal constants)
template <bool CLAMP>
__device__ __forceinline__ float exp_p32_a(float x, const K32& k) {
#pragma clang fp contract(off)
    if (CLAMP) x = fmaxf(x, k.lo);
    const float kf = __builtin_rintf(x * k.log2e);
    float r = __builtin_fmaf(kf, k.nln2hi, x);
    r = __builtin_fmaf(kf, k.nln2lo, r);
    float p = __builtin_fmaf(k.c6, r, k.c5);
    p = __builtin_fmaf(p, r, k.c4);
    p = __builtin_fmaf(p, r, k.c3);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __builtin_ldexpf(__builtin_fmaf(p, r, 1.0f), (int)kf);
}
// exponent-field add instead of ldexp (argument clamped at -86.6: no denormal results)
__device__ __forceinline__ float exp_p32_expadd(float x, const K32& k, float magic) {
#pragma clang fp contract(off)
    x = fmaxf(x, k.lo);
    const float t = __builtin_fmaf(x, k.log2e, magic);
    const float kf = t - magic;
    float r = __builtin_fmaf(kf, k.nln2hi, x);
    r = __builtin_fmaf(kf, k.nln2lo, r);
    float p = __builtin_fmaf(k.c6, r, k.c5);
    p = __builtin_fmaf(p, r, k.c4);
    p = __builtin_fmaf(p, r, k.c3);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    const float y = __builtin_fmaf(p, r, 1.0f);
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, t) << 23) + __builtin_bit_cast(uint32_t, y));
}

// form B: rounding by the magic shift, exponent-field add of k + 64, one multiply by 2^-64 (rounds denormal results once)
template <bool CLAMP>
__device__ __forceinline__ float exp_p32_b(float x, const K32& k, float magic64, float two_m64) {
#pragma clang fp contract(off)
    if (CLAMP) x = fmaxf(x, k.lo);
    const float t = __builtin_fmaf(x, k.log2e, magic64);
    const float kf = t - magic64;
    float r = __builtin_fmaf(kf, k.nln2hi, x);
    r = __builtin_fmaf(kf, k.nln2lo, r);
    float p = __builtin_fmaf(k.c6, r, k.c5);
    p = __builtin_fmaf(p, r, k.c4);
    p = __builtin_fmaf(p, r, k.c3);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    const float y = __builtin_fmaf(p, r, 1.0f);
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, t) << 23) + __builtin_bit_cast(uint32_t, y)) * two_m64;
}
static float exp_p32_b_host(float x) {
    x = fmaxf(x, -104.0f);
    const float M = 12582976.0f;
    const float t = fmaf(x, 0x1.715476p+0f, M);
    const float kf = t - M;
    float r = fmaf(kf, -0x1.62e430p-1f, x);
    r = fmaf(kf, 0x1.05c610p-29f, r);
    float p = fmaf(0x1.6b6e26p-10f, r, 0x1.122faep-7f);
    p = fmaf(p, r, 0x1.555688p-5f);
    p = fmaf(p, r, 0x1.5554a4p-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    const float y = fmaf(p, r, 1.0f);
    uint32_t tb, yb; memcpy(&tb, &t, 4); memcpy(&yb, &y, 4);
    const uint32_t b = (tb << 23) + yb; float f; memcpy(&f, &b, 4);
    return f * 0x1p-64f;
}

template <int V>
__global__ void __launch_bounds__(256) acc_exp(const float* x, float* out, int n) {
    __shared__ uint64_t tab[32];
    exp_ref_table_fill(tab, threadIdx.x);
    __syncthreads();
    const K32 kv = k32<true>(), ks = k32<false>();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        out[i] = V == 0 ? exp_p32_a<true>(x[i], kv) : V == 1 ? exp_p32_a<true>(x[i], ks) : V == 3 ? exp_p32_b<true>(x[i], ks, 12582976.0f, 0x1p-64f) : V == 4 ? exp_p32<true>(x[i]) : exp_glibc<true>(x[i], tab);
}

constexpr int ITER = 2048;
template <int V>
__global__ void __launch_bounds__(256, 8) time_exp(float* out, float seed) {
    __shared__ uint64_t tab[32];
    exp_ref_table_fill(tab, threadIdx.x);
    __syncthreads();
    const K32 kv = k32<true>(), ks = k32<false>();
    K32 k86 = kv; k86.lo = vconst(-86.6f);
    const float magic = vconst(12582912.0f), magic64 = vconst(12582976.0f), twom64 = vconst(0x1p-64f);
    float x0 = -seed - 1e-3f * threadIdx.x, x1 = x0 - 0.1f, x2 = x0 - 0.2f, x3 = x0 - 0.3f, acc = 0.f;
    for (int i = 0; i < ITER; i++) {
        float e0, e1, e2, e3;
#define EV(F) e0 = F(x0); e1 = F(x1); e2 = F(x2); e3 = F(x3);
        if (V == 0) { EV([&](float x) { return exp_p32_a<true>(x, kv); }) }
        else if (V == 1) { EV([&](float x) { return exp_p32_a<true>(x, ks); }) }
        else if (V == 2) { EV([&](float x) { return exp_p32_a<false>(x, kv); }) }
        else if (V == 3) { EV([&](float x) { return exp_p32_expadd(x, k86, magic); }) }
        else if (V == 4) { EV([&](float x) { return exp_glibc<true>(x, tab); }) }
        else if (V == 7) { EV([&](float x) { return exp_p32_b<true>(x, ks, 12582976.0f, 0x1p-64f); }) }
        else if (V == 8) { EV([&](float x) { return exp_p32_b<false>(x, ks, 12582976.0f, 0x1p-64f); }) }
        else if (V == 10) { EV([&](float x) { return exp_p32<true>(x); }) }
        else if (V == 11) { EV([&](float x) { return exp_p32<false>(x); }) }
        else if (V == 9) { EV([&](float x) { return exp_p32_b<true>(x, kv, magic64, twom64); }) }
        else if (V == 5) { EV([&](float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }) }
        else { EV([&](float x) { return x * 0.5f; }) }
        acc += (e0 + e1) + (e2 + e3);
        x0 = __builtin_fmaf(e0, -1e-3f, x0 * 0.999f); x1 = __builtin_fmaf(e1, -1e-3f, x1 * 0.999f);
        x2 = __builtin_fmaf(e2, -1e-3f, x2 * 0.999f); x3 = __builtin_fmaf(e3, -1e-3f, x3 * 0.999f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// instruction rates: 4 independent chains of one instruction, 32 per asm block
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256, 8) rate(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float m = vconst(1.0001f), c = vconst(0.5f);
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));) }
        else if (MODE == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %4, 1.0\n v_fma_f32 %1, %1, %4, 1.0\n v_fma_f32 %2, %2, %4, 1.0\n v_fma_f32 %3, %3, %4, 1.0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        else if (MODE == 2) { REP8(asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 3) { REP8(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 4) { REP8(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0));) }
        else if (MODE == 5) { REP8(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        else if (MODE == 6) { REP8(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(c));) }
        else if (MODE == 7) { REP8(asm volatile("v_bfi_b32 %0, %4, %0, 0\n v_bfi_b32 %1, %4, %1, 0\n v_bfi_b32 %2, %4, %2, 0\n v_bfi_b32 %3, %4, %3, 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0));) }
        else if (MODE == 8) { REP8(asm volatile("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 9) { REP8(asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 10) { REP8(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 11) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 12) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 23, %1\n v_lshl_add_u32 %1, %1, 23, %2\n v_lshl_add_u32 %2, %2, 23, %3\n v_lshl_add_u32 %3, %3, 23, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 13) { REP8(asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 2.0, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 2.0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 14) { REP8(asm volatile("v_sub_f32 %0, 1.0, %0\n v_sub_f32 %1, 1.0, %1\n v_sub_f32 %2, 1.0, %2\n v_sub_f32 %3, 1.0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 15) { REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));) }
        else if (MODE == 16) { REP8(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %2, %0\n v_cvt_f64_f32 %1, %3\n v_cvt_f32_f64 %3, %1" : "+v"(*(double*)&i0), "+v"(*(double*)&i2), "+v"(a0), "+v"(a1));) }
        else if (MODE == 17) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 18) { REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 19) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        else if (MODE == 20) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        else if (MODE == 22) { REP8(asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_mul_f32 %1, 0x3fb8aa3b, %1\n v_mul_f32 %2, 0x3fb8aa3b, %2\n v_mul_f32 %3, 0x3fb8aa3b, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 23) { REP8(asm volatile("v_fmaak_f32 %0, %0, %4, 0x3d2aab44\n v_fmaak_f32 %1, %1, %4, 0x3d2aab44\n v_fmaak_f32 %2, %2, %4, 0x3d2aab44\n v_fmaak_f32 %3, %3, %4, 0x3d2aab44" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        else if (MODE == 24) { REP8(asm volatile("v_fmac_f32 %0, 0x3102e308, %4\n v_fmac_f32 %1, 0x3102e308, %4\n v_fmac_f32 %2, 0x3102e308, %4\n v_fmac_f32 %3, 0x3102e308, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        else if (MODE == 25) { REP8(asm volatile("v_add_f32 %0, 0x3102e308, %0\n v_add_f32 %1, 0x3102e308, %1\n v_add_f32 %2, 0x3102e308, %2\n v_add_f32 %3, 0x3102e308, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 26) { REP8(asm volatile("v_lshlrev_b32 %0, 23, %0\n v_lshlrev_b32 %1, 23, %1\n v_lshlrev_b32 %2, 23, %2\n v_lshlrev_b32 %3, 23, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 27) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 28) { REP8(asm volatile("v_or_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 29) { REP8(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        else if (MODE == 30) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc"); REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        else if (MODE == 31) { REP8(asm volatile("v_and_b32 %0, 0x7fffff, %0\n v_and_b32 %1, 0x7fffff, %1\n v_and_b32 %2, 0x7fffff, %2\n v_and_b32 %3, 0x7fffff, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        else if (MODE == 32) { REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 33) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 34) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 35) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 36) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_i32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_i32 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        else if (MODE == 37) { REP8(asm volatile("v_cmpx_lt_f32 exec, %0, %1\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 exec, %2, %3\n s_mov_b64 exec, -1" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        else if (MODE == 21) { REP8(asm volatile("v_mul_legacy_f32 %0, %0, %4\n v_mul_legacy_f32 %1, %1, %4\n v_mul_legacy_f32 %2, %2, %4\n v_mul_legacy_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(i0 + i1 + i2 + i3);
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
    printf("  %-26s differs on %10zu of %zu, largest difference %ld ulp\n", name, bad, ref.size(), maxulp);
}

int main() {
    const int n = 1 << 26;
    std::vector<float> x(n), ref(n), got(n), refd(n);
    // every 16th float of [-104, -0] (2^26 of the 1.12e9) + a few specials
    for (int i = 0; i < n; i++) { uint32_t u = 0x80000000u + (uint32_t)((uint64_t)i * 0x42d00000ull / n); memcpy(&x[i], &u, 4); }
    x[1] = -104.0f; x[2] = -103.97f; x[3] = -87.33655f; x[4] = -1e4f; x[5] = -3e38f; x[6] = -INFINITY; x[7] = NAN; x[8] = -88.0f;
    for (int i = 0; i < n; i++) { ref[i] = exp_p32_host(x[i]); refd[i] = (float)exp((double)fmaxf(x[i], -104.0f)); }
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    printf("exp on every 16th float of [-104, -0] and specials, %d arguments\n", n);
#define ACC(V, NAME, REF) hipLaunchKernelGGL(acc_exp<V>, dim3(2048), dim3(256), 0, 0, dx, dout, n); \
    hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost); report(NAME, got, REF);
    { std::vector<float> refb(n); for (int i = 0; i < n; i++) refb[i] = exp_p32_b_host(x[i]);
      ACC(3, "form B vs host form B", refb) ACC(4, "exact_math.h exp_p32 vs host B", refb) report("host form B vs host p32", refb, ref); }
    ACC(0, "p32 (VGPR k) vs host p32", ref) ACC(1, "p32 (free k) vs host p32", ref) ACC(0, "p32 vs (float)exp(double)", refd) ACC(2, "glibc form vs (float)exp", refd)
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int blocks = 256 * 8;
    const double per = 1e-3 * clk_khz * 1e3 / (8.0 * ITER * 4);
    const float base = timed(time_exp<6>, blocks, dout, 1.0f);
    printf("issue cost at 8 waves per SIMD, %d MHz (cycles per wave-evaluation per SIMD, loop baseline %.1f subtracted)\n", clk_khz / 1000, base * per);
    printf("  p32 clamp+VGPR consts %.1f | p32 clamp, compiler's consts %.1f | p32 no clamp, VGPR %.1f | p32 exponent add %.1f | glibc form (double pipe) %.1f | v_exp_f32 %.1f\n",
           (timed(time_exp<0>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<1>, blocks, dout, 1.0f) - base) * per,
           (timed(time_exp<2>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<3>, blocks, dout, 1.0f) - base) * per,
           (timed(time_exp<4>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<5>, blocks, dout, 1.0f) - base) * per);
    printf("  form B (magic shift, exponent add of k+64, x 2^-64): clamp, literals %.1f | no clamp, literals %.1f | clamp, VGPR consts %.1f\n",
           (timed(time_exp<7>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<8>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<9>, blocks, dout, 1.0f) - base) * per);
    printf("  exact_math.h exp_p32: clamp %.1f | no clamp %.1f\n", (timed(time_exp<10>, blocks, dout, 1.0f) - base) * per, (timed(time_exp<11>, blocks, dout, 1.0f) - base) * per);
    const char* names[38] = {"v_fma_f32 3 VGPR", "v_fma_f32 inline 1.0", "v_rndne_f32", "v_cvt_i32_f32", "v_ldexp_f32", "v_max_f32 VGPR", "v_bfi_b32 VGPR",
                             "v_bfi_b32 inline 0", "v_or3_b32", "v_sub_u32 VGPR", "v_ashrrev_i32 inline", "v_and_b32 VGPR", "v_lshl_add_u32 inline", "v_mul_f32 inline 0.5",
                             "v_sub_f32 inline 1.0", "v_med3_f32", "v_cvt f64<->f32", "v_mov_b32", "v_mov_b32 dpp quad_perm", "v_cmp_lt_f32 -> vcc", "v_cndmask vcc (VOP2)", "v_mul_legacy_f32", "v_mul_f32 literal", "v_fmaak_f32 (literal addend)", "v_fmac_f32 literal", "v_add_f32 literal", "v_lshlrev_b32 inline",
                             "v_add_u32 VGPR", "v_or_b32 / v_xor_b32", "v_min_f32 VGPR", "v_cndmask vcc, vcc set once", "v_and_b32 literal", "v_mul_f32 x,x", "v_fma_f32 x,x,x", "v_fma_f32 3 different VGPRs", "v_sub/v_mul 2 different",
                             "v_cmp f32/i32 -> vcc", "v_cmpx + s_mov exec (x2)"};
    float ms[38] = {timed(rate<0>, blocks, dout, 1.f), timed(rate<1>, blocks, dout, 1.f), timed(rate<2>, blocks, dout, 1.f), timed(rate<3>, blocks, dout, 1.f),
                    timed(rate<4>, blocks, dout, 1.f), timed(rate<5>, blocks, dout, 1.f), timed(rate<6>, blocks, dout, 1.f), timed(rate<7>, blocks, dout, 1.f),
                    timed(rate<8>, blocks, dout, 1.f), timed(rate<9>, blocks, dout, 1.f), timed(rate<10>, blocks, dout, 1.f), timed(rate<11>, blocks, dout, 1.f),
                    timed(rate<12>, blocks, dout, 1.f), timed(rate<13>, blocks, dout, 1.f), timed(rate<14>, blocks, dout, 1.f), timed(rate<15>, blocks, dout, 1.f),
                    timed(rate<16>, blocks, dout, 1.f), timed(rate<17>, blocks, dout, 1.f), timed(rate<18>, blocks, dout, 1.f), timed(rate<19>, blocks, dout, 1.f),
                    timed(rate<20>, blocks, dout, 1.f), timed(rate<21>, blocks, dout, 1.f), timed(rate<22>, blocks, dout, 1.f), timed(rate<23>, blocks, dout, 1.f),
                    timed(rate<24>, blocks, dout, 1.f), timed(rate<25>, blocks, dout, 1.f), timed(rate<26>, blocks, dout, 1.f), timed(rate<27>, blocks, dout, 1.f),
                    timed(rate<28>, blocks, dout, 1.f), timed(rate<29>, blocks, dout, 1.f), timed(rate<30>, blocks, dout, 1.f), timed(rate<31>, blocks, dout, 1.f),
                    timed(rate<32>, blocks, dout, 1.f), timed(rate<33>, blocks, dout, 1.f), timed(rate<34>, blocks, dout, 1.f), timed(rate<35>, blocks, dout, 1.f),
                    timed(rate<36>, blocks, dout, 1.f), timed(rate<37>, blocks, dout, 1.f)};
    for (int i = 0; i < 38; i++) printf("  %-26s %6.2f cycles per wave-instruction per SIMD\n", names[i], ms[i] * 1e-3 * clk_khz * 1e3 / (8.0 * ITER * 32));
    return 0;
}

// Microbenchmark (not part of the product): does VALU issue on MI355X depend on instruction-level parallelism INSIDE a wave
// when the SIMD holds 8 waves?  ITER x 32 v_fma_f32 per wave arranged as 1, 2, 4 or 8 independent dependency chains, plus
// mixes the blend loops are made of (v_mul/v_fma with VOP3 SGPR-mask selects, compares into SGPR pairs, DPP adds).
// cycles per wave-instruction per SIMD = time * clock / (waves per SIMD * instructions).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_ilp valu_ilp.hip ; run: ./valu_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {m, m};
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) {  // 1 chain: every v_fma depends on the previous one
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                         : "+v"(a0) : "v"(m), "v"(c));)
        } else if (MODE == 1) {  // 2 chains
            REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3"
                         : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
        } else if (MODE == 2) {  // 4 chains
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
        } else if (MODE == 3) {  // 8 chains
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 4) {  // 1 chain of v_mul_f32 (VOP2 encoding, 32-bit instruction)
            REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));)
        } else if (MODE == 5) {  // 4 chains of v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)
        } else if (MODE == 6) {  // 4 chains of v_fmac_f32 (VOP2)
            REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
        } else if (MODE == 7) {  // 4 chains: v_cmp into an SGPR pair + v_cndmask from it (VOP3)
            REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %2, %3\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[22:23]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");)
        } else if (MODE == 8) {  // 4 chains of DPP adds
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 9) {  // 4 chains of v_mul_f32 with an SGPR operand
            REP8(asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s20, %1\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s20, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20");)
        } else if (MODE == 10) {  // 4 chains of v_add_f32 between two VGPRs each (two different sources)
            REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 11) {  // permlane32 swaps
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 13) {  // v_min_f32 with a 32-bit literal
            REP8(asm volatile("v_min_f32 %0, 0x3f7d70a4, %0\n v_min_f32 %1, 0x3f7d70a4, %1\n v_min_f32 %2, 0x3f7d70a4, %2\n v_min_f32 %3, 0x3f7d70a4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 14) {  // v_cmp (VGPR, VGPR) -> vcc + v_cndmask from vcc (VOP2/VOPC encodings)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (MODE == 15) {  // v_cmp with an SGPR source -> SGPR pair, alone
            REP8(asm volatile("v_cmp_lt_f32 s[20:21], s24, %0\n v_cmp_lt_f32 s[22:23], s24, %1\n v_cmp_lt_f32 s[20:21], s24, %2\n v_cmp_lt_f32 s[22:23], s24, %3"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24");)
        } else if (MODE == 16) {  // v_cmp (VGPR, VGPR) -> SGPR pair, alone
            REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[20:21], %2, %3\n v_cmp_lt_f32 s[22:23], %3, %0"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
        } else if (MODE == 17) {  // v_cndmask from an SGPR pair, alone
            REP8(asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %0, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21");)
        } else if (MODE == 18) {  // v_pk_mul_f32 / v_pk_add_f32 alternating
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2"
                         : "+v"(p0), "+v"(p1) : "v"(p2));)
        } else if (MODE == 19) {  // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 20) {  // v_add_u32 with an SGPR operand
            REP8(asm volatile("v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20");)
        } else if (MODE == 12) {  // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y;
}
template <int MODE>
float run(float* out, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(a);
        k<MODE><<<blocks, 256>>>(out, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char* names[21] = {"v_fma_f32, 1 chain", "v_fma_f32, 2 chains", "v_fma_f32, 4 chains", "v_fma_f32, 8 chains", "v_mul_f32 (VOP2), 1 chain",
                             "v_mul_f32 (VOP2), 4 chains", "v_fmac_f32 (VOP2), 4 chains", "v_cmp->SGPR + v_cndmask", "v_add_f32 dpp, 4 chains",
                             "v_mul_f32 SGPR operand", "v_add_f32 two VGPR sources", "v_permlane32_swap", "v_exp_f32", "v_min_f32 literal",
                             "v_cmp->vcc + v_cndmask vcc", "v_cmp SGPR src -> SGPR pair", "v_cmp VGPRs -> SGPR pair", "v_cndmask from SGPR pair",
                             "v_pk_mul/add_f32", "v_rcp_f32", "v_add_u32 SGPR operand"};
    for (int wg = 2; wg <= 8; wg *= 4) {  // workgroups per CU = waves per SIMD
        const int blocks = 256 * wg;
        float ms[21] = {run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks), run<3>(out, blocks), run<4>(out, blocks), run<5>(out, blocks),
                        run<6>(out, blocks), run<7>(out, blocks), run<8>(out, blocks), run<9>(out, blocks), run<10>(out, blocks), run<11>(out, blocks), run<12>(out, blocks),
                        run<13>(out, blocks), run<14>(out, blocks), run<15>(out, blocks), run<16>(out, blocks), run<17>(out, blocks), run<18>(out, blocks), run<19>(out, blocks), run<20>(out, blocks)};
        printf("clock %d MHz, %d waves per SIMD\n", clk_khz / 1000, wg);
        for (int i = 0; i < 21; i++)
            printf("  %-30s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[i], ms[i], ms[i] * 1e-3 * clk_khz * 1e3 / ((double)wg * ITER * 32));
    }
    return 0;
}

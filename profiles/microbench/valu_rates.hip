// Microbenchmark (not part of the product): VALU issue cost on MI355X of the instructions the blend loops are made of,
// at the blend kernels' occupancy (8 waves per SIMD).  Each kernel runs ITER x 32 independent-chain instructions of one
// kind per wave; cycles per wave-instruction per SIMD = time * clock / (waves per SIMD * instructions).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f, c = 0.5f;
    const f2 mm = {m, m}, cc = {c, c};
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) {  // v_fma_f32 x 32
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
        } else if (MODE == 1) {  // v_pk_fma_f32 x 32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));)
        } else if (MODE == 2) {  // v_pk_mul_f32 with a broadcast operand x 32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));)
        } else if (MODE == 3) {  // v_exp_f32 x 32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 4) {  // v_rcp_f32 x 32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 5) {  // v_permlane32_swap x 32 (two pairs, hazards covered by the alternation)
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 6) {  // v_add_f32 with DPP row_mirror x 32
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 7) {  // v_cmp_lt_f32 into an SGPR pair + v_cndmask x 16 each
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (MODE == 8) {  // v_permlane16_swap x 32
            REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
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
    const int cus = 256, wgs_per_cu = 8;  // 8 workgroups x 4 waves = 8 waves per SIMD
    const int blocks = cus * wgs_per_cu;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char* names[9] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32 op_sel_hi", "v_exp_f32", "v_rcp_f32", "v_permlane32_swap_b32",
                            "v_add_f32 dpp row_mirror", "v_cmp + v_cndmask (pair)", "v_permlane16_swap_b32"};
    float ms[9] = {run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks), run<3>(out, blocks), run<4>(out, blocks),
                   run<5>(out, blocks), run<6>(out, blocks), run<7>(out, blocks), run<8>(out, blocks)};
    printf("clock %d MHz, %d workgroups of 256 (8 waves per SIMD), %d instructions per wave\n", clk_khz / 1000, blocks, ITER * 32);
    for (int i = 0; i < 9; i++) {
        const double cyc = ms[i] * 1e-3 * clk_khz * 1e3 / (8.0 * ITER * 32);
        printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[i], ms[i], cyc);
    }
    return 0;
}

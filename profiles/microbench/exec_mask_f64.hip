// Microbenchmark (not part of the product): does a wave64 VALU instruction get cheaper when only a quarter / a half of the
// lanes are enabled in EXEC?  (If the SIMD skipped all-disabled 16-lane passes of the double pipe, running exp_ref only on
// the lanes that pass the blend loops' cheap pre-test would pay.)  v_fma_f64 and v_fma_f32 chains under EXEC masks
// 0xffff (lanes 0-15), 0xffffffff (0-31), 0x0000ffff0000ffff (two quarters), full.
// build: hipcc --offload-arch=gfx950 -O3 -o exec_mask_f64 exec_mask_f64.hip ; run: ./exec_mask_f64
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2048;
template <int F64>
__global__ void __launch_bounds__(256, 8) k(float* out, unsigned long long mask) {
    double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b0 = 1.0f + threadIdx.x * 1e-3f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    const double m = 1.0000001, c = 1e-9;
    const float mf = 1.0000001f, cf = 1e-9f;
    const unsigned long long lanebit = 1ull << (threadIdx.x & 63);
    if (mask & lanebit) {  // divergent region: EXEC = mask for everything inside
        for (int i = 0; i < ITER; i++) {
            if (F64) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
            } else {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(mf), "v"(cf));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3) + b0 + b1 + b2 + b3;
}
template <int F64>
static float run(float* out, unsigned long long mask) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        (void)hipEventRecord(e0);
        k<F64><<<256 * 8, 256>>>(out, mask);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const unsigned long long masks[5] = {~0ull, 0xffffffffull, 0xffffull, 0x0000ffff0000ffffull, 0x1ull};
    const char* names[5] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-15 and 32-47", "lane 0"};
    printf("cycles per wave-instruction per SIMD at 8 waves per SIMD, %d MHz\n", clk_khz / 1000);
    for (int i = 0; i < 5; i++) {
        const double per = 1e-3 * clk_khz * 1e3 / (8.0 * ITER * 16);
        printf("  %-22s v_fma_f64 %.2f   v_fma_f32 %.2f\n", names[i], run<1>(out, masks[i]) * per, run<0>(out, masks[i]) * per);
    }
    return 0;
}

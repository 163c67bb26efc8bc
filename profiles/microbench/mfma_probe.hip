// Microbenchmark (not part of the product): operand / result lane layout and issue rate of v_mfma_f32_4x4x1_16b_f32
// and v_mfma_f32_16x16x4_f32 on gfx950 -- the candidates for the per-Gaussian gradient reduction of render_bwd.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout4x4(float* out) {  // two passes: D = A[la] * 1, then D = 1 * B[lb]
    const int lane = threadIdx.x;
    f4 d = {0.f, 0.f, 0.f, 0.f}, e = d;
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.0f, d, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(lane + 1), e, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = d[r] * 1000.0f + e[r];
}
__global__ void layout16(float* out) {
    const int lane = threadIdx.x;
    f4 d = {0.f, 0.f, 0.f, 0.f};
    // one-hot in k: only lanes with k = lane >> 4 == 0 contribute
    const float a = (lane < 16) ? (float)(lane + 1) : 0.f, b = (lane < 16) ? 1000.0f * (lane + 1) : 0.f;
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = d[r];
}

template <int MODE>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 4x4x1, one accumulator (dependent chain)
#pragma unroll
            for (int u = 0; u < 16; u++) d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        } else if (MODE == 1) {  // 4x4x1, two accumulators
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, d1, 0, 0, 0);
            }
        } else if (MODE == 2) {  // 16x16x4, two accumulators
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d1, 0, 0, 0);
            }
        } else if (MODE == 3) {  // 4x4x1 two accumulators interleaved with 6 independent v_fma per MFMA (co-issue check)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, d1, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            }
        } else if (MODE == 4) {  // the same 96 v_fma alone
#pragma unroll
            for (int u = 0; u < 16; u++)
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + v0 + v1 + v2 + v3;
}

template <int MODE>
float run(float* out, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(a);
        rate<MODE><<<blocks, 256>>>(out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; hipMalloc(&out, 1 << 24);
    float h[256];
    layout4x4<<<1, 64>>>(out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("4x4x1_16b: D[lane][reg] = A[la] * B[lb] -> (la, lb)\n");
    for (int lane = 0; lane < 64; lane++) {
        printf("lane %2d:", lane);
        for (int r = 0; r < 4; r++) {
            const int v = (int)h[lane * 4 + r];
            printf("  r%d=(A%2d,B%2d)", r, v / 1000 - 1, v % 1000 - 1);
        }
        printf("\n");
    }
    layout16<<<1, 64>>>(out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("16x16x4 (k = 0 only): D[lane][reg] -> (A lane i, B lane j)\n");
    for (int lane = 0; lane < 64; lane += 5) {
        printf("lane %2d:", lane);
        for (int r = 0; r < 4; r++) {
            const double v = h[lane * 4 + r];
            int fa = -1, fb = -1;
            for (int la = 0; la < 16 && fa < 0; la++)
                for (int lb = 0; lb < 16; lb++)
                    if (std::fabs(v - 1000.0 * (la + 1) * (lb + 1)) < 0.5) { fa = la; fb = lb; break; }
            printf("  r%d=(i%2d,j%2d)", r, fa, fb);
        }
        printf("\n");
    }
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int iters = 2048;
    const char* names[5] = {"4x4x1_16b x16, 1 accumulator", "4x4x1_16b x16, 2 accumulators", "16x16x4 x16, 2 accumulators",
                            "4x4x1_16b x16 + 96 v_fma interleaved", "96 v_fma alone"};
    for (int wg = 1; wg <= 4; wg *= 2) {
        const int blocks = 256 * wg;  // wg waves per SIMD
        float ms[5] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                       run<4>(out, blocks, iters)};
        for (int i = 0; i < 5; i++)
            printf("%d waves/SIMD  %-40s %8.3f ms  %7.1f cycles per loop iteration per SIMD (at %d MHz)\n", wg, names[i], ms[i],
                   ms[i] * 1e-3 * clk_khz * 1e3 / (double)(iters * wg), clk_khz / 1000);
    }
    return 0;
}

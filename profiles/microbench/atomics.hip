// Microbenchmark (not part of the product): throughput of tile-counter atomics on MI355X.
// R atomics spread over T counters (the instance histogram / slot allocation pattern of binning.hip):
//   A device scope, no return      B device scope, returning
//   C workgroup scope on per-XCD counter copies (executes in that XCD's L2), no return     D same, returning
// build: hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip ; run: ./atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }  // HW_REG_XCC_ID[3:0]
template <int MODE, int STRIDE>
__global__ void k(const unsigned* __restrict__ tiles, unsigned* cnt, unsigned* sink, int n, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned acc = 0;
    const unsigned x = xcc_id();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned t = tiles[i * 4 + r] * STRIDE;
        if (MODE == 0) __hip_atomic_fetch_add(&cnt[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) acc += __hip_atomic_fetch_add(&cnt[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_add(&cnt[x * T + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) acc += __hip_atomic_fetch_add(&cnt[x * T + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE & 1) sink[i] = acc;
}
int main() {
    const int n = 425824, T = 8160;  // ~1.7M atomics
    std::vector<unsigned> h(n * 4);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % T; }
    unsigned *tiles, *cnt, *sink;
    hipMalloc(&tiles, h.size() * 4); hipMalloc(&cnt, 32 * T * 4); hipMalloc(&sink, n * 4);
    hipMemcpy(tiles, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[7] = {"dense no-return", "dense returning", "64B-padded no-return", "64B-padded returning", "128B-padded no-return", "128B-padded returning", "16B-padded returning"};
    for (int mode = 0; mode < 7; mode++) {
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            hipMemset(cnt, 0, 32 * T * 4);
            hipEventRecord(a);
            if (mode == 0) k<0, 1><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 1) k<1, 1><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 2) k<0, 16><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 3) k<1, 16><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 4) k<0, 32><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 5) k<1, 32><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            if (mode == 6) k<1, 4><<<(n + 255) / 256, 256>>>(tiles, cnt, sink, n, T);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        std::vector<unsigned> c(32 * T); hipMemcpy(c.data(), cnt, 32 * T * 4, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (auto v : c) tot += v;
        printf("%-22s %8.1f us  %6.1f G atomics/s  (sum %llu, expect %d)\n", names[mode], best * 1e3, n * 4 / (best * 1e-3) / 1e9, tot, n * 4);
    }
    return 0;
}

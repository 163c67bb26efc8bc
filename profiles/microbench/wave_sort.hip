// Microbenchmark (not part of the product): one wave sorting one tile list in its registers (csrc/tile_sort.h), round 8's
// routine (tile_sort_r8.h, a copy of the header at the start of round 9) against the current one, at the list lengths and
// wave counts bin_tiles meets: lists of ~200 entries two per wave (the uniform scene), ~1000 entries one per wave with the
// other waves idle (a clustered frame at four tiles per segment).
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o wave_sort wave_sort.hip ; run: ./wave_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
#include <type_traits>
#define dgr dgr_r8
#include "tile_sort_r8.h"
#undef dgr
#include "../../diff-gaussian-rasterization_amd/csrc/tile_sort.h"

template <int WHICH>
__device__ __forceinline__ void sort_one(uint64_t* list, int n, uint32_t* dst, int lane) {
    if constexpr (WHICH == 0) {
        using namespace dgr_r8;
        if (n <= 64) sort_wave_trunc<1>(list, n, dst, lane);
        else if (n <= 128) sort_wave_trunc<2>(list, n, dst, lane);
        else if (n <= 256) sort_wave_trunc<4>(list, n, dst, lane);
        else if (n <= 512) sort_wave_trunc<8>(list, n, dst, lane);
        else sort_wave_trunc_1024(list, n, dst, lane);
    } else if constexpr (WHICH == 2) {  // the network alone: words in, sorted words out (no key fetch, no fix-up)
        using namespace dgr;
        auto run = [&](auto tag) {
            constexpr int NCH = decltype(tag)::value;
            uint32_t v[NCH];
            for (int c = 0; c < NCH; c++) { const int e = c * 64 + lane; v[c] = e < n ? ((uint32_t)(list[e] >> 32) & ~1023u) | (uint32_t)e : 0xffffffffu; }
            const NetLane k = net_lane(lane);
            sort_net<NCH>(v, k);
            for (int c = 0; c < NCH; c++) dst[c * 64 + lane] = v[c];
        };
        if (n <= 64) run(std::integral_constant<int, 1>{});
        else if (n <= 128) run(std::integral_constant<int, 2>{});
        else if (n <= 256) run(std::integral_constant<int, 4>{});
        else if (n <= 512) run(std::integral_constant<int, 8>{});
        else run(std::integral_constant<int, 16>{});
    } else {
        using namespace dgr;
        if (n <= 64) sort_wave_trunc<1>(list, n, dst, lane);
        else if (n <= 128) sort_wave_trunc<2>(list, n, dst, lane);
        else if (n <= 256) sort_wave_trunc<4>(list, n, dst, lane);
        else if (n <= 512) sort_wave_trunc<8>(list, n, dst, lane);
        else sort_wave_trunc<16>(list, n, dst, lane);
    }
}

// every wave: `reps` times {copy its list from global into LDS, sort it, ids to global}; active waves < waves of the block idle
template <int WHICH>
__global__ void __launch_bounds__(512, 6) k(const uint64_t* __restrict__ src, uint32_t* __restrict__ dst, int n, int active, int reps,
                                            unsigned long long* __restrict__ ticks) {
    __shared__ uint64_t lds[6144];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= active) return;
    uint64_t* list = lds + wave * (6144 / 8 > n ? 6144 / 8 : n);
    if ((wave + 1) * n > 6144 && active > 1) return;
    const uint64_t* mine = src + ((size_t)blockIdx.x * 8 + wave) * 1024;
    uint32_t* out = dst + ((size_t)blockIdx.x * 8 + wave) * 1024;
    unsigned long long total = 0;
    for (int r = 0; r < reps; r++) {
        for (int i = lane; i < n; i += 64) list[i] = mine[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long t0 = wall_clock64();
        sort_one<WHICH>(list, n, out, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        total += wall_clock64() - t0;
    }
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = total;
}

int main(int argc, char** argv) {
    const int tie_every = argc > 1 ? atoi(argv[1]) : 37;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256, reps = 20;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> h((size_t)blocks * 8 * 1024);
    // keys as bin_tiles meets them: depth bits of z in [1, 6] << 32 | unique id; some depths tied in their upper 22 bits
    std::uniform_real_distribution<float> z(1.0f, 6.0f);
    for (size_t i = 0; i < h.size(); i++) {
        float d = z(rng);
        if (tie_every > 0 && (i % tie_every) == 0 && i > 0) { uint32_t b; memcpy(&b, &d, 4); uint32_t pb = (uint32_t)(h[i - 1] >> 32); b = (pb & ~7u) | (b & 7u); memcpy(&d, &b, 4); }
        uint32_t b; memcpy(&b, &d, 4);
        h[i] = ((uint64_t)b << 32) | (uint32_t)(rng() & 0x0fffffff);
    }
    uint64_t* d_src; uint32_t* d_dst; unsigned long long* d_t;
    hipMalloc(&d_src, h.size() * 8); hipMalloc(&d_dst, h.size() * 4); hipMalloc(&d_t, blocks * 8 * 8);
    hipMemcpy(d_src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> out(h.size());
    std::vector<unsigned long long> t(blocks * 8);
    const int cases[][2] = {{200, 8}, {200, 4}, {330, 8}, {500, 8}, {500, 4}, {700, 4}, {1000, 4}, {1000, 1}, {1024, 4}};
    for (auto& cs : cases) {
        const int n = cs[0], active = cs[1];
        for (int which = 0; which < 3; which++) {
            hipMemset(d_t, 0, blocks * 8 * 8);
            hipMemset(d_dst, 0xff, h.size() * 4);
            if (which == 0) k<0><<<blocks, 512>>>(d_src, d_dst, n, active, reps, d_t);
            else if (which == 1) k<1><<<blocks, 512>>>(d_src, d_dst, n, active, reps, d_t);
            else k<2><<<blocks, 512>>>(d_src, d_dst, n, active, reps, d_t);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(out.data(), d_dst, out.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
            // check wave 0 of a few blocks
            int bad = 0;
            for (int b = 0; b < blocks; b += 17)
                for (int w = 0; w < active && (w + 1) * std::max(n, 768) <= 6144; w++) {
                    std::vector<uint64_t> ref(h.begin() + ((size_t)b * 8 + w) * 1024, h.begin() + ((size_t)b * 8 + w) * 1024 + n);
                    std::sort(ref.begin(), ref.end());
                    for (int i = 0; i < n && which < 2; i++) bad += out[((size_t)b * 8 + w) * 1024 + i] != (uint32_t)ref[i];
                }
            double sum = 0; int cnt = 0; unsigned long long mx = 0;
            for (auto v : t) if (v) { sum += (double)v; cnt++; mx = std::max(mx, v); }
            printf("n %4d waves %d %s: %.2f us per sort (mean), %.2f (slowest wave)  mismatches %d\n", n, active, which == 2 ? "net only" : which ? "round 9" : "round 8",
                   cnt ? sum / cnt / reps / 100.0 : 0.0, mx / (double)reps / 100.0, bad);
        }
    }
    return 0;
}

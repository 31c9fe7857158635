// Round 4, VERDICT round 3 item 2 option (a): what does it cost to put 64 sorted-by-group records into (group, voxel,
// pixel) order INSIDE a wave - the step a 3-pass group-key sort leaves to the sums kernel - against the radix pass it
// would replace?  Stand-alone probe (not part of libo3dr):
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/r04_wave_order_probe.hip -o /tmp/probe && /tmp/probe
//
// Three kernels over the same 149.6 M synthetic (group-sorted) 28-bit indices, one record per lane and step, 16 steps per
// wave, eight waves per SIMD resident:
//   heads   : read the index, head flag against the lane below (DPP), count the flags          (what k_run_heads does)
//   bitonic : + a 64-lane bitonic network on (group rank in step | low 7 bits | lane) before the flags
//   match   : + match-any masks over the 13 varying bits by ballots (the other way to find a voxel's records)
// Prints ms per pass over all records and VALU-free bytes; the scatter pass this would save takes 0.485 ms + 0.12 ms of
// histogram per 149.6 M records (profiles/r03_kernel_stats_200frames.csv).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int kSteps = 16;

__device__ __forceinline__ uint32_t shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); }

// compare-exchange with the lane `lane ^ mask`-style partner value `other`; `up`: this lane keeps the smaller one
__device__ __forceinline__ uint32_t cx(uint32_t v, uint32_t other, bool keep_min)
{
    const uint32_t lo = v < other ? v : other, hi = v < other ? other : v;
    return keep_min ? lo : hi;
}

__device__ __forceinline__ uint32_t bitonic64(uint32_t v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)v, j, 64);
            const bool up = (lane & k) == 0;          // ascending block
            const bool lower = (lane & j) == 0;       // this lane is the lower partner
            v = cx(v, other, up == lower);
        }
    }
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t total = 0;
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        const int64_t i = (wave * kSteps + s) * 64 + lane;
        const uint32_t key = i < n ? keys[i] : 0xffffffffu;
        uint32_t v = key;
        if (MODE == 1) {
            // group rank inside the step (heads of key >> 7), then the composite (rank | low 7 bits | lane)
            const uint32_t g = key >> 7;
            uint32_t pg = shr1(g);
            const unsigned long long gh = __ballot(lane == 0 || g != pg);
            const uint32_t rank = (uint32_t)__popcll(gh & ((2ull << lane) - 1ull)) - 1u;
            const uint32_t comp = (rank << 13) | ((key & 127u) << 6) | (uint32_t)lane;
            const uint32_t sorted = bitonic64(comp, lane);
            v = sorted >> 6;  // (rank | voxel): heads where it changes
        } else if (MODE == 2) {
            const uint32_t g = key >> 7;
            uint32_t pg = shr1(g);
            const unsigned long long gh = __ballot(lane == 0 || g != pg);
            const uint32_t rank = (uint32_t)__popcll(gh & ((2ull << lane) - 1ull)) - 1u;
            const uint32_t comp = (rank << 7) | (key & 127u);
            unsigned long long peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 13; ++b) {
                const bool bit = (comp >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            // a voxel's first record = the lowest lane of its peers
            total += (peers & ((1ull << lane) - 1ull)) == 0ull ? 1u : 0u;
            continue;
        }
        const uint32_t prev = shr1(v);
        total += (lane == 0 || v != prev) ? 1u : 0u;
    }
    // per-wave count (one store per wave)
    for (int o = 32; o > 0; o >>= 1) total += (uint32_t)__shfl_down((int)total, o, 64);
    if (lane == 0) counts[wave] = total;
}

int main()
{
    const int64_t n = 149600000;
    std::vector<uint32_t> h((size_t)n);
    // group-sorted indices with ~14.5 records per group of 128 cells and ~1.5 records per voxel, pixel order inside a group
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    uint32_t grp = 0;
    for (int64_t i = 0; i < n;) {
        const int len = 1 + (int)(next() % 28);
        for (int k = 0; k < len && i < n; ++k, ++i) h[(size_t)i] = ((grp & 0x1fffffu) << 7) | (uint32_t)(next() % 20 * 6 % 128);
        grp += 1 + (uint32_t)(next() % 3);
    }
    uint32_t *d_keys, *d_counts;
    const int64_t waves = (n + 64 * kSteps - 1) / (64 * kSteps), blocks = (waves + 3) / 4;
    CHECK(hipMalloc(&d_keys, (size_t)n * 4));
    CHECK(hipMalloc(&d_counts, (size_t)blocks * 4 * 4));
    CHECK(hipMemcpy(d_keys, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[3] = {"heads", "bitonic", "match"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3((unsigned)blocks), dim3(256), 0, 0, d_keys, n, d_counts);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3((unsigned)blocks), dim3(256), 0, 0, d_keys, n, d_counts);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3((unsigned)blocks), dim3(256), 0, 0, d_keys, n, d_counts);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        std::vector<uint32_t> c((size_t)blocks * 4);
        CHECK(hipMemcpy(c.data(), d_counts, c.size() * 4, hipMemcpyDeviceToHost));
        uint64_t sum = 0;
        for (int64_t w = 0; w < waves; ++w) sum += c[(size_t)w];
        printf("{\"probe\": \"%s\", \"records\": %lld, \"ms\": %.4f, \"GBps_of_index_reads\": %.1f, \"heads_counted\": %llu}\n", names[mode],
               (long long)n, best, (double)n * 4 / (best * 1e-3) / 1e9, (unsigned long long)sum);
    }
    return 0;
}

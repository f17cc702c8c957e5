// HBM read ceiling probe: what a kernel that does nothing but read reaches on this GPU, for the load shapes the attention
// streamers use (16 bytes per lane, nt policy, a few wave loads in flight per wave, 8 or more waves per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/read_peak.hip -o tools/probe/read_peak ; run: ./read_peak [GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LG>
__global__ __launch_bounds__(512) void read_kernel(const u32x4* __restrict__ src, size_t n_vec, unsigned* sink) {
    const size_t lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    u32x4 acc = {0u, 0u, 0u, 0u};
    // a wave reads LG consecutive KiB per step; steps are dealt round-robin over all waves
    for (size_t base = wave * LG * 64; base + LG * 64 <= n_vec; base += n_waves * LG * 64) {
        u32x4 v[LG];
#pragma unroll
        for (int i = 0; i < LG; ++i) v[i] = __builtin_nontemporal_load(src + base + i * 64 + lane);
#pragma unroll
        for (int i = 0; i < LG; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;   // keeps the loads alive
}

template <int LG>
static void run(const u32x4* src, size_t bytes, unsigned* sink, int wgs_per_cu, int waves, int n_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(n_cu * wgs_per_cu), block(waves * 64);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read_kernel<LG>, grid, block, 0, 0, src, bytes / 16, sink);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(read_kernel<LG>, grid, block, 0, 0, src, bytes / 16, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d wave loads in flight, %d waves/CU: %.3f ms  %.2f TB/s\n", LG, wgs_per_cu * waves, ms / reps,
           bytes / (ms / reps) / 1e9);
}

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gib << 30;
    void* src = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMemset(src, 1, bytes);
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%zu GiB, %d CUs\n", gib, n_cu);
    for (int waves_per_cu : {8, 16, 32}) {
        const int wgs = waves_per_cu / 8;
        run<1>((const u32x4*)src, bytes, sink, wgs, 8, n_cu);
        run<2>((const u32x4*)src, bytes, sink, wgs, 8, n_cu);
        run<4>((const u32x4*)src, bytes, sink, wgs, 8, n_cu);
        run<8>((const u32x4*)src, bytes, sink, wgs, 8, n_cu);
    }
    hipFree(src);
    hipFree(sink);
    return 0;
}

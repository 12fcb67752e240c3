// Dependent-load latency of one wave over buffers of growing size (is the 1.7 us per round trip the query kernels see
// the memory system's latency at that footprint -- TLB reach, MALL misses -- or queueing under load?).
// build: hipcc --offload-arch=gfx950 -O2 profiles/probes/lat_probe.hip -o profiles/lat_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
__global__ void chase(const uint32_t* next, uint32_t start, uint32_t steps, uint32_t stride_dw, uint32_t* out, unsigned long long* ticks) {
    uint32_t i = start + threadIdx.x * 97u; // one chain per lane: vector loads, 64 different lines per step
    const unsigned long long t0 = wall_clock64();
    for (uint32_t s = 0; s < steps; ++s) i = next[(size_t)i * stride_dw];
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *out = i; *ticks = t1 - t0; }
}
int main() {
    const uint32_t stride_dw = 16; // one 64-byte line per element
    for (size_t mb : {16, 64, 256, 1024, 3072, 8192}) {
        const size_t n = mb * 1024 * 1024 / 64;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937_64 rng(1234);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> host(n * stride_dw, 0);
        for (size_t k = 0; k < n; ++k) host[(size_t)perm[k] * stride_dw] = perm[(k + 1) % n]; // one cycle through all lines
        uint32_t *d, *out; unsigned long long* ticks;
        if (hipMalloc(&d, host.size() * 4) != hipSuccess) { printf("%zu MB: alloc failed\n", mb); continue; }
        hipMalloc(&out, 4); hipMalloc(&ticks, 8);
        hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice);
        const uint32_t steps = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0u, steps, stride_dw, out, ticks);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 1u, steps, stride_dw, out, ticks);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("%5zu MB footprint: %.0f ns per dependent gather step by wall_clock64 at 100 MHz, %.0f ns by hipEvents (idle GPU, one wave, 64 chains)\n", mb, t * 10.0 / steps, ms * 1e6 / steps);
        hipFree(d); hipFree(out); hipFree(ticks);
    }
    return 0;
}

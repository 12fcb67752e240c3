// FETCH_SIZE calibration by access shape (run under rocprofv3 --pmc, one counter group per run):
//   k_stream4   coalesced 4 B per lane over `bytes` of a table          -> known bytes
//   k_stream16  coalesced 16 B per lane                                  -> known bytes
//   k_gather1   one random byte per lane (a distinct 64-byte line each, table >> every cache) -> known number of gathers
//   k_gather1s  one byte per lane, consecutive lanes 5 bytes apart (the sorted-candidate shape of the range-table gathers)
// build: hipcc --offload-arch=gfx950 -O2 calib_gather.hip -o calib_gather ; usage: ./calib_gather [table GiB] [gathers 2^n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void k_stream4(const uint32_t* t, size_t ndw, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndw; i += (size_t)gridDim.x * blockDim.x) acc ^= t[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream16(const uint4* t, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ void k_gather1(const uint8_t* t, size_t bytes, size_t per_thread, uint32_t* out) {
    uint32_t acc = 0;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t k = 0; k < per_thread; ++k) acc += t[mix(tid * per_thread + k + 1) % bytes];
    if (acc == 0x12345678u) out[0] = acc;
}
// every wave reads 64 bytes that are 5 apart, starting at a random place: ~5 lines per wave instruction
__global__ void k_gather1s(const uint8_t* t, size_t bytes, size_t per_thread, uint32_t* out) {
    uint32_t acc = 0;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (size_t k = 0; k < per_thread; ++k) acc += t[(mix(wave * per_thread + k + 1) % (bytes - 512)) + 5 * lane];
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const int lg = argc > 2 ? atoi(argv[2]) : 28;
    const size_t bytes = gib << 30;
    uint8_t* t; uint32_t* out;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 1, bytes);
    hipDeviceSynchronize();
    const size_t stream_bytes = 2ull << 30;
    hipLaunchKernelGGL(k_stream4, dim3(256 * 32), dim3(256), 0, 0, (const uint32_t*)t, stream_bytes / 4, out);
    hipLaunchKernelGGL(k_stream16, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)(t + stream_bytes), stream_bytes / 16, out);
    const size_t n = 1ull << lg, threads = 256ull * 32 * 256, per = n / threads;
    hipLaunchKernelGGL(k_gather1, dim3(256 * 32), dim3(256), 0, 0, t, bytes, per, out);
    hipLaunchKernelGGL(k_gather1s, dim3(256 * 32), dim3(256), 0, 0, t, bytes, per, out);
    hipDeviceSynchronize();
    printf("calib: table %zu GiB; k_stream4 %zu bytes; k_stream16 %zu bytes; k_gather1 %zu gathers (x64 B = %zu bytes of distinct lines); "
           "k_gather1s %zu wave instructions x 64 lanes, 5 B apart (~5-6 lines of 64 B each)\n",
           gib, stream_bytes, stream_bytes, per * threads, per * threads * 64, per * threads / 64);
    return 0;
}

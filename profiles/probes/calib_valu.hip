// Issue rate of plain 32-bit integer VALU instructions on gfx950 (is a wave64 VALU instruction 2 or 4 cycles
// of its SIMD?). W waves per SIMD, each running ITER x 64 independent v_add_u32 / v_xor_b32 / v_lshl_add_u32 (8 accumulators,
// no memory, no LDS); cycles per wave-instruction per SIMD = elapsed shader cycles x 1 / (W x instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 -o calib_valu calib_valu.hip && ./calib_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(64) k_valu(unsigned* out, int iter, unsigned long long* cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iter; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %2, %2, %3\n\tv_lshl_add_u32 %4, %4, 1, %5\n\tv_add_u32 %6, %6, %7\n\t"
                         "v_and_b32 %1, %1, %0\n\tv_or_b32 %3, %3, %2\n\tv_sub_u32 %5, %5, %4\n\tv_max_u32 %7, %7, %6"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iter = 20000;
    printf("%s: %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = cus * 4 * w;
        unsigned* out;
        unsigned long long* cyc;
        hipMalloc(&out, blocks * 64 * 4);
        hipMalloc(&cyc, blocks * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, out, 100, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, out, iter, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
        hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < blocks; ++i) mean += (double)h[i];
        mean /= blocks;
        const double ninstr = (double)iter * 64.0; // per wave
        printf("W=%d waves/SIMD: %.3f ms, %.1f G wave-instr/s chip-wide, wave's own clock: %.2f shader cycles per instruction => %.2f cycles per wave-instruction per SIMD\n",
               w, ms, blocks * ninstr / ms / 1e6, mean / ninstr, mean / ninstr / w);
        free(h);
        hipFree(out);
        hipFree(cyc);
    }
    return 0;
}

// probe: LDS-DMA semantics on gfx950 (global_load_lds_ubyte / _dword): LDS destination stride per lane, the effect of the
// instruction offset on the LDS side, and counted vmcnt waits. Build: hipcc --offload-arch=gfx950 -O2 ldsdma_probe.hip -o ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(const uint8_t* tab, const uint32_t* words, uint32_t* out) {
    __shared__ uint32_t L[512];
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) L[i] = 0xAAAAAAAAu;
    __syncthreads();
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&L[0];
    uint32_t keep;
    // (1) one byte per lane: lane l reads tab[l * 3]
    const uint32_t off = lane * 3u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(tab), "s"(lds) : "memory");
    // (2) dword per lane with instruction offset 256 on a fresh M0 = lds + 1024: where does it land?
    const uint32_t voff = lane * 4u, lds2 = lds + 1024u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:256\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(words), "s"(lds2) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = L[i];
}

int main() {
    std::vector<uint8_t> tab(4096);
    for (int i = 0; i < 4096; ++i) tab[i] = (uint8_t)(i & 0xFF);
    std::vector<uint32_t> words(1024);
    for (int i = 0; i < 1024; ++i) words[i] = 0x10000000u + i;
    uint8_t* dt; uint32_t *dw, *dout;
    hipMalloc(&dt, 4096); hipMalloc(&dw, 4096); hipMalloc(&dout, 2048);
    hipMemcpy(dt, tab.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(dw, words.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dt, dw, dout);
    std::vector<uint32_t> out(512);
    hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost);
    printf("ubyte region (first 72 dwords):\n");
    for (int i = 0; i < 72; ++i) printf("%08x%c", out[i], (i % 8 == 7) ? '\n' : ' ');
    printf("dword region from L[256] (M0 = +1024, inst offset 256):\n");
    for (int i = 256; i < 256 + 136; ++i) printf("%08x%c", out[i], (i % 8 == 7) ? '\n' : ' ');
    return 0;
}

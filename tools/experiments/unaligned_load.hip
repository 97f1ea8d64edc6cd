// Does a global_load_dword at a byte-unaligned address return the 4 bytes at that address on gfx950 under ROCm's default
// SH_MEM_CONFIG (unaligned access mode)?  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/unaligned unaligned_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint8_t *src, uint32_t *out, uint2 *out2)
{
    const int o = threadIdx.x;  // byte offset 0..63
    uint32_t v;
    const uint8_t *p = src + o;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    out[o] = v;
    uint2 w;
    asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    out2[o] = w;
}
int main()
{
    uint8_t h[256];
    for (int i = 0; i < 256; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o; uint2 *o2;
    hipMalloc(&d, 256); hipMalloc(&o, 64 * 4); hipMalloc(&o2, 64 * 8);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT\n"); return 1; }
    uint32_t r[64]; uint2 r2[64];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, sizeof r2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t e = h[i] | (h[i + 1] << 8) | (h[i + 2] << 16) | ((uint32_t)h[i + 3] << 24);
        uint32_t e2 = h[i + 4] | (h[i + 5] << 8) | (h[i + 6] << 16) | ((uint32_t)h[i + 7] << 24);
        if (r[i] != e || r2[i].x != e || r2[i].y != e2) ++bad;
    }
    printf("unaligned dword/dwordx2 loads: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    return bad != 0;
}

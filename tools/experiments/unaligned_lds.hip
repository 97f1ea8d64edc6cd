// Do ds_read_b32 / ds_read_b64 / ds_read_u16 at byte-unaligned LDS addresses return the bytes at that address on gfx950
// under ROCm's default SH_MEM_CONFIG?  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/unaligned_lds unaligned_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out32, uint2 *out64, uint32_t *out16)
{
    __shared__ uint8_t lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x;  // LDS byte address 0..63 (+ base)
    uint32_t a; uint2 b; uint32_t c;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr) : "memory");
    out32[threadIdx.x] = a; out64[threadIdx.x] = b; out16[threadIdx.x] = c;
}
int main()
{
    uint32_t *o32, *o16; uint2 *o64;
    (void)hipMalloc(&o32, 256); (void)hipMalloc(&o16, 256); (void)hipMalloc(&o64, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o32, o64, o16);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT\n"); return 1; }
    uint32_t r32[64], r16[64]; uint2 r64[64];
    (void)hipMemcpy(r32, o32, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(r16, o16, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r64, o64, 512, hipMemcpyDeviceToHost);
    auto B = [](int i) { return (uint32_t)(uint8_t)(i * 7 + 3); };
    int bad32 = 0, bad64 = 0, bad16 = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t e0 = B(i) | (B(i + 1) << 8) | (B(i + 2) << 16) | (B(i + 3) << 24);
        const uint32_t e1 = B(i + 4) | (B(i + 5) << 8) | (B(i + 6) << 16) | (B(i + 7) << 24);
        bad32 += r32[i] != e0; bad64 += r64[i].x != e0 || r64[i].y != e1; bad16 += r16[i] != (e0 & 0xFFFF);
    }
    printf("unaligned LDS reads: b32 %s, b64 %s, u16 %s\n", bad32 ? "WRONG" : "OK", bad64 ? "WRONG" : "OK", bad16 ? "WRONG" : "OK");
    return 0;
}

// ring_rate.hip — what vector-instruction rate do k_pyr_fast's own inner bodies reach when NOTHING else is in the way?
// The real device functions (ring test, pre-test, corner score) run in dense loops on LDS-resident random pixels, 4 workgroups of 512
// threads per CU (8 waves per SIMD, as the kernel).  rocprofv3 --pmc SQ_INSTS_VALU + the kernel trace give wave-instructions and time:
// cycles per vector instruction per SIMD = time x 1024 SIMDs x clock / SQ_INSTS_VALU.  That is the practical issue peak of this
// instruction mix - the number k_pyr_fast's 4.4 cycles per instruction has to be compared with.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I wut-cuda-orb-slam3_amd/csrc tools/experiments/ring_rate.hip -o /tmp/ring_rate
#include "../../wut-cuda-orb-slam3_amd/csrc/orbx_kernels.hip"

#define RR_ROWS 48
#define RR_TP 320

__device__ __forceinline__ void rr_fill(uint8_t *lds, unsigned seed)
{
    uint32_t *w = (uint32_t *)lds;
    for (int i = threadIdx.x; i < RR_ROWS * RR_TP / 4; i += blockDim.x) {
        unsigned x = (i + seed) * 2654435761u;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        // mostly smooth with some structure: low-amplitude noise around 100, a bright block every 16 dwords
        const unsigned base = ((i >> 4) & 1) ? 0x90909090u : 0x64646464u;
        w[i] = base + (x & 0x07070707u);
    }
    __syncthreads();
}

__global__ __launch_bounds__(512, 8) void k_rate_ring(uint32_t *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    rr_fill(lds, blockIdx.x);
    const uint32_t T4 = 20u * 0x01010101u, C4 = 235u * 0x01010101u;
    uint32_t acc = 0;
    int r = (threadIdx.x >> 6) * 5 % 40, g = 2 + (threadIdx.x & 63);  // a wave's lanes = consecutive groups of one row (as the queue mostly has them)
    for (int i = 0; i < iters; ++i) {
        acc += fast_group_flags_lds(lds + r * RR_TP + 4 * g, RR_TP, T4, C4);
        r = r + 1 >= 40 ? 0 : r + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(512, 8) void k_rate_pretest(uint32_t *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    rr_fill(lds, blockIdx.x);
    const uint32_t T4 = 20u * 0x01010101u, C4 = 235u * 0x01010101u;
    uint32_t acc = 0;
    int r = (threadIdx.x >> 6) * 5 % 40, s = 1 + (threadIdx.x & 31) + 0 * (threadIdx.x & 32);  // consecutive strips; two half-waves on the same row segment
    for (int i = 0; i < iters; ++i) {
        const uint8_t *wp = lds + r * RR_TP + 8 * s;
        uint32_t w0[4], w3[4], w6[4];
        const uint2 a0 = *(const uint2 *)(wp), a1 = *(const uint2 *)(wp + 8);
        const uint2 b0 = *(const uint2 *)(wp + 3 * RR_TP), b1 = *(const uint2 *)(wp + 3 * RR_TP + 8);
        const uint2 c0 = *(const uint2 *)(wp + 6 * RR_TP), c1 = *(const uint2 *)(wp + 6 * RR_TP + 8);
        w0[0] = a0.x; w0[1] = a0.y; w0[2] = a1.x; w0[3] = a1.y;
        w3[0] = b0.x; w3[1] = b0.y; w3[2] = b1.x; w3[3] = b1.y;
        w6[0] = c0.x; w6[1] = c0.y; w6[2] = c1.x; w6[3] = c1.y;
        acc += fast_group_pretest<0>(w0, w3, w6, T4, C4) + 3 * fast_group_pretest<1>(w0, w3, w6, T4, C4);
        r = r + 1 >= 40 ? 0 : r + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(512, 8) void k_rate_score(uint32_t *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    rr_fill(lds, blockIdx.x);
    uint32_t acc = 0;
    int r = 3 + (threadIdx.x >> 6) * 5 % 36, c = 8 + (threadIdx.x & 63) * 4;
    for (int i = 0; i < iters; ++i) {
        acc += (uint32_t)fast_corner_score(lds + r * RR_TP + c, RR_TP);
        r = r + 1 >= 40 ? 3 : r + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// a stream of independent full-rate operations (v_and / v_add / v_bitop3 on 8 accumulators) for the same occupancy: the floor
__global__ __launch_bounds__(512, 8) void k_rate_simple(uint32_t *out, int iters)
{
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * (2 * k + 3);
    const uint32_t b = blockIdx.x * 2654435761u + threadIdx.x, c = b ^ 0x5bd1e995u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_and_b32 %0, %0, %1\nv_add_u32 %0, %0, %2\nv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n" : "+v"(a[k]) : "v"(b), "v"(c));
        }
    }
    uint32_t x = 0;
    for (int k = 0; k < 8; ++k) x ^= a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main(int argc, char **argv)
{
    const int blocks = 256 * 4 * 6;  // six rounds of 4 workgroups per CU
    uint32_t *d;
    if (hipMalloc(&d, (size_t)blocks * 512 * 4) != hipSuccess) return 1;
    const size_t lds = RR_ROWS * RR_TP;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; void (*k)(uint32_t *, int); int iters; size_t lds; } tests[] = {
        {"ring test (fast_group_flags_lds)", k_rate_ring, 200, lds}, {"pre-test (2 x fast_group_pretest)", k_rate_pretest, 400, lds},
        {"corner score (fast_corner_score)", k_rate_score, 300, lds}, {"and/add/bitop3 stream", k_rate_simple, 2000, 0}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(512), t.lds, 0, d, t.iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(512), t.lds, 0, d, t.iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-36s %8.3f ms  (%d workgroups x %d iterations per lane)\n", t.name, ms, blocks, t.iters);
    }
    return 0;
}

// valu_bench.hip — per-instruction VALU throughput microbenchmark for gfx950 (development tool).
// Each kernel issues ITER x 16 independent instances of one instruction per lane (8 accumulators x 2),
// full occupancy; reports wave-instructions per cycle per SIMD-equivalent as Gops (lane-ops/s).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048

#define BENCH_KERNEL(NAME, ASM)                                                             \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)               \
    {                                                                                       \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;            \
        unsigned a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                    \
        unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5bd1e995u;                 \
        for (int i = 0; i < ITER; ++i) {                                                    \
            asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)    \
                         ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                 \
        }                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }

#define A_ADD(x) "v_add_u32 " #x ", " #x ", %8\n"
#define A_SUB_SDWA(x) "v_sub_u32_sdwa " #x ", %8, " #x " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_ALIGNBIT(x) "v_alignbit_b32 " #x ", " #x ", %8, 31\n"
#define A_AND(x) "v_and_b32 " #x ", " #x ", %8\n"
#define A_MIN3(x) "v_min3_i32 " #x ", " #x ", %8, %9\n"
#define A_MIN(x) "v_min_i32 " #x ", " #x ", %8\n"
#define A_PKSUB(x) "v_pk_sub_i16 " #x ", " #x ", %8\n"
#define A_PKMIN(x) "v_pk_min_i16 " #x ", " #x ", %8\n"
#define A_PKMAXU(x) "v_pk_max_u16 " #x ", " #x ", %8\n"
#define A_PKLSHR(x) "v_pk_lshrrev_b16 " #x ", 15, " #x "\n"
#define A_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9\n"
#define A_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define A_MINF(x) "v_min_f32 " #x ", " #x ", %8\n"
#define A_PKMINH(x) "v_pk_min_f16 " #x ", " #x ", %8\n"
#define A_PKADDH(x) "v_pk_add_f16 " #x ", " #x ", %8\n"
#define A_BFI(x) "v_bfi_b32 " #x ", %8, %9, " #x "\n"
#define A_LSHLOR(x) "v_lshl_or_b32 " #x ", " #x ", 1, %8\n"
#define A_MAD24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9\n"
#define A_SAD(x) "v_sad_u8 " #x ", " #x ", %8, %9\n"
#define A_DOT4(x) "v_dot4_u32_u8 " #x ", " #x ", %8, %9\n"
#define A_BITOP3(x) "v_bitop3_b32 " #x ", " #x ", %8, %9 bitop3:0x80\n"
#define A_LSHR(x) "v_lshrrev_b32 " #x ", 1, " #x "\n"
#define A_CMPADDC(x) "v_cmp_lt_u32 vcc, " #x ", %8\nv_addc_co_u32 " #x ", vcc, " #x ", " #x ", vcc\n"
#define A_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8\n"
#define A_BCNT(x) "v_bcnt_u32_b32 " #x ", %8, " #x "\n"
#define A_MSAD(x) "v_msad_u8 " #x ", " #x ", %8, %9\n"
#define A_PKMADU16(x) "v_pk_mad_u16 " #x ", " #x ", %8, %9\n"
#define A_ALIGNBYTE(x) "v_alignbyte_b32 " #x ", " #x ", %8, 1\n"
#define A_MAX3U(x) "v_max3_u32 " #x ", " #x ", %8, %9\n"
#define A_PKFMA32(x) "v_and_b32 " #x ", " #x ", %8\n"

BENCH_KERNEL(k_add, A_ADD)
BENCH_KERNEL(k_sub_sdwa, A_SUB_SDWA)
BENCH_KERNEL(k_alignbit, A_ALIGNBIT)
BENCH_KERNEL(k_and, A_AND)
BENCH_KERNEL(k_min3, A_MIN3)
BENCH_KERNEL(k_min, A_MIN)
BENCH_KERNEL(k_pksub, A_PKSUB)
BENCH_KERNEL(k_pkmin, A_PKMIN)
BENCH_KERNEL(k_pkmaxu, A_PKMAXU)
BENCH_KERNEL(k_pklshr, A_PKLSHR)
BENCH_KERNEL(k_perm, A_PERM)
BENCH_KERNEL(k_fma, A_FMA)
BENCH_KERNEL(k_minf, A_MINF)
BENCH_KERNEL(k_pkminh, A_PKMINH)
BENCH_KERNEL(k_pkaddh, A_PKADDH)
BENCH_KERNEL(k_bfi, A_BFI)
BENCH_KERNEL(k_lshlor, A_LSHLOR)
BENCH_KERNEL(k_mad24, A_MAD24)
BENCH_KERNEL(k_sad, A_SAD)
BENCH_KERNEL(k_dot4, A_DOT4)
BENCH_KERNEL(k_bitop3, A_BITOP3)
BENCH_KERNEL(k_lshr, A_LSHR)
BENCH_KERNEL(k_cmpaddc, A_CMPADDC)
BENCH_KERNEL(k_mullo, A_MULLO)
BENCH_KERNEL(k_bcnt, A_BCNT)
BENCH_KERNEL(k_msad, A_MSAD)
BENCH_KERNEL(k_pkmadu16, A_PKMADU16)
BENCH_KERNEL(k_alignbyte, A_ALIGNBYTE)
BENCH_KERNEL(k_max3u, A_MAX3U)

typedef void (*kern_t)(unsigned *, unsigned);
struct T { const char *name; kern_t k; int instrPerAsm; };

int main()
{
    T tests[] = {{"v_add_u32", k_add, 1}, {"v_sub_u32_sdwa", k_sub_sdwa, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_and_b32", k_and, 1},
                 {"v_min3_i32", k_min3, 1}, {"v_min_i32", k_min, 1}, {"v_pk_sub_i16", k_pksub, 1}, {"v_pk_min_i16", k_pkmin, 1},
                 {"v_pk_max_u16", k_pkmaxu, 1}, {"v_pk_lshrrev_b16", k_pklshr, 1}, {"v_perm_b32", k_perm, 1}, {"v_fma_f32", k_fma, 1},
                 {"v_min_f32", k_minf, 1}, {"v_pk_min_f16", k_pkminh, 1}, {"v_pk_add_f16", k_pkaddh, 1}, {"v_bfi_b32", k_bfi, 1},
                 {"v_lshl_or_b32", k_lshlor, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_sad_u8", k_sad, 1}, {"v_dot4_u32_u8", k_dot4, 1},
                 {"v_bitop3_b32", k_bitop3, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_cmp+v_addc", k_cmpaddc, 2}, {"v_mul_lo_u32", k_mullo, 1},
                 {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_msad_u8", k_msad, 1}, {"v_pk_mad_u16", k_pkmadu16, 1}, {"v_alignbyte_b32", k_alignbyte, 1},
                 {"v_max3_u32", k_max3u, 1}};
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU = 32 waves/CU
    unsigned *d;
    hipMalloc(&d, blocks * 256 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-20s %10s %14s %s\n", "instruction", "ms", "Tlane-ops/s", "cycles/wave-instr/SIMD @2.4GHz");
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        double laneops = (double)blocks * 256 * ITER * 16 * t.instrPerAsm;
        double rate = laneops / (ms * 1e-3);
        double waveInstrPerSimd = laneops / 64 / (256.0 * 4);
        double cyc = (ms * 1e-3 * 2.4e9) / waveInstrPerSimd;
        printf("%-20s %10.3f %14.2f %8.2f\n", t.name, ms, rate / 1e12, cyc);
    }
    return 0;
}

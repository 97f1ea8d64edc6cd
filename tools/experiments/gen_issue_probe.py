#!/usr/bin/env python3
"""gen_issue_probe.py - writes tools/experiments/issue_probe.hip: one-variable-at-a-time probes of gfx950 vector-instruction ISSUE.

Question ("what's weak" 2): k_pyr_fast's inner bodies issue at ~3.3 real cycles per vector instruction where their
instruction classes price out at <= 2.6.  Every kernel below is ONE hand-allocated asm loop (explicit VGPR numbers, so the register
banks of every operand are known), timed in-kernel with s_memtime (shader cycles, independent of DVFS) around the loop; W workgroups of
256 threads per CU = W waves per SIMD.  Reported: shader cycles per vector instruction per SIMD = dt / (instructions per wave x W).

Variables: encoding size (VOP2 e32 4 B / +literal 8 B / VOP3 e64 8 B), source banks of 3-source VOP3 (reg mod 4 equal / distinct),
SGPR operand, opcode class, dependent chain vs 16 accumulators, LDS reads + s_waitcnt in the stream, waves per SIMD.
"""
import os

ACC0, NACC = 8, 16          # accumulators v8..v23
SRC0 = 24                   # sources v24..v31 (banks 0..3, 0..3)
NB = 32                     # vector instructions per loop body (unless a body says otherwise)


def acc(i):
    return ACC0 + (i % NACC)


def srcs_distinct(k):
    """two source registers whose banks differ from each other and from v<k>'s"""
    return SRC0 + (k + 1) % 4, SRC0 + 4 + (k + 2) % 4


def srcs_same(k):
    return SRC0 + k % 4, SRC0 + 4 + k % 4


bodies = {}


def body(name, lines, nvalu=None, note=""):
    bodies[name] = (lines, nvalu if nvalu is not None else len([l for l in lines if l.startswith("v_")]), note)


body("and_e32", [f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(NB)], note="VOP2, 4 bytes")
body("and_lit", [f"v_and_b32_e32 v{acc(i)}, 0x7f7f7f7f, v{acc(i)}" for i in range(NB)], note="VOP2 + literal, 8 bytes")
body("and_e64", [f"v_and_b32_e64 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(NB)], note="VOP3 encoding of the same op, 8 bytes")
body("and_sgpr", [f"v_and_b32_e32 v{acc(i)}, s21, v{acc(i)}" for i in range(NB)], note="VOP2 with an SGPR source")
body("and_samebank", [f"v_and_b32_e32 v{acc(i)}, v{srcs_same(acc(i))[0]}, v{acc(i)}" for i in range(NB)], note="VOP2, both sources in one bank")
body("add_e32", [f"v_add_u32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(NB)])
body("and_3reg", [f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)], note="dst != sources (no read of the previous result)")
body("bitop3_distinct", [f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]} bitop3:0x96" for i in range(NB)],
     note="3 VGPR sources in 3 banks")
body("bitop3_samebank", [f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_same(acc(i))[0]}, v{srcs_same(acc(i))[1]} bitop3:0x96" for i in range(NB)],
     note="3 VGPR sources in ONE bank")
body("bitop3_2same", [f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_same(acc(i))[0]}, v{srcs_distinct(acc(i))[1]} bitop3:0x96" for i in range(NB)],
     note="two of three sources in one bank")
body("bitop3_sgpr", [f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, s21 bitop3:0x96" for i in range(NB)], note="2 VGPR + 1 SGPR")
body("bitop3_rep", [f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[0]} bitop3:0x96" for i in range(NB)],
     note="src1 == src2 (one register read twice)")
body("or3_distinct", [f"v_or3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("and_or_distinct", [f"v_and_or_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("add3_distinct", [f"v_add3_u32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("xad_distinct", [f"v_xad_u32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("lshl_add_distinct", [f"v_lshl_add_u32 v{acc(i)}, v{acc(i)}, 1, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("alignbyte_distinct", [f"v_alignbyte_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, 1" for i in range(NB)])
body("alignbyte_samebank", [f"v_alignbyte_b32 v{acc(i)}, v{acc(i)}, v{srcs_same(acc(i))[0]}, 1" for i in range(NB)])
body("perm_distinct", [f"v_perm_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("min3_distinct", [f"v_min3_i32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("min_e32", [f"v_min_i32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(NB)])
body("mad24_distinct", [f"v_mad_u32_u24 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}" for i in range(NB)])
body("mul_u24_e32", [f"v_mul_u32_u24_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(NB)])
body("lshr_e32", [f"v_lshrrev_b32_e32 v{acc(i)}, 1, v{acc(i)}" for i in range(NB)])
body("mov_e32", [f"v_mov_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}" for i in range(NB)])
body("dep_chain_and", ["v_and_b32_e32 v8, v25, v8" if i % 2 else "v_add_u32_e32 v8, v26, v8" for i in range(NB)], note="ONE accumulator: every instruction reads the previous result")
body("dep_chain_bitop3", ["v_bitop3_b32 v8, v8, v25, v30 bitop3:0x96"] * NB, note="ONE accumulator, 3 sources in 3 banks")


def ring_like(reads, wait, nbanked=True):
    """six vector instructions per ring position as in fast_group_flags_lds (and-literal, add, sub, 2 x bitop3, alignbyte), 5 positions +
    2 fillers = 32 vector instructions; optionally the 2 LDS reads per position and the wait in front of their use"""
    out = []
    for p in range(5):
        P, Pl, td, tb = 8 + 4 * p, 9 + 4 * p, 10 + 4 * p, 11 + 4 * p   # v8..v27
        if reads == "ondemand":
            out += [f"ds_read2_b32 v[40:41], v7 offset0:{4 * p} offset1:{4 * p + 1}"]
            if wait:
                out += ["s_waitcnt lgkmcnt(0)"]
            out += [f"v_alignbyte_b32 v{P}, v41, v40, 1"]
        elif reads == "ahead":
            # the read for position p+1 is in flight while position p computes (two register pairs)
            a, b = (40, 41) if p % 2 == 0 else (42, 43)
            na, nb_ = (42, 43) if p % 2 == 0 else (40, 41)
            out += [f"ds_read2_b32 v[{na}:{nb_}], v7 offset0:{4 * p + 4} offset1:{4 * p + 5}", "s_waitcnt lgkmcnt(1)", f"v_alignbyte_b32 v{P}, v{b}, v{a}, 1"]
        else:
            out += [f"v_alignbyte_b32 v{P}, v{P}, v{29 + p % 3}, 1"]
        out += [f"v_and_b32_e32 v{Pl}, 0x7f7f7f7f, v{P}",
                f"v_add_u32_e32 v{td}, v28, v{Pl}",
                f"v_sub_u32_e32 v{tb}, v29, v{Pl}",
                f"v_bitop3_b32 v{td}, v{P}, v30, v{td} bitop3:0x4d",
                f"v_bitop3_b32 v{tb}, v31, v{P}, v{tb} bitop3:0x4d"]
    out += ["v_bitop3_b32 v32, v10, v14, v18 bitop3:0x80", "v_bitop3_b32 v33, v11, v15, v19 bitop3:0x80"]
    return out


body("ring_mix_nolds", ring_like("none", False), note="the ring test's instruction mix, operands in registers")
body("ring_mix_lds_wait0", ring_like("ondemand", True), note="+ ds_read2_b32 per position, s_waitcnt lgkmcnt(0) right behind it (as compiled)")
body("ring_mix_lds_ahead", ring_like("ahead", True), note="+ ds_read2_b32 per position issued one position ahead, counted wait")
body("ring_mix_sgprconst", [l.replace("0x7f7f7f7f", "s21") for l in ring_like("none", False)], note="ring mix, the and-mask in an SGPR (4-byte encoding)")

ALLREGS = ", ".join(f'"v{i}"' for i in range(7, 48))


def kernel(name, lines, nvalu):
    asm = "\n".join(f'        "{l}\\n"' for l in lines)
    init = "\n".join(f'        "v_add_u32_e32 v{i}, {hex((i * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF)}, %[tid]\\n"' for i in range(8, 48))
    fold = "\n".join(f'        "v_xor_b32_e32 v8, v{i}, v8\\n"' for i in range(9, 44))
    return f'''
__global__ __launch_bounds__(256) void k_{name}(unsigned long long *out, int iters)
{{
    extern __shared__ unsigned char lds[];
    for (unsigned i = threadIdx.x; i < 1024; i += 256) ((volatile unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    unsigned accv;
    const unsigned tid = threadIdx.x, la = (threadIdx.x & 63) * 4;
    asm volatile(
{init}
        "v_mov_b32_e32 v7, %[la]\\n"
        "s_mov_b32 s21, 0x7f7f7f7f\\n"
        "s_mov_b32 s20, %[iters]\\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\\n"
        "s_barrier\\n"
        "s_memtime %[t0]\\n"
        "s_memrealtime %[r0]\\n"
        "s_waitcnt lgkmcnt(0)\\n"
        ".Lloop_{name}:\\n"
{asm}
        "s_sub_u32 s20, s20, 1\\n"
        "s_cmp_lg_u32 s20, 0\\n"
        "s_cbranch_scc1 .Lloop_{name}\\n"
        "s_waitcnt lgkmcnt(0)\\n"
        "s_memtime %[t1]\\n"
        "s_memrealtime %[r1]\\n"
        "s_waitcnt lgkmcnt(0)\\n"
{fold}
        "v_mov_b32_e32 %[accv], v8\\n"
        : [t0] "=&s"(t0), [r0] "=&s"(r0), [t1] "=&s"(t1), [r1] "=&s"(r1), [accv] "=&v"(accv)
        : [tid] "v"(tid), [la] "v"(la), [iters] "s"(iters)
        : {ALLREGS}, "s20", "s21", "scc", "vcc", "memory");
    const unsigned w = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {{ out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }}
    if (accv == 0x12345u && iters < 0) out[0] = accv;
}}
'''


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    src = ['// GENERATED by gen_issue_probe.py - do not edit.  Build: hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe',
           '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <cstring>', '#include <vector>', '#include <algorithm>']
    for name, (lines, nvalu, note) in bodies.items():
        src.append(kernel(name, lines, nvalu))
    src.append('typedef void (*kern_t)(unsigned long long *, int);')
    src.append('struct Probe { const char *name; kern_t k; int nvalu; int bytes; const char *note; };')
    src.append('static Probe probes[] = {')
    for name, (lines, nvalu, note) in bodies.items():
        src.append(f'    {{"{name}", k_{name}, {nvalu}, 0, "{note}"}},')
    src.append('};')
    src.append(r'''
int main(int argc, char **argv)
{
    const char *only = argc > 1 ? argv[1] : nullptr;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    unsigned long long *d;
    const int maxWaves = cus * 32;
    if (hipMalloc(&d, (size_t)maxWaves * 16) != hipSuccess) return 1;
    std::vector<unsigned long long> h((size_t)maxWaves * 2);
    printf("%-22s %3s %10s %8s %8s %9s  %s\n", "probe", "W", "cyc/instr", "min", "max", "clock GHz", "note  (W = waves per SIMD; cycles = shader cycles per vector instruction per SIMD)");
    for (auto &p : probes) {
        if (only && !strstr(p.name, only)) continue;
        for (int W : {8, 4, 2, 1}) {
            if (W != 8 && !(strstr(p.name, "and_e32") || strstr(p.name, "bitop3_distinct") || strstr(p.name, "ring_mix") || strstr(p.name, "dep_chain") || strstr(p.name, "alignbyte_distinct"))) continue;
            const size_t ldsBytes = W == 8 ? 16 * 1024 : W == 4 ? 36 * 1024 : W == 2 ? 72 * 1024 : 150 * 1024;  // W workgroups fit a CU's 160 KB
            hipFuncSetAttribute((const void *)p.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
            const int blocks = cus * W, iters = 24000 / W;
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), ldsBytes, 0, d, iters);   // warm-up
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), ldsBytes, 0, d, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", p.name); return 1; }
            hipMemcpy(h.data(), d, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost);
            std::vector<double> c, clk;
            for (int w = 0; w < blocks * 4; ++w) {
                c.push_back((double)h[2 * w] / ((double)iters * p.nvalu * W));
                if (h[2 * w + 1]) clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
            }
            std::sort(c.begin(), c.end()); std::sort(clk.begin(), clk.end());
            printf("%-22s %3d %10.3f %8.3f %8.3f %9.2f  %s\n", p.name, W, c[c.size() / 2], c.front(), c.back(), clk.empty() ? 0.0 : clk[clk.size() / 2], p.note);
            fflush(stdout);
        }
    }
    return 0;
}
''')
    with open(os.path.join(here, "issue_probe.hip"), "w") as f:
        f.write("\n".join(src))


if __name__ == "__main__":
    main()

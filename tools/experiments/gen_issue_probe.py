#!/usr/bin/env python3
"""gen_issue_probe.py - writes issue_probe.hip (into --out-dir, default a temp directory; never into the tree): one-variable-at-a-time probes of gfx950 vector-instruction ISSUE.

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


body("alt_and_bitop3", [(f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" if i % 2 else
                         f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]} bitop3:0x96") for i in range(NB)],
     note="4-byte and 8-byte encodings alternating")
body("alt_and_alignbyte", [(f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" if i % 2 else
                            f"v_alignbyte_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, 1") for i in range(NB)], note="VOP2 and the slow VOP3 class alternating")
body("and3_e32_x3", [f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" for i in range(3 * NB)], note="96 instructions per loop body (loop overhead / fetch of a longer body)")
# LDS-only streams: 16 reads, one wait.  v7 = 4 * lane (dword-aligned, consecutive lanes = consecutive dwords); v6 = 8 * lane; v5 = 16 * lane
def lds_body(ins):
    out = []
    for k in range(16):
        d = 40 + (k % 2) * 4
        out.append(ins.format(k=k, k1=k + 1, o4=4 * k, o8=8 * k, o8p4=8 * k + 4, o16=16 * k, d=d, d1=d + 1, d2=d + 2, d3=d + 3))
    return out + ["s_waitcnt lgkmcnt(0)"]
body("lds_read_b32", lds_body("ds_read_b32 v{d}, v7 offset:{o4}"), nvalu=16, note="LDS only: 16 x ds_read_b32 (lane stride 4 B), one wait")
body("lds_read2_b32", lds_body("ds_read2_b32 v[{d}:{d}+1], v7 offset0:{k} offset1:{k}+1".replace("{d}+1", "{d1}").replace("{k}+1", "{k1}")), nvalu=16, note="LDS only: ds_read2_b32 of two adjacent dwords")
body("lds_read_b64_al8", lds_body("ds_read_b64 v[{d}:{d1}], v6 offset:{o8}"), nvalu=16, note="LDS only: ds_read_b64, 8-byte aligned (lane stride 8 B)")
body("lds_read_b64_al4", lds_body("ds_read_b64 v[{d}:{d1}], v6 offset:{o8p4}"), nvalu=16, note="LDS only: ds_read_b64 at addresses = 4 mod 8")
body("lds_read_b64_s4", lds_body("ds_read_b64 v[{d}:{d1}], v7 offset:{o4}"), nvalu=16, note="LDS only: ds_read_b64, lane stride 4 B (overlapping, as the ring test's pairs)")
body("lds_read_b128_al16", lds_body("ds_read_b128 v[{d}:{d3}], v5 offset:{o16}"), nvalu=16, note="LDS only: ds_read_b128, 16-byte aligned")
body("lds_read_b96_s4", lds_body("ds_read_b96 v[{d}:{d2}], v7 offset:{o4}"), nvalu=16, note="LDS only: ds_read_b96, lane stride 4 B (3 dwords of a row window)")
body("lds_read_b128_s4", lds_body("ds_read_b128 v[{d}:{d3}], v7 offset:{o4}"), nvalu=16, note="LDS only: ds_read_b128, lane stride 4 B, dword-aligned only")
# ---- mixing rules: runs of k full-rate instructions (F = v_and_b32_e32 / v_bitop3) followed by ONE instruction of another class, 96 per body
def F(i):
    return f"v_and_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}" if i % 2 else \
           f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]} bitop3:0x96"
SLOW = {
    "alignbyte": lambda i: f"v_alignbyte_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, 1",
    "sgprsrc": lambda i: f"v_and_b32_e32 v{acc(i)}, s21, v{acc(i)}",
    "min": lambda i: f"v_min_i32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "mul24": lambda i: f"v_mul_u32_u24_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "or3": lambda i: f"v_or3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}",
    "cvt": lambda i: f"v_cvt_f32_u32_e32 v{acc(i)}, v{acc(i)}",
    "lit": lambda i: f"v_and_b32_e32 v{acc(i)}, 0x7f7f7f7f, v{acc(i)}",
    "cndmask": lambda i: f"v_cndmask_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}, vcc",
    "cmp": lambda i: f"v_cmp_lt_u32_e32 vcc, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "readlane": lambda i: f"v_readfirstlane_b32 s22, v{acc(i)}",
    "dpp": lambda i: f"v_add_u32_dpp v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]} row_shr:1 row_mask:0xf bank_mask:0xf",
    "mbcnt": lambda i: f"v_mbcnt_lo_u32_b32 v{acc(i)}, -1, v{acc(i)}",
    "lshl_add": lambda i: f"v_lshl_add_u32 v{acc(i)}, v{acc(i)}, 1, v{srcs_distinct(acc(i))[1]}",
    "mad_u64": lambda i: f"v_mad_u64_u32 v[44:45], s[22:23], v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v[44:45]",
}
for sname, fn in SLOW.items():
    body(f"pure_{sname}", [fn(i) for i in range(96)], note=f"96 x {sname}")
    for k in (1, 3, 7):
        lines = []
        for i in range(96):
            lines.append(fn(i) if i % (k + 1) == k else F(i))
        body(f"F{k}_{sname}", lines, note=f"runs of {k} full-rate instructions + one {sname}")
# ---- how long does the slow rate last after one slow-class instruction?  runs of k full-rate instructions + 1 v_alignbyte, k up to 255
for k in (15, 31, 63, 127, 255):
    n = max(256, k + 1)
    body(f"F{k}_alignbyte", [SLOW["alignbyte"](i) if i % (k + 1) == k else F(i) for i in range(n)], note=f"runs of {k} full-rate instructions + one v_alignbyte")
# clustered: the same 1 : 7 ratio as F7_alignbyte, but the slow instructions together
for g in (4, 16, 32):
    body(f"S{g}_F{7 * g}", [SLOW["alignbyte"](i) for i in range(g)] + [F(i) for i in range(7 * g)], note=f"{g} x v_alignbyte together, then {7 * g} full-rate instructions")
# ---- class of further opcodes, by their effect on the neighbours: F7_<op> ~ 2.2 -> full rate, ~ 4 -> slow class
CLASSIFY = {
    "xor": lambda i: f"v_xor_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "or": lambda i: f"v_or_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "sub": lambda i: f"v_sub_u32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "subrev": lambda i: f"v_subrev_u32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "lshl": lambda i: f"v_lshlrev_b32_e32 v{acc(i)}, 3, v{acc(i)}",
    "lshl_v": lambda i: f"v_lshlrev_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "ashr": lambda i: f"v_ashrrev_i32_e32 v{acc(i)}, 3, v{acc(i)}",
    "mov": lambda i: f"v_mov_b32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}",
    "not": lambda i: f"v_not_b32_e32 v{acc(i)}, v{acc(i)}",
    "add_f32": lambda i: f"v_add_f32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "mul_f32": lambda i: f"v_mul_f32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "fmac_f32": lambda i: f"v_fmac_f32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}",
    "fma_f32": lambda i: f"v_fma_f32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}",
    "max_f32": lambda i: f"v_max_f32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "cndmask_s": lambda i: f"v_cndmask_b32_e64 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}, s[24:25]",
    "bfe": lambda i: f"v_bfe_u32 v{acc(i)}, v{acc(i)}, 3, 8",
    "and_or": lambda i: f"v_and_or_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}",
    "lshl_or": lambda i: f"v_lshl_or_b32 v{acc(i)}, v{acc(i)}, 3, v{srcs_distinct(acc(i))[0]}",
    "add_co": lambda i: f"v_add_co_u32_e32 v{acc(i)}, vcc, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "mul_lo": lambda i: f"v_mul_lo_u32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}",
    "sdwa": lambda i: f"v_add_u32_sdwa v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
    "bcnt": lambda i: f"v_bcnt_u32_b32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "max_u32": lambda i: f"v_max_u32_e32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
    "pk_add_u16": lambda i: f"v_pk_add_u16 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}",
    "bitop3_lit": lambda i: f"v_bitop3_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, s21 bitop3:0x96",
    "alignbit": lambda i: f"v_alignbit_b32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, 8",
    "and_inl": lambda i: f"v_and_b32_e32 v{acc(i)}, 63, v{acc(i)}",
    "bfi": lambda i: f"v_bfi_b32 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{srcs_distinct(acc(i))[1]}, v{acc(i)}",
    "add_lshl": lambda i: f"v_add_lshl_u32 v{acc(i)}, v{acc(i)}, v{srcs_distinct(acc(i))[0]}, 2",
    "sub_i32_e64": lambda i: f"v_sub_u32_e64 v{acc(i)}, v{srcs_distinct(acc(i))[0]}, v{acc(i)}",
}
for cname, fn in CLASSIFY.items():
    body(f"pure_{cname}", [fn(i) for i in range(96)], note=f"96 x {cname}")
    body(f"F7_{cname}", [fn(i) if i % 8 == 7 else F(i) for i in range(96)], note=f"runs of 7 full-rate instructions + one {cname}")
# ---- what ends the slow rate?  one v_alignbyte, ONE candidate instruction, then 126 full-rate instructions (and the loop's s_sub / s_cmp / s_cbranch)
RESET = {
    "nothing": [],
    "s_nop0": ["s_nop 0"],
    "s_nop7": ["s_nop 7"],
    "waitcnt_lgkm": ["s_waitcnt lgkmcnt(0)"],
    "waitcnt_vm": ["s_waitcnt vmcnt(0)"],
    "s_add": ["s_add_u32 s22, s22, 1"],
    "s_mov": ["s_mov_b32 s22, 5"],
    "s_cmp": ["s_cmp_lg_u32 s22, 0"],
    "s_branch_next": ["s_branch .Lnext_{name}_{u}", ".Lnext_{name}_{u}:"],
    "s_cbranch_nottaken": ["s_cmp_eq_u32 s20, 0", "s_cbranch_scc1 .Lloop_{name}"],
    "s_cbranch_taken_next": ["s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 .Lnext_{name}_{u}", ".Lnext_{name}_{u}:"],
    "ds_read_wait": ["ds_read_b32 v40, v7", "s_waitcnt lgkmcnt(0)"],
    "ds_read_nowait": ["ds_read_b32 v40, v7"],
    "s_barrier": ["s_barrier"],
    "s_sleep": ["s_sleep 1"],
    "s_setprio": ["s_setprio 0"],
    "v_nop": ["v_nop"],
    "v_mov": ["v_mov_b32_e32 v41, v24"],
    "s_setreg": ["s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 4), 0"],
    "s_waitcnt_depctr": ["s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"],
    "two_F_then_nop": None,
}
for rname, cand in RESET.items():
    lines = []
    for rep in range(2):   # two such sequences per loop body (256 vector instructions)
        lines.append(SLOW["alignbyte"](0))
        if cand is None:
            lines += [F(1), F(2), "s_nop 0"]
            lines += [F(i) for i in range(3, 128)]
        else:
            lines += [c.replace("{u}", str(rep)) for c in cand]
            lines += [F(i) for i in range(1, 128)]
    body(f"R_{rname}", ["@@" + l for l in lines], nvalu=len([l for l in lines if l.startswith("v_")]), note=f"v_alignbyte, then [{rname}], then 127 full-rate instructions (x 2 per loop body)")
# ---- the same mixes with the s_nop 0 that tools/sq_slowmode.py inserts behind every slow-class instruction: per wave the slow rate should
# then hold for the slow instruction alone - unless the slow rate is a state of the SIMD that the OTHER waves' instructions keep alive
ALLCLS = dict(SLOW); ALLCLS.update(CLASSIFY)
for sname in ("alignbyte", "min", "cmp", "lshl", "mul24", "cvt", "mul_lo", "lshl_add"):
    fn = ALLCLS[sname]
    for k in (1, 3, 7, 15):
        lines = []
        for i in range(96):
            if i % (k + 1) == k:
                lines += [fn(i), "s_nop 0"]
            else:
                lines.append(F(i))
        body(f"N{k}_{sname}", lines, note=f"runs of {k} full-rate instructions + one {sname} + s_nop 0")
# ---- scalar instructions beside the vector stream: k_pyr_fast issues one scalar instruction per 2.7 vector instructions
def with_salu(k, salu):
    out = []
    i = 0
    while i < 96:
        out += [F(i + t) for t in range(k)]
        out += salu
        i += k
    return out
body("V1_sadd", with_salu(1, ["s_add_u32 s22, s22, 1"]), note="1 full-rate vector instruction : 1 s_add_u32")
body("V2_sadd", with_salu(2, ["s_add_u32 s22, s22, 1"]), note="2 : 1 s_add_u32")
body("V3_sadd", with_salu(3, ["s_add_u32 s22, s22, 1"]), note="3 : 1 s_add_u32")
body("V3_sadd2", with_salu(3, ["s_add_u32 s22, s22, 1", "s_and_b32 s23, s22, 7"]), note="3 : 2 scalar ALU")
body("V3_exec", with_salu(3, ["s_and_saveexec_b64 s[24:25], vcc", "s_or_b64 exec, exec, s[24:25]"]), note="3 : s_and_saveexec_b64 + s_or_b64 exec (vcc = all lanes)")
body("V3_branch", with_salu(3, ["s_cbranch_execz .Lloop_{name}"]), note="3 : 1 s_cbranch_execz, never taken")
body("V6_exec_branch", with_salu(6, ["s_and_saveexec_b64 s[24:25], vcc", "s_cbranch_execz .Lloop_{name}", "s_or_b64 exec, exec, s[24:25]"]),
     note="6 : saveexec + branch (not taken) + restore: a divergent if")
body("V3_waitcnt", with_salu(3, ["s_waitcnt lgkmcnt(0)"]), note="3 : 1 s_waitcnt with nothing outstanding")
body("pure_F", [F(i) for i in range(96)], note="96 x full-rate (v_and_b32_e32 / v_bitop3 alternating)")
body("F7_salu", sum([[F(8 * j + t) for t in range(7)] + ["s_add_u32 s22, s22, 1"] for j in range(12)], []), note="7 full-rate + one s_add_u32 (84 vector instructions)")
body("F7_nop", sum([[F(8 * j + t) for t in range(7)] + ["s_nop 0"] for j in range(12)], []), note="7 full-rate + s_nop 0")
body("F7_waitcnt", sum([[F(8 * j + t) for t in range(7)] + ["s_waitcnt lgkmcnt(0)"] for j in range(12)], []), note="7 full-rate + s_waitcnt lgkmcnt(0) with nothing outstanding")
body("F7_ldsread", sum([[F(8 * j + t) for t in range(7)] + [f"ds_read_b32 v{40 + j % 4}, v7 offset:{16 * j}"] for j in range(12)] + [["s_waitcnt lgkmcnt(0)"]], []),
     note="7 full-rate + one ds_read_b32, one wait per body (84 vector instructions, 12 reads)")
body("F7_ldsread_wait", sum([[F(8 * j + t) for t in range(7)] + [f"ds_read_b32 v{40 + j % 4}, v7 offset:{16 * j}", "s_waitcnt lgkmcnt(0)"] for j in range(12)], []),
     note="7 full-rate + ds_read_b32 + s_waitcnt lgkmcnt(0)")
body("ring_mix_nolds", ring_like("none", False), note="the ring test's instruction mix, operands in registers")
body("ring_mix_lds_wait0", ring_like("ondemand", True), note="+ ds_read2_b32 per position, s_waitcnt lgkmcnt(0) right behind it (as compiled)")
body("ring_mix_lds_ahead", ring_like("ahead", True), note="+ ds_read2_b32 per position issued one position ahead, counted wait")
body("ring_mix_sgprconst", [l.replace("0x7f7f7f7f", "s21") for l in ring_like("none", False)], note="ring mix, the and-mask in an SGPR (4-byte encoding)")

ALLREGS = ", ".join(f'"v{i}"' for i in range(5, 48))


def kernel(name, lines, nvalu):
    lines = [l[2:].replace("{name}", name) if l.startswith("@@") else l.replace("{name}", name) for l in lines]
    asm = "\n".join(f'        "{l}\\n"' for l in lines)
    init = "\n".join(f'        "v_add_u32_e32 v{i}, {hex((i * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF)}, %[tid]\\n"' for i in range(8, 48))
    fold = "\n".join(f'        "v_xor_b32_e32 v8, v{i}, v8\\n"' for i in range(9, 44))
    return f'''
__global__ __launch_bounds__(512) void k_{name}(unsigned long long *out, int iters)
{{
    extern __shared__ unsigned char lds[];
    for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) ((volatile unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    unsigned accv;
    const unsigned tid = threadIdx.x, la = (threadIdx.x & 63) * 4;
    asm volatile(
{init}
        "v_mov_b32_e32 v7, %[la]\\n"
        "v_lshlrev_b32_e32 v6, 1, v7\\n"
        "v_lshlrev_b32_e32 v5, 2, v7\\n"
        "s_mov_b32 s21, 0x7f7f7f7f\\n"
        "s_mov_b64 vcc, -1\\n"
        "s_mov_b32 s24, 0x0f0f3355\\n"
        "s_mov_b32 s25, 0x00ff5533\\n"
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
        : {ALLREGS}, "s20", "s21", "s22", "s23", "s24", "s25", "scc", "vcc", "memory");
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {{ out[4 * w] = t1 - t0; out[4 * w + 1] = r1 - r0; out[4 * w + 2] = r0; out[4 * w + 3] = r1; }}
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
    if (hipMalloc(&d, (size_t)maxWaves * 32) != hipSuccess) return 1;
    std::vector<unsigned long long> h((size_t)maxWaves * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d.  W = waves per SIMD, T = threads per workgroup.  in-kernel: s_memtime ticks of one wave / (its instructions x W), median [min max] over the waves;\n"
           "wall: hipEvent time of the launch x in-kernel clock / (instructions per SIMD); start spread = (latest - earliest wave start) / wave duration.\n", cus);
    printf("%-22s %2s %4s %9s %7s %7s %9s %7s %7s  %s\n", "probe", "W", "T", "in-kernel", "min", "max", "wall", "clk GHz", "spread", "note");
    for (auto &p : probes) {
        if (only && !strstr(p.name, only)) continue;
        for (int cfg = 0; cfg < 6; ++cfg) {
            const int W = cfg == 0 ? 8 : cfg == 1 ? 8 : cfg == 2 ? 4 : cfg == 3 ? 2 : cfg == 4 ? 1 : 16, T = cfg == 1 ? 512 : 256;
            if (cfg == 5) continue;
            const bool sweep = strstr(p.name, "and_e32") || strstr(p.name, "bitop3_distinct") || strstr(p.name, "ring_mix") || strstr(p.name, "alignbyte_distinct") || strstr(p.name, "lds_read");
            const bool mixing = !strncmp(p.name, "pure_", 5) || !strncmp(p.name, "R_", 2) || (p.name[0] == 'V' && p.name[1] >= '0' && p.name[1] <= '9') || (p.name[0] == 'N' && p.name[1] >= '0' && p.name[1] <= '9') || (p.name[0] == 'F' && p.name[1] >= '0' && p.name[1] <= '9');
            if (mixing) { if (cfg != 1 && cfg != 3) continue; }   // the kernel's shape (8 waves per SIMD, 512-thread workgroups) and 2 waves per SIMD
            else if (cfg >= 1 && !sweep) continue;
            const int wavesPerBlock = T / 64, blocksPerCu = W * 4 / wavesPerBlock;
            const size_t ldsBytes = (size_t)(152 * 1024 / blocksPerCu) & ~(size_t)1023;   // exactly blocksPerCu workgroups fit a CU's 160 KB
            hipFuncSetAttribute((const void *)p.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
            const int blocks = cus * blocksPerCu, iters = 24000 / W;
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(T), ldsBytes, 0, d, iters);   // warm-up
            hipEventRecord(e0);
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(T), ldsBytes, 0, d, iters);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", p.name); return 1; }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const int nw = blocks * wavesPerBlock;
            hipMemcpy(h.data(), d, (size_t)nw * 32, hipMemcpyDeviceToHost);
            std::vector<double> c, clk;
            unsigned long long r0min = ~0ull, r0max = 0, durSum = 0;
            for (int w = 0; w < nw; ++w) {
                c.push_back((double)h[4 * w] / ((double)iters * p.nvalu * W));
                if (h[4 * w + 1]) clk.push_back((double)h[4 * w] / (double)h[4 * w + 1] * 0.1);
                r0min = std::min(r0min, h[4 * w + 2]); r0max = std::max(r0max, h[4 * w + 2]); durSum += h[4 * w + 1];
            }
            std::sort(c.begin(), c.end()); std::sort(clk.begin(), clk.end());
            const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
            const double wall = ms * 1e-3 * ghz * 1e9 / ((double)iters * p.nvalu * W);
            printf("%-22s %2d %4d %9.3f %7.3f %7.3f %9.3f %7.2f %7.2f  %s\n", p.name, W, T, c[c.size() / 2], c.front(), c.back(), wall, ghz,
                   (double)(r0max - r0min) / ((double)durSum / nw), p.note);
            fflush(stdout);
        }
    }
    return 0;
}
''')
    # the 52 k generated lines are NOT tracked: they are written where they are used (default: a fresh temp directory) and, with
    # --build, compiled there; tools/experiments/run_issue_probe.sh does both and runs the probe.
    import argparse, subprocess, tempfile
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=None, help="directory for issue_probe.hip (default: a new temp directory)")
    ap.add_argument("--build", action="store_true", help="also compile it with hipcc for gfx950")
    a = ap.parse_args()
    out = a.out_dir or tempfile.mkdtemp(prefix="issue_probe_")
    os.makedirs(out, exist_ok=True)
    hip = os.path.join(out, "issue_probe.hip")
    with open(hip, "w") as f:
        f.write("\n".join(src))
    exe = os.path.join(out, "issue_probe")
    if a.build:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", hip, "-o", exe])
    print(exe if a.build else hip)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""sq_slowmode.py - gfx950 vector-issue "slow mode" tools for compiler-generated assembly (.s).

Measured on MI355X (tools/experiments/gen_issue_probe.py -> profiles/r04_issue_probe_v3..v5*.txt):
  * a small set of vector opcodes issues at ~2.1 cycles per wave64 instruction per SIMD ("full rate": v_and/or/xor/not, v_add/sub_u32,
    v_lshrrev/v_ashrrev, v_mov, v_bitop3, v_cndmask, v_add/mul/fma_f32 ...); every other vector opcode at ~4.1 ("slow class":
    v_lshlrev, v_alignbyte/bit, min/max, every integer multiply, v_cvt, v_cmp, v_readlane, DPP / SDWA forms, v_mbcnt, v_lshl_add, v_lshl_or,
    v_and_or, v_or3, v_add3, v_bfe, v_bfi, v_perm, v_bcnt, v_pk_* ...);
  * AFTER a slow-class instruction the wave keeps issuing ALL its following vector instructions at the slow rate - 127 full-rate
    instructions behind one v_alignbyte run at 4.1 cycles each - until it issues ANY non-vector instruction (s_nop, s_waitcnt, a scalar ALU
    op, a branch, an LDS / memory instruction; also v_mov_b32, but not v_nop).  An `s_nop 0` is free (separate issue port).

`annotate` reports how many full-rate instructions of a .s file sit in such a shadow; `fix` inserts `s_nop 0` at every slow -> full-rate
transition that has no natural reset in between.  The Makefile pipes the device assembly of every .hip source through `fix`.
"""
import re
import sys

FULL_RATE = {
    "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_mov_b32", "v_bitop3_b32", "v_cndmask_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32",
    "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_xnor_b32",
}
RESETTING_VALU = {"v_mov_b32"}          # measured: a v_mov_b32 ends the slow rate like a scalar instruction does
NOT_RESETTING = {"v_nop"}


def opcode(line):
    s = line.strip()
    if not s or s[0] in ".;#/" or s.endswith(":"):
        return None
    m = re.match(r"([a-z_0-9]+)", s)
    return m.group(1) if m else None


def base(op):
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if op.endswith(suf):
            return op[: -len(suf)], suf
    return op, ""


def is_valu(op):
    return op.startswith("v_")


def is_full_rate(op, line):
    b, suf = base(op)
    if suf in ("_dpp", "_sdwa") or " row_" in line or "quad_perm" in line or "_sel:" in line:
        return False
    return b in FULL_RATE


def walk(lines):
    """yields (index, opcode, kind) with kind in 'slow', 'full', 'full_in_shadow', 'reset', 'other'"""
    slow_mode = False
    for i, ln in enumerate(lines):
        op = opcode(ln)
        if op is None:
            continue
        if is_valu(op):
            b, _ = base(op)
            if b in NOT_RESETTING:
                yield i, op, "other"
            elif is_full_rate(op, ln):
                if slow_mode and b not in RESETTING_VALU:
                    yield i, op, "full_in_shadow"
                else:
                    slow_mode = False if b in RESETTING_VALU else slow_mode
                    yield i, op, "full"
            else:
                slow_mode = True
                yield i, op, "slow"
        else:
            slow_mode = False
            yield i, op, "reset"


def annotate(path):
    lines = open(path).read().splitlines()
    kernel, stats = None, {}
    for ln in lines:
        pass
    cur = "?"
    names = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", ln)
        if m:
            names[i] = m.group(1)
    order = sorted(names)
    import bisect
    for i, op, kind in walk(lines):
        k = bisect.bisect_right(order, i) - 1
        name = names[order[k]] if k >= 0 else "?"
        st = stats.setdefault(name, {"slow": 0, "full": 0, "full_in_shadow": 0})
        if kind in st:
            st[kind] += 1
    for name, st in stats.items():
        tot = st["slow"] + st["full"] + st["full_in_shadow"]
        if tot:
            print(f"{name[:70]:70s} vector {tot:6d}  slow-class {st['slow']:6d}  full-rate {st['full']:6d}  full-rate in a slow shadow {st['full_in_shadow']:6d} ({100 * st['full_in_shadow'] / tot:.0f} %)")


def fix(src, dst):
    lines = open(src).read().splitlines()
    out, n = [], 0
    pending_slow = False   # the last vector instruction was slow-class and nothing has reset the mode since
    for ln in lines:
        op = opcode(ln)
        if op is not None:
            if is_valu(op):
                b, _ = base(op)
                if b in NOT_RESETTING:
                    pass
                elif is_full_rate(op, ln):
                    if pending_slow and b not in RESETTING_VALU:
                        out.append("\ts_nop 0")
                        n += 1
                    pending_slow = False
                else:
                    pending_slow = True
            else:
                pending_slow = False
        out.append(ln)
    open(dst, "w").write("\n".join(out) + "\n")
    return n


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "annotate":
        annotate(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "fix":
        print(f"sq_slowmode: {fix(sys.argv[2], sys.argv[3])} s_nop inserted into {sys.argv[3]}", file=sys.stderr)
    else:
        sys.exit("usage: sq_slowmode.py annotate file.s | fix in.s out.s")

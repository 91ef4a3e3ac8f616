"""Static instruction mix of one kernel in a hipcc -S dump (`hipcc -S --cuda-device-only`):
counts by class and a crude issue-cycle estimate (fp64 VALU 4 cycles per wave64 instruction on a
SIMD-32... the vector-instruction ISSUE cost row of MI355X_MICROARCH.md), to tell a compute floor
from a latency problem.  Loops are counted once (straight-line view).

usage: isa_stats.py kernels.s <mangled-name substring> [...]
"""
import collections
import re
import sys


def kernel_bodies(path):
    name, body = None, []
    for ln in open(path, errors="replace"):
        if ln.startswith("_Z") and "; @_Z" in ln:
            name, body = ln.split(":")[0], []
            continue
        if name is None:
            continue
        s = ln.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith("s_endpgm"):
            if s.startswith("s_endpgm"):
                body.append("s_endpgm")
            yield name, body
            name = None
            continue
        if s and not s.startswith((";", ".", "BB", "L")) and not s.endswith(":"):
            body.append(s.split()[0])


def classify(op):
    if op.startswith("v_") and ("_f64" in op):
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
            return "valu_f64_trans", 16
        return "valu_f64", 4
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans32", 8
    if op.startswith("v_"):
        return "valu_other", 4
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0
    if op.startswith("s_waitcnt"):
        return "waitcnt", 0
    if op.startswith("s_barrier"):
        return "barrier", 0
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernel_bodies(path):
        if not any(p in name for p in pats):
            continue
        cnt, cyc = collections.Counter(), 0
        ops = collections.Counter()
        for op in body:
            k, c = classify(op)
            cnt[k] += 1
            cyc += c
            ops[op] += 1
        print(name[:110])
        print("  ", dict(cnt), "valu issue cycles (straight line):", cyc)
        print("  top:", ", ".join(f"{o}:{n}" for o, n in ops.most_common(14)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static instruction census of one kernel in a device assembly file (hipcc --cuda-device-only -S): vector ALU by class,
scalar, LDS, global memory, per basic block.  The quadrature update is bound by vector issue; its kernels are straight
line code apart from the viscosity / eigen-decomposition branches, so the static count of a path is what a wavefront
issues - instruction budgets can be worked on without a GPU.
usage: isa_count.py file.s kernel-name-substring [--blocks]"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            if "fma" in op or "mul" in op or "add" in op or "mac" in op:
                return "valu_f64_arith"
            return "valu_f64_other"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "valu_lane"
        if "dpp" in op or "permlane" in op:
            return "valu_dpp"
        return "valu_32"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")):
            return "s_ctrl"
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem" if not op.startswith("scratch_") else "scratch"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    blocks = OrderedDict()
    cur, inside = None, False
    for line in open(path):
        if not inside:
            if re.match(r"^[_A-Za-z0-9]*%s[_A-Za-z0-9]*:" % re.escape(name), line):
                inside, cur = True, "entry"
                blocks[cur] = Counter()
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if m and not line.strip().startswith((";", ".")):
            op = m.group(1)
            blocks[cur][classify(op)] += 1
            if "dpp" in line and classify(op).startswith("valu") and classify(op) != "valu_dpp":
                blocks[cur]["(of which dpp operand)"] += 1
    tot = Counter()
    for b, c in blocks.items():
        tot.update(c)
    valu = sum(v for k, v in tot.items() if k.startswith("valu"))
    print("kernel %s: %d blocks; VALU %d, SALU %d, LDS %d, VMEM %d, SMEM %d, scratch %d" %
          (name, len(blocks), valu, tot["salu"], tot["lds"], tot["vmem"], tot["smem"], tot["scratch"]))
    for k in sorted(tot):
        print("   %-24s %6d" % (k, tot[k]))
    if "--blocks" in sys.argv:
        for b, c in blocks.items():
            v = sum(x for k, x in c.items() if k.startswith("valu"))
            if v + c["lds"] + c["vmem"] > 10:
                print("   block %-12s VALU %5d (f64 arith %5d) SALU %4d LDS %4d VMEM %3d" % (b, v, c["valu_f64_arith"], c["salu"], c["lds"], c["vmem"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static instruction census of one kernel BY SOURCE REGION: the device assembly is compiled with -gline-tables-only, every
instruction is attributed to the innermost source line the compiler names for it (.loc), and lines are grouped by the function
they sit in or, inside the kernel's own body, by its `// ---- Pn:` phase markers.  Columns: fp64 arithmetic (fma / mul / add),
other fp64 (rcp, rsq, max, cmp, ldexp ...), 64-bit moves, 32-bit moves, selects, other 32-bit vector (addresses, masks), scalar ALU,
LDS, global memory.  Static counts over ALL paths of the kernel (both sides of every branch); what a wavefront issues is less.
usage: isa_by_source.py <file.hip> <kernel-name-substring> [extra hipcc flags...]   (run in laghos_amd/csrc; needs hipcc only)"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict

BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=fast",
        "--cuda-device-only", "-gline-tables-only", "-S"]
COLS = ["f64 arith", "f64 other", "mov64", "mov32", "select", "valu32", "salu", "lds", "vmem"]


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_cndmask"):
            return "select"
        if op.startswith("v_mov_b64") or op.startswith("v_accvgpr"):
            return "mov64"
        if op.startswith("v_mov_b32"):
            return "mov32"
        if "f64" in op:
            return "f64 arith" if any(k in op for k in ("fma", "mul", "add", "mac")) else "f64 other"
        return "valu32"
    if op.startswith("s_"):
        return None if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_load", "s_buffer_load")) else "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return None


FUNC = re.compile(r"^\s*(?:template\s*<.*>\s*)?(?:(?:static|inline|__device__|__forceinline__|__host__|__global__|constexpr|LGH_HD|LGH_DEV)\s+)+[\w:<>\*&,\s]*?\b(\w+)\s*\(")
MARK = re.compile(r"^\s*// ---- (.*?)(?::|$)")


def regions(path):
    out = []
    pending_template = False
    try:
        lines = open(path, errors="replace").read().split("\n")
    except OSError:
        return out
    for i, l in enumerate(lines, 1):
        m = MARK.match(l)
        if m:
            out.append((i, "  " + m.group(1).strip()[:60]))
            continue
        m = FUNC.match(l)
        if m and not l.strip().endswith(";") and m.group(1) not in ("if", "for", "while", "switch", "return", "sizeof", "__launch_bounds__"):
            out.append((i, m.group(1)))
        elif "__launch_bounds__" in l:
            pass
    return out


def main():
    src, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "k.s")
        subprocess.run(BASE + extra + [src, "-o", s], check=True, stderr=subprocess.DEVNULL)
        text = open(s).read().split("\n")
    files, reg = {}, {}
    for l in text:
        m = re.match(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            d, f = m.group(2), m.group(3)
            p = f if os.path.isabs(f) else os.path.join(d if os.path.isabs(d) else os.path.join(os.getcwd(), d), f)
            files[int(m.group(1))] = p
    inside = False
    cur = ("?", 0)
    table = OrderedDict()
    for l in text:
        if not inside:
            if re.match(r"^[_A-Za-z0-9]*%s[_A-Za-z0-9]*:" % re.escape(name), l):
                inside = True
            continue
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", l)
        if not m or l.strip().startswith((";", ".")):
            continue
        c = classify(m.group(1))
        if c is None:
            continue
        f, line = cur
        if f not in reg:
            reg[f] = regions(f)
        label = os.path.basename(f)
        own = os.path.dirname(os.path.abspath(f)) == os.getcwd()
        if own:
            best = None
            for (ln, lab) in reg[f]:
                if ln <= line:
                    best = lab
                else:
                    break
            label = "%s: %s" % (os.path.basename(f), best.strip() if best else "(top)")
        table.setdefault(label, Counter())[c] += 1
    print("kernel *%s* of %s %s" % (name, src, " ".join(extra)))
    print("%-58s" % "source region" + "".join("%10s" % c for c in COLS) + "%10s" % "VALU")
    tot = Counter()
    for lab, cnt in sorted(table.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k not in ("salu", "lds", "vmem"))):
        valu = sum(v for k, v in cnt.items() if k not in ("salu", "lds", "vmem"))
        print("%-58s" % lab[:58] + "".join("%10d" % cnt[c] for c in COLS) + "%10d" % valu)
        tot.update(cnt)
    print("%-58s" % "total" + "".join("%10d" % tot[c] for c in COLS) + "%10d" % sum(v for k, v in tot.items() if k not in ("salu", "lds", "vmem")))


if __name__ == "__main__":
    main()

"""Developer tool: write profiles/r4_pmc_traffic.json (what bench.py quotes as roofline.traffic) from rocprofv3 --pmc
passes (FETCH_SIZE and WRITE_SIZE in separate runs, --kernel-trace only) summarised by tools/pmc_summary.py.
usage: python tools/update_pmc_traffic.py <workload>=<fetch_summary.txt>,<write_summary.txt> [...]
Per kernel: per-launch medians, the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request:
doubled), bytes = 2 * FETCH + WRITE.  The sha256 over the kernel sources (laghos_amd/csrc/*.h*) is recorded: bench.py reports
a figure only for that very build."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    out, cur = {}, None
    for l in open(path).read().splitlines():
        m = re.match(r"(\S.*?)\s+n=(\d+) dur_med=([0-9.]+) us", l)
        if m:
            cur = m.group(1).strip()
            out[cur] = {"n": int(m.group(2)), "us": float(m.group(3))}
            continue
        m = re.match(r"\s+(\w+)\s+med=([0-9.eE+-]+)", l)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def main(specs):
    import bench
    d = {"_comment": "rocprofv3 PMC, separate --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of ./laghos_amd/laghos -p 1 -m "
                     "data/cube01_hex.mesh -rs {4,5} -ok 3 -ot 2 -ms 3 -pa on MI355X (tools/gpu_pmc_traffic.sh); per-launch medians in KB. "
                     "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE doubled; WRITE_SIZE as is. The counters sit at the "
                     "L2's memory side: at c2 (working set ~ Infinity Cache) this is fabric traffic, at c3 (64^3) it is HBM traffic.",
         "kernel_sources_sha16": bench.kernel_sources_sha(), "workloads": {}}
    for spec in specs:
        name, files = spec.split("=")
        ff, wf = files.split(",")
        f, w = parse(ff), parse(wf)
        kern = {}
        for k in f:
            if k in w and "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w[k]:
                kern[k] = {"FETCH_SIZE_KB": f[k]["FETCH_SIZE"], "WRITE_SIZE_KB": w[k]["WRITE_SIZE"], "launches": f[k]["n"], "us_median": f[k]["us"],
                           "bytes_per_launch": (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0}
        k1 = [k for k in kern if "vcg_apply" in k]
        entry = {"kernels": kern}
        if k1:
            k1 = max(k1, key=lambda k: kern[k]["launches"])
            entry.update({"k1_kernel": k1, "k1_bytes_per_launch": kern[k1]["bytes_per_launch"]})
        d["workloads"][name] = entry
    pj = os.path.join(ROOT, "profiles", bench.PMC_FILE)
    json.dump(d, open(pj, "w"), indent=1)
    for n, e in d["workloads"].items():
        print(n, e.get("k1_kernel"), e.get("k1_bytes_per_launch"))


if __name__ == "__main__":
    main(sys.argv[1:])

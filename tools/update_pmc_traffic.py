"""Developer tool: refresh profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic) from the two
rocprofv3 --pmc passes of tools/gpu_profile_r2.sh.
usage: python tools/update_pmc_traffic.py <fetch_summary.txt> <write_summary.txt>
Takes the per-launch medians of K1 (vcg_apply_plane<4, 6, 13>) from the pmc_summary.py outputs, applies the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE doubled) and records the sha256 of the lgh_vcg.hip the counters were
collected on: bench.py only reports the figure for that very source."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def median_of(path, kernel, counter):
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if kernel in l:
            for m in lines[i + 1:i + 6]:
                if m.strip().startswith(counter):
                    return float(re.search(r"med=([0-9.eE+-]+)", m).group(1)), int(re.search(r"n=(\d+)", l).group(1))
    raise SystemExit(f"{kernel} / {counter} not found in {path}")


def main(fsum, wsum):
    k = "vcg_apply_plane<4, 6, 13"
    f, n = median_of(fsum, k, "FETCH_SIZE")
    w, _ = median_of(wsum, k, "WRITE_SIZE")
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(pj))
    d["FETCH_SIZE_KB_per_launch"] = f
    d["WRITE_SIZE_KB_per_launch"] = w
    d["mass_apply_cg_h1_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    d["launches"] = n
    d["kernel_source_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "laghos_amd", "csrc", "lgh_vcg.hip"), "rb").read()).hexdigest()[:16]
    json.dump(d, open(pj, "w"), indent=1)
    print("pmc_traffic.json:", d["mass_apply_cg_h1_bytes_per_launch"], "B per launch, source", d["kernel_source_sha16"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

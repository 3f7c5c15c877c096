"""Developer tool: per-kernel summary of rocprofv3 --pmc counter_collection.csv files.
usage: python tools/pmc_summary.py <dir-or-csv> [kernel-substring ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main(path, filt):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if filt and not any(s in n for s in filt):
                continue
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n in sorted(acc, key=lambda k: -sum(dur[k])):
        d = sorted(dur[n])
        print(f"{n[:70]}  n={len(acc[n][next(iter(acc[n]))])} dur_med={d[len(d)//2]:.1f} us")
        for c in sorted(acc[n]):
            v = sorted(acc[n][c])
            print(f"    {c:28s} med={v[len(v)//2]:.4e}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

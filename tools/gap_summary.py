"""Developer tool: idle time between consecutive kernels in a rocprofv3 kernel trace, grouped by (previous kernel -> next kernel).
usage: python tools/gap_summary.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("lgh::", "")[:28], r.get("Queue_Id", "0")) for r in csv.DictReader(open(f))]
rows.sort()
# busy union over all queues
t0, t1 = rows[0][0], max(r[1] for r in rows)
gaps = defaultdict(lambda: [0, 0.0])
cur_end, prev = rows[0][1], rows[0][2]
idle = 0
for s, e, n, q in rows[1:]:
    if s > cur_end:
        g = s - cur_end
        idle += g
        k = (prev, n)
        gaps[k][0] += 1
        gaps[k][1] += g
    if e > cur_end:
        cur_end, prev = e, n
print(f"span {1e-6*(t1-t0):.1f} ms, idle (no kernel on any queue) {1e-6*idle:.2f} ms = {100.0*idle/(t1-t0):.1f} %")
for k, (c, g) in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]:
    print(f"{k[0]:28s} -> {k[1]:28s} n={c:5d} total {1e-3*g:9.1f} us  mean {1e-3*g/c:7.2f} us")

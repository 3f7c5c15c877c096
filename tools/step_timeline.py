"""Developer tool: where the time of a steady-state RK step goes on the GPU's timeline - busy time per kernel and idle time per
(previous kernel -> next kernel) pair, per RK step, over the last third of a rocprofv3 kernel trace (bench.py's timed steps).
usage: python tools/step_timeline.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("lgh::", "")[:34]) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: between the 2nd and the last qrows update of the final third
q = [i for i, r in enumerate(rows) if r[2].startswith(("qrows_kernel", "qpoint_kernel<3, 6, 10, 5, 1, 6, 0"))]
q = q[len(q) * 2 // 3:]
q = q[: (len(q) // 4) * 4 + 1]           # whole RK4 steps (4 updates each; the dt estimate comes out of the 4th stage's data)
a, b = q[0], q[-1]
nsteps = (len(q) - 1) / 4.0
win = rows[a:b]
span = rows[b][0] - rows[a][0]
busy = defaultdict(lambda: [0, 0.0])
gaps = defaultdict(lambda: [0, 0.0])
cur_end, prev = win[0][1], win[0][2]
union = win[0][1] - win[0][0]
busy[win[0][2]][0] += 1; busy[win[0][2]][1] += win[0][1] - win[0][0]
for s, e, n in win[1:]:
    busy[n][0] += 1; busy[n][1] += e - s
    if s > cur_end:
        gaps[(prev, n)][0] += 1; gaps[(prev, n)][1] += s - cur_end
        union += e - s
    elif e > cur_end:
        union += e - cur_end
    if e > cur_end:
        cur_end, prev = e, n
print(f"{nsteps:.0f} RK4 steps, {1e-6 * span / nsteps:.3f} ms per step; some kernel running {1e-6 * union / nsteps:.3f} ms, idle {1e-6 * (span - union) / nsteps:.3f} ms per step")
print("busy per step (kernels overlap across the two streams):")
for n, (c, t) in sorted(busy.items(), key=lambda x: -x[1][1])[:22]:
    print(f"  {n:36s} {c / nsteps:7.1f} x {1e-3 * t / c:8.1f} us = {1e-3 * t / nsteps:8.1f} us")
print("idle per step by (previous -> next):")
for k, (c, g) in sorted(gaps.items(), key=lambda x: -x[1][1])[:22]:
    print(f"  {k[0]:34s} -> {k[1]:34s} {c / nsteps:6.1f} x {1e-3 * g / c:7.2f} us = {1e-3 * g / nsteps:7.1f} us")

"""Summary of LGH_VCG_TRACE=<file> (per-workgroup stamps of the last K1 launch of a solve, lgh_vcg.hip /
lgh_vcg_mfma.hip): wall-clock stamps in 10 ns ticks; column 5 of the matrix-core K1 packs
(cycles waiting for loads << 32) | cycles in the loop of wave 0."""
import sys
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
a = np.array([r for r in rows if r[1] > 0], dtype=np.int64)
if len(a) == 0:
    print("no records"); sys.exit(0)
t0 = a[:, 1].min()
us = lambda x: (x - t0) / 100.0
q = lambda v: "min/med/max %7.2f / %7.2f / %7.2f" % (np.min(v), np.median(v), np.max(v))
print(len(a), "workgroups")
print("start   ", q(us(a[:, 1])))
print("loop end", q(us(a[:, 2])))
print("end     ", q(us(a[:, 3])))
if a.shape[1] >= 14 and a[:, 13].min() > 0:
    print("enter   ", q(us(a[:, 13])), " (workgroup on its CU; `start` is the first pass: the difference is the prologue)")
    print("prologue", q((a[:, 1] - a[:, 13]) / 100.0))
if len(sys.argv) > 2 and sys.argv[2] == "mfma":
    wait, loop = a[:, 4] >> 32, a[:, 4] & 0xffffffff
    print("loop cycles (wave 0)   ", q(loop))
    print("wait cycles (wave 0)   ", q(wait), " = %.0f %% of the loop" % (100.0 * wait.sum() / loop.sum()))
    print("shader clock over the loop: %.2f GHz" % (np.median(loop / ((a[:, 2] - a[:, 1]) * 10.0))))
    if a.shape[1] >= 13 and a[:, 5:13].sum() > 0:
        names = ["load issue", "forward x,y", "transpose", "z", "transpose back", "backward y,x", "wait+convert", "stores"]
        tot = a[:, 5:13].sum(axis=1)
        for k, nm in enumerate(names):
            print("   %-16s median %8.0f cycles  %5.1f %%" % (nm, np.median(a[:, 5 + k]), 100.0 * a[:, 5 + k].sum() / tot.sum()))

#!/usr/bin/env python3
"""Summarise LGH_PCG_TRACE output: per-phase durations of the persistent solve kernel (lgh_pcg.hip).
Columns of the trace: block iter t0..t13 in 10 ns ticks:
 0 A start | 1 A loop end | 2 release start | 3 release done | 4 poll done | 5 acquire done | 6 A sync done
 7 B loop end | 8 release start | 9 release done | 10 poll done | 11 acquire done | 12 B sync done"""
import sys
import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
its = sorted(set(d[:, 1]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for it in its:
    if it < lo:
        continue
    r = d[d[:, 1] == it]
    if (r[:, 2 + 12] == 0).any():
        continue
    t = r[:, 2:].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    def st(x):
        return "%6.2f/%6.2f/%6.2f" % (x.min(), np.median(x), x.max())
    rows.append((it, t[:, 12].max() - t0,
                 st(t[:, 1] - t[:, 0]), st(t[:, 3] - t[:, 2]), st(t[:, 4] - t[:, 3]), st(t[:, 5] - t[:, 4]), st(t[:, 6] - t[:, 5]),
                 st(t[:, 7] - t[:, 6]), st(t[:, 9] - t[:, 8]), st(t[:, 10] - t[:, 9]), st(t[:, 11] - t[:, 10]), st(t[:, 12] - t[:, 11]),
                 t[:, 1].max() - t0, t[:, 6].max() - t0, t[:, 7].max() - t0))
print("per iteration (us; min/med/max over workgroups)")
print("it  total |   A loop            | A release          | A poll             | A acquire          | A fold             |"
      "   B loop            | B release          | B poll             | B acquire          | B fold   | lastAend lastAsync lastBend")
for r in rows:
    print("%2d %6.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %6.1f %6.1f %6.1f" % r)
# per-batch end times of phase A relative to the phase start (columns 14..21), old vs young half of the grid
if d.shape[1] >= 2 + 22:
    r = d[d[:, 1] == its[min(len(its) - 1, 10)]]
    t = r[:, 2:].astype(np.float64) * 0.01
    half = r[:, 0].max() // 2 + 1
    for name, m in (("first half of the grid", r[:, 0] < half), ("second half", r[:, 0] >= half)):
        rel = t[m][:, 14:22] - t[m][:, 0:1]
        rel[t[m][:, 14:22] == 0] = np.nan
        print("A batch end times (median us after phase start), %s:" % name, np.round(np.nanmedian(rel, axis=0), 1))
if rows:
    print("mean iteration: %.1f us over %d iterations" % (np.mean([r[1] for r in rows]), len(rows)))

#!/usr/bin/env python3
"""Summary of an LGH_Q_TRACE file (lgh_qrows.hpp): where a workgroup of the quadrature update spends its life, and how
busy a CU is.  Stamps are 100 MHz wall-clock ticks (10 ns)."""
import sys
import numpy as np

STAGES = ["issue loads", "wait loads (barrier)", "X stage", "Y stage", "Z stage (wave 0)", "point body (wave 0)",
          "wait other waves' bodies", "qz contraction", "qy contraction", "qx contraction + stores", "dt reduction / exit"]


def main(path):
    d = np.loadtxt(path, dtype=np.uint64)
    t = d[:, 1:13].astype(np.int64)
    hw = d[:, 13]
    ok = t[:, 0] > 0
    t, hw = t[ok], hw[ok]
    t0 = t[:, 0].min()
    t = (t - t0) * 10.0  # ns
    life = t[:, 11] - t[:, 0]
    print("workgroups %d, kernel span %.1f us, workgroup life mean %.2f us (p10 %.2f, p90 %.2f)" %
          (len(t), (t[:, 11].max()) * 1e-3, life.mean() * 1e-3, np.percentile(life, 10) * 1e-3, np.percentile(life, 90) * 1e-3))
    have_f = t[:, 7] > 0
    for k, name in enumerate(STAGES):
        a, b = k, k + 1
        if not have_f.all() and k in (6, 7, 8, 9):
            continue
        dur = t[:, b] - t[:, a]
        print("  %-28s mean %7.3f us  (%4.1f %% of a life)" % (name, dur.mean() * 1e-3, 100 * dur.mean() / life.mean()))
    # residency per CU: (xcc, se, sh, cu)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus = np.unique(key)
    print("distinct CUs seen: %d" % len(cus))
    span = t[:, 11].max()
    grid = np.arange(0, span, 100.0)  # 0.1 us
    tot = np.zeros((len(grid), 5))
    for c in cus[:64]:  # a sample of CUs
        m = key == c
        st, en = t[m, 0], t[m, 11]
        w0, w1 = t[m, 1], t[m, 2]  # waiting for the loads
        res = ((st[None, :] <= grid[:, None]) & (grid[:, None] < en[None, :])).sum(1)
        wait = ((w0[None, :] <= grid[:, None]) & (grid[:, None] < w1[None, :])).sum(1)
        comp = res - wait
        for k in range(5):
            tot[:, k] += (comp == k)
    tot /= min(len(cus), 64)
    inner = (grid > 0.05 * span) & (grid < 0.95 * span)
    print("workgroups of a CU that are past their load wait (sample of 64 CUs, middle 90 %% of the kernel):")
    for k in range(5):
        print("   %d computing: %5.1f %% of the time" % (k, 100 * tot[inner, k].mean()))


if __name__ == "__main__":
    main(sys.argv[1])

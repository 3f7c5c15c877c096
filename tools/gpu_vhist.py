import numpy as np, sys
sys.path.insert(0, "/root/repo")
from laghos_amd import host_lib
sim = host_lib.Sim(["-m", "data/cube01_hex.mesh", "-rs", 4, "-p", 1, "-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", 100, "-vs", 10**9, "-q"])
for n in (5, 20, 60):
    while sim.ti < n:
        sim.step()
    S = sim.state(); sz = sim.sizes(); H1V = 3 * sz["N"]
    v = np.abs(S[H1V:2 * H1V]).reshape(3, -1).max(axis=0)
    e = S[2 * H1V:]
    lg = np.where(v > 0, np.log10(np.maximum(v, 1e-320)), -400)
    edges = [-401, -300, -100, -60, -40, -30, -25, -20, -16, -12, -8, -4, 0, 10]
    h, _ = np.histogram(lg, bins=edges)
    print("ti", sim.ti, "t", sim.t, "vmax", v.max(), "frac per log10|v| bin", dict(zip(edges[1:], np.round(h / v.size, 4))))
    print("   e>0 frac", np.mean(e > 0), "e max", e.max())
sim.close()

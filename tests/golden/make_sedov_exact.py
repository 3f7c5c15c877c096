"""Emit tests/golden/sedov_exact.json from the REFERENCE's own exact Sedov solution.

Runs only in the build container: needs oracle/_ref/libsedov_ref.so, i.e. the reference's
sedov/sedov_sol.cpp compiled where it lies by `make -C oracle ref`.  The fixture holds
inputs and outputs only (parameter block, shock state, (rho, v, P) at sample radii);
omega = 0 throughout, the only value the reference driver uses (laghos.cpp:1012).

    python tests/golden/make_sedov_exact.py
"""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsedov_ref.so"))
    D, L = ctypes.c_double, ctypes.c_long
    P = ctypes.POINTER(D)

    def ptr(a):
        return a.ctypes.data_as(P)

    cases = []
    for dim, gamma, E, t in [(3, 1.4, 0.25, 0.6), (2, 1.4, 0.25, 0.8), (3, 5.0 / 3.0, 1.0, 1.0),
                             (2, 1.4, 1.0, 0.3), (1, 1.4, 0.25, 0.5), (3, 1.4, 2.0, 0.05)]:
        par = np.zeros(21)
        lib.ref_sedov_setup(dim, D(gamma), D(1.0), D(E), D(0.0), ptr(par))
        shock = np.zeros(6)
        r0 = np.zeros(1)
        out0 = [np.zeros(1) for _ in range(3)]
        lib.ref_sedov_eval(dim, D(gamma), D(1.0), D(E), D(0.0), D(t), L(1), ptr(r0), ptr(shock), *[ptr(x) for x in out0])
        r2 = shock[0]
        # inside (clustered towards the shock), exactly at, and beyond the shock
        r = np.concatenate([r2 * np.array([1e-9, 1e-4, 0.01, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.85, 0.9,
                                           0.95, 0.98, 0.99, 0.999, 0.999999]),
                            [r2], r2 * np.array([1.000001, 1.1, 2.0])])
        rho, v, p = (np.zeros(r.size) for _ in range(3))
        lib.ref_sedov_eval(dim, D(gamma), D(1.0), D(E), D(0.0), D(t), L(r.size), ptr(r), ptr(shock), ptr(rho), ptr(v), ptr(p))
        cases.append({"dim": dim, "gamma": gamma, "rho0": 1.0, "blast_energy": E, "omega": 0.0, "t": t,
                      "par": [float.hex(x) for x in par], "shock": [float.hex(x) for x in shock],
                      "r": [float.hex(x) for x in r], "rho": [float.hex(x) for x in rho],
                      "v": [float.hex(x) for x in v], "P": [float.hex(x) for x in p]})
    doc = {"source": "reference sedov/sedov_sol.cpp compiled into oracle/_ref/libsedov_ref.so (g++ -O2, x86-64)",
           "par_order": "dim gamma rho0 E omega a b c d e alpha0..alpha5 V0 Vv V2 Vs alpha",
           "shock_order": "r2 U rho1 rho2 v2 p2", "encoding": "float.hex", "cases": cases}
    with open(os.path.join(HERE, "sedov_exact.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()

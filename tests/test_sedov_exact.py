"""Exact Sedov solution (`laghos -err`): the oracle's restatement against the
reference's own sedov/sedov_sol.cpp — golden values emitted by the compiled reference
(tests/golden/sedov_exact.json, tests/golden/make_sedov_exact.py) and, where it has been built
(oracle/_ref, build container only), the compiled reference itself on seeded radii."""
import ctypes
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    with open(os.path.join(ROOT, "tests", "golden", "sedov_exact.json")) as f:
        doc = json.load(f)
    out = []
    for c in doc["cases"]:
        d = dict(c)
        for k in ("par", "shock", "r", "rho", "v", "P"):
            d[k] = np.array([float.fromhex(x) for x in c[k]])
        out.append(d)
    return out


CASES = _cases()


def close_to(got, ref, rel, shock_r=None, r=None):
    """element-wise |got-ref| <= rel*max(|ref|, tiny); points within 1e-12 of the shock radius are
    skipped (a 1-ulp difference in r2 flips them between the two states)."""
    got, ref = np.asarray(got), np.asarray(ref)
    keep = np.ones(ref.shape, dtype=bool)
    if shock_r is not None:
        keep = np.abs(np.asarray(r) - shock_r) > 1e-12 * shock_r
    return bool(np.all(np.abs(got - ref)[keep] <= rel * np.maximum(np.abs(ref)[keep], 1e-300)))


@pytest.mark.parametrize("case", CASES, ids=[f"dim{c['dim']}-g{c['gamma']:.3f}-t{c['t']}" for c in CASES])
def test_oracle_reproduces_reference_values(case):
    from oracle.sedov_error import SedovSol
    s = SedovSol(case["dim"], case["gamma"], case["rho0"], case["blast_energy"], case["omega"])
    assert close_to(s.par, case["par"], 2e-15)        # constants and the energy integral alpha
    s.set_time(case["t"])
    assert close_to(s.shock, case["shock"], 2e-15)
    rho, v, P = s.eval(case["r"])
    for got, key in ((rho, "rho"), (v, "v"), (P, "P")):
        assert close_to(got, case[key], 1e-12, shock_r=case["shock"][0], r=case["r"]), key


def test_golden_values_are_physical():
    """Rankine–Hugoniot jump at the shock and the self-similar scaling r2 ~ t^(2/(dim+2))."""
    from oracle.sedov_error import SedovSol
    for c in CASES:
        g = c["gamma"]
        r2, U, rho1, rho2, v2, p2 = c["shock"]
        assert abs(rho2 / rho1 - (g + 1) / (g - 1)) < 1e-14 * rho2
        assert abs(v2 - 2 * U / (g + 1)) < 1e-15
        s = SedovSol(c["dim"], g, c["rho0"], c["blast_energy"])
        s.set_time(2 * c["t"])
        assert abs(s.r2 / r2 - 2 ** (2.0 / (c["dim"] + 2))) < 1e-14


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsedov_ref.so")),
                    reason="oracle/_ref is built only where /root/reference exists")
def test_oracle_vs_compiled_reference():
    from oracle.sedov_error import SedovSol
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsedov_ref.so"))
    D = ctypes.c_double
    P = ctypes.POINTER(D)
    rng = np.random.default_rng(11)
    for dim, gamma, E, t in [(2, 1.4, 0.25, 0.8), (3, 1.4, 1.0, 0.6), (3, 1.2, 0.5, 0.2), (1, 5.0 / 3.0, 1.0, 1.0)]:
        s = SedovSol(dim, gamma, 1.0, E)
        s.set_time(t)
        r = np.sort(rng.uniform(0, 1.3 * s.r2, 4000))
        shock = np.zeros(6)
        out = [np.zeros(r.size) for _ in range(3)]
        ref.ref_sedov_eval(dim, D(gamma), D(1.0), D(E), D(0.0), D(t), ctypes.c_long(r.size), r.ctypes.data_as(P),
                           shock.ctypes.data_as(P), *[o.ctypes.data_as(P) for o in out])
        assert close_to(s.shock, shock, 2e-15)
        for got, want in zip(s.eval(r), out):
            assert close_to(got, want, 1e-12, shock_r=shock[0], r=r)


def test_density_projection_and_error_integral_oracle():
    """Oracle-level consistency of the -err post-processing: at t = 0 on the undeformed mesh the
    projected density is rho0 = 1 exactly, total mass is conserved by the projection on a deformed
    mesh, and the error integral of the constant state against a solution whose shock has not
    started (r2 -> 0) vanishes."""
    from helpers import deformed_state
    from oracle import sedov_error as se
    from oracle.driver import Hydro
    from oracle.fem import Problem
    prob = Problem(breaks=[np.linspace(0, 1, 5)] * 2, order_v=2, order_e=1, problem=1, blast_energy=0.25)
    h = Hydro(prob)
    S0 = h.S0.copy()
    rdj = np.array(h.rho0DetJ0w)
    rho = se.compute_density(prob, S0, rdj)
    assert np.max(np.abs(rho - 1.0)) < 1e-12
    S = deformed_state(prob)
    rho_d = se.compute_density(prob, S, rdj)
    _, detJ = se._zone_geometry(prob, S, prob.B, prob.G)
    psi = se._tensor([prob.Bl] * 2)
    mass = np.sum(prob.W[None, :] * detJ * (rho_d.reshape(prob.NE, -1) @ psi.T))
    assert abs(mass - rdj.sum()) < 1e-12 * rdj.sum()
    sol = se.SedovSol(2, 1.4, 1.0, 0.25)
    sol.set_time(1e-12)
    assert se.density_error(prob, S0, rho, sol, [0, 0, 0], se.err_order(2, 1)) < 1e-6
    h.close()


@pytest.mark.parametrize("case", CASES, ids=[f"dim{c['dim']}-g{c['gamma']:.3f}-t{c['t']}" for c in CASES])
def test_library_host_entries_reproduce_reference_values(case):
    """lgh_sedov_setup / _shock / _eval_point are the host-side scalar API of the reference's SedovSol
    class (constructor, SetTime, EvalSol): pure host code in liblaghos_hip.so, so they are checked
    here without a GPU (the array evaluation on the GPU and the error integral: test_gpu_sedov.py)."""
    from laghos_amd import context as C
    par = C.sedov_setup(case["dim"], case["gamma"], case["rho0"], case["blast_energy"])
    assert close_to(par, case["par"], 2e-15)
    assert close_to(C.sedov_shock(par, case["t"]), case["shock"], 2e-15)
    got = np.array([C.sedov_eval_point(par, case["t"], r) for r in case["r"]])
    for i, key in enumerate(("rho", "v", "P")):
        assert close_to(got[:, i], case[key], 1e-12, shock_r=case["shock"][0], r=case["r"]), key


def test_library_rejects_what_the_reference_cannot_do():
    """omega != 0: the reference's own set-up returns NaN or does not terminate (its header says
    "currently only supports uniform initial density"); the library refuses it."""
    from laghos_amd import _lib
    L = _lib.load()
    par = np.zeros(21)
    p = par.ctypes.data_as(_lib.c_dbl_p)
    assert L.lgh_sedov_setup(3, 1.4, 1.0, 1.0, 0.5, p) == 2          # LGH_ERR_UNSUPPORTED
    assert b"omega" in L.lgh_last_error()
    assert L.lgh_sedov_setup(4, 1.4, 1.0, 1.0, 0.0, p) == 1          # LGH_ERR_ARG
    assert L.lgh_sedov_setup(3, 1.0, 1.0, 1.0, 0.0, p) == 1
    assert L.lgh_sedov_setup(3, 1.4, 1.0, 1.0, 0.0, p) == 0 and par[20] > 0

"""`laghos -err` on the GPU: exact Sedov solution, density projection and the
error integral through the C ABI, against the reference-generated golden values and the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_sedov_exact import CASES, close_to

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def small_ctx():
    from helpers import make_gpu
    from oracle.fem import Problem
    prob = Problem(breaks=[np.linspace(0, 1, 3)] * 2, problem=1, blast_energy=0.25)
    h = make_gpu(prob)
    yield h
    h.close()


@pytest.mark.parametrize("case", CASES, ids=[f"dim{c['dim']}-g{c['gamma']:.3f}-t{c['t']}" for c in CASES])
def test_exact_solution_vs_reference_values(small_ctx, case):
    """lgh_sedov_setup / _shock / _eval_point (host, scalar API of SedovSol) and lgh_sedov_eval
    (GPU) against the values of the compiled reference.  The device pow differs from glibc's in
    the last bits and the bisection amplifies that slightly: 1e-11."""
    import torch
    from laghos_amd import context as C
    par = C.sedov_setup(case["dim"], case["gamma"], case["rho0"], case["blast_energy"])
    assert close_to(par, case["par"], 2e-15)
    assert close_to(C.sedov_shock(par, case["t"]), case["shock"], 2e-15)
    host = np.array([C.sedov_eval_point(par, case["t"], r) for r in case["r"]])
    ctx = small_ctx.ctx
    r = ctx.to_dev(case["r"])
    out = [ctx.zeros(r.numel()) for _ in range(3)]
    ctx.sedov_eval(par, case["t"], r, *out)
    torch.cuda.synchronize()
    for i, key in enumerate(("rho", "v", "P")):
        assert close_to(host[:, i], case[key], 1e-12, shock_r=case["shock"][0], r=case["r"]), key
        assert close_to(out[i].cpu().numpy(), case[key], 1e-11, shock_r=case["shock"][0], r=case["r"]), key


def test_exact_solution_many_radii_vs_oracle(small_ctx):
    import torch
    from laghos_amd import context as C
    from oracle.sedov_error import SedovSol
    rng = np.random.default_rng(5)
    for dim, gamma, E, t in [(3, 1.4, 0.25, 0.6), (2, 1.4, 0.25, 0.8)]:
        sol = SedovSol(dim, gamma, 1.0, E)
        sol.set_time(t)
        r_h = rng.uniform(0, 1.2 * sol.r2, 200000)
        par = C.sedov_setup(dim, gamma, 1.0, E)
        ctx = small_ctx.ctx
        r = ctx.to_dev(r_h)
        out = [ctx.zeros(r.numel()) for _ in range(3)]
        ctx.sedov_eval(par, t, r, *out)
        torch.cuda.synchronize()
        for got, want in zip(out, sol.eval(r_h)):
            assert close_to(got.cpu().numpy(), want, 1e-11, shock_r=sol.r2, r=r_h)


@pytest.mark.parametrize("dim,n,ok,ot", [(2, 8, 2, 1), (3, 4, 2, 1), (2, 4, 3, 2), (3, 3, 3, 2)])
def test_density_and_error_vs_oracle(dim, n, ok, ot):
    """ComputeDensity and the error integral on a deformed state and after a short Sedov run:
    HIP kernels vs the numpy restatement (same formulas, different summation order): 1e-10."""
    from helpers import deformed_state, make_gpu, rel_err
    from laghos_amd import context as C
    from laghos_amd.hydro import TimeLoop
    from oracle import sedov_error as se
    from oracle.fem import Problem, bernstein_table, gauss_legendre, lagrange_tables
    prob = Problem(breaks=[np.linspace(0, 1, n + 1)] * dim, order_v=ok, order_e=ot, problem=1, blast_energy=0.25)
    h = make_gpu(prob)
    rdj = np.asarray(h.ctx.rho0DetJ0w)
    t = 0.05
    loop = TimeLoop(h, t_final=t)
    while loop.step():
        pass
    eo = se.err_order(ok, ot)
    pts, wts = gauss_legendre(eo // 2 + 1)
    B, G = lagrange_tables(prob.gll, pts)
    Bl = bernstein_table(ot, pts)
    sol = se.SedovSol(dim, 1.4, 1.0, 0.25)
    sol.set_time(t)
    par = C.sedov_setup(dim, 1.4, 1.0, 0.25)
    for S in (loop.S, h.ctx.to_dev(deformed_state(prob))):
        S_h = S.cpu().numpy()
        rho = h.compute_density(S)
        rho_o = se.compute_density(prob, S_h, rdj)
        assert rel_err(rho.cpu().numpy(), rho_o) < 1e-10
        err = h.sedov_density_error(S, rho, par, t, [0.0, 0.0, 0.0], wts, B, G, Bl)
        err_o = se.density_error(prob, S_h, rho_o, sol, [0, 0, 0], eo)
        assert abs(err - err_o) < 1e-10 * err_o, (err, err_o)
    h.close()


@pytest.mark.parametrize("args,ok,ot", [(["-dim", "2", "-nx", "8", "-ny", "8", "-rs", "0", "-tf", "0.3", "-E0", "0.25"], 2, 1),
                                        (["-dim", "3", "-nx", "4", "-ny", "4", "-nz", "4", "-rs", "0", "-tf", "0.1", "-E0", "0.25",
                                          "-ok", "3", "-ot", "2"], 3, 2)])
def test_cpp_driver_err_option(args, ok, ot):
    """`laghos -p 1 -err` (default mesh): the printed "Density L2 error" against the oracle's value for
    the oracle's own final state of the same run (states agree to CG tolerance -> 1e-6), and
    through the library object against the value computed from the driver's own state."""
    from laghos_amd import host_lib
    from oracle import sedov_error as se
    from oracle.driver import Hydro, run as orun
    from oracle.fem import Problem
    dim = int(args[1])
    n = int(args[3])
    tf = float(args[args.index("-tf") + 1])
    exe = os.path.join(ROOT, "laghos_amd", "laghos")
    p = subprocess.run([exe, "-p", "1", "-pa", "-err"] + args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    line = next(l for l in p.stdout.splitlines() if l.startswith("Density L2 error:"))
    printed = float(line.split(":")[1])
    prob = Problem(breaks=[np.linspace(0, 1, n + 1)] * dim, order_v=ok, order_e=ot, problem=1, blast_energy=0.25)
    ho = Hydro(prob)
    out = orun(prob, t_final=tf, hydro=ho)
    sol = se.SedovSol(dim, 1.4, 1.0, 0.25)
    sol.set_time(tf)
    rho_o = se.compute_density(prob, out["S"], np.array(ho.rho0DetJ0w))
    err_o = se.density_error(prob, out["S"], rho_o, sol, [0, 0, 0], se.err_order(ok, ot))
    ho.close()
    assert abs(printed - err_o) < 1e-5 * err_o, (printed, err_o)
    sim = host_lib.Sim(["-p", 1, "-pa"] + args)
    while sim.step() == 1:
        pass
    err = sim.sedov_error()
    assert abs(err - printed) <= 1e-6 * printed          # printed with 7 digits
    S = sim.state()
    ho = Hydro(prob)
    rho_s = se.compute_density(prob, S, np.array(ho.rho0DetJ0w))
    ho.close()
    assert abs(err - se.density_error(prob, S, rho_s, sol, [0, 0, 0], se.err_order(ok, ot))) < 1e-9 * err
    sim.close()


def test_cpp_driver_err_option_guards():
    """-err needs problem 1 on the default mesh, and a shock that is still inside the box
    (laghos.cpp:303-309, :1018-1025)."""
    exe = os.path.join(ROOT, "laghos_amd", "laghos")
    for extra in (["-p", "0", "-dim", "2"], ["-p", "1", "-m", "data/square01_quad.mesh"]):
        p = subprocess.run([exe, "-pa", "-err", "-ms", "1"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0
    p = subprocess.run([exe, "-p", "1", "-pa", "-err", "-dim", "2", "-nx", "4", "-ny", "4", "-rs", "0", "-ms", "2", "-tf", "5.0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "reflections" in (p.stdout + p.stderr)


@pytest.mark.parametrize("nranks", [2, 8])
def test_err_option_on_emulated_ranks(nranks):
    """-err on several ranks (loopback communicator, see test_multi_rank_run_on_one_gpu): every rank
    integrates its own zones, the squared error is all-reduced; must match the single-rank value."""
    import threading
    from laghos_amd import host_lib
    args = ["-p", 1, "-dim", 3, "-nx", 8, "-ny", 8, "-nz", 8, "-rs", 0, "-ok", 2, "-ot", 1, "-pa", "-E0", 0.25, "-tf", 0.05, "-q"]
    ref = host_lib.Sim(args)
    while ref.step() == 1:
        pass
    want = ref.sedov_error()
    ref.close()
    assert want > 0
    cid = (b"LGHLOCAL" + os.urandom(16).hex().encode()).ljust(128, b"\0")
    out, err = {}, {}

    def rank_main(rank):
        try:
            sim = host_lib.Sim(args, nranks=nranks, rank=rank, nccl_id=cid)
            while sim.step() == 1:
                pass
            out[rank] = sim.sedov_error()
            sim.close()
        except Exception as ex:  # noqa: BLE001 - reported below
            err[rank] = repr(ex)

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a rank did not finish (collective mismatch?)"
    assert not err, err
    for r in range(nranks):
        assert abs(out[r] - want) <= 1e-9 * want, (r, out[r], want)

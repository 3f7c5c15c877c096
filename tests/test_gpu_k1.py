"""Kernel-level parity of K1, the mass-apply kernel of the lockstep velocity solve (the kernel an RK step spends most
of its time in): ONE launch through the C ABI (lgh_test_vcg_k1), in every form the solve can dispatch
(column / plane / two-lane plane / slab / Kronecker) and with both forms of the mass data, against the oracle's
element-level mass apply (MassPAOperator::Mult before the E->L sum, /root/reference/laghos_assembly.cpp:117-121,
on d = r/diag + beta d_old as upstream CGSolver::Mult forms it).  What is compared is what K1 hands to K2: the
E-vector of A_e d_e for the three components and (d, A d).  Tolerance 1e-13 relative to the largest entry (the
operator tolerance of tests/test_gpu_kernels.py; the compact mass data moves the operator by <= 1e-12 of itself,
DESIGN.md §4, so those cases are held to 2e-12)."""
import os

import numpy as np
import pytest

from helpers import make_gpu, make_oracle, rel_err, seeded

pytestmark = pytest.mark.gpu

# (id, mesh, rs, order_v, order_e, LGH_VCG_VARIANT or None, expected lgh_k1_form)
K1_CASES = [
    ("Q3Q2-64-default", "cube01_hex", 1, 3, 2, None, "plane"),
    ("Q3Q2-64-column", "cube01_hex", 1, 3, 2, "0", "column"),
    ("Q3Q2-64-plane", "cube01_hex", 1, 3, 2, "2", "plane"),
    ("Q3Q2-64-slab", "cube01_hex", 1, 3, 2, "4", "slab"),
    ("Q3Q2-16-slab", "box01_hex", 0, 3, 2, "4", "slab"),      # ragged last set (sets of 5 elements)
    ("Q3Q2-16-plane", "box01_hex", 0, 3, 2, "2", "plane"),    # ragged last batch (batches of 13)
    ("Q3Q2-512-slab", "cube01_hex", 2, 3, 2, "4", "slab"),    # more sets than wavefronts of one workgroup
    ("Q3Q2-64-kron", "cube01_hex", 1, 3, 2, "5", "kron"),
    ("Q3Q2-16-kron", "box01_hex", 0, 3, 2, "5", "kron"),
    ("Q2Q1-64-kron", "cube01_hex", 1, 2, 1, "5", "kron"),
    ("Q1Q0-64-kron", "cube01_hex", 1, 1, 0, "5", "kron"),
    ("Q4Q3-16-kron", "box01_hex", 0, 4, 3, "5", "kron"),
    ("Q2Q1-64-default", "cube01_hex", 1, 2, 1, None, "plane"),
    ("Q2Q1-64-column", "cube01_hex", 1, 2, 1, "0", "column"),
    ("Q1Q0-64-default", "cube01_hex", 1, 1, 0, None, "plane"),
    ("Q4Q3-16-default", "box01_hex", 0, 4, 3, None, "plane"),
    ("Q4Q3-16-twolane", "box01_hex", 0, 4, 3, "1", "plane"),
    ("Q4Q3-16-column", "box01_hex", 0, 4, 3, "0", "column"),
    ("Q5Q4-16-default", "box01_hex", 0, 5, 4, None, "plane"),
    ("Q5Q4-16-column", "box01_hex", 0, 5, 4, "0", "column"),
]


def _oracle_k1(prob, o, r, d_old, beta, first):
    """d = r/diag + beta d_old per component, then A_e d_e and (d, A d) with the oracle's element mass apply."""
    from oracle.driver import _dp
    N, NE, ND = prob.N, prob.NE, prob.ND
    dinv = 1.0 / np.asarray(o.diagV)
    hmap = np.asarray(prob.h1map).reshape(NE, ND)
    yE = np.empty((3, NE * ND))
    den = np.zeros(3)
    for c in range(3):
        d = r[c * N:(c + 1) * N] * dinv
        if not first:
            d = d + beta[c] * d_old[c * N:(c + 1) * N]
        xE = np.ascontiguousarray(d[hmap].reshape(-1))
        o.L.lgo_mass_apply_E(o.h, 0, _dp(xE), _dp(yE[c]))
        den[c] = float(np.dot(xE, yE[c]))
    return yE, den


def _run_case(prob, monkeypatch, variant, form, rank1, first):
    monkeypatch.delenv("LGH_VCG_VARIANT", raising=False)
    monkeypatch.delenv("LGH_MASS_RANK1", raising=False)
    if variant is not None:
        monkeypatch.setenv("LGH_VCG_VARIANT", variant)
    if not rank1:
        monkeypatch.setenv("LGH_MASS_RANK1", "0")
    g, o = make_gpu(prob), make_oracle(prob)
    try:
        # (default dispatch with compact mass data on a tensor-product rule: the Kronecker form; the slab form applies it inside)
        kron_ok = rank1 and os.environ.get("LGH_MASS_KRON") != "0"
        if variant == "5":
            expect = "kron" if kron_ok else "plane"  # (no compact data: nothing to take the Kronecker product of)
        else:
            expect = "kron" if (variant is None and kron_ok and form == "plane" and prob.D1D >= 5) else form
        assert g.ctx.k1_form() == expect
        N = prob.N
        r = seeded(3 * N, 101)
        d_old = seeded(3 * N, 102)
        dinv = 1.0 / np.asarray(o.diagV)
        rz = np.array([float(np.dot(r[c * N:(c + 1) * N] ** 2, dinv)) for c in range(3)])
        rz_prev = rz * np.array([1.7, 0.6, 1.1])  # beta = rz / rz_prev
        beta = rz / rz_prev
        yE_o, den_o = _oracle_k1(prob, o, r, d_old, beta, first)
        yE, den = g.ctx.test_vcg_k1(g.ctx.to_dev(r), None if first else g.ctx.to_dev(d_old), rz, rz_prev, first)
        assert g.ctx.mass_data_form() == ("rank1" if rank1 else "stored")
        yE = yE.cpu().numpy()
        # Merged E-vector layout of the slab K1 (round 5): where the zones of a set are x-neighbours the two contributions to
        # a node of a shared x-face leave K1 as ONE value - reported in the left zone's dx = 3 entry, 0.0 in the right zone's
        # dx = 0 entry (lgh_test_vcg_merged_faces marks those).  The oracle's element contributions are summed the same way.
        mask, n_merged = g.ctx.test_vcg_merged_faces()
        merged = os.environ.get("LGH_SLAB_MERGE") != "0" and expect == "slab"
        assert merged or n_merged == 0  # (whether a mesh HAS x-chains of five zones depends on its rows: the callers below say where it must)
        if n_merged:
            idx = np.nonzero(mask)[0]
            assert np.all(idx % prob.D1D == 0) and np.all(idx >= prob.ND)  # dx = 0 entries of a zone with a left neighbour
            yE_o = yE_o.copy()
            for c in range(3):
                yE_o[c, idx - prob.ND + (prob.D1D - 1)] += yE_o[c, idx]    # (same dy, dz; dx = 3 of the zone before)
                yE_o[c, idx] = 0.0
    finally:
        g.close()
        o.close()
    tol = 2e-12 if rank1 else 1e-13
    for c in range(3):
        assert rel_err(yE[c], yE_o[c]) < tol, (c, "E-vector")
        assert abs(den[c] - den_o[c]) <= tol * abs(den_o[c]), (c, "den", den[c], den_o[c])
    return n_merged


@pytest.mark.parametrize("first", [True, False], ids=["first", "later"])
@pytest.mark.parametrize("rank1", [True, False], ids=["compact", "stored"])
@pytest.mark.parametrize("case", K1_CASES, ids=[c[0] for c in K1_CASES])
def test_k1_one_launch_vs_oracle(case, rank1, first, monkeypatch):
    from oracle.fem import Problem
    _, mesh, rs, ok, ot, variant, form = case
    prob = Problem(mesh=mesh, rs=rs, order_v=ok, order_e=ot, problem=1)
    n_merged = _run_case(prob, monkeypatch, variant, form, rank1, first)
    if case[0] == "Q3Q2-512-slab":  # rows of 8 zones: every other set of five is an x-chain (4 faces x 16 nodes merged in each)
        assert n_merged == 51 * 64


@pytest.mark.parametrize("first", [True, False], ids=["first", "later"])
@pytest.mark.parametrize("case", [c for c in K1_CASES if c[6] == "slab"], ids=[c[0] for c in K1_CASES if c[6] == "slab"])
def test_k1_slab_through_the_quadrature_points(case, first, monkeypatch):
    """With compact mass data on a tensor-product rule the slab K1 applies the element matrix as a Kronecker product of
    1-D mass tiles (no quadrature-point values; the default in the cases above).  LGH_MASS_KRON=0 keeps the
    contraction through the quadrature points with the compact data: same operator, same tolerance."""
    from oracle.fem import Problem
    _, mesh, rs, ok, ot, variant, form = case
    monkeypatch.setenv("LGH_MASS_KRON", "0")
    prob = Problem(mesh=mesh, rs=rs, order_v=ok, order_e=ot, problem=1)
    _run_case(prob, monkeypatch, variant, form, True, first)


@pytest.mark.parametrize("rank1", [True, False], ids=["compact", "stored"])
def test_k1_default_dispatch_at_bench_size(rank1, monkeypatch):
    """Config 2's mesh (32^3 zones, Q3Q2): the kernel bench.py's headline is dominated by, as dispatched by default
    (slab form from 20 000 zones), one launch against the oracle's mass apply on all 32 768 elements."""
    from oracle.fem import Problem
    prob = Problem(mesh="cube01_hex", rs=4, order_v=3, order_e=2, problem=1)
    n_merged = _run_case(prob, monkeypatch, None, "slab", rank1, False)
    assert n_merged > 4000 * 64  # (rows of 32 zones: five or six of every 6.4 sets are x-chains)


def test_k1_static_schedule_at_64_cubed(monkeypatch):
    """64^3 zones (configs 3 / 4 on one GPU): the slab K1 runs its STATIC interleaved schedule there (51 passes per
    wavefront; the workgroup queues only up to 16) and its merged E-vector has whole x-chains and sets that straddle
    rows of zones.  An oracle context of that size holds ~10 GB of quadrature data the comparison does not need, so this
    case is a size-independent property instead: one launch of the slab form and one of the plane form - the form the
    cases above hold to the oracle at every size the oracle is cheap at - on the same vectors must hand K2 the same
    assembled A d (E -> L sum through the element -> node map) and the same (d, A d), to the 2e-12 of the compact mass
    data.  Entries K1 has merged are reported once (lgh_test_vcg_merged_faces): the assembled sum does not care."""
    from oracle.fem import Problem
    prob = Problem(mesh="cube01_hex", rs=5, order_v=3, order_e=2, problem=1)
    N, NE, ND = prob.N, prob.NE, prob.ND
    hmap = np.asarray(prob.h1map).reshape(-1)
    r, d_old = seeded(3 * N, 111), seeded(3 * N, 112)
    out = {}
    for variant, form in (("4", "slab"), ("2", "plane")):
        monkeypatch.setenv("LGH_VCG_VARIANT", variant)
        g = make_gpu(prob)
        try:
            assert g.ctx.k1_form() == form
            dinv = 1.0 / np.asarray(g.ctx.mass_diag)
            rz = np.array([float(np.dot(r[c * N:(c + 1) * N] ** 2, dinv)) for c in range(3)])
            yE, den = g.ctx.test_vcg_k1(g.ctx.to_dev(r), g.ctx.to_dev(d_old), rz, rz * np.array([1.7, 0.6, 1.1]), False)
            yE = yE.cpu().numpy()
            _, n_merged = g.ctx.test_vcg_merged_faces()
            assert (n_merged > 0) == (form == "slab")
            Ad = np.zeros((3, N))
            for c in range(3):
                Ad[c] = np.bincount(hmap, weights=yE[c], minlength=N)
            out[form] = (Ad, den)
        finally:
            g.close()
    for c in range(3):
        assert rel_err(out["slab"][0][c], out["plane"][0][c]) < 2e-12, c
        assert abs(out["slab"][1][c] - out["plane"][1][c]) <= 2e-12 * abs(out["plane"][1][c]), c

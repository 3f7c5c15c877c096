"""Kernel-level parity of K1, the mass-apply kernel of the lockstep velocity solve (the kernel an RK step spends most
of its time in): ONE launch through the C ABI (lgh_test_vcg_k1), in every form the solve can dispatch
(column / plane / two-lane plane / matrix-core / slab / Kronecker) and with both forms of the mass data, against the oracle's
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
    ("Q3Q2-64-mfma", "cube01_hex", 1, 3, 2, "3", "mfma"),
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
    finally:
        g.close()
        o.close()
    tol = 2e-12 if rank1 else 1e-13
    for c in range(3):
        assert rel_err(yE[c], yE_o[c]) < tol, (c, "E-vector")
        assert abs(den[c] - den_o[c]) <= tol * abs(den_o[c]), (c, "den", den[c], den_o[c])


@pytest.mark.parametrize("first", [True, False], ids=["first", "later"])
@pytest.mark.parametrize("rank1", [True, False], ids=["compact", "stored"])
@pytest.mark.parametrize("case", K1_CASES, ids=[c[0] for c in K1_CASES])
def test_k1_one_launch_vs_oracle(case, rank1, first, monkeypatch):
    from oracle.fem import Problem
    _, mesh, rs, ok, ot, variant, form = case
    prob = Problem(mesh=mesh, rs=rs, order_v=ok, order_e=ot, problem=1)
    _run_case(prob, monkeypatch, variant, form, rank1, first)


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
    _run_case(prob, monkeypatch, None, "slab", rank1, False)

"""Kernel-level parity of K2, the node kernel of the lockstep velocity solve and the kernel with the largest share of an RK
step: ONE launch through the C ABI (lgh_test_vcg_k2) in every form the one-rank solve can dispatch - the bounded-grid
kernel with the ticketed fold of (r, z), the same with the exact accumulators (slab K1), one or two nodes per thread,
and the round-1 kernel - against a numpy restatement of what one iteration of upstream's CGSolver::Mult does between
two operator applications for each of the three component solves of /root/reference/laghos_solver.cpp:383-392:
E -> L sum of the element contributions (H1 restriction transposed, laghos_assembly.cpp:121), essential rows,
alpha = (r, z) / (d, A d), r -= alpha A d, d = r_old / diag + beta d_old (the direction K1 of the same iteration
applied the operator to, stored for the next one), x += alpha d - deferred to every second iteration in the bounded-grid
kernel, where x then takes the terms of two iterations - and (r, z) of the new residual with z = r / diag.
Tolerance 1e-13 of the largest entry."""
import numpy as np
import pytest

from helpers import make_gpu, make_oracle, rel_err, seeded

pytestmark = pytest.mark.gpu

# (id, mesh, rs, order_v, order_e, environment, bounded-grid kernel expected)
K2_CASES = [
    ("Q3Q2-64-default", "cube01_hex", 1, 3, 2, {}, True),
    ("Q3Q2-128-slab-exact-rz", "box01_hex", 1, 3, 2, {"LGH_VCG_VARIANT": "4"}, True),
    ("Q3Q2-128-slab-ticketed-rz", "box01_hex", 1, 3, 2, {"LGH_VCG_VARIANT": "4", "LGH_RZ_LIMBS": "0"}, True),
    ("Q3Q2-512-slab-exact-rz", "cube01_hex", 2, 3, 2, {"LGH_VCG_VARIANT": "4"}, True),
    ("Q3Q2-64-two-nodes-per-thread", "cube01_hex", 1, 3, 2, {"LGH_K2_U": "2"}, True),
    ("Q3Q2-64-all-slots", "cube01_hex", 1, 3, 2, {"LGH_K2_SKIP": "0"}, True),
    ("Q3Q2-64-round-1-kernel", "cube01_hex", 1, 3, 2, {"LGH_K2P": "0"}, False),
    ("Q2Q1-64-default", "cube01_hex", 1, 2, 1, {}, True),
    ("Q4Q3-16-default", "box01_hex", 0, 4, 3, {}, True),
    ("Q5Q4-16-default", "box01_hex", 0, 5, 4, {}, True),
]


def _numpy_k2(prob, dinv, it, yE, r, d_old, x, den, rz, rz_prev, alpha_prev, deferred_x):
    N, NE, ND = prob.N, prob.NE, prob.ND
    hmap = np.asarray(prob.h1map).reshape(-1)
    r_n, d_n, x_n, rz_n = np.empty_like(r), np.empty_like(d_old), x.copy(), np.zeros(3)
    for c in range(3):
        s = slice(c * N, (c + 1) * N)
        Ad = np.zeros(N)
        np.add.at(Ad, hmap, yE[c])  # the E -> L sum
        ess = np.asarray(prob.ess[c], dtype=np.int64)
        if len(ess):
            Ad[ess] = 0.0  # (r and d vanish there)
        alpha, beta = rz[c] / den[c], rz[c] / rz_prev[c]
        z_old = r[s] * dinv
        d_n[s] = z_old if it == 1 else z_old + beta * d_old[s]
        r_n[s] = r[s] - alpha * Ad
        if deferred_x:
            if it % 2 == 0:
                x_n[s] = (x[s] if it > 2 else 0.0) + alpha * d_n[s] + alpha_prev[c] * d_old[s]
        else:
            x_n[s] = x[s] + alpha * d_n[s]
        rz_n[c] = float(np.dot(r_n[s] ** 2, dinv))
    return r_n, d_n, x_n, rz_n


@pytest.mark.parametrize("it", [1, 2, 3, 4], ids=lambda i: f"it{i}")
@pytest.mark.parametrize("case", K2_CASES, ids=[c[0] for c in K2_CASES])
def test_k2_one_launch_vs_numpy(case, it, monkeypatch):
    from oracle.fem import Problem
    _, mesh, rs, ok, ot, env, bounded = case
    for k in ("LGH_VCG_VARIANT", "LGH_RZ_LIMBS", "LGH_K2_U", "LGH_K2_SKIP", "LGH_K2P"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    prob = Problem(mesh=mesh, rs=rs, order_v=ok, order_e=ot, problem=1)
    N, NE, ND = prob.N, prob.NE, prob.ND
    g, o = make_gpu(prob), make_oracle(prob)
    try:
        if "LGH_VCG_VARIANT" in env:
            assert g.ctx.k1_form() == "slab"
        dinv = 1.0 / np.asarray(o.diagV)
        r, d_old, x = seeded(3 * N, 201), seeded(3 * N, 202), seeded(3 * N, 203)
        for c in range(3):  # (the solve keeps r, d and x zero at the essential dofs of a component)
            ess = np.asarray(prob.ess[c], dtype=np.int64)
            if len(ess):
                r[c * N + ess] = 0.0
                d_old[c * N + ess] = 0.0
                x[c * N + ess] = 0.0
        yE = seeded(3 * NE * ND, 204).reshape(3, NE * ND)
        rz = np.array([float(np.dot(r[c * N:(c + 1) * N] ** 2, dinv)) for c in range(3)])
        rz_prev = rz * np.array([1.7, 0.6, 1.1])
        den = rz * np.array([2.3, 0.9, 1.4])
        alpha_prev = np.array([0.31, 1.9, 0.77])
        rd, dd, xd = g.ctx.to_dev(r), g.ctx.to_dev(d_old), g.ctx.to_dev(x)
        rz_g, deferred = g.ctx.test_vcg_k2(it, g.ctx.to_dev(np.ascontiguousarray(yE.reshape(-1))), rd, dd, xd, den, rz, rz_prev, alpha_prev)
        assert deferred == bounded
        r_o, d_o, x_o, rz_o = _numpy_k2(prob, dinv, it, yE, r, d_old, x, den, rz, rz_prev, alpha_prev, deferred)
        r_g, d_g, x_g = rd.cpu().numpy(), dd.cpu().numpy(), xd.cpu().numpy()
    finally:
        g.close()
        o.close()
    tol = 1e-13
    assert rel_err(r_g, r_o) < tol, "r"
    assert rel_err(d_g, d_o) < tol, "d"
    assert rel_err(x_g, x_o) < tol, "x"
    for c in range(3):
        assert abs(rz_g[c] - rz_o[c]) <= tol * abs(rz_o[c]), (c, "(r, z)", rz_g[c], rz_o[c])

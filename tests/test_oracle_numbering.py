"""CPU: the oracle knows a mesh by its element -> node map only.  On a randomly renumbered, randomly reordered copy of a
problem (tests/helpers.py::PermutedProblem - what tests/test_gpu_general_numbering.py runs the HIP path on) it must return
the structured problem's dS/dt and time-step estimate mapped through the permutation, to round-off."""
import numpy as np
import pytest

from helpers import PermutedProblem, deformed_state, make_oracle, rel_err


@pytest.mark.parametrize("kw", [dict(mesh="cube01_hex", rs=1, order_v=3, order_e=2, problem=1),
                                dict(mesh="square01_quad", rs=2, order_v=2, order_e=1, problem=1)], ids=["3D-Q3Q2", "2D-Q2Q1"])
def test_oracle_on_a_permuted_mesh(kw):
    from oracle.fem import Problem
    base = Problem(**kw)
    perm = PermutedProblem(base, seed=5)
    assert sorted(np.unique(perm.h1map).tolist()) == list(range(base.N))
    S_b = deformed_state(base, seed=29)
    S_p = perm.state(S_b)
    res = {}
    for name, prob, S in (("base", base, S_b), ("perm", perm, S_p)):
        o = make_oracle(prob, cg_tol=1e-13)
        try:
            dS = np.empty_like(S)
            o.qdata_is_current = False
            o.reset_time_step_estimate()
            o.mult(S, dS)
            res[name] = (dS, o.get_time_step_estimate(S))
        finally:
            o.close()
    assert rel_err(res["perm"][0], perm.state(res["base"][0])) < 1e-10
    assert abs(res["perm"][1] - res["base"][1]) <= 1e-13 * res["base"][1]

"""Shared helpers for the parity tests: oracle-side and GPU-side operators built
from the same oracle.fem.Problem and the same seeded inputs."""
import numpy as np


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / den)


def seeded(n, seed):
    """uniform(-0.5, 0.5) from a fixed seed (SURVEY §8d synthetic inputs)"""
    return np.random.default_rng(seed).uniform(-0.5, 0.5, size=n)


def deformed_state(prob, seed=3, amp=0.02, e_amp=0.5):
    """A smooth random perturbation of the initial state: distorted mesh, non-zero
    velocity, positive energies -> exercises every branch of the qpoint body."""
    S, rho_l2, gamma, rho0_q = prob.initial_state()
    rng = np.random.default_rng(seed)
    H1V = prob.H1V
    h = min(np.min(np.diff(b)) for b in prob.breaks) / prob.order_v
    S = S.copy()
    S[:H1V] += amp * h * rng.uniform(-1, 1, H1V)
    S[H1V:2 * H1V] = rng.uniform(-1, 1, H1V)
    S[2 * H1V:] = 1.0 + e_amp * rng.uniform(-1, 1, prob.L2V)
    return S


def make_gpu(prob, **kw):
    from laghos_amd.hydro import HydroOperator
    return HydroOperator(prob, **kw)


def make_oracle(prob, **kw):
    from oracle.driver import Hydro
    return Hydro(prob, **kw)

"""ORACLE (test infrastructure only) — the `-err` post-processing of the reference:
density of the final state against the exact Sedov blast wave.

  * exact solution ....... ctypes over oracle/_build/libsedov_oracle.so (oracle/sedov_exact.cpp,
                           restating /root/reference/sedov/sedov_sol.cpp; pinned against the
                           compiled reference, see that file's header)
  * compute_density ...... LagrangianHydroOperator::ComputeDensity, laghos_solver.cpp:542-563,
                           with DensityIntegrator, laghos_assembly.cpp:26-41 (numpy, per zone)
  * density_error ........ laghos.cpp:1007-1086 (error rule, projection of both fields on
                           the quadrature space, Integrate)

Only tests/ import this.
"""
import ctypes
import os
import subprocess

import numpy as np

from .fem import bernstein_table, gauss_legendre, lagrange_tables

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_DP = ctypes.POINTER(ctypes.c_double)


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_DP)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libsedov_oracle.so")
        src = os.path.join(_HERE, "sedov_exact.cpp")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["make", "-C", _HERE, "_build/libsedov_oracle.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
    return _LIB


class SedovSol:
    """sedov_sol.hpp:21-76 (omega = 0 only, as the reference driver uses it)."""

    def __init__(self, dim, gamma, rho0, blast_energy, omega=0.0):
        assert omega == 0.0
        self.par = np.zeros(21)
        lib().lgo_sedov_setup(int(dim), ctypes.c_double(gamma), ctypes.c_double(rho0), ctypes.c_double(blast_energy),
                              ctypes.c_double(omega), _p(self.par))
        self.alpha = self.par[20]
        self.t = None

    def set_time(self, t):
        self.t = float(t)
        self.shock = np.zeros(6)
        lib().lgo_sedov_shock(_p(self.par), ctypes.c_double(self.t), _p(self.shock))
        self.r2, self.U, self.rho1, self.rho2, self.v2, self.p2 = self.shock

    def eval(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64).ravel()
        rho, v, p = (np.zeros(r.size) for _ in range(3))
        lib().lgo_sedov_eval(_p(self.par), ctypes.c_double(self.t), ctypes.c_long(r.size), _p(r), _p(rho), _p(v), _p(p))
        return rho, v, p


def _tensor(mats):
    """Kronecker product with the LAST axis fastest-varying first: index p = p0 + n*(p1 + n*p2)."""
    out = mats[0]
    for m in mats[1:]:
        out = np.kron(m, out)
    return out


def _zone_geometry(prob, S, B, G):
    """positions X (NE, P, dim) and det J (NE, P) at the tensor points of tables B, G (n1, D1D)."""
    dim, NE, ND = prob.dim, prob.NE, prob.ND
    xe = np.stack([S[c * prob.N + prob.h1map.reshape(NE, ND)] for c in range(dim)], axis=1)  # (NE, dim, ND)
    val = _tensor([B] * dim)                                    # (P, ND)
    X = np.einsum("pd,ecd->epc", val, xe)
    J = np.empty((NE, val.shape[0], dim, dim))
    for k in range(dim):
        dk = _tensor([G if a == k else B for a in range(dim)])  # d/d xi_k
        J[:, :, :, k] = np.einsum("pd,ecd->epc", dk, xe)
    return X, np.linalg.det(J)


def compute_density(prob, S, rho0DetJ0w):
    """rho grid function (NE*NL): per zone M^{-1} b on the current mesh."""
    dim = prob.dim
    _, detJ = _zone_geometry(prob, S, prob.B, prob.G)
    psi = _tensor([prob.Bl] * dim)                               # (NQ, NL)
    wdet = prob.W[None, :] * detJ                                # (NE, NQ)
    M = np.einsum("qi,eq,qj->eij", psi, wdet, psi)
    b = np.einsum("qi,eq->ei", psi, np.asarray(rho0DetJ0w).reshape(prob.NE, prob.NQ))
    return np.linalg.solve(M, b[:, :, None])[:, :, 0].reshape(-1)


def density_error(prob, S, rho, sol, origin, err_order):
    """sqrt(int (rho_exact - rho_h)^2) with the tensor Gauss-Legendre rule of order err_order."""
    dim = prob.dim
    n1 = err_order // 2 + 1
    pts, wts = gauss_legendre(n1)
    B, G = lagrange_tables(prob.gll, pts)
    Bl = bernstein_table(prob.order_e, pts)
    X, detJ = _zone_geometry(prob, S, B, G)
    W = _tensor([wts[:, None]] * dim)[:, 0]
    r = np.sqrt(((X - np.asarray(origin, dtype=float)[None, None, :dim]) ** 2).sum(axis=2))
    rho_x = sol.eval(r)[0].reshape(r.shape)
    rho_h = np.einsum("pl,el->ep", _tensor([Bl] * dim), np.asarray(rho).reshape(prob.NE, prob.NL))
    return float(np.sqrt(np.sum(W[None, :] * detJ * (rho_x - rho_h) ** 2)))


def err_order(order_v, order_e, order_q=-1):
    """laghos.cpp:1027"""
    return max((max(order_v, order_e) + 1) * 2, order_q) * 2

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


class PermutedProblem:
    """The same discrete problem as `base` under a random renumbering of the H1 nodes and a random order of the zones -
    what a general mesh library hands the operators (MFEM numbers vertices, then edge, face and interior dofs, and orders
    its elements its own way; the reference takes whatever `H1.GetElementRestriction` / the mesh give it).  Only the
    element-local dof order stays lexicographic: that IS the interface (laghos_assembly.cpp:296-514 index the E-vector so).
    None of the structure the kernels exploit when they find it survives: no x-chains of zones (merged E-vector), no
    consecutive x-rows of nodes (row loads of the slab K1), no locality between a zone and its neighbours."""

    def __init__(self, base, seed=7, permute_nodes=True, permute_elements=True):
        rng = np.random.default_rng(seed)
        self.base = base
        N, NE = base.N, base.NE
        self.node_perm = rng.permutation(N) if permute_nodes else np.arange(N)       # old node i -> new node node_perm[i]
        self.elem_perm = rng.permutation(NE) if permute_elements else np.arange(NE)  # new zone j = old zone elem_perm[j]
        hm = np.asarray(base.h1map).reshape(NE, base.ND)
        self.h1map = np.ascontiguousarray(self.node_perm[hm[self.elem_perm]].astype(np.int32))
        self.ess = [np.sort(self.node_perm[np.asarray(e, dtype=np.int64)]).astype(np.int32) for e in base.ess]
        own = np.empty(N)
        own[self.node_perm] = np.asarray(base.owner)
        self.owner = own

    def __getattr__(self, name):  # everything that does not depend on the numbering
        return getattr(self.base, name)

    def nodes(self, v):
        """a node vector of `dim` (or 1) components in the new numbering"""
        v = np.asarray(v).reshape(-1, self.base.N)
        out = np.empty_like(v)
        out[:, self.node_perm] = v
        return out.reshape(-1)

    def zones(self, a, per_zone):
        return np.asarray(a).reshape(self.base.NE, per_zone)[self.elem_perm].reshape(-1)

    def state(self, S):
        """a state vector [x | v | e] of the base problem in the new numbering"""
        b = self.base
        return np.concatenate([self.nodes(S[:b.H1V]), self.nodes(S[b.H1V:2 * b.H1V]), self.zones(S[2 * b.H1V:], b.NL)])

    def initial_state(self):
        b = self.base
        S, rho_l2, gamma, rho0_q = b.initial_state()
        return self.state(S), self.zones(rho_l2, b.NL), np.asarray(gamma)[self.elem_perm].copy(), self.zones(rho0_q, b.NQ)

    def accel_source(self):
        return self.nodes(self.base.accel_source())


class CurvedInitialMesh:
    """The same problem on an initial mesh whose zones are NOT affine: every interior node of the structured mesh is moved by
    a smooth random fraction of the mesh width before anything is set up.  Nothing the kernels find on a Cartesian mesh
    holds any more: Jac0inv varies inside a zone (no compact form), rho0 detJ0 w is not W[q] s_e (stored mass table, no
    Kronecker form) - the reference's data makes neither assumption (laghos_solver.cpp:1170-1261, laghos_assembly.cpp:92-95)."""

    def __init__(self, base, amp=0.08, seed=3):
        self.base = base
        S, self._rho_l2, self._gamma, self._rho0_q = base.initial_state()
        rng = np.random.default_rng(seed)
        h = min(np.min(np.diff(b)) for b in base.breaks) / base.order_v
        X = S[:base.H1V].reshape(base.dim, base.N).copy()
        free = np.ones(base.N, dtype=bool)
        for e in base.ess:                     # boundary nodes stay on the boundary (all of them: simplest)
            free[np.asarray(e, dtype=np.int64)] = False
        X[:, free] += amp * h * rng.uniform(-1, 1, (base.dim, int(free.sum())))
        self._S = np.concatenate([X.reshape(-1), S[base.H1V:]])

    def __getattr__(self, name):
        return getattr(self.base, name)

    def initial_state(self):
        # (rho0 at the quadrature points of the moved mesh would be rho0(x_q); problems with constant rho0 - Sedov, Taylor-Green -
        #  keep their values)
        return self._S.copy(), self._rho_l2, self._gamma, self._rho0_q


class TwoBlocks:
    """Two copies of `base` side by side that share no node (the second one shifted along x): a mesh of two structured blocks.
    Zones and nodes of the first block first; PermutedProblem on top scrambles both numberings."""

    def __init__(self, base, shift=10.0):
        self.base, self.shift = base, shift
        self.N, self.NE = 2 * base.N, 2 * base.NE
        self.H1V, self.L2V = base.dim * self.N, 2 * base.L2V
        hm = np.asarray(base.h1map).reshape(base.NE, base.ND)
        self.h1map = np.ascontiguousarray(np.concatenate([hm, hm + base.N]).astype(np.int32))
        self.ess = [np.concatenate([np.asarray(e), np.asarray(e) + base.N]).astype(np.int32) for e in base.ess]
        self.owner = np.ones(self.N)

    def __getattr__(self, name):
        return getattr(self.base, name)

    def state(self, Sa, Sb):
        """[x | v | e] of the two-block mesh from a state of each block"""
        b, N0 = self.base, self.base.N
        out = []
        for blk in range(2 * b.dim):   # x components, then v components
            a_, b_ = Sa[blk * N0:(blk + 1) * N0], Sb[blk * N0:(blk + 1) * N0]
            out += [a_, b_ + (self.shift if blk == 0 else 0.0)]
        out += [Sa[2 * b.H1V:], Sb[2 * b.H1V:]]
        return np.concatenate(out)

    def initial_state(self):
        S, rho_l2, gamma, rho0_q = self.base.initial_state()
        return self.state(S, S), np.concatenate([rho_l2, rho_l2]), np.concatenate([np.asarray(gamma)] * 2), np.concatenate([rho0_q, rho0_q])

    def accel_source(self):
        a, N0 = self.base.accel_source(), self.base.N
        return np.concatenate([np.concatenate([a[c * N0:(c + 1) * N0]] * 2) for c in range(self.base.dim)])

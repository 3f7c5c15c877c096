"""ORACLE (test infrastructure only) — problem setup in numpy.

Restates, independently of the product's C++ host code, the pieces of upstream
MFEM / laghos.cpp that sit *around* the hot path and are needed to reproduce the
reference's golden |e| values:

  * 1-D tables: Gauss-Legendre rule on [0,1], Gauss-Lobatto nodes, Lagrange
    (H1) basis + derivative, Bernstein (L2) basis          (SURVEY A1-A3, A5;
    corroborated by /root/reference/amr/laghos_assembly.cpp:32-58)
  * Cartesian tensor-product meshes equivalent to data/square01_quad.mesh,
    data/cube01_hex.mesh, data/box01_hex.mesh, data/rectangle01_quad.mesh after
    `-rs` uniform refinements (laghos.cpp:378-392; the mesh files are all
    axis-aligned structured grids with element axes = global axes)
  * H1 numbering / lexicographic element maps, byNODES vectors
  * essential dofs from boundary attributes 1/2/3 (laghos.cpp:499-515)
  * initial conditions for problems 0-7 incl. the Sedov delta function and
    the nodal-L2 -> Bernstein projection (laghos.cpp:568-632, :1094-1275; A11, A12)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------
# 1-D tables
# ----------------------------------------------------------------------------
def gauss_legendre(n):
    """n-point Gauss-Legendre rule on [0,1] (points ascending, weights)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_lobatto(n):
    """n Gauss-Lobatto points on [0,1] (n >= 2): endpoints + roots of P'_{n-1}."""
    if n == 2:
        return np.array([0.0, 1.0])
    P = np.polynomial.legendre.Legendre.basis(n - 1)
    dP = P.deriv()
    r = np.sort(np.real(dP.roots()))
    d2P = dP.deriv()
    for _ in range(3):  # Newton polish
        r = r - dP(r) / d2P(r)
    # enforce exact symmetry about 0
    r = 0.5 * (r - r[::-1])
    x = np.concatenate([[-1.0], r, [1.0]])
    return 0.5 * (x + 1.0)


def lagrange_tables(nodes, pts):
    """B[q,d] = l_d(pts[q]), G[q,d] = l_d'(pts[q]) for the Lagrange basis on nodes."""
    nd, nq = len(nodes), len(pts)
    B = np.zeros((nq, nd))
    G = np.zeros((nq, nd))
    for d in range(nd):
        others = [m for m in range(nd) if m != d]
        denom = np.prod([nodes[d] - nodes[m] for m in others])
        for q in range(nq):
            x = pts[q]
            B[q, d] = np.prod([x - nodes[m] for m in others]) / denom
            s = 0.0
            for k in others:
                s += np.prod([x - nodes[m] for m in others if m != k])
            G[q, d] = s / denom
    return B, G


def bernstein_table(p, pts):
    """B[q,l] = C(p,l) x^l (1-x)^(p-l)."""
    B = np.zeros((len(pts), p + 1))
    for l in range(p + 1):
        B[:, l] = math.comb(p, l) * pts ** l * (1.0 - pts) ** (p - l)
    return B


def quad_points_1d(order_v, order_e, order_q=-1):
    """laghos_solver.cpp:145-147: order = 3*ok+ot-1; n = order/2+1 GL points."""
    order = order_q if order_q > 0 else 3 * order_v + order_e - 1
    return order // 2 + 1


# ----------------------------------------------------------------------------
# Mesh (tensor-product, axis aligned)
# ----------------------------------------------------------------------------
MESHES = {
    # name: per-axis break points of the coarse mesh file
    "square01_quad": [[0.0, 0.5, 1.0], [0.0, 0.5, 1.0]],
    "cube01_hex": [[0.0, 0.5, 1.0], [0.0, 0.5, 1.0], [0.0, 0.5, 1.0]],
    "box01_hex": [[0.0, 1.0, 3.0, 5.0, 7.0], [0.0, 1.5, 3.0], [0.0, 1.5, 3.0]],
    "rectangle01_quad": [[float(i) for i in range(8)], [0.0, 1.0, 2.0, 3.0]],
    "square_gresho": [[-0.5, 0.0, 0.5], [-0.5, 0.0, 0.5]],
    "rt2D": [[0.0, 0.5], [-1.0, -0.5, 0.0, 0.5, 1.0]],
}


def refine_breaks(b):
    b = np.asarray(b, dtype=float)
    out = np.empty(2 * len(b) - 1)
    out[0::2] = b
    out[1::2] = 0.5 * (b[:-1] + b[1:])
    return out


class Problem:
    """Everything the hot path needs for one (mesh, order, problem) configuration,
    optionally restricted to one rank's sub-box of elements (block partition)."""

    def __init__(self, mesh="cube01_hex", rs=0, order_v=2, order_e=1, problem=1,
                 blast_energy=1.0, order_q=-1, rank=0, pgrid=None, breaks=None):
        self.problem = problem
        self.order_v, self.order_e = order_v, order_e
        brk = [np.asarray(b, dtype=float) for b in (breaks if breaks is not None else MESHES[mesh])]
        for _ in range(rs):
            brk = [refine_breaks(b) for b in brk]
        self.dim = dim = len(brk)
        self.gbreaks = brk
        gne = [len(b) - 1 for b in brk]
        self.global_ne = gne
        # block partition of the element grid over pgrid ranks (rank-major x fastest)
        if pgrid is None:
            pgrid = [1] * dim
        self.pgrid = list(pgrid)
        self.nranks = int(np.prod(pgrid))
        self.rank = rank
        rc = np.unravel_index(rank, pgrid[::-1])[::-1]  # x fastest
        self.rcoord = [int(c) for c in rc]
        self.eoff = []
        self.ne = []
        for a in range(dim):
            assert gne[a] % pgrid[a] == 0, "element grid must divide evenly"
            n = gne[a] // pgrid[a]
            self.ne.append(n)
            self.eoff.append(self.rcoord[a] * n)
        self.breaks = [brk[a][self.eoff[a]: self.eoff[a] + self.ne[a] + 1] for a in range(dim)]
        self.NE = int(np.prod(self.ne))
        self.global_NE = int(np.prod(gne))

        p = order_v
        self.D1D, self.L1D = p + 1, order_e + 1
        self.Q1D = quad_points_1d(order_v, order_e, order_q)
        self.qpts, self.qwts = gauss_legendre(self.Q1D)
        self.gll = gauss_lobatto(self.D1D)
        self.B, self.G = lagrange_tables(self.gll, self.qpts)      # H1: (Q,D)
        self.Bl = bernstein_table(order_e, self.qpts)              # L2: (Q,L)
        W = self.qwts
        for _ in range(dim - 1):
            W = np.multiply.outer(self.qwts, W)                    # index [qz,qy,qx]
        self.W = W.reshape(-1)                                      # q = qx + Q*(qy + Q*qz)
        self.ND, self.NQ, self.NL = self.D1D ** dim, self.Q1D ** dim, self.L1D ** dim

        # H1 nodes (local to this rank's box), lexicographic, x fastest
        self.nn = [n * p + 1 for n in self.ne]
        self.N = int(np.prod(self.nn))
        self.gnn = [n * p + 1 for n in gne]
        self.global_N = int(np.prod(self.gnn))
        coords1d = []
        for a in range(dim):
            b = self.breaks[a]
            c = np.empty(self.nn[a])
            for e in range(self.ne[a]):
                hlen = b[e + 1] - b[e]
                for d in range(p):
                    c[e * p + d] = b[e] + hlen * self.gll[d]
            c[-1] = b[-1]
            coords1d.append(c)
        self.coords1d = coords1d
        self.H1V, self.L2V = dim * self.N, self.NE * self.NL

        # element -> node map, lexicographic local dofs
        self.h1map = self._build_h1map()

        # essential (scalar-node) lists per velocity component: attribute a+1 =
        # faces normal to axis a on the GLOBAL boundary (data/cube01_hex.mesh:28-53)
        idx = np.arange(self.N).reshape(self.nn[::-1])  # [iz,iy,ix]
        self.ess = []
        for a in range(dim):
            sel = []
            ax = dim - 1 - a
            if self.rcoord[a] == 0:
                sel.append(np.take(idx, 0, axis=ax).reshape(-1))
            if self.rcoord[a] == self.pgrid[a] - 1:
                sel.append(np.take(idx, self.nn[a] - 1, axis=ax).reshape(-1))
            self.ess.append(np.unique(np.concatenate(sel)).astype(np.int32) if sel
                            else np.zeros(0, dtype=np.int32))

        # ownership for dot products: a shared node is owned by the lower-rank
        # side, i.e. a rank does NOT own its low faces unless on the global boundary
        own = np.ones(self.nn[::-1])
        for a in range(dim):
            if self.rcoord[a] > 0:
                ax = dim - 1 - a
                sl = [slice(None)] * dim
                sl[ax] = 0
                own[tuple(sl)] = 0.0
        self.owner = own.reshape(-1)
        self.blast_energy = blast_energy

    # ---- numbering ---------------------------------------------------------
    def _build_h1map(self):
        dim, p, D = self.dim, self.order_v, self.D1D
        m = np.empty((self.NE, self.ND), dtype=np.int32)
        if dim == 2:
            nx, ny = self.ne
            Nx = self.nn[0]
            ex, ey = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
            ex, ey = ex.reshape(-1), ey.reshape(-1)  # e = ex + nx*ey
            for dy in range(D):
                for dx in range(D):
                    m[:, dx + D * dy] = (ex * p + dx) + Nx * (ey * p + dy)
        else:
            nx, ny, nz = self.ne
            Nx, Ny = self.nn[0], self.nn[1]
            e = np.arange(self.NE)
            ex, ey, ez = e % nx, (e // nx) % ny, e // (nx * ny)
            for dz in range(D):
                for dy in range(D):
                    for dx in range(D):
                        m[:, dx + D * (dy + D * dz)] = (ex * p + dx) + Nx * ((ey * p + dy) + Ny * (ez * p + dz))
        return m

    def neighbors(self):
        """Ranks sharing H1 nodes with this one (faces, edges, corners) and the shared
        local node lists, enumerated in local lexicographic order (same order on both
        sides).  Returns (ranks, [node arrays])."""
        dim = self.dim
        idx = np.arange(self.N).reshape(self.nn[::-1])
        ranks, lists = [], []
        rng = [(-1, 0, 1)] * dim
        import itertools
        for off in itertools.product(*rng[::-1]):
            off = off[::-1]  # (ox, oy[, oz]), x fastest ordering of the loop
            if all(o == 0 for o in off):
                continue
            nc = [self.rcoord[a] + off[a] for a in range(dim)]
            if any(c < 0 or c >= self.pgrid[a] for a, c in enumerate(nc)):
                continue
            nr, stride = 0, 1
            for a in range(dim):
                nr += stride * nc[a]
                stride *= self.pgrid[a]
            sl = []
            for a in range(dim):
                if off[a] < 0:
                    sl.append(slice(0, 1))
                elif off[a] > 0:
                    sl.append(slice(self.nn[a] - 1, self.nn[a]))
                else:
                    sl.append(slice(None))
            ranks.append(nr)
            lists.append(idx[tuple(sl[::-1])].reshape(-1).astype(np.int32))
        return ranks, lists

    def node_coords(self):
        """(dim, N) initial node positions, byNODES."""
        grids = np.meshgrid(*self.coords1d[::-1], indexing="ij")  # [z,y,x] order
        return np.stack([g.reshape(-1) for g in grids[::-1]])

    def elem_index(self):
        """(NE, dim) local element coordinates (ex,ey[,ez])."""
        e = np.arange(self.NE)
        out = []
        stride = 1
        for a in range(self.dim):
            out.append((e // stride) % self.ne[a])
            stride *= self.ne[a]
        return np.stack(out, axis=1)

    def elem_points(self, ref1d):
        """Physical coordinates of the tensor points ref1d^dim in every element:
        (NE, npts, dim) with point index x fastest."""
        ei = self.elem_index()
        n = len(ref1d)
        per_axis = []
        for a in range(self.dim):
            b = self.breaks[a]
            lo, hi = b[ei[:, a]], b[ei[:, a] + 1]
            per_axis.append(lo[:, None] + (hi - lo)[:, None] * ref1d[None, :])  # (NE, n)
        pts = np.empty((self.NE, n ** self.dim, self.dim))
        if self.dim == 2:
            for j in range(n):
                for i in range(n):
                    pts[:, i + n * j, 0] = per_axis[0][:, i]
                    pts[:, i + n * j, 1] = per_axis[1][:, j]
        else:
            for k in range(n):
                for j in range(n):
                    for i in range(n):
                        q = i + n * (j + n * k)
                        pts[:, q, 0] = per_axis[0][:, i]
                        pts[:, q, 1] = per_axis[1][:, j]
                        pts[:, q, 2] = per_axis[2][:, k]
        return pts

    def elem_volumes(self):
        ei = self.elem_index()
        v = np.ones(self.NE)
        for a in range(self.dim):
            b = self.breaks[a]
            v *= b[ei[:, a] + 1] - b[ei[:, a]]
        return v

    # ---- problem definitions (laghos.cpp:1094-1275) --------------------------
    def rho0(self, x):
        p, dim = self.problem, self.dim
        if p in (0, 1, 4):
            return np.ones(x.shape[:-1])
        if p == 7:  # Rayleigh-Taylor, laghos.cpp:1117
            return np.where(x[..., 1] >= 0.0, 2.0, 1.0)
        if p in (5, 6):  # 2D Riemann problems, laghos.cpp:1105-1116
            hx, hy = x[..., 0] >= 0.5, x[..., 1] >= 0.5
            if p == 5:
                return np.where(hx & hy, 0.5313, np.where(~hx & ~hy, 0.8, 1.0))
            return np.where(~hx & hy, 2.0, np.where(hx & ~hy, 3.0, 1.0))
        if p == 2:
            return np.where(x[..., 0] < 0.5, 1.0, 0.1)
        if p == 3:
            if dim == 2:
                return np.where((x[..., 0] > 1.0) & (x[..., 1] > 1.5), 0.125, 1.0)
            c = (x[..., 0] > 1.0) & (((x[..., 1] < 1.5) & (x[..., 2] < 1.5)) |
                                     ((x[..., 1] > 1.5) & (x[..., 2] > 1.5)))
            return np.where(c, 0.125, 1.0)
        raise ValueError("problem not supported by the oracle harness")

    def gamma_func(self, x):
        p = self.problem
        if p in (0, 4, 7):
            return np.full(x.shape[:-1], 5.0 / 3.0)
        if p in (1, 2, 5, 6):
            return np.full(x.shape[:-1], 1.4)
        if p == 3:
            return np.where((x[..., 0] > 1.0) & (x[..., 1] <= 1.5), 1.4, 1.5)
        raise ValueError

    def v0(self, x):
        p, dim = self.problem, self.dim
        v = np.zeros(x.shape)
        if p == 0:
            v[..., 0] = np.sin(np.pi * x[..., 0]) * np.cos(np.pi * x[..., 1])
            v[..., 1] = -np.cos(np.pi * x[..., 0]) * np.sin(np.pi * x[..., 1])
            if dim == 3:
                v[..., 0] *= np.cos(np.pi * x[..., 2])
                v[..., 1] *= np.cos(np.pi * x[..., 2])
        if p in (5, 6):  # laghos.cpp:1144-1145, :1178-1197
            x0, x1 = x[..., 0], x[..., 1]
            atn = np.power(x0 * (1.0 - x0) * 4 * x1 * (1.0 - x1) * 4.0, 0.4)
            hx, hy = x0 >= 0.5, x1 >= 0.5
            if p == 5:
                v[..., 0] = np.where(~hx & hy, 0.7276 * atn, 0.0 * atn)
                v[..., 1] = np.where(hx & ~hy, 0.7276 * atn, 0.0 * atn)
            else:
                v[..., 0] = np.where(hy, 0.75 * atn, -0.75 * atn)
                v[..., 1] = np.where(hx, -0.5 * atn, 0.5 * atn)
        if p == 7:  # laghos.cpp:1198-1203
            v[..., 1] = 0.02 * np.exp(-2 * np.pi * x[..., 1] * x[..., 1]) * np.cos(2 * np.pi * x[..., 0])
        if p == 4:  # Gresho vortex, laghos.cpp:1161-1177
            x0, x1 = x[..., 0], x[..., 1]
            r = np.sqrt(x0 * x0 + x1 * x1)
            rs = np.where(r > 0, r, 1.0)
            inner, ring = r < 0.2, (r >= 0.2) & (r < 0.4)
            v[..., 0] = np.where(inner, 5.0 * x1, np.where(ring, 2.0 * x1 / rs - 5.0 * x1, 0.0))
            v[..., 1] = np.where(inner, -5.0 * x0, np.where(ring, -2.0 * x0 / rs + 5.0 * x0, 0.0))
        return v

    def e0(self, x):
        p, dim = self.problem, self.dim
        if p == 0:
            denom = 2.0 / 3.0
            if dim == 2:
                val = 1.0 + (np.cos(2 * np.pi * x[..., 0]) + np.cos(2 * np.pi * x[..., 1])) / 4.0
            else:
                val = 100.0 + ((np.cos(2 * np.pi * x[..., 2]) + 2) *
                               (np.cos(2 * np.pi * x[..., 0]) + np.cos(2 * np.pi * x[..., 1])) - 2) / 16.0
            return val / denom
        if p == 1:
            return np.zeros(x.shape[:-1])
        if p == 3:
            return np.where(x[..., 0] > 1.0, 0.1, 1.0) / self.rho0(x) / (self.gamma_func(x) - 1.0)
        if p == 2:  # laghos.cpp:1228-1229
            return np.where(x[..., 0] < 0.5, 1.0, 0.1) / self.rho0(x) / (self.gamma_func(x) - 1.0)
        if p in (5, 6):  # laghos.cpp:1248-1267
            irg = 1.0 / self.rho0(x) / (self.gamma_func(x) - 1.0)
            if p == 5:
                return np.where((x[..., 0] >= 0.5) & (x[..., 1] >= 0.5), 0.4, 1.0) * irg
            return irg
        if p == 7:  # laghos.cpp:1268-1272
            rho, gamma = self.rho0(x), self.gamma_func(x)
            return (6.0 - rho * x[..., 1]) / (gamma - 1.0) / rho
        if p == 4:  # laghos.cpp:1232-1247
            x0, x1 = x[..., 0], x[..., 1]
            rsq = x0 * x0 + x1 * x1
            r = np.sqrt(rsq)
            gamma = 5.0 / 3.0
            rs = np.where(r > 0, r, 1.0)
            inner = (5.0 + 25.0 / 2.0 * rsq) / (gamma - 1.0)
            t1 = 9.0 - 4.0 * np.log(0.2) + 25.0 / 2.0 * rsq
            t2 = 20.0 * r - 4.0 * np.log(rs)
            ring = (t1 - t2) / (gamma - 1.0)
            outer = (3.0 + 4.0 * np.log(2.0)) / (gamma - 1.0)
            return np.where(r < 0.2, inner, np.where(r < 0.4, ring, outer))
        raise ValueError

    def source_type(self):
        if self.problem == 7:
            return 2  # gravity (laghos.cpp:645)
        return 1 if (self.problem == 0 and self.dim == 2) else 0

    def use_viscosity(self):
        return self.problem not in (0, 4)

    def use_vorticity(self):
        return self.problem == 7  # laghos.cpp:645

    def accel_source(self):
        """nodal projection of RTCoefficient (laghos_solver.hpp:221-231): (0, -1) at every node"""
        a = np.zeros((self.dim, self.N))
        a[1, :] = -1.0
        return a.reshape(-1)

    # ---- nodal L2 (Gauss-Legendre) -> Bernstein, exact change of basis --------
    def _nodal_to_bernstein(self, vals):
        """vals: (NE, L1D^dim) nodal values at the order_e+1 Gauss-Legendre points.
        GridFunction::ProjectGridFunction with a positive target basis is a local L2
        projection (laghos.cpp:583-588, :595, :622); both bases span Q_p on an affine
        element, so it equals the exact change of basis done here."""
        L = self.L1D
        nodes, _ = gauss_legendre(L)
        V = bernstein_table(self.order_e, nodes)       # V[i,l] = B_l(node_i)
        Vinv = np.linalg.inv(V)
        t = vals.reshape((self.NE,) + (L,) * self.dim)  # [e, (z,) y, x]
        for ax in range(1, self.dim + 1):
            t = np.moveaxis(np.tensordot(Vinv, t, axes=([1], [ax])), 0, ax)
        return t.reshape(self.NE, self.NL)

    def initial_state(self):
        """S = [x | v | e], plus rho0 (Bernstein L2 grid function), gamma per
        element and rho0 evaluated at the physical quadrature points."""
        dim = self.dim
        X = self.node_coords()                                   # (dim, N)
        v = self.v0(X.T).T.copy()                                # (dim, N) pointwise (A11)
        for a in range(dim):
            v[a, self.ess[a]] = 0.0                              # laghos.cpp:576-579
        gl_nodes, _ = gauss_legendre(self.L1D)
        xn = self.elem_points(gl_nodes)                          # (NE, NL, dim)
        rho_nodal = self.rho0(xn)
        rho_l2 = self._nodal_to_bernstein(rho_nodal)
        if self.problem == 1:
            e_l2 = self._sedov_delta()
        else:
            e_l2 = self._nodal_to_bernstein(self.e0(xn))
        ctr = self.elem_points(np.array([0.5]))[:, 0, :]
        gamma = self.gamma_func(ctr)                             # order-0 L2: centre value
        xq = self.elem_points(self.qpts)
        rho0_q = self.rho0(xq)                                   # (NE, NQ) coefficient at qpts (A8)
        S = np.concatenate([X.reshape(-1), v.reshape(-1), e_l2.reshape(-1)])
        return S, rho_l2.reshape(-1), gamma, rho0_q.reshape(-1)

    def _sedov_delta(self):
        """DeltaCoefficient at the origin scaled to E0/2^dim (laghos.cpp:597-616).
        Upstream ProjectDeltaCoefficient: in every element having the closest mesh
        vertex (the origin) as a vertex, the nodal L2 values are the vertex-peaked
        shape prod_a (1-x_a)^p (L2 element ProjectDelta), then everything is scaled
        by scale / integral.  In the Bernstein basis that shape is exactly the single
        corner dof, so only dof 0 of the origin element is non-zero."""
        L, p, dim = self.L1D, self.order_e, self.dim
        gl_nodes, gl_w = gauss_legendre(L)
        e_nodal = np.zeros((self.NE, self.NL))
        ei = self.elem_index()
        integral_local = 0.0
        origin_elem = np.all(ei == 0, axis=1) & all(c == 0 for c in self.rcoord)
        if np.any(origin_elem):
            assert all(abs(self.gbreaks[a][0]) < 1e-12 for a in range(dim))
            e_id = int(np.nonzero(origin_elem)[0][0])
            shp1 = (1.0 - gl_nodes) ** p
            shp = shp1
            w = gl_w
            for _ in range(dim - 1):
                shp = np.multiply.outer(shp1, shp)
                w = np.multiply.outer(gl_w, w)
            e_nodal[e_id] = shp.reshape(-1)
            integral_local = float(np.sum(shp * w)) * self.elem_volumes()[e_id]
        self._delta_integral_local = integral_local
        scale = self.blast_energy / 2 ** dim
        # the integral is global (MPI sum upstream); only one element contributes
        integral = integral_local
        if self.nranks > 1:
            # every rank can compute it: origin element volume is known globally
            h = [self.gbreaks[a][1] - self.gbreaks[a][0] for a in range(dim)]
            integral = float(np.prod(h)) / (p + 1) ** dim
        if integral_local:
            e_nodal *= scale / integral
        return self._nodal_to_bernstein(e_nodal)

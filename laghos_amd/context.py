"""Thin Python handle over the C ABI (include/laghos_hip.h) using torch CUDA
tensors as device memory.  Plumbing only: every method is one C-ABI call.

Used by the GPU parity tests and by bench.py; the product host layer that
mirrors the reference's C++ classes lives in laghos_amd/host/.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import LghConfig, check


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _np_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(t):
    """device pointer of a contiguous float64 CUDA tensor"""
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _dbl(a):
    """host pointer of a contiguous float64 numpy array (kept alive by the caller)"""
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_lib.c_dbl_p)


def sedov_setup(dim, gamma, rho0, blast_energy, omega=0.0):
    """SedovSol::SedovSol (sedov/sedov_sol.cpp:27-117): the 21-entry parameter block."""
    par = np.zeros(21)
    check(_lib.load().lgh_sedov_setup(int(dim), float(gamma), float(rho0), float(blast_energy), float(omega), _dbl(par)))
    return par


def sedov_shock(par, t):
    """SedovSol::SetTime: (r2, U, rho1, rho2, v2, p2)."""
    out = np.zeros(6)
    check(_lib.load().lgh_sedov_shock(_dbl(par), float(t), _dbl(out)))
    return out


def sedov_eval_point(par, t, r):
    """SedovSol::EvalSol at one radius (host)."""
    rho, v, p = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    check(_lib.load().lgh_sedov_eval_point(_dbl(par), float(t), float(r), ctypes.byref(rho), ctypes.byref(v),
                                           ctypes.byref(p)))
    return rho.value, v.value, p.value


class Context:
    """lgh_ctx owner.  Arguments follow struct lgh_config: tables are (Q,D)
    arrays B[q,d]; h1_map is (NE, ND); ess is a list of dim int arrays."""

    def __init__(self, dim, NE, D1D, Q1D, L1D, N, h1_map, B_h1, G_h1, B_l2, weights, gamma, ess,
                 owner=None, use_viscosity=True, use_vorticity=False, cfl=0.5, order_v=None,
                 device=0):
        self.lib = _lib.load()
        self.dim, self.NE, self.D1D, self.Q1D, self.L1D, self.N = dim, NE, D1D, Q1D, L1D, N
        self.ND, self.NQ, self.NL = D1D ** dim, Q1D ** dim, L1D ** dim
        self.H1V, self.L2V = dim * N, NE * self.NL
        self.device = torch.device("cuda", device)
        keep = dict(
            h1=_np_i32(np.asarray(h1_map).reshape(-1)),
            B=_np_f64(np.asarray(B_h1).T.reshape(-1)),   # -> [q + Q*d]
            G=_np_f64(np.asarray(G_h1).T.reshape(-1)),
            Bl=_np_f64(np.asarray(B_l2).T.reshape(-1)),
            W=_np_f64(weights), gamma=_np_f64(gamma),
            ess=[_np_i32(e) if len(e) else np.zeros(1, np.int32) for e in ess],
            owner=None if owner is None else _np_f64(owner),
        )
        self._keep = keep
        cfg = LghConfig()
        cfg.dim, cfg.NE, cfg.D1D, cfg.Q1D, cfg.L1D, cfg.N = dim, NE, D1D, Q1D, L1D, N
        ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
        cfg.h1_map = keep["h1"].ctypes.data_as(ip)
        cfg.B_h1 = keep["B"].ctypes.data_as(dp)
        cfg.G_h1 = keep["G"].ctypes.data_as(dp)
        cfg.B_l2 = keep["Bl"].ctypes.data_as(dp)
        cfg.weights = keep["W"].ctypes.data_as(dp)
        cfg.gamma = keep["gamma"].ctypes.data_as(dp)
        for k in range(3):
            cfg.ess_count[k] = len(ess[k]) if k < dim else 0
            cfg.ess[k] = keep["ess"][k].ctypes.data_as(ip) if k < dim else None
        cfg.owner = keep["owner"].ctypes.data_as(dp) if owner is not None else None
        cfg.use_viscosity, cfg.use_vorticity = int(use_viscosity), int(use_vorticity)
        cfg.cfl = cfl
        cfg.order_v = order_v if order_v is not None else D1D - 1
        cfg.device = device
        cfg.stream = None
        h = ctypes.c_void_p()
        check(self.lib.lgh_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.lgh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ----------------------------------------------------------------
    def empty(self, n):
        return torch.empty(int(n), dtype=torch.float64, device=self.device)

    # The library works on its own (non-blocking) HIP stream; torch fills tensors on torch's current
    # stream.  A tensor handed to the library must be complete first, so the helpers that WRITE device
    # memory through torch wait for torch's stream before they return (empty() writes nothing).
    def _torch_done(self):
        torch.cuda.current_stream(self.device).synchronize()

    def zeros(self, n):
        t = torch.zeros(int(n), dtype=torch.float64, device=self.device)
        self._torch_done()
        return t

    def to_dev(self, a):
        t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        self._torch_done()
        return t

    def clone(self, t):
        """torch clone of a device tensor, complete before the library may touch it"""
        c = t.clone()
        self._torch_done()
        return c

    def sync(self):
        check(self.lib.lgh_sync(self.h))

    def _view(self, ptr, n):
        """read a ctx-owned device array into a numpy array"""
        out = self.empty(n)
        self.sync()
        check(self.lib.lgh_vec_copy(self.h, _ptr(out), ctypes.c_void_p(ptr), int(n)))
        self.sync()
        return out.cpu().numpy()

    def _write(self, ptr, arr):
        t = self.to_dev(arr)
        torch.cuda.synchronize()
        check(self.lib.lgh_vec_copy(self.h, ctypes.c_void_p(ptr), _ptr(t), t.numel()))
        self.sync()

    @property
    def stressJinvT(self):
        return self._view(self.lib.lgh_qdata_stressJinvT(self.h), self.NE * self.NQ * self.dim ** 2)

    def set_stressJinvT(self, arr):
        self._write(self.lib.lgh_qdata_stressJinvT(self.h), arr)

    @property
    def Jac0inv(self):
        return self._view(self.lib.lgh_qdata_Jac0inv(self.h), self.NE * self.NQ * self.dim ** 2)

    @property
    def rho0DetJ0w(self):
        return self._view(self.lib.lgh_qdata_rho0DetJ0w(self.h), self.NE * self.NQ)

    @property
    def massD(self):
        return self._view(self.lib.lgh_mass_D(self.h), self.NE * self.NQ)

    @massD.setter
    def massD(self, arr):
        """overwrite the mass quadrature data (lgh_mass_D is called again after the write: its contract)"""
        self._write(self.lib.lgh_mass_D(self.h), np.ascontiguousarray(arr, dtype=np.float64))
        self.lib.lgh_mass_D(self.h)

    def write_massD_in_place(self, arr):
        """overwrite the table through the pointer as a caller that kept it would (no second lgh_mass_D call);
        lgh_mass_data_changed() is then the caller's duty"""
        if not hasattr(self, "_massD_ptr"):
            self._massD_ptr = self.lib.lgh_mass_D(self.h)
        self._write(self._massD_ptr, np.ascontiguousarray(arr, dtype=np.float64))

    def mass_data_changed(self):
        check(self.lib.lgh_mass_data_changed(self.h))

    @property
    def mass_diag(self):
        return self._view(self.lib.lgh_mass_diag(self.h), self.N)

    # ---- one method per C-ABI entry ------------------------------------------------
    def set_h0(self, h0):
        check(self.lib.lgh_set_h0(self.h, h0))

    def set_dt_est(self, v):
        check(self.lib.lgh_set_dt_est(self.h, v))

    def get_dt_est(self):
        v = ctypes.c_double()
        check(self.lib.lgh_get_dt_est(self.h, ctypes.byref(v)))
        return v.value

    def setup_rho0detj0(self, x0, rho0_l2, rho0_q):
        vol = ctypes.c_double()
        torch.cuda.synchronize()
        check(self.lib.lgh_setup_rho0detj0(self.h, _ptr(x0), _ptr(rho0_l2), _ptr(rho0_q), ctypes.byref(vol)))
        return vol.value

    def force_mult(self, x_l2, y_h1):
        check(self.lib.lgh_force_mult(self.h, _ptr(x_l2), _ptr(y_h1)))

    def force_mult_transpose(self, v_h1, y_l2):
        check(self.lib.lgh_force_mult_transpose(self.h, _ptr(v_h1), _ptr(y_l2)))

    def mass_set_ess(self, comp):
        check(self.lib.lgh_mass_set_essential_tdofs(self.h, comp))

    def mass_eliminate_rhs(self, b):
        check(self.lib.lgh_mass_eliminate_rhs(self.h, _ptr(b)))

    def mass_mult(self, space, x, y, full=False):
        fn = self.lib.lgh_mass_mult_full if full else self.lib.lgh_mass_mult
        check(fn(self.h, space, _ptr(x), _ptr(y)))

    def cg_solve(self, space, b, x, rel_tol, max_iter):
        it = ctypes.c_int(0)
        check(self.lib.lgh_cg_solve(self.h, space, _ptr(b), _ptr(x), rel_tol, max_iter, ctypes.byref(it)))
        return it.value

    def qupdate(self, S):
        check(self.lib.lgh_qupdate(self.h, _ptr(S)))

    def qupdate_set_tiny_grad(self, v):
        check(self.lib.lgh_qupdate_set_tiny_grad(self.h, float(v)))

    def table_symmetry(self):
        """(h1, l2): whether lgh_create found the 1-D tables mirror symmetric (plane-form mass kernels need it)."""
        a, b = ctypes.c_int(-1), ctypes.c_int(-1)
        check(self.lib.lgh_table_symmetry(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def k1_form(self):
        """Form of the lockstep velocity solve's mass-apply kernel: 'column', 'plane', 'slab', 'kron' or None."""
        f = ctypes.c_int(-2)
        check(self.lib.lgh_k1_form(self.h, ctypes.byref(f)))
        return {0: "column", 2: "plane", 4: "slab", 5: "kron"}.get(f.value)

    def mesh_order(self):
        """lgh_mesh_order: dict(structured, identity, components, extent) - the library's own zone order / node numbering."""
        o = (ctypes.c_long * 8)()
        check(self.lib.lgh_mesh_order(self.h, o))
        return dict(structured=bool(o[0]), identity=bool(o[1]), components=int(o[2]), extent=(int(o[3]), int(o[4]), int(o[5])))

    def test_vcg_k1(self, r, d_old, rz, rz_prev, first):
        """One launch of the lockstep solve's K1 (lgh_test_vcg_k1): (E-vector planes [3, NE*ND] tensor, den[3])."""
        y = self.empty(3 * self.NE * self.ND)
        rz, rzp, den = _np_f64(rz), _np_f64(rz_prev), np.zeros(3)
        check(self.lib.lgh_test_vcg_k1(self.h, _ptr(r), _ptr(d_old) if d_old is not None else None, _dbl(rz), _dbl(rzp),
                                       int(bool(first)), _ptr(y), _dbl(den)))
        return y.view(3, -1), den

    def test_vcg_merged_faces(self):
        """uint8 mask [NE*ND] of the E-vector entries K1 has already summed into their left x-neighbour's (slab K1, merged
        layout; all zero otherwise) and their number (lgh_test_vcg_merged_faces)."""
        mask = np.zeros(self.NE * self.ND, dtype=np.uint8)
        n = ctypes.c_long(0)
        check(self.lib.lgh_test_vcg_merged_faces(self.h, mask.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
        return mask, n.value

    def test_vcg_k2(self, it, y_E, r, d, x, den, rz, rz_prev, alpha_prev):
        """One launch of the lockstep solve's K2 (lgh_test_vcg_k2); r, d, x are updated in place.
        Returns ((r, z) of the new residual [3], deferred_x)."""
        den, rz, rzp, al, out = _np_f64(den), _np_f64(rz), _np_f64(rz_prev), _np_f64(alpha_prev), np.zeros(3)
        form = ctypes.c_int(-1)
        check(self.lib.lgh_test_vcg_k2(self.h, int(it), _ptr(y_E), _ptr(r), _ptr(d), _ptr(x), _dbl(den), _dbl(rz), _dbl(rzp),
                                       _dbl(al), _dbl(out), ctypes.byref(form)))
        return out, bool(form.value)

    def mass_data_form(self):
        """'rank1' when the mass kernels read D[q, e] = W[q] s_e (one double per element), 'stored' otherwise."""
        f = ctypes.c_int(-1)
        check(self.lib.lgh_mass_data_form(self.h, ctypes.byref(f)))
        return "rank1" if f.value == 1 else "stored"

    def jac0inv_form(self):
        """'compact' when the row-form update reads one Jac0inv per zone (lgh_jac0inv_form), 'stored' otherwise."""
        f = ctypes.c_int(-1)
        check(self.lib.lgh_jac0inv_form(self.h, ctypes.byref(f)))
        return "compact" if f.value == 1 else "stored"

    def qupdate_store_stress(self, on):
        """0: lgh_qupdate keeps the stress in registers (stressJinvT is not written; its readers refuse)."""
        check(self.lib.lgh_qupdate_store_stress(self.h, int(bool(on))))

    def set_fused_forces(self, on):
        check(self.lib.lgh_set_fused_forces(self.h, 1 if on else 0))

    def reset_quadrature_data(self):
        check(self.lib.lgh_reset_quadrature_data(self.h))

    def fused_force_mult(self, y_h1):
        """F.1 formed by the last qupdate, summed to the H1 L-vector; False when it is not on hand"""
        return self.lib.lgh_fused_force_mult(self.h, _ptr(y_h1)) == 0

    def fused_force_mult_transpose(self, y_l2):
        return self.lib.lgh_fused_force_mult_transpose(self.h, _ptr(y_l2)) == 0

    def quadrature_generation(self):
        """(generation, fused F.1 on hand, fused F^T v on hand)"""
        g, a, b = ctypes.c_ulong(0), ctypes.c_int(-1), ctypes.c_int(-1)
        check(self.lib.lgh_quadrature_generation(self.h, ctypes.byref(g), ctypes.byref(a), ctypes.byref(b)))
        return g.value, a.value, b.value

    def solve_velocity(self, S, dS, one, rhs, work, rel_tol, max_iter):
        """one = None: the operator's own constant-one vector (laghos_solver.cpp:170-171)"""
        it = ctypes.c_int(0)
        check(self.lib.lgh_solve_velocity(self.h, _ptr(S), _ptr(dS), _ptr(one) if one is not None else None, _ptr(rhs), _ptr(work),
                                          rel_tol, max_iter, ctypes.byref(it)))
        return it.value

    def solve_energy(self, S, v, dS, e_rhs, rel_tol, max_iter, e_source=None):
        it = ctypes.c_int(0)
        check(self.lib.lgh_solve_energy(self.h, _ptr(S), _ptr(v), _ptr(dS), _ptr(e_rhs),
                                        _ptr(e_source) if e_source is not None else None,
                                        rel_tol, max_iter, ctypes.byref(it)))
        return it.value

    def set_velocity_source(self, accel):
        check(self.lib.lgh_set_velocity_source(self.h, _ptr(accel) if accel is not None else None))

    def tg_source_2d(self, S, out):
        check(self.lib.lgh_tg_source_2d(self.h, _ptr(S), _ptr(out)))

    # ---- `-err`: density against the exact Sedov solution (include/laghos_hip.h) ----
    def compute_density(self, x, rho_l2):
        """ComputeDensity (laghos_solver.cpp:542-563); x = S (positions first)."""
        check(self.lib.lgh_compute_density(self.h, _ptr(x), _ptr(rho_l2)))

    def sedov_eval(self, par, t, r, rho, v, P):
        """SedovSol::EvalSol for a device array of radii."""
        check(self.lib.lgh_sedov_eval(self.h, _dbl(par), float(t), r.numel(), _ptr(r), _ptr(rho), _ptr(v), _ptr(P)))

    def sedov_density_error(self, x, rho_l2, par, t, origin, weights, B_h1, G_h1, B_l2):
        """Integral of (rho_exact - rho_h)^2 over the current mesh (laghos.cpp:1027-1080);
        the tables are host arrays of the error rule: weights[n], B/G [p + n*d], B_l2 [p + n*l]."""
        import numpy as np
        w = np.ascontiguousarray(weights, dtype=np.float64)
        tabs = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in (B_h1, G_h1, B_l2)]
        org = np.zeros(3)
        org[:len(origin)] = origin
        out = ctypes.c_double()
        check(self.lib.lgh_sedov_density_error(self.h, _ptr(x), _ptr(rho_l2), _dbl(par), float(t), _dbl(org),
                                               int(w.size), _dbl(w), _dbl(tabs[0]), _dbl(tabs[1]), _dbl(tabs[2]),
                                               ctypes.byref(out)))
        return out.value

    def solve_energy_begin(self, S, v, dS, e_rhs, rel_tol, max_iter, e_source=None):
        check(self.lib.lgh_solve_energy_begin(self.h, _ptr(S), _ptr(v), _ptr(dS), _ptr(e_rhs),
                                              _ptr(e_source) if e_source is not None else None,
                                              rel_tol, max_iter))

    def solve_energy_end(self):
        it = ctypes.c_int(0)
        check(self.lib.lgh_solve_energy_end(self.h, ctypes.byref(it)))
        return it.value

    def vec_axpby(self, z, a, x, b, y):
        check(self.lib.lgh_vec_axpby(self.h, _ptr(z), a, _ptr(x), b, _ptr(y), z.numel()))

    def vec_axpby_pair(self, z1, a1, x1, b1, z2, a2, x2, b2, y):
        check(self.lib.lgh_vec_axpby_pair(self.h, _ptr(z1), a1, _ptr(x1), b1, _ptr(z2), a2, _ptr(x2), b2, _ptr(y), z1.numel()))

    def vec_copy(self, y, x):
        check(self.lib.lgh_vec_copy(self.h, _ptr(y), _ptr(x), y.numel()))

    def vec_dot(self, x, y):
        v = ctypes.c_double()
        check(self.lib.lgh_vec_dot(self.h, _ptr(x), _ptr(y), x.numel(), ctypes.byref(v)))
        return v.value

    def internal_energy(self, e):
        v = ctypes.c_double()
        check(self.lib.lgh_internal_energy(self.h, _ptr(e), ctypes.byref(v)))
        return v.value

    def kinetic_energy(self, vel):
        v = ctypes.c_double()
        check(self.lib.lgh_kinetic_energy(self.h, _ptr(vel), ctypes.byref(v)))
        return v.value

    def timers(self):
        t = (ctypes.c_double * 4)()
        c = (ctypes.c_long * 3)()
        check(self.lib.lgh_get_timers(self.h, t, c))
        return dict(cgH1=t[0], cgL2=t[1], force=t[2], qdata=t[3], H1iter=c[0], L2iter=c[1],
                    quad_tstep=c[2])

    def reset_timers(self):
        check(self.lib.lgh_reset_timers(self.h))

    def enable_timers(self, on):
        check(self.lib.lgh_enable_timers(self.h, int(on)))

    # E-level (tests)
    def force_mult_E(self, sJit, xE, yE):
        check(self.lib.lgh_force_mult_E(self.h, _ptr(sJit), _ptr(xE), _ptr(yE)))

    def force_mult_transpose_E(self, sJit, vE, yE):
        check(self.lib.lgh_force_mult_transpose_E(self.h, _ptr(sJit), _ptr(vE), _ptr(yE)))

    def mass_apply_E(self, space, xE, yE):
        check(self.lib.lgh_mass_apply_E(self.h, space, _ptr(xE), _ptr(yE)))

    def test_eig(self, dim, A, lam, vec):
        check(self.lib.lgh_test_eig(self.h, dim, lam.numel(), _ptr(A), _ptr(lam), _ptr(vec)))

    def test_singular(self, dim, A, sv):
        check(self.lib.lgh_test_singular(self.h, dim, sv.numel(), _ptr(A), _ptr(sv)))

    def test_sqrt(self, x, y):
        check(self.lib.lgh_test_sqrt(self.h, y.numel(), _ptr(x), _ptr(y)))

    # multi-GPU
    def comm_init(self, nranks, rank, unique_id):
        check(self.lib.lgh_comm_init(self.h, nranks, rank, unique_id))

    def comm_set_neighbors(self, nbr_rank, nbr_nodes):
        n = len(nbr_rank)
        ranks = _np_i32(nbr_rank) if n else np.zeros(1, np.int32)
        counts = _np_i32([len(x) for x in nbr_nodes]) if n else np.zeros(1, np.int32)
        lists = [_np_i32(x) for x in nbr_nodes]
        ip = ctypes.POINTER(ctypes.c_int)
        arr = (ip * max(n, 1))(*[l.ctypes.data_as(ip) for l in lists])
        check(self.lib.lgh_comm_set_neighbors(self.h, n, ranks.ctypes.data_as(ip), counts.ctypes.data_as(ip), arr))

    def test_word_peers(self, nwords):
        cap = ctypes.c_long(-1)
        check(self.lib.lgh_test_word_peers(self.h, nwords, ctypes.byref(cap)))
        return cap.value

    def test_set_rank(self, nranks, rank):
        check(self.lib.lgh_test_set_rank(self.h, nranks, rank))

    def test_halo_pack(self, v, ncomp, out):
        check(self.lib.lgh_test_halo_pack(self.h, _ptr(v), ncomp, _ptr(out)))

    def test_halo_combine(self, inp, v, ncomp):
        check(self.lib.lgh_test_halo_combine(self.h, _ptr(inp), _ptr(v), ncomp))

    def halo_sum(self, v, ncomp):
        check(self.lib.lgh_halo_sum(self.h, _ptr(v), ncomp))

    def allreduce(self, value, op=0):
        v = ctypes.c_double(value)
        check(self.lib.lgh_allreduce(self.h, ctypes.byref(v), op))
        return v.value


def unique_id():
    buf = ctypes.create_string_buffer(128)
    check(_lib.load().lgh_comm_unique_id(buf))
    return buf.raw

"""ORACLE (test infrastructure only) — orchestration of the CPU restatement.

Mirrors, in Python over oracle/_build/liblaghos_oracle.so:
  * LagrangianHydroOperator ctor/Mult/GetTimeStepEstimate/ResetTimeStepEstimate
    (/root/reference/laghos_solver.cpp:104-294, :308-327, :527-540)
  * RK4Solver (upstream MFEM) and RK2AvgSolver (laghos_solver.cpp:1447-1487)
  * the time loop with adaptive dt control (laghos.cpp:706-778) and the |e|
    report / --checks probe points (laghos.cpp:792-839, :903-919)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

from .fem import Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = ctypes.POINTER(ctypes.c_double)
c_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    so = os.path.join(_HERE, "_build", "liblaghos_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("laghos_oracle.cpp", "smallmat.hpp", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


_LIB_NATIVE = None
PARITY_FLAGS = "-O3 -march=x86-64-v3 -fopenmp -fno-fast-math -ffp-contract=off"   # oracle/Makefile CXXFLAGS: the checker
NATIVE_FLAGS = "-O3 -march=native -mprefer-vector-width=256 -fopenmp -fno-fast-math -ffp-contract=fast"     # oracle/Makefile NATIVEFLAGS: timing only


def build_native():
    """The timing build of the same source (`make native`: -march=native, FMA contraction), compiled on the machine that
    runs it - bench.py's cpu_baseline only; never the checker."""
    so = os.path.join(_HERE, "_build", "liblaghos_oracle_native.so")
    # (-B: always for THIS host - a copy that travelled with the tree was built for another machine's -march=native)
    subprocess.check_call(["make", "-B", "-C", _HERE, "native"], stdout=subprocess.DEVNULL)
    return so


def lib(native=False):
    global _LIB, _LIB_NATIVE
    if native:
        if _LIB_NATIVE is None:
            _LIB_NATIVE = _bind(ctypes.CDLL(build_native()))
        return _LIB_NATIVE
    if _LIB is None:
        _LIB = _bind(ctypes.CDLL(build()))
    return _LIB


def _bind(L):
    L.lgo_create.restype = ctypes.c_void_p
    for name in ("lgo_stressJinvT", "lgo_Jac0inv", "lgo_rho0DetJ0w", "lgo_massD", "lgo_diagV",
                 "lgo_q_dx", "lgo_q_dv", "lgo_q_e"):
        getattr(L, name).restype = c_dp
        getattr(L, name).argtypes = [ctypes.c_void_p]
    for name in ("lgo_get_h0", "lgo_get_dt_est"):
        getattr(L, name).restype = ctypes.c_double
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.lgo_set_h0.argtypes = [ctypes.c_void_p, ctypes.c_double]
    L.lgo_set_dt_est.argtypes = [ctypes.c_void_p, ctypes.c_double]
    L.lgo_setup_rho0detj0.restype = ctypes.c_double
    L.lgo_internal_energy.restype = ctypes.c_double
    L.lgo_kinetic_energy.restype = ctypes.c_double
    L.lgo_sv3.restype = ctypes.c_double
    L.lgo_sv2.restype = ctypes.c_double
    L.lgo_cg.restype = ctypes.c_int
    return L


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


HALO_FN = ctypes.CFUNCTYPE(None, c_dp, ctypes.c_int, ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p)


class Hydro:
    """The oracle's LagrangianHydroOperator (PA branch, dim >= 2)."""

    def __init__(self, prob: Problem, cfl=0.5, cg_tol=1e-8, cg_max_iter=300, comm=None, native=False):
        self.p = prob
        self.L = L = lib(native)  # native: the timing build (bench.py's cpu_baseline), never the checker
        self.cg_tol, self.cg_max_iter = cg_tol, cg_max_iter
        self.comm = comm
        S, rho_l2, gamma, rho0_q = prob.initial_state()
        self.S0 = S
        self._keep = dict(
            h1map=np.ascontiguousarray(prob.h1map.reshape(-1)),
            B=np.ascontiguousarray(prob.B.T.reshape(-1)),    # (Q,D) q fastest
            G=np.ascontiguousarray(prob.G.T.reshape(-1)),
            Bl=np.ascontiguousarray(prob.Bl.T.reshape(-1)),
            W=np.ascontiguousarray(prob.W),
            gamma=np.ascontiguousarray(gamma, dtype=np.float64),
            essc=np.array([len(e) for e in prob.ess] + [0] * (3 - prob.dim), dtype=np.int32),
            ess=[np.ascontiguousarray(e) for e in prob.ess] + [np.zeros(1, np.int32)] * (3 - prob.dim),
            owner=np.ascontiguousarray(prob.owner),
        )
        k = self._keep
        ess = [e if len(e) else np.zeros(1, np.int32) for e in k["ess"]]
        self.h = ctypes.c_void_p(L.lgo_create(
            prob.dim, prob.NE, prob.D1D, prob.Q1D, prob.L1D, prob.N, _ip(k["h1map"]),
            _dp(k["B"]), _dp(k["G"]), _dp(k["Bl"]), _dp(k["W"]), _dp(k["gamma"]), _ip(k["essc"]),
            _ip(ess[0]), _ip(ess[1]), _ip(ess[2]), _dp(k["owner"]),
            int(prob.use_viscosity()), int(prob.use_vorticity()), ctypes.c_double(cfl), prob.order_v))
        if comm is not None:
            self._install_comm(comm)
        # Rho0DetJ0Vol + h0 (laghos_solver.cpp:223-262)
        x0 = np.ascontiguousarray(S[:prob.H1V])
        vol = L.lgo_setup_rho0detj0(self.h, _dp(x0), _dp(np.ascontiguousarray(rho_l2)),
                                    _dp(np.ascontiguousarray(rho0_q)))
        ne = float(prob.NE)
        if comm is not None:
            vol = comm.allreduce_sum(vol)
            ne = comm.allreduce_sum(ne)
        self.volume = vol
        h0 = (vol / ne) ** (1.0 / prob.dim) / prob.order_v
        L.lgo_set_h0(self.h, ctypes.c_double(h0))
        L.lgo_mass_assemble_diag(self.h)  # OperatorJacobiSmoother (laghos_solver.cpp:266-270)
        self.source_type = prob.source_type()
        self._accel = np.ascontiguousarray(prob.accel_source()) if self.source_type == 2 else None
        self.qdata_is_current = False

    def _install_comm(self, comm):
        prob = self.p

        def halo(ptr, ncomp, _user):
            arr = np.ctypeslib.as_array(ptr, shape=(ncomp * prob.N,))
            comm.halo_sum(arr, ncomp)

        def allred(v, op, _user):
            return comm.allreduce_sum(v) if op == 0 else comm.allreduce_min(v)

        self._halo_cb, self._ar_cb = HALO_FN(halo), ALLREDUCE_FN(allred)
        self.L.lgo_set_comm_hooks(self._halo_cb, self._ar_cb, None)

    def close(self):
        if self.h:
            self.L.lgo_set_comm_hooks(None, None, None)
            self.L.lgo_destroy(self.h)
            self.h = None

    # -- views on QuadratureData ------------------------------------------------
    def _view(self, fn, n):
        return np.ctypeslib.as_array(getattr(self.L, fn)(self.h), shape=(n,))

    @property
    def stressJinvT(self):
        return self._view("lgo_stressJinvT", self.p.NE * self.p.NQ * self.p.dim ** 2)

    @property
    def Jac0inv(self):
        return self._view("lgo_Jac0inv", self.p.NE * self.p.NQ * self.p.dim ** 2)

    @property
    def rho0DetJ0w(self):
        return self._view("lgo_rho0DetJ0w", self.p.NE * self.p.NQ)

    @property
    def massD(self):
        return self._view("lgo_massD", self.p.NE * self.p.NQ)

    @property
    def diagV(self):
        return self._view("lgo_diagV", self.p.N)

    @property
    def h0(self):
        return self.L.lgo_get_h0(self.h)

    # -- reference API ------------------------------------------------------------
    def reset_time_step_estimate(self):
        self.L.lgo_set_dt_est(self.h, ctypes.c_double(np.inf))

    def update_quadrature_data(self, S):
        """laghos_solver.cpp:807-814: no-op while the quadrature data is current."""
        if self.qdata_is_current:
            return
        self.qdata_is_current = True
        self.L.lgo_qupdate(self.h, _dp(S))

    def reset_quadrature_data(self):
        self.qdata_is_current = False

    def get_time_step_estimate(self, S):
        """laghos_solver.cpp:527-535."""
        self.update_quadrature_data(S)
        dt = self.L.lgo_get_dt_est(self.h)
        if self.comm is not None:
            dt = self.comm.allreduce_min(dt)
        return dt

    def mult(self, S, dS):
        # SolveVelocity's UpdateQuadratureData (:332) honours qdata_is_current (:809)
        self.update_quadrature_data(S)
        if self.source_type == 2:  # gravity: the split calls carry the acceleration source
            h1v = self.p.H1V
            dS[:h1v] = S[h1v:2 * h1v]
            self.solve_velocity(S, dS)
            self.solve_energy(S, S[h1v:2 * h1v], dS)
            self.qdata_is_current = False
            return
        src = None
        if self.source_type == 1:
            src = np.empty(self.p.L2V)
            self.L.lgo_tg_source_2d(self.h, _dp(S), _dp(src))
        self.L.lgo_hydro_mult(self.h, _dp(S), _dp(dS), ctypes.c_double(self.cg_tol),
                              self.cg_max_iter, _dp(src) if src is not None else None, 1)
        self.qdata_is_current = False  # :326

    def solve_velocity(self, S, dS):
        self.update_quadrature_data(S)
        if self.source_type == 2:
            self.L.lgo_solve_velocity_src(self.h, _dp(S), _dp(dS), ctypes.c_double(self.cg_tol), self.cg_max_iter, 1,
                                          _dp(self._accel))
            return
        self.L.lgo_solve_velocity(self.h, _dp(S), _dp(dS), ctypes.c_double(self.cg_tol), self.cg_max_iter, 1)

    def solve_energy(self, S, V, dS):
        self.update_quadrature_data(S)
        src = None
        if self.source_type == 1:
            src = np.empty(self.p.L2V)
            self.L.lgo_tg_source_2d(self.h, _dp(S), _dp(src))
        V = np.ascontiguousarray(V)
        self.L.lgo_solve_energy(self.h, _dp(V), _dp(dS), ctypes.c_double(self.cg_tol), self.cg_max_iter,
                                _dp(src) if src is not None else None)

    def force_mult(self, x_l2):
        y = np.empty(self.p.H1V)
        self.L.lgo_force_mult(self.h, _dp(x_l2), _dp(y))
        return y

    def force_mult_transpose(self, v):
        y = np.empty(self.p.L2V)
        self.L.lgo_force_mult_transpose(self.h, _dp(v), _dp(y))
        return y

    def mass_mult(self, space, x, comp=-1, full=False):
        y = np.empty_like(x)
        self.L.lgo_mass_set_ess(self.h, comp)
        self.L.lgo_mass_mult(self.h, space, int(full), _dp(x), _dp(y))
        return y

    def cg(self, space, b, x=None, comp=-1, rel_tol=None, max_iter=None):
        x = np.zeros_like(b) if x is None else x
        self.L.lgo_mass_set_ess(self.h, comp)
        it = self.L.lgo_cg(self.h, space, _dp(b), _dp(x),
                           ctypes.c_double(self.cg_tol if rel_tol is None else rel_tol),
                           self.cg_max_iter if max_iter is None else max_iter)
        return x, it

    def cg_state(self, space=0):
        """(r, d, [nom, den, alpha, betanom]) of the recurrence as the last cg() call left it: the residual, the direction
        of the last iteration performed, and that iteration's scalars (lgo_cg_vec / lgo_cg_scalars)."""
        n = self.p.N if space == 0 else self.p.L2V
        self.L.lgo_cg_vec.restype = ctypes.POINTER(ctypes.c_double)
        self.L.lgo_cg_vec.argtypes = [ctypes.c_void_p, ctypes.c_int]
        r = np.ctypeslib.as_array(self.L.lgo_cg_vec(self.h, 0), shape=(n,)).copy()
        d = np.ctypeslib.as_array(self.L.lgo_cg_vec(self.h, 2), shape=(n,)).copy()
        sc = np.zeros(4)
        self.L.lgo_cg_scalars(self.h, _dp(sc))
        return r, d, sc

    def internal_energy(self, S):
        e = np.ascontiguousarray(S[2 * self.p.H1V:])
        return self.L.lgo_internal_energy(self.h, _dp(e))

    def kinetic_energy(self, S):
        v = np.ascontiguousarray(S[self.p.H1V:2 * self.p.H1V])
        return self.L.lgo_kinetic_energy(self.h, _dp(v))

    def timers(self):
        t = (ctypes.c_double * 4)()
        c = (ctypes.c_long * 3)()
        self.L.lgo_get_timers(self.h, t, c)
        return dict(cgH1=t[0], cgL2=t[1], force=t[2], qdata=t[3], H1iter=c[0], L2iter=c[1],
                    quad_tstep=c[2])

    def reset_timers(self):
        self.L.lgo_reset_timers(self.h)

    def e_norm(self, S):
        e = S[2 * self.p.H1V:]
        n2 = float(np.dot(e, e))
        if self.comm is not None:
            n2 = self.comm.allreduce_sum(n2)
        return np.sqrt(n2)


def rk4_step(hydro, S, t, dt, work):
    """Classical RK4 exactly as upstream RK4Solver::Step."""
    k, y, z = work
    hydro.mult(S, k)
    np.add(S, (dt / 2) * k, out=y)
    np.add(S, (dt / 6) * k, out=z)
    hydro.mult(y, k)
    np.add(S, (dt / 2) * k, out=y)
    z += (dt / 3) * k
    hydro.mult(y, k)
    np.add(S, dt * k, out=y)
    z += (dt / 3) * k
    hydro.mult(y, k)
    np.add(z, (dt / 6) * k, out=S)
    return t + dt


def rk1_step(hydro, S, t, dt, work):
    """upstream ForwardEulerSolver::Step"""
    k = work[0]
    hydro.mult(S, k)
    S += dt * k
    return t + dt


def rk2_step(hydro, S, t, dt, work, a=0.5):
    """upstream RK2Solver(a)::Step (laghos.cpp:522 uses a = 0.5)"""
    k, x1, _ = work
    b = 0.5 / a
    hydro.mult(S, k)
    np.add(S, ((1.0 - b) * dt) * k, out=x1)
    S += (a * dt) * k
    hydro.mult(S, k)
    np.add(x1, (b * dt) * k, out=S)
    return t + dt


def rk3ssp_step(hydro, S, t, dt, work):
    """upstream RK3SSPSolver::Step (laghos.cpp:523)"""
    k, y, _ = work
    hydro.mult(S, k)
    np.add(S, dt * k, out=y)
    hydro.mult(y, k)
    y += dt * k
    y[:] = (3.0 / 4) * S + (1.0 / 4) * y
    hydro.mult(y, k)
    y += dt * k
    S[:] = (1.0 / 3) * S + (2.0 / 3) * y
    return t + dt


# upstream RK6Solver (laghos.cpp:525): Verner's 8-stage 6th-order method, upstream's coefficients; the
# order conditions through order 6 hold to 1e-30 (tests/test_host_setup.py)
RK6_A = [
    .6e-1,
    .1923996296296296296296296296296296296296e-1, .7669337037037037037037037037037037037037e-1,
    .35975e-1, 0., .107925,
    1.318683415233148260919747276431735612861, 0., -5.042058063628562225427761634715637693344,
    4.220674648395413964508014358284402080483,
    -41.87259166432751461803757780644346812905, 0., 159.4325621631374917700365669070346830453,
    -122.1192135650100309202516203389242140663, 5.531743066200053768252631238332999150076,
    -54.43015693531650433250642051294142461271, 0., 207.0672513650184644273657173866509835987,
    -158.6108137845899991828742424365058599469, 6.991816585950242321992597280791793907096,
    -.1859723106220323397765171799549294623692e-1,
    -54.66374178728197680241215648050386959351, 0., 207.9528062553893734515824816699834244238,
    -159.2889574744995071508959805871426654216, 7.018743740796944434698170760964252490817,
    -.1833878590504572306472782005141738268361e-1, -.5119484997882099077875432497245168395840e-3]
RK6_B = [
    .3438957868357036009278820124728322386520e-1, 0., 0., .2582624555633503404659558098586120858767,
    .4209371189673537150642551514069801967032, 4.405396469669310170148836816197095664891,
    -176.4831190242986576151740942499002125029, 172.3641334014150730294022582711902413315]


def rk6_step(hydro, S, t, dt, work):
    """upstream ExplicitRKSolver::Step with the RK6Solver tableau (laghos.cpp:525): x_i = x + dt sum_j a_ij k_j
    accumulated term by term in upstream's order, then x += dt b_i k_i in order."""
    k = [np.empty_like(S) for _ in range(8)]
    y = work[1]
    hydro.mult(S, k[0])
    l = 0
    for i in range(1, 8):
        np.multiply(k[0], RK6_A[l] * dt, out=y)
        y += S          # add(x, a dt, k0, y): y = x + (a dt) k0
        l += 1
        for j in range(1, i):
            y += (RK6_A[l] * dt) * k[j]
            l += 1
        hydro.mult(y, k[i])
    for i in range(8):
        S += (RK6_B[i] * dt) * k[i]
    return t + dt


def rk2avg_step(hydro, S, t, dt, work):
    """RK2AvgSolver::Step (laghos_solver.cpp:1447-1487)."""
    dS, S0, _ = work
    h1v = hydro.p.H1V
    S0[:] = S
    v0 = S0[h1v:2 * h1v]
    dS[:] = 0.0  # dS_dt = 0 at Init; every block is overwritten below
    # -- 1. S is S0
    hydro.solve_velocity(S, dS)
    V = v0 + (0.5 * dt) * dS[h1v:2 * h1v]
    hydro.solve_energy(S, V, dS)
    dS[:h1v] = V
    # -- 2. S = S0 + 0.5 dt dS_dt
    np.add(S0, (0.5 * dt) * dS, out=S)
    hydro.reset_quadrature_data()
    hydro.solve_velocity(S, dS)
    V = v0 + (0.5 * dt) * dS[h1v:2 * h1v]
    hydro.solve_energy(S, V, dS)
    dS[:h1v] = V
    # -- 3. S = S0 + dt dS_dt
    np.add(S0, dt * dS, out=S)
    hydro.reset_quadrature_data()
    return t + dt


def run(prob: Problem, t_final=0.6, cfl=0.5, cg_tol=1e-8, cg_max_iter=300, max_steps=-1,
        vis_steps=5, probe_steps=(), verbose=False, comm=None, hydro=None, ode_solver=4):
    """The reference time loop (laghos.cpp:706-778).  Returns a dict with the last
    printed (step, t, dt, |e|), |e| at each step in probe_steps, and the FOM data."""
    own = hydro is None
    if own:
        hydro = Hydro(prob, cfl=cfl, cg_tol=cg_tol, cg_max_iter=cg_max_iter, comm=comm)
    S = hydro.S0.copy()
    work = (np.empty_like(S), np.empty_like(S), np.empty_like(S))
    hydro.reset_time_step_estimate()
    t = 0.0
    dt = hydro.get_time_step_estimate(S)
    last_step = False
    steps = 0
    S_old = S.copy()
    probes = {}
    last = None
    repeats = 0
    e_init = hydro.internal_energy(S) + hydro.kinetic_energy(S)
    ti = 1
    while not last_step:
        if t + dt >= t_final:
            dt = t_final - t
            last_step = True
        if steps == max_steps:
            last_step = True
        S_old[:] = S
        t_old = t
        hydro.reset_time_step_estimate()
        t = {1: rk1_step, 2: rk2_step, 3: rk3ssp_step, 4: rk4_step, 6: rk6_step, 7: rk2avg_step}[ode_solver](hydro, S, t, dt, work)
        steps += 1
        dt_est = hydro.get_time_step_estimate(S)
        if dt_est < dt:
            dt *= 0.85
            if dt < np.finfo(float).eps:
                raise RuntimeError("The time step crashed!")
            t = t_old
            S[:] = S_old
            hydro.reset_quadrature_data()
            repeats += 1
            if verbose:
                print(f"Repeating step {ti}")
            if steps < max_steps:
                last_step = False
            continue  # ti unchanged (ti--; continue; ti++)
        elif dt_est > 1.25 * dt:
            dt *= 1.02
        if last_step or (ti % vis_steps) == 0:
            en = hydro.e_norm(S)
            last = dict(step=ti, t=t, dt=dt, e_norm=en)
            if verbose:
                print(f"step {ti:5d},\tt = {t:.4f},\tdt = {dt:.6f},\t|e| = {en:.10e}")
                sys.stdout.flush()
        if ti in probe_steps:
            probes[ti] = hydro.e_norm(S)
        ti += 1
    e_final = hydro.internal_energy(S) + hydro.kinetic_energy(S)
    out = dict(last=last, probes=probes, steps=steps, repeats=repeats, S=S,
               energy_diff=abs(e_init - e_final), timers=hydro.timers())
    if own:
        hydro.close()
    return out

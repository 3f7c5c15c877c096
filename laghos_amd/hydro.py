"""Python mirror of LagrangianHydroOperator + RK4 + the adaptive time loop over
the C ABI (reference: /root/reference/laghos_solver.cpp:104-540, laghos.cpp:706-778).

All numerics run in liblaghos_hip.so on the GPU; this file only sequences C-ABI
calls (the same sequencing the C++ host layer in laghos_amd/host/ performs).
The problem description `prob` is duck-typed: it must provide dim, NE, D1D, Q1D,
L1D, N, H1V, L2V, h1map, B, G, Bl, W, ess, owner, order_v, use_viscosity(),
initial_state() -> (S, rho0_l2, gamma, rho0_q).
"""
import numpy as np
import torch

from .context import Context

H1, L2 = 0, 1


class HydroOperator:
    def __init__(self, prob, cfl=0.5, cg_tol=1e-8, cg_max_iter=300, device=0, comm=None):
        self.p = prob
        self.cg_tol, self.cg_max_iter = cg_tol, cg_max_iter
        S, rho_l2, gamma, rho0_q = prob.initial_state()
        multi = comm is not None and comm["nranks"] > 1
        self.ctx = ctx = Context(prob.dim, prob.NE, prob.D1D, prob.Q1D, prob.L1D, prob.N, prob.h1map,
                                 prob.B, prob.G, prob.Bl, prob.W, gamma, prob.ess,
                                 owner=prob.owner if multi else None,
                                 use_viscosity=prob.use_viscosity(),
                                 use_vorticity=getattr(prob, "use_vorticity", lambda: False)(),
                                 cfl=cfl, order_v=prob.order_v,
                                 device=device)
        self.multi = multi
        if multi:
            ctx.comm_init(comm["nranks"], comm["rank"], comm["unique_id"])
            ctx.comm_set_neighbors(comm["nbr_rank"], comm["nbr_nodes"])
        self.S0 = ctx.to_dev(S)
        x0 = ctx.clone(self.S0[:prob.H1V])
        vol = ctx.setup_rho0detj0(x0, ctx.to_dev(rho_l2), ctx.to_dev(rho0_q))
        ne = float(prob.NE)
        if multi:
            vol = ctx.allreduce(vol, 0)
            ne = ctx.allreduce(ne, 0)
        self.volume = vol
        self.h0 = (vol / ne) ** (1.0 / prob.dim) / prob.order_v   # laghos_solver.cpp:251-262
        ctx.set_h0(self.h0)
        # scratch (laghos_solver.cpp:158-164): one, rhs, e_rhs, B
        self.one = torch.ones(prob.L2V, dtype=torch.float64, device=ctx.device)
        torch.cuda.current_stream(ctx.device).synchronize()  # torch-stream fill complete before the library reads it
        self.rhs = ctx.zeros(prob.H1V)
        self.e_rhs = ctx.zeros(prob.L2V)
        self.work = ctx.zeros(prob.N)
        # source_type 1 = 2D Taylor-Green (laghos.cpp:636-647, laghos_solver.cpp:448)
        self.e_source = ctx.zeros(prob.L2V) if (prob.problem == 0 and prob.dim == 2) else None
        # source_type 2 = gravity of problem 7 (laghos.cpp:645, laghos_solver.cpp:340-347)
        self.accel = ctx.to_dev(prob.accel_source()) if prob.problem == 7 else None
        if self.accel is not None:
            ctx.set_velocity_source(self.accel)
        self.qdata_is_current = False
        torch.cuda.synchronize()

    def close(self):
        self.ctx.close()

    # ---- `-err` post-processing (laghos.cpp:1007-1086) ----
    def compute_density(self, S):
        """LagrangianHydroOperator::ComputeDensity (laghos_solver.cpp:542-563)"""
        rho = self.ctx.zeros(self.p.L2V)
        self.ctx.compute_density(S, rho)
        return rho

    def sedov_density_error(self, S, rho, par, t, origin, weights, B_h1, G_h1, B_l2):
        """sqrt of the integrated squared density error against the exact Sedov solution `par`
        (context.sedov_setup); the error rule comes as host tables [p, d] (transposed here to the
        library's [p + n*d] layout)."""
        import numpy as np
        err2 = self.ctx.sedov_density_error(S, rho, par, t, origin, weights, np.asarray(B_h1).T.copy(),
                                            np.asarray(G_h1).T.copy(), np.asarray(B_l2).T.copy())
        return float(np.sqrt(err2))

    def reset_time_step_estimate(self):
        self.ctx.set_dt_est(float("inf"))

    def reset_quadrature_data(self):
        self.qdata_is_current = False
        self.ctx.reset_quadrature_data()  # the force products formed with the stale data go with it

    def update_quadrature_data(self, S):
        if self.qdata_is_current:
            return
        self.qdata_is_current = True
        self.ctx.qupdate(S)

    def get_time_step_estimate(self, S):
        self.update_quadrature_data(S)
        dt = self.ctx.get_dt_est()
        if self.multi:
            dt = self.ctx.allreduce(dt, 1)
        return dt

    def mult(self, S, dS):
        p, ctx = self.p, self.ctx
        ctx.vec_copy(dS[:p.H1V], S[p.H1V:2 * p.H1V])             # dx_dt = v
        self.update_quadrature_data(S)
        # SolveEnergy takes v from S, not from SolveVelocity: the library overlaps the two
        if self.e_source is not None:
            ctx.tg_source_2d(S, self.e_source)
        ctx.solve_energy_begin(S, S[p.H1V:2 * p.H1V], dS, self.e_rhs, self.cg_tol, self.cg_max_iter,
                               e_source=self.e_source)
        ctx.solve_velocity(S, dS, None, self.rhs, self.work, self.cg_tol, self.cg_max_iter)  # (one: the operator's own)
        ctx.solve_energy_end()
        self.qdata_is_current = False

    def e_norm(self, S):
        e = S[2 * self.p.H1V:]
        n2 = self.ctx.vec_dot(e, e)
        if self.multi:
            n2 = self.ctx.allreduce(n2, 0)
        return float(np.sqrt(n2))


def rk4_step(hydro, S, t, dt, work):
    """Classical RK4, the update sequence of upstream RK4Solver::Step."""
    ctx = hydro.ctx
    k, y, z = work
    hydro.mult(S, k)
    ctx.vec_axpby(y, 1.0, S, dt / 2, k)
    ctx.vec_axpby(z, 1.0, S, dt / 6, k)
    hydro.mult(y, k)
    ctx.vec_axpby(y, 1.0, S, dt / 2, k)
    ctx.vec_axpby(z, 1.0, z, dt / 3, k)
    hydro.mult(y, k)
    ctx.vec_axpby(y, 1.0, S, dt, k)
    ctx.vec_axpby(z, 1.0, z, dt / 3, k)
    hydro.mult(y, k)
    ctx.vec_axpby(S, 1.0, z, dt / 6, k)
    return t + dt


def rk1_step(hydro, S, t, dt, work):
    """upstream ForwardEulerSolver::Step"""
    k = work[0]
    hydro.mult(S, k)
    hydro.ctx.vec_axpby(S, 1.0, S, dt, k)
    return t + dt


def rk2_step(hydro, S, t, dt, work, a=0.5):
    """upstream RK2Solver(a)::Step; Laghos uses a = 0.5, the midpoint rule (laghos.cpp:522)"""
    ctx = hydro.ctx
    k, x1, _ = work
    b = 0.5 / a
    hydro.mult(S, k)
    ctx.vec_axpby(x1, 1.0, S, (1.0 - b) * dt, k)
    ctx.vec_axpby(S, 1.0, S, a * dt, k)
    hydro.mult(S, k)
    ctx.vec_axpby(S, 1.0, x1, b * dt, k)
    return t + dt


def rk3ssp_step(hydro, S, t, dt, work):
    """upstream RK3SSPSolver::Step (laghos.cpp:523)"""
    ctx = hydro.ctx
    k, y, _ = work
    hydro.mult(S, k)
    ctx.vec_axpby(y, 1.0, S, dt, k)
    hydro.mult(y, k)
    ctx.vec_axpby(y, 1.0, y, dt, k)
    ctx.vec_axpby(y, 3.0 / 4, S, 1.0 / 4, y)
    hydro.mult(y, k)
    ctx.vec_axpby(y, 1.0, y, dt, k)
    ctx.vec_axpby(S, 1.0 / 3, S, 2.0 / 3, y)
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
    """upstream ExplicitRKSolver::Step with the RK6Solver tableau (same order of operations)"""
    ctx = hydro.ctx
    if getattr(hydro, "_rk6_k", None) is None or hydro._rk6_k[0].numel() != S.numel():
        hydro._rk6_k = [torch.empty_like(S) for _ in range(8)]
        torch.cuda.synchronize()
    k, y = hydro._rk6_k, work[1]
    hydro.mult(S, k[0])
    l = 0
    for i in range(1, 8):
        ctx.vec_axpby(y, 1.0, S, RK6_A[l] * dt, k[0])
        l += 1
        for j in range(1, i):
            ctx.vec_axpby(y, 1.0, y, RK6_A[l] * dt, k[j])
            l += 1
        hydro.mult(y, k[i])
    for i in range(8):
        ctx.vec_axpby(S, 1.0, S, RK6_B[i] * dt, k[i])
    return t + dt


def rk2avg_step(hydro, S, t, dt, work):
    """RK2AvgSolver::Step (laghos_solver.cpp:1447-1487): two SolveVelocity /
    SolveEnergy sub-steps, the energy one with the half-step average velocity V."""
    ctx, p = hydro.ctx, hydro.p
    dS, S0, y = work
    h1v = p.H1V
    V = y[:h1v]
    ctx.vec_copy(S0, S)
    v0, dv = S0[h1v:2 * h1v], dS[h1v:2 * h1v]
    for sub in (0, 1):
        if sub == 1:
            ctx.vec_axpby(S, 1.0, S0, 0.5 * dt, dS)                # S = S0 + dt/2 dS_dt
            hydro.reset_quadrature_data()
        hydro.update_quadrature_data(S)
        ctx.solve_velocity(S, dS, None, hydro.rhs, hydro.work, hydro.cg_tol, hydro.cg_max_iter)
        ctx.vec_axpby(V, 1.0, v0, 0.5 * dt, dv)                   # V = v0 + dt/2 dv_dt
        if hydro.e_source is not None:
            ctx.tg_source_2d(S, hydro.e_source)
        ctx.solve_energy(S, V, dS, hydro.e_rhs, hydro.cg_tol, hydro.cg_max_iter, e_source=hydro.e_source)
        ctx.vec_copy(dS[:h1v], V)                                  # dx_dt = V
    ctx.vec_axpby(S, 1.0, S0, dt, dS)                              # S = S0 + dt dS_dt
    hydro.reset_quadrature_data()
    return t + dt


class TimeLoop:
    """laghos.cpp:706-778 as a resumable object (bench.py steps it K times)."""

    def __init__(self, hydro, t_final=0.6, max_steps=-1, ode_solver=4):
        steppers = {1: rk1_step, 2: rk2_step, 3: rk3ssp_step, 4: rk4_step, 6: rk6_step, 7: rk2avg_step}
        if ode_solver not in steppers:
            raise ValueError("ode_solver: 1 (Euler), 2 (RK2), 3 (RK3 SSP), 4 (RK4), 6 (RK6) or 7 (RK2Avg)")
        self.stepper = steppers[ode_solver]
        self.h = hydro
        self.t_final, self.max_steps = t_final, max_steps
        self.S = hydro.S0.clone()
        self.S_old = self.S.clone()
        self.work = tuple(torch.empty_like(self.S) for _ in range(3))
        torch.cuda.synchronize()  # torch-stream clones visible to the context stream
        hydro.reset_time_step_estimate()
        self.t = 0.0
        self.dt = hydro.get_time_step_estimate(self.S)
        self.ti = 1
        self.steps = 0
        self.repeats = 0
        self.last_step = False

    def step(self):
        """Advance one accepted time step (repeating with smaller dt as needed).
        Returns False when the run is finished."""
        h = self.h
        while True:
            if self.last_step:
                return False
            if self.t + self.dt >= self.t_final:
                self.dt = self.t_final - self.t
                self.last_step = True
            if self.steps == self.max_steps:
                self.last_step = True
            h.ctx.vec_copy(self.S_old, self.S)
            t_old = self.t
            h.reset_time_step_estimate()
            self.t = self.stepper(h, self.S, self.t, self.dt, self.work)
            self.steps += 1
            dt_est = h.get_time_step_estimate(self.S)
            if dt_est < self.dt:
                self.dt *= 0.85
                if self.dt < np.finfo(float).eps:
                    raise RuntimeError("The time step crashed!")
                self.t = t_old
                h.ctx.vec_copy(self.S, self.S_old)
                h.reset_quadrature_data()
                self.repeats += 1
                if self.steps < self.max_steps:
                    self.last_step = False
                continue
            elif dt_est > 1.25 * self.dt:
                self.dt *= 1.02
            self.ti += 1
            return True


def run(prob, t_final=0.6, cfl=0.5, cg_tol=1e-8, cg_max_iter=300, max_steps=-1, probe_steps=(),
        device=0, comm=None, ode_solver=4, timers=True):
    hydro = HydroOperator(prob, cfl=cfl, cg_tol=cg_tol, cg_max_iter=cg_max_iter, device=device, comm=comm)
    hydro.ctx.enable_timers(timers)  # off: the energy solve overlaps the velocity solve (lgh_solve_energy_begin)
    loop = TimeLoop(hydro, t_final=t_final, max_steps=max_steps, ode_solver=ode_solver)
    probes = {}
    while loop.step():
        done_ti = loop.ti - 1
        if done_ti in probe_steps:
            probes[done_ti] = hydro.e_norm(loop.S)
    out = dict(probes=probes, steps=loop.steps, repeats=loop.repeats, ti=loop.ti - 1, t=loop.t,
               dt=loop.dt, e_norm=hydro.e_norm(loop.S), S=loop.S.cpu().numpy(), timers=hydro.ctx.timers())
    hydro.close()
    return out

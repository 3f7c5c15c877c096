"""Developer probe (not a test, not the bench): times the HIP path on the full
3D Sedov Q3Q2 configuration and prints per-region timers.  Lives under tests/
because it borrows the oracle's numpy problem setup."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--problem", type=int, default=1)
    ap.add_argument("--ok", type=int, default=3)
    ap.add_argument("--ot", type=int, default=2)
    ap.add_argument("--no-timers", action="store_true")
    a = ap.parse_args()
    import torch
    from laghos_amd.hydro import HydroOperator, TimeLoop
    from oracle.fem import Problem
    t0 = time.time()
    prob = Problem(mesh="cube01_hex", rs=a.rs, order_v=a.ok, order_e=a.ot, problem=a.problem)
    g = HydroOperator(prob)
    print(f"setup {time.time()-t0:.1f}s NE={prob.NE} N={prob.N} L2V={prob.L2V}", flush=True)
    loop = TimeLoop(g, t_final=1e9, max_steps=a.warmup + a.steps)
    for _ in range(a.warmup):
        loop.step()
    g.ctx.sync()
    g.ctx.reset_timers()
    if a.no_timers:
        g.ctx.enable_timers(False)
    t0 = time.time()
    s0 = loop.steps
    for _ in range(a.steps):
        loop.step()
    g.ctx.sync()
    wall = time.time() - t0
    rk_steps = loop.steps - s0
    tm = g.ctx.timers()
    dofs = prob.dim * prob.N + prob.L2V
    out = dict(wall_s=wall, accepted_steps=a.steps, rk4_steps=rk_steps, ms_per_rk4=1e3 * wall / rk_steps,
               mdofs_steps_per_s=1e-6 * dofs * 4 * rk_steps / wall, timers=tm, dt=loop.dt, t=loop.t,
               e_norm=g.e_norm(loop.S))
    if not a.no_timers and tm["cgH1"] > 0:
        T1, T2, T3 = tm["cgH1"], tm["force"], tm["qdata"]
        out["FOM1"] = 1e-6 * prob.N * (tm["H1iter"] / prob.dim) / T1
        out["FOM2"] = 1e-6 * 4 * rk_steps * dofs / T2
        out["FOM3"] = 1e-6 * tm["quad_tstep"] * prob.NQ / T3
        out["H1_iters_per_solve"] = tm["H1iter"] / (4 * rk_steps * prob.dim)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

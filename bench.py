#!/usr/bin/env python3
"""bench.py — throughput of the MI355X Laghos partial-assembly hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" is one accepted RK4 time step of 3D Sedov Q3/Q2 (-p 1 -ok 3 -ot 2 -pa):
5 quadrature-data updates, 4 Force Mult, 4 Force MultTranspose, 12 H1 PCG solves
and 4 L2 CG solves, all in liblaghos_hip.so through the C++ host layer.  At N=1
the workload is BASELINE.json configs[1] (cube01_hex -rs 4: 32^3 elements); for
N>1 every rank keeps a 32^3 block (weak scaling, element-sharded, RCCL for the
shared-node sums and the CG dot products), N=8 is configs[3] (64^3 elements).

Prints ONE JSON line on rank 0.  `value` = 1e-6 * (H1 + L2 global dofs) * RK
stages executed / wall time of the K timed steps (the reference's FOM0 with the
whole-step wall time in the denominator; laghos_solver.cpp:727, laghos.cpp:928-935).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL across processes needs this (already exported on the
# GPU boxes; kept for environments built by hand)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: what a float4 copy kernel reaches (79 % of the spec)


def flush_c_stdio():
    """RCCL writes its version banner through C stdio; push it out so that it cannot land behind
    the JSON line when stdout is a pipe."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def partition_grid(ne, n):
    """Python mirror of laghos::Partition (laghos_amd/host/fem.cpp): split the axis with the most local
    zones that is divisible by the smallest prime factor of what is left.  None: not evenly divisible.
    tests/test_bench_contract.py holds it against the C++ code."""
    loc, pg, left = list(ne), [1, 1, 1], n
    while left > 1:
        f = 2
        while left % f:
            f += 1
        best = -1
        for a in range(3):
            if loc[a] % f == 0 and (best < 0 or loc[a] > loc[best]):
                best = a
        if best < 0:
            return None
        loc[best] //= f
        pg[best] *= f
        left //= f
    return tuple(pg)


def block_grid(n, block=32):
    """ranks -> (px,py,pz): the most cubic grid of `block`^3-zone blocks that laghos::Partition itself
    produces for the (block*px, block*py, block*pz) mesh - so that the process grid and the per-GPU
    workload in the bench line are what the library really runs (N=6: 1x6x1, not 2x3x1)."""
    cands = []
    for px in range(1, n + 1):
        if n % px:
            continue
        for py in range(1, n // px + 1):
            if (n // px) % py:
                continue
            pz = n // (px * py)
            p = (px, py, pz)
            if partition_grid((block * px, block * py, block * pz), n) == p:
                cands.append((max(p) / min(p), -px, -py, p))
    if not cands:
        raise SystemExit("no block grid for %d ranks" % n)
    return list(sorted(cands)[0][3])


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _time_oracle(threads, native, budget_s, want_ref):
    """RK4 steps of the bench's own problem from t = 0 with the restated CPU path until `budget_s` of wall time are spent"""
    import numpy as np
    from oracle.driver import Hydro, lib, rk4_step
    from oracle.fem import Problem
    lib(native).lgo_set_num_threads(threads)
    prob = Problem(mesh="cube01_hex", rs=4, order_v=3, order_e=2, problem=1)
    h = Hydro(prob, native=native)
    S = h.S0.copy()
    work = (np.empty_like(S), np.empty_like(S), np.empty_like(S))
    h.reset_time_step_estimate()
    dt = h.get_time_step_estimate(S)
    steps, wall, t = 0, 0.0, 0.0
    ref = None
    while (wall < budget_s or (want_ref and steps < PARITY_STEPS)) and steps < 40:
        t0 = time.time()
        h.reset_time_step_estimate()
        t = rk4_step(h, S, t, dt, work)
        dt_est = h.get_time_step_estimate(S)
        wall += time.time() - t0
        steps += 1
        if dt_est < dt and ref is None:
            ref = {"error": "the oracle's step %d would be repeated with a smaller dt: no parity sample" % steps}
        if dt_est > 1.25 * dt:
            dt *= 1.02
        if want_ref and steps == PARITY_STEPS and ref is None:
            # the checker's state after PARITY_STEPS whole RK4 steps of the bench problem: main() holds the HIP path's against it
            ref = {"steps": steps, "t": t, "dt": dt, "e_norm": h.e_norm(S), "S": S.copy(), "H1V": prob.H1V}
    dofs = prob.dim * prob.N + prob.L2V
    tm = h.timers()
    h.close()
    return dict(value=1e-6 * dofs * 4 * steps / wall, seconds=wall, rk4_steps=steps, h1_cg_iters=tm["H1iter"]), ref


def cpu_baseline(threads):
    """The oracle (CPU restatement of the reference -pa path) timed on the host cores on a bounded sample of the same 32^3
    Q3/Q2 Sedov workload: whole RK4 steps from t = 0 with the real dt controller inputs (each step = 4 stages of 1 QUpdate,
    1 Force, 1 ForceT, 3 H1 PCG, 1 L2 CG, + the dt-estimate QUpdate).  Two builds of the same source (round-5 verdict, item 6):
    `value` is the TIMING build SURVEY 8(d) asks for (-O3 -march=native, FMA contraction; compiled on this very host by
    `make -C oracle native`), `value_parity_build` the checker itself (-march=x86-64-v3 -ffp-contract=off: the build every
    parity test and the `parity` block of this line are held against).  Reported next to the GPU number; never shipped."""
    from oracle.driver import NATIVE_FLAGS, PARITY_FLAGS
    par, ref = _time_oracle(threads, False, 8.0, True)
    out = dict(unit="Mdofs*steps/s", cores=threads, cpu_model=cpu_model(), kind="port",
               value_parity_build=par["value"], flags_parity_build=PARITY_FLAGS)
    try:
        # (in a process of its own: a timing build that dies on this host - illegal instruction, a compiler's bad day - must not
        #  take the bench line with it; round 6 met exactly that with 512-bit auto-vectorisation, oracle/Makefile)
        import subprocess
        code = ("import json, sys; sys.path.insert(0, %r); import bench; r, _ = bench._time_oracle(%d, True, 8.0, False); "
                "print('NATIVE ' + json.dumps(r))" % (ROOT, threads))
        pr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
        lines = [l for l in pr.stdout.splitlines() if l.startswith("NATIVE ")]
        if pr.returncode != 0 or not lines:
            raise RuntimeError("timing build exited with %d: %s" % (pr.returncode, (pr.stderr or "").strip()[-120:]))
        nat = json.loads(lines[-1][len("NATIVE "):])
        out.update(value=nat["value"], flags=NATIVE_FLAGS, seconds=nat["seconds"] + par["seconds"], rk4_steps=nat["rk4_steps"], h1_cg_iters=nat["h1_cg_iters"])
        note = "%d steps in %.1f s with the -march=native build, %d in %.1f s with the parity build" % (nat["rk4_steps"], nat["seconds"], par["rk4_steps"], par["seconds"])
    except Exception as e:  # no compiler on this host: the parity build's time stands in, and the line says so
        out.update(value=par["value"], flags=PARITY_FLAGS, seconds=par["seconds"], rk4_steps=par["rk4_steps"], h1_cg_iters=par["h1_cg_iters"],
                   native_build_error=repr(e)[:200])
        note = "%d steps in %.1f s with the parity build (the -march=native build could not be made here)" % (par["rk4_steps"], par["seconds"])
    out["sample"] = "RK4 steps from t=0 of the same 3D Sedov Q3Q2 32^3 problem (oracle/ C++ kernels, OpenMP %d threads): %s" % (threads, note)
    return out, ref


PARITY_STEPS = 3


def parity_block(host_lib, args, dev, ref):
    """`parity` of the bench line: PARITY_STEPS RK4 steps of the bench's own problem from t = 0 through the HIP path
    (a fresh simulation, outside every timed region) against the oracle's state after the same steps (the ones the
    cpu_baseline leg ran anyway).  tests/test_gpu_pipeline.py::test_config2_full_size_vs_oracle asserts the same
    comparison; here it is printed with the number it belongs to."""
    import numpy as np
    if ref is None or "error" in ref:
        return {"error": (ref or {}).get("error", "no oracle sample")}
    sim = host_lib.Sim(args + ["-dev", dev, "-q"])
    sim.enable_timers(False)
    for _ in range(ref["steps"]):
        sim.step()
    sim.sync()
    out = {"checker": "oracle/ (CPU restatement of the reference -pa path, pinned to the reference's --checks table and README runs)",
           "rk4_steps": ref["steps"], "rk4_steps_executed_hip": sim.rk_steps, "t_hip": sim.t, "t_oracle": ref["t"],
           "dt_hip": sim.dt, "dt_oracle": ref["dt"], "e_norm_hip": sim.e_norm(), "e_norm_oracle": ref["e_norm"]}
    S, So, H1V = sim.state(), ref["S"], ref["H1V"]
    sim.close()
    out["e_norm_rel_diff"] = abs(out["e_norm_hip"] - ref["e_norm"]) / abs(ref["e_norm"])
    out["dt_rel_diff"] = abs(out["dt_hip"] - ref["dt"]) / ref["dt"]
    for name, sl in (("x", slice(0, H1V)), ("v", slice(H1V, 2 * H1V)), ("e", slice(2 * H1V, None))):
        out["state_%s_max_rel_diff" % name] = float(np.abs(S[sl] - So[sl]).max() / np.abs(So[sl]).max())
    out["tolerances"] = {"e_norm": 1e-9, "state": 1e-8, "dt": 1e-9}
    out["pass"] = bool(out["rk4_steps_executed_hip"] == ref["steps"] and out["e_norm_rel_diff"] <= 1e-9 and out["dt_rel_diff"] <= 1e-9
                       and all(out["state_%s_max_rel_diff" % n] <= 1e-8 for n in "xve"))
    return out


KERNEL_NAMES = {0: "vcg_apply_plane (H1 CG K1, 3 velocity components per launch)",
                1: "vcg_update_p_k (H1 CG K2, 3 velocity components per launch)",
                2: "qpoint_kernel (fused QUpdate + both force products)", 3: "force_mult_3d", 4: "force_mult_t_3d",
                5: "mass_apply_l2 (L2 CG K1)",
                6: "halo_sum (pack + grouped ncclSend/Recv + combine)", 7: "ncclAllReduce of device scalars"}
K1_FORMS = {0: "vcg_apply_3d", 2: "vcg_apply_plane", 4: "vcg_apply_slab346", 5: "vcg_apply_kron"}


def kernel_names(L, ctx):
    """KERNEL_NAMES with the K1 form this context really launches (lgh_k1_form)"""
    f = ctypes.c_int(-2)
    L.lgh_k1_form(ctx, ctypes.byref(f))
    names = dict(KERNEL_NAMES)
    names[0] = K1_FORMS.get(f.value, "vcg_apply") + " (H1 CG K1, 3 velocity components per launch)"
    if os.environ.get("LGH_K2P") == "0":
        names[1] = names[1].replace("vcg_update_p_k", "vcg_update_k")
    q = ctypes.c_int(0)
    L.lgh_qupdate_form(ctx, ctypes.byref(q))
    if q.value == 1:
        names[2] = names[2].replace("qpoint_kernel", "qrows_kernel")
    return names


def mass_data_note(sim):
    """Which mass quadrature data the kernels read (lgh_mass_data_form): the compact form is an operator substitution
    decided by a device check (every stored entry within 1e-12 relative of W[q]*s_e), so the line names it."""
    from laghos_amd import _lib
    L = _lib.load()
    form = ctypes.c_int(-1)
    L.lgh_mass_data_form(sim.L.laghos_sim_context(sim.h), ctypes.byref(form))
    j0 = ctypes.c_int(0)
    L.lgh_jac0inv_form(sim.L.laghos_sim_context(sim.h), ctypes.byref(j0))
    return {1: "compact W[q]*s_e (device check of every stored entry, rel 1e-12; affine zones, zone-constant rho0)",
            0: "stored table D[q,e] (the reference's form)"}.get(form.value, "not decided yet (no mass apply ran)") + \
        ("; Jac0inv one per zone (zone-constant, device check, 1e-12)" if j0.value == 1 else "; Jac0inv per point")


def mesh_order_note(sim):
    """What the library found of the mesh (lgh_mesh_order): it orders zones and nodes itself from the element -> node map; on
    the generator's own numbering that order is the caller's and nothing is permuted (the legs c2mfem / c2perm measure the rest)."""
    from laghos_amd import _lib
    o = (ctypes.c_long * 8)()
    _lib.check(_lib.load().lgh_mesh_order(sim.L.laghos_sim_context(sim.h), o))
    if not o[0]:
        return "no structured block found (or LGH_ORDER=0): the caller's zone order and node numbering are kept"
    return "structured block %dx%dx%d found by face adjacency; internal zone order / node numbering %s" % (
        o[3], o[4], o[5], "= the caller's (nothing permuted)" if o[1] else "differ from the caller's (velocity solve and zone walk run in the library's own)")


def algorithmic_bytes(sz):
    """SURVEY §8(d)'s algorithmic bytes per launch, fp64: what the REFERENCE's form of each kernel has to move."""
    dim, D, Q, Ld = sz["dim"], sz["D1D"], sz["Q1D"], sz["L1D"]
    NQ, ND, NL, NE, N = sz["NQ"], D ** dim, Ld ** dim, sz["NE"], sz["N"]
    return {
        0: NE * 8 * (NQ + dim * 2 * ND),                        # lockstep mass apply: D once + (in + out) per component
        # K2, per launch (mean over a solve): r, d read and written, x read and written every second iteration (5 passes
        # per component), 1/diag, the element contributions (E-vector), their ELL index table (8 slots), the flag bytes
        1: 8 * N * (dim * 5 + 1) + 8 * dim * NE * ND + 4 * 8 * N + N,
        2: NE * (8 * (2 * dim * ND + NL + dim * dim * NQ + NQ + dim * dim * NQ) + 8),  # fused QUpdate
        3: NE * 8 * (dim * dim * NQ + NL + dim * ND),           # ForceMult
        4: NE * 8 * (dim * dim * NQ + NL + dim * ND),           # ForceMultTranspose
        5: NE * 8 * (NQ + 2 * NL),                              # L2 mass apply
    }


def moved_bytes(sz, L, ctx, k1_name):
    """Bytes per launch the kernel form that RAN has to move (its operand footprint: every array it reads or writes
    counted once) - what `GBs`, `roofline.achieved` and the aggregates are priced with, so that no figure can exceed
    what the memory system delivered.  Differs from SURVEY §8(d)'s figure (kept beside it as `sec8d_*`) where this
    implementation departs from the reference's kernel:
      K1 (lockstep mass apply): gathers r, d_old (dim components) and 1/diag from node vectors instead of reading one
         E-vector, and with compact mass data (lgh_mass_data_form) reads one double per element instead of NQ;
      L2 mass apply: compact data likewise (plane form);
    everything else moves what §8(d) says."""
    from laghos_amd import _lib
    dim, D, Ld = sz["dim"], sz["D1D"], sz["L1D"]
    NQ, ND, NL, NE, N = sz["NQ"], D ** dim, Ld ** dim, sz["NE"], sz["N"]
    b = dict(algorithmic_bytes(sz))
    form = ctypes.c_int(-1)
    L.lgh_mass_data_form(ctx, ctypes.byref(form))
    h1s, l2s = ctypes.c_int(0), ctypes.c_int(0)
    L.lgh_table_symmetry(ctx, ctypes.byref(h1s), ctypes.byref(l2s))
    k1_compact = form.value == 1 and k1_name.split(" ")[0] in ("vcg_apply_slab346", "vcg_apply_plane", "vcg_apply_plane_ho", "vcg_apply_kron")
    l2f, l2c = ctypes.c_int(0), ctypes.c_int(0)
    L.lgh_l2_mass_form(ctx, ctypes.byref(l2f), ctypes.byref(l2c))  # the kernel the library really launches (round-4 advisor)
    l2_compact = l2c.value == 1
    st = ctypes.c_int(1)
    L.lgh_qupdate_stores_stress(ctx, ctypes.byref(st))
    if st.value == 0:  # stress kept in registers: the nine stressJinvT planes are not written
        b[2] -= NE * 8 * dim * dim * NQ
    j0c, qf = ctypes.c_int(0), ctypes.c_int(0)
    L.lgh_jac0inv_form(ctx, ctypes.byref(j0c))
    L.lgh_qupdate_form(ctx, ctypes.byref(qf))
    jac0_compact = j0c.value == 1 and qf.value == 1  # (the row form reads one inverse Jacobian per zone; the point form the stored values)
    if jac0_compact:
        b[2] -= NE * 8 * dim * dim * (NQ - 1)
    # the E-vector between K1 and K2 and K2's table as THIS context lays them out (lgh_vcg_layout_stats): the slab K1 sums
    # the shared x-faces of its sets itself (merged layout: ~17 % fewer values), K2 reads 16 bytes of table per node and
    # the second 16 only in the wavefronts that need them
    st4 = (ctypes.c_long * 4)()
    _lib.check(L.lgh_vcg_layout_stats(ctx, st4))
    evec, tab = int(st4[0]), int(st4[1]) + int(st4[2])
    b[0] = 8 * (N * (2 * dim + 1) + dim * evec) + (8 * NE if k1_compact else 8 * NE * NQ)
    b[1] = 8 * N * (dim * 5 + 1) + 8 * dim * evec + tab + N
    b[5] = NE * 8 * ((1 if l2_compact else NQ) + 2 * NL)
    notes = {0: "r, d_old (dim components) and 1/diag gathered from node vectors, E-vector out; mass data: %s"
                % ("compact, one factor per element" if k1_compact else "stored table, NQ per element"),
             1: "r, d read and written, x every second iteration, 1/diag, flag bytes; E-vector %d of %d values per component (%s); "
                "transposed-restriction table %.1f of %.1f MB fetched" % (evec, NE * ND, "x-faces of a set summed by K1" if st4[3] else "element-local", 1e-6 * tab, 1e-6 * 32 * N),
             5: "mass data: %s" % ("compact, one factor per element" if l2_compact else "stored table"),
             2: "Jac0inv %s; stressJinvT %s" % ("one per zone (zone-constant: device check at set-up)" if jac0_compact else "per point", "kept in registers (both force products formed in the kernel; lgh_qupdate_store_stress(ctx, 0))" if st.value == 0 else "written (9 planes)")}
    return b, notes, k1_compact


def measure_kernels(sim, sz):
    """HIP-event timing (on the library's stream) of every launch of each hot kernel during one RK step per
    kernel, after the timed region.  Returns (per-kernel dict, aggregates, raw)."""
    from laghos_amd import _lib
    L = _lib.load()
    ctx = sim.L.laghos_sim_context(sim.h)
    KERNEL_NAMES = kernel_names(L, ctx)
    sec8d = algorithmic_bytes(sz)
    bts, notes, k1_compact = moved_bytes(sz, L, ctx, KERNEL_NAMES[0])
    kern, raw = {}, {}
    for kid in (0, 1, 2, 3, 4, 5):
        # In production the two force products come out of the fused QUpdate; the ForcePAOperator kernels are
        # timed with that fusion switched off for the one step they are sampled in.
        _lib.check(L.lgh_set_fused_forces(ctx, 0 if kid in (3, 4) else 1))
        _lib.check(L.lgh_ktime_begin(ctx, kid, 4096))
        sim.step()
        n = ctypes.c_int()
        mean = ctypes.c_double()
        _lib.check(L.lgh_ktime_end(ctx, ctypes.byref(n), ctypes.byref(mean)))
        _lib.check(L.lgh_set_fused_forces(ctx, 1))
        if n.value:
            raw[kid] = (n.value, mean.value)
            kern[KERNEL_NAMES[kid]] = {"launches": n.value, "mean_us": 1e6 * mean.value, "bytes_per_launch": bts[kid],
                                       "GBs": 1e-9 * bts[kid] / mean.value, "frac": 1e-9 * bts[kid] / mean.value / HBM_PEAK_GBS,
                                       "sec8d_bytes_per_launch": sec8d[kid], "sec8d_GBs": 1e-9 * sec8d[kid] / mean.value,
                                       "us_per_rk_step": 1e6 * mean.value * n.value}
            if kid in notes:
                kern[KERNEL_NAMES[kid]]["moves"] = notes[kid]

    def aggregate(ids):
        if any(k not in raw for k in ids):
            return None
        t = sum(raw[k][0] * raw[k][1] for k in ids)
        b = sum(raw[k][0] * bts[k] for k in ids)
        b8 = sum(raw[k][0] * sec8d[k] for k in ids)
        return {"kernels": [KERNEL_NAMES[k].split(" ")[0] for k in ids], "launches_per_rk_step": {KERNEL_NAMES[k].split(" ")[0]: raw[k][0] for k in ids},
                "bytes_per_rk_step": b, "seconds_per_rk_step": t, "achieved": 1e-9 * b / t,
                "frac": 1e-9 * b / t / HBM_PEAK_GBS, "frac_of_achievable": 1e-9 * b / t / HBM_ACHIEVABLE_GBS,
                # the same with SURVEY 8(d)'s bytes (the reference's kernels: quadrature table and E-vector in for every mass apply)
                "sec8d_bytes_per_rk_step": b8, "sec8d_achieved": 1e-9 * b8 / t, "sec8d_frac": 1e-9 * b8 / t / HBM_PEAK_GBS}
    # north_star: "Force+Mass operator apply" = ForceMult + ForceMultTranspose + the mass applies of the H1 CG (K1);
    # the node kernel of the CG (K2) listed with it in a second figure
    agg = {"force_mass_aggregate": aggregate((3, 4, 0)), "force_mass_cg_aggregate": aggregate((3, 4, 0, 1)),
           "force_products_in_production": "formed inside qpoint_kernel (fused QUpdate): no force kernel runs in the timed steps",
           "accounting": "achieved / frac / GBs: bytes the kernel form that ran has to move (operand footprint, moved_bytes() in bench.py); "
                         "sec8d_*: SURVEY 8(d)'s bytes of the reference's kernels, for comparison"}
    return kern, agg, raw, KERNEL_NAMES


def measure_comm(sim, world):
    """What the exchanges cost: HIP-event time of every halo exchange / all-reduce of one RK step (library stream),
    messages per step, size of the largest message.  Several ranks, or the N-rank code path on one (LGH_FORCE_MULTI)."""
    from laghos_amd import _lib
    L = _lib.load()
    ctx = sim.L.laghos_sim_context(sim.h)
    out = {}
    # (first: the sampled steps below run the energy solve after the velocity solve - kernel timing has sequential semantics)
    ls = (ctypes.c_long * 4)()
    _lib.check(L.lgh_energy_lockstep_stats(ctx, ls))
    out["energy_lockstep"] = {"solves": int(ls[0]), "iterations_inside_velocity_solves": int(ls[1]), "iterations_after": int(ls[2])}
    for kid, key in ((6, "halo_exchange"), (7, "allreduce")):
        _lib.check(L.lgh_ktime_begin(ctx, kid, 8192))
        sim.step()
        n, mean = ctypes.c_int(), ctypes.c_double()
        _lib.check(L.lgh_ktime_end(ctx, ctypes.byref(n), ctypes.byref(mean)))
        out[key] = {"per_rk_step": n.value, "mean_us": 1e6 * mean.value if n.value else None}
    nn, ap, c2 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    mx, sh = ctypes.c_long(), ctypes.c_long()
    _lib.check(L.lgh_comm_stats(ctx, ctypes.byref(nn), ctypes.byref(mx), ctypes.byref(sh), ctypes.byref(ap), ctypes.byref(c2)))
    out.update({"neighbours": nn.value, "largest_message_bytes_3_components": 3 * 8 * mx.value, "shared_nodes": sh.value,
                "all_pairs_partition": bool(ap.value), "second_channel": bool(c2.value), "ranks": world})
    return out


def kernel_sources_sha():
    """sha-256 over the HIP sources of the library: counter figures are quoted only for the build they were taken on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "laghos_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_FILE = "r6_pmc_traffic.json"


def pmc_traffic(workload, kernel):
    """Memory-side bytes per launch of `kernel` (substring of its name) from the committed rocprofv3 PMC passes
    (profiles/r4_pmc_traffic.json, written by tools/update_pmc_traffic.py) - only if they were taken with the kernel
    sources this build has; otherwise null."""
    pj = os.path.join(ROOT, "profiles", PMC_FILE)
    try:
        d = json.load(open(pj))
        sha = kernel_sources_sha()
        if d.get("kernel_sources_sha16") != sha:
            return None, "profiles/%s is from another build of the kernel sources (%s != %s): not reported" % (PMC_FILE, d.get("kernel_sources_sha16"), sha)
        ks = d["workloads"][workload]["kernels"]
        hits = [k for k in ks if kernel in k]
        if not hits:
            return None, "no launch of %s in profiles/%s[%s]" % (kernel, PMC_FILE, workload)
        # (several instantiations of one kernel - K2 with and without the update of x - count by their launches)
        nl = sum(ks[k]["launches"] for k in hits)
        mean = sum(ks[k]["launches"] * ks[k]["bytes_per_launch"] for k in hits) / nl
        return mean, ("profiles/%s[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH + WRITE per launch, launch-weighted "
                      "over %s, kernel sources @%s)" % (PMC_FILE, workload, " + ".join(sorted(hits)), sha))
    except Exception as e:
        return None, "unavailable: %r" % (e,)


def roofline_block(kern, agg, raw, names, workload_key):
    """`roofline` of the bench line: the kernel with the largest share of the RK step (by sampled launches x mean
    duration), priced with the bytes its form moves; the other CG kernel and the QUpdate beside it."""
    ids = [k for k in (0, 1, 2, 5) if k in raw]
    if not ids:
        return None
    dom = max(ids, key=lambda k: raw[k][0] * raw[k][1])
    short = {k: names[k].split(" ")[0] for k in ids}
    d = kern[names[dom]]
    traffic, src = pmc_traffic(workload_key, short[dom].split("<")[0]) if workload_key else (None, "not collected for this workload")
    r = {"bound": "hbm", "kernel": names[dom], "achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["GBs"] / HBM_PEAK_GBS,
         "achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": d["GBs"] / HBM_ACHIEVABLE_GBS,
         "traffic": traffic, "traffic_source": src, "mean_launch_us": d["mean_us"], "launches_sampled": d["launches"],
         "bytes_per_launch": d["bytes_per_launch"], "sec8d_bytes_per_launch": d["sec8d_bytes_per_launch"],
         "sec8d_achieved": d["sec8d_GBs"], "sec8d_frac": d["sec8d_GBs"] / HBM_PEAK_GBS,
         "dominant_by": "largest launches x mean duration of one sampled RK step",
         "time_share_us_per_rk_step": {short[k]: 1e6 * raw[k][0] * raw[k][1] for k in ids}}
    beside = {}
    for k in ids:
        if k == dom:
            continue
        e = kern[names[k]]
        t, tsrc = pmc_traffic(workload_key, short[k].split("<")[0]) if workload_key else (None, None)
        beside[short[k]] = {"achieved": e["GBs"], "frac": e["frac"], "mean_launch_us": e["mean_us"], "launches_sampled": e["launches"],
                            "bytes_per_launch": e["bytes_per_launch"], "sec8d_bytes_per_launch": e["sec8d_bytes_per_launch"],
                            "sec8d_frac": e["sec8d_GBs"] / HBM_PEAK_GBS, "traffic": t}
    r["other_kernels"] = beside
    r.update(agg)
    return r


def run_leg(host_lib, args, steps, warmup, dev, force_multi=False, env=None, pmc_key=None):
    """One extra single-GPU workload (a BASELINE.json config other than the one `value` is quoted on)."""
    import torch
    env = dict(env or {})
    if force_multi:
        env["LGH_FORCE_MULTI"] = "1"  # read by the host layer when it builds the operator
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _run_leg(host_lib, args, steps, warmup, dev, force_multi, pmc_key)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run_leg(host_lib, args, steps, warmup, dev, force_multi, pmc_key):
    import torch
    sim = host_lib.Sim(args + ["-dev", dev, "-q"])
    sim.enable_timers(False)
    sz = sim.sizes()
    for _ in range(warmup):
        sim.step()
    sim.sync()
    torch.cuda.synchronize()
    r0 = sim.rk_steps
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    sim.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rk = sim.rk_steps - r0
    dofs = sz["H1GTV"] + sz["L2GTV"]
    kern, agg, raw, names = measure_kernels(sim, sz)
    out = {"value": 1e-6 * dofs * 4 * rk / wall, "unit": "Mdofs*steps/s", "ms_per_step": 1e3 * wall / steps, "steps": steps,
           "elements": sz["global_NE"], "h1_dofs": sz["H1GTV"], "l2_dofs": sz["L2GTV"], "e_norm": sim.e_norm(), "t": sim.t,
           "kernels": {k: {kk: v[kk] for kk in ("mean_us", "GBs", "frac", "sec8d_GBs", "launches", "bytes_per_launch")} for k, v in kern.items()}}
    out.update({k: ({kk: v[kk] for kk in ("achieved", "frac", "frac_of_achievable", "sec8d_achieved", "sec8d_frac")} if (isinstance(v, dict) and "achieved" in v) else v)
                for k, v in agg.items() if k != "accounting"})
    rl = roofline_block(kern, agg, raw, names, pmc_key)
    if rl:
        out["roofline"] = {k: rl[k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_source", "mean_launch_us", "bytes_per_launch",
                                               "sec8d_frac", "time_share_us_per_rk_step")}
    if force_multi:
        out["comm"] = measure_comm(sim, 1)
        out["energy_lockstep"] = out["comm"]["energy_lockstep"]
    # K1 / K2 / update microseconds per launch and what the velocity solve found of the mesh's structure (lgh_vcg_layout_stats)
    out["k_us"] = {key: 1e6 * raw[k][1] for k, key in ((0, "k1"), (1, "k2"), (2, "q")) if k in raw}
    try:
        from laghos_amd import _lib
        st4 = (ctypes.c_long * 4)()
        _lib.check(_lib.load().lgh_vcg_layout_stats(sim.L.laghos_sim_context(sim.h), st4))
        out["vcg_layout"] = {"evector_values_per_component": int(st4[0]), "table_bytes": int(st4[1]) + int(st4[2]), "merged_entries": int(st4[3])}
    except Exception as e:
        out["vcg_layout"] = {"error": repr(e)}
    sim.close()
    return out


WORKLOADS = {
    # BASELINE.json configs[1]: the one `value` is quoted on
    "c2": (["-m", "data/cube01_hex.mesh", "-rs", 4, "-p", 1],
           "3D Sedov -p 1 -m cube01_hex -rs 4 -ok 3 -ot 2 -pa (32^3 elements, E0/2^dim = 0.125)"),
    # configs[3] on one GPU: the 64^3 mesh (HBM-resident: stressJinvT alone is 3.8 GiB)
    "c3": (["-m", "data/cube01_hex.mesh", "-rs", 5, "-p", 1],
           "3D Sedov -p 1 -m cube01_hex -rs 5 -ok 3 -ot 2 -pa (64^3 elements, HBM-resident)"),
    # configs[2]: the bandwidth roofline run (smooth flow, no artificial viscosity)
    "tg": (["-m", "data/cube01_hex.mesh", "-rs", 5, "-p", 0],
           "3D Taylor-Green -p 0 -m cube01_hex -rs 5 -ok 3 -ot 2 -pa (64^3 elements, visc off)"),
}
# further legs of the default run (never the headline):
#  c5     BASELINE.json configs[4] on one GPU: 3D triple point Q5/Q4, 65 536 zones (the high-order kernels)
#  c2dev  configs[1] after 300 time steps: the QUpdate on a developed flow (the timed window of `value` starts at t = 0,
#         where its eigen-decomposition shortcut is at its most favourable)
#  c2multi configs[1] through the N-rank code path on one rank (LGH_FORCE_MULTI=1: real RCCL calls on a communicator of
#         size 1): what the multi-rank sequencing itself costs
LEGS = {
    "c3": dict(args=WORKLOADS["c3"][0], order=(3, 2), steps=5, warmup=2, workload=WORKLOADS["c3"][1]),
    "tg": dict(args=WORKLOADS["tg"][0], order=(3, 2), steps=5, warmup=2, workload=WORKLOADS["tg"][1]),
    "c5": dict(args=["-m", "data/box01_hex.mesh", "-rs", 4, "-p", 3], order=(5, 4), steps=2, warmup=1,
               workload="3D triple point -p 3 -m box01_hex -rs 4 -ok 5 -ot 4 -pa (65 536 zones, Q5/Q4; BASELINE config 5 on one GPU)"),
    "c2dev": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=300,
                  workload=WORKLOADS["c2"][1] + ", after 300 time steps (developed flow)"),
    # configs[1] on the GENERAL-MESH path (round-5 verdict, item 6): the stored mass quadrature table (LGH_MASS_RANK1=0) AND
    # Jac0inv per quadrature point (LGH_JAC0_COMPACT=0) - what a curved or graded initial mesh, or a density that varies inside
    # a zone, takes; the reference's data makes neither assumption (laghos_assembly.cpp:92-95, laghos_solver.cpp:1243-1251)
    "c2general": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, env={"LGH_MASS_RANK1": "0", "LGH_JAC0_COMPACT": "0"},
                      workload=WORKLOADS["c2"][1] + ", general-mesh path: stored mass table, Jac0inv per point (LGH_MASS_RANK1=0 LGH_JAC0_COMPACT=0)"),
    "c2multi": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, force_multi=True,
                    workload=WORKLOADS["c2"][1] + ", N-rank code path on one rank (LGH_FORCE_MULTI=1, RCCL communicator of size 1)"),
    # ... on ONE communicator (LGH_COMM2=0: what N ranks over RCCL run by default - no second channel): the energy CG in lockstep
    # with the velocity CG (lgh_energy_lockstep_stats), and, for the A/B, after it (LGH_ENERGY_LOCKSTEP=0)
    "c2multi1c": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, force_multi=True, env={"LGH_COMM2": "0"},
                      workload=WORKLOADS["c2"][1] + ", N-rank code path on one rank, one communicator: energy CG in lockstep with the velocity CG"),
    "c2multi1cseq": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, force_multi=True, env={"LGH_COMM2": "0", "LGH_ENERGY_LOCKSTEP": "0"},
                         workload=WORKLOADS["c2"][1] + ", N-rank code path on one rank, one communicator: energy CG after the velocity CG"),
    # the general-mesh twins of the N-rank and the HBM-resident legs
    "c2multigeneral": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, force_multi=True, env={"LGH_MASS_RANK1": "0", "LGH_JAC0_COMPACT": "0"},
                           workload=WORKLOADS["c2"][1] + ", N-rank code path on one rank, general-mesh path"),
    "c3general": dict(args=WORKLOADS["c3"][0], order=(3, 2), steps=4, warmup=2, env={"LGH_MASS_RANK1": "0", "LGH_JAC0_COMPACT": "0"},
                      workload=WORKLOADS["c3"][1] + ", general-mesh path: stored mass table, Jac0inv per point"),
    # configs[1] in the numbering the reference's operator API would hand over (round-5 verdict, item 1): every other leg runs
    # on this repository's own generator (nodes lexicographic, zones x-fastest).  c2mfem: MFEM's numbering of the same mesh
    # (`-renumber mfem`, laghos_amd/host/fem.cpp::MfemLikeNumbering: vertex / edge / face / interior dofs, zones in
    # refinement-tree order - /root/reference/laghos.cpp:391, laghos_assembly.cpp:133-134); c2perm: random nodes and zones
    "c2mfem": dict(args=WORKLOADS["c2"][0] + ["-renumber", "mfem"], order=(3, 2), steps=10, warmup=3,
                   workload=WORKLOADS["c2"][1] + ", MFEM-like numbering of nodes and zones (-renumber mfem)"),
    "c2perm": dict(args=WORKLOADS["c2"][0] + ["-renumber", "random"], order=(3, 2), steps=10, warmup=3,
                   workload=WORKLOADS["c2"][1] + ", random numbering of nodes and zones (-renumber random)"),
}


LINE_LIMIT = 8192   # the driver keeps the last 8 KB of stdout: the contract line has to fit with room to spare


def _sig(x, n=6):
    """floats of the line to n significant digits (the detail file keeps full precision)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, x))


def _round_all(o, n=6):
    if isinstance(o, dict):
        return {k: _round_all(v, n) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_all(v, n) for v in o]
    return _sig(o, n)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, detail_path=None):
    """The ONE line the driver parses: the contract fields, `roofline` of the dominant kernel (+ the three kernels
    beside it and the Force+Mass aggregates, figures only), `cpu_baseline`, `parity` and per leg value / ms_per_step /
    roofline kernel and fraction.  Everything else bench.py measures (per-kernel tables, the reference's FOM table, the
    exchange statistics, SURVEY 8(d)'s bytes of the reference's kernel forms) is in the detail record written next to
    it.  The reference's own reporter is eleven numbers (laghos_solver.cpp:699-797).  Pure function of the full record:
    tests/test_bench_contract.py applies it to committed records."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    line["config"] = _pick(full.get("config", {}), ("workload", "transport", "elements", "h1_dofs", "l2_dofs", "quad_points_per_element",
                                                   "rk_stages_executed", "ode", "cg_rel_tol", "parallelism", "zones_per_gpu",
                                                   "qupdate_division", "mass_data", "mesh_order", "e_norm", "t", "dt"))
    r = full.get("roofline")
    if r:
        rl = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "mean_launch_us", "launches_sampled",
                       "bytes_per_launch", "time_share_us_per_rk_step"))
        rl["kernel"] = rl.get("kernel", "").split(" ")[0]
        if isinstance(rl.get("traffic_source"), str):  # file[workload] only; the passes and the source hash are in the detail record
            rl["traffic_source"] = rl["traffic_source"].split(" (")[0][:120]
        rl["other_kernels"] = {k: _pick(v, ("frac", "mean_launch_us", "bytes_per_launch", "traffic")) for k, v in r.get("other_kernels", {}).items()}
        for key in ("force_mass_aggregate", "force_mass_cg_aggregate"):
            if isinstance(r.get(key), dict):
                rl[key] = _pick(r[key], ("achieved", "frac", "bytes_per_rk_step", "seconds_per_rk_step"))
        line["roofline"] = rl
    if "comm" in full:
        c = full["comm"]
        line["comm"] = {k: c[k] for k in c if k in ("halo_exchange", "allreduce", "neighbours", "largest_message_bytes_3_components",
                                                    "all_pairs_partition", "second_channel", "ranks", "energy_lockstep")}
    if "cpu_baseline" in full:
        line["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "cpu_model", "flags", "value_parity_build", "flags_parity_build",
                                                           "native_build_error", "error"))
    if "parity" in full:
        line["parity"] = _pick(full["parity"], ("pass", "rk4_steps", "e_norm_rel_diff", "dt_rel_diff", "state_x_max_rel_diff",
                                                "state_v_max_rel_diff", "state_e_max_rel_diff", "tolerances", "error"))
    if "legs" in full:
        legs = {}
        for name, g in full["legs"].items():
            if "error" in g:
                legs[name] = {"error": str(g["error"])[:160]}
                continue
            e = _pick(g, ("value", "ms_per_step", "ms_per_step_minus_single_rank_path"))
            if isinstance(g.get("roofline"), dict):
                e["kernel"] = str(g["roofline"].get("kernel", "")).split(" ")[0]
                e["frac"] = g["roofline"].get("frac")
            if isinstance(g.get("force_mass_aggregate"), dict):
                e["force_mass_frac"] = g["force_mass_aggregate"].get("frac")
            if isinstance(g.get("k_us"), dict):
                e["k_us"] = g["k_us"]
            if isinstance(g.get("energy_lockstep"), dict) and g["energy_lockstep"].get("solves"):
                e["lockstep_solves"] = g["energy_lockstep"]["solves"]
            if name in ("c2mfem", "c2perm") and isinstance(g.get("vcg_layout"), dict):
                e["merged_entries"] = g["vcg_layout"].get("merged_entries")
            legs[name] = e
        line["legs"] = legs
    if detail_path:
        line["detail"] = detail_path
    line = _round_all(line)
    # `value` and `ms_per_step` are what the driver checks against its own clock: full precision
    for k in ("value", "ms_per_step"):
        if k in full:
            line[k] = full[k]
    # ... and the verification values of the run (|e|, t, dt: what the reference prints to compare runs by) keep every bit
    for k in ("e_norm", "t", "dt"):
        if k in full.get("config", {}):
            line["config"][k] = full["config"][k]
    return line


def write_detail(full, path):
    """the full record (per-kernel tables, legs, FOMs, exchange statistics) as a side file; never on stdout"""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        return os.path.relpath(path, ROOT)
    except Exception as e:
        return "not written: %r" % (e,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="single-GPU workload `value` is measured on (default: BASELINE.json configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs (64^3 Sedov / Taylor-Green, Q5Q4, developed flow, N-rank path)")
    ap.add_argument("--legs", default="c2mfem,c2perm,c3,tg,c5,c2dev,c2general,c2multi,c2multi1c,c2multi1cseq,c2multigeneral,c3general", help="comma-separated extra legs of a single-GPU run")
    ap.add_argument("--transport", choices=("rccl", "shm"), default="rccl",
                    help="several ranks: rccl = the product transport (one GPU per rank, RCCL over xGMI); shm = the cross-process loopback "
                         "transport of lgh_comm.hip (ranks may share one GPU: the torchrun / id broadcast / N-rank code path on a one-GPU box)")
    ap.add_argument("--block", type=int, default=32, help="several ranks: zones per rank and axis (32 = BASELINE.json's weak-scaling block; smaller: tests)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file the full record goes to (per-kernel tables, legs, FOM table, exchange statistics); the stdout line carries the contract only")
    ap.add_argument("--watchdog", type=float, default=300.0,
                    help="several ranks: seconds a rank may spend without finishing a step before it reports and exits (a mismatched collective would otherwise hang silently)")
    a = ap.parse_args()

    import torch
    from laghos_amd import _lib, host_lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback)")
    shm = a.transport == "shm"
    if shm:
        local_rank = local_rank % torch.cuda.device_count()  # the ranks may share a GPU
    torch.cuda.set_device(local_rank)
    nccl_id = None
    dist = None
    if world > 1:
        # torch.distributed only carries the rendezvous (the 128-byte communicator id) and the host barriers around the
        # timed region, over gloo on either transport: the one RCCL instance of the process is the library's own (torch's
        # nccl backend would load a second copy of RCCL beside the one lgh_comm.hip opens, for a broadcast of 128 bytes),
        # and it is the path the one-GPU box can execute (tests/test_gpu_multiproc.py).  The ranks' GPU work is fenced by
        # sim.sync() + torch.cuda.synchronize() on both sides of every barrier.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: no dependence on the host name resolving
        dist.init_process_group("gloo", rank=rank, world_size=world)
        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            cid = ctypes.create_string_buffer(128)
            _lib.check(_lib.load().lgh_comm_unique_id_shm(cid) if shm else _lib.load().lgh_comm_unique_id(cid))
            buf.copy_(torch.tensor(list(cid.raw), dtype=torch.uint8))
        dist.broadcast(buf, 0)
        nccl_id = bytes(buf.cpu().tolist())

    B = a.block
    px, py, pz = block_grid(world, B)
    if world == 1:
        args, workload = WORKLOADS[a.workload]
        args = list(args)
    else:
        args = ["-dim", 3, "-nx", B * px, "-ny", B * py, "-nz", B * pz, "-Sx", px, "-Sy", py, "-Sz", pz,
                "-rs", 0, "-p", 1]
        workload = ("3D Sedov -p 1 Cartesian %dx%dx%d elements (%d^3 per GPU, h = 1/%d), -ok 3 -ot 2 -pa"
                    % (B * px, B * py, B * pz, B, B))
    common = ["-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", a.warmup + a.steps + 64, "-vs", 10 ** 9]
    sim = host_lib.Sim(args + common + ["-dev", local_rank, "-q"], nranks=world, rank=rank, nccl_id=nccl_id)
    flush_c_stdio()  # RCCL's banner (C stdio) out now, on every rank, not at process exit after the JSON line
    sim.enable_timers(False)  # region stopwatches synchronise; keep them out of the timed loop
    sz = sim.sizes()
    if world > 1:
        # the library's own process grid and per-rank block: what the line reports must be what ran
        assert tuple(sz["pgrid"]) == (px, py, pz), (sz["pgrid"], (px, py, pz))
        assert tuple(sz["local_ne"]) == (B, B, B), sz["local_ne"]

    def barrier():
        sim.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # several ranks: a collective that does not match on all ranks blocks inside a stream synchronisation that nothing
    # can interrupt; the watchdog turns that into a rank-tagged message and a non-zero exit instead of a silent hang
    progress = {"where": "start", "t": time.time()}

    def mark(where):
        progress["where"], progress["t"] = where, time.time()

    def watchdog():
        while not progress.get("done"):
            time.sleep(2.0)
            if time.time() - progress["t"] > a.watchdog:
                sys.stderr.write("[bench.py rank %d/%d] no progress for %.0f s in '%s' - a collective probably does not match "
                                 "across the ranks (try LGH_COMM2=0, then LGH_RZ_LIMBS=0, then LGH_HALO_PIGGYBACK=0 on every rank); giving up\n" % (rank, world, a.watchdog, progress["where"]))
                sys.stderr.flush()
                os._exit(3)
    if world > 1:
        import threading
        threading.Thread(target=watchdog, daemon=True).start()

    for i in range(a.warmup):
        mark("warm-up step %d" % i)
        sim.step()
    mark("barrier after warm-up")
    barrier()
    rk0 = sim.rk_steps
    t0 = time.perf_counter()
    for i in range(a.steps):
        mark("timed step %d" % i)
        sim.step()
    mark("barrier after the timed steps")
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        w = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    rk_steps = sim.rk_steps - rk0
    dofs = sz["H1GTV"] + sz["L2GTV"]
    value = 1e-6 * dofs * 4 * rk_steps / wall

    out = {
        "metric": "Mdofs×steps/s on 3D Sedov -pa (Q3/Q2)", "value": value, "unit": "Mdofs*steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * wall / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        # no dataset: the state the kernels run on is the live flow evolved from the problem's initial condition
        "data": "synthetic (live Sedov state evolved from the analytic initial condition; no dataset)",
        "config": {"workload": workload, "transport": ("shm: cross-process loopback through shared memory, ranks share GPUs (NOT the product transport; "
                                                        "exercises the N-rank code path only)" if shm else "rccl") if world > 1 else "none (one rank)",
                   "elements": sz["global_NE"], "h1_dofs": sz["H1GTV"],
                   "l2_dofs": sz["L2GTV"], "quad_points_per_element": sz["NQ"], "rk_stages_executed": 4 * rk_steps,
                   "ode": "RK4", "cg_rel_tol": 1e-8, "parallelism": "elements%dx%dx%d" % tuple(sz["pgrid"]),
                   "zones_per_gpu": "%dx%dx%d" % tuple(sz["local_ne"]),
                   # precision note: fp64 throughout; the fused QUpdate (lgh_qupdate.hip only) divides by reciprocal +
                   # 2 Newton steps + residual correction (<= 2 ulp) instead of the IEEE divide sequence
                   "qupdate_division": "fp64 reciprocal + 2 Newton steps + correction, <= 2 ulp (-freciprocal-math -fapprox-func)",
                   "mass_data": mass_data_note(sim),
                   "mesh_order": mesh_order_note(sim),
                   "e_norm": sim.e_norm(), "t": sim.t, "dt": sim.dt},
    }

    # ---- per-region FOMs (reference formulas, laghos_solver.cpp:722-727) and the
    # roofline of the dominant kernel: extra steps AFTER the timed region so the
    # stopwatch synchronisations / event records do not perturb `value`.
    if not a.no_roofline:
        sim.enable_timers(True)
        sim.reset_timers()
        r0 = sim.rk_steps
        sim.step()
        sim.step()
        sim.sync()
        tm = sim.timers()
        n_rk = sim.rk_steps - r0
        dim = sz["dim"]
        if tm["cgH1"] > 0 and world == 1:
            out["fom"] = {
                "FOM1_cgH1": 1e-6 * sz["H1GTV"] * (tm["H1iter"] / dim) / tm["cgH1"],
                "FOM2_forces": 1e-6 * 4 * n_rk * dofs / tm["force"],
                "FOM3_qdata": 1e-6 * tm["quad_tstep"] * sz["NQ"] / tm["qdata"],
                "h1_cg_iters_per_solve": tm["H1iter"] / (4.0 * n_rk * dim),
                "seconds": {k: tm[k] for k in ("cgH1", "cgL2", "force", "qdata")},
            }
        sim.enable_timers(False)
        mark("per-kernel timing")
        kern, agg, raw, names = measure_kernels(sim, sz)
        rl = roofline_block(kern, agg, raw, names, a.workload if (world == 1 and a.workload in ("c2", "c3", "tg")) else None)
        if rl:
            out["roofline"] = rl
        out["kernels"] = kern
        if world > 1 or os.environ.get("LGH_FORCE_MULTI") == "1":
            mark("exchange timing")
            out["comm"] = measure_comm(sim, world)
    progress["done"] = True
    sim.close()

    # ---- the other single-GPU configs of BASELINE.json (and two views of configs[1]) as short extra legs, not part of `value`
    if world == 1 and not a.no_legs and not a.no_roofline:
        legs = {}
        for name in [n for n in a.legs.split(",") if n in LEGS]:
            if name == a.workload:
                continue
            leg = LEGS[name]
            try:
                largs = list(leg["args"]) + ["-ok", leg["order"][0], "-ot", leg["order"][1], "-pa", "-tf", 1e9, "-ms", 10 ** 6, "-vs", 10 ** 9]
                legs[name] = run_leg(host_lib, largs, steps=leg["steps"], warmup=leg["warmup"], dev=local_rank, force_multi=leg.get("force_multi", False),
                                     env=leg.get("env"), pmc_key=name if name in ("c3", "tg") else None)
                legs[name]["workload"] = leg["workload"]
            except Exception as e:  # an extra leg must not cost the headline number
                legs[name] = {"error": repr(e)}
        for name in ("c2multi", "c2multi1c", "c2multi1cseq"):
            if name in legs and "value" in legs[name]:
                legs[name]["ms_per_step_minus_single_rank_path"] = legs[name]["ms_per_step"] - out["ms_per_step"]
        out["legs"] = legs

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"], ref = cpu_baseline(min(usable_cpus(), 64))
            if a.workload == "c2":
                out["parity"] = parity_block(host_lib, list(WORKLOADS["c2"][0]) + common, local_rank, ref)
        except Exception as e:  # the checker is optional for the measurement
            out.setdefault("cpu_baseline", {"error": repr(e)})
            out.setdefault("parity", {"error": repr(e)})
    flush_c_stdio()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio: flush that first so that the JSON line is
        # the last line of stdout
        flush_c_stdio()
        sys.stdout.flush()
        line = compact_line(out, write_detail(out, a.detail))
        text = json.dumps(line, separators=(",", ":"))
        if len(text) > LINE_LIMIT:  # never let an oversized line cost the driver its record again
            for key in ("legs", "comm"):
                line.pop(key, None)
            text = json.dumps(line, separators=(",", ":"))
        print(text, flush=True)


if __name__ == "__main__":
    main()

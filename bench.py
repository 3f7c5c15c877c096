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


def cpu_baseline(threads):
    """The oracle (CPU restatement of the reference -pa path) timed on the host
    cores on a bounded sample of the same 32^3 Q3/Q2 Sedov workload: whole RK4
    steps from t=0 with the real dt controller inputs (each step = 4 stages of
    1 QUpdate, 1 Force, 1 ForceT, 3 H1 PCG, 1 L2 CG, + the dt-estimate QUpdate)
    until ~12 s of CPU work.  Reported next to the GPU number; never shipped."""
    import numpy as np
    from oracle.driver import Hydro, lib, rk4_step
    from oracle.fem import Problem
    lib().lgo_set_num_threads(threads)
    prob = Problem(mesh="cube01_hex", rs=4, order_v=3, order_e=2, problem=1)
    h = Hydro(prob)
    S = h.S0.copy()
    work = (np.empty_like(S), np.empty_like(S), np.empty_like(S))
    h.reset_time_step_estimate()
    dt = h.get_time_step_estimate(S)
    steps, wall, t = 0, 0.0, 0.0
    while wall < 12.0 and steps < 40:
        t0 = time.time()
        h.reset_time_step_estimate()
        t = rk4_step(h, S, t, dt, work)
        dt_est = h.get_time_step_estimate(S)
        wall += time.time() - t0
        steps += 1
        if dt_est > 1.25 * dt:
            dt *= 1.02
    dofs = prob.dim * prob.N + prob.L2V
    tm = h.timers()
    h.close()
    return dict(value=1e-6 * dofs * 4 * steps / wall, unit="Mdofs*steps/s", cores=threads, cpu_model=cpu_model(), kind="port",
                sample="%d RK4 steps from t=0 of the same 3D Sedov Q3Q2 32^3 problem (oracle/ C++ kernels, "
                       "OpenMP %d threads), %.1f s" % (steps, threads, wall),
                seconds=wall, rk4_steps=steps, h1_cg_iters=tm["H1iter"])


KERNEL_NAMES = {0: "vcg_apply_plane (H1 CG K1, 3 velocity components per launch)",
                1: "vcg_update_k (H1 CG K2, 3 velocity components per launch)",
                2: "qpoint_kernel (fused QUpdate)", 3: "force_mult_3d", 4: "force_mult_t_3d",
                5: "mass_apply_3d (L2 CG K1)",
                6: "halo_sum (pack + grouped ncclSend/Recv + combine)", 7: "ncclAllReduce of device scalars"}
K1_FORMS = {0: "vcg_apply_3d", 2: "vcg_apply_plane", 3: "vcg_apply_mfma346", 4: "vcg_apply_slab346"}


def kernel_names(L, ctx):
    """KERNEL_NAMES with the K1 form this context really launches (lgh_k1_form)"""
    f = ctypes.c_int(-2)
    L.lgh_k1_form(ctx, ctypes.byref(f))
    names = dict(KERNEL_NAMES)
    names[0] = K1_FORMS.get(f.value, "vcg_apply") + " (H1 CG K1, 3 velocity components per launch)"
    return names


def algorithmic_bytes(sz):
    """Algorithmic bytes per launch, fp64 (SURVEY §8d / DESIGN.md "Roofline accounting")."""
    dim, D, Q, Ld = sz["dim"], sz["D1D"], sz["Q1D"], sz["L1D"]
    NQ, ND, NL, NE, N = sz["NQ"], D ** dim, Ld ** dim, sz["NE"], sz["N"]
    return {
        0: NE * 8 * (NQ + dim * 2 * ND),                        # lockstep mass apply: D once + (in + out) per component
        # K2, per launch: r, d, x read and written, 1/diag, the element contributions (E-vector), their ELL index table
        1: 8 * N * (dim * 6 + 1) + 8 * dim * NE * ND + 4 * 8 * N,
        2: NE * (8 * (2 * dim * ND + NL + dim * dim * NQ + NQ + dim * dim * NQ) + 8),  # fused QUpdate
        3: NE * 8 * (dim * dim * NQ + NL + dim * ND),           # ForceMult
        4: NE * 8 * (dim * dim * NQ + NL + dim * ND),           # ForceMultTranspose
        5: NE * 8 * (NQ + 2 * NL),                              # L2 mass apply
    }


def measure_kernels(sim, sz):
    """HIP-event timing (on the library's stream) of every launch of each hot kernel during one RK step per
    kernel, after the timed region.  Returns (per-kernel dict, aggregates)."""
    from laghos_amd import _lib
    L = _lib.load()
    ctx = sim.L.laghos_sim_context(sim.h)
    bts = algorithmic_bytes(sz)
    KERNEL_NAMES = kernel_names(L, ctx)
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
            kern[KERNEL_NAMES[kid]] = {"launches": n.value, "mean_us": 1e6 * mean.value, "algorithmic_bytes": bts[kid],
                                       "GBs": 1e-9 * bts[kid] / mean.value}

    def aggregate(ids):
        if any(k not in raw for k in ids):
            return None
        b = sum(raw[k][0] * bts[k] for k in ids)
        t = sum(raw[k][0] * raw[k][1] for k in ids)
        return {"kernels": [KERNEL_NAMES[k].split(" ")[0] for k in ids], "launches_per_rk_step": {KERNEL_NAMES[k].split(" ")[0]: raw[k][0] for k in ids},
                "algorithmic_bytes_per_rk_step": b, "seconds_per_rk_step": t, "achieved": 1e-9 * b / t,
                "frac": 1e-9 * b / t / HBM_PEAK_GBS, "frac_of_achievable": 1e-9 * b / t / HBM_ACHIEVABLE_GBS}
    # north_star: "Force+Mass operator apply" = ForceMult + ForceMultTranspose + the mass applies of the H1 CG (K1);
    # the node kernel of the CG (K2) listed with it in a second figure
    agg = {"force_mass_aggregate": aggregate((3, 4, 0)), "force_mass_cg_aggregate": aggregate((3, 4, 0, 1)),
           "force_products_in_production": "formed inside qpoint_kernel (fused QUpdate): no force kernel runs in the timed steps"}
    if 0 in raw:
        # What `achieved` counts and what the kernel that ran really has to move (DESIGN.md "Roofline accounting").
        # `achieved` is SURVEY 8(d)'s figure for the mass apply, 8 (NQ + 2 dim D1D^3) bytes per element: the quadrature
        # data once, one E-vector in, one out.  The lockstep K1 departs from it on both sides: with compact mass data
        # (lgh_mass_data_form: D[q, e] = W[q] s_e) it reads 8 bytes of quadrature data per element instead of 8 NQ,
        # and it forms the search direction itself, d = z + beta d_old, z = r / diag - it gathers r, d_old (dim
        # components) and 1 / diag where 8(d) has a single input vector.  `compulsory` is the unique footprint of those
        # operands (every node vector once, the element output once): the least the launch can take from memory.
        dim, D, N, NE, NQ = sz["dim"], sz["D1D"], sz["N"], sz["NE"], sz["NQ"]
        form = ctypes.c_int(-1)
        L.lgh_mass_data_form(ctx, ctypes.byref(form))
        compact = form.value == 1 and KERNEL_NAMES[0].split(" ")[0] in ("vcg_apply_slab346", "vcg_apply_plane")
        comp = 8 * (N * (2 * dim + 1) + NE * dim * D ** dim) + (8 * NE if compact else 8 * NE * NQ)
        t = raw[0][1]
        agg["k1_accounting"] = {
            "achieved_counts": "SURVEY 8(d): 8 (NQ + 2 dim D1D^3) bytes per element and launch",
            "sec8d_bytes_per_launch": bts[0],
            "mass_data": "compact: one factor per element (8 NE bytes instead of 8 NQ NE)" if compact else "stored table",
            "inputs": "r, d_old (dim components each) and 1/diag gathered; d = r/diag + beta d_old formed in the kernel",
            "compulsory_bytes_per_launch": comp, "compulsory_GBs": 1e-9 * comp / t, "compulsory_frac": 1e-9 * comp / t / HBM_PEAK_GBS}
    return kern, agg


def measure_comm(sim, world):
    """What the exchanges cost: HIP-event time of every halo exchange / all-reduce of one RK step (library stream),
    messages per step, size of the largest message.  Several ranks, or the N-rank code path on one (LGH_FORCE_MULTI)."""
    from laghos_amd import _lib
    L = _lib.load()
    ctx = sim.L.laghos_sim_context(sim.h)
    out = {}
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


def k1_sources_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ("lgh_vcg.hpp", "lgh_vcg.hip", "lgh_vcg_slab.hip"):
        h.update(open(os.path.join(ROOT, "laghos_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload="c2"):
    """Memory-side bytes per K1 launch from the committed rocprofv3 PMC passes (profiles/r3_pmc_traffic.json, written by
    tools/update_pmc_traffic.py) - only if they were taken with the K1 sources this build has; otherwise null."""
    pj = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")
    try:
        d = json.load(open(pj))
        sha = k1_sources_sha()
        if d.get("k1_sources_sha16") != sha:
            return None, "profiles/r3_pmc_traffic.json is from another build of the K1 sources (%s != %s): not reported" % (d.get("k1_sources_sha16"), sha)
        w = d["workloads"][workload]
        return w["k1_bytes_per_launch"], ("profiles/r3_pmc_traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH + WRITE per "
                                          "launch of %s, K1 sources @%s)" % (workload, w["k1_kernel"], sha))
    except Exception as e:
        return None, "unavailable: %r" % (e,)


def run_leg(host_lib, args, steps, warmup, dev, force_multi=False):
    """One extra single-GPU workload (a BASELINE.json config other than the one `value` is quoted on)."""
    import torch
    if force_multi:
        os.environ["LGH_FORCE_MULTI"] = "1"  # read by the host layer when it builds the operator
    try:
        sim = host_lib.Sim(args + ["-dev", dev, "-q"])
    finally:
        os.environ.pop("LGH_FORCE_MULTI", None)
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
    kern, agg = measure_kernels(sim, sz)
    out = {"value": 1e-6 * dofs * 4 * rk / wall, "unit": "Mdofs*steps/s", "ms_per_step": 1e3 * wall / steps, "steps": steps,
           "elements": sz["global_NE"], "h1_dofs": sz["H1GTV"], "l2_dofs": sz["L2GTV"], "e_norm": sim.e_norm(), "t": sim.t,
           "kernels": {k: {"mean_us": v["mean_us"], "GBs": v["GBs"], "launches": v["launches"]} for k, v in kern.items()}}
    out.update({k: ({"achieved": v["achieved"], "frac": v["frac"], "frac_of_achievable": v["frac_of_achievable"]} if (isinstance(v, dict) and "achieved" in v) else v)
                for k, v in agg.items()})
    if force_multi:
        out["comm"] = measure_comm(sim, 1)
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
    "c2multi": dict(args=WORKLOADS["c2"][0], order=(3, 2), steps=10, warmup=3, force_multi=True,
                    workload=WORKLOADS["c2"][1] + ", N-rank code path on one rank (LGH_FORCE_MULTI=1, RCCL communicator of size 1)"),
}


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
    ap.add_argument("--legs", default="c3,tg,c5,c2dev,c2multi", help="comma-separated extra legs of a single-GPU run")
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
    torch.cuda.set_device(local_rank)
    nccl_id = None
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        buf = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            cid = ctypes.create_string_buffer(128)
            _lib.check(_lib.load().lgh_comm_unique_id(cid))
            buf.copy_(torch.tensor(list(cid.raw), dtype=torch.uint8))
        dist.broadcast(buf, 0)
        nccl_id = bytes(buf.cpu().tolist())

    px, py, pz = block_grid(world)
    if world == 1:
        args, workload = WORKLOADS[a.workload]
        args = list(args)
    else:
        args = ["-dim", 3, "-nx", 32 * px, "-ny", 32 * py, "-nz", 32 * pz, "-Sx", px, "-Sy", py, "-Sz", pz,
                "-rs", 0, "-p", 1]
        workload = ("3D Sedov -p 1 Cartesian %dx%dx%d elements (32^3 per GPU, h = 1/32), -ok 3 -ot 2 -pa"
                    % (32 * px, 32 * py, 32 * pz))
    common = ["-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", a.warmup + a.steps + 64, "-vs", 10 ** 9]
    sim = host_lib.Sim(args + common + ["-dev", local_rank, "-q"], nranks=world, rank=rank, nccl_id=nccl_id)
    flush_c_stdio()  # RCCL's banner (C stdio) out now, on every rank, not at process exit after the JSON line
    sim.enable_timers(False)  # region stopwatches synchronise; keep them out of the timed loop
    sz = sim.sizes()
    if world > 1:
        # the library's own process grid and per-rank block: what the line reports must be what ran
        assert tuple(sz["pgrid"]) == (px, py, pz), (sz["pgrid"], (px, py, pz))
        assert tuple(sz["local_ne"]) == (32, 32, 32), sz["local_ne"]

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
                                 "across the ranks (try LGH_COMM2=0 LGH_HALO_PIGGYBACK=0); giving up\n" % (rank, world, a.watchdog, progress["where"]))
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
        w = torch.tensor([wall], dtype=torch.float64, device="cuda")
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
        "config": {"workload": workload, "elements": sz["global_NE"], "h1_dofs": sz["H1GTV"],
                   "l2_dofs": sz["L2GTV"], "quad_points_per_element": sz["NQ"], "rk_stages_executed": 4 * rk_steps,
                   "ode": "RK4", "cg_rel_tol": 1e-8, "parallelism": "elements%dx%dx%d" % tuple(sz["pgrid"]),
                   "zones_per_gpu": "%dx%dx%d" % tuple(sz["local_ne"]),
                   # precision note: fp64 throughout; the fused QUpdate (lgh_qupdate.hip only) divides by reciprocal +
                   # 2 Newton steps + residual correction (<= 2 ulp) instead of the IEEE divide sequence
                   "qupdate_division": "fp64 reciprocal + 2 Newton steps + correction, <= 2 ulp (-freciprocal-math -fapprox-func)",
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
        kern, agg = measure_kernels(sim, sz)
        k1_name = [k for k in kern if k.startswith("vcg_apply")]
        dom = kern.get(k1_name[0]) if k1_name else None
        if dom:
            traffic, traffic_source = pmc_traffic(a.workload) if (world == 1 and a.workload in ("c2", "c3")) else (None, "not collected for this workload")
            out["roofline"] = {"bound": "hbm", "kernel": k1_name[0], "achieved": dom["GBs"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": dom["GBs"] / HBM_PEAK_GBS,
                               "achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": dom["GBs"] / HBM_ACHIEVABLE_GBS,
                               "traffic": traffic, "traffic_source": traffic_source,
                               "mean_launch_us": dom["mean_us"], "launches_sampled": dom["launches"],
                               "algorithmic_bytes_per_launch": dom["algorithmic_bytes"]}
            out["roofline"].update(agg)
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
                legs[name] = run_leg(host_lib, largs, steps=leg["steps"], warmup=leg["warmup"], dev=local_rank, force_multi=leg.get("force_multi", False))
                legs[name]["workload"] = leg["workload"]
                if name == "c3":
                    t, src = pmc_traffic("c3")
                    legs[name]["roofline_traffic"] = {"k1_bytes_per_launch": t, "source": src}
            except Exception as e:  # an extra leg must not cost the headline number
                legs[name] = {"error": repr(e)}
        if "c2multi" in legs and "value" in legs["c2multi"]:
            legs["c2multi"]["ms_per_step_minus_single_rank_path"] = legs["c2multi"]["ms_per_step"] - out["ms_per_step"]
        out["legs"] = legs

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(min(usable_cpus(), 64))
        except Exception as e:  # the checker is optional for the measurement
            out["cpu_baseline"] = {"error": repr(e)}
    flush_c_stdio()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio: flush that first so that the JSON line is
        # the last line of stdout
        flush_c_stdio()
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

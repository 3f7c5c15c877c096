"""`bench.py --gpus N` as the driver launches it - torch.distributed.run, one process per rank, id drawn on rank 0 and
broadcast, lgh_comm_init / lgh_comm_set_neighbors, watchdog, max-over-ranks timing, ONE JSON line from rank 0 - executed
on the one-GPU box with the cross-process loopback transport (`--transport shm`, lgh_comm.hip: the ranks are processes
that share the GPU; RCCL itself refuses two ranks on one device).  Everything above the transport is what a multi-GPU
node runs.  The run must reproduce the in-process two-rank run (threads + "LGHLOCAL" communicator) of the same problem
BIT FOR BIT: same kernels, same canonical (rank-ordered) sums - only the way bytes travel differs."""
import json
import os
import socket
import subprocess
import sys
import threading

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun_bench(nproc, extra, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line (rank 0): %r" % (r.stdout[-1000:],)
    return json.loads(lines[0])


def _in_process(nranks, pgrid, block, steps):
    from laghos_amd import host_lib
    px, py, pz = pgrid
    args = ["-dim", 3, "-nx", block * px, "-ny", block * py, "-nz", block * pz, "-Sx", px, "-Sy", py, "-Sz", pz, "-rs", 0, "-p", 1,
            "-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", 10 ** 6, "-vs", 10 ** 9, "-q"]
    cid = (b"LGHLOCAL" + os.urandom(16).hex().encode()).ljust(128, b"\0")
    out, err = {}, {}

    def rank_main(rank):
        try:
            sim = host_lib.Sim(args, nranks=nranks, rank=rank, nccl_id=cid)
            sim.enable_timers(False)
            for _ in range(steps):
                sim.step()
            sim.sync()
            out[rank] = dict(e=sim.e_norm(), t=sim.t, dt=sim.dt, rk=sim.rk_steps)
            sim.close()
        except Exception as ex:  # noqa: BLE001 - reported below
            err[rank] = repr(ex)

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th) and not err, err
    return out[0]


@pytest.mark.parametrize("nproc,pgrid,slab", [(2, (2, 1, 1), 0), (4, (2, 2, 1), 0), (2, (2, 1, 1), 1), (2, (2, 1, 1), 2), (4, (2, 2, 1), 2)],
                         ids=["2ranks", "4ranks", "2ranks-slab-words", "2ranks-slab-words-one-communicator", "4ranks-slab-words-one-communicator"])
def test_bench_gpus_n_over_the_cross_process_transport(nproc, pgrid, slab, monkeypatch):
    # slab: the kernels config 4's ranks run (slab K1 with the merged E-vector, exact accumulators) on 8^3-zone ranks -
    # (r, z) then crosses the ranks as accumulator WORDS (exchange_words, round 5), here through the shm transport
    # slab == 2: no second channel (LGH_COMM2=0, what a run over RCCL has by default) - the energy CG in lockstep with the
    # velocity CG, its sums on the velocity iteration's messages (round 6)
    if slab:
        monkeypatch.setenv("LGH_VCG_VARIANT", "4")
    if slab == 2:
        monkeypatch.setenv("LGH_COMM2", "0")
    steps, warmup, block = 3, 2, 8
    d = _torchrun_bench(nproc, ["--transport", "shm", "--block", str(block), "--steps", str(steps), "--warmup", str(warmup)])
    assert d["n_gpus"] == nproc and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "elements%dx%dx%d" % pgrid and d["config"]["zones_per_gpu"] == "%dx%dx%d" % (block, block, block)
    assert d["config"]["elements"] == nproc * block ** 3
    assert d["config"]["transport"].startswith("shm")
    # whole-job value from the max-over-ranks wall time
    c = d["config"]
    wall = d["ms_per_step"] * 1e-3 * steps
    assert abs(d["value"] - 1e-6 * (c["h1_dofs"] + c["l2_dofs"]) * c["rk_stages_executed"] / wall) < 1e-6 * d["value"]
    # the comm block of an N-rank line: exchanges per RK step and who talks to whom
    m = d["comm"]
    assert m["ranks"] == nproc and m["neighbours"] == nproc - 1 and m["all_pairs_partition"] is True
    assert m["halo_exchange"]["per_rk_step"] > 0 and m["largest_message_bytes_3_components"] > 0
    assert m["second_channel"] is (slab != 2)
    ls = m["energy_lockstep"]
    assert (ls["solves"] >= 4 * (steps + warmup) - 1 and ls["iterations_inside_velocity_solves"] > 4 * ls["solves"]) if slab == 2 else ls["solves"] == 0, ls
    ref = _in_process(nproc, pgrid, block, warmup + steps)
    assert c["rk_stages_executed"] == 4 * steps
    assert (c["e_norm"], c["t"], c["dt"]) == (ref["e"], ref["t"], ref["dt"]), (c["e_norm"], ref)


def test_cross_process_transport_reports_a_missing_rank(tmp_path):
    """A rank that never shows up must end in an error with a message, not in a hang: lgh_comm_init over the shm
    transport waits for its peers with a time-out (here shortened through the segment never being created)."""
    import ctypes
    from helpers import make_gpu
    from oracle.fem import Problem
    from laghos_amd import _lib
    L = _lib.load()
    prob = Problem(mesh="cube01_hex", rs=0, order_v=2, order_e=1, problem=1)
    g = make_gpu(prob)
    try:
        # rank 1 of 2 with an id nobody created: shm_open_group gives up after its time-out (LGH_SHM_TIMEOUT seconds)
        os.environ["LGH_SHM_TIMEOUT"] = "2"
        cid = ctypes.create_string_buffer(b"LGHSHM_test_missing_%d" % os.getpid(), 128)
        rc = L.lgh_comm_init(g.ctx.h, 2, 1, cid)
        assert rc != 0 and b"shm transport" in L.lgh_last_error()
    finally:
        os.environ.pop("LGH_SHM_TIMEOUT", None)
        g.close()

"""ctypes binding of liblaghos_host.so: the C++ host layer (reference API mirror,
driver, time loop).  bench.py and the end-to-end tests drive it through these
entry points; numerics run in liblaghos_hip.so."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblaghos_host.so")
_lib = None


def _preload_torch_hip():
    """torch bundles its own libamdhip64; if this library pulled in /opt/rocm's copy
    first, a later `import torch` would bring a second HIP runtime into the process
    and fail with "No HIP GPUs are available".  Importing torch first makes both
    share one runtime (the standalone `laghos` executable does not involve torch)."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load():
    global _lib
    if _lib is None:
        _preload_torch_hip()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_PATH)
        P, I, D, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_long
        L.laghos_sim_create.restype = P
        L.laghos_sim_create.argtypes = [I, ctypes.POINTER(ctypes.c_char_p), I, I, ctypes.c_char_p]
        L.laghos_sim_destroy.argtypes = [P]
        L.laghos_sim_step.restype = I
        L.laghos_sim_step.argtypes = [P]
        for n in ("laghos_sim_time", "laghos_sim_dt", "laghos_sim_enorm", "laghos_sim_energy", "laghos_sim_sedov_error"):
            getattr(L, n).restype = D
            getattr(L, n).argtypes = [P]
        for n in ("laghos_sim_steps", "laghos_sim_ti"):
            getattr(L, n).restype = I
            getattr(L, n).argtypes = [P]
        L.laghos_sim_sync.argtypes = [P]
        L.laghos_sim_enable_timers.argtypes = [P, I]
        L.laghos_sim_reset_timers.argtypes = [P]
        L.laghos_sim_timers.argtypes = [P, ctypes.POINTER(D), ctypes.POINTER(Lg)]
        L.laghos_sim_sizes.argtypes = [P, ctypes.POINTER(Lg)]
        L.laghos_sim_context.restype = P
        L.laghos_sim_context.argtypes = [P]
        L.laghos_sim_state_size.restype = Lg
        L.laghos_sim_state_size.argtypes = [P]
        L.laghos_sim_get_state.argtypes = [P, P]
        L.laghos_main.restype = I
        L.laghos_main.argtypes = [I, ctypes.POINTER(ctypes.c_char_p)]
        L.laghos_host_partition.restype = I
        L.laghos_host_partition.argtypes = [I, I, I, I, I, ctypes.POINTER(I)]
        L.laghos_host_tables.restype = I
        L.laghos_host_tables.argtypes = [I, I, P, P, P, P, P, P]
        L.laghos_host_disc_create.restype = P
        L.laghos_host_disc_create.argtypes = [ctypes.c_char_p, I, I, I, I, D, I, I]
        L.laghos_host_disc_create_renumbered.restype = P
        L.laghos_host_disc_create_renumbered.argtypes = [ctypes.c_char_p, I, I, I, I, D, I, I, ctypes.c_char_p, I]
        L.laghos_host_disc_destroy.argtypes = [P]
        L.laghos_host_disc_size.restype = Lg
        L.laghos_host_disc_size.argtypes = [P, I]
        L.laghos_host_disc_get.argtypes = [P, I, P]
        _lib = L
    return _lib


def _argv(args):
    arr = (ctypes.c_char_p * len(args))(*[str(a).encode() for a in args])
    return len(args), arr


class Sim:
    """laghos_sim: one simulation on one GPU (one rank)."""

    def __init__(self, args, nranks=1, rank=0, nccl_id=None):
        self.L = load()
        n, arr = _argv(args)
        self.h = self.L.laghos_sim_create(n, arr, nranks, rank, nccl_id)
        if not self.h:
            raise RuntimeError("laghos_sim_create failed (see stderr)")

    def close(self):
        if self.h:
            self.L.laghos_sim_destroy(self.h)
            self.h = None

    def step(self):
        return self.L.laghos_sim_step(self.h)

    def sync(self):
        self.L.laghos_sim_sync(self.h)

    @property
    def t(self):
        return self.L.laghos_sim_time(self.h)

    @property
    def dt(self):
        return self.L.laghos_sim_dt(self.h)

    @property
    def rk_steps(self):
        return self.L.laghos_sim_steps(self.h)

    @property
    def ti(self):
        return self.L.laghos_sim_ti(self.h)

    def e_norm(self):
        return self.L.laghos_sim_enorm(self.h)

    def energy(self):
        return self.L.laghos_sim_energy(self.h)

    def sedov_error(self):
        """`-err`: L2 error of the density against the exact Sedov solution at t_final."""
        return self.L.laghos_sim_sedov_error(self.h)

    def enable_timers(self, on):
        self.L.laghos_sim_enable_timers(self.h, int(on))

    def reset_timers(self):
        self.L.laghos_sim_reset_timers(self.h)

    def timers(self):
        t = (ctypes.c_double * 4)()
        c = (ctypes.c_long * 3)()
        self.L.laghos_sim_timers(self.h, t, c)
        return dict(cgH1=t[0], cgL2=t[1], force=t[2], qdata=t[3], H1iter=c[0], L2iter=c[1], quad_tstep=c[2])

    def sizes(self):
        s = (ctypes.c_long * 16)()
        self.L.laghos_sim_sizes(self.h, s)
        keys = ["dim", "NE", "global_NE", "N", "H1GTV", "L2GTV", "NQ", "D1D", "Q1D", "L1D"]
        out = dict(zip(keys, list(s)[:10]))
        out["pgrid"] = tuple(s[10:13])      # process grid of laghos::Partition
        out["local_ne"] = tuple(s[13:16])   # zones of this rank per axis
        return out

    def state(self):
        n = self.L.laghos_sim_state_size(self.h)
        out = np.empty(n)
        self.L.laghos_sim_get_state(self.h, out.ctypes.data)
        return out


def host_partition(dim, nx, ny, nz, nranks):
    """Process grid laghos::Partition picks for an nx x ny x nz zone grid (None: not evenly divisible)."""
    L = load()
    pg = (ctypes.c_int * 3)()
    if L.laghos_host_partition(dim, nx, ny, nz, nranks, pg) != 0:
        return None
    return tuple(pg)


def host_tables(order_v, order_e):
    L = load()
    D, Ld = order_v + 1, order_e + 1
    Q = (3 * order_v + order_e - 1) // 2 + 1
    qp, qw, gll = np.empty(Q), np.empty(Q), np.empty(D)
    B, G, Bl = np.empty(Q * D), np.empty(Q * D), np.empty(Q * Ld)
    q = L.laghos_host_tables(order_v, order_e, qp.ctypes.data, qw.ctypes.data, gll.ctypes.data,
                             B.ctypes.data, G.ctypes.data, Bl.ctypes.data)
    assert q == Q
    return dict(qpts=qp, qwts=qw, gll=gll, B=B.reshape(D, Q).T, G=G.reshape(D, Q).T, Bl=Bl.reshape(Ld, Q).T)


def host_disc(mesh, rs, order_v, order_e, problem, blast_energy=1.0, nranks=1, rank=0, renumber=None, seed=1):
    """Arrays of the C++ Discretization for one rank (host only, no GPU); renumber = "mfem" / "random": after
    Discretization::Renumber (`-renumber`), with node_perm / elem_perm in the result."""
    L = load()
    h = L.laghos_host_disc_create_renumbered(mesh.encode(), rs, order_v, order_e, problem, blast_energy, nranks, rank,
                                             renumber.encode() if renumber else None, seed)
    if not h:
        raise RuntimeError("laghos_host_disc_create failed")

    def get(kind, dtype):
        n = L.laghos_host_disc_size(h, kind)
        a = np.empty(max(n, 0), dtype=dtype)
        if n > 0:
            L.laghos_host_disc_get(h, kind, a.ctypes.data)
        return a
    out = dict(h1map=get(0, np.int32), S0=get(1, np.float64), rho0_l2=get(2, np.float64),
               gamma=get(3, np.float64), rho0_q=get(4, np.float64),
               ess=[get(5, np.int32), get(6, np.int32), get(7, np.int32)], owner=get(8, np.float64),
               W=get(9, np.float64), nbr_rank=get(10, np.int32))
    out["nbr_nodes"] = [get(11 + k, np.int32) for k in range(len(out["nbr_rank"]))]
    out["node_perm"], out["elem_perm"] = get(100, np.int32), get(101, np.int32)
    L.laghos_host_disc_destroy(h)
    return out

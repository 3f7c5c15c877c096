"""ctypes binding of liblaghos_hip.so (the C ABI declared in include/laghos_hip.h).

The HIP library is the product; there is no CPU fallback.  Importing this module
loads the in-tree shared object and raises if it is missing or cannot be loaded.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblaghos_hip.so")

c_dp = ctypes.c_void_p  # device pointers travel as integers
c_int_p = ctypes.POINTER(ctypes.c_int)
c_dbl_p = ctypes.POINTER(ctypes.c_double)


class LghConfig(ctypes.Structure):
    """struct lgh_config (include/laghos_hip.h)."""
    _fields_ = [
        ("dim", ctypes.c_int), ("NE", ctypes.c_int),
        ("D1D", ctypes.c_int), ("Q1D", ctypes.c_int), ("L1D", ctypes.c_int),
        ("N", ctypes.c_int),
        ("h1_map", c_int_p),
        ("B_h1", c_dbl_p), ("G_h1", c_dbl_p), ("B_l2", c_dbl_p),
        ("weights", c_dbl_p), ("gamma", c_dbl_p),
        ("ess_count", ctypes.c_int * 3),
        ("ess", c_int_p * 3),
        ("owner", c_dbl_p),
        ("use_viscosity", ctypes.c_int), ("use_vorticity", ctypes.c_int),
        ("cfl", ctypes.c_double),
        ("order_v", ctypes.c_int),
        ("device", ctypes.c_int),
        ("stream", ctypes.c_void_p),
    ]


# every symbol include/laghos_hip.h declares: name -> (restype, argtypes)
_I, _D, _P, _L = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_long
SYMBOLS = {
    "lgh_last_error": (ctypes.c_char_p, []),
    "lgh_version": (ctypes.c_char_p, []),
    "lgh_create": (_I, [ctypes.POINTER(LghConfig), ctypes.POINTER(_P)]),
    "lgh_destroy": (_I, [_P]),
    "lgh_sync": (_I, [_P]),
    "lgh_stream": (_P, [_P]),
    "lgh_qdata_stressJinvT": (_P, [_P]),
    "lgh_qdata_Jac0inv": (_P, [_P]),
    "lgh_qdata_rho0DetJ0w": (_P, [_P]),
    "lgh_mass_D": (_P, [_P]),
    "lgh_mass_diag": (_P, [_P]),
    "lgh_set_h0": (_I, [_P, _D]),
    "lgh_get_h0": (_I, [_P, c_dbl_p]),
    "lgh_set_dt_est": (_I, [_P, _D]),
    "lgh_get_dt_est": (_I, [_P, c_dbl_p]),
    "lgh_setup_rho0detj0": (_I, [_P, _P, _P, _P, c_dbl_p]),
    "lgh_force_mult": (_I, [_P, _P, _P]),
    "lgh_force_mult_transpose": (_I, [_P, _P, _P]),
    "lgh_mass_set_essential_tdofs": (_I, [_P, _I]),
    "lgh_mass_eliminate_rhs": (_I, [_P, _P]),
    "lgh_mass_mult": (_I, [_P, _I, _P, _P]),
    "lgh_mass_mult_full": (_I, [_P, _I, _P, _P]),
    "lgh_cg_solve": (_I, [_P, _I, _P, _P, _D, _I, c_int_p]),
    "lgh_qupdate": (_I, [_P, _P]),
    "lgh_solve_velocity": (_I, [_P, _P, _P, _P, _P, _P, _D, _I, c_int_p]),
    "lgh_solve_energy": (_I, [_P, _P, _P, _P, _P, _P, _D, _I, c_int_p]),
    "lgh_solve_energy_begin": (_I, [_P, _P, _P, _P, _P, _P, _D, _I]),
    "lgh_tg_source_2d": (_I, [_P, _P, _P]),
    "lgh_set_velocity_source": (_I, [_P, _P]),
    "lgh_solve_energy_end": (_I, [_P, c_int_p]),
    "lgh_vec_set": (_I, [_P, _P, _D, _L]),
    "lgh_vec_copy": (_I, [_P, _P, _P, _L]),
    "lgh_vec_axpby": (_I, [_P, _P, _D, _P, _D, _P, _L]),
    "lgh_vec_axpby_pair": (_I, [_P, _P, _D, _P, _D, _P, _D, _P, _D, _P, _L]),
    "lgh_vec_dot": (_I, [_P, _P, _P, _L, c_dbl_p]),
    "lgh_internal_energy": (_I, [_P, _P, c_dbl_p]),
    "lgh_kinetic_energy": (_I, [_P, _P, c_dbl_p]),
    "lgh_sedov_setup": (_I, [_I, _D, _D, _D, _D, c_dbl_p]),
    "lgh_sedov_shock": (_I, [c_dbl_p, _D, c_dbl_p]),
    "lgh_sedov_eval_point": (_I, [c_dbl_p, _D, _D, c_dbl_p, c_dbl_p, c_dbl_p]),
    "lgh_sedov_eval": (_I, [_P, c_dbl_p, _D, _L, _P, _P, _P, _P]),
    "lgh_compute_density": (_I, [_P, _P, _P]),
    "lgh_sedov_density_error": (_I, [_P, _P, _P, c_dbl_p, _D, c_dbl_p, _I, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]),
    "lgh_get_timers": (_I, [_P, c_dbl_p, ctypes.POINTER(ctypes.c_long)]),
    "lgh_reset_timers": (_I, [_P]),
    "lgh_enable_timers": (_I, [_P, _I]),
    "lgh_ktime_begin": (_I, [_P, _I, _I]),
    "lgh_ktime_end": (_I, [_P, c_int_p, c_dbl_p]),
    "lgh_table_symmetry": (_I, [_P, c_int_p, c_int_p]),
    "lgh_k1_form": (_I, [_P, c_int_p]),
    "lgh_l2_mass_form": (_I, [_P, c_int_p, c_int_p]),
    "lgh_vcg_layout_stats": (_I, [_P, ctypes.POINTER(ctypes.c_long)]),
    "lgh_energy_lockstep_stats": (_I, [_P, ctypes.POINTER(ctypes.c_long)]),
    "lgh_mesh_order": (_I, [_P, ctypes.POINTER(ctypes.c_long)]),
    "lgh_mesh_order_host": (_I, [_I, _I, _I, _I, c_int_p, c_int_p, c_int_p, ctypes.POINTER(ctypes.c_long)]),
    "lgh_mass_data_form": (_I, [_P, c_int_p]),
    "lgh_jac0inv_form": (_I, [_P, c_int_p]),
    "lgh_mass_data_changed": (_I, [_P]),
    "lgh_comm_stats": (_I, [_P, c_int_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), c_int_p, c_int_p]),
    "lgh_qupdate_set_tiny_grad": (_I, [_P, _D]),
    "lgh_set_fused_forces": (_I, [_P, _I]),
    "lgh_qupdate_store_stress": (_I, [_P, _I]),
    "lgh_qupdate_stores_stress": (_I, [_P, c_int_p]),
    "lgh_qupdate_form": (_I, [_P, c_int_p]),
    "lgh_reset_quadrature_data": (_I, [_P]),
    "lgh_fused_force_mult": (_I, [_P, _P]),
    "lgh_fused_force_mult_transpose": (_I, [_P, _P]),
    "lgh_quadrature_generation": (_I, [_P, ctypes.POINTER(ctypes.c_ulong), c_int_p, c_int_p]),
    "lgh_get_fused_forces": (_I, [_P, c_int_p, c_int_p]),
    "lgh_comm_unique_id": (_I, [ctypes.c_char_p]),
    "lgh_comm_unique_id_shm": (_I, [ctypes.c_char_p]),
    "lgh_comm_init": (_I, [_P, _I, _I, ctypes.c_char_p]),
    "lgh_comm_set_neighbors": (_I, [_P, _I, c_int_p, c_int_p, ctypes.POINTER(c_int_p)]),
    "lgh_groups_to_neighbors": (_I, [_I, _I, _I, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p,
                                     c_int_p, _I, c_int_p, _L]),
    "lgh_halo_sum": (_I, [_P, _P, _I]),
    "lgh_allreduce": (_I, [_P, c_dbl_p, _I]),
    "lgh_force_mult_E": (_I, [_P, _P, _P, _P]),
    "lgh_force_mult_transpose_E": (_I, [_P, _P, _P, _P]),
    "lgh_mass_apply_E": (_I, [_P, _I, _P, _P]),
    "lgh_test_vcg_k1": (_I, [_P, _P, _P, c_dbl_p, c_dbl_p, _I, _P, c_dbl_p]),
    "lgh_test_vcg_merged_faces": (_I, [_P, _P, ctypes.POINTER(ctypes.c_long)]),
    "lgh_test_vcg_k2": (_I, [_P, _I, _P, _P, _P, _P, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "lgh_test_set_rank": (_I, [_P, _I, _I]),
    "lgh_test_word_peers": (_I, [_P, _I, ctypes.POINTER(ctypes.c_long)]),
    "lgh_test_halo_pack": (_I, [_P, _P, _I, _P]),
    "lgh_test_halo_combine": (_I, [_P, _P, _P, _I]),
    "lgh_test_rccl_self_sendrecv": (_I, [_P, _I, c_dbl_p]),
    "lgh_test_eig": (_I, [_P, _I, _I, _P, _P, _P]),
    "lgh_test_singular": (_I, [_P, _I, _I, _P, _P]),
    "lgh_test_sqrt": (_I, [_P, _I, _P, _P]),
}

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
    """Load liblaghos_hip.so; raises RuntimeError when the extension is missing."""
    global _lib
    if _lib is None:
        _preload_torch_hip()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and header diverge
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class LghError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise LghError(f"laghos_hip error {rc}: {load().lgh_last_error().decode()}")

"""The C-ABI library loads without a GPU and exports every symbol that
include/laghos_hip.h declares (and nothing is bound that the header lacks)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "laghos_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lgh_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_header():
    from laghos_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in laghos_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "laghos_amd/_lib.py and the header diverge"
    _lib.load()


def test_no_gpu_fails_loudly():
    """Without a device the product path must refuse to run (no CPU fallback)."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        return
    from laghos_amd import _lib
    L = _lib.load()
    cfg = _lib.LghConfig()
    one_i = np.zeros(8, np.int32)
    one_d = np.ones(8)
    ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    cfg.dim, cfg.NE, cfg.D1D, cfg.Q1D, cfg.L1D, cfg.N = 3, 1, 2, 2, 1, 8
    cfg.h1_map = one_i.ctypes.data_as(ip)
    for f in ("B_h1", "G_h1", "B_l2", "weights", "gamma"):
        setattr(cfg, f, one_d.ctypes.data_as(dp))
    h = ctypes.c_void_p()
    rc = L.lgh_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0
    assert b"no HIP device" in L.lgh_last_error() or b"HIP" in L.lgh_last_error()


def test_host_library_loads():
    from laghos_amd import host_lib
    host_lib.load()


def test_product_does_not_touch_oracle():
    """The product (laghos_amd/, bench.py outside cpu_baseline) never imports oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "laghos_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert 'oracle/' not in txt or f.endswith(".md"), f

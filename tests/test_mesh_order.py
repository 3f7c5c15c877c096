"""The library's own zone order and node numbering (laghos_amd/csrc/lgh_order.hip, lgh_mesh_order_host): found from the
element -> node map alone, by face adjacency.  Host analysis only - no GPU.  Whatever numbering the caller hands over
(this repository's lexicographic generator, an MFEM-like one, a random one), the INTERNAL map - zones in internal order,
nodes by internal number - is the same: the one of the lexicographic generator, i.e. x-chains of zones and consecutive
x-rows of nodes, which is what the fast paths of the velocity solve are built on."""
import ctypes

import numpy as np
import pytest

from laghos_amd import _lib, host_lib


def order_of(h1map, NE, N, D, dim=3):
    L = _lib.load()
    hm = np.ascontiguousarray(h1map, dtype=np.int32)
    z, n = np.empty(NE, np.int32), np.empty(N, np.int32)
    out = (ctypes.c_long * 8)()
    ip = ctypes.POINTER(ctypes.c_int)
    _lib.check(L.lgh_mesh_order_host(dim, NE, N, D, hm.ctypes.data_as(ip), z.ctypes.data_as(ip), n.ctypes.data_as(ip), out))
    return z.astype(np.int64), n.astype(np.int64), dict(structured=bool(out[0]), identity=bool(out[1]), components=int(out[2]), extent=tuple(out[3:6]))


def internal_map(h1map, z, n, NE, ND):
    return n[np.asarray(h1map, dtype=np.int64).reshape(NE, ND)[z]]


CASES = [("cube01_hex", 2, 3), ("cube01_hex", 1, 2), ("box01_hex", 1, 3), ("box01_hex", 0, 5), ("cube01_hex", 2, 1)]


@pytest.mark.parametrize("mesh,rs,ok", CASES)
def test_lexicographic_generator_is_already_in_internal_order(mesh, rs, ok):
    d = host_lib.host_disc(mesh, rs, ok, ok - 1 if ok > 1 else 0, 1)
    NE, N = d["gamma"].size, d["owner"].size
    z, n, st = order_of(d["h1map"], NE, N, ok + 1)
    assert st["structured"] and st["identity"] and st["components"] == 1
    assert np.array_equal(z, np.arange(NE)) and np.array_equal(n, np.arange(N))


@pytest.mark.parametrize("mesh,rs,ok", CASES)
@pytest.mark.parametrize("mode", ["mfem", "random"])
def test_any_numbering_leads_to_the_same_internal_map(mesh, rs, ok, mode):
    ot = ok - 1 if ok > 1 else 0
    base = host_lib.host_disc(mesh, rs, ok, ot, 1)
    d = host_lib.host_disc(mesh, rs, ok, ot, 1, renumber=mode, seed=3)
    NE, N = base["gamma"].size, base["owner"].size
    ND = (ok + 1) ** 3
    z, n, st = order_of(d["h1map"], NE, N, ok + 1)
    assert st["structured"] and not st["identity"] and st["components"] == 1
    assert np.array_equal(np.sort(z), np.arange(NE)) and np.array_equal(np.sort(n), np.arange(N))
    # the internal map is the generator's: zone i of the internal order is the generator's zone i, node by node
    assert np.array_equal(internal_map(d["h1map"], z, n, NE, ND), base["h1map"].reshape(NE, ND))
    # ... and the permutations found are the inverses of the ones the renumbering applied
    assert np.array_equal(d["elem_perm"][z], np.arange(NE))
    assert np.array_equal(n[d["node_perm"]], np.arange(N))


def test_two_blocks_that_do_not_touch_are_two_components():
    base = host_lib.host_disc("cube01_hex", 1, 2, 1, 1)
    NE, N, ND = base["gamma"].size, base["owner"].size, 27
    hm = base["h1map"].reshape(NE, ND).astype(np.int64)
    two = np.concatenate([hm, hm + N])          # a second copy of the mesh with its own nodes
    rng = np.random.default_rng(5)
    zp, npm = rng.permutation(2 * NE), rng.permutation(2 * N)
    z, n, st = order_of(npm[two[zp]], 2 * NE, 2 * N, 3)
    assert st["structured"] and st["components"] == 2
    im = internal_map(npm[two[zp]], z, n, 2 * NE, ND)
    # each component comes out as the generator's block, one after the other (which one first depends on the zone order)
    assert np.array_equal(im[:NE], hm) and np.array_equal(im[NE:], hm + N)


def test_a_mesh_that_wraps_around_keeps_the_callers_order():
    """periodic in x (the last zone of every x-row shares its high face with the first one's low face): integer coordinates
    cannot be consistent - the analysis says so and leaves the caller's numbering alone (general path)."""
    base = host_lib.host_disc("cube01_hex", 1, 2, 1, 1)   # 4 x 4 x 4 zones, Q2: 9^3 nodes
    NE, ND, nn = 64, 27, 9
    hm = base["h1map"].reshape(NE, ND).astype(np.int64)
    ix = hm % nn
    wrapped = np.where(ix == nn - 1, hm - (nn - 1), hm)    # identify x = 1 with x = 0
    used = np.unique(wrapped)
    renum = np.full(nn ** 3, -1)
    renum[used] = np.arange(used.size)
    z, n, st = order_of(renum[wrapped], NE, used.size, 3)
    assert not st["structured"] and st["identity"]
    assert np.array_equal(z, np.arange(NE)) and np.array_equal(n, np.arange(used.size))


def test_two_dimensions_keep_the_callers_order():
    d = host_lib.host_disc("square01_quad", 2, 2, 1, 1)
    NE, N = d["gamma"].size, d["owner"].size
    z, n, st = order_of(d["h1map"], NE, N, 3, dim=2)
    assert st["identity"] and np.array_equal(z, np.arange(NE))


def test_an_l_shaped_block_with_a_hole_is_still_one_structured_component():
    """Not a box: an L-shaped domain (a quarter of the 4 x 4 x 4 mesh removed).  The flood fill still gives consistent integer
    coordinates; zones come out row by row (rows of different length), nodes along those rows - every x-row of every zone is
    consecutive in the internal numbering, and zones that follow each other inside a row are x-neighbours."""
    base = host_lib.host_disc("cube01_hex", 1, 3, 2, 1)      # 4 x 4 x 4 zones, Q3
    NE0, ND, n = 64, 64, 4
    hm0 = base["h1map"].reshape(NE0, ND).astype(np.int64)
    e = np.arange(NE0)
    keep = ~((e % n >= 2) & ((e // n) % n >= 2))              # drop i >= 2 and j >= 2
    hm = hm0[keep]
    used = np.unique(hm)
    renum = np.full(hm0.max() + 1, -1)
    renum[used] = np.arange(used.size)
    rng = np.random.default_rng(9)
    zp, npm = rng.permutation(hm.shape[0]), rng.permutation(used.size)
    given = npm[renum[hm]][zp]
    z, nn, st = order_of(given, hm.shape[0], used.size, 4)
    assert st["structured"] and st["components"] == 1 and not st["identity"]
    im = internal_map(given, z, nn, hm.shape[0], ND).reshape(-1, 4, 4, 4)      # [zone][dz][dy][dx]
    assert np.all(np.diff(im, axis=3) == 1)                                     # x-rows of nodes are consecutive numbers
    chains = np.all(im[:-1, :, :, 3] == im[1:, :, :, 0], axis=(1, 2))           # zone i + 1 is the x-neighbour of zone i
    assert chains.sum() == (3 * 2 + 1 * 2) * 4                                  # rows of 4 zones (j < 2) and of 2 zones (j >= 2), per k

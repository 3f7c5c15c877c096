"""The operators on a mesh whose numbering has no structure: random renumbering of the H1 nodes, random order of the zones
(tests/helpers.py::PermutedProblem).  Every mesh of BASELINE.json comes out of this repository's own Cartesian generator -
lexicographic nodes, zones x-fastest; what the reference hands its operators is MFEM's numbering (vertices, then edge /
face / interior dofs; `H1.GetElementRestriction`, laghos_assembly.cpp:557-565), which has none of that structure.
Round 6: the library finds the structure of the MESH itself (lgh_order.hip: face adjacency of the element -> node map) and
runs the velocity solve in its own zone order and node numbering, so the fast paths (x-chains of zones -> merged E-vector,
consecutive x-rows of nodes -> row loads) are taken on the permuted mesh as well - `order` = "own"; `order` = "callers"
(LGH_ORDER=0) keeps the caller's numbering: the general path of every kernel, as in rounds 1-5.  Both against the oracle on
the same permuted problem (which knows nothing but the element -> node map), and against the un-permuted run - the
discrete problem is the same, so `|e|`, the time step and the state agree to round-off (sums run in another order)."""
import numpy as np
import pytest

from helpers import CurvedInitialMesh, PermutedProblem, deformed_state, make_gpu, make_oracle, rel_err, seeded

pytestmark = pytest.mark.gpu

CASES = [
    # (id, problem kwargs, LGH_VCG_VARIANT)
    ("3D-Q3Q2-512-default", dict(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1), None),
    ("3D-Q3Q2-512-slab", dict(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1), "4"),
    ("3D-Q2Q1-512", dict(mesh="cube01_hex", rs=2, order_v=2, order_e=1, problem=1), None),
    ("3D-Q4Q3-16-kron", dict(mesh="box01_hex", rs=0, order_v=4, order_e=3, problem=3), None),
    ("2D-Q3Q2-64", dict(mesh="square01_quad", rs=2, order_v=3, order_e=2, problem=1), None),
]


@pytest.mark.parametrize("order", ["own", "callers"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_rhs_on_a_permuted_mesh_vs_oracle_and_vs_the_structured_mesh(case, order, monkeypatch):
    """One evaluation of dS/dt (quadrature update, both force products, three H1 solves, the L2 solve; CG to 1e-13) on a
    distorted state: HIP path on the permuted problem against (a) the oracle on the same permuted problem, (b) the HIP path
    on the structured problem, mapped through the permutation."""
    from oracle.fem import Problem
    _, kw, variant = case
    monkeypatch.delenv("LGH_VCG_VARIANT", raising=False)
    monkeypatch.delenv("LGH_ORDER", raising=False)
    if variant is not None:
        monkeypatch.setenv("LGH_VCG_VARIANT", variant)
    if order == "callers":
        monkeypatch.setenv("LGH_ORDER", "0")
    base = Problem(**kw)
    perm = PermutedProblem(base, seed=11)
    S_b = deformed_state(base, seed=23)
    S_p = perm.state(S_b)
    H1V = base.H1V

    def rhs_gpu(prob, S):
        g = make_gpu(prob, cg_tol=1e-13)
        try:
            form = g.ctx.k1_form()
            mo = g.ctx.mesh_order()
            assert mo["identity"] == (prob is base or order == "callers" or prob.dim == 2), mo
            _, n_merged = g.ctx.test_vcg_merged_faces() if prob.dim == 3 else (None, 0)
            Sd, dS = g.ctx.to_dev(S), g.ctx.zeros(S.size)
            g.reset_quadrature_data()
            g.reset_time_step_estimate()
            g.mult(Sd, dS)
            dt = g.get_time_step_estimate(Sd)
            g.ctx.sync()
            return dS.cpu().numpy(), dt, form, n_merged
        finally:
            g.close()

    dS_p, dt_p, form_p, merged_p = rhs_gpu(perm, S_p)
    dS_b, dt_b, form_b, merged_b = rhs_gpu(base, S_b)
    assert form_p == form_b
    if variant == "4":
        # the library's own zone order finds the same x-chains on the permuted mesh; in the caller's order the zones of a set are
        # no neighbours: nothing to merge
        assert form_p == "slab" and merged_b > 0 and merged_p == (merged_b if order == "own" else 0)
    o = make_oracle(perm, cg_tol=1e-13)
    try:
        dS_o = np.empty_like(S_p)
        o.qdata_is_current = False
        o.reset_time_step_estimate()
        o.mult(S_p, dS_o)
        dt_o = o.get_time_step_estimate(S_p)
    finally:
        o.close()
    for name, sl in (("dv", slice(H1V, 2 * H1V)), ("de", slice(2 * H1V, None))):
        assert rel_err(dS_p[sl], dS_o[sl]) < 1e-9, (name, "vs the oracle on the permuted mesh")
        assert rel_err(dS_p[sl], perm.state(dS_b)[sl]) < 1e-9, (name, "vs the structured mesh")
    assert np.array_equal(dS_p[:H1V], S_p[H1V:2 * H1V])  # dx/dt = v
    assert abs(dt_p - dt_o) <= 1e-12 * dt_o and abs(dt_p - dt_b) <= 1e-12 * dt_b


@pytest.mark.parametrize("order", ["own", "callers"])
def test_k1_and_mass_operators_on_a_permuted_mesh(order, monkeypatch):
    """The mass operators alone, kernel by kernel, on the permuted 512-zone Q3Q2 mesh: MassPAOperator::Mult on both spaces
    and one launch of the slab K1 vs the oracle - in the caller's order (node gathers instead of row loads, element-local
    E-vector in every set) and in the library's own (row loads, merged sets; the hook speaks the caller's numbering on both
    sides: vectors and E-vector go through the permutation)."""
    from oracle.driver import _dp
    from oracle.fem import Problem
    base = Problem(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1)
    perm = PermutedProblem(base, seed=13)
    monkeypatch.setenv("LGH_VCG_VARIANT", "4")
    monkeypatch.delenv("LGH_ORDER", raising=False)
    if order == "callers":
        monkeypatch.setenv("LGH_ORDER", "0")
    g, o = make_gpu(perm), make_oracle(perm)
    try:
        N, NE, ND = perm.N, perm.NE, perm.ND
        xv, xe = seeded(N, 301), seeded(perm.L2V, 302)
        yv, ye = g.ctx.empty(N), g.ctx.empty(perm.L2V)
        g.ctx.mass_set_ess(-1)
        g.ctx.mass_mult(0, g.ctx.to_dev(xv), yv)
        g.ctx.mass_mult(1, g.ctx.to_dev(xe), ye)
        g.ctx.sync()
        assert rel_err(yv.cpu().numpy(), o.mass_mult(0, xv)) < 2e-12
        assert rel_err(ye.cpu().numpy(), o.mass_mult(1, xe)) < 2e-12
        r, d_old = seeded(3 * N, 303), seeded(3 * N, 304)
        dinv = 1.0 / np.asarray(o.diagV)
        rz = np.array([float(np.dot(r[c * N:(c + 1) * N] ** 2, dinv)) for c in range(3)])
        rz_prev = rz * np.array([1.7, 0.6, 1.1])
        yE, den = g.ctx.test_vcg_k1(g.ctx.to_dev(r), g.ctx.to_dev(d_old), rz, rz_prev, False)
        yE = yE.cpu().numpy()
        mask, n_merged = g.ctx.test_vcg_merged_faces()
        assert (n_merged > 0) == (order == "own")
        mask = np.asarray(mask).reshape(NE, ND).astype(bool)
        hmap = np.asarray(perm.h1map).reshape(NE, ND)
        # zone -> its left x-neighbour (the zone whose dx = 3 face is this zone's dx = 0 face), for the merged pairs
        left = {tuple(hmap[e].reshape(4, 4, 4)[:, :, 3].ravel()): e for e in range(NE)}
        for c in range(3):
            d = r[c * N:(c + 1) * N] * dinv + (rz[c] / rz_prev[c]) * d_old[c * N:(c + 1) * N]
            xE = np.ascontiguousarray(d[hmap].reshape(-1))
            yE_o = np.empty(NE * ND)
            o.L.lgo_mass_apply_E(o.h, 0, _dp(xE), _dp(yE_o))
            exp = yE_o.reshape(NE, 4, 4, 4).copy()   # [e][dz][dy][dx]
            if n_merged:
                # K1 reports the sum of a merged pair in the LEFT zone's dx = 3 entry and 0.0 in the right zone's dx = 0 entry
                m4 = mask.reshape(NE, 4, 4, 4)
                for e in range(NE):
                    if not m4[e, :, :, 0].any():
                        continue
                    l = left[tuple(hmap[e].reshape(4, 4, 4)[:, :, 0].ravel())]
                    sel = m4[e, :, :, 0]
                    exp[l, :, :, 3][sel] += exp[e, :, :, 0][sel]
                    exp[e, :, :, 0][sel] = 0.0
            assert rel_err(yE[c], exp.reshape(-1)) < 2e-12, c
            assert abs(den[c] - float(np.dot(xE, yE_o))) <= 2e-12 * abs(den[c]), c
    finally:
        g.close()
        o.close()


def test_time_steps_on_a_permuted_mesh_reproduce_the_structured_run():
    """Ten RK4 steps of 3D Sedov Q3Q2 (512 zones) from t = 0 with the real dt controller: the permuted mesh takes the same
    steps with the same dt and ends at the same |e| as the structured one (round-off), and at the oracle's."""
    from laghos_amd.hydro import TimeLoop
    from oracle.fem import Problem
    base = Problem(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1)
    perm = PermutedProblem(base, seed=17)
    out = {}
    for name, prob in (("structured", base), ("permuted", perm)):
        g = make_gpu(prob)
        try:
            loop = TimeLoop(g, t_final=1e9)
            for _ in range(10):
                assert loop.step()
            out[name] = (loop.t, loop.dt, loop.steps, g.e_norm(loop.S))
        finally:
            g.close()
    (t_s, dt_s, ti_s, e_s), (t_p, dt_p, ti_p, e_p) = out["structured"], out["permuted"]
    assert ti_s == ti_p  # (accepted + repeated steps)
    assert abs(t_p - t_s) <= 1e-11 * t_s and abs(dt_p - dt_s) <= 1e-10 * dt_s
    assert abs(e_p - e_s) <= 1e-9 * e_s


@pytest.mark.parametrize("kw,variant", [(dict(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1), "4"),
                                        (dict(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1), None),
                                        (dict(mesh="box01_hex", rs=0, order_v=4, order_e=3, problem=3), None)],
                         ids=["Q3Q2-512-slab", "Q3Q2-512-default", "Q4Q3-16"])
def test_rhs_on_a_curved_initial_mesh_vs_oracle(kw, variant, monkeypatch):
    """A curved initial mesh (tests/helpers.py::CurvedInitialMesh): Jac0inv varies inside every zone and the mass data is
    not W[q] s_e - the library must FIND that (lgh_jac0inv_form, lgh_mass_data_form) and run the stored-data paths of the
    quadrature update, of K1 and of the L2 mass apply; dS/dt and the time-step estimate against the oracle on the same mesh.
    The same problem on the Cartesian mesh finds both compact forms."""
    from oracle.fem import Problem
    monkeypatch.delenv("LGH_VCG_VARIANT", raising=False)
    if variant is not None:
        monkeypatch.setenv("LGH_VCG_VARIANT", variant)
    base = Problem(**kw)
    curved = CurvedInitialMesh(base)
    g0 = make_gpu(base)
    try:
        assert g0.ctx.jac0inv_form() == "compact" and g0.ctx.mass_data_form() == "rank1"
    finally:
        g0.close()
    S0 = curved.initial_state()[0]
    rng = np.random.default_rng(31)
    S = S0.copy()
    H1V = base.H1V
    S[H1V:2 * H1V] = rng.uniform(-1, 1, H1V)
    S[2 * H1V:] = 1.0 + 0.5 * rng.uniform(-1, 1, base.L2V)
    # (iteration cap lifted: the unpreconditioned L2 solve on a distorted order-3 zone needs more than the reference's 300
    #  to reach 1e-13, and two solves cut off unconverged differ by what is left of the error, not by round-off)
    g, o = make_gpu(curved, cg_tol=1e-13, cg_max_iter=5000), make_oracle(curved, cg_tol=1e-13, cg_max_iter=5000)
    try:
        assert g.ctx.jac0inv_form() == "stored" and g.ctx.mass_data_form() == "stored"
        Sd, dS = g.ctx.to_dev(S), g.ctx.zeros(S.size)
        g.reset_quadrature_data()
        g.reset_time_step_estimate()
        g.mult(Sd, dS)
        dt = g.get_time_step_estimate(Sd)
        g.ctx.sync()
        dS = dS.cpu().numpy()
        dS_o = np.empty_like(S)
        o.qdata_is_current = False
        o.reset_time_step_estimate()
        o.mult(S, dS_o)
        dt_o = o.get_time_step_estimate(S)
    finally:
        g.close()
        o.close()
    assert rel_err(dS[H1V:2 * H1V], dS_o[H1V:2 * H1V]) < 1e-9 and rel_err(dS[2 * H1V:], dS_o[2 * H1V:]) < 1e-9
    assert abs(dt - dt_o) <= 1e-12 * dt_o


@pytest.mark.parametrize("mode", ["mfem", "random"])
def test_cpp_driver_on_a_renumbered_mesh_reproduces_the_structured_run(mode):
    """`laghos -renumber mfem|random` (the legs c2mfem / c2perm of bench.py at a small size): eight RK4 steps of 3D Sedov Q3Q2
    on 512 zones through the C++ host layer, in an MFEM-like / a random numbering of nodes and zones - same steps, same dt,
    same |e| and the same state (through the permutation) as in the generator's own numbering."""
    from laghos_amd import host_lib
    common = ["-m", "data/cube01_hex.mesh", "-rs", 2, "-p", 1, "-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", 8, "-vs", 10 ** 9, "-q"]
    out = {}
    for name, extra in (("base", []), (mode, ["-renumber", mode, "-renumber-seed", 9])):
        sim = host_lib.Sim(common + extra)
        try:
            sim.enable_timers(False)
            while sim.step() == 1:
                pass
            sim.sync()
            out[name] = (sim.t, sim.dt, sim.rk_steps, sim.e_norm(), sim.state())
        finally:
            sim.close()
    d = host_lib.host_disc("cube01_hex", 2, 3, 2, 1, renumber=mode, seed=9)
    npm, epm = d["node_perm"].astype(np.int64), d["elem_perm"].astype(np.int64)
    (t_b, dt_b, n_b, e_b, S_b), (t_r, dt_r, n_r, e_r, S_r) = out["base"], out[mode]
    assert n_b == n_r and abs(t_r - t_b) <= 1e-11 * t_b and abs(dt_r - dt_b) <= 1e-10 * dt_b
    assert abs(e_r - e_b) <= 1e-9 * e_b
    N, NE = npm.size, epm.size
    H1V, NL = 3 * N, 27
    for b in range(6):
        assert rel_err(S_r[b * N:(b + 1) * N][npm], S_b[b * N:(b + 1) * N]) < 1e-8, b
    assert rel_err(S_r[2 * H1V:].reshape(NE, NL), S_b[2 * H1V:].reshape(NE, NL)[epm]) < 1e-8


@pytest.mark.parametrize("mode", ["mfem", "random"])
def test_config2_full_size_on_a_renumbered_mesh(mode):
    """BASELINE configs[1] at FULL size (32^3 zones: the slab K1 by default dispatch, the bounded-grid K2, the row-form update)
    in an MFEM-like / a random numbering against the same run in the generator's numbering - which
    tests/test_gpu_pipeline.py::test_config2_full_size_vs_oracle holds to the oracle: three RK4 steps from t = 0, same step
    count and dt, |e| to 1e-9, the state through the permutation to 1e-8; the library must have found the block (lgh_mesh_order:
    structured, not the identity, 32 x 32 x 32) and merged the same x-faces."""
    import ctypes
    from laghos_amd import _lib, host_lib
    common = ["-m", "data/cube01_hex.mesh", "-rs", 4, "-p", 1, "-ok", 3, "-ot", 2, "-pa", "-tf", 1e9, "-ms", 3, "-vs", 10 ** 9, "-q"]
    out = {}
    for name, extra in (("base", []), (mode, ["-renumber", mode, "-renumber-seed", 4])):
        sim = host_lib.Sim(common + extra)
        try:
            sim.enable_timers(False)
            while sim.step() == 1:
                pass
            sim.sync()
            L, ctx = _lib.load(), sim.L.laghos_sim_context(sim.h)
            o, st = (ctypes.c_long * 8)(), (ctypes.c_long * 4)()
            _lib.check(L.lgh_mesh_order(ctx, o))
            _lib.check(L.lgh_vcg_layout_stats(ctx, st))
            form = ctypes.c_int(-9)
            _lib.check(L.lgh_k1_form(ctx, ctypes.byref(form)))
            out[name] = (sim.t, sim.dt, sim.rk_steps, sim.e_norm(), sim.state(), tuple(o[:6]), int(st[3]), form.value)
        finally:
            sim.close()
    (t_b, dt_b, n_b, e_b, S_b, o_b, merged_b, form_b), (t_r, dt_r, n_r, e_r, S_r, o_r, merged_r, form_r) = out["base"], out[mode]
    assert o_b == (1, 1, 1, 32, 32, 32) and o_r == (1, 0, 1, 32, 32, 32)
    assert form_b == form_r == 4 and merged_r == merged_b > 0
    assert n_b == n_r and abs(t_r - t_b) <= 1e-11 * t_b and abs(dt_r - dt_b) <= 1e-9 * dt_b
    assert abs(e_r - e_b) <= 1e-9 * e_b
    d = host_lib.host_disc("cube01_hex", 4, 3, 2, 1, renumber=mode, seed=4)
    npm, epm = d["node_perm"].astype(np.int64), d["elem_perm"].astype(np.int64)
    N, NE = npm.size, epm.size
    H1V, NL = 3 * N, 27
    for b in range(6):
        assert rel_err(S_r[b * N:(b + 1) * N][npm], S_b[b * N:(b + 1) * N]) < 1e-8, b
    assert rel_err(S_r[2 * H1V:].reshape(NE, NL), S_b[2 * H1V:].reshape(NE, NL)[epm]) < 1e-8


@pytest.mark.parametrize("name,args", [
    ("README-4", ["-p", 1, "-m", "data/cube01_hex.mesh", "-rs", 2, "-tf", 0.6, "-E0", 2]),
    ("README-7", ["-p", 3, "-m", "data/box01_hex.mesh", "-rs", 1, "-tf", 5.0]),
    ("README-2", ["-p", 0, "-m", "data/cube01_hex.mesh", "-rs", 1, "-tf", 0.75]),
])
def test_published_3d_runs_in_mfem_numbering(name, args, golden):
    """The reference's published 3D runs (README.md:216, :218, :221 with the values of :228, :230, :233) from t = 0 to t_final
    through the C++ driver with the mesh handed over in an MFEM-like numbering (`-renumber mfem`): the velocity solve runs in
    the library's own order, the final step count, the printed dt and `|e|` are the published ones - the golden values do not
    know how the nodes are numbered."""
    from laghos_amd import host_lib
    g = next(c for c in golden["readme"] if c["name"] == name)
    sim = host_lib.Sim(args + ["-pa", "-q", "-renumber", "mfem"])
    try:
        import ctypes
        from laghos_amd import _lib
        o = (ctypes.c_long * 8)()
        _lib.check(_lib.load().lgh_mesh_order(sim.L.laghos_sim_context(sim.h), o))
        assert o[0] == 1 and o[1] == 0  # a structured block was found, and its order is not the caller's
        while sim.step() == 1:
            pass
        e, ti, dt = sim.e_norm(), sim.ti, sim.dt
    finally:
        sim.close()
    assert ti == g["step"]
    assert f"{dt:.6f}" == g["dt"]
    assert abs(e - g["e_norm"]) / g["e_norm"] < 1e-9, (e, g["e_norm"])


@pytest.mark.parametrize("variant", ["4", None], ids=["slab", "default"])
def test_rhs_on_two_blocks_that_share_no_node(variant, monkeypatch):
    """A mesh of two structured blocks (tests/helpers.py::TwoBlocks: two copies of the 512-zone Q3Q2 cube side by side, no
    common node), nodes and zones randomly renumbered across BOTH: the library must find two components (lgh_mesh_order), put
    them one after the other in its own order - sets of five zones that straddle the two are no x-chains and keep the
    element-local layout - and the right-hand side of a differently distorted state on each block must be the oracle's."""
    from helpers import TwoBlocks
    from oracle.fem import Problem
    monkeypatch.delenv("LGH_VCG_VARIANT", raising=False)
    monkeypatch.delenv("LGH_ORDER", raising=False)
    if variant is not None:
        monkeypatch.setenv("LGH_VCG_VARIANT", variant)
    base = Problem(mesh="cube01_hex", rs=2, order_v=3, order_e=2, problem=1)
    two = TwoBlocks(base)
    perm = PermutedProblem(two, seed=21)
    S = perm.state(two.state(deformed_state(base, seed=5), deformed_state(base, seed=6)))
    H1V = two.H1V
    g, o = make_gpu(perm, cg_tol=1e-13), make_oracle(perm, cg_tol=1e-13)
    try:
        mo = g.ctx.mesh_order()
        assert mo["structured"] and not mo["identity"] and mo["components"] == 2 and mo["extent"] == (8, 8, 8)
        if variant == "4":
            _, n_merged = g.ctx.test_vcg_merged_faces()
            assert g.ctx.k1_form() == "slab" and n_merged > 0
        Sd, dS = g.ctx.to_dev(S), g.ctx.zeros(S.size)
        g.reset_quadrature_data()
        g.reset_time_step_estimate()
        g.mult(Sd, dS)
        dt = g.get_time_step_estimate(Sd)
        g.ctx.sync()
        dS = dS.cpu().numpy()
        dS_o = np.empty_like(S)
        o.qdata_is_current = False
        o.reset_time_step_estimate()
        o.mult(S, dS_o)
        dt_o = o.get_time_step_estimate(S)
    finally:
        g.close()
        o.close()
    assert rel_err(dS[H1V:2 * H1V], dS_o[H1V:2 * H1V]) < 1e-9 and rel_err(dS[2 * H1V:], dS_o[2 * H1V:]) < 1e-9
    assert abs(dt - dt_o) <= 1e-12 * dt_o

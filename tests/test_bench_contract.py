"""The bench line committed under profiles/ (the output of `python bench.py` on an MI355X) carries
every field of the driver's contract, with consistent values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _line("r4_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    # BASELINE.json writes the metric with a multiplication sign; "*" in lines produced before that was matched
    assert d["metric"].replace("*", "\u00d7") == base["metric"].split(";")[0].strip()
    assert d["unit"] == "Mdofs*steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                      # BASELINE.json "published": {} - no number for this metric
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic") and d["n_gpus"] == 1
    assert "qupdate_division" in d["config"]  # the one precision relaxation of the path is stated in the line
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = dofs * RK stages / wall time of the timed steps
    c = d["config"]
    dofs = c["h1_dofs"] + c["l2_dofs"]
    wall = d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(d["value"] - 1e-6 * dofs * c["rk_stages_executed"] / wall) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "sec8d_frac", "other_kernels"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac_of_achievable"] - r["achieved"] / r["achievable"]) < 1e-12
    assert "traffic_source" in r  # a counter figure is only reported for the build it was measured on
    # the roofline kernel is the one with the largest share of the sampled RK step
    share = r["time_share_us_per_rk_step"]
    assert r["kernel"].split(" ")[0] == max(share, key=share.get)
    # Honest bytes (round-3 advisor): `achieved` counts what the kernel form that ran has to MOVE; no figure may exceed
    # what the memory system delivered - for the roofline kernel and for every kernel beside it the counter traffic
    # covers the moved bytes - and nothing runs above the HBM peak
    assert r["traffic"] is not None and r["bytes_per_launch"] <= r["traffic"] <= 1.5 * r["bytes_per_launch"]
    assert 0 < r["frac"] < 1
    for name, o in r["other_kernels"].items():
        assert 0 < o["frac"] < 1, name
        assert o["traffic"] is not None and o["bytes_per_launch"] <= o["traffic"], name
    for name, k in d["kernels"].items():
        assert k["GBs"] < r["peak"], name
    # K1 beside it: what it moves is less than SURVEY 8(d)'s figure (compact mass data: no quadrature table)
    k1 = r["other_kernels"]["vcg_apply_slab346"]
    assert k1["bytes_per_launch"] < k1["sec8d_bytes_per_launch"] and k1["frac"] < k1["sec8d_frac"]
    # north_star quotes its target on the Force+Mass operator apply: F.1 + F^T v + the mass applies of the H1 CG, in both accountings
    for key in ("force_mass_aggregate", "force_mass_cg_aggregate"):
        a = r[key]
        assert abs(a["frac"] - a["achieved"] / r["peak"]) < 1e-12
        assert abs(a["achieved"] - 1e-9 * a["bytes_per_rk_step"] / a["seconds_per_rk_step"]) < 1e-6 * a["achieved"]
        assert a["bytes_per_rk_step"] <= a["sec8d_bytes_per_rk_step"] and a["frac"] <= a["sec8d_frac"] < 1
    assert r["force_mass_aggregate"]["kernels"] == ["force_mult_3d", "force_mult_t_3d", "vcg_apply_slab346"]
    # parity block: the bench's own problem against the oracle, printed with the number it belongs to
    assert d["parity"]["pass"] is True and d["parity"]["e_norm_rel_diff"] <= 1e-9 and d["parity"]["rk4_steps"] == 3
    # the other single-GPU configs of BASELINE.json as extra legs: 64^3 Sedov (HBM-resident) and 64^3 Taylor-Green
    for leg in ("c3", "tg"):
        g = d["legs"][leg]
        assert g["elements"] == 262144 and g["value"] > 0 and 0 < g["force_mass_aggregate"]["frac"] < 1
        assert all(k["GBs"] < r["peak"] for k in g["kernels"].values())
        assert g["roofline"]["traffic"] is None or g["roofline"]["bytes_per_launch"] <= g["roofline"]["traffic"]
    assert "Taylor-Green" in d["legs"]["tg"]["workload"] and "-rs 5" in d["legs"]["c3"]["workload"]
    # config 5 (Q5Q4) on one GPU, the developed-flow view of C2, the general-mesh path (stored mass table) and the
    # N-rank code path on one rank
    assert d["legs"]["c5"]["elements"] == 65536 and d["legs"]["c5"]["value"] > 0
    assert d["legs"]["c2dev"]["value"] > 0
    assert "stored" in d["legs"]["c2stored"]["workload"] and 0 < d["legs"]["c2stored"]["value"] <= 1.02 * d["value"]
    m = d["legs"]["c2multi"]
    assert m["comm"]["ranks"] == 1 and m["comm"]["allreduce"]["per_rk_step"] > 0 and abs(m["ms_per_step_minus_single_rank_path"]) < 1.0
    b = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert key in b, key
    assert b["kind"] in ("port", "reference") and b["unit"] == d["unit"] and b["cores"] >= 1


def test_profiled_run_agrees_with_the_plain_run():
    """The same command under rocprofv3 --kernel-trace --stats: same workload, throughput within
    the profiler's overhead."""
    a, b = _line("r4_bench.json"), _line("r4_bench_under_rocprofv3.json")
    assert a["config"]["workload"] == b["config"]["workload"]
    assert 0.8 * a["value"] < b["value"] <= 1.05 * a["value"]


def test_weak_scaling_layout_of_the_bench():
    """N ranks keep 32^3 elements each in a block grid whose ranks are all neighbours of each other
    up to N = 8 (the partitions for which the CG sums ride on the halo exchange)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [tuple(bench.block_grid(n)) for n in (1, 2, 4, 8)] == [(1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2)]
    for n in (1, 2, 3, 4, 6, 8):
        px, py, pz = bench.block_grid(n)
        assert px * py * pz == n
        # the grid is the one laghos::Partition builds for the mesh bench.py hands it (tests/test_host_setup.py
        # holds the Python mirror against the C++ code)
        assert bench.partition_grid((32 * px, 32 * py, 32 * pz), n) == (px, py, pz)
    assert bench.usable_cpus() >= 1

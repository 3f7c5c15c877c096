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
    d = _line("r1_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    # BASELINE.json writes the metric with a multiplication sign; "*" in lines produced before that was matched
    assert d["metric"].replace("*", "\u00d7") == base["metric"].split(";")[0].strip()
    assert d["unit"] == "Mdofs*steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                      # BASELINE.json "published": {} - no number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = dofs * RK stages / wall time of the timed steps
    c = d["config"]
    dofs = c["h1_dofs"] + c["l2_dofs"]
    wall = d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(d["value"] - 1e-6 * dofs * c["rk_stages_executed"] / wall) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    b = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in b, key
    assert b["kind"] in ("port", "reference") and b["unit"] == d["unit"] and b["cores"] >= 1


def test_profiled_run_agrees_with_the_plain_run():
    """The same command under rocprofv3 --kernel-trace --stats: same workload, throughput within
    the profiler's overhead."""
    a, b = _line("r1_bench.json"), _line("r1_bench_under_rocprofv3.json")
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
    assert bench.usable_cpus() >= 1

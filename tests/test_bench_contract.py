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


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _walk(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _walk(v, path + "/" + str(k))
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _walk(v, path + "/%d" % i)
    else:
        yield path, o


def _check_line(d, text):
    """what the driver needs from the ONE stdout line (round-4 verdict: a 23.8 KB line left BENCH_r04.parsed null)"""
    assert len(text) <= 8192, "the driver keeps 8 KB of stdout: the contract line has to fit (%d bytes)" % len(text)
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
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "other_kernels", "mean_launch_us", "bytes_per_launch"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["mean_launch_us"] * 1e-6) / 1e9) < 1e-4 * r["achieved"]
    assert "traffic_source" in r  # a counter figure is only reported for the build it was measured on
    # the roofline kernel is the one with the largest share of the sampled RK step
    share = r["time_share_us_per_rk_step"]
    assert r["kernel"] == max(share, key=share.get)
    assert 0 < r["frac"] < 1
    for name, o in r["other_kernels"].items():
        assert 0 < o["frac"] < 1, name
    # Honest bytes: where counter traffic is quoted (only for the build it was collected on) it covers the moved bytes
    if r["traffic"] is not None:
        assert r["bytes_per_launch"] <= r["traffic"] <= 1.5 * r["bytes_per_launch"]
        for name, o in r["other_kernels"].items():
            assert o["traffic"] is None or o["bytes_per_launch"] <= o["traffic"], name
    # north_star quotes its target on the Force+Mass operator apply: F.1 + F^T v + the mass applies of the H1 CG
    for key in ("force_mass_aggregate", "force_mass_cg_aggregate"):
        a = r[key]
        assert abs(a["frac"] - a["achieved"] / r["peak"]) < 1e-5 and 0 < a["frac"] < 1
        assert abs(a["achieved"] - 1e-9 * a["bytes_per_rk_step"] / a["seconds_per_rk_step"]) < 1e-4 * a["achieved"]
    # No figure anywhere in the line above the HBM peak: no GB/s entry, no fraction > 1 (round-4 verdict, weak #3:
    # SURVEY 8(d)'s bytes of the reference's kernel forms priced at this implementation's times gave 1.42 of peak)
    for path, v in _walk(d):
        leaf = path.rsplit("/", 1)[-1]
        assert not leaf.startswith("sec8d"), path
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            if leaf in ("frac", "force_mass_frac") or leaf.endswith("_frac"):
                assert 0 <= v < 1, (path, v)
            if leaf in ("achieved", "GBs"):
                assert v < 8000.0, (path, v)
    # parity block: the bench's own problem against the oracle, printed with the number it belongs to
    assert d["parity"]["pass"] is True and d["parity"]["e_norm_rel_diff"] <= 1e-9 and d["parity"]["rk4_steps"] == 3
    # the other single-GPU configs of BASELINE.json as extra legs: value, ms_per_step, roofline kernel and fraction only
    for leg in [l for l in ("c2mfem", "c2perm", "c3", "tg", "c5", "c2dev", "c2stored", "c2general", "c2multi", "c2multistored", "c2multigeneral", "c3stored", "c3general") if l in d["legs"]]:
        g = d["legs"][leg]
        assert g["value"] > 0 and g["ms_per_step"] > 0 and 0 < g["frac"] < 1 and g["kernel"], leg
        assert set(g) <= {"value", "ms_per_step", "kernel", "frac", "force_mass_frac", "ms_per_step_minus_single_rank_path", "k_us", "merged_entries"}, leg
    general = "c2general" if "c2general" in d["legs"] else "c2stored"  # (round 6 renamed the general-mesh legs: they also read Jac0inv per point)
    assert 0 < d["legs"][general]["value"] <= 1.02 * d["value"]
    assert abs(d["legs"]["c2multi"]["ms_per_step_minus_single_rank_path"]) < 1.0
    b = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert key in b, key
    assert b["kind"] in ("port", "reference") and b["unit"] == d["unit"] and b["cores"] >= 1


def test_compact_line_of_a_full_record_fits_the_driver():
    """bench.compact_line() is a pure function of the full record: applied to round 4's 23.8 KB record it must give a line
    the driver can keep (<= 8 KB) that still holds the whole contract."""
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "r4_bench.json")) as f:
        full = json.loads([l for l in f.read().splitlines() if l.startswith("{")][0])
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096
    _check_line(json.loads(text), text)
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]  # what the driver's clock is held against: unrounded


def test_committed_line_is_what_the_driver_reads():
    """profiles/r6_bench.json is the stdout line of `python bench.py` on an MI355X (tools/gpu_final_r6.sh), byte for byte what
    the driver's `python bench.py --gpus 1 --steps 20 --warmup 5` prints last: it holds the contract, fits the driver's
    8 KB, quotes counter traffic of this very build (sha over the kernel sources) and is the compact form of the full
    record written beside it (profiles/r6_bench_detail.json)."""
    with open(os.path.join(ROOT, "profiles", "r6_bench.json")) as f:
        text = [l for l in f.read().splitlines() if l.startswith("{")][0]
    d = json.loads(text)
    _check_line(d, text)
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic_source"].startswith("profiles/r6_pmc_traffic.json")
    assert "compact" in d["config"]["mass_data"]  # the operator substitution decided by a device check is named in the line
    assert "structured block 32x32x32" in d["config"]["mesh_order"] and "nothing permuted" in d["config"]["mesh_order"]
    for leg in ("c2multigeneral", "c3general"):    # the general-mesh twins of the N-rank and the 64^3 legs
        assert 0 < d["legs"][leg]["value"] < d["legs"][leg.replace("general", "")]["value"]
    # round-5 verdict, item 1: configs[1] under the numbering the reference's API hands over - the library orders zones and
    # nodes itself, so the legs stay within a few per cent of the headline and merge the same x-faces
    assert d["legs"]["c2mfem"]["value"] >= 0.9 * d["value"] and d["legs"]["c2perm"]["value"] >= 0.85 * d["value"]
    assert d["legs"]["c2mfem"]["merged_entries"] == d["legs"]["c2perm"]["merged_entries"] > 0
    # round-5 verdict, item 6: the CPU baseline is timed on the build SURVEY 8(d) names, the parity build beside it, flags stated
    b = d["cpu_baseline"]
    assert "-march=native" in b["flags"] and "-ffp-contract=fast" in b["flags"] and "-ffp-contract=off" in b["flags_parity_build"]
    assert b["value"] > 0 and b["value_parity_build"] > 0
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "r6_bench_detail.json")) as f:
        full = json.load(f)
    again = bench.compact_line(full, d["detail"])
    assert again == d
    # the full record prices K2 with the layout that ran: merged E-vector (fewer values than NE * ND), two-table ELL
    k2 = [v for k, v in full["kernels"].items() if k.startswith("vcg_update_p_k")][0]
    assert "summed by K1" in k2["moves"] and k2["bytes_per_launch"] == d["roofline"]["bytes_per_launch"]


def test_round5_line_still_reads_as_a_contract_line():
    """the committed line of the round before (other leg names, no numbering legs) still passes the same checks - compact_line and
    the contract did not drift"""
    with open(os.path.join(ROOT, "profiles", "r5_bench.json")) as f:
        text = [l for l in f.read().splitlines() if l.startswith("{")][0]
    _check_line(json.loads(text), text)


def test_profiled_run_agrees_with_the_plain_run():
    """The same command under rocprofv3 --kernel-trace --stats: same workload, throughput within
    the profiler's overhead."""
    a, b = _line("r6_bench.json"), _line("r6_bench_under_rocprofv3.json")
    assert a["config"]["workload"] == b["config"]["workload"]
    assert 0.8 * a["value"] < b["value"] <= 1.05 * a["value"]


def test_weak_scaling_layout_of_the_bench():
    """N ranks keep 32^3 elements each in a block grid whose ranks are all neighbours of each other
    up to N = 8 (the partitions for which the CG sums ride on the halo exchange)."""
    bench = _bench()
    assert [tuple(bench.block_grid(n)) for n in (1, 2, 4, 8)] == [(1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2)]
    for n in (1, 2, 3, 4, 6, 8):
        px, py, pz = bench.block_grid(n)
        assert px * py * pz == n
        # the grid is the one laghos::Partition builds for the mesh bench.py hands it (tests/test_host_setup.py
        # holds the Python mirror against the C++ code)
        assert bench.partition_grid((32 * px, 32 * py, 32 * pz), n) == (px, py, pz)
    assert bench.usable_cpus() >= 1


def test_line_survives_an_oversized_record():
    """Whatever a future leg or rank count adds to the record, the stdout line stays parseable: compact_line keeps the
    per-leg entries to five figures, and main() drops `legs` and `comm` before it would print more than 8 KB (the emergency
    exit that round 4 did not have).  Here: a record with 200 legs."""
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "r6_bench_detail.json")) as f:
        full = json.load(f)
    one = full["legs"]["c3"]
    full["legs"] = {"leg%03d" % i: one for i in range(200)}
    line = bench.compact_line(full, "x")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) > bench.LINE_LIMIT            # (this record would not fit ...)
    for key in ("legs", "comm"):                   # ... and this is what main() does about it
        line.pop(key, None)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.LINE_LIMIT
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "cpu_baseline", "parity", "config"):
        assert key in d

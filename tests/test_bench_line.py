"""bench.py's ONE stdout line must fit the driver's window (a 26 KB line left BENCH_r04.json.parsed = null).

The shape comes from a real run (profiles/bench_r04_default.json: the headline + four `configs` with per-class `rooflines` + five
`progressive` blocks); `compact_line` has to carry the contract's fields, `roofline`, `cpu_baseline` and one entry per other config in
at most 3 500 characters, and the last 4 000 characters of stdout have to parse as JSON on their own."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _round4_record():
    with open(os.path.join(ROOT, "profiles", "bench_r04_default.json")) as f:
        return json.load(f)


def test_compact_line_fits_and_keeps_the_contract():
    bench = _bench()
    d = _round4_record()
    assert len(json.dumps(d)) > 20000                      # the record that did not parse
    line = bench.compact_line(d)
    text = json.dumps(line)
    assert len(text) < 3500
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"]
    assert "workload" in line["config"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    names = [c["config"] for c in line["configs"]]
    assert names == ["cornell_two_spheres", "cloud", "sky", "manylight"]
    for c in line["configs"]:
        assert c["seconds_per_frame"] > 0 and c["value"] > 0
        assert {"kernel", "bound", "frac"} <= set(c["roofline"])
    assert "ms_per_call" in line["progressive"] and "ms_per_call_with_readback" in line["progressive"]


def test_compact_line_sheds_before_it_overflows():
    bench = _bench()
    d = _round4_record()
    d["config"]["workload"] = "x" * 5000
    d["configs"] = d["configs"] * 6                        # 24 entries: more than the window could ever hold at full width
    for c in d["configs"]:
        c["workload"] = "y" * 900
    text = json.dumps(bench.compact_line(d))
    assert len(text) <= 3500
    line = json.loads(text)
    assert line["roofline"]["frac"] == d["roofline"]["frac"] and line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"]


def test_emit_prints_one_line_the_driver_can_parse(tmp_path):
    bench = _bench()
    d = _round4_record()
    out, err = io.StringIO(), io.StringIO()
    detail = tmp_path / "bench_detail.json"
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        bench.emit(d, str(detail))
    stdout = out.getvalue()
    assert stdout.count("\n") == 1 and stdout.endswith("\n")
    parsed = json.loads(stdout[-4000:])                    # what survives in the driver's tail
    assert parsed["metric"] == "Mrays/s" and parsed["detail"] == "bench_detail.json"
    full = json.loads(detail.read_text())
    assert full["rooflines"] == d["rooflines"] and len(full["configs"]) == 4


def test_compact_line_carries_the_frame_check():
    """round 6: bench.py looks at the film it timed (finite, mean ratio and same-sample parity against the oracle image of its CPU leg); the
    verdict travels in the driver-visible line, and a failing configuration is marked there too."""
    bench = _bench()
    d = _round4_record()
    d["frame_check"] = {"finite": True, "mean_weight_per_pixel": 378.9, "mean_rgb_sum": 398.03, "ok": True, "oracle_spp": 14, "rel_mse_vs_oracle_14spp": 0.066,
                        "mean_ratio": 0.99976, "same_samples_rel_mse": 1.24e-09, "same_samples_frac_pixels_within_1e-2": 1.0, "tolerance": "x" * 300}
    d["configs"][1]["frame_check"] = {"finite": False, "mean_rgb_sum": 0.0, "ok": False}
    line = bench.compact_line(d)
    assert len(json.dumps(line)) <= 3500
    assert line["frame_check"]["ok"] is True and line["frame_check"]["mean_ratio"] == 0.99976 and "tolerance" not in line["frame_check"]
    assert line["configs"][1]["frame_ok"] is False and "frame_ok" not in line["configs"][0]


def test_frame_check_verdict_rules():
    """The rule that ends a bench run with exit code 4: surfaces are held to SURVEY 8(d)'s frame tolerance on the same sample indices, media
    scenes (paths decorrelate at the first ulp) to the means and to most pixels agreeing sample for sample.  The measured records of round 6
    pass, planted failures do not."""
    v = _bench().frame_check_verdict
    cornell = {"finite": True, "mean_ratio": 0.99979, "same_samples_mean_ratio": 1.0, "same_samples_rel_mse": 1.15e-09, "same_samples_frac_pixels_within_1e-2": 1.0}
    cloud = {"finite": True, "mean_ratio": 0.99989, "same_samples_mean_ratio": 1.00001, "same_samples_rel_mse": 0.203, "same_samples_frac_pixels_within_1e-2": 0.93634}
    assert v(cornell, False) and v(cloud, True)
    assert not v(cloud, False)                                               # a surface scene with the cloud's scatter would be a bug
    assert not v(dict(cornell, finite=False), False)
    assert not v(dict(cornell, mean_ratio=1.03), False) and not v(dict(cloud, mean_ratio=0.97), True)
    assert not v(dict(cornell, same_samples_rel_mse=2e-3), False)
    assert not v(dict(cornell, **{"same_samples_frac_pixels_within_1e-2": 0.98}), False)
    assert not v(dict(cloud, **{"same_samples_frac_pixels_within_1e-2": 0.627}), True)     # what the build without correctly rounded division measured
    assert not v(dict(cloud, same_samples_mean_ratio=1.05), True)
    assert not v({}, False) and not v({}, True)


def test_utilisation_digests_carry_the_measured_valu_occupancy():
    """profiles/utilisation_<config>.json (tools/pmc_utilisation.py): every kernel family of every bench config carries rocprof's VALUBusy
    (`valu_busy`), never above 1, never below the 2-cycles-per-instruction lower bound (`valu_issue_frac`), and the resident waves per SIMD
    that pin the counters' unit (<= 8)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cfg in ("cornell", "sky", "cloud", "manylight", "cornell_sphere_box"):
        d = json.load(open(os.path.join(root, "profiles", "utilisation_%s.json" % cfg)))
        fams = {k: e for k, e in d.items() if isinstance(e, dict)}
        assert fams, cfg
        for k, e in fams.items():
            assert 0.0 < e["valu_issue_frac"] <= e["valu_busy"] <= 1.0, (cfg, k, e)
            assert 0.0 < e["lane_util"] <= 1.0 and 0.5 < e["waves_per_simd"] <= 8.0, (cfg, k, e)

"""CPU tests of bench.py's bookkeeping: the roofline object (what a launch's duration is when
launches overlap, which resource `binding` names), the algorithmic byte and flop counts of SURVEY
section 8(d), and the shapes the bench derives from a replica count."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def fake(streams, launches=40, elapsed=0.0102, span_us=350.0, timed=8):
    st = {"launches": launches, "moves": 32768 * launches, "timed_launches": timed,
          "kernel_ms": span_us * 1e-3 * timed}
    return {"st": st, "elapsed": elapsed, "streams": streams}


ARGS = argparse.Namespace(kernel=3, no_events=False)


def test_survey_counts():
    # 78.7 KB per trial move and 111.1 KB / 3.5e7 flop per full evaluation at 750 molecules
    assert bench.algorithmic_bytes_per_move(750, 30.0) == pytest.approx(78657, rel=1e-4)
    assert bench.algorithmic_bytes_full_eval(750) == 36 * 2250 + 24 * 750 + 36 * 337
    assert bench.algorithmic_flops_full_eval(750, 30.0) == pytest.approx(3.5e7, rel=0.02)


def test_launch_duration_is_cost_when_launches_overlap():
    two = bench.roofline_object(fake(2), 65536, ARGS, {"groups": 2}, 750, 30.0, 1)
    one = bench.roofline_object(fake(1, span_us=262.0), 65536, ARGS, {"groups": 2}, 750, 30.0, 1)
    # two streams: cost = timed region / launches, the event span is reported beside it
    assert two["avg_launch_us"] == pytest.approx(1e6 * 0.0102 / 40)
    assert two["launch_span_us"] == pytest.approx(350.0) and "wall time / launches" in two["avg_launch_us_is"]
    # one stream: the event duration IS the cost
    assert one["avg_launch_us"] == pytest.approx(262.0) and "HIP events" in one["avg_launch_us_is"]
    for r in (two, one):
        assert r["kernel"] == "k_move_eval_wave" and r["bound"] == "hbm" and r["peak"] == 8000.0
        assert r["frac"] == pytest.approx(r["achieved"] / 8000.0)
        assert r["achieved"] == pytest.approx(bench.algorithmic_bytes_per_move(750, 30.0) * 32768
                                              / (r["avg_launch_us"] * 1e-6) / 1e9)


def test_steps_per_launch_is_read_from_the_counts():
    st = {"launches": 10, "moves": 30720 * 8 * 10, "timed_launches": 0, "kernel_ms": 0.0}
    r = bench.roofline_object({"st": st, "elapsed": 0.017, "streams": 2}, 61440, ARGS, {"groups": 2}, 750, 30.0, 1)
    assert r["steps_per_launch"] == 8 and r["moves_per_launch"] == 30720 * 8


def test_binding_block_names_the_larger_fraction_and_stays_below_one():
    import glob
    import json
    shape = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    mpl = int(json.load(open(shape[-1]))["moves_per_launch"]) if shape else 32768   # the committed launch shape
    R = 61440 if mpl % 30720 == 0 else 65536
    st = {"launches": 40, "moves": mpl * 40, "timed_launches": 8, "kernel_ms": 0.35 * 8 * mpl / 32768}
    r = bench.roofline_object({"st": st, "elapsed": 40 * 250e-6 * mpl / 32768, "streams": 2}, R, ARGS,
                              {"groups": 2}, 750, 30.0, 1)
    b = r.get("binding")
    if b is None:
        pytest.skip("no committed PMC summary for this launch shape under profiles/")
    v, h = b["fp64_valu_issue"], b["hbm"]
    assert v["peak"] == pytest.approx(1024 * v["clock_ghz"] / 4.0)
    assert v["achieved"] == pytest.approx(v["valu_insts_per_move"] * mpl / (r["avg_launch_us"] * 1e-6) / 1e9)
    assert 0 < v["frac"] <= 1.0 and 0 < h["frac"] <= 1.0
    cands = {"fp64_valu_issue": v["frac"]}
    if "access_pattern" in b:   # the kernel's loads and stores without arithmetic: their time / the launch's
        assert 0 < b["access_pattern"]["frac"] <= 1.0
        assert b["access_pattern"]["frac"] == pytest.approx(b["access_pattern"]["floor_us"] / r["avg_launch_us"])
        cands["hbm_access_pattern"] = b["access_pattern"]["frac"]
    assert b["bound"] == max(cands, key=cands.get) and b["frac"] == pytest.approx(max(cands.values()))
    assert "not measured in this run" in b["counters_source"]


def test_shapes():
    a = argparse.Namespace(groups=0, threads=8, steps=None, warmup=None, zero_copy_moves=-1)
    big, small = bench.shape_for(65536, a), bench.shape_for(32, a)
    assert big["groups"] == 2 and big["steps"] == 600 and big["warmup"] == 64 and big["prewarm"] == 56
    assert small["threads"] == 4 and small["steps"] == 3000 and bench.shape_for(1, a)["groups"] == 1
    assert bench.default_parts(65536, 750) == 1 and bench.default_parts(32, 750) == 5
    assert bench.server_lat_parts(1, 750) == 16 and bench.server_lat_parts(1, 10000) == 84
    assert bench.server_lat_parts(512, 750) == 0
    assert bench.kernel_name(3, 32768, 1) == "k_move_eval_wave" and bench.kernel_name(3, 32, 5) == "k_move_eval_fast"


def test_a_call_whose_last_launch_is_shorter_still_gets_its_binding_block():
    """The driver's `--steps 20`: launches of 8 + 8 + 4 steps.  The committed counters belong to the
    8-step launch; per move they hold for the call's average launch."""
    import glob
    import json
    shape = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not shape or int(json.load(open(shape[-1]))["moves_per_launch"]) != 30720 * 8:
        pytest.skip("the committed counters are not those of 8-step launches of 30720 replicas")
    st = {"launches": 6, "moves": 61440 * 20, "timed_launches": 2, "kernel_ms": 4.4}
    r = bench.roofline_object({"st": st, "elapsed": 0.0090, "streams": 2}, 61440, ARGS, {"groups": 2}, 750, 30.0, 1)
    assert r["steps_per_launch"] == 8 and r["moves_per_launch"] == pytest.approx(61440 * 20 / 6)
    b = r["binding"]
    assert 0 < b["frac"] <= 1.0 and b["bound"] in ("fp64_valu_issue", "hbm_access_pattern")
    assert b["hbm"]["bytes_per_move"] == pytest.approx(json.load(open(shape[-1]))["bytes_per_move"])
    assert b["access_pattern"]["floor_us"] == pytest.approx(b["access_pattern"]["frac"] * r["avg_launch_us"])

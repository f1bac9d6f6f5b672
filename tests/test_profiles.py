"""profiles/ must be evidence of the batches the bench line reports (rocprof passes of sibling batches).  CPU side, on
the committed JSON of the newest round-4+ tag: every `<tag>_<config>_pmc.json` (tools/profile_round.py: `bench.py --config <c>` under
rocprofv3) ran the batch with the seed bench.py's CONFIGS names, and its iteration counts -- which determine the launch duration --
equal the same config's entry in `<tag>_bench.json` (the plain `python bench.py` line of the same box)."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_tag():
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench.json")):
        m = re.match(r"r(\d+)_v(\d+)_bench\.json", os.path.basename(f))
        if m and int(m.group(1)) >= 4:
            key = (int(m.group(1)), int(m.group(2)))
            best = max(best, key) if best else key
    return "r%02d_v%d" % best if best else None


def test_profiles_are_of_the_batches_the_bench_line_reports():
    tag = _newest_tag()
    if tag is None:
        pytest.skip("no round-4+ profile set committed yet")
    sys.path.insert(0, ROOT)
    import bench

    line = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
    entries = {c["config"]: c for c in line["configs"]}
    entries["c1"] = dict(line["solver"], kernel_ms=line["roofline"]["kernel_ms"], active_set_kernel_ms=line["roofline"]["kernel_ms"])
    seen = 0
    for key, cfg in bench.CONFIGS.items():
        f = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (tag, key))
        if not os.path.exists(f):
            continue
        pm = json.load(open(f))
        under = pm["bench_under_rocprof"]
        assert under["config"]["baseline_config"] == key and under["config"]["batch_seed"] == cfg["seed"], (key, under["config"])
        assert pm["qps_per_launch"] == cfg["agents"]
        e = entries[key]
        assert abs(under["solver"]["iters_mean"] - e["iters_mean"]) <= 1e-12 and under["solver"]["iters_max"] == e["iters_max"], (
            key, under["solver"], e["iters_mean"], e["iters_max"])
        # the rocprofv3 average of the timed launches and the HIP-event time of the bench line: same batch, same kernel -- recorded side
        # by side (a ratio, not an assertion: the two runs are different processes on a shared box)
        tl = pm["timed_launches"][pm["kernel"]] if pm.get("kernel") in pm.get("timed_launches", {}) else None
        if tl:
            ev = e.get("active_set_kernel_ms") if "das_kernel" in pm["kernel"] else None  # (the dominant kernel's own HIP-event time, where the line carries it)
            print("%s %s: rocprofv3 %.1f us (%s), bench line %.1f us" % (tag, key, tl["avg_ns"] / 1e3, pm["kernel"].split("(")[0][-40:], (ev or e["kernel_ms"]) * 1e3))
        seen += 1
    assert seen >= 5, "profile set %s is incomplete (%d configs)" % (tag, seen)


def test_round6_profiles_use_one_calibrated_counter_rule():
    """FETCH_SIZE / WRITE_SIZE calibrated in the SAME rocprofv3 session on known-byte streaming kernels, ONE rule for every
    config, counters attributed per kernel instantiation, and the cold 4096 x M5 launch moving what the byte model says it moves."""
    tag = _newest_tag()
    if tag is None or int(tag[1:3]) < 6:
        pytest.skip("no round-6 profile set committed yet")
    hows, seen = set(), 0
    for f in glob.glob(os.path.join(ROOT, "profiles", tag + "_*_pmc.json")):
        pm = json.load(open(f))
        cal = pm["fetch_calibration"]
        hows.add(cal["how"])
        # the stack's own behaviour, measured not assumed: half of the bytes of a coalesced read at every width, all of a write
        for lb, r in cal["all_read_ratios"].items():
            assert 0.45 <= r <= 0.55, (f, lb, r)
        assert 0.95 <= cal["write_counted_per_known_byte"] <= 1.05
        assert set(pm["traffic_per_kernel"]) == set(pm["kernels"]) and pm["traffic_bytes_per_launch"] == pm["traffic_per_kernel"][pm["kernel"]]["traffic"]
        assert "--cold" in pm["command_cold"] and "--calibrate-counters" in pm["command_cold"]
        seen += 1
    assert seen >= 5 and len(hows) == 1, hows
    c4 = json.load(open(os.path.join(ROOT, "profiles", tag + "_c4_f64_pmc.json")))
    assert 0.95 <= c4["traffic_over_algorithmic"] <= 1.15, c4["traffic_over_algorithmic"]
    line = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
    assert line["roofline"]["frac_cold"] > 0 and line["spread"]["repeats"] >= 5

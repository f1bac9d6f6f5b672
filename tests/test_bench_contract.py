"""bench.py contract: one JSON line with the fields the driver reads (GPU only: there is no CPU path to bench)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--no-extra"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 20 and b["warmup"] == 3 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["dtype"] == "f64" and b["data"] == "synthetic" and b["vs_baseline"] is None and "workload" in b["config"]
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "QP/s" and c["sample"]
    assert b["solver"]["non_optimal"] == 0
    assert b["parity"]["max_rel_dobj"] <= 1e-8 and b["parity"]["max_abs_dx"] <= 1e-6
    # BASELINE's metric has two halves: the p99 latency and the roof that binds travel inside objects the driver keeps
    assert b["config"]["latency_ms"]["p99"] >= b["config"]["latency_ms"]["p50"] > 0 and b["config"]["latency_ms"]["calls"] >= 1000
    # round 5: which kernel finished what (the dual active-set phase in front, the interior-point kernel behind it); the fp64 vector roofline is
    # the interior-point kernel's and is only there when that kernel solved something
    pth = b["solver"]["paths"]
    assert pth["active_set_solved"] + pth["interior_point_solved"] == 64 and pth["active_set_solved"] > 0
    assert ("das_kernel" in r["kernel"]) == (pth["active_set_solved"] >= 32) and r["kernel_ms"] > 0 and r["step_gpu_ms"] >= 0.5 * r["kernel_ms"]
    assert ("fp64_valu" in r) == (pth["interior_point_solved"] > 0) and (pth["interior_point_solved"] == 0 or 0 < r["fp64_valu"]["frac"] < 1)
    assert b["config"]["baseline_config"] == "c1" and b["config"]["batch_seed"] == 1000
    # value is whole-job throughput of the timed steps
    assert abs(b["value"] - 64 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]


@pytest.mark.gpu
def test_bench_line_covers_every_baseline_config_at_its_own_shape():
    """Next to the configs[1] headline, every other BASELINE config is measured at its own shape on this GPU -- QP/s, HBM fraction,
    iterations, oracle parity and CPU baseline.  (--no-latency: the 1000-call percentile loops are the driver's bench run, not a test;
    nothing here asserts a duration -- timings are recorded in the line, never judged by the correctness suite.)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-latency"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    b = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    by = {c["config"]: c for c in b["configs"]}
    base = {"c0", "c2", "c3s", "c3", "c4_f64", "c4"}
    # round 6: the same swarms later in their exchange, and SURVEY 8d's controlled fraction of infeasible instances
    loaded = {"c1_loaded", "c0_loaded", "c2_loaded", "c4_loaded"}
    infeasible = {"c1_infeasible_1pct", "c4_infeasible_1pct"}
    assert set(by) == base | loaded | infeasible
    shapes = {"c0": (10, 10, 2), "c2": (512, 6, 3), "c3s": (128, 10, 3), "c3": (1024, 10, 3), "c4_f64": (4096, 5, 3), "c4": (4096, 5, 3),
              "c1_loaded": (64, 5, 3), "c0_loaded": (10, 10, 2), "c2_loaded": (512, 6, 3), "c4_loaded": (4096, 5, 3),
              "c1_infeasible_1pct": (64, 5, 3), "c4_infeasible_1pct": (4096, 5, 3)}
    for k in infeasible:  # every instance made infeasible is reported so -- with the phase on and off -- and its neighbours in the batch are solved
        c = by[k]
        assert "error" not in c, c
        assert c["infeasible"]["all_reported_non_optimal"] and c["infeasible"]["others_optimal"] and c["non_optimal"] == c["infeasible"]["count"]
        assert c["phase_off"]["non_optimal"] == c["infeasible"]["count"] and c["phase_off"]["statuses_equal"]
    for k in loaded:  # busier than the 3-replan batches, still finished, still the oracle's optimum; the phase on and off agree
        c = by[k]
        assert "error" not in c, c
        quiet = {"c0_loaded": "c0", "c2_loaded": "c2", "c4_loaded": "c4_f64"}.get(k)
        assert c["paths"]["active_set_steps_mean"] > (by[quiet]["paths"]["active_set_steps_mean"] if quiet else 0.3)
        assert c["phase_off"]["statuses_equal"] and c["phase_off"]["max_abs_dx_vs_phase_on"] <= 2e-6
    for k, c in by.items():
        assert "error" not in c, c
        assert (c["agents"], c["segments"], c["dim"]) == shapes[k]
        assert c["ms_spread"]["min"] <= c["ms_spread"]["median"] <= c["ms_spread"]["max"]
        if k in infeasible:
            continue
        assert c["non_optimal"] == 0 and c["latency_ms"]["calls"] >= 1 and c["qp_per_s"] > 0 and c["hbm_frac"] > 0
        assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] == "port"
        # both launch orders are in the line; the sorted one only where a launch runs more than one round of workgroups
        # (round 5: with the dual active-set phase in front a launch has no iteration tail to sort: every config runs as given)
        assert c["kernel_ms_as_given"] > 0 and "longest first" not in c["work_order"] and c["paths"]["active_set_solved"] > 0
        if c["rows"] == "f64":
            assert c["parity_vs_oracle"]["max_abs_dx"] <= 1e-6 and c["parity_vs_oracle"]["max_rel_dobj"] <= 1e-8
            assert c["parity_vs_oracle"]["status_disagreements"] == 0
    assert by["c4"]["precision"] == "mixed" and by["c4"]["rows"] == "f32" and by["c3"]["lsc_neighbours"] == 40
    assert b["config"]["baseline_config"] == "c1" and "mixed_vs_fp64_at_4096" in b
    # the line carries its own spread, the cold-HBM figure beside the hot one, and the kernel behind the phase measured on its own
    assert b["spread"]["repeats"] >= 5 and b["spread"]["ms_per_step"]["min"] <= b["ms_per_step"] <= b["spread"]["ms_per_step"]["max"]
    assert b["roofline"]["frac_cold"] > 0 and b["roofline"]["cold"]["copies_bit_identical"] and b["roofline"]["cold"]["bytes_all_copies"] >= 512 * 2**20
    assert b["phase_off"]["non_optimal"] == 0 and b["phase_off"]["fp64_valu"]["frac_of_78.6e12"] > 0
    for k in ("c2", "c3", "c4_f64"):
        assert by[k]["cold"]["frac"] > 0 and by[k]["cold"]["copies_bit_identical"], by[k]["cold"]
    # the whole replan of the forest10 mission as one device chain, eager and as a hipGraph (informational; durations not asserted)
    rc = b["replan_chain"]
    assert "error" not in rc, rc
    assert rc["graph_nodes"] >= 8 and rc["eager"]["failed_qps_last_replan"] == 0 and rc["graph"]["failed_qps_last_replan"] == 0
    c1c = rc["c1_class"]  # the chain at the headline's class: 64 agents x M5 in 3-D
    assert c1c["failed_qps"] == 0 and c1c["mean_distance_flown_m"] > 1.0, c1c


def _run_bench(argv, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    return out


def test_gpus_flag_and_world_size_must_agree():
    """`--gpus N` is not a label: a launch whose WORLD_SIZE differs is refused before anything runs (no device needed)."""
    out = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0"}, timeout=120)
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in out.stderr
    out = _run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0"}, timeout=120)
    assert out.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in out.stderr


@pytest.mark.gpu
def test_bare_gpus_2_measures_the_sharded_config_with_the_allgather():
    """`python bench.py --gpus 2` started WITHOUT a launcher and without any other flag than the driver's runs two ranks (here: sharing the
    one device, collectives over gloo -- LSCQP_BENCH_BACKEND=gloo exists for this test) on what BASELINE's multi-GPU configs name: configs[3]
    (1024 agents x M10 x 40) as ONE batch in contiguous blocks, the plans all-gathered every step; followed by the configs[4] shape, configs[2]
    and the weak configs[1] block.  Every rank's block is checked against the oracle."""
    out = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "2"], {"LSCQP_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    b = json.loads(lines[0])
    c = b["config"]
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and c["baseline_config"] == "c3" and c["allgather"] is True
    assert c["ranks"] == 2 and c["agents_total"] == 1024 and c["agents_per_gpu"] == 512 and c["agents_solved_per_step_all_ranks"] == 1024
    assert c["segments"] == 10 and c["lsc_neighbours"] == 40 and c["allgather_bytes_per_rank_per_step"] == 512 * 180 * 8
    assert "gloo" in c["collective_backend"] and "2 ranks" in c["collective_backend"] and c["distinct_devices"] >= 1
    assert abs(b["value"] - 1024 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]
    assert c["latency_ms"]["p99"] >= c["latency_ms"]["p50"] > 0 and c["latency_ms"]["calls"] >= 200
    one = c["one_gpu_same_workload"]
    assert one["qp_per_s"] > 0 and one["rank0_block_bit_identical_to_sharded_solve"] is True
    pr = b["parity_all_ranks"]
    assert pr["ranks"] == 2 and pr["compared"] == 12 and pr["max_abs_dx"] <= 1e-6 and pr["max_rel_dobj"] <= 1e-8
    assert b["solver"]["non_optimal"] == 0 and b["roofline"]["qps_per_launch"] == 512 and 0 < b["roofline"]["whole_job"]["frac"] < 1
    more = {(w["baseline_config"], w["scaling"]): w for w in b["other_workloads"]}
    assert set(more) == {("c4_f64", "strong"), ("c2", "strong"), ("c1", "weak")}
    for w in more.values():
        assert "error" not in w, w
        assert w["non_optimal"] == 0 and w["parity_all_ranks"]["max_abs_dx"] <= 1e-6 and w["parity_all_ranks"]["max_rel_dobj"] <= 1e-8
        assert abs(w["value"] - w["agents_total"] * 1e3 / w["ms_per_step"]) <= 1e-6 * w["value"]
    assert more[("c4_f64", "strong")]["allgather"] is True and more[("c4_f64", "strong")]["agents_total"] == 4096 and more[("c4_f64", "strong")]["agents_per_gpu"] == 2048
    assert more[("c2", "strong")]["agents_total"] == 512 and more[("c1", "weak")]["allgather"] is False and more[("c1", "weak")]["agents_total"] == 128


@pytest.mark.gpu
def test_weak_scaling_of_the_headline_config_stays_available():
    """`--config c1 --scaling weak`: every rank owns its own 64-agent swarm, no data-path collective (the N = 1 headline replicated)."""
    out = _run_bench(["--gpus", "2", "--config", "c1", "--scaling", "weak", "--steps", "10", "--warmup", "3", "--no-extra", "--no-latency"], {"LSCQP_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    b = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    c = b["config"]
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and c["ranks"] == 2 and c["agents_total"] == 128 and c["agents_solved_per_step_all_ranks"] == 128
    assert c["allgather"] is False and abs(b["value"] - 128 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]
    pr = b["parity_all_ranks"]
    assert pr["ranks"] == 2 and pr["compared"] == 32 and pr["max_abs_dx"] <= 1e-6 and pr["max_rel_dobj"] <= 1e-8
    assert b["solver"]["non_optimal"] == 0
    # without the test backend two ranks need two devices: on a one-GPU box the launch is refused, loudly
    import torch

    if torch.cuda.device_count() == 1:
        out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra", "--no-latency"], timeout=300)
        assert out.returncode != 0 and "distinct devices" in out.stderr


@pytest.mark.gpu
def test_strong_scaling_splits_one_batch_and_gathers_the_plans():
    """BASELINE configs[2]'s mode: 512 agents as ONE batch over 2 ranks (256 each), the plans all-gathered every step; each rank's
    block against the oracle; the gathered buffer holds the rank's own block."""
    out = _run_bench(["--gpus", "2", "--scaling", "strong", "--config", "c2", "--steps", "10", "--warmup", "3", "--no-extra"],
                     {"LSCQP_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    b = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    c = b["config"]
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and c["agents_total"] == 512 and c["agents_per_gpu"] == 256 and c["allgather"] is True
    assert c["agents_solved_per_step_all_ranks"] == 512 and c["devices_by_crossover_rule"] == 2 and c["baseline_config"] == "c2"
    assert c["allgather_bytes_per_rank_per_step"] == 256 * 108 * 8
    assert abs(b["value"] - 512 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]
    assert b["roofline"]["qps_per_launch"] == 256
    pr = b["parity_all_ranks"]
    assert pr["compared"] == 32 and pr["max_abs_dx"] <= 1e-6 and pr["max_rel_dobj"] <= 1e-8
    assert b["latency_ms"]["sharded_step"]["calls"] >= 200 and b["solver"]["non_optimal"] == 0
    # a ragged split: 10 agents over 3 ranks (4 + 4 + 2), the last block padded for the collective, never solved
    out = _run_bench(["--gpus", "3", "--scaling", "strong", "--config", "c0", "--steps", "5", "--warmup", "2", "--no-extra", "--no-latency"],
                     {"LSCQP_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-3000:]
    b = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert b["n_gpus"] == 3 and b["config"]["agents_per_gpu"] == 4 and b["config"]["agents_solved_per_step_all_ranks"] == 10
    assert b["parity_all_ranks"]["compared"] == 10 and b["parity_all_ranks"]["max_abs_dx"] <= 1e-6


@pytest.mark.gpu
def test_single_process_communicator_bench():
    """--single-process: the same split driven by ONE process through lscqp_comm (ncclCommInitAll + lscqp_solve_batch_sharded_device +
    lscqp_allgather); one device here, the code path is the one G devices take."""
    out = _run_bench(["--single-process", "--gpus", "1", "--config", "c2", "--steps", "10", "--warmup", "3", "--no-latency"])
    assert out.returncode == 0, out.stderr[-3000:]
    b = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert b["n_gpus"] == 1 and b["scaling"] == "strong" and b["config"]["agents_total"] == 512 and "rccl" in b["config"]["collective_backend"]
    assert b["solver"]["non_optimal"] == 0 and b["parity_all_ranks"]["max_abs_dx"] <= 1e-6 and b["parity_all_ranks"]["max_rel_dobj"] <= 1e-8
    import torch

    if torch.cuda.device_count() == 1:
        out = _run_bench(["--single-process", "--gpus", "2", "--config", "c2", "--steps", "2", "--warmup", "1"], timeout=300)
        assert out.returncode != 0 and "needs 2 devices" in out.stderr

#!/usr/bin/env python
"""bench.py — QP solves/sec of the batched trajectory-QP hot path on N MI355X GPUs of one node.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU -- by the driver, or BY ITSELF: `python bench.py --gpus N` started without
WORLD_SIZE in the environment re-launches itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1`, and every rank refuses to run unless WORLD_SIZE == --gpus and (over RCCL) N distinct devices
exist.  Rank 0 prints ONE JSON line.

  * A "step" is one pass of the hot path (lscqp_solve_batch_device: the HIP PDIP kernel) over one batch of synthetic
    agents, inputs already resident in HBM.  The workload at every N is BASELINE.json configs[1] PER GPU:
    64 agents, M = 5 segments, 20 LSC neighbours per agent (~20 half-spaces per segment and control point), dim 3,
    fp64 — produced by a synthetic swarm after 3 warm-up replans.  Weak scaling: every rank owns its
    own 64-agent swarm, no data-path collective (the QPs of one replan step are independent,
    reference src/multi_sync_simulator.cpp:354-362).  `--allgather` adds the RCCL all-gather of the solved control
    points after every step (the device analogue of broadcastMsgs, :305-352) for those who want it in the number.
  * `--scaling strong --config c2|c3|c4` is the mode BASELINE's 8-GPU configs describe: ONE global batch (512 / 1024 / 4096
    agents) cut into contiguous blocks of ceil(N/G) agents (sharding.shard_range == lscqp_shard_range), every rank solves its
    block and the plans are all-gathered over RCCL EVERY step (broadcastMsgs, :305-352); value = global agents x steps / time.
    `--single-process` runs the same split from ONE host process through lscqp_comm (ncclCommInitAll, one stream per device:
    lscqp_solve_batch_sharded_device + lscqp_allgather), the deployment the reference's single ROS process would use.
  * roofline: algorithmic bytes (20 432 B per QP at this shape) x QPs per launch / the kernel's average
    launch duration, measured with HIP events on the launch stream over the timed region, against 8.0 TB/s.
  * cpu_baseline: the CPU oracle (oracle/, a port: CPLEX cannot exist here) on the same batch, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VECTOR_PEAK = 78.6e12  # flop/s, MI355X fp64 vector (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)


def valu_roofline(sol, n, n_obs, iterations, kernel_ms):
    """The roof that binds: fp64 vector work of ONE launch = sum over its instances of
    flops_fixed + iterations x flops_per_iteration + flops_last_pass, where the iterations are the kernel's own count of THIS run
    (lscqp_info.iterations) and the per-iteration figure is read off the selected kernel instance's machine code at build time
    (lscqp_instance_work), over the launch duration measured in this run; against MI355X's 78.6 TFLOP/s fp64 vector peak."""
    try:
        w = sol.instance_work(n, n_obs)
    except Exception:  # the run-time-shaped kernel carries no instruction counts
        return None
    it = np.asarray(iterations, dtype=np.float64)
    if it.size == 0:
        return None
    flops = float((w["flops_fixed"] + w["flops_last_pass"] + it * w["flops_per_iteration"]).sum())
    rate = flops / (kernel_ms * 1e-3)
    # the launch lasts as long as its slowest instance: that wavefront's own issue rate (VALU instructions per second of the launch)
    crit = w["valu_insts_fixed"] + float(it.max()) * w["valu_insts_per_iteration"]
    return {"bound": "fp64 vector ALU", "fp64_flops_per_launch": flops, "fp64_flops_per_s": rate, "peak_flops_per_s": FP64_VECTOR_PEAK,
            "frac_of_78.6e12": rate / FP64_VECTOR_PEAK, "kernel": w["kernel"], "wavefronts_per_qp": w["wavefronts"],
            "flops_per_iteration": w["flops_per_iteration"], "valu_insts_per_iteration_per_wavefront": w["valu_insts_per_iteration"],
            "iterations_sum": float(it.sum()), "iterations_max": int(it.max()),
            "slowest_wavefront_valu_insts_per_us": crit / (kernel_ms * 1e3),
            "source": "iterations: lscqp_info of this run; per-iteration counts: lscqp_instance_work (machine code of the instance, build time)"}


def make_batch(api, synth, solver_factory, N, M, dim, n_obs, seed, style, warm_steps):
    """Swarm after `warm_steps` replans (carried forward by the HIP solver itself); returns ABI arrays."""
    sw = synth.Swarm(N, M=M, dim=dim, n_obs=n_obs, seed=seed, style=style)
    sol = solver_factory(sw)
    for _ in range(warm_steps):
        b = sw.build()
        hdr, rows, off, sfc = api.batch_from_swarm(b, sw.n_obs, M)
        x0 = api.x_init_from_swarm(b, dim)
        r = sol.solve_host(hdr, rows, off, sfc, want_info=False, x_init=x0)
        bad = r["status"] != 0
        if bad.mean() > 0.02:
            raise RuntimeError("warm-up replan produced non-optimal instances: %s" % np.bincount(r["status"]))
        # what the planner does with a failed QP: it keeps the initial trajectory (reference src/traj_planner.cpp:767-797)
        r["x"][bad] = x0[bad]
        sw.advance(r["x"])
    b = sw.build()
    return sw, sol, b, api.batch_from_swarm(b, sw.n_obs, M)


def slice_build(build, n, lo, hi):
    """The agents [lo, hi) of a synth.Swarm.build() dict (per-agent arrays are the ones whose first dimension is n)."""
    return {k: (v[lo:hi] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == n else v) for k, v in build.items()}


def to_dev(torch, a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


COLD_BYTES = 512 * 1024 * 1024  # >= 2 x the 256 MiB Infinity Cache (/opt/skills/guides/MI355X_MICROARCH.md): see cold_measure


def time_calls(torch, calls, warm=3):
    """Average duration of one call of `calls` (zero-argument callables that enqueue on torch's current stream), back to back between two
    HIP events; `warm` calls of the list's head first, untimed."""
    for c in calls[:warm]:
        c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for c in calls:
        c()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / len(calls)


def cold_copies(torch, tensors, n, nv, max_copies=1024):
    """K copies of a device-resident batch -- inputs (header, rows, offsets, boxes, initial trajectories) AND outputs -- with K x bytes >= 512 MiB
    (twice the Infinity Cache), every buffer kind in ONE allocation, every copy 256-byte aligned.  Returns (K, bytes per copy, lists of K views)."""
    dh, dr, do, ds, dxi = tensors
    per = sum(int(t.numel() * t.element_size()) for t in (dh, dr, do, ds) if t is not None) + (int(dxi.numel() * dxi.element_size()) if dxi is not None else 0)
    per_out = n * nv * 8 + n * 8 + n * 4 + n * 32
    K = int(min(max_copies, max(2, -(-COLD_BYTES // max(per + per_out, 1)))))

    def rep(t):
        if t is None:
            return [None] * K
        b = t.contiguous().view(torch.uint8).reshape(-1)
        stride = (b.numel() + 255) // 256 * 256
        big = torch.zeros(K * stride, dtype=torch.uint8, device=t.device)
        big.view(K, stride)[:, : b.numel()] = b
        return [big[k * stride: k * stride + b.numel()] for k in range(K)]

    dev = dh.device
    X = torch.zeros(K, n * nv, dtype=torch.float64, device=dev)
    OB = torch.zeros(K, n, dtype=torch.float64, device=dev)
    ST = torch.full((K, n), -1, dtype=torch.int32, device=dev)
    INF = torch.zeros(K, n * 32, dtype=torch.uint8, device=dev)
    return K, per + per_out, (rep(dh), rep(dr), rep(do), rep(ds), rep(dxi), X, OB, ST, INF)


def cold_measure(torch, api, sol, desc_kwargs, n, n_obs, tensors, max_copies=1024):
    """The same batch with NOTHING of it in a cache when a launch reads it.  The hot figures of this file re-solve ONE device-resident batch:
    from the second launch on its rows come out of L2 / the 256 MiB Infinity Cache (FETCH_SIZE counts those hits as traffic; the guide says to
    scale past L3 before reading it).  Here K copies of the batch -- inputs AND outputs, K x bytes >= 512 MiB, twice the Infinity Cache --
    are solved round robin: by the time copy j comes round again, 512 MiB of other copies have passed through every cache level.  One untimed
    round, then one timed round of K launches (HIP events on the launch stream) of (a) the whole solve call and (b) the dual active-set
    phase alone (LSCQP_ACTIVE_SET_ONLY: the kernel the roofline is quoted on).  The copies hold the same numbers: the results are the hot
    run's, bit for bit (checked)."""
    nv = sol.nv
    K, per_all, (H, R, O, S_, XI, X, OB, ST, INF) = cold_copies(torch, tensors, n, nv, max_copies)
    only = api.Solver(api.make_desc(active_set=api.ACTIVE_SET_ONLY, **desc_kwargs))
    out = {"copies": K, "bytes_per_copy": per_all, "bytes_all_copies": K * per_all}
    for tag, sv in (("step_ms", sol), ("kernel_ms", only)):
        calls = [sv.bind_device(n, n_obs, H[k], R[k], O[k], S_[k], X[k], OB[k], ST[k], INF[k], d_x_init=XI[k]) for k in range(K)]
        for c in calls:  # one untimed round: every copy has been read once and pushed out again by the K - 1 after it
            c()
        torch.cuda.synchronize()
        out[tag] = time_calls(torch, calls, warm=0)
        if tag == "step_ms":
            same = bool(torch.equal(X[0], X[K - 1]) and torch.equal(X[0], X[K // 2]) and torch.equal(ST[0], ST[K - 1]))
            out["copies_bit_identical"] = same
            out["non_optimal"] = int((ST[K - 1] != 0).sum().item())
    only.close()
    bq = sol.algorithmic_bytes(n_obs)
    out["frac"] = bq * n / (out["kernel_ms"] * 1e-3) / HBM_PEAK
    out["frac_step"] = bq * n / (out["step_ms"] * 1e-3) / HBM_PEAK
    out["achieved_GBps"] = bq * n / (out["kernel_ms"] * 1e-3) / 1e9
    out["what"] = ("%d copies of the batch (inputs and outputs, %.0f MiB in all: twice the 256 MiB Infinity Cache) solved round robin, one untimed round, "
                   "then one timed round; kernel_ms = the dual active-set phase alone, step_ms = the whole solve call" % (K, K * per_all / 2**20))
    del H, R, O, S_, XI, X, OB, ST, INF
    return out


# BASELINE.json configs at their own shapes (SURVEY.md section 8 size table).  "c1" is the headline (the metric is quoted on it);
# the others are measured on one GPU in the `configs` section of the JSON line and can be made the timed workload with
# --config (that is how tools/profile_round.py traces each kernel instance on its own).  `seed`: ONE batch per config -- the `configs`
# block of the default line, `--config cX` (what the rocprofv3 passes under profiles/ run) and the sharded `--scaling strong` runs all
# build the swarm from it (weak scaling: + rank), so the profiles are of the very batches the bench line reports.
CONFIGS = {
    "c0": dict(seed=3020, agents=10, segments=10, obs=9, dim=2, style="forest", precision="f64", rows="f64",
               what="configs[0] forest10 replica: 10 agents x M=10, dim 2 (the reference's own launch shape; CPLEX there)"),
    "c1": dict(seed=1000, agents=64, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64",
               what="configs[1]: 64 agents x M=5 x ~20 LSC half-spaces/seg, fp64"),
    "c2": dict(seed=3518, agents=512, segments=6, obs=20, dim=3, style="maze", precision="f64", rows="f64",
               what="configs[2]: 512 agents dense-maze LSC set, M=6, fp64 (whole batch on one GPU)"),
    "c3s": dict(seed=3138, agents=128, segments=10, obs=40, dim=3, style="forest", precision="f64", rows="f64",
                what="configs[3], the per-GPU shard of 8: 128 agents x M=10 x 40 LSC + SFC, fp64 (nz = 84)"),
    "c3": dict(seed=4034, agents=1024, segments=10, obs=40, dim=3, style="forest", precision="f64", rows="f64",
               what="configs[3] whole: 1024 agents x M=10 x 40 LSC + SFC, fp64, on ONE GPU"),
    "c4": dict(seed=7101, agents=4096, segments=5, obs=20, dim=3, style="forest", precision="mixed", rows="f32",
               what="configs[4]: 4096 agents x M=5, fp32 PDIP (float32 factorisation, 16-byte rows) with fp64 residual check"),
    "c4_f64": dict(seed=7101, agents=4096, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64",
                   what="configs[4] shape in fp64 (32-byte rows): the comparison the mixed-precision instance is judged against"),
    # round 6: the same swarms LATER in their exchange (BASELINE's batches are taken 3 replans after hover, when most QPs hold no row at the
    # optimum; the reference's own missions are busier: log/simulation_..., forest10 mid-mission) and with SURVEY.md 8d's controlled fraction of
    # infeasible instances.  Checkpoints from tools/loaded_probe.py: the replans where each swarm's step counts peak.
    "c1_loaded": dict(seed=1000, agents=64, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64", warm_steps=25,
                      what="configs[1] swarm 25 replans into its exchange"),
    "c0_loaded": dict(seed=3020, agents=10, segments=10, obs=9, dim=2, style="forest", precision="f64", rows="f64", warm_steps=12,
                      what="configs[0] forest10 replica 12 replans into its exchange (the class's busiest stretch: > 20 active rows at one agent)"),
    "c2_loaded": dict(seed=3518, agents=512, segments=6, obs=20, dim=3, style="maze", precision="f64", rows="f64", warm_steps=20,
                      what="configs[2] dense-maze swarm 20 replans into its exchange"),
    "c4_loaded": dict(seed=7101, agents=4096, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64", warm_steps=8,
                      what="configs[4] shape (fp64 rows) 8 replans into its exchange"),
    "c1_infeasible_1pct": dict(seed=1000, agents=64, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64", infeasible_frac=1 / 64,
                               what="configs[1] batch with ONE of its 64 instances (1.6 %) made infeasible (SURVEY.md 8d)"),
    "c4_infeasible_1pct": dict(seed=7101, agents=4096, segments=5, obs=20, dim=3, style="forest", precision="f64", rows="f64", infeasible_frac=0.01,
                               what="configs[4] shape (fp64 rows) with 1 % of its instances (41) made infeasible (SURVEY.md 8d)"),
}


def percentile_latency(torch, call, max_calls=1050, max_seconds=6.0, skip=50):
    """p50 / p99 of enqueue -> results readable over up to `max_calls` calls (SURVEY.md 8d asks for >= 1000), cut at a time budget."""
    lat = []
    t_end = time.perf_counter() + max_seconds
    for i in range(max_calls):
        a = time.perf_counter()
        call()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - a)
        if time.perf_counter() > t_end and i >= skip + 200:
            break
    lat = np.array(lat[skip:]) * 1e3
    return float(np.percentile(lat, 50)), float(np.percentile(lat, 99)), int(len(lat))


def path_stats(api, info, status):
    """Which of the two kernels finished how many instances of a launch (round 5: the dual active-set phase runs in front of the
    interior-point kernel; lscqp_info.iterations counts active-set STEPS for the former and interior-point ITERATIONS for the latter)."""
    by_as = (info["flags"] & api.INFO_ACTIVE_SET) != 0
    ip = ~by_as & (status == 0)
    it = info["iterations"]
    return {"active_set_solved": int(by_as.sum()), "active_set_steps_mean": float(it[by_as].mean()) if by_as.any() else 0.0,
            "active_set_steps_max": int(it[by_as].max()) if by_as.any() else 0,
            "interior_point_solved": int(ip.sum()), "interior_point_iters_mean": float(it[ip].mean()) if ip.any() else 0.0,
            "interior_point_iters_max": int(it[ip].max()) if ip.any() else 0}


def active_set_kernel_ms(torch, api, desc_kwargs, n, n_obs, tensors, reps=20):
    """Average duration of ONE launch of the dual active-set kernel (lscqp_das::das_kernel) on the given device-resident batch: the phase
    alone (LSCQP_ACTIVE_SET_ONLY), `reps` launches back to back between two HIP events on the launch stream.  None if the phase is off."""
    if os.environ.get("LSCQP_ACTIVE_SET", "1")[:1] == "0" or os.environ.get("LSCQP_ACTIVE_SET_NOW", "1")[:1] == "0":
        return None
    sol = api.Solver(api.make_desc(active_set=api.ACTIVE_SET_ONLY, **desc_kwargs))
    dh, dr, do, ds, dxi = tensors
    dx = torch.zeros(n * sol.nv, dtype=torch.float64, device=dh.device)
    dob = torch.zeros(n, dtype=torch.float64, device=dh.device)
    dst = torch.zeros(n, dtype=torch.int32, device=dh.device)
    dinfo = torch.zeros(n * 32, dtype=torch.uint8, device=dh.device)
    for _ in range(3):
        sol.solve_device(n, n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sol.solve_device(n, n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)
    e1.record()
    torch.cuda.synchronize()
    sol.close()
    return e0.elapsed_time(e1) / reps


def oracle_sample(O, sw, build, M, dim, n_obs_eff, sample, threads=4):
    """The first `sample` QPs of the batch solved once by the CPU oracle (the checker): (result dict, selected indices)."""
    N = len(build["p0"])
    sel = np.arange(min(sample, N))
    cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    ag = np.zeros(len(sel), O.AGENT_DTYPE)
    for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
        ag[f] = build[f][sel]
    ag["vmax"], ag["amax"], ag["radius"], ag["nominal_velocity"], ag["n_obs"] = 1.0, 2.0, 0.15, 1.0, n_obs_eff
    R = O.solve_batch(cls, ag, np.ascontiguousarray(build["lsc"][sel]).reshape(-1), np.arange(len(sel)) * n_obs_eff * M * 6,
                      np.ascontiguousarray(build["sfc"][sel]).reshape(-1), threads=threads)
    return R, sel


def usable_cpus():
    """Host threads this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            avail = max(1, min(avail, int(float(q) / float(p))))
    except Exception:
        pass
    return avail


def oracle_baseline(O, sw, build, M, dim, n_obs_eff, sample, budget_s=2.5, single_core=False):
    """The CPU oracle (a port: CPLEX cannot exist here) on a bounded sample of the batch: QP/s on a FIXED thread count --
    min(64, usable CPUs, sample size), one QP per thread at a time (OpenMP over agents) -- the same rule for every config."""
    N = len(build["p0"])
    sel = np.arange(min(sample, N))
    cls = O.make_class(M=M, dim=dim, use_sfc=True, world_min=sw.world_min, world_max=sw.world_max)
    ag = np.zeros(len(sel), O.AGENT_DTYPE)
    for f in ("p0", "v0", "a0", "goal", "next_waypoint"):
        ag[f] = build[f][sel]
    ag["vmax"], ag["amax"], ag["radius"], ag["nominal_velocity"], ag["n_obs"] = 1.0, 2.0, 0.15, 1.0, n_obs_eff
    lsc = np.ascontiguousarray(build["lsc"][sel]).reshape(-1)
    loff = np.arange(len(sel)) * n_obs_eff * M * 6
    sfc_o = np.ascontiguousarray(build["sfc"][sel]).reshape(-1)
    avail = usable_cpus()
    cores = max(1, min(64, avail, len(sel)))
    R = O.solve_batch(cls, ag, lsc, loff, sfc_o, threads=cores)  # warm (thread pool, caches)
    reps, tcpu = 0, 0.0
    a = time.perf_counter()
    while tcpu < budget_s and reps < 200:
        R = O.solve_batch(cls, ag, lsc, loff, sfc_o, threads=cores)
        reps += 1
        tcpu = time.perf_counter() - a
    out = dict(value=len(sel) * reps / tcpu, unit="QP/s", cores=cores, kind="port", visible_cpus=avail,
               sample="first %d QPs of the batch x %d repetitions, oracle/lscqp_oracle.c (dense fp64 PDIP, OpenMP over agents, %d threads = "
                      "min(64, usable CPUs, sample)), %.1f s wall" % (len(sel), reps, cores, tcpu))
    if single_core:
        k = min(16, len(sel))
        t1c = 1e30  # single core: the first 16 QPs, best of 5 (a lone 30 ms sample is at the mercy of the host's clock ramp)
        for _ in range(5):
            a1 = time.perf_counter()
            O.solve_batch(cls, ag[:k], lsc, loff[:k], sfc_o, threads=1)
            t1c = min(t1c, time.perf_counter() - a1)
        out["single_core_value"] = k / t1c
        out["sample"] += "; single core: %.1f QP/s" % (k / t1c)
    return out, R, sel


def make_infeasible(api, rows, hdr, n_obs, M, frac, seed):
    """"a controlled fraction of deliberately infeasible instances (e.g. 1 %: overlapping agents) ... to exercise the
    status/fallback path" (the reference: QPFAILED, src/traj_optimizer.cpp:143,152 -> the caller keeps initial_traj, src/traj_planner.cpp:767-797).
    The chosen instances get what two agents INSIDE each other's collision model produce: the rows of their second neighbour become the
    mirror image of the first neighbour's, 0.4 m apart -- n.c >= b and n.c <= b - 0.4 for every control point -- a row system without a
    point.  Returns (rows copy, indices)."""
    N = len(hdr)
    rng = np.random.default_rng(seed)
    k = max(1, int(round(frac * N)))
    sel = np.sort(rng.choice(N, size=k, replace=False))
    r = np.array(rows, copy=True).reshape(N, n_obs, M * 6)
    for q in sel:
        for f in ("nx", "ny", "nz"):
            r[f][q, 1] = -r[f][q, 0]
        r["b"][q, 1] = -r["b"][q, 0] + 0.4
    return r.reshape(-1), sel


def measure_config(torch, api, synth, dev, key, cfg, O=None, reps=20, lat_seconds=4.0, cold=False, phase_off=True):
    """One BASELINE config at its own shape on this GPU: kernel time (HIP events), QP/s, latency percentiles, HBM fraction,
    iterations, oracle parity + CPU baseline on a bounded sample.  cfg may carry `warm_steps` (replans after hover before the timed batch is
    taken: 3 for BASELINE's batches, more for the *_loaded blocks) and `infeasible_frac` (make_infeasible).  phase_off: the same batch with the
    dual active-set phase switched off as well (the interior-point kernel alone, rounds 1-4) -- `phase_off` in the block.  cold: cold_measure."""
    N, M, dim, n_obs = cfg["agents"], cfg["segments"], cfg["dim"], cfg["obs"]

    def factory(sw, **kw):
        return api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, **kw))

    sw, sol64, b, (hdr, rows, off, sfc) = make_batch(api, synth, factory, N, M, dim, n_obs, seed=cfg["seed"], style=cfg["style"],
                                                     warm_steps=cfg.get("warm_steps", 3))
    bad_sel = None
    if cfg.get("infeasible_frac"):
        rows, bad_sel = make_infeasible(api, rows, hdr, sw.n_obs, M, cfg["infeasible_frac"], cfg["seed"] + 17)
    kw = {}
    if cfg.get("warm_start") == "tight":
        kw["warm_start"] = api.WARM_TIGHT
    if cfg["precision"] == "mixed":
        kw["precision"] = api.PRECISION_MIXED
    if cfg["rows"] == "f32":
        kw["row_format"] = api.ROWS_F32
    sol = factory(sw, **kw) if kw else sol64
    nv = sol.nv
    dh, do, ds = (to_dev(torch, a, dev) for a in (hdr, off, sfc))
    dr = to_dev(torch, sol.rows_in_format(rows), dev)
    dxi = torch.from_numpy(api.x_init_from_swarm(b, dim)).to(dev)
    dx = torch.zeros(N * nv, dtype=torch.float64, device=dev)
    dob = torch.zeros(N, dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    dinfo = torch.zeros(N * 32, dtype=torch.uint8, device=dev)

    d_order = [None]
    bound = [sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)]

    def call():
        if d_order[0] is not None:  # what a replan does: sort by the previous solve's counts (they are in dinfo), then solve in that order
            sol.order_by_work_device(N, dinfo, d_order[0])
        bound[0]()

    def timed(k):
        return time_calls(torch, [call] * k)

    ms_as_given = timed(max(5, reps // 2))
    # the work order a planner has: the agents whose previous QP took the most iterations first (lscqp_order_by_work_device on the info
    # records of the previous solve -- lscqp_plan carries it from replan to replan).  Here the previous solve is of the SAME batch: the hint
    # is perfect; tools/lpt_probe.py shows the same figures with a hint that is off by one iteration on half of the instances.
    # Only where a launch can have a tail (more instances than lscqp_launch_capacity; lscqp_plan's rule): the sort is part of
    # every timed call, as it is part of every replan.
    phase_on = os.environ.get("LSCQP_ACTIVE_SET", "1")[:1] != "0" and os.environ.get("LSCQP_ACTIVE_SET_NOW", "1")[:1] != "0"
    if N > sol.launch_capacity(N, sw.n_obs) and not phase_on:  # (round 5: with the dual active-set phase in front there is no iteration tail to sort)
        d_order[0] = torch.zeros(N, dtype=torch.int32, device=dev)
        bound[0] = sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi, d_order=d_order[0])
        ms = timed(reps)
    else:
        ms = ms_as_given
    spread = sorted(timed(reps) for _ in range(5))
    # >= 1000 calls where that fits a bounded time (the slowest config, 1024 x M=10, needs ~10 s for them)
    p50, p99, nlat = percentile_latency(torch, call, max_seconds=min(14.0, max(lat_seconds, 1.15e-3 * ms * 1100)))
    info = dinfo.cpu().numpy().view(api.INFO_DTYPE)
    st = dst.cpu().numpy()
    bq = sol.algorithmic_bytes(sw.n_obs)
    by_as = ((info["flags"] & api.INFO_ACTIVE_SET) != 0) & (st == 0)
    out = {"config": key, "what": cfg["what"], "agents": N, "segments": M, "dim": dim, "lsc_neighbours": sw.n_obs, "style": cfg["style"],
           "batch_seed": cfg["seed"], "replans_after_hover": cfg.get("warm_steps", 3),
           "precision": cfg["precision"], "rows": cfg["rows"], "rows_per_qp": sol.num_inequalities(sw.n_obs),
           "kernel_ms": ms, "qp_per_s": N / (ms * 1e-3), "latency_ms": {"p50": p50, "p99": p99, "calls": nlat},
           "ms_spread": {"min": spread[0], "median": spread[2], "max": spread[-1], "repeats": 5, "launches_per_repeat": reps},
           "work_order": ("longest first: lscqp_order_by_work_device on the previous solve's iteration counts, inside every timed call (same batch: "
                          "a perfect hint)") if d_order[0] is not None else "as given (the launch starts every instance at once)",
           "kernel_ms_as_given": ms_as_given, "qp_per_s_as_given": N / (ms_as_given * 1e-3),
           "algorithmic_bytes_per_qp": bq, "hbm_GBps": bq * N / (ms * 1e-3) / 1e9, "hbm_frac": bq * N / (ms * 1e-3) / HBM_PEAK,
           "iters_mean": float(info["iterations"].mean()), "iters_max": int(info["iterations"].max()), "paths": path_stats(api, info, st),
           "active_set_steps_histogram": np.bincount(np.minimum(info["iterations"][by_as], 64), minlength=1).tolist() if by_as.any() else [],
           "active_set_kernel_ms": active_set_kernel_ms(torch, api, dict(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max,
                                                                       **{k_: v_ for k_, v_ in kw.items() if k_ == "row_format"}), N, sw.n_obs, (dh, dr, do, ds, dxi)),
           "roofline_valu": (valu_roofline(sol, N, sw.n_obs, info["iterations"][((info["flags"] & api.INFO_ACTIVE_SET) == 0) & (st == 0)], ms)
                             if cfg["precision"] == "f64" else None),
           "non_optimal": int((st != 0).sum()), "second_pass": int(((info["flags"] & api.INFO_REPAIRED) != 0).sum()),
           "floor_accepted": int(((info["flags"] & api.INFO_FLOOR_ACCEPTED) != 0).sum()),
           "statuses": np.bincount(st, minlength=5).tolist()}
    out["paths"]["proven_infeasible_by_phase"] = int((((info["flags"] & api.INFO_ACTIVE_SET) != 0) & (st == api.STATUS_INFEASIBLE)).sum())
    x_on = dx.cpu().numpy().reshape(N, nv).copy()
    if phase_off and phase_on and cfg["precision"] == "f64":
        # the SAME batch with the phase off: the interior-point kernel alone (rounds 1-4) -- a launch shape where the phase does not pay shows here
        sol.set_knob("active_set_off", 1)
        try:
            off_call = sol.bind_device(N, sw.n_obs, dh, dr, do, ds, dx, dob, dst, dinfo, d_x_init=dxi)
            ms_off = time_calls(torch, [off_call] * max(5, reps // 2))
            st_off = dst.cpu().numpy()
            io = dinfo.cpu().numpy().view(api.INFO_DTYPE)
            both = (st == 0) & (st_off == 0)
            out["phase_off"] = {"kernel_ms": ms_off, "qp_per_s": N / (ms_off * 1e-3), "speedup_of_phase": ms_off / ms, "non_optimal": int((st_off != 0).sum()),
                                "iters_mean": float(io["iterations"].mean()), "iters_max": int(io["iterations"].max()),
                                "statuses_equal": bool(np.array_equal(st_off != 0, st != 0)),
                                "fp64_valu": valu_roofline(sol, N, sw.n_obs, io["iterations"][st_off == 0], ms_off),
                                "max_abs_dx_vs_phase_on": float(np.abs(dx.cpu().numpy().reshape(N, nv) - x_on)[both].max()) if both.any() else None}
        finally:
            sol.set_knob("active_set_off", 0)
        bound[0]()  # (the buffers hold the default path's results again)
        torch.cuda.synchronize()
    if bad_sel is not None:
        out["infeasible"] = {"instances": [int(v) for v in bad_sel[:16]], "count": int(len(bad_sel)), "fraction": float(len(bad_sel)) / N,
                             "all_reported_non_optimal": bool((st[bad_sel] != 0).all()), "others_optimal": bool((np.delete(st, bad_sel) == 0).all()),
                             "how": "second neighbour's rows := mirror image of the first neighbour's, 0.4 m apart (two agents inside each other's model)"}
    if cold and phase_on:
        try:
            out["cold"] = cold_measure(torch, api, sol, dict(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max,
                                                             **{k_: v_ for k_, v_ in kw.items() if k_ == "row_format"}), N, sw.n_obs, (dh, dr, do, ds, dxi))
        except Exception as ex:  # noqa: BLE001
            out["cold"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    if O is not None:
        # (rows rounded to float32 are a different problem instance than the oracle's fp64 rows: parity is taken on fp64 rows)
        cpu, R, sel = oracle_baseline(O, sw, b, M, dim, sw.n_obs, sample=64 if M < 10 else 32)
        out["cpu_baseline"] = cpu
        if cfg["rows"] == "f64" and bad_sel is None:
            xg, og = x_on[sel], dob.cpu().numpy()[sel]
            ok = (R["status"] == 0) & (st[sel] == 0)
            if ok.any():
                out["parity_vs_oracle"] = {"max_abs_dx": float(np.abs(xg - R["x"])[ok].max()), "compared": int(ok.sum()),
                                           "max_rel_dobj": float((np.abs(og - R["obj"]) / np.maximum(1.0, np.abs(R["obj"])))[ok].max()),
                                           "status_disagreements": int(((R["status"] == 0) != (st[sel] == 0)).sum())}
    return out


def newest_profile(pattern):
    """profiles/<tag>_... with the highest (round, version) tag, numerically: r02_v11 > r02_v9 > r01_v11."""
    import glob
    import re

    def tagkey(path):
        m = re.match(r"r(\d+)_v(\d+)", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)

    fs = sorted(glob.glob(os.path.join(HERE, "profiles", pattern)), key=tagkey)
    return fs[-1] if fs else None


def replan_chain(torch, api, replans=60):
    """The whole replan of configs[0]'s mission as ONE chain of device work (lscqp_plan: shifted plans, range filter, CLSC rows,
    corridors over the forest10 voxel map, goal LP, QP, failsafe, doStep), eager and through the captured hipGraph: informational,
    next to the QP figures above.  Closed loop on the device; every agent flies towards a waypoint one grid step from its start."""
    import time

    W = json.load(open(os.path.join(HERE, "tests", "golden", "forest10_world.json")))
    N = len(W["starts"])
    sol = api.Solver(api.make_desc(M=10, dim=2, dt=0.2, world_min=W["world_min"], world_max=W["world_max"]))
    wmap = api.WorldMap(W["boxes"], W["world_min"], W["world_max"], W["resolution"], W["max_dist"])
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = W["radius"], 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, N - 1, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, closed_loop=True, z_2d=W["z_2d"])
    starts, goals = np.array(W["starts"], dtype=np.float64), np.array(W["goals"], dtype=np.float64)
    way = starts.copy()
    way[:, :2] += 0.5 * np.sign(np.round(goals[:, :2] - starts[:, :2], 6))
    res = {"workload": "forest10 mission: 10 agents x M10 x 9 neighbour slots, CLSC + corridors + goal LP + QP per replan", "replans": replans}
    for mode in ("eager", "graph"):
        plan.reset(starts)
        plan.put(api.PLAN_WAYPOINT, starts)
        plan.step()
        plan.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
        plan.step(graph=(mode == "graph"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(replans):
            plan.step(graph=(mode == "graph"))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[mode] = {"us_per_replan": (t2 - t0) / replans * 1e6, "host_submit_us_per_replan": (t1 - t0) / replans * 1e6,
                     "failed_qps_last_replan": int((plan.get(api.PLAN_STATUS) != 0).sum())}
    res["graph_nodes"] = plan.graph_nodes()
    plan.close()
    wmap.close()
    res["c1_class"] = replan_chain_3d(torch, api)
    return res


def replan_chain_3d(torch, api, N=64, replans=41, radii=(10.0, 10.0, 4.0), skip=1):
    """The same chain at configs[1]'s class, in the reference's default modes: 64 agents x M = 5 in three dimensions (downwash 2), 20
    neighbour slots, CLSC rows, corridors from the convex hull over a room with 24 boxes, goal LP -- agents on a sphere swapping sides,
    closed loop through the captured graph.  The host plays the router (a waypoint 0.75 m ahead on the straight line to the goal,
    written before every replan); the time is the step's alone (launch to completion), over the mission from the second replan on
    (the first one differs: initializeSFC, and it is not in the graph); `us_per_replan_after_12`: the same without the first dozen,
    when the agents have left the walls of the room (the corridors along a wall are the expensive ones)."""
    import time

    rng = np.random.default_rng(7)
    i = np.arange(N) + 0.5
    phi, th = np.arccos(1 - 2 * i / N), np.pi * (1 + 5 ** 0.5) * i  # Fibonacci lattice
    starts = np.round((np.c_[np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)] * list(radii) + [0, 0, radii[2] + 1.0]) * 4) / 4
    goals = np.c_[-starts[:, 0], -starts[:, 1], 2 * (radii[2] + 1.0) - starts[:, 2]]
    boxes = []
    while len(boxes) < 24:
        c = np.r_[rng.uniform(-0.7 * radii[0], 0.7 * radii[0], 2), rng.uniform(1.5, 2 * radii[2] + 0.5)]
        if np.abs(starts - c).max(axis=1).min() > 1.2:
            boxes.append([c[0], c[1], c[2], 0.8, 0.8, 0.8])
    wmin, wmax = [-radii[0] - 2.0, -radii[1] - 2.0, 0.0], [radii[0] + 2.0, radii[1] + 2.0, 2 * radii[2] + 2.0]
    sol = api.Solver(api.make_desc(M=5, dim=3, dt=0.2, world_min=wmin, world_max=wmax))
    wmap = api.WorldMap(boxes, wmin, wmax, 0.1, 1.0)
    ag = np.zeros(N, api.AGENT_PARAM_DTYPE)
    ag["radius"], ag["downwash"], ag["max_vel"], ag["max_acc"], ag["nominal_velocity"] = 0.15, 2.0, 1.0, 2.0, 1.0
    plan = api.Plan(sol, wmap, N, 20, ag, constraint_mode=api.GEN_CLSC, sfc_mode=api.SFC_FROM_HULL, optimize_goal=True, closed_loop=True)
    plan.reset(starts)
    t_dev, failed, cut, it_sum, it_max = 0.0, 0, 0, 0.0, 0
    t_late, n_late = 0.0, 0
    for k in range(skip + replans):
        pos = plan.get(api.PLAN_STATE).reshape(N, 9)[:, :3]
        d = goals - pos
        dist = np.linalg.norm(d, axis=1, keepdims=True)
        way = pos + d / np.maximum(dist, 1e-9) * np.minimum(dist, 0.75)
        plan.put(api.PLAN_WAYPOINT, np.float32(way).astype(np.float64))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.step(graph=True)
        torch.cuda.synchronize()
        if k >= skip:
            dt_step = time.perf_counter() - t0
            t_dev += dt_step
            if k >= 12:
                t_late, n_late = t_late + dt_step, n_late + 1
            info = plan.get(api.PLAN_INFO)
            failed += int((plan.get(api.PLAN_STATUS) != 0).sum())
            cut += int((plan.get(api.PLAN_IN_RANGE) > 20).sum())
            it_sum += float(info["iterations"].mean())
            it_max = max(it_max, int(info["iterations"].max()))
    moved = float(np.linalg.norm(plan.get(api.PLAN_STATE).reshape(N, 9)[:, :3] - starts, axis=1).mean())
    res = {"workload": "64 agents x M5 x 20 neighbour slots in 3-D: CLSC rows + corridors + goal LP + QP per replan, hipGraph, waypoints from the host",
           "replans": replans, "us_per_replan": t_dev / replans * 1e6, "us_per_replan_after_12": t_late / max(n_late, 1) * 1e6, "failed_qps": failed, "cut_neighbour_lists": cut,
           "iters_mean": it_sum / replans, "iters_max": it_max, "mean_distance_flown_m": moved}
    plan.close()
    wmap.close()
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def single_process_main(args, torch):
    """--single-process: ONE host process drives the G devices through lscqp_comm (the deployment of the reference's single ROS
    process): the config's global batch in contiguous blocks of ceil(N/G), lscqp_solve_batch_sharded_device on the
    communicator's streams, then lscqp_allgather of the plans (ncclCommInitAll communicators, grouped ncclAllGather) every step."""
    from lsc_dr_planner_amd import api, sharding, synth

    G, M, dim, n_obs, n_glob = args.gpus, args.segments, args.dim, args.obs, args.agents
    torch.cuda.set_device(0)
    kw = {}
    if args.precision == "mixed":
        kw["precision"] = api.PRECISION_MIXED
    if args.rows == "f32":
        kw["row_format"] = api.ROWS_F32

    def factory(sw):
        return api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, **kw))

    sw, sol, build, (hdr, rows, off, sfc) = make_batch(api, synth, factory, n_glob, M, dim, n_obs, seed=CONFIGS[args.config]["seed"], style=args.style, warm_steps=3)
    nv, n_obs_eff, per = sol.nv, sw.n_obs, -(-n_glob // G)
    comm = api.Comm(G)
    comm.set_min_agents_per_device(1)  # this run is TOLD how many devices to use; the library's own rule is reported below
    rows2, sfc2, x0 = sol.rows_in_format(rows).reshape(n_glob, -1), sfc.reshape(n_glob, M), api.x_init_from_swarm(build, dim)
    blocks, T = [], {k: [] for k in ("hdr", "rows", "off", "sfc", "xi", "x", "obj", "st", "info", "all")}
    for g in range(G):
        lo, hi = sharding.shard_range(n_glob, G, g)
        blocks.append((lo, hi))
        dev = torch.device("cuda", g)
        n = hi - lo
        T["hdr"].append(to_dev(torch, hdr[lo:hi], dev))
        T["rows"].append(to_dev(torch, rows2[lo:hi], dev))
        T["off"].append(to_dev(torch, np.arange(n + 1, dtype=np.uint64) * np.uint64(n_obs_eff * M * 6), dev))
        T["sfc"].append(to_dev(torch, sfc2[lo:hi], dev))
        T["xi"].append(None if args.cold_start else torch.from_numpy(np.ascontiguousarray(x0[lo:hi])).to(dev))
        T["x"].append(torch.zeros(per * nv, dtype=torch.float64, device=dev))
        T["obj"].append(torch.zeros(max(n, 1), dtype=torch.float64, device=dev))
        T["st"].append(torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev))
        T["info"].append(torch.zeros(max(n, 1) * 32, dtype=torch.uint8, device=dev))
        T["all"].append(torch.zeros(G * per * nv, dtype=torch.float64, device=dev))
    counts = [hi - lo for lo, hi in blocks]

    def solve():
        sol.solve_sharded_device(comm, counts, n_obs_eff, T["hdr"], T["rows"], T["off"], T["sfc"], T["x"], T["obj"], T["st"], T["info"],
                                 d_x_init=None if args.cold_start else T["xi"])

    def step():
        solve()
        comm.allgather(T["x"], T["all"], per * nv)

    for _ in range(args.warmup):
        step()
    comm.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    comm.synchronize()
    elapsed = time.perf_counter() - t0
    # the solve kernel alone, HIP events on device 0's communicator stream (the stream the kernel is launched on)
    s0 = torch.cuda.ExternalStream(comm.stream(0), device=torch.device("cuda", 0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    for _ in range(20):
        solve()
    e1.record(s0)
    comm.synchronize()
    kernel_ms = e0.elapsed_time(e1) / 20
    lat = []
    for _ in range(0 if args.no_latency else 260):
        a = time.perf_counter()
        step()
        comm.synchronize()
        lat.append(time.perf_counter() - a)
    lat = np.array(lat[10:] or [float("nan")]) * 1e3
    step()
    comm.synchronize()
    st = np.concatenate([T["st"][g].cpu().numpy()[: counts[g]] for g in range(G)])
    it = np.concatenate([T["info"][g].cpu().numpy().view(api.INFO_DTYPE)["iterations"][: counts[g]] for g in range(G)])
    x_all = np.concatenate([T["x"][g].cpu().numpy()[: counts[g] * nv] for g in range(G)]).reshape(n_glob, nv)
    for g in range(G):  # every device must hold every owner's block after the exchange
        view = T["all"][g].cpu().numpy().reshape(G, per * nv)
        got = np.concatenate([view[o][: counts[o] * nv] for o in range(G)]).reshape(n_glob, nv)
        if not np.array_equal(got, x_all):
            raise SystemExit("bench.py: device %d does not hold the gathered plans of all owners" % g)
    parity = None
    if not args.no_cpu_baseline and args.rows == "f64":
        from oracle import oracle as O

        dxm, dom, cmp_n = 0.0, 0.0, 0
        for g, (lo, hi) in enumerate(blocks):  # a sample of EVERY device's block against the oracle
            bg = slice_build(build, n_glob, lo, hi)
            Rk, sel = oracle_sample(O, sw, bg, M, dim, n_obs_eff, sample=min(hi - lo, 8 if M < 10 else 4))
            og = T["obj"][g].cpu().numpy()[sel]
            ok = (Rk["status"] == 0) & (st[lo:hi][sel] == 0)
            dxm = max(dxm, float(np.abs(x_all[lo:hi][sel] - Rk["x"])[ok].max()))
            dom = max(dom, float((np.abs(og - Rk["obj"]) / np.maximum(1.0, np.abs(Rk["obj"])))[ok].max()))
            cmp_n += int(ok.sum())
        parity = {"max_abs_dx": dxm, "max_rel_dobj": dom, "compared": cmp_n, "devices": G}
    bq = sol.algorithmic_bytes(n_obs_eff)
    achieved = bq * counts[0] / (kernel_ms * 1e-3)
    out = {"metric": "qp_solves_per_sec", "value": n_glob * args.steps / elapsed, "unit": "QP/s", "n_gpus": G, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64" if args.precision == "f64" else "f64 iterate and residuals, f32 factorisation", "data": "synthetic",
           "config": {"workload": "%s -- ONE batch of %d agents cut into blocks of <= %d per GPU x M=%d x %d LSC neighbours (dim=%d, %s swarm "
                                  "after 3 warm-up replans), solve + all-gather of the plans per step" % (CONFIGS[args.config]["what"], n_glob, per, M, n_obs_eff, dim, args.style),
                      "baseline_config": args.config, "precision": args.precision, "row_format": args.rows,
                      "collective_backend": comm.backend, "launch": "single process, lscqp_comm (one stream + one RCCL communicator per device)",
                      "agents_per_gpu": per, "agents_total": n_glob, "ranks": 1, "distinct_devices": G,
                      "devices_by_crossover_rule": int(max(1, min(G, n_glob // 256))), "allgather": True,
                      "allgather_bytes_per_rank_per_step": int(per * nv * 8), "segments": M, "lsc_neighbours": n_obs_eff, "dim": dim,
                      "parallelism": "agents sharded over %d GPU(s), one RCCL all-gather of the plans per step" % G},
           "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                        "traffic": None, "kernel": "lscqp_pdip_kernel<%d,%d,true,NSLOT,W,%s>" % (M, dim, "float" if args.precision == "mixed" else "double"),
                        "kernel_ms": kernel_ms, "algorithmic_bytes_per_qp": bq, "qps_per_launch": counts[0]},
           "latency_ms": {"sharded_step": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "calls": int(len(lat)),
                                           "what": "solve of every block + all-gather, enqueue -> all streams synchronised"}},
           "solver": {"non_optimal": int((st != 0).sum()), "iters_mean": float(it.mean()), "iters_max": int(it.max())}}
    if parity is not None:
        out["parity_all_ranks"] = parity
    print(json.dumps(out))
    comm.close()


class Ctx:
    """What every timed workload of one bench.py process shares: the device, the process group and the reductions over it."""

    def __init__(self, torch, dist, rank, world, dev_index, backend, coll_backend):
        self.torch, self.dist, self.rank, self.world, self.dev_index = torch, dist, rank, world, dev_index
        self.dev = torch.device("cuda", dev_index)
        self.backend, self.coll_backend = backend, coll_backend
        self.host_collectives = coll_backend.startswith("gloo")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, vals, op):
        """all-reduce of a few doubles over the ranks (host tensors under gloo, device tensors under RCCL)."""
        if self.dist is None:
            return [float(v) for v in vals]
        d = self.dist
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=("cpu" if self.host_collectives else self.dev))
        d.all_reduce(t, op={"max": d.ReduceOp.MAX, "min": d.ReduceOp.MIN, "sum": d.ReduceOp.SUM}[op])
        return [float(v) for v in t.cpu()]


def workload_args(base, config, **over):
    """A copy of the parsed arguments with `config`'s shape filled in (explicit --agents/--segments/... of `base` win only when
    base.config is that same config)."""
    a = argparse.Namespace(**vars(base))
    a.config = config
    cfg = CONFIGS[config]
    for k in ("agents", "segments", "obs", "dim", "style", "precision", "rows"):
        if getattr(a, k, None) is None or base.config != config:
            setattr(a, k, cfg[k])
    for k, v in over.items():
        setattr(a, k, v)
    return a


def timed_workload(ctx, a):
    """ONE timed workload on every rank of the job: build the batch (device resident), `a.warmup` untimed steps, then exactly `a.steps`
    steps between barrier + synchronize on both sides; max over ranks.  Returns a namespace with everything the line is made of
    (collective results are identical on every rank) and the device state rank 0 needs for its latency / baseline legs."""
    from types import SimpleNamespace

    from lsc_dr_planner_amd import api, sharding, synth

    torch, dist, rank, world, dev = ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.dev
    M, dim, n_obs, N = a.segments, a.dim, a.obs, a.agents
    cfg_seed = CONFIGS[a.config]["seed"]
    cfg_warm = CONFIGS[a.config].get("warm_steps", 3)
    solver_kw = {}
    if a.precision == "mixed":
        solver_kw["precision"] = api.PRECISION_MIXED
    if a.rows == "f32":
        solver_kw["row_format"] = api.ROWS_F32

    def warm_factory(sw):  # the warm-up replans are always carried in fp64 on fp64 rows: one batch per config whatever the timed precision
        return api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max))

    def all_ranks(fn):
        """Run the collective-free `fn` on every rank and agree on the outcome: a rank that failed alone would leave the others
        waiting in the next collective for ever."""
        err = None
        try:
            res = fn()
        except Exception as ex:  # noqa: BLE001
            res, err = None, "%s: %s" % (type(ex).__name__, str(ex)[:300])
        bad, = ctx.reduce([0.0 if err is None else 1.0], "sum")
        if bad:
            raise RuntimeError(err or "batch construction failed on %d other rank(s)" % int(bad))
        return res

    strong = a.scaling == "strong"
    if strong and (a.pipeline or a.graph):
        raise SystemExit("bench.py: --scaling strong times the solve + all-gather step; --pipeline / --graph are weak-scaling options")
    n_glob = N  # agents of the whole job's step: one global batch (strong) or `world` independent swarms of N (weak)
    if strong:
        # ONE swarm for the whole job, built identically on every rank (the config's seed; the warm-up replans are carried by the rank's
        # own GPU, the kernel is deterministic), then cut: rank r owns the contiguous block shard_range(n_glob, world, r) (reference
        # agent order, src/mission.cpp:140-153).  It is the SAME batch the one-GPU `configs` entry of this config solves whole.
        sw, sol, build, (hdr, rows, off, sfc) = all_ranks(lambda: make_batch(api, synth, warm_factory, n_glob, M, dim, n_obs, seed=cfg_seed, style=a.style, warm_steps=cfg_warm))
        lo, hi = sharding.shard_range(n_glob, world, rank)
        per = -(-n_glob // world)
        N = hi - lo
        if N <= 0:
            raise SystemExit("bench.py: rank %d owns no agent (%d agents over %d ranks)" % (rank, n_glob, world))
        whole = SimpleNamespace(hdr=hdr, rows=rows, off=off, sfc=sfc, build=build) if (rank == 0 and world > 1) else None
        build = slice_build(build, n_glob, lo, hi)
        hdr, sfc = hdr[lo:hi], sfc.reshape(n_glob, M)[lo:hi]
        rows = rows.reshape(n_glob, -1)[lo:hi].reshape(-1)
        off = np.arange(N + 1, dtype=np.uint64) * np.uint64(sw.n_obs * M * 6)
    else:
        whole = None
        sw, sol, build, (hdr, rows, off, sfc) = all_ranks(lambda: make_batch(api, synth, warm_factory, N, M, dim, n_obs, seed=cfg_seed + rank,
                                                                             style=a.style, warm_steps=cfg_warm))
        if CONFIGS[a.config].get("infeasible_frac"):
            rows, _ = make_infeasible(api, rows, hdr, sw.n_obs, M, CONFIGS[a.config]["infeasible_frac"], cfg_seed + 17)
        n_glob = world * N
    if solver_kw:
        sol = api.Solver(api.make_desc(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max, **solver_kw))
    n_obs_eff = sw.n_obs
    nv = sol.nv
    d_hdr, d_off, d_sfc = (to_dev(torch, x, dev) for x in (hdr, off, sfc))
    d_rows = to_dev(torch, sol.rows_in_format(rows), dev)
    # TrajOptimizer::solve's initial_traj (the shifted previous plan): the solver's primal start
    d_xinit = None if a.cold_start else torch.from_numpy(api.x_init_from_swarm(build, dim)).to(dev)
    n_pad = per if strong else N  # equal blocks for the collective: the last block of a ragged split is padded (never solved)
    d_x = torch.zeros(n_pad * nv, dtype=torch.float64, device=dev)
    d_obj = torch.zeros(N, dtype=torch.float64, device=dev)
    d_st = torch.full((N,), -1, dtype=torch.int32, device=dev)
    d_info = torch.zeros(N * 32, dtype=torch.uint8, device=dev)
    gather = (a.allgather or a.pipeline or strong) and world > 1
    d_all = torch.zeros(world * n_pad * nv, dtype=torch.float64, device=dev) if gather else None
    # gloo (test-only backend, ranks sharing a device) carries the all-gather through host memory
    h_all = torch.zeros(world * n_pad * nv, dtype=torch.float64) if (gather and ctx.host_collectives) else None

    def all_gather_plans():
        if h_all is None:
            dist.all_gather_into_tensor(d_all, d_x)
        else:
            dist.all_gather_into_tensor(h_all, d_x.cpu())
            d_all.copy_(h_all)

    d_xwarm = torch.zeros(N * nv, dtype=torch.float64, device=dev)
    if a.pipeline:
        # every rank's swarm is independent (weak scaling); global agent id = rank * N + local id.  The rows are
        # regenerated every step from the CURRENT plans of all agents (replanning from the same state: the previous
        # plans satisfy the new rows by the supporting-hyperplane argument of SURVEY 8d), so the exchange is load bearing.
        n_total = world * N
        d_traj = torch.zeros(n_total * M * 6 * 3, dtype=torch.float64, device=dev)
        d_nbr = torch.from_numpy((build["nbr"] + rank * N).astype(np.int32)).to(dev)
        d_rad = torch.full((n_total,), sw.radius, dtype=torch.float64, device=dev)
        d_dw = torch.full((n_total,), sw.downwash, dtype=torch.float64, device=dev)
        d_goal = torch.from_numpy(np.ascontiguousarray(build["goal"], dtype=np.float64)).to(dev)
        sol.solve_device(N, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_x_init=d_xinit)  # plans to start from
        torch.cuda.synchronize()

    d_order = [None]  # work order of the launch: set after the warm-up steps from their iteration counts (see below)

    def solve_only():
        sol.solve_device(N, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_x_init=d_xinit, d_order=d_order[0])

    # the plain step (the solve of BASELINE's metric alone, in the order given): the same C entry with its arguments converted once
    # (api.Solver.bind_device) -- a 64-QP step lasts ~14 us on the device, and building eleven ctypes pointers per call is of that order
    plain = not (a.pipeline or a.graph or gather)
    bound_solve = sol.bind_device(N, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_x_init=d_xinit) if plain else None

    cold_calls, cold_i = None, [0]
    if a.cold:
        # (profiling aid, tools/profile_round.py: the timed steps rotate over K copies of the batch, K x bytes >= 512 MiB -- every launch reads its
        # rows from HBM, which is what makes rocprofv3's FETCH_SIZE of the launch comparable with the algorithmic bytes; cold_measure)
        if not plain:
            raise SystemExit("bench.py: --cold rotates the plain solve step over copies of the batch; not with --pipeline / --graph / a collective")
        Kc, _, (cH, cR, cO, cS, cXI, cX, cOB, cST, cINF) = cold_copies(torch, (d_hdr, d_rows, d_off, d_sfc, d_xinit), N, nv)
        cold_calls = [sol.bind_device(N, n_obs_eff, cH[k], cR[k], cO[k], cS[k], cX[k], cOB[k], cST[k], cINF[k], d_x_init=cXI[k]) for k in range(Kc)]

    def step():
        if cold_calls is not None:
            cold_calls[cold_i[0] % len(cold_calls)]()
            cold_i[0] += 1
            return
        if plain and d_order[0] is None:
            bound_solve()
            return
        if d_order[0] is not None:  # (what a replan does: the previous step's iteration counts are in d_info)
            sol.order_by_work_device(N, d_info, d_order[0])
        if a.pipeline:
            src = d_x
            if d_all is not None:
                all_gather_plans()
                src = d_all
            sol.shift_traj_device(world * N, src, d_traj, z_2d=float(build["p0"][0][2]), shift=0)
            sol.generate_lsc_device(N, n_obs_eff, rank * N, d_traj, d_nbr, d_rad, d_dw, d_goal, d_rows)
            if d_xinit is not None:
                d_xwarm.copy_(d_x)  # the plans the rows were generated from are the primal start of the re-solve
        sol.solve_device(N, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info,
                         d_x_init=(d_xwarm if (a.pipeline and d_xinit is not None) else d_xinit), d_order=d_order[0])
        if d_all is not None and not a.pipeline:
            all_gather_plans()

    if a.graph:
        if world != 1:
            raise SystemExit("--graph is a single-GPU option")
        eager_step = step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm up on the capture stream, as graph capture requires
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph, stream=side):
            eager_step()
        step = hip_graph.replay  # noqa: F811

    # before the contract's W warm-up steps: untimed steps until the device has been busy for ~0.25 s.  A 20-step timed region of a 14 us
    # step lasts 0.3 ms -- on a box that has just been handed over the clocks are still ramping through all of it (round 5: the driver's
    # line was 14 % below the same command's line from a warm process).  Untimed, like the warm-up itself; `clock_warm_steps` says how many.
    if a.calibrate_counters:
        # known-byte streaming kernels at the row stream's access widths (csrc/lscqp_diag.hip: lscqp_calib), in THIS process: a rocprofv3 --pmc
        # session around this command measures them with the same counters as the solver's kernels (tools/profile_round.py)
        rc = api.lib().lscqp_debug_calibrate_(1 << 30, None)
        if rc != 0:
            raise SystemExit("bench.py: counter calibration failed: " + api.lib().lscqp_last_error().decode())
    clock_warm_steps = 0
    if not a.no_clock_warm:
        # (several ranks: a FIXED count -- a step may hold a collective, and ranks that warm by their own clocks would call it unequally often)
        t_w = time.perf_counter()
        while (clock_warm_steps < 300) if world > 1 else (time.perf_counter() - t_w < 0.25 and clock_warm_steps < 200000):
            for _ in range(50):
                step()
            clock_warm_steps += 50
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    phase_on = os.environ.get("LSCQP_ACTIVE_SET", "1")[:1] != "0" and os.environ.get("LSCQP_ACTIVE_SET_NOW", "1")[:1] != "0"
    if a.warmup > 0 and not a.graph and not a.no_work_order and not phase_on and N > sol.launch_capacity(N, n_obs_eff):
        # the work order a planner has from the previous replan (lscqp_plan carries it by itself): longest previous solve first, re-sorted
        # inside every step.  Only where a launch runs more than one round of workgroups (configs[3], the configs[4] shape): results are
        # bit-identical, smaller launches start every instance at once anyway.
        d_order[0] = torch.zeros(N, dtype=torch.int32, device=dev)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    ctx.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()  # (an event is created by its first record -- tens of microseconds of host time: not inside a 0.3 ms timed region)
    ev1.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ctx.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / a.steps  # average launch duration on the launch stream
    # the run's spread: the same K-step region six more times (barrier + synchronize on both sides each time; `value` stays the FIRST region's,
    # the contract's); max over ranks per repeat
    repeats = [elapsed / a.steps * 1e3]
    for _ in range(0 if a.no_spread else 6):
        ctx.barrier()
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ctx.barrier()
        repeats.append((time.perf_counter() - t_r) / a.steps * 1e3)
    repeats = [ctx.reduce([r], "max")[0] for r in repeats]

    if d_all is not None and not a.pipeline:
        # roofline.kernel_ms is the SOLVE kernel's launch duration: time it without the exchange that shares the step's stream
        ek0, ek1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ek0.record()
        for _ in range(20):
            solve_only()
        ek1.record()
        torch.cuda.synchronize()
        kernel_ms = ek0.elapsed_time(ek1) / 20
    my_elapsed = elapsed
    elapsed, kernel_ms = ctx.reduce([elapsed, kernel_ms], "max")

    if cold_calls is not None:
        solve_only()  # (the timed steps wrote the copies' own output buffers)
        torch.cuda.synchronize()
    status = d_st.cpu().numpy()
    info = d_info.cpu().numpy().view(api.INFO_DTYPE)
    iters = info["iterations"]
    paths = path_stats(api, info, status)  # (this rank's block)
    paths = {k: (ctx.reduce([float(v)], "max" if k.endswith("_max") else "sum")[0]) for k, v in paths.items()}
    for k in ("active_set_steps_mean", "interior_point_iters_mean"):  # (sums of per-rank means -> mean over ranks: blocks are equal up to one agent)
        paths[k] = paths[k] / max(ctx.world, 1)
    das_ms = active_set_kernel_ms(torch, api, dict(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max,
                                                   **{k_: v_ for k_, v_ in solver_kw.items() if k_ == "row_format"}), N, n_obs_eff,
                                  (d_hdr, d_rows, d_off, d_sfc, d_xinit))
    if das_ms is not None:
        das_ms, = ctx.reduce([das_ms], "max")
    n_bad = int(ctx.reduce([float((status != 0).sum())], "sum")[0])
    n_floor = int(ctx.reduce([float(((info["flags"] & api.INFO_FLOOR_ACCEPTED) != 0).sum())], "sum")[0])
    it_sum, = ctx.reduce([float(iters.sum())], "sum")
    it_max, = ctx.reduce([float(iters.max())], "max")
    n_ranks_seen, n_devices_seen, n_agents_seen = 1, 1, N
    rank_parity = None
    step_lat = None
    if dist is not None:
        # what actually ran: ranks, DISTINCT devices (by PCI bus id), agents solved per step over all ranks
        bus = torch.cuda.get_device_properties(ctx.dev_index)
        dev_id = hash((getattr(bus, "pci_bus_id", ctx.dev_index), getattr(bus, "pci_device_id", 0), getattr(bus, "pci_domain_id", 0), ctx.dev_index)) % (1 << 40)
        ids = [None] * world
        dist.all_gather_object(ids, (dev_id, N, my_elapsed))
        n_ranks_seen, n_devices_seen, n_agents_seen = len(ids), len({i[0] for i in ids}), sum(i[1] for i in ids)
        if d_all is not None and not a.pipeline:
            # the exchange must have delivered every owner's block: compare this rank's view with its own block
            mine = d_all.view(world, n_pad * nv)[rank][: N * nv]
            if not torch.equal(mine, d_x[: N * nv]):
                raise SystemExit("bench.py: rank %d: the all-gathered plans do not contain this rank's block" % rank)
        if not a.no_rank_parity and not a.no_cpu_baseline and a.rows == "f64":
            # every rank checks its OWN block against the oracle (a bounded sample: the oracle is a dense CPU solver)
            from oracle import oracle as O

            k = min(N, 16 if M < 10 else 6)
            Rk, sel = oracle_sample(O, sw, build, M, dim, n_obs_eff, sample=k)
            xg, og = d_x.cpu().numpy()[: N * nv].reshape(N, nv)[sel], d_obj.cpu().numpy()[sel]
            ok = (Rk["status"] == 0) & (status[sel] == 0)
            dxm = float(np.abs(xg - Rk["x"])[ok].max()) if ok.any() else float("inf")
            dom = float((np.abs(og - Rk["obj"]) / np.maximum(1.0, np.abs(Rk["obj"])))[ok].max()) if ok.any() else float("inf")
            dxm, dom = ctx.reduce([dxm, dom], "max")
            rank_parity = {"max_abs_dx": dxm, "max_rel_dobj": dom, "compared_per_rank": int(k), "ranks": world,
                           "compared": int(ctx.reduce([float(ok.sum())], "sum")[0])}
        if strong and not a.no_latency:
            # per-step latency of the sharded step (solve of the block + the all-gather), every rank taking part: max over ranks
            ls = []
            for _ in range(260):
                t_a = time.perf_counter()
                step()
                torch.cuda.synchronize()
                ls.append(time.perf_counter() - t_a)
            ls = np.array(ls[10:]) * 1e3
            p50, p99 = ctx.reduce([float(np.percentile(ls, 50)), float(np.percentile(ls, 99))], "max")
            step_lat = {"p50": p50, "p99": p99, "calls": int(len(ls)), "what": "solve of the rank's block + all-gather, enqueue -> readable, max over ranks"}
    one_gpu = None
    if whole is not None and not a.no_one_gpu_reference:
        # the SAME global batch solved whole by rank 0's GPU alone (the other ranks wait at the barrier): the one-GPU figure of this very
        # workload, so that the line carries its own strong-scaling reference
        w_h, w_o, w_s = (to_dev(torch, x, dev) for x in (whole.hdr, whole.off, whole.sfc))
        w_r = to_dev(torch, sol.rows_in_format(whole.rows), dev)
        w_xi = None if a.cold_start else torch.from_numpy(api.x_init_from_swarm(whole.build, dim)).to(dev)
        w_x, w_ob = torch.zeros(n_glob * nv, dtype=torch.float64, device=dev), torch.zeros(n_glob, dtype=torch.float64, device=dev)
        w_st, w_in = torch.zeros(n_glob, dtype=torch.int32, device=dev), torch.zeros(n_glob * 32, dtype=torch.uint8, device=dev)
        reps1 = max(5, min(a.steps, 20))
        w_ord = torch.zeros(n_glob, dtype=torch.int32, device=dev) if (n_glob > sol.launch_capacity(n_glob, n_obs_eff) and not a.no_work_order) else None

        def whole():  # the same rule as the sharded steps: sorted by the previous solve's counts where the launch can have a tail
            if w_ord is not None:
                sol.order_by_work_device(n_glob, w_in, w_ord)
            sol.solve_device(n_glob, n_obs_eff, w_h, w_r, w_o, w_s, w_x, w_ob, w_st, w_in, d_x_init=w_xi, d_order=w_ord)

        sol.solve_device(n_glob, n_obs_eff, w_h, w_r, w_o, w_s, w_x, w_ob, w_st, w_in, d_x_init=w_xi)
        for i_ in range(2):
            whole()
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for _ in range(reps1):
            whole()
        torch.cuda.synchronize()
        t_1 = (time.perf_counter() - t_a) / reps1
        same = bool(torch.equal(w_x[lo * nv: hi * nv], d_x[: N * nv]))
        one_gpu = {"qp_per_s": n_glob / t_1, "ms_per_step": t_1 * 1e3, "steps": reps1,
                   "what": "the whole %d-agent batch on rank 0's GPU alone, no collective (wall clock, synchronised)" % n_glob,
                   "rank0_block_bit_identical_to_sharded_solve": same}
        del w_h, w_o, w_s, w_r, w_xi, w_x, w_ob, w_st, w_in, w_ord
    ctx.barrier()
    return SimpleNamespace(a=a, api=api, sw=sw, sol=sol, build=build, hdr=hdr, rows=rows, off=off, sfc=sfc, N=N, M=M, dim=dim, nv=nv, n_obs_eff=n_obs_eff,
                           n_glob=n_glob, n_pad=n_pad, strong=strong, d_hdr=d_hdr, d_rows=d_rows, d_off=d_off, d_sfc=d_sfc, d_xinit=d_xinit, d_x=d_x,
                           d_obj=d_obj, d_st=d_st, d_info=d_info, d_all=d_all, elapsed=elapsed, kernel_ms=kernel_ms, status=status, iters=iters,
                           n_bad=n_bad, n_floor=n_floor, iters_mean=it_sum / max(n_agents_seen, 1), iters_max=int(it_max), n_ranks_seen=n_ranks_seen,
                           n_devices_seen=n_devices_seen, n_agents_seen=n_agents_seen, rank_parity=rank_parity, step_lat=step_lat, one_gpu=one_gpu,
                           solve_only=solve_only, ordered=d_order[0] is not None, paths=paths, das_ms=das_ms, repeats=repeats, clock_warm_steps=clock_warm_steps,
                           solver_kw=solver_kw)


def check_ran_as_asked(ctx, a, S):
    if S.n_ranks_seen != a.gpus or S.n_agents_seen != S.n_glob or (ctx.world > 1 and ctx.backend == "nccl" and S.n_devices_seen != ctx.world):
        raise SystemExit("bench.py: asked for %d GPUs and %d agents per step; ran %d ranks on %d distinct devices solving %d agents" % (
            a.gpus, S.n_glob, S.n_ranks_seen, S.n_devices_seen, S.n_agents_seen))


def workload_config(ctx, a, S):
    """The `config` object of a line: names the workload and what ran."""
    cfg0 = CONFIGS[a.config]
    gathered = S.d_all is not None
    c = {
        "workload": "%s -- %s x M=%d segments x %d LSC neighbours (dim=%d, %s swarm after 3 warm-up replans), "
                    "%s: dual active-set phase (one workgroup per QP) with the batched PDIP behind it for what it leaves" % (
                        cfg0["what"] if (a.agents, S.M, a.obs, S.dim) == (cfg0["agents"], cfg0["segments"], cfg0["obs"], cfg0["dim"]) else "custom shape",
                        ("ONE batch of %d agents cut into blocks of <= %d per GPU" % (S.n_glob, S.n_pad)) if S.strong else ("%d agents/GPU" % S.N),
                        S.M, S.n_obs_eff, S.dim, a.style, "fp64" if a.precision == "f64" else "mixed-precision"),
        "baseline_config": a.config, "batch_seed": CONFIGS[a.config]["seed"], "precision": a.precision, "row_format": a.rows,
        "collective_backend": ctx.coll_backend, "agents_per_gpu": S.n_pad, "agents_total": S.n_glob, "agents_solved_per_step_all_ranks": S.n_agents_seen,
        "ranks": S.n_ranks_seen, "distinct_devices": S.n_devices_seen, "launch": "one process per GPU (torch.distributed)" if ctx.world > 1 else "single process",
        "segments": S.M, "lsc_neighbours": S.n_obs_eff, "dim": S.dim,
        "rows_per_qp": S.sol.num_inequalities(S.n_obs_eff), "allgather": bool(gathered), "pipeline": bool(a.pipeline), "hip_graph": bool(a.graph),
        "warm_start": "initial_traj (shifted previous plan) as primal start" if S.d_xinit is not None else "none",
        "work_order": ("longest previous solve first (lscqp_order_by_work_device on the previous step's iteration counts, inside every step)"
                       if S.ordered else "as given (the launch starts every instance at once)"),
        "parallelism": ("agents sharded over %d GPU(s), " % ctx.world) +
                       ("one RCCL all-gather of the plans per step" if gathered else "no data-path collective"),
    }
    if S.strong:
        # north_star: "RCCL all-gather ... only when agent count justifies it" -- the library's stated rule (lscqp_comm_devices_for:
        # clamp(n / 256, 1, G)) next to what this run was TOLD to use
        c["devices_by_crossover_rule"] = int(max(1, min(ctx.world, S.n_glob // 256)))
        c["allgather_bytes_per_rank_per_step"] = int(S.n_pad * S.nv * 8)
        c["allgather_bytes_received_per_rank_per_step"] = int(ctx.world * S.n_pad * S.nv * 8)
    if S.one_gpu is not None:
        c["one_gpu_same_workload"] = S.one_gpu
    return c


def workload_roofline(ctx, a, S):
    """The dominant kernel of the step.  Round 5: the dual active-set kernel (lscqp_das::das_kernel) finishes most -- on BASELINE's batches
    all -- instances and is where the step's time goes; `kernel_ms` is ITS average launch duration (the phase alone, 20 launches back to
    back between HIP events on the launch stream), `step_gpu_ms` the whole solve call's (phase + the interior-point pass behind it,
    events around the timed steps).  With the phase switched off (LSCQP_ACTIVE_SET=0) the kernel is the interior-point instance, as in
    rounds 1-4."""
    bq = S.sol.algorithmic_bytes(S.n_obs_eff)
    pd = "lscqp_pdip_kernel<%d,%d,true,NSLOT,W,%s>" % (S.M, S.dim, "float" if a.precision == "mixed" else "double")
    if S.das_ms is not None and S.paths["active_set_solved"] >= 0.5 * S.n_agents_seen:
        n_cu = ctx.torch.cuda.get_device_properties(ctx.dev_index).multi_processor_count
        kms, kname = S.das_ms, "lscqp_das::das_kernel<%d,%s,false,%s>" % (4 if S.N <= 8 * n_cu else 1, "true" if a.rows == "f32" else "false",
                                                                                  "false" if S.N <= 2 * n_cu else "true")  # (wavefronts, row format, lean form, first look peeled)
    else:
        kms, kname = S.kernel_ms, pd
    achieved = bq * S.N / (kms * 1e-3)
    return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None,
            "kernel": kname, "kernel_ms": kms, "step_gpu_ms": S.kernel_ms, "behind_it": pd,
            "algorithmic_bytes_per_qp": bq, "qps_per_launch": S.N}


def compact_block(ctx, a, S):
    """One of the additional workloads of a multi-GPU line: the same quantities as the headline, without the single-GPU legs."""
    check_ran_as_asked(ctx, a, S)
    r = workload_roofline(ctx, a, S)
    c = workload_config(ctx, a, S)
    out = {"baseline_config": a.config, "scaling": a.scaling, "value": S.n_glob * a.steps / S.elapsed, "unit": "QP/s", "ms_per_step": S.elapsed / a.steps * 1e3,
           "steps": a.steps, "warmup": a.warmup, "agents_total": S.n_glob, "agents_per_gpu": S.n_pad, "allgather": c["allgather"],
           "collective_backend": ctx.coll_backend, "distinct_devices": S.n_devices_seen, "workload": c["workload"],
           "hbm_frac_whole_job": r["algorithmic_bytes_per_qp"] * S.n_glob * a.steps / S.elapsed / (ctx.world * HBM_PEAK),
           "kernel_ms": S.kernel_ms, "active_set_kernel_ms": S.das_ms, "non_optimal": S.n_bad, "floor_accepted": S.n_floor, "iters_mean": S.iters_mean,
           "iters_max": S.iters_max, "paths": S.paths}
    for k in ("allgather_bytes_per_rank_per_step", "devices_by_crossover_rule", "one_gpu_same_workload"):
        if k in c:
            out[k] = c[k]
    if S.rank_parity is not None:
        out["parity_all_ranks"] = S.rank_parity
    if S.step_lat is not None:
        out["latency_ms"] = {"sharded_step": S.step_lat}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="BASELINE config that is the timed workload.  Default: c1 (the headline, configs[1]) on one GPU; with --gpus N > 1 the "
                         "default is what BASELINE's multi-GPU configs name: c3 (1024 agents x M10 x 40) as ONE batch sharded over the ranks "
                         "with the RCCL all-gather of the plans every step, followed by the c4-shape, c2 and weak-c1 blocks")
    ap.add_argument("--agents", type=int, default=None, help="agents per GPU per step (overrides the config's)")
    ap.add_argument("--segments", type=int, default=None)
    ap.add_argument("--obs", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--style", default=None)
    ap.add_argument("--precision", default=None, choices=["f64", "mixed"])
    ap.add_argument("--rows", default=None, choices=["f64", "f32"])
    ap.add_argument("--allgather", action="store_true", help="all-gather solved trajectories over RCCL every step")
    ap.add_argument("--pipeline", action="store_true",
                    help="step = all-gather of the plans (N > 1) -> LSC generation on the device -> QP solve (SURVEY 8f-1); "
                         "the default step is the QP solve of BASELINE's metric alone")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (all its kernel launches) in a HIP graph and replay it: removes the launch gaps of the "
                         "multi-kernel --pipeline step (single GPU only; no *_device entry point synchronises or allocates)")
    ap.add_argument("--cold-start", action="store_true", help="do not hand the initial trajectories to the solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional blocks (one GPU: every BASELINE config at its own shape; "
                                                            "N > 1: the c4-shape / c2 / weak-c1 workloads after the headline)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the latency loops (their launches would mix into a kernel trace of the timed steps)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: every rank owns --agents agents of its own swarm, no data-path collective.  strong: the config's agents are "
                         "ONE global batch cut into contiguous blocks of ceil(N/G) (BASELINE configs[2..4]), the plans all-gathered every "
                         "step.  Default: weak with an explicit --config or one GPU; strong for a bare --gpus N > 1")
    ap.add_argument("--single-process", action="store_true",
                    help="drive the --gpus devices from THIS process through lscqp_comm (ncclCommInitAll; one stream per device) "
                         "instead of one process per GPU; implies --scaling strong")
    ap.add_argument("--no-rank-parity", action="store_true", help="multi-rank runs: skip the per-rank oracle check of each rank's own block")
    ap.add_argument("--no-work-order", action="store_true",
                    help="launch the instances in the order given instead of longest-previous-solve first (lscqp_order_by_work_device)")
    ap.add_argument("--no-clock-warm", action="store_true", help="skip the untimed ~0.25 s of steps in front of the warm-up (see timed_workload)")
    ap.add_argument("--no-spread", action="store_true", help="skip the six extra timed repeats of the K-step region (the line's `spread`)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-HBM measurement (512 MiB of batch copies solved round robin)")
    ap.add_argument("--cold", action="store_true", help="profiling aid: the timed steps rotate over >= 512 MiB of copies of the batch (every launch reads HBM)")
    ap.add_argument("--calibrate-counters", action="store_true",
                    help="profiling aid: launch the known-byte streaming kernels (lscqp_calib) once before the warm-up, for a rocprofv3 --pmc session to see")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="strong scaling: skip rank 0's solve of the WHOLE batch on its own GPU (config.one_gpu_same_workload)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    # what a bare `--gpus N` measures (north_star / BASELINE configs[2..4]): N = 1 the configs[1] headline; N > 1 the sharded
    # configs[3] batch + all-gather as the line's value, then the other multi-GPU configs and the weak-c1 block
    multi_default = args.gpus > 1 and args.config is None and args.scaling is None and not (args.single_process or args.pipeline or args.graph)
    if args.config is None:
        args.config = "c3" if multi_default else ("c2" if args.single_process else "c1")
    if args.single_process:
        args.scaling = "strong"
    if args.scaling is None:
        args.scaling = "strong" if multi_default else "weak"
    if args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ:
        # started bare with --gpus N: launch one rank per GPU ourselves, exactly as the driver would
        raise SystemExit(launch_ranks(args.gpus))
    cfg0 = CONFIGS[args.config]
    for k in ("agents", "segments", "obs", "dim", "style", "precision", "rows"):
        if getattr(args, k) is None:
            setattr(args, k, cfg0[k])

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.single_process:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the flag and the launch must agree (start it bare and it launches "
                         "its own ranks, or under torch.distributed.run with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    # one process per GPU.  (LSCQP_BENCH_BACKEND=gloo lets a 1-GPU box exercise the multi-rank code path: the ranks then
    # share the device and the collectives go through gloo; never used for reported numbers.)
    backend = os.environ.get("LSCQP_BENCH_BACKEND", "nccl")
    if args.single_process:
        if world != 1:
            raise SystemExit("bench.py: --single-process drives all devices from one process; do not start it under a launcher")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit("bench.py: --single-process --gpus %d needs %d devices, %d visible" % (args.gpus, args.gpus, torch.cuda.device_count()))
        return single_process_main(args, torch)
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks need %d distinct devices, %d visible (LSCQP_BENCH_BACKEND=gloo shares devices, for "
                         "tests only)" % (world, world, torch.cuda.device_count()))
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    coll_backend = "none (single process)"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
            try:  # RCCL carries the barrier, the timing reduction and the all-gather of the plans
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:  # RCCL unusable on this node: keep going over gloo, and SAY so in the line
                sys.stderr.write("bench.py: RCCL probe failed (%s); falling back to gloo for barrier / reductions / all-gather\n" % e)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
                coll_backend = "gloo (RCCL probe failed), %d ranks" % dist.get_world_size()
            else:
                coll_backend = "rccl via torch.distributed nccl backend, %d ranks" % dist.get_world_size()
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            coll_backend = "%s (LSCQP_BENCH_BACKEND), %d ranks" % (backend, dist.get_world_size())

    from lsc_dr_planner_amd import api, synth

    ctx = Ctx(torch, dist, rank, world, dev_index, backend, coll_backend)
    S = timed_workload(ctx, args)
    # the other workloads of a bare multi-GPU run: every rank takes part, rank 0 keeps the blocks
    more = []
    if multi_default and not args.no_extra:
        for key, scal in (("c4_f64", "strong"), ("c2", "strong"), ("c1", "weak")):
            a2 = workload_args(args, key, scaling=scal, allgather=False, no_latency=True)
            try:
                more.append(compact_block(ctx, a2, timed_workload(ctx, a2)))
            except Exception as ex:  # (a collective failure takes every rank here; a SystemExit ends the job, as it must)
                more.append({"baseline_config": key, "scaling": scal, "error": "%s: %s" % (type(ex).__name__, str(ex)[:300])})
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    check_ran_as_asked(ctx, args, S)

    sol, N, M, dim, nv, n_obs_eff, n_glob = S.sol, S.N, S.M, S.dim, S.nv, S.n_obs_eff, S.n_glob
    d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_xinit = S.d_hdr, S.d_rows, S.d_off, S.d_sfc, S.d_x, S.d_obj, S.d_st, S.d_info, S.d_xinit
    hdr, rows, off, sfc, build, sw, status, iters, kernel_ms, elapsed = S.hdr, S.rows, S.off, S.sfc, S.build, S.sw, S.status, S.iters, S.kernel_ms, S.elapsed

    # ---- single-launch latency distribution (enqueue -> results readable), device-resident inputs ----------
    n_lat, n_lath = (60, 60) if args.no_latency else (1050, 250)
    lat = []
    for _ in range(n_lat):  # p50 / p99 over >= 1000 timed calls
        t_a = time.perf_counter()
        S.solve_only()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t_a)
    lat = np.array(lat[50:]) * 1e3
    # batch-of-1 latency (the unchanged sequential simulator loop, src/multi_sync_simulator.cpp:357-362)
    lat1 = []
    for _ in range(n_lat if not args.no_latency else 0):
        t_a = time.perf_counter()
        sol.solve_device(1, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_x_init=d_xinit)
        torch.cuda.synchronize()
        lat1.append(time.perf_counter() - t_a)
    lat1 = np.array(lat1[50:] or [float('nan')]) * 1e3
    # the host-pointer entry (lscqp_solve_batch: H2D of the inputs, solve, D2H of the results) -- PCIe-inclusive, never `value`
    lath = []
    x0_host = None if d_xinit is None else d_xinit.cpu().numpy().reshape(N, nv)
    for _ in range(n_lath if not args.no_latency else 0):
        t_a = time.perf_counter()
        sol.solve_host(hdr, rows, off, sfc, want_info=False, x_init=x0_host)
        lath.append(time.perf_counter() - t_a)
    lath = np.array(lath[50:] or [float('nan')]) * 1e3
    # ... and for a single QP: what one TrajOptimizer::solve call of the unchanged sequential planner loop costs end to end
    lath1 = []
    for _ in range(n_lath if not args.no_latency else 0):
        t_a = time.perf_counter()
        sol.solve_host(hdr[:1], rows.reshape(N, -1)[0], off[:2], sfc.reshape(N, M)[:1], want_info=False,
                       x_init=None if x0_host is None else x0_host[:1])
        lath1.append(time.perf_counter() - t_a)
    lath1 = np.array(lath1[50:] or [float('nan')]) * 1e3
    S.solve_only()
    torch.cuda.synchronize()

    roof = workload_roofline(ctx, args, S)
    # HBM-side traffic per launch: PMC counters cannot be read from inside this process, so the value comes from the
    # committed rocprofv3 --pmc passes of THIS command (profiles/<tag>_<config>_pmc.json of the newest tag, numerically
    # sorted; tools/profile_round.py) and is only reported when kernel shape, precision and batch size match; otherwise null.
    traffic, traffic_src, valu, pm, pf = None, None, None, {}, None
    try:
        pf = newest_profile("*_%s_pmc.json" % args.config)
        pm = json.load(open(pf)) if pf else {}
        kname = (pm.get("kernel") or "").replace(" ", "")
        want_t = "float" if args.precision == "mixed" else "double"
        is_das = "das_kernel" in kname and "das_kernel" in roof["kernel"] and pm.get("segments", M) == M and pm.get("dim", dim) == dim
        is_pdip = ("lscqp_pdip_kernel<%d,%d,true" % (M, dim)) in kname and want_t in kname and "pdip" in roof["kernel"]
        if pm.get("qps_per_launch") == N and (is_das or is_pdip) and pm.get("lsc_neighbours") == n_obs_eff:
            traffic = pm["traffic_bytes_per_launch"]
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes)" % os.path.basename(pf)
            sq = pm.get("sq") or {}
            if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_WAVES"):
                # the limit this kernel actually runs into (SURVEY.md 8d asks for it next to the HBM figure): the share of a
                # wavefront's lifetime in which a vector-ALU instruction of it is executing, from the same PMC passes
                valu = {"valu_active_frac_of_wave_lifetime": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
                        "valu_instructions_per_wavefront": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
                        "lds_instructions_per_wavefront": sq["SQ_INSTS_LDS"] / sq["SQ_WAVES"],
                        "wavefronts_per_launch": sq["SQ_WAVES"], "dispatch": pm.get("dispatch"),
                        "source": "profiles/%s (rocprofv3 --pmc SQ_*)" % os.path.basename(pf)}
    except Exception:
        pass
    roof.update({"traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src, "valu": valu})
    # the same kernel with nothing of its batch in a cache (cold_measure): `frac` above is the batch re-solved in place
    roof["frac_hot"] = roof["frac"]
    roof["frac_cold"], roof["cold"] = None, None
    if world == 1 and not args.no_cold and "das_kernel" in roof["kernel"] and not (args.pipeline or args.graph):
        try:
            cm = cold_measure(torch, api, sol, dict(M=M, dim=dim, world_min=sw.world_min, world_max=sw.world_max,
                                                    **{k_: v_ for k_, v_ in S.solver_kw.items() if k_ == "row_format"}), N, n_obs_eff,
                              (d_hdr, d_rows, d_off, d_sfc, d_xinit))
            roof["frac_cold"], roof["cold"] = cm["frac"], cm
        except Exception as ex:  # noqa: BLE001
            roof["cold"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    info_h = d_info.cpu().numpy().view(api.INFO_DTYPE)
    rv = (valu_roofline(sol, N, n_obs_eff, iters[((info_h["flags"] & api.INFO_ACTIVE_SET) == 0) & (status == 0)], kernel_ms)
          if args.precision == "f64" else None)
    if rv:  # the roof that binds, inside the object the driver keeps (the full record is `roofline_valu` below)
        roof["fp64_valu"] = {"achieved_TFLOPs": rv["fp64_flops_per_s"] / 1e12, "peak_TFLOPs": FP64_VECTOR_PEAK / 1e12, "frac": rv["frac_of_78.6e12"],
                             "flops_per_launch": rv["fp64_flops_per_launch"], "what": "fp64 vector flops of this launch (iterations of this run x "
                             "machine-code count of the instance) / kernel_ms, against the fp64 vector peak"}
    latency = {"batch_p50": float(np.percentile(lat, 50)), "batch_p99": float(np.percentile(lat, 99)),
               "single_qp_p50": float(np.percentile(lat1, 50)), "single_qp_p99": float(np.percentile(lat1, 99)),
               "host_pointers_batch_p50": float(np.percentile(lath, 50)), "host_pointers_batch_p99": float(np.percentile(lath, 99)),
               "host_pointers_single_qp_p50": float(np.percentile(lath1, 50)), "host_pointers_single_qp_p99": float(np.percentile(lath1, 99)),
               "samples": {"device_resident": int(len(lat)), "single_qp": int(len(lat1)), "host_pointers": int(len(lath))}}
    if S.step_lat is not None:
        latency["sharded_step"] = S.step_lat
    conf = workload_config(ctx, args, S)
    # BASELINE's metric is "QP solves/sec + p99 solve latency": the second half travels inside `config` (an object the driver keeps)
    conf["latency_ms"] = ({"p50": S.step_lat["p50"], "p99": S.step_lat["p99"], "calls": S.step_lat["calls"], "of": "one sharded step: solve of the rank's block + "
                           "all-gather, enqueue -> readable, max over ranks"} if S.step_lat is not None else
                          {"p50": latency["batch_p50"], "p99": latency["batch_p99"], "calls": int(len(lat)),
                           "of": "one %d-QP launch, enqueue -> results readable, device-resident inputs" % N})
    out = {
        "metric": "qp_solves_per_sec",
        "value": n_glob * args.steps / elapsed,
        "unit": "QP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64" if args.precision == "f64" else "f64 iterate and residuals, f32 factorisation",
        "data": "synthetic",
        "spread": {"ms_per_step": {"min": float(min(S.repeats)), "median": float(np.median(S.repeats)), "max": float(max(S.repeats))},
                   "value": {"min": n_glob / (max(S.repeats) * 1e-3), "median": n_glob / (float(np.median(S.repeats)) * 1e-3), "max": n_glob / (min(S.repeats) * 1e-3)},
                   "repeats": len(S.repeats), "what": "the timed K-step region repeated (the first repeat IS `value` / `ms_per_step`); wall clock, synchronised, max over ranks",
                   "clock_warm_steps_before_warmup": S.clock_warm_steps},
        "config": conf,
        "roofline": roof,
        "roofline_valu": rv,
        "latency_ms": latency,
        "solver": {"non_optimal": S.n_bad, "floor_accepted": S.n_floor, "iters_mean": S.iters_mean, "iters_max": S.iters_max, "paths": S.paths,
                   "what": "iters_*: lscqp_info.iterations over the batch = active-set steps for instances the dual active-set phase finished "
                           "(LSCQP_INFO_ACTIVE_SET), interior-point iterations for the others; `paths` tells them apart"},
    }
    if world == 1 and args.precision == "f64" and S.das_ms is not None and not (args.pipeline or args.graph):
        # the kernel BEHIND the phase, measured by this run too (the interior-point kernel north_star names: with the phase on it has nothing to
        # do on this batch): the same step with the phase switched off -- rounds 1-4's solver -- and its own roof, fp64 vector work
        sol.set_knob("active_set_off", 1)
        try:
            off_call = sol.bind_device(N, n_obs_eff, d_hdr, d_rows, d_off, d_sfc, d_x, d_obj, d_st, d_info, d_x_init=d_xinit)
            ms_off = time_calls(torch, [off_call] * 20)
            io = d_info.cpu().numpy().view(api.INFO_DTYPE)
            st_off = d_st.cpu().numpy()
            out["phase_off"] = {"ms_per_step": ms_off, "value": N / (ms_off * 1e-3), "unit": "QP/s", "speedup_of_phase": ms_off / out["ms_per_step"],
                                "kernel": roof["behind_it"], "non_optimal": int((st_off != 0).sum()), "iters_mean": float(io["iterations"].mean()),
                                "iters_max": int(io["iterations"].max()), "fp64_valu": valu_roofline(sol, N, n_obs_eff, io["iterations"][st_off == 0], ms_off),
                                "hbm_frac": roof["algorithmic_bytes_per_qp"] * N / (ms_off * 1e-3) / HBM_PEAK,
                                "what": "the same batch, the same call, the dual active-set phase switched off: the interior-point kernel alone (HIP events, 20 launches)"}
        finally:
            sol.set_knob("active_set_off", 0)
        S.solve_only()
        torch.cuda.synchronize()
    if world > 1:
        out["roofline"]["whole_job"] = {"achieved": roof["algorithmic_bytes_per_qp"] * out["value"] / 1e9, "peak": world * HBM_PEAK / 1e9, "unit": "GB/s",
                                        "frac": roof["algorithmic_bytes_per_qp"] * out["value"] / (world * HBM_PEAK),
                                        "what": "QP/s x algorithmic bytes per QP over n_gpus x 8 TB/s (SURVEY.md 8d)"}
    try:  # hardware cross-check of the machine-code count, from the committed PMC pass of this command (labelled as such)
        if out.get("roofline_valu") and pm.get("f64_flops_per_launch_pmc") and pm.get("qps_per_launch") == N:
            out["roofline_valu"]["pmc_cross_check"] = {
                "fp64_flops_per_launch_pmc": pm["f64_flops_per_launch_pmc"], "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64)" % os.path.basename(pf),
                "machine_code_count_over_pmc": out["roofline_valu"]["fp64_flops_per_launch"] / pm["f64_flops_per_launch_pmc"],
                "note": "valid while the iteration counts of this run equal the profiled run's (same seeds: they do)"}
    except Exception:
        pass
    if S.rank_parity is not None:
        out["parity_all_ranks"] = S.rank_parity
    if more:
        out["other_workloads"] = more

    # ---- CPU baseline: the oracle port on the same batch, all host cores --------------------------------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O

        cpu, R, sel = oracle_baseline(O, sw, build, M, dim, n_obs_eff, sample=N if N <= 64 else (64 if M < 10 else 32), budget_s=3.0, single_core=True)
        cpu["reference_published"] = ("CPLEX 20.1, 6 threads: 4.58-6.64 ms/QP (151-218 QP/s) at M=10 dim=2 <=9 neighbours "
                                      "(reference log/summary_LSC_10agents.csv)")
        out["cpu_baseline"] = cpu
        xg = d_x.cpu().numpy()[: N * nv].reshape(N, nv)[sel]
        og = d_obj.cpu().numpy()[sel]
        ok = (R["status"] == 0) & (status[sel] == 0)
        out["parity"] = {
            "max_abs_dx": float(np.abs(xg - R["x"])[ok].max()),
            "max_rel_dobj": float((np.abs(og - R["obj"]) / np.maximum(1.0, np.abs(R["obj"])))[ok].max()),
            "compared": int(ok.sum()),
        }

    # ---- every BASELINE config at its own shape, on this GPU (the headline above stays configs[1]) ---------------------
    if world == 1 and not args.no_extra:
        O = None
        if not args.no_cpu_baseline:
            from oracle import oracle as O  # the checker / CPU baseline, never the thing measured
        sweep = []
        for key in ("c0", "c2", "c3s", "c3", "c4_f64", "c4", "c1_loaded", "c0_loaded", "c2_loaded", "c4_loaded", "c1_infeasible_1pct", "c4_infeasible_1pct"):
            try:
                extra = key not in ("c0", "c2", "c3s", "c3", "c4_f64", "c4")
                sweep.append(measure_config(torch, api, synth, dev, key, CONFIGS[key], O=(O if not key.endswith("_1pct") or True else None),
                                            lat_seconds=1.0 if (args.no_latency or extra) else 3.0,
                                            cold=(key in ("c2", "c3", "c4_f64", "c4", "c4_loaded") and not args.no_cold)))
            except Exception as ex:  # one failing shape must not take the headline line with it
                sweep.append({"config": key, "error": "%s: %s" % (type(ex).__name__, str(ex)[:300])})
        out["configs"] = sweep
        by = {c.get("config"): c for c in sweep}
        if "kernel_ms" in by.get("c4", {}) and "kernel_ms" in by.get("c4_f64", {}):
            out["mixed_vs_fp64_at_4096"] = {"fp64_qp_per_s": by["c4_f64"]["qp_per_s"], "mixed_qp_per_s": by["c4"]["qp_per_s"],
                                            "ratio": by["c4"]["qp_per_s"] / by["c4_f64"]["qp_per_s"],
                                            "iters_fp64": by["c4_f64"]["iters_mean"], "iters_mixed": by["c4"]["iters_mean"]}
        # the run-time-shaped kernel (csrc/lscqp_generic.hip: shapes / neighbour counts no compiled instance serves): informational --
        # the headline's class forced onto it, a 64-neighbour batch no compiled instance holds, and a shape that has none (M = 9, DLSC)
        try:
            gk = {}
            os.environ["LSCQP_FORCE_GENERIC"] = "1"
            t = measure_config(torch, api, synth, dev, "c1", dict(CONFIGS["c1"], what="configs[1] shape forced onto the run-time-shaped kernel"), O=None, lat_seconds=0.5, phase_off=False)
            gk["c1_shape_forced"] = {k: t[k] for k in ("kernel_ms", "qp_per_s", "iters_mean", "iters_max", "non_optimal")}
            os.environ.pop("LSCQP_FORCE_GENERIC")
            t = measure_config(torch, api, synth, dev, "n64", dict(agents=128, segments=5, obs=64, dim=3, style="forest", precision="f64", rows="f64", seed=3133,
                                                                  what="128 agents x M=5 x 64 LSC neighbours (beyond every compiled instance)"), O=None, lat_seconds=0.5, phase_off=False)
            gk["m5_64_neighbours"] = {k: t[k] for k in ("kernel_ms", "qp_per_s", "iters_mean", "iters_max", "non_optimal", "lsc_neighbours")}
            gk["compiled_instance_at_c1"] = by.get("c1", {}).get("kernel_ms") or out["roofline"]["kernel_ms"]
            out["generic_kernel"] = gk
        except Exception as ex:
            out["generic_kernel"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        finally:
            os.environ.pop("LSCQP_FORCE_GENERIC", None)

        try:
            out["replan_chain"] = replan_chain(torch, api)
        except Exception as ex:
            out["replan_chain"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}

    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

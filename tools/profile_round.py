"""Collect the profile artefacts of one round ON THE GPU BOX and write them under gpurun_out/profiles_<tag>/.

    python tools/profile_round.py r02_v1 [c1 c4_f64 c4 c3s ...]      (default: every BASELINE config of bench.py)

Per config (bench.py --config <c>: that config is then the timed workload, nothing else launches the QP kernel):
  1. rocprofv3 --kernel-trace --stats -- python bench.py --config <c> ...   -> <tag>_<c>_kernel_stats.csv (+ the bench line)
  2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate passes, never together with a trace)
                                                                             -> <tag>_<c>_pmc.json
and once: the plain `python bench.py` line (-> <tag>_bench.json) and the kernels either side of the QP.
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KB and need
separate passes; on gfx950 wide coalesced reads are tallied at a fraction of their bytes and "other access widths and WRITE_SIZE are
uncalibrated: calibrate on a known byte count in your own access pattern" -- round 6: the memory-side passes run the COLD form of the
bench command (>= 512 MiB of batch copies: every launch reads HBM, "scale past L3 before reading FETCH_SIZE") together with known-byte
streaming kernels at the row stream's access widths (csrc/lscqp_diag.hip: lscqp_calib) in the SAME session; ONE rule for every config.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
which = sys.argv[2:] or ["c1", "c4_f64", "c4", "c3s", "c3", "c2", "c0"]
OUT = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
STEPS = {"c1": 400, "c0": 200, "c2": 100, "c3s": 60, "c3": 20, "c4": 60, "c4_f64": 60, "c4_loaded": 60, "c1_loaded": 400, "c2_loaded": 100, "c0_loaded": 200}


def bench_cmd(c, cold=False, calibrate=False):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--steps", str(STEPS.get(c, 50)), "--warmup", "10",
            "--no-cpu-baseline", "--no-extra", "--no-latency", "--no-spread", "--no-cold", "--no-clock-warm"] + (  # no side loops: the trace averages the timed launches alone
        ["--cold"] if cold else []) + (["--calibrate-counters"] if calibrate else [])


def run(cmd, log):
    with open(os.path.join(OUT, log), "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT).returncode


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except Exception:
                pass
    return None


CALIB_BYTES = float(1 << 30)  # bench.py --calibrate-counters: lscqp_debug_calibrate_(1 << 30): every calibration launch moves exactly this


def is_solver(name):
    return "lscqp_pdip_kernel" in name or "das_kernel" in name


def profile_config(c):
    from bench import CONFIGS

    N = CONFIGS[c]["agents"]

    def timed_shape(name, grid, wg):
        """a launch over the whole batch: one workgroup per instance (grid = N workgroups), or the persistent form of the kernel (its last
        template argument is `true`; the grid is then what the chip holds)"""
        return grid == N * wg or name.replace(" ", "").split(">(")[0].endswith(",true")

    res = {"config": c, "what": CONFIGS[c]["what"], "batch_seed": CONFIGS[c]["seed"], "command": " ".join(["python", "bench.py"] + bench_cmd(c)[2:]),
           "command_cold": " ".join(["python", "bench.py"] + bench_cmd(c, cold=True, calibrate=True)[2:])}

    # 1. kernel traces: the batch re-solved in place (what the bench line's kernel_ms is), and rotating over >= 512 MiB of copies (cold)
    def trace(cold):
        name = "cold" if cold else "hot"
        d = os.path.join(OUT, "trace_%s_%s" % (c, name))
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench_cmd(c, cold=cold), "%s_%s_trace_%s.log" % (tag, c, name))
        bj = last_json(os.path.join(OUT, "%s_%s_trace_%s.log" % (tag, c, name)))
        stats, timed = None, {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            rows = list(csv.reader(open(f)))
            keep = [rows[0]] + [r for r in rows[1:] if is_solver(r[0])]
            with open(os.path.join(OUT, "%s_%s_kernel_stats%s.csv" % (tag, c, "_cold" if cold else "")), "w", newline="") as g:
                csv.writer(g, quoting=csv.QUOTE_ALL).writerows(keep)
            stats = [dict(zip(rows[0], r)) for r in keep[1:]]
        # (the launches of the warm-up replans through the host entry have the same grid: the stats above average them in too; the
        # per-dispatch trace gives the timed launches alone: the LAST `steps` dispatches of each kernel)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if is_solver(r["Kernel_Name"]) and timed_shape(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])):
                    per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            for k, v in per.items():
                v.sort()
                last = [dur for _, dur in v[-STEPS.get(c, 50):]]
                timed[k] = {"launches": len(last), "avg_ns": sum(last) / len(last), "min_ns": min(last), "max_ns": max(last)}
        return bj, stats, timed

    bj, res["kernel_stats"], res["timed_launches"] = trace(False)
    bjc, res["kernel_stats_cold"], res["timed_launches_cold"] = trace(True)
    res["bench_under_rocprof"] = bj and {k: bj.get(k) for k in ("value", "ms_per_step", "roofline", "roofline_valu", "solver", "config")}
    res["phase"] = "off" if os.environ.get("LSCQP_ACTIVE_SET", "1")[:1] == "0" else "dual active-set phase + interior point behind it"

    # 2. counters, one pass each (never together with a trace).  The memory-side passes run the COLD form of the command -- every launch reads
    # its batch from HBM, as the guide asks before FETCH_SIZE is read -- with the calibration kernels of known byte count in the same session.
    def pmc(counters, name, cold, calibrate):
        dd = os.path.join(OUT, "pmc_%s_%s" % (c, name))
        run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", dd, "--"] + bench_cmd(c, cold=cold, calibrate=calibrate), "%s_%s_pmc_%s.log" % (tag, c, name))
        acc, disp, cal = {}, {}, {}
        steps = STEPS.get(c, 50)
        for f in glob.glob(os.path.join(dd, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            for r in rows:
                kn = r["Kernel_Name"]
                if "lscqp_calib" in kn:
                    cal.setdefault(kn.replace(" ", "").split("(")[0], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    continue
                if not is_solver(kn) or not timed_shape(kn, int(r["Grid_Size"]), int(r["Workgroup_Size"])):
                    continue
                acc.setdefault(kn, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                disp[kn] = {k: r[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size") if k in r}
        # (the timed launches are the last `steps` dispatches of each kernel: the warm-up replans of the batch's construction launch the same kernels)
        return ({kn: {k: sum(v[-steps:]) / len(v[-steps:]) for k, v in cs.items()} for kn, cs in acc.items()}, disp,
                {kn: {k: sum(v) / len(v) for k, v in cs.items()} for kn, cs in cal.items()})

    fetch, disp, cal_f = pmc(["FETCH_SIZE"], "fetch", True, True)
    write, _, cal_w = pmc(["WRITE_SIZE"], "write", True, True)
    sq, _, _ = pmc(["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
                    "SQ_WAIT_INST_ANY"], "sq", False, False)
    kernels = sorted(disp)  # the dual active-set kernel, the interior-point instance behind it (mixed precision: the float instance and the fp64 second pass)
    das = [k for k in kernels if "das_kernel" in k]  # (a mixed-precision config launches the fp64-row instantiation in its warm-up replans: the timed one has the launches)
    main = max(das, key=lambda k: res["timed_launches"].get(k, {}).get("launches", 0)) if das else next((k for k in kernels if "float" in k), kernels[0] if kernels else None)
    res.update({"kernels": kernels, "kernel": main, "qps_per_launch": N, "lsc_neighbours": bj and bj["config"]["lsc_neighbours"],
                "segments": bj and bj["config"]["segments"], "dim": bj and bj["config"]["dim"],
                "dispatch": disp.get(main), "dispatch_all": disp, "sq": sq.get(main), "sq_all": sq,
                "FETCH_SIZE_KB_raw": {k: v.get("FETCH_SIZE") for k, v in fetch.items()},
                "WRITE_SIZE_KB_raw": {k: v.get("WRITE_SIZE") for k, v in write.items()}})
    if bj:
        alg = bj["roofline"]["algorithmic_bytes_per_qp"] * N
        res["algorithmic_bytes_per_launch"] = alg
        nv = bj["config"]["dim"] * bj["config"]["segments"] * 6
        in_bytes = alg - N * (8 * nv + 16)
        row_b = 16 if (bj["config"].get("row_format") == "f32") else 32
        # ONE rule, the same for every config and every kernel (round 6): the counter is divided by what the SAME session counted per known byte
        # of the calibration kernel that reads the way the row stream is read -- LB bytes per lane, contiguous across the wavefront, LB = the
        # row's size (32: lscqp_row, two dwordx4 per lane; 16: lscqp_row_f32) -- the rows being 94 % of a launch's input; WRITE_SIZE by the
        # 8-byte-per-lane store kernel (x_out is written one double per lane).
        def ratio(cal, kind, lb, counter):
            for kn, v in cal.items():
                if ("%s_kernel<%d>" % (kind, lb)) in kn and v.get(counter):
                    return 1024.0 * v[counter] / CALIB_BYTES
            return None

        rf, rw = ratio(cal_f, "read", row_b, "FETCH_SIZE"), ratio(cal_w, "write", 8, "WRITE_SIZE")
        res["fetch_calibration"] = {
            "how": "same rocprofv3 session: FETCH_SIZE / (FETCH_SIZE per known byte of lscqp_calib::read_kernel<row bytes>), WRITE_SIZE / (WRITE_SIZE per known "
                   "byte of lscqp_calib::write_kernel<8>); cold run (>= 512 MiB of batch copies, every launch reads HBM); per kernel instantiation",
            "row_bytes": row_b, "fetch_counted_per_known_byte": rf, "write_counted_per_known_byte": rw,
            "all_read_ratios": {lb: ratio(cal_f, "read", lb, "FETCH_SIZE") for lb in (8, 16, 32)},
            "all_write_ratios": {lb: ratio(cal_w, "write", lb, "WRITE_SIZE") for lb in (8, 16)},
            "calibration_bytes_per_launch": CALIB_BYTES, "known_input_bytes": in_bytes}
        per_kernel = {}
        for kn in kernels:
            f_raw = 1024.0 * (fetch.get(kn, {}).get("FETCH_SIZE") or 0.0)
            w_raw = 1024.0 * (write.get(kn, {}).get("WRITE_SIZE") or 0.0)
            per_kernel[kn] = {"fetch_raw": f_raw, "write_raw": w_raw, "fetch": f_raw / rf if rf else None, "write": w_raw / rw if rw else None}
            if rf and rw:
                per_kernel[kn]["traffic"] = f_raw / rf + w_raw / rw
        res["traffic_per_kernel"] = per_kernel
        res["traffic_bytes_per_launch"] = per_kernel.get(main, {}).get("traffic")  # of the dominant kernel alone
        res["traffic_over_algorithmic"] = (res["traffic_bytes_per_launch"] / alg) if res["traffic_bytes_per_launch"] else None
        for tl, key in ((res["timed_launches"], "roofline_from_trace"), (res["timed_launches_cold"], "roofline_from_trace_cold")):
            if main in tl:
                t = tl[main]["avg_ns"] * 1e-9
                res[key] = {"kernel_time_s": t, "achieved_GBps": alg / t / 1e9, "frac_of_8TBps": alg / t / 8.0e12}
    json.dump(res, open(os.path.join(OUT, "%s_%s_pmc.json" % (tag, c)), "w"), indent=1)
    return res


summary = {}
if "nobench" not in os.environ.get("PROFILE_SKIP", ""):
    run([sys.executable, os.path.join(ROOT, "bench.py")], tag + "_bench.log")
    b = last_json(os.path.join(OUT, tag + "_bench.log"))
    json.dump(b, open(os.path.join(OUT, tag + "_bench.json"), "w"))
    summary["bench"] = b and {k: b[k] for k in ("value", "ms_per_step")}
for c in which:
    r = profile_config(c)
    summary[c] = {"timed": r.get("timed_launches"), "traffic": r.get("traffic_bytes_per_launch"), "alg": r.get("algorithmic_bytes_per_launch"),
                  "dispatch": r.get("dispatch")}
if "nonext" not in os.environ.get("PROFILE_SKIP", ""):
    # the kernels either side of the QP (generation, neighbour selection, safety metrics, voxel map, corridors)
    NEXT = [sys.executable, os.path.join(ROOT, "tools", "bench_next_rows.py"), "4096"]
    dn = os.path.join(OUT, "trace_next_rows")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", dn, "--"] + NEXT, tag + "_next_rows_trace.log")
    lines = [l for l in open(os.path.join(OUT, tag + "_next_rows_trace.log")).read().splitlines() if l.startswith("{")]
    with open(os.path.join(OUT, tag + "_next_rows.jsonl"), "w") as f:
        f.write("\n".join(lines) + "\n")
    for f in glob.glob(os.path.join(dn, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ("generate_lsc_kernel", "safety_metrics", "construct_sfc", "nearest_",
                                                                            "rasterise", "select_neighbours", "shift_traj", "goal_kernel", "validate_step"))]
        with open(os.path.join(OUT, tag + "_next_rows_kernel_stats.csv"), "w", newline="") as g:
            csv.writer(g, quoting=csv.QUOTE_ALL).writerows(keep)
    # the two replan chains of bench.py (replan_chain): which kernel takes what of a replan
    for mode, name in (([], "c1class"), (["forest10"], "forest10")):
        dc = os.path.join(OUT, "trace_chain_" + name)
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", dc, "--", sys.executable, os.path.join(ROOT, "tools", "chain_profile.py")] + mode,
            "%s_chain_%s_trace.log" % (tag, name))
        for f in glob.glob(os.path.join(dc, "**", "*kernel_stats.csv"), recursive=True):
            with open(os.path.join(OUT, "%s_chain_%s_kernel_stats.csv" % (tag, name)), "w") as g:
                g.write(open(f).read())
print(json.dumps(summary))

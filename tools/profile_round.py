"""Collect the profile artefacts of one round ON THE GPU BOX and write them under gpurun_out/profiles_<tag>/.

    python tools/profile_round.py r02_v1 [c1 c4_f64 c4 c3s ...]      (default: every BASELINE config of bench.py)

Per config (bench.py --config <c>: that config is then the timed workload, nothing else launches the QP kernel):
  1. rocprofv3 --kernel-trace --stats -- python bench.py --config <c> ...   -> <tag>_<c>_kernel_stats.csv (+ the bench line)
  2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate passes, never together with a trace)
                                                                             -> <tag>_<c>_pmc.json
and once: the plain `python bench.py` line (-> <tag>_bench.json) and the kernels either side of the QP.
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KB and need
separate passes; contiguous 16 B/lane streaming reads are tallied at half their size on gfx950 (x2), other patterns are to be
calibrated on a known byte count -- done below against the kernel's exactly known input volume.
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
STEPS = {"c1": 200, "c0": 200, "c2": 100, "c3s": 60, "c3": 20, "c4": 60, "c4_f64": 60}


def bench_cmd(c):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--steps", str(STEPS.get(c, 50)), "--warmup", "10",
            "--no-cpu-baseline", "--no-extra", "--no-latency"]  # no side loops: the trace averages the timed launches alone


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


def profile_config(c):
    from bench import CONFIGS

    N = CONFIGS[c]["agents"]

    def timed_shape(name, grid, wg):
        """a launch over the whole batch: one workgroup per instance (grid = N workgroups), or the persistent form of the kernel (its last
        template argument is `true`; the grid is then what the chip holds)"""
        return grid == N * wg or name.replace(" ", "").split(">(")[0].endswith(",true")

    res = {"config": c, "what": CONFIGS[c]["what"], "command": " ".join(["python", "bench.py"] + bench_cmd(c)[2:])}
    # 1. kernel trace
    d = os.path.join(OUT, "trace_" + c)
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench_cmd(c), "%s_%s_trace.log" % (tag, c))
    bj = last_json(os.path.join(OUT, "%s_%s_trace.log" % (tag, c)))
    res["bench_under_rocprof"] = bj and {k: bj.get(k) for k in ("value", "ms_per_step", "roofline", "roofline_valu", "solver", "config")}
    res["phase"] = "off" if os.environ.get("LSCQP_ACTIVE_SET", "1")[:1] == "0" else "dual active-set phase + interior point behind it"
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if "lscqp_pdip_kernel" in r[0] or "das_kernel" in r[0]]
        with open(os.path.join(OUT, "%s_%s_kernel_stats.csv" % (tag, c)), "w", newline="") as g:
            csv.writer(g, quoting=csv.QUOTE_ALL).writerows(keep)
        res["kernel_stats"] = [dict(zip(rows[0], r)) for r in keep[1:]]
    # (the launches of the warm-up replans through the host entry have the same grid: the stats above average them in too; the
    # per-dispatch trace gives the timed launches alone: the LAST steps+warmup dispatches of each kernel)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if ("lscqp_pdip_kernel" in r["Kernel_Name"] or "das_kernel" in r["Kernel_Name"]) and timed_shape(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])):
                per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        res["timed_launches"] = {}
        for k, v in per.items():
            v.sort()
            last = [dur for _, dur in v[-STEPS.get(c, 50):]]
            res["timed_launches"][k] = {"launches": len(last), "avg_ns": sum(last) / len(last), "min_ns": min(last), "max_ns": max(last)}

    # 2. counters, one pass each
    def pmc(counters, name):
        dd = os.path.join(OUT, "pmc_%s_%s" % (c, name))
        run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", dd, "--"] + bench_cmd(c), "%s_%s_pmc_%s.log" % (tag, c, name))
        acc, disp = {}, {}
        for f in glob.glob(os.path.join(dd, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if ("lscqp_pdip_kernel" not in r["Kernel_Name"] and "das_kernel" not in r["Kernel_Name"]) or not timed_shape(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"])):
                    continue
                kn = r["Kernel_Name"]
                acc.setdefault(kn, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                disp[kn] = {k: r[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size") if k in r}
        return {kn: {k: sum(v) / len(v) for k, v in cs.items()} for kn, cs in acc.items()}, disp

    fetch, disp = pmc(["FETCH_SIZE"], "fetch")
    write, _ = pmc(["WRITE_SIZE"], "write")
    sq, _ = pmc(["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
                 "SQ_WAIT_INST_ANY"], "sq")
    # instruction fetch (is a kernel whose loop body exceeds the 64 KB instruction cache thrashing it?) and the hardware's own count of
    # fp64 vector instructions (cross-check of lscqp_instance_work's machine-code count); separate passes, the SQC block has few slots
    ic, _ = pmc(["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"], "icache")
    ic2, _ = pmc(["SQC_TC_INST_REQ", "SQ_IFETCH", "SQ_WAIT_ANY", "SQ_WAVES"], "ifetch")
    fl, _ = pmc(["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_WAVES"], "f64")
    kernels = sorted(disp)  # the dual active-set kernel, the interior-point instance behind it (mixed precision: the float instance and the fp64 second pass)
    das = [k for k in kernels if "das_kernel" in k]  # (a mixed-precision config launches the fp64-row instantiation in its warm-up replans: the timed one has the launches)
    main = max(das, key=lambda k: res["timed_launches"].get(k, {}).get("launches", 0)) if das else next((k for k in kernels if "float" in k), kernels[0] if kernels else None)
    res.update({"kernels": kernels, "kernel": main, "qps_per_launch": N, "lsc_neighbours": bj and bj["config"]["lsc_neighbours"],
                "segments": bj and bj["config"]["segments"], "dim": bj and bj["config"]["dim"],
                "dispatch": disp.get(main), "dispatch_all": disp, "sq": sq.get(main), "sq_all": sq,
                "icache": dict(ic.get(main) or {}, **(ic2.get(main) or {})), "f64_insts": fl.get(main),
                "FETCH_SIZE_KB_raw": {k: v.get("FETCH_SIZE") for k, v in fetch.items()},
                "WRITE_SIZE_KB_raw": {k: v.get("WRITE_SIZE") for k, v in write.items()},
                "correction": "MI355X_MICROARCH.md (HBM): both counters are KB; FETCH_SIZE is doubled for contiguous 16 B/lane streaming "
                              "reads and must be calibrated on a known byte count for other patterns (fetch_calibration); WRITE_SIZE as is"})
    if res["icache"].get("SQC_ICACHE_REQ"):
        i = res["icache"]
        res["icache"]["hit_rate"] = i.get("SQC_ICACHE_HITS", 0.0) / i["SQC_ICACHE_REQ"]
        # one SQC->TC instruction request fetches one 64-byte line
        res["icache"]["inst_bytes_from_L2_per_launch"] = 64.0 * i.get("SQC_TC_INST_REQ", 0.0)
    if res.get("f64_insts") and res["f64_insts"].get("SQ_WAVES"):
        f = res["f64_insts"]
        per_wave = {k: f.get(k, 0.0) / f["SQ_WAVES"] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
        res["f64_insts_per_wave"] = per_wave
        res["f64_flops_per_launch_pmc"] = 64.0 * (f.get("SQ_INSTS_VALU_ADD_F64", 0) + f.get("SQ_INSTS_VALU_MUL_F64", 0) + 2.0 * f.get("SQ_INSTS_VALU_FMA_F64", 0) + f.get("SQ_INSTS_VALU_TRANS_F64", 0))
        if bj and bj.get("roofline_valu"):
            res["f64_flops_per_launch_static"] = bj["roofline_valu"]["fp64_flops_per_launch"]
    if bj:
        alg = bj["roofline"]["algorithmic_bytes_per_qp"] * N
        res["algorithmic_bytes_per_launch"] = alg
        raw_f = 1024.0 * sum(v.get("FETCH_SIZE", 0.0) for v in fetch.values())
        raw_w = 1024.0 * sum(v.get("WRITE_SIZE", 0.0) for v in write.values())
        nv = bj["config"]["dim"] * bj["config"]["segments"] * 6
        in_bytes = alg - N * (8 * nv + 16)
        # Every input byte is read once per launch by the kernel that dominates (the dual active-set kernel reads rows, header, boxes,
        # offset; the interior-point pass behind it reads only statuses).  The guide's correction -- contiguous 16 B/lane streaming reads
        # are tallied at half their size on gfx950 -- concerns the ROW stream only (dwordx4 loads); headers, boxes, tables and instruction
        # fetch are counted as they are.  So the raw counter is compared with the two volumes it can stand for and corrected by exactly
        # the rows' missing half when it matches the second; a counter that matches neither is reported RAW with that said (never "x 2").
        row_b = 16 if (bj["config"].get("row_format") == "f32") else 32
        rows_bytes = N * bj["config"]["lsc_neighbours"] * bj["config"]["segments"] * 6 * row_b
        full, half = in_bytes, in_bytes - 0.5 * rows_bytes
        if in_bytes and abs(raw_f - full) <= 0.15 * full:
            corr, how = raw_f, "as counted (matches the input volume)"
        elif in_bytes and abs(raw_f - half) <= 0.15 * half:
            corr, how = raw_f + 0.5 * rows_bytes, "row stream tallied at half its size: + rows / 2"
        else:
            corr, how = raw_f, "uncalibrated: the raw counter matches neither the input volume nor the volume with the row stream at half; reported raw"
        res["fetch_calibration"] = {"how": how, "known_input_bytes": in_bytes, "row_stream_bytes": rows_bytes, "raw_fetch_bytes": raw_f,
                                    "instruction_bytes_from_L2": res["icache"].get("inst_bytes_from_L2_per_launch")}
        res["traffic_bytes_per_launch"] = corr + raw_w
        if main in res.get("timed_launches", {}):
            t = sum(v["avg_ns"] for v in res["timed_launches"].values()) * 1e-9
            res["roofline_from_trace"] = {"step_time_s": t, "achieved_GBps": alg / t / 1e9, "frac_of_8TBps": alg / t / 8.0e12}
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

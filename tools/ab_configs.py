"""Development aid: A/B of two builds of the library on the bench configs, alternating A, B, A, B in separate processes.

    LSCQP_AB=<name> LSCQP_EXTRA_FLAGS=-D... python -m lsc_dr_planner_amd.build       (here; the .so travels to the GPU box)
    python tools/ab_configs.py <name> [c1 c0 c3s ...]                                 (on the GPU box)

Prints, per config, the launch duration (HIP events over the timed steps of `bench.py --config <c>`) of the product library and of
liblscqp_<name>.so, best of the rounds, and their ratio."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
configs = sys.argv[2:] or ["c1", "c0", "c3s", "c3", "c4_f64", "c2"]
libs = {"product": os.path.join(ROOT, "lsc_dr_planner_amd", "liblscqp.so"), name: os.path.join(ROOT, "lsc_dr_planner_amd", "liblscqp_%s.so" % name)}
STEPS = {"c1": 200, "c0": 200, "c2": 100, "c3s": 60, "c3": 20, "c4": 40, "c4_f64": 40}
out = {}
for c in configs:
    best = {k: 1e30 for k in libs}
    its = {}
    for rnd in range(3):
        for k, lib in libs.items():
            env = dict(os.environ, LSCQP_LIB=lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--steps", str(STEPS.get(c, 50)), "--warmup", "10",
                                "--no-cpu-baseline", "--no-extra", "--no-latency"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                print(c, k, "FAILED", r.stderr[-400:])
                continue
            b = json.loads(line[-1])
            best[k] = min(best[k], b["roofline"]["kernel_ms"])
            its[k] = (b["solver"]["iters_mean"], b["solver"]["iters_max"], b["solver"]["non_optimal"])
    out[c] = {"kernel_ms": best, "iterations": its, "ratio_%s_over_product" % name: best[name] / best["product"]}
    print(json.dumps({c: out[c]}))

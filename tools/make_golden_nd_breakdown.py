"""Generates tests/golden/nd_breakdown_m10d2.json: the one instance of tools/sweep_class_params.py (seed 0, trial 33, instance 4:
M = 10 in 2-D, dt 0.2, w_c 0.038, w_t 0.84, no communication range, no neighbours) on which the nested-dissection instance ends NUMERIC
at iteration 11 -- a pivot of the late-iteration matrix cancels to <= 0 -- while the natural elimination order (and the oracle) solve it.
Pure CPU (the draws of the sweep + the oracle's optimum)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from oracle import oracle

    import sweep_class_params as S

    oracle.build()
    rng = np.random.default_rng(0)
    for trial in range(34):
        par, ags, Ls, boxes = S.draw_trial(rng, trial, oracle)
    q = 4
    cls = oracle.make_class(use_sfc=True, **par)
    o = oracle.solve(cls, ags[q], Ls[q], boxes[q])
    assert o["status"] == 0 and Ls[q] is None
    ag = ags[q]
    flat = lambda v: np.asarray(v).reshape(-1).tolist()  # noqa: E731
    fx = dict(source="tools/sweep_class_params.py seed 0, trial 33, instance 4 (tools/make_golden_nd_breakdown.py)",
              params={k: (flat(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in par.items()},
              agent={k: (flat(ag[k]) if np.asarray(ag[k]).size > 1 else float(flat(ag[k])[0]))
                     for k in ("p0", "v0", "a0", "goal", "next_waypoint", "vmax", "amax", "radius", "nominal_velocity")},
              sfc_min=boxes[q]["bmin"].tolist(), sfc_max=boxes[q]["bmax"].tolist(), oracle_obj=o["obj"], oracle_x=o["x"].tolist())
    json.dump(fx, open(os.path.join(ROOT, "tests", "golden", "nd_breakdown_m10d2.json"), "w"))
    print("nd_breakdown_m10d2.json written; oracle objective", o["obj"], "in", o["iters"], "iterations")


if __name__ == "__main__":
    main()

"""All rows of the path chained in closed loop on the device, in the reference's default configuration (generateCLSC +
constructSFCFromConvexHull + goal LP + trajectory QP + isSolValid / doStep + safety metrics), on the reference's forest10
world with its 10 agents (tools/closed_loop.py; the waypoints come from a host-side stand-in for the out-of-scope grid
planner, without MAPF conflict resolution)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_forest10_closed_loop_is_safe_and_feasible():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop

    log = closed_loop.run(os.path.join(ROOT, "tests", "golden", "forest10_world.json"), steps=60)
    # every QP of 60 replans x 10 agents solves and passes isSolValid: the generated constraints are mutually consistent
    assert log["qp_failed"] == 0 and log["invalid"] == 0 and log["sfc_kept"] == 0, log
    # the LSC guarantee: agents never come closer than the sum of their radii (reference summary: safety_ratio_agent >= 1)
    # (up to the float32 truncation of the control points, which the reference applies as well: 5e-7 m on a 5 m coordinate
    # is 3e-6 of the 0.3 m the ratio is measured in)
    assert log["min_safety_ratio"] >= 1.0 - 5e-6, log
    assert log["max_vel_excess"] <= 1e-5 and log["max_acc_excess"] <= 1e-5, log
    assert log["mean_progress_m"] > 1.5, log  # and they do fly towards their goals
    assert log["max_iters"] <= 30, log

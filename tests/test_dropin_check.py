"""The drop-in claim checked mechanically (tools/check_dropin.py): the shim's traj_optimizer.cpp / goal_optimizer.cpp type-check against
the REFERENCE's own headers (Param, Mission, Agent, CollisionConstraints, LSC, Box, Trajectory as /root/reference/include declares
them).  Build container only -- skipped where the reference checkout does not exist (the GPU box)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="needs the reference checkout (build container only)")


def _tool():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dropin

    return check_dropin


def test_shim_sources_type_check_against_the_reference_headers():
    res, used = _tool().check()
    assert [src for src, rc, _ in res if rc != 0] == [], [err[-1500:] for _, rc, err in res if rc != 0]
    # the types came from the reference's headers, not from the stand-ins of shim/include
    for h in ("param.hpp", "mission.hpp", "sp_const.hpp", "collision_constraints.hpp", "trajectory.hpp", "polynomial.hpp"):
        assert h in used, used
    assert "traj_optimizer.hpp" not in used and "goal_optimizer.hpp" not in used  # the two headers the shim replaces (CPLEX-free)


def test_the_check_bites(tmp_path):
    """Negative control: a source that touches a member the reference's Param does not have fails; the same source with a real
    member passes."""
    bad, good = tmp_path / "bad.cpp", tmp_path / "good.cpp"
    bad.write_text("#include <traj_optimizer.hpp>\nint f(const DynamicPlanning::Param& p) { return p.world_dimension + p.not_a_member_of_param; }\n")
    good.write_text("#include <traj_optimizer.hpp>\nint f(const DynamicPlanning::Param& p, const DynamicPlanning::Agent& a, const DynamicPlanning::CollisionConstraints& c) {\n"
                    "    DynamicPlanning::LSC l = c.getLSC(0, 0, 0);\n    return p.world_dimension + a.id + (int)c.getObsSize() + (int)l.d + (int)c.getSFC(0).box_min.x();\n}\n")
    res, _ = _tool().check(extra_sources=[str(bad), str(good)])
    rc = {os.path.basename(s): r for s, r, _ in res}
    assert rc["bad.cpp"] != 0 and rc["good.cpp"] == 0

#!/usr/bin/env python
"""Mechanical check of the drop-in claim ("second variant compiled inside the reference tree").

`g++ -std=c++17 -fsyntax-only` on the shim's traj_optimizer.cpp and goal_optimizer.cpp with THE REFERENCE'S OWN HEADERS on the include
path -- /root/reference/include/{param,mission,sp_const,collision_constraints,trajectory,polynomial,...}.hpp -- instead of the shim's
stand-ins: only the two headers the shim replaces (traj_optimizer.hpp, goal_optimizer.hpp; they lose <ilcplex/ilocplex.h>) come from
lsc_dr_planner_amd/shim/include.  If the shim touched a member the reference's Param / Mission / Agent / State / CollisionConstraints /
LSC / Box / Trajectory does not have, or used it with another type, this fails to compile.

ROS, octomap, dynamicEDT3D and Eigen do not exist in this image; tools/dropin_stubs/ declares just their NAMES so that the reference
headers parse (its README says what that is and is not).  Build container only: it reads /root/reference, so nothing on the GPU box or
in the -m gpu tests may call it; tests/test_dropin_check.py runs it when the reference checkout is present.

usage: python tools/check_dropin.py [-v]      exit code 0 = both files type-check against the reference's headers
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LSC_REFERENCE", "/root/reference")
SHIM = os.path.join(ROOT, "lsc_dr_planner_amd", "shim")
STUBS = os.path.join(ROOT, "tools", "dropin_stubs")
REPLACED = ["traj_optimizer.hpp", "goal_optimizer.hpp", "eigen_standin.hpp"]  # the last one resolves to <Eigen/Dense> when that exists
SOURCES = ["traj_optimizer.cpp", "goal_optimizer.cpp"]


def check(verbose=False, extra_sources=()):
    """extra_sources: further .cpp files type-checked the same way (the test suite's negative control)."""
    if not os.path.isdir(os.path.join(REF, "include")):
        raise SystemExit("check_dropin: no reference checkout at %s (build container only)" % REF)
    results = []
    with tempfile.TemporaryDirectory() as td:
        hdr = os.path.join(td, "replaced")
        os.makedirs(hdr)
        for f in REPLACED:
            shutil.copy(os.path.join(SHIM, "include", f), hdr)
        for src in list(SOURCES) + list(extra_sources):
            path = src if os.path.isabs(src) else os.path.join(SHIM, "src", src)
            cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-sign-compare", "-Wno-comment",
                   "-I", hdr, "-I", os.path.join(REF, "include"), "-I", STUBS, "-I", os.path.join(ROOT, "include"), path]
            r = subprocess.run(cmd, capture_output=True, text=True)
            results.append((src, r.returncode, r.stderr))
            if verbose or r.returncode != 0:
                sys.stderr.write("$ %s\n%s\n" % (" ".join(cmd), r.stderr[-6000:]))
        # which reference headers were actually parsed (proof that the stand-ins of shim/include were NOT on the path)
        deps = subprocess.run(["g++", "-std=c++17", "-MM", "-I", hdr, "-I", os.path.join(REF, "include"), "-I", STUBS, "-I", os.path.join(ROOT, "include"),
                               os.path.join(SHIM, "src", SOURCES[0])], capture_output=True, text=True).stdout
    used = sorted({os.path.basename(t) for t in deps.replace("\\\n", " ").split() if t.startswith(os.path.join(REF, "include")) and t.count("/") == REF.count("/") + 2})
    return results, used


if __name__ == "__main__":
    res, used = check(verbose="-v" in sys.argv)
    for src, rc, _ in res:
        print("%-22s %s" % (src, "type-checks against the reference's headers" if rc == 0 else "FAILED"))
    print("reference headers parsed:", ", ".join(used))
    sys.exit(0 if all(rc == 0 for _, rc, _ in res) else 1)

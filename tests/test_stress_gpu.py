"""Randomized sweeps (-m gpu), bounded seeds: the acceptance rules of the solver -- the fallback acceptance at the rounding floor,
the early INFEASIBLE verdicts, the re-centring of jammed warm starts -- are heuristics, so they are regression-tested on batches
the fixed parity cases do not contain.  A disagreement is one side OPTIMAL and the other not (a false INFEASIBLE silently becomes
"fall back to the initial trajectory" in the planner), or two optima outside the parity bar (objective 1e-8, x 1e-6 m).
The sweeps are the development tools of tools/ run with fewer seeds (8 shapes x 2 seeds x 3 replans each)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, *args, timeout=1500):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + list(args), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("args", [("2",), ("2", "--warm"), ("2", "--warm", "--gen", "2")], ids=["cold", "warm", "warm-bvc-rows-from-the-device"])
def test_solver_agrees_with_the_oracle_on_randomized_batches(args):
    out = _run("stress_parity.py", *args)
    m = re.search(r"TOTAL mismatches (\d+)", out)
    assert m, out[-1500:]
    assert int(m.group(1)) == 0, "\n".join(l for l in out.splitlines() if "MISMATCH" in l or "seeds" in l)


def test_generators_agree_with_the_oracle_on_randomized_swarms():
    out = _run("sweep_generators.py")
    m = re.search(r"nbad (\d+)", out)
    assert m and int(m.group(1)) == 0, out[-1500:]


def test_corridors_agree_with_the_oracle_on_randomized_worlds():
    out = _run("sweep_corridors.py")
    m = re.search(r"bad (\d+) of (\d+)", out)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) > 0, out[-1500:]

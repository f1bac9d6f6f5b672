import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "pdip_only: a test of the interior-point kernel's own mechanics -- runs with the dual active-set phase switched off")


# Round 5: a dual active-set phase (csrc/lscqp_das.hip) runs in front of the interior-point kernel and finishes most instances of every
# workload the fixtures hold.  The parity modules below therefore run every test TWICE -- through the product's default path
# ("active_set": the phase, then the interior-point kernel for what it leaves) and with the phase switched off ("interior_point":
# rounds 1-4, still the engine behind the phase) -- so neither kernel can hide behind the other; tests of the interior-point kernel's own
# mechanics (re-centring, rescue pass, elimination orders, compiled-instance determinism ...) carry @pytest.mark.pdip_only and run on it
# alone.  The switch is the environment knob the library reads at every launch (LSCQP_ACTIVE_SET_NOW).
_DUAL_PATH = {"test_gpu_parity", "test_generic", "test_kat_3d", "test_row_format", "test_active_set"}


def pytest_generate_tests(metafunc):
    if "solver_path" not in metafunc.fixturenames:
        return
    mod = metafunc.module.__name__.split(".")[-1]
    if metafunc.definition.get_closest_marker("pdip_only"):
        metafunc.parametrize("solver_path", ["interior_point"], indirect=True)
    elif mod in _DUAL_PATH and metafunc.definition.get_closest_marker("gpu"):
        metafunc.parametrize("solver_path", ["active_set", "interior_point"], indirect=True)


@pytest.fixture(autouse=True)
def solver_path(request, monkeypatch):
    path = getattr(request, "param", "active_set")
    monkeypatch.setenv("LSCQP_ACTIVE_SET_NOW", "0" if path == "interior_point" else "1")
    return path


# Collection order of the suite (the driver runs it with -x: whatever comes first must be what matters most).  Oracle / golden
# parity of the QP first (SURVEY.md section 8 rows a, c), then the C++ class surface, then the rows of 8f, then sharding /
# communicator / closed loop, then the full-size property and stress tests, and LAST the tests that spawn bench.py as a
# subprocess (they re-check parity through the bench line and cost the most wall-clock per assertion).
_ORDER = ["test_oracle", "test_abi", "test_row_format", "test_synth", "test_dropin_check",
          "test_gpu_parity", "test_active_set", "test_round6_host", "test_kat_3d", "test_shim", "test_generic", "test_mixed_precision", "test_floor_audit", "test_diag",
          "test_lscgen", "test_lscmode", "test_prediction", "test_goal", "test_post", "test_sfc",
          "test_plan", "test_closed_loop", "test_comm", "test_dist_cpu", "test_race_twin",
          "test_full_size_properties", "test_stress_gpu", "test_profiles", "test_bench_contract"]


def pytest_collection_modifyitems(session, config, items):
    rank = {m: i for i, m in enumerate(_ORDER)}

    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        return rank.get(mod, len(_ORDER) - 2)  # unknown modules: before the profile / bench-subprocess tests

    items.sort(key=key)  # stable: the order inside a module is the file's own


class _TightOracle:
    """oracle/oracle.py with the CHECKER's last step of round 6: every OPTIMAL point replaced by tests/helpers.py: polish_primal -- the optimum of the same row-for-row model by an exact method (null space + least-distance problem +
    Lawson-Hanson NNLS + one equality-constrained solve on the active rows).  At its default tol = 1e-11 the oracle's interior-point iterate
    sits up to 2.7e-6 m off the optimum on flat instances (profiles/r05_v6_stress_parity.txt), and a few 1e-8 m off on some instances at ANY
    tolerance (the reference's log pipeline, case 37: both device kernels reach a lower objective) -- looser than the kernel it checks.
    The polished point is within ~1e-10 m of a 1e-14 iterate wherever that one is good, and is what lets the parity modules hold the
    dual active-set phase to 1e-8 m (tests/helpers.py: PathTol).  Objective, multipliers and statuses are the oracle's own."""

    TOL, MAX_ITER = 1e-11, 200  # (the module's own defaults: a tighter gap target does not help -- see above -- and costs the hard KATs their convergence)

    def __init__(self, mod):
        self._m = mod

    def __getattr__(self, name):
        return getattr(self._m, name)

    def solve(self, cls, agent, lsc=None, sfc=None, tol=None, max_iter=None, polish=True):
        import numpy as np

        from tests import helpers as H

        r = self._m.solve(cls, agent, lsc, sfc, tol=self.TOL if tol is None else tol, max_iter=self.MAX_ITER if max_iter is None else max_iter)
        if polish and r["status"] == 0:
            x, ok = H.polish_primal(self._m, cls, np.ascontiguousarray(agent).reshape(-1)[:1], lsc, sfc, x0=r["x"])
            if ok:
                r["x_interior_point"], r["x"] = r["x"], x
        return r

    def solve_batch(self, cls, agents, lsc=None, lsc_off=None, sfc=None, tol=None, max_iter=None, threads=1, polish=True):
        import numpy as np

        from tests import helpers as H

        R = self._m.solve_batch(cls, agents, lsc, lsc_off, sfc, tol=self.TOL if tol is None else tol,
                                max_iter=self.MAX_ITER if max_iter is None else max_iter, threads=threads)
        if polish:
            from concurrent.futures import ThreadPoolExecutor

            agents = np.ascontiguousarray(agents)
            M = cls.M

            def one(q):
                if R["status"][q] != 0:
                    return
                lq = None
                if lsc is not None and lsc_off is not None:
                    n_rows = int(agents["n_obs"][q]) * M * 6
                    lq = np.ascontiguousarray(lsc[int(lsc_off[q]): int(lsc_off[q]) + n_rows])
                sq = None if sfc is None else np.ascontiguousarray(sfc[q * M:(q + 1) * M])
                x, ok = H.polish_primal(self._m, cls, agents[q:q + 1], lq, sq, x0=R["x"][q])
                if ok:
                    R["x"][q] = x

            # (dense linear algebra releases the interpreter lock: the instances of a batch are polished side by side)
            with ThreadPoolExecutor(max_workers=max(1, min(8, len(agents)))) as ex:
                list(ex.map(one, range(len(agents))))
        return R


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return _TightOracle(O)


@pytest.fixture(scope="session")
def api():
    """The C-ABI library.  Built in-tree by __graft_entry__.build(); if missing it is built here (hipcc
    cross-compiles without a GPU)."""
    from lsc_dr_planner_amd import api as A
    from lsc_dr_planner_amd import build as B

    if not os.path.exists(A.LIB_PATH):
        B.build()
    A.lib()
    return A


@pytest.fixture(scope="session")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch

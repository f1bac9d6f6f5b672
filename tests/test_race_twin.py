"""The dual active-set kernel against its full-synchronisation twin, over fresh processes (round 6; VERDICT r05 "weak" 2, "next" 3).

csrc/lscqp_das.hip hands LDS data over between wavefronts with `s_waitcnt lgkmcnt(0); s_barrier` -- a barrier that waits for the LDS counter
only, so that the global loads of the first pass stay in flight across the prologue -- and inside one wavefront with wavefront-scope fences.
That is the kernel's one assumption about ordering, it had one LDS race in round 5 (commit 43ef9bd), and two process-level observations of that
round never reproduced (NOTES.md section 13).  liblscqp_sync.so (lsc_dr_planner_amd/build.py: -DLSCQP_DAS_FULL_SYNC) is the same source with
every hand-over replaced by workgroup-scope fences + a wait for EVERYTHING in flight + the compiler's __syncthreads(): free of the assumption by
construction.  Here: >= 200 FRESH processes (tests/_race_worker.py: no torch, HIP initialised by the library), each solving the forest10
replica at its busiest stretch (55 steps, > 20 active rows at one agent: every step form of the kernel -- joins, leaving rows, polish) and
the dense-maze batch 10 replans in, twice, with both libraries; every process must report bit-identical results from product and twin, and
every process the same digest."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PROCESSES = int(os.environ.get("LSCQP_RACE_PROCESSES", "200"))


def _fixture(api, path):
    import bench
    from lsc_dr_planner_amd import synth

    out = {}
    keys = [("c0_loaded", None), ("c2", 10)]
    for b, (key, warm) in enumerate(keys):
        cfg = bench.CONFIGS[key]
        N, M, dim = cfg["agents"], cfg["segments"], cfg["dim"]
        sw, sol, build, (hdr, rows, off, sfc) = bench.make_batch(
            api, synth, lambda s: api.Solver(api.make_desc(M=M, dim=dim, world_min=s.world_min, world_max=s.world_max)), N, M, dim, cfg["obs"],
            seed=cfg["seed"], style=cfg["style"], warm_steps=warm or cfg.get("warm_steps", 3))
        out.update({"hdr_%d" % b: np.ascontiguousarray(hdr).view(np.uint8), "rows_%d" % b: np.ascontiguousarray(rows).view(np.uint8),
                    "off_%d" % b: np.ascontiguousarray(off, dtype=np.uint64), "sfc_%d" % b: np.ascontiguousarray(sfc).view(np.uint8),
                    "x0_%d" % b: np.ascontiguousarray(api.x_init_from_swarm(build, dim)), "M_%d" % b: M, "dim_%d" % b: dim,
                    "wmin_%d" % b: np.asarray(sw.world_min, dtype=np.float64), "wmax_%d" % b: np.asarray(sw.world_max, dtype=np.float64)})
    out["n_batches"] = len(keys)
    np.savez(path, **out)


@pytest.mark.gpu
def test_product_and_full_sync_twin_agree_bit_for_bit_over_fresh_processes(api, torch_cuda, tmp_path):
    from lsc_dr_planner_amd import build as B

    assert os.path.exists(B.SYNC_LIB), "liblscqp_sync.so not built (lsc_dr_planner_amd/build.py builds it with the product)"
    fx = str(tmp_path / "race_fixture.npz")
    _fixture(api, fx)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_race_worker.py"), fx, api.LIB_PATH, B.SYNC_LIB]

    def one(_):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-1500:]
        return r.stdout.strip().splitlines()[-1]

    with ThreadPoolExecutor(max_workers=8) as ex:
        lines = list(ex.map(one, range(N_PROCESSES)))
    assert all("DIFFER" not in l and l.count("same") == 2 for l in lines), [l for l in lines if "DIFFER" in l][:3]
    assert len(set(lines)) == 1, sorted(set(lines))[:4]
    # the fixture really is the busy one: the forest10 replica's slowest agent takes tens of steps, every instance is solved
    w = lines[0].split()
    assert int(w[1][5:]) >= 30 and w[2] == "nonopt0" and w[5] == "nonopt0", lines[0]

"""Development aid: the 3-D replan chain of bench.py (replan_chain_3d) alone, for `rocprofv3 --kernel-trace --stats`.
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/chain -- python tools/chain_profile.py [N] [replans]
       ... python tools/chain_profile.py forest10        (the forest10 mission's chain instead: bench.replan_chain, eager + graph)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lsc_dr_planner_amd import api  # noqa: E402

if __name__ == "__main__":
    if "forest10" in sys.argv:
        bench.replan_chain_3d = lambda *a, **k: None
        print(json.dumps(bench.replan_chain(torch, api)))
        sys.exit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    replans = int(sys.argv[2]) if len(sys.argv) > 2 else 41
    print(json.dumps(bench.replan_chain_3d(torch, api, N=N, replans=replans)))

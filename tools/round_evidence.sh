#!/bin/bash
# Development aid: the evidence of a round in one GPU lease -- the -m gpu suite, the profile set (tools/profile_round.py <tag>), the stress sweep.
#   bash tools/round_evidence.sh r05_v2      (GPU box; writes under gpurun_out/)
tag=${1:-r05_v2}
cd "$(dirname "$0")/.."
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${tag}_gpu_suite.txt
python tools/profile_round.py $tag > gpurun_out/profile_round_${tag}.log 2>&1
(for m in "" "--warm" "--dlsc" "--warm --dlsc" "--nd" "--nd --dlsc" "--warm --gen 1" "--warm --gen 2"; do echo "== stress_parity 25 --seed0 400 $m"; timeout 600 python tools/stress_parity.py 25 --seed0 400 $m 2>&1 | grep -v amdgpu.ids; done) > gpurun_out/${tag}_stress_parity.txt 2>&1
tail -3 gpurun_out/${tag}_gpu_suite.txt

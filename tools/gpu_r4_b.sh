#!/bin/bash
# Round 4, visit B: hunt the once-red in-flight test: uninitialised-memory poison legs, stress legs, and the driver's own
# conditions (the file's first tests in one process, in the order GPUTEST_r03 ran them), three times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== poison"; timeout 300 python tools/r4_stress.py --poison 2>&1 | grep -v amdgpu.ids
  echo "== stress"; timeout 300 python tools/r4_stress.py --stress 60 2>&1 | grep -v amdgpu.ids
  for i in 1 2 3; do
    echo "== driver conditions $i"
    timeout 300 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -p no:cacheprovider \
      -k "encoder_vs_oracle or rope or decoder_logprobs or lm_scorer or beam66 or greedy or properties or in_flight" 2>&1 | tail -5
  done
} 2>&1 | tee gpurun_out/r4_b.log

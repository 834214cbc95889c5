#!/bin/bash
# Round 4, visit A: (1) discriminators for the red in-flight test, (2) the prepared split-operand variants timed,
# (3) the whole GPU suite in its new collection order, without -x (everything that is red, not just the first)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== bisect"; timeout 420 python tools/r4_bisect_inflight.py 2>&1 | grep -v amdgpu.ids
  echo "== x3 variants"; timeout 240 python tools/microbench.py --x3 --x3-short 2>&1 | grep -v amdgpu.ids
  echo "== suite"; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40
} 2>&1 | tee gpurun_out/r4_a.log

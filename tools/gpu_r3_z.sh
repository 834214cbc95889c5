#!/bin/bash
# Round 3, call Z2: split-operand kernel on the search's internal contractions; workers x batches-per-search with the faster encoder
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== microbench --x3 --x3-decode"
  timeout 300 python tools/microbench.py --x3 --x3-decode 2>&1 | grep -v amdgpu.ids
  for v in "--streams 8 --group 4" "--streams 10 --group 4" "--streams 12 --group 4" "--streams 8 --group 6" "--streams 6 --group 6"; do
    echo "== bench $v"
    timeout 300 python bench.py $v --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2> gpurun_out/r3z.err | tail -1 | cut -c1-140
  done
} 2>&1 | tee gpurun_out/r3_z2.log

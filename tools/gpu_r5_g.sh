#!/bin/bash
# Round 5, visit G: the full GPU suite on the pruned tree (ABI 10, 15 knobs), then the headline and the latency probe as a
# regression check of the prune.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== suite"; timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -12
  echo "== latency"; timeout 120 python tools/latency_probe.py --overlap 3 2>&1 | grep "latency probe"
  echo "== bench (12 steps, no extras)"
  for rep in 1 2; do timeout 120 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5g.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['config']['gpu_memory_reserved_gb'])"; done
} 2>&1 | tee gpurun_out/r5_g.log

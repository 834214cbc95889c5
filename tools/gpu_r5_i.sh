#!/bin/bash
# Round 5, visit I (short): search workspaces kept per stream across a transcriber's jobs; an outgrown buffer trimmed (SBK_WS_TRIM)
# or left in its pool -- reserved memory and the headline on ONE box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for rep in 1 2 3; do for trim in 1 0; do echo "SBK_WS_TRIM=$trim"; SBK_WS_TRIM=$trim timeout 120 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5i.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['config']['gpu_memory_reserved_gb'])"; done; done
} 2>&1 | tee gpurun_out/r5_i.log

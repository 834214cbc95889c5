#!/bin/bash
# Round 5, visit I (short): search workspaces kept per stream across a transcriber's jobs -- reserved memory and the headline.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/ -q -m gpu -k "concurrent or in_flight or grouped or headline_mode or handoffs" 2>&1 | tail -3
  for rep in 1 2; do timeout 120 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5i.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['config']['gpu_memory_reserved_gb'])"; done
} 2>&1 | tee gpurun_out/r5_i.log

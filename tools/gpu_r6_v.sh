#!/bin/bash
# Round 6, visit V: 4 against 6 workers at the driver's 20 + 5 steps, twice (reserved memory 78 against 100 GB).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6v.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  for rep in 1 2; do for w in 4 6 5; do echo "-- $w x 4, 20 steps, warmup 5"; bench --steps 20 --warmup 5 --streams $w --group 4; done; done
} 2>&1 | tee gpurun_out/r6_v.log

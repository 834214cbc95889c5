#!/bin/bash
# Round 6, visit S: batches per encoder pass of a worker's group (0 = one pass per batch, 2, 4 = the whole group): headline and reserved memory.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6s.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  for rep in 1 2; do for ge in 0 2 1000; do echo "-- group-encoder $ge, 12 steps"; bench --steps 12 --group-encoder $ge; done; done
  for ge in 0 2 1000; do echo "-- group-encoder $ge, 20 steps, warmup 5"; bench --steps 20 --warmup 5 --group-encoder $ge; done
} 2>&1 | tee gpurun_out/r6_s.log

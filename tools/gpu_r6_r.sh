#!/bin/bash
# Round 6, visit R: one encoder pass over the rows of a group's four batches (--group-encoder) against one pass per batch, with this
# round's kernels; 12 and 20 steps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6r.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  for rep in 1 2; do
    echo "-- per batch, 12 steps"; bench --steps 12
    echo "-- group encoder, 12 steps"; bench --steps 12 --group-encoder
  done
  echo "-- per batch, 20 steps, warmup 5"; bench --steps 20 --warmup 5
  echo "-- group encoder, 20 steps, warmup 5"; bench --steps 20 --warmup 5 --group-encoder
} 2>&1 | tee gpurun_out/r6_r.log

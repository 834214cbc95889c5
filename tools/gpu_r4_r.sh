#!/bin/bash
# Round 4, visit R (measurement only): fewer, larger grouped searches -- is the headline bound by the kernels that fill
# the chip (then 2 x 16 ~ 8 x 4) or by the overlap of the small ones (then it drops)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { echo "== $*"; timeout 70 python bench.py --steps 16 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>gpurun_out/r4r.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; tail -2 gpurun_out/r4r.err | cut -c1-200; }
{
run --streams 2 --group 16
run --streams 1 --group 32
run --streams 4 --group 8
} 2>&1 | tee gpurun_out/r4_r.log

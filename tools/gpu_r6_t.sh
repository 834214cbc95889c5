#!/bin/bash
# Round 6, visit T: schedule switches again with the group encoder on: CTC scorer on the helper stream of each worker's search
# (--overlap-ctc 3 / 1), workers x batches per search.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6t.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  echo "-- default"; bench
  echo "-- overlap-ctc 3"; bench --overlap-ctc 3
  echo "-- overlap-ctc 1"; bench --overlap-ctc 1
  echo "-- 6 x 4"; bench --streams 6 --group 4
  echo "-- 6 x 6"; bench --streams 6 --group 6
  echo "-- 10 x 4"; bench --streams 10 --group 4
  echo "-- 8 x 3"; bench --streams 8 --group 3
  echo "-- default"; bench
  echo "-- no search priority"; bench --no-search-priority
} 2>&1 | tee gpurun_out/r6_t.log

#!/bin/bash
# Round 5, visit H: counters and kernel statistics at the final kernels (no kernel source changes after visit G's full suite),
# then two schedule questions the round's kernel mix re-opens: the allocator (expandable segments) and workers x batches per search.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 120 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5h.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  echo "== counters (FETCH_SIZE / WRITE_SIZE / MFMA busy: three separate --pmc passes)"; bash tools/run_pmc_r5.sh 2>&1 | tail -40
  echo "== kernel statistics, single stream, 4 steps"
  (cd /tmp && rm -rf /tmp/kst && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --streams 1 --steps 4 --warmup 1 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>&1 | tail -1 | cut -c1-200)
  f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_final_kernel_stats_single_stream.csv && head -16 "$f" | cut -c1-220
  f=$(find /tmp/kst -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_final_domain_stats_single_stream.csv
  echo "== allocator: default vs expandable segments"
  bench
  PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True bench
  echo "== workers x batches per search"
  bench --streams 10 --group 4
  bench --streams 6 --group 6
  bench --streams 12 --group 3
  bench --streams 8 --group 4
} 2>&1 | tee gpurun_out/r5_h.log

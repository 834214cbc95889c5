#!/bin/bash
# Round 6, visit A (VERDICT r5 "do this" 1): the ring cross-attention inside oracle comparisons on the GPU at the geometry and
# knobs it ships with; kernel statistics (rocprofv3 --kernel-trace --stats) of the bench at HEAD; fabric-traffic / MFMA-busy
# counters of gemm_x3r / gemm_nt_x3p and of the decode step's memory-bound kernels as shipped; then the first A/B of
# non-temporal loads on the streamed-once data (knob 53: 1 = ring K tiles, 2 = ring V tiles, 4 = CTC posteriors).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 150 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6a.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  echo "== ring cross-attention tests (emulator-only cases of round 5 now on the GPU; Conformer-L geometry under the default knobs)"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_full_size_gpu.py -q -m gpu -k "ring" 2>&1 | tail -6
  echo "== kernel statistics, single stream, 4 steps"
  (cd /tmp && rm -rf /tmp/kst && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --streams 1 --steps 4 --warmup 1 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>&1 | tail -1 | cut -c1-200)
  f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_a_kernel_stats_single_stream.csv && head -20 "$f" | cut -c1-220
  f=$(find /tmp/kst -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_a_domain_stats_single_stream.csv
  echo "== counters: contraction kernels (FETCH_SIZE / WRITE_SIZE / MFMA busy: three separate --pmc passes)"; bash tools/run_pmc_r6.sh 2>&1 | tail -40
  echo "== counters: the decode step's memory-bound kernels"; bash tools/run_pmc_r6_decode.sh 2>&1 | tail -8
  echo "== non-temporal loads, decode probe (one stream, HIP events per kernel class)"
  for k in 0 1 2 3 4 7; do echo "-- knob 53=$k"; timeout 120 python tools/decode_probe.py --steps 24 --reps 2 --report --knob 53=$k 2>&1 | head -9; done
  echo "== non-temporal loads, bench (8 workers)"
  for k in 0 7 3 4 0 7; do echo "-- knob 53=$k"; bench --knob 53=$k; done
} 2>&1 | tee gpurun_out/r6_a.log

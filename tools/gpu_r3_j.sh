#!/bin/bash
# Round 3, call J: who ends the idle gaps of the eight-worker timed region (kernel trace), 8 vs 12 workers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for W in 8 12; do
    echo "== kernel trace, $W workers"
    (cd /tmp && rm -rf /tmp/trJ && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trJ -o t -- python $OLDPWD/bench.py --steps 12 --warmup 2 --streams $W --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 > $OLDPWD/gpurun_out/r3j_bench_$W.json 2> $OLDPWD/gpurun_out/r3j_bench_$W.err)
    tail -1 gpurun_out/r3j_bench_$W.json | cut -c1-160
    f=$(find /tmp/trJ -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/trace_overlap.py "$f" 0.3 0.9 | cut -c1-420 && python tools/trace_gaps.py "$f" 0.3 0.9
  done
} 2>&1 | tee gpurun_out/r3_j.log

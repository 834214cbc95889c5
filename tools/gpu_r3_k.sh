#!/bin/bash
# Round 3, call K: 64-wide persistent tiles on the decode-step shapes (tests, microbench, in-situ decode probe) and the
# idle time of the eight-worker timed region cut by marker kernels (SBK_TRACE_MARK=1).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== stream-K tests"
  timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "stream_k" 2>&1 | tail -3
  echo "== microbench --sk64"
  timeout 600 python tools/microbench.py --sk64
  echo "== decode probe, in situ"
  for kn in "25=0" "25=512" "25=512 --knob 26=8" "25=512 --knob 26=32" "25=256"; do
    echo "-- knob $kn"
    timeout 300 python tools/decode_probe.py --steps 24 --reps 2 --report --knob $kn 2>&1 | head -14
  done
  echo "== kernel trace, 8 workers, markers"
  (cd /tmp && rm -rf /tmp/trK && SBK_TRACE_MARK=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trK -o t -- python $OLDPWD/bench.py --steps 12 --warmup 2 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 > $OLDPWD/gpurun_out/r3k_bench.json 2> $OLDPWD/gpurun_out/r3k_bench.err)
  tail -1 gpurun_out/r3k_bench.json | cut -c1-200
  f=$(find /tmp/trK -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py "$f" | cut -c1-600 && python tools/trace_gaps.py "$f"
} 2>&1 | tee gpurun_out/r3_k.log

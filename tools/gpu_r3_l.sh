#!/bin/bash
# Round 3, call L: group encoder A/B with the persistent GEMM (driver command), idle time of the timed region cut by markers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== bench --group-encoder (driver defaults otherwise)"
  timeout 900 python bench.py --group-encoder --no-extras --no-cpu-baseline --latency-runs 0 > gpurun_out/r3l_bench_groupenc.json 2> gpurun_out/r3l_bench_groupenc.err
  tail -1 gpurun_out/r3l_bench_groupenc.json | cut -c1-1500
  echo "== kernel trace, 8 workers, markers (up to 3 tries: rocprofv3 crashes now and then with 8 launching threads)"
  for try in 1 2 3; do
    (cd /tmp && rm -rf /tmp/trL && SBK_TRACE_MARK=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trL -o t -- python $OLDPWD/bench.py --steps 12 --warmup 2 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 > $OLDPWD/gpurun_out/r3l_bench.json 2> $OLDPWD/gpurun_out/r3l_bench.err)
    f=$(find /tmp/trL -name "*kernel_trace.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      tail -1 gpurun_out/r3l_bench.json | cut -c1-200
      python tools/trace_overlap.py "$f" | cut -c1-700; python tools/trace_gaps.py "$f"
      break
    fi
    echo "try $try: no trace"
  done
} 2>&1 | tee gpurun_out/r3_l.log

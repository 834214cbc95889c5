#!/bin/bash
# Round 3, call M: group encoder with 6 / 10 / 12 workers; idle time of the timed region cut by markers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for W in 6 10 12; do
    echo "== bench --group-encoder --streams $W"
    timeout 600 python bench.py --group-encoder --streams $W --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2> gpurun_out/r3m_$W.err | tail -1 | cut -c1-170
  done
  echo "== bench --streams 10 (no group encoder)"
  timeout 600 python bench.py --streams 10 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2> gpurun_out/r3m_p10.err | tail -1 | cut -c1-170
  echo "== kernel trace, 8 workers, markers"
  for try in 1 2 3; do
    (cd /tmp && rm -rf /tmp/trM && SBK_TRACE_MARK=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trM -o t -- python $OLDPWD/bench.py --steps 12 --warmup 2 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 > $OLDPWD/gpurun_out/r3m_bench.json 2> $OLDPWD/gpurun_out/r3m_bench.err)
    f=$(find /tmp/trM -name "*kernel_trace.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      tail -1 gpurun_out/r3m_bench.json | cut -c1-200
      python tools/trace_overlap.py "$f" | cut -c1-900; python tools/trace_gaps.py "$f"
      break
    fi
    echo "try $try: no trace"
  done
} 2>&1 | tee gpurun_out/r3_m.log

#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
{
  if [ "${TESTS:-1}" = "1" ]; then
  echo "== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
  fi
  echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-16} --warmup 1 --verbose ${BENCH_ARGS:-} 2>&1 | tail -12
} > gpurun_out/round.log 2>&1
if [ "${PROFILE:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python "$OLDPWD/bench.py" --steps ${BENCH_STEPS:-16} --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --latency-runs 0 > "$OLDPWD/gpurun_out/rocprof_run.log" 2>&1)
  find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/ \; 2>/dev/null
  ls -la /tmp/prof >> gpurun_out/rocprof_run.log 2>&1; find /tmp/prof | head -20 >> gpurun_out/rocprof_run.log
fi
cat gpurun_out/round.log

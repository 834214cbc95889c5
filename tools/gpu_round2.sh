#!/bin/bash
# Round-2 GPU visit: parity tests, the new bench (default + knobs), optional multi-stream kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  if [ "${TESTS:-1}" = "1" ]; then
    echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
    echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
  fi
  if [ "${BENCH:-1}" = "1" ]; then
    echo "== bench"; timeout 1200 python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} --verbose ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
  fi
  if [ -n "${FORCE_DIST:-}" ]; then
    echo "== bench, RCCL leg forced on one GPU"
    SBK_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 > gpurun_out/bench_dist.json 2> gpurun_out/bench_dist.err; tail -3 gpurun_out/bench_dist.err; cat gpurun_out/bench_dist.json
  fi
  if [ -n "${TRACE_ARGS:-}" ]; then
    echo "== multi-stream kernel trace: $TRACE_ARGS"
    (cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 $TRACE_ARGS > "$OLDPWD/gpurun_out/trace_run.log" 2>&1)
    f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); echo "trace file: $f"; tail -2 gpurun_out/trace_run.log
    [ -n "$f" ] && python tools/trace_overlap.py "$f" | tee gpurun_out/trace_overlap.json
  fi
} 2>&1 | tee gpurun_out/round2.log
if [ -n "${STATS_ARGS:-}" ]; then
  # single-stream rocprofv3 kernel statistics (true kernel durations, no HIP-event overhead)
  (cd /tmp && rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o st -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 --streams 1 $STATS_ARGS > "$OLDPWD/gpurun_out/stats_run.log" 2>&1)
  d=$(find /tmp/prof2 -name "*domain_stats.csv" | head -1); [ -n "$d" ] && cp "$d" gpurun_out/${STATS_NAME:-stats}_domain_stats.csv
  f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; tail -1 gpurun_out/stats_run.log
  [ -n "$f" ] && cp "$f" gpurun_out/${STATS_NAME:-stats}_kernel_stats.csv && head -40 "$f" | cut -c1-200
fi
if [ -n "${MICRO:-}" ]; then
  for m in $MICRO; do echo "== microbench $m"; timeout 600 python tools/microbench.py $m 2>&1 | tail -80; done | tee gpurun_out/micro2.log
fi

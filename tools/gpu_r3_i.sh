#!/bin/bash
# Round 3, call I: fp8 conversion probe, the whole GPU suite, the driver's bench command, its rocprofv3 kernel stats
# (single stream), the strong-scaling mode on one GPU.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== fp8 probe"; timeout 200 python tools/fp8_probe.py 2>&1 | grep -v amdgpu.ids | tail -22
  echo "== pytest -m gpu (all)"
  SECONDS=0
  timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -18
  echo "suite seconds: $SECONDS"
  echo "== bench (driver command)"
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
  tail -3 gpurun_out/r3_final_bench.err; cat gpurun_out/r3_final_bench.json | cut -c1-400
  echo "== strong-scaling mode, one GPU"
  timeout 600 python bench.py --gpus 1 --job-utts 1000 --warmup 1 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>/dev/null | cut -c1-700
} 2>&1 | tee gpurun_out/r3_i.log
(cd /tmp && rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o st -- python "$OLDPWD/bench.py" --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 --streams 1 > "$OLDPWD/gpurun_out/r3_stats_run.log" 2>&1)
d=$(find /tmp/prof2 -name "*domain_stats.csv" | head -1); [ -n "$d" ] && cp "$d" gpurun_out/r03_final_domain_stats_single_stream.csv
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; tail -1 gpurun_out/r3_stats_run.log | cut -c1-300
[ -n "$f" ] && cp "$f" gpurun_out/r03_final_kernel_stats_single_stream.csv && head -14 "$f" | cut -c1-160

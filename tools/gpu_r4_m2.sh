#!/bin/bash
# Round 4, visit M2: the profiles of the round's final kernels -- rocprofv3 kernel stats of the single-stream bench, the
# counters passes of the two contraction kernels, the decode-step trace, and the MX fp8 MFMA probe (facts for the next round)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== rocprofv3 --kernel-trace --stats: bench.py --streams 1 --steps 4"
  (cd /tmp && rm -rf /tmp/prof_m && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o s -- python $OLDPWD/bench.py --streams 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-roofline --latency-runs 0 > $OLDPWD/gpurun_out/r4m_prof.log 2>&1)
  f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4m_kernel_stats_single_stream.csv
  f=$(find /tmp/prof_m -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4m_domain_stats_single_stream.csv
  tail -1 gpurun_out/r4m_prof.log | cut -c1-200
  head -14 gpurun_out/r4m_kernel_stats_single_stream.csv | cut -c1-170
  echo "== counters (FETCH_SIZE / WRITE_SIZE / MFMA busy: separate passes)"
  timeout 900 bash tools/run_pmc_r4.sh 2>&1 | tail -50
  echo "== decode trace (final kernels)"
  (cd /tmp && rm -rf /tmp/dtr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $OLDPWD/tools/decode_probe.py --steps 16 --reps 3 2>&1 | grep "decode probe"); f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 48 | head -22
  echo "== MX fp8 MFMA probe"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mx_probe tools/mx_probe.hip 2>/dev/null && timeout 120 /tmp/mx_probe
} 2>&1 | tee gpurun_out/r4_m2.log

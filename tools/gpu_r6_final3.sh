#!/bin/bash
# Round 6, final visit 3 (after gemm_lp256, the base-2 softmax and the x3p epilogue): the full GPU suite, smoke(), the driver's own bench command at HEAD; rocprofv3 kernel statistics of the single-stream bench (CTC scorer on
# the search's own stream, like the line's instrumented repetition); the contraction kernels' counters at HEAD (gemm_x3r's load
# schedule changed this round).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  echo "== suite"; timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -4
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
  echo "== driver bench"
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final3_bench.json 2> gpurun_out/r06_final3_bench.err
  tail -c 600 gpurun_out/r06_final3_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final3_bench.json").read().strip().splitlines()[-1])
keys = ["value", "ms_per_step", "value_batch128", "value_encoder_gemms_bf16", "value_fp32_mfma_contractions", "p50_latency_ms", "p50_latency_ms_by_mode",
        "decode_step_ms", "launches_per_decode_step"]
print({k: d.get(k) for k in keys})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "launches", "share_of_gpu_time", "traffic")})
print("top3", [(e["kernel"], e["frac"], e["avg_launch_ms"], e.get("traffic_over_algorithmic")) for e in d["roofline_top3"]])
print("checks", d["parity_check"]["ids_equal"], d["determinism_check"]["ids_equal"], d.get("token_error_rate_vs_oracle_12x10s_peaked_heads", {}).get("ids_equal"),
      d.get("token_error_rate_vs_oracle_12x10s_peaked_heads", {}).get("decode_kernels_run"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), "mem", d["config"]["gpu_memory_reserved_gb"])
print("whisper", {k: v.get("ms_per_batch") for k, v in d.get("config5_whisper_encoder", {}).items() if isinstance(v, dict)})
print("breakdown", d.get("kernel_breakdown_ms"))
PY
  echo "== kernel statistics, single stream, 4 steps, CTC scorer on the search's stream"
  (cd /tmp && rm -rf /tmp/kst && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --streams 1 --steps 4 --warmup 1 --overlap-ctc 0 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>&1 | tail -1 | cut -c1-200)
  f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_final3_kernel_stats_single_stream.csv && head -14 "$f" | cut -c1-200
  f=$(find /tmp/kst -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_final3_domain_stats_single_stream.csv
  echo "== counters: contraction kernels at HEAD"; bash tools/run_pmc_r6.sh 2>&1 | tail -40
  for t in fetch write mfma; do cp gpurun_out/pmc_r6_$t.csv gpurun_out/r06_final3_pmc_x3r_x3p_$t.csv; done
} 2>&1 | tee gpurun_out/r6_final3.log

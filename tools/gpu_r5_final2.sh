#!/bin/bash
# Round 5, last visit: the driver's command `python bench.py` at HEAD (the full GPU suite ran at the same kernel sources and
# native.py in tools/gpu_r5_final.sh; only bench.py changed since: the latency leg in a process of its own).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  echo "== bench (python bench.py)"
  timeout 1500 python bench.py 2>gpurun_out/r5final2.err | tail -1 > gpurun_out/r05_final_bench.json
  python -c "
import json; d = json.load(open('gpurun_out/r05_final_bench.json'))
for k in ('value', 'ms_per_step', 'value_batch128', 'value_encoder_gemms_bf16', 'bf16_vs_fp32_token_error_rate_percent', 'value_fp32_mfma_contractions', 'p50_latency_ms', 'p50_latency_ms_by_mode', 'decode_step_ms', 'launches_per_decode_step', 'parity_check', 'determinism_check', 'cpu_baseline', 'config1_encoder_S'): print(k, d.get(k))
print(d['config'].get('gpu_memory_reserved_gb')); print({k: v for k, v in d['config'].items() if 'error' in k})
print(d.get('roofline')); print(d.get('roofline_top3')); print(d.get('roofline_end_to_end')); w = d.get('config5_whisper_encoder') or {}; print({k: v for k, v in w.items() if isinstance(v, dict)})"
} 2>&1 | tee gpurun_out/r5_final2.log

#!/bin/bash
# Round 5, visit C: the persistent few-row step with every phase's loads in flight at once (second version), the one-run
# cross-attention as the default, the bench's secondary legs in processes of their own, the RCCL leg on one GPU.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  echo "== tests"; timeout 600 python -m pytest tests/ -q -m gpu -k "persistent_few_row or cross_attention or concurrent_transcriber_matches or conformer_l_decoder or properties_at_bench_shape or greedy_beam1" 2>&1 | tail -6
  echo "== latency"
  timeout 120 python tools/latency_probe.py --knob 47=0 --overlap 3 2>&1 | grep "latency probe"
  for g in 64 128; do for ov in 0 3; do timeout 120 python tools/latency_probe.py --knob 48=$g --overlap $ov 2>&1 | grep "latency probe"; done; done
  timeout 120 python tools/latency_probe.py --overlap 3 --report 2>&1 | grep -A12 "latency probe"
  timeout 120 python tools/latency_probe.py --seconds 28 --overlap 3 2>&1 | grep "latency probe"
  echo "== kernel trace of the persistent search"
  (cd /tmp && rm -rf /tmp/ltr && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ltr -o t -- python $R/tools/latency_probe.py --runs 3 2>&1 | grep "latency probe")
  f=$(find /tmp/ltr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
  echo "== RCCL leg on one GPU (process group, scatter_object_list, gather, barriers under RCCL)"
  SBK_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5c.err | tail -1 > gpurun_out/r5c_forced_dist_bench.json
  python -c "
import json; d = json.load(open('gpurun_out/r5c_forced_dist_bench.json')); print('forced dist:', d['value'], 'rccl_world', d['rccl_world'], d['per_rank'])"
  echo "== bench with the secondary legs in their own processes (8 steps)"
  timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>>gpurun_out/r5c.err | tail -1 > gpurun_out/r5c_bench.json
  python -c "
import json; d = json.load(open('gpurun_out/r5c_bench.json'))
for k in ('value', 'value_batch128', 'value_encoder_gemms_bf16', 'bf16_vs_fp32_token_error_rate_percent', 'value_fp32_mfma_contractions', 'p50_latency_ms', 'p50_latency_ms_by_mode', 'decode_step_ms', 'launches_per_decode_step', 'parity_check', 'determinism_check'): print(k, d.get(k))
print(d['config'].get('gpu_memory_reserved_gb')); print({k: v for k, v in d['config'].items() if 'error' in k})
print(d.get('decode_step_probe')); print(d.get('roofline')); print(d.get('config5_whisper_encoder'))"
} 2>&1 | tee gpurun_out/r5_c.log

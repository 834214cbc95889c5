#!/bin/bash
# Round 4, visit G: fused-scoring bit-identity after the contraction-proof arithmetic, LayerNorm -> panel with full-line
# stores, kernel trace of a single-stream decoding step (true durations + gaps), bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"; timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -p no:cacheprovider -x -k "layernorm_x3p or fused_scoring or panel_route or input_normalization or golden_model" 2>&1 | tail -12
  echo "== microbench"; timeout 300 python tools/microbench.py --ln-x3p 2>&1 | grep -v amdgpu.ids
  echo "== decode trace"
  (cd /tmp && rm -rf /tmp/dtr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $OLDPWD/tools/decode_probe.py --steps 16 --reps 3 2>&1 | grep "decode probe")
  f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 48
  echo "== bench default"; timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0 2>/dev/null | tee gpurun_out/r4_g_bench.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:14]}); print(json.dumps(d.get('roofline_top3'))); print(d['config'].get('gpu_memory_reserved_gb'))"
} 2>&1 | tee gpurun_out/r4_g.log

#!/bin/bash
# Round 6, visit X: the base-2 softmax (one v_exp_f32 per score, rescale skipped when no lane has a new maximum) in RelPosMHAXL's fp32
# kernel (knob 60) -- tests through it, isolated A/B, headline A/B -- and the Whisper encoder at all 32 layers with the bf16 attention of visit W.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6x.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_check']['ids_equal'], d['determinism_check']['ids_equal'])
except Exception as e: print('no result', e)"; }
{
  echo "== tests through the attention kernels"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_streaming.py tests/test_full_size_gpu.py tests/test_whisper.py tests/test_model_parity.py -q -m gpu -x -k "relpos or encoder or streaming or whisper or bf16" 2>&1 | tail -3
  echo "== isolated"
  timeout 300 python tools/microbench.py --attn 2>&1 | grep -v amdgpu.ids
  echo "== whisper, 32 layers"
  timeout 300 python tools/whisper_probe.py --layers 32 --prec bf16,fp8 2>&1 | grep -v amdgpu.ids | head -40
  for rep in 1 2; do for k in 0 1; do echo "-- headline, knob 60 = $k"; bench --steps 20 --warmup 5 --knob 60=$k; done; done
} 2>&1 | tee gpurun_out/r6_x.log

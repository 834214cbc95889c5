#!/bin/bash
# Round 4, visit E: the step's scoring as one pass per hypothesis row (score_topk_row_kernel + beam_merge_update_kernel):
# GPU parity, decode-step times with / without, bench A/B, workers x group sweep at the new state
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:10]})"; }
{
  echo "== tests"; timeout 900 python -m pytest tests/test_model_parity.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "fused_scoring or golden_model or grouped_search or device_side_step or wide_beam or headline or decoder_logprobs or lm_scorer or greedy or properties or transformerlm" 2>&1 | tail -6
  echo "== decode probe, fused scoring"; timeout 300 python tools/decode_probe.py --steps 16 --report 2>&1 | grep -v amdgpu.ids
  echo "== decode probe, separate kernels"; timeout 300 python tools/decode_probe.py --steps 16 --report --knob 40=0 2>&1 | grep -v amdgpu.ids | head -3
  echo "== decode probe, no events, fused / separate"; timeout 300 python tools/decode_probe.py --steps 16 --reps 3 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/decode_probe.py --steps 16 --reps 3 --knob 40=0 2>&1 | grep -v amdgpu.ids
  echo "== bench separate kernels"; timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0 --knob 40=0 2>/dev/null | short
  echo "== bench fused"; timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0 2>/dev/null | short
  for sg in "6 6" "4 8" "12 3" "8 6"; do set -- $sg
    echo "== bench fused, streams $1 group $2"; timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --latency-runs 0 --streams $1 --group $2 2>/dev/null | short
  done
} 2>&1 | tee gpurun_out/r4_e.log

#!/bin/bash
# Round 5, visit M: cross-attention of a step on a register ring (cross_attn_ring_kernel, knob 4 = 6 / 8 / 9 = 4 / 6 / 3 tiles deep)
# against the LDS-DMA kernel (knob 4 = 7 default; knob 8 = 4: its 256-workgroup form).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 150 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5m.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d.get('parity'))
except Exception as e: print('no result', e)"; }
{
  timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "cross_attention_lds_dma" 2>&1 | tail -2
  for k in "4=7" "4=6" "4=8" "4=9" "8=4"; do echo "== decode probe, knob $k"; timeout 90 python tools/decode_probe.py --steps 16 --reps 3 --knob $k --report 2>&1 | grep -E "decode probe|cross_"; done
  for rep in 1 2; do for k in "4=7" "4=6"; do echo "== bench, knob $k"; bench --knob $k; done; done
} 2>&1 | tee gpurun_out/r5_m.log

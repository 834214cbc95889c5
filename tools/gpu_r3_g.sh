#!/bin/bash
# Round 3, call G: tile size / workgroup count of the LDS-DMA cross-attention step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "cross_attention_lds_dma" 2>&1 | tail -3
  for K in "4=5 --knob 8=1" "4=5 --knob 8=0" "4=5 --knob 8=2" "4=6 --knob 8=1" "4=6 --knob 8=0" "4=6 --knob 8=2"; do
    echo "== decode probe, knob $K"
    timeout 300 python tools/decode_probe.py --steps 16 --report --knob $K 2>&1 | grep -E "decode probe|cross_attn"
  done
  echo "== single batch (32 utterances)"
  for K in "4=0" "4=5 --knob 8=0" "4=6 --knob 8=0" "4=6 --knob 8=2"; do
    timeout 300 python tools/decode_probe.py --steps 16 --batches 1 --report --knob $K 2>&1 | grep -E "decode probe|cross_attn"
  done
} 2>&1 | tee gpurun_out/r3_g.log

#!/bin/bash
# Round 4, visit T (staged change, see gpu_r4_s.sh): a decoding step of a 4 x 32-utterance grouped search with the decoder's
# LayerNorms inside the projections (knob 45 = 1) and as launches of their own (0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
  for k in 0 1 0 1; do echo "== knob 45 = $k"; timeout 25 python tools/decode_probe.py --steps 16 --reps 3 --knob 45=$k 2>&1 | grep "decode probe"; done
} 2>&1 | tee gpurun_out/r4_t.log

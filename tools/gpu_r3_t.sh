#!/bin/bash
# Round 3, call T: the bench's bf16 leg (Conformer-L, eight workers) with bf16 activations from 16 000 rows / from 256 rows / off.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for v in "SBK_BF16A_MIN_ROWS=16000" "SBK_BF16_ACTIVATIONS=0" "SBK_BF16A_MIN_ROWS=256"; do
    echo "== bench --precision bf16, $v"
    env $v timeout 600 python bench.py --precision bf16 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2> gpurun_out/r3t.err | tail -1 | cut -c1-170
  done
} 2>&1 | tee gpurun_out/r3_t.log

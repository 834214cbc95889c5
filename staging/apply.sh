#!/bin/bash
# Apply the staged series on a clean tree, rebuild both libraries, run the CPU suite.  (Next round, first thing.)
set -eu
cd "$(dirname "$0")/.."
test -z "$(git status --porcelain -- speechbrain_amd include tests tools INTEGRATION.md DESIGN.md)" || { echo "tree not clean"; exit 1; }
git apply staging/r5_decoder_layernorm_in_projections.patch
git apply staging/r5_2_relpos_attention_split_operands.patch
python -m speechbrain_amd.csrc.build | tail -1
python tools/kernel_emu/build_emu.py | tail -1
python -m pytest tests/ -x -q -m "not gpu" | tail -3
echo "then: gpurun -- 'bash tools/gpu_r5_a.sh', and git rm -r staging/ once the series is committed"

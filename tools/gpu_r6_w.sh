#!/bin/bash
# Round 6, visit W: the Whisper encoder's bf16 attention kernel with the base-2 softmax on raw scores and the lazy rescale (configs[4]):
# tests, per-kernel times of the fp8 / bf16 pipelines (tools/whisper_probe.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py tests/test_full_size_gpu.py -q -m gpu -x -k "bf16 or whisper" 2>&1 | tail -3
  timeout 300 python tools/whisper_probe.py --layers 8 --prec bf16,fp8 2>&1 | grep -v amdgpu.ids | head -40
} 2>&1 | tee gpurun_out/r6_w.log

#!/bin/bash
# Round 4, visit N: the fp8 activation pipeline of the Whisper encoder (sbk_layernorm_fp8o / sbk_quant_rows_fp8 /
# sbk_gemm_nt_fp8a on v_mfma_f32_32x32x64_f8f6f4): parity of the kernels, tolerances at the large-v3 shape, per-kernel time
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"; timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -s -k "fp8 or whisper_large_v3 or bf16_activation" 2>&1 | grep -v "^$" | tail -12
  echo "== whisper probe, 32 layers, 8 x 30 s"; timeout 600 python tools/whisper_probe.py --layers 32 --prec fp32,bf16,fp8 2>&1 | grep -v amdgpu.ids
  echo "== the round-3 fp8 path (fp32 activations, per-tensor scales)"; SBK_FP8_ACTIVATIONS=0 timeout 300 python tools/whisper_probe.py --layers 32 --prec fp32,fp8 2>&1 | grep -v amdgpu.ids | grep -E "^fp8"
} 2>&1 | tee gpurun_out/r4_n.log

#!/bin/bash
# Round 4, visit O (the round's last): the fp8 pipeline's targeted tests and timing first; only if they pass, the whole GPU
# suite at this HEAD's kernels and the driver's bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== fp8 tests"; timeout 600 python -m pytest tests/test_kernels.py tests/test_whisper.py -q -m gpu -p no:cacheprovider -x -k "fp8" 2>&1 | tail -6
  if [ "${PIPESTATUS[0]}" != "0" ]; then echo "fp8 tests failed: stopping here"; exit 1; fi
  echo "== whisper probe, 32 layers, 8 x 30 s"; timeout 600 python tools/whisper_probe.py --layers 32 --prec fp32,bf16,fp8 2>&1 | grep -v amdgpu.ids
  echo "== suite"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15
  echo "== bench (python bench.py)"; timeout 900 python bench.py 2> gpurun_out/r4o_bench.err | tee gpurun_out/r4o_bench.json | cut -c1-1500
  tail -3 gpurun_out/r4o_bench.err
} 2>&1 | tee gpurun_out/r4_o.log

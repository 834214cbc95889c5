#!/bin/bash
# Round 3, final visit 3: the last-arriver reductions on the GPU (parity tests through them) and a bench run with them on
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 60 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "tiled_splitk or cross_attention_lds_dma or grouped_search or golden_model or split_operand or device_side_step or transformerlm" 2>&1 | tail -3
  echo "== bench, reductions by the last arriver (default)"
  timeout 60 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2> gpurun_out/r3h.err | tail -1 | cut -c1-140
  echo "== bench, separate reduce / merge launches"
  timeout 40 python bench.py --knob 36=0 --knob 37=0 --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>> gpurun_out/r3h.err | tail -1 | cut -c1-140
} 2>&1 | tee gpurun_out/r3_f3.log

#!/bin/bash
# Round 4, visit D: first run of the pre-split panel contraction (csrc/gemm_x3p.hip): kernel tests, microbench, bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"; timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "x3p" -p no:cacheprovider 2>&1 | tail -15
  echo "== microbench"; timeout 300 python tools/microbench.py --x3p --x3p-short 2>&1 | grep -v amdgpu.ids
  echo "== bench x3p off"; SBK_X3P=0 timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: v for k, v in list(d['kernel_breakdown_ms'].items())[:6]})"
  echo "== bench x3p on"; SBK_X3P=1 timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: v for k, v in list(d['kernel_breakdown_ms'].items())[:8]}); print(json.dumps(d['roofline']))"
} 2>&1 | tee gpurun_out/r4_d.log

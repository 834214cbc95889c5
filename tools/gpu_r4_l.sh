#!/bin/bash
# Round 4, visit L: schedule knobs at the new kernel mix (x3r decode projections): CTC scoring on the helper stream beside
# the decoder GEMMs inside the workers, workers x batches per search; 16-step runs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_l_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'))"; }
B="python bench.py --warmup 1 --no-cpu-baseline --no-extras --no-roofline --latency-runs 0"
{
  echo "== tests (x3r full-size oracle comparison, no-workspace error)"; timeout 600 python -m pytest tests/test_kernels.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "no_stream_workspace or decoder_logprobs or gemm_x3r" 2>&1 | tail -4
  echo "== default (8 x 4, overlap 0)"; timeout 400 $B 2>/dev/null | short
  echo "== overlap-ctc 3"; timeout 400 $B --overlap-ctc 3 2>/dev/null | short
  echo "== overlap-ctc 1"; timeout 400 $B --overlap-ctc 1 2>/dev/null | short
  echo "== 6 x 6"; timeout 400 $B --streams 6 --group 6 2>/dev/null | short
  echo "== 4 x 8"; timeout 400 $B --streams 4 --group 8 2>/dev/null | short
  echo "== 6 x 8"; timeout 400 $B --streams 6 --group 8 2>/dev/null | short
  echo "== default again"; timeout 400 $B 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_l.log

#!/bin/bash
# A/B of tuning knobs on the bench: ARGS="--batch 64" KNOBS="4=0" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
show='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_breakdown_ms"])'
echo "== gpu tests (kernels + parity)"; timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x 2>&1 | tail -3
while read -r line; do
  [ -z "$line" ] && continue
  echo "== bench $line"
  timeout 900 python bench.py --warmup 1 --no-cpu-baseline --latency-runs 0 $line 2>&1 | grep '^{' | tee -a gpurun_out/ab.jsonl | python -c "$show"
done <<< "$AB_CASES"

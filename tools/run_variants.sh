#!/bin/bash
# GPU-box visit: parity tests, then the bench on the widened rows (RoPEMHA encoder, TransformerLM scorer).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
show='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_breakdown_ms"])'
for v in "--attention RoPEMHA" "--lm"; do
  echo "== bench $v"
  timeout 900 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --latency-runs 0 $v 2>&1 | grep '^{' | tee -a gpurun_out/variants.jsonl | python -c "$show"
done

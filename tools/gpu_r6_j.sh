#!/bin/bash
# Round 6, visit J: what the kernels cost when eight workers share the chip -- HIP events around every launch of the timed region
# itself (bench.py --prof-concurrent): per kernel class the time its launches took co-resident with the other workers' kernels,
# and the mean number of kernels in flight; the same with 4 and 2 workers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 250 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 --prof-concurrent "$@" 2>>gpurun_out/r6j.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['concurrent_kernels']
print('value', d['value'], 'workers', d['config']['workers_per_gpu'], 'launches', c['launches'], 'kernel_ms_sum', c['kernel_ms_sum'], 'wall_ms', c['wall_ms'], 'in flight', c['mean_kernels_in_flight'])
for k, v in c['by_class'].items(): print(f'   {k:22s} {v[\"launches\"]:7d} launches {v[\"ms\"]:9.1f} ms {v[\"us_each\"]:8.1f} us each')
"; }
{
  echo "== 8 workers x 4"; run
  echo "== 4 workers x 4"; run --streams 4 --group 4
  echo "== 2 workers x 4"; run --streams 2 --group 4
  echo "== 1 worker x 4 (overlap_ctc 0)"; run --streams 1 --group 4 --overlap-ctc 0
} 2>&1 | tee gpurun_out/r6_j.log

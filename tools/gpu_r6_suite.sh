#!/bin/bash
# Round 6: the full GPU suite and smoke() at the current tree.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== suite"; timeout 1700 python -m pytest tests/ -q -m gpu 2>&1 | tail -8
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r6_suite.log

#!/bin/bash
# Round 5, visit Q: ownership of the tile space by the XCDs.  gemm_x3r: knob 51 = 8 (rounds 4-5a: every XCD fetches all of A) / 4 / 2 /
# 1 column groups, 0 = the split with the fewest bytes across the fabric; gemm_nt_x3p: knob 52 = 1 (row bands) / 2 / 4 column groups.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
probe() { timeout 120 python tools/decode_probe.py --steps 60 --reps 2 --report "$@" 2>&1 | grep -E "decode probe|gemm_x3r|gemm_ln_x3r"; }
{
  timeout 400 python -m pytest tests/test_kernels.py -q -m gpu -x -k "test_gemm_x3p or test_gemm_x3r or test_gemm_ln_x3r" 2>&1 | tail -2
  for k in 8 0 4 2 1 8 0; do echo "== x3r knob 51=$k"; probe --knob 51=$k; done
  echo "== x3r per shape, isolated"
  timeout 200 python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from speechbrain_amd import native as nat
dev = torch.device("cuda:0"); lib = nat.load()
def ev(fn, n=40):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in [(1280, 512, 512), (1280, 1536, 512), (1280, 2048, 512), (1280, 512, 2048), (1280, 5000, 512)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); r = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
    line = f"x3r M={M} N={N} K={K}:"
    for xc in (8, 4, 2, 1, 0):
        lib.sbk_prof_set_knob(51, xc)
        line += f"  xc={xc}: {ev(lambda: nat.gemm_nt_x3r(a, w, b, r)):6.1f} us"
    lib.sbk_prof_set_knob(51, 0)
    print(line, flush=True)
for (M, N, K) in [(12800, 2048, 512), (12800, 512, 2048), (14016, 1536, 512), (14016, 512, 512), (14016, 1024, 512), (24032, 2048, 512), (24032, 1536, 512)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    pa = nat.split_x3p(a)
    line = f"x3p M={M} N={N} K={K}:"
    for cg in (1, 2, 4, 1, 2, 4):
        lib.sbk_prof_set_knob(52, cg)
        t = ev(lambda: nat.gemm_nt_x3p(pa, w), 20)
        line += f"  cg={cg}: {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s"
    lib.sbk_prof_set_knob(52, 0)
    print(line, flush=True)
PY
} 2>&1 | tee gpurun_out/r5_q.log

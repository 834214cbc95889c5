"""Does v_cvt_pk_fp8_f32 (sbk_f32_to_fp8) agree with torch.float8_e4m3fn?  Prints mismatches by magnitude class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechbrain_amd import native as nat

lib = nat.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.cat([torch.randn(1 << 16, generator=g) * s for s in (1e-3, 1e-2, 0.1, 1.0, 30.0, 200.0)]).to(dev)
y = torch.empty(x.numel(), dtype=torch.uint8, device=dev)
assert lib.sbk_f32_to_fp8(nat._p(x), nat._p(y), x.numel(), 1.0, nat._stream(x)) == 0
ref = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
bad = (y != ref).nonzero().flatten()
print("elements", x.numel(), "mismatches", bad.numel())
for i in bad[:12].tolist():
    print(f"  x={float(x[i]):.8g} hw=0x{int(y[i]):02x} ({float(y[i:i+1].view(torch.float8_e4m3fn).float()):.6g}) torch=0x{int(ref[i]):02x} ({float(ref[i:i+1].view(torch.float8_e4m3fn).float()):.6g})")
ax = x.abs()
for lo, hi in ((0, 2 ** -10), (2 ** -10, 2 ** -9), (2 ** -9, 2 ** -6), (2 ** -6, 1), (1, 448), (448, 1e9)):
    m = (ax >= lo) & (ax < hi)
    print(f"  |x| in [{lo:g},{hi:g}): {int(m.sum())} elements, {int(((y != ref) & m).sum())} mismatches")

"""Isolated kernel timings on the MI355X (back-to-back launches between two events)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechbrain_amd import native as nat

dev = torch.device("cuda:0")
nat.load()


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000.0  # us


def gemm_case(M, N, K, slices, iters=200):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    ws = torch.empty(max(1, slices * M * N), device=dev)
    out = torch.empty(M, N, device=dev)
    lib = nat.load()
    us = ctypes.c_float(0)
    rc = lib.sbk_prof_gemm_repeat_f32(nat._p(a), nat._p(w), nat._p(out), M, N, K, nat._p(ws) if slices else None,
                                      ws.numel() if slices else 0, iters, ctypes.byref(us), nat._stream(a))
    assert rc == 0
    print(f"gemm M={M} N={N} K={K} slices={slices}: {us.value:8.2f} us  {2.0*M*N*K/us.value/1e6:7.2f} TFLOP/s", flush=True)


def ctc_case(B, T, V, beam, prefix_len):
    x = torch.log_softmax(torch.randn(B, T, V, device=dev), -1)
    enc_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    last = torch.randint(3, V, (B * beam,), dtype=torch.int32, device=dev)
    psi = torch.empty(B * beam, V, device=dev)
    work = torch.empty(8 * B * beam * T + B * T + B * beam + 4096, device=dev)
    us = ctypes.c_float(0)
    rc = nat.load().sbk_prof_ctc_psi_repeat_f32(nat._p(x), nat._p(enc_len), nat._p(last), nat._p(psi), nat._p(work), B, T,
                                                V, beam, prefix_len, 20, ctypes.byref(us), nat._stream(x))
    assert rc == 0, nat.load().sbk_last_error()
    print(f"ctc_psi B={B} T={T} V={V} beam={beam}: {us.value:8.1f} us  {4.0*B*T*V/us.value/1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    if "--pmc-workload" in sys.argv:  # short, single-stream: the kernels whose HBM traffic is read from PMC counters
        # encoder GEMMs of one Conformer-L layer at the bench's batch (M = 128 utterances x 438 frames)
        for (M, N, K) in [(56064, 1536, 512), (56064, 512, 512), (56064, 2048, 512), (56064, 512, 2048),
                          (56064, 2048, 512), (56064, 512, 2048), (56064, 1024, 512), (56064, 512, 512)]:
            gemm_case(M, N, K, 0, iters=10)
        # decode-step GEMMs at M = 128 x beam 10
        for (M, N, K) in [(1280, 1536, 512), (1280, 512, 512), (1280, 512, 512), (1280, 512, 512), (1280, 2048, 512),
                          (1280, 512, 2048), (1280, 5000, 512)]:
            gemm_case(M, N, K, 8, iters=20)
        ctc_case(128, 440, 5000, 10, 5)
        sys.exit(0)
    if "--attn" in sys.argv:  # encoder attention kernels (RelPosMHAXL / RoPE) in isolation
        import math
        from speechbrain_amd.nnet.attention import PrecomputedRoPESinusoids
        H, Dh = 8, 64
        d = H * Dh
        tab = PrecomputedRoPESinusoids(1024, Dh, torch.float32, dev)
        for (B, T) in [(16, 300), (16, 600), (64, 440)]:
            qkv = torch.randn(B, T, 3 * d, device=dev)
            P = torch.randn(2 * T - 1, d, device=dev)
            u = torch.randn(d, device=dev) * 0.1
            kl = torch.full((B,), T, dtype=torch.int32, device=dev)
            nat.load().sbk_prof_set_knob(60, 0)  # softmax weights through libm's expf (before round 6)
            t_rel0 = timeit(lambda: nat.relpos_attention(qkv, P, u, u, kl, H, 1 / math.sqrt(d)), n=20, warm=3)
            nat.load().sbk_prof_set_knob(60, 1)  # 2^(.) on one v_exp_f32 + the rescale skipped when no lane has a new maximum (default)
            t_rel = timeit(lambda: nat.relpos_attention(qkv, P, u, u, kl, H, 1 / math.sqrt(d)), n=20, warm=3)
            t_rope = timeit(lambda: nat.rope_attention(qkv, tab.cosines, tab.sines, kl, H, 1 / math.sqrt(d)), n=20, warm=3)
            fl = B * H * T * T * Dh
            print(f"attn B={B} T={T}: relpos {t_rel:8.1f} us {6.0*fl/t_rel/1e6:6.1f} TF/s (expf form {t_rel0:8.1f} us) | rope {t_rope:8.1f} us {4.0*fl/t_rope/1e6:6.1f} TF/s", flush=True)
        if "--gemm" not in sys.argv:
            sys.exit(0)
        print("decode-step GEMMs: register-operand 32x32 tiles (skinny) vs LDS-tiled")
        for M in (320, 640, 1280):
            for (N, K) in [(512, 512), (1536, 512), (2048, 512), (512, 2048), (5000, 512)]:
                nat.load().sbk_prof_set_knob(2, 0)
                gemm_case(M, N, K, 8)
                nat.load().sbk_prof_set_knob(2, 1)
                gemm_case(M, N, K, 8)
        nat.load().sbk_prof_set_knob(2, 0)
        sys.exit(0)
    # (Modes that toggled knobs removed in round 5 -- --enc-gemm, --relpos-t, --attn2, --k512, --skinny, --sk, --sk64, --pmc-decode,
    #  --flat64, the tile / grid / measurement-mode columns of --x3p / --x3 / --bf16a -- went with the kernels they compared: their
    #  logs are under profiles/r02_* .. r04_*.)
    if "--ffn2" in sys.argv:  # few rows, K = 2048: register-operand split-K (default) vs 64x64 LDS tiles with a 4-way K split
        for knob in (0, 1):
            nat.load().sbk_prof_set_knob(14, knob)
            print("K = 2048 path:", "64x64 LDS tiles, 4-way K split" if knob else "register-operand tiles")
            for M in (10, 40, 80, 160, 320, 640, 1280, 2560):
                gemm_case(M, 512, 2048, 8)
            gemm_case(1280, 768, 3072, 8)
        nat.load().sbk_prof_set_knob(14, 0)
        sys.exit(0)
    if "--stream" in sys.argv:  # hand-written float4 streaming kernels of the library (sbk_prof_stream_f32): the HBM calibration
        for mb in (64, 256, 1024, 4096):
            n = mb * 1024 * 1024 // 4
            x = torch.empty(n, device=dev).normal_()
            y = torch.empty_like(x)
            us = ctypes.c_float(0)
            res = {}
            for mode, tag, nbytes in ((0, "copy", 8.0 * n), (1, "read", 4.0 * n)):
                rc = nat.load().sbk_prof_stream_f32(nat._p(x), nat._p(y), n, mode, 20, ctypes.byref(us), nat._stream(x))
                assert rc == 0
                res[tag] = f"{us.value:9.1f} us {nbytes / us.value / 1e6:6.2f} TB/s"
            print(f"stream {mb} MiB:", res, flush=True)
        sys.exit(0)
    if "--enc-layer" in sys.argv:  # the Conformer-L encoder in situ (32 utterances x 5 / 10 / 20 / 30 s): per-kernel event times by GEMM routing
        from speechbrain_amd.inference.builders import build_asr
        asr = build_asr("L", vocab=5000, seed=0, device="cuda:0")
        variants = (("tile grid", {18: 0}), ("routed (default)", {18: 1}))
        for sec in (5, 10, 20, 30):
            wav = (0.1 * torch.randn(32, sec * 16000, generator=torch.Generator().manual_seed(sec))).to(dev)
            lens = torch.ones(32, device=dev)
            for tag, knobs in variants:
                for k, v in knobs.items():
                    nat.load().sbk_prof_set_knob(k, v)
                with torch.no_grad():
                    for _ in range(2):
                        asr.encode_batch(wav, lens)
                    torch.cuda.synchronize()
                    nat.prof_reset(); nat.prof_enable(True)
                    for _ in range(3):
                        asr.encode_batch(wav, lens)
                    torch.cuda.synchronize()
                    nat.prof_enable(False)
                rep = nat.prof_report(); nat.prof_reset()
                gem = {k: v for k, v in rep.items() if k.startswith("gemm")}
                tot = sum(v["ms"] for v in rep.values())
                gms, gfl = sum(v["ms"] for v in gem.values()), sum(v["flops"] for v in gem.values())
                print(f"enc 32x{sec}s {tag:18s}: all kernels {tot / 3:8.2f} ms | GEMMs {gms / 3:8.2f} ms {gfl / gms / 1e9:6.1f} TF/s |",
                      {k: (round(v["ms"] / 3, 2), round(v["flops"] / v["ms"] / 1e9, 1)) for k, v in gem.items()}, flush=True)
        nat.load().sbk_prof_set_knob(18, 1)
        sys.exit(0)
    if "--enc-bf16" in sys.argv:  # Conformer-L encoder under precision bf16: bf16 activations in memory (feed-forward pairs) vs fp32 activations rounded on load
        from speechbrain_amd.inference.builders import build_asr
        asr = build_asr("L", vocab=5000, seed=0, device="cuda:0")
        asr.eval_precision = "bf16"
        for sec in (5, 10, 20):
            wav = (0.1 * torch.randn(32, sec * 16000, generator=torch.Generator().manual_seed(sec))).to(dev)
            lens = torch.ones(32, device=dev)
            for flag in (True, False):
                nat.BF16_ACTIVATIONS = flag
                with torch.no_grad():
                    for _ in range(2):
                        asr.encode_batch(wav, lens)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        asr.encode_batch(wav, lens)
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t0) / 5
                    nat.prof_reset(); nat.prof_enable(True)
                    for _ in range(3):
                        asr.encode_batch(wav, lens)
                    torch.cuda.synchronize()
                    nat.prof_enable(False)
                rep = nat.prof_report(); nat.prof_reset()
                tot = sum(v["ms"] for v in rep.values())
                print(f"enc 32x{sec}s bf16 activations {flag}: wall {1e3 * wall:7.2f} ms, kernel events {tot / 3:7.2f} ms |",
                      {k: (v["count"] // 3, round(v["ms"] / 3, 2), round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:7]}, flush=True)
        nat.BF16_ACTIVATIONS = True
        sys.exit(0)
    if "--mfma-peak" in sys.argv:  # f32 matrix-core ceiling under DVFS (registers only), then the stream-K GEMM with its panel loads off
        sink = torch.zeros(4096, device=dev)
        tf = ctypes.c_float(0)
        for wgs in (256, 512, 1024):
            for rnd in (0, 1):
                rc = nat.load().sbk_prof_mfma_peak_f32(nat._p(sink), wgs, 20000, rnd, ctypes.byref(tf), nat._stream(sink))
                assert rc == 0
                print(f"mfma peak: {wgs} workgroups x 4 waves, {'random' if rnd else 'zero'} operands: {tf.value:7.1f} TFLOP/s", flush=True)
        for tag, knobs in (("stream-K", {18: 1}), ("tile grid", {18: 0})):
            for k, v in knobs.items():
                nat.load().sbk_prof_set_knob(k, v)
            print("variant:", tag, flush=True)
            for (M, N, K) in [(56064, 2048, 512), (56064, 512, 2048), (56064, 512, 512)]:
                gemm_case(M, N, K, 0, iters=30)
        nat.load().sbk_prof_set_knob(18, 1)
        print("zero-filled operands (DVFS give-back), stream-K then tile grid")
        for mode in (1, 0):
            nat.load().sbk_prof_set_knob(18, mode)
            for (M, N, K) in [(56064, 2048, 512), (56064, 512, 2048)]:
                a = torch.zeros(M, K, device=dev); w = torch.zeros(N, K, device=dev); out = torch.empty(M, N, device=dev)
                us = ctypes.c_float(0)
                rc = nat.load().sbk_prof_gemm_repeat_f32(nat._p(a), nat._p(w), nat._p(out), M, N, K, None, 0, 30, ctypes.byref(us), nat._stream(a))
                assert rc == 0
                print(f"gemm zeros M={M} N={N} K={K}: {us.value:8.2f} us  {2.0*M*N*K/us.value/1e6:7.2f} TFLOP/s", flush=True)
        nat.load().sbk_prof_set_knob(18, 1)
        sys.exit(0)
    if "--sk-pmc" in sys.argv:  # short: the two big-GEMM kernels for a counters pass
        for mode in (1, 0):
            nat.load().sbk_prof_set_knob(18, mode)
            for (M, N, K) in [(56064, 2048, 512), (56064, 512, 2048), (12800, 2048, 512)]:
                gemm_case(M, N, K, 0, iters=5)
        nat.load().sbk_prof_set_knob(18, 1)
        sys.exit(0)
    if "--x3-pmc" in sys.argv:  # short: the split-operand kernel at a bench-sized shape for a counters pass
        nat.F32X3, nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = True, 1, 1
        for (M, N, K) in [(12800, 2048, 512), (12800, 512, 2048), (24032, 1536, 512)]:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
            for _ in range(6):
                nat.gemm_nt(a, w)
            torch.cuda.synchronize()
        sys.exit(0)
    if "--r4-pmc" in sys.argv:  # short: round 4's two contraction kernels at bench-sized shapes for the counters passes
        nat.F32X3, nat.X3P = True, True
        for (M, N, K) in [(1280, 512, 512), (1280, 2048, 512), (1280, 512, 2048), (1280, 5000, 512)]:  # gemm_x3r_kernel
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); r = torch.randn(M, N, device=dev)
            for _ in range(6):
                nat.gemm_nt_x3r(a, w, residual=r)
            torch.cuda.synchronize()
        for (M, N, K) in [(12800, 2048, 512), (12800, 512, 2048), (24032, 1536, 512)]:  # gemm_nt_x3p_kernel (A panel made once)
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
            pa = nat.split_x3p(a)
            for _ in range(6):
                nat.gemm_nt_x3p(pa, w)
            torch.cuda.synchronize()
        sys.exit(0)
    if "--x3r" in sys.argv:  # the decode step's projections: fp32-MFMA default route vs sbk_gemm_nt_x3r (both load schedules)
        def ev_time(fn, n=40):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        lib = nat.load()
        for M in (320, 640, 1280, 2560):
            for (N, K) in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (5000, 512)):
                a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); r = torch.randn(M, N, device=dev)
                t0 = ev_time(lambda: nat.gemm_nt_splitk(a, w, residual=r, slices=4))  # (the search's own route: gemm_nt_ws)
                line = f"x3r M={M} N={N} K={K}: fp32-MFMA route {t0:6.1f} us {2.0*M*N*K/t0/1e6:6.1f} TF/s |"
                t = ev_time(lambda: nat.gemm_nt_x3r(a, w, residual=r))
                line += f" x3r, A fp32: {t:6.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s |"
                ref = a.double() @ w.double().t() + r.double()
                e3 = float((nat.gemm_nt_x3r(a, w, residual=r).double() - ref).pow(2).mean().sqrt())
                e0 = float((nat.gemm_nt_splitk(a, w, residual=r, slices=4).double() - ref).pow(2).mean().sqrt())
                print(line + f" rms err vs fp64: x3r {e3:.3e} fp32 {e0:.3e}", flush=True)
        sys.exit(0)
    if "--x3r-ln" in sys.argv:  # LayerNorm + sbk_gemm_nt_x3r (two launches) vs sbk_gemm_ln_nt_x3r (the LayerNorm in the prologue)
        def ev_time(fn, n=40):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        K = 512
        for M in (640, 1280, 2560):
            for N in (512, 1536, 2048, 5000):
                a = torch.randn(M, K, device=dev) * 2 + 0.5; w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
                g = 1 + 0.1 * torch.randn(K, device=dev); bt = 0.1 * torch.randn(K, device=dev)
                wf, bf = nat._fold_ln(w, b, g, bt)
                h = torch.empty_like(a)
                t_ln = ev_time(lambda: nat.layernorm(a, g, bt, 1e-5, out=h))
                t_g = ev_time(lambda: nat.gemm_nt_x3r(h, w, b))
                t2 = ev_time(lambda: nat.gemm_nt_x3r(nat.layernorm(a, g, bt, 1e-5, out=h), w, b))
                t1 = ev_time(lambda: nat.gemm_ln_nt_x3r(a, wf, bf, 1e-5))
                ref = torch.nn.functional.layer_norm(a.double(), (K,), g.double(), bt.double(), 1e-5) @ w.double().t() + b.double()
                e1 = float((nat.gemm_ln_nt_x3r(a, wf, bf, 1e-5).double() - ref).abs().max())
                e2 = float((nat.gemm_nt_x3r(nat.layernorm(a, g, bt, 1e-5), w, b).double() - ref).abs().max())
                print(f"x3r-ln M={M} N={N}: layernorm {t_ln:5.1f} us + x3r {t_g:5.1f} us, back to back {t2:5.1f} us | pre-pass {t1:5.1f} us "
                      f"({2.0*M*N*K/t1/1e6:5.1f} TF/s) | max err vs fp64: {e1:.2e}, two launches {e2:.2e}", flush=True)
        sys.exit(0)
    if "--ln-x3p" in sys.argv:  # LayerNorm written as the next contraction's panel operand vs LayerNorm + split pass
        def ev_time(fn, n=30):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        for M in (4032, 12800, 24032):
            x = torch.randn(M, 512, device=dev); g = torch.randn(512, device=dev); b = torch.randn(512, device=dev)
            t0 = ev_time(lambda: nat.layernorm(x, g, b, 1e-5))
            y = nat.layernorm(x, g, b, 1e-5)
            t1 = ev_time(lambda: nat.split_x3p(y))
            t2 = ev_time(lambda: nat.layernorm_x3p(x, g, b, 1e-5))
            print(f"ln-x3p M={M} d=512: layernorm {t0:6.1f} us ({8.0*M*512/t0/1e3:5.0f} GB/s) + split {t1:6.1f} us | layernorm_x3p {t2:6.1f} us ({10.0*M*512/t2/1e3:5.0f} GB/s)", flush=True)
        sys.exit(0)
    if "--x3p-modes" in sys.argv:  # measurement builds of gemm_nt_x3p_kernel (key 64) at the encoder's shapes
        def ev_time(fn, n=20):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        lib = nat.load()
        names = {0: "kernel", 1: "no LDS-DMA in the loop", 2: "no MFMAs", 4: "no epilogue", 8: "no fragment fetches"}
        for (M, N, K, act, res, pout) in [(56064, 2048, 512, nat.ACT_SWISH, False, True), (56064, 512, 2048, nat.ACT_NONE, True, False),
                                          (56064, 1536, 512, nat.ACT_NONE, False, False), (56064, 512, 512, nat.ACT_NONE, True, False),
                                          (14016, 2048, 512, nat.ACT_SWISH, False, True), (14016, 512, 2048, nat.ACT_NONE, True, False)]:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
            r = torch.randn(M, N, device=dev) if res else None
            pa = nat.split_x3p(a)
            fn = (lambda: nat.gemm_nt_x3p(pa, w, b, r, act=act, alpha=0.5, panel_out=True, fp32_out=False)) if pout else \
                 (lambda: nat.gemm_nt_x3p(pa, w, b, r, act=act, alpha=0.5))
            line = f"x3p M={M} N={N} K={K} act={act}{' + residual' if res else ''}{' -> panel' if pout else ''}:"
            modes = (0, 0) if "--x3p-epi" in sys.argv else (0, 1, 2, 4, 8, 0)
            outs = []
            for fe in ((0, 1, 0, 1) if "--x3p-epi" in sys.argv else (1,)):  # key 63: the generic epilogue / the straight-line forms
                lib.sbk_prof_set_knob(63, fe)
                if "--x3p-epi" in sys.argv:
                    line += f"  | key 63 = {fe}:"
                    o = fn()
                    outs.append(o.data.clone() if hasattr(o, "data") and not torch.is_tensor(o) else o.clone())
                for mode in modes:
                    lib.sbk_prof_set_knob(64, mode)
                    t = ev_time(fn)
                    line += f"  [{names[mode]}] {t:6.1f} us"
            lib.sbk_prof_set_knob(64, 0)
            lib.sbk_prof_set_knob(63, 1)
            same = f"  identical: {bool(torch.equal(outs[0], outs[1]))}" if outs else ""
            print(line + f"  ({2.0*M*N*K/t/1e6:5.1f} TF/s){same}", flush=True)
        sys.exit(0)
    if "--x3p" in sys.argv:  # both operands pre-split, panel layout, 256-wide tiles (csrc/gemm_x3p.hip) vs the f32x3 kernel
        def ev_time(fn, n=30):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        shapes = [(M, N, K) for M in (4032, 8000, 14016, 24032) for (N, K) in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 512))]
        if "--x3p-short" in sys.argv:
            shapes = [(12800, 2048, 512), (12800, 512, 2048), (14016, 1536, 512), (14016, 512, 512), (4032, 1024, 512)]
        nat.F32X3, nat.X3P = True, False
        for (M, N, K) in shapes:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
            nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = 1, 1
            t3 = ev_time(lambda: nat.gemm_nt(a, w))
            ref3 = nat.gemm_nt(a, w)
            nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = 2048, 192
            tsp = ev_time(lambda: nat.split_x3p(a))
            pa = nat.split_x3p(a)
            line = f"x3p M={M} N={N} K={K}: f32x3 {t3:7.1f} us {2.0*M*N*K/t3/1e6:6.1f} TF/s | split A {tsp:6.1f} us {10.0*M*K/tsp/1e3:6.0f} GB/s |"
            t = ev_time(lambda: nat.gemm_nt_x3p(pa, w))
            tp = ev_time(lambda: nat.gemm_nt_x3p(pa, w, panel_out=True, fp32_out=False)) if N % 16 == 0 else float("nan")
            line += f" x3p (256x128 tiles): {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s (panel out {tp:7.1f} us) |"
            out = nat.gemm_nt_x3p(pa, w)
            ref = (a.double() @ w.double().t())
            line += f" rms err vs fp64: x3p {float((out.double() - ref).pow(2).mean().sqrt()):.3e} f32x3 {float((ref3.double() - ref).pow(2).mean().sqrt()):.3e}"
            print(line, flush=True)
        sys.exit(0)
    if "--x3" in sys.argv:  # fp32 contraction on the bf16 matrix pipe (three-way operand split) vs the fp32-MFMA persistent kernel
        def ev_time(fn, n=30):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        shapes = [(M, N, K) for M in (4032, 8000, 12800, 24032) for (N, K) in ((2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 512))]
        if "--x3-decode" in sys.argv:  # the decoder's vocabulary projection and the memory projections (internal calls of the search)
            shapes = [(640, 5000, 512), (1280, 5000, 512), (2560, 5000, 512), (12800, 1024, 512), (12800, 5000, 512)]
        if "--x3-short" in sys.argv:
            shapes = [(12800, 2048, 512), (12800, 512, 2048), (4032, 512, 512)]
        for (M, N, K) in shapes:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
            nat.F32X3 = False
            t32 = ev_time(lambda: nat.gemm_nt(a, w))
            ref = nat.gemm_nt(a, w)
            nat.F32X3, nat.F32X3_MIN_ROWS, nat.F32X3_MIN_TILES = True, 1, 1
            t = ev_time(lambda: nat.gemm_nt(a, w))
            line = f"f32x3 M={M} N={N} K={K}: fp32-MFMA {t32:7.1f} us {2.0*M*N*K/t32/1e6:6.1f} TF/s | f32x3 {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF/s |"
            if "--x3-zero" in sys.argv:  # DVFS probe: the same launches on zero-filled operands (no toggling in the multipliers)
                az, wz = torch.zeros_like(a), torch.zeros_like(w)
                t1 = ev_time(lambda: nat.gemm_nt(a, w), n=50)
                t0 = ev_time(lambda: nat.gemm_nt(az, wz), n=50)
                line += f" random {t1:6.1f} us, zeros {t0:6.1f} us |"
            out = nat.gemm_nt(a, w)
            exact = a.double() @ w.double().t()
            e3, e32 = float((out.double() - exact).pow(2).mean().sqrt()), float((ref.double() - exact).pow(2).mean().sqrt())
            line += f" rms err vs fp64: x3 {e3:.3e} fp32-MFMA {e32:.3e} (output rms {float(exact.pow(2).mean().sqrt()):.1f})"
            print(line, flush=True)
        nat.F32X3_MIN_ROWS = 2048
        sys.exit(0)
    if "--bf16a" in sys.argv:  # bf16 activations + weights through the LDS-DMA pipeline vs the kernel that reads fp32 activations
        def ev_time(fn, n=20):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        shapes = [(12000, 3840, 1280), (12000, 1280, 1280), (12000, 5120, 1280), (12000, 1280, 5120), (12800, 2048, 512), (12800, 512, 2048), (48000, 1280, 1280)]
        for (M, N, K) in shapes:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); ab = a.bfloat16()
            t_old = ev_time(lambda: nat.gemm_nt_bf16(a, w))
            line = f"bf16 gemm M={M} N={N} K={K}: fp32-A kernel {t_old:7.1f} us {2.0*M*N*K/t_old/1e6:7.1f} TF/s |"
            t32 = ev_time(lambda: nat.gemm_nt_bf16a(ab, w))
            line += f" bf16-A LDS-DMA kernel: {t32:7.1f} us {2.0*M*N*K/t32/1e6:7.1f} TF/s |"
            print(line, flush=True)
        sys.exit(0)
    if "--lp256-modes" in sys.argv:  # measurement builds / schedule variants of gemm_nt_lp256_kernel (key 62), bf16 operands
        def ev_time(fn, n=20):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        lib = nat.load()
        names = {0: "kernel", 1: "no LDS-DMA in the loop", 2: "no MFMAs", 4: "no epilogue", 8: "no fragment fetches", 16: "MFMA phase at priority 1",
                 32: "LDS reads awaited behind the barrier", 48: "both"}
        for (M, N, K, od, res) in [(12000, 3840, 1280, torch.bfloat16, False), (12000, 1280, 5120, torch.float32, True),
                                   (12000, 1280, 1280, torch.float32, True), (8192, 8192, 8192, torch.bfloat16, False)]:
            a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if res else None
            fn = lambda: nat.gemm_nt_bf16a(a, w, b, r, out_dtype=od)
            lib.sbk_prof_set_knob(61, 0)
            t = ev_time(fn)
            print(f"M={M} N={N} K={K} out={str(od).replace('torch.', '')}{' + residual' if res else ''}: 128^2 kernel {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF/s", flush=True)
            lib.sbk_prof_set_knob(61, 2)
            for mode in (0, 1, 2, 4, 8, 16, 32, 48, 0):
                lib.sbk_prof_set_knob(62, mode)
                t = ev_time(fn)
                print(f"    mode {mode:2d} ({names[mode]}): {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF/s", flush=True)
            lib.sbk_prof_set_knob(62, 0)
            lib.sbk_prof_set_knob(61, 1)
        sys.exit(0)
    if "--lp256-pmc" in sys.argv:  # short: the Whisper layer's four contractions on the 256 x 256 kernel, 6 launches each, for the counters passes
        shapes = [(12000, 3840, 1280, nat.ACT_NONE, torch.bfloat16), (12000, 1280, 1280, nat.ACT_NONE, torch.float32),
                  (12000, 5120, 1280, nat.ACT_GELU, "hidden"), (12000, 1280, 5120, nat.ACT_NONE, torch.float32)]
        for fp8 in (False, True):
            for (M, N, K, act, od) in shapes:
                a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; ab = a.bfloat16(); aq = nat.quant_rows_fp8(a)
                b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if od is torch.float32 else None
                nat.gemm_nt_fp8a(aq, w, b, r) if fp8 else nat.gemm_nt_bf16a(ab, w, b, r)  # (weight images cached before the counted launches)
                torch.cuda.synchronize()
                for _ in range(6):
                    if fp8:
                        nat.gemm_nt_fp8a(aq, w, b, r, act=act, out_dtype="fp8" if od == "hidden" else od)
                    else:
                        nat.gemm_nt_bf16a(ab, w, b, r, act=act, out_dtype=torch.bfloat16 if od == "hidden" else od)
                torch.cuda.synchronize()
        sys.exit(0)
    if "--lp256" in sys.argv:  # bf16 / e4m3 activation x weight contractions: 128 x 128 tiles (key 61 = 0) vs 256 x 256 (csrc/gemm_lp256.hip)
        def ev_time(fn, n=20):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        lib = nat.load()
        # (rows, N, K, activation, output): the four projections of a Whisper large-v3 encoder layer at 8 x 30 s, then Conformer-L's
        if "--lp256-small" in sys.argv:  # where the 256 x 256 route starts to pay: 36 ... 360 tiles (1 / 2 / 4 / 8 x 30 s of Whisper rows)
            small = [(M, N, K, nat.ACT_NONE, od) for M in (1500, 3000, 6000) for (N, K, od) in ((3840, 1280, torch.bfloat16), (1280, 1280, torch.float32), (5120, 1280, "hidden"), (1280, 5120, torch.float32))]
        shapes = [(12000, 3840, 1280, nat.ACT_NONE, torch.bfloat16), (12000, 1280, 1280, nat.ACT_NONE, torch.float32),
                  (12000, 5120, 1280, nat.ACT_GELU, "hidden"), (12000, 1280, 5120, nat.ACT_NONE, torch.float32),
                  (12000, 5120, 1280, nat.ACT_NONE, "hidden"), (48000, 1280, 1280, nat.ACT_NONE, torch.float32),
                  (14016, 2048, 512, nat.ACT_SWISH, torch.bfloat16), (14016, 512, 2048, nat.ACT_NONE, torch.float32)]
        if "--lp256-small" in sys.argv:
            shapes = small
        for (M, N, K, act, od) in shapes:
            a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; ab = a.bfloat16(); aq = nat.quant_rows_fp8(a)
            b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev) if od is torch.float32 else None
            line = f"M={M} N={N} K={K} ({-(-M // 256) * -(-N // 256)} tiles) act={act} out={'fp8/bf16' if od == 'hidden' else str(od).replace('torch.', '')}:"
            for name, fn in (("bf16a", lambda: nat.gemm_nt_bf16a(ab, w, b, r, act=act, out_dtype=torch.bfloat16 if od == "hidden" else od)),
                             ("fp8a", lambda: nat.gemm_nt_fp8a(aq, w, b, r, act=act, out_dtype="fp8" if od == "hidden" else od))):
                ts = []
                for knob in (0, 2):
                    lib.sbk_prof_set_knob(61, knob)
                    ts.append(ev_time(fn))
                line += f"  {name} 128^2 {ts[0]:7.1f} us {2.0*M*N*K/ts[0]/1e6:7.1f} TF/s | 256^2 {ts[1]:7.1f} us {2.0*M*N*K/ts[1]/1e6:7.1f} TF/s;"
            lib.sbk_prof_set_knob(61, 1)
            print(line, flush=True)
        sys.exit(0)
    if "--copy" in sys.argv:  # what a plain streaming kernel reaches on this box (calibrates the HBM rooflines)
        for mb in (256, 1024, 4096):
            x = torch.empty(mb * 1024 * 1024 // 4, device=dev).normal_()
            y = torch.empty_like(x)
            t_copy = timeit(lambda: y.copy_(x), n=20, warm=3)
            t_read = timeit(lambda: x.sum(), n=20, warm=3)
            print(f"copy {mb} MiB: {2 * x.numel() * 4 / t_copy / 1e6:7.2f} TB/s (read+write) | "
                  f"sum-reduce read {x.numel() * 4 / t_read / 1e6:7.2f} TB/s", flush=True)
        sys.exit(0)
    if "--ctc" in sys.argv:
        for (B, T) in [(32, 251), (32, 440), (32, 751), (8, 440)]:
            ctc_case(B, T, 5000, 10, 5)
        sys.exit(0)
    print("launch overhead (empty-ish layernorm 4 rows):", end=" ")
    x = torch.randn(4, 512, device=dev); g = torch.ones(512, device=dev); bb = torch.zeros(512, device=dev)
    print(f"{timeit(lambda: nat.layernorm(x, g, bb, 1e-5)):.2f} us")
    print("decode-step shapes at 320 rows: register-operand (skinny) kernels, then the LDS-tiled ones (knob 2 = 1)")
    for (M, N, K) in [(320, 512, 512), (320, 1536, 512), (320, 2048, 512), (320, 512, 2048), (320, 5000, 512)]:
        gemm_case(M, N, K, 8)
    nat.load().sbk_prof_set_knob(2, 1)
    for (M, N, K) in [(320, 512, 512), (320, 1536, 512), (320, 2048, 512), (320, 512, 2048), (320, 5000, 512), (640, 2048, 512), (640, 5000, 512)]:
        gemm_case(M, N, K, 8)
    nat.load().sbk_prof_set_knob(2, 0)
    sys.exit(0)

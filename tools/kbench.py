#!/usr/bin/env python
"""Kernel micro-benchmarks at the real Hi3D stage-2 / stage-1 shapes (no model build).
usage: python tools/kbench.py [gemm|conv|attn|norm|all] [--s1]
Prints one line per shape: ms, TFLOP/s or GB/s.  Random data (never zero-filled)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hi3d-official_amd"))
import torch  # noqa: E402

from hi3d_hip import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rb(*shape):
    return (torch.randn(shape, device=dev) * 0.5).to(torch.bfloat16)


def bench_gemm(F=32, lat=128):
    print("== dense GEMM (M,N,K,kind)")
    for ds, C in ((1, 320), (2, 640), (4, 1280)):
        M = F * (lat // ds) ** 2
        for name, N, K, kw in (("qkv", 3 * C, C, {}), ("proj", C, C, {"res": True}), ("geglu", 8 * C, C, {"geglu": True}),
                               ("ff2", C, 4 * C, {"res": True})):
            A, W = rb(M, K), rb(N, K)
            bias = torch.randn(N, device=dev)
            R1 = rb(M, N) if kw.get("res") else None
            out = torch.empty((M, N // 2 if kw.get("geglu") else N), device=dev, dtype=torch.bfloat16)
            ms = timeit(lambda: ops.gemm(A, W, M=M, N=N, K=K, bias=bias, R1=R1, geglu=bool(kw.get("geglu")), out=out))
            print(f"  {name:6s} M={M:7d} N={N:5d} K={K:5d}: {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")


def bench_conv(F=32, lat=128):
    print("== conv3x3 implicit GEMM (H, Cin, Cout)")
    for H, Cin, Cout in ((lat, 320, 320), (lat, 640, 320), (lat, 960, 320), (lat // 2, 640, 640), (lat // 2, 1280, 640),
                         (lat // 4, 1280, 1280), (lat // 4, 2560, 1280), (lat // 8, 1280, 1280)):
        M, K = F * H * H, 9 * Cin
        A, W = rb(M, Cin), rb(Cout, K)
        bias = torch.randn(Cout, device=dev)
        out = torch.empty((M, Cout), device=dev, dtype=torch.bfloat16)
        geo = dict(Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, up2x=0)
        ms = timeit(lambda: ops.gemm(A, W, M=M, N=Cout, K=K, bias=bias, conv3x3=geo, out=out))
        print(f"  H={H:4d} {Cin:5d}->{Cout:5d}: {ms:8.3f} ms  {2.0 * M * Cout * K / ms / 1e9:8.1f} TFLOP/s")
    print("== conv temporal (3,1,1)")
    for H, C in ((lat, 320), (lat // 2, 640), (lat // 4, 1280)):
        M, K = F * H * H, 3 * C
        A, W = rb(M, C), rb(C, K)
        out = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(A, W, M=M, N=C, K=K, convt3=dict(T=16, HW=H * H, Cin=C), out=out))
        print(f"  H={H:4d} C={C:5d}: {ms:8.3f} ms  {2.0 * M * C * K / ms / 1e9:8.1f} TFLOP/s")


def bench_attn(F=32, lat=128):
    print("== spatial attention d64 (B,H,S)")
    for ds, C in ((1, 320), (2, 640), (4, 1280)):
        S, H = (lat // ds) ** 2, C // 64
        qkv = rb(F * S, 3 * C)
        vt = ops.transpose_v(qkv[:, 2 * C:], F, H, S, 3 * C)
        out = torch.empty((F * S, C), device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.attention_d64(qkv, qkv[:, C:], vt, F, H, S, S, 3 * C, 3 * C, 0.125, out=out), iters=5)
        ms_t = timeit(lambda: ops.transpose_v(qkv[:, 2 * C:], F, H, S, 3 * C), iters=5)
        ms_v = timeit(lambda: ops.attention_d64_v(qkv, qkv[:, C:], qkv[:, 2 * C:], F, H, S, S, 3 * C, 3 * C, 3 * C, 0.125, out=out), iters=5)
        print(f"  B={F} H={H:2d} S={S:6d}: V^T operand {ms:8.3f} ms (+ transpose {ms_t:6.3f})  {4.0 * F * H * S * S * 64 / ms / 1e9:8.1f} TFLOP/s | "
              f"row-major V {ms_v:8.3f} ms  {4.0 * F * H * S * S * 64 / ms_v / 1e9:8.1f} TFLOP/s")
    print("== temporal attention")
    for ds, C in ((1, 320), (2, 640), (4, 1280)):
        S, H = (lat // ds) ** 2, C // 64
        qkv = rb(F * S, 3 * C)
        ms = timeit(lambda: ops.attention_temporal_fused_qkv(qkv, 2, F // 2, S, H))
        print(f"  S={S:6d} H={H:2d}: {ms:8.3f} ms  {2.0 * 4 * F * S * C / ms / 1e6:8.1f} GB/s(q,k,v,o)")


def bench_norm(F=32, lat=128):
    print("== GroupNorm+SiLU (2-D: inst=F ; 3-D: inst=2)")
    for ds, C in ((1, 320), (1, 960), (2, 640), (4, 1280), (4, 2560)):
        P = (lat // ds) ** 2
        x = rb(F * P, C)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        for inst, PP in ((F, P), (2, F // 2 * P)):
            ms = timeit(lambda: ops.groupnorm_silu(x, g, b, inst, PP, C, 1e-5, True, out=out))
            print(f"  inst={inst:3d} P={PP:7d} C={C:5d}: {ms:8.3f} ms  {2.0 * 2 * F * P * C / ms / 1e6:8.1f} GB/s(alg r+w)")
    print("== LayerNorm")
    for ds, C in ((1, 320), (2, 640), (4, 1280)):
        R = F * (lat // ds) ** 2
        x = rb(R, C)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        ms = timeit(lambda: ops.layernorm(x, g, b, R, C, out=out))
        print(f"  R={R:7d} C={C:5d}: {ms:8.3f} ms  {2.0 * 2 * R * C / ms / 1e6:8.1f} GB/s")


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("one", "attn1", "ffn", "sweep", "conv1", "tattn", "vaesweep", "upphase")):
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    lat = 64 if "--s1" in sys.argv else 128
    if which in ("gemm", "all"):
        bench_gemm(lat=lat)
    if which in ("conv", "all"):
        bench_conv(lat=lat)
    if which in ("attn", "all"):
        bench_attn(lat=lat)
    if which in ("norm", "all"):
        bench_norm(lat=lat)


def bench_one():
    """kbench.py one <kind> M N K : a single GEMM shape, few iterations (for rocprofv3 --pmc)."""
    kind, M, N, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    A, W = rb(M, K), rb(N, K)
    if len(sys.argv) > 6 and sys.argv[6] == "zero":      # power probe
        A.zero_(); W.zero_()
    bias = torch.randn(N, device=dev)
    geglu = kind == "geglu"
    R1 = rb(M, N) if kind == "res" else None
    out = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(A, W, M=M, N=N, K=K, bias=bias, R1=R1, geglu=geglu, out=out), iters=3, warm=1)
    print(f"{kind} M={M} N={N} K={K}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


if len(sys.argv) > 1 and sys.argv[1] == "one":
    bench_one()


def bench_attn_one():
    """kbench.py attn1 B H S : one spatial-attention shape (for rocprofv3 --pmc)."""
    B, H, S = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    C = H * 64
    qkv = rb(B * S, 3 * C)
    pre = len(sys.argv) > 5 and sys.argv[5] in ("pre", "zero", "const")     # the UNet's form: scale * log2(e) folded into q, scale argument 0
    if pre:
        qkv[:, :C] *= 0.125 * 1.4426950408889634
    if len(sys.argv) > 5 and sys.argv[5] == "zero":      # power probe: no operand toggling at all
        qkv.zero_()
    if len(sys.argv) > 5 and sys.argv[5] == "const":     # power probe: every element the same value
        qkv.fill_(0.25)
    vt = ops.transpose_v(qkv[:, 2 * C:], B, H, S, 3 * C)
    out = torch.empty((B * S, C), device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.attention_d64(qkv, qkv[:, C:], vt, B, H, S, S, 3 * C, 3 * C, 0.0 if pre else 0.125, out=out), iters=3, warm=1)
    print(f"attn B={B} H={H} S={S}{' pre-scaled q' if pre else ''}: {ms:.3f} ms {4.0 * B * H * S * S * 64 / ms / 1e9:.1f} TFLOP/s")


if len(sys.argv) > 1 and sys.argv[1] == "attn1":
    bench_attn_one()


def bench_ffn():
    """kbench.py ffn : fused GEGLU feed-forward vs GEGLU GEMM + second GEMM at the 320-channel level."""
    C = 320
    zero = len(sys.argv) > 2 and sys.argv[2] == "zero"     # power probe: no operand toggling
    for M in (524288, 131072):
        x, R1 = rb(M, C), rb(M, C)
        w1, w2 = rb(8 * C, C), rb(C, 4 * C)
        if zero:
            x.zero_(); w1.zero_(); w2.zero_(); R1.zero_()
        b1, b2 = torch.randn(8 * C, device=dev), torch.randn(C, device=dev)
        out = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
        gg = torch.empty((M, 4 * C), device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
        ms = timeit(lambda: ops.ffn_geglu(x, w1, b1, w2, b2, M=M, C=C, R1=R1, out=out))
        print(f"  fused   M={M:7d}: {ms:8.3f} ms {fl / ms / 1e9:8.1f} TFLOP/s")

        def two():
            ops.gemm(x, w1, M=M, N=8 * C, K=C, bias=b1, geglu=True, out=gg)
            ops.gemm(gg, w2, M=M, N=C, K=4 * C, bias=b2, R1=R1, out=out)
        ms = timeit(two)
        print(f"  2 GEMMs M={M:7d}: {ms:8.3f} ms {fl / ms / 1e9:8.1f} TFLOP/s")


if len(sys.argv) > 1 and sys.argv[1] == "ffn":
    bench_ffn()


def bench_sweep():
    """kbench.py sweep [variants...] : every GEMM / conv shape of the stage-2 step (and the VAE decoder's convs)
    timed under each HI3D_GEMM_VARIANT (default: the host heuristic 'h', 0, 2, 6, 7; GEGLU also 3, 5), same
    buffers, interleaved rounds -- the table the dispatch heuristic in csrc/gemm.hip is set from."""
    variants = sys.argv[2:] or ["h", "0", "2", "6", "7"]
    dense = [  # (M, N, K, kind)
        (131072, 5120, 640, "geglu"), (32768, 10240, 1280, "geglu"), (8192, 10240, 1280, "geglu"),
        (524288, 960, 320, "plain"), (131072, 1920, 640, "plain"), (32768, 3840, 1280, "plain"),
        (524288, 320, 320, "res"), (131072, 640, 640, "res"), (32768, 1280, 1280, "res"),
        (131072, 640, 2560, "res"), (32768, 1280, 5120, "res"), (8192, 1280, 5120, "res"),
        (524288, 320, 960, "res"), (131072, 640, 1920, "res"), (32768, 1280, 2560, "res"),   # skip_connection 1x1
        (1048576, 128, 256, "plain"), (16384, 16384, 512, "plain"),                          # VAE nin_shortcut, q k^T
        (524288, 2560, 320, "geglu"), (524288, 320, 1280, "res"),                            # the two GEMMs the fused FFN replaces
    ]
    conv = [  # (frames, Hin, Cin, Cout, stride, up2x, res)
        (32, 128, 320, 320, 1, 0, True), (32, 128, 640, 320, 1, 0, False), (32, 128, 960, 320, 1, 0, False),
        (32, 64, 640, 640, 1, 0, True), (32, 64, 1280, 640, 1, 0, False), (32, 64, 1920, 640, 1, 0, False),
        (32, 32, 1280, 1280, 1, 0, True), (32, 32, 2560, 1280, 1, 0, False), (32, 16, 1280, 1280, 1, 0, True),
        (32, 128, 320, 320, 2, 0, False), (32, 64, 640, 640, 1, 1, False), (32, 32, 1280, 1280, 1, 1, False),
        (1, 1024, 128, 128, 1, 0, True), (1, 512, 256, 256, 1, 0, True), (1, 256, 512, 512, 1, 0, True),
        (1, 128, 512, 512, 1, 0, True), (1, 512, 256, 256, 1, 1, False), (1, 1024, 256, 128, 1, 0, False),
    ]
    convt = [(128, 320), (64, 640), (32, 1280)]

    # KB_SWEEP_ENV=HI3D_GEMM_GN sweeps the column-group width of the tile raster instead of the tile variant
    ENV = os.environ.get("KB_SWEEP_ENV", "HI3D_GEMM_VARIANT")

    def run(label, fl, fn, vs):
        best = {}
        for rnd_ in range(2):
            for v in vs:
                if v == "h":
                    os.environ.pop(ENV, None)
                else:
                    os.environ[ENV] = v
                ops.gemm_reload_env()          # (the dispatch caches its switches per process)
                try:
                    ms = timeit(fn, iters=4, warm=1)
                except Exception as e:  # noqa: BLE001
                    ms = float("nan")
                best[v] = min(best.get(v, 1e9), ms)
        os.environ.pop(ENV, None)
        ops.gemm_reload_env()
        cells = "  ".join(f"{v}:{best[v]:7.3f}ms {fl / best[v] / 1e9:6.0f}TF" for v in vs)
        print(f"  {label:52s} {cells}", flush=True)

    # KB_HALF=1: the launches of the two-stream step -- the CFG halves of the two largest levels run as half-batch launches
    half = os.environ.get("KB_HALF", "0") == "1"
    if half:
        dense = [(M // 2 if M >= 131072 else M, N, K, kind) for M, N, K, kind in dense]
        conv = [(Fr // 2 if (Fr == 32 and H >= 64) else Fr, H, Cin, Cout, st, up, res) for Fr, H, Cin, Cout, st, up, res in conv]
    # KB_DIV=8: the launches of ONE RANK of a clip-parallel job (cfg 2 x sp 4: an eighth of every M; VAE rows dropped)
    div = int(os.environ.get("KB_DIV", "1"))
    if div > 1:
        dense = [(M // div, N, K, kind) for M, N, K, kind in dense if M % (div * 128) == 0 and N != 128 and N != 16384]
        conv = [(Fr // div, H, Cin, Cout, st, up, res) for Fr, H, Cin, Cout, st, up, res in conv if Fr == 32]
    print("== dense" + (" (half-batch launches of the two largest levels)" if half else "") + (f" (M / {div})" if div > 1 else ""))
    for M, N, K, kind in dense:
        A, W = rb(M, K), rb(N, K)
        bias = torch.randn(N, device=dev)
        geglu = kind == "geglu"
        R1 = rb(M, N) if kind == "res" else None
        out = torch.empty((M, N // 2 if geglu else N), device=dev, dtype=torch.bfloat16)
        vs = variants + (["3", "5"] if geglu and not sys.argv[2:] and ENV == "HI3D_GEMM_VARIANT" else [])
        run(f"dense M={M} N={N} K={K} {kind}", 2.0 * M * N * K,
            lambda: ops.gemm(A, W, M=M, N=N, K=K, bias=bias, R1=R1, geglu=geglu, out=out), vs)
        del A, W, out, R1
    print("== conv3x3")
    for Fr, H, Cin, Cout, stride, up, res in conv:
        Ho = H * 2 if up else H // stride
        M, K = Fr * Ho * Ho, 9 * Cin
        A, W = rb(Fr * H * H, Cin), rb(Cout, K)
        bias = torch.randn(Cout, device=dev)
        R1 = rb(M, Cout) if res else None
        out = torch.empty((M, Cout), device=dev, dtype=torch.bfloat16)
        geo = dict(Hin=H, Win=H, Cin=Cin, Hout=Ho, Wout=Ho, stride=stride, up2x=up)
        run(f"conv F={Fr} {H}->{Ho} {Cin}->{Cout}{' +R1' if res else ''}", 2.0 * M * Cout * K,
            lambda: ops.gemm(A, W, M=M, N=Cout, K=K, bias=bias, R1=R1, conv3x3=geo, out=out), variants)
        del A, W, out, R1
    print("== conv temporal")
    for H, C in convt:
        M, K = 32 * H * H // div, 3 * C
        HWc = H * H if div == 1 else H * H * 2 // div       # (a rank holds all 16 frames of 2 / div of the pixels of one CFG half)
        A, W, R2 = rb(M, C), rb(C, K), rb(M, C)
        bias = torch.randn(C, device=dev)
        out = torch.empty((M, C), device=dev, dtype=torch.bfloat16)
        run(f"convt H={H} C={C} HW={HWc} +R2", 2.0 * M * C * K,
            lambda: ops.gemm(A, W, M=M, N=C, K=K, bias=bias, R2=R2, convt3=dict(T=16, HW=HWc, Cin=C), out=out), variants)
        del A, W, out, R2


if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    bench_sweep()


def bench_conv_one():
    """kbench.py conv1 F H Cin Cout [res] : one stride-1 conv3x3 shape, few iterations (for rocprofv3 --pmc); prints
    the algorithmic bytes of a launch (input once + weights + output [+ residual]) next to the time."""
    Fr, H, Cin, Cout = (int(a) for a in sys.argv[2:6])
    res = len(sys.argv) > 6 and sys.argv[6] == "res"
    M, K = Fr * H * H, 9 * Cin
    A, W = rb(M, Cin), rb(Cout, K)
    bias = torch.randn(Cout, device=dev)
    R1 = rb(M, Cout) if res else None
    out = torch.empty((M, Cout), device=dev, dtype=torch.bfloat16)
    geo = dict(Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, up2x=0)
    ms = timeit(lambda: ops.gemm(A, W, M=M, N=Cout, K=K, bias=bias, R1=R1, conv3x3=geo, out=out), iters=3, warm=1)
    alg = 2.0 * (M * Cin + Cout * K + M * Cout * (2 if res else 1))
    print(f"conv F={Fr} H={H} {Cin}->{Cout}{' +R1' if res else ''}: {ms:.3f} ms {2.0 * M * Cout * K / ms / 1e9:.1f} TFLOP/s  "
          f"algorithmic {alg / 1e6:.1f} MB/launch (weights {2.0 * Cout * K / 1e6:.1f} MB)")


if len(sys.argv) > 1 and sys.argv[1] == "conv1":
    bench_conv_one()


def bench_temporal_ab():
    """usage: kbench.py tattn   -- hi3d_attn_temporal_d64's three kernels on the same inputs (HI3D_ATTNT_MFMA = 0: VALU kernel of
    rounds 1-3, 1: default dispatch -- one-block matrix-core kernel for T <= 16, two-block above --, 2: two-block kernel for every
    T), at the stage-2 levels for 16 and 32 views (B = 2 clips)."""
    for T in (16, 32):
        for ds, C in ((1, 320), (2, 640), (4, 1280)):
            S, H = (128 // ds) ** 2, C // 64
            qkv = rb(2 * T * S, 3 * C)
            line = f"  T={T} S={S:6d} H={H:2d}:"
            for mode in ("0", "1", "2"):
                os.environ["HI3D_ATTNT_MFMA"] = mode
                ms = timeit(lambda: ops.attention_temporal_fused_qkv(qkv, 2, T, S, H))
                line += f"   mode {mode}: {ms:7.3f} ms {2.0 * 4 * 2 * T * S * C / ms / 1e6:7.1f} GB/s"
            os.environ.pop("HI3D_ATTNT_MFMA", None)
            print(line, flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "tattn":
    bench_temporal_ab()


def bench_vae_sweep():
    """kbench.py vaesweep [variants...] : the VAE decoder's conv shapes (one frame) under each tile variant (HI3D_GEMM_VARIANT,
    re-read through hi3d_gemm_reload_env); 'h' = the host heuristic."""
    variants = sys.argv[2:] or ["h", "0", "1", "2", "3", "6", "8"]
    shapes = ((1, 1024, 128, 128, True), (1, 1024, 128, 128, False), (1, 1024, 256, 128, False), (1, 512, 256, 256, True),
              (1, 512, 512, 256, False), (1, 256, 512, 512, True), (4, 1024, 128, 128, True), (1, 128, 512, 512, True),
              (1, 128, 512, 512, False), (16, 512, 128, 128, True))
    for Fr, H, Cin, Cout, res in shapes:
        M, K = Fr * H * H, 9 * Cin
        A, W = rb(M, Cin), rb(Cout, K)
        bias = torch.randn(Cout, device=dev)
        R1 = rb(M, Cout) if res else None
        out = torch.empty((M, Cout), device=dev, dtype=torch.bfloat16)
        geo = dict(Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, up2x=0)
        ref = None
        line = f"conv F={Fr} H={H} {Cin}->{Cout}{' +R1' if res else ''}:"
        for v in variants:
            if v == "h":
                os.environ.pop("HI3D_GEMM_VARIANT", None)
            else:
                os.environ["HI3D_GEMM_VARIANT"] = v
            ops.gemm_reload_env()
            try:
                ms = timeit(lambda: ops.gemm(A, W, M=M, N=Cout, K=K, bias=bias, R1=R1, conv3x3=geo, out=out), iters=5, warm=2)
            except Exception as e:      # (a variant the shape does not admit)
                line += f"  {v}: n/a ({str(e)[:30]})"
                continue
            same = ""
            if ref is None:
                ref = out.clone()
            else:
                same = "" if torch.equal(ref, out) else " (differs)"
            line += f"  {v}: {ms:.3f} ms {2.0 * M * Cout * K / ms / 1e9:.0f} TF{same}"
        print(line, flush=True)
    os.environ.pop("HI3D_GEMM_VARIANT", None)
    ops.gemm_reload_env()


if len(sys.argv) > 1 and sys.argv[1] == "vaesweep":
    bench_vae_sweep()


def bench_up_phases():
    """kbench.py upphase [variants...] : the VAE decoder's up-sampling convs (one frame) as four placed phase launches under each
    tile variant; checks every variant against the planar + interleave form of the same tile.  CAUTION (round 6): one pass in a
    fixed order -- the first variant of a shape runs on cooler clocks and reads 5-10 % slower than the same tile measured last
    (profiles/r06w_up_phase_variants*.log: `h` = variant 3 there, 0.657 vs 0.582 ms); decide with an alternated A/B
    (tools/ab_lib_vae.sh), which is what refuted this sweep's apparent 11 % for the four-block tile on the phase launches."""
    from hi3d_hip.pack import pack_conv3x3_up_phases
    variants = sys.argv[2:] or ["h", "8", "3"]
    for Fr, H, C in ((1, 512, 256), (1, 256, 512), (1, 128, 512), (16, 256, 256), (16, 64, 640)):
        x = rb(Fr * H * H, C)
        w = (torch.randn((C, C, 3, 3)) * (9 * C) ** -0.5).to(torch.bfloat16).float()
        b = torch.randn(C, device=dev)
        wp = [p_.to(dev) for p_, _ in pack_conv3x3_up_phases(w)]
        line = f"up-conv F={Fr} {H}->{2 * H} C={C}:"
        fl = 2.0 * 4 * Fr * H * H * C * 4 * C
        for v in variants:
            if v == "h":
                os.environ.pop("HI3D_GEMM_VARIANT", None)
            else:
                os.environ["HI3D_GEMM_VARIANT"] = v
            ops.gemm_reload_env()
            ops._PHASE_PLACED_OK.clear()
            try:
                out = ops.upsample_conv_phases(x, wp, b, Fr, H, H, C, placed=True)
                placed_ok = ops._PHASE_PLACED_OK.get((x.device.index, Fr * H * H, H, H, C), True)
                ms = timeit(lambda: ops.upsample_conv_phases(x, wp, b, Fr, H, H, C, placed=True), iters=5, warm=2)
                ref = ops.upsample_conv_phases(x, wp, b, Fr, H, H, C, placed=False)
                line += f"  {v}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF ({'placed' if placed_ok else 'planar fallback'}{'' if torch.equal(out, ref) else ', DIFFERS from its planar form'})"
            except Exception as e:      # noqa: BLE001
                line += f"  {v}: n/a ({str(e)[:40]})"
        print(line, flush=True)
    os.environ.pop("HI3D_GEMM_VARIANT", None)
    ops.gemm_reload_env()
    ops._PHASE_PLACED_OK.clear()


if len(sys.argv) > 1 and sys.argv[1] == "upphase":
    bench_up_phases()

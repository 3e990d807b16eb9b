"""Tensor-level wrappers over the C ABI: torch supplies device memory and the
current HIP stream, every FLOP runs in libhi3d_hip.so.

Activations: torch.bfloat16, channels-last tokens [frames, H*W, C] (contiguous).
"""
import os

import torch

from . import lib as _l

_lib = _l.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


class Profiler:
    """Per-kernel timing with HIP events on the stream the kernels are launched on
    (torch's current stream), plus the algorithmic FLOPs / bytes of each launch.
    Enabled by bench.py / tests only; `None` (default) costs one attribute read per op."""

    def __init__(self):
        self.records = []       # (family, flops, bytes, start_event, end_event)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, family, flops, nbytes, start, detail=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((family, float(flops), float(nbytes), start, e, detail))

    def summary(self, by_shape=False):
        """{family: dict(ms, launches, flops, bytes)} -- call after a device sync.
        by_shape: key on the launch's shape string instead (where the op records one)."""
        out = {}
        for fam, fl, nb, s, e, detail in self.records:
            if by_shape:
                fam = f"{fam} {detail}" if detail else fam
            d = out.setdefault(fam, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            d["ms"] += s.elapsed_time(e); d["launches"] += 1; d["flops"] += fl; d["bytes"] += nb
        return out


PROFILER = None   # set to a Profiler() to record


def _chk_dev(*ts):
    """Every operand on ONE GPU, and that GPU is the current device: the launch goes to
    torch.cuda.current_stream() of the current device, so a tensor of another GPU would be addressed from
    the wrong stream (the runtimes wrap their passes in `torch.cuda.device(...)`)."""
    idx = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _l.Hi3dError("hi3d ops need device tensors (got a CPU tensor); there is no CPU path")
        if idx is None:
            idx = t.device.index
        elif t.device.index != idx:
            raise _l.Hi3dError(f"hi3d op operands live on different GPUs (cuda:{idx} and cuda:{t.device.index})")
    if idx is not None and idx != torch.cuda.current_device():
        raise _l.Hi3dError(f"operands are on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: "
                           "run under `torch.cuda.device(...)` / torch.cuda.set_device")


def gemm_desc(A, W, *, M, N, K, out=None, bias=None, rowvec=None, ldrv=0, rows_per_group=1,
         R1=None, R2=None, a1=None, a2=None, out_fp32=False, geglu=False,
         lda=None, ldw=0, conv3x3=None, convt3=None, tile_n=0, A2=None, K1=0, lda2=None, w_group_stride=0):
    """(hi3d_gemm_desc, out) of gemm(...) with the same arguments -- built, not launched (gemm() launches it; the ISA stress
    tool hands it to hi3d_debug_gemm_launch_info).  out[M, N(/2 if geglu)] = epilogue(A (*) W^T).  See include/hi3d_hip.h.

    conv3x3 = dict(Hin, Win, Cin, Hout, Wout, stride, up2x); convt3 = dict(T, HW, Cin).
    """
    _chk_dev(A, W, out, bias, rowvec, R1, R2, a1, a2, A2)
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=A.device, dtype=torch.float32 if out_fp32 else torch.bfloat16)
    d = _l.GemmDesc()
    d.A, d.W, d.bias, d.rowvec = _p(A), _p(W), _p(bias), _p(rowvec)
    d.R1, d.R2, d.a1, d.a2, d.out = _p(R1), _p(R2), _p(a1), _p(a2), _p(out)
    d.M, d.N, d.K = M, N, K
    d.lda = (K if A2 is None else K1) if lda is None else lda
    if A2 is not None:                       # logical A = [A | A2]: columns [0, K1) from A, [K1, K) from A2
        d.A2, d.K1, d.lda2 = _p(A2), K1, (K - K1 if lda2 is None else lda2)
    d.ldo = out.stride(0) if out.dim() == 2 else n_out
    d.ldr1 = R1.stride(-2) if R1 is not None else 0
    d.ldr2 = R2.stride(-2) if R2 is not None else 0
    d.ldrv = ldrv
    d.ldw = ldw
    d.rows_per_group = rows_per_group
    d.epi = _l.EPI_GEGLU if geglu else _l.EPI_AFFINE
    d.out_fp32 = 1 if out_fp32 else 0
    d.tile_n = tile_n
    d.w_group_stride = int(w_group_stride)
    if conv3x3 is not None:
        d.amode = _l.A_CONV3X3
        for k in ("Hin", "Win", "Cin", "Hout", "Wout", "stride", "up2x"):
            setattr(d, k, int(conv3x3[k]))
        d.pad_br_only = int(conv3x3.get("pad_br_only", 0))
        taps = conv3x3.get("taps")                   # a tap subset (ky*3 + kx each): K = len(taps) * Cin
        if taps:
            d.conv_ntap = len(taps)
            d.conv_taps = sum(int(t) << (4 * k) for k, t in enumerate(taps))
        if conv3x3.get("phase") is not None:         # (a, b): rows go to pixel (2 i + a, 2 j + b) of `out`, the full 2x image
            d.conv_phase = 1 + 2 * int(conv3x3["phase"][0]) + int(conv3x3["phase"][1])
    elif convt3 is not None:
        d.amode = _l.A_CONVT3
        d.T, d.HW, d.Cin = int(convt3["T"]), int(convt3["HW"]), int(convt3["Cin"])
    else:
        d.amode = _l.A_DENSE
    return d, out


_GEMM_WS = {}        # (device index, stream handle) -> the split-K scratch registered for that stream (held for the life of the process)


def _ensure_gemm_workspace(dev, stream=None):
    """Register hi3d_gemm_bf16's split-K scratch for (`dev`, the current stream -- or `stream`) once (HI3D_GEMM_WS_MB, default
    96 MiB = 8 partial tiles of the largest launch that is ever split; 0 = none).  One buffer PER STREAM: two split-K GEMMs in
    flight on two streams of a GPU must not share partial tiles (hi3d_gemm_set_workspace_for_stream).  Not during a graph
    capture: the allocation would belong to the capture's pool -- fused_step registers its capture stream before it captures."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    handle = (torch.cuda.current_stream(idx) if stream is None else stream).cuda_stream
    if (idx, handle) in _GEMM_WS:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    mb = int(os.environ.get("HI3D_GEMM_WS_MB", "96"))
    ws = torch.empty(mb << 20, dtype=torch.uint8, device=dev) if mb > 0 else None
    if ws is not None:
        with torch.cuda.device(idx):
            rc = _lib.hi3d_gemm_set_workspace_for_stream(_p(ws), mb << 20, handle)
        if rc != 0:          # all per-stream slots taken (a process cycling through > 63 streams): this stream does not split
            import warnings
            warnings.warn("hi3d: no split-K scratch slot left for this stream (63 per device); its long-K GEMMs run unsplit")
            ws = None
    _GEMM_WS[(idx, handle)] = ws


def gemm(A, W, *, M, N, K, out=None, bias=None, rowvec=None, ldrv=0, rows_per_group=1,
         R1=None, R2=None, a1=None, a2=None, out_fp32=False, geglu=False,
         lda=None, ldw=0, conv3x3=None, convt3=None, tile_n=0, A2=None, K1=0, lda2=None, gn=None, w_group_stride=0, gn_at=None):
    """out[M, N(/2 if geglu)] = epilogue(A (*) W^T).  See include/hi3d_hip.h.

    conv3x3 = dict(Hin, Win, Cin, Hout, Wout, stride, up2x); convt3 = dict(T, HW, Cin).
    A2 / K1: two-source dense A -- logical A = [A[:, :K1] | A2[:, :K - K1]] (the decoder's skip concat, never materialised).
    gn = (inst, P): the output is the input of a GroupNorm(32) over `inst` instances of P rows -- ask the producer to emit the
    norm's partial sums from its accumulators (hi3d_gemm_desc.gn_partial).  Returns (out, ws): ws is the GroupNorm workspace
    holding them (pass it to groupnorm_silu(..., partials=ws): the statistics pass is skipped), or None when this launch
    cannot provide them (the caller then runs the plain groupnorm_silu).
    gn_at = (ws, first): like gn, with the caller's workspace -- this launch's M / 64 row blocks of partial sums go to block
    `first` onwards (float offset 64 * first) of ws: launches that each produce PART of a tensor's rows (the four phase launches
    of an up-sampling conv).  Returns (out, ws or None)."""
    d, out = gemm_desc(A, W, M=M, N=N, K=K, out=out, bias=bias, rowvec=rowvec, ldrv=ldrv, rows_per_group=rows_per_group, R1=R1, R2=R2,
                       a1=a1, a2=a2, out_fp32=out_fp32, geglu=geglu, lda=lda, ldw=ldw, conv3x3=conv3x3, convt3=convt3, tile_n=tile_n,
                       A2=A2, K1=K1, lda2=lda2, w_group_stride=w_group_stride)
    _ensure_gemm_workspace(A.device)
    n_out = N // 2 if geglu else N
    gn_ws = None
    if gn is not None and GN_FUSED and gn[1] % 64 == 0 and gn[0] * gn[1] == M:
        gn_ws = _gn_workspace(A.device, gn[0], gn[1], n_out)
        d.gn_partial = _p(gn_ws)
    elif gn_at is not None and GN_FUSED and M % 64 == 0:
        gn_ws = gn_at[0]
        d.gn_partial = gn_ws.data_ptr() + 4 * 64 * int(gn_at[1])
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_gemm_bf16(d, _stream()), "hi3d_gemm_bf16")
    if gn_ws is not None and _lib.hi3d_gemm_last_gn_fused() != 1:        # (asked after the launch: one pass through the dispatch)
        gn_ws = None
    if prof:
        fam = ("gemm_conv3x3" if conv3x3 is not None else "gemm_convt3" if convt3 is not None else "gemm_dense")
        nres = (R1 is not None) + (R2 is not None)
        epi = "geglu" if geglu else "+".join(x for x, on in (("b", bias is not None), ("rv", rowvec is not None),
                                                            ("R1", R1 is not None), ("R2", R2 is not None)) if on)
        # algorithmic bytes: every input element ONCE (the conv gathers read a pixel for up to 9 taps, but
        # im2col never exists: M_in * Cin, not M * K), the weights, the output and the residual tiles
        if conv3x3 is not None:
            a_elems = (M // (d.Hout * d.Wout)) * d.Hin * d.Win * d.Cin
            geo = f" {d.Hin}x{d.Win}->{d.Hout}x{d.Wout}"
        elif convt3 is not None:
            a_elems, geo = M * d.Cin, f" T={d.T}"
        else:
            a_elems, geo = M * K, ""
        osz = 2.0 if out_fp32 else 1.0
        prof.end(fam, 2.0 * M * N * K, 2.0 * (a_elems + N * K + M * n_out * (osz + nres)), t0,
                 detail=f"M={M} N={N} K={K}{geo} {epi}")
    return out if (gn is None and gn_at is None) else (out, gn_ws)


def gemm_reload_env():
    """The GEMM dispatch reads its HI3D_GEMM_* / HI3D_GN_FUSED_OFF switches from the environment once per process; call this
    after changing one of them in a running process (tests, tools/kbench.py sweeps)."""
    _lib.hi3d_gemm_reload_env()


def transpose_v(v_view, B, H, S, ldv):
    """v_view: tensor whose data_ptr is V[b=0,s=0,h=0,d=0]; returns vt [B,H,64,S_pad]."""
    S_pad = (S + 63) // 64 * 64
    vt = torch.empty((B, H, 64, S_pad), device=v_view.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_transpose_v(_p(v_view), _p(vt), B, H, S, S_pad, ldv, _stream()), "hi3d_transpose_v")
    if prof:
        prof.end("transpose_v", 0.0, 2.0 * 2 * B * H * S * 64, t0)
    return vt


def attention_d64(q, k, vt, B, H, S_q, S_kv, ldq, ldk, scale, out=None):
    _chk_dev(q, k, vt, out)
    if out is None:
        out = torch.empty((B * S_q, H * 64), device=q.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_d64(_p(q), _p(k), _p(vt), _p(out), B, H, S_q, S_kv, ldq, ldk,
                                vt.shape[-1], out.stride(0), float(scale), _stream()), "hi3d_attn_d64")
    if prof:
        prof.end("attn_d64", 4.0 * B * H * S_q * S_kv * 64, 2.0 * B * H * 64 * (2 * S_q + 2 * S_kv), t0)
    return out


def attention_d64_v(q, k, v, B, H, S_q, S_kv, ldq, ldk, ldv, scale, out=None):
    """attention_d64 with V ROW-major (v: tensor whose data_ptr is V[b=0, s=0, h=0, d=0], row pitch ldv): the kernel forms the
    V^T fragments with gfx950's transposing LDS read -- no transpose pass, no V^T buffer."""
    _chk_dev(q, k, v, out)
    if out is None:
        out = torch.empty((B * S_q, H * 64), device=q.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_d64_v(_p(q), _p(k), _p(v), _p(out), B, H, S_q, S_kv, ldq, ldk, ldv, out.stride(0), float(scale), _stream()),
             "hi3d_attn_d64_v")
    if prof:
        prof.end("attn_d64", 4.0 * B * H * S_q * S_kv * 64, 2.0 * B * H * 64 * (2 * S_q + 2 * S_kv), t0)
    return out


def attention_d512(qkv, B, S, scale=None, out=None):
    """Single-head attention of head dim 512 on a fused [B*S, 3*512] q | k | v projection (the VAE mid block): flash-style,
    no score matrix in memory.  Returns [B*S, 512]."""
    _chk_dev(qkv, out)
    assert qkv.shape == (B * S, 3 * 512) and qkv.stride(1) == 1
    if out is None:
        out = torch.empty((B * S, 512), device=qkv.device, dtype=torch.bfloat16)
    scale = 512 ** -0.5 if scale is None else scale
    ld = qkv.stride(0)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_d512(_p(qkv), _p(qkv[:, 512:]), _p(qkv[:, 1024:]), _p(out), B, S, ld, ld, ld, out.stride(0),
                                 float(scale), _stream()), "hi3d_attn_d512")
    if prof:
        prof.end("attn_d512", 4.0 * B * S * S * 512, 2.0 * B * S * 512 * 4, t0)
    return out


Q_PRESCALE = 64 ** -0.5 * 1.4426950408889634   # softmax scale * log2(e): folded into to_q by pack_qkv(..., q_scale=)
# HI3D_ATTN_VROW=0: the round-1..3 form (transpose_v pass + pre-transposed V^T operand) -- A/B switch, both are HIP paths
ATTN_VROW = os.environ.get("HI3D_ATTN_VROW", "1") != "0"


def self_attention_fused_qkv(qkv, B, S, H, scale=None, q_prescaled=False):
    """qkv: [B*S, 3*H*64] bf16 (q | k | v column blocks). Returns [B*S, H*64].
    q_prescaled: the q block was produced by weights carrying Q_PRESCALE (scale is then ignored)."""
    C = H * 64
    assert qkv.shape == (B * S, 3 * C) and qkv.is_contiguous()
    scale = 0.0 if q_prescaled else (64 ** -0.5 if scale is None else scale)
    if ATTN_VROW:
        return attention_d64_v(qkv, qkv[:, C:], qkv[:, 2 * C:], B, H, S, S, 3 * C, 3 * C, 3 * C, scale)
    vt = transpose_v(qkv[:, 2 * C:], B, H, S, 3 * C)
    return attention_d64(qkv, qkv[:, C:], vt, B, H, S, S, 3 * C, 3 * C, scale)


def self_attention_fused_qkv_fp8qk(qkv, B, S, H):
    """self_attention_fused_qkv(..., q_prescaled=True) with the score product on the fp8 matrix path (BASELINE
    config 5): q | k are quantised to e4m3 with MX block scales (one pass), S = Q K^T runs at twice the bf16
    MFMA rate, softmax and P V are unchanged.  qkv: [B*S, 3*H*64] bf16, q carrying Q_PRESCALE."""
    C = H * 64
    assert qkv.shape == (B * S, 3 * C) and qkv.is_contiguous()
    _chk_dev(qkv)
    vt = transpose_v(qkv[:, 2 * C:], B, H, S, 3 * C)
    ws = torch.empty(_lib.hi3d_attn_fp8_workspace_bytes(B, H, S), device=qkv.device, dtype=torch.uint8)
    out = torch.empty((B * S, C), device=qkv.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_quant_qk(_p(qkv), _p(ws), B, H, S, 3 * C, _stream()), "hi3d_attn_quant_qk")
    if prof:
        prof.end("quant_qk", 0.0, 2.0 * 2 * B * S * C + 2 * B * S * (C + 2 * H), t0)
        t0 = prof.begin()
    _l.check(_lib.hi3d_attn_d64_fp8qk(_p(ws), _p(vt), _p(out), B, H, S, vt.shape[-1], C, _stream()), "hi3d_attn_d64_fp8qk")
    if prof:
        prof.end("attn_d64_fp8qk", 4.0 * B * H * S * S * 64, B * H * 64.0 * (1 * S + 1 * S + 2 * S + 2 * S), t0)
    return out


def self_attention_fused_qkv_fp8(qkv, B, S, H):
    """Both attention products on the fp8 matrix path (the full form of BASELINE config 5): q | k quantised as in
    self_attention_fused_qkv_fp8qk, v quantised to e4m3 V^T tiles (one e8m0 exponent per d row and 64-key tile), P converted
    to e4m3 in registers; one MFMA per (32 x 32) block and key tile for each product.  qkv: [B*S, 3*H*64] bf16, q carrying
    Q_PRESCALE.  Reduced precision: own tolerance."""
    C = H * 64
    assert qkv.shape == (B * S, 3 * C) and qkv.is_contiguous()
    _chk_dev(qkv)
    ws = torch.empty(_lib.hi3d_attn_fp8_workspace_bytes(B, H, S), device=qkv.device, dtype=torch.uint8)
    ws_v = torch.empty(_lib.hi3d_attn_fp8_v_workspace_bytes(B, H, S), device=qkv.device, dtype=torch.uint8)
    out = torch.empty((B * S, C), device=qkv.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_quant_qk(_p(qkv), _p(ws), B, H, S, 3 * C, _stream()), "hi3d_attn_quant_qk")
    _l.check(_lib.hi3d_attn_quant_v(_p(qkv[:, 2 * C:]), _p(ws_v), B, H, S, 3 * C, _stream()), "hi3d_attn_quant_v")
    if prof:
        prof.end("quant_qkv", 0.0, 2.0 * 3 * B * S * C + 3 * B * S * (C + 2 * H), t0)
        t0 = prof.begin()
    _l.check(_lib.hi3d_attn_d64_fp8(_p(ws), _p(ws_v), _p(out), B, H, S, C, _stream()), "hi3d_attn_d64_fp8")
    if prof:
        prof.end("attn_d64_fp8", 4.0 * B * H * S * S * 64, B * H * 64.0 * (1 * S + 1 * S + 1 * S + 2 * S), t0)
    return out


def attention_temporal_fused_qkv(qkv, B, T, S, H, scale=None):
    """qkv: [(B*T*S), 3*H*64] bf16 in frame-major (b t s) row order."""
    C = H * 64
    assert qkv.shape == (B * T * S, 3 * C) and qkv.is_contiguous()
    scale = 64 ** -0.5 if scale is None else scale
    out = torch.empty((B * T * S, C), device=qkv.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_attn_temporal_d64(_p(qkv), _p(qkv[:, C:]), _p(qkv[:, 2 * C:]), _p(out),
                                         B, T, S, H, 3 * C, C, float(scale), _stream()),
             "hi3d_attn_temporal_d64")
    if prof:
        prof.end("attn_temporal", 4.0 * B * S * H * T * T * 64, 2.0 * 4 * B * T * S * C, t0)
    return out


_gn_ws = {}
# HI3D_GN_FUSED=0: every GroupNorm runs its own statistics pass (rounds 1-3; A/B switch)
GN_FUSED = os.environ.get("HI3D_GN_FUSED", "1") != "0"


def _gn_workspace(dev, inst, P, C):
    """the per-(device, stream) GroupNorm scratch (partial sums + per-instance statistics), grown on demand"""
    n = _lib.hi3d_gn_workspace_floats(inst, P, C)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 16), device=dev, dtype=torch.float32)
        _gn_ws[key] = ws
    return ws


def groupnorm_silu(x, gamma, beta, inst, P, C, eps, silu=True, out=None, x2=None, partials=None):
    """x: bf16 [inst*P, C] contiguous. 32 groups, statistics over (P, C/32).
    x2: second source -- the normalised tensor is the channel concatenation [x | x2] (C = C1 + C2 total channels, x holds
    C1 = x.shape[-1] of them), read in place; the result is the full-width [inst*P, C] tensor.
    partials: the workspace a producing gemm(..., gn=(inst, P)) filled with this tensor's partial sums: finalize + apply only.
    (Tried, MI355X: processing runs of instances that fit the 256 MB Infinity Cache so that the second read of x
    hits it -- 0.204 -> 0.236-0.248 ms at [32 x 16384 x 320]: the smaller grids cost more than the re-read.)"""
    _chk_dev(x, gamma, beta, out, x2)
    if x2 is None:
        assert x.is_contiguous() and x.numel() == inst * P * C
        if out is None:
            out = torch.empty_like(x)
    else:
        C1 = x.numel() // (inst * P)
        assert x.is_contiguous() and x2.is_contiguous() and x.numel() == inst * P * C1 and x2.numel() == inst * P * (C - C1)
        if out is None:
            out = torch.empty((inst * P, C), device=x.device, dtype=torch.bfloat16)
    return _groupnorm_silu(x, gamma, beta, inst, P, C, eps, silu, out, x2, partials)


def _groupnorm_silu(x, gamma, beta, inst, P, C, eps, silu, out, x2=None, partials=None):
    ws = _gn_workspace(x.device, inst, P, C) if partials is None else partials
    prof = PROFILER
    t0 = prof.begin() if prof else None
    if partials is not None:
        assert x2 is None
        _l.check(_lib.hi3d_groupnorm_silu_from_partials(_p(x), _p(out), _p(gamma), _p(beta), _p(ws), inst, P, C,
                                                        float(eps), 1 if silu else 0, _stream()), "hi3d_groupnorm_silu_from_partials")
    elif x2 is None:
        _l.check(_lib.hi3d_groupnorm_silu(_p(x), _p(out), _p(gamma), _p(beta), _p(ws), inst, P, C,
                                          float(eps), 1 if silu else 0, _stream()), "hi3d_groupnorm_silu")
    else:
        C1 = x.numel() // (inst * P)
        _l.check(_lib.hi3d_groupnorm_silu_cat2(_p(x), _p(x2), _p(out), _p(gamma), _p(beta), _p(ws), inst, P, C1, C - C1,
                                               float(eps), 1 if silu else 0, _stream()), "hi3d_groupnorm_silu_cat2")
    if prof:   # algorithmic bytes: read x once + write y once; the three-pass form reads x twice
        prof.end("groupnorm_silu", 0.0, 2.0 * 2 * inst * P * C, t0,
                 detail=f"inst={inst} P={P} C={C} {'from-partials' if partials is not None else 'cat2' if x2 is not None else '3-pass'}")
    return out


def groupnorm_silu_sharded(x, gamma, beta, inst, P, C, eps, allreduce_, world, silu=True, out=None):
    """GroupNorm(32)[+SiLU] of an instance whose P positions are this rank's share of world * P:
    local partial sums -> allreduce_(sums) (in place, SUM over the group) -> normalise with the global
    statistics.  x: bf16 [inst*P, C]."""
    _chk_dev(x, gamma, beta, out)
    assert x.is_contiguous() and x.numel() == inst * P * C
    if out is None:
        out = torch.empty_like(x)
    n = _lib.hi3d_gn_workspace_floats(inst, P, C)
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 16), device=x.device, dtype=torch.float32)
        _gn_ws[key] = ws
    sums = torch.empty((inst, 32, 2), device=x.device, dtype=torch.float64)
    _l.check(_lib.hi3d_groupnorm_partial_sums(_p(x), _p(ws), _p(sums), inst, P, C, _stream()), "hi3d_groupnorm_partial_sums")
    allreduce_(sums)
    _l.check(_lib.hi3d_groupnorm_apply_sums(_p(x), _p(out), _p(gamma), _p(beta), _p(sums), _p(ws), inst, P, C,
                                            int(world) * P * (C // 32), float(eps), 1 if silu else 0, _stream()),
             "hi3d_groupnorm_apply_sums")
    return out


# HI3D_GN_FOLD=1: the transformer's GroupNorm as a per-frame rescaling of proj_in's weights.  Opt-in: it removes the norm's apply
# pass (GroupNorm family 9.85 -> 9.40 ms per stage-2 step) but the step's wall time does not move (198.96 / 198.69 vs 198.49 /
# 199.12 ms: those passes already ran under the other CFG half's matrix-core kernels) -- profiles/r04s_gn_fold_ab.log.  Round 6:
# with the statistics from the producer's partial sums the norm reads no activation at all (GroupNorm family 8.48 -> 8.02 ms) --
# and the step still does not move (194.78 vs 194.81 ms over three alternations, profiles/r06k_ab_gn_fold_from_partials.log).
GN_FOLD = os.environ.get("HI3D_GN_FOLD", "0") == "1"


def groupnorm_fold_linear(x, gamma, beta, inst, P, C, eps, W, bias, N, partials=None):
    """The statistics of GroupNorm(32; no activation) over x [inst * P, C] folded into the linear layer (W [N, C] bf16 K-major,
    bias [N] fp32 or None) that consumes the normalised tensor: returns (Wf [inst, N, C] bf16, biasf [inst, N] fp32) for
    gemm(x, Wf, ..., rowvec=biasf, rows_per_group=P, w_group_stride=N * C).  See include/hi3d_hip.h.
    partials: the workspace in which the producer of x left its partial sums (gemm(..., gn=)): x is then not read at all.
    Reference: SpatialTransformer.norm + proj_in, sgm/modules/attention.py:702-712."""
    _chk_dev(x, gamma, beta, W, bias)
    assert W.dtype == torch.bfloat16 and W.is_contiguous() and W.shape[-1] == C and W.shape[0] == N
    ws = _gn_workspace(x.device, inst, P, C) if partials is None else partials
    Wf = torch.empty((inst, N, C), device=x.device, dtype=torch.bfloat16)
    biasf = torch.empty((inst, N), device=x.device, dtype=torch.float32)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    if partials is not None and P % 64 == 0:
        _l.check(_lib.hi3d_groupnorm_fold_linear_from_partials(_p(ws), _p(gamma), _p(beta), float(eps), inst, P, C, _p(W), C, _p(bias), N,
                                                               _p(Wf), _p(biasf), _stream()), "hi3d_groupnorm_fold_linear_from_partials")
    else:
        _l.check(_lib.hi3d_groupnorm_fold_linear(_p(x), _p(ws), _p(gamma), _p(beta), float(eps), inst, P, C, _p(W), C, _p(bias), N,
                                                 _p(Wf), _p(biasf), _stream()), "hi3d_groupnorm_fold_linear")
    if prof:
        prof.end("groupnorm_silu", 0.0, 2.0 * inst * P * C, t0)
    return Wf, biasf


# HI3D_UP_PHASE_PLACED=0: planar phase images + hi3d_permute_rows (rounds 4-5).  Default (round 6): the four phase convolutions of an
# up-sampling conv store their rows straight into the 2x image (hi3d_gemm_desc.conv_phase) -- round 4's parked experiment, rebuilt
# with the per-pass base rows computed BEFORE the K loop (the parked form divided inside the store loop and left a few rows of
# store pass 0 unwritten); bit-identical to the planar form at every UNet / VAE shape, 50 repetitions (tests/test_kernels_gpu.py)
UP_PHASE_PLACED = os.environ.get("HI3D_UP_PHASE_PLACED", "1") != "0"
_PHASE_PLACED_OK = {}


def upsample_conv_phases(x, w_phases, bias, frames, H, Wd, C, placed=None, gn=False):
    """Upsample(nearest 2x) + conv3x3 pad 1 on x [frames * H * Wd, C] (channels-last rows) -> [frames * 2H * 2Wd, C] as four 2x2
    phase convolutions on the low-resolution image (pack.pack_conv3x3_up_phases: w_phases[a * 2 + b] = [C, 4 * C]): the phases are
    written planar [a][b][(f i)][j] and interleaved to [(f i)][a][j][b] by hi3d_permute_rows -- or, `placed` (default:
    HI3D_UP_PHASE_PLACED) and where the launch qualifies (wide tile, Wd % 16 == 0), every phase stores its rows straight into the 2x
    image (hi3d_gemm_desc.conv_phase).
    gn=True: returns (out, ws) -- ws the GroupNorm workspace with the partial sums of `out` as ONE instance of 4 * frames * H * Wd
    rows (a one-frame VAE call: pass it to groupnorm_silu(..., 1, rows, C, partials=ws)), each placed phase launch filling its quarter
    of the row blocks (the order of the blocks inside an instance does not matter to the sums) -- or None when frames != 1 or a
    launch cannot provide them."""
    Ml = frames * H * Wd
    placed = UP_PHASE_PLACED if placed is None else placed
    geo = lambda ph: dict(Hin=H, Win=Wd, Cin=C, Hout=H, Wout=Wd, stride=1, up2x=0,
                          taps=tuple(((ph >> 1) + dy) * 3 + ((ph & 1) + dx) for dy in (0, 1) for dx in (0, 1)))
    key = (x.device.index, Ml, H, Wd, C)
    if placed and _PHASE_PLACED_OK.get(key, True):
        out = torch.empty((4 * Ml, C), device=x.device, dtype=torch.bfloat16)
        ws = _gn_workspace(x.device, 1, 4 * Ml, C) if (gn and frames == 1 and Ml % 64 == 0 and GN_FUSED) else None
        try:
            for ph in range(4):
                kw = dict(M=Ml, N=C, K=4 * C, bias=bias, out=out, conv3x3=dict(geo(ph), phase=(ph >> 1, ph & 1)))
                if ws is None:
                    gemm(x, w_phases[ph], **kw)
                elif gemm(x, w_phases[ph], gn_at=(ws, ph * (Ml // 64)), **kw)[1] is None:
                    ws = None
            return (out, ws) if gn else out
        except _l.Hi3dError:
            if ph != 0:                              # (the shape check is the same for the four phases: it fails at the first or never)
                raise
            _PHASE_PLACED_OK[key] = False
    tmp = torch.empty((4, Ml, C), device=x.device, dtype=torch.bfloat16)
    for ph in range(4):
        gemm(x, w_phases[ph], M=Ml, N=C, K=4 * C, bias=bias, out=tmp[ph], conv3x3=geo(ph))
    out = permute_rows(tmp, (2, 2, frames * H, Wd), (2, 0, 3, 1)).reshape(4 * Ml, C)
    return (out, None) if gn else out


def layernorm(x, gamma, beta, R, C, eps=1e-5, addvec=None, rows_per_group=1, sum_out=None, out=None):
    _chk_dev(x, gamma, beta, addvec, sum_out, out)
    if out is None:
        out = torch.empty((R, C), device=x.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_layernorm(_p(x), _p(out), _p(sum_out), _p(gamma), _p(beta), _p(addvec),
                                 rows_per_group, R, C, float(eps), _stream()), "hi3d_layernorm")
    if prof:
        prof.end("layernorm", 0.0, 2.0 * R * C * (3 if sum_out is not None else 2), t0, detail=f"R={R} C={C}")
    return out


def concat_channels(a, b, rows, C0, C1):
    out = torch.empty((rows, C0 + C1), device=a.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_concat_channels(_p(a), _p(b), _p(out), rows, C0, C1, _stream()), "hi3d_concat_channels")
    if prof:
        prof.end("concat_channels", 0.0, 2.0 * 2 * rows * (C0 + C1), t0)
    return out


def timestep_embedding(t, dim, max_period=10000.0, out_bf16=False):
    t = t.to(torch.float32).contiguous()
    _chk_dev(t)
    out = torch.empty((t.numel(), dim), device=t.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _l.check(_lib.hi3d_timestep_embedding(_p(t), _p(out), t.numel(), dim, float(max_period),
                                          1 if out_bf16 else 0, _stream()), "hi3d_timestep_embedding")
    return out


def silu_to_bf16(x):
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_silu_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "hi3d_silu_f32_to_bf16")
    return out


def cfg_prepare(x, concat_uc, concat_c, Cp, sigma, out=None):
    """x: fp32 [T,4,H,W]; concat_*: fp32 [T,Cc,H,W] or None -> bf16 [2*T*HW, Cp]."""
    _chk_dev(x, concat_uc, concat_c, out)
    T, _, H, W = x.shape
    Cc = 0 if concat_c is None else concat_c.shape[1]
    if out is None:
        out = torch.empty((2 * T * H * W, Cp), device=x.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_cfg_prepare(_p(x), _p(concat_uc), _p(concat_c), _p(out), T, H * W, Cc, Cp,
                                   float(sigma), _stream()), "hi3d_cfg_prepare")
    return out


def sampler_step(x, net, scale, ldn, sigma, sigma_next):
    T, _, H, W = x.shape
    _l.check(_lib.hi3d_sampler_step(_p(x), _p(net), _p(scale), T, H * W, ldn, float(sigma),
                                    float(sigma_next), _stream()), "hi3d_sampler_step")
    return x


def cfg_update_x(x, tokens, sig, tvec, T, HW, Cp):
    """Per-step part of cfg_prepare: tokens[u][t][p][0:4] = x * c_in(sig[0]) for both CFG halves, tvec[0:2T] =
    ln(sig[0])/4; sigma is read on the device (graph-replayable)."""
    _chk_dev(x, tokens, sig, tvec)
    _l.check(_lib.hi3d_cfg_update_x(_p(x), _p(tokens), _p(sig), _p(tvec), T, HW, Cp, _stream()), "hi3d_cfg_update_x")
    return tokens


def sampler_step_dev(x, x_out, net, scale, sig, T, HW, ldn):
    """x_out = Euler-EDM update of x from the CFG-doubled network output `net`; sig = device [sigma, sigma_next]."""
    _chk_dev(x, x_out, net, scale, sig)
    _l.check(_lib.hi3d_sampler_step_dev(_p(x), _p(x_out), _p(net), _p(scale), _p(sig), T, HW, ldn, _stream()),
             "hi3d_sampler_step_dev")
    return x_out


def nchw_to_tokens(x, Cpad):
    """fp32 NCHW -> bf16 [N*HW, Cpad] (zero padded channels)."""
    x = x.to(torch.float32).contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N * H * W, Cpad), device=x.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_nchw_f32_to_nhwc_bf16(_p(x), _p(out), N, Cc, H * W, Cpad, _stream()),
             "hi3d_nchw_f32_to_nhwc_bf16")
    return out


def tokens_to_nchw(x, N, Cc, H, W, ldx):
    out = torch.empty((N, Cc, H, W), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_nhwc_to_nchw_f32(_p(x), _p(out), N, Cc, H * W, ldx,
                                        1 if x.dtype == torch.float32 else 0, _stream()),
             "hi3d_nhwc_to_nchw_f32")
    return out


def vae_latent_prepare(z, w, b, Cpad=64):
    """z fp32 [N,Cz,h,w]; w fp32 [Cz,Cz]; b fp32 [Cz] -> bf16 [N*h*w, Cpad]."""
    z = z.to(torch.float32).contiguous()
    N, Cz, H, W = z.shape
    out = torch.empty((N * H * W, Cpad), device=z.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_vae_latent_prepare(_p(z), _p(w), _p(b), _p(out), N, Cz, H * W, Cpad, _stream()),
             "hi3d_vae_latent_prepare")
    return out


def softmax_rows(s, R, N, ldp, scale):
    """s fp32 [R, N] -> bf16 [R, ldp] (columns >= N zero)."""
    p = torch.empty((R, ldp), device=s.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_softmax_rows(_p(s), _p(p), R, N, s.stride(0), ldp, float(scale), _stream()), "hi3d_softmax_rows")
    if prof:
        prof.end("softmax_rows", 0.0, 4.0 * R * N + 2.0 * R * ldp, t0)
    return p


_ACT = {"gelu": 0, "quick_gelu": 1, "relu": 2, "identity": 3}


def act_(x, kind):
    """In-place activation of a contiguous bf16 tensor: 'gelu' (exact erf), 'quick_gelu' (x sigmoid(1.702 x)), 'relu'."""
    _chk_dev(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise _l.Hi3dError("act_: contiguous bf16 tensor required")
    _l.check(_lib.hi3d_act_bf16(_p(x), x.numel(), _ACT[kind], _stream()), "hi3d_act_bf16")
    return x


def add_act(x, y=None, kind="identity", out=None):
    """out = act(x + y) on contiguous bf16 tensors of one shape (y optional; out may be x)."""
    _chk_dev(x, y, out)
    if out is None:
        out = torch.empty_like(x)
    for t in (x, y, out):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or t.numel() != x.numel()):
            raise _l.Hi3dError("add_act: contiguous bf16 tensors of one size required")
    _l.check(_lib.hi3d_add_act_bf16(_p(x), _p(y), _p(out), x.numel(), _ACT[kind], _stream()), "hi3d_add_act_bf16")
    return out


def dpt_stem_conv(x, w):
    """x fp32 [N, H, W, 3] channels-last, w fp32 [7, 7, 3, 64] (standardised) -> bf16 [N, ceil(H/2), ceil(W/2), 64]."""
    _chk_dev(x, w)
    if x.dtype != torch.float32 or w.dtype != torch.float32 or not x.is_contiguous() or not w.is_contiguous():
        raise _l.Hi3dError("dpt_stem_conv: contiguous fp32 tensors required")
    if x.dim() != 4 or x.shape[-1] != 3 or tuple(w.shape) != (7, 7, 3, 64):
        raise _l.Hi3dError("dpt_stem_conv: x [N,H,W,3], w [7,7,3,64]")
    N, H, W = x.shape[:3]
    y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, 64), device=x.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_dpt_stem_conv(_p(x), _p(w), _p(y), N, H, W, _stream()), "hi3d_dpt_stem_conv")
    return y


def pool2(x, mode):
    """x bf16 [N, H, W, C] -> [N, ceil(H/2), ceil(W/2), C]; mode 'max3' (3x3 max, SAME) or 'pick' (pixel 2oy, 2ox)."""
    _chk_dev(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.dim() != 4:
        raise _l.Hi3dError("pool2: contiguous bf16 [N,H,W,C] required")
    N, H, W, C = x.shape
    y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), device=x.device, dtype=torch.bfloat16)
    _l.check(_lib.hi3d_pool2_nhwc(_p(x), _p(y), N, H, W, C, {"max3": 0, "pick": 1}[mode], _stream()), "hi3d_pool2_nhwc")
    return y


def resize_bilinear(x, Ho, Wo, align_corners=False):
    """F.interpolate(mode='bilinear') on a channels-last [N, Hi, Wi, C] tensor, bf16 (C % 8 == 0) or fp32."""
    _chk_dev(x)
    if x.dtype not in (torch.bfloat16, torch.float32) or not x.is_contiguous() or x.dim() != 4:
        raise _l.Hi3dError("resize_bilinear: contiguous bf16 / fp32 [N,Hi,Wi,C] required")
    N, Hi, Wi, C = x.shape
    y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=x.dtype)
    _l.check(_lib.hi3d_resize_bilinear_nhwc(_p(x), _p(y), N, Hi, Wi, Ho, Wo, C, 1 if align_corners else 0,
                                            1 if x.dtype == torch.float32 else 0, _stream()), "hi3d_resize_bilinear_nhwc")
    return y


def permute_rows(x, dims, perm):
    """x: contiguous tensor viewed as [dims[0], dims[1], dims[2], dims[3], row] -> the same rows in the order perm (a new
    contiguous tensor [dims[perm[0]], ..., row]).  Rows must be a multiple of 16 bytes."""
    import ctypes as C
    _chk_dev(x)
    n = dims[0] * dims[1] * dims[2] * dims[3]
    if not x.is_contiguous() or x.numel() % n:
        raise _l.Hi3dError("permute_rows: contiguous tensor whose size is a multiple of prod(dims) required")
    row = x.numel() // n
    out = torch.empty([dims[p] for p in perm] + [row], device=x.device, dtype=x.dtype)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    _l.check(_lib.hi3d_permute_rows(_p(x), _p(out), (C.c_int32 * 4)(*dims), (C.c_int32 * 4)(*perm), row * x.element_size(), _stream()),
             "hi3d_permute_rows")
    if prof:
        prof.end("permute_rows", 0.0, 2.0 * x.numel() * x.element_size(), t0)
    return out


def resample_image(x, kind, scale=None, shift=None):
    """x fp32 [N, C, H, W] -> fp32 [N, C, Ho, Wo]: the separable resampling `kind` of hi3d_hip/resample.py (two banded
    passes, W then H) followed by the per-channel affine y * scale[c] + shift[c] (fused into the second pass)."""
    from . import resample
    _chk_dev(x, scale, shift)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise _l.Hi3dError("resample_image: fp32 [N,C,H,W] required")
    x = x.contiguous()
    N, C, H, W = x.shape
    (sh, wh), (sw, ww), (Ho, Wo) = resample.tables(kind, H, W, x.device)
    tmp = torch.empty((N, C, H, Wo), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_resample_axis(_p(x), _p(tmp), _p(sw), _p(ww), ww.shape[1], N * C * H, W, Wo, 1, 0, 0, 1, 1, _stream()),
             "hi3d_resample_axis")
    y = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_resample_axis(_p(tmp), _p(y), _p(sh), _p(wh), wh.shape[1], N * C, H, Ho, Wo, _p(scale), _p(shift), 1, C,
                                     _stream()), "hi3d_resample_axis")
    return y


def dpt_head_out(x, w, b):
    """x bf16 [M, C], w fp32 [C], b float -> fp32 [M]: relu(b + w . relu(x[m]))."""
    _chk_dev(x, w)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or w.dtype != torch.float32 or w.numel() != x.shape[-1]:
        raise _l.Hi3dError("dpt_head_out: x bf16 [M,C] contiguous, w fp32 [C]")
    M = x.numel() // x.shape[-1]
    y = torch.empty((M,), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_dpt_head_out(_p(x), _p(w), float(b), _p(y), M, x.shape[-1], _stream()), "hi3d_dpt_head_out")
    return y


def depth_normalize_unshuffle(d, s):
    """d fp32 [B, Hs, Ws] -> fp32 [B, s*s, Hs/s, Ws/s]: per-image min-max normalisation, then pixel-unshuffle."""
    _chk_dev(d)
    if d.dtype != torch.float32 or not d.is_contiguous() or d.dim() != 3:
        raise _l.Hi3dError("depth_normalize_unshuffle: contiguous fp32 [B,Hs,Ws] required")
    B, Hs, Ws = d.shape
    out = torch.empty((B, s * s, Hs // s, Ws // s), device=d.device, dtype=torch.float32)
    _l.check(_lib.hi3d_depth_normalize_unshuffle(_p(d), _p(out), B, Hs, Ws, s, _stream()), "hi3d_depth_normalize_unshuffle")
    return out


def l2_normalize_rows_(x):
    """x fp32 [R, C] contiguous: every row divided by its L2 norm, in place."""
    _chk_dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 2:
        raise _l.Hi3dError("l2_normalize_rows_: contiguous fp32 [R, C] required")
    _l.check(_lib.hi3d_l2_normalize_rows(_p(x), x.shape[0], x.shape[1], _stream()), "hi3d_l2_normalize_rows")
    return x


def vae_posterior(mom, wq, bq, noise, N, Cz, H, W):
    """mom fp32 [N*H*W, ldm] -> z fp32 [N, Cz, H, W] (sample if noise is given, else the mode)."""
    z = torch.empty((N, Cz, H, W), device=mom.device, dtype=torch.float32)
    _l.check(_lib.hi3d_vae_posterior(_p(mom), _p(wq), _p(bq), _p(noise), _p(z), N, Cz, H * W, mom.stride(0), _stream()),
             "hi3d_vae_posterior")
    return z


def v02_blend(lat, noise, z, alpha, sigma):
    for t in (lat, noise, z):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _l.Hi3dError("v02_blend: fp32 contiguous tensors required")
    _l.check(_lib.hi3d_v02_blend(_p(lat), _p(noise), _p(z), lat.numel(), float(alpha), float(sigma), _stream()), "hi3d_v02_blend")
    return lat


def time_mix_small(x, w, b, B, T, H, W, C):
    """x fp32 [(B*T*H*W), ldx] -> fp32 NCHW [B*T, C, H, W]: Conv3d (3,1,1) over the frame axis."""
    out = torch.empty((B * T, C, H, W), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_time_mix_small(_p(x), _p(w), _p(b), _p(out), B, T, H * W, C, x.stride(0), _stream()), "hi3d_time_mix_small")
    return out


def time_mix_small_k3(x, w, b, B, T, H, W, C):
    """x fp32 [(B*T*H*W), ldx] -> fp32 NCHW [B*T, C, H, W]: isotropic Conv3d (3,3,3), padding 1 (w [C, C, 3, 3, 3] fp32)."""
    assert w.is_contiguous() and tuple(w.shape) == (C, C, 3, 3, 3)
    out = torch.empty((B * T, C, H, W), device=x.device, dtype=torch.float32)
    _l.check(_lib.hi3d_time_mix_small_k3(_p(x), _p(w), _p(b), _p(out), B, T, H, W, C, x.stride(0), _stream()), "hi3d_time_mix_small_k3")
    return out


FFN_FUSED_WIDTHS = (320,)   # channel counts hi3d_ffn_geglu is built for


LN_FUSED = os.environ.get("HI3D_LN_FUSED", "1") != "0"   # the LayerNorm in front of a fused feed-forward runs inside its launch


def ffn_geglu(x, w1, b1, w2, b2, *, M, C, R1=None, R2=None, a1=None, a2=None, rows_per_group=1, out=None,
              ln=None, addvec=None, addvec_rows_per_group=1):
    """out[M, C] = (GEGLU(x w1^T + b1) w2^T + b2 + R1) [* a1 + a2 * R2] with the 4C hidden tensor kept on
    the CU (one launch instead of the GEGLU GEMM + the second GEMM).  See include/hi3d_hip.h.
    ln = (gamma, beta, eps): x is the raw residual stream and the kernel normalises its rows first (hi3d_ffn_geglu_ln);
    addvec [groups, C] fp32 is then added to x before the norm and to the residual R1 (the frame-position embedding)."""
    _chk_dev(x, w1, b1, w2, b2, R1, R2, a1, a2, out, addvec)
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.bfloat16)
    prof = PROFILER
    t0 = prof.begin() if prof else None
    if ln is not None:
        g, b, eps = ln
        _chk_dev(x, g, b)
        assert g.dtype == torch.float32 and b.dtype == torch.float32 and (addvec is None or addvec.dtype == torch.float32)
        assert addvec is None or (addvec.is_contiguous() and addvec.shape[-1] == C)
        _l.check(_lib.hi3d_ffn_geglu_ln(_p(x), _p(g), _p(b), float(eps), _p(addvec), int(addvec_rows_per_group),
                                        _p(w1), _p(b1), _p(w2), _p(b2), _p(R1), _p(R2), _p(a1), _p(a2), _p(out),
                                        M, C, x.stride(0), out.stride(0), R1.stride(0) if R1 is not None else 0,
                                        R2.stride(0) if R2 is not None else 0, rows_per_group, _stream()), "hi3d_ffn_geglu_ln")
    else:
        assert addvec is None
        _l.check(_lib.hi3d_ffn_geglu(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(R1), _p(R2), _p(a1), _p(a2), _p(out),
                                     M, C, x.stride(0), out.stride(0), R1.stride(0) if R1 is not None else 0,
                                     R2.stride(0) if R2 is not None else 0, rows_per_group, _stream()), "hi3d_ffn_geglu")
    if prof:
        nres = (R1 is not None) + (R2 is not None)
        prof.end("ffn_fused", 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C,
                 2.0 * (M * C * (2 + nres) + 12 * C * C), t0, detail=f"M={M} C={C}")
    return out

// torch.ops.hi3d.* -- the PyTorch-ROCm custom-op face of the C ABI in include/hi3d_hip.h.
//
// The reference has no native boundary on this path; its operator plug-in point is the attention-mode dictionary
// (sgm/modules/attention.py:457-460, video_attention.py:16-19) and the YAML `target:` registry (sgm/util.py:168-185).
// This shim makes the gfx950 kernels reachable there as ordinary dispatcher ops (binding B): a module
// written against torch tensors calls torch.ops.hi3d.self_attention(...) and never sees a raw pointer.  Host-only C++
// (no device code): shape / dtype / device violations become RuntimeError through TORCH_CHECK before anything is
// launched, outputs are fresh torch tensors (PyTorch owns all memory), every launch goes to the current HIP stream of
// the operands' device, and a non-zero status of the C ABI is raised with hi3d_last_error().  There is no CPU or ATen
// fallback: the ops are registered for the CUDA (= HIP) dispatch key only.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm tensors carry DeviceType::CUDA
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include "../../include/hi3d_hip.h"

namespace {

using at::Tensor;

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", hi3d_last_error()); }

void want_bf16_rows(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, ": device tensor required (this framework has no CPU path)");
  TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, ": bf16 required");
  TORCH_CHECK(t.dim() >= 2 && t.stride(-1) == 1, name, ": rows must be contiguous");
}
void want_f32(const Tensor& t, const char* name, int64_t n) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.numel() == n, name, ": contiguous fp32 [", n, "] on the GPU required");
}
const void* opt_ptr(const c10::optional<Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }
// every operand on the device of the first one: the launch runs under THAT device's guard and stream, a pointer of another
// GPU would be dereferenced there (fault / peer read) instead of raising
void same_device(const Tensor& x, std::initializer_list<const Tensor*> others, const char* name) {
  for (const Tensor* t : others)
    if (t && t->defined()) TORCH_CHECK(t->device() == x.device(), name, ": operands on different devices (", x.device(), " and ", t->device(), ")");
}
const Tensor* opt_t(const c10::optional<Tensor>& t) { return t.has_value() && t->defined() ? &*t : nullptr; }
// the C ABI takes int32 sizes
void fits_i32(std::initializer_list<int64_t> v, const char* name) {
  for (int64_t x : v) TORCH_CHECK(x >= 0 && x <= 0x7fffffffLL, name, ": a size (", x, ") does not fit the int32 of the C ABI");
}

// softmax(q k^T * scale) v for head dim 64 on a fused [B*S, 3*H*64] q|k|v projection: what
// MemoryEfficientCrossAttention.forward / CrossAttention.forward compute for self-attention (attention.py:332-336, 427-439)
Tensor self_attention(const Tensor& qkv, int64_t B, int64_t S, int64_t H, double scale) {
  want_bf16_rows(qkv, "hi3d::self_attention qkv");
  const int64_t C = H * 64;
  TORCH_CHECK(qkv.dim() == 2 && qkv.size(0) == B * S && qkv.size(1) == 3 * C, "hi3d::self_attention: qkv must be [B*S, 3*H*64]");
  TORCH_CHECK(scale > 0.0, "hi3d::self_attention: scale must be > 0");
  fits_i32({B, H, S, B * S, qkv.stride(0)}, "hi3d::self_attention");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
  const int64_t ld = qkv.stride(0);
  Tensor out = at::empty({B * S, C}, qkv.options());
  const char* base = (const char*)qkv.data_ptr();
  // V is read row-major from the fused projection (hi3d_attn_d64_v): no transpose pass, no V^T buffer
  check_rc(hi3d_attn_d64_v(base, base + C * 2, base + 4 * C, out.data_ptr(), (int)B, (int)H, (int)S, (int)S, (int)ld, (int)ld, (int)ld, (int)C,
                           (float)scale, stream_of(qkv)), "hi3d_attn_d64_v");
  return out;
}

// general form: q [B*S_q, >= H*64], k [B*S_kv, ...], v likewise (row strides free), e.g. cross-attention with many keys
Tensor attn_d64(const Tensor& q, const Tensor& k, const Tensor& v, int64_t B, int64_t H, int64_t S_q, int64_t S_kv, double scale) {
  want_bf16_rows(q, "hi3d::attn_d64 q"); want_bf16_rows(k, "hi3d::attn_d64 k"); want_bf16_rows(v, "hi3d::attn_d64 v");
  TORCH_CHECK(q.dim() == 2 && k.dim() == 2 && v.dim() == 2, "hi3d::attn_d64: 2-D token matrices required");
  TORCH_CHECK(q.size(0) == B * S_q && k.size(0) == B * S_kv && v.size(0) == B * S_kv, "hi3d::attn_d64: row counts do not match B, S_q, S_kv");
  TORCH_CHECK(q.size(1) >= H * 64 && k.size(1) >= H * 64 && v.size(1) >= H * 64, "hi3d::attn_d64: fewer than H*64 columns");
  TORCH_CHECK(q.device() == k.device() && q.device() == v.device(), "hi3d::attn_d64: operands on different devices");
  TORCH_CHECK(scale > 0.0, "hi3d::attn_d64: scale must be > 0");
  fits_i32({B, H, S_q, S_kv, B * S_q, B * S_kv, q.stride(0), k.stride(0), v.stride(0)}, "hi3d::attn_d64");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
  Tensor out = at::empty({B * S_q, H * 64}, q.options());
  check_rc(hi3d_attn_d64_v(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), (int)B, (int)H, (int)S_q, (int)S_kv, (int)q.stride(0),
                           (int)k.stride(0), (int)v.stride(0), (int)(H * 64), (float)scale, stream_of(q)), "hi3d_attn_d64_v");
  return out;
}

// attention over the frame axis at every pixel, frame-major tokens [(b t) s, 3*H*64] (video_attention.py:114-125)
Tensor attn_temporal(const Tensor& qkv, int64_t B, int64_t T, int64_t S, int64_t H, double scale) {
  want_bf16_rows(qkv, "hi3d::attn_temporal qkv");
  const int64_t C = H * 64;
  TORCH_CHECK(qkv.dim() == 2 && qkv.size(0) == B * T * S && qkv.size(1) == 3 * C, "hi3d::attn_temporal: qkv must be [B*T*S, 3*H*64]");
  fits_i32({B, T, S, H, B * T * S, qkv.stride(0)}, "hi3d::attn_temporal");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
  Tensor out = at::empty({B * T * S, C}, qkv.options());
  const char* base = (const char*)qkv.data_ptr();
  check_rc(hi3d_attn_temporal_d64(base, base + C * 2, base + 4 * C, out.data_ptr(), (int)B, (int)T, (int)S, (int)H, (int)qkv.stride(0), (int)C,
                                  (float)scale, stream_of(qkv)), "hi3d_attn_temporal_d64");
  return out;
}

// GroupNorm(32) (+ SiLU) over channels-last tokens x [inst*P, C]: GroupNorm32 + nn.SiLU (util.py:259-276, openaimodel.py:328-333)
Tensor groupnorm_silu(const Tensor& x, const Tensor& gamma, const Tensor& beta, int64_t inst, int64_t P, int64_t C, double eps, bool silu) {
  want_bf16_rows(x, "hi3d::groupnorm_silu x");
  TORCH_CHECK(x.is_contiguous() && x.numel() == inst * P * C, "hi3d::groupnorm_silu: x must be contiguous [inst*P, C]");
  want_f32(gamma, "hi3d::groupnorm_silu gamma", C); want_f32(beta, "hi3d::groupnorm_silu beta", C);
  same_device(x, {&gamma, &beta}, "hi3d::groupnorm_silu");
  fits_i32({inst, P, C}, "hi3d::groupnorm_silu");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor y = at::empty_like(x);
  Tensor ws = at::empty({hi3d_gn_workspace_floats((int)inst, (int)P, (int)C)}, x.options().dtype(at::kFloat));
  check_rc(hi3d_groupnorm_silu(x.data_ptr(), y.data_ptr(), gamma.data_ptr<float>(), beta.data_ptr<float>(), ws.data_ptr<float>(), (int)inst, (int)P,
                               (int)C, (float)eps, silu ? 1 : 0, stream_of(x)), "hi3d_groupnorm_silu");
  return y;
}

// nn.LayerNorm over the last dim (attention.py:520-522)
Tensor layernorm(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
  want_bf16_rows(x, "hi3d::layernorm x");
  TORCH_CHECK(x.is_contiguous(), "hi3d::layernorm: contiguous x required");
  const int64_t C = x.size(-1), R = x.numel() / C;
  want_f32(gamma, "hi3d::layernorm gamma", C); want_f32(beta, "hi3d::layernorm beta", C);
  same_device(x, {&gamma, &beta}, "hi3d::layernorm");
  fits_i32({R, C}, "hi3d::layernorm");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor y = at::empty_like(x);
  check_rc(hi3d_layernorm(x.data_ptr(), y.data_ptr(), nullptr, gamma.data_ptr<float>(), beta.data_ptr<float>(), nullptr, 1, (int)R, (int)C, (float)eps,
                          stream_of(x)), "hi3d_layernorm");
  return y;
}

// nn.Linear on tokens: out = x w^T + bias (+ residual); w [N, K] bf16 as nn.Linear stores it ([out, in], K-major)
Tensor linear(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, const c10::optional<Tensor>& residual) {
  want_bf16_rows(x, "hi3d::linear x"); want_bf16_rows(w, "hi3d::linear w");
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && w.is_contiguous() && x.size(1) == w.size(1), "hi3d::linear: x [M, K], w [N, K] required");
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  if (bias.has_value()) want_f32(*bias, "hi3d::linear bias", N);
  if (residual.has_value()) { want_bf16_rows(*residual, "hi3d::linear residual"); TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == N, "hi3d::linear: residual must be [M, N]"); }
  same_device(x, {&w, opt_t(bias), opt_t(residual)}, "hi3d::linear");
  fits_i32({M, N, K, x.stride(0), residual.has_value() ? residual->stride(0) : 0}, "hi3d::linear");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor out = at::empty({M, N}, x.options());
  hi3d_gemm_desc d = {};
  d.A = x.data_ptr(); d.W = w.data_ptr(); d.bias = (const float*)opt_ptr(bias); d.R1 = opt_ptr(residual); d.out = out.data_ptr();
  d.M = (int)M; d.N = (int)N; d.K = (int)K; d.lda = (int)x.stride(0); d.ldo = (int)N; d.ldr1 = residual.has_value() ? (int)residual->stride(0) : 0;
  d.rows_per_group = 1; d.amode = HI3D_A_DENSE; d.epi = HI3D_EPI_AFFINE;
  check_rc(hi3d_gemm_bf16(&d, stream_of(x)), "hi3d_gemm_bf16");
  return out;
}

// 3x3 convolution (padding 1) on channels-last tokens x [N*H*W, Cin] with weights packed [Cout][(ky, kx, Cin)]
// (hi3d_hip.pack.pack_conv3x3): nn.Conv2d of ResBlock / Downsample / Upsample (openaimodel.py:107-207, 328-354)
Tensor conv3x3(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, int64_t N, int64_t H, int64_t W, int64_t stride, bool up2x,
               const c10::optional<Tensor>& residual) {
  want_bf16_rows(x, "hi3d::conv3x3 x"); want_bf16_rows(w, "hi3d::conv3x3 w");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous() && x.size(0) == N * H * W, "hi3d::conv3x3: x must be contiguous [N*H*W, Cin]");
  const int64_t Cin = x.size(1), Cout = w.size(0);
  TORCH_CHECK(w.dim() == 2 && w.is_contiguous() && w.size(1) == 9 * Cin, "hi3d::conv3x3: w must be [Cout, 9*Cin] (pack_conv3x3 layout)");
  TORCH_CHECK((stride == 1 || stride == 2) && !(up2x && stride == 2), "hi3d::conv3x3: stride 1 or 2; nearest-2x only with stride 1");
  const int64_t Ho = up2x ? 2 * H : (stride == 2 ? (H + 1) / 2 : H), Wo = up2x ? 2 * W : (stride == 2 ? (W + 1) / 2 : W), M = N * Ho * Wo;
  if (bias.has_value()) want_f32(*bias, "hi3d::conv3x3 bias", Cout);
  if (residual.has_value()) { want_bf16_rows(*residual, "hi3d::conv3x3 residual"); TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == Cout, "hi3d::conv3x3: residual must be [M, Cout]"); }
  same_device(x, {&w, opt_t(bias), opt_t(residual)}, "hi3d::conv3x3");
  fits_i32({M, N * H * W, Cout, 9 * Cin, residual.has_value() ? residual->stride(0) : 0}, "hi3d::conv3x3");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor out = at::empty({M, Cout}, x.options());
  hi3d_gemm_desc d = {};
  d.A = x.data_ptr(); d.W = w.data_ptr(); d.bias = (const float*)opt_ptr(bias); d.R1 = opt_ptr(residual); d.out = out.data_ptr();
  d.M = (int)M; d.N = (int)Cout; d.K = (int)(9 * Cin); d.lda = (int)(9 * Cin); d.ldo = (int)Cout; d.ldr1 = residual.has_value() ? (int)residual->stride(0) : 0;
  d.rows_per_group = 1; d.amode = HI3D_A_CONV3X3; d.epi = HI3D_EPI_AFFINE;
  d.Hin = (int)H; d.Win = (int)W; d.Cin = (int)Cin; d.Hout = (int)Ho; d.Wout = (int)Wo; d.stride = (int)stride; d.up2x = up2x ? 1 : 0;
  check_rc(hi3d_gemm_bf16(&d, stream_of(x)), "hi3d_gemm_bf16");
  return out;
}

// FeedForward(glu=True) of a transformer block (attention.py:83-119): out = GEGLU(x w1^T + b1) w2^T + b2 (+ residual);
// w1 / b1 in the interleaved GEGLU row order of hi3d_hip.pack.pack_geglu.  The fused single-kernel form where the width
// has one (C = 320), otherwise the two GEMMs with the GEGLU epilogue.
Tensor ffn_geglu(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, const c10::optional<Tensor>& residual) {
  want_bf16_rows(x, "hi3d::ffn_geglu x"); want_bf16_rows(w1, "hi3d::ffn_geglu w1"); want_bf16_rows(w2, "hi3d::ffn_geglu w2");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous(), "hi3d::ffn_geglu: contiguous x [M, C] required");
  const int64_t M = x.size(0), C = x.size(1);
  TORCH_CHECK(w1.is_contiguous() && w1.size(0) == 8 * C && w1.size(1) == C && w2.is_contiguous() && w2.size(0) == C && w2.size(1) == 4 * C,
              "hi3d::ffn_geglu: w1 [8C, C], w2 [C, 4C] required");
  want_f32(b1, "hi3d::ffn_geglu b1", 8 * C); want_f32(b2, "hi3d::ffn_geglu b2", C);
  if (residual.has_value()) { want_bf16_rows(*residual, "hi3d::ffn_geglu residual"); TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == C, "hi3d::ffn_geglu: residual must be [M, C]"); }
  same_device(x, {&w1, &b1, &w2, &b2, opt_t(residual)}, "hi3d::ffn_geglu");
  fits_i32({M, 8 * C, residual.has_value() ? residual->stride(0) : 0}, "hi3d::ffn_geglu");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  Tensor out = at::empty({M, C}, x.options());
  const int ldr = residual.has_value() ? (int)residual->stride(0) : 0;
  int rc = hi3d_ffn_geglu(x.data_ptr(), w1.data_ptr(), b1.data_ptr<float>(), w2.data_ptr(), b2.data_ptr<float>(), opt_ptr(residual), nullptr, nullptr,
                          nullptr, out.data_ptr(), (int)M, (int)C, (int)C, (int)C, ldr, 0, 1, stream_of(x));
  if (rc == HI3D_ESHAPE) {            // no fused kernel at this width: the two GEMMs (hidden tensor in HBM)
    Tensor hid = at::empty({M, 4 * C}, x.options());
    hi3d_gemm_desc d = {};
    d.A = x.data_ptr(); d.W = w1.data_ptr(); d.bias = b1.data_ptr<float>(); d.out = hid.data_ptr();
    d.M = (int)M; d.N = (int)(8 * C); d.K = (int)C; d.lda = (int)C; d.ldo = (int)(4 * C); d.rows_per_group = 1; d.amode = HI3D_A_DENSE; d.epi = HI3D_EPI_GEGLU;
    check_rc(hi3d_gemm_bf16(&d, stream_of(x)), "hi3d_gemm_bf16 (GEGLU)");
    hi3d_gemm_desc e = {};
    e.A = hid.data_ptr(); e.W = w2.data_ptr(); e.bias = b2.data_ptr<float>(); e.R1 = opt_ptr(residual); e.out = out.data_ptr();
    e.M = (int)M; e.N = (int)C; e.K = (int)(4 * C); e.lda = (int)(4 * C); e.ldo = (int)C; e.ldr1 = ldr; e.rows_per_group = 1; e.amode = HI3D_A_DENSE; e.epi = HI3D_EPI_AFFINE;
    check_rc(hi3d_gemm_bf16(&e, stream_of(x)), "hi3d_gemm_bf16");
    return out;
  }
  check_rc(rc, "hi3d_ffn_geglu");
  return out;
}

}  // namespace

TORCH_LIBRARY(hi3d, m) {
  m.def("self_attention(Tensor qkv, int B, int S, int H, float scale) -> Tensor");
  m.def("attn_d64(Tensor q, Tensor k, Tensor v, int B, int H, int S_q, int S_kv, float scale) -> Tensor");
  m.def("attn_temporal(Tensor qkv, int B, int T, int S, int H, float scale) -> Tensor");
  m.def("groupnorm_silu(Tensor x, Tensor gamma, Tensor beta, int inst, int P, int C, float eps, bool silu) -> Tensor");
  m.def("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps) -> Tensor");
  m.def("linear(Tensor x, Tensor w, Tensor? bias, Tensor? residual) -> Tensor");
  m.def("conv3x3(Tensor x, Tensor w, Tensor? bias, int N, int H, int W, int stride, bool up2x, Tensor? residual) -> Tensor");
  m.def("ffn_geglu(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor? residual) -> Tensor");
}

// CompositeExplicitAutograd would also catch CPU tensors; registering for CUDA only makes a CPU call fail in the dispatcher
// ("no kernel for backend CPU") -- and the TORCH_CHECKs above say the same in words when reached.
TORCH_LIBRARY_IMPL(hi3d, CUDA, m) {
  m.impl("self_attention", &self_attention);
  m.impl("attn_d64", &attn_d64);
  m.impl("attn_temporal", &attn_temporal);
  m.impl("groupnorm_silu", &groupnorm_silu);
  m.impl("layernorm", &layernorm);
  m.impl("linear", &linear);
  m.impl("conv3x3", &conv3x3);
  m.impl("ffn_geglu", &ffn_geglu);
}

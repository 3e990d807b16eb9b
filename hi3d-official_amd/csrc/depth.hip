// The non-GEMM pieces of the stage-2 depth conditioner (vtdm/encoders.py:15-53: MiDaS DPT-hybrid, once per clip),
// gfx950.  Everything with a K dimension -- the BiT bottleneck convolutions, the ViT-B blocks, the reassemble / fusion /
// head convolutions -- runs on the GEMM, attention and norm kernels of gemm.hip / attention.hip / norm.hip
// (hi3d_hip/runtime_dpt.py).  Here: the 3-channel stem convolution, the stem's max pool and the stride-2 pixel pick of
// the 1x1 shortcut convolutions, bilinear resampling, the one-channel output convolution, and the depth map's
// min-max normalisation + 3x3 pixel-unshuffle.  All are bandwidth-trivial (a clip's 16 frames at 384 x 384):
// straightforward coalesced VALU kernels, channels-last like every other tensor of the runtime.
#include "common.h"

namespace {

// ---- (1) stem: 7x7 stride-2 convolution with TensorFlow 'SAME' padding, 3 -> 64 channels, fp32 in, bf16 out.
// timm StdConv2dSame(3, 64, 7, stride=2) of the BiT stem (annotator/midas/vit.py:499 builds it through timm); the
// weights arrive already standardised, laid out [ky][kx][ci][64].  One thread = one output pixel x 16 channels.
__global__ __launch_bounds__(256) void dpt_stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           unsigned short* __restrict__ y, int H, int W, int Ho, int Wo,
                                                           int pt, int pl, long npix) {
  __shared__ float sw[147 * 64];
  for (int i = threadIdx.x; i < 147 * 64; i += 256) sw[i] = w[i];
  __syncthreads();
  const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pix >= npix) return;
  const int c0 = (threadIdx.x & 3) * 16;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
  const long n = pix / ((long)Wo * Ho);
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    const int iy = oy * 2 + ky - pt;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 7; ++kx) {
      const int ix = ox * 2 + kx - pl;
      if (ix < 0 || ix >= W) continue;
      const float* xp = x + ((n * H + iy) * W + ix) * 3;
      const float* wp = sw + (ky * 7 + kx) * 3 * 64 + c0;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float xv = xp[ci];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += xv * wp[ci * 64 + j];
      }
    }
  }
  uint4* yp = (uint4*)(y + pix * 64 + c0);
  yp[0] = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
  yp[1] = make_uint4(pack_bf16x2(acc[8], acc[9]), pack_bf16x2(acc[10], acc[11]), pack_bf16x2(acc[12], acc[13]), pack_bf16x2(acc[14], acc[15]));
}

// ---- (2) stride-2 window over channels-last bf16, 8 channels per thread.  MODE 0: 3x3 max, 'SAME' padding with
// -inf (timm MaxPool2dSame of the stem); MODE 1: the top-left pixel only (a 1x1 stride-2 convolution's gather:
// the BiT downsample shortcuts, 'SAME' padding is empty for a 1x1 window).
template <int MODE>
__global__ __launch_bounds__(256) void dpt_pool2_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int H, int W,
                                                       int Ho, int Wo, int C8, int pt, int pl, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const long pix = i / C8;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
  const long n = pix / ((long)Wo * Ho);
  if (MODE == 1) {
    y[i] = x[((n * H + oy * 2) * W + ox * 2) * C8 + c];
    return;
  }
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 + ky - pt;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 + kx - pl;
      if (ix < 0 || ix >= W) continue;
      const uint4 v = x[((n * H + iy) * W + ix) * C8 + c];
      const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        m[2 * j] = fmaxf(m[2 * j], bf16_to_f32(u[j] & 0xffff));
        m[2 * j + 1] = fmaxf(m[2 * j + 1], bf16_to_f32(u[j] >> 16));
      }
    }
  }
  y[i] = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
}

// ---- (3) bilinear resampling of channels-last images, torch.nn.functional.interpolate(mode='bilinear') semantics:
// align_corners: src = dst * (in - 1) / (out - 1); otherwise src = max((dst + 0.5) * in / out - 0.5, 0); the upper
// neighbour is clamped to the last pixel.  (DepthEmbedder's two resizes, vtdm/encoders.py:41,46, align_corners False;
// the x2 of the fusion blocks and of the head, annotator/midas/blocks.py:383-386, dpt_depth.py:96, align_corners True.)
struct ResizeGeom { int Hi, Wi, Ho, Wo; float sh, sw; int align; };

__device__ __forceinline__ void resize_src(int d, float scale, int align, int in, int& i0, int& i1, float& l1) {
  float s = align ? d * scale : fmaxf((d + 0.5f) * scale - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = s - i0;
}

__global__ __launch_bounds__(256) void resize_bilinear_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y,
                                                                  ResizeGeom g, int C8, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const long pix = i / C8;
  const int ox = (int)(pix % g.Wo), oy = (int)((pix / g.Wo) % g.Ho);
  const long n = pix / ((long)g.Wo * g.Ho);
  int y0, y1, x0, x1; float ly, lx;
  resize_src(oy, g.sh, g.align, g.Hi, y0, y1, ly);
  resize_src(ox, g.sw, g.align, g.Wi, x0, x1, lx);
  const uint4* b = x + n * g.Hi * g.Wi * C8 + c;
  const uint4 v00 = b[((long)y0 * g.Wi + x0) * C8], v01 = b[((long)y0 * g.Wi + x1) * C8];
  const uint4 v10 = b[((long)y1 * g.Wi + x0) * C8], v11 = b[((long)y1 * g.Wi + x1) * C8];
  const unsigned int a[4] = {v00.x, v00.y, v00.z, v00.w}, bb[4] = {v01.x, v01.y, v01.z, v01.w};
  const unsigned int cc[4] = {v10.x, v10.y, v10.z, v10.w}, d[4] = {v11.x, v11.y, v11.z, v11.w};
  unsigned int o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sft = h * 16;
      const float p00 = bf16_to_f32((a[j] >> sft) & 0xffff), p01 = bf16_to_f32((bb[j] >> sft) & 0xffff);
      const float p10 = bf16_to_f32((cc[j] >> sft) & 0xffff), p11 = bf16_to_f32((d[j] >> sft) & 0xffff);
      r[h] = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
    }
    o[j] = pack_bf16x2(r[0], r[1]);
  }
  y[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void resize_bilinear_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 ResizeGeom g, int C, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long pix = i / C;
  const int ox = (int)(pix % g.Wo), oy = (int)((pix / g.Wo) % g.Ho);
  const long n = pix / ((long)g.Wo * g.Ho);
  int y0, y1, x0, x1; float ly, lx;
  resize_src(oy, g.sh, g.align, g.Hi, y0, y1, ly);
  resize_src(ox, g.sw, g.align, g.Wi, x0, x1, lx);
  const float* b = x + n * g.Hi * g.Wi * C + c;
  const float p00 = b[((long)y0 * g.Wi + x0) * C], p01 = b[((long)y0 * g.Wi + x1) * C];
  const float p10 = b[((long)y1 * g.Wi + x0) * C], p11 = b[((long)y1 * g.Wi + x1) * C];
  y[i] = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
}

// ---- (4) the head's last two layers after its 128 -> 32 convolution: ReLU, 1x1 convolution to one channel, ReLU
// (annotator/midas/dpt_depth.py:97-100): y[m] = relu(b + sum_c w[c] * relu(x[m][c])), x bf16 [M][C], y fp32 [M]
__global__ __launch_bounds__(256) void dpt_head_out_kernel(const uint4* __restrict__ x, const float* __restrict__ w, float b,
                                                          float* __restrict__ y, int C8, long M) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float acc = b;
  for (int c = 0; c < C8; ++c) {
    const uint4 v = x[m * C8 + c];
    const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc += w[c * 8 + 2 * j] * fmaxf(bf16_to_f32(u[j] & 0xffff), 0.f);
      acc += w[c * 8 + 2 * j + 1] * fmaxf(bf16_to_f32(u[j] >> 16), 0.f);
    }
  }
  y[m] = fmaxf(acc, 0.f);
}

// ---- (5) per image: y = (d - min d) / max(max(d - min d), 1e-6), then 'b 1 (h h0) (w w0) -> b (h0 w0) h w' with
// h0 = w0 = s (vtdm/encoders.py:47-50).  One block per image; d fp32 [B][Hs][Ws] -> out fp32 [B][s*s][Hs/s][Ws/s]
__global__ __launch_bounds__(1024) void depth_normalize_unshuffle_kernel(const float* __restrict__ d, float* __restrict__ out,
                                                                        int Hs, int Ws, int s) {
  __shared__ float smin[16], smax[16];
  const long n = (long)Hs * Ws;
  const float* img = d + blockIdx.x * n;
  float mn = INFINITY, mx = -INFINITY;
  for (long i = threadIdx.x; i < n; i += 1024) { const float v = img[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  mn = -wave_max(-mn); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = smin[0]; mx = smax[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) { mn = fminf(mn, smin[i]); mx = fmaxf(mx, smax[i]); }
  const float den = fmaxf(mx - mn, 1e-6f);
  const int Hq = Hs / s, Wq = Ws / s;
  float* o = out + blockIdx.x * n;
  for (long i = threadIdx.x; i < n; i += 1024) {          // i walks the OUTPUT (coalesced stores)
    const int wq = (int)(i % Wq), hq = (int)((i / Wq) % Hq), ch = (int)(i / ((long)Wq * Hq));
    const int h0 = ch / s, w0 = ch % s;
    o[i] = (img[(long)(hq * s + h0) * Ws + wq * s + w0] - mn) / den;
  }
}

inline void same_pad(int in, int k, int& out, int& before) {
  out = (in + 1) / 2;
  const int total = (out - 1) * 2 + k - in;
  before = total > 0 ? total / 2 : 0;
}


// ------------------------------------------------------------------------------------
// One axis of a separable image resampling with precomputed banded taps (hi3d_hip/resample.py):
//   out[outer][o][inner] = a[c] * sum_t w[o][t] * in[outer][start[o] + t][inner] + b[c],   c = (outer / chan_div) % chan_mod
// Replaces kornia.geometry.resize (bicubic, align_corners, antialias: Gaussian blur folded into the taps) in
// FrozenOpenCLIPImageEmbedder.preprocess (sgm/modules/encoders/modules.py:619-628) and F.interpolate(bilinear) in
// AesEmbedder.forward (vtdm/encoders.py:80-83), together with their (x + 1) / 2 and mean / std normalisation (the
// per-channel affine of the second pass).  Thread = one output element, `inner` fastest (coalesced for the H pass;
// the W pass, inner == 1, reads ~ntap contiguous floats per thread).
__global__ __launch_bounds__(256) void resample_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const int* __restrict__ start, const float* __restrict__ w,
                                                            int ntap, long total, int n_in, int n_out, int inner,
                                                            const float* __restrict__ a, const float* __restrict__ b,
                                                            int chan_div, int chan_mod) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)(idx % inner);
  const long r = idx / inner;
  const int o = (int)(r % n_out);
  const long ou = r / n_out;
  const float* src = in + (ou * n_in + start[o]) * inner + i;
  const float* wo = w + (long)o * ntap;
  const int lim = n_in - start[o];                 // taps beyond the input carry zero weight; do not touch memory there
  float acc = 0.f;
  for (int t = 0; t < ntap; ++t)
    if (t < lim) acc += wo[t] * src[(long)t * inner];
  if (a) {
    const int c = (int)((ou / chan_div) % chan_mod);
    acc = acc * a[c] + b[c];
  }
  out[idx] = acc;
}

}  // namespace

extern "C" int hi3d_dpt_stem_conv(const float* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, void* stream) {
  if (!x || !w || !y) HI3D_FAIL(HI3D_EINVAL, "dpt_stem_conv: null pointer");
  if (N <= 0 || H <= 0 || W <= 0) HI3D_FAIL(HI3D_EINVAL, "dpt_stem_conv: non-positive size");
  if ((uintptr_t)y & 15) HI3D_FAIL(HI3D_EALIGN, "dpt_stem_conv: y not 16-byte aligned");
  int Ho, Wo, pt, pl;
  same_pad(H, 7, Ho, pt); same_pad(W, 7, Wo, pl);
  const long npix = (long)N * Ho * Wo;
  hipLaunchKernelGGL(dpt_stem_conv_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     x, w, (unsigned short*)y, H, W, Ho, Wo, pt, pl, npix);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_pool2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t mode, void* stream) {
  if (!x || !y) HI3D_FAIL(HI3D_EINVAL, "pool2: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "pool2: non-positive size");
  if (C % 8) HI3D_FAIL(HI3D_ESHAPE, "pool2: C must be a multiple of 8");
  if (mode != 0 && mode != 1) HI3D_FAIL(HI3D_EINVAL, "pool2: mode must be 0 (3x3 max, SAME) or 1 (pixel pick)");
  if (((uintptr_t)x | (uintptr_t)y) & 15) HI3D_FAIL(HI3D_EALIGN, "pool2: x / y not 16-byte aligned");
  int Ho, Wo, pt, pl;
  same_pad(H, mode == 0 ? 3 : 1, Ho, pt); same_pad(W, mode == 0 ? 3 : 1, Wo, pl);
  const long total = (long)N * Ho * Wo * (C / 8);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (mode == 0)
    hipLaunchKernelGGL(dpt_pool2_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, H, W, Ho, Wo, C / 8, pt, pl, total);
  else
    hipLaunchKernelGGL(dpt_pool2_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, H, W, Ho, Wo, C / 8, pt, pl, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_resize_bilinear_nhwc(const void* x, void* y, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                         int32_t C, int32_t align_corners, int32_t is_f32, void* stream) {
  if (!x || !y) HI3D_FAIL(HI3D_EINVAL, "resize_bilinear: null pointer");
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "resize_bilinear: non-positive size");
  ResizeGeom g;
  g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.align = align_corners ? 1 : 0;
  g.sh = align_corners ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f) : (float)Hi / (float)Ho;
  g.sw = align_corners ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f) : (float)Wi / (float)Wo;
  if (is_f32) {
    const long total = (long)N * Ho * Wo * C;
    hipLaunchKernelGGL(resize_bilinear_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (float*)y, g, C, total);
  } else {
    if (C % 8) HI3D_FAIL(HI3D_ESHAPE, "resize_bilinear: bf16 needs C % 8 == 0");
    if (((uintptr_t)x | (uintptr_t)y) & 15) HI3D_FAIL(HI3D_EALIGN, "resize_bilinear: x / y not 16-byte aligned");
    const long total = (long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(resize_bilinear_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, (uint4*)y, g, C / 8, total);
  }
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_dpt_head_out(const void* x, const float* w, float b, float* y, int64_t M, int32_t C, void* stream) {
  if (!x || !w || !y) HI3D_FAIL(HI3D_EINVAL, "dpt_head_out: null pointer");
  if (M <= 0 || C <= 0 || (C % 8)) HI3D_FAIL(HI3D_ESHAPE, "dpt_head_out: M > 0 and C a positive multiple of 8");
  if ((uintptr_t)x & 15) HI3D_FAIL(HI3D_EALIGN, "dpt_head_out: x not 16-byte aligned");
  hipLaunchKernelGGL(dpt_head_out_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, w, b, y, C / 8, (long)M);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_depth_normalize_unshuffle(const float* d, float* out, int32_t B, int32_t Hs, int32_t Ws, int32_t s, void* stream) {
  if (!d || !out) HI3D_FAIL(HI3D_EINVAL, "depth_normalize_unshuffle: null pointer");
  if (B <= 0 || Hs <= 0 || Ws <= 0 || s <= 0) HI3D_FAIL(HI3D_EINVAL, "depth_normalize_unshuffle: non-positive size");
  if (Hs % s || Ws % s) HI3D_FAIL(HI3D_ESHAPE, "depth_normalize_unshuffle: size not a multiple of the shuffle size");
  hipLaunchKernelGGL(depth_normalize_unshuffle_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, d, out, Hs, Ws, s);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_resample_axis(const float* in, float* out, const int32_t* start, const float* w, int32_t ntap,
                                  int64_t outer, int32_t n_in, int32_t n_out, int32_t inner, const float* scale,
                                  const float* shift, int32_t chan_div, int32_t chan_mod, void* stream) {
  if (!in || !out || !start || !w) HI3D_FAIL(HI3D_EINVAL, "resample_axis: null pointer");
  if (ntap <= 0 || outer <= 0 || n_in <= 0 || n_out <= 0 || inner <= 0) HI3D_FAIL(HI3D_EINVAL, "resample_axis: non-positive size");
  if ((scale == nullptr) != (shift == nullptr)) HI3D_FAIL(HI3D_EINVAL, "resample_axis: scale and shift come together");
  if (scale && (chan_div <= 0 || chan_mod <= 0)) HI3D_FAIL(HI3D_EINVAL, "resample_axis: bad channel decomposition");
  const long total = (long)outer * n_out * inner;
  if ((total + 255) / 256 > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, "resample_axis: grid too large");
  hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     (const int*)start, w, ntap, total, n_in, n_out, inner, scale, shift, chan_div, chan_mod);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

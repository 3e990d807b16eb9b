// Small layout / elementwise kernels of the sampler step (gfx950).  All are
// bandwidth-trivial next to the UNet; what matters is that each replaces a chain of
// tiny launches in the reference (guiders.py:88-99, denoiser.py:36-39, wrappers.py:26,
// sampling.py:93-107) with a single pass.
#include "common.h"

namespace {

__global__ void concat_channels_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                       uint4* __restrict__ out, long rows, int v0, int v1) {
  const int vt = v0 + v1;
  const long total = rows * vt;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vt; const int c = (int)(i - r * vt);
    out[i] = c < v0 ? a[r * v0 + c] : b[r * v1 + (c - v0)];
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, void* __restrict__ out, int n, int dim,
                                          float neg_log_period_over_half, int out_bf16) {
  const int half = dim / 2;
  const int i = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
  if (i >= n || j >= dim) return;
  float val = 0.f;
  if (j < 2 * half) {
    const int jj = j < half ? j : j - half;
    const float arg = t[i] * expf(neg_log_period_over_half * (float)jj);
    val = j < half ? cosf(arg) : sinf(arg);
  }
  if (out_bf16) ((unsigned short*)out)[(long)i * dim + j] = f32_to_bf16(val);
  else ((float*)out)[(long)i * dim + j] = val;
}

__global__ void silu_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f32_to_bf16(silu_f(x[i]));
}

// Elementwise activation on bf16, 8 elements per thread: out = act(x [+ y]); out may alias x.
// kind 0 = exact-erf GELU (nn.GELU: OpenCLIP ViT-H/14, the ViT-B of MiDaS), 1 = QuickGELU x * sigmoid(1.702 x) (OpenAI
// ViT-L/14), 2 = ReLU (BiT bottlenecks and the fusion blocks of MiDaS DPT-hybrid), 3 = identity (a plain add)
__global__ void add_act_bf16_kernel(const uint4* x, const uint4* __restrict__ y, uint4* out, long n8, int kind) {   // (out may be x)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const uint4 v = x[i];
    unsigned int u[4] = {v.x, v.y, v.z, v.w};
    unsigned int q[4] = {0u, 0u, 0u, 0u};
    if (y) { const uint4 t = y[i]; q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bf16_to_f32(u[j] & 0xffff) + bf16_to_f32(q[j] & 0xffff), b = bf16_to_f32(u[j] >> 16) + bf16_to_f32(q[j] >> 16);
      if (kind == 0) { a = gelu_erf_f(a); b = gelu_erf_f(b); }
      else if (kind == 1) {
        a = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * a));
        b = b * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * b));
      } else if (kind == 2) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
      u[j] = pack_bf16x2(a, b);
    }
    out[i] = make_uint4(u[0], u[1], u[2], u[3]);
  }
}

// x[r][:] /= ||x[r]||_2 (a zero row stays zero), fp32, one wave per row: tools/aes_score.py `normalized`
__global__ __launch_bounds__(64) void l2_normalize_rows_kernel(float* __restrict__ x, int C) {
  float* row = x + (long)blockIdx.x * C;
  float ss = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) ss += row[c] * row[c];
  ss = wave_sum(ss);
  const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 1.0f;
  for (int c = threadIdx.x; c < C; c += 64) row[c] *= inv;
}

// one thread per (u, t, pixel): gathers 4 + Cc channel planes (NCHW fp32) into a
// padded channels-last bf16 row
__global__ void cfg_prepare_kernel(const float* __restrict__ x, const float* __restrict__ cu,
                                   const float* __restrict__ cc, unsigned short* __restrict__ out,
                                   int T, int HW, int Cc, int Cp, float c_in) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = 2L * T * HW;
  if (idx >= total) return;
  const int p = (int)(idx % HW); const int t = (int)((idx / HW) % T); const int u = (int)(idx / ((long)HW * T));
  unsigned short* o = out + idx * Cp;
  const float* xs = x + (long)t * 4 * HW + p;
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = f32_to_bf16(xs[(long)c * HW] * c_in);
  const float* cs = u ? cc : cu;
  for (int c = 0; c < Cc; ++c) o[4 + c] = cs ? f32_to_bf16(cs[((long)t * Cc + c) * HW + p]) : (unsigned short)0;
  for (int c = 4 + Cc; c < Cp; ++c) o[c] = 0;
}

__global__ void sampler_step_kernel(float* __restrict__ x, const float* __restrict__ net,
                                    const float* __restrict__ scale, int T, int HW, int ldn,
                                    float c_skip, float c_out, float dt_over_sigma) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * HW;
  if (idx >= total) return;
  const int p = (int)(idx % HW), t = (int)(idx / HW);
  const float* nu = net + idx * ldn;
  const float* nc = net + (total + idx) * ldn;
  const float s = scale[t];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float* xp = x + ((long)t * 4 + c) * HW + p;
    const float xv = *xp;
    const float du = nu[c] * c_out + xv * c_skip;
    const float dc = nc[c] * c_out + xv * c_skip;
    const float d = du + s * (dc - du);
    *xp = xv + dt_over_sigma * (xv - d);
  }
}

// Per-step half of cfg_prepare for a graph-replayed sampler: only the 4 latent channels of both
// CFG halves are rewritten (the conditioning channels of the token buffer are constant for a clip
// and stay where hi3d_cfg_prepare put them); sigma is read from DEVICE memory so that one captured
// launch serves every step.  Also emits c_noise = ln(sigma)/4 for the 2T rows of the batch.
__global__ void cfg_update_x_kernel(const float* __restrict__ x, unsigned short* __restrict__ tok,
                                    const float* __restrict__ sig, float* __restrict__ tvec,
                                    int T, int HW, int Cp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * HW;
  const float s = sig[0];
  if (idx < 2 * T && tvec) tvec[idx] = 0.25f * logf(s);
  if (idx >= total) return;
  const float c_in = 1.0f / sqrtf(s * s + 1.0f);
  const int p = (int)(idx % HW), t = (int)(idx / HW);
  const float* xs = x + (long)t * 4 * HW + p;
  const uint2 v = make_uint2(pack_bf16x2(xs[0] * c_in, xs[(long)HW] * c_in),
                             pack_bf16x2(xs[2L * HW] * c_in, xs[3L * HW] * c_in));
  *(uint2*)(tok + idx * Cp) = v;
  *(uint2*)(tok + (total + idx) * Cp) = v;
}

// hi3d_sampler_step with sigma / sigma_next in device memory and an explicit output (may alias x)
// (x and xo carry no __restrict__: FusedStepper passes the same buffer for both)
__global__ void sampler_step_dev_kernel(const float* x, float* xo,
                                        const float* __restrict__ net, const float* __restrict__ scale,
                                        const float* __restrict__ sig, int T, int HW, int ldn) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * HW;
  if (idx >= total) return;
  const float sg = sig[0], sn = sig[1];
  const float c_skip = 1.0f / (sg * sg + 1.0f);
  const float c_out = -sg / sqrtf(sg * sg + 1.0f);
  const float dt_over_sigma = (sn - sg) / sg;
  const int p = (int)(idx % HW), t = (int)(idx / HW);
  const float* nu = net + idx * ldn;
  const float* nc = net + (total + idx) * ldn;
  const float s = scale[t];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const long o = ((long)t * 4 + c) * HW + p;
    const float xv = x[o];
    const float du = nu[c] * c_out + xv * c_skip;
    const float dc = nc[c] * c_out + xv * c_skip;
    const float d = du + s * (dc - du);
    xo[o] = xv + dt_over_sigma * (xv - d);
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                    int C, int HW, int Cpad, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over N*HW*Cpad
  if (idx >= total) return;
  const int c = (int)(idx % Cpad); const long np = idx / Cpad;
  const int p = (int)(np % HW); const long n = np / HW;
  y[idx] = c < C ? f32_to_bf16(x[(n * C + c) * HW + p]) : (unsigned short)0;
}

__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y, int C, int HW,
                                    int ldx, int x_is_f32, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over N*C*HW (output order)
  if (idx >= total) return;
  const int p = (int)(idx % HW); const long nc = idx / HW;
  const int c = (int)(nc % C); const long n = nc / C;
  const long src = (n * HW + p) * ldx + c;
  y[idx] = x_is_f32 ? ((const float*)x)[src] : bf16_to_f32(((const unsigned short*)x)[src]);
}

__global__ void vae_latent_prepare_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                          const float* __restrict__ b, unsigned short* __restrict__ out,
                                          int Cz, int HW, int Cpad, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;    // over N*HW pixels
  if (idx >= total) return;
  const int p = (int)(idx % HW); const long n = idx / HW;
  float zi[8];
  for (int c = 0; c < Cz; ++c) zi[c] = z[(n * Cz + c) * HW + p];
  unsigned short* o = out + idx * Cpad;
  for (int co = 0; co < Cz; ++co) {
    float acc = b[co];
    for (int ci = 0; ci < Cz; ++ci) acc += w[co * Cz + ci] * zi[ci];
    o[co] = f32_to_bf16(acc);
  }
  for (int c = Cz; c < Cpad; ++c) o[c] = 0;
}

// one 256-thread block per row, the row (<= 16384 fp32) lives in registers
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, unsigned short* __restrict__ p,
                                                          int N, int lds_, int ldp, float scale_log2) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* row = s + (long)blockIdx.x * lds_;
  unsigned short* prow = p + (long)blockIdx.x * ldp;
  float v[64];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = tid + i * 256;
    v[i] = c < N ? row[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    v[i] = __builtin_amdgcn_exp2f((v[i] - mx) * scale_log2);     // exp2(-inf) = 0 for the padding
    sum += v[i];
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = tid + i * 256;
    if (c < ldp) prow[c] = f32_to_bf16(v[i] * inv);
  }
}

__global__ void vae_posterior_kernel(const float* __restrict__ mom, const float* __restrict__ wq,
                                     const float* __restrict__ bq, const float* __restrict__ noise,
                                     float* __restrict__ z, int Cz, int HW, int ldm, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;    // over N*HW pixels
  if (idx >= total) return;
  const int p = (int)(idx % HW); const long n = idx / HW;
  const int C2 = 2 * Cz;
  float mi[16], mo[16];
  for (int c = 0; c < C2; ++c) mi[c] = mom[idx * ldm + c];
  for (int co = 0; co < C2; ++co) {
    float acc = bq[co];
    for (int ci = 0; ci < C2; ++ci) acc += wq[co * C2 + ci] * mi[ci];
    mo[co] = acc;
  }
  for (int c = 0; c < Cz; ++c) {
    float v = mo[c];
    if (noise) {
      const float lv = fminf(fmaxf(mo[Cz + c], -30.0f), 20.0f);
      v += expf(0.5f * lv) * noise[(n * Cz + c) * HW + p];
    }
    z[(n * Cz + c) * HW + p] = v;
  }
}

__global__ void v02_blend_kernel(float* __restrict__ lat, const float* __restrict__ noise,
                                 const float* __restrict__ z, long n, float alpha, float sigma) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    lat[i] = lat[i] * (1.0f - alpha) + (noise[i] * sigma + z[i]) * alpha;
}

__global__ void time_mix_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                      const float* __restrict__ b, float* __restrict__ out, int T, int HW,
                                      int C, int ldx, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;    // over (b t) * HW
  if (idx >= total) return;
  const int p = (int)(idx % HW); const long f = idx / HW; const int t = (int)(f % T);
  float acc[4];
  for (int co = 0; co < C; ++co) acc[co] = b[co];
  for (int kt = 0; kt < 3; ++kt) {
    const int tt = t + kt - 1;
    if (tt < 0 || tt >= T) continue;
    const float* xp = x + (idx + (long)(kt - 1) * HW) * ldx;
    for (int ci = 0; ci < C; ++ci) {
      const float v = xp[ci];
      for (int co = 0; co < C; ++co) acc[co] += w[(co * C + ci) * 3 + kt] * v;
    }
  }
  for (int co = 0; co < C; ++co) out[(f * C + co) * HW + p] = acc[co];
}

// ... with an isotropic 3 x 3 x 3 kernel (VideoDecoder(video_kernel_size=3), the reference class's default): w [co][ci][kt][ky][kx]
__global__ void time_mix_small_k3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ b, float* __restrict__ out, int T, int H, int W,
                                         int C, int ldx, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;    // over (b t) * H * W
  if (idx >= total) return;
  const int HW = H * W;
  const int p = (int)(idx % HW); const long f = idx / HW; const int t = (int)(f % T);
  const int y = p / W, xx = p - y * W;
  float acc[4];
  for (int co = 0; co < C; ++co) acc[co] = b[co];
  for (int kt = 0; kt < 3; ++kt) {
    const int tt = t + kt - 1;
    if (tt < 0 || tt >= T) continue;
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int xc = xx + kx - 1;
        if (xc < 0 || xc >= W) continue;
        const float* xp = x + ((f + (kt - 1)) * HW + (long)yy * W + xc) * ldx;
        for (int ci = 0; ci < C; ++ci) {
          const float v = xp[ci];
          for (int co = 0; co < C; ++co) acc[co] += w[(((co * C + ci) * 3 + kt) * 3 + ky) * 3 + kx] * v;
        }
      }
    }
  }
  for (int co = 0; co < C; ++co) out[(f * C + co) * HW + p] = acc[co];
}

inline unsigned grid_for(long n, int block, long cap = 65536) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}


// Row permutation of a 4-D array of rows: in[d0][d1][d2][d3][row] -> out ordered by (p0, p1, p2, p3), rows of `vec` 16-byte
// vectors.  The pack / unpack around the frame <-> space all-to-all of a clip spread over several GPUs (SURVEY 8e;
// hi3d_hip/parallel.py: FrameSpaceGroup) -- "(b tl (dst sl)) c -> dst (b tl sl) c" and back -- which the reference never
// needs (it has no inference parallelism) and round 2 did with ATen permute().contiguous().
__global__ __launch_bounds__(256) void permute_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long total, int vec,
                                                           int e0, int e1, int e2, int e3, long s0, long s1, long s2, long s3) {
  // e* : extents of the OUTPUT dims; s* : stride (in rows) in the INPUT of each output dim
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int v = (int)(idx % vec);
  long r = idx / vec;
  const int i3 = (int)(r % e3); r /= e3;
  const int i2 = (int)(r % e2); r /= e2;
  const int i1 = (int)(r % e1); r /= e1;
  const int i0 = (int)r;
  out[idx] = in[(i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3) * vec + v];
}

}  // namespace

extern "C" int hi3d_concat_channels(const void* a, const void* b, void* out, int64_t rows,
                                    int32_t C0, int32_t C1, void* stream) {
  if (!a || !b || !out) HI3D_FAIL(HI3D_EINVAL, "concat: null pointer");
  if (rows <= 0 || C0 <= 0 || C1 <= 0) HI3D_FAIL(HI3D_EINVAL, "concat: non-positive size");
  if ((C0 % 8) || (C1 % 8)) HI3D_FAIL(HI3D_ESHAPE, "concat: channel counts must be multiples of 8");
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) HI3D_FAIL(HI3D_EALIGN, "concat: misaligned pointer");
  const long total = rows * ((C0 + C1) / 8);
  hipLaunchKernelGGL(concat_channels_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)a, (const uint4*)b, (uint4*)out, (long)rows, C0 / 8, C1 / 8);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim,
                                       float max_period, int32_t out_bf16, void* stream) {
  if (!t || !out) HI3D_FAIL(HI3D_EINVAL, "timestep_embedding: null pointer");
  if (n <= 0 || dim < 2 || max_period <= 0.f) HI3D_FAIL(HI3D_EINVAL, "timestep_embedding: bad size");
  const int half = dim / 2;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(n, (dim + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     t, out, n, dim, -logf(max_period) / (float)half, out_bf16);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_silu_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (!x || !y) HI3D_FAIL(HI3D_EINVAL, "silu: null pointer");
  if (n <= 0) HI3D_FAIL(HI3D_EINVAL, "silu: non-positive size");
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, (long)n);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_add_act_bf16(const void* x, const void* y, void* out, int64_t n, int32_t kind, void* stream) {
  if (!x || !out) HI3D_FAIL(HI3D_EINVAL, "add_act: null pointer");
  if (n <= 0 || (n % 8)) HI3D_FAIL(HI3D_ESHAPE, "add_act: n must be a positive multiple of 8");
  if (kind < 0 || kind > 3) HI3D_FAIL(HI3D_EINVAL, "add_act: kind must be 0 (gelu), 1 (quick_gelu), 2 (relu) or 3 (identity)");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) HI3D_FAIL(HI3D_EALIGN, "add_act: pointer not 16-byte aligned");
  hipLaunchKernelGGL(add_act_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (const uint4*)y, (uint4*)out, (long)(n / 8), kind);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_act_bf16(void* x, int64_t n, int32_t kind, void* stream) {
  if (!x) HI3D_FAIL(HI3D_EINVAL, "act: null pointer");
  if (kind < 0 || kind > 2) HI3D_FAIL(HI3D_EINVAL, "act: kind must be 0 (gelu), 1 (quick_gelu) or 2 (relu)");
  return hi3d_add_act_bf16(x, nullptr, x, n, kind, stream);
}

extern "C" int hi3d_l2_normalize_rows(float* x, int32_t R, int32_t C, void* stream) {
  if (!x) HI3D_FAIL(HI3D_EINVAL, "l2_normalize: null pointer");
  if (R <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "l2_normalize: non-positive size");
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, x, C);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_cfg_prepare(const float* x, const float* concat_uc, const float* concat_c, void* out,
                                int32_t T, int32_t HW, int32_t Cc, int32_t Cp, float sigma, void* stream) {
  if (!x || !out) HI3D_FAIL(HI3D_EINVAL, "cfg_prepare: null pointer");
  if (T <= 0 || HW <= 0 || Cc < 0 || Cp < 4 + Cc) HI3D_FAIL(HI3D_EINVAL, "cfg_prepare: bad size");
  if (sigma < 0.f) HI3D_FAIL(HI3D_EINVAL, "cfg_prepare: negative sigma");
  const float c_in = 1.0f / sqrtf(sigma * sigma + 1.0f);
  const long total = 2L * T * HW;
  hipLaunchKernelGGL(cfg_prepare_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, concat_uc, concat_c, (unsigned short*)out, T, HW, Cc, Cp, c_in);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_sampler_step(float* x, const float* net, const float* scale, int32_t T, int32_t HW,
                                 int32_t ldn, float sigma, float sigma_next, void* stream) {
  if (!x || !net || !scale) HI3D_FAIL(HI3D_EINVAL, "sampler_step: null pointer");
  if (T <= 0 || HW <= 0 || ldn < 4) HI3D_FAIL(HI3D_EINVAL, "sampler_step: bad size");
  if (!(sigma > 0.f)) HI3D_FAIL(HI3D_EINVAL, "sampler_step: sigma must be > 0");
  const float c_skip = 1.0f / (sigma * sigma + 1.0f);
  const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
  const float dt_over_sigma = (sigma_next - sigma) / sigma;
  const long total = (long)T * HW;
  hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, net, scale, T, HW, ldn, c_skip, c_out, dt_over_sigma);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_cfg_update_x(const float* x, void* tokens, const float* sigma_dev, float* c_noise_out,
                                 int32_t T, int32_t HW, int32_t Cp, void* stream) {
  if (!x || !tokens || !sigma_dev) HI3D_FAIL(HI3D_EINVAL, "cfg_update_x: null pointer");
  if (T <= 0 || HW <= 0 || Cp < 4 || Cp % 4) HI3D_FAIL(HI3D_EINVAL, "cfg_update_x: bad size");
  if ((uintptr_t)tokens & 7) HI3D_FAIL(HI3D_EALIGN, "cfg_update_x: tokens not 8-byte aligned");
  const long total = (long)T * HW;
  hipLaunchKernelGGL(cfg_update_x_kernel, dim3(grid_for(total > 2 * T ? total : 2 * T, 256, 1L << 30)), dim3(256), 0,
                     (hipStream_t)stream, x, (unsigned short*)tokens, sigma_dev, c_noise_out, T, HW, Cp);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_sampler_step_dev(const float* x, float* x_out, const float* net, const float* scale,
                                     const float* sigma_dev, int32_t T, int32_t HW, int32_t ldn, void* stream) {
  if (!x || !x_out || !net || !scale || !sigma_dev) HI3D_FAIL(HI3D_EINVAL, "sampler_step_dev: null pointer");
  if (T <= 0 || HW <= 0 || ldn < 4) HI3D_FAIL(HI3D_EINVAL, "sampler_step_dev: bad size");
  const long total = (long)T * HW;
  hipLaunchKernelGGL(sampler_step_dev_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, x_out, net, scale, sigma_dev, T, HW, ldn);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_nchw_f32_to_nhwc_bf16(const float* x, void* y, int32_t N, int32_t C, int32_t HW,
                                          int32_t Cpad, void* stream) {
  if (!x || !y) HI3D_FAIL(HI3D_EINVAL, "nchw_to_nhwc: null pointer");
  if (N <= 0 || C <= 0 || HW <= 0 || Cpad < C) HI3D_FAIL(HI3D_EINVAL, "nchw_to_nhwc: bad size");
  const long total = (long)N * HW * Cpad;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, (unsigned short*)y, C, HW, Cpad, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_nhwc_to_nchw_f32(const void* x, float* y, int32_t N, int32_t C, int32_t HW,
                                     int32_t ldx, int32_t x_is_f32, void* stream) {
  if (!x || !y) HI3D_FAIL(HI3D_EINVAL, "nhwc_to_nchw: null pointer");
  if (N <= 0 || C <= 0 || HW <= 0 || ldx < C) HI3D_FAIL(HI3D_EINVAL, "nhwc_to_nchw: bad size");
  const long total = (long)N * C * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, y, C, HW, ldx, x_is_f32, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_vae_latent_prepare(const float* z, const float* w, const float* b, void* out,
                                       int32_t N, int32_t Cz, int32_t HW, int32_t Cpad, void* stream) {
  if (!z || !w || !b || !out) HI3D_FAIL(HI3D_EINVAL, "vae_latent_prepare: null pointer");
  if (N <= 0 || HW <= 0 || Cz <= 0 || Cz > 8 || Cpad < Cz || Cpad % 8) HI3D_FAIL(HI3D_EINVAL, "vae_latent_prepare: bad size");
  const long total = (long)N * HW;
  hipLaunchKernelGGL(vae_latent_prepare_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     z, w, b, (unsigned short*)out, Cz, HW, Cpad, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_softmax_rows(const float* s, void* p, int32_t R, int32_t N, int32_t lds_, int32_t ldp,
                                 float scale, void* stream) {
  if (!s || !p) HI3D_FAIL(HI3D_EINVAL, "softmax_rows: null pointer");
  if (R <= 0 || N <= 0 || lds_ < N || ldp < N) HI3D_FAIL(HI3D_EINVAL, "softmax_rows: bad size");
  if (N > 16384 || ldp > 16384) HI3D_FAIL(HI3D_ESHAPE, "softmax_rows: rows longer than 16384 not supported");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, s, (unsigned short*)p, N, lds_, ldp,
                     scale * 1.4426950408889634f);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_vae_posterior(const float* mom, const float* wq, const float* bq, const float* noise,
                                  float* z, int32_t N, int32_t Cz, int32_t HW, int32_t ldm, void* stream) {
  if (!mom || !wq || !bq || !z) HI3D_FAIL(HI3D_EINVAL, "vae_posterior: null pointer");
  if (N <= 0 || HW <= 0 || Cz <= 0 || Cz > 8 || ldm < 2 * Cz) HI3D_FAIL(HI3D_EINVAL, "vae_posterior: bad size");
  const long total = (long)N * HW;
  hipLaunchKernelGGL(vae_posterior_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     mom, wq, bq, noise, z, Cz, HW, ldm, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_v02_blend(float* lat, const float* noise, const float* z, int64_t n, float alpha,
                              float sigma, void* stream) {
  if (!lat || !noise || !z) HI3D_FAIL(HI3D_EINVAL, "v02_blend: null pointer");
  if (n <= 0) HI3D_FAIL(HI3D_EINVAL, "v02_blend: non-positive size");
  hipLaunchKernelGGL(v02_blend_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, lat, noise, z, (long)n, alpha, sigma);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_time_mix_small(const float* x, const float* w, const float* b, float* out, int32_t B,
                                   int32_t T, int32_t HW, int32_t C, int32_t ldx, void* stream) {
  if (!x || !w || !b || !out) HI3D_FAIL(HI3D_EINVAL, "time_mix_small: null pointer");
  if (B <= 0 || T <= 0 || HW <= 0 || C <= 0 || C > 4 || ldx < C) HI3D_FAIL(HI3D_EINVAL, "time_mix_small: bad size");
  const long total = (long)B * T * HW;
  hipLaunchKernelGGL(time_mix_small_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, w, b, out, T, HW, C, ldx, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_time_mix_small_k3(const float* x, const float* w, const float* b, float* out, int32_t B,
                                      int32_t T, int32_t H, int32_t W, int32_t C, int32_t ldx, void* stream) {
  if (!x || !w || !b || !out) HI3D_FAIL(HI3D_EINVAL, "time_mix_small_k3: null pointer");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || ldx < C) HI3D_FAIL(HI3D_EINVAL, "time_mix_small_k3: bad size");
  const long total = (long)B * T * H * W;
  hipLaunchKernelGGL(time_mix_small_k3_kernel, dim3(grid_for(total, 256, 1L << 30)), dim3(256), 0, (hipStream_t)stream,
                     x, w, b, out, T, H, W, C, ldx, total);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_permute_rows(const void* in, void* out, const int32_t* dims, const int32_t* perm, int32_t row_bytes, void* stream) {
  if (!in || !out || !dims || !perm) HI3D_FAIL(HI3D_EINVAL, "permute_rows: null pointer");
  if (row_bytes <= 0 || row_bytes % 16) HI3D_FAIL(HI3D_EALIGN, "permute_rows: rows must be a multiple of 16 bytes");
  if (((uintptr_t)in | (uintptr_t)out) & 15) HI3D_FAIL(HI3D_EALIGN, "permute_rows: misaligned pointer");
  long stride[4], total = 1;
  int seen = 0;
  for (int i = 3; i >= 0; --i) {
    if (dims[i] <= 0) HI3D_FAIL(HI3D_EINVAL, "permute_rows: non-positive extent");
    stride[i] = total; total *= dims[i];
  }
  for (int i = 0; i < 4; ++i) {
    if (perm[i] < 0 || perm[i] > 3) HI3D_FAIL(HI3D_EINVAL, "permute_rows: perm entries must be 0..3");
    seen |= 1 << perm[i];
  }
  if (seen != 15) HI3D_FAIL(HI3D_EINVAL, "permute_rows: perm is not a permutation");
  const int vec = row_bytes / 16;
  const long n = total * vec;
  if ((n + 255) / 256 > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, "permute_rows: grid too large");
  hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, n, vec,
                     dims[perm[0]], dims[perm[1]], dims[perm[2]], dims[perm[3]], stride[perm[0]], stride[perm[1]], stride[perm[2]], stride[perm[3]]);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// GroupNorm(32)+SiLU and LayerNorm for channels-last bf16 activations (gfx950).
// Both are HBM-bound: 16-byte vector loads/stores, fp32 statistics, one read of x
// for the statistics and one read + one write for the normalisation.
//
// GroupNorm works on x[inst][P][C]; an (instance, group) spans P pixels x C/32
// channels.  The 2-D ResBlock norm is inst = frames, P = H*W; the 3-D time_stack norm
// (statistics over t,h,w -- sgm/modules/diffusionmodules/video_model.py:71-76) is the
// same memory viewed as inst = b, P = T*H*W: no permute is ever materialised.
//   pass 1  gn_stats    : per-block partial (sum, sumsq) per group      -> ws
//   pass 2  gn_finalize : fixed-order fp64 combine -> (mean, rstd)      -> ws
//   pass 3  gn_apply    : y = silu(x * a_c + b_c), a_c = rstd*gamma_c, b_c = beta_c - mean*a_c
// Fixed summation order => bitwise reproducible run to run.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int GN_PPB_MAX = 512;    // pixels per stats block: halved (down to GN_PPB_MIN) until the grid
constexpr int GN_PPB_MIN = 64;     // has >= GN_MIN_BLOCKS blocks -- few, long blocks leave HBM idle
constexpr int GN_MIN_BLOCKS = 1024;
inline int gn_ppb(int inst, int P) {
  int ppb = GN_PPB_MAX;
  while (ppb > GN_PPB_MIN && (long)((P + ppb - 1) / ppb) * inst < GN_MIN_BLOCKS) ppb >>= 1;
  return ppb;
}
constexpr int GN_APPLY_PPB = 256;  // pixels per apply block (fewer on small grids, like the stats blocks)

__host__ __device__ inline int gn_threads(int C) {
  const int vec = C / 8;                       // 16-byte vectors per pixel
  if (vec >= 256) return vec;                  // one pixel per sweep
  return vec * (256 / vec);                    // whole pixels per sweep, <= 256 threads
}

// Two-source form (x2 != nullptr): the normalised tensor is the channel concatenation [x | x2] -- C1 = 8 * vec1 channels from
// x, the rest from x2 (the decoder's th.cat([h, hs.pop()], dim=1), video_model.py:490-499) -- read in place: a thread's
// 16-byte channel chunk lies in exactly one source, so the only change is which base / pitch it walks.
__global__ void gn_stats_kernel(const uint4* __restrict__ x, const uint4* __restrict__ x2, int vec1,
                                float* __restrict__ partial, int P, int C, int nblk, int ppb) {
  __shared__ float gs[64];
  const int vec = C >> 3, cpg = C >> 5;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int chunk_all = tid % vec, rsub = tid / vec, rows_per_sweep = nthr / vec;
  const int inst = blockIdx.y, blk = blockIdx.x;
  const int p_begin = blk * ppb, p_end = min(P, p_begin + ppb);
  const bool from2 = x2 != nullptr && chunk_all >= vec1;
  const int pitch = x2 == nullptr ? vec : (from2 ? vec - vec1 : vec1);
  const int chunk = from2 ? chunk_all - vec1 : chunk_all;
  const uint4* base = (from2 ? x2 : x) + (long)inst * P * pitch;
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
  // 8 independent 16-byte loads in flight per thread (HBM latency hiding)
  constexpr int UNR = 8;
  int p = p_begin + rsub;
  for (; p + (UNR - 1) * rows_per_sweep < p_end; p += UNR * rows_per_sweep) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = base[(long)(p + u * rows_per_sweep) * pitch + chunk];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned int w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_to_f32(w[j] & 0xffff), b = bf16_to_f32(w[j] >> 16);
        s[2 * j] += a; ss[2 * j] += a * a;
        s[2 * j + 1] += b; ss[2 * j + 1] += b * b;
      }
    }
  }
  for (; p < p_end; p += rows_per_sweep) {
    const uint4 v = base[(long)p * pitch + chunk];
    const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16_to_f32(u[j] & 0xffff), b = bf16_to_f32(u[j] >> 16);
      s[2 * j] += a; ss[2 * j] += a * a;
      s[2 * j + 1] += b; ss[2 * j + 1] += b * b;
    }
  }
  // fixed-order reduction: every thread parks its 8 per-channel sums in LDS, then
  // thread (group, sum|sumsq) walks its group's channels and pixel sub-rows in order.
  extern __shared__ float red[];            // [nthr][16]
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s[j]; red[tid * 16 + 8 + j] = ss[j]; }
  __syncthreads();
  if (tid < 64) {
    const int g = tid >> 1, which = tid & 1;
    float acc = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const int ch = c >> 3, j = c & 7;
      for (int r = 0; r < rows_per_sweep; ++r) acc += red[(r * vec + ch) * 16 + which * 8 + j];
    }
    gs[tid] = acc;
  }
  __syncthreads();
  if (tid < 64) partial[((long)inst * nblk + blk) * 64 + tid] = gs[tid];
}

// one block per (instance, group): 256 threads walk the partial blocks in a fixed interleave (four 8-byte loads in flight each),
// fp64, then a fixed-order tree -- bitwise reproducible.  (Rounds 1-3 ran ONE block per instance; with the producing conv's
// 64-row partials a 3-D norm has 4096 partial blocks per instance and two instances: 110 us on two CUs.)
constexpr int GN_FIN_PARTS = 16;        // (gn_reduce_kernel below: the sharded form keeps the one-block-per-instance walk)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int nblk,
                                                          double inv_count, float eps) {
  const int g = blockIdx.x, inst = blockIdx.y, t = threadIdx.x;
  const float2* src = (const float2*)(partial + (long)inst * nblk * 64) + g;       // block b: + 32 b
  double s = 0.0, q = 0.0;
  int b = t;
  for (; b + 768 < nblk; b += 1024) {
    const float2 v0 = src[(long)b * 32], v1 = src[(long)(b + 256) * 32], v2 = src[(long)(b + 512) * 32], v3 = src[(long)(b + 768) * 32];
    s += (double)v0.x; q += (double)v0.y; s += (double)v1.x; q += (double)v1.y;
    s += (double)v2.x; q += (double)v2.y; s += (double)v3.x; q += (double)v3.y;
  }
  for (; b < nblk; b += 256) { const float2 v = src[(long)b * 32]; s += (double)v.x; q += (double)v.y; }
  __shared__ double shs[256], shq[256];
  shs[t] = s; shq[t] = q;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { shs[t] += shs[t + o]; shq[t] += shq[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    const double mean = shs[0] * inv_count;
    double var = shq[0] * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[inst * 64 + 2 * g] = (float)mean;
    stats[inst * 64 + 2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- split form for a GroupNorm whose instance is spread over several GPUs (the 3-D time_stack norm
// of a space-sharded clip, hi3d_hip.parallel.FrameSpaceGroup): pass 1 as above, then
//   gn_reduce      : fixed-order fp64 combine of this GPU's partials -> sums[inst][32][2] (double)
//   (all-reduce of `sums` over the group: the caller's collective)
//   gn_from_sums   : (mean, rstd) from the global sums and the GLOBAL element count
// and gn_apply as above.
__global__ __launch_bounds__(64 * GN_FIN_PARTS) void gn_reduce_kernel(const float* __restrict__ partial,
                                                                      double* __restrict__ sums, int nblk) {
  const int inst = blockIdx.x, tid = threadIdx.x & 63, part = threadIdx.x >> 6;
  __shared__ double sh[GN_FIN_PARTS][64];
  double acc = 0.0;
  for (int b = part; b < nblk; b += GN_FIN_PARTS) acc += (double)partial[((long)inst * nblk + b) * 64 + tid];
  sh[part][tid] = acc;
  __syncthreads();
  if (part == 0) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < GN_FIN_PARTS; ++q) tot += sh[q][tid];
    sums[inst * 64 + tid] = tot;
  }
}

__global__ void gn_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ stats, int n,
                                    double inv_count, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over inst * 32 groups
  if (i >= n) return;
  const double mean = sums[2 * i] * inv_count;
  double var = sums[2 * i + 1] * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

template <bool SILU>
__global__ void gn_apply_kernel(const uint4* __restrict__ x, const uint4* __restrict__ x2, int vec1, uint4* __restrict__ y,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ stats, int P, int C, int appb) {
  const int vec = C >> 3, cpg = C >> 5;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int chunk = tid % vec, rsub = tid / vec, rows_per_sweep = nthr / vec;
  const int inst = blockIdx.y;
  // source of this thread's channel chunk (two-source form: see gn_stats_kernel); y is always the full-width tensor
  const bool from2 = x2 != nullptr && chunk >= vec1;
  const int xpitch = x2 == nullptr ? vec : (from2 ? vec - vec1 : vec1);
  const uint4* xs = (from2 ? x2 + (chunk - vec1) : x + chunk) + (long)inst * P * xpitch;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = chunk * 8 + j, g = c / cpg;
    const float mean = stats[inst * 64 + g * 2], rstd = stats[inst * 64 + g * 2 + 1];
    a[j] = rstd * gamma[c];
    b[j] = beta[c] - mean * a[j];
  }
  const int p_begin = blockIdx.x * appb, p_end = min(P, p_begin + appb);
  const long base = (long)inst * P * vec;
  auto norm8 = [&](const uint4 v) {
    const unsigned int u[4] = {v.x, v.y, v.z, v.w};
    unsigned int o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = bf16_to_f32(u[j] & 0xffff) * a[2 * j] + b[2 * j];
      float hi = bf16_to_f32(u[j] >> 16) * a[2 * j + 1] + b[2 * j + 1];
      if (SILU) { lo = silu_f(lo); hi = silu_f(hi); }
      o[j] = pack_bf16x2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  int p = p_begin + rsub;
  for (; p + 3 * rows_per_sweep < p_end; p += 4 * rows_per_sweep) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = xs[(long)(p + u * rows_per_sweep) * xpitch];
#pragma unroll
    for (int u = 0; u < 4; ++u) y[base + (long)(p + u * rows_per_sweep) * vec + chunk] = norm8(v[u]);
  }
  for (; p < p_end; p += rows_per_sweep) {
    y[base + (long)p * vec + chunk] = norm8(xs[(long)p * xpitch]);
  }
}

// ---- single-pass GroupNorm for SMALL (instance, group) slabs (round 6): one block per (instance, group) holds
// the slab -- P pixels x C/32 channels, as 16-byte chunks -- in REGISTERS: ONE read of x, the statistics (mean first, then the
// centred squares: fp32, fixed order -- the LayerNorm kernels' arithmetic), normalise [+ SiLU], ONE write, ONE launch.  The
// three-pass form above costs three launches and two reads, and at the 16 x 16 / 32 x 32 levels of the UNet (a few MB per tensor,
// 29 norms per stage-2 step, every norm of stage 1's lower half) it is pure launch latency: ~33 us per norm against ~12.
// Needs C/32 % 8 == 0 (whole chunks per group: 1280- and 2560-wide tensors) and P * C/256 <= MAXC * blockDim chunks per slab
// (gn_onepass_ok below: which shapes it wins on).
// Two-source form as in gn_stats_kernel (a chunk lies in exactly one source).
template <bool SILU, int MAXC>
__global__ __launch_bounds__(1024) void gn_onepass_kernel(const uint4* __restrict__ x, const uint4* __restrict__ x2, int vec1, uint4* __restrict__ y,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, int P, int C, float eps) {
  const int vec = C >> 3, cpg = C >> 5, cg = cpg >> 3;
  const int g = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const int chunk0 = g * cg, total = P * cg;
  const int p1 = x2 == nullptr ? vec : vec1, p2 = vec - vec1;
  uint4 v[MAXC];                                   // (8 chunks = 32 registers; 16 waves of a 1024-thread block get 128 each)
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int id = tid + k * nthr;
    if (id < total) {
      const int pix = id / cg, c = id - pix * cg, chunk = chunk0 + c;
      const bool from2 = x2 != nullptr && chunk >= vec1;
      v[k] = from2 ? x2[((long)inst * P + pix) * p2 + (chunk - vec1)] : x[((long)inst * P + pix) * p1 + chunk];
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    if (tid + k * nthr < total) {
      const unsigned int u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) s += bf16_to_f32(u[j] & 0xffff) + bf16_to_f32(u[j] >> 16);
    }
  }
  __shared__ float red[2][16];
  const int wv = tid >> 6, nw = nthr >> 6, lane = tid & 63;
  auto block_sum = [&](float t, int slot) {        // wave butterfly, then the waves' sums in wave order: the same for every thread
    t = wave_sum(t);
    if (lane == 0) red[slot][wv] = t;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += red[slot][i];
    return tot;
  };
  const float inv_n = 1.0f / ((float)P * (float)cpg);
  const float mean = block_sum(s, 0) * inv_n;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    if (tid + k * nthr < total) {
      const unsigned int u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16_to_f32(u[j] & 0xffff) - mean, b = bf16_to_f32(u[j] >> 16) - mean;
        ss += a * a + b * b;
      }
    }
  }
  const float rstd = rsqrtf(block_sum(ss, 1) * inv_n + eps);
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int id = tid + k * nthr;
    if (id < total) {
      const int pix = id / cg, chunk = chunk0 + (id - pix * cg);
      const f32x4 g0 = *(const f32x4*)(gamma + chunk * 8), g1 = *(const f32x4*)(gamma + chunk * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + chunk * 8), b1 = *(const f32x4*)(beta + chunk * 8 + 4);
      const float ga[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const float be[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      const unsigned int u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      unsigned int o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = (bf16_to_f32(u[j] & 0xffff) - mean) * rstd * ga[2 * j] + be[2 * j];
        float hi = (bf16_to_f32(u[j] >> 16) - mean) * rstd * ga[2 * j + 1] + be[2 * j + 1];
        if (SILU) { lo = silu_f(lo); hi = silu_f(hi); }
        o[j] = pack_bf16x2(lo, hi);
      }
      y[((long)inst * P + pix) * vec + chunk] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// the single-pass form applies: whole 16-byte chunks per group, a slab of at most 8 chunks per thread of a 1024-thread block, and a
// tensor of at most 48 MB (HI3D_GN_ONEPASS=0 switches it off, =<MB> moves the bound).  Measured per shape on the MI355X
// (profiles/r06g_gn_onepass_per_shape.log, three-pass -> single-pass): [32 x 256 x 1280] 30.0 -> 16.3 us, [32 x 256 x 2560 concat]
// 45.0 -> 30.9, [32 x 64 x 1280] 26.6 -> 13.6, [2 x 1024 x 1280] 26.1 -> 14.1; [16 x 1024 x 1280] (42 MB) 33.3 -> 33.4; beyond
// that the group-strided 80-byte reads lose to the three-pass form's full rows ([32 x 1024 x 1280], 84 MB: 49 -> 66 us).  A
// streaming variant for larger slabs (3-D norms of the 16 x 16 level: 64 blocks x 0.3 MB, read twice through the L2) was built
// and dropped: 29.5 -> 46 us -- two instances x 32 groups do not fill the chip.
inline bool gn_onepass_ok(int inst, int P, int C) {
  static const int mb = [] { const char* e = getenv("HI3D_GN_ONEPASS"); return e ? atoi(e) : 48; }();
  const int cpg = C >> 5;
  return mb > 0 && cpg % 8 == 0 && (long)P * (cpg >> 3) <= 8 * 1024 && (long)inst * P * C * 2 <= (long)mb << 20;
}
template <bool SILU>
int gn_onepass_launch(const void* x, const void* x2, int C1, void* y, const float* gamma, const float* beta, int inst, int P, int C,
                      float eps, hipStream_t s) {
  const long total = (long)P * (C >> 8);
  const dim3 grid(32, inst);
  hipLaunchKernelGGL((gn_onepass_kernel<SILU, 8>), grid, dim3(total <= 8 * 256 ? 256 : 1024), 0, s, (const uint4*)x, (const uint4*)x2, C1 / 8,
                     (uint4*)y, gamma, beta, P, C, eps);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// ---- LayerNorm: one wave per row, up to 4 16-byte vectors per lane (C <= 2048)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const uint4* __restrict__ x, uint4* __restrict__ y, uint4* __restrict__ sum_out,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ addvec, int rpg, int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int vec = C >> 3;
  const uint4* xr = x + row * vec;
  const float* av = addvec ? addvec + (row / rpg) * (long)C : nullptr;
  float f[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = lane + i * 64;
    if (v < vec) {
      const uint4 q = xr[v];
      const unsigned int u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[i][2 * j] = bf16_to_f32(u[j] & 0xffff);
        f[i][2 * j + 1] = bf16_to_f32(u[j] >> 16);
      }
      if (av) {
        const f32x4 a0 = *(const f32x4*)(av + v * 8), a1 = *(const f32x4*)(av + v * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[i][j] += a0[j]; f[i][4 + j] += a1[j]; }
        if (sum_out) {
          sum_out[row * vec + v] = make_uint4(pack_bf16x2(f[i][0], f[i][1]), pack_bf16x2(f[i][2], f[i][3]),
                                              pack_bf16x2(f[i][4], f[i][5]), pack_bf16x2(f[i][6], f[i][7]));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (lane + i * 64 < vec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = lane + i * 64;
    if (v < vec) {
      const f32x4 g0 = *(const f32x4*)(gamma + v * 8), g1 = *(const f32x4*)(gamma + v * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + v * 8), b1 = *(const f32x4*)(beta + v * 8 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
        o[4 + j] = (f[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      y[row * vec + v] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                    pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
  }
}

// ---- LayerNorm for C = 40 * LPR (320 / 640 / 1280: every transformer width of the UNet): LPR lanes per row, five
// 16-byte vectors per lane, 64 / LPR rows per wave.  The one-wave-per-row kernel above leaves 24 of 64 lanes idle at
// C = 320 (40 vectors) and runs at 3.3 TB/s there against 4.6 at C = 1280; here every lane is busy at every width and a
// load instruction covers whole 128-byte lines (LPR consecutive vectors of each of its rows).
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(
    const uint4* __restrict__ x, uint4* __restrict__ y, uint4* __restrict__ sum_out,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ addvec, int rpg, int R, float eps) {
  constexpr int NV = 5, RPW = 64 / LPR, vec = NV * LPR, C = vec * 8;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  if (row >= R) return;                        // (uniform over the LPR lanes of a row: the shuffles below stay inside them)
  const uint4* xr = x + row * vec;
  const float* av = addvec ? addvec + (row / rpg) * (long)C : nullptr;
  float f[NV][8];
  uint4 q[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) q[i] = xr[sub + i * LPR];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = sub + i * LPR;
    const unsigned int u[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[i][2 * j] = bf16_to_f32(u[j] & 0xffff);
      f[i][2 * j + 1] = bf16_to_f32(u[j] >> 16);
    }
    if (av) {
      const f32x4 a0 = *(const f32x4*)(av + v * 8), a1 = *(const f32x4*)(av + v * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[i][j] += a0[j]; f[i][4 + j] += a1[j]; }
      if (sum_out) {
        sum_out[row * vec + v] = make_uint4(pack_bf16x2(f[i][0], f[i][1]), pack_bf16x2(f[i][2], f[i][3]),
                                            pack_bf16x2(f[i][4], f[i][5]), pack_bf16x2(f[i][6], f[i][7]));
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[i][j];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; ss += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = rsqrtf(ss / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = sub + i * LPR;
    const f32x4 g0 = *(const f32x4*)(gamma + v * 8), g1 = *(const f32x4*)(gamma + v * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(beta + v * 8), b1 = *(const f32x4*)(beta + v * 8 + 4);
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
      o[4 + j] = (f[i][4 + j] - mean) * rstd * g1[j] + b1[j];
    }
    y[row * vec + v] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                  pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

}  // namespace

extern "C" int32_t hi3d_gn_partial_blocks(int32_t P, int32_t C) {
  (void)C;
  return (P + GN_PPB_MIN - 1) / GN_PPB_MIN;   // upper bound over the block sizes the launcher may pick
}

extern "C" int64_t hi3d_gn_workspace_floats(int32_t inst, int32_t P, int32_t C) {
  return (int64_t)inst * hi3d_gn_partial_blocks(P, C) * 64 + (int64_t)inst * 64;
}

namespace {
int groupnorm_launch(const void* x, const void* x2, int C1, void* y, const float* gamma, const float* beta,
                     float* ws, int32_t inst, int32_t P, int32_t C, float eps, int32_t apply_silu, void* stream) {
  if (!x || !y || !gamma || !beta || !ws) HI3D_FAIL(HI3D_EINVAL, "groupnorm: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (((uintptr_t)x | (uintptr_t)y) & 15) HI3D_FAIL(HI3D_EALIGN, "groupnorm: x/y not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (gn_onepass_ok(inst, P, C) && (x2 == nullptr || (C1 % 8 == 0 && ((uintptr_t)x2 & 15) == 0)))
    return apply_silu ? gn_onepass_launch<true>(x, x2, C1, y, gamma, beta, inst, P, C, eps, s)
                      : gn_onepass_launch<false>(x, x2, C1, y, gamma, beta, inst, P, C, eps, s);
  const int ppb = gn_ppb(inst, P);
  const int nblk = (P + ppb - 1) / ppb;
  float* partial = ws;
  float* stats = ws + (long)inst * hi3d_gn_partial_blocks(P, C) * 64;
  const int nthr = gn_threads(C);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, inst), dim3(nthr), nthr * 16 * sizeof(float), s, (const uint4*)x, (const uint4*)x2, C1 / 8, partial, P, C, nblk, ppb);
  HI3D_LAUNCH_CHECK();
  const double inv_count = 1.0 / ((double)P * (double)(C / 32));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, inst), dim3(256), 0, s, partial, stats, nblk, inv_count, eps);
  HI3D_LAUNCH_CHECK();
  const int appb = ppb < GN_APPLY_PPB ? ppb : GN_APPLY_PPB;
  const int ablk = (P + appb - 1) / appb;
  if (apply_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)x2, C1 / 8, (uint4*)y, gamma, beta, stats, P, C, appb);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)x2, C1 / 8, (uint4*)y, gamma, beta, stats, P, C, appb);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}
// ---- GroupNorm (no activation) folded into the following linear layer: one block per (output row n, instance f)
//   Wf[f][n][k] = bf16(W[n][k] * a[f][k]),  biasf[f][n] = bias[n] + sum_k W[n][k] * b[f][k]     a = rstd * gamma, b = beta - mean * a
// 64 threads walk the C input channels in a fixed interleave, fixed-order wave reduction: reproducible.
__global__ __launch_bounds__(64) void gn_fold_linear_kernel(const unsigned short* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ stats, unsigned short* __restrict__ Wf,
                                                            float* __restrict__ biasf, int C, int N) {
  const int n = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
  const int cpg = C >> 5;
  const unsigned short* wr = W + (long)n * ldw;
  unsigned short* wo = Wf + ((long)f * N + n) * C;
  float acc = 0.f;
  for (int k = lane * 2; k < C; k += 128) {                   // two channels per lane and sweep (C % 32 == 0: even)
    const unsigned int wp = *(const unsigned int*)(wr + k);
    const float w0 = bf16_to_f32(wp & 0xffff), w1 = bf16_to_f32(wp >> 16);
    const float* st0 = stats + f * 64 + (k / cpg) * 2;
    const float* st1 = stats + f * 64 + ((k + 1) / cpg) * 2;
    const float a0 = st0[1] * gamma[k], a1 = st1[1] * gamma[k + 1];
    const float b0 = beta[k] - st0[0] * a0, b1 = beta[k + 1] - st1[0] * a1;
    *(unsigned int*)(wo + k) = pack_bf16x2(w0 * a0, w1 * a1);
    acc += w0 * b0 + w1 * b1;
  }
  acc = wave_sum(acc);
  if (lane == 0) biasf[(long)f * N + n] = acc + (bias ? bias[n] : 0.f);
}

}  // namespace

extern "C" int hi3d_groupnorm_fold_linear(const void* x, float* ws, const float* gamma, const float* beta, float eps,
                                          int32_t inst, int32_t P, int32_t C, const void* W, int32_t ldw, const float* bias,
                                          int32_t N, void* Wf, float* biasf, void* stream) {
  if (!x || !ws || !gamma || !beta || !W || !Wf || !biasf) HI3D_FAIL(HI3D_EINVAL, "groupnorm_fold_linear: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0 || N <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm_fold_linear: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (ldw < C || ldw % 2) HI3D_FAIL(HI3D_EALIGN, "groupnorm_fold_linear: ldw < C or odd");
  if (((uintptr_t)x & 15) || ((uintptr_t)W & 3) || ((uintptr_t)Wf & 15)) HI3D_FAIL(HI3D_EALIGN, "groupnorm_fold_linear: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const int ppb = gn_ppb(inst, P);
  const int nblk = (P + ppb - 1) / ppb;
  const int nthr = gn_threads(C);
  float* stats = ws + (long)inst * nblk * 64;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, inst), dim3(nthr), nthr * 16 * sizeof(float), s, (const uint4*)x, (const uint4*)nullptr, 0, ws, P, C, nblk, ppb);
  HI3D_LAUNCH_CHECK();
  const double inv_count = 1.0 / ((double)P * (double)(C / 32));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, inst), dim3(256), 0, s, ws, stats, nblk, inv_count, eps);
  HI3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_fold_linear_kernel, dim3(N, inst), dim3(64), 0, s, (const unsigned short*)W, ldw, bias, gamma, beta, stats,
                     (unsigned short*)Wf, biasf, C, N);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// ... with the partial sums of x already in ws (the producer of x emitted them: hi3d_gemm_desc.gn_partial, 64-row blocks): finalize
// + fold only -- x itself is not read at all by the norm (round 6)
extern "C" int hi3d_groupnorm_fold_linear_from_partials(float* ws, const float* gamma, const float* beta, float eps,
                                                        int32_t inst, int32_t P, int32_t C, const void* W, int32_t ldw,
                                                        const float* bias, int32_t N, void* Wf, float* biasf, void* stream) {
  if (!ws || !gamma || !beta || !W || !Wf || !biasf) HI3D_FAIL(HI3D_EINVAL, "groupnorm_fold_linear: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0 || N <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm_fold_linear: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (P % 64) HI3D_FAIL(HI3D_ESHAPE, "groupnorm_fold_linear_from_partials: P must be a multiple of the 64-row partial blocks");
  if (ldw < C || ldw % 2) HI3D_FAIL(HI3D_EALIGN, "groupnorm_fold_linear: ldw < C or odd");
  if (((uintptr_t)W & 3) || ((uintptr_t)Wf & 15)) HI3D_FAIL(HI3D_EALIGN, "groupnorm_fold_linear: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = P / 64;
  float* stats = ws + (long)inst * nblk * 64;
  const double inv_count = 1.0 / ((double)P * (double)(C / 32));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, inst), dim3(256), 0, s, ws, stats, nblk, inv_count, eps);
  HI3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_fold_linear_kernel, dim3(N, inst), dim3(64), 0, s, (const unsigned short*)W, ldw, bias, gamma, beta, stats,
                     (unsigned short*)Wf, biasf, C, N);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_groupnorm_silu(const void* x, void* y, const float* gamma, const float* beta,
                                   float* ws, int32_t inst, int32_t P, int32_t C, float eps,
                                   int32_t apply_silu, void* stream) {
  return groupnorm_launch(x, nullptr, C, y, gamma, beta, ws, inst, P, C, eps, apply_silu, stream);
}

extern "C" int hi3d_groupnorm_silu_from_partials(const void* x, void* y, const float* gamma, const float* beta,
                                                 float* ws, int32_t inst, int32_t P, int32_t C, float eps,
                                                 int32_t apply_silu, void* stream) {
  if (!x || !y || !gamma || !beta || !ws) HI3D_FAIL(HI3D_EINVAL, "groupnorm: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (P % 64) HI3D_FAIL(HI3D_ESHAPE, "groupnorm_from_partials: P must be a multiple of the 64-row partial blocks");
  if (((uintptr_t)x | (uintptr_t)y) & 15) HI3D_FAIL(HI3D_EALIGN, "groupnorm: x/y not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (gn_onepass_ok(inst, P, C))                         // small slab: one launch and one read either way -- the partial sums are not needed
    return apply_silu ? gn_onepass_launch<true>(x, nullptr, C, y, gamma, beta, inst, P, C, eps, s)
                      : gn_onepass_launch<false>(x, nullptr, C, y, gamma, beta, inst, P, C, eps, s);
  const int nblk = P / 64;                               // == hi3d_gn_partial_blocks(P, C): the stats slot sits right behind
  float* stats = ws + (long)inst * nblk * 64;
  const double inv_count = 1.0 / ((double)P * (double)(C / 32));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, inst), dim3(256), 0, s, ws, stats, nblk, inv_count, eps);
  HI3D_LAUNCH_CHECK();
  const int ppb = gn_ppb(inst, P);
  const int appb = ppb < GN_APPLY_PPB ? ppb : GN_APPLY_PPB;
  const int ablk = (P + appb - 1) / appb;
  const int nthr = gn_threads(C);
  if (apply_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)nullptr, 0, (uint4*)y, gamma, beta, stats, P, C, appb);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)nullptr, 0, (uint4*)y, gamma, beta, stats, P, C, appb);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_groupnorm_silu_cat2(const void* x1, const void* x2, void* y, const float* gamma, const float* beta,
                                        float* ws, int32_t inst, int32_t P, int32_t C1, int32_t C2, float eps,
                                        int32_t apply_silu, void* stream) {
  if (!x2) HI3D_FAIL(HI3D_EINVAL, "groupnorm_cat2: null pointer");
  if (C1 <= 0 || C2 <= 0 || C1 % 8 || C2 % 8) HI3D_FAIL(HI3D_ESHAPE, "groupnorm_cat2: C1 and C2 must be positive multiples of 8");
  if ((uintptr_t)x2 & 15) HI3D_FAIL(HI3D_EALIGN, "groupnorm_cat2: x2 not 16-byte aligned");
  return groupnorm_launch(x1, x2, C1, y, gamma, beta, ws, inst, P, C1 + C2, eps, apply_silu, stream);
}

extern "C" int hi3d_groupnorm_partial_sums(const void* x, float* ws, double* sums, int32_t inst, int32_t P,
                                           int32_t C, void* stream) {
  if (!x || !ws || !sums) HI3D_FAIL(HI3D_EINVAL, "groupnorm_partial_sums: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (((uintptr_t)x & 15) || ((uintptr_t)sums & 7)) HI3D_FAIL(HI3D_EALIGN, "groupnorm_partial_sums: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const int ppb = gn_ppb(inst, P);
  const int nblk = (P + ppb - 1) / ppb;
  const int nthr = gn_threads(C);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, inst), dim3(nthr), nthr * 16 * sizeof(float), s, (const uint4*)x, (const uint4*)nullptr, 0, ws, P, C, nblk, ppb);
  HI3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_reduce_kernel, dim3(inst), dim3(64 * GN_FIN_PARTS), 0, s, ws, sums, nblk);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_groupnorm_apply_sums(const void* x, void* y, const float* gamma, const float* beta,
                                         const double* sums, float* ws, int32_t inst, int32_t P, int32_t C,
                                         int64_t count_per_group, float eps, int32_t apply_silu, void* stream) {
  if (!x || !y || !gamma || !beta || !sums || !ws) HI3D_FAIL(HI3D_EINVAL, "groupnorm_apply_sums: null pointer");
  if (inst <= 0 || P <= 0 || C <= 0 || count_per_group <= 0) HI3D_FAIL(HI3D_EINVAL, "groupnorm: non-positive size");
  if (C % 32 || C > 8192) HI3D_FAIL(HI3D_ESHAPE, "groupnorm: C must be a multiple of 32 (<= 8192)");
  if (((uintptr_t)x | (uintptr_t)y) & 15) HI3D_FAIL(HI3D_EALIGN, "groupnorm: x/y not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* stats = ws;                         // inst * 64 floats
  const int n = inst * 32;
  hipLaunchKernelGGL(gn_from_sums_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sums, stats, n, 1.0 / (double)count_per_group, eps);
  HI3D_LAUNCH_CHECK();
  const int ppb = gn_ppb(inst, P);
  const int appb = ppb < GN_APPLY_PPB ? ppb : GN_APPLY_PPB;
  const int ablk = (P + appb - 1) / appb;
  const int nthr = gn_threads(C);
  if (apply_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)nullptr, 0, (uint4*)y, gamma, beta, stats, P, C, appb);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ablk, inst), dim3(nthr), 0, s, (const uint4*)x, (const uint4*)nullptr, 0, (uint4*)y, gamma, beta, stats, P, C, appb);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_layernorm(const void* x, void* y, void* sum_out, const float* gamma,
                              const float* beta, const float* addvec, int32_t rows_per_group,
                              int32_t R, int32_t C, float eps, void* stream) {
  if (!x || !y || !gamma || !beta) HI3D_FAIL(HI3D_EINVAL, "layernorm: null pointer");
  if (R <= 0 || C <= 0) HI3D_FAIL(HI3D_EINVAL, "layernorm: non-positive size");
  if (C % 8 || C > 2048) HI3D_FAIL(HI3D_ESHAPE, "layernorm: C must be a multiple of 8 and <= 2048");
  if (addvec && rows_per_group < 1) HI3D_FAIL(HI3D_EINVAL, "layernorm: rows_per_group < 1");
  if (sum_out && !addvec) HI3D_FAIL(HI3D_EINVAL, "layernorm: sum_out without addvec");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)sum_out) & 15) HI3D_FAIL(HI3D_EALIGN, "layernorm: not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int rpg = rows_per_group < 1 ? 1 : rows_per_group;
  // widths 320 / 640 / 1280: packed rows, every lane busy (HI3D_LN_PACKED=0 falls back to one wave per row)
  static const bool packed_on = [] { const char* e = getenv("HI3D_LN_PACKED"); return !e || atoi(e) != 0; }();
  if (packed_on && (C == 320 || C == 640 || C == 1280)) {
    const int lpr = C / 40, rows_per_block = 4 * (64 / lpr);
    const int g = (R + rows_per_block - 1) / rows_per_block;
#define LNP_LAUNCH(LPR) hipLaunchKernelGGL(layernorm_packed_kernel<LPR>, dim3(g), dim3(256), 0, s, (const uint4*)x, (uint4*)y, (uint4*)sum_out, gamma, beta, addvec, rpg, R, eps)
    if (lpr == 8) LNP_LAUNCH(8); else if (lpr == 16) LNP_LAUNCH(16); else LNP_LAUNCH(32);
#undef LNP_LAUNCH
    HI3D_LAUNCH_CHECK();
    return HI3D_OK;
  }
  const int grid = (R + 3) / 4, vec = C / 8;
#define LN_LAUNCH(NV) hipLaunchKernelGGL(layernorm_kernel<NV>, dim3(grid), dim3(256), 0, s, (const uint4*)x, (uint4*)y, (uint4*)sum_out, gamma, beta, addvec, rpg, R, C, eps)
  if (vec <= 64) LN_LAUNCH(1);
  else if (vec <= 128) LN_LAUNCH(2);
  else if (vec <= 192) LN_LAUNCH(3);
  else LN_LAUNCH(4);
#undef LN_LAUNCH
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// Fused GEGLU feed-forward for the 320-channel level, second form (round 2): the same computation as ffn.hip
//
//     out = ( GEGLU(X W1^T + b1) W2^T + b2 + R1 ) [ * a1 + a2 * R2 ]
//
// (FeedForward of BasicTransformerBlock / VideoTransformerBlock, sgm/modules/attention.py:83-119; the bracket is the
// AlphaBlender of the temporal block, sgm/modules/video_attention.py:290-294), restructured around what the round-2
// measurements showed: the first form runs its MFMAs at 37 % utilisation because all 8 waves move in lock step through
// 16-MFMA K steps separated by barriers -- both waves of a SIMD wait for LDS together, then both compute, then both do
// their GELU.  Here
//   * a wave owns 32 rows and keeps ALL of its X tile in registers (80): the first GEMM reads only weights from LDS
//     and the 48 KiB of X slabs are gone;
//   * the 64-column chunk of the hidden dimension is cut into six phases of 20 MFMAs,
//         [load segment: fragment reads, a slice of GELU, LDS-DMA requests] s_barrier [20 MFMAs] s_barrier,
//     and waves 4..7 (rows 64..127: their hg rows are their own) run one barrier behind waves 0..3, so on every SIMD
//     one wave is in its MFMA segment while the other reads fragments / evaluates GELU;
//   * the chunk is software-pipelined so that no phase is GELU only: the second half of a chunk's GELU and the second
//     half of its down-projection ride in the first three phases of the NEXT chunk,
//         Ph1  W1A(c) k 0-159   x X -> acc1A        | GELU_B(c-1) rows 0-15   -> hgB
//         Ph2  W1A(c) k 160-319 x X -> acc1A        | GELU_B(c-1) rows 16-31  -> hgB
//         Ph3  hgB(c-1) x W2(c-1)[:, 32:64] -> acc2 | GELU_A(c)   rows 0-15   -> hgA     request W1A(c+1)
//         Ph4  W1B(c) k 0-159   x X -> acc1B        | GELU_A(c)   rows 16-31  -> hgA     request W2(c)[:, 32:64]
//         Ph5  W1B(c) k 160-319 x X -> acc1B
//         Ph6  hgA(c)   x W2(c)[:, 0:32]    -> acc2 |                                    request W1B(c+1)
//     (A / B: the first / second 32 hidden columns of the chunk = its first / second 64 packed W1 rows.)
// LDS: W1A 40 KiB + W1B 40 KiB (five 8 KiB K slabs each, the swizzled stage layout of gemm.hip) + the two 32-column
// halves of the chunk's W2 slab 20 + 20 KiB + hg 16 KiB + the bias vector 10 KiB = 146 KiB.  All waits are counted vmcnt's, all barriers raw.
// Operand layouts (packed GEGLU rows, K-major W2) are those of ffn.hip / gemm.hip: the packed weights are shared.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Ffn2Params {
  const char* X; const char* W1; const float* b1; const char* W2; const float* b2;
  const unsigned short* R1; const unsigned short* R2; const float* a1; const float* a2;
  unsigned short* out;
  int M, ldx, ldo, ldr1, ldr2, rpg;
};

constexpr int GC = 320;                    // channels
constexpr int GH = 4 * GC;                 // hidden width after GEGLU
constexpr int GBM = 128;                   // rows per block
constexpr int GNCH = GH / 64;              // 20 chunks of 64 hidden columns
constexpr int W1S = 5 * 8192;              // one W1 stage: 64 packed rows x 320 k = five [64][64] K slabs
constexpr int W2H = 320 * 64;              // half of a chunk's W2 slab: 320 rows x 32 k, 64-byte rows
constexpr int HGB = GBM * 128;             // hg: 128 rows x 64 hidden
constexpr int B1B = 8 * GC * 4;            // the packed first-layer bias, fp32 (read in every GELU slice: no VMEM loads inside the loop)
constexpr int FFN2_LDS = 2 * W1S + 2 * W2H + HGB + B1B;      // 149,504 B
constexpr int W1_OPS = 5;                  // LDS-DMA instructions per wave and W1 stage
constexpr int W2_OPS_MIN = 2;              // ... per W2 half: waves 0-3 issue 3, waves 4-7 issue 2

__global__ __launch_bounds__(512, 2) void ffn2_geglu_c320_kernel(const Ffn2Params p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sW1 = smem;                                   // stage A at 0, stage B at W1S
  char* const sW2 = smem + 2 * W1S;                         // k 0..31 half at 0, k 32..63 half at W2H
  char* const sHG = sW2 + 2 * W2H;
  float* const sB1 = (float*)(sHG + HGB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = w >= 4;
  const int wmg = w >> 1, wn = w & 1;                       // 32-row block of the tile (0..3), column half
  const int fr = lane & 15, fg = lane >> 4;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int m0 = blockIdx.x * GBM;
  constexpr unsigned INV = 0x80000000u;

  // ---- X: the wave's 32 rows, all 320 channels, in MFMA B-operand layout (lane (fr, fg): row fr of each 16-row block,
  // k = 32*kk + 8*fg .. +7)
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (long)m0 * p.ldx * 2), 0, 0x7fffffff, 0x00020000);
  bf16x8 xr[10][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int r = wmg * 32 + mt * 16 + fr;
    const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + fg * 16) : INV;
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) xr[kk][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, kk * 64, 0));
  }

  // ---- weight loaders.  W1 stage = 40 pieces of 1 KiB (8 packed rows x 128 B of one K slab): wave w moves row group w
  // of every slab (piece w + 8 i -> slab i); W2 half = 20 pieces (16 rows x 64 B): wave w moves pieces w, w+8, w+16 (< 20).
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, 0x7fffffff, 0x00020000);
  unsigned w1_voff;
  {
    const int j = w * 8 + lrow, jw = j & 31;                 // packed row within the stage; row within its wave tile (32)
    const int fi = (jw >> 3) * 4 + (jw & 3);                 // MFMA row index that reads it
    w1_voff = (unsigned)(j * GC * 2 + ((lslot ^ ((fi >> 1) & 7)) << 4));
  }
  unsigned w2_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) w2_voff[i] = (unsigned)(((w + 8 * i) * 16 + (lane >> 2)) * GH * 2 + (lane & 3) * 16);
  auto issue_w1 = [&](int c, int stage) {                    // stage 0 = A (packed rows 128c .. +63), 1 = B
    const int soff = (c * 128 + stage * 64) * GC * 2;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (LDS_AS void*)(sW1 + stage * W1S + i * 8192 + w * 1024), 16, w1_voff, soff + i * 128, 0, 0);
  };
  auto issue_w2 = [&](int c, int half) {                     // hidden columns 64c + 32 half .. +31 of every W2 row
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (w + 8 * i < 20)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (LDS_AS void*)(sW2 + half * W2H + (w + 8 * i) * 1024), 16, w2_voff[i], c * 128 + half * 64, 0, 0);
  };

  // ---- fragment read offsets
  const int f_sw = (fr >> 1) & 7;
  const int w1_off = (wn * 32 + (fr >> 2) * 8 + (fr & 3)) * 128;                  // + nt * 512, + slab * 8192
  const int w2_off = (wn * 160 + (fr >> 2) * 40 + (fr & 3)) * 64 + fg * 16;       // + nt * 256   (64-byte rows, linear)
  const int hg_off = (wmg * 32 + fr) * 128;                                        // + mt * 2048, + swizzled chunk
  // this lane's packed columns of a stage: wn*32 + fg*8 + nt*4 .. +3  (= stage-local hidden columns wn*16 + fg*4 + nt*2, +1)
  const float* b1p = sB1 + wn * 32 + fg * 8;
  // its 8 bytes of hg (4 hidden columns) for row fr of a 16-row block: + stage * 64 bytes (4 chunks of 16 B) handled by the xor below
  auto hg_wptr = [&](int stage, int mt) {
    return sHG + hg_off + mt * 2048 + ((((stage * 4 + wn * 2 + (fg >> 1))) ^ f_sw) << 4) + (fg & 1) * 8;
  };

  f32x4 acc2[2][10];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1a[2][2], acc1b[2][2];

  // value * gelu(gate) of one 16-row block of a stage's accumulators -> 4 hidden columns per lane -> hg
  auto gelu_store = [&](const f32x4 (&acc)[2][2], int c, int stage, int mt) {
    unsigned int u[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 b = *(const f32x4*)(b1p + (c * 128 + stage * 64) + nt * 4);
      f32x4 v = acc[mt][nt];
      v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
      const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
      u[nt] = pack_bf16x2(v[0] * gl[0], v[1] * gl[1]);
    }
    *(uint2*)hg_wptr(stage, mt) = make_uint2(u[0], u[1]);
  };
  // first GEMM, one phase: 5 of the 10 half K steps of a stage
  auto g1_read = [&](bf16x8 (&wf)[5][2], int stage, int half) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int kk = half * 5 + q;
      const char* s = sW1 + stage * W1S + (kk >> 1) * 8192 + w1_off + ((((kk & 1) * 4 + fg) ^ f_sw) << 4);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[q][nt] = *(const bf16x8*)(s + nt * 512);
    }
  };
  auto g1_mfma = [&](const bf16x8 (&wf)[5][2], f32x4 (&acc)[2][2], int half) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][nt], xr[half * 5 + q][mt],
                                                                  (half == 0 && q == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nt], 0, 0, 0);
  };
  // second GEMM, one phase: K half `kh` (32 hidden columns) of the chunk whose hg half is ready
  auto g2_read = [&](bf16x8 (&xf)[2], bf16x8 (&wf)[10], int kh) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xf[mt] = *(const bf16x8*)(sHG + hg_off + mt * 2048 + (((kh * 4 + fg) ^ f_sw) << 4));
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) wf[nt] = *(const bf16x8*)(sW2 + kh * W2H + w2_off + nt * 256);
  };
  auto g2_mfma = [&](const bf16x8 (&xf)[2], const bf16x8 (&wf)[10]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc2[mt][nt], 0, 0, 0);
  };
  auto seg_end = [&]() {                          // end of a load segment: fragment reads / hg writes retired, then the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto cseg_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: the bias vector and both W1 stages of chunk 0
  for (int i = tid; i < 8 * GC / 4; i += 512) *(f32x4*)(sB1 + i * 4) = *(const f32x4*)(p.b1 + i * 4);
  issue_w1(0, 0); issue_w1(0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();          // the stagger

  for (int c = 0; c <= GNCH; ++c) {
    const bool cur = c < GNCH, prev = c > 0;       // this chunk's first GEMM exists / the previous chunk's second half is pending
    // ---- Ph1
    {
      bf16x8 wf[5][2];
      if (cur) { issue_w2(c, 0); g1_read(wf, 0, 0); }
      if (prev) gelu_store(acc1b, c - 1, 1, 0);
      seg_end();
      if (cur) g1_mfma(wf, acc1a, 0);
      cseg_end();
    }
    // ---- Ph2
    {
      bf16x8 wf[5][2];
      if (prev) {                                  // W2(c-1)[:, 32:64] (requested in Ph4 of the previous chunk) has landed
        if (cur) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + W2_OPS_MIN) : "memory");   // younger: W1B(c), W2(c)[:, 0:32]
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (cur) g1_read(wf, 0, 1);
      if (prev) gelu_store(acc1b, c - 1, 1, 1);
      seg_end();
      if (cur) g1_mfma(wf, acc1a, 1);
      cseg_end();
    }
    // ---- Ph3
    {
      bf16x8 xf[2], wf[10];
      if (cur) {                                   // W1B(c) has landed (younger: W2(c)[:, 0:32]); W1A is read out: request W1A(c+1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2_OPS_MIN) : "memory");
        if (c + 1 < GNCH) issue_w1(c + 1, 0);
      }
      if (prev) g2_read(xf, wf, 1);
      if (cur) gelu_store(acc1a, c, 0, 0);
      seg_end();
      if (prev) g2_mfma(xf, wf);
      cseg_end();
    }
    if (!cur) break;
    // ---- Ph4
    {
      bf16x8 wf[5][2];
      issue_w2(c, 1);
      g1_read(wf, 1, 0);
      gelu_store(acc1a, c, 0, 1);
      seg_end();
      g1_mfma(wf, acc1b, 0);
      cseg_end();
    }
    // ---- Ph5
    {
      bf16x8 wf[5][2];
      // W2(c)[:, 0:32] (requested in Ph1) has landed; younger: W1A(c+1) if it exists, W2(c)[:, 32:64]
      if (c + 1 < GNCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + W2_OPS_MIN) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2_OPS_MIN) : "memory");
      g1_read(wf, 1, 1);
      seg_end();
      g1_mfma(wf, acc1b, 1);
      cseg_end();
    }
    // ---- Ph6
    {
      bf16x8 xf[2], wf[10];
      if (c + 1 < GNCH) {                          // W1A(c+1) has landed (younger: W2(c)[:, 32:64]); W1B is read out: request W1B(c+1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2_OPS_MIN) : "memory");
        issue_w1(c + 1, 1);
      }
      g2_read(xf, wf, 0);
      seg_end();
      g2_mfma(xf, wf);
      cseg_end();
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();         // the leading half catches the stagger up
  __syncthreads();                                 // every wave is done with the slabs

  // ---- epilogue: every wave stages its 32 x 160 tile, 16 rows at a time, through a private fp32 slab and issues its
  // residual loads / stores row-contiguous, 16 bytes per lane (as the wide GEMM tiles do)
  constexpr int EW = 160, EROW = EW * 4 + 16, ESLAB = 16 * EROW, ECPR = EW / 8, ECH = (16 * ECPR + 63) / 64;   // 5 chunks per lane and pass
  static_assert(8 * ESLAB <= FFN2_LDS, "epilogue slabs");
  char* const slab = smem + w * ESLAB;
  const int ecol0 = wn * EW;
  const long erow0 = (long)m0 + wmg * 32;
  const int rows_left = p.M - (int)erow0;
  int e_row[ECH], e_col[ECH];
#pragma unroll
  for (int i = 0; i < ECH; ++i) {
    const int cidx = lane + 64 * i;
    e_row[i] = cidx / ECPR;
    e_col[i] = (cidx - e_row[i] * ECPR) * 8;
  }
  auto e_ok = [&](int i, int mt) { return e_row[i] + mt * 16 < rows_left; };
  auto fetch = [&](const unsigned short* R, int ldr, int mt, int i) -> u32x4 {
    if (!R || !e_ok(i, mt)) return u32x4{0u, 0u, 0u, 0u};
    const unsigned short* rp = R + (erow0 + mt * 16 + e_row[i]) * ldr + ecol0 + e_col[i];
    if ((ldr & 7) == 0 && (((uintptr_t)R) & 15) == 0) return *(const u32x4*)rp;
    const u32x2 lo = *(const u32x2*)rp, hi = *(const u32x2*)(rp + 4);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  const bool o16 = (p.ldo & 7) == 0 && (((uintptr_t)p.out) & 15) == 0;
  u32x4 q1[2][ECH], q2[2][ECH];
#pragma unroll
  for (int i = 0; i < ECH; ++i) { q1[0][i] = fetch(p.R1, p.ldr1, 0, i); q2[0][i] = fetch(p.R2, p.ldr2, 0, i); }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) *(f32x4*)(slab + fr * EROW + (fg * 40 + nt * 4) * 4) = acc2[mt][nt];
    if (mt + 1 < 2) {
#pragma unroll
      for (int i = 0; i < ECH; ++i) { q1[1][i] = fetch(p.R1, p.ldr1, 1, i); q2[1][i] = fetch(p.R2, p.ldr2, 1, i); }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ECH; ++i) {
      if (!e_ok(i, mt)) continue;
      const long m = erow0 + mt * 16 + e_row[i];
      const int n = ecol0 + e_col[i];
      const char* sp = slab + e_row[i] * EROW + e_col[i] * 4;
      const f32x4 lo = *(const f32x4*)sp, hi = *(const f32x4*)(sp + 16);
      const f32x4 b0 = *(const f32x4*)(p.b2 + n), b1 = *(const f32x4*)(p.b2 + n + 4);
      const int grp = (p.a1 || p.a2) ? (int)(m / p.rpg) : 0;
      const float s1 = p.a1 ? p.a1[grp] : 1.0f;
      const float s2 = p.a2 ? p.a2[grp] : 1.0f;
      const u32x4 r1 = q1[mt][i], r2 = q2[mt][i];
      float v[8] = {lo[0] + b0[0], lo[1] + b0[1], lo[2] + b0[2], lo[3] + b0[3], hi[0] + b1[0], hi[1] + b1[1], hi[2] + b1[2], hi[3] + b1[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] = (v[2 * j] + __uint_as_float(r1[j] << 16)) * s1 + s2 * __uint_as_float(r2[j] << 16);
        v[2 * j + 1] = (v[2 * j + 1] + __uint_as_float(r1[j] & 0xffff0000u)) * s1 + s2 * __uint_as_float(r2[j] & 0xffff0000u);
      }
      const u32x4 pk = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      unsigned short* op = p.out + m * p.ldo + n;
      if (o16) *(u32x4*)op = pk;
      else { *(u32x2*)op = u32x2{pk[0], pk[1]}; *(u32x2*)(op + 4) = u32x2{pk[2], pk[3]}; }
    }
    __builtin_amdgcn_wave_barrier();
  }
#endif
}

}  // namespace

// entry point shared with ffn.hip: hi3d_ffn_geglu() dispatches here unless HI3D_FFN_V=1
int hi3d_ffn2_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                     const void* r1, const void* r2, const float* a1, const float* a2, void* out,
                     int32_t M, int32_t ldx, int32_t ldo, int32_t ldr1, int32_t ldr2, int32_t rows_per_group, void* stream) {
  Ffn2Params p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.b1 = b1; p.W2 = (const char*)w2; p.b2 = b2;
  p.R1 = (const unsigned short*)r1; p.R2 = (const unsigned short*)r2; p.a1 = a1; p.a2 = a2;
  p.out = (unsigned short*)out; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr1 = ldr1; p.ldr2 = ldr2;
  p.rpg = rows_per_group < 1 ? 1 : rows_per_group;
  static bool attr_done[HI3D_MAX_DEVICES] = {};
  if (int rc = hi3d_raise_lds_limit((const void*)ffn2_geglu_c320_kernel, FFN2_LDS, attr_done)) return rc;
  hipLaunchKernelGGL(ffn2_geglu_c320_kernel, dim3((M + GBM - 1) / GBM), dim3(512), FFN2_LDS, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

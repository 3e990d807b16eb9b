// Fused GEGLU feed-forward for the 320-channel level of the Hi3D VideoUNet, gfx950:
//
//     out = ( GEGLU(X W1^T + b1) W2^T + b2 + R1 ) [ * a1 + a2 * R2 ]
//
// (FeedForward of BasicTransformerBlock / VideoTransformerBlock, sgm/modules/attention.py:
// 83-119; the bracket is the AlphaBlender of the temporal block, video_attention.py:290-294.)
// At the 128^2 level the two GEMMs of this block are the largest item of a denoising step and
// the 4C-wide hidden tensor between them (1.3 GB per call, written and read back) is what
// they spend their time on.  Here one block owns 128 rows of X and walks the hidden dimension
// in chunks of 64 columns:
//
//     acc1[128 x 128]  = X[128 x 320] . W1[chunk]^T          (value/gate interleaved, K = 320)
//     hg  [128 x 64]   = value * gelu(gate)  -> bf16 -> LDS, already in A-operand layout
//     acc2[128 x 320] += hg . W2[:, chunk]^T                 (K = 64)
//
// so the hidden tensor never leaves the CU and each 128-row tile is loaded once.  One block of
// 8 waves (2 x 4: wave tiles 64 x 32 packed / 64 x 80 output columns) per CU, two waves per SIMD
// so that one wave's GELU and LDS waits sit under the other's MFMAs.  Two of the five K steps
// of the wave's X rows stay in registers (64), the other three in LDS slabs; W1 streams
// through a 3-slot ring of 16 KiB K-steps running two steps ahead across chunk boundaries, the
// chunk's W2 slab (40 KiB) is requested at the chunk's first step, hg has its own 16 KiB slab:
// 155,648 B of LDS.  Waits are counted vmcnt's that leave the younger streams in flight.
// Operand layouts, swizzles and the swapped-MFMA column order are those of gemm.hip, so the
// packed weights are shared with the unfused path.
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct FfnParams {
  const char* X; const char* W1; const float* b1; const char* W2; const float* b2;
  const unsigned short* R1; const unsigned short* R2; const float* a1; const float* a2;
  unsigned short* out;
  int M, ldx, ldo, ldr1, ldr2, rpg;
};

constexpr int FC = 320;                    // channels
constexpr int FH = 4 * FC;                 // hidden width after GEGLU
constexpr int FBM = 128;                   // rows per block
constexpr int FHC = 64;                    // hidden columns per chunk
constexpr int NCHUNK = FH / FHC;           // 20
constexpr int KX = FC / 64;                // K steps of the first GEMM
constexpr int KX_REG = 2;                  // ... of which the X operand sits in registers (the others: LDS slabs)
constexpr int SLAB = FBM * 128;            // 128 rows x 64 k: 16 KiB (an X K-step, the hg slab)
constexpr int W1_STAGE = 2 * FHC * 128;    // 128 packed rows x 64 k: 16 KiB
constexpr int W1_SLOTS = 3;
constexpr int W2_BYTES = FC * 128;         // 320 rows x 64 k: 40 KiB
constexpr int XS_BYTES = (KX - KX_REG) * SLAB;
constexpr int FFN_LDS = W1_SLOTS * W1_STAGE + W2_BYTES + SLAB + XS_BYTES;   // 155,648 B
constexpr int FNW = 8;                     // waves: 2 (M) x 4 (N)
constexpr int W1_OPS = 16 / FNW, W2_OPS = 40 / FNW;   // LDS-DMA instructions per wave and stage / slab

__global__ __launch_bounds__(FNW * 64, 2) void ffn_geglu_c320_kernel(const FfnParams p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const W1R = smem;                                  // 3-slot ring of W1 K-steps
  char* const W2S = smem + W1_SLOTS * W1_STAGE;            // the chunk's W2 slab
  char* const HG = W2S + W2_BYTES;                         // hg slab
  char* const XS = HG + SLAB;                              // K steps KX_REG.. of the X tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;                       // wave tile: 64 rows x (32 packed | 80 output) columns
  const int fr = lane & 15, fg = lane >> 4;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int m0 = blockIdx.x * FBM;
  constexpr unsigned INV = 0x80000000u;

  // ---- X: the first KX_REG K steps of the wave's 64 rows live in registers, in MFMA B-operand
  // layout (lane (fr, fg): row fr of each 16-row block, k = 32*j + 8*fg .. +7); the rest of the
  // tile goes to LDS slabs in the A-stage layout of gemm.hip
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (long)m0 * p.ldx * 2), 0, 0x7fffffff, 0x00020000);
  bf16x8 xr[2 * KX_REG][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int r = wm * 64 + mt * 16 + fr;
    const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + fg * 16) : INV;
#pragma unroll
    for (int j = 0; j < 2 * KX_REG; ++j) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, j * 64, 0);
      xr[j][mt] = __builtin_bit_cast(bf16x8, t);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 8 + lrow;
    const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + ((lslot ^ ((r >> 1) & 7)) << 4)) : INV;
#pragma unroll
    for (int kt = KX_REG; kt < KX; ++kt)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (LDS_AS void*)(XS + (kt - KX_REG) * SLAB + (w * 2 + i) * 1024), 16, vo, kt * 128, 0, 0);
  }

  // ---- weight loaders (per-lane byte offsets fixed; K step / chunk walk in the scalar offset)
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, 0x7fffffff, 0x00020000);
  unsigned w1_voff[2], w2_voff[5];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (w + FNW * i) * 8 + lrow, jw = j & 31;          // packed W1 row of the chunk; row within its wave tile
    const int fi = (jw >> 3) * 4 + (jw & 3);                       // MFMA row index that reads it
    w1_voff[i] = (unsigned)(j * FC * 2 + ((lslot ^ ((fi >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = (w + FNW * i) * 8 + lrow, jw = j % 80;           // W2 row = output channel
    const int fi = (jw / 20) * 4 + (jw & 3);
    w2_voff[i] = (unsigned)(j * FH * 2 + ((lslot ^ ((fi >> 1) & 7)) << 4));
  }
  // W1 stage s = 5*chunk + k: rows 128*chunk .. +127 of the packed matrix, k columns 64*k .. +63
  auto issue_w1 = [&](int c, int k, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (LDS_AS void*)(W1R + slot * W1_STAGE + (w + FNW * i) * 1024), 16,
                                               c < NCHUNK ? w1_voff[i] : INV, c * (2 * FHC * FC * 2) + k * 128, 0, 0);
  };
  auto issue_w2 = [&](int c) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (LDS_AS void*)(W2S + (w + FNW * i) * 1024), 16, w2_voff[i], c * 128, 0, 0);
  };

  // ---- fragment read offsets
  const int f_sw = (fr >> 1) & 7;
  const int x_off = (wm * 64 + fr) * 128;                                   // + mt * 2048   (X slabs, hg slab)
  const int w1_off = (wn * 32 + (fr >> 2) * 8 + (fr & 3)) * 128;            // + nt * 512
  const int w2_off = (wn * 80 + (fr >> 2) * 20 + (fr & 3)) * 128;           // + nt * 512
  // this lane's packed columns of a chunk: wn*32 + fg*8 + nt*4 .. +3  (= hidden columns wn*16 + fg*4 + nt*2, +1)
  const float* b1p = p.b1 + wn * 32 + fg * 8;
  // its 8 bytes of hg (4 hidden columns) for row fr of every 16-row block
  char* const hg_w = HG + (wm * 64 + fr) * 128 + (((wn * 2 + (fg >> 1)) ^ f_sw) << 4) + (fg & 1) * 8;    // + mt * 2048

  f32x4 acc2[4][5];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_w1(0, 0, 0);
  issue_w1(0, 1, 1);

  int slot = 0;                                  // ring slot of the current stage
  for (int c = 0; c < NCHUNK; ++c) {
    f32x4 acc1[4][2];
    f32x4 b1v[2];                                // consumed after the K loop: the latency hides there
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b1v[nt] = *(const f32x4*)(b1p + c * (2 * FHC) + nt * 4);
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      // Stage s = (c, k) must have landed.  Loads retire in order; younger than stage s are stage
      // s+1 (W1_OPS) plus, at k = 0, this chunk's two bias loads and, at k = 1, the chunk's W2 slab
      // requested in step 0.
      if (k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + 2) : "memory");
      else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + W2_OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS) : "memory");
      __builtin_amdgcn_s_barrier();              // ... for every wave; stage s-1 is read out, its slot is free
      asm volatile("" ::: "memory");
      {                                          // stage s+2 -> the slot of stage s-1 (void past the end: keeps the op count uniform)
        const int k2 = k + 2 < KX ? k + 2 : k + 2 - KX, c2 = k + 2 < KX ? c : c + 1;
        issue_w1(c2, k2, slot == 0 ? 2 : slot - 1);
      }
      if (k == 0) issue_w2(c);                   // every wave is past the previous chunk's second GEMM
      const char* ws = W1R + slot * W1_STAGE + w1_off;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int cx = ((kh * 4 + fg) ^ f_sw) << 4;
        bf16x8 wf[2], xl[4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wf[nt] = *(const bf16x8*)(ws + nt * 512 + cx);
        if (k >= KX_REG) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xl[mt] = *(const bf16x8*)(XS + (k - KX_REG) * SLAB + x_off + mt * 2048 + cx);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], k < KX_REG ? xr[2 * (k < KX_REG ? k : 0) + kh][mt] : xl[mt],
                                                                   (k == 0 && kh == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc1[mt][nt], 0, 0, 0);
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    // GEGLU: value * gelu(gate) -> 4 consecutive hidden columns per lane and row, written into the
    // hg slab in the swizzled A-operand layout.  (Every wave passed the step barriers of this
    // chunk after its previous hg reads, so the slab is free.)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      unsigned int u[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 v = acc1[mt][nt];
        v[0] += b1v[nt][0]; v[1] += b1v[nt][1]; v[2] += b1v[nt][2]; v[3] += b1v[nt][3];
        const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
        u[nt] = pack_bf16x2(v[0] * gl[0], v[1] * gl[1]);
      }
      lds_store_b64_nodrain(hg_w + mt * 2048, u[0], u[1]);      // (a plain store would drain the weight streams: common.h)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // hg visible (no vmcnt drain: the weight streams stay in flight)
    asm volatile("" ::: "memory");
    // acc2 += hg . W2[:, chunk]^T   (the slab was requested at step 0 and the waits of steps 2.. covered it)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int cx = ((kh * 4 + fg) ^ f_sw) << 4;
      bf16x8 xf[4], wf[5];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(HG + x_off + mt * 2048 + cx);
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) wf[nt] = *(const bf16x8*)(W2S + w2_off + nt * 512 + cx);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
          acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc2[mt][nt], 0, 0, 0);
    }
  }

  // ---- epilogue.  Lane (fg, fr) owns row m0 + wm*64 + mt*16 + fr, columns wn*80 + fg*20 + nt*4 .. +3: stored from that
  // layout every load / store instruction is 64 scattered 8-byte requests (~20 k cycles per block, a tenth of its
  // time).  As in the wide GEMM tiles every wave stages its 64 x 80 tile, 16 rows at a time, through a private fp32
  // slab in the (now free) LDS and issues residual loads and stores row-contiguous, 16 bytes per lane; residual
  // slabs are requested one pass ahead.
  constexpr int EW = 80, EROW = EW * 4 + 16, ESLAB = 16 * EROW, ECPR = EW / 8, ECH = (16 * ECPR + 63) / 64;
  static_assert(FNW * ESLAB <= FFN_LDS, "epilogue slabs");
  __syncthreads();                               // every wave is done with the weight / hg slabs
  char* const slab = smem + w * ESLAB;
  const int ecol0 = wn * EW;
  const long erow0 = (long)m0 + wm * 64;
  const int rows_left = p.M - (int)erow0;
  int e_row[ECH], e_col[ECH], e_lds[ECH];
#pragma unroll
  for (int i = 0; i < ECH; ++i) {
    const int c = lane + 64 * i;
    e_row[i] = c / ECPR;
    e_col[i] = (c - e_row[i] * ECPR) * 8;
    e_lds[i] = c < 16 * ECPR ? e_row[i] * EROW + e_col[i] * 4 : -1;
  }
  auto e_ok = [&](int i, int mt) { return e_lds[i] >= 0 && e_row[i] + mt * 16 < rows_left; };
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
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) *(f32x4*)(slab + fr * EROW + (fg * 20 + nt * 4) * 4) = acc2[mt][nt];
    if (mt + 1 < 4) {
#pragma unroll
      for (int i = 0; i < ECH; ++i) { q1[(mt + 1) & 1][i] = fetch(p.R1, p.ldr1, mt + 1, i); q2[(mt + 1) & 1][i] = fetch(p.R2, p.ldr2, mt + 1, i); }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ECH; ++i) {
      if (!e_ok(i, mt)) continue;
      const long m = erow0 + mt * 16 + e_row[i];
      const int n = ecol0 + e_col[i];
      const f32x4 lo = *(const f32x4*)(slab + e_lds[i]), hi = *(const f32x4*)(slab + e_lds[i] + 16);
      const f32x4 b0 = *(const f32x4*)(p.b2 + n), b1 = *(const f32x4*)(p.b2 + n + 4);
      const int grp = (p.a1 || p.a2) ? (int)(m / p.rpg) : 0;
      const float s1 = p.a1 ? p.a1[grp] : 1.0f;
      const float s2 = p.a2 ? p.a2[grp] : 1.0f;
      const u32x4 r1 = q1[mt & 1][i], r2 = q2[mt & 1][i];
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

int hi3d_ffn2_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                     const void* r1, const void* r2, const float* a1, const float* a2, void* out,
                     int32_t M, int32_t ldx, int32_t ldo, int32_t ldr1, int32_t ldr2, int32_t rows_per_group, void* stream);

extern "C" int hi3d_ffn_geglu(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                              const void* r1, const void* r2, const float* a1, const float* a2, void* out,
                              int32_t M, int32_t C, int32_t ldx, int32_t ldo, int32_t ldr1, int32_t ldr2,
                              int32_t rows_per_group, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !out) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu: null pointer");
  if (M <= 0) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu: non-positive M");
  if (C != FC) HI3D_FAIL(HI3D_ESHAPE, "ffn_geglu: only the 320-channel level is fused (use two GEMMs otherwise)");
  if (ldx < C || ldo < C || (ldx % 8) || (ldo % 4)) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu: bad ldx / ldo");
  if ((r1 && (ldr1 < C || ldr1 % 4)) || (r2 && (ldr2 < C || ldr2 % 4))) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu: bad residual leading dim");
  if ((a1 || a2) && rows_per_group < 1) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu: rows_per_group < 1");
  if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2) & 15) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu: x / w1 / w2 not 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)r1 | (uintptr_t)r2) & 7) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu: out / residuals not 8-byte aligned");
  if (((uintptr_t)b1 | (uintptr_t)b2) & 15) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu: biases not 16-byte aligned");
  // second form (ffn2.hip: 32 rows per wave with X in registers, ping-pong phases, software-pipelined GELU) unless HI3D_FFN_V=1
  {
    const char* e = getenv("HI3D_FFN_V");
    if (!e || atoi(e) != 1)
      return hi3d_ffn2_launch(x, w1, b1, w2, b2, r1, r2, a1, a2, out, M, ldx, ldo, ldr1, ldr2, rows_per_group, stream);
  }
  FfnParams p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.b1 = b1; p.W2 = (const char*)w2; p.b2 = b2;
  p.R1 = (const unsigned short*)r1; p.R2 = (const unsigned short*)r2; p.a1 = a1; p.a2 = a2;
  p.out = (unsigned short*)out; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr1 = ldr1; p.ldr2 = ldr2;
  p.rpg = rows_per_group < 1 ? 1 : rows_per_group;
  static bool attr_done[HI3D_MAX_DEVICES] = {};
  if (int rc = hi3d_raise_lds_limit((const void*)ffn_geglu_c320_kernel, FFN_LDS, attr_done)) return rc;
  hipLaunchKernelGGL(ffn_geglu_c320_kernel, dim3((M + FBM - 1) / FBM), dim3(FNW * 64), FFN_LDS, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

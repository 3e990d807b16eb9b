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
// 4 waves (64 x 64 / 64 x 160 wave tiles) per CU and one wave per SIMD, which buys 512
// registers per lane: the wave's slice of X (160), both accumulators (64 + 160) and the
// operand fragments all live there, and LDS (147,456 B) belongs to the weight streams -- W1
// through a 3-slot ring of 16 KiB K-steps running two steps ahead across chunk boundaries,
// W2 double-buffered a whole chunk ahead, plus the 16 KiB hg slab.  With a single wave per
// SIMD nothing hides a load, so every wait is a counted vmcnt that leaves the younger streams
// in flight.  Operand layouts, swizzles and the swapped-MFMA column order are those of
// gemm.hip, so the packed weights are shared with the unfused path.
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
constexpr int SLAB = FBM * 128;            // the hg slab: 128 rows x 64 k, 16 KiB
constexpr int W1_STAGE = 2 * FHC * 128;    // 128 packed rows x 64 k: 16 KiB
constexpr int W1_SLOTS = 3;
constexpr int W2_BYTES = FC * 128;         // 320 rows x 64 k: 40 KiB
constexpr int FFN_LDS = W1_SLOTS * W1_STAGE + 2 * W2_BYTES + SLAB;   // 147,456 B
constexpr int W1_OPS = 4, W2_OPS = 10, B1_OPS = 4;   // memory instructions per wave

__global__ __launch_bounds__(256, 1) void ffn_geglu_c320_kernel(const FfnParams p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const W1R = smem;                                  // 3-slot ring of W1 K-steps
  char* const W2S = smem + W1_SLOTS * W1_STAGE;            // two W2 slabs (chunk parity)
  char* const HG = W2S + 2 * W2_BYTES;                     // hg slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int m0 = blockIdx.x * FBM;
  constexpr unsigned INV = 0x80000000u;

  // ---- the wave's 64 x 320 slice of X lives in registers for the whole kernel, already in
  // MFMA B-operand layout (lane (fr, fg): row fr of each 16-row block, k = 32*j + 8*fg .. +7):
  // 160 VGPRs, which one wave per SIMD can afford and which leaves LDS to the weight streams
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (long)m0 * p.ldx * 2), 0, 0x7fffffff, 0x00020000);
  bf16x8 xr[2 * KX][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int r = wm * 64 + mt * 16 + fr;
    const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + fg * 16) : INV;
#pragma unroll
    for (int j = 0; j < 2 * KX; ++j) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, j * 64, 0);
      xr[j][mt] = __builtin_bit_cast(bf16x8, t);
    }
  }

  // ---- weight loaders (per-lane byte offsets fixed; K step / chunk walk in the scalar offset)
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, 0x7fffffff, 0x00020000);
  unsigned w1_voff[4], w2_voff[10];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (w + 4 * i) * 8 + lrow, jw = j & 63;            // packed W1 row of the chunk
    const int fi = (jw >> 4) * 4 + (jw & 3);                       // MFMA row index that reads it
    w1_voff[i] = (unsigned)(j * FC * 2 + ((lslot ^ ((fi >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int j = (w + 4 * i) * 8 + lrow, jw = j % 160;            // W2 row = output channel
    const int fi = (jw / 40) * 4 + (jw & 3);
    w2_voff[i] = (unsigned)(j * FH * 2 + ((lslot ^ ((fi >> 1) & 7)) << 4));
  }
  // W1 K-step s = 5*chunk + k: rows 128*chunk .. +127 of the packed matrix, k columns 64*k .. +63
  auto issue_w1 = [&](int c, int k, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (LDS_AS void*)(W1R + slot * W1_STAGE + (w + 4 * i) * 1024), 16, w1_voff[i],
                                               c * (2 * FHC * FC * 2) + k * 128, 0, 0);
  };
  auto issue_w2 = [&](int c) {
#pragma unroll
    for (int i = 0; i < 10; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (LDS_AS void*)(W2S + (c & 1) * W2_BYTES + (w + 4 * i) * 1024), 16, w2_voff[i], c * 128, 0, 0);
  };

  // ---- fragment read offsets
  const int f_sw = (fr >> 1) & 7;
  const int x_off = (wm * 64 + fr) * 128;                                   // + mt * 2048   (hg slab)
  const int w1_off = (wn * 64 + (fr >> 2) * 16 + (fr & 3)) * 128;           // + nt * 512
  const int w2_off = (wn * 160 + (fr >> 2) * 40 + (fr & 3)) * 128;          // + nt * 512
  // this lane's packed columns of a chunk: wn*64 + fg*16 + nt*4 .. +3  (= hidden columns wn*32 + fg*8 + nt*2, +1)
  const float* b1p = p.b1 + wn * 64 + fg * 16;
  // its 16-byte hg chunk (8 hidden columns) for row fr of every 16-row block
  char* const hg_w = HG + (wm * 64 + fr) * 128 + (((wn * 4 + fg) ^ f_sw) << 4);    // + mt * 2048

  f32x4 acc2[4][10];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 b1v[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b1v[nt] = *(const f32x4*)(b1p + nt * 4);
  issue_w2(0);
  issue_w1(0, 0, 0);
  issue_w1(0, 1, 1);

  int slot = 0;                                  // ring slot of the current K step
  for (int c = 0; c < NCHUNK; ++c) {
    f32x4 acc1[4][4];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      // Stage s = (c, k) must have landed.  Loads retire in order; younger than stage s are stage
      // s+1 (4 ops) and whatever the previous step issued after it: the next chunk's W2 slab
      // (after k = 0) or its bias registers (after k = 1).
      if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + W2_OPS) : "memory");
      else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS + B1_OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1_OPS) : "memory");
      __builtin_amdgcn_s_barrier();              // ... for every wave; slot (s-1)%3 is free again
      asm volatile("" ::: "memory");
      {                                          // stage s+2 -> the slot stage s-1 used
        const int k2 = k + 2 < KX ? k + 2 : k + 2 - KX, c2 = k + 2 < KX ? c : c + 1;
        const int slot2 = slot == 0 ? 2 : slot - 1;
        if (c2 < NCHUNK) issue_w1(c2, k2, slot2);
        else {                                   // keep the per-step op count uniform for the vmcnt arithmetic
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (LDS_AS void*)(W1R + slot2 * W1_STAGE + (w + 4 * i) * 1024), 16, INV, 0, 0, 0);
        }
      }
      if (k == 0) {                              // next chunk's W2 slab into the other buffer
        if (c + 1 < NCHUNK) issue_w2(c + 1);
        else {
#pragma unroll
          for (int i = 0; i < 10; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (LDS_AS void*)(W2S + ((c + 1) & 1) * W2_BYTES + (w + 4 * i) * 1024), 16, INV, 0, 0, 0);
        }
      }
      if (k == 1) {                              // next chunk's bias columns (b1v was consumed by step 0; the last chunk re-reads its own)
        const int cn = c + 1 < NCHUNK ? c + 1 : c;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b1v[nt] = *(const f32x4*)(b1p + cn * (2 * FHC) + nt * 4);
      }
      const char* ws = W1R + slot * W1_STAGE;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int cx = ((kh * 4 + fg) ^ f_sw) << 4;
        bf16x8 wf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(ws + w1_off + nt * 512 + cx);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xr[2 * k + kh][mt], (k == 0 && kh == 0) ? b1v[nt] : acc1[mt][nt], 0, 0, 0);
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    // GEGLU: value * gelu(gate) -> 8 consecutive hidden columns per lane and row = one 16-byte
    // chunk of the hg slab, written in the swizzled A-operand layout.  (Every wave passed the
    // K-step barriers of this chunk after its previous hg reads, so the slab is free.)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      unsigned int u[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 v = acc1[mt][nt];
        const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
        u[nt] = pack_bf16x2(v[0] * gl[0], v[1] * gl[1]);
      }
      *(uint4*)(hg_w + mt * 2048) = make_uint4(u[0], u[1], u[2], u[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // hg visible (no vmcnt drain: the weight streams stay in flight)
    asm volatile("" ::: "memory");
    // acc2 += hg . W2[:, chunk]^T   (the slab was requested a whole chunk ago and every later
    // wait covered it)
    const char* w2s = W2S + (c & 1) * W2_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int cx = ((kh * 4 + fg) ^ f_sw) << 4;
      bf16x8 xf[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(HG + x_off + mt * 2048 + cx);
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        bf16x8 wf[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) wf[nt] = *(const bf16x8*)(w2s + w2_off + (nh * 5 + nt) * 512 + cx);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 5; ++nt)
            acc2[mt][nh * 5 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc2[mt][nh * 5 + nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane (fg, fr) owns row m0 + wm*64 + mt*16 + fr, columns wn*160 + fg*40 + nt*4 .. +3
  const int nb = wn * 160 + fg * 40;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + fr;
    if (m >= p.M) continue;
    const int grp = (p.a1 || p.a2) ? m / p.rpg : 0;
    const float s1 = p.a1 ? p.a1[grp] : 1.0f;
    const float s2 = p.a2 ? p.a2[grp] : 1.0f;
    uint2 r1[10], r2[10];
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
      r1[nt] = p.R1 ? *(const uint2*)(p.R1 + (long)m * p.ldr1 + nb + nt * 4) : make_uint2(0, 0);
      r2[nt] = p.R2 ? *(const uint2*)(p.R2 + (long)m * p.ldr2 + nb + nt * 4) : make_uint2(0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
      const f32x4 b = *(const f32x4*)(p.b2 + nb + nt * 4);
      float v[4];
      v[0] = (acc2[mt][nt][0] + b[0] + bf16_to_f32(r1[nt].x & 0xffff)) * s1 + s2 * bf16_to_f32(r2[nt].x & 0xffff);
      v[1] = (acc2[mt][nt][1] + b[1] + bf16_to_f32(r1[nt].x >> 16)) * s1 + s2 * bf16_to_f32(r2[nt].x >> 16);
      v[2] = (acc2[mt][nt][2] + b[2] + bf16_to_f32(r1[nt].y & 0xffff)) * s1 + s2 * bf16_to_f32(r2[nt].y & 0xffff);
      v[3] = (acc2[mt][nt][3] + b[3] + bf16_to_f32(r1[nt].y >> 16)) * s1 + s2 * bf16_to_f32(r2[nt].y >> 16);
      *(uint2*)(p.out + (long)m * p.ldo + nb + nt * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
#endif
}

}  // namespace

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
  FfnParams p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.b1 = b1; p.W2 = (const char*)w2; p.b2 = b2;
  p.R1 = (const unsigned short*)r1; p.R2 = (const unsigned short*)r2; p.a1 = a1; p.a2 = a2;
  p.out = (unsigned short*)out; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr1 = ldr1; p.ldr2 = ldr2;
  p.rpg = rows_per_group < 1 ? 1 : rows_per_group;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_geglu_c320_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS);
    if (e != hipSuccess) { hi3d_set_error(hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  hipLaunchKernelGGL(ffn_geglu_c320_kernel, dim3((M + FBM - 1) / FBM), dim3(256), FFN_LDS, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

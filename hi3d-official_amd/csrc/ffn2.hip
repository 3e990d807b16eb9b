// Fused GEGLU feed-forward for the 320-channel level, second form (round 2): the same computation as ffn.hip
//
//     out = ( GEGLU(X W1^T + b1) W2^T + b2 + R1 ) [ * a1 + a2 * R2 ]
//
// (FeedForward of BasicTransformerBlock / VideoTransformerBlock, sgm/modules/attention.py:83-119; the bracket is the
// AlphaBlender of the temporal block, sgm/modules/video_attention.py:290-294), restructured around what the round-2
// measurements showed: the first form runs its MFMAs at 37 % utilisation because all 8 waves move in lock step through
// 16-MFMA K steps separated by barriers -- both waves of a SIMD wait for LDS together, then both compute, then both do
// their GELU.  Here
//   * a wave owns 32 rows x 160 output columns and keeps 9/10 of its X rows in registers (72; the last 32 channels sit
//     in an 8 KiB LDS slab): the first GEMM reads almost only weights from LDS and the 48 KiB of X slabs are gone;
//   * the loop runs over HALF chunks (32 hidden columns = 64 packed W1 rows) in three phases of 20 MFMAs,
//         [load segment: first fragments, a slice of GELU, LDS-DMA requests and their waits] s_barrier
//         [compute segment: 20 MFMAs + the rolling fragment reads] s_barrier,
//     two for the first GEMM (K halves) and one for the down-projection of the PREVIOUS half chunk; every code site
//     exists once, so the accumulators stay in place (two sites for the second GEMM cost a rotating copy of the
//     80-register output tile and spilled);
//   * waves 4..7 (rows 64..127: their hg rows are their own) run one barrier behind waves 0..3, so on every SIMD one
//     wave is in a compute segment while the other does its loads, GELU and DMA requests;
//   * the first GEMM's accumulators start from the bias; hg is stored with an LDS store the compiler does not drain
//     the weight streams for (common.h: lds_store_b64_nodrain -- the first form waited vmcnt(0) in every chunk).
// Measured (profiles/r02d_*): 1.50 ms at M = 524288 (860 TFLOP/s) against 1.54 ms for the first form; the ablations
// (r02d_ffn2_ablation_v1.log) put the limit at the non-MFMA issue work per SIMD -- GELU ~700, DMA requests ~550,
// fragment reads ~600 cycles per wave and half chunk beside 966 cycles of MFMA -- and at the exposed prologue /
// epilogue of a kernel that fits one block per CU (~15 %).
// LDS: two W1 slots of 40 KiB (five 8 KiB K slabs each, the swizzled stage layout of gemm.hip) + two W2 slots of 20 KiB
// (320 rows x 32 hidden columns) + hg 16 KiB + the bias vector 10 KiB + the X slab 8 KiB = 154 KiB.  All waits are
// counted vmcnt's, all barriers raw.  Operand layouts (packed GEGLU rows, K-major W2) are those of ffn.hip / gemm.hip:
// the packed weights are shared.
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
  // LN = true (hi3d_ffn_geglu_ln): X is the RAW residual stream; the kernel normalises the rows itself
  const float* ln_g; const float* ln_b; const float* av; float ln_eps; int rpg_av;
};

constexpr int GC = 320;                    // channels
constexpr int GH = 4 * GC;                 // hidden width after GEGLU
constexpr int GBM = 128;                   // rows per block
constexpr int GNCH = GH / 64;              // 20 chunks of 64 hidden columns
constexpr int W1S = 5 * 8192;              // one W1 stage: 64 packed rows x 320 k = five [64][64] K slabs
constexpr int W2H = 320 * 64;              // half of a chunk's W2 slab: 320 rows x 32 k, 64-byte rows
constexpr int HGB = GBM * 128;             // hg: 128 rows x 64 hidden
constexpr int B1B = 8 * GC * 4;            // the packed first-layer bias, fp32 (read in every GELU slice: no VMEM loads inside the loop)
constexpr int X9B = GBM * 64;              // the last 32 channels of the X tile (k = 288..319): the 80 registers of a whole X tile, the
                                           // output tile (80) and the rest do not fit in 256 -- 9 of the 10 half K steps stay in registers
constexpr int G1D = 2;                     // first GEMM: fragment pairs in flight (of the 5 per phase; 3 / 5 below measured the same, more registers)
constexpr int G2D = 3;                     // second GEMM: W2 fragments in flight (of the 10 per phase)
constexpr int FFN2_LDS = 2 * W1S + 2 * W2H + HGB + B1B + X9B;      // 157,696 B

// LN: the LayerNorm in front of the feed-forward (x = ff(norm3(x)) + x, attention.py:570; x = ff_in(norm_in(x)) + x and
// x = ff(norm3(x)) + x, video_attention.py:119-133) runs in the prologue on the X registers: a lane holds 80 of its row's 320
// channels (the 4 lanes fr + 16 fg share a row), so the statistics are two register passes + two cross-lane adds, and the
// normalised bf16 values replace the raw ones in place -- layernorm_packed_kernel's arithmetic (fp32, mean first, then the
// centred sum of squares), its rounding points, and the optional per-row-group vector (the frame-position embedding,
// video_attention.py:276-283: statistics on the UNROUNDED sum, the residual takes bf16(x + vector)).  Removes the norm's
// launch, its read of x and its write of the normalised tensor (and, with the vector, the write of the sum).
template <int LNM>       // 0: X is already normalised; 1: LayerNorm in the prologue; 2: ... of x + per-row-group vector
__global__ __launch_bounds__(512, 2) void ffn2_geglu_c320_kernel(const Ffn2Params p) {
  constexpr bool LN = LNM != 0, AV = LNM == 2;
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sW1 = smem;                                   // stage A at 0, stage B at W1S
  char* const sW2 = smem + 2 * W1S;                         // k 0..31 half at 0, k 32..63 half at W2H
  char* const sHG = sW2 + 2 * W2H;
  float* const sB1 = (float*)(sHG + HGB);
  char* const sX9 = (char*)sB1 + B1B;                       // [mt][32-row block][lane] x 16 B: fragments of k = 288..319, shared by the two column halves

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
  bf16x8 xr[9][2];
  u32x4 mine[10];                                           // (LN: the raw 16-row block this wave normalises, all 320 channels)
  const int x9_off = wmg * 1024 + lane * 16;                // + mt * 4096
  if (!LN) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int r = wmg * 32 + mt * 16 + fr;
      const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + fg * 16) : INV;
#pragma unroll
      for (int kk = 0; kk < 9; ++kk) xr[kk][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, kk * 64, 0));
      if (wn == 0) *(u32x4*)(sX9 + mt * 4096 + x9_off) = __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, 9 * 64, 0);
    }
  } else {
    // the two waves of a 32-row block (column halves wn = 0, 1) each fetch and normalise ONE of its 16-row blocks (mt = wn)
    // and hand the result to each other through LDS
    const int r = wmg * 32 + wn * 16 + fr;
    const unsigned vo = (m0 + r < p.M) ? (unsigned)(r * p.ldx * 2 + fg * 16) : INV;
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) mine[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, kk * 64, 0);
  }
  if (LN) {      // gamma | beta into the hg slab (free until the first GELU; read after the prologue barrier)
    float* const sG = (float*)sHG;
    if (tid < GC / 4) *(f32x4*)(sG + tid * 4) = *(const f32x4*)(p.ln_g + tid * 4);
    else if (tid < GC / 2) *(f32x4*)(sG + GC + (tid - GC / 4) * 4) = *(const f32x4*)(p.ln_b + (tid - GC / 4) * 4);
    else if (AV && tid < GC) {
      // ... and the (at most two: rows_per_group >= 128, checked by the host) vectors the block's 128 rows belong to
      const int t = tid - GC / 2, which = t / (GC / 4), c4 = t % (GC / 4);
      const int g0 = m0 / p.rpg_av, glast = (p.M - 1) / p.rpg_av;
      const int g = g0 + which < glast ? g0 + which : glast;
      *(f32x4*)(sG + (2 + which) * GC + c4 * 4) = *(const f32x4*)(p.av + (long)g * GC + c4 * 4);
    }
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
  // W2 rows are 64 B (4 chunks of 16 B): the 16 lanes of a ds_read_b128 lane group hold two k groups fg and, per fg, two
  // row quads q = fr >> 2 -- {0,3} or {1,2} -- whose rows sit 40 apart (a multiple of 4 rows = 256 B: the same banks).
  // Chunk g of a row of quad q is stored at position g ^ ((4 - q) & 3): every lane group then covers 16 distinct
  // (row mod 4, position) pairs.  q differs between a wave's three pieces: three lane offsets.
  unsigned w2_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = (w + 8 * i) * 16 + (lane >> 2), q = (r % 160) / 40;
    w2_voff[i] = (unsigned)(r * GH * 2 + (((lane & 3) ^ ((4 - q) & 3)) << 4));
  }
  auto issue_w1 = [&](int hc, int slot) {                   // packed rows 64 hc .. +63 (half chunk hc) -> W1 slot
    const int soff = hc * 64 * GC * 2;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (LDS_AS void*)(sW1 + slot * W1S + i * 8192 + w * 1024), 16, w1_voff, soff + i * 128, 0, 0);
  };
  auto issue_w2 = [&](int hc, int slot) {                   // hidden columns 32 hc .. +31 of every W2 row -> W2 slot
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (w + 8 * i < 20)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (LDS_AS void*)(sW2 + slot * W2H + (w + 8 * i) * 1024), 16, w2_voff[i], hc * 64, 0, 0);
  };

  // ---- fragment read offsets
  const int f_sw = (fr >> 1) & 7;
  const int w1_off = (wn * 32 + (fr >> 2) * 8 + (fr & 3)) * 128;                  // + nt * 512, + slab * 8192
  const int w2_off = (wn * 160 + (fr >> 2) * 40 + (fr & 3)) * 64 + ((fg ^ ((4 - (fr >> 2)) & 3)) << 4);   // + nt * 256   (64-byte rows)
  const int hg_off = (wmg * 32 + fr) * 128;                                        // + mt * 2048, + swizzled chunk
  // this lane's packed columns of a stage: wn*32 + fg*8 + nt*4 .. +3  (= stage-local hidden columns wn*16 + fg*4 + nt*2, +1)
  const float* b1p = sB1 + wn * 32 + fg * 8;
  // its 8 bytes of hg (4 hidden columns) for row fr of a 16-row block: + stage * 64 bytes (4 chunks of 16 B) handled by the xor below
  auto hg_wptr = [&](int stage, int mt) {
    return sHG + hg_off + mt * 2048 + ((((stage * 4 + wn * 2 + (fg >> 1))) ^ f_sw) << 4) + (fg & 1) * 8;
  };

  f32x4 acc2[2][10];                               // (zeroed right in front of the loop: the LayerNorm prologue needs the registers)
  f32x4 acc1[2][2];

  // value * gelu(gate) of one 16-row block of a stage's accumulators -> 4 hidden columns per lane -> hg
  auto gelu_store = [&](const f32x4 (&acc)[2][2], int stage, int mt) {
    unsigned int u[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 v = acc[mt][nt];
      const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
      u[nt] = pack_bf16x2(v[0] * gl[0], v[1] * gl[1]);
    }
    lds_store_b64_nodrain(hg_wptr(stage, mt), u[0], u[1]);      // (a plain store would drain the weight streams: common.h)
  };
  // first GEMM, one phase: 5 of the 10 half K steps of a stage.  Two of the five fragment pairs are read in the load
  // segment; every later pair is requested into the registers of the pair just issued to the matrix core (4 MFMAs ahead
  // of its use) -- all five at once do not fit beside X and the output tile.
  auto g1_frag = [&](bf16x8 (&f)[2], int stage, int kk) {
    const char* s = sW1 + stage * W1S + (kk >> 1) * 8192 + w1_off + ((((kk & 1) * 4 + fg) ^ f_sw) << 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) f[nt] = *(const bf16x8*)(s + nt * 512);
  };
  auto g1_read = [&](bf16x8 (&wf)[G1D][2], int stage, int half) {
#pragma unroll
    for (int q = 0; q < G1D; ++q) g1_frag(wf[q], stage, half * 5 + q);
  };
  auto g1_step = [&](const bf16x8 (&f)[2], f32x4 (&acc)[2][2], int kk, const bf16x8 (&x9)[2], const f32x4 (&bias)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[nt], kk < 9 ? xr[kk < 9 ? kk : 0][mt] : x9[mt],
                                                                kk == 0 ? bias[nt] : acc[mt][nt], 0, 0, 0);
  };
  auto g1_mfma = [&](bf16x8 (&wf)[G1D][2], f32x4 (&acc)[2][2], int stage, int half, const f32x4 (&bias)[2]) {
    const int k0 = half * 5;
    bf16x8 x9[2];
    if (half == 1) {                               // the X fragments of the last half K step (requested three steps ahead of their use)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) x9[mt] = *(const bf16x8*)(sX9 + mt * 4096 + x9_off);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      g1_step(wf[q % G1D], acc, k0 + q, x9, bias);
      if (q + G1D < 5) {                           // this pair's registers take the pair G1D steps ahead
        __builtin_amdgcn_sched_barrier(0);
        g1_frag(wf[q % G1D], stage, k0 + q + G1D);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the late fragment reads are retired before the closing barrier)
  };
  // second GEMM, one phase: K half `kh` (32 hidden columns) of the chunk whose hg half is ready.  The ten W2 fragments
  // do not fit beside X and the output tile: three are read in the load segment, the others into the registers of the
  // fragment just issued, three column blocks (6 MFMAs) ahead of their use.
  auto g2_read = [&](bf16x8 (&xf)[2], bf16x8 (&wf)[G2D], int kh) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xf[mt] = *(const bf16x8*)(sHG + hg_off + mt * 2048 + (((kh * 4 + fg) ^ f_sw) << 4));
#pragma unroll
    for (int nt = 0; nt < G2D; ++nt) wf[nt] = *(const bf16x8*)(sW2 + kh * W2H + w2_off + nt * 256);
  };
  auto g2_mfma = [&](const bf16x8 (&xf)[2], bf16x8 (&wf)[G2D], int kh) {
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt % G2D], xf[mt], acc2[mt][nt], 0, 0, 0);
      if (nt + G2D < 10) {                         // this fragment's registers take the one G2D column blocks ahead
        __builtin_amdgcn_sched_barrier(0);
        wf[nt % G2D] = *(const bf16x8*)(sW2 + kh * W2H + w2_off + (nt + G2D) * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the late fragment reads are retired before the closing barrier)
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

  // ---- prologue: the bias vector and the first W1 stage
  for (int i = tid; i < 8 * GC / 4; i += 512) *(f32x4*)(sB1 + i * 4) = *(const f32x4*)(p.b1 + i * 4);
  issue_w1(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the bias copy and the X slab are plain LDS stores)
  __builtin_amdgcn_s_barrier();
  if (LN) {
    // ---- LayerNorm of this wave's 16 rows (three passes over the packed registers: the 80 fp32 values of a lane's row share
    // are never held at once), normalised bf16 fragments -> the exchange slab (the second W1 slot and the W2 ring: idle until
    // the loop's first requests, which come after the barriers below) / the X slab; then both waves of the pair read the whole
    // 32-row tile from there.
    // Every step ends in an empty volatile asm on its result: volatile asms keep their order, so a step's loads / unpacking
    // cannot be hoisted over the previous step and its arithmetic cannot be deferred past the next one -- left to itself the
    // scheduler read all 80 gamma / beta values first and spilled ~100-280 registers.
    const float* const sG = (const float*)sHG;
    char* const xchg = sW1 + W1S;                            // [wmg][mt][kk < 9][lane] x 16 B = 72 KiB of the 80 free
    const int r = wmg * 32 + wn * 16 + fr;
    const int row = (m0 + r < p.M) ? m0 + r : 0;
    const int avo = AV ? ((2 + (row / p.rpg_av != m0 / p.rpg_av)) * GC + fg * 8) * 4 : 0;     // byte offset into the staged vectors
    auto vals = [&](int kk, float (&f)[8]) {
      u32x4 q = mine[kk];
      int ao = avo;
      asm volatile("" : "+v"(q), "+v"(ao));
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(q[j] << 16); f[2 * j + 1] = __uint_as_float(q[j] & 0xffff0000u); }
      if (AV) {
        const char* ap = (const char*)sG + ao + kk * 128;
        const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] += a0[j]; f[4 + j] += a1[j]; }
      }
    };
    float sm = 0.f;
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) {
      float f[8]; vals(kk, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) sm += f[j];
      asm volatile("" : "+v"(sm));
    }
    sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / (float)GC);
    float ss = 0.f;
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) {
      float f[8]; vals(kk, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; ss += d * d; }
      asm volatile("" : "+v"(ss));
    }
    ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.0f / (float)GC) + p.ln_eps);
    char* const my_x = xchg + (wmg * 2 + wn) * 9 * 1024 + lane * 16;
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) {
      int go = (kk * 32 + fg * 8) * 4;
      asm volatile("" : "+v"(go));
      const char* gp = (const char*)sG + go;
      const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 16), b0 = *(const f32x4*)(gp + GC * 4), b1 = *(const f32x4*)(gp + GC * 4 + 16);
      float f[8]; vals(kk, f);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (f[j] - mean) * rstd * g0[j] + b0[j];
        o[4 + j] = (f[4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      u32x4 pk = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      asm volatile("" : "+v"(pk));
      // (plain LDS stores: no LDS-DMA is outstanding here -- the first W1 stage was waited for above)
      if (kk < 9) *(u32x4*)(my_x + kk * 1024) = pk;
      else *(u32x4*)(sX9 + (wn ? 4096 : 0) + x9_off) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // both 16-row blocks of every pair are in LDS
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kk = 0; kk < 9; ++kk) xr[kk][mt] = *(const bf16x8*)(xchg + ((wmg * 2 + mt) * 9 + kk) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // the exchange slab and gamma / beta are free again (the loop's first LDS-DMA
  }                                                // into them and the first hg store come later)
  if (late) __builtin_amdgcn_s_barrier();          // the stagger

  // One loop iteration = one HALF chunk hc (32 hidden columns = 64 packed W1 rows; slot s = hc & 1 of every ring).
  // Compute segments hold MFMAs and their rolling fragment reads only; everything else sits in the load segments, under
  // the other group's MFMAs:
  //     Ph1  load: bias, fragments
  //          compute: W1(hc) k 0-159 x X -> acc1 (from the bias)
  //     Ph2  load: wait W2(hc-1); request W1(hc+1) -> W1 slot s^1; fragments
  //          compute: W1(hc) k 160-319 x X -> acc1
  //     Ph3  load: wait W1(hc+1); request W2(hc) -> W2 slot s; GELU(hc) -> hg half s; fragments of hg half s^1 / W2(hc-1)
  //          compute: hg half s^1 x W2(hc-1) -> acc2 (the previous half chunk)
  // Ring protocol.  A phase reads LDS in its load segment AND (the rolling fragments) in its compute segment, and the
  // late group runs one barrier behind: the last read of phase k happens in the slot in which the early group already
  // executes the load segment of phase k+1.  So the request that overwrites what phase k read is issued in the load
  // segment of phase k+2 at the earliest, and a wave waits for its own pieces in the load segment one phase before the
  // first reader (the other group reads one barrier later).  hg rows belong to one group (both column halves of a
  // 32-row block are in the same half of the block), so its hazards are plain program order plus any barrier.
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NHC = 2 * GNCH;
  for (int hc = 0; hc <= NHC; ++hc) {
    const bool cur = hc < NHC, prev = hc > 0;
    const int sl = hc & 1;
    // ---- Ph1
    {
      bf16x8 wf[G1D][2];
      f32x4 bias[2];
      if (cur) {
        g1_read(wf, sl, 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bias[nt] = *(const f32x4*)(b1p + hc * 64 + nt * 4);
      }
      seg_end();
      if (cur) g1_mfma(wf, acc1, sl, 0, bias);
      cseg_end();
    }
    if (cur) {
      // ---- Ph2
      bf16x8 wf[G1D][2];
      f32x4 bias[2];                               // (unused: k > 0)
      // W2(hc-1) (requested in the previous Ph3, read in this Ph3) has landed: nothing younger is in flight
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (hc + 1 < NHC) issue_w1(hc + 1, sl ^ 1);
      g1_read(wf, sl, 1);
      seg_end();
      g1_mfma(wf, acc1, sl, 1, bias);
      cseg_end();
    }
    // ---- Ph3
    {
      bf16x8 xf[2], wf[G2D];
      if (prev) g2_read(xf, wf, sl ^ 1);
      if (cur) {
        // W1(hc+1) (requested in Ph2, read from the next Ph1 on) has landed: nothing younger is in flight.  The last
        // half chunk's W2 has its reader in the very next phase: requested and waited for here.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_w2(hc, sl);
        if (hc + 1 == NHC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gelu_store(acc1, sl, 0);                   // (both row halves here: four independent dependency chains instead of two --
        gelu_store(acc1, sl, 1);                   //  the GELU is latency-bound, ~11 cycles per instruction with two)
      }
      seg_end();
      if (prev) g2_mfma(xf, wf, sl ^ 1);
      cseg_end();
    }
  }
  // ---- epilogue: every wave stages its 32 x 160 tile, 16 rows at a time, through a private fp32 slab and issues its
  // residual loads / stores row-contiguous, 16 bytes per lane (as the wide GEMM tiles do)
  constexpr int EW = 160, EROW = EW * 4 + 16, ESLAB = 16 * EROW, ECPR = EW / 8, ECH = (16 * ECPR + 63) / 64;   // 5 chunks per lane and pass
  static_assert(8 * ESLAB <= FFN2_LDS, "epilogue slabs");
  char* const slab = smem + w * ESLAB;
  const int ecol0 = wn * EW;
  const long erow0 = (long)m0 + wmg * 32;
  const int rows_left = p.M - (int)erow0;
  auto e_row_ = [&](int i) { return (lane + 64 * i) / ECPR; };          // (recomputed at every use: the loop above left no registers to park them in)
  auto e_col_ = [&](int i) { return ((lane + 64 * i) % ECPR) * 8; };
  auto e_ok = [&](int i, int mt) { return e_row_(i) + mt * 16 < rows_left; };
  auto fetch = [&](const unsigned short* R, int ldr, int mt, int i) -> u32x4 {
    if (!R || !e_ok(i, mt)) return u32x4{0u, 0u, 0u, 0u};
    const unsigned short* rp = R + (erow0 + mt * 16 + e_row_(i)) * ldr + ecol0 + e_col_(i);
    if ((ldr & 7) == 0 && (((uintptr_t)R) & 15) == 0) return *(const u32x4*)rp;
    const u32x2 lo = *(const u32x2*)rp, hi = *(const u32x2*)(rp + 4);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  // the residual tiles are requested before the closing barriers (the X registers are dead by now): their latency
  // passes under the barriers and the first slab pass
  u32x4 q1[2][ECH], q2[2][ECH];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < ECH; ++i) { q1[mt][i] = fetch(p.R1, p.ldr1, mt, i); q2[mt][i] = fetch(p.R2, p.ldr2, mt, i); }
  if (!late) __builtin_amdgcn_s_barrier();         // the leading half catches the stagger up
  __builtin_amdgcn_s_barrier();                    // every wave is done with the slabs (their reads were retired in the loop;
                                                   // a __syncthreads() would also wait for the residual tiles just requested)

  const bool o16 = (p.ldo & 7) == 0 && (((uintptr_t)p.out) & 15) == 0;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) *(f32x4*)(slab + fr * EROW + (fg * 40 + nt * 4) * 4) = acc2[mt][nt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < ECH; ++i) {
      if (!e_ok(i, mt)) continue;
      const long m = erow0 + mt * 16 + e_row_(i);
      const int n = ecol0 + e_col_(i);
      const char* sp = slab + e_row_(i) * EROW + e_col_(i) * 4;
      const f32x4 lo = *(const f32x4*)sp, hi = *(const f32x4*)(sp + 16);
      const f32x4 b0 = *(const f32x4*)(p.b2 + n), b1 = *(const f32x4*)(p.b2 + n + 4);
      const int grp = (p.a1 || p.a2) ? (int)(m / p.rpg) : 0;
      const float s1 = p.a1 ? p.a1[grp] : 1.0f;
      const float s2 = p.a2 ? p.a2[grp] : 1.0f;
      u32x4 r1 = q1[mt][i];
      const u32x4 r2 = q2[mt][i];
      if (AV) {                                      // the residual is bf16(x + vector), as layernorm's sum_out rounded it
        const float* a = p.av + (long)((int)m / p.rpg_av) * GC + n;
        const f32x4 a0 = *(const f32x4*)a, a1 = *(const f32x4*)(a + 4);
        const float av8[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          r1[j] = pack_bf16x2(__uint_as_float(r1[j] << 16) + av8[2 * j], __uint_as_float(r1[j] & 0xffff0000u) + av8[2 * j + 1]);
      }
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
template <int LNM>
static int ffn2_launch_t(const Ffn2Params& p, void* stream) {
  static bool attr_done[HI3D_MAX_DEVICES] = {};
  if (int rc = hi3d_raise_lds_limit((const void*)ffn2_geglu_c320_kernel<LNM>, FFN2_LDS, attr_done)) return rc;
  hipLaunchKernelGGL(ffn2_geglu_c320_kernel<LNM>, dim3((p.M + GBM - 1) / GBM), dim3(512), FFN2_LDS, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}
static int ffn2_launch(const Ffn2Params& p, bool ln, void* stream) {
  return !ln ? ffn2_launch_t<0>(p, stream) : p.av ? ffn2_launch_t<2>(p, stream) : ffn2_launch_t<1>(p, stream);
}

int hi3d_ffn2_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                     const void* r1, const void* r2, const float* a1, const float* a2, void* out,
                     int32_t M, int32_t ldx, int32_t ldo, int32_t ldr1, int32_t ldr2, int32_t rows_per_group, void* stream) {
  Ffn2Params p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.b1 = b1; p.W2 = (const char*)w2; p.b2 = b2;
  p.R1 = (const unsigned short*)r1; p.R2 = (const unsigned short*)r2; p.a1 = a1; p.a2 = a2;
  p.out = (unsigned short*)out; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr1 = ldr1; p.ldr2 = ldr2;
  p.rpg = rows_per_group < 1 ? 1 : rows_per_group;
  p.ln_g = nullptr; p.ln_b = nullptr; p.av = nullptr; p.ln_eps = 0.f; p.rpg_av = 1;
  return ffn2_launch(p, false, stream);
}

extern "C" int hi3d_ffn_geglu_ln(const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                 const float* addvec, int32_t addvec_rows_per_group,
                                 const void* w1, const float* b1, const void* w2, const float* b2,
                                 const void* r1, const void* r2, const float* a1, const float* a2, void* out,
                                 int32_t M, int32_t C, int32_t ldx, int32_t ldo, int32_t ldr1, int32_t ldr2,
                                 int32_t rows_per_group, void* stream) {
  if (!x || !ln_gamma || !ln_beta || !w1 || !b1 || !w2 || !b2 || !out) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu_ln: null pointer");
  if (M <= 0) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu_ln: non-positive M");
  if (C != GC) HI3D_FAIL(HI3D_ESHAPE, "ffn_geglu_ln: only the 320-channel level is fused (layernorm + two GEMMs otherwise)");
  if (ldx < C || ldo < C || (ldx % 8) || (ldo % 4)) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu_ln: bad ldx / ldo");
  if ((r1 && (ldr1 < C || ldr1 % 4)) || (r2 && (ldr2 < C || ldr2 % 4))) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu_ln: bad residual leading dim");
  if ((a1 || a2) && rows_per_group < 1) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu_ln: rows_per_group < 1");
  if (addvec && addvec_rows_per_group < GBM)
    HI3D_FAIL(HI3D_ESHAPE, "ffn_geglu_ln: addvec_rows_per_group must be >= 128 (a 128-row block stages at most two vectors)");
  if (addvec && !r1) HI3D_FAIL(HI3D_EINVAL, "ffn_geglu_ln: addvec is added to the normalised input AND to the residual r1 -- r1 missing");
  if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2) & 15) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu_ln: x / w1 / w2 not 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)r1 | (uintptr_t)r2) & 7) HI3D_FAIL(HI3D_EALIGN, "ffn_geglu_ln: out / residuals not 8-byte aligned");
  if (((uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)addvec) & 15)
    HI3D_FAIL(HI3D_EALIGN, "ffn_geglu_ln: biases / layernorm vectors / addvec not 16-byte aligned");
  Ffn2Params p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.b1 = b1; p.W2 = (const char*)w2; p.b2 = b2;
  p.R1 = (const unsigned short*)r1; p.R2 = (const unsigned short*)r2; p.a1 = a1; p.a2 = a2;
  p.out = (unsigned short*)out; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr1 = ldr1; p.ldr2 = ldr2;
  p.rpg = rows_per_group < 1 ? 1 : rows_per_group;
  p.ln_g = ln_gamma; p.ln_b = ln_beta; p.av = addvec; p.ln_eps = ln_eps; p.rpg_av = addvec ? addvec_rows_per_group : 1;
  return ffn2_launch(p, true, stream);
}

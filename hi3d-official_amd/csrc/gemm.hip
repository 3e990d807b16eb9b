// bf16 MFMA GEMM with gathered A operand (dense / 3x3 conv / temporal 3-tap conv)
// and fused epilogues, for gfx950.  One kernel family covers nn.Linear, Conv2d 3x3
// (stride 1/2, nearest-2x upsample folded in), Conv2d 1x1 and Conv3d (3,1,1) of the
// Hi3D VideoUNet / VAE (reference call sites listed in include/hi3d_hip.h).
//
// Tile: BM=128 rows x BN=32*NT cols (NT=4 -> 128, NT=5 -> 160; every channel count
// of the UNet is a multiple of 320 = 2*160) x BK=64.  4 waves as 2(M) x 2(N), wave
// tile 64 x 16*NT built from v_mfma_f32_16x16x32_bf16.  Both operands reach LDS by
// LDS-DMA (global_load_lds, 16 B/lane) into a 2-stage ring; the LDS image is
// lane-linear so the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free is applied to the per-lane *source* chunk and again on the read.
//
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation
// rows) and the weight rows of a wave tile are visited in the order
//   n = g*4NT + nt*4 + r      (g = lane>>4, r = accumulator register)
// so every lane ends up holding 4*NT *consecutive* output columns of one output
// row: the epilogue reads residuals and writes results 8 bytes at a time.
#include "common.h"
#include <stdlib.h>

namespace {

struct GemmParams {
  const char* A; const char* W; const float* bias; const float* rowvec;
  const unsigned short* R1; const unsigned short* R2; const float* a1; const float* a2;
  void* out;
  int M, N, K, lda, ldo, ldr1, ldr2, ldrv, ldw, rpg, out_fp32;
  int Hin, Win, Cin, Hout, Wout, stride, up2x, T, HW;
  int nbm, nbn;
};

constexpr int BM = 128, BK = 64;

// The kernel can walk a contiguous range of tiles per block (persistent form, next tile's
// loads in flight during the epilogue).  On MI355X the dynamic one-block-per-tile dispatch
// measured faster (hardware staggers co-resident blocks; persistent blocks run in lockstep),
// so the default grid is one block per tile; HI3D_GEMM_BLOCKS_PER_CU=k selects k*CUs blocks.
int persistent_grid() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const char* e = getenv("HI3D_GEMM_BLOCKS_PER_CU");   // tuning knob (0 = one block per tile)
    const int per_cu = e ? atoi(e) : 0;   // measured: one block per tile beats the persistent range at every Hi3D shape
    g = per_cu > 0 ? per_cu * cus : 0x7fffffff;
  }
  return g;
}

template <int NT, int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
  constexpr int BN = 32 * NT;
  constexpr int A_BYTES = BM * BK * 2;      // 16 KiB
  constexpr int B_BYTES = BN * BK * 2;      // 16 / 20 KiB
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // ---- persistent block: a contiguous range of output tiles (n fastest, so consecutive
  // tiles re-read the same activation rows from L2).  Blocks are renumbered XCD-aware
  // (bijective for any grid) so that neighbouring ranges share one XCD's L2.
  const int tiles_total = p.nbm * p.nbn;
  const int nblk = gridDim.x;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tpb = (tiles_total + nblk - 1) / nblk;
  const int tile_begin = lid * tpb;
  const int tile_end = min(tiles_total, tile_begin + tpb);
  if (tile_begin >= tile_end) return;

  // ---- per-thread gather state.  LDS row r of a tile is filled by the 8 lanes
  // (r&7 within an 8-row, 1 KiB DMA piece); lane slot s carries source chunk
  // s ^ swz(r).  Chunk assignment is tile independent; row bases are set per tile.
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_chunk[4], b_chunk[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_chunk[i] = lslot ^ ((((w * 4 + i) * 8 + lrow) >> 1) & 7);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int j = (w * NT + i) * 8 + lrow;        // row of the W tile, 0..BN-1
    const int jw = j % (16 * NT);                  // row within its wave tile
    const int fi = (jw / (4 * NT)) * 4 + (jw & 3); // MFMA row index that reads it
    b_chunk[i] = lslot ^ ((fi >> 1) & 7);
  }
  int a_row_valid = 0, b_row_valid = 0;
  long a_base[4], b_base[NT];
  int a_p0[4], a_p1[4];
  int tap = 0, c0 = 0;   // conv modes: current tap and channel offset of the K chunk

  auto setup = [&](int tile) {
    const int m0 = (tile / p.nbn) * BM, n0 = (tile % p.nbn) * BN;
    a_row_valid = 0; b_row_valid = 0; tap = 0; c0 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + (w * 4 + i) * 8 + lrow;
      const bool ok = m < p.M;
      a_row_valid |= ok ? (1 << i) : 0;
      const int mm = ok ? m : 0;
      if (AMODE == HI3D_A_DENSE) {
        a_base[i] = (long)mm * p.lda * 2;
        a_p0[i] = a_p1[i] = 0;
      } else if (AMODE == HI3D_A_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int f = mm / hw, rem = mm - f * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        a_base[i] = (long)f * p.Hin * p.Win;       // pixel index of frame origin
        a_p0[i] = oy * p.stride; a_p1[i] = ox * p.stride;
      } else {
        const int f = mm / p.HW;                    // frame index (b*T + t)
        a_base[i] = (long)mm * p.Cin * 2;
        a_p0[i] = f % p.T; a_p1[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + (w * NT + i) * 8 + lrow;
      const bool ok = n < p.N;
      b_row_valid |= ok ? (1 << i) : 0;
      b_base[i] = (long)(ok ? n : 0) * p.ldw * 2;
    }
  };

  const char* zero = (const char*)hi3d_zero_page;
  auto issue = [&](int kt, int st) {
    char* sA = smem + st * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = kt * BK;
    int dy = 0, dx = 0;
    if (AMODE == HI3D_A_CONV3X3) { dy = tap / 3; dx = tap - dy * 3; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src;
      bool ok = (a_row_valid >> i) & 1;
      if (AMODE == HI3D_A_DENSE) {
        src = p.A + a_base[i] + (long)(k0 + a_chunk[i] * 8) * 2;
      } else if (AMODE == HI3D_A_CONV3X3) {
        int iy = a_p0[i] + dy - 1, ix = a_p1[i] + dx - 1;
        if (p.up2x) {
          ok = ok && iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else {
          ok = ok && iy >= 0 && ix >= 0 && iy < p.Hin && ix < p.Win;
        }
        src = p.A + ((a_base[i] + (long)iy * p.Win + ix) * p.Cin + c0 + a_chunk[i] * 8) * 2;
      } else {
        const int tt = a_p0[i] + tap - 1;
        ok = ok && tt >= 0 && tt < p.T;
        src = p.A + a_base[i] + ((long)(tap - 1) * p.HW * p.Cin + c0 + a_chunk[i] * 8) * 2;
      }
      lds_dma16(ok ? src : zero, sA + (w * 4 + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const char* src = p.W + b_base[i] + (long)(k0 + b_chunk[i] * 8) * 2;
      lds_dma16(((b_row_valid >> i) & 1) ? src : zero, sB + (w * NT + i) * 1024);
    }
    if (AMODE != HI3D_A_DENSE) { c0 += BK; if (c0 >= p.Cin) { c0 = 0; ++tap; } }
  };

  // ---- fragment read addresses (bytes within a stage)
  const int fr = lane & 15, fg = lane >> 4;
  const int x_sw = (fr >> 1) & 7;
  int x_off[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) x_off[mt] = (wm * 64 + mt * 16 + fr) * 128;
  int w_off[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    w_off[nt] = A_BYTES + (wn * 16 * NT + (fr >> 2) * 4 * NT + nt * 4 + (fr & 3)) * 128;
  const int w_sw = (fr >> 1) & 7;

  const int nk = p.K / BK;
  // issue cursor runs one K-step ahead of the compute cursor, across tile boundaries:
  // the first loads of tile t+1 are in flight while tile t runs its epilogue.
  int i_tile = tile_begin, i_k = 0;
  auto advance = [&]() {
    if (++i_k == nk) { i_k = 0; ++i_tile; if (i_tile < tile_end) setup(i_tile); }
  };
  setup(i_tile);
  issue(0, 0);
  advance();
  int st = 0;

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int m0 = (tile / p.nbn) * BM, n0 = (tile % p.nbn) * BN;
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane (fg, fr) owns rows m = m0 + wm*64 + mt*16 + fr, columns nb .. nb + 4*NT - 1
    const int nb = n0 + wn * 16 * NT + fg * 4 * NT;
    uint2 r1v[4][NT];

    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                      // stage st landed; stage st^1 free again
      if (i_tile < tile_end) { issue(i_k, st ^ 1); advance(); }
      if (EPI == HI3D_EPI_AFFINE && kt == nk - 1 && p.R1) {
        // residual tile: fetch under the last K-step's MFMAs instead of in the epilogue
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int m = m0 + wm * 64 + mt * 16 + fr;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = nb + nt * 4;
            r1v[mt][nt] = (m < p.M && n < p.N) ? *(const uint2*)(p.R1 + (long)m * p.ldr1 + n) : make_uint2(0, 0);
          }
        }
      }
      const char* s = smem + st * STAGE;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        bf16x8 xf[4], wf[NT];
        const int cx = ((kh * 4 + fg) ^ x_sw) << 4;
        const int cw = ((kh * 4 + fg) ^ w_sw) << 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(s + x_off[mt] + cx);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const bf16x8*)(s + w_off[nt] + cw);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
      }
      st ^= 1;
    }

    // ---- epilogue
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + fr;
      if (m >= p.M) continue;
      const int grp = m / p.rpg;
      const float s1 = p.a1 ? p.a1[grp] : 1.0f;
      const float s2 = p.a2 ? p.a2[grp] : 1.0f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nb + nt * 4;
        if (n >= p.N) continue;
        float v[4] = {acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
        if (p.bias) {
          const f32x4 b = *(const f32x4*)(p.bias + n);
          v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        }
        if (EPI == HI3D_EPI_GEGLU) {
          const float o0 = v[0] * gelu_erf_f(v[2]);
          const float o1 = v[1] * gelu_erf_f(v[3]);
          unsigned int* o = (unsigned int*)((unsigned short*)p.out + (long)m * p.ldo + (n >> 1));
          *o = pack_bf16x2(o0, o1);
        } else {
          if (p.rowvec) {
            const f32x4 b = *(const f32x4*)(p.rowvec + (long)grp * p.ldrv + n);
            v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
          }
          if (p.R1) {
            const uint2 r = r1v[mt][nt];
            v[0] += bf16_to_f32(r.x & 0xffff); v[1] += bf16_to_f32(r.x >> 16);
            v[2] += bf16_to_f32(r.y & 0xffff); v[3] += bf16_to_f32(r.y >> 16);
          }
          v[0] *= s1; v[1] *= s1; v[2] *= s1; v[3] *= s1;
          if (p.R2) {
            const uint2 r = *(const uint2*)(p.R2 + (long)m * p.ldr2 + n);
            v[0] += s2 * bf16_to_f32(r.x & 0xffff); v[1] += s2 * bf16_to_f32(r.x >> 16);
            v[2] += s2 * bf16_to_f32(r.y & 0xffff); v[3] += s2 * bf16_to_f32(r.y >> 16);
          }
          if (p.out_fp32) {
            *(f32x4*)((float*)p.out + (long)m * p.ldo + n) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((unsigned short*)p.out + (long)m * p.ldo + n) = o;
          }
        }
      }
    }
  }
}

template <int NT, int AMODE, int EPI>
int launch(const GemmParams& p, hipStream_t stream) {
  constexpr int BN = 32 * NT;
  constexpr int smem = 2 * (BM * BK * 2 + BN * BK * 2);
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<NT, AMODE, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { hi3d_set_error(hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int tiles = p.nbm * p.nbn;
  const int grid = tiles < persistent_grid() ? tiles : persistent_grid();
  hipLaunchKernelGGL((gemm_bf16_kernel<NT, AMODE, EPI>), dim3(grid), dim3(256), smem, stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

template <int NT>
int dispatch(const GemmParams& p, int amode, int epi, hipStream_t s) {
  if (epi == HI3D_EPI_GEGLU) {
    if (amode != HI3D_A_DENSE) HI3D_FAIL(HI3D_ESHAPE, "gemm: GEGLU epilogue only with dense A");
    return launch<NT, HI3D_A_DENSE, HI3D_EPI_GEGLU>(p, s);
  }
  switch (amode) {
    case HI3D_A_DENSE: return launch<NT, HI3D_A_DENSE, HI3D_EPI_AFFINE>(p, s);
    case HI3D_A_CONV3X3: return launch<NT, HI3D_A_CONV3X3, HI3D_EPI_AFFINE>(p, s);
    case HI3D_A_CONVT3: return launch<NT, HI3D_A_CONVT3, HI3D_EPI_AFFINE>(p, s);
  }
  HI3D_FAIL(HI3D_EINVAL, "gemm: bad amode");
}

}  // namespace

extern "C" int hi3d_gemm_bf16(const hi3d_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->out) HI3D_FAIL(HI3D_EINVAL, "gemm: null pointer");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) HI3D_FAIL(HI3D_EINVAL, "gemm: non-positive size");
  if (d->K % 64) HI3D_FAIL(HI3D_ESHAPE, "gemm: K must be a multiple of 64");
  if (d->N % 4) HI3D_FAIL(HI3D_ESHAPE, "gemm: N must be a multiple of 4");
  if (d->rows_per_group < 1) HI3D_FAIL(HI3D_EINVAL, "gemm: rows_per_group < 1");
  if (d->epi != HI3D_EPI_AFFINE && d->epi != HI3D_EPI_GEGLU) HI3D_FAIL(HI3D_EINVAL, "gemm: bad epi");
  if (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->out) & 15) HI3D_FAIL(HI3D_EALIGN, "gemm: A/W/out not 16-byte aligned");
  GemmParams p;
  p.A = (const char*)d->A; p.W = (const char*)d->W; p.bias = d->bias; p.rowvec = d->rowvec;
  p.R1 = (const unsigned short*)d->R1; p.R2 = (const unsigned short*)d->R2; p.a1 = d->a1; p.a2 = d->a2;
  p.out = d->out; p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldo = d->ldo;
  p.ldr1 = d->ldr1; p.ldr2 = d->ldr2; p.ldrv = d->ldrv > 0 ? d->ldrv : d->N; p.ldw = d->ldw > 0 ? d->ldw : d->K; p.rpg = d->rows_per_group; p.out_fp32 = d->out_fp32;
  p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
  p.stride = d->stride; p.up2x = d->up2x; p.T = d->T; p.HW = d->HW;
  if (d->amode == HI3D_A_DENSE) {
    if (d->lda < d->K || (d->lda % 8)) HI3D_FAIL(HI3D_EALIGN, "gemm: lda < K or lda % 8 != 0");
  } else if (d->amode == HI3D_A_CONV3X3) {
    if (d->Cin <= 0 || d->Cin % 64 || d->K != 9 * d->Cin) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: need Cin % 64 == 0 and K == 9*Cin");
    if (d->Hin <= 0 || d->Win <= 0 || d->Hout <= 0 || d->Wout <= 0) HI3D_FAIL(HI3D_EINVAL, "conv3x3: bad geometry");
    if (d->stride != 1 && d->stride != 2) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: stride must be 1 or 2");
    if (d->up2x && d->stride != 1) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: up2x needs stride 1");
    const int eh = d->up2x ? 2 * d->Hin : (d->Hin + 2 - 3) / d->stride + 1;
    const int ew = d->up2x ? 2 * d->Win : (d->Win + 2 - 3) / d->stride + 1;
    if (eh != d->Hout || ew != d->Wout) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: Hout/Wout inconsistent with Hin/Win/stride");
    if (d->M % (d->Hout * d->Wout)) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: M not a multiple of Hout*Wout");
  } else if (d->amode == HI3D_A_CONVT3) {
    if (d->Cin <= 0 || d->Cin % 64 || d->K != 3 * d->Cin) HI3D_FAIL(HI3D_ESHAPE, "convt3: need Cin % 64 == 0 and K == 3*Cin");
    if (d->T <= 0 || d->HW <= 0 || d->M % (d->T * d->HW)) HI3D_FAIL(HI3D_ESHAPE, "convt3: M not a multiple of T*HW");
  } else {
    HI3D_FAIL(HI3D_EINVAL, "gemm: bad amode");
  }
  const int n_out = d->epi == HI3D_EPI_GEGLU ? d->N / 2 : d->N;
  if (d->ldo < n_out) HI3D_FAIL(HI3D_EINVAL, "gemm: ldo < N");
  if (p.ldw < d->K || p.ldw % 8) HI3D_FAIL(HI3D_EALIGN, "gemm: ldw < K or ldw % 8 != 0");
  if (d->rowvec && (p.ldrv < d->N || p.ldrv % 4)) HI3D_FAIL(HI3D_EALIGN, "gemm: bad ldrv");
  if ((d->ldo % 4) || (d->R1 && d->ldr1 % 4) || (d->R2 && d->ldr2 % 4)) HI3D_FAIL(HI3D_EALIGN, "gemm: ld % 4 != 0");
  int tile = d->tile_n;
  if (tile == 0) {
    // 160 suits every multiple of 320; otherwise pick the tile that wastes fewer columns
    const int w128 = (d->N + 127) / 128 * 128, w160 = (d->N + 159) / 160 * 160;
    tile = (w160 < w128) ? 160 : 128;
  }
  if (tile != 128 && tile != 160) HI3D_FAIL(HI3D_EINVAL, "gemm: tile_n must be 0, 128 or 160");
  p.nbm = (d->M + BM - 1) / BM;
  p.nbn = (d->N + tile - 1) / tile;
  hipStream_t s = (hipStream_t)stream;
  return tile == 160 ? dispatch<5>(p, d->amode, d->epi, s) : dispatch<4>(p, d->amode, d->epi, s);
}

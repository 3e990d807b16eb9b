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
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "gemm_params.h"

namespace {

// WM   : waves along M (2 -> 128-row tile, 4 waves; 4 -> 256-row tile, 8 waves); 2 waves along N
// NT   : 16-column MFMA tiles per wave along N (4 -> BN = 128, 5 -> BN = 160)
// NS   : LDS ring stages.  NS = 2: loads of K-step k+1 fly during K-step k (vmcnt(0) per step).
//        NS = 3: two K-steps in flight, counted vmcnt (the newest stage's LDS-DMA stays in
//        flight across the raw s_barrier) -- hides HBM latency for the short-K shapes.
// PP   : "ping-pong" K loop (8-wave tiles only, see below): the two waves of a SIMD run half a phase apart, one in
//        its MFMA segment while the other reads fragments / issues LDS-DMA.
// per-wave epilogue (every wave stages its own 64 x 16*NT tile through a private LDS slab): the wide ping-pong tiles, and -- round 6 --
// the single-stage 128 x 128 tile (three blocks per CU: the VAE's 128-channel convs, whose 18 K steps are no longer than the
// block-wide epilogue's barriers)
constexpr bool WEPI_OF(int WM, int NT, int NS) { return (NT > 5 && WM == 4) || (NT == 4 && WM == 2 && NS == 1); }

template <int WM, int NT, int NS, int AMODE_, int EPI, bool PP = false>
__global__ __launch_bounds__(WM * 128, (NT == 4 && WM == 2 && NS == 1) ? 4 : 2) void gemm_bf16_kernel(const GemmParams p) {
#if __HIP_DEVICE_COMPILE__   // buffer-resource builtins exist only in the device pass; the host pass needs just the stub
  constexpr int NW = WM * 2;                 // waves per block
  constexpr bool UP2X = AMODE_ == A_CONV3X3_UP2X;
  constexpr bool DENSE2 = AMODE_ == A_DENSE2;
  constexpr bool PHASE = AMODE_ == A_CONV3X3_PHASE;      // (wide tiles only: the per-wave store loop below places the rows)
  constexpr int AMODE = (UP2X || PHASE) ? HI3D_A_CONV3X3 : DENSE2 ? HI3D_A_DENSE : AMODE_;
  constexpr int BM = WM * 64, BN = 32 * NT;
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_PIECES = BM / 8 / NW;                  // 1 KiB LDS-DMA pieces per wave: always 4
  constexpr int W_PIECES = (4 * NT + NW - 1) / NW;       // per wave, upper bound (piece q = w + NW*i < 4*NT)
  constexpr int LPS_MIN = A_PIECES + (4 * NT) / NW;      // fewest LDS-DMA ops any wave issues per stage
  static_assert(A_PIECES == 4, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // ---- block -> tile, XCD-aware: consecutive logical ids (which share the A tile
  // and sweep W) stay on one XCD's L2.  Bijective for any grid size.
  const int nblk = p.nbm * p.nbn;
  int lid, ks = 0;
  {
    int bid = blockIdx.x;
    if (p.ksplit > 1) { ks = bid / nblk; bid -= ks * nblk; }     // K split outermost: the tiles of one split run side by side
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // logical id -> tile.  n fastest (measured 15x less L2 -> fabric fetch than m fastest); and when the weight matrix does not
  // fit an XCD's 4 MB L2 (GEGLU: 6.5 - 26 MB; round-2 PMC: W re-streamed from the Infinity Cache for every pair of m-tiles,
  // 15x its size per launch), in COLUMN GROUPS of gn n-tiles: the blocks resident on an XCD cover (32 / gn) m-tiles x gn
  // n-tiles, the group's slice of W (gn * BN * K * 2 bytes <= ~2 MB) stays in L2 while m is walked, and an A tile is shared
  // by the gn blocks that run side by side.  Bijective for any nbm, nbn, gn (last group narrower).
  int tn, tm;
  if (p.gn <= 0 || p.gn >= p.nbn) {
    tn = lid % p.nbn; tm = lid / p.nbn;
  } else {
    const int per = p.gn * p.nbm, g = lid / per;
    const int gw = min(p.gn, p.nbn - g * p.gn), rem = lid - g * per;
    tm = rem / gw; tn = g * p.gn + rem - tm * gw;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  // Conv3d (3,1,1), clip boundaries: when the block's rows lie in one frame (HW % BM == 0) and that frame is the first / last
  // of its clip, tap 0 / tap 2 reads nothing but the clip's zero padding -- its K steps are left out (2 of 3 T taps of such a
  // block: 4 % of the launch at T = 16).  Block-uniform: scalar registers only.
  int tap_lo = 0, tap_hi = AMODE_ == HI3D_A_CONV3X3 || AMODE_ == A_CONV3X3_UP2X || AMODE_ == A_CONV3X3_PHASE ? p.ntap : 3;
  int nk_ = p.ksplit > 1 ? p.nk_split : p.K / BK;
  if (AMODE_ == HI3D_A_CONVT3 && p.tskip && p.ksplit <= 1 && p.HW % BM == 0 && p.T > 1) {
    const int t = (m0 / p.HW) % p.T;
    if (t == 0) tap_lo = 1;
    if (t == p.T - 1) tap_hi = 2;
    nk_ = (tap_hi - tap_lo) * (p.Cin / BK);
  }
  const int nk = nk_;
  // A_CONV3X3_PHASE: the 16 rows of store pass mt of this wave are 16 consecutive pixels of ONE low-resolution image row (16 | Win):
  // they land on every second row of the 2x image from a per-pass base row -- twice the row pitch, the base as the store's scalar
  // offset.  The four bases are computed HERE, before the K loop (an integer division each, wave-uniform), not in the store loop.
  int so_ph[4] = {0, 0, 0, 0};
  if constexpr (PHASE) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int r0 = __builtin_amdgcn_readfirstlane(m0 + wm * 64 + mt * 16);
      so_ph[mt] = __builtin_amdgcn_readfirstlane((2 * r0 + 2 * p.Win * (r0 / p.Win) + p.phase_c) * p.ldo * 2);
    }
  }


  // ---- per-thread gather state.  LDS row r of a tile is filled by the 8 lanes
  // (r&7 within an 8-row, 1 KiB DMA piece); lane slot s carries source chunk s ^ swz(r).
  // All global addressing is 32-bit: a buffer descriptor based at this block's first
  // row (or frame), a per-lane byte offset fixed for the whole K loop, and a scalar
  // offset that walks K (and the conv taps).  Out-of-range lanes (M/N tails, conv
  // padding) carry an offset beyond num_records: the buffer unit returns zeros for them.
  constexpr unsigned INV = 0x80000000u;
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned a_voff[4];           // byte offset of this lane's 16-byte chunk (or INV)
  int a_mask[4];                // conv: bit `tap` set when the tap is inside the image / clip
  int a_p0[4], a_p1[4];         // up2x only: output pixel coordinates
  const char* a_origin;
  unsigned a_voff2[DENSE2 ? 4 : 1];      // DENSE2: the same rows in the second source (its own pitch)
  const int kt0 = DENSE2 ? ks * nk : 0;  // DENSE2 walks GLOBAL K chunks (the source switches at K1); plain dense folds ks into the bases
  if (DENSE2) {
    a_origin = p.A + (long)m0 * p.lda * 2;
  } else if (AMODE == HI3D_A_DENSE) {
    a_origin = p.A + (long)m0 * p.lda * 2 + (long)ks * nk * (BK * 2);     // (split-K: this block's first K chunk)
  } else if (AMODE == HI3D_A_CONV3X3) {
    const int f0 = m0 / (p.Hout * p.Wout);
    // origin shifted back by one row + one pixel so that every tap offset is >= 0
    a_origin = p.A + ((long)f0 * p.Hin * p.Win - (UP2X ? 0 : (p.Win + 1) * p.pad)) * p.Cin * 2;
  } else {
    a_origin = p.A + ((long)m0 - p.HW) * p.Cin * 2;      // one frame back: temporal tap 0
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (w * 4 + i) * 8 + lrow;
    const int m = m0 + r;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const bool ok = m < p.M;
    a_mask[i] = 0; a_p0[i] = a_p1[i] = 0;
    if (AMODE == HI3D_A_DENSE) {
      a_voff[i] = ok ? (unsigned)(r * p.lda * 2 + chunk * 16) : INV;
      if (DENSE2) a_voff2[i] = ok ? (unsigned)(r * p.lda2 * 2 + chunk * 16) : INV;
    } else if (AMODE == HI3D_A_CONV3X3) {
      const int hw = p.Hout * p.Wout;
      const int mm = ok ? m : m0;
      const int f = mm / hw, rem = mm - f * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      const int frel = f - m0 / hw;
      if (UP2X) {
        a_p0[i] = oy; a_p1[i] = ox;
        a_voff[i] = ok ? (unsigned)((frel * p.Hin * p.Win) * p.Cin * 2 + chunk * 16) : INV;
      } else {
        const int iy = oy * p.stride, ix = ox * p.stride;
        a_voff[i] = ok ? (unsigned)((((frel * p.Hin + iy) * p.Win + ix) * p.Cin) * 2 + chunk * 16) : INV;
        int ym = 0, xm = 0;                          // bit d: tap row/column d lies inside the image
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          ym |= (iy + d - p.pad >= 0 && iy + d - p.pad < p.Hin) ? (1 << d) : 0;
          xm |= (ix + d - p.pad >= 0 && ix + d - p.pad < p.Win) ? (1 << d) : 0;
        }
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) mk |= (((ym >> (t / 3)) & (xm >> (t % 3))) & 1) << t;
        a_mask[i] = mk;
      }
    } else {
      const int t = ((ok ? m : m0) / p.HW) % p.T;
      a_voff[i] = ok ? (unsigned)(r * p.Cin * 2 + chunk * 16) : INV;
      a_mask[i] = (t >= 1 ? 1 : 0) | 2 | (t + 1 < p.T ? 4 : 0);
    }
  }
  unsigned b_voff[W_PIECES];
#pragma unroll
  for (int i = 0; i < W_PIECES; ++i) {
    const int q = w + NW * i;                      // W piece (8 rows) handled by this wave
    const int j = q * 8 + lrow;                    // row of the W tile, 0..BN-1
    const int jw = j % (16 * NT);                  // row within its wave tile
    const int fi = (jw / (4 * NT)) * 4 + (jw & 3); // MFMA row index that reads it
    const int chunk = lslot ^ ((fi >> 1) & 7);
    b_voff[i] = (n0 + j < p.N) ? (unsigned)(j * p.ldw * 2 + chunk * 16) : INV;
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_origin, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(DENSE2 ? p.A2 + (long)m0 * p.lda2 * 2 : a_origin), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.W + (long)n0 * p.ldw * 2 + (AMODE == HI3D_A_DENSE ? (long)ks * nk * (BK * 2) : 0L) +
              (p.wgs ? (long)(m0 / p.rpg) * p.wgs * 2 : 0L)),     // (per-row-group weights: the tile lies inside ONE group)
      0, 0x7fffffff, 0x00020000);

  // conv modes: current tap and channel offset of the K chunk (split-K starts at a slab boundary: nk is a multiple of the taps)
  int tap = tap_lo, c0 = AMODE == HI3D_A_DENSE ? 0 : ks * (nk / (AMODE == HI3D_A_CONV3X3 ? p.ntap : 3)) * BK;

  // LDS-DMA pieces [LO, HI) of K chunk `kt` into ring slot `st`: pieces 0..3 are this wave's A rows, 4.. its W rows
  // (the ping-pong loop spreads the pieces of a stage over its phases; the plain loops issue them all at once)
  auto issue_pieces = [&](int kt, int st, auto lo_, auto hi_) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
    char* sA = smem + st * STAGE;
    char* sB = sA + A_BYTES;
    unsigned soff = 0;                             // scalar byte offset of this K chunk
    bool second = false;                           // DENSE2: this chunk lies in the second source (block-uniform)
    if (DENSE2) { const int kc = (kt0 + kt) * BK; second = kc >= p.K1; soff = (second ? kc - p.K1 : kc) * 2; }
    else if (AMODE == HI3D_A_DENSE) soff = kt * (BK * 2);
    // (tap subset: the k-th K slab of a channel block reads image tap p.taps[k]; its weights are slab k of W)
    const int tp = (AMODE == HI3D_A_CONV3X3 && !UP2X && p.taps) ? (int)((p.taps >> (4 * tap)) & 15u) : tap;
    if (DENSE2 || AMODE == HI3D_A_DENSE) {}
    else if (AMODE == HI3D_A_CONV3X3) soff = UP2X ? c0 * 2 : (((tp / 3) * p.Win + (tp % 3)) * p.Cin + c0) * 2;
    else soff = (tap * p.HW * p.Cin + c0) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < LO || i >= HI) continue;
      unsigned vo = a_voff[i];
      if (AMODE == HI3D_A_CONV3X3 && UP2X) {
        const int iy = a_p0[i] + tap / 3 - 1, ix = a_p1[i] + tap % 3 - 1;   // on the virtual 2H x 2W grid
        const bool in = iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
        vo = (in && vo != INV) ? vo + (unsigned)(((iy >> 1) * p.Win + (ix >> 1)) * p.Cin * 2) : INV;
      } else if (AMODE != HI3D_A_DENSE) {
        vo = ((a_mask[i] >> tp) & 1) ? vo : INV;
      }
      if (DENSE2 && second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (LDS_AS void*)(sA + (w * 4 + i) * 1024), 16, a_voff2[i], soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(sA + (w * 4 + i) * 1024), 16, vo, soff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      if (4 + i < LO || 4 + i >= HI) continue;
      const int q = w + NW * i;
      if (q < 4 * NT)                               // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (LDS_AS void*)(sB + q * 1024), 16, b_voff[i],
                                                 AMODE == HI3D_A_DENSE ? kt * (BK * 2) : (tap * p.Cin + c0) * 2, 0, 0);
    }
  };
  // K walks the taps INNERMOST: the 9 (3) shifted reads of one 64-channel slab of the input happen in
  // consecutive K steps, in every block of the wave front at about the same time, so a slab comes from
  // HBM / the fabric once and the other taps hit L2.  (Round 1 walked taps outermost and re-fetched the
  // input per tap: FETCH_SIZE 6.5x the algorithmic bytes.)
  auto advance_k = [&]() {
    if (AMODE != HI3D_A_DENSE) { ++tap; if (tap >= tap_hi) { tap = tap_lo; c0 += BK; } }
  };
  auto issue = [&](int kt, int st) {
    issue_pieces(kt, st, std::integral_constant<int, 0>{}, std::integral_constant<int, 4 + W_PIECES>{});
    advance_k();
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

  f32x4 acc[4][NT];                                // first written by the MFMAs of K step 0 (C = 0)
  constexpr bool PEEL0 = NT <= 5 && AMODE == HI3D_A_DENSE;   // (wide tile, conv gathers: zero-fill instead -- peeling costs them registers / time)
  // wide ping-pong tiles: the accumulators START at bias (+ the tile's row vector) -- 160 moves once per tile instead
  // of 320 adds and 40 LDS reads per wave in the epilogue, which is VALU-bound there (set below, once the vectors landed)
  constexpr bool BIAS_INIT = PP && NT > 5;
  if (!PEEL0 && !BIAS_INIT) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // lane (fg, fr) owns rows m = m0 + wm*64 + mt*16 + fr, columns nb .. nb + 4*NT - 1
  const int nb = n0 + wn * 16 * NT + fg * 4 * NT;

  constexpr int BN_OUT = (EPI == HI3D_EPI_GEGLU) ? BN / 2 : BN;
  constexpr int LROW = BN_OUT * 4 + 16;            // bytes; +16 spreads ds_write_b128 lanes over banks
  constexpr int MPP = (NS == 1 || NT > 5) ? 1 : 2; // 16-row accumulator blocks per wave and pass
  constexpr int NPASS = 4 / MPP;
  constexpr int HR = WM * 16 * MPP;                // rows per pass (fits the ring: checked below)
  static_assert(HR * LROW <= NS * STAGE, "epilogue slab does not fit the LDS ring");
  constexpr int CPR = BN_OUT / 8;                  // 8-column chunks per row
  constexpr int NTHR = NW * 64;
  const int n0_out = (EPI == HI3D_EPI_GEGLU) ? n0 / 2 : n0;
  const int N_out = (EPI == HI3D_EPI_GEGLU) ? p.N / 2 : p.N;
  constexpr int CH = (HR * CPR + NTHR - 1) / NTHR;  // 8-column chunks per thread and pass

  // ---- epilogue operands that do not depend on the accumulators are fetched early, not in the
  // epilogue: a global load issued there queues behind the other resident block's LDS-DMA stream
  // and costs 4-12 K cycles (measured with s_memtime stamps) -- as much as a short K loop.
  // Bias and (when every row of the tile is in one group) the group's row vector and blend
  // factors are fetched before the K loop; the R1 tile is requested two K steps before the
  // loop ends, R2 one pass ahead of its use, both in the row-contiguous 16 B/lane layout of
  // the store pass, through buffer descriptors (out-of-range -> zeros).
  const bool ugrp = p.rpg > 0 && (p.rpg % BM) == 0;
  const int tgrp = ugrp ? m0 / p.rpg : 0;
  // bias[n0 ..] and rowvec[tgrp][n0 ..] land in two LDS slots behind the ring (LDS-DMA by
  // wave 0; a zero-length descriptor zero-fills the slot of an absent vector)
  constexpr int VSLOT = (BN * 4 + 1023) / 1024 * 1024;   // bytes per vector slot (1 KiB per DMA)
  char* const vec_lds = smem + NS * STAGE;
  if (w == 0) {
    const int nrem = (p.N - n0) * 4;
    const bool rvt = EPI == HI3D_EPI_AFFINE && p.rowvec && ugrp;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        p.bias ? (void*)(p.bias + n0) : (void*)p.W, 0, p.bias ? nrem : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        rvt ? (void*)(p.rowvec + (long)tgrp * p.ldrv + n0) : (void*)p.W, 0, rvt ? nrem : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < VSLOT / 1024; ++c) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(vec_lds + c * 1024), 16, lane * 16 + c * 1024, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(vec_lds + VSLOT + c * 1024), 16, lane * 16 + c * 1024, 0, 0, 0);
    }
  }
  float ts1 = 1.0f, ts2 = 1.0f;
  if (EPI == HI3D_EPI_AFFINE && ugrp) { if (p.a1) ts1 = p.a1[tgrp]; if (p.a2) ts2 = p.a2[tgrp]; }

  // chunk c = tid + i*NTHR of a pass covers LDS row lr = c / CPR, columns 8*(c % CPR) .. +7;
  // its tile row is trow(lr) + pass * 16*MPP.
  // (computed once: the divisions by CPR would otherwise be redone in every pass)
  int c_row[CH], c_col[CH], c_lds[CH];             // tile row at pass 0, column, byte offset in the slab
  bool c_in[CH];                                   // chunk exists (inside the slab and left of N)
  auto init_chunks = [&]() {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + i * NTHR, lr = c / CPR;
      c_col[i] = (c - lr * CPR) * 8;
      c_row[i] = (lr / (16 * MPP)) * 64 + (lr % (16 * MPP));
      c_lds[i] = lr * LROW + c_col[i] * 4;
      c_in[i] = c < HR * CPR && n0_out + c_col[i] < N_out;
    }
  };
  // dense GEMMs keep them in registers; the conv gathers have none to spare and recompute
  constexpr bool PRECHUNK = AMODE == HI3D_A_DENSE;
  if (PRECHUNK && NT <= 5) init_chunks();          // (wide tile: after the K loop)
  auto chunk_lr = [&](int i) { return (tid + i * NTHR) / CPR; };
  auto chunk_row = [&](int i) { if (PRECHUNK) return c_row[i]; const int lr = chunk_lr(i); return (lr / (16 * MPP)) * 64 + (lr % (16 * MPP)); };
  auto chunk_col = [&](int i) { return PRECHUNK ? c_col[i] : ((tid + i * NTHR) % CPR) * 8; };
  auto chunk_in = [&](int i) { return PRECHUNK ? c_in[i] : (tid + i * NTHR < HR * CPR && n0_out + chunk_col(i) < N_out); };
  auto chunk_lds = [&](int i) { return PRECHUNK ? c_lds[i] : chunk_lr(i) * LROW + chunk_col(i) * 4; };
  auto chunk_ok = [&](int i, int pass) { return chunk_in(i) && m0 + chunk_row(i) + pass * 16 * MPP < p.M; };
  constexpr int RP = NPASS;                         // R1 passes held in registers (narrow tiles)
  u32x4 r1v[RP][CH], r2v[CH];
  // chunk i of pass `pass` of a residual tile (row-contiguous 16 bytes; void chunks read zeros)
  auto fetch_chunk = [&](const unsigned short* R, int ldr, int pass, int i) -> u32x4 {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(R + (long)m0 * ldr + n0_out), 0, 0x7fffffff, 0x00020000);
    const int so = pass * 16 * MPP * ldr * 2;
    const int col = chunk_col(i);
    const bool has8 = n0_out + col + 8 <= N_out;
    const unsigned vo = chunk_ok(i, pass) ? (unsigned)((chunk_row(i) * ldr + col) * 2) : INV;
    if (has8 && p.vec8) return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);   // 8-byte pieces: narrow or unaligned rows
    const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rs, has8 ? vo + 8 : INV, so, 0);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  auto fetch_residual = [&](const unsigned short* R, int ldr, int pass, u32x4 (&dst)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) dst[i] = fetch_chunk(R, ldr, pass, i);
  };
  // The wide tile (NT > 5) has 160 accumulator registers and none to park residual slabs in during the K loop:
  // it requests them one store pass ahead in the epilogue (below).
  constexpr bool RES_EARLY = NT <= 5 && !WEPI_OF(WM, NT, NS);

  const int pf_kt = nk >= 2 ? nk - 2 : 0;
  int st = 0;
  if constexpr (PP) {
    // ---- ping-pong K loop (256-row tiles: 8 waves, two per SIMD, one block per CU).
    // A K step is cut into phases of 4 x NTH MFMAs; every phase is
    //     [load segment: LDS-DMA pieces of a later stage, ds_read of this phase's fragments, lgkmcnt(0)]
    //     s_barrier  [compute segment: the MFMAs]  s_barrier
    // and waves 4..7 (the second wave of each SIMD) pass one extra barrier before the loop (waves 0..3 pass it
    // after), so the two waves of a SIMD are always in opposite segments: the matrix pipe of a SIMD sees one
    // wave's MFMA cluster while the other waits for LDS -- in the lock-step loop both wait, then both compute.
    // Ring protocol (D = NS-1 stages ahead; all waits counted, raw barriers, never vmcnt(0) in steady state):
    //   WAR  pieces of stage kt+D overwrite the slot last read in step kt-1; those ds_reads were retired by the
    //        lgkmcnt(0) that precedes the barrier ending the reader's load segment, and the earliest writer
    //        (a wave of the leading half) issues one barrier later;
    //   RAW  every wave waits for its own pieces of stage kt+1 (the D-1 newer stages may stay in flight) at the
    //        end of the LAST load segment of step kt; the trailing half does so one barrier before the leading
    //        half's first read of stage kt+1.
    static_assert(WM == 4 && NS >= 2 && NS <= 3, "ping-pong K loop: 8 waves, ring of 2 or 3 stages");
    constexpr int D = NS - 1;
    constexpr int NTH = (NT > 5) ? NT / 2 : NT;   // weight fragments per phase
    constexpr int NHALF = NT / NTH;
    constexpr int NPH = 2 * NHALF;                // phases per K step
    constexpr int NISS = (D == 1) ? NPH - 1 : NPH;   // phases that issue LDS-DMA (ring of 2: none in the phase that waits)
    constexpr int TP = 4 + W_PIECES;              // LDS-DMA pieces per wave and stage
    const bool late = w >= 4;
#pragma unroll
    for (int d = 0; d < D; ++d) if (d < nk) issue(d, d);
    if (D > 1 && nk >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPS_MIN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // stage 0 is in LDS for everyone
    if (BIAS_INIT) {                              // so are both vector slots (requested before stage 0)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cl = (wn * 16 * NT + fg * 4 * NT + nt * 4) * 4;
        f32x4 b = *(const f32x4*)(vec_lds + cl);
        if (EPI == HI3D_EPI_AFFINE) {
          const f32x4 g = *(const f32x4*)(vec_lds + VSLOT + cl);
          b[0] += g[0]; b[1] += g[1]; b[2] += g[2]; b[3] += g[3];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = b;
      }
    } else if (EPI == HI3D_EPI_AFFINE && tid < VSLOT / 16) {   // fold them
      f32x4 b = *(const f32x4*)(vec_lds + tid * 16);    // (here, not inside the loop: there it cost 40 registers)
      const f32x4 g = *(const f32x4*)(vec_lds + VSLOT + tid * 16);
      b[0] += g[0]; b[1] += g[1]; b[2] += g[2]; b[3] += g[3];
      *(f32x4*)(vec_lds + tid * 16) = b;
    }
    if (late) __builtin_amdgcn_s_barrier();       // the stagger
    auto kstep_pp = [&](const int kt, auto first) {
      constexpr bool FIRST = decltype(first)::value;
      const char* s = smem + st * STAGE;
      const int st_pf = st == 0 ? NS - 1 : st - 1;  // ring slot of stage kt + D
      const bool pf = kt + D < nk;
      bf16x8 xf[4];
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        const int kh = ph / NHALF, nh = ph % NHALF;
        const int cx = ((kh * 4 + fg) ^ x_sw) << 4;
        const int cw = ((kh * 4 + fg) ^ w_sw) << 4;
        // ---- load segment
        if (ph < NISS && pf) {
          if (ph == 0) issue_pieces(kt + D, st_pf, std::integral_constant<int, 0>{}, std::integral_constant<int, TP / NISS>{});
          if (ph == 1) issue_pieces(kt + D, st_pf, std::integral_constant<int, TP / NISS>{}, std::integral_constant<int, (NISS == 2) ? TP : 2 * TP / NISS>{});
          if (ph == 2) issue_pieces(kt + D, st_pf, std::integral_constant<int, 2 * TP / NISS>{}, std::integral_constant<int, (NISS == 3) ? TP : 3 * TP / NISS>{});
          if (ph == 3) issue_pieces(kt + D, st_pf, std::integral_constant<int, 3 * TP / NISS>{}, std::integral_constant<int, TP>{});
          if (ph == NISS - 1) advance_k();
        }
        if (ph == 0 && EPI == HI3D_EPI_AFFINE && NT <= 5 && p.R1 && kt == pf_kt) {
#pragma unroll
          for (int pass = 0; pass < RP; ++pass) fetch_residual(p.R1, p.ldr1, pass, r1v[pass]);
        }
        bf16x8 wf[NTH];
        if (nh == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(s + x_off[mt] + cx);
        }
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) wf[nt] = *(const bf16x8*)(s + w_off[nh * NTH + nt] + cw);
        if (ph == NPH - 1) {                        // stage kt+1 (own pieces) has landed
          if (D > 1 && pf) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPS_MIN) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute segment
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTH; ++nt)
            acc[mt][nh * NTH + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                wf[nt], xf[mt], (FIRST && PEEL0 && kh == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nh * NTH + nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      st = (st + 1 == NS) ? 0 : st + 1;
    };
    if (PEEL0) kstep_pp(0, std::true_type{});
    for (int kt = PEEL0 ? 1 : 0; kt < nk; ++kt) kstep_pp(kt, std::false_type{});
    if (!late) __builtin_amdgcn_s_barrier();      // the leading half catches the stagger up
  } else {
    if (NS != 1) issue(0, 0);
    if (NS == 3 && nk > 1) issue(1, 1);
    // one K step; `first` selects the C = 0 form of the MFMAs (saves zero-filling 16*NT registers)
    auto kstep = [&](const int kt, auto first) {
      constexpr bool FIRST = decltype(first)::value;
      if (NS == 1) {
        // single stage, 36-40 KiB LDS: 3 blocks per CU hide each other's loads and epilogues
        if (kt) __syncthreads();                // everyone is done reading the stage
        issue(kt, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else if (NS == 3) {
        // stage kt must have landed; the LDS-DMA of stage kt+1 (>= LPS_MIN ops per wave) may stay in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // raw barrier: no implicit vmcnt(0) drain
        if (kt + 2 < nk) issue(kt + 2, st == 0 ? 2 : st - 1);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // stage st landed; stage st^1 free again
        if (kt + 1 < nk) issue(kt + 1, st ^ 1);
      }
      if (EPI == HI3D_EPI_AFFINE && NT <= 5 && kt == 0 && tid < VSLOT / 16) {   // both vector slots have landed: fold them
        f32x4 b = *(const f32x4*)(vec_lds + tid * 16);            // (read again only after later barriers)
        const f32x4 g = *(const f32x4*)(vec_lds + VSLOT + tid * 16);
        b[0] += g[0]; b[1] += g[1]; b[2] += g[2]; b[3] += g[3];
        *(f32x4*)(vec_lds + tid * 16) = b;
      }
      if (EPI == HI3D_EPI_AFFINE && NT <= 5 && p.R1 && kt == pf_kt) {
  #pragma unroll
        for (int pass = 0; pass < RP; ++pass) fetch_residual(p.R1, p.ldr1, pass, r1v[pass]);
      }
      const char* s = smem + st * STAGE;
      __builtin_amdgcn_s_setprio(1);   // MFMA cluster first: measured +0.4 % on the dense shapes (hurts in attention)
  #pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        constexpr int NTH = (NT > 5) ? NT / 2 : NT;   // weight fragments held at a time
        bf16x8 xf[4];
        const int cx = ((kh * 4 + fg) ^ x_sw) << 4;
        const int cw = ((kh * 4 + fg) ^ w_sw) << 4;
  #pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8*)(s + x_off[mt] + cx);
  #pragma unroll
        for (int nh = 0; nh < NT / NTH; ++nh) {
          bf16x8 wf[NTH];
  #pragma unroll
          for (int nt = 0; nt < NTH; ++nt) wf[nt] = *(const bf16x8*)(s + w_off[nh * NTH + nt] + cw);
  #pragma unroll
          for (int mt = 0; mt < 4; ++mt)
  #pragma unroll
            for (int nt = 0; nt < NTH; ++nt)
              acc[mt][nh * NTH + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  wf[nt], xf[mt], (FIRST && PEEL0 && kh == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nh * NTH + nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (NS == 3) st = (st == 2) ? 0 : st + 1;
      else if (NS == 2) st ^= 1;
    };
    if (PEEL0) kstep(0, std::true_type{});
    for (int kt = PEEL0 ? 1 : 0; kt < nk; ++kt) kstep(kt, std::false_type{});
  }
  if (!PP && NT > 5 && EPI == HI3D_EPI_AFFINE && tid < VSLOT / 16) {   // wide tile: the fold inside the loop cost it 40 registers (spills)
    f32x4 b = *(const f32x4*)(vec_lds + tid * 16);                     // (slots landed before the first K step; read after the barrier below)
    const f32x4 g = *(const f32x4*)(vec_lds + VSLOT + tid * 16);
    b[0] += g[0]; b[1] += g[1]; b[2] += g[2]; b[3] += g[3];
    *(f32x4*)(vec_lds + tid * 16) = b;
  }
  if (PRECHUNK && NT > 5) init_chunks();

  // ---- epilogue.  The MFMA C layout gives a lane 4 columns of 16 different rows: stored
  // directly that is 64 scattered 8-byte requests per instruction, and the L2 request rate --
  // not bandwidth -- bounds every short-K GEMM.  So the fp32 tile goes through LDS (the
  // ring is free now), NPASS slabs of HR rows, and all global traffic (residual loads,
  // stores) is issued row-contiguous, 16 bytes per lane.
  if (EPI == HI3D_EPI_AFFINE && RES_EARLY && p.R2) fetch_residual(p.R2, p.ldr2, 0, r2v);
  // wide tile: residual slabs one store pass ahead, in two alternating register sets (the accumulator blocks
  // already staged free the registers); pass 0 is requested together with pass 1, once the first accumulator block
  // is in LDS (earlier, its registers would spill), and is the only one whose latency is exposed
  u32x4 rw1[2][CH];   // (R1 only: a second pair of sets for R2 spills; R2 -- AlphaBlender tails -- stays a load at the point of use)
  if (p.abl & 2) return;
  const int osz = p.out_fp32 ? 4 : 2;
  // ---- wide tiles (256 x 256 / 320, one block per CU: nothing else on the CU hides the epilogue): every WAVE stages
  // its own 64 x 16*NT tile, 16 rows at a time, through a private LDS slab -- no block barrier after the first one,
  // the LDS unit keeps a wave's writes and reads in order -- and issues its residual loads / stores row-contiguous,
  // 16 bytes per lane.  The block-wide form below cost 16 k cycles per tile before any store traffic (ablation:
  // profiles/r02c_gemm_epilogue_ablation.log), as much as 6 K steps: 8 barriers, each waiting for the slowest wave.
  constexpr bool WEPI = WEPI_OF(WM, NT, NS);
  if constexpr (WEPI) {
    constexpr int WOUT = (EPI == HI3D_EPI_GEGLU) ? 8 * NT : 16 * NT;   // output columns of a wave tile
    constexpr int WROW = WOUT * 4 + 16;                                  // slab row pitch (bytes)
    constexpr int WSLAB = 16 * WROW;
    static_assert(NW * WSLAB <= NS * STAGE, "per-wave epilogue slabs do not fit the LDS ring");
    constexpr int WCPR = WOUT / 8;                                       // 8-column chunks per row
    constexpr int WCH = (16 * WCPR + 63) / 64;                           // chunks per lane and pass
    char* const slab = smem + w * WSLAB;
    const int wcol0 = n0_out + wn * WOUT;                                // first output column of this wave
    const long wrow0 = (long)m0 + wm * 64;                               // first row of this wave
    // ---- GroupNorm statistics of this wave's 64 x 16*NT output block, from the accumulators (GemmParams.gn_part).  A lane's
    // 4*NT consecutive columns start at a multiple of 4*NT and so cover WHOLE groups of N/32 channels at every width the host
    // admits (4*NT = 40: 10 / 20 / 40 channels per group = 320 / 640 / 1280 wide; 4*NT = 32: 8 / 16 / 32): per lane NG
    // (sum, sum of squares) pairs over its 4 x 4*NT values, a 4-step butterfly over the 16 lanes that hold the other rows,
    // one 8*NG-byte store per (64-row block, column quarter).  Every (block, group) pair is written by exactly one lane of the
    // grid: plain stores, fixed summation order, bitwise reproducible.  The accumulators already carry bias + row vector.
    auto emit_gn_stats = [&]() {
      constexpr int LC = 4 * NT;
      auto gn_block = [&](auto ng_) {
        constexpr int NG = decltype(ng_)::value, CPG = LC / NG;
        float s[NG], q[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) { s[j] = 0.f; q[j] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = acc[mt][nt][r];
              s[(nt * 4 + r) / CPG] += v;
              q[(nt * 4 + r) / CPG] += v * v;
            }
        // sum over the 16 lanes of this lane's row group (fr): rotate-and-add inside the DPP row -- v_add_f32 with a row_ror
        // modifier, no LDS round trip (__shfl_xor lowers to ds_bpermute: 32 of them per tile cost more than the pass they save)
        auto row_sum = [](float v) {
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
          v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
          return v;
        };
#pragma unroll
        for (int j = 0; j < NG; ++j) { s[j] = row_sum(s[j]); q[j] = row_sum(q[j]); }
        if (fr == 0) {
          float* dst = p.gn_part + (wrow0 >> 6) * 64 + ((n0 + wn * 16 * NT + fg * LC) / CPG) * 2;
#pragma unroll
          for (int j = 0; j < NG; ++j) { dst[2 * j] = s[j]; dst[2 * j + 1] = q[j]; }
        }
      };
      const int cpg = p.N >> 5;
      if (cpg * 4 == LC) gn_block(std::integral_constant<int, 4>{});
      else if (cpg * 2 == LC) gn_block(std::integral_constant<int, 2>{});
      else gn_block(std::integral_constant<int, 1>{});
    };
    // tiles whose accumulators do not start from bias + row vector (the single-stage 128 x 128 tile): a launch that takes
    // statistics adds them HERE, in the accumulator layout, and its store loop skips them -- the same fp32 additions in the same order
    const bool bias_done = !BIAS_INIT && EPI == HI3D_EPI_AFFINE && p.gn_part != nullptr;
    if (bias_done) {
      __syncthreads();                             // (the fold of bias + row vector into vec_lds happened inside the K loop)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 b = *(const f32x4*)(vec_lds + (wn * 16 * NT + fg * 4 * NT + nt * 4) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { acc[mt][nt][0] += b[0]; acc[mt][nt][1] += b[1]; acc[mt][nt][2] += b[2]; acc[mt][nt][3] += b[3]; }
      }
    }
    if (EPI == HI3D_EPI_AFFINE && p.gn_part && !p.gn_post) emit_gn_stats();
    const __amdgpu_buffer_rsrc_t rsOw =
        __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (PHASE ? (long)wcol0 : ((long)ks * p.M * p.ldo + wrow0 * p.ldo + wcol0)) * osz, 0, 0x7fffffff, 0x00020000);
    // chunk i of a pass = lane + 64 i: slab row / column, the same in all four passes -- its LDS, bias, output and
    // residual offsets are computed once per tile (the store loop was VALU-bound on this index arithmetic: ~80
    // instructions per chunk, 20 chunks per wave; the accumulators leave ~30 registers for it)
    int w_row[WCH], w_col[WCH], w_lds[WCH];
    unsigned w_out[WCH], w_r1[WCH], w_r2[WCH];
    const int rows_left = p.M - (int)wrow0;                   // rows of this wave's 64 that exist
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = lane + 64 * i;
      w_row[i] = c / WCPR;
      w_col[i] = (c - w_row[i] * WCPR) * 8;
      const bool in = c < 16 * WCPR && wcol0 + w_col[i] < N_out;
      w_lds[i] = in ? w_row[i] * WROW + w_col[i] * 4 : 0;      // (void chunks read slab bytes 0..31 and store nothing)
      w_out[i] = in ? (unsigned)((w_row[i] * (PHASE ? 2 * p.ldo : p.ldo) + w_col[i]) * osz) : INV;
      w_r1[i] = in ? (unsigned)((w_row[i] * p.ldr1 + w_col[i]) * 2) : INV;
      w_r2[i] = in ? (unsigned)((w_row[i] * p.ldr2 + w_col[i]) * 2) : INV;
    }
    const __amdgpu_buffer_rsrc_t rsR1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.R1 ? p.R1 : (const unsigned short*)p.W) + wrow0 * p.ldr1 + wcol0), 0, p.R1 ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((p.R2 ? p.R2 : (const unsigned short*)p.W) + wrow0 * p.ldr2 + wcol0), 0, p.R2 ? 0x7fffffff : 0, 0x00020000);
    auto wc_ok = [&](int i, int mt) { return w_row[i] + mt * 16 < rows_left; };
    auto wfetch1 = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned (&off)[WCH], int ldr, int mt, int i) -> u32x4 {
      const int so = mt * 16 * ldr * 2;
      const bool has8 = wcol0 + w_col[i] + 8 <= N_out;
      const unsigned vo = wc_ok(i, mt) ? off[i] : INV;
      if (has8 && p.vec8) return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
      const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);   // 8-byte pieces: narrow or unaligned rows
      const u32x2 hi2 = __builtin_amdgcn_raw_buffer_load_b64(rs, has8 ? vo + 8 : INV, so, 0);
      return u32x4{lo[0], lo[1], hi2[0], hi2[1]};
    };
    auto wfetch = [&](int mt, u32x4 (&dst)[WCH]) {
#pragma unroll
      for (int i = 0; i < WCH; ++i) dst[i] = wfetch1(rsR1, w_r1, p.ldr1, mt, i);
    };
    const bool wfast = !p.out_fp32 && p.vec8 && rows_left >= 64 && wcol0 + WOUT <= N_out && (ugrp || !(p.rowvec || p.a1 || p.a2));
    u32x4 q1[2][WCH];        // (R2 -- AlphaBlender tails -- is loaded at the point of use: a third register set spills)
    __syncthreads();                               // every wave is done with the operand ring
    // ---- GroupNorm statistics of a tile that adds residual / blend terms AFTER its accumulators (GemmParams.gn_post; round 6): the
    // terms are added HERE, in the accumulator layout, so that the statistics come from the same registers and the same DPP
    // butterfly as above and the store loop below runs its residual-free path.  The R tile of 32 rows x WOUT columns (bf16, row-
    // contiguous: the chunk offsets of the store pass) goes to this wave's slab by LDS-DMA -- no registers -- and is read back
    // as the lane's 4*NT consecutive columns of its row, 8 at a time.  Same operations in the same order as the store loop
    // ((acc + R1) * a1 + a2 * R2, fp32).  (A first form took the sums in the store loop with ds_add_f32 into a pair table: the LDS
    // atomics cost 20 ms per step, profiles/r06b_ab_gn_post_atomics.log.)
    const bool gn_post = EPI == HI3D_EPI_AFFINE && p.gn_part && p.gn_post;
    if constexpr (EPI == HI3D_EPI_AFFINE) if (gn_post) {
      constexpr int RROW = WOUT * 2;                                       // bytes of a slab row of the bf16 R tile
      static_assert(32 * RROW <= WSLAB && 16 * RROW == WCH * 1024, "R tile: 16 rows = WCH LDS-DMA pieces, 32 rows fit the slab");
      auto add_tile = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned (&off)[WCH], int ldr, auto second) {
        constexpr bool R2T = decltype(second)::value;                    // false: acc += R1;  true: acc += ts2 * R2
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < WCH; ++i)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(slab + sub * 16 * RROW + i * 1024), 16, off[i],
                                                       (half * 2 + sub) * 16 * ldr * 2, 0, 0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const int mt = half * 2 + sub;
            const char* rrow = slab + (sub * 16 + fr) * RROW + fg * (8 * NT);
#pragma unroll
            for (int n2 = 0; n2 < NT / 2; ++n2) {
              const u32x4 r = *(const u32x4*)(rrow + n2 * 16);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float lo = __uint_as_float(r[j] << 16), hi = __uint_as_float(r[j] & 0xffff0000u);
                f32x4& a = acc[mt][2 * n2 + (j >> 1)];
                if (R2T) { a[(j & 1) * 2] += ts2 * lo; a[(j & 1) * 2 + 1] += ts2 * hi; }
                else { a[(j & 1) * 2] += lo; a[(j & 1) * 2 + 1] += hi; }
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the reads are done before the next pieces land
          __builtin_amdgcn_wave_barrier();
        }
      };
      static_assert(NT % 2 == 0, "R tile read-back: 8 columns at a time");
      if (p.R1) add_tile(rsR1, w_r1, p.ldr1, std::false_type{});
      if (p.a1) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { acc[mt][nt][0] *= ts1; acc[mt][nt][1] *= ts1; acc[mt][nt][2] *= ts1; acc[mt][nt][3] *= ts1; }
      }
      if (p.R2) add_tile(rsR2, w_r2, p.ldr2, std::true_type{});
      emit_gn_stats();
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      char* trow = slab + fr * WROW;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cw = fg * 4 * NT + nt * 4;       // wave-local column of acc[mt][nt][0]
        f32x4 v = acc[mt][nt];
        if (EPI == HI3D_EPI_GEGLU) {
          if (!BIAS_INIT) {
            const f32x4 b = *(const f32x4*)(vec_lds + (wn * 16 * NT + cw) * 4);
            v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
          }
          const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
          float2 o; o.x = v[0] * gl[0]; o.y = v[1] * gl[1];
          *(float2*)(trow + (cw >> 1) * 4) = o;
        } else {
          *(f32x4*)(trow + cw * 4) = v;
        }
      }
      if (EPI == HI3D_EPI_AFFINE && p.R1 && !gn_post) {        // residual slabs one pass ahead (pass 0 together with pass 1)
        __builtin_amdgcn_sched_barrier(0);
        if (mt == 0) wfetch(0, q1[0]);
        if (mt + 1 < 4) wfetch(mt + 1, q1[(mt + 1) & 1]);
      }
      __builtin_amdgcn_wave_barrier();
      // interior wave tiles of the usual call (bf16 out, 16-byte rows, one group per tile, no ragged chunk): no
      // per-lane predicate and no per-chunk mode branch -- ~25 VALU instructions per chunk instead of ~80
      if (wfast) {
        const int so = PHASE ? so_ph[mt] : mt * 16 * p.ldo * osz;
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
          const f32x4 lo = *(const f32x4*)(slab + w_lds[i]), hi = *(const f32x4*)(slab + w_lds[i] + 16);
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (EPI == HI3D_EPI_AFFINE) {
            if (!BIAS_INIT && !bias_done) {
              const char* bp = vec_lds + (wn * WOUT + w_col[i]) * 4;
              const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 16);
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
            }
            if (p.R1 && !gn_post) {
              const u32x4 r = q1[mt & 1][i];
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(r[j] << 16); v[2 * j + 1] += __uint_as_float(r[j] & 0xffff0000u); }
            }
            if (p.a1 && !gn_post) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= ts1;
            }
            if (p.R2 && !gn_post) {
              const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsR2, w_r2[i], mt * 16 * p.ldr2 * 2, 0);
#pragma unroll
              for (int j = 0; j < 4; ++j) { v[2 * j] += ts2 * __uint_as_float(r[j] << 16); v[2 * j + 1] += ts2 * __uint_as_float(r[j] & 0xffff0000u); }
            }
          }
          const u32x4 pk = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          __builtin_amdgcn_raw_buffer_store_b128(pk, rsOw, (p.abl & 1) ? INV : w_out[i], so, 0);
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
#pragma unroll
      for (int i = 0; i < WCH; ++i) {
        const int col = w_col[i], row = w_row[i];
        const bool ok = wc_ok(i, mt) && w_out[i] != INV;
        const int n = wcol0 + col;
        const bool has8 = n + 8 <= N_out;
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (w_out[i] != INV) { lo = *(const f32x4*)(slab + w_lds[i]); hi = *(const f32x4*)(slab + w_lds[i] + 16); }
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (EPI == HI3D_EPI_AFFINE) {
          if (!BIAS_INIT && !bias_done) {
            const char* bp = vec_lds + (wn * WOUT + col) * 4;
            const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
          }
          const long m = wrow0 + mt * 16 + row;
          const int grp = (ok && !ugrp && (p.rowvec || p.a1 || p.a2)) ? (int)(m / p.rpg) : 0;
          if (p.rowvec && !ugrp && ok) {
            const float* rv = p.rowvec + (long)grp * p.ldrv + n;
            const f32x4 r0 = *(const f32x4*)rv;
            v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
            if (has8) { const f32x4 r1 = *(const f32x4*)(rv + 4); v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3]; }
          }
          if (p.R1) {
            const u32x4 r = q1[mt & 1][i];
            v[0] += bf16_to_f32(r[0] & 0xffff); v[1] += bf16_to_f32(r[0] >> 16); v[2] += bf16_to_f32(r[1] & 0xffff); v[3] += bf16_to_f32(r[1] >> 16);
            v[4] += bf16_to_f32(r[2] & 0xffff); v[5] += bf16_to_f32(r[2] >> 16); v[6] += bf16_to_f32(r[3] & 0xffff); v[7] += bf16_to_f32(r[3] >> 16);
          }
          if (p.a1) { const float s1 = ugrp ? ts1 : p.a1[grp];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= s1; }
          if (p.R2) {
            const float s2 = ugrp ? ts2 : (p.a2 ? p.a2[grp] : 1.0f);
            const u32x4 r = wfetch1(rsR2, w_r2, p.ldr2, mt, i);
            v[0] += s2 * bf16_to_f32(r[0] & 0xffff); v[1] += s2 * bf16_to_f32(r[0] >> 16); v[2] += s2 * bf16_to_f32(r[1] & 0xffff); v[3] += s2 * bf16_to_f32(r[1] >> 16);
            v[4] += s2 * bf16_to_f32(r[2] & 0xffff); v[5] += s2 * bf16_to_f32(r[2] >> 16); v[6] += s2 * bf16_to_f32(r[3] & 0xffff); v[7] += s2 * bf16_to_f32(r[3] >> 16);
          }
        }
        const unsigned vo = (ok && !(p.abl & 1)) ? w_out[i] : INV;
        const int so = PHASE ? so_ph[mt] : mt * 16 * p.ldo * osz;
        if (p.out_fp32) {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rsOw, vo, so, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rsOw, has8 ? vo + 16 : INV, so, 0);
        } else {
          const u32x4 pk = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          if (has8 && p.vec8) {
            __builtin_amdgcn_raw_buffer_store_b128(pk, rsOw, vo, so, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[0], pk[1]}, rsOw, vo, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[2], pk[3]}, rsOw, has8 ? vo + 8 : INV, so, 0);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rsO =
      __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (((long)ks * p.M + m0) * p.ldo + n0_out) * osz, 0, 0x7fffffff, 0x00020000);
  __syncthreads();                                 // every wave is done with the operand ring
#pragma unroll
  for (int half = 0; half < NPASS; ++half) {
#pragma unroll
    for (int mh = 0; mh < MPP; ++mh) {
      const int mt = half * MPP + mh;
      char* trow = smem + (wm * 16 * MPP + mh * 16 + fr) * LROW;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cl = wn * 16 * NT + fg * 4 * NT + nt * 4;    // tile-local column of acc[mt][nt][0]
        f32x4 v = acc[mt][nt];
        if (EPI == HI3D_EPI_GEGLU) {
          const f32x4 b = *(const f32x4*)(vec_lds + cl * 4);
          v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
          const hi3d_f2 gl = gelu_erf_f2(hi3d_f2{v[2], v[3]});
          float2 o; o.x = v[0] * gl[0]; o.y = v[1] * gl[1];
          *(float2*)(trow + (cl >> 1) * 4) = o;
        } else {
          *(f32x4*)(trow + cl * 4) = v;
        }
      }
    }
    if (EPI == HI3D_EPI_AFFINE && !RES_EARLY) {
      __builtin_amdgcn_sched_barrier(0);           // not before the accumulator block above has left its registers
      if (half == 0 && p.R1) fetch_residual(p.R1, p.ldr1, 0, rw1[0]);
      if (half + 1 < NPASS && p.R1) fetch_residual(p.R1, p.ldr1, half + 1, rw1[(half + 1) & 1]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int col = chunk_col(i);
      const bool ok = chunk_ok(i, half);
      const int n = n0_out + col;
      const bool has8 = n + 8 <= N_out;
      const char* tp = smem + chunk_lds(i);
      f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
      if (chunk_in(i)) { lo = *(const f32x4*)tp; hi = *(const f32x4*)(tp + 16); }
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (EPI == HI3D_EPI_AFFINE) {
        {                                            // bias + the tile's row vector (zeros when absent)
          const f32x4 b0 = *(const f32x4*)(vec_lds + col * 4), b1 = *(const f32x4*)(vec_lds + col * 4 + 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        }
        const int m = m0 + chunk_row(i) + half * 16 * MPP;
        const int grp = (ok && !ugrp && (p.rowvec || p.a1 || p.a2)) ? m / p.rpg : 0;
        if (p.rowvec && !ugrp && ok) {               // tile spans groups (tiny grids only)
          const float* rv = p.rowvec + (long)grp * p.ldrv + n;
          const f32x4 r0 = *(const f32x4*)rv;
          v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
          if (has8) { const f32x4 r1 = *(const f32x4*)(rv + 4); v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3]; }
        }
        if (p.R1) {
          const u32x4 r = RES_EARLY ? r1v[half % RP][i] : rw1[half & 1][i];
          v[0] += bf16_to_f32(r[0] & 0xffff); v[1] += bf16_to_f32(r[0] >> 16); v[2] += bf16_to_f32(r[1] & 0xffff); v[3] += bf16_to_f32(r[1] >> 16);
          v[4] += bf16_to_f32(r[2] & 0xffff); v[5] += bf16_to_f32(r[2] >> 16); v[6] += bf16_to_f32(r[3] & 0xffff); v[7] += bf16_to_f32(r[3] >> 16);
        }
        if (p.a1) { const float s1 = ugrp ? ts1 : p.a1[grp];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= s1; }
        if (p.R2) {
          const float s2 = ugrp ? ts2 : (p.a2 ? p.a2[grp] : 1.0f);
          const u32x4 r = RES_EARLY ? r2v[i] : fetch_chunk(p.R2, p.ldr2, half, i);
          v[0] += s2 * bf16_to_f32(r[0] & 0xffff); v[1] += s2 * bf16_to_f32(r[0] >> 16); v[2] += s2 * bf16_to_f32(r[1] & 0xffff); v[3] += s2 * bf16_to_f32(r[1] >> 16);
          v[4] += s2 * bf16_to_f32(r[2] & 0xffff); v[5] += s2 * bf16_to_f32(r[2] >> 16); v[6] += s2 * bf16_to_f32(r[3] & 0xffff); v[7] += s2 * bf16_to_f32(r[3] >> 16);
        }
      }
      // stores through the descriptor: out-of-range chunks carry the INV offset and are dropped
      const unsigned vo = (ok && !(p.abl & 1)) ? (unsigned)((chunk_row(i) * p.ldo + col) * osz) : INV;
      const int so = half * 16 * MPP * p.ldo * osz;
      if (p.out_fp32) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rsO, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rsO, has8 ? vo + 16 : INV, so, 0);
      } else {
        const u32x4 pk = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (has8 && p.vec8) {
          __builtin_amdgcn_raw_buffer_store_b128(pk, rsO, vo, so, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[0], pk[1]}, rsO, vo, so, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[2], pk[3]}, rsO, has8 ? vo + 8 : INV, so, 0);
        }
      }
    }
    if (EPI == HI3D_EPI_AFFINE && RES_EARLY && p.R2 && half + 1 < NPASS) fetch_residual(p.R2, p.ldr2, half + 1, r2v);
    if (half + 1 < NPASS) __syncthreads();
  }
#endif
}

// debug capture (hi3d_debug_gemm_launch_info): the launch that hi3d_gemm_bf16 WOULD make -- kernel instantiation, geometry,
// packed kernel argument -- written here instead of launched.  Used by hi3d_hip/devtools/isa_stress.py, which re-assembles
// the kernel's device code with timing perturbations and launches it through the HIP module API.
struct GemmCapture { GemmParams p; int grid, block, smem, WM, NT, NS, AMODE, EPI, PP; };
thread_local int g_gn_fused = 0;     // set by hi3d_gemm_bf16: the launch it made (or described) fills GemmParams.gn_part
thread_local GemmCapture* g_capture = nullptr;

template <int WM, int NT, int NS, int AMODE, int EPI, bool PP = false>
int launch(const GemmParams& p, hipStream_t stream) {
  constexpr int smem = NS * (WM * 64 * BK * 2 + 32 * NT * BK * 2) + 2 * ((32 * NT * 4 + 1023) / 1024 * 1024);   // ring + bias / row-vector slots
  if (g_capture) {
    *g_capture = GemmCapture{p, p.nbm * p.nbn * (p.ksplit > 1 ? p.ksplit : 1), WM * 128, smem, WM, NT, NS, AMODE, EPI, PP ? 1 : 0};
    return HI3D_OK;
  }
  static bool attr_done[HI3D_MAX_DEVICES] = {};
  if (int rc = hi3d_raise_lds_limit((const void*)gemm_bf16_kernel<WM, NT, NS, AMODE, EPI, PP>, smem, attr_done)) return rc;
  hipLaunchKernelGGL((gemm_bf16_kernel<WM, NT, NS, AMODE, EPI, PP>), dim3(p.nbm * p.nbn * (p.ksplit > 1 ? p.ksplit : 1)), dim3(WM * 128), smem, stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

template <int WM, int NT, int NS, bool PP = false>
int dispatch(const GemmParams& p, int amode, int epi, hipStream_t s) {
  if (epi == HI3D_EPI_GEGLU) {
    if (amode != HI3D_A_DENSE) HI3D_FAIL(HI3D_ESHAPE, "gemm: GEGLU epilogue only with dense A");
    return launch<WM, NT, NS, HI3D_A_DENSE, HI3D_EPI_GEGLU, PP>(p, s);
  }
  switch (amode) {
    case HI3D_A_DENSE: return launch<WM, NT, NS, HI3D_A_DENSE, HI3D_EPI_AFFINE, PP>(p, s);
    case A_DENSE2:
      // (instantiated for the three tiles hi3d_gemm_bf16 restricts a two-source launch to)
      if constexpr ((WM == 2 && NS == 2 && !PP) || (WM == 4 && NT == 10 && NS == 2 && PP))
        return launch<WM, NT, NS, A_DENSE2, HI3D_EPI_AFFINE, PP>(p, s);
      else
        HI3D_FAIL(HI3D_EINVAL, "gemm: two-source A has no instantiation for this tile");
    case HI3D_A_CONV3X3: return p.up2x ? launch<WM, NT, NS, A_CONV3X3_UP2X, HI3D_EPI_AFFINE, PP>(p, s)
                                        : launch<WM, NT, NS, HI3D_A_CONV3X3, HI3D_EPI_AFFINE, PP>(p, s);
    case A_CONV3X3_PHASE:
      if constexpr (NT > 5 && WM == 4 && PP && NS == 2) return launch<WM, NT, NS, A_CONV3X3_PHASE, HI3D_EPI_AFFINE, PP>(p, s);
      else HI3D_FAIL(HI3D_EINVAL, "gemm: phase-placed output has no instantiation for this tile");
    case HI3D_A_CONVT3: return launch<WM, NT, NS, HI3D_A_CONVT3, HI3D_EPI_AFFINE, PP>(p, s);
  }
  HI3D_FAIL(HI3D_EINVAL, "gemm: bad amode");
}

// ---- split-K second pass: out = a1 * (sum_s part[s] + bias + rowvec[g] + R1) + a2 * R2 (the AFFINE epilogue of the kernel
// above, same order of operations), partials summed in split order -- bitwise reproducible.  One thread = 4 columns of a row.
struct CombineParams {
  const float* part; const float* bias; const float* rowvec; const unsigned short* R1; const unsigned short* R2;
  const float* a1; const float* a2; void* out;
  int M, N, S, ldo, ldr1, ldr2, ldrv, rpg, out_fp32;
};
__global__ __launch_bounds__(256) void splitk_combine_kernel(const CombineParams c) {
  const int nq = c.N >> 2;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)c.M * nq) return;
  const int m = (int)(id / nq), n = (int)(id - (long)m * nq) * 4;
  const long plane = (long)c.M * c.N;
  const float* pp = c.part + (long)m * c.N + n;
  f32x4 v = *(const f32x4*)pp;
  for (int s = 1; s < c.S; ++s) { const f32x4 t = *(const f32x4*)(pp + s * plane); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
  const int g = m / c.rpg;
  if (c.bias) { const f32x4 b = *(const f32x4*)(c.bias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
  if (c.rowvec) { const f32x4 r = *(const f32x4*)(c.rowvec + (long)g * c.ldrv + n); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
  if (c.R1) {
    const u32x2 r = *(const u32x2*)(c.R1 + (long)m * c.ldr1 + n);
    v[0] += bf16_to_f32(r[0] & 0xffff); v[1] += bf16_to_f32(r[0] >> 16); v[2] += bf16_to_f32(r[1] & 0xffff); v[3] += bf16_to_f32(r[1] >> 16);
  }
  if (c.a1) { const float s1 = c.a1[g]; v[0] *= s1; v[1] *= s1; v[2] *= s1; v[3] *= s1; }
  if (c.R2) {
    const float s2 = c.a2 ? c.a2[g] : 1.0f;
    const u32x2 r = *(const u32x2*)(c.R2 + (long)m * c.ldr2 + n);
    v[0] += s2 * bf16_to_f32(r[0] & 0xffff); v[1] += s2 * bf16_to_f32(r[0] >> 16); v[2] += s2 * bf16_to_f32(r[1] & 0xffff); v[3] += s2 * bf16_to_f32(r[1] >> 16);
  }
  if (c.out_fp32) *(f32x4*)((float*)c.out + (long)m * c.ldo + n) = v;
  else *(u32x2*)((unsigned short*)c.out + (long)m * c.ldo + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// Caller-provided scratch for the fp32 partial tiles; no workspace = no split-K.  A scratch buffer belongs to ONE stream:
// two split-K launches in flight on two streams of a device would write the same partial tiles.  So the table is
// keyed by (device, stream): hi3d_gemm_set_workspace_for_stream registers a buffer for a stream (also a capture stream: the
// pointer is baked into the captured graph), and the stream-less hi3d_gemm_set_workspace registers one that the FIRST stream
// to split with it claims -- a launch on any other stream finds no workspace and simply does not split (same result up to
// the fp32 summation order, never a race).
struct GemmWorkspace { void* ptr; long bytes; hipStream_t stream; bool any_stream, claimed; };
constexpr int WS_SLOTS = 64;                      // per device: slot 0 = the stream-less registration (8 ran out in one pytest process: round 4)
GemmWorkspace g_ws[HI3D_MAX_DEVICES][WS_SLOTS] = {};
std::mutex g_ws_mu;
// the workspace `stream` may use on `dev` (claims the stream-less one for it on first use), or nullptr
GemmWorkspace* ws_for(int dev, hipStream_t stream, bool claim) {
  if (dev < 0 || dev >= HI3D_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (int i = 1; i < WS_SLOTS; ++i)
    if (g_ws[dev][i].ptr && g_ws[dev][i].stream == stream) return &g_ws[dev][i];
  GemmWorkspace& w = g_ws[dev][0];
  if (!w.ptr) return nullptr;
  if (!w.claimed) { if (claim) { w.claimed = true; w.stream = stream; } return &w; }
  return w.stream == stream ? &w : nullptr;
}

// K split of a launch with `tiles` 128-row tiles: as many splits (<= 8) as keep tiles * S within the chip's 512 block slots
// (256 CUs x 2 resident blocks), each at least 8 K steps long and a whole number of `units` (K steps for dense A, 64-channel
// slabs for the conv gathers)
int splitk_choose(long tiles, int units, int ksteps_per_unit, long part_bytes, long ws_bytes) {
  int best = 1;
  for (int S = 2; S <= 8; ++S)
    if (units % S == 0 && (units / S) * ksteps_per_unit >= 8 && tiles * S <= 576 && part_bytes * S <= ws_bytes) best = S;
  return best;
}

}  // namespace

extern "C" int hi3d_gemm_set_workspace(void* ptr, int64_t bytes) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HI3D_MAX_DEVICES) HI3D_FAIL(HI3D_EINVAL, "gemm_set_workspace: no current device");
  if (bytes < 0 || (ptr == nullptr) != (bytes == 0) || ((uintptr_t)ptr & 15)) HI3D_FAIL(HI3D_EINVAL, "gemm_set_workspace: bad pointer / size");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_ws[dev][0] = GemmWorkspace{ptr, (long)bytes, nullptr, true, false};
  return HI3D_OK;
}

extern "C" int hi3d_gemm_set_workspace_for_stream(void* ptr, int64_t bytes, void* stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HI3D_MAX_DEVICES) HI3D_FAIL(HI3D_EINVAL, "gemm_set_workspace: no current device");
  if (bytes < 0 || (ptr == nullptr) != (bytes == 0) || ((uintptr_t)ptr & 15)) HI3D_FAIL(HI3D_EINVAL, "gemm_set_workspace: bad pointer / size");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  int slot = -1;
  for (int i = 1; i < WS_SLOTS; ++i) if (g_ws[dev][i].ptr && g_ws[dev][i].stream == (hipStream_t)stream) slot = i;   // replace / withdraw
  if (slot < 0) for (int i = 1; i < WS_SLOTS && slot < 0; ++i) if (!g_ws[dev][i].ptr) slot = i;
  if (!ptr) { if (slot > 0) g_ws[dev][slot] = GemmWorkspace{}; return HI3D_OK; }
  if (slot < 0) HI3D_FAIL(HI3D_EINVAL, "gemm_set_workspace_for_stream: all per-stream slots of this device are taken");
  g_ws[dev][slot] = GemmWorkspace{ptr, (long)bytes, (hipStream_t)stream, false, true};
  return HI3D_OK;
}

namespace {
// The dispatch's A/B switches (HI3D_GEMM_TILE_N / _NO_NARROW / _VARIANT / _ABL / _GN / _SPLITK, HI3D_GN_FUSED_OFF), read from the
// environment ONCE per process -- round 5 called getenv up to 7 times per launch, invisible under graph replay but paid ~750 times
// per rank-step on the eager clip-parallel path.  A caller that changes one of them inside a running process
// (the tests, tools/kbench.py sweeps) calls hi3d_gemm_reload_env() afterwards.
struct GemmEnv {
  int tile_n = 0;                   // HI3D_GEMM_TILE_N: 128 / 160, 0 = host heuristic
  bool no_narrow = false;           // HI3D_GEMM_NO_NARROW set: no 128 x 32 tile for N <= 32
  bool has_variant = false; int variant = 0;      // HI3D_GEMM_VARIANT
  int abl = 0;                      // HI3D_GEMM_ABL (ablation bits of the epilogue study)
  bool has_gn = false; int gn = 0;  // HI3D_GEMM_GN: column-group width of the tile raster
  bool gn_fused_off = false;        // HI3D_GN_FUSED_OFF set
  bool gn_post_off = false;         // HI3D_GN_POST=0: no statistics from residual / blend epilogues (round-5 behaviour; A/B switch)
  bool convt_skip = true;           // HI3D_CONVT_SKIP=0: Conv3d (3,1,1) walks all three taps in every block (A/B switch)
  int splitk = -1;                  // HI3D_GEMM_SPLITK: 0 = never, S > 1 = force S, -1 = heuristic
  void load() {
    *this = GemmEnv{};
    if (const char* e = getenv("HI3D_GEMM_TILE_N")) { const int t = atoi(e); if (t == 128 || t == 160) tile_n = t; }
    no_narrow = getenv("HI3D_GEMM_NO_NARROW") != nullptr;
    if (const char* e = getenv("HI3D_GEMM_VARIANT")) { has_variant = true; variant = atoi(e); }
    if (const char* e = getenv("HI3D_GEMM_ABL")) abl = atoi(e);
    if (const char* e = getenv("HI3D_GEMM_GN")) { has_gn = true; gn = atoi(e); }
    gn_fused_off = getenv("HI3D_GN_FUSED_OFF") != nullptr;
    if (const char* e = getenv("HI3D_GN_POST")) gn_post_off = atoi(e) == 0;
    if (const char* e = getenv("HI3D_CONVT_SKIP")) convt_skip = atoi(e) != 0;
    if (const char* e = getenv("HI3D_GEMM_SPLITK")) splitk = atoi(e);
  }
};
GemmEnv g_env_storage;
std::mutex g_env_mu;
const GemmEnv& gemm_env() {
  static const bool once = [] { std::lock_guard<std::mutex> lk(g_env_mu); g_env_storage.load(); return true; }();
  (void)once;
  return g_env_storage;
}

}  // namespace

extern "C" int hi3d_gemm_reload_env(void) {
  (void)gemm_env();                                  // (the first-use load happens-before this one)
  std::lock_guard<std::mutex> lk(g_env_mu);
  g_env_storage.load();
  return HI3D_OK;
}

extern "C" int hi3d_gemm_bf16(const hi3d_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->out) HI3D_FAIL(HI3D_EINVAL, "gemm: null pointer");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) HI3D_FAIL(HI3D_EINVAL, "gemm: non-positive size");
  if (d->K % 64) HI3D_FAIL(HI3D_ESHAPE, "gemm: K must be a multiple of 64");
  if (d->N % 4) HI3D_FAIL(HI3D_ESHAPE, "gemm: N must be a multiple of 4");
  if (d->rows_per_group < 1) HI3D_FAIL(HI3D_EINVAL, "gemm: rows_per_group < 1");
  if (d->epi != HI3D_EPI_AFFINE && d->epi != HI3D_EPI_GEGLU) HI3D_FAIL(HI3D_EINVAL, "gemm: bad epi");
  if (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->out) & 15) HI3D_FAIL(HI3D_EALIGN, "gemm: A/W/out not 16-byte aligned");
  const GemmEnv& env = gemm_env();
  GemmParams p;
  p.A = (const char*)d->A; p.W = (const char*)d->W; p.bias = d->bias; p.rowvec = d->rowvec;
  p.R1 = (const unsigned short*)d->R1; p.R2 = (const unsigned short*)d->R2; p.a1 = d->a1; p.a2 = d->a2;
  p.out = d->out; p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldo = d->ldo;
  p.ldr1 = d->ldr1; p.ldr2 = d->ldr2; p.ldrv = d->ldrv > 0 ? d->ldrv : d->N; p.ldw = d->ldw > 0 ? d->ldw : d->K; p.rpg = d->rows_per_group; p.out_fp32 = d->out_fp32;
  p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
  p.stride = d->stride; p.up2x = d->up2x; p.T = d->T; p.HW = d->HW; p.pad = d->pad_br_only ? 0 : 1;
  p.A2 = (const char*)d->A2; p.lda2 = d->lda2; p.K1 = d->K1;
  p.ntap = (d->amode == HI3D_A_CONV3X3 && d->conv_ntap > 0) ? d->conv_ntap : 9;
  p.taps = (d->amode == HI3D_A_CONV3X3 && d->conv_ntap > 0) ? d->conv_taps : 0u;
  p.wgs = d->w_group_stride;
  if (p.wgs) {
    if (p.wgs < 0 || p.wgs % 8) HI3D_FAIL(HI3D_EALIGN, "gemm: w_group_stride must be a non-negative multiple of 8 elements");
    if (d->rows_per_group % 256 || d->M % d->rows_per_group)
      HI3D_FAIL(HI3D_ESHAPE, "gemm: per-group weights need rows_per_group % 256 == 0 (the tallest tile) and M % rows_per_group == 0");
  }
  p.gn_part = nullptr; p.gn_post = 0; g_gn_fused = 0;
  p.tskip = env.convt_skip ? 1 : 0;
  const bool two = d->A2 != nullptr;
  if (two) {
    if (d->amode != HI3D_A_DENSE || d->epi != HI3D_EPI_AFFINE) HI3D_FAIL(HI3D_ESHAPE, "gemm: A2 (two-source A) needs dense A and the affine epilogue");
    if (d->K1 <= 0 || d->K1 >= d->K || d->K1 % 64) HI3D_FAIL(HI3D_ESHAPE, "gemm: K1 must be a multiple of 64 inside (0, K)");
    if (d->lda < d->K1 || d->lda % 8 || d->lda2 < d->K - d->K1 || d->lda2 % 8) HI3D_FAIL(HI3D_EALIGN, "gemm: lda < K1, lda2 < K - K1 or a pitch % 8 != 0");
    if ((uintptr_t)d->A2 & 15) HI3D_FAIL(HI3D_EALIGN, "gemm: A2 not 16-byte aligned");
  } else if (d->amode == HI3D_A_DENSE) {
    if (d->lda < d->K || (d->lda % 8)) HI3D_FAIL(HI3D_EALIGN, "gemm: lda < K or lda % 8 != 0");
  } else if (d->amode == HI3D_A_CONV3X3) {
    const int ntap = d->conv_ntap > 0 ? d->conv_ntap : 9;
    if (d->conv_ntap < 0 || d->conv_ntap > 8) HI3D_FAIL(HI3D_EINVAL, "conv3x3: conv_ntap must be 0 (all nine taps) or 1..8");
    for (int k = 0; k < d->conv_ntap; ++k)
      if (((d->conv_taps >> (4 * k)) & 15u) > 8u) HI3D_FAIL(HI3D_EINVAL, "conv3x3: conv_taps holds a tap index > 8");
    if (d->conv_ntap && (d->up2x || d->stride != 1)) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: a tap subset needs stride 1 and no up2x");
    if (d->Cin <= 0 || d->Cin % 64 || d->K != ntap * d->Cin) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: need Cin % 64 == 0 and K == 9*Cin (conv_ntap*Cin with a tap subset)");
    if (d->Hin <= 0 || d->Win <= 0 || d->Hout <= 0 || d->Wout <= 0) HI3D_FAIL(HI3D_EINVAL, "conv3x3: bad geometry");
    if (d->stride != 1 && d->stride != 2) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: stride must be 1 or 2");
    if (d->up2x && d->stride != 1) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: up2x needs stride 1");
    if (d->pad_br_only && d->up2x) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: pad_br_only cannot be combined with up2x");
    const int padt = d->pad_br_only ? 1 : 2;     // total padding per axis
    const int eh = d->up2x ? 2 * d->Hin : (d->Hin + padt - 3) / d->stride + 1;
    const int ew = d->up2x ? 2 * d->Win : (d->Win + padt - 3) / d->stride + 1;
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
  // 16-byte epilogue I/O needs 8-element alignment of every row; otherwise 8-byte pieces
  p.vec8 = (d->ldo % 8 == 0) && (!d->R1 || d->ldr1 % 8 == 0) && (!d->R2 || d->ldr2 % 8 == 0) &&
           (((uintptr_t)d->out | (uintptr_t)d->R1 | (uintptr_t)d->R2) % 16 == 0);
  int tile = d->tile_n;
  if (tile == 0) {
    // 160 suits every multiple of 320; otherwise pick the tile that wastes fewer columns
    const int w128 = (d->N + 127) / 128 * 128, w160 = (d->N + 159) / 160 * 160;
    tile = (w160 <= w128) ? 160 : 128;
    if (env.tile_n) tile = env.tile_n;
  }
  // N <= 32 (the 4-channel output convs: UNet out.2, VAE conv_out): a 128 x 32 tile -- those launches are all
  // A stream (M = 0.5-1 M rows, K = 1152-2880) and a 128-column tile spent 97 % of its MFMAs and W traffic on padding
  if (d->tile_n == 0 && d->N <= 32 && d->epi == HI3D_EPI_AFFINE && !env.no_narrow) tile = 32;
  if (tile != 128 && tile != 160 && tile != 32) HI3D_FAIL(HI3D_EINVAL, "gemm: tile_n must be 0, 32, 128 or 160");
  // tile height / ring depth: 0 = 128 rows, 2-stage ring, 2 blocks/CU (default: fastest at every
  // Hi3D shape once the loaders went to buffer addressing); 1 = 128 rows, 3 stages;
  // 2 = 256 rows, 8 waves, 3-stage ring with counted vmcnt, 1 block/CU;
  // 5 = 256 x 320 tile, 8 waves of 64 x 160 (half the L2->LDS bytes per FLOP), 1 block/CU.
  // 3 = 128 rows, single stage, 3 blocks/CU (round 6: the 128-column form is held to 128 registers -- FOUR blocks/CU -- and runs
  // the wide tiles' per-wave epilogue, statistics included).  Measured per shape (tools/kbench.py, MI355X):
  // the GEGLU GEMMs (erf epilogue, VALU heavy) gain 9-15 % from the third resident block while
  // K is short, and 12-16 % from the 256 x 320 tile (variant 5) at K >= 640; plain GEMMs gain
  // 3-7 % from the 256-row tile when both K and N are long; everything else, and every conv,
  // is fastest at 0.
  // 7 = 256 x 320 tile with the ping-pong K loop (2-stage ring), 8 = the same at 256 x 256 (N multiples of 256: the
  // VAE), 6 = 256 x 128 / 160 ping-pong, 3-stage ring (never the fastest; kept for A/B).  Measured (kbench.py
  // sweep, profiles/r02b_gemm_variant_sweep.log): the ping-pong wide tile wins +10-13 % on every conv3x3 /
  // strided / 2x-upsampling conv, +15-24 % on the QKV projections, +8-20 % on dense GEMMs with K >= 1920, as long
  // as the grid still has >= 1 tile per CU; it loses below 256 tiles (the 16^2 level, M = 8192) and on N that is
  // not close to a multiple of 320 (the VAE's 128 / 256 / 512 channels).
  int variant = 0;
  auto wide_fits = [&](int tn) {          // >= 256 tiles of 256 x tn and <= 7 % of the columns wasted
    const long nbn = (d->N + tn - 1) / tn;
    return d->tile_n == 0 && d->N >= tn && nbn * tn * 100 <= (long)d->N * 107 && ((long)(d->M + 255) / 256) * nbn >= 256;
  };
  if (d->amode == HI3D_A_DENSE) {
    if (d->epi == HI3D_EPI_GEGLU) variant = wide_fits(320) ? 7 : (d->K >= 640 && d->N % 320 == 0 && d->M >= 32768) ? 5 : (d->K >= 1280 ? 2 : 3);
    else if (wide_fits(320)) variant = 7;       // (since the residual slabs are requested a store pass ahead: also N = 320 / 640 with short K)
    else if (d->K >= 2560 || (d->K >= 1280 && d->N >= 2560)) variant = 2;
  } else if (wide_fits(320)) {
    variant = 7;
  } else if (d->amode == HI3D_A_CONV3X3 && d->N % 128 == 0 && d->N <= 512 && d->tile_n == 0 && d->stride == 1 && !d->up2x &&
             d->conv_ntap == 0 && ((long)d->M / 128) * (d->N / 128) >= 1024) {
    // round 6: the VAE's 128- / 256- / 512-channel convs at >= 1024 tiles of 128 x 128 (K = 1152 ... 4608) on the single-stage
    // tile, FOUR blocks per CU (128 registers, per-wave epilogue): 16 waves per CU hide a tile's start-up and store phase better
    // than the second ring stage or the 256 x 256 ping-pong tile did -- 128 -> 128 at 1024^2: 0.484 -> 0.377 ms with the skip,
    // 0.441 -> 0.330 without; 256 -> 256 at 512^2: 0.315 (256 x 256 tile) -> 0.290; 512 -> 512 at 256^2: 0.279 -> 0.267; the
    // 128^2 level (512 tiles) stays on the two-stage tile (0.080 vs 0.092)  (profiles/r06w_vae_variant_sweep_4blocks.log)
    variant = 3;
  } else if (d->N % 256 == 0 && wide_fits(256)) {
    variant = 8;                                  // (the up-sampling convs' phase launches: the placed store needs the wide tile)
  }
  if (env.has_variant) variant = env.variant;
  if (two && variant != 7) variant = 0;           // two-source A: built for the 128-row 2-stage tile and the 256 x 320 ping-pong tile
  // split-K: a long-K launch that leaves most of the chip's 512 block slots (256 CUs x 2 blocks of the 128-row tile) empty
  // -- M = 1-4 K rows: the 8x8 level of stage 1 ran at 370 TFLOP/s, every level does on the ranks of a clip-parallel job --
  // is cut along K into `ksplit` blocks per tile in ONE grid; fp32 partial tiles go to the caller's workspace
  // (hi3d_gemm_set_workspace) and splitk_combine_kernel applies the epilogue.  HI3D_GEMM_SPLITK=0 disables, =S forces S.
  int ksplit = 1;
  GemmWorkspace* ws = nullptr;
  if ((variant == 0 || variant == 2) && (tile == 128 || tile == 160) && d->epi == HI3D_EPI_AFFINE && d->K >= 2048 &&
      (long)d->M * d->N >= (1L << 18) && (((uintptr_t)d->out | (uintptr_t)d->R1 | (uintptr_t)d->R2) & 7) == 0) {
    int dev = -1;
    const int force = env.splitk;
    if (force != 0 && hipGetDevice(&dev) == hipSuccess && (ws = ws_for(dev, (hipStream_t)stream, g_capture == nullptr)) != nullptr) {
      const int taps = d->amode == HI3D_A_CONV3X3 ? p.ntap : d->amode == HI3D_A_CONVT3 ? 3 : 1;
      const int units = d->K / BK / taps;
      const long part = (long)d->M * d->N * 4, tiles = (long)((d->M + 127) / 128) * ((d->N + tile - 1) / tile);
      ksplit = splitk_choose(tiles, units, taps, part, ws->bytes);
      if (force > 1 && units % force == 0 && part * force <= ws->bytes) ksplit = force;
      if (ksplit > 1) variant = 0;                 // (the 256-row tile of variant 2 would halve the tile count again)
    }
  }
  if (variant == 5 || variant == 7) tile = 320;   // 256 x 320 tile: 8 waves of 64 x 160, one block per CU
  if (variant == 8) tile = 256;                   // 256 x 256 tile: 8 waves of 64 x 128
  const int bm = (variant == 2 || variant >= 5) ? 256 : 128;
  p.nbm = (d->M + bm - 1) / bm;
  p.nbn = (d->N + tile - 1) / tile;
  p.abl = env.abl;
  // column-group raster for weight matrices well beyond one XCD's L2 (4 MB): groups whose W slice is <= 2 MB, at least 2 wide
  // (dense / GEGLU and the conv gathers alike; HI3D_GEMM_GN overrides: 0 = off)
  p.gn = 0;
  {
    const long wbytes = (long)d->N * d->K * 2;
    if (wbytes > (6L << 20) && p.nbn > 2) {      // measured (profiles/r03c_gemm_gn_sweep.log): 3.3 MB matrices lose 5 %, >= 6.5 MB gain 4 - 17 %
      long g = (2L << 20) / ((long)tile * d->K * 2);
      p.gn = (int)(g < 2 ? 2 : g);
    }
    if (env.has_gn) p.gn = env.gn;
  }
  hipStream_t s = (hipStream_t)stream;
  int amode = two ? A_DENSE2 : d->amode;
  p.phase_c = 0;
  if (d->conv_phase) {
    const int ph = d->conv_phase - 1;
    if (ph < 0 || ph > 3) HI3D_FAIL(HI3D_EINVAL, "conv3x3: conv_phase must be 0 or 1..4");
    const bool ok = d->amode == HI3D_A_CONV3X3 && d->conv_ntap > 0 && !d->up2x && d->stride == 1 && d->Hout == d->Hin &&
                    d->Wout == d->Win && d->Win % 16 == 0 && d->M % 256 == 0 && d->N % tile == 0 && (variant == 7 || variant == 8) &&
                    ksplit == 1 && !d->R1 && !d->R2 && !d->rowvec && !d->a1 && !d->a2 && !d->out_fp32 && p.vec8 &&
                    4L * d->M * d->ldo * 2 < (1L << 31);
    if (!ok) HI3D_FAIL(HI3D_ESHAPE, "conv3x3: phase-placed output needs a tap subset, a wide tile (full tiles, >= 256 of them), Win % 16 == 0, "
                                     "a bias-only bf16 epilogue and a 2x image below 2 GiB");
    p.phase_c = (ph >> 1) * 2 * d->Win + (ph & 1);
    amode = A_CONV3X3_PHASE;
  }
  p.ksplit = 1; p.nk_split = d->K / BK;
  if (ksplit > 1) {
    int dev = -1;
    hipGetDevice(&dev);
    GemmParams q = p;
    q.ksplit = ksplit; q.nk_split = d->K / BK / ksplit;
    q.out = ws->ptr; q.out_fp32 = 1; q.ldo = d->N; q.vec8 = d->N % 8 == 0;
    q.bias = nullptr; q.rowvec = nullptr; q.R1 = nullptr; q.R2 = nullptr; q.a1 = nullptr; q.a2 = nullptr;
    const int rc = tile == 160 ? dispatch<2, 5, 2>(q, amode, d->epi, s) : dispatch<2, 4, 2>(q, amode, d->epi, s);
    if (rc || g_capture) return rc;
    CombineParams c{(const float*)ws->ptr, d->bias, d->rowvec, (const unsigned short*)d->R1, (const unsigned short*)d->R2,
                    d->a1, d->a2, d->out, d->M, d->N, ksplit, d->ldo, d->ldr1, d->ldr2, p.ldrv, d->rows_per_group, d->out_fp32};
    const long nthr = (long)d->M * (d->N / 4);
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, c);
    HI3D_LAUNCH_CHECK();
    return HI3D_OK;
  }
  // GroupNorm statistics of the output from the producer's accumulators (GemmParams.gn_part): the wide ping-pong tile, full
  // tiles only, nothing added after the accumulators (no residual / blend; a row vector only when it is per tile), and a
  // channel count whose groups are whole inside a lane's 4*NT columns
  p.gn_post = 0;
  // (round 6: also the single-stage 128 x 128 tile, which runs the same per-wave epilogue -- the VAE's 128-channel level)
  const bool wepi = variant == 7 || variant == 8 || (variant == 3 && tile == 128);
  if (d->gn_partial && wepi && ksplit == 1 && d->epi == HI3D_EPI_AFFINE &&
      !d->out_fp32 && d->M % bm == 0 && d->N % tile == 0 && d->N % 32 == 0 && (!d->rowvec || d->rows_per_group % bm == 0) &&
      ((uintptr_t)d->gn_partial & 7) == 0 && !env.gn_fused_off) {
    const int lc = tile == 320 ? 40 : tile == 256 ? 32 : 16, cpg = d->N / 32;
    const bool whole = cpg == lc || cpg * 2 == lc || cpg * 4 == lc;        // groups are whole inside a lane's 4*NT columns
    if (!d->R1 && !d->R2 && !d->a1 && !d->a2) {
      if (whole) { p.gn_part = d->gn_partial; g_gn_fused = 1; }
    } else if (whole && !env.gn_post_off && p.vec8 && ((!d->a1 && !d->a2) || d->rows_per_group % bm == 0)) {
      // residual / blend terms after the accumulators (round 6): they are added in the accumulator layout before the store loop
      // (GemmParams.gn_post), the statistics taken from the final values.  Full tiles, bf16 out, 16-byte rows, one row group per
      // tile: every wave tile takes the kernel's interior store path
      p.gn_part = d->gn_partial; p.gn_post = 1; g_gn_fused = 1;
    }
  }
  if (tile == 256) return dispatch<4, 8, 2, true>(p, amode, d->epi, s);
  if (tile == 320) return variant == 7 ? dispatch<4, 10, 2, true>(p, amode, d->epi, s) : dispatch<4, 10, 2>(p, amode, d->epi, s);
  if (tile == 32) {
    if (d->epi != HI3D_EPI_AFFINE) HI3D_FAIL(HI3D_ESHAPE, "gemm: the 32-column tile has no GEGLU form");
    if (two) HI3D_FAIL(HI3D_ESHAPE, "gemm: two-source A has no 32-column tile");
    p.nbm = (d->M + 127) / 128;
    switch (d->amode) {
      case HI3D_A_DENSE: return launch<2, 1, 2, HI3D_A_DENSE, HI3D_EPI_AFFINE>(p, s);
      case HI3D_A_CONV3X3: return d->up2x ? launch<2, 1, 2, A_CONV3X3_UP2X, HI3D_EPI_AFFINE>(p, s)
                                           : launch<2, 1, 2, HI3D_A_CONV3X3, HI3D_EPI_AFFINE>(p, s);
      case HI3D_A_CONVT3: return launch<2, 1, 2, HI3D_A_CONVT3, HI3D_EPI_AFFINE>(p, s);
    }
  }
  if (tile == 160) {
    if (variant == 6) return dispatch<4, 5, 3, true>(p, amode, d->epi, s);
    if (variant == 2) return dispatch<4, 5, 3>(p, amode, d->epi, s);
    if (variant == 1) return dispatch<2, 5, 3>(p, amode, d->epi, s);
    if (variant == 3) return dispatch<2, 5, 1>(p, amode, d->epi, s);
    return dispatch<2, 5, 2>(p, amode, d->epi, s);
  }
  if (variant == 6) return dispatch<4, 4, 3, true>(p, amode, d->epi, s);
  if (variant == 2) return dispatch<4, 4, 3>(p, amode, d->epi, s);
  if (variant == 1) return dispatch<2, 4, 3>(p, amode, d->epi, s);
  if (variant == 3) return dispatch<2, 4, 1>(p, amode, d->epi, s);
  return dispatch<2, 4, 2>(p, amode, d->epi, s);
}

// 1 when the LAST hi3d_gemm_bf16 launch of the calling host thread filled its descriptor's gn_partial, 0 when it did not (the
// consumer then runs the full hi3d_groupnorm_silu): the answer hi3d_gemm_gn_partial_supported gives BEFORE a launch, without the
// second pass through the dispatch (the wrappers probed, then launched -- every eager conv paid the host side twice)
extern "C" int hi3d_gemm_last_gn_fused(void) { return g_gn_fused; }

// debug aid: resident blocks per CU the runtime predicts for a kernel variant
extern "C" int hi3d_debug_gemm_occupancy(int wm, int nt, int ns) {
  int n = -1;
#define OCC(WM, NT, NS) if (wm == WM && nt == NT && ns == NS) { \
    constexpr int smem = NS * (WM * 64 * BK * 2 + 32 * NT * BK * 2) + 2 * ((32 * NT * 4 + 1023) / 1024 * 1024); \
    hipFuncSetAttribute((const void*)gemm_bf16_kernel<WM, NT, NS, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_bf16_kernel<WM, NT, NS, 0, 0>, WM * 128, smem); }
  OCC(2, 5, 2) OCC(2, 4, 2) OCC(4, 5, 3) OCC(4, 4, 3) OCC(2, 5, 3) OCC(2, 4, 3)
#undef OCC
  return n;
}

// debug aid: what hi3d_gemm_bf16(d) would launch.  params_out receives the kernel argument (struct GemmParams, at most 512
// bytes); info[0..9] = {bytes of the argument, grid, block, dynamic LDS bytes, WM, NT, NS, AMODE, EPI, PP} -- the template
// arguments name the instantiation gemm_bf16_kernel<WM, NT, NS, AMODE, EPI, PP>.  Nothing is launched.
extern "C" int hi3d_debug_gemm_launch_info_on(const hi3d_gemm_desc* d, void* stream, void* params_out, int32_t* info) {
  if (!params_out || !info) HI3D_FAIL(HI3D_EINVAL, "debug_gemm_launch_info: null pointer");
  static_assert(sizeof(GemmParams) <= 512, "GemmParams grew beyond the debug buffer");
  GemmCapture c;
  g_capture = &c;
  const int rc = hi3d_gemm_bf16(d, stream);        // (the stream decides which split-K scratch -- if any -- the launch would use)
  g_capture = nullptr;
  if (rc) return rc;
  memcpy(params_out, &c.p, sizeof(GemmParams));
  const int v[10] = {(int)sizeof(GemmParams), c.grid, c.block, c.smem, c.WM, c.NT, c.NS, c.AMODE, c.EPI, c.PP};
  for (int i = 0; i < 10; ++i) info[i] = v[i];
  return HI3D_OK;
}

extern "C" int hi3d_debug_gemm_launch_info(const hi3d_gemm_desc* d, void* params_out, int32_t* info) {
  return hi3d_debug_gemm_launch_info_on(d, nullptr, params_out, info);
}

// 1 when hi3d_gemm_bf16(d, stream) fills d->gn_partial (the consumer may then call hi3d_groupnorm_silu_from_partials), 0 when
// the launch it would make cannot (tile variant, tails, residual terms ...: the consumer runs the full hi3d_groupnorm_silu),
// negative on an invalid descriptor.  Nothing is launched.
extern "C" int hi3d_gemm_gn_partial_supported(const hi3d_gemm_desc* d, void* stream) {
  GemmCapture c;
  g_capture = &c;
  const int rc = hi3d_gemm_bf16(d, stream);
  g_capture = nullptr;
  return rc ? rc : g_gn_fused;
}

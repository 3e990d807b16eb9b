// Attention kernels for gfx950.
//
// (1) attn_d64_kernel -- flash-style forward for head_dim 64 (the spatial
//     self-attention of BasicTransformerBlock, sgm/modules/attention.py:332-336 /
//     427-439; S up to 16384 tokens).  One block = 256 query rows of one (batch, head),
//     4 waves x 64 rows (two 32-row MFMA blocks per wave share every K / V^T fragment).  K and V^T tiles of 64 keys are LDS-DMA'd (global_load_lds)
//     into a 2-stage ring with a source-side XOR swizzle (conflict-free ds_read_b128).
//     Both matrix products run on v_mfma_f32_32x32x16_bf16 in "swapped" form,
//         S^T[kv][q] = K . Q^T         O^T[d][q] = V^T . P^T
//     so that every lane owns ONE query row: softmax statistics, the running-max
//     rescale and the final 1/l are lane-local, and P (C-layout of the first product)
//     is already the B-operand layout of the second -- no LDS round trip, no cross-lane
//     permutes.  The key rows of each 32-key block are fed to the MFMA with bits 2 and
//     3 of the row index swapped, which makes the accumulator registers of a lane hold
//     8 *consecutive* keys per group -- exactly one 16-byte V^T fragment.
//
// (2) transpose_v_kernel -- V[(b,s), h*64+d] -> V^T[b][h][d][s] (zero padded to 64).
//
// (3) attn_temporal_kernel -- softmax attention over the frame axis (T <= 32) at every
//     pixel (sgm/modules/video_attention.py:114-125), reading the frame-major token
//     layout directly; VALU kernel on v_dot2c_f32_bf16 (0.05 % of the step FLOPs, its
//     cost is the q/k/v/o traffic).
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int swap_bits23(int i) {
  return (i & ~0xC) | ((i & 4) << 1) | ((i & 8) >> 1);
}

struct AttnParams {
  const char* q; const char* k; const char* vt; unsigned short* out;
  int B, H, Sq, Skv, ldq, ldk, ldvt, ldo, nqt;
  int force_exact;    // debug (HI3D_ATTN_FORCE_EXACT=1): every key tile runs the exact pre-pass
  float scale_log2;   // softmax scale * log2(e)
};

constexpr int KV_TILE = 64;
constexpr float SUM_MAX = 2048.0f;     // a key tile's 32-term row sum of P above this triggers the exact rescale
constexpr int ATT_STAGE = 2 * KV_TILE * 128;   // K tile 8 KiB + V^T tile 8 KiB

constexpr int QB = 2;                  // 32-row query blocks per wave
constexpr int Q_TILE = 4 * QB * 32;    // query rows per block

// PRE: q arrives pre-multiplied by scale*log2(e) (the UNet runtime folds it into the to_q weights, one
// rounding); otherwise the scale is applied to the fp32 score differences (one packed multiply per two scores).
// VROW: V arrives ROW-major ([b][s][ldv] with this head's 64 channels at column h*64, e.g. the v block of a fused qkv
// buffer; AttnParams.vt / ldvt then hold that pointer / row pitch) instead of as the pre-transposed V^T of
// hi3d_transpose_v.  The 64-key tile is LDS-DMA'd as two [64 keys][32 d] images with 64-byte rows and the V^T fragments
// of the second product are formed by ds_read_b64_tr_b16 (gfx950's transposing LDS read): a 16-lane group reads a
// [4 keys][16 d] block and each lane receives 4 consecutive keys of ONE d column, so two reads give the 8-key A fragment
// that the pre-transposed layout delivered with one ds_read_b128.  A 32-lane half touches 4 rows x 64 B = 256 contiguous
// bytes per read (all 64 banks once).  Removes the transpose pass (1.2 ms per stage-2 step) and its 2 x V bytes.
template <bool PRE, bool VROW>
__global__ __launch_bounds__(256, 2) void attn_d64_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ATT_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;

  // XCD-aware mapping: the query tiles of one (b, h) are consecutive logical ids and
  // stay on one XCD, so its K / V^T stream is fetched into a single L2.
  const int nblk = gridDim.x;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int h = bh % p.H, b = bh / p.H;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[row li][d = ks*16 + hi*8 ..+7]
  // of each of the wave's QB query blocks
  int qrow[QB]; bool qok[QB];
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = qt * Q_TILE + (w * QB + qb) * 32 + li;
    qok[qb] = qrow[qb] < p.Sq;
    const char* qp = p.q + (((long)b * p.Sq + (qok[qb] ? qrow[qb] : 0)) * p.ldq + h * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[qb][ks] = qok[qb] ? *(const bf16x8*)(qp + (ks * 16 + hi * 8) * 2) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // ---- LDS-DMA gather state: K tile rows (keys) and V^T tile rows (d), 8 KiB each =
  // 8 pieces of 1 KiB; wave w moves pieces 2w, 2w+1 of both.
  const int lrow = lane >> 3, lslot = lane & 7;
  const char* kbase = p.k + ((long)b * p.Skv * p.ldk + h * 64) * 2;
  const char* vbase = VROW ? p.vt + ((long)b * p.Skv * p.ldvt + h * 64) * 2
                           : p.vt + ((long)(b * p.H + h) * 64) * (long)p.ldvt * 2;
  // buffer-addressed LDS-DMA (as in gemm.hip): one descriptor per operand based at this (b, h), a 32-bit
  // per-lane byte offset fixed for the whole loop, a scalar offset walking the key tiles.  Key rows >= Skv lie
  // beyond num_records and read as zeros (no per-lane select, no 64-bit per-lane pointers).
#if __HIP_DEVICE_COMPILE__
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
      (void*)kbase, 0, (int)min((long)0x7fffffff, ((long)p.Skv - 1) * p.ldk * 2 + 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(   // (VROW: rows >= Skv read as zeros, like K)
      (void*)vbase, 0, VROW ? (int)min((long)0x7fffffff, ((long)p.Skv - 1) * p.ldvt * 2 + 128) : 0x7fffffff, 0x00020000);
  int k_vo[2], v_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 8 + lrow;               // LDS row 0..63
    const int kc = lslot ^ ((swap_bits23(r & 31) >> 1) & 7);   // read by MFMA row swap(r)
    const int vc = lslot ^ ((r >> 1) & 7);
    k_vo[i] = r * p.ldk * 2 + kc * 16;
    if (VROW) {
      // piece q = 2w + i of the V tile: image db = q >> 2 ([64 keys][32 d], 64-byte rows), key rows (q & 3) * 16 .. + 15;
      // lane -> key row lane >> 2, 16-byte chunk lane & 3 (the LDS image is lane-linear: row pitch 4 lanes x 16 B)
      const int q = w * 2 + i;
      v_vo[i] = ((q & 3) * 16 + (lane >> 2)) * p.ldvt * 2 + ((q >> 2) * 32 + (lane & 3) * 8) * 2;
    } else {
      v_vo[i] = r * p.ldvt * 2 + vc * 16;
    }
  }
  auto issue = [&](int j, int st) {
    char* sK = smem + st * ATT_STAGE;
    char* sV = sK + KV_TILE * 128;
    const int kv0 = j * KV_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (LDS_AS void*)(sK + (w * 2 + i) * 1024), 16, k_vo[i], kv0 * p.ldk * 2, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(sV + (w * 2 + i) * 1024), 16, v_vo[i],
                                               VROW ? kv0 * p.ldvt * 2 : kv0 * 2, 0, 0);   // (V^T: padded to 64, in range)
  };
#else
  auto issue = [&](int, int) {};
#endif

  // fragment read addresses: one LDS pointer per k-step (the XOR swizzle is lane-specific),
  // moved between the ring stages once per tile; key block / d block are ds_read offsets
  const int f_sw = (li >> 1) & 7;
  const char* k_ptr[4]; const char* v_ptr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    k_ptr[ks] = smem + swap_bits23(li) * 128 + (((ks * 2 + hi) ^ f_sw) << 4);
    v_ptr[ks] = smem + KV_TILE * 128 + li * 128 + (((ks * 2 + hi) ^ f_sw) << 4);
  }
  // VROW: one base for all 16 transposing reads of a tile (image db: + db * 4096, k-step: + ks * 1024, second 4 keys: + 256):
  // lane (g = lane >> 4, i = lane & 15) addresses key row hi * 8 + (i >> 2), columns (g & 1) * 16 + (i & 3) * 4 .. + 3
  const char* v_tr = smem + KV_TILE * 128 + (hi * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  int stage_step = ATT_STAGE;

  f32x16 o[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  // Softmax bookkeeping relative to a running reference point m_ref, kept in the domain of the RAW scores
  // q.k (PRE: q arrives pre-multiplied by scale*log2(e), so raw == log2 domain and kexp == 1; otherwise
  // P = exp2((q.k - m_ref) * kexp) with kexp = scale*log2(e) > 0 applied to the fp32 differences).  The common
  // pass does not compute the tile maximum: -m_ref is the C operand of each 32-key block's first score MFMA
  // (cneg: 16 equal registers per query block), so the matrix core delivers q.k - m_ref and P costs one
  // v_exp_f32 per score (plus one packed multiply for !PRE); only the row sums (needed anyway) are checked
  // at the end of the tile.  Softmax is invariant to the shift, so any m_ref that keeps P bounded is as good
  // as the true maximum.  When a row sum exceeds SUM_MAX (or is not finite), and on the first tile, an exact
  // pre-pass runs first: scores, true tile maximum, reference point / O / l moved in place -- and then the
  // same common pass.  One P register set, one PV block; all branches are wave-uniform.
  //
  // Keys beyond S_kv (a last tile with S_kv % 64 != 0) take the SAME instruction stream as every other tile:
  // their C-operand registers are set to -inf before the block's first score MFMA, so the matrix core
  // returns -inf, exp2 gives P = 0, and neither the maximum nor the row sum sees them.  (Round 2 peeled that
  // tile into a second code copy with a per-score select; that copy is gone)
  float m_ref[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m_ref[qb] = 0.f; l_run[qb] = 0.f; }
  f32x16 cneg[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[qb][r] = 0.f;
  const float kexp = PRE ? 1.0f : p.scale_log2;

  const int ntile = (p.Skv + KV_TILE - 1) / KV_TILE;
  const int nfull = p.Skv / KV_TILE;                   // j == nfull < ntile: the tile with keys >= S_kv
  issue(0, 0);
  for (int j = 0; j < ntile; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntile) issue(j + 1, (j & 1) ^ 1);
    const bool tail = (j == nfull);                    // wave-uniform; true at most once

    // q.k - m_ref for 32 keys x the wave's 64 query rows: every K fragment feeds QB MFMAs.
    // lane registers: sc[qb][r] = score of key  j*64 + kb*32 + (r>>3)*16 + hi*8 + (r&7)
    auto scores = [&](const int kb, f32x16 (&sc)[QB]) {
      if (tail) {
        // -inf into the C registers of keys >= S_kv, -m_ref into the others (rebuilt from m_ref on every
        // call: the two key blocks of the tile have different masks).  `rem` passes through an empty asm so
        // that the compare / select chain stays inside this block (hoisted, it ran on every tile).
        int rem = p.Skv - j * KV_TILE;
        asm volatile("" : "+s"(rem));
        const int lim = rem - kb * 32 - hi * 8;        // register r is beyond S_kv iff (r>>3)*16 + (r&7) >= lim
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cneg[qb][r] = ((r >> 3) * 16 + (r & 7) >= lim) ? -INFINITY : -m_ref[qb];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(k_ptr[ks] + kb * 32 * 128);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          sc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], ks == 0 ? cneg[qb] : sc[qb], 0, 0, 0);
      }
    };

    bf16x8 pf[QB][4];
    float psum[QB];
    bool exact = (j == 0) || p.force_exact;
    for (;;) {
      if (exact) {
        // exact pre-pass: move the rows' reference point to the true maximum seen so far
        float mx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) mx[qb] = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          f32x16 sc[QB];
          scores(kb, sc);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[qb] = fmaxf(mx[qb], sc[qb][r]);
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));   // the row's other 32 keys of this tile
          // t is relative to m_ref (finite: every tile has >= 1 valid key); never move down,
          // except to establish the reference on the first tile (O = l = 0 there)
          const float d = (j == 0) ? t : fmaxf(t, 0.f);
          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d * kexp);
          m_ref[qb] += d;
          l_run[qb] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) cneg[qb][r] -= d;     // in place (a fresh definition costs 32 moves per tile)
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
        }
      }
      // common pass
      hi3d_f2 ps[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) ps[qb] = hi3d_f2{0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 sc[QB];
        scores(kb, sc);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            union { bf16x8 v; unsigned int u[4]; } pk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              hi3d_f2 x = hi3d_f2{sc[qb][half * 8 + 2 * t], sc[qb][half * 8 + 2 * t + 1]};
              if (!PRE) x = x * hi3d_f2{kexp, kexp};             // one packed multiply per two scores
              hi3d_f2 e;
              e[0] = __builtin_amdgcn_exp2f(x[0]);
              e[1] = __builtin_amdgcn_exp2f(x[1]);
              ps[qb] += e;
              pk.u[t] = pack_bf16x2(e[0], e[1]);
            }
            pf[qb][kb * 2 + half] = pk.v;
          }
        }
      }
      bool ok = true;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { psum[qb] = ps[qb][0] + ps[qb][1]; ok = ok && (psum[qb] <= SUM_MAX); }   // NaN / inf fail too
      if (exact || __all(ok)) break;
      exact = true;                                    // rare: redo this tile through the pre-pass
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];

    // O^T += V^T P^T   (k-step ks covers keys ks*16 .. ks*16+15 of the tile); every V^T
    // fragment feeds QB MFMAs.  Keys >= S_kv: P = 0 and the V^T columns are the zero padding.
    if (VROW) {
#if __HIP_DEVICE_COMPILE__
      // The 16 transposing reads of the tile in inline assembly: the compiler's waitcnt pass puts `s_waitcnt vmcnt(0)` in front
      // of the first ds_read_b64_tr_b16 it sees while LDS-DMA is in flight (the builtin is not disambiguated against the DMA
      // target the way plain ds_read is) -- a wait for the NEXT tile's K / V requests in the middle of this tile.  All 16 are
      // issued, the first product block starts when its 8 have returned (LDS returns in order: lgkmcnt(8)), the second at 0;
      // the waits carry the registers as operands so that the MFMAs cannot be scheduled above them.
      bf16x4 vt[2][4][2];
      const unsigned va = (unsigned)(__UINTPTR_TYPE__)(LDS_AS char*)v_tr;
#define HI3D_TR(DB, KS, H) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[DB][KS][H]) : "v"(va), "n"((DB) * 4096 + (KS) * 1024 + (H) * 256))
      HI3D_TR(0, 0, 0); HI3D_TR(0, 0, 1); HI3D_TR(0, 1, 0); HI3D_TR(0, 1, 1); HI3D_TR(0, 2, 0); HI3D_TR(0, 2, 1); HI3D_TR(0, 3, 0); HI3D_TR(0, 3, 1);
      HI3D_TR(1, 0, 0); HI3D_TR(1, 0, 1); HI3D_TR(1, 1, 0); HI3D_TR(1, 1, 1); HI3D_TR(1, 2, 0); HI3D_TR(1, 2, 1); HI3D_TR(1, 3, 0); HI3D_TR(1, 3, 1);
#undef HI3D_TR
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        if (db == 0)
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]),
                                                "+v"(vt[0][2][0]), "+v"(vt[0][2][1]), "+v"(vt[0][3][0]), "+v"(vt[0][3][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]),
                                                "+v"(vt[1][2][0]), "+v"(vt[1][2][1]), "+v"(vt[1][3][0]), "+v"(vt[1][3][1]));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x4 v0 = vt[db][ks][0], v1 = vt[db][ks][1];
          const bf16x8 vf = bf16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], o[qb][db], 0, 0, 0);
        }
      }
#endif
    } else {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 vf = *(const bf16x8*)(v_ptr[ks] + db * 32 * 128);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], o[qb][db], 0, 0, 0);
        }
    }
    // flip the fragment pointers to the other ring stage
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { k_ptr[ks] += stage_step; v_ptr[ks] += stage_step; }
    v_tr += stage_step;
    stage_step = -stage_step;
  }

  // ---- finish: both half-waves hold partial row sums of the same query row
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
    if (qok[qb]) {
      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // C rows (r&3) + 8*(r>>2) + 4*hi  ->  d = db*32 + g*8 + hi*4 + (0..3)
          uint2 v;
          v.x = pack_bf16x2(o[qb][db][g * 4 + 0] * inv, o[qb][db][g * 4 + 1] * inv);
          v.y = pack_bf16x2(o[qb][db][g * 4 + 2] * inv, o[qb][db][g * 4 + 3] * inv);
          *(uint2*)(op + db * 32 + g * 8 + hi * 4) = v;
        }
    }
  }
}

// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_v_kernel(const unsigned short* __restrict__ v,
                                                          unsigned short* __restrict__ vt,
                                                          int H, int S, int S_pad, int ldv) {
  __shared__ unsigned short tile[64][66];   // [s][d], +2 pad
  const int st = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int s0 = st * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;            // 512 chunks of 8 bf16
    const int sr = c >> 3, dc = (c & 7) * 8;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (s0 + sr < S) q = *(const uint4*)(v + ((long)b * S + s0 + sr) * ldv + h * 64 + dc);
    const unsigned int u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[sr][dc + 2 * j] = (unsigned short)(u[j] & 0xffff);
      tile[sr][dc + 2 * j + 1] = (unsigned short)(u[j] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
    const int d = c >> 3, sc = (c & 7) * 8;
    unsigned int u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      u[j] = (unsigned int)tile[sc + 2 * j][d] | ((unsigned int)tile[sc + 2 * j + 1][d] << 16);
    *(uint4*)(vt + (((long)b * H + h) * 64 + d) * S_pad + s0 + sc) = make_uint4(u[0], u[1], u[2], u[3]);
  }
}

// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
__device__ __forceinline__ float dot2_bf16(unsigned int a, unsigned int b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, a), __builtin_bit_cast(v2bf, b), c, false);
}

// block = PB pixels x TP frames (PB*TP = 256) of one (batch, head); thread = (pixel, query frame)
template <int TP>
__global__ __launch_bounds__(256) void attn_temporal_kernel(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k,
    const unsigned short* __restrict__ v, unsigned short* __restrict__ out,
    int T, int S, int ld, int ldo, float scale_log2, int hfast) {
  constexpr int PB = 256 / TP;
  constexpr int KPIX = TP * 128 + 16;            // bytes of keys per pixel (+16: spreads pixels over banks)
  constexpr int VROW = TP * 2;                   // bytes per (pixel, d) row of V^T
  constexpr int VPIX = 64 * VROW + 16;
  __shared__ __attribute__((aligned(16))) char sK[PB * KPIX];
  __shared__ __attribute__((aligned(16))) char sV[PB * VPIX];
  const int tid = threadIdx.x;
  // hfast: heads on the fastest grid axis -- the blocks in flight at one time then read ALL heads' 128-byte pieces of the
  // same token rows (one contiguous 3C-wide row per pixel and frame) instead of one piece out of every row
  const int s0 = (hfast ? blockIdx.y : blockIdx.x) * PB, h = hfast ? blockIdx.x : blockIdx.y, b = blockIdx.z;

  // ---- stage K rows and transposed V: 8 chunks of 16 B per (pixel, frame).  PB * TP * 8 / 256 = 8 chunks per
  // thread: all 16 loads are issued before the first LDS write (the rolled loop had two loads in flight per thread
  // and ran the kernel at 3.0 TB/s)
  constexpr int NCH = PB * TP * 8 / 256;
  uint4 kq[NCH], vq[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    const int ch = c & 7, row = c >> 3;
    const int t = row % TP, px = row / TP;
    kq[i] = make_uint4(0, 0, 0, 0); vq[i] = make_uint4(0, 0, 0, 0);
    if (t < T && s0 + px < S) {
      const long off = (((long)b * T + t) * S + s0 + px) * ld + h * 64 + ch * 8;
      kq[i] = *(const uint4*)(k + off);
      vq[i] = *(const uint4*)(v + off);
    }
  }
  // this thread's query row is requested now, with the K / V loads still in flight (it used to be loaded after the barrier,
  // a second exposed HBM round trip per block)
  const int tq = tid % TP, px = tid / TP;
  const bool active = tq < T && s0 + px < S;
  unsigned int qr[32];
  {
    const long qoff = (((long)b * T + (active ? tq : 0)) * S + (active ? s0 + px : s0)) * ld + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 t4 = *(const uint4*)(q + qoff + c * 8);
      qr[c * 4] = t4.x; qr[c * 4 + 1] = t4.y; qr[c * 4 + 2] = t4.z; qr[c * 4 + 3] = t4.w;
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    const int ch = c & 7, row = c >> 3;
    const int t = row % TP, px = row / TP;
    *(uint4*)(sK + px * KPIX + t * 128 + ch * 16) = kq[i];
    const unsigned int u[4] = {vq[i].x, vq[i].y, vq[i].z, vq[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(unsigned short*)(sV + px * VPIX + (ch * 8 + 2 * j) * VROW + t * 2) = (unsigned short)(u[j] & 0xffff);
      *(unsigned short*)(sV + px * VPIX + (ch * 8 + 2 * j + 1) * VROW + t * 2) = (unsigned short)(u[j] >> 16);
    }
  }
  __syncthreads();

  if (!active) return;
  float sc[TP];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    const char* kr = sK + px * KPIX + t * 128;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 k4 = *(const uint4*)(kr + c * 16);
      a0 = dot2_bf16(qr[c * 4], k4.x, a0); a1 = dot2_bf16(qr[c * 4 + 1], k4.y, a1);
      a0 = dot2_bf16(qr[c * 4 + 2], k4.z, a0); a1 = dot2_bf16(qr[c * 4 + 3], k4.w, a1);
    }
    sc[t] = (t < T) ? (a0 + a1) * scale_log2 : -INFINITY;
    mx = fmaxf(mx, sc[t]);
  }
  float l = 0.f;
  unsigned int pp[TP / 2];
#pragma unroll
  for (int t = 0; t < TP; t += 2) {
    const float e0 = exp2f(sc[t] - mx), e1 = exp2f(sc[t + 1] - mx);
    l += e0 + e1;
    pp[t / 2] = pack_bf16x2(e0, e1);
  }
  const float inv = 1.0f / l;
  unsigned short* op = out + (((long)b * T + tq) * S + s0 + px) * ldo + h * 64;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float od[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* vr = sV + px * VPIX + (c * 8 + j) * VROW;
      float a = 0.f;
#pragma unroll
      for (int t4 = 0; t4 < TP / 8; ++t4) {
        const uint4 vv = *(const uint4*)(vr + t4 * 16);
        a = dot2_bf16(pp[t4 * 4], vv.x, a); a = dot2_bf16(pp[t4 * 4 + 1], vv.y, a);
        a = dot2_bf16(pp[t4 * 4 + 2], vv.z, a); a = dot2_bf16(pp[t4 * 4 + 3], vv.w, a);
      }
      od[j] = a * inv;
    }
    *(uint4*)(op + c * 8) = make_uint4(pack_bf16x2(od[0], od[1]), pack_bf16x2(od[2], od[3]),
                                       pack_bf16x2(od[4], od[5]), pack_bf16x2(od[6], od[7]));
  }
}


// ---- temporal attention on the matrix core, T <= 16 (round 4).  One WAVE per (clip b, pixel, head): the whole problem is
// S^T = K Q^T (16 x 16 x 64: two v_mfma_f32_16x16x32_bf16), a softmax over <= 16 keys, O^T = V^T P^T (four
// v_mfma_f32_16x16x16_bf16).  Layouts are chosen so that nothing is staged that does not have to be:
//   * Q and K rows are loaded from the frame-major token tensor STRAIGHT into MFMA operand registers: lane (t = lane & 15,
//     g = lane >> 4) holds 8 channels (d = ks * 32 + g * 8 ..) of frame t -- one 16-byte load per k-step and operand;
//   * the swapped product leaves acc[r] = S^T[key 4g + r][query = lane & 15]: the softmax of a query is 4 registers x the 4
//     lanes {q, q + 16, q + 32, q + 48} (two shuffles), and the packed P is ALREADY the B operand of the K = 16 MFMA;
//   * V goes through LDS once, as four [16 keys][4 x 4 channels] images that are lane-linear for the writer (4 ds_write_b64
//     per lane at j * 512 + lane * 8) AND for the transposing reader (ds_read_b64_tr_b16 at the same addresses): image j
//     holds channels 16 a + 4 j + (0..3), a = 0..3, so MFMA j's output row i means channel 16 (i >> 2) + 4 j + (i & 3) and a
//     lane ends up with 16 CONSECUTIVE channels of one query frame -- two 16-byte stores, 128 contiguous bytes per 4 lanes.
// The round-1..3 kernel (below, still used for 16 < T <= 32) transposed V with 64 ds_write_b16 per thread (8-16-way bank
// conflicts) and ran every product on v_dot2: 3.5 ms per stage-2 step at 3.4 TB/s; this form is bound by its q/k/v/o traffic.
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
__global__ __launch_bounds__(256) void attn_temporal_mfma_kernel(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k,
    const unsigned short* __restrict__ v, unsigned short* __restrict__ out,
    int B, int T, int S, int H, int ld, int ldo, float scale_log2) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) char sV[4 * 2048];     // one 2 KiB V image set per wave
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const myV = sV + w * 2048;
  const int t16 = lane & 15, g = lane >> 4;
  const unsigned nitem = (unsigned)B * S * H;      // (< 2^31: checked by the host)
  const unsigned stride = gridDim.x * 4;
  // operand-row validity (frames >= T do not exist: zero rows; their keys are masked, their query columns are not stored)
  const bool t_ok = t16 < T;
  const int vkey = lane >> 2;                     // V loader: key row of this lane, channels (lane & 3) * 16 .. + 15
  const bool v_ok = vkey < T;
  // (the per-lane offsets below stay under 2^31: the host checks 16 frames x S x ld x 2 bytes)
  // per-lane byte offsets inside an item's (clip, pixel, head) window: loop-invariant; rows that do not exist carry an offset
  // beyond the descriptor's range and read as zeros (no per-load branch)
  constexpr unsigned INV = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const long frame_pitch = (long)S * ld * 2;                                  // bytes between frames of one pixel
  const unsigned off_qk = t_ok ? (unsigned)(t16 * frame_pitch + g * 16) : INV;
  const unsigned off_v = v_ok ? (unsigned)(vkey * frame_pitch + (lane & 3) * 32) : INV;
  const unsigned off_o = t_ok ? (unsigned)(t16 * (long)S * ldo * 2 + g * 32) : INV;
  for (unsigned it = blockIdx.x * 4 + w; it < nitem; it += stride) {
    const unsigned bp = it / (unsigned)H, h = it - bp * H;
    const unsigned b = bp / (unsigned)S, px = bp - b * S;
    const long base = (((long)b * T * S + px) * ld + h * 64) * 2;            // wave-uniform
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)q + base), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)k + base), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)v + base), 0, 0x7fffffff, 0x00020000);
    // ---- loads (all issued before the first use)
    bf16x8 qf[2], kf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, off_qk, ks * 64, 0));
      kf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsK, off_qk, ks * 64, 0));
    }
    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsV, off_v, 0, 0);      // channels 16 a .. + 7      (a = lane & 3)
    const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rsV, off_v, 16, 0);     // channels 16 a + 8 .. + 15
    // ---- S^T = K Q^T
    f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], sc, 0, 0, 0);
    // ---- V images: image j <- channels 16 a + 4 j .. + 3 of key `vkey` (8 bytes) at j * 512 + lane * 8
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    *(u32x2*)(myV + 0 * 512 + lane * 8) = u32x2{v0[0], v0[1]};
    *(u32x2*)(myV + 1 * 512 + lane * 8) = u32x2{v0[2], v0[3]};
    *(u32x2*)(myV + 2 * 512 + lane * 8) = u32x2{v1[0], v1[1]};
    *(u32x2*)(myV + 3 * 512 + lane * 8) = u32x2{v1[2], v1[3]};
    // ---- softmax over the keys of query t16: registers r (keys 4 g + r) x lanes {t16 + 16 g'}
    float s4[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s4[r] = (4 * g + r < T) ? sc[r] * scale_log2 : -INFINITY;
      mx = fmaxf(mx, s4[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float e[4], l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(s4[r] - mx); l += e[r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l);
    union { bf16x4_t v; unsigned int u[2]; } pk;
    pk.u[0] = pack_bf16x2(e[0], e[1]); pk.u[1] = pack_bf16x2(e[2], e[3]);
    // ---- O^T = V^T P^T: MFMA j, output row i = 4 g + r  <->  channel 16 g + 4 j + r of query t16
    float o[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x4_t vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4_t*)(myV + j * 512 + lane * 8));
      const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pk.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[j * 4 + r] = acc[r] * inv;
    }
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)out + (((long)b * T * S + px) * ldo + h * 64) * 2), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])},
                                           rsO, off_o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(o[8], o[9]), pack_bf16x2(o[10], o[11]), pack_bf16x2(o[12], o[13]), pack_bf16x2(o[14], o[15])},
                                           rsO, off_o, 16, 0);
  }
#endif
}

// ---- the same for 16 < T <= 32 (BASELINE config 4: 32 views), round 5.  Still one wave per (clip, pixel, head); the frames are
// cut into two blocks of 16 on BOTH sides: S^T is 2 x 2 blocks of 16 keys x 16 queries (two v_mfma_f32_16x16x32_bf16 each, K =
// the 64 channels), the softmax of a query runs over 2 blocks x 4 registers x 4 lanes, and O^T[16 channels][16 queries] sums
// BOTH key blocks in ONE v_mfma_f32_16x16x32_bf16 whose contraction index k = 8 g + 4 kb + r stands for key 16 kb + 4 g + r:
// on the B side that is exactly the two packed-P register pairs of a lane (block 0 then block 1), on the A side two transposing
// LDS reads of the same V image position in the two key blocks' image sets.  Everything else -- operand loads straight from
// the frame-major tensor, lane-linear V images, 16 consecutive output channels per lane -- is the T <= 16 kernel's layout with
// a block index added (tests/test_kernels_gpu.py runs both kernels on the same T <= 16 inputs).
// Rounds 1-4 ran this case on the VALU kernel above: 8.35 ms per 32-view step at 2.8 TB/s against 4.3-5.3 TB/s for this form.
__global__ __launch_bounds__(256) void attn_temporal_mfma32_kernel(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k,
    const unsigned short* __restrict__ v, unsigned short* __restrict__ out,
    int B, int T, int S, int H, int ld, int ldo, float scale_log2) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) char sV[4 * 4096];     // two 2 KiB V image sets (key blocks 0 / 1) per wave
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const myV = sV + w * 4096;
  const int t16 = lane & 15, g = lane >> 4;
  const unsigned nitem = (unsigned)B * S * H;      // (< 2^31: checked by the host)
  const unsigned stride = gridDim.x * 4;
  const int vkey = lane >> 2;                     // V loader: key row (within a block) of this lane, channels (lane & 3) * 16 .. + 15
  constexpr unsigned INV = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const long frame_pitch = (long)S * ld * 2;                                  // bytes between frames of one pixel
  // per-lane byte offsets of frame block fb (frames 16 fb + ..): loop-invariant; rows that do not exist (frame >= T) carry an
  // offset beyond the descriptor's range and read as zeros / are not stored (the host checks 32 frames x S x ld x 2 < 2^31)
  unsigned off_qk[2], off_v[2], off_o[2];
#pragma unroll
  for (int fb = 0; fb < 2; ++fb) {
    off_qk[fb] = (16 * fb + t16 < T) ? (unsigned)((16 * fb + t16) * frame_pitch + g * 16) : INV;
    off_v[fb] = (16 * fb + vkey < T) ? (unsigned)((16 * fb + vkey) * frame_pitch + (lane & 3) * 32) : INV;
    off_o[fb] = (16 * fb + t16 < T) ? (unsigned)((16 * fb + t16) * (long)S * ldo * 2 + g * 32) : INV;
  }
  for (unsigned it = blockIdx.x * 4 + w; it < nitem; it += stride) {
    const unsigned bp = it / (unsigned)H, h = it - bp * H;
    const unsigned b = bp / (unsigned)S, px = bp - b * S;
    const long base = (((long)b * T * S + px) * ld + h * 64) * 2;            // wave-uniform
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)q + base), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)k + base), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)v + base), 0, 0x7fffffff, 0x00020000);
    // ---- loads: 12 x 16 bytes per lane, all issued before the first use
    bf16x8 qf[2][2], kf[2][2];
    u32x4 vv[2][2];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[fb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsK, off_qk[fb], ks * 64, 0));
        qf[fb][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, off_qk[fb], ks * 64, 0));
      }
    }
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      vv[fb][0] = __builtin_amdgcn_raw_buffer_load_b128(rsV, off_v[fb], 0, 0);       // channels 16 a .. + 7      (a = lane & 3)
      vv[fb][1] = __builtin_amdgcn_raw_buffer_load_b128(rsV, off_v[fb], 16, 0);      // channels 16 a + 8 .. + 15
    }
    // ---- S^T blocks: sc[kb][qb][r] = S^T[key 16 kb + 4 g + r][query 16 qb + t16]
    f32x4 sc[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][ks], qf[qb][ks], a, 0, 0, 0);
        sc[kb][qb] = a;
      }
    // ---- V images of key block kb at kb * 2048: image j <- channels 16 a + 4 j .. + 3 of key `vkey` (8 bytes) at j * 512 + lane * 8
    // (the previous item's transposing reads of these addresses were consumed by its MFMAs, whose results its stores waited for)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      *(u32x2*)(myV + kb * 2048 + 0 * 512 + lane * 8) = u32x2{vv[kb][0][0], vv[kb][0][1]};
      *(u32x2*)(myV + kb * 2048 + 1 * 512 + lane * 8) = u32x2{vv[kb][0][2], vv[kb][0][3]};
      *(u32x2*)(myV + kb * 2048 + 2 * 512 + lane * 8) = u32x2{vv[kb][1][0], vv[kb][1][1]};
      *(u32x2*)(myV + kb * 2048 + 3 * 512 + lane * 8) = u32x2{vv[kb][1][2], vv[kb][1][3]};
    }
    // ---- softmax of query (qb, t16) over 2 key blocks x 4 registers x the 4 lanes {t16 + 16 g'}; P packed = the B operand
    bf16x8 pk[2];
    float inv[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float s8[8], mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s8[kb * 4 + r] = (16 * kb + 4 * g + r < T) ? sc[kb][qb][r] * scale_log2 : -INFINITY;
          mx = fmaxf(mx, s8[kb * 4 + r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float e[8], l = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { e[i] = __builtin_amdgcn_exp2f(s8[i] - mx); l += e[i]; }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      inv[qb] = __builtin_amdgcn_rcpf(l);
      union { bf16x8 v; unsigned int u[4]; } p8;
      p8.u[0] = pack_bf16x2(e[0], e[1]); p8.u[1] = pack_bf16x2(e[2], e[3]);       // k = 8 g + 0..3 : keys      4 g + r
      p8.u[2] = pack_bf16x2(e[4], e[5]); p8.u[3] = pack_bf16x2(e[6], e[7]);       // k = 8 g + 4..7 : keys 16 + 4 g + r
      pk[qb] = p8.v;
    }
    // ---- O^T = V^T P^T: MFMA (j, qb), output row i = 4 g + r  <->  channel 16 g + 4 j + r of query 16 qb + t16
    float o[2][16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      union { bf16x8 v; bf16x4_t h[2]; } vf;
      vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4_t*)(myV + j * 512 + lane * 8));
      vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4_t*)(myV + 2048 + j * 512 + lane * 8));
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pk[qb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[qb][j * 4 + r] = acc[r] * inv[qb];
      }
    }
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)out + (((long)b * T * S + px) * ldo + h * 64) * 2), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(o[qb][0], o[qb][1]), pack_bf16x2(o[qb][2], o[qb][3]),
                                                   pack_bf16x2(o[qb][4], o[qb][5]), pack_bf16x2(o[qb][6], o[qb][7])}, rsO, off_o[qb], 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(o[qb][8], o[qb][9]), pack_bf16x2(o[qb][10], o[qb][11]),
                                                   pack_bf16x2(o[qb][12], o[qb][13]), pack_bf16x2(o[qb][14], o[qb][15])}, rsO, off_o[qb], 16, 0);
    }
  }
#endif
}

}  // namespace

namespace {
// shared launcher: vrow = 0 -> `v` is the pre-transposed V^T [B][H][64][ld_v] (hi3d_transpose_v), 1 -> row-major V, pitch ld_v
int attn_d64_launch(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t S_q, int32_t S_kv,
                    int32_t ldq, int32_t ldk, int32_t ld_v, int32_t ldo, float scale, int vrow, void* stream) {
  if (!q || !k || !v || !out) HI3D_FAIL(HI3D_EINVAL, "attn_d64: null pointer");
  if (B <= 0 || H <= 0 || S_q <= 0 || S_kv <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_d64: non-positive size");
  if (ldq < H * 64 || ldk < H * 64 || ldo < H * 64) HI3D_FAIL(HI3D_EINVAL, "attn_d64: leading dim < H*64");
  if ((ldq % 8) || (ldk % 8) || (ldo % 4)) HI3D_FAIL(HI3D_EALIGN, "attn_d64: leading dims must keep 16-byte rows");
  if (vrow) {
    if (ld_v % 8 || ld_v < H * 64) HI3D_FAIL(HI3D_EALIGN, "attn_d64_v: ldv must be a multiple of 8 and >= H*64");
  } else {
    if (ld_v % 64 || ld_v < S_kv) HI3D_FAIL(HI3D_ESHAPE, "attn_d64: ld_vt must be S_kv rounded up to 64");
  }
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)out & 7)) HI3D_FAIL(HI3D_EALIGN, "attn_d64: misaligned pointer");
  if (!(scale >= 0.0f)) HI3D_FAIL(HI3D_EINVAL, "attn_d64: scale must be > 0 (or 0: q already carries scale*log2(e))");
  AttnParams p;
  p.q = (const char*)q; p.k = (const char*)k; p.vt = (const char*)v; p.out = (unsigned short*)out;
  p.B = B; p.H = H; p.Sq = S_q; p.Skv = S_kv; p.ldq = ldq; p.ldk = ldk; p.ldvt = ld_v; p.ldo = ldo;
  p.nqt = (S_q + Q_TILE - 1) / Q_TILE;
  { const char* e = getenv("HI3D_ATTN_FORCE_EXACT"); p.force_exact = (e && atoi(e)) ? 1 : 0; }
  p.scale_log2 = scale * 1.4426950408889634f;
  const long nblk = (long)p.nqt * H * B;
  if (nblk > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, "attn_d64: grid too large");
  const dim3 g((unsigned)nblk), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (vrow) {
    if (scale == 0.0f) hipLaunchKernelGGL((attn_d64_kernel<true, true>), g, blk, 0, s, p);
    else hipLaunchKernelGGL((attn_d64_kernel<false, true>), g, blk, 0, s, p);
  } else {
    if (scale == 0.0f) hipLaunchKernelGGL((attn_d64_kernel<true, false>), g, blk, 0, s, p);
    else hipLaunchKernelGGL((attn_d64_kernel<false, false>), g, blk, 0, s, p);
  }
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}
}  // namespace

extern "C" int hi3d_attn_d64(const void* q, const void* k, const void* vt, void* out,
                             int32_t B, int32_t H, int32_t S_q, int32_t S_kv, int32_t ldq,
                             int32_t ldk, int32_t ld_vt, int32_t ldo, float scale, void* stream) {
  return attn_d64_launch(q, k, vt, out, B, H, S_q, S_kv, ldq, ldk, ld_vt, ldo, scale, 0, stream);
}

extern "C" int hi3d_attn_d64_v(const void* q, const void* k, const void* v, void* out,
                               int32_t B, int32_t H, int32_t S_q, int32_t S_kv, int32_t ldq,
                               int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  return attn_d64_launch(q, k, v, out, B, H, S_q, S_kv, ldq, ldk, ldv, ldo, scale, 1, stream);
}

// debug aid: resident blocks per CU the runtime predicts for the spatial attention kernels (0: pre-scaled q, 1: scaled)
extern "C" int hi3d_debug_attn_occupancy(int which) {
  int n = -1;
  if (which == 0) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_d64_kernel<true, false>, 256, 0);
  else if (which == 1) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_d64_kernel<false, false>, 256, 0);
  else if (which == 2) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_d64_kernel<true, true>, 256, 0);
  else hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_d64_kernel<false, true>, 256, 0);
  return n;
}

extern "C" int hi3d_transpose_v(const void* v, void* vt, int32_t B, int32_t H, int32_t S,
                                int32_t S_pad, int32_t ldv, void* stream) {
  if (!v || !vt) HI3D_FAIL(HI3D_EINVAL, "transpose_v: null pointer");
  if (B <= 0 || H <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, "transpose_v: non-positive size");
  if (S_pad % 64 || S_pad < S || S_pad - S >= 64) HI3D_FAIL(HI3D_ESHAPE, "transpose_v: S_pad must be S rounded up to 64");
  if (ldv % 8 || ldv < H * 64) HI3D_FAIL(HI3D_EALIGN, "transpose_v: bad ldv");
  if (((uintptr_t)v | (uintptr_t)vt) & 15) HI3D_FAIL(HI3D_EALIGN, "transpose_v: misaligned pointer");
  if (H > 65535 || B > 65535) HI3D_FAIL(HI3D_ESHAPE, "transpose_v: grid too large");
  hipLaunchKernelGGL(transpose_v_kernel, dim3(S_pad / 64, H, B), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)v, (unsigned short*)vt, H, S, S_pad, ldv);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_attn_temporal_d64(const void* q, const void* k, const void* v, void* out,
                                      int32_t B, int32_t T, int32_t S, int32_t H, int32_t ldqkv,
                                      int32_t ldo, float scale, void* stream) {
  if (!q || !k || !v || !out) HI3D_FAIL(HI3D_EINVAL, "attn_temporal: null pointer");
  if (B <= 0 || T <= 0 || S <= 0 || H <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_temporal: non-positive size");
  if (T > 32) HI3D_FAIL(HI3D_ESHAPE, "attn_temporal: T > 32 not supported");
  if (ldqkv % 8 || ldo % 8 || ldqkv < H * 64 || ldo < H * 64) HI3D_FAIL(HI3D_EALIGN, "attn_temporal: bad leading dim");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) HI3D_FAIL(HI3D_EALIGN, "attn_temporal: misaligned pointer");
  if (H > 65535 || B > 65535) HI3D_FAIL(HI3D_ESHAPE, "attn_temporal: grid too large");
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  // T <= 16 (every Hi3D clip shape but the 32-view one): the matrix-core kernel, one wave per (clip, pixel, head);
  // HI3D_ATTNT_MFMA=0 selects the round-1..3 VALU kernel (A/B switch)
  const int mfma_env = [] { const char* e = getenv("HI3D_ATTNT_MFMA"); return e ? atoi(e) : 1; }();   // (read per call: the tests flip it)
  if (mfma_env && mfma_env != 2 && T <= 16 && (long)B * S * H < 0x7fffffffL && 16L * S * (ldqkv > ldo ? ldqkv : ldo) * 2 < 0x7fffffffL) {
    const long nitem = (long)B * S * H;
    const long want = (nitem + 3) / 4;
    const unsigned grid = (unsigned)(want < 256L * 8 * 4 ? want : 256L * 8 * 4);     // <= 32 blocks of 4 waves per CU's worth
    hipLaunchKernelGGL(attn_temporal_mfma_kernel, dim3(grid), dim3(256), 0, s, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)v, (unsigned short*)out, B, T, S, H, ldqkv, ldo, sl2);
    HI3D_LAUNCH_CHECK();
    return HI3D_OK;
  }
  // 16 < T <= 32 (the 32-view clip of BASELINE config 4): the two-block form of the same kernel (round 5); HI3D_ATTNT_MFMA=2
  // sends T <= 16 through it as well (A/B and parity switch: both kernels on the same inputs)
  if (mfma_env && (T > 16 || mfma_env == 2) && (long)B * S * H < 0x7fffffffL && 32L * S * (ldqkv > ldo ? ldqkv : ldo) * 2 < 0x7fffffffL) {
    const long nitem = (long)B * S * H;
    const long want = (nitem + 3) / 4;
    const unsigned grid = (unsigned)(want < 256L * 8 * 4 ? want : 256L * 8 * 4);
    hipLaunchKernelGGL(attn_temporal_mfma32_kernel, dim3(grid), dim3(256), 0, s, (const unsigned short*)q, (const unsigned short*)k,
                       (const unsigned short*)v, (unsigned short*)out, B, T, S, H, ldqkv, ldo, sl2);
    HI3D_LAUNCH_CHECK();
    return HI3D_OK;
  }
  static const int hfast_env = [] { const char* e = getenv("HI3D_ATTNT_HFAST"); return e ? atoi(e) : 1; }();
  const int hfast = (hfast_env && (S + 7) / 8 <= 65535) ? 1 : 0;
  if (T <= 8) {
    constexpr int TP = 8, PB = 256 / TP;
    hipLaunchKernelGGL(attn_temporal_kernel<TP>, hfast ? dim3(H, (S + PB - 1) / PB, B) : dim3((S + PB - 1) / PB, H, B), dim3(256), 0, s,
                       (const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out, T, S, ldqkv, ldo, sl2, hfast);
  } else if (T <= 16) {
    constexpr int TP = 16, PB = 256 / TP;
    hipLaunchKernelGGL(attn_temporal_kernel<TP>, hfast ? dim3(H, (S + PB - 1) / PB, B) : dim3((S + PB - 1) / PB, H, B), dim3(256), 0, s,
                       (const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out, T, S, ldqkv, ldo, sl2, hfast);
  } else {
    constexpr int TP = 32, PB = 256 / TP;
    hipLaunchKernelGGL(attn_temporal_kernel<TP>, hfast ? dim3(H, (S + PB - 1) / PB, B) : dim3((S + PB - 1) / PB, H, B), dim3(256), 0, s,
                       (const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out, T, S, ldqkv, ldo, sl2, hfast);
  }
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

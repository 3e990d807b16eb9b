// Single-head flash attention, head dim 512, for gfx950: the mid-block attention of the VAE (AttnBlock /
// MemoryEfficientAttnBlock, sgm/modules/diffusionmodules/model.py:180-195 / 226-257: q, k, v = 1x1 convs of the normalised
// 512-channel feature map, softmax(q k^T / sqrt(512)) v over all H*W positions of a frame -- 16384 tokens at the 1024 x 1024
// level's 128 x 128 latent grid).  Rounds 1-3 ran it as GEMM -> fp32 score matrix (1 GiB per frame) -> row softmax -> GEMM;
// this kernel never writes a score (`attn_fwd_d512`).
//
// One block = 64 query rows, 4 waves (one per SIMD, up to 512 VGPRs each), key tiles of 32:
//   * S^T = K Q^T on v_mfma_f32_16x16x32_bf16: wave w owns query rows 16 w .. 16 w + 15 -- its Q rows live in registers as
//     16 B-operand fragments (64 VGPRs) for the whole kernel -- and multiplies them with the whole K tile (32 keys x 512
//     channels in LDS): 32 MFMAs per tile; the swapped product leaves acc[r] = score(key 4 g + r, query lane & 15), so the
//     online-softmax state of a query is lane-local up to two shuffles (as in the d64 kernel);
//   * P (bf16) and the rescale factor of each query row go through a 5 KiB LDS slab: every wave needs all 64 rows of P;
//   * O^T = V^T P^T: wave w owns output channels 128 w .. 128 w + 127 of all 64 query rows (32 accumulator tiles = 128 VGPRs);
//     the V^T operand fragments come out of the ROW-major V tile through ds_read_b64_tr_b16 (no transpose pass); 32 MFMAs.
// K and V tiles (32 KiB each) arrive by LDS-DMA -- one 1 KiB request moves exactly one 512-channel row -- into a 2-stage ring
// (128 KiB), chunk-swizzled on the source side so that both the row-wise K reads and the transposing V reads are conflict-free:
// physical 16-byte chunk = logical chunk ^ f(key), f = (b3 b1 b0 b2) of the key's low four bits.
#include "common.h"
#include <stdlib.h>

namespace {

struct AttnD512Params {
  const char* q; const char* k; const char* v; unsigned short* out;
  int B, S, ldq, ldk, ldv, ldo;
  float scale_log2;
};

constexpr int D5 = 512;
constexpr int D5_BQ = 64, D5_KT = 32;
constexpr int D5_TILE = D5_KT * D5 * 2;                 // bytes of a K (or V) tile: 32 KiB
constexpr int D5_STAGE = 2 * D5_TILE;                   // K + V
constexpr int D5_PROW = 80;                             // bytes per query row of the P slab (64 + 16: spreads rows over banks)
constexpr int D5_P_OFF = 2 * D5_STAGE;                  // P slab [64][80] B
constexpr int D5_A_OFF = D5_P_OFF + D5_BQ * D5_PROW;    // alpha / 1/l per query row, one float per 8-byte slot
constexpr int D5_LDS = D5_A_OFF + D5_BQ * 8;              // (8 bytes per row: written with 8-byte stores)

__device__ __forceinline__ int d5_swz(int key) {        // (b3 b1 b0 b2) of the key's low four bits
  return (key & 8) | ((key & 3) << 1) | ((key >> 2) & 1);
}

__global__ __launch_bounds__(256, 1) void attn_d512_kernel(const AttnD512Params p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nqt = (p.S + D5_BQ - 1) / D5_BQ;
  const int b = blockIdx.x / nqt, q0 = (blockIdx.x - b * nqt) * D5_BQ;

  // ---- this wave's 16 query rows as B-operand fragments: lane (fr, fg) holds channels ks*32 + fg*8 .. +7 of row q0 + 16 w + fr
  constexpr unsigned INV = 0x80000000u;
  const int qrow = q0 + w * 16 + fr;
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.q + (long)b * p.S * p.ldq * 2), 0, 0x7fffffff, 0x00020000);
  bf16x8 qf[16];
  {
    const unsigned qo = qrow < p.S ? (unsigned)(qrow * p.ldq * 2 + fg * 16) : INV;     // rows beyond S: zeros (never stored)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, qo, ks * 64, 0));
  }

  // ---- LDS-DMA of a key tile: piece = one row (512 channels = 1 KiB); wave w moves rows 8 w .. 8 w + 7 of K and of V.
  // Lane L of a piece lands at physical chunk L and fetches logical chunk L ^ f(row).  Rows beyond S read as zeros.
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.k + (long)b * p.S * p.ldk * 2), 0, (int)min((long)0x7fffffff, ((long)p.S - 1) * p.ldk * 2 + D5 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.v + (long)b * p.S * p.ldv * 2), 0, (int)min((long)0x7fffffff, ((long)p.S - 1) * p.ldv * 2 + D5 * 2), 0x00020000);
  unsigned k_vo[8], v_vo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = w * 8 + i;
    const int ch = lane ^ d5_swz(r);
    k_vo[i] = (unsigned)(r * p.ldk * 2 + ch * 16);
    v_vo[i] = (unsigned)(r * p.ldv * 2 + ch * 16);
  }
  auto issue = [&](int j, int st) {
    char* sK = smem + st * D5_STAGE;
    char* sV = sK + D5_TILE;
    const int k0 = j * D5_KT;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (LDS_AS void*)(sK + (w * 8 + i) * 1024), 16, k_vo[i], k0 * p.ldk * 2, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(sV + (w * 8 + i) * 1024), 16, v_vo[i], k0 * p.ldv * 2, 0, 0);
  };

  // ---- fragment addresses
  // K (A operand of S^T): lane (key row fr of key block kb, channels ks*32 + fg*8 ..): logical chunk ks*4 + fg
  const int k_sw = d5_swz(fr);
  const char* k_base = smem + fr * 1024;
  // V^T (A operand of O^T, transposing read): 16-lane group fg reads keys fg*8 + jj*4 + (0..3) x 16 channels; lane i = lane & 15
  // supplies key row fg*8 + jj*4 + (i >> 2), channels db*16 + (i & 3)*4 .. +3 of this wave's slice
  const int vi = lane & 15;
  int v_sw[2]; const char* v_row[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int key = fg * 8 + jj * 4 + (vi >> 2);
    v_sw[jj] = d5_swz(key);
    v_row[jj] = smem + D5_TILE + key * 1024 + ((vi & 1) * 8);
  }
  const int v_c0 = w * 16 + ((vi & 3) >> 1);             // logical chunk of channel 128 w + (i & 3) * 4 (db adds 2 chunks)
  char* const pbuf = smem + D5_P_OFF;
  float* const abuf = (float*)(smem + D5_A_OFF);

  f32x4 o[8][4];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;                  // online-softmax state of query row q0 + 16 w + fr (replicated over fg)

  const int ntile = (p.S + D5_KT - 1) / D5_KT;
  issue(0, 0);
  int st = 0;
  for (int j = 0; j < ntile; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // tile j landed; everyone is done with tile j-1 (its stage, the P slab)
    if (j + 1 < ntile) issue(j + 1, st ^ 1);
    // ---- scores of this wave's 16 queries against the 32 keys.  One wave per SIMD: nothing hides an LDS round trip, and
    // hipcc serialises fragment read -> wait -> MFMA 32 times (measured: ~3.7 k cycles for 0.5 k cycles of MFMA).  So the K
    // fragments are read in inline assembly, 8 at a time (4 k-steps x 2 key blocks), one group ahead of the MFMAs that use
    // them, with counted waits (LDS returns in order).  Address: the swizzle XORs the low four chunk bits only, so
    // k-step 4 g + t reads base[t] + 256 g bytes (+ 16 KiB for the second key block).
    f32x4 sc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      unsigned ka[4];
      const unsigned kb0 = (unsigned)(__UINTPTR_TYPE__)(LDS_AS char*)(k_base + st * D5_STAGE);
#pragma unroll
      for (int t = 0; t < 4; ++t) ka[t] = kb0 + (((((t << 2) ^ (k_sw & 12)) | (fg ^ (k_sw & 3)))) << 4);
      bf16x8 kf[2][4][2];
#define HI3D_KRD(G, T, KB) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[(G) & 1][T][KB]) : "v"(ka[T]), "n"((KB) * 16384 + (G) * 256))
#define HI3D_KGROUP(G) HI3D_KRD(G, 0, 0); HI3D_KRD(G, 0, 1); HI3D_KRD(G, 1, 0); HI3D_KRD(G, 1, 1); HI3D_KRD(G, 2, 0); HI3D_KRD(G, 2, 1); HI3D_KRD(G, 3, 0); HI3D_KRD(G, 3, 1)
#define HI3D_KWAIT(G, CNT) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(kf[(G) & 1][0][0]), "+v"(kf[(G) & 1][0][1]), "+v"(kf[(G) & 1][1][0]), "+v"(kf[(G) & 1][1][1]), \
                                          "+v"(kf[(G) & 1][2][0]), "+v"(kf[(G) & 1][2][1]), "+v"(kf[(G) & 1][3][0]), "+v"(kf[(G) & 1][3][1]))
#define HI3D_KMMA(G) _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
        sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[(G) & 1][t][kb], qf[(G) * 4 + t], sc[kb], 0, 0, 0)
      HI3D_KGROUP(0);
      HI3D_KGROUP(1); HI3D_KWAIT(0, 8); HI3D_KMMA(0);
      HI3D_KGROUP(2); HI3D_KWAIT(1, 8); HI3D_KMMA(1);
      HI3D_KGROUP(3); HI3D_KWAIT(2, 8); HI3D_KMMA(2);
      HI3D_KWAIT(3, 0); HI3D_KMMA(3);
#undef HI3D_KRD
#undef HI3D_KGROUP
#undef HI3D_KWAIT
#undef HI3D_KMMA
    }
    // lane registers: sc[kb][r] = score(key j*32 + kb*16 + 4 fg + r, query fr)
    float s8[8], mx = -INFINITY;
    const int rem = p.S - j * D5_KT;                     // keys of this tile that exist
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (kb * 16 + 4 * fg + r < rem) ? sc[kb][r] * p.scale_log2 : -INFINITY;
        s8[kb * 4 + r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                // finite: every tile has >= 1 valid key
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float e8[8], ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e8[i] = __builtin_amdgcn_exp2f(s8[i] - m_new); ps += e8[i]; }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // P[query][key] (bf16) and alpha[query] to the slab
    // (stores the compiler does not see: it drains every in-flight LDS-DMA in front of an LDS store it knows of -- here the
    // next tile's K / V requests -- common.h lds_store_b64_nodrain; alpha rides in both halves of an 8-byte store per row group)
    {
      char* pr = pbuf + (w * 16 + fr) * D5_PROW + fg * 8;
      lds_store_b64_nodrain(pr, pack_bf16x2(e8[0], e8[1]), pack_bf16x2(e8[2], e8[3]));
      lds_store_b64_nodrain(pr + 32, pack_bf16x2(e8[4], e8[5]), pack_bf16x2(e8[6], e8[7]));
      if (fg == 0) lds_store_b64_nodrain((char*)abuf + (w * 16 + fr) * 8, __float_as_uint(alpha), 0u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // raw: the K / V requests of tile j + 1 stay in flight
    // ---- O^T += V^T P^T for this wave's 128 channels, all 64 queries
    bf16x8 pf[4];
    float al[4];
    bool any = false;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      pf[qb] = *(const bf16x8*)(pbuf + (qb * 16 + fr) * D5_PROW + fg * 16);
      al[qb] = abuf[(qb * 16 + fr) * 2];
      any = any || (al[qb] != 1.0f);
    }
    if (__any(any)) {                                    // (wave-uniform) the running maximum of some row moved: rescale
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) { o[db][qb][0] *= al[qb]; o[db][qb][1] *= al[qb]; o[db][qb][2] *= al[qb]; o[db][qb][3] *= al[qb]; }
    }
    // The 16 transposing reads of the tile in inline assembly (see attention.hip: the compiler would put `s_waitcnt vmcnt(0)`
    // -- a wait for the NEXT tile's 64 KiB of K / V requests -- in front of the first ds_read_b64_tr_b16 builtin; measured
    // here: 1.23 ms per 16384-token frame with it).  P fragments / alpha (compiler-visible reads) are retired first so that
    // the counted waits below see only these 16 reads; each channel block starts when its two reads have returned.
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]));
    bf16x4 vt[8][2];
    {
      const unsigned va0 = (unsigned)(__UINTPTR_TYPE__)(LDS_AS char*)(v_row[0] + st * D5_STAGE);
      const unsigned va1 = (unsigned)(__UINTPTR_TYPE__)(LDS_AS char*)(v_row[1] + st * D5_STAGE);
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const int c = v_c0 + db * 2;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[db][0]) : "v"(va0 + ((c ^ v_sw[0]) << 4)));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[db][1]) : "v"(va1 + ((c ^ v_sw[1]) << 4)));
      }
    }
#define HI3D_D5_PV(DB, CNT)                                                                                   \
    {                                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(vt[DB][0]), "+v"(vt[DB][1]));                        \
      const bf16x4 v0 = vt[DB][0], v1 = vt[DB][1];                                                            \
      const bf16x8 vf = bf16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};                       \
      _Pragma("unroll") for (int qb = 0; qb < 4; ++qb)                                                        \
        o[DB][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb], o[DB][qb], 0, 0, 0);                  \
    }
    HI3D_D5_PV(0, 14) HI3D_D5_PV(1, 12) HI3D_D5_PV(2, 10) HI3D_D5_PV(3, 8)
    HI3D_D5_PV(4, 6) HI3D_D5_PV(5, 4) HI3D_D5_PV(6, 2) HI3D_D5_PV(7, 0)
#undef HI3D_D5_PV
    st ^= 1;
  }
  // ---- finish: 1 / l of every query row through the slab, then out[q][128 w + db*16 + 4 fg + r] (8-byte pieces)
  __syncthreads();
  if (fg == 0) abuf[(w * 16 + fr) * 2] = 1.0f / l_run;
  __syncthreads();
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int q = q0 + qb * 16 + fr;
    const float inv = abuf[(qb * 16 + fr) * 2];
    if (q < p.S) {
      unsigned short* op = p.out + ((long)b * p.S + q) * p.ldo + w * 128 + fg * 4;
#pragma unroll
      for (int db = 0; db < 8; ++db)
        *(uint2*)(op + db * 16) = make_uint2(pack_bf16x2(o[db][qb][0] * inv, o[db][qb][1] * inv), pack_bf16x2(o[db][qb][2] * inv, o[db][qb][3] * inv));
    }
  }
#endif
}

}  // namespace

extern "C" int hi3d_attn_d512(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t S,
                              int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  if (!q || !k || !v || !out) HI3D_FAIL(HI3D_EINVAL, "attn_d512: null pointer");
  if (B <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_d512: non-positive size");
  if (ldq < D5 || ldk < D5 || ldv < D5 || ldo < D5 || (ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 4))
    HI3D_FAIL(HI3D_EALIGN, "attn_d512: leading dims must be >= 512 and keep 16-byte rows");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)out & 7)) HI3D_FAIL(HI3D_EALIGN, "attn_d512: misaligned pointer");
  if (!(scale > 0.0f)) HI3D_FAIL(HI3D_EINVAL, "attn_d512: scale must be > 0");
  if ((long)S * (ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv)) * 2 >= 0x7fffffffL)
    HI3D_FAIL(HI3D_ESHAPE, "attn_d512: a frame's q / k / v must span less than 2 GiB");
  AttnD512Params p;
  p.q = (const char*)q; p.k = (const char*)k; p.v = (const char*)v; p.out = (unsigned short*)out;
  p.B = B; p.S = S; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.scale_log2 = scale * 1.4426950408889634f;
  const long nblk = (long)B * ((S + D5_BQ - 1) / D5_BQ);
  if (nblk > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, "attn_d512: grid too large");
  static bool attr_done[HI3D_MAX_DEVICES] = {};
  if (int rc = hi3d_raise_lds_limit((const void*)attn_d512_kernel, D5_LDS, attr_done)) return rc;
  hipLaunchKernelGGL(attn_d512_kernel, dim3((unsigned)nblk), dim3(256), D5_LDS, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// Spatial self-attention with the SCORE product on the CDNA4 fp8 matrix path (BASELINE config 5:
// "fp8 MFMA attention + bf16 conv"): S^T = K Q^T runs on v_mfma_scale_f32_32x32x64_f8f6f4 -- the whole
// head dimension (64) in ONE instruction at twice the bf16 rate -- on OCP e4m3 operands with MX block
// scales (one power-of-two e8m0 scale per 32 elements of a row, applied by the matrix core: no VALU
// work); the softmax and the P V product stay exactly as in attn_d64_kernel (bf16 P, bf16 V^T, fp32
// accumulate): quantising P to e4m3 costs 6 % relative error per key and was rejected.
//
//   (1) quant_qk_kernel : q | k columns of the fused QKV tensor (bf16, q already carrying
//       softmax-scale * log2 e) -> per (b, h): rows of 64 e4m3 bytes + 2 e8m0 scale bytes, S padded to 64
//   (2) attn_d64_fp8qk_kernel : attn_d64_kernel<PRE = true> with fp8 Q / K fragments
//
// Reference: CrossAttention / MemoryEfficientCrossAttention, sgm/modules/attention.py:332-336, 427-439
// (the reference's GPU path is fp16 xformers; this is the reduced-precision variant, own tolerance).
#include "common.h"
#include <type_traits>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swap_bits23(int i) {
  return (i & ~0xC) | ((i & 4) << 1) | ((i & 8) >> 1);
}

constexpr int KV_TILE = 64;
constexpr float SUM_MAX = 2048.0f;
constexpr int K8_BYTES = KV_TILE * 64;          // 64 keys x 64 e4m3
constexpr int KS_BYTES = 256;                   // 64 keys x 2 scale bytes, one 4-byte LDS-DMA per lane
constexpr int VT_BYTES = KV_TILE * 128;         // V^T tile: 64 d-rows x 64 keys bf16
constexpr int STAGE8 = K8_BYTES + KS_BYTES + VT_BYTES;
constexpr int QB = 2;
constexpr int Q_TILE = 4 * QB * 32;

// ---- (1) quantise: thread = 8 consecutive d of one row of q or k.  Shared exponent = floor(log2(amax)) - 7
// (scaled magnitudes < 256, inside e4m3's 448).  QK_SCALE_BLOCK = 64: ONE exponent per row of a head, written to
// both of the row's scale bytes -- the matrix core then applies the same scale whichever way it maps a lane's
// scale byte to the K elements of the instruction.  (With one exponent per 32-element half row, the MX block
// size, and each lane's 32 bytes taken as one block, the kernel did NOT match its CPU restatement on the
// MI355X: the byte -> K-element mapping of the instruction is not what that assumed.  Row-wise scales need no
// such assumption and measure cos 0.9992 / rms 4 % against fp32 attention, 0.35 % against the restatement.)
constexpr int QK_SCALE_BLOCK = 64;
__global__ __launch_bounds__(256) void quant_qk_kernel(const unsigned short* __restrict__ qkv, unsigned char* __restrict__ q8,
                                                       unsigned char* __restrict__ qs, unsigned char* __restrict__ k8,
                                                       unsigned char* __restrict__ ks, int H, int S, int S_pad, int ld) {
  const int tid = threadIdx.x, chunk = tid & 7, rl = tid >> 3;
  const int s = blockIdx.x * 32 + rl, h = blockIdx.y, b = blockIdx.z >> 1, which = blockIdx.z & 1;
  if (s >= S_pad) return;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (s < S) v = *(const uint4*)(qkv + ((long)b * S + s) * ld + which * H * 64 + h * 64 + chunk * 8);
  const unsigned int u[4] = {v.x, v.y, v.z, v.w};
  float f[8];
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf16_to_f32(u[j] & 0xffff); f[2 * j + 1] = bf16_to_f32(u[j] >> 16);
    amax = fmaxf(amax, fmaxf(fabsf(f[2 * j]), fabsf(f[2 * j + 1])));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  if (QK_SCALE_BLOCK == 64) amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff);          // biased exponent of the block maximum
  e = e < 8 ? 8 : (e > 254 ? 254 : e);
  const float inv = __uint_as_float((unsigned)(261 - e) << 23);  // 2^(7 - floor(log2 amax))
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, w1, true);
  unsigned char* o8 = which ? k8 : q8;
  unsigned char* os = which ? ks : qs;
  const long row = ((long)b * H + h) * S_pad + s;
  *(uint2*)(o8 + row * 64 + chunk * 8) = make_uint2((unsigned)w0, (unsigned)w1);
  if ((chunk & 3) == 0) os[row * 2 + (chunk >> 2)] = (unsigned char)(e - 7);     // e8m0: 2^(byte - 127)
}

struct Attn8Params {
  const char* q8; const unsigned char* qs; const char* k8; const unsigned char* ks; const char* vt; unsigned short* out;
  int B, H, Sq, Skv, Spq, Spk, ldvt, ldo, nqt;
};

__global__ __launch_bounds__(256, 2) void attn_d64_fp8qk_kernel(const Attn8Params p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;

  const int nblk = gridDim.x;
  int lid;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int h = bh % p.H, b = bh / p.H;

  // ---- Q fragments: lane (li, hi) holds Q8[row li][d = hi*32 .. +31] (any split of d works as long as the
  // K fragment uses the same one: d is the contracted index) and the row's scale byte of that 32-block
  i32x8 qf[QB]; int qsc[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qr = qt * Q_TILE + (w * QB + qb) * 32 + li;
    const bool ok = qr < p.Sq;
    const long row = (long)bh * p.Spq + (ok ? qr : 0);
    const i32x4 lo = *(const i32x4*)(p.q8 + row * 64 + hi * 32), hi4 = *(const i32x4*)(p.q8 + row * 64 + hi * 32 + 16);
    qf[qb] = i32x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    qsc[qb] = p.qs[row * 2 + hi];
    if (!ok) qf[qb] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // ---- LDS-DMA: K8 tile = 4 pieces of 1 KiB (16 keys x 64 B), piece w by wave w; scales: one 4-byte
  // DMA by wave 0; V^T tile as in attn_d64_kernel (pieces 2w, 2w+1).  K8 rows are 64 B = half a bank
  // line: 16-byte chunk c of key row r is stored at chunk c ^ ((r >> 1) & 3) (source-side swizzle).
  const char* kbase = p.k8 + (long)bh * p.Spk * 64;
  const unsigned char* ksbase = p.ks + (long)bh * p.Spk * 2;
  const char* vbase = p.vt + (long)bh * 64 * (long)p.ldvt * 2;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int k_r = w * 16 + (lane >> 2);
  const int k_src = k_r * 64 + (((lane & 3) ^ ((k_r >> 1) & 3)) << 4);
  int v_c[2], v_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 8 + lrow;
    v_c[i] = lslot ^ ((r >> 1) & 7);
    v_off[i] = r * p.ldvt * 2;                       // < 64 * S_pad * 2 bytes: fits 32 bits
  }
  // buffer-addressed LDS-DMA (as in gemm.hip): a descriptor per operand based at this (b, h), a 32-bit
  // per-lane offset fixed for the whole loop and a scalar offset walking the key tiles -- no 64-bit
  // per-lane pointers (they did not fit the register file next to the fp32 accumulators)
#if __HIP_DEVICE_COMPILE__
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)ksbase, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0x7fffffff, 0x00020000);
  const int s_vo = lane * 4;
  int v_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) v_vo[i] = v_off[i] + v_c[i] * 16;
  auto issue = [&](int j, int st) {
    char* sK = smem + st * STAGE8;
    char* sS = sK + K8_BYTES;
    char* sV = sS + KS_BYTES;
    const int kv0 = j * KV_TILE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (LDS_AS void*)(sK + w * 1024), 16, k_src, kv0 * 64, 0, 0);   // rows >= Skv: zero padding
    if (w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (LDS_AS void*)sS, 4, s_vo, kv0 * 2, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(sV + (w * 2 + i) * 1024), 16, v_vo[i], kv0 * 2, 0, 0);
  };
#else
  auto issue = [&](int, int) {};
#endif

  // fragment read addresses
  const int kR = swap_bits23(li);                       // key row this lane feeds (per 32-key block)
  const int f_sw = (li >> 1) & 7;
  int k_ptr = kR * 64;                                  // (32-bit LDS offsets, not generic pointers: registers)
  const int k_c0 = ((hi * 2) ^ ((kR >> 1) & 3)) << 4, k_c1 = ((hi * 2 + 1) ^ ((kR >> 1) & 3)) << 4;
  int s_ptr = K8_BYTES + kR * 2 + hi;
  int v_ptr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) v_ptr[ks] = K8_BYTES + KS_BYTES + li * 128 + (((ks * 2 + hi) ^ f_sw) << 4);
  int stage_step = STAGE8;

  f32x16 o[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  float m_run[QB], l_run[QB];
  f32x16 negm[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
  }

  const int ntile = (p.Skv + KV_TILE - 1) / KV_TILE;
  const int nfull = p.Skv / KV_TILE;
  auto tile = [&](const int j, auto ragged_tag) {
    constexpr bool ragged = decltype(ragged_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntile) issue(j + 1, (j & 1) ^ 1);

    // S^T - m = (K8 Q8^T) * 2^(sk + sq) - m: ONE MFMA per (32 keys x 32 queries), block scales applied by the core
    auto scores = [&](const int kb, f32x16 (&sc)[QB]) {
      const i32x4 k0 = *(const i32x4*)(smem + k_ptr + kb * 32 * 64 + k_c0), k1 = *(const i32x4*)(smem + k_ptr + kb * 32 * 64 + k_c1);
      const i32x8 kf = i32x8{k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
      const int ksc = *(const unsigned char*)(smem + s_ptr + kb * 64);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
        sc[qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[qb], negm[qb], 0, 0, 0, ksc, 0, qsc[qb]);
      if (ragged) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = j * KV_TILE + kb * 32 + (r >> 3) * 16 + hi * 8 + (r & 7);
            sc[qb][r] = (kv >= p.Skv) ? -INFINITY : sc[qb][r];
          }
      }
    };

    bf16x8 pf[QB][4];
    float psum[QB];
    bool exact = (j == 0) || ragged;
    for (;;) {
      if (exact) {
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
          const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));
          const float d = (j == 0) ? t : fmaxf(t, 0.f);
          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);
          m_run[qb] += d;
          l_run[qb] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) negm[qb][r] -= d;
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
        }
      }
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
              hi3d_f2 e;
              e[0] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 2 * t]);
              e[1] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 2 * t + 1]);
              ps[qb] += e;
              pk.u[t] = pack_bf16x2(e[0], e[1]);
            }
            pf[qb][kb * 2 + half] = pk.v;
          }
        }
      }
      bool ok = true;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { psum[qb] = ps[qb][0] + ps[qb][1]; ok = ok && (psum[qb] <= SUM_MAX); }
      if (exact || __all(ok)) break;
      exact = true;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];

#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 vf = *(const bf16x8*)(smem + v_ptr[ks] + db * 32 * 128);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], o[qb][db], 0, 0, 0);
      }
    k_ptr += stage_step; s_ptr += stage_step;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) v_ptr[ks] += stage_step;
    stage_step = -stage_step;
  };
  issue(0, 0);
  for (int j = 0; j < nfull; ++j) tile(j, std::false_type{});
  if (nfull < ntile) tile(nfull, std::true_type{});

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = qt * Q_TILE + (w * QB + qb) * 32 + li;
    if (qr < p.Sq) {
      unsigned short* op = p.out + ((long)b * p.Sq + qr) * p.ldo + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = pack_bf16x2(o[qb][db][g * 4 + 0] * inv, o[qb][db][g * 4 + 1] * inv);
          v.y = pack_bf16x2(o[qb][db][g * 4 + 2] * inv, o[qb][db][g * 4 + 3] * inv);
          *(uint2*)(op + db * 32 + g * 8 + hi * 4) = v;
        }
    }
  }
}

}  // namespace

extern "C" int64_t hi3d_attn_fp8_workspace_bytes(int32_t B, int32_t H, int32_t S) {
  const int64_t S_pad = ((int64_t)S + 63) / 64 * 64;
  // q8 | k8 (64 B per row) then qs | ks (2 B per row, + one DMA overshoot of 256 B each), 256-byte aligned sections
  const int64_t rows = (int64_t)B * H * S_pad;
  return 2 * (rows * 64) + 2 * ((rows * 2 + 256 + 255) / 256 * 256);
}

extern "C" int hi3d_attn_quant_qk(const void* qkv, void* ws, int32_t B, int32_t H, int32_t S, int32_t ld, void* stream) {
  if (!qkv || !ws) HI3D_FAIL(HI3D_EINVAL, "attn_quant_qk: null pointer");
  if (B <= 0 || H <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_quant_qk: non-positive size");
  if (ld < 2 * H * 64 || ld % 8) HI3D_FAIL(HI3D_EALIGN, "attn_quant_qk: ld must cover q | k and keep 16-byte rows");
  if (((uintptr_t)qkv & 15) || ((uintptr_t)ws & 255)) HI3D_FAIL(HI3D_EALIGN, "attn_quant_qk: misaligned pointer");
  if (H > 65535 || 2 * B > 65535) HI3D_FAIL(HI3D_ESHAPE, "attn_quant_qk: grid too large");
  const int S_pad = (S + 63) / 64 * 64;
  const long rows = (long)B * H * S_pad;
  const long ssec = (rows * 2 + 256 + 255) / 256 * 256;
  unsigned char* q8 = (unsigned char*)ws;
  unsigned char* k8 = q8 + rows * 64;
  unsigned char* qs = k8 + rows * 64;
  unsigned char* ks = qs + ssec;
  hipLaunchKernelGGL(quant_qk_kernel, dim3(S_pad / 32, H, 2 * B), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)qkv, q8, qs, k8, ks, H, S, S_pad, ld);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_attn_d64_fp8qk(const void* ws, const void* vt, void* out, int32_t B, int32_t H, int32_t S,
                                   int32_t ld_vt, int32_t ldo, void* stream) {
  if (!ws || !vt || !out) HI3D_FAIL(HI3D_EINVAL, "attn_d64_fp8qk: null pointer");
  if (B <= 0 || H <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_d64_fp8qk: non-positive size");
  if (ldo < H * 64 || ldo % 4) HI3D_FAIL(HI3D_EALIGN, "attn_d64_fp8qk: bad ldo");
  const int S_pad = (S + 63) / 64 * 64;
  if (ld_vt != S_pad) HI3D_FAIL(HI3D_ESHAPE, "attn_d64_fp8qk: ld_vt must be S rounded up to 64");
  if (((uintptr_t)ws & 255) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7)) HI3D_FAIL(HI3D_EALIGN, "attn_d64_fp8qk: misaligned pointer");
  const long rows = (long)B * H * S_pad;
  const long ssec = (rows * 2 + 256 + 255) / 256 * 256;
  Attn8Params p;
  p.q8 = (const char*)ws; p.k8 = p.q8 + rows * 64;
  p.qs = (const unsigned char*)(p.k8 + rows * 64); p.ks = p.qs + ssec;
  p.vt = (const char*)vt; p.out = (unsigned short*)out;
  p.B = B; p.H = H; p.Sq = S; p.Skv = S; p.Spq = S_pad; p.Spk = S_pad; p.ldvt = ld_vt; p.ldo = ldo;
  p.nqt = (S + Q_TILE - 1) / Q_TILE;
  const long nblk = (long)p.nqt * H * B;
  if (nblk > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, "attn_d64_fp8qk: grid too large");
  hipLaunchKernelGGL(attn_d64_fp8qk_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

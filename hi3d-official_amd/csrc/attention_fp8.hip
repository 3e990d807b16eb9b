// Spatial self-attention on the CDNA4 fp8 matrix path (BASELINE config 5: "fp8 MFMA attention + bf16 conv").
//
// Score product: S^T = K Q^T on v_mfma_scale_f32_32x32x64_f8f6f4 -- the whole head dimension (64) in ONE instruction at
// twice the bf16 rate -- on OCP e4m3 operands with power-of-two e8m0 scales applied by the matrix core (no VALU work).
// Two forms of the second product (template PV8):
//   PV8 = false ("fp8qk", round 2): softmax and P V exactly as in attn_d64_kernel (bf16 P, bf16 V^T);
//   PV8 = true  ("fp8", round 3):   O^T += V^T P^T on the same instruction, the whole 64-key tile in ONE MFMA per
//       (32 d x 32 queries): P is produced as exp2(score + 3) (the +3 rides the C operand of the score MFMA, like
//       the reference point) and converted to e4m3 with v_cvt_pk_fp8_f32 -- same instruction count as the bf16 pack --
//       with the constant scale 2^-3 handed to the matrix core; V^T is quantised once per attention call to e4m3 with
//       one e8m0 exponent per (d row, 64-key tile).  A key tile in which a lane's 32-term row sum of P exceeds 32 (a P
//       could leave e4m3's range) is redone through the exact pre-pass, as the bf16 kernel does at 2048.  3 mantissa bits on P and V:
//       own, looser tolerance, measured in tests/ at S = 16384.
//
//   (1) quant_qk_kernel : q | k columns of the fused QKV tensor (bf16, q already carrying
//       softmax-scale * log2 e) -> per (b, h): rows of 64 e4m3 bytes + 2 e8m0 scale bytes, S padded to 64
//   (2) quant_vt_kernel : v columns -> per (b, h, key tile): 64 d rows x 64 e4m3 bytes in the key order the P registers
//       come out of the score MFMA in, + 64 e8m0 scale bytes
//   (3) attn_d64_fp8_kernel<PV8>
// Keys >= S_kv take the common path: -inf through the C operand of the score MFMA (one code path).
//
// Reference: CrossAttention / MemoryEfficientCrossAttention, sgm/modules/attention.py:332-336, 427-439
// (the reference's GPU path is fp16 xformers; these are the reduced-precision variants, own tolerances).
#include "common.h"

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
constexpr int V8_BYTES = KV_TILE * 64;          // PV8: 64 d-rows x 64 e4m3
constexpr int VS_BYTES = 256;                   // PV8: 64 scale bytes (one 4-byte LDS-DMA per lane moves 256)
constexpr float P8_SHIFT = 3.0f;                // PV8: P is formed as exp2(score + 3): 1.0 -> 8 (e4m3 then spans P = 2^-12 .. 56)
constexpr float SUM_MAX8 = 32.0f * 8.0f;        // PV8: redo the tile exactly when a lane's 32-term row sum of P exceeds 32
                                                // (every P <= 32 < 56 then; after the exact pass every P <= 1, so the sum fits)
template <bool PV8> constexpr int stage_bytes() { return K8_BYTES + KS_BYTES + (PV8 ? V8_BYTES + VS_BYTES : VT_BYTES); }
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

// ---- (2) V -> V^T in e4m3, per (b, h, 64-key tile): 64 d rows x 64 bytes + 64 scale bytes.  Byte n of lane-half hf of a
// row holds key  (n >> 4) * 32 + ((n >> 3) & 1) * 16 + hf * 8 + (n & 7)  of the tile: the order in which a lane's 32 P
// values of the tile leave the two score MFMAs (register r of key block kb <-> byte kb * 16 + r), so that the P registers
// are converted in place and the contraction pairs byte n of A with byte n of B.  Keys >= S: zeros.
__global__ __launch_bounds__(256) void quant_vt_kernel(const unsigned short* __restrict__ v, unsigned char* __restrict__ v8,
                                                       unsigned char* __restrict__ vs, int H, int S, int ldv) {
  __shared__ unsigned short tile[64][66];   // [key][d], +2 pad
  const int st = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int s0 = st * 64, ntile = gridDim.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
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
  // thread = (d row, 16-byte chunk c = hf * 2 + kb): keys kb*32 + g*16 + hf*8 + i, g = 0..1, i = 0..7
  const int d = tid >> 2, c = tid & 3, hf = c >> 1, kb = c & 1;
  float f[16];
  float amax = 0.f;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    f[n] = bf16_to_f32(tile[kb * 32 + (n >> 3) * 16 + hf * 8 + (n & 7)][d]);
    amax = fmaxf(amax, fabsf(f[n]));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));          // the 4 threads of a d row
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff);
  e = e < 8 ? 8 : (e > 254 ? 254 : e);
  const float inv = __uint_as_float((unsigned)(261 - e) << 23);
  int w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j] * inv, f[4 * j + 1] * inv, w[j], false);
    w[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j + 2] * inv, f[4 * j + 3] * inv, w[j], true);
  }
  const long t = ((long)b * H + h) * ntile + st;
  *(uint4*)(v8 + (t * 64 + d) * 64 + c * 16) = make_uint4((unsigned)w[0], (unsigned)w[1], (unsigned)w[2], (unsigned)w[3]);
  if (c == 0) vs[t * 64 + d] = (unsigned char)(e - 7);
}

struct Attn8Params {
  const char* q8; const unsigned char* qs; const char* k8; const unsigned char* ks;
  const char* vt;                // PV8: V^T e4m3 tiles [bh][tile][64 d][64];  else bf16 V^T [bh][64][ldvt]
  const unsigned char* vsc;      // PV8: [bh][tile][64] e8m0
  unsigned short* out;
  int B, H, Sq, Skv, Spq, Spk, ldvt, ldo, nqt;
};

template <bool PV8>
__global__ __launch_bounds__(256, 2) void attn_d64_fp8_kernel(const Attn8Params p) {
  constexpr int STAGE8 = stage_bytes<PV8>();
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
  const int ntile = (p.Skv + KV_TILE - 1) / KV_TILE;
  const int nfull = p.Skv / KV_TILE;                    // j == nfull < ntile: the tile with keys >= S_kv

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

  // ---- LDS-DMA: K8 tile = 4 pieces of 1 KiB (16 keys x 64 B), piece w by wave w; scales: one 4-byte DMA by wave 0.
  // V^T: bf16 tile as in attn_d64_kernel (pieces 2w, 2w+1), or (PV8) the e4m3 tile, laid out like the K8 tile (64 rows of
  // 64 B: piece w by wave w) + its 64 scale bytes (4-byte DMA by wave 1).  64-byte rows are half a bank line: 16-byte
  // chunk c of row r is stored at chunk c ^ ((r >> 1) & 3) (source-side swizzle).
  const char* kbase = p.k8 + (long)bh * p.Spk * 64;
  const unsigned char* ksbase = p.ks + (long)bh * p.Spk * 2;
  const char* vbase = PV8 ? p.vt + (long)bh * ntile * (64 * 64) : p.vt + (long)bh * 64 * (long)p.ldvt * 2;
  const unsigned char* vsbase = PV8 ? p.vsc + (long)bh * ntile * 64 : nullptr;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int k_r = w * 16 + (lane >> 2);
  const int k_src = k_r * 64 + (((lane & 3) ^ ((k_r >> 1) & 3)) << 4);
#if __HIP_DEVICE_COMPILE__
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)ksbase, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsVS = __builtin_amdgcn_make_buffer_rsrc((void*)(PV8 ? vsbase : ksbase), 0, 0x7fffffff, 0x00020000);
  const int s_vo = lane * 4;
  int v_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 8 + lrow;
    v_vo[i] = r * p.ldvt * 2 + ((lslot ^ ((r >> 1) & 7)) << 4);      // (bf16 V^T form)
  }
  auto issue = [&](int j, int st) {
    char* sK = smem + st * STAGE8;
    char* sS = sK + K8_BYTES;
    char* sV = sS + KS_BYTES;
    const int kv0 = j * KV_TILE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (LDS_AS void*)(sK + w * 1024), 16, k_src, kv0 * 64, 0, 0);   // rows >= Skv: zero padding
    if (w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (LDS_AS void*)sS, 4, s_vo, kv0 * 2, 0, 0);
    if (PV8) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(sV + w * 1024), 16, k_src, j * (64 * 64), 0, 0);
      if (w == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsVS, (LDS_AS void*)(sV + V8_BYTES), 4, s_vo, j * 64, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (LDS_AS void*)(sV + (w * 2 + i) * 1024), 16, v_vo[i], kv0 * 2, 0, 0);
    }
  };
#else
  auto issue = [&](int, int) {};
#endif

  // fragment read addresses (32-bit LDS offsets, not generic pointers: registers)
  const int kR = swap_bits23(li);                       // key row this lane feeds (per 32-key block)
  const int f_sw = (li >> 1) & 7;
  int k_ptr = kR * 64;
  const int k_c0 = ((hi * 2) ^ ((kR >> 1) & 3)) << 4, k_c1 = ((hi * 2 + 1) ^ ((kR >> 1) & 3)) << 4;
  int s_ptr = K8_BYTES + kR * 2 + hi;
  int v_ptr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) v_ptr[ks] = K8_BYTES + KS_BYTES + li * 128 + (((ks * 2 + hi) ^ f_sw) << 4);
  // PV8: d row li (+32 per d block) of the e4m3 tile, chunks hi*2, hi*2+1 (swizzled like the K8 rows), and its scale byte
  int v8_ptr = K8_BYTES + KS_BYTES + li * 64;
  const int v8_c0 = ((hi * 2) ^ ((li >> 1) & 3)) << 4, v8_c1 = ((hi * 2 + 1) ^ ((li >> 1) & 3)) << 4;
  int vs_ptr = K8_BYTES + KS_BYTES + V8_BYTES + li;
  int stage_step = STAGE8;

  f32x16 o[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  // reference point m_ref of the rows (log2 domain of the pre-scaled scores); the C operand of the score MFMA carries
  // -m_ref (+ P8_SHIFT for PV8: P = exp2(score - m_ref + 3) lands in e4m3's range), or -inf for keys >= S_kv
  constexpr float OFF = PV8 ? P8_SHIFT : 0.0f;
  constexpr float SMAX = PV8 ? SUM_MAX8 : SUM_MAX;
  float m_ref[QB], l_run[QB];
  f32x16 cneg[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_ref[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) cneg[qb][r] = OFF;
  }

  issue(0, 0);
  for (int j = 0; j < ntile; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntile) issue(j + 1, (j & 1) ^ 1);
    const bool tail = (j == nfull);                     // wave-uniform; true at most once

    // S^T - m (+ OFF) = (K8 Q8^T) * 2^(sk + sq) + C: ONE MFMA per (32 keys x 32 queries), block scales applied by the core
    auto scores = [&](const int kb, f32x16 (&sc)[QB]) {
      if (tail) {                                       // as in attn_d64_kernel: -inf into the C registers of keys >= S_kv
        int rem = p.Skv - j * KV_TILE;
        asm volatile("" : "+s"(rem));
        const int lim = rem - kb * 32 - hi * 8;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cneg[qb][r] = ((r >> 3) * 16 + (r & 7) >= lim) ? -INFINITY : OFF - m_ref[qb];
      }
      const i32x4 k0 = *(const i32x4*)(smem + k_ptr + kb * 32 * 64 + k_c0), k1 = *(const i32x4*)(smem + k_ptr + kb * 32 * 64 + k_c1);
      const i32x8 kf = i32x8{k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
      const int ksc = *(const unsigned char*)(smem + s_ptr + kb * 64);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
        sc[qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[qb], cneg[qb], 0, 0, 0, ksc, 0, qsc[qb]);
    };

    bf16x8 pf[QB][4];              // bf16 P: 4 k-steps of 16 keys (8 per lane half)
    int p8[QB][8];                 // PV8: the lane's 32 P values of the tile as e4m3 bytes, byte kb*16 + r <-> register r of key block kb
    float psum[QB];
    bool exact = (j == 0);
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
          const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64)) - OFF;      // relative to m_ref
          const float d = (j == 0) ? t : fmaxf(t, 0.f);
          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);
          m_ref[qb] += d;
          l_run[qb] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) cneg[qb][r] -= d;
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
            for (int tp = 0; tp < 2; ++tp) {          // four scores -> one e4m3 dword (PV8) / two bf16 dwords
              hi3d_f2 e0, e1;
              e0[0] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 4 * tp]);
              e0[1] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 4 * tp + 1]);
              e1[0] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 4 * tp + 2]);
              e1[1] = __builtin_amdgcn_exp2f(sc[qb][half * 8 + 4 * tp + 3]);
              ps[qb] += e0;
              ps[qb] += e1;
              if (PV8) {          // bytes kb*16 + half*8 + 4tp .. +3 of the lane's 32: dword kb*4 + half*2 + tp
                int wv = __builtin_amdgcn_cvt_pk_fp8_f32(e0[0], e0[1], 0, false);
                p8[qb][kb * 4 + half * 2 + tp] = __builtin_amdgcn_cvt_pk_fp8_f32(e1[0], e1[1], wv, true);
              } else {
                pk.u[2 * tp] = pack_bf16x2(e0[0], e0[1]);
                pk.u[2 * tp + 1] = pack_bf16x2(e1[0], e1[1]);
              }
            }
            if (!PV8) pf[qb][kb * 2 + half] = pk.v;
          }
        }
      }
      bool ok = true;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { psum[qb] = ps[qb][0] + ps[qb][1]; ok = ok && (psum[qb] <= SMAX); }
      if (exact || __all(ok)) break;
      exact = true;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];

    if (PV8) {
      // O^T += (V8^T P8^T) * 2^(sv - 3): ONE MFMA per (32 d x 32 queries) for the whole 64-key tile
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const i32x4 a0 = *(const i32x4*)(smem + v8_ptr + db * 32 * 64 + v8_c0), a1 = *(const i32x4*)(smem + v8_ptr + db * 32 * 64 + v8_c1);
        const i32x8 vf = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const int vsc = *(const unsigned char*)(smem + vs_ptr + db * 32);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const i32x8 pb = i32x8{p8[qb][0], p8[qb][1], p8[qb][2], p8[qb][3], p8[qb][4], p8[qb][5], p8[qb][6], p8[qb][7]};
          o[qb][db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pb, o[qb][db], 0, 0, 0, vsc, 0, 127 - (int)P8_SHIFT);
        }
      }
    } else {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 vf = *(const bf16x8*)(smem + v_ptr[ks] + db * 32 * 128);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], o[qb][db], 0, 0, 0);
        }
    }
    k_ptr += stage_step; s_ptr += stage_step; v8_ptr += stage_step; vs_ptr += stage_step;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) v_ptr[ks] += stage_step;
    stage_step = -stage_step;
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = (PV8 ? 8.0f : 1.0f) / l_tot;       // l sums exp2(score + 3) for PV8
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

// workspace of the V^T e4m3 tiles + their scales: [B*H*ntile][64][64] bytes, then [B*H*ntile][64] + 256 (DMA overshoot)
extern "C" int64_t hi3d_attn_fp8_v_workspace_bytes(int32_t B, int32_t H, int32_t S) {
  const int64_t tiles = (int64_t)B * H * (((int64_t)S + 63) / 64);
  return tiles * 4096 + (tiles * 64 + 256 + 255) / 256 * 256;
}

extern "C" int hi3d_attn_quant_v(const void* v, void* ws_v, int32_t B, int32_t H, int32_t S, int32_t ldv, void* stream) {
  if (!v || !ws_v) HI3D_FAIL(HI3D_EINVAL, "attn_quant_v: null pointer");
  if (B <= 0 || H <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, "attn_quant_v: non-positive size");
  if (ldv < H * 64 || ldv % 8) HI3D_FAIL(HI3D_EALIGN, "attn_quant_v: ldv must cover the heads and keep 16-byte rows");
  if (((uintptr_t)v & 15) || ((uintptr_t)ws_v & 255)) HI3D_FAIL(HI3D_EALIGN, "attn_quant_v: misaligned pointer");
  if (H > 65535 || B > 65535) HI3D_FAIL(HI3D_ESHAPE, "attn_quant_v: grid too large");
  const int ntile = (S + 63) / 64;
  unsigned char* v8 = (unsigned char*)ws_v;
  unsigned char* vs = v8 + (long)B * H * ntile * 4096;
  hipLaunchKernelGGL(quant_vt_kernel, dim3(ntile, H, B), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)v, v8, vs, H, S, ldv);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

static int attn8_params(Attn8Params& p, const void* ws, const void* vt, void* out, int32_t B, int32_t H, int32_t S, int32_t ldo, const char* who) {
  if (!ws || !vt || !out) HI3D_FAIL(HI3D_EINVAL, who);
  if (B <= 0 || H <= 0 || S <= 0) HI3D_FAIL(HI3D_EINVAL, who);
  if (ldo < H * 64 || ldo % 4) HI3D_FAIL(HI3D_EALIGN, who);
  if (((uintptr_t)ws & 255) || ((uintptr_t)vt & 15) || ((uintptr_t)out & 7)) HI3D_FAIL(HI3D_EALIGN, who);
  const int S_pad = (S + 63) / 64 * 64;
  const long rows = (long)B * H * S_pad;
  const long ssec = (rows * 2 + 256 + 255) / 256 * 256;
  p.q8 = (const char*)ws; p.k8 = p.q8 + rows * 64;
  p.qs = (const unsigned char*)(p.k8 + rows * 64); p.ks = p.qs + ssec;
  p.vt = (const char*)vt; p.vsc = nullptr; p.out = (unsigned short*)out;
  p.B = B; p.H = H; p.Sq = S; p.Skv = S; p.Spq = S_pad; p.Spk = S_pad; p.ldvt = S_pad; p.ldo = ldo;
  p.nqt = (S + Q_TILE - 1) / Q_TILE;
  if ((long)p.nqt * H * B > 0x7fffffffL) HI3D_FAIL(HI3D_ESHAPE, who);
  return HI3D_OK;
}

// both products on the fp8 matrix path: ws from hi3d_attn_quant_qk, ws_v from hi3d_attn_quant_v
extern "C" int hi3d_attn_d64_fp8(const void* ws, const void* ws_v, void* out, int32_t B, int32_t H, int32_t S, int32_t ldo, void* stream) {
  Attn8Params p;
  if (((uintptr_t)ws_v & 255)) HI3D_FAIL(HI3D_EALIGN, "attn_d64_fp8: misaligned V workspace");
  const int rc = attn8_params(p, ws, ws_v, out, B, H, S, ldo, "attn_d64_fp8: bad argument");
  if (rc) return rc;
  p.vsc = (const unsigned char*)ws_v + (long)B * H * ((S + 63) / 64) * 4096;
  hipLaunchKernelGGL(attn_d64_fp8_kernel<true>, dim3((unsigned)((long)p.nqt * H * B)), dim3(256), 0, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

extern "C" int hi3d_attn_d64_fp8qk(const void* ws, const void* vt, void* out, int32_t B, int32_t H, int32_t S,
                                   int32_t ld_vt, int32_t ldo, void* stream) {
  const int S_pad = (S + 63) / 64 * 64;
  if (ld_vt != S_pad) HI3D_FAIL(HI3D_ESHAPE, "attn_d64_fp8qk: ld_vt must be S rounded up to 64");
  Attn8Params p;
  const int rc = attn8_params(p, ws, vt, out, B, H, S, ldo, "attn_d64_fp8qk: bad argument");
  if (rc) return rc;
  hipLaunchKernelGGL(attn_d64_fp8_kernel<false>, dim3((unsigned)((long)p.nqt * H * B)), dim3(256), 0, (hipStream_t)stream, p);
  HI3D_LAUNCH_CHECK();
  return HI3D_OK;
}

// Does VALU work issue under a running MFMA on gfx950?  One wave per SIMD (256-thread blocks, 1 per CU);
// the instruction stream is fixed by inline asm: 4 independent v_mfma_f32_32x32x16_bf16, each followed by NF
// independent VALU "fillers" of one kind.  Prints time per MFMA relative to the bare-MFMA stream.
// A second mode runs TWO waves per SIMD (512-thread blocks): waves 0-3 bare MFMA, waves 4-7 pure VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define F_EXP(r) "v_exp_f32 %" #r ", %" #r "\n"
#define F_FMA(r) "v_fma_f32 %" #r ", %" #r ", %14, %15\n"
#define F_ADD(r) "v_add_f32 %" #r ", %" #r ", %14\n"
#define F_CVT(r) "v_cvt_pk_bf16_f32 %" #r ", %" #r ", %14\n"
#define MF(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %12, %13, %" #acc "\n"
// fillers use operands 4..11 (e0..e7)
#define FILL0(F)
#define FILL1(F) F(4)
#define FILL2(F) F(4) F(5)
#define FILL3(F) F(4) F(5) F(6)
#define FILL4(F) F(4) F(5) F(6) F(7)
#define FILL5(F) F(4) F(5) F(6) F(7) F(8)
#define FILL6(F) F(4) F(5) F(6) F(7) F(8) F(9)
#define FILL8(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
#define FILL12(F) FILL8(F) FILL4(F)
#define FILL16(F) FILL8(F) FILL8(F)

#define KERNEL(NAME, FILL, F)                                                                         \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                \
    f32x16 a0, a1, a2, a3;                                                                            \
    for (int j = 0; j < 16; ++j) { a0[j] = a1[j] = a2[j] = a3[j] = 0.f; }                             \
    bf16x8 x = (bf16x8)(short)(0x3c00 + threadIdx.x), y = (bf16x8)(short)(0x3b80 + threadIdx.x);     \
    float e0 = threadIdx.x * 1e-3f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7; \
    float c1 = 0.999f, c2 = 0.001f;                                                                   \
    for (int it = 0; it < iters; ++it) {                                                              \
      asm volatile(MF(0) FILL(F) MF(1) FILL(F) MF(2) FILL(F) MF(3) FILL(F)                            \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),  \
                     "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)                                           \
                   : "v"(x), "v"(y), "v"(c1), "v"(c2));                                               \
    }                                                                                                 \
    float s = a0[0] + a1[3] + a2[5] + a3[7] + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;                  \
    if (s == 1234.5f) out[0] = s;                                                                     \
  }

KERNEL(k_bare, FILL0, F_EXP)
KERNEL(k_exp1, FILL1, F_EXP) KERNEL(k_exp2, FILL2, F_EXP) KERNEL(k_exp3, FILL3, F_EXP) KERNEL(k_exp4, FILL4, F_EXP)
KERNEL(k_exp5, FILL5, F_EXP) KERNEL(k_exp6, FILL6, F_EXP) KERNEL(k_exp8, FILL8, F_EXP)
KERNEL(k_fma2, FILL2, F_FMA) KERNEL(k_fma4, FILL4, F_FMA) KERNEL(k_fma5, FILL5, F_FMA) KERNEL(k_fma6, FILL6, F_FMA)
KERNEL(k_fma8, FILL8, F_FMA) KERNEL(k_fma12, FILL12, F_FMA) KERNEL(k_fma16, FILL16, F_FMA)
KERNEL(k_add4, FILL4, F_ADD) KERNEL(k_add8, FILL8, F_ADD)
KERNEL(k_cvt4, FILL4, F_CVT) KERNEL(k_cvt8, FILL8, F_CVT)

// pure VALU streams (no MFMA) for the solo cost of the fillers
#define VKERNEL(NAME, F)                                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                \
    float e0 = threadIdx.x * 1e-3f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7; \
    float c1 = 0.999f, c2 = 0.001f; f32x16 d0, d1, d2, d3; bf16x8 x = (bf16x8)(short)1, y = x;       \
    for (int j = 0; j < 16; ++j) { d0[j] = d1[j] = d2[j] = d3[j] = 0.f; }                             \
    for (int it = 0; it < iters; ++it) {                                                              \
      asm volatile(FILL8(F) FILL8(F) FILL8(F) FILL8(F)                                                \
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),  \
                     "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)                                           \
                   : "v"(x), "v"(y), "v"(c1), "v"(c2));                                               \
    }                                                                                                 \
    float s = e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + d0[0];                                          \
    if (s == 1234.5f) out[0] = s;                                                                     \
  }
VKERNEL(v_exp, F_EXP) VKERNEL(v_fma, F_FMA) VKERNEL(v_add, F_ADD) VKERNEL(v_cvt, F_CVT)

// two waves per SIMD: waves 0-3 bare MFMA stream, waves 4-7 a pure VALU stream of NV x 32 fillers per iteration
template <int KIND>
__global__ __launch_bounds__(512) void k_two(float* out, int iters, int valu_iters) {
  const int w = threadIdx.x >> 6;
  f32x16 a0, a1, a2, a3;
  for (int j = 0; j < 16; ++j) { a0[j] = a1[j] = a2[j] = a3[j] = 0.f; }
  bf16x8 x = (bf16x8)(short)(0x3c00 + threadIdx.x), y = (bf16x8)(short)(0x3b80 + threadIdx.x);
  float e0 = threadIdx.x * 1e-3f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7;
  float c1 = 0.999f, c2 = 0.001f;
  if (w < 4) {
    for (int it = 0; it < iters; ++it)
      asm volatile(MF(0) MF(1) MF(2) MF(3)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                   : "v"(x), "v"(y), "v"(c1), "v"(c2));
  } else {
    for (int it = 0; it < valu_iters; ++it) {
      if (KIND == 0)
        asm volatile(FILL8(F_EXP) FILL8(F_EXP) FILL8(F_EXP) FILL8(F_EXP)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                     : "v"(x), "v"(y), "v"(c1), "v"(c2));
      else
        asm volatile(FILL8(F_FMA) FILL8(F_FMA) FILL8(F_FMA) FILL8(F_FMA)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                     : "v"(x), "v"(y), "v"(c1), "v"(c2));
    }
  }
  float s = a0[0] + a1[3] + a2[5] + a3[7] + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
  if (s == 1234.5f) out[0] = s;
}

template <typename K> float run(K kern, int threads, float* out, int a, int b = -1) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  float ms = 0, best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(s);
    if (b < 0) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, a);
    hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
    if (rep && ms < best) best = ms;
  }
  return best;
}
int main() {
  float* out; hipMalloc(&out, 4);
  const int it = 20000;   // 4 MFMA per iteration
  const float base = run(k_bare, 256, out, it);
  printf("bare 4 MFMA/iter                 : %.3f ms = 1.00 (32 cyc/MFMA if the pipe is the limit)\n", base);
#define R(K, nf, kind) { float t = run(K, 256, out, it); printf("%-6s x%2d per MFMA               : %.3f ms = %.2f x bare -> %.1f cyc/MFMA\n", kind, nf, t, t / base, 32.0 * t / base); }
  R(k_exp1, 1, "exp") R(k_exp2, 2, "exp") R(k_exp3, 3, "exp") R(k_exp4, 4, "exp") R(k_exp5, 5, "exp") R(k_exp6, 6, "exp") R(k_exp8, 8, "exp")
  R(k_fma2, 2, "fma") R(k_fma4, 4, "fma") R(k_fma5, 5, "fma") R(k_fma6, 6, "fma") R(k_fma8, 8, "fma") R(k_fma12, 12, "fma") R(k_fma16, 16, "fma")
  R(k_add4, 4, "add") R(k_add8, 8, "add") R(k_cvt4, 4, "cvtpk") R(k_cvt8, 8, "cvtpk")
#define V(K, kind) { float t = run(K, 256, out, it); printf("solo %-6s 32/iter                : %.3f ms -> %.1f cyc each (bare-MFMA clock)\n", kind, t, 128.0 * t / base / 32.0); }
  V(v_exp, "exp") V(v_fma, "fma") V(v_add, "add") V(v_cvt, "cvtpk")
  // two waves per SIMD
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int kind = 0; kind < 2; ++kind)
    for (int vi : {0, it / 4, it / 2, it}) {
      float ms = 0, best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s);
        if (kind == 0) hipLaunchKernelGGL(k_two<0>, dim3(256), dim3(512), 0, 0, out, it, vi);
        else hipLaunchKernelGGL(k_two<1>, dim3(256), dim3(512), 0, 0, out, it, vi);
        hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms, s, e);
        if (rep && ms < best) best = ms;
      }
      printf("two waves/SIMD: MFMA wave + %s wave with %5d x 32 fillers: %.3f ms = %.2f x bare\n", kind ? "fma" : "exp", vi, best, best / base);
    }
  return 0;
}

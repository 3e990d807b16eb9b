// Shared device helpers for the gfx950 kernels of libhi3d_hip.so.
// Written for CDNA4 only: 64-wide wavefronts, bf16 MFMA, LDS-DMA (global_load_lds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hi3d_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// 256 B of zeros in device memory: out-of-range gather lanes of an LDS-DMA read
// this instead of branching (global_load_lds has no predicated-zero form).
static __device__ uint4 hi3d_zero_page[16];  // zero-initialised, one copy per TU

__device__ __forceinline__ float bf16_to_f32(unsigned short u) {
  return __uint_as_float(((unsigned int)u) << 16);
}
// fp32 -> bf16 through the gfx950 hardware converter (v_cvt_pk_bf16_f32: round to
// nearest even, NaN preserved); written as native __bf16 casts so hipcc emits it.
typedef __attribute__((ext_vector_type(2))) __bf16 hi3d_bf2;
typedef __attribute__((ext_vector_type(2))) float hi3d_f2;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const hi3d_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hi3d_bf2));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return (unsigned short)(pack_bf16x2(f, 0.0f) & 0xffffu);
}
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}
// exact-erf GELU (F.gelu default) with erf from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 output step); ~12 VALU ops instead of erff's ~40.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = 1.0f - poly * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

// two GELUs at once on packed fp32 (v_pk_mul/fma_f32): halves the VALU work of the GEGLU epilogue
__device__ __forceinline__ hi3d_f2 gelu_erf_f2(hi3d_f2 x) {
  hi3d_f2 z = {fabsf(x[0]), fabsf(x[1])};
  z = z * 0.70710678118654752f;
  const hi3d_f2 d = z * 0.3275911f + 1.0f;
  const hi3d_f2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  hi3d_f2 poly = t * 1.061405429f - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const hi3d_f2 a = z * z * -1.4426950408889634f;
  const hi3d_f2 ex = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const hi3d_f2 e = 1.0f - poly * t * ex;
  const hi3d_f2 se = {copysignf(e[0], x[0]), copysignf(e[1], x[1])};
  return x * 0.5f * (se + 1.0f);
}

// 16-byte LDS-DMA: each lane supplies its own global source, the LDS destination
// is the wave-uniform `lds_wave_base` + lane*16 (hardware adds the lane part).
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// host side ---------------------------------------------------------------
void hi3d_set_error(const char* msg);
#define HI3D_FAIL(code, msg) do { hi3d_set_error(msg); return (code); } while (0)
#define HI3D_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) {                                   \
      hi3d_set_error(hipGetErrorString(e_));                  \
      return (int)e_;                                         \
    }                                                         \
  } while (0)

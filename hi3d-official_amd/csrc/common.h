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

// Two GELUs at once on packed fp32, for the GEGLU epilogue (where VALU time adds to MFMA time).
// x * Phi(x) with Phi(x) = sigmoid(x * (c0 + c1 x^2 + c2 x^4)); coefficients fitted to the exact
// erf form over [-9, 9]: max abs error 2.6e-5 (the output is rounded to bf16, relative step
// 3.9e-3).  6 packed ops + 2 min + 2 exp2 + 2 rcp per pair -- the Abramowitz-Stegun erf above
// needs 14 packed ops, the same four transcendentals and two bit-field inserts.
// x^2 is clamped at 64: beyond |x| = 8 the sigmoid is saturated and the quartic must not turn.
// The coefficients carry the factor -log2(e) so that the exponential is a bare v_exp_f32.
__device__ __forceinline__ hi3d_f2 gelu_erf_f2(hi3d_f2 x) {
  hi3d_f2 x2 = x * x;
  x2[0] = fminf(x2[0], 64.0f); x2[1] = fminf(x2[1], 64.0f);
  hi3d_f2 inner = x2 * 1.01536637e-3f - 1.06782578e-1f;     // -log2(e) * (c2 x^2 + c1)
  inner = inner * x2 - 2.30111379f;                         // -log2(e) * c0
  const hi3d_f2 t = x * inner;
  const hi3d_f2 d = hi3d_f2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
  return x * hi3d_f2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// 16-byte LDS-DMA: each lane supplies its own global source, the LDS destination
// is the wave-uniform `lds_wave_base` + lane*16 (hardware adds the lane part).
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// 8-byte LDS store that the compiler does not order against in-flight LDS-DMA.  The waitcnt pass puts `s_waitcnt vmcnt(0)`
// in front of every ds_write while `buffer_load ... lds` requests are outstanding (it cannot tell that the DMA targets and
// the store are different slabs) -- which drains a kernel's whole weight pipeline at each such store.  Kernels that keep
// streams in flight across a store use this form and retire it with their own `s_waitcnt lgkmcnt(0)` before the barrier.
__device__ __forceinline__ void lds_store_b64_nodrain(void* lds_ptr, unsigned int lo, unsigned int hi) {
  typedef unsigned int u2_ __attribute__((ext_vector_type(2)));
  const unsigned int a = (unsigned int)(__UINTPTR_TYPE__)(LDS_AS char*)lds_ptr;
  asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(u2_{lo, hi}) : "memory");
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
// hipFuncSetAttribute is a per-device property: a kernel that needs more than 64 KiB of dynamic LDS
// raises its limit once per (kernel, device).  `done` is the caller's static per-kernel table.
constexpr int HI3D_MAX_DEVICES = 64;
inline int hi3d_raise_lds_limit(const void* fn, int bytes, bool (&done)[HI3D_MAX_DEVICES]) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { hi3d_set_error(hipGetErrorString(e)); return (int)e; }
  const bool tracked = dev >= 0 && dev < HI3D_MAX_DEVICES;
  if (tracked && done[dev]) return 0;              // benign race: idempotent
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { hi3d_set_error(hipGetErrorString(e)); return (int)e; }
  if (tracked) done[dev] = true;
  return 0;
}
#define HI3D_FAIL(code, msg) do { hi3d_set_error(msg); return (code); } while (0)
#define HI3D_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) {                                   \
      hi3d_set_error(hipGetErrorString(e_));                  \
      return (int)e_;                                         \
    }                                                         \
  } while (0)

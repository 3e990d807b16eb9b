// Which K elements does a lane's e8m0 scale byte apply to in v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 operands)?
//   H1: a lane's 32 bytes (8 VGPRs) are ONE scale block, scaled by that lane's own scale byte
//   H2: VGPRs 0-3 of both lane halves form block 0 (scale from lanes 0-31), VGPRs 4-7 block 1 (lanes 32-63)
// A = 1.0 (e4m3 0x38) only in VGPRs 4-7 of lanes 0-31, B = 1.0 everywhere; scale_a: lanes 0-31 -> 2^0,
// lanes 32-63 -> 2^1; scale_b = 2^0.  D[0][0] = 16 under H1, 32 under H2.  Second case: A only in VGPRs 0-3 of
// lanes 32-63: 32 under H1 (own scale 2^1), 16 under H2.
// build: hipcc --offload-arch=gfx950 -O2 mx_scale_layout.hip -o mx_scale_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float* out, int which) {
  const int lane = threadIdx.x, g = lane >> 5;
  i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
  const int one4 = 0x38383838;
  for (int i = 0; i < 8; ++i) b[i] = one4;
  if (which == 0 && g == 0) { a[4] = a[5] = a[6] = a[7] = one4; }
  if (which == 1 && g == 1) { a[0] = a[1] = a[2] = a[3] = one4; }
  if (which == 2) { for (int i = 0; i < 8; ++i) a[i] = one4; }            // all ones: 32*s0 + 32*s1 = 96 either way
  const int sa = 127 + g, sb = 127;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  if (lane == 0) out[which] = c[0];
}

int main() {
  float* d; hipMalloc(&d, 16);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, w);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("case0 (A in VGPR4-7 of lanes 0-31): %g  -> %s\n", h[0], h[0] == 16.f ? "H1 (per-lane block)" : h[0] == 32.f ? "H2 (register-half blocks)" : "?");
  printf("case1 (A in VGPR0-3 of lanes 32-63): %g -> %s\n", h[1], h[1] == 32.f ? "H1" : h[1] == 16.f ? "H2" : "?");
  printf("case2 (all ones): %g (expect 96)\n", h[2]);
  return 0;
}

// Does an MFMA whose destination overlaps, but is not, its C operand run slower?  (The register
// allocator produces such "rotating" accumulators; the fused feed-forward kernel hit them.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
__global__ __launch_bounds__(256) void k(int mode, float* out) {
#if __HIP_DEVICE_COMPILE__
  if (mode == 0) {
    for (int it = 0; it < N; ++it)
      asm volatile(
          "v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"
          "v_mfma_f32_16x16x32_bf16 a[4:7], v[0:3], v[4:7], a[4:7]\n"
          "v_mfma_f32_16x16x32_bf16 a[8:11], v[0:3], v[4:7], a[8:11]\n"
          "v_mfma_f32_16x16x32_bf16 a[12:15], v[0:3], v[4:7], a[12:15]\n"
          "v_mfma_f32_16x16x32_bf16 a[16:19], v[0:3], v[4:7], a[16:19]\n"
          "v_mfma_f32_16x16x32_bf16 a[20:23], v[0:3], v[4:7], a[20:23]\n"
          "v_mfma_f32_16x16x32_bf16 a[24:27], v[0:3], v[4:7], a[24:27]\n"
          "v_mfma_f32_16x16x32_bf16 a[28:31], v[0:3], v[4:7], a[28:31]\n" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33");
  } else {
    // destination = C operand shifted down by two registers (what the allocator emitted)
    for (int it = 0; it < N; ++it)
      asm volatile(
          "v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[2:5]\n"
          "v_mfma_f32_16x16x32_bf16 a[4:7], v[0:3], v[4:7], a[6:9]\n"
          "v_mfma_f32_16x16x32_bf16 a[8:11], v[0:3], v[4:7], a[10:13]\n"
          "v_mfma_f32_16x16x32_bf16 a[12:15], v[0:3], v[4:7], a[14:17]\n"
          "v_mfma_f32_16x16x32_bf16 a[16:19], v[0:3], v[4:7], a[18:21]\n"
          "v_mfma_f32_16x16x32_bf16 a[20:23], v[0:3], v[4:7], a[22:25]\n"
          "v_mfma_f32_16x16x32_bf16 a[24:27], v[0:3], v[4:7], a[26:29]\n"
          "v_mfma_f32_16x16x32_bf16 a[28:31], v[0:3], v[4:7], a[30:33]\n" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33");
  }
  if (threadIdx.x == 9999) out[0] = 1.f;
#endif
}
int main() {
  float* out; hipMalloc(&out, 4);
  const char* names[] = {"dst == C (accumulate in place)", "dst = C shifted by 2 registers (overlapping)"};
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, mode, out);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("%-48s: %.3f ms -> %.1f cycles per MFMA @2.4GHz (8 independent accumulators, 1 wave/SIMD)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / (N * 8.0));
  }
  return 0;
}

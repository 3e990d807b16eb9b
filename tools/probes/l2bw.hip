// L2 -> LDS streaming bandwidth probe: the GEMM loader's access pattern without the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_AS __attribute__((address_space(3)))
__global__ __launch_bounds__(256) void stream(const char* base, long region, long span, int steps, int rows, int pitch, int* sink, int rowsW, int pitchW, int nW) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // each block streams `steps` stages of `rows` rows x 128 B (row pitch `pitch`), starting at a block-specific origin
  const char* origin = base + ((long)blockIdx.x * region) % span;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, 0x7fffffff, 0x00020000);
  const int npieces = rows / 8;              // 1 KiB pieces (8 rows x 128 B)
  const char* worigin = base + span + (long)(blockIdx.x % nW) * rowsW * pitchW;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)worigin, 0, 0x7fffffff, 0x00020000);
  const int npw = rowsW / 8;
  int st = 0;
  for (int k = 0; k < steps; ++k) {
    for (int q = w; q < npieces; q += 4) {
      const unsigned vo = (unsigned)((q * 8 + (lane >> 3)) * pitch + (lane & 7) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + st * (rows + rowsW) * 128 + q * 1024), 16, vo, k * 128, 0, 0);
    }
    for (int q = w; q < npw; q += 4) {
      const unsigned vo = (unsigned)((q * 8 + (lane >> 3)) * pitchW + (lane & 7) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (LDS_AS void*)(smem + st * (rows + rowsW) * 128 + (npieces + q) * 1024), 16, vo, k * 128, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    st ^= 1;
  }
  if (tid == 0 && smem[5] == 77) sink[0] = 1;
#endif
}
int main(int argc, char** argv) {
  const long span = atol(argv[1]);          // bytes of the buffer actually touched
  const int rows = atoi(argv[2]);           // rows per stage (288 = GEMM 128x160)
  const int steps = atoi(argv[3]);
  const int pitch = atoi(argv[4]);          // row pitch in bytes (>= 128*steps)
  const int blocks_per_cu = atoi(argv[5]);
  const int rowsW = atoi(argv[6]), pitchW = atoi(argv[7]), nW = atoi(argv[8]);
  char* buf; int* sink;
  hipMalloc(&buf, span + (1l << 30)); hipMemset(buf, 1, span + (1l << 30)); hipMalloc(&sink, 4);
  const int lds = 2 * (rows + rowsW) * 128;
  hipFuncSetAttribute((const void*)stream, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256 * blocks_per_cu * 16;
  const long region = (long)rows * pitch;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(stream, dim3(grid), dim3(256), lds, 0, buf, region, span, steps, rows, pitch, sink, rowsW, pitchW, nW);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * steps * (rows + rowsW) * 128;
    if (it == 2) printf("span %6.1f MB rowsA %d pitchA %d rowsW %d pitchW %d nW %d steps %d bpc %d: %.3f ms  %.2f TB/s\n", span / 1e6, rows, pitch, rowsW, pitchW, nW, steps, blocks_per_cu, ms, bytes / ms / 1e9);
  }
  return 0;
}

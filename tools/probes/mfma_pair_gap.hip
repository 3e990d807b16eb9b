// Probe distilled from the round-2 attention incident (tools/attn_lab/): two v_mfma_f32_32x32x16_bf16 of ONE wave
// that share their A operand registers, with idle wait states between them, while a SECOND wave of the same SIMD issues
// MFMAs of its own.  In the failing kernel a gap of >= 16 wait states between the two (an s_waitcnt stall, an instruction
// fetch delay) made 16 of the 32 result columns of one of them wrong -- only with two waves per SIMD.
//
// Every wave works on its own operands (A, B1, B2 depend on the wave id), computes the reference with the pair issued
// back to back (never seen failing), then runs `iters` times:   MFMA1 ; s_nop * gap ; MFMA2   and counts result registers
// that differ from the reference, per 16-lane group, separately for MFMA1 and MFMA2.
// Register numbers are those of the failing kernel (D1 = v[80:95], D2 = v[64:79], A = v[112:115], B1 = v[132:135],
// B2 = v[148:151]); variants: srcC = inline 0 / a zeroed VGPR tuple, shared / separate A registers.
//     hipcc --offload-arch=gfx950 -O2 mfma_pair_gap.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned int u32;

#define GAP_0 ""
#define GAP_4 "s_nop 3\n"
#define GAP_8 "s_nop 7\n"
#define GAP_16 "s_nop 15\n"
#define GAP_32 "s_nop 15\n s_nop 15\n"
#define GAP_64 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"

#define C_LIT "0"
#define C_REG1 "v[160:175]"
#define C_REG2 "v[176:191]"
#define A2_SAME "v[112:115]"
#define A2_COPY "v[116:119]"

#define PAIR_BODY(GAP, C1, C2, A2)                                                                       \
  asm volatile(                                                                                         \
      "v_mov_b32 v112, %[a0]\n v_mov_b32 v113, %[a1]\n v_mov_b32 v114, %[a2]\n v_mov_b32 v115, %[a3]\n"    \
      "v_mov_b32 v116, %[a0]\n v_mov_b32 v117, %[a1]\n v_mov_b32 v118, %[a2]\n v_mov_b32 v119, %[a3]\n"    \
      "v_mov_b32 v132, %[b0]\n v_mov_b32 v133, %[b1]\n v_mov_b32 v134, %[b2]\n v_mov_b32 v135, %[b3]\n"    \
      "v_mov_b32 v148, %[c0]\n v_mov_b32 v149, %[c1]\n v_mov_b32 v150, %[c2]\n v_mov_b32 v151, %[c3]\n"    \
      "v_mov_b32 v160, 0\n v_mov_b32 v161, 0\n v_mov_b32 v162, 0\n v_mov_b32 v163, 0\n v_mov_b32 v164, 0\n v_mov_b32 v165, 0\n v_mov_b32 v166, 0\n v_mov_b32 v167, 0\n" \
      "v_mov_b32 v168, 0\n v_mov_b32 v169, 0\n v_mov_b32 v170, 0\n v_mov_b32 v171, 0\n v_mov_b32 v172, 0\n v_mov_b32 v173, 0\n v_mov_b32 v174, 0\n v_mov_b32 v175, 0\n" \
      "v_mov_b32 v176, 0\n v_mov_b32 v177, 0\n v_mov_b32 v178, 0\n v_mov_b32 v179, 0\n v_mov_b32 v180, 0\n v_mov_b32 v181, 0\n v_mov_b32 v182, 0\n v_mov_b32 v183, 0\n" \
      "v_mov_b32 v184, 0\n v_mov_b32 v185, 0\n v_mov_b32 v186, 0\n v_mov_b32 v187, 0\n v_mov_b32 v188, 0\n v_mov_b32 v189, 0\n v_mov_b32 v190, 0\n v_mov_b32 v191, 0\n" \
      "s_nop 15\n s_nop 15\n"                                                                           \
      "v_mfma_f32_32x32x16_bf16 v[80:95], v[112:115], v[132:135], " C1 "\n"                             \
      GAP                                                                                               \
      "v_mfma_f32_32x32x16_bf16 v[64:79], " A2 ", v[148:151], " C2 "\n"                                 \
      "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                     \
      "v_mov_b32 %[o0], v64\n v_mov_b32 %[o1], v65\n v_mov_b32 %[o2], v66\n v_mov_b32 %[o3], v67\n"       \
      "v_mov_b32 %[o4], v68\n v_mov_b32 %[o5], v69\n v_mov_b32 %[o6], v70\n v_mov_b32 %[o7], v71\n"       \
      "v_mov_b32 %[o8], v72\n v_mov_b32 %[o9], v73\n v_mov_b32 %[o10], v74\n v_mov_b32 %[o11], v75\n"     \
      "v_mov_b32 %[o12], v76\n v_mov_b32 %[o13], v77\n v_mov_b32 %[o14], v78\n v_mov_b32 %[o15], v79\n"   \
      "v_mov_b32 %[p0], v80\n v_mov_b32 %[p1], v81\n v_mov_b32 %[p2], v82\n v_mov_b32 %[p3], v83\n"       \
      "v_mov_b32 %[p4], v84\n v_mov_b32 %[p5], v85\n v_mov_b32 %[p6], v86\n v_mov_b32 %[p7], v87\n"       \
      "v_mov_b32 %[p8], v88\n v_mov_b32 %[p9], v89\n v_mov_b32 %[p10], v90\n v_mov_b32 %[p11], v91\n"     \
      "v_mov_b32 %[p12], v92\n v_mov_b32 %[p13], v93\n v_mov_b32 %[p14], v94\n v_mov_b32 %[p15], v95\n"   \
      : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]),     \
        [o6] "=&v"(o[6]), [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), \
        [o12] "=&v"(o[12]), [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), [o15] "=&v"(o[15]),                               \
        [p0] "=&v"(q[0]), [p1] "=&v"(q[1]), [p2] "=&v"(q[2]), [p3] "=&v"(q[3]), [p4] "=&v"(q[4]), [p5] "=&v"(q[5]),     \
        [p6] "=&v"(q[6]), [p7] "=&v"(q[7]), [p8] "=&v"(q[8]), [p9] "=&v"(q[9]), [p10] "=&v"(q[10]), [p11] "=&v"(q[11]), \
        [p12] "=&v"(q[12]), [p13] "=&v"(q[13]), [p14] "=&v"(q[14]), [p15] "=&v"(q[15])                                \
      : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z),        \
        [b3] "v"(b.w), [c0] "v"(c.x), [c1] "v"(c.y), [c2] "v"(c.z), [c3] "v"(c.w)                                       \
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v132", "v133", "v134", "v135", "v148", "v149", \
        "v150", "v151", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", \
        "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", \
        "v186", "v187", "v188", "v189", "v190", "v191")

#define DEF_KERNEL(NAME, GAP, C1, C2, A2)                                                               \
  __global__ __launch_bounds__(256, 2) void NAME(const uint4* __restrict__ ain, const uint4* __restrict__ bin, \
                                                 u32* __restrict__ bad, int iters) {                  \
    const int lane = threadIdx.x & 63;                                                                 \
    const int wv = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 63;      /* 64 operand sets */              \
    const uint4 a = ain[wv * 64 + lane], b = bin[wv * 64 + lane], c = bin[((wv + 17) & 63) * 64 + lane]; \
    float r2[16], r1[16], o[16], q[16];                                                                \
    { PAIR_BODY(GAP_0, C_LIT, C_LIT, A2_SAME); }                                                       \
    for (int r = 0; r < 16; ++r) { r2[r] = o[r]; r1[r] = q[r]; }                                       \
    u32 bad1 = 0, bad2 = 0;                                                                            \
    for (int it = 0; it < iters; ++it) {                                                               \
      { PAIR_BODY(GAP, C1, C2, A2); }                                                                  \
      bool m1 = false, m2 = false;                                                                     \
      for (int r = 0; r < 16; ++r) {                                                                   \
        m2 = m2 || (__float_as_uint(o[r]) != __float_as_uint(r2[r]));                                  \
        m1 = m1 || (__float_as_uint(q[r]) != __float_as_uint(r1[r]));                                  \
      }                                                                                                \
      bad1 += m1; bad2 += m2;                                                                          \
    }                                                                                                  \
    atomicAdd(&bad[lane], bad1);                                                                       \
    atomicAdd(&bad[64 + lane], bad2);                                                                  \
  }

DEF_KERNEL(lit_same_g0, GAP_0, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(lit_same_g4, GAP_4, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(lit_same_g8, GAP_8, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(lit_same_g16, GAP_16, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(lit_same_g32, GAP_32, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(lit_same_g64, GAP_64, C_LIT, C_LIT, A2_SAME)
DEF_KERNEL(reg_same_g16, GAP_16, C_REG1, C_REG2, A2_SAME)
DEF_KERNEL(reg_same_g32, GAP_32, C_REG1, C_REG2, A2_SAME)
DEF_KERNEL(reg_same_g64, GAP_64, C_REG1, C_REG2, A2_SAME)
DEF_KERNEL(lit_copy_g16, GAP_16, C_LIT, C_LIT, A2_COPY)
DEF_KERNEL(lit_copy_g32, GAP_32, C_LIT, C_LIT, A2_COPY)
DEF_KERNEL(lit_copy_g64, GAP_64, C_LIT, C_LIT, A2_COPY)
DEF_KERNEL(reg_copy_g32, GAP_32, C_REG1, C_REG2, A2_COPY)

typedef void (*kern_t)(const uint4*, const uint4*, u32*, int);
struct Test { const char* name; kern_t fn; };
#define T(N) {#N, N}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 64;
  Test tests[] = {T(lit_same_g0), T(lit_same_g4), T(lit_same_g8), T(lit_same_g16), T(lit_same_g32), T(lit_same_g64),
                  T(reg_same_g16), T(reg_same_g32), T(reg_same_g64), T(lit_copy_g16), T(lit_copy_g32), T(lit_copy_g64), T(reg_copy_g32)};
  const int n = 64 * 64 * 8;
  unsigned short* ha = (unsigned short*)malloc(n * 2); unsigned short* hb = (unsigned short*)malloc(n * 2);
  srand(11);
  auto bf = [](int v) { float f = (float)v; u32 u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
  for (int i = 0; i < n; ++i) { ha[i] = bf(rand() % 9 - 4); hb[i] = bf(rand() % 9 - 4); }
  uint4 *da, *db; u32* dbad;
  if (hipMalloc(&da, n * 2) != hipSuccess || hipMalloc(&db, n * 2) != hipSuccess || hipMalloc(&dbad, 128 * 4) != hipSuccess) return 1;
  (void)hipMemcpy(da, ha, n * 2, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, n * 2, hipMemcpyHostToDevice);
  printf("two MFMAs of one wave, second wave on the SIMD (blocks of 4 waves, 2 blocks per CU); %ld pairs per lane and test\n", (long)blocks * 4 * iters);
  printf("%-14s | MFMA1 wrong, lanes 0-15 16-31 32-47 48-63 | MFMA2 wrong, lanes 0-15 16-31 32-47 48-63\n", "test");
  for (const Test& t : tests) {
    (void)hipMemset(dbad, 0, 128 * 4);
    hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, da, db, dbad, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", t.name, hipGetErrorString(e)); return 1; }
    u32 h[128]; (void)hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < 128; ++l) g[l / 16] += h[l];
    printf("%-14s | %9lu %9lu %9lu %9lu | %9lu %9lu %9lu %9lu\n", t.name, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
  }
  return 0;
}

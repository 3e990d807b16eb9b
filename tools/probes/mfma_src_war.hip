// Probe for the round-2 attention incident: does v_mfma_f32_32x32x16_bf16 on gfx950 read its A / B
// operands again AFTER issue -- i.e. is a write to srcA / srcB shortly after the MFMA (by a following VALU instruction,
// or by the MFMA's own result when vDst overlaps srcA) a hazard that neither the hardware nor hipcc's hazard
// recognizer covers?  Every test issues F independent "filler" MFMAs, then the victim MFMA, then N wait states, then
// the write, and compares the victim's 16 result registers with a reference run of the same operands without the
// write.  Mismatches are counted per 16-lane group (columns 0-15 / 16-31 of the 32x32 result live in lanes 0-15,32-47 /
// 16-31,48-63).       build + run:  hipcc --offload-arch=gfx950 -O2 mfma_src_war.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned int u32;

#define MFMA_FILL "v_mfma_f32_32x32x16_bf16 v[112:127], v[104:107], v[108:111], 0\n"
#define FILL_0 ""
#define FILL_1 MFMA_FILL
#define FILL_2 MFMA_FILL MFMA_FILL
#define NOP_0 ""
#define NOP_1 "s_nop 0\n"
#define NOP_2 "s_nop 1\n"
#define NOP_4 "s_nop 3\n"
#define NOP_8 "s_nop 7\n"
#define NOP_16 "s_nop 15\n"

// victim: D = v[80:95], A = v[64:67], B = v[68:71];  WRITE is the instruction under test
#define PROBE_BODY(PRE, FILL, VICTIM, NOPS, WRITE)                                                       \
  asm volatile(                                                                                     \
      "v_mov_b32 v64, %[a0]\n v_mov_b32 v65, %[a1]\n v_mov_b32 v66, %[a2]\n v_mov_b32 v67, %[a3]\n"    \
      "v_mov_b32 v68, %[b0]\n v_mov_b32 v69, %[b1]\n v_mov_b32 v70, %[b2]\n v_mov_b32 v71, %[b3]\n"    \
      "v_mov_b32 v104, %[a0]\n v_mov_b32 v105, %[a1]\n v_mov_b32 v106, %[a2]\n v_mov_b32 v107, %[a3]\n" \
      "v_mov_b32 v108, %[b0]\n v_mov_b32 v109, %[b1]\n v_mov_b32 v110, %[b2]\n v_mov_b32 v111, %[b3]\n" \
      PRE "s_nop 15\n s_nop 15\n"                                                                   \
      FILL VICTIM NOPS WRITE                                                                        \
      "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                 \
      "v_mov_b32 %[o0], v80\n v_mov_b32 %[o1], v81\n v_mov_b32 %[o2], v82\n v_mov_b32 %[o3], v83\n"   \
      "v_mov_b32 %[o4], v84\n v_mov_b32 %[o5], v85\n v_mov_b32 %[o6], v86\n v_mov_b32 %[o7], v87\n"   \
      "v_mov_b32 %[o8], v88\n v_mov_b32 %[o9], v89\n v_mov_b32 %[o10], v90\n v_mov_b32 %[o11], v91\n" \
      "v_mov_b32 %[o12], v92\n v_mov_b32 %[o13], v93\n v_mov_b32 %[o14], v94\n v_mov_b32 %[o15], v95\n" \
      : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]),     \
        [o6] "=&v"(o[6]), [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]), [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), \
        [o12] "=&v"(o[12]), [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), [o15] "=&v"(o[15])                                \
      : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z),        \
        [b3] "v"(b.w), [poison] "v"(poison)                                                                             \
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
        "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
        "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")

#define V_PLAIN "v_mfma_f32_32x32x16_bf16 v[80:95], v[64:67], v[68:71], 0\n"
// vDst overlapping srcA at its first register (what hipcc emitted in the round-2 kernel: v[80:95] = mfma(v[80:83], ..., 0))
#define P_NONE ""
#define P_OVL_A "v_mov_b32 v80, v64\n v_mov_b32 v81, v65\n v_mov_b32 v82, v66\n v_mov_b32 v83, v67\n"
#define V_OVL_A "v_mfma_f32_32x32x16_bf16 v[80:95], v[80:83], v[68:71], 0\n"
// vDst overlapping srcA at registers 4..7 (the form inside the round-2 key loop, never seen failing)
#define P_OVL_A4 "v_mov_b32 v84, v64\n v_mov_b32 v85, v65\n v_mov_b32 v86, v66\n v_mov_b32 v87, v67\n"
#define V_OVL_A4 "v_mfma_f32_32x32x16_bf16 v[80:95], v[84:87], v[68:71], 0\n"
#define P_OVL_B "v_mov_b32 v80, v68\n v_mov_b32 v81, v69\n v_mov_b32 v82, v70\n v_mov_b32 v83, v71\n"
#define V_OVL_B "v_mfma_f32_32x32x16_bf16 v[80:95], v[64:67], v[80:83], 0\n"
#define W_NONE ""
#define W_A "v_mov_b32 v64, %[poison]\n"
#define W_A3 "v_mov_b32 v67, %[poison]\n"
#define W_B "v_mov_b32 v68, %[poison]\n"

#define DEF_KERNEL(NAME, PRE, FILL, VICTIM, NOPS, WRITE)                                                 \
  __global__ __launch_bounds__(256) void NAME(const uint4* __restrict__ ain, const uint4* __restrict__ bin, \
                                              u32* __restrict__ bad, int iters, u32 poison) {         \
    const int lane = threadIdx.x & 63;                                                                 \
    const uint4 a = ain[lane], b = bin[lane];                                                          \
    float ref[16], o[16];                                                                              \
    { PROBE_BODY(P_NONE, FILL_0, V_PLAIN, NOP_16, W_NONE); }                                                   \
    for (int r = 0; r < 16; ++r) ref[r] = o[r];                                                        \
    u32 nbad = 0;                                                                                      \
    for (int it = 0; it < iters; ++it) {                                                               \
      { PROBE_BODY(PRE, FILL, VICTIM, NOPS, WRITE); }                                                       \
      bool m = false;                                                                                  \
      for (int r = 0; r < 16; ++r) m = m || (__float_as_uint(o[r]) != __float_as_uint(ref[r]));        \
      nbad += m ? 1u : 0u;                                                                             \
    }                                                                                                  \
    atomicAdd(&bad[lane], nbad);                                                                       \
  }

#define DEF_WAR(T, W)                                                       \
  DEF_KERNEL(war_##T##_f0_n0, P_NONE, FILL_0, V_PLAIN, NOP_0, W)                    \
  DEF_KERNEL(war_##T##_f0_n1, P_NONE, FILL_0, V_PLAIN, NOP_1, W)                    \
  DEF_KERNEL(war_##T##_f0_n2, P_NONE, FILL_0, V_PLAIN, NOP_2, W)                    \
  DEF_KERNEL(war_##T##_f0_n4, P_NONE, FILL_0, V_PLAIN, NOP_4, W)                    \
  DEF_KERNEL(war_##T##_f0_n8, P_NONE, FILL_0, V_PLAIN, NOP_8, W)                    \
  DEF_KERNEL(war_##T##_f1_n0, P_NONE, FILL_1, V_PLAIN, NOP_0, W)                    \
  DEF_KERNEL(war_##T##_f1_n1, P_NONE, FILL_1, V_PLAIN, NOP_1, W)                    \
  DEF_KERNEL(war_##T##_f1_n2, P_NONE, FILL_1, V_PLAIN, NOP_2, W)                    \
  DEF_KERNEL(war_##T##_f1_n4, P_NONE, FILL_1, V_PLAIN, NOP_4, W)                    \
  DEF_KERNEL(war_##T##_f1_n8, P_NONE, FILL_1, V_PLAIN, NOP_8, W)                    \
  DEF_KERNEL(war_##T##_f1_n16, P_NONE, FILL_1, V_PLAIN, NOP_16, W)                  \
  DEF_KERNEL(war_##T##_f2_n0, P_NONE, FILL_2, V_PLAIN, NOP_0, W)                    \
  DEF_KERNEL(war_##T##_f2_n4, P_NONE, FILL_2, V_PLAIN, NOP_4, W)                    \
  DEF_KERNEL(war_##T##_f2_n8, P_NONE, FILL_2, V_PLAIN, NOP_8, W)                    \
  DEF_KERNEL(war_##T##_f2_n16, P_NONE, FILL_2, V_PLAIN, NOP_16, W)

DEF_KERNEL(ctl_plain_f0, P_NONE, FILL_0, V_PLAIN, NOP_0, W_NONE)
DEF_KERNEL(ctl_plain_f2, P_NONE, FILL_2, V_PLAIN, NOP_0, W_NONE)
DEF_WAR(a, W_A)
DEF_WAR(a3, W_A3)
DEF_WAR(b, W_B)
DEF_KERNEL(ovl_a_f0, P_OVL_A, FILL_0, V_OVL_A, NOP_0, W_NONE)
DEF_KERNEL(ovl_a_f1, P_OVL_A, FILL_1, V_OVL_A, NOP_0, W_NONE)
DEF_KERNEL(ovl_a_f2, P_OVL_A, FILL_2, V_OVL_A, NOP_0, W_NONE)
DEF_KERNEL(ovl_a4_f0, P_OVL_A4, FILL_0, V_OVL_A4, NOP_0, W_NONE)
DEF_KERNEL(ovl_a4_f1, P_OVL_A4, FILL_1, V_OVL_A4, NOP_0, W_NONE)
DEF_KERNEL(ovl_a4_f2, P_OVL_A4, FILL_2, V_OVL_A4, NOP_0, W_NONE)
DEF_KERNEL(ovl_b_f0, P_OVL_B, FILL_0, V_OVL_B, NOP_0, W_NONE)
DEF_KERNEL(ovl_b_f1, P_OVL_B, FILL_1, V_OVL_B, NOP_0, W_NONE)
DEF_KERNEL(ovl_b_f2, P_OVL_B, FILL_2, V_OVL_B, NOP_0, W_NONE)

typedef void (*kern_t)(const uint4*, const uint4*, u32*, int, u32);
struct Test { const char* name; kern_t fn; };
#define T(N) {#N, N}
#define T_WAR(X) T(war_##X##_f0_n0), T(war_##X##_f0_n1), T(war_##X##_f0_n2), T(war_##X##_f0_n4), T(war_##X##_f0_n8), \
    T(war_##X##_f1_n0), T(war_##X##_f1_n1), T(war_##X##_f1_n2), T(war_##X##_f1_n4), T(war_##X##_f1_n8), T(war_##X##_f1_n16), \
    T(war_##X##_f2_n0), T(war_##X##_f2_n4), T(war_##X##_f2_n8), T(war_##X##_f2_n16)

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 64;
  Test tests[] = {T(ctl_plain_f0), T(ctl_plain_f2), T_WAR(a), T_WAR(a3), T_WAR(b), T(ovl_a_f0), T(ovl_a_f1), T(ovl_a_f2),
                  T(ovl_a4_f0), T(ovl_a4_f1), T(ovl_a4_f2), T(ovl_b_f0), T(ovl_b_f1), T(ovl_b_f2)};
  // operands: small integers in bf16 (every product and sum exact in fp32, no order dependence)
  unsigned short ha[64 * 8], hb[64 * 8];
  srand(7);
  auto bf = [](int v) { float f = (float)v; u32 u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
  for (int i = 0; i < 64 * 8; ++i) { ha[i] = bf(rand() % 9 - 4); hb[i] = bf(rand() % 9 - 4); }
  uint4 *da, *db; u32* dbad;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dbad, 64 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  const u32 poison = 0x41004100u;   // bf16 pair (8.0, 8.0)
  printf("%-18s %10s | wrong results per 16-lane group (of %ld MFMAs per lane): lanes 0-15, 16-31, 32-47, 48-63\n", "test", "waves", (long)blocks * 4 * iters);
  for (const Test& t : tests) {
    hipMemset(dbad, 0, 64 * 4);
    hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, da, db, dbad, iters, poison);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", t.name, hipGetErrorString(e)); return 1; }
    u32 hbad[64]; hipMemcpy(hbad, dbad, sizeof(hbad), hipMemcpyDeviceToHost);
    unsigned long g[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) g[l / 16] += hbad[l];
    printf("%-18s %10d | %10lu %10lu %10lu %10lu\n", t.name, blocks * 4, g[0], g[1], g[2], g[3]);
  }
  return 0;
}

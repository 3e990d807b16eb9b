// Kernel argument and internal constants shared by the GEMM translation units (gemm.hip: the tile family;
// gemm_persist.hip: the persistent 256-row ping-pong tile).
#pragma once
#include "common.h"

namespace {

struct GemmParams {
  const char* A; const char* W; const float* bias; const float* rowvec;
  const unsigned short* R1; const unsigned short* R2; const float* a1; const float* a2;
  void* out;
  int M, N, K, lda, ldo, ldr1, ldr2, ldrv, ldw, rpg, out_fp32, vec8;
  int Hin, Win, Cin, Hout, Wout, stride, up2x, T, HW, pad;
  int nbm, nbn;
  int gn;      // tile raster: column groups of gn n-tiles, m walked inside a group (0 / >= nbn: one group = n fastest over the row)
  int abl;     // timing ablations (HI3D_GEMM_ABL; wrong results by design): 1 = output stores dropped, 2 = no epilogue at all
  // split-K (long-K launches with too few tiles for the chip: the 8x8 / 16x16 levels of stage 1, the ranks of a clip-parallel
  // job): the grid is ksplit x (nbm * nbn); block (ks, tile) accumulates K steps [ks * nk_split, (ks + 1) * nk_split) -- whole
  // channel slabs for the conv gathers, whose K walks the taps innermost -- into the fp32 partial tile ks of `out`
  // ([ksplit][M][ldo], no bias / residual); splitk_combine_kernel sums the partials in a fixed order and applies the epilogue.
  int ksplit, nk_split;
  // two-source dense A (internal amode A_DENSE2): columns [0, K1) of the logical A come from A (pitch lda), columns [K1, K) from
  // A2 (pitch lda2) -- the decoder's skip concat `th.cat([h, hs.pop()], dim=1)` (video_model.py:490-499) as two K segments
  // of the 1x1 skip_connection GEMM instead of a materialised [M, C1 + C2] tensor
  const char* A2; int lda2, K1;
  int ntap;    // conv3x3: K slabs per 64-channel block -- 9, or the length of a tap SUBSET (`taps`: 4-bit tap indices ky*3+kx,
  unsigned taps;   // first in the low nibble; W then holds [N][ntap][Cin]) -- the 2x2 phase filters of an up-sampling conv
  long wgs;    // per-row-group weight matrices: elements between the W of consecutive groups of rpg rows (0 = one shared W)
  // GroupNorm statistics of the OUTPUT, emitted by the producer (wide ping-pong tile, affine epilogue without residual / blend
  // terms; the host checks the geometry): per 64-row block of the output and per group of N / 32 channels the (sum, sum of
  // squares) of the fp32 results, at gn_part[(row / 64) * 64 + 2 * group + {0, 1}] -- the partial-sum layout gn_finalize
  // reads, so the consuming GroupNorm skips its statistics pass over the tensor (openaimodel.py:292-294 out_layers.0 after
  // in_layers.2; video_model.py time_stack likewise).  nullptr = off.
  float* gn_part;
  // 1: the launch adds residual / blend terms after its accumulators (R1 / R2 / a1 / a2): the statistics are taken in the store
  // loop instead, from the FINAL values as they are rounded to bf16 (round 6) -- the GroupNorm that reads a ResBlock's or a
  // transformer's output (openaimodel.py:328-354 in_layers.0; video_model.py:62-81 time_stack.in_layers.0; attention.py:702 norm)
  int gn_post;
  // Conv3d (3,1,1): a block whose rows lie in ONE frame at either end of its clip skips the K steps of the tap that reads the
  // clip's zero padding (round 6; 0 = walk all three taps, HI3D_CONVT_SKIP=0)
  int tskip;
  // A_CONV3X3_PHASE: the phase constant c = a * 2 Win + b of the up-sampled image this tap-subset launch writes into (see below)
  int phase_c;
};

constexpr int BK = 64;
constexpr int A_CONV3X3_UP2X = 3;   // internal: HI3D_A_CONV3X3 with up2x (own instantiation: the plain gather stays lean)
constexpr int A_CONV3X3_PHASE = 5;  // internal: HI3D_A_CONV3X3 (tap subset) whose output rows go to phase (a, b) of a 2x up-sampled image: row
                                    // p = (f Hin + i) Win + j is stored as row 2 p + 2 Win (p / Win) + c of `out`, c = a * 2 Win + b
                                    // (hi3d_gemm_desc.conv_phase; wide ping-pong tiles only: the per-wave store loop places the rows)
constexpr int A_DENSE2 = 4;         // internal: HI3D_A_DENSE with two K segments from two tensors (GemmParams.A2)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

}  // namespace

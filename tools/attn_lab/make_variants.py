"""Round-3 incident lab, build side (runs in the build container): variants of the round-2 attention kernel whose
peeled ragged-key-tile copy returned wrong rows in 1-3 of 30 launches.  Each variant is the source of
commit 812d8e7 (the failing build) with ONE change, compiled into its own small shared library
gpurun_tmp/attn_lab/<name>/lib.so (ISA beside it: -save-temps) (attention + host plumbing only) which tools/attn_lab/run_lab.py loads on the GPU box.

  old        812d8e7 as it was (expected to fail)
  keepq      the Q fragments are kept live to the end of the kernel (their registers cannot be re-used in the peeled tile)
  schedbar   no VALU instruction may be scheduled between the score MFMAs of a 32-key block
  nops       2 x s_nop 15 after the score MFMAs of the peeled tile (any missing MFMA -> VALU wait state would be covered)
  noexact    the peeled tile skips the exact pre-pass (mask select only)
  pad100     -mllvm -amdgpu-mfma-padding-ratio=100 (s_nops between back-to-back MFMAs)
  nooob      K descriptor without a record limit (no out-of-range LDS-DMA lanes; the harness pads the buffer)
  r2ship     the round-2 shipped kernel (force_exact parameter; clean in round 2)
  unified    the round-3 kernel of this tree (one code path, -inf through the C operand)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "hi3d-official_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_tmp", "attn_lab")
HIPCC = "/opt/rocm/bin/hipcc"


def git_show(rev, path):
    return subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{path}"], check=True, stdout=subprocess.PIPE).stdout.decode()


def must_replace(s, a, b):
    assert a in s, a
    return s.replace(a, b)


def main():
    os.makedirs(OUT, exist_ok=True)
    old = git_show("812d8e7", "hi3d-official_amd/csrc/attention.hip")
    variants = {"old": (old, [])}
    variants["keepq"] = (must_replace(old, "  if (nfull < ntile) tile(nfull, std::true_type{});",
                                      "  if (nfull < ntile) tile(nfull, std::true_type{});\n"
                                      "  for (int qb = 0; qb < QB; ++qb) for (int ks = 0; ks < 4; ++ks) asm volatile(\"\" :: \"v\"(qf[qb][ks]));"), [])
    anchor = "      if (!PRE) {                                // scores -> log2 domain, relative to the reference point"
    variants["schedbar"] = (must_replace(old, anchor, "      __builtin_amdgcn_sched_barrier(0);\n" + anchor), [])
    variants["nops"] = (must_replace(old, anchor, "      if (ragged) asm volatile(\"s_nop 15\\ns_nop 15\" ::: \"memory\");\n" + anchor), [])
    variants["noexact"] = (must_replace(old, "bool exact = (j == 0) || ragged;", "bool exact = (j == 0);"), [])
    variants["pad100"] = (old, ["-mllvm", "-amdgpu-mfma-padding-ratio=100"])
    variants["nooob"] = (must_replace(old, "(int)min((long)0x7fffffff, ((long)p.Skv - 1) * p.ldk * 2 + 128)", "0x7fffffff"), [])
    variants["r2ship"] = (git_show("5c0b83c", "hi3d-official_amd/csrc/attention.hip"), [])
    variants["unified"] = (open(os.path.join(CSRC, "attention.hip")).read(), [])
    # ---- second round: everything on top of `nops` (fails in every launch, in ~2 % of the waves) ----
    nops = variants["nops"][0]
    variants["n_noexact"] = (must_replace(nops, "bool exact = (j == 0) || ragged;", "bool exact = (j == 0);"), [])
    # pre-pass runs, but its result is not applied (reference point, l and O untouched)
    variants["n_noapply"] = (must_replace(nops, "const float d = (j == 0) ? t : fmaxf(t, 0.f);",
                                          "const float d = (j == 0) ? t : (ragged ? 0.f * fminf(t, 0.f) : fmaxf(t, 0.f));"), [])
    # no O rescale in the peeled tile (alpha forced to 1 there; m / l still move)
    variants["n_noscaleO"] = (must_replace(nops, "for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;",
                                           "for (int r = 0; r < 16; ++r) o[qb][db][r] *= (ragged ? 1.0f : alpha);"), [])
    # the score MFMAs of the common pass cannot be merged with the pre-pass ones (the compiler reuses the pre-pass
    # results in the peeled tile: both passes see the same operands)
    variants["n_nocse"] = (must_replace(nops, "      // common pass\n", "      // common pass\n      if (ragged) asm volatile(\"\" : \"+v\"(qf[0][0]), \"+v\"(qf[1][0]));\n"), [])
    # no cross-lane exchange through the LDS crossbar: the other half-wave's maximum through v_permlane32_swap
    variants["n_permlane"] = (must_replace(nops, "const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));",
                                           "float t; { unsigned a_ = __float_as_uint(mx[qb]), b_ = a_; auto r_ = __builtin_amdgcn_permlane32_swap(a_, b_, false, false);"
                                           " t = fmaxf(mx[qb], __uint_as_float(hi ? r_[0] : r_[1])); }"), [])
    variants["n_O1"] = (nops, ["-O1"])
    # dump of the peeled tile's softmax state per (row, head, half-wave): m before, own max, tile max, alpha, row sum, l after
    dump = must_replace(nops, "          const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));",
                        "          const float t = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, 64));\n"
                        "          if (ragged) { dbg[qb][0] = m_run[qb]; dbg[qb][1] = mx[qb]; dbg[qb][2] = t; }")
    dump = must_replace(dump, "    bf16x8 pf[QB][4];\n    float psum[QB];", "    bf16x8 pf[QB][4];\n    float psum[QB];\n    (void)0;")
    dump = must_replace(dump, "  float m_run[QB], l_run[QB];", "  float m_run[QB], l_run[QB];\n  float dbg[QB][6] = {};")
    dump = must_replace(dump, "    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];",
                        "    for (int qb = 0; qb < QB; ++qb) { if (ragged) { dbg[qb][3] = l_run[qb]; dbg[qb][4] = psum[qb]; } l_run[qb] += psum[qb]; if (ragged) dbg[qb][5] = l_run[qb]; }")
    dump = must_replace(dump, "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;",
                        "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;\n"
                        "      { float* dp = (float*)(p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + p.H * 64) + (h * 2 + hi) * 6;\n"
                        "        for (int i = 0; i < 6; ++i) dp[i] = dbg[qb][i]; }")
    variants["n_dump"] = (dump, [])
    # second dump: l at the entry of the peeled tile, d, alpha, l after the pre-pass, m after, row sum
    d2 = must_replace(nops, "  float m_run[QB], l_run[QB];", "  float m_run[QB], l_run[QB];\n  float dbg[QB][6] = {};")
    d2 = must_replace(d2, "          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);",
                      "          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);\n"
                      "          if (ragged) { dbg[qb][0] = l_run[qb]; dbg[qb][1] = d; dbg[qb][2] = alpha; }")
    d2 = must_replace(d2, "          l_run[qb] *= alpha;", "          l_run[qb] *= alpha;\n          if (ragged) { dbg[qb][3] = l_run[qb]; dbg[qb][4] = m_run[qb]; }")
    d2 = must_replace(d2, "    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];",
                      "    for (int qb = 0; qb < QB; ++qb) { if (ragged) dbg[qb][5] = psum[qb]; l_run[qb] += psum[qb]; }")
    d2 = must_replace(d2, "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;",
                      "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;\n"
                      "      { float* dp = (float*)(p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + p.H * 64) + (h * 2 + hi) * 6;\n"
                      "        for (int i = 0; i < 6; ++i) dp[i] = dbg[qb][i]; }")
    variants["n2_dump"] = (d2, [])
    # where does the state go wrong: in the loop over the full tiles (with the LDS-DMA of the ragged tile in flight during
    # the last of them), or inside the peeled tile?
    variants["n_skiplast"] = (must_replace(nops, "  if (nfull < ntile) tile(nfull, std::true_type{});", "  /* peeled tile skipped */"), [])
    late = must_replace(nops, "    if (j + 1 < ntile) issue(j + 1, (j & 1) ^ 1);",
                        "    if (j + 1 < (ragged ? ntile : nfull)) issue(j + 1, (j & 1) ^ 1);")
    late = must_replace(late, "  if (nfull < ntile) tile(nfull, std::true_type{});",
                        "  if (nfull < ntile) { if (nfull > 0) { __syncthreads(); issue(nfull, nfull & 1); } tile(nfull, std::true_type{}); }")
    variants["n_lateissue"] = (late, [])
    # third dump: history of l over the last three FULL tiles + at the entry of the peeled one (same binary runs S = 576, too)
    d3 = must_replace(nops, "  float m_run[QB], l_run[QB];", "  float m_run[QB], l_run[QB];\n  float dbg[QB][6] = {};")
    d3 = must_replace(d3, "    for (int qb = 0; qb < QB; ++qb) l_run[qb] += psum[qb];",
                      "    for (int qb = 0; qb < QB; ++qb) {\n"
                      "      if (ragged) { dbg[qb][3] = l_run[qb]; dbg[qb][5] = psum[qb]; }\n"
                      "      l_run[qb] += psum[qb];\n"
                      "      if (!ragged) { if (j == nfull - 3) dbg[qb][0] = l_run[qb]; if (j == nfull - 2) dbg[qb][1] = l_run[qb];\n"
                      "                     if (j == nfull - 1) { dbg[qb][2] = l_run[qb]; dbg[qb][4] = psum[qb]; } }\n"
                      "    }")
    d3 = must_replace(d3, "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;",
                      "      unsigned short* op = p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + h * 64;\n"
                      "      { float* dp = (float*)(p.out + ((long)b * p.Sq + qrow[qb]) * p.ldo + p.H * 64) + (h * 2 + hi) * 6;\n"
                      "        for (int i = 0; i < 6; ++i) dp[i] = dbg[qb][i]; }")
    variants["n3_dump"] = (d3, [])
    only = sys.argv[1:]
    procs = []
    for name, (src, extra) in variants.items():
        if only and name not in only:
            continue
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "attention.hip"), "w") as fh:
            fh.write(src)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value",
               "-I", CSRC, "-save-temps=obj"] + extra + [os.path.join(d, "attention.hip"), os.path.join(CSRC, "host.hip"),
                                                           "-o", os.path.join(d, "lib.so")]
        procs.append((name, subprocess.Popen(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        out = p.communicate()[0].decode()
        print(name, "rc", p.returncode, out[-400:] if p.returncode else "")
        d = os.path.join(OUT, name)
        for f in os.listdir(d):                      # keep the source, the device ISA and the library
            if not (f in ("attention.hip", "lib.so") or f.endswith("gfx950.s") and f.startswith("attention")):
                os.remove(os.path.join(d, f))


if __name__ == "__main__":
    main()

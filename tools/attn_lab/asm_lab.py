"""Round-3 incident lab, ISA-patching stage (build side).  Source-level variants perturb register allocation and
scheduling of the whole kernel, which moves the failure around; here the DEVICE ASSEMBLY of one failing variant is
patched instead -- wait states / waits inserted after one class of instruction, in one region -- re-assembled
(clang -x assembler, ld.lld) into a code object and run through hipModuleLaunchKernel by run_asm_lab.py.  Everything else
in the binary stays bit-identical, so a patch that makes the failure disappear names the instruction pair.

usage: asm_lab.py [base-variant]          (default: nops)   ->  gpurun_tmp/attn_lab/asm/<patch>.hsaco
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAB = os.path.join(ROOT, "gpurun_tmp", "attn_lab")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN12_GLOBAL__N_115attn_d64_kernelILb0EEEvNS_10AttnParamsE"


def kernel_span(lines):
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    return a, b


def opcode(line):
    m = re.match(r"\s+([a-z_0-9]+)", line)
    return m.group(1) if m else None


def patch(lines, when, insert, before=False, region=None):
    """insert `insert` (list of asm lines) after (or before) every instruction of the <false> kernel for which when(op, line)."""
    a, b = kernel_span(lines)
    lo, hi = (a, b) if region is None else region(lines, a, b)
    out, n = [], 0
    for i, l in enumerate(lines):
        hit = lo <= i <= hi and opcode(l) is not None and when(opcode(l), l)
        if hit and before:
            out += insert
        out.append(l)
        if hit and not before:
            out += insert
        n += hit
    return out, n


def label_region(first, last):
    def f(lines, a, b):
        lo = next(i for i in range(a, b) if lines[i].startswith(first + ":"))
        hi = b if last is None else next(i for i in range(lo, b) if lines[i].startswith(last + ":"))
        return lo, hi
    return f


NOP16 = ["\ts_nop 15"]
PATCHES = {
    "ctl": (lambda op, l: False, [], False),
    # every MFMA followed by 32 wait states: any MFMA -> consumer / MFMA -> MFMA distance is covered
    "mfma_nop32": (lambda op, l: op.startswith("v_mfma"), NOP16 * 2, False),
    # every MFMA preceded by 16 wait states: VALU / LDS write -> MFMA source distances are covered
    "nop16_mfma": (lambda op, l: op.startswith("v_mfma"), NOP16, True),
    "lgkm0_mfma": (lambda op, l: op.startswith("v_mfma"), ["\ts_waitcnt lgkmcnt(0)"], True),
    # transcendental unit: wait states after every v_exp
    "exp_nop8": (lambda op, l: op.startswith("v_exp"), ["\ts_nop 7"], False),
    # packed fp32 / conversions
    "pk_nop4": (lambda op, l: op.startswith("v_pk_"), ["\ts_nop 3"], False),
    "cvt_nop4": (lambda op, l: op.startswith("v_cvt_pk"), ["\ts_nop 3"], False),
    # cross-lane exchange and LDS reads fully retired before anything else issues
    "bperm_wait": (lambda op, l: op.startswith("ds_bpermute"), ["\ts_waitcnt lgkmcnt(0)", "\ts_nop 7"], False),
    "dsread_wait": (lambda op, l: op.startswith("ds_read"), ["\ts_waitcnt lgkmcnt(0)"], False),
    # the LDS-DMA issue: wait states around the M0 writes and after each request
    "dma_nop": (lambda op, l: op.startswith("buffer_load") or "m0" in l, NOP16, False),
    # all VALU spaced out
    "valu_nop2": (lambda op, l: op.startswith("v_") and not op.startswith("v_mfma"), ["\ts_nop 1"], False),
    "barrier_nop": (lambda op, l: op == "s_barrier", NOP16 * 4, False),
}


REGIONS = {          # labels of the <false> kernel in the `old` build (.LBB2_*): first-tile pre-pass, common pass (+ redo), P V block
    "r_first": (".LBB2_24", ".LBB2_26"), "r_common": (".LBB2_26", ".LBB2_29"), "r_pv": (".LBB2_29", ".LBB2_31"),
}


def main():
    base = sys.argv[1] if len(sys.argv) > 1 else "nops"
    if len(sys.argv) > 2 and sys.argv[2] == "bisect":
        return bisect(base)
    if len(sys.argv) > 2 and sys.argv[2] == "pair":
        return pair(base)
    src = os.path.join(LAB, base, "attention-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(src).read().split("\n")
    out_dir = os.path.join(LAB, "asm")
    os.makedirs(out_dir, exist_ok=True)
    for name, (when, ins, before) in PATCHES.items():
        pl, n = patch(lines, when, ins, before)
        s_path = os.path.join(out_dir, f"{base}.{name}.s")
        with open(s_path, "w") as fh:
            fh.write("\n".join(pl))
        o_path = s_path[:-2] + ".o"
        subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o_path], check=True)
        subprocess.run([f"{LLVM}/ld.lld", "-shared", o_path, "-o", s_path[:-2] + ".hsaco"], check=True)
        os.remove(o_path)
        print(f"{base}.{name}: {n} insertion points")


def build(lines, name, base, out_dir):
    s_path = os.path.join(out_dir, f"{base}.{name}.s")
    with open(s_path, "w") as fh:
        fh.write("\n".join(lines))
    o_path = s_path[:-2] + ".o"
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o_path], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", o_path, "-o", s_path[:-2] + ".hsaco"], check=True)
    os.remove(o_path)


def pair(base):
    """Experiments on ONE MFMA pair of the `old` build's common pass (first two score MFMAs of a key tile: same A operand,
    query block 0 then 1): how long a gap between them is needed, is the stale value a late RESULT (extra wait before the
    first consumer cures it) or a wrong OPERAND, and does the victim follow the issue order."""
    src = os.path.join(LAB, base, "attention-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(src).read().split("\n")
    out_dir = os.path.join(LAB, "asm")
    os.makedirs(out_dir, exist_ok=True)
    a, b = kernel_span(lines)
    lo, hi = label_region(*REGIONS["r_common"])(lines, a, b)
    m = [i for i in range(lo, hi) if opcode(lines[i]) and opcode(lines[i]).startswith("v_mfma")]
    i0, i1 = m[0], m[1]                                   # qb0 ks0, qb1 ks0
    cons = next(i for i in range(m[7], hi) if lines[i].strip().startswith("s_nop 10"))
    for ws in (1, 2, 4, 8, 16):
        build(lines[:i0 + 1] + [f"\ts_nop {ws - 1}"] + lines[i0 + 1:], f"pair.gap{ws:02d}", base, out_dir)
    gap = NOP16 * 2
    build(lines[:i0 + 1] + gap + lines[i0 + 1:cons + 1] + NOP16 * 4 + lines[cons + 1:], "pair.gap32_lateconsumer", base, out_dir)
    build(lines[:i0 + 1] + gap + lines[i0 + 1:i1 + 1] + gap + lines[i1 + 1:], "pair.gap32_both", base, out_dir)
    sw = list(lines)
    sw[i0], sw[i1] = lines[i1], lines[i0]                 # query block 1 first
    build(sw[:i0 + 1] + gap + sw[i0 + 1:], "pair.gap32_swapped", base, out_dir)
    build(lines[:i1 + 1] + gap + lines[i1 + 1:], "pair.gap32_after_second", base, out_dir)
    # the gap filled with a wait for all LDS reads instead of idle wait states
    build(lines[:i0 + 1] + ["\ts_waitcnt lgkmcnt(0)"] + lines[i0 + 1:], "pair.lgkm0_between", base, out_dir)
    build(lines[:i0] + ["\ts_waitcnt lgkmcnt(0)"] + lines[i0:], "pair.lgkm0_before", base, out_dir)
    build(lines[:i0] + ["\ts_waitcnt lgkmcnt(0)"] + lines[i0:i0 + 1] + gap + lines[i0 + 1:], "pair.lgkm0_before_gap32", base, out_dir)
    build(lines[:i0 + 1] + gap + lines[i0 + 1:], "pair.gap32", base, out_dir)
    print("pair patches around lines", i0 + 1, i1 + 1, "consumer", cons + 1)


def bisect(base):
    """mfma_nop32 restricted to one region, and -- inside the first-tile pre-pass and the common pass -- to ONE MFMA at a time."""
    src = os.path.join(LAB, base, "attention-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(src).read().split("\n")
    out_dir = os.path.join(LAB, "asm")
    os.makedirs(out_dir, exist_ok=True)
    is_mfma = lambda op, l: op.startswith("v_mfma")
    for rname, (a_, b_) in REGIONS.items():
        pl, n = patch(lines, is_mfma, NOP16 * 2, False, region=label_region(a_, b_))
        build(pl, "mfma_nop32." + rname, base, out_dir)
        print(f"{base}.mfma_nop32.{rname}: {n} insertion points")
    for rname in ("r_first", "r_common"):
        a, b = kernel_span(lines)
        lo, hi = label_region(*REGIONS[rname])(lines, a, b)
        idx = [i for i in range(lo, hi) if opcode(lines[i]) and opcode(lines[i]).startswith("v_mfma")]
        for k, i in enumerate(idx):
            pl = lines[:i + 1] + NOP16 * 2 + lines[i + 1:]
            build(pl, f"one.{rname}.{k:02d}", base, out_dir)
            print(f"{base}.one.{rname}.{k:02d}: after line {i + 1}: {lines[i].strip()}")


if __name__ == "__main__":
    main()

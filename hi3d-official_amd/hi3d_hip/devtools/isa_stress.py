"""Timing-perturbation stress of a gfx950 kernel at the ISA level.

Background: the round-2 build of the flash-attention kernel returned wrong rows in a few launches out of a
thousand.  Every source-level experiment moved the failure around, because any source change re-schedules the whole kernel.
What finally characterised it was patching the DEVICE ASSEMBLY of the failing build: idle wait states inserted after one
class of instruction, everything else bit-identical.  A kernel without a latent timing dependence is indifferent to such
patches (wait states cannot change the result of a correct program); the round-2 build broke in EVERY launch as soon as
16 idle wait states separated two MFMAs that are adjacent in its instruction stream, with two waves on the SIMD.

This module turns that into a reusable check:  device assembly of a .hip file (hipcc -S), a set of perturbation patches
applied to ONE kernel symbol, re-assembly into code objects (clang -x assembler, ld.lld) and launch through the HIP module
API on the HIP runtime torch is linked to.  tests/test_kernels_gpu.py runs the shipped attention kernels under every patch
and requires bit-identical output.
"""
import ctypes as C
import os
import re
import struct
import subprocess
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")       # the compiler build.py uses (same override)
LLVM = os.environ.get("HI3D_LLVM_BIN", os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin"))
if not os.path.isdir(LLVM):
    LLVM = "/opt/rocm/lib/llvm/bin"


ISA_CACHE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_isa")


def _source_digest(hip_file):
    """sha256 over the translation unit and every header beside it (what the device code can depend on)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in [hip_file] + sorted(glob.glob(os.path.join(os.path.dirname(hip_file), "*.h"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def device_asm(hip_file, include_dirs=(), cache=True):
    """gfx950 device assembly of a .hip translation unit (list of lines), compiled with the flags of build.py.
    Cached under hi3d_hip/_isa/ keyed by a digest of the sources: `build.py` fills the cache in the build container
    (gemm.hip takes ~50 s of hipcc), so the GPU suite only patches and re-assembles."""
    cs = os.path.join(ISA_CACHE, os.path.basename(hip_file) + ".s")
    dig = _source_digest(hip_file) if cache else None
    if cache and os.path.exists(cs) and os.path.exists(cs + ".digest") and open(cs + ".digest").read().strip() == dig:
        return open(cs).read().split("\n")
    lines = _compile_device_asm(hip_file, include_dirs)
    if cache:
        try:
            os.makedirs(ISA_CACHE, exist_ok=True)
            with open(cs, "w") as fh:
                fh.write("\n".join(lines))
            with open(cs + ".digest", "w") as fh:
                fh.write(dig)
        except OSError:
            pass
    return lines


def _compile_device_asm(hip_file, include_dirs=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        # (the code-generation flags of build.py's FLAGS, so the stressed ISA is the shipped one)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
               "-Wno-c++20-extensions", "-S", "--cuda-device-only", "-o", out, hip_file] + [f"-I{i}" for i in include_dirs]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {hip_file}:\n" + r.stdout.decode(errors="replace")[-4000:])
        return open(out).read().split("\n")


def opcode(line):
    m = re.match(r"\s+([a-z_0-9]+)", line)
    return m.group(1) if m else None


def kernel_span(lines, symbol):
    a = next(i for i, l in enumerate(lines) if l.startswith(symbol + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    return a, b


def insert(lines, symbol, when, extra, before=False):
    """`extra` (asm lines) after -- or before -- every instruction of kernel `symbol` for which when(opcode, line)."""
    a, b = kernel_span(lines, symbol)
    out, n = [], 0
    for i, l in enumerate(lines):
        hit = a <= i <= b and opcode(l) is not None and when(opcode(l), l)
        if hit and before:
            out += extra
        out.append(l)
        if hit and not before:
            out += extra
        n += hit
    return out, n


NOP16 = ["\ts_nop 15"]
is_mfma = lambda op, l: op.startswith("v_mfma")          # noqa: E731
PATCHES = {
    # name: (predicate, inserted lines, before?)
    "mfma_then_32_idle": (is_mfma, NOP16 * 2, False),                       # what broke the round-2 build in every launch
    "16_idle_then_mfma": (is_mfma, NOP16, True),
    "lds_drained_before_mfma": (is_mfma, ["\ts_waitcnt lgkmcnt(0)"], True),
    "packed_fp32_then_4_idle": (lambda op, l: op.startswith("v_pk_"), ["\ts_nop 3"], False),
    "exp_then_8_idle": (lambda op, l: op.startswith("v_exp"), ["\ts_nop 7"], False),
    "every_valu_then_2_idle": (lambda op, l: op.startswith("v_") and not op.startswith("v_mfma"), ["\ts_nop 1"], False),
    "valu_noop_after_mfma": (is_mfma, ["\tv_nop"], False),                   # adjacent MFMAs split by one VALU instruction
    "barrier_then_64_idle": (lambda op, l: op == "s_barrier", NOP16 * 4, False),
    "sleep_after_mfma": (is_mfma, ["\ts_sleep 2"], False),
}


def assemble(lines, path):
    """asm lines -> loadable code object at `path` (.hsaco)."""
    s_path = path + ".s"
    with open(s_path, "w") as fh:
        fh.write("\n".join(lines))
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", path + ".o"], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", path + ".o", "-o", path], check=True)
    os.remove(path + ".o")
    os.remove(s_path)
    return path


class Module:
    """A code object loaded through hipModuleLoad on torch's HIP runtime; launch(kernel, grid, block, packed kernarg bytes)."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import torch
            h = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
            h.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
            h.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
            h.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]
            h.hipModuleUnload.argtypes = [C.c_void_p]
            cls._hip = h
        return cls._hip

    def __init__(self, path):
        self.mod = C.c_void_p()
        rc = self.hip().hipModuleLoad(C.byref(self.mod), path.encode())
        if rc:
            raise RuntimeError(f"hipModuleLoad({path}) -> {rc}")
        self.fns = {}

    def launch(self, symbol, grid, block, kernarg, stream):
        fn = self.fns.get(symbol)
        if fn is None:
            fn = C.c_void_p()
            rc = self.hip().hipModuleGetFunction(C.byref(fn), self.mod, symbol.encode())
            if rc:
                raise RuntimeError(f"hipModuleGetFunction({symbol}) -> {rc}")
            self.fns[symbol] = fn
        kernarg += b"\0" * (-len(kernarg) % 8)
        buf = C.create_string_buffer(kernarg, len(kernarg))
        size = C.c_size_t(len(kernarg))
        extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p).value, 2, C.cast(C.pointer(size), C.c_void_p).value, 3)
        rc = self.hip().hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, stream, None, extra)
        if rc:
            raise RuntimeError(f"hipModuleLaunchKernel({symbol}) -> {rc}")

    def close(self):
        self.hip().hipModuleUnload(self.mod)


# ---- the attention kernel of csrc/attention.hip ---------------------------------------------------------------------
def attn_symbol(pre, vrow):
    """attn_d64_kernel<PRE, VROW>: q pre-scaled / V row-major (transposing LDS reads) instead of the pre-transposed V^T."""
    return f"_ZN12_GLOBAL__N_115attn_d64_kernelILb{int(pre)}ELb{int(vrow)}EEEvNS_10AttnParamsE"


ATTN_SYMBOL = {True: attn_symbol(True, False), False: attn_symbol(False, False)}      # (the V^T form, by `pre`)


def attn_kernarg(q, k, vt, out, B, H, S_q, S_kv, ldq, ldk, ld_vt, ldo, scale):
    """struct AttnParams of csrc/attention.hip (4 pointers, 10 ints incl. force_exact = 0, scale * log2 e).
    VROW instantiations: vt = the row-major V pointer, ld_vt = its row pitch."""
    nqt = (S_q + 255) // 256
    return struct.pack("<4Q10if", q, k, vt, out, B, H, S_q, S_kv, ldq, ldk, ld_vt, ldo, nqt, 0, scale * 1.4426950408889634), nqt * H * B


# ---- the GEMM / implicit-GEMM convolution kernel of csrc/gemm.hip --------------------------------------------------------
def gemm_symbol(WM, NT, NS, AMODE, EPI, PP):
    return f"_ZN12_GLOBAL__N_116gemm_bf16_kernelILi{WM}ELi{NT}ELi{NS}ELi{AMODE}ELi{EPI}ELb{PP}EEEvNS_10GemmParamsE"


def gemm_launch_info(desc):
    """(symbol, grid, block, dynamic LDS bytes, kernarg bytes) of the launch hi3d_gemm_bf16(desc) would make
    (hi3d_debug_gemm_launch_info: the library's own dispatch heuristics decide, nothing is launched)."""
    from .. import lib as _l
    lib = _l.load()
    buf = C.create_string_buffer(512)
    info = (C.c_int32 * 10)()
    import torch
    _l.check(lib.hi3d_debug_gemm_launch_info_on(desc, torch.cuda.current_stream().cuda_stream, buf, info), "hi3d_debug_gemm_launch_info_on")
    nbytes, grid, block, smem, WM, NT, NS, AMODE, EPI, PP = list(info)
    return gemm_symbol(WM, NT, NS, AMODE, EPI, PP), grid, block, smem, buf.raw[:nbytes]


def launch_dyn_lds(mod, symbol, grid, block, smem, kernarg, stream):
    """Module.launch with dynamic LDS (raises the function's limit first, as the library does)."""
    h = mod.hip()
    if symbol not in mod.fns:
        fn = C.c_void_p()
        rc = h.hipModuleGetFunction(C.byref(fn), mod.mod, symbol.encode())
        if rc:
            raise RuntimeError(f"hipModuleGetFunction({symbol}) -> {rc}")
        h.hipFuncSetAttribute.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if smem > 65536:
            rc = h.hipFuncSetAttribute(fn, 8, smem)             # hipFuncAttributeMaxDynamicSharedMemorySize
            if rc:
                raise RuntimeError(f"hipFuncSetAttribute -> {rc}")
        mod.fns[symbol] = fn
    fn = mod.fns[symbol]
    kernarg += b"\0" * (-len(kernarg) % 8)
    buf = C.create_string_buffer(kernarg, len(kernarg))
    size = C.c_size_t(len(kernarg))
    extra = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p).value, 2, C.cast(C.pointer(size), C.c_void_p).value, 3)
    rc = h.hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, smem, stream, None, extra)
    if rc:
        raise RuntimeError(f"hipModuleLaunchKernel({symbol}) -> {rc}")

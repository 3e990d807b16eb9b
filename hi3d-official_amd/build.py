"""Build libhi3d_hip.so (all gfx950 kernels + the C ABI of include/hi3d_hip.h).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container; the resulting .so sits in-tree (git-ignored) and travels to the GPU box.
"""
import glob
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIB = os.path.join(ROOT, "hi3d_hip", "libhi3d_hip.so")
STAMP = LIB + ".stamp"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-Wno-unused-result", "-Wno-unused-value"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
            os.path.join(ROOT, "..", "include", "hi3d_hip.h")]:
        with open(f, "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile the library if sources changed. Returns the .so path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        if open(STAMP).read().strip() == dig:
            return LIB
    objs = []
    tmpdir = os.path.join(ROOT, "build")
    os.makedirs(tmpdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(tmpdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    with open(STAMP, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"[hi3d build] {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

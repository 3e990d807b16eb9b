"""Static check of the Python side of the boundary: every attribute that any file of the tree reads from a `hi3d_hip` module
(`ops.<name>`, `L.<name>`, `pack.<name>` ...), every `hi3d_*` symbol reached through the ctypes handle and every
`torch.ops.hi3d.<name>` must resolve, and every C export must have a Python caller.

Round 4 ended red because one commit deleted `ops.groupnorm_fold_linear` while `runtime_unet.py` and three GPU tests still called
it -- no CPU test noticed, `pytest -m gpu -x` stopped at the first of them and 150 GPU tests never ran.  This test is
the CPU tripwire for that class of mistake: it needs no GPU, only that the library loads."""
import ast
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hi3d-official_amd")

SCAN_DIRS = ["hi3d-official_amd", "tests", "tools", "oracle"]
SCAN_FILES = ["bench.py", "__graft_entry__.py"]

# C exports that deliberately have no Python caller on the product path (each with the reason)
EXPORT_NO_CALLER_OK = {
}


def _py_files():
    out = [os.path.join(ROOT, f) for f in SCAN_FILES]
    for d in SCAN_DIRS:
        for dp, dn, fn in os.walk(os.path.join(ROOT, d)):
            dn[:] = [x for x in dn if x not in ("__pycache__", "build", "_isa")]
            out += [os.path.join(dp, f) for f in fn if f.endswith(".py")]
    return sorted(out)


def _hi3d_submodules():
    d = os.path.join(PKG, "hi3d_hip")
    mods = {f[:-3] for f in os.listdir(d) if f.endswith(".py") and f != "__init__.py"}
    mods |= {f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f, "__init__.py"))}
    return mods


class _Scan(ast.NodeVisitor):
    """Collects (alias -> hi3d_hip submodule) bindings per scope-insensitive file (aliases are unique per file in this tree, and
    a name re-bound to something else would only produce a false alarm, never hide a miss) and every `<alias>.<attr>` load,
    `monkeypatch.setattr(<alias>, "<attr>", ...)`, `getattr(<alias>, "<attr>")`."""

    def __init__(self, path, submods):
        self.path, self.submods = path, submods
        self.in_pkg = os.path.dirname(path) == os.path.join(PKG, "hi3d_hip")
        self.alias = {}
        self.uses = []          # (module, attr, lineno)
        self.c_syms = []     # (.hi3d_xxx attribute, lineno)
        self.torch_ops = []     # (torch.ops.hi3d.<name>, lineno)

    def visit_ImportFrom(self, node):
        mod = node.module or ""
        if mod == "hi3d_hip" or (self.in_pkg and node.level == 1 and mod == ""):
            for a in node.names:
                if a.name in self.submods:
                    self.alias[a.asname or a.name] = a.name
        elif mod.startswith("hi3d_hip.") and mod.split(".")[1] in self.submods and len(mod.split(".")) == 2:
            for a in node.names:                       # from hi3d_hip.ops import gemm
                self.uses.append((mod.split(".")[1], a.name, node.lineno))
        elif self.in_pkg and node.level == 1 and mod in self.submods:
            for a in node.names:                       # from .ops import gemm
                self.uses.append((mod, a.name, node.lineno))
        self.generic_visit(node)

    def visit_Import(self, node):
        for a in node.names:
            p = a.name.split(".")
            if len(p) == 2 and p[0] == "hi3d_hip" and p[1] in self.submods and a.asname:
                self.alias[a.asname] = p[1]
        self.generic_visit(node)

    def visit_Attribute(self, node):
        if isinstance(node.value, ast.Name) and node.value.id in self.alias and isinstance(node.ctx, ast.Load):
            self.uses.append((self.alias[node.value.id], node.attr, node.lineno))
        if node.attr.startswith("hi3d_") and isinstance(node.ctx, ast.Load):
            self.c_syms.append((node.attr, node.lineno))
        # torch.ops.hi3d.<name>
        v = node.value
        if (isinstance(v, ast.Attribute) and v.attr == "hi3d" and isinstance(v.value, ast.Attribute) and v.value.attr == "ops"
                and isinstance(v.value.value, ast.Name) and v.value.value.id == "torch"):
            self.torch_ops.append((node.attr, node.lineno))
        self.generic_visit(node)

    def visit_Call(self, node):
        f = node.func
        name = f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else None)
        if name in ("setattr", "getattr", "hasattr", "delattr") and len(node.args) >= 2:
            tgt, key = node.args[0], node.args[1]
            if (isinstance(tgt, ast.Name) and tgt.id in self.alias and isinstance(key, ast.Constant) and isinstance(key.value, str)
                    and name != "hasattr"):
                self.uses.append((self.alias[tgt.id], key.value, node.lineno))
        self.generic_visit(node)


def _scan_all():
    submods = _hi3d_submodules()
    scans = []
    for p in _py_files():
        with open(p) as fh:
            src = fh.read()
        s = _Scan(p, submods)
        s.visit(ast.parse(src, p))
        scans.append(s)
    return scans


@pytest.fixture(scope="module")
def scans():
    return _scan_all()


def test_scanner_sees_the_known_call_sites(scans):
    """The scanner itself: it must find the call that round 4 lost, in the places round 4 lost it."""
    by = {os.path.relpath(s.path, ROOT): s for s in scans}
    rt = by["hi3d-official_amd/hi3d_hip/runtime_unet.py"]
    assert ("ops", "groupnorm_fold_linear") in {(m, a) for m, a, _ in rt.uses}
    tk = by["tests/test_kernels_gpu.py"]
    assert ("ops", "groupnorm_fold_linear") in {(m, a) for m, a, _ in tk.uses}
    tu = by["tests/test_unet_gpu.py"]
    assert ("ops", "GN_FOLD") in {(m, a) for m, a, _ in tu.uses}            # monkeypatch.setattr(ops, "GN_FOLD", ...)
    ops = by["hi3d-official_amd/hi3d_hip/ops.py"]
    assert "hi3d_groupnorm_fold_linear" in {a for a, _ in ops.c_syms}
    assert sum(len(s.uses) for s in scans) > 500


def test_every_hi3d_hip_attribute_used_anywhere_resolves(scans):
    missing = []
    mods = {}
    for s in scans:
        for mod, attr, line in s.uses:
            if mod not in mods:
                mods[mod] = importlib.import_module("hi3d_hip." + mod)
            ok = hasattr(mods[mod], attr)
            if not ok and os.path.isdir(os.path.join(PKG, "hi3d_hip", mod)):      # from hi3d_hip.devtools import isa_stress
                ok = os.path.exists(os.path.join(PKG, "hi3d_hip", mod, attr + ".py"))
            if not ok:
                missing.append(f"{os.path.relpath(s.path, ROOT)}:{line}: hi3d_hip.{mod}.{attr}")
    assert not missing, "names that do not exist in the module they are read from:\n  " + "\n  ".join(missing)


def test_every_hi3d_symbol_reached_from_python_is_exported_and_typed(scans):
    from hi3d_hip import lib as L
    l = L.load()
    bad = []
    for s in scans:
        for name, line in s.c_syms:
            fn = getattr(l, name, None)                          # (ctypes resolves the symbol here: AttributeError -> None)
            product = not os.path.relpath(s.path, ROOT).startswith("tools" + os.sep)
            if fn is None or (product and (name not in L.EXPORTS or fn.argtypes is None)):   # tools/ may poke un-exported debug hooks
                bad.append(f"{os.path.relpath(s.path, ROOT)}:{line}: {name}")
    assert not bad, "hi3d_* symbols used from Python that lib.EXPORTS / lib.load()'s signature table do not carry:\n  " + "\n  ".join(bad)


def test_every_export_has_a_python_caller(scans):
    from hi3d_hip import lib as L
    called = set()
    for s in scans:
        if os.path.basename(s.path) == "lib.py":
            continue
        called |= {a for a, _ in s.c_syms}
    orphans = [n for n in L.EXPORTS if n not in called and n not in EXPORT_NO_CALLER_OK]
    assert not orphans, f"C exports nothing in the tree calls (add a caller or an EXPORT_NO_CALLER_OK entry with the reason): {orphans}"
    stale = [n for n in EXPORT_NO_CALLER_OK if n in called or n not in L.EXPORTS]
    assert not stale, f"EXPORT_NO_CALLER_OK entries that are called after all / no longer exported: {stale}"


def test_every_torch_ops_hi3d_name_is_registered(scans):
    from hi3d_hip import torch_ops
    torch_ops.load()
    import torch
    bad = []
    for s in scans:
        for name, line in s.torch_ops:
            if not hasattr(torch.ops.hi3d, name):
                bad.append(f"{os.path.relpath(s.path, ROOT)}:{line}: torch.ops.hi3d.{name}")
    assert not bad, "\n  ".join(bad)
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.hi3d, name), name


def test_env_switches_read_by_the_runtime_are_documented():
    """Every HI3D_* environment switch the product path reads is named in DESIGN.md or INTEGRATION.md (an undocumented switch is
    how an opt-in experiment ends up in front of the goldens unnoticed)."""
    docs = open(os.path.join(ROOT, "DESIGN.md")).read() + open(os.path.join(ROOT, "INTEGRATION.md")).read()
    used = set()
    for dp, dn, fn in os.walk(PKG):
        dn[:] = [x for x in dn if x not in ("__pycache__", "build", "_isa")]
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                used |= set(re.findall(r'(?:getenv\(|environ\.get\(|environ\[)\s*"(HI3D_[A-Z0-9_]+)"',
                                       open(os.path.join(dp, f), errors="replace").read()))
    assert len(used) > 20, used
    missing = sorted(u for u in used if u not in docs)
    assert not missing, f"undocumented switches: {missing}"

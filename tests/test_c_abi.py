"""The C-ABI library loads without a GPU and exports every symbol include/pyascore_hip.h
and include/pyascore_aux.h declare; without a device the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADERS = [os.path.join(ROOT, "include", n) for n in ("pyascore_hip.h", "pyascore_aux.h", "pyascore_debug.h")]


@pytest.fixture(scope="module")
def lib():
    from pyascore_amd import build
    build.build()                      # hipcc cross-compiles gfx950 without a GPU
    from pyascore_amd import _lib
    return _lib.load()


def declared_symbols():
    text = "".join(open(h).read() for h in HEADERS)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pya_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 55
    from pyascore_amd import _lib
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (pya_[a-z_0-9]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, "declared but not exported: %s" % missing
    for n in names:
        assert getattr(lib, n) is not None


def test_the_environment_reaches_the_library_through_four_variables():
    """Route and debug switches are per-handle calls (include/pyascore_debug.h: pya_set_debug); the only getenv
    calls of the library are the two sizes and the two diagnostics host_tables.cpp:read_env names."""
    names = set()
    csrc = os.path.join(ROOT, "pyascore_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".cpp", ".hip", ".h", ".c")):
            text = open(os.path.join(csrc, f), errors="replace").read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            for m in re.finditer(r"getenv\(\s*(\w+|\"[A-Z_0-9]+\")", text):
                names.add((f, m.group(1)))
    assert {f for f, _ in names} <= {"host_tables.cpp"}, names
    text = open(os.path.join(csrc, "host_tables.cpp")).read()
    body = text[text.index("static void read_env"): text.index("void read_knobs")]
    env = set(re.findall(r'"(PYA_[A-Z_]+)"', body))
    assert env == {"PYA_WORKSPACE_MB", "PYA_CHUNK_MB", "PYA_HOST_TIMING", "PYA_STAMPS"}
    assert len(re.findall(r"getenv", text)) - len(re.findall(r"getenv", body)) == 0


def test_version_string(lib):
    assert b"gfx950" in lib.pya_version()


def test_library_names_the_tree_it_was_built_from(lib):
    """pya_version() carries the SHA-256 of every file under csrc/ and include/ (build.py:tree_digest)."""
    from pyascore_amd import build
    assert ("src=" + build.tree_digest()).encode() in lib.pya_version()


def test_every_object_was_compiled_from_the_files_in_the_tree(lib):
    """Each object's record (flags + SHA-256 over the compiler's own dependency list) matches the tree now,
    the dependency lists cover every header under csrc/ and include/, and the library is not older than
    its objects."""
    from pyascore_amd import build, _lib
    objs = sorted(f for f in os.listdir(build.CSRC) if f.endswith(".o"))
    srcs = sorted(f + ".o" for f in os.listdir(build.CSRC) if f.endswith((".hip", ".cpp", ".c")))
    if not os.path.exists(build.FAST):           # (no Python.h on this box: the CPython extension is optional and was not built)
        srcs = [s for s in srcs if s != build.FAST_SRC + ".o"]
        objs = [o for o in objs if o != build.FAST_SRC + ".o"]
    assert objs == srcs
    seen = set()
    for o in objs:
        path = os.path.join(build.CSRC, o)
        deps = build._deps_of(path)
        assert deps, o
        seen.update(deps)
        rec = open(path + ".flags").read().split("\n")
        assert rec[1] == build._digest(deps), "%s is stale" % o
        linked = build.FAST if o == build.FAST_SRC + ".o" else _lib.LIB_PATH      # (the CPython extension is its own library)
        assert os.path.getmtime(linked) >= os.path.getmtime(path), o
    unused = [p for p in build.source_files() if p not in seen]
    assert not unused, "in the tree digest but read by no compile: %s" % unused


def test_fails_loudly_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from pyascore_amd import PyAscore
    with pytest.raises(RuntimeError, match="no CPU path"):
        PyAscore(100.0, 10, "STY", 79.966331)


def test_product_does_not_import_the_oracle():
    """Nothing under pyascore_amd/ may import, load or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "pyascore_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "libascore_oracle" not in text and "libascore_ref" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle_abi.h" not in text, f


def test_one_hip_runtime_whichever_is_imported_first():
    """The library loaded BEFORE torch must not leave the process with two HIP runtimes (torch would
    then report "No HIP GPUs" and could not share device pointers with device.DevicePlan)."""
    import subprocess
    import sys
    code = ("import pyascore_amd._lib as L; L.load(); import torch; "
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "1"

"""Builds pyascore_amd/libpyascore_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m pyascore_amd.build [--force]

Device code (*.hip) is compiled by hipcc for gfx950 only; the host side (*.cpp) by g++ so the float32
score-table chain is evaluated by the same compiler family and libm as the reference (DESIGN.md
"Exactness").  -ffp-contract=off everywhere.

What makes an object stale is CONTENT, not a hand-kept header list and not time stamps: every compile
writes the compiler's own dependency list (-MD), and `<obj>.flags` keeps the flags and a SHA-256 over
the bytes of every file of that list inside this repository.  An object is rebuilt when that record
differs from what the tree holds now.  The library carries `tree_digest()` — a SHA-256 over every file
under csrc/ and include/ — in `pya_version()`, so a test can prove that the binary it ran is the
source it sees (tests/test_c_abi.py, tests/test_gpu_parity.py).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libpyascore_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

SOURCE_EXT = (".hip", ".cpp", ".h", ".c")
import sysconfig  # noqa: E402
# CPython extension behind PyAscore.score() (csrc/pyfast.c), named with the interpreter's ABI tag
# (_fast.cpython-310-x86_64-linux-gnu.so): another interpreter does not pick it up
FAST = os.path.join(HERE, "_fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
FAST_SRC = "pyfast.c"
VERSION_SRC = "version.cpp"          # compiled at every link with the tree digest as a macro

DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                "-fno-fast-math", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-fPIC", "-pthread", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
              "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")]


def source_files():
    """Every source file of the library: csrc/ and include/, sorted by repository-relative path."""
    out = []
    for d in (CSRC, INCLUDE):
        for name in os.listdir(d):
            if name.endswith(SOURCE_EXT):
                out.append(os.path.join(d, name))
    return sorted(out, key=lambda p: os.path.relpath(p, ROOT))


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        rel = os.path.relpath(p, ROOT).replace(os.sep, "/")
        with open(p, "rb") as f:
            body = f.read()
        h.update(rel.encode() + b"\0" + str(len(body)).encode() + b"\0" + body)
    return h.hexdigest()


def tree_digest():
    """SHA-256 over (path, bytes) of every file under csrc/ and include/ with a source extension."""
    return _digest(source_files())


def _deps_of(obj):
    """The files of this repository the compiler read for `obj` (from its -MD output), or None."""
    try:
        with open(obj + ".d") as f:
            text = f.read()
    except OSError:
        return None
    words = text.replace("\\\n", " ").split()
    deps = set()
    for w in words:
        if w.endswith(":"):
            continue
        p = os.path.realpath(w)
        if p.startswith(ROOT + os.sep):
            deps.add(p)
    return sorted(deps)


def _record(flags, deps):
    return " ".join(flags) + "\n" + _digest(deps) + "\n"


def _stale(obj, flags):
    if not os.path.exists(obj):
        return True
    deps = _deps_of(obj)
    if deps is None or any(not os.path.exists(d) for d in deps):
        return True
    try:
        with open(obj + ".flags") as f:
            return f.read() != _record(flags, deps)
    except OSError:
        return True


def _compile(cmd, flags, src, obj):
    for stale in (obj, obj + ".flags"):
        if os.path.exists(stale):
            os.remove(stale)
    _run(cmd + flags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])
    with open(obj + ".flags", "w") as f:
        f.write(_record(flags, _deps_of(obj)))


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False):
    dflags = list(DEVICE_FLAGS)
    if os.environ.get("PYA_DEFS"):              # -D switches for A/B builds
        dflags += os.environ["PYA_DEFS"].split()
    if os.environ.get("PYA_BUILD_STAMPS"):      # diagnostic build with in-kernel phase stamps
        dflags.append("-DPYA_STAMPS")
    jobs, objs = [], []
    for name in sorted(os.listdir(CSRC)):
        if name == VERSION_SRC or not name.endswith((".hip", ".cpp")):
            continue
        s, o = os.path.join(CSRC, name), os.path.join(CSRC, name + ".o")
        cmd, flags = ([HIPCC], dflags) if name.endswith(".hip") else (["g++"], HOST_FLAGS)
        if force or _stale(o, flags):
            jobs.append((cmd, flags, s, o))
        objs.append(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        for fut in [pool.submit(_compile, *j) for j in jobs]:
            fut.result()
    # the version object names the tree it was linked from; it is stale whenever anything changed
    digest = tree_digest()
    vflags = HOST_FLAGS + ['-DPYA_TREE_DIGEST="%s"' % digest]
    vs, vo = os.path.join(CSRC, VERSION_SRC), os.path.join(CSRC, VERSION_SRC + ".o")
    # relink when anything was compiled, the library is missing, or it is OLDER than one of its objects (a link that failed
    # after the compiles of an earlier call leaves fresh objects beside a stale library)
    relink = force or bool(jobs) or not os.path.exists(LIB) or \
        any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs if os.path.exists(o))
    if relink or _stale(vo, vflags):
        _compile(["g++"], vflags, vs, vo)
        relink = True
    if relink:
        try:
            _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + [vo])
        except BaseException:
            # the version object names a tree no library was linked from: the next call must link again
            for stale in (vo, vo + ".flags"):
                if os.path.exists(stale):
                    os.remove(stale)
            raise
    build_fast(force)
    return LIB


def build_fast(force=False):
    """The CPython extension PyAscore.score() calls pya_score_one through (no ctypes on the per-PSM path).  Built
    for the interpreter that runs this; pyascore_amd works without it (ascore.py falls back to ctypes)."""
    import sysconfig
    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        print("no Python.h under %s: pyascore_amd/_fast.so not built (score() will go through ctypes)" % inc, flush=True)
        return None
    flags = ["-O2", "-fPIC", "-Wall", "-I" + inc]
    src, obj = os.path.join(CSRC, FAST_SRC), os.path.join(CSRC, FAST_SRC + ".o")
    if force or _stale(obj, flags) or not os.path.exists(FAST):
        _compile(["gcc"], flags, src, obj)
        _run(["gcc", "-shared", "-o", FAST, obj])
        untagged = os.path.join(HERE, "_fast.so")           # (builds before r05 left the untagged name)
        if untagged != FAST and os.path.exists(untagged):
            os.remove(untagged)
    return FAST


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Builds pyascore_amd/libpyascore_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m pyascore_amd.build [--force]

Device code (*.hip) is compiled by hipcc for gfx950 only; the host side (host.cpp,
score_table.cpp) by g++ so the float32 score-table chain is evaluated by the same compiler
family and libm as the reference (DESIGN.md "Exactness").  -ffp-contract=off everywhere.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpyascore_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

DEVICE_SRC = ["bin_spectra.hip", "score_signatures.hip", "score_big.hip", "rank_and_localize.hip", "score_localize.hip", "tiny_batch.hip"]
HOST_SRC = ["host.cpp", "score_table.cpp", "aux_api.cpp"]
HEADERS = ["common.h", "device_common.hip.h", "bin_core.hip.h", "walk_core.hip.h", "score_core.hip.h",
           "localize_core.hip.h", "localize_body.hip.h", "fused_core.hip.h", "fused_pack.hip.h", "binom_chain.h", os.path.join("..", "..", "include", "pyascore_hip.h"),
           os.path.join("..", "..", "include", "pyascore_aux.h")]

DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                "-fno-fast-math", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-fPIC", "-pthread", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
              "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")]


def _stale(target, deps, flags=None):
    """Older than a dependency, or built with other flags (kept in <target>.flags)."""
    if not os.path.exists(target):
        return True
    if flags is not None:
        try:
            with open(target + ".flags") as f:
                if f.read() != " ".join(flags):
                    return True
        except OSError:
            return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(cmd, flags, src, obj):
    _run(cmd + flags + ["-c", src, "-o", obj])
    with open(obj + ".flags", "w") as f:
        f.write(" ".join(flags))


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False):
    dflags = list(DEVICE_FLAGS)
    if os.environ.get("PYA_LOC_WAVES"):         # A/B experiments on the localize kernel's occupancy target
        dflags.append("-DLOC_WAVES=" + os.environ["PYA_LOC_WAVES"])
    if os.environ.get("PYA_DEFS"):              # other -D switches for A/B builds
        dflags += os.environ["PYA_DEFS"].split()
    if os.environ.get("PYA_BUILD_STAMPS"):      # diagnostic build with in-kernel phase stamps
        dflags.append("-DPYA_STAMPS")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in DEVICE_SRC:
        s, o = os.path.join(CSRC, src), os.path.join(CSRC, src + ".o")
        if force or _stale(o, [s] + hdrs, dflags):
            _compile([HIPCC], dflags, s, o)
        objs.append(o)
    for src in HOST_SRC:
        s, o = os.path.join(CSRC, src), os.path.join(CSRC, src + ".o")
        if force or _stale(o, [s] + hdrs, HOST_FLAGS):
            _compile(["g++"], HOST_FLAGS, s, o)
        objs.append(o)
    if force or _stale(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

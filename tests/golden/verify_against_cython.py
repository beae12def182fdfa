"""Proves that the committed golden vectors are outputs of the reference ITSELF -- its real Cython extension, not the
shim tests/golden/make_golden.py drives (oracle/_ref = the reference's C++ core behind oracle/ref_shim.cpp).

    python tests/golden/verify_against_cython.py [case ...]

BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box, and nothing of the reference is kept in this
repository: the copy and the build live under /tmp).  Steps: copy the reference tree to /tmp, `python3 setup.py
build_ext --inplace` there (its own setup.py: Cython over pyascore/ptm_scoring/*.pyx + cpp/*.cpp), load the built
`pyascore.ptm_scoring` extension module directly (the package's __init__ imports parsers that need pyteomics, which this
image lacks; the extension needs nothing), replay every tests/golden/*.npz through `ptm_scoring.PyAscore` with the
harness every parity test uses (oracle/harness.collect: score() + every property, PSM by PSM) and require BIT equality
of every field -- counts, scores, sort order, sequences, Ascores, alternative sites.

tests/test_golden_pin.py runs this where /root/reference exists.
"""
import glob
import hashlib
import importlib.util
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_digest():
    h = hashlib.sha256()
    base = os.path.join(REFERENCE, "pyascore", "ptm_scoring")
    for path in sorted(glob.glob(os.path.join(base, "**", "*"), recursive=True)) + [os.path.join(REFERENCE, "setup.py")]:
        if os.path.isfile(path):
            with open(path, "rb") as f:
                h.update(os.path.relpath(path, REFERENCE).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def build_extension():
    """The reference's extension built by the reference's own setup.py in a copy under /tmp (reused while the
    reference's sources are unchanged).  Returns the path of the built ptm_scoring*.so."""
    work = os.path.join("/tmp", "pya_refbuild_" + reference_digest())
    found = glob.glob(os.path.join(work, "pyascore", "ptm_scoring*.so"))
    if found:
        return found[0]
    if os.path.exists(work):
        shutil.rmtree(work)
    shutil.copytree(REFERENCE, work, ignore=shutil.ignore_patterns(".git", "docs", "test"))
    log = subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=work, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    found = glob.glob(os.path.join(work, "pyascore", "ptm_scoring*.so"))
    if log.returncode or not found:
        raise RuntimeError("the reference's setup.py build_ext failed:\n" + log.stdout[-3000:])
    return found[0]


def load_extension(path):
    spec = importlib.util.spec_from_file_location("pyascore.ptm_scoring", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main(cases):
    from oracle import harness
    from pyascore_amd import synth
    ext = load_extension(build_extension())
    names = sorted(f[:-4] for f in os.listdir(HERE) if f.endswith(".npz"))
    if cases:
        names = [n for n in names if n in cases]
    bad = 0
    n_psm = n_rec = 0
    for name in names:
        settings, batch, expected = harness.load_case(os.path.join(HERE, name + ".npz"))
        scorer = harness.make_scorer(ext.PyAscore, settings)
        got = harness.collect(scorer, batch, synth.unpack_psm)
        diff = harness.compare(got, expected, exact_float=True)
        n_psm += batch["n_psm"]
        n_rec += int(expected["ps_bits"].size)
        print("%-16s %5d PSMs %7d pep_scores  %s" % (name, batch["n_psm"], expected["ps_bits"].size,
                                                      "bit-equal" if not diff else "DIFFERS: " + "; ".join(diff)))
        bad += bool(diff)
    print("%d golden files, %d PSMs, %d pep_scores records replayed through the reference's Cython extension: %s"
          % (len(names), n_psm, n_rec, "ALL BIT-EQUAL" if not bad else "%d FILES DIFFER" % bad))
    return 1 if bad or not names else 0


if __name__ == "__main__":
    if not os.path.isdir(REFERENCE):
        sys.exit("verify_against_cython: %s is not here (build container only)" % REFERENCE)
    sys.exit(main(sys.argv[1:]))

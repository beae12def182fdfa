"""Generates the committed golden vectors under tests/golden/ -- run HERE only (needs
/root/reference to build oracle/_ref and to read the reference's own test fixtures).

    python tests/golden/make_golden.py

Inputs:  * the reference's fixture data test/match_spectra_pairs/*.pkl (31 real Velos PSMs),
           converted to plain CSR arrays (no pyteomics types);
         * seeded synthetic batches of the BASELINE configs (pyascore_amd.synth);
         * hand-made edge cases.
Outputs: every result the reference produces for them, obtained by running the reference's own
         C++ core (oracle/_ref/libascore_ref.so, built by oracle/Makefile from the sources
         where they lie) through oracle/harness.collect.
"""
import os
import pickle
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import harness, orc  # noqa: E402
from pyascore_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = "/root/reference/test/match_spectra_pairs"
PHOSPHO = 79.966331


class _unitfloat(float):
    def __new__(cls, value=0.0, unit_info=None):
        return float.__new__(cls, value)


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod.startswith("pyteomics"):
            return _unitfloat
        return super().find_class(mod, name)


def _load(name):
    with open(os.path.join(FIX, name), "rb") as src:
        return _Unpickler(src).load()


def settings(mod_group="STY", mod_mass=PHOSPHO, mz_error=0.5, fragment_types="by", nl=()):
    return dict(bin_size=100.0, n_top=10, mod_group=mod_group, mod_mass=mod_mass,
                mz_error=mz_error, fragment_types=fragment_types,
                neutral_losses=[list(x) for x in nl])


def velos_psms(charge_mode):
    psms = []
    for tag in ("1_mods", "2_mods", "3_mods", "aux"):
        matches = _load("velos_matches_%s.pkl" % tag)
        spectra = _load("velos_spectra_%s.pkl" % tag)
        for m, s in zip(matches, spectra):
            var = np.isclose(m["mod_masses"], PHOSPHO)
            z = 1 if charge_mode == "z1" else max(1, int(m["charge_state"]) - 1)
            psms.append(dict(mz=np.asarray(s["mz_values"], np.float64),
                             intensity=np.asarray(s["intensity_values"], np.float64),
                             peptide=m["peptide"], n_of_mod=int(var.sum()), max_charge=z,
                             aux_pos=np.asarray(m["mod_positions"])[~var].astype(np.uint32),
                             aux_mass=np.asarray(m["mod_masses"])[~var].astype(np.float32)))
    return psms


def toy_spectrum(rng, peptide, mods=(), n_noise=120, err=0.05, lo=100.0, hi=1800.0):
    """b/y z1 peaks of `peptide` with PHOSPHO on 0-based residues `mods` + uniform noise."""
    mass = np.array([synth.RESIDUE_MASS[c] for c in peptide])
    for i in mods:
        mass[i] += PHOSPHO
    b = np.cumsum(mass)[:-1] + synth.PROTON
    y = np.cumsum(mass[::-1])[:-1] + synth.WATER + synth.PROTON
    sig = np.concatenate([b, y])
    sig = sig[rng.random(sig.size) < 0.7] + rng.uniform(-0.4 * err, 0.4 * err, size=None)
    mz = np.concatenate([sig, rng.uniform(lo, hi, n_noise)])
    it = np.concatenate([rng.lognormal(6, 1.2, sig.size), rng.lognormal(4.5, 1, n_noise)])
    o = np.argsort(mz)
    return mz[o], it[o]


def edge_psms(rng, err):
    P = []

    def add(pep, k, mods=(), z=1, aux_pos=(), aux_mass=(), **kw):
        mz, it = toy_spectrum(rng, pep, mods, err=err, **kw)
        P.append(dict(mz=mz, intensity=it, peptide=pep, n_of_mod=k, max_charge=z,
                      aux_pos=np.asarray(aux_pos, np.uint32), aux_mass=np.asarray(aux_mass, np.float32)))

    add("AGLSPEDLKR", 1, (3,))                       # k == n : unambiguous
    add("AGLSPEDLKR", 2, (3,))                       # k > n  : no pep_scores, best_score -1
    add("ASTLGYKR", 0)                               # k == 0
    add("ASTLGYKR", 3, (1, 2, 5))                    # k == n == 3
    add("MSTYLKAGSR", 1, (2,), aux_pos=(0, 1), aux_mass=(42.010565, 15.9949))   # n-term + M1 aux
    add("ACSTYLKAGSR", 2, (3, 9), z=2, aux_pos=(2,), aux_mass=(57.021464,))
    add("SSSSSSAK", 3, (0, 2, 4))                    # dense sites, many near-ties
    add("KSTAYSGLSTR", 2, (1, 8), z=3)
    add("ASTLGYKRSTY", 2, (1, 2), lo=5000.0, hi=5300.0)   # nothing can match: all scores tie at 0
    add("TYASGLK", 1, (0,), n_noise=3)               # tiny spectrum
    add("PEPTIDESEQWENCEK", 2, (3, 7))               # mixed
    add("ST", 1, (0,))                               # shortest useful peptide
    add("LLLLSLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLLTLLLLK", 1, (4,))   # long, 2 sites
    return P


ONLY = sys.argv[1:]          # optional: regenerate only the cases whose names start with these


def run_case(name, st, psms):
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    batch = synth.pack_batch(psms) if isinstance(psms, list) else psms
    ref = harness.make_scorer(orc.OracleAscore, st, kind="ref")
    exp = harness.collect(ref, batch, synth.unpack_psm)
    path = os.path.join(HERE, name + ".npz")
    harness.save_case(path, st, batch, exp)
    print("%-16s %5d PSMs  %7d pep_scores  %8.1f KB" % (name, batch["n_psm"], exp["ps_bits"].size,
                                                       os.path.getsize(path) / 1024))


def synth_case(cfg, n, seed):
    batch, s = synth.make_batch(cfg, n_psm=n, seed=seed)
    return s, batch


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    rng = np.random.default_rng(20261003)
    run_case("velos_z1", settings(), velos_psms("z1"))
    run_case("velos_zprec", settings(), velos_psms("zprec"))
    run_case("velos_nl", settings(nl=[("ST", 18.01528)]), velos_psms("zprec"))
    for cfg, n, seed in (("cfg1", 16, 11), ("cfg2", 64, 12), ("cfg3", 160, 13), ("cfg4", 64, 14),
                         ("cfg5", 16, 15)):
        s, b = synth_case(cfg, n, seed)
        run_case("synth_" + cfg, s, b)
    run_case("edge_default", settings(mz_error=0.05), edge_psms(rng, 0.05))
    run_case("edge_err05", settings(mz_error=0.5), edge_psms(rng, 0.5))
    run_case("edge_yb", settings(mz_error=0.05, fragment_types="yb"), edge_psms(rng, 0.05))
    run_case("edge_nKc", settings(mod_group="nKc", mod_mass=42.010565, mz_error=0.05),
             edge_psms(rng, 0.05))
    run_case("edge_nl", settings(mz_error=0.05, nl=[("ST", 18.01528), ("sty", 97.9769)]),
             edge_psms(rng, 0.05))
    run_case("edge_highres", settings(mz_error=0.02, fragment_types="bycz",
                                      nl=[("sty", 97.9769)]), edge_psms(rng, 0.02))
    run_case("edge_Zc", settings(mz_error=0.1, fragment_types="Zc", nl=[("m", 63.998)]),
             edge_psms(rng, 0.1))
    # equal intensities inside the windows: the reference keeps what std::nth_element + std::sort
    # leave (Spectra.cpp:24-41).  First half: three intensity levels, second half: count-like.
    s, b = synth_case("cfg2", 48, 16)
    it = b["intensity"]
    half = int(b["peak_off"][24])
    q = np.concatenate([np.floor(it[:half] / np.median(it) * 3.0), np.floor(it[half:] / np.median(it) * 40.0)]) + 1.0
    run_case("ties_cfg2", s, dict(b, intensity=q))


if __name__ == "__main__":
    main()

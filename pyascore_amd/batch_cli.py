"""Batched counterpart of pyAscore's CLI scoring loop.

The reference's `pyascore/__main__.py:127-172` walks the identifications one PSM at a time:
split each PSM's modifications into the unlocalised (variable) ones and the fixed ones
(`process_mods`, :83-103), pick the fragment-charge limit from the PSM charge (:138-145), call
``PyAscore.score`` and read four properties, then write a TSV.  With scoring on the GPU that
Python loop is the end-to-end bottleneck, so here the same decisions are taken per PSM on the
host, all PSMs are packed into ONE batch, scored with one ``PyAscore.score_batch`` call, and the
rows are produced from the batch results.  Inputs are the already-parsed records the reference's
parsers produce (dicts); file parsing itself (pyteomics) stays out of scope.

Output rows/columns follow docs/source/cli.rst:135-180: Scan, LocalizedSequence, PepScore,
Ascores (';' separated), AltSites (',' within ';').
"""
from itertools import groupby

import numpy as np

from .ascore import PyAscore
from .synth import pack_batch

COLUMNS = ("Scan", "LocalizedSequence", "PepScore", "Ascores", "AltSites")


def process_mods(residues, mod_mass, sequence, positions, masses, mod_correction_tol=1.0,
                 zero_based=False):
    """Variable/fixed split of one PSM's modifications (`__main__.py:83-103`).

    A modification counts as one of the unlocalised ones when its mass matches ``mod_mass``
    (numpy.isclose, rtol 1e-6, atol ``mod_correction_tol``) AND it sits on a residue of
    ``residues`` ('n' for position 0).  Everything else is returned as a fixed modification at its
    1-based position (0 = n-terminus).  Returns (uint32 positions, float32 masses, n_variable)."""
    shift = 1 if zero_based else 0
    n_variable = 0
    const_pos, const_masses = [], []
    for pos, mass in zip(positions, masses):
        pos = int(pos)
        aa = "n" if pos + shift == 0 else sequence[pos - 1 + shift]
        if np.isclose(mod_mass, mass, rtol=1e-6, atol=mod_correction_tol) and aa in residues:
            n_variable += 1
        else:
            const_pos.append(pos + shift)
            const_masses.append(mass)
    return np.array(const_pos, dtype=np.uint32), np.array(const_masses, dtype=np.float32), n_variable


def psm_charge(match, spectrum):
    """Charge used to bound the fragment charge (`__main__.py:138-145`): the identification's
    charge, else the spectrum's precursor charge, else 2; never below 2."""
    if match.get("charge_state") is not None and match["charge_state"] != 0:
        z = match["charge_state"]
    elif spectrum.get("precursor_charge") is not None and spectrum["precursor_charge"] != 0:
        z = spectrum["precursor_charge"]
    else:
        z = 2
    return max(int(z), 2)


def save_match(spectra, match):
    """``--match_save`` (`__main__.py:106-111`): the spectrum and the identification of a PSM as the two
    pickle files the reference writes, in the working directory."""
    import pickle
    with open("dump_spectra.pkl", "wb") as dst:
        pickle.dump([spectra], dst)
    with open("dump_match.pkl", "wb") as dst:
        pickle.dump([match], dst)


def select_psms(psms, spectra_map, residues, mod_mass, hit_depth=1, max_fragment_charge=5,
                mod_correction_tol=1.0, zero_based=False, match_save=False):
    """The reference's loop header (`__main__.py:129-147`): group by scan (input sorted by scan),
    take the first ``hit_depth`` hits of a scan (negative = all), drop PSMs without an
    unlocalised modification.  Returns (list of PSM dicts for pack_batch, list of scans).
    ``match_save``: the reference dumps every PSM it is about to score over the previous one
    (`__main__.py:148-149`), so what it leaves behind is the last one: that is what is written here."""
    picked, scans = [], []
    last = None
    for _, group in groupby(psms, lambda m: m["scan"]):
        for ind, match in enumerate(group):
            if ind == hit_depth:
                break
            spectrum = spectra_map[match["scan"]]
            const_pos, const_masses, n_variable = process_mods(
                residues, mod_mass, match["peptide"], match["mod_positions"], match["mod_masses"],
                mod_correction_tol, zero_based)
            if n_variable <= 0:
                continue
            picked.append(dict(mz=spectrum["mz_values"], intensity=spectrum["intensity_values"],
                               peptide=match["peptide"], n_of_mod=n_variable,
                               max_charge=min(max_fragment_charge, psm_charge(match, spectrum) - 1),
                               aux_pos=const_pos, aux_mass=const_masses))
            scans.append(match["scan"])
            last = (spectrum, match)
    if match_save and last is not None:
        save_match(*last)
    return picked, scans


def localize(ascore, psms, spectra_map, residues, mod_mass, hit_depth=1, max_fragment_charge=5,
             mod_correction_tol=1.0, zero_based=False, match_save=False, log=None):
    """Scores every selected PSM in one batched call and returns the TSV rows
    ``[scan, localized_sequence, pep_score, "a;b", "1,2;3"]`` in input order.  PSMs the library sets
    aside (invalid, or beyond one of its documented limits) keep their row -- empty localisation, PepScore
    nan -- and are reported through ``log`` (a callable taking one string) with their count, scans and codes."""
    if not isinstance(ascore, PyAscore):
        raise TypeError("ascore must be a pyascore_amd.PyAscore")
    picked, scans = select_psms(psms, spectra_map, residues, mod_mass, hit_depth, max_fragment_charge,
                                mod_correction_tol, zero_based, match_save)
    if not picked:
        return []
    batch = pack_batch(picked)
    # One PSM the kernels cannot take (longer than 64 residues, more than 15 000 site assignments,
    # an unknown residue, ...) must not cost the whole run its output: such PSMs are set aside by the
    # library, reported here, and written as rows without a localisation.
    res = ascore.score_batch(batch, skip_invalid=True)
    bad = np.flatnonzero(res["status"])
    if bad.size:
        import warnings
        warnings.warn("%d of %d PSMs were not scored (first: %s); their rows carry no localisation"
                      % (bad.size, len(picked), res["status_message"]), RuntimeWarning)
        if log is not None:
            shown = ", ".join("%s (code %d)" % (scans[i], int(res["status"][i])) for i in bad[:50])
            log("%d of %d PSMs set aside (rows written with an empty LocalizedSequence and PepScore nan); first: %s"
                % (bad.size, len(picked), res["status_message"]))
            log("set-aside scans: %s%s" % (shown, " ..." if bad.size > 50 else ""))
    ok = (res["status"] == 0) & (res["n_sig"] > 0)
    seqs = ascore.format_batch(batch, res["best_sig"], valid=ok.astype(np.int32))   # every string in one call
    rows = []
    for i, psm in enumerate(picked):
        if res["status"][i]:
            rows.append([scans[i], "", float("nan"), "", ""])
            continue
        k = psm["n_of_mod"]
        ascores = ";".join(str(s) for s in res["ascores"][i, :k])
        alts = ";".join(",".join(str(q) for q in ascore.alt_positions(m, psm["peptide"].encode("utf8")))
                        for m in res["alt_mask"][i, :k])
        rows.append([scans[i], seqs[i], float(res["best_score"][i]), ascores, alts])
    return rows


def write_tsv(rows, path):
    """Same file pandas' ``DataFrame(rows, columns=COLUMNS).to_csv(path, sep="\\t", index=False)``
    writes in the reference (`__main__.py:166-172`)."""
    with open(path, "w") as out:
        out.write("\t".join(COLUMNS) + "\n")
        for scan, seq, pep_score, ascores, alts in rows:
            out.write("%s\t%s\t%s\t%s\t%s\n" % (scan, seq, repr(float(pep_score)), ascores, alts))

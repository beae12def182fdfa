"""Host logic of the batched CLI driver: the variable/fixed modification split,
the charge heuristic and the PSM selection, against the behaviour of the reference's loop
(pyascore/__main__.py:83-103, :129-147), and -- on the GPU -- the produced rows against a
per-PSM loop over the CPU checker written the way the reference's loop is."""
import numpy as np
import pytest

from conftest import checker_kind

from pyascore_amd import batch_cli

PHOSPHO = 79.966331


def test_process_mods_split():
    pos, mass, nvar = batch_cli.process_mods("STY", PHOSPHO, "ASMTK", [0, 2, 3, 4],
                                             [42.010565, 79.9663, 15.994915, 79.966331])
    assert nvar == 2 and pos.dtype == np.uint32 and mass.dtype == np.float32
    assert pos.tolist() == [0, 3] and np.allclose(mass, [42.010565, 15.994915])
    # a phospho-mass modification on a residue outside the group is a fixed modification
    pos, mass, nvar = batch_cli.process_mods("ST", PHOSPHO, "AYK", [2], [79.966331])
    assert nvar == 0 and pos.tolist() == [2]
    # tolerance: |80.9 - 79.966| < 1.0 counts, 81.0 does not
    assert batch_cli.process_mods("S", PHOSPHO, "ASK", [2], [80.9])[2] == 1
    assert batch_cli.process_mods("S", PHOSPHO, "ASK", [2], [81.0])[2] == 0
    assert batch_cli.process_mods("S", PHOSPHO, "ASK", [2], [80.9], mod_correction_tol=0.5)[2] == 0
    # zero-based positions shift by one; -1 becomes the n-terminus
    pos, mass, nvar = batch_cli.process_mods("nS", 42.010565, "ASK", [-1, 1], [42.010565, 42.010565], zero_based=True)
    assert nvar == 2 and pos.size == 0
    pos, _, nvar = batch_cli.process_mods("S", PHOSPHO, "ASK", [-1, 1], [42.010565, 79.966331], zero_based=True)
    assert nvar == 1 and pos.tolist() == [0]


def test_charge_heuristic():
    assert batch_cli.psm_charge({"charge_state": 3}, {}) == 3
    assert batch_cli.psm_charge({"charge_state": 1}, {}) == 2
    assert batch_cli.psm_charge({"charge_state": 0}, {"precursor_charge": 4}) == 4
    assert batch_cli.psm_charge({"charge_state": None}, {"precursor_charge": None}) == 2
    assert batch_cli.psm_charge({}, {"precursor_charge": 0}) == 2


def _toy_inputs():
    rng = np.random.default_rng(5)
    spectra, psms = {}, []
    peptides = ["ASTLGYKR", "KSTAYSGLSTR", "PEPTIDEK", "MSTYLKAGSR", "AGLSPEDLKR", "ACSTYLKAGSR"]
    for scan, pep in enumerate(peptides, start=100):
        mz = np.sort(rng.uniform(120.0, 1500.0, 180))
        spectra[scan] = dict(mz_values=mz, intensity_values=rng.lognormal(5, 1, 180), precursor_charge=3)
        sites = [i + 1 for i, c in enumerate(pep) if c in "STY"]
        for hit in range(2):
            mods_pos = sites[hit:hit + 1] if sites else []
            mods_mass = [PHOSPHO] * len(mods_pos)
            if "C" in pep:
                mods_pos.append(pep.index("C") + 1)
                mods_mass.append(57.021464)
            if "M" in pep and hit == 0:
                mods_pos.append(0)
                mods_mass.append(42.010565)
            psms.append(dict(scan=scan, charge_state=[2, 3, 0][scan % 3], peptide=pep,
                             mod_positions=np.array(mods_pos, np.int32), mod_masses=np.array(mods_mass)))
    return spectra, psms


def test_selection_follows_the_reference_loop():
    spectra, psms = _toy_inputs()
    picked, scans = batch_cli.select_psms(psms, spectra, "STY", PHOSPHO, hit_depth=1, max_fragment_charge=5)
    assert scans == [100, 101, 102, 103, 104, 105]          # first hit of every scan with a variable mod
    assert [p["n_of_mod"] for p in picked] == [1] * 6
    assert [p["max_charge"] for p in picked] == [2, 2, 1, 2, 2, 1]   # min(5, max(z, 2) - 1); z = 3, 0 (-> precursor 3), 2, ...
    assert picked[3]["aux_pos"].tolist() == [0] and picked[5]["aux_pos"].tolist() == [2]
    picked, scans = batch_cli.select_psms(psms, spectra, "STY", PHOSPHO, hit_depth=-1)
    assert len(scans) == 10                                   # negative hit_depth: every hit that carries the modification
    picked, scans = batch_cli.select_psms(psms, spectra, "K", 14.01565)
    assert scans == []                                        # no PSM carries the modification of interest


def test_match_save_leaves_the_last_scored_psm(tmp_path, monkeypatch):
    """--match_save (`__main__.py:106-111, 148-149`): the reference dumps each PSM it is about to score over the
    previous one; what remains is the last PSM with a variable modification, as two pickled one-element lists."""
    import pickle
    spectra, psms = _toy_inputs()
    monkeypatch.chdir(tmp_path)
    picked, scans = batch_cli.select_psms(psms, spectra, "STY", PHOSPHO, hit_depth=1, max_fragment_charge=5, match_save=True)
    with open(tmp_path / "dump_match.pkl", "rb") as f:
        (match,) = pickle.load(f)
    with open(tmp_path / "dump_spectra.pkl", "rb") as f:
        (spec,) = pickle.load(f)
    assert match["scan"] == scans[-1] and match["peptide"] == picked[-1]["peptide"]
    assert np.array_equal(spec["mz_values"], spectra[scans[-1]]["mz_values"])
    # without the flag nothing is written
    monkeypatch.chdir(tmp_path.parent)
    batch_cli.select_psms(psms, spectra, "STY", PHOSPHO)
    assert not (tmp_path.parent / "dump_match.pkl").exists()


@pytest.mark.gpu
def test_rows_match_per_psm_reference_loop(tmp_path):
    from oracle import orc
    from pyascore_amd import PyAscore
    spectra, psms = _toy_inputs()
    gpu = PyAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by")
    rows = batch_cli.localize(gpu, psms, spectra, "STY", PHOSPHO, hit_depth=2, max_fragment_charge=3)
    kind = checker_kind()
    chk = orc.OracleAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by", kind=kind)
    want = []
    for match in psms:                                        # the reference's loop, PSM by PSM
        spectrum = spectra[match["scan"]]
        cpos, cmass, nvar = batch_cli.process_mods("STY", PHOSPHO, match["peptide"], match["mod_positions"],
                                                   match["mod_masses"])
        if nvar > 0:
            chk.score(spectrum["mz_values"], spectrum["intensity_values"], match["peptide"], nvar,
                      min(3, batch_cli.psm_charge(match, spectrum) - 1), cpos, cmass)
            want.append([match["scan"], chk.best_sequence, chk.best_score,
                         ";".join(str(s) for s in chk.ascores),
                         ";".join(",".join(str(s) for s in alt) for alt in chk.alt_sites)])
    assert rows == want
    out = tmp_path / "out.tsv"
    batch_cli.write_tsv(rows, str(out))
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == list(batch_cli.COLUMNS) and len(lines) == len(rows) + 1


@pytest.mark.gpu
def test_one_unscorable_psm_does_not_cost_the_run_its_output():
    """A PSM beyond a limit the reference does not have (here: 600 residues) is written as a row
    without a localisation, with a warning; the other rows are unchanged.  (70 residues -- beyond the fast kernels,
    not beyond the general one -- is scored like any other PSM.)"""
    from pyascore_amd import PyAscore
    spectra, psms = _toy_inputs()
    gpu = PyAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by")
    want = batch_cli.localize(gpu, psms, spectra, "STY", PHOSPHO, hit_depth=2, max_fragment_charge=3)
    long_psm = dict(psms[0], scan=psms[-1]["scan"] + 1, peptide="A" * 590 + "STSTSTSTSK",
                    mod_positions=np.array([591], np.int32), mod_masses=np.array([PHOSPHO]))
    spectra2 = dict(spectra)
    spectra2[long_psm["scan"]] = spectra[psms[0]["scan"]]
    said = []
    with pytest.warns(RuntimeWarning, match="not scored"):
        rows = batch_cli.localize(gpu, psms + [long_psm], spectra2, "STY", PHOSPHO, hit_depth=2, max_fragment_charge=3,
                                  log=said.append)
    # the run's log names how many PSMs were set aside, why, and which scans
    assert len(said) == 2 and said[0].startswith("1 of ") and "peptide length 600" in said[0]
    assert str(long_psm["scan"]) in said[1] and "code" in said[1]
    assert rows[:-1] == want
    assert rows[-1][0] == long_psm["scan"] and rows[-1][1] == "" and np.isnan(rows[-1][2]) and rows[-1][3:] == ["", ""]
    mid_psm = dict(psms[0], scan=psms[-1]["scan"] + 2, peptide="A" * 60 + "STSTSTSTSK",
                   mod_positions=np.array([61], np.int32), mod_masses=np.array([PHOSPHO]))
    spectra2[mid_psm["scan"]] = spectra[psms[0]["scan"]]
    rows = batch_cli.localize(gpu, psms + [mid_psm], spectra2, "STY", PHOSPHO, hit_depth=2, max_fragment_charge=3)
    assert rows[:-1] == want and rows[-1][1].startswith("A" * 60) and "[80]" in rows[-1][1] and not np.isnan(rows[-1][2])

"""Spectrum / identification ingest: the reference's parser tests restated against
the dependency-free readers (test/test_spec_parsers.py, test/test_id_parsers.py), on the reference's
own example files (tests/golden/ingest/, copied data), plus the two table formats on small files
written here, and -- on the GPU -- files in, TSV rows out against a per-PSM loop over the checker."""
import os
import warnings

import numpy as np
import pytest

from conftest import ROOT, checker_kind

from pyascore_amd import batch_cli, ingest

DATA = os.path.join(ROOT, "tests", "golden", "ingest")
PHOSPHO = 79.966331

# test/test_spec_parsers.py:5-9
SCAN_NUMBERS = [14760, 18330, 20462, 21996, 26219, 26962, 27845, 31328, 32257, 35669]
PRECURSOR_MZ = [846.306451825194, 871.696163579367, 858.378601074219, 1116.095703125,
                858.408142089844, 1427.79736328125, 827.992004394531, 1078.430162374319,
                1023.712707519531, 885.028560474252]

# test/test_id_parsers.py:146-185 (the known answers of the reference's reader tests)
TIDE = [
    (14760, 4.48925829, "KMSDDEDDDEEEYGKEEHEK", [3], [79.966331]),
    (14760, 1.60863817, "KMSDDEDDDEEEYGKEEHEK", [13], [79.966331]),
    (18330, 2.81006455, "EDLPAENGETKTEESPASDEAGEK", [18], [79.966331]),
    (18330, 2.54836297, "EDLPAENGETKTEESPASDEAGEK", [15], [79.966331]),
    (20462, 3.30343485, "RRASWASENGETDAEGTQMTPAK", [4], [79.966331]),
    (20462, 2.74408746, "RRASWASENGETDAEGTQMTPAK", [7], [79.966331]),
    (21996, 3.88421941, "AEEPPSQLDQDTQVQDMDEGSDDEEEGQK", [17, 21], [15.9949, 79.966331]),
    (21996, 2.35982895, "AEEPPSQLDQDTQVQDMDEGSDDEEEGQK", [12, 17], [79.966331, 15.9949]),
    (26219, 3.62030768, "GKEELAEAEIIKDSPDSPEPPNK", [17], [79.966331]),
    (26219, 3.5756743, "GKEELAEAEIIKDSPDSPEPPNK", [14], [79.966331]),
    (26962, 3.6686945, "KEDSDEEEDDDSEEDEEDDEDEDEDEDEIEPAAMK", [4, 12], [79.966331, 79.966331]),
    (26962, 0.7472344, "NASNVKHHDSSALGVYSYIPLVENPYFSSWPPSGTSSK", [11, 29], [79.966331, 79.966331]),
    (27845, 2.49656582, "DLGSTEDGDGTDDFLTDKEDEK", [16], [79.966331]),
    (27845, 2.15698767, "DLGSTEDGDGTDDFLTDKEDEK", [11], [79.966331]),
    (31328, 5.46427345, "EGHSLEMENENLVENGADSDEDDNSFLK", [7, 19], [15.9949, 79.966331]),
    (31328, 4.58486271, "EGHSLEMENENLVENGADSDEDDNSFLK", [7, 25], [15.9949, 79.966331]),
    (32257, 3.3405838, "KPATPAEDDEDDDIDLFGSDNEEEDK", [4, 19], [79.966331, 79.966331]),
    (32257, 1.01620567, "AGDMGNCVSGQQQEGGVSEEMKGPVQEDK", [7], [57.021464]),
    (35669, 3.87117219, "VEEESTGDPFGFDSDDESLPVSSK", [14], [79.966331]),
    (35669, 3.24795341, "VEEESTGDPFGFDSDDESLPVSSK", [18], [79.966331]),
]


def test_residue_masses():
    """pyteomics.mass.std_aa_mass to the digits it is usually quoted with."""
    known = dict(G=57.02146, A=71.03711, S=87.03203, P=97.05276, V=99.06841, T=101.04768, C=103.00919,
                 L=113.08406, I=113.08406, N=114.04293, D=115.02694, Q=128.05858, K=128.09496, E=129.04259,
                 M=131.04049, H=137.05891, F=147.06841, R=156.10111, Y=163.06333, W=186.07931)
    for aa, m in known.items():
        assert abs(ingest.STD_AA_MASS[aa] - m) < 6e-6, aa


# ---- MassCorrector: test/test_id_parsers.py:11-100 ----------------------------------------------
def test_corrector_n_terminus():
    c = ingest.MassCorrector()
    for digits in range(6):
        assert [x[0] for x in c.correct("X", 0, round(42.010565, digits))] == ["n", 0, 42.010565]
        m = ingest.STD_AA_MASS["M"] + 42.010565
        assert [x[0] for x in c.correct("M", 1, round(m, digits))] == ["n", 0, 42.010565]


def test_corrector_n_terminus_combined_with_a_residue_modification():
    c = ingest.MassCorrector()
    m = ingest.STD_AA_MASS["M"] + 42.010565 + 15.9949
    for digits in range(6):
        res, pos, mass = c.correct("M", 1, round(m, digits))
        assert (res, pos, mass) == (("n", "M"), (0, 1), (42.010565, 15.9949))


def test_corrector_residue_and_unknown():
    c = ingest.MassCorrector()
    for digits in range(6):
        got = c.correct("S", 5, round(ingest.STD_AA_MASS["S"] + PHOSPHO, digits))
        assert got == (("S",), (5,), (PHOSPHO,))
    # a modification the table does not know is passed through with the residue mass taken off, and a warning
    with pytest.warns(UserWarning, match="Unrecognized mod on M at position 5"):
        res, pos, mass = c.correct("M", 5, ingest.STD_AA_MASS["M"] + PHOSPHO)
    assert res == ("M",) and pos == (5,) and abs(mass[0] - PHOSPHO) < 1e-9
    # n_mod_ind moves where an n-terminal modification is reported
    assert ingest.MassCorrector(n_mod_ind=1).correct("X", 0, 42.01)[1] == (1,)


def test_corrector_whole_peptide():
    c = ingest.MassCorrector()
    pos, mass = c.correct_multiple("MRAMSLVSNEGDSEQNEIR", np.array([1, 5]),
                                   np.array([ingest.STD_AA_MASS["M"] + 42.010565 + 15.9949,
                                             ingest.STD_AA_MASS["S"] + PHOSPHO]))
    assert pos.tolist() == [0, 1, 5] and mass.tolist() == [42.010565, 15.9949, PHOSPHO]


# ---- spectra: test/test_spec_parsers.py:11-45 --------------------------------------------------
@pytest.mark.parametrize("fmt,name", [("mzML", "test_spectra.mzML"), ("mzXML", "test_spectra.mzXML")])
def test_spectra_reader(fmt, name):
    spectra = ingest.SpectraParser(os.path.join(DATA, name), fmt).to_list()
    assert len(spectra) == len(SCAN_NUMBERS)
    for rec, scan, pmz in zip(spectra, SCAN_NUMBERS, PRECURSOR_MZ):
        assert rec["scan"] == scan and rec["ms_level"] == 2
        assert rec["precursor_mz"] == pmz and rec["precursor_charge"] == 3
        assert rec["mz_values"].dtype == np.float64 and rec["intensity_values"].dtype == np.float64
        assert rec["mz_values"].shape == rec["intensity_values"].shape and rec["mz_values"].size > 0
        assert np.all(np.diff(rec["mz_values"]) > 0)
    as_dict = ingest.SpectraParser(os.path.join(DATA, name), fmt).to_dict()
    assert sorted(as_dict) == SCAN_NUMBERS and "scan" not in as_dict[SCAN_NUMBERS[0]]


def test_the_two_spectrum_formats_hold_the_same_peaks():
    a = ingest.SpectraParser(os.path.join(DATA, "test_spectra.mzML"), "mzML").to_list()
    b = ingest.SpectraParser(os.path.join(DATA, "test_spectra.mzXML"), "mzXML").to_list()
    for x, y in zip(a, b):
        assert np.array_equal(x["mz_values"], y["mz_values"])
        assert np.array_equal(x["intensity_values"], y["intensity_values"])


def test_spectra_reader_arguments():
    path = os.path.join(DATA, "test_spectra.mzML")
    assert ingest.SpectraParser(path, "mzML", ms_level=1).to_list() == []
    assert len(ingest.SpectraParser(path, "mzML", ms_level=0).to_list()) == 10
    keep = ingest.SpectraParser(path, "mzML", custom_filter=lambda r: r["scan"] > 30000).to_list()
    assert [r["scan"] for r in keep] == [31328, 32257, 35669]
    with pytest.raises(ValueError, match="not supported"):
        ingest.SpectraParser(path, "mgf")
    with pytest.raises(ValueError, match="ms_level"):
        ingest.SpectraParser(path, "mzML", ms_level=-1)


def test_compressed_and_single_precision_arrays(tmp_path):
    """zlib-compressed 32-bit arrays in mzML, zlib + 32-bit network-order pairs in mzXML."""
    import base64
    import zlib
    mz = np.array([100.5, 200.25, 300.125], np.float32)
    it = np.array([10.0, 20.0, 30.0], np.float64)
    b64 = lambda raw: base64.b64encode(zlib.compress(raw)).decode()
    (tmp_path / "a.mzML").write_text("""<?xml version="1.0"?><mzML xmlns="http://psi.hupo.org/ms/mzml"><run><spectrumList>
<spectrum index="0" id="scan=7"><cvParam name="ms level" value="2"/><binaryDataArrayList count="2">
<binaryDataArray><cvParam name="32-bit float"/><cvParam name="zlib compression"/><cvParam name="m/z array"/><binary>%s</binary></binaryDataArray>
<binaryDataArray><cvParam name="64-bit float"/><cvParam name="zlib compression"/><cvParam name="intensity array"/><binary>%s</binary></binaryDataArray>
</binaryDataArrayList></spectrum></spectrumList></run></mzML>""" % (b64(mz.astype("<f4").tobytes()), b64(it.astype("<f8").tobytes())))
    rec = ingest.SpectraParser(str(tmp_path / "a.mzML"), "mzML").to_list()[0]
    assert rec["scan"] == 7 and rec["precursor_mz"] is None and rec["precursor_charge"] is None
    assert rec["mz_values"].tolist() == mz.astype(np.float64).tolist() and rec["intensity_values"].tolist() == it.tolist()
    pairs = np.empty(6, ">f4")
    pairs[0::2], pairs[1::2] = mz, it
    (tmp_path / "a.mzXML").write_text("""<?xml version="1.0"?><mzXML><msRun><scan num="3" msLevel="1" peaksCount="0">
<scan num="9" msLevel="2" peaksCount="3"><precursorMz precursorCharge="2">445.12</precursorMz>
<peaks precision="32" byteOrder="network" contentType="m/z-int" compressionType="zlib">%s</peaks></scan></scan></msRun></mzXML>"""
                                     % b64(pairs.tobytes()))
    recs = ingest.SpectraParser(str(tmp_path / "a.mzXML"), "mzXML").to_list()
    assert [r["scan"] for r in recs] == [9]                      # the MS1 parent is filtered, its nested MS2 is kept
    assert recs[0]["precursor_mz"] == 445.12 and recs[0]["precursor_charge"] == 2
    assert recs[0]["mz_values"].tolist() == mz.astype(np.float64).tolist() and recs[0]["intensity_values"].tolist() == it.tolist()


# ---- identifications: test/test_id_parsers.py:187-227 ------------------------------------------
def _check(psm, answer):
    scan, score, pep, pos, mass = answer
    assert psm["scan"] == scan and psm["charge_state"] == 3 and psm["score"] == score and psm["peptide"] == pep
    assert psm["mod_positions"].tolist() == pos and psm["mod_masses"].tolist() == mass


def test_pepxml_reader():
    psms = ingest.IdentificationParser(os.path.join(DATA, "test_psms.pep.xml"), "pepXML",
                                       score_string="xcorr_score").to_list()
    assert len(psms) == 20
    for psm, answer in zip(psms, TIDE):
        _check(psm, answer)


def test_mzidentml_reader():
    psms = ingest.IdentificationParser(os.path.join(DATA, "test_psms.mzid"), "mzIdentML",
                                       score_string="SEQUEST:xcorr").to_list()
    assert len(psms) == 20
    for psm, answer in zip(psms[::2], TIDE[::2]):          # (this writer collapses localisations: every other hit)
        _check(psm, answer)


def test_score_threshold_and_score_function():
    path = os.path.join(DATA, "test_psms.pep.xml")
    # kept when score < threshold: the reference's `(-1 ** score_lower_better)` is -1 whatever the flag says
    for flag in (True, False):
        psms = ingest.IdentificationParser(path, "pepXML", score_string="xcorr_score", score_threshold=2.0,
                                           score_lower_better=flag).to_list()
        assert [p["score"] for p in psms] == [a[1] for a in TIDE if a[1] < 2.0]
    psms = ingest.IdentificationParser(path, "pepXML", score_string="xcorr_score", score_func=lambda s: -s,
                                       score_threshold=-4.0).to_list()
    assert [p["score"] for p in psms] == [-a[1] for a in TIDE if a[1] > 4.0]
    # a score that is not in the file: None without a threshold, dropped with one
    assert ingest.IdentificationParser(path, "pepXML", score_string="nope").to_list()[0]["score"] is None
    assert ingest.IdentificationParser(path, "pepXML", score_string="nope", score_threshold=1.0).to_list() == []
    as_dict = ingest.IdentificationParser(path, "pepXML", score_string="xcorr_score").to_dict()
    assert sorted(as_dict) == SCAN_NUMBERS and as_dict[14760]["score"] == 1.60863817      # later hits overwrite
    with pytest.raises(ValueError, match="not supported"):
        ingest.IdentificationParser(path, "sqt")


def test_pepxml_terminal_modifications(tmp_path):
    (tmp_path / "t.pep.xml").write_text("""<?xml version="1.0"?><msms_pipeline_analysis xmlns="http://regis-web.systemsbiology.net/pepXML">
<msms_run_summary><spectrum_query start_scan="5" assumed_charge="2"><search_result>
<search_hit hit_rank="1" peptide="MSTK"><modification_info mod_nterm_mass="43.0184" mod_cterm_mass="18.5">
<mod_aminoacid_mass position="2" mass="166.9984"/></modification_info><search_score name="xcorr" value="1.5"/></search_hit>
</search_result></spectrum_query><spectrum_query start_scan="2" assumed_charge="3"><search_result>
<search_hit hit_rank="1" peptide="ACK"><search_score name="xcorr" value="0.5"/></search_hit></search_result></spectrum_query>
</msms_run_summary></msms_pipeline_analysis>""")
    ex = ingest.PepXMLExtractor("xcorr")
    import xml.etree.ElementTree as ET
    queries = [q for q in ET.parse(str(tmp_path / "t.pep.xml")).getroot().iter() if q.tag.endswith("spectrum_query")]
    rec = ex.extract(queries[0])
    assert rec["scans"] == [5] and rec["charge_states"] == [2] and rec["peptides"] == ["MSTK"] and rec["scores"] == [1.5]
    assert rec["mod_positions"][0].tolist() == [0, 2, 5] and rec["mod_positions"][0].dtype == np.int32
    assert np.allclose(rec["mod_masses"][0], [43.0184, 166.9984, 18.5]) and rec["mod_masses"][0].dtype == np.float32
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        psms = ingest.IdentificationParser(str(tmp_path / "t.pep.xml"), "pepXML", score_string="xcorr").to_list()
    assert [p["scan"] for p in psms] == [2, 5]                # sorted by scan
    assert psms[0]["mod_positions"].size == 0
    assert psms[1]["mod_positions"].tolist()[:2] == [0, 2] and psms[1]["mod_masses"].tolist()[1] == PHOSPHO


def test_percolator_and_mokapot_tables(tmp_path):
    (tmp_path / "p.txt").write_text(
        "file_idx\tscan\tcharge\tpercolator score\tsequence\n"
        "0\t30\t2\t1.25\tn[42.01]AS[79.97]TCK\n"
        "0\t12\t3\t0.75\tPEPT[79.97]IDEC[57.02]K\n"
        "0\t30\t2\t0.5\tAST[79.97]CK\n")
    psms = ingest.IdentificationParser(str(tmp_path / "p.txt"), "percolatorTXT").to_list()
    assert [(p["scan"], p["charge_state"], p["score"], p["peptide"]) for p in psms] == [
        (12, 3, 0.75, "PEPTIDECK"), (30, 2, 1.25, "ASTCK"), (30, 2, 0.5, "ASTCK")]
    assert psms[0]["mod_positions"].tolist() == [4, 8] and psms[0]["mod_masses"].tolist() == [PHOSPHO, 57.021464]
    # unmodified C gets the static modification; the n-terminal acetyl comes first
    assert psms[1]["mod_positions"].tolist() == [0, 2, 4] and psms[1]["mod_masses"].tolist() == [42.010565, PHOSPHO, 57.021464]
    assert psms[2]["mod_positions"].tolist() == [3, 4]
    (tmp_path / "m.txt").write_text(
        "SpecId\tLabel\tScanNr\tmokapot score\tPeptide\n"
        "a\tTrue\t8\t2.5\tK.AS[79.97]TM[15.99]K.R\n"
        "b\tTrue\t3\t-0.5\t-.n[42.01]MSK.A\n")
    psms = ingest.IdentificationParser(str(tmp_path / "m.txt"), "mokapotTXT", static_mods={}).to_list()
    assert [(p["scan"], p["charge_state"], p["score"], p["peptide"]) for p in psms] == [(3, None, -0.5, "MSK"), (8, None, 2.5, "ASTMK")]
    assert psms[0]["mod_positions"].tolist() == [0] and psms[0]["mod_masses"].tolist() == [42.010565]
    assert psms[1]["mod_positions"].tolist() == [2, 4] and psms[1]["mod_masses"].tolist() == [PHOSPHO, 15.9949]


def test_files_to_one_batch():
    spectra = ingest.SpectraParser(os.path.join(DATA, "test_spectra.mzML"), "mzML").to_dict()
    psms = ingest.IdentificationParser(os.path.join(DATA, "test_psms.pep.xml"), "pepXML", score_string="xcorr_score").to_list()
    batch, scans = ingest.to_batch(psms, spectra, "STY", PHOSPHO, hit_depth=1)
    assert scans == SCAN_NUMBERS and batch["n_psm"] == 10
    assert batch["n_of_mod"].tolist() == [1, 1, 1, 1, 1, 2, 1, 1, 2, 1]
    assert batch["max_charge"].tolist() == [2] * 10               # min(5, max(3, 2) - 1)
    assert int(batch["aux_off"][-1]) == 2                          # the two oxidised methionines are fixed modifications
    assert int(batch["peak_off"][1]) == 313
    batch2, scans2 = ingest.to_batch(psms, spectra, "STY", PHOSPHO, hit_depth=2)
    assert len(scans2) == 19                                       # the carbamidomethyl-only hit of scan 32257 carries no phosphate


def test_reference_import_names():
    import pyascore
    from pyascore import id_parsers, spec_parsers
    assert id_parsers.MassCorrector is ingest.MassCorrector and spec_parsers.SpectraParser is ingest.SpectraParser
    assert pyascore.IdentificationParser is ingest.IdentificationParser and pyascore.COMMON_MODS["S"] == PHOSPHO
    assert pyascore.STD_AA_MASS is ingest.STD_AA_MASS and pyascore.MzMLExtractor is ingest.MzMLExtractor


@pytest.mark.gpu
def test_files_in_rows_out(tmp_path):
    """The reference CLI's whole job on its own example files: parse, score, write -- one batched call
    on the GPU against the per-PSM loop of `pyascore/__main__.py:127-164` over the checker."""
    from oracle import orc
    from pyascore_amd import PyAscore
    spectra = ingest.SpectraParser(os.path.join(DATA, "test_spectra.mzML"), "mzML").to_dict()
    psms = ingest.IdentificationParser(os.path.join(DATA, "test_psms.pep.xml"), "pepXML", score_string="xcorr_score").to_list()
    gpu = PyAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by")
    rows = batch_cli.localize(gpu, psms, spectra, "STY", PHOSPHO, hit_depth=2, max_fragment_charge=3)
    chk = orc.OracleAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by", kind=checker_kind())
    want = []
    for match in psms:
        spectrum = spectra[match["scan"]]
        cpos, cmass, nvar = batch_cli.process_mods("STY", PHOSPHO, match["peptide"], match["mod_positions"], match["mod_masses"])
        if nvar > 0:
            chk.score(spectrum["mz_values"], spectrum["intensity_values"], match["peptide"], nvar,
                      min(3, batch_cli.psm_charge(match, spectrum) - 1), cpos, cmass)
            want.append([match["scan"], chk.best_sequence, chk.best_score, ";".join(str(s) for s in chk.ascores),
                         ";".join(",".join(str(s) for s in alt) for alt in chk.alt_sites)])
    assert len(rows) == 19 and rows == want
    batch_cli.write_tsv(rows, str(tmp_path / "out.tsv"))
    assert len((tmp_path / "out.tsv").read_text().splitlines()) == 20


def test_command_line_arguments(tmp_path):
    """Option names, defaults and the parameter file of the reference's CLI (config.py:5-93)."""
    from pyascore_amd import __main__ as cli
    args = cli.parse_args(["a.mzML", "b.pep.xml", "out.tsv"])
    assert (args.residues, args.mod_mass, args.mz_error, args.mod_correction_tol, args.fragment_types) == \
        ("STY", 79.966331, 0.5, 1.0, "by")
    assert (args.max_fragment_charge, args.hit_depth, args.spec_file_type, args.ident_file_type) == (5, 1, "mzML", "pepXML")
    assert cli.static_mods_of(args) == {"C": 57.021464}
    (tmp_path / "p.txt").write_text("# comment\nresidues = ST   # trailing\nmz_error=0.05\nhit_depth = 2\nstatic_mod_groups = C,K\n"
                                    "static_mod_masses = 57.021464,8.014199\nnot a parameter line\n")
    args = cli.parse_args(["--parameter_file", str(tmp_path / "p.txt"), "--hit_depth", "3", "a", "b", "c"])
    assert (args.residues, args.mz_error, args.hit_depth) == ("ST", 0.05, 3)        # the command line wins
    assert cli.static_mods_of(args) == {"C": 57.021464, "K": 8.014199}
    for bad in (["--residues", "STB"], ["--fragment_types", "bx"], ["--max_fragment_charge", "0"]):
        with pytest.raises(ValueError):
            cli.parse_args(bad + ["a", "b", "c"])


@pytest.mark.gpu
def test_command_line_end_to_end(tmp_path):
    from pyascore_amd import PyAscore, __main__ as cli
    out = tmp_path / "ascores.tsv"
    rows = cli.run(cli.parse_args(["--mz_error", "0.05", "--hit_depth", "2", os.path.join(DATA, "test_spectra.mzML"),
                                   os.path.join(DATA, "test_psms.pep.xml"), str(out)]), log=lambda *_: None)
    spectra = ingest.SpectraParser(os.path.join(DATA, "test_spectra.mzML"), "mzML").to_dict()
    psms = ingest.IdentificationParser(os.path.join(DATA, "test_psms.pep.xml"), "pepXML").to_list()
    want = batch_cli.localize(PyAscore(100.0, 10, "STY", PHOSPHO, 0.05, "by"), psms, spectra, "STY", PHOSPHO, hit_depth=2)
    assert rows == want and len(rows) == 19
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == list(batch_cli.COLUMNS) and len(lines) == 20
    # the other pair of formats gives the same rows for the hits both files list in full
    rows2 = cli.run(cli.parse_args(["--mz_error", "0.05", "--spec_file_type", "mzXML", "--ident_file_type", "mzIdentML",
                                    os.path.join(DATA, "test_spectra.mzXML"), os.path.join(DATA, "test_psms.mzid"),
                                    str(tmp_path / "b.tsv")]), log=lambda *_: None)
    first_hits = [r for i, r in enumerate(want) if i == 0 or want[i - 1][0] != r[0]]
    assert rows2 == first_hits


def test_mass_corrector_numpy_form():
    """MassCorrector.correct_numpy, the reference's deprecated array form (id_parsers.py:133-180), on cases whose
    answers follow from its rules by hand: residue + modification masses come back as the known modification;
    an n-terminal acetylation reported at position 0, folded into residue 1, or merged with a modification of
    residue 1 is split off; an unknown mass is an error that names position and mass."""
    from pyascore_amd.ingest import MassCorrector, STD_AA_MASS, COMMON_MODS
    mc = MassCorrector(mz_tol=1.5)
    ser, met, n_ac = STD_AA_MASS["S"], STD_AA_MASS["M"], COMMON_MODS["n"]
    # empty
    pos, mass = mc.correct_numpy("PEPTIDE", np.array([]), np.array([]))
    assert pos.size == 0 and mass.size == 0
    # two modified residues, masses reported as residue + (rounded) modification
    pos, mass = mc.correct_numpy("ASMK", np.array([2, 3]), np.array([ser + 79.97, met + 16.0]))
    assert pos.tolist() == [2, 3] and np.allclose(mass, [COMMON_MODS["S"], COMMON_MODS["M"]], rtol=0, atol=1e-12)
    # n-terminal modification reported at position 0
    pos, mass = mc.correct_numpy("SAK", np.array([0, 1]), np.array([42.01, ser + 79.97]))
    assert pos.tolist() == [0, 1] and np.allclose(mass, [n_ac, COMMON_MODS["S"]], rtol=0, atol=1e-12)
    # ... folded into the first residue's mass
    pos, mass = mc.correct_numpy("AMK", np.array([1, 2]), np.array([STD_AA_MASS["A"] + 42.01, met + 15.99]))
    assert pos.tolist() == [0, 2] and np.allclose(mass, [n_ac, COMMON_MODS["M"]], rtol=0, atol=1e-12)
    # ... merged with a modification of the first residue: split, and the caller's mass loses the n-terminal part
    reported = np.array([ser + 79.966331 + n_ac])
    pos, mass = mc.correct_numpy("SAK", np.array([1]), reported)
    assert pos.tolist() == [0, 1] and np.allclose(mass, [n_ac, COMMON_MODS["S"]], rtol=0, atol=1e-12)
    assert abs(reported[0] - (ser + 79.966331)) < 1e-9
    # an unknown mass
    with pytest.raises(ValueError, match="Unrecognized mod at positions"):
        mc.correct_numpy("ASK", np.array([2]), np.array([ser + 120.0]))

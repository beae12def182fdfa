"""Spectrum and identification ingest: the step before the hot path.

The reference reads its inputs through pyteomics (`pyascore/parsing/spec_parsers.py:49-282`,
`pyascore/parsing/id_parsers.py:22-815`); this module provides the same two front ends --
:class:`SpectraParser` (mzML, mzXML) and :class:`IdentificationParser` (pepXML, mzIdentML,
percolatorTXT, mokapotTXT) -- and :class:`MassCorrector` with nothing but the standard library
(``xml.etree`` streaming, ``base64``, ``zlib``, ``csv``) and numpy.  Output records have the
reference's schema, field for field:

* spectra: ``{scan, ms_level, precursor_mz, precursor_charge, mz_values f64, intensity_values f64}``,
  sorted by scan (spec_parsers.py:243-282);
* identifications: ``{scan, charge_state, score, peptide, mod_positions, mod_masses}`` in file order
  within a spectrum, spectra sorted by scan (id_parsers.py:735-815).

These are exactly what :func:`pyascore_amd.batch_cli.localize` takes, so
``localize(ascore, IdentificationParser(...).to_list(), SpectraParser(...).to_dict(), ...)`` is the
whole pipeline from files to TSV rows; :func:`to_batch` packs them into the CSR arrays of one
``score_batch`` call directly.
"""
import base64
import csv
import re
import warnings
import xml.etree.ElementTree as ET
import zlib

import numpy as np

__all__ = ["STD_AA_MASS", "COMMON_MODS", "MassCorrector", "SpectraParser", "IdentificationParser",
           "MzMLExtractor", "MzXMLExtractor", "PepXMLExtractor", "MzIdentMLExtractor",
           "PercolatorTXTExtractor", "MokapotTXTExtractor", "to_batch"]

# ------------------------------------------------------------------------------------------------
# masses
# ------------------------------------------------------------------------------------------------
# Monoisotopic element masses (the NIST values the reference's dependency tabulates) and residue
# compositions (C, H, N, O, S, Se): residue mass = sum of its atoms, as pyteomics.mass.std_aa_mass
# computes it (id_parsers.py:11).
_ELEMENT = dict(H=1.00782503207, C=12.0, N=14.0030740048, O=15.99491461956, S=31.972071, Se=79.9165213)
_RESIDUE_FORMULA = {
    "G": (2, 3, 1, 1, 0, 0), "A": (3, 5, 1, 1, 0, 0), "S": (3, 5, 1, 2, 0, 0), "P": (5, 7, 1, 1, 0, 0),
    "V": (5, 9, 1, 1, 0, 0), "T": (4, 7, 1, 2, 0, 0), "C": (3, 5, 1, 1, 1, 0), "L": (6, 11, 1, 1, 0, 0),
    "I": (6, 11, 1, 1, 0, 0), "J": (6, 11, 1, 1, 0, 0), "N": (4, 6, 2, 2, 0, 0), "D": (4, 5, 1, 3, 0, 0),
    "Q": (5, 8, 2, 2, 0, 0), "K": (6, 12, 2, 1, 0, 0), "E": (5, 7, 1, 3, 0, 0), "M": (5, 9, 1, 1, 1, 0),
    "H": (6, 7, 3, 1, 0, 0), "F": (9, 9, 1, 1, 0, 0), "R": (6, 12, 4, 1, 0, 0), "Y": (9, 9, 1, 2, 0, 0),
    "W": (11, 10, 2, 1, 0, 0), "U": (3, 5, 1, 1, 0, 1), "O": (12, 19, 3, 2, 0, 0),
}
STD_AA_MASS = {
    aa: (c * _ELEMENT["C"] + h * _ELEMENT["H"] + n * _ELEMENT["N"] + o * _ELEMENT["O"] + s * _ELEMENT["S"]
         + se * _ELEMENT["Se"])
    for aa, (c, h, n, o, s, se) in _RESIDUE_FORMULA.items()
}

# id_parsers.py:14-20: used when the caller supplies no modification table
COMMON_MODS = {"n": 42.010565, "M": 15.9949, "K": 8.014199, "S": 79.966331, "T": 79.966331,
               "Y": 79.966331, "C": 57.021464}


def _close(a, b, tol):
    """numpy.isclose(a, b, rtol=0, atol=tol) for scalars (inf on either side never matches)."""
    return bool(np.isfinite(b)) and abs(a - b) <= tol


class MassCorrector:
    """Undoes the two things search engines do to modification masses (id_parsers.py:22-131): rounding
    (79.97 for 79.966331) and folding the residue mass -- or an n-terminal modification -- into the
    reported number.

    ``mod_mass_dict``: known modification per residue ('n' = n-terminus); ``aa_mass_dict`` residue
    masses; ``mz_tol`` how far a reported mass may be from a known combination; ``n_mod_ind`` the
    position an n-terminal modification is reported at (0 = before the first residue)."""

    def __init__(self, mod_mass_dict=COMMON_MODS, aa_mass_dict=STD_AA_MASS, mz_tol=1.5, n_mod_ind=0):
        self.mod_mass_dict = mod_mass_dict
        self.aa_mass_dict = aa_mass_dict
        self.mz_tol = mz_tol
        self.n_mod_ind = n_mod_ind

    def correct(self, res, pos, mass):
        """One reported (residue, position, mass) -> tuples (residues), (positions), (masses) of length 1,
        or 2 when an n-terminal modification had been merged with one on the first residue."""
        inf = float("inf")
        std = STD_AA_MASS.get(res, inf)
        mod = self.mod_mass_dict.get(res, inf)
        n_mod = self.mod_mass_dict.get("n", inf)
        if pos == 0 and _close(mass, n_mod, self.mz_tol):
            return ("n",), (self.n_mod_ind,), (n_mod,)
        if pos == 1 and _close(mass, std + n_mod, self.mz_tol):
            return ("n",), (self.n_mod_ind,), (n_mod,)
        if pos == 1 and _close(mass, std + mod + n_mod, self.mz_tol):
            return ("n", res), (self.n_mod_ind, pos), (n_mod, mod)
        if _close(mass, std + mod, self.mz_tol):
            return (res,), (pos,), (mod,)
        guess = mass - STD_AA_MASS.get(res, 0.0)
        warnings.warn("Unrecognized mod on {} at position {} with mass: {}"
                      " Using uncorrected mass.".format(res, pos, guess))
        return (res,), (pos,), (guess,)

    def correct_multiple(self, peptide, positions, masses):
        """All modifications of one peptide -> (positions array, masses array)."""
        out_pos, out_mass = [], []
        for pos, mass in zip(positions, masses):
            pos = int(pos)
            # (0 = n-terminus; len + 1 = c-terminus, where the reference's own loop runs off the sequence)
            res = "n" if pos == 0 else ("c" if pos == len(peptide) + 1 else peptide[pos - 1])
            _, p, m = self.correct(res, pos, float(mass))
            out_pos.extend(p)
            out_mass.extend(m)
        return np.array(out_pos), np.array(out_mass)

    def correct_numpy(self, peptide, positions, masses):
        """The reference's deprecated array form (id_parsers.py:133-180), kept for callers that still use it: the
        first reported modification is tried as an n-terminal one (at position 0; folded into the first residue's
        mass at position 1; or merged with a modification of the first residue, in which case the n-terminal part
        is split off and -- as in the reference -- subtracted from ``masses[0]`` in place); every other reported
        mass must equal residue + known modification within ``mz_tol`` or a ValueError names the offenders.
        Returns (positions array, masses array)."""
        positions = np.asarray(positions)
        if positions.size == 0:
            return np.array([]), np.array([])
        inf = float("inf")
        out_pos, out_mass = [], []
        first = peptide[0]
        n_mod = self.mod_mass_dict.get("n", inf)
        res0 = STD_AA_MASS.get(first, inf)
        if positions[0] == 0 and _close(masses[0], n_mod, self.mz_tol):
            out_pos.append(0.)
            out_mass.append(n_mod)
            positions, masses = positions[1:], masses[1:]
        elif positions[0] == 1 and _close(masses[0], res0 + n_mod, self.mz_tol):
            out_pos.append(0.)
            out_mass.append(n_mod)
            positions, masses = positions[1:], masses[1:]
        elif positions[0] == 1 and _close(masses[0], res0 + self.mod_mass_dict.get(first, inf) + n_mod, self.mz_tol):
            out_pos.append(0)
            out_mass.append(n_mod)
            masses[0] -= n_mod
        if positions.size > 0:
            on_residue = [int(p) for p in positions if p != 0]
            std = np.array([STD_AA_MASS.get(peptide[p - 1], inf) for p in on_residue])
            mods = np.array([self.mod_mass_dict.get(peptide[p - 1], inf) for p in on_residue])
            ok = np.isclose(masses, std + mods, rtol=0., atol=self.mz_tol)
            if np.any(~ok):
                raise ValueError("Unrecognized mod at positions, {}, with masses, {}".format(
                    positions[np.where(~ok)], np.asarray(masses)[np.where(~ok)]))
            out_pos.extend(positions)
            out_mass.extend(mods)
        return np.array(out_pos), np.array(out_mass)


# ------------------------------------------------------------------------------------------------
# XML helpers
# ------------------------------------------------------------------------------------------------
def _local(tag):
    return tag.rsplit("}", 1)[-1]


def _children(elem, name):
    return [c for c in elem if _local(c.tag) == name]


def _first(elem, name):
    for c in elem:
        if _local(c.tag) == name:
            return c
    return None


def _descend(elem, *names):
    for name in names:
        if elem is None:
            return None
        elem = _first(elem, name)
    return elem


def _stream(path, wanted):
    """Yields every element whose local tag is in `wanted` as its end tag arrives, and frees what has
    been read (the files are large; nothing is kept but the element in hand)."""
    stack = []
    for event, elem in ET.iterparse(path, events=("start", "end")):
        if event == "start":
            stack.append(elem)
            continue
        stack.pop()
        name = _local(elem.tag)
        if name in wanted and not any(_local(s.tag) in wanted for s in stack):
            yield elem
            elem.clear()
        elif not any(_local(s.tag) in wanted for s in stack):
            elem.clear()                       # outside anything wanted: never needed again


def _number(text):
    """int when the text is one, else float, else the text (how XML attribute values come out of the
    reference's reader)."""
    try:
        return int(text)
    except (TypeError, ValueError):
        pass
    try:
        return float(text)
    except (TypeError, ValueError):
        return text


def _first_scan_number(text, liberal=True):
    m = re.search(r"(?<=scan=)([0-9]+)", text)
    if m is None and liberal:
        m = re.search(r"[0-9]+", text)
    return int(m.group()) if m else None


# ------------------------------------------------------------------------------------------------
# spectra
# ------------------------------------------------------------------------------------------------
_EMPTY = np.array([], dtype=np.float64)


def _decode(text, dtype, compressed):
    raw = base64.b64decode(text or "")
    if compressed and raw:
        raw = zlib.decompress(raw)
    return np.frombuffer(raw, dtype=dtype)


class MzMLExtractor:
    """<spectrum> element -> record (spec_parsers.py:49-113)."""

    def extract(self, spectrum):
        params = {}
        for cv in _children(spectrum, "cvParam"):
            params[cv.get("name")] = cv.get("value")
        sid = spectrum.get("id")
        scan = -1
        if sid is not None:
            scan = _first_scan_number(sid, liberal=False)
            if scan is None:
                raise AttributeError("no 'scan=' in spectrum id %r" % sid)
        ms_level = int(params["ms level"]) if "ms level" in params else 0
        precursor_mz, precursor_charge = None, None
        plist = _first(spectrum, "precursorList")
        if plist is not None:
            if int(plist.get("count", "1")) > 1:
                raise ValueError("Multiple precursors not supported at this time")
            ion = _descend(plist, "precursor", "selectedIonList", "selectedIon")
            found = {}
            if ion is not None:
                for cv in _children(ion, "cvParam"):
                    found[cv.get("name")] = cv.get("value")
            if "selected ion m/z" in found and "charge state" in found:
                precursor_mz, precursor_charge = float(found["selected ion m/z"]), int(found["charge state"])
        mz, inten = _EMPTY, _EMPTY
        arrays = _first(spectrum, "binaryDataArrayList")
        got = {}
        if arrays is not None:
            for arr in _children(arrays, "binaryDataArray"):
                names = {cv.get("name") for cv in _children(arr, "cvParam")}
                dtype = "<f8" if "64-bit float" in names else ("<f4" if "32-bit float" in names else None)
                kind = "mz" if "m/z array" in names else ("int" if "intensity array" in names else None)
                if dtype is None or kind is None:
                    continue
                binary = _first(arr, "binary")
                got[kind] = _decode(binary.text if binary is not None else "", dtype,
                                    "zlib compression" in names).astype(np.float64)
        if "mz" in got and "int" in got:
            mz, inten = got["mz"], got["int"]
        return {"scan": scan, "ms_level": ms_level, "precursor_mz": precursor_mz,
                "precursor_charge": precursor_charge, "mz_values": mz, "intensity_values": inten}


class MzXMLExtractor:
    """<scan> element -> record (spec_parsers.py:115-172)."""

    def extract(self, scan):
        num = int(scan.get("num")) if scan.get("num") is not None else -1
        ms_level = int(scan.get("msLevel")) if scan.get("msLevel") is not None else 0
        precursor_mz, precursor_charge = None, None
        precursors = _children(scan, "precursorMz")
        if precursors:
            if len(precursors) > 1:
                raise ValueError("Multiple precursors not supported at this time")
            p = precursors[0]
            if p.get("precursorCharge") is not None:
                precursor_mz, precursor_charge = float(p.text), int(p.get("precursorCharge"))
        mz, inten = _EMPTY, _EMPTY
        peaks = _first(scan, "peaks")
        if peaks is not None and (peaks.text or "").strip():
            width = "f8" if peaks.get("precision", "32") == "64" else "f4"
            order = "<" if peaks.get("byteOrder", "network") != "network" else ">"
            pairs = _decode(peaks.text.strip(), order + width, peaks.get("compressionType", "none") == "zlib")
            pairs = pairs.astype(np.float64)
            mz, inten = pairs[0::2].copy(), pairs[1::2].copy()
        return {"scan": num, "ms_level": ms_level, "precursor_mz": precursor_mz,
                "precursor_charge": precursor_charge, "mz_values": mz, "intensity_values": inten}


def _mzxml_scans(path):
    """<scan> elements of an mzXML file, nested ones (MSn inside their MS1 parent) included; each is
    handed over as its own end tag arrives, before its parent is complete."""
    for event, elem in ET.iterparse(path, events=("end",)):
        if _local(elem.tag) == "scan":
            yield elem
            for child in list(elem):
                if _local(child.tag) == "peaks":
                    child.clear()


class SpectraParser:
    """Spectra of one mzML / mzXML file (spec_parsers.py:175-282).

    ``ms_level``: only scans of this MSn level are returned (0 = all); ``custom_filter``: a callable
    record -> bool.  ``to_list()`` is sorted by scan number, ``to_dict()`` maps scan -> record (without
    its "scan" field)."""

    def __init__(self, spec_file_name, spec_file_format, ms_level=2, custom_filter=None):
        if spec_file_format == "mzML":
            self._records = lambda: (MzMLExtractor().extract(s) for s in _stream(spec_file_name, {"spectrum"}))
        elif spec_file_format == "mzXML":
            self._records = lambda: (MzXMLExtractor().extract(s) for s in _mzxml_scans(spec_file_name))
        else:
            raise ValueError("{} not supported at this time."
                             " Should be one of: mzML or mzXML".format(spec_file_format))
        if ms_level < 0:
            raise ValueError("ms_level must be an integer greater than or equal to 0")
        self.ms_level = ms_level
        if custom_filter is not None and not callable(custom_filter):
            raise ValueError("custom_filter must be callable.")
        self.custom_filter = custom_filter
        self._spectra = []

    def _keep(self, rec):
        if self.ms_level and rec["ms_level"] != self.ms_level:
            return False
        return self.custom_filter is None or bool(self.custom_filter(rec))

    def _load(self):
        if not self._spectra:
            self._spectra = sorted((r for r in self._records() if self._keep(r)), key=lambda r: r["scan"])

    def to_list(self):
        self._load()
        return self._spectra

    def to_dict(self):
        self._load()
        return {rec.pop("scan"): rec for rec in self._spectra}


# ------------------------------------------------------------------------------------------------
# identifications
# ------------------------------------------------------------------------------------------------
class _Extractor:
    """One spectrum's entry -> parallel lists over its hits (id_parsers.py:223-284)."""

    def __init__(self, score_string=None, static_mods=None):
        self.score_string = score_string
        self.static_mods = static_mods or {}

    def extract(self, entry):
        hits = self._hits(entry)
        out = {k: [None] * len(hits) for k in ("scans", "scores", "charge_states", "peptides",
                                                "mod_positions", "mod_masses")}
        for i, hit in enumerate(hits):
            out["scans"][i] = self._scan(entry, hit)
            out["scores"][i] = self._score(entry, hit)
            out["charge_states"][i] = self._charge(entry, hit)
            out["peptides"][i] = self._peptide(entry, hit)
            out["mod_positions"][i], out["mod_masses"][i] = self._mods(entry, hit)
        return out


def _no_mods():
    return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32)


class PepXMLExtractor(_Extractor):
    """entry = a <spectrum_query> element (id_parsers.py:388-473)."""

    def _hits(self, q):
        return [h for r in _children(q, "search_result") for h in _children(r, "search_hit")]

    def _scan(self, q, hit):
        return int(q.get("start_scan")) if q.get("start_scan") is not None else -1

    def _charge(self, q, hit):
        return int(q.get("assumed_charge")) if q.get("assumed_charge") is not None else 0

    def _score(self, q, hit):
        for s in _children(hit, "search_score"):
            if s.get("name") == self.score_string:
                return float(s.get("value"))
        return None

    def _peptide(self, q, hit):
        return hit.get("peptide", "")

    def _mods(self, q, hit):
        infos = _children(hit, "modification_info")
        if not infos:
            return _no_mods()
        info = infos[-1]                                     # (a hit with two of them: the last one counts)
        pos, mass = [], []
        if info.get("mod_nterm_mass") is not None:           # reported at position 0, ahead of the residues
            pos.append(0)
            mass.append(float(info.get("mod_nterm_mass")))
        for m in _children(info, "mod_aminoacid_mass"):
            pos.append(int(m.get("position")))
            mass.append(float(m.get("mass")))
        if info.get("mod_cterm_mass") is not None:
            pos.append(len(hit.get("peptide", "")) + 1)
            mass.append(float(info.get("mod_cterm_mass")))
        return np.array(pos, dtype=np.int32), np.array(mass, dtype=np.float32)


class MzIdentMLExtractor(_Extractor):
    """entry = (<SpectrumIdentificationResult>, {peptide id -> <Peptide>}) (id_parsers.py:286-386)."""

    def _hits(self, entry):
        return _children(entry[0], "SpectrumIdentificationItem")

    def _scan(self, entry, hit):
        sid = entry[0].get("spectrumID")
        if sid is None:
            return -1
        return _first_scan_number(sid)

    def _charge(self, entry, hit):
        return int(hit.get("chargeState")) if hit.get("chargeState") is not None else 0

    def _score(self, entry, hit):
        for p in list(_children(hit, "cvParam")) + list(_children(hit, "userParam")):
            if p.get("name") == self.score_string:
                return float(p.get("value"))
        return None

    def _peptide_elem(self, entry, hit):
        return entry[1].get(hit.get("peptide_ref"))

    def _peptide(self, entry, hit):
        pep = self._peptide_elem(entry, hit)
        seq = _first(pep, "PeptideSequence") if pep is not None else None
        return seq.text if seq is not None and seq.text else ""

    def _mods(self, entry, hit):
        pep = self._peptide_elem(entry, hit)
        mods = _children(pep, "Modification") if pep is not None else []
        if not mods:
            return _no_mods()
        sequence = self._peptide(entry, hit)
        pos = np.zeros(len(mods), dtype=np.int32)
        mass = np.zeros(len(mods), dtype=np.float32)
        for i, m in enumerate(mods):
            pos[i] = int(m.get("location"))
            aa = m.get("residues")[0] if m.get("residues") else ("n" + sequence + "c")[pos[i]]
            mass[i] = STD_AA_MASS.get(aa, 0.0) + float(m.get("monoisotopicMassDelta"))
        return pos, mass


_BRACKET = r"\[[^A-z]+\]"


def _bracket_mods(sequence, static_mods):
    """Positions / masses from a sequence with inline deltas, 'n[42.01]PEPT[79.97]IDE'
    (id_parsers.py:530-562): the residue mass is added to every reported delta; residues without one
    get their static modification."""
    pos, mass = [], []
    nterm = re.match(r"n?" + _BRACKET, sequence)
    if nterm is not None:
        mass.append(float(re.search(r"(?<=\[)[^A-z]+(?=\])", nterm.group()).group()))
        pos.append(0)
    elif "n" in static_mods:
        mass.append(static_mods["n"])
        pos.append(0)
    for ind, res in enumerate((m.group() for m in re.finditer(r"[A-Z](" + _BRACKET + r")?", sequence)), 1):
        delta = re.search(r"(?<=\[)[^A-z]+(?=\])", res)
        if delta is not None:
            mass.append(STD_AA_MASS[res[0]] + float(delta.group()))
            pos.append(ind)
        elif res[0] in static_mods:
            mass.append(STD_AA_MASS[res[0]] + static_mods[res[0]])
            pos.append(ind)
    return np.array(pos, dtype=np.int32), np.array(mass, dtype=np.float32)


class PercolatorTXTExtractor(_Extractor):
    """entry = the rows (dicts) of one scan of a Percolator tab-delimited file (id_parsers.py:475-562)."""

    def _hits(self, rows):
        return rows

    def _scan(self, rows, row):
        return _number(row["scan"])

    def _charge(self, rows, row):
        return _number(row["charge"])

    def _score(self, rows, row):
        return _number(row["percolator score"])

    def _peptide(self, rows, row):
        return re.sub(r"n|(" + _BRACKET + r")", "", row["sequence"])

    def _mods(self, rows, row):
        return _bracket_mods(row["sequence"], self.static_mods)


class MokapotTXTExtractor(_Extractor):
    """entry = the rows of one scan of a mokapot PSM table (id_parsers.py:564-654): flanking residues
    'K.PEPTIDE.R' are stripped; the table has no charge column."""

    @staticmethod
    def _bare(row):
        return re.sub(r"(^.\.)|(\..$)", "", row["Peptide"])

    def _hits(self, rows):
        return rows

    def _scan(self, rows, row):
        return _number(row["ScanNr"])

    def _charge(self, rows, row):
        return None

    def _score(self, rows, row):
        return _number(row["mokapot score"])

    def _peptide(self, rows, row):
        return re.sub(r"n|(" + _BRACKET + r")", "", self._bare(row))

    def _mods(self, rows, row):
        return _bracket_mods(self._bare(row), self.static_mods)


def _table_groups(path, key):
    """Rows of a tab-delimited file grouped by `key`, groups in ascending key order, rows of a group in
    file order (what the reference gets from a pandas group-by)."""
    groups = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f, delimiter="\t"):
            groups.setdefault(_number(row[key]), []).append(row)
    return [groups[k] for k in sorted(groups)]


def _mzid_entries(path):
    """(result element, peptide table) pairs of an mzIdentML file.  <Peptide> elements come first in the
    document (SequenceCollection), so the table is complete when the first result arrives."""
    peptides = {}
    for elem in _stream(path, {"Peptide", "SpectrumIdentificationResult"}):
        if _local(elem.tag) == "Peptide":
            keep = ET.Element(elem.tag, elem.attrib)          # (the stream clears what it has handed out)
            keep.extend(list(elem))
            peptides[elem.get("id")] = keep
        else:
            yield elem, peptides


class IdentificationParser:
    """PSMs of one identification file (id_parsers.py:656-815).

    ``id_file_format``: "pepXML", "mzIdentML", "percolatorTXT" or "mokapotTXT"; ``score_string`` the
    name of the score to report (pepXML search_score / mzIdentML cvParam; the two table formats have
    their own); ``score_threshold``: PSMs are kept when ``score < score_threshold`` -- the reference
    computes ``(score - threshold) * (-1 ** score_lower_better) > 0``, in which the power binds before
    the minus sign, so ``score_lower_better`` has no effect there either and it is kept only as an
    argument; ``score_func`` is applied to the score first; ``static_mods`` (residue -> mass) are added
    to unmodified residues by the two table formats."""

    def __init__(self, id_file_name, id_file_format, mass_corrector=None, score_string=None,
                 score_threshold=None, score_lower_better=True, score_func=None, static_mods=None,
                 spec_file_name=None):
        static_mods = {"C": 57.021464} if static_mods is None else static_mods
        if id_file_format == "mzIdentML":
            ex = MzIdentMLExtractor(score_string)
            self._entries = lambda: (ex.extract(e) for e in _mzid_entries(id_file_name))
        elif id_file_format == "pepXML":
            ex = PepXMLExtractor(score_string)
            self._entries = lambda: (ex.extract(q) for q in _stream(id_file_name, {"spectrum_query"}))
        elif id_file_format == "percolatorTXT":
            ex = PercolatorTXTExtractor(score_string, static_mods)
            self._entries = lambda: (ex.extract(g) for g in _table_groups(id_file_name, "scan"))
        elif id_file_format == "mokapotTXT":
            ex = MokapotTXTExtractor(score_string, static_mods)
            self._entries = lambda: (ex.extract(g) for g in _table_groups(id_file_name, "ScanNr"))
        else:
            raise ValueError("{} not supported at this time."
                             " Must be on of: mzIdentML, pepXML,"
                             " percolatorTXT, or mokapotTXT".format(id_file_format))
        self.mass_corrector = MassCorrector() if mass_corrector is None else mass_corrector
        self.score_threshold = score_threshold
        self.score_lower_better = score_lower_better
        self.score_func = score_func
        self.spec_file_name = spec_file_name
        self._match_records = []

    def _load(self):
        if not self._match_records:
            recs = [e for e in self._entries() if len(e["peptides"]) > 0]
            self._match_records = sorted(recs, key=lambda e: e["scans"][0])

    def _passes(self, score):
        if self.score_threshold is None:
            return True
        if score is None:
            return False
        return (score - self.score_threshold) * -1 > 0

    def _hits(self):
        self._load()
        for rec in self._match_records:
            for i in range(len(rec["peptides"])):
                pos, mass = self.mass_corrector.correct_multiple(rec["peptides"][i], rec["mod_positions"][i],
                                                                 rec["mod_masses"][i])
                score = rec["scores"][i]
                if self.score_func is not None and score is not None:
                    score = self.score_func(score)
                if not self._passes(score):
                    continue
                yield {"scan": rec["scans"][i], "charge_state": rec["charge_states"][i], "score": score,
                       "peptide": rec["peptides"][i], "mod_positions": pos, "mod_masses": mass}

    def to_list(self):
        return list(self._hits())

    def to_dict(self):
        return {hit.pop("scan"): hit for hit in self._hits()}


# ------------------------------------------------------------------------------------------------
# straight into one batch
# ------------------------------------------------------------------------------------------------
def to_batch(psms, spectra_map, residues, mod_mass, hit_depth=1, max_fragment_charge=5,
             mod_correction_tol=1.0):
    """Parsed identifications + spectra -> (CSR batch for ``PyAscore.score_batch``, the scan of each of
    its PSMs), with the reference CLI's per-PSM decisions (`__main__.py:127-164`: hit depth, variable /
    fixed split, charge heuristic).  PSMs without an unlocalised modification are left out, as there."""
    from .batch_cli import select_psms
    from .synth import pack_batch
    picked, scans = select_psms(sorted(psms, key=lambda p: p["scan"]), spectra_map, residues, mod_mass, hit_depth,
                                max_fragment_charge, mod_correction_tol)
    return (pack_batch(picked) if picked else None), scans

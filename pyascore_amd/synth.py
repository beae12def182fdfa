"""Synthetic PSM batches for the BASELINE.json configs.

A *batch* is the CSR layout the C ABI (include/pyascore_hip.h) takes:

    mz, intensity : float64[total_peaks]   raw peaks of all spectra, back to back
    peak_off      : int64[n_psm + 1]
    pep           : uint8[total_residues]  peptide letters, back to back
    pep_off       : int64[n_psm + 1]
    n_of_mod      : int32[n_psm]           unlocalised mods per PSM
    max_charge    : int32[n_psm]           max fragment charge per PSM
    aux_pos       : uint32[total_aux]      fixed-mod positions (0 = n-term, else 1-based)
    aux_mass      : float32[total_aux]
    aux_off       : int64[n_psm + 1]

Residue masses are the reference's table (Types.h:7-30) in float64, as section 8(d) asks.
"""
import numpy as np

# Types.h:7-30
RESIDUE_MASS = {
    "G": 57.02146, "A": 71.03711, "S": 87.03203, "P": 97.05276, "V": 99.06841,
    "T": 101.04768, "C": 103.00919, "L": 113.08406, "I": 113.08406, "N": 114.04293,
    "D": 115.02694, "Q": 128.05858, "K": 128.09496, "E": 129.04259, "M": 131.04049,
    "H": 137.05891, "F": 147.06841, "U": 150.95364, "R": 156.10111, "Y": 163.06333,
    "W": 186.07931, "O": 237.14773,
}
BASE_ALPHABET = "ACDEFGHIKLMNPQRVW"
PHOSPHO = 79.966331
PROTON = 1.007825
WATER = 18.010565

_MASS_LUT = np.zeros(256)
for _k, _v in RESIDUE_MASS.items():
    _MASS_LUT[ord(_k)] = _v

CONFIGS = {
    # name: generator shape + scorer settings
    "cfg1": dict(n_psm=1, L=12, n_sites=2, n_mod=1, mz_error=0.5, fragment_types="by",
                 max_charge=1, neutral_loss=None),
    "cfg2": dict(n_psm=100_000, L=20, n_sites=6, n_mod=3, mz_error=0.05, fragment_types="by",
                 max_charge=1, neutral_loss=None),
    "cfg3": dict(n_psm=1_000_000, L=None, n_sites=None, n_mod=None, mz_error=0.05,
                 fragment_types="by", max_charge=1, neutral_loss=None),
    "cfg4": dict(n_psm=250_000, L=20, n_sites=6, n_mod=3, mz_error=0.02, fragment_types="bycz",
                 max_charge=4, neutral_loss=("sty", 97.9769)),
    "cfg5": dict(n_psm=50_000, L=30, n_sites=15, n_mod=5, mz_error=0.05, fragment_types="by",
                 max_charge=1, neutral_loss=None),
}


def _fixed_shape(rng, n, L, n_sites, n_mod, mz_error, n_noise=300, keep_p=0.6):
    """n PSMs of one (L, n_sites, n_mod) shape -> (pep uint8[n,L], mz list, inten list, counts)."""
    base = np.frombuffer(BASE_ALPHABET.encode(), dtype=np.uint8)
    pep = base[rng.integers(0, len(base), size=(n, L))]
    sites = np.sort(np.argsort(rng.random((n, L)), axis=1)[:, :n_sites], axis=1)
    sty = np.frombuffer(b"STY", dtype=np.uint8)[rng.choice(3, size=(n, n_sites), p=(0.5, 0.35, 0.15))]
    rows = np.arange(n)[:, None]
    pep[rows, sites] = sty
    truth = np.argsort(rng.random((n, n_sites)), axis=1)[:, :n_mod]
    mass = _MASS_LUT[pep]
    np.add.at(mass, (np.repeat(np.arange(n), n_mod), sites[rows, truth].ravel()), PHOSPHO)
    fwd = np.cumsum(mass, axis=1)[:, : L - 1]
    rev = np.cumsum(mass[:, ::-1], axis=1)[:, : L - 1]
    sig = np.concatenate([fwd + PROTON, rev + WATER + PROTON], axis=1)
    keep = rng.random(sig.shape) < keep_p
    sig = sig + rng.uniform(-0.4 * mz_error, 0.4 * mz_error, size=sig.shape)
    sig_int = rng.lognormal(6.0, 1.2, size=sig.shape)
    noise = rng.uniform(100.0, 2000.0, size=(n, n_noise))
    noise_int = rng.lognormal(4.5, 1.0, size=(n, n_noise))
    mz = np.concatenate([np.where(keep, sig, np.inf), noise], axis=1)
    inten = np.concatenate([sig_int, noise_int], axis=1)
    order = np.argsort(mz, axis=1, kind="stable")
    mz = np.take_along_axis(mz, order, axis=1)
    inten = np.take_along_axis(inten, order, axis=1)
    counts = keep.sum(axis=1) + n_noise
    valid = np.arange(mz.shape[1])[None, :] < counts[:, None]
    return pep, mz[valid], inten[valid], counts.astype(np.int64)


def make_batch(config="cfg2", n_psm=None, seed=0, **override):
    """Synthetic batch of one BASELINE config (``n_psm`` overrides the config's size)."""
    cfg = dict(CONFIGS[config])
    cfg.update(override)
    n = int(n_psm if n_psm is not None else cfg["n_psm"])
    rng = np.random.default_rng(seed)
    err = cfg["mz_error"]
    if cfg["L"] is not None:
        pep, mz, inten, counts = _fixed_shape(rng, n, cfg["L"], cfg["n_sites"], cfg["n_mod"], err)
        pep_len = np.full(n, cfg["L"], np.int64)
        pep_flat = pep.ravel()
        n_of_mod = np.full(n, cfg["n_mod"], np.int32)
    else:
        # cfg3: L ~ U{8..40}, n_mod ~ U{1..4}, n_sites ~ U{n_mod+1 .. min(12, L-1)}
        Ls = rng.integers(8, 41, size=n)
        ks = rng.integers(1, 5, size=n)
        hi = np.minimum(12, Ls - 1)
        ns = ks + 1 + (rng.random(n) * (hi - ks)).astype(np.int64)
        ns = np.minimum(ns, hi)
        key = (Ls * 64 + ns) * 8 + ks
        order = np.argsort(key, kind="stable")
        uniq, starts = np.unique(key[order], return_index=True)
        ends = np.append(starts[1:], n)
        # one vectorised generator call per (L, n_sites, n_mod) shape, then scattered back into input
        # order with index arithmetic (no per-PSM Python work: 1M PSMs in seconds)
        groups, cnts = [], np.zeros(n, np.int64)
        for s, e in zip(starts, ends):
            idx = order[s:e]
            L, nsit, k = int(Ls[idx[0]]), int(ns[idx[0]]), int(ks[idx[0]])
            p, m, it, c = _fixed_shape(rng, e - s, L, nsit, k, err)
            cnts[idx] = c
            groups.append((idx, p, m, it, c))
        peak_off = np.concatenate([[0], np.cumsum(cnts)])
        pep_off = np.concatenate([[0], np.cumsum(Ls)])
        mz = np.empty(peak_off[-1], np.float64)
        inten = np.empty(peak_off[-1], np.float64)
        pep_flat = np.empty(pep_off[-1], np.uint8)
        for idx, p, m, it, c in groups:
            src_start = np.concatenate([[0], np.cumsum(c)[:-1]])
            dest = np.repeat(peak_off[idx] - src_start, c) + np.arange(m.size)
            mz[dest] = m
            inten[dest] = it
            pep_flat[(pep_off[idx][:, None] + np.arange(p.shape[1])[None, :]).ravel()] = p.ravel()
        del groups
        counts = cnts
        pep_len = Ls.astype(np.int64)
        n_of_mod = ks.astype(np.int32)
    batch = dict(
        n_psm=n,
        mz=np.ascontiguousarray(mz, dtype=np.float64),
        intensity=np.ascontiguousarray(inten, dtype=np.float64),
        peak_off=np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
        pep=np.ascontiguousarray(pep_flat, dtype=np.uint8),
        pep_off=np.concatenate([[0], np.cumsum(pep_len)]).astype(np.int64),
        n_of_mod=n_of_mod,
        max_charge=np.full(n, cfg["max_charge"], np.int32),
        aux_pos=np.zeros(0, np.uint32),
        aux_mass=np.zeros(0, np.float32),
        aux_off=np.zeros(n + 1, np.int64),
    )
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=PHOSPHO, mz_error=err,
                    fragment_types=cfg["fragment_types"],
                    neutral_losses=[list(cfg["neutral_loss"])] if cfg["neutral_loss"] else [])
    return batch, settings


# ---------------------------------------------------------------------------------------------
# Batch *descriptions* and slices: what the multi-GPU path needs.  A description holds only the
# per-PSM shape (L, n_sites, n_mod, max_charge) of a whole job -- a few bytes per PSM, so every rank
# can hold it and cut the same work-balanced partition -- and spectra are generated per fixed block
# of BLOCK PSMs from a seed that depends only on (seed, block).  A rank generates just the blocks
# its slice touches, and the job's data do not depend on the number of ranks.
# ---------------------------------------------------------------------------------------------
BLOCK = 16384


def describe(config="cfg2", n_psm=None, seed=0, n_noise=300, isotopes=False, **override):
    """Per-PSM shapes of a synthetic job of one BASELINE config (no spectra).  ``n_noise`` / ``isotopes``:
    denser spectra than section 8(d)'s 300 noise peaks (bench.py's `dense` legs), see _var_shape."""
    cfg = dict(CONFIGS[config])
    cfg.update(override)
    n = int(n_psm if n_psm is not None else cfg["n_psm"])
    if cfg["L"] is not None:
        Ls = np.full(n, cfg["L"], np.int64)
        ns = np.full(n, cfg["n_sites"], np.int64)
        ks = np.full(n, cfg["n_mod"], np.int64)
    else:
        rng = np.random.default_rng([int(seed), 0xD35C])
        Ls = rng.integers(8, 41, size=n)
        ks = rng.integers(1, 5, size=n)
        hi = np.minimum(12, Ls - 1)
        ns = np.minimum(ks + 1 + (rng.random(n) * (hi - ks)).astype(np.int64), hi)
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=PHOSPHO, mz_error=cfg["mz_error"],
                    fragment_types=cfg["fragment_types"],
                    neutral_losses=[list(cfg["neutral_loss"])] if cfg["neutral_loss"] else [])
    return dict(config=config, n_psm=n, seed=int(seed), L=Ls, n_sites=ns, n_mod=ks,
                max_charge=np.full(n, cfg["max_charge"], np.int64), settings=settings,
                n_noise=int(n_noise), isotopes=bool(isotopes))


ISOTOPE_STEP = 1.00335            # 13C - 12C


def _var_shape(rng, Ls, ns, ks, mz_error, n_noise=300, keep_p=0.6, isotopes=False):
    """The generator of _fixed_shape for PSMs of mixed shapes in one vectorised pass (rows padded to
    the longest peptide and masked): -> (pep uint8 flat, mz flat, intensity flat, peak counts).
    ``isotopes``: every peak (fragment or noise) brings two satellites at +1.00335/z and +2.0067/z with
    0.5x / 0.2x its intensity (z = 1 for the fragments, 1..3 for the noise): ``n_noise`` then counts the
    noise peaks WITH their satellites (a third of them are drawn)."""
    n, Lmax = Ls.size, int(Ls.max())
    col = np.arange(Lmax)[None, :]
    inside = col < Ls[:, None]
    base = np.frombuffer(BASE_ALPHABET.encode(), dtype=np.uint8)
    pep = base[rng.integers(0, len(base), size=(n, Lmax))]
    key = rng.random((n, Lmax))
    key[~inside] = 2.0
    is_site = np.argsort(np.argsort(key, axis=1), axis=1) < ns[:, None]
    sty = np.frombuffer(b"STY", dtype=np.uint8)[rng.choice(3, size=(n, Lmax), p=(0.5, 0.35, 0.15))]
    pep = np.where(is_site, sty, pep)
    key = rng.random((n, Lmax))
    key[~is_site] = 2.0
    is_mod = np.argsort(np.argsort(key, axis=1), axis=1) < ks[:, None]
    mass = np.where(inside, _MASS_LUT[pep] + PHOSPHO * is_mod, 0.0)
    rev_idx = np.clip(Ls[:, None] - 1 - col, 0, Lmax - 1)
    mass_rev = np.where(inside, np.take_along_axis(mass, rev_idx, axis=1), 0.0)
    frag_ok = (col < (Ls[:, None] - 1))[:, : Lmax - 1]
    fwd = np.cumsum(mass, axis=1)[:, : Lmax - 1]
    rev = np.cumsum(mass_rev, axis=1)[:, : Lmax - 1]
    sig = np.concatenate([fwd + PROTON, rev + WATER + PROTON], axis=1)
    keep = (rng.random(sig.shape) < keep_p) & np.concatenate([frag_ok, frag_ok], axis=1)
    sig = sig + rng.uniform(-0.4 * mz_error, 0.4 * mz_error, size=sig.shape)
    sig_int = rng.lognormal(6.0, 1.2, size=sig.shape)
    if isotopes:
        n_base = max(1, n_noise // 3)
        noise = rng.uniform(100.0, 2000.0, size=(n, n_base))
        noise_int = rng.lognormal(4.5, 1.0, size=(n, n_base))
        step = ISOTOPE_STEP / rng.integers(1, 4, size=(n, n_base))
        sig = np.where(keep, sig, np.inf)
        mz = np.concatenate([sig, sig + ISOTOPE_STEP, sig + 2 * ISOTOPE_STEP, noise, noise + step, noise + 2 * step], axis=1)
        inten = np.concatenate([sig_int, 0.5 * sig_int, 0.2 * sig_int, noise_int, 0.5 * noise_int, 0.2 * noise_int], axis=1)
        counts = (3 * keep.sum(axis=1) + 3 * n_base).astype(np.int64)
    else:
        noise = rng.uniform(100.0, 2000.0, size=(n, n_noise))
        noise_int = rng.lognormal(4.5, 1.0, size=(n, n_noise))
        mz = np.concatenate([np.where(keep, sig, np.inf), noise], axis=1)
        inten = np.concatenate([sig_int, noise_int], axis=1)
        counts = (keep.sum(axis=1) + n_noise).astype(np.int64)
    order = np.argsort(mz, axis=1, kind="stable")
    mz = np.take_along_axis(mz, order, axis=1)
    inten = np.take_along_axis(inten, order, axis=1)
    valid = np.arange(mz.shape[1])[None, :] < counts[:, None]
    return pep[inside], mz[valid], inten[valid], counts


def _make_block(desc, b):
    """PSMs [b * BLOCK, min((b + 1) * BLOCK, n)) of a described job."""
    i0, i1 = b * BLOCK, min((b + 1) * BLOCK, desc["n_psm"])
    n = i1 - i0
    rng = np.random.default_rng([desc["seed"], 0xB10C, b])
    Ls, ns, ks = desc["L"][i0:i1], desc["n_sites"][i0:i1], desc["n_mod"][i0:i1]
    pep, mz, inten, counts = _var_shape(rng, Ls, ns, ks, desc["settings"]["mz_error"], n_noise=desc.get("n_noise", 300),
                                        isotopes=desc.get("isotopes", False))
    return dict(n_psm=n, mz=mz, intensity=inten, peak_off=np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
                pep=pep, pep_off=np.concatenate([[0], np.cumsum(Ls)]).astype(np.int64),
                n_of_mod=ks.astype(np.int32), max_charge=desc["max_charge"][i0:i1].astype(np.int32),
                aux_pos=np.zeros(0, np.uint32), aux_mass=np.zeros(0, np.float32), aux_off=np.zeros(n + 1, np.int64))


def concat_batches(parts):
    """CSR batches back to back."""
    out = dict(n_psm=int(sum(p["n_psm"] for p in parts)))
    for key in ("mz", "intensity", "pep", "n_of_mod", "max_charge", "aux_pos", "aux_mass"):
        out[key] = np.concatenate([p[key] for p in parts]) if parts else np.zeros(0)
    for key, data in (("peak_off", "mz"), ("pep_off", "pep"), ("aux_off", "aux_pos")):
        offs, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            offs.append(np.asarray(p[key][1:], np.int64) - int(p[key][0]) + base)
            base += int(p[key][-1]) - int(p[key][0])
        out[key] = np.concatenate(offs)
    return out


def make_slice(desc, lo=0, hi=None, threads=None):
    """PSMs [lo, hi) of a described job as a CSR batch (blocks generated on a thread pool: numpy's
    sorts and generators release the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    hi = desc["n_psm"] if hi is None else int(hi)
    lo = int(lo)
    if not 0 <= lo <= hi <= desc["n_psm"]:
        raise ValueError("slice outside the job")
    if hi == lo:
        return concat_batches([])
    blocks = list(range(lo // BLOCK, (hi - 1) // BLOCK + 1))
    nt = max(1, min(len(blocks), threads or min(8, os.cpu_count() or 1)))

    def one(b):
        part = _make_block(desc, b)
        b0 = b * BLOCK
        a, z = max(lo, b0) - b0, min(hi, b0 + part["n_psm"]) - b0
        return part if (a == 0 and z == part["n_psm"]) else slice_batch(part, a, z)

    if nt == 1:
        parts = [one(b) for b in blocks]
    else:
        with ThreadPoolExecutor(nt) as ex:
            parts = list(ex.map(one, blocks))
    return concat_batches(parts)


def pack_batch(psms):
    """List of dicts {mz, intensity, peptide, n_of_mod, max_charge, aux_pos, aux_mass} -> CSR."""
    n = len(psms)
    mz = [np.asarray(p["mz"], np.float64) for p in psms]
    it = [np.asarray(p["intensity"], np.float64) for p in psms]
    pep = [np.frombuffer(p["peptide"].encode(), dtype=np.uint8) for p in psms]
    ap = [np.asarray(p.get("aux_pos", ()), np.uint32) for p in psms]
    am = [np.asarray(p.get("aux_mass", ()), np.float32) for p in psms]

    def cat(xs, dt):
        return np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0), dtype=dt)

    def off(xs):
        return np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)

    return dict(
        n_psm=n, mz=cat(mz, np.float64), intensity=cat(it, np.float64), peak_off=off(mz),
        pep=cat(pep, np.uint8), pep_off=off(pep),
        n_of_mod=np.asarray([p["n_of_mod"] for p in psms], np.int32),
        max_charge=np.asarray([p.get("max_charge", 1) for p in psms], np.int32),
        aux_pos=cat(ap, np.uint32), aux_mass=cat(am, np.float32), aux_off=off(ap),
    )


def unpack_psm(batch, i):
    """PSM ``i`` of a CSR batch as the keyword arguments of ``PyAscore.score``."""
    a, b = batch["peak_off"][i], batch["peak_off"][i + 1]
    c, d = batch["pep_off"][i], batch["pep_off"][i + 1]
    e, f = batch["aux_off"][i], batch["aux_off"][i + 1]
    out = dict(mz_arr=batch["mz"][a:b], int_arr=batch["intensity"][a:b],
               peptide=bytes(batch["pep"][c:d]).decode(), n_of_mod=int(batch["n_of_mod"][i]),
               max_fragment_charge=int(batch["max_charge"][i]))
    if f > e:
        out["aux_mod_pos"] = batch["aux_pos"][e:f]
        out["aux_mod_mass"] = batch["aux_mass"][e:f]
    return out


def slice_batch(batch, lo, hi):
    """Contiguous PSM range [lo, hi) of a CSR batch (used for sharding across ranks)."""
    a, b = batch["peak_off"][lo], batch["peak_off"][hi]
    c, d = batch["pep_off"][lo], batch["pep_off"][hi]
    e, f = batch["aux_off"][lo], batch["aux_off"][hi]
    return dict(
        n_psm=hi - lo, mz=batch["mz"][a:b], intensity=batch["intensity"][a:b],
        peak_off=(batch["peak_off"][lo:hi + 1] - a), pep=batch["pep"][c:d],
        pep_off=(batch["pep_off"][lo:hi + 1] - c), n_of_mod=batch["n_of_mod"][lo:hi],
        max_charge=batch["max_charge"][lo:hi], aux_pos=batch["aux_pos"][e:f],
        aux_mass=batch["aux_mass"][e:f], aux_off=(batch["aux_off"][lo:hi + 1] - e),
    )


# ---------------------------------------------------------------------------------------------
# Spectra that look like an instrument's (r06): isotope envelopes, doublets inside the tolerance, repeated
# m/z, count-like intensities, 800-3000 peaks -- the data on which the rare branches of the count-node
# marking (walk_core.hip.h) and of the site-ion closed forms (localize_hash.hip.h) are NOT rare.
# ---------------------------------------------------------------------------------------------
ACETYL = 42.010565
OXIDATION = 15.994915
H3PO4 = 97.9769
H2O = 18.01528


def make_realistic(n_psm, seed=0, general=True, mz_error=0.02, fragment_types=None, max_sites=10, max_mod=4,
                   peaks=(800, 3000)):
    """PSMs shaped like test/test_ascore.py:10-61's data (a phosphopeptide, fixed oxidation / n-terminal acetyl as
    auxiliary mods) with spectra that cluster the way centroided MS2 spectra do.

    general=True : phospho on STY, neutral losses ("sty", H3PO4) and ("ST", H2O), fragment charges 1..4,
                   `fragment_types` "bycz" -- the settings of cfg4's kernels;
    general=False: no losses, charge 1, "by" -- the plain kernels (fused / count nodes / big).
    Per PSM: L 8..32, 2..max_sites STY residues, 1..max_mod of them modified; M carries oxidation and the n-terminus
    acetyl on a share of the peptides (aux mods, never on a modifiable residue: the reference reads out of bounds there).
    Signal: b / y (+ c / z when asked) fragments of the true assignment at every charge, loss variants on a third, each kept
    with p = 0.6, with an isotope envelope (+1.00335/z, +2.0067/z at 0.5x / 0.2x); a tenth of the signal peaks get a doublet
    partner 0.2..1.5 mz_error away, 2 % of all peaks an exact repeat of their m/z.  Noise: envelopes at charge 1..3 up to
    the target peak count.  Intensities are integer counts (ties everywhere among the weak peaks)."""
    rng = np.random.default_rng([int(seed), 0x4EA1])
    ftypes = fragment_types or ("bycz" if general else "by")
    base = list(BASE_ALPHABET)
    psms = []
    for _ in range(int(n_psm)):
        L = int(rng.integers(8, 33))
        n_sites = int(rng.integers(2, min(max_sites, L - 1) + 1))
        k = int(rng.integers(1, min(max_mod, n_sites - 1) + 1))
        pep = [base[i] for i in rng.integers(0, len(base), size=L)]
        site_pos = np.sort(rng.choice(L, size=n_sites, replace=False))
        for p in site_pos:
            pep[p] = "STY"[int(rng.choice(3, p=(0.5, 0.35, 0.15)))]
        truth = set(int(p) for p in rng.choice(site_pos, size=k, replace=False))
        mass = np.array([RESIDUE_MASS[c] for c in pep])
        aux_pos, aux_mass = [], []
        for i, c in enumerate(pep):
            if c == "M" and rng.random() < 0.7:
                aux_pos.append(i + 1)
                aux_mass.append(OXIDATION)
                mass[i] += OXIDATION
        if rng.random() < 0.3 and 0 not in site_pos:
            aux_pos.append(0)
            aux_mass.append(ACETYL)
            mass[0] += ACETYL
        for p in truth:
            mass[p] += PHOSPHO
        zmax = int(rng.integers(1, 5)) if general else 1
        is_p = np.array([i in truth for i in range(L)])
        is_st = np.array([(c in "ST") and (i not in truth) for i, c in enumerate(pep)])
        sig_mz, sig_z = [], []
        for direction in (0, 1):
            order = np.arange(L) if direction == 0 else np.arange(L)[::-1]
            m1 = np.cumsum(mass[order])[:-1]
            has_p = np.cumsum(is_p[order])[:-1] > 0
            has_st = np.cumsum(is_st[order])[:-1] > 0
            offs = []
            if direction == 0:
                if "b" in ftypes:
                    offs.append(0.0)
                if "c" in ftypes:
                    offs.append(17.026549)
            else:
                if "y" in ftypes:
                    offs.append(WATER)
                if "z" in ftypes or "Z" in ftypes:
                    offs.append(WATER - 17.026549 + (1.007825 if "Z" in ftypes else 0.0))
            for off in offs:
                neutral = m1 + off
                variants = [neutral]
                if general:
                    variants.append(np.where(has_p & (rng.random(m1.size) < 0.35), neutral - H3PO4, np.nan))
                    variants.append(np.where(has_st & (rng.random(m1.size) < 0.2), neutral - H2O, np.nan))
                for v in variants:
                    for z in range(1, zmax + 1):
                        x = (v + z * PROTON) / z
                        keep = np.isfinite(x) & (rng.random(x.size) < (0.6 if z == 1 else 0.3))
                        sig_mz.append(x[keep])
                        sig_z.append(np.full(int(keep.sum()), z))
        s_mz = np.concatenate(sig_mz) if sig_mz else np.zeros(0)
        s_z = np.concatenate(sig_z) if sig_z else np.zeros(0, np.int64)
        s_mz = s_mz + rng.uniform(-0.4 * mz_error, 0.4 * mz_error, size=s_mz.size)
        s_int = rng.lognormal(6.0, 1.2, size=s_mz.size)
        # doublets: a partner 0.2 .. 1.5 tolerances away from a tenth of the signal peaks
        dbl = rng.random(s_mz.size) < 0.1
        d_mz = s_mz[dbl] + rng.choice([-1.0, 1.0], size=int(dbl.sum())) * rng.uniform(0.2, 1.5, size=int(dbl.sum())) * mz_error
        d_int = rng.lognormal(5.5, 1.2, size=d_mz.size)
        target = int(rng.integers(peaks[0], peaks[1] + 1))
        n_noise = max(0, (target - 3 * (s_mz.size + d_mz.size)) // 3)
        n_mz = rng.uniform(100.0, 2200.0, size=n_noise)
        n_z = rng.integers(1, 4, size=n_noise)
        n_int = rng.lognormal(4.0, 1.0, size=n_noise)
        b_mz = np.concatenate([s_mz, d_mz, n_mz])
        b_z = np.concatenate([s_z, np.ones(d_mz.size, np.int64), n_z]).astype(np.float64)
        b_int = np.concatenate([s_int, d_int, n_int])
        mz = np.concatenate([b_mz, b_mz + ISOTOPE_STEP / b_z, b_mz + 2 * ISOTOPE_STEP / b_z])
        it = np.concatenate([b_int, 0.5 * b_int, 0.2 * b_int])
        rep = rng.random(mz.size) < 0.02                        # the same m/z twice
        mz = np.concatenate([mz, mz[rep]])
        it = np.concatenate([it, rng.lognormal(4.5, 1.0, size=int(rep.sum()))])
        it = np.floor(it / 8.0) + 1.0                           # counts
        order = np.argsort(mz, kind="stable")
        psms.append(dict(mz=mz[order], intensity=it[order], peptide="".join(pep), n_of_mod=k, max_charge=zmax,
                         aux_pos=np.asarray(aux_pos, np.uint32), aux_mass=np.asarray(aux_mass, np.float32)))
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=PHOSPHO, mz_error=float(mz_error),
                    fragment_types=ftypes,
                    neutral_losses=[["sty", H3PO4], ["ST", H2O]] if general else [])
    return pack_batch(psms), settings

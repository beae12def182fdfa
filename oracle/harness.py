"""Parity harness -- TEST INFRASTRUCTURE ONLY.

``collect(scorer, batch)`` drives any object with the reference's ``PyAscore`` surface
(oracle.orc.OracleAscore or pyascore_amd.PyAscore) over a CSR batch through the *public* API
(score() + properties) and flattens every result into numpy arrays so that two scorers -- or a
scorer and a committed golden file -- can be compared field by field.
"""
import json
import numpy as np


def sig_bits(sig):
    b = 0
    for j, s in enumerate(sig):
        if int(s):
            b |= 1 << j
    return b


def collect(scorer, batch, unpack):
    n = batch["n_psm"]
    max_k = max(1, int(batch["n_of_mod"].max()) if n else 1)
    out = dict(
        n_sig=np.zeros(n, np.int32), best_score=np.zeros(n, np.float32),
        best_sig=np.zeros(n, np.uint64), n_ascores=np.zeros(n, np.int32),
        ascores=np.zeros((n, max_k), np.float32), alt_mask=np.zeros((n, max_k), np.uint64),
        sig_len=np.zeros(n, np.int32),
    )
    seqs, bits, counts, scores, ws, nfrag, off = [], [], [], [], [], [], [0]
    for i in range(n):
        scorer.score(**unpack(batch, i))
        ps = scorer.pep_scores
        out["n_sig"][i] = len(ps)
        out["best_score"][i] = scorer.best_score
        seqs.append(scorer.best_sequence)
        a = np.asarray(scorer.ascores, np.float32)
        out["n_ascores"][i] = a.size
        out["ascores"][i, : a.size] = a
        for j, alt in enumerate(scorer.alt_sites):
            m = 0
            for s in alt:
                m |= 1 << (int(s) - 1)
            out["alt_mask"][i, j] = m
        if ps:
            out["best_sig"][i] = sig_bits(ps[0]["signature"])
            out["sig_len"][i] = len(ps[0]["signature"])
        for p in ps:
            bits.append(sig_bits(p["signature"]))
            counts.append(np.asarray(p["counts"], np.int32))
            scores.append(np.asarray(p["scores"], np.float32))
            ws.append(p["weighted_score"])
            nfrag.append(p["total_fragments"])
        off.append(len(bits))
    out["best_sequence"] = np.asarray(seqs, dtype=np.str_)
    out["ps_off"] = np.asarray(off, np.int64)
    out["ps_bits"] = np.asarray(bits, np.uint64)
    out["ps_counts"] = np.asarray(counts, np.int32).reshape(-1, 10) if counts else np.zeros((0, 10), np.int32)
    out["ps_scores"] = np.asarray(scores, np.float32).reshape(-1, 10) if scores else np.zeros((0, 10), np.float32)
    out["ps_ws"] = np.asarray(ws, np.float32)
    out["ps_nfrag"] = np.asarray(nfrag, np.int64)
    return out


INT_FIELDS = ("n_sig", "best_sig", "n_ascores", "alt_mask", "sig_len", "ps_off", "ps_bits",
              "ps_counts", "ps_nfrag")
FLOAT_FIELDS = ("best_score", "ascores", "ps_scores", "ps_ws")


def compare(got, want, exact_float=True, rtol=1e-6, atol=1e-6):
    """Returns a list of human-readable mismatch strings (empty = parity)."""
    bad = []
    for k in INT_FIELDS:
        if not np.array_equal(got[k], want[k]):
            bad.append("%s differs (%d entries)" % (k, int(np.sum(np.asarray(got[k]) != np.asarray(want[k]))) if np.shape(got[k]) == np.shape(want[k]) else -1))
    if not np.array_equal(got["best_sequence"], want["best_sequence"]):
        bad.append("best_sequence differs")
    for k in FLOAT_FIELDS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        if g.shape != w.shape:
            bad.append("%s shape %s vs %s" % (k, g.shape, w.shape))
        elif exact_float:
            if not np.array_equal(g, w):
                bad.append("%s not bit-equal (max abs diff %g)" % (k, float(np.nanmax(np.abs(np.where(np.isfinite(g) & np.isfinite(w), g - w, 0))))))
        else:
            fin = np.isfinite(w)
            if not (np.array_equal(fin, np.isfinite(g)) and np.array_equal(g[~fin], w[~fin])
                    and np.allclose(g[fin], w[fin], rtol=rtol, atol=atol)):
                bad.append("%s outside tolerance" % k)
    return bad


def make_scorer(factory, settings, **kw):
    """factory(bin_size, n_top, mod_group, mod_mass, mz_error, fragment_types, **kw)."""
    s = factory(settings["bin_size"], settings["n_top"], settings["mod_group"],
                settings["mod_mass"], settings["mz_error"], settings["fragment_types"], **kw)
    for group, mass in settings.get("neutral_losses", []):
        s.add_neutral_loss(group, mass)
    return s


def save_case(path, settings, batch, expected):
    arrays = {"settings": np.asarray(json.dumps(settings))}
    for k, v in batch.items():
        arrays["in_" + k] = np.asarray(v)
    for k, v in expected.items():
        arrays["exp_" + k] = v
    np.savez_compressed(path, **arrays)


def load_case(path):
    z = np.load(path, allow_pickle=False)
    settings = json.loads(str(z["settings"]))
    batch = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    batch["n_psm"] = int(batch["n_psm"])
    expected = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return settings, batch, expected

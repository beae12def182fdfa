"""Random scorer settings x random small batches shared by the CPU and GPU fuzz tests."""
import numpy as np

from pyascore_amd import synth

AA = "ACDEFGHIKLMNPQRSTVWY"


def random_case(rng):
    mod_group = "".join(rng.choice(list("STYKMC"), size=rng.integers(1, 4), replace=False))
    if rng.random() < 0.3:
        mod_group += rng.choice(["n", "c", "nc"])
    mod_mass = float(rng.choice([79.966331, 42.010565, 15.994915, 14.01565, 114.042927]))
    ftypes = "".join(rng.choice(list("bycz") + ["Z"], size=rng.integers(1, 4), replace=False))
    err = float(rng.choice([0.01, 0.02, 0.05, 0.3, 0.5]))
    nls = []
    for _ in range(rng.integers(0, 3)):
        grp = "".join(rng.choice(list("STYstym"), size=rng.integers(1, 4), replace=False))
        nls.append([grp, float(rng.choice([18.01528, 97.9769, 63.998, 17.0265]))])
    settings = dict(bin_size=float(rng.choice([100.0, 100.0, 50.0, 150.0])), n_top=10, mod_group=mod_group,
                    mod_mass=mod_mass, mz_error=err, fragment_types=ftypes, neutral_losses=nls)
    psms = []
    for _ in range(int(rng.integers(8, 25))):
        L = int(rng.integers(2, 41))
        pep = "".join(rng.choice(list(AA), size=L))
        sites = [i for i, c in enumerate(pep) if c in mod_group or (i == 0 and "n" in mod_group)
                 or (i == L - 1 and "c" in mod_group)]
        n = len(sites)
        if n > 14 or n == 0:          # no modifiable residue: the reference dereferences end() (UB)
            continue
        k = int(rng.integers(0, min(n, 4) + 2))
        zmax = int(rng.integers(1, 4))
        truth = set(rng.choice(sites, size=min(k, n), replace=False)) if n and k else set()
        mass = np.array([synth.RESIDUE_MASS[c] for c in pep])
        for i in truth:
            mass[i] += mod_mass
        aux_pos, aux_mass = [], []
        if rng.random() < 0.4:
            for _ in range(rng.integers(1, 3)):
                pos = int(rng.integers(0, L + 1))
                idx = pos - 1 if pos > 0 else 0
                if pep[idx] in mod_group or (idx == 0 and "n" in mod_group) or (idx == L - 1 and "c" in mod_group):
                    continue          # the reference reads out of bounds for NL on such residues
                aux_pos.append(pos)
                aux_mass.append(float(rng.choice([15.9949, 57.021464, 42.010565])))
                mass[idx] += aux_mass[-1]
        b = np.cumsum(mass)[:-1] + synth.PROTON
        y = np.cumsum(mass[::-1])[:-1] + synth.WATER + synth.PROTON
        sig = np.concatenate([b, y, (b + synth.PROTON) / 2, (y + synth.PROTON) / 2])
        sig = sig[rng.random(sig.size) < 0.6] + rng.uniform(-0.4 * err, 0.4 * err, size=None)
        n_noise = int(rng.integers(0, 250))
        mz = np.concatenate([sig, rng.uniform(80.0, 2500.0, n_noise)])
        if mz.size == 0 or (mz.size == 1 and mz[0] % 100.0 == 0.0):
            mz = np.append(mz, 512.3)
        it = rng.lognormal(5.0, 1.3, mz.size)
        in_order = rng.random() < 0.8
        if rng.random() < 0.3:
            # count-like intensities: equal values inside the windows (the reference's choice among
            # them is std::nth_element's, on the window's peaks in input order)
            it = np.floor(it / np.median(it) * float(rng.choice([2.0, 10.0, 50.0]))) + 1.0
        order = np.argsort(mz) if in_order else rng.permutation(mz.size)
        psms.append(dict(mz=mz[order], intensity=it[order], peptide=pep, n_of_mod=k, max_charge=zmax,
                         aux_pos=np.asarray(aux_pos, np.uint32), aux_mass=np.asarray(aux_mass, np.float32)))
    return settings, synth.pack_batch(psms)



"""Known answers the reference's own unit tests hold for this path,
re-typed as literal peptide-chemistry facts and checked against the CPU restatement:
  test_modified_peptide_container.py:28-66 (signature order), :68-106 (b/c/y/z masses),
  :137-226 (all ion types, charges 0-3), :228-265 (terminal mods), :267-346 (aux mods),
  :348-392 (neutral losses), :394-512 (site-determining ions), :514-523 (get_peptide);
  test_spectra_container.py:15-67; test_util.py:11-96.
The oracle walks every signature from the start (no resume-from-common-node), so the
"from beginning" variants of the reference's expectations are the ones stated here."""
import numpy as np
import pytest
from scipy.special import binom as binom_coef
from scipy.special import logsumexp
from scipy.stats import binom as binom_dist

from oracle import orc

PHOSPHO = 79.966331


def pep(mod_group="STY", mod_mass=PHOSPHO, mz_error=0.5, ftypes="by", nl=()):
    s = orc.OracleAscore(100.0, 10, mod_group, mod_mass, mz_error, ftypes, kind="oracle")
    for g, m in nl:
        s.add_neutral_loss(g, m)
    return s


def check(scorer, ftype, charge, sig, neutral):
    want = (np.asarray(neutral) + charge * 1.007825) / max(1, charge)
    got = scorer.fragments(ftype, charge, sig)[0]
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-6, atol=0)


def test_signature_order():
    s = pep()
    s.consume_peptide("ASTK", 1)
    assert s.signature_order("b").tolist() == [[1, 0], [0, 1]]
    assert s.signature_order("y").tolist() == [[0, 1], [1, 0]]
    s.consume_peptide("PASSSSSEFK", 2)
    b, y = s.signature_order("b"), s.signature_order("y")
    assert len(b) == len(y) == 10
    assert b[0].tolist() == [1, 1, 0, 0, 0] and b[4].tolist() == [0, 1, 1, 0, 0]
    assert y[0].tolist() == [0, 0, 0, 1, 1] and y[4].tolist() == [0, 0, 1, 1, 0]
    s.consume_peptide("ASK", 1)
    assert len(s.signature_order("b")) == 1
    s.consume_peptide("PASSEFK", 2)
    assert len(s.signature_order("y")) == 1


def test_set_signature_masses():
    s = pep()
    s.consume_peptide("PASSSSSEFK", 2)
    sig = [0, 1, 1, 0, 0]
    # these literals already include the proton (charge 1)
    for ftype, masses in {
        "b": [98.06058, 169.09769, 256.12972, 423.12808, 590.12644, 677.15847, 764.19050,
              893.23309, 1040.30150],
        "c": [115.08713, 186.12424, 273.15627, 440.15463, 607.15299, 694.18502, 781.21705,
              910.25964, 1057.32805],
        "y": [147.11334, 294.18176, 423.22435, 510.25638, 597.28841, 764.28677, 931.28513,
              1018.3171, 1089.3542],
        "z": [130.08680, 277.15521, 406.19780, 493.22983, 580.26186, 747.26022, 914.25858,
              1001.29061, 1072.32772],
    }.items():
        got = s.fragments(ftype, 1, sig)[0]
        assert np.allclose(got, masses, rtol=1e-6, atol=0)


@pytest.mark.parametrize("charge", [0, 1, 2, 3])
def test_all_ion_types(charge):
    s = pep()
    s.consume_peptide("ASMTK", 1)
    first, second = [1, 0], [0, 1]
    check(s, "b", charge, first, [71.03711, 238.03547, 369.07596, 470.12364])
    check(s, "b", charge, second, [71.03711, 158.06914, 289.10963, 470.12364])
    check(s, "c", charge, first, [88.06365, 255.06201, 386.10251, 487.15019])
    check(s, "c", charge, second, [88.06365, 175.09568, 306.13617, 487.15019])
    # y-type iteration starts with the C-terminal site modified
    check(s, "y", charge, second, [146.10552, 327.11953, 458.160025, 545.19205])
    check(s, "y", charge, first, [146.105525, 247.15320, 378.19369, 545.192056])
    check(s, "z", charge, second, [129.07897, 310.09298, 441.13347, 528.16550])
    check(s, "z", charge, first, [129.07897, 230.12665, 361.16714, 528.16550])
    check(s, "Z", charge, second, [130.086795, 311.100805, 442.141295, 529.173325])
    check(s, "Z", charge, first, [130.086795, 231.134475, 362.174965, 529.173325])


@pytest.mark.parametrize("charge", [0, 1, 3])
def test_terminal_mod_group(charge):
    s = pep("nKc", 42.010565)
    s.consume_peptide("ASKTR", 1)
    assert s.signature_order("b").tolist() == [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    check(s, "b", charge, [1, 0, 0], [113.047675, 200.079705, 328.174664, 429.222344])
    check(s, "b", charge, [0, 1, 0], [71.03711, 158.06914, 328.174664, 429.222344])
    check(s, "b", charge, [0, 0, 1], [71.03711, 158.06914, 286.16409, 387.21178])
    check(s, "y", charge, [0, 0, 1], [216.12223, 317.16992, 445.26487, 532.29691])
    check(s, "y", charge, [0, 1, 0], [174.11167, 275.15935, 445.26487, 532.2969])
    check(s, "y", charge, [1, 0, 0], [174.11167, 275.15935, 403.254314, 490.286345])


@pytest.mark.parametrize("charge", [0, 1, 2])
def test_aux_mods(charge):
    s = pep()
    s.consume_peptide("ASMTK", 1, 1, np.array([0, 3], np.uint32),
                      np.array([42.010565, 15.994915], np.float32))
    check(s, "b", charge, [1, 0], [113.04767, 280.04603, 427.08144, 528.12912])
    check(s, "b", charge, [0, 1], [113.04767, 200.07970, 347.11510, 528.12912])
    check(s, "c", charge, [1, 0], [130.07422, 297.07258, 444.10798, 545.15567])
    check(s, "c", charge, [0, 1], [130.07422, 217.10625, 364.14165, 545.15567])
    check(s, "y", charge, [0, 1], [146.10552, 327.11953, 474.15494, 561.18697])
    check(s, "y", charge, [1, 0], [146.10552, 247.15320, 394.18861, 561.18697])
    check(s, "z", charge, [0, 1], [129.07897, 310.09298, 457.12839, 544.16042])
    check(s, "z", charge, [1, 0], [129.07897, 230.12665, 377.16206, 544.16042])
    check(s, "Z", charge, [0, 1], [130.086795, 311.100805, 458.136215, 545.168245])
    check(s, "Z", charge, [1, 0], [130.086795, 231.134475, 378.169885, 545.168245])


@pytest.mark.parametrize("charge", [0, 1, 3])
def test_neutral_losses(charge):
    s = pep(nl=[("ST", 18.01528)])
    s.consume_peptide("ASMTK", 1)
    check(s, "b", charge, [1, 0], [71.03711, 238.035471, 369.075961, 470.123641, 452.108361])
    check(s, "b", charge, [0, 1], [71.03711, 158.06914, 140.05386, 289.10963, 271.09435,
                                   470.123641, 452.108361])
    check(s, "y", charge, [0, 1], [146.10552, 327.11953, 458.16002, 545.19205, 527.17677])
    check(s, "y", charge, [1, 0], [146.10552, 247.15320, 229.13792, 378.19369, 360.17841,
                                   545.19205, 527.17677])
    mz, size, loss = s.fragments("b", 1, [0, 1])
    assert size.tolist() == [1, 2, 2, 3, 3, 4, 4] and loss.tolist() == [0, 0, 1, 0, 1, 0, 1]


def sdi(s, a, b, ftype, z, want):
    got = s.site_determining(a, b, ftype, z)
    for g, w in zip(got, want):
        assert g.shape == np.shape(w)
        assert np.allclose(g, w, rtol=1e-5, atol=0)


def test_site_determining_ions():
    s = pep()
    s.consume_peptide("ASMSK", 1)
    sdi(s, [1, 0], [0, 1], "b", 1, ([239.0427475, 370.08323747], [159.07641647, 290.11690647]))
    sdi(s, [1, 0], [0, 1], "y", 1, ([234.14537, 365.18586], [314.11171, 445.15220]))
    s.consume_peptide("ASMSK", 1, 1, np.array([3], np.uint32), np.array([15.9949146202], np.float32))
    sdi(s, [1, 0], [0, 1], "b", 1, ([239.0427475, 386.078152], [159.07641647, 306.111821]))
    sdi(s, [1, 0], [0, 1], "y", 1, ([234.14537, 381.18078], [314.11171, 461.14711]))

    s.consume_peptide("PASSSMSSEFK", 2)
    sdi(s, [1, 0, 0, 1, 0], [0, 1, 0, 1, 0], "b", 1, ([336.09550747], [256.12917647]))
    sdi(s, [1, 0, 0, 1, 0], [0, 1, 0, 1, 0], "y", 1, ([982.35929], [1062.32562]))
    sdi(s, [1, 0, 1, 0, 0], [0, 0, 1, 0, 1], "b", 1,
        ([336.09550747, 423.12753747, 590.12589847, 721.16638847, 808.19841847],
         [256.12917647, 343.16120647, 510.15956747, 641.20005747, 728.23208747]))
    sdi(s, [1, 0, 1, 0, 0], [0, 0, 1, 0, 1], "y", 1,
        ([510.25638, 597.28841, 728.32890, 895.32726, 982.35929],
         [590.22271, 677.25474, 808.29523, 975.29359, 1062.32562]))
    s.consume_peptide("PASSSMSSEFK", 2, 1, np.array([6], np.uint32),
                      np.array([15.9949146202], np.float32))
    sdi(s, [1, 0, 1, 0, 0], [0, 0, 1, 0, 1], "b", 1,
        ([336.09550747, 423.12753747, 590.12589847, 737.1613030902, 824.1933330902],
         [256.12917647, 343.16120647, 510.15956747, 657.1949720902, 744.2270020902]))
    sdi(s, [1, 0, 1, 0, 0], [0, 0, 1, 0, 1], "y", 1,
        ([510.25638, 597.28841, 744.32381, 911.32217, 998.35420],
         [590.22271, 677.25474, 824.29014, 991.28850, 1078.32053]))

    s.consume_peptide("ASMHSK", 1, 2)
    sdi(s, [1, 0], [0, 1], "b", 2,
        ([120.02556, 185.545805, 239.0427475, 254.07526, 370.083786, 507.142696],
         [80.042395, 145.56264, 159.076965, 214.092095, 290.117455, 427.176365]))
    sdi(s, [1, 0], [0, 1], "y", 2,
        ([117.576602, 186.106057, 234.14537, 251.626302, 371.20428, 502.24477],
         [157.559767, 226.089222, 291.609468, 314.111710, 451.17062, 582.211111]))


def test_peptide_print():
    s = pep()
    s.consume_peptide("ASMTK", 1, 1, np.array([0, 3], np.uint32),
                      np.array([42.010565, 15.994915], np.float32))
    assert s.get_peptide() == "n[42]AS[80]M[16]TK"
    assert s.get_peptide([0, 1]) == "n[42]ASM[16]T[80]K"
    s.consume_peptide("PASSSSSEFK", 2)
    assert s.get_peptide() == "PAS[80]S[80]SSSEFK"
    assert s.get_peptide([0, 1, 0, 1, 0]) == "PASS[80]SS[80]SEFK"


def test_spectral_processing_toy():
    masses = np.array([100., 300., 325., 350., 375., 400., 425., 450., 475., 500., 550., 1000.])
    intens = np.array([50., 200., 100., 1000., 500., 100., 1200., 200., 300., 400., 500., 50.])
    s = orc.OracleAscore(200.0, 6, "STY", PHOSPHO, kind="oracle")
    s.consume_spectra(masses, intens)
    b = s.binned()
    assert b["min_mz"] == 100.0 and b["max_mz"] == 1000.0 and b["n_bins"] == 5
    n_peaks = np.bincount(b["bin"], minlength=5)
    assert n_peaks[n_peaks > 0].tolist() == [1, 6, 2, 1]
    assert b["mz"][b["rank"] == 0].tolist() == [100., 425., 550., 1000.]


def test_full_spectra_parse():
    n_top, bin_size, n_peaks = 10, 100.0, 500
    rng = np.random.RandomState(2345)
    masses = rng.uniform(500.0, 2000.0, n_peaks)
    intens = 100.0 * rng.randn(n_peaks) + 300.0
    s = orc.OracleAscore(bin_size, n_top, "STY", PHOSPHO, kind="oracle")
    s.consume_spectra(masses, intens)
    b = s.binned()
    lo = np.floor(masses.min() / 100.0) * 100.0
    for ind in range(int(b["n_bins"])):
        sel = (masses >= lo + ind * bin_size) & (masses < lo + (ind + 1) * bin_size)
        order = np.argsort(intens[sel])[::-1][:n_top]
        got = b["bin"] == ind
        assert np.array_equal(b["mz"][got], masses[sel][order])
        assert np.array_equal(b["intensity"][got], intens[sel][order])
        assert b["rank"][got].tolist() == list(range(len(order)))


def test_log_math_vs_scipy():
    lib = orc.load("oracle")
    rng = np.random.RandomState(2345)
    pairs = [(-np.inf, 0.0), (0.0, -np.inf)] + list(zip(rng.randn(100), rng.randn(100)))
    for a, b in pairs:
        assert np.isclose(lib.orc_log_sum(a, b), logsumexp([a, b]), rtol=0, atol=1e-6)
    for n in range(1, 51):
        for k in range(1, n + 1):
            assert np.isclose(lib.orc_log_bin_coef(k, n), np.log(binom_coef(n, k)), rtol=0, atol=5e-5)


def test_binomial_vs_scipy():
    lib = orc.load("oracle")
    for p in (.1, .25, .5, .75, .9):
        for n in range(50):
            for k in range(1, n + 1):
                assert np.isclose(lib.orc_binom_log_pmf(p, k, n), binom_dist.logpmf(k, n, p),
                                  rtol=0, atol=5e-5)
                want = logsumexp([binom_dist.logpmf(k, n, p), binom_dist.logsf(k, n, p)])
                assert np.isclose(lib.orc_binom_log_pvalue(p, k, n), want, rtol=0, atol=5e-5)
                assert np.isclose(lib.orc_binom_log10_pvalue(p, k, n), np.log10(np.exp(want)),
                                  rtol=0, atol=5e-5)


def test_power_set_sums():
    lib = orc.load("oracle")
    out = np.zeros(64, np.float32)

    def sums(target, depth):
        t = np.asarray(target, np.float32)
        n = lib.orc_power_set_sums(t.ctypes.data, t.size, depth, out.ctypes.data, 64)
        return out[:n].tolist()

    assert sums([], 2) == [0.0]
    assert sums([1., 2., 3.], 2) == [0., 1., 2., 3., 4., 5.]
    assert sums([4., 5., 6.], 2) == [0., 4., 5., 6., 9., 10., 11.]

"""The bench line the round commits (profiles/*_bench_cfg2.json, written by bench.py on the GPU box) keeps the driver's
contract and is consistent with itself and with the counters it names -- checked on the CPU, so that a hand-edited or
stale record fails here: every contract key; value = PSMs / step time; roofline.achieved = algorithmic bytes of the
workload (recomputed from the synthetic batch) / the dominant kernel family's duration; frac = achieved /
peak; traffic = the named profile's own counters for that kernel; the CPU baseline's fields."""
import csv
import glob
import json
import os

import numpy as np
import pytest

from conftest import ROOT


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        pytest.skip("no committed bench record")
    return files[-1]


@pytest.fixture(scope="module")
def line():
    with open(_newest("*_bench_cfg2.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_contract_keys_and_types(line):
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["vs_baseline"] is None          # BASELINE.md has no number for this metric
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["unit"] == "PSMs/s" and "synthetic" in line["data"]
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["config"]["workload"].startswith("cfg2")
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "PSMs/s"


def test_value_and_roofline_follow_from_the_measured_times(line):
    from pyascore_amd import synth
    import bench
    psms = line["config"]["psms_total"]
    assert line["value"] == pytest.approx(psms / (line["ms_per_step"] * 1e-3), rel=1e-6)
    blocks = line["blocks"]["ms_per_step"]
    assert len(blocks) >= 5 and sorted(blocks)[len(blocks) // 2] == pytest.approx(line["ms_per_step"], rel=1e-9)
    r = line["roofline"]
    kern = r["kernel_ms"]
    dom = max(kern, key=kern.get)
    assert r["kernel"] == dom
    # the families are inside the step -- unless the line says they overlap (a batch of mixed shapes: the fused family runs
    # beside the others on the plan's side stream; never the case for cfg2, which has only that family)
    assert sum(kern.values()) <= line["ms_per_step"] * 1.001 or r.get("families_overlap")
    assert not r.get("families_overlap")
    # the algorithmic bytes of the workload, from the same seeded generator bench.py uses
    desc = synth.describe("cfg2", seed=1000)
    batch = synth.make_slice(desc)
    assert batch["n_psm"] == psms
    alg = bench.algorithmic_bytes(batch, int(np.max(batch["n_of_mod"])))
    assert r["algorithmic_bytes_per_launch"] == alg
    assert r["achieved"] == pytest.approx(alg / (kern[dom] * 1e-3) / 1e9, rel=1e-9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12)
    assert 0.0 < r["frac"] < 1.0 and r["achieved"] < r["peak"]


def test_traffic_is_the_named_profiles_counter(line):
    r = line["roofline"]
    src = r.get("traffic_source")
    if not src:
        pytest.skip("the record names no counters")
    path = os.path.join(ROOT, src)
    assert os.path.exists(path), "the record names counters that are not committed: %s" % src
    fetch = write = 0.0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"].strip('"').startswith(r["kernel"]):
                v = float(row.get("per_step") or row["mean_value"])
                fetch += v if row["counter"] == "FETCH_SIZE" else 0.0
                write += v if row["counter"] == "WRITE_SIZE" else 0.0
    assert r["traffic"] == pytest.approx((2.0 * fetch + write) * 1024.0, rel=1e-6)
    # wasted re-reads would show here first: the whole path moves less than twice its algorithmic bytes on cfg2
    assert r["traffic_over_algorithmic"] < 2.0


def test_design_quotes_the_newest_counters():
    """DESIGN.md section 8 quotes, per config, the HBM traffic of the whole step over its algorithmic bytes ("= N.NN×" in the
    row's last cell).  The figure must be the one the NEWEST committed counters of that config give (r05 verdict: the text
    quoted one profile while bench.py read the next): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernels of
    profiles/<tag>_rocprof_<cfg>/pmc_summary.csv over roofline.algorithmic_bytes_per_launch of <tag>_bench_<cfg>.json."""
    import re
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        text = f.read()
    sec = text[text.index("## 8. Measurements"):text.index("## 9. Multi-GPU")]
    checked = 0
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
        dirs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_" + cfg)))
        if not dirs:
            continue
        newest = dirs[-1]
        tag = os.path.basename(newest)[: -len("_rocprof_" + cfg)]
        bench_json = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, cfg))
        if not os.path.exists(bench_json):
            continue
        with open(bench_json) as f:
            alg = json.loads(f.read().strip().splitlines()[-1])["roofline"]["algorithmic_bytes_per_launch"]
        fetch = write = 0.0
        with open(os.path.join(newest, "pmc_summary.csv"), newline="") as f:
            for row in csv.DictReader(f):
                v = float(row.get("per_step") or row["mean_value"])
                fetch += v if row["counter"] == "FETCH_SIZE" else 0.0
                write += v if row["counter"] == "WRITE_SIZE" else 0.0
        ratio = (2.0 * fetch + write) * 1024.0 / alg
        rows = [ln for ln in sec.splitlines() if ln.startswith("| " + cfg + " ")]
        assert rows, "DESIGN.md section 8 has no row for " + cfg
        quoted = re.findall(r"=\s*([0-9]+\.[0-9]+)\s*×", rows[0].rstrip().rstrip("|").split("|")[-1])
        assert quoted, "the %s row of DESIGN.md section 8 quotes no traffic ratio" % cfg
        assert abs(float(quoted[-1]) - ratio) <= 0.02 * ratio + 0.005, \
            "%s: DESIGN.md quotes %s x, the newest counters (%s) give %.3f x" % (cfg, quoted[-1], os.path.basename(newest), ratio)
        assert tag in sec, "DESIGN.md section 8 does not name the profile it quotes (%s)" % tag
        checked += 1
    if not checked:
        pytest.skip("no committed counters")

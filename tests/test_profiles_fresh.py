"""Repository consistency (no GPU): the committed round-2 evidence was measured on the kernel sources that are committed.
bench.py stamps nothing itself -- it REFUSES a PMC summary whose `source_hash` (sha256 over csrc/ + include/) differs from
the tree's and then prints null roofline fields; this test makes that situation fail here, before the GPU box sees it."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _load(name):
    with open(os.path.join(PROF, name)) as f:
        return json.load(f)


def test_pmc_and_lookahead_evidence_match_the_sources():
    import bench
    h = bench.source_hash()
    assert _load("r02_pmc_summary.json")["source_hash"] == h, "re-run scripts/gpu/profile_r02.sh: kernel sources changed after the PMC passes"
    assert _load("r02_lookahead.json")["source_hash"] == h


def test_committed_bench_line_keeps_the_contract():
    d = _load("r02_bench_n1.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64" and "workload" in d["config"]
    for r in (d["roofline"], d["roofline_check"]):
        assert r["bound"] in ("valu", "lds", "latency", "hbm", "mfma")
        assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["traffic"] is not None and r["traffic"] > 0
    assert d["roofline"]["pmc_source"].endswith(_load("r02_pmc_summary.json")["source_hash"] + ")")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the lookahead changes the time of a step, never a result
    assert d["config"]["expansion_lookahead"] is True and d["without_lookahead"]["identical_results"] is True
    assert d["without_lookahead"]["ms_per_step"] > d["ms_per_step"]
    for k in ("batch4096", "c3", "c5", "saturating_batch"):
        assert k in d, k


def test_lookahead_evidence_is_consistent():
    l = _load("r02_lookahead.json")
    assert l["identical_results"] is True and l["with_lookahead"]["lookahead_used"] and not l["without_lookahead"]["lookahead_used"]
    w = l["with_lookahead"]
    assert w["children_halves_made"] == w["jobs_posted"] == w["shot_halves_made"]          # every posted half-job was served
    assert 0 < w["records_used"] <= w["pops"] and w["pops"] == l["without_lookahead"]["pops"]

"""The bench line contract (task statement: one JSON line with `roofline` and `cpu_baseline`), checked on the newest
line committed under profiles/ — the file bench.py itself produced on the GPU box."""
import json
from pathlib import Path

PROFILES = Path(__file__).resolve().parent.parent / "profiles"


def _newest_line():
    files = sorted(p for p in PROFILES.rglob("*_bench_line.json"))
    assert files, "no committed bench line"
    return json.loads(files[-1].read_text()), files[-1]


def test_committed_bench_line_keeps_the_contract():
    d, path = _newest_line()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (path.name, k)
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # "valu": the round-2 review asked for the vector-issue roofline as the headline figure of the VALU-bound blend kernel,
    # with the HBM figure on algorithmic bytes kept next to it (`hbm`)
    assert r["bound"] in ("hbm", "mfma", "valu") and r["peak"] > 0
    assert r["unit"] in ("GB/s", "TFLOP/s") or (r["bound"] == "valu" and "instructions/s" in r["unit"])
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if r["bound"] == "valu":
        h = r["hbm"]
        assert h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3 and h["algorithmic_bytes"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    # value is consistent with the step time it was derived from
    px = d["config"]["width"] * d["config"]["height"]
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_headline_pmc_summary_is_found_next_to_the_other_summaries():
    """bench.py attaches the committed PMC figures of the headline workload to its roofline; the SSIM / SDF summaries that
    live in the same directory must not shadow it (they did once: the line fell back to "bound": "hbm")."""
    import sys

    sys.path.insert(0, str(PROFILES.parent))
    import bench

    traffic, src, valu = bench.pmc_traffic("blend_bwd", dict(gaussians=1_000_000, width=1920, height=1080, mode="surfel"))
    assert traffic and valu and src.endswith("pmc_traffic.json") and "ssim" not in src and "sdf" not in src

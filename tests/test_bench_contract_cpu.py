"""The driver's contract for bench.py, checked without a GPU: the flags it is launched with exist, and the last committed
bench line (profiles/r1_bench.json, written by a real run on an MI355X) carries every field the contract names."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_and_single_json_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert '"%s"' % flag in src, flag
    # exactly one place prints the result, on rank 0 only
    assert len(re.findall(r"print\(json\.dumps\(out\)\)", src)) == 2     # the headline path and --workload epoch, each once
    # rank 0 must not issue a collective the other ranks do not (the profile steps run without the gradient all-reduce)
    assert "step(reduce=False)" in src
    # RANK / LOCAL_RANK / WORLD_SIZE come from the environment torch.distributed.run sets
    for env in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert 'os.environ.get("%s"' % env in src, env


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r1_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # consistency of the line itself: value = images per step / step time
    assert abs(d["value"] - 32 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # second half of the BASELINE metric: lane-coefficient error against the CPU oracle within the stated tolerance
    assert d["parity"]["lane_coeff_max_rel_err"] < d["parity"]["tolerance_rel"]

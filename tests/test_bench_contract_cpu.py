"""The driver's contract for bench.py, checked without a GPU: the flags it is launched with exist, the last committed
bench line (profiles/r2_bench.json, written by a real run on an MI355X) carries every field the contract names, and the
real entry point started PLAINLY with --gpus 2 launches its own two ranks (--dry-run: gloo, CPU stand-in step)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_and_single_json_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert '"%s"' % flag in src, flag
    # exactly one place prints the result, on rank 0 only
    assert len(re.findall(r"print\(json\.dumps\(out\)\)", src)) == 2     # the headline path and --workload epoch, each once (+ --dry-run's own line)
    # rank 0 must not issue a collective the other ranks do not (the profile steps run without the gradient all-reduce)
    assert "step(reduce=False)" in src
    # RANK / LOCAL_RANK / WORLD_SIZE come from the environment torch.distributed.run sets
    for env in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert 'os.environ.get("%s"' % env in src, env


def _run_bench(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, whatever the number of ranks
    return json.loads(lines[0])


def test_plain_launch_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` (the shape of the driver's N = 1 command) must work on its own: it re-executes under
    torch.distributed.run, both ranks rendezvous on 127.0.0.1, rank 0 prints the line, n_gpus = the ranks that formed."""
    d = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["allreduce_mean_ok"] is True


def test_launched_by_torch_distributed_run():
    """The driver's own N > 1 form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_world_size_4_dry_run():
    """Rank plumbing beyond two: four ranks rendezvous over gloo, every block is bracketed by the barrier on all of them, the
    all-reduced stand-in gradient is the mean over FOUR ranks, one line comes out (the 8-GPU scaling run is the driver's: this is
    the same code path with a CPU step)."""
    d = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 4 and d["allreduce_mean_ok"] is True


def test_measured_path_of_bench_imports_nothing_from_the_oracle():
    """bench.py may use oracle/ only as the checker: inside cpu_baseline_and_parity() and miopen_baseline() (both run after the
    clock, rank 0, N = 1).  Every other function -- the measured path -- must not import the package; its synthetic inputs
    come from the neutral synthetic_inputs module."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    allowed = {"cpu_baseline_and_parity", "miopen_baseline"}
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the legs that DO touch the oracle run after every timed region of the process: other_configs() is called before them
    assert src.index('out["other_configs"] = other_configs(a)') < src.index('out["miopen_baseline"] = miopen_baseline(') < \
        src.index('out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(')
    for node in tree.body:
        names = []
        for sub in ast.walk(node):
            if isinstance(sub, ast.ImportFrom) and (sub.module or "").split(".")[0] == "oracle":
                names.append(sub.module)
            if isinstance(sub, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in sub.names):
                names.append("import")
        if names:
            assert isinstance(node, ast.FunctionDef) and node.name in allowed, (getattr(node, "name", node), names)


def test_single_rank_dry_run():
    assert _run_bench(["--steps", "2", "--warmup", "1", "--dry-run"])["n_gpus"] == 1


def test_committed_bench_line_has_the_contract_fields():
    path = next(p for p in (os.path.join(ROOT, "profiles", "r%d_bench.json" % r) for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p))
    d = json.load(open(path))
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
    assert c["kind"].split(" ")[0] in ("reference", "port")          # round 6: "port (reference ATen call sequence)"
    if "other_configs" in d:                            # round 6: the other BASELINE configs ride in the driver-run line
        oc = d["other_configs"]
        assert set(oc) == {"config3_bp_4lanes_320x640_b64_bf16", "config5_seg_512x1024_b16_shard", "config4_epoch_3626_frames_1gpu"}
        for name, e in oc.items():
            assert e.get("value"), (name, e)
            for k in ("unit", "ms_per_step", "dtype", "workload"):
                assert k in e, (name, k)
        for name in ("config3_bp_4lanes_320x640_b64_bf16", "config5_seg_512x1024_b16_shard"):
            r2 = oc[name]["roofline"]
            assert abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-3
        assert "roofline_hbm" in oc["config3_bp_4lanes_320x640_b64_bf16"]
        assert r.get("traffic_algorithmic") and r.get("traffic")      # the pair to divide: mean measured / mean algorithmic bytes per launch
    # consistency of the line itself: value = images per step / step time
    assert abs(d["value"] - 32 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # second half of the BASELINE metric: lane-coefficient error against the CPU oracle
    par = d["parity"]
    if "tolerance_rel" in par:                       # round-1 line: fit only
        assert par["lane_coeff_max_rel_err"] < par["tolerance_rel"]
    else:                                            # backbone + fit: the three-number triple and its criterion
        t = par["lane_coeff_max_rel_err"]
        assert set(t) == {"hip_vs_cpu64", "hip_vs_cpu32", "cpu32_vs_cpu64"}
        assert par["ok"] is True and t["hip_vs_cpu64"] <= max(2 * t["cpu32_vs_cpu64"], 1e-5)
        if "train_ok" in par:                        # round 6: the gates are reported one by one; ok = all of them
            assert par["fit_ok"] and par["eval_ok"] and par["train_ok"] and par["train_mode_gated"] and not par["failed"]
        assert par["fit_only_lane_coeff_max_rel_err"] <= 1e-5

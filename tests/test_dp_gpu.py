"""The real N > 1 control flow of bench.py on the GPU box: two ranks, both on device 0 (one-GPU
box: RCCL refuses two ranks on one device, so the collectives go over gloo -- LF_BENCH_SINGLE_DEVICE / LF_BENCH_BACKEND are the
bench's test hooks), each running the real step: parameter broadcast -> HIP forward / backward -> in-place all-reduce of the
engine's flat gradient bucket -> one JSON line from rank 0.  No scaling number is taken from this."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_device_real_kernels():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(LF_BENCH_SINGLE_DEVICE="1", LF_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--min-seconds", "0", "--no-cpu-baseline", "--no-vendor-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert "2 ranks" in d["config"]["grad_allreduce"]
    chk = d["grad_allreduce_check"]
    assert chk["ranks"] == 2 and chk["bit_identical_across_ranks"] is True
    assert chk["bucket_elements"] == 2063344 + 8          # every parameter of the backbone + the reducer's tail (dp.TAIL)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0

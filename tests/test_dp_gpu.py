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


def _busy(ms=150):
    """Enqueue ~ms of independent GPU work on the current stream and return an event recorded behind it."""
    import torch
    a = torch.randn(4096, 4096, device="cuda")
    b = a @ a                                             # (library initialisation stays out of the timing)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    b = a @ a
    t1.record()
    torch.cuda.synchronize()
    reps = max(4, int(ms / max(t0.elapsed_time(t1), 1e-3)))

    def go():
        x = a
        for _ in range(reps):
            x = a @ a
        ev = torch.cuda.Event()
        ev.record()
        return ev, x
    return go


def test_reducer_and_ce_check_do_not_sync_the_host():
    """`check()` read the reduced tail with `.tolist()` on a device tensor, i.e. a D2H copy enqueued
    BEHIND the next step's forward + backward -- one full host sync per step on every rank.  Now the tail is copied to pinned
    memory right behind the collective and `check()` waits for that copy's event only.  Asserted on the stream: with ~150 ms of
    GPU work queued behind call k, call k + 1 (which inspects call k's signature) returns while that work is still running.
    Same for the deferred target-range check of the cross entropy (losses.CrossEntropyLoss2d.flush)."""
    import torch
    import torch.distributed as dist
    from lanedetection_end2end_amd import dp, losses
    own = not dist.is_initialized()
    if own:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1)      # RCCL with one rank: the bucket path through the real backend
    try:
        ps = [torch.nn.Parameter(torch.randn(1000, 1000, device="cuda")) for _ in range(4)]
        for p in ps:
            p.grad = torch.ones_like(p)
        red = dp.FlatGradAllReduce(ps)
        go = _busy()
        red(force=True)                                   # warm-up (communicator setup, pinned buffer)
        red(force=True)
        torch.cuda.synchronize()
        ev_a, _ = go()                                    # "backward of step k"
        red(force=True)                                   # step k's reduce: enqueued behind it
        assert not ev_a.query(), "the reducer waited for the GPU (step k)"
        ev_b, _ = go()                                    # "forward + backward of step k + 1"
        red(force=True)                                   # inspects step k's signature: may wait for work A, never for work B
        assert not ev_b.query(), "check() of the previous signature waited for work launched AFTER its collective"
        red.check()
        torch.cuda.synchronize()
        assert all(float(p.grad.mean()) == 1.0 for p in ps)          # AVG over one rank
        # the cross entropy's deferred label check
        ce = losses.CrossEntropyLoss2d(30.0, seg=True, nclasses=2).cuda()
        logits = torch.randn(2, 3, 64, 128, device="cuda", requires_grad=True)
        tgt = torch.randint(0, 3, (2, 64, 128), device="cuda")
        ce(logits, tgt)
        torch.cuda.synchronize()
        ce(logits, tgt)
        ev_c, _ = go()
        ce(logits, tgt)                                   # flushes the previous call's count
        assert not ev_c.query(), "the cross-entropy target check waited for the GPU"
        ce.flush()
        bad = tgt.clone()
        bad[0, 0, 0] = 7
        ce(logits, bad)                                   # out-of-range label: reported by the next call / flush / train()
        with pytest.raises(RuntimeError):
            ce.train()
    finally:
        if own:
            dist.destroy_process_group()

"""world_size-2 gloo test of the data-parallel path (runs on CPU): parameter broadcast and the flat
gradient all-reduce, with the never-trained encoder.output_conv left without a gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lanedetection_end2end_amd import dp, erfnet
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    net = erfnet.Net(in_channels=3, out_channels=2)
    dp.broadcast_parameters(net, src=0)
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(ref) for _ in range(world)]
    dist.all_gather(gathered, ref)
    same_params = all(torch.equal(gathered[0], g) for g in gathered)
    names = [n for n, _ in net.named_parameters()]
    for i, (n, p) in enumerate(net.named_parameters()):
        if n.startswith("encoder.output_conv"):
            continue                                    # no grad, as in training (ERFNet.py:84,92-93)
        p.grad = torch.full_like(p, float(rank + 1)) * (1 + i % 3)
    reducer = dp.FlatGradAllReduce(net.parameters())
    n = reducer()
    ok = True
    for i, (nme, p) in enumerate(net.named_parameters()):
        if nme.startswith("encoder.output_conv"):
            ok &= p.grad is None
        else:
            ok &= bool(torch.allclose(p.grad, torch.full_like(p, 1.5 * (1 + i % 3))))     # mean of 1 and 2
    n2 = reducer()                                      # second step: same active set
    # the active set may change between steps when it changes on EVERY rank (the pretrained schedule's head switch)
    saved = net.decoder.output_conv.weight.grad
    net.decoder.output_conv.weight.grad = None
    n3 = reducer()
    ok &= n3 == n - net.decoder.output_conv.weight.numel()
    net.decoder.output_conv.weight.grad = saved
    # in-place path: gradients that are views of one flat buffer are reduced without copies
    act = [p for nme, p in net.named_parameters() if not nme.startswith("encoder.output_conv")]
    # (the bucket spans ALL parameters + dp.TAIL slots: the parameters without a gradient are zero-filled holes)
    allp = list(net.parameters())
    total = sum(p.numel() for p in allp)
    flat = torch.zeros(total + dp.TAIL)
    off = 0
    for nme, p in net.named_parameters():
        if not nme.startswith("encoder.output_conv"):
            flat[off: off + p.numel()] = float(rank + 1)
            p.grad = flat[off: off + p.numel()].view(p.shape)
        else:
            p.grad = None
        off += p.numel()
    red2 = dp.FlatGradAllReduce(net.parameters(), flat_provider=lambda: flat)
    active = [i for i, p in enumerate(allp) if p.grad is not None]
    assert red2._in_place_flat([allp[i].grad for i in active], active) is flat
    red2()
    red2.check()
    ok &= bool(torch.allclose(torch.cat([p.grad.reshape(-1) for p in act]), torch.full((sum(p.numel() for p in act),), 1.5)))
    ok &= act[3].grad.data_ptr() >= flat.data_ptr() and red2.last_flat is flat
    # both ranks hold bit-identical buckets after the reduce
    mine = flat.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok &= bool(torch.equal(both[0], both[1]))
    # a rank-dependent active set is detected on EVERY rank -- by the signature riding in the SAME fixed-size collective (no
    # size mismatch, nobody hangs), reported when the reduced tail is inspected: at the next call, or by check()
    failed = False
    if rank == 0:
        net.encoder.output_conv.weight.grad = torch.zeros_like(net.encoder.output_conv.weight)
    reducer()                                           # returns: the host does not wait for the collective
    try:
        reducer.check()
    except RuntimeError:
        failed = True
    out.put((rank, same_params, ok, n, n2, failed, len(names)))
    dist.barrier() if False else None
    dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_params, ok, n, n2, failed, nnames in res:
        assert same_params and ok
        assert n == n2 == 2063344 - (128 * 2 + 2)        # all parameters but encoder.output_conv
        assert nnames == 228
    assert res[0][5] is True and res[1][5] is True       # both ranks saw the disagreement


def test_epoch_batches_config4_sharding():
    """BASELINE config 4: 3626 frames, 32 per GPU on 8 GPUs -> 14 steps, 42 frames dropped, ranks disjoint and in step."""
    import numpy as np
    from lanedetection_end2end_amd import dp
    per_rank = [list(dp.epoch_batches(3626, 32, rank=r, world=8, seed=5)) for r in range(8)]
    assert all(len(b) == 14 for b in per_rank)                       # same step count on every rank
    seen = np.concatenate([np.concatenate(b) for b in per_rank])
    assert seen.size == 14 * 256 == 3584 and np.unique(seen).size == seen.size and seen.max() < 3626
    for s in range(14):                                               # a step's global batch = 8 disjoint slices of 32
        assert np.unique(np.concatenate([per_rank[r][s] for r in range(8)])).size == 256
    one = list(dp.epoch_batches(3626, 32, rank=0, world=1, seed=5))
    assert len(one) == 113 and all(len(b) == 32 for b in one)        # 1 GPU: 113 steps, 10 frames dropped
    assert not np.array_equal(one[0], next(dp.epoch_batches(3626, 32, seed=5, epoch=1)))    # reshuffled per epoch
    assert np.array_equal(np.concatenate(list(dp.epoch_batches(64, 32, shuffle=False))), np.arange(64))


def test_epoch_plan_dry_run_world2_gloo():
    """BASELINE config 4's sharded epoch plan exercised with MORE than one rank before the first real SCALE run: bench.py
    --workload epoch --dry-run at world size 2 over gloo -- both ranks walk their dp.epoch_batches index batches in step (56 steps
    of 2 x 32, 42 frames dropped), every step's global batch is disjoint across the ranks, the flat gradient all-reduce runs once
    per step, one JSON line comes out of rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "epoch", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    ep = d["epoch_plan"]
    assert d["n_gpus"] == 2 and d["allreduce_mean_ok"] is True
    assert ep["steps_min_max_over_ranks"] == [56, 56] and d["steps"] == 56
    assert ep["frames_seen"] == 56 * 64 == 3584 and ep["frames_dropped"] == 42
    assert ep["all_indices_distinct"] and ep["every_step_disjoint_across_ranks"] and ep["index_max"] < 3626


def _premul_avg(per_rank_values, world):
    """What RCCL's ReduceOp.AVG computes for fp32: every rank's value times fp32(1 / world), then an fp32 sum (rank order)."""
    import numpy as np
    inv = np.float32(1.0) / np.float32(world)
    acc = np.zeros(per_rank_values.shape[1], dtype=np.float32)
    for r in range(world):
        acc = (acc + per_rank_values[r].astype(np.float32) * inv).astype(np.float32)
    return acc


def test_signature_survives_premul_avg_rounding():
    """with ReduceOp.AVG at a world size whose reciprocal is inexact in fp32 (3, 5, 6, 7, 12) the
    rank-agreement signature must neither fire on identical sets (every hash value) nor miss a single deviating rank."""
    import numpy as np
    from lanedetection_end2end_amd import dp
    sig = lambda h1, h2, h3: np.array([1.0, h1, h1 * h1, h2, h2 * h2, h3, h3 * h3, 0.0], dtype=np.float32)
    rng = np.random.default_rng(0)
    for world in (2, 3, 5, 6, 7, 8, 12, 64):
        for h1 in range(dp.HASH_MODS[0]):                      # identical on every rank: all values of every hash
            for h2, h3 in ((h1 * 2 % dp.HASH_MODS[1], h1 * 2 % dp.HASH_MODS[2]), (dp.HASH_MODS[1] - 1, dp.HASH_MODS[2] - 1)):
                same = np.stack([sig(h1, h2, h3)] * world)
                assert not dp.signature_disagrees(_premul_avg(same, world), world), (world, h1, h2, h3)
                assert not dp.signature_disagrees(same.sum(0), world)                      # gloo: SUM, divided by check
        for _ in range(200):                                   # ONE rank differs by one in ONE hash: must be seen
            h = [int(rng.integers(1, m - 1)) for m in dp.HASH_MODS]
            rows = [sig(*h)] * world
            k = int(rng.integers(0, 3))
            h2 = list(h)
            h2[k] += 1 if rng.random() < 0.5 else -1
            rows[int(rng.integers(0, world))] = sig(*h2)
            assert dp.signature_disagrees(_premul_avg(np.stack(rows), world), world), (world, h, h2)
    # the real hashes of real active sets: the reference's head switch changes every hash
    a = dp.FlatGradAllReduce._hashes(list(range(226)))
    b = dp.FlatGradAllReduce._hashes(list(range(224)) + [226, 227])
    assert a != b and max(a) < 64

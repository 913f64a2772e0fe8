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
    # in-place path: gradients that are views of one flat buffer are reduced without copies
    act = [p for nme, p in net.named_parameters() if not nme.startswith("encoder.output_conv")]
    flat = torch.full((sum(p.numel() for p in act),), float(rank + 1))
    off = 0
    for p in act:
        p.grad = flat[off: off + p.numel()].view(p.shape)
        off += p.numel()
    red2 = dp.FlatGradAllReduce(net.parameters(), flat_provider=lambda: flat)
    assert red2._in_place_flat([p.grad for p in act]) is flat
    red2()
    ok &= bool(torch.allclose(flat, torch.full_like(flat, 1.5))) and act[3].grad.data_ptr() >= flat.data_ptr()
    # a rank-dependent active set must be detected
    failed = False
    if rank == 0:
        net.encoder.output_conv.weight.grad = torch.zeros_like(net.encoder.output_conv.weight)
        try:
            reducer()
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
    assert res[0][5] is True                             # rank 0 detected the changed active set

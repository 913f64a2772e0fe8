"""Data parallelism for the hot path: one process per GPU, one flat gradient all-reduce per step.

The reference is single-device; this is the MI355X addition.  Every image (and every
(image, lane) fit) is independent, so the batch shards across ranks with no data-path collective;
the only exchange is the gradient sum: 2,063,344 fp32 = 8.25 MB, one RCCL all-reduce over xGMI
(latency-bound: ~15-100 us against a ~20 ms step, so one bucket, no overlap machinery).
BatchNorm statistics stay per replica, exactly like stock DDP without SyncBN.
``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU (tests).
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers (one flat broadcast each)."""
    for tensors in ([p.data for p in module.parameters()],
                    [b.data for b in module.buffers() if b.is_floating_point()]):
        if not tensors:
            continue
        flat = _flatten_dense_tensors(tensors)
        dist.broadcast(flat, src, group=group)
        for t, f in zip(tensors, _unflatten_dense_tensors(flat, tensors)):
            t.copy_(f)


class FlatGradAllReduce:
    """Average ``p.grad`` over the ranks in ONE all-reduce.

    Parameters whose grad is None on this rank (``encoder.output_conv`` never gets one, ERFNet.py:84,92-93;
    the unused head when ``pretrained``) are skipped.  The set of participating parameters may CHANGE between steps -- the
    reference's pretrained schedule flips ``end_to_end`` mid-run (BEV/main.py get_flags): ``decoder.output_conv2`` stops
    and ``decoder.output_conv`` starts receiving gradients -- but it must change identically on every rank (all ranks run
    the same schedule).  That is verified, not assumed: each call first all-reduces (MAX) a two-word signature of the local
    set; ranks that disagree all raise instead of hanging in a size-mismatched collective.
    """

    def __init__(self, params, group=None, flat_provider=None):
        """``flat_provider``: optional callable returning one contiguous tensor of which every ``p.grad`` is a
        view, in parameter order (``erfnet.Net.flat_grad``): then the all-reduce runs in place on it, without
        the flatten / unflatten copies."""
        self.params = list(params)
        self.group = group
        self.active = None
        self.flat_provider = flat_provider

    def _agree(self, active):
        """All ranks must hold gradients for the same parameters: MAX over ranks of (h, -h) is (h, -h) only if every h is equal."""
        h = 0
        for i in active:
            h = (h * 1000003 + i + 1) % 2147483629
        h = h * 4096 + (len(active) % 4096)
        dev = self.params[0].device if self.params else torch.device("cpu")
        sig = torch.tensor([h, -h], dtype=torch.int64, device=dev)
        dist.all_reduce(sig, op=dist.ReduceOp.MAX, group=self.group)
        if int(sig[0]) != -int(sig[1]):
            raise RuntimeError("data-parallel ranks disagree on the set of parameters with gradients (%d tensors here)" % len(active))

    def _in_place_flat(self, grads):
        flat = self.flat_provider() if self.flat_provider is not None else None
        if flat is None or flat.numel() != sum(g.numel() for g in grads):
            return None
        ptr, esz = flat.data_ptr(), flat.element_size()
        for g in grads:                       # every gradient must sit exactly at its running offset
            if g.data_ptr() != ptr or not g.is_contiguous():
                return None
            ptr += g.numel() * esz
        return flat

    def __call__(self):
        world = dist.get_world_size(self.group)
        active = [i for i, p in enumerate(self.params) if p.grad is not None]
        self.active = active
        if world == 1:
            return 0
        self._agree(active)
        if not active:
            return 0
        grads = [self.params[i].grad for i in active]
        flat = self._in_place_flat(grads)
        if flat is not None:
            dist.all_reduce(flat, group=self.group)
            flat.div_(world)
            return flat.numel()
        flat = _flatten_dense_tensors(grads)
        dist.all_reduce(flat, group=self.group)
        flat.div_(world)
        for g, f in zip(grads, _unflatten_dense_tensors(flat, grads)):
            g.copy_(f)
        return flat.numel()


def epoch_batches(n_items, per_rank_batch, rank=0, world=1, seed=0, epoch=0, shuffle=True):
    """Index batches of one epoch for this rank (BASELINE config 4: 3626 frames, 32 per GPU, 8 GPUs -> 14 steps, the last
    42 frames dropped).  Every rank draws the SAME permutation (seed + epoch) and takes its own ``per_rank_batch`` slice
    of each global batch, so the ranks run the same number of steps (drop_last over the GLOBAL batch: no rank ever waits
    in an all-reduce the others skipped) and no frame is seen twice.  The reference's loader is single-process:
    ``get_loader`` truncates the index list to whole batches and samples it with ``SubsetRandomSampler``
    (BEV/Dataloader/Load_Data_new.py:305-320); this is the data-parallel form of that rule."""
    import numpy as np
    if not (0 <= rank < world) or per_rank_batch < 1:
        raise ValueError("epoch_batches: rank %d of %d, batch %d" % (rank, world, per_rank_batch))
    gb = per_rank_batch * world
    steps = n_items // gb
    order = np.random.default_rng(seed + epoch).permutation(n_items) if shuffle else np.arange(n_items)
    for s in range(steps):
        lo = s * gb + rank * per_rank_batch
        yield order[lo: lo + per_rank_batch]

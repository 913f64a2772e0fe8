"""The reference's training-loop bodies executed on the mirror (Missing #5; round 3, Missing #4: the BP tree).

The reference itself is absent on the GPU box, so its main.py cannot run here; this is a replica of the statement sequence of
BEV/main.py:200-266 -- same import names through the path arrangement of tools/run_reference_main.py (the mirrored tree first
on sys.path: ``from Networks.LSQ_layer import Net``, ``from Loss_crit import define_loss_crit, polynomial``), a stub loader
yielding the loader's 6-tuples, ``model(input, end_to_end)`` -> per-lane ``criterion`` -> ``loss.item()`` -> ``zero_grad /
backward / step``, the ``except RuntimeError: continue`` skip of a singular batch, the --clas branch, and the exact-area metric
on ``.cpu()`` copies.  ``test_bp_main_loop_body`` does the same for BP/main.py:232-337: the ``skip`` branch (``early_return=True`` ->
``criterion_seg``, :256-263), ``model(input, gt_line, end_to_end, gt=gt)`` -> ``criterion(beta_k, gt_k, valid_points[:, k])`` per lane
-> ``/ nclasses`` (:286-305), the segmentation-mode branch with its ``no_grad`` metric (:306-318), and the --clas losses (:321-326)."""
import importlib
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree_on_path(name):
    tree = os.path.join(ROOT, "lanedetection_end2end_amd", name)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "Networks" or k.startswith("Networks.") or k == "Loss_crit"}
    sys.path.insert(0, tree)
    try:
        yield
    finally:
        sys.path.remove(tree)
        for k in [k for k in sys.modules if k == "Networks" or k.startswith("Networks.") or k == "Loss_crit"]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.fixture()
def bev_tree_on_path():
    yield from _tree_on_path("bev")


@pytest.fixture()
def bp_tree_on_path():
    yield from _tree_on_path("bp")


class AverageMeter:          # BEV/Networks/utils.py AverageMeter, the four lines the loop uses
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n


@pytest.mark.parametrize("clas", [False, True])
def test_bev_main_loop_body(bev_tree_on_path, clas):
    Net = importlib.import_module("Networks.LSQ_layer").Net              # BEV/main.py:24
    Loss_crit = importlib.import_module("Loss_crit")                      # BEV/main.py:23
    define_loss_crit, polynomial = Loss_crit.define_loss_crit, Loss_crit.polynomial
    from lanedetection_end2end_amd.optim import define_optim             # same signature as Networks.utils.define_optim
    N, R = 2, (256 if clas else 64)      # the --clas heads are built for the 32 x 64 encoder output of resize 256 (LSQ_layer.py:270-277)
    args = Namespace(batch_size=N, nclasses=2, resize=R, end_to_end=True, mod="erfnet", layers=18, channels_in=3, pretrained=False,
                     pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=0.0, use_cholesky=False,
                     mask_percentage=0.3, clas=clas, loss_policy="area", weight_funct="none", weight_seg=30, optimizer="adam",
                     learning_rate=1e-4, weight_decay=0.0, clip_grad_norm=0, weight_fit=1.0, weight_class=1.0)
    torch.manual_seed(3)
    model = Net(args)                                                      # main.py:74
    if not args.no_cuda:
        model = model.cuda()                                               # main.py:77-83
    optimizer = define_optim(args.optimizer, model.parameters(), args.learning_rate, args.weight_decay)   # main.py:80-81
    criterion, criterion_seg = define_loss_crit(args)                      # main.py:87
    criterion_line_class = nn.CrossEntropyLoss().cuda()                    # main.py:88
    criterion_horizon = nn.BCEWithLogitsLoss().cuda()                      # main.py:89
    rng = np.random.default_rng(11)

    def loader(nbatches):                                                  # the 6-tuple of Load_Data_new.__getitem__ batches
        for i in range(nbatches):
            yield (torch.from_numpy(inputs.images(N, R, 2 * R, seed=300 + i)), torch.zeros(N, R, 2 * R, dtype=torch.int64),
                   torch.from_numpy(inputs.bev_gt_params(N, seed=400 + i)), torch.arange(N) + i * N,
                   torch.from_numpy(rng.integers(0, 3, (N, 4))), torch.from_numpy((rng.uniform(0, 1, (N, R)) > 0.5).astype(np.float32)))

    losses, exact_area = AverageMeter(), AverageMeter()
    model.train()                                                          # main.py:197
    skipped, stepped = 0, 0
    w0 = model.net.encoder.initial_block.conv.weight.detach().clone()
    for i, (input, gt, params, idx, gt_line, gt_horizon) in enumerate(loader(4)):
        if not args.no_cuda:
            input, params = input.cuda(non_blocking=True), params.cuda(non_blocking=True)
            input = input.float()
        assert params.size(1) == 4
        gt0, gt1, gt2, gt3 = params[:, 0, :], params[:, 1, :], params[:, 2, :], params[:, 3, :]
        if i == 1:          # a batch whose weight maps vanish: singular normal matrix -> RuntimeError -> skipped (main.py:216-219)
            keep = (model.net.decoder.output_conv.weight.detach().clone(), model.net.decoder.output_conv.bias.detach().clone())
            with torch.no_grad():
                model.net.decoder.output_conv.weight.zero_()
                model.net.decoder.output_conv.bias.zero_()
        try:
            beta0, beta1, beta2, beta3, weightmap_zeros, M, output_net, outputs_line, outputs_horizon = model(input, args.end_to_end)
        except RuntimeError as e:
            assert i == 1, e
            skipped += 1
            with torch.no_grad():
                model.net.decoder.output_conv.weight.copy_(keep[0])
                model.net.decoder.output_conv.bias.copy_(keep[1])
            continue
        assert beta2 is None and beta3 is None and tuple(M.shape) == (N, 3, 3)
        loss = criterion(beta0, gt0) + criterion(beta1, gt1)                # main.py:223
        if args.clas:                                                       # main.py:246-253
            gt_horizon, gt_line = gt_horizon.cuda(non_blocking=True), gt_line.cuda(non_blocking=True)
            _, line_pred = torch.max(outputs_line, 1)
            loss_horizon = criterion_horizon(outputs_horizon, gt_horizon)
            loss_line = criterion_line_class(outputs_line, gt_line)
            loss = loss * args.weight_fit + (loss_line + loss_horizon) * args.weight_class
            assert line_pred.shape == (N, 4)
        else:
            assert outputs_line is None and outputs_horizon is None
        losses.update(loss.item(), input.size(0))                          # main.py:257
        optimizer.zero_grad()                                               # main.py:264-266
        loss.backward()
        optimizer.step()
        stepped += 1
        with torch.no_grad():                                               # main.py:273-280: exact area on .cpu() copies
            trap_left = polynomial(beta0.cpu()).trapezoidal(polynomial(gt0.cpu()))
            trap_right = polynomial(beta1.cpu()).trapezoidal(polynomial(gt1.cpu()))
            exact_area.update(((trap_left + trap_right) / 2).mean().item(), input.size(0))
    assert skipped == 1 and stepped == 3 and losses.count == 3 * N
    assert np.isfinite(losses.sum) and np.isfinite(exact_area.sum) and exact_area.sum > 0
    assert not torch.equal(model.net.encoder.initial_block.conv.weight.detach(), w0)       # the optimizer moved the weights
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.parametrize("clas,nclasses", [(False, 4), (False, 2), (True, 4)])
def test_bp_main_loop_body(bp_tree_on_path, clas, nclasses):
    Net = importlib.import_module("Networks.LSQ_layer").Net              # BP/main.py:25
    Loss_crit = importlib.import_module("Loss_crit")                      # BP/main.py:24
    define_loss_crit = Loss_crit.define_loss_crit
    from lanedetection_end2end_amd.optim import define_optim             # same signature as Networks.utils.define_optim
    N, R = 2, (256 if clas else 64)      # the --clas heads are built for the 32 x 64 encoder output of resize 256
    # --pretrained True: the decoder carries output_conv2 (nclasses + 1 logits) for the segmentation pretraining epochs and
    # output_conv for the end-to-end ones; get_flags (main.py:32-44) flips args.end_to_end between them
    args = Namespace(batch_size=N, nclasses=nclasses, resize=R, end_to_end=True, mod="erfnet", layers=18, channels_in=3,
                     pretrained=True, pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=0.0,
                     use_cholesky=False, mask_percentage=0.2, clas=clas, no_mapping=False, loss_policy="backproject",
                     weight_funct="none", weight_seg=30, optimizer="adam", learning_rate=1e-4, weight_decay=0.0,
                     clip_grad_norm=0, weight_fit=1.0, weight_class=1.0)
    torch.manual_seed(5)
    model = Net(args)                                                      # main.py:94
    if not args.no_cuda:
        model = model.cuda()                                               # main.py:97-100
    optimizer = define_optim(args.optimizer, model.parameters(), args.learning_rate, args.weight_decay)   # main.py:102-103
    criterion, criterion_seg = define_loss_crit(args)                      # main.py:108
    criterion_horizon = nn.BCEWithLogitsLoss().cuda()                      # main.py:109
    criterion_line_class = nn.BCEWithLogitsLoss().cuda()                   # main.py:110
    rng = np.random.default_rng(13)

    def loader(nbatches):                                                  # the 7-tuple of BP Load_Data_new.__getitem__ batches
        for i in range(nbatches):
            lanes, valid = inputs.bp_targets(N, 4, R, seed=700 + i)
            yield (torch.from_numpy(inputs.images(N, R, 2 * R, seed=300 + i)),
                   torch.from_numpy(inputs.seg_targets(N, R, 2 * R, nclasses + 1, seed=500 + i)).unsqueeze(1),
                   torch.from_numpy(lanes), torch.arange(N) + i * N, torch.zeros(N, 4),
                   torch.from_numpy((rng.uniform(0, 1, (N, R)) > 0.5).astype(np.float32)), torch.from_numpy(valid))

    losses, rmse_metric, losses_skip = AverageMeter(), AverageMeter(), AverageMeter()
    model.train()                                                          # main.py:226
    w0 = model.net.encoder.initial_block.conv.weight.detach().clone()
    # get_flags over the epochs of a pretrained run: skip (epoch < skip_epochs), segmentation mode (epoch < pretrain_epochs), end to end
    schedule = [(True, False), (False, False), (False, True), (False, True)]
    for i, (input, gt, lanes, idx, gt_line, gt_horizon, valid_points) in enumerate(loader(len(schedule))):
        skip, args.end_to_end = schedule[i]
        if not args.no_cuda:                                               # main.py:245-248
            input, lanes = input.cuda(), lanes.cuda()
            valid_points = valid_points.cuda()
            gt = gt.cuda().squeeze(1)
        assert lanes.size(1) == 4
        gt0, gt1, gt2, gt3 = lanes[:, 0, :], lanes[:, 1, :], lanes[:, 2, :], lanes[:, 3, :]
        if skip:                                                           # main.py:254-262
            output_net = model(input, gt_line, args.end_to_end, early_return=True)
            assert tuple(output_net.shape) == (N, nclasses + 1, R, 2 * R)          # decoder.output_conv2
            loss = criterion_seg(output_net, gt)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            losses_skip.update(loss.item(), input.size(0))
            continue
        beta0, beta1, beta2, beta3, weightmap_zeros, output_net, outputs_line, outputs_horizon, output_seg = \
            model(input, gt_line, args.end_to_end, gt=gt)                  # main.py:287-289
        assert output_net.shape[1] == nclasses + int(not args.end_to_end)
        if args.end_to_end:                                                # main.py:296-305
            loss_left, x_cal0 = criterion(beta0, gt0, valid_points[:, 0])
            loss_right, x_cal1 = criterion(beta1, gt1, valid_points[:, 1])
            if args.nclasses > 3:
                loss_left1, x_cal2 = criterion(beta2, gt2, valid_points[:, 2])
                loss_right1, x_cal3 = criterion(beta3, gt3, valid_points[:, 3])
                loss_left += loss_left1
                loss_right += loss_right1
            else:
                assert beta2 is None and beta3 is None
            loss = (loss_left + loss_right) / args.nclasses
            assert x_cal0.shape == (N, 56) and beta0.dtype == torch.float64
        else:                                                              # main.py:306-318
            loss = criterion_seg(output_net, gt)
            with torch.no_grad():
                loss_left, x_cal0 = criterion(beta0, gt0, valid_points[:, 0])
                loss_right, x_cal1 = criterion(beta1, gt1, valid_points[:, 1])
                if args.nclasses > 3:
                    loss_left1, x_cal2 = criterion(beta2, gt2, valid_points[:, 2])
                    loss_right1, x_cal3 = criterion(beta3, gt3, valid_points[:, 3])
                    loss_left += loss_left1
                    loss_right += loss_right1
                loss_metric = (loss_left + loss_right) / args.nclasses
                rmse_metric.update(loss_metric.item(), input.size(0))
        if args.clas and args.end_to_end:                                  # main.py:321-326 (the heads run with end_to_end only)
            gt_horizon, gt_line = gt_horizon.cuda(), gt_line.cuda()
            loss_horizon = criterion_horizon(outputs_horizon, gt_horizon).double()
            loss_line = criterion_line_class(outputs_line, gt_line).double()
            loss = loss * args.weight_fit + (loss_line + loss_horizon) * args.weight_class
            assert tuple(outputs_line.shape) == (N, 4) and tuple(outputs_horizon.shape) == (N, R)
        losses.update(loss.item(), input.size(0))                          # main.py:331
        optimizer.zero_grad()                                              # main.py:338-340
        loss.backward()
        optimizer.step()
    criterion_seg.flush()                  # (the deferred target-range check of the last batch: losses.CrossEntropyLoss2d)
    assert losses_skip.count == N and losses.count == 3 * N and rmse_metric.count == N
    # the head switch of the schedule: both heads were trained at some point, each only in its own phase
    assert model.net.decoder.output_conv.weight.grad is not None
    assert np.isfinite(losses.sum) and np.isfinite(losses_skip.sum) and np.isfinite(rmse_metric.sum)
    assert not torch.equal(model.net.encoder.initial_block.conv.weight.detach(), w0)       # the optimizer moved the weights
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.parametrize("end_to_end", [True, False])
def test_bev_validate_body(bev_tree_on_path, end_to_end):
    """BEV/main.py:362-501, validate(): ``model.eval()`` under ``torch.no_grad()``, the loader's 7-tuples, ``model(input,
    args.end_to_end)`` inside the ``except RuntimeError: continue`` guard, ``criterion`` per lane (end-to-end) or
    ``criterion_seg(output_net, gt)`` + the area metric (segmentation mode), the exact-area metric on ``.cpu()`` copies through
    ``polynomial(...).trapezoidal(...)``, ``AverageMeter`` updates from ``.item()`` -- statement by statement on the mirror's import
    names.  Checked against the fp64 oracle in the same mode (eval: running statistics) and for what eval mode promises: no
    gradient graph, running statistics and ``num_batches_tracked`` untouched, a second pass bit-identical."""
    Net = importlib.import_module("Networks.LSQ_layer").Net
    Loss_crit = importlib.import_module("Loss_crit")
    define_loss_crit, polynomial = Loss_crit.define_loss_crit, Loss_crit.polynomial
    from oracle import e2e_oracle, erfnet_oracle
    N, R = 2, 64
    args = Namespace(batch_size=N, nclasses=2, resize=R, end_to_end=end_to_end, mod="erfnet", layers=18, channels_in=3, pretrained=False,
                     pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=0.0, use_cholesky=False,
                     mask_percentage=0.3, clas=False, loss_policy="area", weight_funct="none", weight_seg=30, evaluate=True, print_freq=1)
    torch.manual_seed(5)
    model = Net(args).cuda()
    criterion, criterion_seg = define_loss_crit(args)
    # non-trivial running statistics: one train-mode step first (what a checkpoint carries into validate())
    model.train()
    with torch.no_grad():
        model(torch.from_numpy(inputs.images(N, R, 2 * R, seed=700)).cuda(), True)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def loader(nbatches):                                                  # the 7-tuple of the validation loader (main.py:378)
        rng = np.random.default_rng(21)                                    # the same batches on every pass
        for i in range(nbatches):
            yield (torch.from_numpy(inputs.images(N, R, 2 * R, seed=710 + i)),
                   torch.from_numpy(rng.integers(0, 3, (N, 1, R, 2 * R))), torch.from_numpy(inputs.bev_gt_params(N, seed=720 + i)),
                   torch.arange(N) + i * N, torch.from_numpy(rng.integers(0, 3, (N, 4))),
                   torch.from_numpy((rng.uniform(0, 1, (N, R)) > 0.5).astype(np.float32)), list(range(i * N, (i + 1) * N)))

    def validate():
        losses, avg_area, avg_trapezium_rule = AverageMeter(), AverageMeter(), AverageMeter()
        model.eval()                                                        # main.py:373
        seen = []
        with torch.no_grad():                                               # main.py:376
            for i, (input, gt, params, idx, gt_line, gt_horizon, index) in enumerate(loader(3)):
                if not args.no_cuda:
                    input, params = input.cuda(non_blocking=True), params.cuda(non_blocking=True)
                    input = input.float()
                gt0, gt1, gt2, gt3 = params[:, 0, :], params[:, 1, :], params[:, 2, :], params[:, 3, :]
                try:                                                        # main.py:386-392
                    beta0, beta1, beta2, beta3, weightmap_zeros, M, output_net, outputs_line, outputs_horizon = model(input, args.end_to_end)
                except RuntimeError as e:
                    print("Batch with idx {} skipped due to singular matrix".format(idx.numpy()), e)
                    continue
                if args.end_to_end:                                         # main.py:395-396
                    loss = criterion(beta0, gt0) + criterion(beta1, gt1)
                else:                                                       # main.py:412-416
                    gt = gt.cuda(non_blocking=True)
                    loss = criterion_seg(output_net, gt)
                    area = criterion(beta0, gt0) + criterion(beta1, gt1)
                    avg_area.update(area.item(), input.size(0))
                gt_left_lines = polynomial(gt0.cpu())                       # main.py:437-445
                gt_right_lines = polynomial(gt1.cpu())
                pred_left_lines = polynomial(beta0.cpu())
                pred_right_lines = polynomial(beta1.cpu())
                trap_left = pred_left_lines.trapezoidal(gt_left_lines)
                trap_right = pred_right_lines.trapezoidal(gt_right_lines)
                avg_trapezium_rule.update(((trap_left + trap_right) / 2).mean().item(), input.size(0))
                losses.update(loss.item(), input.size(0))
                assert not loss.requires_grad and not output_net.requires_grad and beta0.grad_fn is None
                seen.append((input.cpu(), params.cpu().numpy(), torch.stack([beta0, beta1], 1)[..., 0].cpu().numpy(), loss.item(),
                             output_net.cpu(), gt if not args.end_to_end else None))
        return losses, avg_area, avg_trapezium_rule, seen

    l1, a1, t1, seen = validate()
    assert l1.count == 3 * N and np.isfinite(l1.sum) and np.isfinite(t1.sum) and (end_to_end or np.isfinite(a1.sum))
    # eval mode leaves the model untouched
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    l2, _, t2, seen2 = validate()
    assert l2.sum == l1.sum and t2.sum == t1.sum and all(np.array_equal(u[2], v[2]) for u, v in zip(seen, seen2))
    # against the CPU oracle in eval mode: logits of every batch, coefficients / loss where the fit is end to end
    P = {k: v.detach().cpu() for k, v in model.net.state_dict().items()}
    for x, params, beta, loss, logits, gt in seen:
        o = e2e_oracle.bev_step(x, P, params, torch.float64, R, training=False)
        scale = np.abs(o["logits"]).max()
        assert np.abs(logits.double().numpy() - o["logits"]).max() <= 2e-5 * scale
        if end_to_end:
            assert e2e_oracle.relerr(beta, o["beta"]) <= 1e-5 and abs(loss - o["loss"]) <= 1e-5 * abs(o["loss"])

"""On-device input pipeline: decoded uint8 frames + label maps already in HBM -> the
``(image, gt[, horizon])`` tensors ``LaneDataset.__getitem__`` produces (BEV/Dataloader/Load_Data_new.py:62-117,
BP/Dataloader/Load_Data_new.py:110-197), batched, bit-identical to the PIL/torchvision result.

Only the pixel work lives here (crop, BILINEAR / NEAREST resize, class remap, flip, ToTensor).  The per-sample
label metadata (polynomial parameters, lane x-coordinates, line-type lists) stays host-side numpy as in the
reference; ``flip_params_bev`` / ``flip_lanes_bp`` / ``mirror_list`` below mirror its flip statements.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class InputPipeline:
    """``InputPipeline(resize, tree='bev'|'bp', nclasses=2, frame_hw=(720, 1280), crop=640)``;
    ``__call__(frames_u8, labels_u8=None, flip=None) -> (image, gt, horizon)``.

    frames_u8 (N, H, W, 3) uint8 and labels_u8 (N, H, W) uint8 on the GPU; flip (N,) bool (the reference draws
    ``np.random.uniform() > 0.5 and flip_on`` per training sample).  image (N,3,R,2R) fp32, gt (N,1,R,2R) int64,
    horizon (N,R) fp32 for tree='bev' (the BP tree derives its horizon from the lane coordinates on the host)."""

    def __init__(self, resize, tree="bev", nclasses=2, frame_hw=(720, 1280), crop=640):
        assert tree in ("bev", "bp")
        lib = _lib.load()
        H, W = frame_hw
        self.resize, self.tree, self.nclasses, self.frame_hw = resize, tree, nclasses, (H, W)
        self.handle = lib.lf_pipeline_plan_create(H, W, H - crop, crop, resize, 2 * resize)
        if not self.handle:
            raise _lib.LaneFitLibraryError("lf_pipeline_plan_create failed: %s" % lib.lf_last_error().decode())
        self.mode = (1 if (tree == "bev" or nclasses < 3) else 0) | (2 if tree == "bp" else 0)
        self._tables = None
        self._lut = None
        # Out-of-pool indices (``index`` form): the kernels count them on the device and read pool entry 0 instead; the count of
        # call k is copied to pinned memory right behind its kernels (``_lib.DeferredRead``) and inspected at the START of call
        # k + 1 or by flush() -- IndexError like the reference's tensor indexing, one call late, without a host sync per step.
        self._bad = None
        self._pending = None

    def flush(self):
        """Raise now if the previous indexed call saw an index outside the pool."""
        pend, self._pending = self._pending, None
        if pend is not None:
            n = int(pend[0].get()[0])
            if n:
                self._bad.zero_()
                raise IndexError("InputPipeline: %d index value(s) outside the pool of %d frames" % (n, pend[1]))

    def __del__(self):
        try:
            _lib.load().lf_pipeline_plan_destroy(self.handle)
        except Exception:
            pass

    def host_tables(self):
        """(bounds_x, weights_x, bounds_y, weights_y, nearest_x, nearest_y) as numpy int32 (tests)."""
        lib = _lib.load()
        ksx, ksy = ctypes.c_int(), ctypes.c_int()
        lib.lf_pipeline_tables_host(self.handle, ctypes.byref(ksx), ctypes.byref(ksy), None, None, None, None, None, None)
        R = self.resize
        arrs = [np.zeros(n, dtype=np.int32) for n in (4 * R, 2 * R * ksx.value, 2 * R, R * ksy.value, 2 * R, R)]
        ptrs = [a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
        lib.lf_pipeline_tables_host(self.handle, None, None, *ptrs)
        bx, kx, by, ky, ntx, nty = arrs
        return bx.reshape(-1, 2), kx.reshape(2 * R, -1), by.reshape(-1, 2), ky.reshape(R, -1), ntx, nty

    def _device_state(self, device):
        lib = _lib.load()
        if self._tables is None or self._tables.device != device:
            self._tables = torch.empty(lib.lf_pipeline_table_bytes(self.handle), dtype=torch.uint8, device=device)
            _lib.check(lib.lf_pipeline_upload(self.handle, _lib.ptr(self._tables), _lib.stream()), "lf_pipeline_upload")
            # (ToTensor()(gt) * 255).long() per palette index, evaluated with the reference's fp32 ops
            v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
            self._lut = (v * 255).long().to(device)
        return self._tables, self._lut

    def __call__(self, frames_u8, labels_u8=None, flip=None, index=None):
        """``index`` (N,) int64 on the device: ``frames_u8`` / ``labels_u8`` are a resident POOL and batch element n is pool entry
        ``index[n]`` -- gathered inside the kernels (``lf_pipeline_image_indexed``), no copy of the N selected frames first."""
        lib = _lib.load()
        if not frames_u8.is_cuda:
            raise _lib.LaneFitLibraryError("InputPipeline needs the decoded frames on the MI355X; there is no CPU path")
        pool, H, W, C = frames_u8.shape
        assert (H, W) == self.frame_hw and C == 3 and frames_u8.dtype == torch.uint8
        dev = frames_u8.device
        tables, lut = self._device_state(dev)
        R = self.resize
        sel = None
        N = pool
        if index is not None:
            self.flush()
            sel = index.to(device=dev, dtype=torch.int64).contiguous()
            N = sel.numel()
            if self._bad is None or self._bad.device != dev:
                self._bad = torch.zeros(1, dtype=torch.int32, device=dev)
        fl = None if flip is None else flip.to(device=dev, dtype=torch.uint8).contiguous()
        frames_u8 = frames_u8.contiguous()
        image = torch.empty(N, 3, R, 2 * R, dtype=torch.float32, device=dev)
        if sel is None:
            _lib.check(lib.lf_pipeline_image(self.handle, _lib.ptr(frames_u8), N, _lib.ptr(tables), _lib.ptr(fl),
                                             _lib.ptr(image), _lib.stream()), "lf_pipeline_image")
        else:
            _lib.check(lib.lf_pipeline_image_indexed(self.handle, _lib.ptr(frames_u8), pool, _lib.ptr(sel), N, _lib.ptr(tables),
                                                     _lib.ptr(fl), _lib.ptr(image), _lib.ptr(self._bad), _lib.stream()),
                       "lf_pipeline_image_indexed")
        gt = horizon = None
        if labels_u8 is not None:
            assert labels_u8.shape == (pool, H, W) and labels_u8.dtype == torch.uint8
            labels_u8 = labels_u8.contiguous()
            gt = torch.empty(N, 1, R, 2 * R, dtype=torch.int64, device=dev)
            if self.tree == "bev":
                horizon = torch.empty(N, R, dtype=torch.float32, device=dev)
            if sel is None:
                _lib.check(lib.lf_pipeline_label(self.handle, _lib.ptr(labels_u8), N, _lib.ptr(tables), _lib.ptr(fl),
                                                 self.mode, _lib.ptr(lut), _lib.ptr(gt), _lib.ptr(horizon), _lib.stream()),
                           "lf_pipeline_label")
            else:
                _lib.check(lib.lf_pipeline_label_indexed(self.handle, _lib.ptr(labels_u8), pool, _lib.ptr(sel), N, _lib.ptr(tables),
                                                         _lib.ptr(fl), self.mode, _lib.ptr(lut), _lib.ptr(gt), _lib.ptr(horizon),
                                                         None, _lib.stream()), "lf_pipeline_label_indexed")
        if sel is not None:
            self._pending = (_lib.DeferredRead(self._bad), pool)      # (the image kernel counted: the label kernel sees the same indices)
        return image, gt, horizon


def mirror_list(lst):
    """Load_Data_new.py ``mirror_list``: mirror the line-type list for a flipped sample."""
    middle = len(lst) // 2
    return list(reversed(lst[middle:])) + list(reversed(lst[:middle]))


def flip_params_bev(params):
    """BEV/Dataloader/Load_Data_new.py:93-96 -- params (4, 3) of a flipped sample."""
    p = -np.asarray(params, dtype=np.float64)[[1, 0, 3, 2]]
    p[:, -1] = 1 + p[:, -1]
    return p


def flip_lanes_bp(lanes, resize):
    """BP/Dataloader/Load_Data_new.py:166-168 -- lanes (4, 56) in resized pixels, -2 = absent."""
    lanes = np.asarray(lanes, dtype=np.float64)
    track = lanes < 0
    out = (2 * resize - 1) - lanes
    out[track] = -2
    return out[[1, 0, 3, 2]]

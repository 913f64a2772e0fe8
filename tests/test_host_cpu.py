"""CPU-only checks of the host layer: the C-ABI library loads and exports every symbol the header
declares, the engine plan (a host object) is consistent with the model, module surfaces mirror the
reference's, and the product refuses to run without a GPU instead of falling back."""
import os
import re
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import ROOT, relerr
from oracle import erfnet_oracle, fit_oracle


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "lanefit.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "liblanefit_hip.so does not export " + s
    assert lib.lf_abi_version() == 5
    declared = set(_lib.exported_symbols())
    assert set(syms) <= declared | {"lf_erfnet_plan"}, sorted(set(syms) - declared)


def test_plan_matches_model():
    from lanedetection_end2end_amd import _lib, erfnet
    lib = _lib.load()
    for cout, heads in ((2, 1), (4, 1), (2, 2)):
        net = erfnet.Net(in_channels=3, out_channels=cout, pretrained=heads == 2)
        plan = erfnet._Plan(32, 256, 512, 3, cout, heads)
        assert plan.n_params == len(list(net.parameters())) == 228 + 2 * (heads - 1)
        assert plan.n_bn == len(net._batchnorms()) == 39
        assert plan.n_drop == len(net._dropouts()) == 13
        assert plan.drop_floats == 32 * (5 * 64 + 8 * 128)
        # sized per precision mode, independent of the mode the plan was last set to
        f32, b16 = plan.workspace_bytes("fp32"), plan.workspace_bytes("bf16")
        assert 0 < f32 < 16 << 30 and f32 == plan.workspace_bytes("fp32x9") and b16 > f32      # read-once weight gradient: larger partial-row regions
        assert lib.lf_erfnet_set_precision(plan.handle, 2) == 0 and plan.workspace_bytes("fp32") == f32
        assert lib.lf_erfnet_workspace_bytes(plan.handle) == b16 and lib.lf_erfnet_workspace_bytes_for(plan.handle, 1) == 0
        assert lib.lf_erfnet_set_precision(plan.handle, 1) != 0 and lib.lf_erfnet_set_precision(plan.handle, 4) != 0    # removed modes
        assert sum(net._used_param_mask(0)) == 228 - 2
    assert not lib.lf_erfnet_plan_create(1, 100, 200, 3, 2, 1)        # H % 16 != 0 -> error, no crash
    assert b"unsupported" in lib.lf_last_error()


def test_state_dict_surface_and_reference_init():
    from lanedetection_end2end_amd.bev.Networks import define_model
    net = define_model('erfnet', layers=18, in_channels=3, out_channels=2, pretrained=False, pool=True)
    spec = erfnet_oracle.param_spec(3, 2)
    sd = net.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    with pytest.raises(KeyError):
        define_model('resnet')
    # the reference's weights_init_kaiming matches on class names (BEV/Networks/utils.py:490-503)
    touched = []

    def init_like_reference(m):
        name = m.__class__.__name__
        if name.find('Conv') != -1 or name.find('Linear') != -1:
            touched.append(m.weight.shape)
            torch.nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in', nonlinearity='relu')
            m.bias.data.zero_()
        elif name.find('BatchNorm2d') != -1:
            torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
    net.apply(init_like_reference)
    assert len(touched) == 74 + 1            # 71 Conv2d + 3 ConvTranspose2d + the unused encoder.output_conv


def test_geometry_matches_oracle(golden_fit):
    from lanedetection_end2end_amd import geometry
    M, Mi = geometry.bev_homography()
    Mo, Mio = fit_oracle.bev_homography()
    assert np.abs(M - Mo).max() < 1e-14 and np.abs(Mi - Mio).max() < 1e-14
    for r in (256, 320):
        assert np.abs(geometry.get_homography(r)[0] - golden_fit["bp_M_%d" % r]).max() < 1e-12
    g = geometry.projective_grid(64, 128, M, True).numpy()
    assert np.abs(g - golden_fit["bev_grid_64x128_f32"]).max() <= 1e-7
    assert np.array_equal(geometry.get_homography(256, True)[0], np.identity(3))


def test_split_lanes_contract():
    from lanedetection_end2end_amd.fit import split_lanes
    beta = torch.arange(2 * 4 * 3, dtype=torch.float64).view(2, 4, 3)
    b0, b1, b2, b3 = split_lanes(beta[:, :2], 2, torch.float32)
    assert b2 is None and b3 is None and b0.shape == (2, 3, 1) and b0.dtype == torch.float32
    outs = split_lanes(beta, 4, torch.float64)
    assert all(o.shape == (2, 3, 1) and o.dtype == torch.float64 for o in outs)
    assert torch.equal(outs[3][:, :, 0], beta[:, 3])


def test_no_cpu_fallback():
    from lanedetection_end2end_amd import _lib, fit, losses
    from lanedetection_end2end_amd.bev.Networks import define_model
    with pytest.raises(_lib.LaneFitLibraryError):
        fit.fit_lanes(torch.zeros(1, 2, 8, 8), torch.zeros(64, 2), 0)
    with pytest.raises(_lib.LaneFitLibraryError):
        losses.Area_Loss(2, "none")(torch.zeros(2, 3, 1), torch.ones(2, 3))
    net = define_model('erfnet', layers=18, in_channels=3, out_channels=2, pretrained=False, pool=True)
    with pytest.raises(_lib.LaneFitLibraryError):
        net(torch.zeros(1, 3, 64, 128), True)
    with pytest.raises(NotImplementedError):
        losses.Area_Loss(3, "none")
    with pytest.raises(NotImplementedError):
        losses.Area_Loss(2, "cubic")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lanedetection_end2end_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), os.path.join(dp, f)


def test_trapezoid_cpu(golden_fit):
    from lanedetection_end2end_amd.losses import polynomial
    b = torch.tensor([[0.1, -0.2, 0.5], [0, 0.1, 0.4]], dtype=torch.float64)
    g = torch.tensor([[0.05, -0.1, 0.45], [0.01, 0.2, 0.5]], dtype=torch.float64)
    tz = polynomial(b.unsqueeze(2)).trapezoidal(polynomial(g))
    assert np.allclose(tz.numpy(), golden_fit["trapezoid_survey"], atol=1e-12)


@pytest.mark.skipif(not os.path.isdir("/root/reference/Birds_Eye_View_Loss"), reason="reference tree not present")
@pytest.mark.parametrize("tree,d", [("bev", "Birds_Eye_View_Loss"), ("bp", "Backprojection_Loss")])
def test_mirror_shadows_reference_imports(tree, d):
    """The reference's own import statements resolve to this package when the mirror tree precedes the
    reference tree on sys.path, while Networks.utils still comes from the reference."""
    import subprocess
    import sys
    code = r'''
import os, sys, types
sys.path.insert(0, %r)
sys.modules["cv2"] = types.ModuleType("cv2")
os.environ["LANEFIT_REFERENCE_ROOT"] = "/root/reference"
sys.path.insert(0, "/root/reference/%s")
sys.path.insert(0, %r)
import Networks
from Networks.LSQ_layer import Net
from Networks.utils import define_args
from Loss_crit import define_loss_crit, polynomial
assert Net.__module__ == "lanedetection_end2end_amd.lsq" and define_args.__module__ == "Networks.utils"
assert Networks.model_dict["erfnet"].__module__.startswith("lanedetection_end2end_amd")
print("ok")
''' % (ROOT, d, os.path.join(ROOT, "lanedetection_end2end_amd", tree))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def test_classification_module_and_chain_plan():
    """--clas heads: the module's state_dict is the reference's (oracle spec pinned by the golden generator),
    and the conv-chain plan (host-side object) sizes its workspace without a device."""
    import ctypes
    from lanedetection_end2end_amd import _lib, clas
    from oracle import clas_oracle
    for ct in ("line", "horizon"):
        m = clas.Classification(ct, size=(32, 64), channels_in=128, resize=256)
        spec = clas_oracle.clas_param_spec(ct)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    plan = clas._ChainPlan(4, 32, 64, (128, 128, 128, 64, 64), (1, 3, 3, 3))
    saved = 4 * 32 * 64 * (128 + 128 + 64 + 64) * 4
    assert plan.ws_bytes > saved and plan.ws_bytes < 12 * saved
    lib = _lib.load()
    bad = (ctypes.c_int * 2)(128, 24)
    ks = (ctypes.c_int * 1)(3)
    assert not lib.lf_convchain_plan_create(1, 8, 8, 1, bad, ks)
    assert b"multiples of 16" in lib.lf_last_error()
    with pytest.raises(_lib.LaneFitLibraryError):
        m(torch.zeros(1, 128, 32, 64))          # no CPU path


def test_blocks_are_bound_to_their_layer_range():
    """Block-level surface (round 4): every sub-module knows its layer range of the engine plan; the plan's layer geometry (a
    host object) agrees with the modules' channel counts and the cumulative-stride table of erfnet.py; the binding survives
    deepcopy and pickling (weak references are re-bound), and a block refuses host tensors like the whole network does."""
    import copy
    import ctypes
    import io
    from lanedetection_end2end_amd import _lib, erfnet
    lib = _lib.load()
    net = erfnet.Net(in_channels=3, out_channels=2, pretrained=True)
    blocks = net._blocks()
    assert len(blocks) == 22 and net.encoder._lf_range == (0, 16) and net.decoder._lf_range == (16, 22)
    plan = erfnet._Plan(2, 64, 128, 3, 2, 2)
    assert lib.lf_erfnet_num_layers(plan.handle) == 22
    io6 = (ctypes.c_int * 6)()
    for i, b in enumerate(blocks):
        assert b._lf_range == (i, i + 1) and b._owner() is net
        assert lib.lf_erfnet_layer_io(plan.handle, i, io6) == 0
        s = erfnet._LAYER_IN_STRIDE[i]
        assert (io6[1], io6[2]) == (64 // s, 128 // s), (i, list(io6))
        conv = b.conv if hasattr(b, "conv") else b.conv3x1_1
        cin = conv.in_channels
        assert io6[0] == cin, (i, io6[0], cin)
        mask = net._range_param_mask(i, i + 1, -1)
        assert sum(mask) == len(list(b.parameters()))
    assert lib.lf_erfnet_layer_io(plan.handle, 22, io6) != 0
    for clone in (copy.deepcopy(net),):
        assert clone.encoder._owner() is clone and clone.decoder.layers[2]._owner() is clone
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert back.encoder.layers[4]._owner() is back
    with pytest.raises(_lib.LaneFitLibraryError):
        net.encoder.layers[1](torch.zeros(1, 64, 16, 32))
    orphan = erfnet.non_bottleneck_1d(64, 0.0, 1)
    with pytest.raises(RuntimeError):
        orphan(torch.zeros(1, 64, 16, 32))

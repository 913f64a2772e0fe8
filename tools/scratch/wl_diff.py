import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from lanedetection_end2end_amd import _lib
lib = _lib.load(); st = _lib.stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.lf_debug_set_ops_precision(2)
N, C, H, W, axis, d = 64, 128, 40, 80, 1, 8
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device="cuda").bfloat16(); w = torch.randn(C, C, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
out = {}
for mode in (0, 4):
    lib.lf_debug_set_bf16_lds(mode)
    y = torch.zeros_like(x)
    _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
    torch.cuda.synchronize()
    out[mode] = y.float().reshape(-1, C)
diff = (out[0] != out[4])
print("differing elements", int(diff.sum()), "of", diff.numel())
px = diff.any(1).nonzero().flatten()
print("differing pixels", len(px), "first", px[:20].tolist(), "last", px[-5:].tolist())
if len(px):
    items = torch.unique(px // 256)
    print("items (256 px)", len(items), items[:40].tolist())
    print("sub-tiles within item", torch.unique((px % 256) // 64).tolist(), "channels", torch.unique(diff.nonzero()[:, 1] // 32).tolist())
    p = int(px[0]); print("pixel", p, "ref", out[0][p, :8].tolist(), "wl", out[4][p, :8].tolist())
lib.lf_debug_set_ops_precision(0)

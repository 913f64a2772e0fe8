import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from lanedetection_end2end_amd import _lib
from tools.bf16_ab import timeit
lib = _lib.load(); st = _lib.stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.lf_debug_set_ops_precision(2)
for (N, C, H, W) in ((64, 16, 160, 320), (32, 16, 128, 256)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, device="cuda") * 0.1; b = torch.randn(C, device="cuda")
    y = torch.empty_like(x); gx = torch.empty_like(x); scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    for axis in (0, 1):
        out = []
        res = {}
        for name, mode in (("general", 2), ("lean", 4)):
            lib.lf_debug_set_bf16_lds(mode)
            f = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, 1, 1, P(scratch), st), "fwd")
            gfn = lambda: _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), P(x), P(gx), N, H, W, C, axis, 1, P(scratch), st), "dgrad")
            out.append("%s fwd %6.1f dgrad %6.1f us" % (name, timeit(f, 100), timeit(gfn, 100)))
            res[name] = (y.float().clone(), gx.float().clone())
        e = max((res["general"][0] - res["lean"][0]).abs().max().item(), (res["general"][1] - res["lean"][1]).abs().max().item())
        nb = 2 * N * H * W * C * 2
        print("N=%d C=%d %dx%d axis %d | %s | max |general - lean| %.3g | tensors %.0f MB" % (N, C, H, W, axis, " | ".join(out), e, nb / 1e6), flush=True)
lib.lf_debug_set_ops_precision(0)

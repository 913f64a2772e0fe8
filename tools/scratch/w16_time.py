import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from lanedetection_end2end_amd import _lib
from tools.bf16_ab import timeit
lib = _lib.load(); st = _lib.stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.lf_debug_set_ops_precision(2)
for (N, C, H, W) in ((64, 16, 160, 320), (32, 16, 128, 256), (3, 16, 20, 48)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    for axis in (0, 1):
        out, res = [], {}
        for name, mode in (("scalar loads", 2), ("transposing reads", 4)):
            lib.lf_debug_set_bf16_lds(mode)
            gw = torch.empty(C, C, 3, device="cuda"); gb = torch.empty(C, device="cuda")
            f = lambda: _lib.check(lib.lf_conv1d_bwd_weight(P(x), P(gy), P(gw), P(gb), N, H, W, C, axis, 1, P(scratch), st), "wgrad")
            out.append("%s %6.1f us" % (name, timeit(f, 50)))
            res[name] = (gw.clone(), gb.clone())
        a, b = res["scalar loads"], res["transposing reads"]
        e = ((a[0] - b[0]).abs().max() / a[0].abs().max()).item(); e2 = ((a[1] - b[1]).abs().max() / a[1].abs().max()).item()
        print("N=%d C=%d %dx%d axis %d | %s | relative difference gw %.2e gb %.2e" % (N, C, H, W, axis, " | ".join(out), e, e2), flush=True)
lib.lf_debug_set_ops_precision(0); lib.lf_debug_set_bf16_lds(4)

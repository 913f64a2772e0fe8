import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from lanedetection_end2end_amd import _lib
lib = _lib.load(); st = _lib.stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.lf_debug_set_ops_precision(2)
for (N, C, H, W, axis, d) in ((64, 128, 40, 80, 1, 8), (64, 64, 80, 160, 0, 1), (32, 128, 32, 64, 0, 4)):
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); w = torch.randn(C, C, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    lib.lf_debug_set_bf16_lds(0)
    ref = torch.zeros_like(x)
    _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(ref), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
    torch.cuda.synchronize()
    for name, mode in (("ring", 2), ("whole-line", 4)):
        lib.lf_debug_set_bf16_lds(mode)
        bad, badpx = 0, []
        for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
            y = torch.zeros_like(x)
            _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
            torch.cuda.synchronize()
            dpx = (y.reshape(-1, C) != ref.reshape(-1, C)).any(1).nonzero().flatten()
            if len(dpx):
                bad += 1
                badpx.append((len(dpx), int(dpx[0]) % 256))
        print("N=%d C=%d %dx%d | %-10s runs with a mismatch: %d  %s" % (N, C, H, W, name, bad, badpx[:8]), flush=True)
lib.lf_debug_set_ops_precision(0)

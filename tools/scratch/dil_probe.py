import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from lanedetection_end2end_amd import _lib
from tools.bf16_ab import timeit
lib = _lib.load(); st = _lib.stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.lf_debug_set_ops_precision(2)
for (N, C, H, W) in ((64, 128, 40, 80), (64, 64, 80, 160), (32, 128, 32, 64)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); w = torch.randn(C, C, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    y = torch.empty_like(x); scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    for axis in (0, 1):
        for d in (0, 1, 4, 8, 16):
            out = []
            for name, mode in (("LDS", 1), ("ring64", 6), ("ring128", 10)):
                if C == 64 and name == "ring128": continue
                lib.lf_debug_set_bf16_lds(mode)
                f = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
                out.append("%s %6.1f" % (name, timeit(f, 100)))
            print("N=%d C=%d %dx%d axis %d dil %2d | %s" % (N, C, H, W, axis, d, " | ".join(out)), flush=True)
lib.lf_debug_set_ops_precision(0)

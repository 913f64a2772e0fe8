"""No hot kernel may spill vector registers to scratch.

Compiles csrc/lf_conv.hip for gfx950 with -save-temps (hipcc cross-compiles without a GPU) and reads the
.vgpr_spill_count / .private_segment_fixed_size metadata of every instantiation the backbone launches in the
fp32 and fp32x9 modes (tap-GEMM with every compiled-in epilogue, the 16-channel kernels, the weight gradients).
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def conv_kernels(tmp_path_factory):
    from lanedetection_end2end_amd import build
    import isa_meta
    d = tmp_path_factory.mktemp("isa")
    src = os.path.join(build.CSRC, "lf_conv.hip")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.FLAGS + ["-c", src, "-o", str(d / "lf_conv.o"), "-save-temps=obj"]
    subprocess.check_call(cmd, cwd=str(d))
    asm = glob.glob(str(d / "*gfx950*.s"))
    assert asm, "no device assembly produced"
    return isa_meta.kernels(asm[0])


def _select(kernels, prefix):
    return [k for k in kernels if k["name"].startswith(prefix)]


def test_no_vgpr_spills_in_network_kernels(conv_kernels):
    hot = []
    # the 64-channel-slab tap-GEMM, every (prologue, epilogue) the network launches, both row forms; the run-time-flag forms
    hot += _select(conv_kernels, "tapgemm_kernel<")
    hot += _select(conv_kernels, "tapgemm_lean_kernel<")
    hot += _select(conv_kernels, "tapgemm_split_kernel<")
    hot += _select(conv_kernels, "tapgemm_bf16_kernel<")
    hot += _select(conv_kernels, "tapgemm_bf16_ring_kernel<")
    hot += _select(conv_kernels, "tapgemm_bf16_wl_kernel<")
    hot += _select(conv_kernels, "tapgemm_bf16_wv_kernel<")
    hot += _select(conv_kernels, "tapgemm_bf16_lean_kernel<")
    hot += _select(conv_kernels, "tapwgrad_kernel<")
    hot += _select(conv_kernels, "tapwgrad16_kernel<")
    assert len(hot) > 100
    # no scratch access inside any matrix loop ...
    bad = [(k["name"], k["loop_scratch"]) for k in hot if k["loop_scratch"] != 0]
    assert not bad, "kernels spilling inside the MFMA loop: %r" % bad
    # ... and none anywhere else
    bad = [(k["name"], k["vgpr_spill"], k["scratch"]) for k in hot if k["vgpr_spill"] != 0]
    assert not bad, "kernels spilling vector registers: %r" % bad


def test_register_budgets(conv_kernels):
    by = {k["name"]: k for k in conv_kernels}
    # round 4: two accumulator sets (128 registers) + two operand sets (64) -- two waves per SIMD (<= 256 registers) for every
    # compiled-in variant the step launches, both row forms (round 3: one accumulator set, three waves per SIMD at <= 168)
    hot = [k for k in conv_kernels if k["name"].startswith("tapgemm_kernel<4, ") and ", -1, " not in k["name"]]
    assert len(hot) >= 18
    for k in hot:
        assert k["vgpr"] + k["agpr"] <= 256, (k["name"], k["vgpr"], k["agpr"])
    # the run-time-flag forms (cold paths) take the whole register file instead of spilling: one wave per SIMD
    for name in ("tapgemm_kernel<4, 0, -1, true, false>", "tapgemm_kernel<4, 1, -1, false, false>"):
        assert by[name]["vgpr_spill"] == 0 and by[name]["vgpr"] + by[name]["agpr"] <= 512, (name, by[name])
    # the 16-channel kernels keep four workgroups per CU (<= 128 registers)
    for k in _select(conv_kernels, "tapgemm_lean_kernel<") + _select(conv_kernels, "tapgemm_bf16_lean_kernel<"):
        assert k["vgpr"] <= 128, (k["name"], k["vgpr"])
    # the bf16 LDS kernels run two workgroups per CU (<= 256 registers; their LDS budgets are checked by the launcher's occupancy query)
    for k in _select(conv_kernels, "tapgemm_bf16_ring_kernel<") + _select(conv_kernels, "tapgemm_bf16_wl_kernel<"):
        assert k["vgpr"] + k["agpr"] <= 256 and k["scratch"] == 0, (k["name"], k["vgpr"], k["scratch"])
    # the wave-private 64-channel kernel: two workgroups per CU (<= 256 registers) except with two staged operand tiles / run-time flags
    # (one workgroup per CU by LDS: the whole register file instead of spills)
    for k in _select(conv_kernels, "tapgemm_bf16_wv_kernel<"):
        two = not any(k["name"].startswith("tapgemm_bf16_wv_kernel<%s>" % e) for e in ("-1", "34", "38"))
        assert k["vgpr"] + k["agpr"] <= (256 if two else 512) and k["scratch"] == 0, (k["name"], k["vgpr"], k["agpr"], k["scratch"])
    # the phase-stamp code exists only in the DBG instantiations
    dbg = [k["name"] for k in conv_kernels if k["name"].endswith(", true>") and k["name"].startswith(("tapgemm_kernel<", "tapwgrad_kernel<"))
           and k["name"].count("true>")]
    assert any(n.startswith("tapgemm_kernel<4, 0, 0, true, true>") for n in dbg)

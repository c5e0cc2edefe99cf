"""the decision-exact compositing (csrc/wg_alpha.h: ref_expf_nonpos) restates, by hand, the float32 `exp` expansion llvm emits
on gfx950 for the reference's `exp(power)` (forward.cu:364, backward.cu:543).  This device unit test compares the restatement with the
compiler's OWN `exp(float)` -- same toolchain, -ffp-contract=off like oracle/_ref's no-contraction build -- on every float32 argument in
[-104, -0] (1.12e9 values): a ROCm upgrade that changes the lowering fails here directly, on the GPU box, not only in the parity tests."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "native", "libexp_probe.so")


def test_exp_probe_is_built_and_exports_its_entry_point():
    assert os.path.exists(LIB), "run __graft_entry__.build() (wild-gaussians_amd/build.py: build_exp_probe)"
    import subprocess
    syms = subprocess.run(["nm", "-D", LIB], capture_output=True, text=True).stdout   # (no dlopen: the library pulls in the HIP runtime)
    assert " T exp_probe_run" in syms


@pytest.mark.gpu
def test_restated_exp_expansion_equals_the_compilers_own_exp_on_every_argument_of_interest():
    lib = C.CDLL(LIB)
    out = (C.c_ulonglong * 4)()
    assert lib.exp_probe_run(out) == 0
    mismatches, first_bits, tiny_bad, tested = (int(v) for v in out)
    assert tested == 0x42D00000 + 1                    # every float32 bit pattern from -0.0 down to -104.0
    assert mismatches == 0, (mismatches, hex(first_bits))   # [-87, -0]: bit for bit
    assert tiny_bad == 0                               # [-104, -87): both far below any alpha >= 1/255

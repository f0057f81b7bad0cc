"""The epilogue's specialised atan / tan (csrc/inflx_ops.h: OCML's algorithms restricted to the arguments
ops::complete_analysis can produce, src/anguelova.rs:128,132) equal OCML's general atan / tan bit for bit on that
domain -- 16 million arguments compared on the device, the special values included -- and so does the quick epilogue's
square root (the compiler's sqrt without operand scaling, behind a range guard)."""

import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("horner_mode", [0, 2])
def test_specialised_atan_and_tan_equal_ocml_bit_for_bit(gpu_lib, tmp_path, horner_mode):
    """Random order (every wavefront mixes the branches) and sorted (wave-uniform: the scalar branches skip what no lane
    needs), both spellings of the Horner step (the kernels' default and the plain one), and the reciprocal without
    special-case handling that the quick epilogue uses inside atan."""
    from inflatox_amd.compiler import hipcc_path

    exe = tmp_path / "epilogue_math_probe"
    csrc = os.path.join(ROOT, "inflatox_amd", "csrc")
    subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-fno-fast-math", "-ffp-contract=on", f"-DINFLX_HORNER_MODE={horner_mode}", f"-I{csrc}", os.path.join(ROOT, "tests", "epilogue_math_probe.hip"), "-o", str(exe)], check=True)
    proc = subprocess.run([str(exe), "16"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "atan mismatches 0, tan mismatches 0" in proc.stdout, proc.stdout
    # the quick epilogue's square root (inflx_sqrt_quick) equals the compiler's wherever its guard accepts the argument
    assert ", mismatches 0\n" in proc.stdout and "quick sqrt:" in proc.stdout, proc.stdout

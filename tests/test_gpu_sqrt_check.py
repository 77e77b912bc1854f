"""msd_sqrt_cr's sign-bit form (msd_mag_impl.h) gives 0 for x = 0 only because this hardware's fused
multiply-add hands an all-ones NaN through with its sign bit -- IEEE 754 does not promise that.  The exhaustive
comparison with the compiler's correctly rounded sqrtf (every float in [2^-40, 2) and zero: 343 932 929 values,
scripts/micro/sqrt_check.hip) therefore runs on the target as a test: another toolchain or architecture that
canonicalises NaNs fails here, not in a receiver's magnitudes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_msd_sqrt_cr_equals_sqrtf_on_this_hardware(torch_cuda, tmp_path):
    exe = str(tmp_path / "sqrt_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "readsb-protobuf_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "scripts", "micro", "sqrt_check.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "343932929 values, 0 differ" in out.stdout, out.stdout + out.stderr

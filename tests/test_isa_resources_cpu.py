"""(CPU) The production GEMM kernels keep their spill-free register budget: hipcc's kernel-resource-usage remarks for gemm.hip, with the package's
own flags, must report NO scratch segment for any kernel.  Round 6 found out why this is a test: two epilogue features compiled into the shared
store_tile pushed the 256 x 256 tile (254 registers of 256) into 276 bytes of scratch per lane - in every launch of that tile, whether or not the
feature was used, invisible to every same-binary A/B - until tools/isa_resources.py was run."""
import re
import subprocess

from ultravox_amd import build as B


def test_no_gemm_kernel_has_a_scratch_segment():
    src = B.CSRC / "gemm.hip"
    cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src.name, []), "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", str(src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    names = re.findall(r"Function Name: (\S+)", err)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)]
    assert len(names) == len(scratch) and len(names) >= 20, (len(names), len(scratch))
    spilled = [(n, s) for n, s in zip(names, scratch) if s]
    assert not spilled, spilled

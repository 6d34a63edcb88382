"""No kernel of the library uses scratch memory: the gfx950 assembly of dsim_api.hip, compiled
with the build's flags, reports ScratchSize 0 for every kernel.  CPU only (hipcc cross-compiles); ~1 minute."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_kernel_uses_scratch(tmp_path):
    import __graft_entry__ as ge
    flags = [f for f in ge.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    assert "-fno-slp-vectorize" in flags
    out = tmp_path / "dsim.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "dronesim_amd", "csrc", "dsim_api.hip")], stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for ln in out.read_text().split("\n"):
        m = re.match(r"^(_Z\w+|k_\w+):", ln)
        if m:
            cur = m.group(1)
        m = re.match(r"\s*;\s*ScratchSize: (\d+)", ln)
        if m and cur:
            kernels[cur] = int(m.group(1))
    assert len(kernels) > 150                                  # every instance of every kernel template
    spilling = {k: v for k, v in kernels.items() if v != 0}
    assert not spilling, spilling

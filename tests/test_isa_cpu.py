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


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_halo_pack_ticket_is_fenced_at_device_scope(tmp_path):
    """VERDICT r03 (weak 6): the completion ticket of k_halo_pack — the last workgroup to take it reads what the others
    produced and writes the message headers — sits between a device-scope release (L2 write-back, buffer_wbl2 sc1, behind
    the workgroup barrier) and a device-scope acquire (buffer_inv sc1) in the gfx950 code of the build's flags."""
    import __graft_entry__ as ge
    flags = [f for f in ge.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = tmp_path / "dsim.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "dronesim_amd", "csrc", "dsim_api.hip")], stderr=subprocess.DEVNULL)
    txt = out.read_text()
    body = txt[txt.index("_Z11k_halo_pack5HaloK:"):]
    body = body[:body.index(".end_amdhsa_kernel")].split("\n")
    ops = [ln.strip().split()[0] for ln in body if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
    # the ticket: the LAST 32-bit returning atomic add of the kernel (the slot reservations come first)
    tick = max(i for i, ln in enumerate(body) if "global_atomic_add " in ln and " sc0" in ln)
    before = [ln for ln in body[:tick] if "buffer_wbl2" in ln and "sc1" in ln]
    after = [ln for ln in body[tick:] if "buffer_inv" in ln and "sc1" in ln]
    barrier_before = any("s_barrier" in ln for ln in body[:tick])
    assert before and after and barrier_before, (len(before), len(after), barrier_before)
    assert "s_barrier" in ops

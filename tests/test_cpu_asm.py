"""not gpu: static checks of the gfx950 code hipcc generates for kernels whose correctness leans on hand-placed waits."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_asm_transpose_reads_are_waited_for(tmp_path):
    """binhip_wgrad.hip issues ds_read_b64_tr_b16 from inline asm (so that the compiler does not fence the LDS-DMA prefetch)
    and waits with its own s_waitcnt lgkmcnt(0): nothing may touch a destination register in between."""
    from check_asm_lds_hazard import check
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "wgrad.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", os.path.join(ROOT, "bin_amd", "csrc", "binhip_wgrad.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=300)
    bad, kernels, n_reads = check(str(out), "wgrad1x1")
    assert kernels == 6 and n_reads > 100, (kernels, n_reads)
    assert not bad, bad[:5]
    bad, kernels, n_reads = check(str(out), "wgrad_mfma_kernel")      # the generic double-buffered kernel (5x5, wide 1x1)
    assert kernels == 4 and n_reads > 100 and not bad, (kernels, n_reads, bad[:5])
    bad, kernels, n_reads = check(str(out), "wgrad3x3_db_kernel")     # the 3x3 kernel: 80 (f16x3) + 40 (f16) reads per tile
    assert kernels == 2 and n_reads == 120 and not bad, (kernels, n_reads, bad[:5])
    # and the reason for the asm: no compiler-inserted vmcnt(0) between the prefetch DMA and the fragment reads of a stage
    text = out.read_text()
    for name in ("_Z15wgrad1x1_kernelILi3ELi1ELi2EEv10WgradKArgs", "_Z15wgrad1x1_kernelILi3ELi2ELi1EEv10WgradKArgs"):
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = [l.split(";")[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l]
        for i, l in enumerate(lines):
            if l.startswith("ds_read_b64_tr_b16"):
                prev = [p for p in lines[max(0, i - 6):i] if p.startswith("s_waitcnt")]
                assert not any("vmcnt" in p for p in prev), (name, lines[max(0, i - 6):i + 1])

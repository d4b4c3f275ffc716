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
    assert "wgrad3x3_db_kernel" not in out.read_text()               # round 2's form lives in tools/experiments (side builds)
    bad, kernels, n_reads = check(str(out), "wgrad3x3_xrow_kernel")   # wave = X row (round 3): 48 + 24 reads per tile
    assert kernels == 2 and n_reads == 72 and not bad, (kernels, n_reads, bad[:5])
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


def _write_kernel(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n")
    return str(f)


def test_hazard_checker_on_handwritten_streams(tmp_path):
    """the checker itself: counted lgkmcnt retires reads in issue order, in-flight state follows branches, barriers and
    address reuse are flagged."""
    from check_asm_lds_hazard import check
    ok = """\tds_read_b64_tr_b16 v[10:11], v2
\tds_read_b64_tr_b16 v[12:13], v2 offset:128
\tds_read_b64_tr_b16 v[14:15], v3
\ts_waitcnt lgkmcnt(1)
\tv_mfma_f32_32x32x16_f16 v[20:35], v[10:13], v[40:43], v[20:35]
\ts_waitcnt lgkmcnt(0)
\tv_mov_b32_e32 v16, v14
"""
    bad, kernels, reads = check(_write_kernel(tmp_path, ok))
    assert kernels == 1 and reads == 3 and not bad, bad
    early = ok.replace("lgkmcnt(1)", "lgkmcnt(2)")                      # the MFMA now reads v[12:13] while it is in flight
    bad, _, _ = check(_write_kernel(tmp_path, early))
    assert len(bad) == 1 and "v_mfma" in bad[0][3]
    branchy = """\tds_read_b64_tr_b16 v[10:11], v2
\ts_cbranch_scc1 .LBB0_2
\tv_add_u32_e32 v5, v6, v7
.LBB0_2:
\tv_mov_b32_e32 v8, v10
\ts_waitcnt lgkmcnt(0)
"""
    bad, _, _ = check(_write_kernel(tmp_path, branchy))                    # reached with the read in flight on BOTH paths
    assert len(bad) == 1 and "v_mov_b32_e32 v8, v10" in bad[0][3]
    barrier = "\tds_read_b64_tr_b16 v[10:11], v2\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)\n"
    bad, _, _ = check(_write_kernel(tmp_path, barrier))
    assert len(bad) == 1 and "barrier" in bad[0][2]
    reuse = "\tds_read_b64_tr_b16 v[10:11], v2\n\tds_read_b64_tr_b16 v[12:13], v10\n\ts_waitcnt lgkmcnt(0)\n"
    bad, _, _ = check(_write_kernel(tmp_path, reuse))
    assert len(bad) == 1 and "in-flight destination" in bad[0][2]

"""Three-phase dense-block launch (BINHIP_PLAN_RDB3) vs the per-conv launches: every output bit of whole RDN calls must
agree, on ragged / multi-image shapes, repeatedly, and while another stream keeps the chip unevenly busy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import _lib as L, ops
from bin_amd.models.archs.RDN import PRECISIONS
from bin_amd.rdn_plan import RdnWeights, rdn_forward
from bin_amd.weights import canonical_weights

canon = {k: torch.from_numpy(v).cuda() for k, v in canonical_weights(0).items()}
g = torch.Generator().manual_seed(5)
wts = RdnWeights(canon, 3, 3, prefix="model2.")
bad = 0
noise_a = torch.randn(4096, 4096, device="cuda")
side = torch.cuda.Stream()
for (n, h, w) in ((1, 64, 96), (2, 40, 72), (1, 256, 256), (3, 130, 190), (1, 768, 1344), (5, 320, 320)):
    ins = [torch.rand(n, 3, h, w, generator=g).cuda() for _ in range(3)]
    ref = rdn_forward(wts, ins, flags=0).clone()
    for rep in range(6):
        if rep >= 3:                      # uneven background load on another stream
            with torch.cuda.stream(side):
                for _ in range(3):
                    noise_a @ noise_a
        out = rdn_forward(wts, ins, flags=L.PLAN_RDB3)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        bad += 0 if same else 1
        if not same:
            print(f"MISMATCH n={n} {h}x{w} rep {rep}: max diff {float((out - ref).abs().max()):.3e}")
    try:
        ops.check_status()
    except RuntimeError as e:
        print("status:", e); bad += 1
    t = []
    for flags in (0, L.PLAN_RDB3):
        for _ in range(3):
            rdn_forward(wts, ins, flags=flags)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            rdn_forward(wts, ins, flags=flags)
        torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"n={n} {h}x{w}: per-conv launches {t[0]:.3f} ms, three-phase {t[1]:.3f} ms per RDN call", flush=True)
print("RDB3 CHECK", "FAILED" if bad else "OK", bad)

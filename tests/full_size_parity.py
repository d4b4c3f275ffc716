"""One-off (GPU box, ~1-2 min of host CPU): BASELINE config 1 at FULL size — a 1280x720 window padded to 768x1344 —
through the HIP path vs the oracle (the CPU restatement pinned to the reference), all 14 outputs.  Not part of the
pytest suites (the oracle forward alone takes about a minute); run by hand, result recorded in DESIGN.md §4."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from bin_amd.models.archs.RDN import bin_stage4_lstm          # noqa: E402
from bin_amd.utils import util                                # noqa: E402
from bin_amd.weights import canonical_weights, reference_state_dict, synthetic_frames   # noqa: E402
from oracle import rdn_oracle as O                            # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
frames = synthetic_frames(1234, 1, 720, 1280, 6)
pads = util.pad_sizes(720, 1280)
padded = [util.replicate_pad(f, pads) for f in frames]
canon = {k: torch.from_numpy(v) for k, v in canonical_weights(0).items()}
t0 = time.time()
with torch.no_grad():
    ref = O.bin_stage4_forward(padded, canon)
print(f"oracle forward at {tuple(padded[0].shape)}: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads", flush=True)
l, r, t, b = pads
target = util.tensor2img(frames[3][0])
for prec in ("f16", "f16x3"):
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.cuda().eval().set_precision(prec)
    with torch.no_grad():
        out = net(*[p.cuda() for p in padded])
    errs = [float((o.cpu() - q).abs().max()) for o, q in zip(out, ref)]
    dps = []
    for idx in (13, 8, 12):
        a = util.tensor2img(out[idx][0])[t:t + 720, l:l + 1280]
        c = util.tensor2img(ref[idx][0])[t:t + 720, l:l + 1280]
        dps.append(abs(util.calculate_psnr(a, target) - util.calculate_psnr(c, target)))
    print(f"{prec}: max-abs over 14 outputs {max(errs):.3e} (per output min {min(errs):.1e}), "
          f"|dPSNR| of Ft_p[13,8,12] vs the oracle's {max(dps):.5f} dB", flush=True)

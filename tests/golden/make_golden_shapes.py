#!/usr/bin/env python3
"""G10: the reference's RDN classes with constructor arguments OTHER than bin_stage4's (reference RDN.py:168-186 takes any
G0 / D / C / G): forward output and the gradients of every input and parameter for a seeded upstream gradient, from the
REFERENCE modules' own autograd (build container only: imports /root/reference).  In the same run the oracle restatement is
asserted equal to the reference on every case.

Run:  python tests/golden/make_golden_shapes.py     (writes tests/golden/g10_rdn_shapes.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference")

import models.archs.RDN as REF                  # noqa: E402  (the reference)

from shape_cases import CASES                   # noqa: E402
from bin_amd.weights import general_rdn_weights  # noqa: E402
from oracle import rdn_oracle as O              # noqa: E402

torch.set_num_threads(8)
CLS = {2: REF.RDN_residual_interp_2_input, 3: REF.RDN_residual_interp_2_1_input, 5: REF.RDN_residual_interp_4_1_input}

if __name__ == "__main__":
    out = {}
    for tag, (k, shape, n, h, w) in CASES.items():
        G0, D, C, G = shape
        W = {nm: torch.from_numpy(v) for nm, v in general_rdn_weights(0, k, shape).items()}
        mod = CLS[k](G0=G0, D=D, C=C, G=G)
        mod.load_state_dict(W, strict=True)
        g = torch.Generator().manual_seed(100 + k)
        ins = [torch.rand(n, 3, h, w, generator=g).requires_grad_(True) for _ in range(k)]
        gout = torch.randn(n, 3, h, w, generator=g) * 1e-3
        y = mod(*ins)
        y.backward(gout)
        # the oracle restatement on the same weights: equal to the reference, value and gradients
        Wo = {f"m.{nm}": v.clone().requires_grad_(True) for nm, v in W.items()}
        ins_o = [t.detach().clone().requires_grad_(True) for t in ins]
        yo = O.rdn(ins_o, Wo, "m")
        yo.backward(gout)
        assert float((yo - y).abs().max()) == 0.0, tag
        for nm, prm in mod.named_parameters():
            assert float((Wo[f"m.{nm}"].grad - prm.grad).abs().max()) <= 1e-7 * max(1.0, float(prm.grad.abs().max())), (tag, nm)
        out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.gout"] = gout.numpy()
        for i, t in enumerate(ins):
            out[f"{tag}.in{i}"] = t.detach().numpy()
            out[f"{tag}.gin{i}"] = t.grad.numpy()
        names = [nm for nm, _ in mod.named_parameters()]
        out[f"{tag}.grad_norms"] = np.array([float(prm.grad.double().norm()) for _, prm in mod.named_parameters()])
        # full gradients of a spread of layers (first / middle / last dense block, the 1x1s, the up-sampler); the norms pin the rest
        keep = [nm for nm in names if nm.split(".")[0] in ("SFENet1", "GFF", "UPNet") or nm.startswith("RDBs.0.")
                or nm.startswith(f"RDBs.{D - 1}.")]
        for nm in keep:
            if dict(mod.named_parameters())[nm].numel() <= 20000:
                out[f"{tag}.grad.{nm}"] = dict(mod.named_parameters())[nm].grad.numpy()
        print(tag, "params", sum(p.numel() for p in mod.parameters()), "y", tuple(y.shape), "oracle == reference")
    path = os.path.join(HERE, "g10_rdn_shapes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")

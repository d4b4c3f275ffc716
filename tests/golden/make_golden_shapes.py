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

from shape_cases import CASES, LSTM_CASES       # noqa: E402
from bin_amd.weights import general_rdn_weights, general_lstm_weights  # noqa: E402
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
    # ---- ConvLSTM cells of other sizes (RDN.py:14-24): the reference module's outputs and autograd gradients
    for tag, (a, b, ks, n, h, w, with_state) in LSTM_CASES.items():
        Wg, Bg = (torch.from_numpy(v) for v in general_lstm_weights(0, a, b, ks))
        cell = REF.ConvLSTMCell(a, b, kernel_size=ks, padding=ks // 2)
        cell.load_state_dict({"Gates.weight": Wg, "Gates.bias": Bg}, strict=True)
        g = torch.Generator().manual_seed(200 + a + b)
        x = (torch.rand(n, a, h, w, generator=g) - 0.3).requires_grad_(True)
        state = [(torch.randn(n, b, h, w, generator=g) * 0.5).requires_grad_(True) for _ in range(2)] if with_state else None
        gh = torch.randn(n, b, h, w, generator=g) * 1e-2
        gc = torch.randn(n, b, h, w, generator=g) * 1e-2
        h1, (c1, h1b) = cell(x, state)
        assert h1b is h1
        ((h1 * gh).sum() + (c1 * gc).sum()).backward()
        xo = x.detach().clone().requires_grad_(True)
        so = [t.detach().clone().requires_grad_(True) for t in state] if with_state else None
        wo, bo = Wg.clone().requires_grad_(True), Bg.clone().requires_grad_(True)
        ho, (co, _) = O.convlstm_cell(xo, so, wo, bo)
        ((ho * gh).sum() + (co * gc).sum()).backward()
        assert float((ho - h1).abs().max()) == 0.0 and float((co - c1).abs().max()) == 0.0, tag
        assert float((wo.grad - cell.Gates.weight.grad).abs().max()) <= 1e-7, tag
        out[f"{tag}.x"], out[f"{tag}.gh"], out[f"{tag}.gc"] = x.detach().numpy(), gh.numpy(), gc.numpy()
        out[f"{tag}.h"], out[f"{tag}.c"] = h1.detach().numpy(), c1.detach().numpy()
        out[f"{tag}.gx"] = x.grad.numpy()
        out[f"{tag}.dw"], out[f"{tag}.db"] = cell.Gates.weight.grad.numpy(), cell.Gates.bias.grad.numpy()
        if with_state:
            out[f"{tag}.c0"], out[f"{tag}.h0"] = state[0].detach().numpy(), state[1].detach().numpy()
            out[f"{tag}.gc0"], out[f"{tag}.gh0"] = state[0].grad.numpy(), state[1].grad.numpy()
        print(tag, "oracle == reference")
    path = os.path.join(HERE, "g10_rdn_shapes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")

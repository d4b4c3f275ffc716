"""Diagnostic (GPU box, run by hand): per-parameter relative error of one RDN backward vs torch autograd of the oracle.
Lives under tests/ because it uses oracle/ (test infrastructure)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd.models.archs import RDN as A
from bin_amd.weights import rdn_param_shapes, canonical_weights
from oracle import rdn_oracle as O
canon = {k: torch.from_numpy(v) for k, v in canonical_weights(0).items()}
set_name, k = "model1", 2
gen = torch.Generator().manual_seed(11)
ins = [torch.rand(1, 3, 32, 48, generator=gen) for _ in range(k)]
gout = torch.randn(1, 3, 32, 48, generator=gen) * 1e-3
W = {f"{set_name}.{n}": canon[f"{set_name}.{n}"].clone().requires_grad_(True) for n in rdn_param_shapes(k)}
ins_cpu = [t.clone().requires_grad_(True) for t in ins]
O.rdn(ins_cpu, W, set_name).backward(gout)
for prec in ("f16", "f16x3"):
    mod = A.RDN_residual_interp_2_input(G0=96, D=12)
    mod.load_state_dict({n: canon[f"{set_name}.{n}"] for n in rdn_param_shapes(k)})
    mod = mod.cuda(); mod.precision = prec
    ins_gpu = [t.cuda().requires_grad_(True) for t in ins]
    mod(*ins_gpu).backward(gout.cuda())
    named = dict(mod.named_parameters())
    print(prec)
    for n in list(rdn_param_shapes(k))[::-1]:
        if 'weight' in n and ('convs.0' in n or 'LFF' in n or 'RDBs' not in n):
            r = W[f"{set_name}.{n}"].grad
            e = float((named[n].grad.cpu() - r).abs().max() / r.abs().max())
            print(f"  {n:40s} rel err {e:.2e}  |ref|max {float(r.abs().max()):.2e}")
    for a, b in zip(ins_gpu, ins_cpu):
        print("  input grad rel err %.2e" % float((a.grad.cpu() - b.grad).abs().max() / b.grad.abs().max()))

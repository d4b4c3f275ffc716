#!/usr/bin/env python3
"""Does a reduced-precision training mode CONVERGE like the fp32-class one?  (BASELINE config 4 says "bf16"; the
supported modes here are f16x3 and the mixed mode `backward_precision: f16`, and single-product `f16` training is gated because its
per-parameter gradients are only verified to 25 % on white-noise upstream gradients.)

Trains `bin_stage4` from the same seeded initialiser on the same stream of synthetic moving-texture clips (bin_amd/data/synthetic.py)
in each mode with `bin_model.optimize_parameters` and reports, per mode: the loss curve in windows, the loss of the FINAL weights
on held-out clips evaluated in f16x3 (so the comparison is of the weights, not of the evaluating arithmetic), the held-out PSNR of
the third-level I6 estimate, the step time, and the status word.

  python tools/train_convergence.py [--steps 1500] [--batch 8] [--size 128] [--modes f16x3,f16x3eps,mixed,f16] [--out gpurun_out/conv]
(`f16x3eps` = the f16x3 run again from weights perturbed by 1e-6 relative: how far two fp32-class runs drift apart by themselves.)
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bin_amd import ops  # noqa: E402
from bin_amd.data.synthetic import moving_texture_batch  # noqa: E402
from bin_amd.models import create_model  # noqa: E402
from bin_amd.weights import reference_state_dict  # noqa: E402


def make_model(mode, lr):
    tmp = tempfile.mkdtemp()
    prec, bwd = {"f16x3": ("f16x3", None), "f16x3eps": ("f16x3", None), "mixed": ("f16x3", "f16"), "f16": ("f16", None)}[mode]
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": prec, "backward_precision": bwd},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": tmp, "training_state": tmp},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": lr, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [10 ** 9],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    if mode == "f16x3eps":                               # noise floor of the comparison: the same run from weights perturbed by 1e-6 relative
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for prm in m.netG.module.parameters():
                prm.mul_(1.0 + 1e-6 * torch.randn(prm.shape, generator=g).to(prm.device))
    if mode == "f16":                                    # the gate of bin_amd/autograd.py, opened for this diagnostic
        for mod in m.netG.module.rdn_modules():
            mod.allow_f16_training = True
    return m


def evaluate(state_dict, clips):
    """Held-out loss and PSNR of a weight set, always evaluated in f16x3 through the wrapper's own loss."""
    m = make_model("f16x3", 1e-4)
    m.netG.module.load_state_dict(state_dict, strict=True)
    m.netG.eval()
    tot, psnr = 0.0, 0.0
    for d in clips:
        m.feed_data(d)
        with torch.no_grad():
            m.Ft_p = m.forward()
            loss, _ = m.get_loss(ret=1)
        tot += float(loss)
        gt = d["GTinp"][:, 2].cuda()
        psnr += float(-10.0 * torch.log10(((m.Ft_p[9].clamp(0, 1) - gt) ** 2).mean()))
    return tot / len(clips), psnr / len(clips)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--pool", type=int, default=150, help="distinct batches rendered; the run cycles through them")
    ap.add_argument("--modes", default="f16x3,f16x3eps,mixed,f16")
    ap.add_argument("--out", default="gpurun_out/train_convergence")
    a = ap.parse_args()
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    held_out = [moving_texture_batch(777, 4 * i, 4, a.size) for i in range(4)]
    # the clips are rendered once and shared by all modes (identical data, identical order)
    t0 = time.time()
    pool = min(a.steps, a.pool)          # host memory: a batch of 8 x 128^2 is 27 MB; the pool is cycled (epochs of `pool` batches)
    data = [moving_texture_batch(6, s * a.batch, a.batch, a.size) for s in range(pool)]
    print(f"rendered {pool} batches of {a.batch} x {a.size}^2 in {time.time() - t0:.0f} s", flush=True)
    res = {"steps": a.steps, "batch": a.batch, "size": a.size, "lr": a.lr, "pool": pool, "modes": {}}
    init_loss, init_psnr = evaluate(reference_state_dict(0), held_out)
    res["initialiser"] = {"heldout_loss": init_loss, "heldout_psnr_db": init_psnr}
    for mode in a.modes.split(","):
        m = make_model(mode, a.lr)
        losses = []
        status = "clean"
        torch.cuda.synchronize()
        t0 = time.time()
        for s in range(a.steps):
            m.feed_data(data[s % pool])
            m.optimize_parameters(s + 1)
            losses.append(m.loss.detach())
        torch.cuda.synchronize()
        dt = time.time() - t0
        try:
            ops.check_status()
        except RuntimeError as e:
            status = str(e)[:200]
        losses = [float(v) for v in losses]
        sd = {k: v.detach().clone() for k, v in m.netG.module.state_dict().items()}
        hl, hp = evaluate(sd, held_out)
        n = a.steps
        def win(lo, hi):
            lo, hi = max(0, lo), min(n, hi)
            return sum(losses[lo:hi]) / max(1, hi - lo)
        row = {"ms_per_step": dt / n * 1e3, "status": status,
               "loss_first10": win(0, 10), "loss_at_10pct": win(n // 10 - 5, n // 10 + 5), "loss_at_50pct": win(n // 2 - 10, n // 2 + 10),
               "loss_last50": win(n - 50, n), "heldout_loss_f16x3_eval": hl, "heldout_psnr_db": hp,
               "finite": all(v == v for v in losses), "curve_every_50": [round(win(i, i + 50), 6) for i in range(0, n, 50)]}
        res["modes"][mode] = row
        print(mode, json.dumps({k: v for k, v in row.items() if k != "curve_every_50"}), flush=True)
        del m
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out + ".json", "w"), indent=1)
    base = res["modes"].get("f16x3")
    with open(a.out + ".md", "w") as f:
        f.write(f"| mode | ms / step ({a.batch} x {a.size}^2) | loss, first 10 | at 10 % | at 50 % | last 50 | held-out loss (f16x3 eval) | vs f16x3 | held-out PSNR of I6 | status |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---|\n")
        f.write(f"| initialiser | | | | | | {init_loss:.5f} | | {init_psnr:.2f} dB | |\n")
        for mode, r in res["modes"].items():
            rel = f"{(r['heldout_loss_f16x3_eval'] / base['heldout_loss_f16x3_eval'] - 1) * 100:+.1f} %" if base else ""
            f.write(f"| {mode} | {r['ms_per_step']:.1f} | {r['loss_first10']:.4f} | {r['loss_at_10pct']:.4f} | {r['loss_at_50pct']:.4f} | "
                    f"{r['loss_last50']:.5f} | {r['heldout_loss_f16x3_eval']:.5f} | {rel} | {r['heldout_psnr_db']:.2f} dB | {r['status']} |\n")
    print(open(a.out + ".md").read())


if __name__ == "__main__":
    main()

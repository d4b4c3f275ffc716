#!/usr/bin/env python3
"""Measured fp16 headroom of the stored planes: the 720p window and a short synthetic training run.

  python tools/fp16_headroom.py [--steps 200] [--out gpurun_out/r03_fp16_headroom]
  python tools/fp16_headroom.py --checkpoint adobe_bin.pth --steps 0      # a holder of the trained weights: one command

--checkpoint PATH: load a reference checkpoint (the `.pth` of model_weights/download_adobe_bin.txt; `module.` / `InterpNet.`
prefixes are cleaned as the reference's load_network does, strict) instead of the seeded initialisation, for the 720p
window and as the starting point of the training part (--steps 0 skips that part).  The pretrained file is not available
in this build environment, so the committed tables are for the seeded weights; the exit code is 1 when any stored plane has
less than 8x headroom.

Part 1: ONE 6-frame 720p window (768x1344 padded, synthetic U[0,1) frames, seeded init) run through the training-forward
        path (four RDN calls with BINHIP_PLAN_KEEP_ACTS, forward values identical to inference): every stored activation
        tensor of every call.
Part 2: BASELINE config 3 (8 x 256x256 crops, Charbonnier, Adam 1e-4) for --steps optimisation steps on fixed synthetic
        data; activations AND gradient planes of steps 1, 50, 200 (weights that have really been updated).
Writes <out>.json (every row) and <out>.md (the summary table to commit under profiles/)."""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bin_amd import ops, range_stats as RS  # noqa: E402
from bin_amd.models import create_model  # noqa: E402
from bin_amd.models.archs.RDN import bin_stage4_lstm  # noqa: E402
from bin_amd.utils import util  # noqa: E402
from bin_amd.weights import reference_state_dict, synthetic_frames  # noqa: E402


TRAINED_LIKE = None          # --trained-like SEED: bin_amd.weights.trained_like_weights instead of the initialiser


def load_weights(net, checkpoint):
    """Seeded initialisation, or a reference checkpoint with the reference's key clean-up (base_model.py:93-102), strict."""
    if not checkpoint and TRAINED_LIKE is not None:
        from bin_amd.weights import state_dict_from_canonical, trained_like_weights
        net.load_state_dict(state_dict_from_canonical(trained_like_weights(TRAINED_LIKE)), strict=True)
        return f"trained-like synthetic distribution (bin_amd/weights.py trained_like_weights, seed {TRAINED_LIKE})"
    if not checkpoint:
        net.load_state_dict(reference_state_dict(0), strict=True)
        return "seeded initialisation (bin_amd/weights.py, seed 0)"
    from bin_amd.models.base_model import clean_state_dict_keys
    sd = torch.load(checkpoint, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd["state_dict"]
    net.load_state_dict(clean_state_dict_keys(sd, True), strict=True)
    return f"checkpoint {os.path.basename(checkpoint)}"


def window_720p(checkpoint=None):
    net = bin_stage4_lstm()
    load_weights(net, checkpoint)
    net = net.cuda().train()
    rec = RS.Recorder().attach(net)
    rec.armed, rec.tag = True, "720p "
    frames = [util.replicate_pad(f, util.pad_sizes(720, 1280)).cuda() for f in synthetic_frames(1234, 1, 720, 1280, 6)]
    with torch.enable_grad():
        out = net(*frames)                 # parameters require grad -> four-call training forward, hooks fire
    torch.cuda.synchronize()
    ops.check_status()
    del out
    rec.detach(net)
    return rec.rows


def training(steps, marks, batch=8, size=256, checkpoint=None):
    tmp = tempfile.mkdtemp()
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": "f16x3",
                         "backward_precision": None},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": tmp, "training_state": tmp},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    net = m.netG.module
    load_weights(net, checkpoint)
    g = torch.Generator().manual_seed(7)
    m.feed_data({"LQs": torch.rand(batch, 6, 3, size, size, generator=g),
                 "GTenh": torch.rand(batch, 6, 3, size, size, generator=g),
                 "GTinp": torch.rand(batch, 5, 3, size, size, generator=g)})
    rec = RS.Recorder().attach(net)
    out = {}
    losses = {}
    for step in range(1, steps + 1):
        rec.armed = step in marks
        rec.tag = f"step {step} "
        rec.rows = []
        m.optimize_parameters(step)
        if rec.armed:
            ops.check_status()
            out[step] = rec.rows
            losses[step] = float(m.loss.detach())
    rec.detach(net)
    return out, losses


def md_table(title, rows):
    lines = [f"### {title}", "",
             "| stored tensor class | tensors | largest abs value | headroom to 65504 | smallest non-zero abs value | share of non-zeros below 6.1e-5 (subnormal in the hi plane) |",
             "|---|---:|---:|---:|---:|---:|"]
    classes = {}
    for r in rows:
        classes.setdefault(r["class"], []).append(r)
    for key, rs in classes.items():
        s = RS.summarize(rs)
        lines.append(f"| {key} | {s['tensors']} | {s['amax']:.4g} | {s['min_headroom']:.3g}x | {s['min_nonzero']:.3g} | "
                     f"{100 * s['max_subnormal_share']:.3f} % |")
    s = RS.summarize(rows)
    lines += ["", f"Worst headroom: **{s['min_headroom']:.3g}x** ({s['worst_tensor']}); largest stored magnitude {s['amax']:.4g}.", ""]
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--marks", default="1,50,200")
    ap.add_argument("--out", default="gpurun_out/r03_fp16_headroom")
    ap.add_argument("--skip-720p", action="store_true")
    ap.add_argument("--checkpoint", default=None, help="reference .pth to measure instead of the seeded initialisation")
    ap.add_argument("--trained-like", type=int, default=None, metavar="SEED",
                    help="weights with a trained network's value distribution (three decades per layer, 30 %% zeros, one layer x50: "
                         "bin_amd.weights.trained_like_weights) instead of the initialiser")
    args = ap.parse_args()
    global TRAINED_LIKE
    TRAINED_LIKE = args.trained_like
    marks = {int(x) for x in args.marks.split(",") if int(x) <= args.steps}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    src = (f"checkpoint {os.path.basename(args.checkpoint)}" if args.checkpoint else
           f"trained-like synthetic distribution, seed {args.trained_like}" if args.trained_like is not None else "seeded initialisation")
    result, md = {}, [f"# fp16 headroom of the stored planes, measured (tools/fp16_headroom.py; weights: {src})", ""]
    if not args.skip_720p:
        rows = window_720p(args.checkpoint)
        result["window_720p"] = rows
        md.append(md_table(f"720p window (768x1344), {src}, forward activations of all 17 call-equivalents (4 batched calls)", rows))
    tr, losses = training(args.steps, marks, checkpoint=args.checkpoint) if args.steps > 0 else ({}, {})
    for step in sorted(tr):
        rows = tr[step]
        result[f"train_step_{step}"] = rows
        fwd = [r for r in rows if r["kind"] == "activation"]
        bwd = [r for r in rows if r["kind"] == "gradient"]
        md.append(md_table(f"training step {step} (8 x 256x256, loss {losses[step]:.5f}): activations", fwd))
        md.append(md_table(f"training step {step}: gradient planes (stored x the call's power-of-two scale; "
                           f"scales seen: {sorted({r['scale'] for r in bwd})})", bwd))
    json.dump(result, open(args.out + ".json", "w"))
    open(args.out + ".md", "w").write("\n".join(md))
    print("\n".join(md))
    worst = min((RS.summarize(rows)["min_headroom"] for rows in result.values() if rows), default=float("inf"))
    print(f"worst headroom over everything measured: {worst:.3g}x")
    return 0 if worst >= 8.0 else 1


if __name__ == "__main__":
    raise SystemExit(main())

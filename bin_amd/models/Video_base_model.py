"""Single-tensor wrapper API over the same generator (reference Video_base_model.py:22-335).

The reference class cannot be imported (it needs `CharbonnierLossPlusSSIM`, which models/loss.py does not define),
so no reference outputs exist for its training step: PARITY UNPINNED there.  What is kept is its surface — `feed_data({'LQs', 'GT'})`,
`optimize_parameters`, `test`, `test_stitch`, `get_current_log`, `get_loss`, `get_current_visuals`, `load`, `save` —
over a `var_L [B,N,C,H,W]` tensor.  The frames along N are handed to bin_stage4 as its six inputs and the 14
outputs come back stacked as `fake_H [B,14,C,H,W]`; `GT` is the matching [B,14,C,H,W] stack.

`test_stitch` is the tiled forward for frames that do not fit in one pass: the reference hard-codes a 960x540 ->
4K SR geometry (scale 4, 320x180 tiles, 32 px halo); here the tile size, halo and scale (1 for bin_stage4) are
arguments with those defaults' roles.  ITS geometry IS pinned (round 4): fixture g12_stitch is the reference's own
`test_stitch` (imported with the one missing loss name injected, tests/golden/make_golden_stitch.py) over a stand-in
single-tensor x4 generator; `test_stitch((180, 320), 32, 4)` over the same generator reproduces it bit for bit."""
import logging
import os.path as osp
from collections import OrderedDict

import torch
import torch.nn as nn

from . import lr_scheduler, networks
from .base_model import BaseModel, _direct_param_grads
from .bin_model import FlatGradAllReduce, SingleProcessParallel, _get
from .loss import CharbonnierLoss, L1SumLoss, L2SumLoss
from ..utils import util

logger = logging.getLogger("base")


class _CbPair(nn.Module):
    """Charbonnier criterion with the (loss, parts) return shape VideoBaseModel's callers unpack."""

    def __init__(self):
        super().__init__()
        self.cb = CharbonnierLoss()

    def forward(self, x, y):
        return self.cb(x, y), None


class VideoBaseModel(BaseModel):
    def __init__(self, opt, netG=None, cri_pix=None):
        super().__init__(opt)
        self.rank = torch.distributed.get_rank() if opt["dist"] else -1
        train_opt = opt["train"] if "train" in opt else None

        net = (netG if netG is not None else networks.define_G(opt)).to(self.device)
        self.netG = SingleProcessParallel(net)
        self.grad_sync = FlatGradAllReduce(self.netG.parameters()) if opt["dist"] else None
        self.print_network()
        self.load()
        if not self.is_train:
            return

        self.netG.train()
        kind = train_opt["pixel_criterion"]
        if cri_pix is not None:
            self.cri_pix = cri_pix
        elif kind == "l1":
            self.cri_pix = L1SumLoss().to(self.device)
        elif kind == "l2":
            self.cri_pix = L2SumLoss().to(self.device)
        elif kind == "cb":
            self.cri_pix = _CbPair().to(self.device)
        else:
            raise NotImplementedError("Loss type [{:s}] is not recognized.".format(kind))
        self.l_pix_w = train_opt["pixel_weight"]

        trainable = []
        for name, p in self.netG.named_parameters():
            if p.requires_grad:
                trainable.append((name, p))
            elif self.rank <= 0:
                logger.warning("Params [%s] will not optimize.", name)
        if _get(train_opt, "ft_tsa_only"):      # Video_base_model.py:61-83: normal parameters first, 'tsa_fusion' second
            params = [{"params": [p for n, p in trainable if "tsa_fusion" not in n], "lr": train_opt["lr_G"]},
                      {"params": [p for n, p in trainable if "tsa_fusion" in n], "lr": train_opt["lr_G"]}]
        else:
            params = [p for _, p in trainable]
        self.optimizer_G = torch.optim.Adam(params, lr=train_opt["lr_G"], weight_decay=_get(train_opt, "weight_decay_G", 0),
                                            betas=(train_opt["beta1"], train_opt["beta2"]))
        self.optimizers.append(self.optimizer_G)
        scheme = train_opt["lr_scheme"]
        if scheme == "MultiStepLR":
            sched = lr_scheduler.MultiStepLR_Restart(
                self.optimizer_G, train_opt["lr_steps"], restarts=_get(train_opt, "restarts"),
                weights=_get(train_opt, "restart_weights"), gamma=train_opt["lr_gamma"],
                clear_state=_get(train_opt, "clear_state", False))
        elif scheme == "CosineAnnealingLR_Restart":
            sched = lr_scheduler.CosineAnnealingLR_Restart(
                self.optimizer_G, train_opt["T_period"], eta_min=train_opt["eta_min"],
                restarts=_get(train_opt, "restarts"), weights=_get(train_opt, "restart_weights"))
        elif scheme == "ReduceLROnPlateau":
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer_G, "min", factor=train_opt["factor"],
                                                               patience=train_opt["patience"])
        else:
            raise NotImplementedError()
        self.schedulers.append(sched)
        self.log_dict = OrderedDict()

    # ------------------------------------------------------------------ data
    def feed_data(self, data, need_GT=True):
        self.var_L = data["LQs"].to(self.device)
        if need_GT:
            self.real_H = data["GT"].to(self.device)

    def _net(self, var_L):
        """[B,N,C,H,W] -> [B,14,C,H,W]: frames along N are bin_stage4's positional inputs.  A generator of the kind the
        reference class was written for (`takes_stacked_frames`: ONE [B,N,C,H,W] tensor in, one tensor out,
        Video_base_model.py:139) gets var_L as it is."""
        from .base_model import unwrap
        if getattr(unwrap(self.netG), "takes_stacked_frames", False):
            return self.netG(var_L)
        return torch.stack(self.netG(*var_L.unbind(dim=1)), dim=1)

    def _pix_loss(self):
        out = self.cri_pix(self.fake_H, self.real_H)
        return out if isinstance(out, tuple) else (out, None)

    # ------------------------------------------------------------------ training
    def set_params_lr_zero(self):
        self.optimizers[0].param_groups[0]["lr"] = 0

    def optimize_parameters(self, step):
        ft = _get(self.opt["train"], "ft_tsa_only")
        if ft and step < ft:
            self.set_params_lr_zero()
        self.optimizer_G.zero_grad()
        if self.grad_sync is not None:
            self.grad_sync.attach()
        self.fake_H = self._net(self.var_L)
        loss, parts = self._pix_loss()
        l_pix = self.l_pix_w * loss
        with _direct_param_grads(self.netG):       # weight gradients land in .grad straight from the kernels
            l_pix.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer_G.step()
        if parts is None:
            self.log_dict["l_pix"] = l_pix.item()
        else:
            self.log_dict["total_loss"] = l_pix.item()
            self.log_dict["l_pix"], self.log_dict["ssim_loss"] = parts[0].item(), parts[1].item()

    optimize_parameters_without_schudlue = optimize_parameters      # the reference's second spelling of the same step

    def get_loss(self):
        return self.l_pix_w * self._pix_loss()[0]

    def get_current_log(self):
        return self.log_dict

    # ------------------------------------------------------------------ inference
    def test(self):
        self.netG.eval()
        with torch.no_grad():
            self.fake_H = self._net(self.var_L)
        self.netG.train()

    def test_stitch(self, tile_hw=(256, 256), halo=32, scale=1):
        """Tiled forward: replicate-pad the frame to whole tiles, run every (tile + halo) crop, keep the tile
        interiors.  The halo is context only; the pyramid's receptive field is larger, so seams are
        approximate exactly as in the reference's stitcher."""
        self.netG.eval()
        th, tw = tile_hw
        with torch.no_grad():
            x = self.var_L
            B, N, C, H, W = x.shape
            ny, nx = -(-H // th), -(-W // tw)
            flat = x.reshape(B * N, C, H, W)
            flat = nn.functional.pad(flat, (0, nx * tw - W, 0, ny * th - H), mode="replicate")
            flat = nn.functional.pad(flat, (halo, halo, halo, halo), mode="replicate")
            xp = flat.reshape(B, N, C, ny * th + 2 * halo, nx * tw + 2 * halo)
            out = None
            for j in range(ny):
                for i in range(nx):
                    crop = xp[..., j * th:(j + 1) * th + 2 * halo, i * tw:(i + 1) * tw + 2 * halo]
                    y = self._net(crop.contiguous())          # [B,K,C,h',w'], or [B,C,h',w'] from a single-output generator
                    if out is None:
                        out = torch.zeros(tuple(y.shape[:-2]) + (ny * th * scale, nx * tw * scale), device=y.device)
                    out[..., j * th * scale:(j + 1) * th * scale, i * tw * scale:(i + 1) * tw * scale] = \
                        y[..., halo * scale:(halo + th) * scale, halo * scale:(halo + tw) * scale]
            self.fake_H = out[..., :H * scale, :W * scale]
        self.netG.train()

    def get_current_visuals(self, need_GT=True, save=False, name=None, save_path=None):
        vis = OrderedDict()
        vis["LQ"] = self.var_L.detach()[0].float().cpu()
        vis["rlt"] = self.fake_H.detach()[0].float().cpu()
        if need_GT:
            vis["GT"] = self.real_H.detach()[0].float().cpu()
        if save:
            util.save_img(util.tensor2img(vis["rlt"][-1]), osp.join(save_path, "{}.png".format(name)))
        return vis

    # ------------------------------------------------------------------ bookkeeping
    def print_network(self):
        s, n = self.get_network_description(self.netG)
        if self.rank <= 0:
            logger.info("Network G structure: %s - %s, with parameters: %s", type(self.netG).__name__,
                        type(self.netG.module).__name__, format(n, ",d"))
            logger.info(s)

    def load(self):
        path = self.opt["path"]["pretrain_model_G"]
        if path is not None:
            logger.info("Loading model for G [%s] ...", path)
            self.load_network(path, self.netG, self.opt["path"]["strict_load"])

    def save(self, iter_label):
        self.save_network(self.netG, "G", iter_label)

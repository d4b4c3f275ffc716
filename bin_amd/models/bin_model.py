"""bin_model — the drop-in wrapper API of the reference (models/bin_model.py:21-615) over the
MI355X-native generator: same method names, argument meaning, stored attributes (`Ft_p`, `loss`,
`loss_list`, `B1..B11`, `I1..I11`), the same 14-output / 17-term loss assembly and checkpoint format.

Differences, all on the parallelism side:
  * one process per GPU.  Non-dist: the generator is wrapped in `SingleProcessParallel`, which only
    provides the `.module` attribute DataParallel users expect (no per-call replicate/scatter/gather).
  * dist (`opt['dist']`): gradients are averaged across ranks with ONE flat all-reduce per step on the
    default process group (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) —
    45.77 MB fp32, the message DistributedDataParallel would send in two buckets.
Only nframes == 6 exists in the reference's factory (`bin_stage4`); other values raise.
"""
import logging
from collections import OrderedDict

import torch
import torch.nn as nn

from . import lr_scheduler
from . import networks
from .base_model import BaseModel, unwrap, _direct_param_grads
from .loss import CharbonnierLoss, L1SumLoss, L2SumLoss
from ..utils import dist_util, util
from ..utils.util import AverageMeter

logger = logging.getLogger("base")


def _get(d, key, default=None):
    """opt access that works for plain dicts and the reference's NoneDict."""
    try:
        v = d[key]
    except (KeyError, TypeError):
        return default
    return default if v is None else v


class SingleProcessParallel(nn.Module):
    """Stand-in for nn.DataParallel in the one-process-per-GPU design: exposes `.module`, forwards calls."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class FlatGradAllReduce:
    """Data-parallel gradient averaging over ONE flat buffer (C1 in SURVEY.md §2c): every parameter's .grad is a view
    into one flat fp32 buffer (11.44 M floats = 45.77 MB for bin_stage4), so autograd / the backward kernels accumulate
    straight into it and the all-reduce (RCCL over xGMI under "nccl", gloo in the CPU tests) runs on the buffer in
    place — no per-parameter gather/scatter copies (540 tensors each way).

    Overlap (SURVEY §8e "launched during backward in >= 2 chunks"): `watch(net)` registers the four RDN weight sets.
    The backward of the pyramid finishes them in the order model4, model3, model2, model1 (each set's gradient is
    complete when the LAST call that shares it has run its backward — bin_amd.autograd counts them); at that moment
    the set's contiguous slice of the flat buffer is all-reduced on a SIDE stream, behind an event recorded on the
    compute stream, while the remaining backward kernels keep running.  `__call__` (after backward) reduces whatever
    is left (ConvLSTM cells; everything, if nothing was watched), joins the side stream and divides by the world size.
    On CPU tensors (gloo tests) there are no streams: the same slices are reduced synchronously."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.force_collective = False      # run the collectives even at world size 1 (tests of the RCCL path)
        self._buckets = []                 # (module, start, end) slices whose completion is signalled during backward
        self._reduced = []                 # slices already all-reduced in this step
        self._stream = None

    def _offsets(self):
        o = 0
        for p in self.params:
            yield o
            o += p.numel()

    def watch(self, net):
        """Register every shared RDN weight set of `net` whose parameters form one contiguous slice of the buffer."""
        from .archs.RDN import _RDNBase
        off = {id(p): o for p, o in zip(self.params, self._offsets())}
        self._buckets = []
        seen = set()
        for mod in net.modules():
            if not isinstance(mod, _RDNBase) or id(mod) in seen:
                continue
            seen.add(id(mod))
            ps = [p for p in mod.parameters() if p.requires_grad]
            if not ps or any(id(p) not in off for p in ps):
                continue
            start = off[id(ps[0])]
            end = start
            ok = True
            for p in ps:
                ok = ok and off[id(p)] == end
                end += p.numel()
            if ok:
                idx = len(self._buckets)
                self._buckets.append((mod, start, end))
                mod._grads_ready_cb = (lambda i=idx: self._bucket_ready(i))
        return self

    def attach(self):
        """Zero the flat buffer and (re)point every .grad at its slice; call after optimizer.zero_grad()."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        else:
            self.flat.zero_()
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        self._reduced = []
        for mod, _, _ in self._buckets:
            mod._bwd_pending = 0

    def _active(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective)

    def _views_intact(self):
        return self.flat is not None and all(p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * o
                                             for p, o in zip(self.params, self._offsets()))

    def _reduce_slice(self, start, end, overlap):
        import torch.distributed as dist
        piece = self.flat[start:end]
        if overlap and piece.is_cuda:
            main = torch.cuda.current_stream(piece.device)
            if self._stream is None or self._stream.device != piece.device:
                self._stream = torch.cuda.Stream(device=piece.device)
            ev = torch.cuda.Event()
            ev.record(main)                          # everything that wrote this slice is queued before this point
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                dist_util.all_reduce(piece)
        else:
            dist_util.all_reduce(piece)
        self._reduced.append((start, end))

    def _bucket_ready(self, idx):
        """Called from the backward pass when weight set `idx` has received its last contribution."""
        if not self._active() or not self._views_intact():
            return
        _, start, end = self._buckets[idx]
        if (start, end) not in self._reduced:
            self._reduce_slice(start, end, overlap=True)

    def _join(self):
        """Order the caller's stream after everything queued on the side stream (the early bucket all-reduces)."""
        if self._stream is not None and self.flat is not None and self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)

    def __call__(self):
        import torch.distributed as dist
        if not self._active():
            return
        world = dist.get_world_size()
        if not self._views_intact():
            # Somebody replaced a .grad (or attach() was skipped).  The early all-reduces may still be in flight on the
            # side stream: join it BEFORE touching the buffer, then copy the strays in and take THEIR ranges out of the
            # already-reduced set (a slice summed before its stray arrived holds stale data for that parameter; the rest
            # of the slice is already a sum over ranks and must not be reduced a second time).
            self._join()
            self._reduced = _subtract(self._reduced, self._gather())
        # the remainder: maximal runs of the buffer not covered by an early bucket
        pos = 0
        for start, end in sorted(self._reduced) + [(self.numel, self.numel)]:
            if start > pos:
                self._reduce_slice(pos, start, overlap=False)
            pos = max(pos, end)
        self._join()
        self._reduced = []
        if world > 1:
            self.flat.div_(world)

    def _gather(self):
        """Re-point every stray .grad at its slice of the flat buffer (copying its values in); returns the strays' ranges."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        base = self.flat.data_ptr()
        strays = []
        for p, o in zip(self.params, self._offsets()):
            view = self.flat[o:o + p.numel()].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() == base + 4 * o:
                continue
            else:
                view.copy_(p.grad.reshape(p.shape))
            p.grad = view
            strays.append((o, o + p.numel()))
        return strays


def _subtract(ranges, holes):
    """Intervals of `ranges` not covered by any of `holes` (half-open [start, end) pairs)."""
    out = []
    for s, e in sorted(ranges):
        pos = s
        for hs, he in sorted(holes):
            if he <= pos or hs >= e:
                continue
            if hs > pos:
                out.append((pos, hs))
            pos = max(pos, he)
        if pos < e:
            out.append((pos, e))
    return out


class bin_model(BaseModel):
    """The model for Blurry Video Frame Interpolation (reference bin_model.py:21)."""

    def __init__(self, opt, netG=None, cri_pix=None):
        """`netG` / `cri_pix` (bin_amd extension, used by the CPU host-logic tests): inject a pre-built
        generator / criterion instead of define_G(opt) / the HIP CharbonnierLoss.  The product objects have
        no CPU path and raise on CPU tensors."""
        super().__init__(opt)
        self.nframes = int(opt["network_G"]["nframes"])
        self.version = int(opt["network_G"]["version"])
        if self.nframes != 6:
            raise NotImplementedError("bin_amd: only nframes == 6 (bin_stage4) exists in the reference factory")
        if opt["dist"]:
            self.rank = torch.distributed.get_rank()
        else:
            self.rank = -1
        train_opt = opt["train"] if "train" in opt else None

        self.netG = (netG if netG is not None else networks.define_G(opt)).to(self.device)
        self.netG = SingleProcessParallel(self.netG)
        self.grad_sync = FlatGradAllReduce(self.netG.parameters()).watch(self.netG) if opt["dist"] else None
        if opt["dist"]:
            self.broadcast_parameters()

        self.print_network()
        self.load()

        if self.is_train:
            self.netG.train()
            self.loss_type = loss_type = train_opt["pixel_criterion"]
            if loss_type == "l1":
                self.cri_pix = L1SumLoss().to(self.device)        # nn.L1Loss(reduction='sum') in the reference
            elif loss_type == "l2":
                self.cri_pix = L2SumLoss().to(self.device)        # nn.MSELoss(reduction='sum')
            elif loss_type == "cb":
                self.cri_pix = CharbonnierLoss().to(self.device)
            else:
                raise NotImplementedError("Loss type [{:s}] is not recognized.".format(loss_type))
            if cri_pix is not None:
                self.cri_pix = cri_pix
            self.l_pix_w = train_opt["pixel_weight"]

            wd_G = _get(train_opt, "weight_decay_G", 0)
            trainable = []
            for k, v in self.netG.named_parameters():
                if v.requires_grad:
                    trainable.append((k, v))
                elif self.rank <= 0:
                    logger.warning("Params [{:s}] will not optimize.".format(k))
            if _get(train_opt, "ft_tsa_only"):
                # bin_model.py:66-87: two groups — everything first, then the 'tsa_fusion' parameters (bin_stage4 has none, so
                # the second group is empty): set_params_lr_zero() freezes group 0, and `.state` files carry both groups
                optim_params = [{"params": [v for k, v in trainable if "tsa_fusion" not in k], "lr": train_opt["lr_G"]},
                                {"params": [v for k, v in trainable if "tsa_fusion" in k], "lr": train_opt["lr_G"]}]
            else:
                optim_params = [v for _, v in trainable]
            self.optimizer_G = torch.optim.Adam(optim_params, lr=train_opt["lr_G"], weight_decay=wd_G,
                                                betas=(train_opt["beta1"], train_opt["beta2"]))
            self.optimizers.append(self.optimizer_G)

            scheme = train_opt["lr_scheme"]
            if scheme == "MultiStepLR":
                for optimizer in self.optimizers:
                    self.schedulers.append(lr_scheduler.MultiStepLR_Restart(
                        optimizer, train_opt["lr_steps"], restarts=_get(train_opt, "restarts"),
                        weights=_get(train_opt, "restart_weights"), gamma=train_opt["lr_gamma"],
                        clear_state=_get(train_opt, "clear_state", False)))
            elif scheme == "CosineAnnealingLR_Restart":
                for optimizer in self.optimizers:
                    self.schedulers.append(lr_scheduler.CosineAnnealingLR_Restart(
                        optimizer, train_opt["T_period"], eta_min=train_opt["eta_min"],
                        restarts=_get(train_opt, "restarts"), weights=_get(train_opt, "restart_weights")))
            elif scheme == "ReduceLROnPlateau":
                for optimizer in self.optimizers:
                    self.schedulers.append(torch.optim.lr_scheduler.ReduceLROnPlateau(
                        optimizer, "min", factor=train_opt["factor"], patience=train_opt["patience"]))
            else:
                raise NotImplementedError()
            self.avg_log_dict = OrderedDict()
            self.inst_log_dict = OrderedDict()

    # ------------------------------------------------------------------ distributed helpers
    def broadcast_parameters(self, force=False):
        """Rank 0's parameters to every rank as ONE bucketed broadcast (a flat copy of the 11.44 M floats) instead of
        540 small collectives; the relayouted kernel weights are invalidated explicitly because writing through
        `.data`-style views does not bump the parameters' version counters."""
        import torch.distributed as dist
        if dist.get_world_size() == 1 and not force:
            return
        params = list(self.netG.parameters())
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in params])
            dist_util.broadcast(flat, src=0)
            o = 0
            for p in params:
                p.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        for mod in self.netG.modules():
            if hasattr(mod, "invalidate_kernel_weights"):
                mod.invalidate_kernel_weights()

    _broadcast_parameters = broadcast_parameters

    # ------------------------------------------------------------------ training step (bin_model.py:130-141)
    def optimize_parameters(self, step):
        ft = _get(self.opt["train"], "ft_tsa_only")
        if ft and step < ft:
            self.set_params_lr_zero()
        self.optimizer_G.zero_grad()
        if self.grad_sync is not None:
            self.grad_sync.attach()
        self.Ft_p = self.forward()
        self.loss, self.loss_list = self.get_loss(ret=1)
        l_pix = self.l_pix_w * self.loss
        with _direct_param_grads(self.netG):       # weight gradients land in .grad straight from the kernels
            l_pix.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer_G.step()

    def set_params_lr_zero(self):
        self.optimizers[0].param_groups[0]["lr"] = 0

    def test_sharp_forward(self):
        """bin_model.py:382-393 interpolates from the SHARP frames for nframes 1 / 3 / 4 / 5 only; with the one generator
        the factory builds (nframes == 6) no branch is taken and `Ft_p` is left as it was — same here."""
        return None

    # ------------------------------------------------------------------ data staging (bin_model.py:147-274)
    def feed_data(self, trainData, need_GT=True):
        LQs, GTenh, GTinp = trainData["LQs"], trainData["GTenh"], trainData["GTinp"]   # B N C H W
        for i, nm in enumerate(("B1", "B3", "B5", "B7", "B9", "B11")):
            setattr(self, nm, LQs[:, i, ...].to(self.device))
        for i, nm in enumerate(("I1", "I3", "I5", "I7", "I9", "I11")):
            setattr(self, nm, GTenh[:, i, ...].to(self.device))
        for i, nm in enumerate(("I2", "I4", "I6", "I8", "I10")):
            setattr(self, nm, GTinp[:, i, ...].to(self.device))
        self.batch, self.channel, self.height, self.width = (self.I1.size(k) for k in range(4))

    def test_set_input(self, testData):
        B1, B3, B5, B7, B9, B11, _ = testData
        self.B1, self.B3, self.B5 = B1.to(self.device), B3.to(self.device), B5.to(self.device)
        self.B7, self.B9, self.B11 = B7.to(self.device), B9.to(self.device), B11.to(self.device)
        self.batch, self.channel, self.height, self.width = (self.B1.size(k) for k in range(4))

    # ------------------------------------------------------------------ forward variants
    def test(self):
        self.netG.eval()
        with torch.no_grad():
            Ft_p = self.netG(self.B1, self.B3, self.B5, self.B7, self.B9, self.B11)
        self.netG.train()
        self.Ft_p = Ft_p
        return Ft_p

    def forward(self):
        Ft_p = self.netG(self.B1, self.B3, self.B5, self.B7, self.B9, self.B11)
        self.Ft_p = Ft_p
        return Ft_p

    def test_forward(self):
        self.Ft_p = self.netG(self.B1, self.B3, self.B5, self.B7, self.B9, self.B11)

    def reset_state(self):
        self.netG.prev_state = None
        self.netG.hidden_state = None

    # ------------------------------------------------------------------ loss (bin_model.py:395-425)
    def get_loss(self, ret=0):
        # reference bin_model.py:395-425: 14 terms against the sharp frames, for nframes 6 / version 2 three cycle terms between
        # outputs, loss = sum(terms) / len(terms); the returned list is trimmed to the 14.  All terms and the mean are ONE
        # autograd node on the device (models/loss.py::multi_term_loss) when the criterion is one of the product's.
        from .loss import multi_term_loss
        num, gt_list = self.get_info(mode=1)
        assert num == len(gt_list)
        pairs = [(self.Ft_p[idx], gt) for idx, gt in enumerate(gt_list)]
        if self.nframes == 6 and self.version == 2:
            pairs += [(self.Ft_p[1], self.Ft_p[7]), (self.Ft_p[5], self.Ft_p[9]), (self.Ft_p[2], self.Ft_p[8])]
        loss, loss_list = multi_term_loss(self.cri_pix, pairs)
        loss_list = loss_list[:num]
        if ret == 1:
            return loss, loss_list
        self.loss = loss
        self.loss_list = loss_list

    def get_info(self, mode=0):
        num = 14
        if mode == 0:
            return num
        gt_list = [self.I2, self.I4, self.I6, self.I8, self.I3, self.I5, self.I7, self.I4, self.I6,
                   self.I5, self.I10, self.I9, self.I8, self.I7]
        if mode == 1:
            return num, gt_list
        return num, gt_list, [self.B1, self.B3, self.B5, self.B7, self.B9, self.B11]

    # ------------------------------------------------------------------ logging / meters
    def get_current_log(self, mode="train"):
        num = self.get_info()
        self.avg_log_dict, self.avg_psnr_dict, self.inst_log_dict = OrderedDict(), OrderedDict(), OrderedDict()
        if mode == "train":
            for i in range(num):
                self.avg_log_dict[str(i)] = self.train_loss_total[i].avg
                self.inst_log_dict[str(i)] = self.loss_list[i].item()
            self.avg_log_dict["Al"] = self.train_loss_total[-1].avg
            return self.inst_log_dict, self.avg_log_dict
        if mode == "val":
            psnr_total_avg = ssim_total_avg = 0
            for i in range(num):
                self.avg_log_dict["Al" + str(i)] = self.val_loss_total[i].avg
                self.avg_psnr_dict["Ap" + str(i)] = self.psnr_interp[i].avg
                psnr_total_avg += self.psnr_interp[i].avg
                ssim_total_avg += self.ssim_interp[i].avg
            self.avg_log_dict["Al"] = self.val_loss_total[-1].avg
            self.avg_psnr_dict["Ap"] = psnr_total_avg / num
            return (self.avg_log_dict, self.avg_psnr_dict, psnr_total_avg / num, ssim_total_avg / num,
                    self.val_loss_total[-1].avg)

    def get_current_visuals(self, need_GT=True):
        num, gt_list, lq_list = self.get_info(mode=2)
        out = OrderedDict()
        out["LQ"] = [d.detach()[0].float().cpu() for d in lq_list]
        out["rlt"] = [self.Ft_p[i].detach()[0].float().cpu() for i in range(num)]
        if need_GT:
            out["GT"] = [d.detach()[0].float().cpu() for d in gt_list]
        return out

    def train_AverageMeter(self):
        self.train_loss_total = [AverageMeter() for _ in range(self.get_info() + 1)]

    def train_AverageMeter_update(self):
        # the .item() reads below synchronise anyway: also surface a saturated fp16 plane (include/binhip.h, "Dynamic
        # range") as an error here instead of training on silently clamped activations / gradients
        from .. import ops
        if self.loss.is_cuda:
            ops.check_status(self.loss.device)
        num = len(self.loss_list)
        for i in range(num):
            self.train_loss_total[i].update(self.loss_list[i].item(), 1)
        self.train_loss_total[num].update(self.loss.item(), 1)

    def train_AverageMeter_reset(self):
        for m in self.train_loss_total:
            m.reset()

    def val_loss_AverageMeter(self):
        self.val_loss_total = [AverageMeter() for _ in range(self.get_info() + 1)]

    def val_loss_AverageMeter_update(self, loss_list, avg_loss):
        from .. import ops
        if avg_loss.is_cuda:
            ops.check_status(avg_loss.device)
        num = len(loss_list)
        for i in range(num):
            self.val_loss_total[i].update(loss_list[i].item(), 1)
        self.val_loss_total[num].update(avg_loss.item(), 1)

    def val_loss_AverageMeter_reset(self):
        for m in self.val_loss_total:
            m.reset()

    def val_AverageMeter_para(self):
        num = self.get_info()
        self.psnr_interp = [AverageMeter() for _ in range(num)]
        self.ssim_interp = [AverageMeter() for _ in range(num)]

    def val_AverageMeter_para_update(self, psnr_interp_t, ssim_interp_t):
        for i in range(len(self.psnr_interp)):
            self.psnr_interp[i].update(psnr_interp_t[i], 1)
            self.ssim_interp[i].update(ssim_interp_t[i], 1)

    def val_AverageMeter_para_reset(self):
        for a, b in zip(self.psnr_interp, self.ssim_interp):
            a.reset()
            b.reset()

    def compute_current_psnr_ssim(self, save=False, name=None, save_path=None):
        """PSNR / SSIM of the 14 outputs vs GT through tensor2img (bin_model.py:564-589); save=True also writes
        `rlt_<name>_<i>.png` / `gt_<name>_<i>.png` under save_path."""
        import os.path as osp
        num = self.get_info()
        visuals = self.get_current_visuals()
        psnr, ssim = [], []
        for i in range(num):
            rlt_img = util.tensor2img(visuals["rlt"][i])
            gt_img = util.tensor2img(visuals["GT"][i])
            psnr.append(util.calculate_psnr(rlt_img, gt_img))
            ssim.append(util.calculate_ssim(rlt_img, gt_img))
            if save:
                util.save_img(rlt_img, osp.join(save_path, "rlt_{}_{}.png".format(name, i)))
                util.save_img(gt_img, osp.join(save_path, "gt_{}_{}.png".format(name, i)))
        return psnr, ssim

    # ------------------------------------------------------------------ IO
    def print_network(self):
        s, n = self.get_network_description(self.netG)
        net_struc_str = "{} - {}".format(self.netG.__class__.__name__, unwrap(self.netG).__class__.__name__)
        if self.rank <= 0:
            logger.info("Network G structure: {}, with parameters: {:,d}".format(net_struc_str, n))
            logger.info(s)

    def load(self):
        load_path_G = _get(self.opt["path"], "pretrain_model_G")
        if load_path_G is not None:
            logger.info("Loading model for G [{:s}] ...".format(load_path_G))
            self.load_network(load_path_G, self.netG, _get(self.opt["path"], "strict_load", True))

    def save(self, iter_label):
        self.save_network(self.netG, "G", iter_label)

    @staticmethod
    def get_lr(optimizer):
        for param_group in optimizer.param_groups:
            return param_group["lr"]

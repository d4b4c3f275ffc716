"""The stand-in generator, options and batch of fixture g13_videobase_step: shared by its generator
(tests/golden/make_golden_videobase_step.py, which drives the REFERENCE's VideoBaseModel with them) and by the tests (which drive
bin_amd's).

The reference class (Video_base_model.py:22-187) hands its generator ONE tensor, `netG(var_L [B,N,C,H,W])`, and compares the result
with `real_H` — `bin_stage4` takes six tensors, so the reference wrapper can only ever train a single-tensor generator.  `StubVSR`
is one: a 3x3 conv over the stacked frames, a ReLU, and a 1x1 "fusion" whose parameters carry the name the wrapper's
`ft_tsa_only` option looks for (`tsa_fusion`, Video_base_model.py:61-83), so both parameter groups exist."""
import numpy as np
import torch
import torch.nn as nn

B, N, K, C, H, W = 2, 6, 14, 3, 16, 24
STEPS = 4
FT_TSA_ONLY = 3                 # steps 1, 2 run with group 0's rate forced to 0 (the option's meaning), steps 3, 4 train everything


class StubVSR(nn.Module):
    takes_stacked_frames = True          # bin_amd's wrapper hands such a generator var_L as ONE tensor, like the reference does

    def __init__(self):
        super().__init__()
        self.feat = nn.Conv2d(N * C, 16, 3, padding=1)
        self.tsa_fusion = nn.Conv2d(16, K * C, 1)
        g = np.random.Generator(np.random.PCG64(1305))
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.from_numpy(g.uniform(-0.2, 0.2, tuple(p.shape)).astype(np.float32)))

    def forward(self, x):                                      # [B, N, C, h, w] -> [B, K, C, h, w]
        b, n, c, h, w = x.shape
        y = self.tsa_fusion(torch.relu(self.feat(x.reshape(b, n * c, h, w))))
        return y.reshape(b, K, c, h, w)


def batch():
    g = np.random.Generator(np.random.PCG64(1306))
    return {"LQs": torch.from_numpy(g.random((B, N, C, H, W), dtype=np.float32)),
            "GT": torch.from_numpy(g.random((B, K, C, H, W), dtype=np.float32))}


def opt(tmp, ft_tsa_only=None, criterion="cb", pixel_weight=0.5):
    """The option keys VideoBaseModel.__init__ reads (Video_base_model.py:24-122)."""
    return {"model": "video_base", "gpu_ids": None, "is_train": True, "dist": False,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp), "training_state": str(tmp)},
            "train": {"pixel_criterion": criterion, "pixel_weight": pixel_weight, "weight_decay_G": 0, "ft_tsa_only": ft_tsa_only,
                      "lr_G": 2e-3, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [2, 3],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


CASES = {                       # name -> (ft_tsa_only, criterion, which step method, (loss, None)-returning criterion?)
    "cb_pair": (None, "cb", "optimize_parameters", True),                     # the method unpacks `loss, loss_tmp = cri_pix(...)`
    "cb_pair_ft": (FT_TSA_ONLY, "cb", "optimize_parameters", True),
    "cb_plain_noschedule": (None, "cb", "optimize_parameters_without_schudlue", False),
    "l1_plain_noschedule_ft": (FT_TSA_ONLY, "l1", "optimize_parameters_without_schudlue", False),
    "l2_plain_noschedule": (None, "l2", "optimize_parameters_without_schudlue", False),
}


class PairCriterion(nn.Module):
    """(loss, None): the return shape Video_base_model.py:141 unpacks (its own 'cb+ssim' class, the only one with that shape,
    does not exist in models/loss.py)."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x, y):
        return self.inner(x, y), None

"""Test helper (CPU): a generator with the PRODUCT module's parameters / state_dict but the ORACLE's
forward (plain torch ops, differentiable), so the host logic of bin_model (loss assembly, optimizer
step, gradient all-reduce, checkpoint IO) can be exercised on CPU.  Lives under tests/ because only
tests may touch oracle/."""
import torch

from bin_amd.models.archs.RDN import bin_stage4_lstm
from oracle import rdn_oracle as O


class OracleNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.inner = bin_stage4_lstm()          # parameter container with the reference's key names

    # expose the inner module's state_dict / parameter namespace unchanged
    def state_dict(self, *a, **k):
        return self.inner.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.inner.load_state_dict(sd, strict=strict)

    def named_parameters(self, *a, **k):
        return self.inner.named_parameters(*a, **k)

    def parameters(self, recurse=True):
        return self.inner.parameters(recurse)

    def forward(self, *frames):
        # named_parameters() lists each shared module once, under its FIRST alias (model1_1, model2_1, model3_1,
        # model4_1) — exactly the names canon_from_state_dict maps to the canonical weight sets
        W = O.canon_from_state_dict(dict(self.inner.named_parameters()))
        return O.bin_stage4_forward(list(frames), W)

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

    # expose the inner module's state_dict namespace unchanged
    def state_dict(self, *a, **k):
        return self.inner.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.inner.load_state_dict(sd, strict=strict)

    def named_parameters(self, *a, **k):
        return self.inner.named_parameters(*a, **k)

    def parameters(self, recurse=True):
        return self.inner.parameters(recurse)

    def forward(self, *frames):
        sd = dict(self.inner.named_parameters())          # de-duplicated (540) but only first aliases
        full = {}
        for k, v in sd.items():
            full[k] = v
        W = O.canon_from_state_dict(_expand_aliases(full))
        return O.bin_stage4_forward(list(frames), W)


def _expand_aliases(named):
    """named_parameters() lists shared modules once (model1_1, model2_1, model3_1, model4_1) — exactly
    the first alias of each set, which is what canon_from_state_dict keys on."""
    return named

"""The LR-schedule scenarios of fixture g6b_lr: shared by its generator (tests/golden/make_golden_lr.py, which runs them
on the REFERENCE's classes) and by tests/test_cpu_host.py (which runs them on bin_amd's)."""
import numpy as np
import torch

CASES = {
    "multistep_warm": ("multistep", dict(milestones=[4, 9, 9, 21], restarts=[14], weights=[0.3], gamma=0.1,
                                         clear_state=True), 6),
    "multistep_unsorted": ("multistep", dict(milestones=[3, 17, 26], restarts=[20, 8], weights=[0.25, 2.0], gamma=0.5,
                                             clear_state=False), 0),
    "cosine_warm": ("cosine", dict(T_period=[8, 12, 6], restarts=[8, 20], weights=[0.7, 1.5], eta_min=3e-7), 5),
    "cosine_long": ("cosine", dict(T_period=[5, 5], restarts=[24], weights=[1.0], eta_min=1e-7), 0),
    # somebody rescales the rates while the cosine sits on its floor (iterations 16 and 24 are bottoms of these cycles):
    # the reference leaves the floor by an INCREMENT on the rate the group has (lr_scheduler.py:57-61)
    "cosine_bottom_rescale": ("cosine", dict(T_period=[16, 8], restarts=[19], weights=[0.5], eta_min=1e-7), 0,
                              {16: 0.1, 28: 3.0}),
}
STEPS, RESUME_AT = 32, 11


def unpack(case):
    """(kind, kwargs, warm-up iterations, {iteration: factor applied to every group's rate after that iteration})."""
    return case if len(case) == 4 else (*case, {})


def drive(mod, kind, kw, warm, resume, rescale=None):
    """The reference loop: optimizer.step(); scheduler.step(); warm-up override (base_model.py:76-87)."""
    def make():
        ps = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        o = torch.optim.Adam([{"params": [ps[0]], "lr": 2e-4}, {"params": [ps[1]], "lr": 5e-5}])
        s = (mod.MultiStepLR_Restart if kind == "multistep" else mod.CosineAnnealingLR_Restart)(o, **kw)
        return o, s
    o, s = make()
    out = []
    for it in range(1, STEPS + 1):
        o.step()
        s.step()
        if it < warm:
            for g in o.param_groups:
                g["lr"] = g["initial_lr"] / warm * it
        if rescale and it in rescale:
            for g in o.param_groups:
                g["lr"] *= rescale[it]
        out.append([g["lr"] for g in o.param_groups])
        if resume and it == RESUME_AT:
            so, ss = o.state_dict(), s.state_dict()
            o, s = make()
            o.load_state_dict(so)
            s.load_state_dict(ss)
    return np.array(out, dtype=np.float64)


#!/usr/bin/env python3
"""G6b: LR curves of the reference's restartable schedulers under the conditions the training loop creates
(reference models/lr_scheduler.py:10-66 driven the way base_model.py:60-87 drives them): two parameter groups, a
warm-up that overrides the rates between scheduler steps, gamma = 0.1, restarts listed out of order, state cleared at a
restart, and a state_dict round trip in the middle of a curve.  Build container only (imports /root/reference).

Run:  python tests/golden/make_golden_lr.py     (writes tests/golden/g6b_lr.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import models.lr_scheduler as REF_LRS          # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
from lr_cases import CASES, drive, unpack              # noqa: E402


if __name__ == "__main__":
    curves = {}
    for tag, case in CASES.items():
        kind, kw, warm, rescale = unpack(case)
        a = drive(REF_LRS, kind, kw, warm, False, rescale)
        b = drive(REF_LRS, kind, kw, warm, True, rescale)
        assert np.array_equal(a, b), tag                  # the reference's own state_dict round trip is exact
        curves[tag] = a
    np.savez(os.path.join(HERE, "g6b_lr.npz"), **curves)
    print("wrote g6b_lr.npz", {k: v.shape for k, v in curves.items()})

"""RDN configurations other than bin_stage4's used by fixture g10_rdn_shapes (tests/golden/make_golden_shapes.py runs them on
the reference's classes, tests/test_gpu_net.py on bin_amd's): tag -> (input frames, (G0, D, C, G), N, H, W)."""
CASES = {
    "rdn2_default_args": (2, (64, 6, 4, 32), 1, 32, 48),      # the classes' own default arguments (RDN.py:169-172)
    "rdn3_wide_growth": (3, (32, 2, 3, 64), 2, 16, 32),       # G = 64, odd C, G0 = 32
    "rdn5_one_block": (5, (128, 1, 2, 32), 1, 16, 16),        # one block, G0 = 128
}

# ConvLSTM cells other than bin_stage4's (3, 3): tag -> (input_size, hidden_size, kernel_size, N, H, W, with previous state)
LSTM_CASES = {
    "lstm_5_7_k3_state": (5, 7, 3, 2, 12, 20, True),
    "lstm_3_16_k5_nostate": (3, 16, 5, 1, 16, 16, False),
    "lstm_20_4_k1_state": (20, 4, 1, 1, 8, 24, True),
}

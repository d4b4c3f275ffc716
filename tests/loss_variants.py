"""Variants of the reference wrapper's loss that g11_loss_variants pins (shared by the generator and the tests):
(tag, network_G.version, train.pixel_criterion, train.pixel_weight)."""
VARIANTS = (
    ("v1_cb", 1, "cb", 1.0),          # no cycle terms: loss = sum of the 14 Charbonnier terms / 14 (bin_model.py:395-403)
    ("v2_l1", 2, "l1", 1.0),          # nn.L1Loss(reduction='sum') incl. the three cycle terms, / 17
    ("v2_l2", 2, "l2", 1.0),          # nn.MSELoss(reduction='sum')
    ("v1_l2", 1, "l2", 1.0),
    ("v2_cb_w", 2, "cb", 0.25),       # pixel_weight scales the loss that is differentiated (bin_model.py:137-138)
)
SAMPLE = ("model.model4_1.UPNet.2.weight", "model.model3_1.RDBs.7.LFF.bias")

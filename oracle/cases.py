"""Seeded test cases shared by the fixture generator and the tests.  TEST INFRASTRUCTURE ONLY.
Inputs are rebuilt from hash tags (oracle/ref_models.hash_uniform), so fixtures only store outputs."""
import torch

from oracle import ref_models as RM


def loss_cases():
    """(name, C, N, side, kwargs) -- labels/logits come from hash tags so tests can rebuild them."""
    return [
        ("c4_n2_s8", 4, 2, 8, {}),
        ("c3_n1_s8", 3, 1, 8, {}),
        ("c4_n4_s64", 4, 4, 64, {}),
        ("c3_n4_s64", 3, 4, 64, {}),
        ("c4_n4_s64_allbg", 4, 4, 64, {"allbg": True}),
        ("c4_n3_s16_allchan", 4, 3, 16, {"is_all_chann": True}),
        ("c4_n2_s16_nosft", 4, 2, 16, {"no_sft": True}),
        ("c4_n2_s16_g0", 4, 2, 16, {"gamma": 0.0}),
        ("c4_n2_s16_g1p5", 4, 2, 16, {"gamma": 1.5}),
        ("c4_n2_s16_alpha", 4, 2, 16, {"alpha": [0.2, 1.0, 2.0, 0.5]}),
        ("c4_n2_s16_sum", 4, 2, 16, {"reduction": "sum"}),
        ("c4_n1_s384", 4, 1, 384, {}),
    ]


def make_loss_inputs(name, C, N, side, kw):
    logits = RM.hash_uniform(f"loss:{name}:z", (N, C, side, side), -3.0, 3.0)
    _, labels = RM.synthetic_batch(N, side, C, tag=f"loss:{name}")
    if kw.get("allbg"):
        labels = torch.zeros_like(labels)
    return logits, labels



"""The fixed "model" of the inference-pipeline golden (make_golden_r3_pipeline.py) and of the test that replays it on the GPU."""
import torch

MESH = (200, 150, 41)          # n_u, n_v, seed -> 30 000 vertices (>= 24 000: the open3d subdivision branch is not taken)


def fixed_model(inputs):
    """{'cls_pred': (1, 17, N)}: class = floor(4 * (7x + 3y + 5z)) mod 17 as a one-hot -- float32 multiply / add / floor / remainder
    only, each exactly rounded, so that the CPU run of the reference and the GPU run of the drop-in see the same logits."""
    x = inputs[0]
    s = x[:, 0] * 7.0
    s = s + x[:, 1] * 3.0
    s = s + x[:, 2] * 5.0
    cls = torch.remainder(torch.floor(s * 4.0), 17.0).long()                   # (1, N)
    return {"cls_pred": torch.nn.functional.one_hot(cls, 17).permute(0, 2, 1).float()}

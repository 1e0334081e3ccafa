"""Loading a checkpoint trained at 224 px into a model built for a higher resolution (reference
network_utils/finetune_state_dict.py:10-65, used by the 280 / 392 px fine-tuning scripts): every positional embedding --
the first stage's (its class-token rows are kept as they are) and those of the spatial-reduction blocks -- is resampled
bicubically from its old square grid to the new one.  Host-side, once per load; the resolutions themselves are served by the
block-streaming attention kernels (csrc/attn_mfma.hip, N > 288).
"""
import math

import torch
import torch.nn.functional as F


def _resample(grid_tokens, new_side):
    """[1, s*s, C] -> [1, new_side^2, C], bicubic, align_corners=False (finetune_state_dict.py:49-54)."""
    side = int(math.sqrt(grid_tokens.shape[1]))
    c = grid_tokens.shape[2]
    img = grid_tokens.reshape(1, side, side, c).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=(new_side, new_side), mode='bicubic', align_corners=False)
    return img.permute(0, 2, 3, 1).flatten(1, 2)


def state_dict_interpolate_pos_embed(model_state_dict, state_dict):
    """Returns `state_dict` with every `*pos_embed*` entry resized to the shape `model_state_dict` expects."""
    assert 'tokens' in model_state_dict
    n_tok = 2 if model_state_dict['tokens'].shape[1] == 2 else 1
    for key, want in model_state_dict.items():
        assert key in state_dict, key
        if 'pos_embed' not in key:
            continue
        have = state_dict[key]
        lead = 0 if 'blocks' in key else n_tok            # SR blocks carry patch positions only
        new_side = int(math.sqrt(want.shape[1] - lead))
        if new_side == int(math.sqrt(have.shape[1] - lead)):
            continue
        grid = _resample(have[:, lead:, :], new_side)
        state_dict[key] = grid if lead == 0 else torch.cat((have[:, :lead, :], grid), dim=1)
    return state_dict


def load_interpolated_state_dict(model_state_dict, ckpt_path):
    """Reference checkpoint layout (main.py:506-512): {'model': ..., 'model_ema': ...}; the EMA weights win when present."""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)      # 'args' is an argparse.Namespace
    sd = ckpt['model_ema'] if 'model_ema' in ckpt else ckpt['model']
    return state_dict_interpolate_pos_embed(model_state_dict, sd)

"""Transformer block of the ViT-Res (super)network: parameter containers + supernet bookkeeping.

Mirror of reference nets/supernet_blocks.py (Mlp :17-71, Attention :74-161, Block :164-259) with the
same constructor arguments, parameter names and `rewiring()` semantics.  The arithmetic of
Block.forward (:209-255) is executed by vitres.functional.{attn,mlp}_branch_{fwd,bwd}.
"""
import torch
import torch.nn as nn

from .channel_drop import ChannelDrop
from .drop import DropPath
from .masked_layer_norm import MaskedLayerNorm

_NUM_WARMUP_EPOCHS_CHANNEL = 15
_EXAMPLE_PER_ARCH = 16


def _maybe_drop(choices, num_warmup_epochs, example_per_arch, single_arch):
    if choices is None:
        return None
    return ChannelDrop(num_channels_to_keep=choices, num_warmup_epochs=num_warmup_epochs,
                       example_per_arch=example_per_arch, single_arch=single_arch)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.,
                 num_channels_to_keep=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS_CHANNEL,
                 example_per_arch=_EXAMPLE_PER_ARCH, single_arch=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert act_layer is nn.GELU and drop == 0., 'HIP path implements erf-GELU and drop=0 only'
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.channel_drop_layer = _maybe_drop(num_channels_to_keep, num_warmup_epochs, example_per_arch, single_arch)

    @torch.no_grad()
    def rewiring(self):
        """Sort hidden units by L1 importance, descending (reference :55-71)."""
        score = self.fc2.weight.abs().sum(dim=0) + self.fc1.weight.abs().sum(dim=1) + self.fc1.bias.abs()
        order = torch.sort(score, descending=True)[1]
        self.fc1.weight.copy_(self.fc1.weight[order, :])
        self.fc1.bias.copy_(self.fc1.bias[order])
        self.fc2.weight.copy_(self.fc2.weight[:, order])


class Attention(nn.Module):
    def __init__(self, dim, num_heads, head_dim=64, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.,
                 num_channels_to_keep=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS_CHANNEL,
                 example_per_arch=_EXAMPLE_PER_ARCH, single_arch=False):
        super().__init__()
        assert qkv_bias and attn_drop == 0. and proj_drop == 0., 'HIP path implements qkv_bias=True, no dropout'
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, num_heads * head_dim * 3, bias=True)
        self.proj = nn.Linear(num_heads * head_dim, dim)
        self.channel_drop_layer = _maybe_drop(num_channels_to_keep, num_warmup_epochs, example_per_arch, single_arch)

    @torch.no_grad()
    def rewiring(self):
        """Sort heads by L1 importance, descending (reference :123-161)."""
        h, d = self.num_heads, self.head_dim
        # same summation order as the reference (proj + qkv bias + qkv weight); matters for fp32 ties only
        score = (self.proj.weight.abs().sum(dim=0).reshape(h, d).sum(dim=1)
                 + self.qkv.bias.abs().reshape(3, h, d).sum(dim=0).sum(dim=1)
                 + self.qkv.weight.abs().sum(dim=1).reshape(3, h, d).sum(dim=0).sum(dim=1))
        order = torch.sort(score, descending=True)[1]
        cin = self.qkv.weight.shape[1]
        self.qkv.weight.copy_(self.qkv.weight.reshape(3, h, d, cin)[:, order].reshape(3 * h * d, cin))
        self.qkv.bias.copy_(self.qkv.bias.reshape(3, h, d)[:, order].reshape(3 * h * d))
        cout = self.proj.weight.shape[0]
        self.proj.weight.copy_(self.proj.weight.reshape(cout, h, d)[:, order].reshape(cout, h * d))


class Block(nn.Module):
    def __init__(self, dim, num_heads, head_dim, mlp_features, qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, num_chs_to_keep_attn=None, num_chs_to_keep_mlp=None,
                 num_chs_to_keep_block=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS_CHANNEL,
                 example_per_arch=_EXAMPLE_PER_ARCH, single_arch=False):
        super().__init__()
        self.layer_drop = _maybe_drop(num_chs_to_keep_block, num_warmup_epochs, example_per_arch, single_arch)
        self.norm1 = MaskedLayerNorm(dim)
        self.attn = Attention(dim, num_heads=num_heads, head_dim=head_dim, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop, num_channels_to_keep=num_chs_to_keep_attn,
                              num_warmup_epochs=num_warmup_epochs, example_per_arch=example_per_arch,
                              single_arch=single_arch)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = MaskedLayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_features, act_layer=act_layer, drop=drop,
                       num_channels_to_keep=num_chs_to_keep_mlp, num_warmup_epochs=num_warmup_epochs,
                       example_per_arch=example_per_arch, single_arch=single_arch)

    def rewiring(self):
        self.attn.rewiring()
        self.mlp.rewiring()

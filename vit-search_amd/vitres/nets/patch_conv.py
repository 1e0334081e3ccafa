"""Patch embeddings: parameter containers with the reference's state_dict layout.

  PatchEmbed      -- timm 0.3.2 PatchEmbed (Conv2d k=s=patch); used by network_def embed type 0
                     (reference nets/vit_sr_supernet.py:227,234).
  PatchConvEmbed  -- reference nets/patch_conv.py:39-73 (3x ConvBnAct @112^2 + residual + Conv 7x7/s7);
                     embed types 4 and 5.
Execution is by the HIP path of the parent model (vitres/functional.py, vitres/stem.py).
"""
import torch.nn as nn


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class ConvBnAct(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), padding=(1, 1), stride=(1, 1)):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=padding, stride=stride,
                              bias=False)
        self.bn = nn.BatchNorm2d(num_features=out_channels)
        self.act = nn.ReLU()


class PatchConvEmbed(nn.Module):
    def __init__(self, embed_dim, img_size=224, patch_size=14, in_chans=3, mid_chans=24):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.patch_grid = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.patch_grid[0] * self.patch_grid[1]
        self.mid_chans = mid_chans
        self.conv1 = ConvBnAct(in_chans, mid_chans, stride=(2, 2))
        self.conv2 = ConvBnAct(mid_chans, mid_chans)
        self.conv3 = ConvBnAct(mid_chans, mid_chans)
        assert self.patch_size[0] % 2 == 0 and self.patch_size[1] % 2 == 0
        k = (self.patch_size[0] // 2, self.patch_size[1] // 2)
        self.conv_proj = nn.Conv2d(mid_chans, embed_dim, kernel_size=k, stride=k)

"""Single-stage (no spatial reduction) flexible ViT and its supernet, patch 16 -- the sibling of the ViT-Res model
(reference nets/vision_transformer_supernet.py:45-283; factories flexible_vit_patch16_{224,192}[_supernet]).

The forward / backward is the ViT-Res hot path without `SpatialReductionPatchEmbedding` entries, so the class is the SR model
restricted to the sibling's grammar and surface:
  * `network_def` = embed (type 0) + transformer entries + head only (`depth == len(network_def) - 2`, :103-104);
  * `dst_head` exists whether or not there is a distillation token (:148-149) -- without one it is never used and never updated;
  * `no_weight_decay()` = {'pos_embed', 'tokens'} (:171-173);
  * the final norm runs over every token before the token rows are taken (:199-200): row-wise, hence the same numbers.
"""
import torch.nn as nn

from ..registry import register_model
from .masked_layer_norm import MaskedLayerNorm
from .vit_sr_supernet import (_BLOCK_TYPE, _TYPE_IS_EMBED, _TYPE_IS_TRANS, _NUM_WARMUP_EPOCHS, _cfg,
                              FlexibleDistillVisionTransformerSR)


class FlexibleDistillVisionTransformer(FlexibleDistillVisionTransformerSR):
    _PATCH_SIZES = (16,)
    _ALWAYS_DST_HEAD = True

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=MaskedLayerNorm, distill_token=True, network_def=None, supernet=False,
                 num_channels_to_keep=None, example_per_arch=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS, single_arch=False):
        depth = sum(1 for e in network_def if e[_BLOCK_TYPE] == _TYPE_IS_TRANS)
        assert depth == len(network_def) - 2, 'Block number error'
        assert network_def[0][_BLOCK_TYPE] == _TYPE_IS_EMBED
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, num_classes=num_classes,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, distill_token=distill_token, network_def=network_def, supernet=supernet,
                         num_channels_to_keep=num_channels_to_keep, example_per_arch=example_per_arch,
                         num_warmup_epochs=num_warmup_epochs, single_arch=single_arch, hybrid_arch=False, patch_output=False)
        if not distill_token:                           # dst_head is registered but unused: the reference never produces a
            for p in self.dst_head.parameters():        # gradient for it, so no optimizer ever touches it
                p.requires_grad_(False)

    def no_weight_decay(self):
        return {'pos_embed', 'tokens'}

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.cls_head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.dst_head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._arena = None


def _factory(**fixed):
    def make(pretrained=False, **kwargs):
        model = FlexibleDistillVisionTransformer(patch_size=16, distill_token=True, **fixed, **kwargs)
        model.default_cfg = _cfg()
        return model
    return make


@register_model
def flexible_vit_patch16_224(pretrained=False, **kwargs):
    return _factory()(pretrained, **kwargs)


@register_model
def flexible_vit_patch16_224_supernet(pretrained=False, **kwargs):
    return _factory(supernet=True)(pretrained, **kwargs)


@register_model
def flexible_vit_patch16_192(pretrained=False, **kwargs):
    return _factory(img_size=192)(pretrained, **kwargs)


@register_model
def flexible_vit_patch16_192_supernet(pretrained=False, **kwargs):
    return _factory(supernet=True, img_size=192)(pretrained, **kwargs)

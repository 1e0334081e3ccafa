"""ViT-Res / ViT-ResNAS (super)network on MI355X: same nn.Module surface as the reference
(nets/vit_sr_supernet.py:185-577: constructor arguments, `network_def` grammar, state_dict keys,
`forward(x, patch_output_type)`, `set_epoch`, `no_weight_decay`, registry factories), executed by
hand-written HIP kernels through the C ABI in include/vitres_hip.h.

Execution model
  * parameters live in one flat fp32 arena (nn.Parameters are views); a bf16 shadow of the arena is
    refreshed by one cast kernel per forward (fast mode) -- fp32 parity mode uses the arena directly;
  * one forward = one autograd node: the layer sequence runs as explicit kernel launches and keeps a
    tape; backward replays the tape in reverse and writes parameter gradients into a flat gradient
    arena (returned to autograd as views, so torch optimizers and DDP see ordinary .grad tensors);
  * all ChannelDrop prefix masks of a forward are sampled on the host first (bit-exact reference RNG
    protocol), shipped as one int32 [n_masks, B] tensor and consumed by kernel epilogues.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import kernels as K
from ..registry import register_model
from .channel_drop import ChannelDrop
from .masked_layer_norm import MaskedLayerNorm
from .patch_conv import PatchConvEmbed, PatchEmbed
from .supernet_blocks import Block

# network_def grammar (reference :20-47)
_BLOCK_EMBED_INDEX = 0
_EMBED_CHANNEL = 1
_EMBED_CONV_MID_CHANNELS = 2
_BLOCK_HEAD_INDEX = -1
_HEAD_OUT_CHANNEL = 2
_HEAD_IN_CHANNEL = 1
_BLOCK_TYPE = 0
_TYPE_IS_EMBED = 0
_TYPE_IS_TRANS = 1
_TYPE_IS_HEAD = 2
_TYPE_IS_SR = 3
_TYPE_IS_CONV_EMBED = 4
_TYPE_IS_FLEXIBLE_CONV_EMBED = 5
_BLOCK_ATTN_IDX = 1
_BLOCK_FFN_IDX = 2
_BLOCK_EXISTS_IDX = 3
_NUM_WARMUP_EPOCHS = 15
_REQUIRE_CUDA = True   # tests that emulate the kernels on CPU lift this guard; the kernels themselves never accept CPU tensors


def _cfg(url='', **kwargs):
    return {'url': url, 'num_classes': 1000, 'input_size': (3, 224, 224), 'pool_size': None, 'crop_pct': .9,
            'interpolation': 'bicubic', 'mean': (0.485, 0.456, 0.406), 'std': (0.229, 0.224, 0.225),
            'first_conv': 'patch_embed.proj', 'classifier': 'head', **kwargs}


def trunc_normal_(t, std):
    return nn.init.trunc_normal_(t, mean=0., std=std, a=-2., b=2.)


_SKIP_DROPPED = __import__("os").environ.get("VITRES_SKIP_DROPPED_LAYERS", "1") != "0"
# VITRES_SKIP_MASKED_WRITES=0: the masked GEMMs store the zeros of their fully masked tiles (measurement / A-B aid)
_SKIP_WRITES = __import__("os").environ.get("VITRES_SKIP_MASKED_WRITES", "1") != "0"

class BypassBlock(nn.Module):
    """Removed transformer block (exists == 0): identity, resets the layer mask (reference :50-56)."""

    def __init__(self, *args, **kwargs):
        super().__init__()


class SpatialReductionPatchEmbedding(nn.Module):
    """Stage transition g x g -> g/2 x g/2 tokens, C -> C' channels (reference :59-182)."""

    def __init__(self, img_size, in_features, out_features, patch_size=2, distill_token=True,
                 num_channels_to_keep=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS, example_per_arch=None,
                 single_arch=False):
        super().__init__()
        assert patch_size == 2
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.distill_token = distill_token
        self.num_tokens = 2 if distill_token else 1
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, out_features))
        self.norm = MaskedLayerNorm(num_channels=in_features)
        self.patch_reduce = nn.Conv2d(in_features, out_features, kernel_size=patch_size + 1, stride=patch_size,
                                      padding=patch_size // 2)
        assert out_features >= in_features
        self.token_transform = nn.Linear(in_features, out_features)
        trunc_normal_(self.pos_embed, std=.02)
        self.channel_drop = None
        if num_channels_to_keep is not None:
            self.channel_drop = ChannelDrop(num_channels_to_keep=num_channels_to_keep,
                                            num_warmup_epochs=num_warmup_epochs,
                                            example_per_arch=example_per_arch, single_arch=single_arch)

    def extra_repr(self):
        return 'num_patches={}, distill_token={}, pos_embed: {}'.format(self.num_patches, self.distill_token,
                                                                      tuple(self.pos_embed.shape))


class _Plan:
    """Host-side description of one forward: which keep vector / drop-path scale each layer uses."""
    __slots__ = ("keep_dev", "rows", "layers", "head", "scales", "batch", "n_dp", "order", "dp_noise", "host", "embed_col", "groups",
                 "keeps_host", "scales_host", "embed_map", "want_tape", "dead_blocks", "skip_writes")

    def __init__(self):
        self.rows, self.layers, self.keep_dev, self.head, self.scales, self.batch, self.n_dp = [], [], None, None, None, 0, 0
        self.order = None
        self.groups = 1          # architecture groups of the batch (contiguous in the arch-grouped order): vr_gemm_args.m_groups
        self.dp_noise = None     # test hook: the uniform draws of drop_path (nets/drop.py:23), [n_dp, B] in the CALLER's sample order
        self.host = None         # (int32 [n_rows, B] keeps, float32 [n_dp, B] DropPath scales) on the host, internal row order
        self.keeps_host = self.scales_host = None
        self.want_tape = True    # False under torch.no_grad(): the forward keeps nothing for a backward
        self.embed_map = None    # int64 device map: internal sample -> caller's sample, consumed by the type-0 patch gather
        self.skip_writes = False # every kernel tile of this batch sees one architecture: fully masked output tiles stay unwritten (kernels.WRITE_SKIP)
        self.dead_blocks = None  # forward-only: indices of the blocks whose layer keep is 0 for every sample (left out on the host)
        self.embed_col = None    # type-0 patch embedding: the patchify operand already gathered (engine.GraphedTrainStep)

    def add(self, keep):
        if keep is None:
            return None
        self.rows.append(keep)
        return len(self.rows) - 1

    def k(self, idx):
        return None if idx is None else self.keep_dev[idx]


class _ViTResFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, plan, with_patch, *params):
        save = plan.want_tape and any(ctx.needs_input_grad[4:])      # (needs_input_grad ignores torch.no_grad())
        cls, pat, tape = model._run_forward(x, plan, with_patch, save)
        ctx.model, ctx.plan, ctx.tape, ctx.with_patch = model, plan, tape, with_patch
        ctx.set_materialize_grads(False)
        if with_patch:
            return cls, pat
        return cls

    @staticmethod
    def backward(ctx, dcls, dpat=None):
        if ctx.tape is None:
            raise RuntimeError('backward called on a forward that saved no tape')
        grads = ctx.model._run_backward(ctx.tape, ctx.plan, dcls, dpat)
        ctx.tape = None
        return (None, None, None, None) + tuple(grads)


class FlexibleDistillVisionTransformerSR(nn.Module):
    _PATCH_SIZES = (14,)
    _ALWAYS_DST_HEAD = False          # the patch-16 sibling registers dst_head even without a distillation token

    def __init__(self, img_size=224, patch_size=14, in_chans=3, num_classes=1000, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=MaskedLayerNorm, distill_token=True, network_def=None,
                 supernet=False, num_channels_to_keep=None, example_per_arch=None,
                 num_warmup_epochs=_NUM_WARMUP_EPOCHS, single_arch=False, hybrid_arch=False, patch_output=False):
        super().__init__()
        assert patch_size in self._PATCH_SIZES
        assert not (patch_output and distill_token), 'Currently support only either ShiftTokenMixup or Distillation.'
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError('drop_rate / attn_drop_rate must be 0 (every shipped recipe uses 0)')
        self.network_def = network_def
        self.num_classes = num_classes
        assert network_def[_BLOCK_HEAD_INDEX][_HEAD_OUT_CHANNEL] == num_classes
        embed_dim = network_def[_BLOCK_EMBED_INDEX][_EMBED_CHANNEL]
        self.num_features = self.embed_dim = embed_dim
        self.img_size, self.patch_size, self.in_chans = img_size, patch_size, in_chans

        etype = network_def[_BLOCK_EMBED_INDEX][_BLOCK_TYPE]
        self.embed_type = etype
        if etype == _TYPE_IS_FLEXIBLE_CONV_EMBED:
            self.patch_embed = PatchConvEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                              embed_dim=embed_dim,
                                              mid_chans=network_def[_BLOCK_EMBED_INDEX][_EMBED_CONV_MID_CHANNELS])
        elif etype == _TYPE_IS_CONV_EMBED:
            self.patch_embed = PatchConvEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                              embed_dim=embed_dim)
        else:
            self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                          embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        grid = img_size // patch_size

        self.distill_token = distill_token
        self.num_tokens = 2 if distill_token else 1           # class token (+ distillation token: *_distill_* factories)
        self.tokens = nn.Parameter(torch.zeros(1, self.num_tokens, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.patch_output = patch_output

        self.embed_channel_drop = None
        if supernet:
            assert num_channels_to_keep is not None, 'Super-network numbers of channels to keep error'
            assert (example_per_arch is not None) or single_arch, 'Super-network forward-backward architecture error'
            assert isinstance(num_channels_to_keep, list), 'Num of channels to keep type error'
            assert len(num_channels_to_keep) == len(network_def), \
                'Lengths of num_channels_to_keep and network_def are not the same'
            self.embed_channel_drop = ChannelDrop(num_channels_to_keep=num_channels_to_keep[0],
                                                  num_warmup_epochs=num_warmup_epochs,
                                                  example_per_arch=example_per_arch,
                                                  single_arch=(single_arch or hybrid_arch))

        depth = sum(1 for b in network_def if b[_BLOCK_TYPE] == _TYPE_IS_TRANS)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth, device='cpu')]
        blocks, depth = [], 0
        for i, block_def in enumerate(network_def):
            btype = block_def[_BLOCK_TYPE]
            if btype not in (_TYPE_IS_SR, _TYPE_IS_TRANS):
                continue
            keep_blk = keep_attn = keep_mlp = keep_layer = None
            if supernet:
                keep_blk = num_channels_to_keep[i]
                if btype == _TYPE_IS_TRANS:
                    assert isinstance(keep_blk, dict)
                    keep_attn, keep_mlp, keep_layer = keep_blk['attn'], keep_blk['mlp'], keep_blk['layer']
                else:
                    assert isinstance(keep_blk, np.ndarray)
            if btype == _TYPE_IS_TRANS:
                attn_def, ffn_def = block_def[_BLOCK_ATTN_IDX], block_def[_BLOCK_FFN_IDX]
                assert attn_def[0] == ffn_def[0], 'Block {}: embedding dim mismatch'.format(depth)
                assert attn_def[0] == embed_dim, \
                    'Block {}: embedding dim is not consistent with patch embedding'.format(depth)
                cls_ = Block if block_def[_BLOCK_EXISTS_IDX] else BypassBlock
                blocks.append(cls_(dim=embed_dim, num_heads=attn_def[1], head_dim=attn_def[2],
                                   mlp_features=ffn_def[1], drop_path=dpr[depth],
                                   num_chs_to_keep_attn=keep_attn, num_chs_to_keep_mlp=keep_mlp,
                                   num_chs_to_keep_block=keep_layer, num_warmup_epochs=num_warmup_epochs,
                                   example_per_arch=example_per_arch, single_arch=single_arch))
                depth += 1
            else:
                assert block_def[1] == embed_dim, 'Block {}: SR input embedding size error'.format(i)
                blocks.append(SpatialReductionPatchEmbedding(
                    img_size=grid, in_features=block_def[1], out_features=block_def[2],
                    num_channels_to_keep=keep_blk, num_warmup_epochs=num_warmup_epochs,
                    example_per_arch=example_per_arch, single_arch=(single_arch or hybrid_arch),
                    distill_token=distill_token))
                embed_dim = block_def[2]
                grid = grid // 2
        self.blocks = nn.ModuleList(blocks)
        self.norm = norm_layer(embed_dim)
        assert embed_dim == network_def[_BLOCK_HEAD_INDEX][_HEAD_IN_CHANNEL]
        self.cls_head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.dst_head = (nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()) \
            if (distill_token or self._ALWAYS_DST_HEAD) else None
        self.patch_head = nn.Linear(embed_dim, num_classes) if patch_output else None

        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.tokens, std=.02)
        self.apply(self._init_weights)

        self.num_warmup_epochs = num_warmup_epochs
        self.epoch_now = None
        self.is_supernet = supernet
        self.compute_dtype = torch.bfloat16          # fast mode; torch.float32 = exact parity mode
        self._arena = None
        self.last_keeps = None                       # keep vectors of the last forward (call order), for tests

    # ---- reference API -----------------------------------------------------------------------
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, MaskedLayerNorm)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        out = ['tokens']
        for name, _ in self.blocks.named_parameters():
            if name.endswith(tuple(out)):
                out.append(name)
        return set(out)

    def get_classifier(self):
        return self.cls_head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.cls_head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        if self.distill_token:                                # (the reference re-creates dst_head unconditionally, :391-394)
            self.dst_head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._arena = None

    def set_epoch(self, epoch):
        self.epoch_now = epoch
        for m in self.modules():
            if isinstance(m, ChannelDrop):
                m.set_epoch(epoch)
        if self.is_supernet and self.num_warmup_epochs >= self.epoch_now:
            for m in self.modules():
                if isinstance(m, Block):
                    m.rewiring()
            self.invalidate_shadow()

    def invalidate_shadow(self):
        """Call after changing parameter VALUES outside an optimizer step (rewiring, load_state_dict, manual edits) when
        vitres.optim.FlatAdamW maintains the bf16 weight shadow: re-casts it right away (forwards then skip their own cast,
        also inside a captured hipGraph)."""
        a = self._arena
        self._stem_fold = None             # BatchNorm-folded evaluation stem (stem.drop_fold)
        if a is not None:
            a["shadow_ver"] = None
        if a is not None and a.get("shadow_ok") and a["flat"].is_cuda:
            K.cast_bf16(a["flat"], a["shadow"])

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_shadow()
        return out

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def channel_drops_in_call_order(self):
        out = [self.embed_channel_drop] if self.embed_channel_drop is not None else []
        for blk in self.blocks:
            if isinstance(blk, Block):
                out += [d for d in (blk.attn.channel_drop_layer, blk.layer_drop, blk.mlp.channel_drop_layer)
                        if d is not None]
            elif isinstance(blk, SpatialReductionPatchEmbedding) and blk.channel_drop is not None:
                out.append(blk.channel_drop)
        return out

    # ---- flat parameter arena ------------------------------------------------------------------
    def _ensure_arena(self, device):
        params = list(self.parameters())
        a = self._arena
        if a is not None and a["flat"].device == device and len(a["params"]) == len(params) and all(
                p.data_ptr() == a["flat"].data_ptr() + 4 * off for p, (off, _) in zip(params, a["offsets"])):
            return a
        offsets, total = [], 0
        for p in params:
            offsets.append((total, p.numel()))
            total += (p.numel() + 7) // 8 * 8                 # 32-byte aligned fp32 / 16-byte aligned bf16 views
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, (off, n) in zip(params, offsets):
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
        a = {"flat": flat, "params": params, "offsets": offsets, "index": {id(p): i for i, p in enumerate(params)},
             "shadow": torch.empty(total, dtype=torch.bfloat16, device=device), "gflat": None, "tmap": {}, "tr": None}
        # transposed bf16 shadows W^T [in, roundup(out, 8)] (one batched transposing cast per step).  Round 2: the data-gradient
        # GEMMs read the forward's weight itself (vr_gemm b_trans on the LDS-DMA kernel: k-major weight slices, transposing LDS
        # reads); only the Linears whose data gradient runs inside vr_gemm_ln (qkv / fc1 of the narrow first stage) still get a
        # transposed copy.  VITRES_WT_SHADOWS=all: every Linear (round-1 layout), none: no copies (no fused LayerNorm backward)
        import os as _os
        which = a["wt_mode"] = _os.environ.get("VITRES_WT_SHADOWS", "fused")
        if self.compute_dtype == torch.bfloat16 and which != "none":
            entries, tot_t = [], 0
            for name, mod in self.named_modules():
                fused = (name.endswith("attn.qkv") or name.endswith("mlp.fc1")) and isinstance(mod, nn.Linear) and \
                    mod.in_features <= Fn.FUSE_LN_MAXN
                if which != "all" and not fused:
                    continue
                if isinstance(mod, nn.Linear) and id(mod.weight) in a["index"]:
                    w = mod.weight
                    out_f, in_f = w.shape
                    ld_t = (out_f + 7) // 8 * 8
                    a["tmap"][id(w)] = (tot_t, ld_t)
                    entries.append((offsets[a["index"][id(w)]][0], tot_t, out_f, in_f, ld_t))
                    tot_t += in_f * ld_t
            if entries:
                a["shadow_t"] = torch.zeros(tot_t, dtype=torch.bfloat16, device=device)
                a["tr"] = K.tr_descs(entries, device)
        self._arena = a
        return a

    def _wc(self, p, rows=None):
        """Compute-dtype 2-D view [out, in] of parameter p."""
        a = self._arena
        rows = p.shape[0] if rows is None else rows
        if self.compute_dtype == torch.float32:
            return p.detach().view(rows, -1)
        off, n = a["offsets"][a["index"][id(p)]]
        return a["shadow"][off:off + n].view(rows, -1)

    def _lin(self, m, wkey=None):
        w = self._wc(m.weight)
        a = self._arena
        w_t, ld_t = None, 0
        if id(m.weight) in a["tmap"]:
            off_t, ld_t = a["tmap"][id(m.weight)]
            w_t = a["shadow_t"][off_t:off_t + m.weight.shape[1] * ld_t].view(m.weight.shape[1], ld_t)
        return Fn.Weights(m.weight, m.bias.detach() if m.bias is not None else None, w, w.shape[1], w_t, ld_t)

    def _zero_grad_arena(self, buf, B):
        """optimizer.zero_grad() of the flat gradient arena (reference engine.py:175): every weight gradient ACCUMULATES (fp32 atomics
        of the token splits), so the whole arena is cleared, in one launch."""
        a = self._arena
        cached = a.get("zero_ranges")
        if cached is None or cached[0] != buf.numel():
            cached = a["zero_ranges"] = (buf.numel(), [(0, buf.numel())], 0)
        if buf.is_cuda:
            K.zero_ranges(buf, cached[1])
        else:
            for lo, hi in cached[1]:
                buf[lo:hi].zero_()
        return cached[2]

    def _gview(self, p):
        a = self._arena
        off, n = a["offsets"][a["index"][id(p)]]
        return a["gcur"][off:off + n].view(p.shape)

    def _ln_parts(self):
        """id(LayerNorm weight) -> [2, LN_COPIES, C] zero-filled rows of partial weight / bias gradient sums
        (functional.LN_COPIES; one allocation for the whole network, made before any capture can be running)."""
        a = self._arena
        parts = a.get("ln_parts")
        if parts is None or a.get("ln_copies") != Fn.LN_COPIES:
            norms = [m for m in self.modules() if isinstance(m, (nn.LayerNorm, MaskedLayerNorm)) and m.weight is not None]
            total = sum(2 * Fn.LN_COPIES * m.weight.numel() for m in norms)
            flat = torch.zeros(total, dtype=torch.float32, device=a["flat"].device)
            parts, off = {}, 0
            for m in norms:
                n = 2 * Fn.LN_COPIES * m.weight.numel()
                parts[id(m.weight)] = flat[off:off + n].view(2, Fn.LN_COPIES, m.weight.numel())
                off += n
            a["ln_parts"], a["ln_parts_flat"], a["ln_copies"] = parts, flat, Fn.LN_COPIES
        return parts

    # ---- host-side mask plan -----------------------------------------------------------------------
    def sample_plan(self, B):
        """Host side of one forward: sample every ChannelDrop (reference RNG protocol, call order) and lay out which
        keep row / drop-path scale each layer uses.  No device work -- see _upload_plan.
        Every keep vector of a forward is `g.repeat(.)` of a short group vector g (one entry per architecture group,
        channel_drop.py:101-109): the plan is assembled on the group vectors and expanded to [rows, B] once (numpy)."""
        plan = _Plan()
        plan.batch = B
        tr = self.training
        groups = []                      # group vectors in ChannelDrop call order (last_keeps)

        def samp(cd, ch):
            if cd is None:
                return None
            g = cd.sample_groups(B, ch, tr)
            groups.append(g)
            return g
        rows = []                        # group vectors of the plan's keep rows

        def add(g):
            if g is None:
                return None
            rows.append(g)
            return len(rows) - 1

        def gmin(a_, b_):                # AND of two prefix masks; group vectors have 1 or G entries
            return np.minimum(a_, b_)
        recipe = self.__dict__.get("_plan_recipe")
        if recipe is None:               # the module walk, once: (kind, ChannelDrops, widths, has DropPath) per entry of self.blocks
            recipe = []
            for blk in self.blocks:
                if isinstance(blk, Block):
                    dpm = blk.drop_path
                    recipe.append((1, blk.attn.channel_drop_layer, blk.attn.num_heads * blk.attn.head_dim, blk.layer_drop,
                                   blk.norm1.num_channels, blk.mlp.channel_drop_layer, blk.mlp.fc1.out_features,
                                   (not isinstance(dpm, nn.Identity)) and dpm.drop_prob > 0))
                elif isinstance(blk, SpatialReductionPatchEmbedding):
                    recipe.append((2, blk.channel_drop, blk.token_transform.out_features))
                else:
                    recipe.append((0,))
            self.__dict__["_plan_recipe"] = recipe
        embed_keep = samp(self.embed_channel_drop, self.embed_dim)
        layer_keep = None
        e_idx = add(embed_keep)
        plan.layers.append({"embed": e_idx})
        n_dp = 0
        for ent in recipe:
            if ent[0] == 1:
                _, cd_a, hd, cd_l, cn, cd_m, fo, has_dp = ent
                ka = samp(cd_a, hd)
                cur = None
                if cd_l is not None:
                    cur = samp(cd_l, cn)
                    if layer_keep is not None:
                        cur = gmin(cur, layer_keep)
                if embed_keep is not None:
                    cur = embed_keep if cur is None else gmin(cur, embed_keep)
                km = samp(cd_m, fo)
                dp = tr and has_dp
                # a dropped layer (layer keep 0 for an architecture group, supernet_blocks.py:243,251: both branch outputs are
                # multiplied by the layer mask) contributes exactly nothing -- no output, no parameter gradient: its attention / MLP
                # widths are zeroed in the rows the KERNELS read (the sampled values, `groups`, are untouched), so that the masked-work
                # rules skip the qkv / fc1 GEMMs, the attention cores and their backward for those samples instead of computing
                # values the layer mask then discards (VITRES_SKIP_DROPPED_LAYERS=0: compute them)
                if _SKIP_DROPPED and cur is not None and ka is not None and km is not None and np.any(np.asarray(cur) == 0):
                    ka = np.where(np.asarray(cur) > 0, ka, 0)
                    km = np.where(np.asarray(cur) > 0, km, 0)
                plan.layers.append({"embed": e_idx, "attn": add(ka), "mlp": add(km), "out": add(cur),
                                    "dp": (n_dp if dp else None)})
                n_dp += 2 if dp else 0
                layer_keep = cur
            elif ent[0] == 2:
                nk = samp(ent[1], ent[2])
                n_idx = add(nk)
                plan.layers.append({"embed": e_idx, "new": n_idx})
                embed_keep, e_idx, layer_keep = nk, n_idx, None
            else:
                plan.layers.append(None)
                layer_keep = None
        plan.head = e_idx
        plan.n_dp = n_dp
        # Arch-grouped execution order: sample b runs architecture (b mod G), G = B / example_per_arch
        # (channel_drop.py:101-105 tiles the G sampled rows).  Running the batch as G contiguous groups makes every
        # GEMM tile / wgrad split see ONE architecture, so masked K slices and output tiles can be skipped.  Results
        # are returned in the caller's order; nothing but the internal row order changes.
        cd = self.embed_channel_drop
        if tr and cd is not None and rows and cd.example_per_arch and B % cd.example_per_arch == 0:
            epa = cd.example_per_arch
            G = B // epa
            if 1 < G < B:
                plan.order = [j * G + g for g in range(G) for j in range(epa)]
                plan.groups = G
            # One architecture per kernel tile (the G contiguous groups above, or one architecture for the whole batch) and a network
            # whose masked Linears all run on the group-pure bf16 kernels: the zeros of fully masked activation tiles -- hidden units /
            # heads beyond an architecture's width, whole dropped layers -- are then never read and need not be written
            plan.skip_writes = bool(_SKIP_WRITES and (plan.groups > 1 or G == 1) and self._masked_writes_skippable(B))

        def expand(gs, who):             # [len(gs), B]: entry b of a group vector g is g[b % len(g)]
            if not gs:
                return None
            out = np.empty((len(gs), B), dtype=np.int64)
            for i, g in enumerate(gs):
                out[i] = g[who % len(g)]
            return out
        caller = np.arange(B)
        plan.keeps_host = expand(rows, np.asarray(plan.order) if plan.order is not None else caller)
        plan.rows = list(torch.from_numpy(plan.keeps_host)) if rows else []
        log = expand(groups, caller)
        self.last_keeps = list(torch.from_numpy(log)) if groups else []
        return plan

    def _masked_writes_skippable(self, B):
        """The part of _Plan.skip_writes that does not depend on the draw: bf16 kernels; every transformer block's widths multiples of
        the 64-wide K slice and at most 4096 (64 slices), every activation a reader walks below 4 GB (32-bit byte offsets) -- the forms
        gemm_ntk.hip covers: a reader of skipped tiles (sched 0x80000) that gemm_ntk.hip declines is refused by every other kernel
        (VR_EUNSUPPORTED), so such a network keeps writing its zeros."""
        dims_ok = getattr(self, "_skippable_dims", None)
        if dims_ok is None:
            ok, widest = True, 0
            for blk in self.blocks:
                if isinstance(blk, Block):
                    dims = (blk.attn.qkv.in_features, blk.attn.num_heads * blk.attn.head_dim, blk.mlp.fc1.out_features)
                    ok = ok and all(d % 64 == 0 and d <= 4096 for d in dims)
                    widest = max(widest, 3 * dims[1], dims[2])
            dims_ok = self._skippable_dims = (ok, widest)
        ok, widest = dims_ok
        tokens = self.pos_embed.shape[1]                                   # (the first stage's: later stages have fewer)
        return ok and B * tokens * widest * 2 < 0xfff00000 and self.compute_dtype == torch.bfloat16

    def _dp_scales_host(self, plan):
        """DropPath scales floor(keep_prob + u) / keep_prob (nets/drop.py:21-26) of one forward, [n_dp, B] float32 on the host in
        the internal row order.  u comes from plan.dp_noise (tests) or from a private CPU generator: the reference draws on the
        device, so the draws never touch the CPU generator the ChannelDrops consume -- neither do these."""
        B = plan.batch
        kp = getattr(self, "_dp_keep_prob_host", None)
        if kp is None or kp.shape[0] != plan.n_dp:
            vals = []
            for blk in self.blocks:
                if isinstance(blk, Block) and not isinstance(blk.drop_path, nn.Identity) and blk.drop_path.drop_prob > 0:
                    vals += [1.0 - blk.drop_path.drop_prob] * 2
            kp = np.asarray(vals, dtype=np.float32).reshape(-1, 1)
            self._dp_keep_prob_host = kp
        if plan.dp_noise is not None:                      # injected draws (caller order) -> the internal, arch-grouped row order
            noise = torch.as_tensor(plan.dp_noise, dtype=torch.float32).reshape(plan.n_dp, B).numpy()
            if plan.order is not None:
                noise = noise[:, np.asarray(plan.order)]
        else:
            noise = torch.rand(plan.n_dp, B, generator=self.drop_path_generator()).numpy()
        return (np.floor(kp + noise.astype(np.float32)) / kp).astype(np.float32)

    def drop_path_generator(self, seed=None):
        """The private CPU generator the DropPath draws come from (the reference draws them on the device, nets/drop.py:23 --
        never from the CPU generator the ChannelDrops consume).  Seeded on first use from torch.initial_seed() and the
        data-parallel rank, so that ranks draw different noise also when `arch_sample='single'` re-seeds the global generator
        identically everywhere; `seed` re-seeds it explicitly.  Its state is part of vitres.checkpoint's RNG bundle
        (drop_path_rng_state / set_drop_path_rng_state): a resumed run continues the same noise stream."""
        gen = getattr(self, "_dp_gen", None)
        if gen is None or seed is not None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            base = torch.initial_seed() if seed is None else int(seed)
            gen = self._dp_gen = torch.Generator(device="cpu")
            gen.manual_seed(((base + rank) * 2654435761 + 97) % (2 ** 63))
        return gen

    def drop_path_rng_state(self):
        return self.drop_path_generator().get_state()

    def set_drop_path_rng_state(self, state):
        self.drop_path_generator().set_state(state)

    def _order_tensors(self, order, device):
        cache = getattr(self, "_order_cache", None)
        key = (len(order), order[1] if len(order) > 1 else 0, str(device))
        if cache is None or cache[0] != key:
            o = torch.tensor(order, dtype=torch.int64)
            cache = (key, o.to(device), torch.argsort(o).to(device))
            self._order_cache = cache
        return cache[1], cache[2]

    def plan_host_buffer(self, plan):
        """(keeps int32 [n_rows, B], scales float32 [n_dp, B]) of a plan as ONE flat int32 host array (scales bit-cast), the unit
        a training step uploads -- see engine.GraphedTrainStep."""
        if plan.keeps_host is None and plan.rows:          # a plan assembled row by row (evo_eval.plan_for_subnet)
            plan.keeps_host = torch.stack([torch.as_tensor(r) for r in plan.rows]).numpy()
        nk = 0 if plan.keeps_host is None else plan.keeps_host.size
        ns = plan.n_dp * plan.batch
        flat = np.empty(nk + ns, dtype=np.int32)
        if nk:
            flat[:nk] = plan.keeps_host.reshape(-1)
        if ns:
            if plan.scales_host is None:
                plan.scales_host = self._dp_scales_host(plan)
            flat[nk:] = plan.scales_host.reshape(-1).view(np.int32)
            # a sample whose branch DropPath dropped (scale 0: nets/drop.py:21-26 multiplies the branch output by it) gets nothing
            # from that branch and gives its parameters no gradient: the attention / MLP width the KERNELS read for it is zeroed, so
            # that its attention cores (and the GEMM tiles that hold only such samples) are skipped instead of computed and discarded
            if _SKIP_DROPPED and nk and self.training and not getattr(plan, "_dp_folded", False):
                kh = flat[:nk].reshape(plan.keeps_host.shape)
                for L in plan.layers:
                    if L is not None and L.get("dp") is not None and L.get("attn") is not None and L.get("mlp") is not None:
                        # (width k -> -(k + 2), not 0: a sample masked on its own shares kernel tiles with the live rows of its
                        # architecture group; every kernel reads a negative width as 0, and gemm_ntk.hip's write skipping decodes
                        # the group's width from it -- the sample's zeros below that width are stored.  A layer dropped for the
                        # whole group keeps its 0: nothing of it is written or read)
                        ra, rm = kh[L["attn"]], kh[L["mlp"]]
                        da, dm = (plan.scales_host[L["dp"]] == 0) & (ra > 0), (plan.scales_host[L["dp"] + 1] == 0) & (rm > 0)
                        ra[da] = -ra[da] - 2
                        rm[dm] = -rm[dm] - 2
        return flat, nk

    def attach_plan_buffer(self, plan, dev_flat, nk):
        """Point a plan at a device copy of plan_host_buffer()'s array."""
        B = plan.batch
        if nk:
            plan.keep_dev = dev_flat[:nk].view(-1, B)
        if plan.n_dp:
            plan.scales = dev_flat[nk:nk + plan.n_dp * B].view(torch.float32).view(plan.n_dp, B)

    def _upload_plan(self, plan, device):
        """Device side of a plan: one H2D copy of all keep rows and DropPath scale vectors (unless a static buffer was
        attached, e.g. by a captured hipGraph)."""
        need_k = (plan.keeps_host is not None or bool(plan.rows)) and plan.keep_dev is None
        need_s = plan.n_dp and plan.scales is None
        if need_k or need_s:
            flat, nk = self.plan_host_buffer(plan)
            host = torch.from_numpy(flat)
            if device.type == 'cuda':
                host = host.pin_memory()
            dev = host.to(device, non_blocking=True)
            keep_dev, scales = plan.keep_dev, plan.scales
            self.attach_plan_buffer(plan, dev, nk)
            if not need_k:
                plan.keep_dev = keep_dev
            if not need_s:
                plan.scales = scales
        return plan

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x, patch_output_type=None, plan=None):
        if _REQUIRE_CUDA and not x.is_cuda:
            raise RuntimeError('vitres runs on MI355X through libvitres_hip.so only; got a %s tensor '
                               '(the CPU restatement lives in oracle/ and is test infrastructure)' % x.device)
        if patch_output_type not in (None, 'seq', 'avg'):
            raise ValueError()
        self._ensure_arena(x.device)
        # second result of the forward: 0 none, 1 patch head per token ('seq'), 2 patch head on the mean token ('avg'),
        # 3 distillation head (two-token variants, training and eval alike, reference :455-458)
        with_patch = int(bool(self.patch_output and self.training)) * (2 if patch_output_type == 'avg' else 1)
        if self.num_tokens == 2:
            with_patch = 3
        if plan is None:
            plan = self.sample_plan(x.shape[0])
        self._upload_plan(plan, x.device)
        params = self._arena["params"]
        x = x.contiguous().float()
        if plan.order is not None:
            fwd_idx, inv_idx = self._order_tensors(plan.order, x.device)
            if self.embed_type == _TYPE_IS_EMBED:
                plan.embed_map = fwd_idx             # the patch gather reads the images in the internal order (vr_im2col_patch_map)
            else:
                x = x.index_select(0, fwd_idx)
        plan.want_tape = torch.is_grad_enabled()
        out = _ViTResFn.apply(self, x, plan, with_patch, *params)
        if plan.order is not None:
            out = tuple(o.index_select(0, inv_idx) for o in out) if isinstance(out, tuple) else out.index_select(0, inv_idx)
        return out

    def loss_and_grad(self, x, targets, patch_targets=None, patch_output_type=None, plan=None, loss_out=None):
        """Forward + soft-target cross entropy of the training loop (engine.py:135-157 with timm's SoftTargetCrossEntropy:
        CE(cls, targets) [+ CE(patch, patch_targets) for 'seq' / CE(patch_mean, targets) for 'avg'; two-token models: CE of the
        class logits only, as the reference's no-teacher branch] + backward, WITHOUT autograd: the logits never leave the internal
        sample order, vr_softce_train scores them against the caller-ordered targets, accumulates the mean loss and writes the
        logit gradients in the layout the head's backward GEMMs read.  Same gradients as `loss.backward()` on forward()'s
        outputs; they land in the flat arena and are exposed as `p.grad`.  Requires zero_grad(set_to_none=True) since the last
        backward.  Returns the loss as a 0-d device tensor (loss_out: a preallocated fp32 [1] buffer, e.g. under hipGraph capture)."""
        if _REQUIRE_CUDA and not x.is_cuda:
            raise RuntimeError('vitres runs on MI355X through libvitres_hip.so only; got a %s tensor' % x.device)
        if not self.training:
            raise RuntimeError('loss_and_grad is a training-step primitive: call model.train() first')
        if patch_output_type not in (None, 'seq', 'avg'):
            raise ValueError()
        a = self._ensure_arena(x.device)
        if any(p_.grad is not None for p_ in a["params"]):
            raise RuntimeError('loss_and_grad needs fresh gradients: zero_grad(set_to_none=True) first')
        use_patch = bool(self.patch_output and patch_targets is not None or (self.patch_output and patch_output_type == 'avg'))
        with_patch = (2 if patch_output_type == 'avg' else 1) if (self.patch_output and use_patch) else 0
        if self.num_tokens == 2:
            with_patch = 3
        if plan is None:
            plan = self.sample_plan(x.shape[0])
        self._upload_plan(plan, x.device)
        x = x.contiguous().float()
        smap = None
        if plan.order is not None:
            smap, _ = self._order_tensors(plan.order, x.device)
            if self.embed_type == _TYPE_IS_EMBED:
                plan.embed_map = smap                # the patch gather reads the images in the internal order (vr_im2col_patch_map)
            elif plan.embed_col is None:
                x = x.index_select(0, smap)
        with torch.no_grad():
            cls, pat, tape = self._run_forward(x, plan, with_patch, True)
            loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=x.device)
            K.zero_(loss)
            dt = self.compute_dtype
            dcls = K.softce_train(cls, targets.float(), smap, 1, loss, dt)
            dpat = None
            if with_patch == 1:
                dpat = K.softce_train(pat, patch_targets.float(), smap, pat.shape[1], loss, dt)
            elif with_patch == 2:
                dpat = K.softce_train(pat, targets.float(), smap, 1, loss, dt)
            self._run_backward(tape, plan, dcls, dpat, ready=True)
            for p_ in a["params"]:
                if p_.requires_grad:
                    p_.grad = self._gview(p_)
        return loss[0]

    def _layer_params(self, blk):
        if isinstance(blk, Block):
            return {"n1w": blk.norm1.weight.detach(), "n1b": blk.norm1.bias.detach(),
                    "n2w": blk.norm2.weight.detach(), "n2b": blk.norm2.bias.detach(),
                    "qkv": self._lin(blk.attn.qkv), "proj": self._lin(blk.attn.proj),
                    "fc1": self._lin(blk.mlp.fc1), "fc2": self._lin(blk.mlp.fc2)}
        if isinstance(blk, SpatialReductionPatchEmbedding):
            w = blk.patch_reduce.weight
            # [co, ci, 3, 3] -> [co, (kh, kw, ci)] from the compute-dtype shadow of the weight: one strided copy, no cast pass
            wsrc = self._wc(w)
            wperm = torch.empty((w.shape[0], 9 * w.shape[1]), dtype=wsrc.dtype, device=wsrc.device)
            K.relayout(wsrc, wperm, w.shape[0], w.shape[1], 9)              # (vr_relayout: no torch permute / copy kernels)
            # (transposed copy for the data gradient only in the round-1 layout: the LDS-DMA kernel reads wperm itself, b_trans)
            wperm_t = wperm.t().contiguous() if self.compute_dtype == torch.bfloat16 and wperm.shape[0] % 8 == 0 and \
                self._arena.get("wt_mode") == "all" else None
            return {"nw": blk.norm.weight.detach(), "nb": blk.norm.bias.detach(),
                    "reduce": Fn.Weights(w, blk.patch_reduce.bias.detach(), wperm, wperm.shape[1], wperm_t,
                                         wperm.shape[0]),
                    "token": self._lin(blk.token_transform), "pos": blk.pos_embed.detach()[0]}
        return None

    def _layer_cfg(self, blk, grid):
        base = {"dtype": self.compute_dtype, "eps": 1e-6, "tokens": self.num_tokens}
        if isinstance(blk, Block):
            base.update(heads=blk.attn.num_heads, head_dim=blk.attn.head_dim, scale=float(blk.attn.scale),
                        hidden=blk.mlp.fc1.out_features)
        elif isinstance(blk, SpatialReductionPatchEmbedding):
            base.update(grid=grid, cout=blk.token_transform.out_features)
        return base

    def _embed_params(self, need_bwd=True):
        pe = self.patch_embed
        if self.embed_type == _TYPE_IS_EMBED:
            w = pe.proj.weight
            k = w.shape[1] * w.shape[2] * w.shape[3]
            if self.compute_dtype == torch.float32:
                wc, ld = w.detach().view(w.shape[0], k), k
            else:
                ld = (k + 7) // 8 * 8
                wc = self._arena.get("embed_wc")            # persistent: the pad columns stay zero, one cast-copy per forward
                if wc is None or wc.shape != (w.shape[0], ld) or wc.device != w.device:
                    wc = self._arena["embed_wc"] = torch.zeros((w.shape[0], ld), dtype=torch.bfloat16, device=w.device)
                K.relayout(w.detach(), wc, w.shape[0], 1, k, dst_ld=ld)             # fp32 [out, 588] -> bf16 [out, 592], pads stay zero
            return {"proj": Fn.Weights(w, pe.proj.bias.detach(), wc, ld), "pos": self.pos_embed.detach(),
                    "tokens": self.tokens.detach()}
        from .. import stem
        return stem.embed_params(self, need_bwd)      # (need_bwd: also the flipped weights of the data-gradient convolutions)

    def _run_forward(self, x, plan, with_patch, save):
        K.WRITE_SKIP[0] = plan.skip_writes       # (for the GEMMs of THIS forward only: other callers of kernels.gemm get whole outputs)
        try:
            return self._run_forward_impl(x, plan, with_patch, save)
        finally:
            K.WRITE_SKIP[0] = False

    def _run_forward_impl(self, x, plan, with_patch, save):
        a = self._arena
        if save and Fn.reset_ln_grads(self) and a.get("ln_parts_flat") is not None:      # (a backward died: see _run_backward)
            a["ln_parts_flat"].zero_()
        # engine.GraphedTrainStep(optimizer=..., deferred): the PREVIOUS replay's AdamW update opens this forward -- the head of the
        # arena (tokens, positional embedding, patch embedding, first stage) here, the rest (most parameters) on the side stream
        # beside the first stage, which does not read them; joined with side_prep in front of the first spatial reduction
        du = getattr(self, "_deferred_update", None)
        du_side = None
        if du is not None:
            opt_, lo_ = du
            n_ = a["flat"].numel()
            if (save and self.compute_dtype == torch.bfloat16 and a["tr"] is not None and Fn.OVERLAP and a["flat"].is_cuda and
                    0 < lo_ < n_):
                Fn.join_side()             # (a previous forward's side work)
                opt_.step_device(0, lo_)
                du_side = (opt_, lo_, n_)
            else:
                opt_.step_device(0, n_)
        if self.compute_dtype == torch.bfloat16:
            if Fn.OVERLAP and a["flat"].is_cuda:
                Fn.join_side()             # a previous forward's side work (if its backward never ran)
            # the bf16 weight shadow: vitres.optim.FlatAdamW writes it with every update (shadow_ok).  Otherwise it is re-cast per
            # forward in training mode; in eval mode (frozen weights: evaluation, candidate scoring) only when a parameter changed --
            # tracked by the parameters' version counters, which every in-place update under autograd's eyes bumps (optimizers,
            # load_state_dict, EMA copies; raw `.data` arithmetic is not seen: call invalidate_shadow() after such edits)
            if not a.get("shadow_ok"):
                ver = None if self.training else sum(p_._version for p_ in a["params"])
                if ver is None or a.get("shadow_ver") != ver:
                    K.cast_bf16(a["flat"], a["shadow"])
                a["shadow_ver"] = ver
        B = x.shape[0]
        tape = [] if save else None
        K.M_GROUPS[0] = plan.groups              # every GEMM of this forward deals the architecture groups to every XCD
        side_params = {}
        ecfg = {"dtype": self.compute_dtype, "patch": self.patch_size, "patches": self.patch_embed.num_patches,
                "dim": self.embed_dim, "tokens": self.num_tokens}
        ep = self._embed_params(save)
        ekeep = plan.k(plan.layers[0]["embed"])
        if self.embed_type == _TYPE_IS_EMBED:
            h, sv = Fn.embed0_fwd(x, ep, ecfg, ekeep, save, sample_map=plan.embed_map, col=plan.embed_col)
        else:
            from .. import stem
            h, sv = stem.embed_conv_fwd(self, x, ep, ecfg, ekeep, save)
        if save:
            tape.append(("embed", ep, ecfg, sv))
        if self.compute_dtype == torch.bfloat16 and a["tr"] is not None and save:
            # only the backward needs W^T: refresh it beside the forward, after the HBM-bound patch gather (joined at the
            # start of _run_backward)
            if Fn.OVERLAP and a["flat"].is_cuda:
                # also off the critical path, beside the first stage: the re-laid-out conv weights of the spatial reductions
                # (torch permute / cast / transpose) and the zeroing of the gradient arena the backward accumulates into
                fresh = a["gflat"] is not None and all(p_.grad is None for p_ in a["params"])

                def side_prep():
                    if du_side is not None:
                        du_side[0].step_device(du_side[1], du_side[2])
                    K.cast_transpose_batch(a["flat"], a["shadow_t"], a["tr"])
                    for blk_ in self.blocks:
                        if isinstance(blk_, SpatialReductionPatchEmbedding):
                            side_params[id(blk_)] = self._layer_params(blk_)
                    if fresh:
                        self._zero_grad_arena(a["gflat"], x.shape[0])
                Fn.on_side(side_prep)                  # enqueued (and side_params filled) by the flush_side() after the first branch
                a["gzeroed"] = fresh
            else:
                K.cast_transpose_batch(a["flat"], a["shadow_t"], a["tr"])
        grid = self.img_size // self.patch_size
        hp = {"nw": self.norm.weight.detach(), "nb": self.norm.bias.detach(), "cls": self._lin(self.cls_head)}
        if with_patch == 3:
            hp["dst"] = self._lin(self.dst_head)
        elif with_patch:
            hp["patch"] = self._lin(self.patch_head)
        hcfg = {"dtype": self.compute_dtype, "eps": 1e-6, "classes": self.num_classes, "tokens": self.num_tokens}
        hk = plan.k(plan.head)
        seq = []
        # forward-only (evaluation, candidate scoring): a block whose layer keep is 0 for EVERY sample -- a removed block of the
        # candidate -- is the identity (supernet_blocks.py:243,251) and is left out on the host: no launch at all; the LayerNorm its
        # predecessor fuses is then the one of the next block that runs
        dead = plan.dead_blocks if not save else None
        if not save and dead is None and _SKIP_DROPPED:
            dead = set()
            rows_h = plan.rows if plan.rows else None
            for j, L in enumerate(plan.layers[1:]):
                if L is not None and L.get("out") is not None and rows_h is not None and L["out"] < len(rows_h):
                    r_ = rows_h[L["out"]]
                    if not getattr(r_, "is_cuda", False) and bool((torch.as_tensor(r_) == 0).all()):
                        dead.add(j)
            plan.dead_blocks = dead
        for j, (blk, L) in enumerate(zip(self.blocks, plan.layers[1:])):
            if L is None:
                continue
            if dead and j in dead and isinstance(blk, Block):
                continue
            lazy = isinstance(blk, SpatialReductionPatchEmbedding) and save      # its weights are re-laid out by side_prep
            seq.append((blk, L, None if lazy else self._layer_params(blk), self._layer_cfg(blk, grid)))
            if not isinstance(blk, Block):
                grid //= 2

        def first_ln(i):
            """(w, b, keep, eps) of the LayerNorm layer i of `seq` starts with (the head's norm past the end): its producer
            computes it in its own epilogue when it can (functional.FUSE_LN)."""
            if i >= len(seq):
                return hp["nw"], hp["nb"], hk, hcfg["eps"]
            blk_, L_, p_, cfg_ = seq[i]
            if isinstance(blk_, Block):
                return p_["n1w"], p_["n1b"], plan.k(L_["embed"]), cfg_["eps"]
            return blk_.norm.weight.detach(), blk_.norm.bias.detach(), plan.k(L_["embed"]), cfg_["eps"]

        pre = None
        for i, (blk, L, p, cfg) in enumerate(seq):
            if isinstance(blk, Block):
                s1 = s2 = None
                if L["dp"] is not None:
                    s1, s2 = plan.scales[L["dp"]], plan.scales[L["dp"] + 1]
                ek, ka, km, ko = plan.k(L["embed"]), plan.k(L["attn"]), plan.k(L["mlp"]), plan.k(L["out"])
                h, sa, pre = Fn.attn_branch_fwd(h, p, cfg, ek, ka, ko, s1, save, pre=pre,
                                                next_ln=(p["n2w"], p["n2b"], ek, cfg["eps"]))
                if i == 0:
                    Fn.flush_side()        # side_prep: enqueued after the main chain's first kernels (functional.SIDE_DEFER)
                h, sm, pre = Fn.mlp_branch_fwd(h, p, cfg, ek, km, ko, s2, save, pre=pre, next_ln=first_ln(i + 1))
                if save:
                    tape.append(("block", blk, p, cfg, (ek, ka, km, ko, s1, s2), sa, sm))
            else:
                if p is None:
                    Fn.flush_side()
                    p = side_params.get(id(blk))
                    if p is None:
                        p = self._layer_params(blk)
                    elif not side_params.get("joined"):
                        Fn.join_side()     # its weights were re-laid out on the side stream
                        side_params["joined"] = True
                ek, nk = plan.k(L["embed"]), plan.k(L["new"])
                h, sv = Fn.sr_fwd(h, p, cfg, ek, nk, save, pre=pre)
                pre = None
                if save:
                    tape.append(("sr", blk, p, cfg, (ek, nk), sv))
        cls, pat, sv = Fn.head_fwd(h, hp, hcfg, hk, with_patch, save, pre=pre)
        if save:
            tape.append(("head", hp, hcfg, hk, sv))
        return cls, pat, tape

    # ---- backward ----------------------------------------------------------------------------------
    def _run_backward(self, tape, plan, dcls, dpat, ready=False):
        a = self._arena
        K.M_GROUPS[0] = plan.groups
        params = a["params"]
        fresh = all(p.grad is None for p in params)
        if a["gflat"] is None:
            a["gflat"] = torch.zeros_like(a["flat"])
        # grads already live in the arena (no zero_grad since the last backward): use a scratch arena so that
        # autograd's accumulation adds a separate buffer
        a["gcur"] = a["gflat"] if fresh else torch.zeros_like(a["flat"])
        if fresh and not a.pop("gzeroed", False):      # (the forward may have zeroed it on the side stream already)
            self._zero_grad_arena(a["gcur"], plan.batch)
        a["gzeroed"] = False
        # leftovers of a backward that raised: weight-gradient calls collected for a block and never launched (join_side would
        # launch them first -- into the arena just zeroed) and LayerNorm partial rows never folded
        stale = Fn.reset_ln_grads(self)
        Fn.claim_pending(self)
        Fn.join_side()                     # transposed weight shadows (issued beside the forward)
        if Fn.LN_COPIES > 1:
            self._ln_parts()
            if stale:
                a["ln_parts_flat"].zero_()
        st = {"rtape": list(reversed(tape)), "plan": plan, "dcls": dcls, "dpat": dpat, "i": 0, "g": None, "gt": None,
              "ready": ready}
        # _bwd_split = j: stop after the head and blocks[j:]; the rest runs in resume_backward() (a second hipGraph, so that the
        # all-reduce of the finished tail of the gradient arena overlaps it -- engine.GraphedTrainStep)
        cuts = getattr(self, "_bwd_split", None)
        cuts = [] if cuts is None else ([cuts] if isinstance(cuts, int) else list(cuts))
        stops = []
        for cut in cuts:                   # head + the entries of blocks[cut:] form a prefix of the reversed tape
            late = {id(b) for b in list(self.blocks)[cut:]}
            stops.append(sum(1 for e in st["rtape"] if e[0] == "head" or (e[0] in ("block", "sr") and id(e[1]) in late)))
        assert stops == sorted(stops), "_bwd_split: block indices must descend (the backward walks the blocks from the end)"
        st["stops"] = stops + [len(st["rtape"])]
        self._bwd_loop(st, st["stops"].pop(0))
        self._bwd_state = st if st["i"] < len(st["rtape"]) else None
        return [self._gview(p) for p in params]

    def resume_backward(self):
        """Next part of a backward that was split by _bwd_split (gradients land in the same arena views); returns True while
        further parts are pending."""
        st = self._bwd_state
        if st is None:
            raise RuntimeError("no split backward is pending")
        K.M_GROUPS[0] = st["plan"].groups
        self._bwd_loop(st, st["stops"].pop(0))
        if st["i"] >= len(st["rtape"]):
            self._bwd_state = None
        return self._bwd_state is not None

    def split_plan(self, parts=2):
        """Where to cut the backward for gradient-exchange overlap.  parts=2: (first block index of part 1, first element of
        the gradient arena that part 1 completes) -- head + the last stage (+ the spatial reduction in front of it), whose
        parameters form the tail of the arena.  parts=3 (or more): a list of such pairs, one cut in front of every spatial
        reduction counted from the end; pair k's arena range [start_k, start_{k-1}) is final after part k.  None if the layout
        does not allow it."""
        a = self._arena
        blocks = list(self.blocks)
        sr = [j for j in range(len(blocks)) if isinstance(blocks[j], SpatialReductionPatchEmbedding)]
        cuts = list(reversed(sr))[:max(parts - 1, 1)] if sr else [len(blocks) // 2]
        heads = list(self.norm.parameters()) + list(self.cls_head.parameters()) + \
            (list(self.patch_head.parameters()) if self.patch_head is not None else []) + \
            (list(self.dst_head.parameters()) if self.dst_head is not None else [])
        out = []
        for cut in cuts:
            tail = [p for m in blocks[cut:] for p in m.parameters()] + heads
            tail_ids = {id(p) for p in tail}
            if not tail:
                return None
            start = min(a["offsets"][a["index"][id(p)]][0] for p in tail)
            for p, (off, _) in zip(a["params"], a["offsets"]):
                if (off >= start) != (id(p) in tail_ids):
                    return None                                 # tail parameters are not contiguous in the arena
            out.append((cut, start))
        return out[0] if parts == 2 else out

    def _bwd_loop(self, st, stop):
        K.WRITE_SKIP[0] = st["plan"].skip_writes
        try:
            return self._bwd_loop_impl(st, stop)
        finally:
            K.WRITE_SKIP[0] = False

    def _bwd_loop_impl(self, st, stop):
        a = self._arena
        gv = self._gview
        dev = a["flat"].device
        rtape, plan, dcls, dpat = st["rtape"], st["plan"], st["dcls"], st["dpat"]
        g = st["g"]
        ekeep0 = plan.k(plan.layers[0]["embed"])

        def consumer_cast(i):
            """(DropPath scale, prefix keep) with which the entry after rtape[i] turns the gradient it receives into its
            compute-dtype branch gradient: LayerNorm backward emits that tensor in the same pass (vr_ln_bwd gt_out)."""
            if i + 1 >= len(rtape) or not Fn.FUSE_CAST:
                return None
            e = rtape[i + 1]
            if e[0] == "block":
                return (e[4][5], e[4][3])
            if e[0] == "sr":
                return (None, e[4][1])
            if e[0] == "embed":
                return (None, ekeep0)
            return None
        gt = st["gt"]
        tail_aux = False
        parts = a.get("ln_parts") if Fn.LN_COPIES > 1 else None

        def with_parts(grads, *pairs):
            if parts is not None:
                for key, w in pairs:
                    grads[key + ".part"] = parts[id(w)]
            return grads
        for ti in range(st["i"], stop):
            entry = rtape[ti]
            kind = entry[0]
            if kind == "head":
                _, hp, hcfg, hk, sv = entry
                grads = {"nw": gv(self.norm.weight), "nb": gv(self.norm.bias), "cls.w": gv(self.cls_head.weight),
                         "cls.b": gv(self.cls_head.bias)}
                if self.patch_head is not None:
                    grads["patch.w"], grads["patch.b"] = gv(self.patch_head.weight), gv(self.patch_head.bias)
                if self.dst_head is not None:
                    grads["dst.w"], grads["dst.b"] = gv(self.dst_head.weight), gv(self.dst_head.bias)
                with_parts(grads, ("nw", self.norm.weight))
                nc = consumer_cast(ti)
                g = Fn.head_bwd(dcls, dpat, sv, hp, grads, hcfg, hk, next_cast=nc, ready=st.get("ready", False))
                g, gt = g if nc is not None else (g, None)
            elif kind == "block":
                _, blk, p, cfg, (ek, ka, km, ko, s1, s2), sa, sm = entry
                grads = {"n1w": gv(blk.norm1.weight), "n1b": gv(blk.norm1.bias), "n2w": gv(blk.norm2.weight),
                         "n2b": gv(blk.norm2.bias), "qkv.w": gv(blk.attn.qkv.weight), "qkv.b": gv(blk.attn.qkv.bias),
                         "proj.w": gv(blk.attn.proj.weight), "proj.b": gv(blk.attn.proj.bias),
                         "fc1.w": gv(blk.mlp.fc1.weight), "fc1.b": gv(blk.mlp.fc1.bias),
                         "fc2.w": gv(blk.mlp.fc2.weight), "fc2.b": gv(blk.mlp.fc2.bias)}
                with_parts(grads, ("n1w", blk.norm1.weight), ("n2w", blk.norm2.weight))
                g, gt = Fn.mlp_branch_bwd(g, sm, p, grads, cfg, ek, km, ko, s2, gt=gt, next_cast=(s1, ko))      # gt: attention branch's
                nc = consumer_cast(ti)
                g = Fn.attn_branch_bwd(g, sa, p, grads, cfg, ek, ka, ko, s1, gt=gt, next_cast=nc)
                g, gt = g if nc is not None else (g, None)
            elif kind == "sr":
                _, blk, p, cfg, (ek, nk), sv = entry
                co, ci = blk.patch_reduce.weight.shape[0], blk.patch_reduce.weight.shape[1]
                nt = self.num_tokens
                ztmp = K.zero_(torch.empty(co * 9 * ci + (nt + blk.num_patches) * co, dtype=torch.float32, device=dev))   # one fill
                wtmp = ztmp[:co * 9 * ci].view(co, 9 * ci)
                ptmp = ztmp[co * 9 * ci:].view(nt + blk.num_patches, co)
                def finish(blk=blk, wtmp=wtmp, ptmp=ptmp, co=co, ci=ci, nt=nt):     # runs on the stream of the weight gradients
                    K.relayout(wtmp, gv(blk.patch_reduce.weight), co, 9, ci)         # [co, (kh, kw), ci] -> [co, ci, kh, kw]
                    gv(blk.pos_embed).copy_(ptmp[nt:].unsqueeze(0))
                grads = {"nw": gv(blk.norm.weight), "nb": gv(blk.norm.bias), "token.w": gv(blk.token_transform.weight),
                         "token.b": gv(blk.token_transform.bias), "reduce.b": gv(blk.patch_reduce.bias),
                         "reduce.w": wtmp, "pos_sum": ptmp, "finish": finish}
                with_parts(grads, ("nw", blk.norm.weight))
                nc = consumer_cast(ti)
                g = Fn.sr_bwd(g, sv, p, grads, cfg, ek, nk, gt=gt, next_cast=nc)
                g, gt = g if nc is not None else (g, None)
            elif kind == "embed":
                _, ep, ecfg, sv = entry
                ekeep = plan.k(plan.layers[0]["embed"])
                if self.embed_type == _TYPE_IS_EMBED:
                    w = self.patch_embed.proj.weight
                    ld = ep["proj"].ld
                    k = w.numel() // w.shape[0]
                    wt = gv(w).view(w.shape[0], k) if ld == k else K.zero_(torch.empty((w.shape[0], ld), dtype=torch.float32,
                                                                                        device=dev))
                    grads = {"proj.w": wt, "proj.b": gv(self.patch_embed.proj.bias), "pos": gv(self.pos_embed)}

                    def tail(wgrad=True, pos=True, g=g, sv=sv, ep=ep, grads=grads, ecfg=ecfg, ekeep=ekeep, gt=gt, wt=wt, w=w, k=k, ld=ld):
                        Fn.embed0_bwd(g, sv, ep, grads, ecfg, ekeep, gt=gt, wgrad=wgrad, pos=pos)
                        if wgrad and ld != k:
                            K.relayout(wt, gv(w), w.shape[0], 1, k, src_ld=ld)   # drop the pad columns
                        if pos:
                            gv(self.tokens).copy_(gv(self.pos_embed)[:, 0:self.num_tokens, :])
                    if Fn.TAIL_AUX and Fn.OVERLAP and g.is_cuda:
                        # nothing downstream but the optimizer: the projection's weight gradient on the auxiliary stream beside the
                        # first block's last weight gradient, the small reductions meanwhile on the main stream
                        Fn.flush_wgrads()
                        Fn.on_aux("tail", tail, g, gt, wt, *sv)
                        tail_aux = True
                    else:
                        tail()
                else:
                    from .. import stem
                    stem.embed_conv_bwd(self, g, sv, ep, ecfg, ekeep, gv, gt=gt)
                    gv(self.tokens).copy_(gv(self.pos_embed)[:, 0:self.num_tokens, :])
        if tail_aux:                       # (behind the patch-embedding weight gradient on the auxiliary stream: the graph executor
            Fn.on_aux("tail", Fn.flush_ln_grads)   #  runs a third branch on the weight gradients' queue, IN FRONT of the last groups)
        else:
            Fn.flush_ln_grads()            # LayerNorm weight / bias gradients of this part: partial rows -> arena
        if stop >= len(rtape) or getattr(self, "_bwd_join_parts", True):
            Fn.join_side()                 # weight-gradient GEMMs trail on the side stream (functional.on_side); an intermediate
                                           # stop joins too unless the next part follows in the same capture (_bwd_join_parts)
        st["i"], st["g"], st["gt"] = stop, g, gt


# ---- registry factories (reference :480-577) -------------------------------------------------------
def _factory(**fixed):
    def make(pretrained=False, **kwargs):
        model = FlexibleDistillVisionTransformerSR(patch_size=14, **fixed, **kwargs)
        model.default_cfg = _cfg()
        return model
    return make


@register_model
def flexible_vit_sr_distill_patch14_224(pretrained=False, **kwargs):
    return _factory(distill_token=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_224(pretrained=False, **kwargs):
    return _factory(distill_token=False)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_distill_patch14_224_supernet(pretrained=False, **kwargs):
    return _factory(distill_token=True, supernet=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_224_supernet(pretrained=False, **kwargs):
    return _factory(distill_token=False, supernet=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_224_patch_output(pretrained=False, **kwargs):
    return _factory(distill_token=False, patch_output=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_224_patch_output_supernet(pretrained=False, **kwargs):
    return _factory(distill_token=False, supernet=True, patch_output=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_280_patch_output(pretrained=False, **kwargs):
    return _factory(img_size=280, distill_token=False, patch_output=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_336_patch_output(pretrained=False, **kwargs):
    return _factory(img_size=336, distill_token=False, patch_output=True)(pretrained, **kwargs)


@register_model
def flexible_vit_sr_patch14_392_patch_output(pretrained=False, **kwargs):
    return _factory(img_size=392, distill_token=False, patch_output=True)(pretrained, **kwargs)

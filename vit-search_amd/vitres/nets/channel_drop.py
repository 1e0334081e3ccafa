"""ChannelDrop: the weight-sharing prefix mask of the supernet, carried as per-sample keep counts.

Host-side mirror of reference nets/channel_drop.py:16-197.  The reference materialises a bool
(B,1,C) mask on the device and multiplies activations by it; every such mask is a channel *prefix*
(channel_drop.py:153-154), so here a mask is the int vector keep[b] = number of leading channels
sample b keeps, and the multiply is folded into the HIP kernels' epilogues (or skipped as work).
The sampling protocol is reproduced exactly -- one `torch.randperm` on the CPU default generator
per training forward, table rows cycling through the sorted choices, warm-up schedule -- so the
keep vectors are bit-identical to the reference's masks for the same RNG state.
"""
import math

import numpy as np
import torch
import torch.nn as nn

_NUM_WARMUP_EPOCHS_CHANNEL = 5


class ChannelDrop(nn.Module):
    def __init__(self, num_channels_to_keep=None, num_warmup_epochs=_NUM_WARMUP_EPOCHS_CHANNEL,
                 example_per_arch=None, single_arch=False):
        super().__init__()
        assert num_channels_to_keep is not None
        assert example_per_arch is not None
        assert isinstance(num_channels_to_keep, np.ndarray), 'num_channels_to_keep data type error'
        self.num_channels_to_keep = np.sort(num_channels_to_keep)[::-1]
        self.epoch_now = None
        self.num_warmup_epochs = num_warmup_epochs
        self.example_per_arch = example_per_arch
        self.single_arch = single_arch
        self.table = None                 # int64 keep count per table row (reference: self.mask)
        self.num_layer_config = None
        self.fixed_keep = None            # debug hook (reference: self.fixed_mask)

    # ---- schedule ---------------------------------------------------------------------------
    def _build_table(self, batch, channels):
        """reference set_mask (channel_drop.py:114-157)."""
        keep = self.num_channels_to_keep
        assert batch % self.example_per_arch == 0, 'Batch size is not divisible by sub-batch size (examples per arch).'
        assert (keep <= channels).all(), 'Some elements in num_channels_to_keep is larger than channel size.'
        assert max(keep) == channels, 'Maximum channel not in num_channels_to_keep'
        assert batch >= len(keep), 'The batch size is smaller than the number of channels to keep.'
        if self.num_warmup_epochs == 0:
            nlc = len(keep)
        else:
            nlc = min(1 + math.floor(self.epoch_now * (len(keep) - 1) / self.num_warmup_epochs), len(keep))
            nlc = max(nlc, 1)
        self.num_layer_config = nlc
        cycles = 1 if self.single_arch else math.ceil((batch // self.example_per_arch) / nlc)
        self.table = torch.tensor([int(keep[i % nlc]) for i in range(nlc * cycles)], dtype=torch.int64)

    def sample_keep(self, batch, channels, training=None):
        """keep[b] (CPU int64 tensor) for one forward; consumes the CPU RNG exactly like the reference
        forward_mask (channel_drop.py:93-111)."""
        training = self.training if training is None else training
        if self.fixed_keep is not None:
            return torch.full((batch,), int(self.fixed_keep), dtype=torch.int64)
        if not training:
            return torch.full((batch,), channels, dtype=torch.int64)      # all-true mask (:84-88)
        if self.table is None:
            self._build_table(batch, channels)
        rows = self.table[torch.randperm(self.table.shape[0])]
        if self.single_arch:
            return rows[0:1].repeat(batch)
        assert batch % self.example_per_arch == 0, \
            'In forward(), batch size is not divisible by sub-batch size (examples per arch).'
        return rows[:batch // self.example_per_arch].repeat(self.example_per_arch)

    def sample_groups(self, batch, channels, training=None):
        """The same draw as sample_keep (one torch.randperm on the CPU generator, reference forward_mask :93-111) returned as
        the short numpy vector g with keep[b] = g[b % len(g)]: the first batch / example_per_arch permuted table rows (one per
        architecture group; `rows[:G].repeat(epa)`), or a single entry (single_arch, eval, fixed mask)."""
        import numpy as np
        if training is None:
            training = self.training
        if self.fixed_keep is not None:
            return np.array([int(self.fixed_keep)], dtype=np.int64)
        if not training:
            return np.array([channels], dtype=np.int64)
        if self.table is None:
            self._build_table(batch, channels)
        tab = self.table.numpy()           # zero-copy view (the table may also have been built by sample_keep)
        perm = torch.randperm(self.table.shape[0]).numpy()
        if self.single_arch:
            return tab[perm[0:1]]
        assert batch % self.example_per_arch == 0, \
            'In forward(), batch size is not divisible by sub-batch size (examples per arch).'
        return tab[perm[:batch // self.example_per_arch]]

    def forward(self, x):
        raise RuntimeError('ChannelDrop has no standalone device op in the HIP path: its multiply is fused into the '
                           'neighbouring kernels; use sample_keep() (host) and the parent model forward.')

    def set_epoch(self, epoch_now):
        self.epoch_now = epoch_now
        self.reset_mask()

    def reset_mask(self):
        self.table = None
        self.fixed_keep = None
        self.num_layer_config = None

    # ---- debug hooks (reference :173-190) -------------------------------------------------------
    def set_fixed_mask(self, mask):
        """Accepts the reference's (1,1,C) bool mask (must be a channel prefix) or an int keep count."""
        if isinstance(mask, int):
            self.fixed_keep = mask
            return
        assert len(mask.shape) == 3 and mask.shape[0] == 1
        m = mask.reshape(-1).to(torch.bool).cpu()
        k = int(m.sum())
        if not bool(m[:k].all()):
            raise ValueError('only prefix masks are supported by the HIP path')
        self.fixed_keep = k

    def set_random_fixed_mask(self):
        if self.table is None:
            self._build_table(self.example_per_arch * len(self.num_channels_to_keep), int(max(self.num_channels_to_keep)))
        self.fixed_keep = int(self.table[torch.randperm(self.table.shape[0])][0])

    def extra_repr(self):
        s = 'num_channels_to_keep={}, num_warmup_epochs={}, example_per_arch={}'.format(
            self.num_channels_to_keep, self.num_warmup_epochs, self.example_per_arch)
        if self.single_arch:
            s += ', single_arch={}'.format(self.single_arch)
        return s

"""MAC estimator of a `network_def` (reference network_utils/compute_flop_mac.py:227-307, MAC mode): the search
constraint of evo_search.py:208-209 and this repo's definition of algorithmic FLOPs (6 x MAC per trained image)."""

_T_EMBED, _T_TRANS, _T_HEAD, _T_SR, _T_CONV, _T_FLEX = 0, 1, 2, 3, 4, 5


class ComputationEstimator:
    def __init__(self, distill, input_resolution, patch_size, num_in_channels=3, return_mac=True):
        if not return_mac:
            raise NotImplementedError('only the MAC mode (return_mac=True) is restated')
        assert input_resolution % patch_size == 0
        self.distill, self.input_resolution, self.patch_size = distill, input_resolution, patch_size
        self.num_in_channels, self.return_mac, self.sr_patch_size = num_in_channels, return_mac, 2

    def __repr__(self):
        return '(distill={}, input_resolution={}, patch_size={}, sr_patch_size={}, num_in_channels={}, return_mac={})'.format(
            self.distill, self.input_resolution, self.patch_size, self.sr_patch_size, self.num_in_channels, self.return_mac)

    def __call__(self, network_def):
        grid = self.input_resolution // self.patch_size
        nt = 2 if self.distill else 1
        n = nt + grid * grid
        e = network_def[0]
        assert e[0] in (_T_EMBED, _T_CONV, _T_FLEX), 'Network def error: embedding'
        dim = e[1]
        if e[0] == _T_EMBED:
            total = dim * self.num_in_channels * self.patch_size ** 2 * grid * grid
        else:
            mid = e[2] if e[0] == _T_FLEX else 24
            ps = self.patch_size // 2
            total = self.num_in_channels * mid * 9 * 112 * 112 + 2 * mid * mid * 9 * 112 * 112 + dim * mid * ps * ps * grid * grid
        for b in network_def:
            if b[0] == _T_TRANS:
                assert b[1][0] == b[2][0] == dim
                if b[3]:
                    c, h, d = b[1]
                    total += c * h * d * 3 * n + 2 * n * n * h * d + n * h * d * c + 2 * n * c * b[2][1]
            elif b[0] == _T_SR:
                assert b[1] == dim and grid % self.sr_patch_size == 0
                grid //= self.sr_patch_size
                total += grid * grid * b[2] * 9 * b[1] + b[1] * b[2] * nt
                n, dim = nt + grid * grid, b[2]
        return total + dim * network_def[-1][2] * nt


def train_flops_per_image(network_def, resolution=224, patch_size=14):
    """6 x (estimator MACs + training-mode patch-head MACs): forward + backward GEMM/conv FLOPs."""
    grid = resolution // patch_size
    for b in network_def:
        if b[0] == _T_SR:
            grid //= 2
    mac = ComputationEstimator(False, resolution, patch_size)(network_def) + grid * grid * network_def[-1][1] * network_def[-1][2]
    return 6 * mac

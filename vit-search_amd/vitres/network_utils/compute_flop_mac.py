"""MAC / FLOP estimator of a `network_def` (reference network_utils/compute_flop_mac.py:227-307 and its helpers :53-194): the
search constraint of evo_search.py:208-209 and this repo's definition of algorithmic FLOPs (6 x MAC per trained image)."""

_T_EMBED, _T_TRANS, _T_HEAD, _T_SR, _T_CONV, _T_FLEX = 0, 1, 2, 3, 4, 5
# per-element FLOP prices of the non-GEMM terms the FLOP mode adds (compute_flop_mac.py:36-38)
_SOFTMAX, _LAYER_NORM, _GELU = 5, 5, 8


class ComputationEstimator:
    """return_mac=True: multiply-accumulates of the GEMMs and convolutions only.  return_mac=False: FLOPs -- every
    multiply-add counts 2, and biases, softmax, score scaling, residual adds, LayerNorm and GELU are added at the reference's
    per-element prices (one 3-factor scheme, compute_flop_mac.py:55-57: multiply-add x2, bias +1, everything else +1)."""

    def __init__(self, distill, input_resolution, patch_size, num_in_channels=3, return_mac=True):
        assert input_resolution % patch_size == 0
        self.distill, self.input_resolution, self.patch_size = distill, input_resolution, patch_size
        self.num_in_channels, self.return_mac, self.sr_patch_size = num_in_channels, return_mac, 2

    def __repr__(self):
        return '(distill={}, input_resolution={}, patch_size={}, sr_patch_size={}, num_in_channels={}, return_mac={})'.format(
            self.distill, self.input_resolution, self.patch_size, self.sr_patch_size, self.num_in_channels, self.return_mac)

    def __call__(self, network_def):
        ma, bf, mf = (1, 0, 0) if self.return_mac else (2, 1, 1)     # multiply-add, bias, everything-else factors
        grid = self.input_resolution // self.patch_size
        nt = 2 if self.distill else 1
        n = nt + grid * grid
        e = network_def[0]
        assert e[0] in (_T_EMBED, _T_CONV, _T_FLEX), 'Network def error: embedding'
        dim, cin, P = e[1], self.num_in_channels, grid * grid
        if e[0] == _T_EMBED:
            total = dim * cin * self.patch_size ** 2 * P * ma + dim * P * bf
        else:
            mid = e[2] if e[0] == _T_FLEX else 24
            ps, px = self.patch_size // 2, 112 * 112                 # (the reference assumes a 224-px input here)
            total = (cin * mid * 9 * px + 2 * mid * mid * 9 * px + dim * mid * ps * ps * P) * ma + (3 * mid * px + dim * P) * bf
        total += dim * n * bf                                        # position embedding
        for b in network_def:
            if b[0] == _T_TRANS:
                assert b[1][0] == b[2][0] == dim
                if b[3]:
                    c, h, d = b[1]
                    f = b[2][1]
                    total += (c * h * d * 3 * n + 2 * n * n * h * d + n * h * d * c + 2 * n * c * f) * ma
                    total += (3 * h * d * n + 2 * n * c + n * f) * bf
                    total += (n * n * h * (_SOFTMAX + 1) + 2 * n * c * (1 + _LAYER_NORM) + n * f * _GELU) * mf
            elif b[0] == _T_SR:
                assert b[1] == dim and grid % self.sr_patch_size == 0
                grid //= self.sr_patch_size
                cout = b[2]
                total += grid * grid * cout * 9 * dim * ma + 2 * grid * grid * cout * bf + grid * grid * cout * _LAYER_NORM * mf
                total += (dim * cout * ma + cout * bf + dim * (_LAYER_NORM + 1) * mf) * nt
                n, dim = nt + grid * grid, cout
        classes = network_def[-1][2]
        return total + (dim * classes * ma + n * classes * bf + dim * _LAYER_NORM * mf) * nt


def train_flops_per_image(network_def, resolution=224, patch_size=14):
    """6 x (estimator MACs + training-mode patch-head MACs): forward + backward GEMM/conv FLOPs."""
    grid = resolution // patch_size
    for b in network_def:
        if b[0] == _T_SR:
            grid //= 2
    mac = ComputationEstimator(False, resolution, patch_size)(network_def) + grid * grid * network_def[-1][1] * network_def[-1][2]
    return 6 * mac

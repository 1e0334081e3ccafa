"""Sub-network weight inheritance: slice a supernet state_dict down to a sub-network's shapes
(reference nets/net_utils.py:10-57; used by evo_search.py:263-264 and main.py:418-424)."""
import torch


def get_qkv_subnet_state_dict(qkv_source, qkv_subnet):
    """q, k and v each occupy one third of the rows: keep the leading rows of every third."""
    assert qkv_subnet.shape[0] % 3 == 0 and qkv_source.shape[0] % 3 == 0
    n_sub, n_src = qkv_subnet.shape[0] // 3, qkv_source.shape[0] // 3
    out = torch.cat([qkv_source[j * n_src: j * n_src + n_sub] for j in range(3)], dim=0)
    if qkv_subnet.dim() == 2:
        out = out[:, :qkv_subnet.shape[1]]
    return out


def get_sub_state_dict(source_dict, sub_dict):
    out = {}
    for key, ref in sub_dict.items():
        src = source_dict[key]
        if 'qkv' in key:
            out[key] = get_qkv_subnet_state_dict(src, ref)
        elif ref.dim() == 0:
            out[key] = ref.clone()              # counters keep the fresh sub-net's value (reference :53-54)
        elif ref.dim() == 4:
            assert ref.shape[2] == src.shape[2] and ref.shape[3] == src.shape[3]
            out[key] = src[:ref.shape[0], :ref.shape[1]]
        elif ref.dim() in (1, 2, 3):
            out[key] = src[tuple(slice(0, n) for n in ref.shape)]
        else:
            raise ValueError
    return out

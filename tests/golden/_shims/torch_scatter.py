"""Stand-in for torch_scatter 2.0.7 (absent from this image; Dockerfile:74 of the
reference pins it).  Used ONLY by tests/golden/make_golden.py to run the
reference's own lav/models/point_pillar.py on CPU.  Restates the CPU kernels'
published semantics: sequential accumulation in index order, output length
index.max()+1, mean = sum / count (count clamped to >= 1), max returns
(values, argmax) with empty rows left at 0 / index.size.
"""
import numpy as np
import torch


def scatter_mean(src, index, dim=0):
    assert dim == 0
    s = src.detach().cpu().numpy().astype(np.float32)
    idx = index.detach().cpu().numpy()
    n = int(idx.max()) + 1 if idx.size else 0
    out = np.zeros((n,) + s.shape[1:], np.float32)
    np.add.at(out, idx, s)                      # unbuffered, in index order
    cnt = np.zeros((n,), np.float32)
    np.add.at(cnt, idx, np.float32(1))
    cnt = np.maximum(cnt, 1)
    out = out / cnt.reshape((n,) + (1,) * (s.ndim - 1))
    return torch.from_numpy(out.astype(np.float32)).to(src.device)


def scatter_max(src, index, dim=0):
    assert dim == 0
    n = int(index.max()) + 1 if index.numel() else 0
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    # argmax is not used by the reference (point_pillar.py:33 takes [0]); keep the tuple shape
    arg = torch.full_like(out, src.shape[0], dtype=torch.long)
    return out, arg

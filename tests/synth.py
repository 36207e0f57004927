"""Seeded synthetic documents in the reference's candidate-tensor layout
(svecalign/utils/embedding_utils.py:149-156): layer k row i = embedding of segments i-k..i,
rows i < k are zero.  The target is a noisy copy of the source so that a monotone path exists."""
import numpy as np


def make_pair(N, M, K, d, seed, noise=0.5, dtype=np.float32, zero_rows=0, deletions=0):
    rng = np.random.default_rng(seed)
    L = max(N, M) + K + deletions
    base = rng.standard_normal((L, d)).astype(np.float32)
    tgt = base + noise * rng.standard_normal((L, d)).astype(np.float32)
    if deletions:  # drop a few target segments so that non 1-1 alignments appear
        keep = np.ones(L, bool)
        keep[rng.choice(np.arange(2, max(3, M - 2)), size=min(deletions, max(1, M - 4)), replace=False)] = False
        tgt = tgt[keep]

    def layers(b, n):
        out = np.zeros((K, n, d), np.float32)
        cs = np.concatenate([np.zeros((1, d), np.float32), np.cumsum(b[:n].astype(np.float64), axis=0).astype(np.float32)])
        for k in range(K):
            if n > k:
                out[k, k:] = cs[k + 1:n + 1] - cs[:n - k]
        return out

    v0, v1 = layers(base, N), layers(tgt, M)
    if zero_rows:  # PAD / ignored candidates (embedding_utils.py:194-201)
        for v in (v0, v1):
            idx = rng.choice(v.shape[1], size=min(zero_rows, v.shape[1]), replace=False)
            v[rng.integers(0, K, size=len(idx)), idx] = 0.0
    if dtype == np.float16:
        return v0.astype(np.float16), v1.astype(np.float16)
    return v0, v1


def round_bf16(a):
    """float32 -> nearest-even bfloat16 values, returned as float32"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def alignment_types(a):
    """vecalign.py:154-162 order"""
    return [(x, y) for x in range(1, a) for y in range(1, a) if x + y <= a]


def make_pair_device(N, M, K, d, seed, device, dtype):
    """make_pair's layout generated on the GPU (torch), for batches too large to build on the host:
    layer k row i = sum of base rows i-k..i (rows i < k zero); target = noisy copy of the source.
    Returns torch tensors [K][N][d], [K][M][d] of `dtype`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = max(N, M)
    base = torch.randn((L, d), generator=g, device=device, dtype=torch.float32)
    tgt = base + 0.5 * torch.randn((L, d), generator=g, device=device, dtype=torch.float32)

    def layers(b, n):
        cs = torch.cat([torch.zeros((1, d), device=device, dtype=torch.float64), torch.cumsum(b[:n].double(), 0)])
        out = torch.zeros((K, n, d), device=device, dtype=torch.float32)
        for k in range(K):
            out[k, k:] = (cs[k + 1:n + 1] - cs[:n - k]).float()
        return out.to(dtype).contiguous()

    return layers(base, N), layers(tgt, M)

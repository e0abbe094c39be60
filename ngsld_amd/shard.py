"""Row sharding of the pair space across the GPUs of one node.

Every SNP pair is independent given the read-only GL matrix (the reference already exploits this per s1,
ngsLD.cpp:159-186), so ranks take contiguous s1 ranges balanced by PAIR COUNT and never exchange data on
the compute path.  The only collective is the one-off distribution of the GL matrix (RCCL broadcast over
xGMI), done by `broadcast_matrix` before any timed region.
"""
from __future__ import annotations

import numpy as np


def pos_dist_from_positions(chrs: list[str], pos: np.ndarray) -> np.ndarray:
    """pos_dist as read_dist builds it (shared/read_data.cpp:198-211): gap to the previous site, the
    first site's gap counted from 0, INFINITY at a chromosome change."""
    n = len(pos)
    pd = np.empty(n, dtype=np.float64)
    if n == 0:
        return pd
    pd[0] = float(pos[0])
    pd[1:] = np.diff(pos).astype(np.float64)
    c = np.asarray(chrs)
    pd[1:][c[1:] != c[:-1]] = np.inf
    return pd


def row_ends(pos_dist: np.ndarray, max_kb_dist: int, max_snp_dist: int) -> np.ndarray:
    """Exclusive end of the s2 walk of every row (ngsLD.cpp:240-262), min_maf = 0, integer gaps."""
    n = len(pos_dist)
    s = np.arange(n, dtype=np.int64)
    end = np.full(n, n, dtype=np.int64)
    if max_kb_dist > 0:
        brk = np.isinf(pos_dist)
        brk[0] = False
        seg = np.cumsum(brk)
        gaps = np.where(np.isinf(pos_dist), 0.0, pos_dist)
        gaps[0] = 0.0                      # the first site's gap (pos - 0) belongs to no pair
        cum = np.cumsum(gaps)
        # sites of later segments are "infinitely far": shift each segment beyond any window
        key = cum + seg * (cum[-1] + max_kb_dist * 1000.0 + 1.0)
        end = np.searchsorted(key, key + max_kb_dist * 1000.0, side="right").astype(np.int64)
    if max_snp_dist > 0:
        end = np.minimum(end, s + 1 + max_snp_dist)
    return np.maximum(np.minimum(end, n), np.minimum(s + 1, n))


def row_pair_counts(pos_dist: np.ndarray, max_kb_dist: int, max_snp_dist: int) -> np.ndarray:
    n = len(pos_dist)
    return row_ends(pos_dist, max_kb_dist, max_snp_dist) - (np.arange(n, dtype=np.int64) + 1)


def split_rows(pair_counts: np.ndarray, world_size: int) -> list[tuple[int, int]]:
    """Contiguous row ranges [lo, hi) per rank with (nearly) equal pair counts."""
    n = len(pair_counts)
    cum = np.concatenate([[0], np.cumsum(pair_counts.astype(np.int64))])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world_size):
        bounds.append(int(np.searchsorted(cum, total * r / world_size, side="left")))
    bounds.append(n)
    bounds = np.maximum.accumulate(np.minimum(bounds, n))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world_size)]


def split_rows_weighted(weights: np.ndarray, world_size: int) -> list[tuple[int, int]]:
    """Contiguous row ranges [lo, hi) per rank with (nearly) equal total WEIGHT -- e.g. a row's pairs times its estimated
    EM iterations per pair (bench.py --balance work): the same cut as split_rows, on float weights."""
    n = len(weights)
    cum = np.concatenate([[0.0], np.cumsum(np.asarray(weights, dtype=np.float64))])
    total = float(cum[-1])
    bounds = [0]
    for r in range(1, world_size):
        bounds.append(int(np.searchsorted(cum, total * r / world_size, side="left")))
    bounds.append(n)
    bounds = np.maximum.accumulate(np.minimum(bounds, n))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world_size)]


def slab_for_rows(row_end: np.ndarray, lo: int, hi: int) -> tuple[int, int]:
    """Sites a rank must hold to compute rows [lo, hi): [lo, max row end) -- the rows plus their halo."""
    if hi <= lo:
        return lo, lo
    return lo, int(max(int(row_end[lo:hi].max()), hi))


def broadcast_matrix(tensor, src: int = 0):
    """One collective: rank `src`'s GL matrix to every rank (RCCL over xGMI when the backend is nccl)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor

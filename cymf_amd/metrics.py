"""Ranking metrics on a 0/1 vector sorted by score (cymf/metrics.pyx): numpy restatements of
dcg_at_k (:24-43), recall_at_k (:71-85), average_precision_at_k (:109-125) and the IPS variants
(:47-67, :89-103, :129-147).  Host-side harness code, not a hot path."""
import numpy as np


def dcg_at_k(y, k):
    y = np.asarray(y, dtype=np.float64)
    counter = y.sum()
    if counter == 0.0:
        return 0.0
    top = y[:k]
    disc = np.ones(len(top))
    disc[1:] = np.log2(np.arange(1, len(top)) + 1.0)
    return float((top / disc).sum() / counter)


def recall_at_k(y, k):
    y = np.asarray(y, dtype=np.float64)
    counter = y.sum()
    if counter == 0.0:
        return 0.0
    return float(y[:k].sum() / counter)


def average_precision_at_k(y, k):
    y = np.asarray(y, dtype=np.float64)
    counter = y.sum()
    if counter == 0.0:
        return 0.0
    top = y[:k]
    hits = np.cumsum(top)
    ranks = np.arange(1, len(top) + 1, dtype=np.float64)
    return float((hits / ranks)[top == 1].sum() / counter)


def dcg_at_k_with_ips(y, p, k):
    y = np.asarray(y, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    sn = (y / p).sum()
    if sn == 0.0:
        return 0.0
    top, pt = y[:k], p[:k]
    disc = np.ones(len(top))
    disc[1:] = np.log2(np.arange(1, len(top)) + 1.0)
    return float((top / disc / pt).sum() / sn)


def recall_at_k_with_ips(y, p, k):
    y = np.asarray(y, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    sn = (y / p).sum()
    if sn == 0.0:
        return 0.0
    return float((y[:k] / p[:k]).sum() / sn)


def average_precision_at_k_with_ips(y, p, k):
    y = np.asarray(y, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    sn_run = np.cumsum(y / p)
    if sn_run[-1] == 0.0:
        return 0.0
    top = y[:k]
    ranks = np.arange(1, len(top) + 1, dtype=np.float64)
    return float((sn_run[:k] / ranks)[top == 1].sum() / sn_run[-1])

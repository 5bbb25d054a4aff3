"""TEST INFRASTRUCTURE - CPU restatement of the reference's k-reciprocal re-ranking (SURVEY.md 8(f) row N2:
utils/metrics.py:275-278 -> utils/reranking.py:30-101, Zhong et al. CVPR 2017).

numpy / torch-CPU only; imported by tests/ (never by the product path editor_amd/).  Pinned against the reference's own
`re_ranking(qf, gf, k1, k2, lambda)` through tests/golden/f17_rerank.npz (captured by capture_golden.py f17): bit-for-bit on
this container's numpy (the float16 storage of the neighbour weights and the float16 arithmetic of the Jaccard distance are
part of the algorithm as shipped, reranking.py:48,84,90-98, and are reproduced here step by step).

The stages, each citing the lines it follows:
    normalised_distance   reranking.py:37-47   d = |a|^2 + |b|^2 - 2ab over ALL images, then d^T / column-max (fp32)
    reciprocal_weights    reranking.py:51-72   k-reciprocal set R(i,k1), its 2/3-overlap expansion by R(c,k1/2), Gaussian weights
    local_expansion       reranking.py:74-79   mean of the k2 nearest rows (fp32 accumulation, fp16 result)
    jaccard               reranking.py:81-96   sum of min over the common non-zeros, in fp16, ascending column order
    re_ranking            reranking.py:98-101  (1 - lambda) * jaccard (fp16 product) + lambda * d   -> (Q, G) fp32
"""
import numpy as np
import torch


def normalised_distance(qf, gf, local_distmat=None, only_local=False):
    """reranking.py:32-47; local_distmat (N, N): added to the global distances (:44-45) or, only_local, used instead (:32-33)."""
    if only_local:
        dist = np.asarray(local_distmat)
    else:
        feat = torch.cat([qf, gf]).float()
        n = feat.shape[0]
        sq = feat.pow(2).sum(dim=1, keepdim=True)
        dist = (sq.expand(n, n) + sq.expand(n, n).t()).clone()
        dist.addmm_(feat, feat.t(), beta=1, alpha=-2)
        dist = dist.numpy()
        if local_distmat is not None:
            dist = dist + np.asarray(local_distmat)
    return np.transpose(dist / np.max(dist, axis=0))          # od[i, j] = dist[j, i] / max_k dist[k, i]


def _reciprocal(rank, i, k):
    """R(i, k): the members of i's k-nearest list (itself included: k + 1 entries) that have i in THEIR list."""
    near = rank[i, :k + 1]
    back = rank[near, :k + 1]
    return near[np.where(back == i)[0]]


def reciprocal_weights(od, rank, k1):
    n = od.shape[0]
    half = int(np.around(k1 / 2))                              # banker's rounding, as numpy's (reranking.py:61)
    v = np.zeros_like(od).astype(np.float16)
    for i in range(n):
        base = _reciprocal(rank, i, k1)
        members = base
        for c in base:
            cand = _reciprocal(rank, c, half)
            if len(np.intersect1d(cand, base)) > 2 / 3 * len(cand):
                members = np.append(members, cand)
        members = np.unique(members)
        w = np.exp(-od[i, members])
        v[i, members] = w / np.sum(w)
    return v


def local_expansion(v, rank, k2):
    if k2 == 1:
        return v
    out = np.zeros_like(v, dtype=np.float16)
    for i in range(v.shape[0]):
        out[i, :] = np.mean(v[rank[i, :k2], :], axis=0)
    return out


def jaccard(v, nq):
    n = v.shape[0]
    cols = [np.where(v[:, j] != 0)[0] for j in range(n)]       # the rows that hold column j
    out = np.zeros((nq, n), dtype=np.float16)
    for i in range(nq):
        acc = np.zeros((1, n), dtype=np.float16)
        for j in np.where(v[i, :] != 0)[0]:
            rows = cols[j]
            acc[0, rows] = acc[0, rows] + np.minimum(v[i, j], v[rows, j])
        out[i] = 1 - acc / (2 - acc)
    return out


def re_ranking(qf, gf, k1, k2, lambda_value, stages=False, local_distmat=None, only_local=False):
    nq = qf.shape[0]
    od = normalised_distance(qf, gf, local_distmat, only_local)
    rank = np.argsort(od).astype(np.int32)
    v = reciprocal_weights(od, rank, k1)
    vq = local_expansion(v, rank, k2)
    jac = jaccard(vq, nq)
    final = jac * (1 - lambda_value) + od[:nq] * lambda_value
    final = final[:, nq:]
    if stages:
        return final, dict(od=od, rank=rank, v=v, vq=vq, jac=jac)
    return final

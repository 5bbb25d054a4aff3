"""TEST INFRASTRUCTURE - CPU restatement of the reference's retrieval evaluation.

numpy / torch-CPU only; imported by tests/ (and never by the product path editor_amd/).  Pinned against the
reference's own utils/metrics.py through tests/golden/f8_retrieval.npz (captured by capture_golden.py f8).

    euclidean_distance   utils/metrics.py:12-18     qq + gg^T - 2 q g^T (squared distances, fp32, addmm_)
    eval_func            utils/metrics.py:132-191   CMC / mAP, gallery entries with the query's (pid, camid) removed
    eval_func_msrv       utils/metrics.py:34-129    same with (pid, sceneid) removal (MSVR310 protocol, :87)
    r1_map_eval          utils/metrics.py:242-283   feature L2-normalisation + the two steps above
"""
import numpy as np
import torch


def euclidean_distance(qf, gf):
    m, n = qf.shape[0], gf.shape[0]
    dist = qf.pow(2).sum(1, keepdim=True).expand(m, n) + gf.pow(2).sum(1, keepdim=True).expand(n, m).t()
    dist = dist.clone()
    dist.addmm_(qf, gf.t(), beta=1, alpha=-2)
    return dist.numpy()


def eval_func(distmat, q_pids, g_pids, q_aux, g_aux, max_rank=50, sort_kind=None):
    """aux = camids (eval_func) or sceneids (eval_func_msrv): the removal rule has the same form in both.
    sort_kind=None is numpy's default (the reference's call); 'stable' fixes the order of exactly tied distances
    (lowest gallery index first), which numpy's default leaves unspecified."""
    num_q, num_g = distmat.shape
    max_rank = min(max_rank, num_g)
    indices = np.argsort(distmat, axis=1) if sort_kind is None else np.argsort(distmat, axis=1, kind=sort_kind)
    matches = (g_pids[indices] == q_pids[:, None]).astype(np.int32)
    all_cmc, all_ap = [], []
    for qi in range(num_q):
        order = indices[qi]
        keep = ~((g_pids[order] == q_pids[qi]) & (g_aux[order] == q_aux[qi]))
        orig = matches[qi][keep]
        if not orig.any():
            continue
        cmc = orig.cumsum()
        cmc[cmc > 1] = 1
        all_cmc.append(cmc[:max_rank])
        cum = orig.cumsum() / (np.arange(1, orig.shape[0] + 1) * 1.0)
        all_ap.append((cum * orig).sum() / orig.sum())
    assert all_cmc, "Error: all query identities do not appear in gallery"
    cmc = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_cmc))
    return cmc, float(np.mean(all_ap)), indices


def r1_map_eval(feats, pids, camids, num_query, max_rank=50, feat_norm=True):
    if feat_norm:
        feats = torch.nn.functional.normalize(feats, dim=1, p=2)
    qf, gf = feats[:num_query], feats[num_query:]
    dist = euclidean_distance(qf, gf)
    pids, camids = np.asarray(pids), np.asarray(camids)
    cmc, m_ap, _ = eval_func(dist, pids[:num_query], pids[num_query:], camids[:num_query], camids[num_query:], max_rank)
    return cmc, m_ap, dist

"""Retrieval evaluation over the hot path's eval-mode features - row N2 of SURVEY.md 8(f).

Host-side mirror of the reference's utils/metrics.py (same names, argument meaning and return values) over the HIP
kernels of csrc/retrieval.hip; there is no CPU fallback.
    euclidean_distance(qf, gf)                                     utils/metrics.py:12-18
    eval_func(distmat, q_pids, g_pids, q_camids, g_camids, ...)    utils/metrics.py:132-191
    eval_func_msrv(..., q_sceneids, g_sceneids, ...)               utils/metrics.py:34-129  (MSVR310 protocol)
    R1_mAP_eval / R1_mAP                                           utils/metrics.py:242-283 / 193-239
Differences, all deliberate: distance matrices stay on the device (pass `.cpu().numpy()` yourself if a numpy array is
wanted - `compute()` does, as the reference returns one); exactly tied distances rank by gallery index (numpy's default
argsort leaves their order unspecified); eval_func_msrv does not write the reference's `re.txt` rank-list dump.
    re_ranking(qf, gf, k1, k2, lambda_value)                       utils/reranking.py:30-101 (R1_mAP_eval(reranking=True), :275-278)
"""
import numpy as np
import torch

from ._lib import call


def _dev(a, dtype, device):
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    return t.to(device=device, dtype=dtype).contiguous()


def _rows(x):
    x = x.float()
    return x if x.stride(-1) == 1 else x.contiguous()


def normalize(feats, eps=1e-12):
    """torch.nn.functional.normalize(feats, dim=1, p=2)."""
    feats = _rows(feats)
    m, d = feats.shape
    out = torch.empty(m, d, dtype=torch.float32, device=feats.device)
    call("editor_l2norm_rows", feats.data_ptr(), feats.stride(0), m, d, float(eps), out)
    return out


def euclidean_distance(qf, gf):
    """(m, n) fp32 squared-distance matrix, on the device."""
    qf, gf = _rows(qf), _rows(gf)
    m, n, d = qf.shape[0], gf.shape[0], qf.shape[1]
    assert gf.shape[1] == d
    dist = torch.empty(m, n, dtype=torch.float32, device=qf.device)
    qq = torch.empty(m, dtype=torch.float32, device=qf.device)
    gg = torch.empty(n, dtype=torch.float32, device=qf.device)
    call("editor_distmat_f32", qf.data_ptr(), qf.stride(0), gf.data_ptr(), gf.stride(0), m, n, d, qq, gg, dist)
    return dist


def argsort_rows(distmat):
    """np.argsort(distmat, axis=1) as an int32 device tensor (ties by ascending column)."""
    q, g = distmat.shape
    p = 2
    while p < g:
        p *= 2
    keys = torch.empty(q * p, dtype=torch.int64, device=distmat.device)
    order = torch.empty(q, g, dtype=torch.int32, device=distmat.device)
    call("editor_rank_sort", distmat, q, g, p, keys, order)
    return order


def re_ranking(probFea, galFea, k1, k2, lambda_value, local_distmat=None, only_local=False):
    """k-reciprocal re-ranking (utils/reranking.py:30-101) on the device: (Q, G) fp32 final distance, a device tensor.
    Dense N x N stages over all N = Q + G images (csrc/rerank.hip).  The reference's float16 storage / arithmetic is followed
    operation by operation: every stage is bit-equal to numpy's FROM THE SAME INPUTS (tests/test_gpu_retrieval.py); end to end the
    result is tolerance-close, not bit-equal - expf differs from numpy's exp in the last place and near-tied distances can order
    differently.  Limits (checked here, documented divergences from the reference, which slices short neighbour lists silently and
    takes any k1): k1 <= 63, N >= k1 + 1, N <= 65 535 - the evaluator's call (utils/metrics.py:278: k1=50, k2=15, lambda=0.3) is inside
    all of them.  local_distmat (N, N; numpy or tensor): added to the global distance matrix before the normalisation
    (reranking.py:44-45) or, with only_local, used instead of it (:32-33) - one elementwise add on the device, the evaluator passes
    neither."""
    if only_local and local_distmat is None:
        raise ValueError("re_ranking(only_local=True) needs local_distmat")
    qf, gf = _rows(probFea), _rows(galFea)
    if not qf.is_cuda:
        raise RuntimeError("re_ranking: features must be on the GPU (there is no CPU fallback)")
    nq, n = qf.shape[0], qf.shape[0] + gf.shape[0]
    if n > 65535:
        raise ValueError("re_ranking: at most 65 535 images (query + gallery) - the device stages index rows through grid.y")
    if int(k1) > 63 or int(k1) + 1 > n:
        raise ValueError("re_ranking: k1 <= 63 and k1 + 1 <= Q + G (the device kernel holds a neighbour list of k1 + 1 in LDS)")
    dev = qf.device
    local = None
    if local_distmat is not None:
        local = torch.as_tensor(np.asarray(local_distmat) if not isinstance(local_distmat, torch.Tensor) else local_distmat)
        local = local.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(local.shape) != (n, n):
            raise ValueError("re_ranking: local_distmat must be (Q + G, Q + G)")
    if only_local:
        dist = local
    else:
        feat = torch.cat([qf, gf]).contiguous()
        dist = euclidean_distance(feat, feat)
        if local is not None:
            dist = dist + local
    od = torch.empty(n, n, dtype=torch.float32, device=dev)
    call("editor_rerank_normalise", dist, n, torch.empty(n, dtype=torch.float32, device=dev), od)
    del dist
    rank = argsort_rows(od)
    v = torch.empty(n, n, dtype=torch.float16, device=dev)
    call("editor_rerank_weights", od, rank, n, int(k1), int(np.around(k1 / 2)), v)
    if k2 != 1:
        vq = torch.empty_like(v)
        call("editor_rerank_expand", v, rank, n, int(k2), vq)
        v = vq
    del rank
    final = torch.empty(nq, n - nq, dtype=torch.float32, device=dev)
    w16 = int(np.float16(1 - lambda_value).view(np.uint16))
    call("editor_rerank_final", v, torch.empty_like(v), od, n, nq, w16, float(np.float32(lambda_value)), final)
    return final


def _evaluate(distmat, q_pids, g_pids, q_aux, g_aux, max_rank):
    if not isinstance(distmat, torch.Tensor):
        distmat = torch.as_tensor(np.asarray(distmat))
    if not distmat.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("eval_func: no GPU (there is no CPU fallback for the retrieval kernels)")
        distmat = distmat.cuda()
    distmat = distmat.float().contiguous()
    dev = distmat.device
    num_q, num_g = distmat.shape
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    order = argsort_rows(distmat)
    ap = torch.empty(num_q, dtype=torch.float64, device=dev)
    first = torch.empty(num_q, dtype=torch.int32, device=dev)
    totals = torch.empty(2, dtype=torch.float64, device=dev)
    counts = torch.empty(max_rank, dtype=torch.int32, device=dev)
    call("editor_rank_metrics", order, _dev(q_pids, torch.int64, dev), _dev(g_pids, torch.int64, dev),
         _dev(q_aux, torch.int64, dev), _dev(g_aux, torch.int64, dev), num_q, num_g, max_rank, ap, first, totals, counts)
    tot = totals.cpu().numpy()
    num_valid_q = float(tot[1])
    assert num_valid_q > 0, "Error: all query identities do not appear in gallery"
    all_cmc = counts.cpu().numpy().astype(np.float32) / num_valid_q
    m_ap = float(tot[0] / num_valid_q)
    return all_cmc, m_ap, order, ap, first


def eval_func(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    """Market-1501 protocol: gallery entries with the query's pid AND camid are discarded."""
    cmc, m_ap, _, _, _ = _evaluate(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
    return cmc, m_ap


def eval_func_msrv(distmat, q_pids, g_pids, q_camids, g_camids, q_sceneids, g_sceneids, max_rank=50):
    """MSVR310 protocol (utils/metrics.py:87): entries with the query's pid AND scene id are discarded."""
    cmc, m_ap, _, _, _ = _evaluate(distmat, q_pids, g_pids, q_sceneids, g_sceneids, max_rank)
    return cmc, m_ap


class R1_mAP_eval():
    def __init__(self, num_query, max_rank=20, feat_norm=True, reranking=False):
        self.num_query = num_query
        self.max_rank = max_rank
        self.feat_norm = feat_norm
        self.reranking = reranking

    def reset(self):
        self.feats = []
        self.pids = []
        self.camids = []

    def update(self, output):
        feat, pid, camid = output
        self.feats.append(feat.detach().float())          # stays on the device
        self.pids.extend(np.asarray(pid.cpu() if isinstance(pid, torch.Tensor) else pid))
        self.camids.extend(np.asarray(camid.cpu() if isinstance(camid, torch.Tensor) else camid))

    def _split(self):
        feats = torch.cat(self.feats, dim=0)
        if self.feat_norm in (True, 'yes'):
            print("The test feature is normalized")
            feats = normalize(feats)
        nq = self.num_query
        return feats[:nq], feats[nq:], np.asarray(self.pids[:nq]), np.asarray(self.pids[nq:]), \
            np.asarray(self.camids[:nq]), np.asarray(self.camids[nq:])

    def compute(self, vis=0):
        qf, gf, q_pids, g_pids, q_camids, g_camids = self._split()
        if self.reranking:
            print('=> Enter reranking')
            distmat = re_ranking(qf, gf, k1=50, k2=15, lambda_value=0.3)          # utils/metrics.py:278
        else:
            print('=> Computing DistMat with euclidean_distance')
            distmat = euclidean_distance(qf, gf)
        # the reference's max_rank attribute is not forwarded to eval_func (utils/metrics.py:282): default 50
        cmc, m_ap = eval_func(distmat, q_pids, g_pids, q_camids, g_camids)
        return cmc, m_ap, distmat.cpu().numpy(), self.pids, self.camids, qf, gf


class R1_mAP(R1_mAP_eval):
    def __init__(self, num_query, max_rank=50, feat_norm='yes'):
        super().__init__(num_query, max_rank, feat_norm)

    def reset(self):
        super().reset()
        self.sceneids = []
        self.img_path = []

    def update(self, output):
        feat, pid, camid, sceneid, img_path = output
        super().update((feat, pid, camid))
        self.sceneids.extend(np.asarray(sceneid.cpu() if isinstance(sceneid, torch.Tensor) else sceneid))
        self.img_path.extend(img_path)

    def compute(self, cfg=None):
        qf, gf, q_pids, g_pids, q_camids, g_camids = self._split()
        nq = self.num_query
        q_s, g_s = np.asarray(self.sceneids[:nq]), np.asarray(self.sceneids[nq:])
        distmat = euclidean_distance(qf, gf)
        cmc, m_ap = eval_func_msrv(distmat, q_pids, g_pids, q_camids, g_camids, q_s, g_s)
        return cmc, m_ap, distmat.cpu().numpy(), self.pids, self.camids, qf, gf

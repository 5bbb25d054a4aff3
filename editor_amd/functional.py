"""Autograd nodes of the EDITOR hot path.  Each Function's forward AND backward is a fixed sequence of
HIP kernel launches through the C ABI (editor_amd.ops); torch only owns the tensors, the streams and
the graph that hands parameter gradients to nn.Parameter.grad (so DDP hooks / optimizers are unchanged).

Granularity is one node per transformer block (12 backbone blocks + 3 modality blocks + 1 joint block),
so gradient buckets become ready block by block and the RCCL all-reduce overlaps the rest of backward.

Activation dtype: torch.bfloat16 (performance mode, bf16 MFMA with fp32 accumulation), torch.float16 (the
reference's own autocast dtype, engine/processor.py:79: same MFMA rate, 3 more mantissa bits; its backward
runs on LOSS-SCALED half gradients as amp.GradScaler does, :94) or torch.float32 (parity mode, exact-f32
MFMA).  The residual stream, LayerNorm statistics, losses and every parameter gradient are fp32 in all modes.
"""
import os
import weakref

import torch

from . import ops

_BF16_CACHE = {}
_WEIGHT_EPOCH = 0

# f16 mode: gradients that travel as half tensors (dy into the dgrad / wgrad products, attention backward, the LayerNorm
# backward's input) are multiplied by this power of two where they are cast from the fp32 residual gradient and divided
# out again where they return to fp32 (wgrad alpha, column sums, LayerNorm backward) - the static form of the
# reference's amp.GradScaler (engine/processor.py:60,94-96; its initial scale is 65536).  Exact whenever nothing
# under- / overflows: half keeps full precision for |g| * scale in [6.1e-5, 65504], i.e. activation-gradient elements
# from 1.9e-9 to 2.0 at the default 2^15 (measured at B = 128: with 2^12 the first layer's gradients, ~1e-8 per
# element, fell into the subnormal range and the patch-embedding weight gradient was 2.3 % off; the mean-reduced
# losses keep elements far below 1).  cfg.MODEL.GRAD_SCALE overrides it.
F16_GRAD_SCALE = 32768.0


def set_f16_grad_scale(v):
    global F16_GRAD_SCALE
    v = float(v)
    if v <= 0 or (v != 2.0 ** round(__import__("math").log2(v))):
        raise ValueError("GRAD_SCALE must be a positive power of two")
    F16_GRAD_SCALE = v


def grad_scale(dtype):
    return F16_GRAD_SCALE if dtype == torch.float16 else 1.0


FWD_GRAD = True        # torch.is_grad_enabled() as EDITOR.forward saw it (it is always False inside Function.forward, and
                       # ctx.needs_input_grad reflects requires_grad whatever the grad mode: ADVICE r4)


def set_model_options(grad_scale_f16=None, act_light=None, grad_enabled=None):
    """Per-MODEL options, installed by EDITOR.forward for the nodes it is about to create (every node captures them in its
    ctx at forward time, so two models with different settings can live in one process).  None = keep."""
    global ACT_LIGHT, FWD_GRAD
    if grad_enabled is not None:
        FWD_GRAD = bool(grad_enabled)
    if grad_scale_f16 is not None:
        set_f16_grad_scale(grad_scale_f16)
    if act_light is not None:
        ACT_LIGHT = bool(act_light)


def invalidate_weight_cache():
    """Call after parameters were modified by something that does not bump tensor version counters
    (editor_amd.optim.FusedSGD's raw HIP update)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def install_weight_copies(pairs, transposed=False):
    """(parameter, 16-bit tensor holding its current value [transposed]) pairs -> entries of the operand cache for the
    current epoch."""
    for w, h in pairs:
        key = (id(w), h.dtype, "T") if transposed else (id(w), h.dtype)
        _BF16_CACHE[key] = (weakref.ref(w), w._version, h, w.data_ptr(), _WEIGHT_EPOCH)


# dgrad products with BOTH operands k-major: needs W^T (K_in, N_out) next to W (N_out, K_in) - 2 400 instead of 2 840
# cycles per K-tile in the 256x256 kernel (the row-k operand goes through ds_read_b64_tr_b16, twice the LDS instructions)
DGRAD_KMAJOR = os.environ.get("EDITOR_DGRAD_KMAJOR", "1") != "0"


def act_weight_t(w, dtype):
    """16-bit W^T (contiguous) of a 2-D fp32 master weight, cached like act_weight (FusedSGD refreshes it in its own launch)."""
    key = (id(w), dtype, "T")
    ent = _BF16_CACHE.get(key)
    ver = w._version
    if ent is None or ent[0]() is not w or ent[1] != ver or ent[3] != w.data_ptr() or ent[4] != _WEIGHT_EPOCH:
        ent = (weakref.ref(w), ver, act_weight(w, dtype).t().contiguous(), w.data_ptr(), _WEIGHT_EPOCH)
        _BF16_CACHE[key] = ent
    return ent[2]


# Activation-light blocks (cfg.MODEL.ACT_LIGHT; BASELINE config 5 at B = 64 per GPU does not fit otherwise): a block saves
# 24 instead of 36 bytes per token-row-element - not the LayerNorm outputs h1 / h2 (recomputed in the backward from the saved
# residual rows and statistics: one LayerNorm pass each) and not the GELU output g (recomputed from the saved pre-activation,
# which then also feeds gelu' in the fc2 dgrad epilogue instead of a saved gelu').  Same forward bits; 16-bit modes only.
ACT_LIGHT = False

F16X2 = "f16x2"      # act_dtype marker of the split-precision forward: half pairs in the forward, plain f16 in the backward
F16X2H = "f16x2h"    # ... of a block whose attention MAP must be fp32-class but whose output need not be: LayerNorm-1, the qkv
                     # product and the attention core on half pairs, the projection and the MLP as the plain f16 mode computes
                     # them (cfg.MODEL.SPLIT_SCOPE = 'selection': the LAST backbone block - nothing downstream of its attention
                     # map feeds the token selection, SFTS.py:145-164)


def act_weight_split(w):
    """fp32 master weight -> the (hi, lo) half pair of w * ops.SPLIT_WSCALE (forward operand of the 'f16x2' mode), cached like
    act_weight (FusedSGD refreshes the pairs in its own launch)."""
    key = (id(w), F16X2)
    ent = _BF16_CACHE.get(key)
    ver = w._version
    if ent is None or ent[0]() is not w or ent[1] != ver or ent[3] != w.data_ptr() or ent[4] != _WEIGHT_EPOCH:
        ent = (weakref.ref(w), ver, ops.split_f32(w.detach(), ops.SPLIT_WSCALE), w.data_ptr(), _WEIGHT_EPOCH)
        _BF16_CACHE[key] = ent
    return ent[2]


def install_weight_pairs(triples):
    """(parameter, hi, lo) -> entries of the operand cache for the current epoch (editor_amd.optim.FusedSGD)."""
    for w, hi, lo in triples:
        _BF16_CACHE[(id(w), F16X2)] = (weakref.ref(w), w._version, (hi, lo), w.data_ptr(), _WEIGHT_EPOCH)


def act_weight(w, dtype):
    """fp32 master parameter -> GEMM operand dtype.  16-bit copies are cached per parameter OBJECT and re-cast when
    the parameter's version counter moves (optimizer step, load_state_dict).  The entry holds a weak reference so
    that a recycled id()/address of a freed parameter can never serve stale weights."""
    if dtype == torch.float32:
        return w.detach()
    key = (id(w), dtype)
    ent = _BF16_CACHE.get(key)
    ver = w._version
    if ent is None or ent[0]() is not w or ent[1] != ver or ent[3] != w.data_ptr() or ent[4] != _WEIGHT_EPOCH:
        if len(_BF16_CACHE) > 4096:                       # drop entries of dead parameters
            for k in [k for k, e in _BF16_CACHE.items() if e[0]() is None]:
                del _BF16_CACHE[k]
        ent = (weakref.ref(w), ver, ops.cast(w.detach().contiguous().view(-1), dtype).view(w.shape),
               w.data_ptr(), _WEIGHT_EPOCH)
        _BF16_CACHE[key] = ent
    return ent[2]


def _linear_fwd(x2d, w_act, bias, out_dtype, m_live=None):
    """y = x W^T + b;  x (M,K) act dtype, W (N,K)."""
    m, k = x2d.shape
    n = w_act.shape[0]
    y = torch.empty(m, n, dtype=out_dtype, device=x2d.device)
    ops.gemm(x2d, w_act, y, m, n, k, k, k, n, 0, 0, bias=bias, m_live=m_live)
    return y


# Weight-gradient products run on a SIDE stream, concurrently with whatever the main stream does next (the dgrad of the
# same layer, LayerNorm / attention backward): a dgrad's last round of tiles and a wgrad's single round of 243-252
# workgroups each leave CUs idle that the other's workgroups take.  Measured: 0.6 ms per step - a 256x256
# tile's workgroup owns its CU, so the two streams take turns on CUs rather than share them.  Discipline: outputs are allocated on the main stream; the side stream waits for an
# event recorded after the producer of dy; the backward function joins (main waits side) before it returns.
_SIDE = {}
_SIDE_KEEP = []
WGRAD_SIDE_STREAM = os.environ.get("EDITOR_WGRAD_STREAM", "1") != "0"


def _side_stream(device):
    st = _SIDE.get(device.index)
    if st is None:
        st = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return st


FUSE_LN_CAST = os.environ.get("EDITOR_FUSE_LN_CAST", "1") != "0"      # measurement switch (TransformerBlockFn.backward)


def join_side_stream(device):
    """Main stream waits for every weight-gradient launch issued so far (call before handing gradients to autograd)."""
    st = _SIDE.get(device.index)
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)
    _SIDE_KEEP.clear()          # (main-stream work enqueued from here on is ordered after the side stream's reads)


# The four weight gradients of a transformer block as ONE grouped launch (ops.gemm_wgrad_group) at the end of the block's
# backward instead of four split-K launches spread over it, on the side stream.
#   * With gradient SINKS (editor_amd.ddp.GradBuckets - bench.py and the training loop always enable them, with or without a
#     process group) the gradients are written in place into bucket slots and autograd is handed nothing, so the launch may
#     still be running when the block's backward returns: it is joined one block LATER (the next block's dgrad chain overlaps
#     it) - by the next block, by a bucket's all-reduce, by GradBuckets.finish() and by FusedSGD.step().
#   * Without sinks the gradients flow through autograd, whose AccumulateGrad may read them (it clones a gradient it cannot
#     steal) on the main stream as soon as the backward returns - a deferred join there handed it unwritten memory (measured:
#     NaN losses) - so the launch is joined before the block's backward returns.
GROUP_WGRAD = os.environ.get("EDITOR_GROUP_WGRAD", "1") != "0"
WGRAD_DEFER_JOIN = os.environ.get("EDITOR_WGRAD_DEFER", "1") != "0"       # measurement switch: 0 = join at the end of each block


DEFER_REDUCE = os.environ.get("EDITOR_DEFER_REDUCE", "1") != "0"     # measurement switch: ops.ReduceQueue in the block backward


# ---- launch requests (round 4) ------------------------------------------------------------------------------------------------
# The forward and backward of a transformer block are written as GENERATORS that yield their large GEMM launches (and the grouped
# weight-gradient launch) as requests instead of issuing them.  One block: `_drive` issues every request as it comes - the same
# launches in the same order as before.  Several blocks of identical shape (the HMA head's per-modality blocks, GroupedBlocksFn):
# `_drive_group` advances them in lockstep and sends requests that agree in everything but their operands out as ONE grouped launch
# (ops.gemm_group: 87 live tiles per product and modality are a third of a round of the 256 CUs).  Same kernels, same tiles: the
# results are bit-identical either way.
class _GemmReq:
    __slots__ = ("args", "kw")

    def __init__(self, *args, **kw):
        self.args, self.kw = args, kw


class _WgradReq:
    __slots__ = ("jobs", "m", "alpha", "m_live", "deferred", "dev")

    def __init__(self, jobs, m, alpha, m_live, deferred, dev):
        self.jobs, self.m, self.alpha, self.m_live, self.deferred, self.dev = jobs, m, alpha, m_live, deferred, dev


def _run_wgrads(reqs):
    """the grouped weight-gradient launches of one or several blocks: side stream (joined one block later when the gradients land in
    sinks, see GROUP_WGRAD below) or inline"""
    dev = reqs[0].dev
    if WGRAD_SIDE_STREAM:
        if any(r.deferred for r in reqs):
            join_side_stream(dev)                   # the PREVIOUS block's grouped launch (had a whole block of slack)
        ready = torch.cuda.current_stream(dev).record_event()
        side = _side_stream(dev)
        side.wait_event(ready)
        for r in reqs:
            _SIDE_KEEP.append([(j[0], j[1]) for j in r.jobs])      # operands stay referenced until the join
        with torch.cuda.stream(side):
            for r in reqs:
                ops.gemm_wgrad_group(r.jobs, r.m, r.alpha, r.m_live)
    else:
        for r in reqs:
            ops.gemm_wgrad_group(r.jobs, r.m, r.alpha, r.m_live)


def _run_req(req):
    if isinstance(req, _GemmReq):
        ops.gemm(*req.args, **req.kw)
    else:
        _run_wgrads([req])


def _drive(gen):
    """run a block generator alone: every request is issued as it is yielded"""
    try:
        while True:
            _run_req(next(gen))
    except StopIteration as e:
        return e.value


_GROUP_SLOT = [0, 1]        # (slot, slots) of the block whose generator is running: its share of per-stream scratch (ops.ReduceQueue)


def _drive_group(gens):
    """run block generators of identical structure in lockstep, grouping the requests they yield together"""
    n = len(gens)
    results = [None] * n
    while True:
        reqs, stopped = [], 0
        for i, g_ in enumerate(gens):
            _GROUP_SLOT[0], _GROUP_SLOT[1] = i, n
            try:
                reqs.append(next(g_))
            except StopIteration as e:
                results[i] = e.value
                stopped += 1
            finally:
                _GROUP_SLOT[0], _GROUP_SLOT[1] = 0, 1
        if stopped == n:
            return results
        if stopped:
            raise RuntimeError("grouped blocks left lockstep (their shapes / options must be identical)")
        if all(isinstance(r, _GemmReq) for r in reqs) and ops.gemm_group_ok([(r.args, r.kw) for r in reqs]):
            ops.gemm_group([(r.args, r.kw) for r in reqs])
        elif all(isinstance(r, _WgradReq) for r in reqs):
            _run_wgrads(reqs)
        else:
            for r in reqs:
                _run_req(r)


class _SubCtx:
    """what a block generator needs of an autograd ctx, for the blocks inside GroupedBlocksFn"""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


# EDITOR_STAGGER_QKV=c (A/B switch, round 5): the qkv forward's first round of workgroups starts spread over c * 2048 cycles
# (ops.EPI_STAGGER) - the one product the spread helped in tools/stagger_sweep.py (rotating operands: 214 -> 186 us; every other
# product of the path is flat or slower with it).  In the step, same box, twice each: 42.56 / 42.57 -> 42.31 / 42.34 ms replay-only.
STAGGER_QKV = ops.EPI_STAGGER(int(os.environ.get("EDITOR_STAGGER_QKV", "24")))
# round 6 A/B switches (tools/stagger_sweep.py, rotating operands: fc2 dgrad x gelu' 295 -> 283 us at 32, proj forward + residual 132 -> 125 at 8)
STAGGER_FC2D = ops.EPI_STAGGER(int(os.environ.get("EDITOR_STAGGER_FC2D", "0")))
STAGGER_PROJ = ops.EPI_STAGGER(int(os.environ.get("EDITOR_STAGGER_PROJ", "0")))

# LayerNorm-1's backward of block i+1 also writes what block i's backward STARTS with: the 16-bit, drop-path- and loss-scaled copy of
# dL/dx (operand of the fc2 dgrad / wgrad) and its column sums (the fc2 bias gradient) - ops.layernorm_bwd_cast, as LayerNorm-2's
# backward already does for the projection - instead of a cast_rows_colsum pass over the fp32 gradient one launch later
# (-152 MB read per backbone layer).  The two autograd nodes meet through a `_CastBox`: block i leaves one in `_CAST_SLOT` when its
# forward returns (holding a weak reference to its output), block i+1's forward picks it up if its input IS that tensor object; in the backward the
# producer fills it and the consumer checks that the gradient it is handed is the very tensor the producer returned, unmodified
# (storage, size, version counter) - anything else (a second consumer of the block output, hooks) falls back to the plain pass.
HANDOFF_CAST = os.environ.get("EDITOR_HANDOFF_CAST", "1") != "0"
# (opt-in) LayerNorm-1's backward as a memory-bound ROLE of the block's weight-gradient launch - see backward_gen, DESIGN 9
WGRAD_LN = os.environ.get("EDITOR_WGRAD_LN", "0") == "1"
_CAST_SLOT = [None]


class _CastBox:
    __slots__ = ("out", "m", "dtype", "rowscale", "has_bias", "gs", "cs_out", "handoff", "perm")

    def __init__(self, out, m, dtype, rowscale, has_bias, gs, cs_out, perm=None):
        self.out, self.m, self.dtype = weakref.ref(out), m, dtype      # (the tensor OBJECT the node returns: an address can be re-used)
        self.rowscale, self.has_bias, self.gs, self.cs_out = rowscale, has_bias, gs, cs_out
        self.perm = perm            # stochastic-depth plan of the consumer's MLP branch: the 16-bit copy goes to its compacted rows
        self.handoff = None

    def put(self, dx, dy16, dbias):
        self.handoff = (dx, dy16, dbias, dx._version)

    def take(self, dx2):
        """(dy16, dbias) if `dx2` is the gradient the producer returned, untouched; else None."""
        h, self.handoff = self.handoff, None
        if h is None or dx2.data_ptr() != h[0].data_ptr() or dx2.numel() != h[0].numel() or dx2.dtype != h[0].dtype:
            return None
        if h[0]._version != h[3] or dx2._version != h[3]:
            return None
        return h[1], h[2]


def _cast_boxes(ctx, x, out, m, d, act_dtype, mask, cu, m_live, rowscale_mlp, has_fc2_bias, sink, plain, perm=None):
    """forward side of HANDOFF_CAST: ctx.feeds_box = the box of the block whose output `x` is; ctx.box = this block's own."""
    prev, _CAST_SLOT[0] = _CAST_SLOT[0], None
    ctx.feeds_box = ctx.box = None
    if not (HANDOFF_CAST and FUSE_LN_CAST and plain and act_dtype in ops.HALF_DTYPES and cu is None and mask is None
            and m_live is None and d % 256 == 0 and d <= 1024):
        return
    if prev is not None and prev.out() is x and prev.m == m and prev.dtype == act_dtype:
        ctx.feeds_box = prev
    ctx.box = _CAST_SLOT[0] = _CastBox(out, m, act_dtype, rowscale_mlp, has_fc2_bias, grad_scale(act_dtype),
                                       sink.views[11] if sink is not None else None, perm)



def _linear_bwd(*args, **kw):
    """_linear_bwd_gen run alone (the small fp32 linears, PatchEmbedFn): every launch issued in place"""
    return _drive(_linear_bwd_gen(*args, **kw))


def _linear_bwd_gen(dy, x2d, w_act, need_bias, gelu_pre=None, m_live=None, db=None, dx_colsum=None, gs=1.0, dw_out=None,
                    db_out=None, dxcs_out=None, w_t=None, defer=None, aux_is_grad=True, rq=None, live_dense=False):
    """dx = dy W (optionally * gelu'(gelu_pre), fused epilogue) ; dW = dy^T x (fp32) ; db = colsum(dy) (or the
    caller's, when the kernel that produced dy summed its columns on the way).  dx_colsum: also return colsum(dx) - the
    bias gradient of the layer BELOW - from the dgrad's own epilogue when it can deliver it (else None).
    gs: loss scale carried by dy (f16 mode): dx keeps it, dW / db / the returned column sums have it divided out.
    dw_out / db_out / dxcs_out: write the weight gradient / bias gradient / column sums of dx THERE (views of a gradient
    bucket, editor_amd.ddp.GradBuckets) instead of into fresh tensors."""
    m, n = dy.shape
    k = x2d.shape[1]
    inv = 1.0 / gs
    use_side = WGRAD_SIDE_STREAM and dy.dtype in ops.HALF_DTYPES and m >= 2048
    dy_ready = torch.cuda.current_stream(dy.device).record_event() if use_side else None   # BEFORE the dgrad launch
    dx = torch.empty(m, k, dtype=x2d.dtype, device=dy.device)
    dxcs = None
    # live_dense: *m_live is the live prefix of stochastic-depth-compacted rows (ops.droppath_plan; rows behind it are zero in dy)
    if dx_colsum and ops.gemm_colsum_ok(m, k, n, dx.dtype, 0, 1, m_live, live_dense):
        dxcs = dxcs_out if dxcs_out is not None else torch.empty(k, dtype=torch.float32, device=dy.device)
    # B = W stored (Kred=n, Nout=k): row-k operand (trans_b = 1); or its k-major copy W^T (Nout=k, Kred=n): trans_b = 0
    wb, ldb, tb = (w_t, n, 0) if w_t is not None else (w_act, k, 1)
    if gelu_pre is None:
        yield _GemmReq(dy, wb, dx, m, k, n, n, ldb, k, 0, tb, m_live=m_live, colsum=dxcs, colsum_scale=inv, tag="dgrad",
                       rq=rq, live_dense=live_dense)
    else:
        ag = ops.EPI_AUX_GRAD if (dy.dtype in ops.HALF_DTYPES and aux_is_grad) else 0   # 16-bit: gelu_pre holds gelu'(pre-activation)
        yield _GemmReq(dy, wb, dx, m, k, n, n, ldb, k, 0, tb, epilogue=ops.EPI_GELU_BWD | ag | (STAGGER_FC2D if (m_live is None or live_dense) and m >= 16384 else 0),
                       aux=gelu_pre, m_live=m_live,
                       colsum=dxcs, colsum_scale=inv, tag="dgrad", rq=rq, live_dense=live_dense)
    dw = dw_out if dw_out is not None else torch.empty(n, k, dtype=torch.float32, device=dy.device)
    if need_bias and db is None:
        db = db_out if db_out is not None else torch.empty(n, dtype=torch.float32, device=dy.device)
        need_colsum = True
    else:
        need_colsum = False
    if need_colsum and m_live is not None:
        #  ops.colsum sums ALL m rows; behind *m_live the rows of dy are unwritten (cast_rows / the dgrads skip dead tiles).
        # The only live-row products are the compacted HMA head's, whose linears have no bias (vit_pytorch.py:232-237,150-156).
        raise RuntimeError("bias gradient of a live-row (compacted) product: editor_colsum has no m_live form")
    sk, skf = _splitk_for(n, k, m)
    if defer is not None:
        # the weight gradient joins the block's grouped launch (issued by the caller once the last dy exists)
        defer.append((dy, x2d, dw, m_live) if live_dense else (dy, x2d, dw))
        if need_colsum:
            ops.colsum(dy, out=db, scale=inv, rq=rq)
    elif use_side:
        side = _side_stream(dy.device)
        side.wait_event(dy_ready)                    # dy is complete; the dgrad above runs concurrently
        # dy may be released by the caller (and its block re-used by a main-stream allocation) before the side stream has
        # read it: keep both operands referenced until the join (cheaper than record_stream, which made the caching
        # allocator hold blocks back and cost 3 ms per eager step)
        _SIDE_KEEP.append((dy, x2d))
        with torch.cuda.stream(side):
            ops.gemm(dy, x2d, dw, n, k, m, n, k, k, 1, 1, alpha=inv, splitk=sk, epilogue=skf, m_live=m_live)
            if need_colsum:
                ops.colsum(dy, out=db, scale=inv)
    else:
        ops.gemm(dy, x2d, dw, n, k, m, n, k, k, 1, 1, alpha=inv, splitk=sk, epilogue=skf, m_live=m_live)   # both stored (Kred=m, .)
        if need_colsum:
            ops.colsum(dy, out=db, scale=inv)
    if dx_colsum is not None:
        return dx, dw, (db if need_bias else None), dxcs
    return dx, dw, (db if need_bias else None)


def _splitk_for(n_out, k_out, m_red, cus=256):
    """Split count AND kernel of a weight gradient (few output tiles, reduction over all token rows): (splitk, flags).
    Outputs that tile into 256 x 256 (every linear layer of the path) take the ping-pong kernel with ONE round of
    workgroups - splitk = CUs // tiles - measured with tools/gemm_bench.py on M = 49 536 token rows against the 256 x 128
    three-stage kernel at ITS best split: (2304,768) 1 026 vs 729 TFLOP/s, (3072,768) 1 117 vs 751, (768,3072) 1 145 vs
    862, (768,768) 659 vs 604 (both waves of a SIMD sit at the same barrier in the three-stage kernel; the ping-pong
    kernel's two wave groups alternate on the matrix core).  Half or double that split loses 25-45 %.
    Other shapes: the 256 x 128 kernel with the cost model fitted in round 1 (units = one 64-deep K-tile of a 256x128
    tile, ~1.3 us): rounds of `cus` workgroups x (K-tiles per split + 15) + 3.3 per split for its slab."""
    nk = max(1, m_red // 64)
    if n_out % 256 == 0 and k_out % 256 == 0:
        tiles = (n_out // 256) * (k_out // 256)
        return max(1, min(cus // tiles, nk // 8)), ops.EPI_FORCE_PP
    tiles = ((n_out + 255) // 256) * ((k_out + 127) // 128)
    best, best_cost = 1, float("inf")
    for sk in range(1, max(1, min(32, nk // 16)) + 1):
        rounds = (tiles * sk + cus - 1) // cus
        cost = rounds * (nk / sk + 15.0) + 3.3 * sk
        if cost < best_cost - 1e-9:
            best, best_cost = sk, cost
    return best, 0


DROP_SKIP = os.environ.get("EDITOR_DROP_SKIP", "1") != "0"      # A/B switch: stochastic-depth compaction of the MLP branch


def _plan_ok(act_dtype, m, d, hidden, cu, mask, m_live, branch16, defer_out, pend_branch, rowscale_mlp):
    """Can this block run its MLP branch on the stochastic-depth-compacted rows?  The plain dense 16-bit block whose backward takes
    the grouped weight gradients, the deferred reductions and the fused LayerNorm casts (every condition backward_gen re-derives)."""
    return (DROP_SKIP and rowscale_mlp is not None and act_dtype in ops.HALF_DTYPES and cu is None and mask is None and m_live is None
            and not branch16 and not defer_out and pend_branch is None and not ACT_LIGHT and GROUP_WGRAD and DEFER_REDUCE
            and FUSE_LN_CAST and m >= 2048 and m % 64 == 0 and d % 256 == 0 and d <= 1024 and hidden % 256 == 0)


class TransformerBlockFn(torch.autograd.Function):
    """Block.forward(get_att=True) (vit_pytorch.py:215-220) and the masked blocks of BlockMask.forward
    (vit_pytorch.py:311-317,327-328 with AttentionMask :240-258 / MlpMasked :158-168).

    x (B,T,D) fp32.  mask (B,T) uint8 or None.  probs_out: (B,heads,T,T) fp32 buffer that receives the
    softmax output (non-differentiable, feeds the rollout) or None.  rowscale: (B*T) fp32 per-row
    drop-path scale keep/keep_prob (vit_pytorch.py:52-69) or None; two independent draws (attn, mlp).
    """

    @staticmethod
    def forward(ctx, *args):
        return _drive(TransformerBlockFn.forward_gen(ctx, *args))

    @staticmethod
    def backward(ctx, dx2, *_unused):
        return _drive(TransformerBlockFn.backward_gen(ctx, dx2))

    @staticmethod
    def forward_gen(ctx, x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, mask, probs_out,
                    heads, eps, act_dtype, rowscale_attn, rowscale_mlp, cu=None, max_t=None, m_live=None, qk_scale=None,
                    sink=None, pend_branch=None, pend_rs=None, defer_out=False, branch16=False, drop_plan=None):
        # (a generator: the four large products of the plain 16-bit path are YIELDED as launch requests, see _drive / _drive_group)
        # branch16 (cfg.MODEL.BRANCH16; bf16 backbone blocks): the projection and fc2 products write their 16-bit branch output with
        # the plain epilogue and the residual add happens inside the LayerNorm that follows (ops.resid_add_layernorm_fwd) - the
        # next block's LayerNorm-1 for the fc2 branch: `defer_out` makes this node return (x1, fc2 branch) instead of x2, and
        # (pend_branch, pend_rs) are the previous block's pair for which `x` is x1.  The backward is unchanged: the gradient
        # that arrives for the x1 output IS dL/dx2 (x2 = x1 + rs * branch), and the branch output's own gradient,
        # rs * dL/dx2 rounded to 16 bits, is what this node's backward computes first anyway.
        # sink (editor_amd.ddp.BlockSink or None): the 12 parameter gradients of this block are written straight into
        # views of a flat all-reduce bucket and the bucket's collective is launched from the end of the backward
        # dense: x (B,T,D), mask (B,T) token mask.  packed (compacted HMA): x (M,D), cu (B+1) sequence row ranges,
        # max_t = longest sequence, mask (M) = 1 for live rows / 0 for the padding rows at the end.
        if cu is None:
            b, t, d = x.shape
        else:
            b, t, d = cu.numel() - 1, int(max_t), x.shape[1]
        m = x.numel() // d
        hd = d // heads
        x2d = x.reshape(m, d)
        amask = None if cu is not None else mask           # packed sequences hold only live tokens
        if (branch16 or defer_out or pend_branch is not None) and (act_dtype not in ops.HALF_DTYPES or cu is not None
                                                                   or mask is not None or m_live is not None or d % 256):
            raise RuntimeError("branch16 blocks: dense 16-bit rows only")
        # drop_plan = (perm, inv, live) of THIS block's MLP branch (ops.droppath_plan; round 6): the samples whose stochastic-depth
        # draw dropped the branch (rowscale_mlp == 0: x2 = x1 + 0, no gradient through it - vit_pytorch.py:66-68,218) are not
        # computed: LayerNorm-2 writes the live samples' rows compacted, fc1 / fc2 and the whole MLP backward run on that prefix
        # (`live` rows, a device scalar: tiles beyond it exit), the fc2 epilogue scatters back.  Same bits for every live row.
        # (the split-precision modes: the MLP's forward on half pairs - or, F16X2H, as plain f16 - and an f16 backward either way)
        plan = drop_plan if _plan_ok(torch.float16 if act_dtype in (F16X2, F16X2H) else act_dtype, m, d, fc1w.shape[0], cu, mask,
                                     m_live, branch16, defer_out, pend_branch, rowscale_mlp) else None
        ctx.plan = plan
        head_done = split_all = False
        if act_dtype in (F16X2, F16X2H):
            # split-precision forward: every product on half PAIRS (three MFMA passes, fp32-class), the exact softmax / GELU
            # on unrounded values; what is saved for the (16-bit) backward are the hi halves - plain f16 tensors
            head_only = act_dtype == F16X2H
            act_dtype = torch.float16
            inv_ws = 1.0 / ops.SPLIT_WSCALE
            wq = act_weight_split(qkvw)
            hidden = fc1w.shape[0]
            h1, h1l, mean1, rstd1 = ops.layernorm_fwd_split(x2d, n1w, n1b, eps, mask, m_live)
            qkv = torch.empty(m, 3 * d, dtype=act_dtype, device=x.device)
            qkvl = torch.empty_like(qkv)
            ops.gemm_split((h1, h1l), wq, qkv, qkvl, m, 3 * d, d, alpha=inv_ws, bias=qkvb, m_live=m_live)
            del h1l
            (ao, aol), attn_saved = ops.attention_fwd_split((qkv, qkvl), b, t, heads, hd, amask,
                                                            None if isinstance(probs_out, list) else probs_out, cu=cu,
                                                            scale=qk_scale)
            if isinstance(probs_out, list):            # no probability tensor: the rollout recomputes it from the q / k pairs
                probs_out.append((qkv, qkvl, attn_saved))
            del qkvl
            head_done, split_all = head_only, not head_only
        if head_done:
            del aol                          # F16X2H: the projection and the MLP below, on the hi halves, as the f16 mode
        if split_all:
            wp, w1, w2 = (act_weight_split(w) for w in (projw, fc1w, fc2w))
            x1 = torch.empty_like(x2d)
            ops.gemm_split((ao, aol), wp, x1, None, m, d, d, alpha=inv_ws, bias=projb, rowscale=rowscale_attn,
                           epilogue=ops.EPI_RESIDUAL, aux=x2d, m_live=m_live)
            del aol
            x2 = torch.empty_like(x2d)
            if plan is not None:         # stochastic-depth compaction (see above): the same three launches on the live prefix
                h2, h2l, mean2, rstd2 = ops.layernorm_fwd_split_perm(x1, n2w, n2b, eps, plan[0], rowscale_mlp, x2)
            else:
                h2, h2l, mean2, rstd2 = ops.layernorm_fwd_split(x1, n2w, n2b, eps, mask, m_live)
            live = plan[2] if plan is not None else m_live
            a = torch.empty(m, hidden, dtype=act_dtype, device=x.device) if (FWD_GRAD and any(ctx.needs_input_grad)) else None
            g = torch.empty(m, hidden, dtype=act_dtype, device=x.device)
            gl = torch.empty_like(g)
            ops.gemm_split((h2, h2l), w1, g, gl, m, hidden, d, alpha=inv_ws, bias=fc1b, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD,
                           aux=a, m_live=live, live_dense=plan is not None)
            del h2l
            ops.gemm_split((g, gl), w2, x2, None, m, d, hidden, alpha=inv_ws, bias=fc2b, rowscale=rowscale_mlp,
                           epilogue=ops.EPI_RESIDUAL, aux=x1, m_live=live, live_dense=plan is not None,
                           rowmap=plan[1] if plan is not None else None)
            del gl
            ctx.save_for_backward(x2d, mean1, rstd1, h1, qkv, ao, x1, mean2, rstd2, h2, a, g, n1w, n2w,
                                  qkvw, projw, fc1w, fc2w, mask, attn_saved, rowscale_attn, rowscale_mlp, cu, m_live)
            ctx.light = None
            ctx.gs = grad_scale(act_dtype)
            ctx.meta = (b, t, d, heads, act_dtype, qkvb is not None, projb is not None, fc1b is not None, fc2b is not None,
                        tuple(x.shape), qk_scale, sink)
            out = x2.view(x.shape)
            _cast_boxes(ctx, x, out, m, d, act_dtype, mask, cu, m_live, rowscale_mlp, fc2b is not None, sink, True,
                        plan[0] if plan is not None else None)
            return out
        wp, w1, w2 = (act_weight(w, act_dtype) for w in (projw, fc1w, fc2w))
        if not head_done:
            wq = act_weight(qkvw, act_dtype)
            if pend_branch is not None:          # x2d <- x1_prev + rs_prev * fc2-branch_prev, h1 = LN1(x2d): one pass
                x2d, h1, mean1, rstd1 = ops.resid_add_layernorm_fwd(x2d, pend_branch, pend_rs, n1w, n1b, eps)
            else:
                h1, mean1, rstd1 = ops.layernorm_fwd(x2d, n1w, n1b, eps, act_dtype, mask, 0, m_live=m_live)
            qkv = torch.empty(m, 3 * d, dtype=act_dtype, device=x.device)
            yield _GemmReq(h1, wq, qkv, m, 3 * d, d, d, d, 3 * d, 0, 0, bias=qkvb, m_live=m_live,
                           epilogue=STAGGER_QKV if m_live is None else 0)
            if isinstance(probs_out, list):            # bf16 backbone: no probability tensor; the rollout recomputes it
                ao, attn_saved = ops.attention_fwd(qkv, b, t, heads, hd, amask, None, cu=cu, scale=qk_scale)
                probs_out.append((qkv, attn_saved))
            else:
                ao, attn_saved = ops.attention_fwd(qkv, b, t, heads, hd, amask, probs_out, cu=cu, scale=qk_scale)
        if branch16:                                # 16-bit branch, plain epilogue; x1 = x + rs * branch inside LayerNorm-2's pass
            br1 = torch.empty(m, d, dtype=act_dtype, device=x.device)
            ops.gemm(ao, wp, br1, m, d, d, d, d, d, 0, 0, bias=projb)
            x1, h2, mean2, rstd2 = ops.resid_add_layernorm_fwd(x2d, br1, rowscale_attn, n2w, n2b, eps)
            del br1
        else:
            x1 = torch.empty_like(x2d)              # x1 = x + rs * (ao Wp^T + b): residual add in the GEMM epilogue
            yield _GemmReq(ao, wp, x1, m, d, d, d, d, d, 0, 0, bias=projb, rowscale=rowscale_attn,
                           epilogue=ops.EPI_RESIDUAL | (STAGGER_PROJ if m_live is None else 0), aux=x2d, m_live=m_live)
            if plan is not None:
                x2 = torch.empty_like(x2d)          # (dropped rows: LayerNorm-2 copies x1 there; live rows: the fc2 epilogue)
                h2, mean2, rstd2 = ops.layernorm_fwd_perm(x1, n2w, n2b, eps, act_dtype, plan[0], rowscale_mlp, x2)
            else:
                h2, mean2, rstd2 = ops.layernorm_fwd(x1, n2w, n2b, eps, act_dtype, mask, 0, m_live=m_live)
        hidden = w1.shape[0]
        # (a no-grad forward - model.eval() under torch.no_grad(), engine/processor.py:217-270 - saves nothing for a backward: the
        #  16-bit fc1 epilogue then writes ONE output instead of two, 304 MB per layer less at B = 128)
        need_a = (FWD_GRAD and any(ctx.needs_input_grad)) or act_dtype not in ops.HALF_DTYPES
        a = torch.empty(m, hidden, dtype=act_dtype, device=x.device) if need_a else None
        g = torch.empty(m, hidden, dtype=act_dtype, device=x.device)
        # 16-bit modes: `a` receives gelu'(pre-activation) - all the backward needs of it (one multiply in the fc2 dgrad
        # epilogue instead of an erfc + exponential per element); the f32 parity kernels keep the pre-activation
        light = ACT_LIGHT and act_dtype in ops.HALF_DTYPES
        ag = ops.EPI_AUX_GRAD if (act_dtype in ops.HALF_DTYPES and not light) else 0     # light: `a` keeps the pre-activation
        live = plan[2] if plan is not None else m_live
        yield _GemmReq(h2, w1, g, m, hidden, d, d, d, hidden, 0, 0, bias=fc1b, epilogue=ops.EPI_GELU | ag, aux=a, m_live=live,
                       live_dense=plan is not None)
        br2 = None
        if defer_out:                               # (x1, fc2 branch): the consumer's LayerNorm adds them
            br2 = torch.empty(m, d, dtype=act_dtype, device=x.device)
            ops.gemm(g, w2, br2, m, d, hidden, hidden, hidden, d, 0, 0, bias=fc2b)
        elif plan is not None:
            yield _GemmReq(g, w2, x2, m, d, hidden, hidden, hidden, d, 0, 0, bias=fc2b, rowscale=rowscale_mlp,
                           epilogue=ops.EPI_RESIDUAL, aux=x1, m_live=live, live_dense=True, rowmap=plan[1])
        else:
            x2 = torch.empty_like(x2d)
            yield _GemmReq(g, w2, x2, m, d, hidden, hidden, hidden, d, 0, 0, bias=fc2b, rowscale=rowscale_mlp,
                           epilogue=ops.EPI_RESIDUAL, aux=x1, m_live=m_live)
        if light:
            h1 = h2 = g = None                      # recomputed by the backward (n1b / n2b / eps ride along)
        ctx.save_for_backward(x2d, mean1, rstd1, h1, qkv, ao, x1, mean2, rstd2, h2, a, g, n1w, n2w,
                              qkvw, projw, fc1w, fc2w, mask, attn_saved, rowscale_attn, rowscale_mlp, cu, m_live)
        ctx.light = (n1b, n2b, float(eps)) if light else None
        ctx.gs = grad_scale(act_dtype)               # (the model's loss scale at forward time: the backward uses THIS one)
        ctx.meta = (b, t, d, heads, act_dtype, qkvb is not None, projb is not None, fc1b is not None, fc2b is not None,
                    tuple(x.shape), qk_scale, sink)
        if defer_out:
            ctx.mark_non_differentiable(br2)
            _CAST_SLOT[0] = None
            ctx.feeds_box = ctx.box = None
            return x1.view(x.shape), br2
        out = x2.view(x.shape)
        _cast_boxes(ctx, x, out, m, d, act_dtype, mask, cu, m_live, rowscale_mlp, fc2b is not None, sink,
                    pend_branch is None and not branch16, plan[0] if plan is not None else None)
        return out

    @staticmethod
    def backward_gen(ctx, dx2):
        (x2d, mean1, rstd1, h1, qkv, ao, x1, mean2, rstd2, h2, a, g, n1w, n2w, qkvw, projw, fc1w, fc2w, mask,
         attn_saved, rs_attn, rs_mlp, cu, m_live) = ctx.saved_tensors
        b, t, d, heads, act_dtype, hb_qkv, hb_proj, hb_fc1, hb_fc2, xshape, qk_scale, sink = ctx.meta
        # gradient outputs in forward-argument order: n1w n1b qkvw qkvb projw projb n2w n2b fc1w fc1b fc2w fc2b
        sv = sink.views if sink is not None else [None] * 12
        gs = ctx.gs                                  # f16: half gradients travel loss-scaled (1.0 otherwise)
        m = x2d.shape[0]
        hd = d // heads
        amask = None if cu is not None else mask
        wq, wp, w1, w2 = (act_weight(w, act_dtype) for w in (qkvw, projw, fc1w, fc2w))
        kmaj = DGRAD_KMAJOR and act_dtype in ops.HALF_DTYPES and m >= 2048
        wqt, wpt, w1t, w2t = ((act_weight_t(w, act_dtype) for w in (qkvw, projw, fc1w, fc2w)) if kmaj else (None,) * 4)
        dx2 = dx2.contiguous().view(m, d)
        light = ctx.light
        if a is None:
            # the forward ran with the grad mode off (FWD_GRAD False: fc1 then writes no gelu' tensor) - a backward through it would
            # silently drop gelu' from the fc2 dgrad; it cannot happen through EDITOR.forward, which restores the flag
            raise RuntimeError("TransformerBlockFn.backward: this block's forward saved no GELU derivative (it ran under "
                               "functional.set_model_options(grad_enabled=False)); re-run the forward with gradients enabled")
        if light is not None:
            # activation-light block: the GELU output for the fc2 weight gradient from the saved pre-activation (the fc2 dgrad
            # epilogue evaluates gelu' from it as well); the LayerNorm outputs right before their weight gradients need them
            g = ops.gelu_fwd(a)
        hidden = fc1w.shape[0]
        jobs = [] if (GROUP_WGRAD and act_dtype in ops.HALF_DTYPES and m >= 2048 and m % 64 == 0 and d % 256 == 0
                      and hidden % 256 == 0) else None
        deferred = jobs is not None and sink is not None and WGRAD_SIDE_STREAM and WGRAD_DEFER_JOIN
        # the block's six second-stage reductions (LayerNorm dgamma / dbeta x2, bias gradients, the fc2 dgrad's column sums) as
        # ONE launch at the end (ops.ReduceQueue); only with the grouped weight gradients, where every producer runs on this stream
        rq = ops.ReduceQueue(dx2.device, _GROUP_SLOT[0], _GROUP_SLOT[1]) if (DEFER_REDUCE and jobs is not None) else None
        # ---- MLP branch:  x2 = x1 + rs * fc2(gelu(fc1(LN2(x1))))
        box = getattr(ctx, "box", None)
        plan = getattr(ctx, "plan", None)            # stochastic-depth compaction of this block's MLP branch (see forward_gen)
        if plan is not None and (rq is None or jobs is None):
            raise RuntimeError("stochastic-depth compaction needs the grouped weight gradients + deferred reductions (its forward check)")
        live_mlp = plan[2] if plan is not None else m_live
        handed = box.take(dx2) if box is not None else None
        if handed is not None:
            dy, dbias = handed                       # left by the next block's LayerNorm-1 backward (HANDOFF_CAST; on plan[0]'s rows)
        elif plan is not None:
            dy, dbias = ops.cast_rows_colsum(dx2, rs_mlp, act_dtype, gs, sv[11], rq=rq, perm=plan[0])
        else:
            dy, dbias = _scaled_cast_colsum(dx2, rs_mlp, act_dtype, m_live, hb_fc2, gs, cs_out=sv[11], rq=rq)
        da, dw2, db2, da_cs = yield from _linear_bwd_gen(dy, g, w2, hb_fc2, gelu_pre=a, m_live=live_mlp, db=dbias,
                                          dx_colsum=hb_fc1, gs=gs, dw_out=sv[10], db_out=sv[11],
                                          dxcs_out=sv[9], w_t=w2t, defer=jobs, aux_is_grad=light is None, rq=rq,
                                          live_dense=plan is not None)   # da = (dy W2) * gelu'(a)
        if light is not None:
            h2 = ops.layernorm_fwd(x1, n2w, light[1], light[2], act_dtype, mask, 0, want_stats=False, m_live=m_live)[0]
        dh2, dw1, db1 = yield from _linear_bwd_gen(da, h2, w1, hb_fc1, m_live=live_mlp, db=da_cs, gs=gs, dw_out=sv[8], db_out=sv[9], w_t=w1t,
                                    defer=jobs, rq=rq, live_dense=plan is not None)
        fuse_cast = (FUSE_LN_CAST and m_live is None and mask is None and act_dtype in ops.HALF_DTYPES
                     and d % 256 == 0 and d <= 1024)
        if fuse_cast:
            # LN2's backward also writes the 16-bit, drop-path-scaled copy of dx1 (and its column sums = proj's bias
            # gradient) that the attention branch's backward starts from: no second pass over dx1
            dx1, dn2w, dn2b, dy, dbias = ops.layernorm_bwd_cast(
                dh2, x1, n2w, mean2, rstd2, dx2, rs_attn, gs, dy_scale=1.0 / gs,
                dgb_out=sink.ln_pair(6) if sink is not None else None, want_colsum=hb_proj, cs_out=sv[5], rq=rq,
                dy_perm=plan[0] if plan is not None else None, dy_live=plan[2] if plan is not None else None)
        else:
            dx1, dn2w, dn2b = ops.layernorm_bwd(dh2, x1, n2w, mean2, rstd2, mask, 0, dx_in=dx2, m_live=m_live,
                                                dy_scale=1.0 / gs, dgb_out=sink.ln_pair(6) if sink is not None else None, rq=rq)
            # ---- attention branch:  x1 = x + rs * proj(attn(qkv(LN1(x))))
            dy, dbias = _scaled_cast_colsum(dx1, rs_attn, act_dtype, m_live, hb_proj, gs, cs_out=sv[5], rq=rq)
        dao, dwp, dbp = yield from _linear_bwd_gen(dy, ao, wp, hb_proj, m_live=m_live, db=dbias, gs=gs, dw_out=sv[4], db_out=sv[5], w_t=wpt,
                                    defer=jobs, rq=rq)
        dbq = None
        if hb_qkv and ops.attention_bwd_colsum_ok(qkv, t, hd, attn_saved):
            # the qkv bias gradient = colsum(dqkv) from the attention backward's own accumulators (one partial row per sequence),
            # not from a pass over the 16-bit dqkv it has just written
            dbq = sv[3] if sv[3] is not None else torch.empty(qkv.shape[1], dtype=torch.float32, device=qkv.device)
        dqkv = ops.attention_bwd(qkv, dao, b, t, heads, hd, amask, attn_saved, ao, cu=cu, scale=qk_scale, colsum=dbq,
                                 colsum_scale=1.0 / gs, rq=rq)
        if light is not None:
            h1 = ops.layernorm_fwd(x2d, n1w, light[0], light[2], act_dtype, mask, 0, want_stats=False, m_live=m_live)[0]
        dh1, dwq, dbq = yield from _linear_bwd_gen(dqkv, h1, wq, hb_qkv, m_live=m_live, db=dbq, gs=gs, dw_out=sv[2], db_out=sv[3], w_t=wqt,
                                    defer=jobs, rq=rq)
        fb = getattr(ctx, "feeds_box", None)
        ln_done = False
        if jobs and WGRAD_LN and fb is not None and fuse_cast and not deferred and plan is None and fb.perm is None:
            # (opt-in, round 4) the block's four weight gradients AND LayerNorm-1's backward as two ROLES of one launch on this stream:
            # the memory-bound rows on a quarter of the CUs beside the tiles on the rest (ops.gemm_wgrad_group_ln)
            dx, dn1w, dn1b, dy_below, db_below = ops.gemm_wgrad_group_ln(
                jobs, m, 1.0 / gs, dh1, x2d, n1w, mean1, rstd1, dx1, fb.rowscale, fb.gs, dy_scale=1.0 / gs,
                dgb_out=sink.ln_pair(0) if sink is not None else None, want_colsum=fb.has_bias, cs_out=fb.cs_out, rq=rq)
            fb.put(dx, dy_below, db_below)
            ln_done = True
        elif jobs:
            # every dy exists: the block's four weight gradients in one launch.  Side stream (joined one block later)
            # unless a gradient sink needs them at the end of THIS block
            yield _WgradReq(jobs, m, 1.0 / gs, m_live, deferred, dx2.device)
        if ln_done:
            pass
        elif fb is not None and fuse_cast:
            # ... and hands the block BELOW the start of its backward (see HANDOFF_CAST): its drop-path row scale, its bias-gradient slot
            dx, dn1w, dn1b, dy_below, db_below = ops.layernorm_bwd_cast(
                dh1, x2d, n1w, mean1, rstd1, dx1, fb.rowscale, fb.gs, dy_scale=1.0 / gs,
                dgb_out=sink.ln_pair(0) if sink is not None else None, want_colsum=fb.has_bias, cs_out=fb.cs_out, rq=rq,
                cast_perm=fb.perm if rq is not None else None)
            if fb.perm is None or rq is not None:
                fb.put(dx, dy_below, db_below)       # (a compacted consumer without the parts form: it casts for itself)
        else:
            dx, dn1w, dn1b = ops.layernorm_bwd(dh1, x2d, n1w, mean1, rstd1, mask, 0, dx_in=dx1, m_live=m_live,
                                               dy_scale=1.0 / gs, dgb_out=sink.ln_pair(0) if sink is not None else None, rq=rq)
        if rq is not None:
            rq.flush()                               # the block's partial rows -> dgamma / dbeta / bias gradients, one launch
        if not deferred:
            join_side_stream(dx.device)              # the four weight gradients (side stream) are complete
        grads = (dn1w, dn1b, dwq, dbq, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2)
        if sink is not None:
            # the gradients already sit in the bucket (= the parameters' .grad): hand autograd nothing for them and let
            # the bucket's all-reduce start now, under the rest of the backward
            grads = tuple(None if v is not None else g_ for g_, v in zip(grads, sv))
            sink.done()
        return (dx.view(xshape),) + grads + (None,) * 18


GROUP_BLOCKS = os.environ.get("EDITOR_GROUP_BLOCKS", "1") != "0"      # A/B switch: GroupedBlocksFn for the HMA modality blocks


class GroupedBlocksFn(torch.autograd.Function):
    """`nblk` transformer blocks of IDENTICAL shape and options on different inputs and weights - the per-modality blocks of
    BlockMask.forward (vit_pytorch.py:311-317: blocksR / blocksN / blocksT) - as one autograd node whose forward and backward run
    the blocks in lockstep (TransformerBlockFn.forward_gen / backward_gen under _drive_group): every large product of the three blocks
    leaves as ONE grouped launch.  Arguments: nblk, then nblk x the argument list of TransformerBlockFn.forward (same length each).
    Returns the nblk outputs.  Bit-identical to nblk TransformerBlockFn nodes (tests/test_gpu_model.py)."""

    @staticmethod
    def forward(ctx, nblk, *flat):
        per = len(flat) // nblk
        assert per * nblk == len(flat) and per >= 20
        subs, gens = [], []
        for i in range(nblk):
            sc = _SubCtx(ctx.needs_input_grad[1 + i * per:1 + (i + 1) * per])
            subs.append(sc)
            gens.append(TransformerBlockFn.forward_gen(sc, *flat[i * per:(i + 1) * per]))
        outs = _drive_group(gens)
        saved = []
        for sc in subs:
            sc.nsaved = len(sc.saved_tensors)
            saved.extend(sc.saved_tensors)
            sc.saved_tensors = ()
        ctx.save_for_backward(*saved)
        ctx.subs, ctx.per = subs, per
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        saved, off = ctx.saved_tensors, 0
        gens = []
        for sc, dout in zip(ctx.subs, douts):
            sc.saved_tensors = saved[off:off + sc.nsaved]
            off += sc.nsaved
            gens.append(TransformerBlockFn.backward_gen(sc, dout))
        res = _drive_group(gens)
        for sc in ctx.subs:
            sc.saved_tensors = ()
        out = (None,)
        for r in res:
            out = out + tuple(r[:ctx.per])
        return out


def _scaled_cast_colsum(dx, rowscale, dtype, m_live, want_colsum, gs=1.0, cs_out=None, rq=None):
    """_scaled_cast, plus the (unscaled) column sums of the result when the consumer has a bias (dense 16-bit rows only)."""
    if (want_colsum and m_live is None and dtype in ops.HALF_DTYPES and dx.dtype == torch.float32
            and dx.shape[1] % 256 == 0 and dx.shape[1] <= 1024):
        return ops.cast_rows_colsum(dx, rowscale, dtype, gs, cs_out, rq=rq)
    return _scaled_cast(dx, rowscale, dtype, m_live, gs), None


def _scaled_cast(dx, rowscale, dtype, m_live=None, gs=1.0):
    if rowscale is None and dx.dtype == dtype and gs == 1.0:
        return dx
    return ops.cast_rows(dx, rowscale, dtype, m_live, gs)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed_overlap (vit_pytorch.py:449-458) + cls/pos/SIE assembly (:625-637) for stride 16.
    img (Btot,3,H,W) fp32 (modalities stacked on the batch axis), cam (Bcam) int64."""

    @staticmethod
    def forward(ctx, img, conv_w, conv_b, cls, pos, sie, cam, coef, act_dtype):
        # img: the stacked (Btot,3,H,W) batch, or the list of per-modality (B,3,H,W) tensors (no stacking copy)
        btot = sum(i.shape[0] for i in img) if isinstance(img, (list, tuple)) else img.shape[0]
        d = conv_w.shape[0]
        kdim = conv_w[0].numel()
        if act_dtype == F16X2:          # split-precision forward: fp32 patch rows from the half pairs; the backward is f16
            act_dtype = torch.float16
            cols, cols_lo = ops.im2col16_split(img)
            wh, wl = act_weight_split(conv_w)
            patch = torch.empty(cols.shape[0], d, dtype=torch.float32, device=cols.device)
            ops.gemm_split((cols, cols_lo), (wh.view(d, kdim), wl.view(d, kdim)), patch, None, cols.shape[0], d, kdim,
                           alpha=1.0 / ops.SPLIT_WSCALE, bias=conv_b)
            del cols_lo
        else:
            cols = ops.im2col16(img, act_dtype)
            w = act_weight(conv_w, act_dtype).view(d, kdim)
            patch = _linear_fwd(cols, w, conv_b, act_dtype)
        t = pos.shape[1]
        x = ops.embed_assemble(patch, cls.view(-1), pos.view(t, d), None if sie is None else sie.view(-1, d),
                               cam, coef, btot, t, d)
        ctx.save_for_backward(cols, conv_w, cam)
        ctx.gs = grad_scale(act_dtype)
        ctx.meta = (coef, act_dtype, None if sie is None else sie.shape[0], cls.shape, pos.shape,
                    None if sie is None else sie.shape)
        return x

    @staticmethod
    def backward(ctx, dx):
        cols, conv_w, cam = ctx.saved_tensors
        coef, act_dtype, ncam, cls_shape, pos_shape, sie_shape = ctx.meta
        join_side_stream(dx.device)      # last node of the backbone's backward: the deferred grouped weight gradients are complete
        dx = dx.contiguous()
        gs = ctx.gs
        dpatch, dpos, dsie = ops.embed_assemble_bwd(dx, cam, ncam or 0, coef, act_dtype, gs)
        d = conv_w.shape[0]
        kdim = cols.shape[1]
        mrows = cols.shape[0]
        dw = torch.empty(d, kdim, dtype=torch.float32, device=dx.device)
        sk, skf = _splitk_for(d, kdim, mrows)
        ops.gemm(dpatch, cols, dw, d, kdim, mrows, d, kdim, kdim, 1, 1, alpha=1.0 / gs, splitk=sk, epilogue=skf)
        db = ops.colsum(dpatch, scale=1.0 / gs)
        dcls = dpos[0].clone().view(cls_shape)
        return (None, dw.view(conv_w.shape), db, dcls, dpos.view(pos_shape),
                None if dsie is None else dsie.view(sie_shape), None, None, None)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm with fp32 output, optionally followed by the row re-mask of BlockMask
    (out_norm + `x * mask3`, vit_pytorch.py:329-332) or the backbone's final norm (:643)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, mask, m_live=None):
        shape = x.shape
        x2d = x.reshape(-1, shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2d, w, b, eps, torch.float32, mask, 0, m_live=m_live)
        ctx.save_for_backward(x2d, w, mean, rstd, mask, m_live)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2d, w, mean, rstd, mask, m_live = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(dy.contiguous().view(x2d.shape), x2d, w, mean, rstd, mask, 0, m_live=m_live)
        return dx.view(dy.shape), dw, db, None, None, None


class SFTSApplyFn(torch.autograd.Function):
    """SFTS.forward's differentiable half (SFTS.py:208-225): zero the unselected patch tokens of every
    modality and (training) the background-consistency loss over the unselected ones."""

    @staticmethod
    def forward(ctx, feat, index, training):
        # Also hands out the (nmod, B, D) cls tokens (SFTS leaves them untouched): taking them with feat[i, :, 0] makes
        # autograd build a zero (nmod,B,T,D) tensor per modality and add it to this node's gradient (three 152 MB fills
        # and adds per step); here their gradient is added into row 0 of the gradient this node writes anyway.
        out, loss = ops.sfts_apply(feat, index, training)
        ctx.save_for_backward(feat, index)
        ctx.training = training
        cls = feat[:, :, 0].contiguous()
        if training:
            return out, loss.view(()), cls
        return out, feat.new_zeros(()), cls

    @staticmethod
    def backward(ctx, dout, dloss, dcls):
        feat, index = ctx.saved_tensors
        dl = dloss.contiguous().view(1).float() if ctx.training else None
        dfeat = ops.sfts_apply_bwd(feat, index, dout.contiguous(), dl)
        if dcls is not None:
            dfeat[:, :, 0] += dcls
        return dfeat, None, None


class PoolFn(torch.autograd.Function):
    """cls + masked-mean pooling per modality of the fused tokens (make_model.py:186-203)."""

    @staticmethod
    def forward(ctx, x, nmod, t):
        out, num = ops.pool_fwd(x.contiguous(), nmod, t)
        ctx.save_for_backward(num)
        ctx.meta = (nmod, t)
        ctx.mark_non_differentiable(num)
        return out, num

    @staticmethod
    def backward(ctx, dout, _dnum):
        (num,) = ctx.saved_tensors
        nmod, t = ctx.meta
        return ops.pool_bwd(dout.contiguous(), num, nmod, t), None, None


class LinearFn(torch.autograd.Function):
    """Small fp32 nn.Linear (REDUCE layers, classifier heads; make_model.py:162-171,205-209)."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = _linear_fwd(x.contiguous(), w.detach(), b, torch.float32)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = _linear_bwd(dy.contiguous(), x.contiguous(), w.detach(), ctx.has_bias)
        return dx, dw, db


class BatchNorm1dFn(torch.autograd.Function):
    """nn.BatchNorm1d (FUSE_BN / BACKBONE_BN / AL_BN, make_model.py:115,120,140); running stats updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, momentum, eps, training):
        y, sm, si = ops.bn1d_fwd(x, gamma, beta, rmean, rvar, momentum, eps, training)
        ctx.save_for_backward(x, gamma, sm, si)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, sm, si = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("BatchNorm1dFn.backward in eval mode is not part of the hot path")
        dx, dg, db = ops.bn1d_bwd(dy.contiguous(), x, gamma, sm, si)
        return dx, dg, db, None, None, None, None, None


class OCFRFn(torch.autograd.Function):
    """OCFR.forward (OCFR.py:44-84) over the modalities: returns the summed intra loss and updates the centre
    tables (non-grad Parameters) in place.  Arguments after nmod: nmod cls-feature tensors, then nmod centre tables."""

    @staticmethod
    def forward(ctx, label, momentum, nmod, *fc):
        feats, centers = fc[:nmod], fc[nmod:]
        loss = torch.empty(1, dtype=torch.float32, device=feats[0].device)
        saved = []
        for i, (f, c) in enumerate(zip(feats, centers)):
            fn, inv = ops.ocfr_fwd(f, label, c.data, momentum, loss, accumulate=i > 0)
            saved += [fn, inv, c.data.clone()]      # centres are overwritten by later steps: keep this step's copy
        ctx.save_for_backward(label, *saved)
        ctx.nmod = nmod
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        label = ctx.saved_tensors[0]
        sv = ctx.saved_tensors[1:]
        dl = dloss.contiguous().view(1).float()
        grads = [ops.ocfr_bwd(sv[3 * i], sv[3 * i + 1], sv[3 * i + 2], label, dl) for i in range(ctx.nmod)]
        return (None, None, None) + tuple(grads) + (None,) * ctx.nmod


class CrossEntropyLabelSmoothFn(torch.autograd.Function):
    """CrossEntropyLabelSmooth.forward (layers/softmax_loss.py:21-34)."""

    @staticmethod
    def forward(ctx, logits, target, eps):
        logits = logits.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        ops.ce_smooth_fwd(logits, target, eps, loss, accumulate=False)
        ctx.save_for_backward(logits, target)
        ctx.eps = eps
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        return ops.ce_smooth_bwd(logits, target, ctx.eps, dloss.contiguous().view(1).float()), None, None


class CenterLossFn(torch.autograd.Function):
    """CenterLoss.forward (layers/center_loss.py:30-51)."""

    @staticmethod
    def forward(ctx, x, centers, label):
        x, centers = x.contiguous().float(), centers.contiguous().float()
        loss, dist = ops.center_loss_fwd(x, centers, label.contiguous())
        ctx.save_for_backward(x, centers, label, dist)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        x, centers, label, dist = ctx.saved_tensors
        dx, dc = ops.center_loss_bwd(x, centers, label, dist, dloss.contiguous().view(1).float(), ctx.needs_input_grad[0],
                                     ctx.needs_input_grad[1])
        return dx, dc, None


class TripletSoftMarginFn(torch.autograd.Function):
    """TripletLoss(margin=None).forward (layers/triplet_loss.py:121-136): batch-hard mining + SoftMarginLoss."""

    @staticmethod
    def forward(ctx, feat, label):
        feat = feat.float()
        if feat.stride(1) != 1:
            feat = feat.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=feat.device)
        idx, coef = ops.triplet_fwd(feat, label, loss, accumulate=False)
        ctx.save_for_backward(feat, idx, coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        feat, idx, coef = ctx.saved_tensors
        return ops.triplet_bwd(feat, idx, coef, dloss.contiguous().view(1).float()), None


class GatherRowsFn(torch.autograd.Function):
    """out[r] = x2d[src[r]] (zeros where src[r] < 0): row movement between the dense token tensor and the packed
    layouts of the compacted HMA head.  Every source row is gathered at most once, so backward is a plain scatter."""

    @staticmethod
    def forward(ctx, x2d, src, live=None, live_mul=1, live_stride=0, bwd_fill="all"):
        # bwd_fill = "none": the gradient's rows that no index names are left UNWRITTEN instead of zero-filled (152 MB of memset,
        # 103 us) - only where the one consumer of that gradient provably reads the gathered rows alone (the layout-A gather of
        # the compacted HMA head: its gradient goes straight into SFTSApplyFn.backward, which reads selected rows only)
        ctx.save_for_backward(src)
        ctx.rows_in = x2d.shape[0]
        ctx.bwd_fill = bwd_fill
        return ops.gather_rows(x2d.contiguous(), src, live, live_mul, live_stride)

    @staticmethod
    def backward(ctx, dy):
        (src,) = ctx.saved_tensors
        return ops.scatter_rows(dy.contiguous(), src, ctx.rows_in, fill=ctx.bwd_fill), None, None, None, None, None


class GatherPairFn(torch.autograd.Function):
    """The two gathers of the compacted HMA head that read the SAME packed tensor - the cls rows of every (modality, sample)
    for OCFR, and the sample-major layout B of the joint block - as one node: backward is one zero-fill + scatter plus a
    384-row add, where two GatherRowsFn nodes made autograd zero-fill a second (rows, D) tensor for the cls rows and add
    the two (2 x 152 MB per step)."""

    @staticmethod
    def forward(ctx, x2d, src_cls, src_b, live, live_mul, live_stride):
        x2d = x2d.contiguous()
        ctx.save_for_backward(src_cls, src_b, live)
        ctx.rows_in = x2d.shape[0]
        ctx.nseg = int(live_mul)
        return ops.gather_rows(x2d, src_cls), ops.gather_rows(x2d, src_b, live, live_mul, live_stride)

    @staticmethod
    def backward(ctx, dcls, dxb):
        src_cls, src_b, live = ctx.saved_tensors
        # the input (layout A) is `nseg` segments of rows_in / nseg rows with `live` live rows each; every live row is the source
        # of exactly one layout-B row, so the scatter writes them all: only the pad rows up to the next multiple of 64 - which the
        # live-row kernels of the per-modality blocks read - are zeroed, not the whole 152 MB
        if live is not None and ctx.rows_in % ctx.nseg == 0:
            dx = ops.scatter_rows(dxb.contiguous(), src_b, ctx.rows_in, fill="tail", live=live, seg_rows=ctx.rows_in // ctx.nseg)
        else:
            dx = ops.scatter_rows(dxb.contiguous(), src_b, ctx.rows_in)
        if dcls is not None:
            dx.index_add_(0, src_cls.long(), dcls.contiguous())      # (every cls row once: deterministic)
        return dx, None, None, None, None, None


class PoolPackedFn(torch.autograd.Function):
    """make_model.py:186-203 on the sample-major packed layout."""

    @staticmethod
    def forward(ctx, x2d, cu, b, nmod, live=None):
        # live (device int32 scalar = nmod * cu[b], the live rows of layout B): the backward then zeroes only the pad rows the
        # live-row LayerNorm reads beyond them instead of zero-filling the worst-case-sized gradient
        out, num = ops.pool_packed_fwd(x2d.contiguous(), cu, b, nmod)
        ctx.save_for_backward(num, cu, live)
        ctx.meta = (b, nmod, x2d.shape[0])
        ctx.mark_non_differentiable(num)
        return out, num

    @staticmethod
    def backward(ctx, dout, _dnum):
        num, cu, live = ctx.saved_tensors
        b, nmod, rows = ctx.meta
        return ops.pool_packed_bwd(dout.contiguous(), num, cu, b, nmod, rows, live), None, None, None, None

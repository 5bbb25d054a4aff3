"""Thin tensor-level wrappers over the C ABI (include/editor_hip.h).  No math happens here: each
function allocates outputs with torch (device memory plumbing) and launches HIP kernels on the
current stream.  Every function raises on CPU tensors - there is no fallback path."""
import ctypes
import os

import torch

from . import _lib

call = _lib.call
_raw_stream = _lib._raw_stream


# ---------------------------------------------------------------------------------------------
# token selection
# ---------------------------------------------------------------------------------------------
def freq_counts(rgb, nir, tir, m4=None):
    """Frequency.py:65-84,42-56 -> (B, N) int32 positive-pixel counts per 16x16 patch (m4: 4-modal extension)."""
    b, c, h, w = rgb.shape
    counts = torch.empty(b, (h // 16) * (w // 16), dtype=torch.int32, device=rgb.device)
    if m4 is not None:
        call("editor_freq_counts_nmod_f32", rgb.contiguous(), nir.contiguous(), tir.contiguous(), m4.contiguous(), 4,
             b, c, h, w, counts)
    else:
        call("editor_freq_counts_f32", rgb.contiguous(), nir.contiguous(),
             None if tir is None else tir.contiguous(), b, c, h, w, counts)
    return counts


def topk_mask(vals, k, group=1):
    """torch.topk -> sort -> scatter_ bool rows (torch CPU tie order); `group` rows OR together."""
    rows, n = vals.shape
    mask = torch.empty(rows // group, n, dtype=torch.uint8, device=vals.device)
    if vals.dtype == torch.int32:
        call("editor_topk_mask_i32", vals.contiguous(), rows, n, int(k), int(group), mask)
    elif vals.dtype == torch.float32:
        call("editor_topk_mask_f32", vals.contiguous(), rows, n, int(k), int(group), mask)
    else:
        raise TypeError(vals.dtype)
    return mask


def frequency_mask(rgb, nir, tir, keep, m4=None):
    counts = freq_counts(rgb, nir, tir, m4)
    return topk_mask(counts, keep), counts


def attn_rollout(probs):
    """probs: (L, B, H, T, ldp) fp32 contiguous (ldp >= T: padded rows) -> (B, H, T-1) CLS-row rollout scores
    (SFTS.py:150-153)."""
    l, b, h, t, ldp = probs.shape
    scores = torch.empty(b, h, t - 1, dtype=torch.float32, device=probs.device)
    call("editor_attn_rollout_f32", probs, l, b * h, t, ldp, b * h * t * ldp, scores)
    return scores


def attn_rollout_qk(layers, b, t, heads, hd, scale=None):
    """Rollout scores (B, H, T-1) from the per-layer (qkv, lse) pairs of the bf16 backbone, first layer first:
    r = e_cls^T A_{L-1}; r <- r A_l for l = L-2 .. 0, each A_l recomputed on the fly (no probability tensor)."""
    dev = layers[0][0].device
    scores = torch.empty(b, heads, t - 1, dtype=torch.float32, device=dev)
    bufs = [torch.empty(b * heads, t, dtype=torch.float32, device=dev) for _ in range(2)]
    r_in = None
    for i, layer in enumerate(reversed(layers)):
        last = i == len(layers) - 1
        out = scores if last else bufs[i & 1]
        if len(layer) == 3:         # split-precision layer: (qkv_hi, qkv_lo, lse) - scores in three passes, fp32-class
            qkv, qkv_lo, lse = layer
            call("editor_attn_rollout_step_f16x2", qkv, qkv_lo, lse, r_in, b, t, heads, hd, float(scale or hd ** -0.5), out,
                 1 if last else 0)
        else:
            qkv, lse = layer
            call(_h16(qkv, "attn_rollout_step"), qkv, lse, r_in, b, t, heads, hd, float(scale or hd ** -0.5), out,
                 1 if last else 0)
        r_in = out
    return scores


def mask_or(a, b=None, c=None, d=None):
    out = torch.empty_like(a)
    call("editor_mask_or", a, b, c, d, out, a.numel())
    return out


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
_WS = {}
_WS_RETIRED = []


def workspace(device, nfloats):
    """fp32 scratch for partial reductions (kernels never allocate): one buffer per device AND stream, so that the
    weight-gradient products running on the side stream never share partial-sum slabs with the main stream.
    A buffer that was handed out is NEVER freed: a captured hipGraph keeps replaying launches that hold its address, so
    when a later request outgrows it the old block is retired (kept alive), not returned to the allocator."""
    key = (device.type, device.index, _raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
    ws = _WS.get(key)
    if ws is None or ws.numel() < nfloats:
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = torch.empty(max(int(nfloats), 1 << 22), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


_ARENA = {}


class ReduceQueue:
    """Deferred second stages of the deterministic two-stage reductions of ONE transformer block's backward (LayerNorm dgamma /
    dbeta, bias gradients, the dgrad epilogue's column sums): the producers leave their partial rows in regions of a per-stream
    arena (the `_parts` entry points) and `flush()` folds up to eight sets with one editor_reduce_rows_multi launch - nothing needs
    the totals before the block ends.  Round 4: 95 -> ~31 reduce launches per step, same bits (same summation order)."""
    ARENA_FLOATS = 16 << 20           # 64 MiB; one block's six sets are ~31 MB at D = 768, ~42 MB at D = 1024

    def __init__(self, device, slot=0, nslots=1):
        # slot / nslots: blocks whose backwards run in LOCKSTEP (functional.GroupedBlocksFn) each take their own part of the arena -
        # their partial rows are live at the same time
        key = (device.index, _raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
        buf = _ARENA.get(key)
        if buf is None:               # never freed: a captured hipGraph replays launches that hold its address
            buf = _ARENA[key] = torch.empty(self.ARENA_FLOATS, dtype=torch.float32, device=device)
        span = (self.ARENA_FLOATS // int(nslots)) // 64 * 64
        self.base, self.limit = int(slot) * span, (int(slot) + 1) * span
        self.buf, self.off, self.jobs = buf, self.base, []

    def region(self, nfloats):
        n = (int(nfloats) + 63) // 64 * 64
        if n > self.limit - self.base:
            return None               # (does not fit at all: the caller reduces on the spot)
        if self.off + n > self.limit or len(self.jobs) >= 7:
            self.flush()
        r = self.buf[self.off:self.off + n]
        self.off += n
        return r

    def add(self, partials, nparts, ncol, out, scale=1.0):
        if len(self.jobs) == 8:
            self.flush(reset=False)   # (the region of the job being added is live: keep the arena offset)
        self.jobs.append((partials, int(nparts), int(ncol), out, float(scale)))

    def flush(self, reset=True):
        n = len(self.jobs)
        if n:
            parts = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in self.jobs])
            cnt = (ctypes.c_int * n)(*[j[1] for j in self.jobs])
            ncol = (ctypes.c_long * n)(*[j[2] for j in self.jobs])
            outs = (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in self.jobs])
            scale = (ctypes.c_float * n)(*[j[4] for j in self.jobs])
            with torch.cuda.device(self.buf.device):
                call("editor_reduce_rows_multi", n, parts, cnt, ncol, outs, scale)
        self.jobs = []
        if reset:
            self.off = self.base


_DT_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
HALF_DTYPES = (torch.bfloat16, torch.float16)


def _is_bf16(t):
    """dtype code of the C ABI (include/editor_hip.h): 0 = fp32, 1 = bf16, 2 = f16."""
    return _DT_CODE[t.dtype]


def _h16(t, name):
    """entry point of the 16-bit kernel family for tensor t's dtype (editor_<name>_bf16 / _f16)."""
    return "editor_" + name + ("_f16" if t.dtype == torch.float16 else "_bf16")


def _ptr(t, off=0):
    """raw device address of element `off` of tensor t (for sub-matrix operands)."""
    if not t.is_cuda:
        raise RuntimeError("EDITOR ops need GPU tensors (no CPU fallback)")
    return t.data_ptr() + off * t.element_size()


WS_ROWS = 1024


# ---------------------------------------------------------------------------------------------
# row kernels
# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps, out_dtype, rowmask=None, mask_period=0, want_stats=True, m_live=None):
    m, d = x2d.shape
    y = torch.empty(m, d, dtype=out_dtype, device=x2d.device)
    mean = torch.empty(m, dtype=torch.float32, device=x2d.device) if want_stats else None
    rstd = torch.empty(m, dtype=torch.float32, device=x2d.device) if want_stats else None
    call("editor_layernorm_fwd", x2d, gamma, beta, float(eps), m, d, rowmask, int(mask_period), y, _is_bf16(y), mean, rstd,
         m_live)
    return y, mean, rstd


def layernorm_fwd_perm(x2d, gamma, beta, eps, out_dtype, perm, rowscale, copy_out):
    """layernorm_fwd onto COMPACTED rows (stochastic-depth skipping, droppath_plan): row r's 16-bit output at row perm[r] of y;
    a dropped row (rowscale[r] == 0) leaves zeros there and copies its x row into copy_out.  -> y, mean, rstd (original rows)."""
    m, d = x2d.shape
    y = torch.empty(m, d, dtype=out_dtype, device=x2d.device)
    mean = torch.empty(m, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2d.device)
    call("editor_layernorm_fwd_perm", x2d, gamma, beta, float(eps), m, d, y, _is_bf16(y), mean, rstd, perm, rowscale, copy_out)
    return y, mean, rstd


def resid_add_layernorm_fwd(x2d, branch, rowscale, gamma, beta, eps):
    """x_out = x + rowscale[:, None] * branch (branch: 16-bit, plain-epilogue output of the projection / fc2 product), y = LN(x_out)
    in branch's dtype -> (x_out fp32, y, mean, rstd).  Dense rows, D a multiple of 256 (cfg.MODEL.BRANCH16)."""
    m, d = x2d.shape
    x_out = torch.empty_like(x2d)
    y = torch.empty(m, d, dtype=branch.dtype, device=x2d.device)
    mean = torch.empty(m, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2d.device)
    call("editor_resid_add_layernorm_fwd", x2d, branch, _is_bf16(branch), rowscale, gamma, beta, float(eps), m, d, x_out, y, mean, rstd)
    return x_out, y, mean, rstd


def layernorm_bwd(dy, x2d, gamma, mean, rstd, rowmask=None, mask_period=0, dx_in=None, want_param_grads=True,
                  m_live=None, dy_scale=1.0, dgb_out=None, rq=None):
    """dgb_out: optional (2, D) fp32 view that receives [dgamma; dbeta] (adjacent slots of a gradient bucket).
    rq (ReduceQueue): leave the dgamma / dbeta partial rows for the queue's one fold at the end of the block's backward."""
    m, d = x2d.shape
    dx = torch.empty(m, d, dtype=torch.float32, device=x2d.device)
    if dgb_out is not None:
        dgb = dgb_out
    else:
        dgb = torch.empty(2, d, dtype=torch.float32, device=x2d.device) if want_param_grads else None
    dg = dgb[0] if want_param_grads else None
    db = dgb[1] if want_param_grads else None
    ws = rq.region(WS_ROWS * 2 * d) if (rq is not None and want_param_grads) else None
    if ws is not None:
        npart = ctypes.c_int(0)
        call("editor_layernorm_bwd_parts", dy, _is_bf16(dy), float(dy_scale), x2d, gamma, mean, rstd, m, d, rowmask, int(mask_period),
             dx_in, dx, ws, WS_ROWS, m_live, ctypes.byref(npart))
        rq.add(ws, npart.value, 2 * d, dgb, 1.0)
        return dx, dg, db
    ws = workspace(x2d.device, WS_ROWS * 2 * d)
    call("editor_layernorm_bwd", dy, _is_bf16(dy), float(dy_scale), x2d, gamma, mean, rstd, m, d, rowmask, int(mask_period), dx_in, dx,
         dg, db, ws, WS_ROWS, m_live)
    return dx, dg, db


def layernorm_bwd_cast(dy, x2d, gamma, mean, rstd, dx_in, rowscale, scale=1.0, dy_scale=1.0, dgb_out=None, want_colsum=True,
                       cs_out=None, rq=None, dy_perm=None, dy_live=None, cast_perm=None):
    """layernorm_bwd (dense 16-bit rows) that also hands out what cast_rows_colsum(dx, rowscale, dy.dtype, scale) would:
    -> dx, dgamma, dbeta, dx16, colsum(dx16) / scale (None unless want_colsum).
    dy_perm / dy_live: dy sits on the compacted rows of a stochastic-depth plan (slots >= *dy_live: dropped rows, gradient zero);
    cast_perm: dx16 is written onto the compacted rows of the branch that consumes it (droppath_plan)."""
    m, d = x2d.shape
    if dy_perm is not None or cast_perm is not None:
        if rq is None:
            raise RuntimeError("layernorm_bwd_cast on compacted rows: the deferred-reduction (parts) form only")
        dx = torch.empty(m, d, dtype=torch.float32, device=x2d.device)
        dgb = dgb_out if dgb_out is not None else torch.empty(2, d, dtype=torch.float32, device=x2d.device)
        dx16 = torch.empty(m, d, dtype=dy.dtype, device=x2d.device)
        cs = (cs_out if cs_out is not None else torch.empty(d, dtype=torch.float32, device=x2d.device)) if want_colsum else None
        ws = rq.region(WS_ROWS * 3 * d)
        npart = ctypes.c_int(0)
        call("editor_layernorm_bwd_cast_perm_parts", dy, _is_bf16(dy), float(dy_scale), x2d, gamma, mean, rstd, m, d, dx_in, dx, ws,
             WS_ROWS, dx16, rowscale, float(scale), 1 if want_colsum else 0, dy_perm, dy_live, cast_perm, ctypes.byref(npart))
        rq.add(ws, npart.value, 2 * d, dgb, 1.0)
        if want_colsum:
            rq.add(ws[WS_ROWS * 2 * d:], npart.value, d, cs, 1.0 / float(scale))
        return dx, dgb[0], dgb[1], dx16, cs
    dx = torch.empty(m, d, dtype=torch.float32, device=x2d.device)
    dgb = dgb_out if dgb_out is not None else torch.empty(2, d, dtype=torch.float32, device=x2d.device)
    dx16 = torch.empty(m, d, dtype=dy.dtype, device=x2d.device)
    cs = None
    if want_colsum:
        cs = cs_out if cs_out is not None else torch.empty(d, dtype=torch.float32, device=x2d.device)
    ws = rq.region(WS_ROWS * 3 * d) if rq is not None else None
    if ws is not None:
        npart = ctypes.c_int(0)
        call("editor_layernorm_bwd_cast_parts", dy, _is_bf16(dy), float(dy_scale), x2d, gamma, mean, rstd, m, d, dx_in, dx, ws, WS_ROWS,
             dx16, rowscale, float(scale), 1 if want_colsum else 0, ctypes.byref(npart))
        rq.add(ws, npart.value, 2 * d, dgb, 1.0)
        if want_colsum:
            rq.add(ws[WS_ROWS * 2 * d:], npart.value, d, cs, 1.0 / float(scale))
        return dx, dgb[0], dgb[1], dx16, cs
    ws = workspace(x2d.device, WS_ROWS * 3 * d)
    call("editor_layernorm_bwd_cast", dy, _is_bf16(dy), float(dy_scale), x2d, gamma, mean, rstd, m, d, dx_in, dx, dgb[0], dgb[1],
         ws, WS_ROWS, dx16, rowscale, float(scale), cs, 1.0 / float(scale))
    return dx, dgb[0], dgb[1], dx16, cs


def colsum(dy, out=None, scale=1.0, rq=None):
    m, n = dy.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dy.device)
    ws = rq.region(WS_ROWS * n) if rq is not None else None
    if ws is not None:
        npart = ctypes.c_int(0)
        call("editor_colsum_parts", dy, _is_bf16(dy), m, n, n, ws, WS_ROWS, ctypes.byref(npart))
        rq.add(ws, npart.value, n, out, float(scale))
        return out
    ws = workspace(dy.device, WS_ROWS * n)
    call("editor_colsum", dy, _is_bf16(dy), m, n, n, out, ws, WS_ROWS, float(scale))
    return out


def gelu_fwd(a):
    g = torch.empty_like(a)
    call("editor_gelu_fwd", a, g, a.numel(), _is_bf16(a))
    return g


def gelu_bwd(a, dg):
    da = torch.empty_like(a)
    call("editor_gelu_bwd", a, dg, da, a.numel(), _is_bf16(a))
    return da


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    if x.dtype == torch.float32 and dtype == torch.bfloat16:
        call("editor_cast_f32_to_bf16", x, out, x.numel())
    elif x.dtype == torch.bfloat16 and dtype == torch.float32:
        call("editor_cast_bf16_to_f32", x, out, x.numel())
    elif x.dtype == torch.float32 and dtype == torch.float16:
        call("editor_cast_f32_to_f16", x, out, x.numel())
    elif x.dtype == torch.float16 and dtype == torch.float32:
        call("editor_cast_f16_to_f32", x, out, x.numel())
    else:
        raise TypeError((x.dtype, dtype))
    return out


def cast_rows(x2d, rowscale, dtype, m_live=None, scale=1.0):
    """x * rowscale[:, None] * scale cast to `dtype` (one pass)."""
    m, d = x2d.shape
    out = torch.empty(m, d, dtype=dtype, device=x2d.device)
    call("editor_cast_rows", x2d, rowscale, m, d, out, _is_bf16(out), m_live, float(scale))
    return out


def cast_rows_colsum(x2d, rowscale, dtype, scale=1.0, cs_out=None, rq=None, perm=None):
    """cast_rows + the column sums of its output (bias gradient, with the scale removed again) in the same pass.
    perm: row r of the result goes to row perm[r] (compacted consumer rows, droppath_plan)."""
    m, d = x2d.shape
    out = torch.empty(m, d, dtype=dtype, device=x2d.device)
    cs = cs_out if cs_out is not None else torch.empty(d, dtype=torch.float32, device=x2d.device)
    ws = rq.region(WS_ROWS * d) if rq is not None else None
    if perm is not None:
        if ws is None:
            raise RuntimeError("cast_rows_colsum onto compacted rows: the deferred-reduction (parts) form only")
        npart = ctypes.c_int(0)
        call("editor_cast_rows_colsum_perm_parts", x2d, rowscale, m, d, out, _is_bf16(out), ws, WS_ROWS, float(scale), perm,
             ctypes.byref(npart))
        rq.add(ws, npart.value, d, cs, 1.0 / float(scale))
        return out, cs
    if ws is not None:
        npart = ctypes.c_int(0)
        call("editor_cast_rows_colsum_parts", x2d, rowscale, m, d, out, _is_bf16(out), ws, WS_ROWS, float(scale), ctypes.byref(npart))
        rq.add(ws, npart.value, d, cs, 1.0 / float(scale))
        return out, cs
    ws = workspace(x2d.device, WS_ROWS * d)
    call("editor_cast_rows_colsum", x2d, rowscale, m, d, out, _is_bf16(out), cs, ws, WS_ROWS, float(scale), 1.0 / float(scale))
    return out, cs


def im2col16(img, dtype):
    """img: (B,C,H,W) fp32, or a list of such tensors (the modalities) laid out as if stacked on the batch axis - without
    the 151 MB `torch.cat` that stacking them costs per step."""
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    b, c, h, w = imgs[0].shape
    rows = b * (h // 16) * (w // 16)
    out = torch.empty(len(imgs) * rows, c * 256, dtype=dtype, device=imgs[0].device)
    for i, im in enumerate(imgs):
        if tuple(im.shape) != (b, c, h, w):
            raise ValueError("im2col16: modality tensors of different shapes")
        call("editor_im2col16", im, b, c, h, w, _ptr(out, i * rows * c * 256), _is_bf16(out))
    return out


def embed_assemble(patch, cls, pos, sie, cam, coef, btot, t, d):
    x = torch.empty(btot, t, d, dtype=torch.float32, device=patch.device)
    call("editor_embed_assemble", patch, _is_bf16(patch), cls, pos, sie, cam, 0 if cam is None else cam.numel(),
         float(coef), btot, t, d, x)
    return x


def embed_assemble_bwd(dx, cam, ncam, coef, dtype, scale=1.0):
    btot, t, d = dx.shape
    dpatch = torch.empty(btot * (t - 1), d, dtype=dtype, device=dx.device)
    dpos = torch.empty(t, d, dtype=torch.float32, device=dx.device)
    dsie = torch.empty(ncam, d, dtype=torch.float32, device=dx.device) if ncam else None
    ws = workspace(dx.device, max(btot * d, 8 * t * d))          # EDITOR_EMBED_POS_SPLITS partial rows / per-sample row sums
    call("editor_embed_assemble_bwd", dx, cam, 0 if cam is None else cam.numel(), int(ncam), float(coef), btot, t, d,
         dpatch, _is_bf16(dpatch), float(scale), dpos, dsie, ws)
    return dpatch, dpos, dsie


def sfts_apply(feat, index, want_loss):
    nmod, b, t, d = feat.shape
    out = torch.empty_like(feat)
    loss = torch.empty(1, dtype=torch.float32, device=feat.device) if want_loss else None
    nblk = (b * t + 7) // 8                      # one partial BCC sum per block of 8 token rows (csrc/norm.hip SFTS_ROWS)
    ws = workspace(feat.device, nblk)
    call("editor_sfts_apply", feat, index, nmod, b, t, d, out, loss, ws, nblk)
    return out, loss


def sfts_apply_bwd(feat, index, dout, dloss):
    nmod, b, t, d = feat.shape
    dfeat = torch.empty_like(feat)
    call("editor_sfts_apply_bwd", feat, index, dout, dloss, nmod, b, t, d, dfeat)
    return dfeat


def pool_fwd(x, nmod, t):
    b, _, d = x.shape
    out = torch.empty(nmod, b, 2 * d, dtype=torch.float32, device=x.device)
    num = torch.empty(b, dtype=torch.float32, device=x.device)
    call("editor_pool_fwd", x, b, nmod, t, d, out, num)
    return out, num


def pool_bwd(dout, num, nmod, t):
    _, b, d2 = dout.shape
    dx = torch.empty(b, nmod * t, d2 // 2, dtype=torch.float32, device=dout.device)
    call("editor_pool_bwd", dout, num, b, nmod, t, d2 // 2, dx)
    return dx


# ---------------------------------------------------------------------------------------------
# contractions
# ---------------------------------------------------------------------------------------------
EPI_NONE, EPI_RESIDUAL, EPI_GELU, EPI_GELU_BWD = 0, 1, 2, 3


EPI_COLSUM = 0x100
EPI_AUX_GRAD = 0x400          # with EPI_GELU / EPI_GELU_BWD (16-bit): aux = gelu'(pre-activation) instead of the pre-activation
EPI_FORCE_PP = 0x200          # run the 256x256 ping-pong kernel whatever the shape heuristic says (tests)
def EPI_STAGGER(c):
    """OR-able (ping-pong kernel, > 256 tiles): the first round's workgroups start spread over c * 2048 shader cycles
    (include/editor_hip.h) - the CUs leave lockstep, HBM-bound epilogues run beside the other CUs' K loops.  Same bits."""
    return (int(c) & 63) << 17


EPI_PIPE128 = 0x800           # prefer the 256x128 three-stage kernel (few token rows; gemm_tile_plan below)
SHORT_TILES = os.environ.get("EDITOR_SHORT_TILES", "1") != "0"      # gemm_tile_rows below (measurement switch)


def EPI_TILE_ROWS(h):
    """OR-able: rows per output tile of the ping-pong kernel (208 | 256), include/editor_hip.h."""
    return ((h // 16) & 15) << 12


def gemm_tile_rows(m, n, cus=256):
    """Tile height for a forward / dgrad product on the ping-pong kernel (k-major operands, staged epilogue): 208 when that
    does not add a round of `cus` workgroups, else 256.  The path's M = 49 536 rows x 768 columns: 582 full tiles = 2.27
    rounds -> 717 short tiles = 2.8 rounds, still three.  Measured (tools/gemm_bench.py): a 208-row tile takes 0.95 of a
    256-row tile's time, not 0.81 - the main loop is bound by its eight barrier intervals per K-tile (~300 cycles each
    whatever the MFMA count of the phase) - so the gain is +4-7 % on the 768-wide products (proj+residual 480 -> 517,
    fc2+residual 912 -> 962 TFLOP/s) and a 240-row tile (N = 3072: 10 rounds either way) LOSES 4-8 %."""
    tn = (n + 255) // 256
    best, best_cost = 256, None
    for h in (256, 208):
        rounds = -(-(-(-m // h) * tn) // cus)
        cost = rounds * (0.77 + 0.23 * h / 256.0)
        if best_cost is None or cost < best_cost * 0.99:
            best, best_cost = h, cost
    return best


def gemm_tile_plan(m, n, cus=256):
    """-> (tile rows of the ping-pong kernel, prefer the 256x128 kernel?) for a forward / dgrad product with k-major operands.
    Rounds of `cus` workgroups x relative tile time: a 208-row tile costs 0.95 of a 256-row one (gemm_tile_rows); a 256x128 tile
    of the three-stage kernel is half the work at ~0.72 of the ping-pong kernel's rate (measured at M = 49 536: 596 vs 828
    TFLOP/s).  The 256x128 kernel wins only when the 256-wide tiles leave most CUs idle: the strong-scaling series' B_local = 16
    (M = 6 192 rows x 768 columns: 90 tiles) - measured there: 1 369 -> 1 421 img/s."""
    th = gemm_tile_rows(m, n, cus)
    tiles_pp = -(-m // th) * ((n + 255) // 256)
    tiles_pipe = -(-m // 256) * ((n + 127) // 128)
    cost_pp = -(-tiles_pp // cus) * (0.95 if th == 208 else 1.0)
    cost_pipe = -(-tiles_pipe // cus) * 0.69
    return th, cost_pipe < 0.97 * cost_pp


def gemm_colsum_ok(m, n, k, c_dtype, trans_a, splitk, m_live, live_dense=False):
    """Can the 16-bit GEMM also deliver the column sums of its (16-bit) output?  (one-pass 256x256 epilogue only)
    live_dense: m_live counts the live prefix of stochastic-depth-compacted rows whose tail rows are ZERO (droppath_plan) - the
    skipped tiles zero their partial rows; the compacted HMA head's live rows (unwritten tails) do not qualify."""
    return (c_dtype in HALF_DTYPES and not trans_a and splitk == 1 and (m_live is None or live_dense) and m >= 2048 and n >= 512
            and n % 8 == 0 and k % 64 == 0)


def gemm(a, b, c, m, n, k, lda, ldb, ldc, trans_a=0, trans_b=0, alpha=1.0, beta=0.0, bias=None, rowscale=None,
         splitk=1, a_off=0, b_off=0, c_off=0, epilogue=0, aux=None, m_live=None, colsum=None, colsum_scale=1.0, tag=None,
         rq=None, live_dense=False, rowmap=None):
    """c = alpha * op(a) op(b) (+bias) (+beta*c) (*rowscale); dtype picks the kernel family.  tag: free-form label
    ("dgrad", ...) for measurement wrappers (bench.py's probe); not used here.
    (fp32 -> exact-f32 MFMA, bf16 -> bf16 MFMA with fp32 accumulate; c may be fp32 for bf16 inputs)."""
    if a.dtype == torch.float32:
        assert b.dtype == torch.float32 and c.dtype == torch.float32 and m_live is None
        if (splitk == 1 and not trans_a and not trans_b and m <= 512 and k >= 1024 and k % 128 == 0 and epilogue == 0
                and rowscale is None and beta == 0.0 and ldc == n and a_off == b_off == c_off == 0):
            # few output tiles, long reduction (REDUCE layers, classifier heads: 128 x 768 x 1536 is 24 workgroups of 96
            # k-steps, 71 us): eight reduction chunks as a BATCH - chunk 0 (+bias) straight into C, chunks 1..7 into
            # workspace slabs - then one fixed-order reduction.  Deterministic, unlike the fp32-atomic split-K.
            s_ = 8
            kc = k // s_
            slabs = workspace(a.device, (s_ - 1) * m * n)
            call("editor_gemm_f32", a, b, c, m, n, kc, lda, ldb, ldc, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, float(alpha), 0.0, bias,
                 None, 1, 0, None, n)
            call("editor_gemm_f32", _ptr(a, kc), _ptr(b, kc), slabs, m, n, kc, lda, ldb, n, 0, 0, s_ - 1, kc, kc, m * n,
                 1, 0, 0, 0, float(alpha), 0.0, None, None, 1, 0, None, n)
            call("editor_reduce_rows", slabs, s_ - 1, m * n, c, 1, 1.0)
            return
        call("editor_gemm_f32", _ptr(a, a_off), _ptr(b, b_off), _ptr(c, c_off), m, n, k, lda, ldb, ldc,
             int(trans_a), int(trans_b), 1, 0, 0, 0, 1, 0, 0, 0, float(alpha), float(beta), bias, rowscale, int(splitk),
             int(epilogue) & 0xFF, aux, n)          # (kernel-selection flags are a 16-bit-family matter)
    elif a.dtype in HALF_DTYPES:
        assert b.dtype == a.dtype
        entry = _h16(a, "gemm")
        if m_live is not None and (m < 256 or n < 128 or k % 64):
            raise RuntimeError("live-row GEMM needs the pipelined path (M >= 256, N >= 128, K % 64 == 0)")
        if trans_a and trans_b and splitk > 1 and k % 64 and k > 64 and m_live is None and beta == 0.0:
            # weight gradient whose reduction length (token rows) is not a multiple of the 64-row K-tile: the pipelined
            # kernel (LDS-DMA, deterministic split-K slabs) takes the whole tiles and a single-split product adds the
            # < 64-row tail - instead of the generic kernel with fp32-atomic split-K for the whole reduction
            k0 = (k // 64) * 64
            gemm(a, b, c, m, n, k0, lda, ldb, ldc, 1, 1, alpha=alpha, splitk=splitk, a_off=a_off, b_off=b_off, c_off=c_off,
                 epilogue=epilogue)
            gemm(a, b, c, m, n, k - k0, lda, ldb, ldc, 1, 1, alpha=alpha, beta=1.0, splitk=1, a_off=a_off + k0 * lda,
                 b_off=b_off + k0 * ldb, c_off=c_off)
            return
        th = ((int(epilogue) >> 12) & 15) * 16 or 256            # explicit EPI_TILE_ROWS, else the shape heuristic
        if (SHORT_TILES and not trans_a and not trans_b and splitk == 1 and beta == 0.0 and (m_live is None or live_dense) and m >= 2048
                and n >= 512 and n % 8 == 0 and ldc % 8 == 0 and k % 64 == 0 and not (int(epilogue) & 0xF000)):
            th, narrow = gemm_tile_plan(m, n)
            if narrow and colsum is None and not (int(epilogue) & EPI_FORCE_PP):
                th, epilogue = 256, int(epilogue) | EPI_PIPE128
            elif th != 256:
                epilogue = int(epilogue) | EPI_TILE_ROWS(th)
        if colsum is not None:
            # colsum (n) fp32 <- column sums of the rounded output (bias gradient of the layer this gradient feeds):
            # per-tile-row partials from the GEMM epilogue, folded in a fixed order
            assert gemm_colsum_ok(m, n, k, c.dtype, trans_a, splitk, m_live, live_dense) and ldc == n and rowmap is None
            tiles_m = (m + th - 1) // th
            part = rq.region(tiles_m * n) if rq is not None else None
            deferred = part is not None
            if not deferred:
                part = workspace(a.device, tiles_m * n)
            call(entry, _ptr(a, a_off), _ptr(b, b_off), _ptr(c, c_off), 0, m, n, k, lda, ldb, ldc,
                 int(trans_a), int(trans_b), float(alpha), float(beta), bias, rowscale, 1, int(epilogue) | EPI_COLSUM, aux,
                 n, part, m_live)
            if deferred:
                rq.add(part, tiles_m, n, colsum, float(colsum_scale))
            else:
                call("editor_reduce_rows", part, tiles_m, n, colsum, 0, float(colsum_scale))
            return
        if rowmap is not None:
            # compacted rows scattered back by the fp32 residual epilogue (stochastic-depth skipping: editor_gemm_h16_rows)
            assert c.dtype == torch.float32 and not trans_a and not trans_b and splitk == 1 and beta == 0.0
            assert a_off == b_off == c_off == 0 and (int(epilogue) & 0xFF) == EPI_RESIDUAL
            call("editor_gemm_h16_rows", _DT_CODE[a.dtype], a, b, c, m, n, k, lda, ldb, ldc, float(alpha), bias, rowscale,
                 int(epilogue), aux, n, m_live, rowmap)
            return
        call(entry, _ptr(a, a_off), _ptr(b, b_off), _ptr(c, c_off), 1 if c.dtype == torch.float32 else 0,
             m, n, k, lda, ldb, ldc, int(trans_a), int(trans_b), float(alpha), float(beta), bias, rowscale, int(splitk),
             int(epilogue), aux, n, workspace(a.device, int(splitk) * m * n) if splitk > 1 else None, m_live)
    else:
        raise TypeError(a.dtype)


# ---------------------------------------------------------------------------------------------
# split-precision forward (COMPUTE_DTYPE 'f16x2'): operands travel as pairs of half tensors x = hi + lo
# ---------------------------------------------------------------------------------------------
SPLIT_WSCALE = 256.0          # power of two folded into the weight pairs (|w| ~ 0.02: keeps the low-order halves normal); the
                              # products' alpha divides it out


def split_f32(x, scale=1.0):
    """fp32 tensor -> (hi, lo) half tensors of the same shape with hi + lo = x * scale to 2^-22."""
    x = x.contiguous()
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    call("editor_split_f32", x, hi, lo, x.numel(), float(scale))
    return hi, lo


def layernorm_fwd_split(x2d, gamma, beta, eps, rowmask=None, m_live=None):
    m, d = x2d.shape
    hi = torch.empty(m, d, dtype=torch.float16, device=x2d.device)
    lo = torch.empty(m, d, dtype=torch.float16, device=x2d.device)
    mean = torch.empty(m, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2d.device)
    call("editor_layernorm_fwd_f16x2", x2d, gamma, beta, float(eps), m, d, rowmask, 0, hi, lo, mean, rstd, m_live)
    return hi, lo, mean, rstd


def layernorm_fwd_split_perm(x2d, gamma, beta, eps, perm, rowscale, copy_out):
    """layernorm_fwd_perm as the half pair of the split-precision forward -> hi, lo, mean, rstd."""
    m, d = x2d.shape
    hi = torch.empty(m, d, dtype=torch.float16, device=x2d.device)
    lo = torch.empty(m, d, dtype=torch.float16, device=x2d.device)
    mean = torch.empty(m, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2d.device)
    call("editor_layernorm_fwd_perm_f16x2", x2d, gamma, beta, float(eps), m, d, hi, lo, mean, rstd, perm, rowscale, copy_out)
    return hi, lo, mean, rstd


def im2col16_split(img):
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    b, c, h, w = imgs[0].shape
    rows = b * (h // 16) * (w // 16)
    hi = torch.empty(len(imgs) * rows, c * 256, dtype=torch.float16, device=imgs[0].device)
    lo = torch.empty_like(hi)
    for i, im in enumerate(imgs):
        if tuple(im.shape) != (b, c, h, w):
            raise ValueError("im2col16: modality tensors of different shapes")
        call("editor_im2col16_f16x2", im, b, c, h, w, _ptr(hi, i * rows * c * 256), _ptr(lo, i * rows * c * 256))
    return hi, lo


def gemm_split(a, b, c, c_lo, m, n, k, alpha=1.0, bias=None, rowscale=None, epilogue=0, aux=None, m_live=None, live_dense=False,
               rowmap=None):
    """c (fp32, c_lo None) or (c, c_lo) half pair = alpha * (a_hi + a_lo)(b_hi + b_lo)^T (+bias) (*rowscale) (+epilogue);
    a = (hi, lo) (M,K), b = (hi, lo) (N,K), all contiguous.  live_dense / rowmap: as ops.gemm (stochastic-depth-compacted rows)."""
    if SHORT_TILES and m >= 2048 and n >= 512 and (m_live is None or live_dense) and not (int(epilogue) & 0xF000):
        th = gemm_tile_rows(m, n)
        if th != 256:
            epilogue = int(epilogue) | EPI_TILE_ROWS(th)
    if rowmap is not None:
        assert c.dtype == torch.float32 and c_lo is None and (int(epilogue) & 0xFF) == EPI_RESIDUAL
        call("editor_gemm_f16x2_rows", a[0], a[1], b[0], b[1], c, m, n, k, k, k, n, float(alpha), bias, rowscale, int(epilogue), aux,
             n, m_live, rowmap)
        return
    call("editor_gemm_f16x2", a[0], a[1], b[0], b[1], c, c_lo, 1 if c.dtype == torch.float32 else 0, m, n, k, k, k, n,
         float(alpha), bias, rowscale, int(epilogue), aux, n, m_live)


def attention_fwd_split(qkv, b, t, heads, hd, mask=None, probs=None, cu=None, scale=None):
    """qkv = (hi, lo) half pair (rows, 3*heads*hd) -> (out_hi, out_lo), lse."""
    hi, lo = qkv
    d = heads * hd
    rows = hi.shape[0]
    scale = float(scale or hd ** -0.5)
    if hd not in ATTN_HEAD_WIDTHS:                       # (see attention_fwd) the pair is summed back into fp32 - exact - first
        if cu is not None:
            raise RuntimeError("variable-length attention needs 32 / 64 / 96-wide heads (use the dense-masked HMA form)")
        out32, probs = attention_fwd(hi.float() + lo.float(), b, t, heads, hd, mask, probs, cu=None, scale=scale)
        return split_f32(out32), probs
    out_hi = _packed_alloc(rows, d, torch.float16, hi.device, cu)
    out_lo = _packed_alloc(rows, d, torch.float16, hi.device, cu)
    lse = torch.empty(heads * rows, dtype=torch.float32, device=hi.device)
    call("editor_attention_fwd_f16x2", hi, lo, b, t, heads, hd, scale, mask, out_hi, out_lo, probs,
         0 if probs is None else probs.shape[-1], lse, cu, rows)
    return (out_hi, out_lo), lse


def _packed_alloc(rows, cols, dtype, device, cu):
    """Output of a variable-length kernel: only the rows inside sequences are written, so the pad rows between the live
    extent cu[-1] and the next multiple of 64 are zeroed (the live-row reductions read whole 64-row tiles); rows beyond
    that are never read by anyone (m_live contract of the GEMM / LayerNorm kernels) and stay uninitialised."""
    out = torch.empty(rows, cols, dtype=dtype, device=device)
    if cu is not None:
        call("editor_zero_tail_rows", out, cols * out.element_size(), rows, cu[-1:])
    return out


def wgrad_group_split(tiles, ktiles, cus=256):
    """Reduction split of a grouped weight-gradient launch: tiles x split workgroups should fill whole rounds of the CUs
    (108 tiles of a ViT-B block x 7 = 756 = 2.95 rounds; ViT-L: 192 x 4 = 768 = 3.0) with >= 24 K-tiles per workgroup."""
    best, best_eff = 1, 0.0
    for rounds in (1, 2, 3, 4):
        s_ = (cus * rounds) // tiles
        if s_ < 1 or ktiles // s_ < 24:
            continue
        eff = tiles * s_ / float(cus * rounds)
        if eff > best_eff + 0.02:
            best, best_eff = s_, eff
    return best


GROUP_FWD = os.environ.get("EDITOR_GROUP_FWD", "1") != "0"     # A/B switch: the HMA modality blocks' products as grouped launches


def gemm_group_ok(reqs):
    """Can these ops.gemm requests (args tuples + keyword dicts of IDENTICAL shape) leave as ONE editor_gemm_group launch?  Products of
    the 256 x 256 ping-pong kernel only: 16-bit k-major operands, full 256-wide column tiles, no split-K / colsum / offsets, and the
    tile plan of ops.gemm must be the full-tile ping-pong kernel for them (the compacted HMA head's live-row products)."""
    if not GROUP_FWD or len(reqs) < 2 or len(reqs) > 4:
        return False
    a0, kw0 = reqs[0]
    (a, b, c, m, n, k, lda, ldb, ldc), ta, tb = a0[:9], a0[9] if len(a0) > 9 else 0, a0[10] if len(a0) > 10 else 0
    if a.dtype not in HALF_DTYPES or ta or tb or m < 2048 or n < 256 or n % 256 or k % 64 or kw0.get("m_live") is None:
        return False
    if lda != k or ldb != k or ldc != n:
        return False
    keys = ("alpha", "beta", "splitk", "a_off", "b_off", "c_off", "epilogue", "colsum")      # (rq only matters with colsum)
    for args, kw in reqs:
        if tuple(args[3:]) != tuple(a0[3:]) or args[0].dtype != a.dtype or args[1].dtype != a.dtype or args[2].dtype != c.dtype:
            return False
        if any(kw.get(q) != kw0.get(q) for q in keys) or kw.get("m_live") is not kw0.get("m_live"):
            return False
        if kw.get("beta", 0.0) != 0.0 or kw.get("splitk", 1) != 1 or kw.get("colsum") is not None:
            return False
        if kw.get("a_off", 0) or kw.get("b_off", 0) or kw.get("c_off", 0) or (int(kw.get("epilogue", 0)) & (0xF000 | EPI_PIPE128 | EPI_COLSUM)):
            return False
        if (kw.get("bias") is None) != (kw0.get("bias") is None) or (kw.get("rowscale") is None) != (kw0.get("rowscale") is None) \
                or (kw.get("aux") is None) != (kw0.get("aux") is None):
            return False
        if not (args[0].is_contiguous() and args[1].is_contiguous() and args[2].is_contiguous()):
            return False
    return True


def gemm_group(reqs):
    """The requests of gemm_group_ok as one launch (editor_gemm_group): bit-identical to the separate ops.gemm calls."""
    import ctypes
    cnt = len(reqs)
    a0, kw0 = reqs[0]
    a, b, c, m, n, k = a0[:6]
    arr = lambda ts: None if ts[0] is None else (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(a.device):
        call("editor_gemm_group", _DT_CODE[a.dtype], cnt, arr([r[0][0] for r in reqs]), arr([r[0][1] for r in reqs]),
             arr([r[0][2] for r in reqs]), 1 if c.dtype == torch.float32 else 0, m, n, k, k, k, n, float(kw0.get("alpha", 1.0)),
             arr([r[1].get("bias") for r in reqs]), arr([r[1].get("rowscale") for r in reqs]),
             int(kw0.get("epilogue", 0)) & ~EPI_FORCE_PP, arr([r[1].get("aux") for r in reqs]), n, kw0.get("m_live"))


def gemm_wgrad_group(jobs, m, alpha=1.0, m_live=None):
    """jobs: list of (dy (m, n_i), x (m, k_i), dw (n_i, k_i) fp32[, live_i]) of ONE block -> one launch (editor_gemm_wgrad_group).
    live_i (optional 4th entry): device scalar, live token rows of THAT problem (rows beyond are zero) - a block whose MLP branch
    ran on stochastic-depth-compacted rows next to its dense attention branch (editor_gemm_wgrad_group_live)."""
    import ctypes
    cnt = len(jobs)
    lives = [j[3] if len(j) > 3 else None for j in jobs]
    jobs = [j[:3] for j in jobs]
    dt = _DT_CODE[jobs[0][0].dtype]
    ns = [j[0].shape[1] for j in jobs]
    ks = [j[1].shape[1] for j in jobs]
    tiles = sum((n // 256) * (k // 256) for n, k in zip(ns, ks))
    sk = wgrad_group_split(tiles, m // 64)
    ws = workspace(jobs[0][0].device, sk * sum(n * k for n, k in zip(ns, ks)))
    arr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in ts])
    ia = lambda v: (ctypes.c_int * cnt)(*v)
    for dy, x, dw in jobs:
        assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous() and dy.shape[0] == m == x.shape[0]
    if any(l_ is not None for l_ in lives):
        assert m_live is None
        la = (ctypes.c_void_p * cnt)(*[None if l_ is None else l_.data_ptr() for l_ in lives])
        call("editor_gemm_wgrad_group_live", dt, cnt, arr([j[0] for j in jobs]), arr([j[1] for j in jobs]), arr([j[2] for j in jobs]),
             ia(ns), ia(ks), m, float(alpha), sk, ws, la)
        return
    call("editor_gemm_wgrad_group", dt, cnt, arr([j[0] for j in jobs]), arr([j[1] for j in jobs]), arr([j[2] for j in jobs]),
         ia(ns), ia(ks), m, float(alpha), sk, ws, m_live)


WGRAD_LN_CUS = int(os.environ.get("EDITOR_WGRAD_LN_CUS", "64"))      # workgroups (= CUs) of the memory role, a multiple of 8


def gemm_wgrad_group_ln(jobs, m, alpha, dy, x2d, gamma, mean, rstd, dx_in, rowscale, scale=1.0, dy_scale=1.0, dgb_out=None,
                        want_colsum=True, cs_out=None, rq=None, nmem=None):
    """gemm_wgrad_group(jobs, m, alpha) AND layernorm_bwd_cast(dy, x2d, gamma, mean, rstd, dx_in, rowscale, scale, dy_scale, ...) in ONE
    launch: the LayerNorm backward as a memory-bound role on `nmem` CUs beside the weight-gradient tiles on the others
    (include/editor_hip.h: editor_gemm_wgrad_group_ln).  -> dx, dgamma, dbeta, dx16, colsum(dx16) / scale (None unless want_colsum)."""
    cnt = len(jobs)
    nmem = int(nmem or WGRAD_LN_CUS)
    dt = _DT_CODE[jobs[0][0].dtype]
    ns = [j[0].shape[1] for j in jobs]
    ks = [j[1].shape[1] for j in jobs]
    tiles = sum((n // 256) * (k // 256) for n, k in zip(ns, ks))
    sk = wgrad_group_split(tiles, m // 64)
    ws = workspace(jobs[0][0].device, sk * sum(n * k for n, k in zip(ns, ks)))
    arr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in ts])
    ia = lambda v: (ctypes.c_int * cnt)(*v)
    assert all(len(j) == 3 for j in jobs)                    # (no per-problem live counts in the role form)
    for jdy, jx, jdw in jobs:
        assert jdy.is_contiguous() and jx.is_contiguous() and jdw.is_contiguous() and jdy.shape[0] == m == jx.shape[0]
    rows, d = x2d.shape
    assert dy.dtype == jobs[0][0].dtype and dy.shape == x2d.shape and d in (768, 1024)
    dev = x2d.device
    dx = torch.empty(rows, d, dtype=torch.float32, device=dev)
    dx16 = torch.empty(rows, d, dtype=dy.dtype, device=dev)
    dgb = dgb_out if dgb_out is not None else torch.empty(2, d, dtype=torch.float32, device=dev)
    cs = None
    if want_colsum:
        cs = cs_out if cs_out is not None else torch.empty(d, dtype=torch.float32, device=dev)
    parts = rq.region(nmem * 3 * d) if rq is not None else None
    queued = parts is not None
    if not queued:
        parts = torch.empty(nmem * 3 * d, dtype=torch.float32, device=dev)
    cparts = parts[nmem * 2 * d:]
    call("editor_gemm_wgrad_group_ln", dt, cnt, arr([j[0] for j in jobs]), arr([j[1] for j in jobs]), arr([j[2] for j in jobs]),
         ia(ns), ia(ks), m, float(alpha), sk, ws,
         dy, float(dy_scale), x2d, gamma, mean, rstd, rows, d, dx_in, dx, parts, dx16, rowscale, float(scale),
         cparts if want_colsum else None, nmem)
    if queued:
        rq.add(parts, nmem, 2 * d, dgb, 1.0)
        if want_colsum:
            rq.add(cparts, nmem, d, cs, 1.0 / float(scale))
    else:
        call("editor_reduce_rows", parts, nmem, 2 * d, dgb, 0, 1.0)
        if want_colsum:
            call("editor_reduce_rows", cparts, nmem, d, cs, 0, 1.0 / float(scale))
    return dx, dgb[0], dgb[1], dx16, cs


# head widths the fused 16-bit attention / rollout kernels are built for (csrc/attention_bf16.hip compiled per width, round 4:
# 64 = ViT-B/L, DeiT-B; 96 = ViT-small's backbone; 32 = DeiT-small's HMA heads)
ATTN_HEAD_WIDTHS = (32, 64, 96)


def attention_fwd(qkv, b, t, heads, hd, mask=None, probs=None, want_lse=True, cu=None, scale=None):
    """Attention / AttentionMask core on packed qkv (rows, 3*heads*hd) -> (rows, heads*hd).
    Dense: rows = b*t.  Variable length (compacted HMA): cu (b+1 int32) = packed row range of every sequence,
    t = longest sequence; rows outside every sequence (padding) come out as zeros.
    Returns (out, saved): `saved` is what the backward needs besides qkv/out - the probabilities in fp32 mode,
    the per-row log-sum-exp in bf16 mode."""
    d = heads * hd
    rows = qkv.shape[0]
    scale = float(scale or hd ** -0.5)          # qk_scale of the architecture (vit_pytorch.py:176), default head_dim ** -0.5
    if qkv.dtype == torch.float32:
        if cu is not None:
            raise RuntimeError("variable-length attention exists in the bf16 kernels only (f32 = dense parity mode)")
        out = torch.empty(rows, d, dtype=qkv.dtype, device=qkv.device)
        if probs is None:
            probs = torch.empty(b, heads, t, t, dtype=torch.float32, device=qkv.device)
        call("editor_attention_fwd_f32", qkv, b, t, heads, hd, scale, mask, out, probs)
        return out, probs
    if hd not in ATTN_HEAD_WIDTHS:
        # head widths the fused 16-bit kernels are not built for (they exist for 32, 64 and 96: every factory of the reference,
        # vit_pytorch.py:693-727): this product runs on the exact-f32 attention kernels between two casts - correct, not fast.
        # `saved` is then the fp32 probability tensor (4-D), which attention_bwd recognises.
        if cu is not None:
            raise RuntimeError("variable-length attention needs 32 / 64 / 96-wide heads (use the dense-masked HMA form)")
        if probs is not None and probs.shape[-1] != t:
            raise RuntimeError("attention_fwd: probability rows of such a head must be unpadded (ldp == T)")
        out32, probs = attention_fwd(qkv.float(), b, t, heads, hd, mask, probs, cu=None, scale=scale)
        return out32.to(qkv.dtype), probs
    out = _packed_alloc(rows, d, qkv.dtype, qkv.device, cu)
    lse = torch.empty(heads * rows, dtype=torch.float32, device=qkv.device) if want_lse else None
    call(_h16(qkv, "attention_fwd"), qkv, b, t, heads, hd, scale, mask, out, probs,
         0 if probs is None else probs.shape[-1], lse, cu, rows)
    return out, lse




ATTN_BWD_COLSUM = os.environ.get("EDITOR_ATTN_COLSUM", "1") == "1"


def attention_bwd_colsum_ok(qkv, t, hd, saved=None):
    """True if attention_bwd can deliver colsum(dqkv) - the qkv bias gradient - from its own accumulators (the fused 16-bit two-pass
    kernels: sequences of <= 608 tokens, 96-wide heads <= 160), include/editor_hip.h: editor_attention_bwd_colsum_*."""
    return (ATTN_BWD_COLSUM and qkv.dtype in HALF_DTYPES and hd in ATTN_HEAD_WIDTHS and not (saved is not None and saved.dim() == 4)
            and t <= (608 if hd <= 64 else 160))


def attention_bwd(qkv, dout, b, t, heads, hd, mask=None, saved=None, out=None, cu=None, scale=None, colsum=None, colsum_scale=1.0,
                  rq=None):
    """-> dqkv.  colsum (fp32 vector of 3*heads*hd, only when attention_bwd_colsum_ok): also filled with colsum(dqkv) * colsum_scale -
    one partial row per sequence from the kernels, folded here or, with a ReduceQueue, at the end of the block."""
    scale = float(scale or hd ** -0.5)
    rows = qkv.shape[0]
    if colsum is not None and not attention_bwd_colsum_ok(qkv, t, hd, saved):
        raise ValueError("attention_bwd: column sums are not available for this call (attention_bwd_colsum_ok)")
    if qkv.dtype == torch.float32:
        dqkv = torch.empty_like(qkv)
        ws = torch.empty(b, heads, t, t, dtype=torch.float32, device=qkv.device)
        call("editor_attention_bwd_f32", qkv, dout, saved, b, t, heads, hd, scale, dqkv, ws)
    elif hd not in ATTN_HEAD_WIDTHS or (saved is not None and saved.dim() == 4):
        # (see attention_fwd: exact-f32 kernels between two casts; a 4-D `saved` is the fp32 probability tensor such a forward -
        #  or the split-precision forward of a non-64-wide head, attention_fwd_split - handed back)
        dqkv = attention_bwd(qkv.float(), dout.float(), b, t, heads, hd, mask, saved, None, None, scale).to(qkv.dtype)
    else:
        dqkv = _packed_alloc(rows, qkv.shape[1], qkv.dtype, qkv.device, cu)
        ws = torch.empty(heads * rows, dtype=torch.float32, device=qkv.device)
        if colsum is None:
            call(_h16(qkv, "attention_bwd"), qkv, dout, out, saved, b, t, heads, hd, scale, mask, dqkv, ws, cu, rows)
        else:
            ncol = qkv.shape[1]
            parts = rq.region(b * ncol) if rq is not None else None
            queued = parts is not None
            if not queued:
                parts = torch.empty(b * ncol, dtype=torch.float32, device=qkv.device)
            call(_h16(qkv, "attention_bwd_colsum"), qkv, dout, out, saved, b, t, heads, hd, scale, mask, dqkv, ws, cu, rows, parts)
            if queued:
                rq.add(parts, b, ncol, colsum, float(colsum_scale))
            else:
                call("editor_reduce_rows", parts, b, ncol, colsum, 0, float(colsum_scale))
    return dqkv


# ---------------------------------------------------------------------------------------------
# compacted (variable-length) HMA
# ---------------------------------------------------------------------------------------------
class CompactPlan:
    """Packing plan of the HMA head for one batch, built ON DEVICE from the SFTS index with no host round trip:
    buffers and launches are sized for the worst case (every token kept) and the live row counts stay in device
    memory (`live_a` = sum_b L_b, `live_b` = nmod * that), which the kernels read through their `m_live` argument."""

    def __init__(self, index, t, nmod=3, pad=64):
        b, n = index.shape
        dev = index.device
        self.b, self.t, self.nmod = b, t, nmod
        self.cu = torch.empty(b + 1, dtype=torch.int32, device=dev)
        tok = torch.empty(b * (n + 1), dtype=torch.int32, device=dev)
        call("editor_compact_plan", index, b, n, self.cu, tok)
        self.ma = (b * t + pad - 1) // pad * pad                   # worst-case rows per modality (layout A)
        self.mb = (nmod * b * t + pad - 1) // pad * pad            # worst-case rows of layout B
        self.map_a = torch.empty(nmod * self.ma, dtype=torch.int32, device=dev)
        self.map_b = torch.empty(self.mb, dtype=torch.int32, device=dev)
        self.map_cls = torch.empty(nmod * b, dtype=torch.int32, device=dev)
        self.mask_a = torch.empty(self.ma, dtype=torch.uint8, device=dev)
        self.mask_b = torch.empty(self.mb, dtype=torch.uint8, device=dev)
        self.cu3 = torch.empty(b + 1, dtype=torch.int32, device=dev)
        call("editor_compact_maps", self.cu, tok, b, t, nmod, self.ma, self.mb, self.map_a, self.map_b, self.map_cls,
             self.mask_a, self.mask_b, self.cu3)
        self.live_a = self.cu[b:b + 1]                             # device scalars
        self.live_b = self.cu3[b:b + 1]

    @property
    def total(self):
        """Host copy of the kept-row count (synchronises; for tests / logging only)."""
        return int(self.cu[-1].item())


def gather_rows(x2d, src, live=None, live_mul=1, live_stride=0):
    """out[r] = x2d[src[r]] (0 where src[r] < 0).  live (device int32 scalar): only rows below
    roundup64(live_mul * live) of every `live_stride`-row segment are produced."""
    r = src.numel()
    out = torch.empty(r, x2d.shape[1], dtype=torch.float32, device=x2d.device)
    call("editor_gather_rows", x2d, src, r, x2d.shape[1], out, live, int(live_mul), int(live_stride))
    return out


# debug: the row-movement backwards below leave rows NO consumer reads unwritten (fill "none" / "tail", the live-row pool
# backward) - correct only while that contract holds.  EDITOR_POISON_UNWRITTEN=1 (or ops.POISON_UNWRITTEN = True) fills those buffers
# with NaN first, so a consumer that does read an unwritten row shows up as a non-finite / different gradient
# (tests/test_gpu_model.py::test_unwritten_gradient_rows_are_never_read).
POISON_UNWRITTEN = os.environ.get("EDITOR_POISON_UNWRITTEN", "0") == "1"


def _unwritten(rows, d, device):
    if POISON_UNWRITTEN:
        return torch.full((rows, d), float("nan"), dtype=torch.float32, device=device)
    return torch.empty(rows, d, dtype=torch.float32, device=device)


def scatter_rows(dy2d, src, rows_out, fill="all", live=None, seg_rows=0):
    """dx[src[r]] = dy[r].  fill: "all" = dx zero-filled first (rows no index names are 0); "none" = no fill at all - ONLY for a
    consumer that never reads those rows; "tail" = only the pad rows [live, roundup64(live)) of every `seg_rows`-row segment are
    zeroed (live: device int32 scalar) - what the live-row kernels read beyond the live extent."""
    d = dy2d.shape[1]
    if fill == "all":
        dx = torch.empty(rows_out, d, dtype=torch.float32, device=dy2d.device)
        call("editor_scatter_rows", dy2d, src, src.numel(), d, rows_out, dx)
        return dx
    dx = _unwritten(rows_out, d, dy2d.device)
    call("editor_scatter_rows_nofill", dy2d, src, src.numel(), d, dx)
    if fill == "tail":
        seg = int(seg_rows) or rows_out
        for s0 in range(0, rows_out, seg):
            call("editor_zero_tail_rows", dx[s0:s0 + seg], d * 4, min(seg, rows_out - s0), live)
    return dx


def pool_packed_fwd(x2d, cu, b, nmod):
    d = x2d.shape[1]
    out = torch.empty(nmod, b, 2 * d, dtype=torch.float32, device=x2d.device)
    num = torch.empty(b, dtype=torch.float32, device=x2d.device)
    call("editor_pool_packed_fwd", x2d, cu, b, nmod, d, out, num)
    return out, num


def pool_packed_bwd(dout, num, cu, b, nmod, rows, live=None):
    if live is not None:                      # live: device scalar = nmod * cu[b] live rows; no 152 MB zero fill
        dx = _unwritten(rows, dout.shape[-1] // 2, dout.device)
        call("editor_pool_packed_bwd_nofill", dout, num, cu, b, nmod, dout.shape[-1] // 2, dx)
        call("editor_zero_tail_rows", dx, dx.shape[1] * 4, rows, live)
        return dx
    return _pool_packed_bwd_filled(dout, num, cu, b, nmod, rows)


def _pool_packed_bwd_filled(dout, num, cu, b, nmod, rows):
    d = dout.shape[2] // 2
    dx = torch.empty(rows, d, dtype=torch.float32, device=dout.device)
    call("editor_pool_packed_bwd", dout, num, cu, b, nmod, d, rows, dx)
    return dx


# ---------------------------------------------------------------------------------------------
# head kernels
# ---------------------------------------------------------------------------------------------
def _rows_view(x):
    """(B, C) view with unit column stride -> (tensor, row stride)."""
    assert x.dim() == 2 and x.stride(1) == 1, "head kernels need unit column stride"
    return x, x.stride(0)


def bn1d_fwd(x, gamma, beta, rmean, rvar, momentum, eps, training):
    x, ldx = _rows_view(x)
    b, c = x.shape
    y = torch.empty(b, c, dtype=torch.float32, device=x.device)
    sm = torch.empty(c, dtype=torch.float32, device=x.device) if training else None
    si = torch.empty(c, dtype=torch.float32, device=x.device) if training else None
    call("editor_bn1d_fwd", _ptr(x), ldx, b, c, gamma, beta, rmean, rvar, float(momentum), float(eps),
         1 if training else 0, y, sm, si)
    return y, sm, si


def bn1d_bwd(dy, x, gamma, sm, si):
    x, ldx = _rows_view(x)
    b, c = x.shape
    dx = torch.empty(b, c, dtype=torch.float32, device=x.device)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    call("editor_bn1d_bwd", dy, _ptr(x), ldx, b, c, gamma, sm, si, dx, dg, db)
    return dx, dg, db


def ocfr_fwd(feat, label, centers, momentum, loss, accumulate):
    feat, ldf = _rows_view(feat)
    b, d = feat.shape
    fn = torch.empty(b, d, dtype=torch.float32, device=feat.device)
    inv = torch.empty(b, dtype=torch.float32, device=feat.device)
    ws = workspace(feat.device, b)
    call("editor_ocfr_fwd", _ptr(feat), ldf, label, b, d, centers.shape[0], centers, float(momentum), fn, inv, ws, loss,
         1 if accumulate else 0)
    return fn, inv


def ocfr_bwd(fn, inv, centers, label, dloss):
    b, d = fn.shape
    df = torch.empty(b, d, dtype=torch.float32, device=fn.device)
    call("editor_ocfr_bwd", fn, inv, centers, label, dloss, b, d, df)
    return df


def center_loss_fwd(x, centers, label):
    """-> loss (1) fp32, dist (B) (own-class squared distances, saved for the backward)."""
    b, d = x.shape
    dist = torch.empty(b, dtype=torch.float32, device=x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    call("editor_center_loss_fwd", x, centers, label, b, centers.shape[0], d, dist, workspace(x.device, b), loss)
    return loss, dist


def center_loss_bwd(x, centers, label, dist, dloss, want_x=True, want_c=True):
    b, d = x.shape
    dx = torch.empty_like(x) if want_x else None
    dc = torch.empty_like(centers) if want_c else None
    call("editor_center_loss_bwd", x, centers, label, dist, dloss, b, centers.shape[0], d, dx, dc)
    return dx, dc


def ce_smooth_fwd(logits, target, eps, loss, accumulate):
    b, c = logits.shape
    call("editor_ce_smooth_fwd", logits, target, b, c, float(eps), workspace(logits.device, b), loss,
         1 if accumulate else 0)


def ce_smooth_bwd(logits, target, eps, dloss):
    b, c = logits.shape
    d = torch.empty_like(logits)
    call("editor_ce_smooth_bwd", logits, target, b, c, float(eps), dloss, d)
    return d


def triplet_fwd(feat, label, loss, accumulate):
    """Returns (idx (2B) int32, coef (3B) fp32) saved for backward."""
    feat, ldf = _rows_view(feat)
    b, d = feat.shape
    dev = feat.device
    gram = torch.empty(9, b, b, dtype=torch.float32, device=dev)       # Gram matrix + 8 reduction-chunk slabs
    sq = torch.empty(b, dtype=torch.float32, device=dev)
    idx = torch.empty(2 * b, dtype=torch.int32, device=dev)
    coef = torch.empty(3 * b, dtype=torch.float32, device=dev)
    call("editor_triplet_fwd", _ptr(feat), ldf, label, b, d, gram, sq, idx, coef, workspace(dev, b), loss,
         1 if accumulate else 0)
    return idx, coef


def triplet_bwd(feat, idx, coef, dloss):
    feat, ldf = _rows_view(feat)
    b, d = feat.shape
    df = torch.empty(b, d, dtype=torch.float32, device=feat.device)
    call("editor_triplet_bwd", _ptr(feat), ldf, b, d, idx, coef, dloss, df)
    return df


def droppath_scales_dev(rates, b, t, state):
    """droppath_scales keyed by the int64 device scalar `state`, which the call advances (hipGraph-replay safe)."""
    l = rates.numel()
    out = torch.empty(l, 2, b * t, dtype=torch.float32, device=rates.device)
    call("editor_droppath_scales_dev", rates, l, b, t, state, out)
    return out


def droppath_plan(scales, l, b, t):
    """Stochastic-depth compaction plan of a scales tensor (l, 2, b*t): -> perm (l,2,b*t) int32 (token row -> slot, live samples
    first), inv (l,2,b*t) (slot -> token row), live (l,2) int32 (live rows).  include/editor_hip.h: editor_droppath_plan."""
    perm = torch.empty(l, 2, b * t, dtype=torch.int32, device=scales.device)
    inv = torch.empty_like(perm)
    live = torch.empty(l, 2, dtype=torch.int32, device=scales.device)
    call("editor_droppath_plan", scales, l, b, t, perm, inv, live)
    return perm, inv, live


def droppath_scales(rates, b, t, seed):
    """(L,2,b*t) fp32 per-row drop-path scales keep/keep_prob (vit_pytorch.py:52-69) for every block and branch."""
    l = rates.numel()
    out = torch.empty(l, 2, b * t, dtype=torch.float32, device=rates.device)
    call("editor_droppath_scales", rates, l, b, t, int(seed), out)
    return out

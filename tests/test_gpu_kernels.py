"""Unit parity of the HIP row / contraction kernels against plain PyTorch fp32 on CPU
(per-kernel numerics; the end-to-end parity against the oracle is tests/test_gpu_model.py)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def _t(dtype, f32, b16):
    """tolerance per activation dtype: f16 carries 3 more mantissa bits than bf16 (asserted at 1/6 of the bf16 bound)"""
    return f32 if dtype == torch.float32 else (b16 if dtype == torch.bfloat16 else b16 / 6)


@pytest.fixture(scope="module")
def ops():
    from editor_amd import ops
    return ops


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [768, 256, 1024, 384, 196])        # 384 (DeiT-small), 196: widths that are not multiples of 256
def test_layernorm_fwd_bwd(ops, dtype, d):
    m = 517
    x = torch.randn(m, d, generator=_g(1)) * 2 + 0.3
    w = torch.rand(d, generator=_g(2)) + 0.5
    b = torch.randn(d, generator=_g(3)) * 0.1
    mask = (torch.rand(m, generator=_g(4)) > 0.4).to(torch.uint8)
    for use_mask in (False, True):
        xr = x.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y_ref = F.layer_norm(xr, (d,), wr, br, 1e-6)
        if use_mask:
            y_ref = y_ref * mask.view(-1, 1).float()
        dy = torch.randn(m, d, generator=_g(5))
        dyq = dy.to(dtype).float()
        y_ref.backward(dyq)
        mk = mask.cuda() if use_mask else None
        y, mean, rstd = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-6, dtype, mk, 0)
        tol = _t(dtype, 1e-5, 6e-3)
        assert rel_err(y.float().cpu(), y_ref.detach()) < tol
        res = torch.randn(m, d, generator=_g(6))
        dx, dg, db = ops.layernorm_bwd(dy.to(dtype).cuda(), x.cuda(), w.cuda(), mean, rstd, mk, 0, dx_in=res.cuda())
        assert rel_err(dx.cpu(), xr.grad + res) < 2e-5
        assert rel_err(dg.cpu(), wr.grad) < 2e-5
        assert rel_err(db.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,d", [(5161, 768), (517, 1024), (3, 256)])
def test_layernorm_bwd_cast_fused(ops, m, d, dtype):
    """editor_layernorm_bwd_cast = editor_layernorm_bwd followed by editor_cast_rows_colsum on its dx: the same bits for dx,
    the 16-bit copy and the LN parameter gradients; the column sums (different per-block grouping) to fp32 rounding."""
    x = (torch.randn(m, d, generator=_g(1)) * 2 + 0.3).cuda()
    w = (torch.rand(d, generator=_g(2)) + 0.5).cuda()
    dy = torch.randn(m, d, generator=_g(5)).to(dtype).cuda()
    res = torch.randn(m, d, generator=_g(6)).cuda()
    rs = (torch.rand(m, generator=_g(7)) + 0.5).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, w, torch.zeros_like(w), 1e-6, dtype, None, 0)
    for rowscale, scale in ((rs, 1.0), (None, 4.0), (rs, 1.0 / 8)):
        dx0, dg0, db0 = ops.layernorm_bwd(dy, x, w, mean, rstd, None, 0, dx_in=res, dy_scale=0.5)
        c0, cs0 = ops.cast_rows_colsum(dx0, rowscale, dtype, scale)
        dx1, dg1, db1, c1, cs1 = ops.layernorm_bwd_cast(dy, x, w, mean, rstd, res, rowscale, scale, dy_scale=0.5)
        for a, b in ((dx0, dx1), (dg0, dg1), (db0, db1), (c0, c1)):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
        assert rel_err(cs1.cpu(), cs0.cpu().double()) < 1e-5
        assert rel_err(cs1.cpu(), c1.double().sum(0).cpu() / scale) < 1e-5
        _, _, _, c2, cs2 = ops.layernorm_bwd_cast(dy, x, w, mean, rstd, res, rowscale, scale, dy_scale=0.5, want_colsum=False)
        assert cs2 is None and torch.equal(c2.view(torch.uint8), c1.view(torch.uint8))


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_f32_layouts(ops, ta, tb):
    m, n, k = 197, 171, 203
    a = torch.randn((k, m) if ta else (m, k), generator=_g(1))
    b = torch.randn((k, n) if tb else (n, k), generator=_g(2))
    bias = torch.randn(n, generator=_g(3))
    c0 = torch.randn(m, n, generator=_g(4))
    ref = 0.7 * ((a.t() if ta else a).double() @ (b if tb else b.t()).double()) + bias.double() + 0.5 * c0.double()
    c = c0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, a.shape[1], b.shape[1], n, ta, tb, alpha=0.7, beta=0.5, bias=bias.cuda())
    assert rel_err(c.cpu(), ref) < 1e-6
    # split-K with atomics
    c = torch.empty(m, n, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, a.shape[1], b.shape[1], n, ta, tb, splitk=5)
    assert rel_err(c.cpu(), (a.t() if ta else a).double() @ (b if tb else b.t()).double()) < 1e-6


def test_gemm_f32_rowscale(ops):
    m, n, k = 130, 64, 96
    a, b = torch.randn(m, k, generator=_g(1)), torch.randn(n, k, generator=_g(2))
    rs = torch.rand(m, generator=_g(3))
    c0 = torch.randn(m, n, generator=_g(4))
    c = c0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, k, k, n, 0, 0, beta=1.0, rowscale=rs.cuda())
    # C = rowscale * (A B^T) + 1.0 * C   (drop-path scaled branch added to the residual)
    assert rel_err(c.cpu(), rs.view(-1, 1) * (a @ b.t()) + c0) < 1e-6


def _attn_ref(qkv, b, t, heads, hd, mask):
    d = heads * hd
    q, k, v = qkv.view(b, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    if mask is not None:
        mm = mask.float().view(b, 1, t, 1)
        s = s.masked_fill((mm @ mm.transpose(-2, -1)) == 0, -65504.0)
        p = s.softmax(-1) * mm
    else:
        p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(b * t, d), p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("t,use_mask", [(129, False), (129, True), (193, False), (387, True), (50, False),
                                        # the edges of "the last tile of the even-sized image is all padding" (round 6: the rollout
                                        # step's skip, the ATTN_PAIR_SKIP builds): 16 (NT - 1) tokens, one more; NT = 10 and NT = 14
                                        (144, False), (145, False), (160, False), (208, False), (209, False)])
def test_attention_fwd_bwd(ops, dtype, t, use_mask):
    b, heads, hd = 3, 12, 64
    d = heads * hd
    qkv = (torch.randn(b * t, 3 * d, generator=_g(1)) * 1.5).to(dtype).float()
    mask = None
    if use_mask:
        mask = (torch.rand(b, t, generator=_g(2)) > 0.5).to(torch.uint8)
        mask[:, 0] = 1
    qr = qkv.clone().requires_grad_(True)
    o_ref, p_ref = _attn_ref(qr, b, t, heads, hd, mask)
    do = torch.randn(b * t, d, generator=_g(3)).to(dtype).float()
    o_ref.backward(do)
    ldp = t if dtype == torch.float32 else (t + 3) // 4 * 4
    probs = torch.zeros(b, heads, t, ldp, device="cuda")
    mk = None if mask is None else mask.cuda()
    o, saved = ops.attention_fwd(qkv.to(dtype).cuda(), b, t, heads, hd, mk, probs)
    p = probs[..., :t]
    tol = _t(dtype, 2e-5, 1.5e-2)
    assert rel_err(o.float().cpu(), o_ref.detach()) < tol
    assert rel_err(p.cpu(), p_ref.detach()) < _t(dtype, 2e-5, 1e-2)
    dqkv = ops.attention_bwd(qkv.to(dtype).cuda(), do.to(dtype).cuda(), b, t, heads, hd, mk, saved, o)
    assert rel_err(dqkv.float().cpu(), qr.grad) < _t(dtype, 3e-5, 2.5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gelu(ops, dtype):
    a = (torch.randn(4096 * 4, generator=_g(1)) * 2).to(dtype)
    ar = a.float().requires_grad_(True)
    g_ref = F.gelu(ar)
    dg = torch.randn(a.shape, generator=_g(2)).to(dtype)
    g_ref.backward(dg.float())
    g = ops.gelu_fwd(a.cuda())
    da = ops.gelu_bwd(a.cuda(), dg.cuda())
    tol = _t(dtype, 1e-6, 5e-3)
    assert rel_err(g.float().cpu(), g_ref.detach()) < tol
    assert rel_err(da.float().cpu(), ar.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum_cast(ops, dtype):
    x = torch.randn(3001, 768, generator=_g(1)).to(dtype)
    s = ops.colsum(x.cuda())
    assert rel_err(s.cpu(), x.float().sum(0)) < 1e-5
    y = torch.randn(1024, 64, generator=_g(2))
    assert torch.equal(ops.cast(y.cuda(), torch.bfloat16).cpu(), y.bfloat16())
    assert torch.equal(ops.cast(y.cuda(), torch.float16).cpu(), y.half())
    assert torch.equal(ops.cast(y.half().cuda(), torch.float32).cpu(), y.half().float())


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed_fn(dtype):
    from editor_amd import functional as fn
    # unit-scale synthetic gradients: the f16 loss scale (sized for real, mean-reduced gradients) would overflow half
    old_gs = fn.F16_GRAD_SCALE
    fn.set_f16_grad_scale(1.0)
    try:
        _patch_embed_case(fn, dtype)
    finally:
        fn.set_f16_grad_scale(old_gs)


def _patch_embed_case(fn, dtype):
    b, cams, h, w, d = 4, 3, 64, 32, 256
    n = (h // 16) * (w // 16)
    g = _g(7)
    img = torch.randn(2 * b, 3, h, w, generator=g)
    cw = (torch.randn(d, 3, 16, 16, generator=g) * 0.05)
    cb, cls = torch.randn(d, generator=g) * 0.1, torch.randn(1, 1, d, generator=g)
    pos, sie = torch.randn(1, n + 1, d, generator=g), torch.randn(cams, 1, d, generator=g)
    cam = torch.randint(0, cams, (b,), generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (cw, cb, cls, pos, sie)]
    x = F.conv2d(img, leaves[0], leaves[1], stride=16).flatten(2).transpose(1, 2)
    x = torch.cat([leaves[2].expand(2 * b, -1, -1), x], 1) + leaves[3] + 3.0 * leaves[4][cam.repeat(2)]
    dx = torch.randn(x.shape, generator=g)
    x.backward(dx)
    dl = [t.clone().cuda().requires_grad_(True) for t in (cw, cb, cls, pos, sie)]
    y = fn.PatchEmbedFn.apply(img.cuda(), *dl, cam.cuda(), 3.0, dtype)
    y.backward(dx.cuda())
    tol = _t(dtype, 1e-5, 1e-2)
    assert rel_err(y.cpu(), x.detach()) < tol
    for a, r in zip(dl, leaves):
        assert rel_err(a.grad.cpu(), r.grad) < tol


def test_sfts_apply_and_pool(oracle):
    from editor_amd import functional as fn
    b, t, d = 6, 33, 256
    g = _g(3)
    feat = torch.randn(3, b, t, d, generator=g)
    index = torch.rand(b, t - 1, generator=g) > 0.5
    fr = feat.clone().requires_grad_(True)
    outs, loss = oracle.sfts_apply([fr[0], fr[1], fr[2]], index, True)
    dout = torch.randn(3, b, t, d, generator=g)
    (sum((o * dd).sum() for o, dd in zip(outs, dout)) + 1.7 * loss).backward()
    fg = feat.clone().cuda().requires_grad_(True)
    o, l, _cls = fn.SFTSApplyFn.apply(fg, index.to(torch.uint8).cuda(), True)
    ((o * dout.cuda()).sum() + 1.7 * l).backward()
    assert rel_err(o.cpu(), torch.stack(outs).detach()) == 0
    assert rel_err(l.cpu(), loss.detach()) < 1e-5
    assert rel_err(fg.grad.cpu(), fr.grad) < 1e-5
    # pooling
    x = torch.stack(outs).detach().permute(1, 0, 2, 3).reshape(b, 3 * t, d).contiguous()
    xr = x.clone().requires_grad_(True)
    parts = [xr[:, i * t:(i + 1) * t] for i in range(3)]
    num = (parts[0][:, 1:].sum(2) != 0).sum(1, keepdim=True)
    ref = torch.stack([torch.cat([p[:, 0], p[:, 1:].sum(1) / num], -1) for p in parts])
    dr = torch.randn(ref.shape, generator=g)
    ref.backward(dr)
    xg = x.clone().cuda().requires_grad_(True)
    out, n_ = fn.PoolFn.apply(xg, 3, t)
    out.backward(dr.cuda())
    assert torch.equal(n_.cpu().long(), num.view(-1))
    assert rel_err(out.cpu(), ref.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 384, 192), (387 * 4, 768, 768), (129, 128, 64)])
def test_gemm_bf16_layouts(ops, ta, tb, m, n, k):
    if ta and m % 8:
        m = (m // 8) * 8
    a = torch.randn((k, m) if ta else (m, k), generator=_g(1)).bfloat16()
    b = torch.randn((k, n) if tb else (n, k), generator=_g(2)).bfloat16()
    bias = torch.randn(n, generator=_g(3))
    ref = (a.float().t() if ta else a.float()).double() @ (b.float() if tb else b.float().t()).double()
    # bf16 output with bias
    c = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, a.shape[1], b.shape[1], n, ta, tb, bias=bias.cuda())
    assert rel_err(c.float().cpu(), ref + bias.double()) < 4e-3
    # fp32 output, residual accumulate with per-row scale
    c0 = torch.randn(m, n, generator=_g(4))
    rs = torch.rand(m, generator=_g(5))
    c = c0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, a.shape[1], b.shape[1], n, ta, tb, alpha=0.5, beta=1.0, bias=bias.cuda(),
             rowscale=rs.cuda())
    assert rel_err(c.cpu(), rs.view(-1, 1).double() * (0.5 * ref + bias.double()) + c0.double()) < 1e-5
    # split-K atomics into fp32
    c = torch.empty(m, n, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c, m, n, k, a.shape[1], b.shape[1], n, ta, tb, splitk=3)
    assert rel_err(c.cpu(), ref) < 1e-5


def test_gemm_bf16_wgrad_shape(ops):
    """dW = dy^T x at the backbone's real reduction length (M = 3*128*129 token rows)."""
    m, n, k = 49536, 256, 384
    dy = (torch.randn(m, n, generator=_g(1)) * 0.1).bfloat16()
    x = torch.randn(m, k, generator=_g(2)).bfloat16()
    dw = torch.empty(n, k, device="cuda")
    ops.gemm(dy.cuda(), x.cuda(), dw, n, k, m, n, k, k, 1, 1, splitk=24)
    assert rel_err(dw.cpu(), dy.float().t().double() @ x.float().double()) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(ops, dtype):
    m, n, k = 520, 384, 256
    a = torch.randn(m, k, generator=_g(1)).to(dtype)
    w = (torch.randn(n, k, generator=_g(2)) * 0.1).to(dtype)
    bias = torch.randn(n, generator=_g(3)) * 0.1
    ref = a.float() @ w.float().t() + bias
    tol = _t(dtype, 1e-5, 6e-3)
    # residual: C = rowscale * (A W^T + b) + R
    res = torch.randn(m, n, generator=_g(4))
    rs = torch.rand(m, generator=_g(5))
    c = torch.empty(m, n, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), c, m, n, k, k, k, n, 0, 0, bias=bias.cuda(), rowscale=rs.cuda(),
             epilogue=ops.EPI_RESIDUAL, aux=res.cuda())
    assert rel_err(c.cpu(), rs.view(-1, 1) * ref + res) < tol
    # GELU forward: aux <- pre-activation, C <- gelu
    pre = torch.empty(m, n, dtype=dtype, device="cuda")
    act = torch.empty(m, n, dtype=dtype, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), act, m, n, k, k, k, n, 0, 0, bias=bias.cuda(), epilogue=ops.EPI_GELU, aux=pre)
    assert rel_err(pre.float().cpu(), ref) < tol
    assert rel_err(act.float().cpu(), F.gelu(pre.float().cpu())) < tol
    # GELU backward on a dgrad: C = (dy W) * gelu'(pre)
    dy = torch.randn(m, n, generator=_g(6)).to(dtype)
    pre2 = (torch.randn(m, k, generator=_g(7))).to(dtype)
    pr = pre2.float().requires_grad_(True)
    F.gelu(pr).backward(dy.float() @ w.float())
    out = torch.empty(m, k, dtype=dtype, device="cuda")
    ops.gemm(dy.cuda(), w.cuda(), out, m, k, n, n, k, k, 0, 1, epilogue=ops.EPI_GELU_BWD, aux=pre2.cuda())
    assert rel_err(out.float().cpu(), pr.grad) < tol
    if dtype != torch.float32:
        # derivative-saving form (EPI_AUX_GRAD) on the small-tile kernels, and the k-major W^T form of the same dgrad
        dsave = torch.empty(m, n, dtype=dtype, device="cuda")
        ops.gemm(a.cuda(), w.cuda(), act, m, n, k, k, k, n, 0, 0, bias=bias.cuda(), epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=dsave)
        pq = pre.float().cpu().requires_grad_(True)
        F.gelu(pq).sum().backward()
        assert rel_err(dsave.float().cpu(), pq.grad) < tol
        gsave = (pr.detach() * 0 + torch.autograd.functional.jacobian(lambda z: F.gelu(z).sum(), pre2.float())).to(dtype)
        out2 = torch.empty(m, k, dtype=dtype, device="cuda")
        ops.gemm(dy.cuda(), w.t().contiguous().cuda(), out2, m, k, n, n, n, k, 0, 0, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD,
                 aux=gsave.cuda())
        assert rel_err(out2.float().cpu(), pr.grad) < 2 * tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("m,n,k", [(2100, 768, 512), (2560, 520, 128), (4100, 1024, 768), (2049, 256, 64)])
def test_gemm_h16_pingpong_kernel(ops, tb, m, n, k, dtype):
    """The 256x256 ping-pong kernel (selected for M >= 2048, N >= 512, A k-major; forced here for the narrower N too):
    ragged M and N edges, short and odd K-tile counts, every fused epilogue, 16-bit (bf16 / f16) and fp32 outputs, the
    live-row form."""
    FP = ops.EPI_FORCE_PP                          # explicit kernel choice (include/editor_hip.h), no environment switch
    tol = 4e-3 if dtype == torch.bfloat16 else 5e-4             # output rounding: 2^-9 / 2^-12 relative

    def gemm(*args, epilogue=0, **kw):
        ops.gemm(*args, epilogue=epilogue | FP, **kw)
    a = torch.randn(m, k, generator=_g(1)).to(dtype)
    b = (torch.randn((k, n) if tb else (n, k), generator=_g(2)) * 0.1).to(dtype)
    bias = torch.randn(n, generator=_g(3)) * 0.1
    rs = torch.rand(m, generator=_g(5)) + 0.5
    ref = a.float().double() @ (b.float() if tb else b.float().t()).double()
    ldb = b.shape[1]
    ag, bg = a.cuda(), b.cuda()
    # plain, and bias + row scale, 16-bit out (one-pass staged epilogue)
    c = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, c, m, n, k, k, ldb, n, 0, tb)
    assert rel_err(c.float().cpu(), ref) < tol
    gemm(ag, bg, c, m, n, k, k, ldb, n, 0, tb, alpha=0.5, bias=bias.cuda(), rowscale=rs.cuda())
    want = rs.view(-1, 1).double() * (0.5 * ref + bias.double())
    assert rel_err(c.float().cpu(), want) < tol
    # GELU forward (pre-activation saved)
    pre = torch.empty(m, n, dtype=dtype, device="cuda")
    act = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, act, m, n, k, k, ldb, n, 0, tb, bias=bias.cuda(), epilogue=ops.EPI_GELU, aux=pre)
    assert rel_err(pre.float().cpu(), ref + bias.double()) < tol
    assert rel_err(act.float().cpu(), F.gelu(pre.float().cpu())) < tol
    # the packed-math erfc form against exact erf on the stored pre-activations: well inside the output rounding
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (act.float().cpu() - F.gelu(pre.float().cpu())).abs().max() <= ulp * act.float().abs().max().item()
    # GELU' epilogue
    pre2 = torch.randn(m, n, generator=_g(7)).to(dtype)
    pr = pre2.float().requires_grad_(True)
    F.gelu(pr).backward(ref.float())
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, out, m, n, k, k, ldb, n, 0, tb, epilogue=ops.EPI_GELU_BWD, aux=pre2.cuda())
    assert rel_err(out.float().cpu(), pr.grad) < 1.5 * tol
    # the derivative-saving form the training step uses (EPI_AUX_GRAD): forward stores gelu'(rounded pre-activation),
    # the dgrad epilogue multiplies by it - same result as the erfc form above up to one more 16-bit rounding
    dsave = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, act, m, n, k, k, ldb, n, 0, tb, bias=bias.cuda(), epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=dsave)
    pq = pre.float().cpu().requires_grad_(True)
    F.gelu(pq).sum().backward()
    assert rel_err(dsave.float().cpu(), pq.grad) < tol
    assert rel_err(act.float().cpu(), F.gelu(pre.float().cpu())) < tol
    out2 = torch.empty(m, n, dtype=dtype, device="cuda")
    pre2_grad = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, torch.empty_like(act), m, n, k, k, ldb, n, 0, tb, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=pre2_grad)
    pg = F.gelu  # (reference: gelu' of the rounded product)
    prod = torch.empty(m, n, dtype=dtype, device="cuda")
    gemm(ag, bg, prod, m, n, k, k, ldb, n, 0, tb)
    pr2 = prod.float().cpu().requires_grad_(True)
    F.gelu(pr2).sum().backward()
    gemm(ag, bg, out2, m, n, k, k, ldb, n, 0, tb, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD, aux=pre2_grad)
    assert rel_err(out2.float().cpu(), ref * pr2.grad.double()) < 2 * tol
    # fp32 residual epilogue and fp32 plain
    res = torch.randn(m, n, generator=_g(4))
    cf = torch.empty(m, n, device="cuda")
    gemm(ag, bg, cf, m, n, k, k, ldb, n, 0, tb, bias=bias.cuda(), rowscale=rs.cuda(), epilogue=ops.EPI_RESIDUAL,
         aux=res.cuda())
    assert rel_err(cf.cpu(), rs.view(-1, 1).double() * (ref + bias.double()) + res.double()) < 1e-5
    gemm(ag, bg, cf, m, n, k, k, ldb, n, 0, tb)
    assert rel_err(cf.cpu(), ref) < 1e-5
    # live-row form (compacted HMA): tiles at or beyond *m_live are skipped, rows below it are exact
    live = 1000
    c.fill_(7.0)
    gemm(ag, bg, c, m, n, k, k, ldb, n, 0, tb, m_live=torch.tensor([live], dtype=torch.int32, device="cuda"))
    assert rel_err(c[:live].float().cpu(), ref[:live]) < tol
    assert float((c[1024:] - 7.0).abs().max()) == 0.0          # 256-row tiles from row 1024 on were never touched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h", [208])
@pytest.mark.parametrize("m,n,k", [(3 * 128 * 129, 768, 768), (5000, 1024, 128), (2049, 512, 192)])
def test_gemm_h16_short_tiles(ops, m, n, k, h, dtype, monkeypatch):
    """EDITOR_EPI_TILE_ROWS (208-row tiles of the ping-pong kernel, chosen by ops.gemm_tile_rows to fill the last
    round of workgroups): every epilogue gives the SAME BITS as the full 256-row tiles - the accumulation order along K is
    unchanged - and the per-tile column sums fold to the same totals (different grouping: fp32 rounding only)."""
    a = torch.randn(m, k, generator=_g(1)).to(dtype).cuda()
    b = (torch.randn(n, k, generator=_g(2)) * 0.1).to(dtype).cuda()
    bias = (torch.randn(n, generator=_g(3)) * 0.1).cuda()
    rs = (torch.rand(m, generator=_g(5)) + 0.5).cuda()
    res = torch.randn(m, n, generator=_g(4)).cuda()
    dsave = torch.randn(m, n, generator=_g(7)).to(dtype).cuda()

    def run(flag):
        outs = []
        c = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        ops.gemm(a, b, c, m, n, k, k, k, n, 0, 0, alpha=0.5, bias=bias, rowscale=rs, epilogue=flag)
        outs.append(c)
        act, sav = torch.full_like(c, float("nan")), torch.full_like(c, float("nan"))
        ops.gemm(a, b, act, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD | flag, aux=sav)
        outs += [act, sav]
        d = torch.full_like(c, float("nan"))
        ops.gemm(a, b, d, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_GELU_BWD | ops.EPI_AUX_GRAD | flag, aux=dsave)
        outs.append(d)
        cf = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm(a, b, cf, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL | flag, aux=res)
        outs.append(cf)
        cs = None
        if ops.gemm_colsum_ok(m, n, k, dtype, 0, 1, None):
            cs = torch.zeros(n, device="cuda")
            c2 = torch.full_like(c, float("nan"))
            ops.gemm(a, b, c2, m, n, k, k, k, n, 0, 0, epilogue=flag, colsum=cs)
            outs.append(c2)
        torch.cuda.synchronize()
        return outs, cs

    monkeypatch.setattr(ops, "SHORT_TILES", False)               # explicit flag below instead of the shape heuristic
    full, cs_full = run(ops.EPI_FORCE_PP)
    short, cs_short = run(ops.EPI_TILE_ROWS(h))
    for x, y in zip(full, short):
        assert not torch.isnan(y.float()).any()
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    if cs_full is not None:
        assert rel_err(cs_short.cpu(), cs_full.cpu().double()) < 1e-5
        assert rel_err(cs_full.cpu(), full[-1].double().sum(0).cpu()) < 1e-5
    # invalid uses are errors, not silent full tiles (the column-sum layout depends on h)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b.t().contiguous(), full[0], m, n, k, k, n, n, 0, 1, epilogue=ops.EPI_TILE_ROWS(h))


def test_attention_varlen_matches_dense_reference(ops):
    """Compacted (variable-length) attention == per-sequence dense softmax attention (fp32 reference)."""
    heads, hd = 12, 64
    d = heads * hd
    lens = [129, 60, 1, 77, 128]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    total = int(cu[-1])
    rows = (total + 63) // 64 * 64
    g = _g(5)
    qkv = (torch.randn(rows, 3 * d, generator=g) * 1.2).bfloat16()
    qkv[total:] = 0
    do = torch.randn(rows, d, generator=g).bfloat16()
    do[total:] = 0
    qr = qkv.float().requires_grad_(True)
    outs = []
    for i, n in enumerate(lens):
        s0 = int(cu[i])
        o, _ = _attn_ref(qr[s0:s0 + n], 1, n, heads, hd, None)
        outs.append(o)
    ref = torch.cat(outs + [torch.zeros(rows - total, d)])
    ref.backward(do.float())
    o, lse = ops.attention_fwd(qkv.cuda(), len(lens), max(lens), heads, hd, None, None, cu=cu.cuda())
    assert rel_err(o.float().cpu(), ref.detach()) < 1.5e-2
    assert float(o[total:].abs().max()) == 0.0
    dqkv = ops.attention_bwd(qkv.cuda(), do.cuda(), len(lens), max(lens), heads, hd, None, lse, o, cu=cu.cuda())
    assert rel_err(dqkv.float().cpu(), qr.grad) < 2.5e-2
    assert float(dqkv[total:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hd,heads", [(32, 12), (96, 8)])
@pytest.mark.parametrize("t,use_mask", [(129, False), (129, True), (193, False), (387, True), (50, False), (700, False)])
def test_attention_other_head_widths(ops, dtype, hd, heads, t, use_mask):
    """Round 4: the fused 16-bit kernels built for the factory's other head widths (vit_small_patch16_224: 8 heads of 96,
    vit_pytorch.py:704-713; DeiT-small's 12 HMA heads of 32, :716-727) - forward, probabilities, lse and the two-pass backward
    against the fp32 softmax attention, in the whole-sequence form (T <= 416 at 96 columns) and the chunked one (T = 700)."""
    b = 3
    d = heads * hd
    assert hd in ops.ATTN_HEAD_WIDTHS
    qkv = (torch.randn(b * t, 3 * d, generator=_g(1)) * 1.5).to(dtype).float()
    mask = None
    if use_mask:
        mask = (torch.rand(b, t, generator=_g(2)) > 0.5).to(torch.uint8)
        mask[:, 0] = 1
    qr = qkv.clone().requires_grad_(True)
    o_ref, p_ref = _attn_ref(qr, b, t, heads, hd, mask)
    do = torch.randn(b * t, d, generator=_g(3)).to(dtype).float()
    o_ref.backward(do)
    ldp = (t + 3) // 4 * 4
    probs = torch.zeros(b, heads, t, ldp, device="cuda") if t <= 416 else None      # (no probability output in the chunked form)
    mk = None if mask is None else mask.cuda()
    o, lse = ops.attention_fwd(qkv.to(dtype).cuda(), b, t, heads, hd, mk, probs)
    assert lse.dim() == 1 and lse.numel() == heads * b * t                            # the fused path, not the fp32 detour
    assert rel_err(o.float().cpu(), o_ref.detach()) < 1.5e-2
    if probs is not None:
        assert rel_err(probs[..., :t].cpu(), p_ref.detach()) < 1e-2
    dqkv = ops.attention_bwd(qkv.to(dtype).cuda(), do.to(dtype).cuda(), b, t, heads, hd, mk, lse, o)
    assert rel_err(dqkv.float().cpu(), qr.grad) < 2.5e-2
    # same operands through the exact-f32 kernels: the 16-bit result differs from it by 16-bit rounding only
    if t <= 416:
        o32, _ = ops.attention_fwd(qkv.cuda(), b, t, heads, hd, mk, None)
        assert rel_err(o.float().cpu(), o32.cpu()) < 1.5e-2


@pytest.mark.parametrize("hd,heads", [(32, 12), (96, 8)])
def test_attention_varlen_other_head_widths(ops, hd, heads):
    """Compacted (variable-length) attention at 32- and 96-wide heads == per-sequence dense softmax attention."""
    d = heads * hd
    lens = [129, 60, 1, 77, 128]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    total = int(cu[-1])
    rows = (total + 63) // 64 * 64
    g = _g(5)
    qkv = (torch.randn(rows, 3 * d, generator=g) * 1.2).bfloat16()
    qkv[total:] = 0
    do = torch.randn(rows, d, generator=g).bfloat16()
    do[total:] = 0
    qr = qkv.float().requires_grad_(True)
    outs = []
    for i, n in enumerate(lens):
        s0 = int(cu[i])
        o, _ = _attn_ref(qr[s0:s0 + n], 1, n, heads, hd, None)
        outs.append(o)
    ref = torch.cat(outs + [torch.zeros(rows - total, d)])
    ref.backward(do.float())
    o, lse = ops.attention_fwd(qkv.cuda(), len(lens), max(lens), heads, hd, None, None, cu=cu.cuda())
    assert rel_err(o.float().cpu(), ref.detach()) < 1.5e-2
    assert float(o[total:].abs().max()) == 0.0
    dqkv = ops.attention_bwd(qkv.cuda(), do.cuda(), len(lens), max(lens), heads, hd, None, lse, o, cu=cu.cuda())
    assert rel_err(dqkv.float().cpu(), qr.grad) < 2.5e-2
    assert float(dqkv[total:].abs().max()) == 0.0


def _colsum_case(ops, dtype, hd, heads, b, t, form, seed):
    """attention_bwd(colsum=...) against the plain entry: same dqkv bits, and the column sums equal the sums of the fp32 values the
    stored 16-bit dqkv was rounded from - i.e. they differ from colsum(stored dqkv) by at most the rounding of each entry."""
    d = heads * hd
    g = _g(seed)
    mask = cu = None
    if form == "varlen":
        lens = [t] + [max(1, int(x)) for x in torch.randint(1, t + 1, (b - 1,), generator=g)]
        cu_h = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
        total = int(cu_h[-1])
        rows = (total + 63) // 64 * 64
        cu = cu_h.cuda()
    else:
        rows = total = b * t
        if form == "masked":
            mask = (torch.rand(b, t, generator=g) > 0.3)
            mask[:, 0] = True
            mask = mask.to(torch.uint8).cuda()
    qkv = (torch.randn(rows, 3 * d, generator=g) * 0.9).to(dtype)
    do = torch.randn(rows, d, generator=g).to(dtype)
    qkv[total:] = 0
    do[total:] = 0
    qkv, do = qkv.cuda(), do.cuda()
    o, lse = ops.attention_fwd(qkv, b, t, heads, hd, mask, None, cu=cu)
    ref = ops.attention_bwd(qkv, do, b, t, heads, hd, mask, lse, o, cu=cu)
    cs = torch.full((3 * d,), float("nan"), device="cuda")
    got = ops.attention_bwd(qkv, do, b, t, heads, hd, mask, lse, o, cu=cu, colsum=cs, colsum_scale=0.5)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    stored = ref.double()
    want = stored.sum(0)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # every entry within half an ulp of its fp32 value (+ fp32 summation noise)
    bound = stored.abs().sum(0) * (0.5 * eps + 4e-6) + 1e-6
    err = (cs.double() * 2.0 - want).abs()
    assert torch.isfinite(cs).all()
    assert bool((err <= bound).all()), (float((err / bound).max()), form, t, hd)
    # and far inside that bound on average: the sums are of the unrounded values, not garbage that happens to fit
    assert float(err.mean() / stored.abs().sum(0).mean()) < 0.1 * eps
    # deterministic
    cs2 = torch.empty_like(cs)
    ops.attention_bwd(qkv, do, b, t, heads, hd, mask, lse, o, cu=cu, colsum=cs2, colsum_scale=0.5)
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form,t", [("dense", 129), ("dense", 100), ("dense", 193), ("dense", 300), ("masked", 129), ("varlen", 129),
                                    ("varlen", 387)])
def test_attention_bwd_column_sums(ops, dtype, form, t):
    """The qkv bias gradient from the attention backward's own accumulators (editor_attention_bwd_colsum_*), 64-wide heads: every
    kernel form (unrolled dense, short dense, 14- and 26-tile, token-masked, packed sequences incl. the joint block's 387)."""
    _colsum_case(ops, dtype, 64, 12, 6, t, form, seed=t)


@pytest.mark.parametrize("hd,heads,t", [(32, 12, 129), (32, 12, 200), (96, 8, 129), (96, 8, 77)])
@pytest.mark.parametrize("form", ["dense", "varlen"])
def test_attention_bwd_column_sums_other_head_widths(ops, hd, heads, t, form):
    _colsum_case(ops, torch.bfloat16, hd, heads, 5, t, form, seed=hd + t)


def test_attention_bwd_column_sums_refused_where_not_built(ops):
    """96-wide heads beyond 160 tokens and sequences beyond 608 tokens: no in-kernel column sums - ops says so up front."""
    import editor_amd.ops as ops_mod
    q96 = torch.zeros(4, 3 * 8 * 96, dtype=torch.bfloat16, device="cuda")
    q64 = torch.zeros(4, 3 * 12 * 64, dtype=torch.bfloat16, device="cuda")
    assert ops_mod.attention_bwd_colsum_ok(q96, 160, 96) and not ops_mod.attention_bwd_colsum_ok(q96, 193, 96)
    assert ops_mod.attention_bwd_colsum_ok(q64, 608, 64) and not ops_mod.attention_bwd_colsum_ok(q64, 609, 64)
    assert not ops_mod.attention_bwd_colsum_ok(q64.float(), 129, 64)
    # the C entry itself: NULL partial rows, a sequence beyond 608 tokens and 96-wide heads beyond 160 are errors, not silent no-ops;
    # ops.attention_bwd(colsum=...) refuses what attention_bwd_colsum_ok refuses
    def raw(hd, heads, t, parts=True):
        b = 2
        qkv = (torch.randn(b * t, 3 * heads * hd) * 0.5).bfloat16().cuda()
        o, lse = ops.attention_fwd(qkv, b, t, heads, hd, None, None)
        dq = torch.empty_like(qkv)
        ws = torch.empty(heads * b * t, device="cuda")
        cp = torch.empty(b * 3 * heads * hd, device="cuda") if parts else None
        ops_mod.call("editor_attention_bwd_colsum_bf16", qkv, o, o, lse, b, t, heads, hd, hd ** -0.5, None, dq, ws, None, b * t, cp)
    raw(64, 12, 129)
    for bad in ((64, 12, 129, False), (64, 12, 640), (96, 8, 193)):
        with pytest.raises(RuntimeError):
            raw(*bad)
    with pytest.raises(ValueError):
        qkv = torch.zeros(2 * 193, 3 * 8 * 96, dtype=torch.bfloat16, device="cuda")
        ops.attention_bwd(qkv, qkv[:, :768].contiguous(), 2, 193, 8, 96, None, torch.zeros(8 * 2 * 193, device="cuda"),
                          qkv[:, :768].contiguous(), colsum=torch.empty(3 * 768, device="cuda"))


@pytest.mark.parametrize("hd,heads", [(32, 12), (96, 8), (64, 12)])
@pytest.mark.parametrize("t", [129, 193, 144, 145])
def test_rollout_recomputed_other_head_widths(ops, hd, heads, t):
    """Rollout steps that recompute P from (qkv, lse) at 32- / 96-wide heads == the rollout over the materialised probabilities;
    the one-launch form == the per-layer steps, bit for bit."""
    import editor_amd.ops as ops_mod
    b, layers = 4, 4
    g = torch.Generator().manual_seed(3)
    ldp = (t + 3) // 4 * 4
    probs = torch.empty(layers, b, heads, t, ldp, device="cuda")
    pairs = []
    for l in range(layers):
        qkv = (torch.randn(b * t, 3 * heads * hd, generator=g) * 0.7).bfloat16().cuda()
        _, lse = ops.attention_fwd(qkv, b, t, heads, hd, None, probs[l])
        pairs.append((qkv, lse))
    ref = ops.attn_rollout(probs)
    got = ops.attn_rollout_qk(pairs, b, t, heads, hd)
    assert got.shape == ref.shape == (b, heads, t - 1)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-5


@pytest.mark.parametrize("m,n,k", [(512, 256, 64), (777, 768, 768), (2049, 512, 1536)])
def test_four_wave_gemm_tile_equals_ping_pong(ops, m, n, k):
    """Round 4 bring-up (csrc/gemm_w4.hip, probe library): the 256 x 256 tile on four wavefronts - one per SIMD, 128 x 128 each,
    accumulators in the accumulator file, LDS-DMA pieces and fragment reads between the wave's own MFMAs, two workgroup barriers
    per K-tile - is BIT-IDENTICAL to the eight-wave ping-pong kernel (same fragment maps, same summation order).  Measured equal,
    not faster (profiles/r04_gemm_w4_probe.txt), so the product path keeps the ping-pong kernel; this pins the alternative."""
    import ctypes
    from editor_amd import _lib
    fn = _lib.probe_lib().editor_probe_gemm_w4
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_int, ctypes.c_void_p]
    for dtype, f16 in ((torch.bfloat16, 0), (torch.float16, 1)):
        x = torch.randn(m, k, generator=_g(21)).to(dtype).cuda()
        w = (torch.randn(n, k, generator=_g(22)) * 0.05).to(dtype).cuda()
        y0 = torch.empty(m, n, dtype=dtype, device="cuda")
        y1 = torch.full((m, n), 3.0, dtype=dtype, device="cuda")
        ops.gemm(x, w, y0, m, n, k, k, k, n, 0, 0, epilogue=ops.EPI_FORCE_PP)
        rc = fn(x.data_ptr(), w.data_ptr(), y1.data_ptr(), f16, m, n, k, k, k, n, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)
        assert rel_err(y1.float().cpu(), x.float().cpu() @ w.float().cpu().t()) < 1e-2


def test_compact_plan_and_rows(ops):
    g = _g(11)
    b, n, d, nmod = 7, 128, 256, 3
    t = n + 1
    index = (torch.rand(b, n, generator=g) > 0.55).to(torch.uint8)
    plan = ops.CompactPlan(index.cuda(), t, nmod)
    lens = 1 + index.sum(1)
    assert plan.total == int(lens.sum())
    assert plan.cu.cpu().tolist() == [0] + lens.cumsum(0).tolist()
    dense = torch.randn(nmod * b * t, d, generator=g)
    xa = ops.gather_rows(dense.cuda(), plan.map_a).cpu().view(nmod, plan.ma, d)
    cu = plan.cu.cpu()
    for m in range(nmod):
        for bb in range(b):
            toks = [0] + [int(i) + 1 for i in torch.nonzero(index[bb]).flatten()]
            ref = dense.view(nmod, b, t, d)[m, bb, toks]
            assert torch.equal(xa[m, cu[bb]:cu[bb + 1]], ref)
        assert float(xa[m, plan.total:].abs().max()) == 0.0
    assert plan.mask_a.cpu().sum() == plan.total and plan.mask_b.cpu().sum() == nmod * plan.total
    xb = ops.gather_rows(xa.reshape(-1, d).cuda(), plan.map_b).cpu()
    for bb in range(b):
        ln = int(cu[bb + 1] - cu[bb])
        for m in range(nmod):
            assert torch.equal(xb[nmod * cu[bb] + m * ln: nmod * cu[bb] + (m + 1) * ln], xa[m, cu[bb]:cu[bb + 1]])
    # scatter is the exact adjoint of gather
    back = ops.scatter_rows(xa.reshape(-1, d).cuda(), plan.map_a, nmod * b * t).cpu().view(nmod, b, t, d)
    keep = torch.cat([torch.ones(b, 1, dtype=torch.bool), index.bool()], 1)
    assert torch.equal(back, dense.view(nmod, b, t, d) * keep.view(1, b, t, 1))


def test_fused_sgd_matches_torch():
    from editor_amd.optim import FusedSGD
    g = _g(21)
    shapes = [(768, 768), (3072,), (171, 2304), (1, 1, 768), (50001,), (3,)]
    names = ["a.weight", "a.bias", "b.weight", "cls_token", "c.weight", "d.bias"]
    ps = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    groups = [{"params": [r], "lr": 2e-3 if "bias" in n else 1e-3, "weight_decay": 1e-4} for n, r in zip(names, ref)]
    topt = torch.optim.SGD(groups, momentum=0.9)
    fopt = FusedSGD(list(zip(names, ps)), base_lr=1e-3, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4,
                    momentum=0.9)
    for step in range(3):
        for p, r in zip(ps, ref):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad = gr.clone() if not (step == 1 and p.numel() == 3) else None      # a parameter without a gradient
            r.grad = gr.clone() if p.grad is not None else None
        fopt.step()
        topt.step()
        for p, r in zip(ps, ref):
            assert rel_err(p.detach().cpu(), r.detach().cpu()) < 1e-6
        # the bf16 shadows written by the same launch == a cast of the updated weights, and the operand cache serves them
        from editor_amd import functional as fnc
        for p, h in zip(ps, fopt.shadows):
            if h is not None and p.grad is not None:
                assert torch.equal(h, p.detach().bfloat16())
                assert fnc.act_weight(p, torch.bfloat16).data_ptr() == h.data_ptr()


def test_droppath_scales_device_state(ops):
    rates = torch.linspace(0, 0.1, 12).cuda()
    state = torch.full((1,), 77, dtype=torch.int64, device="cuda")
    a = ops.droppath_scales_dev(rates, 384, 129, state)
    b = ops.droppath_scales_dev(rates, 384, 129, state)
    assert int(state.item()) == 79                                     # advanced once per call
    assert torch.equal(a, ops.droppath_scales(rates, 384, 129, 77)) and torch.equal(b, ops.droppath_scales(rates, 384, 129, 78))
    assert not torch.equal(a, b)


def test_droppath_scales(ops):
    rates = torch.linspace(0, 0.1, 12).cuda()
    s = ops.droppath_scales(rates, 384, 129, 1234).cpu()
    assert s.shape == (12, 2, 384 * 129)
    assert torch.all(s[0] == 1.0)                                   # rate 0: always kept, scale 1
    per_sample = s.view(12, 2, 384, 129)
    assert torch.all(per_sample == per_sample[..., :1])             # constant over the tokens of a sample
    kp = 1 - 0.1
    vals = per_sample[11, :, :, 0].flatten()
    assert set(torch.unique(vals).tolist()) <= {0.0, float(torch.tensor(1.0) / torch.tensor(kp))}
    assert abs(float((vals > 0).float().mean()) - kp) < 0.06         # ~90 % kept
    assert not torch.equal(per_sample[11, 0, :, 0], per_sample[11, 1, :, 0])   # independent draws per branch


def test_cast_rows_colsum(ops):
    m, d = 1037, 768
    x = torch.randn(m, d, generator=_g(1)) * 3
    rs = torch.rand(m, generator=_g(2)) * 1.3
    out, cs = ops.cast_rows_colsum(x.cuda(), rs.cuda(), torch.bfloat16)
    ref = (x * rs.view(-1, 1)).bfloat16()
    assert torch.equal(out.cpu(), ref)
    assert rel_err(cs.cpu(), ref.float().sum(0)) < 1e-5
    assert rel_err(cs.cpu(), ops.colsum(out).cpu()) < 1e-6          # == the separate pass it replaces


def test_gemm_bf16_colsum_side_output(ops):
    """dgrad + GELU' with the column sums of its output (the next layer's bias gradient) from the same epilogue."""
    m, n, k = 2300, 1024, 256
    dy = torch.randn(m, k, generator=_g(1)).bfloat16()
    w = (torch.randn(k, n, generator=_g(2)) * 0.1).bfloat16()
    pre = torch.randn(m, n, generator=_g(3)).bfloat16()
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    cs = torch.empty(n, device="cuda")
    ops.gemm(dy.cuda(), w.cuda(), out, m, n, k, k, n, n, 0, 1, epilogue=ops.EPI_GELU_BWD, aux=pre.cuda(), colsum=cs)
    ref = torch.empty_like(out)
    ops.gemm(dy.cuda(), w.cuda(), ref, m, n, k, k, n, n, 0, 1, epilogue=ops.EPI_GELU_BWD, aux=pre.cuda())
    assert torch.equal(out, ref)
    assert rel_err(cs.cpu(), ref.float().sum(0).cpu()) < 1e-5
    assert rel_err(cs.cpu(), ops.colsum(ref).cpu()) < 1e-5
    cs2 = torch.empty(n, device="cuda")
    ops.gemm(dy.cuda(), w.cuda(), out, m, n, k, k, n, n, 0, 1, colsum=cs2)              # plain epilogue too
    assert rel_err(cs2.cpu(), out.float().sum(0).cpu()) < 1e-5


def test_gemm_bf16_wgrad_ragged_reduction_is_deterministic(ops):
    """Token-row counts that are not a multiple of 64 (e.g. B = 32: 3*32*129 = 12 384): whole K-tiles through the
    split-K slab kernel + one tail product; bit-identical run to run and equal to the fp64 reference."""
    m, n, k = 12384, 768, 384
    dy = (torch.randn(m, n, generator=_g(1)) * 0.1).bfloat16().cuda()
    x = torch.randn(m, k, generator=_g(2)).bfloat16().cuda()
    outs = []
    for _ in range(3):
        dw = torch.empty(n, k, device="cuda")
        ops.gemm(dy, x, dw, n, k, m, n, k, k, 1, 1, splitk=6)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_err(outs[0].cpu(), dy.float().t().double().cpu() @ x.float().double().cpu()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_weight_gradients_match_separate_products(dtype):
    """editor_gemm_wgrad_group: the four dW = dy^T x products of a block in one launch == the four separate split-K
    products (same operands, fp32 accumulation; only the split of the reduction differs), deterministic run to run, and
    with a device-side live-row count (compacted HMA)."""
    from editor_amd import ops
    g = _g(21)
    m = 4160
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    jobs, refs = [], []
    for n, k in shapes:
        dy = (torch.randn(m, n, generator=g) * 0.5).to(dtype).cuda()
        x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).cuda()
        jobs.append((dy, x, torch.empty(n, k, device="cuda")))
        refs.append(dy.float().t() @ x.float())
    ops.gemm_wgrad_group(jobs, m, alpha=0.5)
    first = [j[2].clone() for j in jobs]
    for (dy, x, dw), ref in zip(jobs, refs):
        assert rel_err(dw.cpu(), 0.5 * ref.cpu()) < 2e-5
    ops.gemm_wgrad_group(jobs, m, alpha=0.5)
    assert all(torch.equal(a, j[2]) for a, j in zip(first, jobs))                # fixed-order slab reduction
    live = 2500
    for dy, x, _ in jobs:
        dy[live:] = 0
        x[live:] = 0
    ops.gemm_wgrad_group(jobs, m, alpha=1.0, m_live=torch.tensor([live], dtype=torch.int32, device="cuda"))
    for dy, x, dw in jobs:
        assert rel_err(dw.cpu(), (dy[:live].float().t() @ x[:live].float()).cpu()) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,d", [(1031, 768), (516, 1024), (77, 256)])
def test_resid_add_layernorm_fwd(ops, m, d, dtype):
    """Round 4, cfg.MODEL.BRANCH16: x_out = x + rowscale * branch (16-bit branch), y = LN(x_out) in one pass == the residual add in
    fp32 followed by the stand-alone LayerNorm kernel, BIT FOR BIT (same per-row arithmetic on the same fp32 rows), and close to
    torch's fp32 layer_norm."""
    x = torch.randn(m, d, generator=_g(1)).cuda() * 3
    br = (torch.randn(m, d, generator=_g(2)) * 0.7).to(dtype).cuda()
    rs = ((torch.rand(m, generator=_g(3)) > 0.2).float() / 0.8).cuda()
    g = (torch.rand(d, generator=_g(4)) + 0.5).cuda()
    b = (torch.randn(d, generator=_g(5)) * 0.1).cuda()
    for scale in (rs, None):
        xo, y, mean, rstd = ops.resid_add_layernorm_fwd(x, br, scale, g, b, 1e-6)
        want_x = x + (br.float() * scale[:, None] if scale is not None else br.float())
        assert torch.equal(xo, want_x)
        y2, mean2, rstd2 = ops.layernorm_fwd(want_x, g, b, 1e-6, dtype)
        assert torch.equal(y.view(torch.int16), y2.view(torch.int16)) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
        ref = F.layer_norm(want_x, (d,), g, b, 1e-6)
        assert rel_err(y.float().cpu(), ref.cpu()) < (5e-3 if dtype == torch.bfloat16 else 6e-4)
    with pytest.raises(RuntimeError):                      # D = 384: not a multiple of 256 - an error, not a silent fallback
        ops.resid_add_layernorm_fwd(torch.zeros(8, 384, device="cuda"), torch.zeros(8, 384, dtype=dtype, device="cuda"), None,
                                    torch.ones(384, device="cuda"), torch.zeros(384, device="cuda"), 1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_stagger_and_tile_height_do_not_change_a_bit(ops, dtype):
    """EDITOR_EPI_STAGGER (the first round's workgroups start spread in time: a per-workgroup s_memtime spin, ON by default
    for the dense qkv forward) and the counted-vmcnt choreography behind it must leave every output bit as it was - plain, bias,
    GELU + saved gelu', fp32 residual, 208-row tiles, column sums - at a size with more than one round of workgroups."""
    m, k = 3 * 32 * 129, 768                                        # 12 384 rows: 49 x 9 = 441 tiles of a 2304-wide output
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(m, k, device="cuda", generator=g).to(dtype)
    st = ops.EPI_STAGGER(24)

    def run(n, epilogue, c_dtype, **kw):
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dtype)
        bias = torch.randn(n, device="cuda", generator=g)
        outs = []
        for extra in (0, st):
            c = torch.empty(m, n, dtype=c_dtype, device="cuda")
            aux = kw.get("aux")
            kw2 = dict(kw)
            if aux == "new":
                kw2["aux"] = torch.empty(m, n, dtype=dtype, device="cuda")
            cs = torch.empty(n, dtype=torch.float32, device="cuda") if kw.get("colsum") else None
            if cs is not None:
                kw2["colsum"] = cs
            ops.gemm(a, w, c, m, n, k, k, k, n, 0, 0, bias=bias, epilogue=epilogue | extra | ops.EPI_FORCE_PP, **kw2)
            outs.append((c, kw2.get("aux") if aux == "new" else None, cs))
        torch.cuda.synchronize()
        for x, y in zip(outs[0], outs[1]):
            if x is not None:
                assert torch.equal(x, y), (n, epilogue)

    run(2304, 0, dtype)                                               # qkv forward (+ bias): the product the stagger ships for
    run(3072, ops.EPI_GELU | ops.EPI_AUX_GRAD, dtype, aux="new")      # fc1 + GELU, saved gelu'
    resid = torch.randn(m, 768, device="cuda", generator=g)
    run(768, ops.EPI_RESIDUAL, torch.float32, aux=resid, rowscale=torch.rand(m, device="cuda", generator=g))
    run(768, ops.EPI_RESIDUAL | ops.EPI_TILE_ROWS(208), torch.float32, aux=resid)
    run(3072, 0, dtype, colsum=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_k_tile_choreographies_agree_bit_for_bit(dtype):
    """the two-phase K-tile (32-MFMA clusters; default for the weight-gradient layout) against the four-phase one, built
    as libeditor_gemm_alt.so by `python -m editor_amd.build --alt` (EDITOR_ALT_PHASES=4): a block's grouped weight gradients and a
    forward product, every bit.  Skipped when the alternative library was not built / is older than the kernel source."""
    import ctypes
    import os
    from editor_amd import _lib, build, ops as ops_mod
    if not build.ab_lib_current(build.LIB_ALT):
        pytest.skip("libeditor_gemm_alt.so not built from the current gemm_bf16.hip (python -m editor_amd.build --alt)")
    lib = _lib.lib()
    names = ("editor_gemm_bf16", "editor_gemm_f16", "editor_gemm_wgrad_group")

    def route(alt):
        srcl = ctypes.CDLL(build.LIB_ALT) if alt else lib.cdll
        for name in names:
            fn_ = getattr(srcl, name)
            fn_.argtypes = lib.protos[name]
            fn_.restype = ctypes.c_int
            lib._fn[name] = fn_
    m = 3 * 64 * 129                                                  # 24 768 token rows (the grouped launch reduces over whole 64-row K-tiles)
    g = torch.Generator(device="cuda").manual_seed(5)
    x768 = torch.randn(m, 768, device="cuda", generator=g).to(dtype)
    x3072 = torch.randn(m, 3072, device="cuda", generator=g).to(dtype)
    dy2304 = torch.randn(m, 2304, device="cuda", generator=g).to(dtype)
    dy768 = torch.randn(m, 768, device="cuda", generator=g).to(dtype)
    w = (torch.randn(2304, 768, device="cuda", generator=g) * 0.05).to(dtype)
    res = []
    try:
        for alt in (False, True):
            route(alt)
            dws = [torch.empty(2304, 768, device="cuda"), torch.empty(768, 768, device="cuda"), torch.empty(3072, 768, device="cuda"),
                   torch.empty(768, 3072, device="cuda")]
            ops_mod.gemm_wgrad_group([(dy2304, x768, dws[0]), (dy768, x768, dws[1]), (x3072, x768, dws[2]), (dy768, x3072, dws[3])], m)
            y = torch.empty(m, 2304, dtype=dtype, device="cuda")
            ops_mod.gemm(x768, w, y, m, 2304, 768, 768, 768, 2304, 0, 0, epilogue=ops_mod.EPI_FORCE_PP)
            torch.cuda.synchronize()
            res.append(dws + [y])
    finally:
        route(False)
    for p_, q_ in zip(*res):
        assert torch.equal(p_, q_)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mfma_32x32x16_form_agrees_bit_for_bit(dtype):
    """the v_mfma_f32_32x32x16 form of the ping-pong kernel (EDITOR_PP_MI32, built as libeditor_gemm_mi32.so by
    `python -m editor_amd.build --mi32`; 32-row fragments, (row >> 1) & 7 image swizzle, the one-pass 16-bit staged epilogue at the
    32x32 accumulator coordinates) against the shipped 16x16x32 form on the three products it applies to - qkv forward (+ bias, row
    scale), fc1 + GELU with the saved gelu', fc2 dgrad x gelu' - plus a ragged M / N edge: every bit.  Skipped when that library was
    not built / is older than the kernel source.  (It measures 3 - 12 % slower: profiles/r06_gemm_mi32_ab.txt; it is not shipped.)"""
    import ctypes
    import os
    from editor_amd import _lib, build, ops as ops_mod
    if not build.ab_lib_current(build.LIB_MI32):
        pytest.skip("libeditor_gemm_mi32.so not built from the current gemm_bf16.hip (python -m editor_amd.build --mi32)")
    lib = _lib.lib()
    names = ("editor_gemm_bf16", "editor_gemm_f16")
    keep = {n_: lib._fn.get(n_) for n_ in names}

    def route(alt):
        srcl = ctypes.CDLL(build.LIB_MI32) if alt else lib.cdll
        for name in names:
            fn_ = getattr(srcl, name)
            fn_.argtypes = lib.protos[name]
            fn_.restype = ctypes.c_int
            lib._fn[name] = fn_
    g = torch.Generator(device="cuda").manual_seed(11)
    res = []
    try:
        for alt in (False, True):
            route(alt)
            outs = []
            for m, n, k in ((3 * 32 * 129, 2304, 768), (1000, 776, 768)):          # whole tiles + a ragged edge (M % 256, N % 256 != 0)
                g.manual_seed(11 + m)
                x = torch.randn(m, k, device="cuda", generator=g).to(dtype)
                w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dtype)
                bias = torch.randn(n, device="cuda", generator=g)
                rs = torch.rand(m, device="cuda", generator=g)
                aux_in = torch.randn(m, n, device="cuda", generator=g).to(dtype)
                y0 = torch.zeros(m, n, dtype=dtype, device="cuda")
                ops_mod.gemm(x, w, y0, m, n, k, k, k, n, 0, 0, bias=bias, rowscale=rs, epilogue=ops_mod.EPI_FORCE_PP)
                y1 = torch.zeros(m, n, dtype=dtype, device="cuda")
                a1 = torch.zeros(m, n, dtype=dtype, device="cuda")
                ops_mod.gemm(x, w, y1, m, n, k, k, k, n, 0, 0, bias=bias,
                             epilogue=ops_mod.EPI_GELU | ops_mod.EPI_AUX_GRAD | ops_mod.EPI_FORCE_PP, aux=a1)
                y2 = torch.zeros(m, n, dtype=dtype, device="cuda")
                ops_mod.gemm(x, w, y2, m, n, k, k, k, n, 0, 0,
                             epilogue=ops_mod.EPI_GELU_BWD | ops_mod.EPI_AUX_GRAD | ops_mod.EPI_FORCE_PP, aux=aux_in)
                torch.cuda.synchronize()
                outs += [y0, y1, a1, y2]
            res.append(outs)
    finally:
        for n_, f_ in keep.items():
            if f_ is not None:
                lib._fn[n_] = f_
        route(False)
    for p_, q_ in zip(*res):
        assert p_.float().abs().max() > 0
        assert torch.equal(p_, q_)


@pytest.mark.parametrize("t", [129, 144, 145, 193])
def test_attention_padding_tile_switches_do_not_change_a_bit(t):
    """Round 6: csrc/attention_bf16.hip's two padding-tile switches - ATTN_ROLLOUT_SKIP (shipped: the rollout step ends at the last
    populated query tile, its one-hot first step reads one) and ATTN_PAIR_SKIP (not shipped: the q / kv passes take their last tile pair
    with one tile; measured no gain) - against the build with both flipped (libeditor_attn_alt.so, `python -m editor_amd.build
    --attn-alt`): forward output + log-sum-exps, backward dqkv + its column sums, a three-layer rollout - every bit.  Skipped when that
    library was not built from the current source."""
    import ctypes
    from editor_amd import _lib, build, ops as ops_mod
    if not build.ab_lib_current(build.LIB_ATTN_ALT):
        pytest.skip("libeditor_attn_alt.so not built from the current attention_bf16.hip (python -m editor_amd.build --attn-alt)")
    lib = _lib.lib()
    names = ("editor_attention_fwd_bf16", "editor_attention_bwd_bf16", "editor_attention_bwd_colsum_bf16", "editor_attn_rollout_step_bf16")
    keep = {n_: lib._fn.get(n_) for n_ in names}

    def route(alt):
        srcl = ctypes.CDLL(build.LIB_ATTN_ALT) if alt else lib.cdll
        for name in names:
            fn_ = getattr(srcl, name)
            fn_.argtypes = lib.protos[name]
            fn_.restype = ctypes.c_int
            lib._fn[name] = fn_
    b, heads, hd = 16, 12, 64
    g = torch.Generator(device="cuda").manual_seed(t)
    qkv = (torch.randn(b * t, 3 * heads * hd, device="cuda", generator=g) * 0.7).bfloat16()
    do = torch.randn(b * t, heads * hd, device="cuda", generator=g).bfloat16()
    res = []
    try:
        for alt in (False, True):
            route(alt)
            o, lse = ops_mod.attention_fwd(qkv, b, t, heads, hd, None, None)
            cs = torch.zeros(3 * heads * hd, device="cuda")
            dqkv = ops_mod.attention_bwd(qkv, do, b, t, heads, hd, None, lse, o, colsum=cs)
            roll = ops_mod.attn_rollout_qk([(qkv, lse)] * 3, b, t, heads, hd)
            torch.cuda.synchronize()
            res.append([o, lse, dqkv, cs, roll])
    finally:
        for n_, f_ in keep.items():
            if f_ is not None:
                lib._fn[n_] = f_
        route(False)
    for p_, q_ in zip(*res):
        assert p_.float().abs().max() > 0
        assert torch.equal(p_, q_)

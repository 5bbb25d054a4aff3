"""Split-precision forward (COMPUTE_DTYPE 'f16x2'): the kernels that carry operands as half PAIRS x = hi + lo and form every
product as hi.hi + hi.lo + lo.hi on the half matrix cores, against float64 references of the reference's fp32 operators
(nn.Linear / nn.GELU / softmax attention, vit_pytorch.py:139-145,184-198,240-258).  The bar is fp32-class: the error of
torch's own fp32 CPU matmul against float64 on the same inputs (~3.5e-7) - asserted tolerances are written per test."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _pair(x, scale=1.0):
    from editor_amd import ops
    return ops.split_f32(x.cuda().float(), scale)


def _join(hi, lo):
    return hi.double().cpu() + lo.double().cpu()


def test_split_f32_pair_reconstructs():
    from editor_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1024, 768, generator=g) * torch.logspace(-3, 1, 768)
    hi, lo = ops.split_f32(x.cuda(), 1.0)
    # 2^-22 relative while the low-order half is a normal half number (|x| >= 2^-3); below that it is a SUBNORMAL half with
    # absolute resolution 2^-24 (the matrix core multiplies subnormals exactly: test_probe_mfma_f16_keeps_subnormal_operands)
    err = ((_join(hi, lo) - x.double()).abs() / (x.double().abs() * 2.0 ** -22).clamp_min(2.0 ** -24)).max().item()
    assert err <= 1.0
    w = torch.randn(768, 768, generator=g) * 0.02
    hi, lo = ops.split_f32(w.cuda(), ops.SPLIT_WSCALE)
    assert rel_err(_join(hi, lo) / ops.SPLIT_WSCALE, w.double()) < 1e-7


@pytest.mark.parametrize("m,n,k", [(387, 2304, 768), (4200, 768, 3072), (49536 // 8, 3072, 768), (300, 768, 768)])
def test_gemm_f16x2_vs_float64(m, n, k):
    from editor_amd import ops
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.02
    bias = torch.randn(n, generator=g) * 0.1
    rs = (torch.rand(m, generator=g) > 0.1).float() / 0.9
    res = torch.randn(m, n, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    f32_err = rel_err((a @ w.t() + bias).double(), ref)
    ap, wp = _pair(a), _pair(w, ops.SPLIT_WSCALE)
    # fp32 output + residual epilogue (proj / fc2)
    c = torch.empty(m, n, device="cuda")
    ops.gemm_split(ap, wp, c, None, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias.cuda(), rowscale=rs.cuda(),
                   epilogue=ops.EPI_RESIDUAL, aux=res.cuda())
    ref_res = ref * rs.double()[:, None] + res.double()
    e1 = rel_err(c.double().cpu(), ref_res)
    # pair output (qkv)
    hi = torch.empty(m, n, dtype=torch.float16, device="cuda")
    lo = torch.empty_like(hi)
    ops.gemm_split(ap, wp, hi, lo, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias.cuda())
    e2 = rel_err(_join(hi, lo), ref)
    # GELU pair + gelu' for the backward (fc1)
    aux = torch.empty(m, n, dtype=torch.float16, device="cuda")
    ops.gemm_split(ap, wp, hi, lo, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, bias=bias.cuda(),
                   epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD, aux=aux)
    gref = torch.nn.functional.gelu(ref)
    e3 = rel_err(_join(hi, lo), gref)
    x = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    e4 = rel_err(aux.double().cpu(), x.grad)
    print("gemm_f16x2 %dx%dx%d: residual %.2e pair %.2e gelu %.2e gelu' %.2e | torch fp32 matmul %.2e" % (m, n, k, e1, e2, e3, e4, f32_err))
    # fp32-class: within 2x torch's own fp32 CPU matmul on the same inputs (its error grows with sqrt(K) as well; a pair
    # OUTPUT adds its own 2^-23 representation error on top of the product's)
    assert max(e1, e2, e3) < max(3e-7, 2.0 * f32_err) and e4 < 6e-4


def test_gemm_f16x2_live_rows():
    """m_live (compacted HMA): tiles of dead rows are skipped, live rows exact as above."""
    from editor_amd import ops
    g = torch.Generator().manual_seed(5)
    m, n, k, live = 1024, 768, 768, 517
    a = torch.randn(m, k, generator=g)
    a[live:] = 0
    w = torch.randn(n, k, generator=g) * 0.02
    ap, wp = _pair(a), _pair(w, ops.SPLIT_WSCALE)
    hi = torch.zeros(m, n, dtype=torch.float16, device="cuda")
    lo = torch.zeros_like(hi)
    ops.gemm_split(ap, wp, hi, lo, m, n, k, alpha=1.0 / ops.SPLIT_WSCALE, m_live=torch.tensor([live], dtype=torch.int32, device="cuda"))
    ref = a.double() @ w.double().t()
    assert rel_err(_join(hi, lo)[:live], ref[:live]) < 3e-7


def test_layernorm_split_pair():
    from editor_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(777, 768, generator=g) * 3 + 0.5
    gm, bt = torch.rand(768, generator=g) + 0.5, torch.randn(768, generator=g) * 0.1
    hi, lo, mean, rstd = ops.layernorm_fwd_split(x.cuda(), gm.cuda(), bt.cuda(), 1e-6)
    ref = torch.nn.functional.layer_norm(x.double(), (768,), gm.double(), bt.double(), 1e-6)
    assert rel_err(_join(hi, lo), ref) < 2e-7
    y16, m2, r2 = ops.layernorm_fwd(x.cuda(), gm.cuda(), bt.cuda(), 1e-6, torch.float16)
    assert torch.equal(y16, hi) and torch.equal(m2, mean) and torch.equal(r2, rstd)   # hi half == the f16 mode's operand


def _attn_ref(qkv, b, t, heads, mask=None, lens=None):
    """float64 softmax attention per sequence; returns out (rows, D), probs list."""
    d = qkv.shape[1] // 3
    hd = d // heads
    out = torch.zeros(qkv.shape[0], d, dtype=torch.float64)
    probs = []
    row = 0
    for i in range(b):
        n = t if lens is None else lens[i]
        x = qkv[row:row + n].double()
        q, k, v = (x[:, j * d:(j + 1) * d].view(n, heads, hd).transpose(0, 1) for j in range(3))
        s = q @ k.transpose(1, 2) * hd ** -0.5
        if mask is not None:
            mk = mask[i].bool()
            s = s.masked_fill(~mk[None, None, :], float("-inf"))
        p = torch.softmax(s, dim=-1)
        if mask is not None:
            p = p * mask[i].double()[None, :, None]
        probs.append(p)
        out[row:row + n] = (p @ v).transpose(0, 1).reshape(n, d)
        row += n
    return out, probs


@pytest.mark.parametrize("t", [129, 193, 387, 579])
def test_attention_f16x2_dense_vs_float64(t):
    from editor_amd import ops
    b, heads, hd = 3, 12, 64
    g = torch.Generator().manual_seed(t)
    qkv = torch.randn(b * t, 3 * heads * hd, generator=g) * 0.8
    ref, probs_ref = _attn_ref(qkv, b, t, heads)
    pair = _pair(qkv)
    ldp = (t + 3) // 4 * 4
    probs = torch.zeros(b, heads, t, ldp, device="cuda")
    (oh, ol), lse = ops.attention_fwd_split(pair, b, t, heads, hd, None, probs)
    e_out = rel_err(_join(oh, ol), ref)
    e_p = rel_err(probs[..., :t].double().cpu(), torch.stack(probs_ref))
    # the lse the 16-bit backward consumes: log2 of sum exp2 of the scaled scores
    s = torch.stack([(qkv[i * t:(i + 1) * t, :768].double().view(t, heads, hd).transpose(0, 1) @
                      qkv[i * t:(i + 1) * t, 768:1536].double().view(t, heads, hd).transpose(0, 1).transpose(1, 2)) * hd ** -0.5
                     for i in range(b)])
    lse_ref = torch.logsumexp(s, dim=-1) / torch.log(torch.tensor(2.0, dtype=torch.float64))     # (b, heads, t)
    e_lse = (lse.view(heads, b, t).permute(1, 0, 2).double().cpu() - lse_ref).abs().max().item()
    print("attention f16x2 T=%d: out %.2e probs %.2e lse abs %.2e" % (t, e_out, e_p, e_lse))
    assert e_out < 1e-6 and e_p < 1e-6 and e_lse < 1e-5
    # the hi half is a valid f16-mode output: what the 16-bit attention kernel computes from the hi operands, to half rounding
    o16, _ = ops.attention_fwd(pair[0], b, t, heads, hd)
    assert rel_err(o16.double().cpu(), ref) < 2e-3


@pytest.mark.parametrize("t", [129, 193, 513])
def test_rollout_f16x2_recomputed_from_pairs(t):
    """Round 4: the split-precision rollout recomputes every layer's probabilities from the q / k half pairs and the forward's lse
    (editor_attn_rollout_step_f16x2) instead of reading the materialised fp32 probabilities: both against the float64 rollout
    (SFTS.py:150-153: CLS row of A_{L-1} ... A_0 without the CLS column) - fp32-class, and no worse than the materialised form."""
    from editor_amd import ops
    b, heads, hd, layers = 3, 12, 64, 4
    g = torch.Generator().manual_seed(100 + t)
    ldp = (t + 3) // 4 * 4
    probs = torch.zeros(layers, b, heads, t, ldp, device="cuda")
    triples, refs = [], []
    for l in range(layers):
        qkv = torch.randn(b * t, 3 * heads * hd, generator=g) * 0.8
        pair = _pair(qkv)
        _, lse = ops.attention_fwd_split(pair, b, t, heads, hd, None, probs[l])
        triples.append((pair[0], pair[1], lse))
        refs.append(torch.stack(_attn_ref(qkv, b, t, heads)[1]))           # (b, heads, t, t) float64
    r = refs[-1][:, :, 0:1, :]                                              # CLS row of the last layer
    for l in range(layers - 2, -1, -1):
        r = r @ refs[l]
    ref = r[:, :, 0, 1:]
    got = ops.attn_rollout_qk(triples, b, t, heads, hd)
    mat = ops.attn_rollout(probs)
    assert got.shape == (b, heads, t - 1)
    e_got, e_mat = rel_err(got.double().cpu(), ref), rel_err(mat.double().cpu(), ref)
    print("rollout f16x2 T=%d: recomputed %.2e materialised %.2e" % (t, e_got, e_mat))
    assert e_got < 2e-6 and e_mat < 5e-7        # measured 6.4e-7 / 7.9e-8 at T = 129: one lse rounding per row vs exp2(s - max) / sum
    # a single step, non-final form: r_out (B*heads, T) = CLS row of the last layer's map
    one = torch.empty(b * heads, t, device="cuda")
    ops.call("editor_attn_rollout_step_f16x2", triples[-1][0], triples[-1][1], triples[-1][2], None, b, t, heads, hd, hd ** -0.5,
             one, 0)
    assert rel_err(one.view(b, heads, t).double().cpu(), refs[-1][:, :, 0, :]) < 2e-6


@pytest.mark.parametrize("hd,heads", [(32, 12), (96, 8)])
@pytest.mark.parametrize("t", [129, 193, 387])
def test_attention_f16x2_other_head_widths(hd, heads, t):
    """Round 4: the split-precision attention and rollout kernels at the factory's other head widths (32: DeiT-small's HMA heads,
    96: ViT-small's backbone) - fp32-class against float64, as at 64 columns; T = 193 / 387 take the chunked form at 96 columns
    (four images of the whole sequence do not fit the LDS there)."""
    from editor_amd import ops
    b = 3
    d = heads * hd
    g = torch.Generator().manual_seed(7 * t + hd)
    qkv = torch.randn(b * t, 3 * d, generator=g) * 0.8
    x = qkv.double()
    outs, probs_ref = [], []
    for i in range(b):
        q, k, v = (x[i * t:(i + 1) * t, j * d:(j + 1) * d].view(t, heads, hd).transpose(0, 1) for j in range(3))
        p = torch.softmax(q @ k.transpose(1, 2) * hd ** -0.5, dim=-1)
        probs_ref.append(p)
        outs.append((p @ v).transpose(0, 1).reshape(t, d))
    ref = torch.cat(outs)
    pair = _pair(qkv)
    ldp = (t + 3) // 4 * 4
    probs = torch.zeros(b, heads, t, ldp, device="cuda")
    (oh, ol), lse = ops.attention_fwd_split(pair, b, t, heads, hd, None, probs)
    assert lse.dim() == 1                                     # the split kernel, not the exact-f32 detour
    e_out, e_p = rel_err(_join(oh, ol), ref), rel_err(probs[..., :t].double().cpu(), torch.stack(probs_ref))
    print("attention f16x2 hd=%d T=%d: out %.2e probs %.2e" % (hd, t, e_out, e_p))
    assert e_out < 1e-6 and e_p < 1e-6
    # the 16-bit backward runs on the hi halves with this lse (the f16x2 mode's backward)
    do = torch.randn(b * t, d, generator=g).half().cuda()
    dqkv = ops.attention_bwd(pair[0], do, b, t, heads, hd, None, lse, oh)
    qr = qkv.clone().requires_grad_(True)
    o32 = []
    for i in range(b):
        q, k, v = (qr[i * t:(i + 1) * t, j * d:(j + 1) * d].view(t, heads, hd).transpose(0, 1) for j in range(3))
        o32.append((torch.softmax(q @ k.transpose(1, 2) * hd ** -0.5, dim=-1) @ v).transpose(0, 1).reshape(t, d))
    torch.cat(o32).backward(do.float().cpu())
    assert rel_err(dqkv.float().cpu(), qr.grad) < 5e-3
    if t <= 416:
        r = probs_ref_roll = torch.stack(probs_ref)[:, :, 0:1, :]
        got = ops.attn_rollout_qk([(pair[0], pair[1], lse)], b, t, heads, hd)
        assert rel_err(got.double().cpu(), r[:, :, 0, 1:]) < 2e-6


def test_attention_f16x2_masked_and_varlen():
    from editor_amd import ops
    b, t, heads, hd = 4, 129, 12, 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(b * t, 3 * heads * hd, generator=g) * 0.8
    mask = (torch.rand(b, t, generator=g) > 0.6).to(torch.uint8)
    mask[:, 0] = 1
    ref, _ = _attn_ref(qkv, b, t, heads, mask=mask)
    (oh, ol), lse = ops.attention_fwd_split(_pair(qkv), b, t, heads, hd, mask.cuda())
    assert rel_err(_join(oh, ol), ref) < 5e-7
    # packed sequences of different lengths (compacted HMA form), one of them long enough for the streamed kernel when alone
    lens = [37, 129, 5, 100]
    rows = sum(lens)
    pad = (rows + 63) // 64 * 64
    q2 = torch.zeros(pad, 3 * heads * hd)
    q2[:rows] = torch.randn(rows, 3 * heads * hd, generator=g) * 0.8
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ref2, _ = _attn_ref(q2[:rows], len(lens), max(lens), heads, lens=lens)
    (oh, ol), lse = ops.attention_fwd_split(_pair(q2), len(lens), max(lens), heads, hd, None, None, cu=cu.cuda())
    assert rel_err(_join(oh, ol)[:rows], ref2) < 5e-7
    assert float(oh[rows:].abs().max()) == 0.0
    lens = [300, 64, 387]
    rows = sum(lens)
    pad = (rows + 63) // 64 * 64
    q3 = torch.zeros(pad, 3 * heads * hd)
    q3[:rows] = torch.randn(rows, 3 * heads * hd, generator=g) * 0.8
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ref3, _ = _attn_ref(q3[:rows], len(lens), max(lens), heads, lens=lens)
    (oh, ol), lse = ops.attention_fwd_split(_pair(q3), len(lens), max(lens), heads, hd, None, None, cu=cu.cuda())
    assert rel_err(_join(oh, ol)[:rows], ref3) < 5e-7


def test_fused_sgd_refreshes_split_pairs():
    from editor_amd import functional as fn, ops
    from editor_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(4)
    ps = [torch.nn.Parameter((torch.randn(768, 768, generator=g) * 0.02).cuda()), torch.nn.Parameter(torch.zeros(768).cuda())]
    opt = FusedSGD(list(zip(["w.weight", "w.bias"], ps)), base_lr=1e-2, momentum=0.9, shadow_dtype=torch.float16, split_pairs=True)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda()
    opt.step()
    hi, lo = fn.act_weight_split(ps[0])
    assert hi.data_ptr() == opt.pairs[0][0].data_ptr()                     # served from the optimizer's own launch
    assert rel_err(_join(hi, lo) / ops.SPLIT_WSCALE, ps[0].detach().double().cpu()) < 1e-7

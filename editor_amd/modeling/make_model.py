"""Drop-in counterpart of the reference's `modeling.make_model` for the MI355X-native hot path.

Same factory, same forward signature and return tuples, same 216(+6 with AL) state-dict keys as
/root/reference/modeling/make_model.py:86-258,371-374 - so engine/processor.py, tools/train.py, the
optimizer's name-based parameter groups (solver/make_optimizer.py:6-19) and checkpoints work unchanged.
The nn.Module tree only CONTAINS parameters; every forward/backward computation is a HIP kernel launched
through libeditor_hip.so (editor_amd.functional).  There is no PyTorch/CPU fallback: calling forward on a
CPU tensor raises.

MI355X-first differences in HOW (not WHAT) it computes:
  * the shared backbone runs ONCE on the three modalities stacked on the batch axis (3B samples), so every
    GEMM sees M = 3*B*T rows and the shared weights stream from HBM once;
  * the attention rollout is a row-vector recurrence (no 129x129 matmul chain): the 16-bit modes recompute each layer's
    probabilities from its saved q / k and row log-sum-exps (no probability tensor at all), the f32 parity mode and the
    split-precision 'f16x2' mode read the (L,3B,h,T,T) softmax buffer their attention kernels materialise;
  * top-k tie order of torch's CPU kernel is reproduced on device (libstdc++ heap/introselect).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as fn
from .. import ops

_ARCH = {
    # name: (embed_dim, depth, heads, mlp_ratio, qkv_bias, qk_scale)   vit_pytorch.py:693-727
    "vit_base_patch16_224": (768, 12, 12, 4.0, True, None),
    "deit_base_patch16_224": (768, 12, 12, 4.0, True, None),
    "deit_small_patch16_224": (384, 12, 6, 4.0, True, None),
    "vit_small_patch16_224": (768, 8, 8, 3.0, False, 768 ** -0.5),        # vit_pytorch.py:712: qk_scale=768 ** -0.5
    # extension (BASELINE.json config 5): not in the reference's factory (make_model.py:363-368)
    "vit_large_patch16_224": (1024, 24, 16, 4.0, True, None),
}

# (input dict key, name of the REDUCE layer / OCFR centre table, BlockMask tag): the reference's three modalities
# (make_model.py:153-155,106-108; vit_pytorch.py:268-290; OCFR.py:16-18) and the 4th of the synthetic 4-modal
# configuration (BASELINE.json config 5) - an extension, the reference's forward hard-codes three keys
_MODALITIES = (("RGB", "RGB", "R"), ("NI", "NIR", "N"), ("TI", "TIR", "T"), ("M4", "M4", "M4"))


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def _act_dtype(cfg):
    name = getattr(cfg.MODEL, "COMPUTE_DTYPE", "bf16")
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    if name in ("f16", "fp16", "float16", "half", "f16x2"):     # 'f16x2': f16 tensors, split-precision forward (EDITOR.split_fwd)
        return torch.float16
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError("cfg.MODEL.COMPUTE_DTYPE must be 'bf16', 'f16', 'f16x2' or 'f32', got %r" % (name,))


# ------------------------------------------------------------------------------------------------
# parameter containers (names == reference state-dict keys)
# ------------------------------------------------------------------------------------------------
class _Attn(nn.Module):
    def __init__(self, dim, bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=bias)
        self.proj = nn.Linear(dim, dim, bias=bias)


class _BackboneAttn(nn.Module):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)                      # vit_pytorch.py:181: proj always has a bias


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=bias)
        self.fc2 = nn.Linear(hidden, dim, bias=bias)


class _Block(nn.Module):
    def __init__(self, dim, hidden, qkv_bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _BackboneAttn(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden)


class _PatchEmbed(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=16, stride=16)


def _init_linear_ln(mod):
    for m in mod.modules():
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, 0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class Trans(nn.Module):
    """Parameter layout of the reference's ViT `Trans` (vit_pytorch.py:461-534)."""

    def __init__(self, img_size, embed_dim, depth, heads, mlp_ratio, qkv_bias, camera, sie_xishu, drop_path_rate,
                 qk_scale=None):
        super().__init__()
        self.embed_dim, self.depth, self.heads = embed_dim, depth, heads
        self.qk_scale = qk_scale
        self.num_y, self.num_x = img_size[0] // 16, img_size[1] // 16
        self.num_patches = self.num_y * self.num_x
        self.img_size = tuple(img_size)
        self.patch_embed = _PatchEmbed(embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim))
        self.cam_num = camera
        self.sie_xishu = sie_xishu
        if camera > 1:
            self.sie_embed = nn.Parameter(torch.zeros(camera, 1, embed_dim))
            _trunc_normal_(self.sie_embed, 0.02)
        self.drop_rates = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # vit_pytorch.py:511
        self.blocks = nn.ModuleList([_Block(embed_dim, int(embed_dim * mlp_ratio), qkv_bias, 1e-6)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc = nn.Linear(embed_dim, 1000)                 # unused in forward, kept for key compatibility
        _trunc_normal_(self.cls_token, 0.02)
        _trunc_normal_(self.pos_embed, 0.02)
        _init_linear_ln(self)
        w = self.patch_embed.proj.weight                     # vit_pytorch.py:439-442
        w.data.normal_(0, math.sqrt(2.0 / (w.shape[2] * w.shape[3] * w.shape[0])))

    def load_param(self, model_path):
        """ImageNet checkpoint loading (vit_pytorch.py:646-671): skip head/dist, resize pos_embed."""
        param_dict = torch.load(model_path, map_location="cpu")
        for key in ("model", "state_dict"):
            if key in param_dict:
                param_dict = param_dict[key]
        sd = self.state_dict()
        for k, v in param_dict.items():
            if "head" in k or "dist" in k:
                continue
            if "patch_embed.proj.weight" in k and v.dim() < 4:
                v = v.reshape(self.patch_embed.proj.weight.shape[0], -1, 16, 16)
            elif k == "pos_embed" and v.shape != self.pos_embed.shape:
                if "distilled" in model_path:
                    v = torch.cat([v[:, 0:1], v[:, 2:]], dim=1)
                v = resize_pos_embed(v, self.num_y, self.num_x)
            try:
                sd[k].copy_(v)
            except Exception:                                # the reference prints and continues (:665-671)
                print("shape do not match in k :{}".format(k))


def resize_pos_embed(posemb, height, width):
    """vit_pytorch.py:674-690: bilinear resize of the grid part of a (1, 1+g*g, D) position table."""
    tok, grid = posemb[:, :1], posemb[0, 1:]
    gs = int(math.sqrt(len(grid)))
    grid = grid.reshape(1, gs, gs, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(height, width), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, height * width, -1)
    return torch.cat([tok, grid], dim=1)


class build_transformer(nn.Module):
    """make_model.py:35-82 (holds `.base`)."""

    def __init__(self, num_classes, cfg, camera_num):
        super().__init__()
        ttype = cfg.MODEL.TRANSFORMER_TYPE
        dim, depth, heads, mlp_ratio, qkv_bias, qk_scale = _ARCH[ttype]
        self.token_dim = dim
        cams = camera_num if cfg.MODEL.SIE_CAMERA else 0
        self.base = Trans(cfg.INPUT.SIZE_TRAIN, dim, depth, heads, mlp_ratio, qkv_bias, cams, cfg.MODEL.SIE_COE,
                          cfg.MODEL.DROP_PATH, qk_scale)
        if cfg.MODEL.PRETRAIN_CHOICE == "imagenet":
            self.base.load_param(cfg.MODEL.PRETRAIN_PATH_T)


class _Wavelet(nn.Module):
    def __init__(self, names):
        super().__init__()
        s = 1.0 / math.sqrt(2.0)
        taps = {"h0": [s, s], "h1": [s, -s], "g0": [s, s], "g1": [s, -s]}   # reversed dec_* / rec_* (lowlevel.py:970-974,916-920)
        for n in names:
            shape = (1, 1, 2, 1) if n.endswith("col") else (1, 1, 1, 2)
            self.register_buffer(n, torch.tensor(taps[n[:2]], dtype=torch.float32).reshape(shape))


class FrequencyIndex(nn.Module):
    """Buffer layout of Frequency_based_Token_Selection (Frequency.py:10-18); compute is in HIP."""

    def __init__(self, keep, stride=16):
        super().__init__()
        self.DWT = _Wavelet(["h0_col", "h1_col", "h0_row", "h1_row"])
        self.IDWT = _Wavelet(["g0_col", "g1_col", "g0_row", "g1_row"])
        self.keep = int(keep)
        self.stride = stride
        if stride != 16:
            raise NotImplementedError("the HIP frequency kernel tiles 16x16 windows (STRIDE_SIZE 16)")

    def forward(self, x, y, z=None, w=None, **_):
        mask, _ = ops.frequency_mask(x, y, z, self.keep, w)
        return mask.bool()


class OCFRCenters(nn.Module):
    def __init__(self, dim, num_class, names=("RGB", "NIR", "TIR")):
        super().__init__()
        for n in names:
            setattr(self, n + "_centers", nn.Parameter(torch.zeros(num_class, dim), requires_grad=False))


class BlockMask(nn.Module):
    """Parameter layout of the HMA head (vit_pytorch.py:261-307)."""

    def __init__(self, dim, num_class, mlp_ratio=4.0, momentum=0.8, modalities=_MODALITIES[:3]):
        super().__init__()
        hidden = int(dim * mlp_ratio)
        for tag in [m[2] for m in modalities]:
            setattr(self, "norm" + tag, nn.LayerNorm(dim))
            setattr(self, "attn" + tag, _Attn(dim, False))
            setattr(self, "norm" + tag + "_", nn.LayerNorm(dim))
            setattr(self, "mlp" + tag, _Mlp(dim, hidden, False))
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = _Attn(dim, False)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, hidden, False)
        self.out_norm = nn.LayerNorm(dim)
        self.memory_cls = OCFRCenters(dim, num_class, [m[1] for m in modalities])
        self.momentum = momentum
        _init_linear_ln(self)


def _block_args(norm1, attn, norm2, mlp):
    return (norm1.weight, norm1.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias,
            norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias)


# ------------------------------------------------------------------------------------------------
# EDITOR
# ------------------------------------------------------------------------------------------------
class EDITOR(nn.Module):
    def __init__(self, num_classes, cfg, camera_num):
        super().__init__()
        self.BACKBONE = build_transformer(num_classes, cfg, camera_num)
        dim = self.BACKBONE.token_dim
        self.num_patches = (cfg.INPUT.SIZE_TRAIN[0] // cfg.MODEL.STRIDE_SIZE[0]) * \
                           (cfg.INPUT.SIZE_TRAIN[1] // cfg.MODEL.STRIDE_SIZE[1])
        self.ratio = (1 / self.num_patches) * int(cfg.MODEL.HEAD_KEEP)           # make_model.py:92
        self.head_k = int(self.num_patches * self.ratio)                        # SFTS.py:155
        self.FREQ_INDEX = FrequencyIndex(cfg.MODEL.FREQUENCY_KEEP, cfg.MODEL.STRIDE_SIZE[0])
        self.hma_heads = getattr(cfg.MODEL, "HMA_HEADS", 12 if dim % 12 == 0 else 16)    # make_model.py:97
        nmod = int(getattr(cfg.MODEL, "NUM_MODALITIES", 3))
        if nmod not in (3, 4):
            raise NotImplementedError("EDITOR fuses 3 modalities (make_model.py:153-155); NUM_MODALITIES = 4 is the "
                                      "synthetic extension of BASELINE config 5")
        self.modalities = _MODALITIES[:nmod]
        self.nmod = nmod
        self.FUSE_block = BlockMask(dim, num_classes, 4.0, 0.8, self.modalities)
        for tag in [m[1] for m in self.modalities]:
            lin = nn.Linear(2 * dim, dim)
            nn.init.kaiming_normal_(lin.weight, a=0, mode="fan_out")            # make_model.py:10-14
            nn.init.constant_(lin.bias, 0.0)
            setattr(self, tag + "_REDUCE", lin)
        self.FUSE_HEAD = nn.Linear(nmod * dim, num_classes, bias=False)
        self.FUSE_BN = nn.BatchNorm1d(nmod * dim)
        nn.init.normal_(self.FUSE_HEAD.weight, std=0.001)                       # make_model.py:26-31
        self.BACKBONE_HEAD = nn.Linear(dim, num_classes, bias=False)
        self.BACKBONE_BN = nn.BatchNorm1d(dim)
        nn.init.normal_(self.BACKBONE_HEAD.weight, std=0.001)
        self.AL = cfg.MODEL.AL
        if self.AL:
            self.AL_HEAD = nn.Linear(nmod * dim, num_classes, bias=False)
            self.AL_BN = nn.BatchNorm1d(nmod * dim)
            nn.init.normal_(self.AL_HEAD.weight, std=0.001)
        self.act_dtype = _act_dtype(cfg)
        # 'f16x2': the forward runs every product on split-precision half pairs (hi.hi + hi.lo + lo.hi on the half matrix
        # cores, fp32-class: token selection as the f32 parity mode, at ~2/3 of the f16 mode's speed); the backward is f16's
        self.split_fwd = getattr(cfg.MODEL, "COMPUTE_DTYPE", "bf16") == "f16x2"
        self.fn_dtype = fn.F16X2 if self.split_fwd else self.act_dtype       # what the autograd nodes are told
        # cfg.MODEL.SPLIT_SCOPE (f16x2 only): 'all' = every forward product on half pairs (features fp32-class, 9e-7);
        # 'selection' = only what the token selection depends on - backbone blocks 0 .. L-2 whole, of the last block LayerNorm-1,
        # the qkv product and the attention core (its MAP enters the rollout, SFTS.py:145-153; its output does not) - and the
        # rest (last block's projection + MLP, the whole HMA head) as the f16 mode computes it: selection still bit-identical
        # to the reference, features within the north star's 1e-3 like the f16 mode's, ~3.5 ms per step cheaper
        scope = str(getattr(cfg.MODEL, "SPLIT_SCOPE", "all"))
        if scope not in ("all", "selection"):
            raise ValueError("cfg.MODEL.SPLIT_SCOPE must be 'all' or 'selection'")
        self.split_selection_only = self.split_fwd and scope == "selection"
        self.fn_dtype_last = fn.F16X2H if self.split_selection_only else self.fn_dtype     # last backbone block
        self.fn_dtype_hma = self.act_dtype if self.split_selection_only else self.fn_dtype  # HMA head
        base = self.BACKBONE.base
        # the fused 16-bit and split-precision attention / rollout kernels are built for 32-, 64- and 96-wide heads (round 4:
        # ViT-small's 96 and the 32-wide HMA heads of DeiT-small used to take the detour below).  Any other width keeps its 16-bit
        # GEMMs and runs the attention product itself on the exact-f32 kernels between two casts (ops.attention_fwd): the
        # probabilities are then materialised for the rollout as in the f32 mode, and the HMA head takes its dense-masked form
        self.bb_attn_f32 = self.act_dtype != torch.float32 and dim // base.heads not in ops.ATTN_HEAD_WIDTHS
        self.hma_attn_f32 = self.act_dtype != torch.float32 and dim // self.hma_heads not in ops.ATTN_HEAD_WIDTHS
        # per-model options of the autograd nodes (installed at the top of every forward, captured by the nodes' ctx)
        self.grad_scale_f16 = float(cfg.MODEL.GRAD_SCALE) if hasattr(cfg.MODEL, "GRAD_SCALE") else None
        self.act_light = bool(getattr(cfg.MODEL, "ACT_LIGHT", False))   # 24 instead of 36 saved bytes per token-row-element
        fn.set_model_options(self.grad_scale_f16, self.act_light)
        self.hma_compact = bool(getattr(cfg.MODEL, "HMA_COMPACT", True))
        # cfg.MODEL.BRANCH16 (default OFF; EDITOR_BRANCH16=1 switches it on for A/B runs): the backbone blocks' projection / fc2
        # products write their branch output in 16 bits with the plain epilogue and the residual add happens inside the LayerNorm
        # that follows (functional.TransformerBlockFn branch16 / defer_out).  Measured in the step: the GEMM
        # family gets 1.0 ms shorter (0.357 -> 0.369 of peak: the fp32 residual epilogues leave it) and the LayerNorm family 1.0
        # ms longer - in situ the stand-alone LayerNorm reads rows the epilogue has just left in the Infinity Cache, the fused one
        # reads the residual from HBM - so the STEP does not move (42.57 / 42.35 vs 42.57 / 42.47 ms), while every branch is
        # rounded to 16 bits before the add (cls4t x 1.068).  Hence an option, not the default.
        b16_default = os.environ.get("EDITOR_BRANCH16") == "1" and self.act_dtype == torch.bfloat16 and not self.split_fwd
        self.branch16 = bool(getattr(cfg.MODEL, "BRANCH16", b16_default)) and self.act_dtype in (torch.bfloat16, torch.float16) \
            and not self.split_fwd and not self.bb_attn_f32 and dim % 256 == 0
        # cfg.MODEL.DROP_SKIP (default on; EDITOR_DROP_SKIP=0 for A/B runs): stochastic depth skips what it multiplies by zero - the
        # MLP branch of a block runs on the samples its draw kept (functional.TransformerBlockFn drop_plan); same bits per live row
        self.drop_skip = bool(getattr(cfg.MODEL, "DROP_SKIP", os.environ.get("EDITOR_DROP_SKIP", "1") != "0"))
        self.rollout_probs = bool(getattr(cfg.MODEL, "ROLLOUT_PROBS", False))
        self.split_rollout_recompute = bool(getattr(cfg.MODEL, "SPLIT_ROLLOUT_RECOMPUTE",
                                                    os.environ.get("EDITOR_SPLIT_ROLLOUT") == "1"))
        self.teacher_index = None            # optional (B,N) bool: force the SFTS selection (bf16 protocol)
        # optional (nmod, depth, 2, B) 0/1: force the stochastic-depth keep masks (parity tests against the oracle / the reference's
        # recorded torch.rand draws: [modality, block, branch (0 attention, 1 MLP), sample], vit_pytorch.py:64-68,217-218)
        self.teacher_drop_keep = None
        self.grad_buckets = None             # editor_amd.ddp.GradBuckets once enable_grad_buckets() was called
        self._drop_rates_dev = None
        self._drop_state = None
        self.last_drop_scales = None
        self.last_aux = {}

    # -- checkpoint compatibility (make_model.py:144-148) ---------------------------------------
    def load_param(self, trained_path):
        param_dict = torch.load(trained_path, map_location="cpu")
        sd = self.state_dict()
        for k in param_dict:
            sd[k.replace("module.", "")].copy_(param_dict[k])
        print("Loading pretrained model from {}".format(trained_path))

    # -- data-parallel gradient buckets (editor_amd.ddp.GradBuckets) ------------------------------------------------
    def grad_segments(self):
        """Transformer blocks in the order their gradients become READY in the backward: joint HMA block, the per-modality
        HMA blocks (last modality first), backbone blocks depth-1 .. 0; plus the remaining parameters (tail)."""
        fb = self.FUSE_block
        segs = [("hma.joint", list(_block_args(fb.norm1, fb.attn1, fb.norm2, fb.mlp)))]
        for m_ in reversed(self.modalities):
            tag = m_[2]
            segs.append(("hma." + tag, list(_block_args(getattr(fb, "norm" + tag), getattr(fb, "attn" + tag),
                                                        getattr(fb, "norm" + tag + "_"), getattr(fb, "mlp" + tag)))))
        blocks = self.BACKBONE.base.blocks
        for i in reversed(range(len(blocks))):
            blk = blocks[i]
            segs.append(("backbone.%d" % i, list(_block_args(blk.norm1, blk.attn, blk.norm2, blk.mlp))))
        inseg = {id(p) for _, ps in segs for p in ps if p is not None}
        tail = [p for p in self.parameters() if id(p) not in inseg]
        return segs, tail

    def enable_grad_buckets(self, bucket_bytes=64 << 20, process_group=None, force=False, wire_dtype=None):
        from ..ddp import GradBuckets
        segs, tail = self.grad_segments()
        self.grad_buckets = GradBuckets(segs, tail, bucket_bytes, process_group, force, wire_dtype)
        self._seg_index = {name: i for i, (name, _) in enumerate(segs)}
        return self.grad_buckets

    def _sink(self, name):
        if self.grad_buckets is None or not torch.is_grad_enabled():
            return None
        return self.grad_buckets.sink(self._seg_index[name])

    # -- stages ------------------------------------------------------------------------------------
    def _backbone(self, imgs, cam):
        """Trans.forward (vit_pytorch.py:623-644) on the modalities' (B,3,H,W) batches as if stacked to (3B,3,H,W) (the
        patch embedding writes their im2col rows side by side; no stacking copy).  Returns final-LN tokens (3B,T,D) fp32
        and the (L,3B,h,T,T) softmax buffer."""
        base = self.BACKBONE.base
        dev = imgs[0].device
        for im in imgs:
            if tuple(im.shape[-2:]) != base.img_size:
                raise AssertionError(f"Input image size ({im.shape[-2]}*{im.shape[-1]}) doesn't match model "
                                     f"({base.img_size[0]}*{base.img_size[1]}).")
        btot = sum(im.shape[0] for im in imgs)
        t = base.num_patches + 1
        sie = base.sie_embed if base.cam_num > 1 else None
        x = fn.PatchEmbedFn.apply(imgs, base.patch_embed.proj.weight, base.patch_embed.proj.bias, base.cls_token,
                                  base.pos_embed, sie, cam if sie is not None else None, float(base.sie_xishu),
                                  self.fn_dtype)
        # softmax outputs of every layer; rows padded to a multiple of 4 floats in bf16 mode (16-byte stores)
        # f32 parity mode: the (L,3B,h,T,T) softmax outputs are materialised as the reference does.  bf16 mode: every
        # block hands back its (qkv, row log-sum-exp) instead and the rollout recomputes the probabilities from them
        # (cfg.MODEL.ROLLOUT_PROBS = True keeps the materialised form: rows padded to a multiple of 4 floats).
        # (split-precision mode: the fp32 probabilities of the split attention kernel are materialised, as in f32 mode.  Round 4
        # built the recomputing form for it too - editor_attn_rollout_step_f16x2, scores from the q / k half PAIRS in three passes -
        # and measured it in the step: 61.9 / 65.1 ms against 61.1 - 62.0 / 64.7 - 65.0, i.e. nothing (a step then reads 304 MB of
        # operand pairs where the probabilities were 314 MB), and its scores are 8x less accurate (6e-7 against 8e-8: one lse
        # rounding per row).  cfg.MODEL.SPLIT_ROLLOUT_RECOMPUTE / EDITOR_SPLIT_ROLLOUT=1 selects it: 1.1 GB less at B = 128.)
        recompute = self.act_dtype != torch.float32 and not self.rollout_probs and not self.bb_attn_f32 and \
            (not self.split_fwd or self.split_rollout_recompute)
        ldp = t if (self.act_dtype == torch.float32 or self.bb_attn_f32) else (t + 3) // 4 * 4
        probs = [] if recompute else torch.empty(base.depth, btot, base.heads, t, ldp, dtype=torch.float32,
                                                 device=dev)
        scales = plans = None
        if self.training and max(base.drop_rates) > 0.0:                       # vit_pytorch.py:52-69: one launch for all
            if self._drop_rates_dev is None or self._drop_rates_dev.device != dev:
                self._drop_rates_dev = torch.tensor(base.drop_rates, dtype=torch.float32, device=dev)
            if self.teacher_drop_keep is not None:
                keep = self.teacher_drop_keep.to(dev).float()                   # (nmod, depth, 2, B)
                if tuple(keep.shape) != (len(imgs), base.depth, 2, btot // len(imgs)):
                    raise ValueError("teacher_drop_keep must be (nmod, depth, 2, B)")
                keep = keep.permute(1, 2, 0, 3).reshape(base.depth, 2, btot)    # the modalities are stacked on the batch axis
                per_sample = keep / (1.0 - self._drop_rates_dev).view(-1, 1, 1) # what droppath_kernel writes: floor(kp + u) / kp
                scales = per_sample.unsqueeze(-1).expand(-1, -1, -1, t).reshape(base.depth, 2, btot * t).contiguous()
            else:
                if self._drop_state is None or self._drop_state.device != dev:  # device-resident RNG counter
                    seed0 = (int(torch.initial_seed()) * 1000003) & 0x3FFFFFFFFFFFFFFF
                    self._drop_state = torch.full((1,), seed0, dtype=torch.int64, device=dev)
                scales = ops.droppath_scales_dev(self._drop_rates_dev, btot, t, self._drop_state)
            self.last_drop_scales = scales             # (depth, 2, nmod*B*T) per-row branch scales of this forward (tests read them)
            if self.drop_skip and self.act_dtype in ops.HALF_DTYPES and not self.branch16:
                # round 6: the samples a branch's draw dropped are not computed (functional.TransformerBlockFn `drop_plan`):
                # token rows of every (block, branch) ordered live samples first, one small launch for the whole backbone
                plans = ops.droppath_plan(scales, base.depth, btot, t)
        pend_branch = pend_rs = None                 # BRANCH16: the previous block's (fc2 branch, drop-path scales); x is then its x1
        for i, blk in enumerate(base.blocks):
            rs_a = rs_m = plan = None
            if scales is not None and base.drop_rates[i] > 0.0:
                rs_a, rs_m = scales[i, 0], scales[i, 1]
                if plans is not None:
                    plan = (plans[0][i, 1], plans[1][i, 1], plans[2][i, 1:2])       # MLP branch: (perm, inv, live rows)
            last = i == len(base.blocks) - 1
            defer = self.branch16 and not last        # (the last block adds its own fc2 branch: the final norm takes plain rows)
            out = fn.TransformerBlockFn.apply(x, *_block_args(blk.norm1, blk.attn, blk.norm2, blk.mlp), None,
                                              probs if recompute else probs[i], base.heads, 1e-6,
                                              self.fn_dtype_last if last else self.fn_dtype, rs_a, rs_m,
                                              None, None, None, base.qk_scale, self._sink("backbone.%d" % i),
                                              pend_branch, pend_rs, defer, self.branch16, plan)
            if defer:
                x, pend_branch = out
                pend_rs = rs_m
            else:
                x, pend_branch, pend_rs = out, None, None
        x = fn.LayerNormFn.apply(x, base.norm.weight, base.norm.bias, 1e-6, None)
        return x, probs

    def _select(self, probs, mask_fre, b):
        """Part_Attention x3 + union with the frequency mask (SFTS.py:145-164,183-187) -> (B,N) uint8."""
        if isinstance(probs, list):                                             # bf16: per-layer (qkv, lse)
            base = self.BACKBONE.base
            h, t = base.heads, base.num_patches + 1
            btot = probs[0][0].shape[0] // t
            scores = ops.attn_rollout_qk(probs, btot, t, h, probs[0][0].shape[1] // (3 * h), base.qk_scale)
        else:
            l, btot, h, t = probs.shape[:4]
            scores = ops.attn_rollout(probs)                                    # (3B, h, N)
        m = ops.topk_mask(scores.view(btot * h, t - 1), self.head_k, group=h)    # (3B, N)
        nmod = btot // b
        index = ops.mask_or(m[:b], m[b:2 * b], m[2 * b:3 * b] if nmod > 2 else None, mask_fre)
        if nmod > 3:
            index = ops.mask_or(index, m[3 * b:4 * b])
        self.last_aux = {"scores": scores, "attn_masks": m.view(nmod, b, t - 1), "mask_fre": mask_fre, "index": index}
        return index

    def _hma(self, feats_s, index, label):
        """BlockMask.forward (vit_pytorch.py:309-352): feats_s (3,B,T,D) fp32 -> fused (B,3T,D), loss_ocfr."""
        fb = self.FUSE_block
        nmod, b, t, d = feats_s.shape
        mask = torch.cat([torch.ones(b, 1, dtype=torch.uint8, device=index.device), index], dim=1).contiguous()
        mods = []
        feats_mod = feats_s.unbind(0)                  # (unbind's backward is one stack; per-index selects zero-fill and add)
        for i, tag in enumerate(m_[2] for m_ in self.modalities):
            args = _block_args(getattr(fb, "norm" + tag), getattr(fb, "attn" + tag), getattr(fb, "norm" + tag + "_"),
                               getattr(fb, "mlp" + tag))
            mods.append(fn.TransformerBlockFn.apply(feats_mod[i], *args, mask, None, self.hma_heads, 1e-5,
                                                    self.fn_dtype_hma, None, None, None, None, None, None,
                                                    self._sink("hma." + tag)))
        loss_ocfr = None
        if self.training:
            loss_ocfr = self._ocfr([m_[:, 0] for m_ in mods], label)
        x = torch.cat(mods, dim=1)
        mask3 = mask.repeat(1, nmod).contiguous()
        x = fn.TransformerBlockFn.apply(x, *_block_args(fb.norm1, fb.attn1, fb.norm2, fb.mlp), mask3, None,
                                        self.hma_heads, 1e-5, self.fn_dtype_hma, None, None, None, None, None, None,
                                        self._sink("hma.joint"))
        x = fn.LayerNormFn.apply(x, fb.out_norm.weight, fb.out_norm.bias, 1e-5, mask3.view(-1))
        return x, loss_ocfr

    def _hma_compact(self, feats_s, index, label):
        """BlockMask.forward on the kept tokens only (SURVEY.md 5 "HMA exact-zero invariant"): the same four blocks on
        packed variable-length sequences.  Returns pooled (3,B,2D), num (B) and loss_ocfr."""
        fb = self.FUSE_block
        nmod, b, t, d = feats_s.shape
        plan = ops.CompactPlan(index, t, nmod)
        # (feats_s has this ONE consumer and SFTSApplyFn.backward reads the selected rows of its gradient only - exactly the rows
        #  this gather names - so the gather's backward does not zero-fill the others: bwd_fill="none")
        xa = fn.GatherRowsFn.apply(feats_s.reshape(nmod * b * t, d), plan.map_a, plan.live_a, 1, plan.ma, "none")   # layout A
        mods = []
        xa_mod = torch.split(xa, plan.ma, dim=0)       # (split's backward is one cat; slices would zero-fill and add)
        flat = []
        for i, tag in enumerate(m_[2] for m_ in self.modalities):
            args = _block_args(getattr(fb, "norm" + tag), getattr(fb, "attn" + tag), getattr(fb, "norm" + tag + "_"),
                               getattr(fb, "mlp" + tag))
            flat.append((xa_mod[i], *args, plan.mask_a, None, self.hma_heads, 1e-5, self.fn_dtype_hma, None, None, plan.cu, t,
                         plan.live_a, None, self._sink("hma." + tag)))
        if fn.GROUP_BLOCKS and self.fn_dtype_hma in ops.HALF_DTYPES and not self.act_light:
            # the per-modality blocks are identical in shape and independent: one node, their products as grouped launches
            # (round 4: ~7 400 live rows per modality are 87 tiles of a 768-wide product - a third of a round of the 256 CUs)
            mods = list(fn.GroupedBlocksFn.apply(len(flat), *[v for blk in flat for v in blk]))
        else:
            mods = [fn.TransformerBlockFn.apply(*blk) for blk in flat]
        xa = torch.cat(mods, dim=0)
        loss_ocfr = None
        if self.training:
            cls, xb = fn.GatherPairFn.apply(xa, plan.map_cls, plan.map_b, plan.live_a, nmod, plan.mb)   # cls rows; layout B (MB, D)
            loss_ocfr = self._ocfr(list(cls.view(nmod, b, d).unbind(0)), label)
        else:
            xb = fn.GatherRowsFn.apply(xa, plan.map_b, plan.live_a, nmod, plan.mb)
        xb = fn.TransformerBlockFn.apply(xb, *_block_args(fb.norm1, fb.attn1, fb.norm2, fb.mlp), plan.mask_b, None,
                                         self.hma_heads, 1e-5, self.fn_dtype_hma, None, None, plan.cu3, nmod * t, plan.live_b,
                                         None, self._sink("hma.joint"))
        xb = fn.LayerNormFn.apply(xb, fb.out_norm.weight, fb.out_norm.bias, 1e-5, plan.mask_b, plan.live_b)
        pooled, num = fn.PoolPackedFn.apply(xb, plan.cu, b, nmod, plan.live_b)
        self.last_aux["plan"] = plan
        return pooled, num, loss_ocfr

    def _ocfr(self, cls_feats, label):
        """OCFR.forward (OCFR.py:44-84): normalise, per-label centre update (momentum 0.8), MSE to own-class centre."""
        mc = self.FUSE_block.memory_cls
        centers = [getattr(mc, m_[1] + "_centers") for m_ in self.modalities]
        return fn.OCFRFn.apply(label.contiguous(), float(self.FUSE_block.momentum), len(cls_feats), *cls_feats, *centers)

    # -- forward (make_model.py:150-258) ----------------------------------------------------------
    def forward(self, x, cam_label=None, label=None, view_label=None, img_path=None, mode=1, writer=None, epoch=None):
        # the grad mode is an option of THIS forward's nodes only: restored afterwards, so that autograd nodes applied outside
        # EDITOR.forward (tests, library users) never inherit a no-grad forward's "save nothing" setting
        fn.set_model_options(self.grad_scale_f16, self.act_light, torch.is_grad_enabled())
        try:
            return self._forward(x, cam_label, label, view_label, img_path, mode, writer, epoch)
        finally:
            fn.set_model_options(grad_enabled=True)

    def _forward(self, x, cam_label=None, label=None, view_label=None, img_path=None, mode=1, writer=None, epoch=None):
        mods = [x[m_[0]].contiguous() for m_ in self.modalities]             # make_model.py:153-155
        rgb = mods[0]
        nmod = self.nmod
        if not rgb.is_cuda:
            raise RuntimeError("EDITOR (MI355X build): inputs must be on the GPU; there is no CPU fallback path")
        b = rgb.shape[0]
        dim = self.BACKBONE.token_dim
        # The frequency branch (wavelet counts + the serial top-10 selection, ~0.25 ms of mostly latency) depends only on the
        # images and is needed only when the masks are OR-ed after the backbone: it runs on the side stream, beside the
        # patch embedding and the first blocks (a parallel branch of the captured graph as well).
        cur = torch.cuda.current_stream(rgb.device)
        side = fn._side_stream(rgb.device)
        side.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(side):
            mask_fre, _ = ops.frequency_mask(mods[0], mods[1], mods[2], self.FREQ_INDEX.keep, mods[3] if nmod > 3 else None)
            fre_done = side.record_event()
        feats, probs = self._backbone(mods, cam_label)
        t = feats.shape[1]
        cur.wait_event(fre_done)
        with torch.no_grad():
            index = self._select(probs, mask_fre, b)
            if self.teacher_index is not None:
                index = self.teacher_index.to(index.device).to(torch.uint8).contiguous()
                self.last_aux["index"] = index
        del probs
        feats = feats.view(nmod, b, t, dim)
        training = self.training
        feats_s, loss_bcc, cls_all = fn.SFTSApplyFn.apply(feats, index, training)
        cls_tri = list(cls_all.unbind(0))
        if training:
            if self.AL:
                ori = torch.cat(cls_tri, dim=-1)
                ori_score = fn.LinearFn.apply(self._bn(self.AL_BN, ori), self.AL_HEAD.weight, None)
            else:
                mod_scores = [fn.LinearFn.apply(self._bn(self.BACKBONE_BN, c), self.BACKBONE_HEAD.weight, None)
                              for c in cls_tri]
        if self.hma_compact and self.act_dtype != torch.float32 and b * t >= 256 and not self.hma_attn_f32:
            pooled, num, loss_ocfr = self._hma_compact(feats_s, index, label)
        else:                      # dense-masked form, as the reference computes it (always used in f32 parity mode)
            fused, loss_ocfr = self._hma(feats_s, index, label)
            pooled, num = fn.PoolFn.apply(fused, nmod, t)
        if training and writer is not None:
            writer.add_scalar("num_count", num.mean(), epoch)                      # make_model.py:199-200
        pooled_m = pooled.unbind(0)              # (unbind's backward is one stack; per-index selects zero-fill and add)
        red = [fn.LinearFn.apply(pooled_m[i], getattr(self, m_[1] + "_REDUCE").weight, getattr(self, m_[1] + "_REDUCE").bias)
               for i, m_ in enumerate(self.modalities)]
        cls4t = torch.cat(red, dim=-1)
        self.last_aux.update(num=num, loss_bcc=loss_bcc, loss_ocfr=loss_ocfr)
        if not training:
            return cls4t
        score = fn.LinearFn.apply(self._bn(self.FUSE_BN, cls4t), self.FUSE_HEAD.weight, None)
        aux_loss = loss_bcc + loss_ocfr
        if self.AL:
            return score, cls4t, ori_score, ori, aux_loss
        return (score, cls4t) + tuple(v for pair in zip(mod_scores, cls_tri) for v in pair) + (aux_loss,)

    def _bn(self, bn, x):
        if self.training:
            bn.num_batches_tracked += 1
        return fn.BatchNorm1dFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                      self.training)


def make_model(cfg, num_class, camera_num):
    model = EDITOR(num_class, cfg, camera_num)
    print("===========Building EDITOR===========")
    return model

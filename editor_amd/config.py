"""Duck-typed stand-in for the reference's yacs config (config/defaults.py).

Only the keys the model reads are present (SURVEY.md 5.6;
/root/reference/modeling/make_model.py:37-60,66,90-96,137).  A real yacs
CfgNode from the reference's config/defaults.py works unchanged as well - the
model only does attribute access.
"""
from types import SimpleNamespace

# dataset presets: (SIZE_TRAIN, AL, num_class, camera_num) - see SURVEY.md 8(d)
PRESETS = {
    # configs/RGBNT201/EDITOR.yml ; C=171 (data/datasets/RGBNT201.py:28), cams=4 assumed
    "RGBNT201": dict(size=(256, 128), al=1, num_class=171, cams=4),
    # configs/RGBNT100/EDITOR.yml ; C=50, cams=8 (params.py:65)
    "RGBNT100": dict(size=(128, 256), al=0, num_class=50, cams=8),
    # configs/MSVR310/EDITOR.yml but 384x128 input (BASELINE.json config 4); C=155 chosen
    "MSVR310": dict(size=(384, 128), al=0, num_class=155, cams=8),
    # BASELINE.json config 5 (synthetic extension, no dataset): 4 modalities, ViT-L/16, 512x256 input -> 512 patch tokens
    "SYNTH4L": dict(size=(512, 256), al=0, num_class=171, cams=4,
                    extra=dict(transformer_type="vit_large_patch16_224", num_modalities=4)),
}
MODALITY_KEYS = ("RGB", "NI", "TI", "M4")


def make_cfg(size_train=(256, 128), al=1, transformer_type="vit_base_patch16_224",
             head_keep=2, frequency_keep=10, drop_path=0.1, sie_camera=True,
             sie_coe=3.0, stride=(16, 16), **extra):
    """Build a cfg carrying the 14 keys of SURVEY.md 5.6 with the shipped defaults
    (/root/reference/config/defaults.py:34-60, configs/*/EDITOR.yml)."""
    model = SimpleNamespace(
        PRETRAIN_PATH_T="", PRETRAIN_CHOICE="none",
        TRANSFORMER_TYPE=transformer_type, SIE_CAMERA=sie_camera, SIE_COE=sie_coe,
        SIE_VIEW=False, STRIDE_SIZE=list(stride), DROP_PATH=drop_path, DROP_OUT=0.0,
        ATT_DROP_RATE=0.0, ID_LOSS_TYPE="softmax", HEAD_KEEP=head_keep,
        FREQUENCY_KEEP=frequency_keep, AL=al, DIST_TRAIN=False, NAME="EDITOR")
    # extension knobs of this build (absent from the reference cfg => defaults)
    model.COMPUTE_DTYPE = extra.pop("compute_dtype", "bf16")   # 'bf16' | 'f16' | 'f16x2' | 'f32'
    if model.COMPUTE_DTYPE == "f16x2s":                          # shorthand: split-precision forward, selection scope only
        model.COMPUTE_DTYPE, model.SPLIT_SCOPE = "f16x2", "selection"
    model.HMA_COMPACT = extra.pop("hma_compact", True)
    model.ACT_LIGHT = extra.pop("act_light", False)              # blocks save 24 B instead of 36 B per token-row-element
    model.ROLLOUT_PROBS = extra.pop("rollout_probs", False)      # bf16 mode: keep the materialised (L,3B,h,T,T) probabilities
    for k, v in extra.items():
        setattr(model, k.upper(), v)
    # config/defaults.py:96-138 (the keys solver/make_optimizer.py and solver/scheduler_factory.py read)
    solver = SimpleNamespace(OPTIMIZER_NAME="SGD", MAX_EPOCHS=70, BASE_LR=0.001, LARGE_FC_LR=False, BIAS_LR_FACTOR=2,
                             MOMENTUM=0.9, WEIGHT_DECAY=0.0001, WEIGHT_DECAY_BIAS=0.0001, WARMUP_ITERS=10,
                             CENTER_LR=0.5, MARGIN=0.3, SEED=1111, IMS_PER_BATCH=128)
    return SimpleNamespace(MODEL=model, SOLVER=solver,
                           DATALOADER=SimpleNamespace(SAMPLER="softmax_triplet", NUM_INSTANCE=16, NUM_WORKERS=14),
                           INPUT=SimpleNamespace(SIZE_TRAIN=list(size_train),
                                                 SIZE_TEST=list(size_train)))


def preset(name, **over):
    p = PRESETS[name]
    kw = dict(p.get("extra", {}))
    kw.update(over)
    al = kw.pop("al", p["al"])
    size = kw.pop("size_train", p["size"])
    cfg = make_cfg(size_train=size, al=al, **kw)
    return cfg, p["num_class"], p["cams"]

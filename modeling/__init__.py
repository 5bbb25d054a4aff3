"""Top-level drop-in for the reference's `modeling` package: `from modeling import make_model`
(/root/reference/modeling/__init__.py:1) resolves to the MI355X-native implementation, so
engine/processor.py, tools/train.py, test_net.py and params.py run unchanged."""
from editor_amd.modeling import make_model  # noqa: F401

"""Experiment helper: run tools/attn_bench.py against a variant build of the library (EDITOR_LIB_VARIANT=<path to .so>)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib
v = os.environ.get("EDITOR_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.abspath(v)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_bench.py"), run_name="__main__")

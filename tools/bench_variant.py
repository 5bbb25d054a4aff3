"""Same-box A/B helper: run bench.py (in-process, pass --graph) against a variant build of the library:
    EDITOR_LIB_VARIANT=editor_amd/exp_x.so python tools/bench_variant.py --graph --steps 10 --no-cpu-baseline --no-h2d"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editor_amd import _lib
v = os.environ.get("EDITOR_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.abspath(v)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")

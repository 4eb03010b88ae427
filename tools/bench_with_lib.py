"""Dev tool: bench.py against a side library from tools/build_variant.py.   python tools/bench_with_lib.py <tag> [bench.py flags]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantv2x_amd import lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "cache", "abl", f"libqv2x_{sys.argv[1]}.so")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

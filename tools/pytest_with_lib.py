"""Dev tool: run pytest against a side library from tools/build_variant.py.   python tools/pytest_with_lib.py <tag> [pytest args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantv2x_amd import lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "cache", "abl", f"libqv2x_{sys.argv[1]}.so")
import pytest
sys.exit(pytest.main(sys.argv[2:]))

"""Dev tool: build a side library tools/cache/abl/libqv2x_<tag>.so with extra -D flags on ONE source, the other objects reused.
    python tools/build_variant.py <tag> <source.hip> -DFOO=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantv2x_amd import build as B
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
out_dir = os.path.join(ROOT, "tools", "cache", "abl"); os.makedirs(out_dir, exist_ok=True)
srcp = os.path.join(B.HERE, "csrc", src)
obj = os.path.join(out_dir, f"{tag}_{src[:-4]}.o")
subprocess.check_call(["hipcc"] + B.CFLAGS + flags + ["-c", srcp, "-o", obj])
objs = [obj if s == srcp else B._obj(s) for s in B.SOURCES]
lib = os.path.join(out_dir, f"libqv2x_{tag}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)

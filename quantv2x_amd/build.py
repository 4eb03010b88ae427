"""Build ``libqv2x.so`` for gfx950 with hipcc (cross-compiles without a GPU).

One object per ``csrc/*.hip`` (compiled in parallel, rebuilt only when the source or a header changed), linked into
``quantv2x_amd/libqv2x.so``.  Objects live in ``quantv2x_amd/build/`` (git-ignored)."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HEADERS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(HERE, "..", "include", "qv2x.h")]
OUT = os.path.join(HERE, "libqv2x.so")
OBJDIR = os.path.join(HERE, "build")
# -ffp-contract=off: every fma on the parity-critical paths is written as fmaf(); the compiler must not fuse
# a separate multiply and add (oracle/qv2x_oracle.c is built the same way).
# -Wno-inline-asm: conv_i8_wide.hip issues its LDS-DMA as inline asm that writes M0 (see the comment there); hipcc warns that M0 is a
# reserved register -- the kernel has no other M0 user (no builtin LDS-DMA, no ds_*_addtid, no GWS / sendmsg).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-inline-asm"]
FLAGS = CFLAGS + ["-shared"]            # one-shot form (tools that build a variant library use it with SOURCES)


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(OUT, SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(_obj(s), [s] + HEADERS)]

    def compile_one(src):
        cmd = ["hipcc"] + CFLAGS + list(extra_flags) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as pool:
        list(pool.map(compile_one, todo))
    # the dynamic symbol table = the C entry points of include/qv2x.h (-fvisibility=hidden + the header's visibility pragma) and nothing
    # else: hipcc keeps every __global__ kernel's host-side handle at default visibility, the version script makes those local too
    vmap = os.path.join(OBJDIR, "exports.map")
    with open(vmap, "w") as f:
        f.write("{ global: qv2x_*; local: *; };\n")
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vmap, "-o", OUT] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)

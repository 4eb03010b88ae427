"""Build ``libqv2x.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
OUT = os.path.join(HERE, "libqv2x.so")
# -ffp-contract=off: every fma on the parity-critical paths is written as fmaf(); the compiler must not fuse
# a separate multiply and add (oracle/qv2x_oracle.c is built the same way).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=default"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SOURCES + glob.glob(os.path.join(HERE, "csrc", "*.h")) + [os.path.join(HERE, "..", "include", "qv2x.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = ["hipcc"] + FLAGS + ["-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)

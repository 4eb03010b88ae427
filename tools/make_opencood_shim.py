"""Writes ``integration/opencood/``: one re-export file per mirrored module of ``quantv2x_amd.plugin`` under the REFERENCE's package names
(INTEGRATION.md §2).  With ``integration/`` on ``sys.path`` in front of (or instead of) the reference tree,
``importlib.import_module("opencood.models." + core_method)`` -- the lookup of opencood/tools/train_utils.py:272-291 -- ``from
opencood.quant import QuantModel`` and ``opencood.utils.transformation_utils`` resolve to the MI355X-backed mirror.  A maintainer of the
reference copies single files of this tree over the reference's own (each is one line).  Regenerate after adding a plugin module."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "quantv2x_amd", "plugin")
DST = os.path.join(ROOT, "integration", "opencood")


def main():
    n = 0
    for dirpath, _, files in os.walk(SRC):
        rel = os.path.relpath(dirpath, SRC)
        if "__pycache__" in rel:
            continue
        out_dir = DST if rel == "." else os.path.join(DST, rel)
        os.makedirs(out_dir, exist_ok=True)
        for f in sorted(files):
            if not f.endswith(".py"):
                continue
            mod = "quantv2x_amd.plugin" + ("" if rel == "." else "." + rel.replace(os.sep, ".")) + ("" if f == "__init__.py" else "." + f[:-3])
            with open(os.path.join(out_dir, f), "w") as fh:
                fh.write(f"import {mod} as _m  # the MI355X-backed mirror of the reference module of this name (tools/make_opencood_shim.py)\n"
                         "globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})\n")
            n += 1
    print("wrote", n, "files under", DST)


if __name__ == "__main__":
    main()

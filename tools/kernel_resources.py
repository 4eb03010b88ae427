"""Dev tool (CPU): registers, scratch and LDS of every kernel in libqv2x.so, from the code objects' metadata notes.
    python tools/kernel_resources.py [path/to/lib.so] [--scratch-only]
Exit code 1 when any kernel has a private (scratch) segment: the product library must not spill."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = os.path.abspath(args[0] if args else os.path.join(root, "quantv2x_amd", "libqv2x.so"))
tmp = tempfile.mkdtemp()
try:
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=tmp, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
            rows.append((name, int(g("vgpr_count")), int(blk.split()[0]), int(g("sgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
    bad = 0
    print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s}")
    for r in sorted(rows):
        if "--scratch-only" in sys.argv and r[4] == 0:
            continue
        print(f"{r[0][:110]:110s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:8d} {r[5]:7d}")
        bad += r[4] > 0
    print(f"{len(rows)} kernels, {sum(r[4] > 0 for r in rows)} with scratch")
    sys.exit(1 if any(r[4] > 0 for r in rows) else 0)
finally:
    shutil.rmtree(tmp, ignore_errors=True)

"""CPU sanitizer pass (GPU sanitizers are not available on the pool): the checker rebuilt with AddressSanitizer + UBSan (``make -C oracle asan``,
no OpenMP) runs the golden-vector suite and the C ABI's host-side argument paths in a child interpreter with the sanitizer runtime
preloaded -- any heap overflow, use-after-free or undefined operation in ``oracle/qv2x_oracle.c`` (or in the host wrappers of
``libqv2x.so`` that the ABI tests reach without a GPU) aborts the child.  And the conv epilogues' quantizer: the reciprocal-multiply form the
oracle and the kernels share (``q_code_mul``) against the reference's division form (``make -C oracle qdiv``; quant_layer.py:132-133) on
layers with zero and NON-ZERO zero points -- the measured flip rate is asserted."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _gcc_file(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_golden_and_abi_argument_paths_under_asan_ubsan():
    asan = _gcc_file("libasan.so")
    if asan is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-C", ORACLE, "-s", "asan"])
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "QV2X_ORACLE_LIB": os.path.join(ORACLE, "_build", "libqv2x_oracle_asan.so"), "OMP_NUM_THREADS": "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_cabi_cpu.py"), "-k", "not collapsed_encoder_algebra and not adaround_mse"],         # (without the torch-heavy ones: minutes under ASan)
                       env=env, capture_output=True, text=True, timeout=3000, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def _conv_codes(lib, x, wq, zw, scale, bias, zx, da, za):
    n, h, w, cin = x.shape
    cout = wq.shape[0]
    out = np.zeros((n, h, w, cout), np.uint8)
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    gc0, gc, gzx = i32([0]), i32([cin]), i32([zx])
    zw_, sc, bi = i32(zw), np.ascontiguousarray(scale, np.float32), np.ascontiguousarray(bias, np.float32)
    lib.orc_conv3x3(p(x), n, h, w, cin, 1, 1, p(gc0), p(gc), p(gzx), p(wq), p(zw_), cout, p(sc), p(bi), 0, ctypes.c_float(da), ctypes.c_float(za),
                    p(out), cout, 0)
    return out


@pytest.mark.parametrize("za", [0.0, 131.0])
def test_reciprocal_multiply_quantizer_against_the_division_form(za):
    """``rint(fma(y, fl(1 / delta), zp))`` against ``rint(y / delta) + zp`` on a 3x3 layer's outputs: the codes differ by at most one LSB and
    only where y / delta sits within rounding of a half-integer.  With zp != 0 the single rounding of ``p + zp`` widens that window from
    ~1.2e-7 |p| to ~ulp(zp) / 2 (|p + zp| < 512: 3e-5): a flip rate of a few 1e-5, still two orders below the layer tests' 5e-4 bound."""
    subprocess.check_call(["make", "-C", ORACLE, "-s", "qdiv"])
    subprocess.check_call(["make", "-C", ORACLE, "-s"])
    mul = ctypes.CDLL(os.path.join(ORACLE, "_build", "libqv2x_oracle.so"))
    div = ctypes.CDLL(os.path.join(ORACLE, "_build", "libqv2x_oracle_qdiv.so"))
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 2, 48, 64, 64, 64
    x = rng.integers(0, 256, (n, h, w, cin), dtype=np.uint8)
    wq = rng.integers(0, 256, (cout, cin, 3, 3), dtype=np.uint8)
    zw = rng.integers(100, 156, cout)
    scale = (rng.uniform(0.5, 2.0, cout) * 3e-6).astype(np.float32)
    bias = rng.normal(0, 0.05, cout).astype(np.float32)
    da = 0.0137
    a = _conv_codes(mul, x, wq, zw, scale, bias, 7, da, za)
    b = _conv_codes(div, x, wq, zw, scale, bias, 7, da, za)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    rate = float((d > 0).mean())
    inner = float(((a > 0) & (a < 255)).mean())
    print(f"q_code_mul vs q_code, zp = {za}: {int((d > 0).sum())} of {d.size} codes differ ({rate:.2e}); {inner:.2f} of the codes unsaturated")
    assert inner > 0.2 and d.max() <= 1
    assert rate <= (2e-4 if za else 2e-5)

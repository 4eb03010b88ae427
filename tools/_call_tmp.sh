set -x
mkdir -p gpurun_out/cl
python -m pytest tests/test_hip_conv_wide.py -q -m gpu -x 2>&1 | tail -3
python tools/pytest_with_lib.py wscl tests/test_hip_conv_wide.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python tools/bench_kernels.py conv 32 2>&1 | tee gpurun_out/cl/default_$i.txt | tail -25
QV2X_LIB_TAG=wscl python tools/bench_kernels.py conv 32 2>&1 | tee gpurun_out/cl/wscl_$i.txt | tail -25
done

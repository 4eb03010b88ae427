#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_a -o p -- python3 $R/tools/bench_kernels.py conv 32 > $R/$out/pmc_a.out 2> $R/$out/pmc_a.err
python3 $R/tools/pmc_table.py /tmp/pmc_a 20 > $R/$out/pmc_sq_conv32.csv; head -30 $R/$out/pmc_sq_conv32.csv
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA --output-format csv -d /tmp/pmc_b -o p -- python3 $R/tools/bench_kernels.py conv 32 > $R/$out/pmc_b.out 2> $R/$out/pmc_b.err
python3 $R/tools/pmc_table.py /tmp/pmc_b 20 > $R/$out/pmc_sq2_conv32.csv; head -30 $R/$out/pmc_sq2_conv32.csv
tail -3 $R/$out/pmc_b.err

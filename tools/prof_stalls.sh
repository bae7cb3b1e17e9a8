#!/bin/bash
# Runs on the GPU box: the stall counters of the accumulate's matrix kernel (what do the matrix pipe's idle cycles
# wait on?), two SQ passes of the C2-sized launches of tools/prof_kernels.py fit.
#   tools/prof_stalls.sh <tag>          (select a library build with TD_HOTPATH_LIB)
set -u
tag=$1
out=gpurun_out/stalls_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT"; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 tools/prof_kernels.py fit > $out/pmc$i.log 2>&1
  python3 tools/prof_summary.py $out/pmc$i --pmc > $out/${tag}_pmc$i.txt 2>&1
  i=$((i+1))
done
rm -rf $out/pmc0 $out/pmc1 $out/pmc2
cat $out/${tag}_pmc*.txt | grep -A12 "lagcov_split\|lagcov_w1"

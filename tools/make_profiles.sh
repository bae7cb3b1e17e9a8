#!/bin/bash
# Runs on the GPU box (through gpurun): regenerates every rocprofv3 summary that DESIGN.md and
# bench.py cite, into gpurun_out/profiles_<tag>/ -- copy them to profiles/ afterwards.
#   tools/make_profiles.sh r01
set -u
tag=$1
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
# 1. the bench command itself (pipelined, the default) and its serial variant
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o t -- python3 bench.py --steps 20 --warmup 3 --no-extra > $out/bench.log 2>&1
python3 tools/prof_summary.py $out/bench > $out/${tag}_bench_kernel_stats.txt
cp $(ls $out/bench/*/t_kernel_stats.csv $out/bench/t_kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_bench_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -o t -- python3 bench.py --steps 20 --warmup 3 --serial --no-cpu --no-extra > $out/serial.log 2>&1
python3 tools/prof_summary.py $out/serial > $out/${tag}_bench_serial_kernel_stats.txt
# 2. the hot kernels alone, whole chip: trace, then PMC in separate passes
rocprofv3 --kernel-trace --stats --output-format csv -d $out/hot -o t -- python3 tools/prof_kernels.py > $out/hot.log 2>&1
python3 tools/prof_summary.py $out/hot > $out/${tag}_hotkernels_kernel_stats.txt
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  # (without the CCA dense stage: its small lagged accumulate runs the same lagcov template and
  # would dilute the per-launch averages of the C2-sized launches)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 tools/prof_kernels.py fit cg cgt shapes cca decode > $out/pmc$i.log 2>&1
  python3 tools/prof_summary.py $out/pmc$i --pmc > $out/${tag}_hotkernels_pmc$i.txt
  i=$((i+1))
done
# 3. (the bench line of an un-profiled run is taken AFTERWARDS, once profiles/<tag>_lagcov_pmc.json
#    has been rebuilt from the PMC passes above -- bench.py reads its `traffic` from there:
#      cp gpurun_out/profiles_<tag>/<tag>_* profiles/ ; python tools/make_pmc_json.py <tag>
#      gpurun -- 'python bench.py > gpurun_out/bench_line.json' ; cp ... profiles/<tag>_bench_line.json)
rm -rf $out/bench $out/serial $out/hot $out/pmc0 $out/pmc1 $out/pmc2
ls -la $out

#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace (+ optional PMC passes) of a command.
#   tools/prof.sh <tag> [--pmc "C1 C2" ...] -- <python args...>
# Outputs land in gpurun_out/<tag>/ ; a per-kernel summary is printed.
set -u
tag=$1; shift
pmcs=()
while [ "$1" != "--" ]; do
  if [ "$1" == "--pmc" ]; then pmcs+=("$2"); shift 2; else shift; fi
done
shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 "$@" > $out/trace.log 2>&1
python3 tools/prof_summary.py $out/trace > $out/kernel_stats.txt 2>&1
head -25 $out/kernel_stats.txt
i=0
for c in "${pmcs[@]}"; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc$i -o p -- python3 "$@" > $out/pmc$i.log 2>&1
  python3 tools/prof_summary.py $out/pmc$i --pmc > $out/pmc$i.txt 2>&1
  head -20 $out/pmc$i.txt
  i=$((i+1))
done

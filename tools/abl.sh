#!/bin/bash
# GPU box: time the decode kernels with each ablation build in build/abl/
for a in 0 1 2 3 4; do
  cp build/abl/libtd_abl$a.so telluride_decoding_amd/libtd_hotpath.so
  tools/prof.sh abl$a -- tools/prof_kernels.py decode > /dev/null 2>&1
  echo "abl$a: $(grep -E 'fir' gpurun_out/abl$a/kernel_stats.txt)"
done
cp build/abl/libtd_abl0.so telluride_decoding_amd/libtd_hotpath.so

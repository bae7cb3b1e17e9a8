timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for v in "" variants/lib_chain4.so variants/lib_chain2.so; do
echo "--- bias chain lib=$v"; TD_HOTPATH_LIB=$v timeout 120 python tools/bias_probe.py 2>&1 | tail -1
TD_HOTPATH_LIB=$v timeout 120 python tools/time_strong_share.py 1 2>&1 | tail -1
done
echo "--- share default"; timeout 120 python tools/time_strong_share.py 1 2 4 8 2>&1 | tail -4
timeout 200 bash tools/prof.sh acc1 -- tools/time_strong_share.py 1 > /dev/null 2>&1; python tools/timeline.py gpurun_out/acc1/trace 6
timeout 200 bash tools/prof.sh acc8 -- tools/time_strong_share.py 8 > /dev/null 2>&1; python tools/timeline.py gpurun_out/acc8/trace 6
grep ridge_c2 gpurun_out/parity.jsonl | tail -4

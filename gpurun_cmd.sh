timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for v in "" variants/lib_setprio.so; do
echo "--- lib=$v"; TD_HOTPATH_LIB=$v timeout 120 python tools/bias_probe.py 2>&1 | tail -1
TD_HOTPATH_LIB=$v timeout 120 python tools/time_strong_share.py 1 8 2>&1 | tail -2
done
timeout 200 bash tools/prof.sh acc1 -- tools/time_strong_share.py 1 > /dev/null 2>&1; python tools/timeline.py gpurun_out/acc1/trace 3

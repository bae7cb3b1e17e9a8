timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "--- new"; timeout 120 python tools/time_c3.py 2>&1 | tail -3
echo "--- old proj f32"; TD_PROJECT_F32=1 timeout 120 python tools/time_c3.py 2>&1 | tail -3
timeout 200 bash tools/prof.sh c3 -- tools/time_c3.py 5 > /dev/null 2>&1; head -8 gpurun_out/c3/kernel_stats.txt

rm -f gpurun_out/parity.jsonl
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_stdout.log 2>gpurun_out/bench_stderr.log; tail -1 gpurun_out/bench_stdout.log > gpurun_out/r03_bench_line.json
timeout 120 python tools/time_strong_share.py 1 2 4 8 2>&1 | tail -4
timeout 120 python tools/time_solve.py 1 20 160 2>&1 | tail -3
timeout 120 python tools/bias_probe.py 2>&1 | tail -1

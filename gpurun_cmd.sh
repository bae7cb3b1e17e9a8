rm -f gpurun_out/parity.jsonl
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2

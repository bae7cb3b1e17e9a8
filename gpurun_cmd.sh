timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_decoder_train.py -m gpu -x -q 2>&1 | tail -3
echo "--- old"; timeout 120 python tools/time_decode.py 2>&1 | grep "W 1000 hop  100"
echo "--- tile16"; TD_FIR_TILE16=1 timeout 120 python tools/time_decode.py 2>&1 | grep "W 1000 hop  100"
TD_FIR_TILE16=1 timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_decoder_train.py -m gpu -x -q 2>&1 | tail -3

python -m pytest tests/test_gpu_random.py -q -x -k indefinite 2>&1 | tail -15
echo "=== single GPU 2048x2048"; python tools/configs_full.py 5full 2>&1 | tail -3

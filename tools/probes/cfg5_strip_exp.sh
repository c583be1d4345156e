echo "=== single GPU 2048x2048"; timeout 900 python tools/configs_full.py 5full 2>&1 | tail -3

export PLFX_SOLVE_DEBUG=1
echo "=== single GPU 2048x2048, 12 load steps"; CFG5_STEPS=12 timeout 1200 python tools/configs_full.py 5 2>&1 | grep "minres\|MINRES\|gave up\|indefinite generator\|Jacobi fall\|config 5\|solves above" | cut -c1-330 | head -150

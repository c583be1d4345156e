# experiments behind DESIGN.md section 6 (config 5 on strips): 8 ranks on the GPUs that exist, host-staged transport
export PLFX_TOOL_TRANSPORT=host
run() { echo "=== $*"; env "$@" timeout 1300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29544 tools/configs_full.py 5full 2>&1 | grep -v "Gloo\|^W0\|\*\*\*\|^\[rank[1-7]\]" | grep "rank \|config 5\|PlfxError\|Error" | head -14; }
echo "=== single GPU 512x64"; CFG5_NX=512 CFG5_NY=64 python tools/configs_full.py 5full 2>&1 | tail -2
echo "=== single GPU 1024x256"; CFG5_NX=1024 CFG5_NY=256 python tools/configs_full.py 5full 2>&1 | tail -2
run CFG5_NX=2048 CFG5_NY=2048

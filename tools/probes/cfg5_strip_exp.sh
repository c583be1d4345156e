# config 5 at full size on one GPU (tools/configs_full.py 5full); with PLFX_TOOL_TRANSPORT=host and torch.distributed.run
# --nproc-per-node 8 the same on 8 strips sharing the GPUs that exist
echo "=== single GPU 2048x2048"; timeout 900 python tools/configs_full.py 5full 2>&1 | tail -3

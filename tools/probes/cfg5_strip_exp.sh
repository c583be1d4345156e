# config 5 at full size on 8 strips (all ranks on the GPUs that exist, host-staged transport): functional check against the
# single-GPU run of the same build (profiles/r02i_config5_full_solve.txt)
export PLFX_TOOL_TRANSPORT=host
timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29544 tools/configs_full.py 5full 2>&1 | grep -v "Gloo\|^W0\|\*\*\*\|^\[rank[1-7]\]" | grep "rank \|config 5\|PlfxError\|Error\|solves above" | head -14

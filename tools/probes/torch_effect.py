import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
mode = sys.argv[1]
if mode != 'none':
    import torch
import bench, pylabfea_amd as FE
fe = bench.tension_model(FE, bench.hill_material(FE), 1024, 0.005, device=0)
eng = fe._ensure_engine()
if mode in ('init', 'sync'):
    torch.zeros(1, device='cuda:0'); torch.cuda.synchronize()
ts = []
def hook(il):
    eng.sync()
    if mode == 'sync':
        torch.cuda.synchronize()
    ts.append(time.perf_counter())
fe._step_hook = hook
fe._max_load_steps = 14
fe.solve(min_step=50)
d = [1e3 * (b - a) for a, b in zip(ts[:-1], ts[1:])]
print(mode, 'steps 8..10: %.3f %.3f %.3f   steps 11..14 avg %.3f' % (d[6], d[7], d[8], sum(d[9:13]) / 4))

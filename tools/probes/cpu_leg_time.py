import sys, time, json
sys.path.insert(0, '/root/repo')
import bench
n = int(sys.argv[1]); nthr = int(sys.argv[2])
t = time.perf_counter()
r = bench.cpu_run(n, 3, 1, nthr, 'pcg', pcg_threads=nthr)
print(n, 'total', time.perf_counter() - t, json.dumps(r))

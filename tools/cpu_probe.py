"""What the host of the GPU box gives a CPU baseline: CPU quota of the container, NUMA layout, and the
thread scaling of the oracle's block SpMV on a first-touch-placed synthetic matrix (one process per
thread count, so that every count places its own pages)."""
import os, subprocess, sys
print("affinity cpus:", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/memory.max"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
subprocess.run("lscpu | egrep 'Model name|Socket|NUMA|Thread|Core' ; grep MemTotal /proc/meminfo", shell=True)
code = r'''
import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from oracle import binding as ol
L = ol.load("oracle/liboracle.so")
n, bs, w = 4000000, 2, 7
rp = (np.arange(n + 1, dtype=np.int64) * w).astype(np.int32)
ci = np.empty(n * w, dtype=np.int32)
base = np.arange(n, dtype=np.int64)
for k, off in enumerate((-40000, -200, -1, 0, 1, 200, 40000)):
    ci[k::w] = np.clip(base + off, 0, n - 1)
val = np.empty(n * w * bs * bs); x = np.empty(n * bs); y = np.empty(n * bs)
# first touch by the team: a parallel axpy-like pass through the oracle is not exposed, so touch through SpMV once
val[:] = 1.0; x[:] = 1.0
L.wo_bcsr_spmv(n, bs, ol.ip(rp), ol.ip(ci), ol.dp(val), ol.dp(x), ol.dp(y))
t = time.time()
for _ in range(5):
    L.wo_bcsr_spmv(n, bs, ol.ip(rp), ol.ip(ci), ol.dp(val), ol.dp(x), ol.dp(y))
el = (time.time() - t) / 5
gb = (n * w * (bs * bs * 8 + 4) + 2 * n * bs * 8) / 1e9
print("threads %4s: SpMV %.4f s  %.1f GB/s" % (os.environ["OMP_NUM_THREADS"], el, gb / el))
'''
for t in sys.argv[1:]:
    env = dict(os.environ, OMP_NUM_THREADS=t, OMP_PROC_BIND="spread", OMP_PLACES="cores")
    subprocess.run([sys.executable, "-c", code], env=env)

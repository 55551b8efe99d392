import numpy as np, time, ctypes
n = 2 * 2**30
t0 = time.perf_counter(); a = np.empty(n, np.int32); t1 = time.perf_counter(); a[::1024] = 1; t2 = time.perf_counter()
print(f"np.empty {a.nbytes/2**30:.0f} GiB: {t1-t0:.4f} s; touch 1 thread: {a.nbytes/(t2-t1)/1e9:.2f} GB/s")
print(open('/proc/self/smaps_rollup').read())

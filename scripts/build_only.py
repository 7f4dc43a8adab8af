"""build a graph and nothing else (for rocprofv3 of build_insert_kernel): rows, expansion_add, corpus mode, max_batch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rad_amd.device import DeviceIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mb = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
idx = DeviceIndex(1024, 8, 16, ef)
idx.synth_vectors(n, seed=20260101, mode=mode)
t0 = time.perf_counter(); idx.link_resident(seed=777, max_batch=mb); dt = time.perf_counter() - t0
print(f"{n} rows, expansion_add {ef}, batches of {mb}: {dt:.2f} s = {n / dt / 1e6:.3f} M inserts/s", flush=True)

"""K1 scan throughput (kernel time by HIP events) for corpus sizes that do / do not fit the caches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex
from rad_amd import _lib
for n in (1_000_000, 10_000_000, 100_000_000):
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=1, mode=1)
    for nq in (1, 8, 64):
        q = idx.read_vectors(7, nq)
        best = 1e9
        for rep in range(3):
            idx.scan(q, 0, n) if nq <= 8 else idx.scan(q, 0, n)
            best = min(best, _lib.lib().radhip_last_kernel_ms())
        print(f"n {n} nq {nq}: last-pass kernel {best:.3f} ms -> {min(nq, 8) * n / best / 1e6:.1f} G eval/s per pass, {n * 128 / best / 1e6:.0f} GB/s of rows", flush=True)
    idx.close()

"""K1 scan (8 queries per pass) and the exact top-10 scan on 2048-bit rows: run once per kernel form
(RADHIP_TOPK_ROWS / RADHIP_SCAN_ROWS = 0: a row across sixteen lanes; default: a row per lane)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rad_amd.device import DeviceIndex
from rad_amd import _lib
L = _lib.lib()
ndim = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
B = ndim // 8
idx = DeviceIndex(ndim, 8, 16, 64)
idx.synth_vectors(n, seed=3, mode=1)
Q = idx.read_vectors(1234, 8)
chunk = 12_500_000
for nq in (1, 8):
    ms = 0.0
    for f in range(0, n, chunk):
        idx.scan(Q[:nq], f, min(chunk, n - f)); ms += L.radhip_last_kernel_ms()
    print(f"scan {nq}q: {ms:.3f} ms, rows {n * B / ms / 1e6:.0f} GB/s, rows + results {n * (B + 8 * nq) / ms / 1e6:.0f} GB/s", flush=True)
idx.topk(Q, 10)
t0 = time.perf_counter(); s, a, o, c = idx.topk(Q, 10); dt = time.perf_counter() - t0
assert (s[:, 0] == np.arange(1234, 1242)).all()
print(f"top-10 of 8 queries (wall): {dt * 1e3:.3f} ms, {n * B / dt / 1e9:.0f} GB/s", flush=True)

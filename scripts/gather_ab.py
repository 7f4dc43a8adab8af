import sys, time, numpy as np
sys.path.insert(0, ".")
from rad_amd.device import DeviceIndex
from rad_amd import _lib
L = _lib.lib()
n = 100_000_000
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=3, mode=1)
q = idx.read_vectors(7, 8)
rng = np.random.default_rng(5)
for m in (20_000_000, 100_000_000):
    slots = rng.integers(0, n, m, dtype=np.uint32)
    off = (np.arange(5, dtype=np.uint64) * (m // 4)).astype(np.uint64); off[-1] = m
    best = 1e9
    for _ in range(3):
        idx.gather(q[:4], slots, off); best = min(best, L.radhip_last_kernel_ms())
    print(f"{m} pairs: {best:.3f} ms  {m / best / 1e6:.2f} G pairs/s  {m * 132 / best / 1e6:.0f} GB/s", flush=True)

"""Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts in this path's own access
patterns (MI355X_MICROARCH.md: FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950):
K1 scan (streaming 16 B/lane) and K2 gather (random 128-B rows, 16 B/lane)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex
n = 40_000_000
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=1, mode=1)
q = idx.read_vectors(5, 1)
idx.scan(q, 0, n)                                   # reads n*128 B, writes n*8 B
rng = np.random.default_rng(0)
m = 20_000_000
slots = rng.integers(0, n, m).astype(np.uint32)
idx.gather(q, slots, np.array([0, m], np.uint64))   # reads m*128 B rows + m*8 B pair arrays, writes m*8 B
print("scan_read_bytes", n * 128, "scan_write_bytes", n * 8)
print("gather_read_bytes", m * 128 + m * 8, "gather_write_bytes", m * 8)

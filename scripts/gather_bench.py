"""K2 gather throughput on random slots of a resident corpus (kernel time by HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex
from rad_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for ndim in (1024, 2048):
    idx = DeviceIndex(ndim, 8, 16, 64)
    nn = n if ndim == 1024 else n // 2
    idx.synth_vectors(nn, seed=1, mode=1)
    NQ = int(os.environ.get("GB_NQ", 4))
    q = idx.read_vectors(5, NQ)
    rng = np.random.default_rng(0)
    m = int(os.environ.get("GB_PAIRS", 40_000_000))
    slots = rng.integers(0, nn, m).astype(np.uint32)
    off = (np.arange(NQ + 1, dtype=np.uint64) * (m // NQ)).astype(np.uint64); off[-1] = m
    for rep in range(3):
        idx.gather(q, slots, off)
        ms = _lib.lib().radhip_last_kernel_ms()
        B = ndim // 8
        print(f"{ndim}-bit: {m/1e6:.0f}M pairs in {ms:.2f} ms = {m/ms/1e6:.2f} G pairs/s, rows {m*B/ms/1e6:.0f} GB/s, algorithmic (B+4+8 written) {m*(B+12)/ms/1e6:.0f} GB/s", flush=True)
    idx.close()

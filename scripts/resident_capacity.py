import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
from rad_amd.device import DeviceIndex
from rad_amd import _lib
for ndim, M in ((1024, 8), (2048, 32), (2048, 8), (1024, 16)):
    idx = DeviceIndex(ndim, M, 2 * M, 64)
    idx.synth_vectors(1000, seed=1, mode=1)
    cap = C.c_uint32(0)
    rc = _lib.lib().radhip_traversal_resident_capacity(idx._h, C.byref(cap))
    print(ndim, M, "resident capacity", cap.value, "=", cap.value / 256, "per CU", rc)

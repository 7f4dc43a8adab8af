#!/usr/bin/env python3
"""A/B: the traversal kernel of the library named by RADHIP_LIB on one built graph (hash table only).
    python scripts/ab_bench.py [n_rows] [corpus_mode] [nq] [n_to_score]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
lib = C.CDLL(os.environ["RADHIP_LIB"])
n = int(sys.argv[1]); mode = int(sys.argv[2]); nq = int(sys.argv[3]); nts = int(sys.argv[4])
V = C.c_void_p
def ck(rc):
    if rc: lib.radhip_last_error.restype = C.c_char_p; raise RuntimeError(lib.radhip_last_error())
h = V(); ck(lib.radhip_index_create(C.c_uint32(1024), C.c_uint32(8), C.c_uint32(16), C.c_uint32(64), C.c_int(0), C.byref(h)))
ck(lib.radhip_index_synth_vectors(h, C.c_uint64(n), C.c_uint64(0), C.c_uint64(n), C.c_uint64(20260101), C.c_int(mode)))
X = np.empty((n, 128), np.uint8)
for f in range(0, n, 4_000_000):
    c = min(4_000_000, n - f); ck(lib.radhip_index_read_vectors(h, C.c_uint64(f), C.c_uint64(c), X[f:f + c].ctypes.data_as(V)))
lib.radhip_index_destroy(h)
h = V(); ck(lib.radhip_index_create(C.c_uint32(1024), C.c_uint32(8), C.c_uint32(16), C.c_uint32(64), C.c_int(0), C.byref(h)))
t0 = time.time()
for f in range(0, n, 5_000_000):
    c = min(5_000_000, n - f); ck(lib.radhip_index_add(h, X[f:f + c].ctypes.data_as(V), C.c_uint64(c), C.c_uint64(777), C.c_uint32(16384)))
print(f"build {time.time() - t0:.1f}s", flush=True)
Q = np.ascontiguousarray(X[np.random.default_rng(0).integers(0, n, nq)])
del X
os.environ.setdefault("RADHIP_TRAV", "4"); os.environ["RADHIP_TABLE"] = "hash"
t = V(); ck(lib.radhip_traversal_create(h, Q.ctypes.data_as(V), C.c_uint32(nq), C.c_uint64(nts), C.c_uint32(0), C.byref(t)))
for rep in range(4):
    if rep: ck(lib.radhip_traversal_reset(t, Q.ctypes.data_as(V)))
    run = C.c_uint32(0); ck(lib.radhip_traversal_run(t, C.c_uint64(0), C.byref(run)))
    ms = C.c_double(0); nl = C.c_uint64(0); ck(lib.radhip_traversal_kernel_time(t, C.byref(ms), C.byref(nl)))
    print(f"{os.environ['RADHIP_LIB']}: rep {rep} {ms.value:.1f} ms", flush=True)

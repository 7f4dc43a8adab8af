"""The device queue key (rad_amd/csrc/common.h rh_make_key, exported on the host as
radhip_rad_key) must order exactly like the reference's Redis ZSET
(rad/priority_queue.py:22-42): ascending float score, ties by the bytes of the
member string "{node_id}:{level}".  Host logic only — no GPU needed."""
import ctypes as C
import functools

import numpy as np


def _lib():
    from rad_amd import _lib
    return _lib.lib()


def test_q24_is_strictly_monotone_in_float32_distance():
    """All (and, or) pairs with or <= 2048: sorting by the 24-bit q equals sorting by the
    float32 edge value 1 - and/or, and distinct rationals never collide in either."""
    L = _lib()
    ors = np.arange(1, 2049, dtype=np.int64)
    a = np.concatenate([np.arange(0, o + 1, dtype=np.int64) for o in ors])
    o = np.concatenate([np.full(o + 1, o, dtype=np.int64) for o in ors])
    q = ((o - a) << 23) // o                       # what rh_q24 must compute
    d32 = (np.float32(1.0) - (a.astype(np.float32) / o.astype(np.float32))).astype(np.float32)
    # spot-check the exported host restatement of the device function (double division)
    rng = np.random.default_rng(0)
    for i in rng.integers(0, a.size, 20000):
        key = L.radhip_rad_key(int(a[i]), int(o[i]), 0, 0)
        assert (key >> 38) == int(q[i])
    # exact rational order: compare via cross products on a sorted-by-q permutation
    order = np.lexsort((a, q))
    qs, ds, as_, os_ = q[order], d32[order], a[order], o[order]
    same_q = qs[1:] == qs[:-1]
    # equal q  <=> equal rational  <=> equal float32
    cross_eq = (os_[1:] - as_[1:]) * os_[:-1] == (os_[:-1] - as_[:-1]) * os_[1:]
    assert np.array_equal(same_q, cross_eq)
    assert np.array_equal(same_q, ds[1:] == ds[:-1])
    # increasing q => strictly increasing float32 distance
    assert np.all(ds[1:][~same_q] > ds[:-1][~same_q])


def _redis_cmp(x, y):
    (sx, mx), (sy, my) = x, y
    if sx != sy:
        return -1 if sx < sy else 1
    return -1 if mx < my else (1 if mx > my else 0)


def test_key_order_equals_redis_zset_order():
    L = _lib()
    rng = np.random.default_rng(1)
    slots = [0, 1, 9, 10, 11, 19, 99, 100, 101, 109, 110, 199, 999, 1000, 1999, 19999, 99999999,
             100000000, 199999999, 999999999, 12, 123, 1234, 12345, 123456, 1234567, 12345678,
             123456789, 2, 20, 200, 29, 299]
    slots += [int(x) for x in rng.integers(0, 1_000_000_000, 300)]
    slots += [int(x) for x in rng.integers(0, 5000, 300)]
    slots = sorted(set(slots))
    scores = [(3, 7), (6, 14), (1, 3), (0, 5), (5, 5), (100, 1024), (0, 0), (50, 97), (49, 95)]
    items = []
    for s in slots:
        for lv in (0, 1, 2, 9, 10, 11, 15):
            a, o = scores[rng.integers(0, len(scores))]
            d = float(np.float32(1.0) - np.float32(a) / np.float32(o)) if o else 0.0
            member = f"{s}:{lv}".encode()
            items.append(((d, member), L.radhip_rad_key(a, o, s, lv), s, lv))
    assert len({k for _, k, _, _ in items}) == len({(m) for (_, m), _, _, _ in items})
    by_redis = sorted(items, key=functools.cmp_to_key(lambda x, y: _redis_cmp(x[0], y[0])))
    by_key = sorted(items, key=lambda x: x[1])
    assert [x[0] for x in by_redis] == [x[0] for x in by_key]
    # decode round trip
    for _, k, s, lv in items[:2000]:
        ds, dl = C.c_uint32(), C.c_uint32()
        L.radhip_rad_key_decode(k, C.byref(ds), C.byref(dl))
        assert (ds.value, dl.value) == (s, lv)

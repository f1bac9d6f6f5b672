"""One FRESH process of tests/test_race_twin.py: loads the product library and its full-synchronisation twin (no torch: ctypes and numpy only,
HIP initialised by the library's own runtime), solves every batch of the fixture file with both through the host-pointer entry of the C ABI,
and prints one line: `same` / `DIFFER` (product vs twin, bit for bit, per batch) and a digest of the product's results.

    python tests/_race_worker.py <fixture.npz> <liblscqp.so> <liblscqp_sync.so>
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsc_dr_planner_amd import api  # noqa: E402  (struct layouts and make_desc only: api.lib() -- which imports torch -- is never called)


def bind(path):
    L = C.CDLL(path)
    vp = C.c_void_p
    L.lscqp_create.restype = C.c_int
    L.lscqp_create.argtypes = [C.POINTER(api.ClassDesc), C.POINTER(vp)]
    L.lscqp_destroy.argtypes = [vp]
    L.lscqp_solve_batch.restype = C.c_int
    L.lscqp_solve_batch.argtypes = [vp, C.c_int64] + [vp] * 9
    L.lscqp_last_error.restype = C.c_char_p
    return L


def solve(L, desc, hdr, rows, off, sfc, x0, nv):
    h = C.c_void_p()
    rc = L.lscqp_create(C.byref(desc), C.byref(h))
    assert rc == 0, L.lscqp_last_error()
    n = len(hdr)
    x, obj, st, info = np.zeros((n, nv)), np.zeros(n), np.full(n, -1, np.int32), np.zeros(n, api.INFO_DTYPE)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for _ in range(2):  # twice: the second call runs on warm caches and a warm staging slot -- another interleaving of the same work
        rc = L.lscqp_solve_batch(h, n, p(hdr), p(rows), p(off), p(sfc), p(x0), p(x), p(obj), p(st), p(info))
        assert rc == 0, L.lscqp_last_error()
    L.lscqp_destroy(h)
    return x, obj, st, info


def main():
    fx = np.load(sys.argv[1], allow_pickle=False)
    prod, twin = bind(sys.argv[2]), bind(sys.argv[3])
    words, dig = [], hashlib.md5()
    for b in range(int(fx["n_batches"])):
        g = lambda k: np.ascontiguousarray(fx["%s_%d" % (k, b)])  # noqa: E731
        M, dim = int(fx["M_%d" % b]), int(fx["dim_%d" % b])
        desc = api.make_desc(M=M, dim=dim, world_min=tuple(fx["wmin_%d" % b]), world_max=tuple(fx["wmax_%d" % b]))
        hdr, rows, off, sfc, x0 = g("hdr").view(api.HEADER_DTYPE).reshape(-1), g("rows").view(api.ROW_DTYPE).reshape(-1), g("off"), g("sfc").view(api.BOX_DTYPE).reshape(-1), g("x0")
        nv = dim * M * 6
        a = solve(prod, desc, hdr, rows, off, sfc, x0, nv)
        t = solve(twin, desc, hdr, rows, off, sfc, x0, nv)
        same = all(np.array_equal(u.view(np.uint8), v.view(np.uint8)) for u, v in zip(a, t))
        words.append("same" if same else "DIFFER")
        for u in a:
            dig.update(np.ascontiguousarray(u).view(np.uint8).tobytes())
        words.append("steps%d" % int(a[3]["iterations"].max()))
        words.append("nonopt%d" % int((a[2] != 0).sum()))
    print(" ".join(words), dig.hexdigest())


if __name__ == "__main__":
    main()

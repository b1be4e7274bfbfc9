"""ctypes front end of oracle/knn_oracle.c (test infrastructure only; parity unpinned -- see the C header)."""
import ctypes as C

import numpy as np

from . import surfel_oracle as _so


def knn_mean_dist2(points, K=3, reference=None, take_sqrt=False):
    """Mean of the K smallest squared distances from every point to the other points (reference=None), or from every
    point of `points` to the cloud `reference`.  float32 brute force."""
    lib = _so.lib()
    q = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 3))
    self_search = reference is None
    r = q if self_search else np.ascontiguousarray(np.asarray(reference, np.float32).reshape(-1, 3))
    out = np.empty(q.shape[0], np.float32)
    fp = C.POINTER(C.c_float)
    lib.so_knn_mean_dist2(C.c_int(q.shape[0]), q.ctypes.data_as(fp), C.c_int(r.shape[0]), r.ctypes.data_as(fp), C.c_int(int(K)),
                          C.c_int(int(self_search)), C.c_int(int(bool(take_sqrt))), out.ctypes.data_as(fp))
    return out

"""ctypes loader of oracle/_build/libsaa_oracle.so (test infrastructure — see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsaa_oracle.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if build and (not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "saa_oracle.c"))):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        _lib.rato_oracle_drone.argtypes = [C.c_int, C.c_int, C.c_double] + [dp] * 10 + [C.c_int]
        _lib.rato_oracle_drone.restype = None
        _lib.rato_oracle_drone_stream.argtypes = [C.c_int, C.c_int, C.c_double] + [dp] * 8 + [C.c_int]
        _lib.rato_oracle_drone_stream.restype = None
        _lib.rato_oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def drone(us, DWs, masses, obs_Qs, dt, nthreads=0, want=("xs", "v_final_du", "val_final", "g_obs_du", "g_up", "Z"),
          out=None):
    """All samples through the C oracle -> dict of fp64 arrays in the reference's shapes.
    ``out``: a dict returned by an earlier call with the same shapes (buffers are reused, so repeated
    timing runs do not pay first-touch page faults)."""
    lib = load()
    us = np.ascontiguousarray(us, dtype=np.float64)
    DWs = np.ascontiguousarray(DWs, dtype=np.float64)
    masses = np.ascontiguousarray(masses, dtype=np.float64)
    obs_Qs = np.ascontiguousarray(obs_Qs, dtype=np.float64)
    M, S = DWs.shape[0], DWs.shape[1]
    shapes = {"xs": (M, S + 1, 6), "v_final_du": (M, 6, 3 * S), "val_final": (M, 6),
              "g_obs_du": (M, 3, S, 3 * S), "g_up": (M, 3, S), "Z": (M,)}
    if out is None:
        out = {k: (np.empty(shapes[k]) if k in want else None) for k in shapes}
    lib.rato_oracle_drone(M, S, float(dt), _p(us), _p(DWs), _p(masses), _p(obs_Qs), _p(out["xs"]),
                          _p(out["v_final_du"]), _p(out["val_final"]), _p(out["g_obs_du"]), _p(out["g_up"]),
                          _p(out["Z"]), int(nthreads))
    return out


def drone_stream(us, DWs, masses, obs_Qs, dt, nthreads=0):
    """The same per-sample work with the dense outputs formed in per-thread buffers and reduced on the fly (batches
    whose (M,3,S,3S) Jacobian does not fit the host) -> dict(sum_final_du (6,3S), sum_val_final (6,), Z (M,), checksum)."""
    lib = load()
    us = np.ascontiguousarray(us, dtype=np.float64)
    DWs = np.ascontiguousarray(DWs, dtype=np.float64)
    masses = np.ascontiguousarray(masses, dtype=np.float64)
    obs_Qs = np.ascontiguousarray(obs_Qs, dtype=np.float64)
    M, S = DWs.shape[0], DWs.shape[1]
    out = {"sum_final_du": np.zeros((6, 3 * S)), "sum_val_final": np.zeros(6), "Z": np.empty(M), "checksum": np.zeros(1)}
    lib.rato_oracle_drone_stream(M, S, float(dt), _p(us), _p(DWs), _p(masses), _p(obs_Qs), _p(out["sum_final_du"]),
                                 _p(out["sum_val_final"]), _p(out["Z"]), _p(out["checksum"]), int(nthreads))
    return out


def max_threads():
    return load().rato_oracle_max_threads()

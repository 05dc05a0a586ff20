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
        _lib.rato_oracle_place_threads.argtypes = [C.c_int, C.c_int]
        _lib.rato_oracle_place_threads.restype = C.c_int
        ip = C.POINTER(C.c_int)
        _lib.rato_oracle_drone_rowmax.argtypes = [C.c_int, C.c_int, C.c_double] + [dp] * 6 + [ip, C.c_int]
        _lib.rato_oracle_drone_rowmax.restype = None
        _lib.rato_oracle_drone_tail_rows.argtypes = ([C.c_int, C.c_int, C.c_double] + [dp] * 4 + [C.c_int, dp, ip, dp, dp,
                                                                                              C.c_int])
        _lib.rato_oracle_drone_tail_rows.restype = None
        _lib.rato_oracle_car.argtypes = [C.c_int, C.c_int] + [dp] * 9 + [C.c_int]
        _lib.rato_oracle_car.restype = None
        _lib.rato_oracle_car_rowmax.argtypes = [C.c_int, C.c_int] + [dp] * 7 + [ip, C.c_int]
        _lib.rato_oracle_car_rowmax.restype = None
        _lib.rato_oracle_car_tail_rows.argtypes = [C.c_int, C.c_int] + [dp] * 5 + [C.c_int, dp, ip, dp, dp, C.c_int]
        _lib.rato_oracle_car_tail_rows.restype = None
    return _lib


def place_threads(nthreads, spread=True):
    """OMP_PROC_BIND=spread by hand for the next calls with this ``nthreads`` (workers only; the calling thread is not
    pinned); ``spread=False`` gives the workers the whole affinity mask back.  -> cpus in the process's mask."""
    return int(load().rato_oracle_place_threads(int(nthreads), 1 if spread else 0))


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


def default_threads():
    """threads of the streaming cut oracle: given EXPLICITLY to every call -- the cutting-plane loop that calls it runs
    under threadpoolctl's 4-thread limit (cvar_cuts.CvarCutSolver.solve), which would otherwise clamp OpenMP too"""
    return max(1, min(128, os.cpu_count() or 1))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class DroneCutOracle:
    """Streaming fp64 cut oracle on a drone batch (rato_oracle_drone_rowmax / _tail_rows): rows of the reference's QP
    (drone_risk.py:357-364, without kappa, y_i, t) formed one sample at a time, nothing stored per sample."""

    def __init__(self, DWs, masses, obs_Qs, dt, nthreads=0):
        self.DWs, self.masses, self.obs_Qs = _f64(DWs), _f64(masses), _f64(obs_Qs)
        self.M, self.S, self.dt, self.nU = self.DWs.shape[0], self.DWs.shape[1], float(dt), 3 * self.DWs.shape[1]
        self.nthreads = nthreads or default_threads()

    def rowmax(self, us_k, u):
        """-> (m (M,), arg (M,)):  m_i = max_r (G_i(u_k) u - g_up_i(u_k))_r"""
        m, arg = np.empty(self.M), np.empty(self.M, dtype=np.int32)
        load().rato_oracle_drone_rowmax(self.M, self.S, self.dt, _p(_f64(us_k)), _p(_f64(u).reshape(-1)), _p(self.DWs),
                                        _p(self.masses), _p(self.obs_Qs), _p(m), _ip(arg), int(self.nthreads))
        return m, arg

    def tail_rows(self, us_k, w, arg):
        """w, arg (K, M) -> (sum_i w_ki G_i[arg_ki] (K, nU), sum_i w_ki g_up_i[arg_ki] (K,))"""
        w, arg = np.atleast_2d(_f64(w)), np.ascontiguousarray(np.atleast_2d(arg), dtype=np.int32)
        K = w.shape[0]
        grad, gup = np.empty((K, self.nU)), np.empty(K)
        load().rato_oracle_drone_tail_rows(self.M, self.S, self.dt, _p(_f64(us_k)), _p(self.DWs), _p(self.masses),
                                           _p(self.obs_Qs), K, _p(w), _ip(arg), _p(grad), _p(gup), int(self.nthreads))
        return grad, gup

    def final_rows(self, us_k):
        """-> sample means (final_du (6, 3S), val_final (6,)) of drone_risk.py:294-300"""
        st = drone_stream(us_k, self.DWs, self.masses, self.obs_Qs, self.dt, self.nthreads)
        return st["sum_final_du"] / self.M, st["sum_val_final"] / self.M


def car(us, states_init, omegas_speed, omegas_rep, DWs, nthreads=0):
    """Driving, all samples, dense -> dict(xs (M,S+1,8), g_obs_du (M,S,2S), g_up (M,S), Z (M,))."""
    DWs = _f64(DWs)
    M, S = DWs.shape[0], DWs.shape[1]
    out = {"xs": np.empty((M, S + 1, 8)), "g_obs_du": np.empty((M, S, 2 * S)), "g_up": np.empty((M, S)), "Z": np.empty(M)}
    load().rato_oracle_car(M, S, _p(_f64(us)), _p(_f64(states_init)), _p(_f64(omegas_speed)), _p(_f64(omegas_rep)), _p(DWs),
                           _p(out["xs"]), _p(out["g_obs_du"]), _p(out["g_up"]), _p(out["Z"]), int(nthreads))
    return out


class CarCutOracle:
    """Streaming fp64 cut oracle on a driving batch (driving.py:358-363 rows, one sample at a time)."""

    def __init__(self, states_init, omegas_speed, omegas_rep, DWs, nthreads=0):
        self.x0, self.ws, self.wr, self.DWs = _f64(states_init), _f64(omegas_speed), _f64(omegas_rep), _f64(DWs)
        self.M, self.S, self.nU = self.DWs.shape[0], self.DWs.shape[1], 2 * self.DWs.shape[1]
        self.nthreads = nthreads or default_threads()

    def rowmax(self, us_k, u):
        m, arg = np.empty(self.M), np.empty(self.M, dtype=np.int32)
        load().rato_oracle_car_rowmax(self.M, self.S, _p(_f64(us_k)), _p(_f64(u).reshape(-1)), _p(self.x0), _p(self.ws),
                                      _p(self.wr), _p(self.DWs), _p(m), _ip(arg), int(self.nthreads))
        return m, arg

    def tail_rows(self, us_k, w, arg):
        w, arg = np.atleast_2d(_f64(w)), np.ascontiguousarray(np.atleast_2d(arg), dtype=np.int32)
        K = w.shape[0]
        grad, gup = np.empty((K, self.nU)), np.empty(K)
        load().rato_oracle_car_tail_rows(self.M, self.S, _p(_f64(us_k)), _p(self.x0), _p(self.ws), _p(self.wr), _p(self.DWs),
                                         K, _p(w), _ip(arg), _p(grad), _p(gup), int(self.nthreads))
        return grad, gup


def max_threads():
    return load().rato_oracle_max_threads()

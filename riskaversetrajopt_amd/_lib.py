"""ctypes binding of include/rato_saa.h.  There is NO CPU fallback: if the HIP
library is missing, loading fails loudly."""
import ctypes as C
import os

from . import _build

_LIB = None
ABI_VERSION = 12             # RATO_ABI_VERSION of include/rato_saa.h this binding was written against

c_float_p = C.c_void_p   # device pointers travel as integers
c_stream = C.c_void_p


class DroneParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("ld", C.c_int32), ("S", C.c_int32), ("dt", C.c_float), ("beta", C.c_float),
                ("drag", C.c_float), ("kp", C.c_float), ("kd", C.c_float), ("tol", C.c_float),
                ("x_init", C.c_float * 6), ("x_final", C.c_float * 6),
                ("obs_xy", (C.c_float * 2) * 3), ("rows_out", C.c_int32),      # 28 words: the doubles start 8-byte aligned
                ("dt64", C.c_double), ("beta64", C.c_double), ("drag64", C.c_double), ("kp64", C.c_double),
                ("kd64", C.c_double), ("tol64", C.c_double), ("x_init64", C.c_double * 6),
                ("x_final64", C.c_double * 6), ("obs_xy64", (C.c_double * 2) * 3),
                # statistics in the same launch (row-parallel linearize kernel): workspace, record, tail level, threshold
                ("stats_workspace", C.c_void_p), ("stats_out", C.c_void_p), ("stats_alpha", C.c_double),
                ("stats_thr", C.c_float), ("stats_flags", C.c_int32)]


class CarParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("S", C.c_int32), ("dt", C.c_float), ("beta", C.c_float),
                ("speed_ped_des", C.c_float), ("d_min", C.c_float), ("tol", C.c_float),
                ("ego_init", C.c_float * 4), ("ego_goal", C.c_float * 4), ("rows_out", C.c_int32),   # 16 words
                ("dt64", C.c_double), ("beta64", C.c_double), ("speed_ped_des64", C.c_double), ("d_min64", C.c_double),
                ("ego_init64", C.c_double * 4),
                ("stats_workspace", C.c_void_p), ("stats_out", C.c_void_p), ("stats_alpha", C.c_double),
                ("stats_thr", C.c_float), ("stats_flags", C.c_int32)]


class CutConfig(C.Structure):
    """rato_cut_config (include/rato_saa.h)"""
    _fields_ = [(k, C.c_int32) for k in ("system", "S", "cap", "keep_max", "keep_recent", "keep_idle", "mode_saa",
                                         "recycle")] + \
               [("M", C.c_int64)] + [(k, C.c_double) for k in ("alpha", "alphaM", "c_s", "rhs0", "u_min", "u_max")] + \
               [("thr", C.c_float), ("params", C.c_void_p)] + [(k, C.c_void_p) for k in ("s0", "s1", "s2", "s3")] + \
               [(k, C.c_void_p) for k in ("uk_dev", "uk_host", "x_host", "x_dev", "ring_m", "ring_arg", "ring_res",
                                          "workspace")] + [("workspace_bytes", C.c_size_t)] + \
               [(k, C.c_void_p) for k in ("part", "part_b", "sums_b_host", "slots_dev", "slots_host", "res_host",
                                          "p_diag", "q")]


class CutResult(C.Structure):
    """rato_cut_result (include/rato_saa.h)"""
    _fields_ = [("us", C.c_void_p)] + \
               [(k, C.c_double) for k in ("slack", "t_risk", "phi", "oracle_s", "master_s", "lam_slack")] + \
               [(k, C.c_int32) for k in ("cuts", "recycled", "status", "uncertified_cuts")] + \
               [("cut_slot", C.c_void_p), ("cut_lambda", C.c_void_p), ("cut_capacity", C.c_int32),
                ("n_cut_rows", C.c_int32), ("bound_var", C.c_void_p), ("bound_sign", C.c_void_p),
                ("bound_lambda", C.c_void_p), ("bound_capacity", C.c_int32), ("n_bounds", C.c_int32)]


class ScpIter(C.Structure):
    """rato_scp_iter (include/rato_saa.h)"""
    _fields_ = [(k, C.c_double) for k in ("define_s", "solve_s", "oracle_s", "master_s", "t_risk", "slack", "phi")] + \
               [(k, C.c_int32) for k in ("cuts", "status", "recycled", "reserved")]


# name -> (restype, argtypes); mirrors include/rato_saa.h one to one
SIGNATURES = {
    "rato_abi_version": (C.c_int, []),
    "rato_packed_tile_stride": (C.c_size_t, [C.c_size_t]),
    "rato_packed_buffer_floats": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "rato_device_clock_probe": (C.c_int, [c_float_p, C.c_int32, c_stream]),
    "rato_device_occupy": (C.c_int, [C.c_int32, C.c_int64, c_stream]),
    "rato_drone_eval": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 7 + [c_stream]),
    "rato_drone_linearize_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rato_drone_linearize": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 10 + [C.c_int32, C.c_int32, c_stream]),
    "rato_drone_rowmax_implicit": (C.c_int, [C.POINTER(DroneParams), c_float_p, c_float_p, C.c_int32, c_float_p,
                                             c_float_p, C.c_double, c_float_p, c_float_p, c_float_p, c_stream]),
    "rato_drone_linearize_generators": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 9 + [c_stream]),
    "rato_drone_tail_rows_implicit": (C.c_int, [C.POINTER(DroneParams), c_float_p, c_float_p, C.c_int32] +
                                      [c_float_p] * 5 + [C.c_int64, c_float_p, C.c_int32, C.c_double, c_float_p,
                                                         c_stream]),
    "rato_drone_rowmax_rollout": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 7 + [c_stream]),
    "rato_drone_tail_rows_rollout": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 7 +
                                     [C.c_int64, c_float_p, C.c_int32, C.c_double, c_float_p, c_stream]),
    "rato_car_rowmax_rollout": (C.c_int, [C.POINTER(CarParams)] + [c_float_p] * 8 + [c_stream]),
    "rato_car_tail_rows_rollout": (C.c_int, [C.POINTER(CarParams)] + [c_float_p] * 8 +
                                   [C.c_int64, c_float_p, C.c_int32, C.c_double, c_float_p, c_stream]),
    "rato_cut_oracle_rollout": (C.c_int, [C.c_int32] + [c_float_p] * 10 + [C.c_double, C.c_float, C.c_double, C.c_void_p,
                                          C.c_size_t, c_float_p, c_float_p, c_float_p, c_stream]),
    "rato_kkt_sums": (C.c_int, [c_float_p, C.c_int64, c_float_p, c_float_p, C.c_int64, c_float_p, c_float_p, C.c_int32,
                                C.c_double, C.c_double, c_float_p, c_stream]),
    "rato_cut_solver_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CutConfig)]),
    "rato_cut_solver_destroy": (None, [C.c_void_p]),
    "rato_cut_config_bytes": (C.c_size_t, []),
    "rato_cut_result_bytes": (C.c_size_t, []),
    "rato_cut_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_stream]),
    "rato_scp_iter_bytes": (C.c_size_t, []),
    "rato_scp_run_drone": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_int32] +
                           [C.c_void_p] * 11 + [c_stream]),
    "rato_cut_define_drone": (C.c_int, [C.c_void_p] * 6 + [C.c_int64] + [C.c_void_p] * 5 + [C.c_int32, c_stream]),
    "rato_cut_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double,
                                 C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32,
                                 C.POINTER(CutResult), c_stream]),
    "rato_copy_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "rato_stream_synchronize": (C.c_int, [c_stream]),
    "rato_master_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "rato_master_add_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rato_master_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rato_master_rows": (C.c_int32, [C.c_void_p]),
    "rato_master_destroy": (None, [C.c_void_p]),
    "rato_drone_obstacle_constraints": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 3 + [c_stream]),
    "rato_car_ego_scratch_floats": (C.c_size_t, [C.c_int32]),
    "rato_car_linearize_plan": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rato_car_eval": (C.c_int, [C.POINTER(CarParams)] + [c_float_p] * 9 + [c_stream]),
    "rato_car_separation_distances": (C.c_int, [C.POINTER(CarParams), c_float_p, c_float_p, c_stream]),
    "rato_car_linearize": (C.c_int, [C.POINTER(CarParams)] + [c_float_p] * 11 + [C.c_int32, c_stream]),
    "rato_hopper_nblocks": (C.c_int, [C.c_int32]),
    "rato_hopper_slip": (C.c_int, [C.c_int32, C.c_int32] + [c_float_p] * 12 + [c_stream]),
    "rato_hopper_slip_host_inputs": (C.c_int, [C.c_int32, C.c_int32] + [c_float_p] * 12 + [c_stream]),
    "rato_hopper_slip_hessian": (C.c_int, [C.c_int32, C.c_int32] + [c_float_p] * 3 + [C.c_int32] + [c_float_p] * 9 + [c_stream]),
    "rato_hopper_jacobian_nnz": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "rato_hopper_emit_jacobian_values": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double] + [c_float_p] * 4 +
                                         [C.c_int32, c_float_p, c_stream]),
    "rato_emit_csc_values": (C.c_int, [c_float_p, c_float_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int64, C.c_float, c_float_p, c_stream]),
    "rato_saa_rowmax": (C.c_int, [c_float_p, c_float_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                  c_float_p, C.c_double, c_float_p, C.c_int32, c_float_p, c_float_p, c_stream]),
    "rato_saa_tail_rows_batch": (C.c_int, [c_float_p, c_float_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                           c_float_p, c_float_p, c_float_p, c_float_p, C.c_int64, c_float_p,
                                           C.c_int32, C.c_double, c_float_p, c_stream]),
    "rato_unpack_records": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, c_float_p, c_float_p,
                                      c_stream]),
    "rato_nnls_warm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "rato_sum_partials": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_double, c_float_p, c_stream]),
    "rato_sum_partials_f64": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_double, c_float_p, c_stream]),
    "rato_count_nonfinite": (C.c_int, [c_float_p, C.c_int64, c_float_p, c_stream]),
    "rato_count_nonfinite_acc": (C.c_int, [c_float_p, C.c_int64, c_float_p, c_stream]),
    "rato_comm_available": (C.c_int, []),
    "rato_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rato_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32]),
    "rato_comm_world": (C.c_int, [C.c_void_p]),
    "rato_comm_rank": (C.c_int, [C.c_void_p]),
    "rato_comm_allgather": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int64, c_stream]),
    "rato_comm_exchange": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int64, C.c_int32, C.c_int64, c_float_p,
                                     c_float_p, c_stream]),
    "rato_comm_destroy": (C.c_int, [C.c_void_p]),
    "rato_philox_u32": (C.c_int, [c_float_p, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, c_stream]),
    "rato_philox_normal": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), c_stream]),
    "rato_philox_uniform": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float), c_stream]),
    "rato_drone_sample": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_float, C.c_float,
                                    C.POINTER(C.c_float), C.c_float, c_float_p, c_float_p, c_float_p, c_stream]),
    "rato_drone_linearize_philox": (C.c_int, [C.POINTER(DroneParams), c_float_p, C.c_uint64, C.c_float] + [c_float_p] * 8
                                    + [c_stream]),
    "rato_drone_tiled_noise_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "rato_drone_tile_noise": (C.c_int, [c_float_p, C.c_int64, C.c_int64, C.c_int32, c_float_p, c_stream]),
    "rato_drone_linearize_tiled": (C.c_int, [C.POINTER(DroneParams)] + [c_float_p] * 10 + [c_stream]),
    "rato_drone_eval_philox": (C.c_int, [C.POINTER(DroneParams), c_float_p, C.c_uint64, C.c_float] + [c_float_p] * 5 +
                               [c_stream]),
    "rato_car_sample": (C.c_int, [C.c_int64, C.c_int32, C.c_float, C.c_uint64] + [C.c_float] * 4 +
                        [C.POINTER(C.c_float), C.POINTER(C.c_float)] + [c_float_p] * 4 + [c_stream]),
    "rato_car_linearize_philox": (C.c_int, [C.POINTER(CarParams), c_float_p, C.c_uint64, C.c_float] + [c_float_p] * 9
                                  + [c_stream]),
    "rato_car_tiled_noise_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "rato_car_tile_noise": (C.c_int, [c_float_p, C.c_int64, C.c_int32, c_float_p, c_stream]),
    "rato_car_linearize_tiled": (C.c_int, [C.POINTER(CarParams)] + [c_float_p] * 11 + [c_stream]),
    "rato_car_eval_philox": (C.c_int, [C.POINTER(CarParams), c_float_p, C.c_uint64, C.c_float] + [c_float_p] * 7 +
                             [c_stream]),
    "rato_hopper_sample": (C.c_int, [C.c_int64, C.c_uint64, c_float_p, c_float_p, c_float_p, c_stream]),
    "rato_risk_stats_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "rato_risk_stats_init": (C.c_int, [C.c_void_p, C.c_size_t, c_stream]),
    "rato_sums_and_risk_stats": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_double, c_float_p, c_float_p, C.c_int64,
                                           C.c_double, C.c_float, C.c_void_p, C.c_size_t, c_float_p, c_stream]),
    "rato_risk_stats": (C.c_int, [c_float_p, C.c_int64, C.c_double, C.c_float, C.c_void_p, C.c_size_t,
                                  c_float_p, c_stream]),
    "rato_drone_stats_in_launch": (C.c_int, [C.c_int32, C.c_int32]),
    "rato_drone_rows_streaming_stores": (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    "rato_car_rows_streaming_stores": (C.c_int, [C.c_int64, C.c_int32]),
    "rato_drone_eval_stats_in_launch": (C.c_int, [C.c_int32]),
    "rato_drone_eval_batch": (C.c_int, [C.POINTER(DroneParams), C.c_int32, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                        C.c_int64, C.c_double, C.c_float, C.c_void_p, C.c_size_t, c_float_p, c_stream]),
    "rato_car_eval_batch": (C.c_int, [C.POINTER(CarParams), C.c_int32, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                      c_float_p, C.c_int64, C.c_double, C.c_float, C.c_void_p, C.c_size_t, c_float_p, c_stream]),
    "rato_risk_stats_batch": (C.c_int, [c_float_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_float, C.c_void_p,
                                        C.c_size_t, c_float_p, c_stream]),
    "rato_car_eval_stats_in_launch": (C.c_int, [C.c_int32]),
    "rato_car_stats_in_launch": (C.c_int, [C.c_int32, C.c_int32]),
    "rato_risk_stats_recover": (C.c_int, [c_float_p, C.c_int64, C.c_double, C.c_float, C.c_void_p, C.c_size_t,
                                  c_float_p, c_stream]),
}


RATO_ENONFINITE, RATO_EINFEASIBLE, RATO_ERANK, RATO_ESELECT, RATO_ENNLS = -2, -4, -6, -7, -8     # rato_saa.h


class RatoError(RuntimeError):
    pass


class RatoNonFiniteError(RatoError):
    """RATO_ENONFINITE: a checked device output holds NaN/Inf (the reference only prints, drone_risk.py:458-459)."""
    status = -2


def lib_path():
    # RATO_SAA_LIB: alternate build of the same ABI (diagnostic builds under tools/)
    return os.environ.get("RATO_SAA_LIB") or _build.LIB_PATH


def load():
    """Load librato_saa.so (built by ``__graft_entry__.build()`` /
    ``python -m riskaversetrajopt_amd._build``).  Raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64; it must be the HIP runtime already in the
    # process when our library binds, or launches land in a second runtime that owns
    # none of torch's device allocations (seen as hipErrorNoDevice).
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise RatoError(
            f"HIP extension {path} is missing — run `python -m riskaversetrajopt_amd._build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the SAA hot path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so is stale: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.rato_abi_version() != ABI_VERSION:
        raise RatoError(f"{path} reports ABI version {lib.rato_abi_version()}, this binding needs {ABI_VERSION}: "
                        "rebuild with `python -m riskaversetrajopt_amd._build`")
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        raise RatoError(f"{what} failed with status {status}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    """the current torch stream of the current device as a C pointer (the raw getter: torch.cuda.current_stream()
    builds a Stream object through three layers of device-index helpers, ~10 us a call)"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def copy_async(dst, src, stream=None):
    """dst <- src (torch tensors: device or host -- pinned for the copy to be asynchronous --, contiguous, same byte
    size) on the current stream (rato_copy_async)."""
    nbytes = src.numel() * src.element_size()
    if nbytes != dst.numel() * dst.element_size() or not (dst.is_contiguous() and src.is_contiguous()):
        raise RatoError("copy_async needs contiguous tensors of the same byte size")
    check(load().rato_copy_async(dst.data_ptr(), src.data_ptr(), nbytes, current_stream() if stream is None else stream),
          "rato_copy_async")


def synchronize(stream=None):
    check(load().rato_stream_synchronize(current_stream() if stream is None else stream), "rato_stream_synchronize")


def require_f32_device(t, name):
    """Kernels read raw fp32 device memory: reject anything else loudly."""
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RatoError(f"{name} must be a device tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        raise RatoError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RatoError(f"{name} must be contiguous")
    return t


PACKED_ALIGN_BYTES = int(os.environ.get("RATO_PACKED_ALIGN_BYTES", 2 << 20))   # tiles of >= 1 MiB start on 2 MiB boundaries
#                                  (rato_packed_tile_stride, rato_saa.h; the variable only serves A/B builds of tools/)


def packed_tile_stride(shape):
    """Floats between consecutive tiles of a packed tile-blocked buffer of ``shape`` = (n_tiles, rows..., TILE)."""
    payload = 1
    for d in shape[1:]:
        payload *= int(d)
    return int(load().rato_packed_tile_stride(payload)), payload


def packed_buffer(shape, device):
    """Device buffer for a packed, tile-blocked Jacobian [n_tiles][rows...][TILE] in the layout the kernels use: a
    plain contiguous tensor when the tiles are packed back to back, otherwise a strided VIEW (tile stride =
    rato_packed_tile_stride, first tile on a 2 MiB boundary) of a larger flat allocation; ``.data_ptr()`` is what the
    kernels take either way."""
    import torch
    stride, payload = packed_tile_stride(shape)
    if stride == payload:
        return torch.empty(tuple(shape), dtype=torch.float32, device=device)
    pad = PACKED_ALIGN_BYTES // 4
    flat = torch.empty(int(load().rato_packed_buffer_floats(int(shape[0]), payload)) + pad, dtype=torch.float32, device=device)
    off = (-(flat.data_ptr() // 4)) % pad
    strides = [stride]
    acc = payload
    for d in shape[1:]:
        acc //= int(d)
        strides.append(acc)
    return torch.as_strided(flat, tuple(int(d) for d in shape), tuple(strides), storage_offset=off)


def is_packed_layout(t, shape):
    """``t`` has ``shape`` and the tile stride / alignment ``packed_buffer`` would give it."""
    if t is None or tuple(t.shape) != tuple(int(d) for d in shape):
        return False
    stride, payload = packed_tile_stride(shape)
    if stride == payload:
        return t.is_contiguous()
    inner = t[0]
    return inner.is_contiguous() and t.stride(0) == stride and t.data_ptr() % PACKED_ALIGN_BYTES == 0

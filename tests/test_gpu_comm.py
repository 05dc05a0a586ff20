"""GPU: the RCCL collective behind the C ABI (rato_comm_*, include/rato_saa.h).  The test boxes have ONE GPU and
RCCL refuses two ranks on one device, so what runs here is a world-size-1 communicator: library binding (dlopen of
the librccl already in the process), unique id, init, stream-ordered all-gather, the fused exchange (all-gather +
rato_unpack_records) against the host layout, teardown.  Multi-rank logic: tests/test_dist_gloo.py (CPU, gloo) and
tests/test_gpu_dist.py (two ranks sharing the GPU through the staged path)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_exchange_matches_host_layout():
    import torch
    from riskaversetrajopt_amd import _lib, dist as rdist
    lib = _lib.load()
    torch.cuda.set_device(0)
    ident = (C.c_uint8 * 128)()
    assert lib.rato_comm_unique_id(ident) == 0
    assert any(ident)                                               # an id was written
    comm = C.c_void_p()
    assert lib.rato_comm_init(C.byref(comm), bytes(ident), 0, 1) == 0
    try:
        assert lib.rato_comm_world(comm) == 1 and lib.rato_comm_rank(comm) == 0
        n_sums, M = 7, 1001
        rec = rdist.Record(n_sums, M, "cuda:0", z_row=M + 3)
        rng = np.random.RandomState(0)
        sums, Z = rng.randn(n_sums), rng.randn(M).astype(np.float32)
        rec.sums.copy_(torch.as_tensor(sums))
        rec.Z.copy_(torch.as_tensor(Z))
        all_, Z_all, total = rec._buffers(1)
        st = _lib.current_stream()
        assert lib.rato_comm_allgather(comm, _lib.ptr(rec.buf), _lib.ptr(all_), rec.rec_bytes, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(all_, rec.buf)
        all_.zero_()
        assert lib.rato_comm_exchange(comm, _lib.ptr(rec.buf), _lib.ptr(all_), rec.rec_bytes, n_sums, M,
                                      _lib.ptr(total), _lib.ptr(Z_all), st) == 0
        torch.cuda.synchronize()
        assert np.array_equal(total.cpu().numpy(), sums) and np.array_equal(Z_all.cpu().numpy(), Z)
        # bad arguments are refused, not executed
        assert lib.rato_comm_exchange(comm, _lib.ptr(rec.buf), _lib.ptr(all_), rec.rec_bytes - 4, n_sums, M,
                                      _lib.ptr(total), _lib.ptr(Z_all), st) == -1
        assert lib.rato_comm_allgather(None, _lib.ptr(rec.buf), _lib.ptr(all_), 8, st) == -1
    finally:
        assert lib.rato_comm_destroy(comm) == 0
    assert lib.rato_comm_init(C.byref(comm), bytes(ident), 1, 1) == -1      # rank out of range


def test_nonfinite_status_is_raised_by_the_facade():
    """RATO_ENONFINITE: a NaN in the inputs reaches g_up / Z and linearize_device(check_finite) raises"""
    import torch
    from oracle import drone as od
    from riskaversetrajopt_amd import _lib, drone_risk, scp
    S, M = 20, 300
    DWs, masses, obs_Qs = od.sample_uncertain_parameters(np.random.RandomState(0), 'saa', M=M, S=S)
    d = drone_risk.Model(S, DWs, masses, obs_Qs, 'saa', 0.1, check_finite=True)
    us = d.initial_guess_us_mat()
    d.linearize_device(us)                                          # finite: no error
    d.linearize_generators_device(us)
    d._dW[3, 1, 17] = float("nan")
    with pytest.raises(_lib.RatoNonFiniteError):
        d.linearize_device(us)
    with pytest.raises(_lib.RatoNonFiniteError):
        d.linearize_generators_device(us)
    d.check_finite = False
    d.linearize_device(us)                                          # unchecked: the reference's behaviour
    with pytest.raises(_lib.RatoNonFiniteError):                    # the SCP drivers switch the check on
        scp.run_drone_reduced(d, num_scp_iters_max=2)
    assert _lib.RatoNonFiniteError.status == -2


def test_nonfinite_status_in_the_table_free_driving_solve():
    """the table-free reduced solve has no linearization to scan: the oracle's statistics of m are checked instead"""
    from oracle import driving as ocar
    from riskaversetrajopt_amd import _lib, driving, scp
    S, M = 20, 300
    samples = ocar.sample_uncertain_parameters(np.random.RandomState(0), M, 'saa', S)
    d = driving.Model(M, 'saa', 0.1, S=S, samples=samples)
    scp.run_driving_reduced(d, num_scp_iters_max=3)                 # finite: no error
    d._dW[3, 1, 17] = float("nan")
    d._cut_solver = None
    with pytest.raises(_lib.RatoNonFiniteError):
        scp.run_driving_reduced(d, num_scp_iters_max=3)

"""GPU: the sample-sharded reduced SCP (Model.shard + cvar_cuts with a process group).  Two ranks share the one
GPU of the test box (gloo, collectives staged through the host); each owns half of the samples.  Both ranks must
return the single-process iterate on the full batch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _samples(M, S):
    from oracle import drone as od
    return od.sample_uncertain_parameters(np.random.RandomState(5), 'saa', M=M, S=S)


def _worker(rank, world, port, M, S, iters, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from riskaversetrajopt_amd import dist as rdist, drone_risk, scp
    rdist.init_from_env(backend="gloo")
    DWs, masses, Q = _samples(M, S)
    lo, hi = rdist.shard_bounds(M, rank, world)
    model = drone_risk.Model(S, DWs[lo:hi], masses[lo:hi], Q[lo:hi], 'saa', 0.1).shard()
    out = scp.run_drone_reduced(model, num_scp_iters_max=iters)
    np.save(os.path.join(tmpdir, f"us_{rank}.npy"), out["us"])
    np.save(os.path.join(tmpdir, f"cuts_{rank}.npy"), out["cuts"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_reduced_scp_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from riskaversetrajopt_amd import drone_risk, scp
    M, S, iters, world = 2000, 20, 8, 2
    mp.spawn(_worker, args=(world, _free_port(), M, S, iters, str(tmp_path)), nprocs=world, join=True)
    us0, us1 = np.load(tmp_path / "us_0.npy"), np.load(tmp_path / "us_1.npy")
    assert np.array_equal(us0, us1)                                   # the ranks stay bit-identical
    assert np.array_equal(np.load(tmp_path / "cuts_0.npy"), np.load(tmp_path / "cuts_1.npy"))
    DWs, masses, Q = _samples(M, S)
    single = scp.run_drone_reduced(drone_risk.Model(S, DWs, masses, Q, 'saa', 0.1), num_scp_iters_max=iters)
    # same cuts up to fp32 summation order inside the shards
    np.testing.assert_allclose(us0, single["us"], rtol=0, atol=2e-5)


def _car_samples(M, S):
    from oracle import driving as ocar
    return ocar.sample_uncertain_parameters(np.random.RandomState(3), M, 'saa', S)


def _car_worker(rank, world, port, M, S, iters, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from riskaversetrajopt_amd import dist as rdist, driving, scp
    rdist.init_from_env(backend="gloo")
    lo, hi = rdist.shard_bounds(M, rank, world)
    samples = tuple(np.asarray(a)[lo:hi] for a in _car_samples(M, S))
    model = driving.Model(hi - lo, 'saa', 0.05, S=S, samples=samples).shard()
    out = scp.run_driving_reduced(model, num_scp_iters_max=iters)
    np.save(os.path.join(tmpdir, f"car_us_{rank}.npy"), out["us"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_driving_reduced_scp_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from riskaversetrajopt_amd import driving, scp
    M, S, iters, world = 3000, 20, 6, 2
    mp.spawn(_car_worker, args=(world, _free_port(), M, S, iters, str(tmp_path)), nprocs=world, join=True)
    us0, us1 = np.load(tmp_path / "car_us_0.npy"), np.load(tmp_path / "car_us_1.npy")
    assert np.array_equal(us0, us1)
    single = scp.run_driving_reduced(driving.Model(M, 'saa', 0.05, S=S, samples=_car_samples(M, S)),
                                     num_scp_iters_max=iters)
    np.testing.assert_allclose(us0, single["us"], rtol=0, atol=5e-5)


def _lin_worker(rank, world, port, M, S, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from riskaversetrajopt_amd import dist as rdist, drone_risk, stats
    rdist.init_from_env(backend="gloo")
    DWs, masses, Q = _samples(M, S)
    lo, hi = rdist.shard_bounds(M, rank, world)
    model = drone_risk.Model(S, DWs[lo:hi], masses[lo:hi], Q[lo:hi], 'saa', 0.1)
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    r = model.linearize_device(us)
    # the bench's zero-copy record: the kernels write Z and the sums straight into the send buffer
    rec = rdist.Record(6 * S + 6, hi - lo, "cuda:0", z_row=r["_Z"].numel())
    r["_Z"], r["sums"] = rec.Z_row[:r["_Z"].numel()], rec.sums
    r = model.linearize_device(us, out=r)
    total, Z_all = rdist.exchange_record(rec)
    st = stats.risk_stats_device(Z_all, 0.1).cpu().numpy()
    np.save(os.path.join(tmpdir, f"lin_{rank}.npy"), np.concatenate([total.cpu().numpy(), st, Z_all.double().cpu().numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_record_exchange_equals_single_process(tmp_path):
    """One bench step on two ranks (linearize -> record all-gather -> unpack kernel -> statistics on the gathered Z)
    == the same step on the full batch in one process."""
    import torch.multiprocessing as mp
    from riskaversetrajopt_amd import drone_risk, stats
    M, S, world = 3000, 20, 2
    mp.spawn(_lin_worker, args=(world, _free_port(), M, S, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "lin_0.npy"), np.load(tmp_path / "lin_1.npy")
    assert np.array_equal(a, b)                                       # bitwise identical on both ranks
    DWs, masses, Q = _samples(M, S)
    model = drone_risk.Model(S, DWs, masses, Q, 'saa', 0.1)
    t = np.arange(S)[:, None]
    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
    r = model.linearize_device(us)
    n = 6 * S + 6
    np.testing.assert_array_equal(a[n + stats.N_STATS:], r["Z"].double().cpu().numpy())         # Z: same kernel, rank order
    st = stats.risk_stats_device(r["Z"], 0.1).cpu().numpy()
    np.testing.assert_array_equal(a[n:n + stats.N_STATS], st)                                   # exact selection on the same Z
    np.testing.assert_allclose(a[:n], r["sums"].cpu().numpy(), rtol=1e-6, atol=1e-4)  # fp32 block partials differ


def _poison_worker(rank, world, port, M, S, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from riskaversetrajopt_amd import dist as rdist, drone_risk
    rdist.init_from_env(backend="gloo")
    DWs, masses, Q = _samples(M, S)
    lo, hi = rdist.shard_bounds(M, rank, world)
    model = drone_risk.Model(S, DWs[lo:hi], masses[lo:hi], Q[lo:hi], 'saa', 0.1).shard()
    us = model.initial_guess_us_mat()
    for k in range(6):
        if k == 3 and rank == 1:
            # ONE rank's selection workspace holds a stale count: its one-launch selection gives up (NaN record) while
            # rank 0's succeeds -- the redo contains collectives, so both ranks must take it together
            model._cut_solver.ws.view(torch.int32)[100] = 7
            torch.cuda.synchronize()
        us, t_risk, info = model.solve_reduced(us, k)
    np.save(os.path.join(tmpdir, f"poison_us_{rank}.npy"), us)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_oracle_recovers_together_when_one_rank_gives_up(tmp_path):
    """the NaN-recovery branch of CvarCutSolver.evaluate re-runs collectives, so whether it is taken must be
    the same on every rank (decided from the rank-ordered sum of the ranks' thresholds, not from a rank's own record)."""
    import torch.multiprocessing as mp
    from riskaversetrajopt_amd import drone_risk
    M, S, world = 16000, 20, 2                       # M_total > 12,288: the one-launch cooperative selection
    mp.spawn(_poison_worker, args=(world, _free_port(), M, S, str(tmp_path)), nprocs=world, join=True)
    us0, us1 = np.load(tmp_path / "poison_us_0.npy"), np.load(tmp_path / "poison_us_1.npy")
    assert np.array_equal(us0, us1)
    DWs, masses, Q = _samples(M, S)
    single = drone_risk.Model(S, DWs, masses, Q, 'saa', 0.1)
    us = single.initial_guess_us_mat()
    for k in range(6):
        us, _, _ = single.solve_reduced(us, k)
    np.testing.assert_allclose(us0, us, rtol=0, atol=2e-5)


def test_pipelined_steps_on_the_device_equal_the_serial_order():
    """dist.PipelinedSteps with real streams: linearize of step n + 1 on the main stream beside [unpack of 4 emulated
    records + exact selection over all of them] of step n on the side stream, two output slots -- the statistics of every
    step equal the serial order bit for bit (what bench.py --gpus N runs by default, minus the wire)."""
    import torch
    from riskaversetrajopt_amd import _lib, dist as rdist, driving, stats
    M, S, world, K = 30000, 40, 4, 9
    dW, x0, ws_, wr = driving.sample_uncertain_parameters_device(M, S, seed=9)
    d = driving.Model.from_device(S, dW, x0, ws_, wr, 'saa', 0.05)
    lib, dev = _lib.load(), d.device
    t = np.arange(S)[:, None]
    us = [np.hstack([0.4 * np.cos(0.3 * t + 0.1 * k) + 0.1, 0.03 * np.sin(0.5 * t) + 0.004]) * (20.0 / S) for k in range(K)]
    outs = [d.linearize_device(us[0]), d.linearize_device(us[0])]
    recs = []
    for o in outs:
        rec = rdist.Record(0, M, dev)
        o["Z"] = rec.Z
        recs.append(rec)
    keys = ("G", "g_up", "Z", "final_du", "final_rhs")
    outs = [{k: o[k] for k in keys} for o in outs]
    Z_all = [torch.empty(world * M, dtype=torch.float32, device=dev) for _ in range(2)]
    total = torch.empty(1, dtype=torch.float64, device=dev)
    wss = [stats.new_workspace(world * M, dev) for _ in range(2)]
    rec_out = torch.empty((2, stats.N_STATS), dtype=torch.float64, device=dev)

    def consume(slot, _):
        all_ = recs[slot].buf.repeat(world)               # the other ranks' records: copies of this one
        _lib.check(lib.rato_unpack_records(_lib.ptr(all_), world, 0, M, recs[slot].rec_bytes, _lib.ptr(total),
                                           _lib.ptr(Z_all[slot]), _lib.current_stream()), "rato_unpack_records")
        stats.risk_stats_device(Z_all[slot], 0.05, workspace=wss[slot], out=rec_out[slot])
        return rec_out[slot].clone()

    serial = []
    for k in range(K):
        d.linearize_device(us[k], out=outs[k & 1])
        serial.append(consume(k & 1, None).cpu().numpy())
    pipe = rdist.PipelinedSteps(2, dev)
    got = []

    def consume_and_keep(slot, _):
        r = consume(slot, _)
        got.append(r)
        return r
    for k in range(K):
        pipe.step(lambda s, k=k: d.linearize_device(us[k], out=outs[s]), consume_and_keep)
    pipe.drain()
    assert len(got) == K
    for k in range(K):
        assert np.array_equal(got[k].cpu().numpy(), serial[k]), k
    assert not np.array_equal(serial[0], serial[1])


def test_pipelined_steps_keep_fresh_producer_tensors_alive_for_the_side_stream():
    """``produce`` may return freshly allocated tensors; the consumer reads them on the side stream after
    PipelinedSteps has dropped its last reference, and the caching allocator would hand the block to the next main-stream
    allocation while the side stream still reads it.  The tensors are marked in use on the side stream (record_stream): a
    slow consumer must still see the producer's numbers, not those of the allocations that follow."""
    import torch
    from riskaversetrajopt_amd import dist as rdist
    dev = torch.device("cuda:0")
    pipe = rdist.PipelinedSteps(2, dev)
    n_el, K = 1 << 20, 12
    seen = []

    def produce_n(n):
        def f(slot):
            return {"x": torch.full((n_el,), float(n), device=dev), "n": n}      # fresh allocation every step
        return f

    def consume(slot, out):
        torch.cuda._sleep(3_000_000)                       # the consumer is slow: the main stream runs far ahead
        seen.append((out["n"], out["x"].sum()))            # (read on the side stream)
        return None
    for n in range(K):
        pipe.step(produce_n(n), consume)
        junk = [torch.full((n_el,), -1.0, device=dev) for _ in range(3)]   # main-stream allocations of the same size
        del junk
    pipe.drain()
    torch.cuda.synchronize()
    assert [n for n, _ in seen] == list(range(K))
    for n, s in seen:
        assert float(s) == float(n) * n_el, (n, float(s))
    assert len(pipe.issued) <= rdist.PipelinedSteps.ISSUED_KEPT

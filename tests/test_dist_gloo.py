"""CPU, world_size 2 over gloo: the sample-axis sharding and the single
all-gather exchange of [partial sums | Z shard] (riskaversetrajopt_amd/dist.py).
The merged statistics must equal the single-process ones on the concatenated
batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M_total, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from riskaversetrajopt_amd import dist as rdist
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.RandomState(11)
    Z_full = rng.randn(M_total).astype(np.float32)
    sums_per_sample = rng.randn(M_total, 7)
    lo, hi = rdist.shard_bounds(M_total, rank, world)
    Z_local = torch.from_numpy(Z_full[lo:hi].copy())
    sums_local = torch.from_numpy(sums_per_sample[lo:hi].sum(0))
    total, Z_all = rdist.exchange(sums_local, Z_local)
    np.save(os.path.join(tmpdir, f"Z_{rank}.npy"), Z_all.numpy())
    np.save(os.path.join(tmpdir, f"sums_{rank}.npy"), total.numpy())
    # the zero-copy record (producers write into the send buffer; row stride > M_local, odd -> padded)
    rec = rdist.Record(7, hi - lo, "cpu", z_row=hi - lo + 3)
    rec.sums.copy_(sums_local)
    rec.Z.copy_(Z_local)
    total_r, Z_all_r = rdist.exchange_record(rec)
    assert torch.equal(total_r, total) and torch.equal(Z_all_r, Z_all) and rec.rec_bytes % 8 == 0
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world2_gloo(tmp_path):
    world, M_total = 2, 2048
    port = _free_port()
    mp.spawn(_worker, args=(world, port, M_total, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.RandomState(11)
    Z_full = rng.randn(M_total).astype(np.float32)
    sums_per_sample = rng.randn(M_total, 7)
    Z0, Z1 = np.load(tmp_path / "Z_0.npy"), np.load(tmp_path / "Z_1.npy")
    s0, s1 = np.load(tmp_path / "sums_0.npy"), np.load(tmp_path / "sums_1.npy")
    assert np.array_equal(Z0, Z_full) and np.array_equal(Z1, Z_full)       # contiguous shards, rank order
    assert np.array_equal(s0, s1)                                          # bitwise identical on every rank
    half = M_total // 2
    expect = sums_per_sample[:half].sum(0) + sums_per_sample[half:].sum(0)
    np.testing.assert_array_equal(s0, expect)
    # merged statistics == single-process statistics on the full batch
    from oracle import stats as ostats
    assert ostats.monte_carlo_var(Z0, 0.1) == ostats.monte_carlo_var(Z_full, 0.1)
    assert ostats.monte_carlo_avar(Z0, 0.1) == ostats.monte_carlo_avar(Z_full, 0.1)


def test_shard_bounds_cover_everything():
    from riskaversetrajopt_amd import dist as rdist
    for M_total in (1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [rdist.shard_bounds(M_total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == M_total
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    from riskaversetrajopt_amd import dist as rdist
    sums = torch.arange(5, dtype=torch.float64) * 1.5
    Z = torch.arange(9, dtype=torch.float32) - 3
    rec = rdist.pack_record(sums, Z)
    assert rec.dtype == torch.uint8 and rec.numel() == 8 * 5 + 4 * 9
    s, z = rdist.unpack_records(torch.cat([rec, rec]), 2, 5, 9)
    assert torch.equal(s[0], sums) and torch.equal(s[1], sums) and torch.equal(z, torch.cat([Z, Z]))


def test_single_process_exchange_is_identity():
    from riskaversetrajopt_amd import dist as rdist
    sums, Z = torch.ones(3, dtype=torch.float64), torch.zeros(4)
    a, b = rdist.exchange(sums, Z)
    assert a is sums and b is Z


def _mismatch_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from riskaversetrajopt_amd import dist as rdist
    rdist.init_from_env(backend="gloo")
    ok = rdist.gather_concat(torch.arange(5, dtype=torch.float32) + rank)           # equal lengths: fine
    assert ok.numel() == 10
    # a LATER call where only ONE rank's length changed: a rank-local cache keyed on the local length would send that
    # rank into the length check while the other, finding its length cached, is already in the all-gather -- every
    # rank must raise instead
    n = 5 if rank == 0 else 6
    try:
        rdist.gather_concat(torch.zeros(n))
        outcome = "no error"
    except ValueError as e:
        outcome = "ValueError" if "differ" in str(e) else repr(e)
    flags = [rdist.any_rank(rank == 1, "cpu"), rdist.any_rank(False, "cpu")]
    with open(os.path.join(tmpdir, f"mismatch_{rank}.txt"), "w") as f:
        f.write(f"{outcome} {flags}")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_concat_length_mismatch_raises_on_every_rank_after_a_good_call(tmp_path):
    world = 2
    mp.spawn(_mismatch_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"mismatch_{r}.txt").read_text() == "ValueError [True, False]"


def _selfcheck_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from riskaversetrajopt_amd import dist as rdist
    rdist.init_from_env(backend="gloo")
    rng = np.random.RandomState(5 + rank)
    rec = rdist.Record(7, 1000, "cpu", z_row=1003)
    rec.sums.copy_(torch.from_numpy(rng.randn(7)))
    rec.Z.copy_(torch.from_numpy(rng.randn(1000).astype(np.float32)))
    good = rdist.comm_selfcheck(rec)

    def corrupted(r, group=None):                    # an exchange that delivers ONE wrong float, on rank 1 only
        total, Z_all = rdist.exchange_record(r, group)
        Z_all = Z_all.clone()
        if rank == 1:
            Z_all[17] += 1.0
        return total, Z_all
    bad = rdist.comm_selfcheck(rec, exchange_fn=corrupted)
    import json
    with open(os.path.join(tmpdir, f"selfcheck_{rank}.json"), "w") as f:
        json.dump({"good": good, "bad": bad}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_comm_selfcheck_agrees_on_every_rank(tmp_path):
    """bench.py --gpus N validates the exchange before it times it (dist.comm_selfcheck): every rank returns the same
    verdict, a single wrong float on one rank fails the run on ALL ranks."""
    import json
    world = 2
    mp.spawn(_selfcheck_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"selfcheck_{r}.json")) for r in range(world)]
    assert res[0] == res[1]
    assert res[0]["good"]["ok"] and res[0]["good"]["world"] == 2 and res[0]["good"]["rccl_ranks"] == 0
    assert res[0]["good"]["bitwise_vs_torch_all_gather"] and res[0]["good"]["identical_on_every_rank"]
    assert not res[0]["bad"]["ok"] and not res[0]["bad"]["identical_on_every_rank"]
    from riskaversetrajopt_amd import dist as rdist
    assert rdist.comm_selfcheck(None)["ok"]                 # single process: nothing to check


# ---- round 5: the N > 1 step as a two-slot pipeline (dist.PipelinedSteps: what bench.py --gpus N runs by default) ----------
def _pipeline_worker(rank, world, port, M_local, K, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import stats as ostats
    from riskaversetrajopt_amd import dist as rdist
    rdist.init_from_env(backend="gloo")
    alpha, n_sums = 0.1, 5

    def data(n):                                         # this rank's shard of step n
        rng = np.random.RandomState(1000 * n + rank)
        return torch.from_numpy(rng.randn(M_local).astype(np.float32) - 1.0), torch.from_numpy(rng.randn(n_sums))

    def statistics(total, Z_all):
        Z = Z_all.numpy().astype(np.float64)
        return np.concatenate([[ostats.monte_carlo_var(Z, alpha), ostats.monte_carlo_avar(Z, alpha), np.mean(Z <= 1e-6)],
                               total.numpy()])

    # serial order: produce(n), exchange(n), statistics(n)
    rec = rdist.Record(n_sums, M_local, "cpu")
    serial = []
    for n in range(K):
        Z, sums = data(n)
        rec.Z.copy_(Z)
        rec.sums.copy_(sums)
        serial.append(statistics(*rdist.exchange_record(rec)))
    # pipelined order: produce(n + 1) is issued BEFORE the exchange + statistics of step n, two output slots
    recs = [rdist.Record(n_sums, M_local, "cpu"), rdist.Record(n_sums, M_local, "cpu")]
    pipe = rdist.PipelinedSteps(2, None)
    got = {}

    def produce(n):
        def f(slot):
            Z, sums = data(n)
            recs[slot].Z.copy_(Z)
            recs[slot].sums.copy_(sums)
            return n
        return f

    def consume(slot, n):
        got[n] = statistics(*rdist.exchange_record(recs[slot]))
        return n
    for n in range(K):
        pipe.step(produce(n), consume)
    assert len(got) == K - 1                              # the last step's consumer is still pending ...
    pipe.drain()
    assert len(got) == K                                  # ... until the pipeline is drained
    order = list(pipe.issued)
    for n in range(K - 1):
        assert order.index(("produce", n + 1)) < order.index(("consume", n))     # the overlap: next producer first
        assert order.index(("consume", n)) < (order.index(("produce", n + 2)) if n + 2 < K else len(order))   # slot reuse
    for n in range(K):
        assert np.array_equal(got[n], serial[n]), (n, got[n], serial[n])         # bit for bit the serial statistics
    np.save(os.path.join(tmpdir, f"pipe_{rank}.npy"), np.stack([got[n] for n in range(K)]))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_steps_equal_the_serial_order_world2_gloo(tmp_path):
    """bench.py --gpus N (N > 1) overlaps the exchange + statistics of step n with the hot kernel of step n + 1
    (dist.PipelinedSteps).  World 2 over gloo, host records: same statistics as the serial order bit for bit, identical on
    both ranks, the consumer of a slot always issued before the slot is produced into again."""
    world, K = 2, 7
    mp.spawn(_pipeline_worker, args=(world, _free_port(), 1500, K, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "pipe_0.npy"), np.load(tmp_path / "pipe_1.npy")
    assert np.array_equal(a, b) and a.shape == (K, 3 + 5)

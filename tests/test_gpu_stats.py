"""GPU parity: risk statistics (exact radix select + CVaR closed form) and the
deterministic partial-sum stage, bit-exact / fp64-tight against NumPy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np_stats(Z32, alpha, thr=1e-6):
    from oracle import stats as ostats
    Z = Z32.astype(np.float64)
    return dict(var=ostats.monte_carlo_var(Z, alpha), cvar=ostats.monte_carlo_avar(Z, alpha),
                frac=np.mean(Z32 <= np.float32(thr)), mean=Z.mean(), max=Z.max())


@pytest.mark.parametrize("M", [1, 2, 63, 64, 1000, 1 << 13, (1 << 13) + 1, 10000, 12 * 1024, 12 * 1024 + 1, 20 * 1024 + 1, 1 << 15, 123457,
                               1 << 19, (1 << 19) + 1, 1 << 20, (1 << 20) + 1])
@pytest.mark.parametrize("alpha", [0.01, 0.05, 0.3, 1.0])
def test_risk_stats_exact(M, alpha):
    from riskaversetrajopt_amd import stats
    rng = np.random.RandomState(M % 1000 + 7)
    Z = (rng.randn(M) * 0.7 - 0.3).astype(np.float32)
    st = stats.risk_stats(Z, alpha)
    ref = _np_stats(Z, alpha)
    assert st["var"] == ref["var"]                      # selection is exact (bit for bit)
    assert st["rank"] == max(M - int(np.floor(alpha * M)) - 1, 0)
    np.testing.assert_allclose(st["cvar"], ref["cvar"], rtol=1e-12, atol=1e-12)
    assert st["frac_satisfied"] == ref["frac"]
    np.testing.assert_allclose(st["mean"], ref["mean"], rtol=1e-12, atol=1e-13)
    assert st["max"] == ref["max"]


# one workgroup with LDS-resident keys (M <= 12,288) / one launch, keys in registers (<= 1,048,576) / five launches
@pytest.mark.parametrize("M", [5000, 12288, 20480, 100000, 1048576, 1100000])
def test_risk_stats_edge_distributions(M):
    from riskaversetrajopt_amd import stats
    cases = {
        "constant": np.full(M, -1.25, np.float32),
        "ties": np.repeat(np.float32([-3.0, -1.0, 0.0, 2.5]), M // 4),
        "signed_zero": np.concatenate([np.full(M // 2, -0.0, np.float32), np.full(M // 2, 0.0, np.float32)]),
        "wide": (np.random.RandomState(1).randn(M) * 1e4).astype(np.float32),
        "tiny": (np.random.RandomState(2).randn(M) * 1e-30).astype(np.float32),
        "clustered": (1.0 + 1e-6 * np.random.RandomState(3).rand(M)).astype(np.float32),
        "sorted_desc": np.linspace(5, -5, M).astype(np.float32),
        "two_bins": (1.25 + 1e-4 * (np.random.RandomState(5).rand(M) - 0.5)).astype(np.float32),
        "one_outlier": np.concatenate([np.full(M - 1, 0.5, np.float32), np.float32([1e6])]),
        "negative_cluster": (-2.0 - 1e-5 * np.random.RandomState(6).rand(M)).astype(np.float32),
    }
    for name, Z in cases.items():
        for alpha in (0.05, 0.5):
            st = stats.risk_stats(Z, alpha)
            ref = _np_stats(Z, alpha)
            assert st["var"] == ref["var"], name
            np.testing.assert_allclose(st["cvar"], ref["cvar"], rtol=1e-11, atol=1e-30, err_msg=name)
            assert st["cvar"] >= st["var"] - 1e-12
            assert st["frac_satisfied"] == ref["frac"], name


def test_monte_carlo_avar_and_var_helpers():
    from riskaversetrajopt_amd import stats
    from oracle import stats as ostats
    Z = (np.random.RandomState(4).randn(4096) - 1).astype(np.float32)
    assert stats.monte_carlo_var(Z, 0.1) == ostats.monte_carlo_var(Z.astype(np.float64), 0.1)
    np.testing.assert_allclose(stats.monte_carlo_avar(Z, 0.1), ostats.monte_carlo_avar(Z.astype(np.float64), 0.1),
                               rtol=1e-12)


def test_risk_stats_deterministic_and_rejects_bad_arguments():
    import torch
    from riskaversetrajopt_amd import stats, _lib
    Z = torch.randn(100000, device="cuda")
    a = stats.risk_stats_device(Z, 0.1).cpu().numpy()
    b = stats.risk_stats_device(Z, 0.1).cpu().numpy()
    assert np.array_equal(a, b)
    with pytest.raises(_lib.RatoError):
        stats.risk_stats_device(Z, 0.0)
    with pytest.raises(_lib.RatoError):
        stats.risk_stats_device(torch.randn(8), 0.1)      # host tensor: no CPU fallback


@pytest.mark.parametrize("nblocks,ncols", [(1, 1), (4, 306), (391, 306), (40000, 126), (33, 33)])
def test_sum_partials(nblocks, ncols):
    import torch
    from riskaversetrajopt_amd import stats
    part = torch.randn(nblocks, ncols, device="cuda") * 100
    out = stats.sum_partials(part, scale=0.5).cpu().numpy()
    ref = 0.5 * part.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-9)
    again = stats.sum_partials(part, scale=0.5).cpu().numpy()
    assert np.array_equal(out, again)


def test_count_nonfinite():
    import torch
    from riskaversetrajopt_amd import stats
    x = torch.randn(1000003, device="cuda")
    assert stats.count_nonfinite(x) == 0
    x[5] = float("nan")
    x[77777] = float("inf")
    x[-1] = -float("inf")
    assert stats.count_nonfinite(x) == 3
    assert stats.count_nonfinite(torch.full((70000,), float("nan"), device="cuda")) == 70000


def test_unpack_records_matches_host_layout():
    """rato_unpack_records: gathered [fp64 sums | fp32 Z row] records -> contiguous Z (rank order) and the sums
    added in rank order; bit-identical to the host restatement of the same layout."""
    import torch
    from riskaversetrajopt_amd import _lib, dist as rdist
    lib = _lib.load()
    rng = np.random.RandomState(3)
    for world, n_sums, M_local, z_row in ((1, 5, 7, 7), (3, 306, 1000, 1000), (8, 6, 1237, 1240), (2, 0, 64, 64),
                                           (4, 906, 999, 1001), (8, 0, 125000, 125000)):   # odd row stride; C5: 8 x 125,000
        recs = [rdist.Record(n_sums, M_local, "cuda:0", z_row=z_row) for _ in range(world)]
        for r in recs:
            if n_sums:
                r.sums.copy_(torch.as_tensor(rng.randn(n_sums) * 1e3))
            r.Z_row.copy_(torch.as_tensor(rng.randn(r.Z_row.numel()).astype(np.float32)))
        all_ = torch.cat([r.buf for r in recs])
        total = torch.empty(max(n_sums, 1), dtype=torch.float64, device="cuda:0")
        Z_all = torch.empty(world * M_local, dtype=torch.float32, device="cuda:0")
        _lib.check(lib.rato_unpack_records(_lib.ptr(all_), world, n_sums, M_local, recs[0].rec_bytes, _lib.ptr(total),
                                           _lib.ptr(Z_all), _lib.current_stream()), "rato_unpack_records")
        assert torch.equal(Z_all, torch.cat([r.Z for r in recs]))
        if n_sums:
            ref = recs[0].sums.clone()
            for r in recs[1:]:
                ref += r.sums
            assert torch.equal(total[:n_sums], ref)


@pytest.mark.parametrize("M", [1000, 10000, 50000, 100000, 300000])
def test_sums_and_risk_stats_single_launch_equals_the_two_calls(M):
    """rato_sums_and_risk_stats (ONE launch for M <= 20,480) == rato_sum_partials + rato_risk_stats, bit for bit"""
    import torch
    from riskaversetrajopt_amd import stats
    rng = np.random.RandomState(M % 97)
    Z = torch.as_tensor((rng.randn(M) * 0.5 - 0.2).astype(np.float32), device="cuda")
    part = torch.as_tensor(rng.randn((M + 63) // 64, 306).astype(np.float32), device="cuda")
    sums_a = stats.sum_partials(part)
    st_a = stats.risk_stats_device(Z, 0.1)
    ws = stats.new_workspace(M, Z.device)
    for _ in range(3):                                           # the workspace cleans itself for the next call
        sums_b, st_b = stats.sums_and_risk_stats_device(part, Z, 0.1, workspace=ws)
        assert torch.equal(sums_a, sums_b) and torch.equal(st_a, st_b)
    Zs = np.sort(Z.cpu().numpy())
    assert st_b[0].item() == float(Zs[M - int(np.floor(0.1 * M)) - 1]) == st_b[10].item()


def test_uninitialised_workspace_fails_loudly():
    import torch
    from riskaversetrajopt_amd import stats, _lib
    lib = _lib.load()
    M = 50000                                    # the paths for M > 12,288 are the ones that use the workspace
    Z = torch.randn(M, device="cuda")
    ws = torch.full((lib.rato_risk_stats_workspace_bytes(M),), 0x5A, dtype=torch.uint8, device="cuda")   # never initialised
    out = stats.risk_stats_device(Z, 0.1, workspace=ws)
    assert torch.isnan(out).all()
    good = stats.risk_stats_device(Z, 0.1)
    assert torch.isfinite(good).all()


def test_unclean_workspace_fails_loudly_instead_of_waiting_forever():
    """The one-launch path waits for its histograms to add up; a workspace that some aborted call left with counts in
    it can never add up -> bounded wait, NaN statistics, the workspace loses its tag (NaN until re-initialised)."""
    import torch
    from riskaversetrajopt_amd import stats
    M = 50000
    Z = torch.randn(M, device="cuda")
    ws = stats.new_workspace(M, Z.device)
    good = stats.risk_stats_device(Z, 0.1, workspace=ws).clone()
    assert torch.isfinite(good).all()
    ws.view(torch.int32)[100] = 7                # a stale count in the first-pass histogram
    torch.cuda.synchronize()
    bad = stats.risk_stats_device(Z, 0.1, workspace=ws)
    assert torch.isnan(bad).all()
    again = stats.risk_stats_device(Z, 0.1, workspace=ws)           # stays loud
    assert torch.isnan(again).all()
    ws2 = stats.new_workspace(M, Z.device)
    assert torch.equal(stats.risk_stats_device(Z, 0.1, workspace=ws2), good)


CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from riskaversetrajopt_amd import stats
out = {}
for M in (300, 10000, 12289, 50000, 100000, 1048576):
    Z = (np.random.RandomState(M).randn(M) * 0.05 + 0.9).astype(np.float32)
    for alpha in (0.05, 1.0):
        ws = stats.new_workspace(M, torch.device("cuda:0"))
        Zd = torch.from_numpy(Z).cuda()
        for rep in range(3):
            st = stats.risk_stats_device(Zd, alpha, workspace=ws)
        out["%%d_%%g" %% (M, alpha)] = st.cpu().numpy()
np.savez(%(path)r, **out)
'''


def test_launch_structures_agree(tmp_path):
    """default (one workgroup / one launch of a few workgroups / five launches by size), the five launches forced, the
    one-launch form forced: selection, counts and maxima identical, fp64 sums equal to summation order."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in (("default", {}), ("multi", {"RATO_RS_PATH": "multi"}), ("coop", {"RATO_RS_PATH": "coop"})):
        path = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=root, path=path)], env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = np.load(path)
    for k in res["default"].files:
        a = res["default"][k]
        for name in ("multi", "coop"):
            b = res[name][k]
            for slot in (0, 2, 4, 5, 7, 8, 9, 10):                  # VaR, fraction, max, counts, rank, t*: exact
                assert a[slot] == b[slot], (k, name, slot)
            np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-13, err_msg=f"{k} {name}")


def test_alpha_one_keeps_the_minimiser_separate_from_the_wrapped_var():
    """floor(alpha M) == M: VaR wraps to max(Z) like the reference's index -1 (drone_main_plot.py:651), while the
    Rockafellar-Uryasev minimiser t (slot 10), CVaR and the counts are those of t = min(Z)"""
    from riskaversetrajopt_amd import stats
    for M in (7, 5000, 20000, 200000):
        Z = (np.random.RandomState(M).randn(M)).astype(np.float32)
        st = stats.risk_stats(Z, 1.0)
        assert st["var"] == float(Z.max()) and st["t_star"] == float(Z.min())
        assert st["count_above_var"] + st["count_at_var"] == M
        np.testing.assert_allclose(st["cvar"], Z.astype(np.float64).mean(), rtol=1e-12, atol=1e-12)


def test_one_launch_selection_on_concurrent_streams():
    """The one-launch selection waits inside the launch for its own workgroups; calls on different streams (each with
    its own workspace) may overlap on the device, beside a bandwidth-heavy kernel, and still finish and agree."""
    import torch
    from riskaversetrajopt_amd import stats, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    Ms = (50000, 200000, 1000000, 30000)
    Zs = [0.9 + 0.05 * torch.randn(M, generator=g, device=dev) for M in Ms]
    refs = [stats.risk_stats_device(Z, 0.1).clone() for Z in Zs]
    streams = [torch.cuda.Stream() for _ in Ms]
    wss = [stats.new_workspace(M, dev) for M in Ms]
    outs = [torch.empty(stats.N_STATS, dtype=torch.float64, device=dev) for _ in Ms]
    big = torch.empty(1 << 28, device=dev)
    torch.cuda.synchronize()
    for rep in range(20):
        big.fill_(float(rep))                                    # keeps every CU busy on the default stream
        for Z, st, ws, out in zip(Zs, streams, wss, outs):
            with torch.cuda.stream(st):
                stats.risk_stats_device(Z, 0.1, workspace=ws, out=out, stream=_lib.current_stream())
    torch.cuda.synchronize()
    for out, ref in zip(outs, refs):
        assert torch.equal(out, ref)


@pytest.mark.parametrize("M", [10000, 50000, 2000000])
def test_signed_zeros_at_the_threshold(M):
    """+0.0 and -0.0 around the VaR threshold: the three launch forms (one workgroup / one launch / five launches: the
    sizes above pick one each) count #{Z > t}, #{Z == t} like the float comparisons of the consumers (cvar.hip: m > t,
    m == t) -- -0.0 and +0.0 are the same value."""
    from riskaversetrajopt_amd import stats
    rng = np.random.RandomState(M)
    Z = rng.randn(M).astype(np.float32)
    k = int(0.1 * M)                                       # alpha M samples strictly above zero, then a band of zeros
    order = np.argsort(-Z)
    Z[order[:k - 5]] = np.abs(Z[order[:k - 5]]) + 1.0
    zeros = order[k - 5:k + 6]
    Z[zeros] = np.where(np.arange(zeros.size) % 2 == 0, 0.0, -0.0).astype(np.float32)
    Z[order[k + 6:]] = -np.abs(Z[order[k + 6:]]) - 1.0
    st = stats.risk_stats(Z, 0.1)
    assert st["var"] == 0.0 and st["t_star"] == 0.0
    assert st["count_above_var"] == float(np.sum(Z > 0.0)) == k - 5
    assert st["count_at_var"] == float(np.sum(Z == 0.0)) == 11
    tail = np.sort(Z.astype(np.float64))[::-1][:k]
    np.testing.assert_allclose(st["cvar"], tail.mean(), rtol=1e-12, atol=1e-12)


def test_one_launch_selection_waits_out_a_chip_owned_by_another_stream():
    """rs_coop's workgroups wait inside the launch for each other.  With 480 of the 512 wave-slot pairs of the chip held
    for 0.8 s by another stream (rato_device_occupy), only part of the 64 workgroups of a selection over M = 1e6 can be
    resident; they keep polling until the rest has been scheduled -- the exact np.sort answer, late, instead of the
    NaN statistics the first version produced after ~0.5 s of polling."""
    import time
    import torch
    from riskaversetrajopt_amd import stats, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = 1000000
    Zh = (0.9 + 0.05 * np.random.RandomState(5).randn(M)).astype(np.float32)
    Z = torch.from_numpy(Zh).to(dev)
    ws = stats.new_workspace(M, dev)
    out = torch.empty(stats.N_STATS, dtype=torch.float64, device=dev)
    hog, sel = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(hog):
        _lib.check(lib.rato_device_occupy(480, 800000, _lib.current_stream()), "rato_device_occupy")
    time.sleep(0.02)                                            # the hog is resident before the selection is queued
    with torch.cuda.stream(sel):
        stats.risk_stats_device(Z, 0.1, workspace=ws, out=out, stream=_lib.current_stream())
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 > 0.7                       # the selection really sat behind / beside the hog
    st = out.cpu().numpy()
    srt = np.sort(Zh)
    assert st[0] == float(srt[M - int(np.floor(0.1 * M)) - 1])  # VaR: bit-identical to np.sort
    k = int(np.floor(0.1 * M))
    np.testing.assert_allclose(st[1], srt[::-1][:k].astype(np.float64).mean(), rtol=1e-12)
    again = stats.risk_stats_device(Z, 0.1, workspace=ws)       # and the workspace was left clean
    assert torch.equal(again, out)


def test_facades_recover_from_a_selection_that_gave_up():
    """a NaN record on finite input is repeated through rato_risk_stats_recover (workspace re-initialised on the stream,
    launch-per-pass selection) -- by stats.risk_stats and by the cut oracle's round trip -- and gives the exact answer"""
    import torch
    from riskaversetrajopt_amd import stats
    M = 50000
    Z = torch.randn(M, device="cuda")
    ws = stats.new_workspace(M, Z.device)
    good = stats.risk_stats(Z, 0.1, workspace=ws)
    ws.view(torch.int32)[100] = 7                # a stale count: the one-launch form gives up and un-tags the workspace
    torch.cuda.synchronize()
    assert torch.isnan(stats.risk_stats_device(Z, 0.1, workspace=ws)).all()
    ws.view(torch.int32)[100] = 7
    rec = stats.risk_stats(Z, 0.1, workspace=ws)                       # facade: recovers by itself
    for k in good:
        assert rec[k] == good[k] or abs(rec[k] - good[k]) <= 1e-12 * abs(good[k]), k      # fp64 sums: equal to summation order
    assert rec["var"] == good["var"] and rec["count_satisfied"] == good["count_satisfied"]
    again = stats.risk_stats(Z, 0.1, workspace=ws)                     # and the workspace is usable again
    assert again == good


@pytest.mark.parametrize("M", [7, 5000, 20000, 1500000])
def test_tail_of_less_than_one_sample(M):
    """floor(alpha M) = 0 (drone_main_plot.py:640-652: index M - 0 - 1 = the maximum; drone_risk.py:694: the closed form
    with an empty tail sum): VaR = CVaR's threshold = max(Z), in every launch form of the selection."""
    import torch
    from oracle import stats as ostats
    from riskaversetrajopt_amd import stats
    Z = torch.randn(M, device="cuda") * 0.3 - 0.1
    alpha = 0.5 / M
    st = stats.risk_stats(Z, alpha)
    Zh = Z.double().cpu().numpy()
    assert st["rank"] == M - 1 and st["var"] == Zh.max() and st["count_above_var"] == 0
    assert st["var"] == ostats.monte_carlo_var(Zh, alpha)
    assert abs(st["cvar"] - ostats.monte_carlo_avar(Zh, alpha)) <= 1e-12 * max(1.0, abs(st["cvar"]))
    assert st["cvar"] == st["var"]                       # empty tail: t + 0 / (alpha M)


@pytest.mark.parametrize("M", [3000, 12288, 50000])
def test_selection_with_nans_of_both_signs_and_a_key_range_near_the_full_word(M):
    """ADVICE r5 (rato_select.h): the range-normalised selection filtered candidates with the unsigned test ``key - lo <
    width``; for a key BELOW lo the subtraction wraps and passes when lo - key > 2^32 - width -- possible only when the keys
    span almost the whole 32-bit word, i.e. when Z holds NaNs of both signs (their keys sit at the two ends of the order).
    The selection must still return the key of the requested ascending rank: NaNs with the sign bit below -inf, the others
    above +inf, exactly the order of the kernels' key map, emulated here in NumPy."""
    from riskaversetrajopt_amd import stats
    rng = np.random.RandomState(M)
    Z = (rng.randn(M) * 0.5).astype(np.float32)
    n_nan = max(M // 100, 2)
    bits = Z.view(np.uint32).copy()
    idx = rng.permutation(M)
    bits[idx[:n_nan]] = 0xffc00001                       # -NaN: the smallest keys
    bits[idx[n_nan:2 * n_nan]] = 0x7fc00001              # +NaN: the largest keys
    bits[idx[2 * n_nan]] = 0xff800000                    # -inf
    bits[idx[2 * n_nan + 1]] = 0x7f800000                # +inf
    Zb = bits.view(np.float32)
    u = bits.copy()
    u[u == 0x80000000] = 0
    key = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)
    order = np.sort(key)
    for alpha in (0.1, 0.3, 0.004):
        k = max(M - int(np.floor(alpha * M)) - 1, 0)
        want = order[k]
        wu = (want & 0x7fffffff) if (want & 0x80000000) else (~want & 0xffffffff)
        want_val = np.array([wu], dtype=np.uint32).view(np.float32)[0]
        st = stats.risk_stats(Zb, alpha)
        assert st["rank"] == k
        if np.isnan(want_val):
            assert np.isnan(st["var"])
        else:
            assert st["var"] == want_val, (M, alpha, st["var"], want_val)

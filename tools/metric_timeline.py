"""Store-rate-vs-time curve of the metric launch (drone_linearize_rows_kernel<false,false>, M = 1e5, S = 50, products) from a
-DRATO_DIAG=4 build: per work unit the workgroup that ran it, its XCD, and the 100 MHz clock at unit start / noise staged /
rollout done / first row with a store picked / last row task done.  One launch on one stream, and two launches overlapping
on two streams (the bench line's `two_streams` form).

    RATO_SAA_LIB=scratch/librato_diag4.so RATO_DYN_TAIL_SPLIT=1 python tools/metric_timeline.py [M] [S]

(RATO_DYN_TAIL_SPLIT=1: whole tiles only, so that every unit has its own record; the product's default hands the last 128
tiles out as quarter tiles, which shortens the drain further -- the second run of this tool, without the variable, reports
the kernel span with them.)  The store rate is modelled per unit as its bytes spread evenly between its first store and its
end -- the stamps are issue times, the stores complete later; the launch's span by HIP events is printed beside it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from riskaversetrajopt_amd import drone_risk, drone_utils  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=7, device=dev)
d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = d._us_device(np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20 / S))
outs = [d.linearize_device(us, factored=False) for _ in range(2)]
for _ in range(5):
    for o in outs:
        d.linearize_device(us, out=o, factored=False)
torch.cuda.synchronize()
TILE_BYTES = 3 * S * (S - 1) * 64 * 4 + 3 * S * 64 * 4


def stamps(o):
    raw = o["part"].cpu().numpy()
    tl = np.ascontiguousarray(raw[:, :18]).view(np.uint64)                 # (tiles, 9)
    meta = tl[:, 5]
    global KERNEL_IN_OUT
    k_out = tl[:, 8][tl[:, 8] > 0]
    KERNEL_IN_OUT = (tl[:, 7].min() * 1e-2, (k_out.max() if len(k_out) else 0) * 1e-2, tl[:, 0].min() * 1e-2, tl[:, 6].max() * 1e-2)
    # columns: start, staged, rolled, first store, all row tasks issued, [all stores acknowledged | = issued]
    cols = tl[:, [0, 1, 2, 4, 3, 6]].astype(np.float64) * 1e-2
    return cols, (meta & 0xffffffff).astype(np.int64), ((meta >> 32) & 0xf).astype(np.int64)


def curve(label, sets, t0=None, bins=10.0):
    """sets: list of (tl us [tiles][5], bid, xcc)"""
    tl = np.concatenate([s[0] for s in sets])
    t0 = tl[:, 0].min() if t0 is None else t0
    tl = tl - t0
    start, staged, rolled, first, end, acked = tl.T
    span = end.max()
    print(f"== {label}: {len(tl)} units, span by the stamps {span:.1f} us  (first unit start .. last row task issued)")
    edges = np.arange(0.0, span + bins, bins)
    rate = np.zeros(len(edges) - 1)
    for f, e in zip(first, end):
        lo, hi = np.searchsorted(edges, [f, e], side="right") - 1
        dur = max(e - f, 1e-3)
        for b in range(max(lo, 0), min(hi, len(rate) - 1) + 1):
            ov = min(e, edges[b + 1]) - max(f, edges[b])
            if ov > 0:
                rate[b] += TILE_BYTES * ov / dur
    rate = rate / (bins * 1e-6) / 1e12                                       # TB/s per bin
    plateau = np.median(rate[len(rate) // 4: 3 * len(rate) // 4])
    print(f"   store rate (modelled), TB/s per {bins:.0f} us bin; plateau (median of the middle half) {plateau:.2f}:")
    print("   " + " ".join(f"{r:.1f}" for r in rate))
    ramp = edges[np.argmax(rate >= 0.9 * plateau)]
    below = np.flatnonzero(rate >= 0.9 * plateau)
    drain = span - edges[below[-1] + 1] if len(below) else float("nan")
    print(f"   ramp: {ramp:.0f} us until 90 % of the plateau; drain: the last {drain:.0f} us are below 90 %")
    print(f"   bytes / span = {len(tl) * TILE_BYTES / (span * 1e-6) / 1e12:.2f} TB/s; bytes / (span - ramp/2 - drain/2) = "
          f"{len(tl) * TILE_BYTES / ((span - 0.5 * ramp - 0.5 * drain) * 1e-6) / 1e12:.2f}")
    print(f"   per unit: staging {np.mean(staged - start):.1f} us (p90 {np.percentile(staged - start, 90):.1f}), "
          f"start -> first store {np.mean(first - start):.1f} (p90 {np.percentile(first - start, 90):.1f}), "
          f"first store -> end {np.mean(end - first):.1f} (p10 {np.percentile(end - first, 10):.1f}, p90 {np.percentile(end - first, 90):.1f})")
    # per workgroup: gap between a unit's end and the next unit's first store
    gaps, per_wg = [], []
    for s_ in sets:
        tls, bid, _ = s_
        tls = tls - t0
        for b in np.unique(bid):
            sel = np.flatnonzero(bid == b)
            sel = sel[np.argsort(tls[sel, 0])]
            per_wg.append(len(sel))
            gaps += list(tls[sel[1:], 3] - tls[sel[:-1], 4])
    gaps = np.array(gaps)
    print(f"   workgroups {len(per_wg)}, units per workgroup {np.min(per_wg)}..{np.max(per_wg)}; between a unit's last row task and "
          f"the next unit's first store: mean {gaps.mean():.1f} us, p50 {np.median(gaps):.1f}, p90 {np.percentile(gaps, 90):.1f} "
          f"-> {gaps.sum() / (len(per_wg) * span) * 100:.1f} % of the workgroup-time of the launch")
    first_start = start.min()
    print(f"   first store of the launch at {first.min() - first_start:.1f} us; 50 % of the workgroups storing by "
          f"{np.median([tls[bid == b, 3].min() - t0 for tls, bid, _ in sets for b in np.unique(bid)]):.1f} us")
    print(f"   all row tasks issued -> unit record written (RATO_DIAG_WAIT=1: every store acknowledged): mean {np.mean(acked - end):.2f} us, "
          f"p90 {np.percentile(acked - end, 90):.2f}, max {np.max(acked - end):.2f}; last record at {acked.max():.1f} us")
    ends = np.sort(end)
    print(f"   last unit ends: {' '.join(f'{v:.0f}' for v in ends[-8:])}; units still running in the last 20 / 10 / 5 us: "
          f"{int((end > span - 20).sum())} / {int((end > span - 10).sum())} / {int((end > span - 5).sum())}")
    xcc = np.concatenate([s[2] for s in sets])
    print("   per XCD: units " + " ".join(str(int((xcc == x).sum())) for x in range(8)) +
          "; mean unit time " + " ".join(f"{np.mean((end - start)[xcc == x]):.0f}" for x in range(8)))
    return span


a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# ---- one launch, one stream
torch.cuda.synchronize()
# the events are queued BEHIND a spin kernel, so that they bracket the launch on the GPU's clock and not the host's time
# inside the facade call (on an idle GPU the first event is taken at once and the ~35 us of Python before the launch
# would be counted into the "kernel")
torch.cuda._sleep(2_000_000)
a.record()
d.linearize_device(us, out=outs[0], factored=False)
b.record()
torch.cuda.synchronize()
ev_us = a.elapsed_time(b) * 1e3
s_one = stamps(outs[0])
print(f"one stream: launch by HIP events (queued behind a spin kernel) {ev_us:.1f} us; first unit start -> last record "
      f"{s_one[0][:, 5].max() - s_one[0][:, 0].min():.1f} us; queue split "
      f"{os.environ.get('RATO_DYN_TAIL_SPLIT', 'default (last 128 tiles as quarters)')}")
ki, ko, u0, r1 = KERNEL_IN_OUT
print(f"            first instruction of any workgroup {u0 - ki:.2f} us before the first unit's start stamp; last instruction of any "
      f"workgroup {ko - r1:.2f} us after the last record; first instruction -> last instruction {ko - ki:.1f} us "
      f"(events - that = {ev_us - (ko - ki):.1f} us of dispatch + release)")
curve("one launch, one stream", [s_one])

# ---- back to back on one stream (what `value` times): the gap between two launches
torch.cuda.synchronize()
d.linearize_device(us, out=outs[0], factored=False)
d.linearize_device(us, out=outs[1], factored=False)
torch.cuda.synchronize()
s0, s1 = stamps(outs[0]), stamps(outs[1])
print(f"two launches back to back on ONE stream: last row task of the first at {s0[0][:, 4].max() - s0[0][:, 0].min():.1f} us, "
      f"last record at {s0[0][:, 5].max() - s0[0][:, 0].min():.1f}; first unit of the second starts at "
      f"{s1[0][:, 0].min() - s0[0][:, 0].min():.1f} us, its first store at {s1[0][:, 3].min() - s0[0][:, 0].min():.1f} us")

# ---- two launches on two streams
st = [torch.cuda.Stream(), torch.cuda.Stream()]
for s_ in st:
    s_.wait_stream(torch.cuda.current_stream())
for i in range(4):
    with torch.cuda.stream(st[i & 1]):
        d.linearize_device(us, out=outs[i & 1], factored=False)
torch.cuda.synchronize()
t_host = __import__("time").perf_counter()
N2 = 8
for i in range(N2):                                  # the bench line's `two_streams` form: launches alternate between the streams
    with torch.cuda.stream(st[i & 1]):
        d.linearize_device(us, out=outs[i & 1], factored=False)
torch.cuda.synchronize()
t_host = (__import__("time").perf_counter() - t_host) / N2 * 1e6
s0, s1 = stamps(outs[0]), stamps(outs[1])            # the last launch of each stream
t0 = min(s0[0][:, 0].min(), s1[0][:, 0].min())
print(f"two streams, {N2} launches alternating: {t_host:.1f} us per launch by the host clock; the last two launches start "
      f"{abs(s1[0][:, 0].min() - s0[0][:, 0].min()):.1f} us apart")
curve("the last launch of stream 0", [s0])
curve("the last launch of stream 1", [s1])
curve("both (rate of the two together)", [s0, s1], t0=t0)

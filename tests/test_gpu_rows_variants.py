"""GPU: the launch-structure variants of the drone row kernel compute the same thing.  The dynamic tile queue (the
default for large batches), with or without part tiles at its end, is the SAME kernel code as the
one-tile-per-workgroup launch and must reproduce it bit for bit, whatever order the tiles are taken in.  The variant
is chosen by an environment variable that the library reads once, hence one subprocess per variant."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from riskaversetrajopt_amd import drone_risk, drone_utils
from riskaversetrajopt_amd.drone_risk import untile
S, M = %(S)d, %(M)d
dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=3)
d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
t = np.arange(S)[:, None]
us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)
out = {}
for fact in (False, True):
    r = d.linearize_device(us, factored=fact)
    G = untile(r["G"], M)
    out["G%%d" %% fact] = G.reshape(-1)[::97].cpu().numpy()                        # strided sample of the entries
    out["Gsum%%d" %% fact] = G.double().sum().item()
    out["gup%%d" %% fact] = r["g_up"].cpu().numpy()
    out["Z%%d" %% fact] = r["Z"].cpu().numpy()
    out["part%%d" %% fact] = r["part"].cpu().numpy()
np.savez(%(path)r, **out)
'''


def run_variant(tmp_path, name, env, S, M):
    path = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, S=S, M=M, path=path)], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return np.load(path)


@pytest.mark.parametrize("S,M", [(50, 100000), (20, 70000)])
def test_launch_structure_variants_are_bit_identical(tmp_path, S, M):
    off = {"RATO_ROWS_DYNAMIC": "0", "RATO_DYN_TAIL_SPLIT": "1"}
    base = run_variant(tmp_path, "base", off, S, M)                                  # one tile per workgroup
    for name, env in (("dynamic", dict(off, RATO_ROWS_DYNAMIC="1")),                 # global tile queue, whole tiles
                      ("dynamic_tail", dict(off, RATO_ROWS_DYNAMIC="1", RATO_DYN_TAIL_SPLIT="4")),   # + quarter tiles last
                      ("dynamic_halves", dict(off, RATO_ROWS_DYNAMIC="2", RATO_DYN_TAIL_SPLIT="2", RATO_DYN_TAIL_TILES="700")),
                      ("default", {})):
        v = run_variant(tmp_path, name, env, S, M)
        for k in base.files:
            assert np.array_equal(base[k], v[k]), (name, k)


def test_two_captured_graphs_replayed_concurrently_do_not_share_a_tile_queue():
    """torch.cuda.graph captures every graph on one shared side stream; a tile queue keyed by the launch stream would be
    shared by both captured launches, and replaying the graphs concurrently on two streams would then skip tiles
    (stale rows survive silently).  A launch recorded into a graph takes a queue of its own (rato::TileQueuePool)."""
    import torch
    from riskaversetrajopt_amd import drone_risk, drone_utils
    from riskaversetrajopt_amd.drone_risk import untile
    S, M = 20, 70000                                        # 1094 tiles >= 1024 resident slots: the queue form
    t = np.arange(S)[:, None]
    us_np = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)])
    models, refs, graphs, outs = [], [], [], []
    for seed in (3, 4):
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(M, S, seed=seed)
        d = drone_risk.Model.from_device(S, dW, mass, Qsym, 'saa', 0.1, M=M)
        us = d._us_device(us_np * (1.0 + 0.1 * seed))
        r = d.linearize_device(us)
        refs.append({k: r[k].clone() for k in ("g_up", "Z", "part")} | {"G": untile(r["G"], M).clone()})
        out = d.linearize_device(us)                       # the buffers the captured launch writes
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            d.linearize_device(us, out=out, reduce=False)
        models.append((d, us)); graphs.append(g); outs.append(out)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(15):
        for out in outs:
            out["G"].zero_(); out["_g_up"].zero_(); out["part"].zero_()
        torch.cuda.synchronize()
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
        torch.cuda.synchronize()
        for out, ref in zip(outs, refs):
            assert torch.equal(untile(out["G"], M), ref["G"]), rep
            assert torch.equal(out["g_up"], ref["g_up"]) and torch.equal(out["part"], ref["part"]), rep
    # an eager launch on the capture-time stream beside a replay does not share a queue with it either
    d, us = models[0]
    with torch.cuda.stream(streams[1]):
        graphs[1].replay()
    r = d.linearize_device(us)
    torch.cuda.synchronize()
    assert torch.equal(untile(r["G"], M), refs[0]["G"]) and torch.equal(untile(outs[1]["G"], M), refs[1]["G"])



def test_streaming_stores_change_no_output(tmp_path):
    """RATO_NT_STORES=2 (the Jacobian written with non-temporal stores at every size) against RATO_NT_STORES=0 (never): the
    launcher's choice between the two is a matter of cache residency only -- every output must be bit for bit the same.
    (The switch is read once per process: two child processes.)"""
    import subprocess
    import sys
    script = tmp_path / "nt_digest.py"
    script.write_text(
        "import sys, hashlib, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from riskaversetrajopt_amd import drone_risk, drone_utils, driving\n"
        "h = hashlib.sha256()\n"
        "for M, S, fact in ((10007, 50, False), (3000, 20, True)):\n"
        "    dW, mass, Q = drone_utils.sample_uncertain_parameters_device(M, S, seed=9)\n"
        "    d = drone_risk.Model.from_device(S, dW, mass, Q, 'saa', 0.1, M=M)\n"
        "    t = np.arange(S)[:, None]\n"
        "    us = np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S)\n"
        "    r = d.linearize_device(us, factored=fact)\n"
        "    for a in (drone_risk.untile(r['G'], M), r['g_up'], r['Z'], r['sums']):\n"
        "        h.update(a.cpu().numpy().tobytes())\n"
        "for M, S in ((10000, 40), (300, 20)):\n"
        "    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=9)\n"
        "    c = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', 0.05)\n"
        "    t = np.arange(S)[:, None]\n"
        "    us = np.hstack([0.4 * np.cos(0.3 * t) + 0.1, 0.03 * np.sin(0.5 * t) + 0.004]) * (20.0 / S)\n"
        "    r = c.linearize_device(us)\n"
        "    for a in (driving.untile(r['G'], M), r['g_up'], r['Z']):\n"
        "        h.update(a.cpu().numpy().tobytes())\n"
        "print(h.hexdigest())\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    digests = []
    for v in ("0", "2"):
        env = dict(os.environ, RATO_NT_STORES=v)
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(out.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1] and len(digests[0]) == 64

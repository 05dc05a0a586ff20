#!/usr/bin/env python3
"""bench.py — SAA inner-loop throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic samples
resident in HBM: batched rollout -> control-Jacobian linearization -> sample
mean -> chance-constraint / VaR / CVaR statistics (what one SCP iteration asks
of the SAA inner loop).  Default workload = the configuration the metric is
quoted on: drone_risk, M = 1e5 samples per GPU, S = 50 steps, fp32.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the dominant kernel's algorithmic HBM bytes / its launch time
                 measured live with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline — the oracle ("port": C + OpenMP for the drone, NumPy otherwise) timed on
                 this box's host cores, on a bounded sample of the same workload (rank 0, N=1 only).
Other workloads (--workload driving|hopper, --M, --S, --mode eval) are for
sweeps; they print the same line shape.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# diagnostic (EXPERIMENTS 6): time the K steps WITHOUT the HIP events around every launch -- what the events themselves cost;
# the line then has no kernel time (`roofline` is NaN) and is not a valid bench line
NO_EVENTS = bool(os.environ.get("RATO_BENCH_NO_EVENTS"))
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
B = 4                       # fp32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=["drone", "driving", "hopper"])
    ap.add_argument("--mode", default="linearize", choices=["linearize", "eval"])
    ap.add_argument("--M", type=int, default=0, help="samples per GPU (default: 1e5 drone/driving, 5e4 hopper)")
    ap.add_argument("--S", type=int, default=0, help="steps (default: 50 drone, 40 driving, 60 hopper)")
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--cols-per-thread", type=int, default=0, help="0 auto; -1 row-parallel kernel (drone)")
    ap.add_argument("--samples-per-lane", type=int, default=0)
    ap.add_argument("--config", default="metric", choices=["metric", "C2", "C3", "C4", "C5"],
                    help="BASELINE.json configuration: metric = drone M=1e5 S=50 (the one the metric is quoted on); "
                         "C2 drone M=1e4 S=50; C3 driving M=1e4 S=40; C4 hopper M=5e4 S=60; C5 driving 125,000 "
                         "samples per GPU, S=40 (M=1e6 over 8 GPUs).  --workload/--M/--S override.")
    ap.add_argument("--jacobian", default="both", choices=["both", "products", "factored", "regenerated"],
                    help="drone linearize: representation the kernel writes.  products = every structural nonzero, "
                         "3S(S-1) numbers per sample (SURVEY 8d; this is what `value` and `roofline` are quoted on); "
                         "factored = W[j,t,a] * Phi[t,s,a], S(S-1)+6S numbers (reported as *_factored); regenerated = the "
                         "products output with the noise regenerated in the kernel instead of read (*_regenerated); both "
                         "= one timed region for each of the three, same samples")
    ap.add_argument("--philox", action="store_true",
                    help="drone, driving: no noise array in HBM, the Brownian increments are regenerated inside the "
                         "kernels (Philox4x32-10: rato_*_eval_philox, rato_*_linearize_philox); the algorithmic bytes "
                         "lose the 12 (8) B per sample-step of the noise")
    ap.add_argument("--no-scp", action="store_true", help="skip the SCP wall-clock block (drone, N=1)")
    ap.add_argument("--scp-iters", type=int, default=60)
    ap.add_argument("--dry-run", action="store_true", help="rank start-up + barrier only (no GPU work)")
    ap.add_argument("--strict-comm", action="store_true",
                    help="N > 1: exit non-zero if the library's own RCCL communicator (rato_comm_init) cannot be created "
                         "on every rank, instead of falling back to torch.distributed's collective with a warning")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="N=1: replay the step as ONE captured hipGraph (kernel time then comes from an eager pre-pass "
                         "with HIP events, since events cannot bracket a node inside a graph).  auto = for "
                         "M <= 50,000 (BASELINE C2-C4) probe replayed against eager back-to-back steps and time the faster form")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=None,
                    help="run the exchange + risk statistics of step i on a side stream while the hot kernel of step "
                         "i+1 runs (two output slots; dist.PipelinedSteps).  N > 1 without either flag: both forms are "
                         "probed over 50 untimed steps and the faster one (slowest rank's clock) is timed.  Off by default "
                         "at N = 1, where it gains nothing (DESIGN.md 5)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="N > 1: kernel -> sums -> all-gather -> selection serially on one stream (A/B against the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE C2-C5 as whole steps after the metric configuration)")
    ap.add_argument("--cpu-samples", type=int, default=0)
    args = ap.parse_args()
    preset = {"metric": ("drone", 100000, 50), "C2": ("drone", 10000, 50), "C3": ("driving", 10000, 40),
              "C4": ("hopper", 50000, 60), "C5": ("driving", 125000, 40)}[args.config]
    if args.workload is None:
        args.workload = preset[0]
    if args.workload == preset[0]:
        args.M = args.M or preset[1]
        args.S = args.S or preset[2]
    return args


def graze_us(S, n_u):
    t = np.arange(S)[:, None]
    if n_u == 3:
        return (np.hstack([0.6 * np.cos(0.3 * t) + 0.3, 0.15 * np.sin(0.5 * t) + 0.02, 0.05 * np.cos(t)]) * (20.0 / S))
    return np.hstack([0.4 * np.cos(0.4 * t) - 0.2, 0.05 * np.sin(0.35 * t) + 0.01]) * (20.0 / S)


# ----------------------------------------------------------------- workloads
class DroneWork:
    name = "drone_risk"
    kernel = "drone_linearize_kernel"

    def __init__(self, args, device, seed):
        from riskaversetrajopt_amd import drone_risk, drone_utils
        self.S = args.S or 50
        self.M = args.M or 100000
        self.mode = args.mode
        self.cpt, self.spl = args.cols_per_thread, args.samples_per_lane
        self.fact = False if args.packed_products else (True if getattr(args, "force_factored", False) else None)
        self.philox = bool(getattr(args, "philox", False))      # noise regenerated in the kernels instead of read
        dW, mass, Qsym = drone_utils.sample_uncertain_parameters_device(self.M, self.S, seed=seed, device=device,
                                                                        want_dW=not self.philox)
        self.model = drone_risk.Model.from_device(self.S, dW, mass, Qsym, 'saa', args.alpha, M=self.M,
                                                  noise_seed=seed if self.philox else None)
        self.us = self.model._us_device(graze_us(self.S, 3))
        self.out = None
        self.retile_us = None
        if self.mode == "linearize":
            if not self.philox:
                # the one-time re-tiling of the batch's noise (what the row kernel reads): setup, timed here so that the line
                # can say what it costs -- once per batch, then every linearization of the batch reads the copy
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.model._tiled_noise(self.model._dW, self.M, self.model._mass.numel())     # (cold: loads the kernel)
                self.model.invalidate_noise()
                torch.cuda.synchronize()
                a.record()
                tiled = self.model._tiled_noise(self.model._dW, self.M, self.model._mass.numel())
                b.record()
                torch.cuda.synchronize()
                if tiled is not None:
                    self.retile_us = a.elapsed_time(b) * 1e3
            r = self.model.linearize_device(self.us, cols_per_thread=self.cpt, samples_per_lane=self.spl,
                                            factored=self.fact)
            self.fact = r["factored"]
            # two output slots, reused alternately: the statistics of step i (side stream) read Z / sums of slot
            # i%2 while step i+1 writes the other slot
            self.outs = [r, self.model.linearize_device(self.us, cols_per_thread=self.cpt, samples_per_lane=self.spl,
                                                        factored=self.fact)]
            self.variant = "cols_per_thread=%d samples_per_lane=%d jacobian=%s%s" % (
                r["cols_per_thread"], r["samples_per_lane"], "factored(W,Phi)" if self.fact else "products",
                " noise=regenerated(Philox4x32-10)" if self.philox else "")
            self.kernel = "drone_linearize_rows_kernel" if r["cols_per_thread"] == -1 else "drone_linearize_kernel"
            # the kernels write Z and the partial sums straight into the record the all-gather sends (dist.Record)
            from riskaversetrajopt_amd import dist as rdist
            self.records = []
            for o in self.outs:
                ld = o["_Z"].numel()
                rec = rdist.Record(6 * self.S + 6, self.M, device, z_row=ld)
                o["_Z"], o["sums"] = rec.Z_row[:ld], rec.sums
                self.records.append(rec)
        else:
            self.kernel = "drone_eval_kernel<philox>" if self.philox else "drone_eval_tiles_kernel"

    def stats_in_launch(self):
        """small batches: the linearize launch itself carries the statistics of its Z (rato_saa.h: params.stats_*)"""
        if self.mode == "eval":      # the Monte-Carlo step: tiled rollout + exact selection in ONE launch (small batches)
            return not self.philox and bool(self.model._lib.rato_drone_eval_stats_in_launch(self.M))
        return (self.mode == "linearize" and self.kernel == "drone_linearize_rows_kernel"
                and bool(self.model._lib.rato_drone_stats_in_launch(self.M, self.S)))

    def hot_kernel(self, events=None, slot=0, reduce=True, stats_request=None):
        """One pass; ``events`` bracket ONLY the dominant kernel's launch.  ``reduce=False``: the partial sums stay
        unreduced (the single-GPU step folds their reduction into the launch of the risk statistics).
        ``stats_request`` = (workspace, record, alpha): the statistics ride in the linearize launch (small batches)."""
        if self.mode == "linearize":
            return self.model.linearize_device(self.us, cols_per_thread=self.cpt, samples_per_lane=self.spl,
                                               out=self.outs[slot], events=events, factored=self.fact, reduce=reduce,
                                               stats_request=stats_request)
        if events is not None:
            events[0].record()
        self.eval_bufs = getattr(self, "eval_bufs", [{}, {}])
        Z, _, _ = self.model.eval_device(self.us, out=self.eval_bufs[slot], stats_request=stats_request)
        if events is not None:
            events[1].record()
        return {"Z": Z, "du_sum": None}

    def sums(self, r):
        import torch
        if r["du_sum"] is None:
            if getattr(self, "_no_sums", None) is None:
                self._no_sums = torch.zeros(1, dtype=torch.float64, device=r["Z"].device)
            return self._no_sums
        return r["sums"]

    def algorithmic_bytes(self):
        M, S = self.M, self.S
        eval_in = M * B * (3 * S + 1 + 9 + 1) + 3 * S * B          # dW, mass, Qsym, Z | us
        if self.mode == "eval":
            return eval_in - (M * B * 3 * S if self.philox else 0)
        nblk = (M + 255) // 256
        # Jacobian as written: products (3S(S-1) numbers per sample) or its two factors
        # Phi[t,s,a] (S(S-1)) and W[j,t,a] (6S) -- the bytes the launch really has to move
        jac = (S * (S - 1) + 6 * S) if self.fact else 3 * S * (S - 1)
        noise = M * B * 3 * S if self.philox else 0                            # not read when it is regenerated
        return eval_in - noise + M * B * (3 * S + jac) + nblk * (6 * S + 6) * B       # g_up, Jacobian | partials

    def needed_bytes(self):
        """eval: what the tiled kernel has to move -- the vertical axis' noise is read by no obstacle row
        (drone_risk.py:174) and is not loaded: 8 S + 44 bytes per sample instead of SURVEY 8(d)'s 12 S + 44"""
        if self.mode != "eval" or self.philox:
            return None
        return self.M * B * (2 * self.S + 1 + 9 + 1) + 3 * self.S * B

    def cpu_baseline(self, n, alpha):
        """The oracle's C restatement (oracle/saa_oracle.c, OpenMP over samples) on the same workload at the SAME M:
        every sample's dense linearization is formed in the reference's shapes ((3,S,3S) Jacobian rows, g_up, the
        final rows) in a per-thread buffer and reduced on the fly to what the SCP consumes (sample sums, Z) -- the
        reference's (M,3,S,3S) array itself is 18 GB at M = 1e5.  -> step(nthreads) callable; ``step.numpy`` = the
        NumPy restatement (1 process) on 2,000-sample chunks of the same batch."""
        from oracle import c_oracle, drone as od, stats as ostats
        rng = np.random.RandomState(0)
        DWs, masses, obs_Qs = od.sample_uncertain_parameters(rng, 'saa', M=n, S=self.S)
        us = graze_us(self.S, 3)
        dt = od.T / self.S

        def step(nthreads):
            if self.mode == "linearize":
                c = c_oracle.drone_stream(us, DWs, masses, obs_Qs, dt, nthreads=nthreads)
                c["sum_final_du"] / n, c["sum_val_final"] / n
            else:
                c = c_oracle.drone(us, DWs, masses, obs_Qs, dt, nthreads=nthreads, want=("Z",))
            Z = c["Z"]
            return ostats.monte_carlo_var(Z, alpha), ostats.monte_carlo_avar(Z, alpha), np.mean(Z <= 1e-6)

        def numpy_chunk(k, chunk=2000):
            """NumPy oracle (vectorised over the samples of one chunk, fp64, dense outputs) on chunk k"""
            lo = (k * chunk) % max(n - chunk + 1, 1)
            sl = slice(lo, min(lo + chunk, n))
            o = od.Model(self.S, DWs[sl], masses[sl], obs_Qs[sl], 'saa', alpha)
            if self.mode == "linearize":
                fdu, flo, _, gdu, gup = o.get_all_constraints_coeffs(us)
                fdu.mean(0), flo.mean(0)
            o.monte_carlo_no_collisions_constraint_verification(us)
            return sl.stop - sl.start
        step.threaded = True
        step.numpy = numpy_chunk
        return step


class DrivingWork:
    name = "driving"
    kernel = "car_linearize_kernel"

    def __init__(self, args, device, seed):
        from riskaversetrajopt_amd import driving
        self.S = args.S or 40
        self.M = args.M or 100000
        self.mode = args.mode
        self.cpt = args.cols_per_thread
        self.philox = bool(getattr(args, "philox", False))
        dW, x0, ws, wr = driving.sample_uncertain_parameters_device(self.M, self.S, seed=seed, device=device,
                                                                    want_dW=not self.philox)
        self.model = driving.Model.from_device(self.S, dW, x0, ws, wr, 'saa', args.alpha,
                                               noise_seed=seed if self.philox else None)
        self.us = self.model._us_device(graze_us(self.S, 2))
        self.out = None
        if self.mode == "linearize":
            r = self.model.linearize_device(self.us, cols_per_thread=self.cpt)
            keys = ("G", "g_up", "Z", "final_du", "final_rhs")
            r2 = self.model.linearize_device(self.us, cols_per_thread=self.cpt)
            self.outs = [{k: r[k] for k in keys}, {k: r2[k] for k in keys}]
            from riskaversetrajopt_amd import dist as rdist
            self.records = []                                # Z is written straight into the all-gather's send buffer
            for o in self.outs:
                rec = rdist.Record(0, self.M, device)
                o["Z"] = rec.Z
                self.records.append(rec)
            self.variant = "cols_per_thread=%d%s" % (r["cols_per_thread"],
                                                     " noise=regenerated(Philox4x32-10)" if self.philox else "")
            self.kernel = "car_linearize_rows_kernel" if r["cols_per_thread"] == -1 else "car_linearize_kernel"
        else:
            self.kernel = "car_eval_kernel<philox>" if self.philox else ("car_eval_tiles_kernel" if self.M <= (1 << 20) else "car_eval_kernel")

    def stats_in_launch(self):
        if self.mode == "eval":
            return not self.philox and bool(self.model._lib.rato_car_eval_stats_in_launch(self.M))
        return (self.mode == "linearize" and self.kernel == "car_linearize_rows_kernel"
                and bool(self.model._lib.rato_car_stats_in_launch(self.M, self.S)))

    def hot_kernel(self, events=None, slot=0, stats_request=None):
        if events is not None:
            events[0].record()
        if self.mode == "linearize":
            r = self.model.linearize_device(self.us, cols_per_thread=self.cpt, out=self.outs[slot],
                                            stats_request=stats_request)
        else:
            self.eval_bufs = getattr(self, "eval_bufs", [{}, {}])
            r = {"Z": self.model.eval_device(self.us, out=self.eval_bufs[slot], stats_request=stats_request)[0]}
        if events is not None:
            events[1].record()
        return r

    def sums(self, r):
        import torch
        if getattr(self, "_no_sums", None) is None:          # final rows are sample independent: nothing to sum
            self._no_sums = torch.zeros(1, dtype=torch.float64, device=r["Z"].device)
        return self._no_sums

    def algorithmic_bytes(self):
        M, S = self.M, self.S
        eval_in = M * B * (2 * S + 6 + 1) + 2 * S * B
        if self.mode == "eval":
            return eval_in - (M * B * 2 * S if self.philox else 0)
        return eval_in - (M * B * 2 * S if self.philox else 0) + M * B * (S + S * (S - 1))

    def cpu_baseline(self, n, alpha):
        from oracle import driving as ocar, stats as ostats
        o = ocar.Model(*ocar.sample_uncertain_parameters(np.random.RandomState(0), n, 'saa', self.S))
        us = graze_us(self.S, 2)

        def step():
            if self.mode == "linearize":
                o.get_all_constraints_coeffs(us)
            _, Z = o.monte_carlo_separation_constraints_verification(us)
            return ostats.monte_carlo_var(Z, alpha), ostats.monte_carlo_avar(Z, alpha)
        return step


# hopper: px / forces by value in the kernel arguments (the eager product path); RATO_BENCH_STAGED=1: pinned upload
STAGED = bool(os.environ.get("RATO_BENCH_STAGED"))


class HopperWork:
    name = "hopper"
    kernel = "hopper_slip_kernel"

    def __init__(self, args, device, seed):
        from riskaversetrajopt_amd import hopper
        self.S = args.S or 60
        self.M = args.M or 50000
        self.mode = args.mode
        a, th, tau = hopper.sample_friction_fields_device(self.M, seed=seed + 1, device=device)
        self.model = hopper.Model.from_device(a, th, tau, 'saa', args.alpha, S=self.S)
        tj, tl = hopper.phase_times(self.S)
        self.C = tj + (self.S - tl)
        rng = np.random.RandomState(5)
        self.px = np.linspace(0.0, 0.2, self.C)
        fz = 32.0 + rng.randn(self.C)
        self.forces = np.stack([0.08 * fz + 0.3 * rng.randn(self.C), fz], axis=1)
        import torch
        self.lam = torch.rand((self.C, self.M), device=device)

    def hot_kernel(self, events=None, slot=0, reduce=True):
        if events is not None:
            events[0].record()
        if self.mode == "linearize":
            r = self.model.slip_device(self.px, self.forces, lam=self.lam, want_deriv=True, reduce=reduce, staged=STAGED)
        else:
            r = self.model.slip_device(self.px, self.forces, want_h=False, staged=STAGED)
        if events is not None:
            events[1].record()
        return r

    def sums(self, r):
        import torch
        if r.get("hess") is None:
            if getattr(self, "_no_sums", None) is None:
                self._no_sums = torch.zeros(1, dtype=torch.float64, device=r["Z"].device)
            return self._no_sums
        return r["hess"].reshape(-1)

    def algorithmic_bytes(self):
        M, C = self.M, self.C
        if self.mode == "eval":
            return M * B * (90 + 1)
        return M * B * (90 + 1) + M * C * 4 * B       # h, dh_dfz, dh_dpx out + lam in

    def cpu_baseline(self, n, alpha):
        from oracle import hopper as oh, stats as ostats
        o = oh.Model(*oh.sample_friction_fields(np.random.RandomState(1), n), method='saa', alpha=alpha, S=self.S)
        lam = np.random.RandomState(2).rand(n, self.C)

        def step():
            if self.mode == "linearize":
                o.slip_partials(self.px, self.forces)
                o.slip_hessian_sums(self.px, self.forces, lam)
            _, Z = o.no_slip_constraints_verification(self.px, self.forces)
            return ostats.monte_carlo_var(Z, alpha), ostats.monte_carlo_avar(Z, alpha)
        return step


WORKLOADS = {"drone": DroneWork, "driving": DrivingWork, "hopper": HopperWork}
CPU_SAMPLES = {"drone": 0, "driving": 3000, "hopper": 50000}      # 0: the workload's own M (streaming C oracle)


def pmc_traffic(workload, mode, M, S, jacobian=None):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC profile, if one matches — measured in a
    separate rocprofv3 --pmc run of this same command (profiles/README.md), not in this run; reported as
    ``traffic_from_profile`` together with the file it came from."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        for rec in reversed(json.load(open(path))):          # later entries = later rounds
            if (rec["workload"], rec["mode"], rec["M"], rec["S"]) == (workload, mode, M, S) \
                    and rec.get("jacobian") == jacobian:
                return {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "file": "profiles/pmc_traffic.json",
                        "profile": rec.get("profile", rec.get("correction", ""))}
    except Exception:
        pass
    return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves, as a CHILD process
    (``python -m torch.distributed.run``), before anything in this process touches the GPU; the parent only relays
    the exit code (a process that has initialised the GPU must never exec another program on this pool)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def board_info():
    """Serial number, VBIOS and HBM vendor of the visible GPU from sysfs, for the ``device`` block of the line: boxes
    of one pool run the same binary several percent apart.  Plain file reads (no child process: under rocprofv3 the
    GPU is initialised before this program starts, and such a process must not exec anything); failures are swallowed."""
    import glob
    try:
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device"), key=lambda x: int("".join(c for c in x.split("/")[4] if c.isdigit()) or 0)):
            try:
                info = {}
                for key, f in (("serial", "serial_number"), ("vbios", "vbios_version"), ("hbm_vendor", "mem_info_vram_vendor")):
                    info[key] = open(os.path.join(d, f)).read().strip()
                cards.append(info)
            except OSError:
                continue
        if not cards:
            return None
        return cards[min(int(os.environ.get("LOCAL_RANK", "0")), len(cards) - 1)]
    except Exception:
        return None


def roofline_block(work, kern_ms, workload, mode, M, S, jacobian, kern_src):
    alg = work.algorithmic_bytes()
    achieved = alg / (kern_ms * 1e-3) / 1e9
    tr = pmc_traffic(workload, mode, M, S, jacobian)
    needed = work.needed_bytes() if hasattr(work, "needed_bytes") else None
    extra = {}
    if needed:      # SURVEY 8(d)'s figure counts an input the kernel does not need: the fraction on the bytes it moves
        extra = {"bytes_needed_per_launch": needed, "achieved_on_bytes_needed": needed / (kern_ms * 1e-3) / 1e9,
                 "frac_on_bytes_needed": needed / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    return {**extra, "bound": "hbm", "kernel": work.kernel, "variant": getattr(work, "variant", ""),
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "frac_of_measured_copy_6290": achieved / 6290.0,
            # store-only replay of the row kernels' pattern (512 resident workgroups, one 1.88 MB tile each, tiles on
            # 2 MiB boundaries; tools/store_pattern5.hip): 5680-5740 GB/s; a linear fill: 6900-7000
            "frac_of_store_only_replay_5700": (achieved / 5700.0) if mode == "linearize" else None,
            "algorithmic_bytes_per_launch": alg, "bytes_per_sample_step": alg / (M * S),
            "kernel_ms": kern_ms, "kernel_ms_source": kern_src,
            # HBM bytes per launch from the PMC counters: they cannot be collected inside this run (rocprofv3 --pmc is
            # a separate pass of the same command), so this is the committed pass for this workload, or null
            "traffic": (tr or {}).get("hbm_bytes_per_launch"),
            "traffic_over_algorithmic": ((tr["hbm_bytes_per_launch"] / alg) if tr else None),
            "traffic_from_profile": tr}


def timed_region(work, args, world, rank, device, stats, rdist, dist, torch, probe_clock=True, two_streams=False):
    """W warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize; -> dict."""
    M = work.M
    stats_out = torch.empty((2, stats.N_STATS), dtype=torch.float64, device=device)
    wss = [stats.new_workspace(M * world, device), stats.new_workspace(M * world, device)]

    def barrier():
        if dist.is_initialized():
            dist.barrier()

    selfcheck = None
    if dist.is_initialized():
        rdist.check_equal_shards(M)       # collective, once, outside the timed region: every step's all-gather relies on it
        # The first multi-GPU run validates the exchange it is about to time (dist.comm_selfcheck): one untimed step, then
        # the product exchange of its record against torch.distributed's own all-gather of the same bytes, bit for bit, and
        # the gathered Z / rank-ordered totals identical on every rank.  A mismatch fails the run on every rank.
        r0 = work.hot_kernel(slot=0)
        recs = getattr(work, "records", None)
        if recs:
            rec = recs[0]
        else:
            sums0 = work.sums(r0).reshape(-1).to(torch.float64)
            rec = rdist.Record(sums0.numel(), M, device)
            rec.sums.copy_(sums0)
            rec.Z.copy_(r0["Z"][:M])
        torch.cuda.synchronize()
        selfcheck = rdist.comm_selfcheck(rec)
        strict = os.environ.get("RATO_STRICT_COMM") == "1"
        if not selfcheck["ok"] or (strict and selfcheck["rccl_ranks"] != world):
            if rank == 0:
                print(f"bench.py: exchange self-check FAILED: {json.dumps(selfcheck)}", file=sys.stderr)
            barrier()
            raise SystemExit(3)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    eager_small, launch_probe = False, None
    use_graph = world == 1 and not args.overlap and (args.graph == "on" or (args.graph == "auto" and M <= 50000))
    # --overlap: the exchange + statistics of step n beside the hot kernel of step n+1 (dist.PipelinedSteps).  N > 1
    # default ("auto"): BOTH forms are probed over untimed steps after the warm-up and the faster one, by the slowest
    # rank's clock, is timed -- beside a store-saturated kernel the selection over M_total samples runs several times
    # slower and takes bandwidth from it, and which form wins differs from board to board (round 5: 178 serial / 191
    # pipelined on the driver's board, 180 / 176 on another).
    auto_form = args.overlap == "auto" and not use_graph
    pipelined = bool(args.overlap) and not use_graph
    pipe = rdist.PipelinedSteps(2, device, high_priority=bool(os.environ.get("RATO_PIPE_PRIORITY"))) if pipelined else None
    form_probe = None
    counter = [0]
    in_launch = (world == 1 and not pipelined and hasattr(work, "stats_in_launch") and work.stats_in_launch()
                 and not os.environ.get("RATO_BENCH_NO_IN_LAUNCH"))

    def finish(slot, r):
        """sample sums -> [one all-gather of the record when N > 1] -> exact VaR / CVaR / fraction satisfied"""
        sums = work.sums(r)
        if getattr(work, "records", None):                # zero-copy record: [sums | Z] already in place
            sums, Z_all = rdist.exchange_record(work.records[slot])
        else:
            sums, Z_all = rdist.exchange(sums, r["Z"], agreed=True)   # the one collective (no-op at N=1; equal shards
            #                                                            were verified once, above)
        stats.risk_stats_device(Z_all, args.alpha, workspace=wss[slot], out=stats_out[slot])
        return sums

    def step(i=None):
        """One pass of the hot path: dominant kernel -> partial sums -> [one all-gather of the record when N > 1] ->
        exact VaR / CVaR / fraction satisfied.  (Pipelined: the exchange + statistics of step n run on a side stream
        beside the hot kernel of step n+1; both streams are drained before the clock stops.)"""
        if pipelined:
            slot = pipe.step(lambda s: work.hot_kernel(events=ev[i] if i is not None else None, slot=s), finish)
            counter[0] += 1
            return None
        slot = counter[0] & 1
        counter[0] += 1
        if in_launch:   # small batches: the kernel's own launch carries the statistics of its Z (+ the sample sums behind it)
            r = work.hot_kernel(events=ev[i] if i is not None else None, slot=slot,
                                stats_request=(wss[slot], stats_out[slot], args.alpha))
            return work.sums(r)
        fold = (world == 1 and isinstance(work, (DroneWork, HopperWork)) and work.mode == "linearize"
                and not os.environ.get("RATO_BENCH_NO_FOLD"))
        if fold:       # single GPU: linearize, then ONE launch for the sample sums + VaR / CVaR (2 launches per step)
            r = work.hot_kernel(events=ev[i] if i is not None else None, slot=slot, reduce=False)
            sums, _ = stats.sums_and_risk_stats_device(r["part"], r["Z"], args.alpha, workspace=wss[slot],
                                                       sums_out=r.get("sums"), out=stats_out[slot])
            return sums
        r = work.hot_kernel(events=ev[i] if i is not None else None, slot=slot)
        return finish(slot, r)

    # setup, not part of the W warm-up steps or of the timed region: ~10 ms of the hot kernel so that a short run
    # (e.g. K = 20, W = 3) does not time the first launches at idle clocks (measured: 0.301 vs 0.289 ms per step)
    for _ in range(40):
        work.hot_kernel(slot=0)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    if pipe is not None:
        pipe.drain()
    if auto_form:
        def probe_form(p, n=50):
            nonlocal pipelined
            pipelined = p
            for _ in range(5):
                step()
            if p:
                pipe.drain()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(n):
                step()
            if p:
                pipe.drain()
            torch.cuda.synchronize()
            t_ = torch.tensor([(time.perf_counter() - t0_) / n * 1e6], dtype=torch.float64,
                              device=device if (dist.is_initialized() and dist.get_backend() == "nccl") else "cpu")
            if dist.is_initialized():
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)      # the slowest rank's figure, identical on every rank:
            return float(t_.item())                            # every rank takes the same decision
        form_probe = {"serial_us": probe_form(False), "pipelined_us": probe_form(True)}
        pipelined = form_probe["pipelined_us"] < form_probe["serial_us"]
        form_probe["timed"] = "pipelined" if pipelined else "serial"
    if use_graph:
        # eager pre-pass: per-launch kernel time with HIP events.  The launches are queued BEHIND a spin kernel long
        # enough for the host to issue all of them, so the events bracket back-to-back GPU work and the host's issue
        # rate (30-100 us per eager step, more than these kernels) does not leak into the kernel time.
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); torch.cuda._sleep(2_000_000); b.record(); torch.cuda.synchronize()
        cycles_per_ms = 2_000_000 / max(a.elapsed_time(b), 1e-3)
        torch.cuda._sleep(int(cycles_per_ms * min(200.0, 5.0 + 0.3 * args.steps)))
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        kern_ms_eager = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        counter[0] = 0
        graph = torch.cuda.CUDAGraph()     # capture/replay plumbing only: the nodes are this library's kernels
        with torch.cuda.graph(graph):
            step()
        # setup (untimed): ~10 ms of replays, so that the K timed replays (often < 1 ms in total) do not run at the
        # clocks the capture pause and the spin kernel left behind; then the W warm-up steps proper
        for _ in range(200 + args.warmup):
            graph.replay()
        torch.cuda.synchronize()
        if args.graph == "auto":
            # --graph auto: the timed steps are replayed or issued eagerly (back to back, no events: the kernel time comes
            # from the pre-pass above), whichever a probe of both finds faster on this host.  A replay costs ~10 us + 3.4 us
            # per node before any work on this stack (configs.launch_floor); an eager step costs the device's launch
            # spacing (~4-5 us per kernel) as long as the host issues faster than the device works.
            def probe(fn, n=150):
                for _ in range(30):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e6
            launch_probe = {"replay_us": probe(graph.replay), "eager_us": probe(step)}
            if launch_probe["eager_us"] < launch_probe["replay_us"]:
                eager_small = True
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    two = None
    if two_streams and not use_graph and not pipelined:   # consecutive (independent) steps on alternating streams: the
        two = [torch.cuda.Stream(), torch.cuda.Stream()]  # line's `two_streams` block, never its `value`
        for s_ in two:
            s_.wait_stream(torch.cuda.current_stream())
        for i in range(6):                                 # untimed: the first launches on a new stream (its tile queue)
            with torch.cuda.stream(two[i & 1]):
                step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if use_graph and not eager_small:
            graph.replay()
        elif eager_small:
            step()
        elif two is not None:
            with torch.cuda.stream(two[i & 1]):
                step()                                   # (no events: two kernels share the chip, their brackets mean nothing)
        else:
            step(None if NO_EVENTS else i)
    if pipe is not None:
        pipe.drain()                                     # the last step's exchange + statistics: inside the timed region
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = kern_ms_eager if use_graph else (float("nan") if (two is not None or NO_EVENTS) else
                                               float(np.mean([a.elapsed_time(b) for a, b in ev])))
    if os.environ.get("RATO_BENCH_TRACE") and not use_graph and two is None and rank == 0:      # per-launch kernel times, to stderr
        ts = [a.elapsed_time(b) for a, b in ev]
        print("kernel ms per launch:", " ".join("%.4f" % t for t in ts), file=sys.stderr)
    sclk = None
    if rank == 0 and probe_clock:
        # diagnostic, outside the timed region: the shader clock the device sustains WHILE the hot kernel runs (one wave
        # on a second stream reads the cycle counter against the 100 MHz counter; rato_device_clock_probe).  Boxes of the pool run this same
        # binary 5-10 % apart while their store-only ceilings agree to 2 %.
        try:
            from riskaversetrajopt_amd import _lib
            lib = _lib.load()
            probe = torch.zeros((3, 3), dtype=torch.float64, device=device)
            side2 = torch.cuda.Stream()
            probe_us = int(max(5, min(2000, kern_ms * 1e3 * 0.5)))                       # ~ half a kernel long
            for i in range(3):
                work.hot_kernel(slot=0)
                with torch.cuda.stream(side2):
                    _lib.check(lib.rato_device_clock_probe(_lib.ptr(probe[i]), probe_us, _lib.current_stream()), "clock probe")
                torch.cuda.synchronize()
            sclk = float(np.median(probe[:, 0].cpu().numpy()))
        except Exception as e:                          # a diagnostic must never take the bench line down
            print(f"note: clock probe skipped ({e})", file=sys.stderr)
    final_stats = stats_out[(counter[0] - 1) & 1].cpu().numpy()
    launch = (("eager steps back to back, one stream (probe: %.1f us per eager step, %.1f replayed)" % (launch_probe["eager_us"], launch_probe["replay_us"]))
              if eager_small else
              "hipGraph replay of the whole step" + (" (the statistics ride in the kernel's own launch)" if in_launch else "")
              + ((" (probe: %.1f us per replayed step, %.1f eager)" % (launch_probe["replay_us"], launch_probe["eager_us"])) if launch_probe else "")) if use_graph else (
        "eager; exchange + VaR/CVaR of step n on a side stream beside the hot kernel of step n+1 (dist.PipelinedSteps)"
        if pipelined else "eager, one stream, no overlap between steps") + (
        (" (probed over 50 untimed steps, slowest rank: %.1f us serial, %.1f us pipelined)" %
         (form_probe["serial_us"], form_probe["pipelined_us"])) if form_probe else "")
    kern_src = ("HIP events around each launch of an eager pre-pass queued behind a spin kernel (back-to-back on the "
                "GPU, host issue rate excluded), mean over K launches" if use_graph
                else "HIP events around the launch, mean over the timed steps")
    kern_ranks = None
    if dist.is_initialized():             # the dominant kernel on every rank (a straggler GPU shows here, not in the max-over-ranks clock)
        every = [None] * world
        dist.all_gather_object(every, float(kern_ms))
        kern_ranks = {"min": min(every), "max": max(every), "per_rank": every}
    return {"elapsed": elapsed, "kern_ms": kern_ms, "kern_src": kern_src, "stats": final_stats, "launch": launch,
            "sclk_mhz": sclk, "comm_selfcheck": selfcheck, "kernel_ms_ranks": kern_ranks, "form_probe": form_probe}


def kkt_block(model, us_final, iters, first_cvar, where):
    """Optimality of what an SCP block timed, against the reference-layout QP (1.5e7 rows at M = 1e5) and without
    forming it: the matrix-free KKT certificate (riskaversetrajopt_amd/certificate.py) of (a) the subproblem at the final
    iterate and (b) the subproblem where the CVaR rows switch on (the hardest one: an O(1) step from the initial guess)
    -- after and outside the timed loop."""
    kkt = {}
    try:
        keys = ("primal", "stationarity", "dual_sign", "complementarity", "multiplier_scale", "active_cuts")
        _, _, info = model.solve_reduced(us_final, iters)
        c = model.certify_reduced(info)
        kkt["final_subproblem"] = {k: c[k] for k in keys}
        model._cut_solver = None
        us = model.initial_guess_us_mat()
        for k in range(first_cvar + 1):
            us, _, info = model.solve_reduced(us, k)
        c = model.certify_reduced(info)
        kkt["switch_on_subproblem"] = {k: c[k] for k in keys}
        kkt["residuals"] = (f"KKT conditions of the reference's full QP ({where}) at the lifted reduced solution; rows "
                            "scaled to unit largest coefficient, dual residuals relative to the multiplier scale; every "
                            "sum over the samples formed on the device (rato_kkt_sums)")
        for k in ("primal", "stationarity", "complementarity"):
            kkt[k] = max(kkt["final_subproblem"][k], kkt["switch_on_subproblem"][k])
    except Exception as e:                           # a diagnostic must never take the bench line down
        kkt["error"] = f"{type(e).__name__}: {e}"
    return kkt


def scp_kernels_block(model, out, args):
    """The device work of the SCP block kernel by kernel: every kernel of an oracle round trip and of a subproblem's
    definition launched 200 x back to back on the bench batch (HIP events; a kernel costs the same between copies of itself
    as between the others: profiles/EXPERIMENTS.md), with the number of launches the timed SCP made of it and what bounds
    it.  Reproducible from `rocprofv3 --kernel-trace --stats -- python scp_bench.py` (profiles/r05_*_scp_kernel_stats.csv)."""
    import torch
    from riskaversetrajopt_amd import _lib, stats as rstats
    cs = model._cut_solver
    M, S, dev = cs.M, cs.S, model.device
    st = _lib.current_stream()
    us = np.asarray(out["us"], dtype=np.float64).reshape(-1)
    cs.set_linearization_point(us)
    cs.evaluate(None, None, 0, None, us * 1.001, slot=cs.cap - 1)          # a cut at a nearby point: a realistic tail
    slot = torch.full((1,), cs.cap - 1, dtype=torch.int32, device=dev)
    m_buf, arg_buf = cs.ring_m[cs.cap - 1], cs.ring_arg[cs.cap - 1]
    part = torch.zeros((cs.nblk, cs.nc), dtype=torch.float64, device=dev)
    ws = rstats.new_workspace(M, dev)
    rec = torch.empty(rstats.N_STATS, dtype=torch.float64, device=dev)
    sums = torch.empty(cs.nc, dtype=torch.float64, device=dev)
    cs._x_np[:] = (us * 0.001).reshape(S, cs.n_u)
    _lib.copy_async(cs.x_dev, cs.x_host, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def us_per_launch(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    K = max(1, len(cs.keep))
    slots_k = torch.as_tensor((list(cs.keep) or [cs.cap - 1])[:K], dtype=torch.int32, device=dev)
    part_k = torch.zeros((cs.nblk, K * cs.nc), dtype=torch.float64, device=dev)
    t_row = us_per_launch(lambda: cs._rollout_rowmax(m_buf, arg_buf, st))
    t_sel = us_per_launch(lambda: rstats.risk_stats_device(m_buf, cs.alpha, workspace=ws, out=rec))
    t_tail = us_per_launch(lambda: cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slot), 1, part, st))
    t_fin = us_per_launch(lambda: rstats.sum_partials(part, out=sums, stream=st))
    t_union = us_per_launch(lambda: cs._rollout_tail_rows(cs.ring_m, cs.ring_arg, cs.ring_res, _lib.ptr(slots_k), K, part_k, st))
    import ctypes as C
    nd = model._native_define                                # the buffers rato_cut_define_drone linearizes into
    dW_, mass_, Q_, _ = model._inputs(None)
    pp = model._params(M, mass_.numel(), 1)
    lib = _lib.load()
    t_gen = us_per_launch(lambda: _lib.check(lib.rato_drone_linearize_generators(
        C.byref(pp), _lib.ptr(nd["us_dev"]), _lib.ptr(dW_), _lib.ptr(mass_), _lib.ptr(Q_), _lib.ptr(nd["A22"]), None, None, None,
        _lib.ptr(nd["part"]), st), "rato_drone_linearize_generators"), 100)
    iters, first = int(args.scp_iters), 2
    trips = int(out["cuts"].sum()) + int((out["cuts"] >= 0).sum() - first)       # one confirming evaluation per CVaR subproblem
    FP64_OPS = 256 * 4 * 16 * 2.4e9                                               # vector fp64 instructions-lanes per second
    rows = {
        "drone_rowmax_rollout_kernel": {"us": t_row, "calls": trips, "bound": "fp64 instruction issue",
                                        "frac": M * S * 57 / (t_row * 1e-6) / FP64_OPS,
                                        "how": "57 fp64 instructions per sample-step (counted in the ISA of the main loop: 38 fma, 7 add, "
                                               "4 mul, 3 max, 3 compare, 2 convert) against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (one "
                                               "wave64 fp64 instruction per 4 cycles per SIMD).  That peak needs 8 waves per SIMD: "
                                               "measured (tools/fp64bench.hip, profiles/r05_fp64bench.txt) a SIMD issues one per "
                                               "10.4 / 5.9 / 5.5 / 4.7 cycles with 1 / 2 / 4 / 8 resident waves whatever their "
                                               "independent chains, and M = 1e5 is 1.5 waves per SIMD"},
        "rs_coop (exact selection over m)": {"us": t_sel, "calls": trips, "bound": "latency: four dependent global phases",
                                             "frac": None},
        "drone_tail_rows_rollout_kernel": {"us": t_tail, "calls": trips, "bound": "latency of one wave's chain per block",
                                           "frac": None},
        "cut_finish_kernel (as sum_partials_kernel<double>)": {"us": t_fin, "calls": trips, "bound": "launch", "frac": None},
        "drone_linearize_generators_kernel<false,false>": {"us": t_gen, "calls": iters, "bound": "fp64 instruction issue",
                                                           "frac": None},
        f"drone_tail_rows_rollout_union_kernel (K = {K} kept cuts)": {"us": t_union, "calls": iters - first - 1,
                                                                      "bound": "latency", "frac": None},
    }
    for r in rows.values():
        r["total_ms"] = r["us"] * r["calls"] * 1e-3
    return {"per_launch": "HIP events over 200 launches back to back on the bench batch (includes ~3.6 us of launch per kernel)",
            "oracle_round_trips": trips, "kernels": rows,
            "device_ms_of_round_trips": sum(r["total_ms"] for k, r in rows.items() if r["calls"] == trips)}


def scp_block(work, args):
    """The second half of the BASELINE metric: SCP wall-clock for the drone at the bench's M and S, with the
    reference's timing protocol (drone_times.py:509-550 / drone_risk.py:510-532: a fixed 60 iterations from the
    initial guess, per-iteration "define" and "solve" times, medians, cumulative).  Runs AFTER and OUTSIDE the timed
    throughput region, on the same resident samples.  Every subproblem is the reference's QP reduced exactly to
    (u, slack) and solved by device CVaR cuts + a host master QP: the reference's own 1.5e7-row QP at
    M = 1e5 is out of reach of any host solver."""
    import torch
    from riskaversetrajopt_amd import scp
    model = work.model
    model.solve_reduced(model.initial_guess_us_mat(), 2)             # warm-up: allocations, first launches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = scp.run_drone_reduced(model, num_scp_iters_max=args.scp_iters, verbose=False)
    wall = time.perf_counter() - t0
    st = model.monte_carlo_statistics(out["us"], alpha=args.alpha)
    try:
        kernels = scp_kernels_block(model, out, args)
    except Exception as e:                              # a diagnostic must never take the bench line down
        kernels = {"error": f"{type(e).__name__}: {e}"}
    kkt = kkt_block(model, out["us"], args.scp_iters, 2, "drone_risk.py:327-368")
    return {"system": "drone_risk", "M": work.M, "S": work.S, "alpha": args.alpha, "iters": args.scp_iters, "kkt": kkt,
            "kernels": kernels,
            "cut_tolerance": 1e-9,
            "loop": f"{out.get('loop', 'python')}; every subproblem: rato_cut_define_drone + rato_cut_solve (the cutting-plane loop native)",
            "protocol": "drone_times.py:509-550: fixed iteration count from the initial guess, per-iteration define / "
                        "solve wall-clock, medians + cumulative; run after the timed throughput region.  Native loop: an "
                        "iteration's clock runs from its first instruction to the moment its solution is on the host, the "
                        "stream is synchronised once inside the last iteration's clock",
            "subproblem": "reference QP reduced exactly to (u, slack): device CVaR cuts (Jacobian-free oracle) + host "
                          "master QP (3S+1 variables); non-finite check on every linearization",
            "define_median_s": float(np.median(out["define_s"])), "solve_median_s": float(np.median(out["solve_s"])),
            "iteration_median_s": float(np.median(out["define_s"] + out["solve_s"])),
            "cumulative_s": float(out["cumulative_s"][-1]), "wall_s": wall,
            "cuts_median": float(np.median(out["cuts"])), "cuts_max": int(out["cuts"].max()),
            "cuts_total": int(out["cuts"].sum()),
            # where the cumulative time goes: the subproblems' definitions (linearization + sums + kept cuts), the oracle
            # round trips of the cutting-plane loops (device work + launch / synchronisation latency), the host master QPs
            "split_s": {"define": float(np.sum(out["define_s"])), "oracle_round_trips": float(np.sum(out["oracle_s"])),
                        "master": float(np.sum(out["solve_s"]) - np.sum(out["oracle_s"]))},
            "L2_error_last": float(out["L2_error"][-1]),
            "in_sample": {k: st[k] for k in ("var", "cvar", "frac_satisfied")}}


def scp_driving_block(device):
    """The driving SCP with the reference's protocol (driving.py:482-529: 15 iterations from the initial guess) at
    M = 1e5, S = 40, alpha = 0.05 on a fresh device-sampled batch -- reduced subproblems with the table-free oracle
    (no Jacobian is formed at all: rato_car_rowmax_rollout / rato_car_tail_rows_rollout)."""
    import torch
    from riskaversetrajopt_amd import driving, scp
    M, S, alpha, iters = 100000, 40, 0.05, 15
    dW, x0, ws, wr = driving.sample_uncertain_parameters_device(M, S, seed=7, device=device)
    model = driving.Model.from_device(S, dW, x0, ws, wr, 'saa', alpha)
    model.solve_reduced(model.initial_guess_us_mat(), 1)             # warm-up: allocations, first launches
    model._cut_solver = None
    torch.cuda.synchronize()
    out = scp.run_driving_reduced(model, num_scp_iters_max=iters, verbose=False)
    st = model.monte_carlo_statistics(out["us"], alpha=alpha)
    return {"system": "driving", "M": M, "S": S, "alpha": alpha, "iters": iters,
            "kkt": kkt_block(model, out["us"], iters, 1, "driving.py:330-373"),
            "define_median_s": float(np.median(out["define_s"])), "solve_median_s": float(np.median(out["solve_s"])),
            "cumulative_s": float(out["cumulative_s"][-1]), "cuts_max": int(out["cuts"].max()),
            "L2_error_last": float(out["L2_error"][-1]),
            "in_sample": {k: st[k] for k in ("var", "cvar", "frac_satisfied")}}


def launch_floor_block(device, torch):
    """What a replayed step costs before any work: the wall-clock of one hipGraph replay holding ONE trivial kernel, and
    the increment per further kernel node (the configurations up to 50,000 samples are timed as one replay per step:
    `ms_per_step` of a small configuration = this floor + its kernels).  Measured live, outside every timed region."""
    from riskaversetrajopt_amd import _lib, stats
    lib = _lib.load()
    ws = stats.new_workspace(1000, device)

    def tiny():
        _lib.check(lib.rato_risk_stats_init(_lib.ptr(ws), ws.numel(), _lib.current_stream()), "rato_risk_stats_init")

    def replay_us(n_nodes, reps=400):
        for _ in range(n_nodes):
            tiny()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n_nodes):
                tiny()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    one, four = replay_us(1), replay_us(4)
    return {"graph_replay_one_tiny_kernel_us": one, "per_further_node_us": (four - one) / 3.0,
            "what": "wall-clock per hipGraph replay of one 256-thread kernel that zeroes 25 KB (rato_risk_stats_init), "
                    "400 replays back to back; the second figure from a 4-node graph"}


def mc_batch_block(work, torch, K=120):
    """The reference's Monte-Carlo report evaluates 4 alpha x 30 repeats = 120 control sequences on one validation batch
    (drone_risk.py:697-725, driving.py:675-740), one jitted call each.  rato_*_eval_batch: all K in ONE call (one rollout
    launch over tiles x K, one launch of K exact selections) -> microseconds per sequence."""
    us = work.us[None].repeat(K, 1, 1).contiguous()
    us = us * (1.0 + 0.001 * torch.arange(K, device=us.device, dtype=us.dtype))[:, None, None]      # K different sequences
    bufs = {}
    for _ in range(5):
        work.model.eval_batch_device(us, out=bufs)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        work.model.eval_batch_device(us, out=bufs)
    torch.cuda.synchronize()
    per_call = (time.perf_counter() - t0) / reps
    return {"sequences_per_call": K, "us_per_call": per_call * 1e6, "us_per_sequence": per_call * 1e6 / K,
            "value": work.M * work.S * K / per_call, "unit": "samples*steps/s", "launch": "eager, one library call per batch"}


def c5_rank_local_block(work, args, device, stats, rdist, torch, world=8, K=100):
    """What EVERY rank of the 8-GPU C5 run pays locally per step, measured on this one GPU: its shard's kernel (125,000
    samples) and, for the exchange, everything but the wire -- rato_unpack_records over `world` records and the exact
    selection over all M_total = 1e6 gathered samples -- serially behind the kernel and pipelined beside the next step's
    kernel (dist.PipelinedSteps).  `bench.py --gpus N` probes both forms and times the faster one.  The records of the other
    ranks are copies of this rank's."""
    from riskaversetrajopt_amd import _lib
    lib = _lib.load()
    M, rec = work.M, work.records[0]
    all_ = rec.buf.repeat(world).contiguous()
    Z_all = torch.empty(world * M, dtype=torch.float32, device=device)
    total = torch.empty(max(rec.n_sums, 1), dtype=torch.float64, device=device)
    ws = [stats.new_workspace(world * M, device) for _ in range(2)]
    out = torch.empty((2, stats.N_STATS), dtype=torch.float64, device=device)

    def local_exchange(slot, r=None):
        _lib.check(lib.rato_unpack_records(_lib.ptr(all_), world, rec.n_sums, M, rec.rec_bytes, _lib.ptr(total),
                                           _lib.ptr(Z_all), _lib.current_stream()), "rato_unpack_records")
        stats.risk_stats_device(Z_all, args.alpha, workspace=ws[slot], out=out[slot])

    def timed(fn, reps):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def events_us(fn, reps=50):
        torch.cuda._sleep(20_000_000)                     # the launches queue up behind a spin: back-to-back device time
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps
    unpack_us = events_us(lambda: _lib.check(lib.rato_unpack_records(
        _lib.ptr(all_), world, rec.n_sums, M, rec.rec_bytes, _lib.ptr(total), _lib.ptr(Z_all), _lib.current_stream()), "unpack"))
    select_us = events_us(lambda: stats.risk_stats_device(Z_all, args.alpha, workspace=ws[0], out=out[0]))
    kernel_us = events_us(lambda: work.hot_kernel(slot=0), 30)
    serial_us = timed(lambda: (work.hot_kernel(slot=0), local_exchange(0)), K)
    pipe = rdist.PipelinedSteps(2, device)

    def piped():
        pipe.step(lambda s: work.hot_kernel(slot=s), local_exchange)
    piped_us = timed(piped, K)
    pipe.drain()
    serial2_us = timed(lambda: (work.hot_kernel(slot=0), local_exchange(0)), K)     # (again: drift between the legs)
    serial_us = min(serial_us, serial2_us)
    return {"world": world, "M_total": world * M,
            "kernel_us": kernel_us, "unpack_records_us": unpack_us, "selection_M_total_us": select_us,
            "step_serial_us": serial_us, "step_pipelined_us": piped_us,
            "default_form": "serial" if serial_us <= piped_us else "pipelined",
            "default_rule": "bench.py --gpus N (N > 1) probes both forms over 50 untimed steps and times the faster one",
            "what": "one GPU: shard kernel + rato_unpack_records(world records) + exact selection over M_total samples; "
                    "eager steps, wall-clock per step over %d steps; kernel / unpack / selection alone by HIP events, "
                    "back to back.  The wire time of the all-gather (4 MB in total over xGMI) is NOT in these numbers" % K}


def beyond_cache_block(args, device, stats, rdist, dist, torch):
    """The headline kernel where the memory-side cache cannot help: drone_risk linearize, products, M = 300,000 (180 MB of
    noise: beyond the 128 MB up to which the launcher keeps the inputs cached by writing the 9 GB Jacobian with streaming
    stores; rato_drone_rows_streaming_stores says 0) -- every input comes from HBM on every launch.  Same protocol as the
    main line, K = 12 steps."""
    import copy
    from riskaversetrajopt_amd import _lib
    a = copy.copy(args)
    a.config, a.workload, a.M, a.S, a.steps, a.warmup = "metric", "drone", 300000, 50, 12, 3
    a.mode, a.philox, a.overlap, a.graph = "linearize", False, False, "off"
    a.packed_products, a.force_factored, a.cols_per_thread, a.samples_per_lane = True, False, 0, 0
    work = WORKLOADS["drone"](a, device, seed=7)
    res = timed_region(work, a, 1, 0, device, stats, rdist, dist, torch, probe_clock=False)
    rb = roofline_block(work, res["kern_ms"], "drone", "linearize", work.M, work.S, "products", res["kern_src"])
    out = {"workload": f"drone_risk linearize M={work.M} S={work.S}, Jacobian written as products",
           "streaming_stores": bool(_lib.load().rato_drone_rows_streaming_stores(work.M, work.S, 0)),
           "noise_bytes": work.M * work.S * 12, "kernel": rb["kernel"], "kernel_ms": res["kern_ms"],
           "achieved": rb["achieved"], "peak": rb["peak"], "unit": "GB/s", "frac": rb["frac"],
           "algorithmic_bytes_per_launch": rb["algorithmic_bytes_per_launch"],
           "ms_per_step": 1e3 * res["elapsed"] / a.steps, "value": work.M * work.S * a.steps / res["elapsed"]}
    del work
    torch.cuda.empty_cache()
    return out


def c5_prediction_block(c5, args, device, stats, rdist, dist, torch):
    """BASELINE C5 as a whole -- driving M = 1e6 -- on ONE GPU (the strong-scaling reference), against what a rank of the
    8-GPU run pays locally (``rank_local``): the predicted 8-GPU step and speed-up, wire time of the one all-gather NOT
    included (no multi-GPU node in this pool; 4 MB in total, ~0.5 MB per peer link)."""
    import copy
    a = copy.copy(args)
    a.config, a.workload, a.M, a.S, a.steps, a.warmup = "C5", "driving", 1000000, 40, 10, 2
    a.mode, a.philox, a.overlap, a.graph = "linearize", False, False, "off"
    a.packed_products, a.force_factored, a.cols_per_thread, a.samples_per_lane = True, False, 0, 0
    work = WORKLOADS["driving"](a, device, seed=7)
    res = timed_region(work, a, 1, 0, device, stats, rdist, dist, torch, probe_clock=False)
    one_gpu_us = 1e6 * res["elapsed"] / a.steps
    del work
    torch.cuda.empty_cache()
    rl = c5["rank_local"]
    return {"one_gpu_M_1e6_step_us": one_gpu_us, "one_gpu_M_1e6_kernel_us": 1e3 * res["kern_ms"],
            "step_serial_us": rl["step_serial_us"], "step_pipelined_us": rl["step_pipelined_us"],
            "speedup_serial": one_gpu_us / rl["step_serial_us"], "speedup_pipelined": one_gpu_us / rl["step_pipelined_us"],
            "speedup_default": one_gpu_us / min(rl["step_serial_us"], rl["step_pipelined_us"]),
            "default_form": rl.get("default_form"),
            "target": ">= 6x at 8 GPUs (BASELINE.json north_star)", "wire": "unknown, not included: the all-gather of 8 x 0.5 MB "
            "records; pipelined it has a whole kernel time (%.0f us) to hide in" % rl["kernel_us"]}


def configs_block(args, device, stats, rdist, dist, torch):
    """BASELINE.json's other single-GPU configurations (C2 drone M=1e4 S=50, C3 driving M=1e4 S=40, C4 hopper M=5e4
    S=60 / 40 contacts, C5's shard: driving 125,000 samples per GPU S=40) as whole steps, after and outside the timed
    region of the metric configuration: per configuration the replayed (or eager, above 50,000 samples) step, the
    dominant kernel by HIP events and its roofline fraction -- the protocol of the main line with K = a few hundred
    steps each (the reference times its configurations the same way: drone_times.py:509-550)."""
    import copy
    out = {}
    for name, (wl, M, S, K, mode) in {"C2": ("drone", 10000, 50, 300, "linearize"), "C3": ("driving", 10000, 40, 300, "linearize"),
                                      "C4": ("hopper", 50000, 60, 300, "linearize"), "C5": ("driving", 125000, 40, 150, "linearize"),
                                      # C2 / C3 in the reference's OWN form: the Monte-Carlo validation (drone_risk.py:643-725,
                                      # driving.py:618-740: rollout -> max -> fraction satisfied / VaR / AVaR at M = 1e4)
                                      "C2_eval": ("drone", 10000, 50, 500, "eval"), "C3_eval": ("driving", 10000, 40, 500, "eval")}.items():
        a = copy.copy(args)
        a.config, a.workload, a.M, a.S, a.steps, a.warmup = name, wl, M, S, K, 10
        a.mode, a.philox, a.overlap, a.graph = mode, False, False, "auto"
        a.packed_products, a.force_factored, a.cols_per_thread, a.samples_per_lane = True, False, 0, 0
        try:
            work = WORKLOADS[wl](a, device, seed=7)
            res = timed_region(work, a, 1, 0, device, stats, rdist, dist, torch, probe_clock=False)
            units = getattr(work, "C", work.S)
            rb = roofline_block(work, res["kern_ms"], wl, mode, work.M, work.S,
                                "products" if (wl == "drone" and mode == "linearize") else None, res["kern_src"])
            ms = 1e3 * res["elapsed"] / K
            out[name] = {"workload": f"{work.name} M={work.M} S={work.S}" + (f" ({work.C} contacts)" if wl == "hopper" else ""),
                         "steps": K, "ms_per_step": ms, "value": work.M * units * K / res["elapsed"],
                         "unit": "samples*steps/s" if wl != "hopper" else "samples*contacts/s",
                         "kernel": rb["kernel"], "kernel_ms": res["kern_ms"], "step_over_kernel": ms / res["kern_ms"],
                         "bound": "hbm" if wl != "hopper" else "valu (transcendental issue; HBM fraction for completeness)",
                         "achieved_GBps": rb["achieved"], "frac": rb["frac"],
                         "frac_on_bytes_needed": rb.get("frac_on_bytes_needed"),
                         "algorithmic_bytes_per_launch": rb["algorithmic_bytes_per_launch"],
                         "traffic_from_profile": rb["traffic_from_profile"], "launch": res["launch"],
                         "stats": {"VaR": res["stats"][0], "CVaR": res["stats"][1], "frac_satisfied": res["stats"][2]}}
            try:   # (see the line's `two_streams` block: consecutive independent steps on alternating streams, eager)
                a2 = copy.copy(a)
                a2.graph = "off"
                r2 = timed_region(work, a2, 1, 0, device, stats, rdist, dist, torch, probe_clock=False, two_streams=True)
                out[name]["two_streams_ms_per_step"] = 1e3 * r2["elapsed"] / K
            except Exception as e:                         # noqa: BLE001
                out[name]["two_streams_ms_per_step"] = repr(e)
            if name == "C5":
                try:
                    out[name]["rank_local"] = c5_rank_local_block(work, a, device, stats, rdist, torch)
                except Exception as e:                     # noqa: BLE001
                    out[name]["rank_local"] = {"error": repr(e)}
            if mode == "eval":
                out[name]["form"] = ("the reference's Monte-Carlo validation step (rollout -> max -> fraction / VaR / AVaR), one "
                                     "control sequence per step: rollout kernel + exact selection, two launches (see `launch`)")
                out[name]["batched"] = mc_batch_block(work, torch)
            del work
            torch.cuda.empty_cache()
        except Exception as e:                             # a side block must never take the bench line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def ordered_line(line):
    """The ONE JSON line, arranged for readers that keep only `config`, `roofline` and the tail of the line: the whole
    metric (the throughput AND the SCP wall-clock, the two-stream figure, the 8-GPU prediction) is summarised inside
    `config`; the long diagnostic blocks (`configs`, `scp` with its per-kernel table, ...) come FIRST and the contract's
    keys LAST.  The blocks are also left in gpurun_out/bench_blocks.json when that directory can be written."""
    cfg = line["config"]
    scp = line.get("scp") or {}
    if "cumulative_s" in scp:
        cfg.update({"scp_cumulative_s": scp["cumulative_s"], "scp_iters": scp["iters"], "scp_cuts_total": scp["cuts_total"],
                    "scp_cuts_max": scp["cuts_max"], "scp_L2_error_last": scp["L2_error_last"],
                    "scp_define_median_s": scp["define_median_s"], "scp_solve_median_s": scp["solve_median_s"],
                    "scp_split_s": scp["split_s"], "scp_oracle_round_trips": (scp.get("kernels") or {}).get("oracle_round_trips"),
                    "scp_kkt": {k: (scp.get("kkt") or {}).get(k) for k in ("primal", "stationarity", "complementarity")},
                    "scp_what": "drone_risk SCP at this M and S, reference protocol drone_times.py:509-550 (60 iterations "
                                "from the initial guess), wall-clock seconds; = the line's scp.cumulative_s"})
    if "cumulative_s" in (line.get("scp_driving") or {}):
        cfg["scp_driving_cumulative_s"] = line["scp_driving"]["cumulative_s"]
    if "ms_per_step" in (line.get("two_streams") or {}):
        cfg["two_streams_ms_per_step"] = line["two_streams"]["ms_per_step"]
    c5 = (line.get("configs") or {}).get("C5") or {}
    p8 = c5.get("predicted_8gpu") or {}
    if "speedup_serial" in p8:
        cfg.update({"c5_predicted_speedup_serial": p8["speedup_serial"], "c5_predicted_speedup_pipelined": p8["speedup_pipelined"],
                    "c5_predicted_speedup_default": p8.get("speedup_default"), "c5_default_form": p8.get("default_form")})
    cfgs = line.get("configs") or {}
    small = {k: {"ms_per_step": v.get("ms_per_step"), "kernel_ms": v.get("kernel_ms"), "frac": v.get("frac")}
             for k, v in cfgs.items() if isinstance(v, dict) and "ms_per_step" in v}
    if small:
        cfg["configs_ms_per_step"] = small
    rb = line.get("roofline_beyond_cache") or {}
    if "frac" in rb:
        cfg["roofline_beyond_cache_frac"] = rb["frac"]
    if "cpu_baseline" in line:
        cfg["gpu_over_cpu"] = line["cpu_baseline"]["gpu_over_cpu"]
        cfg["cpu_baseline_cores"] = line["cpu_baseline"]["cores"]
    last = ("stats", "device", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "cpu_baseline", "roofline", "config")
    blocks = {k: v for k, v in line.items() if k not in last}
    try:
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d) and os.access(d, os.W_OK):
            with open(os.path.join(d, "bench_blocks.json"), "w") as f:
                json.dump(blocks, f)
    except OSError:
        pass
    out = dict(blocks)
    for k in last:
        if k in line:
            out[k] = line[k]
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                      # before any GPU call in this process
    if env_world is not None and int(env_world) != args.gpus:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE={env_world} (launch with --nproc-per-node {args.gpus}, or "
              f"let bench.py start the ranks itself: python bench.py --gpus {args.gpus})", file=sys.stderr)
        sys.exit(2)
    board = board_info() if int(os.environ.get("RANK", "0")) == 0 else None   # before this process touches the GPU
    import torch
    import torch.distributed as dist
    from riskaversetrajopt_amd import dist as rdist, stats

    if args.strict_comm:
        os.environ["RATO_STRICT_COMM"] = "1"
    rank, world, local = rdist.init_from_env()
    if args.dry_run:                                     # launch plumbing only (CPU test of the spawn path)
        devices = [None] * world
        if dist.is_initialized():
            dist.all_gather_object(devices, {"rank": rank, "local_rank": local, "device": f"cuda:{local}"})
            dist.barrier()
        else:
            devices = [{"rank": 0, "local_rank": 0, "device": "cuda:0"}]
        if rank == 0:
            print(json.dumps({"metric": "SAA constraint-eval throughput", "value": None, "n_gpus": world,
                              "dry_run": True, "scaling": "weak",
                              "config": {"baseline_config": args.config, "workload": args.workload,
                                         "M_per_gpu": args.M, "S": args.S, "M_total": world * args.M,
                                         "ranks": devices, "strict_comm": bool(args.strict_comm)}}))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.overlap is None:
        args.overlap = "auto" if world > 1 else False     # N > 1: probe serial against pipelined, time the faster

    is_drone_lin = args.workload == "drone" and args.mode == "linearize"
    variants = [None]
    if is_drone_lin:
        # "regenerated": the products output with the noise regenerated while a tile is staged (rato_drone_linearize_philox)
        variants = {"products": ["products"], "factored": ["factored"], "regenerated": ["regenerated"],
                    "both": ["products", "factored", "regenerated"]}[args.jacobian]
    results = []
    base_philox = bool(args.philox)
    for var in variants:
        args.packed_products = var in ("products", "regenerated")
        args.force_factored = (var == "factored")
        args.philox = base_philox or var == "regenerated"
        work = WORKLOADS[args.workload](args, device, seed=1000 * rank + 7)
        res = timed_region(work, args, world, rank, device, stats, rdist, dist, torch)
        res["work"], res["variant"] = work, var
        if world == 1 and var == variants[0] and args.config == "metric" and is_drone_lin and not args.no_configs:
            try:   # the same K steps with consecutive steps on two alternating streams (an extra block, outside `value`)
                r2 = timed_region(work, args, world, rank, device, stats, rdist, dist, torch, probe_clock=False, two_streams=True)
                res["two_streams"] = {
                    "ms_per_step": 1e3 * r2["elapsed"] / args.steps, "value": work.M * work.S * args.steps / r2["elapsed"],
                    "unit": "samples*steps/s",
                    "algorithmic_GBps_over_the_whole_step": work.algorithmic_bytes() * args.steps / r2["elapsed"] / 1e9,
                    "what": "the metric configuration's K steps with consecutive steps issued on two alternating streams: the "
                            "steps of this benchmark are independent of one another, so the drain of one linearize launch and "
                            "its statistics overlap the ramp of the next (per-stream tile queues).  NOT the line's `value` "
                            "(one stream, no overlap between steps): an SCP's consecutive linearizations depend on each "
                            "other through the host's QP; a Monte-Carlo study over independent batches or control sequences "
                            "can run this way.  Per-launch event times are meaningless here (two kernels share the chip)"}
            except Exception as e:                           # noqa: BLE001
                res["two_streams"] = {"error": repr(e)}
        results.append(res)
        if var != variants[-1]:
            del work.outs, work.records                  # free the 3 GB output slots before the next variant
            torch.cuda.empty_cache()
    head, work = results[0], results[0]["work"]
    M, S = work.M, work.S
    unit_steps = getattr(work, "C", S)        # hopper: the unit is a sample-contact

    if rank == 0:
        value = world * M * unit_steps * args.steps / head["elapsed"]
        jacobian = head["variant"]
        line = {
            "metric": "SAA constraint-eval throughput",
            "value": value,
            "unit": "samples*steps/s" if args.workload != "hopper" else "samples*contacts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * head["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{work.name} {args.mode}: rollout+Jacobian+mean+VaR/CVaR, "
                                   f"M={M} samples/GPU x S={S} steps, alpha={args.alpha}"
                                   + (f", Jacobian written as {jacobian}" if jacobian else ""),
                       "baseline_config": args.config,
                       "inputs": "resident in HBM before the timed region; the row-parallel drone / driving kernels read the "
                                 "batch's noise from a copy re-tiled once per batch (rato_*_tile_noise: one contiguous block per "
                                 "tile of 64 samples, the same numbers)" + (
                                     f"; that copy costs {work.retile_us:.0f} us ONCE per batch (allocation + kernel), outside "
                                     "the timed region -- an SCP that linearizes a batch 60 times pays it once"
                                     if getattr(work, "retile_us", None) else ""),
                       "M_per_gpu": M, "S": S, "M_total": world * M,
                       "value_is_for": ("the SURVEY 8(d) contract: every structural nonzero of the Jacobian written "
                                        "(3S(S-1) numbers per sample)" if jacobian in ("products", "regenerated") else
                                        ("the factored Jacobian (Phi, W): S(S-1)+6S numbers per sample"
                                         if jacobian == "factored" else "the whole step")),
                       "parallelism": f"sample-sharded x{world}, one all-gather of [sums|Z] per step",
                       "transport": rdist.transport(),
                       "comm_selfcheck": head["comm_selfcheck"],
                       "rccl_ranks": (head["comm_selfcheck"] or {}).get("rccl_ranks", 0),
                       "kernel_ms_ranks": head["kernel_ms_ranks"],
                       "step_form_probe": head.get("form_probe"),
                       "launch": head["launch"]},
            "roofline": roofline_block(work, head["kern_ms"], args.workload, args.mode, M, S, jacobian,
                                       head["kern_src"]),
            "stats": {"VaR": head["stats"][0], "CVaR": head["stats"][1], "frac_satisfied": head["stats"][2]},
            "device": {"board": board, "sclk_mhz_beside_hot_kernel": head["sclk_mhz"],
                       "how": "shader-cycle counter against the 100 MHz counter, one wave on a second stream while the hot "
                              "kernel runs (rato_device_clock_probe); diagnostic, outside the timed region"},
        }
        if head.get("two_streams"):
            line["two_streams"] = head["two_streams"]
        for res in results[1:]:                           # the other output representation, same run, same samples
            w = res["work"]
            line["roofline_" + res["variant"]] = roofline_block(w, res["kern_ms"], args.workload, args.mode, M, S,
                                                               res["variant"], res["kern_src"])
            line["value_" + res["variant"]] = world * M * unit_steps * args.steps / res["elapsed"]
            line["ms_per_step_" + res["variant"]] = 1e3 * res["elapsed"] / args.steps
        if world == 1 and is_drone_lin and not args.no_scp:
            # on the resident samples of a variant that keeps its noise materialised (the reduced SCP reads it)
            scp_work = next((r["work"] for r in reversed(results) if not r["work"].philox), None)
            if scp_work is not None:
                line["scp"] = scp_block(scp_work, args)
        if world == 1 and args.config == "metric" and is_drone_lin and not args.no_configs:
            for r in results:                             # free the metric configuration's 3 GB output slots first
                w = r["work"]
                for attr in ("outs", "records"):
                    if hasattr(w, attr):
                        delattr(w, attr)
            torch.cuda.empty_cache()
            line["configs"] = configs_block(args, device, stats, rdist, dist, torch)
            try:
                line["configs"]["C5"]["predicted_8gpu"] = c5_prediction_block(line["configs"]["C5"], args, device, stats, rdist,
                                                                               dist, torch)
            except Exception as e:                        # noqa: BLE001
                line["configs"]["C5"]["predicted_8gpu"] = {"error": repr(e)}
            try:                                          # the cache-independent figure next to the headline
                line["roofline_beyond_cache"] = beyond_cache_block(args, device, stats, rdist, dist, torch)
                p8 = line["configs"]["C5"].get("predicted_8gpu", {})
                if "one_gpu_M_1e6_kernel_us" in p8:       # driving M = 1e6 (320 MB of noise): measured for the C5 prediction
                    alg = 1000000 * 4 * (2 * 40 + 6 + 1) + 2 * 40 * 4 + 1000000 * 4 * (40 + 40 * 39)
                    gbps = alg / (p8["one_gpu_M_1e6_kernel_us"] * 1e-6) / 1e9
                    line["roofline_beyond_cache"]["driving_M_1e6"] = {
                        "kernel": "car_linearize_rows_kernel", "kernel_ms": p8["one_gpu_M_1e6_kernel_us"] * 1e-3,
                        "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": alg}
            except Exception as e:                        # noqa: BLE001
                line["roofline_beyond_cache"] = {"error": repr(e)}
            try:
                line["configs"]["launch_floor"] = launch_floor_block(device, torch)
            except Exception as e:                        # noqa: BLE001
                line["configs"]["launch_floor"] = {"error": repr(e)}
            if not args.no_scp:
                try:                                      # an extra block: its failure must not cost the metric line
                    line["scp_driving"] = scp_driving_block(device)
                except Exception as e:                    # noqa: BLE001
                    line["scp_driving"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            n = args.cpu_samples or CPU_SAMPLES[args.workload] or M
            cpu_step = work.cpu_baseline(n, args.alpha)
            threaded = getattr(cpu_step, "threaded", False)
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            cores = 1

            def timed(fn, budget_s, max_reps):
                fn()                                       # warm-up
                reps, t_cpu = 0, 0.0
                while t_cpu < budget_s and reps < max_reps:
                    t1 = time.perf_counter()
                    fn()
                    t_cpu += time.perf_counter() - t1
                    reps += 1
                return n * unit_steps * reps / t_cpu, reps, t_cpu

            if threaded:
                # 1 thread, 64, half the cpus and EVERY cpu the process may run on (P = os.cpu_count()); the
                # workers of the multi-thread legs are placed like OMP_PROC_BIND=spread (oracle/saa_oracle.c:
                # rato_oracle_place_threads -- by hand: the variable is read when libgomp loads and would also pin this
                # thread, which feeds the GPU).  `value` = the best of them, `cores` = the threads that gave it.
                from oracle import c_oracle
                v1, r1, t1 = timed(lambda: cpu_step(1), 5.0, 40)
                legs = {}
                for nt in sorted({min(avail, 64), max(avail // 2, 1), avail} - {1}):
                    c_oracle.place_threads(nt, True)
                    legs[nt] = timed(lambda: cpu_step(nt), 5.0, 200)
                    c_oracle.place_threads(nt, False)
                if not legs:
                    legs[1] = (v1, r1, t1)
                cores = max(legs, key=lambda k: legs[k][0])
                vp, rp, tp = legs[cores]
                cpu_val, extra = vp, (f"C oracle (oracle/saa_oracle.c, fp64, every sample's dense linearization formed "
                                      f"in the reference's shapes and reduced on the fly, OpenMP over samples, workers "
                                      f"spread over the {avail} cpus of the affinity mask), M={n}: "
                                      + "; ".join(f"{k} threads {v[0]:.3e} ({v[1]} reps, {v[2]:.1f} s)" for k, v in legs.items())
                                      + f"; 1 thread {v1:.3e} ({r1} reps, {t1:.1f} s); value = the best leg ({cores} threads)")
                numpy_1proc = None
                if getattr(cpu_step, "numpy", None) is not None:      # the NumPy restatement, one process, in chunks
                    cpu_step.numpy(0)
                    done, t_np, k = 0, 0.0, 1
                    while t_np < 5.0 and done < n:
                        t1_ = time.perf_counter()
                        done += cpu_step.numpy(k)
                        t_np += time.perf_counter() - t1_
                        k += 1
                    numpy_1proc = done * unit_steps / t_np
                    extra += (f"; NumPy fp64 oracle, 1 process, 2,000-sample chunks of the same batch: "
                              f"{numpy_1proc:.3e} ({done} samples, {t_np:.1f} s)")
            else:
                cpu_val, reps, t_cpu = timed(cpu_step, 10.0, 5)
                extra = f"NumPy fp64 oracle, M={n}, {reps} rep(s), {t_cpu:.1f} s"
            line["cpu_baseline"] = {
                "value": cpu_val, "unit": line["unit"], "cores": cores, "kind": "port",
                "cpus_available": avail,
                "value_by_threads": ({str(k): v[0] for k, v in legs.items()} if threaded else None),
                "value_1_thread": (v1 if threaded else cpu_val),
                "value_numpy_1_process": (numpy_1proc if threaded else cpu_val),
                "sample": extra + f"; restatement of the reference's path (its JAX/XLA-CPU path is not installable "
                                  f"here); host reports {os.cpu_count()} cpus",
                "gpu_over_cpu": value / cpu_val}
        print(json.dumps(ordered_line(line)))
    rdist.destroy_comms()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

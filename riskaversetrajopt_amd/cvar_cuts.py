"""SCP subproblem at large M: cutting planes on the linearized CVaR constraint.

The reference's QP (drone_risk.py:282-423) carries one auxiliary y_i and n_obs*S rows per sample:
1.5e7 rows and 7.4e8 nonzeros at M = 1e5, far beyond any host solver (its own included).  The y_i only
appear in  (M alpha) t + sum_i y_i + slack <= 0,  y_i >= -slack,  y_i >= (G_i u - g_up_i)_r - t,  so they can
be eliminated exactly (y_i = max(-slack, m_i(u) - t),  m_i(u) = max_r [(G_i u)_r - g_up_{i,r}]), and
minimising over t (the Rockafellar-Uryasev identity) leaves a problem in (u, slack) only:

    min  1/2 u' P u + c_s (slack^2/2 + slack)
    s.t. F u = f,   |u| <= u_max,   slack >= 0,
         CVaR_alpha(m(u)) - ((M (1 - alpha) - 1) / (alpha M)) slack <= 0            (*)

with t_risk = VaR_alpha(m(u)) + slack.  (*) is convex and piecewise linear; it is handled by Kelley cutting
planes (the CVaR decomposition of Künzi-Bay & Mayer, 2006): the HOST solves a master QP with 3S+1 variables
and one row per cut, solved exactly (dense_qp: least-distance programming through NNLS); the DEVICE, where the
linearization lives, evaluates m(u) (rato_saa_rowmax / rato_drone_rowmax_implicit), selects the tail exactly
(rato_risk_stats) and forms the cut -- the tail-weighted sums of the arg-max rows and of their offsets
(rato_saa_tail_rows_batch / rato_drone_tail_rows_implicit).  The optimum is the optimum of the reference's
QP (same feasible set and objective after projecting out y, t), so SCP iterates are comparable; checked
against the full QP in tests/test_gpu_scp.py to the north star's 1e-5.

Numerics.  The device arrays are fp32 (the kernels' outputs); everything the master sees is fp64: the rows are
evaluated, maximised and summed in double precision on the device, and a cut's value at the candidate is taken FROM
the cut (value = gradient . x + offset, both from the same fp64 sums), so that value and gradient are consistent to
1e-13 -- with fp32 rows (|G u| ~ 1e2) every cut carried ~1e-5 of noise and the iterates matched the full QP only to
2e-4 (drone) / 2e-3 (driving).  The rows are taken in DELTA form, rows(u) = g + G (u - u_k) (``u_lin``; the linearize
kernels write g instead of g_up = -g + G u_k: params.rows_out = 1): algebraically the reference's
(G u - g_up)_r (drone_risk.py:278, :357-364), without the fp32 rounding of g_up (|g_up| ~ 1e2 against |g| ~ 1e-2 on
the rows that decide the tail).

``mode='baseline'`` (drone_risk.py:303-325, driving.py:320-329): the M R_s rows kappa (G_i u - g_up_i)_r <= -pad are
the single constraint max_i m_i(u) <= -pad / kappa = CVaR_{1/M}(m(u)) <= rhs0: the same machinery with one tail
sample, no slack coupling (y, slack, t_risk are unconstrained there: slack sits at the minimiser -1 of its penalty).
``with_cvar=False`` is the reference's relaxation of the first SCP iterations (drone_risk.py:413-417: rows scaled by
1e-7 inside [-0.1, 0.1], i.e. |row| <= 1e6, never active; driving.py:411-415): u is the minimum-effort control that
meets the final constraints, slack = -1 (its row ``-slack <= 0`` is relaxed with the others), and t_risk, which the
relaxed QP leaves undetermined, is reported as 0.

Cut recycling.  A cut is a tail weighting w (sum = alpha M) and one row r_i per sample; under ANY linearization
CVaR(m(u)) >= (1/(alpha M)) sum_i w_i [(G_i u)_{r_i} - g_up_{i,r_i}].  The (w, r) of the cuts that were active at
the previous SCP iteration's solution are therefore kept on the device (rings of m values / arg-max rows / risk
statistics) and re-evaluated against the new linearization in ONE launch (rato_saa_tail_rows_batch) before the
loop starts: near convergence consecutive linearizations are close and the master starts almost solved.

Sharded over GPUs (``group``: a torch.distributed group, one rank per GPU, equal shards): every rank keeps the
Jacobian of its samples; a cut costs one all-gather of the m shards (the exact VaR needs all of them; every rank
then runs the same selection), and one rank-ordered sum of the 2(S-1) subgradient partial sums.  Every rank
solves the same tiny master; rank 0's solution is broadcast so that the ranks cannot drift apart.
"""
import math
import os
import time

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib, dense_qp, stats
from . import dist as rdist


STALL_VIOLATION, STALL_STEP = 1e-7, 1e-10      # (the stall rule of the cutting-plane loop, see _solve; also in csrc/cutloop.hip)

_CTL = []


def _threadpool_controller():
    """threadpoolctl scans the loaded libraries when a controller is built (~1 ms): build it once."""
    if not _CTL:
        try:
            from threadpoolctl import ThreadpoolController
            _CTL.append(ThreadpoolController())
        except ImportError:                       # pragma: no cover
            _CTL.append(None)
    return _CTL[0]


class CvarCutSolver:
    def __init__(self, lib, device, *, n_u, S, M, ld, R, alpha, dt, Rcost, slack_penalty, u_min, u_max,
                 recycle=True, group=None, world=1, mode='saa', rhs0=0.0):
        self.lib, self.device = lib, device
        self.n_u, self.S, self.M, self.ld, self.R = n_u, S, M, ld, R
        self.group, self.world = group, int(world)      # M = samples of THIS rank; M * world in total
        self.M_total = M * self.world
        self.mode = mode
        self.nU = n_u * S
        self.u_min, self.u_max = float(u_min), float(u_max)
        # drone: (rato_drone_params, mass, A22, a22_axes) -> Jacobian-free evaluation of m(u); with G = None in
        # evaluate / relinearize_kept_cuts the tail rows are regenerated from A22 as well (generators-only mode)
        self.implicit = None
        # table-free: ('drone', rato_drone_params, dW, mass, Qsym) / ('driving', rato_car_params, dW, x0_ped, w_speed,
        # w_rep) -> the oracle re-runs the rollout at u_lin in fp64 from the samples (rato_*_rowmax_rollout /
        # rato_*_tail_rows_rollout); needs the delta form (u_lin)
        self.rollout = None
        self.check_finite = False                        # raise RatoNonFiniteError when an oracle call's m values hold NaN/Inf
        n = self.nU + 1
        Pu = sp.kron(sp.eye(S), sp.csc_matrix(2.0 * dt * np.asarray(Rcost, dtype=np.float64)))
        self.P = sp.block_diag([Pu, sp.csc_matrix([[float(slack_penalty)]])], format="csc")
        self.q = np.zeros(n)
        self.q[-1] = float(slack_penalty)
        self._Pd, self._I = self.P.toarray(), np.eye(n)
        off = self._Pd - np.diag(np.diagonal(self._Pd))
        self._p_diag = np.ascontiguousarray(np.diagonal(self._Pd)) if not off.any() else None   # diagonal cost matrix R
        if mode == 'saa':
            self.alpha = float(alpha)
            self.alphaM = self.alpha * self.M_total           # tail mass of a cut
            self.c_s = (self.M_total * (1.0 - alpha) - 1.0) / (alpha * self.M_total)
            self.rhs0 = 0.0
        elif mode == 'baseline':
            # max_i m_i(u) = CVaR with a tail of exactly one sample; (1 + 1e-9) keeps floor(alpha M) = 1 in fp64
            self.alpha = (1.0 + 1e-9) / self.M_total
            self.alphaM = 1.0
            self.c_s = 0.0
            self.rhs0 = float(rhs0)
        else:
            raise ValueError(f"mode must be 'saa' or 'baseline', got {mode!r}")
        self.recycle = recycle
        self.keep_max = 48
        self.keep_recent = int(os.environ.get("RATO_KEEP_RECENT", 4))   # newest cuts kept beside the ones with a multiplier
        self.cap = (160 if recycle else 1) + 1          # last slot: scratch for calls beyond the ring
        self.nc = 2 * max(S - 1, 0) + 1
        self.nres = stats.N_STATS + self.nc
        self.keep = []                                   # slots kept from the previous solve
        self.idle = {}                                   # slot -> consecutive solves it has carried no multiplier
        self.keep_idle = int(os.environ.get("RATO_KEEP_IDLE", 0))        # (A/B knobs: profiles/EXPERIMENTS.md, "2-cycles")
        self.u_lin = None                                # linearization point of the delta form (None: reference form)
        self._relin_pending = None                       # kept cuts whose re-linearization is already in flight
        self._native = None                              # (key, rato_cut_solver handle, its output arrays)
        self.use_native_loop = True
        if self.world > 1:
            # every collective of the loop moves buffers whose lengths follow from (M, S, n_u, keep_max): agreed on here,
            # once, by every rank (the solver is built collectively, in the first solve_reduced after Model.shard()) --
            # which is what lets the per-cut all-gathers skip their own length check (dist.gather_concat, agreed=True)
            for name, val in (("samples per rank M", M), ("horizon S", S), ("controls n_u", n_u), ("keep_max", self.keep_max)):
                rdist.check_equal_shards(val, group, what=f"cutting-plane solver: {name}")
        if device is not None:                           # (None: a host oracle overrides evaluate / relinearize_kept_cuts
            self._alloc_device(device)                   #  -- tests/_host_cuts.py, the fp64 checker of this loop)

    def _alloc_device(self, device):
        """Device scratch.  Every oracle call writes into one slot of three rings (m values, arg-max rows,
        [statistics | cut sums (2(S-1) gradient entries, then the offset sum)]); slots of cuts worth recycling
        survive the solve."""
        S, M, n_u, recycle = self.S, self.M, self.n_u, self.recycle
        e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=device)
        self.ring_m = e(self.cap, M)
        self.ring_arg = e(self.cap, M, dt=torch.int32)
        self.ring_res = torch.zeros((self.cap, self.nres), dtype=torch.float64, device=device)
        self.nblk = (M + 255) // 256
        self.part = e(self.nblk, self.nc, dt=torch.float64) if S > 1 else None
        self.part_b = e(self.nblk, self.keep_max * self.nc, dt=torch.float64) if (recycle and S > 1) else None
        self.sums_b = torch.zeros(self.keep_max * self.nc, dtype=torch.float64, device=device)
        self.sums_b_host = torch.zeros(self.keep_max * self.nc, dtype=torch.float64).pin_memory()
        self.slots_dev = torch.zeros(self.keep_max, dtype=torch.int32, device=device)
        self.slots_host = torch.zeros(self.keep_max, dtype=torch.int32).pin_memory()
        self.ws = stats.new_workspace(self.M_total, device)
        self.res_host = torch.zeros(self.nres, dtype=torch.float64).pin_memory()
        self.agree_host = torch.zeros(1, dtype=torch.float64).pin_memory()   # (sharded: see _selection_gave_up)
        self.x_host = torch.zeros((S, n_u), dtype=torch.float64).pin_memory()
        self.x_dev = e(S, n_u, dt=torch.float64)
        self.uk_dev = e(S, n_u, dt=torch.float64)        # the linearization point, for the rollout form of the oracle
        # (pointers of the fixed buffers of the one-call round trip, looked up once)
        self._x_np = self.x_host.numpy()
        self._slots_np = self.slots_host.numpy()
        self.uk_host = torch.zeros((S, n_u), dtype=torch.float64).pin_memory()
        self._uk_np = self.uk_host.numpy()
        self._uk_ptr, self._x_host_ptr, self._x_dev_ptr = _lib.ptr(self.uk_dev), _lib.ptr(self.x_host), _lib.ptr(self.x_dev)
        self._ws_ptr, self._res_host_ptr = _lib.ptr(self.ws), _lib.ptr(self.res_host)
        self._part_ptr = _lib.ptr(self.part)

    # ---- the two forms of the rows ------------------------------------------
    def _form(self):
        """-> (sign, x0): rows(u) = G (u - x0) + sign * base  (delta form: base = g, x0 = u_k; reference form:
        base = g_up, x0 = 0)."""
        if self.u_lin is None:
            return -1.0, np.zeros(self.nU)
        return 1.0, self.u_lin

    # ---- device oracle -----------------------------------------------------
    def evaluate(self, G, W, tile, base, u_vec, slot=None):
        """-> (phi = CVaR_alpha(m(u)), t* = VaR, g (nU,) subgradient of phi).  One upload of x = u - x0, four stream-
        ordered calls, ONE read-back (statistics + cut sums).  ``slot``: ring slot that receives the call's m
        values / arg-max rows / statistics (default: the scratch slot)."""
        S, M, n_u = self.S, self.M, self.n_u
        slot = self.cap - 1 if slot is None else slot
        m_buf, arg_buf, res = self.ring_m[slot], self.ring_arg[slot], self.ring_res[slot]
        sign, x0 = self._form()
        x = np.ascontiguousarray(np.asarray(u_vec, dtype=np.float64) - x0)
        if self.rollout is not None and self.world == 1:
            # table-free forms: the whole round trip (upload, rowmax, selection, cut sums, read-back, synchronize) is one
            # library call
            self._x_np[:] = x.reshape(S, n_u)
            kind, p, *samples = self.rollout
            samples = [_lib.ptr(a) for a in samples] + [None] * (4 - len(samples))
            _lib.check(self.lib.rato_cut_oracle_rollout(
                0 if kind == "drone" else 1, _lib.C.byref(p), self._uk_ptr, *samples, self._x_host_ptr, self._x_dev_ptr,
                _lib.ptr(m_buf), _lib.ptr(arg_buf), float(self.alpha), float(stats.SATISFIED_THRESHOLD),
                float(self.alphaM), self._ws_ptr, self.ws.numel(), _lib.ptr(res), self._part_ptr, self._res_host_ptr,
                _lib.current_stream()), "rato_cut_oracle_rollout")
        else:
            self._evaluate_stepwise(G, W, tile, base, x, sign, m_buf, arg_buf, res)
        r = self.res_host.numpy()
        if self._selection_gave_up(r, m_buf):
            # finite m values, NaN statistics: the one-launch selection gave up (a chip some other stream owned for
            # seconds, an unclean workspace) -- the same round trip with the launch-per-pass selection
            self._evaluate_stepwise(G, W, tile, base, x, sign, m_buf, arg_buf, res, recover=True)
            r = self.res_host.numpy()
        if self.check_finite and not (np.isfinite(r[3]) and np.isfinite(r[4])):      # mean and max of the m values
            raise _lib.RatoNonFiniteError("CVaR-cut oracle: non-finite constraint values m_i(u) (RATO_ENONFINITE)")
        g = np.zeros(self.nU)
        if S > 1:
            sums = r[stats.N_STATS:] / self.alphaM
            g.reshape(S, n_u)[:S - 1, 0:2] = sums[:self.nc - 1].reshape(S - 1, 2)
            # the cut's own value at the candidate (fp64, consistent with g).  Every inner product that feeds the master
            # is an exactly rounded sum (math.fsum): independent of summation order, so the native loop
            # (csrc/cutloop.hip, rato_cut_solve) reproduces this one bit for bit
            phi = math.fsum(g * x) + sign * sums[self.nc - 1]
        else:
            phi = float(r[1])                               # no control enters row t = 0: the value is a constant
        return phi, float(r[0]), g

    def _selection_gave_up(self, r, m_buf):
        """Did this round trip's one-launch selection give up on finite m values (-> redo it with the launch-per-pass
        form)?  Sharded, the redo contains collectives, so the answer must be THE SAME ON EVERY RANK: it is taken from
        what every rank holds identically -- the rank-ordered sum of the ranks' thresholds t*, which travels with the
        cut sums (a selection that gave up leaves NaN in every slot of its record, and one NaN poisons the sum on every
        rank), and the gathered m values -- never from this rank's own statistics or its own shard of m."""
        if self.world == 1:
            return bool(np.isnan(r[0])) and stats.count_nonfinite(m_buf) == 0
        if self.S > 1:
            nan_here = bool(np.isnan(self.agree_host.numpy()[0]))
        else:                                       # no cut sums travel when S = 1: agree explicitly
            nan_here = rdist.any_rank(bool(np.isnan(r[0])), self.device, self.group)
        return nan_here and stats.count_nonfinite(self._m_all) == 0

    def _evaluate_stepwise(self, G, W, tile, base, x, sign, m_buf, arg_buf, res, recover=False):
        """the same round trip as separate stream-ordered calls (table forms of the oracle, sharded batches);
        ``recover``: the selection by the launch-per-pass form on a re-initialised workspace"""
        S, M, n_u = self.S, self.M, self.n_u
        tstream = torch.cuda.current_stream()
        st = _lib.C.c_void_p(tstream.cuda_stream)        # one stream lookup per call
        self.x_host.copy_(torch.from_numpy(x.reshape(S, n_u)))
        self.x_dev.copy_(self.x_host, non_blocking=True)
        if self.rollout is not None:
            self._rollout_rowmax(m_buf, arg_buf, st)
        elif self.implicit is not None:
            p, mass, A22, axes = self.implicit
            _lib.check(self.lib.rato_drone_rowmax_implicit(_lib.C.byref(p), _lib.ptr(mass), _lib.ptr(A22), axes,
                                                           _lib.ptr(W), _lib.ptr(base), sign, _lib.ptr(self.x_dev),
                                                           _lib.ptr(m_buf), _lib.ptr(arg_buf), st),
                       "rato_drone_rowmax_implicit")
        else:
            _lib.check(self.lib.rato_saa_rowmax(_lib.ptr(G), _lib.ptr(W), tile, self.R, S, M, self.ld,
                                                _lib.ptr(base), sign, _lib.ptr(self.x_dev), n_u, _lib.ptr(m_buf),
                                                _lib.ptr(arg_buf), st), "rato_saa_rowmax")
        m_all = m_buf if self.world == 1 else rdist.gather_concat(m_buf, self.group, agreed=True)
        self._m_all = m_all
        (stats.risk_stats_recover_device if recover else stats.risk_stats_device)(
            m_all, self.alpha, workspace=self.ws, out=res[:stats.N_STATS], stream=st)
        if S > 1:
            if self.rollout is not None:
                self._rollout_tail_rows(m_buf, arg_buf, res, None, 1, self.part, st)
            elif G is None:              # generators-only linearization: rows regenerated from A22
                p, mass, A22, axes = self.implicit
                _lib.check(self.lib.rato_drone_tail_rows_implicit(
                    _lib.C.byref(p), _lib.ptr(mass), _lib.ptr(A22), axes, _lib.ptr(W), _lib.ptr(base),
                    _lib.ptr(m_buf), _lib.ptr(arg_buf), _lib.ptr(res), self.nres, None, 1, float(self.alphaM),
                    _lib.ptr(self.part), st), "rato_drone_tail_rows_implicit")
            else:
                _lib.check(self.lib.rato_saa_tail_rows_batch(
                    _lib.ptr(G), _lib.ptr(W), self.ld, tile, self.R, S, M, _lib.ptr(base), _lib.ptr(m_buf),
                    _lib.ptr(arg_buf), _lib.ptr(res), self.nres, None, 1, float(self.alphaM), _lib.ptr(self.part),
                    st), "rato_saa_tail_rows_batch")
            stats.sum_partials(self.part, out=res[stats.N_STATS:], stream=st)
            if self.world > 1:
                # [t* | cut sums] in one collective: the sum of the ranks' t* is NaN on EVERY rank as soon as one rank's
                # selection gave up (``_selection_gave_up``)
                tot = rdist.sum_in_rank_order(res[stats.N_STATS - 1:], self.group, agreed=True)
                res[stats.N_STATS:].copy_(tot[1:])
                self.agree_host.copy_(tot[:1], non_blocking=True)
        self.res_host.copy_(res, non_blocking=True)
        tstream.synchronize()

    def _rollout_rowmax(self, m_buf, arg_buf, st):
        kind, p, *samples = self.rollout
        name = {"drone": "rato_drone_rowmax_rollout", "driving": "rato_car_rowmax_rollout"}[kind]
        _lib.check(getattr(self.lib, name)(_lib.C.byref(p), _lib.ptr(self.uk_dev), *[_lib.ptr(a) for a in samples],
                                           _lib.ptr(self.x_dev), _lib.ptr(m_buf), _lib.ptr(arg_buf), st), name)

    def _rollout_tail_rows(self, m_base, arg_base, res_base, slots, K, part, st):
        kind, p, *samples = self.rollout
        name = {"drone": "rato_drone_tail_rows_rollout", "driving": "rato_car_tail_rows_rollout"}[kind]
        _lib.check(getattr(self.lib, name)(_lib.C.byref(p), _lib.ptr(self.uk_dev), *[_lib.ptr(a) for a in samples],
                                           _lib.ptr(m_base), _lib.ptr(arg_base), _lib.ptr(res_base), self.nres, slots, K,
                                           float(self.alphaM), _lib.ptr(part), st), name)

    def set_linearization_point(self, u_lin):
        """The controls the current linearization was taken at (delta form of the rows; None: reference form).  For the
        rollout form of the oracle they also go to the device, in fp64."""
        self.u_lin = None if u_lin is None else np.asarray(u_lin, dtype=np.float64).reshape(-1).copy()
        if self.rollout is not None:
            if self.u_lin is None:
                raise ValueError("the rollout form of the oracle needs the linearization point (delta form)")
            self._uk_np[:] = self.u_lin.reshape(self.S, self.n_u)    # (pinned; every solve ends synchronised, so the
            _lib.copy_async(self.uk_dev, self.uk_host)               #  previous upload from it has long completed)

    def begin(self, u_lin, relinearize, G=None, W=None, tile=0, base=None):
        """Stream-ordered prologue of a subproblem: the linearization point (``set_linearization_point``) and, when
        ``relinearize``, the kept cuts against it (``enqueue_relinearize``) -- one library call (rato_cut_begin) where the
        native loop applies."""
        if u_lin is not None and self.native_loop_applies():
            self.u_lin = np.asarray(u_lin, dtype=np.float64).reshape(-1).copy()
            h = self._native_solver()
            K = len(self.keep) if (relinearize and self.recycle and self.S >= 2) else 0
            out = self._keep_arrays()
            _lib.check(self.lib.rato_cut_begin(h, self.u_lin.ctypes.data, out["keep"].ctypes.data, K,
                                               _lib.current_stream()), "rato_cut_begin")
            self._relin_pending = K if K else None
            return
        self.set_linearization_point(u_lin)
        if relinearize:
            self.enqueue_relinearize(G, W, tile, base)

    def enqueue_relinearize(self, G, W, tile, base):
        """Stream-ordered half of ``relinearize_kept_cuts`` (launches + the read-back into pinned memory, NO
        synchronisation): called right behind the linearize launch, so that the caller's own read-back of the sample
        sums waits for both at once instead of paying a second round trip."""
        self._relin_pending = self._relin_launch(G, W, tile, base)

    def relinearize_kept_cuts(self, G, W, tile, base):
        """The kept cuts under the current linearization -> (rows (K, nU), rhs (K,)):  rows[k].u - c_s s <= rhs[k].
        One batched launch + one partial-sum launch + one read-back (already in flight after ``enqueue_relinearize``)."""
        K, self._relin_pending = self._relin_pending, None
        if K is None:
            K = self._relin_launch(G, W, tile, base)
        if K == 0:
            return np.zeros((0, self.nU)), np.zeros(0)
        S, n_u = self.S, self.n_u
        _lib.synchronize()
        r = self.sums_b_host.numpy()[:K * self.nc].reshape(K, self.nc) / self.alphaM
        rows = np.zeros((K, self.nU))
        rows.reshape(K, S, n_u)[:, :S - 1, 0:2] = r[:, :self.nc - 1].reshape(K, S - 1, 2)
        # cut k:  rows[k].(u - x0) + sign c0_k - c_s s <= rhs0
        sign, x0 = self._form()
        dots = np.array([math.fsum(rows[k] * x0) for k in range(K)])
        return rows, (self.rhs0 + dots) - sign * r[:, self.nc - 1]

    def _relin_launch(self, G, W, tile, base):
        """-> number of kept cuts whose sums are on their way to ``sums_b_host`` (0: nothing to do)"""
        K, S, M, n_u = len(self.keep), self.S, self.M, self.n_u
        if K == 0 or S < 2 or not self.recycle:
            return 0
        st = _lib.current_stream()
        self._slots_np[:K] = self.keep
        _lib.copy_async(self.slots_dev, self.slots_host, st)
        part = self.part_b.view(-1)[:self.nblk * K * self.nc].view(self.nblk, K * self.nc)
        if self.rollout is not None:
            self._rollout_tail_rows(self.ring_m, self.ring_arg, self.ring_res, _lib.ptr(self.slots_dev), K, part, st)
        elif G is None:
            p, mass, A22, axes = self.implicit
            _lib.check(self.lib.rato_drone_tail_rows_implicit(
                _lib.C.byref(p), _lib.ptr(mass), _lib.ptr(A22), axes, _lib.ptr(W), _lib.ptr(base),
                _lib.ptr(self.ring_m), _lib.ptr(self.ring_arg), _lib.ptr(self.ring_res), self.nres,
                _lib.ptr(self.slots_dev), K, float(self.alphaM), _lib.ptr(part),
                _lib.current_stream()), "rato_drone_tail_rows_implicit")
        else:
            _lib.check(self.lib.rato_saa_tail_rows_batch(
                _lib.ptr(G), _lib.ptr(W), self.ld, tile, self.R, S, M, _lib.ptr(base), _lib.ptr(self.ring_m),
                _lib.ptr(self.ring_arg), _lib.ptr(self.ring_res), self.nres, _lib.ptr(self.slots_dev), K,
                float(self.alphaM), _lib.ptr(part), _lib.current_stream()), "rato_saa_tail_rows_batch")
        if self.world == 1:       # the reduction writes straight into pinned host memory: no read-back copy to issue
            stats.sum_partials(part, out=self.sums_b_host[:K * self.nc], stream=st)
            return K
        stats.sum_partials(part, out=self.sums_b[:K * self.nc], stream=st)
        self.sums_b[:K * self.nc].copy_(rdist.sum_in_rank_order(self.sums_b[:K * self.nc], self.group, agreed=True))
        _lib.copy_async(self.sums_b_host, self.sums_b, st)
        return K

    # ---- master QP (host, exact: dense_qp) -----------------------------------------
    def solve(self, *args, **kwargs):
        """-> dict(us (S,n_u), slack, t_risk, cuts, phi, oracle_s, master_s, status).

        The master QP is tiny (3S+1 variables): a BLAS pool with one thread per host core (256 on the MI355X
        boxes) makes every call slower AND starves the thread that feeds the GPU (measured: 2.9 ms instead of
        0.65 ms per oracle call), so the loop runs under a 4-thread limit."""
        if kwargs.get("u_lin") is not None and not kwargs.get("verbose") and self.native_loop_applies():
            return self._solve(*args, **kwargs)   # the native loop: no BLAS call is made on this side
        ctl = _threadpool_controller()
        if ctl is None:                           # pragma: no cover
            return self._solve(*args, **kwargs)
        with ctl.limit(limits=4):
            return self._solve(*args, **kwargs)

    # ---- the loop as one library call (csrc/cutloop.hip) ------------------------------------------------------
    def native_loop_applies(self):
        """rato_cut_solve covers the benchmarked configuration: table-free oracle, one GPU, diagonal cost matrix, the
        device oracle itself (a subclass that overrides ``evaluate`` -- the host checker of tests/_host_cuts.py -- keeps
        the Python loop, as do sharded batches and the table forms).  RATO_PY_CUT_LOOP=1 forces the Python loop (A/B)."""
        return (self.rollout is not None and self.world == 1 and self._p_diag is not None and self.device is not None
                and type(self).evaluate is CvarCutSolver.evaluate and os.environ.get("RATO_PY_CUT_LOOP") != "1"
                and self.use_native_loop)

    def _native_solver(self):
        """the rato_cut_solver for the current samples (rebuilt when they, or the parameters, change)"""
        kind, p, *samples = self.rollout
        nat = self._native
        # (the same parameter object and sample tensors as last time: the handle stands if the struct's BYTES are also
        #  what the native solver copied -- it holds the parameters by value, so an in-place edit of the struct (dt, beta,
        #  S ... of the Model behind it) must rebuild it; ~1 us for the 200 bytes)
        knobs = (bool(self.recycle), self.keep_recent, self.keep_idle, self.keep_max)
        pbytes = bytes(p)
        if nat is not None and len(nat) > 3 and nat[3][0] is p and len(nat[3][1]) == len(samples) and \
                all(a is b for a, b in zip(nat[3][1], samples)) and nat[3][2] == knobs and nat[3][3] == pbytes:
            return nat[1]
        key = (kind, pbytes) + knobs + tuple(a.data_ptr() for a in samples)
        if nat is not None and nat[0] == key:
            self._native = (nat[0], nat[1], nat[2], (p, samples, knobs, pbytes))
            return nat[1]
        self._native_destroy()
        C = _lib.C
        cfg = _lib.CutConfig()
        cfg.system, cfg.S, cfg.cap = (0 if kind == "drone" else 1), self.S, self.cap
        cfg.keep_max, cfg.keep_recent, cfg.keep_idle = self.keep_max, self.keep_recent, self.keep_idle
        cfg.mode_saa, cfg.recycle, cfg.M = int(self.mode == 'saa'), int(bool(self.recycle)), self.M
        cfg.alpha, cfg.alphaM, cfg.c_s, cfg.rhs0 = float(self.alpha), float(self.alphaM), float(self.c_s), float(self.rhs0)
        cfg.u_min, cfg.u_max, cfg.thr = self.u_min, self.u_max, float(stats.SATISFIED_THRESHOLD)
        cfg.params = C.cast(C.pointer(p), C.c_void_p)
        for name, a in zip(("s0", "s1", "s2", "s3"), samples):
            setattr(cfg, name, a.data_ptr())
        q = np.ascontiguousarray(self.q, dtype=np.float64)
        for name, t in (("uk_dev", self.uk_dev), ("uk_host", self.uk_host), ("x_host", self.x_host), ("x_dev", self.x_dev),
                        ("ring_m", self.ring_m), ("ring_arg", self.ring_arg), ("ring_res", self.ring_res),
                        ("workspace", self.ws), ("part", self.part), ("part_b", self.part_b),
                        ("sums_b_host", self.sums_b_host), ("slots_dev", self.slots_dev), ("slots_host", self.slots_host),
                        ("res_host", self.res_host)):
            setattr(cfg, name, None if t is None else t.data_ptr())
        cfg.workspace_bytes = self.ws.numel() * self.ws.element_size()
        cfg.p_diag, cfg.q = self._p_diag.ctypes.data, q.ctypes.data
        h = C.c_void_p()
        _lib.check(self.lib.rato_cut_solver_create(C.byref(h), C.byref(cfg)), "rato_cut_solver_create")
        nU = self.nU
        out = {"us": np.zeros(nU), "cut_slot": np.zeros(self.cap + 8, dtype=np.int32), "cut_lambda": np.zeros(self.cap + 8),
               "bound_var": np.zeros(2 * nU, dtype=np.int32), "bound_sign": np.zeros(2 * nU), "bound_lambda": np.zeros(2 * nU),
               "keep": np.zeros(max(self.keep_max, 1), dtype=np.int32), "idle": np.zeros(max(self.keep_max, 1), dtype=np.int32),
               "n_keep": C.c_int32(0), "samples": samples, "q": q}     # (samples, q: kept alive with the handle)
        res = _lib.CutResult()
        res.us, res.cut_slot, res.cut_lambda = out["us"].ctypes.data, out["cut_slot"].ctypes.data, out["cut_lambda"].ctypes.data
        res.cut_capacity = out["cut_slot"].size
        res.bound_var, res.bound_sign = out["bound_var"].ctypes.data, out["bound_sign"].ctypes.data
        res.bound_lambda, res.bound_capacity = out["bound_lambda"].ctypes.data, out["bound_var"].size
        out["res"] = res
        self._native = (key, h, out, (p, samples, knobs, pbytes))
        return h

    def _native_destroy(self):
        nat, self._native = getattr(self, "_native", None), None
        if nat is not None:
            self.lib.rato_cut_solver_destroy(nat[1])

    def __del__(self):
        try:
            self._native_destroy()
        except Exception:                         # pragma: no cover  (interpreter shutdown)
            pass

    def _keep_arrays(self):
        out = self._native[2]
        K = len(self.keep)
        out["keep"][:K] = self.keep
        out["idle"][:K] = [self.idle.get(sl, 0) for sl in self.keep]
        out["n_keep"].value = K
        return out

    def _solve_native(self, final_du, final_rhs, u_lin, with_cvar, tol, max_cuts, final_cut_above):
        """one rato_cut_solve call -> the same ``info`` dict as the Python loop below"""
        h = self._native_solver()
        out = self._keep_arrays()
        fdu = np.ascontiguousarray(final_du, dtype=np.float64)
        frhs = np.ascontiguousarray(final_rhs, dtype=np.float64)
        u_lin = np.ascontiguousarray(u_lin, dtype=np.float64).reshape(-1)
        in_flight, self._relin_pending = self._relin_pending, None
        res = out["res"]
        rc = self.lib.rato_cut_solve(h, fdu.ctypes.data, frhs.ctypes.data, fdu.shape[0], u_lin.ctypes.data, int(with_cvar),
                                     float(tol), int(max_cuts), float(final_cut_above), int(self.check_finite),
                                     out["keep"].ctypes.data, out["idle"].ctypes.data, _lib.C.byref(out["n_keep"]),
                                     int(bool(in_flight)), _lib.C.byref(res), _lib.current_stream())
        if rc in (_lib.RATO_ERANK, _lib.RATO_ESELECT):
            return None                        # -> the Python loop (general master / recovering selection)
        if rc == _lib.RATO_EINFEASIBLE:
            raise dense_qp.InfeasibleError("master QP infeasible")
        if rc == _lib.RATO_ENONFINITE:
            raise _lib.RatoNonFiniteError("CVaR-cut oracle: non-finite constraint values m_i(u) (RATO_ENONFINITE)")
        _lib.check(rc, "rato_cut_solve")
        if self.recycle and with_cvar:
            K = out["n_keep"].value
            self.keep = [int(v) for v in out["keep"][:K]]
            self.idle = {int(sl): int(c) for sl, c in zip(out["keep"][:K], out["idle"][:K])}
        nb = res.n_bounds
        bounds = []
        if nb:       # grouped by sign run, as the Python loop reports them
            var, sgn, lam = out["bound_var"][:nb], out["bound_sign"][:nb], out["bound_lambda"][:nb]
            cut_at = np.flatnonzero(np.diff(sgn) != 0) + 1
            for idx, sg, la in zip(np.split(var, cut_at), np.split(sgn, cut_at), np.split(lam, cut_at)):
                bounds.append((idx.astype(np.int64).copy(), float(sg[0]), la.copy()))
        nc = res.n_cut_rows
        info = {"oracle_s": res.oracle_s, "master_s": res.master_s,
                "multipliers": {"cuts": [(int(sl), float(la)) for sl, la in zip(out["cut_slot"][:nc], out["cut_lambda"][:nc])],
                                "slack": float(res.lam_slack), "bounds": bounds,
                                "uncertified_cuts": int(res.uncertified_cuts)},
                "us": out["us"].reshape(self.S, self.n_u).copy(), "slack": float(res.slack), "t_risk": float(res.t_risk),
                "cuts": int(res.cuts), "phi": float(res.phi),
                "status": ("solved", "maximum cuts reached", "solved (stalled)")[res.status], "loop": "native"}
        if self.recycle and with_cvar:
            info["recycled"] = int(res.recycled)
        return info

    def _solve(self, G, W, tile, base, final_du, final_rhs, *, u_lin=None, with_cvar=True, tol=1e-9, max_cuts=400,
               verbose=False, final_cut_above=1e-11):
        """``base``: g_up [R][S][ld] of the linearize call (reference form), or -- with ``u_lin`` = the controls the
        linearization was taken at -- its g output (delta form, params.rows_out = 1)."""
        if u_lin is not None and not verbose and self.native_loop_applies():
            new_lin = np.asarray(u_lin, dtype=np.float64).reshape(-1)
            if self.u_lin is None or not np.array_equal(new_lin, self.u_lin):
                self.set_linearization_point(new_lin)
                self._relin_pending = None
            info = self._solve_native(final_du, final_rhs, new_lin, with_cvar, tol, max_cuts, final_cut_above)
            if info is not None:
                return info
        nU, n = self.nU, self.nU + 1
        new_lin = None if u_lin is None else np.asarray(u_lin, dtype=np.float64).reshape(-1)
        if (new_lin is None) != (self.u_lin is None) or (new_lin is not None and not np.array_equal(new_lin, self.u_lin)):
            self.set_linearization_point(new_lin)       # (solve_reduced has usually done this before enqueueing work)
        F = np.hstack([np.asarray(final_du, dtype=np.float64), np.zeros((np.shape(final_du)[0], 1))])
        f = np.asarray(final_rhs, dtype=np.float64)
        Pd, I = self._Pd, self._I                       # (dense P, identity: built once)
        info = {"oracle_s": 0.0, "master_s": 0.0}
        phi = tstar = np.nan
        status = "solved"
        n_cuts = 0
        slack_row = with_cvar and self.mode == 'saa'    # -slack <= 0 (relaxed with the CVaR rows; absent in 'baseline')
        t0 = time.perf_counter()
        master = dense_qp.Master(Pd, self.q, F, f, p_diag=self._p_diag)   # equality elimination + whitening once per SCP iteration
        n_rows = 0
        if slack_row:
            master.add_rows(-I[nU:], [0.0])             # slack >= 0
            n_rows = 1
        cut_rows = []                                   # (row of the master, ring slot) of every CVaR cut
        kept = list(self.keep) if (self.recycle and with_cvar) else []
        if not kept:
            self._relin_pending = None                  # (an enqueued batch for a relaxed iteration is simply not read)
        info["master_s"] += time.perf_counter() - t0
        if kept:
            t0 = time.perf_counter()
            rows, rhs = self.relinearize_kept_cuts(G, W, tile, base)
            info["oracle_s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            master.add_rows(np.hstack([rows, np.full((len(kept), 1), -self.c_s)]), rhs)
            cut_rows += [(n_rows + k, slot) for k, slot in enumerate(kept)]
            n_rows += len(kept)
            info["master_s"] += time.perf_counter() - t0
        free = [sl for sl in range(self.cap - 1) if sl not in set(kept)]
        in_master = np.zeros(2 * nU, dtype=bool)        # control bounds enter lazily: only the violated ones
        bound_rows = []                                 # (first master row, variable indices, +1 upper / -1 lower)
        lam = np.zeros(0)
        def solve_master():
            """the master with the control bounds entering lazily -> (z, multipliers)"""
            nonlocal n_rows
            while True:
                z, lam = master.solve()
                hi = (z[:nU] > self.u_max + 1e-9) & ~in_master[:nU]
                lo = (z[:nU] < self.u_min - 1e-9) & ~in_master[nU:]
                if not (hi.any() or lo.any()):
                    break
                if hi.any():
                    master.add_rows(I[:nU][hi], np.full(int(hi.sum()), self.u_max))
                    bound_rows.append((n_rows, np.flatnonzero(hi), 1.0))
                    n_rows += int(hi.sum())
                    in_master[:nU] |= hi
                if lo.any():
                    master.add_rows(-I[:nU][lo], np.full(int(lo.sum()), -self.u_min))
                    bound_rows.append((n_rows, np.flatnonzero(lo), -1.0))
                    n_rows += int(lo.sum())
                    in_master[nU:] |= lo
            if self.world > 1:                      # every rank solved the same master; keep them bit-identical
                z = rdist.broadcast_from_rank0(z, self.device, self.group)
            return z, lam

        u_last = z_prev = None
        for it in range(max_cuts + 1):
            t0 = time.perf_counter()
            z, lam = solve_master()
            info["master_s"] += time.perf_counter() - t0
            u_vec, s = z[:nU], z[nU]
            if not with_cvar:
                break
            t0 = time.perf_counter()
            slot = free.pop() if free else None
            phi, tstar, g = self.evaluate(G, W, tile, base, u_vec, slot)
            info["oracle_s"] += time.perf_counter() - t0
            viol = phi - self.c_s * s - self.rhs0
            # Stall: the cut added last did not move the master's solution (<= STALL_STEP in every variable) although it
            # was violated -- the oracle has returned a cut the master already holds, i.e. the violation left (1.4e-10 on
            # the bench batch) is the accuracy of the master's own NNLS, and further cuts are repeats.  Treated as
            # converged; it is what lets ``tol`` sit at 1e-9 without the loop ever burning max_cuts on a floor above it.
            if z_prev is not None and tol < viol <= STALL_VIOLATION and np.abs(z - z_prev).max() <= STALL_STEP:
                status = "solved (stalled)"
                if verbose:
                    print(f"   cut {it:3d}: violation {viol:+.3e} and the last cut moved nothing: converged")
                break
            z_prev = z
            if verbose:
                step = np.abs(u_vec - u_last).max() if it else np.nan
                u_last = u_vec.copy()
                print(f"   cut {it:3d}: CVaR {phi:+.6e} slack {s:.3e} violation {viol:+.3e}  |u - u_prev| {step:.2e}")
            if viol <= tol:
                # The loop stops on a violation it can resolve (the device selects the tail on fp32-rounded m: cuts are
                # exact as cuts, optimal as subgradients only to ~1e-9 at M = 1e5, so tol cannot go far below 1e-8 there).
                # The cut just evaluated is already paid for: where it still bites (0 < violation <= tol) it joins the
                # master and the master is solved ONCE more, without another oracle call -- at the subproblem where the
                # CVaR rows switch on that last cut is worth 2.4e-5 -> 2.4e-7 in u; at convergence the recycled cuts
                # leave no violation and nothing is added.
                if viol > final_cut_above and it < max_cuts:
                    t0 = time.perf_counter()
                    master.add_rows(np.concatenate([g, [-self.c_s]])[None, :], [self.rhs0 + (math.fsum(g * u_vec) - phi)])
                    if slot is not None:
                        cut_rows.append((n_rows, slot))
                    n_rows += 1
                    n_cuts += 1
                    z, lam = solve_master()
                    u_vec, s = z[:nU], z[nU]
                    info["master_s"] += time.perf_counter() - t0
                break
            if it == max_cuts:
                status = "maximum cuts reached"
                break
            # phi(u) >= phi_k + g_k.(u - u_k)  =>  g_k.u - c_s s <= rhs0 + g_k.u_k - phi_k
            master.add_rows(np.concatenate([g, [-self.c_s]])[None, :], [self.rhs0 + (math.fsum(g * u_vec) - phi)])
            if slot is not None:
                cut_rows.append((n_rows, slot))
            n_rows += 1
            n_cuts += 1
        if self.recycle and with_cvar:
            # keep the cuts that carry a multiplier at the solution (newest first), plus the newest few
            # (keep_idle > 0: hysteresis -- a cut that carried a multiplier in one of the last keep_idle solves stays.  Tried
            # against the 2-cycles some batches end in, iterates ~1e-6 apart: two sets of active cuts alternating, the
            # subproblems being accurate to the cut tolerance in VALUE only; no keep rule removed them on every batch, so
            # the cheapest one stays the default.)
            for row, sl in cut_rows:
                active = row < lam.shape[0] and lam[row] > 1e-12
                self.idle[sl] = 0 if active else self.idle.get(sl, 0) + 1
            act = [sl for row, sl in reversed(cut_rows) if self.idle[sl] <= self.keep_idle]
            recent = [sl for _, sl in reversed(cut_rows)][:self.keep_recent]
            keep = []
            for sl in act + recent:
                if sl not in keep:
                    keep.append(sl)
            keep = keep[:self.keep_max]
            if self.world > 1:                      # same decision on every rank (slot numbering is identical)
                pad = np.full(self.keep_max + 1, -1.0)
                pad[0] = len(keep)
                pad[1:1 + len(keep)] = keep
                pad = rdist.broadcast_from_rank0(pad, self.device, self.group)
                keep = [int(v) for v in pad[1:1 + int(pad[0])]]
            self.keep = keep
            self.idle = {sl: self.idle[sl] for sl in keep}
            info["recycled"] = len(kept)
        # t_risk: VaR + slack where the CVaR rows are present (y, t eliminated at their optimum); the relaxed QP of
        # the first iterations and the 'baseline' QP leave it undetermined (no row and no cost touches it): 0
        t_risk = float(tstar + s) if slack_row else 0.0
        # the master's multipliers, for whoever wants to certify the solution against the full QP (tests/_host_cuts.py:
        # kkt_certificate): per cut (ring slot, lambda), the slack row, the control bounds that entered
        lam_full = np.zeros(n_rows)
        lam_full[:lam.shape[0]] = lam
        info["multipliers"] = {"cuts": [(sl, float(lam_full[row])) for row, sl in cut_rows],
                               "slack": float(lam_full[0]) if slack_row else 0.0,
                               "bounds": [(idx, sgn, lam_full[r0:r0 + idx.size].copy()) for r0, idx, sgn in bound_rows],
                               "uncertified_cuts": n_cuts + len(kept) - len(cut_rows)}
        info.update(us=u_vec.reshape(self.S, self.n_u).copy(), slack=float(s), t_risk=t_risk,
                    cuts=n_cuts, phi=float(phi), status=status, loop="python")
        return info

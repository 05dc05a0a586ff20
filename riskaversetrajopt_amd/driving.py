"""Driving SAA model — the reference's ``class Model`` (car/driving.py:83-456)
with its sample-axis hot path on the MI355X.

``Model(M, method, alpha)`` samples its own uncertain parameters from the global
``np.random`` stream in the reference's draw order (driving.py:84-120).  The
reference fixes S as a module constant; here it is a keyword (default 20).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, assemble, cvar_cuts, qp, stats
from . import driving_params as P

n_x, n_u = P.n_x, P.n_u
OSQP_TOL = P.OSQP_TOL
std_matrix_ped_initial_state = np.sqrt(P.variance_ped_initial_state)   # driving.py:50-51
BETA = 3e-2                                                            # driving.py:94


from .drone_risk import TILE, num_pairs, num_tiles, untile  # noqa: E402  (same packed layout)


def sample_uncertain_parameters(M, method='saa', S=P.S, rng=None):
    """The sampling half of ``Model.__init__`` (driving.py:95-120), vectorised
    with the reference's draw order -> (states_init (M,8), omegas_speed (M,),
    omegas_repulsive (M,), DWs (M,S,8))."""
    rng = np.random if rng is None else rng
    dt = P.T / S
    omegas_speed = rng.uniform(P.omega_speed_nom - P.omega_speed_del,
                               P.omega_speed_nom + P.omega_speed_del, M)
    omegas_repulsive = rng.uniform(P.omega_repulsive_nom - P.omega_repulsive_del,
                                   P.omega_repulsive_nom + P.omega_repulsive_del, M)
    states_init = np.repeat(P.state_init[None, :], M, axis=0)
    if method == 'saa':
        states_init[:, 4:] += rng.randn(M, 4) * np.diag(std_matrix_ped_initial_state)[None, :]
    DWs = np.sqrt(dt) * rng.randn(M, S, n_x)
    if method == 'baseline':
        DWs = 0 * DWs
        omegas_speed = 0 * omegas_speed
        omegas_repulsive = 0 * omegas_repulsive
    return states_init, omegas_speed, omegas_repulsive, DWs


def sample_uncertain_parameters_device(M, S, seed=0, device='cuda:0', want_dW=True):
    """Synthetic batch (driving.py:84-120 distributions) drawn in HBM by the library's Philox sampler
    (rato_car_sample), in kernel layout: dW [S][2][M], x0_ped [4][M], w_speed [M], w_rep [M] (fp32).
    ``want_dW=False``: no noise array (``Model.from_device(..., noise_seed=seed)`` regenerates it in the kernel)."""
    lib = _lib.load()
    dev = torch.device(device)
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    dW = e(S, 2, M) if want_dW else None
    x0, w_speed, w_rep = e(4, M), e(M), e(M)
    mean = (C.c_float * 4)(*[float(v) for v in P.state_init[4:]])
    std = (C.c_float * 4)(*[float(v) for v in np.diag(std_matrix_ped_initial_state)])
    with torch.cuda.device(dev):
        _lib.check(lib.rato_car_sample(M, S, float(P.T / S), int(seed), float(P.omega_speed_nom),
                                       float(P.omega_speed_del), float(P.omega_repulsive_nom),
                                       float(P.omega_repulsive_del), mean, std, _lib.ptr(dW), _lib.ptr(x0),
                                       _lib.ptr(w_speed), _lib.ptr(w_rep), _lib.current_stream()), "rato_car_sample")
    return dW, x0, w_speed, w_rep


def to_soa_inputs(states_init, omegas_speed, omegas_repulsive, DWs, device):
    """Reference layouts -> kernel layouts (fp32, sample fastest).  Only rows 6..7
    of DWs are used by sigma (driving.py:180-184)."""
    DWs = torch.as_tensor(np.asarray(DWs), device=device)
    dW = DWs[:, :, 6:8].permute(1, 2, 0).contiguous().float()
    x0 = torch.as_tensor(np.asarray(states_init), device=device)
    x0_ped = x0[:, 4:8].t().contiguous().float()
    ws = torch.as_tensor(np.asarray(omegas_speed), device=device).contiguous().float()
    wr = torch.as_tensor(np.asarray(omegas_repulsive), device=device).contiguous().float()
    return dW, x0_ped, ws, wr


class Model:
    def __init__(self, M, method='saa', alpha=0.05, S=P.S, device='cuda:0', rng=None,
                 samples=None, verbose=False, check_finite=False):
        self.check_finite = check_finite        # scan every linearization for NaN/Inf -> RatoNonFiniteError (scp.py)
        if verbose:
            print("Initializing Model with")
            print("> method =", method)
            print("> alpha  =", alpha)
        self.method = method
        self.u_max = P.u_max
        self.u_min = -self.u_max
        self.alpha = alpha
        self.beta = BETA
        self.S, self.dt, self.M = S, P.T / S, M
        self.device = torch.device(device)
        self._lib = _lib.load()
        if samples is None:
            samples = sample_uncertain_parameters(M, method, S, rng)
        if samples != 'device':
            self.states_init, self.omegas_speed, self.omegas_repulsive, self.DWs = samples
            ego0 = np.asarray(self.states_init)[:, :4]
            if not np.all(ego0 == ego0[0:1]):
                raise ValueError("the ego initial state must be sample-independent (driving.py:104-110)")
            self._ego_init = ego0[0].astype(np.float64)
            self._dW, self._x0, self._ws, self._wr = to_soa_inputs(*samples, self.device)
        self._scratch = torch.empty(self._lib.rato_car_ego_scratch_floats(S), dtype=torch.float32,
                                    device=self.device)

    @classmethod
    def from_device(cls, S, dW, x0_ped, w_speed, w_rep, method='saa', alpha=0.05, noise_seed=None):
        """Batch resident in HBM in kernel layout.  ``dW=None`` with ``noise_seed``: the pedestrian noise of
        ``sample_uncertain_parameters_device(seed=noise_seed)`` is regenerated inside the rollout kernel
        (rato_car_eval_philox); the linearization kernels still need a materialised dW."""
        self = cls(w_speed.numel(), method, alpha, S=S, device=w_speed.device, samples='device')
        self._ego_init = P.state_init[:4].astype(np.float64)
        self._x0, self._ws, self._wr = (_lib.require_f32_device(t, n) for t, n in
                                        ((x0_ped, "x0_ped"), (w_speed, "w_speed"), (w_rep, "w_rep")))
        self._dW = _lib.require_f32_device(dW, "dW") if dW is not None else None
        if dW is None and noise_seed is None:
            raise ValueError("from_device needs dW or a noise_seed to regenerate it from")
        self._noise_seed = None if noise_seed is None else int(noise_seed)
        return self

    # ---- layout helpers (driving.py:122-143) -------------------------------
    def convert_us_vec_to_us_mat(self, us_vec):
        return np.reshape(np.asarray(us_vec), (n_u, self.S), 'F').T.copy()

    def convert_us_mat_to_us_jaxvec(self, us_mat):
        return np.reshape(np.asarray(us_mat), (self.S * n_u), 'C')

    def initial_guess_us_mat(self):
        return np.zeros((self.S, n_u)) + (self.u_max + self.u_min) / 2.0 + 1e-2

    # ---- plumbing ----------------------------------------------------------
    def _params(self, M):
        """a fresh rato_car_params (a copy of a template built once per (M, S, dt, beta, ego state): see drone_risk)"""
        key = (M, self.S, self.dt, self.beta, tuple(float(v) for v in self._ego_init))
        cache = self.__dict__.setdefault("_params_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 64:
                cache.clear()
            t = cache[key] = self._params_build(M)
        return _lib.CarParams.from_buffer_copy(t)

    def _params_build(self, M):
        p = _lib.CarParams()
        p.M, p.S, p.dt, p.beta = M, self.S, self.dt, self.beta
        p.speed_ped_des = P.speed_ped_des
        p.d_min = float(P.min_separation_distance)
        p.tol = OSQP_TOL
        goal = np.concatenate((P.position_ego_goal, P.velocity_ego_goal))
        for i in range(4):
            p.ego_init[i] = float(self._ego_init[i])
            p.ego_goal[i] = float(goal[i])
            p.ego_init64[i] = float(self._ego_init[i])
        p.dt64, p.beta64 = float(self.dt), float(self.beta)
        p.speed_ped_des64, p.d_min64 = float(P.speed_ped_des), float(P.min_separation_distance)
        return p

    def _us_device(self, us_mat):
        if isinstance(us_mat, torch.Tensor) and us_mat.is_cuda:
            us = us_mat.float().contiguous()
        else:
            us = torch.as_tensor(np.ascontiguousarray(np.asarray(us_mat), dtype=np.float32), device=self.device)
        if tuple(us.shape) != (self.S, n_u):
            raise ValueError(f"us_mat must be ({self.S},{n_u}), got {tuple(us.shape)}")
        return us

    def _empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    # ---- rollout + separation distances (K3) -------------------------------
    def eval_device(self, us_mat, want_xs=False, want_g=False, inputs=None, out=None, stats_request=None):
        """-> (Z [M], xs [S+1][8][M] or None, g [S][M] or None), device tensors.  ``out``: a dict whose ``_Z`` / ``_g``
        buffers are reused.  ``stats_request`` = (workspace, record, alpha): the call also leaves the ``rato_risk_stats``
        record of Z in ``record`` -- in the SAME launch for small batches without trajectories
        (rato_car_eval_stats_in_launch), by ``rato_risk_stats`` behind the kernel otherwise (``mc_step_device``)."""
        dW, x0, ws, wr = inputs if inputs is not None else (self._dW, self._x0, self._ws, self._wr)
        M = ws.numel()
        us = self._us_device(us_mat)
        o = out if out is not None else {}

        def reuse(key, shape):
            t = o.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t
            return self._empty(*shape)

        Z = reuse("_Z", (M,))
        xs = self._empty(self.S + 1, n_x, M) if want_xs else None
        g = reuse("_g", (self.S, M)) if want_g else None
        o["_Z"] = Z
        if g is not None:                                # (a call without g keeps the reusable g buffer of an earlier one)
            o["_g"] = g
        p = self._params(M)
        if stats_request is not None:        # (workspace, record, alpha[, in_launch])
            stats.request_in_launch(p, *stats_request[:3], flags=(stats.STATS_IN_LAUNCH if (len(stats_request) > 3 and
                                                                                             stats_request[3]) else 0))
        if dW is None:                                   # noise regenerated in the kernel (Philox, csrc/philox.h)
            _lib.check(self._lib.rato_car_eval_philox(C.byref(p), _lib.ptr(us), self._noise_seed, float(self.dt),
                                                      _lib.ptr(x0), _lib.ptr(ws), _lib.ptr(wr),
                                                      _lib.ptr(self._scratch), _lib.ptr(Z), _lib.ptr(xs), _lib.ptr(g),
                                                      _lib.current_stream()), "rato_car_eval_philox")
            if stats_request is not None:
                wsp, rec, alpha = stats_request[:3]
                stats.risk_stats_device(Z, alpha, workspace=wsp, out=rec)
            return Z, xs, g
        _lib.check(self._lib.rato_car_eval(C.byref(p), _lib.ptr(us), _lib.ptr(dW), _lib.ptr(x0), _lib.ptr(ws),
                                           _lib.ptr(wr), _lib.ptr(self._scratch), _lib.ptr(Z), _lib.ptr(xs),
                                           _lib.ptr(g), _lib.current_stream()), "rato_car_eval")
        return Z, xs, g

    def mc_step_device(self, us_mat, alpha=None, out=None, workspace=None, stats_out=None, inputs=None, in_launch=False):
        """One Monte-Carlo validation step on the device (driving.py:630-671: rollout -> max_t(-distance) -> fraction
        satisfied / VaR / AVaR) as ONE library call -- for small batches one launch (ego tables, tiled rollout and the
        exact selection).  -> (Z [M], record double[N_STATS]), device tensors."""
        alpha = self.alpha if alpha is None else alpha
        M = (inputs[2] if inputs is not None else self._ws).numel()
        if workspace is None:
            workspace = stats.new_workspace(M, self.device)
        if stats_out is None:
            stats_out = torch.empty(stats.N_STATS, dtype=torch.float64, device=self.device)
        Z, _, _ = self.eval_device(us_mat, inputs=inputs, out=out, stats_request=(workspace, stats_out, alpha, in_launch))
        return Z, stats_out

    def eval_batch_device(self, us_batch, alpha=None, want_stats=True, out=None, workspace=None):
        """K control sequences on the model's batch in ONE call (rato_car_eval_batch; the reference's Monte-Carlo report,
        driving.py:675-740, evaluates its 4 alpha x 30 repeats one at a time).  ``us_batch`` (K, S, n_u) ->
        (Z [K][M] device, records [K][N_STATS] device double or None); row k equals ``eval_device`` /
        ``stats.risk_stats_device`` on sequence k to the bit."""
        dW, x0, ws, wr = self._dW, self._x0, self._ws, self._wr
        if dW is None:
            raise _lib.RatoError("eval_batch_device reads a materialised dW (this Model regenerates its noise)")
        alpha = self.alpha if alpha is None else alpha
        M = ws.numel()
        if isinstance(us_batch, torch.Tensor) and us_batch.is_cuda:
            us = us_batch.float().contiguous()
        else:
            us = torch.as_tensor(np.ascontiguousarray(np.asarray(us_batch), dtype=np.float32), device=self.device)
        if us.dim() != 3 or tuple(us.shape[1:]) != (self.S, n_u):
            raise ValueError(f"us_batch must be (K,{self.S},{n_u}), got {tuple(us.shape)}")
        K = us.shape[0]
        o = out if out is not None else {}
        Z = o.get("_Zb")
        if Z is None or tuple(Z.shape) != (K, M):
            Z = self._empty(K, M)
        rec = workspace_ = None
        if want_stats:
            rec = o.get("_recb")
            if rec is None or tuple(rec.shape) != (K, stats.N_STATS):
                rec = torch.empty((K, stats.N_STATS), dtype=torch.float64, device=self.device)
            workspace_ = workspace if workspace is not None else o.get("_wsb")
            if workspace_ is None:
                workspace_ = stats.new_workspace(M, self.device)
        o["_Zb"], o["_recb"], o["_wsb"] = Z, rec, workspace_
        p = self._params(M)
        _lib.check(self._lib.rato_car_eval_batch(
            C.byref(p), K, _lib.ptr(us), _lib.ptr(dW), _lib.ptr(x0), _lib.ptr(ws), _lib.ptr(wr), _lib.ptr(Z), M, float(alpha),
            float(stats.SATISFIED_THRESHOLD), _lib.ptr(workspace_), workspace_.numel() if workspace_ is not None else 0,
            _lib.ptr(rec), _lib.current_stream()), "rato_car_eval_batch")
        return Z, rec

    def us_to_state_trajectories(self, us_mat):
        """driving.py:205-214 -> (M, S+1, n_x)."""
        _, xs, _ = self.eval_device(us_mat, want_xs=True)
        return xs.permute(2, 0, 1).double().cpu().numpy()

    def us_to_state_trajectory(self, us_mat, state_init, omega_speed, omega_repulsive, dWs):
        """driving.py:186-203, one sample -> (S+1, n_x)."""
        state_init = np.asarray(state_init, dtype=np.float64)
        if not np.array_equal(state_init[:4], self._ego_init):
            raise ValueError("ego initial state differs from the model's")
        inputs = to_soa_inputs(state_init[None], [omega_speed], [omega_repulsive], np.asarray(dWs)[None], self.device)
        _, xs, _ = self.eval_device(us_mat, want_xs=True, inputs=inputs)
        return xs[:, :, 0].double().cpu().numpy()

    def final_constraints(self, xs):
        goal = np.concatenate((P.position_ego_goal, P.velocity_ego_goal))
        return np.asarray(xs)[..., -1, :4] - goal

    def separation_distances_at_all_times(self, xs):
        """driving.py:232-236: distances along GIVEN trajectories, xs (S+1, n_x) -> (S,); also batched
        (M, S+1, n_x) -> (M, S).  (rato_car_separation_distances; the fused rollout+distance path of the hot loop is
        ``separation_distances_of_samples(us_mat)`` / ``eval_device``.)"""
        xs = np.asarray(xs)
        if xs.shape[-1] != n_x or xs.shape[-2] != self.S + 1:
            raise ValueError(f"xs must be (S+1, n_x) = ({self.S + 1}, {n_x}) or (M, S+1, n_x), got {xs.shape}")
        single = xs.ndim == 2
        if single:
            xs = xs[None]
        M = xs.shape[0]
        xs_d = torch.as_tensor(xs, device=self.device).permute(1, 2, 0).float().contiguous()    # [S+1][8][M]
        dist_d = self._empty(self.S, M)
        p = self._params(M)
        _lib.check(self._lib.rato_car_separation_distances(C.byref(p), _lib.ptr(xs_d), _lib.ptr(dist_d),
                                                           _lib.current_stream()), "rato_car_separation_distances")
        out = dist_d.t().double().cpu().numpy()
        return out[0] if single else out

    def separation_distances_of_samples(self, us_mat):
        """Rollout of the model's samples under ``us_mat`` fused with driving.py:232-236 -> (M, S)  (= -g)."""
        _, _, g = self.eval_device(us_mat, want_g=True)
        return -g.t().double().cpu().numpy()

    # ---- linearization (K4) ------------------------------------------------
    TILED_NOISE = True        # (class-level switch for A/B runs and the equality test)

    def _tiled_noise(self, dW, M):
        """The [tile][2S][64] copy of the MODEL'S OWN noise that the row-parallel kernel reads; made once, kept with
        the source tensor itself (compared by identity, never by address).  A caller's ``inputs`` are not cached
        (``None``: they go through the kernel that reads dW as it lies); an in-place refill of ``self._dW`` through raw
        pointers needs ``set_noise`` / ``invalidate_noise``."""
        if dW is not self._dW:
            return None
        c = getattr(self, "_dW_tiled_cache", None)
        if c is None or c[0] is not dW or c[1] != dW._version or c[3] != (M, self.S):
            n = int(self._lib.rato_car_tiled_noise_floats(M, self.S))
            t = torch.empty(n, dtype=torch.float32, device=dW.device)
            _lib.check(self._lib.rato_car_tile_noise(_lib.ptr(dW), M, self.S, _lib.ptr(t), _lib.current_stream()),
                       "rato_car_tile_noise")
            c = (dW, dW._version, t, (M, self.S))
            self._dW_tiled_cache = c
        return c[2]

    def invalidate_noise(self):
        """Forget every copy derived from ``self._dW`` (after an in-place refill of the noise array)."""
        self._dW_tiled_cache = None

    def set_noise(self, dW):
        """Replace the batch's Brownian increments (kernel layout [S][2][M], fp32, on the model's device)."""
        dW = _lib.require_f32_device(dW, "dW")
        if tuple(dW.shape) != (self.S, 2, int(self._ws.numel())):
            raise ValueError(f"dW must be ({self.S}, 2, {int(self._ws.numel())}), got {tuple(dW.shape)}")
        self._dW = dW
        self.invalidate_noise()

    def linearize_device(self, us_mat, inputs=None, cols_per_thread=0, out=None, want_Z=True, rows_out=0, stats_request=None):
        """-> dict: G [n_tiles][n_pairs][2][TILE], g_up [S][M], Z [M], final_du [4][2S], final_rhs [4]
        (device, fp32; final_* are sample-independent, i.e. already the mean)."""
        dW, x0, ws, wr = inputs if inputs is not None else (self._dW, self._x0, self._ws, self._wr)
        M, S = ws.numel(), self.S
        us = self._us_device(us_mat)
        o = out if out is not None else {}
        cpt, tile = C.c_int32(int(cols_per_thread)), C.c_int32(0)
        if self._lib.rato_car_linearize_plan(M, S, C.byref(cpt), C.byref(tile)) < 0:
            raise _lib.RatoError(f"no car linearize variant for cols_per_thread={cols_per_thread}, S={S}")
        cols_per_thread, tile = cpt.value, tile.value
        if dW is None and cols_per_thread != -1:
            raise _lib.RatoError("a Model that regenerates its noise linearizes with the row-parallel kernel only "
                                 "(cols_per_thread=-1); the column kernel reads a materialised dW")
        def reuse(key, shape):
            """a buffer of an earlier call is reused only if it has exactly the shape this launch writes"""
            t = o.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t
            return self._empty(*shape)

        g_shape = (num_tiles(M, tile), max(num_pairs(S), 1), 2, tile)
        G = o.get("G")               # packed Jacobian: tiles of >= 1 MiB start on 2 MiB boundaries (_lib.packed_buffer)
        if not (_lib.is_packed_layout(G, g_shape) and G.dtype == torch.float32):
            G = _lib.packed_buffer(g_shape, self.device)
        g_up = reuse("g_up", (S, M))
        Z = reuse("Z", (M,)) if want_Z else None
        final_du = reuse("final_du", (4, n_u * S))
        final_rhs = reuse("final_rhs", (4,))
        p = self._params(M)
        p.rows_out = int(rows_out)      # 1: g_up receives g itself (base of the cut oracle's delta form, cvar_cuts.py)
        if stats_request is not None:   # (workspace, out, alpha): the launch also computes the statistics of its Z
            stats.request_in_launch(p, *stats_request)
        tiled = self._tiled_noise(dW, M) if (dW is not None and cols_per_thread == -1 and self.TILED_NOISE) else None
        if dW is None:      # noise regenerated while a tile is staged: the same numbers, no array, no reads
            _lib.check(self._lib.rato_car_linearize_philox(
                C.byref(p), _lib.ptr(us), self._noise_seed, float(self.dt), _lib.ptr(x0), _lib.ptr(ws), _lib.ptr(wr),
                _lib.ptr(self._scratch), _lib.ptr(G), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(final_du),
                _lib.ptr(final_rhs), _lib.current_stream()), "rato_car_linearize_philox")
        elif cols_per_thread == -1 and self.TILED_NOISE and tiled is not None:
            # the row-parallel kernel reads the batch's noise re-tiled ONCE ([tile][2S][64]: a tile's noise is one block
            # instead of 2S rows M floats apart -- reads beside the store stream); same outputs bit for bit
            _lib.check(self._lib.rato_car_linearize_tiled(
                C.byref(p), _lib.ptr(us), _lib.ptr(tiled), _lib.ptr(x0), _lib.ptr(ws), _lib.ptr(wr),
                _lib.ptr(self._scratch), _lib.ptr(G), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(final_du),
                _lib.ptr(final_rhs), _lib.current_stream()), "rato_car_linearize_tiled")
        else:
            _lib.check(self._lib.rato_car_linearize(
                C.byref(p), _lib.ptr(us), _lib.ptr(dW), _lib.ptr(x0), _lib.ptr(ws), _lib.ptr(wr),
                _lib.ptr(self._scratch), _lib.ptr(G), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(final_du),
                _lib.ptr(final_rhs), int(cols_per_thread), _lib.current_stream()), "rato_car_linearize")
        if self.check_finite:
            stats.assert_finite("driving linearize", g_up, Z, final_du, final_rhs)
        return {"G": G, "g_up": g_up, "Z": Z, "final_du": final_du, "final_rhs": final_rhs, "M": M,
                "cols_per_thread": cols_per_thread, "tile": tile, "rows_out": int(rows_out)}

    def step_device(self, us_mat, alpha=None, out=None, workspace=None, stats_out=None, fused=True, **kw):
        """One single-GPU SAA step: the linearize kernel and the exact fraction satisfied / VaR / CVaR of its Z
        (driving.py:630-671).  -> (linearize result dict, stats double[N_STATS]).  ``fused`` (default; row-parallel kernel,
        small batches: rato_car_stats_in_launch): ONE launch -- the statistics are computed by extra workgroups of the linearize launch as
        soon as the last tile's Z has landed (the final rows of the driving problem are sample independent: nothing else
        follows the kernel).  Otherwise the kernel and rato_risk_stats behind it."""
        alpha = self.alpha if alpha is None else alpha
        M = int(self._ws.numel())
        cpt, tile = C.c_int32(int(kw.get("cols_per_thread", 0))), C.c_int32(0)
        self._lib.rato_car_linearize_plan(M, self.S, C.byref(cpt), C.byref(tile))
        if fused and cpt.value == -1 and self._lib.rato_car_stats_in_launch(M, self.S):
            if workspace is None:
                workspace = stats.new_workspace(M, self.device)
            if stats_out is None:
                stats_out = torch.empty(stats.N_STATS, dtype=torch.float64, device=self.device)
            r = self.linearize_device(us_mat, out=out, stats_request=(workspace, stats_out, alpha), **kw)
            return r, stats_out
        r = self.linearize_device(us_mat, out=out, **kw)
        return r, stats.risk_stats_device(r["Z"], alpha, workspace=workspace, out=stats_out)

    def expand_g_obs_du(self, G, M=None):
        """packed G -> dense host (M, S, n_u*S); small M only.  G is either the tile-blocked
        device tensor [n_tiles][n_pairs][2][TILE] or an untiled [n_pairs][2][M'] tensor/ndarray."""
        S = self.S
        if isinstance(G, torch.Tensor):
            if G.dim() == 4:
                G = untile(G, self.M if M is None else M)
            G = G.double().cpu().numpy()
        M = G.shape[-1]                                 # (n_pairs, 2, M)
        dense = np.zeros((M, S, n_u * S))
        for t in range(1, S):
            off = t * (t - 1) // 2
            blk = G[off:off + t]                        # (t, 2, M)
            dense[:, t, :n_u * t] = np.transpose(blk, (2, 0, 1)).reshape(M, n_u * t)
        return dense

    def sample_means(self, us_mat):
        """driving.py:311-313 -> (final_du (4,2S), final_low (4,), final_up (4,))."""
        r = self.linearize_device(us_mat)
        rhs = r["final_rhs"].double().cpu().numpy()
        return r["final_du"].double().cpu().numpy(), rhs, rhs.copy()

    def get_all_constraints_coeffs(self, us_mat, state_init, omega_speed, omega_repulsive, dWs):
        """driving.py:260-298 for ONE sample -> (v_final_du (4,2S), val_final_lower (4,),
        val_final_upper (4,), g_obs_du (S,2S), g_up (S,))."""
        inputs = to_soa_inputs(np.asarray(state_init)[None], [omega_speed], [omega_repulsive],
                               np.asarray(dWs)[None], self.device)
        r = self.linearize_device(us_mat, inputs=inputs)
        rhs = r["final_rhs"].double().cpu().numpy()
        g_obs_du = self.expand_g_obs_du(r["G"], 1)[0]
        return (r["final_du"].double().cpu().numpy(), rhs, rhs.copy(), g_obs_du,
                r["g_up"][:, 0].double().cpu().numpy())

    def get_all_constraints_coeffs_batched(self, us_mat):
        """vmap over the model's samples (driving.py:305-307), dense; small M only."""
        r = self.linearize_device(us_mat)
        return (self.expand_g_obs_du(r["G"], r["M"]),
                r["g_up"].t().double().cpu().numpy())

    # ---- L3: sparse QP assembly (driving.py:243-258, 301-421) --------------
    SLACK_PENALTY = 1000.0      # driving.py:387-388

    def _assemble(self, us_mat, relax):
        r = self.linearize_device(us_mat)
        M = r["M"]
        G = untile(r["G"], M).double().cpu().numpy()[:, :, None, :]          # (n_pairs, 2, 1, M)
        g_up = r["g_up"].double().cpu().numpy()[None]                        # (1, S, M)
        return assemble.saa_constraints(
            r["final_du"].double().cpu().numpy(), r["final_rhs"].double().cpu().numpy(), G, g_up,
            n_u=n_u, S=self.S, M=M, alpha=self.alpha, method=self.method, kappa=1.0, baseline_pad=0.0,
            u_min=self.u_min, u_max=self.u_max, relax=relax)

    def get_objective_coeffs(self):
        """driving.py:375-397 -> (P csc, q)."""
        return assemble.objective(n_u, self.S, self.M, self.dt, P.R, self.SLACK_PENALTY)

    def get_constraints_coeffs(self, us_mat, scp_iter):
        """driving.py:399-421 -> (A csc, l, u).  scp_iter < 1 zeroes rows [n_x:] with n_x = 8 although
        there are only 4 final rows, so the CVaR sum row and the first three -y_i rows survive
        (:411-415); the reference's ``ls *= 0`` turns -inf into nan there, which OSQP's projection
        treats like the l = u = 0 used here."""
        if scp_iter < 1:                            # pattern differs (zeroed rows are dropped): host path
            return self._assemble(us_mat, ('zero', n_x))
        fast = getattr(self, "_fast", None)
        if fast is None:
            A0, l0, u0 = self._assemble(us_mat, None)
            fast = assemble.FastAssembler(A0, l0, u0, n_c=4, n_u=n_u, n_g=2, R=1, S=self.S, M=self.M,
                                          saa=self.method == 'saa')
            self._fast = fast
        if not fast.ok or self.S < 2 or 64 * (1 * (self.S - 1) + 1) * 4 > 160 * 1024:   # rato_emit_csc_values' LDS limit
            return self._assemble(us_mat, None)
        r = self.linearize_device(us_mat)
        M, S = r["M"], self.S
        vals = self._empty(M * S * (S - 1))
        _lib.check(self._lib.rato_emit_csc_values(_lib.ptr(r["G"]), None, 0, r["tile"], 2, 1, S, M, 1.0,
                                                  _lib.ptr(vals), _lib.current_stream()), "rato_emit_csc_values")
        g_up = r["g_up"].t().contiguous().double().cpu().numpy()[:, None, :]          # (M, 1, S)
        return fast.assemble(vals.cpu().numpy(), r["final_du"].double().cpu().numpy(),
                             r["final_rhs"].double().cpu().numpy(), g_up, kappa=1.0, baseline_pad=0.0, relax=None)

    def get_constraints_coeffs_host(self, us_mat, scp_iter):
        """Host (NumPy) assembly from the untiled Jacobian — the checker for the fast path."""
        return self._assemble(us_mat, ('zero', n_x) if scp_iter < 1 else None)

    def get_all_constraints_coeffs_all(self, us_mat):
        """driving.py:301-373 -> dense (constraints_dparams, low, up) without the control bounds; small M only."""
        if self.M > 2000:
            raise MemoryError("the dense QP matrix is O(M^2); use get_constraints_coeffs (sparse)")
        A, l, u = self._assemble(us_mat, None)
        k = n_u * self.S
        return A[:-k].toarray(), l[:-k], u[:-k]

    # ---- L4: host QP (driving.py:423-456) ------------------------------------
    def define_problem(self, us_mat_p, scp_iter=0, verbose=False):
        self.P, self.q = self.get_objective_coeffs()
        self.A, self.l, self.u = self.get_constraints_coeffs(us_mat_p, scp_iter)
        if scp_iter == 0 or scp_iter == 1:          # the sparsity pattern changes: set up again (:429-437)
            self.osqp_prob = qp.OSQP()
            self.osqp_prob.setup(self.P, self.q, self.A, self.l, self.u, eps_abs=OSQP_TOL, eps_rel=OSQP_TOL,
                                 linsys_solver="qdldl", warm_start=True, verbose=verbose, polish=P.OSQP_POLISH)
        else:
            self.osqp_prob.update(l=self.l, u=self.u)
            self.osqp_prob.update(Ax=self.A.data)
        return True

    def solve(self, verbose=False):
        S = self.S
        self.res = self.osqp_prob.solve()
        if self.res.info.status != 'solved':
            print("[solve]: Problem infeasible.")
        us_sol = self.convert_us_vec_to_us_mat(self.res.x[:(n_u * S)])
        ys, t_risk_sol = self.res.x[(n_u * S):-2], self.res.x[-1]
        if verbose:
            print("y_min =", np.min(ys))
            print("slack_var =", self.res.x[-2])
        return us_sol, t_risk_sol

    # ---- L4 at large M: reduced (u, slack) problem with device CVaR cuts ----------------------
    def shard(self, group=None):
        """This Model is one shard of a sample-sharded batch (see drone_risk.Model.shard): the cutting-plane oracle
        of ``solve_reduced`` then runs across the ranks.  (The final rows are sample independent: nothing else to merge.)"""
        import torch.distributed as tdist
        from . import dist as rdist
        rdist.check_equal_shards(self.M, group)          # raises on every rank if the shards differ
        rdist.check_equal_shards(self.S, group, what="horizons S")   # (... the lengths of every exchanged buffer)
        self._group, self._world = group, tdist.get_world_size(group)
        # buffers a single-process solve_reduced may have left behind are single-process shaped (pinned HOST sums that
        # the partial-sum kernel writes into directly): a sharded solve must not inherit them
        self._cut_solver = self._gen_buffers = self._lin_buffers = self._define_host = None
        return self

    def ego_final_rows(self, us_mat):
        """The final-state rows of the ego in double precision -> (final_du (4, 2S), final_rhs (4,)):  the ego carries no
        noise, so x_S[0:4] and its control Jacobian are sample independent (the mean of driving.py:311 is a no-op) and
        O(S^2) numbers -- folded on the host, the way the row kernel's workgroup 0 folds them in fp32.
        final_rhs = -(x_S - goal) + final_du . u  (driving.py:283-288)."""
        S, dt = self.S, float(self.dt)
        us = np.asarray(us_mat, dtype=np.float64).reshape(S, n_u)
        x0, y0, v0, ph0 = (float(a) for a in self._ego_init)
        v = v0 + dt * np.concatenate(([0.0], np.cumsum(us[:, 0])))          # v_k, phi_k, k = 0..S
        ph = ph0 + dt * np.concatenate(([0.0], np.cumsum(us[:, 1])))
        cs, sn = np.cos(ph[:S]), np.sin(ph[:S])
        xS = np.array([x0 + dt * np.sum(v[:S] * cs), y0 + dt * np.sum(v[:S] * sn), v[S], ph[S]])
        # d x_S / d u_{t,0} = dt^2 sum_{k > t} cos phi_k,   d x_S / d u_{t,1} = -dt^2 sum_{k > t} v_k sin phi_k  (y alike)
        after = lambda a: np.concatenate((np.cumsum(a[::-1])[::-1][1:], [0.0]))
        E = np.zeros((4, S, n_u))
        E[0, :, 0], E[0, :, 1] = dt * dt * after(cs), -dt * dt * after(v[:S] * sn)
        E[1, :, 0], E[1, :, 1] = dt * dt * after(sn), dt * dt * after(v[:S] * cs)
        E[2, :, 0] = dt
        E[3, :, 1] = dt
        E = E.reshape(4, S * n_u)
        goal = np.concatenate((P.position_ego_goal, P.velocity_ego_goal)).astype(np.float64)
        return E, -(xS - goal) + E @ us.reshape(-1)

    def solve_reduced(self, us_mat_p, scp_iter=1, tol=1e-9, verbose=False, delta=True, rollout=None):
        """One SCP iteration without the O(M) QP (see cvar_cuts.py / drone_risk.Model.solve_reduced).
        scp_iter < 1 zeroes every separation row (driving.py:411-415), i.e. no CVaR constraint.
        ``method='baseline'`` (driving.py:320-329): the rows (G_i u)_t <= g_up_{i,t} of every sample as the one
        constraint max_i m_i(u) <= 0.  (At scp_iter 0 the reference's ``[n_x:]`` with n_x = 8 leaves the rows
        t = 0..3 of sample 0 in its baseline QP; they are inactive at the initial guess and are not kept here.)

        ``rollout`` (default: on with the delta form and a materialised dW): NO Jacobian is formed at all.  The cut
        oracle re-runs the rollout at ``us_mat_p`` in fp64 from the samples (rato_car_rowmax_rollout /
        rato_car_tail_rows_rollout: 344 bytes per sample at S = 40 instead of the 6240 of the packed Jacobian) and the
        sample-independent final rows come from ``ego_final_rows``."""
        dW, x0, ws, wr = self._dW, self._x0, self._ws, self._wr
        if rollout is None:
            rollout = bool(delta and dW is not None and self.S >= 2)
        if rollout and not (delta and dW is not None):
            raise ValueError("the rollout form of the oracle needs delta=True and a materialised dW")
        M, S = int(ws.numel()), self.S
        cs = getattr(self, "_cut_solver", None)
        if cs is None:
            cs = cvar_cuts.CvarCutSolver(self._lib, self.device, n_u=n_u, S=S, M=M, ld=M, R=1, alpha=self.alpha,
                                         dt=self.dt, Rcost=P.R, slack_penalty=self.SLACK_PENALTY,
                                         u_min=self.u_min, u_max=self.u_max,
                                         group=getattr(self, "_group", None), world=getattr(self, "_world", 1),
                                         mode=self.method, rhs0=0.0)
            self._cut_solver = cs
        u_lin = np.asarray(us_mat_p, dtype=np.float64) if delta else None
        if rollout:
            cs.rollout = ("driving", self._params(M), dW, x0, ws, wr)
            cs.check_finite = self.check_finite          # (no linearization to scan: the oracle's statistics are checked)
            final_du, final_rhs = self.ego_final_rows(us_mat_p)
            if self.check_finite and not (np.isfinite(final_du).all() and np.isfinite(final_rhs).all()):
                raise _lib.RatoNonFiniteError("driving final rows: non-finite values (RATO_ENONFINITE)")
            # unconditionally: the upload of u_k to the device happens only while cs.rollout is set, so a table-form call
            # at the same u before this one must not leave the rollout kernels reading a stale uk_dev
            cs.begin(u_lin, scp_iter >= 1 and getattr(self, "_world", 1) == 1)
            info = cs.solve(None, None, 0, None, final_du, final_rhs, u_lin=u_lin, with_cvar=(scp_iter >= 1), tol=tol,
                            verbose=verbose)
            info["final_du"], info["final_rhs"] = final_du, final_rhs    # (the equality rows: certificate.certify)
            return info["us"], info["t_risk"], info
        cs.rollout = None
        r = self.linearize_device(us_mat_p, out=getattr(self, "_lin_buffers", None), rows_out=1 if delta else 0)
        self._lin_buffers = r
        cs.set_linearization_point(u_lin)
        if scp_iter >= 1 and getattr(self, "_world", 1) == 1:
            cs.enqueue_relinearize(r["G"], None, r["tile"], r["g_up"])      # one device round trip with the read-backs below
        info = cs.solve(r["G"], None, r["tile"], r["g_up"], r["final_du"].double().cpu().numpy(),
                        r["final_rhs"].double().cpu().numpy(), u_lin=u_lin,
                        with_cvar=(scp_iter >= 1), tol=tol, verbose=verbose)
        return info["us"], info["t_risk"], info

    def certify_reduced(self, info):
        """Matrix-free KKT certificate of the last ``solve_reduced`` (table-free oracle, an iteration with the CVaR rows)
        against the reference's full QP (driving.py:330-373): certificate.py."""
        from . import certificate
        return certificate.certify(self._cut_solver, info, info["final_du"], info["final_rhs"], kappa=1.0)

    # ---- Monte-Carlo validation (driving.py:623-671) -----------------------
    def monte_carlo_cost(self, us_mat):
        # driving.py:623-629: the script has ONE dt (driving_params.py:14, dt = T / S), shared with the rollout
        us = np.asarray(us_mat)
        return self.dt * float(np.sum(np.diag(P.R)[None, :] * us * us))

    def monte_carlo_separation_constraints_verification(self, us_mat):
        Z, _, _ = self.eval_device(us_mat)
        Zh = Z.double().cpu().numpy()
        return Zh <= 1e-6, Zh

    def monte_carlo_statistics(self, us_mat, alpha=None):
        """rollout -> Z -> fraction satisfied, VaR, CVaR (``mc_step_device``); a NaN record on finite Z is recovered
        through ``stats.risk_stats``."""
        alpha = self.alpha if alpha is None else alpha
        Z, rec = self.mc_step_device(us_mat, alpha)
        r = rec.cpu().numpy()
        if np.isnan(r[0]):
            return stats.risk_stats(Z, alpha)
        return dict(zip(stats._STAT_NAMES, r.tolist()))

    monte_carlo_avar = staticmethod(stats.monte_carlo_avar)


def L2_error_us(us_mat, us_mat_prev):
    """driving.py:459-464."""
    error = np.mean(np.linalg.norm(us_mat - us_mat_prev, axis=-1))
    return error / np.mean(np.linalg.norm(us_mat, axis=-1))

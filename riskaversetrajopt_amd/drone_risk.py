"""Drone SAA model — the reference's ``class Model`` (drone/drone_risk.py:70-469)
with its sample-axis hot path on the MI355X.

Same constructor, method names, argument meaning and return shapes as the
reference for the L2 layer (rollout, constraints, per-sample linearization,
Monte-Carlo closures).  Device work goes through the C ABI of
``include/rato_saa.h``; torch is used for device memory and streams only.
Host-facing methods return NumPy arrays in the reference's shapes; ``*_device``
methods return the kernels' SoA tensors (sample index fastest).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, assemble, cvar_cuts, qp, stats
from . import drone_params as P

n_x, n_u, n_obs = P.n_x, P.n_u, P.n_obs
OSQP_TOL = P.OSQP_TOL


TILE = 256      # RATO_TILE of include/rato_saa.h (the row-parallel drone kernel uses 64-wide tiles)


def num_pairs(S):
    return S * (S - 1) // 2


def num_tiles(M, tile=TILE):
    return (M + tile - 1) // tile


def untile(G, M):
    """tile-blocked [n_tiles][rows...][TILE] -> [rows...][M] (a view-free copy; small M / tests)."""
    nd = G.dim()
    perm = list(range(1, nd - 1)) + [0, nd - 1]
    rows = G.shape[1:-1]
    return G.permute(*perm).reshape(*rows, G.shape[0] * G.shape[-1])[..., :M]


def pad4(M):
    """Row stride of the SoA arrays: samples padded to a multiple of 4 so that a lane
    can move 4 consecutive samples with one 16-byte access."""
    return (M + 3) // 4 * 4


def to_soa_inputs(DWs, masses, obs_Qs, device):
    """Reference layouts -> kernel layouts (fp32, sample index fastest, row stride
    ld = pad4(M); padding samples are inert: zero noise, unit mass, zero Q).
    DWs (M,S,6) -> dW [S][3][ld] (only rows 3..5 are used by sigma, drone_risk.py:136);
    obs_Qs (M,n_obs,3,3) -> Qsym [n_obs][3][ld] = (Q00, Q01+Q10, Q11) of [:2,:2] (:174).
    Returns (dW, mass, Qsym, M)."""
    DWs = torch.as_tensor(np.asarray(DWs), device=device)
    M, S = DWs.shape[0], DWs.shape[1]
    ld = pad4(M)
    dW = torch.zeros((S, 3, ld), dtype=torch.float32, device=device)
    dW[:, :, :M] = DWs[:, :, 3:6].permute(1, 2, 0).float()
    mass = torch.ones(ld, dtype=torch.float32, device=device)
    mass[:M] = torch.as_tensor(np.asarray(masses), device=device).float()
    Q = torch.as_tensor(np.asarray(obs_Qs), device=device)
    Qs = torch.stack([Q[:, :, 0, 0], Q[:, :, 0, 1] + Q[:, :, 1, 0], Q[:, :, 1, 1]], dim=1)  # (M,3,n_obs)
    Qsym = torch.zeros((n_obs, 3, ld), dtype=torch.float32, device=device)
    Qsym[:, :, :M] = Qs.permute(2, 1, 0).float()
    return dW, mass, Qsym, M


class Model:
    def __init__(self, S, DWs, masses, obs_Qs, method='saa', alpha=0.1, device='cuda:0',
                 verbose=False, check_finite=False):
        # drone_risk.py:71-93
        # check_finite: scan g_up / Z / partial sums of every linearization for NaN/Inf and raise
        # RatoNonFiniteError (the SCP drivers of scp.py switch it on; off in throughput runs: it costs a sync)
        self.check_finite = check_finite
        if verbose:
            print("Initializing Model with")
            print("> method =", method)
            print("> alpha  =", alpha)
            print("> S      =", S)
        self.method = method
        self.S = S
        self.dt = P.T / S
        self.u_max = P.u_max
        self.u_min = -self.u_max
        self.alpha = alpha
        self.beta = P.beta
        self.drag_coefficient = P.drag_coefficient
        self.device = torch.device(device)
        self._lib = _lib.load()
        if DWs is not None:
            self.DWs, self.masses, self.obs_Qs = DWs, masses, obs_Qs
            self._dW, self._mass, self._Qsym, self.M = to_soa_inputs(DWs, masses, obs_Qs, self.device)
            if self._dW.shape[0] != S:
                raise ValueError(f"DWs has {self._dW.shape[0]} steps, Model has S={S}")

    @classmethod
    def from_device(cls, S, dW, mass, Qsym, method='saa', alpha=0.1, M=None, noise_seed=None, sampler_dt=None):
        """Batch already resident in HBM in kernel layout (throughput runs):
        dW [S][3][ld], mass [ld], Qsym [n_obs][3][ld]; M <= ld samples are used.
        ``dW=None`` with ``noise_seed``: the Brownian increments of ``drone_utils.sample_uncertain_parameters_device(
        seed=noise_seed, dt=sampler_dt)`` are REGENERATED inside the rollout kernel (rato_drone_eval_philox) instead
        of being stored: Monte-Carlo validation batches (drone_risk.py:647-662) then cost 40 B per sample of HBM
        instead of 12 S + 40; the linearization kernels still need a materialised dW."""
        self = cls(S, None, None, None, method, alpha, device=mass.device)
        self.DWs = self.masses = self.obs_Qs = None
        self._mass, self._Qsym = (_lib.require_f32_device(t, n) for t, n in ((mass, "mass"), (Qsym, "Qsym")))
        self._dW = _lib.require_f32_device(dW, "dW") if dW is not None else None
        if dW is None and noise_seed is None:
            raise ValueError("from_device needs dW or a noise_seed to regenerate it from")
        self._noise_seed = None if noise_seed is None else int(noise_seed)
        self._sampler_dt = P.T / S if sampler_dt is None else float(sampler_dt)
        self.M = mass.numel() if M is None else M
        return self

    # ---- layout helpers (drone_risk.py:95-120) -----------------------------
    def convert_us_vec_to_us_mat(self, us_vec):
        return np.reshape(np.asarray(us_vec), (n_u, self.S), 'F').T.copy()

    def convert_us_mat_to_us_jaxvec(self, us_mat):
        return np.reshape(np.asarray(us_mat), (self.S * n_u), 'C')

    def initial_guess_us_mat(self):
        us = np.zeros((self.S, n_u))
        us[:, :(n_u - 1)] = (self.u_max + self.u_min) / 2.0 + 1e-2
        return us

    # ---- plumbing ----------------------------------------------------------
    def _inputs(self, inputs):
        return inputs if inputs is not None else (self._dW, self._mass, self._Qsym, self.M)

    def _params(self, M, ld, rows_out=0):
        """a fresh rato_drone_params for this Model (callers set the stats_* fields on it): a copy of a template built once
        per (M, ld, rows_out, S, dt, beta, drag) -- ~40 ctypes field stores cost 15-20 us, more than a small kernel"""
        key = (M, ld, int(rows_out), self.S, self.dt, self.beta, self.drag_coefficient)
        cache = self.__dict__.setdefault("_params_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 64:
                cache.clear()
            t = cache[key] = self._params_build(M, ld, rows_out)
        return _lib.DroneParams.from_buffer_copy(t)

    def _params_build(self, M, ld, rows_out=0):
        p = _lib.DroneParams()
        p.M, p.ld, p.S = M, ld, self.S
        p.rows_out = int(rows_out)      # 1: the linearize kernels write g (not g_up = -g + G u_k) into their g_up buffer
        p.dt, p.beta, p.drag = self.dt, self.beta, self.drag_coefficient
        p_kp, p_kd = -float(P.feedback_gain[0, 0]), -float(P.feedback_gain[0, 3])
        p.kp, p.kd = p_kp, p_kd
        p.tol = OSQP_TOL
        for i in range(6):
            p.x_init[i] = float(P.x_init[i])
            p.x_final[i] = float(P.x_final[i])
        for j in range(n_obs):
            p.obs_xy[j][0] = float(P.obs_positions[j, 0])
            p.obs_xy[j][1] = float(P.obs_positions[j, 1])
        # the same constants in fp64 (the entry points that compute in double precision read these)
        p.dt64, p.beta64, p.drag64, p.kp64, p.kd64, p.tol64 = self.dt, self.beta, self.drag_coefficient, p_kp, p_kd, OSQP_TOL
        for i in range(6):
            p.x_init64[i] = float(P.x_init[i])
            p.x_final64[i] = float(P.x_final[i])
        for j in range(n_obs):
            p.obs_xy64[j][0] = float(P.obs_positions[j, 0])
            p.obs_xy64[j][1] = float(P.obs_positions[j, 1])
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

    def _rows(self, *shape, M):
        """[..][ld] output rows: the kernels never write the ld - M padding lanes, so they are zeroed once here
        (whole-buffer consumers such as the non-finite scan then see defined values)."""
        return self._empty(*shape) if shape[-1] == M else torch.zeros(shape, dtype=torch.float32, device=self.device)

    # ---- rollout + constraint values (K1) ----------------------------------
    def eval_device(self, us_mat, want_xs=False, want_g=False, inputs=None, out=None, stats_request=None):
        """-> (Z [M], xs [S+1][6][M] or None, g [n_obs][S][M] or None): device tensors
        (views of row-stride-ld buffers).
        ``out``: a dict whose ``_Z`` / ``_g`` buffers (shapes of an earlier call) are reused.
        ``stats_request`` = (workspace, record, alpha): the call also leaves the ``rato_risk_stats`` record of Z in
        ``record`` (double[N_STATS], device) -- in the SAME launch for small batches without trajectories
        (rato_drone_eval_stats_in_launch), by ``rato_risk_stats`` behind the kernel otherwise (``mc_step_device``)."""
        dW, mass, Qsym, M = self._inputs(inputs)
        ld = mass.numel()
        us = self._us_device(us_mat)
        o = out if out is not None else {}

        def reuse(key, shape):
            t = o.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t
            return self._empty(*shape)

        Z = reuse("_Z", (ld,))
        xs = self._empty(self.S + 1, n_x, ld) if want_xs else None
        g = reuse("_g", (n_obs, self.S, ld)) if want_g else None
        o["_Z"] = Z
        if g is not None:                                # (a call without g keeps the reusable g buffer of an earlier one)
            o["_g"] = g
        p = self._params(M, ld)
        if stats_request is not None:        # (workspace, record, alpha[, in_launch])
            stats.request_in_launch(p, *stats_request[:3], flags=(stats.STATS_IN_LAUNCH if (len(stats_request) > 3 and
                                                                                             stats_request[3]) else 0))
        if dW is None:                                   # noise regenerated in the kernel (Philox, csrc/philox.h)
            _lib.check(self._lib.rato_drone_eval_philox(C.byref(p), _lib.ptr(us), self._noise_seed, self._sampler_dt,
                                                        _lib.ptr(mass), _lib.ptr(Qsym), _lib.ptr(Z), _lib.ptr(xs),
                                                        _lib.ptr(g), _lib.current_stream()), "rato_drone_eval_philox")
            if stats_request is not None:
                ws, rec, alpha = stats_request[:3]
                stats.risk_stats_device(Z[:M], alpha, workspace=ws, out=rec)
            return Z[:M], (xs[..., :M] if want_xs else None), (g[..., :M] if want_g else None)
        _lib.check(self._lib.rato_drone_eval(C.byref(p), _lib.ptr(us), _lib.ptr(dW), _lib.ptr(mass),
                                             _lib.ptr(Qsym), _lib.ptr(Z), _lib.ptr(xs), _lib.ptr(g),
                                             _lib.current_stream()), "rato_drone_eval")
        return Z[:M], (xs[..., :M] if want_xs else None), (g[..., :M] if want_g else None)

    def mc_step_device(self, us_mat, alpha=None, out=None, workspace=None, stats_out=None, inputs=None, in_launch=False):
        """One Monte-Carlo validation step on the device (drone_risk.py:711-714: rollout -> max over (obstacle, t) ->
        fraction satisfied / VaR / AVaR) as ONE library call -- for small batches one launch.
        -> (Z [M], record double[N_STATS]), device tensors; ``out`` / ``workspace`` / ``stats_out`` are reused when given
        (a captured step must pass them)."""
        alpha = self.alpha if alpha is None else alpha
        M = self._inputs(inputs)[3]
        if workspace is None:
            workspace = stats.new_workspace(M, self.device)
        if stats_out is None:
            stats_out = torch.empty(stats.N_STATS, dtype=torch.float64, device=self.device)
        Z, _, _ = self.eval_device(us_mat, inputs=inputs, out=out, stats_request=(workspace, stats_out, alpha, in_launch))
        return Z, stats_out

    def eval_batch_device(self, us_batch, alpha=None, want_stats=True, out=None, workspace=None):
        """K control sequences on the model's batch in ONE call (rato_drone_eval_batch): what the reference's Monte-Carlo
        report does one sequence at a time for its 4 alpha x 30 repeats (drone_risk.py:697-725).  ``us_batch``
        (K, S, n_u) -> (Z [K][M] device, records [K][N_STATS] device double or None).  Row k equals, to the bit, what
        ``eval_device`` / ``stats.risk_stats_device`` give for sequence k."""
        dW, mass, Qsym, M = self._inputs(None)
        if dW is None:
            raise _lib.RatoError("eval_batch_device reads a materialised dW (this Model regenerates its noise)")
        alpha = self.alpha if alpha is None else alpha
        if isinstance(us_batch, torch.Tensor) and us_batch.is_cuda:
            us = us_batch.float().contiguous()
        else:
            us = torch.as_tensor(np.ascontiguousarray(np.asarray(us_batch), dtype=np.float32), device=self.device)
        if us.dim() != 3 or tuple(us.shape[1:]) != (self.S, n_u):
            raise ValueError(f"us_batch must be (K,{self.S},{n_u}), got {tuple(us.shape)}")
        K, ld = us.shape[0], mass.numel()
        o = out if out is not None else {}
        Z = o.get("_Zb")
        if Z is None or tuple(Z.shape) != (K, ld):
            Z = self._empty(K, ld)
        rec = None
        if want_stats:
            rec = o.get("_recb")
            if rec is None or tuple(rec.shape) != (K, stats.N_STATS):
                rec = torch.empty((K, stats.N_STATS), dtype=torch.float64, device=self.device)
            if workspace is None:
                workspace = o.get("_wsb")
            if workspace is None:
                workspace = stats.new_workspace(M, self.device)
        o["_Zb"], o["_recb"], o["_wsb"] = Z, rec, workspace
        p = self._params(M, ld)
        _lib.check(self._lib.rato_drone_eval_batch(
            C.byref(p), K, _lib.ptr(us), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym), _lib.ptr(Z), ld, float(alpha),
            float(stats.SATISFIED_THRESHOLD), _lib.ptr(workspace), workspace.numel() if workspace is not None else 0,
            _lib.ptr(rec), _lib.current_stream()), "rato_drone_eval_batch")
        return Z[:, :M], rec

    def us_to_state_trajectories(self, us_mat):
        """drone_risk.py:157-162 -> (M, S+1, n_x)."""
        _, xs, _ = self.eval_device(us_mat, want_xs=True)
        return xs.permute(2, 0, 1).double().cpu().numpy()

    def us_to_state_trajectory(self, us_mat, mass, dWs):
        """drone_risk.py:139-155, one sample -> (S+1, n_x)."""
        Q = np.zeros((1, n_obs, 3, 3))
        inputs = to_soa_inputs(np.asarray(dWs)[None], np.asarray([mass]), Q, self.device)
        _, xs, _ = self.eval_device(us_mat, want_xs=True, inputs=inputs)
        return xs[:, :, 0].double().cpu().numpy()

    def final_constraints(self, xs):
        return np.asarray(xs)[..., -1, :] - P.x_final

    def obstacle_avoidance_constraints(self, xs, obs_Q):
        """drone_risk.py:198-213: xs (S+1,n_x), obs_Q (n_obs,3,3) -> (n_obs,S);
        also batched: (M,S+1,n_x), (M,n_obs,3,3) -> (M,n_obs,S)."""
        xs, obs_Q = np.asarray(xs), np.asarray(obs_Q)
        single = xs.ndim == 2
        if single:
            xs, obs_Q = xs[None], obs_Q[None]
        M = xs.shape[0]
        _, _, Qsym, _ = to_soa_inputs(np.zeros((M, 1, 6)), np.ones(M), obs_Q, self.device)
        ld = Qsym.shape[-1]
        xs_d = torch.zeros((self.S + 1, n_x, ld), dtype=torch.float32, device=self.device)
        xs_d[..., :M] = torch.as_tensor(xs, device=self.device).permute(1, 2, 0).float()
        g = self._empty(n_obs, self.S, ld)
        p = self._params(M, ld)
        _lib.check(self._lib.rato_drone_obstacle_constraints(C.byref(p), _lib.ptr(xs_d), _lib.ptr(Qsym),
                                                             _lib.ptr(g), _lib.current_stream()),
                   "rato_drone_obstacle_constraints")
        out = g[..., :M].permute(2, 0, 1).double().cpu().numpy()
        return out[0] if single else out

    # ---- linearization (K2 + K6) -------------------------------------------
    def linearize_plan(self, M, ld, cols_per_thread=0, samples_per_lane=0):
        """-> (nblocks, cols_per_thread, samples_per_lane, tile) as the library resolves them."""
        cpt, spl, tile = C.c_int32(cols_per_thread), C.c_int32(samples_per_lane), C.c_int32(0)
        nblk = self._lib.rato_drone_linearize_plan(M, self.S, ld, C.byref(cpt), C.byref(spl), C.byref(tile))
        if nblk < 0:
            raise _lib.RatoError(f"no linearize variant for cols_per_thread={cols_per_thread}, "
                                 f"samples_per_lane={samples_per_lane}, ld={ld}")
        return nblk, cpt.value, spl.value, tile.value

    TILED_NOISE = True        # (class-level switch for A/B runs and the equality test)

    def _tiled_noise(self, dW, M, ld):
        """The [tile][3S][64] copy of the MODEL'S OWN noise that the row-parallel kernel reads; made once, kept with
        the source tensor itself (compared by identity: an address can be recycled by the allocator, a live tensor
        cannot).  A caller's ``inputs`` are never cached -- ``None`` sends them through the kernel that reads dW as it
        lies.  Whoever rewrites ``self._dW`` in place through raw pointers (the library's samplers do not bump a
        tensor's version counter) must call ``set_noise`` / ``invalidate_noise``."""
        if dW is not self._dW:
            return None
        c = getattr(self, "_dW_tiled_cache", None)
        if c is None or c[0] is not dW or c[1] != dW._version or c[3] != (M, ld, self.S):
            n = int(self._lib.rato_drone_tiled_noise_floats(M, self.S))
            t = torch.empty(n, dtype=torch.float32, device=dW.device)
            _lib.check(self._lib.rato_drone_tile_noise(_lib.ptr(dW), M, ld, self.S, _lib.ptr(t), _lib.current_stream()),
                       "rato_drone_tile_noise")
            c = (dW, dW._version, t, (M, ld, self.S))
            self._dW_tiled_cache = c
        return c[2]

    def invalidate_noise(self):
        """Forget every copy derived from ``self._dW`` (after an in-place refill of the noise array)."""
        self._dW_tiled_cache = None

    def set_noise(self, dW):
        """Replace the batch's Brownian increments (kernel layout [S][3][ld], fp32, on the model's device)."""
        dW = _lib.require_f32_device(dW, "dW")
        if tuple(dW.shape) != (self.S, 3, self._mass.numel()):
            raise ValueError(f"dW must be ({self.S}, 3, {self._mass.numel()}), got {tuple(dW.shape)}")
        self._dW = dW
        self.invalidate_noise()

    def linearize_device(self, us_mat, inputs=None, cols_per_thread=0, samples_per_lane=0, out=None,
                         want_Z=True, events=None, factored=None, want_A22=False, reduce=True, rows_out=0, stats_request=None):
        """One SAA linearization pass on the device (drone_risk.py:239-296).

        -> dict of device tensors:
           G  [n_tiles][n_pairs][2][n_obs][TILE]  packed causal Jacobian, tile-blocked (rato_saa.h), or — when
              ``factored`` (default with the row-parallel kernel) — Phi [n_tiles][n_pairs][2][TILE] together
              with W [n_obs][S][2][M]:  d g[j,t]/d u[s,a] = W[j,t,a] * Phi[t,s,a]  (2.67x fewer bytes at S=50;
              ``packed_jacobian(result)`` materialises the products)
           g_up [n_obs][S][M], Z [M]  (views of row-stride-ld buffers),
           du_sum [S][6] (float64: sums over samples of dx_S/du_{s,axis}),
           rhs_sum [6]   (float64: sums of -v_final + v_final_du.u)
        ``want_A22`` (factored output only): also keep the step-Jacobian table A22 [S][2][M] — with W and g_up
           the whole linearization in 11 S numbers per sample (rato_drone_rowmax_implicit evaluates G.u from it).
        ``out``: a dict returned by an earlier call (same shapes) whose buffers are reused.
        ``events``: optional (start, end) torch.cuda.Event pair recorded tightly around the
        linearize launch on the launch stream (bench.py's roofline timing).
        ``reduce=False``: leave the per-block partial sums unreduced (``part``); ``step_device`` folds their
        reduction into the single launch of the risk statistics.
        ``rows_out=1``: the ``g_up`` buffer receives the constraint values g at ``us_mat`` instead of
        g_up = -g + (grad g).u (the base of the cut oracle's delta form, cvar_cuts.py).
        ``stats_request`` = (workspace, out, alpha) (row-parallel kernel only, M <= stats.FUSED_MAX_M): the launch also
        leaves the ``rato_risk_stats`` record of its Z in ``out`` (extra workgroups of the same launch, ``step_device``).
        """
        dW, mass, Qsym, M = self._inputs(inputs)
        ld, S = mass.numel(), self.S
        us = self._us_device(us_mat)
        nblk, cpt, spl, tile = self.linearize_plan(M, ld, cols_per_thread, samples_per_lane)
        if dW is None and cpt != -1:
            raise _lib.RatoError("a Model that regenerates its noise linearizes with the row-parallel kernel only "
                                 "(cols_per_thread=-1, S <= 126); the column kernels read a materialised dW")
        if factored is None:
            factored = (cpt == -1)
        if factored and cpt != -1:
            raise _lib.RatoError("the factored output exists for the row-parallel kernel (cols_per_thread=-1) only")
        o = out if out is not None else {}
        g_shape = (num_tiles(M, tile), max(num_pairs(S), 1), 2, tile) if factored else \
            (num_tiles(M, tile), max(num_pairs(S), 1), 2, n_obs, tile)

        def reuse(key, shape, alloc):
            """a buffer of an earlier call is reused only if it has exactly the shape this launch writes (``out`` may
            come from another batch size or S: the kernels would write past a smaller buffer)"""
            t = o.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t
            return alloc()

        # the packed Jacobian: tiles of >= 1 MiB start on 2 MiB boundaries (a strided view; _lib.packed_buffer)
        G = o.get("G")
        if not (_lib.is_packed_layout(G, g_shape) and G.dtype == torch.float32):
            G = _lib.packed_buffer(g_shape, self.device)
        Wf = None
        if factored:
            Wf = reuse("_W", (n_obs, S, 2, ld), lambda: self._empty(n_obs, S, 2, ld))
        A22 = None
        if want_A22:
            if not factored:
                raise _lib.RatoError("A22 goes with the factored output (row-parallel kernel)")
            A22 = reuse("_A22", (S, 2, ld), lambda: self._empty(S, 2, ld))
        g_up = reuse("_g_up", (n_obs, S, ld), lambda: self._rows(n_obs, S, ld, M=M))
        Z = reuse("_Z", (ld,), lambda: self._rows(ld, M=M)) if want_Z else None
        part = reuse("part", (nblk, 6 * S + 6), lambda: self._empty(nblk, 6 * S + 6))
        if o.get("sums") is not None and (o["sums"].numel() != 6 * S + 6 or o["sums"].dtype != torch.float64):
            o = dict(o, sums=None)
        p = self._params(M, ld, rows_out)
        if stats_request is not None:
            stats.request_in_launch(p, *stats_request)
        # (re-tiled BEFORE the start event: the one-time copy is not part of the linearize launch that is timed)
        tiled = self._tiled_noise(dW, M, ld) if (dW is not None and cpt == -1 and self.TILED_NOISE) else None
        if events is not None:
            events[0].record()
        if dW is None:      # noise regenerated while a tile is staged: the same numbers, no array, no reads
            _lib.check(self._lib.rato_drone_linearize_philox(
                C.byref(p), _lib.ptr(us), self._noise_seed, self._sampler_dt, _lib.ptr(mass), _lib.ptr(Qsym),
                _lib.ptr(G), _lib.ptr(Wf), _lib.ptr(A22), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(part),
                _lib.current_stream()), "rato_drone_linearize_philox")
        elif cpt == -1 and self.TILED_NOISE and tiled is not None:
            # the row-parallel kernel reads the batch's noise re-tiled ONCE ([tile][3S][64]: a tile's noise is one block
            # instead of 3S rows ld floats apart -- reads beside the store stream cost more than their bytes); same outputs
            _lib.check(self._lib.rato_drone_linearize_tiled(
                C.byref(p), _lib.ptr(us), _lib.ptr(tiled), _lib.ptr(mass), _lib.ptr(Qsym), _lib.ptr(G),
                _lib.ptr(Wf), _lib.ptr(A22), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(part), _lib.current_stream()),
                "rato_drone_linearize_tiled")
        else:
            _lib.check(self._lib.rato_drone_linearize(
                C.byref(p), _lib.ptr(us), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym), _lib.ptr(G), _lib.ptr(Wf),
                _lib.ptr(A22), _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(part), cpt, spl, _lib.current_stream()),
                "rato_drone_linearize")
        if events is not None:
            events[1].record()
        if reduce:
            sums = stats.sum_partials(part, out=o.get("sums"))       # [6S+6] fp64, one launch
        else:
            sums = o["sums"] if o.get("sums") is not None else torch.empty(6 * S + 6, dtype=torch.float64,
                                                                            device=self.device)
        if self.check_finite:
            stats.assert_finite("drone linearize", g_up, Z, part)
        return {"G": G, "g_up": g_up[..., :M], "Z": (Z[:M] if want_Z else None),
                "du_sum": sums[:6 * S].view(S, 6), "rhs_sum": sums[6 * S:], "sums": sums,
                "part": part, "M": M, "_g_up": g_up, "_Z": Z, "cols_per_thread": cpt,
                "samples_per_lane": spl, "tile": tile, "factored": bool(factored), "rows_out": int(rows_out),
                "W": (Wf[..., :M] if factored else None), "_W": Wf,
                "A22": (A22[..., :M] if A22 is not None else None), "_A22": A22}

    def linearize_generators_device(self, us_mat, inputs=None, out=None, rows_out=0, tables=True, defer_check=False):
        """Generators-only linearization (rato_drone_linearize_generators): A22 [S][3][M] (holding 1 - a22), W [n_obs][S][2][M],
        g_up [n_obs][S][M], Z [M] and the sample sums -- everything ``solve_reduced`` needs -- without the S(S-1)
        Jacobian entries per sample (60 B instead of 245 B of HBM traffic per sample-step).  Same dict keys as
        ``linearize_device`` with G = None.  ``tables=False``: W and g_up are not written (keys None) -- a reduced solve whose
        cut oracle re-runs the rollout only needs Z and the sample sums.  ``defer_check``: with ``check_finite`` the count
        of non-finite outputs is left on the device (key ``_nonfinite``) for a caller that reads it back with its other
        results instead of paying a synchronisation here."""
        dW, mass, Qsym, M = self._inputs(inputs)
        if dW is None:
            raise _lib.RatoError("linearize_generators_device reads a materialised dW (this Model regenerates its noise)")
        ld, S = mass.numel(), self.S
        us = self._us_device(us_mat)
        o = out if out is not None else {}

        def reuse(key, shape, alloc):
            t = o.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.is_contiguous():
                return t
            return alloc()

        A22 = reuse("_A22", (S, 3, ld), lambda: self._empty(S, 3, ld))
        Wf = reuse("_W", (n_obs, S, 2, ld), lambda: self._empty(n_obs, S, 2, ld)) if tables else None
        g_up = reuse("_g_up", (n_obs, S, ld), lambda: self._rows(n_obs, S, ld, M=M)) if tables else None
        Z = reuse("_Z", (ld,), lambda: self._rows(ld, M=M))
        nblk = (M + 255) // 256
        part = reuse("part", (nblk, 6 * S + 6), lambda: self._empty(nblk, 6 * S + 6))
        if o.get("sums") is not None and (o["sums"].numel() != 6 * S + 6 or o["sums"].dtype != torch.float64):
            o = dict(o, sums=None)
        p = self._params(M, ld, rows_out)
        _lib.check(self._lib.rato_drone_linearize_generators(
            C.byref(p), _lib.ptr(us), _lib.ptr(dW), _lib.ptr(mass), _lib.ptr(Qsym), _lib.ptr(A22), _lib.ptr(Wf),
            _lib.ptr(g_up), _lib.ptr(Z), _lib.ptr(part), _lib.current_stream()), "rato_drone_linearize_generators")
        sums = stats.sum_partials(part, out=o.get("sums"))
        bad = None
        if self.check_finite:
            bad = stats.enqueue_nonfinite_count("drone linearize (generators)", g_up, Z, part, out=o.get("_nonfinite"))
            if not defer_check:
                stats.raise_if_nonfinite("drone linearize (generators)", int(bad.item()))
        return {"_nonfinite": bad, "G": None, "g_up": g_up[..., :M] if tables else None, "Z": Z[:M], "du_sum": sums[:6 * S].view(S, 6),
                "rhs_sum": sums[6 * S:], "sums": sums, "part": part, "M": M, "_g_up": g_up, "_Z": Z,
                "tile": 64, "factored": True, "W": Wf[..., :M] if tables else None, "_W": Wf, "A22": A22[..., :M],
                "_A22": A22, "a22_axes": 3, "rows_out": int(rows_out)}

    def expand_final_du(self, du_sum, scale):
        """[S][6] sums -> dense (n_x, n_u*S) like the reference's v_final_du."""
        S = self.S
        d = np.asarray(du_sum, dtype=np.float64) * scale
        out = np.zeros((n_x, n_u * S))
        for a in range(3):
            out[a, a::n_u] = d[:, a]
            out[3 + a, a::n_u] = d[:, 3 + a]
        return out

    def packed_jacobian(self, r):
        """linearize_device result -> untiled packed Jacobian [n_pairs][2][n_obs][M] (device tensor), whichever
        representation the kernel wrote (for the factored one the products W * Phi are formed here)."""
        M, S = r["M"], self.S
        if not r.get("factored"):
            return untile(r["G"], M)
        Phi = untile(r["G"], M)                                   # (n_pairs, 2, M)
        t_of = torch.as_tensor(np.concatenate([np.full(t, t) for t in range(1, S)]) if S > 1 else np.zeros(0),
                               dtype=torch.long, device=Phi.device)
        Wt = r["W"].permute(1, 2, 0, 3)[t_of]                      # (n_pairs, 2, n_obs, M): W[j,t,a] at the pair's t
        return Phi[:, :, None, :] * Wt

    def expand_g_obs_du(self, G, M=None):
        """packed G -> dense host (M,n_obs,S,n_u*S); small M only.  G is either the
        tile-blocked device tensor [n_tiles][n_pairs][2][n_obs][TILE], an already untiled
        [n_pairs][2][n_obs][M'] tensor/ndarray, or a whole linearize_device result dict."""
        S = self.S
        if isinstance(G, dict):                        # a linearize_device result (either representation)
            G = self.packed_jacobian(G)
        if isinstance(G, torch.Tensor):
            if G.dim() == 5:
                G = untile(G, self.M if M is None else M)
            G = G.double().cpu().numpy()
        M = G.shape[-1]                                # (n_pairs, 2, n_obs, M)
        dense = np.zeros((M, n_obs, S, n_u * S))
        for t in range(1, S):
            off = t * (t - 1) // 2
            blk = G[off:off + t]                       # (t, 2, n_obs, M) over s < t
            for a in range(2):
                dense[:, :, t, a:n_u * t:n_u] = np.transpose(blk[:, a], (2, 1, 0))
        return dense

    def sample_means(self, us_mat):
        """drone_risk.py:294-296 -> (final_du (6,3S), final_low (6,), final_up (6,))."""
        r = self.linearize_device(us_mat)
        final_du = self.expand_final_du(r["du_sum"].cpu().numpy(), 1.0 / r["M"])
        rhs = r["rhs_sum"].cpu().numpy() / r["M"]
        return final_du, rhs, rhs.copy()

    def get_all_constraints_coeffs(self, us_mat, mass, dWs, obs_Q):
        """drone_risk.py:239-280 for ONE sample -> (v_final_du (6,3S),
        val_final_lower (6,), val_final_upper (6,), g_obs_du (n_obs,S,3S), g_up (n_obs,S))."""
        inputs = to_soa_inputs(np.asarray(dWs)[None], np.asarray([mass]), np.asarray(obs_Q)[None], self.device)
        r = self.linearize_device(us_mat, inputs=inputs)
        v_final_du = self.expand_final_du(r["du_sum"].cpu().numpy(), 1.0)
        rhs = r["rhs_sum"].cpu().numpy()
        g_obs_du = self.expand_g_obs_du(r)[0]
        g_up = r["g_up"][:, :, 0].double().cpu().numpy()
        return v_final_du, rhs, rhs.copy(), g_obs_du, g_up

    def get_all_constraints_coeffs_batched(self, us_mat):
        """vmap of the above over the model's samples (drone_risk.py:288-290), dense;
        small M only.  -> (g_obs_du (M,n_obs,S,3S), g_up (M,n_obs,S))."""
        r = self.linearize_device(us_mat)
        g_obs_du = self.expand_g_obs_du(r)
        g_up = r["g_up"].permute(2, 0, 1).double().cpu().numpy()
        return g_obs_du, g_up

    def step_device(self, us_mat, alpha=None, out=None, workspace=None, stats_out=None, events=None, fused=True, **kw):
        """One single-GPU SAA step: linearize, the sample sums (drone_risk.py:294-296) and fraction satisfied / VaR / CVaR of
        Z (:661, :663-695, drone_main_plot.py:640-652).  -> (linearize result dict, stats double[N_STATS]).
        ``fused`` (default; row-parallel kernel, batches small enough for the whole launch to be resident at once:
        rato_drone_stats_in_launch): the statistics are computed by extra workgroups of the linearize launch itself as soon
        as the last tile's Z has landed -- before the Jacobian has been stored -- and only the tiny reduction of the sample
        sums (complete when the kernel ends) follows it.  Otherwise: the kernel, then ONE launch for sums + statistics
        (rato_sums_and_risk_stats) -- beside a store-saturated kernel the selection would run slower than behind it."""
        alpha = self.alpha if alpha is None else alpha
        _, mass, _, M = self._inputs(kw.get("inputs"))
        if fused and self._lib.rato_drone_stats_in_launch(M, self.S) and \
                self.linearize_plan(M, mass.numel(), kw.get("cols_per_thread", 0), kw.get("samples_per_lane", 0))[1] == -1:
            if workspace is None:
                workspace = stats.new_workspace(M, self.device)
            if stats_out is None:
                stats_out = torch.empty(stats.N_STATS, dtype=torch.float64, device=self.device)
            r = self.linearize_device(us_mat, out=out, events=events, reduce=True,
                                      stats_request=(workspace, stats_out, alpha), **kw)
            return r, stats_out
        r = self.linearize_device(us_mat, out=out, events=events, reduce=False, **kw)
        _, st = stats.sums_and_risk_stats_device(r["part"], r["Z"], alpha, workspace=workspace, sums_out=r["sums"],
                                                 out=stats_out)
        return r, st

    # ---- hipGraph: one SCP-iteration's device work as a single replayable graph -------------
    def capture_step(self, alpha=None, cols_per_thread=0, samples_per_lane=0, factored=None, fused=True):
        """Capture linearize -> sample means -> VaR/CVaR into ONE hipGraph (torch.cuda.CUDAGraph is
        only the capture/replay plumbing; the nodes are this library's kernels).  Returns a
        ``StepGraph``: write the controls into ``.us`` (device tensor, (S, n_u)), call ``.replay()``
        and read ``.out`` (same dict as linearize_device) and ``.stats`` (double[10], rato_saa.h).
        ``fused``: see ``step_device``."""
        alpha = self.alpha if alpha is None else alpha
        us = torch.zeros((self.S, n_u), dtype=torch.float32, device=self.device)
        out = self.linearize_device(us, cols_per_thread=cols_per_thread, samples_per_lane=samples_per_lane,
                                    factored=factored)
        ws = stats.new_workspace(out["M"], self.device)
        st = torch.empty(stats.N_STATS, dtype=torch.float64, device=self.device)
        kw = dict(cols_per_thread=out["cols_per_thread"], samples_per_lane=out["samples_per_lane"],
                  factored=out["factored"])
        self.step_device(us, alpha, out=out, workspace=ws, stats_out=st, fused=fused, **kw)   # warm-up: uncaptured first call
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            res, _ = self.step_device(us, alpha, out=out, workspace=ws, stats_out=st, fused=fused, **kw)
        sg = StepGraph(graph, us, res, st)
        sg.workspace = ws
        return sg

    # ---- L3: sparse QP assembly (drone_risk.py:221-237, 282-423) -----------
    MULTIPLIER = 0.01           # drone_risk.py:307,353: constraint rows are scaled by 0.01
    SLACK_PENALTY = 10000.0     # :389-390

    def _host_linearization(self, us_mat):
        r = self.linearize_device(us_mat)
        M = r["M"]
        final_du = self.expand_final_du(r["du_sum"].cpu().numpy(), 1.0 / M)
        final_rhs = r["rhs_sum"].cpu().numpy() / M
        G = self.packed_jacobian(r).double().cpu().numpy()           # (n_pairs, 2, n_obs, M)
        g_up = r["g_up"].double().cpu().numpy()                      # (n_obs, S, M)
        return final_du, final_rhs, G, g_up, M

    def _assemble(self, us_mat, relax):
        final_du, final_rhs, G, g_up, M = self._host_linearization(us_mat)
        return assemble.saa_constraints(
            final_du, final_rhs, G, g_up, n_u=n_u, S=self.S, M=M, alpha=self.alpha, method=self.method,
            kappa=self.MULTIPLIER, baseline_pad=(1e-3 if self.method == 'baseline' else 0.0),
            u_min=self.u_min, u_max=self.u_max, relax=relax)

    def get_objective_coeffs(self):
        """drone_risk.py:376-399 -> (P csc, q)."""
        return assemble.objective(n_u, self.S, self.M, self.dt, P.R, self.SLACK_PENALTY)

    def get_constraints_coeffs_host(self, us_mat, scp_iter):
        """Host (NumPy) assembly from the untiled Jacobian — the checker for the fast path."""
        relax = ('scale', n_x, 1e-7, -0.1, 0.1) if scp_iter < 2 else None
        return self._assemble(us_mat, relax)

    def get_constraints_coeffs(self, us_mat, scp_iter):
        """drone_risk.py:401-423 -> (A csc, l, u) with the reference's row/column order and
        dropped-zero pattern.  scp_iter < 2 relaxes every row after the n_x final rows
        (A *= 1e-7, l = -0.1, u = 0.1, :413-417).

        After the first call the sparsity pattern is cached and an iteration only moves the value block
        that the device wrote in CSC order (rato_emit_csc_values)."""
        relax = ('scale', n_x, 1e-7, -0.1, 0.1) if scp_iter < 2 else None
        fast = getattr(self, "_fast", None)
        if fast is None:
            A0, l0, u0 = self._assemble(us_mat, None)
            saa = self.method == 'saa'
            fast = assemble.FastAssembler(A0, l0, u0, n_c=n_x, n_u=n_u, n_g=2, R=n_obs, S=self.S, M=self.M, saa=saa)
            self._fast = fast
        if not fast.ok or self.S < 2 or 64 * (n_obs * (self.S - 1) + 1) * 4 > 160 * 1024:   # rato_emit_csc_values' LDS limit
            return self._assemble(us_mat, relax)
        r = self.linearize_device(us_mat)
        M, S = r["M"], self.S
        factor = relax[2] if relax is not None else 1.0
        vals = self._empty(M * n_obs * S * (S - 1))
        ldw = r["_W"].shape[-1] if r["factored"] else 0
        _lib.check(self._lib.rato_emit_csc_values(_lib.ptr(r["G"]), _lib.ptr(r["_W"]), ldw, r["tile"], 2, n_obs,
                                                  S, M, float(self.MULTIPLIER * factor), _lib.ptr(vals),
                                                  _lib.current_stream()), "rato_emit_csc_values")
        final_du = self.expand_final_du(r["du_sum"].cpu().numpy(), 1.0 / M)
        final_rhs = r["rhs_sum"].cpu().numpy() / M
        g_up = r["g_up"].permute(2, 0, 1).contiguous().double().cpu().numpy()       # (M, n_obs, S)
        return fast.assemble(vals.cpu().numpy(), final_du, final_rhs, g_up, kappa=self.MULTIPLIER,
                             baseline_pad=(1e-3 if self.method == 'baseline' else 0.0), relax=relax)

    def get_all_constraints_coeffs_all(self, us_mat):
        """drone_risk.py:282-374 -> dense (constraints_dparams, low, up) WITHOUT the control bounds;
        O(M^2) memory like the reference, so small M only."""
        if self.M > 2000:
            raise MemoryError("the dense QP matrix is O(M^2); use get_constraints_coeffs (sparse)")
        A, l, u = self._assemble(us_mat, None)
        k = n_u * self.S
        return A[:-k].toarray(), l[:-k], u[:-k]

    # ---- L4: host QP (drone_risk.py:425-469) --------------------------------
    def define_problem(self, us_mat_p, verbose=False):
        scp_iter = 2            # define with the collision-avoidance pattern (:426-427)
        self.P, self.q = self.get_objective_coeffs()
        self.A, self.l, self.u = self.get_constraints_coeffs(us_mat_p, scp_iter)
        self.osqp_prob = qp.OSQP()
        self.osqp_prob.setup(self.P, self.q, self.A, self.l, self.u, eps_abs=OSQP_TOL, eps_rel=OSQP_TOL,
                             linsys_solver="qdldl", warm_start=True, verbose=verbose, polish=P.OSQP_POLISH)
        return True

    def update_problem(self, us_mat_p, scp_iter=0, verbose=False):
        self.P, self.q = self.get_objective_coeffs()
        self.A, self.l, self.u = self.get_constraints_coeffs(us_mat_p, scp_iter)
        self.osqp_prob.update(l=self.l, u=self.u)
        self.osqp_prob.update(Ax=self.A.data)
        return True

    def solve(self, verbose=True):
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
        """Declare this Model one shard of a sample-sharded batch (one process per GPU, torch.distributed already
        initialised, equal shard sizes): ``solve_reduced`` then merges the sample means and runs the cutting-plane
        oracle across the ranks (cvar_cuts.py); every rank returns the same iterate."""
        import torch.distributed as tdist
        from . import dist as rdist
        rdist.check_equal_shards(self.M, group)          # raises on every rank if the shards differ
        rdist.check_equal_shards(self.S, group, what="horizons S")   # (... the lengths of every exchanged buffer)
        self._group, self._world = group, tdist.get_world_size(group)
        # buffers a single-process solve_reduced may have left behind are single-process shaped (pinned HOST sums that
        # the partial-sum kernel writes into directly): a sharded solve must not inherit them
        self._cut_solver = self._gen_buffers = self._lin_buffers = self._define_host = None
        return self

    def solve_reduced(self, us_mat_p, scp_iter=2, tol=1e-9, verbose=False, implicit=True, generators_only=None,
                      delta=True, factored=None, rollout=None):
        """One SCP iteration without the O(M) QP: linearize at ``us_mat_p`` on the device, eliminate the
        y_i / t_risk of the reference's QP exactly and solve the remaining problem in (u, slack) by cutting
        planes (cvar_cuts.py): host master QP with 3S+1 variables, device oracle for the CVaR constraint.
        Same optimum as define/update_problem + solve; -> (us_sol (S,n_u), t_risk, info).
        ``method='baseline'`` (drone_risk.py:303-325): the rows 0.01 (G_i u - g_up_i)_r <= -1e-3 of every sample, as
        the one constraint max_i m_i(u) <= -0.1 (row generation with the same oracle).
        ``scp_iter < 2``: the reference's relaxation (drone_risk.py:413-417), see cvar_cuts.py.
        ``implicit``: evaluate the constraint rows of a candidate u from the step-Jacobian table (O(S) per sample,
        rato_drone_rowmax_implicit) instead of reading the packed Jacobian (O(S^2), rato_saa_rowmax).
        ``generators_only`` (default: same as ``implicit``): do not even write the Jacobian -- linearize to
        (A22, W, g) only and regenerate the few rows the subgradients need (rato_drone_tail_rows_implicit).
        ``delta``: rows as g + G (u - u_k) (the kernels write g) instead of G u - g_up (they write g_up).
        ``factored`` (with the Jacobian written): its representation, see ``linearize_device``.
        ``rollout`` (default: on whenever it applies -- implicit, generators-only, delta form, materialised noise): the
        oracle reads no linearization table at all; it re-runs the rollout at ``us_mat_p`` in fp64 from the samples
        (rato_drone_rowmax_rollout / rato_drone_tail_rows_rollout): less traffic, and no fp32 intermediate in the rows."""
        if generators_only is None:
            generators_only = implicit
        if generators_only and not implicit:
            raise ValueError("generators_only needs the implicit oracle")
        rows_out = 1 if delta else 0
        dW, mass, Qsym, _ = self._inputs(None)
        if rollout is None:
            rollout = bool(implicit and generators_only and delta and dW is not None)
        if rollout and not (implicit and delta and dW is not None):
            raise ValueError("the rollout form of the oracle needs implicit=True, delta=True and a materialised dW")
        world = getattr(self, "_world", 1)
        if rollout and world == 1 and not verbose:
            # the benchmarked configuration: define (rato_cut_define_drone) and solve (rato_cut_solve) are one native call each
            cs = self._reduced_cut_solver(int(self._inputs(None)[3]), mass.numel())
            cs.implicit = None
            rp = getattr(self, "_rollout_params", None)      # (built once: ~20 us of ctypes field stores per call otherwise)
            if rp is None or rp[0] != (cs.M, mass.numel()):
                rp = self._rollout_params = ((cs.M, mass.numel()), self._params(cs.M, mass.numel()))
            cs.rollout = ("drone", rp[1], dW, mass, Qsym)
            cs.check_finite = False
            if cs.native_loop_applies():
                return self._solve_reduced_native(cs, us_mat_p, scp_iter, tol)
        if generators_only:
            # (the table-free oracle reads neither W nor g: only Z and the sample sums are produced then)
            bufs = getattr(self, "_gen_buffers", None)
            if bufs is None and getattr(self, "_world", 1) == 1:
                # single GPU: the sample sums are consumed on the host only, so the partial-sum kernel writes them straight
                # into pinned host memory (visible after the synchronisation below) -- no read-back copy to issue
                bufs = {"sums": torch.zeros(6 * self.S + 6, dtype=torch.float64).pin_memory()}
            r = self.linearize_generators_device(us_mat_p, out=bufs, rows_out=rows_out, tables=not rollout,
                                                 defer_check=True)
            self._gen_buffers = r
        else:
            r = self.linearize_device(us_mat_p, out=getattr(self, "_lin_buffers", None), want_A22=implicit,
                                      rows_out=rows_out, factored=factored)
            self._lin_buffers = r
        M, S = r["M"], self.S
        cs = self._reduced_cut_solver(M, mass.numel())
        cs.implicit = cs.rollout = None
        if implicit:
            cs.implicit = (self._params(M, mass.numel()), mass, r["_A22"], r.get("a22_axes", 2))
        if rollout:
            cs.rollout = ("drone", self._params(M, mass.numel()), dW, mass, Qsym)
        # the linearization point and the kept cuts against this linearization: launched now, so that the read-back of the
        # sample sums below waits for both (one device round trip instead of two)
        cs.begin(np.asarray(us_mat_p, dtype=np.float64) if delta else None, scp_iter >= 2 and world == 1,
                 r["G"], r["_W"], r["tile"], r["_g_up"])
        sums = r["sums"]
        if world > 1:                                     # sample means over ALL shards, summed in rank order
            from . import dist as rdist
            sums = rdist.sum_in_rank_order(sums, getattr(self, "_group", None), agreed=True)   # 6S + 6 numbers: S agreed in shard()
        # ONE synchronisation for the sample sums, the non-finite count and (enqueued above) the kept cuts
        host = getattr(self, "_define_host", None)
        if host is None or host[0].numel() != sums.numel():
            host = (torch.zeros(sums.numel(), dtype=torch.float64).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory())
            self._define_host = host
        st = _lib.current_stream()
        if sums.is_cuda:
            _lib.copy_async(host[0], sums, st)
        bad = r.get("_nonfinite")
        if bad is not None:
            _lib.copy_async(host[1], bad, st)
        _lib.synchronize(st)
        if bad is not None:
            stats.raise_if_nonfinite("drone linearize (generators)", int(host[1][0]))
        sums = (host[0] if sums.is_cuda else sums).numpy()
        final_du = self.expand_final_du(sums[:6 * S].reshape(S, 6), 1.0 / (M * world))
        final_rhs = sums[6 * S:] / (M * world)
        info = cs.solve(r["G"], r["_W"], r["tile"], r["_g_up"], final_du, final_rhs,
                        u_lin=(np.asarray(us_mat_p, dtype=np.float64) if delta else None),
                        with_cvar=(scp_iter >= 2), tol=tol, verbose=verbose)
        info["final_du"], info["final_rhs"] = final_du, final_rhs        # (the equality rows: certificate.certify)
        return info["us"], info["t_risk"], info

    def _reduced_cut_solver(self, M, ld):
        cs = getattr(self, "_cut_solver", None)
        if cs is None:
            cs = cvar_cuts.CvarCutSolver(self._lib, self.device, n_u=n_u, S=self.S, M=M, ld=ld,
                                         R=n_obs, alpha=self.alpha, dt=self.dt, Rcost=P.R,
                                         slack_penalty=self.SLACK_PENALTY, u_min=self.u_min, u_max=self.u_max,
                                         group=getattr(self, "_group", None), world=getattr(self, "_world", 1),
                                         mode=self.method, rhs0=-1e-3 / self.MULTIPLIER)
            self._cut_solver = cs
        return cs

    def _solve_reduced_native(self, cs, us_mat_p, scp_iter, tol):
        """``solve_reduced`` for the benchmarked configuration (table-free oracle, one GPU) as TWO library calls:
        rato_cut_define_drone (controls up, generators-only linearization, sample sums into pinned memory, non-finite
        count, u_k and the kept cuts for the oracle, one synchronisation) and rato_cut_solve (the cutting-plane loop)."""
        S, M = self.S, cs.M
        b = self._native_define_buffers(cs)
        us64 = np.ascontiguousarray(us_mat_p, dtype=np.float64)
        if us64.shape != (S, n_u):
            raise ValueError(f"us_mat must be ({S},{n_u}), got {us64.shape}")
        h = cs._native_solver()
        out = cs._keep_arrays()
        K = len(cs.keep) if (scp_iter >= 2 and cs.recycle and S >= 2) else 0
        chk = bool(self.check_finite)
        # (no device-side scan here: every quantity this path consumes goes through the sample sums -- a NaN / Inf anywhere
        #  in a rollout reaches its final state and poisons them -- and the oracle checks its own m values)
        rc = self._lib.rato_cut_define_drone(
            h, us64.ctypes.data, b["us_host"].data_ptr(), b["us_dev"].data_ptr(), b["A22"].data_ptr(), None, 0,
            b["part"].data_ptr(), b["sums_host"].data_ptr(), None, None, out["keep"].ctypes.data, K, _lib.current_stream())
        _lib.check(rc, "rato_cut_define_drone")
        cs.u_lin = us64.reshape(-1).copy()
        cs._relin_pending = K if K else None
        cs.check_finite = chk
        sums = b["sums_np"]
        if chk and not np.isfinite(sums).all():
            raise _lib.RatoNonFiniteError("drone linearize (generators): non-finite sample sums (RATO_ENONFINITE)")
        final_du = self.expand_final_du(sums[:6 * S].reshape(S, 6), 1.0 / M)
        final_rhs = sums[6 * S:] / M
        info = cs._solve(None, None, 64, None, final_du, final_rhs, u_lin=us64, with_cvar=(scp_iter >= 2), tol=tol)
        info["final_du"], info["final_rhs"] = final_du, final_rhs
        return info["us"], info["t_risk"], info

    def _native_define_buffers(self, cs):
        """the device / pinned buffers rato_cut_define_drone and rato_scp_run_drone work in (made once per shape)"""
        S, M, ld = self.S, cs.M, self._mass.numel()
        b = getattr(self, "_native_define", None)
        if b is None or b["key"] != (S, M, ld):
            e = lambda *sh, dt=torch.float32: torch.empty(sh, dtype=dt, device=self.device)
            b = {"key": (S, M, ld), "us_host": torch.zeros((S, n_u), dtype=torch.float32).pin_memory(), "us_dev": e(S, n_u),
                 "A22": e(S, 3, ld), "Z": e(ld), "part": e((M + 255) // 256, 6 * S + 6),
                 "sums_host": torch.zeros(6 * S + 6, dtype=torch.float64).pin_memory(),
                 "bad_dev": e(1, dt=torch.int32), "bad_host": torch.zeros(1, dtype=torch.int32).pin_memory()}
            b["sums_np"] = b["sums_host"].numpy()
            self._native_define = b
        return b

    def scp_run_native(self, us0, iters, first_cvar=2, tol=1e-9, max_cuts=400, final_cut_above=1e-11):
        """The whole reduced SCP as ONE library call (rato_scp_run_drone: ``iters`` x [define, solve] with the per-iteration
        clocks of the reference's protocol taken natively) -- for the configuration ``solve_reduced`` runs natively (table-free
        oracle, one GPU).  -> dict(us_hist (iters, S, n_u), define_s, solve_s, oracle_s, cuts, t_risk, status) or None when the
        configuration is not the native one / the native loop handed back (rank-deficient master, a selection that gave
        up): the caller then runs the per-iteration loop."""
        dW, mass, Qsym, _ = self._inputs(None)
        if dW is None or getattr(self, "_world", 1) != 1 or self.S < 2:
            return None
        cs = self._reduced_cut_solver(int(self._inputs(None)[3]), mass.numel())
        cs.implicit = None
        rp = getattr(self, "_rollout_params", None)
        if rp is None or rp[0] != (cs.M, mass.numel()):
            rp = self._rollout_params = ((cs.M, mass.numel()), self._params(cs.M, mass.numel()))
        cs.rollout = ("drone", rp[1], dW, mass, Qsym)
        if not cs.native_loop_applies():
            return None
        S = self.S
        b = self._native_define_buffers(cs)
        us0 = np.ascontiguousarray(us0, dtype=np.float64)
        if us0.shape != (S, n_u):
            raise ValueError(f"us0 must be ({S},{n_u}), got {us0.shape}")
        h = cs._native_solver()
        out = cs._keep_arrays()
        assert C.sizeof(_lib.ScpIter) == self._lib.rato_scp_iter_bytes()
        rec = (_lib.ScpIter * max(iters, 1))()
        us_hist = np.zeros((max(iters, 1), S, n_u))
        done = C.c_int32(0)
        cs.check_finite = bool(self.check_finite)
        rc = self._lib.rato_scp_run_drone(
            h, us0.ctypes.data, int(iters), int(first_cvar), float(tol), int(max_cuts), float(final_cut_above),
            int(bool(self.check_finite)), b["us_host"].data_ptr(), b["us_dev"].data_ptr(), b["A22"].data_ptr(),
            b["part"].data_ptr(), b["sums_host"].data_ptr(), out["keep"].ctypes.data, out["idle"].ctypes.data,
            C.addressof(out["n_keep"]), us_hist.ctypes.data, C.addressof(rec), C.addressof(done), _lib.current_stream())
        # the solver's Python-side state follows the native one (a later solve_reduced continues from here)
        K = out["n_keep"].value
        cs.keep = [int(v) for v in out["keep"][:K]]
        cs.idle = {int(sl): int(c) for sl, c in zip(out["keep"][:K], out["idle"][:K])}
        cs._relin_pending = None
        n = done.value
        if n:
            cs.u_lin = (us_hist[n - 2] if n >= 2 else us0).reshape(-1).copy()      # the last linearization point
        if rc in (_lib.RATO_ERANK, _lib.RATO_ESELECT):
            _lib.synchronize()
            return None
        if rc == _lib.RATO_EINFEASIBLE:
            raise cvar_cuts.dense_qp.InfeasibleError("master QP infeasible")
        if rc == _lib.RATO_ENONFINITE:
            raise _lib.RatoNonFiniteError("reduced SCP (native loop): non-finite sample sums / constraint values (RATO_ENONFINITE)")
        _lib.check(rc, "rato_scp_run_drone")
        recs = rec[:iters]
        f = lambda k: np.array([getattr(r, k) for r in recs])
        return {"us_hist": us_hist[:iters], "define_s": f("define_s"), "solve_s": f("solve_s"), "oracle_s": f("oracle_s"),
                "master_s": f("master_s"), "cuts": f("cuts").astype(np.int64), "t_risk": f("t_risk"), "slack": f("slack"),
                "status": f("status").astype(np.int64)}

    def certify_reduced(self, info):
        """Matrix-free KKT certificate of the last ``solve_reduced`` (its ``info``; an iteration with the CVaR rows,
        before any other solve) against the reference's full QP (drone_risk.py:327-368): certificate.py."""
        from . import certificate
        return certificate.certify(self._cut_solver, info, info["final_du"], info["final_rhs"], kappa=self.MULTIPLIER)

    # ---- Monte-Carlo validation (drone_risk.py:649-695) --------------------
    def monte_carlo_cost(self, us_mat):
        us = np.asarray(us_mat)
        return P.dt * float(np.sum(np.diag(P.R)[None, :] * us * us))

    def monte_carlo_no_collisions_constraint_verification(self, us_mat):
        """vmap of drone_risk.py:656-662 -> (B_satisfied (M,) bool, max_constraint (M,))."""
        Z, _, _ = self.eval_device(us_mat)
        Zh = Z.double().cpu().numpy()
        return Zh <= 1e-6, Zh

    def monte_carlo_statistics(self, us_mat, alpha=None):
        """Fused device path: rollout -> Z -> fraction satisfied, VaR, CVaR (``mc_step_device``: one call, for small
        batches one launch).  A NaN record on finite Z (a one-launch selection that gave up) is recovered through
        ``stats.risk_stats``."""
        alpha = self.alpha if alpha is None else alpha
        Z, rec = self.mc_step_device(us_mat, alpha)
        r = rec.cpu().numpy()
        if np.isnan(r[0]):
            return stats.risk_stats(Z, alpha)
        return dict(zip(stats._STAT_NAMES, r.tolist()))

    monte_carlo_avar = staticmethod(stats.monte_carlo_avar)
    monte_carlo_var = staticmethod(stats.monte_carlo_var)


class StepGraph:
    """A captured device step (see Model.capture_step)."""

    def __init__(self, graph, us, out, stats_out):
        self.graph, self.us, self.out, self.stats = graph, us, out, stats_out

    def replay(self, us_mat=None):
        if us_mat is not None:
            self.us.copy_(torch.as_tensor(np.asarray(us_mat), dtype=torch.float32), non_blocking=True)
        self.graph.replay()
        return self.out, self.stats


def L2_error_us(us_mat, us_mat_prev):
    """drone_risk.py:471-476."""
    error = np.mean(np.linalg.norm(us_mat - us_mat_prev, axis=-1))
    return error / np.mean(np.linalg.norm(us_mat, axis=-1))

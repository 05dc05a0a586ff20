"""Hopper uncertain-friction SAA constraint — the sample-dependent part of the
reference's ``class Model`` (hopper/hopper.py:68-81,90-171,300-367,901-958) on
the MI355X.  The sample-independent NLP rows and the IPOPT glue stay on the
host and are out of scope."""
import ctypes as C

import numpy as np
import torch

from . import _lib, stats

# hopper.py:44-69
MAX_HOST_CONTACTS = 128   # RATO_HOPPER_MAX_HOST_CONTACTS (include/rato_saa.h)
S = 30
M = 30
T = 2.0
n_x = 8
n_u = 4
mu_nom = 0.10
num_mu_features = 30


def phase_times(S):
    """time_jump, time_land (hopper.py:48-49: 10, 20 at S=30)."""
    return S // 3, (2 * S) // 3


def sample_friction_fields(M, rng=None):
    """hopper.py:70-74 / :975-979, same draw order on the global stream."""
    rng = np.random if rng is None else rng
    intensities = rng.uniform(0, 1, (M, num_mu_features))
    intensities = np.sqrt(2 / num_mu_features) * intensities
    intensities = 0.025 * intensities
    thetas = rng.uniform(0, np.pi, (M, num_mu_features))
    taus = rng.uniform(0, 2 * np.pi, (M, num_mu_features))
    return intensities, thetas, taus


def sample_friction_fields_device(M, seed=1, device='cuda:0'):
    """Synthetic fields (hopper.py:70-74 distributions) drawn in HBM by the library's Philox sampler
    (rato_hopper_sample), kernel layout [30][M] (fp32)."""
    lib = _lib.load()
    dev = torch.device(device)
    a, th, tau = (torch.empty((num_mu_features, M), dtype=torch.float32, device=dev) for _ in range(3))
    with torch.cuda.device(dev):
        _lib.check(lib.rato_hopper_sample(M, int(seed), _lib.ptr(a), _lib.ptr(th), _lib.ptr(tau),
                                          _lib.current_stream()), "rato_hopper_sample")
    return a, th, tau


class Model:
    def __init__(self, M, method='baseline', alpha=0.1, S=S, fields=None, device='cuda:0', rng=None,
                 verbose=False):
        # hopper.py:91-104
        if verbose:
            print("Initializing Model with")
            print("> method =", method)
            print("> alpha  =", alpha)
        self.method, self.alpha, self.S, self.M = method, alpha, S, M
        self.time_jump, self.time_land = phase_times(S)
        self.device = torch.device(device)
        self._lib = _lib.load()
        self.num_vars = (S + 1) * n_x + S * n_u + M + 2
        if fields is None:
            fields = sample_friction_fields(M, rng)
        if fields != 'device':
            z = 0.0 if method == 'baseline' else 1.0
            self.intensities, self.thetas, self.taus = (z * np.asarray(f, dtype=np.float64) for f in fields)
            self._a, self._th, self._tau = (
                torch.as_tensor(f, device=self.device).t().contiguous().float()
                for f in (self.intensities, self.thetas, self.taus))

    @classmethod
    def from_device(cls, a, theta, tau, method='saa', alpha=0.1, S=S):
        self = cls(a.shape[1], method, alpha, S=S, fields='device', device=a.device)
        self._a, self._th, self._tau = (_lib.require_f32_device(t, n) for t, n in
                                        ((a, "a"), (theta, "theta"), (tau, "tau")))
        return self

    # ---- variable layout (hopper.py:105-132) -------------------------------
    def convert_z_to_variables(self, z):
        nx, nu = (self.S + 1) * n_x, self.S * n_u
        return z[:nx], z[nx:nx + nu], z[nx + nu:-2], z[-2], z[-1]

    def convert_z_to_xs_us_mats(self, z):
        xs_vec, us_vec, _, _, _ = self.convert_z_to_variables(np.asarray(z))
        return (np.reshape(xs_vec, (n_x, self.S + 1), 'F').T.copy(),
                np.reshape(us_vec, (n_u, self.S), 'F').T.copy())

    def end_effector_position(self, x):
        x = np.asarray(x)
        return np.stack([x[..., 0] + x[..., 3] * np.sin(x[..., 2]),
                         x[..., 1] - x[..., 3] * np.cos(x[..., 2])], axis=-1)

    def end_effector_x_derivatives(self, x):
        """Chain-rule factors of the (sample-independent) map state -> end-effector x position p = x0 + x3 sin x2
        (hopper.py:166-171) that carry dh/dpx and d2h/dpx2 to the NLP variables (x0, x2, x3) in jac_g / the
        Hessian (hopper.py:569,577-580):  -> (J (...,3) = dp/d(x0,x2,x3),  H (...,3,3) = d2p/d(x0,x2,x3)^2)."""
        x = np.asarray(x, dtype=np.float64)
        s, c = np.sin(x[..., 2]), np.cos(x[..., 2])
        J = np.stack([np.ones_like(s), x[..., 3] * c, s], axis=-1)
        H = np.zeros(x.shape[:-1] + (3, 3))
        H[..., 1, 1] = -x[..., 3] * s
        H[..., 1, 2] = H[..., 2, 1] = c
        return J, H

    def contact_chain(self, Z):
        """The factors above at the contact steps of ``contact_inputs(Z)``: J (C,3), H (C,3,3).  With the device
        outputs:  dh_ic/d(x0,x2,x3)_c = dh_dpx[i,c] J[c];  sum_i lam_ic d2h_ic/d(.)2 = D2[c] J[c] J[c]' + (sum_i lam_ic
        dh_dpx[i,c]) H[c];  mixed with fz: D1[c] J[c]."""
        xs_mat, _ = self.convert_z_to_xs_us_mats(Z)
        xc = np.concatenate([xs_mat[:self.time_jump], xs_mat[self.time_land:-1]])
        return self.end_effector_x_derivatives(xc)

    def contact_inputs(self, Z):
        """Contact-phase mask of hopper.py:305-311 -> (px (C,), forces (C,2))."""
        xs_mat, us_mat = self.convert_z_to_xs_us_mats(Z)
        ee_x = self.end_effector_position(xs_mat)[:, 0]
        px = np.concatenate([ee_x[:self.time_jump], ee_x[self.time_land:-1]])
        forces = np.concatenate([us_mat[:self.time_jump, 2:], us_mat[self.time_land:, 2:]])
        return px, forces

    # ---- device path (K5) --------------------------------------------------
    def slip_device(self, px, forces, lam=None, want_Z=True, want_h=True, want_deriv=False, reduce=True, staged=None):
        """lam: [C][M] multipliers or None.
        -> dict of device tensors: Z [M], h/dh_dfz/dh_dpx [C][M], hess [C][2] (float64).
        reduce=False: the per-workgroup partial sums of the lambda-weighted second derivatives come back as "part"
        (hess=None) for a caller that folds their second stage into the statistics launch
        (stats.sums_and_risk_stats_device).
        px / forces change with every NLP iterate and come from the host.  staged=False: they travel in the kernel's
        argument block (rato_hopper_slip_host_inputs: no staging buffer, no upload in front of the kernel);
        staged=True: one pinned staging buffer + one asynchronous upload into a device buffer the kernel reads
        (rato_hopper_slip).  Default: by value, except inside a hipGraph capture (where by-value inputs would be frozen
        into the graph) and for more than 128 contacts."""
        px = np.ascontiguousarray(px, dtype=np.float32)
        forces = np.asarray(forces, dtype=np.float32)
        Cn, M = px.shape[0], self._a.shape[1]
        dev = self.device
        capturing = torch.cuda.is_current_stream_capturing()
        if staged is None:
            staged = capturing
        if Cn > MAX_HOST_CONTACTS:
            staged = True
        if not staged:
            fx = np.ascontiguousarray(forces[:, 0])
            fz = np.ascontiguousarray(forces[:, 1])
            hp = lambda x: C.c_void_p(x.ctypes.data)
            entry, pxd, fxd, fzd = self._lib.rato_hopper_slip_host_inputs, hp(px), hp(fx), hp(fz)
        else:
            # ONE pinned staging buffer and one asynchronous upload (three pageable copies cost ~40 us, more than the
            # kernel at M = 5e4)
            st = getattr(self, "_stage", None)
            if st is None or st[0].shape[1] != Cn:
                if capturing:
                    raise RuntimeError("hopper.slip_device: the pinned staging buffer cannot be allocated inside a "
                                       "hipGraph capture; call slip_device(..., staged=True) once before capturing")
                st = (torch.empty((3, Cn), dtype=torch.float32).pin_memory(),
                      torch.empty((3, Cn), dtype=torch.float32, device=dev))
                self._stage = st
            host, devbuf = st
            self._stage_event = getattr(self, "_stage_event", None)
            if self._stage_event is not None and not capturing:   # inside a capture: no host-side waits
                self._stage_event.synchronize()      # the previous upload has left the pinned buffer
            host[0].copy_(torch.from_numpy(px))
            host[1].copy_(torch.from_numpy(np.ascontiguousarray(forces[:, 0])))
            host[2].copy_(torch.from_numpy(np.ascontiguousarray(forces[:, 1])))
            devbuf.copy_(host, non_blocking=True)
            if not capturing:
                if self._stage_event is None:
                    self._stage_event = torch.cuda.Event()
                self._stage_event.record()
            entry, pxd, fxd, fzd = self._lib.rato_hopper_slip, _lib.ptr(devbuf[0]), _lib.ptr(devbuf[1]), _lib.ptr(devbuf[2])
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        Z = e(M) if want_Z else None
        h = e(Cn, M) if want_h else None
        dfz = e(Cn, M) if want_deriv else None
        dpx = e(Cn, M) if want_deriv else None
        lamd = part = None
        if lam is not None:
            lamd = (lam if isinstance(lam, torch.Tensor) else torch.as_tensor(np.asarray(lam), device=dev))
            lamd = lamd.float().contiguous()
            if tuple(lamd.shape) != (Cn, M):
                raise ValueError(f"lam must be [C][M] = ({Cn},{M}), got {tuple(lamd.shape)}")
            part = e(self._lib.rato_hopper_nblocks(M), Cn, 2)
        _lib.check(entry(
            M, Cn, pxd, fxd, fzd, _lib.ptr(self._a), _lib.ptr(self._th),
            _lib.ptr(self._tau), _lib.ptr(lamd), _lib.ptr(Z), _lib.ptr(h), _lib.ptr(dfz), _lib.ptr(dpx),
            _lib.ptr(part), _lib.current_stream()), "rato_hopper_slip")
        hess = stats.sum_partials(part) if (part is not None and reduce) else None
        return {"Z": Z, "h": h, "dh_dfz": dfz, "dh_dpx": dpx, "hess": hess, "part": part}

    def slip_risk_constraints(self, Z):
        """hopper.py:300-367 -> gs (1 + M + M*C + 1,) ['saa'] or (M*C,) ['baseline']."""
        Z = np.asarray(Z, dtype=np.float64)
        _, _, ys, slack_var, t_risk = self.convert_z_to_variables(Z)
        px, forces = self.contact_inputs(Z)
        h = self.slip_device(px, forces, want_Z=False)["h"].t().double().cpu().numpy()   # (M,C)
        M, Cn = h.shape
        if self.method == 'baseline':
            return (h - slack_var).reshape(M * Cn)
        gs = np.zeros(1 + M + M * Cn + 1)
        gs[0] = (M * self.alpha) * t_risk + np.sum(ys)
        gs[1:1 + M] = -ys
        gs[1 + M:1 + M + M * Cn] = (h - t_risk - ys[:, None] - slack_var).reshape(M * Cn)
        return gs

    def slip_partials(self, px, forces):
        """(h, dh/dfz, dh/dpx), each (M,C) — the sample-dependent slices of jac_g."""
        r = self.slip_device(px, forces, want_Z=False, want_deriv=True)
        return tuple(r[k].t().double().cpu().numpy() for k in ("h", "dh_dfz", "dh_dpx"))

    def slip_hessian_sums(self, px, forces, lam):
        """(D1 (C,), D2 (C,)): lambda-weighted d2h/(dpx dfz), d2h/dpx^2 summed over samples."""
        hess = self.slip_device(px, forces, lam=np.asarray(lam).T, want_Z=False, want_h=False,
                                want_deriv=True)["hess"].cpu().numpy()
        return hess[:, 0], hess[:, 1]

    # ---- the reference's own matrices (hopper.py:569, :575-580) ---------------
    def contact_steps(self):
        return np.concatenate([np.arange(0, self.time_jump), np.arange(self.time_land, self.S)])

    def _jacobian_pattern(self, Cn):
        """(indices, indptr) of ``jacrev(slip_risk_constraints)`` with EVERY structural entry present, in the value order
        of rato_hopper_emit_jacobian_values (include/rato_saa.h); built once per (M, C, method)."""
        key = (self._a.shape[1], Cn, self.method)
        pat = getattr(self, "_jac_pattern", None)
        if pat is not None and pat[0] == key:
            return pat[1], pat[2]
        M, S = key[0], self.S
        saa = self.method != 'baseline'
        nX, nU = (S + 1) * n_x, S * n_u
        steps = self.contact_steps()
        r0 = 1 + M if saa else 0
        rows_c = r0 + np.arange(M, dtype=np.int64)[None, :] * Cn + np.arange(Cn, dtype=np.int64)[:, None]     # (C, M)
        counts = np.zeros(self.num_vars, dtype=np.int64)
        for k in (0, 2, 3):
            counts[steps * n_x + k] = M
        counts[nX + steps * n_u + 2] = M
        counts[nX + steps * n_u + 3] = M
        parts = [np.repeat(rows_c[:, None, :], 3, axis=1).reshape(-1), np.repeat(rows_c[:, None, :], 2, axis=1).reshape(-1)]
        rows_i = r0 + np.arange(M, dtype=np.int64)[:, None] * Cn + np.arange(Cn, dtype=np.int64)[None, :]     # (M, C)
        if saa:
            ycols = np.concatenate([np.zeros((M, 1), dtype=np.int64), 1 + np.arange(M, dtype=np.int64)[:, None], rows_i], axis=1)
            parts.append(ycols.reshape(-1))
            counts[nX + nU:nX + nU + M] = 2 + Cn
        parts.append(rows_i.reshape(-1))
        counts[self.num_vars - 2] = M * Cn
        if saa:
            parts.append(np.concatenate([[0], rows_i.reshape(-1)]))
            counts[self.num_vars - 1] = 1 + M * Cn
        indices = np.concatenate(parts).astype(np.int32)
        indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        assert indices.size == int(self._lib.rato_hopper_jacobian_nnz(M, Cn, int(saa))) == indptr[-1]
        self._jac_pattern = (key, indices, indptr)
        return indices, indptr

    def slip_jacobian_device(self, Z, out=None):
        """-> (values [nnz] fp32 device tensor, indices, indptr, shape): ``jacrev(slip_risk_constraints)(Z)`` of the
        reference (hopper.py:569 on the rows of :300-367), rows and columns in its order, as CSC with every structural entry
        present; the values are written on the device (rato_hopper_emit_jacobian_values) from the slip kernel's partials and
        the end-effector chain factors.  ``out``: the value tensor of an earlier call (its constant part is kept)."""
        Z = np.asarray(Z, dtype=np.float64)
        px, forces = self.contact_inputs(Z)
        Cn, M = px.shape[0], self._a.shape[1]
        r = self.slip_device(px, forces, want_Z=False, want_h=False, want_deriv=True)
        Jee, _ = self.contact_chain(Z)
        chain = np.ascontiguousarray(Jee, dtype=np.float32)
        saa = self.method != 'baseline'
        nnz = int(self._lib.rato_hopper_jacobian_nnz(M, Cn, int(saa)))
        fresh = out is None or out.numel() != nnz
        vals = torch.empty(nnz, dtype=torch.float32, device=self.device) if fresh else out
        chain_dev = None
        if Cn > MAX_HOST_CONTACTS:
            chain_dev = torch.as_tensor(chain, device=self.device)
        _lib.check(self._lib.rato_hopper_emit_jacobian_values(
            M, Cn, int(saa), float(self.alpha), _lib.ptr(r["dh_dfz"]), _lib.ptr(r["dh_dpx"]),
            C.c_void_p(chain.ctypes.data), _lib.ptr(chain_dev), int(fresh), _lib.ptr(vals), _lib.current_stream()),
            "rato_hopper_emit_jacobian_values")
        indices, indptr = self._jacobian_pattern(Cn)
        n_rows = (1 + M + M * Cn + 1) if saa else M * Cn
        return vals, indices, indptr, (n_rows, self.num_vars)

    def slip_jacobian(self, Z):
        """The same matrix as a scipy CSC (fp64 values, exact zeros dropped as the reference's ``csc_matrix(dense)`` drops
        them: the sparsity pattern is the reference's)."""
        import scipy.sparse as sp
        vals, indices, indptr, shape = self.slip_jacobian_device(Z)
        data = vals.double().cpu().numpy()
        if self.method != 'baseline':
            data[indptr[-2]] = self._a.shape[1] * self.alpha       # row 0 of the t_risk column: M alpha in fp64 (a constant)
        A = sp.csc_matrix((data, indices, indptr), shape=shape)
        A.eliminate_zeros()
        A.sort_indices()
        return A

    slip_jacobian_rows = slip_jacobian

    def slip_hessian(self, Z, lam):
        """``hessian(lambda . slip_risk_constraints)(Z)`` (hopper.py:575-580; ``lam`` (M, C): the multipliers of the
        per-sample rows -- every other row is linear) as a scipy CSC (num_vars x num_vars).  The three sample sums per contact
        come from ONE launch (rato_hopper_slip_hessian); the 15 entries per contact are placed on the host."""
        import scipy.sparse as sp
        Z = np.asarray(Z, dtype=np.float64)
        px, forces = self.contact_inputs(Z)
        D = self.slip_hessian_sums3(px, forces, lam)
        Jee, Hee = self.contact_chain(Z)
        steps = self.contact_steps()
        nX = (self.S + 1) * n_x
        xi = steps[:, None] * n_x + np.array([0, 2, 3])[None, :]                                   # (C, 3)
        blk = D[:, 1, None, None] * Jee[:, :, None] * Jee[:, None, :] + D[:, 2, None, None] * Hee    # (C, 3, 3)
        mixed = D[:, 0, None] * Jee                                                                  # (C, 3)
        fz = (nX + steps * n_u + 3)[:, None]
        I = np.concatenate([np.repeat(xi, 3, axis=1).reshape(-1), xi.reshape(-1), np.repeat(fz, 3, axis=1).reshape(-1)])
        J = np.concatenate([np.tile(xi, (1, 3)).reshape(-1), np.repeat(fz, 3, axis=1).reshape(-1), xi.reshape(-1)])
        V = np.concatenate([blk.reshape(-1), mixed.reshape(-1), mixed.reshape(-1)])
        H = sp.coo_matrix((V, (I, J)), shape=(self.num_vars, self.num_vars)).tocsc()
        H.eliminate_zeros()
        H.sort_indices()
        return H

    def slip_hessian_sums3(self, px, forces, lam):
        """(C, 3): per contact the lambda-weighted sums over the samples of d2h/(dpx dfz), d2h/dpx^2 and dh/dpx."""
        px = np.ascontiguousarray(px, dtype=np.float32)
        forces = np.asarray(forces, dtype=np.float32)
        Cn, M = px.shape[0], self._a.shape[1]
        lamd = torch.as_tensor(np.ascontiguousarray(np.asarray(lam, dtype=np.float64).T), device=self.device).float().contiguous()
        if tuple(lamd.shape) != (Cn, M):
            raise ValueError(f"lam must be (M, C) = ({M},{Cn})")
        part = torch.empty((self._lib.rato_hopper_nblocks(M), Cn, 3), dtype=torch.float32, device=self.device)
        fx, fz = np.ascontiguousarray(forces[:, 0]), np.ascontiguousarray(forces[:, 1])
        if Cn <= MAX_HOST_CONTACTS:
            args, host = (C.c_void_p(px.ctypes.data), C.c_void_p(fx.ctypes.data), C.c_void_p(fz.ctypes.data)), 1
        else:
            dev = [torch.as_tensor(v, device=self.device) for v in (px, fx, fz)]
            args, host = tuple(_lib.ptr(v) for v in dev), 0
        _lib.check(self._lib.rato_hopper_slip_hessian(
            M, Cn, *args, host, _lib.ptr(self._a), _lib.ptr(self._th), _lib.ptr(self._tau), _lib.ptr(lamd), None, None,
            None, None, _lib.ptr(part), _lib.current_stream()), "rato_hopper_slip_hessian")
        return stats.sum_partials(part).cpu().numpy().reshape(Cn, 3)

    # ---- Monte-Carlo validation (hopper.py:901-958) ------------------------
    def no_slip_constraints_verification(self, px, forces):
        Zh = self.slip_device(px, forces, want_h=False)["Z"].double().cpu().numpy()
        return Zh <= 1e-6, Zh

    def monte_carlo_statistics(self, px, forces, alpha=None):
        Z = self.slip_device(px, forces, want_h=False)["Z"]
        return stats.risk_stats(Z, self.alpha if alpha is None else alpha)

    avar = staticmethod(stats.monte_carlo_avar)

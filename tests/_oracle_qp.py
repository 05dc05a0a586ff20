"""Test helper: the reference's Model API (L3/L4) on top of the fp64 oracle, so that the sparse
assembler, the host QP and the SCP driver can be checked on CPU and the device path can be compared
iterate by iterate."""
import numpy as np

from riskaversetrajopt_amd import assemble, qp


def pack_pairs(dense, S, n_u, n_g):
    """dense (..., S rows t, n_u*S cols) -> packed (n_pairs, n_g, ...) over pairs s < t."""
    out = []
    for t in range(1, S):
        for s in range(t):
            out.append(np.stack([dense[..., t, s * n_u + g] for g in range(n_g)], axis=0))
    return np.stack(out, axis=0) if out else np.zeros((0, n_g) + dense.shape[:-2])


class DroneOracleQP:
    """oracle.drone.Model + sparse assembly + host QP, same method names as the reference."""
    MULT, SLACK = 0.01, 10000.0

    def __init__(self, om):
        from oracle import drone as od
        self.o, self.od = om, od
        self.S, self.M, self.dt, self.alpha, self.method = om.S, om.M, om.dt, om.alpha, om.method

    def initial_guess_us_mat(self):
        return self.o.initial_guess_us_mat()

    def packed(self, us):
        fdu, flo, _, gdu, gup = self.o.get_all_constraints_coeffs(us)
        G = pack_pairs(np.transpose(gdu, (1, 0, 2, 3)), self.S, 3, 2)        # (n_pairs, 2, n_obs, M)
        return fdu.mean(0), flo.mean(0), G, np.transpose(gup, (1, 2, 0))

    def get_constraints_coeffs(self, us, scp_iter):
        fdu, frhs, G, gup = self.packed(us)
        relax = ('scale', 6, 1e-7, -0.1, 0.1) if scp_iter < 2 else None
        return assemble.saa_constraints(fdu, frhs, G, gup, n_u=3, S=self.S, M=self.M, alpha=self.alpha,
                                        method=self.method, kappa=self.MULT,
                                        baseline_pad=(1e-3 if self.method == 'baseline' else 0.0),
                                        u_min=-self.od.u_max, u_max=self.od.u_max, relax=relax)

    def get_objective_coeffs(self):
        return assemble.objective(3, self.S, self.M, self.dt, self.od.R, self.SLACK)

    def define_problem(self, us, verbose=False):
        self.P, self.q = self.get_objective_coeffs()
        self.A, self.l, self.u = self.get_constraints_coeffs(us, 2)
        self.osqp_prob = qp.OSQP()
        self.osqp_prob.setup(self.P, self.q, self.A, self.l, self.u, eps_abs=self.od.OSQP_TOL,
                             eps_rel=self.od.OSQP_TOL, warm_start=True, polish=True)
        return True

    def update_problem(self, us, scp_iter=0, verbose=False):
        self.A, self.l, self.u = self.get_constraints_coeffs(us, scp_iter)
        self.osqp_prob.update(l=self.l, u=self.u)
        self.osqp_prob.update(Ax=self.A.data)
        return True

    def solve(self, verbose=False):
        self.res = self.osqp_prob.solve()
        x = self.res.x
        return self.o.convert_us_vec_to_us_mat(x[:3 * self.S]), x[-1]


class DrivingOracleQP:
    SLACK = 1000.0

    def __init__(self, om):
        from oracle import driving as ocar
        self.o, self.oc = om, ocar
        self.S, self.M, self.dt, self.alpha, self.method = om.S, om.M, om.dt, om.alpha, om.method

    def initial_guess_us_mat(self):
        return self.o.initial_guess_us_mat()

    def get_constraints_coeffs(self, us, scp_iter):
        fdu, flo, _, gdu, gup = self.o.get_all_constraints_coeffs(us)
        G = pack_pairs(gdu, self.S, 2, 2)[:, :, None, :]                     # (n_pairs, 2, 1, M)
        relax = ('zero', 8) if scp_iter < 1 else None
        return assemble.saa_constraints(fdu[0], flo[0], G, gup.T[None], n_u=2, S=self.S, M=self.M,
                                        alpha=self.alpha, method=self.method, kappa=1.0, baseline_pad=0.0,
                                        u_min=-self.oc.u_max, u_max=self.oc.u_max, relax=relax)

    def get_objective_coeffs(self):
        return assemble.objective(2, self.S, self.M, self.dt, self.oc.R, self.SLACK)

    def define_problem(self, us, scp_iter=0, verbose=False):
        self.P, self.q = self.get_objective_coeffs()
        self.A, self.l, self.u = self.get_constraints_coeffs(us, scp_iter)
        if scp_iter in (0, 1):
            self.osqp_prob = qp.OSQP()
            self.osqp_prob.setup(self.P, self.q, self.A, self.l, self.u, eps_abs=self.oc.OSQP_TOL,
                                 eps_rel=self.oc.OSQP_TOL, warm_start=True, polish=True)
        else:
            self.osqp_prob.update(l=self.l, u=self.u)
            self.osqp_prob.update(Ax=self.A.data)
        return True

    def solve(self, verbose=False):
        self.res = self.osqp_prob.solve()
        x = self.res.x
        return self.o.convert_us_vec_to_us_mat(x[:2 * self.S]), x[-1]

import os, sys, numpy as np
sys.argv = [sys.argv[0], "3000", "20", "3", "0.02"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'proto_newton.py')).read().split("us = mdl.initial_guess_us_mat()")[0])
us = mdl.initial_guess_us_mat()
for k in range(3):
    us, _, info = mdl.solve_reduced(us, k, tol=1e-10)
fdu, frhs, gdu, gup = mdl.linearization(us)
G = gdu.reshape(M, -1, nU); gupf = gup.reshape(M, -1)
uk = np.asarray(us, dtype=np.float64).reshape(-1)
base = -(gupf - G @ uk)
x = np.zeros(nU)
for hf in (0.005, 0.02, 0.05, 0.1):
    HFRAC = hf
    phi, t, g, H = oracle(G, base, x)
    rng = np.random.RandomState(1)
    out = []
    for trial in range(3):
        d = rng.randn(nU); d[2::3] = 0; d /= np.linalg.norm(d)
        for eps in (3e-2, 1e-2):
            _, _, g2, _ = oracle(G, base, x + eps * d, want_H=False)
            _, _, g1, _ = oracle(G, base, x - eps * d, want_H=False)
            fd = (g2 - g1) / (2 * eps)
            out.append((float(d @ fd), float(d @ H @ d), float(np.linalg.norm(fd - H @ d) / np.linalg.norm(fd))))
    print("HFRAC", hf, "dHd fd/model, rel err:", " | ".join("%.3f/%.3f %.2f" % o for o in out))
print("eig H top:", np.linalg.eigvalsh(H)[-6:])

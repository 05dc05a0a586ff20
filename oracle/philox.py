"""Oracle (test infrastructure) — Philox4x32-10 and the device sampler's transforms, NumPy.

The reference has no counter-based generator (it draws with NumPy's global MT19937 stream:
drone_utils.py:61-93, driving.py:84-120, hopper.py:70-74); the device-side sampler is this build's own
component.  Its published algorithm is Philox4x32 with 10 rounds — J. K. Salmon,
M. A. Moraes, R. O. Dror, D. E. Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 — restated
here in NumPy integer arithmetic and pinned by the known-answer vectors of Random123's ``kat_vectors`` for
``philox4x32 10`` (tests/test_philox_oracle.py).  The uniform / Box–Muller transforms follow
``riskaversetrajopt_amd/csrc/philox.h`` and are evaluated in fp64.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_DW, STREAM_MASS, STREAM_RADII, STREAM_OMEGA, STREAM_X0, STREAM_FIELD, STREAM_USER = 1, 2, 3, 4, 5, 6, 16


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counter words (arrays broadcast together) and key words (python ints) -> 4 arrays of uint32."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def philox_at(seed, stream, t, m):
    """counter = (m low, m high, t, stream), key = (seed low, seed high): csrc/philox.h philox_at."""
    m = np.asarray(m, dtype=np.uint64)
    return philox4x32_10(m & MASK, m >> np.uint64(32), np.asarray(t, dtype=np.uint64), np.uint64(stream),
                         int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)


def u01(r):
    """24-bit uniform in (0, 1): ((r >> 8) + 0.5) * 2^-24."""
    return ((np.asarray(r, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24


def box_muller(ra, rb):
    """(rho cos 2 pi phi, rho sin 2 pi phi), rho = sqrt(-2 ln(((ra >> 8) + 1) 2^-24)), phi = (rb >> 8) 2^-24."""
    u = ((np.asarray(ra, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    phi = (np.asarray(rb, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    rho = np.sqrt(-2.0 * np.log(u))
    return rho * np.cos(2.0 * np.pi * phi), rho * np.sin(2.0 * np.pi * phi)


def normals(seed, stream, T, M, C):
    """[T][C][M] standard normals of the generic fill (C <= 4)."""
    t, m = np.meshgrid(np.arange(T), np.arange(M), indexing="ij")
    r = philox_at(seed, stream, t, m)
    n0, n1 = box_muller(r[0], r[1])
    n2, n3 = box_muller(r[2], r[3])
    return np.stack([n0, n1, n2, n3][:C], axis=1)


def uniforms(seed, stream, T, M, C):
    t, m = np.meshgrid(np.arange(T), np.arange(M), indexing="ij")
    r = philox_at(seed, stream, t, m)
    return np.stack([u01(x) for x in r][:C], axis=1)


def drone_sample(seed, M, S, sampler_dt, mass_nom=32.0, mass_delta=3.0, obs_radii=(0.3, 0.2, 0.2), delta=0.025):
    """rato_drone_sample in the REFERENCE's layouts: DWs (M,S,6) (position rows zero: only rows 3..5 enter sigma),
    masses (M,), obs_Qs (M,3,3,3) diagonal with the z entry left 0 (it never enters obs_Q[:2,:2])."""
    dW = np.sqrt(sampler_dt) * normals(seed, STREAM_DW, S, M, 3)                 # [S][3][M]
    DWs = np.zeros((M, S, 6))
    DWs[:, :, 3:6] = np.transpose(dW, (2, 0, 1))
    masses = mass_nom + mass_delta * (2.0 * uniforms(seed, STREAM_MASS, 1, M, 1)[0, 0] - 1.0)
    u = uniforms(seed, STREAM_RADII, 3, M, 2)                                    # [obstacle][x, y][M]
    obs_Qs = np.zeros((M, 3, 3, 3))
    for j in range(3):
        for d in range(2):
            obs_Qs[:, j, d, d] = 1.0 / (obs_radii[j] + delta * (2.0 * u[j, d] - 1.0)) ** 2
    return DWs, masses, obs_Qs


def car_sample(seed, M, S, sampler_dt, state_init, x0_std, ws=(0.1, 0.075), wr=(0.05, 0.045)):
    """rato_car_sample in the reference's layouts: states_init (M,8), omegas_speed, omegas_repulsive (M,), DWs (M,S,8)."""
    dW = np.sqrt(sampler_dt) * normals(seed, STREAM_DW, S, M, 2)
    DWs = np.zeros((M, S, 8))
    DWs[:, :, 6:8] = np.transpose(dW, (2, 0, 1))
    u = uniforms(seed, STREAM_OMEGA, 1, M, 2)[0]
    omegas_speed = ws[0] + ws[1] * (2.0 * u[0] - 1.0)
    omegas_repulsive = wr[0] + wr[1] * (2.0 * u[1] - 1.0)
    n = normals(seed, STREAM_X0, 1, M, 4)[0]                                     # [4][M]
    states_init = np.repeat(np.asarray(state_init, dtype=np.float64)[None], M, axis=0)
    states_init[:, 4:] += (np.asarray(x0_std)[:, None] * n).T
    return states_init, omegas_speed, omegas_repulsive, DWs


def hopper_sample(seed, M, K=30):
    u = uniforms(seed, STREAM_FIELD, K, M, 3)                                    # [K][3][M]
    return (0.025 * np.sqrt(2.0 / K) * u[:, 0].T, np.pi * u[:, 1].T, 2.0 * np.pi * u[:, 2].T)

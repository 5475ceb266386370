"""Synthetic S3DIS-like inputs (no dataset is available offline): seeded rooms, tiles' superpoints, weights.
Follows the generator described in SURVEY.md section 8(d)."""
import numpy as np


def make_room(seed, density=5000.0):
    """Axis-aligned room: floor, ceiling, 4 walls and 3-8 boxes sampled at `density` points/m^2 with 2 mm normal
    jitter; u8 colours per surface (+noise); labels 0..12 per surface.  Returns xyz f32 [N,3], rgb u8 [N,3],
    label i32 [N]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W, D, H = rng.uniform(4, 10), rng.uniform(3, 8), 3.0
    parts = []

    def plane(origin, u, v, label):
        area = np.linalg.norm(u) * np.linalg.norm(v)
        n = max(int(area * density), 16)
        a, b = rng.random(n), rng.random(n)
        p = origin[None] + a[:, None] * u[None] + b[:, None] * v[None]
        nrm = np.cross(u, v); nrm /= np.linalg.norm(nrm)
        p = p + rng.normal(0, 0.002, n)[:, None] * nrm[None]
        base = rng.integers(40, 216, 3)
        col = np.clip(base[None] + rng.normal(0, 12, (n, 3)), 0, 255).astype(np.uint8)
        parts.append((p.astype(np.float32), col, np.full(n, label, np.int32)))

    o = np.zeros(3)
    ex, ey, ez = np.array([W, 0, 0.]), np.array([0, D, 0.]), np.array([0, 0, H])
    plane(o, ex, ey, 1)                      # floor
    plane(o + ez, ex, ey, 0)                 # ceiling
    plane(o, ex, ez, 2); plane(o + ey, ex, ez, 2); plane(o, ey, ez, 2); plane(o + ex, ey, ez, 2)   # walls
    for _ in range(int(rng.integers(3, 9))):    # furniture boxes
        sz = rng.uniform(0.3, 1.5, 3); sz[2] = rng.uniform(0.4, 1.2)
        c = np.array([rng.uniform(0, W - sz[0]), rng.uniform(0, D - sz[1]), 0.0])
        lab = int(rng.integers(3, 13))
        bx, by, bz = np.array([sz[0], 0, 0.]), np.array([0, sz[1], 0.]), np.array([0, 0, sz[2]])
        plane(c + bz, bx, by, lab); plane(c, bx, bz, lab); plane(c + by, bx, bz, lab); plane(c, by, bz, lab); plane(c + bx, by, bz, lab)
    xyz = np.concatenate([p[0] for p in parts]); rgb = np.concatenate([p[1] for p in parts]); lab = np.concatenate([p[2] for p in parts])
    perm = rng.permutation(len(xyz))
    return xyz[perm], rgb[perm], lab[perm]


def superpoints_from_tile(xyz, cell=0.3):
    """Stand-in for the cut-pursuit partition (out of scope): connected-looking blobs = occupied cells of a coarse
    grid.  Returns CSR (offsets int32 [S+1], points int32 [N])."""
    k = np.floor((xyz - xyz.min(0)) / cell).astype(np.int64)
    key = k[:, 0] + 4096 * (k[:, 1] + 4096 * k[:, 2])
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    off = np.concatenate([heads, [len(ks)]]).astype(np.int32)
    return off, order.astype(np.int32)

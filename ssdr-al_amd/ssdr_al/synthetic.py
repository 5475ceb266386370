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


# ---- synthetic network weights ("random-init weights of that architecture") -------------------------------------------
def layer_specs(d_out=(16, 64, 128, 256, 512), num_classes=13, in_dim=6):
    """Ordered list of (name, in, out, bias, bn, act, transposed) — also the layer order of the C ABI."""
    specs = [("fc0", in_dim, 8, True, True, True, False)]
    d_in = 8
    for i, d in enumerate(d_out):
        h = d // 2
        p = "Encoder_layer_%d" % i
        specs += [(p + "mlp1", d_in, h, True, True, True, False),
                  (p + "LFAmlp1", 10, h, True, True, True, False),
                  (p + "LFAatt_pooling_1fc", d, d, False, False, False, False),
                  (p + "LFAatt_pooling_1mlp", d, h, True, True, True, False),
                  (p + "LFAmlp2", h, h, True, True, True, False),
                  (p + "LFAatt_pooling_2fc", d, d, False, False, False, False),
                  (p + "LFAatt_pooling_2mlp", d, d, True, True, True, False),
                  (p + "mlp2", d, 2 * d, True, True, False, False),
                  (p + "shortcut", d_in, 2 * d, True, True, False, False)]
        d_in = 2 * d
    specs.append(("decoder_0", d_in, d_in, True, True, True, False))
    enc_ch = [2 * d_out[0]] + [2 * d for d in d_out]        # f_encoder_list channels (RandLANet.py:149-157)
    feat = d_in
    for j in range(len(d_out)):
        skip = enc_ch[-j - 2]
        specs.append(("Decoder_layer_%d" % j, skip + feat, skip, True, True, True, True))
        feat = skip
    specs += [("fc1", feat, 64, True, True, True, False), ("fc2", 64, 32, True, True, True, False),
              ("fc", 32, num_classes, True, False, False, False)]
    return specs


def init_weights(seed=0, d_out=(16, 64, 128, 256, 512), num_classes=13, in_dim=6, trained_like=True):
    """Random-init weights following helper_tf_util.py:43-48 (round(truncated_normal(std=sqrt(2/shape[-1]))*1000)/1000,
    bias 0) and Glorot-uniform for tf.layers.dense.  trained_like=True also randomises the BN statistics and biases
    (a fresh TF graph has gamma=1, beta=0, mean=0, var=1, which would leave the BN fold untested)."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, cin, cout, bias, bn, act, transposed in layer_specs(d_out, num_classes, in_dim):
        shape = (cout, cin) if transposed else (cin, cout)
        if name == "fc0" or name.endswith("fc") and "att_pooling" in name:
            lim = np.sqrt(6.0 / (cin + cout))
            w = rng.uniform(-lim, lim, shape)
        else:
            std = np.sqrt(2.0 / shape[-1])
            w = np.clip(rng.normal(0, std, shape), -2 * std, 2 * std)
            w = np.round(w * 1000) / 1000
        ent = {"W": w.astype(np.float32), "b": None, "bn": None, "act": act, "transposed": transposed}
        if bias:
            ent["b"] = (rng.normal(0, 0.05, cout) if trained_like else np.zeros(cout)).astype(np.float32)
        if bn:
            if trained_like:
                ent["bn"] = tuple(a.astype(np.float32) for a in (rng.uniform(0.7, 1.3, cout), rng.normal(0, 0.1, cout),
                                                                  rng.normal(0, 0.2, cout), rng.uniform(0.5, 1.5, cout)))
            else:
                ent["bn"] = (np.ones(cout, np.float32), np.zeros(cout, np.float32), np.zeros(cout, np.float32), np.ones(cout, np.float32))
        W[name] = ent
    return W

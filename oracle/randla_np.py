"""TEST INFRASTRUCTURE ONLY — NumPy restatement of RandLA-Net *inference* as the reference defines it.

PARITY UNPINNED: the reference network needs TensorFlow 1.x (tf.layers, tf.batch_gather, tf.contrib;
version not pinned anywhere in the reference), which is absent from this image and cannot be installed, and the
reference holds no golden output for the network.  This file follows the op definitions line by line
(/root/reference/SSDR_AL_s3dis/RandLANet.py:140-180, 505-585 and helper_tf_util.py:111-166, 169-246) and is
cross-checked only against an independent torch-CPU formulation (tests/test_randla.py).

Weights: dict name -> dict(W [in,out], b [out] or None, bn (gamma,beta,mean,var) or None, act bool), names are
the reference's variable scopes under 'layers/'.  conv2d_transpose kernels are stored [out,in] as in the
reference (helper_tf_util.py:207-208) and applied as x @ W.T.
"""
import numpy as np

BN_EPS = 1e-6           # helper_tf_util.py:162 / RandLANet.py:145
LRELU = 0.2             # helper_tf_util.py:165, tf.nn.leaky_relu default


# The oracle's OWN layer table (written from RandLANet.py:140-180, 505-585 / SURVEY appendix B, independently of the product's
# ssdr_al/synthetic.py, which bench.py uses): a wrong BN / activation flag in one of the two tables shows up as a parity failure
# (tests/test_randla.py::test_layer_tables_agree compares them field by field).
def layer_specs(d_out=(16, 64, 128, 256, 512), num_classes=13, in_dim=6):
    """[(scope, in, out, has_bias, has_bn, has_act, transposed_kernel)] in graph order."""
    rows = []

    def conv(scope, cin, cout, bn=True, act=True, bias=True, transposed=False):
        rows.append((scope, cin, cout, bias, bn, act, transposed))

    conv("fc0", in_dim, 8)                                              # tf.layers.dense + BN + lrelu (:144-146)
    width = 8
    skips = []
    for level, d in enumerate(d_out):
        scope = "Encoder_layer_%d" % level
        conv(scope + "mlp1", width, d // 2)                            # dilated_res_block :506
        conv(scope + "LFAmlp1", 10, d // 2)                            # building_block :518
        conv(scope + "LFAatt_pooling_1fc", d, d, bn=False, act=False, bias=False)     # att_pooling :578 (tf.layers.dense, use_bias=False)
        conv(scope + "LFAatt_pooling_1mlp", d, d // 2)                 # :583
        conv(scope + "LFAmlp2", d // 2, d // 2)                        # :523
        conv(scope + "LFAatt_pooling_2fc", d, d, bn=False, act=False, bias=False)
        conv(scope + "LFAatt_pooling_2mlp", d, d)
        conv(scope + "mlp2", d, 2 * d, act=False)                      # :509, activation_fn=None
        conv(scope + "shortcut", width, 2 * d, act=False)              # :510-511
        if level == 0:
            skips.append(2 * d)                                        # f_encoder_list starts with the un-sampled level-0 output (:152-153)
        width = 2 * d
        skips.append(width)
    conv("decoder_0", width, width)                                    # :159-161
    for j in range(len(d_out)):
        skip = skips[-j - 2]
        conv("Decoder_layer_%d" % j, skip + width, skip, transposed=True)      # conv2d_transpose over concat[skip, interpolated] (:165-172)
        width = skip
    conv("fc1", width, 64)
    conv("fc2", 64, 32)
    conv("fc", 32, num_classes, bn=False, act=False)                   # :176-178, activation_fn=None
    return rows


def init_weights(seed=0, d_out=(16, 64, 128, 256, 512), num_classes=13, in_dim=6, trained_like=True):
    """Random weights by the reference's initialisers: conv kernels round(truncated_normal(std = sqrt(2 / shape[-1])) * 1000) / 1000 with
    zero bias (helper_tf_util.py:43-48, :158-159), tf.layers.dense Glorot-uniform.  trained_like=True also draws non-trivial biases and
    BN statistics (a fresh graph has gamma 1, beta 0, mean 0, var 1, which would leave the BN fold untested).  Draw order == the
    product generator's, so the same seed gives the same network in both."""
    rng = np.random.default_rng(seed)
    W = {}
    for scope, cin, cout, has_bias, has_bn, has_act, transposed in layer_specs(d_out, num_classes, in_dim):
        shape = (cout, cin) if transposed else (cin, cout)
        if scope == "fc0" or scope.endswith("fc") and "att_pooling" in scope:
            lim = np.sqrt(6.0 / (cin + cout))
            w = rng.uniform(-lim, lim, shape)
        else:
            std = np.sqrt(2.0 / shape[-1])
            w = np.round(np.clip(rng.normal(0, std, shape), -2 * std, 2 * std) * 1000) / 1000
        ent = {"W": w.astype(np.float32), "b": None, "bn": None, "act": has_act, "transposed": transposed}
        if has_bias:
            ent["b"] = (rng.normal(0, 0.05, cout) if trained_like else np.zeros(cout)).astype(np.float32)
        if has_bn:
            if trained_like:
                ent["bn"] = tuple(a.astype(np.float32) for a in (rng.uniform(0.7, 1.3, cout), rng.normal(0, 0.1, cout),
                                                                  rng.normal(0, 0.2, cout), rng.uniform(0.5, 1.5, cout)))
            else:
                ent["bn"] = (np.ones(cout, np.float32), np.zeros(cout, np.float32), np.zeros(cout, np.float32), np.ones(cout, np.float32))
        W[scope] = ent
    return W


def fold_bn(ent, dtype=np.float32):
    """Inference-time fold: returns W [in,out] and b [out] with BN absorbed."""
    w = ent["W"].astype(np.float64)
    if ent["transposed"]:
        w = w.T
    b = np.zeros(w.shape[1]) if ent["b"] is None else ent["b"].astype(np.float64)
    if ent["bn"] is not None:
        g, beta, mu, var = [a.astype(np.float64) for a in ent["bn"]]
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[None, :]
        b = (b - mu) * s + beta
    return w.astype(dtype), b.astype(dtype)


def _lrelu(x):
    return np.where(x > 0, x, x * x.dtype.type(LRELU))


def _conv(x, ent):
    """helper_tf_util.conv2d / conv2d_transpose / tf.layers.dense on the channel axis: x [..., in] -> [..., out]."""
    w = ent["W"].astype(x.dtype)
    y = x @ (w.T if ent["transposed"] else w)
    if ent["b"] is not None:
        y = y + ent["b"].astype(x.dtype)
    if ent["bn"] is not None:
        g, beta, mu, var = [a.astype(x.dtype) for a in ent["bn"]]
        y = (y - mu) / np.sqrt(var + x.dtype.type(BN_EPS)) * g + beta
    if ent["act"]:
        y = _lrelu(y)
    return y


def _gather(pc, idx):
    """gather_neighbour (RandLANet.py:561-570): pc [B,N,d], idx [B,M,K] -> [B,M,K,d]."""
    B = pc.shape[0]
    return np.stack([pc[b][idx[b]] for b in range(B)])


def _att_pooling(fset, W, name):
    """RandLANet.py:572-585: dense d->d (no bias), softmax over the K axis, weighted sum, then conv."""
    act = fset @ W[name + "fc"]["W"].astype(fset.dtype)
    act = act - act.max(axis=2, keepdims=True)
    e = np.exp(act)
    scores = e / e.sum(axis=2, keepdims=True)
    agg = (fset * scores).sum(axis=2)
    return _conv(agg, W[name + "mlp"])


def forward(W, features, xyz, neigh_idx, sub_idx, interp_idx, dtype=np.float32, return_all=False):
    """features [B,N,6]; xyz: list of [B,N_i,3]; neigh_idx [B,N_i,K]; sub_idx [B,N_{i+1},K]; interp_idx [B,N_i,1].
    Returns probs [B*N,C] (softmax, RandLANet.py:84) and last_second_features [B*N,32] (RandLANet.py:45,175)."""
    L = len(neigh_idx)
    f = _conv(features.astype(dtype), W["fc0"])                 # dense + BN + lrelu (:144-146)
    enc = []
    trace = {}
    for i in range(L):
        p = "Encoder_layer_%d" % i
        x = xyz[i].astype(dtype)
        nb = neigh_idx[i]
        f_pc = _conv(f, W[p + "mlp1"])                          # dilated_res_block :506
        # building_block :514-527
        nxyz = _gather(x, nb)
        tile = np.broadcast_to(x[:, :, None, :], nxyz.shape)
        rel = tile - nxyz
        dis = np.sqrt((rel * rel).sum(-1, keepdims=True))
        f_xyz = _conv(np.concatenate([dis, rel, tile, nxyz], -1), W[p + "LFAmlp1"])   # :529-535, :518
        f_cat = np.concatenate([_gather(f_pc, nb), f_xyz], -1)
        agg = _att_pooling(f_cat, W, p + "LFAatt_pooling_1")
        f_xyz = _conv(f_xyz, W[p + "LFAmlp2"])
        f_cat = np.concatenate([_gather(agg, nb), f_xyz], -1)
        agg = _att_pooling(f_cat, W, p + "LFAatt_pooling_2")
        out = _lrelu(_conv(agg, W[p + "mlp2"]) + _conv(f, W[p + "shortcut"]))          # :508-512
        samp = _gather(out, sub_idx[i]).max(axis=2)            # random_sample :537-548
        if i == 0:
            enc.append(out)
        enc.append(samp)
        trace["enc%d" % i] = out
        f = samp
    f = _conv(enc[-1], W["decoder_0"])
    for j in range(L):
        idx = interp_idx[-j - 1][..., 0]
        interp = np.stack([f[b][idx[b]] for b in range(f.shape[0])])                   # nearest_interpolation :550-559
        f = _conv(np.concatenate([enc[-j - 2], interp], -1), W["Decoder_layer_%d" % j])
    f1 = _conv(f, W["fc1"])
    f2 = _conv(f1, W["fc2"])
    logits = _conv(f2, W["fc"])                                # dropout is the identity at inference
    logits = logits.reshape(-1, logits.shape[-1])
    z = logits - logits.max(-1, keepdims=True)
    e = np.exp(z)
    probs = e / e.sum(-1, keepdims=True)
    feat = f2.reshape(-1, 32)
    if return_all:
        return probs, feat, trace
    return probs, feat


def build_pyramid(xyz0, ratios, knn_batch, K=16):
    """tf_map (s3dis_dataset.py:156-183) with a caller-supplied knn_batch(support, query, k) -> int [B,Nq,k]."""
    xyz, neigh, sub, interp = [], [], [], []
    cur = xyz0
    for r in ratios:
        nb = knn_batch(cur, cur, K).astype(np.int32)
        nxt = cur[:, : cur.shape[1] // r]
        xyz.append(cur)
        neigh.append(nb)
        sub.append(nb[:, : cur.shape[1] // r])
        interp.append(knn_batch(nxt, cur, 1).astype(np.int32))
        cur = nxt
    return xyz, neigh, sub, interp

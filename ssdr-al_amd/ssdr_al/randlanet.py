"""Host-side mirror of the inference half of the reference's ``RandLANet.Network``
(/root/reference/SSDR_AL_s3dis/RandLANet.py:140-180, 505-585): weights are addressed by the reference's variable
scopes ('fc0', 'Encoder_layer_0mlp1', ..., 'Decoder_layer_4', 'fc'), batch-norm is folded at load, and
``Network.infer`` returns what ``sess.run([prob_logits, last_second_features])`` returns
(S3/sampler2.py:598, :327).  All arithmetic runs in libssdr_al.so."""
import ctypes as C

import numpy as np

from . import _lib
from .helper_tool import ConfigS3DIS

BN_EPS = 1e-6


def _fold(ent):
    """conv + bias + BN(gamma,beta,mean,var, eps 1e-6) -> (W [in,out], b [out]) (helper_tf_util.py:158-163)."""
    w = np.asarray(ent["W"], np.float64)
    if ent.get("transposed"):          # conv2d_transpose kernels are [out,in] (helper_tf_util.py:207-208)
        w = w.T
    b = np.zeros(w.shape[1]) if ent.get("b") is None else np.asarray(ent["b"], np.float64)
    if ent.get("bn") is not None:
        g, beta, mu, var = [np.asarray(a, np.float64) for a in ent["bn"]]
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[None, :]
        b = (b - mu) * s + beta
    return w, b


class Network:
    """``Network(config).load(weights)`` then ``infer(...)``; config carries k_n, num_layers, d_out,
    sub_sampling_ratio, num_classes as the reference's Config classes do (helper_tool.py:46-117)."""

    def __init__(self, config=ConfigS3DIS, in_dim=6):
        self.config = config
        self.in_dim = in_dim
        self._h = C.c_void_p()
        d = np.asarray(config.d_out, np.int32)
        _lib.check(_lib.lib().ssdr_randla_create(config.num_layers, _lib.ptr(d), config.k_n, config.num_classes, in_dim,
                                                 C.byref(self._h)))
        self._lib = _lib.lib()
        self.precision = "f32"

    def __del__(self):
        try:
            if self._h:
                self._lib.ssdr_randla_destroy(self._h)
                self._h = None
        except Exception:
            pass

    PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}

    def set_precision(self, mode):
        """Arithmetic of the matrix products: "f32" (exact, default), "bf16x3" (split bf16, fp32 accumulate; within the
        1e-3 tolerance of the fp32 path) or "bf16" (BASELINE configuration 3)."""
        _lib.check(_lib.lib().ssdr_randla_set_precision(self._h, self.PRECISIONS[mode]))
        self.precision = mode
        return self

    def set_formulation(self, tiles32=True):
        """bf16 modes: 32 x 32 MFMA tiles with the softmax inside the lane (default) or the 16 x 16-tile kernels (A/B timing, cross-check)."""
        _lib.check(_lib.lib().ssdr_randla_set_formulation(self._h, 1 if tiles32 else 0))
        return self

    def layer_table(self, weights):
        """Reference-named weights -> the ABI's ordered (W, b) list (csrc/randla_model.hip header)."""
        L = self.config.num_layers
        out = [_fold(weights["fc0"])]
        for i in range(L):
            p = "Encoder_layer_%d" % i
            out.append(_fold(weights[p + "mlp1"]))
            out.append(_fold(weights[p + "LFAmlp1"]))
            out.append((np.asarray(weights[p + "LFAatt_pooling_1fc"]["W"], np.float64), None))
            out.append(_fold(weights[p + "LFAatt_pooling_1mlp"]))
            out.append(_fold(weights[p + "LFAmlp2"]))
            out.append((np.asarray(weights[p + "LFAatt_pooling_2fc"]["W"], np.float64), None))
            out.append(_fold(weights[p + "LFAatt_pooling_2mlp"]))
            w2, b2 = _fold(weights[p + "mlp2"])
            ws, bs = _fold(weights[p + "shortcut"])
            out.append((np.concatenate([w2, ws], 0), b2 + bs))      # lrelu(mlp2(agg) + shortcut(feature)), :508-512
        out.append(_fold(weights["decoder_0"]))
        for j in range(L):
            out.append(_fold(weights["Decoder_layer_%d" % j]))      # rows already ordered [skip | interp] (:167)
        out += [_fold(weights["fc1"]), _fold(weights["fc2"]), _fold(weights["fc"])]
        return out

    def load(self, weights):
        table = self.layer_table(weights)
        L = _lib.lib()
        assert L.ssdr_randla_num_layers(self._h) == len(table)
        for i, (w, b) in enumerate(table):
            cin, cout, hb = C.c_int(), C.c_int(), C.c_int()
            _lib.check(L.ssdr_randla_layer_shape(self._h, i, C.byref(cin), C.byref(cout), C.byref(hb)))
            assert w.shape == (cin.value, cout.value), "layer %d: %s vs (%d,%d)" % (i, w.shape, cin.value, cout.value)
            w32 = np.ascontiguousarray(w, np.float32)
            b32 = None if b is None else np.ascontiguousarray(b, np.float32)
            _lib.check(L.ssdr_randla_set_layer(self._h, i, _lib.ptr(w32), _lib.ptr(b32)))
        return self

    def infer_dev(self, B, N, d_features, d_xyz, d_neigh, d_interp, d_probs, d_feat, stream=None):
        """Everything device-resident (pointers as ints); enqueue only."""
        L = self.config.num_layers
        arr = C.c_void_p * L
        r = np.asarray(self.config.sub_sampling_ratio, np.int32)
        _lib.check(_lib.lib().ssdr_randla_infer_dev(self._h, B, N, d_features, d_xyz, _lib.ptr(r), arr(*d_neigh), arr(*d_interp),
                                                    d_probs, d_feat, stream))

    def infer(self, features, xyz):
        """features [B,N,in_dim] f32, xyz [B,N,3] f32 (host) -> (probs [B*N,C], last_second_features [B*N,32]).
        Builds the KNN pyramid on the device (tf_map, s3dis_dataset.py:156-183) and runs the network."""
        cfg = self.config
        features = np.ascontiguousarray(features, np.float32)
        xyz = np.ascontiguousarray(xyz, np.float32)
        B, N = xyz.shape[0], xyz.shape[1]
        L, K = cfg.num_layers, cfg.k_n
        sizes = [N]
        for r in cfg.sub_sampling_ratio:
            sizes.append(sizes[-1] // r)
        d_xyz = _lib.DevArray.from_host(xyz)
        d_feat_in = _lib.DevArray.from_host(features)
        neigh = [_lib.DevArray((B, sizes[i], K), np.int32) for i in range(L)]
        interp = [_lib.DevArray((B, sizes[i], 1), np.int32) for i in range(L)]
        arr = C.c_void_p * L
        r = np.asarray(cfg.sub_sampling_ratio, np.int32)
        _lib.check(_lib.lib().ssdr_knn_pyramid_dev(d_xyz.ptr, B, N, L, _lib.ptr(r), K, arr(*[a.ptr for a in neigh]), None,
                                                   arr(*[a.ptr for a in interp]), None))
        probs = _lib.DevArray((B * N, cfg.num_classes), np.float32)
        feat = _lib.DevArray((B * N, 32), np.float32)
        self.infer_dev(B, N, d_feat_in.ptr, d_xyz.ptr, [a.ptr for a in neigh], [a.ptr for a in interp], probs.ptr, feat.ptr)
        _lib.sync()
        return probs.to_host(), feat.to_host()

"""Evaluation tail (scope row N2): vote smoothing, re-projection, confusion matrix, IoU against NumPy / sklearn
restatements of RandLANet.py:326-334, 353-411 and helper_tool.py:237-262."""
import numpy as np


def _iou_ref(confusions):            # helper_tool.py:237-262, restated
    confusions = np.asarray(confusions)
    TP = np.diagonal(confusions, axis1=-2, axis2=-1)
    TP_plus_FN = np.sum(confusions, axis=-1)
    TP_plus_FP = np.sum(confusions, axis=-2)
    IoU = TP / (TP_plus_FP + TP_plus_FN - TP + 1e-6)
    mask = TP_plus_FN < 1e-3
    counts = np.sum(1 - mask, axis=-1, keepdims=True)
    mIoU = np.sum(IoU, axis=-1, keepdims=True) / (counts + 1e-6)
    IoU += mask * mIoU
    return IoU


def test_vote_smoothing_projection_confusion_iou(backend):
    from sklearn.metrics import confusion_matrix
    from ssdr_al import evaluate
    rng = np.random.default_rng(6)
    n_sub, n_raw, C = (3000, 9000, 13) if backend == "emu" else (60000, 400000, 13)
    sub = rng.random((n_sub, 3), dtype=np.float32) * 4
    raw = (sub[rng.integers(0, n_sub, n_raw)] + rng.normal(0, 0.01, (n_raw, 3))).astype(np.float32)
    sub_lab = rng.integers(0, C - 1, n_sub).astype(np.int32)           # class C-1 absent: exercises the mask branch
    raw_lab = rng.integers(0, C - 1, n_raw).astype(np.int32)
    acc = evaluate.VoteAccumulator(n_sub, C)
    ref = np.zeros((n_sub, C), np.float32)
    for it in range(3):
        m = 2048 if backend == "emu" else 40960
        p_idx = rng.integers(0, n_sub, m).astype(np.int32)
        p_idx[-m // 8:] = p_idx[: m // 8]                               # padded tile: repeated points
        probs = rng.dirichlet(np.ones(C), m).astype(np.float32)
        acc.update(p_idx, probs)
        test_smooth = 0.95
        ref[p_idx] = test_smooth * ref[p_idx] + (1 - test_smooth) * probs                 # RandLANet.py:333, verbatim arithmetic
    assert np.array_equal(acc.probs(), ref)                             # bit-exact, repeated indices included
    preds, conf, iou = acc.confusion(sub_lab)
    assert np.array_equal(preds, np.argmax(ref, 1))
    assert np.array_equal(conf, confusion_matrix(sub_lab, np.argmax(ref, 1), labels=np.arange(C)))
    assert np.allclose(iou, _iou_ref(conf.astype(np.float64)), rtol=1e-12)
    # re-projection to the raw cloud (data_prepare_s3dis.py:69, RandLANet.py:378-395)
    proj = evaluate.project_indices(sub, raw)
    d = ((raw[:, None, :].astype(np.float64) - sub[None, proj[:64]].astype(np.float64)) ** 2).sum(-1) if False else None
    brute = np.array([np.argmin(((sub - r) ** 2).sum(1)) for r in raw[:200]])
    assert np.array_equal(proj[:200], brute)
    preds_r, conf_r, iou_r = acc.confusion(raw_lab, proj)
    assert np.array_equal(preds_r, np.argmax(ref[proj], 1))
    assert np.array_equal(conf_r, confusion_matrix(raw_lab, np.argmax(ref[proj], 1), labels=np.arange(C)))
    assert np.allclose(evaluate.IoU_from_confusions(conf_r), _iou_ref(conf_r.astype(np.float64)), rtol=1e-12)

"""CPU: the stage-b statistics (train_boxpc.py ClassificationStats / BoxDeltaIOUStats) against direct restatements."""
import numpy as np

from fake_t3d import FakeLib
from oracle import ref_box as RB
from transferable3d_amd.boxpc_stats import ALL_CLASSES, BoxDeltaIOUStats, ClassificationStats
from transferable3d_amd.engine import Runtime


def test_classification_stats_formulas():
    s = ClassificationStats(ALL_CLASSES)
    pred = np.array([1, 1, 0, 0, 1, 0, 1, 1])
    y = np.array([1, 0, 1, 0, 1, 0, 0, 1])
    cls = np.array([0, 0, 0, 0, 3, 3, 3, 3])
    s.add_prediction(pred[:5], y[:5], cls[:5])
    s.add_prediction(pred[5:], y[5:], cls[5:])
    s.add_loss(2.0)
    s.add_loss(4.0)
    st = s.get_batch_stats()
    assert sorted(st) == ['bed', 'chair'] and s.get_mean_loss() == 3.0
    prec, rec = 1 / (1 + 1 + 1e-3), 1 / (1 + 1 + 1e-3)                       # bed: tp 1, fp 1, fn 1
    assert np.allclose(st['bed'], (prec, rec, 2 * prec * rec / (prec + rec + 1e-3), 4))
    prec, rec = 2 / (2 + 1 + 1e-3), 2 / (2 + 0 + 1e-3)                       # chair: tp 2, fp 1, fn 0
    assert np.allclose(st['chair'], (prec, rec, 2 * prec * rec / (prec + rec + 1e-3), 4))
    text = s.summarize_stats(st)
    assert text.splitlines()[0].split() == ['Classname', 'Prec', 'Recall', 'F1', 'Supp'] and 'MEAN' in text.splitlines()[-1]


def test_box_delta_iou_stats_against_the_reference_loop():
    r = np.random.RandomState(0)
    n = 40
    cls = r.randint(0, 4, n)

    def boxes(center, noise):
        return (center + r.normal(size=(n, 3)) * noise, r.randint(0, 12, n) * 0 + hc, hr + r.normal(size=n) * noise, cls.copy(),
                sr + r.normal(size=(n, 3)) * noise * 0.3)
    hc, hr, sr = r.randint(0, 12, n), r.uniform(-0.2, 0.2, n), r.normal(size=(n, 3)) * 0.1
    c0 = r.normal(size=(n, 3)) * 0.3 + [0, 0, 3]
    y = (c0, hc, hr, cls, sr)
    ori, dele = boxes(c0, 0.25), boxes(c0, 0.08)
    st = BoxDeltaIOUStats(ALL_CLASSES, Runtime(device='cpu', lib=FakeLib()))
    st.add_prediction([v[:25] for v in ori], [v[:25] for v in dele], [v[:25] for v in y], cls[:25])
    st.add_prediction([v[25:] for v in ori], [v[25:] for v in dele], [v[25:] for v in y], cls[25:])
    got = st.get_batch_stats()

    def iou(a, b, i):
        box = lambda t: RB.get_3d_box(RB.class2size(int(t[3][i]), t[4][i]), RB.class2angle(int(t[1][i]), float(t[2][i])), t[0][i])
        return RB.box3d_iou(box(a), box(b))[0]
    for c in range(4):
        sel = np.nonzero(cls == c)[0]
        before, after = np.mean([iou(y, ori, i) for i in sel]), np.mean([iou(y, dele, i) for i in sel])
        g = got[ALL_CLASSES[c]]
        assert abs(g[0] - before) < 1e-6 and abs(g[1] - after) < 1e-6 and abs(g[2] - (after - before)) < 1e-6 and g[3] == len(sel)
    assert np.mean([got[k][2] for k in got]) > 0                        # the less noisy boxes overlap the label more
    text = st.summarize_stats(got)
    assert text.splitlines()[0].split() == ['Classname', 'Before', 'After', '+/-', 'Supp'] and ' +' in text.splitlines()[-1]

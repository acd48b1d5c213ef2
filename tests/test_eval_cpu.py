"""CPU: detection evaluation (eval_det.py / evaluate.py surface) on the specification library against the oracle's loop."""
import numpy as np

from eval_check import check_eval_det, check_predictions_round_trip
from fake_t3d import FakeLib
from oracle import ref_eval as RE
from transferable3d_amd import eval_det as E
from transferable3d_amd.engine import Runtime


def test_voc_ap_known_answers_and_oracle():
    # one detection, correct: AP = 1; precision 1 up to recall 1/2 then nothing: AP = 1/2
    assert E.voc_ap(np.array([1.0]), np.array([1.0])) == 1.0
    assert E.voc_ap(np.array([0.5]), np.array([1.0])) == 0.5
    # the envelope: a dip in precision is filled by a later higher value
    rec, prec = np.array([0.25, 0.5, 0.5, 0.75, 1.0]), np.array([1.0, 1.0, 2 / 3, 0.75, 0.8])
    assert abs(E.voc_ap(rec, prec) - (0.5 * 1.0 + 0.5 * 0.8)) < 1e-12
    assert abs(E.voc_ap(rec, prec, True) - (6 * 1.0 + 5 * 0.8) / 11) < 1e-12
    r = np.random.RandomState(0)
    for _ in range(20):
        n = r.randint(1, 30)
        tp = np.cumsum(r.uniform(size=n) < 0.6)
        rec, prec = tp / max(tp[-1] + r.randint(0, 3), 1), tp / np.arange(1, n + 1)
        for m07 in (False, True):
            assert abs(E.voc_ap(rec, prec, m07) - RE.voc_ap(rec, prec, m07)) < 1e-12


def test_eval_det_against_the_reference_loop():
    check_eval_det(Runtime(device='cpu', lib=FakeLib()))


def test_predictions_to_boxes_and_perfect_detections():
    check_predictions_round_trip(Runtime(device='cpu', lib=FakeLib()))


def test_test_semisup_evaluate_flag_reports_average_precision():
    """test_semisup --evaluate: inference on synthetic frustums, predictions in the 14-list layout, AP table of evaluate.py."""
    from transferable3d_amd import test_semisup as TS
    logs = []
    FLAGS = TS.build_flags(['--semi_type', 'F', '--use_one_hot', '--num_point', '128', '--num_channels', '4', '--batch_size', '4',
                            '--num_frustums', '8', '--refine', '1', '--pred_prefix', 'F2_', '--test', 'AB', '--synthetic', '--evaluate'])
    preds = TS.test(FLAGS, rt=Runtime(device='cpu', lib=FakeLib()), log=logs.append)
    assert len(preds) == 14 and preds[8] == [0.0] * 8 and preds[11] == list(range(8))
    table = [l for l in logs if str(l).startswith('Average Precision:')]
    assert len(table) == 1 and 'Mean AP:' in table[0]

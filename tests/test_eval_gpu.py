"""GPU: detection evaluation with the overlaps computed by t3d_box3d_iou_corners."""
import pytest

from eval_check import check_eval_det, check_predictions_round_trip
from transferable3d_amd.engine import Runtime

pytestmark = pytest.mark.gpu


def test_eval_det_against_the_reference_loop(hip_lib):
    check_eval_det(Runtime(lib=hip_lib))


def test_predictions_to_boxes_and_perfect_detections(hip_lib):
    check_predictions_round_trip(Runtime(lib=hip_lib))

"""Dataset-level constants of the SUN-RGBD Frustum-PointNet path.

Mirrors the constants the reference model builders import from
`sunrgbd/sunrgbd_detection/roi_seg_box3d_dataset.py:18-35` (class table, per-class mean box
sizes, NUM_HEADING_BIN / NUM_SIZE_CLUSTER / NUM_CLASS) and the derived `MEAN_DIMS_ARR`
(`semisup_models.py:22-24`, `semisup_v1_sunrgbd.py:25-27`, `boxpc_sunrgbd.py:23-25`).
Only the constants are on the hot path; the dataset classes themselves are out of scope.
"""
import numpy as np

type2class = {'bed': 0, 'table': 1, 'sofa': 2, 'chair': 3, 'toilet': 4, 'desk': 5,
              'dresser': 6, 'night_stand': 7, 'bookshelf': 8, 'bathtub': 9}
class2type = {v: k for k, v in type2class.items()}

# (l, w, h) mean box size per class, metres.
type_mean_size = {
    'bathtub':     np.array([0.765840, 1.398258, 0.472728]),
    'bed':         np.array([2.114256, 1.620300, 0.927272]),
    'bookshelf':   np.array([0.404671, 1.071108, 1.688889]),
    'chair':       np.array([0.591958, 0.552978, 0.827272]),
    'desk':        np.array([0.695190, 1.346299, 0.736364]),
    'dresser':     np.array([0.528526, 1.002642, 1.172878]),
    'night_stand': np.array([0.500618, 0.632163, 0.683424]),
    'sofa':        np.array([0.923508, 1.867419, 0.845495]),
    'table':       np.array([0.791118, 1.279516, 0.718182]),
    'toilet':      np.array([0.699104, 0.454178, 0.756250]),
}

NUM_HEADING_BIN = 12
NUM_SIZE_CLUSTER = 10
NUM_CLASS = 10
NUM_SEG_CLASSES = 2
BOX_OUT_DIMS = 3 + NUM_HEADING_BIN * 2 + NUM_SIZE_CLUSTER * 4   # 67
BOXPC_OUT_DIMS = 3 + 3 + 1 + 2                                    # 9

MEAN_DIMS_ARR = np.zeros((NUM_SIZE_CLUSTER, 3))
for _i in range(NUM_SIZE_CLUSTER):
    MEAN_DIMS_ARR[_i, :] = type_mean_size[class2type[_i]]

ORIENT_ANCHORS = np.arange(0, 2 * np.pi, 2 * np.pi / NUM_HEADING_BIN)

BN_EPS = 1e-3   # tf.contrib.layers.batch_norm default epsilon (tf_util.py:1660-1664)

"""ORACLE (test infrastructure only) -- the oracle's OWN copy of the data-set constants of the reference
(sunrgbd/sunrgbd_detection/roi_seg_box3d_dataset.py:18-35: class table, per-class mean box sizes (l, w, h) in metres,
NUM_HEADING_BIN / NUM_SIZE_CLUSTER / NUM_CLASS; MEAN_DIMS_ARR of semisup_models.py:22-24; batch-norm epsilon of
tf.contrib.layers.batch_norm, tf_util.py:1660-1664).  Kept apart from transferable3d_amd/constants.py so that the oracle and the
product do not share one source of truth: tests/test_oracle.py checks the two copies against each other and
tests/test_reference_vectors.py pins both on the values recorded from the reference itself."""
import numpy as np

type2class = {'bed': 0, 'table': 1, 'sofa': 2, 'chair': 3, 'toilet': 4, 'desk': 5, 'dresser': 6, 'night_stand': 7, 'bookshelf': 8,
              'bathtub': 9}
class2type = {v: k for k, v in type2class.items()}
type_mean_size = {
    'bathtub': (0.765840, 1.398258, 0.472728), 'bed': (2.114256, 1.620300, 0.927272), 'bookshelf': (0.404671, 1.071108, 1.688889),
    'chair': (0.591958, 0.552978, 0.827272), 'desk': (0.695190, 1.346299, 0.736364), 'dresser': (0.528526, 1.002642, 1.172878),
    'night_stand': (0.500618, 0.632163, 0.683424), 'sofa': (0.923508, 1.867419, 0.845495), 'table': (0.791118, 1.279516, 0.718182),
    'toilet': (0.699104, 0.454178, 0.756250)}
NUM_HEADING_BIN, NUM_SIZE_CLUSTER, NUM_CLASS = 12, 10, 10
BOX_OUT_DIMS = 3 + 2 * NUM_HEADING_BIN + 4 * NUM_SIZE_CLUSTER          # 67: centre, heading scores + residuals, size scores + residuals
MEAN_DIMS_ARR = np.array([type_mean_size[class2type[i]] for i in range(NUM_SIZE_CLUSTER)], dtype=np.float64)
ORIENT_ANCHORS = np.arange(0, 2 * np.pi, 2 * np.pi / NUM_HEADING_BIN)
BN_EPS = 1e-3

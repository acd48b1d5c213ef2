"""Hyper-parameter namespace + `--UPPER_CASE` command-line flags of the reference (models/config.py:13-205 constants,
208-350 `SpecialArgumentParser` / flag mirror).  Every constant is exposed as a flag of the same name, default and
type (lists via nargs='+', bools via `str2bool`, config.py:240); scripts add their own lower-case flags before
calling `cfg.parse_special_args()`, exactly as train_semisup.py:28-48 does.

The table below is data (names and published default values); the parser is generated from it."""
import argparse

import numpy as np

_STRONG_CLS = ['bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub']
_SEMI_TRAIN = ['bed', 'chair', 'toilet', 'desk', 'bathtub']
_KITTI = ['Car', 'Pedestrian', 'Cyclist']

DEFAULTS = dict(
    # algorithm
    BOX_PC_MASK_REPRESENTATION='', USE_NORMALIZED_BOX2D_AS_FEATS=False, NORMALIZE_PC_BEFORE_SEG=False,
    NORMALIZATION_METHOD='',
    # Box-PC fit sampling
    BOXPC_SAMPLING_METHOD='SAMPLE', BOXPC_SAMPLE_EQUAL_CLASS_WITH_PROB=1., BOXPC_PROPORTION_OF_BOXPC_FIT=0.5,
    BOXPC_NOFIT_BOUNDS=[0.01, 0.25], BOXPC_FIT_BOUNDS=[0.7, 1.0], BOXPC_CENTER_PERTURBATION=0.8,
    BOXPC_SIZE_PERTURBATION=0.2, BOXPC_ANGLE_PERTURBATION=float(np.pi),
    BOXPC_DELTA_LOSS_TYPE='huber', BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF=False, BOXPC_WEIGH_DELTA_LOSS_BY_CLS_CONF=False,
    BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA=True, BOXPC_WEIGH_DELTA_LOSS_BY_CLS_GT=False,
    BOXPC_WEIGHT_CLS=1., BOXPC_WEIGHT_DELTA=1., BOXPC_WEIGHT_DELTA_CENTER_PERCENT=0.34,
    BOXPC_WEIGHT_DELTA_SIZE_PERCENT=0.33, BOXPC_WEIGHT_DELTA_ANGLE_PERCENT=0.33, BOXPC_WEIGHT_CLUSTER=1.,
    # semi-supervised
    SEMI_MODEL='', SEMI_SAMPLING_METHOD='ALTERNATE_BATCH', SEMI_SAMPLE_EQUAL_CLASS_WITH_PROB=0.,
    SEMI_USE_LABELS2D_OF_CLASSES3D=False, SEMI_ADV_INITIAL_ITERS_BEFORE_TRAIN=0, SEMI_ADV_INITIAL_TRAINING_EPOCHS=0,
    SEMI_ADV_ITERS_FOR_D=0, SEMI_ADV_SAMPLE_EQUAL_CLASS_W_PROB=0., SEMI_ADV_DROPOUTS_FOR_G=0.5,
    SEMI_ADV_FLIP_LABELS_FOR_D_PROB=0., SEMI_ADV_SOFT_NOISY_LABELS_FOR_D=False, SEMI_ADV_FEATURE_MATCHING=False,
    SEMI_ADV_NORMALIZE_PC_TO_NEG1_TO_1=False, SEMI_ADV_TANH_FOR_LAST_LAYER_OF_G=True,
    SEMI_ADV_DIFF_MINIBATCH_REAL_VS_FAKE=False, SEMI_ADV_LEAKY_RELU=True, SEMI_ADV_AVERAGE_POOLING=False,
    SEMI_TRAIN_BOXPC_MODEL=False, SEMI_TRAIN_SEG_TRAIN_CLASS_AG_SEG=False, SEMI_TRAIN_BOX_TRAIN_CLASS_AG_TNET=False,
    SEMI_TRAIN_BOX_TRAIN_CLASS_AG_BOX=False, SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE=False,
    SEMI_BOXPC_FIT_ONLY_ON_2D_CLS=False, SEMI_WEIGH_BOXPC_DELTA_DURING_TEST=False,
    SEMI_REFINE_USING_BOXPC_DELTA_NUM=1, SEMI_INTRACLSDIMS_ONLY_ON_2D_CLS=True,
    SEMI_MULTIPLIER_FOR_WEAK_LOSS=1., SEMI_WEIGHT_REG_DELTA_LOSS=0., SEMI_WEIGHT_G_LOSS=0.4,
    SEMI_WEIGHT_G_FEATURE_MATCH_LOSS=0., SEMI_WEIGHT_BOXPC_FIT_LOSS=1., SEMI_WEIGHT_BOXPC_INTRACLS_FIT_LOSS=0.,
    SEMI_WEIGHT_BOXPC_KMEANS_LOSS=0.,
    # weak losses
    WEAK_WEIGHT_CROSS_ENTROPY=5., WEAK_CLS_WEIGHTS_CROSS_ENTROPY=[2., 0.333, 1., 0.333], WEAK_WEIGHT_VARIANCE=1.,
    WEAK_WEIGHT_INACTIVE=4., WEAK_WEIGHT_BINARY=0.002, WEAK_WEIGHT_ANCHOR_CLS=1., WEAK_WEIGHT_CENTER_YVAR=1.,
    WEAK_WEIGHT_INACTIVE_VOLUME=0., WEAK_WEIGHT_REPROJECTION=0.01, WEAK_WEIGHT_SURFACE=1.,
    WEAK_WEIGHT_INTRACLASSVAR=0., WEAK_TRAIN_SEG_W_SURFACE=False, WEAK_TRAIN_BOX_W_REPROJECTION=[True, True, True],
    WEAK_TRAIN_BOX_W_SURFACE=[True, False, True], WEAK_VARIANCE_LOSS_MARGIN=1.5, WEAK_USE_GT_ANCHOR_CLS_LABEL=False,
    WEAK_INACTIVE_VOL_LOSS_MARGINS=[10., 0., 0.], WEAK_INACTIVE_VOL_ONLY_ON_2D_CLS=True,
    WEAK_REPROJECTION_USE_SOFTMAX_PROJ=False, WEAK_REPROJECTION_SOFTMAX_SCALE=10.,
    WEAK_REPROJECTION_ONLY_ON_2D_CLS=False, WEAK_REPROJECTION_CLIP_LOWERB_LOSS=True,
    WEAK_REPROJECTION_CLIP_PRED_BOX=False, WEAK_REPROJECTION_LOSS_TYPE='huber', WEAK_REPROJECTION_DILATE_FACTOR=1.5,
    WEAK_SURFACE_MARGIN=0., WEAK_SURFACE_LOSS_WT_FOR_INNER_PTS=0.8, WEAK_SURFACE_LOSS_SCALE_DIMS=0.9,
    WEAK_DIMS_LOSS_TYPE='huber', WEAK_DIMS_USE_MARGIN_LOSS=True, WEAK_DIMS_SD_MARGIN=0.2, WEAK_DIMS_EMA_DECAY=0.99,
    # strong losses
    STRONG_WEIGHT_CROSS_ENTROPY=1., STRONG_BOX_MULTIPLER=0.1, STRONG_WEIGHT_CENTER=1., STRONG_WEIGHT_ORIENT_CLS=1.,
    STRONG_WEIGHT_ORIENT_REG=20., STRONG_WEIGHT_DIMS_CLS=1., STRONG_WEIGHT_DIMS_REG=20., STRONG_WEIGHT_TNET_CENTER=1.,
    STRONG_WEIGHT_CORNER=1.,
    # class lists
    SUNRGBD_STRONG_TRAIN_CLS=_STRONG_CLS, SUNRGBD_SEMI_TRAIN_CLS=_SEMI_TRAIN,
    SUNRGBD_SEMI_TEST_CLS=[c for c in _STRONG_CLS if c not in _SEMI_TRAIN], SUNRGBD_WEAK_TRAIN_CLS=_STRONG_CLS,
    SUNRGBD_WEAK_TEST_CLS=_STRONG_CLS, KITTI_ALL_CLS=_KITTI, KITTI_SEMI_TRAIN_CLS=[], KITTI_SEMI_TEST_CLS=_KITTI,
)

# the reference declares this float flag with type=str2bool (config.py:321); kept for flag compatibility
_TYPE_QUIRKS = {'WEAK_REPROJECTION_SOFTMAX_SCALE': 'str2bool'}
# int-valued constants whose flags are declared type=int
_INT_FLAGS = {'SEMI_ADV_INITIAL_ITERS_BEFORE_TRAIN', 'SEMI_ADV_INITIAL_TRAINING_EPOCHS', 'SEMI_ADV_ITERS_FOR_D',
              'SEMI_REFINE_USING_BOXPC_DELTA_NUM'}
# list-valued string constants need an explicit element type when the default list is empty
_STR_LISTS = {'KITTI_SEMI_TRAIN_CLS'}


def str2bool(v):
    return v.lower() in ('yes', 'true', 't', '1')


class SpecialArgumentParser(argparse.ArgumentParser):
    """argparse + attributes attached after parsing (config.py:208-238): `set_attributes([(name, value), ...])`,
    `parse_special_args()` returns the namespace with those attributes and a `config_str` summary."""

    def __init__(self):
        argparse.ArgumentParser.__init__(self)
        self.attr_names_and_vals = None

    def set_attributes(self, attr_names_and_vals):
        self.attr_names_and_vals = attr_names_and_vals

    def parse_special_args(self, args=None):
        flags = self.parse_args(args)
        for name, val in (self.attr_names_and_vals or []):
            setattr(flags, name, val)
        setattr(flags, 'config_str', self.get_config_str(flags))
        return flags

    def get_config_str(self, c):
        if not hasattr(c, 'mode'):
            c.mode = None
        lines = ['', ' * REMEMBER TO CHECK THE CONFIGURATIONS * ', '', '  [MODE: %s]' % c.mode, '  [CLASSES]']
        for k in ('SUNRGBD_STRONG_TRAIN_CLS', 'SUNRGBD_SEMI_TRAIN_CLS', 'SUNRGBD_SEMI_TEST_CLS', 'SUNRGBD_WEAK_TRAIN_CLS',
                  'SUNRGBD_WEAK_TEST_CLS'):
            lines.append('    %-40s: %s' % (k, getattr(c, k)))
        return '\n'.join(lines) + '\n'


def _add_flag(parser, name, default):
    quirk = _TYPE_QUIRKS.get(name)
    if quirk == 'str2bool':
        parser.add_argument('--' + name, type=str2bool, default=default)
    elif isinstance(default, bool):
        parser.add_argument('--' + name, type=str2bool, default=default)
    elif isinstance(default, list):
        if name in _STR_LISTS or (default and isinstance(default[0], str)):
            et = str
        elif default and isinstance(default[0], bool):
            et = str2bool
        else:
            et = float
        parser.add_argument('--' + name, nargs='+', type=et, default=default)
    elif name in _INT_FLAGS:
        parser.add_argument('--' + name, type=int, default=default)
    elif isinstance(default, float):
        parser.add_argument('--' + name, type=float, default=default)
    else:
        parser.add_argument('--' + name, type=str, default=default)


def make_parser():
    p = SpecialArgumentParser()
    p.set_attributes([])
    for name, default in DEFAULTS.items():
        _add_flag(p, name, default)
    return p


cfg = make_parser()

# module-level constants, as in the reference (`import config; config.STRONG_WEIGHT_CORNER`)
globals().update(DEFAULTS)

"""Synthetic frustum batches: the metric's inputs.

No dataset exists on the GPU box, so bench.py, smoke() and the parity tests all draw their batches
here, from `np.random.RandomState(seed)`, in the feed layout of the reference's batch tuple
(`roi_semi_dataset.py:531-534`, `semisup_v1_sunrgbd.placeholder_inputs` 37-67,
`boxpc_sunrgbd.placeholder_inputs` 33-54).
"""
import numpy as np

from .constants import NUM_CLASS, NUM_HEADING_BIN, NUM_SIZE_CLUSTER


def make_batch(batch_size=32, num_point=1024, num_channel=4, seed=1234, is_data_2D=0, boxpc=False,
               dropout_scopes=None):
    """One synthetic batch as a dict of NumPy arrays (fp32 / int32).

    pc: x,y ~ N(0,0.6^2), z ~ U(1,6), extra channels ~ U(0,1).  Labels as in SURVEY 8(d).
    `dropout_scopes`: {scope: (shape, keep_prob)} -> injected 0/1 keep masks (parity tests).
    """
    r = np.random.RandomState(seed)
    B, N, C = batch_size, num_point, num_channel
    pc = np.empty((B, N, C), np.float32)
    pc[:, :, 0:2] = r.normal(0.0, 0.6, size=(B, N, 2))
    pc[:, :, 2] = r.uniform(1.0, 6.0, size=(B, N))
    if C > 3:
        pc[:, :, 3:] = r.uniform(0.0, 1.0, size=(B, N, C - 3))
    cls = r.randint(0, NUM_CLASS, size=B)
    one_hot = np.zeros((B, NUM_CLASS), np.float32)
    one_hot[np.arange(B), cls] = 1.0
    y_seg = (r.uniform(size=(B, N)) < 0.3).astype(np.int32)
    cnt = np.maximum(y_seg.sum(1, keepdims=True), 1)
    y_center = ((pc[:, :, 0:3] * y_seg[:, :, None]).sum(1) / cnt + r.normal(0, 0.1, size=(B, 3))).astype(np.float32)
    batch = {
        'pc': pc, 'one_hot_vec': one_hot, 'y_seg': y_seg, 'y_center': y_center,
        'y_orient_cls': r.randint(0, NUM_HEADING_BIN, size=B).astype(np.int32),
        'y_orient_reg': r.uniform(-np.pi / NUM_HEADING_BIN, np.pi / NUM_HEADING_BIN, size=B).astype(np.float32),
        'y_dims_cls': cls.astype(np.int32),
        'y_dims_reg': r.normal(0, 0.1, size=(B, 3)).astype(np.float32),
        'is_data_2D': np.full((B,), int(is_data_2D), np.int32),
    }
    assert NUM_SIZE_CLUSTER == NUM_CLASS
    # camera side of the weak losses (semisup_v1_sunrgbd.placeholder_inputs: Rtilt, K, rot_frust, box2D, img_dim): a SUN-RGBD-like
    # calibration, a small tilt, a frustum angle, and a 2-D label box around the image centre.  Drawn from a generator of its own so
    # that the fields above are what they were before these existed (committed golden fixtures).
    r2 = np.random.RandomState(seed + 7919)
    tilt = r2.normal(0, 0.05, size=B)
    batch['Rtilt'] = np.stack([np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]]) for t in tilt]).astype(np.float32)
    batch['K'] = np.tile(np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]], np.float32), (B, 1, 1))
    batch['rot_frust'] = r2.normal(0, 0.3, size=(B, 1)).astype(np.float32)
    batch['img_dim'] = np.tile(np.array([530.0, 730.0], np.float32), (B, 1))
    cx, cy, hw, hh = r2.uniform(150, 580, B), r2.uniform(100, 430, B), r2.uniform(40, 220, B), r2.uniform(40, 200, B)
    batch['box2D'] = np.stack([cx - hw, cy - hh, cx + hw, cy + hh], 1).astype(np.float32)
    if boxpc:
        batch['y_box_iou'] = r.uniform(0, 1, size=B).astype(np.float32)
        batch['y_center_delta'] = r.uniform(-0.2, 0.2, size=(B, 3)).astype(np.float32)
        batch['y_dims_delta'] = r.uniform(-0.2, 0.2, size=(B, 3)).astype(np.float32)
        batch['y_orient_delta'] = r.uniform(-0.2, 0.2, size=B).astype(np.float32)
    if dropout_scopes:
        batch['dropout_masks'] = {k: (r.uniform(size=shape) < keep).astype(np.float32)
                                  for k, (shape, keep) in dropout_scopes.items()}
    return batch

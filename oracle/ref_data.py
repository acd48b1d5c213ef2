"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's per-sample batch assembly.  get_sample / angle2class /
rotate_pc_along_y are PINNED on outputs of the reference's own ROISegBoxDataset.__getitem__ / get_batch and
ROISemiDataset.get_classes3D, executed in the build container on a synthetic frustum file with the np.random draws recorded
(tests/golden/make_reference_vectors.py -> tests/test_reference_vectors.py).  The Box-PC sample generator below is pinned the same way on BoxPCFitDataset.get, which
the reference ran with oracle/ref_box.box3d_iou standing in for its absent box_util: the generator's law is pinned, the IoU is not.

Follows sunrgbd/sunrgbd_detection/roi_semi_dataset.py:283-347 (`get_classes3D`: resample to N points with replacement,
rotate to the frustum's centre view, labels, flip / shift augmentation, angle and size classes) and 482-535 (`get_batch`),
with the helpers of roi_seg_box3d_dataset.py:37-77,346-368.  The random draws of the reference (np.random.choice /
random / randn) are arguments here, so that the device kernel can be checked on identical draws.
"""
import numpy as np

from .ref_constants import MEAN_DIMS_ARR, NUM_CLASS, NUM_HEADING_BIN


def rotate_pc_along_y(pc, rot_angle):
    """roi_seg_box3d_dataset.py:37-45 (on a copy)."""
    pc = np.array(pc, dtype=np.float64, copy=True)
    c, s = np.cos(rot_angle), np.sin(rot_angle)
    rotmat = np.array([[c, -s], [s, c]])
    pc[:, [0, 2]] = np.dot(pc[:, [0, 2]], rotmat.T)
    return pc


def angle2class(angle, num_class):
    """roi_seg_box3d_dataset.py:47-62."""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(num_class)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def get_sample(points, seg, frustum_angle, box_center, heading, size, cls, choice, flip, shift_randn, height_u, num_channel,
               rotate_to_center=True, random_flip=True, random_shift=True, is_2D=False):
    """One frustum -> (point_set [N,C], seg [N], center [3], angle_class, angle_residual, size_class, size_residual [3],
    rot_angle, one_hot [10]).  `choice` [N] ints, `flip` bool (the reference flips when np.random.random() > 0.5),
    `shift_randn` ~ N(0,1), `height_u` ~ U(0,1)."""
    rot_angle = np.pi / 2.0 + frustum_angle                                   # roi_seg_box3d_dataset.py:346-347
    ps = rotate_pc_along_y(points, rot_angle) if rotate_to_center else np.array(points, dtype=np.float64)
    ps = ps[choice, :]                                                        # roi_semi_dataset.py:301-303
    sg = np.asarray(seg)[choice]
    center = np.array(box_center, dtype=np.float64)
    if rotate_to_center:
        center = rotate_pc_along_y(center[None, :], rot_angle)[0]             # :319-320
        heading_angle = heading - rot_angle                                   # :324-325
    else:
        heading_angle = heading
    size_class, size_residual = int(cls), np.asarray(size, dtype=np.float64) - MEAN_DIMS_ARR[int(cls)]   # :330, size2class
    if random_flip and flip:                                                  # :333-338
        ps[:, 0] *= -1
        center[0] *= -1
        heading_angle = np.pi - heading_angle
    if random_shift:                                                          # :339-346 (the clip bounds are the reference's)
        dist = np.sqrt(np.sum(center[0] ** 2 + center[1] ** 2))
        shift = np.clip(shift_randn * dist * 0.05, dist * 0.8, dist * 1.2)
        ps[:, 2] += shift
        center[2] += shift
        hs = height_u * 0.4 - 0.2
        ps[:, 1] += hs
        center[1] += hs
    acls, ares = angle2class(heading_angle, NUM_HEADING_BIN)
    one_hot = np.zeros(NUM_CLASS)
    one_hot[int(cls)] = 1
    if is_2D:       # get_classes2D (roi_semi_dataset.py:383-452): the caller passes random_flip = random_shift = False; labels are zeros
        return ps[:, :num_channel], np.zeros_like(sg), np.zeros_like(center), 0, 0.0, 0, np.zeros_like(size_residual), rot_angle, one_hot
    return ps[:, :num_channel], sg, center, acls, ares, size_class, size_residual, rot_angle, one_hot


def get_batch(ds, sample, choice, flip, shift_randn, height_u, num_channel, is_2D=None, **kw):
    """roi_semi_dataset.py:482-535 on a ragged data set `ds` (dict: points [total,C], seg [total], offsets [F+1],
    frustum_angle, box_center, heading, size, cls) for the frustum indices `sample` [B].  `is_2D` [B]: the slot holds a frustum of
    the 2-D-label list (get_classes2D: no flip / shift, zero labels, is_data_2D = 1)."""
    out = {k: [] for k in ('pc', 'y_seg', 'y_center', 'y_orient_cls', 'y_orient_reg', 'y_dims_cls', 'y_dims_reg', 'rot_angle',
                           'one_hot_vec')}
    for i, f in enumerate(sample):
        lo, hi = int(ds['offsets'][f]), int(ds['offsets'][f + 1])
        kw_i = dict(kw)
        if is_2D is not None and is_2D[i]:
            kw_i.update(random_flip=False, random_shift=False, is_2D=True)
        r = get_sample(ds['points'][lo:hi], ds['seg'][lo:hi], ds['frustum_angle'][f], ds['box_center'][f], ds['heading'][f],
                       ds['size'][f], ds['cls'][f], choice[i], bool(flip[i]), shift_randn[i], height_u[i], num_channel, **kw_i)
        for k, v in zip(out, r):
            out[k].append(v)
    res = {k: np.asarray(v) for k, v in out.items()}
    if is_2D is not None:
        res['is_data_2D'] = np.asarray(is_2D, dtype=np.int32)
    return res


# ---- Box-PC Fit samples --------------------------------------------------------------------------------------------------------
def perturb_box_to_diff_ious(box3d_center, size, heading_angle, iou_bounds, center_perturbation, size_perturbation, angle_perturbation,
                             draws):
    """box_pc_fit_dataset.py:211-244.  `draws` [T,7] are the uniforms in [0,1) the reference would take from np.random.uniform,
    candidate after candidate: 3 for the centre (mapped to U(-p,p)), 3 for the size, 1 for the angle (mapped to U(0,p)).
    Returns (new_center, new_size, new_heading, iou3d, y_center_delta, y_size_delta, y_angle_delta, tries)."""
    from oracle.ref_box import get_box3d_iou
    iou_mean = np.mean(iou_bounds)
    cp, sp, ap = center_perturbation * (1 - iou_mean), size_perturbation * (1 - iou_mean), angle_perturbation * (1 - iou_mean)
    box3d_center, size = np.asarray(box3d_center, np.float64), np.asarray(size, np.float64)
    for count, u in enumerate(np.asarray(draws, np.float64)):
        y_center_delta = (2 * u[0:3] - 1) * cp
        y_size_delta = np.multiply(size, (2 * u[3:6] - 1) * sp)
        y_angle_delta = u[6] * ap
        new_center, new_size, new_heading = box3d_center + y_center_delta, size + y_size_delta, heading_angle + y_angle_delta
        iou3d, _ = get_box3d_iou(box3d_center, size, heading_angle, new_center, new_size, new_heading)
        assert 0.0 <= iou3d <= 1.0 + 1e-12
        if iou_bounds[0] < iou3d < iou_bounds[1]:                       # inrange(): strict on both sides (44-45)
            return new_center, new_size, new_heading, iou3d, y_center_delta, y_size_delta, y_angle_delta, count + 1
    return None


def boxpc_sample_labels(center, heading, size, cls, is_fit, fit_bounds, nofit_bounds, perturbations, draws):
    """The label half of BoxPCFitDataset.get (box_pc_fit_dataset.py:170-185) for a frustum whose augmented label box is
    (center, heading, size): the perturbed box in class / residual form and the regression targets."""
    out = perturb_box_to_diff_ious(center, size, heading, fit_bounds if is_fit else nofit_bounds, *perturbations, draws)
    if out is None:
        return None
    new_center, new_size, new_heading, iou, dc, ds, da, tries = out
    acls, ares = angle2class(new_heading, NUM_HEADING_BIN)
    return dict(x_center=new_center, x_orient_cls=acls, x_orient_reg=ares, x_dims_cls=cls, x_dims_reg=new_size - MEAN_DIMS_ARR[cls],
                y_box_iou=iou, y_center_delta=dc, y_dims_delta=ds, y_orient_delta=da, tries=tries)


# ---- class-balanced batches -----------------------------------------------------------------------------------------------------
def sample_equal_per_class(bsize, cls_to_idx, shuffled_split, member_draws):
    """roi_semi_dataset.py:558-567 / box_pc_fit_dataset.py:367-377.  `cls_to_idx`: list of index lists, one per class key;
    `shuffled_split`: np.array_split([1]*bsize, n) after random.shuffle (a list of n arrays of ones); `member_draws`: uniforms in
    [0,1) standing for np.random.choice(list, len(group), replace=True) (index = floor(u * len(list)))."""
    choices, k = [], 0
    for i, group in enumerate(shuffled_split):
        lst = cls_to_idx[i]
        for _ in range(len(group)):
            choices.append(lst[min(int(member_draws[k] * len(lst)), len(lst) - 1)])
            k += 1
    return choices

"""TEST INFRASTRUCTURE ONLY (see oracle/README.md): torch-autograd restatement of the reference's weak box losses.

Follows /root/reference/models/weak_losses.py (get_reprojection_loss 69-229, get_surface_loss 231-259,
get_inactive_volume_loss_v1 39-67, loss_for_deviation_from_range 15-37) and the geometry helpers of
/root/reference/models/tf_util.py they call (line numbers at each function).  Parity unpinned: TensorFlow 1.x cannot run here;
every function is a line-by-line transcription, including the behaviours that look unintended and are kept:
  * get_surface_loss multiplies by `soft_mask`, not by the stop-gradient `mask` it prepares (weak_losses.py:238,250): the surface
    loss always back-propagates into the segmentation logits, whatever WEAK_TRAIN_SEG_W_SURFACE says;
  * tf_distance_to_closest_3D_box_surface takes the minimum over the RAW six distances; the "cleaned" ones (points whose ray
    leaves the box set to 1e8) are computed and dropped (tf_util.py:694-707);
  * tf_dilate_2D_bboxes computes height = top - bottom (negative), so the dilated box comes out with top and bottom swapped
    relative to the input convention (tf_util.py:486-514, its own TODO).
Gradient conventions: reduce_min / reduce_max split the gradient evenly between ties (torch.amin / amax do the same), tf.abs has
gradient sign(x), tf.maximum(0, x) passes the gradient where x > 0."""
import torch


def tf_huber(labels, predictions, delta=1.0):
    """tf.losses.huber_loss(..., reduction=NONE)."""
    e = (predictions - labels).abs()
    q = torch.clamp(e, max=delta)
    return 0.5 * q * q + delta * (e - q)


def elem_loss(labels, predictions, loss_type):
    if loss_type == 'huber':
        return tf_huber(labels, predictions)
    if loss_type == 'mse':
        return (predictions - labels) ** 2
    raise Exception('Not implemented: %s' % loss_type)


def loss_for_deviation_from_range(val, lower_b, upper_b, loss='huber'):
    """weak_losses.py:15-37."""
    lo = (val < lower_b).to(val.dtype)
    hi = (val > upper_b).to(val.dtype)
    return lo * elem_loss(lower_b, val, loss) + hi * elem_loss(upper_b, val, loss)


# ---- geometry (tf_util.py) ---------------------------------------------------------------------------------------------------
def rot_box_params_multi(box, angles):
    """tf_util.py:1045-1072: rotation about the y axis by angles[b]."""
    center, dims, orient = box
    a = angles.reshape(-1)
    ca, sa = torch.cos(a), torch.sin(a)
    x, y, z = center[:, 0], center[:, 1], center[:, 2]
    return torch.stack([ca * x + sa * z, y, -sa * x + ca * z], 1), dims, orient + a


def flip_axis_to_camera(pc):
    """tf_util.py:816-823: depth X,Y,Z -> camera X,-Z,Y."""
    return torch.stack([pc[..., 0], -pc[..., 2], pc[..., 1]], -1)


def flip_axis_to_depth(pc):
    """tf_util.py:826-830."""
    return torch.stack([pc[..., 0], pc[..., 2], -pc[..., 1]], -1)


def create_3D_box_by_vertices_multi(box, apply_translation=False):
    """tf_util.py:841-891 -> (B,8,3) in upright camera coordinates."""
    centers, dims, orient = box
    l, w, h = dims[:, 0:1], dims[:, 1:2], dims[:, 2:3]
    c, s = torch.cos(-orient), torch.sin(-orient)
    xc = torch.cat([-l / 2, l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2], 1)
    yc = torch.cat([w / 2, w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2], 1)
    zc = torch.cat([h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2], 1)
    x3 = c[:, None] * xc - s[:, None] * yc
    y3 = s[:, None] * xc + c[:, None] * yc
    corners = flip_axis_to_camera(torch.stack([x3, y3, zc], -1))             # (B,8,3)
    if apply_translation:
        corners = corners + centers[:, None, :]
    return centers, corners


def project_upright_depth_to_image(pc, Rtilt, K):
    """tf_util.py:798-813: (B,N,3) upright depth -> (B,N,2) image uv."""
    pc2 = torch.matmul(Rtilt.transpose(1, 2), pc.transpose(1, 2)).transpose(1, 2)
    pc2 = flip_axis_to_camera(pc2)
    uv = torch.matmul(pc2, K.transpose(1, 2))
    return torch.stack([uv[..., 0] / uv[..., 2], uv[..., 1] / uv[..., 2]], -1)


def get_2D_bbox_of_projection(points, Rtilts, Ks):
    """tf_util.py:364-377, 416-432: hard min / max of the projected points -> (B,4) left, top, right, bottom."""
    uv = project_upright_depth_to_image(flip_axis_to_depth(points), Rtilts, Ks)
    return torch.stack([uv[..., 0].amin(1), uv[..., 1].amin(1), uv[..., 0].amax(1), uv[..., 1].amax(1)], 1)


def get_2D_bbox_of_softmax_projection(points, Rtilts, Ks, scale):
    """tf_util.py:379-414, 434-449: soft extreme = sum of the coordinates weighted by a stop-gradient softmax of the closeness."""
    uv = project_upright_depth_to_image(flip_axis_to_depth(points), Rtilts, Ks)
    u, v = uv[..., 0], uv[..., 1]
    lb, tb, rb, bb = u.amin(1, keepdim=True), v.amin(1, keepdim=True), u.amax(1, keepdim=True), v.amax(1, keepdim=True)
    width, height = (rb - lb).abs().detach(), (bb - tb).abs().detach()
    sm = lambda x: torch.softmax(x * scale, 1).detach()
    return torch.stack([(u * sm((rb - u) / width)).sum(1), (v * sm((bb - v) / height)).sum(1),
                        (u * sm((u - lb) / width)).sum(1), (v * sm((v - tb) / height)).sum(1)], 1)


def dilate_2D_bboxes(b, f):
    """tf_util.py:486-514 (height = top - bottom, as written)."""
    left, top, right, bottom = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    cx, cy = (left + right) / 2., (top + bottom) / 2.
    nw, nh = f * (right - left), f * (top - bottom)
    return torch.stack([cx - nw / 2., cy + nh / 2., cx + nw / 2., cy - nh / 2.], 1)


def clip_2D_bbox_to_image_dims_multi(b, img_dims):
    """tf_util.py:517-540: image_dim = (rows, cols)."""
    rows, cols = img_dims[:, 0], img_dims[:, 1]
    zero = torch.zeros_like(rows)
    return torch.stack([torch.maximum(zero, b[:, 0]), torch.maximum(zero, b[:, 1]), torch.minimum(cols, b[:, 2]),
                        torch.minimum(rows, b[:, 3])], 1)


# ---- losses -------------------------------------------------------------------------------------------------------------------
def get_reprojection_loss(pred_box_reg, box2D, Rtilts, Ks, img_dims, rot_frust, use_softmax_projection, softmax_scale_factor,
                          dilate_factor, clip_lower_b_loss, clip_pred_box, loss_type, train_box, ep=None):
    """weak_losses.py:69-229, reduce_loss=False -> (B,)."""
    center, dims, orient = pred_box_reg
    center = center if train_box[0] else center.detach()
    dims = dims if train_box[1] else dims.detach()
    orient = orient if train_box[2] else orient.detach()
    box = rot_box_params_multi((center, dims, orient), 1 * rot_frust)
    _, pts = create_3D_box_by_vertices_multi(box, apply_translation=True)
    if use_softmax_projection:
        pb = get_2D_bbox_of_softmax_projection(pts, Rtilts, Ks, softmax_scale_factor)
    else:
        pb = get_2D_bbox_of_projection(pts, Rtilts, Ks)
    if clip_pred_box:
        pb = clip_2D_bbox_to_image_dims_multi(pb, img_dims)
        small = clip_2D_bbox_to_image_dims_multi(box2D, img_dims)
        big = clip_2D_bbox_to_image_dims_multi(dilate_2D_bboxes(box2D, dilate_factor), img_dims)
        dev = loss_for_deviation_from_range
        out = dev(pb[:, 0], big[:, 0], small[:, 0], loss_type) + dev(pb[:, 1], big[:, 1], small[:, 1], loss_type) + \
            dev(pb[:, 2], small[:, 2], big[:, 2], loss_type) + dev(pb[:, 3], small[:, 3], big[:, 3], loss_type)
    else:
        small = clip_2D_bbox_to_image_dims_multi(box2D, img_dims)
        big = dilate_2D_bboxes(box2D, dilate_factor)
        bigc = clip_2D_bbox_to_image_dims_multi(big, img_dims)
        notc = (big == bigc).to(pb.dtype)
        if clip_lower_b_loss:
            dev = loss_for_deviation_from_range
            out = notc[:, 0] * dev(pb[:, 0], bigc[:, 0], small[:, 0], loss_type) + \
                notc[:, 1] * dev(pb[:, 1], bigc[:, 1], small[:, 1], loss_type) + \
                notc[:, 2] * dev(pb[:, 2], small[:, 2], bigc[:, 2], loss_type) + \
                notc[:, 3] * dev(pb[:, 3], small[:, 3], bigc[:, 3], loss_type)
            out = torch.clamp(out, max=1000.)
        else:
            sides = []
            for i, less_is_inside in ((0, True), (1, True), (2, False), (3, False)):
                inner = elem_loss(small[:, i], pb[:, i], loss_type)
                outer = elem_loss(bigc[:, i], pb[:, i], loss_type)
                if less_is_inside:      # left / top: no inner violation left of the inner box, no outer violation right of the outer box
                    inner = torch.where(pb[:, i] < small[:, i], torch.zeros_like(inner), inner)
                    outer = torch.where(pb[:, i] > bigc[:, i], torch.zeros_like(outer), outer)
                else:
                    inner = torch.where(pb[:, i] > small[:, i], torch.zeros_like(inner), inner)
                    outer = torch.where(pb[:, i] < bigc[:, i], torch.zeros_like(outer), outer)
                sides.append(torch.clamp(inner + outer * notc[:, i], max=1000.))
            out = sides[0] + sides[1] + sides[2] + sides[3]
    if ep is not None:
        ep['reproj_pred_box3D_pts'], ep['reproj_proj_pred_box'] = pts, pb
    return out


def distance_to_closest_3D_box_surface_multi(pc, box):
    """tf_util.py:610-719 (the minimum over the six RAW ray distances) -> (B,N)."""
    center, dims, orient = box
    l, w, h = dims[:, 0], dims[:, 1], dims[:, 2]
    st, ct = torch.sin(orient), torch.cos(orient)
    zero, one = torch.zeros_like(st), torch.ones_like(st)
    R = torch.stack([torch.stack([ct, zero, st], 1), torch.stack([zero, one, zero], 1), torch.stack([-st, zero, ct], 1)], 1)   # (B,3,3)
    n0 = torch.tensor([[-1., 1., 0., 0., 0., 0.], [0., 0., -1., 1., 0., 0.], [0., 0., 0., 0., -1., 1.]], dtype=pc.dtype)
    z = torch.zeros_like(l)
    sp0 = torch.stack([torch.stack([l / 2, -l / 2, z, z, z, z], 1), torch.stack([z, z, h / 2, -h / 2, z, z], 1),
                       torch.stack([z, z, z, z, w / 2, -w / 2], 1)], 1)                                                    # (B,3,6)
    sp = torch.matmul(R, sp0).transpose(1, 2) + center[:, None, :]                                                        # (B,6,3)
    nrm = torch.matmul(R, n0[None].expand(R.shape[0], 3, 6)).transpose(1, 2)                                              # (B,6,3)
    ray = pc - center[:, None, :]                                                                                         # (B,N,3)
    perp = torch.einsum('bnd,bsd->bns', ray, nrm)
    q = ((sp - center[:, None, :]) * nrm).sum(2)                                                                          # (B,6)
    rn = torch.linalg.norm(ray, dim=2, keepdim=True)
    dcs = rn * (q[:, None, :] / (perp + 1e-5))
    at_center = (ray.abs().sum(2, keepdim=True) == 0)
    half = torch.stack([l / 2, l / 2, h / 2, h / 2, w / 2, w / 2], 1)[:, None, :].expand_as(dcs)
    dcs = torch.where(at_center.expand_as(dcs), half, dcs)
    return (rn - dcs).abs().amin(2)


def get_surface_loss(pred_box_reg, pc_xyz, soft_mask, margin, scale_dims_factor, train_box, ep=None):
    """weak_losses.py:231-259, reduce_loss=False -> (B,).  (train_seg only prepares an unused tensor: see the module docstring.)"""
    center, dims, orient = pred_box_reg
    center = center if train_box[0] else center.detach()
    dims = dims if train_box[1] else dims.detach()
    orient = orient if train_box[2] else orient.detach()
    d = distance_to_closest_3D_box_surface_multi(pc_xyz, (center, dims * scale_dims_factor, orient))
    per_point = torch.clamp(d - margin, min=0.) * soft_mask
    if ep is not None:
        ep['surface_min_dist'] = d
    return per_point.mean(1)


def get_inactive_volume_loss_v1(dims_reg, y_class, train_classes, margins):
    """weak_losses.py:39-67: per trained class mean(max(0, margin_i - l*w*h)), an empty class counts 0; mean over those classes."""
    out = []
    for i, on in enumerate(train_classes):
        if not on:
            continue
        sel = dims_reg[y_class.long() == i]
        if sel.shape[0] == 0:
            out.append(torch.zeros((), dtype=dims_reg.dtype))
            continue
        out.append(torch.clamp(margins[i] - sel.prod(1), min=0.).mean())
    return torch.stack(out).mean()

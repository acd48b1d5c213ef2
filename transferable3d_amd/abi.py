"""ctypes view of include/t3d.h: argument structs + loader for libt3d.so.

The product path has no fallback: `load()` raises if the HIP library is missing or does not export
every entry point the header declares.
"""
import ctypes as C
import os

F = C.POINTER(C.c_float)
I = C.POINTER(C.c_int32)
i32, f32 = C.c_int, C.c_float

TILE_ROWS = 128
F32, BF16 = 0, 1              # t3d.h: T3D_F32 / T3D_BF16 (element type of the per-point layer tensors + GEMM arithmetic)
DTYPE_BY_NAME = {'f32': F32, 'bf16': BF16}
ARITH_AUTO, ARITH_FP32_MFMA, ARITH_BF16X3, ARITH_BF16 = 0, 1, 2, 3      # t3d.h: T3D_ARITH_* (arithmetic of an fp32 GEMM launch)
ARITH_BY_NAME = {'auto': ARITH_AUTO, 'fp32_mfma': ARITH_FP32_MFMA, 'bf16x3': ARITH_BF16X3}
ARITH_NAMES = {ARITH_FP32_MFMA: 'fp32_mfma', ARITH_BF16X3: 'bf16x3', ARITH_BF16: 'bf16'}
ABI_VERSION = 3
ERR_ABI = -4
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_TANH = 0, 1, 2, 3
ACT_BY_NAME = {None: ACT_NONE, 'relu': ACT_RELU, 'leaky_relu': ACT_LEAKY_RELU, 'tanh': ACT_TANH}


class Sized(C.Structure):
    """An argument struct of ABI version 2 that starts with `struct_size` (t3d.h): filled in here, so positional construction
    keeps listing the fields behind it."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class ActSrc(C.Structure):
    _fields_ = [('x', F), ('ldx', i32), ('coff', i32), ('scale', F), ('shift', F), ('relu', i32),
                ('sub', F), ('sub_ld', i32), ('dtype', i32)]


class DySrc(C.Structure):
    _fields_ = [('dz', F), ('y', F), ('coef', F), ('argidx', I), ('dpool', F), ('dtype', i32)]


class PointMlpFwdArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('a', ActSrc), ('w', F), ('bias', F), ('rowbias', F), ('y', F), ('psum', F), ('psumsq', F),
                ('rowmask', F), ('pmax', F), ('pmin', F), ('pamax', I), ('pamin', I),
                ('M', i32), ('K', i32), ('N', i32), ('rows_per_frustum', i32), ('dtype', i32), ('w_x3', C.c_void_p), ('w_x3_stride', C.c_int64), ('arith', i32)]


class BnFwdFinalizeArgs(C.Structure):
    _fields_ = [('psum', F), ('psumsq', F), ('n_tiles', i32), ('count', i32), ('N', i32), ('gamma', F), ('beta', F),
                ('moving_mean', F), ('moving_var', F), ('decay', F), ('eps', f32), ('is_training', i32),
                ('unbiased_ema', i32), ('scale', F), ('shift', F), ('mean', F), ('invstd', F),
                ('pool_pmax', F), ('pool_pmin', F), ('pool_pamax', I), ('pool_pamin', I), ('pool_B', i32), ('pool_tiles_per_frustum', i32),
                ('pooled', F), ('ld_pooled', i32), ('argidx', I), ('ysel', F)]


class PoolFinalizeArgs(C.Structure):
    _fields_ = [('scale', F), ('shift', F), ('pmax', F), ('pmin', F), ('pamax', I), ('pamin', I),
                ('B', i32), ('N', i32), ('tiles_per_frustum', i32), ('pooled', F), ('ld_pooled', i32),
                ('argidx', I), ('ysel', F)]


class PointMlpDgradArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('dy', DySrc), ('w', F), ('add_in', F), ('prev_y', F), ('prev_scale', F), ('prev_shift', F),
                ('out', F), ('psum_dz', F), ('psum_dzy', F), ('M', i32), ('K', i32), ('N', i32),
                ('rows_per_frustum', i32), ('dtype', i32), ('w_x3', C.c_void_p), ('w_x3_stride', C.c_int64), ('arith', i32)]


class PointMlpWgradArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('a', ActSrc), ('dy', DySrc), ('slabs', F), ('M', i32), ('K', i32), ('N', i32),
                ('rows_per_frustum', i32), ('rows_per_split', i32), ('arith', i32)]


class PoolBwdPrepArgs(C.Structure):
    _fields_ = [('w', F), ('bias', F), ('coef', F), ('K', i32), ('N', i32), ('p_slabs', F), ('rc_slabs', F), ('wc', F)]


class PoolSparseRowsArgs(C.Structure):
    _fields_ = [('argidx', I), ('dpool', F), ('wc', F), ('B', i32), ('N', i32), ('K', i32), ('rows_per_frustum', i32), ('s', F),
                ('row_live', I)]


class PointMlpDgradGramArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('a', ActSrc), ('p', F), ('rowconst', F), ('add_in', F), ('add_live', I), ('prev_y', F), ('prev_scale', F),
                ('prev_shift', F),
                ('out', F), ('psum_dz', F), ('psum_dzy', F), ('M', i32), ('K', i32), ('rows_per_frustum', i32), ('dtype', i32), ('arith', i32)]


class PointMlpGramArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('a', ActSrc), ('slabs', F), ('M', i32), ('K', i32), ('rows_per_frustum', i32), ('rows_per_split', i32), ('arith', i32)]


class ActColsumArgs(C.Structure):
    _fields_ = [('a', ActSrc), ('M', i32), ('K', i32), ('rows_per_frustum', i32), ('part', F)]


class PoolWgradFinishArgs(C.Structure):
    _fields_ = [('a', ActSrc), ('argidx', I), ('dpool', F), ('coef', F), ('w', F), ('bias', F), ('g', F), ('abar', F),
                ('B', i32), ('K', i32), ('N', i32), ('rows_per_frustum', i32), ('dw', F)]


class BoxRefineStepArgs(C.Structure):
    _fields_ = [('out9', F), ('center_in', F), ('dims_in', F), ('theta_in', F), ('center_out', F), ('dims_out', F), ('theta_out', F),
                ('total', F), ('fit_prob', F), ('weigh_by_conf', i32), ('first', i32), ('B', i32)]


class ClassGroups(C.Structure):
    _fields_ = [('members', I), ('offsets', I), ('n_groups', i32), ('perm', I), ('perm_len', i32)]


class SampleEqualClassesArgs(C.Structure):
    _fields_ = [('set', ClassGroups * 2), ('B', i32), ('seed', C.c_uint32), ('hyper', F), ('order_draws', F), ('member_draws', F),
                ('equal_prob', f32), ('prob_draw', F), ('sample', I), ('is_data_2D', I)]


class BoxPcPerturbArgs(C.Structure):
    _fields_ = [('center', F), ('orient_cls', I), ('orient_reg', F), ('dims_cls', I), ('dims_reg', F), ('y_box_iou', F),
                ('y_center_delta', F), ('y_dims_delta', F), ('y_orient_delta', F), ('center_perturbation', f32),
                ('size_perturbation', f32), ('angle_perturbation', f32), ('fit_lo', f32), ('fit_hi', f32), ('nofit_lo', f32),
                ('nofit_hi', f32), ('proportion_fit', f32), ('fit_draw', F), ('cand_draws', F), ('max_rounds', i32),
                ('seed', C.c_uint32), ('hyper', F), ('B', i32)]


class Box3dIouArgs(C.Structure):
    _fields_ = [('center1', F), ('size1', F), ('heading1', F), ('center2', F), ('size2', F), ('heading2', F), ('iou3d', F), ('iou2d', F),
                ('n', i32)]


class Box3dIouCornersArgs(C.Structure):
    _fields_ = [('corners1', F), ('corners2', F), ('iou3d', F), ('iou2d', F), ('n', i32)]


class BoxHeadIouArgs(C.Structure):
    _fields_ = [('box', F), ('ld_box', i32), ('stage1_center', F), ('y_center', F), ('y_orient_cls', I), ('y_orient_reg', F),
                ('y_dims_cls', I), ('y_dims_reg', F), ('iou2d', F), ('iou3d', F), ('B', i32)]


class BatchAssembleArgs(C.Structure):
    _fields_ = [('points', F), ('seg', I), ('offsets', C.POINTER(C.c_int64)), ('frustum_angle', F), ('box_center', F), ('heading', F),
                ('size', F), ('cls', I), ('sample', I), ('sample_len', i32), ('choice', I), ('aug', F), ('C_src', i32), ('C', i32), ('B', i32), ('N', i32),
                ('rotate_to_center', i32), ('random_flip', i32), ('random_shift', i32), ('seed', C.c_uint32), ('hyper', F), ('pc', F),
                ('y_seg', I), ('y_center', F), ('y_orient_cls', I), ('y_orient_reg', F), ('y_dims_cls', I), ('y_dims_reg', F),
                ('one_hot', F), ('rot_angle', F), ('sample2', I), ('sample2_len', i32), ('is_data_2D', I), ('frustum_is_2D', I), ('ld_pc', i32),
                ('slot_is_2D', I), ('cam_rtilt', F), ('cam_k', F), ('cam_box2d', F), ('cam_img_dim', F), ('Rtilt', F), ('K', F), ('box2D', F),
                ('img_dim', F)]


class BnBwdFinalizeArgs(C.Structure):
    _fields_ = [('psum_dz', F), ('psum_dzy', F), ('n_tiles', i32), ('dpool_in', F), ('ld_dpool_in', i32),
                ('pooled', F), ('ld_pooled', i32), ('ysel', F), ('dpool', F), ('B', i32), ('count', i32), ('N', i32),
                ('gamma', F), ('mean', F), ('invstd', F), ('scale', F), ('frozen', i32), ('dgamma', F), ('dbeta', F),
                ('coef', F)]


class DyColsumArgs(C.Structure):
    _fields_ = [('psum_dz', F), ('psum_y', F), ('coef', F), ('B', i32), ('N', i32), ('tiles_per_frustum', i32),
                ('rows_per_frustum', i32), ('alpha', f32), ('out', F)]


class FcFwdArgs(C.Structure):
    _fields_ = [('in_', F), ('ld_in', i32), ('K', i32), ('in2', F), ('ld_in2', i32), ('K2', i32), ('w', F), ('bias', F),
                ('gamma', F), ('beta', F), ('moving_mean', F), ('moving_var', F), ('decay', F), ('eps', f32),
                ('is_training', i32), ('unbiased_ema', i32), ('act', i32), ('leaky_alpha', f32), ('drop_mask', F),
                ('keep_prob', f32), ('add_in', F), ('ld_add', i32), ('add_n', i32), ('y', F), ('out', F),
                ('ld_out', i32), ('mean', F), ('invstd', F), ('B', i32), ('N', i32)]


class FcBwdArgs(C.Structure):
    _fields_ = [('dout', F), ('ld_dout', i32), ('dy_next', F), ('w_next', F), ('N_next', i32),
                ('in_', F), ('ld_in', i32), ('K', i32), ('in2', F), ('ld_in2', i32), ('K2', i32),
                ('y', F), ('out', F), ('ld_out', i32), ('gamma', F), ('beta', F), ('mean', F), ('invstd', F),
                ('bn_training', i32), ('act', i32), ('leaky_alpha', f32), ('drop_mask', F), ('keep_prob', f32),
                ('dy', F), ('dw', F), ('dbias', F), ('dgamma', F), ('dbeta', F), ('B', i32), ('N', i32)]


class FcDinputArgs(C.Structure):
    _fields_ = [('dy', F), ('N', i32), ('w', F), ('add_in', F), ('ld_add', i32), ('alpha', f32), ('din', F),
                ('ld_din', i32), ('B', i32), ('K', i32),
                ('bn_pooled', F), ('bn_ld_pooled', i32), ('bn_ysel', F), ('bn_dpool', F), ('bn_count', i32), ('bn_gamma', F), ('bn_mean', F),
                ('bn_invstd', F), ('bn_scale', F), ('bn_frozen', i32), ('bn_dgamma', F), ('bn_dbeta', F), ('bn_coef', F)]


class SegHeadArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('y', F), ('scale', F), ('shift', F), ('drop_mask', F), ('keep_prob', f32), ('w', F), ('bias', F),
                ('labels', I), ('is_data_2D', I), ('pc', F), ('ld_pc', i32), ('ce_weight', f32), ('logits', F),
                ('mask', F), ('part', F), ('dz', F), ('psum_dz', F), ('psum_dzy', F), ('dw_part', F),
                ('M', i32), ('K', i32), ('rows_per_frustum', i32), ('B', i32), ('drop_seed', C.c_uint32), ('drop_hyper', F),
                ('dtype', i32), ('dsoft', F), ('oracle_mask', I)]


SMALL_KIND = {'t3d_bn_bwd_finalize': 1, 't3d_fc_bwd': 2, 't3d_fc_dinput': 3, 't3d_dy_colsum': 4}      # t3d.h T3D_SMALL_* (t3d_small_pair)
RIDER_KIND = dict(SMALL_KIND, t3d_bn_fwd_finalize=5, t3d_fc_fwd=6, t3d_pool_bwd_mid=7)                  # kinds a rider set takes
RIDER_FIELD = {1: 'bn_bwd', 2: 'fc_bwd', 3: 'fc_dinput', 4: 'dy_colsum', 5: 'bn_fwd', 6: 'fc_fwd', 7: 'mid'}
RIDER_WIDE = ('t3d_pool_bwd_mid',)      # hundreds of workgroups, alone in its set


class WeakLossArgs(C.Structure):
    _fields_ = [('center', F), ('reg_dims', F), ('reg_theta', F), ('pc', F), ('ld_pc', i32), ('logits', F), ('Rtilt', F), ('K', F),
                ('rot_frust', F), ('box2D', F), ('img_dim', F), ('is_data_2D', I), ('w_reproj', f32), ('w_surface', f32),
                ('multiplier', f32), ('use_softmax_proj', i32), ('softmax_scale', f32), ('dilate', f32), ('clip_lower_b_loss', i32),
                ('clip_pred_box', i32), ('loss_mse', i32), ('train_box_reproj', i32 * 3), ('train_box_surface', i32 * 3),
                ('surface_margin', f32), ('surface_scale_dims', f32), ('surf_part', F), ('dsoft', F), ('reproj', F), ('surface', F),
                ('dbox7', F), ('total_losses', F), ('loss', F), ('B', i32), ('N', i32), ('one_hot', F), ('w_inactive', f32),
                ('inactive_margins', f32 * 10), ('inactive_train', i32 * 10), ('inactive', F)]


class ActDropoutArgs(C.Structure):
    _fields_ = [('a', ActSrc), ('mask', F), ('keep_prob', f32), ('out', F), ('M', i32), ('K', i32), ('rows_per_frustum', i32)]


class SegFinalizeArgs(C.Structure):
    _fields_ = [('part', F), ('dw_part', F), ('B', i32), ('tiles_per_frustum', i32), ('rows_per_frustum', i32),
                ('K', i32), ('mask_xyz_mean', F), ('seg_loss', F), ('dw', F), ('dbias', F), ('n_correct', F)]


class StrongWeights(C.Structure):
    _fields_ = [(n, f32) for n in ('center', 'orient_cls', 'orient_reg', 'dims_cls', 'dims_reg', 'tnet_center',
                                   'corner', 'box_multiplier', 'cross_entropy')]


class StrongLossArgs(C.Structure):
    _fields_ = [('box', F), ('ld_box', i32), ('stage1_center', F), ('seg_loss', F), ('y_center', F),
                ('y_orient_cls', I), ('y_orient_reg', F), ('y_dims_cls', I), ('y_dims_reg', F), ('is_data_2D', I),
                ('wts', StrongWeights), ('normalize_by_3d_count', i32), ('dbox', F), ('dstage1', F), ('terms', F),
                ('total_losses', F), ('loss', F), ('center', F), ('reg_dims', F), ('reg_theta', F), ('iou2d', F), ('iou3d', F), ('B', i32)]


class X3FragEntry(C.Structure):      # t3d_x3_frag_entry (t3d_split_x3_frag)
    _fields_ = [('off', C.c_int64), ('K', C.c_int32), ('N', C.c_int32), ('fwd', C.c_int32), ('dgrad', C.c_int32), ('blk0', C.c_int32), ('reserved', C.c_int32)]


def x3_frag_table(entries):
    """(bytes of a t3d_x3_frag_entry table as a NumPy uint8 array, number of workgroups) for [(off, K, N), ...] (both arrangements)"""
    import numpy as np
    rec = np.zeros(len(entries), dtype=np.dtype([('off', '<i8'), ('K', '<i4'), ('N', '<i4'), ('fwd', '<i4'), ('dgrad', '<i4'), ('blk0', '<i4'), ('reserved', '<i4')]))
    blk = 0
    for i, (off, K, N) in enumerate(entries):
        rec[i] = (off, K, N, 1, 1, blk, 0)
        blk += (K * N // 8 + 255) // 256
    return rec.view(np.uint8).copy(), blk


class SlabDesc(C.Structure):
    _fields_ = [('slab_off', C.c_int64), ('grad_off', C.c_int64), ('numel', C.c_int32), ('n_slabs', C.c_int32)]


class Schedule(C.Structure):
    _fields_ = [('base_lr', f32), ('lr_decay_rate', f32), ('lr_decay_step', f32), ('bn_init_decay', f32),
                ('bn_decay_rate', f32), ('bn_decay_step', f32), ('bn_decay_clip', f32), ('beta1', f32), ('beta2', f32),
                ('batch_size', i32), ('step_offset', i32)]


class BoxPcRepArgs(Sized):
    _fields_ = [('struct_size', C.c_uint32), ('pc', F), ('ld_pc', i32), ('C', i32), ('center', F), ('dims', F), ('theta', F), ('y_dims_cls', I),
                ('y_orient_cls', I), ('rep', F), ('ld_rep', i32), ('box_out', F), ('M', i32), ('rows_per_frustum', i32), ('rowmask', F)]


class BoxPcRepBwdArgs(C.Structure):
    _fields_ = [('pc', F), ('ld_pc', i32), ('box', F), ('drep', F), ('ld_drep', i32), ('coff', i32), ('dbox', F),
                ('B', i32), ('rows_per_frustum', i32)]


class BoxPcLossArgs(C.Structure):
    _fields_ = [('out', F), ('y_box_iou', F), ('y_center_delta', F), ('y_dims_delta', F), ('y_orient_delta', F),
                ('fit_bound', f32), ('w_cls', f32), ('w_delta', f32), ('w_center', f32), ('w_size', f32), ('w_angle', f32),
                ('weigh_by_cls_conf', i32), ('weigh_by_cls_gt', i32), ('dout', F), ('terms', F), ('loss', F), ('B', i32),
                ('weigh_pred_by_cls_conf', i32), ('grad_cls_via_delta', i32), ('delta_loss_mse', i32)]


class BoxRefineStepBwdArgs(C.Structure):
    _fields_ = [('out9', F), ('dbox_rep', F), ('carry', F), ('tot_out', F), ('dout9', F), ('weigh_by_conf', i32),
                ('grad_via_conf', i32), ('B', i32)]


class Box2dFeatsArgs(C.Structure):
    _fields_ = [('one_hot', F), ('n_oh', i32), ('box2D', F), ('img_dim', F), ('out', F), ('B', i32)]


class DgradNarrowArgs(C.Structure):
    _fields_ = [('dy', DySrc), ('w', F), ('k0', i32), ('kn', i32), ('out', F), ('ld_out', i32), ('M', i32), ('N', i32)]


class SemiFinalLossArgs(C.Structure):
    _fields_ = [('strong_loss', F), ('reg_dims', F), ('one_hot', F), ('is_data_2D', I), ('out9', F),
                ('train_classes', C.c_int32 * 10), ('w_weak', f32), ('w_fit', f32), ('fit_only_2d', i32), ('d_dims', F),
                ('dout9', F), ('fit_prob', F), ('terms', F), ('loss', F), ('B', i32)]


class AnchorRegBwdArgs(C.Structure):
    _fields_ = [('box', F), ('ld_box', i32), ('dbox7', F), ('d_dims', F), ('dbox', F), ('dstage1', F), ('B', i32)]


VP = C.c_void_p
# name -> argtypes.  Struct entry points take (const args*, stream).

class PoolBwdMidArgs(C.Structure):
    _fields_ = [('slab_base', F), ('grad_base', F), ('table_dev', C.POINTER(SlabDesc)), ('n_tensors', i32), ('max_numel', i32),
                ('sparse', PoolSparseRowsArgs)]


class SmallOpU(C.Union):
    _fields_ = [('bn_bwd', BnBwdFinalizeArgs), ('fc_bwd', FcBwdArgs), ('fc_dinput', FcDinputArgs), ('dy_colsum', DyColsumArgs),
                ('bn_fwd', BnFwdFinalizeArgs), ('fc_fwd', FcFwdArgs), ('mid', PoolBwdMidArgs)]


class SmallOp(C.Structure):
    _fields_ = [('kind', i32), ('depends', i32), ('u', SmallOpU)]


RIDER_MAX_OPS = 10            # t3d.h T3D_RIDER_MAX_OPS


class RiderSet(C.Structure):
    _fields_ = [('ops', SmallOp * RIDER_MAX_OPS), ('n_ops', i32), ('n_wg', i32), ('lds_bytes', i32), ('sync', C.POINTER(C.c_uint32))]


ENTRY_POINTS = {
    't3d_abi_version': [],
    't3d_gemm_arithmetic': [i32, i32, i32, i32, i32],
    't3d_source_hash': [C.c_char_p, C.c_int],
    't3d_pointmlp_fwd': [C.POINTER(PointMlpFwdArgs), VP],
    't3d_bn_fwd_finalize': [C.POINTER(BnFwdFinalizeArgs), VP],
    't3d_pool_finalize': [C.POINTER(PoolFinalizeArgs), VP],
    't3d_pointmlp_dgrad': [C.POINTER(PointMlpDgradArgs), VP],
    't3d_pointmlp_wgrad': [C.POINTER(PointMlpWgradArgs), VP],
    't3d_pointmlp_bwd': [C.POINTER(PointMlpDgradArgs), C.POINTER(PointMlpWgradArgs), VP],
    't3d_wgrad_plan': [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    't3d_bwd_plan': [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)],
    't3d_weak_loss': [C.POINTER(WeakLossArgs), VP],
    't3d_small_pair': [C.POINTER(SmallOp), C.POINTER(SmallOp), VP],
    't3d_riders_plan': [C.POINTER(RiderSet)],
    't3d_run_riders': [C.POINTER(RiderSet), VP],
    't3d_pointmlp_fwd_r': [C.POINTER(PointMlpFwdArgs), C.POINTER(RiderSet), VP],
    't3d_pointmlp_wgrad_r': [C.POINTER(PointMlpWgradArgs), C.POINTER(RiderSet), VP],
    't3d_pointmlp_bwd_r': [C.POINTER(PointMlpDgradArgs), C.POINTER(PointMlpWgradArgs), C.POINTER(RiderSet), VP],
    't3d_pool_bwd_stage1_r': [C.POINTER(PointMlpGramArgs), C.POINTER(ActColsumArgs), C.POINTER(PoolBwdPrepArgs), C.POINTER(RiderSet), VP],
    't3d_pool_bwd_stage2_r': [C.POINTER(PoolWgradFinishArgs), C.POINTER(PointMlpDgradGramArgs), C.POINTER(RiderSet), VP],
    't3d_pointmlp_fwd_hosts_riders': [C.POINTER(PointMlpFwdArgs)],
    't3d_pointmlp_wgrad_hosts_riders': [C.POINTER(PointMlpWgradArgs)],
    't3d_pointmlp_bwd_hosts_riders': [C.POINTER(PointMlpDgradArgs), C.POINTER(PointMlpWgradArgs)],
    't3d_pool_bwd_stage1_hosts_riders': [C.POINTER(PointMlpGramArgs), C.POINTER(ActColsumArgs), C.POINTER(PoolBwdPrepArgs)],
    't3d_pool_bwd_stage2_hosts_riders': [C.POINTER(PoolWgradFinishArgs), C.POINTER(PointMlpDgradGramArgs)],
    't3d_gram_plan': [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)],
    't3d_bn_bwd_finalize': [C.POINTER(BnBwdFinalizeArgs), VP],
    't3d_dy_colsum': [C.POINTER(DyColsumArgs), VP],
    't3d_pool_bwd_prep': [C.POINTER(PoolBwdPrepArgs), VP],
    't3d_pool_sparse_rows': [C.POINTER(PoolSparseRowsArgs), VP],
    't3d_pointmlp_dgrad_gram': [C.POINTER(PointMlpDgradGramArgs), VP],
    't3d_pointmlp_gram': [C.POINTER(PointMlpGramArgs), VP],
    't3d_act_colsum': [C.POINTER(ActColsumArgs), VP],
    't3d_pool_wgrad_finish': [C.POINTER(PoolWgradFinishArgs), VP],
    't3d_batch_assemble': [C.POINTER(BatchAssembleArgs), VP],
    't3d_box_refine_step': [C.POINTER(BoxRefineStepArgs), VP],
    't3d_box_refine_step_bwd': [C.POINTER(BoxRefineStepBwdArgs), VP],
    't3d_boxpc_perturb': [C.POINTER(BoxPcPerturbArgs), VP],
    't3d_sample_equal_classes': [C.POINTER(SampleEqualClassesArgs), VP],
    't3d_box3d_iou': [C.POINTER(Box3dIouArgs), VP],
    't3d_box3d_iou_corners': [C.POINTER(Box3dIouCornersArgs), VP],
    't3d_box_head_iou': [C.POINTER(BoxHeadIouArgs), VP],
    't3d_box2d_feats': [C.POINTER(Box2dFeatsArgs), VP],
    't3d_pool_bwd_stage1': [C.POINTER(PointMlpGramArgs), C.POINTER(ActColsumArgs), C.POINTER(PoolBwdPrepArgs), VP],
    't3d_pool_bwd_mid': [F, F, C.POINTER(SlabDesc), i32, i32, C.POINTER(PoolSparseRowsArgs), VP],
    't3d_pool_bwd_stage2': [C.POINTER(PoolWgradFinishArgs), C.POINTER(PointMlpDgradGramArgs), VP],
    't3d_fc_fwd': [C.POINTER(FcFwdArgs), VP],
    't3d_fc_bwd': [C.POINTER(FcBwdArgs), VP],
    't3d_fc_dinput': [C.POINTER(FcDinputArgs), VP],
    't3d_seg_head': [C.POINTER(SegHeadArgs), VP],
    't3d_seg_finalize': [C.POINTER(SegFinalizeArgs), VP],
    't3d_act_dropout': [C.POINTER(ActDropoutArgs), VP],
    't3d_strong_loss': [C.POINTER(StrongLossArgs), VP],
    't3d_boxpc_rep': [C.POINTER(BoxPcRepArgs), VP],
    't3d_boxpc_rep_bwd': [C.POINTER(BoxPcRepBwdArgs), VP],
    't3d_boxpc_loss': [C.POINTER(BoxPcLossArgs), VP],
    't3d_pointmlp_dgrad_narrow': [C.POINTER(DgradNarrowArgs), VP],
    't3d_semi_final_loss': [C.POINTER(SemiFinalLossArgs), VP],
    't3d_anchor_reg_bwd': [C.POINTER(AnchorRegBwdArgs), VP],
    't3d_reduce_slabs': [F, F, C.POINTER(SlabDesc), i32, i32, VP],
    't3d_schedule_step': [F, C.POINTER(Schedule), VP],
    't3d_adam_tf_step': [F, F, F, F, C.c_int64, F, f32, f32, f32, f32, VP],
    't3d_momentum_step': [F, F, F, C.c_int64, F, f32, f32, VP],
    't3d_split_x3': [F, VP, C.c_int64, C.c_int64, VP],
    't3d_split_x3_frag': [F, VP, VP, C.c_int64, VP, i32, i32, VP],
    't3d_dropout_mask': [F, C.c_int64, f32, C.c_uint32, F, VP],
    't3d_cast_bf16': [F, VP, C.c_int64, VP],
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libt3d.so')
ERRORS = {-1: 'T3D_ERR_ARG', -2: 'T3D_ERR_SHAPE', -3: 'T3D_ERR_LAUNCH'}


class T3DError(RuntimeError):
    pass


def source_hash_of(lib):
    """The source hash compiled into a loaded library (csrc/version.hip)."""
    buf = C.create_string_buffer(64)
    check(lib.t3d_source_hash(buf, 64), 't3d_source_hash')
    return buf.value.decode()


def load(path=None):
    """dlopen libt3d.so and bind every entry point of include/t3d.h; raises if anything is missing, and if the library was built
    from other sources than the ones lying next to it (T3D_ALLOW_STALE_LIB=1: experiments only)."""
    # torch bundles its own libamdhip64: import it FIRST so that libt3d.so binds to the same HIP runtime
    # instance that owns torch's streams and allocations (two runtimes in one process cannot share them).
    import torch  # noqa: F401
    variant = path or os.environ.get('T3D_LIB')                # T3D_LIB: an alternative build (tools/build_variant.sh) for same-box A/B
    path = variant or LIB_PATH
    if not os.path.exists(path):
        raise T3DError('HIP library %s not built: run `python -m transferable3d_amd.build` (no CPU fallback exists)' % path)
    lib = C.CDLL(path)
    for name, argtypes in ENTRY_POINTS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise T3DError('%s does not export %s' % (path, name))
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.t3d_abi_version() != ABI_VERSION:      # every library, the variants of the tools included
        raise T3DError('%s speaks ABI version %d, this host side version %d (include/t3d.h): rebuild it' % (path, lib.t3d_abi_version(), ABI_VERSION))
    from .build import CSRC, lib_source_hash
    if not variant and os.path.isdir(CSRC) and os.environ.get('T3D_ALLOW_STALE_LIB', '0') != '1':
        built, src = source_hash_of(lib), lib_source_hash()
        if built != src:
            raise T3DError('%s was built from other sources (library %s, csrc/ + include/t3d.h %s): run '
                           '`python -m transferable3d_amd.build`' % (path, built, src))
    return lib


def fptr(t):
    """float* of a tensor (None -> NULL)."""
    return C.cast(C.c_void_p(0 if t is None else t.data_ptr()), F)


def iptr(t):
    return C.cast(C.c_void_p(0 if t is None else t.data_ptr()), I)


def check(rc, what):
    if rc != 0:
        raise T3DError('%s failed: %s' % (what, ERRORS.get(rc, rc)))

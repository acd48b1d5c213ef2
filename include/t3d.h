/*
 * t3d.h -- C ABI of the MI355X-native Frustum-PointNet training path (libt3d.so).
 *
 * The reference (yewsiang/Transferable3D) has no FFI layer: its hot path is stock TensorFlow-1 ops
 * reached through the Python wrappers in models/tf_util.py.  Each entry point below therefore cites
 * the reference call site(s) whose arithmetic it replaces; the Python side of the boundary
 * (transferable3d_amd/tf_util.py, semisup_models.py, ...) keeps the reference's function names.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, `int` return: 0 = T3D_OK, negative = T3D_ERR_*.
 *   - every buffer is a caller-owned DEVICE pointer (the library never allocates or frees);
 *     scratch (partials, slabs) is passed in explicitly.
 *   - all matrices are row-major, channel fastest: a (B, N, C) point tensor is an (M = B*N, C)
 *     matrix, exactly the reference's NHWC layout with H = N, W = 1.
 *   - `stream` is a hipStream_t; kernels are enqueued, never synchronised.  No global state.
 *   - "partials" are per-128-row-tile column reductions ([M/128, C], T3D_TILE_ROWS rows each) that a
 *     finalize kernel combines deterministically (no float atomics anywhere on the path).
 *   - M must be a multiple of T3D_TILE_ROWS and rows_per_frustum (the point count N) a multiple of
 *     T3D_TILE_ROWS, so that a tile never straddles two frustums.
 */
#ifndef T3D_H_
#define T3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T3D_OK 0
#define T3D_ERR_ARG (-1)     /* null / inconsistent argument */
#define T3D_ERR_SHAPE (-2)   /* unsupported shape (alignment, divisibility) */
#define T3D_ERR_LAUNCH (-3)  /* HIP launch failure */
#define T3D_ERR_ABI (-4)     /* an argument struct of another size than this library was built for (see `struct_size`) */

/* ABI versions 2 and 3.  Version 1 structs were plain; fields appended to four of them in round 4 (w_x3, oracle_mask, rowmask) made a caller
 * built against the older header pass structs the library read past.  Since version 2 every argument struct that has grown, or may
 * grow, starts with `struct_size`: the caller stores sizeof(the struct as ITS header declares it) there.  New fields are only ever
 * appended, and 0 is the documented default of every appended field.  An entry point that takes such a struct accepts
 *     struct_size == the library's sizeof          the same header;
 *     T3D_V2_SIZE_<struct> <= struct_size < sizeof  an OLDER caller: the fields it does not know read 0 (csrc/abi_take.h copies the
 *                                                   caller's bytes into a zeroed struct of the library's own declaration);
 *     struct_size > sizeof, every byte beyond 0     a NEWER caller that uses none of the fields this library does not know;
 * and returns T3D_ERR_ABI for anything else (a struct shorter than its version-2 size; a newer caller's non-zero unknown field).
 * (Until round 5 the test was equality: every append would have refused every older caller.)  The structs WITHOUT `struct_size` are
 * frozen: they never grow -- a change to one of them is a new struct behind a new entry point. */
#define T3D_ABI_VERSION 3      /* 3: `w_x3` means fragment-order planes (t3d_split_x3_frag); layouts and sizes as version 2 */
/* sizeof of the growable structs at ABI version 2 (the smallest struct_size an entry point accepts) */
#define T3D_V2_SIZE_pointmlp_fwd_args 200
#define T3D_V2_SIZE_pointmlp_dgrad_args 168
#define T3D_V2_SIZE_pointmlp_wgrad_args 144
#define T3D_V2_SIZE_pointmlp_dgrad_gram_args 168
#define T3D_V2_SIZE_pointmlp_gram_args 96
#define T3D_V2_SIZE_seg_head_args 208
#define T3D_V2_SIZE_boxpc_rep_args 104

#define T3D_TILE_ROWS 128

typedef void* t3d_stream_t;

enum { T3D_ACT_NONE = 0, T3D_ACT_RELU = 1, T3D_ACT_LEAKY_RELU = 2, T3D_ACT_TANH = 3 };

/* Element type of the per-point [M, C] layer tensors (raw conv outputs y, gradients dz) and, with it, the arithmetic of the GEMM
 * that produces or consumes them (BASELINE.json configs[1..3] vs configs[4]):
 *   T3D_F32   fp32 storage, v_mfma_f32_32x32x2_f32 (exact fp32)
 *   T3D_BF16  bf16 storage (2 bytes per element in HBM), operands rounded to bf16 while they are staged into LDS,
 *             v_mfma_f32_32x32x16_bf16 with fp32 accumulation; statistics, partials, weights, weight gradients, the sparse
 *             arg-max rows, the per-frustum [B, C] tensors, the optimiser and the raw inputs (point cloud, Box-PC
 *             representation) stay fp32.
 * Every `dtype` field below takes one of these; 0 (fp32) is what a zero-initialised struct means. */
enum { T3D_F32 = 0, T3D_BF16 = 1 };

/* Arithmetic of an fp32 (dtype = T3D_F32) per-point GEMM launch -- the `arith` field of the GEMM argument structs:
 *   T3D_ARITH_FP32_MFMA  v_mfma_f32_32x32x2_f32: the fp32 fma chain, 157 TFLOP/s peak
 *   T3D_ARITH_BF16X3     every fp32 operand as the exact sum of three bf16 terms, six v_mfma_f32_32x32x16_bf16 per multiply-add with
 *                        fp32 accumulation (csrc/pointmlp.hip, "fp32 GEMMs on the bf16 matrix pipe"); a launch whose shape has no such
 *                        kernel, or for which the launcher's rule prefers the fp32 MFMA (t3d_gemm_arithmetic says which), takes that.
 *                        Two differences from the fp32 MFMA a caller should know: an operand of magnitude >= 3.39e38 (above bf16's
 *                        largest finite value; fp32 itself reaches 3.40e38) has an infinite leading term and yields inf / NaN where
 *                        the fma chain would still be finite; and the bf16 MFMA's accumulation is not rounded to nearest -- results
 *                        lie a little below the exact sum (about -3e-7 of the mean |dW| on a 32768-row weight gradient, bounded in
 *                        tests/test_kernels_gpu.py), where the fma chain shows no offset
 *   T3D_ARITH_AUTO       (0, a zero-initialised struct) the library's default = T3D_ARITH_BF16X3; the only value for which the
 *                        experiment variables T3D_X3 / T3D_X3_MINKN of the tools are consulted
 * A host fixes the value when it builds its plan (transferable3d_amd.engine.Runtime.gemm_arithmetic), so a captured graph, an eager
 * launch and the line bench.py prints cannot disagree.  Ignored by T3D_BF16 launches. */
enum { T3D_ARITH_AUTO = 0, T3D_ARITH_FP32_MFMA = 1, T3D_ARITH_BF16X3 = 2, T3D_ARITH_BF16 = 3 };
/* The arithmetic a per-point GEMM launch of this request takes: T3D_ARITH_FP32_MFMA, T3D_ARITH_BF16X3 or T3D_ARITH_BF16 (dtype =
 * T3D_BF16).  K x N: the layer's weight matrix (a Gram-form launch: N = K); backward != 0: a data / weight gradient launch. */
/* `kind`: which launcher is asked (0 and 1 are what round 5's `backward` flag meant) */
enum { T3D_GEMM_FWD = 0, T3D_GEMM_BWD = 1, T3D_GEMM_DGRAD = 2, T3D_GEMM_WGRAD = 3, T3D_GEMM_GRAM = 4, T3D_GEMM_DGRAD_GRAM = 5 };
int t3d_gemm_arithmetic(int arith, int dtype, int K, int N, int kind);

int t3d_abi_version(void);
/* Hash (16 hex digits + NUL) over the HIP sources and this header the library was built from (csrc/version.hip; "unknown" for a
 * build outside transferable3d_amd/build.py).  `cap` >= 17.  The host side refuses a library whose hash differs from the sources
 * next to it (abi.load). */
int t3d_source_hash(char* out, int cap);

/* ---- operand descriptors ------------------------------------------------------------------ */

/* A per-point activation operand produced lazily from the RAW output of the previous layer:
 *   a[m,k] = relu?( x[m, coff+k] * scale[k] + shift[k] ) - sub[b(m), k]
 * scale/shift (batch-norm apply, tf_util.py:1316-1322) and sub (per-frustum recentring,
 * semisup_models.py:159,207) are optional (NULL).  ldx and coff must be multiples of 4 (of 8 for a T3D_BF16 source: 16-byte reads). */
typedef struct {
  const float* x;
  int ldx;
  int coff;
  const float* scale;
  const float* shift;
  int relu;
  const float* sub;      /* [B, sub_ld] */
  int sub_ld;
  int dtype;             /* element type of x (T3D_F32 / T3D_BF16: `x` then points at bf16 elements; ldx, coff in elements) */
} t3d_act_src;

/* The gradient w.r.t. a layer's RAW conv output, produced lazily while loading:
 *   dy[m,n] = coef[0,n]*dz[m,n] + coef[1,n]*y[m,n] + coef[2,n]
 * (training-mode batch-norm backward folded into three per-channel constants by
 * t3d_bn_bwd_finalize).  dz is either dense [M,N] or, for a max-pooled layer, sparse:
 * dz[m,n] = dpool[b,n] if argidx[b,n] == m - b*rows_per_frustum else 0. */
typedef struct {
  const float* dz;        /* dense [M,N], or NULL for the pooled-sparse form */
  const float* y;         /* [M,N] raw conv output of this layer */
  const float* coef;      /* [3,N] */
  const int32_t* argidx;  /* [B,N] (pooled-sparse form) */
  const float* dpool;     /* [B,N] (pooled-sparse form) */
  int dtype;              /* element type of dz and y; T3D_BF16 supports the dense form only (pooled layers: K11e) */
} t3d_dy_src;

/* ---- K1: per-point shared-MLP layer, forward ---------------------------------------------------
 * Replaces tf_util.conv2d's tf.nn.conv2d(1x1 | [1,D], VALID) + bias_add (tf_util.py:1308-1314) at
 * semisup_models.py:76-95,115-130 (seg), 172-183 (T-Net), 224-239 (box), 354-369 (Box-PC); the
 * batch-norm + ReLU of the PREVIOUS layer is applied while loading `a` (t3d_act_src), this layer's
 * batch-norm statistics are reduced in the epilogue, and conv6's tile+concat of the global feature
 * (semisup_models.py:107-108) enters as the per-frustum `rowbias` (= global . W[64:]).
 *   y[m,n] = sum_k a[m,k] w[k,n] + bias[n] + rowbias[b(m), n]
 *   psum[t,n] = sum_{m in tile t} y[m,n];  psumsq[t,n] = sum y^2
 * Optional max-pool partials over the rows with rowmask != 0 (tf_util.max_pool2d over N after the
 * mask multiply, semisup_models.py:96,184-188,240-244,375): per tile the max and min of raw y and
 * their row indices within the frustum (the finalize kernel picks max or min by the sign of the
 * batch-norm scale). */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_pointmlp_fwd_args) of the caller's header (see T3D_ABI_VERSION) */
  t3d_act_src a;
  const float* w;        /* [K,N] */
  const float* bias;     /* [N] or NULL */
  const float* rowbias;  /* [B,N] or NULL */
  float* y;              /* [M,N], or NULL: statistics and pool partials only (Gram-form backward) */
  float* psum;           /* [M/128, N] */
  float* psumsq;         /* [M/128, N] */
  const float* rowmask;  /* [M] or NULL (pool over all rows) */
  float* pmax;           /* [M/128, N] or NULL (no pooling) */
  float* pmin;
  int32_t* pamax;
  int32_t* pamin;
  int M, K, N;
  int rows_per_frustum;
  int dtype;             /* element type of y and arithmetic of the GEMM (a.dtype may still be T3D_F32: the raw inputs) */
  /* optional (fp32 layers on the three-term bf16 path): the same [K,N] matrix already split into three bf16 planes IN FRAGMENT ORDER,
   * FORWARD arrangement (t3d_split_x3_frag: planes_fwd + the matrix's offset), plane p at w_x3 + p * w_x3_stride (bf16 elements).
   * The kernel then reads the weight operand of every MFMA straight from these planes (one 16-byte load per lane), without an LDS
   * image.  NULL: the kernel splits w while it stages it through LDS -- same results bit for bit.  (ABI version 3: until version 2 the
   * planes were row-major copies of w, t3d_split_x3.) */
  const void* w_x3;
  int64_t w_x3_stride;
  int arith;               /* T3D_ARITH_* (fp32 launches) */
} t3d_pointmlp_fwd_args;
int t3d_pointmlp_fwd(const t3d_pointmlp_fwd_args* args, t3d_stream_t stream);

/* ---- K2: batch-norm statistics -> per-channel scale/shift (+ EMA) ------------------------------
 * Replaces tf.contrib.layers.batch_norm (tf_util.py:1660-1664; eps 1e-3, updates_collections=None).
 * Training: mean/biased variance from the partials, scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale, moving <- moving*decay + batch*(1-decay) (variance Bessel-corrected
 * when `unbiased_ema`).  Eval: scale/shift from the moving statistics.  `decay` is a DEVICE scalar
 * (it follows the step counter under graph replay). */
typedef struct {
  const float* psum;
  const float* psumsq;
  int n_tiles;
  int count;             /* rows reduced = M */
  int N;
  const float* gamma;
  const float* beta;
  float* moving_mean;
  float* moving_var;
  const float* decay;    /* device scalar */
  float eps;
  int is_training;
  int unbiased_ema;
  float* scale;          /* [N] out */
  float* shift;          /* [N] out */
  float* mean;           /* [N] out (saved for backward) */
  float* invstd;         /* [N] out */
  /* optional: K3 (t3d_pool_finalize) for the same channels in the same launch; pool_pmax == NULL: none */
  const float* pool_pmax; const float* pool_pmin; const int32_t* pool_pamax; const int32_t* pool_pamin;
  int pool_B, pool_tiles_per_frustum;
  float* pooled; int ld_pooled;      /* [B, ld_pooled] */
  int32_t* argidx;                   /* [B,N] */
  float* ysel;                       /* [B,N] */
} t3d_bn_fwd_finalize_args;
int t3d_bn_fwd_finalize(const t3d_bn_fwd_finalize_args* args, t3d_stream_t stream);

/* ---- K3: masked global max-pool finalize -------------------------------------------------------
 * pooled[b,n] = max_m mask*relu(scale*y+shift)  (tf_util.py:1519-1523 after semisup_models.py:185).
 * argidx[b,n] = row (within the frustum) that receives the gradient, -1 when pooled == 0. */
typedef struct {
  const float* scale;
  const float* shift;
  const float* pmax;
  const float* pmin;
  const int32_t* pamax;
  const int32_t* pamin;
  int B, N, tiles_per_frustum;
  float* pooled;         /* [B, ld_pooled] */
  int ld_pooled;
  int32_t* argidx;       /* [B,N] */
  float* ysel;           /* [B,N] raw y at argidx */
} t3d_pool_finalize_args;
int t3d_pool_finalize(const t3d_pool_finalize_args* args, t3d_stream_t stream);

/* ---- K11a: per-point layer, data gradient ------------------------------------------------------
 * The dgrad twin of K1 (autodiff of tf_util.conv2d, train_semisup.py:249):
 *   da[m,k] = sum_n dy[m,n] w[k,n]  (+ add_in[m,k])
 * and, when the producer of `a` has batch-norm + ReLU (prev_y != NULL), the ReLU mask and the
 * producer's batch-norm-backward partials in the epilogue:
 *   out = da * 1[prev_y*prev_scale + prev_shift > 0];  psum_dz = sum out;  psum_dzy = sum out*prev_y */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_pointmlp_dgrad_args) of the caller's header (see T3D_ABI_VERSION) */
  t3d_dy_src dy;
  const float* w;          /* [K,N] */
  const float* add_in;     /* [M,K] or NULL */
  const float* prev_y;     /* [M,K] or NULL */
  const float* prev_scale; /* [K] */
  const float* prev_shift; /* [K] */
  float* out;              /* [M,K] */
  float* psum_dz;          /* [M/128,K] or NULL */
  float* psum_dzy;         /* [M/128,K] or NULL */
  int M, K, N;
  int rows_per_frustum;
  int dtype;               /* element type of prev_y, out and add_in, and the arithmetic; must equal dy.dtype */
  const void* w_x3;        /* optional: w as three bf16 fragment-order planes, DATA-GRADIENT arrangement (t3d_split_x3_frag: planes_dgrad + offset) */
  int64_t w_x3_stride;
  int arith;               /* T3D_ARITH_* (fp32 launches) */
} t3d_pointmlp_dgrad_args;
int t3d_pointmlp_dgrad(const t3d_pointmlp_dgrad_args* args, t3d_stream_t stream);

/* ---- K11b: per-point layer, weight gradient (split over rows) ----------------------------------
 *   slab[s,k,n] = sum_{m in split s} a[m,k] dy[m,n],  s = 0 .. M/rows_per_split - 1
 * t3d_reduce_slabs sums the slabs in a fixed order (deterministic). rows_per_split % 32 == 0. */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_pointmlp_wgrad_args) of the caller's header (see T3D_ABI_VERSION) */
  t3d_act_src a;
  t3d_dy_src dy;
  float* slabs;            /* [M/rows_per_split, K, N] */
  int M, K, N;
  int rows_per_frustum;
  int rows_per_split;      /* the arithmetic follows dy.dtype (bf16: rows_per_split % 64 == 0); slabs are fp32 either way */
  int arith;               /* T3D_ARITH_* (fp32 launches) */
} t3d_pointmlp_wgrad_args;
int t3d_pointmlp_wgrad(const t3d_pointmlp_wgrad_args* args, t3d_stream_t stream);
/* K11a + K11b of one dense layer in ONE launch (the two are independent; sharing a launch saves a kernel's fill/drain
 * latency per layer and lets both kinds of tile share the CUs).  Same arguments and results as the two separate calls;
 * dense dy only (dy.dz != NULL in both), same M, K, N. */
int t3d_pointmlp_bwd(const t3d_pointmlp_dgrad_args* dgrad, const t3d_pointmlp_wgrad_args* wgrad, t3d_stream_t stream);
/* Recommended row split (and the tile the launcher will then use) for a K x N weight gradient over M rows. */
int t3d_wgrad_plan(int M, int K, int N, int* rows_per_split, int* tile_k, int* tile_n);
/* Recommended row split for t3d_pointmlp_bwd.  bf16 layers (dtype = T3D_BF16 for dy, the input and the output) with K and N in
 * {64, 128}, or K x N = 256 x 128 / 128 x 256, run ONE-PASS (*one_pass = 1): one workgroup per split walks its 128-row tiles, reads dz, y and the input once, keeps
 * dW in registers across the tiles and writes one slab at the end; t3d_pointmlp_bwd takes that form whenever the shape is eligible
 * and M / rows_per_split >= min(256, M / 128).  Everything else: t3d_wgrad_plan's split (*one_pass = 0).  The results are those of
 * the two separate calls up to the fp32 summation order.  fp32 layers with K, N in {64, 128} have a one-pass form too (64-row tiles); it is
 * planned when a workgroup gets at least 256 rows (M >= 65536) or when M < 32768 -- at M = 32768 the split form ties and stays. */
int t3d_bwd_plan(int M, int K, int N, int dtype, int* rows_per_split, int* one_pass);

/* ---- K11c: batch-norm backward statistics -> dgamma, dbeta and the three dy coefficients -------
 * dense form: from the dgrad epilogue partials.  pooled form (psum_dz == NULL): from the gradient
 * of the pooled feature, dpool_in[B,N]; writes the ReLU-masked dpool[B,N] the sparse dy form reads.
 * `frozen` (eval-mode batch-norm, stage c): dy = scale*dz, no batch terms, no dgamma/dbeta. */
typedef struct {
  const float* psum_dz;
  const float* psum_dzy;
  int n_tiles;
  const float* dpool_in;   /* [B, ld_dpool_in] (pooled form) */
  int ld_dpool_in;
  const float* pooled;     /* [B, ld_pooled] */
  int ld_pooled;
  const float* ysel;       /* [B,N] */
  float* dpool;            /* [B,N] out */
  int B;
  int count;               /* M */
  int N;
  const float* gamma;
  const float* mean;
  const float* invstd;
  const float* scale;      /* used when frozen */
  int frozen;
  float* dgamma;           /* [N] or NULL */
  float* dbeta;            /* [N] or NULL */
  float* coef;             /* [3,N] */
} t3d_bn_bwd_finalize_args;
int t3d_bn_bwd_finalize(const t3d_bn_bwd_finalize_args* args, t3d_stream_t stream);

/* Per-frustum column sums of a dense dy, from partials only:
 *   out[b,n] = alpha * ( coef0*sum_dz + coef1*sum_y + coef2*rows_per_frustum )
 * (gradient of a per-frustum broadcast: conv6's global feature, semisup_models.py:107-108, and
 * box_est's `xyz - stage1_center`, semisup_models.py:207). */
typedef struct {
  const float* psum_dz;    /* [M/128,N] */
  const float* psum_y;     /* [M/128,N] forward psum of the same layer */
  const float* coef;       /* [3,N] */
  int B, N, tiles_per_frustum, rows_per_frustum;
  float alpha;
  float* out;              /* [B,N] */
} t3d_dy_colsum_args;
int t3d_dy_colsum(const t3d_dy_colsum_args* args, t3d_stream_t stream);

/* ---- K11e: backward of a max-pooled layer in "Gram form" ------------------------------------------
 * The three pooled layers (seg conv5 128->1024, T-Net conv3 128->256, box conv4 256->512;
 * semisup_models.py:92-96, 180-188, 236-244) are the widest of their nets.  Their dy is
 *   dy = c0*dz_sparse + c1*y + c2   with  y = a.w + bias  and dz_sparse non-zero in one row per (b,n),
 * so both gradients factor through the K x K Gram matrix of the layer INPUT and never touch [M,N]:
 *   da = a.P + rowconst + S          P = w diag(c1) w^T,  rowconst = w.(c1*bias + c2)
 *        S[m,:] = sum_{n : argidx[b,n] = m - b*rpf} dpool[b,n] * wc[n,:],   wc[n,k] = c0[n]*w[k,n]
 *   dw = c1*(G.w + abar (x) bias) + abar (x) c2 + c0 * sum_b dpool[b,n] * a[b*rpf + argidx[b,n], :]
 *        G = a^T a,  abar = column sums of a
 * which replaces 4*M*K*N flops by 4*M*K*K and lets the forward pass skip the [M,N] store (y = NULL
 * in t3d_pointmlp_fwd).  Same autodiff semantics as K11a/K11b (train_semisup.py:249). */
typedef struct {
  const float* w;          /* [K,N] */
  const float* bias;       /* [N] or NULL */
  const float* coef;       /* [3,N] from t3d_bn_bwd_finalize */
  int K, N;
  float* p_slabs;          /* [ceil(N/128), K, K] out: partial P per chunk of 128 columns n (sum with t3d_reduce_slabs) */
  float* rc_slabs;         /* [ceil(N/128), K] out: partial rowconst */
  float* wc;               /* [N,K] out, or NULL */
} t3d_pool_bwd_prep_args;
int t3d_pool_bwd_prep(const t3d_pool_bwd_prep_args* args, t3d_stream_t stream);

typedef struct {
  const int32_t* argidx;   /* [B,N] row within the frustum, -1 = no gradient */
  const float* dpool;      /* [B,N] */
  const float* wc;         /* [N,K] */
  int B, N, K;
  int rows_per_frustum;
  float* s;                /* [B*rows_per_frustum, K] out (every row written when row_live is NULL) */
  int32_t* row_live;       /* [B*rows_per_frustum] out or NULL: 1 where the row received a hit, else 0.  When given, rows of `s`
                            * without a hit are NOT written (a few percent of the rows carry arg-max hits; the dense zero rows
                            * were most of this kernel's HBM traffic) and the reader must gate on it (add_live below). */
} t3d_pool_sparse_rows_args;
int t3d_pool_sparse_rows(const t3d_pool_sparse_rows_args* args, t3d_stream_t stream);

/* out = (a.p + rowconst + add_in) * 1[prev_y*prev_scale + prev_shift > 0], with the producer's
 * batch-norm-backward partials, exactly like the epilogue of K11a. */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_pointmlp_dgrad_gram_args) of the caller's header (see T3D_ABI_VERSION) */
  t3d_act_src a;           /* [M,K] input of the pooled layer */
  const float* p;          /* [K,K] */
  const float* rowconst;   /* [K] or NULL */
  const float* add_in;     /* [M,K] or NULL (the sparse rows S) */
  const int32_t* add_live; /* [M] or NULL: add_in is read only on rows whose flag is non-zero (row_live of t3d_pool_sparse_rows) */
  const float* prev_y;     /* [M,K] or NULL */
  const float* prev_scale;
  const float* prev_shift;
  float* out;              /* [M,K] */
  float* psum_dz;          /* [M/128,K] or NULL */
  float* psum_dzy;
  int M, K;
  int rows_per_frustum;
  int dtype;               /* element type of prev_y and out, and the arithmetic (a.dtype gives the operand's); add_in with
                            * add_live (the sparse rows S) is fp32 either way, a dense add_in has this type */
  int arith;               /* T3D_ARITH_* (fp32 launches) */
} t3d_pointmlp_dgrad_gram_args;
int t3d_pointmlp_dgrad_gram(const t3d_pointmlp_dgrad_gram_args* args, t3d_stream_t stream);

/* slab[s] = a_s^T a_s over the rows of split s; rows_per_split from t3d_wgrad_plan(M, K, K). */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_pointmlp_gram_args) of the caller's header (see T3D_ABI_VERSION) */
  t3d_act_src a;
  float* slabs;            /* [M/rows_per_split, K, K] */
  int M, K;
  int rows_per_frustum;
  int rows_per_split;      /* the arithmetic follows a.dtype */
  int arith;               /* T3D_ARITH_* (fp32 launches) */
} t3d_pointmlp_gram_args;
int t3d_pointmlp_gram(const t3d_pointmlp_gram_args* args, t3d_stream_t stream);

/* Recommended row split of the Gram slabs.  bf16 inputs with K = 128 or 256 take the ONE-PASS kernels (*one_pass = 1): t3d_pointmlp_gram
 * / stage 1 with one workgroup per split (the input read and activated once, the per-tile column sums of t3d_act_colsum from the same
 * pass) whenever M / rows_per_split >= min(256, M / 128); t3d_pointmlp_dgrad_gram / stage 2 likewise when prev_y is the input
 * tensor itself (a.x, dense rows) and add_in is absent or the sparse form (add_live).  Otherwise t3d_wgrad_plan(M, K, K)'s split. */
int t3d_gram_plan(int M, int K, int dtype, int* rows_per_split, int* one_pass);

/* part[t,k] = sum over the 128 rows of tile t of a[m,k]. */
typedef struct {
  t3d_act_src a;
  int M, K;
  int rows_per_frustum;
  float* part;             /* [M/128, K] */
} t3d_act_colsum_args;
int t3d_act_colsum(const t3d_act_colsum_args* args, t3d_stream_t stream);

typedef struct {
  t3d_act_src a;
  const int32_t* argidx;   /* [B,N] */
  const float* dpool;      /* [B,N] */
  const float* coef;       /* [3,N] */
  const float* w;          /* [K,N] */
  const float* bias;       /* [N] or NULL */
  const float* g;          /* [K,K] reduced Gram matrix */
  const float* abar;       /* [K] column sums of a (t3d_act_colsum partials, reduced) */
  int B, K, N;
  int rows_per_frustum;
  float* dw;               /* [K,N] out */
} t3d_pool_wgrad_finish_args;
int t3d_pool_wgrad_finish(const t3d_pool_wgrad_finish_args* args, t3d_stream_t stream);

/* The K11e launches that are independent of each other, fused (same arguments and results as the separate calls):
 * stage 1 = t3d_pointmlp_gram + t3d_act_colsum + t3d_pool_bwd_prep; after the slab reduction,
 * stage 2 = t3d_pool_wgrad_finish + t3d_pointmlp_dgrad_gram. */
int t3d_pool_bwd_stage1(const t3d_pointmlp_gram_args* gram, const t3d_act_colsum_args* colsum,
                        const t3d_pool_bwd_prep_args* prep, t3d_stream_t stream);
int t3d_pool_bwd_stage2(const t3d_pool_wgrad_finish_args* finish, const t3d_pointmlp_dgrad_gram_args* dgrad,
                        t3d_stream_t stream);

/* ---- K6: per-frustum fully-connected layer -----------------------------------------------------
 * Replaces tf_util.fully_connected (tf_util.py:1463-1499: matmul + bias [+ batch_norm over the B
 * rows] [+ activation]) and the tf_util.dropout that follows it in mlps_with_dropout /
 * combined_box_pc_mask_features_model (semisup_models.py:56-62, 385-390).
 *   y = [in | in2] . w + bias ; z = BN(y) ; out = dropout(act(z)) (+ add_in on the first add_n cols)
 * drop_mask holds 0/1 keep flags; out *= mask/keep_prob. */
typedef struct {
  const float* in;  int ld_in;  int K;
  const float* in2; int ld_in2; int K2;      /* optional concat (one_hot_vec), K2 = 0 if none */
  const float* w;                            /* [K+K2, N]; NULL = identity (K == N, K2 == 0): y = in, i.e. a standalone
                                              * tf_util.batch_norm_for_fc (tf_util.py:1666-1677) / tf_util.dropout (1720-1741) node */
  const float* bias;                         /* [N] or NULL */
  const float* gamma; const float* beta;     /* NULL -> no batch-norm */
  float* moving_mean; float* moving_var;
  const float* decay;                        /* device scalar */
  float eps; int is_training; int unbiased_ema;
  int act; float leaky_alpha;
  const float* drop_mask; float keep_prob;   /* [B,N] or NULL */
  const float* add_in; int ld_add; int add_n;
  float* y;                                  /* [B,N] raw (pre-BN), may be NULL when no BN */
  float* out; int ld_out;                    /* [B, ld_out] */
  float* mean; float* invstd;                /* [N] saved for backward */
  int B, N;
} t3d_fc_fwd_args;
int t3d_fc_fwd(const t3d_fc_fwd_args* args, t3d_stream_t stream);

/* Backward of one fully-connected layer for all of its columns.  The incoming gradient w.r.t. `out`
 * is either `dout` or formed on the fly as dy_next . w_next^T (the next layer's input gradient).
 * Writes dy (grad w.r.t. the raw matmul output), dW, dbias, dgamma, dbeta. */
typedef struct {
  const float* dout; int ld_dout;                         /* [B, ld_dout] or NULL */
  const float* dy_next; const float* w_next; int N_next;  /* [B,N_next], [*, N_next] rows 0..N-1 */
  const float* in;  int ld_in;  int K;
  const float* in2; int ld_in2; int K2;
  const float* y; const float* out; int ld_out;
  const float* gamma; const float* beta; const float* mean; const float* invstd;
  int bn_training;                                         /* 0: gamma*invstd only (frozen) */
  int act; float leaky_alpha;
  const float* drop_mask; float keep_prob;
  float* dy;                                               /* [B,N] */
  float* dw; float* dbias; float* dgamma; float* dbeta;    /* NULL -> not computed (frozen net) */
  int B, N;
} t3d_fc_bwd_args;
int t3d_fc_bwd(const t3d_fc_bwd_args* args, t3d_stream_t stream);

/* din[b,k] = alpha * sum_n dy[b,n] w[k,n] (+ add_in[b,k]) : the input gradient of an FC layer
 * (pooled-feature gradient, stage1_center gradient, global-feature gradient). */
typedef struct {
  const float* dy; int N;       /* [B,N] */
  const float* w;               /* [K,N] */
  const float* add_in; int ld_add;
  float alpha;
  float* din; int ld_din;
  int B, K;
  /* optional: the pooled form of K11c (t3d_bn_bwd_finalize) on din's columns in the same launch -- din is the gradient
   * of a max-pooled feature and column k is all that channel k's batch-norm backward needs; bn_coef == NULL: none */
  const float* bn_pooled; int bn_ld_pooled;   /* [B, ld] */
  const float* bn_ysel;                       /* [B,K] */
  float* bn_dpool;                            /* [B,K] out */
  int bn_count;
  const float* bn_gamma; const float* bn_mean; const float* bn_invstd; const float* bn_scale;
  int bn_frozen;
  float* bn_dgamma; float* bn_dbeta;          /* [K] or NULL */
  float* bn_coef;                             /* [3,K] out */
} t3d_fc_dinput_args;
int t3d_fc_dinput(const t3d_fc_dinput_args* args, t3d_stream_t stream);

/* ---- K5 + K7 + K10: segmentation head, forward + backward in one pass --------------------------
 * Replaces, for the last seg layer: tf_util.dropout (semisup_models.py:131), conv10 128->2 without
 * BN/activation (133-135), sparse_softmax_cross_entropy_with_logits (semisup_v1_sunrgbd.py:430),
 * the hard mask and masked xyz sums of subtract_points_mean (semisup_models.py:150-158), and their
 * gradients back to conv9's batch-norm output.
 * Per-frustum loss weight: w_b = ce_weight * (1 - is_data_2D[b]) / (B * rows_per_frustum). */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_seg_head_args) of the caller's header (see T3D_ABI_VERSION) */
  const float* y;            /* [M,K] raw conv9 output */
  const float* scale; const float* shift;
  const float* drop_mask;    /* [M,K] 0/1 or NULL */
  float keep_prob;
  const float* w;            /* [K,2] */
  const float* bias;         /* [2] */
  const int32_t* labels;     /* [M] or NULL (inference) */
  const int32_t* is_data_2D; /* [B] */
  const float* pc; int ld_pc;/* [M, ld_pc] xyz in cols 0..2 */
  float ce_weight;
  float* logits;             /* [M,2] */
  float* mask;               /* [M] */
  float* part;               /* [M/128, 8]: ce_sum, mask_cnt, sx, sy, sz, db0, db1, n_correct */
  float* dz;                 /* [M,K] grad wrt conv9 BN output (ReLU-masked) or NULL (no backward) */
  float* psum_dz;            /* [M/128,K] */
  float* psum_dzy;           /* [M/128,K] */
  float* dw_part;            /* [M/128,K,2] */
  int M, K, rows_per_frustum, B;
  /* drop_mask == NULL, keep_prob < 1, drop_hyper != NULL: the keep mask is generated where it is consumed, element (m,k) from
   * (drop_seed, step = drop_hyper[0], index m*K + k) with the generator of t3d_dropout_mask -- no [M,K] mask tensor in HBM. */
  uint32_t drop_seed;
  const float* drop_hyper;
  int dtype;                 /* element type of y and dz (T3D_F32 / T3D_BF16) */
  /* optional: d loss / d soft_mask[m] (soft_mask = softmax(logits)[:,1]) from t3d_weak_loss, added to the logit gradients -- the
   * head is then run a second time, behind the box losses, and rewrites dz, the partials and dw_part (NULL: none) */
  const float* dsoft;        /* [M] or NULL */
  /* optional: the `oracle_mask` of get_semi_model_final (semisup_v1_sunrgbd.py:161-162; test_semisup.py:75 feeds y_seg): the
   * logits that leave the head -- written, masked on, fed to the cross-entropy -- are stack([1 - m, m]) instead of conv10's; no
   * gradient reaches conv9 then (the stacked tensor is a constant of the graph). */
  const int32_t* oracle_mask; /* [M] or NULL */
} t3d_seg_head_args;
int t3d_seg_head(const t3d_seg_head_args* args, t3d_stream_t stream);

/* Combines the seg-head tile partials per frustum: mask_xyz_mean[B,3] (semisup_models.py:157-158),
 * seg CE per frustum (mean over points), and the conv10 weight/bias gradient. */
typedef struct {
  const float* part; const float* dw_part;
  int B, tiles_per_frustum, rows_per_frustum, K;
  float* mask_xyz_mean;      /* [B,3] */
  float* seg_loss;           /* [B] mean CE per frustum */
  float* dw;                 /* [K,2] or NULL */
  float* dbias;              /* [2] or NULL */
  float* n_correct;          /* [1] or NULL */
} t3d_seg_finalize_args;
int t3d_seg_finalize(const t3d_seg_finalize_args* args, t3d_stream_t stream);

/* ---- K9 + K10: box losses, forward + backward ---------------------------------------------------
 * Replaces get_strong_loss (semisup_v1_sunrgbd.py:423-553), huber_loss (555-564), the corner boxes of
 * model_util.get_box3d_corners_sunrgbd / _helper (model_util.py:94-119,145-167) and the anchor->reg
 * conversion tf_convert_box_params_from_anchor_to_reg_format_multi (tf_util.py:1001-1041), for the
 * 67-wide head output `box` and the T-Net centre.  total[b] = w3d[b]*(seg_w*seg_loss[b] + box_l[b]),
 * loss = sum_b total[b] * (mean_over_B ? 1/B : 1/(sum w3d + 1e-3)).   w3d = 1 - is_data_2D. */
typedef struct {
  float center, orient_cls, orient_reg, dims_cls, dims_reg, tnet_center, corner, box_multiplier,
        cross_entropy;
} t3d_strong_weights;
typedef struct {
  const float* box; int ld_box;          /* [B,67] head output (centre residual first) */
  const float* stage1_center;            /* [B,3] */
  const float* seg_loss;                 /* [B] or NULL */
  const float* y_center; const int32_t* y_orient_cls; const float* y_orient_reg;
  const int32_t* y_dims_cls; const float* y_dims_reg; const int32_t* is_data_2D;
  t3d_strong_weights wts;
  int normalize_by_3d_count;             /* 0: mean over B (model A); 1: /(sum w3d + 1e-3) (model F) */
  float* dbox;                           /* [B,67] */
  float* dstage1;                        /* [B,3] */
  float* terms;                          /* [B,8]: mask, center, stage1, hcls, hres, scls, sres, corner */
  float* total_losses;                   /* [B] */
  float* loss;                           /* [1] */
  float* center;                         /* [B,3] end_points['center'] */
  float* reg_dims;                       /* [B,3] anchor->reg dims */
  float* reg_theta;                      /* [B] */
  /* optional per-frustum IoU of the predicted box (arg-max bins) with the label box: the tf.py_func summary
   * get_iou_summary / compute_box3d_iou (semisup_v1_sunrgbd.py:236-246, roi_seg_box3d_dataset.py:103-140).  NULL: skipped. */
  float* iou2d;                          /* [B] ground-plane IoU */
  float* iou3d;                          /* [B] */
  int B;
} t3d_strong_loss_args;
int t3d_strong_loss(const t3d_strong_loss_args* args, t3d_stream_t stream);

/* ---- K9b: weak box losses of get_semi_loss_backbone, forward + backward ------------------------------------------------
 * Replaces weak_losses.get_reprojection_loss (models/weak_losses.py:69-229, with tf_util.tf_rot_box_params_multi 1045-1072,
 * tf_create_3D_box_by_vertices_multi 841-891, tf_get_2D_bbox_of_(softmax_)projection_sunrgbd_multi 364-449, tf_dilate_2D_bboxes
 * 486-514, tf_clip_2D_bbox_to_image_dims_multi 517-540) and weak_losses.get_surface_loss (231-259, with
 * tf_distance_to_closest_3D_box_surface_multi, tf_util.py:610-719) as called at semisup_v1_sunrgbd.py:270-311 on
 * S_pred_box_reg = (center, reg_dims, reg_theta) -- the outputs of t3d_strong_loss:
 *   weak[b] = w_reproj * reprojection[b] + w_surface * surface[b]
 *   total_losses[b] += is_data_2D[b] * multiplier * weak[b];   loss += mean_b of that term
 *   dbox7[b] = d loss / d (center, reg_dims, reg_theta)[b]     (feed t3d_anchor_reg_bwd)
 *   dsoft[m] = d loss / d soft_mask[m]   (soft_mask = softmax(logits)[:,1]; the reference multiplies the surface loss by the
 *              soft mask itself, so this gradient exists whatever WEAK_TRAIN_SEG_W_SURFACE says: feed t3d_seg_head.dsoft)
 * Run after t3d_strong_loss / t3d_semi_final_loss (in/out: total_losses, loss).  Two launches (per-point surface distances; per-frustum finish).
 * A loss is evaluated whenever its inputs are given (pc + logits; Rtilt .. img_dim), whatever its weight -- the reference logs both. */
typedef struct {
  const float* center;       /* [B,3] */
  const float* reg_dims;     /* [B,3] */
  const float* reg_theta;    /* [B] */
  const float* pc; int ld_pc;/* [B*N, ld_pc] xyz in cols 0..2 (surface loss) */
  const float* logits;       /* [B*N,2] (surface loss) */
  const float* Rtilt;        /* [B,3,3] */
  const float* K;            /* [B,3,3] */
  const float* rot_frust;    /* [B] */
  const float* box2D;        /* [B,4] left, top, right, bottom */
  const float* img_dim;      /* [B,2] rows, cols */
  const int32_t* is_data_2D; /* [B], or NULL: every sample counts (get_semi_loss_final without WEAK_REPROJECTION_ONLY_ON_2D_CLS) */
  float w_reproj, w_surface; /* WEAK_WEIGHT_REPROJECTION, WEAK_WEIGHT_SURFACE */
  float multiplier;          /* SEMI_MULTIPLIER_FOR_WEAK_LOSS */
  int use_softmax_proj; float softmax_scale;      /* WEAK_REPROJECTION_USE_SOFTMAX_PROJ, _SOFTMAX_SCALE */
  float dilate;              /* WEAK_REPROJECTION_DILATE_FACTOR */
  int clip_lower_b_loss, clip_pred_box;           /* WEAK_REPROJECTION_CLIP_LOWERB_LOSS, _CLIP_PRED_BOX */
  int loss_mse;              /* WEAK_REPROJECTION_LOSS_TYPE: 0 huber, 1 mse */
  int32_t train_box_reproj[3];                    /* WEAK_TRAIN_BOX_W_REPROJECTION (centre, dims, theta) */
  int32_t train_box_surface[3];                   /* WEAK_TRAIN_BOX_W_SURFACE */
  float surface_margin, surface_scale_dims;       /* WEAK_SURFACE_MARGIN, WEAK_SURFACE_LOSS_SCALE_DIMS */
  float* surf_part;          /* [B, N/128, 8] scratch */
  float* dsoft;              /* [B*N] out or NULL */
  float* reproj;             /* [B] out or NULL */
  float* surface;            /* [B] out or NULL */
  float* dbox7;              /* [B,7] out */
  float* total_losses;       /* [B] in/out or NULL */
  float* loss;               /* [1] in/out */
  int B, N;
  /* stage c (get_semi_loss_final, semisup_v1_sunrgbd.py:345-359): the inactive-volume loss of the box dims per class
   * (weak_losses.get_inactive_volume_loss_v1, weak_losses.py:39-67): loss += multiplier * w_inactive * mean over the trained classes
   * of mean_{b in class} max(0, margin[class] - l*w*h); its gradient joins dbox7[:,3:6].  w_inactive == 0: none */
  const float* one_hot;      /* [B,10] class one-hot (class_ids = arg-max) or NULL */
  float w_inactive;          /* WEAK_WEIGHT_INACTIVE_VOLUME */
  float inactive_margins[10];/* WEAK_INACTIVE_VOL_LOSS_MARGINS */
  int32_t inactive_train[10];/* end_points['inactive_vol_train_classes'] */
  float* inactive;           /* [1] out or NULL */
} t3d_weak_loss_args;
int t3d_weak_loss(const t3d_weak_loss_args* args, t3d_stream_t stream);

/* ---- K8: Box-PC representation ----------------------------------------------------------------------
 * Replaces tf_get_box_pc_representation (tf_util.py:764-795) + tf_create_3D_box_by_surface_centers_multi
 * (893-955): rep[m, 0:C] = pc[m, 0:C]; rep[m, C:C+6] = the six signed point-to-face distances of point m to the
 * box of its frustum (centre, dims (l,w,h), heading); columns C+6 .. ld_rep-1 are zero padding.
 * Box in regression form (center, dims, theta), or, when y_dims_cls != NULL, in label form: dims/theta hold the
 * residuals and the box is max(anchor[cls] + res, 1e-5) / bin[cls] + res (boxpc_sunrgbd.py:206-230).
 * box_out[B,7] (optional) receives (cx,cy,cz,l,w,h,theta) for the backward.  rows_per_frustum % 256 == 0. */
typedef struct {
  uint32_t struct_size;    /* = sizeof(t3d_boxpc_rep_args) of the caller's header (see T3D_ABI_VERSION) */
  const float* pc; int ld_pc; int C;
  const float* center;            /* [B,3] */
  const float* dims;              /* [B,3] */
  const float* theta;             /* [B]   */
  const int32_t* y_dims_cls;      /* [B] or NULL */
  const int32_t* y_orient_cls;    /* [B] or NULL */
  float* rep; int ld_rep;         /* [M, ld_rep] */
  float* box_out;                 /* [B,7] or NULL */
  int M, rows_per_frustum;
  /* optional: `--mask_pc_for_boxpc` of test_semisup.py:103-105 -- the net sees pc * mask (every channel of a background point is 0,
   * so its six distances are those of the origin) */
  const float* rowmask;           /* [M] 0/1 or NULL */
} t3d_boxpc_rep_args;
int t3d_boxpc_rep(const t3d_boxpc_rep_args* args, t3d_stream_t stream);

/* Gradient of the representation w.r.t. the box: dbox[B,7] = d(cx,cy,cz,l,w,h,theta), reduced over the points,
 * from drep[m, coff .. coff+5] = gradient w.r.t. the six distance channels (stage c: train_semisup_adv.py:331-411). */
typedef struct {
  const float* pc; int ld_pc;
  const float* box;               /* [B,7] from t3d_boxpc_rep */
  const float* drep; int ld_drep; int coff;
  float* dbox;                    /* [B,7] */
  int B, rows_per_frustum;
} t3d_boxpc_rep_bwd_args;
int t3d_boxpc_rep_bwd(const t3d_boxpc_rep_bwd_args* args, t3d_stream_t stream);

/* Box-PC loss, forward + backward (boxpc_sunrgbd.py:106-193, huber form): out[B,9] = [dcentre(3), dsize(3), dangle,
 * fit logits(2)];  loss = mean_b( w_cls*CE(logits, iou > fit_bound) + w_delta*wl*(w_center*mean3 Huber + w_size*mean3
 * Huber + w_angle*Huber) ).  terms[B,4] = (CE, delta loss, p_fit, total).
 * weigh_pred_by_cls_conf (BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF, boxpc_sunrgbd.py:84-92): the Huber terms see the PREDICTED deltas
 * times (1 - p_fit).  grad_cls_via_delta = !BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA (boxpc_sunrgbd.py:73-74): p_fit inside wl / the
 * prediction weight is differentiated (the gradient reaches the fit logits); 0 = stop_gradient, the recipes' setting. */
typedef struct {
  const float* out;               /* [B,9] */
  const float* y_box_iou; const float* y_center_delta; const float* y_dims_delta; const float* y_orient_delta;
  float fit_bound, w_cls, w_delta, w_center, w_size, w_angle;
  int weigh_by_cls_conf, weigh_by_cls_gt;
  float* dout;                    /* [B,9] */
  float* terms;                   /* [B,4] */
  float* loss;                    /* [1] */
  int B;
  int weigh_pred_by_cls_conf, grad_cls_via_delta;
  int delta_loss_mse;             /* BOXPC_DELTA_LOSS_TYPE == 'mse' (boxpc_sunrgbd.py:158-164): squared error instead of Huber */
} t3d_boxpc_loss_args;
int t3d_boxpc_loss(const t3d_boxpc_loss_args* args, t3d_stream_t stream);

/* ---- stage-c glue (SEMI_MODEL F with the frozen Box-PC net; train_semisup_adv.py:331-411) ---------------- */

/* Input gradient of a per-point layer restricted to kn <= 8 input channels k0..k0+kn-1 (dense dy only):
 * out[m, j] = sum_n dy[m,n] w[k0+j, n].  Used for the six distance channels of the Box-PC representation.
 * M % 64 == 0, N % 128 == 0. */
typedef struct {
  t3d_dy_src dy;
  const float* w;          /* [K,N] */
  int k0, kn;
  float* out; int ld_out;  /* [M, ld_out] */
  int M, N;
} t3d_pointmlp_dgrad_narrow_args;
int t3d_pointmlp_dgrad_narrow(const t3d_pointmlp_dgrad_narrow_args* args, t3d_stream_t stream);

/* get_semi_loss_final's weak terms (semisup_v1_sunrgbd.py:343-407) on top of the strong loss already computed by
 * t3d_strong_loss(normalize_by_3d_count = 1):
 *   loss = strong + w_weak * intraclass(reg_dims | class) + w_fit * mean_b(-log(0.01 + p_fit[b]) * (fit_only_2d ? is2D : 1))
 * with w_weak = SEMI_MULTIPLIER_FOR_WEAK_LOSS * WEAK_WEIGHT_INTRACLASSVAR, the intraclass-variance loss of
 * weak_losses.py:267-291 (huber), p_fit = softmax(out9[:,7:9])[:,1].  Writes the gradients w.r.t. reg_dims and out9. */
typedef struct {
  const float* strong_loss;       /* [1] device scalar */
  const float* reg_dims;          /* [B,3] */
  const float* one_hot;           /* [B,10] class one-hot (class_ids = argmax) */
  const int32_t* is_data_2D;      /* [B] */
  const float* out9;              /* [B,9] Box-PC output */
  int32_t train_classes[10];
  float w_weak, w_fit;
  int fit_only_2d;
  float* d_dims;                  /* [B,3] */
  float* dout9;                   /* [B,9] */
  float* fit_prob;                /* [B] end_points['boxpc_fit_prob'] */
  float* terms;                   /* [2]: intraclass loss, fit loss */
  float* loss;                    /* [1] */
  int B;
} t3d_semi_final_loss_args;
int t3d_semi_final_loss(const t3d_semi_final_loss_args* args, t3d_stream_t stream);

/* Backward of the anchor->reg conversion (tf_util.py:1017-1031): accumulates the gradient of (centre, dims, theta)
 * -- dbox7[B,7] from the Box-PC path and/or d_dims[B,3] -- into dbox[B,67] and dstage1[B,3] (both in/out). */
typedef struct {
  const float* box; int ld_box;   /* [B,67] head output (scores pick the bins) */
  const float* dbox7;             /* [B,7] or NULL */
  const float* d_dims;            /* [B,3] or NULL */
  float* dbox;                    /* [B,67] in/out */
  float* dstage1;                 /* [B,3] in/out */
  int B;
} t3d_anchor_reg_bwd_args;
int t3d_anchor_reg_bwd(const t3d_anchor_reg_bwd_args* args, t3d_stream_t stream);

/* One step of the iterated Box-PC refinement of the inference graph (test_semisup.py:101-134):
 *   p_fit = softmax(out9[:,7:9])[:,1];  w = (1 - p_fit)^weigh_by_conf, weigh_by_conf = 0, 1 or 2: one factor for
 *   SEMI_WEIGH_BOXPC_DELTA_DURING_TEST (test_semisup.py:119-121), one for BOXPC_WEIGH_DELTA_PRED_BY_CLS_CONF (the Box-PC
 *   model's own weighting of its predicted deltas, boxpc_sunrgbd.py:84-92)
 *   box_out = box_in - w * out9[:,0:7]  (centre 0:3, size 3:6, angle 6);  total (+)= w * out9[:,0:7]  (`first`: =)
 * The F2_ heads are the F_ heads minus `total` (test_semisup.py:136-142). */
typedef struct {
  const float* out9;              /* [B,9] Box-PC net output */
  const float* center_in; const float* dims_in; const float* theta_in;   /* [B,3],[B,3],[B] */
  float* center_out; float* dims_out; float* theta_out;                  /* may alias the inputs */
  float* total;                   /* [B,7] accumulated deltas */
  float* fit_prob;                /* [B] or NULL */
  int weigh_by_conf;
  int first;
  int B;
} t3d_box_refine_step_args;
int t3d_box_refine_step(const t3d_box_refine_step_args* args, t3d_stream_t stream);

/* Backward of one refinement step inside the TRAINING graph (train_semisup_adv.py:362-386 with SEMI_REFINE_USING_BOXPC_DELTA_NUM > 1
 * and SEMI_BOXPC_MIN_FIT_LOSS_AFT_REFINE: the fit loss reads the Box-PC evaluation of the box refined NUM-1 times, so its gradient
 * runs back through every step  box_next = box - w * out9[:,0:7],  w = (1 - p_fit)^weigh_by_conf):
 *   tot = dbox_rep + carry            gradient w.r.t. box_next: through the next evaluation's representation (t3d_boxpc_rep_bwd)
 *                                     + what flows past it to the steps behind (carry, NULL at the last step)
 *   tot_out = tot                     (the identity path: the carry of the step before)
 *   dout9[:,0:7] = -w * tot;  dout9[:,7:9] = (grad_via_conf ? d w / d logits . sum_k tot_k * (-out9_k) : 0)
 * With out9 == NULL only tot_out is written (the total gradient w.r.t. the unrefined box). */
typedef struct {
  const float* out9;              /* [B,9] Box-PC output of THIS step's evaluation, or NULL */
  const float* dbox_rep;          /* [B,7] */
  const float* carry;             /* [B,7] or NULL */
  float* tot_out;                 /* [B,7] (may alias carry or dbox_rep) */
  float* dout9;                   /* [B,9], required with out9 */
  int weigh_by_conf;              /* as t3d_box_refine_step */
  int grad_via_conf;              /* !BOXPC_STOP_GRAD_OF_CLS_VIA_DELTA (boxpc_sunrgbd.py:73-74) */
  int B;
} t3d_box_refine_step_bwd_args;
int t3d_box_refine_step_bwd(const t3d_box_refine_step_bwd_args* args, t3d_stream_t stream);

/* ---- K13: 3-D IoU of upright boxes ---------------------------------------------------------------------
 * Replaces box_util.box3d_iou (the Frustum-PointNets module the reference imports but does not ship) at its call sites:
 * roi_seg_box3d_dataset.py:103-140 (per-step IoU summary), box_pc_fit_dataset.py:38-42,211-244 (perturb a box until its IoU
 * falls inside the fit / no-fit bounds), eval_det.py:60-66 (detection matching).  Ground-plane rectangle intersection (area of
 * the clipped polygon), height overlap, iou3d = inter_vol / (vol1 + vol2 - inter_vol), iou2d = inter / (a1 + a2 - inter).
 * Parameter form: centre (x,y,z), size (l,w,h), heading about y, as get_3d_box takes them (roi_seg_box3d_dataset.py:86-101);
 * negative l / w count by magnitude (a mirrored rectangle), a negative h gives iou3d = 0 (inverted height range), as the
 * reference's get_3d_box + box3d_iou behave on the sizes an untrained head produces. */
typedef struct {
  const float* center1; const float* size1; const float* heading1;      /* [n,3], [n,3], [n] */
  const float* center2; const float* size2; const float* heading2;
  float* iou3d; float* iou2d;                                           /* [n]; iou2d may be NULL */
  int n;
} t3d_box3d_iou_args;
int t3d_box3d_iou(const t3d_box3d_iou_args* args, t3d_stream_t stream);

/* Corner form, box3d_iou's own signature: corners[n,8,3] in get_3d_box order (0-3 the y-max face, 4-7 the y-min face). */
typedef struct {
  const float* corners1; const float* corners2;                         /* [n,8,3] */
  float* iou3d; float* iou2d;
  int n;
} t3d_box3d_iou_corners_args;
int t3d_box3d_iou_corners(const t3d_box3d_iou_corners_args* args, t3d_stream_t stream);

/* The extra fully-connected input columns of v1_tnet / v1_box_est under USE_NORMALIZED_BOX2D_AS_FEATS
 * (semisup_models.py:192-195, 249-252: concat [pooled | one_hot | norm_box2D]): out[B, n_oh + 4] = [one_hot (n_oh = 0 or 10) |
 * tf_util.tf_normalize_2D_bboxes(box2D, img_dim) (tf_util.py:466-484) = left/cols, top/rows, right/cols, bottom/rows with
 * img_dim = (rows, cols)].  Inputs are placeholders: no gradient. */
typedef struct {
  const float* one_hot; int n_oh;        /* [B, n_oh] or NULL with n_oh = 0 */
  const float* box2D;                    /* [B,4] */
  const float* img_dim;                  /* [B,2] */
  float* out;                            /* [B, n_oh + 4] */
  int B;
} t3d_box2d_feats_args;
int t3d_box2d_feats(const t3d_box2d_feats_args* args, t3d_stream_t stream);

/* compute_box3d_iou on raw box heads (roi_seg_box3d_dataset.py:103-140): box[B,67] = [centre - stage1_center (3), heading scores
 * (12), normalised heading residuals (12), size scores (10), normalised size residuals (10x3)]; predicted box from the arg-max
 * bins (class2angle / class2size), label box from the label bins + residuals. */
typedef struct {
  const float* box; int ld_box;
  const float* stage1_center;            /* [B,3] or NULL (centre already absolute) */
  const float* y_center; const int32_t* y_orient_cls; const float* y_orient_reg; const int32_t* y_dims_cls; const float* y_dims_reg;
  float* iou2d; float* iou3d;            /* [B] */
  int B;
} t3d_box_head_iou_args;
int t3d_box_head_iou(const t3d_box_head_iou_args* args, t3d_stream_t stream);

/* ---- device-side batch assembly -------------------------------------------------------------------------------
 * Replaces the host loop ROISemiDataset.get_batch -> get_classes3D (roi_semi_dataset.py:283-347, 482-535; helpers
 * roi_seg_box3d_dataset.py:37-77, 346-368) for a data set of ragged frustums resident in HBM.  Per batch slot b with
 * frustum f = sample[b]: N points drawn with replacement (np.random.choice(count, N, replace=True), 301-303), rotated to
 * the centre view (rot = pi/2 + frustum_angle; [x,z] <- [x c - z s, x s + z c]), flipped in x with probability 1/2,
 * shifted along z by clip(randn*dist*0.05, 0.8 dist, 1.2 dist) with dist = |centre.xy| (the reference's bounds, as
 * written) and along y by U(-0.2, 0.2); labels: rotated/flipped/shifted box centre, heading -> (bin, residual) by
 * angle2class over 12 bins, size -> (class, size - mean size of the class), one-hot class.
 * Draws: `choice` [B,N] and `aug` [B,3] = (flip, randn, u) when given (parity tests), else generated from
 * (seed, hyper[0] = step, b, n) -- so the whole input pipeline can sit inside the captured step.
 * ALTERNATE_BATCH (train_semisup_adv.py:538-565, sample_pure_from_2D_cls / _3D_cls 589-596): see sample2. */
typedef struct {
  const float* points;         /* [total, C_src] all frustums, concatenated; xyz in columns 0..2 */
  const int32_t* seg;          /* [total] per-point labels */
  const int64_t* offsets;      /* [F+1] */
  const float* frustum_angle;  /* [F] */
  const float* box_center;     /* [F,3] */
  const float* heading;        /* [F] */
  const float* size;           /* [F,3] (l,w,h) */
  const int32_t* cls;          /* [F] class id */
  const int32_t* sample;       /* [B] frustum index per batch slot, or (sample_len > 0) a permutation [sample_len] of which slot b
                                  of step s takes entry (s*B + b) mod sample_len */
  int sample_len;
  const int32_t* choice;       /* [B,N] or NULL */
  const float* aug;            /* [B,3] or NULL */
  int C_src, C, B, N;
  int rotate_to_center, random_flip, random_shift;
  uint32_t seed;
  const float* hyper;          /* device step counter (hyper[0]); required when choice == NULL */
  float* pc;                   /* [B,N,C] out */
  int32_t* y_seg;              /* [B,N] out */
  float* y_center;             /* [B,3] */
  int32_t* y_orient_cls;       /* [B] */
  float* y_orient_reg;         /* [B] */
  int32_t* y_dims_cls;         /* [B] */
  float* y_dims_reg;           /* [B,3] */
  float* one_hot;              /* [B,10] */
  float* rot_angle;            /* [B] or NULL */
  const int32_t* sample2;      /* NULL, or the second list of ALTERNATE_BATCH sampling: even steps take slot b from
                                  sample[((s/2)*B + b) mod sample_len] (2-D-label classes, is_data_2D = 1), odd steps from
                                  sample2[((s/2)*B + b) mod sample2_len] (3-D-label classes, is_data_2D = 0) */
  int sample2_len;
  int32_t* is_data_2D;         /* [B] or NULL */
  const int32_t* frustum_is_2D; /* [F] or NULL: per-frustum flag of a combined data set (SEMI_SAMPLING_METHOD BATCH: get_batch walks the
                                  3-D-label list followed by the 2-D-label list, roi_semi_dataset.py:482-535); when given it decides
                                  is_data_2D in the one-list modes */
  int ld_pc;                   /* row stride of pc in floats (>= C; 0 = C).  A multiple of 4 when the layers read pc (C = 6: xyz + rgb
                                  rows padded to 8 floats; the padding is written as zeros) */
  const int32_t* slot_is_2D;   /* [B] or NULL: per-slot flag written earlier in the step by t3d_sample_equal_classes (its is_data_2D)
                                  when neither sample2 nor frustum_is_2D decides */
  /* A slot that holds a frustum of the 2-D-label list is assembled the way ROISemiDataset.get_classes2D does it
   * (roi_semi_dataset.py:383-452): resampled and rotated to the centre view, but NOT flipped or shifted ("2D Classes cannot be
   * augmented because the projection will no longer be accurate"), and every 3-D label (y_seg, y_center, orientation and size
   * class / residual) is written as zero; one_hot and rot_angle are kept. */
  /* optional camera side of the weak losses: per-frustum calibration and 2-D box of the data set (roi_semi_dataset.py:243-246,
   * 355-359) copied to the batch slot (the frustum rotation the reprojection loss undoes is rot_angle).  cam_rtilt == NULL: none */
  const float* cam_rtilt;      /* [F,9] */
  const float* cam_k;          /* [F,9] */
  const float* cam_box2d;      /* [F,4] left, top, right, bottom */
  const float* cam_img_dim;    /* [F,2] rows, cols */
  float* Rtilt;                /* [B,9] out */
  float* K;                    /* [B,9] out */
  float* box2D;                /* [B,4] out */
  float* img_dim;              /* [B,2] out */
} t3d_batch_assemble_args;
int t3d_batch_assemble(const t3d_batch_assemble_args* args, t3d_stream_t stream);

/* Class-balanced batch composition: `equal_samples_per_class` of ROISemiDataset.sample_from_set (roi_semi_dataset.py:558-567) and
 * BoxPCFitDataset.sample_batch (box_pc_fit_dataset.py:367-377).  The B slots are split over the n class groups like
 * np.array_split([1]*B, n) -- B mod n groups get one slot more, WHICH groups is random per step (random.shuffle of the split) --
 * slots are laid out group after group, and each slot draws a frustum of its group with replacement.  Writes sample[B] (the
 * frustum index per slot) for t3d_batch_assemble's explicit-sample mode (sample_len = 0).
 * Two sets = ALTERNATE_BATCH (train_semisup_adv.py:538-565): even steps draw from set[0] (classes with 2-D labels only,
 * is_data_2D = 1), odd steps from set[1] (is_data_2D = 0). */
typedef struct {
  const int32_t* members;        /* frustum indices, grouped by class */
  const int32_t* offsets;        /* [n_groups + 1] into members; every group non-empty */
  int n_groups;                  /* 1..32; 0 = set not used */
  const int32_t* perm;           /* [perm_len] epoch permutation of the set's frustums (the not-balanced batches), or NULL */
  int perm_len;
} t3d_class_groups;
typedef struct {
  t3d_class_groups set[2];
  int B;                         /* <= 1024 */
  uint32_t seed;
  const float* hyper;            /* device step counter (hyper[0]) */
  const float* order_draws;      /* [n_groups] keys: the groups with the smallest keys take the larger share; NULL: generated */
  const float* member_draws;     /* [B] uniforms in [0,1) picking the member; NULL: generated */
  float equal_prob;              /* *_SAMPLE_EQUAL_CLASS_WITH_PROB: a step is class-balanced when its uniform draw < equal_prob
                                    (train_semisup.py:357, train_boxpc.py:323-328); otherwise B distinct frustums,
                                    np.random.choice(len, B, replace=False): slot b takes perm[(steps_of_this_set * B + b) mod perm_len] */
  const float* prob_draw;        /* [1] that uniform, or NULL (generated) */
  int32_t* sample;               /* [B] out */
  int32_t* is_data_2D;           /* [B] out, or NULL */
} t3d_sample_equal_classes_args;
int t3d_sample_equal_classes(const t3d_sample_equal_classes_args* args, t3d_stream_t stream);

/* Box-PC Fit training samples (box_pc_fit_dataset.py:105-185 `get`, 211-244 `perturb_box_to_diff_ious`, fed by
 * train_boxpc.py:343-355): each frustum's label box is perturbed until its 3-D IoU with the label box falls strictly inside the
 * "fit" bounds (with probability proportion_fit) or the "no-fit" bounds.  Candidate t of frustum b:
 *   centre + U(-cp,cp)^3,  size + size*U(-sp,sp)^3,  heading + U(0,ap),   (cp,sp,ap) = perturbation * (1 - mean(bounds)).
 * The reference draws candidates one after another; here the 64 lanes of a wave each test one candidate per round and the lowest
 * accepted candidate index wins -- the same first-accepted law over the same candidate stream.  After max_rounds*64 rejected
 * candidates the last candidate of lane 0 is taken and its true IoU reported (the reference would loop on).
 * In place on the label buffers written by t3d_batch_assemble: centre / heading bin+residual / size residual become the
 * perturbed box (the net's x_* inputs); the class is unchanged (size2class keeps the label's type, 179-180). */
typedef struct {
  float* center;                 /* [B,3] in: label centre; out: perturbed centre */
  int32_t* orient_cls;           /* [B]   in/out heading bin  (angle2class of the perturbed heading) */
  float* orient_reg;             /* [B]   in/out heading residual */
  const int32_t* dims_cls;       /* [B]   size class */
  float* dims_reg;               /* [B,3] in/out size residual w.r.t. the class mean size */
  float* y_box_iou;              /* [B] out */
  float* y_center_delta;         /* [B,3] out: perturbed - label */
  float* y_dims_delta;           /* [B,3] out */
  float* y_orient_delta;         /* [B] out */
  float center_perturbation, size_perturbation, angle_perturbation;     /* BOXPC_*_PERTURBATION (config.py:27-29) */
  float fit_lo, fit_hi, nofit_lo, nofit_hi, proportion_fit;              /* BOXPC_FIT_BOUNDS, BOXPC_NOFIT_BOUNDS, BOXPC_PROPORTION_OF_BOXPC_FIT */
  const float* fit_draw;         /* [B] uniforms deciding fit / no-fit, or NULL (generated) */
  const float* cand_draws;       /* [B, max_rounds*64, 7] uniforms in [0,1) of the candidates, or NULL (generated) */
  int max_rounds;
  uint32_t seed;
  const float* hyper;            /* device step counter; required when draws are generated */
  int B;
} t3d_boxpc_perturb_args;
int t3d_boxpc_perturb(const t3d_boxpc_perturb_args* args, t3d_stream_t stream);

/* ---- K11d / K12 / schedules --------------------------------------------------------------------- */

/* grad[off_i + e] = sum_s slabs_i[s, e]  for every tensor i of a device-side table. */
typedef struct { int64_t slab_off; int64_t grad_off; int32_t numel; int32_t n_slabs; } t3d_slab_desc;
int t3d_reduce_slabs(const float* slab_base, float* grad_base, const t3d_slab_desc* table_dev,
                     int n_tensors, int max_numel, t3d_stream_t stream);
/* t3d_reduce_slabs + t3d_pool_sparse_rows (both wait only for stage 1) in one launch. */
int t3d_pool_bwd_mid(const float* slab_base, float* grad_base, const t3d_slab_desc* table_dev, int n_tensors, int max_numel,
                     const t3d_pool_sparse_rows_args* sparse, t3d_stream_t stream);

/* hyper[0] = step (as float, incremented), [1] = lr, [2] = bn_decay, [3] = adam lr_t.
 * Replaces tf.train.exponential_decay(staircase) for lr and bn momentum (train_semisup.py:127-145)
 * and the Adam bias correction.  Device-resident so a captured graph advances by itself. */
typedef struct {
  float base_lr, lr_decay_rate, lr_decay_step;
  float bn_init_decay, bn_decay_rate, bn_decay_step, bn_decay_clip;
  float beta1, beta2;
  int batch_size;
  int step_offset;         /* the step this call runs is hyper[0] + step_offset (0: every call advances one counter; 1: two counters
                            * advance alternately -- the two contexts of the software-pipelined step, step.PipelinedStep) */
} t3d_schedule;
int t3d_schedule_step(float* hyper, const t3d_schedule* s, t3d_stream_t stream);

/* tf.train.AdamOptimizer (train_semisup.py:230), TF form: w -= lr_t * m / (sqrt(v) + eps).
 * `grad_scale` divides the (all-reduced) gradient sum by the world size. */
int t3d_adam_tf_step(float* params, const float* grads, float* m, float* v, int64_t n,
                     const float* hyper, float beta1, float beta2, float eps, float grad_scale,
                     t3d_stream_t stream);

/* fp32 GEMMs on the bf16 matrix pipe (csrc/pointmlp.hip PathX3): x = h + m + l exactly with h = bf16(x), m = bf16(x - h),
 * l = bf16(x - h - m).  Writes the three planes of src[0..n): planes[p * plane_stride + i] (bf16 elements, plane_stride >= n).  The
 * optimiser's fp32 weights are split ONCE per step this way instead of once per tile that stages them. */
int t3d_split_x3(const float* src, void* planes, int64_t n, int64_t plane_stride, t3d_stream_t stream);

/* ... and in MFMA-FRAGMENT ORDER, which is what the x3 GEMM kernels take as `w_x3` (ABI version 3): for every [K, N] matrix of the
 * device table (K % 32 == 0, N % 32 == 0, off % 8 == 0) the three planes are written at planes_*[p * plane_stride + off ...] as
 *     fragment f = (rt * NB + nb) * 64 + lane  ->  eight bf16 at element off + 8 f:  Op[nb * 32 + (lane & 31)][rt * 16 + 8 * (lane >> 5) + j]
 * with Op[n][k] = w[k][n], NB = N / 32 (planes_fwd: the forward's weight operand) and Op[k][n] = w[k][n], NB = K / 32 (planes_dgrad:
 * the data gradient's) -- the B operand of one v_mfma_f32_32x32x16_bf16 per 64 x 16 bytes.  One launch for every layer, once per step
 * behind the optimiser.  `fwd` / `dgrad` select the arrangements an entry needs.  Replaces nothing in the reference: TensorFlow's
 * conv2d reads its fp32 weights (models/tf_util.py:1308). */
typedef struct {
  int64_t off;             /* element offset of the matrix in `params` (and of its planes in every plane buffer) */
  int32_t K, N;
  int32_t fwd, dgrad;      /* write the forward / the data-gradient arrangement */
  int32_t blk0;            /* first workgroup of this matrix: ascending over the table, matrix e owns ceil(K N / 2048) workgroups */
  int32_t reserved;
} t3d_x3_frag_entry;
/* n_blocks = the sum of ceil(K N / 2048) over the table (one 256-thread workgroup per 256 eight-element fragments) */
int t3d_split_x3_frag(const float* params, void* planes_fwd, void* planes_dgrad, int64_t plane_stride, const t3d_x3_frag_entry* table,
                      int n_entries, int n_blocks, t3d_stream_t stream);

/* tf.train.MomentumOptimizer(learning_rate, momentum) of `--optimizer momentum` (train_semisup.py:226-228, train_boxpc.py:247,
 * train_semisup_adv.py:296), TF form without Nesterov: accum = momentum * accum + g * grad_scale;  w -= lr * accum, with
 * lr = hyper[1] (the staircase schedule t3d_schedule_step evaluates).  `accum` is the variable's `Momentum` slot. */
int t3d_momentum_step(float* params, const float* grads, float* accum, int64_t n, const float* hyper, float momentum,
                      float grad_scale, t3d_stream_t stream);

/* Standalone tf_util.dropout on a per-point tensor (tf_util.py:1720-1741; the reference's one call site, conv9 -> dp1 -> conv10 of
 * v1_inst_seg, semisup_models.py:131, is fused into t3d_seg_head on the hot path): materialises
 *   out[m,k] = act(a)[m,k] * mask[m,k] / keep_prob          (mask == NULL or keep_prob >= 1: out = act(a))
 * from the lazy activation operand, fp32 [M,K] row-major. */
typedef struct {
  t3d_act_src a;
  const float* mask;       /* [M,K] 0/1 keep flags or NULL */
  float keep_prob;
  float* out;              /* [M,K] */
  int M, K;
  int rows_per_frustum;
} t3d_act_dropout_args;
int t3d_act_dropout(const t3d_act_dropout_args* args, t3d_stream_t stream);

/* dst[i] = bf16(src[i]): the copy of the weights the T3D_BF16 GEMMs read (`w` of t3d_pointmlp_fwd / _dgrad / _bwd then points into
 * it); run once per step after the optimiser.  The fp32 master copy stays what Adam updates and what checkpoints hold. */
int t3d_cast_bf16(const float* src, void* dst, int64_t n, t3d_stream_t stream);

/* tf.nn.dropout keep mask (tf_util.py:1738-1740): mask[i] = 1[u_i < keep], u from a counter-based
 * generator keyed by (seed, hyper step, i).  Tests inject masks instead. */
int t3d_dropout_mask(float* mask, int64_t n, float keep_prob, uint32_t seed, const float* hyper,
                     t3d_stream_t stream);

/* ---- two independent small launches in one ---------------------------------------------------------------------------
 * a and b must not depend on each other (the box / T-Net backward and the segmentation-net backward: semisup_models.py:150-151).
 * Same arguments, same results (bit for bit) as the two stand-alone calls; what is saved is one kernel boundary (~3-5 us).
 * Kinds: t3d_bn_bwd_finalize (up to 512 row tiles), t3d_fc_bwd, t3d_fc_dinput, t3d_dy_colsum. */
enum { T3D_SMALL_BN_BWD_FINALIZE = 1, T3D_SMALL_FC_BWD = 2, T3D_SMALL_FC_DINPUT = 3, T3D_SMALL_DY_COLSUM = 4,
       T3D_SMALL_BN_FWD_FINALIZE = 5, T3D_SMALL_FC_FWD = 6, T3D_SMALL_POOL_BWD_MID = 7 };      /* 5 .. 7: rider sets only */
/* the arguments of t3d_pool_bwd_mid as a struct (kind 7: a WIDE rider -- hundreds of workgroups, alone in its set, no barrier) */
typedef struct {
  const float* slab_base;
  float* grad_base;
  const t3d_slab_desc* table_dev;
  int n_tensors, max_numel;
  t3d_pool_sparse_rows_args sparse;
} t3d_pool_bwd_mid_args;
typedef struct {
  int kind;
  int depends;            /* rider sets: 1 = reads what the PREVIOUS op of its set wrote (a barrier among the set's workgroups goes in
                             front of it); 0 = independent of it.  Ignored by t3d_small_pair. */
  union {
    t3d_bn_bwd_finalize_args bn_bwd;
    t3d_fc_bwd_args fc_bwd;
    t3d_fc_dinput_args fc_dinput;
    t3d_dy_colsum_args dy_colsum;
    t3d_bn_fwd_finalize_args bn_fwd;
    t3d_fc_fwd_args fc_fwd;
    t3d_pool_bwd_mid_args mid;
  } u;
} t3d_small_op;
int t3d_small_pair(const t3d_small_op* a, const t3d_small_op* b, t3d_stream_t stream);

/* ---- riders: small launches of one chain inside the GEMM launches of an independent chain ---------------------------------------
 * The reference runs its graph op by op (sess.run, train_semisup.py:405-411) and TensorFlow's executor overlaps whatever is
 * independent.  Here the independence that matters is one: nothing the T-Net / box net compute (forward, loss, backward) reaches the
 * segmentation net's backward -- `mask` is a hard comparison (semisup_models.py:150-151).  A rider SET is a run of small ops
 * (batch-norm finalizers, FC heads and their backward, column sums: the kinds above) executed in order by the first `n_wg`
 * 256-thread workgroups of a GEMM launch that does not depend on them and that they do not depend on; the GEMM's own tiles are the
 * workgroups behind.  An op with `depends` waits for its predecessor behind a barrier among the rider workgroups (agent-scope
 * release / acquire; the GEMM's workgroups never wait).  Results are bit-identical to the stand-alone launches of the same ops.
 *   ops        the set, in chain order (passed to the kernel by value: at most T3D_RIDER_MAX_OPS of them)
 *   n_wg       rider workgroups (<= 32; a set that is ONE t3d_pool_bwd_mid -- slab reduction + sparse arg-max rows of a pooled layer's
 *              backward, hundreds of latency-bound workgroups that use a fraction of the chip -- gets one workgroup per block) and
 *   lds_bytes  dynamic LDS the rider bodies need: both filled in by t3d_riders_plan
 *   sync       DEVICE, 2 * T3D_RIDER_MAX_OPS + 2 zero-initialised 32-bit words owned by this set (self-resetting: graph replays need
 *              no memset); the last but one word is set to 1 if a barrier ever gave up waiting (poisoned state; results invalid)
 * The `_r` launchers take `riders == NULL` (then they ARE the plain entry points).  Where the kernel variant a launch selects has no
 * rider form (bf16 kernels, the one-pass backward, the activation-resident pooled forward) the set runs as its own launch in front
 * of the GEMM (t3d_run_riders) -- same results, nothing hidden. */
#define T3D_RIDER_MAX_OPS 10
typedef struct {
  t3d_small_op ops[T3D_RIDER_MAX_OPS];
  int n_ops;
  int n_wg;
  int lds_bytes;
  unsigned* sync;
} t3d_rider_set;
/* validates the set's ops (kinds, shapes that can ride: B <= 32 for the FC ops, <= 512 row tiles for the finalizers) and fills in n_wg and
 * lds_bytes; T3D_ERR_ARG / T3D_ERR_SHAPE if one of them cannot ride */
int t3d_riders_plan(t3d_rider_set* riders);
int t3d_run_riders(const t3d_rider_set* riders, t3d_stream_t stream);      /* the set as a launch of its own */
int t3d_pointmlp_fwd_r(const t3d_pointmlp_fwd_args* args, const t3d_rider_set* riders, t3d_stream_t stream);
int t3d_pointmlp_wgrad_r(const t3d_pointmlp_wgrad_args* args, const t3d_rider_set* riders, t3d_stream_t stream);
int t3d_pointmlp_bwd_r(const t3d_pointmlp_dgrad_args* dgrad, const t3d_pointmlp_wgrad_args* wgrad, const t3d_rider_set* riders,
                       t3d_stream_t stream);
int t3d_pool_bwd_stage1_r(const t3d_pointmlp_gram_args* gram, const t3d_act_colsum_args* colsum, const t3d_pool_bwd_prep_args* prep,
                          const t3d_rider_set* riders, t3d_stream_t stream);
int t3d_pool_bwd_stage2_r(const t3d_pool_wgrad_finish_args* finish, const t3d_pointmlp_dgrad_gram_args* dgrad,
                          const t3d_rider_set* riders, t3d_stream_t stream);
/* Would the `_r` launcher run a rider set INSIDE the GEMM launch for these arguments (1), or as a launch of its own in front of it
 * (0)?  Answered by the launchers' own dispatch (kernel variant, T3D_X3, tile widths, ...), so a scheduler needs no copy of those
 * rules.  Negative: the arguments would be rejected. */
int t3d_pointmlp_fwd_hosts_riders(const t3d_pointmlp_fwd_args* args);
int t3d_pointmlp_wgrad_hosts_riders(const t3d_pointmlp_wgrad_args* args);
int t3d_pointmlp_bwd_hosts_riders(const t3d_pointmlp_dgrad_args* dgrad, const t3d_pointmlp_wgrad_args* wgrad);
int t3d_pool_bwd_stage1_hosts_riders(const t3d_pointmlp_gram_args* gram, const t3d_act_colsum_args* colsum, const t3d_pool_bwd_prep_args* prep);
int t3d_pool_bwd_stage2_hosts_riders(const t3d_pool_wgrad_finish_args* finish, const t3d_pointmlp_dgrad_gram_args* dgrad);

#ifdef __cplusplus
}
#endif
#endif /* T3D_H_ */

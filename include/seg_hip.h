/* seg_hip.h - C ABI of the MI355X (gfx950) UNet-family hot path.
 *
 * The reference (Sakib1263/TF-1D-2D-Segmentation-End2EndPipelines) has NO plugin / operator / FFI interface: its
 * hot path is the graph of tf.keras layers built in TensorFlow/2DCNN/models/unet_variants.py:7-154,277-343,
 * 459-487,750-792,936-1115 (1D: TensorFlow/1DCNN/Models/unet_variants.py:53-319,544-645, BCDUNet.py:6-174) and
 * executed by Keras fit/predict (2DCNN/Train.py:394-415, 2DCNN/Test.py:153,161).  Each entry point below
 * replaces the TensorFlow-runtime kernel(s) behind one of those Keras layers; the layer is cited per function.
 *
 * Conventions
 *   - all functions return 0 on success, a negative SEG_ERR_* otherwise; seg_last_error() gives the message
 *     (thread-local).  No allocation, no synchronisation, no ownership transfer inside: every call only
 *     enqueues kernels on `stream` (a hipStream_t), so calls may be captured into a hipGraph.
 *   - activations are NHWC fp16 with a channel stride `cs` (elements per pixel, multiple of 8); channels beyond
 *     the logical channel count are zero.  A 1D signal is H = 1.
 *   - "virtual concat": a conv input is a list of up to SEG_MAX_SEG tensors of identical N,H,W; the K loop walks
 *     them, the concatenated tensor is never materialised (Concat_Block, unet_variants.py:27-32).
 *   - GEMM weights are [rows][T][Kpad] (rows = GEMM N extent, T = taps in the kernel window, Kpad = sum of the
 *     segment channel counts each rounded up to 32); fp16 for the MFMA operand, fp32 for master copy / gradient.
 */
#ifndef SEG_HIP_H
#define SEG_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* seg_stream_t; /* hipStream_t */

enum { SEG_OK = 0, SEG_ERR_ARG = -1, SEG_ERR_LAUNCH = -2, SEG_ERR_UNSUPPORTED = -3 };
enum { SEG_ACT_NONE = 0, SEG_ACT_RELU = 1, SEG_ACT_LEAKY = 2, SEG_ACT_SIGMOID = 3, SEG_ACT_SOFTMAX = 4 };
/* the 16 names of 2DCNN/utils/tf_losses.py:4-44, Keras 2.x formulas (eps 1e-7).  CCE / KLD / CATHINGE / COSINE /
 * SPARSE_CCE reduce over the channel axis and average over pixels, the others average over every element. */
enum { SEG_LOSS_MSE = 0, SEG_LOSS_BCE = 1, SEG_LOSS_CCE = 2, SEG_LOSS_MAE = 3, SEG_LOSS_MAPE = 4, SEG_LOSS_MSLE = 5,
       SEG_LOSS_HUBER = 6, SEG_LOSS_LOGCOSH = 7, SEG_LOSS_POISSON = 8, SEG_LOSS_HINGE = 9, SEG_LOSS_SQHINGE = 10,
       SEG_LOSS_FOCAL = 11, SEG_LOSS_KLD = 12, SEG_LOSS_CATHINGE = 13, SEG_LOSS_COSINE = 14, SEG_LOSS_SPARSE_CCE = 15 };

#define SEG_MAX_SEG 8
#define SEG_MAX_TAPS 32        /* taps of all sub-pixel classes together: a 1-D kernel_size of up to 32 (the reference takes any) */
#define SEG_MAX_CLS 4

const char* seg_last_error(void);
int seg_abi_version(void);      /* 28 bits of sha256(this header), compiled in by build_ext.py; the loader recomputes and compares */

/* Geometry of one (transposed / strided) convolution as an implicit GEMM over "logical pixels" (n, i, j),
 * i < Hl, j < Wl, evaluated for ncls output classes (1 for a convolution, 4 = sub-pixel parity classes of a
 * 4x4/s2 transposed conv, 2 for the 1D k2/s2 one).  For class c and tap t (entry e = c*ntaps + t):
 *   input  pixel = (i*isy + tap_dy[e], j*isx + tap_dx[e])   (zero outside [0,Hin) x [0,Win))
 *   output pixel = (i*osy + oy[c],     j*osx + ox[c])       (always inside [0,Hout) x [0,Wout))
 *   weight slice = tap_w[e]  (index into the T_total taps of the weight tensor)                                */
typedef struct seg_geom {
  int32_t N, Hin, Win;
  int32_t Hl, Wl;
  int32_t isy, isx;
  int32_t Hout, Wout;
  int32_t osy, osx;
  int32_t ncls, ntaps;
  int32_t T_total;
  int32_t oy[SEG_MAX_CLS], ox[SEG_MAX_CLS];
  int32_t tap_dy[SEG_MAX_TAPS], tap_dx[SEG_MAX_TAPS], tap_w[SEG_MAX_TAPS];
} seg_geom;

/* ---- implicit-GEMM forward: Conv1D/2D (unet_variants.py:9), Conv1D/2DTranspose (:19), the ConvLSTM input conv
 * (:148) and - with transposed weights - every data gradient of those layers.
 *   out[pix][n] (+)= act( ep_scale[n] * sum_{seg,tap,c} in_seg[pix+tap][c] * w[n][tap_w][woff_seg + c] + bias[n] )
 * MFMA v_mfma_f32_32x32x16_f16, fp32 accumulate.  Optional per-channel partial sums of the pre-activation
 * value (sum, sum of squares) for training-mode BatchNormalization (:11).  ep_scale (NULL = 1) is the folded
 * inference-mode BatchNormalization: with ep_scale = gamma / sqrt(var + eps) and bias = beta + (b - mean) * ep_scale
 * (seg_bn_fold_inference) one launch is the whole Conv_Block of unet_variants.py:7-14 as model.predict runs it
 * (2DCNN/Test.py:153,161).                                                                                      */
#define SEG_VARIANT_A_HI 0x1000  /* OR-ed into `variant` of seg_igemm_fwd_f32 (variant 3 only): the A operand - the incoming gradient of a
                                    data-gradient launch - is read as its hi half, two MFMAs per product instead of three (the weight
                                    operand stays whole: a b ~= ah bh + ah bl / 2048); never for a forward pass                     */
#define SEG_VARIANT_B_HI 0x2000  /* ... or its B operand - the (transposed) weights - instead: a b ~= ah bh + al bh / 2048; dz keeps its 22 bits
                                    and its range (weights are O(1): no element of them sits below fp16's normal range)                */
#define SEG_VARIANT_IN_HI 0x1000 /* OR-ed into `variant` of seg_igemm_wgrad_f32 (variant 2, tiles of 64 output channels): the input operand
                                    (the activations) is read as its hi half - two MFMAs per product, dz stays whole             */
#define SEG_VARIANT_DZ_HI 0x2000 /* ... together with SEG_VARIANT_IN_HI: dz as its hi half as well - ONE MFMA per product (a b ~= ah bh).  The
                                    weight gradient ends a chain: its rounding lands once in dW and is inherited by no other gradient */
typedef struct seg_igemm_fwd_args {
  seg_geom g;
  int32_t nseg;
  int32_t seg_c[SEG_MAX_SEG];    /* channels of segment s taking part (multiple of 8)                      */
  int32_t seg_cs[SEG_MAX_SEG];   /* channel stride of segment s                                            */
  int32_t seg_woff[SEG_MAX_SEG]; /* first K column of segment s in the weight (multiple of 32)             */
  const void* seg_ptr[SEG_MAX_SEG];
  const void* w;                 /* fp16 [n_rows][T_total][Kpad]                                           */
  int32_t Kpad;
  int32_t n_rows;                /* GEMM N extent (multiple of 8)                                          */
  int32_t n_valid;               /* outputs n >= n_valid are written as 0 (channel padding)                */
  void* out;                     /* fp16 [N][Hout][Wout][out_cs], columns out_coff .. out_coff+n_rows      */
  int32_t out_cs, out_coff;
  const float* bias;             /* [n_rows] or NULL                                                       */
  int32_t act;                   /* SEG_ACT_NONE / RELU / LEAKY / SIGMOID                                  */
  float alpha;
  int32_t accumulate;            /* out += result instead of out = result                                  */
  float* stats;                  /* NULL or fp32 [nparts][2][n_rows], nparts = ncls * ceil(M / tile_m)     */
  int32_t tile_m, tile_n;        /* 128x128 | 128x64 | 64x128 | 64x64 | 256x64                              */
  /* split output ("un-concat", used by data gradients of a virtual concat): when nout > 0 the N columns are cut
   * into nout consecutive spans; span o covers out_span[o] columns (multiple of 32) of which the first
   * out_cs_o[o] are stored to tensor out_ptr[o] (channel stride out_cs_o[o], NULL = discard), accumulating when
   * out_acc[o] != 0.  out / out_cs / out_coff / accumulate are ignored in that mode.                          */
  int32_t nout;
  int32_t out_span[SEG_MAX_SEG];
  int32_t out_cs_o[SEG_MAX_SEG];
  int32_t out_acc[SEG_MAX_SEG];
  void* out_ptr[SEG_MAX_SEG];
  /* fused activation backward of the layer that produced span o (split-output mode): when out_y[o] != NULL the value
   * stored to (or accumulated into) out_ptr[o] is first multiplied by act'(y), y = out_y[o][pixel][channel] being that
   * layer's OUTPUT (same layout as out_ptr[o]), act = out_dact[o] (RELU / LEAKY / SIGMOID), slope out_dalpha[o].
   * Replaces a separate seg_act_bwd pass when the span's tensor has no other consumer (Conv2DTranspose + LeakyReLU,
   * unet_variants.py:17-24).                                                                                     */
  const void* out_y[SEG_MAX_SEG];
  int32_t out_dact[SEG_MAX_SEG];
  float out_dalpha[SEG_MAX_SEG];
  int32_t variant;               /* 0: per-tap kernels (any geometry); 3: halo-tile kernel - fp16, 3x3 taps with
                                    offsets in {-1,0,1}^2, stride 1, 'same'; a block owns a 256-pixel patch, so
                                    tile_m must be 256 (tile_n 64 | 128) and seg_fwd_halo_patch(&g) != 0          */
  const float* ep_scale;         /* NULL, or [n_rows] per-channel factor applied to the contraction before bias (folded BN) */
  /* split of the reduction (variant 0, tiles 128x128 / 128x64 / 256x64, taps in {1,2,3,4,9,16}): splitk > 1 launches splitk blocks
   * per output tile, each contracting a range of the K chunks into `partial` (fp32 accumulator images,
   * seg_igemm_fwd_workspace_bytes() bytes, no initialisation needed), and a second launch that adds the images in order and
   * runs the epilogue (bias, activation, statistics, spans) - deterministic.  For grids that leave CUs idle: the 8x8 / 16x16
   * levels of a depth-5 encoder (M = 1 024 .. 4 096 pixels against K = 4 608 .. 18 432).  0 / 1: no split.                       */
  int32_t splitk;
  float* partial;
} seg_igemm_fwd_args;
int seg_igemm_fwd(const seg_igemm_fwd_args* a, seg_stream_t stream);
/* floats of `stats` for these arguments ([ncls * ceil(M / tile_m)][2][n_rows]); the library allocates nothing itself */
int64_t seg_igemm_fwd_stats_floats(const seg_igemm_fwd_args* a);
/* bytes of `partial` for these arguments (0 when splitk <= 1) */
int64_t seg_igemm_fwd_workspace_bytes(const seg_igemm_fwd_args* a);
/* patch width (32: 8x32 pixels, 16: 16x16 pixels) the variant-3 kernel uses for this geometry; 0: not eligible   */
int seg_fwd_halo_patch(const seg_geom* g);

/* ---- implicit-GEMM weight gradient of the same layers:
 *   dw[n][tap_w][woff_seg + c] += scale * (*scale_ptr) * sum_pix dz[outpix][n] * in_seg[pix+tap][c]
 * reduction over pixels split `splitk` ways, fp32 atomics into dw.                                              */
typedef struct seg_igemm_wgrad_args {
  seg_geom g;
  int32_t nseg;
  int32_t seg_c[SEG_MAX_SEG], seg_cs[SEG_MAX_SEG], seg_woff[SEG_MAX_SEG];
  const void* seg_ptr[SEG_MAX_SEG];
  const void* dz;                /* fp16 [N][Hout][Wout][dz_cs], columns dz_coff .. dz_coff+n_rows         */
  int32_t dz_cs, dz_coff;
  float* dw;                     /* fp32 [n_rows][T_total][Kpad]                                           */
  int32_t Kpad, n_rows;
  int32_t splitk;
  float scale;
  const float* scale_ptr;        /* device scalar or NULL                                                  */
  int32_t tile_m, tile_n;        /* rows (n) x K-columns tile: 128 or 64                                   */
  int32_t variant;               /* 0: ds_read_b64_tr_b16 operand reads, 1: transposing LDS stores,
                                    2: LDS-DMA halo tiles (3x3 stride-1 'same' only, fp16; tile_m/tile_n ignored:
                                       64 rows x 64-channel planes x 9 taps per block, splitk splits the
                                       64-pixel patches; requires seg_wgrad_halo_patch(&g) != 0)              */
  float* bias_part;              /* optional, variant 2 on a transposed conv with seg_wgrad_halo_class_groups(&g) == 2:
                                    per-block partial column sums of dz (the bias gradient before the 1/S scale),
                                    fp32 [2 * splitk][2][n_rows] - row (py * splitk + z), plane 0 = sum, plane 1 = 0,
                                    the layout seg_bn_bwd_finalize reduces (dgamma = NULL, dbeta = the bias gradient).
                                    Parity mode (_f32): variant 2 on any 4-class transposed conv, [4 * splitk][2][n_rows],
                                    row (class * splitk + z).  Must be NULL otherwise.                        */
  int32_t overwrite;             /* variant 2 with splitk == 1 only: dw is known to hold zeros where this call writes (the
                                    engine clears the gradient buffer once per step), so the single writer of each tile
                                    stores instead of read-modify-writing - deep layers (8x8 / 16x16 images, 2048-channel
                                    kernels) are otherwise bound by that epilogue.  0: always accumulate                  */
  float* partial;                /* optional workspace for a DETERMINISTIC pixel split, fp32 [splitk][n_rows * T_total * Kpad],
                                    zero-filled once by the caller and owned by this layer: with splitk > 1 every block stores
                                    its partial tile there (plain stores, no atomics) and a second launch inside this call adds
                                    scale * sum_z partial[z] to dw in a fixed order - two runs give bit-identical gradients.
                                    NULL: split blocks add into dw with float atomics (order-dependent last bits).            */
} seg_igemm_wgrad_args;
int seg_igemm_wgrad(const seg_igemm_wgrad_args* a, seg_stream_t stream);
/* bytes of the zero-initialised `partial` workspace for these arguments (splitk * n_rows * T_total * Kpad floats; 0 when
 * splitk == 1): the caller allocates it once per layer, the library allocates nothing                                   */
int64_t seg_igemm_wgrad_workspace_bytes(const seg_igemm_wgrad_args* a);
/* Largest operand (bytes, exclusive: 1.875 GiB) the LDS-DMA kernel families address - one buffer resource per tensor, 32-bit byte
 * offsets: seg_igemm_fwd variant 3 and seg_igemm_wgrad variant 2 refuse larger inputs / gradients / weights; seg_igemm_fwd variant 0
 * then runs its register-staged kernel (64-bit addresses, tile_m <= 128, no splitk), seg_igemm_wgrad variants 0 / 1 likewise.      */
int64_t seg_lds_dma_operand_limit(void);
/* patch width (32 / 16 pixels, patch = 64 pixels) the variant-2 kernel would use for this geometry; 0: not eligible */
int seg_wgrad_halo_patch(const seg_geom* g);
/* blocks along the class axis of the variant-2 kernel (1: conv; 2: transposed conv on the class-pair kernel; 4)     */
int seg_wgrad_halo_class_groups(const seg_geom* g);
/* taps one block of seg_igemm_wgrad handles for this tile (layers with <= 64 output channels use a multi-tap
 * kernel that reads the dz tile once for 3 / 4 / 9 taps); the host sizes `splitk` with it.                       */
int seg_wgrad_taps_per_block(int tile_m, int tile_n, int ntaps, int variant);

/* ---- BatchNormalization() (unet_variants.py:11,70,72,76,96,99), eps 1e-3, momentum 0.99 ------------------ */
/* per-channel partial sums of an existing tensor: stats[part][0][c] = sum x, [1][c] = sum x^2 over the part   */
int seg_bn_stats(const void* x, int64_t npix, int32_t cs, int32_t coff, int32_t C, float* stats, int32_t nparts,
                 int32_t stats_ld, int32_t stats_coff, seg_stream_t stream);
/* reduce the partials -> batch mean / biased var; scale = gamma*rsqrt(var+eps), shift = beta - mean*scale;
 * moving stats update (unbiased variance when unbiased_moving != 0: the fused 4-D Keras path).  training == 0:
 * scale/shift from the moving statistics, nothing else touched.                                               */
int seg_bn_finalize(const float* stats, int32_t nparts, int32_t C, int64_t count, const float* gamma,
                    const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                    int32_t unbiased_moving, int32_t training, float* mean, float* invstd, float* scale,
                    float* shift, seg_stream_t stream);
/* inference-mode BatchNormalization folded into the conv in front of it: scale[c] = gamma / sqrt(moving_var + eps),
 * shift[c] = beta + (conv_bias[c] - moving_mean[c]) * scale[c]  (conv_bias may be NULL) - the ep_scale / bias operands of
 * seg_igemm_fwd for a one-launch Conv_Block                                                                     */
int seg_bn_fold_inference(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                          const float* conv_bias, int32_t C, float* scale, float* shift, seg_stream_t stream);
/* y = act(x*scale + shift) (channels >= c_valid -> 0); optional fused MaxPooling2D(ph,pw) second output
 * (unet_variants.py:790).                                                                                     */
int seg_bn_act_apply(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff,
                     int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale,
                     const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw,
                     seg_stream_t stream);
/* the same with a ONE-CHANNEL 1x1 OUTPUT HEAD on the tensor riding along (a deep-supervision level of UNet++, unet_variants.py:300-302,
 * or the model output, :1106): head->y[pix] = act_h(sum_c y[pix][c] * head->w[c] + head->b[0]) comes out of this pass - the lanes of a
 * pixel reduce their partial dot products - instead of a seg_head_fwd launch that reads the tensor again (w / b / act / y per head k < n).
 * C / 8 a power of two <= 64.                                                                                                      */
typedef struct { int32_t n; const float* w[2]; const float* b[2]; int32_t act[2]; float* y[2]; } seg_head_fwd_desc;   /* n = 1 | 2 heads (UNet++'s X0d feeds level d AND the output) */
int seg_bn_act_apply_head(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff,
                          int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale,
                          const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw,
                          const seg_head_fwd_desc* head, seg_stream_t stream);
/* backward, pass 1: g = dy * act'(y), y = x*scale+shift; partial[part][0][c] = sum g, [1][c] = sum g*xhat       */
int seg_bn_act_bwd_reduce(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                          int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int32_t act, float alpha, float* partial,
                          int32_t nparts, seg_stream_t stream);
/* pass 1b: dgamma += gs * sum g*xhat, dbeta += gs * sum g (gs = gscale * *gscale_ptr); c1 = sum g / count,
 * c2 = sum g*xhat / count                                                                                     */
int seg_bn_bwd_finalize(const float* partial, int32_t nparts, int32_t C, int64_t count, float* dgamma,
                        float* dbeta, float gscale, const float* gscale_ptr, float* c1, float* c2,
                        seg_stream_t stream);
/* passes 1 and 2 with the data gradient of such heads INJECTED: the gradient read is dy[pix][c] + gs * sum_k dpre_k[pix] * inj->w[k][c]
 * (dpre_k = inj->dy[k] * act_h'(inj->y[k]) of head k, gs = *inj->gs_ptr or 1) - the head's seg_head_bwd does not write its dx into the
 * gradient tensor first (a write of the whole tensor and a read-modify-write by the data gradient that follows).  dy == NULL:
 * the head is the tensor's only consumer.                                                                                       */
typedef struct { int32_t n; const float* dy[2]; const float* y[2]; const float* w[2]; int32_t act[2]; const float* gs_ptr; } seg_head_inject;
int seg_bn_act_bwd_reduce_inj(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                              int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift,
                              const float* mean, const float* invstd, int32_t act, float alpha, float* partial,
                              int32_t nparts, const seg_head_inject* inj, seg_stream_t stream);
int seg_bn_act_bwd_apply_inj(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                             int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate,
                             const seg_head_inject* inj, seg_stream_t stream);
/* pass 2 for TWO Conv_Blocks that received the same dOut (the 1x1 shortcut block and the 3x3 block of a ResPath iteration,
 * unet_variants.py:110-112: the add fans one gradient out): dy [npix][C] dense is read once, sets[j].dx = the block's dz (may
 * alias dy), from its raw conv output sets[j].x and its own coefficients (training-mode form).                           */
typedef struct {
  const void* x; void* dx;
  const float *scale, *shift, *mean, *invstd, *c1, *c2;
  int32_t act; float alpha;
} seg_bn_bwd_set;
int seg_bn_act_bwd_apply2(const void* dy, const seg_bn_bwd_set* sets, int64_t npix, int32_t C, seg_stream_t stream);
/* pass 2: dx = scale * (g - c1 - xhat*c2)   (training)   or   dx = scale * g   (c1 == NULL: inference BN)     */
int seg_bn_act_bwd_apply(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                         int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C,
                         const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate,
                         seg_stream_t stream);

/* ---- Activation layers without BN (unet_variants.py:23 LeakyReLU after Conv2DTranspose):
 * dx = dy * act'(y) given the activation OUTPUT y; optional bias gradient dbias[c] += gs * sum_pix dx          */
int seg_act_bwd(const void* dy, const void* y, void* dx, int64_t npix, int32_t cs, int32_t C, int32_t act,
                float alpha, float* dbias, float gscale, const float* gscale_ptr, seg_stream_t stream);

/* ---- MaxPooling1D/2D (unet_variants.py:790), stride = pool size ------------------------------------------- */
int seg_maxpool_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw,
                    seg_stream_t stream);
/* dx (+)= dy routed to the first maximum of each window (TensorFlow MaxPoolGrad tie rule)                      */
int seg_maxpool_bwd(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W,
                    int32_t cs, int32_t ph, int32_t pw, int32_t accumulate, seg_stream_t stream);

/* ---- UpSampling2D(bilinear) (unet_variants.py:37, half-pixel centres, edge clamp) / UpSampling1D nearest     */
int seg_upsample_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw,
                     int32_t bilinear, seg_stream_t stream);
int seg_upsample_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh,
                     int32_t fw, int32_t bilinear, int32_t accumulate, seg_stream_t stream);

/* ---- add([..]) / Add() (+ optional activation) (unet_variants.py:55,73,97,112): y = act(sum_i x_i), n <= 4.  Dense tensors of
 * nelem elements in rows of cs channels; channels >= c_valid are PADDING and are written as zero whatever the activation makes of 0
 * (sigmoid(0) = 0.5 would otherwise reach the padding columns of the next conv's weight gradient)                */
int seg_add_act_fwd(const void* const* xs, int32_t n, void* y, int64_t nelem, int32_t cs, int32_t c_valid, int32_t act, float alpha,
                    seg_stream_t stream);
/* dx_i (+)= dy * act'(y) for every i with dxs[i] != NULL; accumulate is a bit mask over i                      */
int seg_add_act_bwd(const void* dy, const void* y, void* const* dxs, int32_t n, int64_t nelem, int32_t act,
                    float alpha, int32_t accumulate_mask, seg_stream_t stream);
/* The tail of a MultiResBlock / ResPath, BatchNormalization(ReLU(a + b)) (unet_variants.py:97-99, :118-120), as fused passes.
 * A VIRTUAL CONCATENATION (seg_cat_desc) names up to four tensors whose channel counts need not be multiples of 8 - MultiResBlock's
 * cat[c3x3, c5x5, c7x7] (unet_variants.py:95): 5 + 10 + 16 channels at W = 32 -; piece k supplies channels [coff, coff + C) of the
 * concatenation (coff = the sum of the C before it).  The kernels fetch the pieces' 8-channel groups whole and assemble the rows of
 * the concatenation in LDS; the concatenated tensor never exists in memory.
 * forward  y = act(sum_k pre_k(x_k) + [cat * cat_scale + cat_shift]), zero beyond c_valid, plus per-channel partial sums of y and
 *          y^2 (stats[nparts][2][cs], the operand of seg_bn_finalize for the BatchNormalization that follows).
 *          pre_k(x) = x, or relus[k] ? max(x * scales[k] + shifts[k], 0) : x * scales[k] + shifts[k] ([cs] floats each): an input
 *          that is a Conv_Block's output (conv -> BN -> ReLU, unet_variants.py:7-14) with no other consumer is normalised on load
 *          from the conv's raw output and never written.  cat_scale / cat_shift = the BatchNormalization over the concatenation
 *          (MultiResBlock's BN(cat): unet_variants.py:96), NULL = identity.  n = 0 and y = NULL with a cat: only the statistics of
 *          the concatenation (the operand of THAT BatchNormalization's seg_bn_finalize).  Tensors are dense [npix][cs], cs <= 2048.
 * backward dx_k (+)= [x > 0] * scale * (dy - c1 - xhat * c2) for every input k of the add with dxs[k] != NULL (x = the
 *          BatchNormalization's input = the ReLU's output; relu_mask = 0 without the ReLU): replaces seg_bn_act_bwd_apply +
 *          seg_add_act_bwd.  With a cat: additionally cat_partial[nparts][2][cs] = per-block sums of dx and dx * xhat_cat
 *          (xhat_cat from cat_mean / cat_invstd) - the operand of seg_bn_bwd_finalize for the BatchNormalization over the
 *          concatenation, whose input gradient seg_cat_bn_bwd_scatter then writes to the pieces:
 *          piece.dx (+)= scale * (g - c1 - xhat_cat * c2) restricted to the piece's channels, whole groups, padding zero.      */
typedef struct {
  const void* x;      /* piece [npix][cs]: its first C channels belong to the concatenation                                 */
  void* dx;           /* seg_cat_bn_bwd_scatter: gradient of the piece, same layout (NULL: not wanted)                      */
  int32_t cs, C;
  int32_t coff;       /* first channel of the piece inside the concatenation                                                */
  int32_t acc;        /* seg_cat_bn_bwd_scatter: add to dx instead of storing                                               */
} seg_cat_piece;
typedef struct { int32_t n; seg_cat_piece p[4]; } seg_cat_desc;
/* An input of the add that is a Conv_Block normalised on load (see `forward`): the gradient the backward launch writes is that
 * block's dOut, so the launch also delivers the two sums of the block's own BatchNormalization backward - per-block partial rows
 * partial[nparts][2][cs] of sum dOut act'(.) and sum dOut act'(.) xhat_z, the operand of seg_bn_bwd_finalize - from the block's
 * raw conv output z (replaces a seg_bn_act_bwd_reduce pass over dOut and z).                                                       */
typedef struct {
  const void* z;                 /* raw conv output [npix][cs]                                                              */
  const float *scale, *shift;    /* the block's BatchNormalization as y = z * scale + shift (activation mask)               */
  const float *mean, *invstd;    /* its batch statistics (xhat_z)                                                           */
  int32_t relu;                  /* the block ends in a ReLU                                                                */
  float* partial;
} seg_lazy_bn;
int seg_add_act_stats_fwd(const void* const* xs, const float* const* scales, const float* const* shifts, const int32_t* relus, int32_t n,
                          const seg_cat_desc* cat, const float* cat_scale, const float* cat_shift, void* y, int64_t npix, int32_t cs,
                          int32_t c_valid, int32_t act, float alpha, float* stats, int32_t nparts, int32_t stats_ld, int32_t stats_coff,
                          seg_stream_t stream);
/* stats_ld / stats_coff (0 / 0 = the tensor's own rows [nparts][2][cs]): the partial rows have stats_ld columns and this launch fills
 * columns stats_coff .. stats_coff + c_valid - with n = 1, one pre-affine input and act = none the call is "bn_act_apply that also
 * delivers the statistics of what it writes": a piece of a virtual concatenation (c3 / c5 / c7 of a MultiResBlock) hands its sums to
 * the BatchNormalization over the concatenation on its way out (same nparts for all pieces), no pass over the pieces afterwards.   */
int seg_bn_bwd_apply_fanout(const void* dy, const void* x, void* const* dxs, int32_t n, int32_t accumulate_mask, int64_t npix,
                            int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2,
                            int32_t relu_mask, const seg_cat_desc* cat, const float* cat_mean, const float* cat_invstd,
                            float* cat_partial, const seg_lazy_bn* lazy, int32_t nlazy, int32_t nparts, seg_stream_t stream);
int seg_cat_bn_bwd_scatter(const void* g, const seg_cat_desc* cat, int64_t npix, int32_t cs, const float* scale, const float* mean,
                           const float* invstd, const float* c1, const float* c2, seg_stream_t stream);
/* widest row (cs, channels) the three fused-tail entry points above take: one column group of 8 channels per thread of a block.
 * Wider tails (MultiResUNet with model_width >= 128 at depth 5) are lowered to the separate passes by the host planner.            */
int32_t seg_tail_row_limit(void);

/* ---- channel-slice copy (materialised concatenate(axis=-1), unet_variants.py:95, when a consumer cannot walk the
 * pieces: BatchNormalization over cat[c3x3,c5x5,c7x7] in MultiResBlock).  dst[p][dst_coff+c] (+)= src[p][src_coff+c],
 * c < C; element granular (the pieces need not start on 8-channel boundaries).  The backward is the same call with
 * src/dst swapped and accumulate set.                                                                         */
int seg_copy_channels(const void* src, int32_t src_cs, int32_t src_coff, void* dst, int32_t dst_cs, int32_t dst_coff,
                      int32_t C, int64_t npix, int32_t accumulate, seg_stream_t stream);

/* ---- attention gating multiply (unet_variants.py:81): y[p][c] = a[p][c] * r[p][0]                           */
int seg_mul_bcast_fwd(const void* a, const void* r, void* y, int64_t npix, int32_t cs, int32_t r_cs,
                      seg_stream_t stream);
/* da (+)= dy * r ; dr[p][0] (+)= sum_c dy*a                                                                   */
int seg_mul_bcast_bwd(const void* dy, const void* a, const void* r, void* da, void* dr, int64_t npix, int32_t cs,
                      int32_t r_cs, int32_t acc_a, int32_t acc_r, seg_stream_t stream);

/* ---- ConvLSTM2D/1D, sequence length 1, zero state (unet_variants.py:148): z = [i|f|c|o] (4F channels),
 * h = hs(o) * tanh(hs(i) * tanh(c)), hs(x) = clip(0.2x+0.5, 0, 1).                                           */
int seg_convlstm_gate_fwd(const void* z, int32_t z_cs, void* h, int32_t h_cs, int64_t npix, int32_t F,
                          seg_stream_t stream);
int seg_convlstm_gate_bwd(const void* dh, int32_t h_cs, const void* z, int32_t z_cs, void* dz, int64_t npix,
                          int32_t F, seg_stream_t stream);

/* ---- output heads: Conv2D(output_nums, 1x1, activation=final_activation, name='out') (unet_variants.py:1106)
 * and the deep-supervision Conv2D(1, 1x1, name='level{k}') heads (:136, :282, :340).  Cout <= 8.
 * y fp32 [npix][Cout] = act(x[p][:Cin] . w[co][:] + b[co]);  w fp32 [Cout][w_ld], input stride (sy, sx).       */
int seg_head_fwd(const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx,
                 const float* w, int32_t w_ld, const float* b, int32_t Cout, int32_t act, float* y,
                 seg_stream_t stream);
/* dy fp32 = gradient wrt the activated output; dx fp16 (+)= (*gs_ptr) * dpre . w ; dw, db += exact fp32 sums.
 * work: optional caller-owned scratch of work_elems floats (>= Cout * (w_ld + 1); 1024 * Cout * (w_ld + 1) lets the
 * weight-gradient pass use the whole chip): per-block partial rows + a finalize launch instead of same-address
 * atomics (deterministic, 4x faster on a 16 x 256 x 256 x 64 input).  NULL: atomics from <= 256 blocks.           */
int seg_head_bwd(const float* dy, const float* y, const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H,
                 int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, int32_t Cout, int32_t act,
                 void* dx, int32_t accumulate, const float* gs_ptr, float* dw, float* db, float* work,
                 int64_t work_elems, seg_stream_t stream);

/* ---- output heads with more than 8 channels (unet_variants.py:1106 with a large output_nums): the 1x1 conv runs as an
 * implicit GEMM (seg_igemm_fwd / _wgrad, logits in the storage type, channel stride cs); these are the fp32 activation
 * on top.  fwd: y[pix][k] = act(x[pix*cs + k]) (SEG_ACT_NONE / SIGMOID / SOFTMAX);  bwd: dx[pix*cs + k] (+)= gs[0] *
 * d act / d logit . dy, channels C..cs-1 are written as zeros.                                                      */
int seg_head_act_fwd(const void* x, int32_t cs, int64_t npix, int32_t C, int32_t act, float* y, seg_stream_t stream);
int seg_head_act_fwd_f32(const void* x, int32_t cs, int64_t npix, int32_t C, int32_t act, float* y, seg_stream_t stream);
int seg_head_act_bwd(const float* dy, const float* y, int64_t npix, int32_t C, int32_t act, const float* gs_ptr,
                     void* dx, int32_t cs, int32_t accumulate, seg_stream_t stream);
int seg_head_act_bwd_f32(const float* dy, const float* y, int64_t npix, int32_t C, int32_t act, const float* gs_ptr,
                         void* dx, int32_t cs, int32_t accumulate, seg_stream_t stream);

/* ---- losses used by the training harness (2DCNN/Train.py:322, utils/tf_losses.py:4-44): mean reduction.
 * loss_out[0] += loss ; dpred = d loss / d pred  (dpred may be NULL: value only)                                 */
int seg_loss_fwd_bwd(int32_t kind, const float* pred, const float* target, int64_t npix, int32_t C,
                     float* loss_out, float* dpred, seg_stream_t stream);
/* The same with a loss weight (model.compile(loss_weights=...), 1D notebook cell 35): loss_out[0] += weight * loss,
 * dpred = weight * d loss / d pred.  SEG_LOSS_SPARSE_CCE: target holds ONE class index per pixel (npix floats).
 * binary_flag (device, may be NULL = 1): HINGE / SQHINGE read labels as 2t-1 when it is non-zero - Keras converts
 * {0,1} labels to {-1,+1} only when every label of the batch is 0 or 1; seg_labels_binary computes that flag.    */
int seg_loss_fwd_bwd_w(int32_t kind, const float* pred, const float* target, int64_t npix, int32_t C, float weight,
                       const int32_t* binary_flag, float* loss_out, float* dpred, seg_stream_t stream);
int seg_labels_binary(const float* target, int64_t n, int32_t* flag, seg_stream_t stream);
/* ---- metrics of utils/tf_metrics.py: the mean-type ones are seg_loss_fwd_bwd_w with dpred = NULL (value only).
 * counts[0..3] += TP, FP, FN, TN with "pred > threshold" (BinaryAccuracy, Precision, Recall, TruePositives ...),
 * counts[4..7] += the same with "pred >= threshold" (BinaryIoU); a label is positive when > 0.5.               */
int seg_binary_counts(const float* pred, const float* target, int64_t n, float threshold, uint64_t* counts,
                      seg_stream_t stream);
/* counts[0] += pixels whose arg-max channel agrees (CategoricalAccuracy; sparse_target: one class index per pixel) */
int seg_argmax_match(const float* pred, const float* target, int64_t npix, int32_t C, int32_t sparse_target,
                     uint64_t* counts, seg_stream_t stream);
/* amax[0] = max(amax[0], max |x|)  (amax must be zeroed by the caller)                                         */
int seg_absmax(const float* x, int64_t n, float* amax, seg_stream_t stream);
/* gs[0] = 2^k with amax*2^k in [target/2, target), gs[1] = 1/gs[0]  (gs[0] = 1 when amax is 0 or not finite)   */
int seg_grad_scale(const float* amax, float target, float* gs, seg_stream_t stream);

/* ---- Keras Adam (utils/tf_optimizers.py:11): m,v update; p -= alpha_t * m / (sqrt(v) + eps); alpha_t is
 * lr*sqrt(1-b2^t)/(1-b1^t) computed by the host.  g is multiplied by gmul first (1/world_size for data
 * parallel).  Also refreshes the fp16 operand copy p16 (same indexing).  Non-finite gradients are skipped.     */
int seg_adam_step(float* p, const float* g, float* m, float* v, void* p16, int64_t n, float alpha_t, float b1,
                  float b2, float eps, float gmul, seg_stream_t stream);
/* the same with alpha_t read from device memory at run time: the form a captured hipGraph of the training step replays
 * (the host refreshes *alpha_ptr before each replay)                                                             */
int seg_adam_step_dev(float* p, const float* g, float* m, float* v, void* p16, int64_t n, const float* alpha_ptr,
                      float b1, float b2, float eps, float gmul, seg_stream_t stream);
/* The other optimisers of utils/tf_optimizers.py:12-26, Keras (>= 2.11) update rules with the registry's fixed
 * hyper-parameters; g is multiplied by gmul first, non-finite gradients are skipped, p16 as in seg_adam_step.
 * s1 / s2: two state buffers of n floats; c: 5 HOST floats computed per step:
 *   SGD       p -= c0 g                                                     c0 = lr
 *   RMSPROP   s2 = c1 s2 + (1-c1) g^2 ; p -= c0 g / sqrt(s2 + c2)            c0 = lr, c1 = rho 0.9, c2 = eps
 *   ADAGRAD   s2 += g^2 ; p -= c0 g / sqrt(s2 + c2)                          c0 = lr, c2 = eps; s2 starts at 0.1
 *   ADADELTA  s1 = c1 s1 + (1-c1) g^2 ; d = -sqrt(s2+c2) g / sqrt(s1+c2) ;
 *             s2 = c1 s2 + (1-c1) d^2 ; p += c0 d                            c0 = lr, c1 = rho 0.95, c2 = eps
 *   ADAMAX    s1 += (g-s1)(1-c1) ; s2 = max(c2 s2, |g|) ; p -= c0 s1/(s2+c3) c0 = lr/(1-b1^t), c1 = b1, c2 = b2, c3 = eps
 *   NADAM     s1 += 0.1 (g-s1) ; s2 += 0.001 (g^2-s2) ;
 *             p -= c0 (c1 s1 + c2 g) / (sqrt(s2 c3) + c4)                    c1 = u_{t+1}/(1-prod u_{1..t+1}), c2 = (1-u_t)/(1-prod u_{1..t}),
 *                                                                           c3 = 1/(1-b2^t), c4 = eps; u_k = 0.9 (1 - 0.5 * 0.96^k)
 *   FTRL      a = s1 + g^2 ; s2 += g - (sqrt(a)-sqrt(s1))/c0 * p ; s1 = a ;
 *             p = -s2 / (sqrt(a)/c0)                                        c0 = lr; s1 starts at 0.1 (lr_power -0.5, no l1/l2)   */
enum { SEG_OPT_SGD = 0, SEG_OPT_RMSPROP = 1, SEG_OPT_ADAGRAD = 2, SEG_OPT_ADADELTA = 3, SEG_OPT_ADAMAX = 4, SEG_OPT_NADAM = 5,
       SEG_OPT_FTRL = 6 };
int seg_optimizer_step(int32_t kind, float* p, const float* g, float* s1, float* s2, void* p16, int64_t n, const float* c,
                       float gmul, seg_stream_t stream);

/* ---- the forms the engine calls (SegModel.optimizer_step): as above, plus
 *   g_dtype    SEG_GRAD_F32: g is the fp32 gradient buffer the weight-gradient kernels accumulated into;
 *              SEG_GRAD_BF16: g is a bf16 slice that came back from the data-parallel all-reduce (dp.py; the sum over the
 *              ranks, widened to fp32 before the update - state and master weights stay fp32)
 *   alpha_ptr  (Adam) when not NULL the step size is alpha_t * *alpha_ptr, read on the device (hipGraph replay)
 *   nonfinite  when not NULL: *nonfinite += number of elements whose gradient was NaN / +-Inf (their update is skipped,
 *              as in the plain forms - Keras would propagate the NaN; here the count makes the overflow visible and
 *              harness.fit reacts by halving the fp16 loss scale)                                                       */
enum { SEG_GRAD_F32 = 0, SEG_GRAD_BF16 = 1 };
/*   p16_fmt    layout of the operand copy p16 (may be NULL): SEG_OPERAND_F16 = fp16 image, 2 B per element (throughput
 *              mode); SEG_OPERAND_SPLIT = the parity mode's split pair, 4 B per element (see "parity mode" below)           */
enum { SEG_OPERAND_F16 = 0, SEG_OPERAND_SPLIT = 1 };
int seg_adam_step_ex(float* p, const void* g, int32_t g_dtype, float* m, float* v, void* p16, int32_t p16_fmt, int64_t n, float alpha_t,
                     const float* alpha_ptr, float b1, float b2, float eps, float gmul, uint32_t* nonfinite, seg_stream_t stream);
int seg_optimizer_step_ex(int32_t kind, float* p, const void* g, int32_t g_dtype, float* s1, float* s2, void* p16, int32_t p16_fmt,
                          int64_t n, const float* c, float gmul, uint32_t* nonfinite, seg_stream_t stream);
/* y[i] = bf16(x[i]), round to nearest even, NaN / Inf preserved: packs a finished slice of the flat fp32 gradient buffer
 * for the data-parallel exchange (half the bytes over xGMI).  x 32-byte, y 16-byte aligned.               */
int seg_cast_f32_bf16(const float* x, void* y, int64_t n, seg_stream_t stream);

/* fp32 [rows][T][K] -> fp16 [K][T][rows_pad] (the data-gradient operand of the same layer)                     */
int seg_transpose_w(const float* w, void* wt, int32_t rows, int32_t T, int32_t K, int32_t rows_pad,
                    seg_stream_t stream);
/* the same transpose for a DEVICE-resident table of jobs in one launch.  Job d reads the K columns k0 .. k0+kcount of
 * w [rows][T][K] and writes wt[k - k0][t][dst_coff + r] (row length dst_ld; r < rows_pad, zero for r >= rows):
 * k0 = 0, kcount = K, dst_ld = rows_pad, dst_coff = 0 is seg_transpose_w; a column slice + column offset builds the
 * operand of a gathered data gradient (one tensor, several consuming convs side by side along K).  tile0 = number of
 * blocks of the jobs before d (a job has T * ceil(kcount/32) * ceil(rows_pad/32) blocks); total_tiles = their sum.  */
typedef struct seg_transpose_desc {
  const float* w;
  void* wt;
  int32_t rows, T, K;
  int32_t k0, kcount;
  int32_t rows_pad, dst_ld, dst_coff;
  int32_t tile0;
  int32_t src16;                 /* != 0: w points at an fp16 copy of the weight (same layout) instead of the fp32 master */
  int32_t dst_T, dst_t0;         /* dst_T > 0: the destination has dst_T taps and source tap t lands at tap t + dst_t0 (a 1x1
                                    kernel as the centre tap of a 3x3 operand: the shortcut conv of a ResPath / MultiResBlock
                                    joining the gathered data gradient of its 3x3 sibling); the other taps are never written   */
} seg_transpose_desc;
int seg_transpose_w_multi(const seg_transpose_desc* descs_dev, int32_t n, int32_t total_tiles, seg_stream_t stream);
/* fp32 -> fp16 cast of n elements                                                                              */
int seg_cast_f32_f16(const float* x, void* y, int64_t n, seg_stream_t stream);
/* flat fp32 buffer -> its split image (parity mode's GEMM weight operand): n a multiple of 8, 32-byte aligned  */
int seg_cast_f32_split(const float* x, void* y, int64_t n, seg_stream_t stream);
/* fp32 NHWC [npix][C] -> fp16 [npix][cs] zero padded (model input).  overflow (device int32, may be NULL): += the
 * number of input elements that are NaN, +-Inf or |x| >= 65504 - values the reference's fp32 tensors hold and an fp16 /
 * split-pair element cannot (they turn into Inf / NaN downstream; the engine raises on a non-zero count)          */
int seg_pack_input(const float* x, void* y, int64_t npix, int32_t C, int32_t cs, int32_t* overflow, seg_stream_t stream);
/* fp16 [npix][cs] columns [coff, coff+C) -> fp32 [npix][C]                                                     */
int seg_unpack_f32(const void* x, float* y, int64_t npix, int32_t C, int32_t cs, int32_t coff, seg_stream_t stream);

/* ---- parity mode (suffix _f32) ------------------------------------------------------------------------------------
 * Every entry point above that reads or writes activation tensors also exists with the suffix _f32: identical
 * arguments (channel counts / strides / offsets in ELEMENTS as before), 4 bytes per element, but an element is stored
 * as a SPLIT PAIR of fp16 numbers, x ~= hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048): ~22 significant
 * bits for 2^-14 <= |x| < 65504 (absolute error floor ~2^-36 below, Inf / NaN above - loud, not silent).  Eight
 * consecutive channels (an 8-aligned group; every cs / coff is a multiple of 8) occupy 32 bytes = [8 hi | 8 lo], so a
 * 16-byte chunk is an MFMA-ready fragment and the GEMM kernels fetch operands by LDS-DMA exactly as in fp16 mode and
 * contract a b ~= ah bh + (al bh + ah bl) / 2048 with three v_mfma_f32_32x32x16_f16 per product into two fp32
 * accumulators.  The GEMM weight operand `w` (and seg_transpose_w's output) have the same split layout along their
 * innermost axis (seg_cast_f32_split / the optimiser's SEG_OPERAND_SPLIT copy produce it from the fp32 master).
 * seg_pack_input_f32 / seg_unpack_f32_f32 convert from / to plain fp32 at the boundary; model outputs (heads) and
 * every statistic / gradient-of-parameter buffer are plain fp32 in both modes.  This is the mode the reference's fp32
 * arithmetic (2DCNN/Train.py:322-325: no mixed-precision policy) is matched in: every output within 1e-3 and equal
 * masks (tests/test_gpu_fp32.py, tests/test_gpu_fullsize.py); the un-suffixed functions are the fp16 throughput mode. */
int seg_igemm_fwd_f32(const seg_igemm_fwd_args* a, seg_stream_t stream);
int seg_igemm_wgrad_f32(const seg_igemm_wgrad_args* a, seg_stream_t stream);
int seg_bn_stats_f32(const void* x, int64_t npix, int32_t cs, int32_t coff, int32_t C, float* stats, int32_t nparts,
                     int32_t stats_ld, int32_t stats_coff, seg_stream_t stream);
int seg_bn_act_apply_f32(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff,
                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale,
                         const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw,
                         seg_stream_t stream);
int seg_bn_act_apply_head_f32(const void* x, int32_t x_cs, int32_t x_coff, void* y, int32_t y_cs, int32_t y_coff,
                              int32_t N, int32_t H, int32_t W, int32_t C, int32_t c_valid, const float* scale,
                              const float* shift, int32_t act, float alpha, void* pooled, int32_t ph, int32_t pw,
                              const seg_head_fwd_desc* head, seg_stream_t stream);
int seg_bn_act_bwd_reduce_inj_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                                  int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, int32_t act, float alpha, float* partial,
                                  int32_t nparts, const seg_head_inject* inj, seg_stream_t stream);
int seg_bn_act_bwd_apply_inj_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                                 int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C,
                                 const float* scale, const float* shift, const float* mean, const float* invstd,
                                 const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate,
                                 const seg_head_inject* inj, seg_stream_t stream);
int seg_bn_act_bwd_apply2_f32(const void* dy, const seg_bn_bwd_set* sets, int64_t npix, int32_t C, seg_stream_t stream);
int seg_bn_act_bwd_reduce_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                              int32_t x_coff, int64_t npix, int32_t C, const float* scale, const float* shift,
                              const float* mean, const float* invstd, int32_t act, float alpha, float* partial,
                              int32_t nparts, seg_stream_t stream);
int seg_bn_act_bwd_apply_f32(const void* dy, int32_t dy_cs, int32_t dy_coff, const void* x, int32_t x_cs,
                             int32_t x_coff, void* dx, int32_t dx_cs, int32_t dx_coff, int64_t npix, int32_t C,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* c1, const float* c2, int32_t act, float alpha, int32_t accumulate,
                             seg_stream_t stream);
int seg_act_bwd_f32(const void* dy, const void* y, void* dx, int64_t npix, int32_t cs, int32_t C, int32_t act,
                    float alpha, float* dbias, float gscale, const float* gscale_ptr, seg_stream_t stream);
int seg_maxpool_fwd_f32(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t ph, int32_t pw,
                        seg_stream_t stream);
int seg_maxpool_bwd_f32(const void* dy, const void* x, const void* y, void* dx, int32_t N, int32_t H, int32_t W,
                        int32_t cs, int32_t ph, int32_t pw, int32_t accumulate, seg_stream_t stream);
int seg_upsample_fwd_f32(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh, int32_t fw,
                         int32_t bilinear, seg_stream_t stream);
int seg_upsample_bwd_f32(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t cs, int32_t fh,
                         int32_t fw, int32_t bilinear, int32_t accumulate, seg_stream_t stream);
int seg_add_act_fwd_f32(const void* const* xs, int32_t n, void* y, int64_t nelem, int32_t cs, int32_t c_valid, int32_t act, float alpha,
                        seg_stream_t stream);
int seg_add_act_bwd_f32(const void* dy, const void* y, void* const* dxs, int32_t n, int64_t nelem, int32_t act,
                        float alpha, int32_t accumulate_mask, seg_stream_t stream);
int seg_add_act_stats_fwd_f32(const void* const* xs, const float* const* scales, const float* const* shifts, const int32_t* relus,
                              int32_t n, const seg_cat_desc* cat, const float* cat_scale, const float* cat_shift, void* y, int64_t npix,
                              int32_t cs, int32_t c_valid, int32_t act, float alpha, float* stats, int32_t nparts, int32_t stats_ld,
                              int32_t stats_coff, seg_stream_t stream);
int seg_bn_bwd_apply_fanout_f32(const void* dy, const void* x, void* const* dxs, int32_t n, int32_t accumulate_mask, int64_t npix,
                                int32_t cs, const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2,
                                int32_t relu_mask, const seg_cat_desc* cat, const float* cat_mean, const float* cat_invstd,
                                float* cat_partial, const seg_lazy_bn* lazy, int32_t nlazy, int32_t nparts, seg_stream_t stream);
int seg_cat_bn_bwd_scatter_f32(const void* g, const seg_cat_desc* cat, int64_t npix, int32_t cs, const float* scale, const float* mean,
                               const float* invstd, const float* c1, const float* c2, seg_stream_t stream);
int seg_mul_bcast_fwd_f32(const void* a, const void* r, void* y, int64_t npix, int32_t cs, int32_t r_cs,
                          seg_stream_t stream);
int seg_mul_bcast_bwd_f32(const void* dy, const void* a, const void* r, void* da, void* dr, int64_t npix, int32_t cs,
                          int32_t r_cs, int32_t acc_a, int32_t acc_r, seg_stream_t stream);
int seg_convlstm_gate_fwd_f32(const void* z, int32_t z_cs, void* h, int32_t h_cs, int64_t npix, int32_t F,
                              seg_stream_t stream);
int seg_convlstm_gate_bwd_f32(const void* dh, int32_t h_cs, const void* z, int32_t z_cs, void* dz, int64_t npix,
                              int32_t F, seg_stream_t stream);
int seg_head_fwd_f32(const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H, int32_t W, int32_t sy, int32_t sx,
                     const float* w, int32_t w_ld, const float* b, int32_t Cout, int32_t act, float* y,
                     seg_stream_t stream);
int seg_head_bwd_f32(const float* dy, const float* y, const void* x, int32_t cs, int32_t Cin, int32_t N, int32_t H,
                     int32_t W, int32_t sy, int32_t sx, const float* w, int32_t w_ld, int32_t Cout, int32_t act,
                     void* dx, int32_t accumulate, const float* gs_ptr, float* dw, float* db, float* work,
                     int64_t work_elems, seg_stream_t stream);
int seg_copy_channels_f32(const void* src, int32_t src_cs, int32_t src_coff, void* dst, int32_t dst_cs, int32_t dst_coff,
                          int32_t C, int64_t npix, int32_t accumulate, seg_stream_t stream);
int seg_transpose_w_f32(const float* w, void* wt, int32_t rows, int32_t T, int32_t K, int32_t rows_pad,
                        seg_stream_t stream);
int seg_transpose_w_multi_f32(const seg_transpose_desc* descs_dev, int32_t n, int32_t total_tiles, seg_stream_t stream);
int seg_pack_input_f32(const float* x, void* y, int64_t npix, int32_t C, int32_t cs, int32_t* overflow, seg_stream_t stream);
int seg_unpack_f32_f32(const void* x, float* y, int64_t npix, int32_t C, int32_t cs, int32_t coff, seg_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* SEG_HIP_H */

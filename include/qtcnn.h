/* qtcnn.h -- C ABI of the MI355X (gfx950) QuadtreeCNN hot-path library.
 *
 * The reference (Avirup221/Multimodal-Hierarchical-CNN-for-Sun-Salutation-Pose-
 * Classification) has no FFI of its own: its hot path is torch.nn modules calling
 * ATen.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md).  Every entry point names the reference call it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns every buffer (including workspaces); the library allocates
 *     nothing persistent on the device.  Process-wide state is limited to (i) the
 *     kernel-selection switches qt_set_patch_conv / qt_set_pt_conv / qt_set_stem_conv /
 *     qt_set_wgrad_patch_min_width / qt_set_wgrad_patch_variant and the QTCNN_* environment variables they mirror
 *     (read once; DESIGN.md section 5 lists them) -- they pick between kernels that
 *     compute the same result, set them before the first launch -- and (ii) the
 *     per-device "LDS limit raised" bits of the large-LDS kernels;
 *   - the per-op entry points enqueue all work on `stream` (a hipStream_t passed as
 *     void*) with no internal synchronisation and are re-entrant across streams.  A
 *     qt_plan additionally OWNS one side stream and a handful of events (created in
 *     qt_plan_create, destroyed in qt_plan_destroy): qt_plan_backward forks the
 *     weight-gradient launches onto it and joins them back into `stream` before it
 *     returns (qt_plan_side_fence exposes the join to another stream).  One plan is
 *     used by one host thread at a time; one process per GPU under data parallelism;
 *   - return value: QT_OK (0) or a negative qt_status; qt_last_error() returns a
 *     thread-local description of the last failure; nothing throws across the ABI;
 *   - activations are NHWC, element type qt_dtype (bf16 throughput build or the
 *     f32 parity build of the SAME kernels); accumulation is always f32 on MFMA.
 */
#ifndef QTCNN_H_
#define QTCNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qt_status {
  QT_OK = 0,
  QT_ERR_INVALID_ARG = -1,
  QT_ERR_LAUNCH = -2,
  QT_ERR_UNSUPPORTED = -3
} qt_status;

typedef enum qt_dtype { QT_F32 = 0, QT_BF16 = 1 } qt_dtype;

int qt_version(void);
const char* qt_last_error(void);

/* ------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA.
 *   QT_CONV_FWD   replaces nn.Conv2d forward   (ResNet-18 convs reached at
 *                 Quadtree_from scratch/models.py:222-230,241; the quadrant conv
 *                 :235 called 4x at :284-287; resnet/models.py:80-100,158-161)
 *   QT_CONV_DGRAD replaces the conv backward-input ATen call that loss.backward()
 *                 (Quadtree_from scratch/Quadtree_train.py:65) makes per conv.
 * dst[m][n] = epilogue( sum_{tap,k} W[n][tap][k] * src[pixel(m) moved by tap][k] )
 * epilogue(v) = relu_mask( relu( v*scale[n] + shift[n] + residual[m][n] ) ),
 * each part optional; stats_partial (optional) receives per-128-row-tile sums and
 * sums of squares of the raw v per channel: [qt_conv2d_stats_rows][2][n_out] f32.
 * ------------------------------------------------------------------------ */
enum { QT_CONV_FWD = 0, QT_CONV_DGRAD = 1 };

typedef struct qt_conv_desc {
  int dtype;         /* qt_dtype of src / weight / dst / residual / relu_mask */
  int mode;          /* QT_CONV_FWD or QT_CONV_DGRAD */
  int batch;         /* images */
  int in_h, in_w;    /* spatial size of src (per quadrant when quad != 0) */
  int out_h, out_w;  /* spatial size of dst (per quadrant for quad FWD) */
  int k_per_tap;     /* contracted channels per tap; multiple of 64 (bf16) / 32 (f32) */
  int n_out;         /* channels of dst; multiple of 8 */
  int kh, kw, stride, pad;
  long long src_img_stride; /* elements between images of src */
  int src_row_stride;       /* elements between rows of src */
  int src_pix_stride;       /* elements between pixels of src */
  int quad;  /* S = 2 (1 means 2) or 4: FWD reads the S x S regions of a (S*in_h x S*in_w) map as
                S*S*batch images (region rr*S+rc of image n is image n*S*S + rr*S + rc) with zero halo
                at the seams (Quadtree_from scratch/models.py:277-287 quadrants; :62-78 quadrants and
                sub-quadrants of AttentionHierarchicalCNN); DGRAD scatters the per-region gradient
                images back onto the un-split map */
  int relu;
  /* optional strided destination (0 = dense [M][n_out]): row (img, oh, ow) is written to pixel
   * (oh*dst_sub + dst_off_h, ow*dst_sub + dst_off_w) of a dst_h x dst_w image; residual and
   * relu_mask are read at the same place.  Used to run the data gradient of a stride-2 conv as
   * four stride-1 gathers, one per output-pixel parity class (qt_pack_dgrad_s2). */
  int dst_sub, dst_h, dst_w, dst_off_h, dst_off_w;
  /* dst_merge = C > 0 (with dst_sub = 2, n_out = 4*C): the n_out channels of a row are the FOUR parity classes of a
   * stride-2 data gradient at once -- channel n is channel n % C of pixel (2*oh + (n/C >> 1), 2*ow + (n/C & 1)) of the
   * dst_h x dst_w image with C channels; residual / relu_mask / BatchNorm links are read there and the link sums
   * are written as 4 partial rows of C channels per pixel tile.  One 2x2-tap launch over the gradient map instead of
   * four gathers that each re-read it (weight operand: qt_pack_dgrad_s2_merged). */
  int dst_merge;
  int dst_merge_res0;  /* with dst_merge: the residual is added to class (0,0) only (the other pixels of it are never read) */
  /* Conv3d as ONE launch (nn.Conv3d 3x3x3 / pad 1 of /root/reference/3dcnn/models.py:108-139, cnn+lstm/models.py:104-121):
   * kt = 3 frame taps on time-major clips [frames][batch / frames][H][W][C] -- `batch` counts images = frames x clips,
   * image t * (batch / frames) + b is frame t of clip b; tap (kt, kh, kw) of an output image reads the image one frame
   * earlier / same / later (zero where that frame does not exist), weight [n_out][kt][kh][kw][k_per_tap]; f32
   * accumulation over all kt * kh * kw taps, the epilogue (bias, residual, BatchNorm3d statistics) sees the finished value.
   * DGRAD mirrors it.  Needs stride 1 and no region / strided-destination mode; 0 or 1 = plain 2-D. */
  int kt, frames;
  /* With dst_merge: 1 = a FIFTH tap slot (weight rows then hold 5 slots of k_per_tap) that reads qt_conv_io.extra_src -- a
   * second gradient map of the same geometry as src -- at the window's first pixel (r, c) and only reaches class (0,0): the
   * data gradient of the block's 1x1 / stride-2 downsample (torchvision BasicBlock.downsample) accumulated in
   * the same launch as conv1's, instead of a launch of its own whose output the merged launch then re-reads as a residual. */
  int dst_merge_extra;
} qt_conv_desc;

typedef struct qt_conv_io {
  const void* src;
  const void* weight;    /* [n_out][kh*kw][k_per_tap], K contiguous */
  void* dst;             /* [M][n_out] */
  const float* scale;    /* [n_out] or NULL */
  const float* shift;    /* [n_out] or NULL */
  const void* residual;  /* [M][n_out] or NULL */
  const void* relu_mask; /* [M][n_out] or NULL: dst = mask > 0 ? dst : 0 */
  float* stats_partial;  /* or NULL */
  /* Optional (data-gradient launches): up to two BatchNorms consume the value g written to
   * dst; the epilogue then also emits, per 128-row tile, sum g and sum g*(y-mean)*invstd for
   * each of them ([qt_conv2d_stats_rows][2][n_out] f32), which is what qt_bn_bwd_finalize
   * takes -- the separate qt_bn_bwd_reduce pass over g and y is not needed. */
  struct {
    const void* y;       /* BatchNorm input saved by the forward pass, [M][n_out] */
    const float* mean;   /* [n_out] */
    const float* invstd; /* [n_out] */
    float* partial;
  } bwd_bn[2];
  /* The ReLU mask as ONE BIT per element instead of a tensor of the data type: [M][n_out / 8] bytes, bit (n & 7) of byte
   * n / 8 of row m set = the gradient passes (what qt_bn_act_mask writes next to the activation).  NULL or exclusive with
   * relu_mask; read at the destination row like relu_mask (strided / merged destinations included).  n_out % 8 == 0.
   * 1/16 of the bytes of a bf16 mask: the mask is the second-largest epilogue operand of a data-gradient launch. */
  const unsigned char* relu_mask_bits;
  const void* extra_src;  /* qt_conv_desc.dst_merge_extra: [batch][in_h][in_w][k_per_tap] with src's strides, or NULL */
} qt_conv_io;

/* The 56x56 64->64 bf16 3x3 stride-1 convs (ResNet layer1) on the persistent sliding-ring kernel with the input window
 * resident in LDS (csrc/conv_patch.hip) instead of the generic implicit GEMM: 0 never, anything else (default) on.
 * (Rounds 1-3 had an experimental one-tile-per-workgroup kernel behind mode 1; removed in round 4.) */
void qt_set_patch_conv(int mode);
/* 3x3 / stride 1 / pad 1 convolutions on dense 28x28, 14x14 and 7x7 maps with >= 16 images and a multiple of 128 output
 * channels (the twelve such convs of layer2..4, forward and data gradient) take the patch-resident ping-pong kernel
 * (csrc/conv_pt.hip: 196-pixel tiles of whole image rows, input patch + halo of a 128-byte channel chunk in LDS for all
 * nine taps, weight tiles streamed through an LDS-DMA ring, two wave groups half a period apart).  1 (default) on,
 * 0 generic implicit GEMM.  Same arithmetic (f32 accumulation on MFMA) in a different K order: chunk-major, tap-minor;
 * qt_conv2d_stats_rows follows the choice (two rows per 196-pixel tile: one per wave row). */
void qt_set_pt_conv(int mode);
/* The 128-channel-tile instantiations of that kernel are PERSISTENT: one workgroup per CU walks consecutive (channel tile,
 * pixel tile) items without stopping the K-tile stream (the 28x28 stage of the benchmark: four items per workgroup).
 * tests: cap the grid at n workgroups (0 = one per CU; a huge n = one item per workgroup, QTCNN_PT_PERSIST=0 does the
 * same): results are bit-identical for every n. */
void qt_set_pt_conv_max_workgroups(int n);
/* The packed bf16 stem convolution (desc of qt_pack_stem_input: kh 7|8, kw 1, stride 2, k_per_tap 32,
 * 64 outputs, no residual / mask / bwd_bn) takes a dedicated kernel (csrc/conv_stem.hip: input
 * rows of a 4 x 112 pixel tile in LDS, filter in registers, one partial-statistics row per
 * workgroup).  1 (default) on, 0 generic implicit GEMM. */
void qt_set_stem_conv(int mode);
int qt_conv2d_stats_rows(const qt_conv_desc* desc);
/* ---------------------------------------------------------------------------
 * The stride-2 transition of a ResNet stage in ONE launch (csrc/conv_s2.hip): conv1 of layerN.0 (3x3 / stride 2 / pad 1,
 * no bias) and its downsample (1x1 / stride 2, no bias) computed from one staged input patch -- torchvision BasicBlock
 * with `downsample`, i.e. the nn.Conv2d calls behind /root/reference/Quadtree_from scratch/models.py:
 * 228-229,241 (layer2.0, layer3.0, layer4.0) and resnet/models.py:86-88,99.  Replaces two qt_conv2d_igemm launches.
 *   y_conv = epi_conv(conv3x3s2(x, w_conv)),  y_down = epi_down(conv1x1s2(x, w_down)),
 *   epi(v) = relu?(v * scale[c] + shift[c]) with every part optional; stats_* (optional) receive per (pixel tile, wave row)
 *   sums and sums of squares of the RAW v per channel: [qt_conv_s2_pair_stats_rows][2][c_out] f32 (qt_bn_finalize's input).
 * x: NHWC [batch][in_h][in_w][c_in]; w_conv: [c_out][3][3][c_in] (qt_pack_conv_weight's forward operand); w_down:
 * [c_out][c_in]; y_*: NHWC [batch][in_h/2][in_w/2][c_out].  Supported (qt_conv_s2_pair_supported): in_h == in_w in
 * {56, 28, 14}, c_in a multiple of 64 (bf16) / 32 (f32), c_out a multiple of 128, batch >= 16 (a multiple of 4 for
 * 14x14 inputs); other problems take qt_conv2d_igemm.  Same arithmetic as the generic path (f32 accumulation on MFMA)
 * in another K order.  QTCNN_S2_CONV=0 / qt_set_conv_s2(0): report "unsupported" (same-box A/B against the generic pair).
 * ------------------------------------------------------------------------ */
typedef struct qt_conv_s2_desc {
  int dtype;            /* qt_dtype of x, the weights and both outputs */
  int batch;
  int in_h, in_w;
  int c_in, c_out;
  int relu_conv, relu_down;
} qt_conv_s2_desc;
typedef struct qt_conv_s2_io {
  const void* src;
  const void* w_conv;
  const void* w_down;
  void* y_conv;
  void* y_down;
  const float* scale_conv;  /* [c_out] or NULL */
  const float* shift_conv;
  const float* scale_down;
  const float* shift_down;
  float* stats_conv;        /* or NULL */
  float* stats_down;
} qt_conv_s2_io;
int qt_conv_s2_pair_supported(const qt_conv_s2_desc* desc);
int qt_conv_s2_pair_stats_rows(const qt_conv_s2_desc* desc);
int qt_conv_s2_pair(const qt_conv_s2_desc* desc, const qt_conv_s2_io* io, void* stream);
void qt_set_conv_s2(int mode);
/* tests: cap the persistent grid (0 = one workgroup per CU) so that small batches exercise several items per workgroup */
void qt_set_conv_s2_max_workgroups(int n);

int qt_conv2d_igemm(const qt_conv_desc* desc, const qt_conv_io* io, void* stream);

/* Weight gradient of the convolution described by `desc` (mode QT_CONV_FWD):
 *   dw[n][tap][k] += sum_pixels dy[pixel][n] * x[pixel moved by tap][k]     (f32)
 * replaces the conv/linear backward-weight ATen calls under loss.backward()
 * (Quadtree_from scratch/Quadtree_train.py:65).  `dw` must be zeroed (or hold the
 * running sum) by the caller: partial tiles are added with f32 atomics. */
int qt_conv2d_wgrad(const qt_conv_desc* desc, const void* dy, const void* x, float* dw, void* stream);
/* nn.Linear backward-weight (classifier.0 of the reference, Quadtree_from scratch/models.py:264-271): dw [out][in] f32 =
 * dy^T x over `rows` rows of dy [rows][out] and x [rows][in] (both of `dtype`, dense), WRITTEN -- where every tile of dw
 * has a single range of rows (always at rows <= 256) with plain stores, no zero fill and no atomics. */
int qt_linear_wgrad(int dtype, const void* dy, const void* x, float* dw, int rows, int out, int in, void* stream);
/* Same, with a caller-owned scratch buffer of qt_conv2d_wgrad_workspace_bytes(desc) bytes (0: this
 * shape does not use one).  With it the streaming kernel writes one partial filter per range of
 * positions with plain stores and a second kernel adds them to `dw` in a fixed order: no atomics,
 * bit-reproducible.  A NULL or too small workspace falls back to the atomic accumulation. */
size_t qt_conv2d_wgrad_workspace_bytes(const qt_conv_desc* desc);
int qt_conv2d_wgrad_ws(const qt_conv_desc* desc, const void* dy, const void* x, float* dw, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The shapes with a workspace (bf16: 3x3 stride 1; and the stride-2 pair of a ResNet transition block, 3x3 / 2 pad 1 and
 * 1x1 / 2 on an even-sized map, csrc/conv_wgrad_s2.hip): the gradient is WRITTEN (not accumulated) in the
 * reference's OIHW layout by the kernel that sums the partial filters -- no [O][kh][kw][I] scratch,
 * no zero fill, no qt_unpack_conv_wgrad.  QT_ERR_UNSUPPORTED for every other shape. */
int qt_conv2d_wgrad_oihw(const qt_conv_desc* desc, const void* dy, const void* x, float* grad_oihw, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same, with the sum of the partial filters enqueued on `sum_stream` (behind the kernel through an event) instead of the
 * kernel's stream, which is then free for its next launch at once.  `workspace` must stay untouched until that sum has run:
 * the caller orders its next use of it behind sum_stream (csrc/plan.hip alternates two workspaces).  sum_stream NULL or ==
 * stream: qt_conv2d_wgrad_oihw. */
int qt_conv2d_wgrad_oihw_on(const qt_conv_desc* desc, const void* dy, const void* x, float* grad_oihw, void* workspace,
                            size_t workspace_bytes, void* stream, void* sum_stream);
/* bf16 3x3 / stride 1 / pad 1 weight gradients of images at least `min_width` wide take the
 * streaming kernel (csrc/conv_wgrad_patch.hip: one workgroup accumulates all nine taps of a
 * 64x64 channel tile while dY and X stream through LDS once).  0 = never, <0 = default (7: every stage of the
 * model).  Env QTCNN_WGRAD_PATCH_MIN_W. */
void qt_set_wgrad_patch_min_width(int min_width);
/* Which streaming kernel those shapes take: 3 = tile-resident (a tile of 128-256 padded positions + halo double
 * buffered in LDS, fragment reads two taps ahead of the MFMAs, source offsets from a table in LDS; default),
 * 0 / 2 = the round-1 ring kernel with one / two wave groups.  <0 = default; env QTCNN_WP_VARIANT. */
void qt_set_wgrad_patch_variant(int variant);
/* bf16 weight gradients of the stride-2 convolutions (3x3 / 2 pad 1, 1x1 / 2 pad 0, even map, channels % 64 == 0) on the
 * four parity planes of the input, tile-resident, all taps per workgroup, deterministic (csrc/conv_wgrad_s2.hip):
 * 1 = on (default), 0 = the generic kernel with float atomics, <0 = default.  Env QTCNN_WGRAD_S2. */
void qt_set_wgrad_s2(int on);

/* ------------------------------------------------------------------------
 * Layout packing (HBM-bound).
 * ------------------------------------------------------------------------ */
/* Packed stem input: [B][QT_STEM_PAD_H][QT_STEM_PAD_W][4] (3 zero rows/cols before,
 * 3 rows / 5 cols after, channel 3 zero) so that the 7x7/2 stem conv
 * (torchvision conv1, reached at Quadtree_from scratch/models.py:223) becomes 7
 * row taps of 32 contiguous elements: desc {kh=7,kw=1,stride=2,pad=0,k_per_tap=32,
 * src_pix_stride=4}.  Input contract: image [B,3,224,224] f32 NCHW
 * (Quadtree_from scratch/dataloader.py:72-91). */
#define QT_STEM_PAD_H 230
#define QT_STEM_PAD_W 232
int qt_pack_stem_input(int dtype, const float* image_nchw, void* dst, int batch, void* stream);
/* OIHW f32 master weights (reference state_dict layout) -> [O][kh][kw][I] (w_fwd)
 * and/or [I][kh][kw][O] (w_dgrad); either may be NULL.  Linear layers: kh=kw=1. */
int qt_pack_conv_weight(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int O, int I, int kh, int kw,
                        void* stream);
/* Every conv / linear weight of a model in one launch (LDS tile transposes): per item the same
 * result as qt_pack_conv_weight, or, with stride2_dgrad != 0 and k == 3, w_dgrad in the parity-class
 * layout of qt_pack_dgrad_s2 (stride2_dgrad = 1) or the merged layout of qt_pack_dgrad_s2_merged (= 2; only the
 * nine real taps are written: the zero slots of w_dgrad must have been zeroed once).  O and I must be multiples of
 * 32 (k = 3) or 64 (k = 1); at most 32 items. */
typedef struct qt_pack_item {
  const float* w_oihw;
  void* w_fwd;   /* nullable */
  void* w_dgrad; /* nullable */
  int O, I, k, stride2_dgrad;   /* stride2_dgrad: 0 stride 1; 1 parity classes (qt_pack_dgrad_s2); 2 merged, 4 slots
                                   (qt_pack_dgrad_s2_merged); 3 merged, FIVE slots per row (slot 4 belongs to the downsample);
                                   4 (k = 1): w_dgrad is that five-slot operand of the block's conv1 -- this item fills slot 4
                                   of the class-(0,0) rows: element ((i*5 + 4)*O + o) = w[o][i] (rows of the other classes
                                   keep the zeros qt_plan_init_workspace wrote) */
} qt_pack_item;
int qt_pack_weights_batched(int dtype, const qt_pack_item* items, int n, void* stream);
/* Adam with L2-in-gradient weight decay, torch.optim.Adam semantics without amsgrad / maximize
 * (the reference's optimizer, Quadtree_from scratch/Quadtree_train.py:45: lr 1e-4, weight_decay 1e-4):
 *   g = grad*grad_scale + wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)                 all f32
 * qt_adam_multi: any list of tensors in one launch per 48 tensors.
 * qt_adam_pack_weights_batched: the same update INSIDE the one-launch weight packing, so the f32
 * masters, both moments and the packed operand copies are each touched once per step
 * (opt[j].param must be items[j].w_oihw).  `step` counts from 1; grad_scale 0 means 1. */
typedef struct qt_adam_desc {
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
  int step;
} qt_adam_desc;
typedef struct qt_adam_item {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long numel;
} qt_adam_item;
int qt_adam_multi(const qt_adam_item* items, int n, const qt_adam_desc* adam, void* stream);
int qt_adam_pack_weights_batched(int dtype, const qt_pack_item* items, const qt_adam_item* opt, const qt_adam_desc* adam,
                                 int n, void* stream);
/* Data-gradient operand of a stride-2 conv (k = 3 pad 1, or k = 1 pad 0) split by the parity
 * (ph, pw) of the input pixel: class c = ph*2+pw gets [I][taps_c][O] with only the taps that
 * reach it (k=3: 1,2,2,4 taps; k=1: 1,0,0,0), stored back to back in class order.  Row taps of
 * class parity 0: {1}; parity 1: {2, 0} (source row i, i+1).  Element offsets of the four
 * classes are returned in class_offset[4], tap grid in class_kh[4] / class_kw[4] (host arrays). */
int qt_pack_dgrad_s2(int dtype, const float* w_oihw, void* dst, int O, int I, int k, long long* class_offset,
                     int* class_kh, int* class_kw, void* stream);
/* The same four classes as ONE operand [4*I][2*2][O] for a single 2x2-tap launch over the gradient map
 * (qt_conv_desc.dst_merge = I, kh = kw = 2, pad 0, n_out = 4*I): row class*I + i, tap slot th*2 + tw with th / tw = 0
 * the tap on the source row / column itself, 1 the tap on the next one; slots no tap maps to are zero.  k = 3 only;
 * dst holds 16*O*I elements (9/16 of them non-zero: the launch reads the gradient map once instead of four times and
 * needs no per-class launches, at 16/9 of the MFMA work). */
int qt_pack_dgrad_s2_merged(int dtype, const float* w_oihw, void* dst, int O, int I, void* stream);
/* [64][3][7][7] -> [64][taps][8][4] for the packed stem; taps = 7, or 8 (8th row zero)
 * when a 32-element tap is only half a K-step (bf16: desc.kh = 8) */
int qt_pack_stem_weight(int dtype, const float* w_oihw, void* dst, int taps, void* stream);
/* [O][kh][kw][I] f32 gradient -> OIHW f32 (.grad), optionally accumulating */
int qt_unpack_conv_wgrad(const float* dw, float* grad_oihw, int O, int I, int kh, int kw, int accumulate, void* stream);
int qt_unpack_stem_wgrad(const float* dw, float* grad_oihw, int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d (torch defaults eps 1e-5, momentum 0.1; replaces nn.BatchNorm2d of
 * torchvision's ResNet-18 in train() and eval() mode) and fused activations.
 * ------------------------------------------------------------------------ */
/* partial[rows][2][C] (from qt_conv2d_igemm) -> batch mean / invstd, scale = gamma*invstd,
 * shift = beta - mean*scale; updates running stats (unbiased var) if given. */
/* partial buffers must have room for qt_stats_capacity_rows(rows) rows (long row
 * counts are folded 256:1 into the spare rows before the final reduction). */
int qt_stats_capacity_rows(int rows);
int qt_bn_finalize(float* partial, int rows, int C, long long count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics */
int qt_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* scale, float* shift, void* stream);
/* the same for up to 32 BatchNorms in one launch (an eval forward needs all of them up front) */
typedef struct qt_bn_eval_item {
  const float *gamma, *beta, *running_mean, *running_var;
  float *scale, *shift;
  int C;
  float *mean, *invstd; /* optional pair: running_mean and 1/sqrt(running_var + eps), what the backward kernels read */
} qt_bn_eval_item;
int qt_bn_eval_affine_batched(const qt_bn_eval_item* items, int n, float eps, void* stream);
/* out = relu?( y*scale+shift + (residual ? residual*res_scale+res_shift : 0) ), [M][C] */
int qt_bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* residual,
              const float* res_scale, const float* res_shift, int relu, void* out, long long M, int C, void* stream);
/* the same, and mask_bits (nullable) receives out > 0 as one bit per element: [M][C/8] bytes, bit (c & 7) of byte c / 8 --
 * the ReLU mask in the form qt_conv_io.relu_mask_bits takes (the training forward of the plan writes it for every
 * activation whose mask a data-gradient epilogue applies) */
int qt_bn_act_mask(int dtype, const void* y, const float* scale, const float* shift, const void* residual,
                   const float* res_scale, const float* res_shift, int relu, void* out, unsigned char* mask_bits,
                   long long M, int C, void* stream);
/* backward: g = d(loss)/d(BN output) (masked by `mask` > 0 if given).  qt_bn_bwd_finalize with count == 0 is the backward of
 * an eval-mode BatchNorm (running statistics in mean / invstd): dx = g*gamma*invstd without the batch-mean terms. */
int qt_bn_bwd_partial_rows(long long M, int C);
int qt_bn_bwd_reduce(int dtype, const void* g, const void* mask, const void* y, const float* mean, const float* invstd,
                     float* partial, long long M, int C, void* stream);
int qt_bn_bwd_finalize(float* partial, int rows, int C, long long count, const float* gamma, const float* invstd,
                       float* dgamma, float* dbeta, int accumulate, float* coef /*[3][C]*/, void* stream);
int qt_bn_bwd_apply(int dtype, const void* g, const void* mask, const void* y, const float* mean, const float* invstd,
                    const float* coef, void* dy, void* g_out, long long M, int C, void* stream);

/* stem: relu(y*scale+shift) then MaxPool2d(3,2,1): [B][112][112][64] -> [B][56][56][64]
 * (torchvision bn1/relu/maxpool, Quadtree_from scratch/models.py:224-226); argmax (u8, optional)
 * records the winning tap for the backward, y_at_max (optional, same shape and type as pooled)
 * the raw conv1 output at that tap (input of qt_stem_bn_bwd_sums). */
int qt_stem_pool(int dtype, const void* y, const float* scale, const float* shift, void* pooled,
                 unsigned char* argmax, void* y_at_max, int batch, void* stream);
/* Eval forward: conv1 (packed stem descriptor: xpad, qt_pack_stem_weight's [64][taps][32]) + folded BatchNorm
 * (scale / shift) + ReLU + MaxPool2d(3,2,1) in one launch; the conv1 map never reaches memory.  bf16 only
 * (QT_ERR_UNSUPPORTED otherwise: qt_conv2d_igemm + qt_stem_pool). */
int qt_stem_conv_pool(int dtype, const void* xpad, const void* weight, int taps, const float* scale, const float* shift,
                      void* pooled, int batch, void* stream);
/* The same straight from the f32 NCHW image [B][3][224][224] the reference's dataloader hands over (16-byte aligned): the
 * kernel packs its input rows in LDS (same rounding as qt_pack_stem_input), so neither xpad nor the conv1 map exists in
 * memory; bit-identical to qt_pack_stem_input + qt_stem_conv_pool.  QT_ERR_UNSUPPORTED (take the packed form) for f32, an
 * image that is not 16-byte aligned, or QTCNN_STEM_NCHW=0. */
int qt_stem_conv_pool_nchw(int dtype, const float* image_nchw, const void* weight, int taps, const float* scale,
                           const float* shift, void* pooled, int batch, void* stream);
int qt_stem_pool_bwd(int dtype, const void* dpooled, const unsigned char* argmax, const void* y, const float* scale,
                     const float* shift, void* g, int batch, void* stream);
/* The same gradient fused with bn1's backward, without materialising it: _reduce emits
 * qt_stem_bn_bwd_rows(batch) rows of partial sums for qt_bn_bwd_finalize, _apply writes
 * d(loss)/d(conv1 output) directly. */
int qt_stem_bn_bwd_rows(int batch);
int qt_stem_bn_bwd_reduce(int dtype, const void* dpooled, const unsigned char* argmax, const void* y,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          float* partial, int batch, void* stream);
/* The same partial sums from the pooled side (each pooled cell feeds one conv1 position): reads
 * dpooled and y_at_max only.  qt_stem_bn_bwd_sums_rows(batch) rows. */
int qt_stem_bn_bwd_sums_rows(int batch);
int qt_stem_bn_bwd_sums(int dtype, const void* dpooled, const void* y_at_max, const float* scale, const float* shift,
                        const float* mean, const float* invstd, float* partial, int batch, void* stream);
int qt_stem_bn_bwd_apply(int dtype, const void* dpooled, const unsigned char* argmax, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* coef, void* dy,
                         int batch, void* stream);
/* Stem backward in one launch (bf16): the gradient of conv1's output (max-pool backward + ReLU mask + BatchNorm backward,
 * the arithmetic of qt_stem_bn_bwd_apply) is computed tile by tile in LDS and contracted with the packed input at once:
 * dw[64][7][32] (the layout qt_unpack_stem_wgrad reads, zeroed by the caller) += conv1's weight gradient.  The 411 MB map
 * d(loss)/d(conv1 output) is neither written nor read.  QT_ERR_UNSUPPORTED for f32 / QTCNN_STEM_BWD_FUSED=0: use
 * qt_stem_bn_bwd_apply + qt_conv2d_wgrad. */
int qt_stem_bn_bwd_wgrad(int dtype, const void* dpooled, const unsigned char* argmax, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* coef, const void* xpad,
                         float* dw, int batch, void* stream);
/* AdaptiveAvgPool2d(1,1)+flatten into columns [col0, col0+C) of the fused feature
 * matrix (Quadtree_from scratch/models.py:242,289-294) and its backward fused with the
 * ReLU mask of the pooled map. */
int qt_avgpool(int dtype, const void* x, void* dst, int batch, int hw, int C, int ld, int col0, void* stream);
int qt_avgpool_bwd(int dtype, const void* d, const void* x, void* g, int batch, int hw, int C, int ld, int col0,
                   void* stream);
/* quadrant head tail: MaxPool2d(2,2) (7->3) + flatten(1) + torch.cat placement
 * (Quadtree_from scratch/models.py:237,284-294): q [B*4][7][7][128] -> dst[b][col0 + quad*1152 + c*9 + ph*3 + pw] */
int qt_quad_pool(int dtype, const void* q, void* dst, int batch, int ld, int col0, void* stream);
int qt_quad_pool_bwd(int dtype, const void* d, const void* q, void* dq, int batch, int ld, int col0, void* stream);
/* Heads of AttentionHierarchicalCNN (Quadtree_from scratch/models.py:6-101).
 * qt_region_avgpool: AdaptiveAvgPool2d((1,1)) + flatten of the per-region conv+ReLU maps x [batch*split^2][hw][C]
 *   (:21-30) into dst[b*ld + col0 + slot*C + c]; `slot` is the reference's append order -- quadrants TL,TR,BL,BR
 *   (:62-67), and for split 4 quadrant*4 + sub-quadrant (:70-78).  dst_dtype: `dtype` or QT_F32.
 * qt_region_avgpool_bwd: g = x > 0 ? d[...]/hw : 0  (mean-pool backward fused with the ReLU mask).
 * qt_attention_gate (:34-38,:81-89): v [B][16][64] f32; act [B][16][32] = relu(W1 v + b1); alpha [B][16] =
 *   softmax_j(w2 . act_j + b2); out[b*ld + col0 + c] = sum_j alpha_j v_j[c]   (out has type `dtype`).
 * qt_attention_gate_bwd: from d[b*ld + col0 + c] -> dv [B][16][64], and the two row-wise factors of the
 *   parameter gradients: ds [B][16] (d/d score) and dpre [B][16][32] (d/d pre-ReLU hidden), so that
 *   dW1 = dpre^T v, db1 = sum dpre, dw2 = ds^T act, db2 = sum ds (qt_gemm_small / qt_col_sum).
 * qt_relu_mask_cols: out[r][c] = act[r*ld+col0+c] > 0 ? d[r*ld+col0+c]*mul : 0  (f32, dense): backward of the
 *   Linear -> ReLU -> Dropout numerical branch (:43-46) whose output lives inside the fused feature matrix. */
int qt_region_avgpool(int dtype, const void* x, void* dst, int dst_dtype, int batch, int split, int hw, int C, int ld,
                      int col0, void* stream);
int qt_region_avgpool_bwd(int dtype, const void* d, int d_dtype, const void* x, void* g, int batch, int split, int hw,
                          int C, int ld, int col0, void* stream);
int qt_attention_gate(int dtype, const float* v, const float* w1, const float* b1, const float* w2, const float* b2,
                      float* act, float* alpha, void* out, int batch, int ld, int col0, void* stream);
int qt_attention_gate_bwd(int dtype, const void* d, const float* v, const float* act, const float* alpha, const float* w1,
                          const float* w2, float* ds, float* dpre, float* dv, int batch, int ld, int col0, void* stream);
int qt_relu_mask_cols(int dtype, const void* d, const void* act, float* out, long long rows, int cols, int ld, int col0,
                      float mul, void* stream);
/* nn.LSTM recurrence (cnn+lstm/models.py:43-49,81-86; batch_first, one direction, f32, torch gate order i,f,g,o).
 * qt_lstm_forward: xproj [B][T][4H] = x_t W_ih^T + b_ih for every step (a thin product done by the caller; b_hh
 *   [4H], nullable, is added here),
 *   whh_t = W_hh^T [H][4H] (qt_transpose_f32) -> gates [B][T][4H] (post-activation), cell / hprev / hout [B][T][H]
 *   (hprev = h_{t-1}, zeros at t = 0).  One launch, one workgroup per sequence; H in {256, 64}.
 * qt_lstm_backward: dhout [B][T][H] (gradient w.r.t. every h_t, nullable) + dlast [B][H] (w.r.t. h_{T-1}, nullable)
 *   -> dgates [B][T][4H] w.r.t. the pre-activation gates; then dx = dgates W_ih, dW_ih = dgates^T x,
 *   dW_hh = dgates^T hprev, db_ih = db_hh = column sums (qt_gemm_small / qt_col_sum).  whh = W_hh [4H][H].
 * qt_scale_by_nonzero: g = x != 0 ? g*mul : 0 -- backward of the dropout between LSTM layers from the dropped
 *   activations themselves. */
int qt_lstm_forward(const float* xproj, const float* whh_t, const float* bhh, float* gates, float* cell, float* hprev,
                    float* hout, int batch, int T, int H, void* stream);
int qt_lstm_backward(const float* dhout, const float* dlast, const float* gates, const float* cell, const float* whh,
                     float* dgates, int batch, int T, int H, void* stream);
int qt_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream);
int qt_scale_by_nonzero(float* g, const float* x, long long n, float mul, void* stream);
/* dst[i] = (float)src[i] (src of type `dtype`): f32 copy of the fused per-frame features for the f32 LSTM products */
int qt_cast_f32(int dtype, const void* src, float* dst, long long n, void* stream);
/* nn.Dropout (Quadtree_from scratch/models.py:258,269), in place, counter-hash RNG */
int qt_dropout(int dtype, void* x, long long rows, int cols, int ld, unsigned long long seed, float p, void* stream);
/* g = act > 0 ? g*mul : 0 (ReLU / dropout backward from the forward output) */
int qt_relu_mask_scale(int dtype, void* g, const void* act, long long n, float mul, void* stream);
/* out[c] (+)= sum_r x[r*ld+c]  (bias gradients) */
int qt_col_sum(int dtype, const void* x, long long rows, int cols, int ld, float* out, int accumulate, void* stream);

/* nn.Linear on a small batch with a large weight matrix (classifier.0: Linear(5376 -> 2688) + ReLU,
 * Quadtree_from scratch/models.py:266-268, and its backward-input product), bf16:
 *   y[m][n] = relu?( bias[n] + sum_k x[m][k] * w[n][k] ),  x [M][K], w [N][K], y [M][N]
 * split over K so that every CU streams a disjoint block of the weights once; partial products go
 * through `workspace` (qt_linear_workspace_bytes, f32 [S][M][N]) and are added in a fixed order.
 * Covers M <= 256, N % 64 == 0, K % 64 == 0; anything else returns QT_ERR_UNSUPPORTED (use
 * qt_conv2d_igemm with a 1x1 descriptor). */
size_t qt_linear_workspace_bytes(int M, int N, int K);
int qt_linear_bf16(const void* x, const void* w, const float* bias, int relu, void* y, int M, int N, int K,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Thin dense products (nn.Linear 47->94->256 and 2688->12 and their gradients,
 * Quadtree_from scratch/models.py:255-260,270): C[m][n] = relu?(acc? + bias[n] +
 * sum_k A[m*ars+k*aks] * B[n*brs+k*bks]).
 * ------------------------------------------------------------------------ */
typedef struct qt_gemm_small_desc {
  int M, N, K;
  int a_dtype, b_dtype, c_dtype;
  long long a_row_stride, a_k_stride;
  long long b_row_stride, b_k_stride;
  long long c_row_stride;
  int relu, accumulate;
} qt_gemm_small_desc;
int qt_gemm_small(const qt_gemm_small_desc* desc, const void* A, const void* B, const float* bias, void* C,
                  void* stream);

/* ------------------------------------------------------------------------
 * 3-D clip models (Quadtree3DCNN 3dcnn/models.py:96-214, Ji3DCNN cnn+lstm/models.py:93-142).  Clip activations are
 * time-major NHWC [T][B][H][W][C]: a Conv3d(3x3x3, padding 1) is three qt_conv2d_igemm launches over contiguous frame
 * ranges (out[t] += conv2d(in[t+kt-1], W[:,:,kt]), accumulated through `residual` = dst), its gradients three
 * QT_CONV_DGRAD / qt_conv2d_wgrad launches (host side: <pkg>/video3d.py).  The HBM-bound pieces:
 * ------------------------------------------------------------------------ */
/* clips [B][T][3][H][W] f32 (the reference's image_sequence_input) -> [T][B][H][W][128] of `dtype`: the 27 taps x 3
 * channels of pixel (t,b,h,w), element ((kt*3+kh)*3+kw)*3+c, zeros outside the clip and in 81..127: the first Conv3d
 * (3 input channels) becomes a 1x1 convolution with k_per_tap = 128 */
int qt_pack_clip27(int dtype, const float* clips, void* dst, int batch, int frames, int h, int w, void* stream);
/* One launch per Conv3d block (nn.Conv3d 3x3x3 + BatchNorm3d, /root/reference/3dcnn/models.py:108-139): the operands of
 * the 27-tap implicit GEMM (qt_conv_desc.kt = 3) from nn.Conv3d's weight w [O][I][3][3][3] f32 -- w_fwd
 * [O_pad][27][I_pad], w_dgrad [I_pad][27][O_pad] (optional), zero padded; first != 0: w_fwd [O_pad][128] in the K order of
 * qt_pack_clip27 (27 * I <= 128) -- and the padded per-channel vectors: vec_in_dev = DEVICE array of 5 device pointers
 * (conv bias, BatchNorm weight, bias, running_mean, running_var; [O] each, NULL = padding value) -> vec_out [5][O_pad]
 * padded with 0, 1, 0, 0, 1.  qt_unpack_conv3d_wgrad is the way back for the weight gradient: dw = three [O_pad][9][I_pad]
 * f32 blocks, one per frame tap, as qt_conv2d_wgrad writes them (first != 0: [O_pad][128]) -> grad [O][I][3][3][3]. */
int qt_pack_conv3d_block(int dtype, const float* w, void* w_fwd, void* w_dgrad, int O, int I, int O_pad, int I_pad, int first,
                         const float* const* vec_in_dev, float* vec_out, void* stream);
int qt_unpack_conv3d_wgrad(const float* dw, float* grad, int O, int I, int O_pad, int I_pad, int first, void* stream);
/* BatchNorm3d batch statistics of a finished map y [M][C]: partial[qt_bn_stats_rows(M,C)][2][C] sums / sums of squares
 * (fixed summation order), input of qt_bn_finalize (buffer capacity qt_stats_capacity_rows(rows)) */
int qt_bn_stats_rows(long long M, int C);
int qt_bn_stats(int dtype, const void* y, long long M, int C, float* partial, void* stream);
/* nn.MaxPool3d(kernel = stride = (pool_t, 2, 2)), pool_t 1 or 2, floor mode: x [T][B][H][W][C] ->
 * out [T/pool_t][B][H/2][W/2][C]; argmax (u8 per output element, optional) = index of the first maximum in (t,h,w)
 * scan order, which is where torch's backward sends the gradient.  _bwd writes every element of dx. */
int qt_pool3d_max(int dtype, const void* x, void* out, unsigned char* argmax, int frames, int batch, int h, int w, int C,
                  int pool_t, void* stream);
int qt_pool3d_max_bwd(int dtype, const void* dout, const unsigned char* argmax, void* dx, int frames, int batch, int h,
                      int w, int C, int pool_t, void* stream);
/* Round 3: BatchNorm3d (scale / shift) + ReLU + MaxPool3d (pool_t, 2, 2) in one pass over the RAW conv output y -- the
 * activation map relu(bn(y)) is not materialised; y_at_max (nullable) keeps y at each pooled cell's argmax, so that the
 * BatchNorm-backward sums can be taken from the pooled side: qt_bn_bwd_reduce(g = d(loss)/d(pooled), mask = pooled,
 * y = y_at_max, M = pooled cells) followed by qt_bn_bwd_finalize with count = ALL positions of the map.  Values compared are
 * the activation rounded to `dtype`, i.e. exactly what qt_bn_act + qt_pool3d_max compare.  y rows are y_channels wide
 * (<= C, a multiple of 8: qt_conv3d_first_fwd writes 32-channel rows while the next layer wants 64-channel K rows);
 * out / argmax / y_at_max rows are C wide and zero in channels >= y_channels. */
int qt_pool3d_bn_relu_max(int dtype, const void* y, const float* scale, const float* shift, void* out, unsigned char* argmax,
                          void* y_at_max, int frames, int batch, int h, int w, int C, int y_channels, int pool_t, void* stream);
/* ... and the backward of the three in one pass: dy = a (g - b - xhat c) with g = dout at the window's argmax where
 * pooled > 0, zero elsewhere (coef = qt_bn_bwd_finalize's [3][C]); replaces qt_pool3d_max_bwd + qt_bn_bwd_apply and the
 * full-size gradient map between them.  dy rows are dy_channels wide (>= y_channels; zero beyond y_channels). */
/* tests: the smallest launch (in 16-byte channel groups) for which qt_pool3d_bn_bwd_apply takes its resident-grid form (bf16, no
 * padding channels: C = y_channels = dy_channels); 0 restores the default (2^20) */
void qt_set_pool3d_apply_light_min(long long groups);
int qt_pool3d_bn_bwd_apply(int dtype, const void* dout, const unsigned char* argmax, const void* pooled, const void* y,
                           const float* mean, const float* invstd, const float* coef, void* dy, int frames, int batch, int h,
                           int w, int C, int y_channels, int dy_channels, int pool_t, void* stream);
/* Round 3: the first Conv3d of the clip models (3 -> 32 channels, 3x3x3, pad 1; /root/reference/3dcnn/models.py:108)
 * straight from the f32 clip [B][T][3][H][W]: y [T][B][H][W][32] (NO channel padding), bias-free accumulator.  w_packed is
 * qt_pack_conv3d_block(first = 1)'s [>= 32][128] filter.  Either scale / shift (+ relu) of 32 channels (eval: folded
 * BatchNorm3d), or stats: qt_conv3d_first_stats_rows(...) rows of [2][64] partial sums of y for qt_bn_finalize (channels
 * 32..63 zero), or neither.  Replaces qt_pack_clip27 + a 1x1 qt_conv2d_igemm over 256-byte K rows (3.3 GB written and read
 * back at 32 x 8 x 224 x 224).  QT_ERR_UNSUPPORTED (take the packed form) for f32, H % 4, W % 16, W > 256, a clip that is
 * not 16-byte aligned, QTCNN_CONV3D_FIRST=0. */
int qt_conv3d_first_stats_rows(int batch, int frames, int h, int w);
/* Round 3: the clip models' second Conv3d (32 -> 64 channels, 3x3x3, pad 1; /root/reference/3dcnn/models.py:115) with the
 * input frame slabs resident in LDS: x [T][B][H][W][x_channels] (channels 0..31 read: the 64-channel-padded pooled map of
 * block 1, or 32-channel rows), w_packed = qt_pack_conv3d_block's [64][27][64], y [T][B][H][W][64].  Bias-free accumulator
 * with either scale / shift (+ relu) of 64 channels, or stats = qt_conv3d_c32_stats_rows(...) rows of [2][64] partial sums
 * for qt_bn_finalize, or neither.  Same result as qt_conv2d_igemm with kt = 3 on the same operands (other summation order).
 * QT_ERR_UNSUPPORTED (take qt_conv2d_igemm) for f32, odd H, W % 16, W > 128, misaligned operands, QTCNN_CONV3D_SLAB=0. */
int qt_conv3d_c32_stats_rows(int batch, int frames, int h, int w);
int qt_conv3d_c32_fwd(int dtype, const void* x, int x_channels, const void* w_packed, void* y, const float* scale,
                      const float* shift, int relu, float* stats, int batch, int frames, int h, int w, void* stream);
/* ... and its data gradient: dx [T][B][H][W][dx_channels] (channels 0..31 = d(loss)/d(input); dx_channels = 64: channels
 * 32..63 written as zeros, the rows a 64-channel-padded conv3d_block1 reads; 32: no padding, round 4) from dy
 * [T][B][H][W][64] and qt_pack_conv3d_block's data-gradient filter [64][27][64] ([input channel][tap][output channel]): the
 * same slab walk over dy with the flipped filter, as two launches over dy's channel halves joined through an f32 scratch
 * (qt_conv3d_c32_dgrad_scratch_bytes; 0 = shape not covered, take qt_conv2d_igemm in QT_CONV_DGRAD mode). */
size_t qt_conv3d_c32_dgrad_scratch_bytes(int batch, int frames, int h, int w);
int qt_conv3d_c32_dgrad(int dtype, const void* dy, const void* w_dgrad_packed, void* dx, int dx_channels, void* scratch,
                        size_t scratch_bytes, int batch, int frames, int h, int w, void* stream);
/* ... and its weight gradient: dweight [64][32][3][3][3] f32 in nn.Conv3d's layout (every element written) from x (channels
 * 0..31 of x_channels-wide rows) and dy [T][B][H][W][64]; the contraction runs over positions with both operands read
 * through transposing LDS reads; per-workgroup partial filters in `workspace` (qt_conv3d_c32_wgrad_workspace_bytes; 0 = shape
 * not covered, take qt_conv2d_wgrad per frame tap) added in a fixed order: deterministic. */
size_t qt_conv3d_c32_wgrad_workspace_bytes(int batch, int frames, int h, int w);
int qt_conv3d_c32_wgrad(int dtype, const void* x, int x_channels, const void* dy, float* dweight, void* workspace,
                        size_t workspace_bytes, int batch, int frames, int h, int w, void* stream);
/* Eval forward of the whole conv3d_block1 in one launch: Conv3d + folded BatchNorm3d (scale / shift of 32 channels, the conv
 * bias folded into shift) + ReLU + MaxPool3d((1,2,2)) in registers; pooled [T][B][H/2][W/2][pooled_channels]: 64 with channels
 * 32..63 zero (the K rows the implicit GEMM reads) or 32 (round 4: what the slab kernels of block 2 read); the conv map never
 * reaches memory.  Same values as qt_conv3d_first_fwd(scale, shift, relu) followed by qt_pool3d_max.  Shapes as
 * qt_conv3d_first_fwd. */
int qt_conv3d_first_fwd_pool(int dtype, const float* clips, const void* w_packed, void* pooled, int pooled_channels,
                             const float* scale, const float* shift, int batch, int frames, int h, int w, void* stream);
/* ... and its weight gradient from the f32 clip and dy [T][B][H][W][32] (32-channel rows, what qt_pool3d_bn_bwd_apply
 * writes with dy_channels = 32): dweight [32][3][3][3][3] f32 in nn.Conv3d's own layout, every element written; partial
 * filters per workgroup in `workspace` (qt_conv3d_first_wgrad_workspace_bytes, 0 = shape not covered: W % 32, else as
 * qt_conv3d_first_fwd) added in a fixed order: deterministic.  Replaces qt_pack_clip27 + qt_conv2d_wgrad +
 * qt_unpack_conv3d_wgrad. */
size_t qt_conv3d_first_wgrad_workspace_bytes(int batch, int frames, int h, int w);
int qt_conv3d_first_wgrad(int dtype, const float* clips, const void* dy, float* dweight, void* workspace,
                          size_t workspace_bytes, int batch, int frames, int h, int w, void* stream);
/* Round 4: the same weight gradient with d(loss)/dy formed on the way in -- the backward of MaxPool3d((1,2,2)) + ReLU +
 * BatchNorm3d of conv3d_block1 (/root/reference/3dcnn/models.py:108-112) inside the weight-gradient kernel: y = the raw conv
 * output [T][B][H][W][32], dout / argmax [T][B][H/2][W/2][pooled_channels] (pooled_channels 32 or 64) = the gradient of the
 * pooled map and qt_pool3d_bn_relu_max's argmax (pool_t = 1), mean / invstd / scale / shift = qt_bn_finalize's vectors,
 * coef = qt_bn_bwd_finalize's [3][pooled_channels].  Replaces qt_pool3d_bn_bwd_apply (dy_channels = 32) +
 * qt_conv3d_first_wgrad: dy (0.8 GB at 32 clips x 8 frames of 224 x 224) is neither written nor read.  The ReLU mask is
 * recomputed from y (relu(y scale + shift) > 0) instead of read from the pooled map.  Workspace and shapes as
 * qt_conv3d_first_wgrad; QT_ERR_UNSUPPORTED otherwise. */
int qt_conv3d_first_wgrad_fused(int dtype, const float* clips, const void* y, const void* dout, const unsigned char* argmax,
                                int pooled_channels, const float* mean, const float* invstd, const float* scale,
                                const float* shift, const float* coef, float* dweight, void* workspace, size_t workspace_bytes,
                                int batch, int frames, int h, int w, void* stream);
int qt_conv3d_first_fwd(int dtype, const float* clips, const void* w_packed, void* y, const float* scale, const float* shift,
                        int relu, float* stats, int batch, int frames, int h, int w, void* stream);
/* nn.AdaptiveAvgPool3d((1,1,1)) + flatten(1) into columns [col0, col0+C) of an f32 [B][ld] matrix, and its backward */
int qt_avgpool_tb(int dtype, const void* x, float* dst, int frames, int batch, int hw, int C, int ld, int col0, void* stream);
int qt_avgpool_tb_bwd(int dtype, const float* d, void* g, int frames, int batch, int hw, int C, int ld, int col0,
                      void* stream);

/* ------------------------------------------------------------------------
 * Whole-network executor.  One plan = one model variant at a maximum batch:
 *   QT_MODEL_QUADTREE         QuadtreeCNN  (Quadtree_from scratch/models.py:214-305;
 *                             resnet/models.py:70-180 with `mode`)
 *   QT_MODEL_STANDARD_RESNET  StandardResNetCNN (resnet/models.py:7-65)
 *   QT_MODEL_ATTENTION        AttentionHierarchicalCNN (Quadtree_from scratch/models.py:6-101); `mode` is
 *                             ignored; tensors are listed under base_cnn.* names for the ResNet part (the
 *                             reference keeps it as a local: bind features_extractor.{0,1,4,5} /
 *                             global_processor.{0,1} to base_cnn.{conv1,bn1,layer1,layer2} / {layer3,layer4})
 *   QT_MODEL_CNN_LSTM         CnnLstm (cnn+lstm/models.py:14-89): `batch` counts FRAMES (sequences x seq_len); the
 *                             frozen per-frame ResNet-18 (bind cnn_backbone.{0,1,4,5,6,7} to base_cnn.{conv1,bn1,
 *                             layer1..4}) + Linear-ReLU-Linear on the pose vector feed a 2-layer LSTM over seq_len
 *                             steps; logits [batch / seq_len][num_classes].  Gradients: MLP, LSTM, classifier.
 * The tensor table lists every parameter / buffer under the reference's
 * state_dict key (first-seen `base_cnn.*` names); the caller
 * passes one device pointer per entry (f32, reference layouts: OIHW conv
 * weights, [out][in] linear weights, int64 num_batches_tracked).
 *   qt_plan_pack_weights : casts/permutes the master weights into the packed
 *                          operands (call after every optimizer step)
 *   qt_plan_forward      : model(image[B,3,224,224] f32 NCHW, numerical[B,47] f32)
 *                          -> logits[B,num_classes] f32   (forward of models.py:273-305);
 *                          training == 1: BatchNorm batch statistics + running-stat update, dropout with `seed`;
 *                          training == 0: eval mode, BatchNorm / ReLU / residual fused into the conv epilogues, nothing
 *                          kept for a backbone backward; training == 2: eval-mode arithmetic (running statistics, no
 *                          dropout) but every tensor qt_plan_backward needs is kept, so that model.eval() followed by
 *                          logits.backward() with trainable backbone parameters works (Grad-CAM on the all-trainable
 *                          variant, Quadtree_from scratch/grad_cam.py:72-83)
 *   qt_plan_backward     : loss.backward() of Quadtree_train.py:65 given dlogits;
 *                          grads[i] (f32, same layout as tensors[i]) is written
 *                          (not accumulated) when non-NULL; QT_BWD_HEAD = classifier,
 *                          numerical MLP and quadrant head, QT_BWD_LAYER4 = layer4,
 *                          QT_BWD_LAYER32 = layers 3 and 2, QT_BWD_LAYER1 = layer1 and the stem
 *                          (lets the caller start the gradient all-reduce of a bucket while the
 *                          next phase runs; the last bucket is 0.6 MB).
 * ------------------------------------------------------------------------ */
enum { QT_MODEL_QUADTREE = 0, QT_MODEL_STANDARD_RESNET = 1, QT_MODEL_ATTENTION = 2, QT_MODEL_CNN_LSTM = 3 };
enum { QT_MODE_FUSION = 0, QT_MODE_IMAGE_ONLY = 1, QT_MODE_NUMERICAL_ONLY = 2 };
/* call order HEAD, LAYER4, LAYER32, LAYER1 (or any union of consecutive phases in one call);
 * QT_BWD_REST = LAYER32 | LAYER1, QT_BWD_BACKBONE = LAYER4 | REST. */
enum { QT_BWD_HEAD = 1, QT_BWD_LAYER4 = 2, QT_BWD_LAYER32 = 4, QT_BWD_LAYER1 = 8, QT_BWD_REST = 12, QT_BWD_BACKBONE = 14,
       QT_BWD_ALL = 15 };

typedef struct qt_plan_desc {
  int dtype;
  int batch;          /* maximum images per call */
  int num_classes;
  int model;
  int mode;
  int numerical_dim;  /* 47 */
  float dropout_p;    /* 0.5 */
  float bn_eps;       /* 1e-5 */
  float bn_momentum;  /* 0.1 */
  int seq_len;        /* QT_MODEL_CNN_LSTM: frames per sequence (reference default 4) */
  int lstm_hidden;    /* QT_MODEL_CNN_LSTM: 256 */
} qt_plan_desc;

typedef struct qt_plan qt_plan;

int qt_plan_create(const qt_plan_desc* desc, qt_plan** out);
void qt_plan_destroy(qt_plan* plan);
int qt_plan_num_tensors(const qt_plan* plan);
const char* qt_plan_tensor_name(const qt_plan* plan, int i);
int qt_plan_tensor_kind(const qt_plan* plan, int i);            /* 0 parameter, 1 f32 buffer, 2 int64 counter */
int qt_plan_tensor_shape(const qt_plan* plan, int i, int* dims4); /* returns ndim */
size_t qt_plan_workspace_bytes(const qt_plan* plan);
/* byte offset inside the workspace of a named activation / gradient buffer:
 * "stem.pooled", "block<0-7>.out|.a1|.gout", "conv<i>.y|.gy", "fused", "dfused", "hidden";
 * QT_MODEL_ATTENTION also "attention.vectors" (f32 [B][16][64]) and "attention.weights" (f32 [B][16]) */
int qt_plan_find_buffer(const qt_plan* plan, const char* name, size_t* offset);
/* Measurement aid (bench.py roofline): while enabled, every MFMA kernel launch is
 * bracketed by HIP events on its stream; _end sums algorithmic FLOPs, milliseconds
 * and launch counts per kind {0 igemm forward, 1 igemm dgrad, 2 wgrad}. */
int qt_plan_profile_begin(qt_plan* plan);
int qt_plan_profile_end(qt_plan* plan, double* flops3, double* ms3, int* launches3);
/* algorithmic HBM bytes of the launches of the last profile, per kind: every operand of a launch once (source map,
 * weights, destination, per-pixel epilogue operands) -- the figure bench.py puts beside the PMC-measured traffic */
int qt_plan_profile_bytes(const qt_plan* plan, double* bytes3);
/* Weight gradients run on a plan-owned side stream; a partial qt_plan_backward phase returns
 * without joining it.  Before consuming that phase's gradients on another stream (the
 * all-reduce stream), make it wait for the side stream with qt_plan_side_fence (and for the
 * caller's stream as usual).  The last phase joins the side stream into the caller's stream. */
int qt_plan_side_fence(qt_plan* plan, void* waiting_stream);
/* Once per workspace, before the first qt_plan_pack_weights: the constant vectors of the identity BatchNorm affine and the
 * zero tap slots of the merged stride-2 data-gradient operands (qt_pack_dgrad_s2_merged: the packers write the nine real
 * taps only).  Synchronises `stream`. */
int qt_plan_init_workspace(qt_plan* plan, void* workspace, void* stream);
int qt_plan_pack_weights(qt_plan* plan, void* workspace, void* const* tensors, int for_backward, void* stream);
/* Optimizer step fused with the re-packing (replaces optimizer.step() + qt_plan_pack_weights of a
 * training step; Quadtree_from scratch/Quadtree_train.py:66): per plan tensor one gradient pointer
 * (NULL = frozen / unused: only re-packed) and its two Adam moments.  The conv / linear weights are
 * updated inside the one-launch packing kernel, the rest by qt_adam_multi. */
int qt_plan_adam_step(qt_plan* plan, void* workspace, void* const* tensors, float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const qt_adam_desc* adam, int for_backward, void* stream);
int qt_plan_forward(qt_plan* plan, void* workspace, void* const* tensors, const float* image, const float* numerical,
                    float* logits, int batch, int training, unsigned long long seed, void* stream);
int qt_plan_backward(qt_plan* plan, void* workspace, void* const* tensors, float* const* grads, const float* numerical,
                     const float* dlogits, int phases, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QTCNN_H_ */

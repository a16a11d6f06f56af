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
 *     nothing persistent on the device and keeps no mutable global state;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), no
 *     internal synchronisation, re-entrant across streams / one process per GPU;
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
  int quad;  /* 1: FWD reads the four 2x2 quadrants of a (2*in_h x 2*in_w) map as 4*batch
                images with zero halo at the seam (models.py:277-287); DGRAD scatters
                the per-quadrant gradient images back onto the un-split map */
  int relu;
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
} qt_conv_io;

int qt_conv2d_stats_rows(const qt_conv_desc* desc);
int qt_conv2d_igemm(const qt_conv_desc* desc, const qt_conv_io* io, void* stream);

/* Weight gradient of the convolution described by `desc` (mode QT_CONV_FWD):
 *   dw[n][tap][k] += sum_pixels dy[pixel][n] * x[pixel moved by tap][k]     (f32)
 * replaces the conv/linear backward-weight ATen calls under loss.backward()
 * (Quadtree_from scratch/Quadtree_train.py:65).  `dw` must be zeroed (or hold the
 * running sum) by the caller: partial tiles are added with f32 atomics. */
int qt_conv2d_wgrad(const qt_conv_desc* desc, const void* dy, const void* x, float* dw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QTCNN_H_ */

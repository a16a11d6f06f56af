// Arguments shared by the implicit-GEMM convolution kernels (conv_igemm.hip: generic tiles;
// conv_pt.hip: the patch-resident ping-pong kernel for the 3x3 layers of layer2..4).
#pragma once
#include "qt_common.h"

namespace qtc {

struct ConvArgs {
  const void* src;
  const void* wgt;
  void* dst;
  const float* scale;      // per destination channel, nullable
  const float* shift;      // per destination channel, nullable
  const void* residual;    // [M][N] same dtype, nullable
  const void* relu_mask;   // [M][N] same dtype: result *= (mask > 0), nullable
  const unsigned char* relu_mask_bits;   // [M][N/8]: the same mask, one bit per element (qt_conv_io.relu_mask_bits), nullable
  float* stats_partial;    // [gridM][2][N] per-tile sum / sum of squares, nullable
  // BatchNorm-backward partial sums of the value written to dst (g): sum g and sum g*xhat with
  // xhat = (bn_y - mean) * invstd, for up to two BatchNorms that consume g
  const void* bn_y[2];
  const float* bn_mean[2];
  const float* bn_invstd[2];
  float* bn_partial[2];     // [gridM][2][N] each
  long long src_img_stride;
  int src_row_stride, src_pix_stride;
  int M, N;
  int OH, OW, IH, IW;
  int KC;                  // K elements per tap
  int ntaps, KW;
  int stride, pad;
  int quad;
  int relu;
  int gridM, gridN;
  // destination row mapping (0 = dense): row m = (img, oh, ow) of the OHxOW grid is written to
  // pixel (oh*dst_sub + dst_oh, ow*dst_sub + dst_ow) of a dst_h x dst_w image
  int dst_sub, dst_h, dst_w, dst_oh, dst_ow;
  int dst_merge_res0;       // residual only for class (0,0)
  int dst_merge;            // C > 0: N = 4*C, the four parity classes of a stride-2 data gradient in one launch (qtcnn.h)
  long long extra_off;      // dst_merge with FIVE tap slots (ntaps == 5): elements from src to the second gradient map that
                            // slot 4 reads at the window's first pixel (the downsample's gradient; class (0,0) only)
  FastDiv div_ohw, div_ow;
#ifdef QT_KERNEL_PROF
  unsigned long long* prof;   // experiment build only (qt_set_igemm_prof): [workgroup][4] s_memrealtime stamps (10 ns)
#endif
  // Conv3d as ONE implicit GEMM (round 3): KT > 1 frame taps on time-major clips [T][Bf][H][W][C] -- tap (kt, kh, kw) of
  // output image n = t * Bf + b reads image n + (kt - KT/2) * Bf (data gradient: n - (kt - KT/2) * Bf), valid while that
  // frame exists; ntaps = KT * KH * KW <= 32.  KT <= 1: plain 2-D.
  int KT, frames, imgs_per_frame;
  long long frame_stride;   // elements between consecutive frames of the source
};

}  // namespace qtc

// conv_pt.hip (3x3 stride-1 convolutions of the 28x28 / 14x14 / 7x7 stages: input patch resident in LDS, ping-pong MFMA
// schedule): does it take this problem, how many statistics rows (= pixel tiles) it writes, and its launch
bool qt_pt_eligible(const qtc::ConvArgs& a, int dtype, bool dgrad);
int qt_pt_stats_rows(const qtc::ConvArgs& a, bool dgrad);
int qt_pt_launch(const qtc::ConvArgs& a, int dtype, bool dgrad, hipStream_t stream);

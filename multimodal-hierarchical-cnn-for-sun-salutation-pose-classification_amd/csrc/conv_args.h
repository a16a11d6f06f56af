// Arguments shared by the implicit-GEMM convolution kernels (conv_igemm.hip: generic tiles;
// conv_pp.hip: the 8-wave ping-pong kernel for the long 3x3 layers).
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
  FastDiv div_ohw, div_ow;
};

}  // namespace qtc

// conv_pp.hip: does the ping-pong kernel take this problem, with which pixel-tile height, and its launch
bool qt_pp_eligible(const qtc::ConvArgs& a, int dtype, bool dgrad);
int qt_pp_tile_m(const qtc::ConvArgs& a, int dtype);
int qt_pp_launch(const qtc::ConvArgs& a, int dtype, bool dgrad, hipStream_t stream);

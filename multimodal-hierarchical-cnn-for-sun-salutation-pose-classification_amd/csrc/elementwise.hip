// HBM-bound kernels around the MFMA convolutions: BatchNorm statistics /
// apply / backward, ReLU + residual, stem max-pool, global average pool,
// quadrant max-pool + concat, dropout, column sums.  All activations NHWC with
// 8 channels (16 B bf16 / 32 B f32) per thread, f32 arithmetic.
//
// Reference ops replaced: nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d /
// nn.AdaptiveAvgPool2d inside torchvision's ResNet-18 as wired at
// /root/reference/Quadtree_from scratch/models.py:222-243; the quadrant head's
// ReLU + MaxPool2d(2,2) + flatten + torch.cat at :236-237,284-294; nn.Dropout at
// :258,269; and their autograd backward.
#include <type_traits>

#include "qt_common.h"

namespace {

int grid_for(long long total, int block = 256, int cap = 16384) {
  long long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) { QtVec8<float>::load(p, v); }

// ---------------------------------------------------------------------------------
// BatchNorm (training) statistics: partial[rows][2][C] -> mean / invstd / scale / shift
// + running statistics update (torch: momentum 0.1, unbiased running variance).
// ---------------------------------------------------------------------------------
// Sum rows rl, rl+16, ... of a [rows][2][C] partial table for channel c.  Eight rows per trip with
// all sixteen loads issued before the first add: the kernels below are a chain of dependent
// launches on the critical path of every BatchNorm, and a one-row-per-trip loop costs one
// memory latency per 16 rows (9-13 us per launch at 256-1024 rows instead of ~4).
constexpr int kFinC = 16;  // channels per block of the finalize kernels (x 16 row groups)

__device__ __forceinline__ void sum_partial_rows(const float* __restrict__ partial, int rows, int C, int c, int rl,
                                                 double& s1, double& s2) {
  int r = rl;
  for (; r + 16 * 7 < rows; r += 16 * 8) {
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = partial[((long long)(r + 16 * k) * 2 + 0) * C + c];
      b[k] = partial[((long long)(r + 16 * k) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s1 += (double)a[k];
      s2 += (double)b[k];
    }
  }
  for (; r < rows; r += 16) {
    s1 += (double)partial[((long long)r * 2 + 0) * C + c];
    s2 += (double)partial[((long long)r * 2 + 1) * C + c];
  }
}

__global__ __launch_bounds__(16 * kFinC) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* num_batches_tracked,
                                   float momentum, float eps, float* mean, float* invstd, float* scale, float* shift) {
  // kFinC channels x 16 row groups = 256 threads, 4 KB of LDS: one wave per SIMD and <= 64 VGPRs, so that the launch
  // finds room on a CU whose LDS and registers a weight-gradient workgroup of the side stream is holding (as a
  // 1024-thread / 16 KB block it waited 40-50 us per BatchNorm for one of those to retire)
  __shared__ double red[2][16][kFinC];
  const int cl = threadIdx.x % kFinC, rl = threadIdx.x / kFinC;
  const int c = blockIdx.x * kFinC + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) sum_partial_rows(partial, rows, C, c, rl, s1, s2);
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    s1 = s2 = 0.0;
    for (int k = 0; k < 16; ++k) {
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = (float)m;
    invstd[c] = (float)is;
    scale[c] = (float)((double)g * is);
    shift[c] = (float)((double)b - m * (double)g * is);
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  }
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
}

// stage 1 of a long partial-row reduction: block (cb, s) sums rows [64 s, 64 s + 64)
// of partial[rows][2][C] into row (rows + s).  256 threads = 64 channels x 4 row lanes.
constexpr int kFold = 64;
__global__ void stats_stage1_kernel(float* partial, int rows, int C) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * kFold;
  int r1 = r0 + kFold;
  if (r1 > rows) r1 = rows;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      s1 += partial[((long long)r * 2 + 0) * C + c];
      s2 += partial[((long long)r * 2 + 1) * C + c];
    }
  }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    const long long o = (long long)(rows + blockIdx.y) * 2;
    partial[(o + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    partial[(o + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rvar[c] + eps);
  const float s = gamma[c] * is;
  scale[c] = s;
  shift[c] = beta[c] - rmean[c] * s;
}

// every BatchNorm of a model in one launch (eval mode: 20 dependent 4 us launches otherwise)
constexpr int BN_MAX_ITEMS = 32;
struct BnEvalBatchArgs {
  const float* gamma[BN_MAX_ITEMS];
  const float* beta[BN_MAX_ITEMS];
  const float* rmean[BN_MAX_ITEMS];
  const float* rvar[BN_MAX_ITEMS];
  float* scale[BN_MAX_ITEMS];
  float* shift[BN_MAX_ITEMS];
  float* mean[BN_MAX_ITEMS];    // nullable: running statistics in the form the backward kernels take
  float* invstd[BN_MAX_ITEMS];
  int C[BN_MAX_ITEMS];
  float eps;
};
__global__ void bn_eval_affine_batched_kernel(BnEvalBatchArgs a) {
  const int j = blockIdx.x;
  for (int c = threadIdx.x; c < a.C[j]; c += blockDim.x) {
    const float is = 1.f / sqrtf(a.rvar[j][c] + a.eps);
    const float s = a.gamma[j][c] * is;
    a.scale[j][c] = s;
    a.shift[j][c] = a.beta[j][c] - a.rmean[j][c] * s;
    if (a.mean[j]) {
      a.mean[j][c] = a.rmean[j][c];
      a.invstd[j][c] = is;
    }
  }
}

// out = relu?( y*scale + shift + (res ? res*res_scale + res_shift : 0) )
template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                              const T* __restrict__ res, const float* __restrict__ rscale,
                              const float* __restrict__ rshift, int relu, T* __restrict__ out,
                              unsigned char* __restrict__ bits, long long M, int C) {
  const int cgs = C >> 3;
  const long long total = M * cgs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cgs) * 8;
    const long long off = i * 8;
    float v[8], sc[8], sh[8];
    QtVec8<T>::load(y + off, v);
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) {
      float r[8];
      QtVec8<T>::load(res + off, r);
      if (rscale) {
        float rs[8], rb[8];
        load8f(rscale + c0, rs);
        load8f(rshift + c0, rb);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e] * rs[e] + rb[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    QtVec8<T>::store(out + off, v);
    if (bits) {  // the ReLU mask of the data-gradient epilogues, one bit per element (qt_conv_io.relu_mask_bits)
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
      bits[i] = (unsigned char)b;
    }
  }
}

// ---------------------------------------------------------------------------------
// BatchNorm backward.  g = gradient w.r.t. the BN output (ReLU mask already
// applied, or applied here from `mask` > 0).  pass 1: per-block partial sums of
// g and g*xhat;  pass 2 (after bn_bwd_finalize): dy = a*(g - b - xhat*c).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ mask,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ partial, long long M, int C,
                                                            int rows_per_block) {
  extern __shared__ float red[];  // [RL][C][2]
  const int cgs = C >> 3;
  const int RL = 256 / cgs;  // row lanes
  const int cg = threadIdx.x % cgs, rl = threadIdx.x / cgs;
  const int c0 = cg * 8;
  float mu[8], is[8];
  load8f(mean + c0, mu);
  load8f(invstd + c0, is);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const long long r_begin = (long long)blockIdx.x * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  if (rl < RL)
    for (long long r = r_begin + rl; r < r_end; r += RL) {
      const long long off = r * C + c0;
      float gv[8], yv[8];
      QtVec8<T>::load(g + off, gv);
      QtVec8<T>::load(y + off, yv);
      if (mask) {
        float mv[8];
        QtVec8<T>::load(mask + off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = mv[e] > 0.f ? gv[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += gv[e];
        s2[e] += gv[e] * (yv[e] - mu[e]) * is[e];
      }
    }
  if (rl < RL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rl * C + c0 + e) * 2 + 0] = s1[e];
      red[(rl * C + c0 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < RL; ++r) {
      a += red[(r * C + c) * 2 + 0];
      b += red[(r * C + c) * 2 + 1];
    }
    partial[((long long)blockIdx.x * 2 + 0) * C + c] = a;
    partial[((long long)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// coef[0][c] = gamma*invstd, coef[1][c] = sum_g/M, coef[2][c] = sum_gxhat/M
__global__ __launch_bounds__(16 * kFinC) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* dgamma, float* dbeta, int accumulate, float* coef) {
  __shared__ double red[2][16][kFinC];  // small on purpose: see bn_finalize_kernel
  const int cl = threadIdx.x % kFinC, rl = threadIdx.x / kFinC;
  const int c = blockIdx.x * kFinC + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) sum_partial_rows(partial, rows, C, c, rl, s1, s2);
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    s1 = s2 = 0.0;
    for (int k = 0; k < 16; ++k) {
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
    // count == 0: BatchNorm ran on running statistics (eval mode): no batch-mean terms in dx
    coef[C + c] = count > 0.0 ? (float)(s1 / count) : 0.f;
    coef[2 * C + c] = count > 0.0 ? (float)(s2 / count) : 0.f;
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ mask, const T* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ coef, T* __restrict__ dy, T* __restrict__ g_out,
                                    long long M, int C) {
  const int cgs = C >> 3;
  const long long total = M * cgs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cgs) * 8;
    const long long off = i * 8;
    float gv[8], yv[8], mu[8], is[8], ca[8], cb[8], cc[8];
    QtVec8<T>::load(g + off, gv);
    QtVec8<T>::load(y + off, yv);
    if (mask) {
      float mv[8];
      QtVec8<T>::load(mask + off, mv);
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = mv[e] > 0.f ? gv[e] : 0.f;
    }
    if (g_out) QtVec8<T>::store(g_out + off, gv);
    load8f(mean + c0, mu);
    load8f(invstd + c0, is);
    load8f(coef + c0, ca);
    load8f(coef + C + c0, cb);
    load8f(coef + 2 * C + c0, cc);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ca[e] * (gv[e] - cb[e] - (yv[e] - mu[e]) * is[e] * cc[e]);
    QtVec8<T>::store(dy + off, o);
  }
}

// The same pass for the training step's backward chain (bf16, no mask / g_out operand): dy = P g + Q y + R with the
// per-channel P = ca, Q = -ca cc invstd, R = -ca cb - Q mean folded ONCE per thread -- the grid stride is a multiple of the
// channel groups per row, so a thread keeps its eight channels -- and two 16-byte loads of each operand in flight.  In the step
// this kernel runs on the main stream BESIDE the weight-gradient stream's MFMA kernels (which leave 84 of a SIMD's 512
// registers per lane): 44 VGPRs instead of 60 put two of its waves on a SIMD instead of one, and the operand loads of two
// elements overlap -- the pass sits in the chain  data gradient -> finalize -> apply -> data gradient  that bounds the
// backward pass (round 4).  QTCNN_BN_APPLY_LIGHT=0: the general kernel above (same-box A/B).
__global__ __launch_bounds__(256) void bn_bwd_apply_light_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ y,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ coef, bf16_t* __restrict__ dy,
                                                                 long long total, int cgs, int C) {
  // (FOUR channels = 8 bytes per thread and element: 12 coefficient registers instead of 24; total / cgs count 4-channel groups;
  //  32-bit element indices -- the launcher checks total < 2^29 -- so an address is a scalar base + one offset register)
  const unsigned stride = gridDim.x * 256u;   // (a multiple of cgs: checked by the launcher)
  unsigned i = blockIdx.x * 256u + threadIdx.x;
  const unsigned n = (unsigned)total;
  const int c0 = (int)(i % (unsigned)cgs) * 4;
  float P[4], Q[4], R[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ca = coef[c0 + e], cb = coef[C + c0 + e], cc = coef[2 * C + c0 + e];
    P[e] = ca;
    Q[e] = -ca * cc * invstd[c0 + e];
    R[e] = -ca * cb - Q[e] * mean[c0 + e];
  }
  const uint2* __restrict__ g2 = reinterpret_cast<const uint2*>(g);
  const uint2* __restrict__ y2 = reinterpret_cast<const uint2*>(y);
  uint2* __restrict__ d2 = reinterpret_cast<uint2*>(dy);
  auto pack2 = [](float lo, float hi) -> unsigned {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
    return __builtin_bit_cast(unsigned, v);
  };
  auto one = [&](const uint2& gu, const uint2& yu, unsigned at) {
    uint2 o;
    o.x = pack2(P[0] * __uint_as_float(gu.x << 16) + Q[0] * __uint_as_float(yu.x << 16) + R[0],
                P[1] * __uint_as_float(gu.x & 0xffff0000u) + Q[1] * __uint_as_float(yu.x & 0xffff0000u) + R[1]);
    o.y = pack2(P[2] * __uint_as_float(gu.y << 16) + Q[2] * __uint_as_float(yu.y << 16) + R[2],
                P[3] * __uint_as_float(gu.y & 0xffff0000u) + Q[3] * __uint_as_float(yu.y & 0xffff0000u) + R[3]);
    d2[at] = o;
  };
#pragma unroll 1
  for (; i + stride < n; i += 2u * stride) {   // two elements in flight (i + stride < 2^31: total < 2^29)
    const uint2 ga = g2[i], ya = y2[i], gb = g2[i + stride], yb = y2[i + stride];
    one(ga, ya, i);
    one(gb, yb, i + stride);
  }
  for (; i < n; i += stride) one(g2[i], y2[i], i);
}

// ---------------------------------------------------------------------------------
// Stem: a = relu(y*scale+shift) on [B][112][112][64]; 3x3/2 pad 1 max pool -> [B][56][56][64]
// (+ argmax position 0..8 in scan order, first maximum wins like ATen).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void stem_pool_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                 const float* __restrict__ shift, T* __restrict__ pooled,
                                 unsigned char* __restrict__ argmax, T* __restrict__ y_at_max, int batch) {
  constexpr int H = 112, W = 112, C = 64, PH = 56, PW = 56;
  const long long total = (long long)batch * PH * PW * (C / 8);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % (C / 8)) * 8;
    const int pw = (int)((i / (C / 8)) % PW);
    const int ph = (int)((i / ((C / 8) * PW)) % PH);
    const int n = (int)(i / ((long long)(C / 8) * PW * PH));
    float sc[8], sh[8];
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
    float best[8], braw[8];
    int bidx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -INFINITY;
      braw[e] = 0.f;
      bidx[e] = 0;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ph * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = pw * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        float v[8];
        QtVec8<T>::load(y + (((long long)n * H + h) * W + w) * C + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = fmaxf(v[e] * sc[e] + sh[e], 0.f);
          if (a > best[e]) {
            best[e] = a;
            braw[e] = v[e];
            bidx[e] = kh * 3 + kw;
          }
        }
      }
    }
    const long long off = (((long long)n * PH + ph) * PW + pw) * C + c0;
    QtVec8<T>::store(pooled + off, best);
    if (argmax) {
      uint2 pk;
      pk.x = bidx[0] | (bidx[1] << 8) | (bidx[2] << 16) | (bidx[3] << 24);
      pk.y = bidx[4] | (bidx[5] << 8) | (bidx[6] << 16) | (bidx[7] << 24);
      *reinterpret_cast<uint2*>(argmax + off) = pk;
    }
    if (y_at_max) QtVec8<T>::store(y_at_max + off, braw);
  }
}

// d(loss)/d(conv1 output) of the stem in one pass, 2 x 2 conv1 positions per thread: the four
// positions (2a..2a+1, 2b..2b+1) receive gradient from the pooled cells (a,b), (a,b+1), (a+1,b),
// (a+1,b+1) only, so a thread loads each cell once (a per-position gather loads 9 cells for the same
// four outputs) and has 12 independent 16-byte loads in flight.  Same sums in the same order as
// stem_pool_bwd_kernel<T, 2>.
template <typename T>
__global__ __launch_bounds__(256) void stem_bn_bwd_apply2x2_kernel(
    const T* __restrict__ dpooled, const unsigned char* __restrict__ argmax, const T* __restrict__ y,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ out, int batch) {
  constexpr int H = 112, W = 112, C = 64, PH = 56, PW = 56;
  const long long total = (long long)batch * PH * PW * (C / 8);
  const int c0 = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) % (C / 8)) * 8;
  float sc[8], sh[8], mu[8], is[8], ca[8], cb[8], cc[8];
  load8f(scale + c0, sc);
  load8f(shift + c0, sh);
  load8f(mean + c0, mu);
  load8f(invstd + c0, is);
  load8f(coef + c0, ca);
  load8f(coef + C + c0, cb);
  load8f(coef + 2 * C + c0, cc);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)((i / (C / 8)) % PW);
    const int a = (int)((i / ((C / 8) * PW)) % PH);
    const int n = (int)(i / ((long long)(C / 8) * PW * PH));
    const bool a1 = a + 1 < PH, b1 = b + 1 < PW;
    const long long cell = (((long long)n * PH + a) * PW + b) * C + c0;
    uint2 k00 = *reinterpret_cast<const uint2*>(argmax + cell), k01 = make_uint2(0xffffffffu, 0xffffffffu), k10 = k01, k11 = k01;
    float d00[8], d01[8], d10[8], d11[8];
    QtVec8<T>::load(dpooled + cell, d00);
#pragma unroll
    for (int e = 0; e < 8; ++e) d01[e] = d10[e] = d11[e] = 0.f;
    if (b1) {
      k01 = *reinterpret_cast<const uint2*>(argmax + cell + C);
      QtVec8<T>::load(dpooled + cell + C, d01);
    }
    if (a1) {
      k10 = *reinterpret_cast<const uint2*>(argmax + cell + PW * C);
      QtVec8<T>::load(dpooled + cell + PW * C, d10);
    }
    if (a1 && b1) {
      k11 = *reinterpret_cast<const uint2*>(argmax + cell + PW * C + C);
      QtVec8<T>::load(dpooled + cell + PW * C + C, d11);
    }
    const long long pos = (((long long)n * H + 2 * a) * W + 2 * b) * C + c0;
    float y00[8], y01[8], y10[8], y11[8];
    QtVec8<T>::load(y + pos, y00);
    QtVec8<T>::load(y + pos + C, y01);
    QtVec8<T>::load(y + pos + W * C, y10);
    QtVec8<T>::load(y + pos + W * C + C, y11);
    float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sft = (e & 3) * 8;
      const int i00 = ((e < 4 ? k00.x : k00.y) >> sft) & 0xff, i01 = ((e < 4 ? k01.x : k01.y) >> sft) & 0xff;
      const int i10 = ((e < 4 ? k10.x : k10.y) >> sft) & 0xff, i11 = ((e < 4 ? k11.x : k11.y) >> sft) & 0xff;
      // taps kh*3+kw with kh = h - (2*ph - 1), kw = w - (2*pw - 1)
      float g00 = i00 == 4 ? d00[e] : 0.f;
      float g01 = (i00 == 5 ? d00[e] : 0.f) + (i01 == 3 ? d01[e] : 0.f);
      float g10 = (i00 == 7 ? d00[e] : 0.f) + (i10 == 1 ? d10[e] : 0.f);
      float g11 = (i00 == 8 ? d00[e] : 0.f);
      g11 += (i01 == 6 ? d01[e] : 0.f);
      g11 += (i10 == 2 ? d10[e] : 0.f);
      g11 += (i11 == 0 ? d11[e] : 0.f);
      g00 = (y00[e] * sc[e] + sh[e] > 0.f) ? g00 : 0.f;
      g01 = (y01[e] * sc[e] + sh[e] > 0.f) ? g01 : 0.f;
      g10 = (y10[e] * sc[e] + sh[e] > 0.f) ? g10 : 0.f;
      g11 = (y11[e] * sc[e] + sh[e] > 0.f) ? g11 : 0.f;
      o00[e] = ca[e] * (g00 - cb[e] - (y00[e] - mu[e]) * is[e] * cc[e]);
      o01[e] = ca[e] * (g01 - cb[e] - (y01[e] - mu[e]) * is[e] * cc[e]);
      o10[e] = ca[e] * (g10 - cb[e] - (y10[e] - mu[e]) * is[e] * cc[e]);
      o11[e] = ca[e] * (g11 - cb[e] - (y11[e] - mu[e]) * is[e] * cc[e]);
    }
    QtVec8<T>::store(out + pos, o00);
    QtVec8<T>::store(out + pos + C, o01);
    QtVec8<T>::store(out + pos + W * C, o10);
    QtVec8<T>::store(out + pos + W * C + C, o11);
  }
}

// BatchNorm-backward sums of the stem from the POOLED side: every pooled cell sends its gradient
// to exactly one conv1 position (its argmax), so
//   sum_pos g = sum_cells [bn(y*) > 0] d,   sum_pos g*xhat = sum_cells [bn(y*) > 0] d * (y* - mean) * invstd
// with y* = the raw conv1 output at the argmax (saved by stem_pool_kernel).  Reads 2 x 103 MB at
// B = 256 instead of the 822 MB a reduction over the conv1 map costs.  One partial row per block.
template <typename T>
__global__ __launch_bounds__(256) void stem_bn_bwd_sums_kernel(const T* __restrict__ dpooled, const T* __restrict__ y_at_max,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               float* __restrict__ partial, long long total) {
  constexpr int C = 64;
  const int c0 = (threadIdx.x & 7) * 8;  // gridDim.x * blockDim.x is a multiple of 8
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
  load8f(scale + c0, sc);
  load8f(shift + c0, sh);
  load8f(mean + c0, mu);
  load8f(invstd + c0, is);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float d[8], v[8];
    QtVec8<T>::load(dpooled + i * 8, d);
    QtVec8<T>::load(y_at_max + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = (v[e] * sc[e] + sh[e] > 0.f) ? d[e] : 0.f;
      s1[e] += g;
      s2[e] += g * (v[e] - mu[e]) * is[e];
    }
  }
  __shared__ float red[32][64][2];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[rl][cg * 8 + e][0] = s1[e];
    red[rl][cg * 8 + e][1] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < 32; ++r) {
      a += red[r][threadIdx.x][0];
      b += red[r][threadIdx.x][1];
    }
    partial[((long long)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    partial[((long long)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// g[n][h][w][c] = (y*scale+shift > 0) * sum over the <=4 pooled cells whose argmax is (h,w)
// MODE 0: store g.   MODE 1: BatchNorm-backward partial sums of g (nothing stored).
// MODE 2: store dy = a*(g - b - xhat*c) directly.  Modes 1+2 never materialise g (411 MB at B=256).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(
    const T* __restrict__ dpooled, const unsigned char* __restrict__ argmax, const T* __restrict__ y,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ coef, float* __restrict__ partial,
    T* __restrict__ out, int batch) {
  constexpr int H = 112, W = 112, C = 64, PH = 56, PW = 56;
  const long long total = (long long)batch * H * W * (C / 8);
  // gridDim.x * blockDim.x is a multiple of 8, so a thread keeps its channel group
  const int c0 = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) % (C / 8)) * 8;
  float sc[8], sh[8], mu[8], is[8], ca[8], cb[8], cc[8], s1[8], s2[8];
  load8f(scale + c0, sc);
  load8f(shift + c0, sh);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = mu[e] = is[e] = ca[e] = cb[e] = cc[e] = 0.f;
  if (MODE >= 1) {
    load8f(mean + c0, mu);
    load8f(invstd + c0, is);
  }
  if (MODE == 2) {
    load8f(coef + c0, ca);
    load8f(coef + C + c0, cb);
    load8f(coef + 2 * C + c0, cc);
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)((i / (C / 8)) % W);
    const int h = (int)((i / ((C / 8) * W)) % H);
    const int n = (int)(i / ((long long)(C / 8) * W * H));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // pooled rows covering h: 2*ph-1 <= h <= 2*ph+1
    const int ph_lo = h >> 1, ph_hi = (h + 1) >> 1;
    const int pw_lo = w >> 1, pw_hi = (w + 1) >> 1;
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
      if (ph >= PH) continue;
      const int kh = h - (ph * 2 - 1);
      for (int pw = pw_lo; pw <= pw_hi; ++pw) {
        if (pw >= PW) continue;
        const int kw = w - (pw * 2 - 1);
        const int me = kh * 3 + kw;
        const long long off = (((long long)n * PH + ph) * PW + pw) * C + c0;
        const uint2 pk = *reinterpret_cast<const uint2*>(argmax + off);
        float d[8];
        QtVec8<T>::load(dpooled + off, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int idx = ((e < 4 ? pk.x : pk.y) >> ((e & 3) * 8)) & 0xff;
          if (idx == me) acc[e] += d[e];
        }
      }
    }
    const long long off = (((long long)n * H + h) * W + w) * C + c0;
    float v[8];
    QtVec8<T>::load(y + off, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (v[e] * sc[e] + sh[e] > 0.f) ? acc[e] : 0.f;
    if (MODE == 0) {
      QtVec8<T>::store(out + off, acc);
    } else if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += acc[e];
        s2[e] += acc[e] * (v[e] - mu[e]) * is[e];
      }
    } else {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = ca[e] * (acc[e] - cb[e] - (v[e] - mu[e]) * is[e] * cc[e]);
      QtVec8<T>::store(out + off, o);
    }
  }
  if (MODE == 1) {
    __shared__ float red[32][64][2];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;  // blockDim.x == 256: 8 channel groups x 32 lanes
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[rl][cg * 8 + e][0] = s1[e];
      red[rl][cg * 8 + e][1] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < 32; ++r) {
        a += red[r][threadIdx.x][0];
        b += red[r][threadIdx.x][1];
      }
      partial[((long long)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
      partial[((long long)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
    }
  }
}

// ---------------------------------------------------------------------------------
// Global average pool [B][HW][C] -> dst[b*ld + col0 + c], and its backward fused
// with the ReLU mask of the pooled map:  g[b][p][c] = d[b*ld+col0+c]/HW * (x>0).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ dst, int batch, int HW, int C, int ld,
                               int col0) {
  const int cgs = C >> 3;
  const int total = batch * cgs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c0 = (i % cgs) * 8, b = i / cgs;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll 7
    for (int p = 0; p < HW; ++p) {  // 7 independent 16-byte loads in flight (HW = 49 here)
      float v[8];
      QtVec8<T>::load(x + ((long long)b * HW + p) * C + c0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] *= inv;
    QtVec8<T>::store(dst + (long long)b * ld + col0 + c0, s);
  }
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ d, const T* __restrict__ x, T* __restrict__ g, int batch,
                                   int HW, int C, int ld, int col0) {
  const int cgs = C >> 3;
  const long long total = (long long)batch * HW * cgs;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cgs) * 8;
    const int b = (int)(i / ((long long)cgs * HW));
    float dv[8], xv[8];
    QtVec8<T>::load(d + (long long)b * ld + col0 + c0, dv);
    QtVec8<T>::load(x + i * 8, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = xv[e] > 0.f ? dv[e] * inv : 0.f;
    QtVec8<T>::store(g + i * 8, dv);
  }
}

// ---------------------------------------------------------------------------------
// Quadrant head tail: q[B*4][7][7][128] (already conv+bias+ReLU) -> 2x2/2 max pool
// (floor: 7->3) -> dst[b*ld + col0 + quad*1152 + c*9 + ph*3 + pw]  (flatten(1) of NCHW,
// then torch.cat order of models.py:291-294).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void quad_pool_kernel(const T* __restrict__ q, T* __restrict__ dst, int batch, int ld, int col0) {
  constexpr int C = 128;
  const int total = batch * 4 * 9 * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C;
    const int cell = (i / C) % 9;
    const int img = i / (C * 9);  // b*4 + quad
    const int ph = cell / 3, pw = cell % 3;
    const T* s = q + (((long long)img * 7 + ph * 2) * 7 + pw * 2) * C + c;
    float m = qt_to_f32<T>(s[0]);
    m = fmaxf(m, qt_to_f32<T>(s[C]));
    m = fmaxf(m, qt_to_f32<T>(s[7 * C]));
    m = fmaxf(m, qt_to_f32<T>(s[8 * C]));
    const int b = img >> 2, quad = img & 3;
    if constexpr (sizeof(T) == 4)
      dst[(long long)b * ld + col0 + quad * 1152 + c * 9 + cell] = m;
    else
      dst[(long long)b * ld + col0 + quad * 1152 + c * 9 + cell] = (bf16_t)m;
  }
}

// One thread per (region image, 2x2 pooling window, 8 channels): the four window pixels are read and written as 16-byte
// vectors (the round-1 kernel was one thread per ELEMENT with four scalar loads each: 65 us in the train step for 40 MB);
// the 13 pixels of row / column 6, which MaxPool2d(2,2) drops on a 7x7 map, get their zeros from 13 more items per image.
template <typename T>
__global__ void quad_pool_bwd_kernel(const T* __restrict__ d, const T* __restrict__ q, T* __restrict__ dq, int batch,
                                     int ld, int col0) {
  constexpr int C = 128, G = C / 8, ITEMS = (9 + 13) * G;   // per region image
  const long long total = (long long)batch * 4 * ITEMS;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(i / ITEMS), r = (int)(i - (long long)img * ITEMS);
    const int g = r % G, k = r / G;
    const int c0 = g * 8;
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k >= 9) {   // the dropped row / column: pixels (6, 0..6) and (0..5, 6)
      const int e = k - 9;
      const int h = e < 7 ? 6 : e - 7, w = e < 7 ? e : 6;
      QtVec8<T>::store(dq + (((long long)img * 7 + h) * 7 + w) * C + c0, z);
      continue;
    }
    const int ph = k / 3, pw = k - ph * 3;
    const long long base = (((long long)img * 7 + ph * 2) * 7 + pw * 2) * C + c0;
    float v[4][8];
    QtVec8<T>::load(q + base, v[0]);
    QtVec8<T>::load(q + base + C, v[1]);
    QtVec8<T>::load(q + base + 7 * C, v[2]);
    QtVec8<T>::load(q + base + 8 * C, v[3]);
    const int b = img >> 2, quad = img & 3;
    const T* dp = d + (long long)b * ld + col0 + quad * 1152 + ph * 3 + pw;
    float o[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int am = 0;
      float best = v[0][e];
      if (v[1][e] > best) { best = v[1][e]; am = 1; }
      if (v[2][e] > best) { best = v[2][e]; am = 2; }
      if (v[3][e] > best) { best = v[3][e]; am = 3; }
      const float gd = best > 0.f ? qt_to_f32<T>(dp[(c0 + e) * 9]) : 0.f;   // best > 0: ReLU passes gradient
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m][e] = m == am ? gd : 0.f;
    }
    QtVec8<T>::store(dq + base, o[0]);
    QtVec8<T>::store(dq + base + C, o[1]);
    QtVec8<T>::store(dq + base + 7 * C, o[2]);
    QtVec8<T>::store(dq + base + 8 * C, o[3]);
  }
}

// ---------------------------------------------------------------------------------
// Dropout (inverted, keep-probability 1-p) with a counter-based hash RNG:
// keep(i) depends only on (seed, i), so backward re-derives the mask from the
// forward output (kept and positive  <=>  out > 0 after the preceding ReLU).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32);
}

template <typename T>
__global__ void dropout_kernel(T* __restrict__ x, long long rows, int cols, int ld, unsigned long long seed, float p) {
  const long long total = rows * cols;
  const float inv = 1.f / (1.f - p);
  const unsigned thr = (unsigned)((double)p * 4294967296.0);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    T* e = x + r * ld + c;
    const bool keep = hash_u32(seed, (unsigned long long)i) >= thr;
    const float v = keep ? qt_to_f32<T>(*e) * inv : 0.f;
    if constexpr (sizeof(T) == 4)
      *e = v;
    else
      *e = (bf16_t)v;
  }
}

// g = (act > 0) ? g * mul : 0     (ReLU [+ dropout] backward from the forward output)
template <typename T>
__global__ void relu_mask_scale_kernel(T* __restrict__ g, const T* __restrict__ act, long long n, float mul) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = qt_to_f32<T>(act[i]) > 0.f ? qt_to_f32<T>(g[i]) * mul : 0.f;
    if constexpr (sizeof(T) == 4)
      g[i] = v;
    else
      g[i] = (bf16_t)v;
  }
}

// out[c] += sum over this block's row chunk of x[r*ld + c]  (out pre-zeroed unless accumulating)
template <typename T>
__global__ void col_sum_kernel(const T* __restrict__ x, long long rows, int cols, int ld, float* __restrict__ out,
                               int rows_per_block) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < cols)
    for (long long r = r0 + rl; r < r1; r += 4) s += qt_to_f32<T>(x[r * ld + c]);
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

}  // namespace

#define QT_DT_OK(dtype, name) QT_CHECK_ARG((dtype) == QT_F32 || (dtype) == QT_BF16, name ": bad dtype %d", (dtype))

// Long row counts are first folded 256:1 into the spare rows behind `rows`.
static int fold_partial(float*& partial, int& rows, int C, hipStream_t s) {
  if (rows <= 1024) return QT_OK;  // the finalize kernels (16 row groups) cover this directly
  const int S = qt_cdiv(rows, kFold);
  hipLaunchKernelGGL(stats_stage1_kernel, dim3(qt_cdiv(C, 64), S), dim3(256), 0, s, partial, rows, C);
  QT_CHECK_LAUNCH();
  partial += (long long)rows * 2 * C;
  rows = S;
  return QT_OK;
}

extern "C" int qt_stats_capacity_rows(int rows) { return rows <= 1024 ? rows : rows + qt_cdiv(rows, kFold); }

extern "C" int qt_bn_finalize(float* partial, int rows, int C, long long count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float momentum, float eps, float* mean, float* invstd,
                              float* scale, float* shift, void* stream) {
  QT_CHECK_ARG(partial && rows > 0 && C > 0 && count > 0 && mean && invstd && scale && shift,
               "qt_bn_finalize: bad argument");
  QT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "qt_bn_finalize: running stats must come in pairs");
  if (int st = fold_partial(partial, rows, C, static_cast<hipStream_t>(stream))) return st;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(qt_cdiv(C, kFinC)), dim3(16 * kFinC), 0, static_cast<hipStream_t>(stream), partial,
                     rows, C, (double)count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                     mean, invstd, scale, shift);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
  QT_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "qt_bn_eval_affine: bad argument");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(qt_cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), gamma,
                     beta, running_mean, running_var, eps, C, scale, shift);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_eval_affine_batched(const qt_bn_eval_item* items, int n, float eps, void* stream) {
  QT_CHECK_ARG(items && n > 0 && n <= BN_MAX_ITEMS, "qt_bn_eval_affine_batched: 1..%d items", BN_MAX_ITEMS);
  BnEvalBatchArgs a;
  for (int j = 0; j < n; ++j) {
    const qt_bn_eval_item& q = items[j];
    QT_CHECK_ARG(q.gamma && q.beta && q.running_mean && q.running_var && q.scale && q.shift && q.C > 0,
                 "qt_bn_eval_affine_batched: item %d incomplete", j);
    a.gamma[j] = q.gamma; a.beta[j] = q.beta; a.rmean[j] = q.running_mean; a.rvar[j] = q.running_var;
    a.scale[j] = q.scale; a.shift[j] = q.shift; a.C[j] = q.C;
    QT_CHECK_ARG((q.mean == nullptr) == (q.invstd == nullptr), "qt_bn_eval_affine_batched: item %d: mean / invstd come in pairs", j);
    a.mean[j] = q.mean; a.invstd[j] = q.invstd;
  }
  a.eps = eps;
  hipLaunchKernelGGL(bn_eval_affine_batched_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_act_mask(int dtype, const void* y, const float* scale, const float* shift, const void* residual,
                              const float* res_scale, const float* res_shift, int relu, void* out, unsigned char* mask_bits,
                              long long M, int C, void* stream) {
  QT_DT_OK(dtype, "qt_bn_act");
  QT_CHECK_ARG(y && scale && shift && out && M > 0 && C > 0 && C % 8 == 0, "qt_bn_act: bad argument");
  QT_CHECK_ARG((res_scale == nullptr) == (res_shift == nullptr), "qt_bn_act: res_scale/res_shift must come in pairs");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(M * (C / 8));
  if (dtype == QT_F32)
    hipLaunchKernelGGL(bn_act_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)y, scale, shift,
                       (const float*)residual, res_scale, res_shift, relu, (float*)out, mask_bits, M, C);
  else
    hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                       (const bf16_t*)residual, res_scale, res_shift, relu, (bf16_t*)out, mask_bits, M, C);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* residual,
                         const float* res_scale, const float* res_shift, int relu, void* out, long long M, int C,
                         void* stream) {
  return qt_bn_act_mask(dtype, y, scale, shift, residual, res_scale, res_shift, relu, out, nullptr, M, C, stream);
}

static int bn_bwd_rows_per_block(long long M, int C) {
  const int RL = 256 / (C / 8);
  long long rpb = (M + 2047) / 2048;  // at most 2048 blocks
  long long min_rows = (long long)RL * 8;
  if (rpb < min_rows) rpb = min_rows;
  return (int)rpb;
}

extern "C" int qt_bn_bwd_partial_rows(long long M, int C) {
  if (M <= 0 || C <= 0 || C % 8 || C > 2048) return QT_ERR_INVALID_ARG;
  return qt_cdiv(M, bn_bwd_rows_per_block(M, C));
}

extern "C" int qt_bn_bwd_reduce(int dtype, const void* g, const void* mask, const void* y, const float* mean,
                                const float* invstd, float* partial, long long M, int C, void* stream) {
  QT_DT_OK(dtype, "qt_bn_bwd_reduce");
  QT_CHECK_ARG(g && y && mean && invstd && partial && M > 0 && C >= 8 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0,
               "qt_bn_bwd_reduce: bad argument (C=%d)", C);
  const int rpb = bn_bwd_rows_per_block(M, C);
  const int grid = qt_cdiv(M, rpb);
  const int RL = 256 / (C / 8);
  const int lds = RL * C * 2 * 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(grid), dim3(256), lds, s, (const float*)g, (const float*)mask,
                       (const float*)y, mean, invstd, partial, M, C, rpb);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(grid), dim3(256), lds, s, (const bf16_t*)g,
                       (const bf16_t*)mask, (const bf16_t*)y, mean, invstd, partial, M, C, rpb);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_bwd_finalize(float* partial, int rows, int C, long long count, const float* gamma,
                                  const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef,
                                  void* stream) {
  QT_CHECK_ARG(partial && rows > 0 && C > 0 && count >= 0 && invstd && coef, "qt_bn_bwd_finalize: bad argument");
  if (int st = fold_partial(partial, rows, C, static_cast<hipStream_t>(stream))) return st;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(qt_cdiv(C, kFinC)), dim3(16 * kFinC), 0, static_cast<hipStream_t>(stream),
                     partial, rows, C, (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_bn_bwd_apply(int dtype, const void* g, const void* mask, const void* y, const float* mean,
                               const float* invstd, const float* coef, void* dy, void* g_out, long long M, int C,
                               void* stream) {
  QT_DT_OK(dtype, "qt_bn_bwd_apply");
  QT_CHECK_ARG(g && y && mean && invstd && coef && dy && M > 0 && C > 0 && C % 8 == 0, "qt_bn_bwd_apply: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(M * (C / 8));
  static int light = -1;
  if (light < 0) {
    const char* e = getenv("QTCNN_BN_APPLY_LIGHT");
    light = e ? atoi(e) : 1;
  }
  const int cgs = C / 4;   // (4-channel groups per row)
  const int lgrid = grid_for(M * cgs);
  if (light && dtype == QT_BF16 && !mask && !g_out && ((long long)lgrid * 256) % cgs == 0 && M * cgs < (1ll << 29)) {
    hipLaunchKernelGGL(bn_bwd_apply_light_kernel, dim3(lgrid), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)y, mean, invstd, coef,
                       (bf16_t*)dy, M * cgs, cgs, C);
    QT_CHECK_LAUNCH();
    return QT_OK;
  }
  if (dtype == QT_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)g, (const float*)mask,
                       (const float*)y, mean, invstd, coef, (float*)dy, (float*)g_out, M, C);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)mask,
                       (const bf16_t*)y, mean, invstd, coef, (bf16_t*)dy, (bf16_t*)g_out, M, C);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_stem_pool(int dtype, const void* y, const float* scale, const float* shift, void* pooled,
                            unsigned char* argmax, void* y_at_max, int batch, void* stream) {
  QT_DT_OK(dtype, "qt_stem_pool");
  QT_CHECK_ARG(y && scale && shift && pooled && batch > 0, "qt_stem_pool: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)batch * 56 * 56 * 8);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(stem_pool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)y, scale, shift,
                       (float*)pooled, argmax, (float*)y_at_max, batch);
  else
    hipLaunchKernelGGL(stem_pool_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                       (bf16_t*)pooled, argmax, (bf16_t*)y_at_max, batch);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

template <typename T>
static void launch_stem_bwd(int mode, int grid, hipStream_t s, const void* dpooled, const unsigned char* argmax,
                            const void* y, const float* scale, const float* shift, const float* mean,
                            const float* invstd, const float* coef, float* partial, void* out, int batch) {
  if (mode == 0)
    hipLaunchKernelGGL((stem_pool_bwd_kernel<T, 0>), dim3(grid), dim3(256), 0, s, (const T*)dpooled, argmax, (const T*)y,
                       scale, shift, mean, invstd, coef, partial, (T*)out, batch);
  else if (mode == 1)
    hipLaunchKernelGGL((stem_pool_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, (const T*)dpooled, argmax, (const T*)y,
                       scale, shift, mean, invstd, coef, partial, (T*)out, batch);
  else
    hipLaunchKernelGGL((stem_pool_bwd_kernel<T, 2>), dim3(grid), dim3(256), 0, s, (const T*)dpooled, argmax, (const T*)y,
                       scale, shift, mean, invstd, coef, partial, (T*)out, batch);
}

extern "C" int qt_stem_pool_bwd(int dtype, const void* dpooled, const unsigned char* argmax, const void* y,
                                const float* scale, const float* shift, void* g, int batch, void* stream) {
  QT_DT_OK(dtype, "qt_stem_pool_bwd");
  QT_CHECK_ARG(dpooled && argmax && y && scale && shift && g && batch > 0, "qt_stem_pool_bwd: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)batch * 112 * 112 * 8);
  if (dtype == QT_F32)
    launch_stem_bwd<float>(0, grid, s, dpooled, argmax, y, scale, shift, nullptr, nullptr, nullptr, nullptr, g, batch);
  else
    launch_stem_bwd<bf16_t>(0, grid, s, dpooled, argmax, y, scale, shift, nullptr, nullptr, nullptr, nullptr, g, batch);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

static int stem_bwd_rows(int batch) {
  const long long blocks = ((long long)batch * 112 * 112 * 8 + 255) / 256;
  return (int)(blocks > 2048 ? 2048 : blocks);
}
extern "C" int qt_stem_bn_bwd_rows(int batch) { return batch > 0 ? stem_bwd_rows(batch) : QT_ERR_INVALID_ARG; }

extern "C" int qt_stem_bn_bwd_reduce(int dtype, const void* dpooled, const unsigned char* argmax, const void* y,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     float* partial, int batch, void* stream) {
  QT_DT_OK(dtype, "qt_stem_bn_bwd_reduce");
  QT_CHECK_ARG(dpooled && argmax && y && scale && shift && mean && invstd && partial && batch > 0,
               "qt_stem_bn_bwd_reduce: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = stem_bwd_rows(batch);
  if (dtype == QT_F32)
    launch_stem_bwd<float>(1, grid, s, dpooled, argmax, y, scale, shift, mean, invstd, nullptr, partial, nullptr, batch);
  else
    launch_stem_bwd<bf16_t>(1, grid, s, dpooled, argmax, y, scale, shift, mean, invstd, nullptr, partial, nullptr, batch);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// The same sums in a light-footprint form (bf16; round 4): in the step this kernel is the last link of the main stream's chain
// in front of the stem's weight gradient and runs beside the weight-gradient stream's last layer-1 launch, which leaves a CU
// 5 KB of LDS and 84 registers per lane: the kernel above (74 VGPRs, 16 KB of LDS) measured 103 us there against 41 us alone.
// Four channels (8 bytes) per thread, mask threshold and BatchNorm constants folded (sum g v is corrected by mean and invstd
// once at the end), lanes of a channel group added by two butterfly steps, one 2 KB exchange between the four waves.
static __global__ __launch_bounds__(256) void stem_bn_bwd_sums_light_kernel(const bf16_t* __restrict__ dpooled,
                                                                     const bf16_t* __restrict__ y_at_max,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     float* __restrict__ partial, unsigned total4) {
  constexpr int C = 64;
  const int c0 = (threadIdx.x & 15) * 4;   // gridDim.x * 256 is a multiple of 16
  float sc[4], sh[4], s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = scale[c0 + e];
    sh[e] = shift[c0 + e];
    s1[e] = s2[e] = 0.f;
  }
  const uint2* __restrict__ d2 = reinterpret_cast<const uint2*>(dpooled);
  const uint2* __restrict__ v2 = reinterpret_cast<const uint2*>(y_at_max);
  const unsigned stride = gridDim.x * 256u;
  auto add = [&](const uint2& du, const uint2& vu) {
    const float d[4] = {__uint_as_float(du.x << 16), __uint_as_float(du.x & 0xffff0000u), __uint_as_float(du.y << 16),
                        __uint_as_float(du.y & 0xffff0000u)};
    const float v[4] = {__uint_as_float(vu.x << 16), __uint_as_float(vu.x & 0xffff0000u), __uint_as_float(vu.y << 16),
                        __uint_as_float(vu.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = (v[e] * sc[e] + sh[e] > 0.f) ? d[e] : 0.f;
      s1[e] += g;
      s2[e] += g * v[e];
    }
  };
  unsigned i = blockIdx.x * 256u + threadIdx.x;
#pragma unroll 1
  for (; i + stride < total4; i += 2u * stride) {
    const uint2 da = d2[i], va = v2[i], db = d2[i + stride], vb = v2[i + stride];
    add(da, va);
    add(db, vb);
  }
  if (i < total4) add(d2[i], v2[i]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {   // sum g (v - mean) invstd = invstd (sum g v - mean sum g)
    s2[e] = (s2[e] - mean[c0 + e] * s1[e]) * invstd[c0 + e];
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {   // the four lanes of a wave that share a channel group
      s1[e] += __shfl_xor(s1[e], m, 64);
      s2[e] += __shfl_xor(s2[e], m, 64);
    }
  }
  __shared__ float red[4][64][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][c0 + e][0] = s1[e];
      red[wave][c0 + e][1] = s2[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float a = red[0][threadIdx.x][0] + red[1][threadIdx.x][0] + red[2][threadIdx.x][0] + red[3][threadIdx.x][0];
    const float b = red[0][threadIdx.x][1] + red[1][threadIdx.x][1] + red[2][threadIdx.x][1] + red[3][threadIdx.x][1];
    partial[((long long)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    partial[((long long)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

static int stem_sums_rows(int batch) {
  const long long blocks = ((long long)batch * 56 * 56 * 8 + 255) / 256;
  return (int)(blocks > 1024 ? 1024 : blocks);
}
extern "C" int qt_stem_bn_bwd_sums_rows(int batch) { return batch > 0 ? stem_sums_rows(batch) : QT_ERR_INVALID_ARG; }

extern "C" int qt_stem_bn_bwd_sums(int dtype, const void* dpooled, const void* y_at_max, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, float* partial,
                                   int batch, void* stream) {
  QT_DT_OK(dtype, "qt_stem_bn_bwd_sums");
  QT_CHECK_ARG(dpooled && y_at_max && scale && shift && mean && invstd && partial && batch > 0,
               "qt_stem_bn_bwd_sums: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = stem_sums_rows(batch);
  const long long total = (long long)batch * 56 * 56 * 8;
  static int light = -1;   // QTCNN_STEM_SUMS_LIGHT (default 1): 0 = the general kernel (same-box A/B)
  if (light < 0) {
    const char* e = getenv("QTCNN_STEM_SUMS_LIGHT");
    light = e ? atoi(e) : 1;
  }
  if (light && dtype == QT_BF16 && total * 2 < (1ll << 30)) {
    hipLaunchKernelGGL(stem_bn_bwd_sums_light_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)dpooled, (const bf16_t*)y_at_max,
                       scale, shift, mean, invstd, partial, (unsigned)(total * 2));
    QT_CHECK_LAUNCH();
    return QT_OK;
  }
  if (dtype == QT_F32)
    hipLaunchKernelGGL(stem_bn_bwd_sums_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dpooled,
                       (const float*)y_at_max, scale, shift, mean, invstd, partial, total);
  else
    hipLaunchKernelGGL(stem_bn_bwd_sums_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dpooled,
                       (const bf16_t*)y_at_max, scale, shift, mean, invstd, partial, total);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_stem_bn_bwd_apply(int dtype, const void* dpooled, const unsigned char* argmax, const void* y,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const float* coef, void* dy, int batch, void* stream) {
  QT_DT_OK(dtype, "qt_stem_bn_bwd_apply");
  QT_CHECK_ARG(dpooled && argmax && y && scale && shift && mean && invstd && coef && dy && batch > 0,
               "qt_stem_bn_bwd_apply: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((long long)batch * 56 * 56 * 8);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(stem_bn_bwd_apply2x2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dpooled, argmax,
                       (const float*)y, scale, shift, mean, invstd, coef, (float*)dy, batch);
  else
    hipLaunchKernelGGL(stem_bn_bwd_apply2x2_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dpooled, argmax,
                       (const bf16_t*)y, scale, shift, mean, invstd, coef, (bf16_t*)dy, batch);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

template <typename F>
static void by_dtype(int dtype, F&& f) {
  if (dtype == QT_F32)
    f(static_cast<float*>(nullptr));
  else
    f(static_cast<bf16_t*>(nullptr));
}
#define QT_T(tag) std::remove_pointer_t<decltype(tag)>

extern "C" int qt_avgpool(int dtype, const void* x, void* dst, int batch, int hw, int C, int ld, int col0,
                          void* stream) {
  QT_DT_OK(dtype, "qt_avgpool");
  QT_CHECK_ARG(x && dst && batch > 0 && hw > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && col0 % 8 == 0 && col0 + C <= ld,
               "qt_avgpool: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(avgpool_kernel<T>, dim3(grid_for((long long)batch * (C / 8), 64)), dim3(64), 0, s,
                       (const T*)x, (T*)dst, batch, hw, C, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_avgpool_bwd(int dtype, const void* d, const void* x, void* g, int batch, int hw, int C, int ld,
                              int col0, void* stream) {
  QT_DT_OK(dtype, "qt_avgpool_bwd");
  QT_CHECK_ARG(d && x && g && batch > 0 && hw > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && col0 % 8 == 0 && col0 + C <= ld,
               "qt_avgpool_bwd: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(grid_for((long long)batch * hw * (C / 8))), dim3(256), 0, s,
                       (const T*)d, (const T*)x, (T*)g, batch, hw, C, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_quad_pool(int dtype, const void* q, void* dst, int batch, int ld, int col0, void* stream) {
  QT_DT_OK(dtype, "qt_quad_pool");
  QT_CHECK_ARG(q && dst && batch > 0 && col0 >= 0 && col0 + 4 * 1152 <= ld, "qt_quad_pool: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(quad_pool_kernel<T>, dim3(grid_for((long long)batch * 4 * 9 * 128)), dim3(256), 0, s,
                       (const T*)q, (T*)dst, batch, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_quad_pool_bwd(int dtype, const void* d, const void* q, void* dq, int batch, int ld, int col0,
                                void* stream) {
  QT_DT_OK(dtype, "qt_quad_pool_bwd");
  QT_CHECK_ARG(d && q && dq && batch > 0 && col0 >= 0 && col0 + 4 * 1152 <= ld, "qt_quad_pool_bwd: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(quad_pool_bwd_kernel<T>, dim3(grid_for((long long)batch * 4 * 22 * 16)), dim3(256), 0, s,
                       (const T*)d, (const T*)q, (T*)dq, batch, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_dropout(int dtype, void* x, long long rows, int cols, int ld, unsigned long long seed, float p,
                          void* stream) {
  QT_DT_OK(dtype, "qt_dropout");
  QT_CHECK_ARG(x && rows > 0 && cols > 0 && ld >= cols && p >= 0.f && p < 1.f, "qt_dropout: bad argument (p=%f)", p);
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid_for(rows * cols)), dim3(256), 0, s, (T*)x, rows, cols, ld, seed, p);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_relu_mask_scale(int dtype, void* g, const void* act, long long n, float mul, void* stream) {
  QT_DT_OK(dtype, "qt_relu_mask_scale");
  QT_CHECK_ARG(g && act && n > 0, "qt_relu_mask_scale: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(relu_mask_scale_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (T*)g, (const T*)act, n, mul);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_col_sum(int dtype, const void* x, long long rows, int cols, int ld, float* out, int accumulate,
                          void* stream) {
  QT_DT_OK(dtype, "qt_col_sum");
  QT_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "qt_col_sum: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!accumulate && hipMemsetAsync(out, 0, (size_t)cols * 4, s) != hipSuccess) {
    qt_set_error("qt_col_sum: hipMemsetAsync failed");
    return QT_ERR_LAUNCH;
  }
  const int rpb = 128;
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(col_sum_kernel<T>, dim3(qt_cdiv(cols, 64), qt_cdiv(rows, rpb)), dim3(256), 0, s, (const T*)x,
                       rows, cols, ld, out, rpb);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

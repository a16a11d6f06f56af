// Memory-bound pieces of the 3-D clip models (BASELINE config 4):
//   Quadtree3DCNN  /root/reference/3dcnn/models.py:96-214   (5 x Conv3d 3x3x3 + BatchNorm3d + ReLU + MaxPool3d, T = 8 clips)
//   Ji3DCNN        /root/reference/cnn+lstm/models.py:93-142 (3 x Conv3d block)
//
// Layout.  Clip activations are TIME-MAJOR NHWC: [T][B][H][W][C].  A 3x3x3 convolution with padding 1 is then the sum
// over the frame tap kt of three 3x3 2-D convolutions whose operands are CONTIGUOUS sub-batches of frames --
//     out[t] += conv2d(in[t + kt - 1], W[:, :, kt])   for the t with 0 <= t + kt - 1 < T   (images t*B .. t*B+B-1)
// -- so the MFMA work runs on the implicit-GEMM kernels of conv_igemm.hip / conv_pt.hip / conv_wgrad.hip unchanged (frames
// past the clip's ends are simply not part of the launch: no padding frames, no wasted MFMAs); the first layer (3 input
// channels) is packed to one 128-wide K row per pixel (27 taps x 3 channels) and runs as a 1x1 convolution.
// What lives here is HBM-bound: the clip packing, BatchNorm3d batch statistics of a finished map (the conv epilogue's
// statistics cannot be used: a map is finished by the last of three launches), MaxPool3d (1,2,2) / (2,2,2) with its
// backward, AdaptiveAvgPool3d(1,1,1) with its backward.  8 channels (16 B of bf16) per thread, f32 arithmetic.
#include <atomic>

#include "qt_common.h"

namespace {

// ---- clip packing: [B][T][3][H][W] f32 (image_sequence_input, 3dcnn/models.py:189 permutes it to B,C,T,H,W) ->
// [T][B][H][W][128]: element ((kt*3 + kh)*3 + kw)*3 + c = x[b][t+kt-1][c][h+kh-1][w+kw-1] (0 outside), 81..127 = 0
template <typename T>
__global__ void pack_clip27_kernel(const float* __restrict__ x, T* __restrict__ dst, int B, int Tn, int H, int W) {
  const long long rows = (long long)Tn * B * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < rows * 16; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i >> 4;
    const int chunk = (int)(i & 15);
    // (frame, row, column) by 32-bit divisions: rows < 2^31 (checked by the launcher); the 64-bit runtime divisions this
    // replaces were most of the kernel's time (2.0 ms for 3.3 GB: 1.6 TB/s)
    const unsigned urow = (unsigned)row, uw = (unsigned)W, uh = (unsigned)H;
    const unsigned q1 = urow / uw;
    const int w = (int)(urow - q1 * uw);
    const unsigned q2 = q1 / uh;
    const int h = (int)(q1 - q2 * uh);
    const unsigned q3 = q2 / (unsigned)B;
    const int b = (int)(q2 - q3 * (unsigned)B);
    const int t = (int)q3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = chunk * 8 + e;
      v[e] = 0.f;
      if (k < 81) {
        const int c = k % 3, tap = k / 3;
        const int kw = tap % 3, kh = (tap / 3) % 3, kt = tap / 9;
        const int tt = t + kt - 1, hh = h + kh - 1, ww = w + kw - 1;
        if ((unsigned)tt < (unsigned)Tn && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
          v[e] = x[((((long long)b * Tn + tt) * 3 + c) * H + hh) * W + ww];
      }
    }
    QtVec8<T>::store(dst + row * 128 + chunk * 8, v);
  }
}

// ---- per-channel sum / sum of squares of y [M][C] over row slabs -> partial [rows][2][C] (for qt_bn_finalize) ----
template <typename T>
__global__ void bn_stats_kernel(const T* __restrict__ y, long long M, int C, float* __restrict__ partial, int slab) {
  // block = 256 threads = (C/8 channel groups) x (256 / (C/8) row lanes); one partial row per block
  const int groups = C / 8;
  const int lanes = 256 / groups;
  const int g = threadIdx.x % groups, rl = threadIdx.x / groups;
  const long long r0 = (long long)blockIdx.x * slab, r1 = min(M, r0 + slab);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (rl < lanes) {
    for (long long r = r0 + rl; r < r1; r += lanes) {
      float v[8];
      QtVec8<T>::load(y + r * C + g * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += v[e];
        q[e] += v[e] * v[e];
      }
    }
  }
  __shared__ float red[256 * 16];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x * 16 + e] = s[e];
    red[threadIdx.x * 16 + 8 + e] = q[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {   // fixed order over the row lanes: deterministic
    const int which = c / C, ch = c % C;
    const int gg = ch / 8, e = ch % 8;
    float a = 0.f;
    for (int l = 0; l < lanes; ++l) a += red[(l * groups + gg) * 16 + which * 8 + e];
    partial[((long long)blockIdx.x * 2 + which) * C + ch] = a;
  }
}

// ---- MaxPool3d, kernel = stride = (PT, 2, 2), floor mode: x [T][B][H][W][C] -> out [T/PT][B][H/2][W/2][C];
// arg (u8) = index of the FIRST maximum in (t, h, w) scan order, as torch's kernel keeps it ----
template <typename T, int PT>
__global__ void pool3d_max_kernel(const T* __restrict__ x, T* __restrict__ out, unsigned char* __restrict__ arg, int Tn,
                                  int B, int H, int W, int C) {
  const int To = Tn / PT, Ho = H / 2, Wo = W / 2, G = C / 8;
  const long long n = (long long)To * B * Ho * Wo * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int b = (int)(r % B);
    const int to = (int)(r / B);
    float best[8];
    unsigned char idx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; idx[e] = 0; }
#pragma unroll
    for (int dt = 0; dt < PT; ++dt)
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const long long src = ((((long long)(to * PT + dt) * B + b) * H + ho * 2 + dh) * W + wo * 2 + dw) * C + g * 8;
          float v[8];
          QtVec8<T>::load(x + src, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (v[e] > best[e]) { best[e] = v[e]; idx[e] = (unsigned char)((dt * 2 + dh) * 2 + dw); }
        }
    const long long o = ((((long long)to * B + b) * Ho + ho) * Wo + wo) * C + g * 8;
    QtVec8<T>::store(out + o, best);
    if (arg) {
      uint2 pk;
      pk.x = idx[0] | (idx[1] << 8) | (idx[2] << 16) | ((unsigned)idx[3] << 24);
      pk.y = idx[4] | (idx[5] << 8) | (idx[6] << 16) | ((unsigned)idx[7] << 24);
      *reinterpret_cast<uint2*>(arg + o) = pk;
    }
  }
}
// dx [T][B][H][W][C] (every element written: windows tile the pooled part; the floor-mode remainder gets zeros)
template <typename T, int PT>
__global__ void pool3d_max_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg, T* __restrict__ dx,
                                      int Tn, int B, int H, int W, int C) {
  const int To = Tn / PT, Ho = H / 2, Wo = W / 2, G = C / 8;
  const long long n = (long long)Tn * B * H * W * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int b = (int)(r % B);
    const int t = (int)(r / B);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const int to = t / PT, ho = h / 2, wo = w / 2;
    if (to < To && ho < Ho && wo < Wo) {
      const long long o = ((((long long)to * B + b) * Ho + ho) * Wo + wo) * C + g * 8;
      const uint2 pk = *reinterpret_cast<const uint2*>(arg + o);
      const unsigned me = (unsigned)(((t - to * PT) * 2 + (h & 1)) * 2 + (w & 1));
      float d[8];
      QtVec8<T>::load(dout + o, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
        v[e] = a == me ? d[e] : 0.f;
      }
    }
    QtVec8<T>::store(dx + i * 8, v);
  }
}

// ---- BatchNorm3d (as scale / shift) + ReLU + MaxPool3d (PT, 2, 2) in ONE pass over the raw conv output (round 3) ----
// The train forward of a conv block wrote a = relu(bn(y)) (one pass over the map) and pooled it (a second one), although
// nothing but the pool reads `a`: the next block consumes the pooled map, the backward's ReLU mask is `pooled > 0` at the
// argmax and zero gradient elsewhere.  Here a thread applies the affine + ReLU to its window on the fly -- rounded to the
// activation type before the comparison, exactly the values the two-kernel form compared -- and also keeps the RAW conv
// output at the argmax (ymax): the BatchNorm-backward sums then come from the pooled side (every pooled cell sends its
// gradient to exactly one position).  Floor mode, first maximum in (t, h, w) scan order, as pool3d_max_kernel.
template <typename T, int PT>
__global__ void pool3d_bn_relu_max_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                          T* __restrict__ out, unsigned char* __restrict__ arg, T* __restrict__ ymax, int Tn, int B,
                                          int H, int W, int C, int Cy) {
  const int To = Tn / PT, Ho = H / 2, Wo = W / 2, G = C / 8;
  const long long n = (long long)To * B * Ho * Wo * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int b = (int)(r % B);
    const int to = (int)(r / B);
    float sc[8], sh[8], best[8], braw[8];
    const long long o = ((((long long)to * B + b) * Ho + ho) * Wo + wo) * C + g * 8;
    if (g * 8 >= Cy) {   // channels the conv output does not have (y has Cy <= C per row): the pooled map is zero there
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = 0.f;
      QtVec8<T>::store(out + o, best);
      if (arg) *reinterpret_cast<uint2*>(arg + o) = make_uint2(0u, 0u);
      if (ymax) QtVec8<T>::store(ymax + o, best);
      continue;
    }
    QtVec8<float>::load(scale + g * 8, sc);
    QtVec8<float>::load(shift + g * 8, sh);
    unsigned char idx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; braw[e] = 0.f; idx[e] = 0; }
#pragma unroll
    for (int dt = 0; dt < PT; ++dt)
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const long long src = ((((long long)(to * PT + dt) * B + b) * H + ho * 2 + dh) * W + wo * 2 + dw) * Cy + g * 8;
          float v[8];
          QtVec8<T>::load(y + src, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a = (float)(T)fmaxf(v[e] * sc[e] + sh[e], 0.f);   // (the stored activation's value)
            if (a > best[e]) { best[e] = a; braw[e] = v[e]; idx[e] = (unsigned char)((dt * 2 + dh) * 2 + dw); }
          }
        }
    QtVec8<T>::store(out + o, best);
    if (arg) {
      uint2 pk;
      pk.x = idx[0] | (idx[1] << 8) | (idx[2] << 16) | ((unsigned)idx[3] << 24);
      pk.y = idx[4] | (idx[5] << 8) | (idx[6] << 16) | ((unsigned)idx[7] << 24);
      *reinterpret_cast<uint2*>(arg + o) = pk;
    }
    if (ymax) QtVec8<T>::store(ymax + o, braw);
  }
}

// ---- max-pool backward + ReLU mask + BatchNorm backward in one pass: dy [T][B][H][W][C] from the pooled gradient ----
//   g = dout at the window's argmax where pooled > 0, else 0 (also 0 in the floor-mode remainder);  dy = a (g - b - xhat c)
// (coef = qt_bn_bwd_finalize's [3][C]).  The full-size gradient of the ReLU output is never materialised: it was written by
// pool3d_max_bwd and read twice (reduce, apply) -- four passes over the largest maps of the clip models.
template <typename T, int PT>
__global__ void pool3d_bn_bwd_apply_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg,
                                           const T* __restrict__ pooled, const T* __restrict__ y, const float* __restrict__ mean,
                                           const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ dy,
                                           int Tn, int B, int H, int W, int C, int Cy, int Cd) {
  const int To = Tn / PT, Ho = H / 2, Wo = W / 2, G = Cd / 8;
  const long long n = (long long)Tn * B * H * W * G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    long long r = i / G;
    const long long row = r;
    if (g * 8 >= Cy) {   // dy rows are Cd wide, y rows Cy: the padding channels of dy are zero
      float z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = 0.f;
      QtVec8<T>::store(dy + row * Cd + g * 8, z);
      continue;
    }
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int b = (int)(r % B);
    const int t = (int)(r / B);
    float gv[8], yv[8], mu[8], is[8], ca[8], cb[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = 0.f;
    const int to = t / PT, ho = h / 2, wo = w / 2;
    if (to < To && ho < Ho && wo < Wo) {
      const long long o = ((((long long)to * B + b) * Ho + ho) * Wo + wo) * C + g * 8;
      const uint2 pk = *reinterpret_cast<const uint2*>(arg + o);
      const unsigned me = (unsigned)(((t - to * PT) * 2 + (h & 1)) * 2 + (w & 1));
      float d[8], pv[8];
      QtVec8<T>::load(dout + o, d);
      QtVec8<T>::load(pooled + o, pv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
        gv[e] = (a == me && pv[e] > 0.f) ? d[e] : 0.f;
      }
    }
    QtVec8<T>::load(y + row * Cy + g * 8, yv);
    QtVec8<float>::load(mean + g * 8, mu);
    QtVec8<float>::load(invstd + g * 8, is);
    QtVec8<float>::load(coef + g * 8, ca);
    QtVec8<float>::load(coef + C + g * 8, cb);
    QtVec8<float>::load(coef + 2 * C + g * 8, cc);
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = ca[e] * (gv[e] - cb[e] - (yv[e] - mu[e]) * is[e] * cc[e]);
    QtVec8<T>::store(dy + row * Cd + g * 8, o8);
  }
}

// The same pass where no row carries padding channels (C = Cy = Cd: every block but the first), as a grid that stays resident
// (round 4): the thread's channel group -- and with it the five per-channel vectors, 10 loads of 32 B in front of every 16 bytes
// of y in the kernel above -- is fixed for all its rows (the stride is a multiple of C / 8), two rows are in flight per trip.
template <typename T, int PT>
__global__ __launch_bounds__(256) void pool3d_bn_bwd_apply_light_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg,
                                                                        const T* __restrict__ pooled, const T* __restrict__ y,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ coef, T* __restrict__ dy, int Tn, int B,
                                                                        int H, int W, int C) {
  const int To = Tn / PT, Ho = H / 2, Wo = W / 2, G = C / 8;
  const long long n = (long long)Tn * B * H * W * G, stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int g = (int)(i0 % G);
  float mu[8], is[8], ca[8], cb[8], cc[8];
  QtVec8<float>::load(mean + g * 8, mu);
  QtVec8<float>::load(invstd + g * 8, is);
  QtVec8<float>::load(coef + g * 8, ca);
  QtVec8<float>::load(coef + C + g * 8, cb);
  QtVec8<float>::load(coef + 2 * C + g * 8, cc);
  for (long long i = i0; i < n; i += 2 * stride) {
    float gv[2][8], yv[2][8];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long row = (i + u * stride) / G;
      live[u] = i + u * stride < n;
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[u][e] = 0.f;
      if (!live[u]) continue;
      long long r = row;
      const int w = (int)(r % W); r /= W;
      const int h = (int)(r % H); r /= H;
      const int b = (int)(r % B);
      const int t = (int)(r / B);
      const int to = t / PT, ho = h / 2, wo = w / 2;
      QtVec8<T>::load(y + row * C + g * 8, yv[u]);
      if (to < To && ho < Ho && wo < Wo) {
        const long long o = ((((long long)to * B + b) * Ho + ho) * Wo + wo) * C + g * 8;
        const uint2 pk = *reinterpret_cast<const uint2*>(arg + o);
        const unsigned me = (unsigned)(((t - to * PT) * 2 + (h & 1)) * 2 + (w & 1));
        float d[8], pv[8];
        QtVec8<T>::load(dout + o, d);
        QtVec8<T>::load(pooled + o, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned a = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
          gv[u][e] = (a == me && pv[e] > 0.f) ? d[e] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!live[u]) continue;
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = ca[e] * (gv[u][e] - cb[e] - (yv[u][e] - mu[e]) * is[e] * cc[e]);
      QtVec8<T>::store(dy + ((i + u * stride) / G) * C + g * 8, o8);
    }
  }
}

// ---- AdaptiveAvgPool3d((1,1,1)) + flatten: x [T][B][HW][C] -> dst[b*ld + col0 + c] (f32) = mean over t, hw ----
template <typename T>
__global__ __launch_bounds__(256) void avgpool_tb_kernel(const T* __restrict__ x, float* __restrict__ dst, int Tn, int B, int HW, int C,
                                                         int ld, int col0) {
  // workgroup (clip b, slice of <= 32 channel groups): 256 / slice position lanes per channel group, each adds its positions in
  // ascending order (four loads in flight), then the lanes are added in ascending order through LDS: deterministic.
  // (One thread per channel group walked all Tn * HW positions alone: 107 us for 32 clips of 2 x 49 x 512; one workgroup per
  // clip with 256 / G lanes, round 3: 72 us for 32 clips of 196 x 1024 -- 98 dependent loads per thread on 32 CUs.)
  __shared__ float red[256 * 8];
  const int b = blockIdx.x, G = C / 8;
  const int GS = G < 32 ? G : 32;                // channel groups of this workgroup (G is a multiple of 32 or < 32)
  const int R = 256 / GS;                        // position lanes per channel group
  const int gl = (int)threadIdx.x % GS, r = (int)threadIdx.x / GS;
  const int g = blockIdx.y * GS + gl;
  const int np = Tn * HW;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (g < G && r < R) {
    auto at = [&](int q) -> const T* {
      const int t = q / HW, pp = q - t * HW;
      return x + (((long long)t * B + b) * HW + pp) * C + g * 8;
    };
    int q = r;
    for (; q + 3 * R < np; q += 4 * R) {
      float v0[8], v1[8], v2[8], v3[8];
      QtVec8<T>::load(at(q), v0);
      QtVec8<T>::load(at(q + R), v1);
      QtVec8<T>::load(at(q + 2 * R), v2);
      QtVec8<T>::load(at(q + 3 * R), v3);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v0[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v1[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v2[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v3[e];
    }
    for (; q < np; q += R) {
      float v[8];
      QtVec8<T>::load(at(q), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  if (r == 0 && g < G) {
    const float inv = 1.f / (float)np;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
      for (int k = 0; k < R; ++k) a += red[(k * GS + gl) * 8 + e];
      dst[(long long)b * ld + col0 + g * 8 + e] = a * inv;
    }
  }
}
// g [T][B][HW][C] = d[b*ld + col0 + c] / (T*HW)
template <typename T>
__global__ void avgpool_tb_bwd_kernel(const float* __restrict__ d, T* __restrict__ g, int Tn, int B, int HW, int C, int ld, int col0) {
  const int G = C / 8;
  const long long n = (long long)Tn * B * HW * G;
  const float inv = 1.f / (float)(Tn * HW);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int gg = (int)(i % G);
    const int b = (int)((i / ((long long)G * HW)) % B);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = d[(long long)b * ld + col0 + gg * 8 + e] * inv;
    QtVec8<T>::store(g + i * 8, v);
  }
}

inline unsigned grid_for(long long n) {
  const long long b = (n + 255) / 256;
  return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

}  // namespace

namespace {

// ---- one launch per Conv3d block: operands of the 27-tap implicit GEMM + the zero / one padded per-channel vectors ----
// W [O][I][27] f32 (nn.Conv3d weight, taps (kt, kh, kw) row-major) ->
//   FIRST == 0: wf [Op][27][Ip] (forward operand), wd [Ip][27][Op] (data-gradient operand, optional), zero padded;
//   FIRST == 1: wf [Op][128], K index (tap * I + c) < 27 * I as qt_pack_clip27 lays the clip out, zero padded.
// vec_in[5] = bias, gamma, beta, running_mean, running_var ([O] each) -> vec_out [5][Op], padded with 0, 1, 0, 0, 1.
template <typename T>
__global__ void pack_conv3d_block_kernel(const float* __restrict__ W, T* __restrict__ wf, T* __restrict__ wd, int O, int I,
                                         int Op, int Ip, int first, const float* const* vec_in, float* __restrict__ vec_out) {
  const long long nf = first ? (long long)Op * 128 : (long long)Op * 27 * Ip;
  const long long total = nf + 5ll * Op;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (i >= nf) {
      const int v = (int)((i - nf) / Op), c = (int)((i - nf) % Op);
      const float* src = vec_in[v];
      vec_out[v * Op + c] = (c < O && src) ? src[c] : ((v == 1 || v == 4) ? 1.f : 0.f);
      continue;
    }
    if (first) {
      const int o = (int)(i >> 7), k = (int)(i & 127);
      float val = 0.f;
      if (o < O && k < 27 * I) {
        const int tap = k / I, c = k - tap * I;
        val = W[((long long)o * I + c) * 27 + tap];
      }
      wf[i] = (T)val;
    } else {
      const int c = (int)(i % Ip);
      const long long q = i / Ip;
      const int tap = (int)(q % 27), o = (int)(q / 27);
      const float val = (o < O && c < I) ? W[((long long)o * I + c) * 27 + tap] : 0.f;
      wf[i] = (T)val;
      if (wd) wd[((long long)c * 27 + tap) * Op + o] = (T)val;
    }
  }
}

// The same operands (FIRST == 0) through an LDS tile (round 4): the element-per-thread form above gathers W at a stride of 27
// floats and scatters wd at a stride of 27 * Op elements -- 118 us for conv3d_final_features' 7.1 M weights (0.5 TB/s), in
// line in every train forward.  A workgroup owns 32 output x 16 input channels x 27 taps: W arrives as 32 contiguous runs of
// 432 floats, wf leaves as runs of 16 input channels per (o, tap), wd as runs of 32 output channels per (c, tap); the odd row
// pitch keeps both read patterns of the tile off the same bank.  The last workgroup writes the per-channel vectors.
constexpr int PK3_TO = 32, PK3_TI = 16, PK3_PITCH = PK3_TI * 27 + 1;
template <typename T>
__global__ __launch_bounds__(256) void pack_conv3d_tile_kernel(const float* __restrict__ W, T* __restrict__ wf, T* __restrict__ wd, int O,
                                                               int I, int Op, int Ip, const float* const* vec_in,
                                                               float* __restrict__ vec_out) {
  __shared__ float tile[PK3_TO * PK3_PITCH];
  const int tiles_i = Ip / PK3_TI, ntiles = (Op / PK3_TO) * tiles_i;
  if ((int)blockIdx.x >= ntiles) {
    for (int i = threadIdx.x; i < 5 * Op; i += 256) {
      const int v = i / Op, c = i - v * Op;
      const float* src = vec_in[v];
      vec_out[i] = (c < O && src) ? src[c] : ((v == 1 || v == 4) ? 1.f : 0.f);
    }
    return;
  }
  const int o0 = ((int)blockIdx.x / tiles_i) * PK3_TO, c0 = ((int)blockIdx.x % tiles_i) * PK3_TI;
  // 16 bytes per load and per store: with an element per thread the kernel is bound by its instruction count (35 per element)
  constexpr int RUN4 = PK3_TI * 27 / 4, NLOAD = PK3_TO * RUN4, ROUND = 7;   // I % 16 == 0: a tile's runs are whole or padding
  const bool cols = c0 < I;
  for (int e0 = threadIdx.x; e0 < NLOAD; e0 += 256 * ROUND) {
    float4 v[ROUND];
#pragma unroll
    for (int k = 0; k < ROUND; ++k) {
      const int e = e0 + k * 256;
      const int o = e / RUN4, q = e - o * RUN4;
      v[k] = (e < NLOAD && cols && o0 + o < O) ? *reinterpret_cast<const float4*>(W + ((long long)(o0 + o) * I + c0) * 27 + q * 4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < ROUND; ++k) {
      const int e = e0 + k * 256;
      if (e < NLOAD) {
        const int o = e / RUN4, q = e - o * RUN4;
        float* t = tile + o * PK3_PITCH + q * 4;
        t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PK3_TO * 27 * (PK3_TI / 8); e += 256) {   // (o, tap, eight input channels)
    const int h = e % (PK3_TI / 8), ot = e / (PK3_TI / 8);
    const int tap = ot % 27, o = ot / 27;
    const float* t = tile + o * PK3_PITCH + h * 8 * 27 + tap;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t[j * 27];
    QtVec8<T>::store(wf + ((long long)(o0 + o) * 27 + tap) * Ip + c0 + h * 8, v);
  }
  if (wd)
    for (int e = threadIdx.x; e < PK3_TI * 27 * (PK3_TO / 8); e += 256) {   // (c, tap, eight output channels)
      const int oq = e % (PK3_TO / 8), ct = e / (PK3_TO / 8);
      const int tap = ct % 27, c = ct / 27;
      const float* t = tile + oq * 8 * PK3_PITCH + c * 27 + tap;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = t[j * PK3_PITCH];
      QtVec8<T>::store(wd + ((long long)(c0 + c) * 27 + tap) * Op + o0 + oq * 8, v);
    }
}

// weight gradient back to nn.Conv3d's layout: dw [27][Op][Ip]-like pieces -> dW [O][I][27]
//   FIRST == 0: dw = three [Op][9][Ip] f32 blocks (one per frame tap, as qt_conv2d_wgrad writes them);
//   FIRST == 1: dw = [Op][128] (K index tap * I + c).
__global__ void unpack_conv3d_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ dW, int O, int I, int Op, int Ip,
                                           int first) {
  const long long total = (long long)O * I * 27;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 27);
    const long long q = i / 27;
    const int c = (int)(q % I), o = (int)(q / I);
    float v;
    if (first) v = dw[(long long)o * 128 + tap * I + c];
    else {
      const int kt = tap / 9, t2 = tap - kt * 9;
      v = dw[(((long long)kt * Op + o) * 9 + t2) * Ip + c];
    }
    dW[i] = v;
  }
}

}  // namespace

extern "C" int qt_pack_conv3d_block(int dtype, const float* w, void* w_fwd, void* w_dgrad, int O, int I, int O_pad, int I_pad,
                                    int first, const float* const* vec_in_dev, float* vec_out, void* stream) {
  QT_CHECK_ARG(w && w_fwd && vec_in_dev && vec_out && O > 0 && I > 0 && O_pad >= O && (first || I_pad >= I) &&
                   (!first || 27 * I <= 128),
               "qt_pack_conv3d_block: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_conv3d_block: bad dtype %d", dtype);
  const long long total = (first ? (long long)O_pad * 128 : (long long)O_pad * 27 * I_pad) + 5ll * O_pad;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const bool tiled = [] {
    const char* e = getenv("QTCNN_PACK3D_TILED");
    return !(e && e[0] == '0');
  }();
  if (!first && tiled && O_pad % 64 == 0 && I_pad % 64 == 0 && I % PK3_TI == 0 && ((uintptr_t)w % 16) == 0) {
    const int blocks = (O_pad / PK3_TO) * (I_pad / PK3_TI) + 1;
    if (dtype == QT_F32)
      hipLaunchKernelGGL(pack_conv3d_tile_kernel<float>, dim3(blocks), dim3(256), 0, s, w, static_cast<float*>(w_fwd),
                         static_cast<float*>(w_dgrad), O, I, O_pad, I_pad, vec_in_dev, vec_out);
    else
      hipLaunchKernelGGL(pack_conv3d_tile_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, static_cast<bf16_t*>(w_fwd),
                         static_cast<bf16_t*>(w_dgrad), O, I, O_pad, I_pad, vec_in_dev, vec_out);
    QT_CHECK_LAUNCH();
    return QT_OK;
  }
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_conv3d_block_kernel<float>, dim3(grid), dim3(256), 0, s, w, static_cast<float*>(w_fwd),
                       static_cast<float*>(w_dgrad), O, I, O_pad, I_pad, first, vec_in_dev, vec_out);
  else
    hipLaunchKernelGGL(pack_conv3d_block_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, static_cast<bf16_t*>(w_fwd),
                       static_cast<bf16_t*>(w_dgrad), O, I, O_pad, I_pad, first, vec_in_dev, vec_out);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_unpack_conv3d_wgrad(const float* dw, float* grad, int O, int I, int O_pad, int I_pad, int first, void* stream) {
  QT_CHECK_ARG(dw && grad && O > 0 && I > 0 && O_pad >= O && (first || I_pad >= I), "qt_unpack_conv3d_wgrad: bad argument");
  const long long total = (long long)O * I * 27;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(unpack_conv3d_wgrad_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), dw, grad, O, I,
                     O_pad, I_pad, first);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pack_clip27(int dtype, const float* clips, void* dst, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(clips && dst && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_pack_clip27: bad argument");
  QT_CHECK_ARG((long long)frames * batch * h * w < (1ll << 31), "qt_pack_clip27: more than 2^31 pixel rows");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_clip27: bad dtype %d", dtype);
  const long long n = (long long)frames * batch * h * w * 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_clip27_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, clips, (float*)dst, batch, frames, h, w);
  else
    hipLaunchKernelGGL(pack_clip27_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, clips, (bf16_t*)dst, batch, frames, h, w);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// rows of partial sums qt_bn_stats writes for an [M][C] map (at most 1024 slabs of at least 256 rows)
extern "C" int qt_bn_stats_rows(long long M, int C) {
  (void)C;
  const long long slabs = (M + 255) / 256;
  return (int)(slabs > 1024 ? 1024 : (slabs < 1 ? 1 : slabs));
}

extern "C" int qt_bn_stats(int dtype, const void* y, long long M, int C, float* partial, void* stream) {
  QT_CHECK_ARG(y && partial && M > 0, "qt_bn_stats: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_bn_stats: bad dtype %d", dtype);
  QT_CHECK_ARG(C >= 8 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
               "qt_bn_stats: C=%d: C / 8 must divide 256", C);
  const int rows = qt_bn_stats_rows(M, C);
  const int slab = (int)((M + rows - 1) / rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)y, M, C, partial, slab);
  else
    hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)y, M, C, partial, slab);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pool3d_max(int dtype, const void* x, void* out, unsigned char* argmax, int frames, int batch, int h, int w,
                             int C, int pool_t, void* stream) {
  QT_CHECK_ARG(x && out && frames > 0 && batch > 0 && h >= 2 && w >= 2 && C % 8 == 0, "qt_pool3d_max: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pool3d_max: bad dtype %d", dtype);
  QT_CHECK_ARG((pool_t == 1 || pool_t == 2) && frames >= pool_t, "qt_pool3d_max: pool_t=%d (1 or 2, <= frames)", pool_t);
  const long long n = (long long)(frames / pool_t) * batch * (h / 2) * (w / 2) * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n)), blk(256);
#define QT_POOL(TT, PT) hipLaunchKernelGGL((pool3d_max_kernel<TT, PT>), grid, blk, 0, s, (const TT*)x, (TT*)out, argmax, frames, batch, h, w, C)
  if (dtype == QT_F32) { if (pool_t == 1) QT_POOL(float, 1); else QT_POOL(float, 2); }
  else { if (pool_t == 1) QT_POOL(bf16_t, 1); else QT_POOL(bf16_t, 2); }
#undef QT_POOL
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pool3d_max_bwd(int dtype, const void* dout, const unsigned char* argmax, void* dx, int frames, int batch,
                                 int h, int w, int C, int pool_t, void* stream) {
  QT_CHECK_ARG(dout && argmax && dx && frames > 0 && batch > 0 && h >= 2 && w >= 2 && C % 8 == 0, "qt_pool3d_max_bwd: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pool3d_max_bwd: bad dtype %d", dtype);
  QT_CHECK_ARG((pool_t == 1 || pool_t == 2) && frames >= pool_t, "qt_pool3d_max_bwd: pool_t=%d", pool_t);
  const long long n = (long long)frames * batch * h * w * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n)), blk(256);
#define QT_POOLB(TT, PT) hipLaunchKernelGGL((pool3d_max_bwd_kernel<TT, PT>), grid, blk, 0, s, (const TT*)dout, argmax, (TT*)dx, frames, batch, h, w, C)
  if (dtype == QT_F32) { if (pool_t == 1) QT_POOLB(float, 1); else QT_POOLB(float, 2); }
  else { if (pool_t == 1) QT_POOLB(bf16_t, 1); else QT_POOLB(bf16_t, 2); }
#undef QT_POOLB
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pool3d_bn_relu_max(int dtype, const void* y, const float* scale, const float* shift, void* out,
                                     unsigned char* argmax, void* y_at_max, int frames, int batch, int h, int w, int C,
                                     int y_channels, int pool_t, void* stream) {
  QT_CHECK_ARG(y && scale && shift && out && frames > 0 && batch > 0 && h >= 2 && w >= 2 && C % 8 == 0,
               "qt_pool3d_bn_relu_max: bad argument");
  QT_CHECK_ARG(y_channels > 0 && y_channels <= C && y_channels % 8 == 0, "qt_pool3d_bn_relu_max: y_channels=%d of C=%d", y_channels, C);
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pool3d_bn_relu_max: bad dtype %d", dtype);
  QT_CHECK_ARG((pool_t == 1 || pool_t == 2) && frames >= pool_t, "qt_pool3d_bn_relu_max: pool_t=%d", pool_t);
  const long long n = (long long)(frames / pool_t) * batch * (h / 2) * (w / 2) * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(n)), blk(256);
#define QT_POOLF(TT, PT) hipLaunchKernelGGL((pool3d_bn_relu_max_kernel<TT, PT>), grid, blk, 0, s, (const TT*)y, scale, shift, (TT*)out, argmax, (TT*)y_at_max, frames, batch, h, w, C, y_channels)
  if (dtype == QT_F32) { if (pool_t == 1) QT_POOLF(float, 1); else QT_POOLF(float, 2); }
  else { if (pool_t == 1) QT_POOLF(bf16_t, 1); else QT_POOLF(bf16_t, 2); }
#undef QT_POOLF
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// tests: the smallest launch (in 16-byte groups) that takes the resident-grid form of qt_pool3d_bn_bwd_apply (0 = default)
static std::atomic<long long> g_pool3d_light_min{1ll << 20};
extern "C" void qt_set_pool3d_apply_light_min(long long groups) { g_pool3d_light_min.store(groups > 0 ? groups : (1ll << 20)); }

extern "C" int qt_pool3d_bn_bwd_apply(int dtype, const void* dout, const unsigned char* argmax, const void* pooled, const void* y,
                                      const float* mean, const float* invstd, const float* coef, void* dy, int frames, int batch,
                                      int h, int w, int C, int y_channels, int dy_channels, int pool_t, void* stream) {
  QT_CHECK_ARG(dout && argmax && pooled && y && mean && invstd && coef && dy && frames > 0 && batch > 0 && h >= 2 && w >= 2 &&
                   C % 8 == 0,
               "qt_pool3d_bn_bwd_apply: bad argument");
  QT_CHECK_ARG(y_channels > 0 && y_channels <= C && y_channels % 8 == 0 && dy_channels >= y_channels && dy_channels % 8 == 0,
               "qt_pool3d_bn_bwd_apply: y_channels=%d dy_channels=%d of C=%d", y_channels, dy_channels, C);
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pool3d_bn_bwd_apply: bad dtype %d", dtype);
  QT_CHECK_ARG((pool_t == 1 || pool_t == 2) && frames >= pool_t, "qt_pool3d_bn_bwd_apply: pool_t=%d", pool_t);
  const long long n = (long long)frames * batch * h * w * (dy_channels / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const bool light = [] {
    const char* e = getenv("QTCNN_POOL3D_APPLY_LIGHT");
    return !(e && e[0] == '0');
  }();
  if (light && dtype == QT_BF16 && C == y_channels && C == dy_channels && 256 % (C / 8) == 0 && n >= g_pool3d_light_min.load()) {
    const dim3 lgrid(256 * 5), blk(256);   // (92 VGPRs: five waves per SIMD; a multiple of C / 8 threads: the channel group is loop invariant)
    if (pool_t == 1)
      hipLaunchKernelGGL((pool3d_bn_bwd_apply_light_kernel<bf16_t, 1>), lgrid, blk, 0, s, (const bf16_t*)dout, argmax,
                         (const bf16_t*)pooled, (const bf16_t*)y, mean, invstd, coef, (bf16_t*)dy, frames, batch, h, w, C);
    else
      hipLaunchKernelGGL((pool3d_bn_bwd_apply_light_kernel<bf16_t, 2>), lgrid, blk, 0, s, (const bf16_t*)dout, argmax,
                         (const bf16_t*)pooled, (const bf16_t*)y, mean, invstd, coef, (bf16_t*)dy, frames, batch, h, w, C);
    QT_CHECK_LAUNCH();
    return QT_OK;
  }
  const dim3 grid(grid_for(n)), blk(256);
#define QT_POOLA(TT, PT) hipLaunchKernelGGL((pool3d_bn_bwd_apply_kernel<TT, PT>), grid, blk, 0, s, (const TT*)dout, argmax, (const TT*)pooled, (const TT*)y, mean, invstd, coef, (TT*)dy, frames, batch, h, w, C, y_channels, dy_channels)
  if (dtype == QT_F32) { if (pool_t == 1) QT_POOLA(float, 1); else QT_POOLA(float, 2); }
  else { if (pool_t == 1) QT_POOLA(bf16_t, 1); else QT_POOLA(bf16_t, 2); }
#undef QT_POOLA
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_avgpool_tb(int dtype, const void* x, float* dst, int frames, int batch, int hw, int C, int ld, int col0,
                             void* stream) {
  QT_CHECK_ARG(x && dst && frames > 0 && batch > 0 && hw > 0 && C % 8 == 0 && ld >= col0 + C, "qt_avgpool_tb: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_avgpool_tb: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(avgpool_tb_kernel<float>, dim3(batch, (C / 8 + 31) / 32), dim3(256), 0, s, (const float*)x, dst, frames, batch, hw, C, ld, col0);
  else
    hipLaunchKernelGGL(avgpool_tb_kernel<bf16_t>, dim3(batch, (C / 8 + 31) / 32), dim3(256), 0, s, (const bf16_t*)x, dst, frames, batch, hw, C, ld, col0);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_avgpool_tb_bwd(int dtype, const float* d, void* g, int frames, int batch, int hw, int C, int ld, int col0,
                                 void* stream) {
  QT_CHECK_ARG(d && g && frames > 0 && batch > 0 && hw > 0 && C % 8 == 0 && ld >= col0 + C, "qt_avgpool_tb_bwd: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_avgpool_tb_bwd: bad dtype %d", dtype);
  const long long n = (long long)frames * batch * hw * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(avgpool_tb_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, d, (float*)g, frames, batch, hw, C, ld, col0);
  else
    hipLaunchKernelGGL(avgpool_tb_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, d, (bf16_t*)g, frames, batch, hw, C, ld, col0);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

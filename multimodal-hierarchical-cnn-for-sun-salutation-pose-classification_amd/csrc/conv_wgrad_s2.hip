// Weight gradient of the STRIDE-2 convolutions of a ResNet transition block (bf16): conv1 (3x3 / 2, pad 1) and the
// 1x1 / 2 downsample, tile-resident on the four PARITY PLANES of the block input.
//
// Replaces the conv2d backward-weight ATen calls of loss.backward() (/root/reference/Quadtree_from scratch/
// Quadtree_train.py:65) for layer{2,3,4}.0.conv1 and layer{2,3,4}.0.downsample.0 of the torchvision ResNet-18 the
// reference builds on (/root/reference/Quadtree_from scratch/models.py:221-229, :240-243).
//
//   dW[o][kh][kw][i] = sum_{b,oh,ow} dY[b][oh][ow][o] * X[b][2 oh + kh - 1][2 ow + kw - 1][i]
//
// Until round 3 these six launches ran on the generic kernel (conv_wgrad.hip): one workgroup per (tap, tile) that
// fetches dY and X once per tap (32 FLOP per staged byte) and ends in float atomics -- 7-8 % matrix-pipe busy,
// 0.66 ms of the 6 ms step for < 4 % of its FLOPs.
//
// Decomposition (conv_s2.hip's, turned around): plane (r, c) of X holds the pixels (2h + r, 2w + c).  Output pixel
// (oh, ow) meets tap (kh, kw) at plane (kh != 1, kw != 1), plane pixel (oh - (kh == 0), ow - (kw == 0)): over the
// ZERO-PADDED output grid (one pad row above and one pad column left of every image, images back to back, as in
// conv_wgrad_patch.hip) a tap is a plane and a constant row shift in {0, -1, -PW, -PW - 1} -- a stride-1 problem:
//     plane (1,1): taps (0,0) (0,2) (2,0) (2,2)     plane (0,1): taps (1,0) (1,2)
//     plane (1,0): taps (0,1) (2,1)                 plane (0,0): tap (1,1) -- and the whole 1x1 / 2 downsample.
// One workgroup owns a 64(o) x 64(i) x 9(taps) accumulator and streams its range of padded positions ONCE: a tile of
// 64 positions of dY and the same 64 (+ HL halo rows in front: shifts are never positive) of each plane are resident
// in LDS, double buffered, gathered by LDS-DMA through per-position offset tables (a plane is the table of plane (0,0)
// plus a scalar byte offset; pad positions and rows outside the tensor are zero-filled by the buffer range check):
// 288 FLOP per staged byte of dY, as the tile kernel of the stride-1 layers.  Pixel-major MFMA fragments by
// ds_read_b64_tr_b16 with the 32-byte-block XOR swizzle on the DMA source side (same fragment geometry as
// conv_wgrad_tile_kernel).  Two wave groups: group g contracts chunk g (32 positions) of every tile, the groups'
// partial filters are exchanged through LDS.  Partial filters of the position ranges go to a workspace and are summed
// in a fixed order by wgrad_partial_sum_kernel, which writes the gradient in the reference's OIHW layout:
// deterministic, no atomics, no zero fill.
#include <stdlib.h>

#include "qt_common.h"

// conv_wgrad_patch.hip: dw = sum over ranges of part[range][...] (fixed order).  layout 1: j = (n*9 + tap)*KC + c ->
// OIHW element (n*KC + c)*9 + tap;  2: written as is ([N][1][KC] is OIHW already)
int qt_wgrad_partial_sum_launch(const float* part, float* dw, size_t filt, int nsplit, int KC, int layout, hipStream_t stream);

namespace {

constexpr int T = 64;       // positions per tile: two 32-position chunks, one per wave group
constexpr int CH = 32;

struct WS2Args {
  const bf16_t* dy;
  const bf16_t* x;
  float* part;               // [nsplit][N][NTAP][KC]
  unsigned x_is2, dy_is2;    // bytes per image
  int x_rs, x_ps;            // elements between rows / pixels of the (full-resolution) block input
  int N, KC, OH, OW, PW, PP, B;
  int total, pps, nsplit, tiles, tilesC;
  int HL;                    // halo rows in front of a tile (multiple of 8, >= PW + 1)
  unsigned xrb, bufb, tab;   // bytes of one plane region / of a buffer; LDS byte address of the offset tables
  unsigned x_bytes, dy_bytes;
  FastDiv div_pp, div_pw;
  int tap_plane[9], tap_shift[9];
};

__device__ __forceinline__ QT_LDS_AS s16x4* lds_tr_ptr(unsigned lds_byte) { return (QT_LDS_AS s16x4*)(size_t)lds_byte; }

// NPL planes staged (4: conv1; 1: the downsample, plane (0,0) only), NTAP taps accumulated (9 / 1).
// NB tile buffers (2 or 3: NB - 1 tiles of look-ahead -- the kernel is bound by the latency of its LDS-DMA, a tile's MFMAs
// last 0.5 us); NU = LDS-DMA instructions EVERY wave issues per tile (its share of the plane units padded with transfers
// through a zero-record resource into a spare KiB, + one dY unit), so that the counted wait below is an immediate.
template <int NPL, int NTAP, int NB, int NU>
__global__ __launch_bounds__(512) void conv_wgrad_s2_kernel(WS2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (lds_addr_of(smem) != 0) return;  // no static LDS in this kernel; addresses below are absolute

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wq = wave & 3;

  // whole position ranges per XCD; the channel tiles of one range run back to back on it (they share dY / X in its L2)
  int split, tile;
  if ((a.nsplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / a.tiles) * 8 + xcd;
    tile = slot % a.tiles;
  } else {
    split = blockIdx.x % a.nsplit;
    tile = blockIdx.x / a.nsplit;
  }
  const int tc = tile % a.tilesC, tn = tile / a.tilesC;
  const int n0 = tn * 64, c0 = tc * 64;
  const int p0 = split * a.pps;
  if (p0 >= a.total) return;
  const int pend = min(a.total, p0 + a.pps);
  const int ntiles = (pend - p0 + T - 1) / T;

  // ---- offset tables: entry r < PP is padded position r of an image, entries PP .. PP+7 are the first positions of the
  // NEXT image (an 8-row DMA unit may straddle two images).  tx: pixel (2 (ph-1), 2 (pw-1)) of X; ty: (ph-1, pw-1) of dY.
  const int tabn = a.PP + 8;
  {
    unsigned* tx = reinterpret_cast<unsigned*>(smem + a.tab);
    unsigned* ty = tx + tabn;
    for (int r = tid; r < tabn; r += 512) {
      const int rr = r >= a.PP ? r - a.PP : r;
      const int ph = (int)fdiv((unsigned)rr, a.div_pw), pw = rr - ph * a.PW;
      const bool ok = ph >= 1 && pw >= 1;
      const unsigned nx = r >= a.PP ? a.x_is2 : 0u, ny = r >= a.PP ? a.dy_is2 : 0u;
      tx[r] = ok ? (unsigned)(2 * (ph - 1) * a.x_rs + 2 * (pw - 1) * a.x_ps) * 2u + nx : kOob;
      ty[r] = ok ? (unsigned)(((ph - 1) * a.OW + (pw - 1)) * a.N) * 2u + ny : kOob;
    }
  }

  // ---- DMA side: a unit = 8 rows = 1 KiB; lane = (row of the unit, 16-byte LDS slot); swizzle on the source ----
  const i32x4 rx = make_rsrc(a.x, a.x_bytes), ry = make_rsrc(a.dy, a.dy_bytes);
  const int drow = lane >> 3, dslot = lane & 7;
  const int dkey = (drow >> 1) & 3;
  const unsigned lane_c = (unsigned)((((dslot >> 1) ^ dkey) << 1) | (dslot & 1)) * 16u;
  const unsigned tab_x = a.tab + (unsigned)drow * 4u, tab_y = tab_x + (unsigned)tabn * 4u;
  const int xu = (T + a.HL) >> 3;          // units per plane region
  const int xunits = NPL * xu;             // X units per tile (this wave: units wave, wave + 8, ...)
  auto sgpr = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  // image index (+1) and position inside the image of padded position `pos` (>= -PP)
  auto split_pos = [&](int pos, unsigned& img1, unsigned& rem) {
    const unsigned g = (unsigned)(pos + a.PP);
    img1 = fdiv(g, a.div_pp);
    rem = g - img1 * (unsigned)a.PP;
  };
  const unsigned spare = a.tab + 2u * (unsigned)tabn * 4u;   // a KiB behind the tables: where the padding transfers go
  auto dma_tile = [&](int k, unsigned bufbase) {   // tile k -> the buffer at LDS byte `bufbase`: exactly NU instructions per wave
    const int q0 = p0 + k * T;
#pragma unroll
    for (int j = 0; j < NU - 1; ++j) {
      const int u = wave + 8 * j;
      const bool live = u < xunits;   // (uniform)
      const int uc = live ? u : 0;
      const int pl = NPL == 1 ? 0 : uc / xu, uu = NPL == 1 ? uc : uc - pl * xu;
      unsigned img1, rem;
      split_pos(q0 - a.HL + 8 * uu, img1, rem);
      const unsigned plane_off = (unsigned)((pl >> 1) * a.x_rs + (pl & 1) * a.x_ps) * 2u;
      const unsigned base = sgpr((img1 - 1u) * a.x_is2 + (unsigned)c0 * 2u + plane_off);
      const unsigned tv = *reinterpret_cast<const unsigned*>(smem + tab_x + sgpr(rem) * 4u);
      i32x4 r_ = rx;
      r_.z = (int)sgpr(live ? a.x_bytes : 0u);   // zero records: no memory traffic, zeros into the spare KiB
      blds16(r_, tv + base + lane_c, 0u, sgpr(live ? bufbase + (unsigned)pl * a.xrb + (unsigned)uu * 1024u : spare));
    }
    {   // dY: T / 8 = 8 units, one per wave
      unsigned img1, rem;
      split_pos(q0 + 8 * wave, img1, rem);
      const unsigned base = sgpr((img1 - 1u) * a.dy_is2 + (unsigned)n0 * 2u);
      const unsigned tv = *reinterpret_cast<const unsigned*>(smem + tab_y + sgpr(rem) * 4u);
      blds16(ry, tv + base + lane_c, 0u, sgpr(bufbase + (unsigned)NPL * a.xrb + (unsigned)wave * 1024u));
    }
  };

  // ---- MFMA side (fragment geometry of conv_wgrad_tile_kernel) ----
  const int li = lane & 15, lg = lane >> 4;
  const int qq = li >> 2, pp = li & 3;
  const int lrow = 4 * lg + qq;  // second transposing read: lrow + 16 (same swizzle key)
  unsigned a_base[4], b_base[NTAP];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    a_base[i] = (unsigned)NPL * a.xrb + (unsigned)(group * CH + lrow) * 128u + ((unsigned)(i ^ ((lrow >> 1) & 3)) << 5) + pp * 8;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int r = a.HL + a.tap_shift[t] + group * CH + lrow;   // >= 0: HL >= PW + 1
    b_base[t] = (unsigned)a.tap_plane[t] * a.xrb + (unsigned)r * 128u + ((unsigned)(wq ^ ((r >> 1) & 3)) << 5) + pp * 8;
  }
  auto frag = [&](unsigned addr) -> uint4 {   // one 16-byte MFMA operand: rows lrow and lrow + 16 of a chunk
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(addr + 2048));
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };

  f32x4 acc[4][NTAP];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // tables visible, then tiles 0 .. NB-2 into buffers 0 .. NB-2 (tiles past the range read rows nobody multiplies)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int t = 0; t < NB - 1; ++t) dma_tile(t, (unsigned)t * a.bufb);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NU * (NB - 2)) : "memory");   // tile 0 has landed

  unsigned wbuf = (unsigned)(NB - 1) * a.bufb;   // buffer the next DMA fills: (k + NB - 1) % NB
  int rbuf = 0;                                  // buffer of tile k: k % NB
  for (int k = 0; k < ntiles; ++k) {
    dma_tile(k + NB - 1, wbuf);   // lands behind the MFMAs of NB - 1 tiles
    wbuf = wbuf + a.bufb == (unsigned)NB * a.bufb ? 0u : wbuf + a.bufb;
    // positions of the tile at or behind pend belong to the next range (or lie past the tensor: zeros): a range is a whole
    // number of tiles except the last one, whose tail is past the tensor
    uint4 fa[4], fb[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag(a_base[i]);
    fb[0] = frag(b_base[0]);
    if (NTAP > 1) fb[1] = frag(b_base[NTAP > 1 ? 1 : 0]);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (t + 2 < NTAP) fb[(t + 2) % 3] = frag(b_base[t + 2 < NTAP ? t + 2 : 0]);   // X fragments two taps ahead
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                            __builtin_bit_cast(bf16x8, fb[t % 3]), acc[i][t], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    {  // fragment bases -> the next buffer
      rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
      const unsigned delta = rbuf == 0 ? 0u - (unsigned)(NB - 1) * a.bufb : a.bufb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_base[i] += delta;
        asm volatile("" : "+v"(a_base[i]));
      }
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        b_base[t] += delta;
        asm volatile("" : "+v"(b_base[t]));
      }
    }
    // tile k + 1 has landed -- vmcnt retires in issue order: at most the NU (NB - 2) instructions of the tiles behind it are
    // outstanding -- (this wave's share; the barrier makes it everybody's) and nobody reads this tile's buffer again
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NU * (NB - 2)) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the look-ahead tiles past the range: landed before the LDS changes hands)
  __syncthreads();

  // ---- group 1 hands o-blocks 0,1 to group 0 and takes o-blocks 2,3 from it ----
  {
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + (wq * 2 * NTAP) * 64 + lane;  // [wq][2 NTAP][lane] x 16 B
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) xch[(i * NTAP + t) * 64] = acc[i][t];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[i][t] += xch[(i * NTAP + t) * 64];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) xch[(i * NTAP + t) * 64] = acc[2 + i][t];
    }
    __syncthreads();
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[2 + i][t] += xch[(i * NTAP + t) * 64];
    }
  }

  // ---- partial filter of this range: lane holds o = 16*i + 4*lg + r, input channel 16*wq + li of every tap ----
  const int cc = c0 + wq * 16 + li;
  float* base = a.part + (long long)split * a.N * NTAP * a.KC;
  auto flush = [&](int i, const f32x4 (&v)[NTAP]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + i * 16 + lg * 4 + r;
      float* row = base + (long long)n * NTAP * a.KC + cc;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) row[t * a.KC] = v[t][r];
    }
  };
  if (group == 0) {
    flush(0, acc[0]);
    flush(1, acc[1]);
  } else {
    flush(2, acc[2]);
    flush(3, acc[3]);
  }
}

int g_wgrad_s2 = -1;   // QTCNN_WGRAD_S2 (default 1): 0 = the generic kernel (same-box A/B)
bool s2_enabled() {
  if (g_wgrad_s2 < 0) {
    const char* e = getenv("QTCNN_WGRAD_S2");
    g_wgrad_s2 = e ? atoi(e) : 1;
  }
  return g_wgrad_s2 != 0;
}

// QTCNN_S2_BUFFERS (default 3): tile buffers of the stride-2 weight-gradient kernel where the LDS holds them (2: round-4 first form)
int s2_buffers() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("QTCNN_S2_BUFFERS");
    v = e ? atoi(e) : 3;
  }
  return v;
}

void split_ranges(int total, int tiles, int* pps_out, int* nsplit_out) {
  int nsplit = 256 / tiles;
  if (nsplit < 1) nsplit = 1;
  int pps = qt_cdiv(total, nsplit);
  pps = qt_cdiv(pps, T) * T;       // whole tiles per range
  *pps_out = pps;
  *nsplit_out = qt_cdiv(total, pps);
}

}  // namespace

extern "C" void qt_set_wgrad_s2(int on) { g_wgrad_s2 = on < 0 ? 1 : on; }

// bf16, 3x3 / stride 2 / pad 1 or 1x1 / stride 2 / pad 0 on an even-sized map, whole 64-channel tiles
bool qt_wgrad_s2_eligible(const qt_conv_desc* d) {
  if (!s2_enabled() || d->dtype != QT_BF16 || d->stride != 2 || d->quad || d->kt > 1) return false;
  const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1, k1 = d->kh == 1 && d->kw == 1 && d->pad == 0;
  if (!k3 && !k1) return false;
  if ((d->in_h & 1) || (d->in_w & 1) || d->out_h * 2 != d->in_h || d->out_w * 2 != d->in_w) return false;
  if (d->n_out % 64 || d->k_per_tap % 64) return false;
  if (d->out_w < 4 || d->out_w > 120 || d->out_h < 4) return false;
  if ((long long)d->batch * d->src_img_stride * 2 >= (1ll << 30)) return false;                       // 32-bit byte offsets
  if ((long long)d->batch * d->out_h * d->out_w * d->n_out * 2 >= (1ll << 30)) return false;
  if ((long long)d->batch * (d->out_h + 1) * (d->out_w + 1) >= (1ll << 30)) return false;
  const int PW = d->out_w + 1, HL = (PW + 1 + 7) / 8 * 8;
  const unsigned npl = k3 ? 4u : 1u;
  const unsigned lds = 2u * ((npl * (unsigned)(T + HL) + (unsigned)T) * 128u) + 2u * (unsigned)((d->out_h + 1) * PW + 8) * 4u + 1024u;
  const int nu = qt_cdiv((int)npl * ((T + HL) / 8), 8) + 1;   // (the instantiated DMA counts per wave and tile: see the launcher)
  if (k3 ? (nu != 6 && nu != 7) : (nu != 2 && nu != 3)) return false;
  return lds <= 160u * 1024u && HL <= (d->out_h + 1) * PW;
}

size_t qt_wgrad_s2_workspace_bytes(const qt_conv_desc* d) {
  if (!qt_wgrad_s2_eligible(d)) return 0;
  const int total = d->batch * (d->out_h + 1) * (d->out_w + 1);
  int pps, nsplit;
  split_ranges(total, (d->n_out / 64) * (d->k_per_tap / 64), &pps, &nsplit);
  return (size_t)nsplit * d->n_out * d->kh * d->kw * d->k_per_tap * 4;
}

// grad_oihw [n_out][k_per_tap][kh][kw] f32 is WRITTEN (not accumulated)
int qt_wgrad_s2_launch(const qt_conv_desc* d, const void* dy, const void* x, float* grad_oihw, void* workspace,
                       size_t workspace_bytes, void* stream) {
  const bool k3 = d->kh == 3;
  WS2Args a;
  a.dy = static_cast<const bf16_t*>(dy);
  a.x = static_cast<const bf16_t*>(x);
  a.part = static_cast<float*>(workspace);
  a.N = d->n_out; a.KC = d->k_per_tap; a.OH = d->out_h; a.OW = d->out_w; a.B = d->batch;
  a.PW = a.OW + 1; a.PP = a.PW * (a.OH + 1);
  a.x_rs = d->src_row_stride; a.x_ps = d->src_pix_stride;
  a.x_is2 = (unsigned)(d->src_img_stride * 2);
  a.dy_is2 = (unsigned)((long long)a.OH * a.OW * a.N * 2);
  a.x_bytes = (unsigned)((long long)a.B * d->src_img_stride * 2);
  a.dy_bytes = (unsigned)((long long)a.B * a.OH * a.OW * a.N * 2);
  a.total = a.B * a.PP;
  a.HL = (a.PW + 1 + 7) / 8 * 8;
  a.div_pp = make_fastdiv((unsigned)a.PP);
  a.div_pw = make_fastdiv((unsigned)a.PW);
  a.tilesC = a.KC / 64;
  a.tiles = (a.N / 64) * a.tilesC;
  int real_split;
  split_ranges(a.total, a.tiles, &a.pps, &real_split);
  a.nsplit = real_split;
  if (a.nsplit >= 6 && (a.nsplit & 7)) a.nsplit = qt_cdiv(a.nsplit, 8) * 8;  // empty tail ranges exit at once
  const int ntap = k3 ? 9 : 1, npl = k3 ? 4 : 1;
  const size_t filt = (size_t)a.N * ntap * a.KC;
  if (!workspace || workspace_bytes < (size_t)real_split * filt * 4) {
    qt_set_error("qt_conv2d_wgrad_oihw: workspace of %zu bytes needed", (size_t)real_split * filt * 4);
    return QT_ERR_INVALID_ARG;
  }
  a.xrb = (unsigned)(T + a.HL) * 128u;
  a.bufb = (unsigned)npl * a.xrb + (unsigned)T * 128u;
  const unsigned tab_bytes = 2u * (unsigned)(a.PP + 8) * 4u + 1024u;   // offset tables + the spare KiB of the padding transfers
  const int nb = (3u * a.bufb + tab_bytes <= 160u * 1024u && s2_buffers() >= 3) ? 3 : 2;
  a.tab = (unsigned)nb * a.bufb;
  const int nu = qt_cdiv(npl * ((T + a.HL) / 8), 8) + 1;   // plane units per wave (rounded up) + its dY unit
  for (int t = 0; t < 9; ++t) {
    const int kh = k3 ? t / 3 : 1, kw = k3 ? t % 3 : 1;
    a.tap_plane[t] = k3 ? ((kh != 1) * 2 + (kw != 1)) : 0;
    a.tap_shift[t] = -((kh == 0) * a.PW + (kw == 0));
  }
  const unsigned lds = a.tab + tab_bytes;
  // (the exchange at the end lives in the buffers: 4 x 2 NTAP x 64 lanes x 16 B)
  if ((size_t)2 * a.bufb < (size_t)4 * 2 * ntap * 64 * 16) {
    qt_set_error("qt_conv2d_wgrad_oihw: map too small for the stride-2 kernel");
    return QT_ERR_UNSUPPORTED;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // instantiations: (planes, taps, buffers, DMA instructions per wave and tile); nu = 6 / 7 for HL = 16 / 32 with four planes,
  // 2 / 3 (one plane: 10 / 12 units over 8 waves) -- anything else takes the generic kernel
#define QT_S2_LAUNCH(NPL_, NTAP_, NB_, NU_)                                                                     \
  {                                                                                                             \
    auto kern = conv_wgrad_s2_kernel<NPL_, NTAP_, NB_, NU_>;                                                    \
    static std::atomic<unsigned long long> set_{0};                                                             \
    if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024, set_)) return rc;          \
    hipLaunchKernelGGL(kern, dim3(a.tiles * a.nsplit), dim3(512), lds, s, a);                                   \
    launched = true;                                                                                            \
  }
  bool launched = false;
  if (k3 && nb == 3 && nu == 6) QT_S2_LAUNCH(4, 9, 3, 6)
  else if (k3 && nb == 3 && nu == 7) QT_S2_LAUNCH(4, 9, 3, 7)
  else if (k3 && nu == 6) QT_S2_LAUNCH(4, 9, 2, 6)
  else if (k3 && nu == 7) QT_S2_LAUNCH(4, 9, 2, 7)
  else if (!k3 && nb == 3 && nu == 3) QT_S2_LAUNCH(1, 1, 3, 3)
  else if (!k3 && nb == 3 && nu == 2) QT_S2_LAUNCH(1, 1, 3, 2)
  else if (!k3 && nu == 3) QT_S2_LAUNCH(1, 1, 2, 3)
  else if (!k3 && nu == 2) QT_S2_LAUNCH(1, 1, 2, 2)
#undef QT_S2_LAUNCH
  if (!launched) {
    qt_set_error("qt_conv2d_wgrad_oihw: stride-2 kernel: no instantiation for this map width");
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_LAUNCH();
  return qt_wgrad_partial_sum_launch(a.part, grad_oihw, filt, real_split, a.KC, k3 ? 1 : 2, s);
}

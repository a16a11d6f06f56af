// The stride-2 TRANSITION of a ResNet stage in one launch: conv1 (3x3 / stride 2 / pad 1) of layerN.0 AND its 1x1 / stride 2
// downsample, both from ONE staged input patch, persistent workgroups, ping-pong MFMA schedule.
//
// Replaces the ATen calls behind torchvision's BasicBlock with a downsample (SURVEY.md A.1:
// `out = conv1(x)` and `identity = downsample[0](x)`), reached from
// /root/reference/Quadtree_from scratch/models.py:228-229,241 (layer2.0 / layer3.0 / layer4.0) and resnet/models.py:86-88,99.
// Until round 3 both ran on the generic implicit GEMM (conv_igemm.hip) at 0.078 of the MFMA peak: K loops of 9 / 1 taps,
// 6-7 vector instructions per MFMA, the input read twice.
//
// Decomposition.  A stride-2 tap reads input pixel (2*oh + kh - 1, 2*ow + kw - 1): inside ONE of the four PARITY PLANES
// (ih & 1, iw & 1) of the input that is plane pixel (oh + dr, ow + dc) with dr, dc in {-1, 0}: a constant shift.  The
// LDS-DMA takes a per-lane source address, so a plane is staged as its own position-numbered image -- LDS row
// (pr + 1) * PW + (pc + 1), PW = OW + 1: one zero pad column on the left, one pad / halo row on top -- and an output
// position m = r * PW + c reads LDS row m + (dr + 1) * PW + (dc + 1) of the tap's plane: exactly conv_pt.hip's "a tap is a
// row shift", with the nine taps spread over four planes:
//     plane (1,1): taps (0,0) (0,2) (2,0) (2,2)      plane (0,1): (1,0) (1,2)      plane (1,0): (0,1) (2,1)
//     plane (0,0): tap (1,1)  +  the 1x1 / stride-2 downsample (its only tap IS this plane at shift (0,0))
// A K-tile = one tap of one 128-byte channel chunk; ten K-tiles per chunk, the tenth feeds a second accumulator set with
// the downsample's weights.  Planes are the staging unit ("stages"): 256 LDS rows x 128 B = 32 KB each, a ring of
// three; while the K-tiles of one stage run, the passes of the stages 1-2 ahead are in flight (two 64-row passes per
// K-tile), so every input byte is staged ONCE for both convolutions (the generic kernels staged it 9 + 1 times).
//
// Tiles (196 output pixels = whole output rows, as conv_pt.hip):  28x28 outputs: a quarter image (7 rows, PW = 29),
// 14x14: one image (PW = 15), 7x7: four images stacked at pitch 8 / 64.  256 images -> 1024 / 256 / 64 pixel tiles.
// PERSISTENT: a workgroup walks a contiguous range of (pixel tile, channel tile) items; the K-tile stream never stops at
// an item boundary -- the next item's first plane and weight tiles are requested during the last K-tiles of the current
// one, the epilogue runs from the accumulators (no LDS) between two K-tiles.  DMA slots are branch-free: a slot with
// nothing to fetch (after the last item) goes through a zero-record resource (the hardware writes zeros into a dead
// buffer), so the counted vmcnt waits are compile-time constants.
//
// Schedule, barriers and hazards: conv_pt.hip's (two wave groups one raw barrier apart, counted vmcnt, weight ring of
// NBW slots).  Patch stages: stage B / C / D / A' of a chunk are issued in L_0-1 / L_2-3 / L_4-5 / L_6-7, always before the
// weight tile of the stage's first K-tile (so the wait for that tile covers them: vmcnt retires in issue order), into
// the buffer whose last reader finished at least one L segment earlier (ring of three: A -> a, B -> a+1, C -> a+2,
// D -> a, A' -> a+1).
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

namespace {

constexpr int kNT = 512;        // 8 waves: two per SIMD, one of each group
constexpr int kKB = 128;        // bytes of K per row and K-tile (one channel chunk)
constexpr int kNPB = 3;         // plane buffers
constexpr int kPBuf = 256 * kKB; // one plane stage: 4 passes of 64 rows
enum { GEO_ROWS = 0, GEO_STACK = 1 };

struct S2Out {
  void* dst;             // [B][OH][OW][N]
  const float* scale;    // per channel, nullable
  const float* shift;
  float* stats;          // [2 * tiles_m][2][N] per (pixel tile, wave row) sum / sum of squares of the raw value, nullable
  int relu;
};

struct S2Args {
  const void* src;
  const void* wgt;       // conv1 [N][3][3][KC]
  const void* wds;       // downsample [N][KC]
  S2Out out[2];          // 0: conv1, 1: downsample
  int N, KC;
  int OH, OW;
  int R, PW, npos, tpi;  // GEO_ROWS: output rows per tile, plane row pitch OW + 1, (R + 1) * PW, tiles per image
  int tiles_m, gridN, items, ipw, nt_fast;
  int nchunks;
  long long src_img_stride;
  int src_row_stride, src_pix_stride;
  unsigned src_records, wgt_bytes, wds_bytes;
  FastDiv div_pw;
};

// two / four transfers 8 KiB apart in LDS, M0 saved once (as conv_pt.hip)
__device__ __forceinline__ void s2_blds16x2(const i32x4& rsrc, unsigned v0, unsigned v1, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory", "scc");
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void s2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    s2_static_for<N, I + 1>(f);
  }
}

// K-tile t of a chunk: filter tap (kh * 3 + kw; 9 = the downsample), plane shift rows / columns, stage (plane) it reads
__device__ constexpr int kTapOf[10] = {0, 2, 6, 8, 3, 5, 1, 7, 4, 9};
__device__ constexpr int kShR[10] = {0, 0, 1, 1, 1, 1, 0, 1, 1, 1};
__device__ constexpr int kShC[10] = {0, 1, 0, 1, 0, 1, 1, 1, 1, 1};
// stage s = 0..3 holds plane (ph, pw) = (1,1), (0,1), (1,0), (0,0)
__device__ constexpr int kPh[4] = {1, 0, 1, 0};
__device__ constexpr int kPw[4] = {1, 1, 0, 0};

template <typename T, int BN, int NBW, int GEO>
__global__ __launch_bounds__(kNT, 2) void conv_s2_kernel(S2Args q) {
  static_assert(BN == 128, "channel tile");
  static_assert(NBW == 3 || NBW == 4, "weight ring slots");
  constexpr int TM = GEO == GEO_STACK ? 8 : 7;   // 16-row tiles per wave (two wave rows)
  constexpr int TN = BN / 4 / 16;                // 16-channel tiles per wave (4 channel quarters)
  constexpr int NP = TN / 2;
  constexpr int RW = BN / 64;                    // LDS-DMA instructions per wave and weight tile
  constexpr int D = NBW - 1;                     // weight tiles in flight
  constexpr int WSLOT = BN * kKB;
  constexpr int ESZ = (int)sizeof(T);
  static_assert(RW == 2 && NP == 1, "BN = 128");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = grp, wn = wave & 3;
  const int frow = lane & 15, fk = lane >> 4;

  const int item_begin = __builtin_amdgcn_readfirstlane((int)blockIdx.x * q.ipw);
  const int item_end = min(item_begin + q.ipw, q.items);
  if (item_begin >= item_end) return;   // (uniform, before any barrier)

  const unsigned smem_base = lds_addr_of(smem);
  const unsigned wring = smem_base + kNPB * kPBuf;
  // scale / shift of both outputs live in LDS behind the rings ([4][N] f32, identity where a vector is absent): an epilogue
  // in the middle of the K-tile stream must not issue global loads (their vmcnt(0) would drain the DMA pipeline)
  float* aff = reinterpret_cast<float*>(smem + kNPB * kPBuf + NBW * WSLOT);
  for (int i = tid; i < 4 * q.N; i += kNT) {
    const int v = i / q.N, c = i - v * q.N;
    const float* p = (v & 1) ? q.out[v >> 1].shift : q.out[v >> 1].scale;
    aff[i] = p ? p[c] : ((v & 1) ? 0.f : 1.f);
  }
  const int PW = GEO == GEO_STACK ? 8 : q.PW;

  // ---- per-thread staging rows (lane-invariant over tiles: the tile's origin travels in the scalar offset) ----
  const int rbase = tid >> 3;                         // row inside a 64-row pass
  const int chunk = (tid & 7) ^ (rbase & 7);          // source 16-byte chunk of this lane
  const int ce = chunk * (16 / ESZ);
  unsigned voff[4];
  bool top_lane = false;   // GEO_ROWS: LDS row 0 of a plane = the halo row above the tile (zero for the first tile of an image)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pos = i * 64 + rbase;
    voff[i] = kOob;
    if constexpr (GEO == GEO_ROWS) {
      const unsigned lr = fdiv((unsigned)pos, q.div_pw);
      const unsigned lc = (unsigned)pos - lr * (unsigned)q.PW;
      if (pos < q.npos && lc >= 1) {
        // plane pixel (row0 + lr - 1, lc - 1) -> input pixel (2 * (lr - 1) + ph, 2 * (lc - 1) + pw) relative to the tile's
        // first input row; the resource's base sits two rows above it, (ph, pw) travel in the scalar offset
        voff[i] = (unsigned)((long long)(2 * (int)lr) * q.src_row_stride + (long long)(2 * ((int)lc - 1)) * q.src_pix_stride + ce) * (unsigned)ESZ;
        if (lr == 0) top_lane = true;
      }
    } else {
      const int il = pos >> 6, lr = (pos >> 3) & 7, lc = pos & 7;
      if (lr >= 1 && lc >= 1)
        voff[i] = (unsigned)((long long)il * q.src_img_stride + (long long)(2 * (lr - 1)) * q.src_row_stride +
                             (long long)(2 * (lc - 1)) * q.src_pix_stride + ce) * (unsigned)ESZ;
    }
  }
  const unsigned voff0_top = top_lane ? kOob : voff[0];   // pass 0 for a tile at the top of its image
  // LDS weight row rho = 32*g + 16*i + x holds output channel 32*g + 8*(x>>2) + 4*i + (x&3) (conv_pt.hip): a lane's two
  // 16x16 tiles own eight consecutive channels of a pixel
  unsigned w_off[RW], wd_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int rho = rbase + 64 * i;
    const int x = rho & 15, ii = (rho >> 4) & 1;
    const int n = (rho & ~31) + 8 * (x >> 2) + 4 * ii + (x & 3);
    w_off[i] = (unsigned)(n * (9 * q.KC) + ce) * (unsigned)ESZ;
    wd_off[i] = (unsigned)(n * q.KC + ce) * (unsigned)ESZ;
  }
  const unsigned char* src_base = static_cast<const unsigned char*>(q.src);
  if constexpr (GEO == GEO_ROWS) src_base -= (long long)2 * q.src_row_stride * ESZ;
  const i32x4 rs_src = make_rsrc(src_base, q.src_records), rs_wgt = make_rsrc(q.wgt, q.wgt_bytes),
              rs_wds = make_rsrc(q.wds, q.wds_bytes);

  // ---- items: (pixel tile mt, channel tile nt) ----
  struct Item {
    int mt, nt;
    unsigned psoff;   // byte offset of the tile's origin in src
    unsigned wsoff, dsoff;   // byte offsets of the channel tile's first filter in wgt / wds
    int top;          // GEO_ROWS: the tile starts at image row 0 (halo row above it = padding)
    int live;         // 0: no such item (zero-record resources)
  };
  auto make_item = [&](int it) {
    Item r;
    r.live = it < item_end ? 1 : 0;
    const int i2 = r.live ? it : item_begin;
    if (q.nt_fast) { r.mt = i2 / q.gridN; r.nt = i2 - r.mt * q.gridN; }
    else { r.nt = i2 / q.tiles_m; r.mt = i2 - r.nt * q.tiles_m; }
    if constexpr (GEO == GEO_ROWS) {
      const int img = r.mt / q.tpi, part = r.mt - img * q.tpi;
      r.top = part == 0;
      r.psoff = (unsigned)(((long long)img * q.src_img_stride + (long long)(2 * part * q.R) * q.src_row_stride) * ESZ);
    } else {
      r.top = 0;
      r.psoff = (unsigned)((long long)r.mt * 4 * q.src_img_stride * ESZ);
    }
    r.wsoff = (unsigned)((long long)r.nt * BN * 9 * q.KC * ESZ);
    r.dsoff = (unsigned)((long long)r.nt * BN * q.KC * ESZ);
    if (!r.live) r.psoff = r.wsoff = r.dsoff = 0;   // (with zero records: every lane out of range whatever the check subtracts)
    return r;
  };
  auto zrec = [](i32x4 r, int live) { r.z = live ? r.z : 0; return r; };

  // ---- per-lane fragment addressing (conv_pt.hip) ----
  const int a_lane = (wm * (TM * 16) + frow) * kKB;
  int b_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) b_off[kk] = (wn * (BN / 4) + frow) * kKB + (((kk * 4 + fk) ^ (frow & 7)) << 4);

  f32x4 acc[2][TN][TM];   // [0] conv1, [1] downsample

  const unsigned plane_delta[4] = {
      (unsigned)((kPh[0] * q.src_row_stride + kPw[0] * q.src_pix_stride) * ESZ),
      (unsigned)((kPh[1] * q.src_row_stride + kPw[1] * q.src_pix_stride) * ESZ),
      (unsigned)((kPh[2] * q.src_row_stride + kPw[2] * q.src_pix_stride) * ESZ),
      (unsigned)((kPh[3] * q.src_row_stride + kPw[3] * q.src_pix_stride) * ESZ)};
  const unsigned tap_bytes = (unsigned)(q.KC * ESZ);

  Item cur = make_item(item_begin);
  int pa = 0;               // plane buffer of stage A of the current chunk (ring of three)
  int rd = 0, wr = D % NBW; // weight ring slots
  // ---- prologue: stage A of the first chunk, D weight tiles ----
  {
    const unsigned v0 = cur.top ? voff0_top : voff[0];
    const unsigned so = cur.psoff + plane_delta[0];
    s2_blds16x2(rs_src, v0, voff[1], so, smem_base + wave * 1024);
    s2_blds16x2(rs_src, voff[2], voff[3], so, smem_base + 2 * 8192 + wave * 1024);
#pragma unroll
    for (int s = 0; s < D; ++s)   // K-tiles 0 .. D-1 of chunk 0 (D <= 3: conv taps)
      s2_blds16x2(rs_wgt, w_off[0], w_off[1], cur.wsoff + (unsigned)kTapOf[s] * tap_bytes, wring + s * WSLOT + wave * 1024);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(RW * (D - 1)) : "memory");   // (lgkmcnt: the affine vectors)
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
  }

  T* __restrict__ dst0 = static_cast<T*>(q.out[0].dst);
  T* __restrict__ dst1 = static_cast<T*>(q.out[1].dst);

  for (int it = item_begin; it < item_end; ++it) {
    const Item nxt_item = make_item(it + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[s][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cidx = 0; cidx < q.nchunks; ++cidx) {
      // (item, chunk) that follows this chunk in the K-tile stream
      const bool lastc = cidx + 1 == q.nchunks;
      unsigned n_psoff = lastc ? nxt_item.psoff : cur.psoff + (unsigned)(cidx + 1) * kKB;
      unsigned n_wsoff = lastc ? nxt_item.wsoff : cur.wsoff + (unsigned)(cidx + 1) * kKB;
      unsigned n_dsoff = lastc ? nxt_item.dsoff : cur.dsoff + (unsigned)(cidx + 1) * kKB;
      const int n_live = lastc ? nxt_item.live : 1;
      const int n_top = lastc ? nxt_item.top : cur.top;
      unsigned c_psoff = cur.psoff + (unsigned)cidx * kKB;
      unsigned c_wsoff = cur.wsoff + (unsigned)cidx * kKB;
      unsigned c_dsoff = cur.dsoff + (unsigned)cidx * kKB;
      asm volatile("" : "+s"(n_psoff), "+s"(n_wsoff), "+s"(n_dsoff), "+s"(c_psoff), "+s"(c_wsoff), "+s"(c_dsoff));
      const i32x4 rs_src_n = zrec(rs_src, n_live), rs_wgt_n = zrec(rs_wgt, n_live), rs_wds_n = zrec(rs_wds, n_live);
      const unsigned v0_c = cur.top ? voff0_top : voff[0];
      const unsigned v0_n = n_top ? voff0_top : voff[0];
      const int pb1 = pa + 1 >= kNPB ? pa + 1 - kNPB : pa + 1, pb2 = pa + 2 >= kNPB ? pa + 2 - kNPB : pa + 2;
      const bool after_epilogue = D == 2 && cidx == 0 && it != item_begin;

      s2_static_for<10>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        // ---- L_t: two passes of the stage 1-2 ahead, the weight tile D ahead, this K-tile's fragments ----
        if constexpr (t < 8) {
          constexpr int s = t / 2 + 1;          // 1, 2, 3: planes B, C, D of this chunk; 4: plane A of the next chunk / item
          constexpr int pp = (t & 1) * 2;
          const int buf = (s == 1 || s == 4) ? pb1 : (s == 2 ? pb2 : pa);
          const unsigned lds = smem_base + buf * kPBuf + pp * 8192 + wave * 1024;
          if constexpr (s < 4) {
            unsigned so = c_psoff + plane_delta[s];
            s2_blds16x2(rs_src, pp == 0 ? v0_c : voff[2], pp == 0 ? voff[1] : voff[3], so, lds);
          } else {
            unsigned so = n_psoff + plane_delta[0];
            s2_blds16x2(rs_src_n, pp == 0 ? v0_n : voff[2], pp == 0 ? voff[1] : voff[3], so, lds);
          }
        }
        {
          constexpr int u = (t + D) % 10;
          constexpr bool wrap = t + D >= 10;
          const unsigned sw = wring + wr * WSLOT + wave * 1024;
          if constexpr (u == 9) {
            s2_blds16x2(wrap ? rs_wds_n : rs_wds, wd_off[0], wd_off[1], wrap ? n_dsoff : c_dsoff, sw);
          } else {
            unsigned so = (wrap ? n_wsoff : c_wsoff) + (unsigned)kTapOf[u] * tap_bytes;
            s2_blds16x2(wrap ? rs_wgt_n : rs_wgt, w_off[0], w_off[1], so, sw);
          }
          wr = wr + 1 == NBW ? 0 : wr + 1;
        }
        const int rbuf = t < 4 ? pa : (t < 6 ? pb1 : (t < 8 ? pb2 : pa));
        int sh = (kShR[t] * PW + kShC[t]) * kKB;
        // (opaque to the optimiser: the fragment addresses must not be hoisted out of the chunk loop, conv_pt.hip)
        asm volatile("" : "+s"(sh));
        const unsigned char* pap = smem + rbuf * kPBuf + sh;
        const unsigned char* pw = smem + kNPB * kPBuf + rd * WSLOT;
        rd = rd + 1 == NBW ? 0 : rd + 1;
        uint4 fw[2][TN], fa[2][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          fw[0][i] = *reinterpret_cast<const uint4*>(pw + i * (16 * kKB) + b_off[0]);
          fw[1][i] = *reinterpret_cast<const uint4*>(pw + i * (16 * kKB) + b_off[1]);
        }
        {
          const int rowb = a_lane + sh;
          const int a0 = a_lane + (((fk ^ (rowb >> 7)) & 7) << 4);
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            fa[0][j] = *reinterpret_cast<const uint4*>(pap + a0 + j * (16 * kKB));
            fa[1][j] = *reinterpret_cast<const uint4*>(pap + (a0 ^ 64) + j * (16 * kKB));
          }
        }
        // weight tile t+1 (issued D-1 segments ago) and everything older -- the plane passes of the stage that K-tile
        // t+1 may open included -- have landed once only the instructions issued after it are outstanding
        constexpr int in_t = (t < 8 ? 2 : 0) + RW;
        constexpr int in_tm1 = ((t + 9) % 10 < 8 ? 2 : 0) + RW;
        constexpr int allowed = in_t + (D == 3 ? in_tm1 : 0);
        if (t == 0 && after_epilogue) {   // (uniform; see the epilogue)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(allowed) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- C_t ----
        __builtin_amdgcn_s_setprio(1);
        constexpr int S = t == 9 ? 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[S][i][j], fw[kk][i], fa[kk][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
      });
      pa = pb1;
    }

    // ---- epilogue of the item, straight from the accumulators (no LDS traffic but the affine vectors: the next item's DMA
    // is in flight).  Its stores count in vmcnt, in issue order, between the DMA of the last and the next K-tiles: everything
    // the next item's first K-tile needs (its plane A, weight tiles 0 and 1: all issued at least a segment ago) is waited for
    // HERE, so that the first L segment behind the stores needs no vmcnt wait at all and nobody waits for a store to retire
    // before the second one (by then they have) ----
    // Item boundary of the ping-pong.  Group 1 runs one barrier behind group 0: without the next two lines its last barrier of
    // the item pairs with group 0's FIRST barrier of the next item (or the one behind the loop), i.e. group 1 sits out group 0's
    // whole epilogue and group 0 then waits for group 1's -- the two epilogues ran one after the other (measured on conv_pt, scripts/pt_phases.py).  Group 0 takes one extra
    // barrier BEFORE its epilogue (pairs with group 1's last), so both epilogues run side by side; group 1 takes one extra
    // barrier AFTER its epilogue when another item follows, which restores the one-barrier offset exactly as at the start.
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      int drow[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        int m = wm * (TM * 16) + j * 16 + frow;
        asm volatile("" : "+v"(m));   // (opaque: the row / column split must not be hoisted out of the item loop into spilled registers)
        drow[j] = -1;
        if constexpr (GEO == GEO_ROWS) {
          const unsigned r = fdiv((unsigned)m, q.div_pw);
          const unsigned c = (unsigned)m - r * (unsigned)q.PW;
          const int img = cur.mt / q.tpi, part = cur.mt - img * q.tpi;
          if ((int)r < q.R && (int)c < q.OW) drow[j] = (img * q.OH + part * q.R + (int)r) * q.OW + (int)c;
        } else {
          const int r = (m >> 3) & 7, c = m & 7, img = cur.mt * 4 + (m >> 6);
          if (r < 7 && c < 7) drow[j] = (img * 7 + r) * 7 + c;
        }
      }
      const int cl = wn * (BN / 4) + fk * 8;
      const int c0 = cur.nt * BN + cl;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const S2Out& o = q.out[s];
        T* __restrict__ dst = s == 0 ? dst0 : dst1;
        float sc[8], sf[8], s1[8], s2[8];
        if (o.scale || o.shift) {   // (uniform)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sc[e] = aff[(2 * s) * q.N + c0 + e];
            sf[e] = aff[(2 * s + 1) * q.N + c0 + e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          float v[8] = {acc[s][0][j][0], acc[s][0][j][1], acc[s][0][j][2], acc[s][0][j][3],
                        acc[s][1][j][0], acc[s][1][j][1], acc[s][1][j][2], acc[s][1][j][3]};
          if (drow[j] < 0) continue;
          if (o.stats != nullptr) {   // (uniform)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1[e] += v[e];
              s2[e] += v[e] * v[e];
            }
          }
          if (o.scale || o.shift) {   // (uniform)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sf[e];
          }
          if (o.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          QtVec8<T>::store(dst + (long long)drow[j] * q.N + c0, v);
        }
        if (o.stats != nullptr) {
          // sum over the 16 pixels (lanes with equal fk) of the wave, four DPP adds in a fixed order; one partial row per
          // (pixel tile, wave row): no cross-wave reduction, no LDS
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] = qt_row16_sum(s1[e]);
            s2[e] = qt_row16_sum(s2[e]);
          }
          if (frow == 0) {
            float* o0 = o.stats + ((long long)(cur.mt * 2 + wm) * 2) * q.N + c0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              o0[e] = s1[e];
              o0[q.N + e] = s2[e];
            }
          }
        }
      }
    }
    if (grp == 1 && it + 1 < item_end) asm volatile("s_barrier" ::: "memory");   // (see the item boundary above)
    cur = nxt_item;
  }
  // the branch-free slots of the last K-tiles wrote zeros into dead buffers: they must have landed before the LDS is
  // handed to another workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- host side --------------------------------------------------------------------------------------------

int g_s2_enabled = -1, g_s2_max_wgs = 0;
inline int s2_enabled() {
  if (g_s2_enabled < 0) {
    const char* e = getenv("QTCNN_S2_CONV");
    g_s2_enabled = e ? atoi(e) : 1;
  }
  return g_s2_enabled;
}

bool s2_geometry(const qt_conv_s2_desc* d, S2Args& q) {
  const int OH = d->in_h / 2;
  if (d->in_h % 2 != 0 || d->in_h != d->in_w) return false;
  q.OH = q.OW = OH;
  if (OH == 28) { q.R = 7; q.tpi = 4; }
  else if (OH == 14) { q.R = 14; q.tpi = 1; }
  else if (OH == 7) { q.R = 7; q.tpi = 0; }
  else return false;
  q.PW = OH == 7 ? 8 : OH + 1;
  q.npos = (q.R + 1) * q.PW;
  q.div_pw = make_fastdiv((unsigned)q.PW);
  return OH == 7 || q.npos <= 256;
}

template <typename T, int GEO>
int s2_launch(S2Args q, hipStream_t stream) {
  constexpr int NBW = 3, BN = 128;
  constexpr int lds = kNPB * kPBuf + NBW * BN * kKB + 4 * 512 * 4;   // + scale / shift of both outputs (c_out <= 512)
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv_s2_kernel<T, BN, NBW, GEO>;
  static std::atomic<unsigned long long> lds_limit_set{0};
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), lds, lds_limit_set)) return rc;
  const int grid = qt_cdiv(q.items, q.ipw);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kNT), lds, stream, q);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

}  // namespace

extern "C" void qt_set_conv_s2(int mode) { g_s2_enabled = mode < 0 ? 1 : mode; }
extern "C" void qt_set_conv_s2_max_workgroups(int n) { g_s2_max_wgs = n > 0 ? n : 0; }

extern "C" int qt_conv_s2_pair_supported(const qt_conv_s2_desc* d) {
  if (!d || !s2_enabled()) return 0;
  if (d->dtype != QT_F32 && d->dtype != QT_BF16) return 0;
  const int esz = d->dtype == QT_F32 ? 4 : 2;
  S2Args q;
  if (!s2_geometry(d, q)) return 0;
  if ((d->c_in * esz) % kKB != 0 || d->c_out % 128 != 0 || d->c_out > 512) return 0;
  if (d->batch < 16 || (q.OH == 7 && d->batch % 4 != 0)) return 0;   // (a handful of tiles: the generic kernel's small tiles)
  const long long in_bytes = (long long)d->batch * d->in_h * d->in_w * d->c_in * esz;
  if (in_bytes >= (1ll << 31) || (long long)d->c_out * 9 * d->c_in * esz >= (1ll << 31)) return 0;
  return 1;
}

extern "C" int qt_conv_s2_pair_stats_rows(const qt_conv_s2_desc* d) {
  S2Args q;
  if (!d || !s2_geometry(d, q)) return QT_ERR_INVALID_ARG;
  return 2 * (q.OH == 7 ? d->batch / 4 : d->batch * q.tpi);
}

extern "C" int qt_conv_s2_pair(const qt_conv_s2_desc* d, const qt_conv_s2_io* io, void* stream) {
  QT_CHECK_ARG(d && io, "qt_conv_s2_pair: null descriptor");
  QT_CHECK_ARG(qt_conv_s2_pair_supported(d), "qt_conv_s2_pair: unsupported problem (batch %d, %dx%dx%d -> %d)", d->batch,
               d->in_h, d->in_w, d->c_in, d->c_out);
  QT_CHECK_ARG(io->src && io->w_conv && io->w_down && io->y_conv && io->y_down, "qt_conv_s2_pair: null src / weights / outputs");
  for (const void* ptr : {io->src, io->w_conv, io->w_down, (const void*)io->y_conv, (const void*)io->y_down})
    QT_CHECK_ARG(((uintptr_t)ptr % 16) == 0, "qt_conv_s2_pair: pointers must be 16-byte aligned");
  const int esz = d->dtype == QT_F32 ? 4 : 2;
  S2Args q;
  s2_geometry(d, q);
  q.src = io->src; q.wgt = io->w_conv; q.wds = io->w_down;
  q.out[0] = {io->y_conv, io->scale_conv, io->shift_conv, io->stats_conv, d->relu_conv};
  q.out[1] = {io->y_down, io->scale_down, io->shift_down, io->stats_down, d->relu_down};
  q.N = d->c_out; q.KC = d->c_in;
  q.src_pix_stride = d->c_in; q.src_row_stride = d->in_w * d->c_in; q.src_img_stride = (long long)d->in_h * d->in_w * d->c_in;
  q.tiles_m = q.OH == 7 ? d->batch / 4 : d->batch * q.tpi;
  q.gridN = d->c_out / 128;
  q.items = q.tiles_m * q.gridN;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int wgs = g_s2_max_wgs > 0 ? g_s2_max_wgs : cus;
  q.ipw = qt_cdiv(q.items, wgs);
  // a workgroup's items: whole pixel tiles with all their channel tiles when they fit (the patch is re-read from L2),
  // else one channel tile of pixel tile (id % tiles_m): the channel tiles of a pixel tile then share id % 8 = an XCD
  q.nt_fast = q.ipw % q.gridN == 0 ? 1 : 0;
  q.nchunks = d->c_in * esz / kKB;
  // the range check of a raw buffer covers vector + scalar offset against num_records: the records span the whole tensor
  // (+ the two rows the base of the row geometry sits above it); every position that is not padding is inside by construction
  q.src_records = (unsigned)((long long)d->batch * q.src_img_stride * esz + (q.OH == 7 ? 0 : 2ll * q.src_row_stride * esz));
  q.wgt_bytes = (unsigned)((long long)d->c_out * 9 * d->c_in * esz);
  q.wds_bytes = (unsigned)((long long)d->c_out * d->c_in * esz);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == QT_F32) return q.OH == 7 ? s2_launch<float, GEO_STACK>(q, s) : s2_launch<float, GEO_ROWS>(q, s);
  return q.OH == 7 ? s2_launch<bf16_t, GEO_STACK>(q, s) : s2_launch<bf16_t, GEO_ROWS>(q, s);
}

// Convolutions of the CNN turbulence closure on the matrix cores (gfx950 MFMA, exact fp32: v_mfma_f32_16x16x4_f32).
//
// The closure of the reference (diffpiso/networks.py:3-73) is a 7-layer fully convolutional network, 4 -> 16 -> 16 -> 32 -> 64 -> 64
// -> 64 -> 2 channels, kernels 7,5,5,3,3,1,1, stride 1, no bias, leaky ReLU (0.2) after all but the last layer, NHWC tensors,
// HWIO weights, 'SAME' or 'VALID' padding (tf.nn.conv2d = cross-correlation).  Every layer is an implicit GEMM
//     out[pixel][co] = sum over (ky, kx, ci) in[pixel + (ky, kx) - pad][ci] * w[ky][kx][ci][co]
// with M = pixels, N = co, K = taps * ci.  One wavefront owns 64 consecutive pixels of one output row x all output channels
// (up to 4 x 4 tiles of 16 x 16 accumulators = 64 VGPRs); per K-step of 4 input channels of one tap it loads the A fragments
// (lane l: pixel l & 15, channel l >> 4 - 16-byte channel groups of NHWC, served by L1 / L2: a tap re-reads the row band its
// neighbours just touched) and the B fragments (lane l: channel l >> 4, output channel l & 15 - the weights of a layer are at
// most 147 KB and stay in L2) and issues MT x NT MFMAs.  fp32 MFMA is an exact fmaf chain, so results agree with a plain fp32
// convolution to summation order.
//   conv_forward   out = [leaky](conv(in, w)): the forward pass AND the input gradient (the caller passes the gradient of the
//                  layer's pre-activation output as `in` and the flipped, transposed weights; pad' = k - 1 - pad)
//   conv_wgrad     dW[ky][kx][ci][co] = sum over pixels in[pixel + (ky, kx) - pad][ci] * g[pixel][co]: M = ci,
//                  N = co, K = pixels; every workgroup reduces a band of rows into its own partial, a second kernel adds the
//                  partials in a fixed order (deterministic, no atomics)
#include "piso_common.h"
#include "options.h"

namespace piso {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kLeakySlope = 0.2f;

struct ConvGeom {
  int H, W;          // input rows / columns
  int Ho, Wo;        // output rows / columns
  int pad;           // zero padding on every side
  int cin, cout;     // true channel counts of `in` / `out` (the weight tensor is [KS][KS][CINP][COUTP], zero padded)
};

// KS: kernel size; CINP: input channels rounded up to 4 (<= 4 channels) or to 16; NT: output-channel tiles of 16 (COUTP = 16 NT).
// CINP >= 16: the K dimension of a block of 16 channels is PERMUTED so that every operand is one 16-byte load: K-step j of the
// block takes channel 4 (lane >> 4) + j from lane group lane >> 4 - a lane loads the float4 of its pixel's channels
// [4 (lane >> 4), +4) once and feeds component j to step j; the host lays the weights out to match:
//     w[tap][block][lane >> 4][co][j] = W[tap][16 block + 4 (lane >> 4) + j][co]          (piso_conv2d_weight_layout)
template <int KS, int CINP, int NT, bool LEAKY_OUT>
__global__ __launch_bounds__(kBlock) void conv_forward_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ w,
                                                               float* __restrict__ out) {
  constexpr int MT = 4;                                     // 4 x 16 = 64 pixels per wave
  constexpr int COUTP = 16 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (g.Wo + 16 * MT - 1) / (16 * MT);
  const int tile = blockIdx.x * (kBlock / 64) + wave;
  if (tile >= tiles_x * g.Ho) return;
  const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * 16 * MT;
  const int ai = lane & 15, ak = lane >> 4;                 // A: pixel in tile, channel group;  B: channel group = ak, co = ai
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (CINP >= 16) {
    // Software pipeline over the K-blocks (tap row, tap column, block of 16 channels): the operands of block s + 1 are loaded while
    // the 16 MT NT / 4 MFMAs of block s run - issued and consumed in the same block the loop ran at the latency of one L2 round
    // trip per block (forward 3 x 3, 64 -> 64: 264 us at 256 x 896, 40 % of the fp32 MFMA peak).
    constexpr int CB = CINP / 16, NSEQ = KS * CB;
    auto load_ab = [&](int yy, int sidx, f32x4 (&a)[MT], f32x4 (&b)[NT], int ky) __attribute__((always_inline)) {
      const int kx = sidx / CB, cb = sidx - kx * CB;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int xx = x0 + 16 * m + ai + kx - g.pad;
        a[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (xx >= 0 && xx < g.W) a[m] = *reinterpret_cast<const f32x4*>(in + ((size_t)yy * g.W + xx) * g.cin + 16 * cb + 4 * ak);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n)
        b[n] = *reinterpret_cast<const f32x4*>(w + ((((size_t)(ky * KS + kx) * CB + cb) * 4 + ak) * COUTP + 16 * n + ai) * 4);
    };
    for (int ky = 0; ky < KS; ++ky) {
      const int yy = y + ky - g.pad;
      if (yy < 0 || yy >= g.H) continue;                     // (wave-uniform: a whole tap row of zero padding)
      f32x4 a0[MT], b0[NT], a1[MT], b1[NT];
      load_ab(yy, 0, a0, b0, ky);
#pragma unroll
      for (int sq = 0; sq < NSEQ; sq += 2) {
        if (sq + 1 < NSEQ) load_ab(yy, sq + 1, a1, b1, ky);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][j], b0[n][j], acc[m][n], 0, 0, 0);
        if (sq + 1 < NSEQ) {
          if (sq + 2 < NSEQ) load_ab(yy, sq + 2, a0, b0, ky);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m][j], b1[n][j], acc[m][n], 0, 0, 0);
        }
      }
    }
  } else {
  for (int ky = 0; ky < KS; ++ky) {
    const int yy = y + ky - g.pad;
    if (yy < 0 || yy >= g.H) continue;                       // (wave-uniform: a whole tap row of zero padding)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      {
        static_assert(CINP == 4, "up to 4 input channels: one K-step per tap");
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int xx = x0 + 16 * m + ai + kx - g.pad;
          a[m] = (xx >= 0 && xx < g.W && ak < g.cin) ? in[((size_t)yy * g.W + xx) * g.cin + ak] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = w[((size_t)(ky * KS + kx) * 4 + ak) * COUTP + 16 * n + ai];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  }
  // C/D layout: column (co) = lane & 15, row (pixel) = (lane >> 4) * 4 + register
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = 16 * n + ai;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + 16 * m + ak * 4 + r;
        if (x < g.Wo && co < g.cout) {
          float v = acc[m][n][r];
          if (LEAKY_OUT) v = v > 0.f ? v : kLeakySlope * v;
          out[((size_t)y * g.Wo + x) * g.cout + co] = v;
        }
      }
    }
}

// The same convolution with its operands STAGED THROUGH LDS (CINP >= 16, KS >= 3).  The kernel above reads, per K-block of a wave,
// 4 KB of A and NT KB of B from L2 for 16 MT NT MFMAs: at config 4's size that is ~10 TB/s of L2 traffic chip-wide - the 64 -> 64
// layers ran at 53 % of the fp32 MFMA peak, bound by it.  Here the four waves of a workgroup (four consecutive tiles of 64 pixels,
// possibly of two output rows) walk the same stages = (tap row ky, block of 16 input channels) in lock step:
//   * B of the stage - the weights of all KS tap columns, KS NT KB - is loaded ONCE per workgroup and shared by the four waves;
//   * A of the stage - the 64 + KS - 1 input pixels a wave's tile touches over the KS tap columns, 16 channels - is loaded ONCE per
//     wave; the tap columns read it at pixel offsets 0 .. KS - 1 (a lane's 16-byte reads cover a contiguous KB: conflict-free).
// L2 traffic per stage and wave: (64 + KS - 1) 64 B + KS NT KB / 4 instead of KS (4 + NT) KB (3 x 3, 64 -> 64: 7.2 instead of 24 KB).
// Double-buffered: the next stage's operands travel from L2 into registers while the MFMAs of this stage run, are written to the
// other LDS buffer behind them, one barrier per stage.  Same K order per output as the kernel above: the same bits.
template <int KS, int CINP, int NT, bool LEAKY_OUT>
__global__ __launch_bounds__(kBlock) void conv_forward_lds_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ w,
                                                                   float* __restrict__ out) {
  static_assert(CINP >= 16 && KS >= 3, "tap columns share the staged pixels; channels in blocks of 16");
  constexpr int MT = 4, COUTP = 16 * NT, CB = CINP / 16;
  constexpr int P = 16 * MT + KS - 1;                       // pixels of a wave's A segment
  constexpr int NA = (P * 4 + 63) / 64;                     // 16-byte loads per lane for it
  constexpr int BV = KS * NT * 64;                          // 16-byte words of a stage's B
  constexpr int NB = (BV + kBlock - 1) / kBlock;            // ... per thread
  __shared__ f32x4 As[2][kBlock / 64][P * 4];
  __shared__ f32x4 Bs[2][BV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (g.Wo + 16 * MT - 1) / (16 * MT);
  const int tile = blockIdx.x * (kBlock / 64) + wave;
  const bool active = tile < tiles_x * g.Ho;                // (a wave without a tile still helps with B and takes part in the barriers)
  const int y = active ? tile / tiles_x : 0, x0 = active ? (tile - y * tiles_x) * 16 * MT : 0;
  const int ai = lane & 15, ak = lane >> 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // every wave walks ALL tap rows (the weights of a stage are the same for every output row); a tap row outside the image - zero
  // padding above / below - contributes nothing: its pixels are staged as zeros (wave-uniform: no loads are issued)
  constexpr int nstages = KS * CB;
  f32x4 ra[NA], rb[NB];
  auto fetch = [&](int s) __attribute__((always_inline)) {        // stage s: global -> registers
    const int ky = s / CB, cb = s - (s / CB) * CB;
    const int yy = y + ky - g.pad;
    const bool row_ok = active && yy >= 0 && yy < g.H;
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int i = lane + 64 * t, px = i >> 2, grp = i & 3;
      const int xx = x0 + px - g.pad;
      ra[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row_ok && i < P * 4 && xx >= 0 && xx < g.W) ra[t] = *reinterpret_cast<const f32x4*>(in + ((size_t)yy * g.W + xx) * g.cin + 16 * cb + 4 * grp);
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int i = threadIdx.x + kBlock * t;                // [kx][ak][COUTP] 16-byte words: NT x 64 per tap column
      const int kx = i / (NT * 64), r = i - kx * (NT * 64);
      rb[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < BV) rb[t] = *reinterpret_cast<const f32x4*>(w + ((((size_t)(ky * KS + kx) * CB + cb) * 4) * COUTP + r) * 4);
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {      // registers -> LDS
#pragma unroll
    for (int t = 0; t < NA; ++t) { const int i = lane + 64 * t; if (i < P * 4) As[buf][wave][i] = ra[t]; }
#pragma unroll
    for (int t = 0; t < NB; ++t) { const int i = threadIdx.x + kBlock * t; if (i < BV) Bs[buf][i] = rb[t]; }
  };
  fetch(0); stash(0);
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstages) fetch(s + 1);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = As[buf][wave][(16 * m + ai + kx) * 4 + ak];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = Bs[buf][(kx * 4 + ak) * COUTP + 16 * n + ai];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
    }
    if (s + 1 < nstages) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = 16 * n + ai;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + 16 * m + ak * 4 + r;
        if (active && x < g.Wo && co < g.cout) {
          float v = acc[m][n][r];
          if (LEAKY_OUT) v = v > 0.f ? v : kLeakySlope * v;
          out[((size_t)y * g.Wo + x) * g.cout + co] = v;
        }
      }
    }
}

// Weight gradient.  A work item is one (tap, 16-channel tile of ci); wave w of workgroup (band, group) owns the items
// [(4 group + w) IPW, +IPW) x all NT tiles of co and reduces the output rows of its band into part[band][KS][KS][CINP16][COUTP].
// The K loop (pixels) is unrolled 4 x 4 pixels with every operand load issued before the first MFMA: the loop is latency
// bound otherwise (one global round trip per 4 pixels).  CINP16: input channels rounded up to 16.
// PACK4 (cin <= 4, the first layer): the 16 rows of an M tile are 4 consecutive kx taps x 4 channels - one contiguous 64-byte
// segment of NHWC per pixel - instead of 16 channels of which 12 would be padding; an item is then (ky, group of 4 kx).
template <int KS, int MTI, int NT, int IPW, bool PACK4 = false>
__global__ __launch_bounds__(kBlock) void conv_wgrad_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ gout,
                                                             float* __restrict__ part, int rows_per_block) {
  constexpr int TAPS = KS * KS, KXG = (KS + 3) / 4, ITEMS = PACK4 ? KS * KXG : TAPS * MTI, U = 4;
  constexpr int CINP16 = 16 * MTI, COUTP = 16 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ai = lane & 15, ak = lane >> 4;                 // A: ci = ai, pixel-in-step = ak;  B: pixel-in-step = ak, co = ai
  const int item0 = (blockIdx.y * 4 + wave) * IPW;
  if (item0 >= ITEMS) return;
  f32x4 acc[IPW][NT];
#pragma unroll
  for (int t = 0; t < IPW; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int ky[IPW], kx[IPW], ci[IPW];
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    const int item = item0 + t < ITEMS ? item0 + t : ITEMS - 1;     // (a duplicate of the last item: computed, never stored)
    if (PACK4) {
      ky[t] = item / KXG;
      kx[t] = 4 * (item - ky[t] * KXG) + (ai >> 2);                  // this lane's tap of the group; >= KS: padding
      ci[t] = (kx[t] < KS) ? (ai & 3) : g.cin;                       // (channel >= cin reads as zero)
    } else {
      const int tap = item / MTI;
      ky[t] = tap / KS; kx[t] = tap - ky[t] * KS;
      ci[t] = 16 * (item - tap * MTI) + ai;
    }
  }
  const int y_begin = blockIdx.x * rows_per_block, y_end = min(y_begin + rows_per_block, g.Ho);
  for (int y = y_begin; y < y_end; ++y) {
    for (int x0 = 0; x0 < g.Wo; x0 += 4 * U) {
      float a[U][IPW], b[U][NT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int x = x0 + 4 * u + ak;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = 16 * n + ai;
          b[u][n] = (x < g.Wo && co < g.cout) ? gout[((size_t)y * g.Wo + x) * g.cout + co] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < IPW; ++t) {
          const int yy = y + ky[t] - g.pad, xx = x + kx[t] - g.pad;
          a[u][t] = (x < g.Wo && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W && ci[t] < g.cin) ? in[((size_t)yy * g.W + xx) * g.cin + ci[t]] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < IPW; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b[u][n], acc[t][n], 0, 0, 0);
    }
  }
  float* mine = part + (size_t)blockIdx.x * TAPS * CINP16 * COUTP;
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    const int item = item0 + t;
    if (item >= ITEMS) break;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ak * 4 + r, co = 16 * n + ai;                  // row of the tile = (lane >> 4) * 4 + r, column (co) = lane & 15
        if (PACK4) {
          const int kyi = item / KXG, kxi = 4 * (item - kyi * KXG) + (row >> 2);
          if (kxi < KS) mine[((size_t)(kyi * KS + kxi) * CINP16 + (row & 3)) * COUTP + co] = acc[t][n][r];
        } else {
          const int tap = item / MTI, m = item - tap * MTI;
          mine[((size_t)tap * CINP16 + 16 * m + row) * COUTP + co] = acc[t][n][r];
        }
      }
  }
}

// The generic weight gradient with its operands staged through LDS (cin, cout multiples of 4).  The kernel above issues one 4-byte
// load with its own bounds checks per operand element - U (IPW + NT) load instructions and ~8 VALU instructions each per 4 U IPW NT
// MFMAs: the address arithmetic costs as much as the matrix cores.  Here the four waves of a workgroup - the items (tap, 16-channel
// tile of ci) of one group - share a staged chunk of 32 output pixels: gout[32][COUTP] and in[KS rows][32 + KS - 1][CINP16], zero where
// the image ends, loaded with 16-byte accesses once per workgroup, double-buffered, one barrier per chunk.  The inner loop reads
// LDS at addresses that need no checks.  Same pixel order per weight: the same bits.
template <int KS, int MTI, int NT, int IPW>
__global__ __launch_bounds__(kBlock) void conv_wgrad_lds_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ gout,
                                                                 float* __restrict__ part, int rows_per_block) {
  constexpr int TAPS = KS * KS, ITEMS = TAPS * MTI, CH = 32, PA = CH + KS - 1;
  constexpr int CINP16 = 16 * MTI, COUTP = 16 * NT;
  constexpr int GV = CH * COUTP / 4, IV = KS * PA * CINP16 / 4;            // 16-byte words of a staged chunk
  constexpr int NG = (GV + kBlock - 1) / kBlock, NI = (IV + kBlock - 1) / kBlock;
  __shared__ f32x4 Gs[2][GV], Is[2][IV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ai = lane & 15, ak = lane >> 4;
  const int item0 = (blockIdx.y * 4 + wave) * IPW;
  f32x4 acc[IPW][NT];
#pragma unroll
  for (int t = 0; t < IPW; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoff[IPW];                                            // float offset of (ky, kx, ci tile) inside a staged `in` chunk, + my ci
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    const int item = item0 + t < ITEMS ? item0 + t : ITEMS - 1;     // (a duplicate of the last item: computed, never stored)
    const int tap = item / MTI, ky = tap / KS, kx = tap - ky * KS;
    aoff[t] = (ky * PA + kx) * CINP16 + 16 * (item - tap * MTI) + ai;
  }
  const int y_begin = blockIdx.x * rows_per_block, y_end = min(y_begin + rows_per_block, g.Ho);
  const int chunks_x = (g.Wo + CH - 1) / CH, nsteps = (y_end - y_begin) * chunks_x;
  f32x4 rg[NG], ri[NI];
  auto fetch = [&](int s) __attribute__((always_inline)) {
    const int y = y_begin + s / chunks_x, x0 = (s - (s / chunks_x) * chunks_x) * CH;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const int i = threadIdx.x + kBlock * t, px = i / (COUTP / 4), c4 = (i - px * (COUTP / 4)) * 4, x = x0 + px;
      rg[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < GV && x < g.Wo && c4 < g.cout) rg[t] = *reinterpret_cast<const f32x4*>(gout + ((size_t)y * g.Wo + x) * g.cout + c4);
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = threadIdx.x + kBlock * t;
      const int c4 = (i % (CINP16 / 4)) * 4, rest = i / (CINP16 / 4), px = rest % PA, ky = rest / PA;
      const int yy = y + ky - g.pad, xx = x0 + px - g.pad;
      ri[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (a pixel of `in` beyond the output row's last pixel + KS - 1 is never paired with a non-zero gout: the chunk's own bound suffices)
      if (i < IV && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W && c4 < g.cin) ri[t] = *reinterpret_cast<const f32x4*>(in + ((size_t)yy * g.W + xx) * g.cin + c4);
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NG; ++t) { const int i = threadIdx.x + kBlock * t; if (i < GV) Gs[buf][i] = rg[t]; }
#pragma unroll
    for (int t = 0; t < NI; ++t) { const int i = threadIdx.x + kBlock * t; if (i < IV) Is[buf][i] = ri[t]; }
  };
  if (nsteps > 0) { fetch(0); stash(0); }
  __syncthreads();
  const bool working = item0 < ITEMS;                       // (a wave without items still loads its share and takes part in the barriers)
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) fetch(s + 1);
    if (working) {
      const float* gs = reinterpret_cast<const float*>(Gs[buf]);
      const float* is = reinterpret_cast<const float*>(Is[buf]);
#pragma unroll
      for (int q = 0; q < CH / 16; ++q) {
        float a[4][IPW], b[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int px = 16 * q + 4 * u + ak;
#pragma unroll
          for (int n = 0; n < NT; ++n) b[u][n] = gs[px * COUTP + 16 * n + ai];
#pragma unroll
          for (int t = 0; t < IPW; ++t) a[u][t] = is[px * CINP16 + aoff[t]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < IPW; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b[u][n], acc[t][n], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) stash(buf ^ 1);
    __syncthreads();
  }
  if (!working) return;
  float* mine = part + (size_t)blockIdx.x * TAPS * CINP16 * COUTP;
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    const int item = item0 + t;
    if (item >= ITEMS) break;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ak * 4 + r, co = 16 * n + ai;
        const int tap = item / MTI, m = item - tap * MTI;
        mine[((size_t)tap * CINP16 + 16 * m + row) * COUTP + co] = acc[t][n][r];
      }
  }
}

// Weight gradient of the 64 -> 64 channel layers: M and N are PERMUTED (tile m, row i <-> channel 4 i + m; tile n, column j <->
// channel 4 j + n) so that a lane's operand for all four tiles is ONE float4 of its pixel (channels [4 (lane & 15), +4)):
// one 16-byte load of `in`, one of `gout` per 4 pixels and 16 MFMAs.  A wave owns one tap and all 4 x 4 tiles.
template <int KS>
__global__ __launch_bounds__(kBlock) void conv_wgrad64_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ gout,
                                                               float* __restrict__ part, int rows_per_block) {
  constexpr int TAPS = KS * KS, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ai = lane & 15, ak = lane >> 4;
  const int tap = blockIdx.y * 3 + wave;                   // (workgroups of three waves: the nine taps of a 3 x 3 kernel fill three of them)
  if (tap >= TAPS) return;
  const int ky = tap / KS, kx = tap - ky * KS;
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int y_begin = blockIdx.x * rows_per_block, y_end = min(y_begin + rows_per_block, g.Ho);
  for (int y = y_begin; y < y_end; ++y) {
    const int yy = y + ky - g.pad;
    if (yy < 0 || yy >= g.H) continue;
    for (int x0 = 0; x0 < g.Wo; x0 += 4 * U) {
      f32x4 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int x = x0 + 4 * u + ak, xx = x + kx - g.pad;
        a[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        b[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (x < g.Wo) {
          b[u] = *reinterpret_cast<const f32x4*>(gout + ((size_t)y * g.Wo + x) * 64 + 4 * ai);
          if (xx >= 0 && xx < g.W) a[u] = *reinterpret_cast<const f32x4*>(in + ((size_t)yy * g.W + xx) * 64 + 4 * ai);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u][n], acc[m][n], 0, 0, 0);
    }
  }
  float* mine = part + ((size_t)blockIdx.x * TAPS + tap) * 64 * 64;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(size_t)(4 * (ak * 4 + r) + m) * 64 + 4 * ai + n] = acc[m][n][r];
}

// The same with the operands staged through LDS: the three waves of a workgroup are the three tap COLUMNS of one tap row - they read the
// same row of `gout` and the same row of `in`, shifted by one pixel each.  Chunks of 32 output pixels: gout[32][64] and in[34][64]
// are loaded once per workgroup (L2 traffic / 3), double-buffered, one barrier per chunk (32 pixels: 32 MFMAs per wave).  Same pixel order
// per weight as the kernel above: the same bits.
template <int KS>
__global__ __launch_bounds__(64 * KS) void conv_wgrad64_lds_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ gout,
                                                                    float* __restrict__ part, int rows_per_block) {
  constexpr int TAPS = KS * KS, CH = 32, PA = CH + KS - 1, NTH = 64 * KS;   // (chunks of 32 pixels: 33 KB of LDS, four workgroups = 12 waves per CU)
  constexpr int NLB = (CH * 16 + NTH - 1) / NTH, NLA = (PA * 16 + NTH - 1) / NTH;       // 16-byte loads per thread: gout chunk, in chunk
  __shared__ f32x4 Gs[2][CH * 16], Is[2][PA * 16];          // [pixel][16 groups of 4 channels]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ai = lane & 15, ak = lane >> 4;
  const int ky = blockIdx.y, kx = wave;                      // (one workgroup = one tap row, one wave per tap column)
  const int tap = ky * KS + kx;
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int y_begin = blockIdx.x * rows_per_block, y_end = min(y_begin + rows_per_block, g.Ho);
  const int chunks_x = (g.Wo + CH - 1) / CH;
  f32x4 rg[NLB], ri[NLA];
  auto fetch = [&](int y, int c) __attribute__((always_inline)) {
    const int yy = y + ky - g.pad, x0 = c * CH;
    const bool row_ok = yy >= 0 && yy < g.H;
#pragma unroll
    for (int t = 0; t < NLB; ++t) {
      const int i = threadIdx.x + NTH * t, px = i >> 4, grp = i & 15, x = x0 + px;
      rg[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CH * 16 && x < g.Wo) rg[t] = *reinterpret_cast<const f32x4*>(gout + ((size_t)y * g.Wo + x) * 64 + 4 * grp);
    }
#pragma unroll
    for (int t = 0; t < NLA; ++t) {
      const int i = threadIdx.x + NTH * t, px = i >> 4, grp = i & 15, xx = x0 + px - g.pad;
      ri[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row_ok && i < PA * 16 && xx >= 0 && xx < g.W) ri[t] = *reinterpret_cast<const f32x4*>(in + ((size_t)yy * g.W + xx) * 64 + 4 * grp);
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NLB; ++t) { const int i = threadIdx.x + NTH * t; if (i < CH * 16) Gs[buf][i] = rg[t]; }
#pragma unroll
    for (int t = 0; t < NLA; ++t) { const int i = threadIdx.x + NTH * t; if (i < PA * 16) Is[buf][i] = ri[t]; }
  };
  const int nsteps = (y_end - y_begin) * chunks_x;
  if (nsteps > 0) { fetch(y_begin, 0); stash(0); }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) { const int yn = y_begin + (s + 1) / chunks_x, cn = (s + 1) - ((s + 1) / chunks_x) * chunks_x; fetch(yn, cn); }
    // (a tap row outside the image was staged as zeros: its products vanish; same as the `continue` of the direct kernel)
#pragma unroll
    for (int q = 0; q < CH / 16; ++q) {                     // 16 pixels = 4 x (4 pixels, one per lane group ak)
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = 16 * q + 4 * u + ak;
        b[u] = Gs[buf][px * 16 + ai];
        a[u] = Is[buf][(px + kx) * 16 + ai];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u][n], acc[m][n], 0, 0, 0);
    }
    if (s + 1 < nsteps) stash(buf ^ 1);
    __syncthreads();
  }
  float* mine = part + ((size_t)blockIdx.x * TAPS + tap) * 64 * 64;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(size_t)(4 * (ak * 4 + r) + m) * 64 + 4 * ai + n] = acc[m][n][r];
}

// dW[tap][ci][co] (true sizes) = sum of the band partials in a fixed order: a workgroup owns 64 weights, its 4 waves add the
// bands b = wave, wave + 4, ... and the four wave sums are added in wave order
__global__ __launch_bounds__(kBlock) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblocks, int taps,
                                                                    int cinp16, int coutp, int cin, int cout) {
  __shared__ float sm[kBlock];
  const int n = taps * cin * cout;
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  float s = 0.f;
  if (k < n) {
    const int co = k % cout, ci = (k / cout) % cin, tap = k / (cout * cin);
    const size_t src = ((size_t)tap * cinp16 + ci) * coutp + co, stride = (size_t)taps * cinp16 * coutp;
    for (int b = wave; b < nblocks; b += 4) s += part[b * stride + src];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (wave == 0 && k < n) dw[k] = ((sm[threadIdx.x] + sm[64 + threadIdx.x]) + sm[128 + threadIdx.x]) + sm[192 + threadIdx.x];
}

// The same for cout % 4 == 0 (every layer but the last): a lane owns FOUR consecutive output channels of one (tap, ci) - 16-byte
// loads, four independent sums per lane, eight bands in flight per wave; a workgroup owns 256 weights.  Same summation order per
// weight as the scalar kernel (bands b = wave, wave + 4, ... inside a wave, then the four waves in order).
__global__ __launch_bounds__(kBlock) void conv_wgrad_reduce4_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblocks, int taps,
                                                                     int cinp16, int coutp, int cin, int cout) {
  __shared__ f32x4 sm[kBlock];
  const int n4 = taps * cin * (cout >> 2);
  const int k4 = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (k4 < n4) {
    const int cq = cout >> 2;
    const int co = (k4 % cq) * 4, ci = (k4 / cq) % cin, tap = k4 / (cq * cin);
    const size_t src = ((size_t)tap * cinp16 + ci) * coutp + co, stride = (size_t)taps * cinp16 * coutp;
    int b = wave;
    for (; b + 28 < nblocks; b += 32) {                      // eight bands of this wave at once: the loads are independent, the sums ordered
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (size_t)(b + 4 * u) * stride + src);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += 4) s += *reinterpret_cast<const f32x4*>(part + (size_t)b * stride + src);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (wave == 0 && k4 < n4) *reinterpret_cast<f32x4*>(dw + (size_t)k4 * 4) = ((sm[threadIdx.x] + sm[64 + threadIdx.x]) + sm[128 + threadIdx.x]) + sm[192 + threadIdx.x];
}
static void launch_wgrad_reduce(const float* part, float* dw, int nblocks, int taps, int cinp16, int coutp, int cin, int cout, hipStream_t stream) {
  const int n = taps * cin * cout;
  if (cout % 4 == 0 && coutp % 4 == 0) conv_wgrad_reduce4_kernel<<<(n / 4 + 63) / 64, kBlock, 0, stream>>>(part, dw, nblocks, taps, cinp16, coutp, cin, cout);
  else conv_wgrad_reduce_kernel<<<(n + 63) / 64, kBlock, 0, stream>>>(part, dw, nblocks, taps, cinp16, coutp, cin, cout);
}

// row bands = partial sums per weight: what the second stage has to add (and re-read).  256: one output row per band at config 4's
// size - with bands of two rows the 9 x 126 waves of the 64 -> 64 layer left SIMDs with two waves next to SIMDs with one
constexpr int kWgradMaxBlocks = 256;

template <int KS, int CINP, int NT>
static int launch_forward(const ConvGeom& g, const float* in, const float* w, float* out, int leaky, hipStream_t stream) {
  if constexpr (CINP >= 16 && KS >= 3) {
    if (opt(OPT_CONV_LDS) != 0) {                            // operands staged through LDS (option conv_lds 0: the direct kernel)
      const int tiles2 = ((g.Wo + 63) / 64) * g.Ho, grid2 = (tiles2 + kBlock / 64 - 1) / (kBlock / 64);
      if (leaky) conv_forward_lds_kernel<KS, CINP, NT, true><<<grid2, kBlock, 0, stream>>>(g, in, w, out);
      else conv_forward_lds_kernel<KS, CINP, NT, false><<<grid2, kBlock, 0, stream>>>(g, in, w, out);
      PISO_LAUNCH_CHECK();
      return PISO_OK;
    }
  }
  const int tiles = ((g.Wo + 63) / 64) * g.Ho;
  const int grid = (tiles + kBlock / 64 - 1) / (kBlock / 64);
  if (leaky) conv_forward_kernel<KS, CINP, NT, true><<<grid, kBlock, 0, stream>>>(g, in, w, out);
  else conv_forward_kernel<KS, CINP, NT, false><<<grid, kBlock, 0, stream>>>(g, in, w, out);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

template <int KS, int MTI, int NT, int IPW, bool PACK4 = false>
static int launch_wgrad(const ConvGeom& g, const float* in, const float* gout, float* part, float* dw, hipStream_t stream) {
  const int rows_per_block = (g.Ho + kWgradMaxBlocks - 1) / kWgradMaxBlocks;
  const int nblocks = (g.Ho + rows_per_block - 1) / rows_per_block;
  constexpr int items = PACK4 ? KS * ((KS + 3) / 4) : KS * KS * MTI;
  constexpr int groups = (items + 4 * IPW - 1) / (4 * IPW);
  bool staged = false;
  if constexpr (!PACK4) {
    if (opt(OPT_CONV_LDS) != 0 && g.cin % 4 == 0 && g.cout % 4 == 0) {
      conv_wgrad_lds_kernel<KS, MTI, NT, IPW><<<dim3(nblocks, groups), kBlock, 0, stream>>>(g, in, gout, part, rows_per_block);
      staged = true;
    }
  }
  if (!staged) conv_wgrad_kernel<KS, MTI, NT, IPW, PACK4><<<dim3(nblocks, groups), kBlock, 0, stream>>>(g, in, gout, part, rows_per_block);
  PISO_LAUNCH_CHECK();
  launch_wgrad_reduce(part, dw, nblocks, KS * KS, 16 * MTI, 16 * NT, g.cin, g.cout, stream);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace piso

using namespace piso;

namespace piso {
// g' = g * leaky'(pre-activation) from the layer's saved OUTPUT (a leaky ReLU with a positive slope keeps the sign): the gradient of
// the pre-activation that both the input gradient and the weight gradient consume.  One pass (torch: a multiply and a where).
__global__ __launch_bounds__(kBlock) void leaky_backward_kernel(const float* __restrict__ g, const float* __restrict__ out, float* __restrict__ gp, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i], ov = reinterpret_cast<const f32x4*>(out)[i];
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ov[q] > 0.f ? gv[q] : kLeakySlope * gv[q];
    reinterpret_cast<f32x4*>(gp)[i] = r;
  }
}
__global__ __launch_bounds__(kBlock) void leaky_backward_tail_kernel(const float* __restrict__ g, const float* __restrict__ out, float* __restrict__ gp, size_t begin, size_t n) {
  const size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) gp[i] = out[i] > 0.f ? g[i] : kLeakySlope * g[i];
}
}  // namespace piso

extern "C" {
int piso_leaky_relu_backward(const float* grad_out, const float* out, float* grad_pre, size_t n, piso_stream_t stream_) {
  using namespace piso;
  if (!grad_out || !out || !grad_pre) { set_error_msg("piso_leaky_relu_backward: invalid argument"); return PISO_ERR_INVALID_ARG; }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool aligned = ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(grad_pre)) & 15) == 0;
  const size_t n4 = aligned ? n / 4 : 0;
  if (n4 > 0) leaky_backward_kernel<<<grid_for((long long)n4, kBlock * 2, 4096), kBlock, 0, stream>>>(grad_out, out, grad_pre, n4);
  if (n4 * 4 < n) leaky_backward_tail_kernel<<<(int)((n - n4 * 4 + kBlock - 1) / kBlock), kBlock, 0, stream>>>(grad_out, out, grad_pre, n4 * 4, n);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

static inline int padded_cin(int cin) { return cin <= 4 ? 4 : round_up(cin, 16); }

// Weight layout expected by piso_conv2d_forward (zero filled beyond the true channel counts), COUTP = round_up(cout, 16):
//   cin <= 4 : [ks][ks][4][COUTP]                                 (HWIO, channels padded to 4)
//   cin  > 4 : [ks][ks][CINP / 16][4][COUTP][4], CINP = round_up(cin, 16): element [tap][blk][q][co][j] = W[tap][16 blk + 4 q + j][co]
size_t piso_conv2d_weight_elems(int ks, int cin, int cout) { return (size_t)ks * ks * padded_cin(cin) * round_up(cout, 16); }

size_t piso_conv2d_wgrad_workspace_bytes(int ks, int cin, int cout) {
  return (size_t)kWgradMaxBlocks * ks * ks * round_up(cin, 16) * round_up(cout, 16) * sizeof(float);
}

int piso_conv2d_forward(const float* in, const float* w_laid_out, float* out, int H, int W, int cin, int cout, int ks, int pad, int leaky_out,
                        piso_stream_t stream_) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  ConvGeom g;
  g.H = H; g.W = W; g.pad = pad; g.cin = cin; g.cout = cout;
  g.Ho = H + 2 * pad - ks + 1; g.Wo = W + 2 * pad - ks + 1;
  if (!in || !w_laid_out || !out || g.Ho < 1 || g.Wo < 1 || cin < 1 || cout < 1 || cout > 64 || cin > 64 || (cin > 4 && cin % 16 != 0)) {
    set_error_msg("piso_conv2d_forward: invalid argument (channels: 1..4 or a multiple of 16 up to 64 in, 1..64 out; kernel size 1 | 3 | 5 | 7)");
    return PISO_ERR_INVALID_ARG;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int cinp = padded_cin(cin), nt = round_up(cout, 16) / 16;
#define PISO_CONV_FWD(KS, CINP, NT) \
  if (ks == KS && cinp == CINP && nt == NT) return launch_forward<KS, CINP, NT>(g, in, w_laid_out, out, leaky_out, stream)
  // the layers of the closure and of its input-gradient pass (channel roles swapped)
  PISO_CONV_FWD(7, 4, 1);  PISO_CONV_FWD(7, 16, 1);
  PISO_CONV_FWD(5, 16, 1); PISO_CONV_FWD(5, 16, 2); PISO_CONV_FWD(5, 32, 1);
  PISO_CONV_FWD(3, 32, 4); PISO_CONV_FWD(3, 64, 2); PISO_CONV_FWD(3, 64, 4);
  PISO_CONV_FWD(1, 64, 4); PISO_CONV_FWD(1, 64, 1); PISO_CONV_FWD(1, 4, 4);
#undef PISO_CONV_FWD
  set_error_msg("piso_conv2d_forward: this (kernel size, channels) combination is not instantiated");
  return PISO_ERR_INVALID_ARG;
}

int piso_conv2d_wgrad(const float* in, const float* grad_out, float* dw, int H, int W, int cin, int cout, int ks, int pad, void* workspace,
                      size_t workspace_bytes, piso_stream_t stream_) {
  const piso::OptScope knobs;                              // (the call works on a snapshot of the knobs, options.h)
  ConvGeom g;
  g.H = H; g.W = W; g.pad = pad; g.cin = cin; g.cout = cout;
  g.Ho = H + 2 * pad - ks + 1; g.Wo = W + 2 * pad - ks + 1;
  if (!in || !grad_out || !dw || !workspace || g.Ho < 1 || g.Wo < 1 || cin < 1 || cout < 1 || cout > 64 || cin > 64 ||
      workspace_bytes < piso_conv2d_wgrad_workspace_bytes(ks, cin, cout)) {
    set_error_msg("piso_conv2d_wgrad: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  const int mti = round_up(cin, 16) / 16, nt = round_up(cout, 16) / 16;
#define PISO_CONV_WG(KS, MTI, NT, IPW) \
  if (ks == KS && mti == MTI && nt == NT) return launch_wgrad<KS, MTI, NT, IPW>(g, in, grad_out, part, dw, stream)
  if (cin == 64 && cout == 64 && ks == 3) {      // (measured: 304 us against 329 us for the generic kernel at 250 x 876; the 1 x 1
    // layer has a single tap, i.e. one busy wave per workgroup here, and stays on the generic kernel: 130 us against 219 us)
    const int rows_per_block = (g.Ho + kWgradMaxBlocks - 1) / kWgradMaxBlocks, nblocks = (g.Ho + rows_per_block - 1) / rows_per_block;
    if (opt(OPT_CONV_LDS) != 0) conv_wgrad64_lds_kernel<3><<<dim3(nblocks, 3), 192, 0, stream>>>(g, in, grad_out, part, rows_per_block);
    else conv_wgrad64_kernel<3><<<dim3(nblocks, 3), 192, 0, stream>>>(g, in, grad_out, part, rows_per_block);
    PISO_LAUNCH_CHECK();
    launch_wgrad_reduce(part, dw, nblocks, ks * ks, 64, 64, 64, 64, stream);
    PISO_LAUNCH_CHECK();
    return PISO_OK;
  }
  // (items per wave: enough waves to fill the chip, few enough registers for the 4 x 4-pixel prefetch)
  if (ks == 7 && cin <= 4 && nt == 1) return launch_wgrad<7, 1, 1, 1, true>(g, in, grad_out, part, dw, stream);
  PISO_CONV_WG(7, 1, 1, 3); PISO_CONV_WG(5, 1, 1, 2); PISO_CONV_WG(5, 1, 2, 2); PISO_CONV_WG(3, 2, 4, 1); PISO_CONV_WG(3, 4, 4, 1);
  PISO_CONV_WG(1, 4, 4, 1); PISO_CONV_WG(1, 4, 1, 1);
#undef PISO_CONV_WG
  set_error_msg("piso_conv2d_wgrad: this (kernel size, channels) combination is not instantiated");
  return PISO_ERR_INVALID_ARG;
}

}  // extern "C"

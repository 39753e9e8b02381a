// Fused stencil glue of the PISO step on the flat "u-first" face layout (u [ny][nx+1] followed by v [ny+1][nx]).
//
// Replaces the ~25 TensorFlow / PhiFlow-math ops per step of diffpiso/piso_tf.py:36-73 and diffpiso/piso_helpers.py:35-55,
// 169-172, 226-310 (custom_padded, arrange_rhs_term_tf, finite_volume_gradient_tensor, finite_volume_divergence, the
// element-wise velocity updates) by a handful of launches, forward AND reverse mode.  The reverse-mode kernels implement
// the adjoints AS THE REFERENCE'S GRAPH COMPUTES THEM:
//   * periodic axes of the pressure gradient: g[:-1] - g[1:], no wrap term, duplicate face ignored (piso_helpers.py:230-232);
//   * divergence: the custom gradient of piso_helpers.py:291-305, whose periodic branch feeds face 0 with dc[N-2].
// Arithmetic is float32 in the reference's operation order (products / quotients are not re-associated; -ffp-contract=off).
// All kernels are HBM-trivial element-wise / 5-point gathers: one thread per output element, x fastest, coalesced.
// Slab-decomposed step (piso_*_slab entry points): the same kernels, the same WHOLE-GRID index arithmetic - every array access goes
// through the rank's RowMap (piso_common.h), which is the identity on one GPU.
#include "piso_common.h"

namespace piso {

enum { PAD_ZERO = 0, PAD_EDGE = 1, PAD_WRAP = 2 };          // 'constant', 'boundary', 'periodic' of CenteredGrid.padded
enum { FACE_RHS = 0, FACE_CORR1 = 1, FACE_FINAL = 2 };

struct GlueGeom {
  int nx, ny;
  int per_x, per_y;            // velocity periodicity (x, y)
  int px_lo, px_hi, py_lo, py_hi;   // pressure pad mode per side
  float dxdy, hx, hy, beta;
};

__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// ---- custom_padded + flatten (piso_helpers.py:35-55, piso_tf.py:93): padded u [ny+2][nx+3] then padded v [ny+3][nx+2]
// (pw: the padded rows a windowed launch fills - u_lo / v_lo are element offsets into the padded u / padded v array)
__global__ __launch_bounds__(kBlock) void pad_velocity_kernel(const float* __restrict__ vel, float* __restrict__ out, int nx, int ny,
                                                               int per_x, int per_y, FaceWin pw, RowMap M) {
  const int pu = (ny + 2) * (nx + 3);
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < pw.count(); w += gridDim.x * kBlock) {
    const int k = pw.map(w);
    if (k < pu) {
      const int jj = k / (nx + 3), ii = k - jj * (nx + 3);
      const int j = per_y ? wrap(jj - 1, ny) : clampi(jj - 1, 0, ny - 1);                 // cross axis: (1, 1)
      const int i = per_x ? (ii - 1 + nx) % nx : clampi(ii - 1, 0, nx);                   // own axis: duplicate dropped, (1, 2)
      out[M.pad(k)] = vel[M.u(j, i)];
    } else {
      const int q = k - pu;
      const int jj = q / (nx + 2), ii = q - jj * (nx + 2);
      const int j = per_y ? (jj - 1 + ny) % ny : clampi(jj - 1, 0, ny);
      const int i = per_x ? wrap(ii - 1, nx) : clampi(ii - 1, 0, nx - 1);
      out[M.pad(k)] = vel[M.v(j, i)];
    }
  }
}

// ---- A0 = (1 / (beta - A)) * dx_factor on every face, flat "v-first" (piso_tf.py:53-54, piso_cuda_pressure_solver.py:70)
__global__ __launch_bounds__(kBlock) void a0_vfirst_kernel(const float* __restrict__ A, float* __restrict__ a0, int n_u, int n_v, float beta,
                                                            float dx_factor, FaceWin fw, RowMap M) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int src = fw.map(w);                              // u-first index of the face
    const int k = src < n_u ? n_v + src : src - n_u;        // its place in the v-first vector
    a0[M.face_vfirst(k)] = (1.0f / (beta - A[M.face(src)])) * dx_factor;
  }
}

// pressure difference across face k of an axis with n cells (k = 0 .. n): upper[k] - lower[k] of finite_volume_gradient_tensor
// (piso_helpers.py:236-254; periodic: circular_padded_gradient :226-229).  `at(c)` returns p at cell c of that axis line.
template <typename F>
__device__ __forceinline__ float face_difference(F at, int k, int n, int lo_mode, int hi_mode) {
  if (lo_mode == PAD_WRAP) {
    const int c = (k == n) ? 0 : k;                        // the duplicate face repeats face 0
    return at(c) - at(c == 0 ? n - 1 : c - 1);
  }
  const float upper = (k < n) ? at(k) : (hi_mode == PAD_EDGE ? at(n - 1) : 0.0f);
  const float lower = (k > 0) ? at(k - 1) : (lo_mode == PAD_EDGE ? at(0) : 0.0f);
  return upper - lower;
}

// min(accessible_lo, accessible_hi) of a face (piso_helpers.py:255-265); acc is the padded [ny+2][nx+2] mask, or NULL
__device__ __forceinline__ float face_mask(const RowMap& M, const float* __restrict__ acc, int comp, int j, int i, int nx) {
  if (!acc) return 1.0f;
  const int w = nx + 2;
  if (comp == 0) return fminf(acc[M.mask((j + 1) * w + i + 1)], acc[M.mask((j + 1) * w + i)]);       // u face (j, i): cells (j, i-1) | (j, i)
  return fminf(acc[M.mask((j + 1) * w + i + 1)], acc[M.mask(j * w + i + 1)]);                         // v face (j, i): cells (j-1, i) | (j, i)
}

// G(p) on flat face f: ((difference * dxdy) / h) * mask
__device__ __forceinline__ float face_gradient(const GlueGeom& g, const RowMap& M, const float* __restrict__ p, const float* __restrict__ acc, int f) {
  const int nx = g.nx, ny = g.ny, n_u = (nx + 1) * ny;
  if (f < n_u) {
    const int j = f / (nx + 1), i = f - j * (nx + 1);
    const float d = face_difference([&](int c) { return p[M.c(j, c)]; }, i, nx, g.px_lo, g.px_hi);
    return ((d * g.dxdy) / g.hx) * face_mask(M, acc, 0, j, i, nx);
  }
  const int q = f - n_u, j = q / nx, i = q - j * nx;
  const float d = face_difference([&](int c) { return p[M.c(c, i)]; }, j, ny, g.py_lo, g.py_hi);
  return ((d * g.dxdy) / g.hy) * face_mask(M, acc, 1, j, i, nx);
}

// ---- the three face updates that contain a pressure gradient:
//   FACE_RHS   out0 = m ? -dv : (in0 * beta - G(p) [+ in1 * dxdy])          in0 = velocity, in1 = forcing (or NULL), in2 = dv   (piso_tf.py:36-39)
//   FACE_CORR1 out0 = in0 - (G(p) / bmA) / dxdy ; out1 = out0 - in0           in0 = u*                                             (:58, :61)
//   FACE_FINAL out0 = in0 + (in1 - G(p) / dxdy) / bmA                         in0 = u**, in1 = H                                   (:71-72)
template <int MODE>
__global__ __launch_bounds__(kBlock) void face_forward_kernel(GlueGeom g, const float* __restrict__ p, const float* __restrict__ acc,
                                                               const float* __restrict__ A, const float* __restrict__ in0,
                                                               const float* __restrict__ in1, const float* __restrict__ in2,
                                                               const uint8_t* __restrict__ dmask, float* __restrict__ out0,
                                                               float* __restrict__ out1, FaceWin fw, RowMap M) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int fg = fw.map(w);
    const float gp = face_gradient(g, M, p, acc, fg);
    const int f = M.face(fg);                               // (everything below is element-wise on the STORED vectors)
    if (MODE == FACE_RHS) {
      float r = in0[f] * g.beta - gp;
      if (in1) r = r + in1[f] * g.dxdy;
      out0[f] = (dmask && dmask[f]) ? in2[f] * -1.0f : r;
    } else if (MODE == FACE_CORR1) {
      const float bmA = g.beta - A[f];
      const float s2 = in0[f] - (gp / bmA) / g.dxdy;
      out0[f] = s2;
      out1[f] = s2 - in0[f];
    } else {
      const float bmA = g.beta - A[f];
      out0[f] = in0[f] + (in1[f] - gp / g.dxdy) / bmA;
    }
  }
}

// weight of G(p)[f] in the incoming gradient: dL/dG on face f, BEFORE the mask and the dxdy / h scale
template <int MODE>
__device__ __forceinline__ float face_weight(const GlueGeom& g, const float* __restrict__ A, const uint8_t* __restrict__ dmask,
                                              const float* __restrict__ d0, const float* __restrict__ d1, int f) {
  if (MODE == FACE_RHS) return (dmask && dmask[f]) ? 0.0f : -d0[f];
  const float bmA = g.beta - A[f];
  if (MODE == FACE_CORR1) return -(((d0[f] + (d1 ? d1[f] : 0.0f)) / g.dxdy) / bmA);
  return -((d0[f] / bmA) / g.dxdy);
}

// element-wise part of the reverse mode (everything but d p)
template <int MODE>
__global__ __launch_bounds__(kBlock) void face_backward_kernel(GlueGeom g, const float* __restrict__ A, const uint8_t* __restrict__ dmask,
                                                                const float* __restrict__ d0, const float* __restrict__ d1,
                                                                float* __restrict__ g0, float* __restrict__ g1, float* __restrict__ g2, FaceWin fw, RowMap M) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int f = M.face(fw.map(w));
    if (MODE == FACE_RHS) {
      const bool m = dmask && dmask[f];
      const float d = m ? 0.0f : d0[f];
      g0[f] = d * g.beta;                                  // d velocity
      if (g1) g1[f] = d * g.dxdy;                          // d forcing
      if (g2) g2[f] = m ? d0[f] * -1.0f : 0.0f;            // d dirichlet_values
    } else if (MODE == FACE_CORR1) {
      g0[f] = d0[f];                                       // d u*: s2 = u* - q, delta = s2 - u*  ->  (d0 + d1) - d1
    } else {
      g0[f] = d0[f];                                       // d u**
      g1[f] = d0[f] / (g.beta - A[f]);                     // d H
    }
  }
}

// reverse mode of G w.r.t. p, one thread per cell: gather of the face weights with the reference's adjoint stencil
template <int MODE>
__global__ __launch_bounds__(kBlock) void gradient_adjoint_kernel(GlueGeom g, const float* __restrict__ acc, const float* __restrict__ A,
                                                                   const uint8_t* __restrict__ dmask, const float* __restrict__ d0,
                                                                   const float* __restrict__ d1, float* __restrict__ dp, CellWin cw, RowMap M) {
  const int nx = g.nx, ny = g.ny;
  for (int c = cw.lo + blockIdx.x * kBlock + threadIdx.x; c < cw.lo + cw.n; c += gridDim.x * kBlock) {
    const int j = c / nx, i = c - j * nx;
    // scaled, masked gradient w.r.t. the face differences: ((w * mask) / h) * dxdy
    auto wu = [&](int k) { const int f = M.u(j, k); return ((face_weight<MODE>(g, A, dmask, d0, d1, f) * face_mask(M, acc, 0, j, k, nx)) / g.hx) * g.dxdy; };
    auto wv = [&](int k) { const int f = M.v(k, i); return ((face_weight<MODE>(g, A, dmask, d0, d1, f) * face_mask(M, acc, 1, k, i, nx)) / g.hy) * g.dxdy; };
    // x axis: d p[i] = w[i] - w[i+1] (+ replicate-pad terms on a 'boundary' side; nothing more on periodic axes: C-8)
    float s = wu(i) - wu(i + 1);
    if (g.px_lo != PAD_WRAP) {
      if (i == nx - 1 && g.px_hi == PAD_EDGE) s += wu(nx);      // upper[nx] = p[nx-1]
      if (i == 0 && g.px_lo == PAD_EDGE) s -= wu(0);            // lower[0] = p[0]
    }
    float t = wv(j) - wv(j + 1);
    if (g.py_lo != PAD_WRAP) {
      if (j == ny - 1 && g.py_hi == PAD_EDGE) t += wv(ny);
      if (j == 0 && g.py_lo == PAD_EDGE) t -= wv(0);
    }
    dp[M.c(j, i)] = t + s;                                  // (axis 0 = y first, then x: the order the oracle accumulates in)
  }
}

// ---- finite_volume_divergence (piso_helpers.py:277-289) on flat faces
__global__ __launch_bounds__(kBlock) void divergence_kernel(const float* __restrict__ faces, float* __restrict__ div, int nx, int ny, float dxdy,
                                                             float hx, float hy, CellWin cw, RowMap M) {
  for (int c = cw.lo + blockIdx.x * kBlock + threadIdx.x; c < cw.lo + cw.n; c += gridDim.x * kBlock) {
    const int j = c / nx, i = c - j * nx;
    const float dy_term = ((faces[M.v(j + 1, i)] - faces[M.v(j, i)]) * dxdy) / hy;
    const float dx_term = ((faces[M.u(j, i + 1)] - faces[M.u(j, i)]) * dxdy) / hx;
    div[M.c(j, i)] = dy_term + dx_term;
  }
}

// the reference's custom gradient of the divergence (piso_helpers.py:291-305): per axis, faces k = 0 .. n:
//   non-periodic  r[k] = -[k < n] dc[k] f + [k > 0] dc[k-1] f
//   periodic      r[k] = -dc[k < n ? k : 0] f + dc[k > 0 ? k-1 : n-2] f           (face 0 receives dc[n-2]: App. C-7)
template <typename F>
__device__ __forceinline__ float div_adjoint_axis(F at, int k, int n, int periodic, float dxdy, float h) {      // at(c): dc at cell c of the axis line
  float lo_term, hi_term;                                   // -cat(dc, first | 0)[k] , cat(last | 0, dc)[k]
  if (periodic) {
    hi_term = at(k < n ? k : 0);
    lo_term = at(k > 0 ? k - 1 : n - 2);
  } else {
    hi_term = (k < n) ? at(k) : 0.0f;
    lo_term = (k > 0) ? at(k - 1) : 0.0f;
  }
  return -((hi_term * dxdy) / h) + (lo_term * dxdy) / h;
}
__global__ __launch_bounds__(kBlock) void divergence_adjoint_kernel(const float* __restrict__ dc, float* __restrict__ dfaces, int nx, int ny,
                                                                     int per_x, int per_y, float dxdy, float hx, float hy, FaceWin fw, RowMap M) {
  const int n_u = (nx + 1) * ny;
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int f = fw.map(w);
    if (f < n_u) {
      const int j = f / (nx + 1), i = f - j * (nx + 1);
      dfaces[M.u(j, i)] = div_adjoint_axis([&](int c) { return dc[M.c(j, c)]; }, i, nx, per_x, dxdy, hx);
    } else {
      const int q = f - n_u, j = q / nx, i = q - j * nx;
      dfaces[M.v(j, i)] = div_adjoint_axis([&](int c) { return dc[M.c(c, i)]; }, j, ny, per_y, dxdy, hy);
    }
  }
}

// ---- second corrector: H = M d - (A - beta) d on faces (piso_helpers.py:223), div2 = D(H / (beta - A)) (piso_tf.py:66)
__global__ __launch_bounds__(kBlock) void h_kernel(const float* __restrict__ Md, const float* __restrict__ delta, const float* __restrict__ A,
                                                    float beta, float* __restrict__ H, float* __restrict__ Hb, FaceWin fw, RowMap M) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int f = M.face(fw.map(w));
    const float h = Md[f] - (A[f] - beta) * delta[f];
    H[f] = h;
    Hb[f] = h / (beta - A[f]);
  }
}
// reverse: d_Hb = divergence adjoint (computed by the caller into `dHb`), d_H_total = d_H + d_Hb / bmA
__global__ __launch_bounds__(kBlock) void h_adjoint_kernel(const float* __restrict__ dH, const float* __restrict__ dHb, const float* __restrict__ A,
                                                            float beta, float* __restrict__ dMd, float* __restrict__ ddelta, FaceWin fw, RowMap M) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < fw.count(); w += gridDim.x * kBlock) {
    const int f = M.face(fw.map(w));
    const float t = (dH ? dH[f] : 0.0f) + dHb[f] / (beta - A[f]);
    dMd[f] = t;
    ddelta[f] = -((A[f] - beta) * t);
  }
}

static int glue_grid(long long n) { return grid_for(n, kBlock * 2, 2048); }

}  // namespace piso

using namespace piso;

namespace {

#define PISO_SLAB_CHECK(what)                                                                                     \
  if (!slab_ok(slab, ny)) { set_error_msg(what ": invalid slab (rows must lie inside the grid, at least 4, at most ny - 6)"); return PISO_ERR_INVALID_ARG; }

int pad_velocity_impl(const float* vel_flat, float* vel_pad, int nx, int ny, int periodic_x, int periodic_y, piso_stream_t stream_,
                      const piso_slab_t* slab) {
  if (!vel_flat || !vel_pad || nx < 1 || ny < 1) { set_error_msg("piso_pad_velocity: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_pad_velocity");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // slab: only the padded rows the assembly of this rank's face rows reads - padded u rows [j0, j1 + 1 + last), padded v rows
  // [j0, j1 + 2 + last) (assembly.hip: a u row j reads padded u row j + 1 and padded v rows j + 1, j + 2; a v row j reads padded u
  // rows j, j + 1 and padded v rows j .. j + 2)
  const RowMap M = make_row_map(slab, nx, ny);
  const int pu = (ny + 2) * (nx + 3);
  FaceWin pw{0, pu, pu, (ny + 3) * (nx + 2)};
  if (M.on) {
    const int u_hi = M.j1 + 1 + M.last, v_hi = M.j1 + 2 + M.last;
    pw = FaceWin{M.j0 * (nx + 3), (u_hi - M.j0) * (nx + 3), pu + M.j0 * (nx + 2), (v_hi - M.j0) * (nx + 2)};
  }
  pad_velocity_kernel<<<glue_grid(pw.count()), kBlock, 0, stream>>>(vel_flat, vel_pad, nx, ny, periodic_x, periodic_y, pw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int a0_vfirst_impl(const float* a_flat, float* a0_vfirst, int nx, int ny, float beta, float dx_factor, piso_stream_t stream_,
                   const piso_slab_t* slab) {
  if (!a_flat || !a0_vfirst || nx < 1 || ny < 1) { set_error_msg("piso_a0_vfirst: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_a0_vfirst");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int n_u = (nx + 1) * ny, n_v = nx * (ny + 1);
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  a0_vfirst_kernel<<<glue_grid(fw.count()), kBlock, 0, stream>>>(a_flat, a0_vfirst, n_u, n_v, beta, dx_factor, fw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int make_geom(GlueGeom& g, int nx, int ny, int periodic_x, int periodic_y, const int pad_modes[4], float dxdy, float hx, float hy, float beta) {
  if (nx < 1 || ny < 1 || !pad_modes) return PISO_ERR_INVALID_ARG;
  for (int q = 0; q < 4; ++q) if (pad_modes[q] < 0 || pad_modes[q] > 2) return PISO_ERR_INVALID_ARG;
  // a periodic pressure axis needs both sides periodic
  if ((pad_modes[0] == PAD_WRAP) != (pad_modes[1] == PAD_WRAP) || (pad_modes[2] == PAD_WRAP) != (pad_modes[3] == PAD_WRAP)) return PISO_ERR_INVALID_ARG;
  g.nx = nx; g.ny = ny; g.per_x = periodic_x; g.per_y = periodic_y;
  g.px_lo = pad_modes[0]; g.px_hi = pad_modes[1]; g.py_lo = pad_modes[2]; g.py_hi = pad_modes[3];
  g.dxdy = dxdy; g.hx = hx; g.hy = hy; g.beta = beta;
  return PISO_OK;
}

int face_forward_impl(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* p,
                      const float* accessible, const float* a_flat, const float* in0, const float* in1, const float* in2,
                      const uint8_t* dirichlet, float* out0, float* out1, piso_stream_t stream_, const piso_slab_t* slab) {
  GlueGeom g;
  if (make_geom(g, nx, ny, 0, 0, pad_modes, dxdy, hx, hy, beta) != PISO_OK || !p || !in0 || !out0 ||
      (mode == FACE_RHS && dirichlet && !in2) || (mode != FACE_RHS && !a_flat) || (mode == FACE_CORR1 && !out1) || (mode == FACE_FINAL && !in1)) {
    set_error_msg("piso_face_forward: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  PISO_SLAB_CHECK("piso_face_forward");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  const int grid = glue_grid(fw.count());
  if (mode == FACE_RHS) face_forward_kernel<FACE_RHS><<<grid, kBlock, 0, stream>>>(g, p, accessible, a_flat, in0, in1, in2, dirichlet, out0, out1, fw, M);
  else if (mode == FACE_CORR1) face_forward_kernel<FACE_CORR1><<<grid, kBlock, 0, stream>>>(g, p, accessible, a_flat, in0, in1, in2, dirichlet, out0, out1, fw, M);
  else if (mode == FACE_FINAL) face_forward_kernel<FACE_FINAL><<<grid, kBlock, 0, stream>>>(g, p, accessible, a_flat, in0, in1, in2, dirichlet, out0, out1, fw, M);
  else { set_error_msg("piso_face_forward: unknown mode"); return PISO_ERR_INVALID_ARG; }
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int face_backward_impl(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* accessible,
                       const float* a_flat, const uint8_t* dirichlet, const float* d_out0, const float* d_out1, float* d_in0,
                       float* d_in1, float* d_in2, float* d_p, piso_stream_t stream_, const piso_slab_t* slab) {
  GlueGeom g;
  if (make_geom(g, nx, ny, 0, 0, pad_modes, dxdy, hx, hy, beta) != PISO_OK || !d_out0 || !d_in0 || !d_p || (mode != FACE_RHS && !a_flat) ||
      (mode == FACE_FINAL && !d_in1)) {
    set_error_msg("piso_face_backward: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  PISO_SLAB_CHECK("piso_face_backward");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  const CellWin cw = cell_window(M);
  const int gf = glue_grid(fw.count()), gc = glue_grid(cw.n);
#define PISO_FACE_BWD(MD)                                                                                                   \
  do {                                                                                                                      \
    face_backward_kernel<MD><<<gf, kBlock, 0, stream>>>(g, a_flat, dirichlet, d_out0, d_out1, d_in0, d_in1, d_in2, fw, M);  \
    gradient_adjoint_kernel<MD><<<gc, kBlock, 0, stream>>>(g, accessible, a_flat, dirichlet, d_out0, d_out1, d_p, cw, M);    \
  } while (0)
  if (mode == FACE_RHS) PISO_FACE_BWD(FACE_RHS);
  else if (mode == FACE_CORR1) PISO_FACE_BWD(FACE_CORR1);
  else if (mode == FACE_FINAL) PISO_FACE_BWD(FACE_FINAL);
  else { set_error_msg("piso_face_backward: unknown mode"); return PISO_ERR_INVALID_ARG; }
#undef PISO_FACE_BWD
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int divergence_impl(const float* faces, float* div, int nx, int ny, float dxdy, float hx, float hy, piso_stream_t stream_, const piso_slab_t* slab) {
  if (!faces || !div || nx < 1 || ny < 1) { set_error_msg("piso_divergence: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_divergence");
  const RowMap M = make_row_map(slab, nx, ny);
  const CellWin cw = cell_window(M);
  divergence_kernel<<<glue_grid(cw.n), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(faces, div, nx, ny, dxdy, hx, hy, cw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int divergence_adjoint_impl(const float* d_div, float* d_faces, int nx, int ny, int periodic_x, int periodic_y, float dxdy, float hx, float hy,
                            piso_stream_t stream_, const piso_slab_t* slab) {
  if (!d_div || !d_faces || nx < 2 || ny < 2) { set_error_msg("piso_divergence_adjoint: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_divergence_adjoint");
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  divergence_adjoint_kernel<<<glue_grid(fw.count()), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(
      d_div, d_faces, nx, ny, periodic_x, periodic_y, dxdy, hx, hy, fw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int h_contribution_impl(const float* m_delta, const float* delta, const float* a_flat, float beta, float* h, float* h_over_bma, int nx, int ny,
                        piso_stream_t stream_, const piso_slab_t* slab) {
  if (!m_delta || !delta || !a_flat || !h || !h_over_bma || nx < 1 || ny < 1) { set_error_msg("piso_h_contribution: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_h_contribution");
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  h_kernel<<<glue_grid(fw.count()), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(m_delta, delta, a_flat, beta, h, h_over_bma, fw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int h_contribution_adjoint_impl(const float* d_h, const float* d_h_over_bma, const float* a_flat, float beta, float* d_m_delta, float* d_delta,
                                int nx, int ny, piso_stream_t stream_, const piso_slab_t* slab) {
  if (!d_h_over_bma || !a_flat || !d_m_delta || !d_delta || nx < 1 || ny < 1) { set_error_msg("piso_h_contribution_adjoint: invalid argument"); return PISO_ERR_INVALID_ARG; }
  PISO_SLAB_CHECK("piso_h_contribution_adjoint");
  const RowMap M = make_row_map(slab, nx, ny);
  const FaceWin fw = face_window(M);
  h_adjoint_kernel<<<glue_grid(fw.count()), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(d_h, d_h_over_bma, a_flat, beta, d_m_delta, d_delta, fw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}
#undef PISO_SLAB_CHECK

}  // namespace

extern "C" {

int piso_pad_velocity(const float* vel_flat, float* vel_pad, int nx, int ny, int periodic_x, int periodic_y, piso_stream_t stream) {
  return pad_velocity_impl(vel_flat, vel_pad, nx, ny, periodic_x, periodic_y, stream, nullptr);
}
int piso_pad_velocity_slab(const float* vel_flat, float* vel_pad, int nx, int ny, int periodic_x, int periodic_y, piso_stream_t stream,
                           const piso_slab_t* slab) {
  return pad_velocity_impl(vel_flat, vel_pad, nx, ny, periodic_x, periodic_y, stream, slab);
}
int piso_a0_vfirst(const float* a_flat, float* a0_vfirst, int nx, int ny, float beta, float dx_factor, piso_stream_t stream) {
  return a0_vfirst_impl(a_flat, a0_vfirst, nx, ny, beta, dx_factor, stream, nullptr);
}
int piso_a0_vfirst_slab(const float* a_flat, float* a0_vfirst, int nx, int ny, float beta, float dx_factor, piso_stream_t stream,
                        const piso_slab_t* slab) {
  return a0_vfirst_impl(a_flat, a0_vfirst, nx, ny, beta, dx_factor, stream, slab);
}
int piso_face_forward(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* p,
                      const float* accessible, const float* a_flat, const float* in0, const float* in1, const float* in2,
                      const uint8_t* dirichlet, float* out0, float* out1, piso_stream_t stream) {
  return face_forward_impl(mode, nx, ny, pad_modes, dxdy, hx, hy, beta, p, accessible, a_flat, in0, in1, in2, dirichlet, out0, out1, stream, nullptr);
}
int piso_face_forward_slab(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* p,
                           const float* accessible, const float* a_flat, const float* in0, const float* in1, const float* in2,
                           const uint8_t* dirichlet, float* out0, float* out1, piso_stream_t stream, const piso_slab_t* slab) {
  return face_forward_impl(mode, nx, ny, pad_modes, dxdy, hx, hy, beta, p, accessible, a_flat, in0, in1, in2, dirichlet, out0, out1, stream, slab);
}
int piso_face_backward(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* accessible,
                       const float* a_flat, const uint8_t* dirichlet, const float* d_out0, const float* d_out1, float* d_in0,
                       float* d_in1, float* d_in2, float* d_p, piso_stream_t stream) {
  return face_backward_impl(mode, nx, ny, pad_modes, dxdy, hx, hy, beta, accessible, a_flat, dirichlet, d_out0, d_out1, d_in0, d_in1, d_in2, d_p, stream, nullptr);
}
int piso_face_backward_slab(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* accessible,
                            const float* a_flat, const uint8_t* dirichlet, const float* d_out0, const float* d_out1, float* d_in0,
                            float* d_in1, float* d_in2, float* d_p, piso_stream_t stream, const piso_slab_t* slab) {
  return face_backward_impl(mode, nx, ny, pad_modes, dxdy, hx, hy, beta, accessible, a_flat, dirichlet, d_out0, d_out1, d_in0, d_in1, d_in2, d_p, stream, slab);
}
int piso_divergence(const float* faces, float* div, int nx, int ny, float dxdy, float hx, float hy, piso_stream_t stream) {
  return divergence_impl(faces, div, nx, ny, dxdy, hx, hy, stream, nullptr);
}
int piso_divergence_slab(const float* faces, float* div, int nx, int ny, float dxdy, float hx, float hy, piso_stream_t stream, const piso_slab_t* slab) {
  return divergence_impl(faces, div, nx, ny, dxdy, hx, hy, stream, slab);
}
int piso_divergence_adjoint(const float* d_div, float* d_faces, int nx, int ny, int periodic_x, int periodic_y, float dxdy, float hx, float hy,
                            piso_stream_t stream) {
  return divergence_adjoint_impl(d_div, d_faces, nx, ny, periodic_x, periodic_y, dxdy, hx, hy, stream, nullptr);
}
int piso_divergence_adjoint_slab(const float* d_div, float* d_faces, int nx, int ny, int periodic_x, int periodic_y, float dxdy, float hx, float hy,
                                 piso_stream_t stream, const piso_slab_t* slab) {
  return divergence_adjoint_impl(d_div, d_faces, nx, ny, periodic_x, periodic_y, dxdy, hx, hy, stream, slab);
}
int piso_h_contribution(const float* m_delta, const float* delta, const float* a_flat, float beta, float* h, float* h_over_bma, int nx, int ny,
                        piso_stream_t stream) {
  return h_contribution_impl(m_delta, delta, a_flat, beta, h, h_over_bma, nx, ny, stream, nullptr);
}
int piso_h_contribution_slab(const float* m_delta, const float* delta, const float* a_flat, float beta, float* h, float* h_over_bma, int nx, int ny,
                             piso_stream_t stream, const piso_slab_t* slab) {
  return h_contribution_impl(m_delta, delta, a_flat, beta, h, h_over_bma, nx, ny, stream, slab);
}
int piso_h_contribution_adjoint(const float* d_h, const float* d_h_over_bma, const float* a_flat, float beta, float* d_m_delta, float* d_delta,
                                int nx, int ny, piso_stream_t stream) {
  return h_contribution_adjoint_impl(d_h, d_h_over_bma, a_flat, beta, d_m_delta, d_delta, nx, ny, stream, nullptr);
}
int piso_h_contribution_adjoint_slab(const float* d_h, const float* d_h_over_bma, const float* a_flat, float beta, float* d_m_delta, float* d_delta,
                                     int nx, int ny, piso_stream_t stream, const piso_slab_t* slab) {
  return h_contribution_adjoint_impl(d_h, d_h_over_bma, a_flat, beta, d_m_delta, d_delta, nx, ny, stream, slab);
}

}  // extern "C"

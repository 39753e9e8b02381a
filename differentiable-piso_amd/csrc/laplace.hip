// Pressure matrix of the PISO correctors: 5 diagonals (-y, -x, diag, +x, +y) from the face coefficients A0 and the
// active / accessible cell masks.  Replaces calcPISOLaplaceMatrix + setUpData (CUDAsrc/laplace_op.cu.cc:79-189) and the
// index helpers gridIDXWithOffsetShifted / gridIDXForStaggered / CordsByRow (:16-76); the per-row `cords` scratch of the
// reference is not needed.  One thread per cell, one 40-byte (fp64) row store per thread.
#include "piso_common.h"

namespace piso {

template <typename T>
__global__ __launch_bounds__(kBlock) void laplace_kernel(int nx, int ny, const float* __restrict__ active,
                                                         const float* __restrict__ fluid,
                                                         const float* __restrict__ a0, T* __restrict__ L, CellWin cw, RowMap M) {
  const int ms = nx + 2, n_v = nx * (ny + 1);
  for (int row = cw.lo + blockIdx.x * kBlock + threadIdx.x; row < cw.lo + cw.n; row += gridDim.x * kBlock) {
    const int i = row % nx, j = row / nx;
    const int me = (i + 1) + (j + 1) * ms;
    // neighbour order of the reference loops (j = dim_size-1 .. 0): y-before, y-after, x-before, x-after
    const int nb[4] = {me - ms, me + ms, me - 1, me + 1};
    const int fa[4] = {i + j * nx, i + (j + 1) * nx, n_v + i + j * (nx + 1), n_v + i + 1 + j * (nx + 1)};
    const float am = active[M.mask(me)], fm = fluid[M.mask(me)];
    T dg = 0;
    T off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float an = active[M.mask(nb[k])], fn = fluid[M.mask(nb[k])];
      const float c = a0[M.face_vfirst(fa[k])];
      if (!(an == 0.0f && fn == 0.0f) && am != 0.0f) dg -= (T)c;                       // laplace_op.cu.cc:118-135
      off[k] = (an == 1.0f && fn == 1.0f && !(am == 0.0f && fm == 0.0f)) ? (T)c : (T)0;   // :140-177
    }
    T* o = L + (size_t)(row - cw.lo) * 5;                  // (a slab's matrix holds its OWNED rows; one GPU: cw.lo = 0)
    o[0] = off[0]; o[1] = off[2]; o[2] = dg; o[3] = off[3]; o[4] = off[1];
  }
}

template <typename T>
static int laplace_launch(int nx, int ny, const float* active, const float* fluid, const float* a0, T* L,
                          piso_stream_t stream, const piso_slab_t* slab = nullptr) {
  if (nx < 1 || ny < 1 || !active || !fluid || !a0 || !L || !slab_ok(slab, ny)) {
    set_error_msg("piso_laplace_matrix: invalid argument");
    return PISO_ERR_INVALID_ARG;
  }
  const RowMap M = make_row_map(slab, nx, ny);
  const CellWin cw = cell_window(M);                       // (slab-decomposed step: the rows of this rank's cells)
  const int g = grid_for((long long)cw.n, kBlock, 4096);
  laplace_kernel<T><<<g, kBlock, 0, static_cast<hipStream_t>(stream)>>>(nx, ny, active, fluid, a0, L, cw, M);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

}  // namespace piso

extern "C" {
int piso_laplace_matrix_f64(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                            double* laplace, piso_stream_t stream) {
  return piso::laplace_launch<double>(nx, ny, active, fluid, a0_vfirst, laplace, stream);
}
int piso_laplace_matrix_f32(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                            float* laplace, piso_stream_t stream) {
  return piso::laplace_launch<float>(nx, ny, active, fluid, a0_vfirst, laplace, stream);
}
int piso_laplace_matrix_f64_slab(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                                 double* laplace, piso_stream_t stream, const piso_slab_t* slab) {
  return piso::laplace_launch<double>(nx, ny, active, fluid, a0_vfirst, laplace, stream, slab);
}
int piso_laplace_matrix_f32_slab(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                                 float* laplace, piso_stream_t stream, const piso_slab_t* slab) {
  return piso::laplace_launch<float>(nx, ny, active, fluid, a0_vfirst, laplace, stream, slab);
}
}

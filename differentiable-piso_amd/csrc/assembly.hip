// Advection-diffusion matrix assembly on the staggered grid: both 5-point CSR matrices (u and v) in ONE launch.
// Replaces CentralDifferenceMatrixCsrKernelLauncher and its kernels calcDimPad / calcCsrRowPtrGpu / initWithZeros /
// calcAdvetionMatrixX / calcAdvetionMatrixY (CUDAsrc/central_difference_csr_op.cu.cc:148-664): no device->host copies of
// the dimensions, no stream/event creation, no row-pointer pre-pass -- each thread derives its row's CSR window from the
// closed form and ranks its (at most 5) entries by column.  ~72 B/row of HBM traffic, one row per thread.
#include "piso_common.h"

namespace piso {

struct AsmArgs {
  const float* vel_pad;
  float* val;
  int* col;
  int* rowptr;
  float* diag;
  const uint8_t* dirichlet;
  const float* active;
  const float* viscosity;
  const uint8_t* no_slip;
  int visc_is_field, nx, ny, per_x, per_y;
  float area[2], spacing[2], beta;
  int n_u, n_v, nnz_u;
  FaceWin fw;                 // the rows this launch assembles (slab-decomposed step: this rank's face rows)
  RowMap M;                   // where the rows of the arrays live (piso_common.h; one GPU: the identity)
  int pattern_only;           // slab set-up: columns and row pointers of every STORED row, nothing else
  int nnz_u_l;                // stored entries of the u matrix (= nnz_u on one GPU): where the v matrix starts in val / col
};

__global__ __launch_bounds__(kBlock) void assemble_kernel(AsmArgs a) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < a.fw.count(); w += gridDim.x * kBlock) {
    const int g = a.fw.map(w);
    const int comp = g >= a.n_u;                     // 0: u faces (nx+1, ny); 1: v faces (nx, ny+1)
    const int row = comp ? g - a.n_u : g;
    const int W = a.nx + (comp == 0), H = a.ny + (comp == 1);
    const int i = row % W, j = row / W;
    const int loc[2] = {i, j}, dims[2] = {W, H}, per[2] = {a.per_x, a.per_y}, stride[2] = {1, W};
    const RowMap& M = a.M;
    float* val = a.val + (comp ? a.nnz_u_l : 0);
    int* col = a.col + (comp ? a.nnz_u_l : 0);
    int* rp = a.rowptr + (comp ? M.n_u + 1 : 0);          // (M.n_u: the STORED u faces)

    // neighbour existence and column, order: (low_x, high_x, low_y, high_y); wrap skips the duplicate face in the
    // component's own direction (:259-264, :281-286)
    int exists[4], ncol[4];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int own = (d == comp);
      const int in_lo = loc[d] >= 1, in_hi = loc[d] <= dims[d] - 2;
      exists[2 * d] = in_lo || per[d];
      exists[2 * d + 1] = in_hi || per[d];
      ncol[2 * d] = in_lo ? row - stride[d] : row + stride[d] * (dims[d] - 1 - own);
      ncol[2 * d + 1] = in_hi ? row + stride[d] : row - stride[d] * (dims[d] - 1 - own);
    }
    const int end_g = csr_row_end(row, i, j, W, H, a.per_x, a.per_y);
    const int end = M.slot(comp, j, end_g);                // (whole-grid slot numbers -> stored ones; the identity on one GPU)
    const int start = end - (1 + exists[0] + exists[1] + exists[2] + exists[3]);
    const int lrow = M.frow(comp, row);
    rp[lrow + 1] = end;
    rp[lrow] = start;                                // (the same value the row before writes: a windowed launch has no row before its first)
    // (a windowed launch may own neither first row: the two segment starts and the end of the u segment are closed forms)
    if (w == 0) { a.rowptr[0] = 0; a.rowptr[M.n_u] = a.nnz_u_l; a.rowptr[M.n_u + 1] = 0; }
    // slot of an entry = start + number of existing entries with a smaller column (rows are stored column-sorted,
    // which is what the slot arithmetic at :176-210 produces)
    int slot[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      const int ce = (e < 4) ? ncol[e] : row;
      int s = start;
#pragma unroll
      for (int o = 0; o < 5; ++o) {
        if (o == e) continue;
        const int co = (o < 4) ? ncol[o] : row;
        const int eo = (o < 4) ? exists[o] : 1;
        s += (eo && co < ce) ? 1 : 0;
      }
      slot[e] = s;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (exists[e]) { col[slot[e]] = ncol[e]; }
    col[slot[4]] = row;
    if (a.pattern_only) continue;

    const int gl = M.face(g);                        // this face in the stored face vectors
    if (a.dirichlet[gl]) {                            // :214-238
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (exists[e]) val[slot[e]] = 0.f;
      val[slot[4]] = 1.f;
      a.diag[gl] = 0.f;
      continue;
    }

    // face fluxes of this row's control volume from the padded velocities (calcCellFluxesX/Y, :35-101)
    const int pad_stride[2] = {a.nx + 3, a.nx + 2};
    const int pad_offset[2] = {0, (a.nx + 3) * (a.ny + 2)};
    float flux[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int p = pad_offset[c] + (i + 1) + (j + 1) * pad_stride[c];
      const int back = (comp == 0) ? 1 : pad_stride[c];
      float h = a.vel_pad[M.pad(p)];
      flux[2 * c] = (float)(.5 * (h + a.vel_pad[M.pad(p - back)]) * a.area[c]);
      p += (c == 0) ? 1 : pad_stride[c];
      h = a.vel_pad[M.pad(p)];
      flux[2 * c + 1] = (float)(.5 * (h + a.vel_pad[M.pad(p - back)]) * a.area[c]);
    }

    const float nu = a.viscosity[a.visc_is_field ? gl : 0];
    const int ms = a.nx + 2;
    float dv = 0.f;
#pragma unroll
    for (int d = 1; d >= 0; --d) {                   // y first, then x, as the reference accumulates (:248-293)
      const int own = (d == comp);
      const float diff = nu * a.area[d] / a.spacing[d];
      {
        int off[2] = {0, 0};
        off[d] = -1;
        const int nb = M.mask((i + 1 + off[0]) + (j + 1 + off[1]) * ms);
        const int ns = a.no_slip ? a.no_slip[nb] : 0;
        const int open = (a.active[nb] == 1.0f) || ((loc[d] >= 1) && ns);
        const float v = (float)(flux[2 * d] * .5 + diff);
        if (exists[2 * d]) val[slot[2 * d]] = open ? v : 0.f;
        dv = (float)(dv + (flux[2 * d] * (2 - open) * .5 - diff * (open + (!own) * (1 - open) * ns * 2)));
      }
      {
        int off[2] = {0, 0};
        off[d] = 1 - own;
        const int nb = M.mask((i + 1 + off[0]) + (j + 1 + off[1]) * ms);
        const int ns = a.no_slip ? a.no_slip[nb] : 0;
        const int open = (a.active[nb] == 1.0f) || ((loc[d] <= dims[d] - 2) && ns);
        const float v = (float)(-flux[2 * d + 1] * .5 + diff);
        if (exists[2 * d + 1]) val[slot[2 * d + 1]] = open ? v : 0.f;
        dv = (float)(dv + (-flux[2 * d + 1] * (2 - open) * .5 - diff * (open + (!own) * (1 - open) * ns * 2)));
      }
    }
    val[slot[4]] = dv - a.beta;                      // :294
    a.diag[gl] = dv;                                 // :296
  }
}

}  // namespace piso

extern "C" {

void piso_csr_nnz(int nx, int ny, int periodic_x, int periodic_y, int* nnz_u, int* nnz_v) {
  // diffpiso/piso_tf.py:102-106
  const int px = periodic_x ? 1 : 0, py = periodic_y ? 1 : 0;
  if (nnz_u) *nnz_u = 5 * (nx + 1) * ny - 2 * ny * (1 - px) - 2 * (nx + 1) * (1 - py);
  if (nnz_v) *nnz_v = 5 * nx * (ny + 1) - 2 * (ny + 1) * (1 - px) - 2 * nx * (1 - py);
}

static int assemble_impl(const float* vel_pad, float* csr_val, int* csr_col, int* csr_rowptr, float* diag,
                         const uint8_t* dirichlet, const float* active, const float* viscosity, int viscosity_is_field,
                         int nx, int ny, int periodic_x, int periodic_y, float cell_area_x, float cell_area_y,
                         float spacing_x, float spacing_y, const uint8_t* no_slip, float beta, piso_stream_t stream,
                         const piso_slab_t* slab, int pattern_only) {
  using namespace piso;
  if (nx < 3 || ny < 3 || !csr_col || !csr_rowptr || (!pattern_only && (!vel_pad || !csr_val || !diag || !dirichlet || !active || !viscosity))) {
    set_error_msg("piso_assemble_csr: invalid argument (need nx, ny >= 3 and non-NULL arrays)");
    return PISO_ERR_INVALID_ARG;
  }
  if (!slab_ok(slab, ny) || (pattern_only && !slab)) { set_error_msg("piso_assemble_csr_slab: invalid slab"); return PISO_ERR_INVALID_ARG; }
  AsmArgs a;
  a.vel_pad = vel_pad; a.val = csr_val; a.col = csr_col; a.rowptr = csr_rowptr; a.diag = diag;
  a.dirichlet = dirichlet; a.active = active; a.viscosity = viscosity; a.no_slip = no_slip;
  a.visc_is_field = viscosity_is_field ? 1 : 0;
  a.nx = nx; a.ny = ny; a.per_x = periodic_x ? 1 : 0; a.per_y = periodic_y ? 1 : 0;
  a.area[0] = cell_area_x; a.area[1] = cell_area_y; a.spacing[0] = spacing_x; a.spacing[1] = spacing_y; a.beta = beta;
  a.n_u = (nx + 1) * ny; a.n_v = nx * (ny + 1);
  int nnz_v;
  piso_csr_nnz(nx, ny, a.per_x, a.per_y, &a.nnz_u, &nnz_v);
  a.M = make_row_map(slab, nx, ny, a.per_x, a.per_y);
  a.nnz_u_l = a.M.nnz_l[0];
  a.pattern_only = pattern_only ? 1 : 0;
  a.fw = face_window(a.M);
  if (pattern_only) {
    // every STORED face row, walked in two or four whole-grid intervals (the ring's seam cuts an interval in two); the launch is
    // a set-up step: one launch per interval pair keeps the kernel's window a FaceWin
    const RowMap& M = a.M;
    const int Wu = nx + 1, Wv = nx, Hu = ny, Hv = ny + 1;
    const int u_first = M.cb, u_n1 = (M.cb + M.cr <= Hu) ? M.cr : Hu - M.cb, u_n2 = M.cr - u_n1;
    const int v_first = M.vb, v_n1 = (M.vb + M.vr <= Hv) ? M.vr : Hv - M.vb, v_n2 = M.vr - v_n1;
    const FaceWin w1{u_first * Wu, u_n1 * Wu, a.n_u + v_first * Wv, v_n1 * Wv}, w2{0, u_n2 * Wu, a.n_u, v_n2 * Wv};
    for (const FaceWin& w : {w1, w2}) {
      if (w.count() == 0) continue;
      a.fw = w;
      assemble_kernel<<<grid_for((long long)w.count(), kBlock, 8192), kBlock, 0, static_cast<hipStream_t>(stream)>>>(a);
    }
    PISO_LAUNCH_CHECK();
    return PISO_OK;
  }
  const int g = grid_for((long long)a.fw.count(), kBlock, 8192);
  assemble_kernel<<<g, kBlock, 0, static_cast<hipStream_t>(stream)>>>(a);
  PISO_LAUNCH_CHECK();
  return PISO_OK;
}

int piso_assemble_csr(const float* vel_pad, float* csr_val, int* csr_col, int* csr_rowptr, float* diag,
                      const uint8_t* dirichlet, const float* active, const float* viscosity, int viscosity_is_field,
                      int nx, int ny, int periodic_x, int periodic_y, float cell_area_x, float cell_area_y,
                      float spacing_x, float spacing_y, const uint8_t* no_slip, float beta, piso_stream_t stream) {
  return assemble_impl(vel_pad, csr_val, csr_col, csr_rowptr, diag, dirichlet, active, viscosity, viscosity_is_field, nx, ny, periodic_x,
                       periodic_y, cell_area_x, cell_area_y, spacing_x, spacing_y, no_slip, beta, stream, nullptr, 0);
}
int piso_assemble_csr_slab(const float* vel_pad, float* csr_val, int* csr_col, int* csr_rowptr, float* diag,
                           const uint8_t* dirichlet, const float* active, const float* viscosity, int viscosity_is_field,
                           int nx, int ny, int periodic_x, int periodic_y, float cell_area_x, float cell_area_y,
                           float spacing_x, float spacing_y, const uint8_t* no_slip, float beta, piso_stream_t stream,
                           const piso_slab_t* slab, int pattern_only) {
  return assemble_impl(vel_pad, csr_val, csr_col, csr_rowptr, diag, dirichlet, active, viscosity, viscosity_is_field, nx, ny, periodic_x,
                       periodic_y, cell_area_x, cell_area_y, spacing_x, spacing_y, no_slip, beta, stream, slab, pattern_only);
}
int piso_slab_sizes(const piso_slab_t* slab, int nx, int ny, int periodic_x, int periodic_y, int* out8) {
  using namespace piso;
  if (!out8 || nx < 3 || ny < 3 || !slab_ok(slab, ny)) { set_error_msg("piso_slab_sizes: invalid argument"); return PISO_ERR_INVALID_ARG; }
  const RowMap M = make_row_map(slab, nx, ny, periodic_x ? 1 : 0, periodic_y ? 1 : 0);
  out8[0] = M.cr; out8[1] = M.vr; out8[2] = M.n_u; out8[3] = M.n_v; out8[4] = M.nnz_l[0]; out8[5] = M.nnz_l[1]; out8[6] = M.mr;
  out8[7] = M.pur * (nx + 3) + M.pvr * (nx + 2);
  return PISO_OK;
}
}

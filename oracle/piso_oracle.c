/*
 * piso_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the native algorithms on the differentiable-PISO hot path of
 * tum-pbs/differentiable-piso.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / reported CPU baseline.  The product path
 * (differentiable-piso_amd/) never links, imports or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - The reference's CUDA sources need the CUDA toolkit, cuBLAS, cuSPARSE, cuRAND and TensorFlow headers,
 *     none of which exist in this image => the reference is UNBUILDABLE here and no stand-in build is made.
 *   - The reference ships no golden vectors for these kernels.
 *   - assembly / Laplace / CG / BiCGStab below are therefore pinned by: the worked known-answer rows of
 *     SURVEY.md Appendix B, scipy.sparse direct solves (the reference's own cross-check pattern,
 *     diffpiso/linear_solver.py:39-44, diffpiso/piso_helpers.py:326-343) and dense numpy solves.
 *     cuSPARSE csrilu02/csrsv2/csrmv/csr2csc (CUDA 10.0, closed source) are restated from their published
 *     definition (ILU(0) in IKJ order on the CSR pattern, unit-lower / non-unit-upper substitution).
 *   - oracle_laplace_* (at A0 = 1) and the recurrence of oracle_cg_* (no shift, no restart, fixed iteration counts) ARE pinned
 *     against outputs of the reference's own Python: PhiFlow's sparse_pressure_matrix and conjugate_gradient, run here by
 *     tests/golden/make_golden_pressure.py (tests/test_oracle_pressure_phiflow.py: matrix entry for entry, iterates to 1e-11).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 * 2-D only (dimSize == 2), batch size 1 -- the only configuration any reference script uses.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------
 * Advection-diffusion matrix assembly (CentralDifferenceMatrixCsr)
 * ---------------------------------------------------------------------------------------------- */

/* CUDAsrc/central_difference_csr_op.cu.cc:472-505 (calcCsrRowPtrGpu, dimSize==2 branch) plus the
 * rowPtr[0]=0 memcpy at :617-618.  W,H = dims of this component's face array. */
static void csr_row_ptr(int* rp, int W, int H, int per_x, int per_y) {
  rp[0] = 0;
  for (int row = 0; row < W * H; ++row) {
    const int i = row % W, j = row / W;
    int r = (row + 1) * 5;
    r -= imin(j, 1) * (W * (1 - per_y));
    r -= ((1 - imin(j, 1)) + (1 + imax(j + 1 - H, -1))) * (i + 1) * (1 - per_y);
    r -= (j * 2 + 1 + (1 + imax(i + 1 - W, -1))) * (1 - per_x);
    rp[row + 1] = r;
  }
}

/* One component of calcAdvetionMatrixX / calcAdvetionMatrixY
 * (central_difference_csr_op.cu.cc:148-303 and :306-453; helper functions :25-33, :35-101, :132-146).
 * comp = 0 (u faces, dims (nx+1, ny)) or 1 (v faces, dims (nx, ny+1)).
 * vel_pad: [padded u (ny+2, nx+3)] followed by [padded v (ny+3, nx+2)], x fastest (piso_tf.py:93).
 * active/no_slip are indexed in the padded-centred (ny+2, nx+2) space.
 * The arithmetic keeps the reference's C types: float fluxes, double literals .5, float accumulator. */
static void assemble_component(int comp, const float* vel_pad, int nx, int ny, int per_x, int per_y,
                               const uint8_t* dirichlet, const float* active, const float* viscosity,
                               int visc_is_field, const float cell_area[2], const float spacing[2],
                               const uint8_t* no_slip, float beta, float* val, int* col, const int* rp,
                               float* diag) {
  const int W = nx + (comp == 0), H = ny + (comp == 1);
  const int dims[2] = {W, H};
  const int per[2] = {per_x, per_y};
  /* calcDimPad (:508-518) with padDepth == 1 everywhere (piso_tf.py:95) */
  const int pad_stride[2] = {nx + 1 + 2, nx + 2};                    /* row stride of padded u / padded v */
  const int pad_offset[2] = {0, (nx + 3) * (ny + 2)};                /* start of padded u / padded v      */
  const int mask_stride = nx + 2;                                    /* gridIDXpaddedCenteredMasks :143   */

  for (int row = 0; row < W * H; ++row) {
    int loc[2] = {row % W, row / W};                                 /* calcGridLocation :25-33 */
    int inb[4];                                                      /* domainBoundaryBool :166-172: 1 = NOT on that border */
    for (int d = 0; d < 2; ++d) {
      inb[2 * d] = imin(loc[d], 1);
      inb[2 * d + 1] = imin(dims[d] - 1 - loc[d], 1);
    }
    /* slot bookkeeping :176-210. idx order: (low_x, high_x, low_y, high_y, centre) */
    int idx[5], ordered[5];
    for (int d = 0; d < 2; ++d) {
      idx[2 * (1 - d)] = rp[row] + d;
      idx[2 * (1 - d) + 1] = rp[row] + 4 - d;
    }
    idx[4] = rp[row] + 2;
    for (int d = 1; d >= 0; --d) {
      idx[2 * d] += (d + 1) * 2 * (1 - inb[2 * d]) * per[d];
      idx[2 * d] += (1 - inb[2 * d + 1]) * per[d];
      idx[2 * d + 1] -= (d + 1) * 2 * (1 - inb[2 * d + 1]) * per[d];
      idx[2 * d + 1] -= (1 - inb[2 * d]) * per[d];
      for (int s = 0; s < d; ++s) {
        idx[2 * s] += (inb[2 * d] - inb[2 * d + 1]) * per[d];
        idx[2 * s + 1] += (inb[2 * d] - inb[2 * d + 1]) * per[d];
      }
      idx[4] += (inb[2 * d] - inb[2 * d + 1]) * per[d];
    }
    for (int i = 0; i < 5; ++i) ordered[idx[i] - rp[row]] = i;
    for (int d = 0; d < 4; ++d) {
      if (ordered[d] == 4) continue;
      for (int i = d + 1; i < 5; ++i)
        idx[ordered[i]] -= (1 - inb[ordered[d]]) * (1 - per[ordered[d] / 2]);
    }
    const int stride[2] = {1, W};                                    /* currentOffset after /= dimensions[d] */

    /* column indices: identical code for Dirichlet (:216-232) and regular rows (:259-264, :281-286) */
    for (int d = 1; d >= 0; --d) {
      const int own = (d == comp);
      if (inb[2 * d]) col[idx[2 * d]] = row - stride[d];
      else if (per[d]) col[idx[2 * d]] = row + stride[d] * (dims[d] - 1 - own);
      if (inb[2 * d + 1]) col[idx[2 * d + 1]] = row + stride[d];
      else if (per[d]) col[idx[2 * d + 1]] = row - stride[d] * (dims[d] - 1 - own);
    }
    col[idx[4]] = row;

    if (dirichlet[row]) {                                            /* :214-238 */
      val[idx[4]] = 1.f;
      diag[row] = 0.f;
      continue;
    }

    /* calcCellFluxesX/Y (:35-101): fluxes = (x_lo, x_hi, y_lo, y_hi) through the faces of this row's control volume */
    float flux[4];
    for (int c = 0; c < 2; ++c) {
      int p = pad_offset[c] + (loc[0] + 1) + (loc[1] + 1) * pad_stride[c];
      const int back = (comp == 0) ? 1 : pad_stride[c];              /* X: "-1" (:62), Y: "-dimPad[i*dimSize+1]" (:89) */
      float h = vel_pad[p];
      flux[2 * c] = (float)(.5 * (h + vel_pad[p - back]) * cell_area[c]);
      p += (c == 0) ? 1 : pad_stride[c];
      h = vel_pad[p];
      flux[2 * c + 1] = (float)(.5 * (h + vel_pad[p - back]) * cell_area[c]);
    }

    const float nu = viscosity[visc_is_field ? row : 0];
    float dv = 0.f;                                                  /* diagonalValue :246 */
    for (int d = 1; d >= 0; --d) {
      const int own = (d == comp);
      /* lower neighbour (:251-266) */
      int off[2] = {0, 0};
      off[d] = -1;
      int nb = (loc[0] + 1 + off[0]) + (loc[1] + 1 + off[1]) * mask_stride;
      int open = (active[nb] == 1.0f) || (inb[2 * d] && no_slip[nb]);
      if (open && (inb[2 * d] || per[d]))                            /* guard: see DESIGN.md "undefined in reference" */
        val[idx[2 * d]] = (float)(flux[2 * d] * .5 + nu * cell_area[d] / spacing[d]);
      dv = (float)(dv + (flux[2 * d] * (2 - open) * .5 -
                         nu * cell_area[d] / spacing[d] * (open + (!own) * (1 - open) * no_slip[nb] * 2)));
      /* upper neighbour (:273-288) */
      off[d] = 1 - own;
      nb = (loc[0] + 1 + off[0]) + (loc[1] + 1 + off[1]) * mask_stride;
      open = (active[nb] == 1.0f) || (inb[2 * d + 1] && no_slip[nb]);
      if (open && (inb[2 * d + 1] || per[d]))
        val[idx[2 * d + 1]] = (float)(-flux[2 * d + 1] * .5 + nu * cell_area[d] / spacing[d]);
      dv = (float)(dv + (-flux[2 * d + 1] * (2 - open) * .5 -
                         nu * cell_area[d] / spacing[d] * (open + (!own) * (1 - open) * no_slip[nb] * 2)));
    }
    val[idx[4]] = dv - beta;                                         /* :294 */
    diag[row] = dv;                                                  /* :296 */
  }
}

/* CentralDifferenceMatrixCsrKernelLauncher (central_difference_csr_op.cu.cc:543-664).
 * Outputs: val/col [nnz_u+nnz_v], rowptr [n_u+1 + n_v+1] (two 0-based segments), diag [n_u+n_v].
 * dirichlet, viscosity (if a field): u rows first, then v rows (piso_tf.py:30, :654). Returns nnz_u+nnz_v. */
ORACLE_API int oracle_assemble_csr(const float* vel_pad, int nx, int ny, int per_x, int per_y,
                                   const uint8_t* dirichlet, const float* active, const float* viscosity,
                                   int visc_is_field, const float* cell_area, const float* spacing,
                                   const uint8_t* no_slip, float beta, float* val, int* col, int* rowptr,
                                   float* diag) {
  const int n_u = (nx + 1) * ny, n_v = nx * (ny + 1);
  csr_row_ptr(rowptr, nx + 1, ny, per_x, per_y);
  csr_row_ptr(rowptr + n_u + 1, nx, ny + 1, per_x, per_y);
  const int nnz_u = rowptr[n_u], nnz_v = rowptr[n_u + 1 + n_v];
  memset(val, 0, sizeof(float) * (size_t)(nnz_u + nnz_v));          /* initWithZeros :627-628 */
  memset(col, 0, sizeof(int) * (size_t)(nnz_u + nnz_v));
  assemble_component(0, vel_pad, nx, ny, per_x, per_y, dirichlet, active, viscosity, visc_is_field, cell_area,
                     spacing, no_slip, beta, val, col, rowptr, diag);
  assemble_component(1, vel_pad, nx, ny, per_x, per_y, dirichlet + n_u, active,
                     viscosity + (visc_is_field ? n_u : 0), visc_is_field, cell_area, spacing, no_slip, beta,
                     val + nnz_u, col + nnz_u, rowptr + n_u + 1, diag + n_u);
  return nnz_u + nnz_v;
}

/* ------------------------------------------------------------------------------------------------
 * Pressure Laplacian (calcPISOLaplaceMatrix) and CG (LaunchPressureKernel)
 * ---------------------------------------------------------------------------------------------- */

/* CUDAsrc/laplace_op.cu.cc:79-179 (+ setUpData :181-189, index helpers :16-62).
 * L: [N,5] rows (-y, -x, diag, +x, +y).  a0: flat staggered, V FIRST ((ny+1)*nx values of v, then ny*(nx+1) of u;
 * piso_cuda_pressure_solver.py:70).  active/fluid: padded-centred (ny+2, nx+2).  T = double or float. */
#define DEFINE_LAPLACE(NAME, T)                                                                          \
  ORACLE_API void NAME(int nx, int ny, const float* active, const float* fluid, const float* a0, T* L) { \
    const int N = nx * ny, ms = nx + 2, n_v = nx * (ny + 1);                                             \
    for (int k = 0; k < 5 * N; ++k) L[k] = 0;                                                            \
    for (int row = 0; row < N; ++row) {                                                                  \
      const int i = row % nx, j = row / nx;                                                              \
      const int me = (i + 1) + (j + 1) * ms;                                                             \
      /* neighbour mask indices and staggered A0 indices, order: y-before, y-after, x-before, x-after */ \
      const int nb[4] = {me - ms, me + ms, me - 1, me + 1};                                              \
      const int fa[4] = {i + j * nx, i + (j + 1) * nx, n_v + i + j * (nx + 1), n_v + i + 1 + j * (nx + 1)}; \
      T dg = 0.0f;                                                                                       \
      for (int k = 0; k < 4; ++k)                                                                        \
        if (!(active[nb[k]] == 0.0f && fluid[nb[k]] == 0.0f) && active[me] != 0.0f) dg -= a0[fa[k]];     \
      const int slot[4] = {0, 4, 1, 3};                                                                  \
      for (int k = 0; k < 4; ++k)                                                                        \
        if (active[nb[k]] == 1 && fluid[nb[k]] == 1 && !(active[me] == 0 && fluid[me] == 0))             \
          L[row * 5 + slot[k]] = a0[fa[k]];                                                              \
      L[row * 5 + 2] = dg;                                                                               \
    }                                                                                                    \
  }
DEFINE_LAPLACE(oracle_laplace_f64, double)
DEFINE_LAPLACE(oracle_laplace_f32, float)

/* CUDAsrc/pressure_solve_op.cu.cc:140-418 (LaunchPressureKernel), kernels :57-133, batch 1, randomized_restarts 0.
 * Solves (L + c 1 1^T) x = b by plain CG, x0 = 0 (init_with_zeros, piso_cuda_pressure_solver.py:95).
 * Returns the iteration count the reference writes to iterations_gpu. */
#define DEFINE_CG(NAME, T)                                                                                \
  static void NAME##_apply(int nx, int ny, int per_x, int per_y, const T* L, const T* p, T* z, T vsum) {  \
    const int N = nx * ny;                                                                                \
    /* calcDiagonalOffsets :117-133 */                                                                    \
    const int off[5] = {-nx, -1, 0, 1, nx};                                                               \
    const int poff[5] = {N * per_y, nx * per_x, 0, -nx * per_x, -N * per_y};                              \
    for (int row = 0; row < N; ++row) { /* calcZ_v4 :57-92 */                                             \
      const int i = row % nx, j = row / nx;                                                               \
      const int onb[5] = {j == 0, i == 0, 0, i == nx - 1, j == ny - 1};                                   \
      T tmp = 0;                                                                                          \
      for (int s = 0; s < 5; ++s) {                                                                       \
        const T l = L[row * 5 + s];                                                                       \
        const int ci = row + off[s] + onb[s] * poff[s];                                                   \
        tmp += l * p[ci * (l != 0.0)];                                                                    \
      }                                                                                                   \
      z[row] = tmp + vsum;                                                                                \
    }                                                                                                     \
  }                                                                                                       \
  static T NAME##_dot(int N, const T* a, const T* b) {                                                    \
    T s = 0;                                                                                              \
    for (int k = 0; k < N; ++k) s += a[k] * b[k];                                                         \
    return s;                                                                                             \
  }                                                                                                       \
  static T NAME##_sum(int N, const T* a) {                                                                \
    T s = 0;                                                                                              \
    for (int k = 0; k < N; ++k) s += a[k];                                                                \
    return s;                                                                                             \
  }                                                                                                       \
  ORACLE_API int NAME(int nx, int ny, int per_x, int per_y, const T* L, const T* b, T* x, T* p, T* z,     \
                      T* r, float accuracy, int max_iterations, int rank_deficient, int reset_steps) {    \
    const int N = nx * ny;                                                                                \
    T c = 0; /* vectorSum_scaling :161-168: 0.1/N * asum(diag) */                                         \
    if (rank_deficient) {                                                                                 \
      for (int k = 0; k < N; ++k) c += fabs((double)L[k * 5 + 2]);                                        \
      c *= .1 / N;                                                                                        \
    }                                                                                                     \
    for (int k = 0; k < N; ++k) x[k] = 0; /* :186-190 */                                                  \
    NAME##_apply(nx, ny, per_x, per_y, L, x, z, rank_deficient ? c * NAME##_sum(N, x) : 0);               \
    int flag_dev = 0, flag_host = 0; /* threshold_reached (device) / threshold_reached_cpu */             \
    for (int k = 0; k < N; ++k) p[k] = r[k] = b[k] - z[k]; /* initVariablesWithGuess :104-114 */          \
    flag_dev = 0;                                                                                         \
    flag_host = flag_dev; /* :244 */                                                                      \
    int checker = 1, iterations = 0;                                                                      \
    for (; iterations < max_iterations; iterations++) { /* :257-357 */                                    \
      if ((iterations + 1) % reset_steps == 0) { /* residual reset :260-274 */                            \
        NAME##_apply(nx, ny, per_x, per_y, L, x, z, rank_deficient ? c * NAME##_sum(N, x) : 0);           \
        for (int k = 0; k < N; ++k) p[k] = r[k] = b[k] - z[k];                                            \
        flag_dev = 0;                                                                                     \
      }                                                                                                   \
      NAME##_apply(nx, ny, per_x, per_y, L, p, z, rank_deficient ? c * NAME##_sum(N, p) : 0);             \
      const T p_r = NAME##_dot(N, p, r), p_z = NAME##_dot(N, p, z);                                       \
      T alpha = 0.;                                                                                       \
      if (fabs((double)p_z) > 0.) alpha = p_r / p_z; /* :301-302 */                                       \
      for (int k = 0; k < N; ++k) x[k] += alpha * p[k];                                                   \
      for (int k = 0; k < N; ++k) r[k] += -alpha * z[k];                                                  \
      if (checker % 5 == 0) { /* :312-335 */                                                              \
        for (int k = 0; k < N; ++k)                                                                       \
          if (fabs((double)r[k]) >= accuracy) { flag_dev = 0; break; } /* checkResiduum :94-102 */        \
        flag_host = flag_dev;                                                                             \
        if (flag_host) { iterations++; break; }                                                           \
        flag_dev = 1; /* cudaMemset(threshold_reached, 1) :334 */                                         \
      }                                                                                                   \
      checker++;                                                                                          \
      const T r_z = NAME##_dot(N, r, z);                                                                  \
      const T beta = -r_z / p_z; /* :351-352 */                                                           \
      for (int k = 0; k < N; ++k) p[k] = beta * p[k] + r[k];                                              \
    }                                                                                                     \
    return iterations;                                                                                    \
  }
DEFINE_CG(oracle_cg_f64, double)
DEFINE_CG(oracle_cg_f32, float)

/* ------------------------------------------------------------------------------------------------
 * General CSR helpers (cuSPARSE stand-ins restated from their published definitions)
 * ---------------------------------------------------------------------------------------------- */

/* cusparse?csr2csc (multi_bicgstab_ilu_linear_solve_op.cu.cc:117): explicit transpose, columns sorted */
#define DEFINE_CSR(T, S)                                                                                   \
  ORACLE_API void oracle_csr_transpose_##S(int n, const T* val, const int* rp, const int* col, T* tval,    \
                                           int* trp, int* tcol) {                                          \
    const int nnz = rp[n];                                                                                 \
    for (int k = 0; k <= n; ++k) trp[k] = 0;                                                               \
    for (int k = 0; k < nnz; ++k) trp[col[k] + 1]++;                                                       \
    for (int k = 0; k < n; ++k) trp[k + 1] += trp[k];                                                      \
    int* fill = (int*)malloc(sizeof(int) * (size_t)n);                                                     \
    memcpy(fill, trp, sizeof(int) * (size_t)n);                                                            \
    for (int r = 0; r < n; ++r)                                                                            \
      for (int k = rp[r]; k < rp[r + 1]; ++k) {                                                            \
        const int q = fill[col[k]]++;                                                                      \
        tval[q] = val[k];                                                                                  \
        tcol[q] = r;                                                                                       \
      }                                                                                                    \
    free(fill);                                                                                            \
  }                                                                                                        \
  /* cusparseCsrmvEx, non-transpose, alpha=1, beta=0 (:269-277) */                                         \
  ORACLE_API void oracle_csr_spmv_##S(int n, const T* val, const int* rp, const int* col, const T* x, T* y) { \
    for (int r = 0; r < n; ++r) {                                                                          \
      T s = 0;                                                                                             \
      for (int k = rp[r]; k < rp[r + 1]; ++k) s += val[k] * x[col[k]];                                     \
      y[r] = s;                                                                                            \
    }                                                                                                      \
  }                                                                                                        \
  /* csrilu02 (:193-218): ILU(0), IKJ order, in place on `lu` (a copy of the values).                      \
   * keep[k]==0 drops entry k from the preconditioner (treated as structurally absent); keep==NULL keeps   \
   * everything == the reference.  Returns 0 or 1+row of a zero pivot. */                                  \
  ORACLE_API int oracle_ilu0_##S(int n, T* lu, const int* rp, const int* col, const uint8_t* keep) {       \
    int* pos = (int*)malloc(sizeof(int) * (size_t)n);                                                      \
    int* dpos = (int*)malloc(sizeof(int) * (size_t)n);                                                     \
    for (int k = 0; k < n; ++k) pos[k] = -1;                                                               \
    int bad = 0;                                                                                           \
    for (int i = 0; i < n; ++i) {                                                                          \
      dpos[i] = -1;                                                                                        \
      for (int k = rp[i]; k < rp[i + 1]; ++k)                                                              \
        if (!keep || keep[k]) { pos[col[k]] = k; if (col[k] == i) dpos[i] = k; }                           \
      for (int k = rp[i]; k < rp[i + 1]; ++k) {                                                            \
        const int m = col[k];                                                                              \
        if (m >= i || (keep && !keep[k])) continue;                                                        \
        lu[k] /= lu[dpos[m]];                                                                              \
        for (int q = rp[m]; q < rp[m + 1]; ++q) {                                                          \
          if (col[q] <= m || (keep && !keep[q])) continue;                                                 \
          const int t = pos[col[q]];                                                                       \
          if (t >= 0) lu[t] -= lu[k] * lu[q];                                                              \
        }                                                                                                  \
      }                                                                                                    \
      if (dpos[i] < 0 || lu[dpos[i]] == 0) { if (!bad) bad = i + 1; }                                      \
      for (int k = rp[i]; k < rp[i + 1]; ++k) pos[col[k]] = -1;                                            \
    }                                                                                                      \
    free(pos);                                                                                             \
    free(dpos);                                                                                            \
    return bad;                                                                                            \
  }                                                                                                        \
  /* csrsv2 with descrL = lower/unit then descrU = upper/non-unit (:163-173, :321-327): out = U^-1 L^-1 in */ \
  ORACLE_API void oracle_ilu_apply_##S(int n, const T* lu, const int* rp, const int* col,                  \
                                       const uint8_t* keep, const T* in, T* tmp, T* out) {                 \
    for (int i = 0; i < n; ++i) {                                                                          \
      T s = in[i];                                                                                         \
      for (int k = rp[i]; k < rp[i + 1]; ++k)                                                              \
        if (col[k] < i && (!keep || keep[k])) s -= lu[k] * tmp[col[k]];                                    \
      tmp[i] = s;                                                                                          \
    }                                                                                                      \
    for (int i = n - 1; i >= 0; --i) {                                                                     \
      T s = tmp[i], d = 1;                                                                                 \
      for (int k = rp[i]; k < rp[i + 1]; ++k) {                                                            \
        if (keep && !keep[k]) continue;                                                                    \
        if (col[k] > i) s -= lu[k] * out[col[k]];                                                          \
        else if (col[k] == i) d = lu[k];                                                                   \
      }                                                                                                    \
      out[i] = s / d;                                                                                      \
    }                                                                                                      \
  }
DEFINE_CSR(float, f32)
DEFINE_CSR(double, f64)

/* BicgstabIluLinearSolveLauncher[Double] (multi_bicgstab_ilu_linear_solve_op.cu.cc:85-453 / :540-910),
 * one component.  x0 = initial guess, result in x.  warning[0] is set on NaN input norms (:245-256).
 * keep: optional preconditioner drop mask on the entries of the matrix actually factorised (the transposed
 * one when transpose != 0); NULL == reference.  Returns the total iteration count (it_count). */
#define DEFINE_BICG(T, S, SQRT)                                                                            \
  static T bicg_dot_##S(int n, const T* a, const T* b) {                                                   \
    T s = 0;                                                                                               \
    for (int k = 0; k < n; ++k) s += a[k] * b[k];                                                          \
    return s;                                                                                              \
  }                                                                                                        \
  ORACLE_API int oracle_bicgstab_ilu_##S(int n, const T* val_in, const int* rp_in, const int* col_in,      \
                                         const T* rhs, const T* x0, T* x, float tol, int max_it,           \
                                         int transpose, const uint8_t* keep, uint8_t* warning) {           \
    const int nnz = rp_in[n];                                                                              \
    T* val = (T*)malloc(sizeof(T) * (size_t)nnz);                                                          \
    int* rp = (int*)malloc(sizeof(int) * (size_t)(n + 1));                                                 \
    int* col = (int*)malloc(sizeof(int) * (size_t)nnz);                                                    \
    if (transpose) oracle_csr_transpose_##S(n, val_in, rp_in, col_in, val, rp, col); /* :113-134 */        \
    else {                                                                                                 \
      memcpy(val, val_in, sizeof(T) * (size_t)nnz);                                                        \
      memcpy(rp, rp_in, sizeof(int) * (size_t)(n + 1));                                                    \
      memcpy(col, col_in, sizeof(int) * (size_t)nnz);                                                      \
    }                                                                                                      \
    T* lu = (T*)malloc(sizeof(T) * (size_t)nnz);                                                           \
    memcpy(lu, val, sizeof(T) * (size_t)nnz); /* :137 */                                                   \
    oracle_ilu0_##S(n, lu, rp, col, keep);                                                                 \
    T* w = (T*)calloc((size_t)n * 8, sizeof(T));                                                           \
    T *p = w, *p_hat = w + n, *r = w + 2 * n, *rh = w + 3 * n, *v = w + 4 * n, *t = w + 5 * n,             \
      *z = w + 6 * n, *s_hat = w + 7 * n;                                                                  \
    T alpha = 1, rho = 1, rhop = 1, omega = 1, beta, nrm_r = 0;                                            \
    int it_count = 0;                                                                                      \
    { /* NaN guard :238-256 (norms of x0, matrix values, rhs) */                                           \
      T a = SQRT(bicg_dot_##S(n, x0, x0)), m = SQRT(bicg_dot_##S(nnz, val, val)),                          \
        b = SQRT(bicg_dot_##S(n, rhs, rhs));                                                               \
      if (isnan(a) || isnan(m) || isnan(b)) warning[0] = 1;                                                \
    }                                                                                                      \
    memcpy(x, x0, sizeof(T) * (size_t)n); /* :261 */                                                       \
    for (int restart = 0; restart < 2; restart++) { /* :263-408 */                                         \
      oracle_csr_spmv_##S(n, val, rp, col, x, r);                                                          \
      for (int k = 0; k < n; ++k) r[k] = rhs[k] - r[k];                                                    \
      nrm_r = SQRT(bicg_dot_##S(n, r, r));                                                                 \
      if (nrm_r < tol) break; /* goto endofloop :290-292 */                                                \
      for (int k = 0; k < n; ++k) { rh[k] = r[k]; v[k] = 0; p[k] = 0; }                                    \
      for (int i = 0; i < max_it; i++) {                                                                   \
        it_count++;                                                                                        \
        rhop = rho;                                                                                        \
        rho = bicg_dot_##S(n, r, rh);                                                                      \
        beta = (rho / rhop) * (alpha / omega);                                                             \
        for (int k = 0; k < n; ++k) p[k] = (p[k] - omega * v[k]) * beta + r[k]; /* axpy, scal, axpy :316-318 */ \
        oracle_ilu_apply_##S(n, lu, rp, col, keep, p, z, p_hat);                                           \
        oracle_csr_spmv_##S(n, val, rp, col, p_hat, v);                                                    \
        alpha = rho / bicg_dot_##S(n, rh, v);                                                              \
        for (int k = 0; k < n; ++k) x[k] += alpha * p_hat[k];                                              \
        for (int k = 0; k < n; ++k) r[k] -= alpha * v[k];                                                  \
        nrm_r = SQRT(bicg_dot_##S(n, r, r));                                                               \
        if (nrm_r < tol) break;                                                                            \
        oracle_ilu_apply_##S(n, lu, rp, col, keep, r, z, s_hat);                                           \
        oracle_csr_spmv_##S(n, val, rp, col, s_hat, t);                                                    \
        omega = bicg_dot_##S(n, t, r) / bicg_dot_##S(n, t, t);                                             \
        for (int k = 0; k < n; ++k) x[k] += omega * s_hat[k];                                              \
        for (int k = 0; k < n; ++k) r[k] -= omega * t[k];                                                  \
        nrm_r = SQRT(bicg_dot_##S(n, r, r));                                                               \
        if (nrm_r < tol) break;                                                                            \
      }                                                                                                    \
      if (nrm_r > tol * 100 || isnan(nrm_r)) { /* :392-407: zero the solution, retry once */               \
        for (int k = 0; k < n; ++k) x[k] = 0;                                                              \
      } else break;                                                                                        \
    }                                                                                                      \
    free(w); free(lu); free(val); free(rp); free(col);                                                     \
    return it_count;                                                                                       \
  }
DEFINE_BICG(float, f32, sqrtf)
DEFINE_BICG(double, f64, sqrt)

/* Preconditioner drop mask of the MI355X engine (DESIGN.md "structured block ILU0"): keep an entry iff it is the
 * diagonal or a geometric near neighbour (same face row: col==row+-1, adjacent face row: col==row+-W) AND row and
 * column lie in the same band of `band_rows` consecutive face rows.  Not part of the reference; used so that the
 * HIP solver can be compared trajectory-by-trajectory with the oracle running the same preconditioner. */
ORACLE_API void oracle_band_keep_mask(int W, int H, int band_rows, const int* rp, const int* col, uint8_t* keep) {
  for (int row = 0; row < W * H; ++row) {
    const int i = row % W, j = row / W;
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
      const int c = col[k], ci = c % W, cj = c / W;
      int near = (c == row) || (cj == j && (ci == i - 1 || ci == i + 1)) || (ci == i && (cj == j - 1 || cj == j + 1));
      keep[k] = (uint8_t)(near && (cj / band_rows == j / band_rows));
    }
  }
}

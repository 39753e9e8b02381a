/*
 * piso_hip.h -- C ABI of libpiso_hip.so, the MI355X (gfx950) native library behind the differentiable-PISO hot path.
 *
 * Drop-in boundary: each entry point replaces one native launcher of tum-pbs/differentiable-piso (the functions the
 * reference's TensorFlow OpKernel shells call after unpacking their tensors).  Differences from the reference launchers,
 * all deliberate:
 *   - every array argument is a DEVICE pointer owned by the caller; small scalars (sizes, tolerances, flags) are passed
 *     BY VALUE instead of as device arrays that the launcher copies back to the host;
 *   - an explicit HIP stream (hipStream_t passed as void*; NULL = the null stream);
 *   - scratch memory is a caller-provided workspace (query *_workspace_bytes first); nothing is hipMalloc'ed per call;
 *   - an int status is returned (PISO_OK == 0) instead of exit(1)/assert(0);
 *   - outputs are separate arrays (no input buffer is overwritten in place).
 * Numerical failure handling mirrors the reference: non-convergence of BiCGStab yields a zero solution after one restart,
 * NaN input sets warning[0] = 1; the pressure CG reports its iteration count.
 *
 * Layout conventions (identical to the reference):
 *   nx, ny          cell resolution; x is the fastest index everywhere
 *   u faces         [ny][nx+1]     v faces [ny+1][nx]      "u-first" flat vector = u followed by v
 *   padded u        [ny+2][nx+3]   padded v [ny+3][nx+2]   (custom_padded, diffpiso/piso_helpers.py:35-55)
 *   cell masks      [ny+2][nx+2]   (one ghost ring)        pressure / divergence [ny][nx]
 *   pressure matrix [ny*nx][5]     row = (-y, -x, diag, +x, +y)
 *   A0 (face coefficient of the pressure matrix) flat "v-first": v faces followed by u faces
 */
#ifndef PISO_HIP_H
#define PISO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PISO_OK 0
#define PISO_ERR_INVALID_ARG 1       /* bad size / NULL pointer / workspace too small */
#define PISO_ERR_HIP 2               /* a HIP runtime call failed (piso_last_error_string() has the text) */
#define PISO_ERR_UNSUPPORTED_PATTERN 3 /* CSR input is not a 5-point staggered-grid matrix */
#define PISO_ERR_NO_DEVICE 4
#define PISO_ERR_NEEDS_HOST 5        /* piso_cg_solve_async_*: this grid is solved with the host in the loop, call piso_cg_solve_* */

typedef void* piso_stream_t;

/* One rank's part of a grid cut into y-slabs (the slab-decomposed STEP; no counterpart in the reference).  The *_slab
 * twin of an entry point takes the same arguments as the entry point itself - nx, ny and every index rule are the WHOLE grid's -
 * plus this struct; its arrays hold the rank's rows only (LOCAL storage; round 5 - there is no process-wide row window any more):
 *   cell arrays, u faces   the ring rows [row_begin - 2, row_end + 2) (mod ny_global), row_end - row_begin + 4 of them
 *   v faces                the ring rows [row_begin - 3, row_end + 3) of the ring v[0] .. v[ny - 1], v[ny] (the duplicate row),
 *                          row_end - row_begin + 6 of them; flat face vectors: the stored u rows followed by the stored v rows
 *                          (v-first vectors: v then u)
 *   padded cell masks      rows [row_begin, row_begin + (row_end - row_begin) + 3) of the [ny + 2] padded rows (no ring)
 *   padded velocities      padded u rows [row_begin, row_end + 2), then padded v rows [row_begin, row_end + 3)
 *   CSR                    the rows of the stored face rows (u rows, then v rows) in stored order; row pointers
 *                          [stored u rows (nx + 1) + 1][stored v rows nx + 1] are offsets into the stored value / column arrays of
 *                          each component; column indices keep the whole grid's numbering (component-local row numbers)
 *   pressure matrix, CG    the owned rows only ([row_end - row_begin][nx][5]; what piso_cg_solve_slab_* takes)
 * A launch writes the OWNED rows (cells / u rows [row_begin, row_end), v rows [row_begin, row_end + owns_last_face_row)) and reads
 * the stored rows around them, which the caller fills with piso_comm_exchange (element segments of the stored arrays).
 * Constraints: 4 <= row_end - row_begin <= ny_global - 6. */
typedef struct piso_slab {
  int ny_global;              /* cell rows of the whole grid (must equal the entry point's ny) */
  int row_begin, row_end;     /* owned cell rows */
  int owns_last_face_row;     /* the last slab also owns the duplicate face row v[ny] */
} piso_slab_t;

/* Library / build information. */
const char* piso_version(void);
const char* piso_last_error_string(void);
/* Number of visible HIP devices (0 if none; never initialises a context). */
int piso_device_count(void);
/* Tuning / test knobs (no counterpart in the reference).  NONE of them changes what is computed: a knob picks between implementations
 * that return bitwise the same result (which kernel instance runs, how operands are staged, launch shapes) or switches a check / a
 * measurement aid on and off.  Each knob `name` takes its default ONCE, at library load, from the environment variable
 * PISO_<NAME IN UPPER CASE>; -1 = not set (automatic).  The store is process-wide and atomic; every entry point that reads knobs copies
 * ALL of them when it is entered and works on that snapshot, so a piso_set_option() from another thread (a test flipping a knob while an
 * autograd backward thread is inside a solve) never changes a decision in the middle of a call.  The list, with what each one selects,
 * is differentiable-piso_amd/csrc/options.h; the ones callers outside the tests use:
 *   cg_persist (0 forbid / 1 force the persistent CG kernel; default by grid size), cg_persist_r (2|4|16 rows per region),
 *   cg_segment (iterations per persistent launch), cg_verify (0: skip the true-residual check of persistent solves),
 *   slab_force (1: a communicator of ONE rank still runs the slab code paths - a ring with itself),
 *   bicg_fold / bicg_sweep_lds / bicg_fuse_p (0: the un-fused forms of the BiCGStab stages), conv_lds (0: closure convolutions read
 *   their operands straight from L2). */
int piso_set_option(const char* name, int value);
int piso_get_option(const char* name, int* value_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Advection-diffusion matrix assembly.
 * Replaces CentralDifferenceMatrixCsrKernelLauncher (CUDAsrc/central_difference_csr_op.cc:33-36,
 * CUDAsrc/central_difference_csr_op.cu.cc:543-664; kernels :148-453, :472-505).
 *   vel_pad      [(ny+2)(nx+3) + (ny+3)(nx+2)] padded u then padded v
 *   csr_val/col  [nnz_u + nnz_v]; csr_rowptr [n_u+1 + n_v+1] (two 0-based segments); diag [n_u + n_v] (the "A" array)
 *   dirichlet    [n_u + n_v] bytes (u first); active [(ny+2)(nx+2)]; no_slip [(ny+2)(nx+2)] bytes or NULL
 *   viscosity    1 value, or n_u+n_v values when viscosity_is_field != 0
 *   cell_area_x/y = area of the x-/y-normal face (dy, dx); spacing_x/y = (dx, dy); beta = dx*dy/dt
 * nnz_u/nnz_v follow diffpiso/piso_tf.py:102-106; piso_csr_nnz() returns them.
 * ------------------------------------------------------------------------------------------------------------- */
void piso_csr_nnz(int nx, int ny, int periodic_x, int periodic_y, int* nnz_u, int* nnz_v);

int piso_assemble_csr(const float* vel_pad, float* csr_val, int* csr_col, int* csr_rowptr, float* diag,
                      const uint8_t* dirichlet, const float* active, const float* viscosity, int viscosity_is_field,
                      int nx, int ny, int periodic_x, int periodic_y, float cell_area_x, float cell_area_y,
                      float spacing_x, float spacing_y, const uint8_t* no_slip, float beta, piso_stream_t stream);
/* one rank's rows (piso_slab_t).  pattern_only != 0: column indices and row pointers of ALL stored rows (owned and halo: the pattern is
 * geometry), no values - called once per set-up; pattern_only == 0: values, columns, row pointers and diag of the OWNED rows. */
int piso_assemble_csr_slab(const float* vel_pad, float* csr_val, int* csr_col, int* csr_rowptr, float* diag,
                           const uint8_t* dirichlet, const float* active, const float* viscosity, int viscosity_is_field,
                           int nx, int ny, int periodic_x, int periodic_y, float cell_area_x, float cell_area_y,
                           float spacing_x, float spacing_y, const uint8_t* no_slip, float beta, piso_stream_t stream,
                           const piso_slab_t* slab, int pattern_only);
/* sizes of a rank's stored arrays (piso_slab_t): out8 = {stored u rows, stored v rows, stored u faces, stored v faces, stored CSR
 * entries of the u matrix, of the v matrix, stored mask rows, elements of the stored padded velocities} */
int piso_slab_sizes(const piso_slab_t* slab, int nx, int ny, int periodic_x, int periodic_y, int* out8);

/* ---------------------------------------------------------------------------------------------------------------
 * ILU(0)-preconditioned BiCGStab on the u and v matrices (both components advance in the same launches).
 * Replaces MultiBicgstabIluLinearSolveLauncher (CUDAsrc/multi_bicgstab_ilu_linear_solve_op.cc:50-58,
 * .cu.cc:455-531 float / :912-988 double; per-component algorithm :85-453 / :540-910).
 *   csr_*      the concatenated two-matrix CSR produced by piso_assemble_csr (values possibly negated by the caller)
 *   rhs, x0    [n_u + n_v];  x_out [n_u + n_v]
 *   tol        absolute ||r||_2 tolerance; max_it per restart; transpose: bit 0 = solve with A^T (adjoint), bit 1 = the system matrix
 *              is -csr_val (the reference hands the op `-matrix_values`, piso_tf.py:41: the sign is applied where the values are read);
 *              any other bit set: PISO_ERR_INVALID_ARG (a caller meaning "non-zero = transpose" with 2 or -1 must not get another system)
 *   band_rows  rows of faces per preconditioner block: < 0  = one block (global structured ILU0),
 *              0 = automatic (8 rows from ny = 2048 on, 4 from 1024, 2 from 256, 8 below), > 0 = that many.  See DESIGN.md "structured block ILU0".
 *   warning    device byte, set to 1 on NaN input (never cleared);  iterations_out: host int[2] or NULL
 * Scope limits: 4 <= nx <= 8191, ny >= 4; the CSR must be the 5-point staggered-grid pattern piso_assemble_csr produces
 * (any other matrix: PISO_ERR_UNSUPPORTED_PATTERN) -- this is not a general CSR solver.
 * ------------------------------------------------------------------------------------------------------------- */
size_t piso_bicgstab_workspace_bytes(int nx, int ny, int elem_size);

int piso_multi_bicgstab_ilu_f32(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                const float* x0, float* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                size_t workspace_bytes, piso_stream_t stream);
int piso_multi_bicgstab_ilu_f64(const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                const double* x0, double* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                size_t workspace_bytes, piso_stream_t stream);

/* y = A x (transpose == 0) or A^T x on the concatenated two-matrix CSR; the H-operator product of the second corrector
 * (diffpiso/piso_helpers.py:209-223 uses tf.gather/segment_sum for it). */
int piso_csr_matvec_f32(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* x, float* y,
                        int nx, int ny, int transpose, piso_stream_t stream);
int piso_csr_matvec_f32_slab(const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* x, float* y,
                             int nx, int ny, int periodic_x, int periodic_y, int transpose, piso_stream_t stream, const piso_slab_t* slab);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused stencil glue of the step on the flat "u-first" face layout, forward and reverse mode (csrc/glue.hip).  Replaces the
 * TensorFlow / PhiFlow-math ops of diffpiso/piso_tf.py:36-73 and diffpiso/piso_helpers.py:35-55 (custom_padded), :169-172
 * (arrange_rhs_term_tf), :226-274 (finite_volume_gradient_tensor, circular_padded_gradient), :277-310
 * (finite_volume_divergence) and :223 (the H combination).  The reverse-mode entries implement the reference's custom
 * gradients, including their deviations from the exact transpose.
 *   pad_modes   int[4] = pressure extrapolation (x_lo, x_hi, y_lo, y_hi): 0 'constant' (zero), 1 'boundary' (edge), 2 'periodic'
 *   accessible  [(ny+2)(nx+2)] padded cell mask of the gradient (piso_helpers.py:255-265) or NULL (no mask)
 *   a_flat      [n_u+n_v] the "A" array of piso_assemble_csr;  dirichlet [n_u+n_v] bytes or NULL;  hx, hy = (dx, dy)
 * piso_face_forward, by mode (G = masked finite-volume pressure gradient on the faces):
 *   0 RHS    out0 = dirichlet ? -in2 : in0 * beta - G(p) [+ in1 * dxdy]     in0 velocity, in1 forcing or NULL, in2 Dirichlet values
 *   1 CORR1  out0 = in0 - (G(p) / (beta - A)) / dxdy ; out1 = out0 - in0     in0 = u*
 *   2 FINAL  out0 = in0 + (in1 - G(p) / dxdy) / (beta - A)                   in0 = u**, in1 = H
 * piso_face_backward: d_in* and d_p from d_out0 [, d_out1]; NULL d_in1 / d_in2 are skipped.
 * ------------------------------------------------------------------------------------------------------------- */
int piso_pad_velocity(const float* vel_flat, float* vel_pad, int nx, int ny, int periodic_x, int periodic_y, piso_stream_t stream);
int piso_a0_vfirst(const float* a_flat, float* a0_vfirst, int nx, int ny, float beta, float dx_factor, piso_stream_t stream);
int piso_face_forward(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* p,
                      const float* accessible, const float* a_flat, const float* in0, const float* in1, const float* in2,
                      const uint8_t* dirichlet, float* out0, float* out1, piso_stream_t stream);
int piso_face_backward(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta,
                       const float* accessible, const float* a_flat, const uint8_t* dirichlet, const float* d_out0,
                       const float* d_out1, float* d_in0, float* d_in1, float* d_in2, float* d_p, piso_stream_t stream);
int piso_divergence(const float* faces, float* div, int nx, int ny, float dxdy, float hx, float hy, piso_stream_t stream);
int piso_divergence_adjoint(const float* d_div, float* d_faces, int nx, int ny, int periodic_x, int periodic_y, float dxdy, float hx,
                            float hy, piso_stream_t stream);
/* h = m_delta - (A - beta) delta ; h_over_bma = h / (beta - A)   (piso_helpers.py:223, piso_tf.py:66) and its reverse mode:
 * d_m_delta = d_h + d_h_over_bma / (beta - A) ; d_delta = -(A - beta) d_m_delta   (d_h may be NULL) */
int piso_h_contribution(const float* m_delta, const float* delta, const float* a_flat, float beta, float* h, float* h_over_bma, int nx,
                        int ny, piso_stream_t stream);
int piso_h_contribution_adjoint(const float* d_h, const float* d_h_over_bma, const float* a_flat, float beta, float* d_m_delta,
                                float* d_delta, int nx, int ny, piso_stream_t stream);
/* the same on one rank's rows (piso_slab_t: what the arrays hold, which rows a launch writes) */
int piso_pad_velocity_slab(const float* vel_flat, float* vel_pad, int nx, int ny, int periodic_x, int periodic_y, piso_stream_t stream,
                           const piso_slab_t* slab);
int piso_a0_vfirst_slab(const float* a_flat, float* a0_vfirst, int nx, int ny, float beta, float dx_factor, piso_stream_t stream,
                        const piso_slab_t* slab);
int piso_face_forward_slab(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta, const float* p,
                           const float* accessible, const float* a_flat, const float* in0, const float* in1, const float* in2,
                           const uint8_t* dirichlet, float* out0, float* out1, piso_stream_t stream, const piso_slab_t* slab);
int piso_face_backward_slab(int mode, int nx, int ny, const int* pad_modes, float dxdy, float hx, float hy, float beta,
                            const float* accessible, const float* a_flat, const uint8_t* dirichlet, const float* d_out0,
                            const float* d_out1, float* d_in0, float* d_in1, float* d_in2, float* d_p, piso_stream_t stream,
                            const piso_slab_t* slab);
int piso_divergence_slab(const float* faces, float* div, int nx, int ny, float dxdy, float hx, float hy, piso_stream_t stream,
                         const piso_slab_t* slab);
int piso_divergence_adjoint_slab(const float* d_div, float* d_faces, int nx, int ny, int periodic_x, int periodic_y, float dxdy, float hx,
                                 float hy, piso_stream_t stream, const piso_slab_t* slab);
int piso_h_contribution_slab(const float* m_delta, const float* delta, const float* a_flat, float beta, float* h, float* h_over_bma,
                             int nx, int ny, piso_stream_t stream, const piso_slab_t* slab);
int piso_h_contribution_adjoint_slab(const float* d_h, const float* d_h_over_bma, const float* a_flat, float beta, float* d_m_delta,
                                     float* d_delta, int nx, int ny, piso_stream_t stream, const piso_slab_t* slab);

/* ---------------------------------------------------------------------------------------------------------------
 * Pressure matrix.  Replaces LaplaceMatrixKernelLauncher (CUDAsrc/pressure_solve_op.cc:78-84,
 * CUDAsrc/laplace_op.cu.cc:79-239).
 * ------------------------------------------------------------------------------------------------------------- */
int piso_laplace_matrix_f64(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                            double* laplace, piso_stream_t stream);
int piso_laplace_matrix_f32(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                            float* laplace, piso_stream_t stream);
/* one rank's OWNED rows: laplace [(row_end - row_begin) nx][5], masks and a0_vfirst as piso_slab_t stores them */
int piso_laplace_matrix_f64_slab(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                                 double* laplace, piso_stream_t stream, const piso_slab_t* slab);
int piso_laplace_matrix_f32_slab(int nx, int ny, const float* active, const float* fluid, const float* a0_vfirst,
                                 float* laplace, piso_stream_t stream, const piso_slab_t* slab);

/* ---------------------------------------------------------------------------------------------------------------
 * Pressure CG.  Replaces LaunchPressureKernel (CUDAsrc/pressure_solve_op.cc:48-76,
 * CUDAsrc/pressure_solve_op.cu.cc:140-418 double / :420-696 float): plain CG on (L + c 1 1^T) x = b from x0 = 0,
 * c = 0.1*mean|diag L| if rank_deficient, p/r re-initialised every residual_reset iterations, max-norm stop tested
 * every 5th iteration with the reference's flag semantics.
 *   laplace [N][5], divergence [N], x_out [N]; iterations_out: host int* (also the reference's `iterations` output)
 * The call returns when the solve has finished (the host must see the convergence flag, as in the reference).
 * Grids whose rows are a multiple of 128 cells (fp64; 256 for fp32) and that fit the chip run the iterations inside
 * persistent launches (csrc/cg_persist1.h); a grid-wide exchange that times out fails the call with
 * PISO_ERR_HIP.  Tuning / test knobs: piso_set_option() above.
 * ------------------------------------------------------------------------------------------------------------- */
size_t piso_cg_workspace_bytes(int nx, int ny, int elem_size);

int piso_cg_solve_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace, const double* divergence,
                      double* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                      int* iterations_out, void* workspace, size_t workspace_bytes, piso_stream_t stream);
int piso_cg_solve_f32(int nx, int ny, int periodic_x, int periodic_y, const float* laplace, const float* divergence,
                      float* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                      int* iterations_out, void* workspace, size_t workspace_bytes, piso_stream_t stream);

/* The same solve WITHOUT a host round trip, where the library can run it in one launch (today: grids of at most 4 608 cells,
 * csrc/cg_tiny.h - the lid-driven cavity of lid_driven_cavity_2d.py): the call returns as soon as the kernel is queued and the
 * iteration count (the reference's `iterations` output tensor, pressure_solve_op.cc:60-76) is written to DEVICE memory.
 * Any other grid: PISO_ERR_NEEDS_HOST and nothing has been queued - call piso_cg_solve_*. */
int piso_cg_solve_async_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace, const double* divergence,
                            double* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                            int* iterations_dev, void* workspace, size_t workspace_bytes, piso_stream_t stream);
int piso_cg_solve_async_f32(int nx, int ny, int periodic_x, int periodic_y, const float* laplace, const float* divergence,
                            float* x_out, float accuracy, int max_iterations, int rank_deficient, int residual_reset,
                            int* iterations_dev, void* workspace, size_t workspace_bytes, piso_stream_t stream);

/* Fixed-work variant for bandwidth measurements: runs exactly `iterations` CG iterations (no convergence test),
 * optionally timing the kernels with HIP events on `stream`; kernel_ms_out (NULL to skip): host float[2] = average ms per
 * launch of K1 (fused p-update + stencil + dots) and K2 (x/r update + dots), or, when the iterations ran inside persistent
 * launches, {average ms per ITERATION, 0}. */
int piso_cg_fixed_iterations_f64(int nx, int ny, int periodic_x, int periodic_y, const double* laplace,
                                 const double* divergence, double* x_out, int rank_deficient, int iterations,
                                 float* kernel_ms_out, void* workspace, size_t workspace_bytes, piso_stream_t stream);

/* Sampling of K1 / K2 launch durations with HIP events on the launch stream (every `stride`-th iteration); used by
 * bench.py for roofline.achieved.  ms_sum / count: host arrays of FOUR entries: [0] K1 launches, [1] K2 launches,
 * [2] persistent segments (ms_sum[2] = total duration of all segment launches, count[2] = CG ITERATIONS executed inside
 * them), [3] count[3] = number of segment launches (ms_sum[3] unused). */
void piso_cg_profile_enable(int enable, int stride);
void piso_cg_profile_read(double* ms_sum, long long* count);
/* Solves (since load) that were restarted on the two-kernel path because a grid-wide exchange of the persistent kernel timed
 * out (workgroups not co-resident: CU mask, another process on the GPU).  0 on a dedicated GPU. */
int piso_cg_persist_fallbacks(void);
/* With option "cg_xcd_map" = 1: the XCD (0-7) that every workgroup of the calling thread's last solve's LAST chip-wide persistent launch
 * ran on, out[0 .. min(return value, capacity)); returns the number of workgroups (0: the solve ran no such launch).  The exchange adds
 * an XCD's records in workgroup order, so two solves of the same input are bit-for-bit equal whenever the hardware dealt the workgroups
 * to the XCDs the same way - which it does on an otherwise idle GPU; the reproducibility tests check exactly that precondition. */
int piso_cg_last_xcd_map(int* out, int capacity);
/* Solves of grids of at most 4 608 cells run inside ONE workgroup, one launch for the whole solve (csrc/cg_tiny.h: the lid-driven
 * cavity of BASELINE.json's config 1); same iteration and control flow as the chip-wide paths.  Option "cg_tiny": 0 = never. */
long long piso_cg_tiny_solves(void);
/* Every fp64 solve that ran iterations inside the persistent kernel is VERIFIED before it returns: the recurrence residual r must
 * equal b - (L x + c sum x) for the returned x to 1e-5 max|b| (one extra stencil pass).  The persistent kernel publishes perimeter
 * rows without release / acquire fences; a value read before it was visible would break exactly this identity.  A failed check
 * restarts the solve on the two-kernel iteration and is counted here (and in piso_cg_persist_fallbacks).  Option "cg_verify": 0
 * skips the check, 2 treats every check as failed (test knob). */
void piso_cg_verify_stats(long long* runs_out, int* failures_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Convolutions of the CNN turbulence closure on the matrix cores (csrc/conv.hip; exact fp32 MFMA).  Replaces the
 * tf.nn.conv2d / tf.nn.leaky_relu calls of diffpiso/networks.py:3-57 (NHWC activations, batch 1, stride 1, no bias, kernel
 * sizes and channel counts of the closure: 7x7 4->16, 5x5 16->16, 5x5 16->32, 3x3 32->64, 3x3 64->64, 1x1 64->64, 1x1 64->2 and
 * the transposed shapes of the input-gradient pass).
 *   in [H][W][cin], out [Ho][Wo][cout] with Ho = H + 2 pad - ks + 1; leaky_out != 0: leaky ReLU (slope 0.2) on the output.
 *   w_laid_out: the HWIO weights in the kernel's operand layout, zero filled beyond the true channel counts
 *   (piso_conv2d_weight_elems floats, COUTP = round_up(cout, 16)):
 *     cin <= 4: [ks][ks][4][COUTP];   cin > 4 (a multiple of 16): [ks][ks][cin/16][4][COUTP][4] with
 *     element [tap][blk][q][co][j] = W[tap][16 blk + 4 q + j][co]   (every MFMA operand is then one 16-byte load).
 * Input gradient = piso_conv2d_forward(grad of the pre-activation output, flipped + transposed weights, pad = ks - 1 - pad).
 * piso_conv2d_wgrad: dw [ks][ks][cin][cout] (HWIO, true sizes) = sum over pixels of in (x) grad_out (pre-activation);
 * deterministic two-stage reduction through the caller's workspace.
 * ------------------------------------------------------------------------------------------------------------- */
size_t piso_conv2d_weight_elems(int ks, int cin, int cout);
size_t piso_conv2d_wgrad_workspace_bytes(int ks, int cin, int cout);
int piso_conv2d_forward(const float* in, const float* w_laid_out, float* out, int H, int W, int cin, int cout, int ks, int pad,
                        int leaky_out, piso_stream_t stream);
int piso_conv2d_wgrad(const float* in, const float* grad_out, float* dw, int H, int W, int cin, int cout, int ks, int pad,
                      void* workspace, size_t workspace_bytes, piso_stream_t stream);
/* grad_pre = grad_out * leaky_relu'(pre-activation) (slope 0.2), from the layer's saved output (same sign as the pre-activation). */
int piso_leaky_relu_backward(const float* grad_out, const float* out, float* grad_pre, size_t n, piso_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Slab-decomposed pressure CG (no counterpart in the reference, which is single-GPU).
 * The grid is cut along y into `world` slabs of ny_local rows, one rank (process) per GPU.  Two transports:
 *
 *  PEER (piso_comm_peer_create / _connect, or piso_comm_peer_create_fd / _connect_fd; the GPUs of one node, world <= 8): every rank
 *  owns a peer-mapped MAILBOX (uncached device memory exported with hipIpcGetMemHandle - or as a POSIX file descriptor of a
 *  virtual-memory allocation - and mapped by all ranks).  Step 1 creates the mailbox and returns its 64-byte
 *  handle; the caller distributes the handles of all ranks by any means (torch.distributed all_gather, MPI, a pipe); step 2 maps
 *  them.  Reductions and halo rows are written by kernels straight into the consumers' mailboxes (tagged words / release-acquire
 *  at system scope), never through the host.  The NORMAL iterations run inside the persistent kernel: the slab's r, p, x stay on
 *  chip, the z' rows at the slab edges and the per-GPU totals cross xGMI from inside the kernel, one extra hop per iteration;
 *  resets, the first iteration and slab shapes the kernel cannot tile (row length not a multiple of 128, ...) run the two-kernel
 *  iteration with mailbox collectives between the kernels.  A wait on a peer that gives up (peer gone) fails the call, it never
 *  hangs; a failed persistent segment makes ALL ranks restart the solve on the two-kernel iteration (piso_comm_stats counts it).
 *  row_capacity = longest grid row (cells) the communicator will carry.  x_out_global must be NULL (gather the slabs yourself).
 *
 *  RCCL (piso_comm_unique_id / piso_comm_create): per iteration K1, a 3-double all-reduce, K2, a 3-double all-reduce and a
 *  one-row halo exchange of the residual -- stream-ordered RCCL calls, no host sync, two-kernel iteration only.  The same
 *  communicator carries the slab BiCGStab and the halo messages of the sharded step (below).  librccl is
 *  dlopen'ed on first use; the 128-byte unique id is created on one rank and distributed by the caller.
 *
 *   laplace_local [ny_local*nx][5], divergence_local / x_out_local [ny_local*nx]: this rank's rows;
 *   x_out_global  [world*ny_local*nx] or NULL (RCCL transport): every rank receives the full solution (all-gather).
 * piso_cg_solve_slab_emulated_f64 runs `slabs` virtual ranks on ONE device in lock-step with an in-process loopback: same
 * kernels, same halo / partial-sum logic; it exists to test the multi-rank index logic of the two-kernel iteration on one GPU.
 * piso_comm_stats: out6 = {transport (1 RCCL, 2 peer), CG iterations executed inside persistent slab segments, solves restarted
 * on the two-kernel iteration, persistent launches, slab solves verified against the true residual b - A^ x after persistent
 * segments (every one is; the check needs the neighbours' edge rows of x and covers what crossed xGMI), checks that failed on
 * some rank (all ranks then restart together)}.
 * ------------------------------------------------------------------------------------------------------------- */
int piso_comm_peer_create(int rank, int world, int row_capacity, void** comm_out, void* ipc_handle64_out);
int piso_comm_peer_connect(void* comm, const void* ipc_handles64_all_ranks);
/* The same mailboxes for nodes that refuse hipIpc handles across ranks: exportable virtual-memory allocations (hipMemCreate, uncached
 * type) shared as POSIX file descriptors.  Step 1 creates and maps the rank's mailbox and returns a file descriptor; the caller hands a
 * copy to every other rank (a Unix socket with SCM_RIGHTS; diffpiso/distributed.py) and closes its own; step 2 takes the descriptors
 * received from the other ranks (fds_all_ranks[r] for r != rank), imports and maps them.  Everything else is the peer transport above. */
int piso_comm_peer_create_fd(int rank, int world, int row_capacity, void** comm_out, int* fd_out);
int piso_comm_peer_connect_fd(void* comm, const int* fds_all_ranks);
/* Mailbox ping-pong: `iters` round trips of one tagged 8-byte word between ranks a (initiator) and b; a == b: a rank's own mailbox.
 * Called by EVERY rank with the same arguments (others return at once); us_per_round_trip_out is written on rank a. */
int piso_comm_pingpong(void* comm, int a, int b, int iters, float* us_per_round_trip_out, piso_stream_t stream);
int piso_comm_stats(void* comm, long long* out6);
/* The slab-decomposed STEP (new design): the *_slab twins of piso_assemble_csr, piso_pad_velocity, piso_a0_vfirst,
 * piso_face_forward / _backward, piso_divergence(_adjoint), piso_h_contribution(_adjoint), piso_laplace_matrix_* and piso_csr_matvec_f32
 * (declared next to their entry points; piso_slab_t above says what the arrays hold) work on one rank's rows.
 * piso_comm_exchange fills halo rows: msgs28 = 4 x {count <= 3, off[3], len[3]} element segments of `vec` {sent to the upper
 * neighbour, sent to the lower, received from the lower, received from the upper} (ring neighbours); dtype 0 float, 1 double,
 * 2 int32; a no-op on one rank.  Peer transport: one launch, the elements cross xGMI as 8-byte words written into the consumer's
 * mailbox.  RCCL transport: grouped ncclSend / ncclRecv of the segments on the stream, straight from / into `vec`.
 * piso_comm_check: has a wait on a peer given up (agreed over the ranks; peer transport - RCCL has no bounded waits)? */
int piso_comm_exchange(void* comm, void* vec, int dtype, const int* msgs28, piso_stream_t stream);
int piso_comm_check(void* comm, piso_stream_t stream);
/* Slab-decomposed ILU(0)-BiCGStab (either transport): same arguments as piso_multi_bicgstab_ilu_*, all arrays FULL on every rank
 * (the assembly is cheap and replicated); the rank works on the face rows of its ny / world cell rows, which must be whole
 * preconditioner bands (ny / world a multiple of the band height: then the banded ILU(0) is the single-GPU one and the iterates
 * agree to summation order).  Per iteration: the five dot products are all-reduced INSIDE the one-block scalar kernels (tagged
 * words through the mailboxes), the inputs of the two SpMVs receive their neighbours' edge rows (u[j], v[j], v[j+1] downwards,
 * u[j], v[j] upwards, ring).  x_out is valid on the owned rows only.  The communicator's row_capacity must be >= 3 nx + 1.
 * RCCL transport (where the environment refuses hipIpc): every scalar stage is two launches around an 8-double ncclAllReduce, the
 * edge rows travel by grouped ncclSend / ncclRecv on the side stream - stream-ordered, no host sync, same arithmetic. */
int piso_multi_bicgstab_ilu_slab_f32(void* comm, const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                     const float* x0, float* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                     int band_rows, uint8_t* warning, int* iterations_out, void* workspace, size_t workspace_bytes,
                                     piso_stream_t stream);
int piso_multi_bicgstab_ilu_slab_f64(void* comm, const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                     const double* x0, double* x_out, int nx, int ny, float tol, int max_it, int transpose,
                                     int band_rows, uint8_t* warning, int* iterations_out, void* workspace, size_t workspace_bytes,
                                     piso_stream_t stream);
/* The same solver on LOCAL storage (the slab-decomposed step, round 5): csr_*, rhs, x0, x_out and the workspace hold the rank's stored
 * rows only (piso_slab_t; the caller fills the halo rows of csr_val with piso_comm_exchange - the transposed solve gathers from them);
 * nx, ny, periodic_* are the whole grid's.  x_out is written on the owned rows. */
size_t piso_bicgstab_slab_workspace_bytes(int nx, int ny, int elem_size, const piso_slab_t* slab);
int piso_multi_bicgstab_ilu_slab_local_f32(void* comm, const float* csr_val, const int* csr_rowptr, const int* csr_col, const float* rhs,
                                           const float* x0, float* x_out, int nx, int ny, int periodic_x, int periodic_y, float tol, int max_it,
                                           int transpose, int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                           size_t workspace_bytes, piso_stream_t stream, const piso_slab_t* slab);
int piso_multi_bicgstab_ilu_slab_local_f64(void* comm, const double* csr_val, const int* csr_rowptr, const int* csr_col, const double* rhs,
                                           const double* x0, double* x_out, int nx, int ny, int periodic_x, int periodic_y, float tol, int max_it,
                                           int transpose, int band_rows, uint8_t* warning, int* iterations_out, void* workspace,
                                           size_t workspace_bytes, piso_stream_t stream, const piso_slab_t* slab);
int piso_comm_unique_id(void* id128);
int piso_comm_create(const void* id128, int rank, int world, void** comm_out);
int piso_comm_destroy(void* comm);
size_t piso_cg_slab_workspace_bytes(int nx, int ny_local, int local_ranks);
int piso_cg_solve_slab_f64(void* comm, int nx, int ny_local, int periodic_x, int periodic_y, const double* laplace_local,
                           const double* divergence_local, double* x_out_local, double* x_out_global, float accuracy,
                           int max_iterations, int rank_deficient, int residual_reset, int* iterations_out, void* workspace,
                           size_t workspace_bytes, piso_stream_t stream);
int piso_cg_solve_slab_emulated_f64(int slabs, int nx, int ny, int periodic_x, int periodic_y, const double* laplace,
                                    const double* divergence, double* x_out, float accuracy, int max_iterations,
                                    int rank_deficient, int residual_reset, int* iterations_out, void* workspace,
                                    size_t workspace_bytes, piso_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PISO_HIP_H */

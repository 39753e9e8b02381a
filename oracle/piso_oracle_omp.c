/*
 * piso_oracle_omp.c -- CPU ORACLE, multi-threaded pressure CG (test infrastructure, NOT product code).
 *
 * The same algorithm and control flow as oracle_cg_f64 in piso_oracle.c (LaunchPressureKernel,
 * CUDAsrc/pressure_solve_op.cu.cc:140-418; calcZ_v4 :57-92; checkResiduum :94-102; initVariablesWithGuess :104-114),
 * with the vector passes spread over OpenMP threads.  Used for (a) bench.py's cpu_baseline leg on all host cores and
 * (b) generating the full-size fixtures of tests/golden/make_golden_configs.py in minutes instead of hours.
 * Reductions are DETERMINISTIC for any thread count: every vector is cut into fixed chunks of CHUNK elements, each chunk
 * is summed left to right by one thread, the chunk sums are added left to right by the master.  (The summation order
 * therefore differs from the single-threaded oracle, as cuBLAS' differs from both; converged answers agree, iteration
 * counts of the shifted - indefinite - operator need not: DESIGN.md "Oracle", findings.)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))
#define CHUNK 8192

ORACLE_API int oracle_omp_max_threads(void) { return omp_get_max_threads(); }
ORACLE_API void oracle_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

typedef struct { double a, b, c; } sum3;

/* z = L p (+ vsum), calcZ_v4 :57-92, rows [r0, r1) */
static inline void apply_rows(int nx, int ny, int per_x, int per_y, const double* L, const double* p, double* z, double vsum,
                              int r0, int r1) {
  const int N = nx * ny;
  const int off[5] = {-nx, -1, 0, 1, nx};
  const int poff[5] = {N * per_y, nx * per_x, 0, -nx * per_x, -N * per_y};   /* calcDiagonalOffsets :117-133 */
  for (int row = r0; row < r1; ++row) {
    const int i = row % nx, j = row / nx;
    const int onb[5] = {j == 0, i == 0, 0, i == nx - 1, j == ny - 1};
    double tmp = 0;
    for (int s = 0; s < 5; ++s) {
      const double l = L[(size_t)row * 5 + s];
      const int ci = row + off[s] + onb[s] * poff[s];
      tmp += l * p[ci * (l != 0.0)];
    }
    z[row] = tmp + vsum;
  }
}

static double chunk_sum(int N, const double* a, double* part) {
  const int nc = (N + CHUNK - 1) / CHUNK;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < nc; ++c) {
    double s = 0;
    const int e = (c + 1) * CHUNK < N ? (c + 1) * CHUNK : N;
    for (int k = c * CHUNK; k < e; ++k) s += a[k];
    part[c] = s;
  }
  double s = 0;
  for (int c = 0; c < nc; ++c) s += part[c];
  return s;
}

static double chunk_dot(int N, const double* a, const double* b, double* part) {
  const int nc = (N + CHUNK - 1) / CHUNK;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < nc; ++c) {
    double s = 0;
    const int e = (c + 1) * CHUNK < N ? (c + 1) * CHUNK : N;
    for (int k = c * CHUNK; k < e; ++k) s += a[k] * b[k];
    part[c] = s;
  }
  double s = 0;
  for (int c = 0; c < nc; ++c) s += part[c];
  return s;
}

static void apply_all(int nx, int ny, int per_x, int per_y, const double* L, const double* p, double* z, double vsum) {
  const int N = nx * ny, nc = (N + CHUNK - 1) / CHUNK;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < nc; ++c) apply_rows(nx, ny, per_x, per_y, L, p, z, vsum, c * CHUNK, (c + 1) * CHUNK < N ? (c + 1) * CHUNK : N);
}

/* history (optional, [max_iterations]): max|r| after every iteration, for convergence plots of the fixtures */
ORACLE_API int oracle_cg_f64_omp(int nx, int ny, int per_x, int per_y, const double* L, const double* b, double* x, double* p,
                                 double* z, double* r, float accuracy, int max_iterations, int rank_deficient, int reset_steps,
                                 double* history) {
  const int N = nx * ny, nc = (N + CHUNK - 1) / CHUNK;
  double* part = (double*)malloc(sizeof(double) * (size_t)nc);
  int* ipart = (int*)malloc(sizeof(int) * (size_t)nc);
  double c = 0; /* vectorSum_scaling :161-168 */
  if (rank_deficient) {
    for (int k = 0; k < N; ++k) c += fabs(L[(size_t)k * 5 + 2]);
    c *= .1 / N;
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < N; ++k) x[k] = 0; /* :186-190 */
  apply_all(nx, ny, per_x, per_y, L, x, z, rank_deficient ? c * chunk_sum(N, x, part) : 0);
  int flag_dev = 0, flag_host = 0;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < N; ++k) p[k] = r[k] = b[k] - z[k]; /* initVariablesWithGuess :104-114 */
  int checker = 1, iterations = 0;
  for (; iterations < max_iterations; iterations++) { /* :257-357 */
    if ((iterations + 1) % reset_steps == 0) { /* residual reset :260-274 */
      apply_all(nx, ny, per_x, per_y, L, x, z, rank_deficient ? c * chunk_sum(N, x, part) : 0);
#pragma omp parallel for schedule(static)
      for (int k = 0; k < N; ++k) p[k] = r[k] = b[k] - z[k];
      flag_dev = 0;
    }
    apply_all(nx, ny, per_x, per_y, L, p, z, rank_deficient ? c * chunk_sum(N, p, part) : 0);
    const double p_r = chunk_dot(N, p, r, part), p_z = chunk_dot(N, p, z, part);
    double alpha = 0.;
    if (fabs(p_z) > 0.) alpha = p_r / p_z; /* :301-302 */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) { x[k] += alpha * p[k]; r[k] += -alpha * z[k]; }
    if (history || checker % 5 == 0) {
#pragma omp parallel for schedule(static)
      for (int cc = 0; cc < nc; ++cc) {
        int bad = 0;
        double m = 0;
        const int e = (cc + 1) * CHUNK < N ? (cc + 1) * CHUNK : N;
        for (int k = cc * CHUNK; k < e; ++k) { const double a = fabs(r[k]); if (a >= accuracy || a != a) bad = 1; if (a > m || a != a) m = a; }
        ipart[cc] = bad; part[cc] = m;
      }
      if (history) { double m = 0; for (int cc = 0; cc < nc; ++cc) if (part[cc] > m || part[cc] != part[cc]) m = part[cc]; history[iterations] = m; }
    }
    if (checker % 5 == 0) { /* :312-335, checkResiduum :94-102 */
      for (int cc = 0; cc < nc; ++cc) if (ipart[cc]) { flag_dev = 0; break; }
      flag_host = flag_dev;
      if (flag_host) { iterations++; break; }
      flag_dev = 1; /* cudaMemset(threshold_reached, 1) :334 */
    }
    checker++;
    const double r_z = chunk_dot(N, r, z, part);
    const double beta = -r_z / p_z; /* :351-352 */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < N; ++k) p[k] = beta * p[k] + r[k];
  }
  free(part);
  free(ipart);
  return iterations;
}

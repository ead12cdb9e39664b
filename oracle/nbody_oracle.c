/* nbody_oracle.c -- C restatement of the reference's all-pairs force, TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from this file; the product (nbody-deep-sim_amd/) never does.
 *
 * Follows BaseSimulator.compute_accelerations, src/galaxify/simulation.py:71-89:
 *   diff = r_j - r_i (:80); dist_sq = (dx^2 + dy^2) + dz^2 + eps^2 (:82); inv = dist_sq^(-3/2) (:83);
 *   diagonal zeroed (:85); acc_i = G * sum_j m_j * diff * inv (:86-88).
 * Two variants: f32 (per-pair arithmetic in fp32 exactly as above; the row sum is carried in double,
 * which is closer to torch's blocked fp32 reduction than a 65 536-term serial fp32 chain would be) and
 * f64 (everything in double: the accuracy yardstick). Pinned by tests/test_oracle_golden.py against
 * the golden vectors the real reference produced. OpenMP over target rows.
 */
#include <math.h>
#include <stddef.h>

void nbody_oracle_acc_f32(const float* pos, const float* mass, int n, int row_lo, int row_hi, float g,
                          float eps2, float* acc) {
#pragma omp parallel for schedule(static)
  for (int i = row_lo; i < row_hi; ++i) {
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int j = 0; j < n; ++j) {
      const float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
      const float d2 = ((dx * dx + dy * dy) + dz * dz) + eps2;
      float inv = (float)pow((double)d2, -1.5);   /* correctly rounded fp32 pow(-1.5) */
      if (j == i) inv = 0.f;
      ax += (double)(dx * inv * mass[j]);
      ay += (double)(dy * inv * mass[j]);
      az += (double)(dz * inv * mass[j]);
    }
    acc[3 * (size_t)(i - row_lo)] = g * (float)ax;
    acc[3 * (size_t)(i - row_lo) + 1] = g * (float)ay;
    acc[3 * (size_t)(i - row_lo) + 2] = g * (float)az;
  }
}

void nbody_oracle_acc_f64(const float* pos, const float* mass, int n, int row_lo, int row_hi, double g,
                          double eps2, double* acc) {
#pragma omp parallel for schedule(static)
  for (int i = row_lo; i < row_hi; ++i) {
    const double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      const double dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
      const double d2 = dx * dx + dy * dy + dz * dz + eps2;
      const double inv = 1.0 / (d2 * sqrt(d2)) * mass[j];
      ax += dx * inv; ay += dy * inv; az += dz * inv;
    }
    acc[3 * (size_t)(i - row_lo)] = g * ax;
    acc[3 * (size_t)(i - row_lo) + 1] = g * ay;
    acc[3 * (size_t)(i - row_lo) + 2] = g * az;
  }
}

// generators.hip -- initial-condition generators on the device, fp64 as upstream.
//
// The reference's generate_disk / generate_spiral (src/galaxify/galaxies.py:54-192, 195-296) draw from NumPy's
// legacy GLOBAL random stream and then do per-body arithmetic; generate_disk also evaluates, for every star, the
// mass strictly inside its radius with an O(N^2) masked sum (:143-152). The draws stay on the host (the stream
// has to be consumed in the reference's order for a seed to give the same galaxy); everything after them runs
// here: elementwise kernels, a fixed-order two-stage reduction for the Hernquist normalisation, and for the
// enclosed mass a sort by radius + prefix sum + lower-bound search (radix sort and scan from hipCUB -- plain
// library primitives -- the rest hand-written).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>

#include "../../include/nbd.h"

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }
constexpr double kEps32 = 1.1920928955078125e-07;  // np.finfo(np.float32).eps

// radius, height, azimuth, planar position, Hernquist weight (galaxies.py:97-137); body 0 is the black hole
__global__ __launch_bounds__(256) void disk_stage1_kernel(const double* __restrict__ u_r, const double* __restrict__ u_z,
                                                          const double* __restrict__ u_phi, int n, double radial_scale,
                                                          double height_scale, double total_mass, double* __restrict__ dist,
                                                          double* __restrict__ phi, double* __restrict__ pos,
                                                          double* __restrict__ weight, double* __restrict__ partial) {
  __shared__ double red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double w = 0.0;
  if (i < n) {
    double d = -radial_scale * log(1.0 - u_r[i]);
    double z = u_z[i] * height_scale * (1.0 - sqrt(d));
    if (i == 0) { d = 0.0; z = 0.0; }
    const double ph = u_phi[i] * 2.0 * M_PI;
    dist[i] = d; phi[i] = ph;
    pos[3 * i] = cos(ph) * d; pos[3 * i + 1] = sin(ph) * d; pos[3 * i + 2] = z;
    if (i > 0) {
      const double r = d == 0.0 ? kEps32 : d;                                  // avoid_distance_zero (:39-41)
      const double q = 1.0 + r;
      w = (total_mass / (2.0 * M_PI)) * (1.0 / (r * (q * q * q)));             // r0 = 1 (:135)
    }
    weight[i] = w;
  }
  // block sum in a fixed order; blocks are combined in index order by stage 2
  for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n_parts,
                                                           double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int b = threadIdx.x; b < n_parts; b += 256) s += partial[b];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// masses (galaxies.py:128-137)
__global__ __launch_bounds__(256) void disk_mass_kernel(const double* __restrict__ weight, const double* __restrict__ wsum,
                                                        int n, double total_mass, double m_bh, double* __restrict__ mass) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  mass[i] = i == 0 ? m_bh : weight[i] * ((total_mass - m_bh) / wsum[0]);
}

// velocities from the enclosed mass, spin direction, Euler rotation, offsets (galaxies.py:142-192).
// prefix[k] = sum of the k + 1 smallest-radius masses; enclosed(i) = prefix[lower_bound(sorted, dist_i) - 1].
__global__ __launch_bounds__(256) void disk_stage3_kernel(const double* __restrict__ dist, const double* __restrict__ phi,
                                                          const double* __restrict__ sorted_dist,
                                                          const double* __restrict__ prefix, int n, double g_const,
                                                          int clockwise, const double* __restrict__ rot /*3x3 row-major*/,
                                                          double ox, double oy, double oz, double vx0, double vy0,
                                                          double vz0, double* __restrict__ pos, double* __restrict__ vel) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v[3] = {0.0, 0.0, 0.0};
  if (i > 0) {
    const double d = dist[i];
    int lo = 0, hi = n;                                     // first index with sorted_dist >= d
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_dist[mid] < d) lo = mid + 1; else hi = mid; }
    const double m_enc = lo > 0 ? prefix[lo - 1] : 0.0;     // masses[distances < distances[i]].sum()  (:147)
    const double s = sqrt(g_const * m_enc / d);
    v[0] = s * cos(phi[i] + M_PI / 2.0);
    v[1] = s * sin(phi[i] + M_PI / 2.0);
    if (clockwise) { v[0] = -v[0]; v[1] = -v[1]; }
  } else if (clockwise) {
    v[0] = -0.0; v[1] = -0.0;
  }
  const double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  const double off[3] = {ox, oy, oz}, v0[3] = {vx0, vy0, vz0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {                             // row-vector times matrix, then the offset
    pos[3 * i + c] = (p[0] * rot[c] + p[1] * rot[3 + c] + p[2] * rot[6 + c]) + off[c];
    vel[3 * i + c] = (v[0] * rot[c] + v[1] * rot[3 + c] + v[2] * rot[6 + c]) + v0[c];
  }
}

// generate_spiral after its draws (galaxies.py:245-294). raw: (n - 1, 6) = {r, u_phi, g_z, g_R, g_phi, g_vz} per star
__global__ __launch_bounds__(256) void spiral_kernel(const double* __restrict__ raw, int n, double total_mass,
                                                     double radial_scale, double height_scale, double g_const,
                                                     double m_bh, int n_arms, double pitch_angle, double arm_strength,
                                                     double* __restrict__ pos, double* __restrict__ vel,
                                                     double* __restrict__ mass) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i == 0) {
    mass[0] = m_bh;
    for (int c = 0; c < 3; ++c) { pos[c] = 0.0; vel[c] = 0.0; }
    return;
  }
  const double* q = raw + (size_t)(i - 1) * 6;
  const double r = q[0], ph = 2.0 * M_PI * q[1];
  mass[i] = (total_mass - m_bh) / (double)(n - 1);
  double ang = ph;
  if (r > 0.0) ang = ph + arm_strength * sin(n_arms * (ph - log(r / radial_scale) / tan(pitch_angle)));
  const double ca = cos(ang), sa = sin(ang);
  pos[3 * i] = r * ca; pos[3 * i + 1] = r * sa; pos[3 * i + 2] = 0.0 + height_scale * q[2];
  const double m_enc = total_mass * (1.0 - exp(-r / radial_scale) * (1.0 + r / radial_scale));
  const double v_circ = r < 1e-8 ? 0.0 : sqrt(g_const * m_enc / r);
  const double v_r = 0.0 + (0.1 * v_circ) * q[3];
  const double v_phi = v_circ + (0.0 + (0.07 * v_circ) * q[4]);
  const double v_z = 0.0 + (0.05 * v_circ) * q[5];
  vel[3 * i] = v_r * ca - v_phi * sa; vel[3 * i + 1] = v_r * sa + v_phi * ca; vel[3 * i + 2] = v_z;
}

size_t cub_temp_bytes(int n) {
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const double*)nullptr, (double*)nullptr, (const double*)nullptr,
                                           (double*)nullptr, n);      // size queries: nothing runs, nothing can fail
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (const double*)nullptr, (double*)nullptr, n);
  return ((a > b ? a : b) + 255) & ~(size_t)255;
}

}  // namespace

extern "C" {

size_t nbd_disk_workspace_bytes(int n) {
  if (n <= 0) return 0;
  // dist, phi, weight, sorted_dist, sorted_mass (-> prefix): 5 n doubles; the weight sum; library scratch
  return (size_t)5 * n * sizeof(double) + 8 * sizeof(double) + cub_temp_bytes(n) + 256;
}

int nbd_disk_from_draws_f64(const double* u_r, const double* u_z, const double* u_phi, int n, double total_mass,
                            double radial_scale, double height_scale, double g_const, double black_hole_mass,
                            int clockwise, const double* rot3x3, const double* offset3_host,
                            const double* initial_vel3_host, double* pos, double* vel, double* mass, void* workspace,
                            size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!u_r || !u_z || !u_phi || !rot3x3 || !offset3_host || !initial_vel3_host || !pos || !vel || !mass) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_disk_workspace_bytes(n)) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* dist = static_cast<double*>(workspace);
  double* phi = dist + n;
  double* weight = phi + n;
  double* sdist = weight + n;
  double* smass = sdist + n;
  double* wsum = smass + n;
  double* partial = sdist;                               // block sums: sorted_dist is not needed before the sort
  void* cub_ws = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(wsum + 8) + 255) & ~(uintptr_t)255);
  size_t cub_bytes = cub_temp_bytes(n);
  const int blocks = ceil_div(n, 256);
  disk_stage1_kernel<<<blocks, 256, 0, st>>>(u_r, u_z, u_phi, n, radial_scale, height_scale, total_mass, dist, phi, pos,
                                            weight, partial);
  int rc = status();
  if (rc) return rc;
  sum_partials_kernel<<<1, 256, 0, st>>>(partial, blocks, wsum);
  if ((rc = status())) return rc;
  const double m_bh = total_mass * black_hole_mass;
  disk_mass_kernel<<<blocks, 256, 0, st>>>(weight, wsum, n, total_mass, m_bh, mass);
  if ((rc = status())) return rc;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub_ws, cub_bytes, dist, sdist, mass, smass, n, 0, 64, st);
  if (e != hipSuccess) return (int)e;
  e = hipcub::DeviceScan::InclusiveSum(cub_ws, cub_bytes, smass, smass, n, st);
  if (e != hipSuccess) return (int)e;
  disk_stage3_kernel<<<blocks, 256, 0, st>>>(dist, phi, sdist, smass, n, g_const, clockwise, rot3x3, offset3_host[0],
                                            offset3_host[1], offset3_host[2], initial_vel3_host[0], initial_vel3_host[1],
                                            initial_vel3_host[2], pos, vel);
  return status();
}

int nbd_spiral_from_draws_f64(const double* raw, int n, double total_mass, double radial_scale, double height_scale,
                              double g_const, double black_hole_mass, int n_arms, double pitch_angle, double arm_strength,
                              double* pos, double* vel, double* mass, nbd_stream_t stream) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if ((n > 1 && !raw) || !pos || !vel || !mass) return NBD_E_BADARG;
  spiral_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(raw, n, total_mass, radial_scale, height_scale, g_const,
                                                                  total_mass * black_hole_mass, n_arms, pitch_angle,
                                                                  arm_strength, pos, vel, mass);
  return status();
}

}  // extern "C"

/* nbd.h -- C-ABI of libnbd_hip.so: the MI355X (gfx950) hot path of bikuta6/nbody-deep-sim.
 *
 * The reference is pure Python and has no FFI layer of its own; its boundary for this path is
 * the Python API (src/galaxify/simulation.py, gnn.py, contconv.py, trainer.py). Each entry
 * point below names the reference lines whose arithmetic it replaces. The Python host classes
 * in nbody-deep-sim_amd/ keep the reference's names/arguments and call these through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer (HBM) unless it says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous;
 *   - return 0 on success, a positive hipError_t if the HIP runtime failed, a negative
 *     NBD_E_* for argument errors; nothing throws, nothing allocates, no global mutable state;
 *   - all arithmetic is IEEE fp32 ("f32" suffix), indices are int32 on the ABI and int64 where
 *     the reference's tensors are int64 (edge_index).
 */
#ifndef NBD_H_
#define NBD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBD_ABI_VERSION 1
#define NBD_E_BADARG (-1)   /* null pointer / negative size / misaligned buffer              */
#define NBD_E_WORKSPACE (-2) /* workspace smaller than nbd_*_workspace_bytes() reported        */
#define NBD_E_UNSUPPORTED (-3)

#define NBD_SRC_PAD 64      /* packed source arrays are padded to a multiple of this          */

typedef void* nbd_stream_t;

int nbd_abi_version(void);
/* Human-readable text for a return code (static storage). */
const char* nbd_strerror(int code);

/* ---------------------------------------------------------------- direct-force integrator */

/* Length (in float4 entries) of the packed {x,y,z,m} array for n bodies: n rounded up to
 * NBD_SRC_PAD. Padding entries carry m = 0 and contribute exactly 0. */
int nbd_posm_padded_len(int n);

/* Pack positions (n,3) row-major + masses (n,) into float4 {x,y,z,m}[nbd_posm_padded_len(n)].
 * Layout step for nbd_accel_f32; replaces the implicit (N,3)/(N,) tensor reads of
 * simulation.py:80,87. posm must be 16-byte aligned. */
int nbd_pack_posm_f32(const float* pos, const float* mass, int n, float* posm, nbd_stream_t stream);

/* Bytes of scratch nbd_accel_f32 needs for this problem size (partial-force slabs). */
size_t nbd_accel_workspace_bytes(int n_src, int n_tgt);

/* Launch geometry nbd_accel_f32 will use (introspection for tests / bench / DESIGN.md):
 * groups = 128-target workgroup columns, slabs = source splits across workgroups,
 * chunks_per_wave = 64-source chunks each wave streams. Any out pointer may be NULL (host). */
int nbd_accel_plan(int n_src, int n_tgt, int* groups, int* slabs, int* chunks_per_wave);

/* All-pairs softened gravity, BaseSimulator.compute_accelerations (simulation.py:71-89):
 *   acc[i] = g_const * sum_j m_j (r_j - r_i) (|r_j - r_i|^2 + softening_sq)^(-3/2),
 * the pair with global index j == tgt_global_offset + i excluded (fill_diagonal_(0), :85).
 *   posm_src  packed sources, n_src real entries, nbd_posm_padded_len(n_src) allocated
 *   posm_tgt  packed targets (n_tgt entries; may point into posm_src)
 *   tgt_global_offset  index in the source numbering of target 0 (range partition, multi-GPU)
 *   softening_sq       (float)(softening**2), the fp32 scalar of simulation.py:82
 *   acc_out   (n_tgt,3) row-major fp32
 * Deterministic: same inputs give bit-identical outputs on every call. */
int nbd_accel_f32(const float* posm_src, int n_src, const float* posm_tgt, int n_tgt,
                  int tgt_global_offset, float softening_sq, float g_const, float* acc_out,
                  void* workspace, size_t workspace_bytes, nbd_stream_t stream);

/* v += c_kick * a ; x += c_drift * v ; posm = pack(x, m)   (in place on pos, vel)
 * LeapFrogSimulator.step first half, simulation.py:164,166 with c_kick = (float)(0.5*dt),
 * c_drift = (float)dt; two roundings per update (mul then add), as torch eager does.
 * posm may be NULL (no packing). */
int nbd_kick_drift_f32(float* pos, float* vel, const float* acc, const float* mass, int n,
                       float c_kick, float c_drift, float* posm, nbd_stream_t stream);

/* v += c * a   (simulation.py:170, and :185 for Euler). */
int nbd_kick_f32(float* vel, const float* acc, int n, float c, nbd_stream_t stream);

/* x += c * v   (simulation.py:187). */
int nbd_drift_f32(float* pos, const float* vel, int n, float c, nbd_stream_t stream);

/* Bytes of scratch the fused step entry points need (always >= one slab). */
size_t nbd_step_workspace_bytes(int n);

/* One whole LeapFrogSimulator.step (simulation.py:153-170) on one GPU:
 *   kick-drift-pack -> all-pairs force -> kick, three launches on `stream`.
 * acc_in is a(t) (read), acc_out receives a(t+dt) (may alias acc_in).
 * posm: scratch float4[nbd_posm_padded_len(n)]; workspace as nbd_step_workspace_bytes(n). */
int nbd_leapfrog_step_f32(float* pos, float* vel, const float* acc_in, float* acc_out,
                          const float* mass, int n, float dt_half, float dt, float softening_sq,
                          float g_const, float* posm, void* workspace, size_t workspace_bytes,
                          nbd_stream_t stream);

/* Same as nbd_leapfrog_step_f32, additionally recording two caller-created hipEvent_t handles on
 * `stream` immediately before and after the all-pairs force kernel (NULL = skip). Measurement
 * hook for bench.py's roofline leg; the arithmetic and launches are identical. */
int nbd_leapfrog_step_ev_f32(float* pos, float* vel, const float* acc_in, float* acc_out,
                             const float* mass, int n, float dt_half, float dt, float softening_sq,
                             float g_const, float* posm, void* workspace, size_t workspace_bytes,
                             nbd_stream_t stream, void* ev_force_begin, void* ev_force_end);

/* One whole EulerSimulator.step (simulation.py:173-187): force -> kick(dt) -> drift(dt). */
int nbd_euler_step_f32(float* pos, float* vel, float* acc_out, const float* mass, int n, float dt,
                       float softening_sq, float g_const, float* posm, void* workspace,
                       size_t workspace_bytes, nbd_stream_t stream);

/* Bytes of scratch nbd_energy_f32 needs. */
size_t nbd_energy_workspace_bytes(int n);

/* BaseSimulator.compute_energies (simulation.py:91-115):
 *   K = sum_i 0.5 m_i |v_i|^2 ;  U = sum_{i<j} -G m_i m_j / (|r_i - r_j| + softening).
 * out_uk: device double[2] = {U, K} (pair sums are accumulated in fp64 from fp32 terms). */
int nbd_energy_f32(const float* posm, const float* vel, int n, float softening, float g_const,
                   double* out_uk, void* workspace, size_t workspace_bytes, nbd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NBD_H_ */

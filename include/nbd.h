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

#define NBD_ABI_VERSION 2
#define NBD_E_BADARG (-1)   /* null pointer / negative size / misaligned buffer              */
#define NBD_E_WORKSPACE (-2) /* workspace smaller than nbd_*_workspace_bytes() reported        */
#define NBD_E_UNSUPPORTED (-3)

#define NBD_SRC_PAD 64      /* packed source arrays are padded to a multiple of this          */
#define NBD_CC_TILE 128     /* nodes per tile of the fused ContinuousConv kernels             */
#define NBD_CC_GROUPS 8     /* cell groups of the fused ContinuousConv kernel (one per XCD)     */
#define NBD_CC_MAX_RES 4    /* filter resolutions per nbd_contconv_pairs_batch_f32 call       */

typedef void* nbd_stream_t;

int nbd_abi_version(void);
/* sizeof() of the argument structs below as this library was compiled, by name ("nbd_gnn_layer_args", ...; 0 for an
 * unknown name): a binding that mirrors a struct field for field checks its own size against this before the first call
 * (nbd/_lib.py does; tests/test_cabi.py holds every mirror to it). */
size_t nbd_struct_size(const char* name);
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

/* Tuning hook: nbd_accel_f32 with an explicit launch geometry (slabs = source splits across workgroups,
 * 1..64; variant 0 = eight sources in flight per wave, 5 waves/SIMD; variant 1 = four, 8 waves/SIMD) and an
 * optional excluded source range [exclude_lo, exclude_hi) (those sources contribute nothing). The launch
 * plan of the library was chosen with it (tools/sweep_accel_plan.py); every geometry computes the same sums
 * in a different association, so results agree to rounding. Workspace: slabs * n_tgt * 3 floats. */
size_t nbd_accel_tuned_workspace_bytes(int n_tgt, int slabs);
int nbd_accel_tuned_f32(const float* posm_src, int n_src, int exclude_lo, int exclude_hi, const float* posm_tgt,
                        int n_tgt, int tgt_global_offset, float softening_sq, float g_const, float* acc_out,
                        void* workspace, size_t workspace_bytes, int slabs, int variant, nbd_stream_t stream);

/* ---- range-sharded step: one rank of a range partition owns bodies [lo, lo + n_local) of n_total
 * (the reference has no distributed code -- this is the same LeapFrogSimulator.step,
 * simulation.py:153-170, with the force sum of :80-88 split by source ownership). Per step and rank:
 *   nbd_kick_drift_f32(local state) -> posm_local
 *   [all-gather of posm_local into posm_all: issued by the host, asynchronous]
 *   nbd_shard_force_local_f32        own bodies as sources; runs while the gather is in flight
 *   [wait for the gather]
 *   nbd_shard_force_remote_f32       all other bodies as sources, then acc = G * (fixed-order sum of every
 *                                    partial-force slab of both launches) and v += c_kick * acc fused
 * posm_local: float4[nbd_posm_padded_len(n_local)] (padding zero); posm_all: float4[padded(n_total)] in
 * global order. Any lo / n_local is accepted (chunks of posm_all that straddle lo or lo + n_local are
 * walked with an element mask). vel may be NULL (no kick: compute_accelerations / Euler). Workspace as
 * nbd_shard_workspace_bytes, the same buffer for both calls of a step. Deterministic. */
int nbd_shard_plan(int n_total, int lo, int n_local, int* slabs_local, int* chunks_per_wave_local,
                   int* slabs_remote, int* chunks_per_wave_remote);
size_t nbd_shard_workspace_bytes(int n_total, int lo, int n_local);
int nbd_shard_force_local_f32(const float* posm_local, int n_local, float softening_sq, void* workspace,
                              size_t workspace_bytes, int n_total, int lo, nbd_stream_t stream);
int nbd_shard_force_remote_f32(const float* posm_all, int n_total, const float* posm_local, int n_local, int lo,
                               float softening_sq, float g_const, float* acc_out, float* vel, float c_kick,
                               void* workspace, size_t workspace_bytes, nbd_stream_t stream);
/* The same two launches for a system of EQUAL masses (see nbd_leapfrog_step_uniform_f32): the force kernels without their
 * per-pair mass multiply, g_const * mass_value applied once by the finishing pass. The caller vouches that every body's
 * mass equals mass_value. */
int nbd_shard_force_local_uniform_f32(const float* posm_local, int n_local, float softening_sq, void* workspace,
                                      size_t workspace_bytes, int n_total, int lo, nbd_stream_t stream);
int nbd_shard_force_remote_uniform_f32(const float* posm_all, int n_total, const float* posm_local, int n_local, int lo,
                                       float softening_sq, float g_const, float mass_value, float* acc_out, float* vel,
                                       float c_kick, void* workspace, size_t workspace_bytes, nbd_stream_t stream);

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

/* out[0..3n) = pos, out[3n..6n) = vel, out[6n..9n) = acc: the per-step state clones of BaseSimulator.run
 * (simulation.py:135-139) as one launch, so that a chunk of steps can be captured into a hipGraph with its
 * snapshots going to a device ring (one device->host copy per chunk instead of three per step). */
int nbd_snapshot_f32(const float* pos, const float* vel, const float* acc, int n, float* out, nbd_stream_t stream);

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
/* The same step for a system whose bodies all have the SAME mass (round 3; the published configurations: Plummer,
 * m = 1 / N): the mass factors out of the force sum, a = (G m) sum_j d_ij s_ij^3, so the force kernel drops its per-pair
 * multiply by m_j (11 packed fp32 ops + 2 v_rsq_f32 per source and pair of targets instead of 12 + 2) and g_const *
 * mass_value is applied once, to the finished sum. The differences r_j - r_i stay the exact fp32 subtractions of
 * simulation.py:80; one multiplication per sum rounds differently from one per term. The caller vouches that every
 * entry of `mass` equals mass_value. Workspace: nbd_step_workspace_bytes(n). */
int nbd_leapfrog_step_uniform_f32(float* pos, float* vel, const float* acc_in, float* acc_out, const float* mass,
                                  float mass_value, int n, float dt_half, float dt, float softening_sq, float g_const,
                                  float* posm, void* workspace, size_t workspace_bytes, nbd_stream_t stream,
                                  void* ev_force_begin, void* ev_force_end);

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

/* ------------------------------------------------------------ surrogate models: graph build
 * Replace the torch_cluster kernels the reference reaches through PyG. Index-exact rule (the
 * reference delegates ties/truncation to torch_cluster; fixed here, see oracle/surrogate_oracle.py):
 * d2 = (dx*dx + dy*dy) + dz*dz in fp32; edges grouped by centre, ascending centre index;
 * edge_index[0] = neighbour j, edge_index[1] = centre i, both int64, row stride = num_edges.
 * seg_lo/seg_hi (both NULL or both given, int32 per node): candidate range = the node's batch
 * segment, replacing PyG's `batch` vector. */

/* knn_graph(pos, k, batch, loop) -- gnn.py:13, datautils.py:36. Per centre the k smallest (d2, j),
 * ties -> lower j, ascending. out_off[i] = first edge slot of centre i (NULL: i * min(k, n - !loop)).
 * k <= 256. */
int nbd_knn_graph_f32(const float* pos, int n, int k, int loop, const int* seg_lo, const int* seg_hi,
                      const int64_t* out_off, int64_t num_edges, int64_t* edge_index, nbd_stream_t stream);

/* The same search with a HINT: hint[i * kk + t], t < kk = min(k, n - 1 (+1 if loop)), are kk distinct
 * neighbours of centre i from an earlier, similar configuration -- in a rollout, the previous step's own
 * edge_index row 0. The largest of their CURRENT distances bounds the kk-th neighbour distance, which saves
 * the first of the two candidate scans; the result is exactly that of nbd_knn_graph_f32 whatever the hint
 * holds (a hint that is not kk distinct valid neighbours is detected and ignored for that centre). hint may
 * alias edge_index (each centre reads its hint before writing its own list). No batch segments / out_off. */
int nbd_knn_graph_hint_f32(const float* pos, int n, int k, int loop, const int* seg_lo, const int* seg_hi,
                           const int64_t* out_off, int64_t num_edges, int64_t* edge_index, const int64_t* hint,
                           nbd_stream_t stream);

/* The radius search of a ROLLOUT (Trainer.evaluate_rollout calls the model once per step on a configuration that
 * has barely moved): same outputs as nbd_radius_search_f32 (no batch segments), exactly, but the O(n^2) scan runs only
 * when some body has moved more than sqrt(moved_sq) since the cached candidate lists were built -- lists of the first
 * wide_cap indices within sqrt(wide_radius_sq) of every centre, kept in `state` together with the reference
 * positions. Every call: one check launch, the (self-skipping) rebuild launches, one re-test launch (one wave per
 * centre over its cached list; a centre whose truncated list holds fewer than max_num_neighbors current hits scans
 * on behind the list). Exact as long as moved <= (wide_radius - radius) / 2; callers leave a margin (graphops.py uses
 * 0.45). state: nbd_radius_cached_state_bytes bytes, 64-byte aligned, ZEROED before its first use and kept by the
 * caller between calls; workspace: nbd_radius_cached_workspace_bytes (scratch). indeg as in nbd_radius_search_f32. */
size_t nbd_radius_cached_state_bytes(int n, int wide_cap);
size_t nbd_radius_cached_workspace_bytes(int n, int wide_cap);
int nbd_radius_cached_search_f32(const float* pos, int n, float radius_sq, float wide_radius_sq, float moved_sq, int loop,
                                 int max_num_neighbors, int wide_cap, void* state, size_t state_bytes, int* nbr, int* deg,
                                 int* last, int* indeg, void* workspace, size_t workspace_bytes, nbd_stream_t stream);

/* The transposed lists (rows = neighbour j, entries = the centres that list j, ascending) straight from the cache of
 * the nbd_radius_cached_search_f32 call that produced `last` on the SAME positions: rowptr = exclusive scan of that
 * call's indeg. Same result as nbd_radius_transpose_lists, without scatter, atomics or per-row sort. */
int nbd_radius_cached_transpose_f32(const float* pos, int n, float radius_sq, int loop, int wide_cap, const void* state,
                                    size_t state_bytes, const int* last, const int* rowptr, int* centres,
                                    nbd_stream_t stream);

/* radius_graph(pos, r, batch, loop, max_num_neighbors) -- contconv.py:225 -- in padded (ELL) form:
 * nbr[i][0..deg[i]) = the first max_num_neighbors indices j (ascending) with d2 < radius_sq
 * (strict), j == i iff loop; last[i] = the largest listed j (-1 if none). No host sync needed. */
int nbd_radius_search_f32(const float* pos, int n, float radius_sq, int loop, int max_num_neighbors,
                          const int* seg_lo, const int* seg_hi, int* nbr, int* deg, int* last, int* indeg,
                          nbd_stream_t stream);

/* radius_graph(loop = False) as torch_cluster 1.6.3 computes it (contconv.py:225 with self_loops = False): search with
 * self as a candidate and max_num_neighbors + 1 slots (the calls here with loop = 1), THEN drop row == col -- a centre
 * with >= max_num_neighbors + 1 lower-indexed hits keeps all max_num_neighbors + 1. This entry point does the drop on
 * lists of `cap` slots per centre: the self entry is removed, deg / last / indeg (optional) follow. (The `loop = 0`
 * mode of the search entry points -- self excluded BEFORE the cap -- is kept for callers that want that rule.) */
int nbd_radius_drop_self_i32(int* nbr, int* deg, int* last, int* indeg, int n, int cap, nbd_stream_t stream);

/* The same search in streaming form (lane = centre, sources broadcast through LDS, the source range cut
 * into slices whose per-centre hit lists are concatenated in index order by a second kernel): identical
 * outputs, ~10x faster at N = 16 384. Needs nbd_radius_search_workspace_bytes(n, max_num_neighbors). */
size_t nbd_radius_search_workspace_bytes(int n, int max_num_neighbors);
int nbd_radius_search_ws_f32(const float* pos, int n, float radius_sq, int loop, int max_num_neighbors,
                             const int* seg_lo, const int* seg_hi, int* nbr, int* deg, int* last, int* indeg,
                             void* workspace, size_t workspace_bytes, nbd_stream_t stream);
/* indeg (optional, int32 [n], zeroed by the call): indeg[j] = number of lists that hold j -- the
 * in-degree the transpose needs, counted with integer atomics while the lists are built.
 *
 * O(E) transpose of the lists (what the ContinuousConv pipeline uses): with rowptr = exclusive scan
 * of indeg, centres[rowptr[j] ..] = the centres c whose list holds j, ascending (scatter through an
 * atomic cursor, then a per-row rank sort: deterministic). cursor: int32 [n] scratch; scratch: int32
 * [>= E] scratch. Same result as the two nbd_radius_transpose_*_f32 scans below. */
int nbd_radius_transpose_lists(const int* nbr, const int* deg, int n, int cap, const int* rowptr, int* cursor,
                               int* scratch, int* centres, nbd_stream_t stream);

/* Transposed adjacency of those capped lists: indeg[j] = number of centres c whose list holds j;
 * after an exclusive scan, centres[rowptr[j] ..] = those c, ascending. This is the grouping
 * ContinuousConv aggregates over (scatter by edge_index[0], contconv.py:82,95-97). */
int nbd_radius_transpose_count_f32(const float* pos, int n, float radius_sq, int loop, const int* seg_lo,
                                   const int* seg_hi, const int* last, int* indeg, nbd_stream_t stream);
int nbd_radius_transpose_fill_f32(const float* pos, int n, float radius_sq, int loop, const int* seg_lo,
                                  const int* seg_hi, const int* last, const int* rowptr, int* centres,
                                  nbd_stream_t stream);

/* Edge list -> CSR grouped by `key` (a row of an int64 edge_index with values in [0, n)): rowptr[n+1],
 * out[e] = the `val` entries of each key in ascending order (duplicates kept). With key = edge_index[0]
 * (sources) and val = edge_index[1] (targets) this is the transposed adjacency that the backward pass of
 * a gather needs so that it is itself a gather with a fixed summation order (the reference's autograd
 * scatters with atomics: gnn.py:170,183 `loss.backward()`). cursor[n], scratch[n_edges]: caller-owned
 * temporaries; *bad_flag (device int) is set to 1 if any key is outside [0, n). */
int nbd_csr_by_key_i64(const int64_t* key, const int64_t* val, int64_t n_edges, int n, int* rowptr, int* cursor,
                       int* scratch, int* out, int* bad_flag, nbd_stream_t stream);

/* ptr[0] = 0, ptr[i+1] = ptr[i] + counts[i]  (int32, n counts -> n+1 entries). */
int nbd_exclusive_scan_i32(const int* counts, int n, int* ptr, nbd_stream_t stream);

/* rowptr[0 .. n] of an edge list whose targets are already in ascending order (knn_graph's output, and the collation of
 * such graphs): rowptr[i] = first edge with tgt >= i. One launch. */
int nbd_rowptr_sorted_i64(const int64_t* tgt, int64_t n_edges, int n, int* rowptr, nbd_stream_t stream);

/* ELL lists -> compact int64 edge_index[2][num_edges] (what radius_graph returns). */
int nbd_ell_to_edge_index(const int* nbr, const int* deg, const int* ptr, int n, int cap, int64_t num_edges,
                          int64_t* edge_index, nbd_stream_t stream);

/* ------------------------------------------------------------ surrogate models: dense blocks */

/* y[r][c] = act( rowscale[r] * sum_k x[r][k] w[c][k]  +  bias_rowscale[r] * bias[c] ),
 * act 0 = identity, 1 = tanh; bias, rowscale, bias_rowscale may be NULL (= 0, 1, 1).
 * torch.nn.Linear layout (w is out x in); ld* are row strides in floats, so inputs/outputs may be
 * column slices of wider buffers (replaces torch.cat). fp32-input MFMA, exact fp32 arithmetic.
 * Used for gnn.py:57-63,75-93,105-114 and contconv.py:92,136-141,206-216. Deterministic. */
int nbd_linear_f32(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowscale,
                   const float* bias_rowscale, int act, float* y, int ldy, int n_rows, int n_cols, int k,
                   void* workspace, size_t workspace_bytes, nbd_stream_t stream);
/* Scratch for the split-K path of large products (0 for small ones). Without it the call still
 * succeeds through the un-split kernel. */
size_t nbd_linear_workspace_bytes(int n_rows, int n_cols, int k);

/* EdgeConv aggregation (gnn.py:75-93) after the per-node factoring of its first Linear:
 * pq[i] = [P_i (h) | Q_i (h)], s[i] = aggr_j tanh(P_i + Q_j) over the edges of target i, i.e.
 * src[rowptr[i] .. rowptr[i+1]) (rowptr NULL: exactly fixed_k edges per node, i*fixed_k ..).
 * aggr 0 = sum, 1 = mean (sum / max(count,1)), 2 = max (no edges -> 0). */
int nbd_edgeconv_aggregate_f32(const float* pq, int ldpq, int h, const int* rowptr, const int64_t* src,
                               int fixed_k, int n, int aggr, float* s, int lds, nbd_stream_t stream);

/* EdgeConv with aggr = "max" (PyG's own default; gnn.py:79-93 passes the user's aggr through): the
 * second Linear cannot be hoisted out of a max, so messages are materialised per edge,
 * m[e] = tanh(P_tgt[e] + Q_src[e]) (edges grouped by target), nbd_linear_f32 runs over the E rows and
 * nbd_segment_reduce_f32 reduces each target's rows (mode 0 = sum, 1 = mean, 2 = max; empty -> 0; 3 = product in row
 * order, empty -> 1: torch_scatter's scatter(reduce="mul"), which contconv.py:95-97 reaches with agg="mul"). */
int nbd_edge_messages_f32(const float* pq, int ldpq, int h, const int64_t* src, const int64_t* tgt, int64_t n_edges,
                          float* m, int ldm, nbd_stream_t stream);
int nbd_segment_reduce_f32(const float* m, int ldm, int h, const int* rowptr, int n, int mode, float* out, int ldo,
                           nbd_stream_t stream);

/* torch.nn.LayerNorm over the last dim (gnn.py:146, contconv.py:233); gamma/beta may be NULL. */
int nbd_layernorm_f32(const float* x, int ldx, int c, const float* gamma, const float* beta, float eps, float* y,
                      int ldy, int n, nbd_stream_t stream);

/* LayerNorm + decoder in ONE launch: out = MLP(LayerNorm(x)) (gnn.py:105-114,146-148; contconv.py:206-216,233-234) for
 * c <= 256 channels, n_layers in 1..3 Linears with tanh between them, hidden widths <= 64, last width <= 8
 * (dims[0 .. n_layers], dims[0] = c). w[i] for the hidden layers (i < n_layers - 1) is the TRANSPOSED weight, [in][out]
 * contiguous; w[n_layers - 1] is out x in as torch.nn.Linear holds it; b[i] may be NULL. With hidden layers
 * (n_layers >= 2: fp32 MFMA, 16 rows per wave) the LayerNorm's affine part must come FOLDED into the first Linear --
 * w[0] = (W1 diag(gamma))^T, b[0] = b1 + W1 beta -- and gamma = beta = NULL; with n_layers = 1 gamma / beta are applied
 * here. kick_vel (n x last width,
 * contiguous) non-NULL: kick_vel += kick_c * out in the epilogue (Trainer.step's second half-kick, trainer.py:225-226).
 * nbd_ln_mlp_head_lds_bytes returns 0 for shapes it does not cover (the caller then runs nbd_layernorm_f32 +
 * nbd_linear_f32). Host arrays w / b / dims. */
size_t nbd_ln_mlp_head_lds_bytes(int c, int n_layers, const int* dims);
int nbd_ln_mlp_head_f32(const float* x, int ldx, int c, const float* gamma, const float* beta, float eps, int n_layers,
                        const float* const* w, const float* const* b, const int* dims, float* out, int ldout,
                        float* kick_vel, float kick_c, int n, nbd_stream_t stream);

/* ContinuousConv.forward (contconv.py:80-98) feature-side binning:
 * a_out[n][cell][i] = sum over edges e with edge_index[0][e] == n of
 *      window_e * trilinear_weight_e(cell) * feat[centre_e][i],   cell = (z*D + y)*D + x,
 * window/ball_to_cube/grid_sample(align_corners=True) exactly as contconv.py:30-33,53-78,85-90, so
 * that ContinuousConv = scatter_mean(...) = rowscale * (a_out . filters.reshape(D^3*I, O)).
 * rowptr/centres: CSR by aggregation target (nbd_radius_transpose_*). D <= 10.
 * Rows [node_begin, node_begin + n) are produced into a_out[0 .. n).
 * cell_map (NULL = identity, cells_out ignored): int[D^3], cell -> column block of a_out or -1. ball_to_cube
 * keeps every sample inside |mapped| < tanh(R) (contconv.py:30-33 with the window cut at R, :85-87), so grid
 * points further than that from the cube centre are never touched: their columns are structurally zero and
 * the caller may drop them from a_out AND from the filter matrix (D = 6, R = 1: 160 of 216 cells remain);
 * a_out is then (n, cells_out * in_channels). */
int nbd_contconv_bin_f32(const float* pos, const float* feat, int ldf, int in_channels, const int* rowptr,
                         const int* centres, int node_begin, int n, int filter_resolution, float radius_sq,
                         const int* cell_map, int cells_out, float* a_out, nbd_stream_t stream);

/* One whole EdgeConv layer of GraphModel.forward (gnn.py:75-93,140-148) in ONE launch, one node per
 * wave: s_i = aggr_j tanh(P_i + Q_j); y_i = W2 s_i + beta_i b2 (beta = [deg>0] for mean, deg for sum);
 * then the epilogue. At the reference's sizes the forward pass is launch/latency bound, so this
 * replaces 3 launches per layer (and LayerNorm + head) of the general path; same arithmetic. */
#define NBD_GNN_WRITE_X 0    /* out[i][0..h)   = y_i                      (ldout: column slice ok)    */
#define NBD_GNN_NEXT_PQ 1    /* out[i][0..ep_out) = w_ep y_i + b_ep       (next layer's [P|Q])        */
#define NBD_GNN_FINAL_HEAD 2 /* out[i][0..ep_out) = w_ep LayerNorm([enc_i || y_i]) + b_ep, ep_out<=8 */
#define NBD_GNN_FINAL_LN 3   /* out[i][0..e+h) = LayerNorm([enc_i || y_i])                            */
#define NBD_GNN_NEXT_PQ_FOLDED 4 /* out[i][0..ep_out) = (w_ep W2) S_i + beta_i (w_ep b2) + b_ep: as NEXT_PQ with
                                  * the two mat-vecs folded on the host; the caller passes w_ep := (w_ep W2)^T
                                  * [h][ep_out], b2 := w_ep b2 [ep_out]; w2t is not read                   */
typedef struct nbd_gnn_layer_args {
  const int* rowptr;      /* [n+1] edges grouped by target, or NULL: exactly fixed_k edges per node */
  const int64_t* src;     /* source node j of every edge (edge_index[0])                            */
  int fixed_k, n;
  const float* pq;        /* [n][ldpq] = [P (h) | Q (h)], or NULL: form P/Q from x on the fly       */
  int ldpq;
  const float* x;         /* [n][ldx], first f columns (f <= 8) -- used when pq == NULL             */
  int ldx, f;
  const float* wpq;       /* [2h][f] rows P then Q = [W1a - W1b ; W1b]; bpq [h] = b1                */
  const float* bpq;
  int h, aggr;            /* channels (<= 128); 0 = sum, 1 = mean                                    */
  const float* w2t;       /* [h][h] = W2 TRANSPOSED (in x out: w2t[k][o] = W2[o][k]), b2 [h]          */
  const float* b2;
  int epilogue;           /* NBD_GNN_*                                                               */
  const float* w_ep;      /* NEXT_PQ: [h][ep_out] (TRANSPOSED);  FINAL_HEAD: [ep_out][e+h]          */
  const float* b_ep;
  int ep_out;
  const float* enc;       /* FINAL_*: encoder output [n][ldenc], e columns (e <= 256)                */
  int ldenc, e;
  const float* ln_g;      /* LayerNorm weight / bias [e+h]                                           */
  const float* ln_b;
  float ln_eps;
  float* out;
  int ldout;
  float* kick_vel;        /* FINAL_HEAD only, or NULL: vel[n][ep_out] += kick_c * out_i, the second half-kick of  */
  float kick_c;           /* Trainer.step (trainer.py:225) in the layer's epilogue (multiply, then add)           */
  /* Exponential tables (optional, h <= 64): tanh(P_i + Q_j) = 1 - 2 / (EP_i EQ_j + 1) with EP = 2^(c P), EQ = 2^(c Q),
   * c = 2 log2 e -- one reciprocal per edge and channel instead of an exponential and a reciprocal, the edge loop's
   * bound. epq [n][ldepq] = [EP (h) | EQ (h)] belongs to the SAME P/Q that pq (or x, wpq, bpq) describe: an entry whose
   * |c v| exceeds 100 holds NaN, and a node that meets one is recomputed from pq / x exactly as without tables, so the
   * result never depends on the tables' range. out_epq (NEXT_PQ*): the epilogue also writes the table of its `out`. */
  const float* epq;
  int ldepq;
  float* out_epq;
  int ldout_epq;
  /* Pre-advance (optional; FINAL_HEAD with ep_out == 3, kick_vel set, h == 64 -- NBD_E_UNSUPPORTED otherwise): the
   * leapfrog bookkeeping of a rollout step (trainer.py:217-227) in the epilogue, so that a captured step is the search
   * and the layers only. With vh = adv_vel_half[i] (the half-kicked velocity this step's positions were drifted with),
   * x = adv_pos[i] (those positions: what the search saw) and a = out[i]:
   *   kick_vel[i] = vh + kick_c a        this step's velocity (written, not accumulated)
   *   adv_pos_out[i] = x                 this step's position
   *   adv_vel_half[i] = kick_vel[i] + kick_c a ;  adv_pos[i] = x + adv_dt adv_vel_half[i]    the next step's first half,
   * the new position also into the first three columns of adv_posm[i] (rows of 4: the packed model input). Every product
   * and sum is rounded separately, as nbd_kick_f32 / nbd_kick_drift_f32 round them: the trajectory is bit for bit the one
   * of the separate launches. Only node i's own rows are touched. */
  float* adv_vel_half;
  float* adv_pos;
  float* adv_posm;
  float* adv_pos_out;
  float adv_dt;
} nbd_gnn_layer_args;
int nbd_gnn_layer_f32(const nbd_gnn_layer_args* args, nbd_stream_t stream);

/* GraphModel.predict (gnn.py:205-215: transform_to_graph :11-22, then forward :130-148) as ONE call: the kNN graph of
 * `pos` (k nearest, ascending (d2, j), self excluded unless loop -- nbd_knn_graph_hint_f32's rule) into edge_index
 * [2][n * min(k, n - 1 + loop)], then layers[0 .. n_layers) through nbd_gnn_layer_f32 on that graph (the callee sets
 * each layer's rowptr / src / fixed_k / n). use_hint: edge_index's previous content (an earlier result for a similar
 * configuration, e.g. the previous rollout step) bounds the search; the result does not depend on it. */
#define NBD_GNN_MAX_LAYERS 8
typedef struct nbd_gnn_forward_args {
  const float* pos;       /* [n][3] */
  int n, k, loop, use_hint;
  int64_t* edge_index;    /* [2][n * kk] */
  int n_layers;
  nbd_gnn_layer_args layers[NBD_GNN_MAX_LAYERS];
  void* workspace;        /* optional, nbd_gnn_forward_workspace_bytes(): room for the layers' exponential tables (see  */
  size_t workspace_bytes; /* nbd_gnn_layer_args.epq); with it and h <= 64, first layer formed from x, n <= 8192, the    */
                          /* search kernel also emits the first layer's tables (nbd_knn_graph_hint_pq_f32) and every    */
                          /* NEXT_PQ* epilogue the next one's. NULL / too small: the layers run without tables.         */
} nbd_gnn_forward_args;
size_t nbd_gnn_forward_workspace_bytes(const nbd_gnn_forward_args* args);    /* 0: this configuration uses no tables */
int nbd_gnn_forward_f32(const nbd_gnn_forward_args* args, nbd_stream_t stream);

/* nbd_knn_graph_hint_f32 for ONE un-segmented system (n <= 8192, k <= 200, pos 16-byte aligned; NBD_E_UNSUPPORTED
 * otherwise, nothing launched) whose search kernel, one wave per centre, also writes that centre's row of the first
 * EdgeConv layer's tables: epq[i] = [2^(c P_i) | 2^(c Q_i)], P = wpq[0:h] x_i + bpq, Q = wpq[h:2h] x_i (the fma order of
 * nbd_gnn_layer_f32's on-the-fly form; h <= 64, f <= 8; |c v| > 100 -> NaN, see nbd_gnn_layer_args.epq). Replaces
 * torch_cluster.knn (gnn.py:13) plus the first Linear of gnn.py:140-141 for the rollout. */
typedef struct nbd_knn_pq_args {
  const float* x;         /* [n][ldx], first f columns */
  int ldx, f, h;
  const float* wpq;       /* [2h][f] */
  const float* bpq;       /* [h] */
  float* epq;             /* [n][ldepq] = [EP (h) | EQ (h)] */
  int ldepq;
} nbd_knn_pq_args;
int nbd_knn_graph_hint_pq_f32(const float* pos, int n, int k, int loop, int64_t num_edges, int64_t* edge_index,
                              const int64_t* hint, const nbd_knn_pq_args* pq, nbd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Backward kernels: what `loss.backward()` executes in the reference's training step
 * (gnn.py:163-191 train_batch / train_graph_batch, contconv.py:242-247, driven by trainer.py:60-72)
 * for the layers above. All sums run in a fixed order (row slabs reduced by a second kernel, gathers
 * over sorted adjacency lists): gradients are bit-identical run to run, unlike autograd's atomics.
 * The data gradient of a Linear (dX = g W) is nbd_linear_f32 with the transposed weight.
 * ------------------------------------------------------------------------------------------------- */

/* g = rowscale[n] * dy * act'(y): act 0 = identity, 1 = tanh (1 - y^2); y is the forward OUTPUT;
 * rowscale NULL = 1 (the 1/deg of a mean aggregation, contconv.py:95-97). */
int nbd_act_bwd_f32(const float* dy, int lddy, const float* y, int ldy, int act, const float* rowscale, float* g,
                    int ldg, int n, int c, nbd_stream_t stream);

/* out[c] = sum_n rowweight[n] * x[n][c] (rowweight NULL = 1): bias gradients. */
size_t nbd_colsum_workspace_bytes(int n, int c);
int nbd_colsum_f32(const float* x, int ldx, const float* rowweight, int n, int c, float* out, void* workspace,
                   size_t workspace_bytes, nbd_stream_t stream);

/* dW[m][k] = sum_n g[n][m] * x[n][k]: the weight gradient of Y = X W^T (W is m x k), fp32 MFMA. */
size_t nbd_linear_wgrad_workspace_bytes(int n, int m, int k);
int nbd_linear_wgrad_f32(const float* g, int ldg, const float* x, int ldx, int n, int m, int k, float* dw, int lddw,
                         void* workspace, size_t workspace_bytes, nbd_stream_t stream);
/* The same with the bias gradient in the same launches: db[m] = sum_n rowweight[n] * g[n][m] (rowweight NULL = 1) as one
 * more column of the product (what nbd_colsum_f32 would return; its own summation order, fixed). */
size_t nbd_linear_wgrad_bias_workspace_bytes(int n, int m, int k);
int nbd_linear_wgrad_bias_f32(const float* g, int ldg, const float* x, int ldx, const float* rowweight, int n, int m, int k,
                              float* dw, int lddw, float* db, void* workspace, size_t workspace_bytes, nbd_stream_t stream);

/* Backward of nbd_edgeconv_aggregate_f32 for aggr 0 (sum) / 1 (mean): dpq[n][2h] = [dP | dQ] from ds[n][h].
 * (rowptr | fixed_k, src): the forward's by-target lists; (rowptr_t, tgt_t): the same edges grouped by
 * source (nbd_csr_by_key_i64). */
int nbd_edgeconv_aggregate_bwd_f32(const float* pq, int ldpq, int h, const float* ds, int ldds, const int* rowptr,
                                   const int64_t* src, int fixed_k, const int* rowptr_t, const int* tgt_t, int n,
                                   int aggr, float* dpq, int lddpq, nbd_stream_t stream);

/* torch.nn.BatchNorm1d in TRAINING mode fused with the activation that follows it in the PyG MLP
 * (contconv.py:136-141): y = act(gamma (x - mean) rstd + beta) with the batch mean / biased variance per
 * column, which are also returned (the caller updates running_mean / running_var). n >= 2 as in torch.
 * bwd: dx, dgamma, dbeta from dy (and y when act = tanh). */
size_t nbd_batchnorm_train_workspace_bytes(int n, int c);
int nbd_batchnorm_train_fwd_f32(const float* x, int ldx, int n, int c, const float* gamma, const float* beta, float eps,
                                int act, float* y, int ldy, float* mean, float* var, float* rstd, void* workspace,
                                size_t workspace_bytes, nbd_stream_t stream);
int nbd_batchnorm_train_bwd_f32(const float* x, int ldx, int n, int c, const float* gamma, const float* mean,
                                const float* rstd, int act, const float* y, int ldy, const float* dy, int lddy,
                                float* dx, int lddx, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, nbd_stream_t stream);

/* Adjoint of nbd_contconv_bin_f32 with respect to the features: dfeat[c][i] = sum over edges (n <- c) of
 * window * sum_corners t_corner * da[n][cell][i], gathered per SOURCE c over its list of targets n:
 * CSR (rowptr_s, tgt_s) or, with rowptr_s NULL, padded lists tgt_s[c * cap + 0..deg[c]) -- the layout
 * nbd_radius_search_f32 produces. da is (n, cells_out * in_channels) contiguous; cell_map / cells_out as in
 * nbd_contconv_bin_f32. */
int nbd_contconv_bin_bwd_f32(const float* pos, const float* da, int in_channels, const int* rowptr_s, const int* tgt_s,
                             const int* deg, int cap, int n, int filter_resolution, float radius_sq,
                             const int* cell_map, int cells_out, float* dfeat, int lddf, nbd_stream_t stream);

/* Backward of nbd_segment_reduce_f32 mode 2 (max): dm[e][c] = dx[i][c] at the first row e of target i with
 * m[e][c] == x[i][c] (x = the forward output), 0 elsewhere. m may have zero rows for a target. */
int nbd_segment_max_bwd_f32(const float* m, int ldm, int h, const float* x, int ldx, const int* rowptr, int n,
                            const float* dx, int lddx, float* dm, int lddm, nbd_stream_t stream);

/* Backward of nbd_segment_reduce_f32 mode 3 (product): dm[e][c] = dx[i][c] * prod over the OTHER rows e' of target i of
 * m[e'][c] (prefix times suffix: exact when a row holds zeros). */
int nbd_segment_mul_bwd_f32(const float* m, int ldm, int h, const int* rowptr, int n, const float* dx, int lddx,
                            float* dm, int lddm, nbd_stream_t stream);

/* Backward of nbd_layernorm_f32: dx[n][c], dgamma[c], dbeta[c] from x, gamma (NULL = 1) and dy. */
size_t nbd_layernorm_bwd_workspace_bytes(int n, int c);
int nbd_layernorm_bwd_f32(const float* x, int ldx, int c, const float* gamma, float eps, const float* dy, int lddy,
                          float* dx, int lddx, float* dgamma, float* dbeta, int n, void* workspace,
                          size_t workspace_bytes, nbd_stream_t stream);

/* ---- GraphModel's training step with the whole model behind one call per direction (csrc/train_model.hip):
 * forward = gnn.py:130-148 (node encoder MLP or none, message_passing_steps EdgeConv layers with aggr sum / mean in the
 * per-node factored form, LayerNorm over [encoder output | last layer], Linear or MLP decoder), keeping what the
 * backward pass needs in `workspace`; backward = what `loss.backward()` computes for every parameter (gnn.py:170,183,
 * 189), from d loss / d out. Same kernels and summation orders as the per-layer entry points above (bit-reproducible).
 * Weights in torch.nn.Linear layout (out x in, contiguous); w1[l] is the EdgeConv MLP's first Linear (h x 2 F_l),
 * w2[l] its second (h x h). enc_dim[0 .. n_enc] / head_dim[0 .. n_head]: layer widths (tanh between the Linears, none
 * after the last), enc_dim[0] = f, head_dim[0] = encoder output + h. (rowptr | fixed_k, src): edges grouped by target as
 * for nbd_edgeconv_aggregate_f32; (rowptr_t, tgt_t): the same edges grouped by source (nbd_csr_by_key_i64), read by the
 * backward pass only. The same args struct (same workspace) goes to both calls. n = 0: NBD_E_UNSUPPORTED. */
#define NBD_TRAIN_MAX_MLP 8
typedef struct nbd_gnn_train_args {
  int n; const int* rowptr; const int64_t* src; int fixed_k; const int* rowptr_t; const int* tgt_t; int aggr;
  const float* x; int ldx; int f;
  int n_enc; const float* enc_w[NBD_TRAIN_MAX_MLP]; const float* enc_b[NBD_TRAIN_MAX_MLP]; int enc_dim[NBD_TRAIN_MAX_MLP + 1];
  int n_layers; int h;
  const float* w1[NBD_GNN_MAX_LAYERS]; const float* b1[NBD_GNN_MAX_LAYERS];
  const float* w2[NBD_GNN_MAX_LAYERS]; const float* b2[NBD_GNN_MAX_LAYERS];
  const float* ln_g; const float* ln_b; float ln_eps;
  int n_head; const float* head_w[NBD_TRAIN_MAX_MLP]; const float* head_b[NBD_TRAIN_MAX_MLP]; int head_dim[NBD_TRAIN_MAX_MLP + 1];
  float* out; int ldout;
  void* workspace; size_t workspace_bytes;
} nbd_gnn_train_args;
typedef struct nbd_gnn_train_grads {
  float* enc_w[NBD_TRAIN_MAX_MLP]; float* enc_b[NBD_TRAIN_MAX_MLP];
  float* w1[NBD_GNN_MAX_LAYERS]; float* b1[NBD_GNN_MAX_LAYERS]; float* w2[NBD_GNN_MAX_LAYERS]; float* b2[NBD_GNN_MAX_LAYERS];
  float* ln_g; float* ln_b;
  float* head_w[NBD_TRAIN_MAX_MLP]; float* head_b[NBD_TRAIN_MAX_MLP];
} nbd_gnn_train_grads;
size_t nbd_gnn_train_workspace_bytes(const nbd_gnn_train_args* args);
int nbd_gnn_train_forward_f32(const nbd_gnn_train_args* args, nbd_stream_t stream);
int nbd_gnn_train_backward_f32(const nbd_gnn_train_args* args, const float* dout, int lddout,
                               const nbd_gnn_train_grads* grads, nbd_stream_t stream);

/* ---- ContinuousConvModel's training step the same way (contconv.py:218-247; csrc/train_model.hip): node encoder (PyG MLP:
 * Linear -> BatchNorm1d on BATCH statistics (enc_bn = 1; running statistics updated in place when bn_rmean / bn_rvar are
 * given, momentum bn_momentum) -> tanh per hidden layer, plain last Linear; enc_bn = 0: no norm; n_enc = 0: no encoder), the
 * ContinuousConv layers (agg sum / mean, tanh) over the caller's pair lists -- pairs_fwd[l] / pairs_adj[l] from
 * nbd_contconv_pairs_jobs_f32 over the forward (rowptr_fwd, cap_fwd) and adjoint (rowptr_adj, cap_adj) groupings, per
 * layer's filter resolution; pairs_adj[0] may be NULL without an encoder --, LayerNorm over [encoder output | last layer],
 * decoder MLP. filt[l]: the layer's (cells_total, in, out) filters as the module holds them; kept[l] (int64 [n_cells]) /
 * cell_map[l] (int32 [cells_total]) the reachable-cell maps; scale: the 1 / in-degree row scale of a mean aggregation or
 * NULL. Shapes outside nbd_contconv_fused_supported in either direction: workspace_bytes() returns 0 (the caller keeps the
 * per-layer path). Same args struct and workspace for both calls. */
typedef struct nbd_cc_train_args {
  int n; const float* x; int ldx; int in_ch;
  int n_enc; const float* enc_w[NBD_TRAIN_MAX_MLP]; const float* enc_b[NBD_TRAIN_MAX_MLP]; int enc_dim[NBD_TRAIN_MAX_MLP + 1];
  int enc_bn; const float* bn_g[NBD_TRAIN_MAX_MLP]; const float* bn_b[NBD_TRAIN_MAX_MLP]; float bn_eps[NBD_TRAIN_MAX_MLP];
  float* bn_rmean[NBD_TRAIN_MAX_MLP]; float* bn_rvar[NBD_TRAIN_MAX_MLP]; float bn_momentum[NBD_TRAIN_MAX_MLP];
  int n_layers; int cdim;
  const float* filt[NBD_GNN_MAX_LAYERS]; const int64_t* kept[NBD_GNN_MAX_LAYERS]; const int* cell_map[NBD_GNN_MAX_LAYERS];
  int n_cells[NBD_GNN_MAX_LAYERS]; int cells_total[NBD_GNN_MAX_LAYERS];
  const void* pairs_fwd[NBD_GNN_MAX_LAYERS]; const void* pairs_adj[NBD_GNN_MAX_LAYERS];
  const int* rowptr_fwd; int64_t cap_fwd; const int* rowptr_adj; int64_t cap_adj;
  const float* scale;
  const float* ln_g; const float* ln_b; float ln_eps;
  int n_head; const float* head_w[NBD_TRAIN_MAX_MLP]; const float* head_b[NBD_TRAIN_MAX_MLP]; int head_dim[NBD_TRAIN_MAX_MLP + 1];
  float* out; int ldout;
  void* workspace; size_t workspace_bytes;
} nbd_cc_train_args;
typedef struct nbd_cc_train_grads {
  float* enc_w[NBD_TRAIN_MAX_MLP]; float* enc_b[NBD_TRAIN_MAX_MLP]; float* bn_g[NBD_TRAIN_MAX_MLP]; float* bn_b[NBD_TRAIN_MAX_MLP];
  float* filt[NBD_GNN_MAX_LAYERS];
  float* ln_g; float* ln_b;
  float* head_w[NBD_TRAIN_MAX_MLP]; float* head_b[NBD_TRAIN_MAX_MLP];
} nbd_cc_train_grads;
size_t nbd_cc_train_workspace_bytes(const nbd_cc_train_args* args);
int nbd_cc_train_forward_f32(const nbd_cc_train_args* args, nbd_stream_t stream);
int nbd_cc_train_backward_f32(const nbd_cc_train_args* args, const float* dout, int lddout, const nbd_cc_train_grads* grads,
                              nbd_stream_t stream);

/* ---- ContinuousConv.forward (contconv.py:80-98), block-sparse and fused: only the (node, filter cell) blocks
 * some edge touches are multiplied, and the binned features never reach HBM (csrc/contconv_fused.hip).
 * For in_channels % 4 == 0, in_channels <= 128 (nbd_contconv_fused_supported), inference and training alike (the
 * backward pass: nbd_contconv_pairs_jobs_f32 + nbd_contconv_filter_grad_f32 below); other shapes use
 * nbd_contconv_bin_f32 + nbd_linear_f32.
 *
 * nbd_contconv_pairs_f32: the per-(tile of NBD_CC_TILE nodes, cell) lists of (edge corner) pairs {source node,
 *   window * trilinear weight}, grouped by node -- geometry of contconv.py:84-90 evaluated once per edge and
 *   filter resolution. rowptr/centres: edges grouped by aggregation target (edge_index[0]); centres[e] =
 *   edge_index[1]. cell_map (int32 [D^3], -1 = cell dropped) or NULL; n_cells = cells kept. edge_capacity >=
 *   rowptr[n] sizes the lists (host-known bound, e.g. n * max_num_neighbors: no device read-back).
 * nbd_contconv_fused_f32: out[n][:] = act(rowscale[n] * sum_cells A[n][cell] . F[cell]) with
 *   filters_shuffled = the (cell, in, out) filters as nbd_contconv_shuffle_filters_f32 writes them: every element split
 *   into three bf16 terms (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even: together
 *   the 24 significant bits of x), in the fragment order of v_mfma_f32_16x16x32_bf16 -- 16-byte quad index
 *   (((cell * ceil(O/16) + cb) * 4 + s) * 3 + u) * 64 + lane holds term u (0 lo, 1 mid, 2 hi) of
 *   F[cell][32 s + 8 (lane >> 4) + e][16 cb + (lane & 15)], e = 0 .. 7; zero beyond I / O (nbd_contconv_filter_floats
 *   counts it in floats: 4 per quad). act: 0 none, 1 tanh. rowscale may be NULL. Deterministic.
 *   Arithmetic: fp32-equivalent. The features are split the same way on chip and the six term products of order <= 2^-16
 *   (all but mid.lo, lo.mid, lo.lo) are accumulated in fp32 on the bf16 matrix pipe; the row error against an fp64
 *   product is no larger than that of the fp32 matrix instruction (tests/test_surrogate_gpu.py holds it to that).
 * Limit: a node with more than 65 535 edges (the pair kernel counts pairs per (node, cell) in 16 bits) is not
 *   processed -- its whole tile of NBD_CC_TILE nodes comes out as NaN, loudly, instead of being summed wrongly. */
int nbd_contconv_fused_supported(int in_channels, int out_channels, int n_cells);
size_t nbd_contconv_pairs_bytes(int n, int64_t edge_capacity, int n_cells);
int nbd_contconv_pairs_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                           int filter_resolution, float radius_sq, const int* cell_map, int n_cells,
                           void* pair_lists, size_t pair_lists_bytes, nbd_stream_t stream);
/* The pair lists of up to NBD_CC_MAX_RES filter resolutions of ONE graph in a single launch (the layers of a
 * model share the graph and differ in D). Host arrays of n_res entries each. */
int nbd_contconv_pairs_batch_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                                 float radius_sq, int n_res, const int* filter_resolutions, const int* const* cell_maps,
                                 const int* n_cells, void* const* pair_lists, const size_t* pair_lists_bytes,
                                 nbd_stream_t stream);
/* The general form: up to NBD_CC_MAX_RES pair-list jobs over the same n nodes in ONE launch, each with its own
 * grouping of the edges -- what a training step needs (contconv.py:236-247): the FORWARD lists (rows = aggregation
 * targets edge_index[0], listed nodes = feature sources edge_index[1]) and the ADJOINT lists (rows = feature sources,
 * listed nodes = aggregation targets; adjoint = 1 negates the relative position, which is exact, so an edge gets
 * bit-identical cells and weights in both groupings). With the adjoint lists the gradient with respect to the
 * features is the forward kernel itself: dfeat = nbd_contconv_fused_f32(g, adjoint lists, filters transposed per
 * cell), g = scale * act'(out) * dout. rowptr [n + 1] gives every row's first edge; deg NULL = rows end where the next
 * begins (CSR), else rows are padded (ELL: rowptr[i] = i * cap, deg[i] valid entries -- the radius search's own
 * per-centre lists, unchanged). */
typedef struct nbd_cc_pairs_job {
  const int* rowptr; const int* centres; const int* deg;
  int64_t edge_capacity;
  int filter_resolution; const int* cell_map; int n_cells;
  int adjoint;
  void* pair_lists; size_t pair_lists_bytes;
} nbd_cc_pairs_job;
int nbd_contconv_pairs_jobs_f32(const float* pos, int n, float radius_sq, int n_jobs, const nbd_cc_pairs_job* jobs,
                                nbd_stream_t stream);
/* Gradient of ContinuousConv.forward with respect to `filters` (contconv.py:92-98 under loss.backward()) over the
 * forward pair lists: dfilters[cell][i][o] = sum over the touched (node, cell) blocks of A[node][cell][i] * g[node][o]
 * (cells = the n_cells kept ones, compact order; row-major in x out per cell). fp32 MFMA, the binned matrix is not
 * formed, partial sums per slab of tiles added in fixed order: deterministic. in_channels even, <= 128;
 * out_channels <= 128. */
size_t nbd_contconv_filter_grad_workspace_bytes(int n, int n_cells, int in_channels, int out_channels);
int nbd_contconv_filter_grad_f32(const float* feat, int ldf, int in_channels, const float* g, int ldg, int out_channels,
                                 const int* rowptr, int n, int64_t edge_capacity, const void* pair_lists, int n_cells,
                                 float* dfilters, void* workspace, size_t workspace_bytes, nbd_stream_t stream);
/* The same gradient laid out over the FULL filter grid (cells_total = D^3 cells of in x out; cell_map int32
 * [cells_total] -> compact cell or -1, NULL when every cell is kept): unreachable cells get exact zeros -- the
 * reference's (D, D, D, in, out) `.grad` in one call. workspace >= n_cells * in * out * 4 +
 * nbd_contconv_filter_grad_workspace_bytes(...) bytes. */
int nbd_contconv_filter_grad_full_f32(const float* feat, int ldf, int in_channels, const float* g, int ldg, int out_channels,
                                      const int* rowptr, int n, int64_t edge_capacity, const void* pair_lists, int n_cells,
                                      const int* cell_map, int cells_total, float* dfilters_full, void* workspace,
                                      size_t workspace_bytes, nbd_stream_t stream);
/* filters (cells_total, in, out) row-major -> filters_shuffled, the bf16 x 3 fragment order nbd_contconv_fused_f32 reads
 * (above), over the kept cells kept_cells[0 .. n_cells) (int64 indices into the full grid, ascending). transposed = 1
 * re-lays every cell's filter TRANSPOSED (the operand of the feature gradient: in / out swapped).
 * nbd_contconv_filter_floats floats out (16-byte aligned). Once per weight update. */
int nbd_contconv_shuffle_filters_f32(const float* filters, const int64_t* kept_cells, int n_cells, int in_channels,
                                     int out_channels, int transposed, float* filters_shuffled, nbd_stream_t stream);
/* Byte offsets of the sections of a pair-list buffer (for reports and tests; the layout is otherwise opaque):
 * [0] per-(tile, cell) descriptors, [1] rows, [2] pair records (int2 {source, weight bits}), [3] = [4], [4] step records (int4 per 16-row
 * step), [5] steps per tile (int32 [tiles], tiles = ceil(n / NBD_CC_TILE)), [6] cost per tile (int32 [tiles]),
 * [7] total bytes, [8] float [n]: 1 / max(in-degree, 1), the row scale of a mean aggregation (what
 * nbd_degree_scale_f32 mode 0 returns, written by the pair kernel as a by-product). offsets: 9 entries. The fused
 * kernel multiplies 16 x in x out per step. */
int nbd_contconv_pairs_layout(int n, int64_t edge_capacity, int n_cells, size_t* offsets);
size_t nbd_contconv_fused_workspace_bytes(int n, int n_cells, int out_channels);
size_t nbd_contconv_filter_floats(int in_channels, int out_channels, int n_cells);
int nbd_contconv_fused_f32(const float* feat, int ldf, int in_channels, const int* rowptr, int n, int64_t edge_capacity,
                           const void* pair_lists, const float* filters_shuffled, int n_cells, int out_channels,
                           const float* rowscale, int act, float* out, int ldo, void* workspace,
                           size_t workspace_bytes, nbd_stream_t stream);

/* ContinuousConv's public helpers (contconv.py:30-33 and 53-78), as methods of the drop-in layer:
 * nbd_ball_to_cube_f32: out[i] = r[i] / (|r[i]| + 1e-8) * tanh |r[i]|, r and out (n, 3) row-major.
 * nbd_trilinear_interpolate_f32: out (n, in, out) = the filters (D, D, D, in, out) blended at coords (n, 3) in grid units
 * [0, D - 1] exactly as the reference's F.grid_sample call does it (align_corners = True, zero padding outside the grid,
 * coordinate component 0 along the LAST filter axis). */
int nbd_ball_to_cube_f32(const float* r, int n, float* out, nbd_stream_t stream);
int nbd_trilinear_interpolate_f32(const float* filters, int filter_resolution, int in_channels, int out_channels,
                                  const float* coords, int n, float* out, nbd_stream_t stream);

/* scale[i] from the CSR degree d_i: mode 0 = 1/max(d,1), 1 = d, 2 = (d > 0). */
int nbd_degree_scale_f32(const int* rowptr, int n, int mode, float* scale, nbd_stream_t stream);

/* ------------------------------------------------------------ initial-condition generators on the device
 * generate_disk / generate_spiral (src/galaxify/galaxies.py:54-192, 195-296) AFTER their random draws, in fp64 as
 * upstream. The draws stay on the host: they come from NumPy's legacy global stream and must be consumed in the
 * reference's order for a seed to reproduce its galaxy. generate_disk's O(N^2) enclosed-mass loop (:143-152) is a
 * radix sort by radius + prefix sum + lower-bound search.
 *   u_r, u_z, u_phi   device double[n]: the three vector draws of generate_disk in the reference's order --
 *                     uniform(eps32, 1), uniform(-1, 1), rand()  (:98-112)
 *   rot3x3            device double[9], row-major: positions/velocities are multiplied as row vectors (v @ rot),
 *                     rot = Rx^T Ry^T Rz^T of the Euler angles (:160-186)
 *   offset3_host, initial_vel3_host   HOST double[3] (read at call time)
 *   pos, vel (n,3) and mass (n) device double outputs; body 0 is the central black hole. */
size_t nbd_disk_workspace_bytes(int n);
int nbd_disk_from_draws_f64(const double* u_r, const double* u_z, const double* u_phi, int n, double total_mass,
                            double radial_scale, double height_scale, double g_const, double black_hole_mass,
                            int clockwise, const double* rot3x3, const double* offset3_host,
                            const double* initial_vel3_host, double* pos, double* vel, double* mass, void* workspace,
                            size_t workspace_bytes, nbd_stream_t stream);
/* raw: device double[(n-1)*6], per star {gamma radius, rand, normal z, normal v_R, normal v_phi, normal v_z} in the
 * order the reference draws them (:245-262). */
int nbd_spiral_from_draws_f64(const double* raw, int n, double total_mass, double radial_scale, double height_scale,
                              double g_const, double black_hole_mass, int n_arms, double pitch_angle, double arm_strength,
                              double* pos, double* vel, double* mass, nbd_stream_t stream);

/* ------------------------------------------------------------ dataset CSV rows (host code, no device work)
 * The reference writes one row per particle per state with csv.DictWriter (src/s01-dataset-generation.py:218-241);
 * the nine state columns are numpy.float32 scalars, which the csv module prints with str(): the shortest decimal
 * that reads back as the same fp32, positional for 1e-4 <= |x| < 1e16, d.ddde-XX otherwise.
 *   nbd_format_f32        that string for one value into out24 (>= 24 bytes, not NUL-terminated); returns its length
 *   nbd_format_f32_array  n values into fixed slots of `slot` (>= 24) bytes each, NUL-padded (a numpy 'S<slot>' array)
 *   nbd_csv_format_state  the n rows of one state:  prefix + mass_i + ",x,y,z,vx,vy,vz,ax,ay,az" + suffix, where
 *                         prefix = "scene,scene_type,step,step_time," and suffix = ",u,k\r\n" are the per-state
 *                         constants the caller printed, mass_i = mass_chars[mass_off[i] .. mass_off[i+1]) the per-body
 *                         float64 strings (printed once per scene), pos / vel / acc HOST (n,3) fp32. Returns the bytes
 *                         written, or -1 on a bad argument / cap < nbd_csv_state_bound(...). */
int nbd_format_f32(float x, char* out24);
int nbd_format_f32_array(const float* x, int64_t n, char* out, int slot);
size_t nbd_csv_state_bound(int n, size_t prefix_len, size_t mass_chars, size_t suffix_len);
int64_t nbd_csv_format_state(char* out, size_t cap, const char* prefix, size_t prefix_len, const char* mass_chars,
                             const int32_t* mass_off, const float* pos, const float* vel, const float* acc, int n,
                             const char* suffix, size_t suffix_len);

#ifdef __cplusplus
}
#endif
#endif /* NBD_H_ */

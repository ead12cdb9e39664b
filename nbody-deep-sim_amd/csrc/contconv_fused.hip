// contconv_fused.hip -- ContinuousConv.forward (contconv.py:80-98) as a block-sparse contraction for gfx950.
//
//   out[n][o] = scale_n * sum_{edges e -> n} window_e * sum_{8 corners c of e} t_c(e) * sum_i F[cell_c(e)][i][o] * feat[col_e][i]
//
// The trilinear blend is linear in the filter, so it is applied to the FEATURES (as nbd_contconv_bin_f32
// does): per (node n, filter cell k) touched by some edge of n,
//     A[n][k][i] = sum_{(e,c): cell_c(e) = k} window_e t_c(e) feat[col_e][i],      out[n] = sum_k A[n][k] . F[k].
// Only ~20 % (D = 6) / ~43 % (D = 4) of the (node, cell) blocks are touched at the published configuration
// (N = 16 384, mean radius-1 degree 32: 711 273 + 451 048 blocks, 3.1 / 4.9 edge corners per block), so the
// dense product `A (N x D^3 I) . F` of round 1 multiplied ~80 % zeros and moved a 1.3 GB A through HBM.
// Here A never leaves the chip and only touched blocks are multiplied:
//
//   nbd_contconv_pairs_batch_f32   per (tile of 128 nodes, filter resolution): every (edge, corner) pair
//                            source and weight (two arrays), grouped by (cell, node) -- a counting sort held in LDS;
//                            the packed "rows" (distinct nodes) of each (tile, cell) and their pair ranges. ~24 B of
//                            index data per pair, once per graph; all resolutions of a model in one launch (60 us at the
//                            published shape for D = 6 and D = 4 together).
//   nbd_contconv_fused_f32   per (tile, chunk of cells), 16 waves: eight producer waves gather the pairs' feature
//                            rows (pair records by scalar loads, rows by buffer loads with the row offset in an
//                            SGPR: no vector-ALU work per gathered row but the weighted add) and sum them into
//                            packed A rows in LDS (32 rows per step, a ring of four buffers, two waves per buffer); eight consumer waves multiply each step by 16 columns
//                            of the cell's I x O filter with fp32 MFMA (v_mfma_f32_16x16x4_f32; the fragment sits in
//                            registers, pre-shuffled by the host so that every lane loads it with one dwordx4 per 16
//                            k, the next cell's being fetched meanwhile) and add the 32 x 16 result into a
//                            128-node x 128-column accumulator in LDS. Producers and consumers meet through LDS
//                            flags only (no workgroup barrier inside the loop). Cell chunks of one tile are
//                            summed in fixed order by the finishing kernel (scale, activation). No float atomics:
//                            deterministic.
// Measured at the published shape (tools/bench_contconv.py, profiles/r02_contconv_*): layer D = 6 0.47 ms, D = 4
// 0.32 ms against 1.09 / 0.54 ms for binning + dense GEMM; executed 53.7 GFLOP per step (32-row granularity; the
// touched blocks alone are 38.1) against 120; the binned matrix: 0 bytes of HBM traffic against 2.7 GB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

constexpr int TN = NBD_CC_TILE;      // nodes per tile (128)
constexpr int SUB = 32;              // packed rows per MFMA step
constexpr int LDA = 132;             // A row stride in floats: 16-B aligned, conflict-free ds_read_b128 fragments
constexpr int MAXC = 160;            // filter cells kept (reachable) supported: D = 6 at R = 1 has exactly 160; the pair
                                     // kernel's LDS tables (7 bytes per (node, cell) + scan scratch) fill the 160 KiB at that
constexpr int CHUNK_MAX = 64;        // cells per workgroup of the fused kernel

struct Geo { int ix, iy, iz; float tx, ty, tz, window; };

// window, ball_to_cube and trilinear coordinates of one edge (contconv.py:30-33,84-90); same arithmetic as
// nn.hip's edge_geometry (the binning kernel the training path still uses)
__device__ __forceinline__ Geo edge_geo(const float* __restrict__ pos, int c, float xn, float yn, float zn, float r2max,
                                        float half) {
  Geo g;
  const float rx = pos[3 * c] - xn, ry = pos[3 * c + 1] - yn, rz = pos[3 * c + 2] - zn;   // pos[col] - pos[row]
  const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz));
  const float qq = 1.0f - d2 / r2max;
  g.window = (d2 < r2max) ? qq * qq * qq : 0.f;
  const float nrm = sqrtf(d2);
  const float sc = tanhf(nrm) / (nrm + 1e-8f);
  const float gx = (rx * sc + 1.0f) * half, gy = (ry * sc + 1.0f) * half, gz = (rz * sc + 1.0f) * half;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  g.ix = (int)fx; g.iy = (int)fy; g.iz = (int)fz;
  g.tx = gx - fx; g.ty = gy - fy; g.tz = gz - fz;
  return g;
}

// compact cell index of corner (ax, ay, az) of an edge, or -1 (outside the grid = grid_sample's zero padding,
// or a cell no sample can reach), and its weight window * t_corner
__device__ __forceinline__ int corner_cell(const Geo& g, int corner, int D, const int* __restrict__ cell_map, float* w) {
  const int ax = corner & 1, ay = (corner >> 1) & 1, az = corner >> 2;
  const int cx = g.ix + ax, cy = g.iy + ay, cz = g.iz + az;
  // branch-free (the map is read at a clamped index whatever the corner): eight of these run back to back per edge,
  // and an early return made every map read wait for the previous corner's
  const bool in = (unsigned)cx < (unsigned)D && (unsigned)cy < (unsigned)D && (unsigned)cz < (unsigned)D;
  const int cell = in ? (cz * D + cy) * D + cx : 0;                  // filters[z][y][x] (contconv.py:62-75)
  const float wxy = (ax ? g.tx : 1.0f - g.tx) * (ay ? g.ty : 1.0f - g.ty);
  *w = wxy * ((az ? g.tz : 1.0f - g.tz) * g.window);
  const int k = cell_map ? cell_map[cell] : cell;
  return in ? k : -1;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------- pair lists
// One workgroup (16 waves) per (tile of 128 nodes, filter resolution): the lists of every resolution a model
// uses are built by ONE launch (grid.y), which also fills the chip (a tile count of 128 is half the CUs).
// LDS: cnt[node][cell] (u16 pairs of a (node, cell) block, packed two per word), pwithin[node][cell] (u32: pairs
// of the same cell in lower nodes of the tile), rowidx[node][cell] (u8: touched lower nodes of the same cell),
// the tile's slice of rowptr and the cell map.
//   A  lane per edge over the tile's whole edge range (coalesced, every lane busy; the node of an edge by
//      binary search in the LDS copy of rowptr): geometry, 8 LDS counter increments. Order-free.
//   B  prefix sums down the tile per cell, two levels: thread (16-node segment, cell) sums its segment, then
//      walks it again from the sum of the lower segments (LDS reads only, conflict-free: neighbouring threads
//      take neighbouring cells). (First form: one shuffle-based wave scan per cell and quantity, 27 us.)
//   B2 prefix over cells -> desc[tile][cell] = {first row, rows}
//   B3 rows[] = {node_local, first pair}
//   C  half a wave per node, lane per edge: each pair takes the next slot of its (node, cell) block. Nodes are
//      handed out through a counter (dense tiles hold nodes with ~200 edges next to nodes with 5); a node
//      belongs to one half-wave and its edges are visited in CSR order, so the slots -- and with them the order
//      in which the fused kernel sums a block's pairs -- do not depend on which half-wave took the node.
// Global layout, per tile t with e_t = rowptr[128 t]: rows at 8 e_t + t (one sentinel row per tile),
// pairs at 8 e_t: an edge has at most 8 corners, so the bases need no scan across tiles.
struct PairJob {
  int D, n_cells;
  const int* cell_map;
  int2 *desc, *rows;
  int* pair_src;        // [8 * edge_capacity] source node of every (edge, corner) pair ...
  float* pair_w;        // ... and its window * trilinear weight (two arrays: the fused kernel reads them with scalar loads)
};
struct PairJobs { PairJob j[NBD_CC_MAX_RES]; };

#ifdef NBD_PAIRS_TRACE
__device__ long long* g_pairs_trace = nullptr;
#define PT(i) if (threadIdx.x == 0 && g_pairs_trace) g_pairs_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define PT(i)
#endif
constexpr int PAIR_THREADS = 1024;
constexpr int SEG = 16, NSEG = TN / SEG;                   // node segments of the two-level scan
__global__ __launch_bounds__(PAIR_THREADS) void contconv_pairs_kernel(
    const float* __restrict__ pos, const int* __restrict__ rowptr, const int* __restrict__ centres, int n, float r2max,
    const PairJobs jobs) {
  extern __shared__ unsigned smem[];
  const PairJob& job = jobs.j[blockIdx.y];
  const int D = job.D, n_cells = job.n_cells;
  int2* __restrict__ desc = job.desc;
  int2* __restrict__ rows = job.rows;
  int* __restrict__ psrc = job.pair_src;
  float* __restrict__ pwgt = job.pair_w;
  const int kc = (n_cells + 3) & ~3;                       // padded cell count (even: two u16 per word)
  unsigned* cnt32 = smem;                                   // [TN][kc/2]
  unsigned* pwithin = cnt32 + TN * kc / 2;                  // [TN][kc]
  unsigned char* rowidx = reinterpret_cast<unsigned char*>(pwithin + TN * kc);   // [TN][kc]
  __shared__ int cell_rows[MAXC], cell_pairs[MAXC], cell_rowbase[MAXC], cell_pairbase[MAXC];
  __shared__ int seg_pairs[NSEG][MAXC], seg_rows[NSEG][MAXC];
  __shared__ int rp[TN + 1];
  __shared__ int cmap[216];                                 // D <= 6
  __shared__ int next_node;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, n0 = tile * TN, n_here = min(TN, n - n0);
  const float half = (float)(D - 1) / 2.0f;

  PT(0)
  if (tid <= TN) rp[tid] = rowptr[min(n0 + tid, n)];
  if (tid < D * D * D) cmap[tid] = job.cell_map ? job.cell_map[tid] : tid;
  if (tid == 0) next_node = 0;
  for (int i = tid; i < TN * kc / 2; i += PAIR_THREADS) cnt32[i] = 0;
  __syncthreads();
  const int e_t = rp[0], e_end = rp[n_here];
  const size_t row_base = (size_t)8 * e_t + tile, pair_base = (size_t)8 * e_t;

  PT(1)
  // ---- A: counts (lane = edge)
  for (int e = e_t + tid; e < e_end; e += PAIR_THREADS) {
    int lo = 0, hi = n_here;                               // largest nl with rp[nl] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rp[mid] <= e) lo = mid; else hi = mid; }
    const int nl = lo, node = n0 + nl;
    const Geo g = edge_geo(pos, centres[e], pos[3 * node], pos[3 * node + 1], pos[3 * node + 2], r2max, half);
    if (g.window == 0.f) continue;                         // outside the radius: the reference multiplies by 0
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      float w;
      const int k = corner_cell(g, corner, D, cmap, &w);
      if (k >= 0) atomicAdd(&cnt32[(nl * kc + k) >> 1], 1u << (16 * (k & 1)));
    }
  }
  __syncthreads();

  PT(2)
  // ---- B: per cell, prefix over the nodes of the tile, in 16-node segments
  const unsigned short* cnt16 = reinterpret_cast<const unsigned short*>(cnt32);
  for (int w = tid; w < NSEG * n_cells; w += PAIR_THREADS) {
    const int sg = w / n_cells, k = w - sg * n_cells;       // neighbouring threads: neighbouring cells
    int ps = 0, rs = 0;
    for (int i = 0; i < SEG; ++i) { const int v = cnt16[(sg * SEG + i) * kc + k]; ps += v; rs += v > 0; }
    seg_pairs[sg][k] = ps; seg_rows[sg][k] = rs;
  }
  __syncthreads();
  for (int w = tid; w < NSEG * n_cells; w += PAIR_THREADS) {
    const int sg = w / n_cells, k = w - sg * n_cells;
    int pe = 0, re = 0;
    for (int q = 0; q < sg; ++q) { pe += seg_pairs[q][k]; re += seg_rows[q][k]; }
    for (int i = 0; i < SEG; ++i) {
      const int nl = sg * SEG + i, v = cnt16[nl * kc + k];
      pwithin[nl * kc + k] = pe; rowidx[nl * kc + k] = (unsigned char)re;
      pe += v; re += v > 0;
    }
    if (sg == NSEG - 1) { cell_pairs[k] = pe; cell_rows[k] = re; }
  }
  __syncthreads();

  PT(3)
  // ---- B2: prefix over cells (one wave; cells in chunks of 64 with a running carry)
  if (wave == 0) {
    int row_carry = 0, pair_carry = 0;
    for (int k0 = 0; k0 < n_cells; k0 += 64) {
      const int k = k0 + lane;
      const int rv = k < n_cells ? cell_rows[k] : 0, pv = k < n_cells ? cell_pairs[k] : 0;
      const int ri = wave_incl_scan(rv, lane), pi = wave_incl_scan(pv, lane);
      if (k < n_cells) {
        cell_rowbase[k] = row_carry + ri - rv;
        cell_pairbase[k] = pair_carry + pi - pv;
        desc[(size_t)tile * n_cells + k] = make_int2(row_carry + ri - rv, rv);
      }
      row_carry += __shfl(ri, 63);
      pair_carry += __shfl(pi, 63);
    }
    if (lane == 0) rows[row_base + row_carry] = make_int2(0, pair_carry);      // sentinel: end of the last row
  }
  __syncthreads();

  PT(4)
  // ---- B3: row records (wave = nodes, lanes = cells: no division)
  for (int nl = wave; nl < n_here; nl += PAIR_THREADS / 64)
    for (int k = lane; k < n_cells; k += 64)
      if (cnt16[nl * kc + k] > 0)
        rows[row_base + cell_rowbase[k] + rowidx[nl * kc + k]] = make_int2(nl, cell_pairbase[k] + (int)pwithin[nl * kc + k]);
  __syncthreads();          // B3 reads the counters that C counts down

  PT(5)
  // ---- C: place the pairs (counters count down: slot = old - 1). A half-wave takes the next node from the
  // counter; its 32 lanes walk the node's edges in order.
  const int hl = lane & 31;
  for (;;) {
    int nl = 0;
    if (hl == 0) nl = atomicAdd(&next_node, 1);
    nl = __shfl(nl, lane & 32);                            // broadcast inside the half-wave
    if (nl >= n_here) break;
    const int node = n0 + nl;
    const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
    const int e0 = rp[nl], e1 = rp[nl + 1];
    // the source of the NEXT trip is fetched before this trip's counters and stores (a dense tile's nodes have
    // ~200 edges: seven trips per node, each otherwise paying centres -> pos -> store in sequence)
    int c = (e0 + hl < e1) ? centres[e0 + hl] : 0;
    float px = pos[3 * c], py = pos[3 * c + 1], pz = pos[3 * c + 2];
    for (int e = e0 + hl; e < e1; e += 32) {
      const int c_cur = c;
      const float sx = px, sy = py, sz = pz;
      if (e + 32 < e1) { c = centres[e + 32]; px = pos[3 * c]; py = pos[3 * c + 1]; pz = pos[3 * c + 2]; }
      const float src[3] = {sx, sy, sz};
      const Geo g = edge_geo(src, 0, xn, yn, zn, r2max, half);
      if (g.window == 0.f) continue;
      // the eight counters are decremented by eight UNCONDITIONAL returning atomics issued together (a corner that
      // falls outside subtracts 0 from cell 0's word), then the eight stores: with `if (k < 0) continue` in front of
      // each, every corner paid its own LDS round trip -- this phase was 43 of the densest tile's 74 us
      int kk[8];
      float ww[8];
      unsigned old[8];
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) kk[corner] = corner_cell(g, corner, D, cmap, &ww[corner]);
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const int k = max(kk[corner], 0);
        old[corner] = atomicSub(&cnt32[(nl * kc + k) >> 1], kk[corner] >= 0 ? 1u << (16 * (k & 1)) : 0u);
      }
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const int k = kk[corner];
        if (k < 0) continue;
        const int slot = (int)((old[corner] >> (16 * (k & 1))) & 0xffffu) - 1;
        const size_t at = (size_t)pair_base + cell_pairbase[k] + pwithin[nl * kc + k] + slot;
        psrc[at] = c_cur; pwgt[at] = ww[corner];
      }
    }
  }
#ifdef NBD_PAIRS_TRACE
  __syncthreads();
  PT(6)
#endif
}

// ---------------------------------------------------------------------------------------------- fused conv
// grid = (tiles, cell chunks, column groups of 128); block = 1024 threads: waves 0-7 consume (MFMA, 16 output
// columns each), waves 8-15 produce (gather + sum of the packed A rows; two waves per ring buffer).
//
// A "step" is 32 packed rows of one cell. Producer wave p builds the steps p, p+4, p+8, ... into ring buffer p
// (it owns that buffer); consumer waves walk all steps in order. The two sides meet only through LDS flags
// -- full[b] = sequence number of the step buffer b holds, done[b] = consumer waves that have finished with
// it -- so a producer's load latency (row records -> pairs -> feature rows, three dependent trips to L2) is
// hidden behind three other steps, and the consumers never wait for each other: every wave owns its 32
// output columns of the LDS accumulator. (First version: one s_barrier per step, 7 us per step against
// 1.8 us of MFMA.)
#if defined(NBD_CC_ABL) && NBD_CC_ABL == 3
#define CC_MFMA(acc, a, b) acc[0] += a * b;
#else
#define CC_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#endif
#ifndef NBD_CC_SLEEP
#define NBD_CC_SLEEP 4             // s_sleep between two polls of a flag (x64 cycles); 0 / 1 / 4 measured 0.479 / 0.492 / 0.474 ms (D = 6)
#endif
#ifndef NBD_CC_XCD_MAP
#define NBD_CC_XCD_MAP 0
#endif
#ifndef NBD_CC_PRODUCER_PRIO
#define NBD_CC_PRODUCER_PRIO 2
#endif
constexpr int NBUF = 4;
constexpr int CC_CONSUMERS = 8;                  // consumer waves (16 output columns each)
constexpr int CC_PRODUCERS = 2 * NBUF;           // producer waves: two per ring buffer, 16 packed rows each
constexpr int CC_THREADS = (CC_CONSUMERS + CC_PRODUCERS) * 64;
constexpr int HSUB = SUB / 2;
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int MAX_STEPS = CHUNK_MAX * (TN / SUB);

// Flags live in LDS and guard LDS data only: relaxed workgroup-scope atomics + fences restricted to the local
// address space, so that signalling never drains the global loads a wave keeps in flight (filter-fragment and
// row-record prefetches).
#define CC_WAIT(flag, cond)                                                                                  \
  do {                                                                                                       \
    while (!(__hip_atomic_load(&(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) cond))               \
      __builtin_amdgcn_s_sleep(NBD_CC_SLEEP);                                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                          \
  } while (0)
#define CC_RELEASE_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")

#ifdef NBD_CC_TRACE
__device__ long long* g_cc_trace = nullptr;
#endif

// Build-time probes (tools/build_contconv_trace.sh, tools/contconv_trace.py): -DNBD_CC_TRACE stamps every workgroup
// (s_memrealtime) and accumulates the time waves spend on the LDS flags; -DNBD_CC_ABL = 1 / 2 / 3 are timing-only
// ablations (one cell's filters / feature rows from a 64 KiB table / no MFMA). None of it is in the product build.
#ifndef NBD_CC_TRACE
#define DBG_T(x)
#define DBG_ACC(x)
#define DBG_LAT(x, j)
#define DBG_W(x)
#define DBG_PH(i, x)
#endif
template <int KG>
__global__ __launch_bounds__(CC_THREADS) void contconv_fused_kernel(
    const float* __restrict__ feat, int ldf, int I, const int* __restrict__ rowptr, int n,
    const int2* __restrict__ desc, const int2* __restrict__ rows, const int* __restrict__ pair_src,
    const float* __restrict__ pair_w,
    const f4* __restrict__ filt, int n_cells, int kq_count, int colblocks, int cells_per_chunk, int n_tiles,
    int n_chunks, int O, float* __restrict__ partial) {
  // f4-typed so that the dynamic region starts 16-byte aligned behind the static __shared__ variables: declared
  // as float[] it began at an 8-byte offset and EVERY ds_read_b128 / ds_write_b64 below took the unaligned path
  // (SQ_LDS_UNALIGNED_STALL = 85 % of all LDS cycles, LDS array 69 % busy, MFMA pipe 27 %)
  extern __shared__ f4 lds_aligned[];
  float* lds = reinterpret_cast<float*>(lds_aligned);
  float* out_acc = lds;                                    // [TN + 1][128]: row TN swallows a step's padding rows
  float* a_buf = out_acc + (TN + 1) * 128;                 // [NBUF][SUB][LDA]
  int* rowmap = reinterpret_cast<int*>(a_buf + NBUF * SUB * LDA);   // [NBUF][SUB]
  __shared__ int s_cell[CHUNK_MAX], s_rowbeg[CHUNK_MAX], s_nrows[CHUNK_MAX];
  __shared__ unsigned char st_cell[MAX_STEPS], st_sub[MAX_STEPS];  // step -> (compact cell, 32-row slice)
  __shared__ int s_ncell, s_nsteps;
  __shared__ int full[NBUF], done[NBUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NBD_CC_TRACE
  __shared__ long long s_dbg_wait[16];
  long long dbg_wait = 0;
#define DBG_T(x) const long long x = __builtin_amdgcn_s_memrealtime();
#define DBG_ACC(x) dbg_wait += __builtin_amdgcn_s_memrealtime() - x;
#define DBG_W(x) const long long x##_w = dbg_wait;
  long long dbg_ph[3] = {0, 0, 0}; const long long l2 = 0; (void)l2;
#define DBG_PH(i, x) dbg_ph[i] += __builtin_amdgcn_s_memrealtime() - x;
  long long dbg_lat = 0; int dbg_nlat = 0;
#define DBG_LAT(x, j) if ((j) == 0) { dbg_lat += __builtin_amdgcn_s_memrealtime() - x - (dbg_wait - x##_w); ++dbg_nlat; }
  __shared__ int s_dbg_pairs;
  const long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) s_dbg_pairs = 0;
#endif
  // Workgroup -> (tile, cell chunk), tile fastest. Measured and rejected (NBD_CC_XCD_MAP=1): chunk = index mod 8, which
  // pins every chunk -- and its 1.3 MB slice of the filter matrix -- to ONE XCD (workgroups go to the XCDs round-robin
  // by index: confirmed with the probes, 1024 of 1024). The filters then sit in that XCD's L2, but nothing got
  // faster per workgroup (the kernel is not waiting on those loads) while the chunks of the central cells, now all on
  // two XCDs, stretched the launch from 0.50 to 0.65 ms.
  int tile, chunk;
  if (NBD_CC_XCD_MAP && (n_chunks & 7) == 0) {
    const int k = blockIdx.x >> 3;
    tile = k % n_tiles; chunk = (blockIdx.x & 7) + 8 * (k / n_tiles);
  } else {
    tile = blockIdx.x % n_tiles; chunk = blockIdx.x / n_tiles;
  }
  const int n0 = tile * TN;
  const int k_begin = chunk * cells_per_chunk, k_end = min(n_cells, k_begin + cells_per_chunk);
  const int e_t = rowptr[n0];
  const int2* t_rows = rows + (size_t)8 * e_t + tile;
  const int* t_src = pair_src + (size_t)8 * e_t;
  const float* t_w = pair_w + (size_t)8 * e_t;

  // non-empty cells of this chunk, compacted, and the step table (wave 0; cells_per_chunk <= 64)
  if (wave == 0) {
    const int k = k_begin + lane;
    int2 d = make_int2(0, 0);
    if (k < k_end) d = desc[(size_t)tile * n_cells + k];
    const unsigned long long m = __ballot(d.y > 0);
    const int nsub = (d.y + SUB - 1) / SUB;
    int incl = nsub;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (d.y > 0) {
      const int j = __popcll(m & ((1ull << lane) - 1ull));
      s_cell[j] = k; s_rowbeg[j] = d.x; s_nrows[j] = d.y;
      for (int u = 0; u < nsub; ++u) { st_cell[incl - nsub + u] = (unsigned char)j; st_sub[incl - nsub + u] = (unsigned char)u; }
    }
    if (lane == 63) s_nsteps = incl;
    if (lane == 0) s_ncell = __popcll(m);
    if (lane < NBUF) { full[lane] = 0; done[lane] = 0; }
  }
  for (int i = tid; i < (TN + 1) * 128 / 4; i += CC_THREADS) reinterpret_cast<f4*>(out_acc)[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int nsteps = s_nsteps;

  if (wave >= CC_CONSUMERS) {
    const int w4 = (wave - CC_CONSUMERS) & (NBUF - 1);    // ring buffer this wave fills
    const int hf = (wave - CC_CONSUMERS) / NBUF;           // which 16 rows of each step
    // ---------------- producer: rows [16 hf, 16 hf + 16) of the steps w4, w4 + 4, ... into buffer w4
    // The wave is instruction-issue bound (two waves per SIMD), so the per-pair work is kept to a handful of
    // instructions: the pair records of 64 pairs sit one per lane; a row's byte offset is one VALU multiply for
    // all 64; a feature row is then `v_readlane -> s_add -> global_load (scalar base + lane offset)`, and its
    // accumulation `v_readlane -> v_pk_fma_f32`. Rows are fetched 16 at a time with the next 16 in flight
    // (the gather wants tens of KiB outstanding per CU: MI355X_MICROARCH.md "Indexed rows"), and the wave's next
    // step's row records are fetched while this one is summed. (First form: 64-bit index math, a predicate
    // around every load and a clamp per pair -- 25+ instructions per pair, 5.7 us per step.)
    constexpr int PB = 8, NBF = 4;                        // rows per batch, batches in flight per wave
    // The producers are the YOUNGER waves of their SIMDs (waves 8-15 behind the consumers' 0-7): at equal priority
    // the issue arbiter serves age first, and the consumers' back-to-back MFMAs left a producer one instruction
    // slot in ~70 (in-kernel stamps: 2.1 us to ISSUE a batch of 16 row loads, 1.4 us to sum it -- the same with the
    // rows coming from a 64 KiB table as from the 8 MB one, i.e. not a memory effect). Their stream is sparse (about
    // ten instructions per gathered row), so raised priority costs the consumers little.
    __builtin_amdgcn_s_setprio(NBD_CC_PRODUCER_PRIO);
    float* a_dst = a_buf + (w4 * SUB + hf * HSUB) * LDA;
    const bool live = 2 * lane < I;
    const unsigned lane8 = (unsigned)min(2 * lane, I - 2) * 4u;      // clamped: every lane reads inside the row
    const unsigned ldb = (unsigned)ldf * 4u;
    // this half's rows of step s (clamped to the last step: ONE load whatever s, so that hipcc can count its
    // s_waitcnt vmcnt); lane `cnt` holds the row behind the last one (or the tile's sentinel): its first pair ends the step
    auto step_cnt = [&](int s) { return max(0, min(HSUB, s_nrows[st_cell[s]] - st_sub[s] * SUB - hf * HSUB)); };
    auto row_records = [&](int s_want) {
      const int s = min(s_want, nsteps - 1), j = st_cell[s], cnt = step_cnt(s);
      const int2 v = t_rows[s_rowbeg[j] + st_sub[s] * SUB + (cnt > 0 ? hf * HSUB + min(lane, cnt) : 0)];
      return (lane <= cnt && cnt > 0) ? v : make_int2(-1, 0);
    };
    // The feature matrix through a buffer descriptor: a row is fetched by `buffer_load_dwordx2 v, v_lane, s[rsrc], s_row offen`
    // with the row's byte offset in an SGPR -- and the pair records {source, weight} are read with SCALAR loads
    // (uniform addresses), so gathering a row costs the vector ALU nothing and summing it one v_pk_fma_f32 with the
    // weight as a scalar operand. The fp32 MFMAs of the consumer waves keep the SIMDs' vector issue busy: in-kernel
    // stamps showed a producer getting one VALU slot per ~35 cycles (1.5-2 us to ISSUE 16 row loads when each cost a
    // v_readlane + a 64-bit VALU add; the same from a 64 KiB table as from the 8 MB one, so not a memory effect).
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, (int)min((size_t)0x7fffffff, ((size_t)(n - 1) * ldf + I) * 4), 0x00020000);
    int2 rinfo = make_int2(-1, 0), rinfo_n = make_int2(-1, 0);
    if (nsteps > 0) { rinfo = row_records(w4); rinfo_n = row_records(w4 + NBUF); }
    for (int s = w4, use = 0; s < nsteps; s += NBUF, ++use) {
      const int cnt = step_cnt(s);
      const int p_begin = __builtin_amdgcn_readlane(rinfo.y, 0), p_end = __builtin_amdgcn_readlane(rinfo.y, cnt);
#ifdef NBD_CC_TRACE
      if (lane == 0) atomicAdd(&s_dbg_pairs, p_end - p_begin);
#endif
      const int2 rinfo_cur = rinfo;
      rinfo = rinfo_n;
      // the ring buffer is claimed only when the first feature rows are already on their way
      bool claimed = false;
      auto claim = [&]() {
        if (!claimed) {
          DBG_T(p0) if (use > 0) CC_WAIT(done[w4], >= CC_CONSUMERS * use);   // the consumers are done with this buffer
          DBG_ACC(p0)
          if (lane < HSUB) rowmap[w4 * SUB + hf * HSUB + lane] = lane < cnt ? rinfo_cur.x : TN;   // padding rows -> the dummy row
          claimed = true;
        }
      };
      int cur = 0;
      int next_begin = __builtin_amdgcn_readlane(rinfo_cur.y, 1);
      f2 acc = {0.f, 0.f};
      auto flush = [&]() {                                           // row `cur` is complete
        *reinterpret_cast<f2*>(a_dst + cur * LDA + 2 * lane) = live ? acc : f2{0.f, 0.f};
        acc = f2{0.f, 0.f};
        ++cur;
        next_begin = __builtin_amdgcn_readlane(rinfo_cur.y, cur + 1);
      };
      // The step-half's pairs as one stream of 16-row batches, two in flight. EVERY stage issues exactly 16 row loads
      // (past the end: re-reads of the last pair's row, L1 hits), and the row records of the step after next ONE:
      // only then can hipcc count its s_waitcnt vmcnt() and let batch j be summed while batch j + 1 is in flight
      // (with `if (more) issue(...)` the merged paths made it assume no younger loads: summing row u of one batch
      // waited for row u of the NEXT batch -- vmcnt(15..0) behind a conditional block of 16 loads in the ISA).
      const int np = p_end - p_begin, nb = (np + PB - 1) / PB;
      if (nb > 0) {
        // NBF batches of PB rows in flight. A batch's pair records sit in SGPRs, fetched by scalar loads one stage
        // before they are needed: its PB sources (for the row loads) NBF - 1 stages before its PB weights (for the
        // sum), two register sets each, alternating. Inline asm: hipcc selects scalar loads only for memory it can
        // prove unclobbered, and the LDS fences of the flag protocol defeat that proof (it fell back to vector
        // loads + a v_readfirstlane waterfall per row).
        f2 fbuf[NBF][PB];
        i8v srcA, srcB, wgtA, wgtB;
        auto req_src = [&](i8v& src, int j) {                        // batches past the end: no request, row 0 is gathered
          if (j < nb) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(src) : "s"(t_src + p_begin + PB * j) : "memory");
        };
        auto req_wgt = [&](i8v& wgt, int j) {
          if (j < nb) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(wgt) : "s"(t_w + p_begin + PB * j) : "memory");
        };
        auto landed = [&](i8v& a, i8v& b, i8v& c, i8v& d) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
        };
        auto issue = [&](f2* f, const i8v& src, int j) {
          const int valid = np - PB * j;                             // records past the step's last pair: gather row 0 of
#pragma unroll                                                       // the batch again instead of whatever follows
          for (int u = 0; u < PB; ++u) {
            const int sidx = u < valid ? src[u] : (valid > 0 ? src[0] : 0);     // nothing left at all: row 0, a hot line
#if defined(NBD_CC_ABL) && NBD_CC_ABL == 2
            const unsigned ro = (unsigned)(sidx & 127) * ldb;
#else
            const unsigned ro = (unsigned)sidx * ldb;
#endif
            f[u] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(frs, (int)lane8, (int)ro, 0));
          }
        };
        auto sum = [&](const f2* f, const i8v& wgt, int j) {
          const int valid = min(PB, np - PB * j), q0 = p_begin + PB * j;
          if (valid == PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
              if (q0 + u == next_begin) flush();                     // wave-uniform
              const float w = __int_as_float(wgt[u]);
              acc = __builtin_elementwise_fma(f2{w, w}, f[u], acc);
            }
          } else {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
              if (u < valid) {                                       // wave-uniform
                if (q0 + u == next_begin) flush();
                const float w = __int_as_float(wgt[u]);
                acc = __builtin_elementwise_fma(f2{w, w}, f[u], acc);
              }
            }
          }
        };
        static_assert(NBF == 4, "the stage rotation below is written out for four batches in flight");
        DBG_T(l0) DBG_W(l0)
        req_src(srcA, 0); req_src(srcB, 1);
        landed(srcA, srcB, wgtA, wgtB);
        issue(fbuf[0], srcA, 0); issue(fbuf[1], srcB, 1);
        req_src(srcA, 2); req_src(srcB, 3); req_wgt(wgtA, 0);
        rinfo_n = row_records(s + 2 * NBUF);
        landed(srcA, srcB, wgtA, wgtB);
        issue(fbuf[2], srcA, 2);
        DBG_PH(0, l0)
        claim();
        DBG_T(l1)
        // stage K: rows of batch j + K + 3 on their way, next records requested, batch j + K summed meanwhile
#define CC_STAGE(K, SRC_CUR, SRC_OTHER, WGT_CUR, WGT_OTHER)                                                   \
        landed(srcA, srcB, wgtA, wgtB);                                                                      \
        issue(fbuf[(K + NBF - 1) % NBF], SRC_CUR, j + K + NBF - 1);                                          \
        req_src(SRC_OTHER, j + K + NBF); req_wgt(WGT_OTHER, j + K + 1);                                      \
        sum(fbuf[K], WGT_CUR, j + K);                                                                        \
        if (j + K + 1 >= nb) break;
        for (int j = 0;; j += NBF) {
          CC_STAGE(0, srcB, srcA, wgtA, wgtB)
          if (j == 0) { DBG_PH(1, l1) DBG_LAT(l0, j) }
          CC_STAGE(1, srcA, srcB, wgtB, wgtA)
          CC_STAGE(2, srcB, srcA, wgtA, wgtB)
          CC_STAGE(3, srcA, srcB, wgtB, wgtA)
        }
#undef CC_STAGE
      } else {
        rinfo_n = row_records(s + 2 * NBUF);
      }
      claim();
      if (cnt > 0) *reinterpret_cast<f2*>(a_dst + cur * LDA + 2 * lane) = live ? acc : f2{0.f, 0.f};
      CC_RELEASE_FENCE();
      if (lane == 0) __hip_atomic_fetch_add(&full[w4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // both halves -> 2 (use + 1)
    }
  } else {
    // ---------------- consumer: 16 output columns, all steps in order
    // Two consumer waves share a SIMD (waves w and w + 4): while one scatters its results or waits on a flag,
    // the other's MFMAs keep the matrix pipe busy. Each wave multiplies the step's 32 packed rows (two 16-row
    // tiles = two independent accumulator chains) by its 16 columns of the cell's filter with
    // v_mfma_f32_16x16x4_f32; the fragment (I/16 dwordx4 per lane) sits in registers, the next cell's is fetched
    // into a second set while this one multiplies (the two sets alternate: no copies).
    // (With ONE consumer wave per SIMD on 32 columns, 32x32x2 MFMA, a step took 6500 cycles against 4096 of
    // matrix work: the scatter and the flag handling of that one wave were all lost matrix time.)
    const int cw = wave;                                   // 0..7
    const int cb = blockIdx.z * CC_CONSUMERS + cw;         // 16-column block of the output
    const bool has_cols = cb < colblocks;
    f4 bf0[KG], bf1[KG];
    // always exactly KG loads (clamped index, zeroed afterwards): a counted s_waitcnt vmcnt(KG) is only possible
    // when the number of younger loads does not depend on the path taken
    auto load_b = [&](f4* dstv, int cell) {
#if defined(NBD_CC_ABL) && NBD_CC_ABL == 1
      cell = 0;
#endif
      const f4* src = filt + (((size_t)cell * colblocks + min(cb, colblocks - 1)) * kq_count) * 64 + lane;
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 v = src[(size_t)min(g, kq_count - 1) * 64];
        dstv[g] = (g < kq_count) ? v : f4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if (nsteps > 0) load_b(bf0, s_cell[0]);
    float* o_col = out_acc + cw * 16 + (lane & 15);
#define CC_STEP(BC)                                                                                          \
    {                                                                                                        \
      f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};                                          \
      const float* a_base = a_buf + (b * SUB + (lane & 15)) * LDA + (lane >> 4) * 4;                         \
      _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                                       \
        const f4 a0 = *reinterpret_cast<const f4*>(a_base + g * 16);                                         \
        const f4 a1 = *reinterpret_cast<const f4*>(a_base + 16 * LDA + g * 16);                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
          CC_MFMA(acc0, a0[j], BC[g][j]) CC_MFMA(acc1, a1[j], BC[g][j])                                      \
        }                                                                                                    \
      }                                                                                                      \
      /* C row = 4 (lane >> 4) + reg (+ 16 for the second tile), column = lane & 15 -> node of the tile */  \
      const int4 n0v = *reinterpret_cast<const int4*>(rowmap + b * SUB + 4 * (lane >> 4));                   \
      const int4 n1v = *reinterpret_cast<const int4*>(rowmap + b * SUB + 16 + 4 * (lane >> 4));              \
      /* the buffer can go back to its producer: A and the row map are in registers */                      \
      CC_RELEASE_FENCE();                                                                                    \
      if (lane == 0) __hip_atomic_fetch_add(&done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    \
      const int nd[8] = {n0v.x, n0v.y, n0v.z, n0v.w, n1v.x, n1v.y, n1v.z, n1v.w};                            \
      float old[8];                                                                                          \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) old[r] = o_col[nd[r] * 128];                             \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) o_col[nd[r] * 128] = old[r] + acc0[r];                   \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) o_col[nd[4 + r] * 128] = old[4 + r] + acc1[r];           \
    }
    // Cells two at a time: the even cell multiplies with fragment set 0 while set 1 is fetched for the odd cell
    // and vice versa. Written out like this (instead of a parity switch inside one loop) the loads of the NEXT
    // cell are the only ones younger than the current cell's, so the wait before the first MFMA is a counted
    // vmcnt(KG) and the prefetch really stays in flight (with the switch the compiler had to use vmcnt(0):
    // every cell change paid the full L2 / Infinity Cache latency).
    auto cell_steps = [&](int j) { return (s_nrows[j] + SUB - 1) / SUB; };
#define CC_ONE_STEP(BC)                                                                                      \
    {                                                                                                        \
      const int b = s & (NBUF - 1), use = s / NBUF;                                                          \
      DBG_T(c0) CC_WAIT(full[b], == 2 * (use + 1)); DBG_ACC(c0)                                              \
      if (has_cols) CC_STEP(BC) else if (lane == 0)                                                          \
        __hip_atomic_fetch_add(&done[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                 \
      ++s;                                                                                                   \
    }
    /* the cell's first step is peeled: its fragment wait then sits in straight-line code behind the prefetch */ \

#define CC_CELL(BC)                                                                                          \
    {                                                                                                        \
      const int nsub = cell_steps(j);                                                                        \
      CC_ONE_STEP(BC)                                                                                        \
      for (int u = 1; u < nsub; ++u) CC_ONE_STEP(BC)                                                         \
    }
    {
      int s = 0;
      const int ncell = s_ncell;
      for (int j = 0; j < ncell;) {
        load_b(bf1, s_cell[min(j + 1, ncell - 1)]);        // the last cell re-fetches itself: never used
        CC_CELL(bf0)
        if (++j >= ncell) break;
        load_b(bf0, s_cell[min(j + 1, ncell - 1)]);
        CC_CELL(bf1)
        ++j;
      }
    }
#undef CC_CELL
#undef CC_ONE_STEP
#undef CC_STEP
  }
#ifdef NBD_CC_TRACE
  if (lane == 0) s_dbg_wait[wave] = wave == 9 ? dbg_lat : (wave == 13 ? dbg_nlat : (wave == 10 ? dbg_ph[0] : (wave == 11 ? dbg_ph[1] : (wave == 14 ? dbg_ph[2] : (wave == 15 ? dbg_nlat : dbg_wait)))));
#endif
  __syncthreads();
#ifdef NBD_CC_TRACE
  const long long dbg_t1 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- write the tile's partial sums for this cell chunk: partial[chunk][node][column]
  float* dst = partial + ((size_t)chunk * n + n0) * O;
  const int col0 = blockIdx.z * 128;
  const int n_here = min(TN, n - n0), cols = min(128, O - col0);
  for (int i = tid; i < n_here * 128; i += CC_THREADS) {
    const int nl = i >> 7, c = i & 127;
    if (c < cols) dst[(size_t)nl * O + col0 + c] = out_acc[nl * 128 + c];
  }
#ifdef NBD_CC_TRACE
  __syncthreads();
  if (tid == 0 && g_cc_trace) {
    long long* t = g_cc_trace + (size_t)blockIdx.x * 16;
    t[6] = s_dbg_wait[0]; t[7] = s_dbg_wait[4]; t[8] = s_dbg_wait[8]; t[9] = s_dbg_wait[12];
    t[10] = s_dbg_wait[9]; t[11] = s_dbg_wait[13]; t[12] = s_dbg_wait[10]; t[13] = s_dbg_wait[11]; t[14] = s_dbg_wait[14]; t[15] = s_dbg_wait[15];
    t[0] = dbg_t0; t[1] = dbg_t1; t[2] = __builtin_amdgcn_s_memrealtime(); t[3] = nsteps; t[4] = s_dbg_pairs;
    t[5] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
}

// out = act(scale * sum of the chunk partials), fixed chunk order
__global__ __launch_bounds__(256) void contconv_finish_kernel(const float* __restrict__ partial, int n_chunks,
                                                              const float* __restrict__ rowscale, int act,
                                                              float* __restrict__ out, int ldo, int n, int O) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)n * O;
  if (i >= total) return;
  const int row = (int)(i / O), col = (int)(i - (size_t)row * O);
  float v = partial[i];
  for (int s = 1; s < n_chunks; ++s) v += partial[(size_t)s * total + i];
  if (rowscale) v = __fmul_rn(v, rowscale[row]);
  out[(size_t)row * ldo + col] = act == 1 ? tanhf(v) : v;
}

struct FusedPlan { int tiles, chunks, cells_per_chunk, colgroups; };
FusedPlan plan_fused(int n, int n_cells, int O) {
  FusedPlan p;
  p.tiles = ceil_div(n, TN);
  p.colgroups = ceil_div(O, 128);
  // ~2 workgroups per CU over the launch (a workgroup carries ~10 us of fixed cost: fill, drain, 64 KiB of partial
  // sums; measured at N = 16 384, D = 6 / 4: 3 / 4 / 5 / 6 / 8 / 12 / 16 chunks 0.512 / 0.465 / 0.495 / 0.480 /
  // 0.485 / 0.497 / 0.510 and 0.372 / 0.315 / 0.319 / 0.319 / 0.317 / 0.329 / 0.347 ms); <= 64 cells per chunk
  int chunks = ceil_div(512, p.tiles * p.colgroups);
  if (chunks > 16) chunks = 16;                      // bounds the partial-sum traffic of small problems
  if (chunks > n_cells) chunks = n_cells;
  if (chunks < ceil_div(n_cells, CHUNK_MAX)) chunks = ceil_div(n_cells, CHUNK_MAX);
  if (chunks < 1) chunks = 1;
  p.cells_per_chunk = ceil_div(n_cells, chunks);
  p.chunks = ceil_div(n_cells, p.cells_per_chunk);
  return p;
}

}  // namespace

extern "C" {

#ifdef NBD_CC_TRACE
int nbd_debug_cc_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cc_trace), &buf, sizeof(buf)); }
#endif

#ifdef NBD_PAIRS_TRACE
int nbd_debug_pairs_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pairs_trace), &buf, sizeof(buf)); }
#endif

int nbd_contconv_fused_supported(int in_channels, int out_channels, int n_cells) {
  return in_channels > 0 && in_channels % 4 == 0 && in_channels <= 128 && out_channels > 0 && n_cells > 0 &&
         n_cells <= MAXC;
}

size_t nbd_contconv_pairs_bytes(int n, int64_t edge_capacity, int n_cells) {
  if (n <= 0 || edge_capacity < 0 || n_cells <= 0) return 0;
  const size_t tiles = (size_t)ceil_div(n, TN);
  const size_t desc = tiles * n_cells * sizeof(int2);
  const size_t rows = ((size_t)8 * edge_capacity + tiles) * sizeof(int2);
  const size_t pairs = 2 * (((size_t)8 * edge_capacity * sizeof(int) + 255) & ~(size_t)255);   // sources, weights
  return ((desc + 255) & ~(size_t)255) + ((rows + 255) & ~(size_t)255) + pairs + 256;
}

static void split_pairs_buffer(void* buf, int n, int64_t edge_capacity, int n_cells, int2** desc, int2** rows, int** pair_src,
                               float** pair_w) {
  const size_t tiles = (size_t)ceil_div(n, TN);
  char* p = static_cast<char*>(buf);
  *desc = reinterpret_cast<int2*>(p);
  p += (tiles * n_cells * sizeof(int2) + 255) & ~(size_t)255;
  *rows = reinterpret_cast<int2*>(p);
  p += (((size_t)8 * edge_capacity + tiles) * sizeof(int2) + 255) & ~(size_t)255;
  *pair_src = reinterpret_cast<int*>(p);
  p += ((size_t)8 * edge_capacity * sizeof(int) + 255) & ~(size_t)255;
  *pair_w = reinterpret_cast<float*>(p);
}

int nbd_contconv_pairs_batch_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                                 float radius_sq, int n_res, const int* filter_resolutions, const int* const* cell_maps,
                                 const int* n_cells, void* const* pair_lists, const size_t* pair_lists_bytes,
                                 nbd_stream_t stream) {
  if (n < 0 || edge_capacity < 0 || n_res < 1 || n_res > NBD_CC_MAX_RES) return NBD_E_BADARG;
  if (!filter_resolutions || !cell_maps || !n_cells || !pair_lists || !pair_lists_bytes) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !rowptr || !centres) return NBD_E_BADARG;
  PairJobs jobs;
  int kc_max = 0;
  for (int r = 0; r < n_res; ++r) {
    const int d = filter_resolutions[r], nc = n_cells[r];
    if (d < 2 || d > 6 || nc <= 0 || nc > MAXC || nc > d * d * d) return NBD_E_BADARG;
    if (!cell_maps[r] && nc != d * d * d) return NBD_E_BADARG;
    if (!pair_lists[r] || (reinterpret_cast<uintptr_t>(pair_lists[r]) & 15) != 0) return NBD_E_BADARG;
    if (pair_lists_bytes[r] < nbd_contconv_pairs_bytes(n, edge_capacity, nc)) return NBD_E_WORKSPACE;
    PairJob& j = jobs.j[r];
    j.D = d; j.n_cells = nc; j.cell_map = cell_maps[r];
    split_pairs_buffer(pair_lists[r], n, edge_capacity, nc, &j.desc, &j.rows, &j.pair_src, &j.pair_w);
    const int kc = (nc + 3) & ~3;
    if (kc > kc_max) kc_max = kc;
  }
  for (int r = n_res; r < NBD_CC_MAX_RES; ++r) jobs.j[r] = jobs.j[0];
  const size_t lds = (size_t)TN * kc_max / 2 * 4 + (size_t)TN * kc_max * 4 + (size_t)TN * kc_max;
  {   // > 64 KiB of dynamic LDS needs the opt-in (a per-function attribute, idempotent)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_pairs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  contconv_pairs_kernel<<<dim3(ceil_div(n, TN), n_res), PAIR_THREADS, lds, (hipStream_t)stream>>>(
      pos, rowptr, centres, n, radius_sq, jobs);
  return status();
}

int nbd_contconv_pairs_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                           int filter_resolution, float radius_sq, const int* cell_map, int n_cells,
                           void* pair_lists, size_t pair_lists_bytes, nbd_stream_t stream) {
  const int* maps[1] = {cell_map};
  void* lists[1] = {pair_lists};
  return nbd_contconv_pairs_batch_f32(pos, rowptr, centres, n, edge_capacity, radius_sq, 1, &filter_resolution, maps,
                                      &n_cells, lists, &pair_lists_bytes, stream);
}

size_t nbd_contconv_fused_workspace_bytes(int n, int n_cells, int out_channels) {
  if (n <= 0 || n_cells <= 0 || out_channels <= 0) return 0;
  const FusedPlan p = plan_fused(n, n_cells, out_channels);
  return (size_t)p.chunks * n * out_channels * sizeof(float);
}

size_t nbd_contconv_filter_floats(int in_channels, int out_channels, int n_cells) {
  if (in_channels <= 0 || out_channels <= 0 || n_cells <= 0) return 0;
  return (size_t)n_cells * ceil_div(out_channels, 16) * ceil_div(in_channels, 16) * 64 * 4;
}

int nbd_contconv_fused_f32(const float* feat, int ldf, int in_channels, const int* rowptr, int n, int64_t edge_capacity,
                           const void* pair_lists, const float* filters_shuffled, int n_cells, int out_channels,
                           const float* rowscale, int act, float* out, int ldo, void* workspace,
                           size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || !nbd_contconv_fused_supported(in_channels, out_channels, n_cells) || ldf < in_channels ||
      ldo < out_channels || (ldf & 1))
    return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!feat || !rowptr || !pair_lists || !filters_shuffled || !out) return NBD_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(feat) & 7) || (reinterpret_cast<uintptr_t>(filters_shuffled) & 15)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_contconv_fused_workspace_bytes(n, n_cells, out_channels)) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int2 *desc, *rows;
  int* pair_src;
  float* pair_w;
  split_pairs_buffer(const_cast<void*>(pair_lists), n, edge_capacity, n_cells, &desc, &rows, &pair_src, &pair_w);
  const FusedPlan p = plan_fused(n, n_cells, out_channels);
  const size_t lds = (size_t)((TN + 1) * 128 + NBUF * SUB * LDA) * sizeof(float) + NBUF * SUB * sizeof(int);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(p.tiles * p.chunks, 1, p.colgroups);
  const int kq_count = ceil_div(in_channels, 16);
#define CC_LAUNCH(K)                                                                                                \
  do {                                                                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_fused_kernel<K>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                     \
    if (e != hipSuccess) return (int)e;                                                                             \
    contconv_fused_kernel<K><<<grid, CC_THREADS, lds, st>>>(feat, ldf, in_channels, rowptr, n, desc, rows, pair_src, pair_w, \
                                                           reinterpret_cast<const f4*>(filters_shuffled), n_cells, \
                                                           kq_count, ceil_div(out_channels, 16), p.cells_per_chunk, \
                                                           p.tiles, p.chunks, out_channels, partial);              \
  } while (0)
  if (kq_count <= 2) CC_LAUNCH(2); else CC_LAUNCH(8);      // K depth: I <= 32 / I <= 128
#undef CC_LAUNCH
  int rc = status();
  if (rc) return rc;
  const size_t total = (size_t)n * out_channels;
  contconv_finish_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(partial, p.chunks, rowscale, act, out, ldo, n,
                                                                         out_channels);
  return status();
}

}  // extern "C"

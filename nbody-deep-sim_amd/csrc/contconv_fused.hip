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
//                            It also cuts every (tile, cell)'s rows into STEPS of 16 (one MFMA tile of rows) and writes
//                            the tile's step list + step count: the unit of work of the fused kernel.
//   nbd_contconv_fused_f32   PERSISTENT workgroups (one per CU, 16 waves) over the global step sequence (tile-major, cell,
//                            16-row slice), cut into gridDim.x equal contiguous ranges by STEP count ("stream-K": the
//                            matrix work of every workgroup is equal whatever the tiles' densities). Eight producer
//                            waves, one per buffer of an eight-deep LDS ring: a producer gathers the feature rows of
//                            its step's pairs (pair records staged once per 64 pairs in wave-private LDS and read back
//                            as broadcasts: no scalar loads, no dependent round trip per batch of rows; rows by buffer
//                            loads, four batches of 8 in flight) and sums them into the step's 16 packed A rows. Eight
//                            consumer waves multiply each step by 16 columns of the cell's I x O filter with fp32 MFMA
//                            (v_mfma_f32_16x16x4_f32, operands swapped so that a lane ends up with 4 consecutive output
//                            columns of ONE node: the scatter into the 128-node x 128-column LDS accumulator is one
//                            128-bit read-modify-write per lane; the filter fragment sits in registers, pre-shuffled by
//                            the host, the next cell's fetched meanwhile). Producers and consumers meet through LDS flags
//                            only (no workgroup barrier inside a range). When a range crosses into the next tile the
//                            consumers write their columns of the accumulator to partial slot (workgroup + tile) -- a
//                            merge-path numbering, unique and independent of timing -- and the finishing kernel sums a
//                            tile's slots in workgroup order (scale, activation). No float atomics: deterministic.
// Round 2's form (one workgroup per (tile, chunk of cells), 32-row steps, ring of four, pair records by scalar
// loads one stage ahead) measured 0.475 / 0.32 ms per layer (D = 6 / 4) at the published shape with the MFMA pipe
// 47 % busy: its producers paid one exposed scalar-load round trip per 8 gathered rows and ~9 us of dependent
// latencies per half-step, and every workgroup 11 us of fill / drain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

constexpr int TN = NBD_CC_TILE;      // nodes per tile (128)
constexpr int CC_GRID = 256;         // persistent workgroups of the fused kernel (one per CU of an MI355X)
constexpr int MAXC = 160;            // filter cells kept (reachable) supported: D = 6 at R = 1 has exactly 160; the pair
                                     // kernel's LDS tables (7 bytes per (node, cell) + scan scratch) fill the 160 KiB at that
constexpr int SUBR = 16;             // packed rows per step of the fused kernel (one 16 x 16 MFMA tile of rows)
// Cost of a step for the fused kernel's split of the step sequence over its workgroups, in pairs: a step takes the
// consumers ~1.1-1.45 us whatever it holds and the producers ~14.5 ns per pair (in-kernel stamps at the published shape,
// round 3: the gathers are served from the Infinity Cache at ~3.5 TB/s), so below ~80-100 pairs the matrix side sets
// the pace and above it the gather. (The additive model cost = a + pairs, a = 32 .. 96, measured 2-12 % slower.)
constexpr int CC_COST_MIN = 64;    // round 4 (bf16 consumers), same box, D = 6 + D = 4 layers: 24 / 40 / 56 / 64 / 72 / 80 / 112 ->
                                   // 0.741 / 0.662 / 0.611 / 0.609 / 0.613 / 0.620 / 0.658 ms (round 3's fp32 consumers: 80)
__host__ __device__ inline int step_cost(int pairs) { return pairs > CC_COST_MIN ? pairs : CC_COST_MIN; }

// first step record of a tile: a (tile, cell) with r rows has ceil(r / 16) <= r / 16 + 1 steps and a tile's rows
// are <= 8 x its edges, so e_t / 2 + tile * (cells + 2) needs no scan across tiles (one terminal record per tile)
__host__ __device__ inline size_t step_base(int tile, int e_t, int n_cells) {
  return (size_t)(e_t >> 1) + (size_t)tile * (size_t)(n_cells + 2);
}

struct Geo { int ix, iy, iz; float tx, ty, tz, window; };

// window, ball_to_cube and trilinear coordinates of one edge (contconv.py:30-33,84-90); same arithmetic as
// nn.hip's edge_geometry (the binning kernel the training path still uses)
// sign = +1: rows are aggregation targets, the listed node c is the feature source (the forward lists);
// sign = -1: rows are feature sources and c the aggregation target (the adjoint lists of the backward pass): the
// same edge, the same relative position pos[source] - pos[target] -- negation is exact, so both groupings see
// bit-identical cells and weights.
__device__ __forceinline__ Geo edge_geo(const float* __restrict__ pos, int c, float xn, float yn, float zn, float r2max,
                                        float half, float sign = 1.0f) {
  Geo g;
  const float rx = sign * (pos[3 * c] - xn), ry = sign * (pos[3 * c + 1] - yn), rz = sign * (pos[3 * c + 2] - zn);   // pos[col] - pos[row]
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
  const int* rowptr;    // [n + 1] first edge of every row ...
  const int* centres;   // ... and the node each edge lists
  const int* deg;       // NULL: a row ends where the next begins (CSR); else padded rows (ELL), deg[i] entries valid
  float sign;           // +1 forward lists, -1 adjoint lists (see edge_geo)
  const int* cell_map;
  int2 *desc, *rows;
  int2* pair;           // [8 * edge_capacity] {source node, bits of window * trilinear weight} of every (edge, corner) pair: ONE
                        // 8-byte record (two arrays at first: the placement phase is bound by the issue of its scattered stores)
  int4* steps;          // per tile (at step_base): {first row, cell | rows << 8 | steps left in the cell << 16, first pair, 0}
  int* tile_nsteps;     // [tiles]
  float* inv_deg;       // [n] 1 / max(in-degree, 1): the row scale of a mean aggregation, a by-product (saves a launch)
  int2* cellstep;       // [tiles][n_cells + 1] {first step of the cell in the tile's list, running cost in front of it}: what the plan
                        // kernel cuts the cells into groups with ([n_cells]: {steps, cost} of the whole tile)
  int* tile_cost;       // [tiles] sum of the tile's step costs (each step: max(CC_COST_MIN, its pairs)); the running
                        // (inclusive) sum inside the tile is the step records' .w
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
    const float* __restrict__ pos, int n, float r2max, const PairJobs jobs) {
  extern __shared__ unsigned smem[];
  const PairJob& job = jobs.j[blockIdx.y];
  const int D = job.D, n_cells = job.n_cells;
  const int* __restrict__ rowptr = job.rowptr;
  const int* __restrict__ centres = job.centres;
  const float sign = job.sign;
  int2* __restrict__ desc = job.desc;
  int2* __restrict__ rows = job.rows;
  int2* __restrict__ prec = job.pair;
  const int kc = (n_cells + 3) & ~3;                       // padded cell count (even: two u16 per word)
  unsigned* cnt32 = smem;                                   // [TN][kc/2]
  unsigned* pwithin = cnt32 + TN * kc / 2;                  // [TN][kc]
  unsigned char* rowidx = reinterpret_cast<unsigned char*>(pwithin + TN * kc);   // [TN][kc]
  __shared__ int cell_rows[MAXC], cell_pairs[MAXC], cell_rowbase[MAXC], cell_pairbase[MAXC], cell_stepbase[MAXC];
  __shared__ int seg_pairs[NSEG][MAXC], seg_rows[NSEG][MAXC];
  __shared__ int rp[TN + 1];
  __shared__ int rdeg[TN];                                  // edges of every row (CSR: rp[i + 1] - rp[i]; ELL: deg[i])
  __shared__ int cmap[216];                                 // D <= 6
  __shared__ int next_node, s_tile_steps, s_over;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, n0 = tile * TN, n_here = min(TN, n - n0);
  const float half = (float)(D - 1) / 2.0f;

  PT(0)
  if (tid <= TN) rp[tid] = rowptr[min(n0 + tid, n)];
  if (tid < D * D * D) cmap[tid] = job.cell_map ? job.cell_map[tid] : tid;
  if (tid == 0) { next_node = 0; s_over = 0; }
  for (int i = tid; i < TN * kc / 2; i += PAIR_THREADS) cnt32[i] = 0;
  __syncthreads();
  // The (node, cell) pair counters are 16 bits wide (two per LDS word: at D = 6 the tables fill the LDS as it is). A
  // row can only overflow one with more than 65 535 edges (a hub of a > 65 536-body clump under the search's "first 32
  // by index" rule): such a tile is marked instead of counted wrongly -- tile_nsteps = -1, which the fused kernel skips
  // and the finishing kernel turns into NaN outputs for the tile's nodes. Loud, not silent.
  if (tid < n_here) {
    const int dg = job.deg ? job.deg[n0 + tid] : rp[tid + 1] - rp[tid];
    rdeg[tid] = dg;
    if (dg > 65535) s_over = 1;
    job.inv_deg[n0 + tid] = 1.0f / (float)max(dg, 1);      // = nbd_degree_scale_f32 mode 0
  }
  __syncthreads();
  const int e_t = rp[0], e_end = s_over ? rp[0] : rp[n_here - 1] + rdeg[n_here - 1];
  const size_t row_base = (size_t)8 * e_t + tile, pair_base = (size_t)8 * e_t;
  int4* __restrict__ t_steps = job.steps + step_base(tile, e_t, n_cells);

  PT(1)
  // ---- A: counts (lane = edge)
  for (int e = e_t + tid; e < e_end; e += PAIR_THREADS) {
    int lo = 0, hi = n_here;                               // largest nl with rp[nl] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rp[mid] <= e) lo = mid; else hi = mid; }
    const int nl = lo, node = n0 + nl;
    if (e - rp[nl] >= rdeg[nl]) continue;                  // padding of an ELL row
    const Geo g = edge_geo(pos, centres[e], pos[3 * node], pos[3 * node + 1], pos[3 * node + 2], r2max, half, sign);
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
    int row_carry = 0, pair_carry = 0, step_carry = 0;
    for (int k0 = 0; k0 < n_cells; k0 += 64) {
      const int k = k0 + lane;
      const int rv = k < n_cells ? cell_rows[k] : 0, pv = k < n_cells ? cell_pairs[k] : 0;
      const int sv = (rv + SUBR - 1) / SUBR;                 // steps of this cell
      const int ri = wave_incl_scan(rv, lane), pi = wave_incl_scan(pv, lane), si = wave_incl_scan(sv, lane);
      if (k < n_cells) {
        cell_rowbase[k] = row_carry + ri - rv;
        cell_pairbase[k] = pair_carry + pi - pv;
        cell_stepbase[k] = step_carry + si - sv;
        desc[(size_t)tile * n_cells + k] = make_int2(row_carry + ri - rv, rv);
        for (int u = 0; u < sv; ++u) {                       // .z (first pair) comes from B3
          int4* s = t_steps + (step_carry + si - sv + u);
          s->x = row_carry + ri - rv + SUBR * u;
          s->y = k | (min(SUBR, rv - SUBR * u) << 8) | ((sv - u) << 16);      // .w (running cost) comes from B4
        }
      }
      row_carry += __shfl(ri, 63);
      pair_carry += __shfl(pi, 63);
      step_carry += __shfl(si, 63);
    }
    if (lane == 0) {
      rows[row_base + row_carry] = make_int2(0, pair_carry);      // sentinel: end of the last row
      t_steps[step_carry] = make_int4(0, 0, pair_carry, 0);      // terminal record: end of the last step
      job.tile_nsteps[tile] = s_over ? -1 : step_carry;
      s_tile_steps = step_carry;
    }
  }
  __syncthreads();

  PT(4)
  // ---- B3: row records (wave = nodes, lanes = cells: no division)
  for (int nl = wave; nl < n_here; nl += PAIR_THREADS / 64)
    for (int k = lane; k < n_cells; k += 64)
      if (cnt16[nl * kc + k] > 0) {
        const int ri = rowidx[nl * kc + k], first = cell_pairbase[k] + (int)pwithin[nl * kc + k];
        rows[row_base + cell_rowbase[k] + ri] = make_int2(nl, first);
        if ((ri & (SUBR - 1)) == 0) t_steps[cell_stepbase[k] + ri / SUBR].z = first;     // a step's first pair
      }
  __syncthreads();          // B3 reads the counters that C counts down

  // ---- B4: running cost of the tile's steps (a step's pairs = next step's first pair - its own)
  {
    const int ns = s_tile_steps;
    int carry = 0;
    for (int j0 = 0; j0 < ns; j0 += PAIR_THREADS) {
      const int j = j0 + tid;
      const int c = j < ns ? step_cost(t_steps[j + 1].z - t_steps[j].z) : 0;
      const int incl = wave_incl_scan(c, lane);
      if (lane == 63) seg_pairs[0][wave] = incl;              // B's scratch, free by now
      __syncthreads();
      int woff = 0, tot = 0;
      for (int i = 0; i < PAIR_THREADS / 64; ++i) { const int x = seg_pairs[0][i]; woff += i < wave ? x : 0; tot += x; }
      if (j < ns) t_steps[j].w = carry + woff + incl;
      carry += tot;
      __syncthreads();
    }
    if (tid == 0) job.tile_cost[tile] = carry;
  }
  // ---- B5: where every cell's steps begin in the tile's list and the running cost in front of them
  {
    int2* cs = job.cellstep + (size_t)tile * (n_cells + 1);
    for (int k = tid; k <= n_cells; k += PAIR_THREADS) {
      const int sb = k < n_cells ? cell_stepbase[k] : s_tile_steps;      // (an empty cell: where the next one begins)
      cs[k] = s_over ? make_int2(0, 0) : make_int2(sb, sb > 0 ? t_steps[sb - 1].w : 0);
    }
  }

  PT(5)
  // ---- C: place the pairs (counters count down: slot = old - 1). A half-wave takes the next node from the
  // counter; its 32 lanes walk the node's edges in order.
  const int hl = lane & 31;
  for (;;) {
    int nl = 0;
    if (hl == 0) nl = atomicAdd(&next_node, 1);
    nl = __shfl(nl, lane & 32);                            // broadcast inside the half-wave
    if (nl >= n_here || s_over) break;
    const int node = n0 + nl;
    const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
    const int e0 = rp[nl], e1 = e0 + rdeg[nl];
    // the source of the NEXT trip is fetched before this trip's counters and stores (a dense tile's nodes have
    // ~200 edges: seven trips per node, each otherwise paying centres -> pos -> store in sequence)
    int c = (e0 + hl < e1) ? centres[e0 + hl] : 0;
    float px = pos[3 * c], py = pos[3 * c + 1], pz = pos[3 * c + 2];
    for (int e = e0 + hl; e < e1; e += 32) {
      const int c_cur = c;
      const float sx = px, sy = py, sz = pz;
      if (e + 32 < e1) { c = centres[e + 32]; px = pos[3 * c]; py = pos[3 * c + 1]; pz = pos[3 * c + 2]; }
      const float src[3] = {sx, sy, sz};
      const Geo g = edge_geo(src, 0, xn, yn, zn, r2max, half, sign);
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
        prec[at] = make_int2(c_cur, __float_as_int(ww[corner]));
      }
    }
  }
#ifdef NBD_PAIRS_TRACE
  __syncthreads();
  PT(6)
#endif
}

// ---------------------------------------------------------------------------------------------- plan
// Once per pair list (one workgroup per job, right behind the pair kernel). The filter cells are cut into `groups`
// contiguous GROUPS of equal cost (over all tiles), one per XCD of an MI355X: persistent workgroup w of the fused kernel
// works on group w mod groups -- workgroup w runs on XCD w mod 8 -- so the 32 workgroups that share an XCD's 4 MiB L2
// re-read the fragments of the same ~20 cells (1.9 MB at D = 6) all launch long instead of sweeping all 160 (15.7 MB,
// past the L2: in-kernel stamps of round 4 put a cell change at 1.2 us, an Infinity-Cache round trip that the one-step
// reach of the in-place reload cannot cover). A group's step sequence = tile-major, the tile's steps of the group's
// cells (contiguous in the tile's list); it is cut into CC_GRID / groups ranges of equal cost:
//   gcell[g]                  first cell of group g ([groups] = n_cells)
//   tile_base[g][t]           first step of tile t in group g's sequence ([tiles]: its length), cost_base likewise
//   cuts[g][j]                first step (in group g's sequence) of the group's j-th workgroup
// cut(j) = first step whose cost-before is >= C_g j / W. Every cut is one thread: two binary searches (tile by cost
// prefix, step by the running cost inside the tile).
struct PlanJob { const int* tile_nsteps; const int2* cellstep; const int4* steps; const int* rowptr; int n_cells, groups;
                 int* cuts; int* tile_base; int* cost_base; int* gcell; };
struct PlanJobs { PlanJob j[NBD_CC_MAX_RES]; };
__global__ __launch_bounds__(1024) void contconv_plan_kernel(const PlanJobs jobs, int n_tiles) {
  const PlanJob& J = jobs.j[blockIdx.x];
  const int K = J.n_cells, G = J.groups, W = CC_GRID / G;
  __shared__ int ccost[MAXC];                               // cost of cell k over all tiles (the whole list's cost fits an int)
  __shared__ int s_gc[NBD_CC_GROUPS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int2* cs = J.cellstep;
  if (G == 1) {                                              // one group: all cells
    if (tid == 0) { s_gc[0] = 0; s_gc[1] = K; J.gcell[0] = 0; J.gcell[1] = K; }
  } else {
    for (int k = tid; k < K; k += 1024) ccost[k] = 0;
    __syncthreads();
    // thread = (cell, slice of the tiles): neighbouring threads read neighbouring cells of a tile (coalesced), every thread's
    // loads are independent of each other (eight in flight) -- one (tile, cell) entry per trip behind an LDS atomic took 30 us
    {
      const int slices = 1024 / K > 0 ? 1024 / K : 1;          // K <= 160: >= 6 slices
      const int k = tid % K, sl = tid / K;
      if (sl < slices) {
        const int per = (n_tiles + slices - 1) / slices, t_lo = sl * per, t_hi = min(n_tiles, t_lo + per);
        int c = 0;
#pragma unroll 8
        for (int t = t_lo; t < t_hi; ++t) {
          const int a = cs[(size_t)t * (K + 1) + k].y, b = cs[(size_t)t * (K + 1) + k + 1].y;
          c += b - a;                                          // (a refused tile's entries are zero)
        }
        if (c) atomicAdd(&ccost[k], c);
      }
    }
    __syncthreads();
    if (wave == 0) {                                         // prefix over the cells (K <= 160 = three lane chunks), then the
      static_assert(MAXC <= 192, "three chunks of 64 cells");       // group boundaries by ballot: a serial loop of dependent LDS
      int before[3];                                         // reads and 64-bit products was 15 us of this kernel
      int carry = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = 64 * c + lane;
        const int v = k < K ? ccost[k] : 0;
        const int incl = wave_incl_scan(v, lane);
        before[c] = carry + incl - v;
        carry += __shfl(incl, 63);
      }
      const long long tot = carry;
      if (lane == 0) { s_gc[0] = 0; s_gc[G] = K; }
      for (int g = 1; g < G; ++g) {                          // gcell[g] = first cell whose cost-before reaches tot g / G
        int first = K;
#pragma unroll
        for (int c = 2; c >= 0; --c) {
          const int k = 64 * c + lane;
          const unsigned long long m = __ballot(k <= K && (long long)before[c] * G >= tot * g);
          if (m) first = 64 * c + __builtin_ctzll(m);
        }
        if (lane == 0) s_gc[g] = min(first, K);
      }
    }
    __syncthreads();
    if (tid <= G) J.gcell[tid] = s_gc[tid];
  }
  __syncthreads();
  // per group: prefix over the tiles of (steps, cost) of the group's cells -- wave g scans group g, 64 tiles per trip
  if (wave < G) {
    const int g = wave, c0 = s_gc[g], c1 = s_gc[g + 1];
    int* tb = J.tile_base + (size_t)g * (n_tiles + 1);
    int* cb = J.cost_base + (size_t)g * (n_tiles + 1);
    int scarry = 0, ccarry = 0;
    for (int t0 = 0; t0 < n_tiles; t0 += 64) {
      const int t = t0 + lane;
      int vs = 0, vc = 0;
      if (t < n_tiles && J.tile_nsteps[t] > 0) {             // -1: a tile the pair kernel refused
        const int2 a = cs[(size_t)t * (K + 1) + c0], b = cs[(size_t)t * (K + 1) + c1];
        vs = b.x - a.x; vc = b.y - a.y;
      }
      const int is = wave_incl_scan(vs, lane), ic = wave_incl_scan(vc, lane);
      if (t < n_tiles) { tb[t] = scarry + is - vs; cb[t] = ccarry + ic - vc; }
      scarry += __shfl(is, 63); ccarry += __shfl(ic, 63);
    }
    if (lane == 0) { tb[n_tiles] = scarry; cb[n_tiles] = ccarry; }
  }
  __threadfence_block();
  __syncthreads();
  // every cut of every group by its own thread
  const int wshift = 31 - __clz(W);
  for (int i = tid; i < G * (W + 1); i += 1024) {
    const int g = i / (W + 1), w = i - g * (W + 1);
    const int c0 = s_gc[g];
    const int* tb = J.tile_base + (size_t)g * (n_tiles + 1);
    const int* cb = J.cost_base + (size_t)g * (n_tiles + 1);
    const int T = tb[n_tiles];
    const long long Ctot = cb[n_tiles];
    int cut;
    if (w == 0 || Ctot == 0) cut = w == W ? T : 0;
    else if (w >= W) cut = T;
    else {
      const int B = (int)((Ctot * w) >> wshift);             // W is a power of two
      int lo = 0, hi = n_tiles - 1;                          // smallest t with cost_base[t + 1] > B
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (cb[mid + 1] > B) hi = mid; else lo = mid + 1; }
      const int t = lo, rel = B - cb[t], sb = tb[t], ns = tb[t + 1] - sb;
      if (rel == 0) cut = sb;
      else {
        const int2 first = cs[(size_t)t * (K + 1) + c0];     // the group's first step in tile t and the cost in front of it
        const int4* ts_ = J.steps + step_base(t, J.rowptr[t * TN], K) + first.x;
        int a = 0, b = ns;                                   // steps i with running cost (inside the group) < rel (.w increases)
        while (a < b) { const int mid = (a + b) >> 1; if (ts_[mid].w - first.y < rel) a = mid + 1; else b = mid; }
        cut = sb + min(1 + a, ns);
      }
    }
    J.cuts[i] = cut;
  }
}

// ---------------------------------------------------------------------------------------------- fused conv
// grid = (persistent workgroups, 1, column groups of 128); block = 1024 threads: waves 0-7 consume (MFMA, 16 output
// columns each), waves 8-15 produce (gather + sum of the packed A rows).
//
// A "step" is 16 packed rows (distinct nodes) of one (tile, cell). The global step sequence (tile-major, then
// cell, then 16-row slice; the pair kernel's per-tile step lists laid end to end) is cut into gridDim.x contiguous
// ranges of equal cost; a workgroup copies its range's step records into LDS (<= CC_CAP at a time) and walks it.
// Producer p builds the steps p, p + 8, p + 16, ... of the range, step q into ring buffer q mod NBUF; consumer waves walk
// all steps in order. The two sides meet only through LDS flags -- full[b] = number of steps buffer b has held, done[b][w]
// = steps of buffer b consumer wave w has finished with -- so a producer has eight step-times for the latency of its step
// (row records and pair records prefetched one own-step ahead; the feature rows of <= 32 pairs in flight), and the
// consumers never wait for each other: every wave owns its 16 output columns of the LDS accumulator.
//
// ARITHMETIC (round 4): fp32-equivalent on the bf16 matrix pipe. The fp32 matrix instruction (v_mfma_f32_16x16x4_f32)
// runs at the fp32 vector rate, 1/16 of the bf16 rate, and was the kernel's pace-setter (0.93 us of matrix-pipe time per
// step and SIMD). Both operands are now split into three bf16 terms,
//     x = x_hi + x_mid + x_lo,   x_hi = bf16(x), x_mid = bf16(x - x_hi), x_lo = bf16(x - x_hi - x_mid)
// (round to nearest even; both differences are exact in fp32, and the last rounding is too but for a measure-zero set:
// the three terms carry all 24 bits of x), and the product is the six terms of order <= 2^-16,
//     a.b ~= a_hi b_lo + a_mid b_mid + a_hi b_mid + a_mid b_hi + a_lo b_hi + a_hi b_hi,
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (24 instructions of 16 cycles per step and wave instead of 32 of 32
// cycles). What is dropped (a_mid b_lo, a_lo b_mid, a_lo b_lo) is <= 2^-24 |a| |b| per product -- the size of the rounding
// the fp32 instruction commits on every product anyway; tests/test_surrogate_gpu.py holds the row error against an fp64
// product to no more than the fp32-MFMA kernel's on the same inputs (tools/cc_bf16x3.hip: 9e-8 vs 1.2e-7 on N(0,1) data).
// The filters are split once per weight update (contconv_shuffle_kernel); the A rows by the producers as they write them.
constexpr int CC_SLEEP = 2;                      // s_sleep between two polls of a flag (x64 cycles)
constexpr int CC_PRODUCER_PRIO = 3;
// The two consumer waves of a SIMD (w and w + 4) run at DIFFERENT priorities: at equal priority their MFMAs
// interleave issue by issue, both bursts end together and both waves then sit in their LDS round trips (flag poll,
// fragment reads, scatter) at the same time with the matrix pipe idle -- a convoy (in-kernel stamps, round 3). With one
// wave preferred its burst runs uncontended while the other is in its latency phase, and vice versa.
constexpr int CC_PRIO_HI = 2, CC_PRIO_LO = 1;
constexpr int NBUF = 6;                          // ring depth: what the LDS holds beside the accumulator (12 KiB per step)
constexpr int NPROD = 8;                         // producer waves
constexpr int CC_CONSUMERS = 8;                  // consumer waves (16 output columns each)
constexpr int CC_THREADS = (CC_CONSUMERS + NPROD) * 64;
constexpr int LDO = 132;                         // accumulator row stride (floats): spreads the rows of a 128-bit scatter over the banks
constexpr int CC_CAP = 256;                      // step records held in LDS at a time (a multiple of NPROD)
constexpr int PB = 8;                            // gathered rows per batch
constexpr int NBF = 4;                           // batches of PB gathered rows in flight per producer wave
// A step in LDS: three planes (hi, mid, lo) of 16 rows x 128 bf16 = 256 B per row, no padding: the 16-byte chunk c of row
// m sits at chunk c ^ m (XOR swizzle). A consumer lane reads chunk (lane >> 4) + 4 s of row lane & 15: whatever lanes the
// LDS groups together (MI355X_MICROARCH.md, ds_read_b128: {0-3, 12-15, 20-27}, ...), equal lane >> 4 means distinct rows
// -> distinct chunks, and the two lane >> 4 values of a group land on complementary halves: conflict-free. A producer lane
// writes the dword (2 bf16) of columns 2 lane, 2 lane + 1: 64 distinct banks.
constexpr int A_ROW_BYTES = 256, A_PLANE_BYTES = SUBR * A_ROW_BYTES, A_STEP_BYTES = 3 * A_PLANE_BYTES;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef int q4 __attribute__((ext_vector_type(4)));       // a quad of 8 bf16 as the registers see it

// x (two floats) -> the three bf16 terms, packed two to a dword (low half = x[0])
__device__ __forceinline__ unsigned pk_bf16(f2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2)); }
__device__ __forceinline__ f2 unpk_bf16(unsigned h) { return f2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)}; }
__device__ __forceinline__ void split_bf16x3(f2 v, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pk_bf16(v);
  const f2 r1 = v - unpk_bf16(hi);               // exact
  mid = pk_bf16(r1);
  const f2 r2 = r1 - unpk_bf16(mid);             // exact
  lo = pk_bf16(r2);
}

// Flags live in LDS and guard LDS data only: relaxed workgroup-scope atomics + fences restricted to the local
// address space, so that signalling never drains the global loads a wave keeps in flight (filter-fragment and
// record prefetches).
#define CC_WAIT(flag, cond)                                                                                  \
  do {                                                                                                       \
    while (!(__hip_atomic_load(&(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) cond))               \
      __builtin_amdgcn_s_sleep(CC_SLEEP);                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                          \
  } while (0)
#define CC_RELEASE_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
// done[b][w] = uses of buffer b consumer wave w has finished with (plain stores by one lane: no read-modify-write, and
// none of the ballot / count code hipcc wraps around an atomic add); the producer reads the eight words lane-parallel
#define CC_WAIT_ALL_DONE(ptr, want)                                                                          \
  do {                                                                                                       \
    while (__builtin_amdgcn_ballot_w64(__hip_atomic_load((ptr) + (lane & (CC_CONSUMERS - 1)), __ATOMIC_RELAXED, \
                                                         __HIP_MEMORY_SCOPE_WORKGROUP) < (want)) != 0ull)      \
      __builtin_amdgcn_s_sleep(CC_SLEEP);                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                          \
  } while (0)

// Build-time probe (tools/build_contconv_trace.sh, tools/contconv_trace.py): -DNBD_CC_TRACE stamps every workgroup
// (s_memrealtime) and accumulates the time consumer wave 0 / producer wave 0 spend on the LDS flags. Not in the product build.
#ifdef NBD_CC_TRACE
__device__ long long* g_cc_trace = nullptr;
#define DBG_T(x) const long long x = __builtin_amdgcn_s_memrealtime();
#define DBG_ACC(cond, v, x) if (cond) v += __builtin_amdgcn_s_memrealtime() - x;
// per-step timeline of ONE workgroup (g_cc_tl_wg): [wave][step of the range][4] stamps -- consumers: step entered, buffer full,
// step done; producers: step begun, buffer asked for, buffer got, step published (tools/contconv_timeline.py)
__device__ long long* g_cc_tl = nullptr;
__device__ int g_cc_tl_wg = -1;
constexpr int CC_TL_STEPS = 512;
#define DBG_TL(WAVE, STEP, K) if (g_cc_tl && (int)blockIdx.x == g_cc_tl_wg && (tid & 63) == 0 && (STEP) < CC_TL_STEPS) \
    g_cc_tl[(((size_t)(WAVE) * CC_TL_STEPS + (STEP)) << 2) + (K)] = __builtin_amdgcn_s_memrealtime();
#else
#define DBG_T(x)
#define DBG_ACC(cond, v, x)
#define DBG_TL(WAVE, STEP, K)
#endif

// LDS / memory loads are "divergent" to the compiler even at wave-uniform addresses: values that steer control flow
// are moved to SGPRs explicitly, or every `if` on them becomes v_cmp + s_and_saveexec and the counters VGPRs
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
__device__ __forceinline__ int4 uni4(int4 v) { return make_int4(UNI(v.x), UNI(v.y), UNI(v.z), UNI(v.w)); }
__device__ __forceinline__ int2 uni2(int2 v) { return make_int2(UNI(v.x), UNI(v.y)); }

struct CCArgs {
  const float* feat; int ldf, I; const int* rowptr; int n, n_tiles;
  const int2* rows; const int2* pair; const int4* steps; const int* tile_nsteps; const int* tile_cost;
  // the plan, computed once per pair list (contconv_plan_kernel): cell groups, and per group the first step of every
  // workgroup's range ([groups][CC_GRID / groups + 1]) and of every tile ([groups][tiles + 1]) in the group's sequence
  const int* cuts; const int* tile_base; const int* gcell; const int2* cellstep; int groups;
  const uint4* filt; int n_cells, colblocks, OP; float* partial;
};

// LDS carve-up of the fused kernel (dynamic region, 16-byte aligned)
struct CCLds {
  float* out_acc;      // [TN][LDO]
  char* a_buf;         // [NBUF][3 planes][SUBR][256 B]
  int* rowmap;         // [NBUF][SUBR]
  int2* scratch;       // [NPROD producers][2][64] {row byte offset, weight}: two groups of 64 pair records per wave
  int4* st4;           // [CC_CAP] {cell | rows << 8 | steps left in the cell << 16, tile, first row, e_t}
  int2* st2;           // [CC_CAP] {first pair, end pair}
  int* seg;            // [3][CC_CAP], borrows the ring's first buffer while the ring is idle
  int *s_red, *s_nseg, *full, *done;
};
__device__ __forceinline__ CCLds cc_lds(float* lds, int* statics) {
  CCLds L;
  L.out_acc = lds;
  L.a_buf = reinterpret_cast<char*>(L.out_acc + TN * LDO);
  L.rowmap = reinterpret_cast<int*>(L.a_buf + NBUF * A_STEP_BYTES);
  L.scratch = reinterpret_cast<int2*>(L.rowmap + NBUF * SUBR);
  L.st4 = reinterpret_cast<int4*>(L.scratch + NPROD * 128);
  L.st2 = reinterpret_cast<int2*>(L.st4 + CC_CAP);
  L.seg = reinterpret_cast<int*>(L.a_buf);
  L.s_red = statics; L.s_nseg = statics + 16; L.full = statics + 24; L.done = statics + 32;   // full[NBUF], done[NBUF][CC_CONSUMERS]
  return L;
}
constexpr size_t CC_LDS_BYTES = (size_t)TN * LDO * sizeof(float) + (size_t)NBUF * A_STEP_BYTES + NBUF * SUBR * sizeof(int) +
                                NPROD * 128 * sizeof(int2) + CC_CAP * (sizeof(int4) + sizeof(int2));
static_assert(CC_LDS_BYTES + 1024 <= 160 * 1024, "the fused kernel's LDS carve-up");
static_assert(3 * CC_CAP * sizeof(int) <= (size_t)A_STEP_BYTES, "the segment list borrows one ring buffer");
static_assert(NBUF <= 8, "full[] has eight words");

// The step records of [p0, p1) of the global sequence -> LDS (all 1024 threads; both roles call it at the same
// points, so the barriers match). The ring is idle here (first pass: untouched; later passes: every wave has left
// the previous pass), so the segment list may borrow its first buffer.
__device__ __forceinline__ void cc_load_table(const CCArgs& A, const CCLds& L, int grp, int p0, int p1, int tid) {
  const int* tbase = A.tile_base + (size_t)grp * (A.n_tiles + 1);
  const int gc0 = UNI(A.gcell[grp]);                              // the group's first cell
  __syncthreads();
  if (tid == 0) *L.s_nseg = 0;
  __syncthreads();
  for (int t = tid; t < A.n_tiles; t += CC_THREADS) {
    const int base = tbase[t], v = tbase[t + 1] - base;
    if (v > 0 && base < p1 && base + v > p0) {
      const int s = atomicAdd(L.s_nseg, 1);                        // order-free: every segment is copied whole
      L.seg[s] = t; L.seg[CC_CAP + s] = base; L.seg[2 * CC_CAP + s] = v;
    }
  }
  __syncthreads();
  const int nseg = UNI(*L.s_nseg);
  for (int s = 0; s < nseg; ++s) {
    const int t = UNI(L.seg[s]), base = UNI(L.seg[CC_CAP + s]), cnt = UNI(L.seg[2 * CC_CAP + s]);
    const int e_t = UNI(A.rowptr[t * TN]);
    const int4* ts = A.steps + step_base(t, e_t, A.n_cells) + UNI(A.cellstep[(size_t)t * (A.n_cells + 1) + gc0].x);   // the group's steps of the tile
    const int j_lo = max(p0 - base, 0), j_hi = min(p1 - base, cnt);
    for (int j = j_lo + tid; j < j_hi; j += CC_THREADS) {
      const int4 r = ts[j];
      const int nx = ts[j + 1].z;                                  // next step's (or the terminal record's) first pair
      L.st4[base + j - p0] = make_int4(r.y, t, r.x, e_t);
      L.st2[base + j - p0] = make_int2(r.z, nx);
    }
  }
  __syncthreads();
}

// ---------------- producer p: steps p, p + 8, ... of every pass, step q into ring buffer q mod NBUF
// Per gathered row: one broadcast ds_read of its record {byte offset of the row, weight}, one v_add (lane offset),
// one buffer_load_dwordx2, one v_pk_fma_f32. No scalar loads and no round trip to memory between the batches of a
// step: the records of 64 pairs are fetched by ONE coalesced vector load (the first 64 of a step already while the
// wave's previous step was summed) and parked in wave-private LDS. A finished row (fp32 sums of columns 2 lane, 2 lane + 1
// in the lane) is split into its three bf16 terms and stored as three dwords.
__device__ __forceinline__ void cc_producer(const CCArgs& A, const CCLds& L, int grp, int g0, int g1, int tid, long long* dbg) {
  const int lane = tid & 63, p = UNI(tid >> 6) - CC_CONSUMERS;     // the wave index in an SGPR: `p`-dependent branches are scalar
  __builtin_amdgcn_s_setprio(CC_PRODUCER_PRIO);
  int2* my_scr = L.scratch + p * 128;
  const int I = A.I;
  const bool live = 2 * lane < I;
  const unsigned lane8 = (unsigned)min(2 * lane, I - 2) * 4u;      // clamped: every lane reads inside the row
  const unsigned ldb = (unsigned)A.ldf * 4u;
  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A.feat), 0, (int)min((size_t)0x7fffffff, ((size_t)(A.n - 1) * A.ldf + I) * 4), 0x00020000);
  // row records (lane r <= rows: {node, first pair}; lane `rows` = the row behind the last: its first pair ends
  // the step) and the first 128 pair records of step i (s0 / w0: pairs 0 .. 63; s1 / w1: pairs 64 .. 127; past the
  // end: the last pair again)
  struct Pre { int2 rinfo; int s0, s1; float w0, w1; };
  auto fetch = [&](int i, Pre& f) {
    const int4 r = uni4(L.st4[i]);
    const int2 pr = uni2(L.st2[i]);
    const size_t e8 = (size_t)8 * r.w;
    f.rinfo = (A.rows + e8 + r.y)[r.z + min(lane, (r.x >> 8) & 31)];
    const int np = pr.y - pr.x;
    const int at0 = pr.x + min(lane, np - 1), at1 = pr.x + min(64 + lane, np - 1);
    const int2 r0 = (A.pair + e8)[at0], r1 = (A.pair + e8)[at1];
    f.s0 = r0.x; f.w0 = __int_as_float(r0.y);
    f.s1 = r1.x; f.w1 = __int_as_float(r1.y);
  };
  // this lane's dword of row `row` in a plane: chunk (lane >> 2) ^ row, dword lane & 3 of it
  const int lane_chunk = lane >> 2, lane_dw = (lane & 3) * 4;
  int qbase = 0;
  for (int p0 = g0; p0 < g1; p0 += CC_CAP) {
    const int p1 = min(g1, p0 + CC_CAP), npass = p1 - p0;
    DBG_T(t0_)
    cc_load_table(A, L, grp, p0, p1, tid);
    DBG_ACC(true, dbg[2], t0_)
    Pre nx = {make_int2(0, 0), 0, 0, 0.f, 0.f};
    if (p < npass) fetch(p, nx);
    for (int i = p; i < npass; i += NPROD) {
      const int q = qbase + i, b = q % NBUF, use = q / NBUF;
      char* a_dst = L.a_buf + b * A_STEP_BYTES;
      DBG_T(s0_)
      DBG_TL(CC_CONSUMERS + p, q, 0)
      const int4 r = uni4(L.st4[i]);
      const int2 pr = uni2(L.st2[i]);
      const int cnt = (r.x >> 8) & 31, pb = pr.x, np = pr.y - pr.x;
      const int nb = (np + PB - 1) / PB;                           // batches of PB pairs in this step
      const size_t e8 = (size_t)8 * r.w;
      const Pre cu = nx;
      const int2 rinfo = cu.rinfo;
      fetch(min(i + NPROD, npass - 1), nx);                        // unconditional (the last steps re-fetch themselves)
      // the ring buffer is claimed only when the step's first row is complete (its pairs have landed and are summed): six
      // buffers serve eight producers, and a buffer held through the first round trip of a step's gather was a buffer the
      // others waited for (in-kernel stamps, round 4: producers 47 % of their time on `done`, consumers 31 % on `full`)
      bool claimed = false;
      auto claim = [&]() {
        if (!claimed) {
          DBG_T(c0_)
          DBG_TL(CC_CONSUMERS + p, q, 1)
          if (use > 0) CC_WAIT_ALL_DONE(L.done + b * CC_CONSUMERS, use);   // the consumers are done with this buffer
          DBG_ACC(true, dbg[0], c0_)
          DBG_TL(CC_CONSUMERS + p, q, 2)
          if (lane < SUBR) L.rowmap[b * SUBR + lane] = lane < cnt ? rinfo.x : -1;   // padding rows: no node
          claimed = true;
        }
      };
      int cur = 0;
      int next_begin = __builtin_amdgcn_readlane(rinfo.y, 1);
      f2 acc = {0.f, 0.f};
      auto store_row = [&]() {                                     // row `cur` is complete: split, three dwords
        claim();
        unsigned hi, mid, lo;
        split_bf16x3(live ? acc : f2{0.f, 0.f}, hi, mid, lo);
        char* d = a_dst + cur * A_ROW_BYTES + (((lane_chunk ^ cur) & 15) << 4) + lane_dw;
        *reinterpret_cast<unsigned*>(d) = hi;
        *reinterpret_cast<unsigned*>(d + A_PLANE_BYTES) = mid;
        *reinterpret_cast<unsigned*>(d + 2 * A_PLANE_BYTES) = lo;
      };
      auto flush = [&]() {
        store_row();
        acc = f2{0.f, 0.f};
        ++cur;
        next_begin = __builtin_amdgcn_readlane(rinfo.y, cur + 1);
      };
      // The pair records of 128 pairs ("segment") are parked in wave-private LDS -- for the first segment from the
      // registers fetched a step ago -- and read back as broadcasts: no refill inside the stream of row loads, which
      // therefore runs through a whole segment with four batches of 8 rows in flight (the first persistent version
      // re-primed its pipeline every 64 pairs and fetched the next 64 records in between, synchronously: ~4 of the 9 us
      // a D = 4 step of ~80 pairs took). Steps of more than 128 pairs (dense clumps) fetch the next segment's records
      // between segments. (Refilling a ring from inside the loop was tried three ways -- registers, buffer loads,
      // LDS-DMA: hipcc guarded each with vmcnt(1) / vmcnt(0) at the loop head, or, for the DMA, in front of every LDS
      // read, draining the rows in flight once per 32 pairs.) Slots behind the step's last pair: the last pair's row
      // (a hot line) with weight 0.
      int rs0 = cu.s0, rs1 = cu.s1;
      float rw0 = cu.w0, rw1 = cu.w1;
      // NBF batches of PB rows in flight. EVERY stage issues exactly PB row loads (past the end: re-reads of the
      // last pair's row, L1 hits): only then can hipcc count its s_waitcnt vmcnt() and let batch j be summed while
      // the younger batches are in flight.
      f2 fbuf[NBF][PB];
#define CC_ISSUE(SLOT, OFF)                                                                                  \
      {                                                                                                      \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                                     \
          const unsigned ro = (unsigned)my_scr[(OFF) + u].x;                                                 \
          fbuf[SLOT][u] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(frs, (int)(ro + lane8), 0, 0)); \
        }                                                                                                    \
      }
#define CC_SUM(SLOT, OFF, BJ)                                                                                \
      {                                                                                                      \
        const int valid = min(PB, np - PB * (BJ)), q0 = pb + PB * (BJ);                                      \
        float wv[PB];                              /* the weights: broadcast reads of the parked records */  \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) wv[u] = __int_as_float(my_scr[(OFF) + u].y);          \
        if (valid == PB) {                                                                                   \
          _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                                   \
            if (q0 + u == next_begin) flush();                         /* wave-uniform */                    \
            const float w = wv[u];                                                                           \
            acc = __builtin_elementwise_fma(f2{w, w}, fbuf[SLOT][u], acc);                                   \
          }                                                                                                  \
        } else {                                                                                             \
          _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                                   \
            if (u < valid) {                                           /* wave-uniform */                    \
              if (q0 + u == next_begin) flush();                                                             \
              const float w = wv[u];                                                                         \
              acc = __builtin_elementwise_fma(f2{w, w}, fbuf[SLOT][u], acc);                                 \
            }                                                                                                \
          }                                                                                                  \
        }                                                                                                    \
      }
      // stage J of trip h (32 pairs): batch 4 h + J summed with three younger batches in flight, then batch
      // 4 h + J + 4 requested into the slot just freed (the last trip of a segment only sums)
#define CC_STAGE(J, ISSUE_NEXT)                                                                              \
      CC_SUM(J, 32 * HQ + PB * (J), b0 + 4 * HQ + (J))                                                       \
      if (b0 + 4 * HQ + (J) + 1 >= nb) { done = true; break; }                                               \
      if (ISSUE_NEXT) CC_ISSUE(J, 32 * HQ + 32 + PB * (J))
      static_assert(PB == 8 && NBF == 4, "the slot rotation below is written out for four batches of 8 in flight");
      bool done = false;
      for (int b0 = 0; !done; b0 += 16) {                            // one segment = 16 batches = 128 pairs
        if (b0 > 0) {
          const size_t at0 = e8 + pb + min(PB * b0 + lane, np - 1), at1 = e8 + pb + min(PB * b0 + 64 + lane, np - 1);
          const int2 r0 = A.pair[at0], r1 = A.pair[at1];
          rs0 = r0.x; rw0 = __int_as_float(r0.y);
          rs1 = r1.x; rw1 = __int_as_float(r1.y);
        }
        __builtin_amdgcn_wave_barrier();
        my_scr[lane] = make_int2((int)((unsigned)rs0 * ldb), PB * b0 + lane < np ? __float_as_int(rw0) : 0);
        my_scr[64 + lane] = make_int2((int)((unsigned)rs1 * ldb), PB * b0 + 64 + lane < np ? __float_as_int(rw1) : 0);
        __builtin_amdgcn_wave_barrier();
        CC_ISSUE(0, 0)
        CC_ISSUE(1, PB)
        CC_ISSUE(2, 2 * PB)
        CC_ISSUE(3, 3 * PB)
        do {
#define HQ 0
          CC_STAGE(0, true) CC_STAGE(1, true) CC_STAGE(2, true) CC_STAGE(3, true)
#undef HQ
#define HQ 1
          CC_STAGE(0, true) CC_STAGE(1, true) CC_STAGE(2, true) CC_STAGE(3, true)
#undef HQ
#define HQ 2
          CC_STAGE(0, true) CC_STAGE(1, true) CC_STAGE(2, true) CC_STAGE(3, true)
#undef HQ
#define HQ 3
          CC_STAGE(0, false) CC_STAGE(1, false) CC_STAGE(2, false) CC_STAGE(3, false)
#undef HQ
        } while (0);
      }
#undef CC_STAGE
#undef CC_SUM
#undef CC_ISSUE
      // The loop leaves up to three batches in flight (issued past the step's end, never summed). They
      // are "used" here so that hipcc retires them now: left pending, their registers -- reused as temporaries at the
      // loop head -- made every wait there a vmcnt(0) that drained the row loads in flight (seen in the ISA).
#pragma unroll
      for (int sl = 0; sl < NBF; ++sl)
#pragma unroll
        for (int u = 0; u < PB; ++u) asm volatile("" ::"v"(fbuf[sl][u]));
      store_row();
      CC_RELEASE_FENCE();
      if (lane == 0) __hip_atomic_store(&L.full[b], use + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      DBG_TL(CC_CONSUMERS + p, q, 3)
      DBG_ACC(true, dbg[1], s0_)
    }
    qbase += npass;
  }
}

// ---------------- consumer: 16 output columns, all steps in order
// Two consumer waves share a SIMD (waves w and w + 4). A step is ONE 16 x 16 tile per wave, the K = I contraction in NS
// slabs of 32, six bf16 term products per slab; the MFMA takes the filter fragment as its first operand and the packed A
// rows as its second, so the result comes out transposed -- lane l holds output columns 4 (l >> 4) .. + 3 of packed row
// l & 15 -- and the scatter into the node's accumulator row is one ds_read_b128 + ds_write_b128 per lane. When the range
// crosses into the next tile (and at its end) the wave writes its 16 columns of the accumulator to partial slot
// (workgroup + tile) and zeroes them.
//
// Per slab s, in this order (S0 / S1 / BIG = three accumulators; an MFMA never follows one it depends on directly: the
// consumers-alone build ran 4 % slower with the small products in ONE chain):
//     S0 += b_lo[s] a_hi[s]    S1 += b_mid[s] a_mid[s]    S0 += b_mid[s] a_hi[s]
//     S1 += b_hi[s] a_mid[s]   S0 += b_hi[s] a_lo[s]      BIG += b_hi[s] a_hi[s]
// and the step's result is (S0 + S1) + BIG: the five products of order 2^-8 and 2^-16 never meet the leading one before
// the end, whatever the slab order.
//
// THE FILTER FRAGMENT (NS slabs x 3 terms, one quad of 8 bf16 each: 48 VGPRs at I = 128) is single-buffered and reloaded
// IN PLACE, quad by quad, inside the last step of a cell: the moment a quad's last MFMA of the step is issued, the next
// cell's quad is requested into the same registers. The next step uses the quads in the same order, so every quad has 21
// to 23 MFMAs of 24 to arrive -- with the sibling wave's burst in between about twice that in time: more than an L2 hit's
// latency. (Two register sets with the next cell's fetched a whole cell ahead, round 3's scheme, would need 96 VGPRs for
// the fragments alone; a 16-wave workgroup has 128 per lane.) hipcc cannot express either the in-flight registers or
// the counted waits, so the loads are issued from inline asm ("+v": same register in and out) and the waits are written
// by hand: the consumers issue no other vector loads (flush_acc drains its stores on the spot), loads retire in order,
// and quads are always requested in the cyclic order j = 3 s + {lo, mid, hi} in which they are used. With Q = 3 NS quads:
// before quad j's first use, a step that reloads waits vmcnt(Q - 1) (outstanding, oldest first: j .. Q - 1 of the previous
// reload, then this step's 0 .. j - 1); a step that does not, vmcnt(Q - 1 - j) (outstanding: j .. Q - 1, or nothing).
// tools/check_contconv_isa.py (run by the CPU tests) walks the disassembly's control-flow graph and verifies that nothing
// reads or writes a fragment register between its load and the wait that covers it. K slabs beyond I hold zeros on both
// sides (the producers write zeros beyond I, the shuffle kernel pads the filters).
//
// THE A FRAGMENTS (3 quads per slab) are streamed: slabs 0 .. 2 are requested at the step's start (the first two quads of
// slab 0 already in the middle of the previous step, when that step's flag showed the buffer full), slab 3 when slab 0's
// products are issued -- 36 VGPRs instead of 48.
template <int NS>
__device__ __forceinline__ void cc_consumer(const CCArgs& A, const CCLds& L, int grp, int slot0, int g0, int g1, int tid, long long* dbg) {
  const int lane = tid & 63, cw = UNI(tid >> 6);
  if (cw < CC_CONSUMERS / 2) __builtin_amdgcn_s_setprio(CC_PRIO_HI);
  else __builtin_amdgcn_s_setprio(CC_PRIO_LO);
  const int cb = blockIdx.z * CC_CONSUMERS + cw;                   // 16-column block of the output
  const int colblocks = A.colblocks;
  const bool has_cols = cb < colblocks;
  int cur_tile = -1;
  auto flush_acc = [&](int tile) {
    int ln = lane;
    asm volatile("" : "+v"(ln));     // opaque: keeps hipcc from hoisting this rare path's per-lane addresses out of the step
                                     // loops, where they cost VGPRs the step body needs
    float* dst = A.partial + ((size_t)(slot0 + tile) * TN) * A.OP + (size_t)blockIdx.z * 128 + cw * 16 + (ln & 3) * 4;
    float* src = L.out_acc + cw * 16 + (ln & 3) * 4;
#pragma unroll 4
    for (int r0 = 0; r0 < TN; r0 += 16) {
      const int r = r0 + (ln >> 2);
      f4* a = reinterpret_cast<f4*>(src + r * LDO);
      *reinterpret_cast<f4*>(dst + (size_t)r * A.OP) = *a;
      *a = f4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores and loads retire out of order with each other: keep the
  };                                                   // hand-counted vmcnt of the fragment loads among loads only
  static_assert(NS == 1 || NS == 2 || NS == 4, "K slabs of 32");
  constexpr int NQ = 3 * NS;                         // fragment quads per lane
  // (quads are held as 4 x i32, a type hipcc never takes apart: as 8 x bf16 it unpacked and re-packed them by halves around
  // every block boundary -- v_perm / v_lshrrev on registers with loads in flight)
  q4 B[NQ];                                          // [3 s + u], u = 0 lo, 1 mid, 2 hi: the order of use
#pragma unroll
  for (int j = 0; j < NQ; ++j) B[j] = q4{0, 0, 0, 0};
  // Addressing: the wave's column block of cell 0 as a scalar base; the cell's byte offset + 16 lane (+ 4096 per four
  // quads) in a VGPR (the filter matrix is checked < 2 GiB on the host), the quad inside its four as an immediate offset.
  const char* fbase = reinterpret_cast<const char*>(A.filt) + (size_t)min(cb, colblocks - 1) * (NQ * 1024);
  const unsigned cell_stride = (unsigned)colblocks * (NQ * 1024u);                    // bytes between cells
  const unsigned lane16 = (unsigned)lane * 16u;
#define CC_LOAD_QUAD(J, OFFS)                                                                                \
  {                                                                                                          \
    if constexpr (((J) & 3) == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(B[J]) : "v"(OFFS[(J) >> 2]), "s"(fbase) : "memory"); \
    if constexpr (((J) & 3) == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "+v"(B[J]) : "v"(OFFS[(J) >> 2]), "s"(fbase) : "memory"); \
    if constexpr (((J) & 3) == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "+v"(B[J]) : "v"(OFFS[(J) >> 2]), "s"(fbase) : "memory"); \
    if constexpr (((J) & 3) == 3) asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "+v"(B[J]) : "v"(OFFS[(J) >> 2]), "s"(fbase) : "memory"); \
  }
#define CC_QUAD_OFFSETS(NAME, CELL)                                                                          \
  unsigned NAME[3];                                                                                          \
  NAME[0] = (unsigned)(CELL) * cell_stride + lane16; NAME[1] = NAME[0] + 4096u; NAME[2] = NAME[0] + 8192u;
  // "all but the N youngest loads have landed", tied to quad J's registers: every later use of them depends on it
#define CC_QUAD_LANDED(J, N) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(B[J]) : "n"(N));
#define CC_DRAIN()                                                                                           \
  {                                                                                                          \
    if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]));         \
    if constexpr (NS == 2)                                                                                   \
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[NS > 1 ? 3 : 0]), "+v"(B[NS > 1 ? 4 : 0]), "+v"(B[NS > 1 ? 5 : 0])); \
    if constexpr (NS == 4)                                                                                   \
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[NS > 1 ? 3 : 0]), "+v"(B[NS > 1 ? 4 : 0]), "+v"(B[NS > 1 ? 5 : 0]), \
                   "+v"(B[NS > 2 ? 6 : 0]), "+v"(B[NS > 2 ? 7 : 0]), "+v"(B[NS > 2 ? 8 : 0]), "+v"(B[NS > 2 ? 9 : 0]), "+v"(B[NS > 2 ? 10 : 0]), \
                   "+v"(B[NS > 2 ? 11 : 0]));                                                                 \
  }
  const int a_row = (lane & 15) * A_ROW_BYTES;                       // this lane's row of a step's A planes ...
  const int sw0 = (((lane >> 4) ^ (lane & 15)) & 15) << 4;           // ... and its chunk of K slab 0 (slab s: ^ (s << 6))
  const int o_lane = cw * 16 + 4 * (lane >> 4);
  int qbase = 0;
  for (int p0 = g0; p0 < g1; p0 += CC_CAP) {
    const int p1 = min(g1, p0 + CC_CAP), npass = p1 - p0;
    cc_load_table(A, L, grp, p0, p1, tid);
    // pipeline registers: the A quads and the row map entry of the step about to run (valid when `have`), that step's tile
    q4 a[NS][3];                                       // [slab][0 hi, 1 mid, 2 lo]
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) a[s_][0] = a[s_][1] = a[s_][2] = q4{0, 0, 0, 0};
    int node_n = -1;
    bool have = false;
    int tile_n = UNI(L.st4[0].y);
#define CC_A(BASE, PLANE, S) (*reinterpret_cast<const q4*>((BASE) + (PLANE) * A_PLANE_BYTES + (sw0 ^ ((S) << 6))))
    // the three quads of slab S of ring buffer BUF into the slab's registers (whose previous contents are multiplied out)
#define CC_READ_SLAB(BUF, S)                                                                                 \
    if constexpr ((S) < NS) {                                                                                \
      constexpr int S_ = (S) < NS ? (S) : 0;                                                                 \
      const char* ab_ = L.a_buf + (BUF) * A_STEP_BYTES + a_row;                                              \
      a[S_][0] = CC_A(ab_, 0, S_); a[S_][1] = CC_A(ab_, 1, S_); a[S_][2] = CC_A(ab_, 2, S_);                 \
    }
#define CC_MFMA(ACC, BQ, AV) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, BQ), __builtin_bit_cast(b8, AV), ACC, 0, 0, 0);
    // slab S: its six products, every fragment quad re-requested behind its last use when the step reloads
#define CC_SLAB(S, RELOAD)                                                                                   \
    if constexpr ((S) < NS) {                                                                                \
      constexpr int S_ = (S) < NS ? (S) : 0;                                                                 \
      CC_QUAD_LANDED(3 * S_, (RELOAD) ? NQ - 1 : NQ - 1 - 3 * S_)                                            \
      CC_MFMA(accs0, B[3 * S_], a[S_][0])                                                                    \
      if (RELOAD) CC_LOAD_QUAD(3 * S_, offs_)                                                                \
      CC_QUAD_LANDED(3 * S_ + 1, (RELOAD) ? NQ - 1 : NQ - 2 - 3 * S_)                                        \
      CC_MFMA(accs1, B[3 * S_ + 1], a[S_][1])                                                                \
      CC_MFMA(accs0, B[3 * S_ + 1], a[S_][0])                                                                \
      if (RELOAD) CC_LOAD_QUAD(3 * S_ + 1, offs_)                                                            \
      CC_QUAD_LANDED(3 * S_ + 2, (RELOAD) ? NQ - 1 : NQ - 3 - 3 * S_)                                        \
      CC_MFMA(accs1, B[3 * S_ + 2], a[S_][1])                                                                \
      CC_MFMA(accs0, B[3 * S_ + 2], a[S_][2])                                                                \
      CC_MFMA(accb, B[3 * S_ + 2], a[S_][0])                                                                 \
      if (RELOAD) CC_LOAD_QUAD(3 * S_ + 2, offs_)                                                            \
    }
    // SOFTWARE PIPELINE ACROSS STEPS (round 4). In-kernel stamps and a producers-off build of the first bf16 version: a
    // consumer wave took ~2100 cycles per step for 384 cycles of its own MFMAs -- an LDS read takes 300-400 cycles with 16
    // waves on the array, and with each slab's quads requested one slab (96 MFMA cycles) ahead every slab waited out the
    // difference. Now the quads of the NEXT step are requested from inside this one, into the registers this step has just
    // multiplied out: slabs 0-1 behind slab 1's products, slabs 2-3 behind slab 3's -- when the next step's buffer is
    // already full at that point (the normal case with the producers ahead); otherwise at the next step's start, all
    // twelve at once. One set of 48 registers holds "the step about to run".
#define CC_STEP(IDX, RELOAD, NEXT_CELL)                                                                      \
    {                                                                                                        \
      const int q_ = qbase + (IDX), b = q_ % NBUF, use = q_ / NBUF;                                          \
      if (tile_n != cur_tile) {                        /* the range crosses into the next tile */          \
        if (cur_tile >= 0 && has_cols) flush_acc(cur_tile);                                                  \
        cur_tile = tile_n;                                                                                   \
      }                                                                                                      \
      DBG_TL(cw, q_, 0)                                                                                      \
      if (!have) {                                     /* not prefetched: the step was not full yet */     \
        DBG_T(c0_) CC_WAIT(L.full[b], >= use + 1);                                                           \
        DBG_ACC(true, dbg[0], c0_)                                                                           \
        if (has_cols) {                                                                                      \
          node_n = L.rowmap[b * SUBR + (lane & 15)];                                                         \
          CC_READ_SLAB(b, 0) CC_READ_SLAB(b, 1) CC_READ_SLAB(b, 2) CC_READ_SLAB(b, 3)                        \
        }                                                                                                    \
      }                                                                                                      \
      DBG_T(w0_) DBG_TL(cw, q_, 1)                                                                           \
      /* control reads for step IDX + 1, issued now, used half a burst later */                             \
      const int nidx_ = min((IDX) + 1, npass - 1), qn_ = q_ + 1, bn_ = qn_ % NBUF;                           \
      const int tile_v_ = L.st4[nidx_].y;                                                                    \
      const int flag_v_ = __hip_atomic_load(&L.full[bn_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   \
      if (has_cols) {                                                                                        \
        f4v accs0 = {0.f, 0.f, 0.f, 0.f}, accs1 = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};         \
        const int node = node_n;                                                                             \
        CC_QUAD_OFFSETS(offs_, NEXT_CELL)                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        CC_SLAB(0, RELOAD)                                                                                   \
        CC_SLAB(1, RELOAD)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* mid-burst: is step IDX + 1 already full? then its row map entry and the quads of its slabs 0-1 come now */ \
        tile_n = UNI(tile_v_);                                                                               \
        have = (IDX) + 1 < npass && UNI(flag_v_) >= qn_ / NBUF + 1;                                          \
        if (have) {                                                                                          \
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                    \
          node_n = L.rowmap[bn_ * SUBR + (lane & 15)];                                                       \
          CC_READ_SLAB(bn_, 0) CC_READ_SLAB(bn_, 1)                                                          \
        }                                                                                                    \
        /* the node's accumulator quad (padding rows: row 0, discarded; in program order behind the previous   \
           step's write of this wave) */                                                                     \
        f4* o = reinterpret_cast<f4*>(L.out_acc + max(node, 0) * LDO + o_lane);                              \
        const f4 old = *o;                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        CC_SLAB(2, RELOAD)                                                                                   \
        CC_SLAB(3, RELOAD)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* Buffer b goes back to its producer: its quads and row map entry are in registers (the reads were waited for  \
           by the MFMAs). No s_waitcnt here: the LDS executes a wave's instructions in order, so the flag store lands      \
           behind every earlier read of this wave -- the fence is for the compiler only (a release fence proper would     \
           also wait out the reads just issued for the next step). */                                       \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                             \
        if (lane == 0) __hip_atomic_store(&L.done[b * CC_CONSUMERS + cw], use + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                                                             \
        /* the scatter BEFORE the next step's slabs 2-3 are requested: hipcc waits for `old` with lgkmcnt(0) (the reads \
           behind it are conditional), which would also wait out reads issued a moment ago */              \
        if (node >= 0)                                                                                       \
          *o = f4{old[0] + ((accs0[0] + accs1[0]) + accb[0]), old[1] + ((accs0[1] + accs1[1]) + accb[1]),    \
                  old[2] + ((accs0[2] + accs1[2]) + accb[2]), old[3] + ((accs0[3] + accs1[3]) + accb[3])};   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (have) { CC_READ_SLAB(bn_, 2) CC_READ_SLAB(bn_, 3) }                                              \
      } else {                                                                                               \
        tile_n = UNI(tile_v_);                                                                               \
        have = false;                                                                                        \
        if (lane == 0) __hip_atomic_store(&L.done[b * CC_CONSUMERS + cw], use + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      }                                                                                                      \
      DBG_ACC(true, dbg[1], w0_) DBG_TL(cw, q_, 2)                                                           \
    }
    // meta = cell | rows << 8 | steps left in the cell << 16. A cell's steps but the last run in a loop that requests
    // nothing; the last one requests the next cell's fragment (a pass's last step: its own cell's again -- the next pass
    // starts afresh, and loads to one register retire in order).
    auto meta_at = [&](int i) { return L.st4[min(i, npass - 1)].x; };
    int meta = UNI(meta_at(0));
    if (has_cols) {
      CC_QUAD_OFFSETS(offs0_, meta & 0xff)
      CC_LOAD_QUAD(0, offs0_) CC_LOAD_QUAD(1, offs0_) CC_LOAD_QUAD(2, offs0_)
      if constexpr (NS > 1) { CC_LOAD_QUAD(NS > 1 ? 3 : 0, offs0_) CC_LOAD_QUAD(NS > 1 ? 4 : 0, offs0_) CC_LOAD_QUAD(NS > 1 ? 5 : 0, offs0_) }
      if constexpr (NS > 2) {
        CC_LOAD_QUAD(NS > 2 ? 6 : 0, offs0_) CC_LOAD_QUAD(NS > 2 ? 7 : 0, offs0_) CC_LOAD_QUAD(NS > 2 ? 8 : 0, offs0_)
        CC_LOAD_QUAD(NS > 2 ? 9 : 0, offs0_) CC_LOAD_QUAD(NS > 2 ? 10 : 0, offs0_) CC_LOAD_QUAD(NS > 2 ? 11 : 0, offs0_)
      }
      // a pass's first request is waited out on the spot (once per CC_CAP steps): hipcc is free to move the quads into the
      // registers its step loop keeps them in -- it does, for NS < 4 -- and a copy must not see a load in flight
      CC_DRAIN()
    }
    for (int i = 0; i < npass;) {
      const int rem = min((meta >> 16) & 0xff, npass - i);
      const int meta_nv = meta_at(i + rem);                             // the next cell's record, used by the reloading step
      for (int u = 0; u + 1 < rem; ++u) CC_STEP(i + u, false, 0)
      const int meta_n = UNI(meta_nv);
      CC_STEP(i + rem - 1, true, meta_n & 0xff)
      i += rem; meta = meta_n;
    }
    // the pass's last request (its own cell's fragment again) lands before anything else may use the registers
    CC_DRAIN()
#undef CC_STEP
#undef CC_READ_SLAB
#undef CC_SLAB
#undef CC_MFMA
#undef CC_A
    qbase += npass;
  }
#undef CC_DRAIN
#undef CC_QUAD_LANDED
#undef CC_QUAD_OFFSETS
#undef CC_LOAD_QUAD
  if (has_cols && cur_tile >= 0) flush_acc(cur_tile);
}

template <int NS>
__global__ __launch_bounds__(CC_THREADS) void contconv_stream_kernel(const CCArgs A) {
  // f4-typed so that the dynamic region starts 16-byte aligned behind the static __shared__ variables (declared as
  // float[] it began at an 8-byte offset and every ds_read_b128 / ds_write_b64 took the unaligned path)
  extern __shared__ f4 lds_aligned[];
  __shared__ int statics[32 + NBUF * CC_CONSUMERS];                // s_red[16], s_nseg, full[8], done[NBUF][8]
  const CCLds L = cc_lds(reinterpret_cast<float*>(lds_aligned), statics);
  const int tid = threadIdx.x, wave = UNI(tid >> 6);
  DBG_T(dbg_t0)                                     // (probe build: the workgroup's start stamp)
  long long dbg_wait[4] = {0, 0, 0, 0};        // [0] time on the flags, [1] in the steps, [2] in the table loads, [3] whole role

  // ---- this workgroup's range of the global step sequence: cut where the running COST (per step: max(CC_COST_MIN,
  // pairs)) crosses w / G of its total -- equal matrix work where the tiles are sparse, equal gather work where they
  // are dense. The cuts are part of the pair lists (contconv_plan_kernel, once per graph and resolution).
  if (tid < 8) L.full[tid] = 0;
  if (tid < NBUF * CC_CONSUMERS) L.done[tid] = 0;
  // workgroup w = blockIdx.x runs on XCD w mod 8: it takes group w mod groups, and is the group's (w / groups)-th
  const int grp = blockIdx.x % A.groups, wig = blockIdx.x / A.groups, wpg = CC_GRID / A.groups;
  const int g0 = UNI(A.cuts[grp * (wpg + 1) + wig]), g1 = UNI(A.cuts[grp * (wpg + 1) + wig + 1]);
  if (g0 >= g1) return;                                            // uniform: the whole workgroup leaves
  const int slot0 = grp * (wpg + A.n_tiles) + wig;                 // partial slot of (this workgroup, tile t) = slot0 + t
  for (int i = tid; i < TN * LDO / 4; i += CC_THREADS) reinterpret_cast<f4*>(L.out_acc)[i] = f4{0.f, 0.f, 0.f, 0.f};
  // (the first cc_load_table's barriers order the zeroing before any consumer's first scatter)
  DBG_T(r0_)
  if (wave >= CC_CONSUMERS) cc_producer(A, L, grp, g0, g1, tid, dbg_wait);
  else cc_consumer<NS>(A, L, grp, slot0, g0, g1, tid, dbg_wait);
  DBG_ACC(true, dbg_wait[3], r0_)
#ifdef NBD_CC_TRACE
  const int lane = tid & 63;
  __shared__ long long s_dbg[16][4];
  if (lane == 0) { for (int i = 0; i < 4; ++i) s_dbg[wave][i] = dbg_wait[i]; }
  __syncthreads();
  if (tid < 64 && g_cc_trace && blockIdx.z == 0) {
    long long* t = g_cc_trace + (size_t)blockIdx.x * 72;
    if (tid == 0) {
      t[0] = dbg_t0; t[1] = g1 - g0; t[2] = __builtin_amdgcn_s_memrealtime();
      t[3] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    t[8 + tid] = s_dbg[tid >> 2][tid & 3];
  }
#endif
}

// out = act(scale * sum of the tile's partial slots), in (group, workgroup) order. grid (tiles, 16): 8 rows of a tile each,
// one output quad per thread, so that a block's loads of up to FIN_WB slots go out as ONE round trip (the slots were written
// by other XCDs a moment ago: every load is a cold trip through the fabric -- with 32 rows per block and three slots per
// trip the D = 6 layer's ~10 slots per tile took 26.9 us, this form 21.3: 82 MB of partials at 3.9 TB/s, the bytes now).
// The workgroups whose ranges meet the tile's steps of a group, [tile_base[g][t], tile_base[g][t + 1]), are the ones that
// wrote a slot for it: slot(group g, its j-th workgroup, tile t) = g (W + tiles) + j + t -- a merge-path numbering inside
// every group, unique and independent of timing. The list is built by wave 0, 64 workgroups of a group per trip.
constexpr int FIN_ROWS = 8, FIN_WB = 6;
__global__ __launch_bounds__(256) void contconv_stream_finish_kernel(
    const float* __restrict__ partial, const int* __restrict__ tile_nsteps, const int* __restrict__ cuts,
    const int* __restrict__ tile_base, int n_tiles, int groups,
    const float* __restrict__ rowscale, int act, float* __restrict__ out, int ldo, int n, int O, int OP) {
  __shared__ int s_slot[CC_GRID];
  __shared__ int s_ns;
  const int tid = threadIdx.x, tile = blockIdx.x, lane = tid & 63;
  const int W = CC_GRID / groups;
  const int cnt_all = tile_nsteps[tile];
  if (tid < 64) {
    // lane g < groups: the tile's steps in group g
    int base = 0, cnt = 0;
    if (lane < groups && cnt_all > 0) {
      const int* tb = tile_base + (size_t)lane * (n_tiles + 1);
      base = tb[tile]; cnt = tb[tile + 1] - base;
    }
    int ns = 0;
    for (int g = 0; g < groups; ++g) {
      const int gb = __shfl(base, g), gc = __shfl(cnt, g);
      if (gc <= 0) continue;                                         // wave-uniform
      const int* cut = cuts + g * (W + 1);
      for (int w0 = 0; w0 < W; w0 += 64) {
        const int w = w0 + lane;
        const int c0 = w < W ? cut[w] : 0, c1 = w < W ? cut[w + 1] : 0;
        const bool hit = w < W && c0 < c1 && c0 < gb + gc && c1 > gb;     // a non-empty range that meets the tile's steps
        const unsigned long long m = __ballot(hit);
        if (hit) s_slot[ns + __popcll(m & ((1ull << lane) - 1))] = g * (W + n_tiles) + w + tile;
        ns += __popcll(m);
      }
    }
    if (lane == 0) s_ns = ns;
  }
  __syncthreads();
  const int ns = s_ns;
  const int r0 = tile * TN + blockIdx.y * FIN_ROWS;
  const int qpr = OP / 4;                                            // quads per row
  for (int e = tid; e < FIN_ROWS * qpr; e += 256) {
    const int rl = e / qpr, c = (e - rl * qpr) * 4, row = r0 + rl;
    if (row >= n || c >= O) continue;
    f4 v = cnt_all < 0 ? f4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")}      // refused tile
                       : f4{0.f, 0.f, 0.f, 0.f};
    const float* src = partial + (size_t)(row - tile * TN) * OP + c;
    for (int i0 = 0; i0 < ns; i0 += FIN_WB) {
      f4 x[FIN_WB];
#pragma unroll
      for (int q = 0; q < FIN_WB; ++q)
        x[q] = i0 + q < ns ? *reinterpret_cast<const f4*>(src + (size_t)s_slot[i0 + q] * TN * OP) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < FIN_WB; ++q) {
        if (i0 + q >= ns) break;                                     // uniform; a skipped slot adds nothing, not + 0
        v = f4{v[0] + x[q][0], v[1] + x[q][1], v[2] + x[q][2], v[3] + x[q][3]};
      }
    }
    const float sc = rowscale ? rowscale[row] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c + j < O) {
        const float y = rowscale ? __fmul_rn(v[j], sc) : v[j];
        out[(size_t)row * ldo + c + j] = act == 1 ? tanhf(y) : y;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------- filter gradient
// dF[k][i][o] = sum over the touched blocks (node n, cell k) of A[n][k][i] * g[n][o], A[n][k] = the block's weighted
// sum of gathered feature rows exactly as the forward kernel forms it, g = scale * act'(out) * dout (contconv.py:92-98
// differentiated with respect to `filters`). The binned matrix is not formed here either.
// 512 workgroups of 1024 threads, each over ONE contiguous range of the (cell, tile) units in cell-major order, the
// ranges of equal cost (contconv_wplan_kernel). Inside a cell a workgroup walks the rows of its tiles 16 at a time -- rows
// of different tiles share a step (the contraction runs over rows: no padding but a segment's last step) -- and per step
// its sixteen waves build the 16 A rows (one each) and copy the 16 g rows into LDS (double-buffered: one barrier per
// step), then multiply A^T (I x 16) by g (16 x O) with v_mfma_f32_16x16x4_f32; wave w keeps four 16 x 16 blocks of the
// I x O result in registers for the whole cell. When the range leaves a cell its block goes to partial slot
// (range + cell); the finishing kernel adds a cell's slots in range order: deterministic, no float atomics.
constexpr int WG_LD = 144;       // LDS row stride (floats): the four k-rows a fragment read touches land on distinct banks
constexpr int WG_MAXT = 64;      // tiles per slab (one wave scans their row counts)

struct WGArgs {
  const float* feat; int ldf, I; const float* g; int ldg, O;
  const int* rowptr; int n, n_tiles, n_cells;
  const int2* desc; const int2* rows; const int2* pair; const int* tile_nsteps;
  int* ucut;                   // [WG_RANGES + 1] first unit of every workgroup's range (contconv_wplan_kernel)
  float* partial;              // [(range + cell) < WG_RANGES + cells][I][O]
};
constexpr int WG_RANGES = 512;   // workgroups of the filter-gradient kernel (two per CU in turn)

// The units of the filter gradient's work -- (cell, tile) pairs in CELL-major order, u = cell * tiles + tile -- cut into
// WG_RANGES contiguous ranges of equal cost (per unit: its pairs + 8 per row: a row costs its share of the MFMA step and
// of the three round trips, a pair one gathered row). A grid of (cells x slabs of tiles), the first form, ran as long as
// its heaviest cell: with self loops every node sends a row to each of the 8 cells around the grid centre (|r| = 0 maps
// there), so those cells hold N rows each while the outer ones hold a handful -- 200+ us per launch on a 2 000-node
// batch whatever the kernel inside did. One workgroup, once per call.
__global__ __launch_bounds__(1024) void contconv_wplan_kernel(const WGArgs A) {
  __shared__ long long red[16];
  __shared__ long long s_tot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int U = A.n_cells * A.n_tiles;
  auto cost = [&](int u) -> int {
    if (u >= U) return 0;
    const int cell = u / A.n_tiles, tile = u - cell * A.n_tiles;
    if (A.tile_nsteps[tile] < 0) return 0;                 // a tile the pair kernel refused
    const int2 d = A.desc[(size_t)tile * A.n_cells + cell];
    if (d.y <= 0) return 0;
    const int2* rw = A.rows + (size_t)8 * A.rowptr[tile * TN] + tile + d.x;
    return (rw[d.y].y - rw[0].y) + 8 * d.y;
  };
  long long mine = 0;
  for (int u = tid; u < U; u += 1024) mine += cost(u);
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if (lane == 0) red[wave] = mine;
  __syncthreads();
  if (tid == 0) { long long t = 0; for (int i = 0; i < 16; ++i) t += red[i]; s_tot = t; }
  __syncthreads();
  const long long Ctot = s_tot;
  // every cut starts at unit 0 (what a cut at cost 0 is; with no rows at all the last range visits every cell and writes
  // zeros); the units below move the cuts that fall behind them
  for (int r = tid; r < WG_RANGES; r += 1024) A.ucut[r] = 0;
  if (tid == 0) A.ucut[WG_RANGES] = U;
  if (Ctot == 0) return;
  long long carry = 0;
  for (int c0 = 0; c0 < U; c0 += 1024) {
    const int u = c0 + tid;
    const long long c = cost(u);
    long long incl = c;
    for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    __syncthreads();
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    long long woff = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const long long x = red[i]; woff += i < wave ? x : 0; tot += x; }
    const long long e0 = carry + woff + incl - c, e1 = e0 + c;        // cost before / behind unit u
    if (c > 0) {                                             // the cuts B_r = Ctot r / G that fall in (e0, e1] start at u + 1
      long long r = e0 * WG_RANGES / Ctot;
      while (r < WG_RANGES && Ctot * r / WG_RANGES <= e0) ++r;
      for (; r < WG_RANGES && Ctot * r / WG_RANGES <= e1; ++r) A.ucut[r] = u + 1;
    }
    carry += tot;
  }
}

// 16 waves per workgroup: wave w builds row w of the step (in training batches the rows are SKEWED -- radius_graph keeps
// the first 32 hits by index, so the low-index bodies of every graph are listed by everyone around them and their
// blocks hold 50-100 pairs next to blocks of 3 -- and hub rows are consecutive: with four rows per wave a step of hubs
// cost one wave 400 dependent-latency-bound pairs, 20 us per step, 220-350 us per launch in four successive forms of
// this kernel), then multiplies one 16-row block of A^T by four 16-column blocks of g.
constexpr int WG_WAVES = 16;
__global__ __launch_bounds__(64 * WG_WAVES) void contconv_wgrad_kernel(const WGArgs A) {
  __shared__ __attribute__((aligned(16))) float a_s[2][16][WG_LD];
  __shared__ __attribute__((aligned(16))) float g_s[2][16][WG_LD];
  __shared__ int t_pref[WG_MAXT + 1], t_row0[WG_MAXT];
  __shared__ long long t_e8[WG_MAXT];
  const int tid = threadIdx.x, lane = tid & 63, wave = UNI(tid >> 6);
  const int range = blockIdx.x;
  const int u0 = UNI(A.ucut[range]), u1 = UNI(A.ucut[range + 1]);
  if (u0 >= u1) return;                                    // an empty range writes no slot (the finishing kernel skips it)
  const int I = A.I, O = A.O;
  const bool live = 2 * lane < I;
  const int fo = min(2 * lane, I - 2);
  // wave w multiplies 16-row block (w & 7) of A^T by the 16-column blocks 4 (w >> 3) .. + 3 of g
  const int ib = wave & 7, ob0 = (wave >> 3) * 4;
  const int IB = (I + 15) >> 4, OB = (O + 15) >> 4;
  f4v acc[4];
#pragma unroll
  for (int y = 0; y < 4; ++y) acc[y] = f4v{0.f, 0.f, 0.f, 0.f};
  const int cell_a = u0 / A.n_tiles, cell_b = (u1 - 1) / A.n_tiles;
  for (int k = cell_a; k <= cell_b; ++k) {
  const int seg0 = max(u0, k * A.n_tiles) - k * A.n_tiles, seg1 = min(u1, (k + 1) * A.n_tiles) - k * A.n_tiles;
  for (int t0 = seg0; t0 < seg1; t0 += WG_MAXT) {           // the tiles of this cell in this range, <= 64 at a time
  const int nt = min(WG_MAXT, seg1 - t0);
  __syncthreads();                                         // the previous segment's readers of t_pref / a_s / g_s are done
  if (wave == 0) {
    const int t = t0 + lane;
    int cnt = 0, row0 = 0;
    long long e8 = 0;
    if (lane < nt && A.tile_nsteps[t] >= 0) {              // -1: a tile the pair kernel refused (see there)
      const int2 d = A.desc[(size_t)t * A.n_cells + k];
      row0 = d.x; cnt = d.y;
      e8 = (long long)8 * A.rowptr[t * TN];
    }
    const int incl = wave_incl_scan(cnt, lane);
    t_pref[lane + 1] = incl;
    if (lane == 0) t_pref[0] = 0;
    t_row0[lane] = row0; t_e8[lane] = e8;
  }
  __syncthreads();
  const int R = t_pref[nt], nsteps = (R + 15) >> 4;

  // One row per wave: three dependent levels of memory accesses (row record -> pair records -> feature rows),
  // SOFTWARE-PIPELINED across steps: while step s multiplies, the wave has the feature rows of step s + 1, the pair
  // records of step s + 2 and the row record of step s + 3 in flight together, so a step exposes one round trip, not
  // three (with the three levels back to back per step the launch was latency-bound at 5-10 us per step whatever the
  // in-flight depth inside a level: 174-350 us in five successive forms). The pair records of a row are fetched
  // LANE-PARALLEL (lane i = the row's i-th pair: one coalesced load per array and 64 pairs), the feature rows with the
  // source index broadcast out of its lane (v_readlane): the first 8 at once, then 24, then 32 per trip -- every trip
  // inside one 64-record chunk, pairs summed in list order.
  struct S1 { int2 rec; int nexty, tile; long long e8; bool ok; };
  struct S2 { int p0, np, sr; float wr, g0, g1; const int2* pr; };
  auto stage1 = [&](int s) {
    S1 x;
    x.rec = make_int2(0, 0); x.nexty = 0; x.tile = 0; x.e8 = 0; x.ok = false;
    const int q = 16 * s + wave;
    if (s < nsteps && q < R) {                             // wave-uniform
      int j = 0;
      while (j + 1 < nt && t_pref[j + 1] <= q) ++j;        // the tile of row q (LDS broadcasts)
      j = UNI(j);
      x.tile = t0 + j; x.e8 = t_e8[j];
      const int2* rw = A.rows + x.e8 + x.tile + t_row0[j] + (q - t_pref[j]);
      x.rec = rw[0]; x.nexty = rw[1].y;
      x.ok = true;
    }
    return x;
  };
  auto stage2 = [&](const S1& x) {
    S2 y;
    y.p0 = x.ok ? UNI(x.rec.y) : 0;
    y.np = x.ok ? UNI(x.nexty - x.rec.y) : 0;
    const int node = x.ok ? UNI(x.tile * TN + x.rec.x) : 0;
    y.pr = A.pair + x.e8;
    y.sr = 0; y.wr = 0.f; y.g0 = 0.f; y.g1 = 0.f;
    if (y.np > 0) {                                        // wave-uniform (a row of the lists always holds a pair)
      const int at0 = y.p0 + min(lane, y.np - 1);          // past the row's end: its last pair again, weight 0
      const int2 rec = y.pr[at0];
      y.sr = rec.x;
      y.wr = lane < y.np ? __int_as_float(rec.y) : 0.f;
      const float* gr = A.g + (size_t)node * A.ldg;
      y.g0 = lane < O ? gr[lane] : 0.f;
      y.g1 = lane + 64 < O ? gr[lane + 64] : 0.f;
    }
    return y;
  };
  auto stage3 = [&](S2& y, int buf) {
    const int r = wave, np = y.np, p0 = y.p0;
    f2 acc = {0.f, 0.f};
    if (np > 0) {
      {
        f2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = *reinterpret_cast<const f2*>(A.feat + (size_t)__builtin_amdgcn_readlane(y.sr, u) * A.ldf + fo);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y.wr), u));
          acc = __builtin_elementwise_fma(f2{w, w}, v[u], acc);
        }
      }
      for (int base = 8; base < np;) {                     // wave-uniform: long rows (hubs)
        const int next = base < 32 ? 32 : base + 32, len = next - base;
        if ((base & 63) == 0) {                            // the next 64 records
          const int at = p0 + min(base + lane, np - 1);
          const int2 rec = y.pr[at];
          y.sr = rec.x;
          y.wr = base + lane < np ? __int_as_float(rec.y) : 0.f;
        }
        f2 t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u)
          t[u] = *reinterpret_cast<const f2*>(A.feat + (size_t)__builtin_amdgcn_readlane(y.sr, min((base & 63) + u, 63)) * A.ldf + fo);
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y.wr), min((base & 63) + u, 63)));
          w = u < len ? w : 0.f;                           // (a 24-pair trip: the lanes of the next trip carry live weights)
          acc = __builtin_elementwise_fma(f2{w, w}, t[u], acc);
        }
        base = next;
      }
    }
    *reinterpret_cast<f2*>(&a_s[buf][r][2 * lane]) = (live && np > 0) ? acc : f2{0.f, 0.f};      // padding rows: exact zeros
    g_s[buf][r][lane] = y.g0; g_s[buf][r][lane + 64] = y.g1;
  };

  S2 y2;                                                   // the pair records of the step after the one in LDS
  S1 y1;                                                   // the row record of the step after that
  {
    S1 x1 = stage1(0);
    S2 x2 = stage2(x1);
    x1 = stage1(1);
    y1 = stage1(2);
    y2 = stage2(x1);
    stage3(x2, 0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) {
      const S1 z1 = stage1(s + 3);
      S2 z2 = stage2(y1);
      stage3(y2, buf ^ 1);
      y2 = z2; y1 = z1;
    }
    if (ib < IB && ob0 < OB) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kr = kk * 4 + (lane >> 4), c = lane & 15;
        const float av = a_s[buf][kr][ib * 16 + c];
        float bv[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) bv[y] = g_s[buf][kr][min(ob0 + y, 7) * 16 + c];
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (ob0 + y < OB)                                // wave-uniform
            acc[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[y], acc[y], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  }   // tiles of the cell, 64 at a time
  // the cell is complete as far as this range goes: its block of the result to slot (range + cell) -- ranges and cells
  // both run in sequence order, so the numbering is unique and independent of timing
  float* dst = A.partial + (size_t)(range + k) * (size_t)I * O;
  if (ib < IB) {
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (ob0 + y >= OB) continue;
      const int o = (ob0 + y) * 16 + (lane & 15);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = ib * 16 + 4 * (lane >> 4) + v;
        if (i < I && o < O) dst[(size_t)i * O + o] = acc[y][v];
      }
      acc[y] = f4v{0.f, 0.f, 0.f, 0.f};
    }
  }
  }   // cells of the range
}

// dfilters[cell][e] = the slots (range + cell) of the ranges that visited the cell, in range order. grid (cells, chunks)
__global__ __launch_bounds__(256) void contconv_wgrad_finish_kernel(const float* __restrict__ partial, const int* __restrict__ ucut,
                                                                    int n_tiles, int io, float* __restrict__ out) {
  __shared__ int s_r0, s_r1;
  const int k = blockIdx.x;
  if (threadIdx.x == 0) {
    int r0 = -1, r1 = -2;
    if (ucut) {
      const int lo_u = k * n_tiles, hi_u = (k + 1) * n_tiles;
      int a = 0, b = WG_RANGES - 1;                          // last range starting at or before the cell's first unit
      while (a < b) { const int m = (a + b + 1) >> 1; if (ucut[m] <= lo_u) a = m; else b = m - 1; }
      r0 = a; r1 = a;
      while (r1 + 1 < WG_RANGES && ucut[r1 + 1] < hi_u) ++r1;
    }
    s_r0 = r0; s_r1 = r1;
  }
  __syncthreads();
  const int r0 = s_r0, r1 = s_r1, lo_u = k * n_tiles, hi_u = (k + 1) * n_tiles;
  for (int e = blockIdx.y * 256 + threadIdx.x; e < io; e += gridDim.y * 256) {
    float v = 0.f;
    for (int r = r0; r <= r1; ++r) {
      const int a = ucut[r], b = ucut[r + 1];
      if (a >= b || b <= lo_u || a >= hi_u) continue;        // empty, or not in this cell after all
      v += partial[(size_t)(r + k) * io + e];
    }
    out[(size_t)k * io + e] = v;
  }
}

// dfilters over the FULL grid: cell c of the D^3 grid takes the sum of the slabs of its compact cell cell_map[c], or an
// exact zero when no sample can reach it (cell_map[c] < 0) -- the (D, D, D, I, O) gradient tensor in one launch
// (torch.zeros + index_copy_ before)
__global__ __launch_bounds__(256) void contconv_wgrad_finish_full_kernel(const float* __restrict__ partial, int slabs, int n_cells,
                                                                         int io, const int* __restrict__ cell_map, int d3,
                                                                         float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)d3 * io) return;
  const int c = (int)(e / io), r = (int)(e - (size_t)c * io);
  const int k = cell_map ? cell_map[c] : c;
  float v = 0.f;
  if (k >= 0) {
    const size_t per_slab = (size_t)n_cells * io, at = (size_t)k * io + r;
    for (int s = 0; s < slabs; ++s) v += partial[(size_t)s * per_slab + at];
  }
  out[e] = v;
}

// K slabs of 32 the stream kernel runs for `in` input channels (its template parameter; the fragment is padded to it).
// Always 4 (K = 128, zeros beyond I): the kernel is written for 1 and 2 as well, but with the smaller fragments hipcc
// keeps the quads in different registers in the loop of a cell's steps and in the reloading step and copies them in
// between -- with loads in flight (tools/check_contconv_isa.py refuses those builds). Narrow layers multiply zeros: their
// matrix work is small either way.
__host__ __device__ inline int cc_slabs(int in) { (void)in; return 4; }

// filters (cells_total, I, O) -> the fused kernel's MFMA fragment order over the kept cells (include/nbd.h), split into
// three bf16 terms (split_bf16x3): 16-byte quad index (((cell * CB + cb) * NS + s) * 3 + u) * 64 + lane holds bf16 term
// u (0 lo, 1 mid, 2 hi: the order the consumers use them in) of F[kept[cell]][32 s + 8 (lane >> 4) + e][16 cb + (lane & 15)], e = 0 .. 7; zero beyond I / O; transposed = 1
// re-lays F^T (in / out swapped: the feature gradient's operand). One thread per (cell, cb, s, lane): eight strided reads,
// three 16-byte stores. Once per weight update.
__global__ __launch_bounds__(256) void contconv_shuffle_kernel(const float* __restrict__ f, const int64_t* __restrict__ kept,
                                                               int n_cells, int I, int O, int transposed, uint4* __restrict__ out) {
  const int Ii = transposed ? O : I, Oo = transposed ? I : O;           // the operand's own in / out
  const int NS = cc_slabs(Ii), CB = (Oo + 15) >> 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n_cells * CB * NS * 64) return;
  const int lane = (int)(idx & 63);
  size_t rest = idx >> 6;
  const int sl = (int)(rest % NS); rest /= NS;
  const int cb = (int)(rest % CB);
  const int cell = (int)(rest / CB);
  const float* src = f + (size_t)kept[cell] * I * O;
  const int col = 16 * cb + (lane & 15);
  unsigned t[3][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f2 x = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = 32 * sl + 8 * (lane >> 4) + e + j;
      if (kk < Ii && col < Oo) x[j] = transposed ? src[(size_t)col * O + kk] : src[(size_t)kk * O + col];
    }
    split_bf16x3(x, t[0][e >> 1], t[1][e >> 1], t[2][e >> 1]);
  }
#pragma unroll
  for (int term = 0; term < 3; ++term)
    out[((((size_t)cell * CB + cb) * NS + sl) * 3 + (2 - term)) * 64 + lane] = make_uint4(t[term][0], t[term][1], t[term][2], t[term][3]);
}

constexpr size_t WG_HEADER_BYTES = 4096;      // ucut[WG_RANGES + 1] in front of the partial slots
static_assert((WG_RANGES + 1) * sizeof(int) <= WG_HEADER_BYTES, "ucut header");
}  // namespace

extern "C" {

#ifdef NBD_CC_TRACE
int nbd_debug_cc_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cc_trace), &buf, sizeof(buf)); }
int nbd_debug_cc_timeline(void* buf, int workgroup) {       // buf: 16 waves x 512 steps x 4 int64
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_cc_tl), &buf, sizeof(buf));
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cc_tl_wg), &workgroup, sizeof(workgroup));
}
#endif

#ifdef NBD_PAIRS_TRACE
int nbd_debug_pairs_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pairs_trace), &buf, sizeof(buf)); }
#endif

int nbd_contconv_fused_supported(int in_channels, int out_channels, int n_cells) {
  return in_channels > 0 && in_channels % 4 == 0 && in_channels <= 128 && out_channels > 0 && n_cells > 0 &&
         n_cells <= MAXC;
}

namespace {
struct PairsLayout { size_t desc, rows, src, w, steps, nsteps, cost, scale, cuts, tbase, cbase, gcell, cellstep, total; };
// cell groups of a layer with n_cells kept cells (a divisor of CC_GRID): one per XCD when the fragments of all cells are
// well past an XCD's 4 MiB L2 (12 KiB per cell and 16-column block: 15.7 MB at D = 6, 128 -> 128), else one for all -- at
// D = 4 (64 cells, 6 MB) the groups' extra accumulator flushes (a workgroup then crosses ~5 tiles instead of 1.5) cost more
// than the fragment's L2 misses: same box, 0.252 -> 0.276 ms, against 0.385 -> 0.367 ms at D = 6
// (D = 4 with two groups -- the halves the workgroups' parity used to split a tile into anyway, now aligned with the XCDs:
// 0.248 -> 0.244 ms; four groups 0.266)
inline int cc_groups(int n_cells) { return n_cells >= 96 ? NBD_CC_GROUPS : n_cells >= 32 ? 2 : 1; }
PairsLayout pairs_layout(int n, int64_t edge_capacity, int n_cells) {
  const size_t tiles = (size_t)ceil_div(n, TN);
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  PairsLayout L;
  size_t at = 0;
  L.desc = at; at += up(tiles * n_cells * sizeof(int2));
  L.rows = at; at += up(((size_t)8 * edge_capacity + tiles) * sizeof(int2));
  L.src = at; at += up((size_t)8 * edge_capacity * sizeof(int2));      // {source, weight} records
  L.w = at;                                                             // (no separate weight array any more)
  L.steps = at; at += up((step_base((int)tiles, (int)edge_capacity, n_cells) + 2) * sizeof(int4));
  L.nsteps = at; at += up(tiles * sizeof(int));
  L.cost = at; at += up(tiles * sizeof(int));
  L.scale = at; at += up((size_t)n * sizeof(float));
  L.cuts = at; at += up((CC_GRID + NBD_CC_GROUPS) * sizeof(int));          // the plan (contconv_plan_kernel)
  L.tbase = at; at += up(NBD_CC_GROUPS * (tiles + 1) * sizeof(int));
  L.cbase = at; at += up(NBD_CC_GROUPS * (tiles + 1) * sizeof(int));
  L.gcell = at; at += up((NBD_CC_GROUPS + 1) * sizeof(int));
  L.cellstep = at; at += up(tiles * (size_t)(n_cells + 1) * sizeof(int2));
  L.total = at + 256;
  return L;
}
}  // namespace

size_t nbd_contconv_pairs_bytes(int n, int64_t edge_capacity, int n_cells) {
  if (n <= 0 || edge_capacity < 0 || n_cells <= 0) return 0;
  return pairs_layout(n, edge_capacity, n_cells).total;
}

int nbd_contconv_pairs_layout(int n, int64_t edge_capacity, int n_cells, size_t* offsets) {
  if (n <= 0 || edge_capacity < 0 || n_cells <= 0 || !offsets) return NBD_E_BADARG;
  const PairsLayout L = pairs_layout(n, edge_capacity, n_cells);
  offsets[0] = L.desc; offsets[1] = L.rows; offsets[2] = L.src; offsets[3] = L.w; offsets[4] = L.steps;
  offsets[5] = L.nsteps; offsets[6] = L.cost; offsets[7] = L.total; offsets[8] = L.scale;
  return 0;
}

int nbd_contconv_pairs_jobs_f32(const float* pos, int n, float radius_sq, int n_jobs, const nbd_cc_pairs_job* jd,
                                nbd_stream_t stream) {
  if (n < 0 || n_jobs < 1 || n_jobs > NBD_CC_MAX_RES || !jd) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos) return NBD_E_BADARG;
  PairJobs jobs;
  PlanJobs plans;
  int kc_max = 0;
  for (int r = 0; r < n_jobs; ++r) {
    const nbd_cc_pairs_job& q = jd[r];
    const int d = q.filter_resolution, nc = q.n_cells;
    if (!q.rowptr || !q.centres || q.edge_capacity < 0) return NBD_E_BADARG;
    if (q.edge_capacity > (int64_t)0x0fffffff) return NBD_E_BADARG;     // 8 x edges indexes the pair arrays as int
    if (d < 2 || d > 6 || nc <= 0 || nc > MAXC || nc > d * d * d) return NBD_E_BADARG;
    if (!q.cell_map && nc != d * d * d) return NBD_E_BADARG;
    if (!q.pair_lists || (reinterpret_cast<uintptr_t>(q.pair_lists) & 15) != 0) return NBD_E_BADARG;
    if (q.pair_lists_bytes < nbd_contconv_pairs_bytes(n, q.edge_capacity, nc)) return NBD_E_WORKSPACE;
    PairJob& j = jobs.j[r];
    j.D = d; j.n_cells = nc; j.cell_map = q.cell_map;
    j.rowptr = q.rowptr; j.centres = q.centres; j.deg = q.deg; j.sign = q.adjoint ? -1.0f : 1.0f;
    const PairsLayout L = pairs_layout(n, q.edge_capacity, nc);
    char* base = static_cast<char*>(q.pair_lists);
    j.desc = reinterpret_cast<int2*>(base + L.desc); j.rows = reinterpret_cast<int2*>(base + L.rows);
    j.pair = reinterpret_cast<int2*>(base + L.src);
    j.steps = reinterpret_cast<int4*>(base + L.steps); j.tile_nsteps = reinterpret_cast<int*>(base + L.nsteps);
    j.tile_cost = reinterpret_cast<int*>(base + L.cost);
    j.cellstep = reinterpret_cast<int2*>(base + L.cellstep);
    j.inv_deg = reinterpret_cast<float*>(base + L.scale);
    PlanJob& pj = plans.j[r];
    pj.tile_nsteps = j.tile_nsteps; pj.cellstep = j.cellstep; pj.steps = j.steps; pj.rowptr = q.rowptr; pj.n_cells = nc;
    pj.groups = cc_groups(nc); pj.gcell = reinterpret_cast<int*>(base + L.gcell);
    pj.cuts = reinterpret_cast<int*>(base + L.cuts); pj.tile_base = reinterpret_cast<int*>(base + L.tbase);
    pj.cost_base = reinterpret_cast<int*>(base + L.cbase);
    const int kc = (nc + 3) & ~3;
    if (kc > kc_max) kc_max = kc;
  }
  for (int r = n_jobs; r < NBD_CC_MAX_RES; ++r) { jobs.j[r] = jobs.j[0]; plans.j[r] = plans.j[0]; }
  const size_t lds = (size_t)TN * kc_max / 2 * 4 + (size_t)TN * kc_max * 4 + (size_t)TN * kc_max;
  {   // > 64 KiB of dynamic LDS needs the opt-in (a per-function attribute, idempotent)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_pairs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  contconv_pairs_kernel<<<dim3(ceil_div(n, TN), n_jobs), PAIR_THREADS, lds, (hipStream_t)stream>>>(pos, n, radius_sq, jobs);
  int rc = status();
  if (rc) return rc;
  contconv_plan_kernel<<<n_jobs, 1024, 0, (hipStream_t)stream>>>(plans, ceil_div(n, TN));
  return status();
}

int nbd_contconv_pairs_batch_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                                 float radius_sq, int n_res, const int* filter_resolutions, const int* const* cell_maps,
                                 const int* n_cells, void* const* pair_lists, const size_t* pair_lists_bytes,
                                 nbd_stream_t stream) {
  if (n < 0 || edge_capacity < 0 || n_res < 1 || n_res > NBD_CC_MAX_RES) return NBD_E_BADARG;
  if (!filter_resolutions || !cell_maps || !n_cells || !pair_lists || !pair_lists_bytes) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !rowptr || !centres) return NBD_E_BADARG;
  nbd_cc_pairs_job jd[NBD_CC_MAX_RES];
  for (int r = 0; r < n_res; ++r) {
    jd[r].rowptr = rowptr; jd[r].centres = centres; jd[r].deg = nullptr; jd[r].edge_capacity = edge_capacity;
    jd[r].filter_resolution = filter_resolutions[r]; jd[r].cell_map = cell_maps[r]; jd[r].n_cells = n_cells[r];
    jd[r].adjoint = 0; jd[r].pair_lists = pair_lists[r]; jd[r].pair_lists_bytes = pair_lists_bytes[r];
  }
  return nbd_contconv_pairs_jobs_f32(pos, n, radius_sq, n_res, jd, stream);
}

int nbd_contconv_pairs_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                           int filter_resolution, float radius_sq, const int* cell_map, int n_cells,
                           void* pair_lists, size_t pair_lists_bytes, nbd_stream_t stream) {
  const int* maps[1] = {cell_map};
  void* lists[1] = {pair_lists};
  return nbd_contconv_pairs_batch_f32(pos, rowptr, centres, n, edge_capacity, radius_sq, 1, &filter_resolution, maps,
                                      &n_cells, lists, &pair_lists_bytes, stream);
}

// [4 KiB header, unused][partial slots: per cell group (its workgroups + tiles) = CC_GRID + groups x tiles in all, each TN
// rows x (column groups x 128) floats]
constexpr size_t CC_CUTS_BYTES = 4096;
static_assert((CC_GRID + 1) * sizeof(int) <= CC_CUTS_BYTES, "cuts header");
size_t nbd_contconv_fused_workspace_bytes(int n, int n_cells, int out_channels) {
  if (n <= 0 || n_cells <= 0 || out_channels <= 0) return 0;
  return CC_CUTS_BYTES + (size_t)(CC_GRID + (size_t)cc_groups(n_cells) * ceil_div(n, TN)) * TN * (size_t)(ceil_div(out_channels, 128) * 128) * sizeof(float);
}

size_t nbd_contconv_filter_floats(int in_channels, int out_channels, int n_cells) {
  if (in_channels <= 0 || out_channels <= 0 || n_cells <= 0) return 0;
  return (size_t)n_cells * ceil_div(out_channels, 16) * 3 * cc_slabs(in_channels) * 64 * 4;    // 16-byte quads of 8 bf16, in floats
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
  if ((size_t)n * ldf * 4 > (size_t)0x7fffffff) return NBD_E_BADARG;       // a row's byte offset is a 32-bit buffer offset
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
      workspace_bytes < nbd_contconv_fused_workspace_bytes(n, n_cells, out_channels))
    return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const PairsLayout L = pairs_layout(n, edge_capacity, n_cells);
  const char* base = static_cast<const char*>(pair_lists);
  const int2* rows = reinterpret_cast<const int2*>(base + L.rows);
  const int2* pair = reinterpret_cast<const int2*>(base + L.src);
  const int4* steps = reinterpret_cast<const int4*>(base + L.steps);
  const int* tile_nsteps = reinterpret_cast<const int*>(base + L.nsteps);
  const int* tile_cost = reinterpret_cast<const int*>(base + L.cost);
  const int tiles = ceil_div(n, TN), colgroups = ceil_div(out_channels, 128), OP = colgroups * 128;
  const int* cuts = reinterpret_cast<const int*>(base + L.cuts);        // (the workspace keeps its 4 KiB header: unused now)
  const int* tile_base = reinterpret_cast<const int*>(base + L.tbase);
  float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + CC_CUTS_BYTES);
  const dim3 grid(CC_GRID, 1, colgroups);
  if (nbd_contconv_filter_floats(in_channels, out_channels, n_cells) * 4 > (size_t)0x7fffffff) return NBD_E_BADARG;   // 32-bit fragment offsets
  CCArgs A;
  A.feat = feat; A.ldf = ldf; A.I = in_channels; A.rowptr = rowptr; A.n = n; A.n_tiles = tiles;
  A.rows = rows; A.pair = pair; A.steps = steps; A.tile_nsteps = tile_nsteps; A.tile_cost = tile_cost; A.cuts = cuts; A.tile_base = tile_base;
  A.gcell = reinterpret_cast<const int*>(base + L.gcell); A.cellstep = reinterpret_cast<const int2*>(base + L.cellstep);
  A.groups = cc_groups(n_cells);
  A.filt = reinterpret_cast<const uint4*>(filters_shuffled); A.n_cells = n_cells;
  A.colblocks = ceil_div(out_channels, 16); A.OP = OP; A.partial = partial;
#define CC_LAUNCH(K)                                                                                                \
  do {                                                                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_stream_kernel<K>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);                     \
    if (e != hipSuccess) return (int)e;                                                                             \
    contconv_stream_kernel<K><<<grid, CC_THREADS, CC_LDS_BYTES, st>>>(A);                                           \
  } while (0)
  CC_LAUNCH(4);                                            // cc_slabs(): one K depth, I <= 128
#undef CC_LAUNCH
  int rc = status();
  if (rc) return rc;
  contconv_stream_finish_kernel<<<dim3(tiles, TN / FIN_ROWS), 256, 0, st>>>(partial, tile_nsteps, cuts, tile_base, tiles, cc_groups(n_cells), rowscale, act, out,
                                                                 ldo, n, out_channels, OP);
  return status();
}

size_t nbd_contconv_filter_grad_workspace_bytes(int n, int n_cells, int in_channels, int out_channels) {
  if (n <= 0 || n_cells <= 0 || in_channels <= 0 || out_channels <= 0) return 0;
  return WG_HEADER_BYTES + (size_t)(WG_RANGES + n_cells) * in_channels * out_channels * sizeof(float);
}

int nbd_contconv_filter_grad_f32(const float* feat, int ldf, int in_channels, const float* g, int ldg, int out_channels,
                                 const int* rowptr, int n, int64_t edge_capacity, const void* pair_lists, int n_cells,
                                 float* dfilters, void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || in_channels <= 0 || in_channels % 2 != 0 || in_channels > 128 || out_channels <= 0 || out_channels > 128 ||
      n_cells <= 0 || n_cells > MAXC || ldf < in_channels || (ldf & 1) || ldg < out_channels)
    return NBD_E_BADARG;
  if (!dfilters) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int io = in_channels * out_channels;
  const dim3 fgrid(n_cells, ceil_div(io, 1024) < 16 ? ceil_div(io, 1024) : 16);
  if (n == 0) {        // no rows: the gradient is zero
    contconv_wgrad_finish_kernel<<<fgrid, 256, 0, st>>>(nullptr, nullptr, 0, io, dfilters);
    return status();
  }
  if (!feat || !g || !rowptr || !pair_lists || (reinterpret_cast<uintptr_t>(feat) & 7)) return NBD_E_BADARG;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
      workspace_bytes < nbd_contconv_filter_grad_workspace_bytes(n, n_cells, in_channels, out_channels))
    return NBD_E_WORKSPACE;
  const PairsLayout L = pairs_layout(n, edge_capacity, n_cells);
  const char* base = static_cast<const char*>(pair_lists);
  WGArgs A;
  A.feat = feat; A.ldf = ldf; A.I = in_channels; A.g = g; A.ldg = ldg; A.O = out_channels;
  A.rowptr = rowptr; A.n = n; A.n_tiles = ceil_div(n, TN); A.n_cells = n_cells;
  A.desc = reinterpret_cast<const int2*>(base + L.desc); A.rows = reinterpret_cast<const int2*>(base + L.rows);
  A.pair = reinterpret_cast<const int2*>(base + L.src);
  A.tile_nsteps = reinterpret_cast<const int*>(base + L.nsteps);
  A.ucut = static_cast<int*>(workspace);
  A.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + WG_HEADER_BYTES);
  contconv_wplan_kernel<<<1, 1024, 0, st>>>(A);
  int rc = status();
  if (rc) return rc;
  contconv_wgrad_kernel<<<WG_RANGES, 64 * WG_WAVES, 0, st>>>(A);
  rc = status();
  if (rc) return rc;
  contconv_wgrad_finish_kernel<<<fgrid, 256, 0, st>>>(A.partial, A.ucut, A.n_tiles, io, dfilters);
  return status();
}

int nbd_contconv_shuffle_filters_f32(const float* filters, const int64_t* kept_cells, int n_cells, int in_channels,
                                     int out_channels, int transposed, float* filters_shuffled, nbd_stream_t stream) {
  if (n_cells <= 0 || in_channels <= 0 || out_channels <= 0 || !filters || !kept_cells || !filters_shuffled) return NBD_E_BADARG;
  if (reinterpret_cast<uintptr_t>(filters_shuffled) & 15) return NBD_E_BADARG;
  const int Ii = transposed ? out_channels : in_channels, Oo = transposed ? in_channels : out_channels;
  const size_t threads = (size_t)n_cells * ceil_div(Oo, 16) * cc_slabs(Ii) * 64;
  contconv_shuffle_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      filters, kept_cells, n_cells, in_channels, out_channels, transposed ? 1 : 0, reinterpret_cast<uint4*>(filters_shuffled));
  return status();
}

int nbd_contconv_filter_grad_full_f32(const float* feat, int ldf, int in_channels, const float* g, int ldg, int out_channels,
                                      const int* rowptr, int n, int64_t edge_capacity, const void* pair_lists, int n_cells,
                                      const int* cell_map, int cells_total, float* dfilters_full, void* workspace,
                                      size_t workspace_bytes, nbd_stream_t stream) {
  if (cells_total < n_cells || (!cell_map && cells_total != n_cells) || !dfilters_full) return NBD_E_BADARG;
  const size_t compact = (size_t)n_cells * in_channels * out_channels * sizeof(float);
  const size_t inner = nbd_contconv_filter_grad_workspace_bytes(n, n_cells, in_channels, out_channels);
  // workspace: [compact gradient][the slab partials of nbd_contconv_filter_grad_f32]
  if (!workspace || workspace_bytes < compact + inner || (reinterpret_cast<uintptr_t>(workspace) & 15)) return NBD_E_WORKSPACE;
  float* dcompact = static_cast<float*>(workspace);
  int rc = nbd_contconv_filter_grad_f32(feat, ldf, in_channels, g, ldg, out_channels, rowptr, n, edge_capacity, pair_lists,
                                        n_cells, dcompact, static_cast<char*>(workspace) + compact, inner, stream);
  if (rc) return rc;
  const int io = in_channels * out_channels;
  const size_t total = (size_t)cells_total * io;
  contconv_wgrad_finish_full_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      dcompact, 1, n_cells, io, cell_map, cells_total, dfilters_full);
  return status();
}

}  // extern "C"

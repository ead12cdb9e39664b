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
//   nbd_contconv_pairs_f32   per tile of 128 nodes: every (edge, corner) pair {source, weight}, grouped by
//                            (cell, node) -- a counting sort held in LDS; the packed "rows" (distinct nodes) of
//                            each (tile, cell) and their pair ranges. ~24 B per pair of index data, once per
//                            filter resolution and graph.
//   nbd_contconv_fused_f32   per (tile, chunk of cells): producer waves gather the pairs' feature rows and
//                            sum them into packed A rows in LDS (32 rows per step, double-buffered);
//                            consumer waves multiply each step by the cell's I x O filter with fp32 MFMA
//                            (v_mfma_f32_32x32x2_f32; the filter fragment is held in registers, pre-shuffled
//                            by the host so that every lane loads it with one dwordx4 per 8 k) and scatter-add
//                            the 32 x 128 result into a 128-node x 128-column accumulator in LDS.
//                            Cell chunks of one tile are summed in fixed order by the finishing kernel
//                            (scale, activation). No float atomics across waves: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

constexpr int TN = NBD_CC_TILE;      // nodes per tile (128)
constexpr int SUB = 32;              // packed rows per MFMA step
constexpr int LDA = 132;             // A row stride in floats: 16-B aligned, conflict-free ds_read_b128 fragments
constexpr int MAXC = 256;            // filter cells (reachable) supported: D <= 6
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
  if (cx < 0 || cx >= D || cy < 0 || cy >= D || cz < 0 || cz >= D) return -1;
  const int cell = (cz * D + cy) * D + cx;                           // filters[z][y][x] (contconv.py:62-75)
  const float wxy = (ax ? g.tx : 1.0f - g.tx) * (ay ? g.ty : 1.0f - g.ty);
  *w = wxy * ((az ? g.tz : 1.0f - g.tz) * g.window);
  return cell_map ? cell_map[cell] : cell;
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
// One workgroup (8 waves) per tile of 128 nodes. LDS: cnt[node][cell] (u16 pairs of a (node, cell) block,
// packed two per word), pwithin[node][cell] (u32: pairs of the same cell in lower nodes of the tile),
// rowidx[node][cell] (u8: touched lower nodes of the same cell).
//   A  wave per node, lane per edge: geometry, 8 LDS counter increments
//   B  wave per cell, lanes over nodes: prefix sums down the tile -> rows and pairs per cell
//   B2 prefix over cells -> desc[tile][cell] = {first row, rows}
//   B3 rows[] = {node_local, first pair}
//   C  wave per node again: each pair takes the next slot of its (node, cell) block. A node belongs to one
//      wave and its edges are visited in CSR order, so slots are assigned in a fixed order.
// Global layout, per tile t with e_t = rowptr[128 t]: rows at 8 e_t + t (one sentinel row per tile),
// pairs at 8 e_t: an edge has at most 8 corners, so the bases need no scan across tiles.
__global__ __launch_bounds__(512) void contconv_pairs_kernel(
    const float* __restrict__ pos, const int* __restrict__ rowptr, const int* __restrict__ centres, int n, int D,
    float r2max, const int* __restrict__ cell_map, int n_cells, int2* __restrict__ desc, int2* __restrict__ rows,
    int2* __restrict__ pairs) {
  extern __shared__ unsigned smem[];
  const int kc = (n_cells + 3) & ~3;                       // padded cell count (even: two u16 per word)
  unsigned* cnt32 = smem;                                   // [TN][kc/2]
  unsigned* pwithin = cnt32 + TN * kc / 2;                  // [TN][kc]
  unsigned char* rowidx = reinterpret_cast<unsigned char*>(pwithin + TN * kc);   // [TN][kc]
  __shared__ int cell_rows[MAXC], cell_pairs[MAXC], cell_rowbase[MAXC], cell_pairbase[MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, n0 = tile * TN, n_here = min(TN, n - n0);
  const int e_t = rowptr[n0];
  const size_t row_base = (size_t)8 * e_t + tile, pair_base = (size_t)8 * e_t;
  const float half = (float)(D - 1) / 2.0f;

  for (int i = tid; i < TN * kc / 2; i += 512) cnt32[i] = 0;
  __syncthreads();

  // ---- A: counts
  for (int nl = wave; nl < n_here; nl += 8) {
    const int node = n0 + nl;
    const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
    const int e0 = rowptr[node], e1 = rowptr[node + 1];
    for (int e = e0 + lane; e < e1; e += 64) {
      const Geo g = edge_geo(pos, centres[e], xn, yn, zn, r2max, half);
      if (g.window == 0.f) continue;                       // outside the radius: the reference multiplies by 0
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        float w;
        const int k = corner_cell(g, corner, D, cell_map, &w);
        if (k >= 0) atomicAdd(&cnt32[(nl * kc + k) >> 1], 1u << (16 * (k & 1)));
      }
    }
  }
  __syncthreads();

  // ---- B: per cell, prefix over the nodes of the tile (lane holds nodes 2*lane and 2*lane + 1)
  const unsigned short* cnt16 = reinterpret_cast<const unsigned short*>(cnt32);
  for (int k = wave; k < n_cells; k += 8) {
    const int v0 = cnt16[(2 * lane) * kc + k], v1 = cnt16[(2 * lane + 1) * kc + k];
    const int r0 = v0 > 0, r1 = v1 > 0;
    const int pi = wave_incl_scan(v0 + v1, lane), ri = wave_incl_scan(r0 + r1, lane);
    const int pe = pi - (v0 + v1), re = ri - (r0 + r1);     // exclusive
    pwithin[(2 * lane) * kc + k] = pe;      rowidx[(2 * lane) * kc + k] = (unsigned char)re;
    pwithin[(2 * lane + 1) * kc + k] = pe + v0; rowidx[(2 * lane + 1) * kc + k] = (unsigned char)(re + r0);
    if (lane == 63) { cell_pairs[k] = pi; cell_rows[k] = ri; }
  }
  __syncthreads();

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

  // ---- B3: row records
  for (int i = tid; i < TN * n_cells; i += 512) {
    const int nl = i / n_cells, k = i - nl * n_cells;
    if (cnt16[nl * kc + k] > 0)
      rows[row_base + cell_rowbase[k] + rowidx[nl * kc + k]] = make_int2(nl, cell_pairbase[k] + (int)pwithin[nl * kc + k]);
  }

  __syncthreads();          // B3 reads the counters that C counts down

  // ---- C: place the pairs (counters count down: slot = old - 1)
  for (int nl = wave; nl < n_here; nl += 8) {
    const int node = n0 + nl;
    const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
    const int e0 = rowptr[node], e1 = rowptr[node + 1];
    for (int e = e0 + lane; e < e1; e += 64) {
      const int c = centres[e];
      const Geo g = edge_geo(pos, c, xn, yn, zn, r2max, half);
      if (g.window == 0.f) continue;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        float w;
        const int k = corner_cell(g, corner, D, cell_map, &w);
        if (k < 0) continue;
        const unsigned old = atomicSub(&cnt32[(nl * kc + k) >> 1], 1u << (16 * (k & 1)));
        const int slot = (int)((old >> (16 * (k & 1))) & 0xffffu) - 1;
        pairs[pair_base + cell_pairbase[k] + pwithin[nl * kc + k] + slot] = make_int2(c, __float_as_int(w));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- fused conv
// grid = (tiles, cell chunks, column groups of 128); block = 512 threads: waves 0-3 consume (MFMA, 32 output
// columns each), waves 4-7 produce (gather + sum of the packed A rows). One s_barrier per step.
struct Step { int cell, row_begin, n_rows; };

__global__ __launch_bounds__(512) void contconv_fused_kernel(
    const float* __restrict__ feat, int ldf, int I, const int* __restrict__ rowptr, int n,
    const int2* __restrict__ desc, const int2* __restrict__ rows, const int2* __restrict__ pairs,
    const f4* __restrict__ filt, int n_cells, int kq_count, int colblocks, int cells_per_chunk, int O,
    float* __restrict__ partial) {
  extern __shared__ float lds[];
  float* out_acc = lds;                                    // [TN][128]
  float* a_buf = out_acc + TN * 128;                       // [2][SUB][LDA]
  int* rowmap = reinterpret_cast<int*>(a_buf + 2 * SUB * LDA);   // [2][SUB]
  __shared__ int s_cell[CHUNK_MAX], s_rowbeg[CHUNK_MAX], s_nrows[CHUNK_MAX];
  __shared__ int s_ncell;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, n0 = tile * TN;
  const int k_begin = blockIdx.y * cells_per_chunk, k_end = min(n_cells, k_begin + cells_per_chunk);
  const int e_t = rowptr[n0];
  const int2* t_rows = rows + (size_t)8 * e_t + tile;
  const int2* t_pairs = pairs + (size_t)8 * e_t;

  // non-empty cells of this chunk, compacted (wave 0; cells_per_chunk <= 64)
  if (wave == 0) {
    const int k = k_begin + lane;
    int2 d = make_int2(0, 0);
    if (k < k_end) d = desc[(size_t)tile * n_cells + k];
    const unsigned long long m = __ballot(d.y > 0);
    if (d.y > 0) {
      const int j = __popcll(m & ((1ull << lane) - 1ull));
      s_cell[j] = k; s_rowbeg[j] = d.x; s_nrows[j] = d.y;
    }
    if (lane == 0) s_ncell = __popcll(m);
  }
  for (int i = tid; i < TN * 128 / 4; i += 512) reinterpret_cast<f4*>(out_acc)[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int ncell = s_ncell;

  const bool producer = wave >= 4;
  const int w4 = wave & 3;
  const int cb = blockIdx.z * 4 + w4;                      // consumer: 32-column block of the output
  const bool has_cols = cb < colblocks;

  // ---- producer: packed A rows of one step into a_buf[buf]
  auto produce = [&](int j, int sub, int buf) {
    const int nrows = s_nrows[j], rbeg = s_rowbeg[j];
    const int r_first = sub * SUB + 8 * w4;                // this wave's 8 rows of the step
    const int cnt = max(0, min(8, nrows - r_first));
    float* a_dst = a_buf + (buf * SUB + 8 * w4) * LDA;
    int2 rinfo = make_int2(-1, 0);
    if (lane <= cnt && cnt > 0) rinfo = t_rows[rbeg + r_first + lane];     // row `cnt` = the next row (or sentinel)
    if (lane < 8) rowmap[buf * SUB + 8 * w4 + lane] = lane < cnt ? rinfo.x : -1;
    if (cnt == 0) return;
    const int p_begin = __builtin_amdgcn_readlane(rinfo.y, 0), p_end = __builtin_amdgcn_readlane(rinfo.y, cnt);
    const bool live = 2 * lane < I;
    const float* f_lane = feat + 2 * lane;
    int cur = 0;                                           // row being accumulated
    int next_begin = __builtin_amdgcn_readlane(rinfo.y, 1);
    f2 acc = {0.f, 0.f};
    for (int base = p_begin; base < p_end; base += 64) {
      int2 pr = make_int2(0, 0);
      if (base + lane < p_end) pr = t_pairs[base + lane];
      const int here = min(64, p_end - base);
      for (int i0 = 0; i0 < here; i0 += 4) {
        f2 f[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                      // four feature rows in flight
          const int i = min(i0 + u, here - 1);
          const int c = __builtin_amdgcn_readlane(pr.x, i);
          w[u] = __int_as_float(__builtin_amdgcn_readlane(pr.y, i));
          f[u] = live ? *reinterpret_cast<const f2*>(f_lane + (size_t)c * ldf) : f2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u >= here) break;
          if (base + i0 + u == next_begin) {               // row boundary (wave-uniform)
            *reinterpret_cast<f2*>(a_dst + cur * LDA + 2 * lane) = acc;
            acc = f2{0.f, 0.f};
            ++cur;
            next_begin = __builtin_amdgcn_readlane(rinfo.y, min(cur + 1, 63));
          }
          acc.x = fmaf(w[u], f[u].x, acc.x);
          acc.y = fmaf(w[u], f[u].y, acc.y);
        }
      }
    }
    *reinterpret_cast<f2*>(a_dst + cur * LDA + 2 * lane) = acc;
  };

  // ---- consumer state
  f4 bfrag[16];
  auto load_b = [&](int cell) {
    const f4* src = filt + (((size_t)cell * colblocks + cb) * kq_count) * 64 + lane;
#pragma unroll
    for (int kq = 0; kq < 16; ++kq)
      if (kq < kq_count) bfrag[kq] = src[(size_t)kq * 64];
  };
  auto consume = [&](int buf) {
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a_base = a_buf + (buf * SUB + (lane & 31)) * LDA + (lane >> 5) * 4;
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) {
      if (kq < kq_count) {
        const f4 a = *reinterpret_cast<const f4*>(a_base + kq * 8);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], bfrag[kq][c], acc, 0, 0, 0);
      }
    }
    // scatter-add: C row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = lane & 31; packed row -> node of the tile.
    // Within a step the rows are distinct nodes and every column block belongs to one wave: plain
    // read-modify-write by the owner, no cross-wave race.
    const int* rm = rowmap + buf * SUB + 4 * (lane >> 5);
    float* o_col = out_acc + w4 * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int node = rm[(r & 3) + 8 * (r >> 2)];
      if (node >= 0) o_col[node * 128] += acc[r];
    }
  };

  // ---- the step pipeline: producers run one step ahead of the consumers
  int pj = 0, psub = 0;                                    // producer iterator (next step to fill)
  if (producer && ncell > 0) produce(0, 0, 0);
  auto advance = [&](int& j, int& sub) {
    if ((sub + 1) * SUB < s_nrows[j]) ++sub; else { ++j; sub = 0; }
  };
  if (ncell > 0) advance(pj, psub);
  __syncthreads();
  int buf = 0, cur_cell = -1;
  for (int j = 0, sub = 0; j < ncell; advance(j, sub)) {
    if (producer) {
      if (pj < ncell) produce(pj, psub, buf ^ 1);
    } else if (has_cols) {
      if (s_cell[j] != cur_cell) { cur_cell = s_cell[j]; load_b(cur_cell); }
      consume(buf);
    }
    if (pj < ncell) advance(pj, psub);
    __syncthreads();
    buf ^= 1;
  }

  // ---- write the tile's partial sums for this cell chunk: partial[chunk][node][column]
  float* dst = partial + ((size_t)blockIdx.y * n + n0) * O;
  const int col0 = blockIdx.z * 128;
  const int n_here = min(TN, n - n0), cols = min(128, O - col0);
  for (int i = tid; i < n_here * 128; i += 512) {
    const int nl = i >> 7, c = i & 127;
    if (c < cols) dst[(size_t)nl * O + col0 + c] = out_acc[nl * 128 + c];
  }
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
  // ~4 workgroups per CU over the launch so that tiles of different density balance; <= 64 cells per chunk
  int chunks = ceil_div(1024, p.tiles * p.colgroups);
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

int nbd_contconv_fused_supported(int in_channels, int out_channels, int n_cells) {
  return in_channels > 0 && in_channels % 4 == 0 && in_channels <= 128 && out_channels > 0 && n_cells > 0 &&
         n_cells <= MAXC;
}

size_t nbd_contconv_pairs_bytes(int n, int64_t edge_capacity, int n_cells) {
  if (n <= 0 || edge_capacity < 0 || n_cells <= 0) return 0;
  const size_t tiles = (size_t)ceil_div(n, TN);
  const size_t desc = tiles * n_cells * sizeof(int2);
  const size_t rows = ((size_t)8 * edge_capacity + tiles) * sizeof(int2);
  const size_t pairs = (size_t)8 * edge_capacity * sizeof(int2);
  return ((desc + 255) & ~(size_t)255) + ((rows + 255) & ~(size_t)255) + pairs + 256;
}

static void split_pairs_buffer(void* buf, int n, int64_t edge_capacity, int n_cells, int2** desc, int2** rows, int2** pairs) {
  const size_t tiles = (size_t)ceil_div(n, TN);
  char* p = static_cast<char*>(buf);
  *desc = reinterpret_cast<int2*>(p);
  p += (tiles * n_cells * sizeof(int2) + 255) & ~(size_t)255;
  *rows = reinterpret_cast<int2*>(p);
  p += (((size_t)8 * edge_capacity + tiles) * sizeof(int2) + 255) & ~(size_t)255;
  *pairs = reinterpret_cast<int2*>(p);
}

int nbd_contconv_pairs_f32(const float* pos, const int* rowptr, const int* centres, int n, int64_t edge_capacity,
                           int filter_resolution, float radius_sq, const int* cell_map, int n_cells,
                           void* pair_lists, size_t pair_lists_bytes, nbd_stream_t stream) {
  if (n < 0 || edge_capacity < 0 || filter_resolution < 2 || n_cells <= 0 || n_cells > MAXC) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !rowptr || !centres || !pair_lists) return NBD_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(pair_lists) & 15) != 0) return NBD_E_BADARG;
  if (pair_lists_bytes < nbd_contconv_pairs_bytes(n, edge_capacity, n_cells)) return NBD_E_WORKSPACE;
  if (!cell_map && n_cells != filter_resolution * filter_resolution * filter_resolution) return NBD_E_BADARG;
  int2 *desc, *rows, *pairs;
  split_pairs_buffer(pair_lists, n, edge_capacity, n_cells, &desc, &rows, &pairs);
  const int kc = (n_cells + 3) & ~3;
  const size_t lds = (size_t)TN * kc / 2 * 4 + (size_t)TN * kc * 4 + (size_t)TN * kc;
  {   // > 64 KiB of dynamic LDS needs the opt-in (a per-function attribute, idempotent)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_pairs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  contconv_pairs_kernel<<<ceil_div(n, TN), 512, lds, (hipStream_t)stream>>>(
      pos, rowptr, centres, n, filter_resolution, radius_sq, cell_map, n_cells, desc, rows, pairs);
  return status();
}

size_t nbd_contconv_fused_workspace_bytes(int n, int n_cells, int out_channels) {
  if (n <= 0 || n_cells <= 0 || out_channels <= 0) return 0;
  const FusedPlan p = plan_fused(n, n_cells, out_channels);
  return (size_t)p.chunks * n * out_channels * sizeof(float);
}

size_t nbd_contconv_filter_floats(int in_channels, int out_channels, int n_cells) {
  if (in_channels <= 0 || out_channels <= 0 || n_cells <= 0) return 0;
  return (size_t)n_cells * ceil_div(out_channels, 32) * ceil_div(in_channels, 8) * 64 * 4;
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
  int2 *desc, *rows, *pairs;
  split_pairs_buffer(const_cast<void*>(pair_lists), n, edge_capacity, n_cells, &desc, &rows, &pairs);
  const FusedPlan p = plan_fused(n, n_cells, out_channels);
  const size_t lds = (size_t)(TN * 128 + 2 * SUB * LDA) * sizeof(float) + 2 * SUB * sizeof(int);
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  float* partial = static_cast<float*>(workspace);
  contconv_fused_kernel<<<dim3(p.tiles, p.chunks, p.colgroups), 512, lds, st>>>(
      feat, ldf, in_channels, rowptr, n, desc, rows, pairs, reinterpret_cast<const f4*>(filters_shuffled), n_cells,
      ceil_div(in_channels, 8), ceil_div(out_channels, 32), p.cells_per_chunk, out_channels, partial);
  int rc = status();
  if (rc) return rc;
  const size_t total = (size_t)n * out_channels;
  contconv_finish_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(partial, p.chunks, rowscale, act, out, ldo, n,
                                                                         out_channels);
  return status();
}

}  // extern "C"

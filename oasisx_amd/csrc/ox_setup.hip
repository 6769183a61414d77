// Set-up of the IPCS path behind the C ABI: from a simplicial mesh (vertex coordinates + cell->vertex
// table) to everything the kernels of this library read -- kernel cell order and geometry, Lagrange
// P1/P2 dof numbering in SELL-64 row order, cell->dof tables, the SELL-64 sparsity pattern with its
// 16-bit column stream, the dof->cell adjacency with in-row position bytes, the width bins of the
// assembly launches, the rectangular (V x Q, Q x V) patterns and the 1-byte value dictionaries.
// It replaces what DOLFINx does for the reference behind functionspace(), create_matrix() /
// create_sparsity_pattern() and the dofmap (reference fracstep.py:187,293-300,315,324,336,352).
//
// Everything runs on the device (rocPRIM radix sorts / scans + a few kernels of our own); sizes go
// through size_t / int64 so patterns beyond 2^31 storage slots (256^3 P2) are built the same way.
// The numbering is the one DESIGN.md section 2 describes: dofs (and cells, by centroid) ordered by
// (tile_z, tile_y, z, y, x), then, inside windows of `window` rows, stably by decreasing row length.
#include <cstring>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ox_common.h"

namespace {

constexpr int KV = OX_KV;
constexpr int SLICE = OX_SLICE;
constexpr int ROW_CAP = 2048;  // candidate columns of one row (adjacent cells x dofs per cell) held in LDS

struct DevBuf {  // owned device allocation
  void *p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  int alloc(size_t b) {
    release();
    bytes = b;
    if (b == 0) return 0;
    if (hipMalloc(&p, b) != hipSuccess) {
      p = nullptr;
      snprintf(ox_err_buf, sizeof(ox_err_buf), "set-up: hipMalloc of %zu bytes failed", b);
      return -1;
    }
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
  void *detach() {
    void *q = p;
    p = nullptr;
    bytes = 0;
    return q;
  }
};

#define OX_TRY(expr)          \
  do {                        \
    if ((expr) != 0) return -1; \
  } while (0)

// ---- small device helpers -------------------------------------------------------------------
template <class K, class V>
int sort_pairs(K *keys_in, K *keys_out, V *vals_in, V *vals_out, size_t n, int end_bit, hipStream_t st) {
  if (n == 0) return 0;
  size_t tb = 0;
  OX_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
  DevBuf tmp;
  OX_TRY(tmp.alloc(tb));
  OX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

int exclusive_scan_i64(const int64_t *in, int64_t *out, size_t n, hipStream_t st) {
  if (n == 0) return 0;
  size_t tb = 0;
  OX_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), st));
  DevBuf tmp;
  OX_TRY(tmp.alloc(tb));
  OX_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

int bits_for(uint64_t maxval) {
  int b = 1;
  while (b < 64 && (maxval >> b)) ++b;
  return b;
}

// per-dimension min / max of an [n][d] point set: per-block partials, finished on the host
__global__ __launch_bounds__(256) void k_minmax(const double *__restrict__ x, int64_t n, int d, double *__restrict__ part) {
  __shared__ double lo_s[4][3], hi_s[4][3];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    for (int k = 0; k < d; ++k) {
      const double v = x[i * d + k];
      lo[k] = fmin(lo[k], v);
      hi[k] = fmax(hi[k], v);
    }
  for (int k = 0; k < 3; ++k)
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fmin(lo[k], __shfl_down(lo[k], off, 64));
      hi[k] = fmax(hi[k], __shfl_down(hi[k], off, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; ++k) lo_s[threadIdx.x >> 6][k] = lo[k], hi_s[threadIdx.x >> 6][k] = hi[k];
  __syncthreads();
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; ++k) {
      part[blockIdx.x * 6 + k] = fmin(fmin(lo_s[0][k], lo_s[1][k]), fmin(lo_s[2][k], lo_s[3][k]));
      part[blockIdx.x * 6 + 3 + k] = fmax(fmax(hi_s[0][k], hi_s[1][k]), fmax(hi_s[2][k], hi_s[3][k]));
    }
}

// ordering key of a point: (tile_z, tile_y, z, y, x) on a 2^bits lattice
struct KeySpec {
  double lo[3], inv[3];  // (x - lo) * inv in [0, 2^bits - 1]
  int d, bits, tb;
  int curve;  // 0: tiled lexicographic (lattice meshes), 1: Z-order curve (everything else)
  int bs;     // > 0 (tiled order only): bricks of 2^bs lattice steps a side -- (tile, brick_z, brick_y, brick_x, z, y, x)
};
__device__ __forceinline__ uint64_t locality_key(const double *p, const KeySpec &K) {
  uint64_t q[3] = {0, 0, 0};
  for (int k = 0; k < K.d; ++k) {
    double v = rint((p[k] - K.lo[k]) * K.inv[k]);
    const double top = (double)((1ull << K.bits) - 1);
    v = v < 0.0 ? 0.0 : (v > top ? top : v);
    q[k] = (uint64_t)v;
  }
  uint64_t key = 0;
  if (K.curve) {  // bit b of every coordinate, slowest coordinate first, from the top bit down
    for (int b = K.bits - 1; b >= 0; --b)
      for (int k = K.d - 1; k >= 0; --k) key = (key << 1) | ((q[k] >> b) & 1ull);
    return key;
  }
  for (int k = K.d - 1; k >= 1; --k) key = (key << K.tb) | (q[k] >> (K.bits - K.tb));
  if (K.bs > 0) {  // brick order: the rows of 8 consecutive slices then share a compact window of columns (k_spmv_win)
    for (int k = K.d - 1; k >= 0; --k) key = (key << (K.bits - K.bs)) | (q[k] >> K.bs);
    for (int k = K.d - 1; k >= 0; --k) key = (key << K.bs) | (q[k] & ((1ull << K.bs) - 1ull));
    return key;
  }
  for (int k = K.d - 1; k >= 0; --k) key = (key << K.bits) | q[k];
  return key;
}
__host__ __device__ inline int key_end_bit(const KeySpec &K) {
  return K.curve ? K.d * K.bits : (K.d - 1) * K.tb + K.d * K.bits;
}

// distinct values of one coordinate on a 2^20 lattice: bits set in a 128 KiB bitmap
constexpr int LATTICE_BITS = 20;
__global__ __launch_bounds__(256) void k_mark_coord(const double *__restrict__ x, int64_t n, int d, int k, double lo, double inv,
                                                    unsigned *__restrict__ bitmap) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = rint((x[i * d + k] - lo) * inv);
    const double top = (double)((1u << LATTICE_BITS) - 1);
    v = v < 0.0 ? 0.0 : (v > top ? top : v);
    const unsigned q = (unsigned)v;
    atomicOr(&bitmap[q >> 5], 1u << (q & 31));
  }
}

__global__ __launch_bounds__(256) void k_cell_keys(const double *__restrict__ coords, const int32_t *__restrict__ cells,
                                                   int64_t nc, KeySpec K, uint64_t *__restrict__ keys,
                                                   int32_t *__restrict__ ids) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  const int nv = K.d + 1;
  double cen[3] = {0, 0, 0};
  for (int a = 0; a < nv; ++a)
    for (int k = 0; k < K.d; ++k) cen[k] += coords[(int64_t)cells[c * nv + a] * K.d + k];
  for (int k = 0; k < K.d; ++k) cen[k] /= (double)nv;
  keys[c] = locality_key(cen, K);
  ids[c] = (int32_t)c;
}

__global__ __launch_bounds__(256) void k_point_keys(const double *__restrict__ x, int64_t n, KeySpec K,
                                                    uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = locality_key(x + i * K.d, K);
  ids[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_gather_cells(const int32_t *__restrict__ cells, const int32_t *__restrict__ perm,
                                                      int64_t nc, int nv, int32_t *__restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  for (int a = 0; a < nv; ++a) out[c * nv + a] = cells[(int64_t)perm[c] * nv + a];
}

// [n_cells][gs] rows of grad(lambda_1..d) then |detJ| (gs = 6 in 2-D, 10 in 3-D): closed-form inverses
__global__ __launch_bounds__(256) void k_geometry(const double *__restrict__ coords, const int32_t *__restrict__ cells,
                                                  int64_t nc, int d, double *__restrict__ geom) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  const int nv = d + 1, gs = d == 2 ? 6 : 10;
  double *g = geom + c * gs;
  const double *x0 = coords + (int64_t)cells[c * nv] * d;
  double e[3][3];  // rows = edge vectors x_a - x_0
  for (int a = 1; a < nv; ++a)
    for (int k = 0; k < d; ++k) e[a - 1][k] = coords[(int64_t)cells[c * nv + a] * d + k] - x0[k];
  if (d == 2) {
    const double a = e[0][0], b = e[1][0], cc = e[0][1], dd = e[1][1];  // J = [[a, b], [cc, dd]]
    const double det = a * dd - b * cc;
    g[0] = dd / det, g[1] = -b / det, g[2] = -cc / det, g[3] = a / det;
    g[4] = fabs(det);
    g[5] = 0.0;
  } else {
    const double *e1 = e[0], *e2 = e[1], *e3 = e[2];
    double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
    for (int k = 0; k < 3; ++k) g[k] = c23[k] / det, g[3 + k] = c31[k] / det, g[6 + k] = c12[k] / det;
    g[9] = fabs(det);
  }
}

// ---- P2: edges ------------------------------------------------------------------------------
__device__ __constant__ int EDGE2[3][2] = {{1, 2}, {0, 2}, {0, 1}};
__device__ __constant__ int EDGE3[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};

__global__ __launch_bounds__(256) void k_edge_keys(const int32_t *__restrict__ cells, int64_t nc, int d, int64_t nverts,
                                                   uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int ne = d == 2 ? 3 : 6;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nc * ne) return;
  const int64_t c = i / ne;
  const int le = (int)(i - c * ne);
  const int a = d == 2 ? EDGE2[le][0] : EDGE3[le][0], b = d == 2 ? EDGE2[le][1] : EDGE3[le][1];
  const int64_t va = cells[c * (d + 1) + a], vb = cells[c * (d + 1) + b];
  keys[i] = (uint64_t)(va < vb ? va : vb) * (uint64_t)nverts + (uint64_t)(va < vb ? vb : va);
  ids[i] = (int32_t)i;  // nc * ne < 2^31 is checked by the caller
}

// ---- P3 on tetrahedra: faces (one dof each), numbered by ascending sorted vertex triple ---------------------------
__device__ __constant__ int FACE3[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};  // face f = the vertices other than f
__global__ __launch_bounds__(256) void k_face_keys(const int32_t *__restrict__ cells, int64_t nc, int64_t nverts,
                                                   uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nc * 4) return;
  const int64_t c = i >> 2;
  const int f = (int)(i & 3);
  int64_t v0 = cells[c * 4 + FACE3[f][0]], v1 = cells[c * 4 + FACE3[f][1]], v2 = cells[c * 4 + FACE3[f][2]];
  if (v0 > v1) { const int64_t t = v0; v0 = v1; v1 = t; }
  if (v1 > v2) { const int64_t t = v1; v1 = v2; v2 = t; }
  if (v0 > v1) { const int64_t t = v0; v0 = v1; v1 = t; }
  keys[i] = ((uint64_t)v0 * (uint64_t)nverts + (uint64_t)v1) * (uint64_t)nverts + (uint64_t)v2;
  ids[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_heads(const uint64_t *__restrict__ ks, int64_t n, int64_t *__restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) head[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

// eid = exclusive scan of head shifted: id of sorted position i = (#heads at positions <= i) - 1
__global__ __launch_bounds__(256) void k_edge_scatter(const uint64_t *__restrict__ ks, const int32_t *__restrict__ pos,
                                                      const int64_t *__restrict__ head, const int64_t *__restrict__ excl,
                                                      int64_t n, int32_t *__restrict__ cell_edges,
                                                      uint64_t *__restrict__ edge_keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t id = excl[i] + head[i] - 1;
  cell_edges[pos[i]] = (int32_t)id;
  if (head[i]) edge_keys[id] = ks[i];
}

__global__ __launch_bounds__(256) void k_cd0(const int32_t *__restrict__ cells, const int32_t *__restrict__ cell_edges,
                                             int64_t nc, int d, int degree, int64_t nverts, int64_t n_edges,
                                             int32_t *__restrict__ cd0, const int32_t *__restrict__ cell_faces = nullptr,
                                             const int32_t *__restrict__ cell_perm = nullptr) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nc) return;
  const int nv = d + 1, ne = degree >= 2 ? (d == 2 ? 3 : 6) : 0;
  if (degree == 3 && d == 3) {
    // tetrahedra, 20 dofs: vertices; per local edge (a, b) the node nearer a, then the node nearer b (an edge's two dofs
    // are numbered from its LOWER global vertex: nverts + 2 e + {0, 1}); the dofs of the four faces (face f = the
    // vertices other than f): nverts + 2 n_edges + face id
    const int nd = 20;
    for (int a = 0; a < nv; ++a) cd0[c * nd + a] = cells[c * nv + a];
    for (int e = 0; e < ne; ++e) {
      const int flip = cells[c * nv + EDGE3[e][0]] > cells[c * nv + EDGE3[e][1]] ? 1 : 0;
      const int64_t first = nverts + 2 * (int64_t)cell_edges[c * ne + e];
      cd0[c * nd + nv + 2 * e] = (int32_t)(first + flip);
      cd0[c * nd + nv + 2 * e + 1] = (int32_t)(first + 1 - flip);
    }
    for (int f = 0; f < 4; ++f) cd0[c * nd + 16 + f] = (int32_t)(nverts + 2 * n_edges + cell_faces[c * 4 + f]);
    return;
  }
  if (degree == 3) {
    // triangles, 10 dofs: vertices; per local edge (a, b) the node nearer a, then the node nearer b -- an edge's two
    // dofs are numbered from its LOWER global vertex to its higher one: nverts + 2 e + {0, 1}; the cell's own dof:
    // nverts + 2 n_edges + the caller's index of the cell
    const int nd = 10;
    for (int a = 0; a < nv; ++a) cd0[c * nd + a] = cells[c * nv + a];
    for (int e = 0; e < ne; ++e) {
      const int flip = cells[c * nv + EDGE2[e][0]] > cells[c * nv + EDGE2[e][1]] ? 1 : 0;
      const int64_t first = nverts + 2 * (int64_t)cell_edges[c * ne + e];
      cd0[c * nd + nv + 2 * e] = (int32_t)(first + flip);
      cd0[c * nd + nv + 2 * e + 1] = (int32_t)(first + 1 - flip);
    }
    // (the CALLER's cell index, not the kernel order's: the same on every rank of a partitioned mesh, whose parts list
    // their cells in ascending global id -- sender and receiver then order these dofs alike)
    cd0[c * nd + 9] = (int32_t)(nverts + 2 * n_edges + (cell_perm ? cell_perm[c] : c));
    return;
  }
  const int nd = nv + ne;
  for (int a = 0; a < nv; ++a) cd0[c * nd + a] = cells[c * nv + a];
  for (int e = 0; e < ne; ++e) cd0[c * nd + nv + e] = (int32_t)(nverts + cell_edges[c * ne + e]);
}

__global__ __launch_bounds__(256) void k_dof_coords(const double *__restrict__ coords, const uint64_t *__restrict__ edge_keys,
                                                    int64_t nverts, int64_t n, int d, double *__restrict__ x, int degree = 2,
                                                    int64_t n_edges = 0, const int32_t *__restrict__ cells = nullptr,
                                                    const uint64_t *__restrict__ face_keys = nullptr,
                                                    const int32_t *__restrict__ cell_kernel_index = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i < nverts) {
    for (int k = 0; k < d; ++k) x[i * d + k] = coords[i * d + k];
  } else if (degree == 3) {
    // gll_warped P3 (reference fracstep.py:170,181): the edge nodes at the Gauss-Lobatto-Legendre points of the edge,
    // counted from its lower global vertex; the cell's node at its centroid
    if (i < nverts + 2 * n_edges) {
      const int64_t e = (i - nverts) >> 1;
      const double t = ((i - nverts) & 1) ? 0.5 + 0.5 / sqrt(5.0) : 0.5 - 0.5 / sqrt(5.0);
      const uint64_t key = edge_keys[e];
      const int64_t a = (int64_t)(key / (uint64_t)nverts), b = (int64_t)(key % (uint64_t)nverts);
      for (int k = 0; k < d; ++k) x[i * d + k] = (1.0 - t) * coords[a * d + k] + t * coords[b * d + k];
    } else if (d == 3) {  // the node of a face at its centroid
      const uint64_t key = face_keys[i - nverts - 2 * n_edges], nv_ = (uint64_t)nverts;
      const int64_t a = (int64_t)(key / (nv_ * nv_)), b = (int64_t)((key / nv_) % nv_), c = (int64_t)(key % nv_);
      for (int k = 0; k < d; ++k) x[i * d + k] = (coords[a * d + k] + coords[b * d + k] + coords[c * d + k]) / 3.0;
    } else {
      const int64_t cc = i - nverts - 2 * n_edges;  // caller's cell index -> kernel cell index
      const int64_t c = cell_kernel_index ? cell_kernel_index[cc] : cc;
      for (int k = 0; k < d; ++k) {
        double v = 0.0;
        for (int a = 0; a <= d; ++a) v += coords[(int64_t)cells[c * (d + 1) + a] * d + k];
        x[i * d + k] = v / (double)(d + 1);
      }
    }
  } else {
    const uint64_t key = edge_keys[i - nverts];
    const int64_t a = (int64_t)(key / (uint64_t)nverts), b = (int64_t)(key % (uint64_t)nverts);
    for (int k = 0; k < d; ++k) x[i * d + k] = 0.5 * (coords[a * d + k] + coords[b * d + k]);
  }
}

__global__ __launch_bounds__(256) void k_invert(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ rank) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) rank[perm[i]] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_relabel(const int32_t *__restrict__ in, const int32_t *__restrict__ rank, int64_t n,
                                                 int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rank[in[i]];
}

__global__ __launch_bounds__(256) void k_compose(const int32_t *__restrict__ r1, const int32_t *__restrict__ r2, int64_t n,
                                                 int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = r2[r1[i]];
}

__global__ __launch_bounds__(256) void k_permute_points(const double *__restrict__ xin, const int32_t *__restrict__ rank,
                                                        int64_t n, int d, double *__restrict__ xout) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < d; ++k) xout[(int64_t)rank[i] * d + k] = xin[i * d + k];
}

// ---- dof -> (cell, local index) pairs grouped by dof -----------------------------------------
__global__ __launch_bounds__(256) void k_pair_keys(const int32_t *__restrict__ cell_dofs, int64_t npairs,
                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  keys[i] = (uint32_t)cell_dofs[i];
  vals[i] = (uint32_t)i;  // = cell * nd + local index  (< 2^32 is checked by the caller)
}

// start[r] = first sorted position whose key is >= r (every dof occurs at least once)
__global__ __launch_bounds__(256) void k_group_starts(const uint32_t *__restrict__ ks, int64_t npairs, int64_t n,
                                                      int64_t *__restrict__ start) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  const uint32_t k = ks[i];
  if (i == 0) {
    for (uint32_t r = 0; r <= k; ++r) start[r] = 0;
  } else {
    const uint32_t kp = ks[i - 1];
    for (uint32_t r = kp + 1; r <= k; ++r) start[r] = i;
  }
  if (i == npairs - 1)
    for (int64_t r = (int64_t)k + 1; r <= n; ++r) start[r] = npairs;
}

// ---- the row kernel: one wave per row ---------------------------------------------------------
// Candidates = the column-space dofs of the row's adjacent cells (in adjacency order), held in LDS;
// first-occurrence flags and the rank of every candidate among the row's distinct columns by
// all-pairs comparison (rows have 15 .. ~70 distinct columns; ~2 ms for 17 M rows).
// mode 0: len[r] = number of distinct columns.
// mode 1: cols (SELL slots, ascending per row, padding = own row), and per adjacency pair the cell,
//         the row dof's local index and the in-row positions of the cell's column dofs.
struct RowArgs {
  int64_t n_rows, n_cols;
  const int64_t *start;        // [n_rows+1] pair offsets of the rows (pairs grouped by row dof)
  const uint32_t *pair;        // [npairs] cell * nd_r + local index, cells ascending per row
  int nd_r, nd_c;
  const int32_t *col_dofs;     // [n_cells][nd_c]
  int32_t *len;                // mode 0
  const int64_t *slice_ptr;    // mode 1 ...
  int32_t *cols;
  const int64_t *adj_ptr;
  int32_t *adj_cell;
  uint8_t *adj_loc;            // may be NULL (rectangular patterns reuse the row space's)
  uint8_t *adj_pos;
  int pw;
  int *err;                    // 1: a row has more than ROW_CAP candidates; 2: a row is wider than 255
};

template <int MODE>
__device__ __forceinline__ void k_rows_one(const RowArgs &A, int64_t r, int lane, int32_t *cand, uint16_t *rk, uint8_t *fs);

template <int MODE>
__global__ __launch_bounds__(256) void k_rows(RowArgs A) {
  __shared__ int32_t cand_s[4][ROW_CAP];
  __shared__ uint16_t rank_s[4][ROW_CAP];
  __shared__ uint8_t first_s[4][ROW_CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n_slots_rows = ((A.n_rows + SLICE - 1) / SLICE) * SLICE;
  int32_t *cand = cand_s[wave];
  uint16_t *rk = rank_s[wave];
  uint8_t *fs = first_s[wave];
  // grid-stride over the rows (a grid of n_rows / 4 blocks would exceed 2^32 threads beyond 2^26 rows)
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < (MODE == 0 ? A.n_rows : n_slots_rows); r += (int64_t)gridDim.x * 4)
    k_rows_one<MODE>(A, r, lane, cand, rk, fs);
}

template <int MODE>
__device__ __forceinline__ void k_rows_one(const RowArgs &A, int64_t r, int lane, int32_t *cand, uint16_t *rk, uint8_t *fs) {
  if (MODE == 1 && r >= A.n_rows) {  // rows that only pad the last slice: columns point at 0, no pairs
    const int64_t s = r >> 6;
    const int64_t base = A.slice_ptr[s];
    const int width = (int)((A.slice_ptr[s + 1] - base) >> 6);
    for (int k = lane; k < width; k += 64)
      A.cols[base + (int64_t)(k / KV) * (SLICE * KV) + (r & 63) * KV + (k % KV)] = 0;
    const int64_t abase = A.adj_ptr[s];
    const int T = (int)((A.adj_ptr[s + 1] - abase) >> 6);
    for (int t = lane; t < T; t += 64) {
      const int64_t pidx = abase + (int64_t)t * 64 + (r & 63);
      A.adj_cell[pidx] = -1;
      if (A.adj_loc) A.adj_loc[pidx] = 0;
      for (int j = 0; j < A.pw; ++j) A.adj_pos[pidx * A.pw + j] = 0;
    }
    return;
  }
  const int64_t p0 = A.start[r];
  const int m = (int)(A.start[r + 1] - p0);
  const int nc = m * A.nd_c;
  if (nc > ROW_CAP) {
    if (lane == 0) atomicExch(A.err, 1);
    if (MODE == 0 && lane == 0) A.len[r] = 0;
    return;
  }
  for (int i = lane; i < nc; i += 64) {
    const int t = i / A.nd_c, j = i - t * A.nd_c;
    const int64_t cell = A.pair[p0 + t] / (uint32_t)A.nd_r;
    cand[i] = A.col_dofs[cell * A.nd_c + j];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are done
  int nfirst = 0;
  for (int i = lane; i < nc; i += 64) {
    const int32_t v = cand[i];
    bool f = true;
    for (int j = 0; j < i; ++j) f = f && (cand[j] != v);
    fs[i] = f ? 1 : 0;
    nfirst += f ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) nfirst += __shfl_xor(nfirst, off, 64);
  if (MODE == 0) {
    if (lane == 0) A.len[r] = nfirst;
    return;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int i = lane; i < nc; i += 64) {
    const int32_t v = cand[i];
    int q = 0;
    for (int j = 0; j < nc; ++j) q += (fs[j] && cand[j] < v) ? 1 : 0;
    rk[i] = (uint16_t)q;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int64_t s = r >> 6;
  const int rl = (int)(r & 63);
  const int64_t base = A.slice_ptr[s];
  const int width = (int)((A.slice_ptr[s + 1] - base) >> 6);
  if (width > 255 && lane == 0) atomicExch(A.err, 2);
  auto slot = [&](int k) { return base + (int64_t)(k / KV) * (SLICE * KV) + rl * KV + (k % KV); };
  for (int i = lane; i < nc; i += 64)
    if (fs[i]) A.cols[slot(rk[i])] = cand[i];
  const int64_t nlim = A.n_rows < A.n_cols ? A.n_rows : A.n_cols;
  const int32_t own = r < nlim ? (int32_t)r : 0;
  for (int k = nfirst + lane; k < width; k += 64) A.cols[slot(k)] = own;  // padding: own row, value 0
  // adjacency pairs of this row: (t, lane) of slice s at adj_ptr[s] + t*64 + lane
  const int64_t abase = A.adj_ptr[s];
  const int T = (int)((A.adj_ptr[s + 1] - abase) >> 6);
  for (int t = lane; t < T; t += 64) {
    const int64_t pidx = abase + (int64_t)t * 64 + rl;
    if (t < m) {
      const uint32_t pv = A.pair[p0 + t];
      A.adj_cell[pidx] = (int32_t)(pv / (uint32_t)A.nd_r);
      if (A.adj_loc) A.adj_loc[pidx] = (uint8_t)(pv % (uint32_t)A.nd_r);
      for (int j = 0; j < A.pw; ++j) A.adj_pos[pidx * A.pw + j] = j < A.nd_c ? (uint8_t)rk[t * A.nd_c + j] : 0;
    } else {
      A.adj_cell[pidx] = -1;
      if (A.adj_loc) A.adj_loc[pidx] = 0;
      for (int j = 0; j < A.pw; ++j) A.adj_pos[pidx * A.pw + j] = 0;
    }
  }
}

// per slice: storage width (max row length, rounded up to OX_KV, at least OX_KV) and adjacency depth
__global__ __launch_bounds__(256) void k_slice_sizes(const int32_t *__restrict__ row_len, const int64_t *__restrict__ start,
                                                     int64_t n_rows, int64_t n_slices, int64_t *__restrict__ width64,
                                                     int64_t *__restrict__ depth64, int32_t *__restrict__ width32) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + wave;
  if (s >= n_slices) return;
  const int64_t r = s * 64 + lane;
  int w = r < n_rows ? row_len[r] : 0;
  int t = (start && r < n_rows) ? (int)(start[r + 1] - start[r]) : 0;
  for (int off = 32; off > 0; off >>= 1) {
    w = max(w, __shfl_xor(w, off, 64));
    t = max(t, __shfl_xor(t, off, 64));
  }
  w = ((w + KV - 1) / KV) * KV;
  if (w < KV) w = KV;
  if (lane == 0) {
    width64[s] = (int64_t)w * SLICE;
    if (depth64) depth64[s] = (int64_t)t * SLICE;
    width32[s] = w;
  }
}

// (Round 5 tried LENGTH CLASSES here -- lengths within a factor 1 + 2^-s share a sort key, so that the stable sort keeps
// the rows of a class in their order along the locality curve and a slice holds mesh neighbours --: on the refined Delaunay
// mesh of the bench, whose rows take few distinct lengths, s = 3 shrank the mean column window of a window block from 3093
// to 2847 entries for 0.85 % more padding and moved neither assemble_first (13.7 ms) nor the mat-vecs by more than 3 %;
// s >= 4 changed nothing.  Removed again: profiles/r05_assemble_row_blocks.txt.)
__global__ __launch_bounds__(256) void k_window_keys(const int32_t *__restrict__ len, int64_t n, int window, int lmax,
                                                     uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  keys[r] = (uint64_t)(r / window) * (uint64_t)(lmax + 1) + (uint64_t)(lmax - len[r]);
  ids[r] = (int32_t)r;
}

__global__ __launch_bounds__(256) void k_gather_i32(const int32_t *__restrict__ in, const int32_t *__restrict__ idx,
                                                    int64_t n, int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

inline unsigned nblk(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }
// grid of the wave-per-row kernels: 4 rows per block, capped (the kernels stride)
inline unsigned row_grid(int64_t rows) { return (unsigned)std::min<int64_t>((rows + 3) / 4, (int64_t)1 << 22); }

int reduce_max_i32(const int32_t *v, int64_t n, int32_t *out_host, hipStream_t st) {
  *out_host = 0;
  if (n == 0) return 0;
  DevBuf res, tmp;
  OX_TRY(res.alloc(sizeof(int32_t)));
  size_t tb = 0;
  OX_HIP(rocprim::reduce(nullptr, tb, v, res.as<int32_t>(), (int32_t)0, (size_t)n, rocprim::maximum<int32_t>(), st));
  OX_TRY(tmp.alloc(tb));
  OX_HIP(rocprim::reduce(tmp.p, tb, v, res.as<int32_t>(), (int32_t)0, (size_t)n, rocprim::maximum<int32_t>(), st));
  OX_HIP(hipMemcpyAsync(out_host, res.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

int reduce_sum_i32(const int32_t *v, int64_t n, int64_t *out_host, hipStream_t st) {
  *out_host = 0;
  if (n == 0) return 0;
  DevBuf res, tmp;
  OX_TRY(res.alloc(sizeof(int64_t)));
  auto in = rocprim::make_transform_iterator(v, [] __device__(int32_t a) { return (int64_t)a; });
  size_t tb = 0;
  OX_HIP(rocprim::reduce(nullptr, tb, in, res.as<int64_t>(), (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
  OX_TRY(tmp.alloc(tb));
  OX_HIP(rocprim::reduce(tmp.p, tb, in, res.as<int64_t>(), (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
  OX_HIP(hipMemcpyAsync(out_host, res.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // namespace

// ================================== the objects ====================================================
struct ox_mesh {
  int gdim = 0, tile_bits = 0, key_bits = 18;
  int lattice = 1;  // vertices on a tensor grid (few distinct values per coordinate): tiled order; else Z-order
  int64_t nv = 0, nc = 0;
  double lo[3] = {0, 0, 0}, span[3] = {1, 1, 1};
  DevBuf coords;     // [nv][gdim]
  DevBuf cells;      // [nc][gdim+1] int32, KERNEL order
  DevBuf cell_perm;  // [nc] int32: kernel index -> caller's cell index
  DevBuf geom;       // [nc][gs]
};

struct ox_pattern_store {  // SELL-64 pattern + 16-bit stream + width bins
  int64_t n_rows = 0, n_cols = 0, n_slices = 0, size = 0, nnz = 0, n16 = 0;
  DevBuf slice_ptr, cols, row_len, cols16, cbase, bin_slices;
  std::vector<int32_t> widths, bin_width;
  std::vector<int64_t> bin_ptr;
  // row blocks of the one-launch assembly kernels (include/oasisx_hip.h: ox_pattern_info.row_blk_*)
  DevBuf row_blk_ptr;
  int32_t n_row_blocks = 0;
  int64_t row_blk_entries = 0;
};

struct ox_space {
  const ox_mesh *mesh = nullptr;
  int degree = 0, nd = 0, pw = 0, window = 0;
  int64_t n = 0, n_edges = 0, npairs = 0;  // dofs, edges, padded adjacency pairs
  int64_t n_faces = 0;  // P3 on tetrahedra: faces (one dof each)
  int64_t n_rows = 0;   // rows of the space's patterns: n, or the owned dofs of a mesh-partitioned space (they come first)
  DevBuf cell_dofs;     // [nc][nd] int32, final numbering
  DevBuf x;             // [n][gdim]
  DevBuf rank_initial;  // [n] int32: initial dof (vertex id, or nv + edge id) -> final dof
  DevBuf edge_keys;     // [n_edges] uint64: min_vertex * nv + max_vertex, ascending (edge id = position)
  DevBuf face_keys;     // [n_faces] uint64: (v0 * nv + v1) * nv + v2 of the sorted vertex triple, ascending (P3 tetrahedra)
  DevBuf start, pair;   // pairs grouped by final dof (kept for rectangular patterns)
  DevBuf adj_ptr, adj_cell, adj_loc, adj_pos, adj_count;
  ox_pattern_store P;
  DevBuf row_pos;       // [n] int32: position of final row r in the pure locality order (before the length sort)
  // LDS-window stream of the square pattern (ox_space_windows; built on first request)
  DevBuf wb_slices, wb_waves, wb_ptr, wlist, wt_ptr, wcode;
  int32_t n_wblocks = 0, w_max = 0;
  int64_t n_list = 0, n_tiles = 0, n_over_16bit = 0;
  bool windows_built = false;
  int windows_split = 0;  // split_entries the stream was built with
};

struct ox_rect {
  const ox_space *R = nullptr, *C = nullptr;
  int pw = 0;
  DevBuf pos;  // [R.npairs][pw]
  ox_pattern_store P;
};

namespace {

// bits per coordinate: 4 lattice steps per mean point spacing (fem.default_key_bits is the twin of
// this integer arithmetic; OX_KEY_BITS overrides both)
int default_key_bits(int64_t n_points, int d, int tb) {
  const char *e = getenv("OX_KEY_BITS");
  if (e) return atoi(e);
  auto pw = [&](int64_t L) {
    __int128 v = 1;
    for (int k = 0; k < d; ++k) v *= L;
    return v;
  };
  int64_t L = 1;
  while (pw(L) < (__int128)n_points) L += L < 64 ? 1 : std::max<int64_t>(1, L / 64);
  while (L > 1 && pw(L - 1) >= (__int128)n_points) --L;
  int b = 0;
  while (((int64_t)1 << b) < L) ++b;
  return std::min(18, std::max(b + 2, tb + 1));
}

// Brick order of the degree-2 spaces on lattice meshes (fem.default_brick_shift is the twin): bricks of about 8 points
// a side (512 rows: one window block of k_spmv_win); shift = bits - floor(log2(L / 8)) with L = points per direction
// as default_key_bits counts them; 0 (no bricks) below 16 points per direction.
int default_brick_shift(int64_t n_points, int d, int bits) {
  auto pw = [&](int64_t L) {
    __int128 v = 1;
    for (int k = 0; k < d; ++k) v *= L;
    return v;
  };
  int64_t L = 1;
  while (pw(L) < (__int128)n_points) L += L < 64 ? 1 : std::max<int64_t>(1, L / 64);
  while (L > 1 && pw(L - 1) >= (__int128)n_points) --L;
  const int64_t nb = L / 8;
  if (nb < 2) return 0;
  int lg = 0;
  while (((int64_t)2 << lg) <= nb) ++lg;
  const int bs = bits - lg;
  return bs > 0 && bs < bits ? bs : 0;
}

KeySpec key_spec(const ox_mesh *M, int tb, int64_t n_points, bool brick = false) {
  KeySpec K;
  K.d = M->gdim;
  K.tb = tb;
  K.curve = M->lattice ? 0 : 1;
  int bits = K.curve ? 18 : default_key_bits(n_points, M->gdim, tb);
  while (bits > 4 && (K.curve ? K.d * bits : (K.d - 1) * tb + K.d * bits) > 63) --bits;
  if (tb > bits) K.tb = bits;
  K.bits = bits;
  K.bs = (!K.curve && brick) ? default_brick_shift(n_points, M->gdim, bits) : 0;
  for (int k = 0; k < 3; ++k) {
    K.lo[k] = M->lo[k];
    K.inv[k] = (double)((1ull << bits) - 1) / M->span[k];
  }
  return K;
}

// pattern of (rows grouped in `start`/`pair`, columns from col_dofs) with given row lengths
int finish_pattern(ox_pattern_store &P, hipStream_t st) {
  // 16-bit column stream
  if (P.size > 0) {
    OX_TRY(P.cols16.alloc(sizeof(uint16_t) * (size_t)P.size));
    OX_TRY(P.cbase.alloc(sizeof(int32_t) * 2 * (size_t)(P.size / (SLICE * KV))));
    ox_sell S{};
    S.n_rows = P.n_rows, S.n_cols = P.n_cols, S.n_slices = (int32_t)P.n_slices;
    S.slice_ptr = P.slice_ptr.as<int64_t>(), S.cols = P.cols.as<int32_t>();
    if (ox_sell_compress_cols(&S, P.cols16.as<uint16_t>(), P.cbase.as<int32_t>(), &P.n16, st)) return -1;
  }
  // width bins for the LDS-accumulating row kernels: slices sorted (stably) by width
  std::vector<int32_t> order(P.n_slices);
  for (int64_t s = 0; s < P.n_slices; ++s) order[s] = (int32_t)s;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return P.widths[a] < P.widths[b]; });
  P.bin_width.clear();
  P.bin_ptr.assign(1, 0);
  for (int64_t i = 0; i < P.n_slices; ++i) {
    const int32_t w = P.widths[order[i]];
    if (P.bin_width.empty() || P.bin_width.back() != w) {
      if (!P.bin_width.empty()) P.bin_ptr.push_back(i);
      P.bin_width.push_back(w);
    }
  }
  P.bin_ptr.push_back(P.n_slices);
  if (P.bin_width.empty()) P.bin_ptr.assign(1, 0);
  {  // a bin with fewer than max(64, n_slices / 64) slices joins the next wider one (fem.merge_small_bins)
    const int64_t thresh = std::max<int64_t>(64, P.n_slices / 64);
    std::vector<int32_t> bw;
    std::vector<int64_t> bp(1, 0);
    for (size_t b = 0; b < P.bin_width.size(); ++b) {
      const bool last = b + 1 == P.bin_width.size();
      if (P.bin_ptr[b + 1] - bp.back() >= thresh || last) {
        bw.push_back(P.bin_width[b]);
        bp.push_back(P.bin_ptr[b + 1]);
      }
    }
    if (!P.bin_width.empty()) P.bin_width.swap(bw), P.bin_ptr.swap(bp);
  }
  OX_TRY(P.bin_slices.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(P.n_slices, 1)));
  if (P.n_slices)
    OX_HIP(hipMemcpyAsync(P.bin_slices.p, order.data(), sizeof(int32_t) * (size_t)P.n_slices, hipMemcpyHostToDevice, st));
  // row blocks: consecutive slices in storage order, greedily, at most OX_ROW_BLOCK_WAVES per block and
  // OX_ROW_BLOCK_LDS bytes of accumulators (fem.row_blocks is the torch twin)
  std::vector<int32_t> blk(1, 0);
  {
    const int64_t cap = OX_ROW_BLOCK_LDS / (int64_t)sizeof(double);
    int64_t used = 0, big = 0;
    int cnt = 0;
    bool fits = true;
    for (int64_t s = 0; s < P.n_slices; ++s) {
      const int64_t e = (int64_t)P.widths[s] * SLICE;
      if (e > cap) fits = false;
      if (cnt == OX_ROW_BLOCK_WAVES || used + e > cap) {
        blk.push_back((int32_t)s);
        used = 0, cnt = 0;
      }
      used += e, ++cnt;
      big = std::max(big, used);
    }
    if (P.n_slices) blk.push_back((int32_t)P.n_slices);
    if (!fits) blk.assign(1, 0), big = 0;  // a row wider than the budget: the width bins serve this pattern
    P.n_row_blocks = (int32_t)blk.size() - 1;
    P.row_blk_entries = big;
  }
  OX_TRY(P.row_blk_ptr.alloc(sizeof(int32_t) * blk.size()));
  OX_HIP(hipMemcpyAsync(P.row_blk_ptr.p, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

// slice widths, slice_ptr (and optionally adjacency depths, adj_ptr) from the row lengths
int layout_slices(ox_pattern_store &P, const int64_t *start, DevBuf *adj_ptr, int64_t *npairs, hipStream_t st) {
  P.n_slices = (P.n_rows + SLICE - 1) / SLICE;
  const int64_t ns = P.n_slices;
  DevBuf w64, d64, w32;
  OX_TRY(w64.alloc(sizeof(int64_t) * (size_t)(ns + 1)));
  OX_TRY(d64.alloc(sizeof(int64_t) * (size_t)(ns + 1)));
  OX_TRY(w32.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
  OX_HIP(hipMemsetAsync(w64.p, 0, w64.bytes, st));
  OX_HIP(hipMemsetAsync(d64.p, 0, d64.bytes, st));
  if (ns)
    hipLaunchKernelGGL(k_slice_sizes, dim3(nblk(ns, 4)), dim3(256), 0, st, P.row_len.as<int32_t>(), start, P.n_rows, ns,
                       w64.as<int64_t>(), adj_ptr ? d64.as<int64_t>() : nullptr, w32.as<int32_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(P.slice_ptr.alloc(sizeof(int64_t) * (size_t)(ns + 1)));
  OX_TRY(exclusive_scan_i64(w64.as<int64_t>(), P.slice_ptr.as<int64_t>(), (size_t)(ns + 1), st));
  OX_HIP(hipMemcpy(&P.size, P.slice_ptr.as<int64_t>() + ns, sizeof(int64_t), hipMemcpyDeviceToHost));
  P.widths.assign((size_t)ns, 0);
  if (ns) OX_HIP(hipMemcpy(P.widths.data(), w32.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
  if (adj_ptr) {
    OX_TRY(adj_ptr->alloc(sizeof(int64_t) * (size_t)(ns + 1)));
    OX_TRY(exclusive_scan_i64(d64.as<int64_t>(), adj_ptr->as<int64_t>(), (size_t)(ns + 1), st));
    OX_HIP(hipMemcpy(npairs, adj_ptr->as<int64_t>() + ns, sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  OX_TRY(reduce_sum_i32(P.row_len.as<int32_t>(), P.n_rows, &P.nnz, st));
  OX_TRY(P.cols.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(P.size, 1)));
  return 0;
}

int check_row_error(int *err_dev, const char *what) {
  int e = 0;
  OX_HIP(hipMemcpy(&e, err_dev, sizeof(int), hipMemcpyDeviceToHost));
  if (e == 1) OX_FAIL("%s: a dof touches so many cells that its row has more than %d candidate columns", what, ROW_CAP);
  if (e == 2) OX_FAIL("%s: a row is longer than 255 entries (position bytes overflow)", what);
  return 0;
}

// pairs (dof, cell*nd + k) grouped by dof, cells ascending: start [n+1], pair [nc*nd]
int group_pairs(const int32_t *cell_dofs, int64_t nc, int nd, int64_t n, DevBuf &start, DevBuf &pair, hipStream_t st) {
  const int64_t np = nc * nd;
  if (np >= (int64_t)1 << 32) OX_FAIL("set-up: %lld (cell, dof) pairs exceed 2^32", (long long)np);
  DevBuf k_in, k_out, v_in;
  OX_TRY(k_in.alloc(sizeof(uint32_t) * (size_t)np));
  OX_TRY(k_out.alloc(sizeof(uint32_t) * (size_t)np));
  OX_TRY(v_in.alloc(sizeof(uint32_t) * (size_t)np));
  OX_TRY(pair.alloc(sizeof(uint32_t) * (size_t)np));
  hipLaunchKernelGGL(k_pair_keys, dim3(nblk(np)), dim3(256), 0, st, cell_dofs, np, k_in.as<uint32_t>(), v_in.as<uint32_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(sort_pairs(k_in.as<uint32_t>(), k_out.as<uint32_t>(), v_in.as<uint32_t>(), pair.as<uint32_t>(), (size_t)np,
                    bits_for((uint64_t)std::max<int64_t>(n - 1, 1)), st));
  OX_TRY(start.alloc(sizeof(int64_t) * (size_t)(n + 1)));
  OX_HIP(hipMemsetAsync(start.p, 0, start.bytes, st));
  hipLaunchKernelGGL(k_group_starts, dim3(nblk(np)), dim3(256), 0, st, k_out.as<uint32_t>(), np, n, start.as<int64_t>());
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // namespace

// ================================== C ABI ===========================================================
// frame: the bounding box, lattice flag, tile bits and cell count of the WHOLE mesh when `cells` is one rank's part of
// it (ox_mesh_create_sub) -- the ordering keys of all parts then live on the same lattice as the whole mesh's
struct MeshFrame {
  double lo[3], span[3];
  int lattice, tile_bits;
  int64_t n_cells_ref;
};
static int mesh_create_impl(const double *coords, int64_t n_vertices, const int32_t *cells, int64_t n_cells, int gdim,
                            int on_device, int tile_bits, const MeshFrame *frame, ox_mesh **out) {
  if (!coords || !cells || !out) OX_FAIL("ox_mesh_create: null argument");
  if (gdim != 2 && gdim != 3) OX_FAIL("ox_mesh_create: gdim=%d (triangles or tetrahedra)", gdim);
  if (n_vertices < gdim + 1 || n_cells < 1) OX_FAIL("ox_mesh_create: empty mesh");
  if (n_cells * 6 >= ((int64_t)1 << 31)) OX_FAIL("ox_mesh_create: %lld cells: cell-edge indices exceed 2^31", (long long)n_cells);
  hipStream_t st = nullptr;
  ox_mesh *M = new ox_mesh();
  struct Guard {
    ox_mesh *m;
    ~Guard() { delete m; }
  } guard{M};
  M->gdim = gdim, M->nv = n_vertices, M->nc = n_cells;
  const int nv = gdim + 1;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  OX_TRY(M->coords.alloc(sizeof(double) * (size_t)n_vertices * gdim));
  OX_HIP(hipMemcpy(M->coords.p, coords, M->coords.bytes, kind));
  DevBuf cells_in;
  OX_TRY(cells_in.alloc(sizeof(int32_t) * (size_t)n_cells * nv));
  OX_HIP(hipMemcpy(cells_in.p, cells, cells_in.bytes, kind));
  // bounding box
  if (frame) {
    for (int k = 0; k < 3; ++k) M->lo[k] = frame->lo[k], M->span[k] = k < gdim ? std::max(frame->span[k], 1e-300) : 1.0;
  } else {
    const int nb = 512;
    DevBuf part;
    OX_TRY(part.alloc(sizeof(double) * 6 * nb));
    hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, st, M->coords.as<double>(), n_vertices, gdim, part.as<double>());
    OX_LAUNCH_CHECK();
    std::vector<double> h(6 * nb);
    OX_HIP(hipMemcpy(h.data(), part.p, sizeof(double) * 6 * nb, hipMemcpyDeviceToHost));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < nb; ++b)
      for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], h[b * 6 + k]), hi[k] = std::max(hi[k], h[b * 6 + 3 + k]);
    for (int k = 0; k < 3; ++k) {
      M->lo[k] = k < gdim ? lo[k] : 0.0;
      M->span[k] = k < gdim ? std::max(hi[k] - lo[k], 1e-300) : 1.0;
    }
  }
  // Lattice or not (fem.mesh_is_lattice is the twin): a tensor-grid mesh has about n^(1/d) distinct values
  // per coordinate, an unstructured one about n.  Lattice meshes keep whole x-lines contiguous (the gathers
  // of a wave coalesce); anything else is ordered along a Z-order curve: 64 consecutive rows are then a
  // compact cluster instead of a thin tube through the whole x extent (Delaunay mesh, 2.3 M P2 rows:
  // SpMV 350 -> 262 us).  OX_ORDER=tiles|curve overrides.
  if (frame) {
    M->lattice = frame->lattice;
  } else {
    int64_t L = 1;
    while (true) {
      __int128 v = 1;
      for (int k = 0; k < gdim; ++k) v *= L;
      if (v >= (__int128)n_vertices) break;
      ++L;
    }
    DevBuf bm;
    OX_TRY(bm.alloc(sizeof(unsigned) << (LATTICE_BITS - 5)));
    std::vector<unsigned> hb((size_t)1 << (LATTICE_BITS - 5));
    int lattice = 1;
    for (int k = 0; k < gdim && lattice; ++k) {
      OX_HIP(hipMemset(bm.p, 0, bm.bytes));
      hipLaunchKernelGGL(k_mark_coord, dim3(1024), dim3(256), 0, st, M->coords.as<double>(), n_vertices, gdim, k, M->lo[k],
                         (double)((1u << LATTICE_BITS) - 1) / M->span[k], bm.as<unsigned>());
      OX_LAUNCH_CHECK();
      OX_HIP(hipMemcpy(hb.data(), bm.p, bm.bytes, hipMemcpyDeviceToHost));
      int64_t cnt = 0;
      for (unsigned w : hb) cnt += __builtin_popcount(w);
      if (cnt > 4 * L + 4) lattice = 0;
    }
    const char *e = getenv("OX_ORDER");
    if (e) lattice = strcmp(e, "curve") != 0;
    M->lattice = lattice;
  }
  // tiles of about 24 vertex lines a side; OX_TILE_BITS / the argument override
  if (tile_bits < 0) {
    const char *e = getenv("OX_TILE_BITS");
    if (e) tile_bits = atoi(e);
    else {
      const double lines = std::max(std::pow((double)n_vertices, 1.0 / gdim), 1.0);
      tile_bits = (int)std::min(6.0, std::max(0.0, std::nearbyint(std::log2(lines / 24.0))));
    }
  }
  if (frame) tile_bits = frame->tile_bits;
  M->tile_bits = tile_bits;
  const KeySpec K = key_spec(M, tile_bits, frame ? frame->n_cells_ref : n_cells);
  M->key_bits = K.bits;
  // kernel cell order: tiled order of the centroids (stable: ties keep the caller's order)
  {
    DevBuf k_in, k_out, v_in;
    OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)n_cells));
    OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)n_cells));
    OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)n_cells));
    OX_TRY(M->cell_perm.alloc(sizeof(int32_t) * (size_t)n_cells));
    hipLaunchKernelGGL(k_cell_keys, dim3(nblk(n_cells)), dim3(256), 0, st, M->coords.as<double>(), cells_in.as<int32_t>(),
                       n_cells, K, k_in.as<uint64_t>(), v_in.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), M->cell_perm.as<int32_t>(),
                      (size_t)n_cells, key_end_bit(K), st));
  }
  OX_TRY(M->cells.alloc(sizeof(int32_t) * (size_t)n_cells * nv));
  hipLaunchKernelGGL(k_gather_cells, dim3(nblk(n_cells)), dim3(256), 0, st, cells_in.as<int32_t>(), M->cell_perm.as<int32_t>(),
                     n_cells, nv, M->cells.as<int32_t>());
  OX_LAUNCH_CHECK();
  const int gs = gdim == 2 ? 6 : 10;
  OX_TRY(M->geom.alloc(sizeof(double) * (size_t)n_cells * gs));
  hipLaunchKernelGGL(k_geometry, dim3(nblk(n_cells)), dim3(256), 0, st, M->coords.as<double>(), M->cells.as<int32_t>(), n_cells,
                     gdim, M->geom.as<double>());
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  guard.m = nullptr;
  *out = M;
  return 0;
}

extern "C" int ox_mesh_create(const double *coords, int64_t n_vertices, const int32_t *cells, int64_t n_cells, int gdim,
                              int on_device, int tile_bits, ox_mesh **out) {
  return mesh_create_impl(coords, n_vertices, cells, n_cells, gdim, on_device, tile_bits, nullptr, out);
}

extern "C" int ox_mesh_create_sub(const double *coords, int64_t n_vertices, const int32_t *cells, int64_t n_cells, int gdim,
                                  int on_device, const double *lo, const double *span, int lattice, int tile_bits,
                                  int64_t n_cells_whole, ox_mesh **out) {
  if (!lo || !span) OX_FAIL("ox_mesh_create_sub: null argument");
  if (tile_bits < 0 || tile_bits > 6) OX_FAIL("ox_mesh_create_sub: tile_bits=%d (the whole mesh's, 0..6)", tile_bits);
  if (n_cells_whole < n_cells) OX_FAIL("ox_mesh_create_sub: n_cells_whole < n_cells");
  MeshFrame F{};
  for (int k = 0; k < 3; ++k) F.lo[k] = k < gdim ? lo[k] : 0.0, F.span[k] = k < gdim ? span[k] : 1.0;
  F.lattice = lattice ? 1 : 0, F.tile_bits = tile_bits, F.n_cells_ref = n_cells_whole;
  return mesh_create_impl(coords, n_vertices, cells, n_cells, gdim, on_device, tile_bits, &F, out);
}

extern "C" int ox_mesh_destroy(ox_mesh *M) {
  delete M;
  return 0;
}

extern "C" int ox_mesh_view(const ox_mesh *M, ox_mesh_info *v) {
  if (!M || !v) OX_FAIL("ox_mesh_view: null argument");
  memset(v, 0, sizeof(*v));
  v->gdim = M->gdim, v->tile_bits = M->tile_bits;
  v->n_vertices = M->nv, v->n_cells = M->nc;
  for (int k = 0; k < 3; ++k) v->lo[k] = M->lo[k], v->span[k] = M->span[k];
  v->coords = M->coords.as<double>();
  v->cells = M->cells.as<int32_t>();
  v->cell_perm = M->cell_perm.as<int32_t>();
  v->cells_struct.gdim = M->gdim;
  v->cells_struct.n_cells = M->nc;
  v->cells_struct.geom = M->geom.as<double>();
  v->lattice = M->lattice;
  return 0;
}

// ghosts behind the owned dofs, ordered by (owner rank, initial dof id)
__global__ __launch_bounds__(256) void k_point_keys_part(const double *__restrict__ x, int64_t n, KeySpec K,
                                                         const int32_t *__restrict__ owner, int rank,
                                                         uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = owner[i];
  keys[i] = o == rank ? locality_key(x + i * K.d, K) : ((1ull << 63) | ((uint64_t)(uint32_t)o << 32) | (uint64_t)i);
  ids[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void k_owned_flag(const int32_t *__restrict__ owner, int64_t n, int rank, int32_t *__restrict__ f) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) f[i] = owner[i] == rank ? 1 : 0;
}
// window sort of the owned rows only; the ghosts keep their order behind them
__global__ __launch_bounds__(256) void k_window_keys_part(const int32_t *__restrict__ len, int64_t n, int64_t n_owned, int window,
                                                          int lmax, uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t nwin = (uint64_t)((n_owned + window - 1) / window + 1);
  keys[r] = r < n_owned ? (uint64_t)(r / window) * (uint64_t)(lmax + 1) + (uint64_t)(lmax - len[r])
                        : nwin * (uint64_t)(lmax + 1) + (uint64_t)(r - n_owned);
  ids[r] = (int32_t)r;
}

static int space_create_impl(const ox_mesh *M, int degree, int window, const int32_t *owner, int rank, int64_t n_initial,
                             int64_t n_dofs_whole, ox_space **out, int brick = 0) {
  if (!M || !out) OX_FAIL("ox_space_create: null argument");
  if (degree < 1 || degree > 3) OX_FAIL("ox_space_create: Lagrange degree %d (1, 2 and -- on triangles -- 3 are built)", degree);
  if (window < SLICE) window = 4096;
  hipStream_t st = nullptr;
  ox_space *V = new ox_space();
  struct Guard {
    ox_space *v;
    ~Guard() { delete v; }
  } guard{V};
  const int d = M->gdim, nv = d + 1, ne = degree >= 2 ? (d == 2 ? 3 : 6) : 0, nd = degree == 3 ? (d == 2 ? 10 : 20) : nv + ne;
  const int64_t nc = M->nc;
  V->mesh = M, V->degree = degree, V->nd = nd, V->window = window;
  V->pw = nd <= 4 ? 4 : (nd <= 8 ? 8 : (nd <= 16 ? 16 : 32));
  // ---- 1. initial dof ids: vertices, then edges numbered by ascending (min, max) vertex pair -------
  DevBuf cd0;
  OX_TRY(cd0.alloc(sizeof(int32_t) * (size_t)nc * nd));
  {
    DevBuf cell_edges;
    if (degree >= 2) {
      const int64_t nk = nc * ne;
      DevBuf k_in, k_out, v_in, v_out, head, excl;
      OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)nk));
      OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)nk));
      OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)nk));
      OX_TRY(v_out.alloc(sizeof(int32_t) * (size_t)nk));
      hipLaunchKernelGGL(k_edge_keys, dim3(nblk(nk)), dim3(256), 0, st, M->cells.as<int32_t>(), nc, d, M->nv, k_in.as<uint64_t>(),
                         v_in.as<int32_t>());
      OX_LAUNCH_CHECK();
      OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), v_out.as<int32_t>(), (size_t)nk,
                        bits_for((uint64_t)M->nv * (uint64_t)M->nv), st));
      k_in.release();
      v_in.release();
      OX_TRY(head.alloc(sizeof(int64_t) * (size_t)nk));
      OX_TRY(excl.alloc(sizeof(int64_t) * (size_t)nk));
      hipLaunchKernelGGL(k_heads, dim3(nblk(nk)), dim3(256), 0, st, k_out.as<uint64_t>(), nk, head.as<int64_t>());
      OX_LAUNCH_CHECK();
      OX_TRY(exclusive_scan_i64(head.as<int64_t>(), excl.as<int64_t>(), (size_t)nk, st));
      int64_t last[2];
      OX_HIP(hipMemcpy(&last[0], excl.as<int64_t>() + nk - 1, sizeof(int64_t), hipMemcpyDeviceToHost));
      OX_HIP(hipMemcpy(&last[1], head.as<int64_t>() + nk - 1, sizeof(int64_t), hipMemcpyDeviceToHost));
      V->n_edges = last[0] + last[1];
      OX_TRY(V->edge_keys.alloc(sizeof(uint64_t) * (size_t)V->n_edges));
      OX_TRY(cell_edges.alloc(sizeof(int32_t) * (size_t)nk));
      hipLaunchKernelGGL(k_edge_scatter, dim3(nblk(nk)), dim3(256), 0, st, k_out.as<uint64_t>(), v_out.as<int32_t>(),
                         head.as<int64_t>(), excl.as<int64_t>(), nk, cell_edges.as<int32_t>(), V->edge_keys.as<uint64_t>());
      OX_LAUNCH_CHECK();
      OX_HIP(hipStreamSynchronize(st));
    }
    DevBuf cell_faces;
    if (degree == 3 && d == 3) {  // faces of the tetrahedra, like the edges: sorted keys, heads, scan, scatter
      if ((long double)M->nv * M->nv * M->nv >= 9.0e18L) OX_FAIL("ox_space_create: face keys of %lld vertices overflow 64 bits", (long long)M->nv);
      const int64_t nk = nc * 4;
      if (nk >= ((int64_t)1 << 31)) OX_FAIL("ox_space_create: %lld cell faces exceed int32", (long long)nk);
      DevBuf k_in, k_out, v_in, v_out, head, excl;
      OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)nk));
      OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)nk));
      OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)nk));
      OX_TRY(v_out.alloc(sizeof(int32_t) * (size_t)nk));
      hipLaunchKernelGGL(k_face_keys, dim3(nblk(nk)), dim3(256), 0, st, M->cells.as<int32_t>(), nc, M->nv, k_in.as<uint64_t>(),
                         v_in.as<int32_t>());
      OX_LAUNCH_CHECK();
      OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), v_out.as<int32_t>(), (size_t)nk, 64, st));
      k_in.release();
      v_in.release();
      OX_TRY(head.alloc(sizeof(int64_t) * (size_t)nk));
      OX_TRY(excl.alloc(sizeof(int64_t) * (size_t)nk));
      hipLaunchKernelGGL(k_heads, dim3(nblk(nk)), dim3(256), 0, st, k_out.as<uint64_t>(), nk, head.as<int64_t>());
      OX_LAUNCH_CHECK();
      OX_TRY(exclusive_scan_i64(head.as<int64_t>(), excl.as<int64_t>(), (size_t)nk, st));
      int64_t last[2];
      OX_HIP(hipMemcpy(&last[0], excl.as<int64_t>() + nk - 1, sizeof(int64_t), hipMemcpyDeviceToHost));
      OX_HIP(hipMemcpy(&last[1], head.as<int64_t>() + nk - 1, sizeof(int64_t), hipMemcpyDeviceToHost));
      V->n_faces = last[0] + last[1];
      OX_TRY(V->face_keys.alloc(sizeof(uint64_t) * (size_t)V->n_faces));
      OX_TRY(cell_faces.alloc(sizeof(int32_t) * (size_t)nk));
      hipLaunchKernelGGL(k_edge_scatter, dim3(nblk(nk)), dim3(256), 0, st, k_out.as<uint64_t>(), v_out.as<int32_t>(),
                         head.as<int64_t>(), excl.as<int64_t>(), nk, cell_faces.as<int32_t>(), V->face_keys.as<uint64_t>());
      OX_LAUNCH_CHECK();
      OX_HIP(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_cd0, dim3(nblk(nc)), dim3(256), 0, st, M->cells.as<int32_t>(), cell_edges.as<int32_t>(), nc, d, degree,
                       M->nv, V->n_edges, cd0.as<int32_t>(), cell_faces.as<int32_t>(), M->cell_perm.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_HIP(hipStreamSynchronize(st));
  }
  const int64_t n = degree == 3 ? M->nv + 2 * V->n_edges + (d == 3 ? V->n_faces : nc) : M->nv + V->n_edges;
  if (n >= ((int64_t)1 << 31) - 64) OX_FAIL("ox_space_create: %lld dofs exceed int32", (long long)n);
  V->n = n;
  if (owner && n != n_initial)
    OX_FAIL("ox_space_create_part: the mesh part has %lld initial dofs, owners were given for %lld", (long long)n,
            (long long)n_initial);
  int64_t n_owned = n;  // mesh-partitioned space: the dofs this rank owns come first, their rows are the rows
  if (owner) {
    DevBuf flag;
    OX_TRY(flag.alloc(sizeof(int32_t) * (size_t)n));
    hipLaunchKernelGGL(k_owned_flag, dim3(nblk(n)), dim3(256), 0, st, owner, n, rank, flag.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_TRY(reduce_sum_i32(flag.as<int32_t>(), n, &n_owned, st));
  }
  V->n_rows = n_owned;
  // ---- 2. dof coordinates, tiled spatial order ------------------------------------------------------
  DevBuf xL, rank1;
  OX_TRY(xL.alloc(sizeof(double) * (size_t)n * d));
  DevBuf cell_kidx;  // degree 3 on triangles: caller's cell index -> kernel cell index (the cell dofs' coordinates)
  if (degree == 3 && d == 2) {
    OX_TRY(cell_kidx.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nc, 1)));
    hipLaunchKernelGGL(k_invert, dim3(nblk(nc)), dim3(256), 0, st, M->cell_perm.as<int32_t>(), nc, cell_kidx.as<int32_t>());
    OX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_dof_coords, dim3(nblk(n)), dim3(256), 0, st, M->coords.as<double>(), V->edge_keys.as<uint64_t>(), M->nv, n,
                     d, xL.as<double>(), degree, V->n_edges, M->cells.as<int32_t>(), V->face_keys.as<uint64_t>(),
                     cell_kidx.as<int32_t>());
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));  // (cell_kidx is released at the end of this scope's caller: finish the launch first)
  OX_LAUNCH_CHECK();
  OX_TRY(rank1.alloc(sizeof(int32_t) * (size_t)n));
  {
    const KeySpec K = key_spec(M, M->tile_bits, owner ? n_dofs_whole : n, brick != 0);
    DevBuf k_in, k_out, v_in, perm1;
    OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)n));
    OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)n));
    OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)n));
    OX_TRY(perm1.alloc(sizeof(int32_t) * (size_t)n));
    if (owner)
      hipLaunchKernelGGL(k_point_keys_part, dim3(nblk(n)), dim3(256), 0, st, xL.as<double>(), n, K, owner, rank,
                         k_in.as<uint64_t>(), v_in.as<int32_t>());
    else
      hipLaunchKernelGGL(k_point_keys, dim3(nblk(n)), dim3(256), 0, st, xL.as<double>(), n, K, k_in.as<uint64_t>(), v_in.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), perm1.as<int32_t>(), (size_t)n,
                      owner ? 64 : key_end_bit(K), st));
    hipLaunchKernelGGL(k_invert, dim3(nblk(n)), dim3(256), 0, st, perm1.as<int32_t>(), n, rank1.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_HIP(hipStreamSynchronize(st));
  }
  // ---- 3. row lengths in that order (they do not depend on the numbering) ---------------------------
  DevBuf err;
  OX_TRY(err.alloc(sizeof(int)));
  OX_HIP(hipMemset(err.p, 0, sizeof(int)));
  DevBuf len1;
  OX_TRY(len1.alloc(sizeof(int32_t) * (size_t)n));
  OX_HIP(hipMemsetAsync(len1.p, 0, len1.bytes, st));  // (ghost rows of a partitioned space: no row, length 0)
  {
    DevBuf cd1, start1, pair1;
    OX_TRY(cd1.alloc(sizeof(int32_t) * (size_t)nc * nd));
    hipLaunchKernelGGL(k_relabel, dim3(nblk(nc * nd)), dim3(256), 0, st, cd0.as<int32_t>(), rank1.as<int32_t>(), nc * nd,
                       cd1.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_TRY(group_pairs(cd1.as<int32_t>(), nc, nd, n, start1, pair1, st));
    RowArgs A{};
    A.n_rows = n_owned, A.n_cols = n;
    A.start = start1.as<int64_t>(), A.pair = pair1.as<uint32_t>();
    A.nd_r = A.nd_c = nd;
    A.col_dofs = cd1.as<int32_t>();
    A.len = len1.as<int32_t>();
    A.err = err.as<int>();
    hipLaunchKernelGGL(k_rows<0>, dim3(row_grid(n_owned)), dim3(256), 0, st, A);
    OX_LAUNCH_CHECK();
    OX_HIP(hipStreamSynchronize(st));
    OX_TRY(check_row_error(err.as<int>(), "ox_space_create"));
  }
  // ---- 4. inside windows of `window` rows: stably by decreasing row length --------------------------
  int32_t lmax = 0;
  OX_TRY(reduce_max_i32(len1.as<int32_t>(), n, &lmax, st));
  DevBuf perm2, rank2;
  OX_TRY(perm2.alloc(sizeof(int32_t) * (size_t)n));
  OX_TRY(rank2.alloc(sizeof(int32_t) * (size_t)n));
  {
    DevBuf k_in, k_out, v_in;
    OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)n));
    OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)n));
    OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)n));
    if (owner)
      hipLaunchKernelGGL(k_window_keys_part, dim3(nblk(n)), dim3(256), 0, st, len1.as<int32_t>(), n, n_owned, window, (int)lmax,
                         k_in.as<uint64_t>(), v_in.as<int32_t>());
    else
      hipLaunchKernelGGL(k_window_keys, dim3(nblk(n)), dim3(256), 0, st, len1.as<int32_t>(), n, window, (int)lmax, k_in.as<uint64_t>(),
                         v_in.as<int32_t>());
    OX_LAUNCH_CHECK();
    const uint64_t kmax = (uint64_t)((n + window - 1) / window + 3) * (uint64_t)(lmax + 1) + (uint64_t)(n - n_owned);
    OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), perm2.as<int32_t>(), (size_t)n, bits_for(kmax), st));
    hipLaunchKernelGGL(k_invert, dim3(nblk(n)), dim3(256), 0, st, perm2.as<int32_t>(), n, rank2.as<int32_t>());
    OX_LAUNCH_CHECK();
  }
  OX_TRY(V->rank_initial.alloc(sizeof(int32_t) * (size_t)n));
  hipLaunchKernelGGL(k_compose, dim3(nblk(n)), dim3(256), 0, st, rank1.as<int32_t>(), rank2.as<int32_t>(), n,
                     V->rank_initial.as<int32_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(V->cell_dofs.alloc(sizeof(int32_t) * (size_t)nc * nd));
  hipLaunchKernelGGL(k_relabel, dim3(nblk(nc * nd)), dim3(256), 0, st, cd0.as<int32_t>(), V->rank_initial.as<int32_t>(), nc * nd,
                     V->cell_dofs.as<int32_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(V->x.alloc(sizeof(double) * (size_t)n * d));
  hipLaunchKernelGGL(k_permute_points, dim3(nblk(n)), dim3(256), 0, st, xL.as<double>(), V->rank_initial.as<int32_t>(), n, d,
                     V->x.as<double>());
  OX_LAUNCH_CHECK();
  ox_pattern_store &P = V->P;
  P.n_rows = n_owned, P.n_cols = n;
  OX_TRY(P.row_len.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(n_owned, 1)));
  hipLaunchKernelGGL(k_gather_i32, dim3(nblk(std::max<int64_t>(n_owned, 1))), dim3(256), 0, st, len1.as<int32_t>(), perm2.as<int32_t>(),
                     n_owned, P.row_len.as<int32_t>());
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  V->row_pos.p = perm2.detach();  // final row r was row perm2[r] of the locality order: what ox_space_windows clusters by
  V->row_pos.bytes = sizeof(int32_t) * (size_t)n;
  cd0.release(), xL.release(), rank1.release(), rank2.release(), len1.release();
  // ---- 5. final adjacency, SELL layout, columns and position bytes ------------------------------------
  OX_TRY(group_pairs(V->cell_dofs.as<int32_t>(), nc, nd, n, V->start, V->pair, st));
  OX_TRY(layout_slices(P, V->start.as<int64_t>(), &V->adj_ptr, &V->npairs, st));
  OX_TRY(V->adj_cell.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(V->npairs, 1)));
  OX_TRY(V->adj_loc.alloc((size_t)std::max<int64_t>(V->npairs, 1)));
  OX_TRY(V->adj_pos.alloc((size_t)std::max<int64_t>(V->npairs, 1) * V->pw));
  {
    RowArgs A{};
    A.n_rows = n_owned, A.n_cols = n;
    A.start = V->start.as<int64_t>(), A.pair = V->pair.as<uint32_t>();
    A.nd_r = A.nd_c = nd;
    A.col_dofs = V->cell_dofs.as<int32_t>();
    A.slice_ptr = P.slice_ptr.as<int64_t>(), A.cols = P.cols.as<int32_t>();
    A.adj_ptr = V->adj_ptr.as<int64_t>(), A.adj_cell = V->adj_cell.as<int32_t>();
    A.adj_loc = V->adj_loc.as<uint8_t>(), A.adj_pos = V->adj_pos.as<uint8_t>();
    A.pw = V->pw;
    A.err = err.as<int>();
    hipLaunchKernelGGL(k_rows<1>, dim3(row_grid(P.n_slices * SLICE)), dim3(256), 0, st, A);
    OX_LAUNCH_CHECK();
    OX_HIP(hipStreamSynchronize(st));
    OX_TRY(check_row_error(err.as<int>(), "ox_space_create"));
  }
  OX_TRY(finish_pattern(P, st));
  guard.v = nullptr;
  *out = V;
  return 0;
}

extern "C" int ox_space_create(const ox_mesh *M, int degree, int window, ox_space **out) {
  return space_create_impl(M, degree, window, nullptr, 0, 0, 0, out);
}

extern "C" int ox_space_create_ordered(const ox_mesh *M, int degree, int window, int order_flags, ox_space **out) {
  if (order_flags & ~1) OX_FAIL("ox_space_create_ordered: order_flags=%d", order_flags);
  return space_create_impl(M, degree, window, nullptr, 0, 0, 0, out, order_flags & 1);
}

extern "C" int ox_space_create_part(const ox_mesh *M, int degree, int window, const int32_t *owner, int64_t n_initial, int rank,
                                    int64_t n_dofs_whole, ox_space **out) {
  if (!owner) OX_FAIL("ox_space_create_part: null argument");
  if (n_dofs_whole < n_initial) OX_FAIL("ox_space_create_part: n_dofs_whole < n_initial");
  return space_create_impl(M, degree, window, owner, rank, n_initial, n_dofs_whole, out);
}

extern "C" int ox_space_destroy(ox_space *V) {
  delete V;
  return 0;
}

static void fill_pattern_view(const ox_pattern_store &P, ox_pattern_info *v) {
  memset(v, 0, sizeof(*v));
  v->sell.n_rows = P.n_rows, v->sell.n_cols = P.n_cols, v->sell.n_slices = (int32_t)P.n_slices;
  v->sell.slice_ptr = P.slice_ptr.as<int64_t>(), v->sell.cols = P.cols.as<int32_t>();
  v->sell.cols16 = P.cols16.as<uint16_t>(), v->sell.cbase = P.cbase.as<int32_t>();
  v->size = P.size, v->nnz = P.nnz, v->n_compressed = P.n16;
  v->row_len = P.row_len.as<int32_t>();
  v->n_bins = (int32_t)P.bin_width.size();
  v->bin_ptr_host = P.bin_ptr.data(), v->bin_width_host = P.bin_width.data();
  v->bin_slices = P.bin_slices.as<int32_t>();
  v->widths_host = P.widths.data();
  v->n_row_blocks = P.n_row_blocks, v->row_blk_ptr = P.row_blk_ptr.as<int32_t>(), v->row_blk_entries = P.row_blk_entries;
}

extern "C" int ox_space_view(const ox_space *V, ox_space_info *v) {
  if (!V || !v) OX_FAIL("ox_space_view: null argument");
  memset(v, 0, sizeof(*v));
  v->degree = V->degree, v->nd = V->nd, v->pw = V->pw, v->gdim = V->mesh->gdim;
  v->n_dofs = V->n, v->n_edges = V->n_edges, v->n_pairs = V->npairs;
  v->cell_dofs = V->cell_dofs.as<int32_t>();
  v->x = V->x.as<double>();
  v->rank_initial = V->rank_initial.as<int32_t>();
  v->edge_keys = V->edge_keys.as<uint64_t>();
  v->n_faces = V->n_faces, v->face_keys = V->face_keys.as<uint64_t>();
  v->adj.n_slices = (int32_t)V->P.n_slices, v->adj.nd = V->nd;
  v->adj.adj_ptr = V->adj_ptr.as<int64_t>(), v->adj.adj_cell = V->adj_cell.as<int32_t>(), v->adj.adj_loc = V->adj_loc.as<uint8_t>();
  v->adj_pos = V->adj_pos.as<uint8_t>();
  v->pair_start = V->start.as<int64_t>();
  fill_pattern_view(V->P, &v->pattern);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// LDS-window stream of a space's square pattern (ox_sell.wb_* / wlist / wt_ptr / wcode; kernel k_spmv_win, ox_spmv.hip).
// Inside every length-sort window (V->window rows = spw slices, contiguous in storage) the slices are ordered by the
// mean locality position of their rows and cut into blocks of 8: on a box mesh in brick order a block is the rows of
// one brick, whatever their lengths; on any mesh, rows that are neighbours along the ordering curve.  Per block: the
// ascending distinct columns of its entries (the window), a 16-bit index into it for every entry (tile layout: 2
// storage pairs of a lane per 8-byte load) and the wave (0..3) that multiplies each of its slices (longest-processing-
// time schedule).  Everything on the device; the window lists come from radix sorts of (block, column) keys over
// chunks of whole sort windows.
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_win_slice_keys(const int32_t *__restrict__ row_pos, int64_t n_rows, int n_slices, int spw,
                                                        uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int64_t row = (int64_t)slice * 64 + lane;
  long long v = row < n_rows ? (long long)row_pos[row] : 0;
  int cnt = row < n_rows ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) {
    v += __shfl_down(v, off, 64);
    cnt += __shfl_down(cnt, off, 64);
  }
  if (lane == 0) {
    const uint64_t mean64 = cnt > 0 ? (uint64_t)((v * 64) / cnt) : 0;  // < 2^37
    keys[slice] = ((uint64_t)(slice / spw) << 40) | mean64;
    ids[slice] = slice;
  }
}

__global__ __launch_bounds__(256) void k_win_blocks(const int32_t *__restrict__ order, int n_slices, int spw, int bpw, int spb,
                                                    int32_t *__restrict__ wb_slices, int32_t *__restrict__ blk_of_slice) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_slices) return;
  const int slice = order[i];
  const int w = i / spw, j = i - w * spw;  // the windows are contiguous runs of spw slices, in the sorted order too
  const int b = w * bpw + j / spb;
  wb_slices[(size_t)b * 8 + j % spb] = slice;
  blk_of_slice[slice] = b;
}

__global__ __launch_bounds__(256) void k_win_schedule(const int32_t *__restrict__ wb_slices, const int64_t *__restrict__ slice_ptr,
                                                      int n_blocks, uint16_t *__restrict__ wb_waves) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  int w[8], idx[8];
  for (int j = 0; j < 8; ++j) {
    const int s = wb_slices[(size_t)b * 8 + j];
    w[j] = s >= 0 ? (int)((slice_ptr[s + 1] - slice_ptr[s]) >> 7) : 0;
    idx[j] = j;
  }
  for (int a = 1; a < 8; ++a) {  // stable insertion sort by decreasing width
    const int wa = w[a], ia = idx[a];
    int q = a - 1;
    while (q >= 0 && w[q] < wa) {
      w[q + 1] = w[q], idx[q + 1] = idx[q];
      --q;
    }
    w[q + 1] = wa, idx[q + 1] = ia;
  }
  int load[4] = {0, 0, 0, 0};
  unsigned sched = 0;
  for (int a = 0; a < 8; ++a) {  // longest processing time first: the next slice to the least loaded wave
    int m = 0;
    for (int k = 1; k < 4; ++k)
      if (load[k] < load[m]) m = k;
    load[m] += w[a];
    sched |= (unsigned)m << (2 * idx[a]);
  }
  wb_waves[b] = (uint16_t)sched;
}

// (block, column) keys of the slots of slices [s0, s1): key slot i - slot0
__global__ __launch_bounds__(256) void k_win_entry_keys(const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ cols,
                                                        const int32_t *__restrict__ blk_of_slice, int s0, int s1, int64_t slot0,
                                                        uint64_t *__restrict__ keys) {
  const int slice = s0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= s1) return;
  const int64_t base = slice_ptr[slice], end = slice_ptr[slice + 1];
  const uint64_t hi = (uint64_t)(uint32_t)blk_of_slice[slice] << 32;
  for (int64_t e = base + lane; e < end; e += 64) keys[e - slot0] = hi | (uint32_t)cols[e];
}

__global__ __launch_bounds__(256) void k_win_flag_unique(const uint64_t *__restrict__ keys, int64_t n, int64_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_win_compact(const uint64_t *__restrict__ keys, const int64_t *__restrict__ pos, int64_t n,
                                                     uint64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && (i == 0 || keys[i] != keys[i - 1])) out[pos[i]] = keys[i];
}

__global__ __launch_bounds__(256) void k_win_block_ptr(const uint64_t *__restrict__ ukey, int64_t n, int n_blocks,
                                                       int64_t *__restrict__ wb_ptr, int32_t *__restrict__ wsize) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b > n_blocks) return;
  auto lower = [&](uint64_t target) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (ukey[mid] < target) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  const int64_t a = lower((uint64_t)(uint32_t)b << 32);
  wb_ptr[b] = a;
  if (b < n_blocks) wsize[b] = (int32_t)(lower((uint64_t)(uint32_t)(b + 1) << 32) - a);
}
__global__ __launch_bounds__(256) void k_win_list(const uint64_t *__restrict__ ukey, int64_t n, int32_t *__restrict__ wlist) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) wlist[i] = (int32_t)(uint32_t)(ukey[i] & 0xffffffffull);
}

__global__ __launch_bounds__(256) void k_win_ntiles(const int64_t *__restrict__ slice_ptr, int n_slices, int64_t *__restrict__ nt) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n_slices) nt[s] = (((slice_ptr[s + 1] - slice_ptr[s]) >> 7) + 1) >> 1;
  if (s == n_slices) nt[s] = 0;
}

// window index of every entry, written in the tile layout
__global__ __launch_bounds__(256) void k_win_codes(const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ cols, int n_slices,
                                                   const int32_t *__restrict__ blk_of_slice, const int64_t *__restrict__ wb_ptr,
                                                   const int32_t *__restrict__ wlist, const int64_t *__restrict__ wt_ptr,
                                                   uint16_t *__restrict__ wcode, unsigned long long *__restrict__ n_over) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int64_t base = slice_ptr[slice];
  const int npair = (int)((slice_ptr[slice + 1] - base) >> 7);
  const int b = blk_of_slice[slice];
  const int64_t w0 = wb_ptr[b];
  const int W = (int)(wb_ptr[b + 1] - w0);
  const int32_t *__restrict__ wl = wlist + w0;
  const int64_t t0 = wt_ptr[slice];
  const int ntile = (npair + 1) >> 1;
  const bool big = W > 65535;  // a 16-bit index cannot address such a window: codes 0, the kernel multiplies from `cols`
  if (big && lane == 0) atomicAdd(n_over, 1ull);
  for (int t = 0; t < ntile; ++t)
    for (int j = 0; j < 2; ++j) {
      const int k = 2 * t + j;
      uint16_t code[2] = {0, 0};
      if (k < npair && !big) {
        for (int i = 0; i < 2; ++i) {
          const int32_t c = cols[base + (int64_t)k * 128 + lane * 2 + i];
          int lo = 0, hi = W;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (wl[mid] < c) lo = mid + 1;
            else hi = mid;
          }
          code[i] = (uint16_t)lo;
        }
      }
      wcode[((t0 + t) * 64 + lane) * 4 + 2 * j] = code[0];
      wcode[((t0 + t) * 64 + lane) * 4 + 2 * j + 1] = code[1];
    }
}

int build_windows(ox_space *V, hipStream_t st, int split_entries) {
  const ox_pattern_store &P = V->P;
  const int ns = (int)P.n_slices;
  if (ns == 0 || P.size == 0) {
    V->windows_built = true;
    return 0;
  }
  const int spw = std::max(1, V->window / SLICE);
  const int spb = 8;  // slices per window block (4 measured -6 % with three columns, +9 % with one, refined Delaunay mesh, r04)
  const int bpw = (spw + spb - 1) / spb;
  const int nwin = (ns + spw - 1) / spw;
  const int c_last = ns - (nwin - 1) * spw;
  int nb = (nwin - 1) * bpw + (c_last + spb - 1) / spb;
  auto nblk = [](int64_t n) { return (unsigned)((n + 255) / 256); };
  const int64_t *slice_ptr = P.slice_ptr.as<int64_t>();
  const int32_t *cols = P.cols.as<int32_t>();
  // ---- blocks ---------------------------------------------------------------------------------------------------
  DevBuf order, blk_of_slice;
  OX_TRY(order.alloc(sizeof(int32_t) * (size_t)ns));
  OX_TRY(blk_of_slice.alloc(sizeof(int32_t) * (size_t)ns));
  {
    DevBuf k_in, k_out, v_in;
    OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)ns));
    OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)ns));
    OX_TRY(v_in.alloc(sizeof(int32_t) * (size_t)ns));
    hipLaunchKernelGGL(k_win_slice_keys, dim3((ns + 3) / 4), dim3(256), 0, st, V->row_pos.as<int32_t>(), P.n_rows, ns, spw,
                       k_in.as<uint64_t>(), v_in.as<int32_t>());
    OX_LAUNCH_CHECK();
    OX_TRY(sort_pairs(k_in.as<uint64_t>(), k_out.as<uint64_t>(), v_in.as<int32_t>(), order.as<int32_t>(), (size_t)ns,
                      std::min(64, 40 + bits_for((uint64_t)nwin)), st));
  }
  OX_TRY(V->wb_slices.alloc(sizeof(int32_t) * (size_t)nb * 8));
  OX_HIP(hipMemsetAsync(V->wb_slices.p, 0xff, sizeof(int32_t) * (size_t)nb * 8, st));
  hipLaunchKernelGGL(k_win_blocks, dim3(nblk(ns)), dim3(256), 0, st, order.as<int32_t>(), ns, spw, bpw, spb, V->wb_slices.as<int32_t>(),
                     blk_of_slice.as<int32_t>());
  OX_LAUNCH_CHECK();
  order.release();
  // ---- window lists: distinct (block, column) keys, chunks of whole sort windows ------------------------------------
  std::vector<int64_t> sp_h((size_t)ns + 1);
  OX_HIP(hipMemcpyAsync(sp_h.data(), slice_ptr, sizeof(int64_t) * ((size_t)ns + 1), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  const int64_t chunk_slots = (int64_t)1 << 28;  // 2 GiB of keys per chunk (+ the sort's double buffer)
  DevBuf ukey, wsize;
  int64_t total = 0;
  // the sorted distinct (block, column) keys of the current block assignment, the blocks' offsets into them and sizes
  auto make_lists = [&](int nb_) -> int {
    std::vector<std::unique_ptr<DevBuf>> parts;
    std::vector<int64_t> part_n;
    total = 0;
    for (int w0 = 0; w0 < nwin;) {
      int w1 = w0 + 1;
      const int s0 = w0 * spw;
      while (w1 < nwin && sp_h[(size_t)std::min(ns, (w1 + 1) * spw)] - sp_h[(size_t)s0] <= chunk_slots) ++w1;
      const int s1 = std::min(ns, w1 * spw);
      const int64_t slot0 = sp_h[(size_t)s0], m = sp_h[(size_t)s1] - slot0;
      if (m > 0) {
        DevBuf k_in, k_out, flag, pos;
        OX_TRY(k_in.alloc(sizeof(uint64_t) * (size_t)m));
        OX_TRY(k_out.alloc(sizeof(uint64_t) * (size_t)m));
        hipLaunchKernelGGL(k_win_entry_keys, dim3((s1 - s0 + 3) / 4), dim3(256), 0, st, slice_ptr, cols, blk_of_slice.as<int32_t>(), s0,
                           s1, slot0, k_in.as<uint64_t>());
        OX_LAUNCH_CHECK();
        {
          size_t tb = 0;
          const int end_bit = std::min(64, 32 + bits_for((uint64_t)nb_));
          OX_HIP(rocprim::radix_sort_keys(nullptr, tb, k_in.as<uint64_t>(), k_out.as<uint64_t>(), (size_t)m, 0, end_bit, st));
          DevBuf tmp;
          OX_TRY(tmp.alloc(tb));
          OX_HIP(rocprim::radix_sort_keys(tmp.p, tb, k_in.as<uint64_t>(), k_out.as<uint64_t>(), (size_t)m, 0, end_bit, st));
          OX_HIP(hipStreamSynchronize(st));
        }
        k_in.release();
        OX_TRY(flag.alloc(sizeof(int64_t) * ((size_t)m + 1)));
        OX_TRY(pos.alloc(sizeof(int64_t) * ((size_t)m + 1)));
        hipLaunchKernelGGL(k_win_flag_unique, dim3(nblk(m)), dim3(256), 0, st, k_out.as<uint64_t>(), m, flag.as<int64_t>());
        OX_LAUNCH_CHECK();
        OX_HIP(hipMemsetAsync(flag.as<int64_t>() + m, 0, sizeof(int64_t), st));
        OX_TRY(exclusive_scan_i64(flag.as<int64_t>(), pos.as<int64_t>(), (size_t)m + 1, st));
        int64_t nu = 0;
        OX_HIP(hipMemcpyAsync(&nu, pos.as<int64_t>() + m, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        OX_HIP(hipStreamSynchronize(st));
        flag.release();
        parts.emplace_back(new DevBuf());
        OX_TRY(parts.back()->alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(nu, 1)));
        hipLaunchKernelGGL(k_win_compact, dim3(nblk(m)), dim3(256), 0, st, k_out.as<uint64_t>(), pos.as<int64_t>(), m,
                           parts.back()->as<uint64_t>());
        OX_LAUNCH_CHECK();
        OX_HIP(hipStreamSynchronize(st));
        part_n.push_back(nu);
        total += nu;
      }
      w0 = w1;
    }
    OX_TRY(ukey.alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(total, 1)));
    int64_t off = 0;
    for (size_t i = 0; i < parts.size(); ++i) {  // chunks are in block order: the concatenation is sorted
      OX_HIP(hipMemcpyAsync(ukey.as<uint64_t>() + off, parts[i]->p, sizeof(uint64_t) * (size_t)part_n[i], hipMemcpyDeviceToDevice, st));
      off += part_n[i];
    }
    OX_HIP(hipStreamSynchronize(st));
    parts.clear();
    OX_TRY(V->wb_ptr.alloc(sizeof(int64_t) * ((size_t)nb_ + 1)));
    OX_TRY(wsize.alloc(sizeof(int32_t) * (size_t)nb_));
    hipLaunchKernelGGL(k_win_block_ptr, dim3(nblk(nb_ + 1)), dim3(256), 0, st, ukey.as<uint64_t>(), total, nb_, V->wb_ptr.as<int64_t>(),
                       wsize.as<int32_t>());
    OX_LAUNCH_CHECK();
    return 0;
  };
  OX_TRY(make_lists(nb));
  // ---- blocks whose window exceeds `split_entries` are cut in two (round 5; 0: never) -------------------------------------
  // A block of 8 slices whose window holds more entries than the LDS budget of a three-column launch (2176) multiplies
  // from the int32 columns inside that launch, at the lane = row speed; its two halves -- 4 slices each, neighbours along
  // the locality curve, one per wave -- mostly fit.  Refined Delaunay mesh, 18.9 M P2 rows (tools/win_bench.py, A/B in
  // one gpurun call): 36 864 -> 59 575 blocks, mean window 3093 -> 2380 entries, three-column mat-vec 1876-1904 -> 1748 us,
  // ONE-column mat-vec (budget 6144: nothing was over it) 1091-1174 -> 1203-1280 us -- so the caller decides per space
  // (the velocity pattern yes, the pressure pattern no).  One level only: a second cut (blocks of 2 slices, half the
  // waves of a 256-thread block idle) measured 1760 / 1360 us.
  for (int round = 0; round < 1 && split_entries > 0; ++round) {
    std::vector<int32_t> ws((size_t)nb), sl((size_t)nb * 8);
    OX_HIP(hipMemcpyAsync(ws.data(), wsize.p, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, st));
    OX_HIP(hipMemcpyAsync(sl.data(), V->wb_slices.p, sizeof(int32_t) * (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    OX_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> sl2, bos((size_t)ns, 0);
    sl2.reserve(sl.size() + sl.size() / 2);
    int nsplit = 0;
    for (int b = 0; b < nb; ++b) {
      int cnt = 0;
      while (cnt < 8 && sl[(size_t)b * 8 + cnt] >= 0) ++cnt;
      const bool split = ws[(size_t)b] > split_entries && cnt > 4;
      const int cut = split ? (cnt + 1) / 2 : cnt;
      for (int part = 0; part < (split ? 2 : 1); ++part) {
        const int j0 = part ? cut : 0, j1 = part ? cnt : cut;
        const int nbk = (int)(sl2.size() / 8);
        for (int j = 0; j < 8; ++j) {
          const int32_t s_ = (j0 + j < j1) ? sl[(size_t)b * 8 + j0 + j] : -1;
          sl2.push_back(s_);
          if (s_ >= 0) bos[(size_t)s_] = nbk;
        }
      }
      nsplit += split ? 1 : 0;
    }
    if (nsplit > 0) {
      nb = (int)(sl2.size() / 8);
      OX_TRY(V->wb_slices.alloc(sizeof(int32_t) * sl2.size()));
      OX_HIP(hipMemcpyAsync(V->wb_slices.p, sl2.data(), sizeof(int32_t) * sl2.size(), hipMemcpyHostToDevice, st));
      OX_HIP(hipMemcpyAsync(blk_of_slice.p, bos.data(), sizeof(int32_t) * (size_t)ns, hipMemcpyHostToDevice, st));
      OX_HIP(hipStreamSynchronize(st));
      OX_TRY(make_lists(nb));
    } else {
      break;
    }
  }
  OX_TRY(V->wb_waves.alloc(sizeof(uint16_t) * (size_t)nb));
  hipLaunchKernelGGL(k_win_schedule, dim3(nblk(nb)), dim3(256), 0, st, V->wb_slices.as<int32_t>(), slice_ptr, nb,
                     V->wb_waves.as<uint16_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(V->wlist.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1)));
  hipLaunchKernelGGL(k_win_list, dim3(nblk(total)), dim3(256), 0, st, ukey.as<uint64_t>(), total, V->wlist.as<int32_t>());
  OX_LAUNCH_CHECK();
  int32_t wmax = 0;
  OX_TRY(reduce_max_i32(wsize.as<int32_t>(), nb, &wmax, st));
  ukey.release(), wsize.release();
  // ---- tile offsets and the 16-bit window index of every entry ----------------------------------------------------------
  DevBuf nt;
  OX_TRY(nt.alloc(sizeof(int64_t) * ((size_t)ns + 1)));
  OX_TRY(V->wt_ptr.alloc(sizeof(int64_t) * ((size_t)ns + 1)));
  hipLaunchKernelGGL(k_win_ntiles, dim3(nblk(ns + 1)), dim3(256), 0, st, slice_ptr, ns, nt.as<int64_t>());
  OX_LAUNCH_CHECK();
  OX_TRY(exclusive_scan_i64(nt.as<int64_t>(), V->wt_ptr.as<int64_t>(), (size_t)ns + 1, st));
  int64_t n_tiles = 0;
  OX_HIP(hipMemcpyAsync(&n_tiles, V->wt_ptr.as<int64_t>() + ns, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  OX_TRY(V->wcode.alloc(sizeof(uint16_t) * (size_t)std::max<int64_t>(n_tiles, 1) * 256));
  DevBuf n_over;
  OX_TRY(n_over.alloc(sizeof(unsigned long long)));
  OX_HIP(hipMemsetAsync(n_over.p, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_win_codes, dim3((ns + 3) / 4), dim3(256), 0, st, slice_ptr, cols, ns, blk_of_slice.as<int32_t>(),
                     V->wb_ptr.as<int64_t>(), V->wlist.as<int32_t>(), V->wt_ptr.as<int64_t>(), V->wcode.as<uint16_t>(),
                     n_over.as<unsigned long long>());
  OX_LAUNCH_CHECK();
  unsigned long long over = 0;
  OX_HIP(hipMemcpyAsync(&over, n_over.p, sizeof(over), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  V->n_wblocks = nb, V->w_max = wmax, V->n_list = total, V->n_tiles = n_tiles, V->n_over_16bit = (int64_t)over;
  V->windows_built = true;
  V->windows_split = split_entries;
  return 0;
}

}  // namespace

extern "C" int ox_space_windows_split(ox_space *V, int split_entries, ox_window_info *v);
extern "C" int ox_space_windows(ox_space *V, ox_window_info *v) { return ox_space_windows_split(V, 0, v); }

extern "C" int ox_space_windows_split(ox_space *V, int split_entries, ox_window_info *v) {
  if (!V || !v) OX_FAIL("ox_space_windows: null argument");
  if (!V->row_pos.p && V->n > 0) OX_FAIL("ox_space_windows: the space carries no locality positions");
  if (split_entries < 0) split_entries = 0;
  if (V->windows_built && V->windows_split != split_entries) V->windows_built = false;  // (another cut: built anew)
  if (!V->windows_built && build_windows(V, nullptr, split_entries)) {
    // a failed build (out of memory in a 2-GiB key chunk, ...) leaves nothing half-built behind: the space stays usable
    // on the lane = row kernels, and a later call starts from scratch
    V->wb_slices.release(), V->wb_waves.release(), V->wb_ptr.release();
    V->wlist.release(), V->wt_ptr.release(), V->wcode.release();
    V->n_wblocks = 0, V->w_max = 0;
    return -1;
  }
  memset(v, 0, sizeof(*v));
  v->n_wblocks = V->n_wblocks, v->w_max = V->w_max;
  v->n_list = V->n_list, v->n_tiles = V->n_tiles, v->n_over_16bit = V->n_over_16bit;
  v->wb_slices = V->wb_slices.as<int32_t>(), v->wb_waves = V->wb_waves.as<uint16_t>();
  v->wb_ptr = V->wb_ptr.as<int64_t>(), v->wlist = V->wlist.as<int32_t>();
  v->wt_ptr = V->wt_ptr.as<int64_t>(), v->wcode = V->wcode.as<uint16_t>();
  return 0;
}

// Rectangular operator (rows = dofs of R, columns = dofs of C, same mesh): the reference's
// create_matrix of the mixed forms (fracstep.py:315,336,352).
extern "C" int ox_rect_create(const ox_space *R, const ox_space *C, ox_rect **out) {
  if (!R || !C || !out) OX_FAIL("ox_rect_create: null argument");
  if (R->mesh != C->mesh) OX_FAIL("ox_rect_create: the two spaces live on different meshes");
  hipStream_t st = nullptr;
  ox_rect *Q = new ox_rect();
  struct Guard {
    ox_rect *q;
    ~Guard() { delete q; }
  } guard{Q};
  Q->R = R, Q->C = C;
  Q->pw = C->nd <= 4 ? 4 : (C->nd <= 8 ? 8 : (C->nd <= 16 ? 16 : 32));
  ox_pattern_store &P = Q->P;
  P.n_rows = R->n_rows, P.n_cols = C->n;  // (a mesh-partitioned row space: its owned dofs)
  DevBuf err;
  OX_TRY(err.alloc(sizeof(int)));
  OX_HIP(hipMemset(err.p, 0, sizeof(int)));
  OX_TRY(P.row_len.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(R->n_rows, 1)));
  RowArgs A{};
  A.n_rows = R->n_rows, A.n_cols = C->n;
  A.start = R->start.as<int64_t>(), A.pair = R->pair.as<uint32_t>();
  A.nd_r = R->nd, A.nd_c = C->nd;
  A.col_dofs = C->cell_dofs.as<int32_t>();
  A.len = P.row_len.as<int32_t>();
  A.err = err.as<int>();
  hipLaunchKernelGGL(k_rows<0>, dim3(row_grid(R->n_rows)), dim3(256), 0, st, A);
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  OX_TRY(check_row_error(err.as<int>(), "ox_rect_create"));
  OX_TRY(layout_slices(P, nullptr, nullptr, nullptr, st));
  OX_TRY(Q->pos.alloc((size_t)std::max<int64_t>(R->npairs, 1) * Q->pw));
  // the pair bookkeeping (cell, local index) is the row space's: written into scratch here
  DevBuf scratch_cell;
  OX_TRY(scratch_cell.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(R->npairs, 1)));
  A.slice_ptr = P.slice_ptr.as<int64_t>(), A.cols = P.cols.as<int32_t>();
  A.adj_ptr = R->adj_ptr.as<int64_t>(), A.adj_cell = scratch_cell.as<int32_t>();
  A.adj_loc = nullptr, A.adj_pos = Q->pos.as<uint8_t>();
  A.pw = Q->pw;
  hipLaunchKernelGGL(k_rows<1>, dim3(row_grid(P.n_slices * SLICE)), dim3(256), 0, st, A);
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  OX_TRY(check_row_error(err.as<int>(), "ox_rect_create"));
  OX_TRY(finish_pattern(P, st));
  guard.q = nullptr;
  *out = Q;
  return 0;
}

extern "C" int ox_rect_destroy(ox_rect *Q) {
  delete Q;
  return 0;
}

extern "C" int ox_rect_view(const ox_rect *Q, ox_rect_info *v) {
  if (!Q || !v) OX_FAIL("ox_rect_view: null argument");
  memset(v, 0, sizeof(*v));
  v->pw = Q->pw;
  v->pos = Q->pos.as<uint8_t>();
  fill_pattern_view(Q->P, &v->pattern);
  return 0;
}

// ---- value dictionary ("1-byte value codes") ---------------------------------------
namespace {
constexpr int HSLOTS = 4096;
constexpr unsigned long long HEMPTY = 0xffffffffffffffffull;

__global__ __launch_bounds__(256) void k_dict_collect(const unsigned long long *__restrict__ bits, int64_t n,
                                                      unsigned long long *table, int *count) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (*reinterpret_cast<volatile int *>(count) > 256) return;  // hopeless: more than 256 distinct values
    const unsigned long long v = bits[i];
    unsigned h = (unsigned)((v * 0x9e3779b97f4a7c15ull) >> 52) & (HSLOTS - 1);
    for (int probe = 0; probe < HSLOTS; ++probe) {
      const unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == v) break;
      if (cur == HEMPTY) {
        const unsigned long long old = atomicCAS(&table[h], HEMPTY, v);
        if (old == HEMPTY) {
          atomicAdd(count, 1);
          break;
        }
        if (old == v) break;
      }
      h = (h + 1) & (HSLOTS - 1);
    }
  }
}

template <int NCOMP>
__global__ __launch_bounds__(256) void k_dict_encode(const long long *__restrict__ bits, int64_t n_slots,
                                                     const long long *__restrict__ dict, int nd, void *__restrict__ codes) {
  __shared__ long long d_s[256];
  if ((int)threadIdx.x < nd) d_s[threadIdx.x] = dict[threadIdx.x];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_slots; i += stride) {
    unsigned packed = 0;
    for (int c = 0; c < NCOMP; ++c) {
      const long long v = bits[i * NCOMP + c];
      int lo = 0, hi = nd - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (d_s[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      packed |= (unsigned)lo << (8 * c);
    }
    if (NCOMP == 1) static_cast<uint8_t *>(codes)[i] = (uint8_t)packed;
    else static_cast<uint32_t *>(codes)[i] = packed;
  }
}
}  // namespace

// vals: [n_slots][ncomp] doubles.  If they take at most 256 distinct bit patterns: dict (ascending as
// signed 64-bit integers, n_dict of them) and codes (ncomp == 1: one byte per slot; ncomp 2..3: one
// uint32 per slot with byte c = the code of component c); otherwise *n_dict = 0 and nothing is written.
extern "C" int ox_value_dictionary(const double *vals, int64_t n_slots, int ncomp, void *codes, double *dict, int *n_dict,
                                   void *stream) {
  if (!vals || !codes || !dict || !n_dict) OX_FAIL("ox_value_dictionary: null argument");
  if (ncomp < 1 || ncomp > 3) OX_FAIL("ox_value_dictionary: ncomp=%d", ncomp);
  hipStream_t st = ox_stream(stream);
  *n_dict = 0;
  const int64_t n = n_slots * ncomp;
  if (n <= 0) return 0;
  DevBuf table, count;
  OX_TRY(table.alloc(sizeof(unsigned long long) * HSLOTS));
  OX_TRY(count.alloc(sizeof(int)));
  OX_HIP(hipMemsetAsync(table.p, 0xff, table.bytes, st));
  OX_HIP(hipMemsetAsync(count.p, 0, sizeof(int), st));
  const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_dict_collect, dim3(nb), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(vals), n,
                     table.as<unsigned long long>(), count.as<int>());
  OX_LAUNCH_CHECK();
  int cnt = 0;
  std::vector<unsigned long long> h(HSLOTS);
  OX_HIP(hipMemcpyAsync(&cnt, count.p, sizeof(int), hipMemcpyDeviceToHost, st));
  OX_HIP(hipMemcpyAsync(h.data(), table.p, table.bytes, hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  if (cnt > 256) return 0;
  std::vector<long long> d;
  for (auto v : h)
    if (v != HEMPTY) d.push_back((long long)v);
  if ((int)d.size() != cnt || d.empty()) return 0;  // (a value with the all-ones pattern: decline)
  std::sort(d.begin(), d.end());
  OX_HIP(hipMemcpyAsync(dict, d.data(), sizeof(long long) * d.size(), hipMemcpyHostToDevice, st));
  const unsigned nb2 = (unsigned)std::min<int64_t>((n_slots + 255) / 256, 8192);
  const long long *bits = reinterpret_cast<const long long *>(vals);
  const long long *dd = reinterpret_cast<const long long *>(dict);
  if (ncomp == 1) hipLaunchKernelGGL(k_dict_encode<1>, dim3(nb2), dim3(256), 0, st, bits, n_slots, dd, cnt, codes);
  else if (ncomp == 2) hipLaunchKernelGGL(k_dict_encode<2>, dim3(nb2), dim3(256), 0, st, bits, n_slots, dd, cnt, codes);
  else hipLaunchKernelGGL(k_dict_encode<3>, dim3(nb2), dim3(256), 0, st, bits, n_slots, dd, cnt, codes);
  OX_LAUNCH_CHECK();
  OX_HIP(hipStreamSynchronize(st));
  *n_dict = cnt;
  return 0;
}

// ---- pair-slot stream (include/oasisx_hip.h, ox_sell.ps_*) -----------------------------------------------
namespace {
__device__ __forceinline__ int ps_wave_min(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int ps_wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
// entry k of lane `lane` of the slice that starts at `base`
__device__ __forceinline__ int64_t ps_slot(int64_t base, int k, int lane) {
  return base + (int64_t)(k / KV) * (SLICE * KV) + lane * KV + (k % KV);
}

// one wave per slice, lane = row: walk the row's stored entries, an entry whose successor is the next
// column shares its slot.  FILL = false: slots of the slice (max over its rows, padded to 4) * 64.
template <bool FILL>
__global__ __launch_bounds__(256) void k_pair_stream(ox_sell A, const int32_t *__restrict__ row_len, int zero_code,
                                                     int64_t *__restrict__ ps_len, int64_t *__restrict__ ps_ptr,
                                                     uint32_t *__restrict__ ps_code, int32_t *__restrict__ ps_base,
                                                     unsigned long long *__restrict__ n_wide) {
  const int lane = threadIdx.x & 63;
  for (int64_t slice = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); slice < A.n_slices; slice += (int64_t)gridDim.x * 4) {
    const int64_t row = slice * SLICE + lane;
    const int len = row < A.n_rows ? row_len[row] : 0;
    const int64_t base = A.slice_ptr[slice];
    if (!FILL) {
      int slots = 0;
      for (int k = 0; k < len; ++slots) {
        const int c = A.cols[ps_slot(base, k, lane)];
        k += (k + 1 < len && A.cols[ps_slot(base, k + 1, lane)] == c + 1) ? 2 : 1;
      }
      const int m = ps_wave_max(slots);
      if (lane == 0) ps_len[slice] = (int64_t)((m + 3) & ~3) * SLICE;
      continue;
    }
    const int64_t pb = ps_ptr[slice] & ~(int64_t)255;  // (a neighbour may already carry its flag bit)
    const int ngroups = (int)(((ps_ptr[slice + 1] & ~(int64_t)255) - pb) >> 8);
    const int BIG = 0x7fffffff;
    // A slot reads x[col] AND x[col+1], the second possibly with a zero coefficient: it must be a finite,
    // settled number.  Columns >= n_rows of a partitioned operator are ghosts -- stale (or NaN) while the
    // halo exchange that overlaps the interior slices is in flight -- so a zero-coefficient read never
    // crosses from the owned block into the ghost block, nor past the last column.
    const int64_t n_own = min(A.n_rows, A.n_cols);
    const int pad_col = (int)min(max(row, (int64_t)0), n_own - 2);  // x[pad_col], x[pad_col+1]: owned
    bool ok = true;
    int k = 0, used = 0;  // used: slots of this row that hold an entry
    for (int g = 0; g < ngroups; ++g) {
      int c[4];
      unsigned vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k < len) {
          ++used;
          int cc = A.cols[ps_slot(base, k, lane)];
          unsigned a = A.vcode[ps_slot(base, k, lane)], b = (unsigned)zero_code;
          if (k + 1 < len && A.cols[ps_slot(base, k + 1, lane)] == cc + 1) {
            b = A.vcode[ps_slot(base, k + 1, lane)];
            k += 2;
          } else {
            k += 1;
            if (cc + 1 == A.n_cols || cc + 1 == n_own) {  // x[cc+1] is missing or a ghost: start one column earlier, a = 0
              cc -= 1;
              b = a;
              a = (unsigned)zero_code;
            }
          }
          c[j] = cc;
          vv[j] = a | (b << 8);
        } else {
          c[j] = pad_col;
          vv[j] = (unsigned)zero_code | ((unsigned)zero_code << 8);
        }
      }
      const int lo = ps_wave_min(min(min(c[0], c[1]), min(c[2], c[3])));
      bool far[4];
      int mn = BIG, mx = -1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        far[j] = c[j] - lo >= 32768;
        if (far[j]) {
          mn = min(mn, c[j]);
          mx = max(mx, c[j]);
        }
      }
      const int lo2 = ps_wave_min(mn), hi2 = ps_wave_max(mx);
      const bool fits = (lo2 == BIG) || (hi2 - lo2 < 32768);
      ok = ok && fits;
      uint4 out;
      unsigned *o = reinterpret_cast<unsigned *>(&out);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned off = !fits ? 0u : (far[j] ? (0x8000u | (unsigned)(c[j] - lo2)) : (unsigned)(c[j] - lo));
        o[j] = off | (vv[j] << 16);
      }
      reinterpret_cast<uint4 *>(ps_code + pb)[(int64_t)g * SLICE + lane] = out;
      if (lane == 0) {
        ps_base[((pb >> 8) + g) * 2] = lo;
        ps_base[((pb >> 8) + g) * 2 + 1] = lo2 == BIG ? lo : lo2;
      }
    }
    // slots of the LAST group that any row of the slice uses (1..4), in bits 1-3 of the offset: the SpMV issues
    // gathers for those only (a slice of interior P1 rows needs 9 slots and stores 12)
    const int m = ps_wave_max(used);
    const int last = ngroups > 0 ? min(4, max(1, m - 4 * (ngroups - 1))) : 0;
    if (lane == 0 && ngroups > 0) {  // (with the offset: the kernel learns both without another round trip)
      ps_ptr[slice] = pb | (ok ? 0 : 1) | ((int64_t)last << 1);
      if (!ok) atomicAdd(n_wide, 1ull);
    }
  }
}

// code of +0.0 in the matrix's dictionary, or -1
int ps_zero_code(const ox_sell *A, hipStream_t st) {
  if (!A->vdict || A->n_dict < 1 || A->n_dict > 256) return -1;
  long long h[256];
  if (hipMemcpyAsync(h, A->vdict, sizeof(long long) * A->n_dict, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  for (int i = 0; i < A->n_dict; ++i)
    if (h[i] == 0) return i;
  return -1;
}
}  // namespace

extern "C" int ox_pair_stream_size(const ox_sell *A, const int32_t *row_len, int64_t *ps_ptr, int64_t *n_codes,
                                   void *stream) {
  if (!A || !row_len || !ps_ptr || !n_codes) OX_FAIL("ox_pair_stream_size: null argument");
  hipStream_t st = ox_stream(stream);
  *n_codes = 0;
  if (!A->vcode || !A->cols || A->n_slices <= 0 || A->n_cols < 2 || A->n_rows < 2) return 0;
  const int zc = ps_zero_code(A, st);
  if (zc < 0) return 0;  // no 0.0 in the dictionary: singles cannot be expressed
  DevBuf len;
  OX_TRY(len.alloc(sizeof(int64_t) * ((size_t)A->n_slices + 1)));
  OX_HIP(hipMemsetAsync(len.p, 0, len.bytes, st));
  const unsigned nb = (unsigned)std::min<int64_t>(((int64_t)A->n_slices + 3) / 4, 1 << 20);
  hipLaunchKernelGGL(k_pair_stream<false>, dim3(nb), dim3(256), 0, st, *A, row_len, zc, len.as<int64_t>(), nullptr, nullptr,
                     nullptr, nullptr);
  OX_LAUNCH_CHECK();
  OX_TRY(exclusive_scan_i64(len.as<int64_t>(), ps_ptr, (size_t)A->n_slices + 1, st));
  int64_t total = 0;
  OX_HIP(hipMemcpyAsync(&total, ps_ptr + A->n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  *n_codes = total;
  return 0;
}

extern "C" int ox_pair_stream_fill(const ox_sell *A, const int32_t *row_len, int64_t *ps_ptr, uint32_t *ps_code,
                                   int32_t *ps_base, int64_t *n_wide, void *stream) {
  if (!A || !row_len || !ps_ptr || !ps_code || !ps_base) OX_FAIL("ox_pair_stream_fill: null argument");
  hipStream_t st = ox_stream(stream);
  if (n_wide) *n_wide = 0;
  if (!A->vcode || !A->cols || A->n_slices <= 0 || A->n_cols < 2 || A->n_rows < 2)
    OX_FAIL("ox_pair_stream_fill: matrix has no value codes");
  const int zc = ps_zero_code(A, st);
  if (zc < 0) OX_FAIL("ox_pair_stream_fill: no 0.0 in the value dictionary");
  DevBuf cnt;
  OX_TRY(cnt.alloc(sizeof(unsigned long long)));
  OX_HIP(hipMemsetAsync(cnt.p, 0, cnt.bytes, st));
  const unsigned nb = (unsigned)std::min<int64_t>(((int64_t)A->n_slices + 3) / 4, 1 << 20);
  hipLaunchKernelGGL(k_pair_stream<true>, dim3(nb), dim3(256), 0, st, *A, row_len, zc, nullptr, ps_ptr, ps_code, ps_base,
                     cnt.as<unsigned long long>());
  OX_LAUNCH_CHECK();
  unsigned long long h = 0;
  OX_HIP(hipMemcpyAsync(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  if (n_wide) *n_wide = (int64_t)h;
  return 0;
}

// ---- plain device memory for callers without a device array library (numpy + ctypes) ----------------
extern "C" int ox_malloc(size_t bytes, void **out) {
  if (!out) OX_FAIL("ox_malloc: null argument");
  *out = nullptr;
  if (bytes == 0) return 0;
  OX_HIP(hipMalloc(out, bytes));
  return 0;
}
extern "C" int ox_free(void *p) {
  if (p) OX_HIP(hipFree(p));
  return 0;
}
extern "C" int ox_memset(void *p, int value, size_t bytes, void *stream) {
  if (bytes) OX_HIP(hipMemsetAsync(p, value, bytes, ox_stream(stream)));
  return 0;
}
extern "C" int ox_synchronize(void *stream) {
  OX_HIP(hipStreamSynchronize(ox_stream(stream)));
  return 0;
}

/*
 * oasisx_hip.h -- C ABI of the MI355X (gfx950) implementation of the per-time-step
 * IPCS hot path of oasisx's FractionalStep_AB_CN.
 *
 * The reference (ComputationalPhysiology/oasisx, pure Python) has no FFI of its
 * own: on this path it calls DOLFINx (assembly, set_bc) and PETSc (Mat/Vec/KSP).
 * Every entry point below replaces one such call site; the file:line cited is
 * the reference call it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (text via ox_last_error());
 *   - all pointers marked "device" are HIP device pointers owned by the caller
 *     (the Python host allocates them with torch); the library owns no field data;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     stream-ordered and only ox_ksp_solve blocks (it returns host-visible results);
 *   - all floating point data is float64, all indices int32, offsets int64;
 *   - multi-component vectors are interleaved: v[row*ncomp + comp];
 *   - process model: ONE process per GPU and one call at a time per HOST THREAD.  The library keeps
 *     per-thread scratch (the reduction scratch of ox_dot / ox_remove_mean, the pinned copy of the
 *     Krylov state of ox_ksp_solve, the Jacobi-diagonal hook of the merged CG epilogue): two solves in flight from
 *     ONE thread on different streams would share it; host threads that each drive their own handles (e.g. the ranks
 *     of a partitioned job rehearsed inside one process, tests/test_gpu_threads_rehearsal.py) do not.  The profiler's
 *     event list (ox_profile_*) is per process: profile from one thread.  The product runs one rank per process, the
 *     reference's model (one PETSc solve at a time per MPI rank);
 *   - an x handed to ox_spmv / ox_ksp_solve on a matrix with a pair-slot stream (ps_*) must be FINITE in
 *     every owned and ghost column: a slot multiplies two adjacent columns and its second half may be the
 *     code of 0.0 (0 * Inf = NaN where the entry streams would not touch that column; -0.0 sums can come
 *     out as +0.0).  With finite x the results are bit-identical to the entry streams.
 *
 * Matrix layout: SELL-64 ("sliced ELLPACK", one slice = the 64 rows one CDNA
 * wavefront owns, lane = row).  Inside slice s with width w_s (multiple of OX_KV)
 * entry k of lane l sits at
 *      slice_ptr[s] + (k / OX_KV) * 64 * OX_KV + l * OX_KV + (k % OX_KV)
 * so that a wave reads 64 * 16 B of values with one dwordx4 load per lane.
 * The dof numbering of a function space IS the SELL row order (rows are grouped
 * by length inside windows at setup), so no row permutation is applied.
 */
#ifndef OASISX_HIP_H
#define OASISX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OX_KV 2          /* entries of one row stored contiguously (16-B value loads) */
#define OX_ROW_BLOCK_WAVES 8              /* slices (= waves) per row block of the one-launch assembly kernels        */
#define OX_ROW_BLOCK_LDS (134 * 1024)     /* bytes of LDS accumulators per row block: 4 slices of 65-entry rows (the
                                             vertex rows of a P2 box mesh), rows of up to 268 entries               */
#define OX_SLICE 64      /* rows per slice = wavefront width */

/* KSPConvergedReason values the step functions return (reference ksp.py:78,
 * fracstep.py:681,684 assert > 0). */
#define OX_CONVERGED_RTOL 2
#define OX_CONVERGED_ATOL 3
#define OX_CONVERGED_ITS 4
#define OX_DIVERGED_ITS (-3)
#define OX_DIVERGED_DTOL (-4)
#define OX_DIVERGED_BREAKDOWN (-5)
#define OX_DIVERGED_NANORINF (-9)

#define OX_KSP_CG 1      /* PETSc "cg"   */
#define OX_KSP_BCGS 2    /* PETSc "bcgs" */
#define OX_KSP_CG_SINGLE 3 /* PETSc "cg" with -ksp_cg_single_reduction: Chronopoulos-Gear recurrences, ONE
                              merged reduction (one all-reduce in partitioned runs) per iteration */
#define OX_KSP_BCGS_MERGED 4 /* "bcgs" with merged reductions: TWO synchronisation points (all-reduces) per
                              iteration instead of three -- rhat.v behind the first mat-vec; t.t, t.s, rhat.s,
                              rhat.t, s.s behind the second give omega, rho and |r| together (|r| by the
                              recurrence |s - omega t|^2, as PETSc's pipelined / improved BiCGStab variants do).
                              Same Krylov space; iteration counts within +-2 of OX_KSP_BCGS.  Inside the loop the
                              convergence test reads the RECURRENCE norm sqrt(max(0, s.s - 2 omega t.s + omega^2 t.t)),
                              which near 1e-10..1e-12 relative can sit below the norm of the stored residual; once it
                              reports convergence the solver sums r.r of the STORED residual (one extra reduction per
                              solve), takes reason and result->rnorm from that -- the test PETSc's KSPBCGS makes
                              (reference ksp.py:76) -- and, where it fails, re-seeds the shadow residual and carries on */

#define OX_KSP_CG_MERGED 5 /* "cg", one right-hand side, ONE synchronisation point and THREE kernels per iteration instead of
                             two and five: the mat-vec q = A p also sums p.q and q.D^-1 q, from which alpha and -- with
                             r' = r - alpha q:  r'.z' = r.z - 2 alpha q.z + alpha^2 q.D^-1 q,  q.z = p.q - beta_old p.q_old (z =
                             D^-1 r = p - beta_old p_old, A symmetric), plain algebra without any orthogonality assumption --
                             beta follow together; one kernel then updates x, r and p and sums the TRUE r.z, |z|^2 and p.q_old
                             for the next point (the carried r.z is replaced every iteration: no drift; the convergence test
                             sees the true |D^-1 r|, one point later: one surplus mat-vec per solve).  Same Krylov space;
                             iteration counts equal to OX_KSP_CG's up to rounding.  With ncomp > 1 the call runs OX_KSP_CG */

/* SELL-64 matrix: pattern + one value array (PETSc Mat on this path). */
typedef struct {
  int64_t n_rows;            /* rows owned by this rank                                  */
  int64_t n_cols;            /* local columns (owned + ghost)                            */
  int32_t n_slices;          /* ceil(n_rows / 64)                                        */
  int32_t n_dict;            /* entries of vdict (0: no value dictionary)                   */
  const int64_t *slice_ptr;  /* device [n_slices+1], entry offsets, multiples of 64*OX_KV */
  const int32_t *cols;       /* device [slice_ptr[n_slices]] (padding: own row, value 0) */
  double *vals;              /* device [slice_ptr[n_slices]]                             */
  /* optional 16-bit column stream (ox_sell_compress_cols); both NULL = int32 columns only */
  const uint16_t *cols16;    /* device [slice_ptr[n_slices]] codes: cols[e] =
                                cbase[e / (64*OX_KV)][code >> 15] + (code & 0x7fff)         */
  const int32_t *cbase;      /* device [slice_ptr[n_slices] / (64*OX_KV)][2]; [0] < 0 at the
                                first pair of a slice: that slice is read from cols         */
  /* optional value dictionary for matrices with <= 256 distinct values (mass / stiffness on
   * meshes of congruent cells): vals[e] == vdict[vcode[e]] bit for bit; used with cols16 */
  const uint8_t *vcode;      /* device [slice_ptr[n_slices]]                                */
  const double *vdict;       /* device [n_dict]                                             */
  /* optional, mesh-partitioned operators: the slices listed with the INTERIOR ones first (no ghost
   * column in any of their rows), then the boundary ones.  With it the distributed mat-vecs start
   * the halo exchange, multiply the interior slices while it is in flight, and finish with the
   * boundary slices (DOLFINx/PETSc do the same inside MatMult; reference fracstep.py:453,497,632). */
  const int32_t *ib_slices;  /* device [n_slices] or NULL                                   */
  int32_t n_interior;        /* leading entries of ib_slices that are interior              */
  int32_t n_wb_interior;     /* LDS-window stream of a mesh-partitioned operator: its window blocks are listed with the
                                INTERIOR ones (no ghost column in any row of any of their slices) first; this many.
                                The overlapped mat-vec multiplies blocks [0, n_wb_interior) while the halo exchange is in
                                flight and the others after it (0 with ib_slices set: no interior block)  */
  /* optional pair-slot stream of a value-dictionary matrix (ox_pair_stream_size / _fill; all NULL =
   * not built).  A slot multiplies TWO adjacent columns: y += vdict[a] * x[col] + vdict[b] * x[col+1],
   * one 16-byte gather instead of two 8-byte ones -- the SpMV on these matrices is bound by the number
   * of vector-memory instructions (16 cycles of the address unit each), not by bytes.
   * Slots keep the stored order of the row's entries (an entry whose successor is the next column
   * shares its slot; any other entry gets b = the code of 0.0), so sums are bit-identical. */
  const int64_t *ps_ptr;     /* device [n_slices+1], offsets into ps_code, multiples of 256;
                                bit 0 set by _fill: that slice is read from cols / vcode;
                                bits 1-3 set by _fill: slots of the slice's last group of 4 that
                                any row uses (the others are padding: no gather is issued)    */
  const uint32_t *ps_code;   /* device [ps_ptr[n_slices]]; slot j of lane l of slice s at
                                ps_ptr[s] + (j/4)*256 + l*4 + j%4: bits 0-14 column offset, bit 15
                                which base, bits 16-23 code a, bits 24-31 code b                */
  const int32_t *ps_base;    /* device [ps_ptr[n_slices] / 256][2] column bases per group of 4
                                slots x 64 lanes                                                */
  /* optional LDS-window stream (ox_window_size / ox_window_fill; all NULL / 0 = not built).  The mat-vecs on the
   * velocity matrices are bound by their gather instructions, not by bytes (every x operand of a lane = row kernel
   * is one gather wave-instruction: DESIGN.md 3).  A WINDOW BLOCK = up to 8 slices (512 rows) whose rows lie close
   * together in the mesh; the sorted list of the distinct columns its entries touch is its window.  The kernel
   * copies x[window] into LDS once (|window| gathers instead of one per entry: 6-9 x fewer) and reads every operand
   * from there through a 16-bit index.  Same entries, same per-row order of fused multiply-adds: bit-identical
   * results.  Blocks whose window exceeds the kernel's LDS budget are multiplied from `cols` as before. */
  const int32_t *wb_slices;  /* device [n_wblocks][8] slice ids of each block (-1: none)                */
  const uint16_t *wb_waves;  /* device [n_wblocks]: 2 bits per slot j = the wave (0..3) that multiplies
                                slice wb_slices[b][j] -- a schedule that balances the slices' widths     */
  const int64_t *wb_ptr;     /* device [n_wblocks+1] offsets into wlist                                  */
  const int32_t *wlist;      /* device [wb_ptr[n_wblocks]] ascending distinct columns of each block      */
  const int64_t *wt_ptr;     /* device [n_slices+1]: first TILE of each slice in wcode / wvcode; a tile = 2 storage
                                pairs (4 entries) of the 64 lanes, lane-contiguous: entry 2j + i (i = 0, 1) of pair
                                2t + j of lane l of slice s at ((wt_ptr[s] + t) * 64 + l) * 4 + 2j + i; a slice of
                                npair pairs has ceil(npair / 2) tiles, the surplus pair of an odd slice's last tile is
                                unused.  One 8-byte (wcode) and one 4-byte (wvcode) load per lane serve 4 entries     */
  const uint16_t *wcode;     /* device [wt_ptr[n_slices] * 256]: cols[e] == wlist[wb_ptr[b] + wcode[tile slot of e]]
                                for every slot e of a slice of block b                                    */
  const uint8_t *wvcode;     /* device [wt_ptr[n_slices] * 256] or NULL: vcode in the tile layout (per matrix, with
                                its value dictionary: ox_window_retile)                                    */
  int32_t n_wblocks;         /* 0: no window stream                                                      */
  int32_t w_max;             /* largest window (entries)                                                 */
  /* per-matrix schedule knobs of the mat-vec (results never depend on them: every storage level multiplies the same
   * entries in the same order).  0 = the library's defaults */
  int32_t levels;            /* storage levels ox_spmv may use on THIS matrix: OX_SPMV_LEVELS(mask) with mask bit 0
                                nontemporal matrix stream, 1 the 16-bit column stream, 2 the 1-byte value codes, 3 the
                                pair-slot stream, 4 the LDS-window stream -- each where the matrix carries it; 0 = all */
  int32_t w_cap;             /* LDS-window stream: window entries the launch holds in LDS (0 = 6144 / 3072 / 2176 for
                                1 / 2 / 3 right-hand sides); a block whose window is larger multiplies from `cols`  */
} ox_sell;
#define OX_SPMV_LEVELS(mask) (32 | ((mask) & 31))

/* Cells of the mesh as the element kernels read them. */
typedef struct {
  int32_t gdim;              /* 2 (triangles) or 3 (tetrahedra)                          */
  int32_t reserved;
  int64_t n_cells;
  const double *geom;        /* device [n_cells][gs]: grad(lambda_1..d) row-major, |detJ|;
                                gs = 6 (2-D, one pad) or 10 (3-D)                        */
} ox_cells;

/* dof -> cell adjacency of one row space in the same SELL-64 row order:
 * pair (t, lane) of slice s sits at adj_ptr[s] + t*64 + lane. */
typedef struct {
  int32_t n_slices;
  int32_t nd;                /* dofs per cell of the ROW space                           */
  const int64_t *adj_ptr;    /* device [n_slices+1]                                      */
  const int32_t *adj_cell;   /* device [pairs], -1 = padding                             */
  const uint8_t *adj_loc;    /* device [pairs], local index of the row dof in the cell   */
} ox_adj;

/* Halo plan + communicator for mesh-partitioned runs (NULL = single GPU). */
typedef struct ox_dist ox_dist;

typedef struct {
  int32_t reason[4];         /* per component, KSPConvergedReason                        */
  int32_t its[4];            /* per component, iterations taken                          */
  double rnorm[4];           /* per component, final preconditioned residual norm        */
  double bnorm[4];           /* per component, preconditioned norm of b                  */
  int32_t resumed[4];        /* OX_KSP_BCGS_MERGED: times the column was re-opened because its STORED residual failed
                                the test its recurrence norm had passed (0 for every other method)  */
} ox_ksp_result;

/* ---- set-up: mesh -> spaces -> patterns (ox_setup.hip) ----------------------------------
 * What DOLFINx does for the reference behind functionspace() / create_matrix() /
 * create_sparsity_pattern() (reference fracstep.py:187-216,293-300,315,324,336,352).  The objects
 * own their device arrays; the *_view calls hand out pointers that stay valid until the object is
 * destroyed (pointers named *_host are host memory).  A binding needs nothing but these calls and
 * numpy arrays of vertex coordinates and cell->vertex indices. */
typedef struct ox_mesh ox_mesh;
typedef struct ox_space ox_space;
typedef struct ox_rect ox_rect;

typedef struct {
  int32_t gdim, tile_bits;
  int64_t n_vertices, n_cells;
  double lo[3], span[3];     /* bounding box                                                */
  const double *coords;      /* device [n_vertices][gdim]                                   */
  const int32_t *cells;      /* device [n_cells][gdim+1], KERNEL order (tiled by centroid)  */
  const int32_t *cell_perm;  /* device [n_cells]: kernel cell index -> caller's cell index  */
  ox_cells cells_struct;     /* geometry of the cells in kernel order, as the kernels take it */
  int32_t lattice;           /* 1: vertices on a tensor grid (tiled lexicographic order), 0: Z-order curve;
                                with lo / span / tile_bits / n_cells: the frame ox_mesh_create_sub takes  */
  int32_t reserved;
} ox_mesh_info;

typedef struct {
  ox_sell sell;              /* the pattern (vals == NULL: copy the struct and set vals per matrix) */
  int64_t size, nnz;         /* storage slots (padding included) / real entries             */
  int64_t n_compressed;      /* slots read through the 16-bit column stream                 */
  const int32_t *row_len;    /* device [n_rows]                                             */
  int32_t n_bins;            /* width bins of the assembly launches (ox_assemble_matrix)    */
  const int64_t *bin_ptr_host;
  const int32_t *bin_width_host;
  const int32_t *bin_slices; /* device [n_slices]                                           */
  const int32_t *widths_host;/* host [n_slices]                                             */
  /* row blocks of the one-launch assembly (ox_assemble_first_blocks / ox_assemble_matrix_blocks): block b owns the
   * CONSECUTIVE slices [row_blk_ptr[b], row_blk_ptr[b+1]) -- at most OX_ROW_BLOCK_WAVES of them, their storage slots
   * together at most OX_ROW_BLOCK_LDS / 8 (greedy, in storage order); row_blk_entries = the largest block's slots
   * (sizes the launch's LDS).  n_row_blocks = 0: a slice is wider than the budget, use the width bins. */
  int32_t n_row_blocks;
  int32_t reserved_rb;
  const int32_t *row_blk_ptr;/* device [n_row_blocks + 1]                                   */
  int64_t row_blk_entries;
} ox_pattern_info;

typedef struct {
  int32_t degree, nd, pw, gdim; /* dofs per cell; stride of the position bytes              */
  int64_t n_dofs, n_edges, n_pairs;
  const int32_t *cell_dofs;     /* device [n_cells][nd], kernel cell order, final numbering */
  const double *x;              /* device [n_dofs][gdim] dof coordinates                    */
  const int32_t *rank_initial;  /* device [n_dofs]: vertex id (or n_vertices + edge id) -> dof */
  const uint64_t *edge_keys;    /* device [n_edges]: min_vertex * n_vertices + max_vertex, ascending */
  ox_adj adj;                   /* dof -> cell adjacency in slice order                      */
  const uint8_t *adj_pos;       /* device [n_pairs][pw]                                      */
  const int64_t *pair_start;    /* device [n_dofs+1]: cells per dof = pair_start[r+1] - pair_start[r] */
  ox_pattern_info pattern;      /* square operator on the space (M, K, A share it: fracstep.py:293-294) */
  int64_t n_faces;              /* degree 3 on tetrahedra: faces (one dof each: initial id n_vertices + 2 n_edges + face) */
  const uint64_t *face_keys;    /* device [n_faces]: (v0 * n_vertices + v1) * n_vertices + v2 of the sorted triple, ascending */
} ox_space_info;

typedef struct {
  int32_t pw;
  const uint8_t *pos;           /* device [rows' n_pairs][pw]: in-row positions of the column space's cell dofs */
  ox_pattern_info pattern;
} ox_rect_info;

/* LDS-window stream of a space's square pattern: the arrays ox_sell.wb_* / wlist / wt_ptr / wcode point to */
typedef struct {
  int32_t n_wblocks, w_max;     /* window blocks (8 slices each); entries of the largest window           */
  int64_t n_list, n_tiles;      /* entries of wlist; tiles of wcode (256 codes each)                      */
  int64_t n_over_16bit;         /* slices in blocks whose window exceeds 65535 entries (codes 0: read from cols) */
  const int32_t *wb_slices;
  const uint16_t *wb_waves;
  const int64_t *wb_ptr;
  const int32_t *wlist;
  const int64_t *wt_ptr;
  const uint16_t *wcode;
} ox_window_info;

/* coords [n_vertices][gdim] f64, cells [n_cells][gdim+1] int32 (any order, any orientation);
 * on_device: the two pointers are device pointers; tile_bits < 0: default.
 * Cells (and later the dofs) are ordered for the kernels: a lattice mesh (few distinct values per
 * coordinate) by y-z tiles of whole x-lines, any other mesh along a Z-order curve. */
int ox_mesh_create(const double *coords, int64_t n_vertices, const int32_t *cells, int64_t n_cells, int gdim,
                   int on_device, int tile_bits, ox_mesh **out);
int ox_mesh_view(const ox_mesh *mesh, ox_mesh_info *view);
int ox_mesh_destroy(ox_mesh *mesh);
/* scalar Lagrange space of degree 1 or 2 (functionspace(mesh, ("Lagrange", k)), fracstep.py:187-216);
 * window: rows per length-sorting window of the SELL-64 numbering (< 64: the default 4096). */
int ox_space_create(const ox_mesh *mesh, int degree, int window, ox_space **out);
/* The same with a choice of the dof order: order_flags bit 0 = BRICK order on lattice meshes -- (tile, brick_z, brick_y,
 * brick_x, z, y, x) with bricks of about 8 points a side instead of whole x-lines: the rows of 8 consecutive slices
 * then share a compact window of columns, what the LDS-window SpMV (ox_sell.wb_*) needs.  The lane = row kernels
 * gather worse in this order (mass mat-vec 487 -> 630 us at 128^3): only together with the window stream. */
int ox_space_create_ordered(const ox_mesh *mesh, int degree, int window, int order_flags, ox_space **out);
/* One rank's piece of a mesh-partitioned space (what DOLFINx's functionspace() returns on a distributed mesh,
 * reference fracstep.py:186-216): built from that rank's cells only (its own cells + one ghost layer).
 *   ox_mesh_create_sub   the part as a mesh of its own (vertices renumbered 0..n-1 in ascending global id) whose
 *                        ordering keys use the WHOLE mesh's frame -- bounding box lo/span, lattice flag, tile bits,
 *                        cell count -- so every part orders its cells and dofs as the whole mesh would;
 *   ox_space_create_part owner: device [n_initial] owner rank of every initial dof of the part (its vertices, then
 *                        its edges by ascending vertex pair for degree 2).  Dofs owned by `rank` come first (tile
 *                        order, window-sorted by row length: the rows of every pattern of the space), the ghosts
 *                        behind them ordered by (owner, initial id) -- the order in which the owners send them.
 *                        pattern.sell.n_rows = owned dofs, n_cols = n_dofs = owned + ghosts.  n_dofs_whole: dofs of
 *                        the whole space (resolution of the ordering keys). */
int ox_mesh_create_sub(const double *coords, int64_t n_vertices, const int32_t *cells, int64_t n_cells, int gdim,
                       int on_device, const double *lo, const double *span, int lattice, int tile_bits,
                       int64_t n_cells_whole, ox_mesh **out);
int ox_space_create_part(const ox_mesh *mesh, int degree, int window, const int32_t *owner, int64_t n_initial, int rank,
                         int64_t n_dofs_whole, ox_space **out);
int ox_space_view(const ox_space *space, ox_space_info *view);
/* LDS-window stream of the space's square pattern (M, K, A share it), built on the first call and owned by the space:
 * copy the pointers into the ox_sell of every matrix on the pattern (the value codes of a dictionary matrix are
 * re-tiled per matrix: ox_window_retile).  Worth it where the rows of 8 consecutive slices share their columns:
 * any mesh in Z-order, box meshes in brick order (ox_space_create_ordered). */
int ox_space_windows(ox_space *space, ox_window_info *view);
/* The same with blocks whose window holds more than split_entries columns cut in two (4 + 4 slices; 0 = never,
 * ox_space_windows): for patterns whose mat-vecs run with three right-hand sides -- the LDS budget of such a launch is
 * 2176 window entries, a block beyond it multiplies from the int32 columns.  One-column launches (budget 6144) lose
 * ~9 % to the extra lists: the velocity pattern yes, the pressure pattern no.  A call with another split_entries than
 * the stream was built with builds it anew (views handed out before are stale then). */
int ox_space_windows_split(ox_space *space, int split_entries, ox_window_info *view);
int ox_space_destroy(ox_space *space);
/* pattern of a mixed operator, rows = dofs of `rows`, columns = dofs of `cols` (fracstep.py:315,336,352) */
int ox_rect_create(const ox_space *rows, const ox_space *cols, ox_rect **out);
int ox_rect_view(const ox_rect *rect, ox_rect_info *view);
int ox_rect_destroy(ox_rect *rect);
/* Value dictionary of a matrix that will not change any more (M, K, Ap, the rectangular operators):
 * vals [n_slots][ncomp] device doubles.  At most 256 distinct bit patterns: dict (device, room for 256,
 * ascending as signed 64-bit integers), *n_dict of them, and codes (device; ncomp == 1: one byte per
 * slot, ncomp 2..3: one uint32 per slot, byte c = code of component c).  More: *n_dict = 0. */
int ox_value_dictionary(const double *vals, int64_t n_slots, int ncomp, void *codes, double *dict, int *n_dict,
                        void *stream);
/* Pair-slot stream of a matrix that carries a value dictionary (A->vcode, A->vdict, ncomp == 1) whose
 * dictionary contains 0.0.  row_len: device [n_rows] entries per row (ox_pattern_info.row_len).
 * _size writes ps_ptr (device [n_slices+1]) and returns the number of uint32 codes in *n_codes (0: not
 * available for this matrix); _fill writes ps_code [n_codes] and ps_base [n_codes/256][2], marks in
 * ps_ptr (bit 0) the slices whose columns do not fit two 15-bit windows per group -- they keep their
 * int32 columns -- and returns their number in *n_wide.  Then set A->ps_ptr / ps_code / ps_base. */
int ox_pair_stream_size(const ox_sell *A, const int32_t *row_len, int64_t *ps_ptr, int64_t *n_codes, void *stream);
/* Re-tile a per-slot array of A's pattern (elem_bytes = 1: vcode, 2: 16-bit codes) from the pair layout of
 * slice_ptr into the tile layout of wt_ptr (ox_sell.wt_ptr); unused entries of a slice's last tile are zeroed. */
int ox_window_retile(const ox_sell *A, const int64_t *wt_ptr, const void *src, int elem_bytes, void *dst, void *stream);
int ox_pair_stream_fill(const ox_sell *A, const int32_t *row_len, int64_t *ps_ptr, uint32_t *ps_code,
                        int32_t *ps_base, int64_t *n_wide, void *stream);
/* plain device memory for callers without a device array library (numpy + ctypes) */
int ox_malloc(size_t bytes, void **out);
int ox_free(void *p);
int ox_memset(void *p, int value, size_t bytes, void *stream);
int ox_synchronize(void *stream);

/* ---- library -------------------------------------------------------------------- */
int ox_version(void);
const char *ox_last_error(void);
int ox_sell_kv(void);
/* device of the current context: compute units and name (diagnostics for bench.py) */
int ox_device_info(int *n_cu, char *name, int name_len);

/* ---- S1/S2: Mat.mult (reference fracstep.py:452,501,541,615,638,642) -------------- */
/* y[row*ncomp+c] = sum_k A[row,k] * x[col_k*ncomp+c]; ncomp in 1..3. */
int ox_spmv(const ox_sell *A, const double *x, double *y, int ncomp, const ox_dist *dist,
            void *stream);

/* Set-up: 16-bit column stream of a pattern for ox_spmv / ox_ksp_solve (10 instead of 12 bytes
 * per stored entry; results are bit-identical).  cols16: [slice_ptr[n_slices]] uint16, cbase:
 * [slice_ptr[n_slices] / (64*OX_KV)][2] int32, both caller-allocated device arrays that the caller then
 * sets in ox_sell.  n_compressed (host, may be NULL): stored entries now read as 16 bit. */
int ox_sell_compress_cols(const ox_sell *A, uint16_t *cols16, int32_t *cbase, int64_t *n_compressed,
                          void *stream);

/* ---- S2: Mat.mult of the pre-assembled rectangular operators (low_memory_version = False,
 *      reference fracstep.py:499-502, 540-542, 642).  One SELL pattern, gdim values per entry
 *      (A->vals holds [slot][gdim]).
 *      v2s = 0: y[row][d] = base[row][d] + scale * sum_k vals[k][d] * x[col_k]     (P_i ps, G_i dp)
 *      v2s = 1: y[row]    = base[row]    + scale * sum_k sum_d vals[k][d] * x[col_k][d]  (sum_i D_i u_i)
 *      base may be NULL. */
int ox_spmv_multi(int v2s, int gdim, const ox_sell *A, const double *x, const double *base, double scale,
                  double *y, const ox_dist *dist, void *stream);
/* A9: one-off assemble_matrix of those operators (fracstep.py:392-404) into zero-initialised
 * [slot][gdim] values.  family 0: p*v.dx(i)*dx (rows V, cols Q); 1: p.dx(i)*v*dx (rows V, cols Q);
 * 2: u.dx(i)*q*dx (rows Q, cols V).  adj/adj_pos: adjacency of the ROW space with the in-row
 * positions of the COLUMN space's cell dofs. */
int ox_assemble_rect(int family, int row_degree, int col_degree, const ox_cells *cells, const ox_adj *adj,
                     const uint8_t *adj_pos, int pw, const ox_sell *A, void *stream);

/* ---- V1: Vec axpy/copy/scale on .x.array (fracstep.py:432-434,456-458,506,604,622,690-693) */
/* z = a*x + b*y elementwise over n doubles (x, y, z may alias). */
int ox_axpby(int64_t n, double a, const double *x, double b, const double *y, double *z,
             void *stream);
/* V1: out[c] = sum_i x[i*ncomp+c]*y[i*ncomp+c] (Vec.norm/dot, fracstep.py:524); host result. */
int ox_dot(int64_t n_rows, int ncomp, const double *x, const double *y, double *out_host,
           const ox_dist *dist, void *stream);

/* ---- V2: set_bc(vec,[bc]) (reference bcs.py:135-139, fracstep.py:518,550) ---------- */
/* b[dofs[k]*ncomp+comp] = g[k]. */
int ox_set_bc(double *b, const int32_t *dofs, const double *g, int64_t n, int ncomp, int comp,
              void *stream);

/* A11: adding the assembled outlet term to the RHS (reference fracstep.py:461-465):
 * b[rows[k]*ncomp+comp] += scale * y[k]; rows must be unique (no atomics). */
int ox_scatter_add(double *b, const int32_t *rows, const double *y, int64_t n, int ncomp, int comp,
                   double scale, void *stream);

/* ---- S4: Mat.zeroRowsLocal(rows, diag) (fracstep.py:471-472); keeps columns -------- */
int ox_zero_rows(const ox_sell *A, const int32_t *rows, int64_t n, double diag, void *stream);
/* The same, and au[row][c] = diag * u1[row][c] for the zeroed rows (au may be NULL): the identity rows of the
 * product A u1 that ox_assemble_first_au hands to the tentative-velocity solve (fracstep.py:470-472 + :521). */
int ox_zero_rows_au(const ox_sell *A, const int32_t *rows, int64_t n, double diag, double *au,
                    const double *u1, int ncomp, void *stream);
/* DOLFINx assemble_matrix(..., bcs) on the pressure Laplacian (fracstep.py:379):
 * rows AND columns flagged in is_bc[] -> identity. */
int ox_zero_rows_cols(const ox_sell *A, const uint8_t *is_bc, double diag, void *stream);

/* ---- A1/A2/A3: one-off assemble_matrix of mass / stiffness (fracstep.py:373-380) ---- */
/* kind 0: u*v*dx, kind 1: inner(grad u, grad v)*dx.  Square Lagrange space of `degree`
 * on `cells`; cell_dofs [n_cells][nd]; adj_pos [pairs][pw] gives, per (row, cell) pair,
 * the in-row index k of every cell dof.  Overwrites A->vals.  Slices are launched per width
 * bin: bin b covers bin_slices[bin_ptr_host[b] .. bin_ptr_host[b+1]) (device list) whose rows
 * are at most bin_width_host[b] entries wide (sizes the per-wave LDS accumulator). */
int ox_assemble_matrix(int kind, int degree, const ox_cells *cells, const int32_t *cell_dofs,
                       const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                       int n_bins, const int64_t *bin_ptr_host, const int32_t *bin_slices,
                       const int32_t *bin_width_host, void *stream);
/* w[row] = int phi_row dx (body-force vector for constant f: fracstep.py:387-390, and
 * the weights of assemble_scalar(phi*dx): fracstep.py:585-590). */
int ox_assemble_weights(int degree, const ox_cells *cells, const ox_adj *adj, int64_t n_rows,
                        double *w, void *stream);
/* Load vector of a non-constant source: b[row] = int f phi_row dx, the `force * v * dx` of a body force that is a
 * UFL expression (fracstep.py:284-289, assembled once at :387-390) and the `inner(function, v) * dx` of a
 * Projector (function.py:75,110-113).  The caller tabulates: fq[cell][q] = f at the n_q quadrature points of every
 * cell (kernel cell order), wphi[q][i] = w_q phi_i(x_q) on the reference simplex (any rule, any of the n_d basis
 * functions of the space `adj` belongs to); the kernel forms, per row, sum over its (cell, i) pairs in adjacency
 * order of |det J_cell| * sum_q wphi[q][i] fq[cell][q] -- one lane per row, no atomics, fixed order. */
int ox_assemble_load_vector(const ox_cells *cells, const ox_adj *adj, int64_t n_rows, int n_d, int n_q,
                            const double *wphi, const double *fq, double *b, void *stream);

/* ---- A4 + S3 + S1, fused: assemble_first (fracstep.py:432-469) ---------------------- */
/* With uab = 1.5*u1 - 0.5*u2 already formed (ox_axpby), for every row:
 *   C   = assemble_matrix(inner(dot(uab, nabla_grad(u)), v)*dx)            (:435-437)
 *   Ar  = -0.5*C + (1/dt)*M - 0.5*nu*K                                     (:438-442)
 *   b_first[:, i] = Ar @ u1[:, i] + b0[:, i]                               (:449-458)
 *   A   = -Ar + (2/dt)*M                                                   (:468-469)
 * M, K, A share one SELL pattern (fracstep.py:293-294).  Slices are launched per
 * width bin (bin_ptr/bin_slices/bin_width from the host) so each launch sizes its
 * LDS accumulator for the widest row of the bin. */
int ox_assemble_first(int degree, const ox_cells *cells, const int32_t *cell_dofs,
                      const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                      const ox_sell *M, const ox_sell *K, const double *uab,
                      const double *u1, const double *b0, double *b_first, double dt, double nu,
                      int n_bins, const int64_t *bin_ptr_host, const int32_t *bin_slices,
                      const int32_t *bin_width_host, void *stream);

/* The same, and a_u1 (device [n_rows][gdim], may be NULL) = (the assembled A) @ u1 row by row, with the
 * entry order and operations of ox_spmv: where u1 is also the initial guess of the tentative-velocity
 * solve (fracstep.py:521 with -ksp_initial_guess_nonzero) it is that solve's first mat-vec
 * (ox_ksp_solve_ax0), once the rows ox_zero_rows turns into identity rows have been set to u1. */
int ox_assemble_first_au(int degree, const ox_cells *cells, const int32_t *cell_dofs,
                         const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                         const ox_sell *M, const ox_sell *K, const double *uab,
                         const double *u1, const double *b0, double *b_first, double dt, double nu,
                         int n_bins, const int64_t *bin_ptr_host, const int32_t *bin_slices,
                         const int32_t *bin_width_host, void *stream, double *a_u1);

/* The same two in ONE launch over the slices in storage order (round 5): row block b = the consecutive slices
 * [blk_ptr[b], blk_ptr[b+1]), one per wave of a 512-thread block, accumulators for lds_entries storage slots per block
 * (ox_pattern_info.n_row_blocks / row_blk_ptr / row_blk_entries).  The width bins send the slices of one length-sort
 * window -- one compact region of the mesh -- to up to ten launches, each of which fetches that region's cell records
 * and coefficients again (refined Delaunay mesh: 95.9 GB of HBM traffic per call for ~22 GB of streams); in storage
 * order the rows of a cell meet in one L2.  Per slice the same operations in the same order: bit-identical results. */
int ox_assemble_matrix_blocks(int kind, int degree, const ox_cells *cells, const int32_t *cell_dofs,
                              const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A, int n_blocks,
                              const int32_t *blk_ptr, int64_t lds_entries, void *stream);
int ox_assemble_first_blocks(int degree, const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj,
                             const uint8_t *adj_pos, int pw, const ox_sell *A, const ox_sell *M, const ox_sell *K,
                             const double *uab, const double *u1, const double *b0, double *b_first, double dt,
                             double nu, int n_blocks, const int32_t *blk_ptr, int64_t lds_entries, void *stream,
                             double *a_u1);

/* ---- A6 / A8: assemble_vector(p * v.dx(i) * dx) and (dp.dx(i) * v * dx), all i at once
 *      (fracstep.py:487-497 and :618).  kind 0: out[r][i] = base[r][i] + scale * int p d_i(phi_r)
 *      kind 1: out[r][i] = base[r][i] + scale * int d_i(p) phi_r.  base may be NULL (=0). */
int ox_assemble_grad_vector(int kind, int row_degree, int p_degree, const ox_cells *cells,
                            const int32_t *cell_pdofs, const ox_adj *adj, int64_t n_rows,
                            const double *p, const double *base, double scale, double *out,
                            void *stream);
/* ---- A7: assemble_vector(div(u) * q * dx) scaled (fracstep.py:538,546):
 *      out[r] = scale * int div(u) psi_r,  u interleaved [n_u][gdim]. */
int ox_assemble_div_vector(int row_degree, int u_degree, const ox_cells *cells,
                           const int32_t *cell_udofs, const ox_adj *adj, int64_t n_rows,
                           const double *u, double scale, double *out, void *stream);

/* ---- Jacobi preconditioner setup: dinv[row] = 1 / A[row,row] (PCJACOBI) ------------- */
int ox_jacobi_setup(const ox_sell *A, double *dinv, void *stream);

/* ---- K1/K2/K3: KSP.solve (reference ksp.py:71-78; fracstep.py:521,578,634) ---------- */
/* Left-Jacobi-preconditioned CG or BiCGStab on `ncomp` right-hand sides that share A
 * (the velocity components share one matrix, fracstep.py:274,521), run in lockstep with
 * per-component scalars; PETSc conventions: zero initial guess unless nonzero_guess,
 * convergence on ||D^-1 r|| <= max(rtol*||D^-1 b||, atol).  Scalars stay on the device;
 * the host reads the state back every `check_every` iterations. */
size_t ox_ksp_work_bytes(int64_t n_rows, int64_t n_cols, int ncomp, int ksp_type);
int ox_ksp_solve(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x,
                 int ncomp, double rtol, double atol, int max_it, int nonzero_guess,
                 int check_every, int max_restarts, void *work, size_t work_bytes,
                 ox_ksp_result *result, const ox_dist *dist, void *stream);
/* The same with A x0 supplied by the caller (nonzero_guess only; NULL = computed here): the velocity
 * update has M u* at hand from its right-hand side (fracstep.py:615,634), so the solver's first
 * mat-vec is skipped.  ax0: device [n_rows][ncomp], the product with the x passed in. */
int ox_ksp_solve_ax0(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x,
                     int ncomp, double rtol, double atol, int max_it, int nonzero_guess,
                     int check_every, int max_restarts, void *work, size_t work_bytes,
                     ox_ksp_result *result, const ox_dist *dist, void *stream, const double *ax0);
/* The same with a value dictionary of dinv (dinv_code: device [n_rows] bytes, dinv[i] == dinv_dict[dinv_code[i]] bit
 * for bit; ox_value_dictionary builds it where dinv takes <= 256 distinct values, as the diagonals of M and Ap do on
 * meshes of congruent cells): the CG update kernels then read one byte per row instead of eight.  Same iterates. */
int ox_ksp_solve_dc(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x,
                    int ncomp, double rtol, double atol, int max_it, int nonzero_guess,
                    int check_every, int max_restarts, void *work, size_t work_bytes,
                    ox_ksp_result *result, const ox_dist *dist, void *stream, const double *ax0,
                    const uint8_t *dinv_code, const double *dinv_dict, int n_dinv_dict);
/* The same with every option of the solve in one block -- PETSc's KSP options (reference ksp.py:38-53 forwards any key to
 * KSPSetFromOptions) and this library's schedule knobs, per call instead of per process:
 *   divtol       -ksp_divtol: |D^-1 r| >= divtol |D^-1 b| ends the column with OX_DIVERGED_DTOL (KSPConvergedDefault;
 *                PETSc's default 1e4 -- "divergence=10000." in -ksp_view); the older entry points run with that default
 *   fold_blocks  one-column CG on one GPU: blocks of the folded update kernels; 0 = separate synchronisation points
 *                (five kernels per iteration), -1 = ox_ksp_default_fold_blocks()
 *   run_ahead    one-column solves queue one batch ahead of the state the host reads: 1 / 0, -1 = default (1)
 * ox_ksp_options_default fills in PETSc's defaults (rtol 1e-5, atol 1e-50, divtol 1e4, max_it 10000, zero guess). */
typedef struct {
  double rtol, atol, divtol;
  int32_t max_it, nonzero_guess, check_every, max_restarts;
  int32_t fold_blocks, run_ahead;
  const double *ax0;         /* see ox_ksp_solve_ax0                                      */
  const uint8_t *dinv_code;  /* see ox_ksp_solve_dc                                       */
  const double *dinv_dict;
  int32_t n_dinv_dict;
  int32_t reserved;
} ox_ksp_options;
int ox_ksp_options_default(ox_ksp_options *opt);
int ox_ksp_solve_opt(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x, int ncomp,
                     const ox_ksp_options *opt, void *work, size_t work_bytes, ox_ksp_result *result,
                     const ox_dist *dist, void *stream);
/* Workspace of a solve on THIS operator: ox_ksp_work_bytes sizes the partial-sum arrays for the lane = row grid; an
 * operator that carries an LDS-window stream may launch more blocks than that (one per window block).  Callers with the
 * ox_sell at hand use this one; ox_ksp_solve checks against it. */
size_t ox_ksp_work_bytes_for(const ox_sell *A, int ncomp, int ksp_type);
/* max_restarts: BiCGStab only.  0 = PETSc's KSPBCGS: a rho = rhat.r = 0 (or omega = 0) breakdown
 * ends the solve with OX_DIVERGED_BREAKDOWN.  > 0: re-seed the shadow residual (rhat <- r) and
 * continue, at most that many times per component (used when a direct solver was asked for). */

/* ---- Projector into a discontinuous P1 space (reference function.py:13-143 with the "DG" 1 target of its
 *      test_projector.py:26-35).  Dof (cell e, vertex a, component c) at (e * (gdim+1) + a) * ncomp + c, cells in
 *      the order of `cells`.  The mass matrix is block diagonal: its inverse is applied cell by cell in closed
 *      form (the reference asks PETSc for preonly + LU).
 *      ox_dg1_grad_rhs: b = assemble_vector(inner(grad(u), v) * dx), u a Lagrange P1 / P2 field (function.py:74-77,
 *      :110-119), ncomp = gdim;  ox_dg1_mass: out = M^-1 in (inverse != 0: KSP.solve, function.py:132) or M in. */
int ox_dg1_grad_rhs(int u_degree, const ox_cells *cells, const int32_t *cell_dofs, const double *u, double *b,
                    void *stream);
int ox_dg1_mass(int inverse, const ox_cells *cells, int ncomp, const double *in, double *out, void *stream);

/* ---- V3 + A10: nullspace.remove and mean shift (fracstep.py:573-574, 579-591) -------- */
/* m = (sum_{i<n} w[i]*x[i], all ranks) / wsum;  x[i] -= m for i < n_apply (owned + ghost rows);
 * w == NULL -> plain sum (arithmetic mean when wsum = global n). */
int ox_remove_mean(int64_t n, int64_t n_apply, double *x, const double *w, double wsum,
                   const ox_dist *dist, void *stream);

/* ---- measurement: per-kernel HIP-event timing on the launching stream (bench.py) ------- */
/* tags: 10*ncomp+epi for SpMV (epi 0 plain, 1 CG p.q, 2/3 BiCGStab), 100 assemble_first,
 * 110/111 grad vectors, 120 div vector. */
/* One-column CG on one GPU folds the iteration's synchronisation points into its update kernels (3 kernels per iteration
 * instead of 5; replaces nothing in the reference: its PETSc KSPCG has the same two reductions, ksp.py:71-78).  The number of
 * 1024-thread blocks those kernels run is a per-solve option (ox_ksp_options.fold_blocks); this is the library's default
 * for it: one block per compute unit.  Partitioned operators never fold (their points carry an all-reduce). */
int ox_ksp_default_fold_blocks(void);
/* Kernels one iteration of the given method launches on this operator with these options (reporting: bench.py);
 * partitioned != 0: the operator has a halo plan. */
int ox_ksp_kernels_per_iteration(int ksp_type, const ox_sell *A, int ncomp, int check_every, int fold_blocks,
                                 int partitioned);
int ox_profile_begin(int max_records, int sample_every); /* time every sample_every-th launch per tag */
int ox_profile_end(void);
int ox_profile_get(int tag, long long key, long long *count, double *total_ms); /* key: the matrix's
                                   n_rows for SpMV records (tells the pressure matrix from the
                                   velocity matrix), -1 = any */

/* roctx ranges on the calling host thread (rocprofv3 --kernel-trace --marker-trace segments the trace by them):
 * FractionalStep_AB_CN brackets its six phase methods (fracstep.py:411-658) with them -- what PETSc's log stages are
 * to the reference.  No-ops without librocprofiler-sdk-roctx / libroctx64 or without a profiler attached. */
int ox_range_push(const char *name);
int ox_range_pop(void);

/* ---- H1 + collectives: mesh-partitioned runs (one process per GPU, RCCL) -------------- */
int ox_comm_unique_id(char *id128);   /* ncclGetUniqueId on rank 0 (broadcast it out of band) */
int ox_comm_create(const char *id128, int rank, int nranks, void **comm_out); /* ncclCommInitRank */
int ox_comm_info(void *comm, int *nranks, int *rank, int *device); /* ncclCommCount / UserRank / CuDevice: what the
                                   communicator itself reports (bench.py's N > 1 line prints it); NULL outputs are skipped */
int ox_comm_destroy(void *comm);
/* Halo plan of one function space on communicator `comm`.  send_idx: device, owned rows to
 * pack, grouped per peer by send_off; ghosts arrive contiguously per peer at
 * x[n_owned + recv_off[p] ...] (the ghost block of a vector is ordered by (owner, global id)). */
int ox_dist_create(void *comm, int rank, int nranks, int n_peers, const int32_t *peers,
                   const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off,
                   int64_t n_owned, int64_t n_ghost, ox_dist **out);
/* Direct xGMI transport for a plan (alternative to the RCCL calls; `comm` of ox_dist_create may then
 * be NULL).  Every rank creates one uncached window of ox_p2p_window_bytes(nranks, its n_ghost)
 * bytes, hands the 64-byte IPC handle to all ranks out of band, opens theirs, and enables the
 * transport with: rank_wins[nranks] (own window at [rank]), and per peer of the plan the dof offset
 * of THIS rank's block inside the peer's ghost block (the peer's recv_off for this rank) and the
 * peer's n_ghost.  The plan then owns the window and the mappings.  Exchanges wait at most
 * timeout_s for a peer; ox_dist_status reports a time-out (sticky). */
size_t ox_p2p_window_bytes(int nranks, int64_t n_ghost);
int ox_p2p_window_create(size_t bytes, void **win_dev, char *handle64);
int ox_p2p_window_open(const char *handle64, void **win_dev);
int ox_p2p_window_close(void *win_dev);
int ox_p2p_window_free(void *win_dev);
int ox_dist_enable_p2p(ox_dist *d, void *my_win, void *const *rank_wins, const int64_t *peer_recv_off,
                       const int64_t *peer_n_ghost, double timeout_s);
int ox_dist_p2p_timeout(ox_dist *d, double timeout_s); /* change the bound of the peer waits */
/* The plan's distributed mat-vecs: 1 = start the halo exchange, multiply the interior slices while it is in flight, the
 * boundary slices after it has landed; 0 = exchange, then multiply; -1 = the transport's default (overlap on the xGMI
 * windows and the callback transports, off on RCCL plans until the side-stream send/recv has run between two real GPUs).
 * What DOLFINx/PETSc do inside MatMult (reference fracstep.py:453,497,632); never changes a result. */
int ox_dist_set_overlap(ox_dist *d, int overlap);
/* Release protocol of the xGMI-window transport of a plan (after ox_dist_enable_p2p).  conservative = 1 (the DEFAULT):
 * every storing wave of the push kernel fences at system scope behind its payload stores and the flags -- halo and
 * all-reduce -- are system-scope RELEASE stores.  conservative = 0: the fast form (the storing waves only wait for their
 * write-through stores to be acknowledged, ONE system fence by the last block / per all-reduce wave, relaxed flag
 * stores: 32 us instead of ~130 us for the 2.4 MB velocity halo of a 128^3 x 8 rank, 15 us per all-reduce -- figures
 * measured with every rank on ONE device; the form has never crossed a link).  Keep the default until a run between two
 * GPUs has passed the halo self-test, a bit-exact all-reduce and the rehearsal comparison with conservative = 0. */
int ox_dist_set_p2p_release(ox_dist *d, int conservative);
int ox_dist_disable_p2p(ox_dist *d);
int ox_dist_status(const ox_dist *d);
/* Same plan on a caller-supplied transport instead of RCCL (rehearsals on one GPU, other
 * fabrics): halo_cb(user, packed_send_values_dev, ghost_block_dev, ncomp) and
 * allreduce_cb(user, buf_dev, n) are called at the points where ncclSend/ncclRecv and
 * ncclAllReduce would be, after the stream has been drained; both return 0 on success. */
int ox_dist_create_custom(int rank, int nranks, int n_peers, const int32_t *peers,
                          const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off,
                          int64_t n_owned, int64_t n_ghost,
                          int (*halo_cb)(void *, const double *, double *, int),
                          int (*allreduce_cb)(void *, double *, int), void *user, ox_dist **out);
/* blocking copy helper for such transports: to_device = 1 host->device, 0 device->host */
int ox_memcpy(void *dst, const void *src, size_t bytes, int to_device, void *stream);
int ox_dist_destroy(ox_dist *d);
/* scatter_forward (owner -> ghost) of an interleaved vector (fracstep.py:453,...). */
int ox_halo_forward(const ox_dist *d, double *x, int ncomp, void *stream);
int ox_allreduce_sum(const ox_dist *d, double *buf_dev, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif

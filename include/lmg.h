/*
 * lmg.h -- C ABI of the MI355X (gfx950) multigrid V-cycle kernels.
 *
 * This is the drop-in boundary for the hot path of claudiotomasi/LearnMultigrid
 * (learn_multigrid/solvers/Multigrid.py:36-124).  The reference is pure Python
 * and has no FFI of its own; each entry point below replaces one SciPy / pyamg
 * call made on that path and cites it (paths relative to /root/reference).
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the Python side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - Every `const T*` / `T*` named d_* or documented "device" is a DEVICE pointer
 *     owned by the caller; the library never allocates, frees or synchronises
 *     inside a launch function (graph-capture safe) -- scratch is passed in.
 *   - CSR: int32 rowptr[n+1], int32 colidx[nnz], fp64 vals[nnz]; array bases must
 *     be 16-byte aligned (LMG_ERR_ALIGN otherwise).  Rows are accumulated in
 *     storage order with separate multiply and add roundings (no FMA contraction),
 *     which makes SpMV / Jacobi / Gauss-Seidel bit-identical to SciPy's
 *     csr_matvec and to pyamg's sweep on sorted CSR.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Return value: LMG_OK (0) or a negative lmg_status; no exceptions cross the ABI.
 */
#ifndef LMG_H
#define LMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMG_VERSION 100 /* 0.1.0 */

typedef enum {
    LMG_OK = 0,
    LMG_ERR_ARG = -1,       /* null pointer / negative size / bad enum            */
    LMG_ERR_ALIGN = -2,     /* array base not 16-byte aligned                     */
    LMG_ERR_LAUNCH = -3,    /* HIP reported a launch / runtime error              */
    LMG_ERR_CAPACITY = -4,  /* a row exceeds what the kernel's LDS tile can hold  */
    LMG_ERR_NODEVICE = -5   /* no HIP device visible                              */
} lmg_status;

int lmg_version(void);
const char *lmg_status_string(int status);
/* Number of visible HIP devices, or a negative lmg_status. */
int lmg_device_count(void);

/* Runtime tuning knobs (kernel variant selection; used by bench.py for A/B runs).
 *   key "pcsr_ju"       : row entries per step of the packed sweeps (0 = auto, 1, 3, 5).
 *   key "sweep_variant" : tile geometry of the plain-CSR sweep kernels, 0..6 (0 = default: chosen
 *                         per launch from the average row length).
 *   keys "rpat_variant", "rpat_nt_rows", "stencil_nt_rows", "stencil_wgs_per_cu", "fused_seg_lines",
 *        "fused_pf", "gs_single_max": geometry / cache-policy knobs of the twin kernels (parity tests force
 *        the instantiations a launcher would only pick on very large operators).       */
int lmg_tune_set(const char *key, int value);
int lmg_tune_get(const char *key);

/* ---- fine-level sweeps ---------------------------------------------------------
 * Number of fp64 partial sums lmg_csr_residual_norm2 / lmg_dot need as scratch. */
int64_t lmg_partials_count(int64_t n);

/* r = b - A x  and  *d_norm2 = sum_i r_i^2 (deterministic two-stage reduction).
 * Replaces `rhs - A.dot(u)` + `np.linalg.norm` (Multigrid.py:62-63,:90; Jacobi.py:28-29).
 * d_r may be NULL (norm only); d_partials/d_norm2 may both be NULL (residual only). */
int lmg_csr_residual_norm2(int64_t n, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_colidx,
                           const double *d_vals, const double *d_x, const double *d_b, double *d_r,
                           double *d_partials, double *d_norm2, void *stream);

/* x_out = x_in + omega * ((1/a_ii) * (b - A x_in)); rows without a diagonal copy x_in.
 * omega = 1 is the reference update `solution += inv_d * residual_vector`
 * (Jacobi.py:22-35).  x_out must not alias x_in. */
int lmg_csr_jacobi(int64_t n, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_colidx,
                   const double *d_vals, const double *d_x_in, const double *d_b, double omega,
                   double *d_x_out, void *stream);

/* y = alpha * (A x) + beta * y  (beta == 0 never reads y).
 * Restriction  r_c = P^T r  (Multigrid.py:93) with A := R = P^T stored as CSR;
 * prolongation u += P e     (Multigrid.py:115) with alpha = beta = 1. */
int lmg_csr_spmv(int64_t n_rows, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_colidx,
                 const double *d_vals, const double *d_x, double *d_y, double alpha, double beta,
                 void *stream);

/* ---- packed CSR ("PCSR") sweeps ---------------------------------------------------
 * Lossless re-encoding of a CSR matrix, built once at setup (learnmultigrid_amd/ops.py
 * PackedCSR shows how), that keeps the entry order -- results are bit-identical to the CSR
 * entry points above -- and moves fewer bytes:
 *   d_rowlen[n]          uint8 row lengths (rows longer than 255 entries: not packable)
 *   d_tile_base[T+1]     int32 entry offset of every tile of `tile_rows` rows (512 by default
 *                        = lmg_pcsr_tile_rows(); 128 or 64 for matrices with long rows)
 *   d_col                colmode 0: uint16 (column - d_tile_colbase[tile]); 1: int32 column
 *   d_val                valmode 0: uint8 index into d_dict (ndict <= 256); 1: uint16 index
 *                        (ndict <= 65536); 2: raw fp64 (d_dict unused)
 * d_col / d_val are 16-byte aligned and padded with >= 16 readable bytes; tile_cap is the
 * largest number of entries in one tile.  mode: 0 residual (+norm), 1 Jacobi, 2 SpMV, with
 * the argument meaning of lmg_csr_residual_norm2 / lmg_csr_jacobi (alpha = omega) /
 * lmg_csr_spmv.  LMG_ERR_CAPACITY if a tile does not fit the LDS budget (use the CSR path). */
int lmg_pcsr_tile_rows(void);
int lmg_pcsr_sweep(int mode, int64_t n, int64_t nnz, int32_t tile_rows, int32_t tile_cap,
                   const int32_t *d_tile_base,
                   const int32_t *d_tile_colbase, const uint8_t *d_rowlen, const void *d_col,
                   int colmode, const void *d_val, int valmode, const double *d_dict, int32_t ndict,
                   const double *d_x, const double *d_b, double *d_out, double alpha, double beta,
                   double *d_partials, double *d_norm2, void *stream);

/* ---- row-pattern ("RPAT") sweeps ---------------------------------------------------
 * Second lossless twin, for matrices whose rows repeat when written as (length; column - row
 * index and value bits of every entry, in storage order): assembled grid operators.  Each
 * distinct row is stored once, every row carries a uint8 pattern id:
 *   d_pid[n]                 pattern of every row
 *   d_pat_ptr[npat+1], d_pat_off[nent], d_pat_val[nent]   the patterns (CSR-like; off = col - row)
 * Limits from lmg_rpat_limits (255 patterns, 1024 entries in total; held in LDS).  Same modes,
 * argument meaning and bits as lmg_pcsr_sweep; max_len = longest pattern (picks the unrolling).
 * d_partials needs lmg_partials_count(n) doubles.
 * Building (learnmultigrid_amd/ops.py RowPatterns.from_csr is the calling sequence):
 *   lmg_rpat_row_hash   64-bit hash of every row's pattern
 *   -- distinct hashes with lmg_value_set_insert, ids with lmg_value_encode --
 *   lmg_rpat_claim      d_rep[p] (pre-set to -1) = some row with pattern id p
 *   lmg_rpat_verify     compares EVERY row with its pattern entry by entry (and its columns with
 *                       [0, ncols)); *d_mismatch != 0 means the format must not be used. */
int lmg_rpat_limits(int32_t *max_patterns, int32_t *max_entries);
int lmg_rpat_sweep(int mode, int64_t n, const uint8_t *d_pid, int32_t npat, int32_t nent, int32_t max_len,
                   const int32_t *d_pat_ptr, const int32_t *d_pat_off, const double *d_pat_val,
                   const double *d_x, const double *d_b, double *d_out, double alpha, double beta,
                   double *d_partials, double *d_norm2, void *stream);
int lmg_rpat_row_hash(int64_t n, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                      uint64_t *d_hash, void *stream);
int lmg_rpat_claim(int64_t n, const uint8_t *d_pid, int32_t *d_rep, void *stream);
int lmg_rpat_verify(int64_t n, int64_t ncols, const int32_t *d_rowptr, const int32_t *d_colidx,
                    const double *d_vals, const uint8_t *d_pid, int32_t npat, const int32_t *d_pat_ptr,
                    const int32_t *d_pat_off, const double *d_pat_val, int32_t *d_mismatch, void *stream);

/* Row patterns of RECTANGULAR grid operators (transfers between nested grids), SpMV only.  A pattern
 * then stores  column - base(row)  with, for y = row / row_len and x = row % row_len,
 *     base(row) = (y >> ysh) * col_stride + ((x >> xsh) << xshl),
 * h_grid_map = HOST pointer to {row_len, col_stride, ysh, xsh, xshl} (NULL or row_len 0: base(row) =
 * row, i.e. the functions above).  Restriction R = P^T of a tensor-product interpolator between a
 * Wf x Wf and a Wc x Wc grid: {Wc, 2 Wf, 0, 0, 1}; its prolongation P: {Wf, Wc, 1, 1, 0}; 1-D
 * transfers: row_len > number of rows.  The format stays a verified lossless re-encoding: the builder
 * (ops.RowPatterns.from_csr(A, grid_map)) tries a map, and lmg_rpat_verify_grid checks every entry.
 * u += P e and r_c = R r then move the vectors and one byte per row instead of 10-12 bytes per entry.
 * Replaces the same SciPy calls as lmg_csr_spmv (Multigrid.py:93, :115). */
int lmg_rpat_sweep_grid(int mode, int64_t n, const int32_t *h_grid_map, const uint8_t *d_pid, int32_t npat,
                        int32_t nent, int32_t max_len, const int32_t *d_pat_ptr, const int32_t *d_pat_off,
                        const double *d_pat_val, const double *d_x, const double *d_b, double *d_out,
                        double alpha, double beta, double *d_partials, double *d_norm2, void *stream);
int lmg_rpat_row_hash_grid(int64_t n, const int32_t *h_grid_map, const int32_t *d_rowptr,
                           const int32_t *d_colidx, const double *d_vals, uint64_t *d_hash, void *stream);
int lmg_rpat_verify_grid(int64_t n, int64_t ncols, const int32_t *h_grid_map, const int32_t *d_rowptr,
                         const int32_t *d_colidx, const double *d_vals, const uint8_t *d_pid, int32_t npat,
                         const int32_t *d_pat_ptr, const int32_t *d_pat_off, const double *d_pat_val,
                         int32_t *d_mismatch, void *stream);

/* ---- grid-stencil sweeps: row-pattern matrices whose patterns are 3x3 stencils ------------
 * A row-pattern matrix (d_pid as above) qualifies when every entry of every pattern sits at
 * column - row = c * line_stride + d with c, d in {-1, 0, 1} and every pattern lists its entries in
 * ascending column order: the 5-point operator of configs #2 / #4, its 9-point Galerkin
 * coarsenings, tridiagonal 1-D operators.  The pattern table is then stored by SLOT:
 *   d_st_val[npat * 9]   value of slot (c+1)*3 + (d+1) of every pattern (unused slots: anything)
 *   d_st_mask[npat]      bit s set = slot s present in the pattern
 *   union_mask           OR of all d_st_mask (slots nobody uses are never loaded)
 * (learnmultigrid_amd/ops.py StencilTwin.from_patterns derives line_stride and the table from a
 * verified RowPatterns twin and refuses everything else; at most lmg_stencil_limits patterns.)
 * A lane owns two consecutive rows, reads x with three 16-byte loads and shares the left / right
 * neighbours inside the wave -- 7 vector-memory instructions per 128 rows instead of 16-24 -- and
 * accumulates every row's entries in ascending column order: same modes, argument meaning and bits
 * as lmg_rpat_sweep.  d_x, d_b, d_out must be 16-byte aligned; square matrices only (x has n entries). */
int lmg_stencil_limits(int32_t *max_patterns);
int lmg_stencil_sweep(int mode, int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                      const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask, const double *d_x,
                      const double *d_b, double *d_out, double alpha, double beta, double *d_partials,
                      double *d_norm2, void *stream);

/* Fused smoothing pass on a grid-stencil matrix (same format arguments as lmg_stencil_sweep):
 *     x_out = J^sweeps(x_in),  J(x) = x + omega * D^-1 (b - A x),   sweeps = 1..3,
 *     r_out = b - A x_out      (d_r_out may be NULL),
 * i.e. the `smooth_steps` sweeps of Multigrid.py:88 / :121 and the residual of :90 in ONE pass over
 * x_in, b and the pattern ids (temporal blocking in registers: a wave marches down a 128-column strip
 * with all intermediate iterates in registers, see csrc/stencil_fused.hip).  d_x_in == NULL means a zero
 * initial iterate (coarse levels, Multigrid.py:103): the first sweep then is omega * (D^-1 b), the bits
 * of lmg_vmul.  hot_pattern / h_hot_val (HOST pointer to its 9 slot values; -1 / NULL: none) name a
 * pattern that has every slot of union_mask and a non-zero diagonal -- the interior row of a grid
 * operator: lines on which all lanes of a wave hold it run with its values in scalar registers.  Every
 * value equals the one the separate lmg_stencil_sweep launches produce, bit for bit.  x_out (and
 * r_out) must not alias x_in.  Compiled for the union masks lmg_stencil_smooth_supported accepts
 * (5-point 0x0BA, 9-point 0x1FF, 1-D 0x038); LMG_ERR_CAPACITY for others (run the separate sweeps). */
int lmg_stencil_smooth_supported(uint32_t union_mask);
int lmg_stencil_smooth(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                       const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                       int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                       const double *d_b, double omega, double *d_x_out, double *d_r_out, void *stream);

/* The same pass for SMALL levels (10^4 .. 10^6 rows), iterates in LDS instead of registers: a workgroup owns a tile
 * of 64 columns x 16 / 32 lines, loads it once, runs the sweeps between two LDS buffers and stores its inner part
 * (csrc/stencil_tile.hip): one launch instead of sweeps + 1, same bits.  Same arguments as lmg_stencil_smooth;
 * 5- and 9-point union masks (lmg_stencil_smooth_tiled_supported). */
int lmg_stencil_smooth_tiled_supported(uint32_t union_mask);
int lmg_stencil_smooth_tiled(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                             const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                             int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                             const double *d_b, double omega, double *d_x_out, double *d_r_out, void *stream);
/* ... and with the coarse-grid correction folded in, x_out = J^sweeps(x_in + P e_coarse): the tile is loaded as
 * x + P e (row patterns of P as in lmg_stencil_smooth_prolong, no frequent-pair shortcut needed: the correction is
 * formed once per element, with the sums of lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 1) in order). */
int lmg_stencil_smooth_tiled_prolong(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                                     const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                                     int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                                     const double *d_b, double omega, double *d_x_out, int64_t n_coarse,
                                     int32_t coarse_stride, const double *d_e_coarse, const uint8_t *d_p_pid,
                                     int32_t p_npat, const double *d_p_val, const int32_t *d_p_mask, void *stream);
/* ... and with the restriction folded in, b_coarse = R (b - A x_out) instead of the residual (arguments of
 * lmg_stencil_smooth_restrict without the frequent-pattern shortcut): the residual of the tile goes to LDS, every
 * element on an (even line, even column) node sums the nine entries of its row of R in column order. */
int lmg_stencil_smooth_tiled_restrict(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                                      const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                                      int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                                      const double *d_b, double omega, double *d_x_out, int64_t n_coarse,
                                      int32_t coarse_stride, double *d_b_coarse, const uint8_t *d_r_pid,
                                      int32_t r_npat, const double *d_r_val, const int32_t *d_r_mask, void *stream);

/* ---- fused smoothing passes for grid operators with VARIABLE coefficients (csrc/dia_tile.hip) ------------------------
 * The operators the reference's learned transfers are built for (variable-coefficient / jittered-mesh stiffness
 * matrices, Multigrid.py:306-370, :741-765) have the 3x3 slot geometry of lmg_stencil_sweep -- every entry at
 * column - row = c * line_stride + d, c, d in {-1, 0, 1} -- but no repeating rows.  Their DIA twin stores one fp64 array
 * of n values per slot of the union mask, slots ascending: d_dia[q * n + row] (40 B/row for a 5-point operator; a row
 * without that slot holds +0.0, bitwise neutral for finite iterates like an explicit zero of the CSR input).
 *   lmg_dia_fill   : d_dia from sorted CSR; *d_mismatch |= 1 if an entry is no slot of the mask (d_dia = NULL: probe only,
 *                    the slots seen are OR-ed into *d_mask_out);
 *   lmg_dia_smooth : x_out = J^sweeps(x_in) (x_in = NULL: zero iterate), r_out = b - A x_out (optional), sweeps 1..3, in
 *                    ONE pass: the `sweeps` forward relaxations of pyamg's gauss_seidel call sites in their weighted-
 *                    Jacobi form (Multigrid.py:88, :121; Jacobi.py:35) plus the residual of :90 -- a workgroup per
 *                    64-column tile, the matrix values of a lane's rows in registers for all sweeps, the iterate in LDS;
 *                    same bits as lmg_csr_jacobi x sweeps + lmg_csr_residual_norm2.  Union masks: 5-point, both 7-point
 *                    orientations, 9-point (lmg_dia_smooth_supported). */
int lmg_dia_smooth_supported(uint32_t union_mask);
int lmg_dia_fill(int64_t n, int32_t line_stride, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                 uint32_t union_mask, double *d_dia, int32_t *d_mismatch, uint32_t *d_mask_out, void *stream);
int lmg_dia_smooth(int64_t n, int32_t line_stride, uint32_t union_mask, const double *d_dia, int sweeps,
                   const double *d_x_in, const double *d_b, double omega, double *d_x_out, double *d_r_out, void *stream);

/* The same pass with the coarse-grid correction folded in (Multigrid.py:115 + :121 in one pass):
 *     x_out = J^sweeps(x_in + P e_coarse)
 * for a prolongation P whose row (y, x) -- y = row / line_stride, x = row % line_stride -- reads
 * e_coarse only at  ((y >> 1) * coarse_stride + (x >> 1)) + {0, 1, coarse_stride, coarse_stride + 1}  (slots
 * 0..3, ascending columns): the tensor-product interpolation between nested grids, stored as row
 * patterns (d_p_pid: one uint8 id per row of P; d_p_val [p_npat][4] values by slot, d_p_mask [p_npat]
 * slots present; rows on even lines must not use slots 2, 3).  x_in + P e is never written: the
 * correction is formed when a line arrives, with the sums of lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 1)
 * in the same order, so x_out has the bits of prolongation + separate sweeps.  h_hot_pairs (HOST, 2 ints,
 * may be NULL): for even / odd lines the ids (even column | odd column << 8) of the usual pattern pair,
 * with masks {0}, {0,1} / {0,2}, {0,1,2,3}, or -1; h_hot_pval (HOST, 9 doubles): their values in that
 * order.  5- and 9-point union masks (lmg_stencil_smooth_prolong_supported). */
int lmg_stencil_smooth_prolong_supported(uint32_t union_mask);
int lmg_stencil_smooth_prolong(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                               const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                               int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                               const double *d_b, double omega, double *d_x_out, int64_t n_coarse,
                               int32_t coarse_stride, const double *d_e_coarse, const uint8_t *d_p_pid,
                               int32_t p_npat, const double *d_p_val, const int32_t *d_p_mask,
                               const int32_t *h_hot_pairs, const double *h_hot_pval, void *stream);

/* The pre-smoothing pass with the restriction folded in (Multigrid.py:88 + :90 + :93 in one pass):
 *     x_out = J^sweeps(x_in),   b_coarse = R (b - A x_out)
 * for a restriction R whose row (Y, X) -- Y = row / coarse_stride, X = row % coarse_stride -- reads the fine
 * residual only at  (2 Y * line_stride + 2 X) + c * line_stride + d,  c, d in {-1, 0, 1}  (slots 0..8,
 * ascending columns): the transpose of the tensor-product interpolation, stored as row patterns (d_r_pid:
 * one uint8 id per row of R; d_r_val [r_npat][9], d_r_mask [r_npat]).  The residual is never written: its last
 * three lines stay in registers, and every coarse row is summed like lmg_rpat_sweep_grid(SPMV, alpha = 1,
 * beta = 0) would, so b_coarse has the bits of residual + restriction launches.  Every fine node (even line,
 * even column) must have its coarse row (LMG_ERR_ARG otherwise).  hot_r / h_hot_rval (HOST, 9 doubles): a
 * pattern with all nine slots, or -1 / NULL.  d_x_in == NULL: zero initial iterate, as in lmg_stencil_smooth. */
int lmg_stencil_smooth_restrict(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                                const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                                int32_t hot_pattern, const double *h_hot_val, int sweeps, const double *d_x_in,
                                const double *d_b, double omega, double *d_x_out, int64_t n_coarse,
                                int32_t coarse_stride, double *d_b_coarse, const uint8_t *d_r_pid,
                                int32_t r_npat, const double *d_r_val, const int32_t *d_r_mask, int32_t hot_r,
                                const double *h_hot_rval, void *stream);

/* ---- sliced-ELL ("SELL-64") sweeps: matrices with long rows ---------------------------
 * Third lossless twin.  Slice s = rows 64 s .. 64 s + 63, padded to its longest row
 * d_slice_len[s]; entry j of row r lives at d_slice_base[s] + 64 j + (r mod 64) of d_col
 * (colmode 0: uint16 column - d_slice_cmin[s]; 1: int32 column) and d_val (fp64), so a wave
 * reads entry j of its 64 rows with one coalesced load per stream and needs no LDS.  Rows are
 * still accumulated in storage order: same bits as the CSR entry points.  d_rowlen[n] = true
 * row lengths; max_len = longest row (picks the unrolling); modes and arguments as
 * lmg_pcsr_sweep.  Building: lmg_sell_slice_info (padded length and column range per slice),
 * d_slice_base = exclusive int64 scan of 64 * d_slice_len, lmg_sell_fill (d_col_out may be NULL
 * to refresh the values only); the padded arrays must be zero-initialised. */
int lmg_sell_sweep(int mode, int64_t n, const int64_t *d_slice_base, const int32_t *d_slice_len,
                   const int32_t *d_slice_cmin, const int32_t *d_rowlen, const void *d_col, int colmode,
                   const double *d_val, int32_t max_len, const double *d_x, const double *d_b, double *d_out,
                   double alpha, double beta, double *d_partials, double *d_norm2, void *stream);
int lmg_sell_slice_info(int64_t n, const int32_t *d_rowptr, const int32_t *d_colidx, int32_t *d_slice_len,
                        int32_t *d_cmin, int32_t *d_cmax, void *stream);
int lmg_sell_fill(int64_t n, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                  const int64_t *d_slice_base, const int32_t *d_slice_cmin, int colmode, void *d_col_out,
                  double *d_val_out, void *stream);

/* ---- building the packed twin (setup) ----------------------------------------------
 * One streaming pass each; learnmultigrid_amd/ops.py PackedCSR.from_csr is the calling sequence.
 *   lmg_pcsr_tile_colrange   smallest / largest column of every tile of `tile_rows` rows
 *                            (0 / 0 for an empty tile)
 *   lmg_pcsr_encode_cols16   d_out[e] = uint16(d_colidx[e] - d_colbase[tile of e])
 *   lmg_value_set_insert     inserts the bit patterns of d_vals into an open-addressing table
 *                            (d_table: table_slots uint64, a power of two >= 2 * (limit + 262144),
 *                            pre-filled with 0xFF bytes).  d_state[3] (zeroed by the caller):
 *                            [0] distinct values seen, [1] != 0 if more than `limit` (the table
 *                            content is then meaningless), [2] != 0 if the all-ones pattern --
 *                            the empty marker -- occurs among the values.
 *   lmg_value_encode         d_out[e] = index of d_vals[e] in d_dict (width 1: uint8, 2: uint16);
 *                            d_dict sorted ascending as SIGNED 64-bit patterns; *d_missing is set
 *                            when a value is not in the dictionary.
 *   lmg_csr_inverse_diagonal d_dinv[i] = 1 / sum of the diagonal entries of row i, 0 when
 *                            that sum is 0 or the row has no diagonal entry. */
int lmg_pcsr_tile_colrange(int64_t n, int32_t tile_rows, const int32_t *d_rowptr, const int32_t *d_colidx,
                           int32_t *d_cmin, int32_t *d_cmax, void *stream);
int lmg_pcsr_encode_cols16(int64_t n, int32_t tile_rows, const int32_t *d_rowptr, const int32_t *d_colidx,
                           const int32_t *d_colbase, uint16_t *d_out, void *stream);
/* the occupied slots of that table, appended in ANY order to d_out (at most cap of them; *d_count, zeroed by the caller,
 * receives how many there were): the few survivors are sorted on the host */
int lmg_value_set_collect(const uint64_t *d_table, int64_t table_slots, uint64_t *d_out, int32_t cap, int32_t *d_count,
                          void *stream);
int lmg_value_set_insert(int64_t count, const double *d_vals, uint64_t *d_table, int64_t table_slots,
                         int32_t limit, int32_t *d_state, void *stream);
int lmg_value_encode(int64_t count, const double *d_vals, const double *d_dict, int32_t ndict, int width,
                     void *d_out, int32_t *d_missing, void *stream);
int lmg_csr_inverse_diagonal(int64_t n, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                             double *d_dinv, void *stream);

/* CSR transpose by counting sort (setup: the restriction R = P^T as an explicit CSR, `i.T` of
 * Multigrid.py:93).  lmg_csr_transpose_count: d_counts[c] (zeroed by the caller, ncols entries) += number of
 * entries in column c; the caller scans d_counts into d_t_rowptr with lmg_exclusive_scan_i32;
 * lmg_csr_transpose_fill places every entry (d_cursor: ncols zeroed int32) and sorts each row of A^T by
 * column, so the result equals SciPy's A.T.tocsr() for a sorted, duplicate-free A.  Rows of A^T longer than
 * lmg_csr_transpose_max_row() entries are sorted by a one-thread insertion sort too (slow): callers with
 * such rows use a library sort instead (ops.DeviceCSR.transpose does). */
int lmg_csr_transpose_max_row(void);
int lmg_csr_transpose_count(int64_t nnz, int64_t ncols, const int32_t *d_colidx, int32_t *d_counts, void *stream);
int lmg_csr_transpose_fill(int64_t n, int64_t ncols, const int32_t *d_rowptr, const int32_t *d_colidx,
                           const double *d_vals, const int32_t *d_t_rowptr, int32_t *d_cursor, int32_t *d_t_colidx,
                           double *d_t_vals, void *stream);

/* ---- Gauss-Seidel ---------------------------------------------------------------
 * One independent set: for every i in d_rows, in place,
 *     x_i = (b_i - sum_{j != i} a_ij x_j) / a_ii      (skipped when a_ii == 0)
 * -- the row update of pyamg's gauss_seidel (called at Multigrid.py:88,:121).  The rows
 * of one call must be mutually independent in the pattern of A + A^T. */
int lmg_csr_gs_rows(const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                    double *d_x, const double *d_b, const int32_t *d_rows, int64_t nrows,
                    void *stream);

/* `sweeps` sweeps over a schedule of `nsets` independent sets executed in order
 * (level schedule => exact lexicographic forward Gauss-Seidel; colour classes =>
 * multicolour Gauss-Seidel).  d_set_ptr / h_set_ptr are the same nsets+1 offsets into
 * d_set_rows on device and host.  max_set = largest set size. */
int lmg_csr_gs_schedule(const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                        double *d_x, const double *d_b, const int32_t *d_set_rows,
                        const int32_t *d_set_ptr, const int32_t *h_set_ptr, int64_t nsets,
                        int64_t max_set, int sweeps, void *stream);
/* The one-workgroup executor of lmg_csr_gs_schedule on a copy of the PATTERN in schedule order
 * (for schedules of a few thousand rows per set at most: grids up to ~1025^2): d_ell_row[k] the
 * k-th scheduled row, d_ell_start[k] / d_ell_len[k] its entry range in the CSR arrays,
 * d_ell_cols[j * total_rows + k] its j-th column (j < ell_k in {3, 5, 7, 9, 16} >= the longest
 * row; padding entries hold the row itself).  Values are read from d_vals, so the copy stays
 * valid across coefficient changes.  n = length of d_x (up to 18 000 unknowns the iterate is kept
 * in LDS for the whole call).  Same results as lmg_csr_gs_schedule, bit for bit. */
int lmg_csr_gs_schedule_ell(int64_t n, const double *d_vals, double *d_x, const double *d_b, const int32_t *d_ell_row,
                            const int32_t *d_ell_start, const int32_t *d_ell_len, const int32_t *d_ell_cols,
                            int32_t ell_k, int64_t total_rows, const int32_t *d_set_ptr, int64_t nsets,
                            int sweeps, void *stream);

/* Exact forward (lexicographic) Gauss-Seidel on a grid-stencil matrix (format arguments of
 * lmg_stencil_sweep) as a pipelined wavefront in registers: a wave owns 64 grid lines, lane l relaxes
 * column t - SK*l of its line at step t, neighbouring bands hand their boundary line over through memory
 * behind a progress counter (csrc/gs_wave.hip).  `sweeps` forward sweeps in place on d_x: the same bits as
 * lmg_csr_gs_schedule on the level schedule and as pyamg's gauss_seidel (Multigrid.py:88, :121), at ~0.15 us
 * per dependent anti-diagonal instead of a kernel launch each.  Requirements (the caller checks them:
 * ops.StencilTwin.gs_ok): union_mask accepted by lmg_stencil_gs_supported (5-, 7-, 9-point, 1-D), and no
 * pattern used in column 0 / line_stride-1 of a line couples across the line end.  hot_pattern /
 * h_hot_val as in lmg_stencil_smooth (steps in which every lane relaxes that pattern skip the LDS table).  d_work:
 * lmg_stencil_gs_work_bytes(n, line_stride) bytes, 8-byte aligned, its FIRST int32 zeroed by the caller
 * once (it is set when a band had to give up waiting: the result is then invalid).  Up to four sweeps share
 * one launch, pipelined behind each other (band b of sweep s trails band b + 1 of sweep s - 1), so the
 * pipeline fill -- 64 SK steps per band -- is paid once per launch, not once per sweep. */
int lmg_stencil_gs_supported(uint32_t union_mask);
int64_t lmg_stencil_gs_work_bytes(int64_t n, int32_t line_stride);
int lmg_stencil_gs_sweep(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t npat,
                         const double *d_st_val, const int32_t *d_st_mask, uint32_t union_mask,
                         int32_t hot_pattern, const double *h_hot_val, double *d_x, const double *d_b,
                         void *d_work, int sweeps, void *stream);

/* HOST helpers (host pointers, run on the CPU at setup time).
 * level[i] = 1 + max(level[j] : j < i adjacent to i in A + A^T), 0 if none: rows of equal
 * level are independent, executing levels in order IS the lexicographic sweep.
 * Returns the number of levels (>= 0) or a negative lmg_status. */
int64_t lmg_host_gs_levels(int64_t n, const int32_t *h_rowptr, const int32_t *h_colidx,
                           int32_t *h_level_out);
/* Greedy natural-order colouring of A + A^T; returns the number of colours. */
int64_t lmg_host_greedy_colors(int64_t n, const int32_t *h_rowptr, const int32_t *h_colidx,
                               int32_t *h_color_out);

/* ---- vectors --------------------------------------------------------------------*/
int lmg_axpby(int64_t n, double alpha, const double *d_x, double beta, double *d_y, void *stream);
/* out = alpha * (x * y) elementwise.  With x = 1/diag(A), y = b, alpha = omega this is the
 * Jacobi sweep from a zero initial guess (coarse levels start from zeros, Multigrid.py:103). */
int lmg_vmul(int64_t n, double alpha, const double *d_x, const double *d_y, double *d_out, void *stream);
int lmg_copy(int64_t n, const double *d_src, double *d_dst, void *stream);
int lmg_zero(int64_t n, double *d_x, void *stream);
int lmg_dot(int64_t n, const double *d_x, const double *d_y, double *d_partials, double *d_out,
            void *stream);
/* buf[k] = x[idx[k]] (halo pack) and x[idx[k]] = buf[k] (unpack). */
int lmg_gather(int64_t n, const int32_t *d_idx, const double *d_x, double *d_buf, void *stream);
int lmg_scatter(int64_t n, const int32_t *d_idx, const double *d_buf, double *d_x, void *stream);

/* ---- coarsest level --------------------------------------------------------------
 * y = M x for a dense row-major n x m matrix: applies a pre-factored coarse operator
 * (M = A_L^-1 computed once at setup) in place of the per-cycle SuperLU
 * `spsolve(A_coarse, res_coarse)` of Multigrid.py:106. */
int lmg_dense_gemv(int64_t n, int64_t m, const double *d_M, const double *d_x, double *d_y,
                   void *stream);
/* y_b = M_b x_b for nblocks dense bs x bs blocks stored one after the other (bs even):
 * the strip solves of the banded block-elimination coarse solver (coarse.py). */
int lmg_dense_gemv_blockdiag(int64_t nblocks, int64_t bs, const double *d_M, const double *d_x,
                             double *d_y, void *stream);

/* Batched GEMV on windows of a vector, for block k < nblocks and row r < rows:
 *     y[k*y_stride + r] = (d_z ? d_z[k*z_stride + r] : 0) + alpha * sum_c M[k][r][c] * x[k*x_stride + c]
 * (M: nblocks dense rows x cols blocks, row-major; windows may overlap).  The building block of the
 * block-cyclic-reduction coarse solver (coarse.py), which lifts the size limit of the dense / one-level
 * banded solvers: `spsolve` on coarse operators of 10^5 unknowns (2-level runs, Multigrid.py:106).
 * cols and x_stride even, M and x 16-byte aligned.  lmg_block_copy: dst[k*dst_stride + i] =
 * src[k*src_stride + i], i < bs. */
int lmg_dense_gemv_windows(int64_t nblocks, int64_t rows, int64_t cols, const double *d_M, const double *d_x,
                           int64_t x_stride, const double *d_z, int64_t z_stride, double alpha, double *d_y,
                           int64_t y_stride, void *stream);

/* The same with a table of window starts: the x window of block k is d_x[d_x_offsets[k] .. + cols) (any 8-byte
 * aligned start; the caller keeps every window inside the allocation).  Back-substitution of the banded coarse
 * solver: x_I = y_I - (A_II^-1 A_IS) x_S in one launch (coarse.py BandedBlockSolver.apply). */
int lmg_dense_gemv_windows_off(int64_t nblocks, int64_t rows, int64_t cols, const double *d_M, const double *d_x,
                               const int32_t *d_x_offsets, const double *d_z, int64_t z_stride, double alpha,
                               double *d_y, int64_t y_stride, void *stream);

/* First and last product of the banded coarse solver (coarse.py) with its permutation perm = [strip unknowns |
 * separator unknowns] folded in, so that no gather / scatter launch surrounds the solve:
 *   lmg_coarse_front: y = blockdiag(M) b[perm[0 .. nblocks*bs)],  tail_out[i] = b[perm[nblocks*bs + i]], i < ntail;
 *   lmg_coarse_back : out[perm[k*rows + r]] (+)= z[k*z_stride + r] + alpha * M_k[r,:] . x[x_offsets[k] ..),
 *                     out[perm[nblocks*rows + i]] (+)= x[i], i < ntail   (accumulate != 0: += , the refinement step).
 * Every block of d_perm[0 .. nblocks*bs) is a run of consecutive indices (perm[k bs + c] = perm[k bs] + c: a strip of the
 * banded solver), which lmg_coarse_front relies on.
 * Same sums as lmg_dense_gemv_blockdiag / lmg_dense_gemv_windows_off followed by lmg_gather / lmg_scatter / lmg_axpby. */
int lmg_coarse_front(int64_t nblocks, int64_t bs, const double *d_M, const double *d_b, const int32_t *d_perm, double *d_y,
                     int64_t ntail, double *d_tail_out, void *stream);
int lmg_coarse_back(int64_t nblocks, int64_t rows, int64_t cols, const double *d_M, const double *d_x,
                    const int32_t *d_x_offsets, const double *d_z, int64_t z_stride, double alpha, const int32_t *d_perm,
                    double *d_out, int accumulate, int64_t ntail, void *stream);

/* The same two products for blocks that are not runs of consecutive unknowns (coarse.py GridBlockSolver: rectangular
 * blocks of a grid cut by separator lines AND columns, padded to one size) -- every operand index comes from a table:
 *   lmg_coarse_front_gather: y[k*bs + r] = sum_c M_k[r][c] * b[d_idx[k*bs + c]]  (d_idx < 0: padding, contributes 0),
 *                            tail_out[i] = b[d_tail_idx[i]], i < ntail;
 *   lmg_coarse_back_gather : out[d_oidx[k*rows + r]] (+)= z[k*z_stride + r] + alpha * sum_c M_k[r][c] * x[d_xidx[k*cols + c]]
 *                            (d_oidx < 0: padding row, skipped),  out[d_tail_idx[i]] (+)= x[i], i < ntail.
 * bs and cols even, index tables 8-byte aligned.  Replaces the per-cycle SuperLU `spsolve` of Multigrid.py:106 together
 * with lmg_csr_spmv and lmg_dense_gemv (four launches per application). */
int lmg_coarse_front_gather(int64_t nblocks, int64_t bs, const double *d_M, const double *d_b, const int32_t *d_idx,
                            double *d_y, int64_t ntail, const int32_t *d_tail_idx, double *d_tail_out, void *stream);
int lmg_coarse_back_gather(int64_t nblocks, int64_t rows, int64_t cols, const double *d_M, const double *d_x,
                           const int32_t *d_xidx, const double *d_z, int64_t z_stride, double alpha, const int32_t *d_oidx,
                           double *d_out, int accumulate, int64_t ntail, const int32_t *d_tail_idx, void *stream);

/* Dense fp64 helpers of the coarse-solver SETUP (csrc/gemm.hip; what NumPy / SuperLU do on the host inside the reference's
 * per-cycle `spsolve`, Multigrid.py:106, is a one-off factorisation here): row-major, leading dimensions, batch strides.
 *   lmg_batched_gemm: C[b] = alpha * A[b] (M x K) * B[b] (K x N) + beta * C[b]   (beta == 0: C is not read)
 *   lmg_copy2d      : dst[b][r][c] = alpha * src[b][r][c]   (accumulate != 0: += ) */
int lmg_batched_gemm(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double *d_A, int64_t lda,
                     int64_t stride_a, const double *d_B, int64_t ldb, int64_t stride_b, double beta, double *d_C,
                     int64_t ldc, int64_t stride_c, void *stream);
int lmg_copy2d(int64_t batch, int64_t rows, int64_t cols, double alpha, const double *d_src, int64_t ld_src,
               int64_t stride_src, double *d_dst, int64_t ld_dst, int64_t stride_dst, int accumulate, void *stream);

/* Setup helper: d_counts[((y & 1) * 2 + (x & 1)) * 256 + id] += number of rows i = y * line_stride + x with pattern id
 * d_pid[i] = id (d_counts: 1024 int32, zeroed by the caller). */
int lmg_pattern_parity_counts(int64_t n, int32_t line_stride, const uint8_t *d_pid, int32_t *d_counts, void *stream);
int lmg_block_copy(int64_t nblocks, int64_t bs, const double *d_src, int64_t src_stride, double *d_dst,
                   int64_t dst_stride, void *stream);

/* Inverses of nmat dense n x n matrices (n <= 128, row-major, one after the other): one workgroup per
 * matrix, Gauss-Jordan with partial pivoting in LDS.  d_info[k] = 1 when matrix k has an exactly zero
 * pivot (d_Ainv of that matrix is then undefined).  Setup-time helper of the coarse solvers. */
/* d_dense (n x m, row-major, zeroed by the caller) += the CSR matrix (duplicate entries are summed). */
int lmg_csr_to_dense(int64_t n, int64_t m, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                     double *d_dense, void *stream);
int lmg_batched_inverse(int64_t nmat, int32_t n, const double *d_A, double *d_Ainv, int32_t *d_info, void *stream);

/* ---- Galerkin product (SpGEMM)  C = A * B -----------------------------------------
 * Replaces SciPy's csr_matmat behind `i.T @ A @ i` (Multigrid.py:97-98), evaluated as
 * (R A) P like SciPy does.  Row-wise Gustavson with an expand / stable-sort / in-order
 * compress per row, so C has sorted rows and every c_ik is accumulated in the order of
 * the products a_ij*b_jk as SciPy traverses them.
 *   1. lmg_spgemm_count     : d_row_products[i] = sum_j nnz(B_j)  (upper bound of nnz(C_i))
 *   2. lmg_spgemm_symbolic  : d_c_rownnz[i] = nnz(C_i)            (needs step 1)
 *      -- caller turns d_c_rownnz into d_c_rowptr with lmg_exclusive_scan_i32 --
 *   3. lmg_spgemm_numeric   : fills d_c_colidx / d_c_vals.
 * Rows whose product count exceeds LMG_SPGEMM_MAX_ROW_PRODUCTS are skipped by steps 2 and 3
 * and must be handled by lmg_spgemm_long_rows (same results, dense accumulator in global
 * scratch: nsets x b_cols doubles and int32 marks, marks zero-initialised by the caller). */
#define LMG_SPGEMM_MAX_ROW_PRODUCTS 8192
int lmg_spgemm_count(int64_t a_rows, const int32_t *d_a_rowptr, const int32_t *d_a_colidx,
                     const int32_t *d_b_rowptr, int32_t *d_row_products, int32_t *d_max_products,
                     void *stream);
int lmg_spgemm_symbolic(int64_t a_rows, const int32_t *d_a_rowptr, const int32_t *d_a_colidx,
                        const int32_t *d_b_rowptr, const int32_t *d_b_colidx,
                        const int32_t *d_row_products, int32_t max_products,
                        int32_t *d_c_rownnz, void *stream);
int lmg_spgemm_numeric(int64_t a_rows, const int32_t *d_a_rowptr, const int32_t *d_a_colidx,
                       const double *d_a_vals, const int32_t *d_b_rowptr,
                       const int32_t *d_b_colidx, const double *d_b_vals,
                       const int32_t *d_row_products, int32_t max_products,
                       const int32_t *d_c_rowptr, int32_t *d_c_colidx, double *d_c_vals,
                       void *stream);
/* Numeric passes on a RECORDED pattern (coefficient changes on a fixed sparsity pattern:
 * the Galerkin rebuild).  lmg_spgemm_numeric_record is lmg_spgemm_numeric that also stores, for
 * every product, its position in the sorted product list of its row (d_dst, uint16, indexed
 * d_prod_ptr[row] + traversal sequence number; d_prod_ptr = exclusive int64 scan of
 * d_row_products with the long rows counted as 0) and the end of every C entry's segment in that
 * list (d_segend[nnz(C)], uint16).  lmg_spgemm_numeric_replay then recomputes d_c_vals without
 * sorting -- same products, same order of additions, bit-identical values -- at 2 bytes of
 * extra traffic per product.  d_c_colidx is not touched by the replay. */
int lmg_spgemm_numeric_record(int64_t a_rows, const int32_t *d_a_rowptr, const int32_t *d_a_colidx,
                              const double *d_a_vals, const int32_t *d_b_rowptr,
                              const int32_t *d_b_colidx, const double *d_b_vals,
                              const int32_t *d_row_products, int32_t max_products,
                              const int32_t *d_c_rowptr, int32_t *d_c_colidx, double *d_c_vals,
                              const int64_t *d_prod_ptr, uint16_t *d_dst, uint16_t *d_segend,
                              void *stream);
int lmg_spgemm_numeric_replay(int64_t a_rows, const int32_t *d_a_rowptr, const int32_t *d_a_colidx,
                              const double *d_a_vals, const int32_t *d_b_rowptr, const double *d_b_vals,
                              const int32_t *d_row_products, int32_t max_products,
                              const int32_t *d_c_rowptr, double *d_c_vals, const int64_t *d_prod_ptr,
                              const uint16_t *d_dst, const uint16_t *d_segend, void *stream);
int lmg_spgemm_long_rows(int numeric, int64_t nlong, const int32_t *d_long_rows,
                         const int32_t *d_a_rowptr, const int32_t *d_a_colidx, const double *d_a_vals,
                         const int32_t *d_b_rowptr, const int32_t *d_b_colidx, const double *d_b_vals,
                         int64_t b_cols, int32_t nsets, double *d_scratch_val, int32_t *d_scratch_mark,
                         int32_t *d_c_rownnz, const int32_t *d_c_rowptr, int32_t *d_c_colidx,
                         double *d_c_vals, void *stream);
/* out[0] = 0, out[i+1] = in[0] + ... + in[i]  (n inputs, n+1 outputs).
 * d_scratch holds lmg_scan_scratch_count(n) int32 values. */
int64_t lmg_scan_scratch_count(int64_t n);
int lmg_exclusive_scan_i32(int64_t n, const int32_t *d_in, int32_t *d_out, int32_t *d_scratch,
                           void *stream);

/* ---- P1 assembly on triangles (the step before the hot path) ----------------------------------
 * Stiffness (optionally element coefficient d_coeff[e]), mass and load vector (constant load
 * f_const) of linear triangles, node-centric and atomic-free; replaces the element loops of
 * assembly/StiffnessMatrix.py:21-36, MassMatrix.py:21-35, LoadVector.py:20-34.
 *   d_px/d_py[n_nodes], d_conn[3*n_elem] (0-based);  node->element adjacency in element order:
 *   d_n2e_ptr[n_nodes+1], d_n2e_elem[], d_n2e_loc[] (local vertex 0..2 of the node in that element);
 *   d_rowptr/d_colidx: the CSR pattern of the mesh graph (node + neighbours), shared by A and M.
 * Any of d_a_vals / d_m_vals / d_rhs may be NULL. */
int lmg_p1_assemble_2d(int64_t n_nodes, const double *d_px, const double *d_py, const int32_t *d_conn,
                       const int32_t *d_n2e_ptr, const int32_t *d_n2e_elem, const int32_t *d_n2e_loc,
                       const double *d_coeff, double f_const, const int32_t *d_rowptr,
                       const int32_t *d_colidx, double *d_a_vals, double *d_m_vals, double *d_rhs,
                       void *stream);

/* ---- 1-D L2-projection coupling operator (the step before the hot path) -------------------------
 * B[i][j] = integral of fine basis i x coarse basis j over the intersections of fine and coarse elements, by
 * the reference's 3-point rule; replaces Intersection.find_intersections1d (Intersection.py:60-76, an
 * O(ne * ne_c) double loop) + CouplingOperator.compute_b_1d (CouplingOperator.py:32-69) and the row
 * scalings of L2Projection.py:74-90.  d_xf / d_xc: strictly increasing node coordinates (nf, nc >= 2).
 * lmg_l2_coupling_count: entries per row; the caller scans them into d_rowptr (lmg_exclusive_scan_i32);
 * lmg_l2_coupling_fill writes sorted CSR rows.  kind 0: B; 1: "pseudo" Q = diag(colsum M_fine)^-1 B;
 * 2: "quasi" Q = B / rowsum(B).  ("L2", Q = M^-1 B, is dense and stays on the host.) */
int lmg_l2_coupling_count(int64_t nf, int64_t nc, const double *d_xf, const double *d_xc, int32_t *d_rownnz,
                          void *stream);
int lmg_l2_coupling_fill(int kind, int64_t nf, int64_t nc, const double *d_xf, const double *d_xc,
                         const int32_t *d_rowptr, int32_t *d_colidx, double *d_vals, void *stream);

/* ---- hipGraph capture of a launch sequence (one V-cycle) -----------------------------
 * begin/end bracket launches issued on `stream`; end returns an opaque executable graph. */
int lmg_graph_begin(void *stream);
int lmg_graph_end(void *stream, void **graph_exec_out);
int lmg_graph_launch(void *graph_exec, void *stream);
int lmg_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* LMG_H */

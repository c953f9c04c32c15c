// Tile plan of the dense-front path: L as a block-sparse matrix of dense 16 x 16 fp64 tiles.
//
// Replaces, for patterns whose factor is dense (second-order cones of large dimension create dense fronts through
// the two expansion columns per cone, reference src/eicos.cpp:1848-1876), the scalar sliced-ELL programs of plans.hpp:
//   * numeric LDL' (reference ldlt.factorize, src/eicos.cpp:900,1164) = left-looking block factorisation; every
//     update  T_IJ -= L_IK D_K L_JK'  is a 16 x 16 x 16 product = four v_mfma_f64_16x16x4_f64;
//   * the two triangular sweeps (reference ldlt.solve, :1477,1599) = dense 16 x 16 tile mat-vecs, level-scheduled over
//     the BLOCK dependency graph (an order of magnitude fewer levels than the scalar elimination tree).
// Blocks hold <= 16 consecutive elimination positions (Symbolic::blk_ptr) and are padded to 16: padding nodes are
// identity rows (D = 1, no coupling), so kernels never see ragged tiles.  KKT-space vectors live in the padded order:
// slot(node k of block b at offset o) = 16 b + o.
#pragma once
#include <cstdint>
#include <vector>

#include "device_types.hpp"
#include "symbolic.hpp"

namespace eicos {

struct TilePlan {
    int nb = 0, nt = 0, nblev = 0; // blocks, off-diagonal tiles, block levels
    int n0 = 0;                    // first node on tiles (hybrid: the nodes below keep their scalar slots); 0 in pure tile mode
    int N16 = 0;                   // n0 + 16 * nb: length of the KKT-space vectors on the device (blocks padded to 16)
    std::vector<int> slot;         // elimination position -> slot in the padded order
    std::vector<int> blev_ptr;     // nblev+1: block range per level (copy of Symbolic::blev_ptr)
    // off-diagonal tiles, CSC by block column (rows ascending); tile t = (t_row[t], t_col[t])
    std::vector<int> tc_ptr, t_row, t_col;
    // CSR view for the forward sweep: tiles of block row I, ascending block column
    std::vector<int> tr_ptr, tr_tile;
    // factor program.  Targets in execution order, level by level, longest pair list first inside a level:
    // tgt[q] < nb: diagonal tile of block tgt[q]; otherwise off-diagonal tile tgt[q] - nb.
    std::vector<int> tgt, tgt_lev_ptr /* nblev+1 */, tp_ptr /* ntgt+1 */;
    std::vector<int> pa, pb, pk;   // pair: T -= L(pa) D(pk) L(pb)' with tiles pa = (I,K), pb = (J,K), source block pk = K
    // off-diagonal targets of every level again (second phase: L_IJ = T_IJ Linv_JJ' / D_J), level ranges in fin_lev_ptr
    std::vector<int> fin, fin_lev_ptr;
    // KKT entries -> dense image Kt = [nb diagonal tiles | nt off-diagonal tiles], 256 doubles each, element (r, c) at
    // tile_res(r, c) (device_types.hpp: the MFMA result order); only the lower triangle of diagonal tiles is filled.
    // Per permuted entry: L entry e (CSC order of Symbolic) -> Le_img[e]; diagonal j -> D_img[j]; padding diagonals (value 1)
    std::vector<int> Le_img, D_img, pad_img;
    // scalar entry of L (CSC e) -> its tile value position: tile id and in-tile (r, c); for the debug hooks
    std::vector<int> Le_tile, Le_rc;   // Le_rc = 16 r + c
    std::vector<int> ident;            // per block: 1 = its diagonal tile of L is the identity (no coupling inside the block)
    int64_t npairs = 0;
};

// Per-wavefront schedules of the two tile sweeps for a workgroup of NW wavefronts: the blocks of a level are dealt to the
// wavefronts longest-first, and every wavefront gets ONE flat list of tile operations per level -- the off-diagonal tiles of
// its blocks, each block closed by its diagonal operation -- so that its loads can run ahead across block boundaries.
// op = {tile id (diagonal op: the block), vector block the tile multiplies, block being accumulated, flags}
// flags TOP_DIAG / TOP_IDENT: device_types.hpp
// A block whose tile list is much longer than a wavefront's share of the level (the top of the block tree: one or two blocks per level, up
// to 48 tiles each on the dense-front config while seven wavefronts wait) is SPLIT into parts dealt like blocks; a part ends with a
// TOP_PART operation (partial sum -> LDS slot) and the block's diagonal operation moves to a second phase of the level that adds the
// partial sums in slot order (device_types.hpp).  ptr has two segments per (level, wave): phase 0 (tiles, parts, whole blocks), phase 1
// (the split blocks' diagonal operations); split[level] = 1 when phase 1 is not empty.
struct TileSweeps {
    int NW = 0;
    int serial = 0;                    // 1: a small block system, swept by wavefront 0 alone as one flat list (tiles.cpp)
    std::vector<int> fops, bops;       // 4 ints per op
    std::vector<int> fptr, bptr;       // [nblev * 2 * NW + 1]: op range of (level, phase, wave), levels in sweep order (forward: up, backward: down)
    std::vector<int> fend, bend;       // [nblev * 2 * NW]: end of the REAL operations of that range (what follows, up to the next range, is padding)
    std::vector<int> fsplit, bsplit;   // [nblev]
};
TileSweeps build_tile_sweeps(const TilePlan &T, int NW, int pf /* list lengths are padded to a multiple of pf */, int serial_max = 0 /* tiles + blocks up to which the system is swept serially */);

// Per-wavefront schedule of the factorisation's accumulation phase (T_IJ = K_IJ - sum_K L_IK D_K L_JK'): the targets of a
// level are dealt to the wavefronts (most pairs first) and every wavefront gets ONE flat list of operations per level -- per
// target an INIT operation (the target's K tile) followed by its pairs, the last one flagged END -- so that the loads of the
// next TILE_FPF operations are in flight across target boundaries (one operation = two tiles + the D values of a block).
// op = {tile A (INIT: the target's tile in the K image), tile B, source block K, flags | target << FOP_SHIFT}
struct TileFactorOps {
    std::vector<int> ops;  // 4 ints per op
    std::vector<int> ptr;  // [nblev * NW + 1]
};
// img_zero (optional, [nb + nt]): 1 = that tile of the K image holds no KKT entry (a pure fill tile): its INIT operation carries FOP_ZERO
TileFactorOps build_tile_factor_ops(const TilePlan &T, int NW, int pf, const std::vector<char> *img_zero = nullptr);

TilePlan build_tile_plan(const Symbolic &S);

} // namespace eicos

// Internal context of libwaiwera_hip.so: device-resident mesh, fluid state, BCSR Jacobian,
// preconditioner schedule and Krylov work vectors.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/waiwera_hip.h"
#include "physics.hip.h"

namespace wai {

// Element (s, r, k) of block row i in a block-ELL value array of n block rows (layout rationale:
// kernels_linalg.hip, "Matrix entry addressing").  WAI_ELL_ROWS builds the block-row layout for every bs.
// The stride between the element planes of block sizes >= 3 is not n but n rounded up to 512 doubles (4 KB), and never a
// multiple of 2 MB: how a slot's nine planes fall onto the memory channels depends on it.  MEASURED (C4's SpMV alone,
// tools/micro/spmv3_stride.hip, nine strides x three fresh allocations on one box): stride n = 5 029 280 doubles
// 71.8-72.3 % of HBM peak, a multiple of 2 MB 70.0-72.9 %, 4-KB multiples with 0-132 KB added 71.4-78.2 % (mean 75 %).
// Every array indexed through ell_ix with bs >= 3 is allocated with ell_rows(bs, n) rows (-DWAI_ELL_NO_PAD: stride n).
#if defined(WAI_ELL_NO_PAD) && !defined(WAI_ELL_PLANES)
#define WAI_ELL_PLANES   // the 64-row groups inside a slot need a stride that is a multiple of 64
#endif
__host__ __device__ __forceinline__ size_t ell_ld(size_t n) {
#ifdef WAI_ELL_NO_PAD
  return n;
#else
  size_t ld = (n + 511) & ~(size_t)511;
  if ((ld & 262143) == 0) ld += 512;
  return ld;
#endif
}
__host__ __device__ __forceinline__ size_t ell_rows(int bs, size_t n) {
#ifndef WAI_ELL_ROWS
  if (bs >= 3) return ell_ld(n);
#endif
  return n;
}
// Block sizes >= 3: SELL-64 in slices of eight slots (round 5).  The bs^2 elements of 64 consecutive block rows of slot s
// sit together (rounds 4's per-slot form), and the (up to) eight slots of those 64 rows follow one another: everything a
// wave reads of its 64 rows -- 7 slots x 9 elements x 512 B = 32 KB for the 7-point stencil of 3 x 3 blocks -- is ONE
// contiguous run (the eighth slot's 4.6 KB are a hole nobody reads), where the per-slot form had seven streams 360 MB
// apart and the element planes of rounds 2-3 sixty-three.  What the memory channels make of many concurrent streams
// depends on where the allocation lands: C4's k_spmv<3> ran at 64 % of HBM peak on one box and 74-76 % on others, and
// between 72.6 and 85.5 % in eight processes on one box; SELL-64 measured 76.0 % in ten processes out of ten
// (tools/micro/spmv3_variants.hip, profiles/spmv3_variants_r4.log).  Slot s of block row i, element e = r bs + k:
//     (s / 8) * 8 bs^2 ld  +  (((i / 64) * 8 + s % 8) * bs^2 + e) * 64  +  i % 64,      ld = ell_ld(n)
// so wider systems (ILU(k) fill: W > 8) take further slices of eight.  An array of ONE block per row (the inverted
// pivots) keeps the per-slot form through ell_ix1.  Arrays indexed through ell_ix are allocated with ell_size doubles.
// -DWAI_ELL_SLOTWISE builds round 4's per-slot form, -DWAI_ELL_PLANES one plane per element.
__host__ __device__ __forceinline__ size_t ell_ix1(int bs, size_t n, int r, int k, size_t i) {
#ifndef WAI_ELL_ROWS
  if (bs >= 3) {
#ifdef WAI_ELL_PLANES
    return ((size_t)(r * bs + k)) * ell_ld(n) + i;
#else
    return ((i >> 6) * (size_t)(bs * bs) + (size_t)(r * bs + k)) * 64 + (i & 63);
#endif
  }
#endif
  return ((size_t)r * n + i) * bs + k;
}
__host__ __device__ __forceinline__ size_t ell_ix(int bs, size_t n, int s, int r, int k, size_t i) {
#ifndef WAI_ELL_ROWS
  if (bs >= 3) {
#ifdef WAI_ELL_PLANES
    return ((size_t)((s * bs + r) * bs + k)) * ell_ld(n) + i;
#elif defined(WAI_ELL_SLOTWISE)
    const size_t bb = (size_t)(bs * bs);
    return (size_t)s * bb * ell_ld(n) + ((i >> 6) * bb + (size_t)(r * bs + k)) * 64 + (i & 63);
#else
    const size_t bb = (size_t)(bs * bs);
    return (size_t)(s >> 3) * 8 * bb * ell_ld(n) + ((((i >> 6) << 3) + (size_t)(s & 7)) * bb + (size_t)(r * bs + k)) * 64 + (i & 63);
#endif
  }
#endif
  // (2 x 2 blocks: the two block rows of 64 rows together inside a slot, the same idea, MEASURED without effect -- C3's
  // fused launch 0.5326 / 0.5406 / 0.5387 against 0.5409 / 0.5713 / 0.5388 ms, the 108^3 share 0.0826 against 0.0820;
  // profiles/group2_ab_r4.log -- two planes per slot are few enough)
  return ((size_t)(s * bs + r) * n + i) * bs + k;
}
// doubles of a W-slot array indexed through ell_ix
__host__ __device__ __forceinline__ size_t ell_size(int bs, size_t n, int W) {
#if !defined(WAI_ELL_ROWS) && !defined(WAI_ELL_PLANES) && !defined(WAI_ELL_SLOTWISE)
  if (bs >= 3) return (size_t)((W + 7) / 8) * 8 * bs * bs * ell_ld(n);
#endif
  return (size_t)W * bs * bs * ell_rows(bs, n);
}

enum KClass { KC_EOS = 0, KC_RESIDUAL = 1, KC_JACOBIAN = 2, KC_SPMV = 3, KC_PC_APPLY = 4,
              KC_PC_SETUP = 5, KC_VECTOR = 6, KC_TRANSITIONS = 7, KC_COUNT = 8 };

// device scalars of the Krylov solvers (Krylov::scal) and the reduction slots of the same numbers (Krylov::partials)
enum { S_RHO = 0, S_RHOOLD = 1, S_ALPHA = 2, S_OMEGA = 3, S_BETA = 4, S_D1 = 5, S_D2 = 6,
       S_DP2 = 7, S_RHONEW = 8, S_W2 = 9, S_BREAK = 15, S_H = 16 };

struct Comm;  // RCCL state (comm.cpp)

struct DeviceMesh {
  int n_owned = 0, n_halo = 0, n_bc = 0, n_prim = 0, n_local = 0, n_faces = 0, max_deg = 0;
  double* rock = nullptr;    // SoA 8 x n_local
  double* vol = nullptr;     // n_local
  double* fgeom = nullptr;   // SoA 5 x n_faces: area, d1, d2, d12, g.n
  int* fdir = nullptr;       // permeability direction 1..3
  // ELL cell->face adjacency of owned cells, slot-major: [slot * n_owned + cell]
  int* adj_face = nullptr;   // face*2 + side (side 0: cell is cell 1 of the face), -1 = empty
  int* adj_other = nullptr;  // local index of the cell across the face
  int* adj_blk = nullptr;    // matrix slot of (cell, other) in the cell's block row, -1 for a bc cell
  int* adj_tblk = nullptr;   // matrix slot of (other, cell) in the OTHER cell's block row, -1 where the other cell is no owned row (k_jacobian_sym)
  int* diag_blk = nullptr;   // matrix slot of (cell, cell)
  int* cell_src = nullptr;   // first source in the cell or -1
  int* face_cells = nullptr; // [2 n_faces] (cell 1, cell 2) of every face: flux output only
};

struct Sources {
  int n = 0;
  int* cell = nullptr; int* comp = nullptr; int* next = nullptr;
  double* rate = nullptr; double* enth = nullptr;
  SrcCtl* ctl = nullptr;   // state-dependent controls (wai_set_source_controls), null: none
  double* net = nullptr;   // [2 n] what the source network does to each source (source_network_rate), null: no network
};

// Source network: groups and reinjectors (src/source_network_group.F90, source_network_reinjector.F90),
// evaluated on the host between the EOS sweep and the residual kernel of every residual evaluation,
// as source_network%update is (src/source_network.F90:90-130).  Nodes carry the six flow values of
// source_network_node_type.
struct NetNode { double rate = 0, enth = 0, wrate = 0, wenth = 0, srate = 0, senth = 0; };
struct NetRef { int kind = 0, index = -1; };       // 0 none, 1 source, 2 group, 3 reinjector
struct NetGroup {
  std::vector<NetRef> in;
  int scaling = 0;                                 // 0 uniform, 1 progressive
  int n_limit = 0, limit_type[3] = {0, 0, 0};      // 0 total, 1 water, 2 steam
  double limit[3] = {0, 0, 0};
  SrcCtl sep{};                                    // the group's own separator (sep_hf, sep_hg, sep_more); sep_hg = 0: none
  NetNode node;
};
struct NetOutput {
  int flow = 1;                                    // 1 water, 2 steam
  NetRef out;
  double rate = -1.0, proportion = -1.0, enthalpy = -1.0;
  NetNode node;
};
struct NetReinjector {
  NetRef in;                                       // source / group; none: fed by another reinjector
  std::vector<NetOutput> out;
  NetRef overflow;
  NetNode node;                                    // water / steam rate = capacity (-1 unrated)
  double in_w = 0, in_wh = 0, in_s = 0, in_sh = 0; // input flows
  double out_w = 0, out_s = 0;                     // delivered
  NetNode over;                                    // overflow
  bool fed = false;                                // input set by an upstream reinjector this pass
};
struct Network {
  bool on = false;
  std::vector<NetGroup> groups;
  std::vector<NetReinjector> reinjectors;
  std::vector<int> reinj_order;                    // outputs / overflow targets before their feeders
  std::vector<int> rate_specified, enth_specified; // per source
  std::vector<NetNode> src;                        // per source, last pass
  std::vector<double> h_enth0, h_raw, out_rate, out_enth;
  std::vector<char> is_out;                        // sources a reinjector feeds (last pass)
  std::vector<SrcCtl> h_ctl;                       // host copy of the control records (separators)
  double* d_raw = nullptr;                         // device scratch: raw rates and enthalpies, 2 n
  // A network whose sources live on several ranks (source_network_group.F90:494-515, 579-596: gathers over the
  // group's communicator): every rank holds the whole description, numbered by GLOBAL source index; the sources' own
  // rates are all-gathered (an all-reduce of a vector each rank fills at its own entries) before the pass, which
  // every rank then evaluates identically.  gidx: global index of each local source; empty: one rank, identity
  std::vector<int> gidx;
  int n_global = 0;
  double* d_all = nullptr;                         // [2 n_global] all-reduce buffer
  std::vector<double> h_loc, l_net, l_enth;        // local staging: raw rates (2 n_local), factors (2 n_local), enthalpies
  // Jacobian couplings through the network (flow_simulation_modify_jacobian, src/flow_simulation.F90:3023-3084,
  // dependencies of src/source_network.F90:359-498): blocks E[i][j] = d R(cell i) / d y(cell j) *through the
  // network pass* for the cells of the network's sources (the reference inserts the outer product of its
  // dependency rows and columns; pairs without a dependence difference to exact zeros), by the FD rule of
  // the Jacobian.  The operator is A + E; the preconditioner is built from A.
  std::vector<int> h_cell;                         // cell of every source
  bool coupling = true;                            // wai_set_network_couplings
  bool cp_in_pc = true;                            // ... and inside the preconditioner's ILU pattern (wai_set_network_couplings 2; one rank)
  bool cp_valid = false;                           // E belongs to the Jacobian in force and has a nonzero entry
  std::vector<int> cp_cells;                       // distinct (local) cells of the network's sources, ascending
  std::vector<double> h_cp_val;                    // [ml][m][bs][bs] row-major, ml = cp_cells.size() rows, m columns
  // a network on several ranks: the columns are the network's cells of ALL ranks, ordered by (owner rank, local
  // cell) -- this rank's cells are the columns cp_j0 .. cp_j0 + ml; one rank: m = ml, cp_j0 = 0
  bool cp_span = false;
  int cp_m = 0, cp_j0 = 0;
  std::vector<int> cp_owner;                       // [m] owner rank of every column cell
  int* d_cp_cells = nullptr;
  double *d_cp_val = nullptr, *d_cp_f = nullptr, *d_cp_g = nullptr;
  double* d_cp_x = nullptr;                        // [m * bs + 2] x at the column cells, gathered over the ranks
  void free_device() {
    if (d_raw) (void)hipFree(d_raw);
    if (d_all) (void)hipFree(d_all);
    d_all = nullptr;
    if (d_cp_cells) (void)hipFree(d_cp_cells);
    if (d_cp_val) (void)hipFree(d_cp_val);
    if (d_cp_f) (void)hipFree(d_cp_f);
    if (d_cp_g) (void)hipFree(d_cp_g);
    if (d_cp_x) (void)hipFree(d_cp_x);
    d_raw = d_cp_val = d_cp_f = d_cp_g = d_cp_x = nullptr; d_cp_cells = nullptr;
  }
};

// Block matrix in HBM: block-ELL, slot-major struct-of-arrays ("SELL" with one slice):
//   col[s*n + i]            column of slot s of block row i (padding slots: col = i, values 0)
//   val[(s*bb + e)*n + i]   entry e (row-major inside the bs x bs block) of that block
// Slots of a row are in ascending column order, i.e. slot s of row i is BCSR block
// rowptr[i] + s: the BCSR arrays (rowptr / colidx on the host) remain the exchange format of
// the C ABI, the device layout is what lets one-thread-per-row kernels read and write fully
// coalesced (64 consecutive doubles per wave instruction).
struct Bcsr {
  int n = 0, ncols = 0, nnzb = 0, bs = 0, W = 0;
  int* col = nullptr;
  double* val = nullptr;
  int* rowptr = nullptr;  // device copy of the BCSR row pointer (layout conversion kernels)
  std::vector<int> h_rowptr, h_colidx;
};

// Block-Jacobi ILU(0): one workgroup per subdomain, one thread per block row.
struct IluSchedule {
  int nsub = 0, max_rows = 0, max_lev = 0;
  int* sub_ptr = nullptr;   // nsub+1 row ranges
  int* sub_nlev = nullptr;  // per subdomain: forward levels | backward levels << 16
  int* sub_split = nullptr; // per subdomain: leading rows longer than half the block-ELL width (k_pc_rows: MINC bricks), or null
  int* row_info = nullptr;  // per row: lfirst | dslot<<4 | ulast<<8 | lev_f<<12 | lev_b<<22
  double* fval = nullptr;   // factor in the matrix' block-ELL layout; the diagonal slot holds
                            // the inverted pivot block
  double* dinv = nullptr;   // inverted pivot blocks, SoA [bb][n]
  bool diag_only = false;   // ILU(0) touches no off-diagonal block in any subdomain (== DILU)
  bool scaled = true;       // diag_only: rows pre-scaled by the inverted pivots (WAI_ILU_NOSCALE: off)
  bool park = true;         // k_pc_park: upper blocks parked in LDS (WAI_PC_PARK=0: off)
  int* row_uoff = nullptr;  // first parked upper block of a row inside its subdomain
  int* row_tslot = nullptr; // per row: slot of A_ki in row k for each of its (<= 4) in-subdomain lower couplings k, 4 bits each (15: none)
  int max_nl = 0;           // most in-subdomain lower couplings of any row
  bool park2 = false;
  int* sub_int = nullptr;   // subdomains none of whose rows has a partition-ghost column ...
  int* sub_bnd = nullptr;   // ... and the others (device lists; null on a single rank)
  int n_int = 0, n_bnd = 0;
  int* sub_order = nullptr; // launch order of all subdomains when they differ in cost (ragged bricks): inside each XCD's
                            // contiguous eighth the long ones first, so the short ones make the tail; null: uniform
  int max_ublocks = 0;      // most in-subdomain upper blocks of any subdomain
  bool fast3 = false;         // <= 3 lower and <= 3 upper in-subdomain couplings per row, offsets < 4
  int max_nlu = 0;            // most lower or upper in-subdomain couplings of any row
  bool rows_kernel = false;   // k_pc_rows (one thread per scalar row) applies and is selected
  bool wave_kernel = false;   // k_pc_wave (one wave per brick of <= 64 block rows) applies and is selected
  int* row_uoffw = nullptr;   // first parked upper block of a row inside its subdomain, all (<= 4) uppers counted
  int max_ublocks_w = 0;
  // Brick-local 16-bit column indices for k_pc_park (2 x 2 blocks): entry = segment << 13 | offset, column = the brick's
  // sub_seg[segment] + offset.  Segment 0 starts at the brick's own first row; the others cover what its rows reach in
  // other bricks and among the ghost columns, windows of 8192 columns each (a 16 x 16 x 2 brick of a structured mesh:
  // its six neighbour bricks).  Null when some brick would need more than 8 segments: the int32 planes serve then.
  unsigned short* col16 = nullptr;   // [n][8]: the (<= 8) slots of a row together, 16 bytes -- one load per row instead of one per slot
  int* sub_seg = nullptr;            // [nsub][8]
  bool level_sorted = false;  // every subdomain's rows are stored in dependency-level order (forward levels non-decreasing,
                              // backward levels non-increasing with the row index)
  bool factored = false;
  // subdomains of more than 1024 rows ("one block per rank", sub_ptr = NULL, is the reference's
  // PCBJACOBI / PCASM default): rows of equal dependency level are independent across all
  // subdomains, so the factorisation and the two substitutions run as one launch per level over
  // the rows of that level (stored factor, unfused)
  bool big = false;
  int nlev_f = 0, nlev_b = 0;
  int* ord_f = nullptr;       // rows sorted by forward level, ...
  int* ord_b = nullptr;       // ... by backward level
  std::vector<int> lev_f_ptr, lev_b_ptr;   // host: row ranges of each level in ord_f / ord_b
  bool built = false;
};

// PCASM (restricted additive Schwarz) system: every subdomain's overlapped row set is stored as
// its own block of an extended matrix E (couplings leaving the set dropped), so the block-Jacobi
// machinery applies to E unchanged: gather r -> r_ext, ILU(0) solve per block, scatter the owned
// rows back (src/timestepper.F90:1668-1669,1753-1757; PETSc PCASM defaults: overlap 1, restrict)
struct AsmSystem {
  int overlap = -1;           // what E was built for (-1: not built; 0: no overlap, fill only)
  int levels = 0;             // ILU(k) fill levels E's pattern carries
  int n_ext = 0;
  Bcsr E;                     // block-ELL over the n_ext rows, columns in ext numbering
  IluSchedule sched;
  int* ext_row = nullptr;     // [n_ext] row of the global system, bit 31 set: owned by this block
  int* gmap = nullptr;        // [W * n_ext] plane position (slot * n + row) of the source block in J, -1: none
  double* r_ext = nullptr;    // [bs * n_ext] gathered right-hand side / solution
  // overlap across rank boundaries: the matrix rows of the partition-ghost cells, received from
  // their owners at every set-up (block-ELL over the n_halo cells, the sender's slot order), and the residual
  // with its ghost entries filled by one more halo exchange per application
  // the source network's blocks inside the factor's pattern (src/flow_simulation.F90:3023-3084 widens the BAIJ pattern
  // PETSc factors): E's pattern carries the pairs of network cells of one subdomain, net_pos / net_pair name where each
  // block of the network's E goes (plane position t * n_ext + q; pair = row * m + column in the coupling array)
  bool with_net = false;
  int n_net = 0;
  int* net_pos = nullptr;
  int* net_pair = nullptr;
  bool cross = false;
  double* hval = nullptr;     // [W * bs * bs * n_halo]
  double* r_full = nullptr;   // [bs * n_prim]
};

// Residual form of the time stepping method (src/timestepper.F90:345-452), by value to kernels
struct ResForm {
  int method = 0;             // WAI_METHOD_BEULER | BDF2 | DIRECTSS
  double dt = 0.0, ratio = 0.0;
  const double* last = nullptr;   // lhs at the start of the step
  const double* last2 = nullptr;  // BDF2: lhs one step further back
};

// Passive tracers (src/tracer.F90:30-40) and the auxiliary linear problem's solver settings
// (timestepper.F90:2021-2022, 2061-2064: gmres + bjacobi unless configured)
constexpr int MAX_TRACERS = 8;
constexpr int POST_OFF = 64;   // h_scal[POST_OFF ..+2]: {(R,R), 8 * sequence number + code, check word} posted by the device (post_scalars)
constexpr unsigned long long POST_KEY = 0x5bd1e995a5a5c3c3ull;   // check = bits((R,R)) ^ bits(tag) ^ POST_KEY: zeroed memory never verifies
struct Tracers {
  int nt = 0;
  int phase[MAX_TRACERS] = {0};
  double decay[MAX_TRACERS] = {0}, activation[MAX_TRACERS] = {0}, diffusion[MAX_TRACERS] = {0};
  double* bc = nullptr;    // [n_bc][nt] Dirichlet mass fractions
  double* inj = nullptr;   // [n_sources][nt] injection rates
  double* val = nullptr;   // scalar block-ELL values of the system being solved, W x n
  int ksp_type = 1, restart = 30, max_its = 10000;
  double rtol = 1.e-5, atol = 1.e-50;
};

// one tracer's system: which tracer, and the method's combination (timestepper.F90:458-581)
struct TracerForm {
  int method, it, nt, phase;
  double dt, ratio, decay, activation, diffusion;
};

// Finalisation of a producer kernel's partial sums inside its own launch (fin_block, kernels_linalg.hip): one
// extra workgroup waits for the partials of `nslots` consecutive reduction slots, sums them into the device
// scalars -- in the order k_finalize sums them --, derives the BiCGStab scalars of `phase` and, when asked,
// posts the scalars to the pinned host mirror.  Replaces a one-block k_finalize launch (and the 128-byte
// copy) behind every producer.
constexpr int FIN_MAXF = 64;   // most finaliser workgroups of a launch (second-level partials per slot)
struct Fin {
  int count = 0;               // workgroups of this launch that store partials; 0: no finalisation here (no extra workgroup)
  int nb = 0;                  // partials per slot to sum (an earlier launch may have left some of them)
  int nf = 1;                  // finaliser workgroups: the last nf of the grid, each sums one slice of the partials (fin_slices)
  double* part2 = nullptr;     // [slots][FIN_MAXF] slice sums on their way to the last finaliser
  int slot0 = 0, nslots = 0;
  int phase = -1;              // derive_scalars phase, -1: sums only
  int seq = 0;                 // > 0: post (R,R) and the breakdown code with this sequence number to `post`
  double* scal = nullptr;
  double* post = nullptr;      // device address of the pinned host mirror (16 bytes, 16-byte aligned)
};

struct Krylov {
  int n = 0, nl = 0;           // bs*n_owned, bs*n_prim
  double* d_post = nullptr;    // device address of h_scal + POST_OFF
  int seq = 0;                 // last sequence number handed out
  long long n_launch = 0, n_copy = 0;   // kernels launched / copies enqueued by the Krylov helpers (wai_launch_stats)
  double *R = nullptr, *RP = nullptr, *P = nullptr, *V = nullptr, *S = nullptr, *T = nullptr,
         *tmp = nullptr, *X = nullptr;
  double* bl = nullptr;        // BiCGStab(L): r_0..r_L, u_0..u_L, r~ (allocated on first use)
  double* basis = nullptr;     // GMRES: (m+1) vectors of nl
  int basis_m = 0;
  double* partials = nullptr;  // [slots][nb_max]
  double* partials2 = nullptr; // [slots][FIN_MAXF]: slice sums of the finaliser workgroups (fin_block)
  unsigned* started = nullptr; // k_bcgs_xrp<DERIVE>: workgroups of the launch that have read their scalars (device counter, zero between launches)
  bool alpha_pending = false;  // several ranks: alpha = rho / (V,rP) is to be derived by the next pack_halo_axpy launch (no scalar kernel)
  int nb_max = 0;
  double* scal = nullptr;      // device scalars
  double* h_scal = nullptr;    // pinned host mirror
  int nblocks = 0;
  int nb_pc = 0;               // partial-sum blocks the last preconditioner application left per slot
};

}  // namespace wai

namespace wai {
// PCLU / sub-preconditioner "lu" (src/timestepper.F90:1749-1750; the reference lists it "for testing
// purposes"): exact solves of the preconditioner blocks.  Each block's dense inverse, formed on the
// host with partial pivoting at every set-up, applied on the device as one dense product per block.
struct LuBlocks {
  double* inv = nullptr;      // concatenated dense inverses, block b at inv_ptr[b], row-major m_b x m_b
  size_t* inv_ptr = nullptr;  // device, nsub + 1
  std::vector<size_t> h_inv_ptr;
  size_t total = 0;
};
int launch_lu_apply(wai_ctx* c, const double* r, double* z);
}  // namespace wai

struct wai_ctx {
  int device = 0;
  int n_cu = 256;               // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  size_t lds_per_block = 64 * 1024;   // LDS a workgroup may ask for (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KB on gfx950)
  hipStream_t stream = nullptr;
  int kind = 0, np = 0, df = 0;
  wai::EosParams ep{};
  wai_solver_opts opts{};
  wai::DeviceMesh mesh;
  wai::Sources src;
  wai::Network net;
  std::vector<int> src_gidx;    // wai_set_source_global_index: global index of every local source (networks across ranks)
  int src_nglobal = 0;
  wai::Bcsr J;
  wai::IluSchedule ilu;
  wai::AsmSystem as;
  wai::AsmSystem as_aux;   // the extended system of the scalar (tracer) problems, block size 1 (AuxScope swaps it in)
  wai::LuBlocks lu;
  wai::Krylov ks;
  wai::Tracers tr;
  // fluid state, SoA df x n_local each; perturbed states np x df x n_prim
  double *flu = nullptr, *flu_last_iter = nullptr, *flu_last_step = nullptr, *flu_pert = nullptr;
  double* hstep = nullptr;      // FD steps np x n_prim (interleaved like y)
  // work vectors (interleaved [cell][bs], nl entries)
  double *w_y = nullptr, *w_yold = nullptr, *w_delta = nullptr, *w_f = nullptr, *w_lhs = nullptr,
         *w_a = nullptr, *w_b = nullptr, *w_c = nullptr;
  int* d_flags = nullptr;       // [0] err, [1] first bad cell, [2] changed_y, [3] changed_search
  int* h_flags = nullptr;       // pinned
  double* d_red = nullptr;      // reduction scratch
  double* h_red = nullptr;      // pinned
  double* stage[4] = {nullptr, nullptr, nullptr, nullptr};  // host-vector staging
  size_t stage_len = 0;
  wai::Comm* comm = nullptr;
  hipEvent_t ev_scal = nullptr;   // marks the scalar read-back of a Krylov iteration (ksp_bcgs)
  // halo exchange overlapped with the preconditioned operator on the bricks that touch no ghost
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  // WAI_FACE_STREAM=1: the face bricks' launch of the overlapped halo exchange on a stream of its own, ordered behind the
  // unpack only (round 5; measured slower than behind the interior bricks on the compute stream: krylov.hip launch_pc_split)
  hipStream_t face_stream = nullptr;
  hipEvent_t ev_face = nullptr, ev_prior = nullptr;
  // run-time switches of the fused launches, read from the environment once per solve / set-up / probe (read_env),
  // not per launch: WAI_FIN_SEPARATE, WAI_PC_STAGGER (-1: each kernel's default), WAI_WAVE_ROWPTR, WAI_NO_COL16 (k_pc_park on the int32 column planes)
  struct EnvSw { bool fin_separate = false; int stagger = -1; bool wave_rowptr = false; bool no_col16 = false; int stage = -1; bool no_face_stream = false; bool scalar_kernels = false; } env;
  int test_drop_wait = 0;   // fault injection (wai_test_drop_stream_wait): 1 the face bricks' launch does not wait for the halo
  // halo
  int n_nbr = 0;
  std::vector<int> nbr_rank, send_ptr, recv_ptr;
  int* d_send_idx = nullptr; double* d_sendbuf = nullptr; double* d_recvbuf = nullptr;
  int send_total = 0, max_dof_buf = 0;
  // Newton bookkeeping
  double fnorm0 = 0.0;
  // time stepping method: residual form in force, and wai_timestep's own BDF2 history
  int method = 0;               // residual form (wai_set_residual_form)
  double ratio = 0.0;
  double* w_lhs2 = nullptr;     // lhs two steps back, as handed to wai_set_residual_form
  int scheme = 0, taken = 0;    // wai_set_timestep_method, accepted wai_timestep calls since
  double dt_last = 0.0;
  double* w_hist = nullptr;     // lhs at the start of the last accepted wai_timestep
  double* w_hist_prev = nullptr;  // ... and of the one before, to undo an acceptance
  double dt_last_prev = 0.0;
  bool can_reject = false;      // the last wai_timestep converged and has not been rejected
  // measurement
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool prof_on = false;
  bool last_iter_partial = false;   // flu_last_iter holds only the transition sweep's planes (do_newton_step)
  double prof_ms[wai::KC_COUNT] = {0};
  long long prof_n[wai::KC_COUNT] = {0};
  hipEvent_t pev0 = nullptr, pev1 = nullptr;
  std::string err;
  bool bc_set = false;
  int dbg = 0;                  // timing probes (wai_bench_kernel)
};

// ---- kernel launchers (kernels_assembly.hip / kernels_linalg.hip) --------------------------
namespace wai {
int launch_eos(wai_ctx* c, const double* y, int first, int count, bool perturbed);
int launch_residual(wai_ctx* c, double dt, const double* lhs_old, double* f, double* lhs_out,
                    double* rhs_out, const int* only = nullptr, int n_only = 0);   // only: these rows alone (device list)
int launch_jacobian(wai_ctx* c, double dt, const double* lhs_old);
int launch_transitions(wai_ctx* c, const double* y_old, double* search, double* y);
// tracer system of tf.it on the flow Jacobian's pattern: values -> c->tr.val, rhs -> b
int launch_tracer_assemble(wai_ctx* c, const TracerForm& tf, const double* alx_last,
                           const double* alx_last2, double* b);
int launch_tracer_lhs(wai_ctx* c, double* Al);
int launch_separator(wai_ctx* c, double pressure, double* out);   // out[3] on the device: hf, hg, err
int launch_face_fluxes(wai_ctx* c, const int* face_cells, double* out);   // [face][np + nmob]
int launch_source_rates(wai_ctx* c, double* out, bool raw = false);   // out[0..n) rates, out[n..2n) enthalpies (device); raw: before the network pass
// X[cell][nt] <-> x[cell] of tracer it; alx = Al o X
int launch_tracer_pick(wai_ctx* c, const double* X, int it, double* x);
int launch_tracer_put(wai_ctx* c, const double* x, int it, double* X);
int launch_tracer_alx(wai_ctx* c, const double* X, double* alx);
int launch_max_scaled(wai_ctx* c, const double* v, const double* scale, double tol, double* val,
                      int* idx);
int launch_fluid_aos(wai_ctx* c, const double* flu_soa, double* out_aos);
int launch_region_get(wai_ctx* c, double* out);  // regions as doubles, n_prim
int launch_region_set(wai_ctx* c, const double* in, int first, int count);

int launch_spmv(wai_ctx* c, const double* x, double* y);
int launch_ilu_factor(wai_ctx* c);
// the same on any (matrix, schedule) pair: the Jacobian with the brick schedule, or the extended
// ASM system with its own
int launch_ilu_factor_on(wai_ctx* c, const Bcsr& M, IluSchedule& s);
// in2 (optional, fused kernels that can: pc_axpy_capable): the input is in - alpha in2, alpha = the device scalar S_ALPHA
int launch_pc_on(wai_ctx* c, const Bcsr& M, const IluSchedule& s, bool spmv, const double* in, double* z,
                 int dot_mode, const double* aux, const int* list = nullptr, int nrun = 0, const Fin* fin = nullptr,
                 const double* in2 = nullptr);
bool pc_axpy_capable(const wai_ctx* c);
bool pc_axpy_default(const wai_ctx* c);   // is the composed second launch the default for the kernel in force (k_pc_park with col16)
// subdomains of any size: level-by-level launches, in place on z (z = r on entry)
int launch_big_solve(wai_ctx* c, const Bcsr& M, const IluSchedule& s, double* z);
int launch_asm_gather_matrix(wai_ctx* c);                   // E.val <- J.val (and the ghost cells' rows; + the network's blocks)
int launch_pack_rows(wai_ctx* c);                           // d_sendbuf <- matrix rows of the cells sent to neighbours
int launch_unpack_rows(wai_ctx* c);                         // as.hval <- d_recvbuf
int launch_asm_gather(wai_ctx* c, const double* r);        // as.r_ext <- r
int launch_asm_scatter(wai_ctx* c, double* z);             // z[owned] <- as.r_ext
// up to two dot products (a1,b1) -> slot1, (a2,b2) -> slot2 (a2 null: one); partial blocks in ks.nb_pc
int vec_dots(wai_ctx* c, const double* a1, const double* b1, int slot1, const double* a2, const double* b2,
             int slot2, int n);
// z = B^-1 r (spmv = false) or z = B^-1 (A x) (spmv = true: x is `in`, haloed by the caller).
// dot_mode 0: none; 1: scal-partials S_D1 += (z, aux); 2: S_D1 += (in, z), S_D2 += (z, z);
// 3: S_DP2 += (z, z)
// list / nrun: run only the listed subdomains (null: all)
// fin (optional): finalise the dot products in the kernel's last workgroup instead of a k_finalize launch
int launch_pc(wai_ctx* c, bool spmv, const double* in, double* z, int dot_mode, const double* aux,
              const int* list = nullptr, int nrun = 0, const Fin* fin = nullptr, const double* in2 = nullptr);
// finalisation descriptor for slots [slot0, slot0 + nslots) (the launcher fills in the workgroup counts);
// post: mirror the scalars to the host with a fresh sequence number (left in ks.seq)
Fin make_fin(wai_ctx* c, int slot0, int nslots, int phase, bool post = false);
int launch_ell_to_bcsr(wai_ctx* c, const double* ell, double* bcsr);
int launch_bcsr_to_ell(wai_ctx* c, const double* bcsr, double* ell);
// reductions: partial sums live in ks.partials[slot][block]; finalize sums nb partials of
// nslots consecutive slots into ks.scal and (phase >= 0) derives the BiCGStab scalars
int vec_finalize(wai_ctx* c, int nb, int slot0, int nslots, int phase);
int vec_dot(wai_ctx* c, const double* a, const double* b, int n, int slot);
int partials_clear(wai_ctx* c, int slot0, int nslots);   // reduction slots emptied (FIN_EMPTY)
int vec_copy(wai_ctx* c, double* dst, const double* src, size_t n);
int vec_zero(wai_ctx* c, double* dst, size_t n);
int vec_waxpy(wai_ctx* c, double* w, double alpha, const double* x, const double* y, int n);
int bcgs_scalars(wai_ctx* c, int phase, bool post = false);
int bcgs_update_xrp_derive(wai_ctx* c);
int bcgs_post(wai_ctx* c, int seq);
void read_env(wai_ctx* c);   // the launch switches above (kernels_linalg.hip)
int test_drop_partials(wai_ctx* c, int n);   // fault injection (tests): workgroup 0 loses its next n partial sums
int bcgs_update_p(wai_ctx* c);
int bcgs_update_s(wai_ctx* c);
// dots: reduces (R,R), (R,RP) into S_DP2, S_RHONEW and (fin_phase >= -1) finalises them in its last workgroup
int bcgs_update_xr(wai_ctx* c, bool dots = true, int fin_phase = -2, bool post = false);
// S = R - alpha V re-formed; X += alpha P + omega S; R = S - omega T; P = R + beta (P - omega V): one pass, no reduction
int bcgs_update_xrp(wai_ctx* c);
int pack_halo_axpy(wai_ctx* c, const double* a, const double* b, int dof, hipStream_t stream = nullptr);
int gmres_mdot(wai_ctx* c, const double* w, int k);          // scal[16+i] = (w, v_i), i<k
int gmres_maxpy_norm(wai_ctx* c, double* w, int k);          // w -= sum h_i v_i ; scal[8] = |w|^2
int gmres_scale_to(wai_ctx* c, double* dst, const double* src, int slot_norm2, int n);
int gmres_update_x(wai_ctx* c, double* x, const double* ycoef_host, int k);
int pack_halo(wai_ctx* c, const double* vec, int dof, hipStream_t stream = nullptr);
int unpack_halo(wai_ctx* c, double* vec, int dof, hipStream_t stream = nullptr);
}  // namespace wai

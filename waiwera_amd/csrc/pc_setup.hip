// Preconditioner set-up of libwaiwera_hip.so: the symbolic phase of block-Jacobi ILU(0) on the brick subdomains
// (dependency levels, kernel selection), the extended systems of PCASM and ILU(k) (overlapped row sets across rank
// boundaries, level-of-fill patterns), the dense block inverses of PCLU, and the numeric set-up that PCSetUp stands for
// (src/timestepper.F90:1645-1836).
#include "host.hpp"

using namespace wai;

namespace wai {

// Symbolic phase of block-Jacobi ILU(0) on a block matrix given as host CSR (ascending columns):
// per-row slot ranges inside the row's subdomain, dependency levels of both substitutions, whether
// ILU(0) ever touches an off-diagonal block (if not it is DILU and the fused kernels apply), the
// compact / parked kernel conditions -- or, for subdomains of more than 1024 rows, the level sets
// of the launch-per-level path.  `ghosts`: rows may have columns >= n (partition ghosts).
int build_schedule(wai_ctx* c, IluSchedule& s, const std::vector<int>& rowptr, const std::vector<int>& colidx,
                   const std::vector<int>& sub, int N, int W, int np, bool ghosts) {
  s.nsub = (int)sub.size() - 1;
  if (sub.front() != 0 || sub.back() != N) { c->err = "sub_ptr must cover [0, n_owned]"; return -2; }
  std::vector<int> diag(N);
  for (int i = 0; i < N; i++) {
    const int* row = colidx.data() + rowptr[i];
    diag[i] = (int)(std::lower_bound(row, row + (rowptr[i + 1] - rowptr[i]), i) - row);
  }
  std::vector<int> info(N), uoff(N, 0), uoffw(N, 0), levf(N), levb(N), nlev(s.nsub, 0), lfirst(N), ulast(N), tslot(N, 0);
  int max_nu = 0;
  s.max_rows = 0; s.max_lev = 0; s.max_ublocks = 0; s.max_ublocks_w = 0; s.max_nlu = 0; s.max_nl = 0;
  bool offdiag_fill = false, fast3 = true;
  int nlf_all = 0, nlb_all = 0;
  for (int sd = 0; sd < s.nsub; sd++) {
    const int lo = sub[sd], hi = sub[sd + 1];
    if (hi < lo) { c->err = "sub_ptr not monotone"; return -2; }
    s.max_rows = std::max(s.max_rows, hi - lo);
    int nlf = 0, nlb = 0;
    for (int i = lo; i < hi; i++) {
      const int* row = colidx.data() + rowptr[i];
      const int cnt = rowptr[i + 1] - rowptr[i];
      int ls = 0;
      while (ls < cnt && row[ls] < lo) ls++;
      int ue = cnt;
      while (ue > 0 && row[ue - 1] >= hi) ue--;
      lfirst[i] = ls; ulast[i] = ue;
      int lv = 0;
      for (int q = ls; q < diag[i]; q++) lv = std::max(lv, levf[row[q]] + 1);
      levf[i] = lv;
      nlf = std::max(nlf, lv + 1);
    }
    for (int i = hi - 1; i >= lo; i--) {
      const int* row = colidx.data() + rowptr[i];
      int lv = 0;
      for (int q = diag[i] + 1; q < ulast[i]; q++) lv = std::max(lv, levb[row[q]] + 1);
      levb[i] = lv;
      nlb = std::max(nlb, lv + 1);
    }
    // does the IKJ elimination ever update an off-diagonal block of a row in this subdomain?
    for (int i = lo; i < hi && !offdiag_fill; i++) {
      const int* row = colidx.data() + rowptr[i];
      for (int q = lfirst[i]; q < diag[i] && !offdiag_fill; q++) {
        const int k = row[q];
        const int* rk = colidx.data() + rowptr[k];
        for (int r2 = diag[k] + 1; r2 < ulast[k]; r2++) {
          const int j = rk[r2];
          if (j == i) continue;
          if (std::binary_search(row + q + 1, row + ulast[i], j)) { offdiag_fill = true; break; }
        }
      }
    }
    // per in-subdomain lower coupling (i, k): the slot of row k that holds A_ki (15: structurally absent), four
    // bits each -- the pivot recurrence reads A_ki without chasing row k's descriptor and columns
    for (int i = lo; i < hi; i++) {
      const int* row = colidx.data() + rowptr[i];
      int pack = 0;
      for (int q = lfirst[i], p = 0; q < diag[i] && p < 4; q++, p++) {
        const int k = row[q];
        const int* rk = colidx.data() + rowptr[k];
        const int* e = std::lower_bound(rk + diag[k] + 1, rk + ulast[k], i);
        const int r2 = (e < rk + ulast[k] && *e == i) ? (int)(e - rk) : 15;
        pack |= (r2 & 15) << (4 * p);
      }
      tslot[i] = pack;
      s.max_nl = std::max(s.max_nl, diag[i] - lfirst[i]);
    }
    int ucount = 0, ucountw = 0;
    for (int i = lo; i < hi; i++) {
      const int nL = diag[i] - lfirst[i], nU = ulast[i] - diag[i] - 1;
      if (nL > 3 || nU > 3 || lfirst[i] > 3 || diag[i] > 3) fast3 = false;
      s.max_nlu = std::max(s.max_nlu, std::max(nL, nU));
      uoff[i] = ucount;
      ucount += std::min(nU, 3);
      uoffw[i] = ucountw;
      ucountw += nU;
      max_nu = std::max(max_nu, nU);
    }
    s.max_ublocks = std::max(s.max_ublocks, ucount);
    s.max_ublocks_w = std::max(s.max_ublocks_w, ucountw);
    nlev[sd] = (nlf & 0xffff) | (nlb << 16);
    s.max_lev = std::max(s.max_lev, std::max(nlf, nlb));
    nlf_all = std::max(nlf_all, nlf); nlb_all = std::max(nlb_all, nlb);
  }
  // the brick kernels hold a row's <= 8 blocks in registers and pack slot numbers in 4 bits: wider rows (ILU(k)
  // fill) and subdomains of more than 1024 rows take the launch-per-level path, whose descriptor has 8-bit slots
  s.big = s.max_rows > 1024 || W > 8;
  if (!s.big && s.max_lev > 1023) { c->err = "more than 1023 dependency levels in a subdomain"; return -2; }
  for (int i = 0; i < N; i++)
    info[i] = s.big ? (lfirst[i] | (diag[i] << 8) | (ulast[i] << 16))
                    : (lfirst[i] | (diag[i] << 4) | (ulast[i] << 8) | (levf[i] << 12) | (levb[i] << 22));
  // Launch order.  Workgroup b of a fused launch runs on XCD b % 8 and takes position (b & 7) * per + (b >> 3) of the
  // list it is given, so each XCD works through one contiguous eighth in order.  Where bricks differ in cost (the
  // ragged bricks at the upper ends of a rank's box: fewer rows, fewer levels) the long ones go first inside each
  // eighth and the short ones last: a launch ends with its shortest workgroups (the tail of 2646 bricks on 768 slots
  // at 108^3 is a fifth of the launch).  The eighths themselves stay contiguous -- an XCD's L2 keeps serving the
  // neighbour bricks' vector entries.
  auto brick_cost = [&](int sd) { return ((nlev[sd] & 0xffff) + (nlev[sd] >> 16)) * 4096 + (sub[sd + 1] - sub[sd]); };
  auto lpt_order = [&](std::vector<int>& list) {
    const int n = (int)list.size(), per = (n + 7) >> 3;
    for (int j = 0; j < 8; j++) {
      const int a = std::min(j * per, n), b = std::min((j + 1) * per, n);
      std::stable_sort(list.begin() + a, list.begin() + b, [&](int x, int y) { return brick_cost(x) > brick_cost(y); });
    }
  };
  if (!s.big) {
    bool uniform = true;
    for (int sd = 1; sd < s.nsub && uniform; sd++) uniform = brick_cost(sd) == brick_cost(0);
    if (!uniform) {
      std::vector<int> order(s.nsub);
      std::iota(order.begin(), order.end(), 0);
      lpt_order(order);
      if (dev_upload(c, &s.sub_order, order)) return -1;
    }
  }
  if (ghosts) {   // subdomains without / with partition-ghost columns (for the overlapped halo exchange)
    std::vector<int> li, lb;
    for (int sd = 0; sd < s.nsub; sd++) {
      bool ghost = false;
      for (int i = sub[sd]; i < sub[sd + 1] && !ghost; i++)
        for (int q = rowptr[i]; q < rowptr[i + 1]; q++)
          if (colidx[q] >= N) { ghost = true; break; }
      (ghost ? lb : li).push_back(sd);
    }
    if (c->mesh.n_halo == 0 && W == 7) {
      // one rank: for the split-kernel measurement (wai_bench_kernel 9, 10) take the bricks on the
      // faces of the box -- rows with fewer than six neighbours -- as if every face were a partition
      // boundary (what an interior rank of a larger decomposition sees)
      li.clear(); lb.clear();
      for (int sd = 0; sd < s.nsub; sd++) {
        bool face = false;
        for (int i = sub[sd]; i < sub[sd + 1] && !face; i++) face = rowptr[i + 1] - rowptr[i] < 7;
        (face ? lb : li).push_back(sd);
      }
    }
    s.n_int = (int)li.size();
    s.n_bnd = (int)lb.size();
    lpt_order(li); lpt_order(lb);
    if (s.n_int > 0 && s.n_bnd > 0) {
      if (dev_upload(c, &s.sub_int, li) || dev_upload(c, &s.sub_bnd, lb)) return -1;
    }
  }
  if (s.big) {
    // level sets over all subdomains: rows of one level are independent wherever they live
    s.nlev_f = nlf_all; s.nlev_b = nlb_all;
    std::vector<int> of(N), ob(N);
    s.lev_f_ptr.assign(nlf_all + 1, 0); s.lev_b_ptr.assign(nlb_all + 1, 0);
    for (int i = 0; i < N; i++) { s.lev_f_ptr[levf[i] + 1]++; s.lev_b_ptr[levb[i] + 1]++; }
    for (int l = 0; l < nlf_all; l++) s.lev_f_ptr[l + 1] += s.lev_f_ptr[l];
    for (int l = 0; l < nlb_all; l++) s.lev_b_ptr[l + 1] += s.lev_b_ptr[l];
    std::vector<int> pf(s.lev_f_ptr.begin(), s.lev_f_ptr.end() - 1), pb(s.lev_b_ptr.begin(), s.lev_b_ptr.end() - 1);
    for (int i = 0; i < N; i++) { of[pf[levf[i]]++] = i; ob[pb[levb[i]]++] = i; }
    if (dev_upload(c, &s.ord_f, of) || dev_upload(c, &s.ord_b, ob)) return -1;
  }
  if (dev_upload(c, &s.sub_ptr, sub) || dev_upload(c, &s.sub_nlev, nlev) || dev_upload(c, &s.row_info, info) ||
      dev_upload(c, &s.row_uoff, uoff) || dev_upload(c, &s.row_uoffw, uoffw) || dev_upload(c, &s.row_tslot, tslot) ||
      dev_alloc(c, &s.fval, ell_size(np, N, W)) || dev_alloc(c, &s.dinv, (size_t)np * np * ell_rows(np, N)))
    return -1;
  // Kernel-selection switches are build-time (A/B builds: WAI_EXTRA_HIPCC_FLAGS="-DWAI_ILU_GENERAL" ...); the
  // run-time environment only steers what the tests compare in one process (WAI_BCGS_MERGED, WAI_JAC_PARK,
  // WAI_HALO_OVERLAP) and the transport library (WAI_RCCL_LIB).
  s.diag_only = !offdiag_fill && !s.big;
  s.level_sorted = !s.big;
  for (int sd = 0; sd < s.nsub && s.level_sorted; sd++)
    for (int i = sub[sd] + 1; i < sub[sd + 1]; i++)
      if (levf[i] < levf[i - 1] || levb[i] > levb[i - 1]) { s.level_sorted = false; break; }
  s.fast3 = fast3;
  s.scaled = true;
#ifdef WAI_ILU_GENERAL
  s.diag_only = false;     // stored L / U factor everywhere
#endif
#ifdef WAI_ILU_NOFAST
  s.fast3 = false;         // no compacted 3 + 3 couplings
#endif
#ifdef WAI_ILU_NOSCALE
  s.scaled = false;        // DILU with the inverted pivots read per application
#endif
  {
    // 160 KB of LDS per CU; a workgroup may use 64 KB
    const size_t need = ((size_t)(((s.max_rows + 63) / 64) * 64) * np + 32 + (size_t)s.max_ublocks * 4) * sizeof(double);
    s.park = need <= 64 * 1024;
#ifdef WAI_PC_NOPARK
    s.park = false;        // k_pc instead of k_pc_park
#endif
  }
  {
    // one thread per scalar row: needs the pivot-scaled DILU form, <= 4 + 4 couplings and a brick whose
    // scalar rows fit one workgroup.  Default for block sizes 3 and 4, where a whole block row per
    // thread does not fit the register file (-DWAI_PC_ROWS=0 / 1 forces it off / on, bs <= 2 too).
    const bool can = s.diag_only && s.scaled && !s.big && s.max_nlu <= 4 && s.max_rows * np <= 1024 && W <= 8;
#ifdef WAI_PC_ROWS
    s.rows_kernel = can && (WAI_PC_ROWS != 0);
#else
    s.rows_kernel = can && np >= 3;
#endif
  }
  {
    // one wave per brick: <= 64 block rows, <= 3 lower and <= 4 upper in-brick couplings, LDS for four bricks per
    // workgroup within 64 KB (-DWAI_PC_WAVE=0 builds without)
    const size_t lds_w = (size_t)4 * (64 * np + (size_t)s.max_ublocks_w * np * np) * sizeof(double);
    s.wave_kernel = s.rows_kernel && np == 3 && s.max_rows <= 64   // (4 x 4 blocks: 174 VGPRs, two waves per SIMD -- not measured, k_pc_rows keeps them)
                    && s.max_nl <= 3 && max_nu <= 4 && lds_w <= 64 * 1024;
#ifdef WAI_PC_WAVE
    s.wave_kernel = s.wave_kernel && (WAI_PC_WAVE != 0);
#endif
  }
  if (s.rows_kernel) {
    // bricks whose long rows come first (MINC: fracture cells, then their matrix cells with 2 of 8
    // slots): k_pc_rows maps the long rows of all components to the first waves, so that a wave is
    // all-long or all-short and the short ones skip the slot loop instead of idling in it
    std::vector<int> split(s.nsub);
    bool any = false;
    for (int sd = 0; sd < s.nsub; sd++) {
      const int lo = sub[sd], hi = sub[sd + 1];
      int r1 = lo;
      while (r1 < hi && (rowptr[r1 + 1] - rowptr[r1]) * 2 > W) r1++;
      bool sorted = true;
      for (int i = r1; i < hi && sorted; i++) sorted = (rowptr[i + 1] - rowptr[i]) * 2 <= W;
      split[sd] = (sorted && r1 > lo) ? r1 - lo : hi - lo;
      any = any || split[sd] != hi - lo;
      // bits 16+: the most blocks a short row of the brick has, or 15 where short rows are mixed among the long ones.
      // k_pc_wave then knows a row's slot count from the brick's record -- long rows take all W slots (a missing
      // neighbour's padding: a zero block on the own column) -- instead of waiting for rowptr before its first block load
      int short_cnt = 0;
      for (int i = r1; i < hi; i++) short_cnt = std::max(short_cnt, rowptr[i + 1] - rowptr[i]);
      bool mixed = false;
      for (int i = lo; i < hi && !mixed; i++) mixed = (i < r1) != ((rowptr[i + 1] - rowptr[i]) * 2 > W);
      split[sd] |= (mixed || !sorted ? 15 : short_cnt) << 16;
    }
    if (any && dev_upload(c, &s.sub_split, split)) return -1;
  }
  if (np == 2 && W <= 8 && !s.big && s.park && s.diag_only && s.scaled && s.fast3 && s.max_rows <= 512) {
    // k_pc_park will serve: its column indices as 16-bit (segment, offset) pairs -- 14 of a row's 304 bytes less per launch
    std::vector<unsigned short> c16((size_t)8 * N, 0);      // [row][8]: a row's indices are ONE 16-byte load
    std::vector<int> seg((size_t)s.nsub * 8, 0);
    std::vector<int> far;
    bool ok = true;
    // (tests: WAI_COL16_MAX_SEG=<n> lowers the limit so that a structured mesh takes the bail-out an unstructured one would)
    int max_seg = 8;
    if (const char* e = getenv("WAI_COL16_MAX_SEG")) max_seg = std::max(1, std::min(8, atoi(e)));
    for (int sd = 0; sd < s.nsub && ok; sd++) {
      const int lo = sub[sd], hi = sub[sd + 1];
      far.clear();
      for (int i = lo; i < hi; i++)
        for (int q = rowptr[i]; q < rowptr[i + 1]; q++)
          if (colidx[q] < lo || colidx[q] >= hi) far.push_back(colidx[q]);
      std::sort(far.begin(), far.end());
      far.erase(std::unique(far.begin(), far.end()), far.end());
      int* sg = seg.data() + (size_t)sd * 8;
      int nseg = 1;
      sg[0] = lo;
      for (size_t k = 0; k < far.size();) {      // windows of 8192 columns over what the brick reaches outside itself
        if (nseg == max_seg) { ok = false; break; }
        const int base = far[k];
        sg[nseg++] = base;
        while (k < far.size() && far[k] - base < 8192) k++;
      }
      for (int i = lo; i < hi && ok; i++) {
        const int cnt = rowptr[i + 1] - rowptr[i];
        for (int q = 0; q < W; q++) {
          const int cg = q < cnt ? colidx[rowptr[i] + q] : i;      // padding: the own column (a zero block), as Bcsr::col has it
          int sgi = 0;
          if (cg < lo || cg >= hi) {
            sgi = nseg - 1;
            while (sgi > 0 && !(cg >= sg[sgi] && cg - sg[sgi] < 8192)) sgi--;
            if (sgi == 0) { ok = false; break; }
          }
          c16[(size_t)i * 8 + q] = (unsigned short)((sgi << 13) | (cg - sg[sgi]));
        }
      }
    }
#ifdef WAI_NO_COL16
    ok = false;            // A/B builds: the int32 planes everywhere
#endif
    if (ok) {
      if (hipMalloc(reinterpret_cast<void**>(&s.col16), c16.size() * sizeof(unsigned short)) != hipSuccess ||
          hipMemcpy(s.col16, c16.data(), c16.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) {
        c->err = "hipMalloc of the 16-bit column indices failed";
        return -1;
      }
      if (dev_upload(c, &s.sub_seg, seg)) return -1;
    }
  }
  s.built = true;
  s.factored = false;
  return 0;
}

void free_schedule(IluSchedule& s) {
  hipFree(s.col16); hipFree(s.sub_seg);
  hipFree(s.sub_ptr); hipFree(s.sub_nlev); hipFree(s.sub_split); hipFree(s.row_info); hipFree(s.fval); hipFree(s.dinv);
  hipFree(s.row_uoff); hipFree(s.row_uoffw); hipFree(s.row_tslot); hipFree(s.sub_order); hipFree(s.sub_int); hipFree(s.sub_bnd); hipFree(s.ord_f); hipFree(s.ord_b);
  s = IluSchedule();
}
void free_asm(AsmSystem& a) {
  hipFree(a.net_pos); hipFree(a.net_pair);
  free_schedule(a.sched);
  hipFree(a.E.col); hipFree(a.E.val); hipFree(a.ext_row); hipFree(a.gmap); hipFree(a.r_ext); hipFree(a.hval); hipFree(a.r_full);
  a = AsmSystem();
}

// ILU(k) symbolic phase on the blocks of a block matrix (host CSR, ascending columns, all columns inside the
// row's block): level-of-fill rule of PETSc's MatILUFactorSymbolic -- an entry created while row k is
// eliminated from row i gets lev(i,k) + lev(k,j) + 1, an entry reached twice keeps the smaller level, kept when
// <= levels ("sub_preconditioner": {"factor": {"levels": k}}, src/timestepper.F90:1716-1718, 1827).  ILU(k)'s
// numeric phase is ILU(0) on the filled pattern with explicit zeros, which is how it runs here.
// src: per entry the index it is filled from (kept for original entries, -1 for fill).
void iluk_fill(const std::vector<int>& ptr, int levels, std::vector<int>& rp, std::vector<int>& col, std::vector<int>& src) {
  const int n = (int)rp.size() - 1;
  std::vector<int> orp(n + 1, 0), ocol, osrc, olev, odiag(n, 0);
  ocol.reserve(col.size() * (size_t)(1 + 2 * levels)); osrc.reserve(ocol.capacity()); olev.reserve(ocol.capacity());
  std::vector<int> wc, wl, ws;
  for (size_t b = 0; b + 1 < ptr.size(); b++)
    for (int i = ptr[b]; i < ptr[b + 1]; i++) {
      wc.assign(col.begin() + rp[i], col.begin() + rp[i + 1]);
      ws.assign(src.begin() + rp[i], src.begin() + rp[i + 1]);
      wl.assign(wc.size(), 0);
      for (size_t a = 0; a < wc.size() && wc[a] < i; a++) {   // eliminate with row k = wc[a], ascending (fill included)
        const int k = wc[a], lik = wl[a];
        for (int r = odiag[k] + 1; r < orp[k + 1]; r++) {
          const int j = ocol[r], lv = lik + olev[r] + 1;
          if (lv > levels) continue;
          const size_t pos = (size_t)(std::lower_bound(wc.begin() + a + 1, wc.end(), j) - wc.begin());
          if (pos < wc.size() && wc[pos] == j) { wl[pos] = std::min(wl[pos], lv); continue; }
          wc.insert(wc.begin() + pos, j); wl.insert(wl.begin() + pos, lv); ws.insert(ws.begin() + pos, -1);
        }
      }
      orp[i] = (int)ocol.size();
      odiag[i] = -1;
      for (size_t a = 0; a < wc.size(); a++) {
        if (wc[a] == i) odiag[i] = (int)ocol.size();
        ocol.push_back(wc[a]); osrc.push_back(ws[a]); olev.push_back(wl[a]);
      }
      orp[i + 1] = (int)ocol.size();
      if (odiag[i] < 0) odiag[i] = orp[i + 1] - 1;
    }
  rp.swap(orp); col.swap(ocol); src.swap(osrc);
}

// PCASM: the overlapped row set of every subdomain (MatIncreaseOverlap over the matrix graph, owned
// rows only), the extended block-ELL matrix that holds each set as its own block, and the map that
// fills it from the Jacobian.  Local order inside a block = ascending row index (PETSc sorts the
// subdomain index sets).
// levels > 0: ILU(k) fill inside every block; overlap 0 with levels > 0 is block Jacobi + ILU(k) on the same path
int ensure_halo_dof(wai_ctx* c, int dof) {   // halo buffers wide enough for `dof` doubles per cell
  if (dof <= c->max_dof_buf) return 0;
  if (c->d_sendbuf) (void)hipFree(c->d_sendbuf);
  if (c->d_recvbuf) (void)hipFree(c->d_recvbuf);
  c->d_sendbuf = c->d_recvbuf = nullptr;
  c->max_dof_buf = dof;
  if (dev_alloc(c, &c->d_sendbuf, (size_t)c->send_total * dof) || dev_alloc(c, &c->d_recvbuf, (size_t)c->mesh.n_halo * dof)) return -1;
  return 0;
}

// The structure of the partition-ghost cells' matrix rows, from their owners (collective).  Every cell gets the
// identity (owner rank, owner's local index); the identities of the ghost cells arrive by a halo exchange, and a
// second exchange carries, for every cell a rank sends, the identities of its row's columns.  The receiver keeps
// the columns it knows (its owned and ghost cells -- what the overlapped row sets can contain) in ascending local
// order, with the sender's slot each came from.
int ghost_rows(wai_ctx* c, std::vector<int>& grp, std::vector<int>& gci, std::vector<int>& gslot) {
  const Bcsr& J = c->J;
  const int N = J.n, H = c->mesh.n_halo, W = J.W;
  std::vector<double> ids((size_t)N + H, -1.0);
  const double base = (double)c->comm->rank * 4294967296.0;
  for (int i = 0; i < N; i++) ids[i] = base + i;
  double* scratch = c->ks.tmp;   // a Krylov work vector (n_prim * bs + 16 doubles): idle while the preconditioner is set up
  HIPCHK(c, hipMemcpyAsync(scratch, ids.data(), sizeof(double) * (N + H), hipMemcpyHostToDevice, c->stream));
  if (halo_exchange(c, scratch, 1)) return -1;
  HIPCHK(c, hipMemcpyAsync(ids.data(), scratch, sizeof(double) * (N + H), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (ensure_halo_dof(c, W * J.bs * J.bs)) return -1;
  std::vector<int> sidx((size_t)c->send_total);
  HIPCHK(c, hipMemcpy(sidx.data(), c->d_send_idx, sizeof(int) * sidx.size(), hipMemcpyDeviceToHost));
  std::vector<double> sb((size_t)c->send_total * W, -1.0), rb((size_t)H * W, -1.0);
  for (int p = 0; p < c->send_total; p++) {
    const int i = sidx[p];
    for (int q = J.h_rowptr[i]; q < J.h_rowptr[i + 1]; q++) sb[(size_t)p * W + (q - J.h_rowptr[i])] = ids[J.h_colidx[q]];
  }
  HIPCHK(c, hipMemcpyAsync(c->d_sendbuf, sb.data(), sizeof(double) * sb.size(), hipMemcpyHostToDevice, c->stream));
  if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), W, c->d_sendbuf, c->d_recvbuf,
                    c->stream, c->err))
    return -1;
  HIPCHK(c, hipMemcpyAsync(rb.data(), c->d_recvbuf, sizeof(double) * rb.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<std::pair<double, int>> known((size_t)N + H);
  for (int i = 0; i < N + H; i++) known[i] = {ids[i], i};
  std::sort(known.begin(), known.end());
  grp.assign((size_t)H + 1, 0);
  gci.clear(); gslot.clear();
  std::vector<std::pair<int, int>> row;
  for (int h = 0; h < H; h++) {
    row.clear();
    for (int q = 0; q < W; q++) {
      const double id = rb[(size_t)h * W + q];
      if (id < 0.0) continue;
      auto it = std::lower_bound(known.begin(), known.end(), std::make_pair(id, -1));
      if (it != known.end() && it->first == id) row.push_back({it->second, q});
    }
    std::sort(row.begin(), row.end());
    for (auto& e : row) { gci.push_back(e.first); gslot.push_back(e.second); }
    grp[h + 1] = (int)gci.size();
  }
  return 0;
}

int build_asm(wai_ctx* c, int overlap, int levels, bool with_net) {
  AsmSystem& a = c->as;
  free_asm(a);
  const Bcsr& J = c->J;
  const int N = J.n, np = J.bs;
  // Overlap across rank boundaries (the reference's PCASM subdomains are the ranks and MatIncreaseOverlap
  // pulls in the neighbours' rows): the overlapped sets may contain partition-ghost cells, whose matrix rows come
  // from their owners.  One ghost layer exists, so overlap 1 -- the reference's default -- is exact; a deeper overlap would
  // silently stop at that layer on a rank boundary and is refused instead.
  const bool cross = overlap > 0 && c->comm && c->comm->nranks > 1 && c->mesh.n_halo > 0 && c->n_nbr > 0;
  if (cross && overlap > 1) {
    c->err = "preconditioner asm: overlap > 1 across ranks is not supported (the partition carries one ghost layer); use overlap 1";
    return -2;
  }
  const int H = cross ? c->mesh.n_halo : 0, NX = N + H;
  std::vector<int> grp, gci, gslot;
  if (cross && ghost_rows(c, grp, gci, gslot)) return -1;
  // row i of the local matrix: owned rows are the Jacobian's, ghost rows the received ones
  auto row_begin = [&](int i) { return i < N ? J.h_rowptr[i] : grp[i - N]; };
  auto row_end = [&](int i) { return i < N ? J.h_rowptr[i + 1] : grp[i - N + 1]; };
  auto row_col = [&](int i, int e) { return i < N ? J.h_colidx[e] : gci[e]; };
  auto row_src = [&](int i, int e) { return i < N ? (e - J.h_rowptr[i]) * N + i : -(2 + gslot[e] * H + (i - N)); };
  std::vector<int> sub((size_t)c->ilu.nsub + 1);
  HIPCHK(c, hipMemcpy(sub.data(), c->ilu.sub_ptr, sizeof(int) * sub.size(), hipMemcpyDeviceToHost));
  const int nsub = c->ilu.nsub;
  std::vector<int> ext_ptr(nsub + 1, 0), ext_rows, mark(NX, -1), loc(NX, 0);
  ext_rows.reserve((size_t)N * 2);
  for (int sd = 0; sd < nsub; sd++) {
    const size_t start = ext_rows.size();
    for (int i = sub[sd]; i < sub[sd + 1]; i++) { ext_rows.push_back(i); mark[i] = sd; }
    size_t lo = start;
    for (int l = 0; l < overlap; l++) {
      const size_t hi = ext_rows.size();
      for (size_t q = lo; q < hi; q++) {
        const int i = ext_rows[q];
        for (int e = row_begin(i); e < row_end(i); e++) {
          const int j = row_col(i, e);
          if (j >= NX || mark[j] == sd) continue;
          ext_rows.push_back(j); mark[j] = sd;
        }
      }
      lo = hi;
    }
    std::sort(ext_rows.begin() + start, ext_rows.end());
    ext_ptr[sd + 1] = (int)ext_rows.size();
  }
  const int n_ext = (int)ext_rows.size();
  std::fill(mark.begin(), mark.end(), -1);
  std::vector<int> erp(n_ext + 1, 0), ecol, esrc;
  ecol.reserve((size_t)n_ext * 7); esrc.reserve((size_t)n_ext * 7);
  int W = 1;
  for (int sd = 0; sd < nsub; sd++) {
    const int a0 = ext_ptr[sd], b0 = ext_ptr[sd + 1];
    for (int q = a0; q < b0; q++) { mark[ext_rows[q]] = sd; loc[ext_rows[q]] = q; }
    for (int q = a0; q < b0; q++) {
      const int i = ext_rows[q];
      for (int e = row_begin(i); e < row_end(i); e++) {
        const int j = row_col(i, e);
        if (j >= NX || mark[j] != sd) continue;
        ecol.push_back(loc[j]);
        esrc.push_back(row_src(i, e));   // slot * n + row in J's block-ELL planes, or the ghost rows' (<= -2)
      }
      erp[q + 1] = (int)ecol.size();
    }
  }
  // (columns are positions in the extended numbering: block b's rows are ext_ptr[b] .. ext_ptr[b + 1])
  // The source network's blocks (flow_simulation_modify_jacobian, src/flow_simulation.F90:3023-3084: the reference widens
  // the BAIJ pattern by the network's dependencies and PETSc factors what it finds there): every pair of network cells
  // that share a subdomain's row set gets an entry (a structural zero of A where the cells are not neighbours; the
  // values are added after the gather, k_asm_add_couplings), before the fill levels are counted.
  const Network& nw = c->net;
  const int mnet = with_net ? (int)nw.cp_cells.size() : 0;
  if (mnet > 0) {
    std::vector<int> netidx(NX, -1);
    for (int r = 0; r < mnet; r++) netidx[nw.cp_cells[r]] = r;
    std::vector<int> nrp(n_ext + 1, 0), ncol, nsrc;
    ncol.reserve(ecol.size() + (size_t)mnet * mnet); nsrc.reserve(ncol.capacity());
    for (int sd = 0; sd < nsub; sd++) {
      std::vector<int> cells;   // ext positions of the network cells in this subdomain's row set
      for (int q = ext_ptr[sd]; q < ext_ptr[sd + 1]; q++) if (ext_rows[q] < N && netidx[ext_rows[q]] >= 0) cells.push_back(q);
      for (int q = ext_ptr[sd]; q < ext_ptr[sd + 1]; q++) {
        const bool isnet = ext_rows[q] < N && netidx[ext_rows[q]] >= 0 && cells.size() > 1;
        if (!isnet) {
          for (int e = erp[q]; e < erp[q + 1]; e++) { ncol.push_back(ecol[e]); nsrc.push_back(esrc[e]); }
        } else {   // merge the row's columns with the network cells' positions (both ascending)
          size_t a2 = 0;
          int e = erp[q];
          while (e < erp[q + 1] || a2 < cells.size()) {
            const int ca = e < erp[q + 1] ? ecol[e] : 0x7fffffff, cb = a2 < cells.size() ? cells[a2] : 0x7fffffff;
            if (ca <= cb) { ncol.push_back(ca); nsrc.push_back(esrc[e]); e++; if (cb == ca) a2++; }
            else { ncol.push_back(cb); nsrc.push_back(-1); a2++; }
          }
        }
        nrp[q + 1] = (int)ncol.size();
      }
    }
    erp.swap(nrp); ecol.swap(ncol); esrc.swap(nsrc);
  }
  if (levels > 0) iluk_fill(ext_ptr, levels, erp, ecol, esrc);
  for (int q = 0; q < n_ext; q++) W = std::max(W, erp[q + 1] - erp[q]);
  if (W > 255) { c->err = "ILU(k): more than 255 blocks in a factor row"; return -2; }
  std::vector<int> ell_col((size_t)W * n_ext), gmap((size_t)W * n_ext, -1), erow(n_ext);
  for (int sd = 0; sd < nsub; sd++)
    for (int q = ext_ptr[sd]; q < ext_ptr[sd + 1]; q++) {
      const int i = ext_rows[q];
      const bool own = i >= sub[sd] && i < sub[sd + 1];
      erow[q] = own ? (int)((unsigned)i | 0x80000000u) : i;
      const int cnt = erp[q + 1] - erp[q];
      for (int t = 0; t < W; t++) {
        ell_col[(size_t)t * n_ext + q] = t < cnt ? ecol[erp[q] + t] : q;
        gmap[(size_t)t * n_ext + q] = t < cnt ? esrc[erp[q] + t] : -1;
      }
    }
  // (re)build
  IluSchedule fresh;
  a.sched = fresh;
  a.n_ext = n_ext;
  a.E.n = n_ext; a.E.ncols = n_ext; a.E.bs = np; a.E.W = W; a.E.nnzb = (int)ecol.size();
  a.E.h_rowptr = erp; a.E.h_colidx = ecol;
  if (dev_upload(c, &a.E.col, ell_col) || dev_upload(c, &a.gmap, gmap) || dev_upload(c, &a.ext_row, erow) ||
      dev_alloc(c, &a.E.val, ell_size(np, n_ext, W)) || dev_alloc(c, &a.r_ext, (size_t)np * n_ext + 16))
    return -1;
  if (int e = build_schedule(c, a.sched, erp, ecol, ext_ptr, n_ext, W, np, false)) return e;
  a.with_net = with_net;
  a.n_net = 0;
  if (mnet > 0) {   // where the blocks of the network's E land in the extended planes
    std::vector<int> netidx(NX, -1), pos, pair;
    for (int r = 0; r < mnet; r++) netidx[nw.cp_cells[r]] = r;
    for (int q = 0; q < n_ext; q++) {
      const int i = ext_rows[q];
      if (i >= N || netidx[i] < 0) continue;
      for (int t = 0; t < erp[q + 1] - erp[q]; t++) {
        const int j = ext_rows[ecol[erp[q] + t]];
        if (j < N && netidx[j] >= 0) { pos.push_back(t * n_ext + q); pair.push_back(netidx[i] * mnet + netidx[j]); }
      }
    }
    a.n_net = (int)pos.size();
    if (a.n_net && (dev_upload(c, &a.net_pos, pos) || dev_upload(c, &a.net_pair, pair))) return -1;
  }
  if (cross) {
    if (dev_alloc(c, &a.hval, ell_size(np, H, J.W)) || dev_alloc(c, &a.r_full, (size_t)np * NX + 16)) return -1;
    HIPCHK(c, hipMemset(a.r_full, 0, sizeof(double) * ((size_t)np * NX + 16)));
  }
  a.cross = cross;
  a.overlap = overlap;
  a.levels = levels;
  return 0;
}

// PCLU: dense inverse of every preconditioner block (one block per rank with sub_ptr = NULL), by
// Gauss-Jordan elimination with partial pivoting on the host.  Meant for small systems.
int lu_setup(wai_ctx* c) {
  const Bcsr& J = c->J;
  const int bs = J.bs, bb = bs * bs, nsub = c->ilu.nsub;
  std::vector<int> sub((size_t)nsub + 1);
  HIPCHK(c, hipMemcpy(sub.data(), c->ilu.sub_ptr, sizeof(int) * sub.size(), hipMemcpyDeviceToHost));
  LuBlocks& L = c->lu;
  if (L.h_inv_ptr.empty()) {
    L.h_inv_ptr.assign((size_t)nsub + 1, 0);
    for (int s = 0; s < nsub; s++) {
      const size_t m = (size_t)(sub[s + 1] - sub[s]) * bs;
      if (m > 8192) { c->err = "preconditioner lu: a block has more than 8192 unknowns (dense inverses; use ilu)"; return -2; }
      L.h_inv_ptr[s + 1] = L.h_inv_ptr[s] + m * m;
    }
    L.total = L.h_inv_ptr[nsub];
    if (L.total > ((size_t)1 << 29)) { c->err = "preconditioner lu: more than 4 GB of dense block inverses"; return -2; }
    if (dev_alloc(c, &L.inv, L.total) || dev_upload(c, &L.inv_ptr, L.h_inv_ptr)) return -1;
  }
  std::vector<double> val((size_t)J.nnzb * bb), inv(L.total), A;
  {
    double* tmp = nullptr;
    if (dev_alloc(c, &tmp, val.size())) return -1;
    launch_ell_to_bcsr(c, J.val, tmp);
    HIPCHK(c, hipMemcpyAsync(val.data(), tmp, val.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(tmp);
  }
  for (int s = 0; s < nsub; s++) {
    const int lo = sub[s], hi = sub[s + 1], m = (hi - lo) * bs;
    A.assign((size_t)m * m, 0.0);
    double* B = inv.data() + L.h_inv_ptr[s];
    std::fill(B, B + (size_t)m * m, 0.0);
    for (int i = 0; i < m; i++) B[(size_t)i * m + i] = 1.0;
    for (int i = lo; i < hi; i++)
      for (int q = J.h_rowptr[i]; q < J.h_rowptr[i + 1]; q++) {
        const int j = J.h_colidx[q];
        if (j < lo || j >= hi) continue;   // couplings leaving the block are dropped (block Jacobi)
        for (int r = 0; r < bs; r++)
          for (int k = 0; k < bs; k++) A[(size_t)((i - lo) * bs + r) * m + (j - lo) * bs + k] = val[(size_t)q * bb + r * bs + k];
      }
    for (int p = 0; p < m; p++) {   // Gauss-Jordan with partial pivoting on [A | B]
      int piv = p;
      for (int r = p + 1; r < m; r++) if (std::fabs(A[(size_t)r * m + p]) > std::fabs(A[(size_t)piv * m + p])) piv = r;
      if (A[(size_t)piv * m + p] == 0.0) return 1;   // singular block: recoverable (KSP_DIVERGED_PC_FAILED)
      if (piv != p)
        for (int k = 0; k < m; k++) { std::swap(A[(size_t)p * m + k], A[(size_t)piv * m + k]); std::swap(B[(size_t)p * m + k], B[(size_t)piv * m + k]); }
      const double d = 1.0 / A[(size_t)p * m + p];
      for (int k = 0; k < m; k++) { A[(size_t)p * m + k] *= d; B[(size_t)p * m + k] *= d; }
      for (int r = 0; r < m; r++) {
        const double f = A[(size_t)r * m + p];
        if (r == p || f == 0.0) continue;
        for (int k = 0; k < m; k++) { A[(size_t)r * m + k] -= f * A[(size_t)p * m + k]; B[(size_t)r * m + k] -= f * B[(size_t)p * m + k]; }
      }
    }
  }
  HIPCHK(c, hipMemcpyAsync(L.inv, inv.data(), L.total * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int do_pc_setup(wai_ctx* c) {
  read_env(c);
  if (c->opts.pc_type == WAI_PC_NONE) { c->ilu.factored = true; return 0; }
  if (c->opts.pc_type == WAI_PC_LU) {
    Prof p(c, KC_PC_SETUP);
    const int e = lu_setup(c);
    if (e == 0) c->ilu.factored = true;
    return e;
  }
  {
    Prof p(c, KC_PC_SETUP);
    if (pc_extended(c)) {
      const int ov = c->opts.pc_type == WAI_PC_ASM ? (c->opts.asm_overlap > 0 ? c->opts.asm_overlap : 1) : 0;
      const int lv = std::max(c->opts.ilu_levels, 0);
      const bool wn = pc_with_net(c);
      if (c->as.overlap != ov || c->as.levels != lv || c->as.E.bs != c->J.bs || c->as.with_net != wn) { if (int e = build_asm(c, ov, lv, wn)) return e < 0 ? -1 : e; }
      if (c->as.cross) {   // the ghost cells' matrix rows, from their owners
        const int dof = c->J.W * c->J.bs * c->J.bs;
        if (ensure_halo_dof(c, dof)) return -1;
        launch_pack_rows(c);
        if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), dof, c->d_sendbuf,
                          c->d_recvbuf, c->stream, c->err))
          return -1;
        launch_unpack_rows(c);
      }
      launch_asm_gather_matrix(c);
      if (launch_ilu_factor_on(c, c->as.E, c->as.sched)) return -1;
      c->ilu.factored = true;
    } else if (launch_ilu_factor(c)) return -1;
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  return fl[0] ? 1 : 0;
}

}  // namespace wai

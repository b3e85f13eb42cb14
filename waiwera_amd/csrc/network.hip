// The source network (groups, reinjectors: src/source_network*.F90) of libwaiwera_hip.so: its host pass between the EOS
// sweep and the residual kernel, the Jacobian blocks it couples cells with (src/flow_simulation.F90:3023-3084), the
// operator (A + E) x, and the ABI entry points that describe and inspect it.
#include "host.hpp"

using namespace wai;

namespace wai {

// ---- source network: groups and reinjectors, one pass on the host -------------------------------
// source_network%update (src/source_network.F90:90-130) after the sources' own controls: group sums
// (source_network_group.F90:239-287) and limiters with uniform (:479-534) or progressive (:652-763;
// array_progressive_limit, utils.F90:607-647) scaling, reinjector capacities
// (source_network_reinjector.F90:1014-1112) and distribution with overflow (:1115-1292, :970-1010).
// Serial: every source of the network lives on this rank.
void net_separate(const SrcCtl& k, double rate, double enth, NetNode& n) {   // separator.F90:139-166, 212-260
  double q = rate, h = enth, steam_m = 0.0, steam_e = 0.0;
  for (int st = 0; st < 4; st++) {
    const double hf = st == 0 ? k.sep_hf : k.sep_more[2 * (st - 1)], hg = st == 0 ? k.sep_hg : k.sep_more[2 * (st - 1) + 1];
    if (st > 0 && !(hg > 0.0)) break;
    double f, hw, hs;
    if (h <= hf) { f = 0.0; hw = h; hs = 0.0; }
    else if (h <= hg) { f = (h - hf) / (hg - hf); hw = hf; hs = hg; }
    else { f = 1.0; hw = 0.0; hs = h; }
    const double sr = f * q;
    steam_m += sr; steam_e += sr * hs;
    q = (1.0 - f) * q; h = hw;
  }
  n.wrate = q; n.wenth = h; n.srate = steam_m;
  n.senth = std::fabs(steam_m) > 1.e-9 ? steam_e / steam_m : 0.0;
}
void net_zero_separated(NetNode& n) { n.wrate = n.wenth = n.srate = n.senth = 0.0; }
void net_source_set_rate(Network& nw, int i, double rate) {   // source_network_node_set_rate + get_separated_flows
  NetNode& n = nw.src[i];
  n.rate = rate;
  if (rate < 0.0 && i < (int)nw.h_ctl.size() && nw.h_ctl[i].sep_hg > 0.0) net_separate(nw.h_ctl[i], rate, n.enth, n);
  else net_zero_separated(n);
}
NetNode& net_node(Network& nw, const NetRef& r) { return r.kind == 1 ? nw.src[r.index] : nw.groups[r.index].node; }
double net_rate_by_type(const NetNode& n, int type) { return type == 1 ? n.wrate : (type == 2 ? n.srate : n.rate); }
void net_group_sum(Network& nw, NetGroup& g) {   // source_network_group_sum + default_separated_flows
  double q = 0.0, qh = 0.0;
  for (const NetRef& r : g.in) { const NetNode& n = net_node(nw, r); q += n.rate; qh += n.rate * n.enth; }
  g.node.enth = std::fabs(q) > 1.e-9 ? qh / q : 0.0;
  g.node.rate = q;
  if (q < 0.0 && g.sep.sep_hg > 0.0) net_separate(g.sep, q, g.node.enth, g.node);   // the group's own separator (:375-403)
  else if (q < 0.0) {
    double wq = 0, wqh = 0, sq = 0, sqh = 0;
    for (const NetRef& r : g.in) {
      const NetNode& n = net_node(nw, r);
      wq += n.wrate; wqh += n.wrate * n.wenth; sq += n.srate; sqh += n.srate * n.senth;
    }
    g.node.wrate = wq; g.node.srate = sq;
    g.node.wenth = std::fabs(wq) > 1.e-9 ? wqh / wq : 0.0;
    g.node.senth = std::fabs(sq) > 1.e-9 ? sqh / sq : 0.0;
  } else net_zero_separated(g.node);
}
void net_scale(Network& nw, const NetRef& r, double scale) {   // scale_rate, recursive through groups
  if (r.kind == 1) { net_source_set_rate(nw, r.index, nw.src[r.index].rate * scale); return; }
  NetGroup& g = nw.groups[r.index];
  for (const NetRef& q : g.in) net_scale(nw, q, scale);
  net_group_sum(nw, g);
}
bool net_min_limit_scale(const NetNode& n, int nl, const int* type, const double* limit, double& scale) {
  bool over = false;
  scale = 1.0;
  for (int i = 0; i < nl; i++) {
    const double a = std::fabs(net_rate_by_type(n, type[i]));
    if (a > limit[i]) { over = true; if (a > 1.e-6) scale = std::min(scale, limit[i] / a); }
  }
  return over;
}
void net_limit_inputs(Network& nw, const NetRef& r, int nl, const int* type, const double* limit);
void net_limit_rate(Network& nw, const NetRef& r, int nl, const int* type, const double* limit) {
  double scale;
  if (r.kind == 1 || nw.groups[r.index].scaling == 0) {   // node / uniform group: one factor for everything below
    if (net_min_limit_scale(net_node(nw, r), nl, type, limit, scale)) net_scale(nw, r, scale);
    return;
  }
  bool over = false;
  for (int i = 0; i < nl; i++) over = over || std::fabs(net_rate_by_type(nw.groups[r.index].node, type[i])) > limit[i];
  if (over) net_limit_inputs(nw, r, nl, type, limit);
}
void net_limit_inputs(Network& nw, const NetRef& r, int nl, const int* type, const double* limit) {
  if (r.kind == 1 || nw.groups[r.index].scaling == 0) { net_limit_rate(nw, r, nl, type, limit); return; }
  NetGroup& g = nw.groups[r.index];   // progressive: inputs are limited in order until the total is met
  const size_t m = g.in.size();
  std::vector<double> node_limit(m * 3, 0.0);
  for (int il = 0; il < nl; il++) {
    double sum = 0.0;
    for (size_t i = 0; i < m; i++) {
      const double a = std::fabs(net_rate_by_type(net_node(nw, g.in[i]), type[il]));
      if (sum + a > limit[il]) { node_limit[i * 3 + il] = limit[il] - sum; break; }
      node_limit[i * 3 + il] = a;
      sum += a;
    }
  }
  for (size_t i = 0; i < m; i++) net_limit_inputs(nw, g.in[i], nl, type, &node_limit[i * 3]);
  net_group_sum(nw, g);
}
void net_node_limit_rate(double node_rate, double& rate) {   // reinjector.F90:199-215
  if (node_rate > -1.0) rate = rate > -1.0 ? std::min(rate, node_rate) : node_rate;
}
void net_total(double wr, double wh, double sr, double sh, double& rate, double& enth) {
  rate = wr + sr;
  enth = rate > 1.e-6 ? (wr * wh + sr * sh) / rate : 0.0;
}

// one pass of the network on the host: nw.h_raw (rates, then enthalpies of the sources' own controls) ->
// node states, nw.is_out / out_rate / out_enth for the sources the reinjectors feed
void network_evaluate(Network& nw) {
  const int n = (int)nw.src.size();
  for (int i = 0; i < n; i++) { nw.src[i].enth = nw.h_raw[n + i]; net_source_set_rate(nw, i, nw.h_raw[i]); }
  for (NetGroup& g : nw.groups) net_group_sum(nw, g);
  for (size_t gi = 0; gi < nw.groups.size(); gi++) {
    NetGroup& g = nw.groups[gi];
    if (!g.n_limit) continue;
    NetRef self; self.kind = 2; self.index = (int)gi;
    net_limit_rate(nw, self, g.n_limit, g.limit_type, g.limit);
    for (size_t gj = gi + 1; gj < nw.groups.size(); gj++) net_group_sum(nw, nw.groups[gj]);   // sum_out
  }
  // what the injection sources can take: their own specified rate, -1 if none
  auto specified = [&](int i) { return nw.rate_specified[i] ? nw.h_raw[i] : -1.0; };
  std::vector<double>& out_rate = nw.out_rate;
  std::vector<double>& out_enth = nw.out_enth;
  std::vector<char>& is_out = nw.is_out;
  out_rate.assign(n, 0.0); out_enth.assign(n, 0.0); is_out.assign(n, 0);
  for (NetReinjector& r : nw.reinjectors) r.fed = false;
  for (int ri : nw.reinj_order) {   // capacities, downstream first
    NetReinjector& r = nw.reinjectors[ri];
    double cap[3] = {0.0, 0.0, 0.0};
    for (const NetOutput& o : r.out) {
      double node_rate = -1.0;
      if (o.out.kind == 1) node_rate = specified(o.out.index);
      else if (o.out.kind == 3) node_rate = o.flow == 1 ? nw.reinjectors[o.out.index].node.wrate : nw.reinjectors[o.out.index].node.srate;
      else continue;
      double& cc = cap[o.flow];
      if (node_rate > -1.0) { if (cc > -1.0) cc += node_rate; } else cc = -1.0;
    }
    r.node.wrate = cap[1]; r.node.srate = cap[2];
  }
  for (auto it = nw.reinj_order.rbegin(); it != nw.reinj_order.rend(); ++it) {   // distribution, upstream first
    NetReinjector& r = nw.reinjectors[*it];
    if (r.in.kind == 1 || r.in.kind == 2) {
      const NetNode& in = net_node(nw, r.in);
      r.in_w = std::fabs(in.wrate); r.in_wh = in.wenth; r.in_s = std::fabs(in.srate); r.in_sh = in.senth;
    } else if (!r.fed) { r.in_w = r.in_wh = r.in_s = r.in_sh = 0.0; }
    double wbal = r.in_w, sbal = r.in_s;
    r.out_w = r.out_s = 0.0;
    for (NetOutput& o : r.out) {
      double qw = 0.0, qs = 0.0;
      double& q = o.flow == 1 ? qw : qs;
      if (o.rate > -1.0) q = o.rate;                                              // rate output (:463-480)
      else if (o.proportion >= 0.0) q = o.proportion * (o.flow == 1 ? r.in_w : r.in_s);   // proportion output (:484-501)
      else q = -1.0;                                                              // whatever is left
      double node_rate = -1.0;
      if (o.out.kind == 1) node_rate = specified(o.out.index);
      else if (o.out.kind == 3) node_rate = o.flow == 1 ? nw.reinjectors[o.out.index].node.wrate : nw.reinjectors[o.out.index].node.srate;
      if (o.out.kind) net_node_limit_rate(node_rate, q);
      double& bal = o.flow == 1 ? wbal : sbal;
      double& tot = o.flow == 1 ? r.out_w : r.out_s;
      if (q < 0.0) q = bal;
      q = std::min(q, bal);
      bal = std::max(bal - q, 0.0);
      tot += q;
      // enthalpies: specified for this flow type, or the input's
      const double wh = (o.enthalpy > 0.0 && o.flow == 1) ? o.enthalpy : (o.enthalpy > 0.0 ? 0.0 : r.in_wh);
      const double sh = (o.enthalpy > 0.0 && o.flow == 2) ? o.enthalpy : (o.enthalpy > 0.0 ? 0.0 : r.in_sh);
      o.node.wrate = qw; o.node.wenth = wh; o.node.srate = qs; o.node.senth = sh;
      net_total(qw, wh, qs, sh, o.node.rate, o.node.enth);
      if (o.out.kind == 1) {
        const int i = o.out.index;
        is_out[i] = 1; out_rate[i] = o.node.rate; out_enth[i] = o.node.enth;
        nw.src[i].wrate = qw; nw.src[i].wenth = wh; nw.src[i].srate = qs; nw.src[i].senth = sh;
      } else if (o.out.kind == 3) {
        NetReinjector& d = nw.reinjectors[o.out.index];
        if (!d.fed) { d.in_w = d.in_wh = d.in_s = d.in_sh = 0.0; d.fed = true; }
        if (o.flow == 1) { d.in_w += qw; d.in_wh = wh; } else { d.in_s += qs; d.in_sh = sh; }
      }
    }
    r.over.wrate = wbal; r.over.wenth = r.in_wh; r.over.srate = sbal; r.over.senth = r.in_sh;
    net_total(wbal, r.in_wh, sbal, r.in_sh, r.over.rate, r.over.enth);
    if (r.overflow.kind == 3) {
      NetReinjector& d = nw.reinjectors[r.overflow.index];
      d.in_w = wbal; d.in_wh = r.in_wh; d.in_s = sbal; d.in_sh = r.in_sh; d.fed = true;
    } else if (r.overflow.kind == 1) {   // an overflow source takes what is left, whatever its own rate says (:1002-1006)
      const int i = r.overflow.index;
      is_out[i] = 1; out_rate[i] = r.over.rate; out_enth[i] = r.over.enth;
      nw.src[i].wrate = wbal; nw.src[i].wenth = r.in_wh; nw.src[i].srate = sbal; nw.src[i].senth = r.in_sh;
    }
  }
  for (int i = 0; i < n; i++)
    if (is_out[i]) {   // reinjector_output_update (:283-320): rate always, enthalpy unless the source has its own
      nw.src[i].rate = out_rate[i];
      nw.src[i].enth = (nw.enth_specified[i] && i < (int)nw.h_enth0.size()) ? nw.h_enth0[i] : out_enth[i];
    }
}

int network_update(wai_ctx* c) {
  Network& nw = c->net;
  const int n = c->src.n;           // local sources
  if (!nw.on) return 0;
  const bool span = !nw.gidx.empty();
  const int ng = span ? nw.n_global : n;
  if (ng == 0) return 0;
  // the sources' own (controlled) rates and flowing enthalpies on the current fluid
  if (n) {
    launch_source_rates(c, nw.d_raw, true);
    HIPCHK(c, hipMemcpyAsync(span ? nw.h_loc.data() : nw.h_raw.data(), nw.d_raw, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (span) {   // all ranks' sources: every rank fills its own entries, the sum is the gather (collective)
    std::fill(nw.h_raw.begin(), nw.h_raw.end(), 0.0);
    for (int i = 0; i < n; i++) { nw.h_raw[nw.gidx[i]] = nw.h_loc[i]; nw.h_raw[ng + nw.gidx[i]] = nw.h_loc[n + i]; }
    HIPCHK(c, hipMemcpyAsync(nw.d_all, nw.h_raw.data(), sizeof(double) * 2 * ng, hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_all, 2 * (size_t)ng, 0, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(nw.h_raw.data(), nw.d_all, sizeof(double) * 2 * ng, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  network_evaluate(nw);
  if (!n) return 0;
  const std::vector<double>& out_rate = nw.out_rate;
  const std::vector<char>& is_out = nw.is_out;
  // hand the result to the device: scale factors of group members, rates / enthalpies of reinjection sources
  bool enth_changed = false;
  for (int i = 0; i < n; i++) {
    const int g = span ? nw.gidx[i] : i;
    double mode = 0.0, val = 0.0;
    if (is_out[g]) {
      mode = 2.0; val = out_rate[g];
      const double e = nw.src[g].enth;
      if (e != nw.l_enth[i]) { nw.l_enth[i] = e; enth_changed = true; }
    } else if (nw.src[g].rate != nw.h_raw[g]) {
      mode = 1.0; val = nw.h_raw[g] != 0.0 ? nw.src[g].rate / nw.h_raw[g] : 1.0;
    }
    nw.l_net[2 * i] = mode; nw.l_net[2 * i + 1] = val;
  }
  HIPCHK(c, hipMemcpyAsync(c->src.net, nw.l_net.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
  if (enth_changed)
    HIPCHK(c, hipMemcpyAsync(c->src.enth, nw.l_enth.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // l_net / l_enth are reused by the next pass
  return 0;
}

// ---- Jacobian couplings through the source network ---------------------------------------------
// flow_simulation_modify_jacobian (src/flow_simulation.F90:3023-3084) widens the Jacobian's pattern by the
// network's dependencies and MatFDColoring then differences the whole residual function -- network pass
// included -- into it.  Here the 7-point part A is differenced with the network's factors held
// (k_jacobian), and the rest, E = dR/dy *through the network pass*, is differenced separately on the cells
// of the network's sources: for every such cell j and primary k, with y_jk + h (the same h as A's
// columns), E[:, j][:, k] = (R(network pass redone) - R(factors held)) / h on the rows of those cells.
// Two residual launches on the network's rows alone (k_residual's row list: the same code path per row, so
// the same bits as a full sweep) and three host passes per column: the cost does not grow with the mesh.
__global__ void k_gather_rows(int m, int bs, const int* __restrict__ cells, const double* __restrict__ f,
                              double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * bs) return;
  out[t] = f[(size_t)cells[t / bs] * bs + t % bs];
}

// t += E x on the network's rows: thread (i, r) sums its row over the mc column cells (cells are distinct: no race).
// xg != null: x at the column cells, gathered over the ranks ([mc][bs]); else the columns are the row cells themselves
__global__ void k_coupling_apply(int mr, int mc, int bs, const int* __restrict__ cells, const double* __restrict__ val,
                                 const double* __restrict__ x, const double* __restrict__ xg, double* __restrict__ t) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= mr * bs) return;
  const int i = id / bs, r = id % bs;
  double s = 0.0;
  for (int j = 0; j < mc; j++) {
    const double* e = val + ((size_t)(i * mc + j) * bs + r) * bs;
    const double* xj = xg ? xg + (size_t)j * bs : x + (size_t)cells[j] * bs;
    for (int k = 0; k < bs; k++) s += e[k] * xj[k];
  }
  t[(size_t)cells[i] * bs + r] += s;
}

int network_couplings(wai_ctx* c, double dt, double* y, const double* lhs_old) {
  Network& nw = c->net;
  nw.cp_valid = false;
  const bool span = nw.cp_span;
  const int ml = (int)nw.cp_cells.size(), m = span ? nw.cp_m : ml, j0 = span ? nw.cp_j0 : 0;
  if (!nw.on || !nw.coupling || m == 0) return 0;
  const int bs = c->np, mb = ml * bs, me = span ? c->comm->rank : 0;
  if (!nw.d_cp_val) {
    std::vector<int> cells = nw.cp_cells;
    if (cells.empty()) cells.push_back(0);
    if (dev_upload(c, &nw.d_cp_cells, cells) || dev_alloc(c, &nw.d_cp_val, (size_t)std::max(ml, 1) * m * bs * bs) ||
        dev_alloc(c, &nw.d_cp_f, (size_t)c->mesh.n_local * bs) || dev_alloc(c, &nw.d_cp_g, (size_t)2 * std::max(mb, 1)) ||
        dev_alloc(c, &nw.d_cp_x, (size_t)m * bs + 2))
      return -1;
  }
  nw.h_cp_val.assign((size_t)ml * m * bs * bs, 0.0);
  std::vector<double> g((size_t)2 * mb), yc((size_t)bs);
  const int grid = (mb + 63) / 64;
  auto set_y = [&](int cell, int k, double v) -> int {
    HIPCHK(c, hipMemcpyAsync(y + (size_t)cell * bs + k, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    launch_eos(c, y, cell, 1, false);
    return 0;
  };
  // one double from its owner to every rank (the sum over the ranks of {value on the owner, 0 elsewhere})
  auto from_owner = [&](double& v, bool mine) -> int {
    if (!span) return 0;
    const double mineval = mine ? v : 0.0;
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_x, &mineval, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_cp_x, 1, 0, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(&v, nw.d_cp_x, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  };
  if (network_update(c)) return -1;   // the factors A was differenced with
  bool any = false;
  for (int j = 0; j < m; j++) {       // every rank walks the same columns: the network passes are collective
    const bool mine = !span || nw.cp_owner[j] == me;
    const int cell = mine ? nw.cp_cells[j - j0] : -1;
    if (mine) {
      HIPCHK(c, hipMemcpyAsync(yc.data(), y + (size_t)cell * bs, sizeof(double) * bs, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int k = 0; k < bs; k++) {
      double h = 0.0;
      if (mine) {
        double dx = yc[k];   // MatFDColoring "ds" increment, as fd_step (kernels_assembly.hip)
        if (std::fabs(dx) < c->opts.fd_umin) dx = dx >= 0.0 ? c->opts.fd_umin : -c->opts.fd_umin;
        h = dx * c->opts.fd_eps;
      }
      if (from_owner(h, mine)) return -1;
      if (mine && set_y(cell, k, yc[k] + h)) return -1;
      if (ml) {
        launch_residual(c, dt, lhs_old, nw.d_cp_f, nullptr, nullptr, nw.d_cp_cells, ml);   // factors held; the network's rows alone
        hipLaunchKernelGGL(k_gather_rows, grid, 64, 0, c->stream, ml, bs, nw.d_cp_cells, nw.d_cp_f, nw.d_cp_g);
      }
      if (network_update(c)) return -1;
      if (ml) {
        launch_residual(c, dt, lhs_old, nw.d_cp_f, nullptr, nullptr, nw.d_cp_cells, ml);   // network pass redone
        hipLaunchKernelGGL(k_gather_rows, grid, 64, 0, c->stream, ml, bs, nw.d_cp_cells, nw.d_cp_f, nw.d_cp_g + mb);
        HIPCHK(c, hipMemcpyAsync(g.data(), nw.d_cp_g, sizeof(double) * 2 * mb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < ml; i++)
          for (int r = 0; r < bs; r++) {
            const double e = (g[(size_t)mb + i * bs + r] - g[(size_t)i * bs + r]) / h;
            nw.h_cp_val[((size_t)(i * m + j) * bs + r) * bs + k] = e;
            any = any || e != 0.0;
          }
      }
      if (mine && set_y(cell, k, yc[k])) return -1;   // back to the unperturbed state and its network factors
      if (network_update(c)) return -1;
    }
  }
  // a perturbed state outside the EOS's range was already reported by the perturbed-state sweep of A
  HIPCHK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
  if (any) {
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_val, nw.h_cp_val.data(), sizeof(double) * nw.h_cp_val.size(), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  double flag = any ? 1.0 : 0.0;   // every rank applies E (a collective gather of x) or none does
  if (span) {
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_x, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_cp_x, 1, 1, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(&flag, nw.d_cp_x, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  nw.cp_valid = flag != 0.0;
  return 0;
}

// t = (A + E) x: the block-ELL SpMV and, when the network couples cells, its blocks on top.  A network on several
// ranks: x at the network's cells is gathered first (one all-reduce of m * bs doubles per application)
int apply_operator(wai_ctx* c, const double* x, double* t) {
  launch_spmv(c, x, t);
  const Network& nw = c->net;
  if (!nw.cp_valid) return 0;
  const int ml = (int)nw.cp_cells.size(), bs = c->np;
  if (!nw.cp_span) {
    hipLaunchKernelGGL(k_coupling_apply, (ml * bs + 63) / 64, 64, 0, c->stream, ml, ml, bs, nw.d_cp_cells, nw.d_cp_val, x,
                       (const double*)nullptr, t);
    return 0;
  }
  const int m = nw.cp_m;
  HIPCHK(c, hipMemsetAsync(nw.d_cp_x, 0, sizeof(double) * (size_t)m * bs, c->stream));
  if (ml) hipLaunchKernelGGL(k_gather_rows, (ml * bs + 63) / 64, 64, 0, c->stream, ml, bs, nw.d_cp_cells, x, nw.d_cp_x + (size_t)nw.cp_j0 * bs);
  if (comm_allreduce(c->comm, nw.d_cp_x, (size_t)m * bs, 0, c->stream, c->err)) return -1;
  if (ml) hipLaunchKernelGGL(k_coupling_apply, (ml * bs + 63) / 64, 64, 0, c->stream, ml, m, bs, nw.d_cp_cells, nw.d_cp_val, x,
                             (const double*)nw.d_cp_x, t);
  return 0;
}

// the flat description of wai_set_source_network -> Network (no device involved)
int network_build(Network& nw, int n, const int* rate_specified, const int* enthalpy_specified, int n_groups,
                  const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                  const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                  const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                  const int* out_kind, const int* out_node, const double* out_rate, const double* out_proportion,
                  const double* out_enthalpy, const int* rj_overflow_kind, const int* rj_overflow, std::string& err) {
  if (!n || !rate_specified || !enthalpy_specified) { err = "source network without sources"; return -2; }
  auto ok = [&](int kind, int idx) {
    return kind == 0 || (kind == 1 && idx >= 0 && idx < n) || (kind == 2 && idx >= 0 && idx < n_groups) ||
           (kind == 3 && idx >= 0 && idx < n_reinj);
  };
  nw.rate_specified.assign(rate_specified, rate_specified + n);
  nw.enth_specified.assign(enthalpy_specified, enthalpy_specified + n);
  nw.groups.assign(std::max(n_groups, 0), NetGroup());
  for (int g = 0; g < n_groups; g++) {
    NetGroup& G = nw.groups[g];
    for (int q = grp_ptr[g]; q < grp_ptr[g + 1]; q++) {
      if (!ok(grp_in_kind[q], grp_in[q]) || grp_in_kind[q] == 0 || grp_in_kind[q] == 3 || (grp_in_kind[q] == 2 && grp_in[q] >= g)) {
        err = "source network group: inputs are sources or earlier groups"; return -2;
      }
      NetRef r; r.kind = grp_in_kind[q]; r.index = grp_in[q];
      G.in.push_back(r);
    }
    G.scaling = grp_scaling ? grp_scaling[g] : 0;
    for (int l = 0; l < 3; l++)
      if (grp_limit_type && grp_limit_type[3 * g + l] >= 0) {
        G.limit_type[G.n_limit] = grp_limit_type[3 * g + l]; G.limit[G.n_limit] = grp_limit[3 * g + l]; G.n_limit++;
      }
    std::memset(&G.sep, 0, sizeof(G.sep));
    if (grp_sep) {
      G.sep.sep_hf = grp_sep[8 * g]; G.sep.sep_hg = grp_sep[8 * g + 1];
      for (int q = 0; q < 6; q++) G.sep.sep_more[q] = grp_sep[8 * g + 2 + q];
    }
  }
  nw.reinjectors.assign(std::max(n_reinj, 0), NetReinjector());
  for (int r = 0; r < n_reinj; r++) {
    NetReinjector& R = nw.reinjectors[r];
    if (!ok(rj_in_kind[r], rj_in[r]) || rj_in_kind[r] == 3 || !ok(rj_overflow_kind[r], rj_overflow[r]) || rj_overflow_kind[r] == 2) {
      err = "source network reinjector: bad input / overflow reference"; return -2;
    }
    R.in.kind = rj_in_kind[r]; R.in.index = rj_in[r];
    R.overflow.kind = rj_overflow_kind[r]; R.overflow.index = rj_overflow[r];
    for (int q = rj_out_ptr[r]; q < rj_out_ptr[r + 1]; q++) {
      if (!ok(out_kind[q], out_node[q]) || out_kind[q] == 2 || (out_flow[q] != 1 && out_flow[q] != 2)) {
        err = "source network reinjector: bad output"; return -2;
      }
      NetOutput o;
      o.flow = out_flow[q]; o.out.kind = out_kind[q]; o.out.index = out_node[q];
      o.rate = out_rate[q]; o.proportion = out_proportion[q]; o.enthalpy = out_enthalpy[q];
      R.out.push_back(o);
    }
  }
  {   // order: a reinjector after every reinjector it delivers or overflows to
    nw.reinj_order.clear();
    std::vector<int> state(std::max(n_reinj, 0), 0);
    std::function<bool(int)> visit = [&](int r) -> bool {
      if (state[r] == 2) return true;
      if (state[r] == 1) return false;
      state[r] = 1;
      const NetReinjector& R = nw.reinjectors[r];
      for (const NetOutput& o : R.out) if (o.out.kind == 3 && !visit(o.out.index)) return false;
      if (R.overflow.kind == 3 && !visit(R.overflow.index)) return false;
      state[r] = 2;
      nw.reinj_order.push_back(r);
      return true;
    };
    for (int r = 0; r < n_reinj; r++) if (!visit(r)) { err = "source network reinjectors form a cycle"; return -2; }
  }
  nw.src.assign(n, NetNode());
  nw.h_raw.assign(2 * (size_t)n, 0.0);
  if (nw.h_enth0.size() != (size_t)n) nw.h_enth0.assign(n, 0.0);
  return 0;
}
// the cells whose equations and unknowns the network ties together: every source a group, a reinjector input,
// output or overflow names (source_network_identify_source_dependencies, source_network.F90:359-498, walks the
// same lists: production cells of a reinjector's input x cells of the sources it -- or the reinjectors
// it delivers or overflows to -- feeds; the members of a limited group among each other).  The coupling
// blocks E cover all pairs of these cells, a superset of the reference's dependency list.
void network_cells(Network& nw, int n) {
  std::vector<char> in_net((size_t)n, 0);
  auto mark = [&](const NetRef& r) { if (r.kind == 1 && r.index >= 0 && r.index < n) in_net[r.index] = 1; };
  for (const NetGroup& g : nw.groups) for (const NetRef& r : g.in) mark(r);
  for (const NetReinjector& r : nw.reinjectors) { mark(r.in); mark(r.overflow); for (const NetOutput& o : r.out) mark(o.out); }
  nw.cp_cells.clear();
  for (int i = 0; i < n && i < (int)nw.h_cell.size(); i++) if (in_net[i]) nw.cp_cells.push_back(nw.h_cell[i]);
  std::sort(nw.cp_cells.begin(), nw.cp_cells.end());
  nw.cp_cells.erase(std::unique(nw.cp_cells.begin(), nw.cp_cells.end()), nw.cp_cells.end());
}
// the same over several ranks: the network's cells of all ranks, ordered by (owner rank, local cell); this rank's own
// are then one contiguous run of the columns and, in that order, the rows it differences and applies
void network_cells_span(Network& nw, int ng, const std::vector<double>& id, int rank) {
  std::vector<char> in_net((size_t)ng, 0);
  auto mark = [&](const NetRef& r) { if (r.kind == 1 && r.index >= 0 && r.index < ng) in_net[r.index] = 1; };
  for (const NetGroup& g : nw.groups) for (const NetRef& r : g.in) mark(r);
  for (const NetReinjector& r : nw.reinjectors) { mark(r.in); mark(r.overflow); for (const NetOutput& o : r.out) mark(o.out); }
  std::vector<double> u;
  for (int g = 0; g < ng; g++) if (in_net[g]) u.push_back(id[g]);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  nw.cp_span = true;
  nw.cp_m = (int)u.size();
  nw.cp_owner.resize(u.size());
  nw.cp_cells.clear();
  nw.cp_j0 = 0;
  for (size_t j = 0; j < u.size(); j++) {
    const int owner = (int)(u[j] / 4294967296.0);
    nw.cp_owner[j] = owner;
    if (owner == rank) {
      if (nw.cp_cells.empty()) nw.cp_j0 = (int)j;
      nw.cp_cells.push_back((int)(u[j] - (double)owner * 4294967296.0));
    }
  }
}

}  // namespace wai

extern "C" {

// Source network (src/source_network_group.F90, source_network_reinjector.F90; input "network.group",
// "network.reinject").  Node references are (kind, index) pairs: kind 0 none, 1 source, 2 group,
// 3 reinjector.  Groups in dependency order (a group after the groups it takes in).
int wai_set_source_network(wai_ctx* c, const int* rate_specified, const int* enthalpy_specified, int n_groups,
                           const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                           const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                           const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                           const int* out_kind, const int* out_node, const double* out_rate,
                           const double* out_proportion, const double* out_enthalpy, const int* rj_overflow_kind,
                           const int* rj_overflow) {
  if (!c) return -2;
  Network& nw = c->net;
  const int n = c->src.n;
  auto ctl = nw.h_ctl; auto e0 = nw.h_enth0; auto cells = nw.h_cell;
  const bool coupling = nw.coupling;
  nw.free_device();
  if (c->src.net) { (void)hipFree(c->src.net); c->src.net = nullptr; }
  nw = Network();
  nw.h_ctl = ctl; nw.h_enth0 = e0; nw.h_cell = cells; nw.coupling = coupling;
  if (n_groups <= 0 && n_reinj <= 0) return 0;
  const bool span = c->comm && c->comm->nranks > 1;
  int ng = n;
  std::vector<double> span_id;   // several ranks: every source's cell as (owner rank, local cell)
  if (span) {
    // the description is numbered by global source index (wai_set_source_global_index); what the pass needs of
    // the other ranks' sources -- separator enthalpies, specified injection enthalpies -- is gathered once, here
    if ((int)c->src_gidx.size() != n || c->src_nglobal < n) { c->err = "source network on several ranks: wai_set_source_global_index first"; return -2; }
    ng = c->src_nglobal;
    for (int g : c->src_gidx) if (g < 0 || g >= ng) { c->err = "global source index out of range"; return -2; }
    nw.gidx = c->src_gidx;
    nw.n_global = ng;
    const int NG = 10;   // per source: 8 separator enthalpies, the specified enthalpy, the cell's identity (rank * 2^32 + cell)
    std::vector<double> all((size_t)NG * ng, 0.0);
    for (int i = 0; i < n; i++) {
      const int g = nw.gidx[i];
      if (i < (int)ctl.size()) {
        all[(size_t)NG * g] = ctl[i].sep_hf; all[(size_t)NG * g + 1] = ctl[i].sep_hg;
        for (int q = 0; q < 6; q++) all[(size_t)NG * g + 2 + q] = ctl[i].sep_more[q];
      }
      all[(size_t)NG * g + 8] = i < (int)e0.size() ? e0[i] : 0.0;
      all[(size_t)NG * g + 9] = (double)c->comm->rank * 4294967296.0 + (double)(i < (int)cells.size() ? cells[i] : 0);
    }
    double* tmp = nullptr;
    if (dev_upload(c, &tmp, all)) return -1;
    int rc = comm_allreduce(c->comm, tmp, all.size(), 0, c->stream, c->err);
    if (!rc && hipMemcpyAsync(all.data(), tmp, sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -1;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -1;
    (void)hipFree(tmp);
    if (rc) return -1;
    nw.h_ctl.assign((size_t)ng, SrcCtl{});
    nw.h_enth0.assign((size_t)ng, 0.0);
    span_id.assign((size_t)ng, 0.0);
    for (int g = 0; g < ng; g++) {
      nw.h_ctl[g].sep_hf = all[(size_t)NG * g]; nw.h_ctl[g].sep_hg = all[(size_t)NG * g + 1];
      for (int q = 0; q < 6; q++) nw.h_ctl[g].sep_more[q] = all[(size_t)NG * g + 2 + q];
      nw.h_enth0[g] = all[(size_t)NG * g + 8];
      span_id[g] = all[(size_t)NG * g + 9];
    }
  }
  if (int e = network_build(nw, ng, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in, grp_scaling,
                            grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind,
                            out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, c->err))
    return e;
  if (dev_alloc(c, &nw.d_raw, 2 * (size_t)std::max(n, 1)) || dev_alloc(c, &c->src.net, 2 * (size_t)std::max(n, 1))) return -1;
  if (span && dev_alloc(c, &nw.d_all, 2 * (size_t)ng)) return -1;
  HIPCHK(c, hipMemset(c->src.net, 0, sizeof(double) * 2 * std::max(n, 1)));
  nw.h_loc.assign(2 * (size_t)n, 0.0);
  nw.l_net.assign(2 * (size_t)n, 0.0);
  nw.l_enth.assign((size_t)n, 0.0);
  for (int i = 0; i < n; i++) nw.l_enth[i] = span ? nw.h_enth0[nw.gidx[i]] : (i < (int)nw.h_enth0.size() ? nw.h_enth0[i] : 0.0);
  nw.on = true;
  if (!span) network_cells(nw, n);
  else network_cells_span(nw, ng, span_id, c->comm->rank);
  c->as.overlap = -1;   // the factor's pattern carries the network's cell pairs: built again at the next set-up
  return 0;
}

// Global index of every local source, for a source network whose sources live on several ranks: the network
// description handed to wai_set_source_network then refers to sources by these indices (0 .. n_global - 1), the same
// description on every rank.  After wai_set_sources, before wai_set_source_network.
int wai_set_source_global_index(wai_ctx* c, int n_global, const int* global_index) {
  if (!c || n_global < 0 || (c->src.n > 0 && !global_index)) return -2;
  c->src_gidx.assign(global_index, global_index + c->src.n);
  c->src_nglobal = n_global;
  return 0;
}

int wai_set_network_couplings(wai_ctx* c, int on) {
  if (!c) return -2;
  const bool was = pc_with_net(c);
  c->net.coupling = on != 0;
  c->net.cp_in_pc = on != 1;      // 1: in the operator only (rounds 2-3); 2 (and any other non-zero value): in the factor's pattern too
  if (!on) c->net.cp_valid = false;
  // pc_fused() / pc_extended() follow pc_with_net(): a set-up made for the other path (the extended system factored and
  // c->ilu not, or the other way round) must not be applied -- the next solve sets up again, the extended pattern included
  if (pc_with_net(c) != was) {
    c->ilu.factored = false;
    c->as.overlap = -1;
  }
  return 0;
}

int wai_get_network_couplings(wai_ctx* c, int* n_cells, int* cells, double* values) {
  if (!c || !n_cells) return -2;
  const Network& nw = c->net;
  const bool on = nw.on && nw.coupling && nw.cp_valid;
  const int ml = on ? (int)nw.cp_cells.size() : 0, m = on ? (nw.cp_span ? nw.cp_m : ml) : 0;
  *n_cells = m;
  if (cells)
    for (int j = 0; j < m; j++) {
      const bool mine = !nw.cp_span || (j >= nw.cp_j0 && j < nw.cp_j0 + ml);
      cells[j] = mine ? nw.cp_cells[j - (nw.cp_span ? nw.cp_j0 : 0)] : -1 - nw.cp_owner[j];
    }
  if (values && ml) std::memcpy(values, nw.h_cp_val.data(), sizeof(double) * nw.h_cp_val.size());
  return 0;
}
// The same network pass without a context or a device (host logic only; tests): the sources' own rates
// and enthalpies and their separators (8 doubles per source: hf, hg of stage 1, then (hf, hg) of stages
// 2..4, hg = 0: no separator / no further stage) in, node states out -- sources and groups 6 doubles each
// (rate, enthalpy, water_rate, water_enthalpy, steam_rate, steam_enthalpy), reinjectors 8 each as
// wai_get_source_network.
int wai_network_evaluate(int n_sources, const double* rate, const double* enthalpy, const double* src_sep,
                         const int* rate_specified, const int* enthalpy_specified, int n_groups,
                         const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                         const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                         const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                         const int* out_kind, const int* out_node, const double* out_rate,
                         const double* out_proportion, const double* out_enthalpy, const int* rj_overflow_kind,
                         const int* rj_overflow, double* sources_out, double* groups_out, double* reinjectors_out) {
  if (!rate || !enthalpy || n_sources <= 0) return -2;
  Network nw;
  std::string err;
  nw.h_enth0.assign(enthalpy, enthalpy + n_sources);
  if (int e = network_build(nw, n_sources, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in,
                            grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow,
                            out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, err))
    return e;
  nw.h_ctl.assign(n_sources, SrcCtl{});
  for (int i = 0; i < n_sources && src_sep; i++) {
    nw.h_ctl[i].sep_hf = src_sep[8 * i]; nw.h_ctl[i].sep_hg = src_sep[8 * i + 1];
    for (int q = 0; q < 6; q++) nw.h_ctl[i].sep_more[q] = src_sep[8 * i + 2 + q];
  }
  for (int i = 0; i < n_sources; i++) { nw.h_raw[i] = rate[i]; nw.h_raw[n_sources + i] = enthalpy[i]; }
  network_evaluate(nw);
  auto put = [](const NetNode& n, double* o) { o[0] = n.rate; o[1] = n.enth; o[2] = n.wrate; o[3] = n.wenth; o[4] = n.srate; o[5] = n.senth; };
  for (int i = 0; sources_out && i < n_sources; i++) put(nw.src[i], sources_out + 6 * i);
  for (size_t g = 0; groups_out && g < nw.groups.size(); g++) put(nw.groups[g].node, groups_out + 6 * g);
  for (size_t r = 0; reinjectors_out && r < nw.reinjectors.size(); r++) {
    const NetReinjector& R = nw.reinjectors[r];
    const double v[8] = {R.out_w, R.out_s, R.over.rate, R.over.enth, R.over.wrate, R.over.wenth, R.over.srate, R.over.senth};
    std::memcpy(reinjectors_out + 8 * r, v, sizeof(v));
  }
  return 0;
}
// The cells between which wai_jacobian forms the network's coupling blocks, for the given sources' cells and
// network description -- no context, no device (tests pin it on the reference's dependency list).
// cells: room for n_sources entries; *n_cells: how many were written (ascending, distinct)
int wai_network_cells(int n_sources, const int* source_cell, const int* rate_specified, const int* enthalpy_specified,
                      int n_groups, const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                      const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                      const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                      const int* out_kind, const int* out_node, const double* out_rate, const double* out_proportion,
                      const double* out_enthalpy, const int* rj_overflow_kind, const int* rj_overflow, int* n_cells,
                      int* cells) {
  if (!source_cell || !n_cells || !cells || n_sources <= 0) return -2;
  Network nw;
  std::string err;
  if (int e = network_build(nw, n_sources, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in,
                            grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow,
                            out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, err))
    return e;
  nw.h_cell.assign(source_cell, source_cell + n_sources);
  network_cells(nw, n_sources);
  *n_cells = (int)nw.cp_cells.size();
  for (size_t i = 0; i < nw.cp_cells.size(); i++) cells[i] = nw.cp_cells[i];
  return 0;
}
// state of the network after the last pass: groups 6 doubles each (rate, enthalpy, water_rate,
// water_enthalpy, steam_rate, steam_enthalpy); reinjectors 8 each (output water / steam rate, overflow
// rate, enthalpy, water rate, water enthalpy, steam rate, steam enthalpy)
int wai_get_source_network(wai_ctx* c, double* groups, double* reinjectors) {
  if (!c) return -2;
  const Network& nw = c->net;
  for (size_t g = 0; groups && g < nw.groups.size(); g++) {
    const NetNode& n = nw.groups[g].node;
    const double v[6] = {n.rate, n.enth, n.wrate, n.wenth, n.srate, n.senth};
    std::memcpy(groups + 6 * g, v, sizeof(v));
  }
  for (size_t r = 0; reinjectors && r < nw.reinjectors.size(); r++) {
    const NetReinjector& R = nw.reinjectors[r];
    const double v[8] = {R.out_w, R.out_s, R.over.rate, R.over.enth, R.over.wrate, R.over.wenth, R.over.srate, R.over.senth};
    std::memcpy(reinjectors + 8 * r, v, sizeof(v));
  }
  return 0;
}

}  // extern "C"

// Per-cell / per-face physics as gfx950 device functions: EOS w / we, rel-perm and capillary
// curves, accumulation terms, two-point flux, source terms, phase transitions.
//
// Reference arithmetic replaced (file:line under /root/reference/src):
//   eos.F90:186-257, eos_w.F90:126-255, eos_we.F90:149-526, relative_permeability.F90:197-492,
//   capillary_pressure.F90:159-305, cell.F90:114-142, face.F90:282-515, fluid.F90:197-453,
//   rock.F90:142-150, source.F90:386-480, root_finder.F90:127-248, interpolation.F90:388-435.
//
// Data layout in HBM: the fluid state is struct-of-arrays, flu[field * stride + cell], with the
// reference's 23 (we) / 15 (w) fields in the reference's order (fluid.F90:212-267) so that
// field f of cell c of the reference's AoS record is flu[f*stride + c].  One thread owns one
// cell; consecutive lanes read consecutive doubles of one field (512 B per wave instruction).
#pragma once
#include <hip/hip_runtime.h>
#include "if97.hip.h"
#include "ifc67.hip.h"

namespace wai {

enum { EOS_W = 0, EOS_WE = 1, EOS_WCE = 2, EOS_WSE = 3, EOS_WAE = 4, EOS_WSCE = 5, EOS_WSAE = 6 };
// water + non-condensible gas + energy (eos_wge.F90): CO2 (eos_wce.F90) or air (eos_wae.F90)
template <int KIND> constexpr bool is_wge = (KIND == EOS_WCE || KIND == EOS_WAE);
// salt family: water + salt + energy (eos_wse.F90), and with a gas as a third component and fourth
// primary (eos_wsge.F90: wsce, wsae)
template <int KIND> constexpr bool is_wsge = (KIND == EOS_WSCE || KIND == EOS_WSAE);
template <int KIND> constexpr bool is_salt = (KIND == EOS_WSE || is_wsge<KIND>);
template <int KIND> constexpr bool gas_is_air = (KIND == EOS_WAE || KIND == EOS_WSAE);
enum { RP_FULLY_MOBILE = 0, RP_LINEAR = 1, RP_PICKENS = 2, RP_COREY = 3, RP_GRANT = 4,
       RP_VAN_GENUCHTEN = 5, RP_TABLE = 6 };
enum { CP_ZERO = 0, CP_LINEAR = 1, CP_VAN_GENUCHTEN = 2, CP_TABLE = 3 };

// fluid record field indices (fluid.F90:212-267): 6 bulk fields, nc partial pressures, then per
// phase 7 scalars + nc mass fractions
enum { F_P = 0, F_T = 1, F_REGION = 2, F_OLD_REGION = 3, F_PHASES = 4, F_PERMFAC = 5, F_PP = 6 };
enum { PH_RHO = 0, PH_MU = 1, PH_SAT = 2, PH_KR = 3, PH_PC = 4, PH_H = 5, PH_U = 6, PH_X = 7 };
// rock record (rock.F90:97-112)
enum { R_K1 = 0, R_K2 = 1, R_K3 = 2, R_WET = 3, R_DRY = 4, R_PHI = 5, R_RHO = 6, R_CP = 7 };

template <int KIND> struct EosT;
template <> struct EosT<EOS_W> {
  static constexpr int np = 1, nc = 1, nph = 1, nmob = 1, df = 15, f_phase0 = 7, ph_dof = 8;
  static constexpr bool isothermal = true;
};
template <> struct EosT<EOS_WE> {
  static constexpr int np = 2, nc = 1, nph = 2, nmob = 2, df = 23, f_phase0 = 7, ph_dof = 8;
  static constexpr bool isothermal = false;
};
template <> struct EosT<EOS_WCE> {  // water + CO2 + energy (eos_wge.F90 + eos_wce.F90)
  static constexpr int np = 3, nc = 2, nph = 2, nmob = 2, df = 26, f_phase0 = 8, ph_dof = 9;
  static constexpr bool isothermal = false;
};

template <> struct EosT<EOS_WAE> {  // water + air + energy (eos_wge.F90 + eos_wae.F90)
  static constexpr int np = 3, nc = 2, nph = 2, nmob = 2, df = 26, f_phase0 = 8, ph_dof = 9;
  static constexpr bool isothermal = false;
};
template <> struct EosT<EOS_WSE> {  // water + salt + energy (eos_wse.F90): third phase = solid halite, immobile
  static constexpr int np = 3, nc = 2, nph = 3, nmob = 2, df = 35, f_phase0 = 8, ph_dof = 9;
  static constexpr bool isothermal = false;
};

template <> struct EosT<EOS_WSCE> {  // water + salt + CO2 + energy (eos_wsge.F90 + eos_wsce.F90): 4 x 4 blocks
  static constexpr int np = 4, nc = 3, nph = 3, nmob = 2, df = 39, f_phase0 = 9, ph_dof = 10;
  static constexpr bool isothermal = false;
};
template <> struct EosT<EOS_WSAE> {  // water + salt + air + energy (eos_wsae.F90)
  static constexpr int np = 4, nc = 3, nph = 3, nmob = 2, df = 39, f_phase0 = 9, ph_dof = 10;
  static constexpr bool isothermal = false;
};

// piecewise table curve (interpolation_table_type, src/interpolation.F90): n points (x, v), end values
// held outside the data (:494-510); interpolation 0 linear (:388-404), 1 step (:715-720), 2 pchip with
// the Fritsch-Carlson derivatives d computed once on the host (:761-887, polynomial of :891-925)
constexpr int MAX_CURVE_POINTS = 12;
struct CurveTable {
  int n, interp;
  double x[MAX_CURVE_POINTS], v[MAX_CURVE_POINTS], d[MAX_CURVE_POINTS];
};

// run-time EOS parameters (kernel argument, lives in SGPRs / constant cache)
struct EosParams {
  double temperature;     // eos_w
  double scale[9][4];     // primary_scale(var, region)  (eos_we.F90:104-109; eos_wse: regions 1..8); a zero partial-
                          // pressure scale selects adaptive scaling Pg/P (eos_wge.F90:639-674)
  int rp_type, cp_type;
  double rp_par[6], cp_par[6];
  int thermo;             // "thermodynamics": THERMO_IAPWS (default) | THERMO_IFC67
  int perm_type;          // eos wse permeability modifier: 0 none, 1 power, 2 Verma-Pruess
  double perm_par[3];     // exponent, phir, gamma
  CurveTable tab[3];      // "table" curves: liquid / vapour relative permeability, capillary pressure
};

// thermodynamic formulation dispatch (thermodynamics_type: src/thermodynamics.F90, IAPWS.F90,
// IFC67.F90); `thermo` is uniform over a launch.  region 1 = liquid water, 2 = steam.
enum { THERMO_IAPWS = 0, THERMO_IFC67 = 1 };
namespace th {
__device__ __forceinline__ int props(int thermo, int region, double p, double t, double& rho, double& u) {
  if (thermo == THERMO_IFC67) return region == 1 ? ifc67::region1(p, t, rho, u) : ifc67::region2(p, t, rho, u);
  return region == 1 ? if97::region1(p, t, rho, u) : if97::region2(p, t, rho, u);
}
__device__ __forceinline__ double viscosity(int thermo, int region, double t, double p, double rho) {
  return thermo == THERMO_IFC67 ? ifc67::viscosity(region, t, p, rho) : if97::viscosity(t, rho);
}
__device__ __forceinline__ int sat_pressure(int thermo, double t, double& p) {
  return thermo == THERMO_IFC67 ? ifc67::sat_pressure(t, p) : if97::sat_pressure(t, p);
}
__device__ __forceinline__ int sat_temperature(int thermo, double p, double& t) {
  return thermo == THERMO_IFC67 ? ifc67::sat_temperature(p, t) : if97::sat_temperature(p, t);
}
__device__ __forceinline__ int phase_composition(int thermo, int region, double p, double t) {
  return thermo == THERMO_IFC67 ? ifc67::phase_composition(region) : if97::phase_composition(region, p, t);
}
}  // namespace th
}  // namespace wai
#include "salt.hip.h"
namespace wai {

// two-row table lookup with end clamping (interpolation.F90:202-222,388-404,494-510)
__device__ __forceinline__ double lin2(double x, double x0, double x1, double y0, double y1) {
  if (x <= x0) return y0;
  if (x >= x1) return y1;
  const double xi = (x - x0) / (x1 - x0);
  return (1.0 - xi) * y0 + xi * y1;
}

// interpolation_table_interpolate (src/interpolation.F90:533-543): index with val(i) <= x < val(i+1),
// clamped to the end values
__device__ __forceinline__ double curve_table(const CurveTable& t, double x) {
  if (x <= t.x[0]) return t.v[0];
  if (x >= t.x[t.n - 1]) return t.v[t.n - 1];
  int i = 0;
  for (int k = 1; k < MAX_CURVE_POINTS - 1; k++)
    if (k < t.n - 1 && x >= t.x[k]) i = k;
  const double x0 = t.x[i], x1 = t.x[i + 1], v0 = t.v[i], v1 = t.v[i + 1];
  if (t.interp == 1) return v0;
  if (t.interp == 2) {
    const double h = x1 - x0, delta = (v1 - v0) / h;
    const double del1 = (t.d[i] - delta) / h, del2 = (t.d[i + 1] - delta) / h;
    const double c2 = -(2.0 * del1 + del2), c3 = (del1 + del2) / h, dx = x - x0;
    return v0 + dx * (t.d[i] + dx * (c2 + dx * c3));
  }
  const double xi = (x - x0) / (x1 - x0);
  return (1.0 - xi) * v0 + xi * v1;
}

__device__ __forceinline__ void relperm(const EosParams& e, double sl, double& kl, double& kv) {
  const double* par = e.rp_par;
  switch (e.rp_type) {
    case RP_TABLE:   // relative_permeability_table_values, src/relative_permeability.F90:547-558
      kl = curve_table(e.tab[0], sl);
      kv = curve_table(e.tab[1], 1.0 - sl);
      break;
    case RP_FULLY_MOBILE: kl = 1.0; kv = 1.0; break;
    case RP_LINEAR:
      kl = lin2(sl, par[0], par[1], 0.0, 1.0);
      kv = lin2(1.0 - sl, par[2], par[3], 0.0, 1.0);
      break;
    case RP_PICKENS: kl = pow(sl, par[0]); kv = 1.0; break;
    case RP_COREY:
    case RP_GRANT: {
      const double slr = par[0], ssr = par[1], sv = 1.0 - sl;
      if (sv < ssr) { kl = 1.0; kv = 0.0; }
      else if (sv > 1.0 - slr) { kl = 0.0; kv = 1.0; }
      else {
        const double ss = (sl - slr) / (1.0 - slr - ssr), ss2 = ss * ss;
        kl = ss2 * ss2;
        kv = (e.rp_type == RP_COREY) ? (1.0 - 2.0 * ss + ss2) * (1.0 - ss2) : 1.0 - kl;
      }
    } break;
    case RP_VAN_GENUCHTEN: {
      const double lambda = par[0], slr = par[1], sls = par[2], ssr = par[4];
      const double ss = (sl - slr) / (sls - slr);
      if (ss < 0.0) kl = 0.0;
      else if (ss < 1.0) {
        const double w = 1.0 - pow(1.0 - pow(ss, 1.0 / lambda), lambda);
        kl = sqrt(ss) * w * w;
      } else kl = 1.0;
      if (par[3] != 0.0) kv = 1.0 - kl;
      else {
        const double sh = (sl - slr) / (1.0 - slr - ssr), sh2 = sh * sh;
        kv = fmin(1.0, (1.0 - 2.0 * sh + sh2) * (1.0 - sh2));
      }
    } break;
    default: kl = 0.0; kv = 0.0;
  }
}

__device__ __forceinline__ double capillary(const EosParams& e, double sl) {
  const double* par = e.cp_par;
  switch (e.cp_type) {
    case CP_TABLE: return curve_table(e.tab[2], sl);   // src/capillary_pressure.F90:349-358
    case CP_LINEAR: return lin2(sl, par[0], par[1], -fabs(par[2]), 0.0);
    case CP_VAN_GENUCHTEN: {
      const double eps = 1.e-3;
      const double P0 = fabs(par[0]), lambda = par[1], slr = par[2], sls = par[3];
      const double Pmax = fabs(par[4]);
      double cp = 0.0;
      if (sl < 1.0) {
        const double ss = (sl - slr) / (sls - slr);
        if (ss < 0.0) cp = -Pmax;
        else if (ss < 1.0) cp = -P0 * pow(pow(ss, -1.0 / lambda) - 1.0, 1.0 - lambda);
        else cp = 0.0;
        cp = fmin(0.0, cp);
        if (par[5] != 0.0) cp = fmax(-Pmax, cp);
        if (sl > 1.0 - eps) cp = cp * (1.0 - sl) / eps;
      }
      return cp;
    }
    default: return 0.0;
  }
}

// ---- CO2 as non-condensible gas (ncg_co2_thermodynamics.F90:83-292, ncg_thermodynamics.F90) ---
namespace co2 {
constexpr double MW = 44.01, WATER_MW = 18.01528, GAS_CONSTANT = 8.3144598;
__device__ __forceinline__ void properties(double partial_pressure, double t, double& rho, double& h) {
  const double tk = t + if97::TC_K, pp = partial_pressure * 1.0e-6;
  const double tc = pow(0.01 * tk, 3.3333333333);
  const double hci = 1.667 + 0.001542 * tk - 0.7948 * log10(tk) - 41.35 / tk;
  h = 1.e6 * (hci - 0.3571 * pp * (1.0 + 0.07576 * pp) / tc);
  const double vc = 0.00018882 * tk - pp * (0.0824 + 0.01249 * pp) / tc;
  rho = pp / vc;
}
__device__ __forceinline__ double henry_poly(double x) {
  return 0.783666 + x * (1.96025 + x * (8.20574 + x * (-7.40674 + x * (2.18380 + x * -0.220999))));
}
__device__ __forceinline__ double henrys_constant(double t) { return 1.e8 * henry_poly(t / 100.0); }
__device__ __forceinline__ double energy_solution(double t) {
  const double x = t / 100.0;
  const double dpoly = 1.96025 + x * (2.0 * 8.20574 + x * (3.0 * -7.40674 + x * (4.0 * 2.18380 + x * (5.0 * -0.220999))));
  const double hd = 1.e8 * dpoly / (henrys_constant(t) * 100.0);
  const double tk = t + if97::TC_K;
  return -1.e3 * GAS_CONSTANT * tk * tk * hd / MW;
}
__device__ __forceinline__ int viscosity(double partial_pressure, double t, double& visc) {
  if (!(partial_pressure <= 300.e5)) return 1;
  const double P[5] = {0.0, 10.0, 15.0, 20.0, 30.0};
  const double Cf[5][5] = {{1.3578, 3.9189, 9.6607, 13.1566, 14.7968},
                           {4.9227e-3, -35.984e-3, -135.479e-3, -179.352e-3, -160.731e-3},
                           {-2.9661e-6, 0.25825e-3, 0.90087e-3, 1.12474e-3, 0.850257e-3},
                           {2.8529e-9, -7.1178e-7, -2.4727e-6, -2.98864e-6, -1.99076e-6},
                           {-2.1829e-12, 6.9578e-10, 2.4156e-9, 2.85911e-9, 1.73423e-9}};
  const double p = partial_pressure / 1.e6;
  double c[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    // piecewise-linear in pressure with end clamping, written branch-light
    double v = Cf[k][0];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double xi = (p - P[i]) / (P[i + 1] - P[i]);
      const double seg = (1.0 - xi) * Cf[k][i] + xi * Cf[k][i + 1];
      v = (p > P[i] && p < P[i + 1]) ? seg : v;
      v = (p >= P[i + 1]) ? Cf[k][i + 1] : v;
    }
    c[k] = v;
  }
  visc = 1.e-5 * (c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * c[4]))));
  return 0;
}
__device__ __forceinline__ double mole_to_mass(double xmole) {
  const double w = xmole * MW;
  return w / (w + (1.0 - xmole) * WATER_MW);
}
}  // namespace co2

// air NCG thermodynamics (ncg_air_thermodynamics.F90; correlation data as held at :15-41:
// Irvine & Liley 1984 enthalpy; D'Amore & Truesdell 1988 / Cramer 1982 / Cygan 1991 Henry's
// constants of N2 and O2; Hirschfelder et al. 1954 mixture viscosity)
namespace air {
constexpr double MW = 28.96, WATER_MW = 18.01528, GAS_CONSTANT = 8.3144598;
__device__ __forceinline__ double horner4(const double* a, double x) { return a[0] + x * (a[1] + x * (a[2] + x * a[3])); }
__device__ __forceinline__ void properties(double partial_pressure, double t, double& rho, double& h) {   // :90-114
  const double c[4] = {1.20740, 9.24502, 0.115984, -5.63568e-4};
  const double tk = t + if97::TC_K;
  const double shift = horner4(c, (0.01 + if97::TC_K) / 100.0);   // zero at the triple point of water, :77-80
  rho = partial_pressure * MW / (1.e3 * GAS_CONSTANT * 1.0 * tk);
  h = 1.e4 * (horner4(c, tk / 100.0) - shift);
}
__device__ __forceinline__ void henry_constituents(double t, double* hc, double* dhinv) {
  const double p0[2] = {1.01325e5, 1.e5};
  const double a[2][7] = {{0.513726, 1.58603, -5.9378e-1, -6.98282e-1, 5.10330e-1, -1.21388e-1, 1.00041e-2},
                          {0.26234, 0.610628, 7.00732e-1, -0.139299e1, 7.13850e-1, -1.54216e-1, 1.23190e-2}};
  const double x = t / 100.0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    double p = a[i][6], d = 6.0 * a[i][6];
#pragma unroll
    for (int k = 5; k >= 0; k--) p = a[i][k] + x * p;
#pragma unroll
    for (int k = 5; k >= 1; k--) d = k * a[i][k] + x * d;          // polynomial_derivative, Horner
    hc[i] = 1.e5 * p0[i] * p;
    dhinv[i] = p0[i] * (1.e5 * d) / (hc[i] * 100.0);
  }
}
__device__ __forceinline__ double henrys_constant(double t) {      // :118-137
  double hc[2], d[2];
  henry_constituents(t, hc, d);
  return 0.79 * hc[0] + 0.21 * hc[1];
}
__device__ __forceinline__ double energy_solution(double t) {      // :174-199, ncg_thermodynamics.F90:187-231
  double hc[2], d[2];
  henry_constituents(t, hc, d);
  const double tk = t + if97::TC_K;
  return -1.e3 * GAS_CONSTANT * tk * tk * (0.79 * d[0] + 0.21 * d[1]) / MW;
}
__device__ __forceinline__ double mole_to_mass(double xmole) {
  const double w = xmole * MW;
  return w / (w + (1.0 - xmole) * WATER_MW);
}
__device__ __forceinline__ double covis(double trd, double c, double ome, double rm, double f) {
  return 266.93e-7 * sqrt(rm * trd * f) / (c * c * ome * trd);
}
__device__ inline double mixture_viscosity(double water_viscosity, double t, double xg) {   // :260-338, gas phase
  const double fair = 97.0, fwat = 363.0, cair = 3.617, cwat = 2.655;
  const double fmix = sqrt(fair * fwat), cmix = 0.5 * (cair + cwat);
  const double rm1 = MW, rm2 = WATER_MW;
  const double w = xg / rm1, x1 = w / (w + (1.0 - xg) / rm2), x2 = 1.0 - x1;
  const double tk = t + if97::TC_K, trd1 = tk / fair, trd3 = tk / fmix;
  const double ome1 = (1.188 - 0.051 * trd1) / trd1;
  const double ome3 = (1.48 - 0.412 * log(trd3)) / trd3;
  const double ard = 1.095 / trd3;
  const double rm3 = 2.0 * rm1 * rm2 / (rm1 + rm2);
  const double vis1 = covis(trd1, cair, ome1, rm1, fair);
  const double vis2 = 10.0 * water_viscosity;
  const double vis3 = covis(trd3, cmix, ome3, rm3, fmix);
  const double z1 = x1 * x1 / vis1 + 2.0 * x2 * x1 / vis3 + x2 * x2 / vis2;
  const double g = x1 * x1 * rm1 / rm2, h = x2 * x2 * rm2 / rm1;
  const double ee = (2.0 * x1 * x2 * rm1 * rm2 / (rm3 * rm3)) * vis3 / (vis1 * vis2);
  const double z2 = 0.6 * ard * (g / vis1 + ee + h / vis2);
  const double z3 = 0.6 * ard * (g + ee * (vis1 + vis2) - 2.0 * x1 * x2 + h);
  return 0.1 * (1.0 + z3) / (z1 + z2);
}
}  // namespace air

// Henry's constant and energy of solution of the gas in brine (henrys_constant_salt,
// henrys_derivative_salt, ncg_energy_solution_salt: ncg_co2_thermodynamics.F90:139-232,
// ncg_air_thermodynamics.F90:141-238, ncg_thermodynamics.F90:187-261)
template <bool AIR>
__device__ inline void gas_henry_salt(double t, double xs, double& henry, double& esol) {
  const double m = 1.0e3 * xs / (58.443 * (1.0 - xs)), x = t / 100.0, tk = t + if97::TC_K;
  if constexpr (AIR) {
    const double w[2] = {0.79, 0.21};
    const double ks[2][5] = {{0.183369, -0.236905, 0.242438, -7.30134e-2, 8.58723e-3},
                             {0.16218, -1.16909e-1, 5.55185e-2, -8.75443e-3, 9.91567e-4}};
    double hc[2], d0[2], h = 0.0, deriv = 0.0;
    air::henry_constituents(t, hc, d0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const double kb = ks[i][0] + x * (ks[i][1] + x * (ks[i][2] + x * (ks[i][3] + x * ks[i][4])));
      const double dkb = ks[i][1] + x * (2.0 * ks[i][2] + x * (3.0 * ks[i][3] + x * 4.0 * ks[i][4]));
      h += w[i] * hc[i] * pow(10.0, m * kb);
      deriv += w[i] * (d0[i] + log(10.0) * m * (dkb / 100.0));
    }
    henry = h;
    esol = -1.e3 * air::GAS_CONSTANT * tk * tk * deriv / air::MW;
  } else {
    const double ks[5] = {1.19784e-1, -7.17823e-2, 4.93854e-2, -1.03826e-2, 1.08233e-3};
    const double h0 = co2::henrys_constant(t);
    const double kb = ks[0] + x * (ks[1] + x * (ks[2] + x * (ks[3] + x * ks[4])));
    const double dkb = ks[1] + x * (2.0 * ks[2] + x * (3.0 * ks[3] + x * 4.0 * ks[4]));
    const double dpoly = 1.96025 + x * (2.0 * 8.20574 + x * (3.0 * -7.40674 + x * (4.0 * 2.18380 + x * (5.0 * -0.220999))));
    const double deriv = 1.e8 * dpoly / (h0 * 100.0) + log(10.0) * m * (dkb / 100.0);
    henry = h0 * pow(10.0, m * kb);
    esol = -1.e3 * co2::GAS_CONSTANT * tk * tk * deriv / co2::MW;
  }
}

// eos%unscale / eos%scale (eos.F90:186-210; adaptive third variable eos_wge.F90:639-674)
template <int KIND>
__device__ __forceinline__ void eos_unscale(const EosParams& e, const double* y, int region, double* prim) {
  using E = EosT<KIND>;
#pragma unroll
  for (int k = 0; k < E::np; k++) prim[k] = y[k] * e.scale[region][k];
  if constexpr (is_wge<KIND>) { if (e.scale[region][2] == 0.0) prim[2] = y[2] * prim[0]; }
  if constexpr (is_wsge<KIND>) { if (e.scale[region][3] == 0.0) prim[3] = y[3] * prim[0]; }   // eos_wsge.F90:984-998
}
template <int KIND>
__device__ __forceinline__ void eos_scale(const EosParams& e, const double* prim, int region, double* y) {
  using E = EosT<KIND>;
#pragma unroll
  for (int k = 0; k < E::np; k++) y[k] = prim[k] / e.scale[region][k];
  if constexpr (is_wge<KIND>) { if (e.scale[region][2] == 0.0) y[2] = prim[2] / prim[0]; }
  if constexpr (is_wsge<KIND>) { if (e.scale[region][3] == 0.0) y[3] = prim[3] / prim[0]; }   // eos_wsge.F90:962-980
}

// eos_wse: mixture region -> water region / halite presence (eos_wse.F90:131-134)
__device__ __forceinline__ int wse_water_region(int region) { return region > 4 ? region - 4 : region; }
__device__ __forceinline__ bool wse_halite(int region) { return region > 4; }

// fluid_permeability_factor_*_modify (fluid.F90:588-664): factor on the permeability from the pore
// fraction pf = S_l + S_v left open by halite
__device__ __forceinline__ double permeability_factor(const EosParams& e, double pf) {
  if (e.perm_type == 1) return pow(pf, e.perm_par[0]);
  if (e.perm_type == 2) {
    const double n = e.perm_par[0], phir = e.perm_par[1], gamma = e.perm_par[2];
    const double omega = 1.0 + (1.0 / (gamma * (1.0 / phir - 1.0)));
    const double theta = (pf - phir) / (1.0 - phir);
    return pow(theta, n) * (1.0 - gamma + gamma / pow(omega, n)) /
           (1.0 - gamma + gamma * pow(theta / (theta + omega - 1.0), n));
  }
  return 1.0;
}

// ---- cell state in registers ---------------------------------------------------------------
template <int KIND> struct CellState {
  using E = EosT<KIND>;
  double P, T, region, phases, permfac;
  double pp[E::nc];
  double rho[E::nph], mu[E::nph], sat[E::nph], kr[E::nph], pc[E::nph], h[E::nph], u[E::nph];
  double x[E::nph][E::nc];
};

// full EOS evaluation: scaled primaries + region -> state (fluid_properties loop body,
// flow_simulation.F90:2347-2403).  Returns the reference's err flag.
template <int KIND>
__device__ __forceinline__ int eos_eval(const EosParams& e, const double* y, int region,
                                        CellState<KIND>& s) {
  using E = EosT<KIND>;
  s.region = (double)region;
  s.permfac = 1.0;
  if constexpr (KIND == EOS_W) {
    s.P = y[0] * e.scale[region][0];
    s.T = e.temperature;
    s.sat[0] = 1.0;
    const int ph = th::phase_composition(e.thermo, region, s.P, s.T);
    if (ph <= 0) return 1;
    s.phases = (double)ph;
    double rho, u;
    const int err = th::props(e.thermo, region == 1 ? 1 : 2, s.P, s.T, rho, u);
    if (err) return err;
    s.rho[0] = rho; s.u[0] = u; s.h[0] = u + s.P / rho;
    s.kr[0] = 1.0; s.pc[0] = 0.0;
    s.mu[0] = th::viscosity(e.thermo, region == 1 ? 1 : 2, s.T, s.P, rho);
    s.x[0][0] = 1.0; s.pp[0] = s.P;
    return 0;
  } else if constexpr (is_salt<KIND>) {
    // eos_wse_bulk_properties / phase_saturations / phase_properties (eos_wse.F90:645-857), with a
    // gas eos_wsge.F90:625-852: brine pressure = P - Pg, components water, salt, gas
    constexpr bool gas = is_wsge<KIND>;
    const int wr = wse_water_region(region);
    const bool halite = wse_halite(region);
    double prim[E::np];
    eos_unscale<KIND>(e, y, region, prim);
    s.P = prim[0];
    double pg = 0.0;
    if constexpr (gas) pg = prim[3];
    const double pw = s.P - pg;
    s.pp[0] = pw; s.pp[1] = 0.0;
    if constexpr (gas) s.pp[2] = pg;
    if (wr == 4) {
      double xs2 = prim[2], t;
      if (region != 4) { if (salt::halite_solubility_two_phase(e.thermo, pw, xs2)) return 1; }
      if (salt::brine_sat_temperature(e.thermo, pw, xs2, t)) return 1;
      s.T = t;
    } else s.T = prim[1];
    const int ph = th::phase_composition(e.thermo, wr, s.P, s.T);
    if (ph <= 0) return 1;
    s.phases = (double)ph;
    const double ss = (halite || region == 2) ? prim[2] : 0.0, fs = 1.0 - ss;
    if (wr == 1) { s.sat[0] = fs; s.sat[1] = 0.0; }
    else if (wr == 2) { s.sat[0] = 0.0; s.sat[1] = fs; }
    else { s.sat[0] = fs - prim[1]; s.sat[1] = prim[1]; }
    s.sat[2] = ss;
    s.permfac = permeability_factor(e, s.sat[0] + s.sat[1]);
    double xs = 0.0;
    if (halite) { if (salt::halite_solubility(s.T, xs)) return 1; }
    else if (region != 2) xs = prim[2];
    const double sle = s.sat[0] / (1.0 - ss);
    double kl, kv;
    relperm(e, sle, kl, kv);
    const double pcl = capillary(e, sle);
    double gas_rho = 0.0, gas_h = 0.0;
    if constexpr (gas) {
      if constexpr (gas_is_air<KIND>) air::properties(pg, s.T, gas_rho, gas_h);
      else co2::properties(pg, s.T, gas_rho, gas_h);
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
      if (ph & (1 << p)) {
        double rho, u, mu;
        const double bp = (p == 0 || !gas) ? s.P : pw;
        const int err = (p == 0) ? salt::brine_properties(e.thermo, bp, s.T, xs, rho, u)
                                 : th::props(e.thermo, 2, bp, s.T, rho, u);
        if (err) return err;
        const double xp = (p == 0) ? xs : 0.0;
        if (p == 0) { if (salt::brine_viscosity(e.thermo, s.T, s.P, xs, mu)) return 1; }
        else mu = th::viscosity(e.thermo, 2, s.T, s.P, rho);
        double xg = 0.0, grho = 0.0, esol = 0.0;
        if constexpr (gas) {
          if (p == 0) {
            double henry;
            gas_henry_salt<gas_is_air<KIND>>(s.T, xs, henry, esol);
            xg = gas_is_air<KIND> ? air::mole_to_mass(pg / henry) : co2::mole_to_mass(pg / henry);
          } else {
            grho = gas_rho;
            const double tot = grho + rho;
            xg = (tot < 1.e-30) ? 0.0 : grho / tot;
            if constexpr (gas_is_air<KIND>) mu = air::mixture_viscosity(mu, s.T, xg);
            else {
              double gmu;
              if (co2::viscosity(pg, s.T, gmu)) return 1;
              mu = mu * (1.0 - xg) + gmu * xg;
            }
          }
        }
        s.mu[p] = mu;
        s.rho[p] = rho + grho;
        s.x[p][0] = 1.0 - xp - xg; s.x[p][1] = xp;
        if constexpr (gas) s.x[p][2] = xg;
        s.kr[p] = (p == 0) ? kl : kv;
        s.pc[p] = (p == 0) ? pcl : 0.0;
        const double bh = u + bp / rho;
        if constexpr (gas) {
          s.h[p] = bh * (1.0 - xg) + (gas_h + esol) * xg;
          s.u[p] = s.h[p] - s.P / s.rho[p];
        } else {
          s.h[p] = bh;
          s.u[p] = u;
        }
      } else {
        s.rho[p] = 0.0; s.u[p] = 0.0; s.h[p] = 0.0; s.kr[p] = 0.0; s.pc[p] = 0.0; s.mu[p] = 0.0;
#pragma unroll
        for (int q = 0; q < E::nc; q++) s.x[p][q] = 0.0;
      }
    }
    s.kr[2] = 0.0; s.pc[2] = 0.0; s.mu[2] = 0.0;
#pragma unroll
    for (int q = 0; q < E::nc; q++) s.x[2][q] = 0.0;
    if (halite || region == 2) {
      double rho, u;
      salt::halite_properties(s.P, s.T, rho, u);
      s.rho[2] = rho; s.u[2] = u; s.h[2] = u + s.P / rho;
      s.x[2][1] = 1.0;
    } else {
      s.rho[2] = 0.0; s.u[2] = 0.0; s.h[2] = 0.0;
    }
    return 0;
  } else if constexpr (is_wge<KIND>) {
    constexpr bool air_gas = (KIND == EOS_WAE);
    // eos_wge_bulk_properties / phase_properties (eos_wge.F90:350-543) with CO2 (eos_wce.F90)
    double prim[3];
    eos_unscale<KIND>(e, y, region, prim);
    s.P = prim[0];
    const double Pg = prim[2], Pw = s.P - Pg;
    s.pp[0] = Pw; s.pp[1] = Pg;
    if (region == 4) {
      double t;
      if (th::sat_temperature(e.thermo, Pw, t)) return 1;
      s.T = t;
    } else s.T = prim[1];
    const int ph = th::phase_composition(e.thermo, region, s.P, s.T);
    if (ph <= 0) return 1;
    s.phases = (double)ph;
    if (region == 1) { s.sat[0] = 1.0; s.sat[1] = 0.0; }
    else if (region == 2) { s.sat[0] = 0.0; s.sat[1] = 1.0; }
    else { s.sat[0] = 1.0 - prim[1]; s.sat[1] = prim[1]; }
    double kl, kv;
    relperm(e, s.sat[0], kl, kv);
    double gas_rho, gas_h;
    if constexpr (air_gas) air::properties(Pg, s.T, gas_rho, gas_h);
    else co2::properties(Pg, s.T, gas_rho, gas_h);
#pragma unroll
    for (int p = 0; p < 2; p++) {
      if (ph & (1 << p)) {
        const double wpres = (p == 0) ? s.P : Pw;
        double wrho, wu;
        const int err = th::props(e.thermo, p == 0 ? 1 : 2, wpres, s.T, wrho, wu);
        if (err) return err;
        const double grho = (p == 0) ? 0.0 : gas_rho;
        double xg, esol = 0.0;
        if (p == 0) {
          if constexpr (air_gas) {
            xg = air::mole_to_mass(Pg / air::henrys_constant(s.T));
            esol = air::energy_solution(s.T);
          } else {
            xg = co2::mole_to_mass(Pg / co2::henrys_constant(s.T));
            esol = co2::energy_solution(s.T);
          }
        } else {
          const double tot = grho + wrho;
          xg = (tot < 1.e-30) ? 0.0 : grho / tot;
        }
        const double wmu = th::viscosity(e.thermo, p == 0 ? 1 : 2, s.T, wpres, wrho);
        double mu = wmu;
        if (p == 1) {
          if constexpr (air_gas) mu = air::mixture_viscosity(wmu, s.T, xg);
          else {
            double gmu;
            if (co2::viscosity(Pg, s.T, gmu)) return 1;
            mu = wmu * (1.0 - xg) + gmu * xg;
          }
        }
        s.mu[p] = mu;
        s.rho[p] = wrho + grho;
        s.x[p][0] = 1.0 - xg; s.x[p][1] = xg;
        s.kr[p] = (p == 0) ? kl : kv;
        s.pc[p] = (p == 0) ? capillary(e, s.sat[0]) : 0.0;
        const double wh = wu + wpres / wrho;
        s.h[p] = wh * (1.0 - xg) + (gas_h + esol) * xg;
        s.u[p] = s.h[p] - s.P / s.rho[p];
      } else {
        s.rho[p] = 0.0; s.u[p] = 0.0; s.h[p] = 0.0; s.kr[p] = 0.0; s.pc[p] = 0.0; s.mu[p] = 0.0;
        s.x[p][0] = 0.0; s.x[p][1] = 0.0;
      }
    }
    return 0;
  } else {
    const double p0 = y[0] * e.scale[region][0], p1 = y[1] * e.scale[region][1];
    s.P = p0;
    s.pp[0] = p0;
    if (region == 4) {
      double t;
      if (th::sat_temperature(e.thermo, s.P, t)) return 1;
      s.T = t;
    } else s.T = p1;
    const int ph = th::phase_composition(e.thermo, region, s.P, s.T);
    if (ph <= 0) return 1;
    s.phases = (double)ph;
    if (region == 1) { s.sat[0] = 1.0; s.sat[1] = 0.0; }
    else if (region == 2) { s.sat[0] = 0.0; s.sat[1] = 1.0; }
    else { s.sat[0] = 1.0 - p1; s.sat[1] = p1; }
    double kl, kv;
    relperm(e, s.sat[0], kl, kv);
    const double pcl = capillary(e, s.sat[0]);
#pragma unroll
    for (int p = 0; p < E::nph; p++) {
      if (ph & (1 << p)) {
        double rho, u;
        const int err = th::props(e.thermo, p == 0 ? 1 : 2, s.P, s.T, rho, u);
        if (err) return err;
        s.rho[p] = rho; s.u[p] = u; s.h[p] = u + s.P / rho;
        s.kr[p] = (p == 0) ? kl : kv;
        s.pc[p] = (p == 0) ? pcl : 0.0;
        s.mu[p] = th::viscosity(e.thermo, p == 0 ? 1 : 2, s.T, s.P, rho);
        s.x[p][0] = 1.0;
      } else {
        s.rho[p] = 0.0; s.u[p] = 0.0; s.h[p] = 0.0; s.kr[p] = 0.0; s.pc[p] = 0.0; s.mu[p] = 0.0;
        s.x[p][0] = 0.0;
      }
    }
    return 0;
  }
}

// store / load a state to the SoA fluid array (only the fields the EOS rewrites; region and
// old_region are owned by the transition kernel)
template <int KIND>
__device__ __forceinline__ void store_state(double* flu, size_t stride, size_t c,
                                            const CellState<KIND>& s) {
  using E = EosT<KIND>;
  flu[F_P * stride + c] = s.P;
  flu[F_T * stride + c] = s.T;
  flu[F_PHASES * stride + c] = s.phases;
  flu[F_PERMFAC * stride + c] = s.permfac;
#pragma unroll
  for (int q = 0; q < E::nc; q++) flu[(F_PP + q) * stride + c] = s.pp[q];
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    const size_t b = (size_t)(E::f_phase0 + p * E::ph_dof) * stride + c;
    flu[b + PH_RHO * stride] = s.rho[p];
    flu[b + PH_MU * stride] = s.mu[p];
    flu[b + PH_SAT * stride] = s.sat[p];
    flu[b + PH_KR * stride] = s.kr[p];
    flu[b + PH_PC * stride] = s.pc[p];
    flu[b + PH_H * stride] = s.h[p];
    flu[b + PH_U * stride] = s.u[p];
#pragma unroll
    for (int q = 0; q < E::nc; q++) flu[b + (PH_X + q) * stride] = s.x[p][q];
  }
}

template <int KIND>
__device__ __forceinline__ void load_state(const double* __restrict__ flu, size_t stride, size_t c,
                                           CellState<KIND>& s) {
  using E = EosT<KIND>;
  s.P = flu[F_P * stride + c];
  s.T = flu[F_T * stride + c];
  s.phases = flu[F_PHASES * stride + c];
  s.permfac = flu[F_PERMFAC * stride + c];
  s.region = 0.0;
#pragma unroll
  for (int q = 0; q < E::nc; q++) s.pp[q] = 0.0;  // not needed by the sweeps
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    const size_t b = (size_t)(E::f_phase0 + p * E::ph_dof) * stride + c;
    s.rho[p] = flu[b + PH_RHO * stride];
    s.mu[p] = flu[b + PH_MU * stride];
    s.sat[p] = flu[b + PH_SAT * stride];
    s.kr[p] = flu[b + PH_KR * stride];
    s.pc[p] = flu[b + PH_PC * stride];
    s.h[p] = flu[b + PH_H * stride];
    s.u[p] = flu[b + PH_U * stride];
    if constexpr (E::nc == 1) {
      s.x[p][0] = (((int)s.phases >> p) & 1) ? 1.0 : 0.0;  // mass_fraction(1) of a present phase
    } else {
#pragma unroll
      for (int q = 0; q < E::nc; q++) s.x[p][q] = flu[b + (PH_X + q) * stride];
    }
  }
}

struct RockState { double k[3], wet, dry, phi, rho, cp; };
__device__ __forceinline__ void load_rock(const double* __restrict__ rock, size_t stride, size_t c,
                                          RockState& r) {
  r.k[0] = rock[R_K1 * stride + c]; r.k[1] = rock[R_K2 * stride + c]; r.k[2] = rock[R_K3 * stride + c];
  r.wet = rock[R_WET * stride + c]; r.dry = rock[R_DRY * stride + c];
  r.phi = rock[R_PHI * stride + c]; r.rho = rock[R_RHO * stride + c]; r.cp = rock[R_CP * stride + c];
}

// what face_flux reads of the OTHER cell's rock: the permeability along the face's direction and the two conductivities
// (3 of the record's 8 doubles: 40 bytes less per neighbour)
__device__ __forceinline__ void load_rock_face(const double* __restrict__ rock, size_t stride, size_t c, int dir,
                                               RockState& r) {
  const int d = dir - 1;
  const double kd = rock[(size_t)(d == 0 ? R_K1 : (d == 1 ? R_K2 : R_K3)) * stride + c];
  r.k[0] = kd; r.k[1] = kd; r.k[2] = kd;
  r.wet = rock[R_WET * stride + c]; r.dry = rock[R_DRY * stride + c];
  r.phi = 0.0; r.rho = 0.0; r.cp = 0.0;
}

// cell%balance (cell.F90:114-142)
template <int KIND>
__device__ __forceinline__ void cell_balance(const CellState<KIND>& s, const RockState& r, double* bal) {
  using E = EosT<KIND>;
  double m[E::nc], ef = 0.0;
#pragma unroll
  for (int q = 0; q < E::nc; q++) m[q] = 0.0;
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    const double ds = s.rho[p] * s.sat[p];
#pragma unroll
    for (int q = 0; q < E::nc; q++) m[q] += ds * s.x[p][q];
    ef += ds * s.u[p];
  }
#pragma unroll
  for (int q = 0; q < E::nc; q++) bal[q] = r.phi * m[q];
  if constexpr (!E::isothermal) {
    const double er = r.rho * r.cp * s.T;
    bal[E::np - 1] = r.phi * ef + (1.0 - r.phi) * er;
  }
}

struct FaceGeom { double area, d1, d2, d12, gn; int dir; };

__device__ __forceinline__ double harmonic(const FaceGeom& g, double x1, double x2) {
  const double wx = (g.d1 * x2 + g.d2 * x1) / g.d12;
  return (fabs(wx) > 1.e-30) ? x1 * x2 / wx : 0.0;
}

// face%flux (face.F90:443-515): flux[np] from cell 1 to cell 2
template <int KIND>
__device__ __forceinline__ void face_flux(const FaceGeom& g, const CellState<KIND>& a,
                                          const RockState& ra, const CellState<KIND>& b,
                                          const RockState& rb, double* flux) {
  using E = EosT<KIND>;
#pragma unroll
  for (int i = 0; i < E::np; i++) flux[i] = 0.0;
  const int d = g.dir - 1;
  const double ka = (d == 0) ? ra.k[0] : (d == 1 ? ra.k[1] : ra.k[2]);
  const double kb = (d == 0) ? rb.k[0] : (d == 1 ? rb.k[1] : rb.k[2]);
  const double k = harmonic(g, ka * a.permfac, kb * b.permfac);
  if constexpr (!E::isothermal) {
    const double ca = ra.dry + sqrt(a.sat[0]) * (ra.wet - ra.dry);
    const double cb = rb.dry + sqrt(b.sat[0]) * (rb.wet - rb.dry);
    const double cond = harmonic(g, ca, cb);
    const double dtdn = (b.T - a.T) / g.d12;
    flux[E::np - 1] = -cond * dtdn;
  }
  const int pa = (int)a.phases, pb = (int)b.phases, present = pa | pb;
#pragma unroll
  for (int p = 0; p < E::nmob; p++) {
    if (!(present & (1 << p))) continue;
    const double rho_f = (a.sat[p] * a.rho[p] + b.sat[p] * b.rho[p]) / (a.sat[p] + b.sat[p]);
    const double dpdn = ((b.P + b.pc[p]) - (a.P + a.pc[p])) / g.d12;
    const double G = dpdn - rho_f * g.gn;
    const bool up1 = (G <= 0.0);
    const int phup = up1 ? pa : pb;
    if (!(phup & (1 << p))) continue;
    const double kr = up1 ? a.kr[p] : b.kr[p], rho = up1 ? a.rho[p] : b.rho[p];
    const double mu = up1 ? a.mu[p] : b.mu[p], h = up1 ? a.h[p] : b.h[p];
    const double mob = kr * rho / mu;
    const double F = -k * mob * G;
#pragma unroll
    for (int q = 0; q < E::nc; q++) flux[q] += F * (up1 ? a.x[p][q] : b.x[p][q]);
    if constexpr (!E::isothermal) flux[E::np - 1] += h * F;
  }
}

// phase flux of mobile phase p through a face (the entries np+1.. of the reference's flux
// store, face.F90:505-507, flow_simulation.F90:156-205), same arithmetic as face_flux
template <int KIND>
__device__ __forceinline__ double face_phase_flux(const FaceGeom& g, const CellState<KIND>& a,
                                                  const RockState& ra, const CellState<KIND>& b,
                                                  const RockState& rb, int p) {
  using E = EosT<KIND>;
  const int d = g.dir - 1;
  const double ka = (d == 0) ? ra.k[0] : (d == 1 ? ra.k[1] : ra.k[2]);
  const double kb = (d == 0) ? rb.k[0] : (d == 1 ? rb.k[1] : rb.k[2]);
  const double k = harmonic(g, ka * a.permfac, kb * b.permfac);
  const int pa = (int)a.phases, pb = (int)b.phases, present = pa | pb;
  double out = 0.0;
#pragma unroll
  for (int q = 0; q < E::nmob; q++) {
    if (q != p || !(present & (1 << q))) continue;
    const double rho_f = (a.sat[q] * a.rho[q] + b.sat[q] * b.rho[q]) / (a.sat[q] + b.sat[q]);
    const double dpdn = ((b.P + b.pc[q]) - (a.P + a.pc[q])) / g.d12;
    const double G = dpdn - rho_f * g.gn;
    const bool up1 = (G <= 0.0);
    const int phup = up1 ? pa : pb;
    if (!(phup & (1 << q))) continue;
    const double kr = up1 ? a.kr[q] : b.kr[q], rho = up1 ? a.rho[q] : b.rho[q];
    const double mu = up1 ? a.mu[q] : b.mu[q];
    const double F = -k * (kr * rho / mu) * G;
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < E::nc; c++) sum += F * (up1 ? a.x[q][c] : b.x[q][c]);
    out = sum;
  }
  return out;
}

// State-dependent source controls: the record of include/waiwera_hip.h (wai_source_control),
// evaluated on the cell's current -- base or perturbed -- state, so that the FD Jacobian carries
// d(rate)/d(primaries) as it does in the reference, where source_network%update runs inside every
// cell_inflows call (flow_simulation.F90:1469).  Order as the controls are set up
// (source_setup.F90:2381-2412): deliverability (source_control.F90:359-403) or recharge (:553-578)
// gives the rate, the limiter scales it (source_network_node.F90:247-315; water / steam through a
// separator, separator.F90:139-166, :212-260), the direction control zeroes it (:596-620).
struct SrcCtl {
  int kind, direction, limiter, table_coord, n_table;
  double coef, pressure, limit, sep_hf, sep_hg;
  double table[16];
  double factor;
  double sep_more[6];
  double threshold, threshold_pi;
};

// one separator stage (separator.F90:139-166): steam fraction f of a flow of enthalpy h, hw the
// enthalpy of the water that goes on to the next stage
__device__ inline double separator_stage(double hf, double hg, double h, double& hw) {
  if (h <= hf) { hw = h; return 0.0; }
  if (h <= hg) { hw = hf; return (h - hf) / (hg - hf); }
  hw = 0.0;
  return 1.0;
}
// all stages (separator.F90:212-260): total steam rate over the rate fed in
__device__ inline double separator_steam_fraction(const SrcCtl& k, double h) {
  double hw;
  const double f1 = separator_stage(k.sep_hf, k.sep_hg, h, hw);
  if (!(k.sep_more[1] > 0.0)) return f1;
  double steam = f1, water = 1.0 - f1;
  for (int i = 0; i < 3 && k.sep_more[2 * i + 1] > 0.0; i++) {
    const double f = separator_stage(k.sep_more[2 * i], k.sep_more[2 * i + 1], hw, hw);
    steam += f * water;
    water *= 1.0 - f;
  }
  return steam;
}

__device__ inline double ctl_table(const SrcCtl& k, double x) {
  const int n = k.n_table;
  if (x <= k.table[0]) return k.table[1];
  if (x >= k.table[2 * (n - 1)]) return k.table[2 * (n - 1) + 1];
  int i = 0;
  while (x >= k.table[2 * (i + 1)]) i++;
  const double xi = (x - k.table[2 * i]) / (k.table[2 * (i + 1)] - k.table[2 * i]);
  return (1.0 - xi) * k.table[2 * i + 1] + xi * k.table[2 * (i + 1) + 1];
}

// what the source network did to a source in the last network pass (host side, network.hip): net[2 si]
// 0 nothing, 1 its rate is scaled by net[2 si + 1] (member of a limited group), 2 its rate IS
// net[2 si + 1] (output of a reinjector); src/source_network_group.F90, source_network_reinjector.F90
__device__ inline double source_network_rate(const double* net, int si, double rate) {
  if (!net) return rate;
  const double mode = net[2 * si];
  if (mode == 1.0) return rate * net[2 * si + 1];
  if (mode == 2.0) return net[2 * si + 1];
  return rate;
}

template <int KIND>
// commit: an unperturbed residual evaluation -- the threshold deliverability notes its productivity index then
__device__ inline double source_rate(const CellState<KIND>& s, SrcCtl* ctl, int si, double rate,
                                     const double* net = nullptr, bool commit = false) {
  using E = EosT<KIND>;
  if (!ctl) return source_network_rate(net, si, rate);
  const SrcCtl& k = ctl[si];
  const int phases = (int)s.phases;
  double mob[E::nph], sum = 0.0, h = 0.0;
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    mob[p] = (phases & (1 << p)) ? s.kr[p] * s.rho[p] / s.mu[p] : 0.0;
    sum += mob[p];
  }
  if constexpr (!E::isothermal) {
#pragma unroll
    for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) h += (mob[p] / sum) * s.h[p];
  }
  if (k.kind == 1) {
    double pref = k.pressure;
    if (k.table_coord == 1) pref = ctl_table(k, h);
    else if (k.table_coord == 2) pref = ctl_table(k, s.P);
    const double dp = s.P - pref;
    if (k.threshold > 0.0) {
      // deliverability_source_control_iterator (src/source_control.F90:489-503): above the threshold pressure the
      // source keeps its rate and the index that would give exactly that rate is noted (calculate_PI_from_rate,
      // :407-466); below it the deliverability rate with the noted index applies if it is the smaller production
      if (s.P < k.threshold) {
        double qd = 0.0;
#pragma unroll
        for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) qd = qd - k.threshold_pi * s.permfac * mob[p] * dp;
        if (qd > rate) rate = qd;
      } else if (commit) {
        const double fac = sum * dp * s.permfac;
        if (fabs(fac) > 1.0e-9) ctl[si].threshold_pi = fabs(rate) / fac;   // the control records are the sweep's to note into (commit: unperturbed evaluations only)
      }
    } else {
      rate = 0.0;
#pragma unroll
      for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) rate = rate - k.coef * s.permfac * mob[p] * dp;
    }
  } else if (k.kind == 2) {
    rate = -k.coef * (s.P - k.pressure);
  }
  if (k.limiter) {
    double r = rate;
    if (k.limiter > 1) {
      if (rate < 0.0) {
        const double f = separator_steam_fraction(k, h);
        r = k.limiter == 2 ? (1.0 - f) * rate : f * rate;
      } else r = 0.0;
    }
    const double a = fabs(r);
    if (a > k.limit && a > 1.0e-6) rate = rate * (k.limit / a);
  }
  if (k.direction == 1 && !(rate < 0.0)) rate = 0.0;
  if (k.direction == 2 && !(rate > 0.0)) rate = 0.0;
  if (k.factor != 0.0) rate *= k.factor;
  return source_network_rate(net, si, rate);
}

// source term (source.F90:386-480, fluid.F90:377-453): flow[np] for one source
template <int KIND>
__device__ __forceinline__ void source_flow(const CellState<KIND>& s, double rate, double enth,
                                            int comp, double* flow) {
  using E = EosT<KIND>;
#pragma unroll
  for (int k = 0; k < E::np; k++) flow[k] = 0.0;
  double h = 0.0;
  int component;
  if (rate > 0.0) {
    component = comp <= 0 ? 1 : comp;
    h = enth;
    if (component - 1 < E::np) {
#pragma unroll
      for (int k = 0; k < E::np; k++) if (k == component - 1) flow[k] = rate;
    }
  } else {
    component = comp <= 0 ? 0 : comp;
    const int phases = (int)s.phases;
    double frac[E::nph] = {}, sum = 0.0;
    if (component < E::np) {
#pragma unroll
      for (int p = 0; p < E::nph; p++) {
        frac[p] = (phases & (1 << p)) ? s.kr[p] * s.rho[p] / s.mu[p] : 0.0;
        sum += frac[p];
      }
#pragma unroll
      for (int p = 0; p < E::nph; p++) frac[p] /= sum;
      if constexpr (!E::isothermal) {
#pragma unroll
        for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) h += frac[p] * s.h[p];
      }
    }
    if (component <= 0) {
      double cf[E::nc], cs = 0.0;
#pragma unroll
      for (int q = 0; q < E::nc; q++) {
        cf[q] = 0.0;
#pragma unroll
        for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) cf[q] += frac[p] * s.x[p][q];
        cs += cf[q];
      }
#pragma unroll
      for (int q = 0; q < E::nc; q++) flow[q] = rate * (cf[q] / cs);
    } else {
#pragma unroll
      for (int k = 0; k < E::np; k++) if (k == component - 1) flow[k] = rate;
    }
  }
  if constexpr (!E::isothermal) {
    if (component < E::np) flow[E::np - 1] += h * rate;
  }
}

// ---- Brent on the saturation line (root_finder.F90:127-248 / eos_we.F90:530-553) -----------
struct SatLine { double p0, t0, p1, t1, g0, g1; int thermo; };  // g: gas partial pressure (eos_wge.F90:678-701)
__device__ inline double satline_diff(double x, const SatLine& c) {
  const double P = (1.0 - x) * c.p0 + x * c.p1, T = (1.0 - x) * c.t0 + x * c.t1;
  const double Pg = (1.0 - x) * c.g0 + x * c.g1;
  double ps = 0.0;
  th::sat_pressure(c.thermo, T, ps);
  return P - Pg - ps;
}

template <class F>
__device__ inline int brent_unit(const F& fn, double& root) {
  const double xtol = 1.e-8, ftol = 1.e-8, small = 1.e-16;
  double a = 0.0, b = 1.0;
  double fa = fn(a), fb = fn(b);
  root = 0.0;
  if (fa * fb > 0.0) return 1;
  double c = b, fc = fb, d = 0.0, e = 0.0;
  bool found = false;
  for (int iter = 1; iter <= 100; iter++) {
    if (fb * fc > 0.0) { c = a; fc = fa; d = b - a; e = d; }
    if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    const double dx = 0.5 * (c - b);
    if (fabs(dx) <= xtol || fabs(fb) <= ftol) { found = true; break; }
    if (fabs(e) >= xtol && fabs(fa) > fabs(fb)) {
      const double s = fb / fa;
      double p, q;
      if (fabs(a - c) <= small) { p = 2.0 * dx * s; q = 1.0 - s; }
      else {
        q = fa / fc;
        const double r = fb / fc;
        p = s * (2.0 * dx * q * (q - r) - (b - a) * (r - 1.0));
        q = (q - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q = -q; else p = -p;
      const double pc = fmin(3.0 * dx * q - fabs(xtol * q), fabs(e * q));
      if (2.0 * p < pc) { e = d; d = p / q; }
      else { d = dx; e = d; }
    } else { d = dx; e = d; }
    a = b; fa = fb;
    if (fabs(d) > xtol) b += d;
    else b += (dx >= 0.0 ? xtol : -xtol);
    fb = fn(b);
  }
  root = b;
  return found ? 0 : 2;
}

__device__ inline int brent_satline(const SatLine& sl, double& root) {
  return brent_unit([&](double x) { return satline_diff(x, sl); }, root);
}

// eos%transition (eos_we.F90:149-323; eos_wge.F90:154-346; eos_w.F90:103-122).
// prim/oldp are unscaled primaries; region is updated in place.  Returns err.
__device__ __forceinline__ double lerp_clamped(double xi, double a, double b) {
  if (xi <= 0.0) return a;
  if (xi >= 1.0) return b;
  return (1.0 - xi) * a + xi * b;
}

template <int KIND>
__device__ inline int eos_transition(int thermo, const double* oldp, double* prim, int old_region,
                                     double old_temperature, int& region, bool& transition) {
  transition = false;
  if constexpr (KIND == EOS_W) {
    (void)thermo; (void)oldp; (void)old_region; (void)old_temperature; (void)region;
    return 0;
  } else {
    constexpr bool wce = is_wge<KIND>;
    constexpr int ig_ = wce ? 2 : 0;
    const double small = 1.e-6;
    if (old_region == 4) {
      const double sv = prim[1];
      int new_region = 0;
      if (sv < 0.0) new_region = 1; else if (sv > 1.0) new_region = 2;
      if (!new_region) return 0;
      const double bound = (new_region == 1) ? 0.0 : 1.0;
      const double pfac = (new_region == 1) ? 1.0 + small : 1.0 - small;
      if constexpr (wce) prim[2] = fmax(0.0, fmin(prim[2], prim[0]));
      const double v1 = oldp[1], v2 = prim[1], vmax = fmax(fabs(v1), fabs(v2));
      if (fabs(v2 - v1) >= 1.e-8 * vmax) {
        const double vs1 = v1 / vmax, vs2 = v2 / vmax, ys = bound / vmax;
        double xi = (ys - vs1) / (vs2 - vs1);
        xi = (1.0 - xi) * 0.0 + xi * 1.0;
        const double ip = lerp_clamped(xi, oldp[0], prim[0]);
        const double ig = wce ? lerp_clamped(xi, oldp[ig_], prim[ig_]) : 0.0;
        const double iw = ip - ig;
        prim[0] = pfac * iw + ig;
        if constexpr (wce) prim[2] = ig;
        double t;
        const int err = th::sat_temperature(thermo, iw, t);
        if (err == 0) { prim[1] = t; region = new_region; transition = true; }
        return err;
      }
      double ps;
      const int err = th::sat_pressure(thermo, old_temperature, ps);
      if (err == 0) {
        prim[0] = pfac * ps + (wce ? prim[ig_] : 0.0);
        prim[1] = old_temperature;
        region = new_region;
        transition = true;
      }
      return err;
    }
    double ps;
    const int err = th::sat_pressure(thermo, prim[1], ps);
    if (err) return err;
    const double pw = prim[0] - (wce ? prim[ig_] : 0.0);
    if ((old_region == 1 && pw < ps) || (old_region == 2 && pw > ps)) {
      if constexpr (wce) prim[2] = fmax(0.0, fmin(prim[2], prim[0]));
      SatLine sl{oldp[0], oldp[1], prim[0], prim[1], wce ? oldp[ig_] : 0.0, wce ? prim[ig_] : 0.0, thermo};
      double root;
      if (brent_satline(sl, root) == 0) {
        const double ig = wce ? lerp_clamped(root, oldp[ig_], prim[ig_]) : 0.0;
        prim[0] = lerp_clamped(root, oldp[0], prim[0]);
        if constexpr (wce) prim[2] = ig;
      } else prim[0] = ps + (wce ? prim[ig_] : 0.0);
      prim[1] = (old_region == 1) ? small : 1.0 - small;
      region = 4;
      transition = true;
    }
    return 0;
  }
}

// eos_wse transitions (eos_wse.F90:203-617): boiling / condensing along the *brine* saturation
// line, then halite precipitation / dissolution.  cur_old_region = the cell's region on entry
// (fluid%old_region as set by flow_simulation.F90:2502), last_old_region = that field of the
// last-iteration fluid.
template <bool GAS>
__device__ inline int wse_to_single_phase(int thermo, const double* oldp, double* prim, int old_region,
                                          double old_t, int new_region, int& region, bool& transition) {
  const double small = 1.e-6;
  const bool old_halite = wse_halite(old_region);
  const int nwr = wse_water_region(new_region);
  const double ss = old_halite ? prim[2] : 0.0;
  const double bound = (nwr == 1) ? 0.0 : 1.0 - ss;
  const double pfac = (nwr == 1) ? 1.0 + small : 1.0 - small;
  int err = 0;
  if constexpr (GAS) {   // eos_wsge.F90:222-226
    prim[2] = fmax(0.0, prim[2]);
    prim[3] = fmax(0.0, fmin(prim[3], prim[0]));
  }
  const double v1 = oldp[1], v2 = prim[1], vmax = fmax(fabs(v1), fabs(v2));
  if (fabs(v2 - v1) >= 1.e-8 * vmax) {
    const double xi = (bound / vmax - v1 / vmax) / (v2 / vmax - v1 / vmax);
    const double ip = lerp_clamped(xi, oldp[0], prim[0]), is = lerp_clamped(xi, oldp[2], prim[2]);
    double ig = 0.0;
    if constexpr (GAS) ig = lerp_clamped(xi, oldp[3], prim[3]);
    const double ibp = ip - ig;   // interpolated brine pressure
    double t, xs;
    prim[0] = pfac * ibp + ig;
    prim[2] = GAS ? is : fmax(0.0, is);
    if constexpr (GAS) prim[3] = ig;
    if (nwr == 1) {
      if (old_halite) err = salt::halite_solubility_two_phase(thermo, ibp, xs);
      else xs = prim[2];
      if (!err) err = salt::brine_sat_temperature(thermo, ibp, xs, t);
    } else err = th::sat_temperature(thermo, ibp, t);
    if (!err) { prim[1] = t; region = new_region; transition = true; }
  } else {
    double xs, ps;
    if (nwr == 1) {
      if (old_halite) err = salt::halite_solubility(old_t, xs);
      else xs = oldp[2];
      if (!err) { xs = fmax(0.0, xs); err = salt::brine_sat_pressure(thermo, old_t, xs, ps); }
    } else err = th::sat_pressure(thermo, old_t, ps);
    if (!err) {
      double pgk = 0.0;
      if constexpr (GAS) pgk = prim[3];
      prim[0] = pfac * ps + pgk; prim[1] = old_t; region = new_region; transition = true;
    }
  }
  return err;
}

template <bool GAS>
__device__ inline int eos_transition_wse(int thermo, const double* oldp, double* prim, int old_region,
                                         double old_t, int cur_old_region, int last_old_region,
                                         int& region, bool& transition) {
  const double small = 1.e-6;
  transition = false;
  const int owr = wse_water_region(old_region);
  const bool old_halite = wse_halite(old_region);
  int err = 0;
  if (owr == 4) {
    const int off = old_halite ? 4 : 0;
    const double sv = prim[1];
    if (sv < 0.0) err = wse_to_single_phase<GAS>(thermo, oldp, prim, old_region, old_t, off + 1, region, transition);
    else {
      const double ss = old_halite ? prim[2] : 0.0;
      if (sv > 1.0 - ss) err = wse_to_single_phase<GAS>(thermo, oldp, prim, old_region, old_t, off + 2, region, transition);
    }
  } else {
    double xs, ps;
    if (owr == 1) {
      if (old_halite) err = salt::halite_solubility(prim[1], xs);
      else xs = prim[2];
      if (!err) { xs = fmax(0.0, xs); err = salt::brine_sat_pressure(thermo, prim[1], xs, ps); }
    } else err = th::sat_pressure(thermo, prim[1], ps);
    double pwat = prim[0];
    if constexpr (GAS) pwat -= prim[3];
    if (!err && ((owr == 1 && pwat < ps) || (owr == 2 && pwat > ps))) {
      prim[2] = fmax(0.0, prim[2]);
      double a3 = 0.0, b3 = 0.0;
      if constexpr (GAS) { prim[3] = fmax(0.0, fmin(prim[3], prim[0])); a3 = oldp[3]; b3 = prim[3]; }
      const double a0 = oldp[0], a1 = oldp[1], a2 = oldp[2], b0 = prim[0], b1 = prim[1], b2 = prim[2];
      double root;
      const int rerr = brent_unit([&](double x) {   // eos_wse_saturation_difference :942-974
        const double P = (1.0 - x) * a0 + x * b0, T = (1.0 - x) * a1 + x * b1;
        double xq = (1.0 - x) * a2 + x * b2, Ps = 0.0;
        if (owr == 1) {
          if (old_halite) salt::halite_solubility(T, xq);
          salt::brine_sat_pressure(thermo, T, xq, Ps);
        } else th::sat_pressure(thermo, T, Ps);
        return P - ((1.0 - x) * a3 + x * b3) - Ps;   // eos_wsge.F90:1002-1036
      }, root);
      if (rerr == 0) {
        prim[0] = lerp_clamped(root, a0, b0);
        prim[2] = lerp_clamped(root, a2, b2);
        if constexpr (GAS) prim[3] = lerp_clamped(root, a3, b3);
      } else prim[0] = ps + b3;
      const double ss = old_halite ? prim[2] : 0.0;
      prim[1] = (owr == 1) ? small : 1.0 - ss - small;
      region = old_halite ? 8 : 4;
      transition = true;
    }
  }
  if (err) return err;
  // halite_transition :413-525
  double t, sol;
  switch (region) {
    case 1: case 4:
      if (region == 1) t = prim[1];
      else {
        double bpk = prim[0];
        if constexpr (GAS) bpk -= prim[3];
        err = salt::brine_sat_temperature(thermo, bpk, prim[2], t);
      }
      if (!err) {
        err = salt::halite_solubility(t, sol);
        if (prim[2] > sol) { prim[2] = small; region += 4; transition = true; }
      }
      break;
    case 2:
      if (prim[2] > 0.0) { prim[2] = small; region = 6; transition = true; }
      break;
    case 5: case 8:
      if (prim[2] < 0.0) {
        if (region == 5) {
          err = salt::halite_solubility(prim[1], sol);
          if (!err) { prim[2] = sol - small; region = 1; transition = true; }
        } else if (cur_old_region == 6 || last_old_region == 6) {
          prim[2] = small; region = 4; transition = true;
        } else {
          double bpk = prim[0];
          if constexpr (GAS) bpk -= prim[3];
          err = salt::halite_solubility_two_phase(thermo, bpk, sol);
          if (!err) { prim[2] = sol - small; region = 4; transition = true; }
        }
      }
      break;
    case 6:
      if (prim[2] < 0.0) { prim[2] = 0.0; region = 2; transition = true; }
      break;
  }
  return err;
}

// eos%check_primary_variables (eos_we.F90:486-526; eos_w.F90:232-255; eos_wge.F90:573-635:
// the gas partial pressure is clamped to [0, (1-1e-6) P] and reported as `changed`)
template <int KIND>
__device__ __forceinline__ int eos_check_primary(double* prim, int region, bool& changed) {
  changed = false;
  if constexpr (is_wsge<KIND>) {   // eos_wsge.F90:890-958
    const double small = 1.e-6;
    if (!(prim[0] > 0.0)) return 1;
    const double maxpp = (1.0 - small) * prim[0];
    if (prim[3] > maxpp) { prim[3] = maxpp; changed = true; }
    else if (prim[3] < 0.0) { prim[3] = 0.0; changed = true; }
    if (prim[2] < 0.0) { prim[2] = 0.0; changed = true; }
    else if (prim[2] > 1.0) return 1;
    if (prim[0] - prim[3] > 100.e6) return 1;
    if (wse_water_region(region) == 4) { if (prim[1] < -1.0 || prim[1] > 2.0) return 1; }
    else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
    return 0;
  }
  if constexpr (KIND == EOS_WSE) {   // eos_wse.F90:891-938
    if (prim[2] < 0.0) { prim[2] = 0.0; changed = true; }
    else if (prim[2] > 1.0) return 1;
    if (prim[0] < 0.0 || prim[0] > 100.e6) return 1;
    if (wse_water_region(region) == 4) { if (prim[1] < -1.0 || prim[1] > 2.0) return 1; }
    else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
    return 0;
  }
  if constexpr (is_wge<KIND>) {
    const double small = 1.e-6;
    if (!(prim[0] > 0.0)) return 1;
    const double maxpp = (1.0 - small) * prim[0];
    if (prim[2] > maxpp) { prim[2] = maxpp; changed = true; }
    else if (prim[2] < 0.0) { prim[2] = 0.0; changed = true; }
    if (prim[0] - prim[2] > 100.e6) return 1;
  } else {
    const double p = prim[0];
    if (p < 0.0 || p > 100.e6) return 1;
  }
  if constexpr (KIND != EOS_W) {
    if (region == 4) { if (prim[1] < -1.0 || prim[1] > 2.0) return 1; }
    else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
  }
  return 0;
}

}  // namespace wai

/* wo_physics.c -- CPU restatement of Waiwera's per-cell / per-face arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see wai_oracle.h).  Plain C, fp64, written to be read next to the
 * reference: every function cites the reference lines it follows.
 */
#include "wai_oracle.h"
#include "if97_tables.h"
#include <math.h>
#include <string.h>

#define TC_K 273.15            /* src/thermodynamics.F90:37 */
#define RCONST 0.461526e3      /* src/thermodynamics.F90:36 */
#define TCRITICALK 647.096     /* src/IAPWS.F90:273 */
#define TCRITICAL (TCRITICALK - TC_K)
#define PCRITICAL 22.064e6     /* src/IAPWS.F90:275 */
#define DCRITICAL 322.0        /* src/IAPWS.F90:276 */

/* integer power by repeated multiplication.  The reference evaluates the same powers through a
 * precomputed addition chain (src/powertable.F90:261-278); both are products of the same
 * factors and agree to a few ulp -- well inside the 1e-7 tolerance of IAPWS_test.F90:57. */
static double ipow(double x, int n) {
  if (n < 0) { x = 1.0 / x; n = -n; }
  double r = 1.0;
  while (n) {
    if (n & 1) r *= x;
    x *= x;
    n >>= 1;
  }
  return r;
}

/* src/IAPWS.F90:503-542 */
int wo_region1(double p, double t, double *rho, double *u) {
  if (!((t <= 350.0) && (p <= 100.e6))) return 1;
  const double pstar = 16.53e6, tstar = 1386.0;
  double tk = t + TC_K, rt = RCONST * tk, pi = p / pstar, tau = tstar / tk;
  double a = 7.1 - pi, b = tau - 1.222;
  double gampi = 0.0, gamt = 0.0;
  for (int i = 0; i < 34; i++) {
    int I = WO_R1_I[i], J = WO_R1_J[i];
    gampi += (WO_R1_N[i] * I) * ipow(a, I - 1) * ipow(b, J);
    gamt += (WO_R1_N[i] * J) * ipow(a, I) * ipow(b, J - 1);
  }
  gampi = -gampi;
  *rho = pstar / (rt * gampi);
  *u = rt * (tau * gamt - pi * gampi);
  return 0;
}

/* src/IAPWS.F90:596-639 */
int wo_region2(double p, double t, double *rho, double *u) {
  if (!((t <= 800.0) && (p <= 100.e6))) return 1;
  const double pstar = 1.0e6, tstar = 540.0;
  double tk = t + TC_K, rt = RCONST * tk, pi = p / pstar, tau = tstar / tk;
  double b = tau - 0.5;
  double gamt0 = 0.0, gampir = 0.0, gamtr = 0.0;
  for (int i = 0; i < 9; i++)
    gamt0 += (WO_R2_N0[i] * WO_R2_J0[i]) * ipow(tau, WO_R2_J0[i] - 1);
  for (int i = 0; i < 43; i++) {
    int I = WO_R2_I[i], J = WO_R2_J[i];
    gampir += (WO_R2_N[i] * I) * ipow(pi, I - 1) * ipow(b, J);
    gamtr += (WO_R2_N[i] * J) * ipow(pi, I) * ipow(b, J - 1);
  }
  double gampi = 1.0 / pi + gampir;
  *rho = pstar / (rt * gampi);
  *u = rt * (tau * (gamt0 + gamtr) - pi * gampi);
  return 0;
}

/* src/IAPWS.F90:762-789 */
int wo_sat_pressure(double t, double *p) {
  if (!((t >= 0.0) && (t <= TCRITICAL))) return 1;
  const double *n = WO_SAT_N - 1; /* 1-based like the formulation */
  double tk = t + TC_K;
  double theta = tk + n[9] / (tk - n[10]);
  double theta2 = theta * theta;
  double a = theta2 + n[1] * theta + n[2];
  double b = n[3] * theta2 + n[4] * theta + n[5];
  double c = n[6] * theta2 + n[7] * theta + n[8];
  double x = 2.0 * c / (-b + sqrt(b * b - 4.0 * a * c));
  x = x * x;
  *p = 1.0e6 * x * x;
  return 0;
}

/* src/IAPWS.F90:793-818 */
int wo_sat_temperature(double p, double *t) {
  if (!((p >= 611.213) && (p <= PCRITICAL))) return 1;
  const double *n = WO_SAT_N - 1;
  double beta2 = sqrt(p / 1.0e6);
  double beta = sqrt(beta2);
  double e = beta2 + n[3] * beta + n[6];
  double f = n[1] * beta2 + n[4] * beta + n[7];
  double g = n[2] * beta2 + n[5] * beta + n[8];
  double d = 2.0 * g / (-f - sqrt(f * f - 4.0 * e * g));
  double x = n[10] + d;
  *t = 0.5 * (n[10] + d - sqrt(x * x - 4.0 * (n[9] + n[10] * d))) - TC_K;
  return 0;
}

/* src/IAPWS.F90:412-443 */
double wo_viscosity(double t, double rho) {
  double tk = t + TC_K, tau = tk / TCRITICALK, del = rho / DCRITICAL;
  double it = 1.0 / tau;
  double s0 = 0.0;
  for (int k = 0; k < 4; k++) s0 += WO_VISC_H0[k] * ipow(it, k);
  double mu0 = 100.0 * sqrt(tau) / s0;
  double a = it - 1.0, b = del - 1.0, s1 = 0.0;
  for (int i = 0; i < 21; i++)
    s1 += ipow(a, WO_VISC_I[i]) * WO_VISC_H1[i] * ipow(b, WO_VISC_J[i]);
  double mu1 = exp(del * s1);
  return 1.0e-6 * mu0 * mu1;
}

/* src/IAPWS.F90:317-365 */
int wo_phase_composition(int region, double p, double t) {
  if (region == 4) return 3;
  if (t <= TCRITICAL) {
    if (region == 1) return 1;
    if (region == 2) return 2;
    if (region == 3) {
      double ps;
      if (wo_sat_pressure(t, &ps) == 0) return (p >= ps) ? 1 : 2;
      return -1;
    }
    return 0;
  }
  return (p <= PCRITICAL) ? 2 : 4;
}

/* ---- thermodynamic formulation dispatch ("thermodynamics": "iapws" | "ifc67") -------------- */
/* region 1 = liquid water, 2 = steam */
static int th_props(const wo_eos *e, int region, double p, double t, double *rho, double *u) {
  if (e->thermo == 1)
    return region == 1 ? wo_ifc67_region1(p, t, 350.0, rho, u) : wo_ifc67_region2(p, t, rho, u);
  return region == 1 ? wo_region1(p, t, rho, u) : wo_region2(p, t, rho, u);
}
static double th_viscosity(const wo_eos *e, int region, double t, double p, double rho) {
  return e->thermo == 1 ? wo_ifc67_viscosity(region, t, p, rho) : wo_viscosity(t, rho);
}
static int th_sat_pressure(const wo_eos *e, double t, double *p) {
  return e->thermo == 1 ? wo_ifc67_sat_pressure(t, p) : wo_sat_pressure(t, p);
}
static int th_sat_temperature(const wo_eos *e, double p, double *t) {
  return e->thermo == 1 ? wo_ifc67_sat_temperature(p, t) : wo_sat_temperature(p, t);
}
static int th_phase_composition(const wo_eos *e, int region, double p, double t) {
  return e->thermo == 1 ? wo_ifc67_phase_composition(region) : wo_phase_composition(region, p, t);
}

/* ---- CO2 NCG thermodynamics -------------------------------------------------------------- */
#define CO2_MW 44.01             /* src/ncg_co2_thermodynamics.F90:14 */
#define WATER_MW 18.01528        /* src/thermodynamics.F90:38 */
#define GAS_CONSTANT 8.3144598   /* src/thermodynamics.F90:39 */
/* published correlation data (Battistelli et al. 1997 / TOUGH2 EWASG), as held at
 * src/ncg_co2_thermodynamics.F90:15-29 */
static const double CO2_HENRY[6] = {0.783666, 1.96025, 8.20574, -7.40674, 2.18380, -0.220999};
static const double CO2_VISC_P[5] = {0.0, 10.0, 15.0, 20.0, 30.0}; /* MPa */
static const double CO2_VISC_C[5][5] = { /* [coefficient][pressure node] */
    {1.3578, 3.9189, 9.6607, 13.1566, 14.7968},
    {4.9227e-3, -35.984e-3, -135.479e-3, -179.352e-3, -160.731e-3},
    {-2.9661e-6, 0.25825e-3, 0.90087e-3, 1.12474e-3, 0.850257e-3},
    {2.8529e-9, -7.1178e-7, -2.4727e-6, -2.98864e-6, -1.99076e-6},
    {-2.1829e-12, 6.9578e-10, 2.4156e-9, 2.85911e-9, 1.73423e-9}};

static double horner(const double *a, int n, double x) { /* src/utils.F90:224-241 */
  double p = a[n - 1];
  for (int i = n - 2; i >= 0; i--) p = a[i] + x * p;
  return p;
}

/* src/ncg_co2_thermodynamics.F90:83-112 */
int wo_co2_properties(double partial_pressure, double t, double *rho, double *h) {
  double tk = t + TC_K;
  double pp = partial_pressure * 1.0e-6;
  double tc = pow(0.01 * tk, 3.3333333333);
  double hci = 1.667 + 0.001542 * tk - 0.7948 * log10(tk) - 41.35 / tk;
  *h = 1.e6 * (hci - 0.3571 * pp * (1.0 + 0.07576 * pp) / tc);
  double vc = 0.00018882 * tk - pp * (0.0824 + 0.01249 * pp) / tc;
  *rho = pp / vc;
  return 0;
}

/* src/ncg_co2_thermodynamics.F90:116-135 */
double wo_co2_henrys_constant(double t) { return 1.e8 * horner(CO2_HENRY, 6, t / 100.0); }

/* d(ln H)/dT (:183-205) and heat of solution (ncg_thermodynamics.F90:186-231) */
double wo_co2_energy_solution(double t) {
  double d[5];
  for (int i = 0; i < 5; i++) d[i] = (i + 1) * CO2_HENRY[i + 1]; /* polynomial_derivative */
  double H = wo_co2_henrys_constant(t);
  double hd = 1.e8 * horner(d, 5, t / 100.0) / (H * 100.0);
  double tk = t + TC_K;
  return -1.e3 * GAS_CONSTANT * tk * tk * hd / CO2_MW;
}

/* src/ncg_co2_thermodynamics.F90:236-260: coefficients linearly interpolated in pressure */
int wo_co2_viscosity(double partial_pressure, double t, double *visc) {
  if (!(partial_pressure <= 300.e5)) return 1;
  double p = partial_pressure / 1.e6, coefs[5];
  int i = 0;
  if (p <= CO2_VISC_P[0]) { for (int k = 0; k < 5; k++) coefs[k] = CO2_VISC_C[k][0]; }
  else if (p >= CO2_VISC_P[4]) { for (int k = 0; k < 5; k++) coefs[k] = CO2_VISC_C[k][4]; }
  else {
    while (i < 3 && p >= CO2_VISC_P[i + 1]) i++;
    double xi = (p - CO2_VISC_P[i]) / (CO2_VISC_P[i + 1] - CO2_VISC_P[i]);
    for (int k = 0; k < 5; k++) coefs[k] = (1.0 - xi) * CO2_VISC_C[k][i] + xi * CO2_VISC_C[k][i + 1];
  }
  *visc = 1.e-5 * horner(coefs, 5, t);
  return 0;
}

/* ---- air NCG thermodynamics: src/ncg_air_thermodynamics.F90 -------------------------------- */
/* correlation data as held at :15-41 (Irvine & Liley 1984 enthalpy; D'Amore & Truesdell 1988,
 * Cramer 1982, Cygan 1991 Henry's constants of N2 / O2; Hirschfelder et al. 1954 viscosity) */
#define AIR_MW 28.96
#define IS_WGE(e) ((e)->kind == WO_EOS_WCE || (e)->kind == WO_EOS_WAE)
/* salt family: wse, and with a gas (eos_wsge.F90) wsce / wsae: 4th primary = gas partial pressure */
#define IS_WSGE(e) ((e)->kind == WO_EOS_WSCE || (e)->kind == WO_EOS_WSAE)
#define IS_SALT(e) ((e)->kind == WO_EOS_WSE || IS_WSGE(e))
#define GAS_IS_AIR(e) ((e)->kind == WO_EOS_WAE || (e)->kind == WO_EOS_WSAE)
static const double AIR_ENTHALPY[4] = {1.20740, 9.24502, 0.115984, -5.63568e-4};
static const double AIR_WEIGHT[2] = {0.79, 0.21};
static const double AIR_HENRY_P0[2] = {1.01325e5, 1.e5};
static const double AIR_HENRY[2][7] = {
    {0.513726, 1.58603, -5.9378e-1, -6.98282e-1, 5.10330e-1, -1.21388e-1, 1.00041e-2},
    {0.26234, 0.610628, 7.00732e-1, -0.139299e1, 7.13850e-1, -1.54216e-1, 1.23190e-2}};
/* ncg_air_properties :90-114 */
int wo_air_properties(double partial_pressure, double t, double *rho, double *h) {
  double tk = t + TC_K;
  double shift = horner(AIR_ENTHALPY, 4, (0.01 + TC_K) / 100.0); /* zero enthalpy at the triple point, :77-80 */
  *rho = partial_pressure * AIR_MW / (1.e3 * GAS_CONSTANT * 1.0 * tk);
  *h = 1.e4 * (horner(AIR_ENTHALPY, 4, tk / 100.0) - shift);
  return 0;
}
/* ncg_air_henrys_constant :118-137 */
static void air_henry_constituents(double t, double *hc) {
  for (int i = 0; i < 2; i++) hc[i] = 1.e5 * AIR_HENRY_P0[i] * horner(AIR_HENRY[i], 7, t / 100.0);
}
double wo_air_henrys_constant(double t) {
  double hc[2];
  air_henry_constituents(t, hc);
  return AIR_WEIGHT[0] * hc[0] + AIR_WEIGHT[1] * hc[1];
}
/* ncg_air_henrys_derivative :174-199 and the energy of solution (ncg_thermodynamics.F90:187-231) */
double wo_air_energy_solution(double t) {
  double hc[2], d = 0.0;
  air_henry_constituents(t, hc);
  for (int i = 0; i < 2; i++) {
    double dc[6];
    for (int k = 1; k < 7; k++) dc[k - 1] = k * AIR_HENRY[i][k]; /* polynomial_derivative */
    double dhinv = 1.e5 * horner(dc, 6, t / 100.0);
    d += AIR_WEIGHT[i] * (AIR_HENRY_P0[i] * dhinv / (hc[i] * 100.0));
  }
  double tk = t + TC_K;
  return -1.e3 * GAS_CONSTANT * tk * tk * d / AIR_MW;
}
/* ncg_air_mixture_viscosity :260-338, gas phase (the liquid keeps the water viscosity) */
static double air_covis(double trd, double c, double ome, double rm, double f) {
  return 266.93e-7 * sqrt(rm * trd * f) / (c * c * ome * trd);
}
double wo_air_mixture_viscosity(double water_viscosity, double t, double xg) {
  const double fair = 97.0, fwat = 363.0, cair = 3.617, cwat = 2.655;
  const double fmix = sqrt(fair * fwat), cmix = 0.5 * (cair + cwat);
  const double rm1 = AIR_MW, rm2 = WATER_MW;
  double w = xg / rm1, x1 = w / (w + (1.0 - xg) / rm2), x2 = 1.0 - x1;
  double tk = t + TC_K, trd1 = tk / fair, trd3 = tk / fmix;
  double ome1 = (1.188 - 0.051 * trd1) / trd1;
  double ome3 = (1.48 - 0.412 * log(trd3)) / trd3;
  double ard = 1.095 / trd3;
  double rm3 = 2.0 * rm1 * rm2 / (rm1 + rm2);
  double vis1 = air_covis(trd1, cair, ome1, rm1, fair);
  double vis2 = 10.0 * water_viscosity;
  double vis3 = air_covis(trd3, cmix, ome3, rm3, fmix);
  double z1 = x1 * x1 / vis1 + 2.0 * x2 * x1 / vis3 + x2 * x2 / vis2;
  double g = x1 * x1 * rm1 / rm2, h = x2 * x2 * rm2 / rm1;
  double ee = (2.0 * x1 * x2 * rm1 * rm2 / (rm3 * rm3)) * vis3 / (vis1 * vis2);
  double z2 = 0.6 * ard * (g / vis1 + ee + h / vis2);
  double z3 = 0.6 * ard * (g + ee * (vis1 + vis2) - 2.0 * x1 * x2 + h);
  return 0.1 * (1.0 + z3) / (z1 + z2);
}

/* src/ncg_thermodynamics.F90:155-167 */
double wo_ncg_mole_to_mass(double xmole, double mw) {
  double w = xmole * mw;
  return w / (w + (1.0 - xmole) * WATER_MW);
}

/* ---------------------------------------------------------------------------------------- */
/* two-point table lookup with end clamping: interpolation_table "interpolate" on a 2-row table
 * (src/interpolation.F90:202-222 index rule, :388-404 linear interpolant, :494-510 clamping) */
static double lin2(double x, double x0, double x1, double y0, double y1) {
  if (x <= x0) return y0;
  if (x >= x1) return y1;
  double xi = (x - x0) / (x1 - x0);
  return (1.0 - xi) * y0 + xi * y1;
}

/* src/relative_permeability.F90:197-492.  par: linear [l0,l1,v0,v1]; pickens [power];
 * corey/grant [slr,ssr]; van Genuchten [lambda,slr,sls,sum_unity,ssr] */
/* interpolation_table_type (src/interpolation.F90): find :202-222, clamped ends :494-510, linear
 * :388-404, step :715-720, pchip derivatives :810-885 and polynomial :891-925 */
static int sign_test(double a, double b) {
  if ((a > 0.0 && b > 0.0) || (a < 0.0 && b < 0.0)) return 1;
  if (a == 0.0 || b == 0.0) return 0;
  return -1;
}
static void pchip_deriv(int n, const double *x, const double *f, double *d) {
  if (n == 1) { d[0] = 0.0; return; }
  double h1 = x[1] - x[0], del1 = (f[1] - f[0]) / h1;
  if (n == 2) { d[0] = d[1] = del1; return; }
  double h2 = x[2] - x[1], del2 = (f[2] - f[1]) / h2, hsum = h1 + h2;
  double w1 = (h1 + hsum) / hsum, w2 = -h1 / hsum, dmax, dmin;
  d[0] = w1 * del1 + w2 * del2;
  if (sign_test(d[0], del1) <= 0) d[0] = 0.0;
  else if (sign_test(del1, del2) < 0) { dmax = 3.0 * del1; if (fabs(d[0]) > fabs(dmax)) d[0] = dmax; }
  for (int i = 1; i < n - 1; i++) {
    if (i > 1) { h1 = h2; h2 = x[i + 1] - x[i]; hsum = h1 + h2; del1 = del2; del2 = (f[i + 1] - f[i]) / h2; }
    if (sign_test(del1, del2) > 0) {
      w1 = (hsum + h1) / (3.0 * hsum); w2 = (hsum + h2) / (3.0 * hsum);
      dmax = fmax(fabs(del1), fabs(del2)); dmin = fmin(fabs(del1), fabs(del2));
      d[i] = dmin / (w1 * (del1 / dmax) + w2 * (del2 / dmax));
    } else d[i] = 0.0;
  }
  w1 = -h2 / hsum; w2 = (h2 + hsum) / hsum;
  d[n - 1] = w1 * del1 + w2 * del2;
  if (sign_test(d[n - 1], del2) <= 0) d[n - 1] = 0.0;
  else if (sign_test(del1, del2) < 0) { dmax = 3.0 * del2; if (fabs(d[n - 1]) > fabs(dmax)) d[n - 1] = dmax; }
}
int wo_eos_set_curve_table(wo_eos *e, int which, int interp, int n, const double *xy) {
  if (which < 0 || which > 2 || n < 1 || n > WO_MAX_CURVE_POINTS || interp < 0 || interp > 2) return 1;
  wo_curve_table *t = &e->tab[which];
  t->n = n; t->interp = interp;
  for (int i = 0; i < n; i++) { t->x[i] = xy[2 * i]; t->v[i] = xy[2 * i + 1]; t->d[i] = 0.0; }
  if (interp == 2) pchip_deriv(n, t->x, t->v, t->d);
  return 0;
}
double wo_curve_table_value(const wo_curve_table *t, double x) {
  if (x <= t->x[0]) return t->v[0];
  if (x >= t->x[t->n - 1]) return t->v[t->n - 1];
  int i = 0;
  while (i + 1 < t->n - 1 && x >= t->x[i + 1]) i++;
  double x0 = t->x[i], x1 = t->x[i + 1], v0 = t->v[i], v1 = t->v[i + 1];
  if (t->interp == 1) return v0;
  if (t->interp == 2) {
    double h = x1 - x0, delta = (v1 - v0) / h;
    double del1 = (t->d[i] - delta) / h, del2 = (t->d[i + 1] - delta) / h;
    double c2 = -(2.0 * del1 + del2), c3 = (del1 + del2) / h, dx = x - x0;
    return v0 + dx * (t->d[i] + dx * (c2 + dx * c3));
  }
  double xi = (x - x0) / (x1 - x0);
  return (1.0 - xi) * v0 + xi * v1;
}
static void eos_relperm(const wo_eos *e, double sl, double rp[2]) {
  if (e->rp_type == WO_RP_TABLE) {   /* relative_permeability_table_values :547-558 */
    rp[0] = wo_curve_table_value(&e->tab[0], sl);
    rp[1] = wo_curve_table_value(&e->tab[1], 1.0 - sl);
  } else wo_relperm(e->rp_type, e->rp_par, sl, rp);
}
static double eos_capillary(const wo_eos *e, double sl, double t) {
  if (e->cp_type == WO_CP_TABLE) return wo_curve_table_value(&e->tab[2], sl);   /* capillary_pressure.F90:349-358 */
  return wo_capillary(e->cp_type, e->cp_par, sl, t);
}

void wo_relperm(int type, const double *par, double sl, double rp[2]) {
  switch (type) {
  case WO_RP_FULLY_MOBILE:
    rp[0] = 1.0; rp[1] = 1.0; break;
  case WO_RP_LINEAR:
    rp[0] = lin2(sl, par[0], par[1], 0.0, 1.0);
    rp[1] = lin2(1.0 - sl, par[2], par[3], 0.0, 1.0);
    break;
  case WO_RP_PICKENS:
    rp[0] = pow(sl, par[0]); rp[1] = 1.0; break;
  case WO_RP_COREY:
  case WO_RP_GRANT: {
    double slr = par[0], ssr = par[1], sv = 1.0 - sl;
    if (sv < ssr) { rp[0] = 1.0; rp[1] = 0.0; }
    else if (sv > 1.0 - slr) { rp[0] = 0.0; rp[1] = 1.0; }
    else {
      double ss = (sl - slr) / (1.0 - slr - ssr), ss2 = ss * ss;
      rp[0] = ss2 * ss2;
      if (type == WO_RP_COREY) rp[1] = (1.0 - 2.0 * ss + ss2) * (1.0 - ss2);
      else rp[1] = 1.0 - rp[0];
    }
  } break;
  case WO_RP_VAN_GENUCHTEN: {
    double lambda = par[0], slr = par[1], sls = par[2];
    int sum_unity = (par[3] != 0.0);
    double ssr = par[4];
    double ss = (sl - slr) / (sls - slr);
    if (ss < 0.0) rp[0] = 0.0;
    else if (ss < 1.0) {
      double w = 1.0 - pow(1.0 - pow(ss, 1.0 / lambda), lambda);
      rp[0] = sqrt(ss) * w * w;
    } else rp[0] = 1.0;
    if (sum_unity) rp[1] = 1.0 - rp[0];
    else {
      double sh = (sl - slr) / (1.0 - slr - ssr), sh2 = sh * sh;
      rp[1] = (1.0 - 2.0 * sh + sh2) * (1.0 - sh2);
      if (rp[1] > 1.0) rp[1] = 1.0;
    }
  } break;
  default:
    rp[0] = rp[1] = 0.0;
  }
}

/* src/capillary_pressure.F90:159-305.  par: linear [s0,s1,pressure];
 * van Genuchten [P0,lambda,slr,sls,Pmax,apply_Pmax] */
double wo_capillary(int type, const double *par, double sl, double t) {
  (void)t;
  switch (type) {
  case WO_CP_ZERO: return 0.0;
  case WO_CP_LINEAR: return lin2(sl, par[0], par[1], -fabs(par[2]), 0.0);
  case WO_CP_VAN_GENUCHTEN: {
    const double eps = 1.e-3;
    double P0 = fabs(par[0]), lambda = par[1], slr = par[2], sls = par[3];
    double Pmax = fabs(par[4]);
    int apply_Pmax = (par[5] != 0.0);
    double cp;
    if (sl < 1.0) {
      double ss = (sl - slr) / (sls - slr);
      if (ss < 0.0) cp = -Pmax;
      else if (ss < 1.0) cp = -P0 * pow(pow(ss, -1.0 / lambda) - 1.0, 1.0 - lambda);
      else cp = 0.0;
      if (cp > 0.0) cp = 0.0;
      if (apply_Pmax && cp < -Pmax) cp = -Pmax;
      if (sl > 1.0 - eps) cp = cp * (1.0 - sl) / eps;
    } else cp = 0.0;
    return cp;
  }
  }
  return 0.0;
}

/* Brent's method (Press et al., Numerical Recipes, zbrent), as the reference uses it:
 * src/root_finder.F90:127-248.  Returns 0 ok, 1 not bracketed, 2 iterations exceeded. */
int wo_brent(wo_rootfn f, void *ctx, double a, double b, double xtol, double ftol, int maxit,
             double *root, int *iters) {
  const double small = 1.e-16;
  double fa = f(a, ctx), fb = f(b, ctx);
  *root = 0.0;
  if (iters) *iters = 0;
  if (fa * fb > 0.0) return 1;
  double c = b, fc = fb, d = 0.0, e = 0.0;
  int found = 0, iter;
  for (iter = 1; iter <= maxit; iter++) {
    if (fb * fc > 0.0) { c = a; fc = fa; d = b - a; e = d; }
    if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    double dx = 0.5 * (c - b);
    if (fabs(dx) <= xtol || fabs(fb) <= ftol) { found = 1; break; }
    if (fabs(e) >= xtol && fabs(fa) > fabs(fb)) {
      double s = fb / fa, p, q;
      if (fabs(a - c) <= small) { p = 2.0 * dx * s; q = 1.0 - s; }
      else {
        q = fa / fc;
        double r = fb / fc;
        p = s * (2.0 * dx * q * (q - r) - (b - a) * (r - 1.0));
        q = (q - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q = -q; else p = -p;
      double pc1 = 3.0 * dx * q - fabs(xtol * q), pc2 = fabs(e * q);
      double pc = pc1 < pc2 ? pc1 : pc2;
      if (2.0 * p < pc) { e = d; d = p / q; }
      else { d = dx; e = d; }
    } else { d = dx; e = d; }
    a = b; fa = fb;
    if (fabs(d) > xtol) b += d;
    else b += (dx >= 0.0 ? fabs(xtol) : -fabs(xtol));
    fb = f(b, ctx);
  }
  *root = b;
  if (iters) *iters = iter;
  return found ? 0 : 2;
}

/* ---------------------------------------------------------------------------------------- */
/* fluid record offsets (src/fluid.F90:212-267) */
#define F_P 0
#define F_T 1
#define F_REGION 2
#define F_OLD_REGION 3
#define F_PHASES 4
#define F_PERMFAC 5
#define F_PP 6
#define PH_RHO 0
#define PH_MU 1
#define PH_SAT 2
#define PH_KR 3
#define PH_PC 4
#define PH_H 5
#define PH_U 6
#define PH_X 7
static inline int phase_off(const wo_eos *e, int p) { return (7 + e->nc - 1) + p * (8 + e->nc - 1); }

/* eos_wse: mixture region -> water region / halite presence (src/eos_wse.F90:131-134) */
static const int WSE_WATER_REGION[9] = {0, 1, 2, 0, 4, 1, 2, 0, 4};
static const int WSE_HALITE[9] = {0, 0, 0, 0, 0, 1, 1, 0, 1};
static int wse_bulk_properties(const wo_eos *e, const double *primary, double *fl);
static int wse_phase_properties(const wo_eos *e, const double *primary, double *fl);
static int wse_transition(const wo_eos *e, const double *oldp, double *prim, const double *old_fluid,
                          double *fluid, int *transition);

void wo_eos_init(wo_eos *e, int kind) {
  memset(e, 0, sizeof(*e));
  e->kind = kind;
  e->nc = 1;
  e->temperature = 20.0;
  {   /* default tables: k_r = S on [0, 1] (relative_permeability.F90:513-516), P_c = 0 (capillary_pressure.F90:324-325) */
    const double kr[4] = {0.0, 0.0, 1.0, 1.0}, pc[4] = {0.0, 0.0, 1.0, 0.0};
    wo_eos_set_curve_table(e, 0, 0, 2, kr);
    wo_eos_set_curve_table(e, 1, 0, 2, kr);
    wo_eos_set_curve_table(e, 2, 0, 2, pc);
  }
  if (kind == WO_EOS_W) { /* src/eos_w.F90:50-99 */
    e->np = 1; e->nph = 1; e->nmob = 1; e->isothermal = 1;
    e->scale[1][0] = 1.e6; e->scale[2][0] = 1.e6;
  } else {                /* src/eos_we.F90:56-126; src/eos_wge.F90:44-128 + eos_wce.F90:24-47 */
    e->np = 2; e->nph = 2; e->nmob = 2; e->isothermal = 0;
    e->scale[1][0] = 1.e6; e->scale[1][1] = 1.e2;
    e->scale[2][0] = 1.e6; e->scale[2][1] = 1.e2;
    e->scale[4][0] = 1.e6; e->scale[4][1] = 1.0;
    if (kind == WO_EOS_WCE || kind == WO_EOS_WAE) { e->np = 3; e->nc = 2; } /* scale[.][2] = 0: adaptive Pg/P */
    if (kind == WO_EOS_WSE || kind == WO_EOS_WSCE || kind == WO_EOS_WSAE) {
      /* src/eos_wse.F90:123-165, eos_wsge.F90:85-140: solid third phase, regions 5, 6, 8 = 1, 2, 4 + halite;
       * with a gas: components water, salt, gas; 4th primary scale 0 = adaptive Pg / P */
      e->np = 3; e->nc = 2; e->nph = 3;
      if (kind != WO_EOS_WSE) { e->np = 4; e->nc = 3; }
      for (int r = 1; r <= 8; r++) {
        if (r == 3 || r == 7) continue;
        e->scale[r][0] = 1.e6;
        e->scale[r][1] = (r == 4 || r == 8) ? 1.0 : 1.e2;
        e->scale[r][2] = 1.0;
      }
    }
  }
  e->df = (7 + e->nc - 1) + e->nph * (8 + e->nc - 1);
  /* reference defaults: linear [0,1]/[0,1] rel perm, zero Pc
   * (relative_permeability.F90:225-226,591; capillary_pressure.F90:389) */
  e->rp_type = WO_RP_LINEAR;
  e->rp_par[0] = 0.0; e->rp_par[1] = 1.0; e->rp_par[2] = 0.0; e->rp_par[3] = 1.0;
  e->cp_type = WO_CP_ZERO;
}

/* src/eos.F90:186-210 */
void wo_eos_unscale(const wo_eos *e, const double *y, int region, double *primary) {
  for (int k = 0; k < e->np; k++) primary[k] = y[k] * e->scale[region][k];
  if (IS_WGE(e) && e->scale[region][2] == 0.0) primary[2] = y[2] * primary[0]; /* eos_wge.F90:659-674 */
  if (IS_WSGE(e) && e->scale[region][3] == 0.0) primary[3] = y[3] * primary[0];   /* eos_wsge.F90:984-998 */
}
void wo_eos_scale(const wo_eos *e, const double *primary, int region, double *y) {
  for (int k = 0; k < e->np; k++) y[k] = primary[k] / e->scale[region][k];
  if (IS_WGE(e) && e->scale[region][2] == 0.0) y[2] = primary[2] / primary[0];   /* eos_wge.F90:639-655 */
  if (IS_WSGE(e) && e->scale[region][3] == 0.0) y[3] = primary[3] / primary[0];  /* eos_wsge.F90:962-980 */
}

/* src/eos_we.F90:327-390 (we), src/eos_w.F90:126-147 (w); phase composition eos.F90:214-236 */
int wo_eos_bulk_properties(const wo_eos *e, const double *primary, double *fl) {
  int region = (int)lround(fl[F_REGION]);
  int err = 0;
  fl[F_P] = primary[0];
  if (e->kind == WO_EOS_W) {
    fl[F_T] = e->temperature;
    fl[phase_off(e, 0) + PH_SAT] = 1.0;
    int ph = th_phase_composition(e, region, fl[F_P], fl[F_T]);
    if (ph > 0) fl[F_PHASES] = (double)ph; else err = 1;
    fl[F_PERMFAC] = 1.0;
    fl[F_PP] = fl[F_P];
    return err;
  }
  if (IS_SALT(e)) return wse_bulk_properties(e, primary, fl);
  if (IS_WGE(e)) { /* src/eos_wge.F90:350-389 */
    fl[F_PP] = fl[F_P] - primary[2];
    fl[F_PP + 1] = primary[2];
  }
  if (region == 4) {
    double t;
    err = th_sat_temperature(e, IS_WGE(e) ? fl[F_PP] : fl[F_P], &t);
    if (err == 0) fl[F_T] = t;
  } else fl[F_T] = primary[1];
  if (err) return err;
  fl[F_PERMFAC] = 1.0;
  int ph = th_phase_composition(e, region, fl[F_P], fl[F_T]);
  if (ph <= 0) return 1;
  fl[F_PHASES] = (double)ph;
  double *l = fl + phase_off(e, 0), *v = fl + phase_off(e, 1);
  switch (region) {
  case 1: l[PH_SAT] = 1.0; v[PH_SAT] = 0.0; break;
  case 2: l[PH_SAT] = 0.0; v[PH_SAT] = 1.0; break;
  case 4: l[PH_SAT] = 1.0 - primary[1]; v[PH_SAT] = primary[1]; break;
  }
  if (!IS_WGE(e)) fl[F_PP] = fl[F_P];
  return 0;
}

/* src/eos_wge.F90:421-543 with CO2 as the gas (eos_wce.F90) */
static int wce_phase_properties(const wo_eos *e, double *fl) {
  double P = fl[F_P], T = fl[F_T], Pw = fl[F_PP], Pg = fl[F_PP + 1];
  int phases = (int)lround(fl[F_PHASES]);
  double sl = fl[phase_off(e, 0) + PH_SAT];
  double rp[2];
  eos_relperm(e, sl, rp);
  double gas_rho, gas_h;
  const int air = (e->kind == WO_EOS_WAE);
  const double gas_mw = air ? AIR_MW : CO2_MW;
  int err = air ? wo_air_properties(Pg, T, &gas_rho, &gas_h) : wo_co2_properties(Pg, T, &gas_rho, &gas_h);
  if (err) return err;
  for (int p = 0; p < e->nph; p++) {
    double *ph = fl + phase_off(e, p);
    if (phases & (1 << p)) {
      double water_pressure, cap, henry, esol;
      if (p == 0) {
        water_pressure = P;
        cap = eos_capillary(e, sl, T);
        henry = air ? wo_air_henrys_constant(T) : wo_co2_henrys_constant(T);
        esol = air ? wo_air_energy_solution(T) : wo_co2_energy_solution(T);
      } else {
        water_pressure = Pw; cap = 0.0; henry = 0.0; esol = 0.0;
      }
      double wrho, wu;
      err = th_props(e, p == 0 ? 1 : 2, water_pressure, T, &wrho, &wu);
      if (err) return err;
      double grho = (p == 0) ? 0.0 : gas_rho; /* effective_properties: no free gas in liquid */
      double xg;
      if (p == 0) xg = wo_ncg_mole_to_mass(Pg / henry, gas_mw);
      else {
        double tot = grho + wrho;
        xg = (tot < 1.e-30) ? 0.0 : grho / tot;
      }
      double wmu = th_viscosity(e, p == 0 ? 1 : 2, T, water_pressure, wrho), mu;
      if (p == 0) mu = wmu;
      else if (air) mu = wo_air_mixture_viscosity(wmu, T, xg);
      else {
        double gmu;
        err = wo_co2_viscosity(Pg, T, &gmu);
        if (err) return err;
        mu = wmu * (1.0 - xg) + gmu * xg;
      }
      ph[PH_MU] = mu;
      ph[PH_RHO] = wrho + grho;
      ph[PH_X] = 1.0 - xg; ph[PH_X + 1] = xg;
      ph[PH_KR] = rp[p]; ph[PH_PC] = cap;
      double wh = wu + water_pressure / wrho;
      ph[PH_H] = wh * (1.0 - xg) + (gas_h + esol) * xg;
      ph[PH_U] = ph[PH_H] - P / ph[PH_RHO];
    } else {
      ph[PH_RHO] = 0.0; ph[PH_U] = 0.0; ph[PH_H] = 0.0; ph[PH_KR] = 0.0;
      ph[PH_PC] = 0.0; ph[PH_MU] = 0.0; ph[PH_X] = 0.0; ph[PH_X + 1] = 0.0;
    }
  }
  return 0;
}

/* src/eos_we.F90:394-458 (we), src/eos_w.F90:168-213 (w) */
int wo_eos_phase_properties(const wo_eos *e, const double *primary, double *fl) {
  (void)primary;
  double P = fl[F_P], T = fl[F_T];
  if (e->kind == WO_EOS_W) {
    int p = (int)lround(fl[F_REGION]); /* region 1 -> liquid, the only phase */
    double *ph = fl + phase_off(e, 0);
    double rho, u;
    int err = th_props(e, p == 1 ? 1 : 2, P, T, &rho, &u);
    if (err) return err;
    ph[PH_RHO] = rho; ph[PH_U] = u; ph[PH_H] = u + P / rho;
    ph[PH_KR] = 1.0; ph[PH_PC] = 0.0; ph[PH_X] = 1.0;
    ph[PH_MU] = th_viscosity(e, p == 1 ? 1 : 2, T, P, rho);
    return 0;
  }
  if (IS_WGE(e)) return wce_phase_properties(e, fl);
  if (IS_SALT(e)) return wse_phase_properties(e, primary, fl);
  int phases = (int)lround(fl[F_PHASES]);
  double sl = fl[phase_off(e, 0) + PH_SAT];
  double rp[2], cp[2];
  eos_relperm(e, sl, rp);
  cp[0] = eos_capillary(e, sl, T);
  cp[1] = 0.0;
  for (int p = 0; p < e->nph; p++) {
    double *ph = fl + phase_off(e, p);
    if (phases & (1 << p)) {
      double rho, u;
      int err = th_props(e, p == 0 ? 1 : 2, P, T, &rho, &u);
      if (err) return err;
      ph[PH_RHO] = rho; ph[PH_U] = u; ph[PH_H] = u + P / rho;
      ph[PH_X] = 1.0; ph[PH_KR] = rp[p]; ph[PH_PC] = cp[p];
      ph[PH_MU] = th_viscosity(e, p == 0 ? 1 : 2, T, P, rho);
    } else {
      ph[PH_RHO] = 0.0; ph[PH_U] = 0.0; ph[PH_H] = 0.0; ph[PH_KR] = 0.0;
      ph[PH_PC] = 0.0; ph[PH_MU] = 0.0; ph[PH_X] = 0.0;
    }
  }
  return 0;
}

/* saturation-line difference along the old->new primary segment: src/eos_we.F90:530-553 */
typedef struct { double p0, t0, p1, t1, g0, g1; const wo_eos *e; } satline_ctx; /* g: gas partial pressure (eos_wge.F90:678-701) */
static double satline_diff(double x, void *vc) {
  satline_ctx *c = (satline_ctx *)vc;
  double P = (1.0 - x) * c->p0 + x * c->p1, T = (1.0 - x) * c->t0 + x * c->t1, Ps = 0.0;
  double Pg = (1.0 - x) * c->g0 + x * c->g1;
  th_sat_pressure(c->e, T, &Ps); /* error ignored, as the reference does */
  return P - Pg - Ps;
}

/* src/eos_we.F90:149-323 and src/eos_wge.F90:154-346 (wce: water pressure = P - Pg on the
 * saturation line, Pg clipped to [0, P] and interpolated); eos_w has no transitions */
static double lerp_clamped(double xi, double a, double b) { /* interpolate(xi), interpolation.F90:494-543 */
  if (xi <= 0.0) return a;
  if (xi >= 1.0) return b;
  return (1.0 - xi) * a + xi * b;
}
int wo_eos_transition(const wo_eos *e, const double *oldp, double *prim, const double *old_fluid,
                      double *fluid, int *transition) {
  *transition = 0;
  if (e->kind == WO_EOS_W) return 0;
  if (IS_SALT(e)) return wse_transition(e, oldp, prim, old_fluid, fluid, transition);
  const double small = 1.e-6;
  const int wce = IS_WGE(e);
  int old_region = (int)lround(old_fluid[F_REGION]);
  int err = 0;
  if (old_region == 4) {
    double sv = prim[1];
    int new_region = 0;
    if (sv < 0.0) new_region = 1; else if (sv > 1.0) new_region = 2;
    if (!new_region) return 0;
    double bound = (new_region == 1) ? 0.0 : 1.0;
    double pfac = (new_region == 1) ? 1.0 + small : 1.0 - small;
    if (wce) prim[2] = fmax(0.0, fmin(prim[2], prim[0]));
    /* linear inverse interpolant on component 2: src/interpolation.F90:407-435 */
    double v1 = oldp[1], v2 = prim[1];
    double vmax = fmax(fabs(v1), fabs(v2));
    if (fabs(v2 - v1) >= 1.e-8 * vmax) {
      double vs1 = v1 / vmax, vs2 = v2 / vmax, ys = bound / vmax;
      double xi = (ys - vs1) / (vs2 - vs1);
      xi = (1.0 - xi) * 0.0 + xi * 1.0;
      double ip = lerp_clamped(xi, oldp[0], prim[0]);
      double ig = wce ? lerp_clamped(xi, oldp[2], prim[2]) : 0.0;
      double iw = ip - ig; /* interpolated water pressure */
      prim[0] = pfac * iw + ig;
      if (wce) prim[2] = ig;
      double t;
      err = th_sat_temperature(e, iw, &t);
      if (err == 0) { prim[1] = t; fluid[F_REGION] = (double)new_region; *transition = 1; }
    } else {
      double ps;
      err = th_sat_pressure(e, old_fluid[F_T], &ps);
      if (err == 0) {
        prim[0] = pfac * ps + (wce ? prim[2] : 0.0);
        prim[1] = old_fluid[F_T];
        fluid[F_REGION] = (double)new_region;
        *transition = 1;
      }
    }
    return err;
  }
  double ps;
  err = th_sat_pressure(e, prim[1], &ps);
  if (err) return err;
  double pw = prim[0] - (wce ? prim[2] : 0.0);
  if ((old_region == 1 && pw < ps) || (old_region == 2 && pw > ps)) {
    if (wce) prim[2] = fmax(0.0, fmin(prim[2], prim[0]));
    satline_ctx c = {oldp[0], oldp[1], prim[0], prim[1], wce ? oldp[2] : 0.0, wce ? prim[2] : 0.0, e};
    double root;
    int it;
    int rerr = wo_brent(satline_diff, &c, 0.0, 1.0, 1.e-8, 1.e-8, 100, &root, &it);
    if (rerr == 0) {
      double ig = wce ? lerp_clamped(root, oldp[2], prim[2]) : 0.0;
      prim[0] = lerp_clamped(root, oldp[0], prim[0]);
      if (wce) prim[2] = ig;
    } else prim[0] = ps + (wce ? prim[2] : 0.0);
    prim[1] = (old_region == 1) ? small : 1.0 - small;
    fluid[F_REGION] = 4.0;
    *transition = 1;
  }
  return 0;
}

/* src/eos_we.F90:486-526, src/eos_w.F90:232-255, src/eos_wge.F90:573-635 */
int wo_eos_check_primary(const wo_eos *e, const double *fluid, double *prim, int *changed) {
  *changed = 0;
  if (IS_WSGE(e)) { /* src/eos_wsge.F90:890-958 */
    const double small = 1.e-6;
    if (!(prim[0] > 0.0)) return 1;
    double maxpp = (1.0 - small) * prim[0];
    if (prim[3] > maxpp) { prim[3] = maxpp; *changed = 1; }
    else if (prim[3] < 0.0) { prim[3] = 0.0; *changed = 1; }
    if (prim[2] < 0.0) { prim[2] = 0.0; *changed = 1; }
    else if (prim[2] > 1.0) return 1;
    if (prim[0] - prim[3] > 100.e6) return 1;
    if (WSE_WATER_REGION[(int)lround(fluid[F_REGION])] == 4) {
      if (prim[1] < -1.0 || prim[1] > 2.0) return 1;
    } else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
    return 0;
  }
  if (e->kind == WO_EOS_WSE) { /* src/eos_wse.F90:891-938 */
    if (prim[2] < 0.0) { prim[2] = 0.0; *changed = 1; }
    else if (prim[2] > 1.0) return 1;
    if (prim[0] < 0.0 || prim[0] > 100.e6) return 1;
    if (WSE_WATER_REGION[(int)lround(fluid[F_REGION])] == 4) {
      if (prim[1] < -1.0 || prim[1] > 2.0) return 1;
    } else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
    return 0;
  }
  if (IS_WGE(e)) {
    const double small = 1.e-6;
    if (!(prim[0] > 0.0)) return 1;
    double maxpp = (1.0 - small) * prim[0];
    if (prim[2] > maxpp) { prim[2] = maxpp; *changed = 1; }
    else if (prim[2] < 0.0) { prim[2] = 0.0; *changed = 1; }
    if (prim[0] - prim[2] > 100.e6) return 1;
  } else {
    double p = prim[0];
    if (p < 0.0 || p > 100.e6) return 1;
    if (e->kind == WO_EOS_W) return 0;
  }
  int region = (int)lround(fluid[F_REGION]);
  if (region == 4) {
    if (prim[1] < -1.0 || prim[1] > 2.0) return 1;
  } else if (prim[1] < 0.0 || prim[1] > 800.0) return 1;
  return 0;
}

/* src/cell.F90:114-142 with fluid.F90:295-315, :354-370 and rock.F90:142-150 */
void wo_cell_balance(const wo_eos *e, const double *fl, const double *rock, double *bal) {
  double phi = rock[5];
  for (int c = 0; c < e->nc; c++) bal[c] = 0.0;
  double ef = 0.0;
  for (int p = 0; p < e->nph; p++) {
    const double *ph = fl + phase_off(e, p);
    double ds = ph[PH_RHO] * ph[PH_SAT];
    for (int c = 0; c < e->nc; c++) bal[c] += ds * ph[PH_X + c];
    ef += ds * ph[PH_U];
  }
  for (int c = 0; c < e->nc; c++) bal[c] = phi * bal[c];
  if (!e->isothermal) {
    double er = rock[6] * rock[7] * fl[F_T];
    bal[e->np - 1] = phi * ef + (1.0 - phi) * er;
  }
}

/* src/face.F90:358-377 */
double wo_harmonic_average(const double *fg, double x1, double x2) {
  double wx = (fg[1] * x2 + fg[2] * x1) / fg[3];
  return (fabs(wx) > 1.e-30) ? x1 * x2 / wx : 0.0;
}

/* src/eos.F90:240-257 */
double wo_conductivity(const double *rock, const double *fl, const wo_eos *e) {
  double sl = fl[phase_off(e, 0) + PH_SAT];
  return rock[4] + sqrt(sl) * (rock[3] - rock[4]);
}

/* src/face.F90:334-354 */
double wo_face_phase_density(const wo_eos *e, const double *f1, const double *f2, int p) {
  const double *a = f1 + phase_off(e, p), *b = f2 + phase_off(e, p);
  double rho = a[PH_SAT] * a[PH_RHO] + b[PH_SAT] * b[PH_RHO];
  double w = a[PH_SAT] + b[PH_SAT];
  return rho / w;
}

/* src/face.F90:443-515 */
void wo_face_flux(const wo_eos *e, const double *fg, const double *f1, const double *r1,
                  const double *f2, const double *r2, double *flux) {
  int np = e->np, nc = e->nc;
  for (int i = 0; i < np + e->nmob; i++) flux[i] = 0.0;
  int dir = (int)lround(fg[11]);
  double k = wo_harmonic_average(fg, r1[dir - 1] * f1[F_PERMFAC], r2[dir - 1] * f2[F_PERMFAC]);
  if (!e->isothermal) {
    double cond = wo_harmonic_average(fg, wo_conductivity(r1, f1, e), wo_conductivity(r2, f2, e));
    double dtdn = (f2[F_T] - f1[F_T]) / fg[3];
    flux[np - 1] = -cond * dtdn;
  }
  int ph1 = (int)lround(f1[F_PHASES]), ph2 = (int)lround(f2[F_PHASES]);
  int present = ph1 | ph2;
  for (int p = 0; p < e->nmob; p++) {
    if (!(present & (1 << p))) continue;
    const double *a = f1 + phase_off(e, p), *b = f2 + phase_off(e, p);
    double rho_f = wo_face_phase_density(e, f1, f2, p);
    double dpdn = ((f2[F_P] + b[PH_PC]) - (f1[F_P] + a[PH_PC])) / fg[3];
    double G = dpdn - rho_f * fg[7];
    int up_is_1 = (G <= 0.0);
    int phup = up_is_1 ? ph1 : ph2;
    if (!(phup & (1 << p))) continue;
    const double *u = up_is_1 ? a : b;
    double mob = u[PH_KR] * u[PH_RHO] / u[PH_MU];
    double F = -k * mob * G;
    double sum = 0.0;
    for (int c = 0; c < nc; c++) {
      double fc = F * u[PH_X + c];
      flux[c] += fc;
      sum += fc;
    }
    if (!e->isothermal) flux[np - 1] += u[PH_H] * F;
    flux[np + p] = sum;
  }
}

/* separator_stage_init: src/separator.F90:108-136 -- enthalpies of saturated water and steam at
 * the separator pressure */
int wo_separator_enthalpies(const wo_eos *e, double pressure, double *hf, double *hg) {
  double ts, rho, u;
  int err = th_sat_temperature(e, pressure, &ts);
  if (err) return err;
  err = th_props(e, 1, pressure, ts, &rho, &u);
  if (err) return err;
  *hf = u + pressure / rho;
  err = th_props(e, 2, pressure, ts, &rho, &u);
  if (err) return err;
  *hg = u + pressure / rho;
  return 0;
}

/* ==== salt (NaCl) thermodynamics: src/salt_thermodynamics.F90 ================================ */
/* Correlation data as held at src/salt_thermodynamics.F90:12-30 (Driesner 2007; Battistelli 2012;
 * Haas 1976; Phillips et al. 1981). */
#define SALT_MW 58.443
static const double HALITE_DENSITY[3] = {2.1704e3, -2.4599e-1, -9.5797e-5};
static const double HALITE_ENTHALPY[4] = {-5.615174e5, 8.766380e2, 6.413881e-2, 8.810112e-5};
static const double HALITE_SOLUBILITY[7] = {0.2627980, 3.130833e-2, 2.136495, -9.371763, 3.083588e1,
                                            -3.959050e1, 1.711302e1};
static const double HALITE_SOLUBILITY_2PH[5] = {0.2876823, 0.30122157, -0.39877656, 0.31352381, -0.09062578};
static const double BRINE_PSAT_A[4] = {0.0, 5.93582e-1, -5.19386, 1.23156};
static const double BRINE_PSAT_B[6] = {0.0, 1.15420, 1.41254, -1.92476, -1.70717, 1.05390};
static const double BRINE_VISC[4] = {1.0, 0.0816, 0.0122, 1.28e-4};

/* polynomial_single, src/utils.F90:224-241 (Horner) */
static double poly(const double *a, int n, double x) {
  double p = a[n - 1];
  for (int i = n - 2; i >= 0; i--) p = a[i] + x * p;
  return p;
}

/* newton1d_general, src/utils.F90:651-709: FD Newton with relative increment of the start value */
typedef double (*newton_fn)(double x, void *ctx, int *err);
static int newton1d(newton_fn f, void *ctx, double *x, double ftol, double xtol, int maxit, double inc) {
  double delx = inc * (*x);
  int err = 0, found = 0;
  for (int i = 0; i < maxit; i++) {
    double fx = f(*x, ctx, &err);
    if (err) break;
    if (fabs(fx) <= ftol) { found = 1; break; }
    double fxd = f(*x + delx, ctx, &err);
    if (err) break;
    double df = (fxd - fx) / delx, dx = -fx / df;
    *x += dx;
    if (fabs(dx) <= xtol) { found = 1; break; }
  }
  if (!err && !found) err = 1;
  return err;
}

/* halite_solubility :44-61 */
int wo_halite_solubility(double t, double *s) {
  if (0.0 <= t) { *s = poly(HALITE_SOLUBILITY, 7, t * 1.0e-3); return 0; }
  *s = 0.0;
  return 1;
}
/* halite_properties :108-134 -> density, internal energy */
int wo_halite_properties(double p, double t, double *rho, double *u) {
  const double l3 = 5.727e-3, l4 = 2.715e-3, l5 = 733.4;
  double pbar = p / 1.0e5;
  double density0 = poly(HALITE_DENSITY, 3, t);
  double l = l3 + l4 * exp(t / l5);
  *rho = density0 + l * pbar;
  double h1 = poly(HALITE_ENTHALPY, 4, t);
  double h = h1 + 44.14 * (pbar - 1.0);
  *u = h - p / *rho;
  return 0;
}
/* salt_mole_fraction :140-148 (molality-like measure used by the Haas / Phillips fits) */
static double salt_mole_fraction(double xs) { return 1.0e3 * xs / (SALT_MW * (1.0 - xs)); }

/* brine_saturation_pressure :152-176 */
int wo_brine_sat_pressure(const wo_eos *e, double t, double xs, double *ps) {
  double smol = salt_mole_fraction(xs);
  double a = 1.0 + 1.0e-5 * poly(BRINE_PSAT_A, 4, smol);
  double b = 1.0e-5 * poly(BRINE_PSAT_B, 6, 0.1 * smol);
  double tk = t + TC_K;
  double teff = exp(log(tk) / (a + b * tk)) - TC_K;
  return th_sat_pressure(e, teff, ps);
}
struct bst_ctx { const wo_eos *e; double p, xs; };
static double bst_f(double x, void *c, int *err) {
  struct bst_ctx *k = (struct bst_ctx *)c;
  double ps = 0.0;
  *err = wo_brine_sat_pressure(k->e, x, k->xs, &ps);
  return k->p - ps;
}
/* brine_saturation_temperature :180-217 */
int wo_brine_sat_temperature(const wo_eos *e, double p, double xs, double *ts) {
  double t;
  int err = th_sat_temperature(e, p, &t);
  if (err) return err;
  struct bst_ctx k = {e, p, xs};
  err = newton1d(bst_f, &k, &t, 1.0e-10 * p, 1.0e-10, 100, 1.0e-8);
  *ts = t;
  return err;
}
struct hs2_ctx { const wo_eos *e; double p; };
static double hs2_f(double x, void *c, int *err) {
  struct hs2_ctx *k = (struct hs2_ctx *)c;
  double t, s;
  *err = wo_brine_sat_temperature(k->e, k->p, x, &t);
  if (*err) return -1.0;
  *err = wo_halite_solubility(t, &s);
  return x - s;
}
/* halite_solubility_two_phase :65-104 */
int wo_halite_solubility_two_phase(const wo_eos *e, double p, double *s) {
  double xs = poly(HALITE_SOLUBILITY_2PH, 5, p / 1.0e7);
  struct hs2_ctx k = {e, p};
  int err = newton1d(hs2_f, &k, &xs, 1.0e-10, 1.0e-10, 100, 1.0e-8);
  *s = xs;
  return err;
}

/* brine_properties :221-389 (Driesner 2007): density and internal energy */
int wo_brine_properties(const wo_eos *e, double p, double t, double xs, double *rho_out, double *u_out) {
  double pbar = p / 1.0e5;
  double f = 1.0 / (xs + (1.0 - xs) * SALT_MW / WATER_MW);
  double xmol = xs * f, xmol1 = 1.0 - xmol, xmol12 = xmol1 * xmol1;
  double bmw = SALT_MW * f;
  int err = 0;
  /* density */
  double n11 = -54.2958 - 45.7623 * exp(-9.44785e-4 * pbar);
  double n21 = -2.6142 - 0.000239092 * pbar;
  const double c22[3] = {0.0356828, 4.37235e-3, 2.0566e-3};
  double n22 = poly(c22, 3, pbar / 1.0e3);
  double c1[4] = {330.47 + 0.942876 * sqrt(pbar), 8.17193, -2.47556e-4, 3.45052e-4};
  double n1x1 = poly(c1, 4, pbar / 1.0e2);
  double c2[4] = {-0.0370751 + 0.00237723 * sqrt(pbar), 5.42049e-1, 5.84709e-1, -5.99373e-1};
  double n2x1 = poly(c2, 4, pbar / 1.0e4);
  double n10 = n1x1, n20 = 1.0 - n21 * sqrt(n22), n12 = -n11 - n10;
  double n23 = n2x1 - n20 - n21 * sqrt(1.0 + n22);
  double n1 = n10 + n11 * xmol1 + n12 * xmol12;
  double n2 = n20 + n21 * sqrt(xmol + n22) + n23 * xmol;
  /* deviation, eq. 14 */
  double pp = pbar + 472.051;
  double n300 = 7.60664e6 / (pp * pp);
  double n301 = -50.0 - 86.1446 * exp(-6.21128e-4 * pbar);
  double n302 = 294.318 * exp(-5.66735e-3 * pbar);
  double n310 = -0.0732761 * exp(-2.3772e-3 * pbar) - 5.2948e-5 * pbar;
  double n311 = -47.2747 + 24.3653 * exp(-1.25533e-3 * pbar);
  double n312 = -0.278529 - 0.00081381 * pbar;
  double n30 = n300 * (exp(n301 * xmol) - 1.0) + n302 * xmol;
  double n31 = n310 * exp(n311 * xmol) + n312 * xmol;
  double tstar_v = n1 + n2 * t + n30 * exp(n31 * t);
  double pcrit = e->thermo == 1 ? 22.12e6 : PCRITICAL; /* src/IFC67.F90:159, IAPWS.F90:275 */
  double ts = 0.0, rho = 0.0, rw, uw;
  int extrapolate = 0;
  if (p <= pcrit) {
    err = th_sat_temperature(e, p, &ts);
    if (!err) extrapolate = tstar_v > ts;
  }
  if (err) return err;
  if (extrapolate) { /* eq. 17 */
    const double dt = 0.2;
    err = th_props(e, 1, p, ts, &rw, &uw);
    if (err) return err;
    double vws = 1.0e3 * WATER_MW / rw;
    err = th_props(e, 1, p, ts - dt, &rw, &uw);
    if (err) return err;
    double vws1 = 1.0e3 * WATER_MW / rw;
    double dvdt = (vws - vws1) / dt, logp = log(pbar);
    double co[3] = {2.0125e-7 + 3.29977e-9 * exp(-4.31279 * logp), -1.17748e-7, 7.58009e-8};
    double o2 = poly(co, 3, logp), ts2 = ts * ts;
    double o1 = dvdt - 3.0 * o2 * ts2;
    double o0 = vws - ts * (o1 + o2 * ts2);
    double cv[4] = {o0, o1, 0.0, o2};
    double vb = poly(cv, 4, tstar_v);
    rho = 1.0e3 * bmw / vb;
  } else {
    err = th_props(e, 1, p, tstar_v, &rw, &uw);
    if (err) return err;
    rho = rw * bmw / WATER_MW;
  }
  /* internal energy */
  double q11 = -32.1724 + 0.0621255 * pbar;
  const double cq21[3] = {-1.69513, -4.52781, -6.04279};
  double q21 = poly(cq21, 3, pbar / 1.0e4);
  double q22 = 0.0612567 + 1.88082e-5 * pbar;
  const double cq1[3] = {47.9048, -9.36994, 6.51059};
  double q1x1 = poly(cq1, 3, pbar / 1.0e3);
  const double cq2[3] = {0.241022, 3.45087e-1, -4.28356e-1};
  double q2x1 = poly(cq2, 3, pbar / 1.0e4);
  double q10 = q1x1, q20 = 1.0 - q21 * sqrt(q22), q12 = -q11 - q10;
  double q23 = q2x1 - q20 - q21 * sqrt(1.0 + q22);
  double q1 = q10 + q11 * xmol1 + q12 * xmol12;
  double q2 = q20 + q21 * sqrt(xmol + q22) + q23 * xmol;
  double tstar_h = q1 + q2 * t;
  err = th_props(e, 1, p, tstar_h, &rw, &uw);
  if (err) return err;
  double hb = uw + p / rw;
  *rho_out = rho;
  *u_out = hb - p / rho;
  return 0;
}

/* brine_viscosity :393-423 */
int wo_brine_viscosity(const wo_eos *e, double t, double p, double xs, double *mu) {
  double smol = salt_mole_fraction(xs);
  double factor = poly(BRINE_VISC, 4, smol) + 6.29e-4 * t * (1.0 - exp(-0.7 * smol));
  double rw, uw;
  int err = th_props(e, 1, p, t, &rw, &uw);
  if (err) return err;
  *mu = factor * th_viscosity(e, 1, t, p, rw);
  return 0;
}

/* Henry's constant and energy of solution of the gas in brine of salt mass fraction xs
 * (henrys_constant_salt / henrys_derivative_salt + ncg_energy_solution_salt:
 * src/ncg_co2_thermodynamics.F90:139-232, src/ncg_air_thermodynamics.F90:141-238,
 * src/ncg_thermodynamics.F90:187-261) */
static const double CO2_HENRY_SALT[5] = {1.19784e-1, -7.17823e-2, 4.93854e-2, -1.03826e-2, 1.08233e-3};
static const double AIR_HENRY_SALT[2][5] = {{0.183369, -0.236905, 0.242438, -7.30134e-2, 8.58723e-3},
                                            {0.16218, -1.16909e-1, 5.55185e-2, -8.75443e-3, 9.91567e-4}};
void wo_gas_henry_salt(const wo_eos *e, double t, double xs, double *henry, double *esol);
#define gas_henry_salt wo_gas_henry_salt
void wo_gas_henry_salt(const wo_eos *e, double t, double xs, double *henry, double *esol) {
  double m = salt_mole_fraction(xs), x = t / 100.0, tk = t + TC_K, deriv;
  if (GAS_IS_AIR(e)) {
    double hc[2], h = 0.0;
    air_henry_constituents(t, hc);
    deriv = 0.0;
    for (int i = 0; i < 2; i++) {
      double dc[6], ds[4];
      for (int k = 1; k < 7; k++) dc[k - 1] = k * AIR_HENRY[i][k];
      for (int k = 1; k < 5; k++) ds[k - 1] = k * AIR_HENRY_SALT[i][k];
      double kb = horner(AIR_HENRY_SALT[i], 5, x);
      h += AIR_WEIGHT[i] * hc[i] * pow(10.0, m * kb);
      double d0 = AIR_HENRY_P0[i] * (1.e5 * horner(dc, 6, x)) / (hc[i] * 100.0);
      deriv += AIR_WEIGHT[i] * (d0 + log(10.0) * m * (horner(ds, 4, x) / 100.0));
    }
    *henry = h;
    *esol = -1.e3 * GAS_CONSTANT * tk * tk * deriv / AIR_MW;
  } else {
    double h0 = wo_co2_henrys_constant(t), d[5], ds[4];
    for (int i = 0; i < 5; i++) d[i] = (i + 1) * CO2_HENRY[i + 1];
    for (int k = 1; k < 5; k++) ds[k - 1] = k * CO2_HENRY_SALT[k];
    *henry = h0 * pow(10.0, m * horner(CO2_HENRY_SALT, 5, x));
    deriv = 1.e8 * horner(d, 5, x) / (h0 * 100.0) + log(10.0) * m * (horner(ds, 4, x) / 100.0);
    *esol = -1.e3 * GAS_CONSTANT * tk * tk * deriv / CO2_MW;
  }
}

/* ==== eos_wse: water, salt, energy (src/eos_wse.F90) ======================================== */
/* fluid_permeability_factor_{null,power,verma_pruess}_modify: src/fluid.F90:588-664 */
double wo_permeability_factor(const wo_eos *e, double pf) {
  if (e->perm_type == 1) return pow(pf, e->perm_par[0]);
  if (e->perm_type == 2) {
    double n = e->perm_par[0], phir = e->perm_par[1], gamma = e->perm_par[2];
    double omega = 1.0 + (1.0 / (gamma * (1.0 / phir - 1.0)));
    double theta = (pf - phir) / (1.0 - phir);
    return pow(theta, n) * (1.0 - gamma + gamma / pow(omega, n)) /
           (1.0 - gamma + gamma * pow(theta / (theta + omega - 1.0), n));
  }
  return 1.0;
}

/* bulk properties :645-688, phase saturations :692-726 */
static int wse_bulk_properties(const wo_eos *e, const double *primary, double *fl) {
  int region = (int)lround(fl[F_REGION]);
  int wr = WSE_WATER_REGION[region], halite = WSE_HALITE[region];
  int err = 0;
  fl[F_P] = primary[0];
  const double pg = IS_WSGE(e) ? primary[3] : 0.0, pw = fl[F_P] - pg;   /* brine pressure (eos_wsge.F90:645-650) */
  if (wr == 4) {
    double xs = primary[2], t;
    if (region != 4) err = wo_halite_solubility_two_phase(e, pw, &xs);
    if (!err) err = wo_brine_sat_temperature(e, pw, xs, &t);
    if (!err) fl[F_T] = t;
  } else fl[F_T] = primary[1];
  if (err) return err;
  int ph = th_phase_composition(e, wr, fl[F_P], fl[F_T]);
  if (ph <= 0) return 1;
  fl[F_PHASES] = (double)ph;
  double ss = (halite || region == 2) ? primary[2] : 0.0, fs = 1.0 - ss;
  double *l = fl + phase_off(e, 0), *v = fl + phase_off(e, 1), *h = fl + phase_off(e, 2);
  switch (wr) {
  case 1: l[PH_SAT] = fs; v[PH_SAT] = 0.0; break;
  case 2: l[PH_SAT] = 0.0; v[PH_SAT] = fs; break;
  case 4: l[PH_SAT] = fs - primary[1]; v[PH_SAT] = primary[1]; break;
  }
  h[PH_SAT] = ss;
  fl[F_PERMFAC] = wo_permeability_factor(e, l[PH_SAT] + v[PH_SAT]);
  fl[F_PP] = pw;
  fl[F_PP + 1] = 0.0;
  if (IS_WSGE(e)) fl[F_PP + 2] = pg;
  return 0;
}

static void phase_zero(const wo_eos *e, double *ph) {
  ph[PH_RHO] = 0.0; ph[PH_U] = 0.0; ph[PH_H] = 0.0; ph[PH_KR] = 0.0; ph[PH_PC] = 0.0; ph[PH_MU] = 0.0;
  for (int c = 0; c < e->nc; c++) ph[PH_X + c] = 0.0;
}

/* phase properties :730-857 */
static int wse_phase_properties(const wo_eos *e, const double *primary, double *fl) {
  double P = fl[F_P], T = fl[F_T];
  int phases = (int)lround(fl[F_PHASES]), region = (int)lround(fl[F_REGION]);
  int halite = WSE_HALITE[region];
  double xs = 0.0;
  int err = 0;
  if (halite) err = wo_halite_solubility(T, &xs);
  else if (region == 2) xs = 0.0;
  else xs = primary[2];
  if (err) return err;
  double sl = fl[phase_off(e, 0) + PH_SAT], ss = fl[phase_off(e, 2) + PH_SAT];
  double sle = sl / (1.0 - ss);
  double rp[2], cp[2];
  eos_relperm(e, sle, rp);
  cp[0] = eos_capillary(e, sle, T);
  cp[1] = 0.0;
  const int gas = IS_WSGE(e), nc = e->nc;
  const double pw = fl[F_PP], pg = gas ? fl[F_PP + 2] : 0.0;   /* brine pressure, gas partial pressure */
  double gas_rho = 0.0, gas_h = 0.0;
  if (gas) {   /* eos_wsge_phase_properties: src/eos_wsge.F90:675-852 */
    err = GAS_IS_AIR(e) ? wo_air_properties(pg, T, &gas_rho, &gas_h) : wo_co2_properties(pg, T, &gas_rho, &gas_h);
    if (err) return err;
  }
  for (int p = 0; p < e->nmob; p++) {
    double *ph = fl + phase_off(e, p);
    if (phases & (1 << p)) {
      double rho, u, xp, bp, mu;
      if (p == 0) { bp = P; err = wo_brine_properties(e, bp, T, xs, &rho, &u); xp = xs; }
      else { bp = gas ? pw : P; err = th_props(e, 2, bp, T, &rho, &u); xp = 0.0; }
      if (err) return err;
      if (p == 0) { err = wo_brine_viscosity(e, T, P, xs, &mu); if (err) return err; }
      else mu = th_viscosity(e, 2, T, P, rho);
      double xg = 0.0, grho = 0.0, esol = 0.0;
      if (gas) {
        if (p == 0) {
          double henry;
          gas_henry_salt(e, T, xs, &henry, &esol);
          xg = wo_ncg_mole_to_mass(pg / henry, GAS_IS_AIR(e) ? AIR_MW : CO2_MW);
        } else {
          grho = gas_rho;
          double tot = grho + rho;
          xg = (tot < 1.e-30) ? 0.0 : grho / tot;
          if (GAS_IS_AIR(e)) mu = wo_air_mixture_viscosity(mu, T, xg);
          else {
            double gmu;
            err = wo_co2_viscosity(pg, T, &gmu);
            if (err) return err;
            mu = mu * (1.0 - xg) + gmu * xg;
          }
        }
      }
      ph[PH_MU] = mu;
      ph[PH_RHO] = rho + grho;
      ph[PH_X] = 1.0 - xp - xg; ph[PH_X + 1] = xp;
      if (gas) ph[PH_X + 2] = xg;
      ph[PH_KR] = rp[p]; ph[PH_PC] = cp[p];
      double bh = u + bp / rho;
      ph[PH_H] = gas ? bh * (1.0 - xg) + (gas_h + esol) * xg : bh;
      ph[PH_U] = gas ? ph[PH_H] - P / ph[PH_RHO] : u;
    } else phase_zero(e, ph);
  }
  (void)nc;
  double *h = fl + phase_off(e, 2);
  if (halite || region == 2) {
    double rho, u;
    err = wo_halite_properties(P, T, &rho, &u);
    if (err) return err;
    phase_zero(e, h);
    h[PH_RHO] = rho; h[PH_U] = u; h[PH_H] = u + P / rho;
    h[PH_X] = 0.0; h[PH_X + 1] = 1.0;
  } else phase_zero(e, h);
  return 0;
}

/* eos_wse_saturation_difference :942-974 along the old -> new primary segment */
typedef struct { const double *a, *b; int halite, wr; const wo_eos *e; } wse_line;
static double wse_satline_diff(double x, void *vc) {
  wse_line *c = (wse_line *)vc;
  double P = (1.0 - x) * c->a[0] + x * c->b[0], T = (1.0 - x) * c->a[1] + x * c->b[1];
  double xs = (1.0 - x) * c->a[2] + x * c->b[2], Ps = 0.0;
  double Pg = IS_WSGE(c->e) ? (1.0 - x) * c->a[3] + x * c->b[3] : 0.0;   /* eos_wsge.F90:1002-1036 */
  if (c->wr == 1) {
    if (c->halite) wo_halite_solubility(T, &xs);
    wo_brine_sat_pressure(c->e, T, xs, &Ps);
  } else th_sat_pressure(c->e, T, &Ps);
  return P - Pg - Ps;
}

/* transition_to_single_phase :203-337 */
static int wse_to_single_phase(const wo_eos *e, const double *oldp, const double *old_fluid, int new_region,
                               double *prim, double *fluid, int *transition) {
  const double small = 1.e-6;
  int old_region = (int)lround(old_fluid[F_REGION]);
  int old_halite = WSE_HALITE[old_region], nwr = WSE_WATER_REGION[new_region];
  double ss = old_halite ? prim[2] : 0.0;
  double bound = (nwr == 1) ? 0.0 : 1.0 - ss;
  double pfac = (nwr == 1) ? 1.0 + small : 1.0 - small;
  int err = 0;
  const int gas = IS_WSGE(e);
  if (gas) { /* eos_wsge.F90:222-226: salt and gas variables clipped before interpolating */
    prim[2] = fmax(0.0, prim[2]);
    prim[3] = fmax(0.0, fmin(prim[3], prim[0]));
  }
  /* inverse linear interpolant on component 2 (src/interpolation.F90:407-435, :571-584) */
  double v1 = oldp[1], v2 = prim[1], vmax = fmax(fabs(v1), fabs(v2));
  if (fabs(v2 - v1) >= 1.e-8 * vmax) {
    double xi = (bound / vmax - v1 / vmax) / (v2 / vmax - v1 / vmax);
    double ip = lerp_clamped(xi, oldp[0], prim[0]), is = lerp_clamped(xi, oldp[2], prim[2]);
    double ig = gas ? lerp_clamped(xi, oldp[3], prim[3]) : 0.0, ibp = ip - ig;   /* interpolated brine pressure */
    double t, xs;
    prim[0] = pfac * ibp + ig;
    prim[2] = gas ? is : fmax(0.0, is);
    if (gas) prim[3] = ig;
    if (nwr == 1) {
      if (old_halite) err = wo_halite_solubility_two_phase(e, ibp, &xs);
      else xs = prim[2];
      if (!err) err = wo_brine_sat_temperature(e, ibp, xs, &t);
    } else err = th_sat_temperature(e, ibp, &t);
    if (!err) { prim[1] = t; fluid[F_REGION] = (double)new_region; *transition = 1; }
  } else {
    double xs, ps;
    if (nwr == 1) {
      if (old_halite) err = wo_halite_solubility(old_fluid[F_T], &xs);
      else xs = oldp[2];
      if (!err) { xs = fmax(0.0, xs); err = wo_brine_sat_pressure(e, old_fluid[F_T], xs, &ps); }
    } else err = th_sat_pressure(e, old_fluid[F_T], &ps);
    if (!err) {
      prim[0] = pfac * ps + (gas ? prim[3] : 0.0);
      prim[1] = old_fluid[F_T];
      fluid[F_REGION] = (double)new_region;
      *transition = 1;
    }
  }
  return err;
}

/* halite_transition :413-525 (eos_wsge.F90:416-532: brine pressure = P - Pg) */
static int wse_halite_transition(const wo_eos *e, const double *old_fluid, double *prim, double *fluid,
                                 int *transition, int err) {
  const double small = 1.e-6;
  int region = (int)lround(fluid[F_REGION]);
  double t, sol;
  switch (region) {
  case 1: case 4:
    if (region == 1) t = prim[1];
    else err = wo_brine_sat_temperature(e, prim[0] - (IS_WSGE(e) ? prim[3] : 0.0), prim[2], &t);
    if (!err) {
      err = wo_halite_solubility(t, &sol);
      if (prim[2] > sol) { prim[2] = small; fluid[F_REGION] = region + 4; *transition = 1; }
    }
    break;
  case 2:
    if (prim[2] > 0.0) { prim[2] = small; fluid[F_REGION] = 6.0; *transition = 1; }
    break;
  case 5: case 8:
    if (prim[2] < 0.0) {
      if (region == 5) {
        err = wo_halite_solubility(prim[1], &sol);
        if (!err) { prim[2] = sol - small; fluid[F_REGION] = 1.0; *transition = 1; }
      } else {
        int cur_old = (int)lround(fluid[F_OLD_REGION]), last_old = (int)lround(old_fluid[F_OLD_REGION]);
        if (cur_old == 6 || last_old == 6) { prim[2] = small; fluid[F_REGION] = 4.0; *transition = 1; }
        else {
          err = wo_halite_solubility_two_phase(e, prim[0] - (IS_WSGE(e) ? prim[3] : 0.0), &sol);
          if (!err) { prim[2] = sol - small; fluid[F_REGION] = 4.0; *transition = 1; }
        }
      }
    }
    break;
  case 6:
    if (prim[2] < 0.0) { prim[2] = 0.0; fluid[F_REGION] = 2.0; *transition = 1; }
    break;
  }
  return err;
}

/* eos_wse_transition :529-617 with transition_to_two_phase :341-409 */
static int wse_transition(const wo_eos *e, const double *oldp, double *prim, const double *old_fluid,
                          double *fluid, int *transition) {
  const double small = 1.e-6;
  int old_region = (int)lround(old_fluid[F_REGION]);
  int owr = WSE_WATER_REGION[old_region], old_halite = WSE_HALITE[old_region];
  int err = 0;
  if (owr == 4) {
    int off = old_halite ? 4 : 0;
    double sv = prim[1];
    if (sv < 0.0) err = wse_to_single_phase(e, oldp, old_fluid, off + 1, prim, fluid, transition);
    else {
      double ss = old_halite ? prim[2] : 0.0;
      if (sv > 1.0 - ss) err = wse_to_single_phase(e, oldp, old_fluid, off + 2, prim, fluid, transition);
    }
  } else {
    double xs, ps;
    if (owr == 1) {
      if (old_halite) err = wo_halite_solubility(prim[1], &xs);
      else xs = prim[2];
      if (!err) { xs = fmax(0.0, xs); err = wo_brine_sat_pressure(e, prim[1], xs, &ps); }
    } else err = th_sat_pressure(e, prim[1], &ps);
    const int gas = IS_WSGE(e);
    double pwat = prim[0] - (gas ? prim[3] : 0.0);
    if (!err && ((owr == 1 && pwat < ps) || (owr == 2 && pwat > ps))) {
      int new_region = old_halite ? 8 : 4;
      prim[2] = fmax(0.0, prim[2]);
      if (gas) prim[3] = fmax(0.0, fmin(prim[3], prim[0]));
      wse_line c = {oldp, prim, old_halite, owr, e};
      double root;
      int it;
      if (wo_brent(wse_satline_diff, &c, 0.0, 1.0, 1.e-8, 1.e-8, 100, &root, &it) == 0) {
        double ip = lerp_clamped(root, oldp[0], prim[0]), is = lerp_clamped(root, oldp[2], prim[2]);
        double ig = gas ? lerp_clamped(root, oldp[3], prim[3]) : 0.0;
        prim[0] = ip;
        prim[2] = is;
        if (gas) prim[3] = ig;
      } else prim[0] = ps + (gas ? prim[3] : 0.0);
      double ss = old_halite ? prim[2] : 0.0;
      prim[1] = (owr == 1) ? small : 1.0 - ss - small;
      fluid[F_REGION] = (double)new_region;
      *transition = 1;
    }
  }
  if (!err) err = wse_halite_transition(e, old_fluid, prim, fluid, transition, err);
  return err;
}

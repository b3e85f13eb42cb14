// Salt (NaCl) thermodynamics on the device: halite solubility and properties, brine saturation
// line, brine density / internal energy (Driesner 2007) and viscosity (Phillips et al. 1981) --
// src/salt_thermodynamics.F90 of the reference, with the water side through the th:: dispatch.
// Correlation data as held at src/salt_thermodynamics.F90:12-30.  The two 1-D Newton solves
// (newton1d_general, src/utils.F90:651-709: finite-difference slope with an increment relative to
// the *starting* value) are nested exactly as in the reference: the halite solubility on the
// saturation line calls the brine saturation temperature inside its residual.
#pragma once

namespace wai {
namespace salt {

constexpr double SALT_MW = 58.443, WATER_MW = 18.01528, TC_K = 273.15;

template <int N>
__device__ __forceinline__ double poly(const double (&a)[N], double x) {   // Horner, utils.F90:224-241
  double p = a[N - 1];
#pragma unroll
  for (int i = N - 2; i >= 0; i--) p = a[i] + x * p;
  return p;
}

// halite_solubility :44-61
__device__ inline int halite_solubility(double t, double& s) {
  const double c[7] = {0.2627980, 3.130833e-2, 2.136495, -9.371763, 3.083588e1, -3.959050e1, 1.711302e1};
  if (0.0 <= t) { s = poly(c, t * 1.0e-3); return 0; }
  s = 0.0;
  return 1;
}

// halite_properties :108-134
__device__ inline void halite_properties(double p, double t, double& rho, double& u) {
  const double cd[3] = {2.1704e3, -2.4599e-1, -9.5797e-5};
  const double ch[4] = {-5.615174e5, 8.766380e2, 6.413881e-2, 8.810112e-5};
  const double l3 = 5.727e-3, l4 = 2.715e-3, l5 = 733.4;
  const double pbar = p / 1.0e5;
  rho = poly(cd, t) + (l3 + l4 * exp(t / l5)) * pbar;
  const double h = poly(ch, t) + 44.14 * (pbar - 1.0);
  u = h - p / rho;
}

__device__ __forceinline__ double mole_fraction(double xs) { return 1.0e3 * xs / (SALT_MW * (1.0 - xs)); }

// brine_saturation_pressure :152-176 (Haas 1976)
__device__ inline int brine_sat_pressure(int thermo, double t, double xs, double& ps) {
  const double ca[4] = {0.0, 5.93582e-1, -5.19386, 1.23156};
  const double cb[6] = {0.0, 1.15420, 1.41254, -1.92476, -1.70717, 1.05390};
  const double smol = mole_fraction(xs);
  const double a = 1.0 + 1.0e-5 * poly(ca, smol);
  const double b = 1.0e-5 * poly(cb, 0.1 * smol);
  const double tk = t + TC_K;
  const double teff = exp(log(tk) / (a + b * tk)) - TC_K;
  return th::sat_pressure(thermo, teff, ps);
}

// brine_saturation_temperature :180-217
__device__ inline int brine_sat_temperature(int thermo, double p, double xs, double& ts) {
  double x;
  int err = th::sat_temperature(thermo, p, x);
  if (err) return err;
  const double ftol = 1.0e-10 * p, xtol = 1.0e-10, delx = 1.0e-8 * x;
  bool found = false;
  for (int i = 0; i < 100; i++) {
    double ps;
    err = brine_sat_pressure(thermo, x, xs, ps);
    if (err) break;
    const double fx = p - ps;
    if (fabs(fx) <= ftol) { found = true; break; }
    err = brine_sat_pressure(thermo, x + delx, xs, ps);
    if (err) break;
    const double df = ((p - ps) - fx) / delx, dx = -fx / df;
    x += dx;
    if (fabs(dx) <= xtol) { found = true; break; }
  }
  if (!err && !found) err = 1;
  ts = x;
  return err;
}

// halite_solubility_two_phase :65-104
__device__ inline int halite_solubility_two_phase(int thermo, double p, double& s) {
  const double c0[5] = {0.2876823, 0.30122157, -0.39877656, 0.31352381, -0.09062578};
  double x = poly(c0, p / 1.0e7);
  const double delx = 1.0e-8 * x;
  int err = 0;
  bool found = false;
  auto f = [&](double xx, int& e) {
    double t, sol;
    e = brine_sat_temperature(thermo, p, xx, t);
    if (e) return -1.0;
    e = halite_solubility(t, sol);
    return xx - sol;
  };
  for (int i = 0; i < 100; i++) {
    const double fx = f(x, err);
    if (err) break;
    if (fabs(fx) <= 1.0e-10) { found = true; break; }
    const double fxd = f(x + delx, err);
    if (err) break;
    const double df = (fxd - fx) / delx, dx = -fx / df;
    x += dx;
    if (fabs(dx) <= 1.0e-10) { found = true; break; }
  }
  if (!err && !found) err = 1;
  s = x;
  return err;
}

// brine_properties :221-389: density and internal energy
__device__ inline int brine_properties(int thermo, double p, double t, double xs, double& rho_out, double& u_out) {
  const double pbar = p / 1.0e5;
  const double f = 1.0 / (xs + (1.0 - xs) * SALT_MW / WATER_MW);
  const double xmol = xs * f, xmol1 = 1.0 - xmol, xmol12 = xmol1 * xmol1;
  const double bmw = SALT_MW * f;
  const double n11 = -54.2958 - 45.7623 * exp(-9.44785e-4 * pbar);
  const double n21 = -2.6142 - 0.000239092 * pbar;
  const double c22[3] = {0.0356828, 4.37235e-3, 2.0566e-3};
  const double n22 = poly(c22, pbar / 1.0e3);
  const double c1[4] = {330.47 + 0.942876 * sqrt(pbar), 8.17193, -2.47556e-4, 3.45052e-4};
  const double n1x1 = poly(c1, pbar / 1.0e2);
  const double c2[4] = {-0.0370751 + 0.00237723 * sqrt(pbar), 5.42049e-1, 5.84709e-1, -5.99373e-1};
  const double n2x1 = poly(c2, pbar / 1.0e4);
  const double n10 = n1x1, n20 = 1.0 - n21 * sqrt(n22), n12 = -n11 - n10;
  const double n23 = n2x1 - n20 - n21 * sqrt(1.0 + n22);
  const double n1 = n10 + n11 * xmol1 + n12 * xmol12;
  const double n2 = n20 + n21 * sqrt(xmol + n22) + n23 * xmol;
  const double pp = pbar + 472.051;                       // deviation, eq. 14
  const double n300 = 7.60664e6 / (pp * pp);
  const double n301 = -50.0 - 86.1446 * exp(-6.21128e-4 * pbar);
  const double n302 = 294.318 * exp(-5.66735e-3 * pbar);
  const double n310 = -0.0732761 * exp(-2.3772e-3 * pbar) - 5.2948e-5 * pbar;
  const double n311 = -47.2747 + 24.3653 * exp(-1.25533e-3 * pbar);
  const double n312 = -0.278529 - 0.00081381 * pbar;
  const double n30 = n300 * (exp(n301 * xmol) - 1.0) + n302 * xmol;
  const double n31 = n310 * exp(n311 * xmol) + n312 * xmol;
  const double tstar_v = n1 + n2 * t + n30 * exp(n31 * t);
  const double pcrit = thermo == THERMO_IFC67 ? 22.12e6 : 22.064e6;   // IFC67.F90:159, IAPWS.F90:275
  double ts = 0.0, rho, rw, uw;
  int err = 0;
  bool extrapolate = false;
  if (p <= pcrit) {
    err = th::sat_temperature(thermo, p, ts);
    if (!err) extrapolate = tstar_v > ts;
  }
  if (err) return err;
  if (extrapolate) {                                      // eq. 17
    const double dt = 0.2;
    err = th::props(thermo, 1, p, ts, rw, uw);
    if (err) return err;
    const double vws = 1.0e3 * WATER_MW / rw;
    err = th::props(thermo, 1, p, ts - dt, rw, uw);
    if (err) return err;
    const double vws1 = 1.0e3 * WATER_MW / rw;
    const double dvdt = (vws - vws1) / dt, logp = log(pbar);
    const double co[3] = {2.0125e-7 + 3.29977e-9 * exp(-4.31279 * logp), -1.17748e-7, 7.58009e-8};
    const double o2 = poly(co, logp), ts2 = ts * ts;
    const double o1 = dvdt - 3.0 * o2 * ts2;
    const double o0 = vws - ts * (o1 + o2 * ts2);
    const double cv[4] = {o0, o1, 0.0, o2};
    rho = 1.0e3 * bmw / poly(cv, tstar_v);
  } else {
    err = th::props(thermo, 1, p, tstar_v, rw, uw);
    if (err) return err;
    rho = rw * bmw / WATER_MW;
  }
  const double q11 = -32.1724 + 0.0621255 * pbar;
  const double cq21[3] = {-1.69513, -4.52781, -6.04279};
  const double q21 = poly(cq21, pbar / 1.0e4);
  const double q22 = 0.0612567 + 1.88082e-5 * pbar;
  const double cq1[3] = {47.9048, -9.36994, 6.51059};
  const double q1x1 = poly(cq1, pbar / 1.0e3);
  const double cq2[3] = {0.241022, 3.45087e-1, -4.28356e-1};
  const double q2x1 = poly(cq2, pbar / 1.0e4);
  const double q10 = q1x1, q20 = 1.0 - q21 * sqrt(q22), q12 = -q11 - q10;
  const double q23 = q2x1 - q20 - q21 * sqrt(1.0 + q22);
  const double q1 = q10 + q11 * xmol1 + q12 * xmol12;
  const double q2 = q20 + q21 * sqrt(xmol + q22) + q23 * xmol;
  err = th::props(thermo, 1, p, q1 + q2 * t, rw, uw);
  if (err) return err;
  rho_out = rho;
  u_out = (uw + p / rw) - p / rho;
  return 0;
}

// brine_viscosity :393-423
__device__ inline int brine_viscosity(int thermo, double t, double p, double xs, double& mu) {
  const double cv[4] = {1.0, 0.0816, 0.0122, 1.28e-4};
  const double smol = mole_fraction(xs);
  const double factor = poly(cv, smol) + 6.29e-4 * t * (1.0 - exp(-0.7 * smol));
  double rw, uw;
  const int err = th::props(thermo, 1, p, t, rw, uw);
  if (err) return err;
  mu = factor * th::viscosity(thermo, 1, t, p, rw);
  return 0;
}

}  // namespace salt
}  // namespace wai

/* IFC-67 industrial formulation (International Formulation Committee, Duesseldorf 1967) as the
 * reference uses it for "thermodynamics": "ifc67": sub-region 1 (liquid water), sub-region 2
 * (steam), the K-function saturation line and the TOUGH2-style viscosity fits.
 *
 * TEST INFRASTRUCTURE (oracle): restates the published formulation in its tabulated form
 * chi = d(zeta)/d(beta), eps = zeta - theta d(zeta)/d(theta); operating ranges, reduction
 * constants (Tc = 647.3 K, pc = 22.12 MPa, v = 0.00317 chi m3/kg, h = 70120.4 eps J/kg) and error
 * behaviour follow /root/reference/src/IFC67.F90:147-176 (constants), :265-374 (region 1),
 * :378-396 and :580-600 (viscosity), :425-576 (region 2), :606-676 (saturation line).
 * Pinned on the known-answer values of test/unit/src/IFC67_test.F90 (tests/golden). */
#include <math.h>

#include "wai_oracle.h"

#define TC_K 273.15
#define TCK67 647.3
#define PC67 22.12e6
#define VSCALE 0.00317
#define HSCALE 70120.4

/* sub-region 1 coefficients A0..A22, a1..a12 (IFC-67 table) */
static const double A[23] = {
    6.824687741e3, -5.422063673e2, -2.096666205e4, 3.941286787e4, -13.466555478e4, 29.707143084e4,
    -4.375647096e5, 42.954208335e4, -27.067012452e4, 9.926972482e4, -16.138168904e3, 7.982692717,
    -2.616571843e-2, 1.522411790e-3, 2.284279054e-2, 2.421647003e2, 1.269716088e-10,
    2.074838328e-7, 2.174020350e-8, 1.105710498e-9, 1.293441934e1, 1.308119072e-5,
    6.047626338e-14};
static const double SA[12] = {8.438375405e-1, 5.362162162e-4, 1.72, 7.342278489e-2, 4.975858870e-2,
                              6.537154300e-1, 1.150e-6, 1.51080e-5, 1.41880e-1, 7.002753165,
                              2.995284926e-4, 2.040e-1};

int wo_ifc67_region1(double p, double t, double max_temperature, double *rho, double *u) {
  if (!(t <= max_temperature && p <= 100.0e6)) return 1;
  double th[21];
  th[0] = 1.0;
  th[1] = (t + TC_K) / TCK67;
  for (int k = 2; k <= 20; k++) th[k] = th[k - 1] * th[1];
  double b1 = p / PC67, b2 = b1 * b1, b3 = b2 * b1, b4 = b3 * b1;
  /* Y, Z of the leading term */
  double Y = 1.0 - SA[0] * th[2] - SA[1] / th[6];
  double disc = SA[2] * Y * Y - 2.0 * SA[3] * th[1] + 2.0 * SA[4] * b1;
  if (!(disc >= 0.0)) return 1;
  double Z = Y + sqrt(disc);
  double Z517 = pow(Z, 5.0 / 17.0);
  double dY = -2.0 * SA[0] * th[1] + 6.0 * SA[1] / th[7];
  double c1 = SA[5] - th[1], c2 = c1 * c1, c4 = c2 * c2, c8 = c4 * c4, c10 = c8 * c2;
  double a19 = SA[6] + th[19];
  double a11 = SA[7] + th[11];
  double s10 = SA[9] + b1;
  /* reduced volume */
  double v = A[11] * SA[4] / Z517;
  v += A[12] + A[13] * th[1] + A[14] * th[2] + A[15] * c10 + A[16] / a19;
  v -= (A[17] + 2.0 * A[18] * b1 + 3.0 * A[19] * b2) / a11;
  v -= A[20] * th[18] * (SA[8] + th[2]) * (-3.0 / (s10 * s10 * s10 * s10) + SA[10]);
  v += 3.0 * A[21] * (SA[11] - th[1]) * b2 + 4.0 * A[22] / th[20] * b3;
  double V = v * VSCALE;
  /* reduced enthalpy */
  double poly = 0.0;
  for (int k = 10; k >= 3; k--) poly = poly * th[1] + A[k];  /* A3 + A4 th + ... + A10 th^7 */
  poly = poly * th[2] - A[1];                                 /* sum (nu-2) A_nu th^(nu-1), nu >= 3, minus A1 */
  double e = A[0] * th[1] - poly;
  e += A[11] * (Z * (17.0 * (Z / 29.0 - Y / 12.0) + 5.0 * th[1] * dY / 12.0) + SA[3] * th[1] -
                (SA[2] - 1.0) * th[1] * Y * dY) / Z517;
  e += b1 * (A[12] - A[14] * th[2] + A[15] * (9.0 * th[1] + SA[5]) * c8 * c1 +
             A[16] * (19.0 * th[19] + a19) / (a19 * a19));
  e -= (11.0 * th[11] + a11) / (a11 * a11) * (A[17] * b1 + A[18] * b2 + A[19] * b3);
  e += A[20] * th[18] * (17.0 * SA[8] + 19.0 * th[2]) * (1.0 / (s10 * s10 * s10) + SA[10] * b1);
  e += A[21] * SA[11] * b3 + 21.0 * A[22] / th[20] * b4;
  *rho = 1.0 / V;
  *u = e * HSCALE - p * V;
  return 0;
}

/* sub-region 2: B0nu; B_mu,nu with exponents z; b_mu,lambda with exponents x; B9nu; L-function */
static const double B0[6] = {16.83599274, 28.56067796, -54.38923329, 0.4330662834, -0.6547711697,
                             8.565182058e-2};
static const struct { int n; double B[3]; int z[3]; } SER[5] = {
    {2, {6.670375918e-2, 1.388983801, 0.0}, {13, 3, 0}},
    {3, {8.390104328e-2, 2.614670893e-2, -3.373439453e-2}, {18, 2, 1}},
    {2, {4.520918904e-1, 1.069036614e-1, 0.0}, {18, 10, 0}},
    {2, {-5.975336707e-1, -8.847535804e-2, 0.0}, {25, 14, 0}},
    {3, {5.958051609e-1, -5.159303373e-1, 2.075021122e-1}, {32, 28, 24}}};
static const struct { double B[2]; int z[2]; int nl; double b[2]; int x[2]; } RAT[3] = {
    {{1.190610271e-1, -9.867174132e-2}, {12, 11}, 1, {4.006073948e-1, 0.0}, {14, 0}},
    {{1.683998803e-1, -5.809438001e-2}, {24, 18}, 1, {8.636081627e-2, 0.0}, {19, 0}},
    {{6.552390126e-3, 5.710218649e-4}, {24, 14}, 2, {-8.532322921e-1, 3.460208861e-1}, {54, 27}}};
static const double B9[7] = {1.936587558e2, -1.388522425e3, 4.126607219e3, -6.508211677e3,
                             5.745984054e3, -2.693088365e3, 5.235718623e2};
static const double BL[3] = {15.74373327, -34.17061978, 19.31380707};
#define SB 7.633333333e-1
#define RI1 4.260321148

int wo_ifc67_region2(double p, double t, double *rho, double *u) {
  if (!(t <= 800.0 && p <= 100.0e6)) return 1;
  double theta = (t + TC_K) / TCK67, beta = p / PC67;
  double X[55];
  X[0] = 1.0;
  X[1] = exp(SB * (1.0 - theta));
  for (int k = 2; k < 55; k++) X[k] = X[k - 1] * X[1];
  double bt = SB * theta; /* b theta */
  double th2 = theta * theta, th3 = th2 * theta, th4 = th3 * theta;
  double chi = RI1 * theta / beta;
  double eps = B0[0] * theta - (-B0[1] + B0[3] * th2 + 2.0 * B0[4] * th3 + 3.0 * B0[5] * th4);
  double bp = 1.0; /* beta^(mu-1) */
  for (int m = 0; m < 5; m++) {
    double sv = 0.0, se = 0.0;
    for (int k = 0; k < SER[m].n; k++) {
      double term = SER[m].B[k] * X[SER[m].z[k]];
      sv += term;
      se += term * (1.0 + SER[m].z[k] * bt);
    }
    chi -= (m + 1) * bp * sv;
    bp *= beta;
    eps -= bp * se;
  }
  double binv = 1.0 / beta, bneg = binv * binv * binv * binv; /* beta^(2-mu), mu = 6 */
  for (int m = 0; m < 3; m++) {
    int mu = m + 6;
    double D = bneg, dsum = 0.0;
    for (int k = 0; k < RAT[m].nl; k++) {
      double term = RAT[m].b[k] * X[RAT[m].x[k]];
      D += term;
      dsum += RAT[m].x[k] * term;
    }
    double sv = 0.0, se = 0.0;
    for (int k = 0; k < 2; k++) {
      double term = RAT[m].B[k] * X[RAT[m].z[k]];
      sv += term;
      se += term * (1.0 + RAT[m].z[k] * bt - bt * dsum / D);
    }
    chi -= (mu - 2) * (bneg * binv) * sv / (D * D);  /* beta^(1-mu) */
    eps -= se / D;
    bneg *= binv;
  }
  double betaL = BL[0] + BL[1] * theta + BL[2] * th2, dbetaL = BL[1] + 2.0 * BL[2] * theta;
  double r = beta / betaL, r2 = r * r, r4 = r2 * r2, r10 = r4 * r4 * r2;
  double s9 = 0.0, e9 = 0.0, o2 = 1.0 + theta * 10.0 * dbetaL / betaL;
  for (int k = 6; k >= 0; k--) {
    s9 = s9 * X[1] + B9[k];
    e9 = e9 * X[1] + (o2 + k * bt) * B9[k];
  }
  chi += 11.0 * r10 * s9;
  eps += beta * r10 * e9;
  double V = chi * VSCALE;
  *rho = 1.0 / V;
  *u = eps * HSCALE - p * V;
  return 0;
}

/* K-function saturation line */
static const double KA[9] = {-7.691234564, -2.608023696e1, -1.681706546e2, 6.423285504e1,
                             -1.189646225e2, 4.167117320, 2.097506760e1, 1.0e9, 6.0};

int wo_ifc67_sat_pressure(double t, double *p) {
  if (!(t >= 1.0 && t <= TCK67 - TC_K)) return 1;
  double th = (t + TC_K) / TCK67, x = 1.0 - th, x2 = x * x;
  double s = 0.0;
  for (int k = 4; k >= 0; k--) s = (s + KA[k]) * x;
  *p = PC67 * exp(s / (th * (1.0 + KA[5] * x + KA[6] * x2)) - x / (KA[7] * x2 + KA[8]));
  return 0;
}

/* Newton iteration with a forward-difference slope, relative increment 1e-8 of the starting
 * estimate (newton1d, src/utils.F90:651-709): at most 200 iterations, |f| <= 1e-10 p or
 * |dx| <= 1e-10 */
int wo_ifc67_sat_temperature(double p, double *t) {
  if (!(p >= 0.0061e5 && p <= PC67)) return 1;
  double x = fmax(4606.0 / (24.02 - log(p)) - TC_K, 5.0);
  double delx = 1.0e-8 * x, ps;
  for (int i = 0; i < 200; i++) {
    if (wo_ifc67_sat_pressure(x, &ps)) return 1;
    double fx = p - ps;
    if (fabs(fx) <= 1.0e-10 * p) { *t = x; return 0; }
    if (wo_ifc67_sat_pressure(x + delx, &ps)) return 1;
    double dx = -fx / (((p - ps) - fx) / delx);
    x += dx;
    if (fabs(dx) <= 1.0e-10) { *t = x; return 0; }
  }
  return 1;
}

/* viscosity: liquid (pressure-corrected Andrade-type fit), steam (two fits either side of 350 degC) */
double wo_ifc67_viscosity(int region, double t, double p, double rho) {
  if (region == 1) {
    double ps = 0.0;
    wo_ifc67_sat_pressure(t, &ps); /* error ignored, as the reference does */
    double am = 1.0 + 1.0467 * (t - 31.85) * (p - ps) * 1.0e-11;
    return 1.0e-7 * am * 241.4 * pow(10.0, 247.8 / (t + 133.15));
  }
  double v1 = 0.407 * t + 80.4;
  if (t <= 350.0) return 1.0e-7 * (v1 - rho * (1858.0 - 5.9 * t) * 1.0e-3);
  return 1.0e-7 * (v1 + rho * (0.353 + rho * (676.5e-6 + rho * 102.1e-9)));
}

/* IFC67_phase_composition (:200-222): by region alone */
int wo_ifc67_phase_composition(int region) {
  return region == 1 ? 1 : region == 2 ? 2 : region == 4 ? 3 : 0;
}

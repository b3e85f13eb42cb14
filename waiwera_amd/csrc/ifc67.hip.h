// IFC-67 industrial formulation (International Formulation Committee, Duesseldorf 1967) for
// "thermodynamics": "ifc67" -- device side.  Sub-region 1 (liquid water), sub-region 2 (steam),
// K-function saturation line, TOUGH2-style viscosity fits.  Written from the tabulated form of
// the formulation (chi = d zeta / d beta, eps = zeta - theta d zeta / d theta); reduction
// constants, operating ranges and error behaviour are those of /root/reference/src/IFC67.F90:
// :147-176 constants, :265-374 region 1, :425-576 region 2, :378-396,580-600 viscosity,
// :606-676 saturation line (Newton iteration of src/utils.F90:651-709).
// Only the EOS sweeps (k_eos, k_eos_pert, k_transitions) reach this code; it is straight fp64
// VALU work on registers, a few hundred flops per phase.
#pragma once
#include <hip/hip_runtime.h>

namespace wai {
namespace ifc67 {

constexpr double TC_K = 273.15, TCK = 647.3, PC = 22.12e6, VSCALE = 0.00317, HSCALE = 70120.4;

// sub-region 1: A0..A22, a1..a12
__device__ constexpr double A[23] = {
    6.824687741e3, -5.422063673e2, -2.096666205e4, 3.941286787e4, -13.466555478e4, 29.707143084e4,
    -4.375647096e5, 42.954208335e4, -27.067012452e4, 9.926972482e4, -16.138168904e3, 7.982692717,
    -2.616571843e-2, 1.522411790e-3, 2.284279054e-2, 2.421647003e2, 1.269716088e-10,
    2.074838328e-7, 2.174020350e-8, 1.105710498e-9, 1.293441934e1, 1.308119072e-5,
    6.047626338e-14};
__device__ constexpr double SA[12] = {8.438375405e-1, 5.362162162e-4, 1.72, 7.342278489e-2,
                                      4.975858870e-2, 6.537154300e-1, 1.150e-6, 1.51080e-5,
                                      1.41880e-1, 7.002753165, 2.995284926e-4, 2.040e-1};

__device__ __forceinline__ int region1(double p, double t, double& rho, double& u) {
  if (!(t <= 350.0 && p <= 100.0e6)) return 1;
  const double th = (t + TC_K) / TCK;
  const double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th7 = th6 * th, th8 = th4 * th4;
  const double th10 = th8 * th2, th11 = th10 * th, th18 = th10 * th8, th19 = th18 * th, th20 = th10 * th10;
  const double b1 = p / PC, b2 = b1 * b1, b3 = b2 * b1, b4 = b3 * b1;
  const double Y = 1.0 - SA[0] * th2 - SA[1] / th6;
  const double disc = SA[2] * Y * Y - 2.0 * SA[3] * th + 2.0 * SA[4] * b1;
  if (!(disc >= 0.0)) return 1;
  const double Z = Y + sqrt(disc);
  const double Z517 = pow(Z, 5.0 / 17.0);
  const double dY = -2.0 * SA[0] * th + 6.0 * SA[1] / th7;
  const double c1 = SA[5] - th, c2 = c1 * c1, c4 = c2 * c2, c8 = c4 * c4, c10 = c8 * c2;
  const double a19 = SA[6] + th19, a11 = SA[7] + th11, s10 = SA[9] + b1;
  const double s10_2 = s10 * s10;
  double v = A[11] * SA[4] / Z517;
  v += A[12] + A[13] * th + A[14] * th2 + A[15] * c10 + A[16] / a19;
  v -= (A[17] + 2.0 * A[18] * b1 + 3.0 * A[19] * b2) / a11;
  v -= A[20] * th18 * (SA[8] + th2) * (-3.0 / (s10_2 * s10_2) + SA[10]);
  v += 3.0 * A[21] * (SA[11] - th) * b2 + 4.0 * A[22] / th20 * b3;
  const double V = v * VSCALE;
  double poly = 0.0;
#pragma unroll
  for (int k = 10; k >= 3; k--) poly = poly * th + A[k];
  poly = poly * th2 - A[1];
  double e = A[0] * th - poly;
  e += A[11] * (Z * (17.0 * (Z / 29.0 - Y / 12.0) + 5.0 * th * dY / 12.0) + SA[3] * th -
                (SA[2] - 1.0) * th * Y * dY) / Z517;
  e += b1 * (A[12] - A[14] * th2 + A[15] * (9.0 * th + SA[5]) * c8 * c1 +
             A[16] * (19.0 * th19 + a19) / (a19 * a19));
  e -= (11.0 * th11 + a11) / (a11 * a11) * (A[17] * b1 + A[18] * b2 + A[19] * b3);
  e += A[20] * th18 * (17.0 * SA[8] + 19.0 * th2) * (1.0 / (s10_2 * s10) + SA[10] * b1);
  e += A[21] * SA[11] * b3 + 21.0 * A[22] / th20 * b4;
  rho = 1.0 / V;
  u = e * HSCALE - p * V;
  return 0;
}

// sub-region 2 tables: series terms B_mu,nu X^z (mu = 1..5), rational terms (mu = 6..8) with
// denominators beta^(2-mu) + sum b X^x, the B9 polynomial and the L-function
struct Series { int n; double B[3]; int z[3]; };
struct Rational { double B[2]; int z[2]; int nl; double b[2]; int x[2]; };
__device__ constexpr double B0[6] = {16.83599274, 28.56067796, -54.38923329, 0.4330662834,
                                     -0.6547711697, 8.565182058e-2};
__device__ constexpr Series SER[5] = {
    {2, {6.670375918e-2, 1.388983801, 0.0}, {13, 3, 0}},
    {3, {8.390104328e-2, 2.614670893e-2, -3.373439453e-2}, {18, 2, 1}},
    {2, {4.520918904e-1, 1.069036614e-1, 0.0}, {18, 10, 0}},
    {2, {-5.975336707e-1, -8.847535804e-2, 0.0}, {25, 14, 0}},
    {3, {5.958051609e-1, -5.159303373e-1, 2.075021122e-1}, {32, 28, 24}}};
__device__ constexpr Rational RAT[3] = {
    {{1.190610271e-1, -9.867174132e-2}, {12, 11}, 1, {4.006073948e-1, 0.0}, {14, 0}},
    {{1.683998803e-1, -5.809438001e-2}, {24, 18}, 1, {8.636081627e-2, 0.0}, {19, 0}},
    {{6.552390126e-3, 5.710218649e-4}, {24, 14}, 2, {-8.532322921e-1, 3.460208861e-1}, {54, 27}}};
__device__ constexpr double B9[7] = {1.936587558e2, -1.388522425e3, 4.126607219e3, -6.508211677e3,
                                     5.745984054e3, -2.693088365e3, 5.235718623e2};
__device__ constexpr double BL[3] = {15.74373327, -34.17061978, 19.31380707};
constexpr double SB = 7.633333333e-1, RI1 = 4.260321148;

// x^n for a compile-time n by repeated squaring
template <int N>
__device__ __forceinline__ double xpow(double x) {
  if constexpr (N == 0) return 1.0;
  else if constexpr (N == 1) return x;
  else if constexpr (N % 2 == 0) { const double h = xpow<N / 2>(x); return h * h; }
  else return x * xpow<N - 1>(x);
}
__device__ __forceinline__ double xpow_rt(double x, int n) {  // small run-time exponent
  double r = 1.0, b = x;
  while (n) { if (n & 1) r *= b; b *= b; n >>= 1; }
  return r;
}

__device__ __forceinline__ int region2(double p, double t, double& rho, double& u) {
  if (!(t <= 800.0 && p <= 100.0e6)) return 1;
  const double theta = (t + TC_K) / TCK, beta = p / PC;
  const double X = exp(SB * (1.0 - theta));
  const double bt = SB * theta;
  const double th2 = theta * theta, th3 = th2 * theta, th4 = th3 * theta;
  double chi = RI1 * theta / beta;
  double eps = B0[0] * theta - (-B0[1] + B0[3] * th2 + 2.0 * B0[4] * th3 + 3.0 * B0[5] * th4);
  double bp = 1.0;
#pragma unroll
  for (int m = 0; m < 5; m++) {
    double sv = 0.0, se = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k < SER[m].n) {
        const double term = SER[m].B[k] * xpow_rt(X, SER[m].z[k]);
        sv += term;
        se += term * (1.0 + SER[m].z[k] * bt);
      }
    }
    chi -= (m + 1) * bp * sv;
    bp *= beta;
    eps -= bp * se;
  }
  const double binv = 1.0 / beta;
  double bneg = (binv * binv) * (binv * binv);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    double D = bneg, dsum = 0.0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (k < RAT[m].nl) {
        const double term = RAT[m].b[k] * xpow_rt(X, RAT[m].x[k]);
        D += term;
        dsum += RAT[m].x[k] * term;
      }
    }
    double sv = 0.0, se = 0.0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const double term = RAT[m].B[k] * xpow_rt(X, RAT[m].z[k]);
      sv += term;
      se += term * (1.0 + RAT[m].z[k] * bt - bt * dsum / D);
    }
    chi -= (m + 4) * (bneg * binv) * sv / (D * D);
    eps -= se / D;
    bneg *= binv;
  }
  const double betaL = BL[0] + BL[1] * theta + BL[2] * th2, dbetaL = BL[1] + 2.0 * BL[2] * theta;
  const double r = beta / betaL, r2 = r * r, r4 = r2 * r2, r10 = r4 * r4 * r2;
  const double o2 = 1.0 + theta * 10.0 * dbetaL / betaL;
  double s9 = 0.0, e9 = 0.0;
#pragma unroll
  for (int k = 6; k >= 0; k--) {
    s9 = s9 * X + B9[k];
    e9 = e9 * X + (o2 + k * bt) * B9[k];
  }
  chi += 11.0 * r10 * s9;
  eps += beta * r10 * e9;
  const double V = chi * VSCALE;
  rho = 1.0 / V;
  u = eps * HSCALE - p * V;
  return 0;
}

__device__ constexpr double KA[9] = {-7.691234564, -2.608023696e1, -1.681706546e2, 6.423285504e1,
                                     -1.189646225e2, 4.167117320, 2.097506760e1, 1.0e9, 6.0};

__device__ __forceinline__ int sat_pressure(double t, double& p) {
  if (!(t >= 1.0 && t <= TCK - TC_K)) return 1;
  const double th = (t + TC_K) / TCK, x = 1.0 - th, x2 = x * x;
  double s = 0.0;
#pragma unroll
  for (int k = 4; k >= 0; k--) s = (s + KA[k]) * x;
  p = PC * exp(s / (th * (1.0 + KA[5] * x + KA[6] * x2)) - x / (KA[7] * x2 + KA[8]));
  return 0;
}

__device__ __forceinline__ int sat_temperature(double p, double& t) {
  if (!(p >= 0.0061e5 && p <= PC)) return 1;
  double x = fmax(4606.0 / (24.02 - log(p)) - TC_K, 5.0);
  const double delx = 1.0e-8 * x;
  for (int i = 0; i < 200; i++) {
    double ps;
    if (sat_pressure(x, ps)) return 1;
    const double fx = p - ps;
    if (fabs(fx) <= 1.0e-10 * p) { t = x; return 0; }
    if (sat_pressure(x + delx, ps)) return 1;
    const double dx = -fx / (((p - ps) - fx) / delx);
    x += dx;
    if (fabs(dx) <= 1.0e-10) { t = x; return 0; }
  }
  return 1;
}

__device__ __forceinline__ double viscosity(int region, double t, double p, double rho) {
  if (region == 1) {
    double ps = 0.0;
    sat_pressure(t, ps);  // error ignored, as the reference does
    const double am = 1.0 + 1.0467 * (t - 31.85) * (p - ps) * 1.0e-11;
    return 1.0e-7 * am * 241.4 * pow(10.0, 247.8 / (t + 133.15));
  }
  const double v1 = 0.407 * t + 80.4;
  if (t <= 350.0) return 1.0e-7 * (v1 - rho * (1858.0 - 5.9 * t) * 1.0e-3);
  return 1.0e-7 * (v1 + rho * (0.353 + rho * (676.5e-6 + rho * 102.1e-9)));
}

__device__ __forceinline__ int phase_composition(int region) {
  return region == 1 ? 1 : region == 2 ? 2 : region == 4 ? 3 : 0;
}

}  // namespace ifc67
}  // namespace wai

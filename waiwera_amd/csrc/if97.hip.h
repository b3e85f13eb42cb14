// IAPWS-IF97 regions 1/2, saturation line and IAPWS-2008 viscosity as gfx950 device functions.
//
// Replaces the CPU evaluation in the reference: src/IAPWS.F90:503-542 (region 1), :596-639
// (region 2), :762-818 (saturation curve), :412-443 (viscosity), :317-365 (phase composition).
// The reference computes the integer powers through a run-time "powertable" addition chain
// (src/powertable.F90:261-278, pointer-chasing per cell).  Here every Gibbs sum is unrolled at
// compile time with constexpr exponents: powers are products in registers, common
// sub-products are shared by the compiler, no tables are read from memory at run time.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "if97_tables.hip.h"

namespace wai {
namespace if97 {

constexpr double TC_K = 273.15;          // src/thermodynamics.F90:37
constexpr double RCONST = 0.461526e3;    // src/thermodynamics.F90:36
constexpr double TCRITICALK = 647.096;   // src/IAPWS.F90:273
constexpr double TCRITICAL = TCRITICALK - TC_K;
constexpr double PCRITICAL = 22.064e6;
constexpr double DCRITICAL = 322.0;

// x^N for compile-time N >= 0 by binary exponentiation (inlined; shared squares are CSE'd)
template <int N>
__device__ __forceinline__ double powi_pos(double x) {
  if constexpr (N == 0) return 1.0;
  else if constexpr (N == 1) return x;
  else if constexpr (N % 2 == 0) { double h = powi_pos<N / 2>(x); return h * h; }
  else return x * powi_pos<N - 1>(x);
}
// x^N for any compile-time N given x and 1/x
template <int N>
__device__ __forceinline__ double powi(double x, double xinv) {
  if constexpr (N >= 0) return powi_pos<N>(x);
  else return powi_pos<-N>(xinv);
}

template <int... Is>
__device__ __forceinline__ void r1_sums(double a, double b, double binv, double& gampi,
                                        double& gamt, std::integer_sequence<int, Is...>) {
  gampi = ((((R1_N[Is] * R1_I[Is]) * powi<(R1_I[Is] > 0 ? R1_I[Is] - 1 : 0)>(a, 0.0)) *
            powi<R1_J[Is]>(b, binv)) + ...);
  gamt = ((((R1_N[Is] * R1_J[Is]) * powi<R1_I[Is]>(a, 0.0)) * powi<R1_J[Is] - 1>(b, binv)) + ...);
}

// Region 1 (liquid): density and internal energy from (P [Pa], T [deg C]); returns error flag
__device__ __forceinline__ int region1(double p, double t, double& rho, double& u) {
  if (!((t <= 350.0) && (p <= 100.e6))) return 1;
  constexpr double pstar = 16.53e6, tstar = 1386.0;
  const double tk = t + TC_K, rt = RCONST * tk, pi = p / pstar, tau = tstar / tk;
  const double a = 7.1 - pi, b = tau - 1.222, binv = 1.0 / b;
  double gampi, gamt;
  // terms with I = 0 contribute n*I = 0 to gampi exactly as in the reference (nI = n*I)
  r1_sums(a, b, binv, gampi, gamt, std::make_integer_sequence<int, 34>{});
  gampi = -gampi;
  rho = pstar / (rt * gampi);
  u = rt * (tau * gamt - pi * gampi);
  return 0;
}

template <int... Is>
__device__ __forceinline__ double r2_gamt0(double tau, double tinv,
                                           std::integer_sequence<int, Is...>) {
  return (((R2_N0[Is] * R2_J0[Is]) * powi<R2_J0[Is] - 1>(tau, tinv)) + ...);
}
template <int... Is>
__device__ __forceinline__ void r2_sums(double pi, double b, double binv, double& gampir,
                                        double& gamtr, std::integer_sequence<int, Is...>) {
  gampir = ((((R2_N[Is] * R2_I[Is]) * powi<R2_I[Is] - 1>(pi, 0.0)) * powi<R2_J[Is]>(b, binv)) + ...);
  gamtr = ((((R2_N[Is] * R2_J[Is]) * powi<R2_I[Is]>(pi, 0.0)) *
            powi<(R2_J[Is] > 0 ? R2_J[Is] - 1 : 0)>(b, binv)) + ...);
}

// Region 2 (steam)
__device__ __forceinline__ int region2(double p, double t, double& rho, double& u) {
  if (!((t <= 800.0) && (p <= 100.e6))) return 1;
  constexpr double pstar = 1.0e6, tstar = 540.0;
  const double tk = t + TC_K, rt = RCONST * tk, pi = p / pstar, tau = tstar / tk;
  const double b = tau - 0.5, tinv = 1.0 / tau;
  const double gamt0 = r2_gamt0(tau, tinv, std::make_integer_sequence<int, 9>{});
  double gampir, gamtr;
  r2_sums(pi, b, 0.0, gampir, gamtr, std::make_integer_sequence<int, 43>{});
  const double gampi = 1.0 / pi + gampir;
  rho = pstar / (rt * gampi);
  u = rt * (tau * (gamt0 + gamtr) - pi * gampi);
  return 0;
}

__device__ __forceinline__ int sat_pressure(double t, double& p) {
  if (!((t >= 0.0) && (t <= TCRITICAL))) return 1;
  const double tk = t + TC_K;
  const double theta = tk + SAT_N[8] / (tk - SAT_N[9]);
  const double theta2 = theta * theta;
  const double a = theta2 + SAT_N[0] * theta + SAT_N[1];
  const double b = SAT_N[2] * theta2 + SAT_N[3] * theta + SAT_N[4];
  const double c = SAT_N[5] * theta2 + SAT_N[6] * theta + SAT_N[7];
  double x = 2.0 * c / (-b + sqrt(b * b - 4.0 * a * c));
  x = x * x;
  p = 1.0e6 * x * x;
  return 0;
}

__device__ __forceinline__ int sat_temperature(double p, double& t) {
  if (!((p >= 611.213) && (p <= PCRITICAL))) return 1;
  const double beta2 = sqrt(p / 1.0e6);
  const double beta = sqrt(beta2);
  const double e = beta2 + SAT_N[2] * beta + SAT_N[5];
  const double f = SAT_N[0] * beta2 + SAT_N[3] * beta + SAT_N[6];
  const double g = SAT_N[1] * beta2 + SAT_N[4] * beta + SAT_N[7];
  const double d = 2.0 * g / (-f - sqrt(f * f - 4.0 * e * g));
  const double x = SAT_N[9] + d;
  t = 0.5 * (SAT_N[9] + d - sqrt(x * x - 4.0 * (SAT_N[8] + SAT_N[9] * d))) - TC_K;
  return 0;
}

template <int... Is>
__device__ __forceinline__ double visc_s1(double a, double b, std::integer_sequence<int, Is...>) {
  return (((powi<VISC_I[Is]>(a, 0.0) * VISC_H1[Is]) * powi<VISC_J[Is]>(b, 0.0)) + ...);
}

__device__ __forceinline__ double viscosity(double t, double rho) {
  const double tk = t + TC_K, tau = tk / TCRITICALK, del = rho / DCRITICAL;
  const double it = 1.0 / tau;
  const double s0 = ((VISC_H0[0] + VISC_H0[1] * it) + VISC_H0[2] * (it * it)) +
                    VISC_H0[3] * (it * (it * it));
  const double mu0 = 100.0 * sqrt(tau) / s0;
  const double s1 = visc_s1(it - 1.0, del - 1.0, std::make_integer_sequence<int, 21>{});
  return 1.0e-6 * mu0 * exp(del * s1);
}

__device__ __forceinline__ int phase_composition(int region, double p, double t) {
  if (region == 4) return 3;
  if (t <= TCRITICAL) {
    if (region == 1) return 1;
    if (region == 2) return 2;
    if (region == 3) {
      double ps;
      if (sat_pressure(t, ps) == 0) return (p >= ps) ? 1 : 2;
      return -1;
    }
    return 0;
  }
  return (p <= PCRITICAL) ? 2 : 4;
}

}  // namespace if97
}  // namespace wai

// Host code: the alignment optimiser of test-time ensembling, natively.
//
// The reference hands its alignment cost to scipy.optimize.minimize(method="BFGS", tol, options={"maxiter"})
// (marigold/util/ensemble.py:154-173).  Round 2 ran scipy itself with the cost / gradient on the device: ~90 cost
// evaluations per map at ~115 us each, of which 40 us are the device pass and 60-75 us are scipy's Python (BFGS update, the
// DCSRCH line search, ScalarFunction bookkeeping) and the numpy glue of the objective.  This file restates, operation for
// operation, what scipy 1.15 executes for that call:
//   scipy/optimize/_optimize.py::_minimize_bfgs (c1 = 1e-4, c2 = 0.9, norm = inf, xrtol = 0, hess_inv0 = I),
//   _line_search_wolfe12 -> _linesearch.py::line_search_wolfe1 (amin 1e-100, amax 1e100, xtol 1e-14; _dcsrch.py: DCSRCH /
//   dcstep, <= 100 iterations) with the fall-back line_search_wolfe2 (scalar_search_wolfe2 / _zoom / _cubicmin / _quadmin,
//   maxiter 10), and the value / gradient memoisation of ScalarFunction + MemoizeJac (an objective evaluation happens once
//   per distinct point).
// Differences from scipy are confined to floating-point summation order where numpy calls BLAS (dot products of <= 256
// elements); tests/test_host.py runs both on the same objectives and compares iterates and evaluation counts.
// The objective itself (marigold_amd/ensemble.py::DepthAligner.reference_fd_objective) is restated in
// mg_ens_align_minimize: closed-form pair costs from the member statistics (mg_ens_align_cost_grad), the regulariser from
// one device pass per evaluation (the caller's MG_OP_ENS_DEPTH_MEDIAN op writing into host-mapped memory), and the
// forward-difference survival factor of the reference's fp32 parameter cast.
// Built with -ffp-contract=off (Makefile) like ensemble.hip: no fused multiply-adds where numpy has none.
//
// The control flow restated here is scipy's (not the reference's): SciPy is BSD-3-Clause,
//   Copyright (c) 2001-2002 Enthought, Inc. 2003, SciPy Developers.  All rights reserved.
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the
//   conditions of the BSD 3-Clause licence are met (THIRD_PARTY_LICENSES.md at the repository root carries its full text;
//   DCSRCH / dcstep are the MINPACK-2 line search by More' and Thuente as translated in scipy/optimize/_dcsrch.py).
// The parity test pins the version this file follows: tests/test_host.py::test_native_bfgs_follows_scipy skips unless scipy
// is 1.15.x.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

extern "C" int mg_ens_align_cost_grad(int E, const double* s, const double* t, const double* mean, const double* C, double* cost,
                                      double* gs, double* gt);

namespace {

typedef int (*objective_fn)(void* user, int n, const double* x, double* f, double* g);

struct Objective {   // ScalarFunction + MemoizeJac: (f, g) of the last point, recomputed only for a different one
  objective_fn fn;
  void* user;
  int n;
  std::vector<double> x, g;
  double f = 0.0;
  bool valid = false;
  int nfev = 0, rc = 0;
  void at(const double* p) {
    if (valid && memcmp(p, x.data(), sizeof(double) * n) == 0) return;
    const int r = fn(user, n, p, &f, g.data());
    if (r && !rc) rc = r;
    ++nfev;
    memcpy(x.data(), p, sizeof(double) * n);
    valid = true;
  }
};

inline double py_max3(double a, double b, double c) {   // Python's max(): a later argument wins only if it compares greater
  double m = a;
  if (b > m) m = b;
  if (c > m) m = c;
  return m;
}
inline double py_min2(double a, double b) { return b < a ? b : a; }
inline double py_max2(double a, double b) { return b > a ? b : a; }
inline double np_sign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : (v == 0 ? 0.0 : v)); }
inline double np_clip(double v, double lo, double hi) { return v != v ? v : std::min(std::max(v, lo), hi); }

// ---- _dcsrch.py ---------------------------------------------------------------------------------------------------------
void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp, double fp, double dp, bool& brackt,
            double stpmin, double stpmax) {
  const double sgnd = np_sign(dp) * np_sign(dx);
  double stpf;
  if (fp > fx) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma *= -1;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (fabs(stpc - stx) <= fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma *= -1;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = true;
  } else if (fabs(dp) < fabs(dx)) {
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = py_max3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt(py_max2(0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (dx - dp)) + gamma;
    const double r = p / q;
    double stpc;
    if (r < 0 && gamma != 0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = py_min2(stp + 0.66 * (sty - stp), stpf);
      else stpf = py_max2(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = np_clip(stpf, stpmin, stpmax);
    }
  } else {
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = py_max3(fabs(theta), fabs(dy), fabs(dp));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dy;
      const double r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp; fy = fp; dy = dp;
  } else {
    if (sgnd < 0) { sty = stx; fy = fx; dy = dx; }
    stx = stp; fx = fp; dx = dp;
  }
  stp = stpf;
}

enum Task { T_START, T_FG, T_ERROR, T_WARN, T_CONV };

struct Dcsrch {
  double ftol, gtol, xtol, stpmin, stpmax;
  bool brackt = false;
  int stage = 0;
  double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0, stmin = 0, stmax = 0, width = 0, width1 = 0;
  Task iterate(double& stp, double f, double g, Task task) {
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    if (task == T_START) {
      if (stp < stpmin || stp > stpmax || g >= 0 || ftol < 0 || gtol < 0 || xtol < 0 || stpmin < 0 || stpmax < stpmin) return T_ERROR;
      brackt = false;
      stage = 1;
      finit = f; ginit = g;
      gtest = ftol * ginit;
      width = stpmax - stpmin;
      width1 = width / p5;
      stx = 0.0; fx = finit; gx = ginit;
      sty = 0.0; fy = finit; gy = ginit;
      stmin = 0;
      stmax = stp + xtrapu * stp;
      return T_FG;
    }
    const double ftest = finit + stp * gtest;
    if (stage == 1 && f <= ftest && g >= 0) stage = 2;
    Task t = task;
    if (brackt && (stp <= stmin || stp >= stmax)) t = T_WARN;
    if (brackt && stmax - stmin <= xtol * stmax) t = T_WARN;
    if (stp == stpmax && f <= ftest && g <= gtest) t = T_WARN;
    if (stp == stpmin && (f > ftest || g >= gtest)) t = T_WARN;
    if (f <= ftest && fabs(g) <= gtol * -ginit) t = T_CONV;
    if (t == T_WARN || t == T_CONV) return t;
    if (stage == 1 && f <= fx && f > ftest) {
      const double fm = f - stp * gtest;
      double fxm = fx - stx * gtest, fym = fy - sty * gtest;
      const double gm = g - gtest;
      double gxm = gx - gtest, gym = gy - gtest;
      dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
      fx = fxm + stx * gtest;
      fy = fym + sty * gtest;
      gx = gxm + gtest;
      gy = gym + gtest;
    } else {
      dcstep(stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax);
    }
    if (brackt) {
      if (fabs(sty - stx) >= p66 * width1) stp = stx + p5 * (sty - stx);
      width1 = width;
      width = fabs(sty - stx);
    }
    if (brackt) {
      stmin = std::min(stx, sty);
      stmax = std::max(stx, sty);
    } else {
      stmin = stp + xtrapl * (stp - stx);
      stmax = stp + xtrapu * (stp - stx);
    }
    stp = np_clip(stp, stpmin, stpmax);
    if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
    return T_FG;
  }
};

// one line search along pk from xk: phi(s) = f(xk + s pk), derphi(s) = <grad f(xk + s pk), pk>
struct Line {
  Objective& o;
  const double* xk;
  const double* pk;
  std::vector<double> xt;
  std::vector<double> gval;   // gradient at the last derphi point
  double gval_alpha = NAN;
  bool have_gval = false;
  Line(Objective& ob, const double* x, const double* p) : o(ob), xk(x), pk(p), xt(ob.n), gval(ob.n) {}
  void point(double s) {
    for (int i = 0; i < o.n; ++i) xt[i] = xk[i] + s * pk[i];
    o.at(xt.data());
  }
  double phi(double s) { point(s); return o.f; }
  double derphi(double s) {
    point(s);
    gval = o.g;
    gval_alpha = s;
    have_gval = true;
    double d = 0.0;
    for (int i = 0; i < o.n; ++i) d += gval[i] * pk[i];
    return d;
  }
};

// scalar_search_wolfe1 (DCSRCH); returns false when no step was found
bool search_wolfe1(Line& L, double phi0, double old_phi0, double derphi0, double c1, double c2, double amax, double amin, double xtol,
                   double& stp_out, double& phi1_out) {
  double alpha1 = 1.0;
  if (derphi0 != 0) {
    alpha1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
    if (alpha1 < 0) alpha1 = 1.0;
  }
  Dcsrch d{c1, c2, xtol, amin, amax};
  double phi1 = phi0, derphi1 = derphi0, stp = alpha1;
  Task task = T_START;
  bool ok = false;
  int i = 0;
  for (; i < 100; ++i) {
    stp = alpha1;
    task = d.iterate(stp, phi1, derphi1, task);
    if (!isfinite(stp)) { task = T_WARN; break; }
    if (task == T_FG) {
      alpha1 = stp;
      phi1 = L.phi(stp);
      derphi1 = L.derphi(stp);
    } else {
      break;
    }
  }
  if (i == 100) task = T_WARN;   // did not converge within max iterations
  ok = !(task == T_ERROR || task == T_WARN);
  stp_out = stp;
  phi1_out = phi1;
  return ok;
}

bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double& xmin) {
  const double C = fpa, db = b - a, dc = c - a;
  const double denom = ((db * dc) * (db * dc)) * (db - dc);
  const double d00 = dc * dc, d01 = -(db * db), d10 = -(dc * dc * dc), d11 = db * db * db;
  const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
  double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
  if (denom == 0 || !isfinite(denom) || !isfinite(A) || !isfinite(B)) return false;
  A /= denom;
  B /= denom;
  const double radical = B * B - 3 * A * C;
  if (!(radical >= 0) || A == 0 || !isfinite(A) || !isfinite(B)) return false;
  xmin = a + (-B + sqrt(radical)) / (3 * A);
  return isfinite(xmin);
}

bool quadmin(double a, double fa, double fpa, double b, double fb, double& xmin) {
  const double D = fa, C = fpa, db = b - a * 1.0;
  if (db * db == 0) return false;
  const double B = (fb - D - C * db) / (db * db);
  if (B == 0 || !isfinite(B)) return false;
  xmin = a - C / (2.0 * B);
  return isfinite(xmin);
}

bool zoom(double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo, Line& L, double phi0, double derphi0, double c1,
          double c2, double& a_star, double& val_star, double& valprime_star) {
  const int maxiter = 10;
  int i = 0;
  const double delta1 = 0.2, delta2 = 0.1;
  double phi_rec = phi0, a_rec = 0;
  for (;;) {
    const double dalpha = a_hi - a_lo;
    double a, b;
    if (dalpha < 0) { a = a_hi; b = a_lo; } else { a = a_lo; b = a_hi; }
    double a_j = 0, cchk = 0;
    bool have = false;
    if (i > 0) {
      cchk = delta1 * dalpha;
      have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j);
    }
    if (i == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
      const double qchk = delta2 * dalpha;
      have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j);
      if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
    }
    const double phi_aj = L.phi(a_j);
    if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
      phi_rec = phi_hi; a_rec = a_hi;
      a_hi = a_j; phi_hi = phi_aj;
    } else {
      const double derphi_aj = L.derphi(a_j);
      if (fabs(derphi_aj) <= -c2 * derphi0) {
        a_star = a_j; val_star = phi_aj; valprime_star = derphi_aj;
        return true;
      }
      if (derphi_aj * (a_hi - a_lo) >= 0) {
        phi_rec = phi_hi; a_rec = a_hi;
        a_hi = a_lo; phi_hi = phi_lo;
      } else {
        phi_rec = phi_lo; a_rec = a_lo;
      }
      a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
    }
    ++i;
    if (i > maxiter) return false;
  }
}

// scalar_search_wolfe2: returns alpha_star (has_alpha), phi_star, the (possibly replaced) phi0, and whether derphi_star exists
void search_wolfe2(Line& L, double& phi0, double old_phi0, double derphi0, double c1, double c2, double amax, bool& has_alpha,
                   double& alpha_star, double& phi_star, bool& has_derphi) {
  const int maxiter = 10;
  double alpha0 = 0, alpha1 = 1.0;
  if (derphi0 != 0) alpha1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
  if (alpha1 < 0) alpha1 = 1.0;
  alpha1 = py_min2(alpha1, amax);
  double phi_a1 = L.phi(alpha1), phi_a0 = phi0, derphi_a0 = derphi0;
  has_alpha = false;
  has_derphi = false;
  double derphi_star = 0;
  int i = 0;
  for (; i < maxiter; ++i) {
    if (alpha1 == 0 || alpha0 > amax) {
      has_alpha = false;
      phi_star = phi0;
      phi0 = old_phi0;
      has_derphi = false;
      return;
    }
    const bool not_first = i > 0;
    if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && not_first)) {
      has_alpha = zoom(alpha0, alpha1, phi_a0, phi_a1, derphi_a0, L, phi0, derphi0, c1, c2, alpha_star, phi_star, derphi_star);
      has_derphi = has_alpha;
      return;
    }
    const double derphi_a1 = L.derphi(alpha1);
    if (fabs(derphi_a1) <= -c2 * derphi0) {
      has_alpha = true; alpha_star = alpha1; phi_star = phi_a1; has_derphi = true;
      return;
    }
    if (derphi_a1 >= 0) {
      has_alpha = zoom(alpha1, alpha0, phi_a1, phi_a0, derphi_a1, L, phi0, derphi0, c1, c2, alpha_star, phi_star, derphi_star);
      has_derphi = has_alpha;
      return;
    }
    double alpha2 = 2 * alpha1;
    alpha2 = py_min2(alpha2, amax);
    alpha0 = alpha1;
    alpha1 = alpha2;
    phi_a0 = phi_a1;
    phi_a1 = L.phi(alpha1);
    derphi_a0 = derphi_a1;
  }
  // the loop ran out: the last trial step, without a slope
  has_alpha = true; alpha_star = alpha1; phi_star = phi_a1; has_derphi = false;
}

int bfgs(Objective& o, double* x, double gtol, int maxiter, double* fval, int* nit, int* status) {
  const int N = o.n;
  const double c1 = 1e-4, c2 = 0.9, xrtol = 0.0;
  std::vector<double> xk(x, x + N), gfk(N), pk(N), sk(N), yk(N), gfkp1(N), H(N * N, 0.0), T(N * N), A1(N * N), A2(N * N);
  o.at(xk.data());
  double old_fval = o.f;
  gfk = o.g;
  int k = 0;
  for (int i = 0; i < N; ++i) H[i * N + i] = 1.0;
  double nrm = 0.0;
  for (int i = 0; i < N; ++i) nrm += gfk[i] * gfk[i];
  double old_old_fval = old_fval + sqrt(nrm) / 2;
  int warnflag = 0;
  auto infnorm = [&](const std::vector<double>& v) { double m = 0; for (double e : v) { const double a = fabs(e); if (a > m || a != a) m = a; } return m; };
  double gnorm = infnorm(gfk);
  while (gnorm > gtol && k < maxiter) {
    for (int i = 0; i < N; ++i) {
      double a = 0.0;
      for (int j = 0; j < N; ++j) a += H[i * N + j] * gfk[j];
      pk[i] = -a;
    }
    // _line_search_wolfe12
    Line L(o, xk.data(), pk.data());
    double derphi0 = 0.0;
    for (int i = 0; i < N; ++i) derphi0 += gfk[i] * pk[i];
    double alpha_k = 0, new_fval = 0, new_old = old_fval;
    bool have_g = false;
    bool ok = search_wolfe1(L, old_fval, old_old_fval, derphi0, c1, c2, 1e100, 1e-100, 1e-14, alpha_k, new_fval);
    if (ok) {
      have_g = L.have_gval;
      if (have_g) gfkp1 = L.gval;
      new_old = old_fval;
    } else {
      Line L2(o, xk.data(), pk.data());
      double phi0 = old_fval, phi_star = 0, a_star = 0;
      bool has_alpha = false, has_derphi = false;
      search_wolfe2(L2, phi0, old_old_fval, derphi0, c1, c2, 1e100, has_alpha, a_star, phi_star, has_derphi);
      if (!has_alpha) { warnflag = 2; break; }
      alpha_k = a_star;
      new_fval = phi_star;
      new_old = phi0;
      have_g = has_derphi && L2.have_gval;
      if (have_g) gfkp1 = L2.gval;
    }
    old_old_fval = new_old;
    old_fval = new_fval;
    for (int i = 0; i < N; ++i) { sk[i] = alpha_k * pk[i]; xk[i] = xk[i] + sk[i]; }
    if (!have_g) { o.at(xk.data()); gfkp1 = o.g; }
    for (int i = 0; i < N; ++i) yk[i] = gfkp1[i] - gfk[i];
    gfk = gfkp1;
    ++k;
    gnorm = infnorm(gfk);
    if (gnorm <= gtol) break;
    double pn = 0.0, xn = 0.0;
    for (int i = 0; i < N; ++i) { pn += fabs(pk[i]) * fabs(pk[i]); xn += fabs(xk[i]) * fabs(xk[i]); }
    if (alpha_k * sqrt(pn) <= xrtol * (xrtol + sqrt(xn))) break;
    if (!isfinite(old_fval)) { warnflag = 2; break; }
    double rhok_inv = 0.0;
    for (int i = 0; i < N; ++i) rhok_inv += yk[i] * sk[i];
    const double rhok = rhok_inv == 0. ? 1000.0 : 1. / rhok_inv;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const double e = i == j ? 1.0 : 0.0;
        A1[i * N + j] = e - sk[i] * yk[j] * rhok;
        A2[i * N + j] = e - yk[i] * sk[j] * rhok;
      }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double a = 0.0;
        for (int l = 0; l < N; ++l) a += H[i * N + l] * A2[l * N + j];
        T[i * N + j] = a;
      }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double a = 0.0;
        for (int l = 0; l < N; ++l) a += A1[i * N + l] * T[l * N + j];
        H[i * N + j] = a + rhok * sk[i] * sk[j];
      }
  }
  if (warnflag != 2) {
    if (k >= maxiter) warnflag = 1;
    else {
      bool nan = gnorm != gnorm || old_fval != old_fval;
      for (int i = 0; i < N; ++i) nan = nan || xk[i] != xk[i];
      if (nan) warnflag = 3;
    }
  }
  memcpy(x, xk.data(), sizeof(double) * N);
  *fval = old_fval;
  *nit = k;
  *status = warnflag;
  return o.rc;
}

}  // namespace

// Generic entry (tests): minimise fn from x (in / out) exactly as scipy.optimize.minimize(fn, x, jac=True, method="BFGS",
// tol=gtol, options={"maxiter": maxiter}) would.  fn returns 0 and fills (*f, g[n]).
extern "C" int mg_bfgs_minimize(int (*fn)(void*, int, const double*, double*, double*), void* user, int n, double* x, double gtol,
                                int maxiter, double* fval, int* nit, int* nfev, int* status) {
  MG_REQUIRE(fn && x && n >= 1 && fval && nit && nfev && status, "bfgs_minimize: bad arguments");
  Objective o{fn, user, n};
  o.x.resize(n);
  o.g.resize(n);
  const int rc = bfgs(o, x, gtol, maxiter, fval, nit, status);
  *nfev = o.nfev;
  MG_REQUIRE(rc == 0, "bfgs_minimize: the objective callback failed (%d)", rc);
  return 0;
}

// The alignment objective of ensemble_depth (marigold_amd/ensemble.py::DepthAligner.reference_fd_objective), natively.
namespace {
struct AlignCtx {
  int E, affine, reduction;
  double lam;
  const double* mean;
  const double* C;
  const mg_op* reg_op;
  hipStream_t stream;
  float* st_host;         // [2E] scales, shifts read by the device pass
  const float* mm_host;   // [2 + 2E] (min, max of the ensembled prediction, member values at those pixels) written by it
};
inline double q32(double v) { return (double)(float)v; }
const double FD_STEP = 1.4901161193847656e-08;

int align_objective(void* user, int n, const double* p, double* f, double* g) {
  const AlignCtx& c = *(const AlignCtx*)user;
  const int E = c.E;
  std::vector<double> buf((size_t)4 * E);   // any ensemble size (the reference's optimiser has no limit either)
  double* const s = buf.data();
  double* const t = s + E;
  double* const gs = t + E;
  double* const gt = gs + E;
  for (int i = 0; i < E; ++i) { s[i] = q32(p[i]); t[i] = c.affine ? q32(p[E + i]) : 0.0; }
  double cost;
  if (int rc = mg_ens_align_cost_grad(E, s, t, c.mean, c.C, &cost, gs, gt)) return rc;
  if (c.lam > 0) {
    std::vector<float> fbuf((size_t)3 * E);
    float* const s32 = fbuf.data();
    float* const t32 = s32 + E;
    for (int i = 0; i < E; ++i) { s32[i] = (float)s[i]; t32[i] = (float)t[i]; c.st_host[i] = s32[i]; c.st_host[E + i] = t32[i]; }
    if (int rc = mg_launch(c.reg_op, (void*)c.stream)) return rc;
    if (hipStreamSynchronize(c.stream) != hipSuccess) return 3;
    const double mn = (double)c.mm_host[0], mx = (double)c.mm_host[1];
    cost += (fabs(0.0 - mn) + fabs(1.0 - mx)) * c.lam;
    for (int which = 0; which < 2; ++which) {
      const float* draw = c.mm_host + 2 + which * E;
      const double sign = which == 0 ? np_sign(mn) : -np_sign(1.0 - mx);
      if (c.reduction == 0) {   // lower-middle median: which member is it at that pixel? (stable argsort of the fp32 values)
        float* const a = t32 + E;
        std::vector<int> order((size_t)E);
        for (int i = 0; i < E; ++i) { a[i] = draw[i] * s32[i] + t32[i]; order[i] = i; }
        std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return a[u] < a[v]; });
        const int e = order[(E - 1) / 2];
        gs[e] += c.lam * sign * (double)draw[e];
        gt[e] += c.lam * sign;
      } else {
        for (int i = 0; i < E; ++i) { gs[i] += c.lam * sign * (double)draw[i] / E; gt[i] += c.lam * sign / E; }
      }
    }
  }
  // g_i x the share of scipy's forward-difference step that survives the reference's fp32 parameter cast
  for (int i = 0; i < n; ++i) {
    const double k = (q32(p[i] + FD_STEP) - q32(p[i])) / FD_STEP;
    g[i] = (i < E ? gs[i] : gt[i - E]) * k;
  }
  *f = cost;
  return 0;
}
}  // namespace

extern "C" int mg_ens_align_minimize(const mg_op* reg_op, void* stream, int E, int affine, int reduction, double lam, const double* mean,
                                     const double* C, float* st_host, const float* mm_host, double* x, double gtol, int maxiter,
                                     double* fval, int* nit, int* nfev, int* status) {
  MG_REQUIRE(E >= 1 && mean && C && x && fval && nit && nfev && status, "ens_align_minimize: bad arguments");
  MG_REQUIRE(!(lam > 0) || (reg_op && st_host && mm_host), "ens_align_minimize: the regulariser needs its device pass and host-mapped buffers");
  AlignCtx c{E, affine, reduction, lam, mean, C, reg_op, (hipStream_t)stream, st_host, mm_host};
  const int n = affine ? 2 * E : E;
  Objective o{align_objective, &c, n};
  o.x.resize(n);
  o.g.resize(n);
  const int rc = bfgs(o, x, gtol, maxiter, fval, nit, status);
  *nfev = o.nfev;
  MG_REQUIRE(rc == 0, "ens_align_minimize: an evaluation failed (%d): %s", rc, mg_last_error());
  return 0;
}

// gridpf_redispatch.hpp -- the environment's redispatching automaton as a batched kernel.
//
// BaseEnv._compute_dispatch_vect (grid2op/Environment/baseEnv.py:2211-2470) projects the redispatch the agents ask for onto the
// physical limits of the generators, one small quadratic program per environment and step (scipy SLSQP in the reference):
//
//     minimise    sum_{i in M} w_i (x_i - t_i)^2                  t_i = target_i - actual_i,  w_i ~ 1 / (ramp_up_i + ramp_down_i)
//     subject to  sum_{i in G} x_i = rhs                          rhs = storage - curtailment + detached       (:2335-2340)
//                 lo_i <= x_i <= hi_i                             pmin / pmax / ramp limits                    (:2343-2366)
//
// G = participating generators (:2227-2232), M = the generators an action modified (all of G when there is none, :2309-2312);
// afterwards actual_dispatch += x.  The program is separable with ONE coupling constraint, so it is solved exactly instead of
// iteratively: x_i(lambda) = clip(t_i - lambda / (2 w_i), lo_i, hi_i) for i in M, the zero-weight generators of G \ M sit at a
// bound when lambda != 0, and sum_G x_i(lambda) is monotone in the multiplier -> bisection.  When lambda = 0 (the modified
// generators reach their targets) what is left is shared between the generators of G \ M in proportion to 1 / w_i -- the
// starting point the reference hands to SLSQP (:2384-2408), which is where SLSQP stays when it is feasible.
//
// One wavefront per lane, generators across the SIMD lanes (<= 4 per lane: n_gen <= 256), float64 arithmetic, wave reductions
// for the sums; a few hundred instructions per lane -- the kernel exists so that batched agents with redispatch / storage actions
// never leave the device, not because it is heavy.
#pragma once
#include "gridpf_common.hpp"

namespace gpf {

struct RedispDev {
  int n_gen;
  double eps_poly;
  const double *pmin, *pmax, *ramp_up, *ramp_down;   // [n_gen]
  const unsigned char* redispatchable;               // [n_gen]
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = WAVE / 2; off; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = WAVE / 2; off; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = WAVE / 2; off; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

constexpr int RD_PER_LANE = 4;

// new_p, prev_p, actual, target: [n][n_gen] double; modified: [n][n_gen] u8; rhs: [n]; ok: [n] u8; after: [n][n_gen] float;
// delta_out (may be null): [.][n_gen] float rows of the engine's lane_gen_delta buffer starting at lane0
__global__ __launch_bounds__(WAVE) void redispatch_kernel(RedispDev R, int n, const double* __restrict__ new_p, const double* __restrict__ prev_p,
                                                          const double* __restrict__ actual, const double* __restrict__ target,
                                                          const unsigned char* __restrict__ modified, const double* __restrict__ rhs_in,
                                                          unsigned char* __restrict__ ok_out, float* __restrict__ after, float* __restrict__ delta_out) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int tid = threadIdx.x;
  const int ng = R.n_gen;
  const size_t row = (size_t)k * ng;
  bool part[RD_PER_LANE], mod[RD_PER_LANE];
  double lo[RD_PER_LANE], hi[RD_PER_LANE], w[RD_PER_LANE], tv[RD_PER_LANE], act[RD_PER_LANE], x[RD_PER_LANE];
  double s_incr = 0.0, s_up = 0.0, s_down = 0.0, s_coef = 0.0;
  int n_mod = 0;
  const double added = 0.5 * R.eps_poly;
#pragma unroll
  for (int q = 0; q < RD_PER_LANE; ++q) {
    const int i = tid + q * WAVE;
    part[q] = false; mod[q] = false; lo[q] = hi[q] = w[q] = tv[q] = act[q] = x[q] = 0.0;
    if (i < ng) {
      const double np_ = new_p[row + i], pv = prev_p[row + i], a = actual[row + i], t = target[row + i];
      const double pmin = R.pmin[i], pmax = R.pmax[i], ru = R.ramp_up[i], rd = R.ramp_down[i];
      act[q] = a;
      part[q] = ((np_ > 0.0) || (fabs(a) >= 1e-7) || (t != a)) && R.redispatchable[i];
      const double incr = np_ - (pv - a);
      if (part[q]) {
        s_incr += incr;
        s_down += fmax(pmin - pv, -rd);
        s_up += fmin(pmax - pv, ru);
        const double pth = np_ + a;
        lo[q] = fmax(pmin - pth, -rd - incr) - added;
        hi[q] = fmin(pmax - pth, ru - incr) + added;
        w[q] = 1.0 / (ru + rd + R.eps_poly);
        s_coef += w[q];
        tv[q] = t - a;
        mod[q] = modified[row + i] != 0;
        n_mod += mod[q] ? 1 : 0;
      }
    }
  }
  s_incr = wave_sum_f64(s_incr); s_up = wave_sum_f64(s_up); s_down = wave_sum_f64(s_down); s_coef = wave_sum_f64(s_coef);
  n_mod = (int)wave_sum_f64((double)n_mod);
  const double rhs = rhs_in[k];
  bool ok = true;
  const double sum_move = s_incr + rhs;                                   // :2474-2476
  if (sum_move > s_up || sum_move < s_down) ok = false;
  double s_lo = 0.0, s_hi = 0.0;
#pragma unroll
  for (int q = 0; q < RD_PER_LANE; ++q) {
    if (part[q]) { w[q] /= s_coef; if (n_mod == 0) mod[q] = true; s_lo += lo[q]; s_hi += hi[q]; }
  }
  s_lo = wave_sum_f64(s_lo); s_hi = wave_sum_f64(s_hi);
  if (rhs < s_lo || rhs > s_hi) ok = false;
  if (ok) {
    // sum of the modified generators at lambda, and of the free ones at their bounds
    double f_lo = 0.0, f_hi = 0.0, lam_lo = 1e300, lam_hi = -1e300;
#pragma unroll
    for (int q = 0; q < RD_PER_LANE; ++q) {
      if (part[q] && !mod[q]) { f_lo += lo[q]; f_hi += hi[q]; }
      if (part[q] && mod[q]) { lam_lo = fmin(lam_lo, 2.0 * w[q] * (tv[q] - hi[q])); lam_hi = fmax(lam_hi, 2.0 * w[q] * (tv[q] - lo[q])); }
    }
    f_lo = wave_sum_f64(f_lo); f_hi = wave_sum_f64(f_hi); lam_lo = wave_min_f64(lam_lo); lam_hi = wave_max_f64(lam_hi);
    auto sum_mod = [&](double lam) -> double {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < RD_PER_LANE; ++q)
        if (part[q] && mod[q]) s += fmin(fmax(tv[q] - lam / (2.0 * w[q]), lo[q]), hi[q]);
      return wave_sum_f64(s);
    };
    const double s0 = sum_mod(0.0);
    double lam = 0.0;
    int free_at = 0;                              // 0: share the remainder, +1: free generators at hi, -1: at lo
    if (rhs - s0 > f_hi) { free_at = 1; double a = lam_lo, b = 0.0;        // more is needed: lambda < 0
      for (int it = 0; it < 64; ++it) { const double mid = 0.5 * (a + b); if (sum_mod(mid) + f_hi > rhs) a = mid; else b = mid; }
      lam = 0.5 * (a + b);
    } else if (rhs - s0 < f_lo) { free_at = -1; double a = 0.0, b = lam_hi;
      for (int it = 0; it < 64; ++it) { const double mid = 0.5 * (a + b); if (sum_mod(mid) + f_lo > rhs) a = mid; else b = mid; }
      lam = 0.5 * (a + b);
    }
#pragma unroll
    for (int q = 0; q < RD_PER_LANE; ++q)
      if (part[q] && mod[q]) x[q] = fmin(fmax(tv[q] - lam / (2.0 * w[q]), lo[q]), hi[q]);
    double got = 0.0;
#pragma unroll
    for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && mod[q]) got += x[q];
    got = wave_sum_f64(got);
    if (free_at != 0) {
#pragma unroll
      for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && !mod[q]) x[q] = free_at > 0 ? hi[q] : lo[q];
      // the bisection leaves a residual of a few ulps: spread it over the generators that are strictly inside their bounds
      const double rest = rhs - got - (free_at > 0 ? f_hi : f_lo);
      int n_in = 0;
#pragma unroll
      for (int q = 0; q < RD_PER_LANE; ++q) n_in += (part[q] && mod[q] && x[q] > lo[q] && x[q] < hi[q]) ? 1 : 0;
      n_in = (int)wave_sum_f64((double)n_in);
      if (n_in > 0) {
#pragma unroll
        for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && mod[q] && x[q] > lo[q] && x[q] < hi[q]) x[q] += rest / n_in;
      }
    } else {
      // lambda = 0: the free generators share r in proportion to 1 / w_i, clipped: x_i = clip(alpha / w_i)
      const double r = rhs - got;
      double a = -1e300, b = 1e300;
      // bracket alpha: all at lo / all at hi
      double a_lo = 1e300, a_hi = -1e300;
#pragma unroll
      for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && !mod[q]) { a_lo = fmin(a_lo, fmin(lo[q] * w[q], hi[q] * w[q])); a_hi = fmax(a_hi, fmax(lo[q] * w[q], hi[q] * w[q])); }
      a = wave_min_f64(a_lo); b = wave_max_f64(a_hi);
      if (a <= b) {
        for (int it = 0; it < 64; ++it) {
          const double mid = 0.5 * (a + b);
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && !mod[q]) s += fmin(fmax(mid / w[q], lo[q]), hi[q]);
          s = wave_sum_f64(s);
          if (s < r) a = mid; else b = mid;
        }
        const double alpha = 0.5 * (a + b);
#pragma unroll
        for (int q = 0; q < RD_PER_LANE; ++q) if (part[q] && !mod[q]) x[q] = fmin(fmax(alpha / w[q], lo[q]), hi[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RD_PER_LANE; ++q) {
    const int i = tid + q * WAVE;
    if (i < ng) {
      const float v = (float)(act[q] + (ok && part[q] ? x[q] : 0.0));
      after[row + i] = v;
      if (delta_out && ok) delta_out[row + i] = v;
    }
  }
  if (tid == 0) ok_out[k] = ok ? 1 : 0;
}

}  // namespace gpf

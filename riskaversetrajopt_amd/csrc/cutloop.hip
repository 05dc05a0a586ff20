// Host side of the boundary, continued (nnls.hip, master.hip): the CUTTING-PLANE LOOP of one reduced SCP subproblem as
// ONE library call -- HOST code only; the device work it issues is the table-free cut oracle of cvar.hip.
//
// The reference hands every SCP subproblem to OSQP as one QP with M auxiliary variables (drone_risk.py:425-469,
// driving.py:423-456).  Here the subproblem is that QP reduced exactly to (u, slack) (riskaversetrajopt_amd/cvar_cuts.py
// has the derivation) and solved by Kelley cuts: master QP on the host (rato_master_*), oracle on the device
// (rato_cut_oracle_rollout: rowmax -> exact tail selection -> tail-row sums -> read-back, one round trip per cut).
// Round 3 drove that loop from Python (cvar_cuts.CvarCutSolver._solve): ~20 ms of interpreter time per 60 SCP iterations
// against ~35 ms of device time.  This file is the same loop, statement for statement -- lazy control bounds, the "last
// cut joins the master" rule, the keep rule for recycled cuts, the multipliers for the KKT certificate -- so that ONE
// ctypes call per subproblem remains (rato_cut_begin is its stream-ordered prologue).  The Python loop stays as the
// implementation for sharded batches and the table forms of the oracle, and as the checker: both produce BITWISE the
// same iterates (tests/test_gpu_scp.py), which is why every inner product that feeds the master is an exactly rounded
// sum here and there (fsum below == math.fsum).
#pragma clang fp contract(off)   // the exactly-rounded sums below must see individually rounded products

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>

#include "rato_common.h"
#include "rato_saa.h"

namespace {

// Shewchuk's exact summation (the algorithm of CPython's math.fsum, Modules/mathmodule.c): the correctly rounded value
// of the exact sum of the inputs -- independent of their order, hence identical to math.fsum on the same numbers.
double fsum(const double* v, int n) {
  std::vector<double> p;
  p.reserve(32);
  for (int k = 0; k < n; ++k) {
    double x = v[k];
    size_t i = 0;
    for (size_t j = 0; j < p.size(); ++j) {
      double y = p[j];
      if (fabs(x) < fabs(y)) std::swap(x, y);
      const double hi = x + y;
      const double lo = y - (hi - x);
      if (lo != 0.0) p[i++] = lo;
      x = hi;
    }
    p.resize(i);
    p.push_back(x);
  }
  double hi = 0.0;
  size_t n_p = p.size();
  if (n_p > 0) {
    hi = p[--n_p];
    double lo = 0.0;
    while (n_p > 0) {
      const double x = hi;
      const double y = p[--n_p];
      hi = x + y;
      const double yr = hi - x;
      lo = y - yr;
      if (lo != 0.0) break;
    }
    // round half to even across the remaining partials
    if (n_p > 0 && ((lo < 0.0 && p[n_p - 1] < 0.0) || (lo > 0.0 && p[n_p - 1] > 0.0))) {
      const double y = lo * 2.0;
      const double x = hi + y;
      const double yr = x - hi;
      if (y == yr) hi = x;
    }
  }
  return hi;
}

double dot_exact(const double* a, const double* b, int n, std::vector<double>& prod) {
  prod.resize(n);
  for (int i = 0; i < n; ++i) prod[i] = a[i] * b[i];
  return fsum(prod.data(), n);
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

struct rato_cut_solver {
  rato_cut_config c;
  rato_drone_params drone;
  rato_car_params car;
  int nU = 0, n = 0, nc = 0, nres = 0, nblk = 0;
  std::vector<double> p_diag, q;
  hipEvent_t sums_ready = nullptr;   // rato_cut_define_drone: recorded behind the linearization's sample sums (lazily created)
  bool kept_armed = false;           // sums_b_host was pre-set before the kept cuts' launch in flight (readback_arm): the wait may watch it
  const void* params() const { return c.system == 0 ? (const void*)&drone : (const void*)&car; }
  ~rato_cut_solver() {
    if (sums_ready) (void)hipEventDestroy(sums_ready);
  }
};

extern "C" int rato_cut_solver_create(rato_cut_solver** out, const rato_cut_config* cfg) {
  if (!out || !cfg) return RATO_EINVAL;
  const rato_cut_config& c = *cfg;
  if ((c.system != 0 && c.system != 1) || !c.params || c.S < 1 || c.M < 1 || c.cap < 2 || c.keep_max < 0 ||
      c.keep_recent < 0 || c.keep_idle < 0 || !(c.alphaM > 0.0) || !c.s0 || !c.s1 || !c.s2 || (c.system == 1 && !c.s3) ||
      !c.uk_dev || !c.uk_host || !c.x_host || !c.x_dev || !c.ring_m || !c.ring_arg || !c.ring_res || !c.workspace ||
      !c.res_host || !c.p_diag || !c.q)
    return RATO_EINVAL;
  if (c.S > 1 && (!c.part || (c.keep_max > 0 && (!c.part_b || !c.sums_b_host || !c.slots_dev || !c.slots_host))))
    return RATO_EINVAL;
  rato_cut_solver* s = new rato_cut_solver;
  s->c = c;
  int n_u;
  if (c.system == 0) {
    s->drone = *static_cast<const rato_drone_params*>(c.params);
    if (s->drone.S != c.S || s->drone.M != c.M) { delete s; return RATO_EINVAL; }
    n_u = 3;
  } else {
    s->car = *static_cast<const rato_car_params*>(c.params);
    if (s->car.S != c.S || s->car.M != c.M) { delete s; return RATO_EINVAL; }
    n_u = 2;
  }
  s->nU = n_u * c.S;
  s->n = s->nU + 1;
  s->nc = 2 * std::max(c.S - 1, 0) + 1;
  s->nres = RATO_N_STATS + s->nc;
  s->nblk = (int)((c.M + 255) / 256);
  s->p_diag.assign(c.p_diag, c.p_diag + s->n);
  s->q.assign(c.q, c.q + s->n);
  s->c.params = nullptr;   // (the copy above is what is used)
  *out = s;
  return RATO_OK;
}

extern "C" void rato_cut_solver_destroy(rato_cut_solver* s) { delete s; }

extern "C" size_t rato_cut_config_bytes(void) { return sizeof(rato_cut_config); }
extern "C" size_t rato_cut_result_bytes(void) { return sizeof(rato_cut_result); }

namespace {

int tail_rows_launch(rato_cut_solver* s, const float* m_base, const int32_t* arg_base, const double* res_base,
                     const int32_t* slots, int K, double* part, void* stream) {
  const rato_cut_config& c = s->c;
  return c.system == 0 ? rato_drone_tail_rows_rollout(&s->drone, c.uk_dev, c.s0, c.s1, c.s2, m_base, arg_base, res_base,
                                                      s->nres, slots, K, c.alphaM, part, stream)
                       : rato_car_tail_rows_rollout(&s->car, c.uk_dev, c.s0, c.s1, c.s2, c.s3, m_base, arg_base,
                                                    res_base, s->nres, slots, K, c.alphaM, part, stream);
}

// u_k, the controls and the kept slots reach the device IN THE ARGUMENTS of one small launch: an asynchronous copy of a
// kilobyte costs ~25 us of host time on this stack (the define issued three), a launch ~5.
constexpr int STAGE_MAX = 192, STAGE_SLOTS = 64;
struct StageArgs {
  double uk[STAGE_MAX];
  float us[STAGE_MAX];
  int32_t slots[STAGE_SLOTS];
};
__global__ void cut_stage_kernel(const StageArgs a, int n_uk, int n_us, int n_slots, double* __restrict__ uk_dev,
                                 float* __restrict__ us_dev, int32_t* __restrict__ slots_dev) {
  const int i = threadIdx.x;
  if (i < n_uk) uk_dev[i] = a.uk[i];
  if (i < n_us) us_dev[i] = a.us[i];
  if (i < n_slots) slots_dev[i] = a.slots[i];
}

// u_lin -> uk_dev (fp64), us (optional) -> us_dev (fp32), keep -> slots_dev: one launch when they fit its arguments
int stage_inputs(rato_cut_solver* s, const double* u_lin, const float* us_f32, float* us_dev, const int32_t* keep, int n_keep,
                 hipStream_t st) {
  const rato_cut_config& c = s->c;
  const int nU = s->nU;
  if (nU <= STAGE_MAX && n_keep <= STAGE_SLOTS) {
    StageArgs a;
    memcpy(a.uk, u_lin, sizeof(double) * (size_t)nU);
    if (us_f32) memcpy(a.us, us_f32, sizeof(float) * (size_t)nU);
    for (int k = 0; k < n_keep; ++k) a.slots[k] = keep[k];
    hipLaunchKernelGGL(cut_stage_kernel, dim3(1), dim3(STAGE_MAX), 0, st, a, nU, us_f32 ? nU : 0, n_keep, c.uk_dev, us_dev,
                       c.slots_dev);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? RATO_OK : RATO_EHIP - (int)e;
  }
  memcpy(c.uk_host, u_lin, sizeof(double) * (size_t)nU);
  hipError_t e = hipMemcpyAsync(c.uk_dev, c.uk_host, sizeof(double) * (size_t)nU, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return RATO_EHIP - (int)e;
  if (us_f32) {   // (us_f32 is the caller's pinned buffer)
    e = hipMemcpyAsync(us_dev, us_f32, sizeof(float) * (size_t)nU, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  if (n_keep > 0) {
    for (int k = 0; k < n_keep; ++k) c.slots_host[k] = keep[k];
    e = hipMemcpyAsync(c.slots_dev, c.slots_host, sizeof(int32_t) * (size_t)n_keep, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  return RATO_OK;
}

// the kept cuts' tail-row sums under the linearization point staged last, reduced into pinned host memory
int kept_cuts_launch(rato_cut_solver* s, int n_keep, void* stream) {
  const rato_cut_config& c = s->c;
  if (n_keep == 0 || c.S < 2) return RATO_OK;
  int rc = tail_rows_launch(s, c.ring_m, c.ring_arg, c.ring_res, c.slots_dev, n_keep, c.part_b, stream);
  if (rc != RATO_OK) return rc;
  return rato_sum_partials_f64(c.part_b, s->nblk, n_keep * s->nc, 1.0, c.sums_b_host, stream);
}

// A kept-cuts launch is still marked in flight (define / begin followed by another define / begin without the solve that
// waits for it): let it finish before its pinned words are pre-set again -- a late write of the OLD launch would
// otherwise satisfy the NEW wait with stale sums.  Clears the mark whatever happens.
int settle_kept(rato_cut_solver* s, hipStream_t st) {
  if (!s->kept_armed) return RATO_OK;
  s->kept_armed = false;
  const hipError_t e = hipStreamSynchronize(st);
  return e == hipSuccess ? RATO_OK : RATO_EHIP - (int)e;
}

bool keep_ok(const rato_cut_solver* s, const int32_t* keep, int n_keep) {
  for (int k = 0; k < n_keep; ++k)
    if (keep[k] < 0 || keep[k] >= s->c.cap - 1) return false;
  return true;
}

}  // namespace

// Stream-ordered prologue of a subproblem: u_k -> device (fp64), and -- when cuts were kept from the previous
// subproblem -- their tail-row sums under the NEW linearization point, reduced straight into pinned host memory
// (sums_b_host).  Nothing is synchronised: the caller's own read-back of the linearization's sample sums waits for
// this work too (one device round trip per "define" instead of two).
extern "C" int rato_cut_begin(rato_cut_solver* s, const double* u_lin, const int32_t* keep, int32_t n_keep, void* stream) {
  if (!s || !u_lin || n_keep < 0 || n_keep > s->c.keep_max || (n_keep > 0 && !keep)) return RATO_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool with_cuts = n_keep > 0 && s->c.S >= 2;
  if (with_cuts && !keep_ok(s, keep, n_keep)) return RATO_EINVAL;
  int rc = settle_kept(s, st);
  if (rc != RATO_OK) return rc;
  rc = stage_inputs(s, u_lin, nullptr, nullptr, keep, with_cuts ? n_keep : 0, st);
  if (rc != RATO_OK) return rc;
  const bool arm_kept = with_cuts && rato::readback_poll_enabled();
  if (arm_kept) rato::readback_arm(s->c.sums_b_host, n_keep * s->nc);
  rc = kept_cuts_launch(s, with_cuts ? n_keep : 0, stream);
  s->kept_armed = arm_kept && rc == RATO_OK;   // armed words with nothing launched behind them are never waited for
  return rc;
}

// The "define" half of a reduced SCP iteration of the DRONE as one call (table-free oracle: no linearization table is
// kept): the controls to the device, the generators-only linearization at them (what is left of it here: the sample
// sums of the final rows and Z), the reduction of those sums straight into PINNED host memory, the count of
// non-finite outputs, rato_cut_begin (u_k in fp64 + the kept cuts against it) and ONE synchronisation.  Round 3 issued
// these from Python: ~0.15 ms of interpreter time per SCP iteration around 0.1 ms of device work.
//   us [S][3] doubles (host);  us_host (pinned) / us_dev: [S][3] floats;  A22 [S][3][ld] floats (the kernel's scratch);
//   Z [ld] or NULL (nothing downstream of this call reads it: without it no obstacle row is formed at all);
//   part [ceil(M/256)][6S+6] floats;  sums_host (pinned): 6S+6 doubles (sums, not means);
//   bad_dev / bad_host (pinned): one uint32 each, or both NULL (no non-finite check).
extern "C" int rato_cut_define_drone(rato_cut_solver* s, const double* us, float* us_host, float* us_dev, float* A22,
                                     float* Z, int64_t z_floats, float* part, double* sums_host, uint32_t* bad_dev,
                                     uint32_t* bad_host, const int32_t* keep, int32_t n_keep, void* stream) {
  if (!s || s->c.system != 0 || !us || !us_host || !us_dev || !A22 || !part || !sums_host || (!bad_dev != !bad_host) ||
      (bad_dev && !Z) || (Z && z_floats < s->c.M))
    return RATO_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nU = s->nU, S = s->c.S, ncols = 6 * S + 6;
  if (n_keep < 0 || n_keep > s->c.keep_max || (n_keep > 0 && !keep)) return RATO_EINVAL;
  const bool with_cuts = n_keep > 0 && S >= 2;
  if (with_cuts && !keep_ok(s, keep, n_keep)) return RATO_EINVAL;
  for (int i = 0; i < nU; ++i) us_host[i] = (float)us[i];
  hipError_t e = hipSuccess;
  // the sample sums land in pinned memory and are watched for (rato_common.h: readback_*); with the non-finite count
  // (a copy node behind them) the event below is waited for instead
  const bool watch = rato::readback_poll_enabled() && !bad_dev;
  int rc = settle_kept(s, st);   // a kept-cuts launch nobody waited for must not write into words armed anew
  if (rc != RATO_OK) return rc;
  if (watch) rato::readback_arm(sums_host, ncols);
  const bool arm_kept = with_cuts && rato::readback_poll_enabled();
  if (arm_kept) rato::readback_arm(s->c.sums_b_host, n_keep * s->nc);   // (kept_armed only once the launch is out)
  rc = stage_inputs(s, us, us_host, us_dev, keep, with_cuts ? n_keep : 0, st);   // us, u_k (fp64) and the kept slots: one launch
  if (rc != RATO_OK) return rc;
  rc = rato_drone_linearize_generators(&s->drone, us_dev, s->c.s0, s->c.s1, s->c.s2, A22, nullptr, nullptr, Z, part,
                                           stream);
  if (rc != RATO_OK) return rc;
  if ((rc = rato_sum_partials(part, s->nblk, ncols, 1.0, sums_host, stream)) != RATO_OK) return rc;
  if (bad_dev) {
    if ((rc = rato_count_nonfinite(Z, z_floats, bad_dev, stream)) != RATO_OK) return rc;
    if ((rc = rato_count_nonfinite_acc(part, (int64_t)s->nblk * ncols, bad_dev, stream)) != RATO_OK) return rc;
    e = hipMemcpyAsync(bad_host, bad_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  // The host continues (sample sums -> equality rows -> the master's factorisation) as soon as the SUMS are back; the kept
  // cuts' re-linearization behind them is waited for by rato_cut_solve(kept_in_flight = 1), after it has built the master.
  if (!watch) {
    if (!s->sums_ready) {
      e = hipEventCreateWithFlags(&s->sums_ready, hipEventDisableTiming);
      if (e != hipSuccess) return RATO_EHIP - (int)e;
    }
    e = hipEventRecord(s->sums_ready, st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  if ((rc = kept_cuts_launch(s, with_cuts ? n_keep : 0, stream)) != RATO_OK) return rc;
  s->kept_armed = arm_kept;
  e = watch ? rato::readback_wait(sums_host, ncols, st) : hipEventSynchronize(s->sums_ready);
  if (e != hipSuccess) return RATO_EHIP - (int)e;
  if (bad_host && *bad_host) return RATO_ENONFINITE;
  return RATO_OK;
}

extern "C" int rato_cut_solve(rato_cut_solver* s, const double* final_du, const double* final_rhs, int32_t n_c,
                              const double* u_lin, int32_t with_cvar, double tol, int32_t max_cuts,
                              double final_cut_above, int32_t check_finite, int32_t* keep, int32_t* keep_idle_count,
                              int32_t* n_keep_io, int32_t kept_in_flight, rato_cut_result* out, void* stream) {
  if (!s || !final_du || !final_rhs || n_c < 0 || !u_lin || !out || !out->us || !keep || !keep_idle_count || !n_keep_io ||
      max_cuts < 0)
    return RATO_EINVAL;
  const rato_cut_config& c = s->c;
  const int nU = s->nU, n = s->n, S = c.S, nc = s->nc, n_u = nU / S;
  const bool saa = c.mode_saa != 0;
  const bool cvar = with_cvar != 0;
  const bool slack_row = cvar && saa;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double oracle_s = 0.0, master_s = 0.0;
  std::vector<double> prod, row(n), g(nU), x(nU), z(n), lam, lam_tmp;

  auto t0 = std::chrono::steady_clock::now();
  // master: equalities [final_du | 0] z = final_rhs
  std::vector<double> F((size_t)n_c * n, 0.0);
  for (int r = 0; r < n_c; ++r) memcpy(&F[(size_t)r * n], final_du + (size_t)r * nU, sizeof(double) * nU);
  rato_master* master = nullptr;
  int rc = rato_master_create(&master, n, s->p_diag.data(), s->q.data(), n_c, F.data(), final_rhs);
  if (rc != RATO_OK) return RATO_ERANK;
  struct Guard {
    rato_master* m;
    ~Guard() { rato_master_destroy(m); }
  } guard{master};
  int n_rows = 0;
  if (slack_row) {   // -slack <= 0
    std::fill(row.begin(), row.end(), 0.0);
    row[nU] = -1.0;
    const double zero = 0.0;
    if ((rc = rato_master_add_rows(master, 1, row.data(), &zero)) != RATO_OK) return rc;
    n_rows = 1;
  }
  std::vector<std::pair<int, int>> cut_rows;   // (row of the master, ring slot)
  const int n_kept = (cvar && c.recycle) ? *n_keep_io : 0;
  if (n_kept < 0 || n_kept > c.keep_max) return RATO_EINVAL;
  // the kept slots index host tables below (is_kept, idle) whether or not their re-linearization is already in flight:
  // checked here unconditionally, not only inside rato_cut_begin / rato_cut_define_drone
  if (*n_keep_io < 0 || *n_keep_io > c.keep_max || !keep_ok(s, keep, *n_keep_io)) return RATO_EINVAL;
  master_s += seconds_since(t0);

  if (n_kept > 0 && S >= 2) {
    t0 = std::chrono::steady_clock::now();
    if (!kept_in_flight) {
      if ((rc = rato_cut_begin(s, u_lin, keep, n_kept, stream)) != RATO_OK) return rc;
    }
    // (the kept cuts' sums were armed where they were launched: rato_cut_begin / rato_cut_define_drone)
    const bool armed = s->kept_armed;
    s->kept_armed = false;
    hipError_t e = armed ? rato::readback_wait(c.sums_b_host, n_kept * nc, st) : hipStreamSynchronize(st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
    // (not armed: the words were not watched -- a define that failed before its launch leaves them pre-set; never data)
    if (!armed && rato::readback_pending(c.sums_b_host, n_kept * nc)) return RATO_EHIP - (int)hipErrorNotReady;
    oracle_s += seconds_since(t0);
    t0 = std::chrono::steady_clock::now();
    // cut k under the current linearization (delta form):  rows_k . (u - u_k) + c0_k - c_s s <= rhs0
    std::vector<double> rows((size_t)n_kept * n, 0.0), rhs(n_kept);
    for (int k = 0; k < n_kept; ++k) {
      const double* r = c.sums_b_host + (size_t)k * nc;
      double* rk = &rows[(size_t)k * n];
      for (int t = 0; t < S - 1; ++t) {
        rk[t * n_u + 0] = r[2 * t + 0] / c.alphaM;
        rk[t * n_u + 1] = r[2 * t + 1] / c.alphaM;
      }
      rk[nU] = -c.c_s;
      const double d = dot_exact(rk, u_lin, nU, prod);
      rhs[k] = (c.rhs0 + d) - 1.0 * (r[nc - 1] / c.alphaM);
      cut_rows.emplace_back(n_rows + k, keep[k]);
    }
    if ((rc = rato_master_add_rows(master, n_kept, rows.data(), rhs.data())) != RATO_OK) return rc;
    n_rows += n_kept;
    master_s += seconds_since(t0);
  }
  std::vector<int> free_slots;   // ascending; the loop takes from the back
  {
    std::vector<uint8_t> is_kept(c.cap, 0);
    for (int k = 0; k < n_kept; ++k) is_kept[keep[k]] = 1;
    for (int sl = 0; sl < c.cap - 1; ++sl)
      if (!is_kept[sl]) free_slots.push_back(sl);
  }
  std::vector<uint8_t> in_master(2 * (size_t)nU, 0);
  struct BoundRows {
    int r0;
    std::vector<int> idx;
    double sgn;
  };
  std::vector<BoundRows> bound_rows;

  auto solve_master = [&]() -> int {   // the master with the control bounds entering lazily: only the violated ones
    for (;;) {
      lam.assign(n_rows, 0.0);
      const int r = rato_master_solve(master, z.data(), lam.data());
      if (r == RATO_EINFEASIBLE) return RATO_EINFEASIBLE;
      if (r != 1) return RATO_ENNLS;
      std::vector<int> hi, lo;
      for (int i = 0; i < nU; ++i) {
        if (z[i] > c.u_max + 1e-9 && !in_master[i]) hi.push_back(i);
        if (z[i] < c.u_min - 1e-9 && !in_master[nU + i]) lo.push_back(i);
      }
      if (hi.empty() && lo.empty()) return RATO_OK;
      for (int pass = 0; pass < 2; ++pass) {
        const std::vector<int>& idx = pass == 0 ? hi : lo;
        if (idx.empty()) continue;
        const double sgn = pass == 0 ? 1.0 : -1.0;
        std::vector<double> R((size_t)idx.size() * n, 0.0), b(idx.size(), pass == 0 ? c.u_max : -c.u_min);
        for (size_t k = 0; k < idx.size(); ++k) {
          R[k * n + idx[k]] = sgn;
          in_master[(pass == 0 ? 0 : nU) + idx[k]] = 1;
        }
        const int r2 = rato_master_add_rows(master, (int)idx.size(), R.data(), b.data());
        if (r2 != RATO_OK) return r2;
        bound_rows.push_back({n_rows, idx, sgn});
        n_rows += (int)idx.size();
      }
    }
  };

  double phi = NAN, tstar = NAN;
  int n_cuts = 0, status = 0;
  std::vector<double> z_prev(n);
  bool have_prev = false;
  const int scratch = c.cap - 1;
  const size_t M = (size_t)c.M;
  for (int it = 0; it <= max_cuts; ++it) {
    t0 = std::chrono::steady_clock::now();
    if ((rc = solve_master()) != RATO_OK) return rc;
    master_s += seconds_since(t0);
    if (!cvar) break;
    t0 = std::chrono::steady_clock::now();
    int slot = -1;
    if (!free_slots.empty()) {
      slot = free_slots.back();
      free_slots.pop_back();
    }
    const int ring = slot >= 0 ? slot : scratch;
    for (int i = 0; i < nU; ++i) {
      x[i] = z[i] - u_lin[i];
      c.x_host[i] = x[i];
    }
    rc = rato_cut_oracle_rollout(c.system, s->params(), c.uk_dev, c.s0, c.s1, c.s2, c.s3, c.x_host, c.x_dev,
                                 c.ring_m + (size_t)ring * M, c.ring_arg + (size_t)ring * M, c.alpha, c.thr, c.alphaM,
                                 c.workspace, c.workspace_bytes, c.ring_res + (size_t)ring * s->nres, c.part, c.res_host,
                                 stream);
    if (rc != RATO_OK) return rc;
    const double* r = c.res_host;
    if (isnan(r[0])) return RATO_ESELECT;     // the one-launch selection gave up (or the m values hold NaN): the caller
    //                                           repeats the subproblem with the recovering Python loop
    if (check_finite && !(isfinite(r[3]) && isfinite(r[4]))) return RATO_ENONFINITE;
    std::fill(g.begin(), g.end(), 0.0);
    if (S > 1) {
      for (int t = 0; t < S - 1; ++t) {
        g[t * n_u + 0] = r[RATO_N_STATS + 2 * t + 0] / c.alphaM;
        g[t * n_u + 1] = r[RATO_N_STATS + 2 * t + 1] / c.alphaM;
      }
      phi = dot_exact(g.data(), x.data(), nU, prod) + 1.0 * (r[RATO_N_STATS + nc - 1] / c.alphaM);
    } else {
      phi = r[1];
    }
    tstar = r[0];
    oracle_s += seconds_since(t0);
    const double slack = z[nU];
    const double viol = phi - c.c_s * slack - c.rhs0;
    // stall: the cut added last moved nothing although it was violated -- the oracle returns cuts the master already
    // holds; what is left of the violation is the accuracy of the master's own NNLS (cvar_cuts.py: STALL_*)
    if (have_prev && viol > tol && viol <= 1e-7) {
      double step = 0.0;
      for (int i = 0; i < n; ++i) step = fmax(step, fabs(z[i] - z_prev[i]));
      if (step <= 1e-10) {
        status = 2;
        break;
      }
    }
    z_prev = z;
    have_prev = true;
    auto add_cut = [&]() -> int {   // phi(u) >= phi_k + g_k.(u - u_k)  =>  g_k.u - c_s s <= rhs0 + g_k.u_k - phi_k
      memcpy(row.data(), g.data(), sizeof(double) * nU);
      row[nU] = -c.c_s;
      const double rhs = c.rhs0 + (dot_exact(g.data(), z.data(), nU, prod) - phi);
      const int r2 = rato_master_add_rows(master, 1, row.data(), &rhs);
      if (r2 != RATO_OK) return r2;
      if (slot >= 0) cut_rows.emplace_back(n_rows, slot);
      n_rows += 1;
      n_cuts += 1;
      return RATO_OK;
    };
    if (viol <= tol) {
      if (viol > final_cut_above && it < max_cuts) {   // the cut just evaluated is paid for: it joins the master
        t0 = std::chrono::steady_clock::now();
        if ((rc = add_cut()) != RATO_OK) return rc;
        if ((rc = solve_master()) != RATO_OK) return rc;
        master_s += seconds_since(t0);
      }
      break;
    }
    if (it == max_cuts) {
      status = 1;
      break;
    }
    if ((rc = add_cut()) != RATO_OK) return rc;
  }

  // multipliers of the last master solve, for whoever certifies the solution against the full QP
  std::vector<double> lam_full(n_rows, 0.0);
  for (size_t i = 0; i < lam.size() && i < (size_t)n_rows; ++i) lam_full[i] = lam[i];
  if (cvar && c.recycle) {
    // keep rule: the cuts that carry a multiplier (newest first; keep_idle > 0: or did within the last keep_idle solves),
    // plus the newest keep_recent
    std::vector<int> idle(c.cap, -1);
    for (int k = 0; k < *n_keep_io; ++k) idle[keep[k]] = keep_idle_count[k];
    for (auto& cr : cut_rows) {
      const bool active = cr.first < (int)lam.size() && lam[cr.first] > 1e-12;
      idle[cr.second] = active ? 0 : (idle[cr.second] < 0 ? 0 : idle[cr.second]) + 1;
    }
    std::vector<int> new_keep;
    auto push = [&](int sl) {
      if (std::find(new_keep.begin(), new_keep.end(), sl) == new_keep.end()) new_keep.push_back(sl);
    };
    for (auto it2 = cut_rows.rbegin(); it2 != cut_rows.rend(); ++it2)
      if (idle[it2->second] <= c.keep_idle) push(it2->second);
    int cnt = 0;
    for (auto it2 = cut_rows.rbegin(); it2 != cut_rows.rend() && cnt < c.keep_recent; ++it2, ++cnt) push(it2->second);
    if ((int)new_keep.size() > c.keep_max) new_keep.resize(c.keep_max);
    for (size_t k = 0; k < new_keep.size(); ++k) {
      keep[k] = new_keep[k];
      keep_idle_count[k] = idle[new_keep[k]];
    }
    *n_keep_io = (int)new_keep.size();
  }
  memcpy(out->us, z.data(), sizeof(double) * nU);
  out->slack = z[nU];
  out->t_risk = slack_row ? tstar + z[nU] : 0.0;
  out->phi = phi;
  out->oracle_s = oracle_s;
  out->master_s = master_s;
  out->cuts = n_cuts;
  out->recycled = n_kept;
  out->status = status;
  out->lam_slack = slack_row ? lam_full[0] : 0.0;
  out->uncertified_cuts = n_cuts + n_kept - (int)cut_rows.size();
  out->n_cut_rows = 0;
  if (out->cut_slot && out->cut_lambda) {
    for (auto& cr : cut_rows) {
      if (out->n_cut_rows >= out->cut_capacity) break;
      out->cut_slot[out->n_cut_rows] = cr.second;
      out->cut_lambda[out->n_cut_rows] = lam_full[cr.first];
      ++out->n_cut_rows;
    }
  }
  out->n_bounds = 0;
  if (out->bound_var && out->bound_sign && out->bound_lambda) {
    for (auto& br : bound_rows)
      for (size_t k = 0; k < br.idx.size(); ++k) {
        if (out->n_bounds >= out->bound_capacity) break;
        out->bound_var[out->n_bounds] = br.idx[k];
        out->bound_sign[out->n_bounds] = br.sgn;
        out->bound_lambda[out->n_bounds] = lam_full[br.r0 + (int)k];
        ++out->n_bounds;
      }
  }
  return RATO_OK;
}


// The reduced SCP of the drone as ONE call: `iters` iterations of [rato_cut_define_drone at the current controls -> the
// equality rows from the sample sums -> rato_cut_solve], the reference's fixed-count protocol (drone_risk.py:519-532,
// drone_times.py:509-550) with its per-iteration wall clocks taken here.  Each iteration is timed from its first
// instruction to the moment its solution is on the host (the last oracle round trip has been read back: the device has
// nothing of this iteration left to do); the stream is synchronised ONCE, after the last iteration, inside that
// iteration's clock.  scp.run_drone_reduced's Python loop (one define + one solve call per iteration, a device
// synchronisation on both sides of each) stays as the checker: same iterates bit for bit (tests/test_gpu_scp.py).
//   us0 [S][3]: the initial guess;  first_cvar: the first iteration with the CVaR rows (2: drone_risk.py:413-417);
//   us_hist [iters][S][3] (host): the solution of every iteration;  rec [iters];
//   the define's buffers as for rato_cut_define_drone;  keep / keep_idle_count / n_keep_io: in and out as for rato_cut_solve.
// Returns the first non-OK status of a define / solve (RATO_ERANK, RATO_ESELECT: repeat with the per-iteration calls,
// whose Python loop recovers), *done = iterations completed.
extern "C" int rato_scp_run_drone(rato_cut_solver* s, const double* us0, int32_t iters, int32_t first_cvar, double tol,
                                  int32_t max_cuts, double final_cut_above, int32_t check_finite, float* us_host,
                                  float* us_dev, float* A22, float* part, double* sums_host, int32_t* keep,
                                  int32_t* keep_idle_count, int32_t* n_keep_io, double* us_hist, rato_scp_iter* rec,
                                  int32_t* done, void* stream) {
  if (!s || s->c.system != 0 || !us0 || iters < 0 || !us_host || !us_dev || !A22 || !part || !sums_host || !keep ||
      !keep_idle_count || !n_keep_io || !us_hist || !rec || !done)
    return RATO_EINVAL;
  const int S = s->c.S, nU = s->nU, n_c = 6;
  const double M = (double)s->c.M, inv_M = 1.0 / M;
  std::vector<double> us(us0, us0 + nU), final_du((size_t)n_c * nU), final_rhs(n_c), sol(nU);
  std::vector<int32_t> cut_slot(s->c.cap + 8);
  std::vector<double> cut_lambda(s->c.cap + 8);
  *done = 0;
  for (int it = 0; it < iters; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool cvar = it >= first_cvar;
    const int K = (cvar && s->c.recycle && S >= 2) ? *n_keep_io : 0;
    int rc = rato_cut_define_drone(s, us.data(), us_host, us_dev, A22, nullptr, 0, part, sums_host, nullptr, nullptr, keep, K,
                                   stream);
    if (rc != RATO_OK) return rc;
    if (check_finite)
      for (int i = 0; i < 6 * S + 6; ++i)
        if (!std::isfinite(sums_host[i])) return RATO_ENONFINITE;
    // the equality rows: mean of the final-state Jacobian (axis a: row a = position, row 3 + a = velocity; the axes decouple)
    // and of its right-hand side (drone_risk.py:271-273, :294-300)
    std::fill(final_du.begin(), final_du.end(), 0.0);
    for (int t = 0; t < S; ++t)
      for (int a = 0; a < 3; ++a) {
        final_du[(size_t)a * nU + t * 3 + a] = sums_host[t * 6 + a] * inv_M;
        final_du[(size_t)(3 + a) * nU + t * 3 + a] = sums_host[t * 6 + 3 + a] * inv_M;
      }
    for (int r = 0; r < n_c; ++r) final_rhs[r] = sums_host[6 * S + r] / M;
    rato_cut_result res = {};
    res.us = sol.data();
    res.cut_slot = cut_slot.data();
    res.cut_lambda = cut_lambda.data();
    res.cut_capacity = (int)cut_slot.size();
    rc = rato_cut_solve(s, final_du.data(), final_rhs.data(), n_c, us.data(), cvar ? 1 : 0, tol, max_cuts, final_cut_above,
                        check_finite, keep, keep_idle_count, n_keep_io, K > 0 ? 1 : 0, &res, stream);
    if (rc != RATO_OK) return rc;
    if (it == iters - 1) {   // the protocol's closing synchronisation, once: inside the last iteration's clock
      const hipError_t e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
      if (e != hipSuccess) return RATO_EHIP - (int)e;
    }
    const double total = seconds_since(t0);
    rato_scp_iter& r = rec[it];
    r.oracle_s = res.oracle_s;
    r.master_s = res.master_s;
    r.solve_s = res.oracle_s + res.master_s;
    r.define_s = total - r.solve_s;
    r.t_risk = res.t_risk;
    r.slack = res.slack;
    r.phi = res.phi;
    r.cuts = res.cuts;
    r.status = res.status;
    r.recycled = res.recycled;
    r.reserved = 0;
    memcpy(us_hist + (size_t)it * nU, sol.data(), sizeof(double) * nU);
    us = sol;
    *done = it + 1;
  }
  return RATO_OK;
}

extern "C" size_t rato_scp_iter_bytes(void) { return sizeof(rato_scp_iter); }

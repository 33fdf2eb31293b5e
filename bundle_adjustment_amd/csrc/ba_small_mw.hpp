// The window solver of ba_small.hpp on SEVERAL workgroups of one launch, for the reference's windows
// (BundleAdjuster(window_size=5), src/pipeline.py:39,99: five keyframes, a few hundred landmarks; up to eight keyframes).  One LM iteration of
// k_small_lm is ~62 us on one compute unit, every phase a handful of dependent round trips; here G <= 32 workgroups each
// own 64 consecutive landmarks:
//
//   C1   camera half of the normal equations over the workgroup's SLICE of every camera's observation list (the lists
//        are ascending in landmark index, so a landmark range is a contiguous slice: woff, computed by the host)
//   P1   EIGHT LANES PER LANDMARK, one observation per lane: Hpp / bp by a butterfly over the eight lanes, every lane
//        then holds the damped inverse = L L^T, y0, z; its observation's 6 x 3 block of V = W L goes into LDS --
//        the workgroup's columns of V never leave the compute unit
//   G    [V; z][V; z]^T of those columns on the matrix cores (v_mfma_f64_16x16x4_f64, operands from LDS)
//   ---- exchange 1: every workgroup publishes its partial reduced system (upper triangle + V z + bc + diag Hcc + max|bp|:
//        one ~560-word message), a counter barrier, every workgroup sums the G messages in workgroup order
//   S, Cholesky, both triangular solves: redundantly in every workgroup (wave 0; identical inputs, identical bits)
//   camera update (redundant), P2 back substitution for the own landmarks (V from LDS, L / y0 / Hpp / bp still in
//        registers), C3 trial cost over the own slices
//   ---- exchange 2: six step scalars per workgroup; every workgroup takes the same verdict
//
// Cross-workgroup data only moves through agent-scope atomic stores / loads (coherent whatever XCD a workgroup landed
// on; measured ~1.8 doubles / ns per reading workgroup, tools/microbench/grid_barrier.hip -- hence the compact message)
// and the barrier is one agent-scope counter (1.1 - 1.5 us).  All sums keep a fixed order (lanes, waves, workgroups).
//
// Residency.  The workgroups meet at a counter barrier, so all G of them have to be resident together; an ordinary launch
// does not promise that (another stream of the same process, a second handle or a profiler may hold compute units).
// The host checks the occupancy query before it ever chooses this kernel (ba_set_problem), and the kernel never depends
// on the answer: every spin is bounded by the 100 MHz wall clock (MW_SPIN_TICKS, 50 ms), a workgroup that is not
// served reports BA_ERR_HIP in the summary and leaves, and the host then re-solves the SAME window with the
// one-workgroup kernel (k_small_lm) in the same process (small_solve in ba_hip.hip).  The start point survives a failed
// launch: the cameras in global memory are only written by the final block of a successful solve, and every workgroup
// parks its landmarks' start positions in the (otherwise unused) y slots of the current point table before anything
// else, from where the host side restores them.
//
// Limits (checked by the host, else k_small_lm runs): Nc <= 8 (instantiated for two, three and four 16-row tiles of [V; z]:
// up to 5 / 7 / 8 cameras), 64 (G - 1) < Np <= 64 G with G <= 32, no landmark observed twice by one camera (so a track has at
// most Nc <= 8 observations: one per lane), single rank.
#pragma once
#include "ba_small.hpp"

namespace ba {

constexpr int MW_MAX_WG = 32;
constexpr int MW_PTS = 64;                         // landmarks per workgroup
constexpr int MW_LPP = 8;                          // lanes per landmark
constexpr int MW_THREADS = MW_PTS * MW_LPP;        // 512
constexpr int MW_WAVES = MW_THREADS / 64;
constexpr int MW_MAX_CAMS = SMALL_MAX_CAMS;        // 8
// The kernel is instantiated per number of 16-row tiles NT of [V; z]: 2 (up to 5 cameras: 30 + 1 rows), 3 (6 or 7 cameras),
// 4 (8 cameras: 48 rows and the z row alone in the fourth tile) -- what sizes the LDS images.
template <int NT> struct MwDim {
  static constexpr int NMAX = NT == 2 ? 30 : (NT == 3 ? 42 : 48);            // largest 6 Nc with 6 Nc + 1 <= 16 NT
  static constexpr int VR = 16 * NT;                                         // rows of the LDS image of V
  static constexpr int SLAB = VR * 16;                                       // doubles per 16-column slab
  static constexpr int NTILE = NT > 3 ? 9 : NT * (NT + 1) / 2;               // upper tiles (the z row's own square is not needed)
  static constexpr int SYRK_WAVES = NT > 3 ? 2 : 4;                          // waves that multiply (LDS for their partial tiles)
  static constexpr int MSG = NMAX * (NMAX + 1) / 2 + 3 * NMAX + 1;           // words of exchange 1
  static constexpr int SROW = NMAX + 3;                                      // row stride of [S | g]: odd, a column reads conflict-free
  static constexpr int POOL = (SYRK_WAVES * NTILE * 256 > NMAX * SROW) ? SYRK_WAVES * NTILE * 256 : NMAX * SROW;
  static constexpr int EROWS = (NMAX + 7) / 8;                               // elements a thread owns in the elimination
};
constexpr int MW_SLABS = 3 * MW_PTS / 16;          // 12 slabs of 16 columns
constexpr int MW_MSG = MwDim<4>::MSG + 3;          // stride of a workgroup's slot in the exchange buffer (any NT)
constexpr int MW_SCAL = 8;
constexpr int MW_MIN_PTS = 1;                     // (measured: no slower than k_small_lm even with one or two workgroups)
// A barrier gives up after this many ticks of the 100 MHz wall clock (s_memrealtime): 50 ms.  At most three barriers can
// time out one after the other in one launch (workgroups that were queued behind the ones that gave up arrive later and
// wait in turn), far below the 20 s the host waits for the result word.
constexpr long long MW_SPIN_TICKS = 5000000;

struct MwArgs {
  SmallArgs A;
  const int* woff;                 // [(G + 1)][MW_MAX_CAMS]: camera c's observations of landmarks >= 64 g start here (index into c_pt / c_uv)
  int G;
  double* slots;                   // [2][G][MW_MSG]   exchange 1, double-buffered by its own parity
  double* sslots;                  // [2][G][MW_SCAL]  exchange 2, double-buffered by its own parity
  int G_barrier;                   // workgroups every barrier waits for: G (tests pass G + 1 to provoke the time-out path)
  unsigned long long* ctr;         // barrier counter; every launch counts from (its sequence number << 32): no clearing between launches
};

__device__ inline double mw_allreduce8(double x) {     // sum over the 8 lanes of a landmark, the same bits in all of them
  x += dpp_f64<0xB1, 0xf>(x);      // quad_perm [1 0 3 2]
  x += dpp_f64<0x4E, 0xf>(x);      // quad_perm [2 3 0 1]
  x += dpp_f64<0x141, 0xf>(x);     // row_half_mirror: lane i <-> 7 - i of the 8-lane half row
  return x;
}

#define MW_STAMP(k) do { if (A.stamps && g == 0 && tid == 0 && s_it == 1) A.stamps[k] = (long long)wall_clock64(); } while (0)

template <int NT>
__global__ void __launch_bounds__(MW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_small_mw(MwArgs M) {
  using D = MwDim<NT>;
  constexpr int MW_N = D::NMAX, MW_SLAB = D::SLAB, MW_SYRK_WAVES = D::SYRK_WAVES, NTILE = D::NTILE, SROW = D::SROW;
  const SmallArgs& A = M.A;
  __shared__ __align__(16) double l_V[MW_SLABS * MW_SLAB];                 // [slab][row][16 columns]: 48 / 72 / 96 KB
  // one pool for two images that are never alive together: the multiplying waves' partial tiles (product -> message)
  // and the augmented [S | g] (after exchange 1 -> step)
  __shared__ __align__(16) double l_pool[D::POOL];
  double* const l_part = l_pool;
  double (*const l_S)[SROW] = (double (*)[SROW])l_pool;
  __shared__ double l_cs[2][MW_MAX_CAMS][CS];
  __shared__ double l_cam[2][MW_MAX_CAMS][6];
  __shared__ double l_Hccp[MW_MAX_CAMS][27];                               // this workgroup's share of Hcc | bc
  __shared__ double l_msg[D::MSG];
  __shared__ unsigned short l_ij[D::MSG];            // (i | j << 8) of message word t < nS + n: built once
  __shared__ double l_dc[MW_N], l_bc[MW_N], l_dH[MW_N];
  __shared__ double l_red[MW_WAVES * 8], l_tot[8], l_camred[MW_MAX_CAMS][4], l_wcost[MW_WAVES][2], l_sc[MW_SCAL];
  __shared__ double s_lambda, s_cost, s_sse, s_nu;
  __shared__ int s_cur, s_stop, s_it, s_acc, s_status, s_ok;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = M.G;
  const int Nc = A.n_cams, Np = A.n_pts, n = 6 * Nc;
  const bool robust = A.robust != 0;
  const int nS = n * (n + 1) / 2;                    // message: [0, nS) upper triangle | V z (n) | bc (n) | diag Hcc (n) | max |bp|
  const int msg_len = nS + 3 * n + 1;
  const int pl = tid / MW_LPP, sub = tid % MW_LPP;   // landmark of this thread inside the workgroup, lane inside the landmark
  const int p = g * MW_PTS + pl;
  const bool have_p = p < Np;
  // Each exchange buffer has a parity of its own: a workgroup that has passed the barrier of use u of a buffer may write
  // its slot for use u + 1 (the other half) while a slower one still reads use u; it cannot reach use u + 2 (the same
  // half again) before everybody has arrived at use u + 1, i.e. has finished reading use u.  No assumption about how the
  // two kinds of exchange alternate.
  int xpar1 = 0, xpar2 = 0;
  // the counter of this launch starts at (sequence number << 32): whichever workgroup comes first raises it there (an
  // idempotent maximum: later ones find it at or above), so no launch has to clear what the previous one left
  const unsigned long long ctr_base = (unsigned long long)A.seq << 32;
  if (tid == 0) __hip_atomic_fetch_max(M.ctr, ctr_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long bar = 0;                        // barriers passed

  // counter barrier: everything this workgroup stored (agent-scope atomic stores) is visible to whoever passes it
  auto barrier = [&]() -> bool {
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(M.ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long target = ctr_base + (bar + 1) * (unsigned long long)M.G_barrier;
      int spins = 0, ok = 1;
      long long t_start = 0;
      while (__hip_atomic_load(M.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63) == 0) {                       // the clock is read once per 64 polls
          const long long now = (long long)wall_clock64();
          if (t_start == 0) t_start = now;
          else if (now - t_start > MW_SPIN_TICKS) { ok = 0; break; }
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      s_ok = ok;
    }
    __syncthreads();
    ++bar;
    return s_ok != 0;
  };
  // every workgroup's `len` words (src, LDS) summed in workgroup order into dst (LDS); word `max_at` (if >= 0): the maximum
  auto exchange = [&](double* slots, int& xpar, int stride, const double* src, int len, double* dst, int max_at) -> bool {
    double* mine = slots + ((size_t)xpar * G + g) * stride;
    for (int i = tid; i < len; i += MW_THREADS) __hip_atomic_store(mine + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!barrier()) return false;
    const double* all = slots + (size_t)xpar * G * stride;
    for (int i = tid; i < len; i += MW_THREADS) {
      double a = 0.0;                                    // (sums start from +0: every partial of a maximum is >= 0 as well)
      for (int q0 = 0; q0 < G; q0 += 8) {                // eight loads in flight, added in workgroup order
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          v[q] = q0 + q < G ? __hip_atomic_load(all + (size_t)(q0 + q) * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        if (i == max_at) {
#pragma unroll
          for (int q = 0; q < 8; ++q) a = nanmax(a, v[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) a += v[q];
        }
      }
      dst[i] = a;
    }
    xpar ^= 1;
    __syncthreads();
    return true;
  };
  auto give_up = [&]() {                              // a barrier timed out: report, never hang
    if (g == 0 && tid == 0) {
      A.summary->status = BA_ERR_HIP; A.summary->iterations = -1; A.summary->accepted = s_acc;
      *A.cur_out = A.cur;
      publish_flag(A.host_flag, A.seq, 1);
    }
  };

  // ---- start point into LDS (every workgroup its own copy), the LDS image of V cleared once: entries of cameras that do
  //      not see a landmark are never written
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) l_cam[A.cur][tid][q] = A.cams[A.cur][6 * tid + q];
    camera_state(&l_cam[A.cur][tid][0], &l_cs[A.cur][tid][0]);
  }
  for (int i = tid; i < MW_SLABS * MW_SLAB; i += MW_THREADS) l_V[i] = 0.0;
  for (int t = tid; t < nS + n; t += MW_THREADS) {      // row-major upper triangle, then column n (V z)
    int i, j;
    if (t < nS) {
      i = 0;
      int base = 0;
      while (base + (n - i) <= t) { base += n - i; ++i; }
      j = i + (t - base);
    } else { i = t - nS; j = n; }
    l_ij[t] = (unsigned short)(i | (j << 8));
  }
  if (tid == 0) { s_cur = A.cur; s_lambda = A.lambda0; s_nu = 2.0; s_stop = 0; s_it = 0; s_acc = 0; s_status = 0; }
  // the start positions of the own landmarks, parked in the y slots of the current point table (this kernel keeps y in
  // registers and never reads or writes those slots otherwise): what the host restores the window from if a barrier
  // times out after steps have already been taken (k_small_restore)
  if (have_p && sub == 0) {
    const double4 X = *(const double4*)(A.ptab[A.cur] + PT * (size_t)p);
    double* o = A.ptab[A.cur] + PT * (size_t)p + 4;
    o[0] = X.x; o[1] = X.y; o[2] = X.z;
  }
  __syncthreads();

  // cost over this workgroup's slices at parameter set w: wave c < Nc walks camera c's slice -> l_sc[4] = sse, l_sc[5] = sum rho
  auto slice_cost = [&](int w) {
    if (wv < Nc) {
      const int c = wv;
      const double* cs = &l_cs[w][c][0];
      double acc[2] = {0.0, 0.0};
      for (int i = M.woff[g * MW_MAX_CAMS + c] + lane; i < M.woff[(g + 1) * MW_MAX_CAMS + c]; i += 64) {
        const double4 X = *(const double4*)(A.ptab[w] + PT * (size_t)A.c_pt[i]);
        const double2 uv = A.c_uv[i];
        double xh, yh;
        obs_project(cs, X.x, X.y, X.z, xh, yh);
        const double ru = uv.x - (xh * A.fx + A.cx), rv = uv.y - (yh * A.fy + A.cy);
        acc[0] += ru * ru + rv * rv;
        if (robust) { double t0, t1, ww; huber(ru, A.hub_c, t0, ww); huber(rv, A.hub_c, t1, ww); acc[1] += t0 + t1; }
      }
      if (!robust) acc[1] = acc[0];
      acc[0] = wave_total_dpp(acc[0]); acc[1] = wave_total_dpp(acc[1]);
      if (lane == 0) { l_wcost[wv][0] = acc[0]; l_wcost[wv][1] = acc[1]; }
    }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int c = 0; c < Nc; ++c) { a += l_wcost[c][0]; b += l_wcost[c][1]; }
      l_sc[4] = a; l_sc[5] = b;
    }
    __syncthreads();
  };

  slice_cost(A.cur);
  if (tid < 4) l_sc[tid] = 0.0;
  __syncthreads();
  if (!exchange(M.sslots, xpar2, MW_SCAL, l_sc, 6, l_tot, -1)) { give_up(); return; }
  if (tid == 0) {
    s_sse = l_tot[4]; s_cost = 0.5 * l_tot[5];
    if (g == 0) { A.summary->initial_sse = s_sse; A.summary->initial_cost = s_cost; }
    if (!isfinite(s_cost)) { s_stop = 1; s_status = -4; }          // BA_ERR_NUMERIC
    if (A.max_iters <= 0) s_stop = 1;
  }
  __syncthreads();

  // the landmark's observation of this lane (static over the solve: at most one per lane)
  int my_c = -1;
  double2 my_uv = make_double2(0.0, 0.0);
  if (have_p) {
    const int j = A.pt_off[p] + sub;
    if (j < A.pt_off[p + 1]) { my_c = A.p_cam[j]; my_uv = A.p_uv[j]; }
  }
  // column offsets of this landmark in the LDS image: component q sits in column q * MW_PTS + pl
  const int col0 = pl, col1 = MW_PTS + pl, col2 = 2 * MW_PTS + pl;
  auto vofs = [](int col) { return (col >> 4) * MW_SLAB + (col & 15); };      // + 16 * row
  const int o0 = vofs(col0), o1 = vofs(col1), o2 = vofs(col2);

  bool need_lin = true;
  double hp[6] = {0, 0, 0, 0, 0, 0}, bp3[3] = {0, 0, 0};       // undamped Hpp, bp of this landmark (all 8 lanes)
  double Lf[6] = {0, 0, 0, 0, 0, 0}, y0[3] = {0, 0, 0};
  while (!s_stop) {
    const int cur = s_cur, tr = 1 - cur;
    const double lambda = s_lambda;
    MW_STAMP(0);
    if (need_lin) {
      // ---- C1: camera half over the workgroup's slices (post-M rows, as k_small_lm)
      if (wv < Nc) {
        const int c = wv;
        const double* cs = &l_cs[cur][c][0];
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        if (c != A.fixed_cam) {
          for (int i = M.woff[g * MW_MAX_CAMS + c] + lane; i < M.woff[(g + 1) * MW_MAX_CAMS + c]; i += 64) {
            const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)A.c_pt[i]);
            const double2 uv = A.c_uv[i];
            Geom gm;
            obs_geom(cs, X.x, X.y, X.z, A.fx, A.fy, gm);
            const double ru = uv.x - (gm.xh * A.fx + A.cx), rv = uv.y - (gm.yh * A.fy + A.cy);
            double w0 = 1.0, w1 = 1.0;
            if (robust) { double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1); }
            double c0[6], c1[6];
            small_cam_rows(cs, gm, X.x, X.y, X.z, c0, c1);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const double wa0 = w0 * c0[a], wa1 = w1 * c1[a];
#pragma unroll
              for (int b = a; b < 6; ++b) acc[U6(a, b)] += wa0 * c0[b] + wa1 * c1[b];
              acc[21 + a] += wa0 * ru + wa1 * rv;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = wave_total_dpp(acc[q]);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 27; ++q) l_Hccp[c][q] = acc[q];
        }
      }
    }
    MW_STAMP(1);
    // ---- P1: eight lanes per landmark
    double gmp = 0.0;
    {
      double4 X = make_double4(0, 0, 0, 0);
      if (have_p) X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
      Geom gm;
      double w0 = 1.0, w1 = 1.0, ru = 0.0, rv = 0.0;
      double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (my_c >= 0) {
        obs_geom(&l_cs[cur][my_c][0], X.x, X.y, X.z, A.fx, A.fy, gm);
        ru = my_uv.x - (gm.xh * A.fx + A.cx); rv = my_uv.y - (gm.yh * A.fy + A.cy);
        if (robust) { double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1); }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double wa0 = w0 * gm.P[q], wa1 = w1 * gm.P[3 + q];
#pragma unroll
          for (int r = q; r < 3; ++r) a[U3(q, r)] += wa0 * gm.P[r] + wa1 * gm.P[3 + r];
          a[6 + q] -= wa0 * ru + wa1 * rv;
        }
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) a[q] = mw_allreduce8(a[q]);
#pragma unroll
      for (int q = 0; q < 6; ++q) hp[q] = a[q];
#pragma unroll
      for (int q = 0; q < 3; ++q) { bp3[q] = a[6 + q]; if (have_p) gmp = nanmax(gmp, fabs(a[6 + q])); }
      double h[6], hinv[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) h[q] = hp[q];
      h[0] += lambda * fmax(h[0], DIAG_FLOOR);
      h[3] += lambda * fmax(h[3], DIAG_FLOOR);
      h[5] += lambda * fmax(h[5], DIAG_FLOOR);
      sym3_inverse(h, hinv);
      sym3_mul(hinv, bp3, y0);
      const double l00 = sqrt(fmax(hinv[0], 1e-300));
      const double l10 = hinv[1] / l00, l20 = hinv[2] / l00;
      const double l11 = sqrt(fmax(hinv[3] - l10 * l10, 1e-300));
      const double l21 = (hinv[4] - l20 * l10) / l11;
      const double l22 = sqrt(fmax(hinv[5] - l20 * l20 - l21 * l21, 1e-300));
      Lf[0] = l00; Lf[1] = l10; Lf[2] = l11; Lf[3] = l20; Lf[4] = l21; Lf[5] = l22;
      if (have_p && sub == 0) {                          // z = L^T bp: row n of the image
        l_V[o0 + 16 * n] = l00 * bp3[0] + l10 * bp3[1] + l20 * bp3[2];
        l_V[o1 + 16 * n] = l11 * bp3[1] + l21 * bp3[2];
        l_V[o2 + 16 * n] = l22 * bp3[2];
      }
      if (my_c >= 0 && my_c != A.fixed_cam) {
        double c0[6], c1[6];
        small_cam_rows(&l_cs[cur][my_c][0], gm, X.x, X.y, X.z, c0, c1);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double wa0 = -w0 * c0[i], wa1 = -w1 * c1[i];
          const double W0 = wa0 * gm.P[0] + wa1 * gm.P[3], W1 = wa0 * gm.P[1] + wa1 * gm.P[4], W2 = wa0 * gm.P[2] + wa1 * gm.P[5];
          const int r = 16 * (6 * my_c + i);
          l_V[o0 + r] = W0 * l00 + W1 * l10 + W2 * l20;
          l_V[o1 + r] = W1 * l11 + W2 * l21;
          l_V[o2 + r] = W2 * l22;
        }
      }
    }
    gmp = wave_nanmax(gmp);
    if (lane == 0) l_red[wv] = gmp;
    __syncthreads();                                     // V image, l_Hccp, l_red complete
    MW_STAMP(2);
    // ---- G: [V; z][V; z]^T of the workgroup's 192 columns; the multiplying waves take slabs w, w + W, ...
    if (wv < MW_SYRK_WAVES) {
      small_d4 acc[NTILE];
#pragma unroll
      for (int q = 0; q < NTILE; ++q) acc[q] = (small_d4){0.0, 0.0, 0.0, 0.0};
      const double* vrow = l_V + (lane & 15) * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int u = 0; u < MW_SLABS / MW_SYRK_WAVES; ++u) {
        const int sl = wv + MW_SYRK_WAVES * u;
        small_d4 a[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *(const small_d4*)(vrow + MW_SLAB * sl + 256 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int tj = 0; tj < NT; ++tj) {
#pragma unroll
            for (int ti = 0; ti <= tj; ++ti) {
              if (NT > 3 && ti == 3) continue;
              acc[tj * (tj + 1) / 2 + ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti][e], a[tj][e], acc[tj * (tj + 1) / 2 + ti], 0, 0, 0);
            }
          }
        }
      }
      small_d4* out = (small_d4*)l_part + (size_t)(wv * NTILE) * 64 + lane;
#pragma unroll
      for (int q = 0; q < NTILE; ++q) out[q * 64] = acc[q];
    }
    __syncthreads();
    // ---- this workgroup's message: element (i, j), i <= j, of blockdiag(Hcc_g) - (V V^T)_g | (V z)_g | bc_g | diag Hcc_g | max |bp|
    //      tile (ti <= tj) number tj (tj + 1) / 2 + ti; element (ri, rj) of a tile: lane rj + 16 (ri % 4), register ri / 4
    for (int t = tid; t < msg_len; t += MW_THREADS) {
      double out;
      if (t < nS + n) {
        const int i = l_ij[t] & 0xff, j = l_ij[t] >> 8;
        const int ti = i >> 4, tj = j >> 4, ri = i & 15, rj = j & 15;
        const double* ps = l_part + (size_t)(tj * (tj + 1) / 2 + ti) * 256 + 4 * (rj + 16 * (ri & 3)) + (ri >> 2);
        double vv = 0.0;
#pragma unroll
        for (int w = 0; w < MW_SYRK_WAVES; ++w) vv += ps[(size_t)w * NTILE * 256];
        out = (j == n) ? vv : -vv;
        if (j < n && i / 6 == j / 6) out += l_Hccp[i / 6][U6(i % 6, j % 6)];
      } else if (t < nS + 2 * n) {
        const int i = t - nS - n;
        out = l_Hccp[i / 6][21 + i % 6];
      } else if (t < nS + 3 * n) {
        const int i = t - nS - 2 * n;
        out = l_Hccp[i / 6][U6(i % 6, i % 6)];
      } else {
        double m = 0.0;
        for (int w = 0; w < MW_WAVES; ++w) m = nanmax(m, l_red[w]);
        out = m;
      }
      l_msg[t] = out;
    }
    __syncthreads();
    MW_STAMP(3);
    // ---- exchange 1
    if (!exchange(M.slots, xpar1, MW_MSG, l_msg, msg_len, l_msg, msg_len - 1)) { give_up(); return; }
    MW_STAMP(4);
    // ---- S, g (every workgroup the same): damping from the summed diagonal; fixed camera: identity rows / columns
    for (int t = tid; t < nS; t += MW_THREADS) {         // both triangles from the message's upper one
      const int i = l_ij[t] & 0xff, j = l_ij[t] >> 8, ci = i / 6, cj = j / 6;
      double v = l_msg[t];
      if (i == j) v += lambda * fmax(l_msg[nS + 2 * n + i], DIAG_FLOOR);
      if (ci == A.fixed_cam || cj == A.fixed_cam) v = (i == j) ? 1.0 : 0.0;
      l_S[i][j] = v;
      l_S[j][i] = v;
    }
    if (tid < n) {
      const int i = tid, ci = i / 6;
      l_bc[i] = l_msg[nS + n + i];
      l_dH[i] = l_msg[nS + 2 * n + i];
      l_S[i][n] = (ci == A.fixed_cam) ? 0.0 : -(l_msg[nS + n + i] - l_msg[nS + i]);     // right-hand side g: the augmented column
    }
    __syncthreads();
    if (need_lin && wv == 0) {                           // max |gradient| = max(max |bp| over the workgroups, max |bc|)
      double m = lane < n ? fabs(l_bc[lane]) : 0.0;
      if (lane == 0) m = nanmax(m, l_msg[msg_len - 1]);
      m = wave_nanmax(m);
      if (lane == 0) {
        if (!isfinite(m)) { s_stop = 1; s_status = -4; }
        else if (A.gtol > 0 && m <= A.gtol) { s_stop = 1; s_status = 3; }
      }
    }
    __syncthreads();
    if (s_stop) break;
    MW_STAMP(5);
    // ---- S dc = g.  Gauss-Jordan elimination on the augmented [S | g] in LDS, ALL waves: step k subtracts
    //      (S[i][k] / S[k][k]) x row k from every OTHER row i -- (n - 1)(n - k) independent updates, one or two per
    //      thread, a workgroup barrier per step; what is left is a diagonal system.  (k_small_lm factors in one wave, lane = row, 3 j instructions for
    //      column j: 10 us of its 62 us iteration; with the rest of the iteration spread over G workgroups that would
    //      be a third of the time here.)  S is symmetric positive definite: no pivoting, pivots floored like the
    //      Cholesky's diagonal.
    MW_STAMP(12);
    {
      // a thread owns column ja = tid % 64 (column n = the right-hand side) of rows tid / 64 + 8 r, r = 0 .. EROWS - 1, for
      // the whole elimination, kept in registers and stored after every step (the next step's pivot row and column are
      // read by everybody).  Every read of a step is unconditional (clamped addresses) and issued before the first use:
      // one LDS round trip per step.
      constexpr int ER = D::EROWS;
      const int ja = tid & 63, i0 = tid >> 6;
      const bool col_ok = ja <= n;
      const int jc = col_ok ? ja : n;
      int ic[ER];
      double va[ER];
#pragma unroll
      for (int r = 0; r < ER; ++r) { const int i = i0 + 8 * r; ic[r] = i < n ? i : n - 1; va[r] = l_S[ic[r]][jc]; }
      for (int k = 0; k < n; ++k) {
        if (k / 6 == A.fixed_cam) continue;               // identity rows and columns of the held camera: nothing to eliminate
        const double piv0 = l_S[k][k], u = l_S[k][jc];
        double lk[ER];
#pragma unroll
        for (int r = 0; r < ER; ++r) lk[r] = l_S[ic[r]][k];
        const double piv = fmax(piv0, DIAG_FLOOR);
        double rp = __builtin_amdgcn_rcp(piv);
        rp = rp * (2.0 - piv * rp);
        rp = rp * (2.0 - piv * rp);
        const bool act = col_ok && ja > k;
#pragma unroll
        for (int r = 0; r < ER; ++r) {
          const int i = i0 + 8 * r;
          if (act && i != k && i < n) { va[r] -= (lk[r] * rp) * u; l_S[i][ja] = va[r]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: nothing else is in flight here
      }
    }
    MW_STAMP(13);
    // every row is reduced to its diagonal element and the right-hand side (Gauss-Jordan: the rows ABOVE a pivot were
    // swept in the same steps by threads that would have idled)
    if (tid < n) {
      const double dg = fmax(l_S[tid][tid], DIAG_FLOOR);
      l_dc[tid] = l_S[tid][n] / dg;
    }
    __syncthreads();
    MW_STAMP(6);
    // ---- camera update + camera-side scalars (every workgroup the same)
    if (tid < Nc) {
      const int c = tid;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int q = 0; q < 6; ++q) {
        const double d = (c == A.fixed_cam) ? 0.0 : l_dc[6 * c + q];
        const double xq = l_cam[cur][c][q];
        l_cam[tr][c][q] = xq + d;
        a0 += l_bc[6 * c + q] * d;
        a1 += fmax(l_dH[6 * c + q], DIAG_FLOOR) * d * d;
        a2 += d * d;
        a3 += xq * xq;
      }
      l_camred[c][0] = a0; l_camred[c][1] = a1; l_camred[c][2] = a2; l_camred[c][3] = a3;
      camera_state(&l_cam[tr][c][0], &l_cs[tr][c][0]);
    }
    __syncthreads();
    // ---- P2: back substitution for the own landmarks, lane c of a landmark takes camera c's rows of V (LDS)
    double ps[4] = {0, 0, 0, 0};
    {
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      if (sub < Nc) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int r = 16 * (6 * sub + i);
          const double d = l_dc[6 * sub + i];
          t0 += l_V[o0 + r] * d; t1 += l_V[o1 + r] * d; t2 += l_V[o2 + r] * d;
        }
      }
      t0 = mw_allreduce8(t0); t1 = mw_allreduce8(t1); t2 = mw_allreduce8(t2);
      if (have_p && sub == 0) {
        const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
        const double d0 = -(y0[0] + Lf[0] * t0);
        const double d1 = -(y0[1] + Lf[1] * t0 + Lf[2] * t1);
        const double d2 = -(y0[2] + Lf[3] * t0 + Lf[4] * t1 + Lf[5] * t2);
        double* o = A.ptab[tr] + PT * (size_t)p;
        o[0] = X.x + d0; o[1] = X.y + d1; o[2] = X.z + d2;
        const double D0 = fmax(hp[0], DIAG_FLOOR), D1 = fmax(hp[3], DIAG_FLOOR), D2 = fmax(hp[5], DIAG_FLOOR);
        ps[0] = bp3[0] * d0 + bp3[1] * d1 + bp3[2] * d2;
        ps[1] = D0 * d0 * d0 + D1 * d1 * d1 + D2 * d2 * d2;
        ps[2] = d0 * d0 + d1 * d1 + d2 * d2;
        ps[3] = X.x * X.x + X.y * X.y + X.z * X.z;
      }
    }
    small_block_sum<4>(ps, l_red, l_sc);                 // l_sc[0 .. 3]
    __threadfence_block();
    __syncthreads();                                     // trial points visible to the slice cost pass
    MW_STAMP(7);
    slice_cost(tr);                                      // l_sc[4], l_sc[5]
    MW_STAMP(8);
    // ---- exchange 2: the step's scalars; then the verdict, the same in every workgroup
    if (!exchange(M.sslots, xpar2, MW_SCAL, l_sc, 6, l_tot, -1)) { give_up(); return; }
    MW_STAMP(9);
    if (tid == 0) {
      double gTd = l_tot[0], dDd = l_tot[1], step2 = l_tot[2], x2 = l_tot[3];
      for (int c = 0; c < Nc; ++c) { gTd += l_camred[c][0]; dDd += l_camred[c][1]; step2 += l_camred[c][2]; x2 += l_camred[c][3]; }
      const double model = 0.5 * (lambda * dDd - gTd);
      const double sse_new = l_tot[4], cost_new = 0.5 * l_tot[5];
      const double rho = (model > 0.0 && isfinite(cost_new)) ? (s_cost - cost_new) / model : -1.0;
      const int it = ++s_it;
      ba_iter_record rec;
      rec.iteration = it; rec.accepted = (rho > 0.0 && isfinite(cost_new)) ? 1 : 0; rec.pcg_iterations = 0; rec.reserved = 0;
      rec.cost = s_cost; rec.cost_trial = cost_new; rec.sse_trial = sse_new; rec.lambda = lambda; rec.gain_ratio = rho;
      rec.step_norm = sqrt(step2); rec.seconds = 0.0;
      if (g == 0) A.trace[it - 1] = rec;
      int stop = 0;
      if (rec.accepted) {
        const double dcost = s_cost - cost_new;
        s_cur = tr;
        s_cost = cost_new; s_sse = sse_new;
        ++s_acc;
        const double t = 2.0 * rho - 1.0;
        s_lambda = fmax(lambda * fmax(1.0 / 3.0, 1.0 - t * t * t), 1e-12);
        s_nu = 2.0;
        if (dcost <= A.ftol * cost_new) { s_status = 1; stop = 1; }
      } else {
        if (!isfinite(cost_new) && lambda >= 1e12) { s_status = -4; stop = 1; }
        s_lambda = fmin(lambda * s_nu, 1e12);
        s_nu *= 2.0;
      }
      if (!stop && sqrt(step2) <= A.xtol * (A.xtol + sqrt(x2))) { s_status = 2; stop = 1; }
      if (!stop && it >= A.max_iters) { s_status = 0; stop = 1; }
      s_stop = stop;
      l_tot[7] = rec.accepted ? 1.0 : 0.0;
    }
    __syncthreads();
    need_lin = l_tot[7] != 0.0;                          // a rejected step keeps the camera half, re-damps the rest
    __syncthreads();
  }
  // ---- results (workgroup 0): cameras of the accepted set back to global, summary
  __syncthreads();
  if (g != 0) return;
  const int fin = s_cur;
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) A.cams[fin][6 * tid + q] = l_cam[fin][tid][q];
    for (int q = 0; q < CS; ++q) A.cs[fin][CS * tid + q] = l_cs[fin][tid][q];
    for (int q = 0; q < 12; ++q) A.camA[fin][TA * tid + q] = l_cs[fin][tid][q];
  }
  if (tid == 0) {
    A.summary->iterations = s_it; A.summary->accepted = s_acc; A.summary->pcg_iterations = 0; A.summary->status = s_status;
    A.summary->final_sse = s_sse; A.summary->final_cost = s_cost; A.summary->final_lambda = s_lambda;
    *A.cur_out = fin;
    publish_flag(A.host_flag, A.seq, 1);
  }
}

// After a k_small_mw launch that gave up at a barrier: the start positions back from the y slots into the X slots of the
// point table the solve started from (the cameras of that set were never written).
__global__ void k_small_restore(double* __restrict__ ptab, int n_pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  double* o = ptab + PT * (size_t)p;
  o[0] = o[4]; o[1] = o[5]; o[2] = o[6];
}

}  // namespace ba

// libba_hip.so: host side of the MI355X bundle-adjustment solve step and its C ABI
// (include/ba_hip.h).  One handle owns one GPU, one stream, every device buffer of one
// problem, and (multi-rank jobs) one RCCL communicator loaded with dlopen.
//
// LM / Schur / PCG loop (ba_solve), replacing scipy.optimize.least_squares at
// src/bundle_adjuster.py:170-174 of the reference:
//   linearise (K2a camera pass, K2b point pass)            [all-reduce Hcc | bc]
//   repeat with damping lambda until a step is accepted:
//     K3 damp + invert point blocks, y0 = Hpp^-1 bp;  rhs pass (K4b on y0) [all-reduce]
//     preconditioner (block-Jacobi of Hcc, or of the Schur diagonal)      [all-reduce]
//     PCG on S dc = g: per iteration K4a (by point), K4b (by camera) [all-reduce], K5
//     K7a camera update, K6 back substitution, K1 cost at the trial point  [all-reduce]
//     gain ratio -> accept (swap buffers) / reject (raise lambda)
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_kernels.hpp"
#include "ba_triangulate.hpp"
#include "ba_coarse.hpp"
#include "ba_small.hpp"
#include "ba_small_mw.hpp"
#include "ba_setup.hpp"

using namespace ba;

// ---------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHECK(expr)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(BA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* ba_last_error(void) { return g_err.c_str(); }

static const char* kKernelNames[BA_PROFILE_SLOTS] = {
    "cam_prepare", "residual_cam", "linearize_cam", "linearize_pt", "point_invert", "schur_pt",
    "schur_cam", "pcg_step", "precond", "backsub_pt", "misc", "allreduce", "schur_pt_then_backsub", "", "", ""};
extern "C" const char* ba_kernel_name(int slot) {
  return (slot >= 0 && slot < BA_PROFILE_SLOTS) ? kKernelNames[slot] : "";
}

// ------------------------------------------------------------------------------ RCCL
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static int load_rccl() {
  if (g_rccl.lib) return BA_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(BA_ERR_COMM, "dlopen(librccl.so) failed: %s", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail(BA_ERR_COMM, "librccl.so lacks an expected symbol");
  g_rccl.lib = lib;
  return BA_OK;
}

// ----------------------------------------------------------------------------- roctx
// Optional named ranges around the phases of an LM iteration (BA_ROCTX=1): they show up in rocprofv3
// --marker-trace timelines.  libroctx64 is loaded on demand; without it, or without the variable, nothing happens.
struct Roctx {
  bool tried = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
static Roctx g_roctx;
static void roctx_load() {
  if (g_roctx.tried) return;
  g_roctx.tried = true;
  if (!getenv("BA_ROCTX")) return;
  void* lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return;
  g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
  g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
  if (!g_roctx.push || !g_roctx.pop) { g_roctx.push = nullptr; g_roctx.pop = nullptr; }
}
struct Range {        // host-side range: covers the launches (and host waits) of one phase
  bool on;
  explicit Range(const char* name) : on(g_roctx.push != nullptr) { if (on) g_roctx.push(name); }
  ~Range() { if (on) g_roctx.pop(); }
};

// ---------------------------------------------------------------------------- handle
template <typename T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  // a NEW allocation comes back zero-filled (once: callers do not clear buffers on every use)
  hipError_t alloc(size_t count) {
    if (p && n >= count && count > 0) return hipSuccess;
    release();
    if (count == 0) return hipSuccess;
    // small buffers get headroom: consecutive sliding windows differ a little in size, and a hipFree + hipMalloc per
    // buffer on every growth costs more than the solve of such a window
    if (count < ((size_t)1 << 20)) count += count / 2 + 64;
    n = count;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) { p = nullptr; n = 0; return e; }
    // hipMemset of device memory may return before the fill has run (it is queued on the null stream, which the handle's
    // non-blocking stream does not wait for): drain it, or the zeros can land on top of the first upload.  Allocations are rare.
    e = hipMemset(p, 0, count * sizeof(T));
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(nullptr);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct ba_handle {
  int device = 0;
  int n_cu = 256;              // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int xcd_ranges = 1;          // point-pass ranges grouped per XCD (BA_XCD_RANGES=0 turns it off; speed only)
  int model = 0;               // camera model of the running call: 0 = the reference's pinhole (Pinhole), 1 = BAL 9-parameter
                               // (BalCam, ba_models.hpp); every kernel of the loop is instantiated for both
  int lanes = LPP;             // lanes per point in the point passes: 2, or 4 / 8 / 16 for every point of a smaller problem
  int cam_band = 0;            // camera passes: XCD x takes camera range x (1) or partition x of every camera (0); see group_of_block
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second stream: test hook ba_debug_occupy only
  bool have_problem = false, have_params = false, linearized = false;
  int lin_robust = 0;
  double lin_fscale = 1.0;
  int Nc = 0, Np = 0, Nobs = 0, fixed = -1;
  double K4[4] = {1, 1, 0, 0};
  // observation lists (camera order, point order)
  DBuf<int> offk, c_pt, c_orig, pt_off, p_cam, slot, long_pts;
  DBuf<int> c_ptf[2], p_camf[2];  // index streams with the "weights are not (1, 1)" flag (robust loss; c_ptf pairs with c_w, p_camf with p_w)
  int long_thr = 16;           // tracks longer than this get a DPP row each (set in ba_set_problem)
  int n_long = 0, nblkL = 0;   // points with more than LONG_TRACK observations: one DPP row each, own launch
  int long_spb = PT_THREADS / LPP_LONG;   // long-track points per workgroup (a multiple of one round's 64)
  DBuf<int2> blk_win;          // per point-pass workgroup: first camera and number of cameras its points see
  bool uv_f32 = false;         // c_uv / p_uv hold float2 (every pixel of the problem is a float32 value: UvArr, ba_kernels.hpp)
  DBuf<double2> c_uv, p_uv, c_w[2], p_w[2];   // both halves of the linearisation are double-buffered: the next one is
                                              // computed speculatively at the trial point while the host decides
                                              // (camera half: c_w, c_ptf, partL [lb]; point half: p_w, p_camf, Hpp, bp, Hppinv, y0 [pb])
  // parameters (current / trial): cameras, camera state, point table
  DBuf<double> cams[2], cs[2], ptab[2], stage;
  int cur = 0;
  // camera table of the point passes, normal equations
  DBuf<double> camA[2], HccBc, Hpp[2], bp[2], Hppinv[2], y0[2], Hccd, Minv;
  int pb = 0;                  // which point-half buffer set holds the current linearisation
  // partial sums
  DBuf<double> partR, partL[2], part6, partE, partA, partB, partC, partV;
  DBuf<double> linmsg[2];            // multi-rank: a linearisation's camera half, folded, behind 8 header words (k_fold_lin)
  DBuf<double> sysmsg;               // multi-rank: the damped system's message [u.y, 0 | part6 | partE | world maxima] (k_fold_msg)
  DBuf<double> partG[2], partGc;   // per-workgroup max |bp| (point half, double-buffered like it) and max |bc| (k_pcg_setup): the gtol test
  int lb = 0;                  // which c_w / partL buffer holds the current linearisation
  // PCG vectors, comm buffers (multi-rank), scalars
  DBuf<double> gvec, x, r, p, s, z, vin, vx, scal, rbuf, gather;
  DBuf<PcgState> st;
  DBuf<double> tri;            // staging of ba_triangulate
  char* h_up = nullptr;        // pinned staging of ba_set_problem's index uploads (small problems: one memcpy + async copies
  size_t h_up_cap = 0;         //  instead of a blocking staged copy per array; grow-only)
  char* h_small = nullptr;     // k_small_lm's results, host-mapped: ba_summary | int cur | trace records
  char* d_small_host = nullptr;
  size_t h_small_bytes = 0;
  long long small_seq = 0;     // sequence number of h_flags[4], which k_small_lm publishes when its results are written
  DBuf<double> intr[2];        // per-camera intrinsics (f, k1, k2) of the BAL model, current / trial like cams[]
  DBuf<double> small_V, small_gS;   // k_small_lm: V = W L (49 x 3 Np_pad, zero where unwritten), per-wave partial V V^T
  int small_np_pad = -1;
  char* h_par = nullptr; size_t h_par_cap = 0;   // pinned bounce buffer of ba_set_params / ba_get_params (window-sized problems)
  hipEvent_t par_event = nullptr; bool par_pending = false;   // the last upload from it (ba_set_params returns without waiting)
  // k_small_mw (ba_small_mw.hpp): the window solver on mw_G workgroups; mw_ok: this problem fits its limits
  DBuf<int> mw_woff; DBuf<double> mw_buf; bool mw_ok = false; int mw_G = 0;
  int mw_resident[3] = {-1, -1, -1};   // workgroups of k_small_mw<2 / 3 / 4> the device holds at once (occupancy query, once per handle)
  long long stats[BA_STAT_COUNT] = {0};   // ba_get_stat
  DBuf<int> setup_i;                      // scratch of the device build of ba_set_problem (ba_setup.hpp)
  DBuf<char> up_dev;                      // device copy of the pinned upload arena of a window-sized problem (k_unpack_problem)
  hipEvent_t up_event = nullptr; bool up_pending = false;   // ... and its last copy (ba_set_problem returns without waiting for it)
  char* h_setup = nullptr;                // pinned: what that build reads back (track-length histogram, statistics, windows)
  int setup_path = 0;                     // how the current problem's layout was built: 0 host, 1 device
  DBuf<double> stat2;                     // multi-rank: the band statistic (span sum, tracks) summed over the shards
  bool banded_known = false;              // ... already decided for the problem being set (device build that fell back to the host build)
  // two-level preconditioner for band-structured problems (ba_coarse.hpp): structures built by ba_set_problem
  bool banded = false;         // mean camera span of a track <= Nc / 8 (sequential captures): pcg_model_tol's automatic default
  bool two_level_ok = false;   // this problem has them
  bool two_level = false;      // the current solve uses them
  int n_agg = 0, n_runs = 0, n_pairs = 0, coarse_bw = 0;
  DBuf<int> run_beg, run_pt, run_agg;
  DBuf<int2> run_pairs;
  DBuf<double> coarseU, coarseE, coarseEinv, coarse_rc, coarse_info;
  DBuf<long long> coarseEint;
  DBuf<double> dev_lam;        // device word a riding k_scalars stores the next damping in; 0 = not yet (ba_kernels.hpp, ScalarsArgs::lam_slot)
  DBuf<double> verdict;        // PCG verdict words {gamma, zeta, finished, -} x 2 iteration parities (point pass -> camera pass, vector kernel)
  int cam_segl = 64;           // lanes per (camera, partition) segment in the PCG camera pass (BA_CAM_SEGL, tuning)
  int nblkP = 1, ppb = 1, nblkV = 1;   // nblkV: camera-vector workgroups of the pinhole (VEC_CAMS cameras each)
  int nblkVm[2] = {1, 1};              // ... per camera model (CM::VC cameras each)
  size_t lds_bytes_m[2] = {0, 0};   // dynamic LDS of the point passes (largest window that fits), per camera model (row strides differ)
  bool jac_f32 = false;        // PCG passes recompute the Jacobian blocks in fp32 (ba_options.jacobian_precision = 1)
  bool all_lds_m[2] = {true, true};  // every point-pass workgroup's camera window fits in LDS, per camera model
  // pinned host mirror for scalars
  double* h_scal = nullptr;
  double* d_scal_host = nullptr;   // device-side address of h_scal (host-mapped, coherent)
  long long* h_flags = nullptr;    // host-mapped progress words: [0..1] PCG verdicts, [2..3] step scalars
  long long* d_flags = nullptr;
  long long flag_base = 1, step_seq = 1;
  // comm
  int rank = 0, world = 1;
  bool sys_diag = false;   // the last exchange_system carried Schur-Jacobi blocks (layout of sysmsg)
  bool one_part = false;       // multi-rank, thin shards: every camera's local observations in partition 0 (no fold kernels; ba_set_problem)
  bool multi = false;          // the multi-rank control flow is on: world > 1, or a communicator of ONE rank was forced
                               // (BA_COMM_FORCE=1: lets a single GPU execute every fold / all-reduce / decide step of the
                               // multi-rank loop through the real RCCL library)
  ncclComm_t nccl = nullptr;
  // host-staged shared-memory transport (BA_COMM=shm): a test vehicle that lets several ranks
  // share ONE GPU (RCCL refuses that), so the multi-rank control flow can be exercised end to end
  struct ShmComm* shm = nullptr;
  // device-side exchange of the per-PCG-iteration message through IPC-mapped peer buffers (BA_IPC=1; ba_kernels.hpp)
  struct IpcComm* ipc = nullptr;
  // first failed kernel launch since the last check (hipGetLastError right behind every launch)
  hipError_t launch_err = hipSuccess;
  char launch_what[160] = {0};
  int debug_lds_extra = 0;     // BA_DEBUG_LDS_EXTRA: bytes added to the point passes' dynamic LDS (tests provoke a failed launch)
  std::vector<ba_iter_record> trace;   // one record per LM iteration of the last ba_solve
  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_slot;
  size_t ev_used = 0, n_flushes = 0;
  ba_profile prof = {};
  std::vector<float> prof_ms[BA_PROFILE_SLOTS];   // every measured duration, per slot
};

// Every kernel launch goes through BA_LAUNCH: a launch the runtime refuses (dynamic LDS above the limit, an empty
// grid, ...) is not reported by a later stream synchronise, so the error is picked up right here and kept in the
// handle until the next check_launches().
static void note_launch(ba_handle* h, const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && h->launch_err == hipSuccess) {
    h->launch_err = e;
    snprintf(h->launch_what, sizeof h->launch_what, "%s", what);
  }
}
#define BA_LAUNCH(kern, ...)                  \
  do {                                        \
    hipLaunchKernelGGL(kern, __VA_ARGS__);    \
    note_launch(h, #kern);                    \
  } while (0)
static int check_launches(ba_handle* h);
// drain the stream, then report the first launch that failed since the last check (a refused launch leaves the
// outputs stale without making the synchronise fail)
#define BA_SYNC(h)                                            \
  do {                                                        \
    HIPCHECK(hipStreamSynchronize((h)->stream));              \
    if (int rc_sync_ = check_launches(h)) return rc_sync_;    \
  } while (0)
static int check_launches(ba_handle* h) {
  if (h->launch_err == hipSuccess) return BA_OK;
  const hipError_t e = h->launch_err;
  h->launch_err = hipSuccess;
  return fail(BA_ERR_HIP, "launch of kernel %s failed: %s", h->launch_what, hipGetErrorString(e));
}

static int set_device(ba_handle* h) {
  HIPCHECK(hipSetDevice(h->device));
  return BA_OK;
}

extern "C" int ba_device_count(int* n) {
  if (!n) return fail(BA_ERR_INVALID, "null argument");
  HIPCHECK(hipGetDeviceCount(n));
  return BA_OK;
}

// the LDS-table point passes need more than the default 64 KB of dynamic LDS
template <typename F>
static hipError_t allow_big_lds(F* f) {
  return hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TAB_BYTES);   // the largest table a launch asks for
}

static int create_impl(ba_handle* h, int device_id) {
  HIPCHECK(hipSetDevice(device_id));
  HIPCHECK(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device_id));
  if (h->n_cu < 1) h->n_cu = 1;
  HIPCHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHECK(hipHostMalloc((void**)&h->h_scal, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  HIPCHECK(hipHostGetDevicePointer((void**)&h->d_scal_host, h->h_scal, 0));
  HIPCHECK(hipHostMalloc((void**)&h->h_flags, 8 * sizeof(long long), hipHostMallocMapped | hipHostMallocCoherent));
  memset(h->h_flags, 0, 8 * sizeof(long long));
  HIPCHECK(hipHostGetDevicePointer((void**)&h->d_flags, h->h_flags, 0));
#define BA_BIG_LDS(K) HIPCHECK(allow_big_lds(K))
#define BA_BIG_LDS_LIN(CM, R, L) BA_BIG_LDS((k_pt_linearize<CM, R, L, 2>)); BA_BIG_LDS((k_pt_linearize<CM, R, L, 4>));        \
  BA_BIG_LDS((k_pt_linearize<CM, R, L, 8>)); BA_BIG_LDS((k_pt_linearize<CM, R, L, 16>)); BA_BIG_LDS((k_pt_linearize_both<CM, R, L>))
#define BA_BIG_LDS_SCH1(CM, R, M, L, LN) BA_BIG_LDS((k_pt_schur<CM, R, M, L, LN, double>)); BA_BIG_LDS((k_pt_schur<CM, R, M, L, LN, float>))
#define BA_BIG_LDS_SCH(CM, R, M, L) BA_BIG_LDS_SCH1(CM, R, M, L, 2); BA_BIG_LDS_SCH1(CM, R, M, L, 4); BA_BIG_LDS_SCH1(CM, R, M, L, 8);     \
  BA_BIG_LDS_SCH1(CM, R, M, L, 16); BA_BIG_LDS((k_pt_schur_both<CM, R, M, L, double>)); BA_BIG_LDS((k_pt_schur_both<CM, R, M, L, float>))
#define BA_BIG_LDS_MODEL(CM)                                                                                                            \
  BA_BIG_LDS_LIN(CM, true, true); BA_BIG_LDS_LIN(CM, true, false); BA_BIG_LDS_LIN(CM, false, true); BA_BIG_LDS_LIN(CM, false, false);   \
  BA_BIG_LDS_SCH(CM, true, 0, true); BA_BIG_LDS_SCH(CM, true, 0, false); BA_BIG_LDS_SCH(CM, false, 0, true); BA_BIG_LDS_SCH(CM, false, 0, false); \
  BA_BIG_LDS_SCH(CM, true, 1, true); BA_BIG_LDS_SCH(CM, true, 1, false); BA_BIG_LDS_SCH(CM, false, 1, true); BA_BIG_LDS_SCH(CM, false, 1, false)
  BA_BIG_LDS_MODEL(Pinhole);
  BA_BIG_LDS_MODEL(BalCam);
#undef BA_BIG_LDS_MODEL
#undef BA_BIG_LDS_SCH
#undef BA_BIG_LDS_SCH1
#undef BA_BIG_LDS_LIN
#undef BA_BIG_LDS
  if (const char* e = getenv("BA_DEBUG_LDS_EXTRA")) h->debug_lds_extra = atoi(e);
  return BA_OK;
}
extern "C" int ba_destroy(ba_handle* h);
extern "C" int ba_create(int device_id, ba_handle** out) {
  if (!out) return fail(BA_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  HIPCHECK(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return fail(BA_ERR_INVALID, "device %d not in [0,%d)", device_id, n);
  ba_handle* h = new ba_handle();
  h->device = device_id;
  if (int rc = create_impl(h, device_id)) {      // nothing half-built leaks: stream, pinned blocks and the handle go
    const std::string msg = g_err;
    ba_destroy(h);
    g_err = msg;
    return rc;
  }
  *out = h;
  return BA_OK;
}

static void flush_profile(ba_handle* h);
static void shm_destroy(ba_handle* h);
static void ipc_destroy(ba_handle* h);
static int shm_init(ba_handle* h, int rank, int world, const void* id128);
static int ipc_init(ba_handle* h, int rank, int world, const void* id128);

extern "C" int ba_destroy(ba_handle* h) {
  if (!h) return BA_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(h->nccl);
  ipc_destroy(h);
  shm_destroy(h);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  DBuf<int>* ib[] = {&h->offk, &h->c_pt, &h->c_orig, &h->pt_off, &h->p_cam, &h->slot, &h->long_pts, &h->c_ptf[0], &h->c_ptf[1],
                     &h->p_camf[0], &h->p_camf[1]};
  for (auto b : ib) b->release();
  h->blk_win.release();
  DBuf<double2>* d2[] = {&h->c_uv, &h->p_uv, &h->c_w[0], &h->c_w[1], &h->p_w[0], &h->p_w[1]};
  for (auto b : d2) b->release();
  DBuf<double>* db[] = {&h->cams[0], &h->cams[1], &h->cs[0], &h->cs[1], &h->ptab[0], &h->ptab[1], &h->stage,
                        &h->camA[0], &h->camA[1], &h->HccBc, &h->Hpp[0], &h->Hpp[1], &h->bp[0], &h->bp[1],
                        &h->Hppinv[0], &h->Hppinv[1], &h->y0[0], &h->y0[1], &h->Hccd, &h->Minv,
                        &h->partR, &h->partL[0], &h->partL[1], &h->part6, &h->partE, &h->sysmsg, &h->linmsg[0], &h->linmsg[1], &h->partA, &h->partB, &h->partC, &h->partV,
                        &h->partG[0], &h->partG[1], &h->partGc,
                        &h->gvec, &h->x, &h->r, &h->p, &h->s, &h->z, &h->vin, &h->vx, &h->scal, &h->rbuf, &h->gather};
  for (auto b : db) b->release();
  h->st.release();
  h->tri.release();
  h->small_V.release();
  h->mw_woff.release(); h->mw_buf.release();
  if (h->h_par) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->h_par); h->h_par = nullptr; h->h_par_cap = 0; }
  if (h->par_event) { (void)hipEventDestroy(h->par_event); h->par_event = nullptr; }
  if (h->up_event) { (void)hipEventDestroy(h->up_event); h->up_event = nullptr; }
  h->up_pending = false;
  h->par_pending = false;
  h->intr[0].release(); h->intr[1].release();
  h->small_gS.release();
  h->small_np_pad = -1;
  h->run_beg.release(); h->run_pt.release(); h->run_agg.release(); h->run_pairs.release();
  h->coarseU.release(); h->coarseE.release(); h->coarseEinv.release(); h->coarse_rc.release(); h->coarse_info.release();
  h->coarseEint.release();
  h->verdict.release();
  h->dev_lam.release();
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->h_flags) (void)hipHostFree(h->h_flags);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_up) (void)hipHostFree(h->h_up);
  if (h->h_setup) (void)hipHostFree(h->h_setup);
  h->setup_i.release();
  h->up_dev.release();
  h->stat2.release();
  if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BA_OK;
}

extern "C" int ba_synchronize(ba_handle* h) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (set_device(h)) return BA_ERR_HIP;
  BA_SYNC(h);
  return BA_OK;
}

// ------------------------------------------------------------------------------ comm
extern "C" int ba_comm_unique_id(void* id128) {
  if (!id128) return fail(BA_ERR_INVALID, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  { const char* e = getenv("BA_COMM");
    if (e && strcmp(e, "shm") == 0) {               // any 128 unique bytes will do
      unsigned long long seed = (unsigned long long)getpid() * 2654435761ULL ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
      unsigned char* o = (unsigned char*)id128;
      for (int i = 0; i < 128; ++i) { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; o[i] = (unsigned char)(seed >> 33); }
      return BA_OK;
    } }
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  memcpy(id128, &id, 128);
  return BA_OK;
}

extern "C" int ba_comm_init(ba_handle* h, int rank, int world, const void* id128) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (world < 1 || rank < 0 || rank >= world) return fail(BA_ERR_INVALID, "rank %d / world %d", rank, world);
  h->rank = rank;
  h->world = world;
  h->multi = world > 1 || getenv("BA_COMM_FORCE") != nullptr;
  if (!h->multi) return BA_OK;
  if (!id128) return fail(BA_ERR_INVALID, "null id buffer");
  if (set_device(h)) return BA_ERR_HIP;
  // BA_IPC=1: the per-PCG-iteration exchange goes through IPC-mapped peer buffers (device-side stores + flags, consumed
  // inside k_pcg_step) instead of the base transport's all-reduce; everything else stays on the base transport
  if (const char* e = getenv("BA_IPC")) if (atoi(e) != 0) { if (int rc = ipc_init(h, rank, world, id128)) return rc; }
  { const char* e = getenv("BA_COMM"); if (e && strcmp(e, "shm") == 0) return shm_init(h, rank, world, id128); }
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = g_rccl.CommInitRank(&h->nccl, world, id, rank);
  if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return BA_OK;
}

// --------------------------------------------------------------------------- profile
static hipEvent_t next_event(ba_handle* h) {
  if (h->ev_used == h->ev.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    h->ev.push_back(e);
  }
  return h->ev[h->ev_used++];
}
struct Scope {   // brackets one launch with two events when profiling
  ba_handle* h;
  Scope(ba_handle* hh, int slot) : h(hh) {
    if (h->profile) {
      h->ev_slot.push_back(slot);
      (void)hipEventRecord(next_event(h), h->stream);
    }
  }
  ~Scope() {
    if (h->profile) {
      (void)hipEventRecord(next_event(h), h->stream);
      if (h->ev_used >= 8192) flush_profile(h);
    }
  }
};
static void flush_profile(ba_handle* h) {
  if (h->ev_used == 0) return;
  (void)hipStreamSynchronize(h->stream);
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]);
    const int slot = h->ev_slot[i / 2];
    h->prof.launches[slot] += 1;
    h->prof.total_ms[slot] += ms;
    h->prof_ms[slot].push_back(ms);
  }
  h->ev_used = 0;
  h->ev_slot.clear();
  h->n_flushes++;
}
extern "C" int ba_get_profile(ba_handle* h, ba_profile* out) {
  if (!h || !out) return fail(BA_ERR_INVALID, "null argument");
  flush_profile(h);
  for (int sl = 0; sl < BA_PROFILE_SLOTS; ++sl) {
    // working launches: within [0.5, 4] x the 90th percentile (early exits below, rare host-side
    // hiccups between the two events above)
    std::vector<float> v = h->prof_ms[sl];
    h->prof.working_launches[sl] = 0;
    h->prof.working_ms[sl] = 0;
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    const float ref = v[(size_t)(0.9 * (v.size() - 1))];
    for (float d : v) if (d >= 0.5f * ref && d <= 4.0f * ref) { h->prof.working_launches[sl]++; h->prof.working_ms[sl] += d; }
  }
  *out = h->prof;
  return BA_OK;
}
extern "C" int ba_reset_profile(ba_handle* h) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  flush_profile(h);
  memset(&h->prof, 0, sizeof h->prof);
  for (auto& v : h->prof_ms) v.clear();
  return BA_OK;
}

// ------------------------------------------------------------- shared-memory transport
// Same semantics as the RCCL path (in-place sum / max all-reduce of doubles, identical bits
// on every rank), implemented with a POSIX shared-memory segment and host staging.  Slow by
// design; selected with BA_COMM=shm.  Never used unless asked for.
struct ShmComm {
  static constexpr size_t SLOT = 8u << 20;         // bytes per rank
  int rank = 0, world = 1;
  char name[64] = {0};
  unsigned char* base = nullptr;
  size_t bytes = 0;
  double* stage = nullptr;                          // pinned
  unsigned gen = 0;
  std::atomic<unsigned>* counter() { return reinterpret_cast<std::atomic<unsigned>*>(base); }
  std::atomic<unsigned>* sense() { return reinterpret_cast<std::atomic<unsigned>*>(base + 64); }
  double* slot(int r) { return reinterpret_cast<double*>(base + 4096 + (size_t)r * SLOT); }
  int barrier() {
    const unsigned my = ++gen;
    if (counter()->fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)world) {
      counter()->store(0, std::memory_order_relaxed);
      sense()->store(my, std::memory_order_release);
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (sense()->load(std::memory_order_acquire) < my) {
        if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0)
          return -1;
      }
    }
    return 0;
  }
};

static int shm_init(ba_handle* h, int rank, int world, const void* id128) {
  ShmComm* c = new ShmComm();
  c->rank = rank; c->world = world;
  const unsigned char* id = (const unsigned char*)id128;
  unsigned long long tag = 1469598103934665603ULL;
  for (int i = 0; i < 128; ++i) tag = (tag ^ id[i]) * 1099511628211ULL;
  snprintf(c->name, sizeof c->name, "/ba_hip_%016llx", tag);
  c->bytes = 4096 + (size_t)world * ShmComm::SLOT;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return fail(BA_ERR_COMM, "shm_open(%s) failed", c->name); }
  if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return fail(BA_ERR_COMM, "ftruncate failed"); }
  c->base = (unsigned char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return fail(BA_ERR_COMM, "mmap failed"); }
  if (hipHostMalloc((void**)&c->stage, ShmComm::SLOT) != hipSuccess) { delete c; return fail(BA_ERR_HIP, "hipHostMalloc failed"); }
  h->shm = c;
  if (c->barrier()) return fail(BA_ERR_COMM, "shm barrier timed out at init");
  return BA_OK;
}
static void shm_destroy(ba_handle* h) {
  if (!h->shm) return;
  ShmComm* c = h->shm;
  (void)c->barrier();
  if (c->stage) (void)hipHostFree(c->stage);
  if (c->base) munmap(c->base, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  h->shm = nullptr;
}
static int shm_allreduce(ba_handle* h, double* buf, size_t count, bool is_max) {
  ShmComm* c = h->shm;
  if (count * sizeof(double) > ShmComm::SLOT) return fail(BA_ERR_COMM, "shm all-reduce of %zu doubles exceeds the slot", count);
  HIPCHECK(hipMemcpyAsync(c->stage, buf, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  memcpy(c->slot(c->rank), c->stage, count * sizeof(double));
  if (c->barrier()) return fail(BA_ERR_COMM, "shm barrier timed out");
  for (size_t i = 0; i < count; ++i) {            // rank order: identical bits on every rank
    double a = c->slot(0)[i];
    for (int r = 1; r < c->world; ++r) a = is_max ? std::max(a, c->slot(r)[i]) : a + c->slot(r)[i];
    c->stage[i] = a;
  }
  if (c->barrier()) return fail(BA_ERR_COMM, "shm barrier timed out");
  HIPCHECK(hipMemcpyAsync(buf, c->stage, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  BA_SYNC(h);      // the staging buffer is reused by the next call
  return BA_OK;
}

// ------------------------------------------------------ device-side exchange over IPC-mapped peer buffers (BA_IPC=1)
// Every rank allocates a receive buffer ([2 parities][world][stride] doubles + flag lines) in fine-grained device memory,
// exports it with hipIpcGetMemHandle and opens every peer's; the handles travel through a small POSIX shared-memory
// board named after the communicator id (one node: what bench.py --gpus N runs on).  Used for the Schur product's
// exchange inside the PCG loop (inside k_pcg_step); every other collective keeps the base transport.
struct IpcComm {
  static constexpr size_t STRIDE = 32768;             // doubles per (parity, sender) slot: messages up to 256 KB (3600 BAL cameras)
  int rank = 0, world = 1;
  char name[64] = {0};
  unsigned char* board = nullptr;                     // shm: [64-byte counter line][64-byte sense line][world x 128 bytes of handle]
  size_t board_bytes = 0;
  unsigned gen = 0;
  void* local = nullptr;                              // own receive buffer (recv doubles, then the flag lines)
  void* opened[IPC_MAX_WORLD] = {nullptr};
  IpcPeers peers;
  long long seq = 0;
  size_t recv_doubles() const { return (size_t)2 * world * STRIDE; }
  size_t bytes() const { return recv_doubles() * sizeof(double) + (size_t)2 * world * IPC_MAX_BLOCKS * sizeof(unsigned long long); }
  std::atomic<unsigned>* counter() { return reinterpret_cast<std::atomic<unsigned>*>(board); }
  std::atomic<unsigned>* sense() { return reinterpret_cast<std::atomic<unsigned>*>(board + 64); }
  int barrier() {
    const unsigned my = ++gen;
    if (counter()->fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)world) {
      counter()->store(0, std::memory_order_relaxed);
      sense()->store(my, std::memory_order_release);
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (sense()->load(std::memory_order_acquire) < my)
        if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) return -1;
    }
    return 0;
  }
};
static void ipc_destroy(ba_handle* h) {
  if (!h->ipc) return;
  IpcComm* c = h->ipc;
  if (c->board) (void)c->barrier();                   // nobody unmaps a buffer a peer may still be storing into
  for (int r = 0; r < c->world; ++r) if (c->opened[r]) (void)hipIpcCloseMemHandle(c->opened[r]);
  if (c->local) (void)hipFree(c->local);
  if (c->board) { munmap(c->board, c->board_bytes); if (c->rank == 0) shm_unlink(c->name); }
  delete c;
  h->ipc = nullptr;
}
static int ipc_init(ba_handle* h, int rank, int world, const void* id128) {
  if (world > IPC_MAX_WORLD) return fail(BA_ERR_COMM, "BA_IPC: at most %d ranks", IPC_MAX_WORLD);
  IpcComm* c = new IpcComm();
  h->ipc = c;
  c->rank = rank; c->world = world;
  const unsigned char* id = (const unsigned char*)id128;
  unsigned long long tag = 1469598103934665603ULL;
  for (int i = 0; i < 128; ++i) tag = (tag ^ id[i]) * 1099511628211ULL;
  snprintf(c->name, sizeof c->name, "/ba_ipc_%016llx", tag);
  c->board_bytes = 128 + (size_t)world * 128;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return fail(BA_ERR_COMM, "shm_open(%s) failed", c->name);
  if (ftruncate(fd, (off_t)c->board_bytes) != 0) { close(fd); return fail(BA_ERR_COMM, "ftruncate failed"); }
  void* m = mmap(nullptr, c->board_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return fail(BA_ERR_COMM, "mmap failed");
  c->board = (unsigned char*)m;
  // fine-grained device memory: peer stores are visible to this device's loads without cache maintenance
  if (hipExtMallocWithFlags(&c->local, c->bytes(), hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    return fail(BA_ERR_COMM, "BA_IPC: fine-grained device memory for the receive buffer is not available");
  }
  HIPCHECK(hipMemset(c->local, 0, c->bytes()));
  HIPCHECK(hipDeviceSynchronize());
  hipIpcMemHandle_t mine;
  if (hipIpcGetMemHandle(&mine, c->local) != hipSuccess) { (void)hipGetLastError(); return fail(BA_ERR_COMM, "hipIpcGetMemHandle failed"); }
  static_assert(sizeof(hipIpcMemHandle_t) <= 120, "handle does not fit its board slot");
  memcpy(c->board + 128 + (size_t)rank * 128, &mine, sizeof mine);
  const int dev_of_rank = h->device;
  memcpy(c->board + 128 + (size_t)rank * 128 + 120, &dev_of_rank, sizeof(int));
  if (c->barrier()) return fail(BA_ERR_COMM, "BA_IPC: board barrier timed out");
  for (int r = 0; r < world; ++r) {
    void* base = c->local;
    if (r != rank) {
      hipIpcMemHandle_t hd;
      memcpy(&hd, c->board + 128 + (size_t)r * 128, sizeof hd);
      int peer_dev = 0;
      memcpy(&peer_dev, c->board + 128 + (size_t)r * 128 + 120, sizeof(int));
      if (peer_dev != h->device) {                      // another GPU of the node: its memory has to be reachable from this one
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, h->device, peer_dev);
        if (can) { const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0); if (e != hipSuccess) (void)hipGetLastError(); }
      }
      if (hipIpcOpenMemHandle(&base, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return fail(BA_ERR_COMM, "BA_IPC: hipIpcOpenMemHandle of rank %d's buffer failed", r);
      }
      c->opened[r] = base;
    }
    c->peers.recv[r] = (double*)base;
    c->peers.flags[r] = (unsigned long long*)((double*)base + c->recv_doubles());
  }
  if (c->barrier()) return fail(BA_ERR_COMM, "BA_IPC: board barrier timed out");
  return BA_OK;
}

static int allreduce(ba_handle* h, double* buf, size_t count, bool is_max = false) {
  if (!h->multi) return BA_OK;
  Scope sc(h, BA_K_ALLREDUCE);
  if (h->shm) return shm_allreduce(h, buf, count, is_max);
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, is_max ? ncclMax : ncclSum, h->nccl, h->stream);
  if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return BA_OK;
}

// ---------------------------------------------------------------------------- problem
// The band statistic behind pcg_model_tol's automatic default (mean camera span of a track <= Nc / 8).  Tracks are split
// over the ranks of a multi-rank job by landmark, so the sums of the shards are the whole problem's: one small all-reduce
// makes every rank decide what a single rank would.  Collective: every rank's ba_set_problem calls it exactly once.
static int decide_banded(ba_handle* h, double span_sum, double tracks, int Nc) {
  if (h->multi) {
    HIPCHECK(h->stat2.alloc(2));
    const double v[2] = {span_sum, tracks};
    HIPCHECK(hipMemcpyAsync(h->stat2.p, v, sizeof v, hipMemcpyHostToDevice, h->stream));
    if (int rc = allreduce(h, h->stat2.p, 2)) return rc;
    double w[2];
    HIPCHECK(hipMemcpyAsync(w, h->stat2.p, sizeof w, hipMemcpyDeviceToHost, h->stream));
    BA_SYNC(h);
    span_sum = w[0]; tracks = w[1];
  }
  h->banded = tracks > 0 && span_sum / tracks <= Nc / 8.0;
  h->stats[BA_STAT_BANDED] = h->banded ? 1 : 0;
  h->banded_known = true;
  return BA_OK;
}
// Point-pass grid, part 1 (needs the problem's dimensions only): lanes per point, number and length of the point ranges,
// lanes per segment of the PCG camera pass.  Shared by the host and the device build of ba_set_problem.
struct PtGrid { bool table_fits; int per_cu, pts_per_pass, want; };
static PtGrid config_point_grid(ba_handle* h, int Nc, int Np, int No) {
  const size_t full_table = (size_t)Nc * TA * sizeof(double);
  const bool table_fits = full_table <= (size_t)LDS_TAB_BYTES;
  const int per_cu = !table_fits ? 1 : (int)std::max<size_t>(1, std::min<size_t>(2048 / PT_THREADS, (size_t)(160 * 1024) / (full_table + 1024)));
  h->lanes = LPP;
  for (int ln = 16; ln > LPP; ln >>= 1)
    if ((Np + PT_THREADS / ln - 1) / (PT_THREADS / ln) <= h->n_cu * per_cu) { h->lanes = ln; break; }
  if (const char* e = getenv("BA_PT_LANES")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16) h->lanes = v; }
  const int pts_per_pass = PT_THREADS / h->lanes;
  const int want = std::max(1, (Np + pts_per_pass - 1) / pts_per_pass);
  h->nblkP = std::min(want, 4096);
  if (table_fits) h->nblkP = std::min(h->nblkP, h->n_cu * per_cu);
  if (const char* e = getenv("BA_PT_BLOCKS")) h->nblkP = std::max(1, std::min(want, atoi(e)));
  if (const char* e = getenv("BA_XCD_RANGES")) h->xcd_ranges = atoi(e) != 0;
  // lanes per (camera, partition) segment in the PCG camera pass: a wave, or a 16-lane row when segments are short
  // (config 5: ~48 observations per segment -- a wave would walk it in one step with a quarter of its lanes idle and
  // pay the 64-lane reduction of every sum for it; measured 13.4 -> 12.1 us, C3's ~125-observation segments keep the wave)
  h->cam_segl = (Nc > 0 && (long long)No / Nc / NPART < 64) ? 16 : 64;
  if (const char* e = getenv("BA_CAM_SEGL")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) h->cam_segl = v; }
  h->ppb = std::max(1, (Np + h->nblkP - 1) / h->nblkP);
  return PtGrid{table_fits, per_cu, pts_per_pass, want};
}
// part 2a: the track length above which a point gets a 16-lane row of its own (med = element Np / 2 of the sorted lengths)
static void config_long_threshold(ba_handle* h, int med) {
  const char* e = getenv("BA_LONG_TRACK");
  h->long_thr = e ? std::max(1, atoi(e)) : std::max(8, 2 * med);
  if (h->lanes != LPP) h->long_thr = 0x7fffffff;           // more lanes per point already: no separate long-track rows
}
// part 2b: how the long tracks are dealt to workgroups, and -- when ranges plus long-track workgroups would not all be
// resident at once -- the grid with the fewest rounds per workgroup
static void config_long_grid(ba_handle* h, const PtGrid& g, int Np, int n_long) {
  h->n_long = n_long;
  h->long_spb = PT_THREADS / LPP_LONG;
  if (const char* e = getenv("BA_LONG_SLOTS")) h->long_spb = std::max(1, atoi(e)) * (PT_THREADS / LPP_LONG);
  h->nblkL = (h->n_long + h->long_spb - 1) / h->long_spb;
  // A point-pass workgroup is 1024 threads at 128 VGPRs: ONE per compute unit.  When one-round ranges plus one-round
  // long-track workgroups need somewhat more workgroups than the chip has units (config 5: 306 + 175 on 256), the
  // second wave of workgroups runs on part of the chip while the rest idles, and every workgroup pays its launch and its
  // window copy for one round of work.  A round (512 points at 2 lanes, 64 long tracks at 16) is latency-bound and costs
  // about the same whatever it holds, so the cost of a launch is the largest number of ROUNDS any workgroup walks:
  // choose the long-track workgroups' size (m rounds) and the ranges' length such that everything is resident at once and
  // that number is smallest (config 5: 168 ranges of 932 points + 88 x 128 long tracks, two rounds each; Schur point
  // pass 17.0 -> 15.2 us pinhole, 20.0 -> 18.0 us BAL camera).  Much larger WINDOWED problems (more than two rounds per
  // unit) keep one-round ranges: narrow windows matter more there.
  if (!getenv("BA_PT_BLOCKS") && (!g.table_fits || h->n_long > 0)) {
    int best_m = 0, best_cost = g.table_fits ? 0x7fffffff : 3, best_nb = 0;   // (table in LDS: ranges of any length share one fill)
    const bool pick_m = !getenv("BA_LONG_SLOTS") && h->n_long > 0;
    for (int m = 1; m <= (pick_m ? 4 : 1); ++m) {
      const int spb = pick_m ? m * (PT_THREADS / LPP_LONG) : h->long_spb;
      const int nl = (h->n_long + spb - 1) / spb;
      const int avail = h->n_cu - nl;
      if (avail < 1) continue;
      const int nb = std::min(g.want, avail);
      const int rounds = ((Np + nb - 1) / nb + g.pts_per_pass - 1) / g.pts_per_pass;
      const int cost = std::max(rounds, h->n_long > 0 ? spb / (PT_THREADS / LPP_LONG) : 0);
      if (cost < best_cost) { best_cost = cost; best_m = m; best_nb = nb; }
    }
    if (best_m && g.want + h->nblkL > h->n_cu) {
      if (pick_m) { h->long_spb = best_m * (PT_THREADS / LPP_LONG); h->nblkL = (h->n_long + h->long_spb - 1) / h->long_spb; }
      h->nblkP = best_nb;
      h->ppb = (Np + best_nb - 1) / best_nb;
    }
  }
}
// Every device buffer of a problem whose size follows from the problem's dimensions and the point-pass grid alone (h->Nc,
// Np, Nobs, nblkP, nblkL, multi, world are set).  Idempotent: a buffer that is large enough is kept.
static int alloc_solver_buffers(ba_handle* h) {
  const int Nc = h->Nc, Np = h->Np, No = h->Nobs;
  const size_t nobs1 = std::max(No, 1), np1 = std::max(Np, 1);
  const size_t nbv_max = (size_t)std::max(h->nblkVm[0], h->nblkVm[1]);
  HIPCHECK(h->c_pt.alloc(nobs1)); HIPCHECK(h->c_orig.alloc(nobs1)); HIPCHECK(h->p_cam.alloc(nobs1));
  HIPCHECK(h->c_uv.alloc(nobs1)); HIPCHECK(h->p_uv.alloc(nobs1));
  HIPCHECK(h->c_w[0].alloc(nobs1)); HIPCHECK(h->c_w[1].alloc(nobs1)); HIPCHECK(h->p_w[0].alloc(nobs1)); HIPCHECK(h->p_w[1].alloc(nobs1));
  HIPCHECK(h->c_ptf[0].alloc(nobs1)); HIPCHECK(h->c_ptf[1].alloc(nobs1)); HIPCHECK(h->p_camf[0].alloc(nobs1)); HIPCHECK(h->p_camf[1].alloc(nobs1));
  for (int k = 0; k < 2; ++k) {
    HIPCHECK(h->cams[k].alloc(6 * (size_t)Nc)); HIPCHECK(h->cs[k].alloc(CS * (size_t)Nc));
    HIPCHECK(h->ptab[k].alloc(PT * np1));     // (k_pack_points writes whole records of set 0; set 1 gets X from the back
  }                                           //  substitution and y from the point half before either is read)
  HIPCHECK(h->stage.alloc(3 * np1));
  // per-camera buffers are sized for the larger camera model (BAL: 9 parameters, 45 + 9 sums, 26-double table rows)
  constexpr size_t NBX = BalCam::NB, NHX = BalCam::NH, NLX = BalCam::NL;
  HIPCHECK(h->camA[0].alloc(TA_MAX * (size_t)Nc)); HIPCHECK(h->camA[1].alloc(TA_MAX * (size_t)Nc));
  HIPCHECK(h->intr[0].alloc(3 * (size_t)Nc)); HIPCHECK(h->intr[1].alloc(3 * (size_t)Nc));
  HIPCHECK(h->HccBc.alloc(NLX * (size_t)Nc + 8));   // Hcc (NH Nc) | bc (NB Nc): one all-reduce
  for (int k = 0; k < 2; ++k) {
    HIPCHECK(h->Hpp[k].alloc(6 * np1)); HIPCHECK(h->bp[k].alloc(3 * np1)); HIPCHECK(h->Hppinv[k].alloc(6 * np1));
    HIPCHECK(h->y0[k].alloc(3 * np1));
  }
  h->pb = 0;
  HIPCHECK(h->Hccd.alloc(NHX * (size_t)Nc)); HIPCHECK(h->Minv.alloc(NHX * (size_t)Nc));
  HIPCHECK(h->partR.alloc(2 * (size_t)NPART * Nc));
  HIPCHECK(h->partL[0].alloc(NLX * (size_t)NPART * Nc)); HIPCHECK(h->partL[1].alloc(NLX * (size_t)NPART * Nc));
  HIPCHECK(h->part6.alloc(NBX * (size_t)NPART * Nc + 8));   // + the u.y word: one all-reduce carries both
  HIPCHECK(h->partE.alloc(NHX * (size_t)NPART * Nc));
  if (h->multi) { HIPCHECK(h->linmsg[0].alloc(8 + NLX * (size_t)Nc)); HIPCHECK(h->linmsg[1].alloc(8 + NLX * (size_t)Nc)); }
  if (h->multi) HIPCHECK(h->sysmsg.alloc(2 + (size_t)(NBX + NHX) * Nc + (size_t)h->world + 8));
  HIPCHECK(h->partA.alloc(h->nblkP + h->nblkL)); HIPCHECK(h->partB.alloc(4 * (size_t)(h->nblkP + h->nblkL)));
  HIPCHECK(h->partC.alloc(5 * nbv_max)); HIPCHECK(h->partV.alloc(4 * nbv_max));
  HIPCHECK(h->partG[0].alloc(h->nblkP + h->nblkL)); HIPCHECK(h->partG[1].alloc(h->nblkP + h->nblkL)); HIPCHECK(h->partGc.alloc(nbv_max));
  DBuf<double>* v6[] = {&h->gvec, &h->x, &h->r, &h->p, &h->s, &h->z, &h->vin, &h->vx};
  for (auto b : v6) HIPCHECK(b->alloc(NBX * (size_t)Nc));
  HIPCHECK(h->scal.alloc(64));
  HIPCHECK(h->st.alloc(2));
  HIPCHECK(h->verdict.alloc(8));
  HIPCHECK(hipMemsetAsync(h->verdict.p, 0, 8 * sizeof(double), h->stream));
  HIPCHECK(h->dev_lam.alloc(2));
  // (dev_lam is cleared by every back substitution; camA rows: geometry by k_cam_prepare / k_cam_update, vt by the
  //  PCG setup before any pass reads it)
  return BA_OK;
}

// ---- ba_set_problem, device build (ba_setup.hpp): one upload of the caller's arrays, every ordering derived by kernels.
// Returns BA_OK, a negative ba_status, or 1 = "not for this problem" (the caller then runs the host build; nothing the
// host build relies on has been touched).  Chosen for large problems whose whole camera table fits in LDS (the caller's
// point numbering is kept then); bit-equal to the host build (tests/test_gpu_setup.py).
constexpr int SETUP_HIST_BINS = 4096;
constexpr size_t SETUP_PINNED_BYTES = 128 * 1024;
// BA_PIXELS=f64 keeps the pixel streams double2 whatever the values (A / B measurements, tests)
static bool uv_f32_wanted() {
  const char* e = getenv("BA_PIXELS");
  return !(e && strcmp(e, "f64") == 0);
}
static inline UvArr uv_arr(const ba_handle* h, const DBuf<double2>& b) { return UvArr{b.p, h->uv_f32 ? 1 : 0}; }
static int dev_scan(ba_handle* h, const int* in, int n, int* bsum, int* out) {
  const int nb = (n + SETUP_SCAN_BLOCK - 1) / SETUP_SCAN_BLOCK;
  BA_LAUNCH(k_scan_block_sums, dim3(nb), dim3(1024), 0, h->stream, in, n, bsum);
  BA_LAUNCH(k_scan_top, dim3(1), dim3(1024), 0, h->stream, bsum, nb);
  BA_LAUNCH(k_scan_final, dim3(nb), dim3(1024), 0, h->stream, in, n, (const int*)bsum, out);
  return BA_OK;
}
static int set_problem_device(ba_handle* h, int Nc, int Np, int No, const int32_t* cam_idx, const int32_t* pt_idx, const double* uv,
                              const double K4[4], int fixed_cam, bool timed) {
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* name) {
    if (!timed) return;
    (void)hipStreamSynchronize(h->stream);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "ba_set_problem [device] %-24s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  if ((size_t)Np + 1 > (size_t)1024 * SETUP_SCAN_BLOCK) return 1;
  h->Nc = Nc; h->Np = Np; h->Nobs = No; h->fixed = fixed_cam;
  memcpy(h->K4, K4, sizeof h->K4);
  h->nblkV = (Nc + VEC_CAMS - 1) / VEC_CAMS;
  h->nblkVm[0] = (Nc + Pinhole::VC - 1) / Pinhole::VC; h->nblkVm[1] = (Nc + BalCam::VC - 1) / BalCam::VC;
  const PtGrid grid = config_point_grid(h, Nc, Np, No);
  if (!grid.table_fits) return 1;
  if (!h->h_setup) HIPCHECK(hipHostMalloc((void**)&h->h_setup, SETUP_PINNED_BYTES, hipHostMallocDefault));
  // observation-sized buffers the build works in; the flagged index copies serve as scratch until k_init_flagged fills them
  HIPCHECK(h->p_camf[0].alloc(No)); HIPCHECK(h->p_camf[1].alloc(No)); HIPCHECK(h->c_ptf[0].alloc(No)); HIPCHECK(h->c_ptf[1].alloc(No));
  HIPCHECK(h->p_cam.alloc(No)); HIPCHECK(h->c_pt.alloc(No)); HIPCHECK(h->c_orig.alloc(No));
  HIPCHECK(h->c_uv.alloc(No)); HIPCHECK(h->p_uv.alloc(No)); HIPCHECK(h->rbuf.alloc(2 * (size_t)No));
  HIPCHECK(h->pt_off.alloc(Np + 1)); HIPCHECK(h->offk.alloc((size_t)Nc * (NPART + 1))); HIPCHECK(h->slot.alloc(Np));
  // scratch: p_src[No] | cnt[Np+1] fill[Np] big[Np] flag[Np+1] pos[Np+1] | cam_cnt[Nc+1] cam_fill[Nc] cam_off[Nc+1] | bsum[1032] |
  //          hist[SETUP_HIST_BINS] | words[16] (0 bad, 1 n_tracks, 2 n_big) | u64[4] (0 span_sum, 1 in_partition, 2 in_band)
  const size_t o_cnt = (size_t)No, o_fill = o_cnt + Np + 1, o_big = o_fill + Np, o_flag = o_big + Np, o_pos = o_flag + Np + 1;
  const size_t o_ccnt = o_pos + Np + 1, o_cfill = o_ccnt + Nc + 1, o_coff = o_cfill + Nc, o_bsum = o_coff + Nc + 1;
  const size_t o_hist = o_bsum + 1032, o_words = o_hist + SETUP_HIST_BINS, o_u64 = (o_words + 16 + 1) & ~(size_t)1, o_end = o_u64 + 8;
  HIPCHECK(h->setup_i.alloc(o_end));
  int* const S = h->setup_i.p;
  int *d_cam = h->p_camf[0].p, *d_pt = h->p_camf[1].p, *seg = h->c_ptf[0].p, *p_pt = h->c_ptf[1].p, *p_src = S;
  unsigned long long* u64 = (unsigned long long*)(S + o_u64);
  HIPCHECK(hipMemcpyAsync(d_cam, cam_idx, (size_t)No * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(d_pt, pt_idx, (size_t)No * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->rbuf.p, uv, 2 * (size_t)No * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemsetAsync(S + o_cnt, 0, (o_end - o_cnt) * sizeof(int), h->stream));
  HIPCHECK(hipMemsetAsync(S + o_words, 0x7f, sizeof(int), h->stream));                   // bad = 0x7f7f7f7f: "none"
  stage("upload");
  const dim3 go((No + 255) / 256), gp((Np + 255) / 256), b256(256);
  // (words: 0 first bad observation, 1 tracks, 2 big points, 3 "some pixel is not a float32 value")
  BA_LAUNCH(k_setup_hist, go, b256, 0, h->stream, (const int*)d_cam, (const int*)d_pt, No, Nc, Np, S + o_cnt, S + o_words,
            (uv_f32_wanted() && Nc > SMALL_MAX_CAMS) ? (const double2*)h->rbuf.p : (const double2*)nullptr, S + o_words + 3);
  dev_scan(h, S + o_cnt, Np, S + o_bsum, h->pt_off.p);
  BA_LAUNCH(k_setup_scatter_pt, go, b256, 0, h->stream, (const int*)d_pt, No, (const int*)h->pt_off.p, S + o_fill, seg);
  BA_LAUNCH(k_setup_sort_pt, gp, b256, 0, h->stream, (const int*)h->pt_off.p, Np, (const int*)seg, (const int*)d_cam, p_src, h->p_cam.p, p_pt,
            S + o_hist, SETUP_HIST_BINS, u64, S + o_words + 1, S + o_big, S + o_words + 2);
  BA_LAUNCH(k_setup_sort_big, dim3(std::min(Np, 2048)), b256, 0, h->stream, (const int*)(S + o_big), (const int*)(S + o_words + 2),
            (const int*)h->pt_off.p, (const int*)seg, (const int*)d_cam, p_src, h->p_cam.p, p_pt, u64, S + o_words + 1);
  // read back: histogram of track lengths, words, span sum
  int* hh = (int*)h->h_setup;
  HIPCHECK(hipMemcpyAsync(hh, S + o_hist, (SETUP_HIST_BINS + 16) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(hh + SETUP_HIST_BINS + 16, u64, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  stage("point order");
  const int bad = hh[SETUP_HIST_BINS + 0], n_tracks = hh[SETUP_HIST_BINS + 1];
  if (bad >= 0 && bad < No) {
    if (cam_idx[bad] < 0 || cam_idx[bad] >= Nc) return fail(BA_ERR_INVALID, "cam_idx[%lld]=%d out of range", (long long)bad, cam_idx[bad]);
    return fail(BA_ERR_INVALID, "pt_idx[%lld]=%d out of range", (long long)bad, pt_idx[bad]);
  }
  h->uv_f32 = uv_f32_wanted() && Nc > SMALL_MAX_CAMS && hh[SETUP_HIST_BINS + 3] == 0;
  unsigned long long span_sum;
  memcpy(&span_sum, hh + SETUP_HIST_BINS + 16, sizeof span_sum);
  // median track length = element Np / 2 of the sorted lengths
  int med = 0;
  { long long cum = 0; for (int L = 0; L < SETUP_HIST_BINS; ++L) { cum += hh[L]; if (cum > Np / 2) { med = L; break; } } }
  config_long_threshold(h, med);
  if (h->long_thr != 0x7fffffff && h->long_thr >= SETUP_HIST_BINS - 1) return 1;       // (the capped histogram cannot count those)
  int n_long = 0;
  if (h->long_thr != 0x7fffffff) for (int L = h->long_thr + 1; L < SETUP_HIST_BINS; ++L) n_long += hh[L];
  config_long_grid(h, grid, Np, n_long);
  if (int rc = decide_banded(h, (double)span_sum, (double)n_tracks, Nc)) return rc;
  {   // problems the two-level preconditioner's structures would be built for: the host build (it also re-sorts every track by camera)
    bool want = h->banded && !h->multi;
    if (const char* e = getenv("BA_TWO_LEVEL")) want = atoi(e) != 0 && !h->multi;
    if (want && Nc >= 2 * VEC_CAMS) return 1;
  }
  h->two_level_ok = false;
  h->mw_ok = false;
  h->one_part = false;
  HIPCHECK(h->long_pts.alloc(std::max(h->n_long, 1)));
  if (h->n_long > 0) {
    BA_LAUNCH(k_setup_long_flags, gp, b256, 0, h->stream, (const int*)h->pt_off.p, Np, h->long_thr, S + o_flag);
    dev_scan(h, S + o_flag, Np, S + o_bsum, S + o_pos);
    BA_LAUNCH(k_setup_long_list, gp, b256, 0, h->stream, (const int*)h->pt_off.p, Np, h->long_thr, (const int*)(S + o_pos), h->long_pts.p);
  }
  if (h->lanes == LPP && !getenv("BA_NO_BANK_ORDER")) {
    const long long nthreads = (long long)h->nblkP * ((h->ppb + 15) / 16) * 2;
    BA_LAUNCH(k_setup_bank_order, dim3((unsigned)((nthreads + 255) / 256)), b256, 0, h->stream, (const int*)h->pt_off.p, Np, h->nblkP, h->ppb,
              h->p_cam.p, p_src);
  }
  stage("long tracks + visiting order");
  // camera order (keys: positions of the point-ordered list)
  const dim3 gt((No + SETUP_CAM_TILE - 1) / SETUP_CAM_TILE);
  BA_LAUNCH(k_setup_hist_cam, gt, b256, (size_t)Nc * sizeof(int), h->stream, (const int*)h->p_cam.p, No, Nc, S + o_ccnt);
  dev_scan(h, S + o_ccnt, Nc, S + o_bsum, S + o_coff);
  BA_LAUNCH(k_setup_scatter_cam, gt, b256, 2 * (size_t)Nc * sizeof(int), h->stream, (const int*)h->p_cam.p, No, Nc, (const int*)(S + o_coff),
            S + o_cfill, seg);
  BA_LAUNCH(k_setup_sort_cam, dim3(Nc), b256, 0, h->stream, (const int*)(S + o_coff), (const int*)seg, (const int*)p_pt, (const int*)p_src,
            h->c_pt.p, h->c_orig.p);
  BA_LAUNCH(k_setup_offk, dim3((Nc * (NPART + 1) + 255) / 256), b256, 0, h->stream, (const int*)(S + o_coff), Nc, h->offk.p);
  BA_LAUNCH(k_setup_xcd_stat, dim3(std::min((Nc * NPART + 3) / 4, 2 * h->n_cu)), b256, 0, h->stream, (const int*)h->offk.p, (const int*)h->c_pt.p, Nc, Np, u64 + 1);
  const int nwin = h->nblkP + h->nblkL;
  if ((size_t)nwin * sizeof(int2) + 64 > SETUP_PINNED_BYTES) return 1;
  HIPCHECK(h->blk_win.alloc(nwin));
  BA_LAUNCH(k_setup_windows, dim3(nwin), b256, 0, h->stream, (const int*)h->pt_off.p, (const int*)h->p_cam.p, Np, Nc, h->nblkP, h->ppb,
            (const int*)h->long_pts.p, h->n_long, h->long_spb, h->blk_win.p);
  HIPCHECK(hipMemcpyAsync(h->h_setup, u64 + 1, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(h->h_setup + 64, h->blk_win.p, (size_t)nwin * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
  // pixels into both orderings, flagged index copies, identity point numbering
  BA_LAUNCH(k_gather_uv, go, b256, 0, h->stream, (const double2*)h->rbuf.p, (const int*)p_src, No, h->p_uv.p, (int)h->uv_f32);
  BA_LAUNCH(k_gather_uv, go, b256, 0, h->stream, (const double2*)h->rbuf.p, (const int*)h->c_orig.p, No, h->c_uv.p, (int)h->uv_f32);
  BA_LAUNCH(k_init_flagged, go, b256, 0, h->stream, (const int*)h->c_pt.p, (const int*)h->p_cam.p, No, h->c_ptf[0].p, h->c_ptf[1].p,
            h->p_camf[0].p, h->p_camf[1].p);
  BA_LAUNCH(k_setup_iota, gp, b256, 0, h->stream, h->slot.p, Np);
  if (int rc = alloc_solver_buffers(h)) return rc;
  h->lb = 0;
  BA_SYNC(h);
  stage("camera order, windows, pixels");
  unsigned long long st2[2];
  memcpy(st2, h->h_setup, sizeof st2);
  h->cam_band = st2[1] > st2[0];
  if (const char* e = getenv("BA_CAM_BAND")) h->cam_band = atoi(e) != 0;
  {
    const int2* win = (const int2*)(h->h_setup + 64);
    size_t max_win[2] = {0, 0};
    const size_t row_bytes[2] = {Pinhole::TA * sizeof(double), BalCam::TA * sizeof(double)};
    h->all_lds_m[0] = h->all_lds_m[1] = true;
    for (int b = 0; b < nwin; ++b)
      for (int m = 0; m < 2; ++m) {
        const size_t bytes = (size_t)win[b].y * row_bytes[m];
        if (bytes <= (size_t)LDS_TAB_BYTES) max_win[m] = std::max(max_win[m], bytes);
        else h->all_lds_m[m] = false;
      }
    h->lds_bytes_m[0] = max_win[0]; h->lds_bytes_m[1] = max_win[1];
  }
  h->setup_path = 1;
  return BA_OK;
}

extern "C" int ba_set_problem(ba_handle* h, int32_t n_cams, int32_t n_pts, int64_t n_obs, const int32_t* cam_idx,
                              const int32_t* pt_idx_in, const double* uv, const double K4[4], int32_t fixed_cam) {
  const int32_t* pt_idx = pt_idx_in;
  // BA_TIME_SETUP=1: stage times of this call on stderr (host sorts are the bulk of it at C3)
  const bool timed = getenv("BA_TIME_SETUP") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* name) {
    if (!timed) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "ba_set_problem %-28s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (n_cams <= 0 || n_pts < 0 || n_obs < 0 || n_obs > 0x7fffffffLL) return fail(BA_ERR_INVALID, "bad sizes");
  if (n_obs > 0 && (!cam_idx || !pt_idx || !uv)) return fail(BA_ERR_INVALID, "null observation arrays");
  if (!K4) return fail(BA_ERR_INVALID, "null intrinsics");
  if (fixed_cam < -1 || fixed_cam >= n_cams) return fail(BA_ERR_INVALID, "fixed_cam %d out of range", fixed_cam);
  {   // index ranges, before anything is touched (a rejected call keeps the previous problem): branch-free sweep first
    int ok = 1;
    for (int64_t i = 0; i < n_obs; ++i)
      ok &= (int)((unsigned)cam_idx[i] < (unsigned)n_cams) & (int)((unsigned)pt_idx[i] < (unsigned)n_pts);
    if (!ok)
      for (int64_t i = 0; i < n_obs; ++i) {
        if (cam_idx[i] < 0 || cam_idx[i] >= n_cams) return fail(BA_ERR_INVALID, "cam_idx[%lld]=%d out of range", (long long)i, cam_idx[i]);
        if (pt_idx[i] < 0 || pt_idx[i] >= n_pts) return fail(BA_ERR_INVALID, "pt_idx[%lld]=%d out of range", (long long)i, pt_idx[i]);
      }
  }
  if ((n_cams + VEC_CAMS - 1) / VEC_CAMS > 16384) return fail(BA_ERR_INVALID, "more than %d cameras are not supported", 16384 * VEC_CAMS);
  stage("validate");
  // float2 pixel streams (UvArr): only problems the window solvers never take (they read double2), and only when every
  // pixel is a float32 value; the device build looks for itself (k_setup_hist), the host build sweeps them below
  const bool uv_f32_eligible = uv_f32_wanted() && n_cams > SMALL_MAX_CAMS && n_obs > 0;
  if (set_device(h)) return BA_ERR_HIP;
  // From here on the previous problem is gone: should anything below fail (allocation, copy), the handle is left
  // WITHOUT a problem rather than with new sizes over old buffers.
  h->have_problem = false;
  h->have_params = false;
  h->linearized = false;
  h->setup_path = 0;
  h->banded_known = false;
  h->uv_f32 = false;
  // Large problems whose camera table fits in LDS are laid out ON THE DEVICE (set_problem_device, ba_setup.hpp: one upload
  // of the caller's arrays, no host sorts; bit-equal result).  BA_SETUP=host / device forces a path (device: whenever the
  // problem qualifies at all), BA_SETUP_DEVICE_MIN moves the size from which it is chosen (default 50 000 observations:
  // the measured crossover is near 40 000 -- 0.23 against 0.27 ms at 48 k, 0.40 against 1.05 ms at 160 k).
  {
    static const long dev_min = [] { const char* e = getenv("BA_SETUP_DEVICE_MIN"); return e ? atol(e) : 50000L; }();
    const char* mode = getenv("BA_SETUP");
    bool try_dev = n_obs > 0 && n_pts > 0 && n_cams > MW_MAX_CAMS && (size_t)n_cams * TA * sizeof(double) <= (size_t)LDS_TAB_BYTES &&
                   !(getenv("BA_ONE_PART") && atoi(getenv("BA_ONE_PART")) != 0 && h->multi);
    if (mode && strcmp(mode, "host") == 0) try_dev = false;
    else if (!(mode && strcmp(mode, "device") == 0) && n_obs < dev_min) try_dev = false;
    if (try_dev) {
      h->small_np_pad = -1;
      const int rc = set_problem_device(h, n_cams, n_pts, (int)n_obs, cam_idx, pt_idx, uv, K4, fixed_cam, timed);
      if (rc < 0) { const std::string msg = g_err; (void)hipStreamSynchronize(h->stream); h->launch_err = hipSuccess; g_err = msg; return rc; }
      if (rc == BA_OK) {
        stage("device build (total)");
        h->have_problem = true;
        return BA_OK;
      }
      (void)hipStreamSynchronize(h->stream);        // rc == 1: this problem is for the host build
      h->launch_err = hipSuccess;
    }
  }
  // the window solver keeps V = W L resident and only ever writes the columns of points that exist: a new problem on
  // the same handle (fewer landmarks inside the same 16-column padding, or other cameras per point) must not inherit
  // columns of the previous one -> V is cleared again before its next use
  h->small_np_pad = -1;
  const int Nc = n_cams, Np = n_pts, No = (int)n_obs;
  h->uv_f32 = false;           // (a device build that handed the problem back may have decided already: decided again here)
  if (uv_f32_eligible) {       // every pixel a float32 value?  (branch-free sweep; NaN compares unequal: stays double)
    int all = 1;
    for (int64_t i = 0; i < 2 * n_obs; ++i) all &= (int)pixel_is_f32(uv[i]);
    h->uv_f32 = all != 0;
    stage("pixel value sweep");
  }
  // internal point numbering.  When the whole camera table fits in LDS nothing is gained by
  // moving points, so the caller's order is kept.  Otherwise points are sorted by the mean index of
  // the cameras that observe them: consecutive points are then seen from a narrow window of cameras
  // whenever the data has that locality (and per-camera partitions stay balanced when it has not).
  // Pure locality: results do not depend on it.
  std::vector<int> slot(Np);
  if ((size_t)Nc * TA * sizeof(double) <= (size_t)LDS_TAB_BYTES) {
    for (int p = 0; p < Np; ++p) slot[p] = p;
  } else {
    std::vector<double> sum(Np, 0.0);
    std::vector<int> n(Np, 0);
    for (int i = 0; i < No; ++i) { sum[pt_idx[i]] += cam_idx[i]; n[pt_idx[i]]++; }
    // The order a stable sort of the points by key gives, in two linear steps instead of a comparison sort with an indirect
    // key load per comparison (config 5: 9 of this stage's 13 ms): the keys lie in [0, Nc], so a stable counting sort by
    // floor(16 key) -- monotone in the key -- leaves a handful of points per bucket, finished by a stable insertion sort on
    // the exact keys
    constexpr int BK = 16;
    const size_t nbk = (size_t)(Nc + 1) * BK + 2;
    std::vector<int> bfirst(nbk + 1, 0);
    std::vector<std::pair<double, int>> order(Np);
    for (int p = 0; p < Np; ++p) {
      const double key = n[p] ? sum[p] / n[p] : (double)Nc;
      sum[p] = key;
      bfirst[(size_t)(key * BK) + 1]++;
    }
    for (size_t q = 0; q < nbk; ++q) bfirst[q + 1] += bfirst[q];
    {
      std::vector<int> fill(bfirst.begin(), bfirst.end() - 1);
      for (int p = 0; p < Np; ++p) order[fill[(size_t)(sum[p] * BK)]++] = std::make_pair(sum[p], p);
    }
    for (size_t q = 0; q < nbk; ++q)
      for (int a = bfirst[q] + 1; a < bfirst[q + 1]; ++a) {
        const std::pair<double, int> v = order[a];
        int w = a;
        while (w > bfirst[q] && order[w - 1].first > v.first) { order[w] = order[w - 1]; --w; }
        order[w] = v;
      }
    for (int r = 0; r < Np; ++r) slot[order[r].second] = r;
  }
  std::vector<int> pt_new(No);
  for (int i = 0; i < No; ++i) pt_new[i] = slot[pt_idx[i]];
  pt_idx = pt_new.data();
  stage("point numbering");
  // point-pass workgroups: contiguous point ranges, PT_THREADS / LPP points per round.  When the
  // whole camera table fits in LDS (so a wider range cannot overflow it) no more workgroups are
  // started than the chip holds at once -- each then walks several rounds with one table fill
  // (C3 x 10: Schur point pass 133 -> 93 us); with camera windows the ranges stay one round long
  // so that the windows stay narrow.  BA_PT_BLOCKS overrides (tuning only).
  // A smaller problem (a sliding window, a shard of a multi-GPU job) gives every point 4, 8 or 16 lanes
  // instead of 2 -- more workgroups, fewer observations per lane -- the most for which the
  // workgroups are still all resident at once.  BA_PT_LANES overrides (tuning only).
  const PtGrid grid = config_point_grid(h, Nc, Np, No);
  const bool table_fits = grid.table_fits;
  const int pts_per_pass = grid.pts_per_pass, want = grid.want;
  (void)pts_per_pass; (void)want;
  // ---- bank-aware visiting order inside a point (2-lane point passes with the camera table in LDS).
  // A point pass reads a camera's 144-byte LDS row with nine ds_read_b128; the hardware serves such a read in groups of
  // 16 lanes, and two lanes of a group collide when their rows fall into the same of 16 bank classes (row mod 16: the
  // row stride is 36 dwords).  With random cameras a group sees ~3 lanes per class: SQ_LDS_BANK_CONFLICT was 64 % of
  // the LDS cycles (profiles/).  The ORDER in which a point's observations are visited is free, so it is chosen here,
  // greedily per group of eight points and per step, so that the sixteen rows read together are in distinct classes
  // wherever the data allows.  Pure scheduling: every sum keeps a fixed order, results stay bit-reproducible.
  auto bank_aware_order = [&](std::vector<int>& p_cam, std::vector<int>& p_src, const std::vector<int>& pt_off) {
    if (h->lanes != LPP || !table_fits || getenv("BA_NO_BANK_ORDER")) return;
    // ds_read_b128 is served in groups of SIXTEEN CONSECUTIVE LANES (measured on MI355X, tools/microbench/lds_b128_groups.hip:
    // rows with distinct bank classes inside every 16 consecutive lanes read as fast as a broadcast, 14.3 cycles per
    // instruction against 23.8 for random rows; distinct classes inside the lane sets {0-3,12-15,20-27} / {4-11,16-19,28-31}
    // that rounds 1-3 ordered for -- the guide's grouping -- still cost 19.9).  With 2 lanes per point: points 0-7 of a
    // 16-point chunk are one group, points 8-15 the other
    static const int group_of_pair[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};
    std::vector<int> tmp_c, tmp_s;
    for (int b = 0; b < h->nblkP; ++b) {
      const int p0 = std::min(Np, b * h->ppb), p1 = std::min(Np, (b + 1) * h->ppb);
      for (int c0 = p0; c0 < p1; c0 += 16) {              // one 32-lane half: 16 points, two groups of 8
        for (int g = 0; g < 2; ++g) {
          int pts[8], npts = 0;
          for (int q = 0; q < 16 && c0 + q < p1; ++q)
            if (group_of_pair[((c0 - p0) + q) & 15] == g) pts[npts++] = c0 + q;
          int maxlen = 0;
          for (int i = 0; i < npts; ++i) maxlen = std::max(maxlen, pt_off[pts[i] + 1] - pt_off[pts[i]]);
          if (maxlen > 64) continue;                      // (long tracks: their own launch, other mapping)
          // remaining observations of each point as a small list; at every step each point places up to two
          int cur[8];
          for (int i = 0; i < npts; ++i) cur[i] = pt_off[pts[i]];
          for (int step = 0; 2 * step < maxlen; ++step) {
            unsigned used = 0;                            // bank classes taken in this step
            for (int i = 0; i < npts; ++i) {
              const int end = pt_off[pts[i] + 1];
              for (int sub = 0; sub < 2 && cur[i] < end; ++sub) {
                int pick = cur[i];
                for (int j = cur[i]; j < end; ++j)
                  if (!(used >> (p_cam[j] & 15) & 1u)) { pick = j; break; }
                used |= 1u << (p_cam[pick] & 15);
                std::swap(p_cam[pick], p_cam[cur[i]]);
                std::swap(p_src[pick], p_src[cur[i]]);
                ++cur[i];
              }
            }
          }
        }
      }
    }
  };
  // point order: stable counting sort by point (keeps the caller's order inside a point)
  std::vector<int> pt_off(Np + 1, 0);
  for (int i = 0; i < No; ++i) pt_off[pt_idx[i] + 1]++;
  for (int p = 0; p < Np; ++p) pt_off[p + 1] += pt_off[p];
  // (only the indices are permuted on the host; the pixels follow on the device, k_gather_uv)
  std::vector<int> p_cam(No), p_src(No);
  {
    std::vector<int> pc(pt_off.begin(), pt_off.end() - 1);
    for (int i = 0; i < No; ++i) {
      const int b = pc[pt_idx[i]]++;
      p_cam[b] = cam_idx[i]; p_src[b] = i;
    }
  }
  // long tracks: one DPP row (16 lanes) per point in a launch of their own
  std::vector<int> long_pts;
  {   // long = more than max(8, 2 x median track length) observations (BA_LONG_TRACK overrides)
    std::vector<int> len(Np);
    for (int p = 0; p < Np; ++p) len[p] = pt_off[p + 1] - pt_off[p];
    int med = 0;
    if (Np > 0) { std::nth_element(len.begin(), len.begin() + Np / 2, len.end()); med = len[Np / 2]; }
    config_long_threshold(h, med);
  }
  for (int p = 0; p < Np; ++p) if (pt_off[p + 1] - pt_off[p] > h->long_thr) long_pts.push_back(p);
  config_long_grid(h, grid, Np, (int)long_pts.size());
  const int long_per_blk = h->long_spb;
  // (the visiting order below is laid out for the ranges' final length)
  bank_aware_order(p_cam, p_src, pt_off);
  stage("sort by point");
  // camera order: stable counting sort of the POINT-ordered list by camera, so that every
  // camera's observations are ascending in point index (needed by the partition split)
  std::vector<int> cam_off(Nc + 1, 0);
  for (int i = 0; i < No; ++i) cam_off[cam_idx[i] + 1]++;
  for (int c = 0; c < Nc; ++c) cam_off[c + 1] += cam_off[c];
  std::vector<int> c_pt(No), c_orig(No);
  {
    std::vector<int> cc(cam_off.begin(), cam_off.end() - 1);
    for (int p = 0; p < Np; ++p)
      for (int j = pt_off[p]; j < pt_off[p + 1]; ++j) {
        const int a = cc[p_cam[j]]++;
        c_pt[a] = p; c_orig[a] = p_src[j];
      }
  }
  stage("sort by camera");
  // partition split: every camera's (point-sorted) list is cut into NPART equal-count chunks.  For
  // uniformly spread observations chunk k covers about the k-th eighth of the point table (what
  // keeps it resident in XCD k's L2); for band-structured data the chunks stay balanced and are
  // narrow in point index anyway.
  // BA_ONE_PART=1 (experiment, multi-rank only): the whole local list of a camera goes into partition 0, the others stay
  // empty -- the partial sums then come out already "folded" and the fold kernels in front of the all-reduces
  // (fold_and_reduce) disappear.  Measured on a rank's share of C3 with every collective issued (tools/shard_times.py):
  // slower at every shard size (1/8 of the points: 240 against 182 us per LM iteration, 1/4: 279 against 184) -- a
  // camera's list handled by ONE 16-lane row in the camera passes costs more than the 3 us fold it saves.  Off by default.
  h->one_part = false;
  if (const char* e = getenv("BA_ONE_PART")) h->one_part = atoi(e) != 0 && h->multi;
  std::vector<int> offk((size_t)Nc * (NPART + 1));
  for (int c = 0; c < Nc; ++c) {
    const long long n = cam_off[c + 1] - cam_off[c];
    for (int k = 0; k <= NPART; ++k)
      offk[(size_t)c * (NPART + 1) + k] = h->one_part ? (k == 0 ? cam_off[c] : cam_off[c + 1]) : cam_off[c] + (int)((n * k) / NPART);
  }
  // which workgroup -> XCD assignment of the camera passes keeps an XCD on one slice of the point table
  // (group_of_block): count the observations whose point lies in the slice of their partition, and in
  // the slice of their camera's range
  {
    long long in_partition = 0, in_band = 0;
    if (Np > 0) {
      std::vector<unsigned char> slice_of(Np);           // p * NPART / Np without a division per observation
      for (int k = 0; k < NPART; ++k)
        for (long long q = ((long long)k * Np + NPART - 1) / NPART; q < ((long long)(k + 1) * Np + NPART - 1) / NPART; ++q)
          slice_of[q] = (unsigned char)k;
      for (int c = 0; c < Nc; ++c) {
        const int cam_slice = (int)(((long long)c * NPART) / Nc);
        for (int k = 0; k < NPART; ++k)
          for (int a = offk[(size_t)c * (NPART + 1) + k]; a < offk[(size_t)c * (NPART + 1) + k + 1]; ++a) {
            const int pt_slice = slice_of[c_pt[a]];
            in_partition += pt_slice == k;
            in_band += pt_slice == cam_slice;
          }
      }
    }
    h->cam_band = in_band > in_partition;
    if (const char* e = getenv("BA_CAM_BAND")) h->cam_band = atoi(e) != 0;
  }
  stage("partitions + XCD statistic");
  // ---- two-level preconditioner structures (ba_coarse.hpp), for band-structured problems on one rank: observations
  // inside a point sorted by camera, so that a point's observations of one 16-camera aggregate are one RUN
  std::vector<int> run_beg, run_pt, run_agg;
  std::vector<int2> run_pairs;
  h->two_level_ok = false;
  {
    // band statistic: mean camera span of a track against the number of cameras (sequential captures: a few
    // cameras; random visibility: most of the range)
    double span_sum = 0.0;
    long long tracks = 0;
    for (int p = 0; p < Np; ++p) {
      if (pt_off[p + 1] == pt_off[p]) continue;
      int lo = Nc, hi = -1;
      for (int j = pt_off[p]; j < pt_off[p + 1]; ++j) { lo = std::min(lo, p_cam[j]); hi = std::max(hi, p_cam[j]); }
      span_sum += hi - lo;
      ++tracks;
    }
    if (!h->banded_known) { if (int rc = decide_banded(h, span_sum, (double)tracks, Nc)) return rc; }
    bool want = h->banded && !h->multi;
    if (const char* e = getenv("BA_TWO_LEVEL")) want = atoi(e) != 0 && !h->multi;
    if (want && Nc >= 2 * VEC_CAMS && Np > 0 && No > 0) {
      std::vector<std::pair<int, int>> tmp;
      int bwa = 0;
      run_beg.reserve(No / 2 + 16); run_pt.reserve(No / 2 + 16); run_agg.reserve(No / 2 + 16); run_pairs.reserve(No);
      for (int p = 0; p < Np; ++p) {
        const int b = pt_off[p], e = pt_off[p + 1];
        if (e - b <= 32) {             // the usual track: stable insertion sort by camera, in place
          for (int j = b + 1; j < e; ++j) {
            const int c = p_cam[j], sidx = p_src[j];
            int q = j;
            while (q > b && p_cam[q - 1] > c) { p_cam[q] = p_cam[q - 1]; p_src[q] = p_src[q - 1]; --q; }
            p_cam[q] = c; p_src[q] = sidx;
          }
        } else {
          tmp.resize(e - b);
          for (int j = b; j < e; ++j) tmp[j - b] = std::make_pair(p_cam[j], p_src[j]);
          std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
          for (int j = b; j < e; ++j) { p_cam[j] = tmp[j - b].first; p_src[j] = tmp[j - b].second; }
        }
        const int first_run = (int)run_pt.size();
        for (int j = b; j < e;) {
          const int a = p_cam[j] / VEC_CAMS;
          run_beg.push_back(j); run_pt.push_back(p); run_agg.push_back(a);
          while (j < e && p_cam[j] / VEC_CAMS == a) ++j;
        }
        const int last_run = (int)run_pt.size();
        for (int r1 = first_run; r1 < last_run; ++r1)
          for (int r2 = r1; r2 < last_run; ++r2) {
            run_pairs.push_back(make_int2(r1, r2));
            bwa = std::max(bwa, run_agg[r2] - run_agg[r1]);
          }
      }
      run_beg.push_back(No);
      h->n_agg = (Nc + VEC_CAMS - 1) / VEC_CAMS;
      h->n_runs = (int)run_pt.size();
      h->n_pairs = (int)run_pairs.size();
      h->coarse_bw = std::min(6 * h->n_agg - 1, 6 * bwa + 5);
      // the banded factorisation keeps one column of the band in LDS (512 words) and E^-1 is dense: give up the
      // coarse level when the problem is not band-structured enough for either
      h->two_level_ok = h->coarse_bw <= 500 && 6 * h->n_agg <= 4096;
    }
  }
  stage("coarse-level structures");
  h->Nc = Nc; h->Np = Np; h->Nobs = No; h->fixed = fixed_cam;
  memcpy(h->K4, K4, sizeof h->K4);
  h->nblkV = (Nc + VEC_CAMS - 1) / VEC_CAMS;
  h->nblkVm[0] = (Nc + Pinhole::VC - 1) / Pinhole::VC; h->nblkVm[1] = (Nc + BalCam::VC - 1) / BalCam::VC;
  const size_t nbv_max = (size_t)std::max(h->nblkVm[0], h->nblkVm[1]);
  std::vector<int2> win(h->nblkP + h->nblkL);
  // a window is staged in LDS when its rows fit; the row stride depends on the camera model (18 doubles for the
  // reference's pinhole, 26 for the BAL camera), so the LDS size and the "every window fits" flag are kept per model
  size_t max_win[2] = {0, 0};
  const size_t row_bytes[2] = {Pinhole::TA * sizeof(double), BalCam::TA * sizeof(double)};
  h->all_lds_m[0] = h->all_lds_m[1] = true;
  auto window_of = [&](int lo, int hi) {
    if (hi < lo) { lo = 0; hi = -1; }
    for (int m = 0; m < 2; ++m) {
      const size_t bytes = (size_t)(hi - lo + 1) * row_bytes[m];
      if (bytes <= (size_t)LDS_TAB_BYTES) max_win[m] = std::max(max_win[m], bytes);
      else h->all_lds_m[m] = false;
    }
    return make_int2(lo, hi - lo + 1);
  };
  for (int b = 0; b < h->nblkP; ++b) {
    const int p0 = std::min(Np, b * h->ppb), p1 = std::min(Np, (b + 1) * h->ppb);
    int lo = Nc, hi = -1;
    for (int j = pt_off[p0]; j < pt_off[p1]; ++j) { lo = std::min(lo, p_cam[j]); hi = std::max(hi, p_cam[j]); }
    win[b] = window_of(lo, hi);
  }
  for (int b = 0; b < h->nblkL; ++b) {
    int lo = Nc, hi = -1;
    for (int q = b * long_per_blk; q < std::min(h->n_long, (b + 1) * long_per_blk); ++q)
      for (int j = pt_off[long_pts[q]]; j < pt_off[long_pts[q] + 1]; ++j) { lo = std::min(lo, p_cam[j]); hi = std::max(hi, p_cam[j]); }
    win[h->nblkP + b] = window_of(lo, hi);
  }
  h->lds_bytes_m[0] = max_win[0]; h->lds_bytes_m[1] = max_win[1];
  stage("long tracks + windows");
  if (timed)
    fprintf(stderr, "ba_set_problem point passes: %d lanes/point, %d range workgroups x %d points + %d long-track workgroups x %d points "
            "(%d points over %d observations), LDS window %zu / %zu bytes, every window in LDS %d / %d\n", h->lanes, h->nblkP, h->ppb,
            h->nblkL, h->long_spb, h->n_long, h->long_thr, max_win[0], max_win[1], (int)h->all_lds_m[0], (int)h->all_lds_m[1]);
  const size_t nobs1 = std::max(No, 1), np1 = std::max(Np, 1);
  // Index uploads.  A window-sized problem (the reference's own use: a few thousand observations) sends a dozen small
  // arrays; copied from pageable memory each is a blocking staged transfer (~10 us), so they go through ONE pinned
  // staging block and truly asynchronous copies.  Large problems (C3: 36 MB) keep the direct path.
  size_t up_used = 0;
  const size_t up_need = ((size_t)No * (4 * 4 + 16) + (size_t)Np * 12 + (size_t)Nc * 40 + win.size() * 8 + 4096) * 1;
  const bool staged = up_need <= ((size_t)4 << 20);
  if (h->up_pending) { (void)hipEventSynchronize(h->up_event); h->up_pending = false; }      // the arena's last upload has landed
  if (staged && h->h_up_cap < up_need) {       // grow-only, with headroom: consecutive windows differ a little in size
    if (h->h_up) { (void)hipHostFree(h->h_up); h->h_up = nullptr; h->h_up_cap = 0; }
    const size_t cap = std::min<size_t>((size_t)4 << 20, std::max<size_t>(2 * up_need, (size_t)256 << 10));
    HIPCHECK(hipHostMalloc((void**)&h->h_up, cap, hipHostMallocDefault));
    h->h_up_cap = cap;
  }
  // (staged: the sections are only collected here; ONE copy of the arena and ONE kernel deal them out at the end)
  UnpackArgs ua;
  memset(&ua, 0, sizeof ua);
  auto upload = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (bytes == 0) return hipSuccess;
    if (staged && up_used + bytes <= h->h_up_cap && ua.n_sections < UNPACK_MAX_SECTIONS && bytes % 4 == 0) {
      memcpy(h->h_up + up_used, src, bytes);
      ua.off[ua.n_sections] = up_used; ua.dst[ua.n_sections] = (int*)dst; ua.words[ua.n_sections] = (int)(bytes / 4);
      ++ua.n_sections;
      up_used += (bytes + 63) & ~(size_t)63;
      return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
  };
  // a section that only the unpack kernel reads (no array of its own): returns its arena offset, or (size_t)-1
  auto stash = [&](const void* src, size_t bytes) -> size_t {
    if (!(staged && up_used + bytes <= h->h_up_cap)) return (size_t)-1;
    memcpy(h->h_up + up_used, src, bytes);
    const size_t o = up_used;
    up_used += (bytes + 63) & ~(size_t)63;
    return o;
  };
  HIPCHECK(h->offk.alloc((size_t)Nc * (NPART + 1))); HIPCHECK(h->pt_off.alloc(Np + 1));
  HIPCHECK(h->slot.alloc(np1));
  if (Np > 0) HIPCHECK(upload(h->slot.p, slot.data(), Np * sizeof(int)));
  HIPCHECK(h->blk_win.alloc(win.size()));
  HIPCHECK(upload(h->blk_win.p, win.data(), win.size() * sizeof(int2)));
  HIPCHECK(h->long_pts.alloc(std::max(h->n_long, 1)));
  if (h->n_long) HIPCHECK(upload(h->long_pts.p, long_pts.data(), h->n_long * sizeof(int)));
  // the multi-workgroup window solver (ba_small_mw.hpp): eight cameras at most, up to 2048 landmarks in ranges of 64, no
  // landmark seen twice by one camera; every camera's list is ascending in landmark index, so a range is a slice of it
  h->mw_ok = false;
  const char* mw_min_env = getenv("BA_SMALL_MW_MIN");            // (tuning: smallest landmark count that goes to k_small_mw)
  const int mw_min_pts = mw_min_env ? std::max(1, atoi(mw_min_env)) : MW_MIN_PTS;
  if (!h->multi && Nc <= MW_MAX_CAMS && Np >= mw_min_pts && Np <= MW_MAX_WG * MW_PTS && No > 0) {
    bool dup = false;
    for (int p = 0; p < Np && !dup; ++p) {
      unsigned seen = 0;
      for (int j = pt_off[p]; j < pt_off[p + 1]; ++j) { if ((seen >> p_cam[j]) & 1u) { dup = true; break; } seen |= 1u << p_cam[j]; }
    }
    if (!dup) {
      h->mw_G = (Np + MW_PTS - 1) / MW_PTS;
      std::vector<int> woff((size_t)(h->mw_G + 1) * MW_MAX_CAMS, 0);
      for (int c = 0; c < MW_MAX_CAMS; ++c)
        for (int gq = 0; gq <= h->mw_G; ++gq) {
          int v = No;
          if (c < Nc) v = (int)(std::lower_bound(c_pt.begin() + cam_off[c], c_pt.begin() + cam_off[c + 1], gq * MW_PTS) - c_pt.begin());
          woff[(size_t)gq * MW_MAX_CAMS + c] = v;
        }
      HIPCHECK(h->mw_woff.alloc(woff.size()));
      HIPCHECK(upload(h->mw_woff.p, woff.data(), woff.size() * sizeof(int)));
      HIPCHECK(h->mw_buf.alloc((size_t)2 * MW_MAX_WG * (MW_MSG + MW_SCAL) + 8));
      // the workgroups of k_small_mw meet at a counter barrier: the kernel is only chosen when the device can hold all of
      // them at once (occupancy query x compute units; the launch itself is an ordinary one, and a barrier that is not
      // served in time -- somebody else holds the units -- ends in the fall-back of small_solve, not in a hang)
      const int nt = Nc <= 5 ? 0 : (Nc <= 7 ? 1 : 2);
      if (h->mw_resident[nt] < 0) {
        int per_cu = 0;
        hipError_t e = nt == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_small_mw<2>, MW_THREADS, 0)
                     : nt == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_small_mw<3>, MW_THREADS, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_small_mw<4>, MW_THREADS, 0);
        if (e != hipSuccess) { per_cu = 0; (void)hipGetLastError(); }
        h->mw_resident[nt] = per_cu * h->n_cu;
      }
      h->mw_ok = h->mw_resident[nt] >= h->mw_G;
    }
  }
  if (int rc = alloc_solver_buffers(h)) return rc;
  if (h->two_level_ok) {
    const size_t nc6 = 6 * (size_t)h->n_agg;
    HIPCHECK(h->run_beg.alloc(run_beg.size())); HIPCHECK(h->run_pt.alloc(run_pt.size())); HIPCHECK(h->run_agg.alloc(run_agg.size()));
    HIPCHECK(h->run_pairs.alloc(run_pairs.size()));
    HIPCHECK(hipMemcpyAsync(h->run_beg.p, run_beg.data(), run_beg.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->run_pt.p, run_pt.data(), run_pt.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->run_agg.p, run_agg.data(), run_agg.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->run_pairs.p, run_pairs.data(), run_pairs.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h->coarseU.alloc(18 * (size_t)h->n_runs)); HIPCHECK(h->coarseE.alloc(nc6 * nc6)); HIPCHECK(h->coarseEinv.alloc(nc6 * nc6));
    HIPCHECK(h->coarseEint.alloc(nc6 * nc6)); HIPCHECK(h->coarse_rc.alloc(nc6)); HIPCHECK(h->coarse_info.alloc(8));
  }
  h->lb = 0;
  stage("allocations");
  HIPCHECK(upload(h->offk.p, offk.data(), offk.size() * sizeof(int)));
  HIPCHECK(upload(h->pt_off.p, pt_off.data(), (Np + 1) * sizeof(int)));
  bool packed = false;         // the window-sized form: everything through the arena and k_unpack_problem
  if (No > 0 && staged) {
    const int s_cpt = ua.n_sections;
    HIPCHECK(upload(h->c_pt.p, c_pt.data(), No * sizeof(int)));
    const int s_corig = ua.n_sections;
    HIPCHECK(upload(h->c_orig.p, c_orig.data(), No * sizeof(int)));
    const int s_pcam = ua.n_sections;
    HIPCHECK(upload(h->p_cam.p, p_cam.data(), No * sizeof(int)));
    const size_t o_uv = stash(uv, 2 * (size_t)No * sizeof(double)), o_psrc = stash(p_src.data(), No * sizeof(int));
    if (ua.n_sections == s_pcam + 1 && s_pcam == s_corig + 1 && s_corig == s_cpt + 1 && o_uv != (size_t)-1 && o_psrc != (size_t)-1) {
      ua.n_obs = No; ua.off_uv = o_uv; ua.off_psrc = o_psrc;
      ua.off_cpt = ua.off[s_cpt]; ua.off_corig = ua.off[s_corig]; ua.off_pcam = ua.off[s_pcam];
      ua.p_uv = h->p_uv.p; ua.c_uv = h->c_uv.p; ua.uv_f32 = h->uv_f32 ? 1 : 0;
      ua.c_ptf0 = h->c_ptf[0].p; ua.c_ptf1 = h->c_ptf[1].p; ua.p_camf0 = h->p_camf[0].p; ua.p_camf1 = h->p_camf[1].p;
      packed = true;
    } else {
      return fail(BA_ERR_STATE, "ba_set_problem: the staging arena was sized too small (internal)");
    }
  }
  if (ua.n_sections > 0) {     // the arena's one copy and the one kernel that deals it out
    HIPCHECK(h->up_dev.alloc(up_used));
    HIPCHECK(hipMemcpyAsync(h->up_dev.p, h->h_up, up_used, hipMemcpyHostToDevice, h->stream));
    ua.arena = h->up_dev.p;
    const int nthreads = std::max(No, 4096);
    BA_LAUNCH(k_unpack_problem, dim3((nthreads + 255) / 256), dim3(256), 0, h->stream, ua);
  }
  if (No > 0 && !packed) {
    HIPCHECK(upload(h->c_pt.p, c_pt.data(), No * sizeof(int)));
    HIPCHECK(upload(h->c_orig.p, c_orig.data(), No * sizeof(int)));
    HIPCHECK(upload(h->p_cam.p, p_cam.data(), No * sizeof(int)));
    // pixels: uploaded once in the caller's order, permuted into both orderings on the device
    // (staging: the residual buffer for the pixels, a flagged-index buffer for the point-order permutation)
    HIPCHECK(h->rbuf.alloc(2 * (size_t)No));
    HIPCHECK(upload(h->rbuf.p, uv, 2 * (size_t)No * sizeof(double)));
    HIPCHECK(upload(h->c_ptf[0].p, p_src.data(), No * sizeof(int)));
    const dim3 gg((No + 255) / 256), gb(256);
    BA_LAUNCH(k_gather_uv, gg, gb, 0, h->stream, (const double2*)h->rbuf.p, (const int*)h->c_ptf[0].p, No, h->p_uv.p, (int)h->uv_f32);
    BA_LAUNCH(k_gather_uv, gg, gb, 0, h->stream, (const double2*)h->rbuf.p, (const int*)h->c_orig.p, No, h->c_uv.p, (int)h->uv_f32);
    // the flagged copies of the index streams start as the plain streams: a robust linearisation reads them and stores
    // an entry only where its "weights are not (1, 1)" flag changes
    BA_LAUNCH(k_init_flagged, gg, gb, 0, h->stream, (const int*)h->c_pt.p, (const int*)h->p_cam.p, No, h->c_ptf[0].p, h->c_ptf[1].p,
              h->p_camf[0].p, h->p_camf[1].p);
  }
  if (packed || (staged && No == 0 && ua.n_sections > 0)) {
    // everything the device still reads sits in the pinned arena (the host vectors were copied into it): no need to wait
    // for the copy and the kernel -- whatever comes next is ordered behind them on the stream; the arena's next use
    // (the next ba_set_problem) waits for this event first
    if (!h->up_event) HIPCHECK(hipEventCreateWithFlags(&h->up_event, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(h->up_event, h->stream));
    h->up_pending = true;
    if (int rc = check_launches(h)) return rc;
  } else {
    BA_SYNC(h);   // host vectors go out of scope
  }
  stage("upload");
  h->have_problem = true;
  h->have_params = false;
  h->linearized = false;
  return BA_OK;
}

// block sizes of the running camera model
static int nb_of(const ba_handle* h) { return h->model ? BalCam::NB : Pinhole::NB; }
// partitions a consumer of the camera passes' partial sums adds up: all of them on one rank; in a multi-rank job the
// arrays arrive folded into partition 0 and all-reduced (fold_and_reduce), the other partitions are stale
static int nparts_of(const ba_handle* h) { return h->multi ? 1 : NPART; }
// the camera half's running sums a consumer reads: all partitions on one rank; the folded, all-reduced message otherwise
static const double* partL_of(const ba_handle* h, int buf) { return h->multi ? h->linmsg[buf].p + 8 : h->partL[buf].p; }
static int nbv(const ba_handle* h) { return h->nblkVm[h->model]; }      // camera-vector workgroups of the active model
static int nh_of(const ba_handle* h) { return h->model ? BalCam::NH : Pinhole::NH; }
static int nl_of(const ba_handle* h) { return h->model ? BalCam::NL : Pinhole::NL; }
static size_t lds_of(const ba_handle* h) { return h->lds_bytes_m[h->model]; }
static bool all_lds_of(const ba_handle* h) { return h->all_lds_m[h->model]; }
static double* bc_ptr(ba_handle* h) { return h->HccBc.p + nh_of(h) * (size_t)h->Nc; }
static int cam_grid(ba_handle* h, int segl = 64) { const int cpb = 64 * WPB / segl; return ((h->Nc + cpb - 1) / cpb) * NPART; }
static int row_grid(ba_handle* h) { return ((h->Nc + ROWS - 1) / ROWS) * NPART; }

// pinned bounce buffer for parameter transfers of at most 1 MB (grow-only); false: copy from / to the caller's memory
static bool par_bounce(ba_handle* h, size_t bytes) {
  if (bytes == 0 || bytes > ((size_t)1 << 20)) return false;
  if (h->par_pending) { (void)hipEventSynchronize(h->par_event); h->par_pending = false; }     // the buffer's last upload has landed
  if (h->h_par_cap < bytes) {
    if (h->h_par) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->h_par); h->h_par = nullptr; h->h_par_cap = 0; }
    const size_t cap = std::max<size_t>(2 * bytes, (size_t)64 << 10);
    if (hipHostMalloc((void**)&h->h_par, cap, hipHostMallocDefault) != hipSuccess) { h->h_par = nullptr; (void)hipGetLastError(); return false; }
    h->h_par_cap = cap;
  }
  return true;
}
extern "C" int ba_set_params(ba_handle* h, const double* cams, const double* pts) {
  if (!h || !cams || (!pts && h->Np > 0)) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_problem) return fail(BA_ERR_STATE, "ba_set_problem has not been called");
  if (set_device(h)) return BA_ERR_HIP;
  h->cur = 0;
  // a window's worth of parameters goes through a pinned bounce buffer: copies from pageable memory are staged by the
  // runtime one blocking transfer at a time (~10 us each), more than the kernels behind them take
  const size_t cb = 6 * (size_t)h->Nc * sizeof(double), pb = 3 * (size_t)h->Np * sizeof(double);
  const double *cams_src = cams, *pts_src = pts;
  if (par_bounce(h, cb + pb)) {
    memcpy(h->h_par, cams, cb);
    if (pb) memcpy(h->h_par + cb, pts, pb);
    cams_src = (const double*)h->h_par; pts_src = (const double*)(h->h_par + cb);
  }
  HIPCHECK(hipMemcpyAsync(h->cams[0].p, cams_src, cb, hipMemcpyHostToDevice, h->stream));
  if (h->Np > 0) {
    HIPCHECK(hipMemcpyAsync(h->stage.p, pts_src, pb, hipMemcpyHostToDevice, h->stream));
    BA_LAUNCH(k_pack_points, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->stage.p, h->slot.p, h->Np, h->ptab[0].p);
  }
  BA_LAUNCH(k_cam_prepare<Pinhole>, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->cams[0].p, (const double*)h->intr[0].p,
            h->cs[0].p, h->camA[0].p, h->Nc);
  if (cams_src != cams) {
    // the caller's arrays are no longer referenced: no need to wait for the two kernels (whatever comes next is ordered
    // behind them on the stream; a launch failure surfaces at the next synchronisation)
    if (!h->par_event) HIPCHECK(hipEventCreateWithFlags(&h->par_event, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(h->par_event, h->stream));
    h->par_pending = true;
    if (int rc = check_launches(h)) return rc;
  } else {
    BA_SYNC(h);
  }
  h->have_params = true;
  h->linearized = false;
  return BA_OK;
}

extern "C" int ba_get_params(ba_handle* h, double* cams, double* pts) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  const size_t cb = cams ? 6 * (size_t)h->Nc * sizeof(double) : 0, pb = (pts && h->Np > 0) ? 3 * (size_t)h->Np * sizeof(double) : 0;
  const bool bounce = par_bounce(h, cb + pb);
  if (cams) HIPCHECK(hipMemcpyAsync(bounce ? (void*)h->h_par : (void*)cams, h->cams[h->cur].p, cb, hipMemcpyDeviceToHost, h->stream));
  if (pts && h->Np > 0) {
    BA_LAUNCH(k_unpack_points, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->ptab[h->cur].p, h->slot.p, h->Np, h->stage.p);
    HIPCHECK(hipMemcpyAsync(bounce ? (void*)(h->h_par + cb) : (void*)pts, h->stage.p, pb, hipMemcpyDeviceToHost, h->stream));
  }
  BA_SYNC(h);
  if (bounce) {
    if (cb) memcpy(cams, h->h_par, cb);
    if (pb) memcpy(pts, h->h_par + cb, pb);
  }
  return BA_OK;
}

// multi-rank: every rank ends up with the points of all shards (its own at [p_begin, p_begin + Np)):
// zero-filled buffer + own slice, summed over the ranks with the solver's all-reduce
extern "C" int ba_allgather_points(ba_handle* h, int64_t p_begin, int64_t n_total, double* pts_all) {
  if (!h || !pts_all) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (p_begin < 0 || n_total < 0 || p_begin + (int64_t)h->Np > n_total)
    return fail(BA_ERR_INVALID, "shard [%lld, %lld) does not fit in %lld points", (long long)p_begin,
                (long long)(p_begin + h->Np), (long long)n_total);
  if (n_total == 0) return BA_OK;
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(h->gather.alloc(3 * (size_t)n_total));
  HIPCHECK(hipMemsetAsync(h->gather.p, 0, 3 * (size_t)n_total * sizeof(double), h->stream));
  if (h->Np > 0)
    BA_LAUNCH(k_unpack_points, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->ptab[h->cur].p, h->slot.p, h->Np,
                       h->gather.p + 3 * (size_t)p_begin);
  if (int rc = allreduce(h, h->gather.p, 3 * (size_t)n_total)) return rc;
  HIPCHECK(hipMemcpyAsync(pts_all, h->gather.p, 3 * (size_t)n_total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  return BA_OK;
}

extern "C" int ba_get_rotations(ba_handle* h, double* R) {
  if (!h || !R) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(hipMemcpy2DAsync(R, 9 * sizeof(double), h->cs[h->cur].p, CS * sizeof(double), 9 * sizeof(double), h->Nc,
                            hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  return BA_OK;
}

// ---------------------------------------------------------------------- launch helpers
// Every helper dispatches on the camera model of the running call (h->model): the same kernel templates, instantiated
// for the reference's pinhole and for the BAL camera (ba_models.hpp).
#define BA_BY_MODEL(CALL_T)                      \
  do {                                           \
    if (h->model) CALL_T(BalCam);                \
    else CALL_T(Pinhole);                        \
  } while (0)

static void launch_residual(ba_handle* h, int which, bool robust, double fscale, double* r_out) {
  Scope sc(h, BA_K_RESIDUAL);
  if (h->model) {
    auto kern = robust ? k_cam_residual_bal<true> : k_cam_residual_bal<false>;
    BA_LAUNCH(kern, dim3(cam_grid(h)), dim3(64 * WPB), 0, h->stream, h->cs[which].p, (const double*)h->intr[which].p, h->ptab[which].p,
              h->offk.p, h->c_pt.p, uv_arr(h, h->c_uv), h->c_orig.p, fscale, h->Nc, h->cam_band, r_out, h->partR.p);
    return;
  }
  auto kern = robust ? k_cam_residual<true> : k_cam_residual<false>;
  BA_LAUNCH(kern, dim3(cam_grid(h)), dim3(64 * WPB), 0, h->stream, h->cs[which].p, h->ptab[which].p, h->offk.p,
                     h->c_pt.p, uv_arr(h, h->c_uv), h->c_orig.p, h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, h->Nc, h->cam_band, r_out,
                     h->partR.p);
}
// fold the partial arrays of a step into `scal` (residual always; point / camera parts optional)
// with_step: also the step partials and the PCG verdict for iteration k; on a single rank the
// results go straight to the host-mapped mirror (no copy kernel)
// decide (single rank): the same kernel also computes the gain ratio and the next damping (lm_decide)
static ScalarsArgs scalars_args(ba_handle* h, bool with_step, int k, double tol2, int min_iters, long long seq, double cost_cur,
                                double lambda, double lam_floor = 0.0) {
  const bool direct = with_step && !h->multi;         // results straight into host-mapped memory + sequence word
  ScalarsArgs a;
  a.partR = h->partR.p; a.nR = NPART * h->Nc;
  a.partB = h->partB.p; a.nB = (with_step && h->Np > 0) ? h->nblkP + h->nblkL : 0;
  a.partC = h->partC.p; a.nC = with_step ? nbv(h) : 0;
  a.kit = k;
  a.st = with_step ? (const PcgState*)h->st.p : (const PcgState*)nullptr;
  a.partV = h->partV.p; a.nblkV = nbv(h);
  a.tol2 = tol2; a.min_iters = min_iters;
  a.scal = h->scal.p; a.scal_host = direct ? h->d_scal_host : (double*)nullptr;
  a.host_flag = direct ? h->d_flags + 2 : (long long*)nullptr; a.seq = seq;
  a.decide = direct ? 1 : 0; a.cost_cur = cost_cur; a.lambda = lambda; a.lam_floor = lam_floor;
  a.lam_slot = nullptr; a.err_flag = nullptr; a.on = 0;
  return a;
}
static void launch_scalars(ba_handle* h, bool with_step, int k = 0, double tol2 = 0.0, int min_iters = 0, long long seq = 0,
                           double cost_cur = 0.0, double lambda = 0.0, double lam_floor = 0.0) {
  Scope sc(h, BA_K_MISC);
  BA_LAUNCH(k_scalars, dim3(1), dim3(1024), 0, h->stream, scalars_args(h, with_step, k, tol2, min_iters, seq, cost_cur, lambda, lam_floor));
}
// spin on a host-mapped sequence word until it reaches `target` (the device publishes with a
// system-scope release); a wall-clock limit turns a wedged GPU into an error instead of a hang
static int wait_flag(ba_handle* h, int idx, long long target) {
  // a kernel the runtime refused to launch will never publish: say so instead of spinning into the time-out
  if (int rc = check_launches(h)) return rc;
  volatile long long* f = h->h_flags + idx;
  const auto t0 = std::chrono::steady_clock::now();
  double next_query = 2e-3;         // seconds of waiting before the stream is first asked for an error state
  unsigned spins = 0;
  while (*f < target) {
    if ((++spins & 0xfff) != 0) continue;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > next_query) {      // a faulted kernel puts the stream into a sticky error state: report that, not a time-out
      next_query = waited * 2;
      const hipError_t q = hipStreamQuery(h->stream);
      if (q != hipSuccess && q != hipErrorNotReady)
        return fail(BA_ERR_HIP, "the stream reports '%s' while waiting for flag %d", hipGetErrorString(q), idx);
      if (q == hipSuccess && *f < target)          // stream drained and the word was never written
        return fail(BA_ERR_HIP, "the stream drained without publishing flag %d (%lld < %lld)", idx, (long long)*f, target);
    }
    if (waited > 20.0)
      return fail(BA_ERR_HIP, "timed out waiting for the device (flag %d: %lld < %lld)", idx, (long long)*f, target);
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return BA_OK;
}
// camera half of the linearisation at parameter set `which`, into buffer set `buf`
// cost: also the cost partials at that parameter set (partR) -- the pass then doubles as the trial-cost evaluation
template <class CM>
static void launch_lin_cam_t(ba_handle* h, int which, int buf, bool robust, double fscale, bool cost) {
  auto kern = robust ? (cost ? k_camrow_linearize<CM, true, true> : k_camrow_linearize<CM, true, false>)
                     : (cost ? k_camrow_linearize<CM, false, true> : k_camrow_linearize<CM, false, false>);
  BA_LAUNCH(kern, dim3(row_grid(h)), dim3(ROW_LANES * ROWS), 0, h->stream, h->cs[which].p, (const double*)h->intr[which].p,
                     h->ptab[which].p, h->offk.p, h->c_pt.p, uv_arr(h, h->c_uv), h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, h->Nc,
                     h->cam_band, h->c_w[buf].p, h->c_ptf[buf].p, h->partL[buf].p, h->partR.p);
}
static void launch_lin_cam(ba_handle* h, int which, int buf, bool robust, double fscale, bool cost = false) {
  Scope sc(h, BA_K_LINEARIZE_CAM);
#define CALL_T(CM) launch_lin_cam_t<CM>(h, which, buf, robust, fscale, cost)
  BA_BY_MODEL(CALL_T);
#undef CALL_T
}
static void launch_lin_finalize(ba_handle* h) {
  Scope sc(h, BA_K_MISC);
#define CALL_T(CM) BA_LAUNCH(k_lin_finalize<CM::NB>, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, partL_of(h, h->lb), \
                             nparts_of(h), h->cs[h->cur].p, h->Nc, h->fixed, h->HccBc.p, bc_ptr(h))
  BA_BY_MODEL(CALL_T);
#undef CALL_T
}
static PtWork pt_work(ba_handle* h) {        // every point; long tracks skipped when they have a launch of their own
  return PtWork{nullptr, h->Np, h->nblkL ? h->long_thr : 0x7fffffff, 0, h->ppb, h->xcd_ranges};
}
static PtWork pt_work_long(ba_handle* h) { return PtWork{h->long_pts.p, h->n_long, 0x7fffffff, h->nblkP, h->long_spb, 0}; }
// point half at parameter set `w` into point-buffer set `pbuf`, with the damped inverse / y0 at `lambda` fused in
// (lam_dev != null: the damping is read from that device word instead -- a speculated pass, see ba_solve)
template <class CM>
static void launch_lin_pt_t(ba_handle* h, int w, int pbuf, bool robust, double fscale, double lambda, const double* lam_dev,
                            const ScalarsArgs& sa) {
  const int ride = sa.on * NPART;                      // the step's scalar fold + verdict as workgroup 0 of this launch (+ NPART - 1 idle ones)
#define LP_HEAD h->camA[w].p, h->ptab[w].p, h->pt_off.p, h->p_cam.p, uv_arr(h, h->p_uv), h->blk_win.p
#define LP_TAIL h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, lambda, lam_dev, h->Hpp[pbuf].p, h->bp[pbuf].p, h->p_w[pbuf].p,      \
                h->p_camf[pbuf].p, h->Hppinv[pbuf].p, h->y0[pbuf].p, h->partG[pbuf].p, sa
#define LP_LAUNCH(R, L, LN, G, WK) BA_LAUNCH((k_pt_linearize<CM, R, L, LN>), dim3((G) + ride), dim3(PT_THREADS), lds_of(h), h->stream, LP_HEAD, WK, LP_TAIL)
#define LP_BOTH(R, L) BA_LAUNCH((k_pt_linearize_both<CM, R, L>), dim3(h->nblkP + h->nblkL + ride), dim3(PT_THREADS), lds_of(h), h->stream, \
                                         LP_HEAD, wk, h->nblkP, wl, LP_TAIL)
  const PtWork wk = pt_work(h), wl = pt_work_long(h);
  if (h->nblkL) {          // short and long tracks in one launch
    if (all_lds_of(h)) { if (robust) LP_BOTH(true, true); else LP_BOTH(false, true); }
    else               { if (robust) LP_BOTH(true, false); else LP_BOTH(false, false); }
  } else {
#define LP_ONE(R, L)                                             \
  do {                                                           \
    switch (h->lanes) {                                          \
      case 4: LP_LAUNCH(R, L, 4, h->nblkP, wk); break;           \
      case 8: LP_LAUNCH(R, L, 8, h->nblkP, wk); break;           \
      case 16: LP_LAUNCH(R, L, 16, h->nblkP, wk); break;         \
      default: LP_LAUNCH(R, L, 2, h->nblkP, wk); break;          \
    }                                                            \
  } while (0)
    if (all_lds_of(h)) { if (robust) LP_ONE(true, true); else LP_ONE(false, true); }
    else               { if (robust) LP_ONE(true, false); else LP_ONE(false, false); }
#undef LP_ONE
  }
#undef LP_BOTH
#undef LP_LAUNCH
#undef LP_HEAD
#undef LP_TAIL
}
static void launch_lin_pt(ba_handle* h, int w, int pbuf, bool robust, double fscale, double lambda, const double* lam_dev = nullptr,
                          const ScalarsArgs* rider = nullptr) {
  if (h->Np == 0) return;
  Scope sc(h, BA_K_LINEARIZE_PT);
  ScalarsArgs sa;
  if (rider) sa = *rider;
  else { memset(&sa, 0, sizeof sa); }
#define CALL_T(CM) launch_lin_pt_t<CM>(h, w, pbuf, robust, fscale, lambda, lam_dev, sa)
  BA_BY_MODEL(CALL_T);
#undef CALL_T
}
static void launch_point_invert(ba_handle* h, double lambda) {
  if (h->Np == 0) return;
  Scope sc(h, BA_K_POINT_INVERT);
  BA_LAUNCH(k_point_invert, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->Hpp[h->pb].p, h->bp[h->pb].p, lambda,
                     h->Np, h->Hppinv[h->pb].p, h->y0[h->pb].p, h->ptab[h->cur].p);
}
// part6 buffer: [u.y word, pad | NPART x Nc x NB partial sums]; the u.y word sits in FRONT of partition 0 so that a
// multi-rank job all-reduces it together with the folded partition (one contiguous message)
constexpr int GMAX_HOST_SLOT = 40;   // word of the host-mapped scalar block that receives max |gradient| (k_scalars uses [0, S_COUNT))
static double* uy_ptr(ba_handle* h) { return h->part6.p; }
static double* p6_ptr(ba_handle* h) { return h->part6.p + 2; }
// (multi-rank: the damped system's sums are read from the all-reduced message of exchange_system)
static double* sys_p6(ba_handle* h) { return h->multi ? h->sysmsg.p + 2 : p6_ptr(h); }
static double* sys_E(ba_handle* h) { return h->multi ? h->sysmsg.p + 2 + nb_of(h) * (size_t)h->Nc : h->partE.p; }
static size_t sys_nE(ba_handle* h) { return h->sys_diag ? nh_of(h) * (size_t)h->Nc : 0; }
static const double* gmax_parts(ba_handle* h) {   // per-workgroup (single rank) / per-rank (multi-rank) maxima of |bp|
  return h->multi ? h->sysmsg.p + 2 + nb_of(h) * (size_t)h->Nc + sys_nE(h) : h->partG[h->pb].p;
}
static int gmax_count(ba_handle* h) { return h->multi ? h->world : (h->Np > 0 ? h->nblkP + h->nblkL : 0); }
// camera pass of the Schur product on the y slot of the current point table
//   diag: also the Schur-Jacobi blocks; pcg: iteration k with early exit
template <class CM>
static void launch_cam_schur_t(ba_handle* h, bool robust, bool diag, bool pcg, int k) {
  const int w = h->cur;
#define CS_ARGS h->cs[w].p, (const double*)h->intr[w].p, h->ptab[w].p, h->offk.p, (robust ? h->c_ptf[h->lb].p : h->c_pt.p), h->c_w[h->lb].p,  \
                h->K4[0], h->K4[1], h->Nc, h->cam_band, h->fixed,                                                                 \
                p6_ptr(h), k, (const double*)h->verdict.p, h->partA.p, (h->Np > 0 ? h->nblkP + h->nblkL : 0), uy_ptr(h)
  const int segl = pcg ? h->cam_segl : 64;
  const dim3 g(cam_grid(h, segl) + (pcg ? 1 : 0)), b(64 * WPB);
#define CS_PCG(R, JT)                                                                                       \
  do {                                                                                                      \
    if (segl == 16) BA_LAUNCH((k_cam_schur<CM, R, true, JT, 16>), g, b, 0, h->stream, CS_ARGS);        \
    else if (segl == 32) BA_LAUNCH((k_cam_schur<CM, R, true, JT, 32>), g, b, 0, h->stream, CS_ARGS);   \
    else BA_LAUNCH((k_cam_schur<CM, R, true, JT, 64>), g, b, 0, h->stream, CS_ARGS);                   \
  } while (0)
  if (diag) {
    auto kern = robust ? k_camrow_schur_diag<CM, true> : k_camrow_schur_diag<CM, false>;
    BA_LAUNCH(kern, dim3(row_grid(h)), dim3(ROW_LANES * ROWS), 0, h->stream, h->cs[w].p, (const double*)h->intr[w].p, h->ptab[w].p,
                       h->offk.p, (robust ? h->c_ptf[h->lb].p : h->c_pt.p), h->c_w[h->lb].p, h->Hppinv[h->pb].p, h->K4[0],
                       h->K4[1], h->Nc, h->cam_band, h->fixed, p6_ptr(h), h->partE.p);
  } else if (pcg) {
    if (h->jac_f32) { if (robust) CS_PCG(true, float); else CS_PCG(false, float); }
    else            { if (robust) CS_PCG(true, double); else CS_PCG(false, double); }
  } else {
    if (robust) BA_LAUNCH((k_cam_schur<CM, true, false, double, 64>), g, b, 0, h->stream, CS_ARGS);
    else        BA_LAUNCH((k_cam_schur<CM, false, false, double, 64>), g, b, 0, h->stream, CS_ARGS);
  }
#undef CS_PCG
#undef CS_ARGS
}
static void launch_cam_schur(ba_handle* h, bool robust, bool diag, bool pcg, int k, double tol2, int min_iters) {
  Scope sc(h, diag ? BA_K_PRECOND : BA_K_SCHUR_CAM);
  (void)tol2; (void)min_iters;
#define CALL_T(CM) launch_cam_schur_t<CM>(h, robust, diag, pcg, k)
  BA_BY_MODEL(CALL_T);
#undef CALL_T
}
// point pass with the camera vector in vtil; mode 0 = PCG iteration k, mode 1 = back substitution
// gmax_out (first PCG probe behind a fresh linearisation): host-mapped word that receives max |gradient|
template <class CM>
static void launch_pt_schur_t(ba_handle* h, bool robust, int mode, int k, double tol2, int min_iters, long long flag_base,
                              double* gmax_out, const CamUpdateArgs& cu) {
  const int w = h->cur;
  const int ride = (mode == 1 || cu.fuse) ? cu.n_blocks : 0;      // the camera update as extra workgroups of the back substitution
                                                                  // (or of the PCG point pass that may turn into it: cu.fuse)
#define PS_HEAD h->camA[w].p, h->ptab[w].p, h->pt_off.p, (robust ? h->p_camf[h->pb].p : h->p_cam.p), h->p_w[h->pb].p,                 \
                h->Hppinv[h->pb].p, h->blk_win.p
#define PS_TAIL h->K4[0], h->K4[1], h->fixed, h->partA.p, k, h->st.p, h->partV.p, nbv(h), tol2, min_iters, h->y0[h->pb].p,     \
                h->Hpp[h->pb].p, h->bp[h->pb].p, h->ptab[1 - w].p, h->partB.p, flag, flag_base, h->verdict.p,                      \
                gmax_parts(h), gmax_count(h), (const double*)h->partGc.p, nbv(h), gmax_out, cu
  const size_t lds = std::max(lds_of(h), ride ? cu.groups * cam_update_lds_doubles<CM>() * sizeof(double) : (size_t)0) + (size_t)h->debug_lds_extra;
  long long* flag = (mode == 0 && flag_base > 0) ? h->d_flags : (long long*)nullptr;
  // data with long tracks: short and long tracks in one launch (workgroup 0 publishes the verdict)
#define PS_ONE(R, M, L, LN, JT) \
  BA_LAUNCH((k_pt_schur<CM, R, M, L, LN, JT>), dim3(h->nblkP + ride), dim3(PT_THREADS), lds, h->stream, PS_HEAD, wk, PS_TAIL)
#define PS_LAUNCH(R, M, L, JT)                                                                                              \
  do {                                                                                                                      \
    if (h->nblkL) BA_LAUNCH((k_pt_schur_both<CM, R, M, L, JT>), dim3(h->nblkP + h->nblkL + ride), dim3(PT_THREADS), lds,  \
                                     h->stream, PS_HEAD, wk, h->nblkP, wl, PS_TAIL);                                        \
    else {                                                                                                                  \
      switch (h->lanes) {                                                                                                   \
        case 4: PS_ONE(R, M, L, 4, JT); break;                                                                              \
        case 8: PS_ONE(R, M, L, 8, JT); break;                                                                              \
        case 16: PS_ONE(R, M, L, 16, JT); break;                                                                            \
        default: PS_ONE(R, M, L, 2, JT); break;                                                                             \
      }                                                                                                                     \
    }                                                                                                                       \
  } while (0)
#define PS_MODE(R, L)                                                                 \
  do {                                                                                \
    if (mode != 0) PS_LAUNCH(R, 1, L, double);                                        \
    else if (f32) PS_LAUNCH(R, 0, L, float);                                          \
    else PS_LAUNCH(R, 0, L, double);                                                  \
  } while (0)
  const PtWork wk = pt_work(h), wl = pt_work_long(h);
  const bool f32 = h->jac_f32 && flag_base > 0;       // only inside the PCG loop of ba_solve
  if (all_lds_of(h)) { if (robust) PS_MODE(true, true); else PS_MODE(false, true); }
  else               { if (robust) PS_MODE(true, false); else PS_MODE(false, false); }
#undef PS_MODE
#undef PS_LAUNCH
#undef PS_ONE
#undef PS_HEAD
#undef PS_TAIL
}
static void launch_pt_schur(ba_handle* h, bool robust, int mode, int k, double tol2, int min_iters, long long flag_base = 0,
                            double* gmax_out = nullptr, const CamUpdateArgs* rider = nullptr) {
  if (h->Np == 0) {
    // an empty landmark shard (multi-rank): no point pass, but the PCG probe's verdict is still owed
    if (mode == 0 && flag_base > 0) {
      Scope sc(h, BA_K_SCHUR_PT);
      BA_LAUNCH(k_pcg_probe, dim3(1), dim3(64), 0, h->stream, k, h->st.p, h->partV.p, nbv(h), tol2, min_iters, h->d_flags,
                flag_base, h->verdict.p, gmax_parts(h), gmax_count(h), (const double*)h->partGc.p, nbv(h), gmax_out);
    }
    return;
  }
  Scope sc(h, mode == 0 ? BA_K_SCHUR_PT : BA_K_BACKSUB);
  CamUpdateArgs cu;
  if (rider) cu = *rider;
  else memset(&cu, 0, sizeof cu);
#define CALL_T(CM) launch_pt_schur_t<CM>(h, robust, mode, k, tol2, min_iters, flag_base, gmax_out, cu)
  BA_BY_MODEL(CALL_T);
#undef CALL_T
}

// multi-rank: a buffer of NPART per-partition partial sums is folded in place (partition 0 <- the sum
// in partition order, the others <- 0) and only partition 0 is all-reduced; the consumers, which add
// the NPART partitions, then read "total + 0 + ... + 0" and stay the kernels they are on one rank.
// An eighth of the bytes on the wire for two launch-floor kernels.  Single rank: nothing.
static int fold_and_reduce(ba_handle* h, double* parts, size_t n_per_part, double* msg, size_t msg_count) {
  if (!h->multi) return BA_OK;
  if (!h->one_part) {        // (one_part: partitions 1 .. NPART-1 are empty, partition 0 already is the sum)
    Scope sc(h, BA_K_MISC);
    BA_LAUNCH(k_fold_parts, dim3((unsigned)((n_per_part + 255) / 256)), dim3(256), 0, h->stream, parts, n_per_part, NPART);
  }
  return allreduce(h, msg, msg_count);
}
// with_scalars: the step's six local sums (h->scal, written by k_scalars just before) ride in the message's header
static int exchange_partL(ba_handle* h, int buf, bool with_scalars = false) {
  if (!h->multi) return BA_OK;
  const size_t n = nl_of(h) * (size_t)h->Nc;
  HIPCHECK(h->linmsg[buf].alloc(8 + n));           // (sized in ba_set_problem; a communicator that joined after it: here)
  {
    Scope sc(h, BA_K_MISC);
    BA_LAUNCH(k_fold_lin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, with_scalars ? (const double*)h->scal.p : (const double*)nullptr,
              (const double*)h->partL[buf].p, n, h->one_part ? 1 : NPART, h->linmsg[buf].p);
  }
  return allreduce(h, h->linmsg[buf].p, 8 + n);
}
static int exchange_schur(ba_handle* h) {
  if (!h->multi) return BA_OK;
  // the message starts at the u.y word in front of partition 0
  return fold_and_reduce(h, p6_ptr(h), nb_of(h) * (size_t)h->Nc, uy_ptr(h), 2 + nb_of(h) * (size_t)h->Nc);
}
// The damped system's sums of a multi-rank job: W y0 (part6), the Schur-Jacobi blocks (partE, with_diag) and every rank's
// max |bp| travel in ONE message (k_fold_msg gathers and folds them, one all-reduce); k_pcg_setup and the first PCG probe
// read them from there.
static int exchange_system(ba_handle* h, bool with_diag) {
  if (!h->multi) return BA_OK;
  h->sys_diag = with_diag;
  const size_t n6 = nb_of(h) * (size_t)h->Nc, nE = sys_nE(h);
  HIPCHECK(h->sysmsg.alloc(2 + n6 + nE + (size_t)h->world + 8));
  {
    Scope sc(h, BA_K_MISC);
    BA_LAUNCH(k_fold_msg, dim3((unsigned)((n6 + nE + 255) / 256)), dim3(256), 0, h->stream, (const double*)uy_ptr(h),
              (const double*)p6_ptr(h), n6, (const double*)h->partE.p, nE, h->one_part ? 1 : NPART,
              (const double*)h->partG[h->pb].p, h->Np > 0 ? h->nblkP + h->nblkL : 0, h->rank, h->world, h->sysmsg.p);
  }
  return allreduce(h, h->sysmsg.p, 2 + n6 + nE + (size_t)h->world);
}
// finalize = true: fold the fresh camera-half partials into Hcc | bc inside the same kernel
// precond: 0 = Jacobi blocks, 1 = Schur-Jacobi blocks from partE, 2 = keep the blocks already in Minv (k_pcg_setup)
static void launch_pcg_setup(ba_handle* h, double lambda, int precond, bool finalize) {
  Scope sc(h, BA_K_PCG_UPDATE);
#define SU_ARGS partL_of(h, h->lb), h->HccBc.p, bc_ptr(h), sys_p6(h), sys_E(h), nparts_of(h), h->cs[h->cur].p, lambda,           \
                precond, h->Nc, h->fixed, h->Hccd.p, h->Minv.p, h->gvec.p, h->x.p, h->r.p, h->p.p, h->s.p,     \
                h->z.p, h->camA[h->cur].p, h->partV.p, h->st.p, h->partGc.p, (h->two_level ? h->coarse_rc.p : (double*)nullptr), h->vx.p
#define CALL_T(CM)                                                                                               \
  do {                                                                                                           \
    if (finalize) BA_LAUNCH((k_pcg_setup<CM, true>), dim3(nbv(h)), dim3(VEC_BLOCK), 0, h->stream, SU_ARGS);    \
    else          BA_LAUNCH((k_pcg_setup<CM, false>), dim3(nbv(h)), dim3(VEC_BLOCK), 0, h->stream, SU_ARGS);   \
  } while (0)
  BA_BY_MODEL(CALL_T);
#undef CALL_T
#undef SU_ARGS
}

// --------------------------------------------------------------------- K1 entry point
extern "C" int ba_residuals(ba_handle* h, int32_t loss, double f_scale, double* r, double* sse, double* cost) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (loss != BA_LOSS_LINEAR && loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss %d", loss);
  if (!(f_scale > 0)) return fail(BA_ERR_INVALID, "f_scale must be positive");
  if (set_device(h)) return BA_ERR_HIP;
  double* rdev = nullptr;
  if (r && h->Nobs > 0) {
    HIPCHECK(h->rbuf.alloc(2 * (size_t)h->Nobs));
    rdev = h->rbuf.p;
  }
  launch_residual(h, h->cur, loss == BA_LOSS_HUBER, f_scale, rdev);
  launch_scalars(h, false);
  if (int rc = allreduce(h, h->scal.p, 2)) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (rdev) HIPCHECK(hipMemcpyAsync(r, rdev, 2 * (size_t)h->Nobs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  if (sse) *sse = h->h_scal[0];
  if (cost) *cost = 0.5 * h->h_scal[1];
  return BA_OK;
}

// K1 for the BAL 9-parameter camera (row f2): same problem upload and row order as ba_residuals, the cameras'
// (f, k1, k2) handed over per call; the handle's K4 is not used.
extern "C" int ba_residuals_bal(ba_handle* h, const double* intr, int32_t loss, double f_scale, double* r, double* sse,
                                double* cost) {
  if (!h || !intr) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (loss != BA_LOSS_LINEAR && loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss %d", loss);
  if (!(f_scale > 0)) return fail(BA_ERR_INVALID, "f_scale must be positive");
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(h->tri.alloc(3 * (size_t)h->Nc + 8));
  HIPCHECK(hipMemcpyAsync(h->tri.p, intr, 3 * (size_t)h->Nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  double* rdev = nullptr;
  if (r && h->Nobs > 0) {
    HIPCHECK(h->rbuf.alloc(2 * (size_t)h->Nobs));
    rdev = h->rbuf.p;
  }
  {
    Scope sc(h, BA_K_RESIDUAL);
    auto kern = loss == BA_LOSS_HUBER ? k_cam_residual_bal<true> : k_cam_residual_bal<false>;
    BA_LAUNCH(kern, dim3(cam_grid(h)), dim3(64 * WPB), 0, h->stream, h->cs[h->cur].p, (const double*)h->tri.p, h->ptab[h->cur].p,
              h->offk.p, h->c_pt.p, uv_arr(h, h->c_uv), h->c_orig.p, f_scale, h->Nc, h->cam_band, rdev, h->partR.p);
  }
  launch_scalars(h, false);
  if (int rc = allreduce(h, h->scal.p, 2)) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (rdev) HIPCHECK(hipMemcpyAsync(r, rdev, 2 * (size_t)h->Nobs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  if (sse) *sse = h->h_scal[0];
  if (cost) *cost = 0.5 * h->h_scal[1];
  return BA_OK;
}

// --------------------------------------------------------------------- K2 entry point
extern "C" int ba_linearize(ba_handle* h, int32_t loss, double f_scale, double* Hcc, double* bc, double* Hpp, double* bp) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (loss != BA_LOSS_LINEAR && loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss %d", loss);
  if (!(f_scale > 0)) return fail(BA_ERR_INVALID, "f_scale must be positive");
  if (set_device(h)) return BA_ERR_HIP;
  launch_lin_cam(h, h->cur, h->lb, loss == BA_LOSS_HUBER, f_scale);
  if (int rc = exchange_partL(h, h->lb)) return rc;
  launch_lin_finalize(h);
  launch_lin_pt(h, h->cur, h->pb, loss == BA_LOSS_HUBER, f_scale, 1.0);
  h->linearized = true;
  h->lin_robust = (loss == BA_LOSS_HUBER);
  h->lin_fscale = f_scale;
  if (Hcc) HIPCHECK(hipMemcpyAsync(Hcc, h->HccBc.p, 21 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (bc) HIPCHECK(hipMemcpyAsync(bc, bc_ptr(h), 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if ((Hpp || bp) && h->Np) {          // per-point blocks back in the caller's point order
    HIPCHECK(h->rbuf.alloc(6 * (size_t)h->Np));
    if (Hpp) {
      BA_LAUNCH(k_unpermute_rows, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->Hpp[h->pb].p, h->slot.p, h->Np, 6, h->rbuf.p);
      HIPCHECK(hipMemcpyAsync(Hpp, h->rbuf.p, 6 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (bp) {
      BA_LAUNCH(k_unpermute_rows, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->bp[h->pb].p, h->slot.p, h->Np, 3, h->stage.p);
      HIPCHECK(hipMemcpyAsync(bp, h->stage.p, 3 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
  }
  BA_SYNC(h);
  return BA_OK;
}

// Two-level preconditioner: E = P^T S P at the current damping, its banded factor and explicit inverse, then the first
// PCG vectors redone with the coarse term (k_pcg_coarse with k = -1 writes the partials iteration 0's probe sums).
static void coarse_build(ba_handle* h) {
  const int n = 6 * h->n_agg;
  const bool robust = h->lin_robust;
  Scope sc(h, BA_K_PRECOND);
  (void)hipMemsetAsync(h->coarseEint.p, 0, (size_t)n * n * sizeof(long long), h->stream);
  BA_LAUNCH(k_coarse_diag, dim3(1), dim3(1024), 0, h->stream, (const double*)h->Hccd.p, h->Nc, h->fixed, h->n_agg, h->coarseEint.p,
            h->coarse_info.p);
  if (robust)
    BA_LAUNCH(k_coarse_runs<true>, dim3((h->n_runs + 255) / 256), dim3(256), 0, h->stream, (const double*)h->cs[h->cur].p,
              (const double*)h->ptab[h->cur].p, (const int*)h->run_beg.p, (const int*)h->run_pt.p, (const int*)h->p_camf[h->pb].p,
              (const double2*)h->p_w[h->pb].p, h->K4[0], h->K4[1], h->fixed, h->n_runs, h->coarseU.p);
  else
    BA_LAUNCH(k_coarse_runs<false>, dim3((h->n_runs + 255) / 256), dim3(256), 0, h->stream, (const double*)h->cs[h->cur].p,
              (const double*)h->ptab[h->cur].p, (const int*)h->run_beg.p, (const int*)h->run_pt.p, (const int*)h->p_cam.p,
              (const double2*)h->p_w[h->pb].p, h->K4[0], h->K4[1], h->fixed, h->n_runs, h->coarseU.p);
  BA_LAUNCH(k_coarse_pairs, dim3((h->n_pairs + 255) / 256), dim3(256), 0, h->stream, (const int2*)h->run_pairs.p, h->n_pairs,
            (const int*)h->run_pt.p, (const int*)h->run_agg.p, (const double*)h->coarseU.p, (const double*)h->Hppinv[h->pb].p,
            h->n_agg, (const double*)h->coarse_info.p, (unsigned long long*)h->coarseEint.p);
  BA_LAUNCH(k_coarse_to_double, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, h->stream,
            (const long long*)h->coarseEint.p, n, (const double*)h->coarse_info.p, h->coarseE.p);
  BA_LAUNCH(k_coarse_cholesky, dim3(1), dim3(1024), 0, h->stream, h->coarseE.p, n, h->coarse_bw);
  BA_LAUNCH(k_coarse_inverse, dim3((n + 63) / 64), dim3(64), 0, h->stream, (const double*)h->coarseE.p, n, h->coarse_bw,
            h->coarseEinv.p);
  BA_LAUNCH(k_pcg_coarse, dim3(h->n_agg), dim3(VEC_BLOCK), 0, h->stream, -1, (const double*)h->coarseEinv.p,
            (const double*)h->coarse_rc.p, h->n_agg, (const double*)h->Hccd.p, (const double*)h->cs[h->cur].p, h->Nc, h->fixed,
            (const double*)h->r.p, h->z.p, h->camA[h->cur].p, h->partV.p, nbv(h), (const double*)h->verdict.p, 0);
}

// --------------------------------------------------------------------- K4 test hooks
// invert: (re)compute the damped point inverses (not needed right after launch_lin_pt at the
// same lambda); finalize: Hcc | bc still have to be folded from the camera-half partials
static void coarse_build(ba_handle* h);
// keep = true (Schur-Jacobi only, ba_options.precond_lag): the preconditioner blocks in Minv stay as they are -- the
// right-hand side W y0 then comes from the 6-sum camera pass (k_cam_schur) instead of the 27-sum one that also builds
// the blocks' Schur terms, and k_pcg_setup neither folds them nor inverts anything
static int damped_system(ba_handle* h, double lambda, bool schur_diag, bool invert = true, bool finalize = false, bool keep = false) {
  if (invert) launch_point_invert(h, lambda);
  const bool diag_pass = schur_diag && !keep;
  launch_cam_schur(h, h->lin_robust, diag_pass, false, 0, 0.0, 0);
  if (int rc = exchange_system(h, diag_pass)) return rc;
  launch_pcg_setup(h, lambda, schur_diag ? (keep ? 2 : 1) : 0, finalize);
  if (h->two_level) coarse_build(h);
  if (schur_diag) h->stats[keep ? BA_STAT_PRECOND_REUSES : BA_STAT_PRECOND_BUILDS]++;
  return BA_OK;
}

extern "C" int ba_schur_rhs(ba_handle* h, double lambda, double* g) {
  if (!h || !g) return fail(BA_ERR_INVALID, "null argument");
  if (!h->linearized) return fail(BA_ERR_STATE, "ba_linearize first");
  if (set_device(h)) return BA_ERR_HIP;
  if (int rc = damped_system(h, lambda, true)) return rc;
  HIPCHECK(hipMemcpyAsync(g, h->gvec.p, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  return BA_OK;
}

extern "C" int ba_schur_apply(ba_handle* h, double lambda, const double* v, double* out) {
  if (!h || !v || !out) return fail(BA_ERR_INVALID, "null argument");
  if (!h->linearized) return fail(BA_ERR_STATE, "ba_linearize first");
  if (set_device(h)) return BA_ERR_HIP;
  if (int rc = damped_system(h, lambda, false)) return rc;       // Hccd, Hppinv
  HIPCHECK(hipMemcpyAsync(h->vin.p, v, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    Scope sc(h, BA_K_MISC);
    BA_LAUNCH(k_vtil, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->vin.p, h->cs[h->cur].p, h->Nc,
                       h->fixed, h->camA[h->cur].p);
    BA_LAUNCH(k_pcg_reset, dim3(1), dim3(64), 0, h->stream, h->st.p, h->partV.p, nbv(h));
  }
  launch_pt_schur(h, h->lin_robust, 0, 0, -1.0, 1 << 30);        // y = Hppinv W^T v into the point table
  launch_cam_schur(h, h->lin_robust, false, false, 0, 0.0, 0);
  if (int rc = exchange_schur(h)) return rc;
  {
    Scope sc(h, BA_K_MISC);
    BA_LAUNCH(k_schur_combine, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->Hccd.p, h->vin.p, p6_ptr(h),
                       nparts_of(h), h->cs[h->cur].p, h->Nc, h->fixed, h->z.p);
  }
  HIPCHECK(hipMemcpyAsync(out, h->z.p, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  return BA_OK;
}

// ----------------------------------------------------------------------------- solve
extern "C" int ba_default_options(ba_options* o) {
  if (!o) return fail(BA_ERR_INVALID, "null options");
  memset(o, 0, sizeof *o);
  o->loss = BA_LOSS_HUBER;          // src/bundle_adjuster.py:171
  o->max_iters = 50;                // max_nfev=50, :173
  o->f_scale = 1.0;
  o->ftol = 1e-5;                   // :173
  o->xtol = 1e-5;                   // :173
  o->gtol = 1e-8;                   // scipy default (least_squares.py:241-245)
  o->initial_lambda = 1e-4;
  o->pcg_tol = 0.1;
  o->pcg_max_iters = 200;
  o->pcg_min_iters = 1;
  o->preconditioner = BA_PRECOND_SCHUR_JACOBI;
  o->jacobian_precision = 0;
  o->reserved0 = 0;
  o->profile = 0;
  o->verbose = 0;
  o->pcg_model_tol = -1.0;          // Nash & Sofer's truncated-Newton test on the quadratic model: automatic -- 0.5 (their value) on
                                    // band-structured problems, off otherwise (ba_hip.h)
  o->pcg_model_min_iters = 5;
  o->precond_lag = 3;
  return BA_OK;
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int eval_cost(ba_handle* h, int which, bool robust, double fscale, double* sse, double* cost) {
  launch_residual(h, which, robust, fscale, nullptr);
  launch_scalars(h, false);
  if (int rc = allreduce(h, h->scal.p, 2)) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  *sse = h->h_scal[0];
  *cost = 0.5 * h->h_scal[1];
  return BA_OK;
}

static int solve_impl(ba_handle* h, const ba_options* opts, ba_summary* sum);
extern "C" int ba_solve(ba_handle* h, const ba_options* opts, ba_summary* sum) {
  if (!h || !opts || !sum) return fail(BA_ERR_INVALID, "null argument");
  const int rc = solve_impl(h, opts, sum);
  if (rc != BA_OK) {             // leave the handle usable: nothing queued, no per-solve mode left on
    const std::string msg = g_err;
    (void)hipStreamSynchronize(h->stream);
    h->launch_err = hipSuccess;
    if (h->profile) flush_profile(h);
    h->profile = false;
    h->jac_f32 = false;
    h->two_level = false;
    h->linearized = false;
    g_err = msg;
  }
  return rc;
}
// Sliding-window-sized problems (ba_small.hpp): the whole LM loop in one kernel launch, dense Cholesky of the reduced
// system instead of PCG.  Same options, summary and trace as the multi-kernel path.
static bool small_applies(const ba_handle* h, const ba_options* opts) {
  static const long max_obs = [] { const char* e = getenv("BA_SMALL_MAX_OBS"); return e ? atol(e) : (long)SMALL_DEFAULT_MAX_OBS; }();
  // (the observation limit is the measured crossover of the ONE-workgroup kernel with the multi-kernel path; a window that
  // fits the multi-workgroup kernel -- five cameras, 2048 landmarks: at most 10 k observations -- is far below its own)
  const char* mw_env = getenv("BA_SMALL_MW");
  const bool mw = h->mw_ok && (!mw_env || atoi(mw_env) != 0);
  return opts->small_solver == 0 && !h->multi && h->Nc <= SMALL_MAX_CAMS && h->Np > 0 && h->Nobs > 0 && (h->Nobs <= max_obs || mw) &&
         opts->max_iters >= 1 && getenv("BA_NO_SMALL_SOLVER") == nullptr;
}
static int small_solve(ba_handle* h, const ba_options* opts, ba_summary* sum) {
  const double t_begin = now_s();
  const size_t off_cur = sizeof(ba_summary), off_trace = 256;
  const size_t bytes = off_trace + sizeof(ba_iter_record) * (size_t)opts->max_iters;
  if (h->h_small_bytes < bytes) {
    if (h->h_small) { BA_SYNC(h); (void)hipHostFree(h->h_small); h->h_small = nullptr; h->h_small_bytes = 0; }
    HIPCHECK(hipHostMalloc((void**)&h->h_small, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHECK(hipHostGetDevicePointer((void**)&h->d_small_host, h->h_small, 0));
    h->h_small_bytes = bytes;
  }
  SmallArgs A;
  for (int k = 0; k < 2; ++k) { A.cams[k] = h->cams[k].p; A.cs[k] = h->cs[k].p; A.ptab[k] = h->ptab[k].p; A.camA[k] = h->camA[k].p; }
  A.offk = h->offk.p; A.c_pt = h->c_pt.p; A.c_uv = h->c_uv.p;
  A.pt_off = h->pt_off.p; A.p_cam = h->p_cam.p; A.p_uv = h->p_uv.p;
  A.Hpp = h->Hpp[h->pb].p; A.bp = h->bp[h->pb].p; A.Lf = h->Hppinv[h->pb].p; A.y0 = h->y0[h->pb].p;
  A.Np_pad = (h->Np + 15) & ~15;
  A.Kp = 3 * A.Np_pad;
  const char* mw_env = getenv("BA_SMALL_MW");           // BA_SMALL_MW=0: always the one-workgroup kernel
  bool used_mw = h->mw_ok && (!mw_env || atoi(mw_env) != 0);
  HIPCHECK(h->small_gS.alloc((size_t)SMALL_WAVES * SMALL_TILES * 256 + 16));        // + 16 words of diagnostic stamps
  A.gS = h->small_gS.p;
  A.n_cams = h->Nc; A.n_pts = h->Np; A.fixed_cam = h->fixed; A.robust = opts->loss == BA_LOSS_HUBER;
  A.fx = h->K4[0]; A.fy = h->K4[1]; A.cx = h->K4[2]; A.cy = h->K4[3]; A.hub_c = opts->f_scale;
  A.max_iters = opts->max_iters; A.ftol = opts->ftol; A.xtol = opts->xtol; A.gtol = opts->gtol; A.lambda0 = opts->initial_lambda;
  A.cur = h->cur;
  A.summary = (ba_summary*)h->d_small_host;
  A.cur_out = (int*)(h->d_small_host + off_cur);
  A.trace = (ba_iter_record*)(h->d_small_host + off_trace);
  A.host_flag = h->d_flags + 4;
  h->profile = opts->profile != 0;
  A.stamps = getenv("BA_SMALL_STAMPS") ? (long long*)(h->small_gS.p + (size_t)SMALL_WAVES * SMALL_TILES * 256) : nullptr;   // device memory: a host store would stall the wave
  // one launch of either kernel and the wait for its result word: the kernel's last act is a system-scope release of the
  // sequence word -- summary, parameter set and trace are in host memory by then, and whatever the caller queues next on
  // the stream is ordered behind the kernel as usual
  auto run_once = [&](bool mw) -> int {
    if (!mw && h->small_np_pad != A.Np_pad) {             // columns of padding points and rows past 6 Nc stay zero for good
      const size_t nv = (size_t)SMALL_VROWS * A.Kp;       // (k_small_mw keeps its columns of V in LDS: nothing to clear)
      HIPCHECK(h->small_V.alloc(nv));
      HIPCHECK(hipMemsetAsync(h->small_V.p, 0, nv * sizeof(double), h->stream));
      h->small_np_pad = A.Np_pad;
    }
    A.V = h->small_V.p;
    memset(h->h_small, 0, off_trace);
    A.seq = ++h->small_seq;
    if (mw) {           // several workgroups: 64 landmarks each, two exchanges per LM iteration (ba_small_mw.hpp)
      MwArgs M;
      M.A = A; M.woff = h->mw_woff.p; M.G = h->mw_G;
      // (test hook BA_DEBUG_MW_EXTRA_WG: every barrier waits for one workgroup more than the launch has -- the time-out path)
      M.G_barrier = h->mw_G + (getenv("BA_DEBUG_MW_EXTRA_WG") ? 1 : 0);
      M.slots = h->mw_buf.p; M.sslots = h->mw_buf.p + (size_t)2 * MW_MAX_WG * MW_MSG;
      M.ctr = (unsigned long long*)(h->mw_buf.p + (size_t)2 * MW_MAX_WG * (MW_MSG + MW_SCAL));
      Scope sc(h, BA_K_MISC);
      // instantiated per number of 16-row tiles of [V; z]: 6 Nc + 1 rows
      if (h->Nc <= 5)      BA_LAUNCH(k_small_mw<2>, dim3(h->mw_G), dim3(MW_THREADS), 0, h->stream, M);
      else if (h->Nc <= 7) BA_LAUNCH(k_small_mw<3>, dim3(h->mw_G), dim3(MW_THREADS), 0, h->stream, M);
      else                 BA_LAUNCH(k_small_mw<4>, dim3(h->mw_G), dim3(MW_THREADS), 0, h->stream, M);
      h->stats[BA_STAT_WINDOW_MW_LAUNCHES]++;
    } else {
      Scope sc(h, BA_K_MISC);
      BA_LAUNCH(k_small_lm, dim3(1), dim3(SMALL_THREADS), 0, h->stream, A);
      h->stats[BA_STAT_WINDOW_LM_LAUNCHES]++;
    }
    if (int rc = wait_flag(h, 4, A.seq)) return rc;
    memcpy(sum, h->h_small, sizeof(ba_summary));
    return BA_OK;
  };
  if (int rc = run_once(used_mw)) return rc;
  if (used_mw && sum->status == BA_ERR_HIP) {
    // A workgroup of k_small_mw was not served at a barrier within MW_SPIN_TICKS: its G workgroups were not resident
    // together (another stream, handle or tool holds compute units).  The window is solved again, in this process, by the
    // one-workgroup kernel from the SAME start point: the cameras of the start set were never written (only the final block
    // of a successful solve stores them), the landmarks' start positions are restored from the y slots every workgroup
    // parked them in.  Workgroups of the failed launch that were still queued run (and give up) first: stream order.
    if (h->Np > 0)
      BA_LAUNCH(k_small_restore, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->ptab[A.cur].p, h->Np);
    h->stats[BA_STAT_WINDOW_FALLBACKS]++;
    used_mw = false;
    if (int rc = run_once(false)) return rc;
  }
  if (h->profile) flush_profile(h);
  h->profile = false;
  if (sum->status == BA_ERR_HIP) return fail(BA_ERR_HIP, "the window solver reported a device-side failure");
  if (sum->status == BA_ERR_NUMERIC)
    return fail(BA_ERR_NUMERIC, sum->iterations == 0 ? "non-finite cost at the initial parameters"
                                                     : "non-finite cost / gradient during the solve (LM iteration %d)", sum->iterations);
  if (sum->iterations > 0) {
    h->trace.resize((size_t)sum->iterations);
    memcpy(h->trace.data(), h->h_small + off_trace, sizeof(ba_iter_record) * (size_t)sum->iterations);
  }
  memcpy(&h->cur, h->h_small + off_cur, sizeof(int));
  if (A.stamps && used_mw) {
    long long st[16];
    HIPCHECK(hipMemcpy(st, A.stamps, sizeof st, hipMemcpyDeviceToHost));
    static const char* names[] = {"C1 camera half (slices)", "P1 point half + V (8 lanes / landmark)", "G  [V;z][V;z]^T (MFMA, LDS) + message",
                                  "exchange 1 (barrier + gather)", "S, g", "elimination + back substitution", "camera update + P2", "C3 trial cost (slices)", "exchange 2"};
    for (int k = 0; k < 9; ++k) fprintf(stderr, "[k_small_mw, LM iteration 2, workgroup 0] %-40s %7.2f us\n", names[k], (st[k + 1] - st[k]) * 0.01);
    fprintf(stderr, "[k_small_mw] elimination %.2f us, back substitution %.2f us\n", (st[13] - st[12]) * 0.01, (st[6] - st[13]) * 0.01);
  } else if (A.stamps) {
    long long st[16];
    HIPCHECK(hipMemcpy(st, A.stamps, sizeof st, hipMemcpyDeviceToHost));
    static const char* names[] = {"C1 camera half", "P1 point half + V", "G  [V;z][V;z]^T (MFMA)", "S, g from the tiles", "Cholesky + solves (1 wave)",
                                  "camera update", "P2 back substitution", "C3 trial cost"};
    for (int k = 0; k < 8; ++k) fprintf(stderr, "[k_small_lm, LM iteration 2] %-28s %7.2f us\n", names[k], (st[k + 1] - st[k]) * 0.01);
    fprintf(stderr, "[k_small_lm] wave 0: factor %.2f us, forward %.2f us, backward %.2f us\n", (st[13] - st[4]) * 0.01, (st[14] - st[13]) * 0.01, (st[5] - st[14]) * 0.01);
    fprintf(stderr, "[k_small_lm] shader clock over the iteration: %.2f GHz\n", (double)(st[10] - st[9]) / ((st[8] - st[0]) * 10.0));
  }
  h->linearized = false;
  sum->seconds_total = now_s() - t_begin;
  const double per = sum->iterations ? sum->seconds_total / sum->iterations : 0.0;
  for (auto& r : h->trace) r.seconds = per;
  return BA_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// BAL 9-parameter camera [rvec | t | f k1 k2] (BASELINE config 5 is stated on a BAL problem; the
// reference's only camera is cv2.projectPoints(..., distCoeffs=None), src/bundle_adjuster.py:67).  Cameras (rvec, t) and
// points are the handle's (ba_set_params / ba_get_params); the per-camera (f, k1, k2) travel with the call.  K4 of
// ba_set_problem is not used.  The entry points switch the handle to the BalCam instantiation of every kernel
// (ba_models.hpp) for the duration of the call: same LM / Schur / PCG loop, same device-side verdicts and speculation,
// same multi-rank exchange, 9x9 camera blocks.
static int bal_enter(ba_handle* h, const double* intr) {
  HIPCHECK(hipMemcpyAsync(h->intr[h->cur].p, intr, 3 * (size_t)h->Nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->model = 1;
  h->linearized = false;
  BA_LAUNCH(k_cam_prepare<BalCam>, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->cams[h->cur].p, (const double*)h->intr[h->cur].p,
            h->cs[h->cur].p, h->camA[h->cur].p, h->Nc);
  return BA_OK;
}
// back to the pinhole layout of the camera table (what every other entry point reads), at the current parameters
static void bal_leave(ba_handle* h) {
  h->model = 0;
  h->linearized = false;                      // the linearisation buffers hold 9-parameter blocks
  BA_LAUNCH(k_cam_prepare<Pinhole>, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->cams[h->cur].p, (const double*)h->intr[h->cur].p,
            h->cs[h->cur].p, h->camA[h->cur].p, h->Nc);
}

extern "C" int ba_linearize_bal(ba_handle* h, const double* intr, int32_t loss, double f_scale, double* Hcc, double* bc,
                                double* Hpp, double* bp) {
  if (!h || !intr) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (loss != BA_LOSS_LINEAR && loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss %d", loss);
  if (!(f_scale > 0)) return fail(BA_ERR_INVALID, "f_scale must be positive");
  if (set_device(h)) return BA_ERR_HIP;
  if (int rc = bal_enter(h, intr)) return rc;
  auto body = [&]() -> int {
    launch_lin_cam(h, h->cur, h->lb, loss == BA_LOSS_HUBER, f_scale);
    if (int rc = exchange_partL(h, h->lb)) return rc;
    launch_lin_finalize(h);
    launch_lin_pt(h, h->cur, h->pb, loss == BA_LOSS_HUBER, f_scale, 1.0);
    if (Hcc) HIPCHECK(hipMemcpyAsync(Hcc, h->HccBc.p, BalCam::NH * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (bc) HIPCHECK(hipMemcpyAsync(bc, bc_ptr(h), BalCam::NB * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if ((Hpp || bp) && h->Np) {
      HIPCHECK(h->rbuf.alloc(6 * (size_t)h->Np));
      if (Hpp) {
        BA_LAUNCH(k_unpermute_rows, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->Hpp[h->pb].p, h->slot.p, h->Np, 6, h->rbuf.p);
        HIPCHECK(hipMemcpyAsync(Hpp, h->rbuf.p, 6 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      }
      if (bp) {
        BA_LAUNCH(k_unpermute_rows, dim3((h->Np + 255) / 256), dim3(256), 0, h->stream, h->bp[h->pb].p, h->slot.p, h->Np, 3, h->stage.p);
        HIPCHECK(hipMemcpyAsync(bp, h->stage.p, 3 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      }
    }
    BA_SYNC(h);
    return BA_OK;
  };
  const int rc = body();
  const std::string msg = g_err;
  if (rc != BA_OK) { (void)hipStreamSynchronize(h->stream); h->launch_err = hipSuccess; }
  bal_leave(h);
  (void)hipStreamSynchronize(h->stream);
  g_err = msg;
  return rc;
}

static int solve_impl(ba_handle* h, const ba_options* opts, ba_summary* sum);
extern "C" int ba_solve_bal(ba_handle* h, double* intr, const ba_options* opts, ba_summary* sum) {
  if (!h || !intr || !opts || !sum) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (set_device(h)) return BA_ERR_HIP;
  if (int rc = bal_enter(h, intr)) { h->model = 0; return rc; }
  int rc = ba_solve(h, opts, sum);            // (ba_solve drains the stream and clears the per-solve modes on failure)
  const std::string msg = g_err;
  // the adjusted (f, k1, k2) of the accepted parameter set -- also after a failed solve: the last accepted step stands
  if (hipMemcpyAsync(intr, h->intr[h->cur].p, 3 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess && rc == BA_OK)
    rc = fail(BA_ERR_HIP, "copying the intrinsics back failed");
  // a failure in the middle of the loop may have left the camera state of the current set behind its parameters
  // (k_cam_update writes both for the TRIAL set; the current set is always complete) -- rebuilt here either way
  bal_leave(h);
  if (hipStreamSynchronize(h->stream) != hipSuccess && rc == BA_OK) rc = fail(BA_ERR_HIP, "stream synchronise failed");
  if (rc != BA_OK && !msg.empty()) g_err = msg;
  return rc;
}

static int solve_impl(ba_handle* h, const ba_options* opts, ba_summary* sum) {
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (opts->loss != BA_LOSS_LINEAR && opts->loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss");
  if (!(opts->f_scale > 0) || opts->max_iters < 0 || opts->pcg_max_iters < 1 || !(opts->initial_lambda > 0))
    return fail(BA_ERR_INVALID, "bad options");
  if (opts->jacobian_precision != 0 && opts->jacobian_precision != 1)
    return fail(BA_ERR_INVALID, "jacobian_precision must be 0 (f64) or 1 (f32 blocks, f64 accumulation)");
  if (opts->preconditioner < BA_PRECOND_JACOBI || opts->preconditioner > BA_PRECOND_TWO_LEVEL) return fail(BA_ERR_INVALID, "unknown preconditioner");
  if (!(opts->pcg_model_tol >= 0.0 || opts->pcg_model_tol == -1.0) || opts->pcg_model_min_iters < 0)
    return fail(BA_ERR_INVALID, "bad pcg_model_tol (>= 0, or -1 = automatic) / pcg_model_min_iters");
  if (opts->precond_lag < 0) return fail(BA_ERR_INVALID, "precond_lag must not be negative");
  if (set_device(h)) return BA_ERR_HIP;
  memset(sum, 0, sizeof *sum);
  h->trace.clear();
  // window-sized problems: one launch, exact reduced solve -- no PCG, so preconditioner / jacobian_precision (validated
  // above) have nothing to act on, and no per-solve mode of the multi-kernel path is left switched on behind it
  h->jac_f32 = false;
  h->two_level = false;
  if (h->model == 0 && small_applies(h, opts)) return small_solve(h, opts, sum);
  // (the coarse level is built for the pinhole's 6x6 blocks; the BAL camera runs Schur-Jacobi under that option)
  if (h->model == 0 && opts->preconditioner == BA_PRECOND_TWO_LEVEL && !h->two_level_ok)
    return fail(BA_ERR_STATE, "the two-level preconditioner needs a band-structured problem on a single rank "
                              "(ba_set_problem found none for this one)");
  h->jac_f32 = opts->jacobian_precision == 1;
  roctx_load();
  Range r_solve("ba_solve");
  const bool robust = opts->loss == BA_LOSS_HUBER;
  const bool schur_diag = opts->preconditioner != BA_PRECOND_JACOBI;
  h->two_level = opts->preconditioner == BA_PRECOND_TWO_LEVEL && h->model == 0;
  const double fs = opts->f_scale;
  const int Nc = h->Nc;
  h->profile = opts->profile != 0;
  const double tol2 = opts->pcg_tol * opts->pcg_tol;
  // (automatic: the model test trades inner accuracy for outer iterations -- on band-structured problems a large gain at
  // loose outer tolerances like the reference's ftol = 1e-5, a loss where the caller asks for tight convergence)
  const double model_tol = opts->pcg_model_tol >= 0.0 ? opts->pcg_model_tol : ((h->banded && opts->ftol >= 1e-6) ? 0.5 : 0.0);

  BA_SYNC(h);
  const double t_begin = now_s();
  double sse = 0, cost = 0;
  if (int rc = eval_cost(h, h->cur, robust, fs, &sse, &cost)) return rc;
  if (!std::isfinite(cost)) return fail(BA_ERR_NUMERIC, "non-finite cost at the initial parameters");
  sum->initial_sse = sse;
  sum->initial_cost = cost;
  // nothing to adjust.  Single rank only: a rank of a multi-rank job whose landmark shard is empty still has to
  // join every collective of the loop below (with zero partials), or the other ranks wait for it forever.
  if (!h->multi && (h->Np == 0 || h->Nobs == 0)) {
    sum->final_sse = sse; sum->final_cost = cost; sum->final_lambda = opts->initial_lambda;
    sum->seconds_total = now_s() - t_begin;
    h->profile = false;
    return BA_OK;
  }
  const bool debug_poison = getenv("BA_DEBUG_POISON_TRIAL") != nullptr;     // tests: every trial cost comes out NaN
  // BA_RIDERS: bit 0 = the camera update rides along the back substitution, bit 1 = the scalar fold + verdict rides along
  // the speculated point half (ba_kernels.hpp, "riders"), bit 2 = the PCG probe that finds PCG finished goes on as the back
  // substitution in the same launch (needs bit 0); BA_NO_RIDERS / BA_RIDERS=0: launches of their own (tuning, tests)
  const int riders = getenv("BA_NO_RIDERS") ? 0 : (getenv("BA_RIDERS") ? atoi(getenv("BA_RIDERS")) : 7);
  double lambda = opts->initial_lambda, nu = 2.0;
  int it = 0, status = 0;
  // Schur-Jacobi blocks kept over consecutive damped systems (ba_options.precond_lag): when they were built, how often they
  // have been kept since, and what the inner solves cost with them.  Host-side and deterministic: the rule reads options,
  // dampings and PCG iteration counts only (identical on every rank of a multi-rank job).
  const bool cap_floor = getenv("BA_NO_CAP_FLOOR") == nullptr;      // (switch for A / B measurements)
  const int cu_groups = getenv("BA_CU_GROUPS") ? std::min(8, std::max(1, atoi(getenv("BA_CU_GROUPS")))) : CU_GROUPS;
  double lam_floor = 0.0;
  bool have_precond = false;
  double lam_built = 0.0;
  double last_decrease = 1.0;      // relative cost decrease of the last accepted step
  int kept = 0, pcg_at_build = -1, pcg_last = -1;
  const int lag = (schur_diag && !h->two_level) ? opts->precond_lag : 0;
  bool need_linearize = true;      // a linearisation at the current parameters is needed before the next damped system
  bool have_lin = false;           // ... and buffer sets [lb] / [pb] already hold it (speculated at the trial point that was accepted)
  h->linearized = false;

  while (it < opts->max_iters) {
    double t0 = now_s();
    bool fresh = false;
    Range r_iter("lm_iteration");
    if (need_linearize) {
      Range r_lin("linearize");
      if (!have_lin) {
        launch_lin_cam(h, h->cur, h->lb, robust, fs);
        launch_lin_pt(h, h->cur, h->pb, robust, fs, lambda);          // also Hpp^-1, y0 at this lambda
        // multi-rank: the camera-half partials of a pass launched here still have to be all-reduced
        // (a speculated pass was reduced right behind its launch)
        if (int rc = exchange_partL(h, h->lb)) return rc;
      }
      h->lin_robust = robust; h->lin_fscale = fs;
      need_linearize = false;
      have_lin = false;
      fresh = true;
    }
    // ---- damped system, right-hand side, preconditioner, first PCG vectors
    // (!fresh: the same linearisation damped again after a rejected step; else the point the blocks were built at has moved by
    // a step that lowered the cost by no more than 1 % -- early, large steps always rebuild: there a stale preconditioner
    // costs more PCG iterations, at three passes each, than the one 27-sum pass it saves)
    const bool keep = lag > 0 && have_precond && kept < lag && lambda <= 10.0 * lam_built && lambda >= 0.1 * lam_built &&
                      (pcg_at_build < 0 || pcg_last <= pcg_at_build + pcg_at_build / 2 + 2) && (!fresh || last_decrease <= 1e-2);
    { Range r_damp("damped_system"); if (int rc = damped_system(h, lambda, schur_diag, !fresh, fresh, keep)) return rc; }
    if (keep) ++kept;
    else { have_precond = schur_diag; lam_built = lambda; kept = 0; pcg_at_build = -1; }
    bool gtol_pending = false;     // single rank: max |gradient| lands in host-mapped memory, read at the first PCG verdict
    if (fresh && opts->gtol > 0) {
      // max |gradient| = max(|bc|, |bp|): per-workgroup maxima come out of the point half (partG) and of
      // k_pcg_setup (partGc); single rank: the first PCG probe folds them into host-mapped memory
      gtol_pending = true;
      // (multi-rank: bc is all-reduced, identical on every rank; bp is shard-local -- every rank's maximum came with the
      // damped system's message, exchange_system, and the probe folds those instead of partG)
    }
    double t1 = now_s();
    sum->seconds_linearize += t1 - t0;
    // ---- PCG.  The point pass of iteration k is the probe: it publishes the verdict for k (go on /
    // converged after n iterations) in host-mapped memory when it STARTS.  Only after a "go on" are
    // the camera pass and the vector kernel of k queued (the point pass is still running then),
    // followed at once by the probe of k+1.  Convergence costs one early-exit point pass.  The rule
    // only depends on the (deterministic, rank-identical) verdicts, never on timing.
    int k = 0, pcg_done_iters = -1;
    bool gtol_stop = false;
    const long long base = h->flag_base;
    h->flag_base += opts->pcg_max_iters + 8;
    // K7a + K6 arguments.  The camera update does not feed the back substitution except through the step's vt, which the point
    // workgroups drop into the rows of their LDS windows themselves (vx): it rides along the same launch as extra
    // workgroups (ba_kernels.hpp, "riders") whenever every window is staged in LDS; else it is a launch of its own.
    CamUpdateArgs cu;
    memset(&cu, 0, sizeof cu);
    cu.cams = h->cams[h->cur].p; cu.intr = h->intr[h->cur].p; cu.dc = h->x.p; cu.rpcg = h->r.p; cu.Hcc = h->HccBc.p; cu.bc = bc_ptr(h);
    cu.cs = h->cs[h->cur].p; cu.cams_trial = h->cams[1 - h->cur].p; cu.intr_trial = h->intr[1 - h->cur].p; cu.cs_trial = h->cs[1 - h->cur].p;
    cu.vtil = h->camA[h->cur].p; cu.camA_trial = h->camA[1 - h->cur].p; cu.partC = h->partC.p;
    cu.vx = h->vx.p;
    cu.lam_slot = h->dev_lam.p;
    // (riding workgroups: a multiple of NPART, so that the point workgroups behind them keep their XCD = index mod NPART)
    cu.n_cams = Nc; cu.fixed_cam = h->fixed; cu.groups = cu_groups;
    cu.n_blocks = (((nbv(h) + cu.groups - 1) / cu.groups + NPART - 1) / NPART) * NPART;
    const bool ride = (riders & 1) && all_lds_of(h) && h->Np > 0;
    // The probe that finds PCG finished goes on as the back substitution (pt_schur_body, cu.fuse): the camera-update riders
    // then sit behind the point workgroups of EVERY PCG point pass (they only run in the launch that finds PCG finished), so
    // only where they do not add a round of workgroups to those launches -- a point-pass workgroup has a compute unit to
    // itself (config 5: 256 workgroups, one round; 16 more would be a second one in each of ~100 launches per LM iteration);
    // fp64 blocks only (the back substitution is never computed with the PCG passes' fp32 blocks).  Nothing in the launch
    // waits for a rider, so residency is a matter of speed, not of correctness.
    const int pt_wgs = h->nblkP + h->nblkL;
    // ... and where the inner solves are short: every PCG point pass carries the riders (0.6 us each at C3), the launch they
    // save comes once per LM iteration (5 - 8 us) -- beyond ~16 PCG iterations per LM iteration the separate launch is cheaper
    // (decided from the last inner solve's count: host-side, deterministic, identical on every rank; the bits do not depend on it)
    const bool fuse = ride && (riders & 4) && !h->jac_f32 && (pcg_last < 0 || pcg_last <= 16) &&
                      (pt_wgs + cu.n_blocks + h->n_cu - 1) / h->n_cu == (pt_wgs + h->n_cu - 1) / h->n_cu;
    cu.fuse = fuse ? 1 : 0;
    bool backsub_done = false;
    size_t probe_ev = (size_t)-1, probe_flushes = 0;
    auto launch_point_pass = [&](int kk) {
      probe_ev = h->ev_slot.size(); probe_flushes = h->n_flushes;
      launch_pt_schur(h, robust, 0, kk, tol2, opts->pcg_min_iters, base,
                      (kk == 0 && gtol_pending) ? h->d_scal_host + GMAX_HOST_SLOT : (double*)nullptr, fuse ? &cu : nullptr);
    };
    // BA_IPC: the exchange of the Schur product happens inside k_pcg_step, workgroup by workgroup (ba_kernels.hpp,
    // "device-side all-reduce"); every workgroup's record has to fit its slot of the receive buffers
    const bool use_ipc = h->ipc && h->multi && !h->two_level && nbv(h) <= IPC_MAX_BLOCKS &&
                         (size_t)nbv(h) * (2 + (size_t)nb_of(h) * (h->model ? BalCam::VC : Pinhole::VC)) <= IpcComm::STRIDE;
    auto launch_rest = [&](int kk) -> int {
      launch_cam_schur(h, robust, false, true, kk, tol2, opts->pcg_min_iters);
      // the Schur product of the reduced camera system, summed over the ranks: device-side stores into every peer's receive
      // buffer (inside k_pcg_step), or fold + all-reduce on the base transport
      IpcStep ipc;
      memset(&ipc, 0, sizeof ipc);
      if (use_ipc) {
        IpcComm* c = h->ipc;
        ipc.P = c->peers; ipc.on = 1; ipc.rank = c->rank; ipc.world = c->world;
        ipc.seq = ++c->seq;
        ipc.parity = (int)(ipc.seq & 1);
        ipc.stride = IpcComm::STRIDE;
        h->stats[BA_STAT_IPC_EXCHANGES]++;
      } else if (int rc = exchange_schur(h)) return rc;
      Scope sc(h, BA_K_PCG_UPDATE);
      // (device-side exchange: k_cam_schur's raw partitions, folded inside the kernel; else partition 0 holds the all-reduced sums)
      const int step_parts = use_ipc ? (h->one_part ? 1 : NPART) : nparts_of(h);
#define STEP_ARGS kk, (const double*)p6_ptr(h), step_parts, (const double*)uy_ptr(h), h->Hccd.p, h->Minv.p, h->cs[h->cur].p, Nc, h->fixed, tol2,       \
                  opts->pcg_min_iters, h->x.p, h->r.p, h->p.p, h->s.p, h->z.p, h->camA[h->cur].p, h->partV.p, nbv(h), h->st.p, \
                  h->d_flags, base, (const double*)h->verdict.p
#define STEP_TAIL h->vx.p, model_tol, opts->pcg_model_min_iters, ipc, h->d_flags + 6
      if (h->two_level) {
        BA_LAUNCH((k_pcg_step<Pinhole, true>), dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, STEP_ARGS, h->coarse_rc.p, STEP_TAIL);
        BA_LAUNCH(k_pcg_coarse, dim3(h->n_agg), dim3(VEC_BLOCK), 0, h->stream, kk, (const double*)h->coarseEinv.p,
                  (const double*)h->coarse_rc.p, h->n_agg, (const double*)h->Hccd.p, (const double*)h->cs[h->cur].p, Nc, h->fixed,
                  (const double*)h->r.p, h->z.p, h->camA[h->cur].p, h->partV.p, nbv(h), (const double*)h->verdict.p, 1);
      } else if (h->model) {
        BA_LAUNCH((k_pcg_step<BalCam, false>), dim3(nbv(h)), dim3(VEC_BLOCK), 0, h->stream, STEP_ARGS, (double*)nullptr, STEP_TAIL);
      } else {
        BA_LAUNCH((k_pcg_step<Pinhole, false>), dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, STEP_ARGS, (double*)nullptr, STEP_TAIL);
      }
#undef STEP_ARGS
#undef STEP_TAIL
      return BA_OK;
    };
    Range* r_pcg = new Range("pcg");
    struct RangeGuard { Range*& r; ~RangeGuard() { delete r; r = nullptr; } } r_pcg_guard{r_pcg};
    launch_point_pass(0);
    while (true) {
      if (int rc = wait_flag(h, 0, base + k + 1)) return rc;
      if (h->h_flags[6] == 2) { h->h_flags[6] = 0; return fail(BA_ERR_COMM, "LM iteration %d: a peer's share of the reduced camera system's product did not arrive (BA_IPC)", it); }
      // the gradient maximum was written by a kernel ahead of this probe: visible now
      if (gtol_pending) {
        gtol_pending = false;
        const double gmax = h->h_scal[GMAX_HOST_SLOT];
        if (!std::isfinite(gmax)) return fail(BA_ERR_NUMERIC, "non-finite gradient at LM iteration %d", it);
        if (gmax <= opts->gtol) { gtol_stop = true; break; }
      }
      const long long payload = h->h_flags[1];
      if (payload > 0) {
        pcg_done_iters = (int)payload - 1;
        if (fuse) {                   // the launch that published this verdict is doing the back substitution
          backsub_done = true;
          if (h->profile && probe_flushes == h->n_flushes && probe_ev < h->ev_slot.size()) h->ev_slot[probe_ev] = BA_K_SCHUR_PT_BACKSUB;
        }
        break;
      }
      if (int rc = launch_rest(k)) return rc;
      ++k;
      if (k >= opts->pcg_max_iters) break;
      launch_point_pass(k);
    }
    delete r_pcg; r_pcg = nullptr;
    if (gtol_stop) { status = 3; break; }      // converged by gradient: no step (the queued probe exits on its own)
    // Cap-aware damping.  An inner solve that runs into pcg_max_iters says that at this damping the reduced system is
    // beyond what the preconditioned iteration resolves within its budget (long camera chains at small damping: the drift
    // modes; BASELINE config 5).  The step it leaves is still a descent step (truncated CG), but letting the damping fall
    // further only buys more capped solves: from here on the damping stays at or above three times the value that
    // hit the cap -- one Nielsen step back, where the solve still converged.  The floor travels with the step's verdict
    // (lm_decide): the speculated point half reads the next damping on the device.
    if (cap_floor && k >= opts->pcg_max_iters) { lam_floor = std::max(lam_floor, 3.0 * lambda); h->stats[BA_STAT_CAP_FLOOR_RAISES]++; }
    // ---- step, trial point, gain-ratio scalars
    Range r_step("step");
    if (!backsub_done) {              // (PCG ran into its iteration cap, or the fused form is off)
      cu.fuse = 0;
      if (!ride) {
        Scope sc(h, BA_K_MISC);
        if (h->model) BA_LAUNCH(k_cam_update<BalCam>, dim3(nbv(h)), dim3(VEC_BLOCK), 0, h->stream, cu);
        else          BA_LAUNCH(k_cam_update<Pinhole>, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, cu);
        cu.n_blocks = 0;              // (the arguments still travel: the launch clears the riding verdict's damping word)
      }
      launch_pt_schur(h, robust, 1, 0, 0.0, 0, 0, nullptr, &cu);
    }
    // Speculation: unless this is the last iteration, the cost at the trial point comes out of the camera half of
    // the NEXT linearisation computed there (one pass instead of two), into the other c_w / partL buffers; the step's
    // verdict (gain ratio, next damping) is computed on the device right behind it, and while the host reads it the
    // GPU already runs the point half at the trial point with that damping, into the other point buffers.  An accepted
    // step finds its linearisation done; a rejected one ignores both.
    const bool speculated = (it + 1 < opts->max_iters);
    if (speculated) launch_lin_cam(h, 1 - h->cur, 1 - h->lb, robust, fs, true);
    else            launch_residual(h, 1 - h->cur, robust, fs, nullptr);
    if (debug_poison) BA_LAUNCH(k_poison, dim3(1), dim3(64), 0, h->stream, h->partR.p);
    const long long seq = ++h->step_seq;
    // the step's scalar fold + verdict: single rank with a speculated point half behind it -> workgroup 0 of that launch
    // (the point workgroups pick the next damping up through a device word); else a launch of its own
    const bool ride_scalars = (riders & 2) && speculated && !h->multi && h->Np > 0;
    if (!ride_scalars) launch_scalars(h, true, k, tol2, opts->pcg_min_iters, seq, cost, lambda, lam_floor);
    if (h->multi) {
      // the six sums over ranks, then the verdict on the all-reduced block (same host-mapped mirror + word).  A speculated
      // camera half has to be all-reduced anyway: the six words travel in the header of that message, one collective
      const double* reduced6 = nullptr;
      if (speculated) {
        if (int rc = exchange_partL(h, 1 - h->lb, true)) return rc;
        reduced6 = h->linmsg[1 - h->lb].p;
      } else if (int rc = allreduce(h, h->scal.p, 6)) return rc;
      Scope sc(h, BA_K_MISC);
      BA_LAUNCH(k_decide, dim3(1), dim3(64), 0, h->stream, h->scal.p, reduced6, cost, lambda, lam_floor, h->d_scal_host, h->d_flags + 2, seq);
    }
    if (speculated) {
      ScalarsArgs sa = scalars_args(h, true, k, tol2, opts->pcg_min_iters, seq, cost, lambda, lam_floor);
      sa.on = 1; sa.lam_slot = h->dev_lam.p; sa.err_flag = h->d_flags + 6;
      launch_lin_pt(h, 1 - h->cur, 1 - h->pb, robust, fs, 0.0, h->scal.p + S_LAM_NEXT, ride_scalars ? &sa : nullptr);
    }
    if (int rc = wait_flag(h, 2, seq)) return rc;
    if (h->h_flags[6] != 0) {          // a point workgroup of the speculated pass waited RIDER_WAIT_TICKS for the riding verdict
      h->h_flags[6] = 0;
      return fail(BA_ERR_HIP, "LM iteration %d: the point workgroups of the speculated linearisation were not served by the riding scalar fold", it);
    }
    if (pcg_done_iters < 0) pcg_done_iters = (h->h_scal[S_PCG_FIN] != 0.0) ? (int)h->h_scal[S_PCG_ITERS] : k;
    sum->pcg_iterations += pcg_done_iters;
    pcg_last = pcg_done_iters;
    if (pcg_at_build < 0) pcg_at_build = pcg_done_iters;      // the first inner solve with freshly built blocks
    double t2 = now_s();
    sum->seconds_pcg += t2 - t1;
    const double* S = h->h_scal;
    const double sse_new = S[S_SSE], cost_new = 0.5 * S[S_RHO];
    const double step2 = S[S_PT_DD] + S[S_CAM_DD], x2 = S[S_PT_XX] + S[S_CAM_XX];
    const double rho = S[S_GAIN];               // gain ratio and next damping: decided on the device (lm_decide)
    ++it;
    if (opts->verbose)
      fprintf(stderr, "[ba] it %3d cost %.9e -> %.9e lambda %.3e rho %+.3f pcg %d |step| %.3e\n", it, cost, cost_new,
              lambda, rho, pcg_done_iters, std::sqrt(step2));
    {
      ba_iter_record rec = {};
      rec.iteration = it; rec.accepted = (rho > 0 && std::isfinite(cost_new)) ? 1 : 0; rec.pcg_iterations = pcg_done_iters;
      rec.cost = cost; rec.cost_trial = cost_new; rec.sse_trial = sse_new; rec.lambda = lambda; rec.gain_ratio = rho;
      rec.step_norm = std::sqrt(step2); rec.seconds = now_s() - t0;
      h->trace.push_back(rec);
    }
    bool stop = false;
    if (rho > 0 && std::isfinite(cost_new)) {
      const double dcost = cost - cost_new;
      last_decrease = cost_new > 0.0 ? dcost / cost_new : 1.0;
      h->cur = 1 - h->cur;
      if (speculated) { h->lb = 1 - h->lb; h->pb = 1 - h->pb; have_lin = true; }
      cost = cost_new;
      sse = sse_new;
      sum->accepted++;
      lambda = S[S_LAM_NEXT];
      nu = 2.0;
      need_linearize = true;
      if (dcost <= opts->ftol * cost) { status = 1; stop = true; }
    } else {
      // a trial cost that is not finite even at the largest damping the loop allows cannot be stepped away from
      if (!std::isfinite(cost_new) && lambda >= 1e12)
        return fail(BA_ERR_NUMERIC, "non-finite cost at the trial point of LM iteration %d with the damping at its cap", it);
      lambda = std::min(lambda * nu, 1e12);
      nu *= 2.0;
    }
    if (!stop && std::sqrt(step2) <= opts->xtol * (opts->xtol + std::sqrt(x2))) { status = 2; stop = true; }
    sum->seconds_update += now_s() - t2;
    if (stop) break;
  }
  BA_SYNC(h);
  sum->seconds_total = now_s() - t_begin;
  sum->iterations = it;
  sum->status = status;
  sum->final_sse = sse;
  sum->final_cost = cost;
  sum->final_lambda = lambda;
  if (h->profile) flush_profile(h);
  h->profile = false;
  h->jac_f32 = false;
  h->two_level = false;
  return BA_OK;
}

// ------------------------------------------------------------------ counters, test hooks
extern "C" int ba_get_stat(ba_handle* h, int32_t which, int64_t* value) {
  if (!h || !value || which < 0 || which >= BA_STAT_COUNT) return fail(BA_ERR_INVALID, "bad argument");
  *value = which == BA_STAT_PIXELS_F32 ? (int64_t)(h->have_problem && h->uv_f32) : (int64_t)h->stats[which];
  return BA_OK;
}
// Test hook: the layout ba_set_problem built, copied to the host (tests compare the device build with the host build).
// which: 0 pt_off (Np+1 ints), 1 p_cam, 2 c_pt, 3 c_orig (Nobs ints each), 4 offk (Nc x 9 ints), 5 long_pts (n_long ints),
// 6 blk_win ((nblkP + nblkL) x 2 ints), 7 slot (Np ints), 8 scalars (16 ints: lanes, nblkP, ppb, nblkL, long_spb, long_thr,
// n_long, cam_band, banded, cam_segl, all_lds[0], all_lds[1], lds_bytes[0], lds_bytes[1], build path 0 host / 1 device, mw_ok),
// 9 p_uv, 10 c_uv (Nobs x 2 doubles each).  *n = elements (ints, or doubles for 9 / 10) written.
extern "C" int ba_debug_layout(ba_handle* h, int32_t which, void* out, int64_t capacity, int64_t* n) {
  if (!h || !out || !n) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_problem) return fail(BA_ERR_STATE, "ba_set_problem has not been called");
  if (set_device(h)) return BA_ERR_HIP;
  const void* src = nullptr;
  int64_t cnt = 0;
  size_t esz = sizeof(int);
  int sc[16];
  switch (which) {
    case 0: src = h->pt_off.p; cnt = h->Np + 1; break;
    case 1: src = h->p_cam.p; cnt = h->Nobs; break;
    case 2: src = h->c_pt.p; cnt = h->Nobs; break;
    case 3: src = h->c_orig.p; cnt = h->Nobs; break;
    case 4: src = h->offk.p; cnt = (int64_t)h->Nc * (NPART + 1); break;
    case 5: src = h->long_pts.p; cnt = h->n_long; break;
    case 6: src = h->blk_win.p; cnt = 2 * (int64_t)(h->nblkP + h->nblkL); break;
    case 7: src = h->slot.p; cnt = h->Np; break;
    case 8: {
      const int v[16] = {h->lanes, h->nblkP, h->ppb, h->nblkL, h->long_spb, h->long_thr, h->n_long, h->cam_band, (int)h->banded, h->cam_segl,
                         (int)h->all_lds_m[0], (int)h->all_lds_m[1], (int)h->lds_bytes_m[0], (int)h->lds_bytes_m[1], h->setup_path, (int)h->mw_ok};
      memcpy(sc, v, sizeof sc);
      if (capacity < 16) return fail(BA_ERR_INVALID, "capacity");
      memcpy(out, sc, sizeof sc); *n = 16; return BA_OK;
    }
    case 9: src = h->p_uv.p; cnt = 2 * (int64_t)h->Nobs; esz = sizeof(double); break;
    case 10: src = h->c_uv.p; cnt = 2 * (int64_t)h->Nobs; esz = sizeof(double); break;
    default: return fail(BA_ERR_INVALID, "unknown layout array %d", which);
  }
  if (capacity < cnt) return fail(BA_ERR_INVALID, "capacity %lld < %lld", (long long)capacity, (long long)cnt);
  if ((which == 9 || which == 10) && h->uv_f32) {          // float2 streams: widened here, the caller sees the doubles it gave
    std::vector<float> tmp((size_t)cnt);
    if (cnt > 0) HIPCHECK(hipMemcpyAsync(tmp.data(), src, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    BA_SYNC(h);
    for (int64_t i = 0; i < cnt; ++i) ((double*)out)[i] = (double)tmp[i];
    *n = cnt;
    return BA_OK;
  }
  if (cnt > 0) HIPCHECK(hipMemcpyAsync(out, src, (size_t)cnt * esz, hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  *n = cnt;
  return BA_OK;
}
// workgroups that hold their compute unit's LDS for a bounded time and do nothing (ba_debug_occupy)
__global__ void __launch_bounds__(256) k_debug_occupy(long long ticks, double* __restrict__ sink) {
  extern __shared__ __align__(16) double hog[];
  if (threadIdx.x == 0) hog[0] = (double)blockIdx.x;
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
  if (threadIdx.x == 0 && sink && hog[0] < 0.0) sink[0] = hog[0];          // (keeps the LDS word alive; never true)
}
extern "C" int ba_debug_occupy(ba_handle* h, int32_t n_workgroups, int32_t lds_bytes, double milliseconds) {
  if (!h || n_workgroups < 1 || n_workgroups > 4096 || lds_bytes < 8 || lds_bytes > 160 * 1024 || !(milliseconds > 0) || milliseconds > 2000.0)
    return fail(BA_ERR_INVALID, "bad argument");
  if (set_device(h)) return BA_ERR_HIP;
  if (!h->stream2) HIPCHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  HIPCHECK(hipFuncSetAttribute((const void*)k_debug_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipLaunchKernelGGL(k_debug_occupy, dim3(n_workgroups), dim3(256), (size_t)lds_bytes, h->stream2, (long long)(milliseconds * 1e5), (double*)nullptr);
  HIPCHECK(hipGetLastError());
  return BA_OK;
}

// --------------------------------------------------------- two-view triangulation (row f3)
extern "C" int ba_triangulate(ba_handle* h, const double K[9], const double R_rel[9], const double t_rel[3], int64_t n,
                              const double* pts1, const double* pts2, double* xyz, uint8_t* valid) {
  if (!h || !K || !R_rel || !t_rel || n < 0) return fail(BA_ERR_INVALID, "bad argument");
  if (n == 0) return BA_OK;
  if (!pts1 || !pts2 || !xyz || !valid) return fail(BA_ERR_INVALID, "null point arrays");
  if (set_device(h)) return BA_ERR_HIP;
  TriView v;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      v.P1[4 * r + c] = (c < 3) ? K[3 * r + c] : 0.0;                                   // K [I | 0]
      double s2 = 0.0;
      for (int k = 0; k < 3; ++k) s2 += K[3 * r + k] * ((c < 3) ? R_rel[3 * k + c] : t_rel[k]);
      v.P2[4 * r + c] = s2;                                                             // K [R | t]
    }
  memcpy(v.R, R_rel, sizeof v.R);
  memcpy(v.t, t_rel, sizeof v.t);
  // staging: 2n + 2n doubles in, 3n doubles + n bytes out, in the handle's scratch buffer
  const size_t words = 7 * (size_t)n + ((size_t)n + 7) / 8 + 8;
  HIPCHECK(h->tri.alloc(words));
  double* d1 = h->tri.p;
  double* d2 = d1 + 2 * (size_t)n;
  double* dx = d2 + 2 * (size_t)n;
  uint8_t* dv = (uint8_t*)(dx + 3 * (size_t)n);
  HIPCHECK(hipMemcpyAsync(d1, pts1, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(d2, pts2, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  BA_LAUNCH(k_triangulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, v, n, (const double2*)d1, (const double2*)d2, dx, dv);
  HIPCHECK(hipMemcpyAsync(xyz, dx, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(valid, dv, (size_t)n, hipMemcpyDeviceToHost, h->stream));
  BA_SYNC(h);
  return BA_OK;
}

extern "C" int ba_get_trace(ba_handle* h, ba_iter_record* out, int32_t capacity, int32_t* n) {
  if (!h || !n || capacity < 0) return fail(BA_ERR_INVALID, "bad argument");
  *n = (int32_t)h->trace.size();
  if (out) memcpy(out, h->trace.data(), sizeof(ba_iter_record) * (size_t)std::min<int32_t>(capacity, *n));
  return BA_OK;
}

// ------------------------------------------------------------------------ bench hook
extern "C" int ba_time_kernel(ba_handle* h, int slot, int reps, double* mean_us) {
  if (!h || !mean_us || reps < 1) return fail(BA_ERR_INVALID, "bad argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  const bool saved = h->profile;
  h->profile = false;
  const bool robust = h->lin_robust;
  launch_lin_cam(h, h->cur, h->lb, robust, h->lin_fscale);
  launch_lin_finalize(h);
  launch_lin_pt(h, h->cur, h->pb, robust, h->lin_fscale, 1e-4);
  h->linearized = true;
  if (int rc = damped_system(h, 1e-4, true)) return rc;
  BA_LAUNCH(k_pcg_reset, dim3(1), dim3(64), 0, h->stream, h->st.p, h->partV.p, nbv(h));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  auto once = [&]() {
    switch (slot) {
      case BA_K_RESIDUAL: launch_residual(h, h->cur, robust, h->lin_fscale, nullptr); break;
      case BA_K_LINEARIZE_CAM: launch_lin_cam(h, h->cur, h->lb, robust, h->lin_fscale); break;
      case BA_K_LINEARIZE_PT: launch_lin_pt(h, h->cur, h->pb, robust, h->lin_fscale, 1e-4); break;
      case BA_K_SCHUR_PT: launch_pt_schur(h, robust, 0, 0, -1.0, 1 << 30); break;
      case BA_K_SCHUR_CAM: launch_cam_schur(h, robust, false, false, 0, 0.0, 0); break;
      case BA_K_PRECOND: launch_cam_schur(h, robust, true, false, 0, 0.0, 0); break;
      case BA_K_POINT_INVERT: launch_point_invert(h, 1e-4); break;
      default: break;
    }
  };
  once();
  HIPCHECK(hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) once();
  HIPCHECK(hipEventRecord(e1, h->stream));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  h->profile = saved;
  *mean_us = 1e3 * ms / reps;
  return BA_OK;
}

// --------------------------------------------------------------- diagnostic build only
#ifdef BA_STAMPS
extern "C" int ba_debug_mode(ba_handle* h, int mode) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_mode), &mode, sizeof(int)));
  return BA_OK;
}
// copy the stamps of the last launch of kind 0 (PCG point pass), 1 (PCG camera pass), 2 (k_pcg_step)
extern "C" int ba_debug_stamps(ba_handle* h, int kind, unsigned long long* out, int n_blocks) {
  if (!h || !out || kind < 0 || kind > 2 || n_blocks < 1 || n_blocks > STAMP_BLOCKS) return fail(BA_ERR_INVALID, "bad argument");
  if (set_device(h)) return BA_ERR_HIP;
  BA_SYNC(h);
  HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n_blocks * 8 * sizeof(unsigned long long),
                               (size_t)kind * STAMP_BLOCKS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return BA_OK;
}
#endif

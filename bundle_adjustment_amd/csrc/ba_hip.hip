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
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_kernels.hpp"

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
    "schur_cam", "pcg_step", "precond", "backsub_pt", "misc", "allreduce", "", "", "", ""};
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

// ---------------------------------------------------------------------------- handle
template <typename T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    if (p && n >= count && count > 0) return hipSuccess;
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc((void**)&p, count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct ba_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool have_problem = false, have_params = false, linearized = false;
  int lin_robust = 0;
  double lin_fscale = 1.0;
  int Nc = 0, Np = 0, Nobs = 0, fixed = -1;
  double K4[4] = {1, 1, 0, 0};
  // observation lists
  DBuf<int> cam_off, c_pt, c_orig, pt_off, p_cam;
  DBuf<double2> c_uv, p_uv, c_w, p_w;
  // parameters (current / trial) and camera state
  DBuf<double> cams[2], pts[2], cs[2];
  int cur = 0;
  // normal equations
  DBuf<double> Hcc, bc, Hpp, bp, Hppinv, y0, y, Hccd, Minv, E;
  // PCG
  DBuf<double> gvec, x, r, p, s, z, vtil, comm, partA, partV, partB, partC, partR, scal, rbuf;
  DBuf<PcgState> st;
  int nblkA = 0, nblkV = 0;
  // pinned host mirror for scalars
  double* h_scal = nullptr;
  PcgState* h_st = nullptr;
  // comm
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_slot;
  size_t ev_used = 0;
  ba_profile prof = {};
};

static int set_device(ba_handle* h) {
  HIPCHECK(hipSetDevice(h->device));
  return BA_OK;
}

extern "C" int ba_device_count(int* n) {
  if (!n) return fail(BA_ERR_INVALID, "null argument");
  HIPCHECK(hipGetDeviceCount(n));
  return BA_OK;
}

extern "C" int ba_create(int device_id, ba_handle** out) {
  if (!out) return fail(BA_ERR_INVALID, "null out pointer");
  int n = 0;
  HIPCHECK(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return fail(BA_ERR_INVALID, "device %d not in [0,%d)", device_id, n);
  ba_handle* h = new ba_handle();
  h->device = device_id;
  HIPCHECK(hipSetDevice(device_id));
  HIPCHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHECK(hipHostMalloc((void**)&h->h_scal, 64 * sizeof(double)));
  HIPCHECK(hipHostMalloc((void**)&h->h_st, 2 * sizeof(PcgState)));
  *out = h;
  return BA_OK;
}

static void flush_profile(ba_handle* h);

extern "C" int ba_destroy(ba_handle* h) {
  if (!h) return BA_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(h->nccl);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  DBuf<int>* ib[] = {&h->cam_off, &h->c_pt, &h->c_orig, &h->pt_off, &h->p_cam};
  for (auto b : ib) b->release();
  DBuf<double2>* d2[] = {&h->c_uv, &h->p_uv, &h->c_w, &h->p_w};
  for (auto b : d2) b->release();
  DBuf<double>* db[] = {&h->cams[0], &h->cams[1], &h->pts[0], &h->pts[1], &h->cs[0], &h->cs[1], &h->Hcc, &h->bc,
                        &h->Hpp, &h->bp, &h->Hppinv, &h->y0, &h->y, &h->Hccd, &h->Minv, &h->E, &h->gvec, &h->x,
                        &h->r, &h->p, &h->s, &h->z, &h->vtil, &h->comm, &h->partA, &h->partV, &h->partB,
                        &h->partC, &h->partR, &h->scal, &h->rbuf};
  for (auto b : db) b->release();
  h->st.release();
  if (h->h_scal) (void)hipHostFree(h->h_scal);
  if (h->h_st) (void)hipHostFree(h->h_st);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BA_OK;
}

extern "C" int ba_synchronize(ba_handle* h) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(hipStreamSynchronize(h->stream));
  return BA_OK;
}

// ------------------------------------------------------------------------------ comm
extern "C" int ba_comm_unique_id(void* id128) {
  if (!id128) return fail(BA_ERR_INVALID, "null id buffer");
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
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
  if (world == 1) return BA_OK;
  if (!id128) return fail(BA_ERR_INVALID, "null id buffer");
  if (int rc = load_rccl()) return rc;
  if (set_device(h)) return BA_ERR_HIP;
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
  }
  h->ev_used = 0;
  h->ev_slot.clear();
}
extern "C" int ba_get_profile(ba_handle* h, ba_profile* out) {
  if (!h || !out) return fail(BA_ERR_INVALID, "null argument");
  flush_profile(h);
  *out = h->prof;
  return BA_OK;
}
extern "C" int ba_reset_profile(ba_handle* h) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  flush_profile(h);
  memset(&h->prof, 0, sizeof h->prof);
  return BA_OK;
}

static int allreduce(ba_handle* h, double* buf, size_t count) {
  if (h->world == 1) return BA_OK;
  Scope sc(h, BA_K_ALLREDUCE);
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, h->nccl, h->stream);
  if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return BA_OK;
}

// ---------------------------------------------------------------------------- problem
extern "C" int ba_set_problem(ba_handle* h, int32_t n_cams, int32_t n_pts, int64_t n_obs, const int32_t* cam_idx,
                              const int32_t* pt_idx, const double* uv, const double K4[4], int32_t fixed_cam) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (n_cams <= 0 || n_pts < 0 || n_obs < 0 || n_obs > 0x7fffffffLL) return fail(BA_ERR_INVALID, "bad sizes");
  if (n_obs > 0 && (!cam_idx || !pt_idx || !uv)) return fail(BA_ERR_INVALID, "null observation arrays");
  if (!K4) return fail(BA_ERR_INVALID, "null intrinsics");
  if (fixed_cam < -1 || fixed_cam >= n_cams) return fail(BA_ERR_INVALID, "fixed_cam %d out of range", fixed_cam);
  for (int64_t i = 0; i < n_obs; ++i) {
    if (cam_idx[i] < 0 || cam_idx[i] >= n_cams) return fail(BA_ERR_INVALID, "cam_idx[%lld]=%d out of range", (long long)i, cam_idx[i]);
    if (pt_idx[i] < 0 || pt_idx[i] >= n_pts) return fail(BA_ERR_INVALID, "pt_idx[%lld]=%d out of range", (long long)i, pt_idx[i]);
  }
  if (set_device(h)) return BA_ERR_HIP;
  const int Nc = n_cams, Np = n_pts, No = (int)n_obs;
  // stable counting sorts: by camera and by point
  std::vector<int> cam_off(Nc + 1, 0), pt_off(Np + 1, 0);
  for (int i = 0; i < No; ++i) { cam_off[cam_idx[i] + 1]++; pt_off[pt_idx[i] + 1]++; }
  for (int c = 0; c < Nc; ++c) cam_off[c + 1] += cam_off[c];
  for (int p = 0; p < Np; ++p) pt_off[p + 1] += pt_off[p];
  std::vector<int> c_pt(No), c_orig(No), p_cam(No);
  std::vector<double2> c_uv(No), p_uv(No);
  {
    std::vector<int> cc(cam_off.begin(), cam_off.end() - 1), pc(pt_off.begin(), pt_off.end() - 1);
    for (int i = 0; i < No; ++i) {
      const int a = cc[cam_idx[i]]++;
      c_pt[a] = pt_idx[i]; c_orig[a] = i; c_uv[a] = make_double2(uv[2 * (size_t)i], uv[2 * (size_t)i + 1]);
      const int b = pc[pt_idx[i]]++;
      p_cam[b] = cam_idx[i]; p_uv[b] = c_uv[a];
    }
  }
  h->Nc = Nc; h->Np = Np; h->Nobs = No; h->fixed = fixed_cam;
  memcpy(h->K4, K4, sizeof h->K4);
  h->nblkA = (Np + PT_BLOCK - 1) / PT_BLOCK;
  h->nblkV = (Nc + VEC_BLOCK - 1) / VEC_BLOCK;
  const size_t nobs1 = std::max(No, 1), np1 = std::max(Np, 1);
  HIPCHECK(h->cam_off.alloc(Nc + 1)); HIPCHECK(h->pt_off.alloc(Np + 1));
  HIPCHECK(h->c_pt.alloc(nobs1)); HIPCHECK(h->c_orig.alloc(nobs1)); HIPCHECK(h->p_cam.alloc(nobs1));
  HIPCHECK(h->c_uv.alloc(nobs1)); HIPCHECK(h->p_uv.alloc(nobs1));
  HIPCHECK(h->c_w.alloc(nobs1)); HIPCHECK(h->p_w.alloc(nobs1));
  for (int k = 0; k < 2; ++k) {
    HIPCHECK(h->cams[k].alloc(6 * (size_t)Nc)); HIPCHECK(h->pts[k].alloc(3 * np1)); HIPCHECK(h->cs[k].alloc(CS * (size_t)Nc));
  }
  HIPCHECK(h->Hcc.alloc(27 * (size_t)Nc + 8));   // Hcc (21 Nc) | bc (6 Nc): one all-reduce
  HIPCHECK(h->Hpp.alloc(6 * np1)); HIPCHECK(h->bp.alloc(3 * np1)); HIPCHECK(h->Hppinv.alloc(6 * np1));
  HIPCHECK(h->y0.alloc(3 * np1)); HIPCHECK(h->y.alloc(3 * np1));
  HIPCHECK(h->Hccd.alloc(21 * (size_t)Nc)); HIPCHECK(h->Minv.alloc(21 * (size_t)Nc)); HIPCHECK(h->E.alloc(21 * (size_t)Nc));
  DBuf<double>* v6[] = {&h->gvec, &h->x, &h->r, &h->p, &h->s, &h->z, &h->vtil};
  for (auto b : v6) HIPCHECK(b->alloc(6 * (size_t)Nc));
  HIPCHECK(h->comm.alloc(6 * (size_t)Nc + 8));
  HIPCHECK(h->partA.alloc(std::max(h->nblkA, 1)));
  HIPCHECK(h->partV.alloc(4 * (size_t)h->nblkV));
  HIPCHECK(h->partB.alloc(4 * (size_t)std::max(h->nblkA, 1)));
  HIPCHECK(h->partC.alloc(5 * (size_t)h->nblkV));
  HIPCHECK(h->partR.alloc(2 * (size_t)Nc));
  HIPCHECK(h->scal.alloc(64));
  HIPCHECK(h->st.alloc(2));
  HIPCHECK(hipMemcpyAsync(h->cam_off.p, cam_off.data(), (Nc + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->pt_off.p, pt_off.data(), (Np + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (No > 0) {
    HIPCHECK(hipMemcpyAsync(h->c_pt.p, c_pt.data(), No * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->c_orig.p, c_orig.data(), No * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->p_cam.p, p_cam.data(), No * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->c_uv.p, c_uv.data(), No * sizeof(double2), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->p_uv.p, p_uv.data(), No * sizeof(double2), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(hipStreamSynchronize(h->stream));   // host vectors go out of scope
  h->have_problem = true;
  h->have_params = false;
  h->linearized = false;
  return BA_OK;
}

static void launch_cam_prepare(ba_handle* h, int which) {
  Scope sc(h, BA_K_CAM_PREPARE);
  hipLaunchKernelGGL(k_cam_prepare, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->cams[which].p, h->cs[which].p, h->Nc);
}

extern "C" int ba_set_params(ba_handle* h, const double* cams, const double* pts) {
  if (!h || !cams || (!pts && h->Np > 0)) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_problem) return fail(BA_ERR_STATE, "ba_set_problem has not been called");
  if (set_device(h)) return BA_ERR_HIP;
  h->cur = 0;
  HIPCHECK(hipMemcpyAsync(h->cams[0].p, cams, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->Np > 0)
    HIPCHECK(hipMemcpyAsync(h->pts[0].p, pts, 3 * (size_t)h->Np * sizeof(double), hipMemcpyHostToDevice, h->stream));
  launch_cam_prepare(h, 0);
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->have_params = true;
  h->linearized = false;
  return BA_OK;
}

extern "C" int ba_get_params(ba_handle* h, double* cams, double* pts) {
  if (!h) return fail(BA_ERR_INVALID, "null handle");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  if (cams) HIPCHECK(hipMemcpyAsync(cams, h->cams[h->cur].p, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (pts && h->Np > 0)
    HIPCHECK(hipMemcpyAsync(pts, h->pts[h->cur].p, 3 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return BA_OK;
}

extern "C" int ba_get_rotations(ba_handle* h, double* R) {
  if (!h || !R) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  HIPCHECK(hipMemcpy2DAsync(R, 9 * sizeof(double), h->cs[h->cur].p, CS * sizeof(double), 9 * sizeof(double), h->Nc,
                            hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return BA_OK;
}

// ---------------------------------------------------------------------- launch helpers
static void launch_residual(ba_handle* h, int which, bool robust, double fscale, double* r_out) {
  Scope sc(h, BA_K_RESIDUAL);
  auto kern = robust ? k_residual_cam<true> : k_residual_cam<false>;
  hipLaunchKernelGGL(kern, dim3(h->Nc), dim3(CAM_BLOCK), 0, h->stream, h->cs[which].p, h->pts[which].p, h->cam_off.p,
                     h->c_pt.p, h->c_uv.p, h->c_orig.p, h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, r_out,
                     h->partR.p);
}
static void launch_reduce(ba_handle* h, const double* part, int nrows, int ncols, double* out) {
  Scope sc(h, BA_K_MISC);
  hipLaunchKernelGGL(k_reduce_cols, dim3(1), dim3(256), 0, h->stream, part, nrows, ncols, out);
}
static void launch_linearize(ba_handle* h, bool robust, double fscale) {
  const int w = h->cur;
  {
    Scope sc(h, BA_K_LINEARIZE_CAM);
    auto kern = robust ? k_linearize_cam<true> : k_linearize_cam<false>;
    hipLaunchKernelGGL(kern, dim3(h->Nc), dim3(CAM_BLOCK), 0, h->stream, h->cs[w].p, h->pts[w].p, h->cam_off.p,
                       h->c_pt.p, h->c_uv.p, h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, h->fixed, h->Hcc.p,
                       h->Hcc.p + 21 * (size_t)h->Nc, h->c_w.p);
  }
  if (h->Np > 0) {
    Scope sc(h, BA_K_LINEARIZE_PT);
    auto kern = robust ? k_linearize_pt<true> : k_linearize_pt<false>;
    hipLaunchKernelGGL(kern, dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, h->cs[w].p, h->pts[w].p, h->pt_off.p,
                       h->p_cam.p, h->p_uv.p, h->K4[0], h->K4[1], h->K4[2], h->K4[3], fscale, h->Np, h->Hpp.p,
                       h->bp.p, h->p_w.p);
  }
}
static double* bc_ptr(ba_handle* h) { return h->Hcc.p + 21 * (size_t)h->Nc; }

static void launch_damp(ba_handle* h, double lambda) {
  {
    Scope sc(h, BA_K_MISC);
    hipLaunchKernelGGL(k_damp_cameras, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->Hcc.p, lambda, h->Nc,
                       h->fixed, h->Hccd.p);
  }
  if (h->Np > 0) {
    Scope sc(h, BA_K_POINT_INVERT);
    hipLaunchKernelGGL(k_point_invert, dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, h->Hpp.p, h->bp.p, lambda,
                       h->Np, h->Hppinv.p, h->y0.p);
  }
}
// camera pass on `yvec`; MODE 1 = unconditional (rhs / test hooks), MODE 0 = PCG iteration k
static void launch_schur_cam(ba_handle* h, bool robust, int mode, const double* yvec, int k, double tol2, int min_iters) {
  Scope sc(h, BA_K_SCHUR_CAM);
  const int w = h->cur;
#define SC_ARGS h->cs[w].p, h->pts[w].p, h->cam_off.p, h->c_pt.p, h->c_w.p, yvec, h->K4[0], h->K4[1], h->Nc, h->fixed, \
                h->comm.p, h->partA.p, h->nblkA, k, h->st.p, h->partV.p, h->nblkV, tol2, min_iters
  if (mode == 0) {
    if (robust) hipLaunchKernelGGL((k_schur_cam<true, 0>), dim3(h->Nc + 1), dim3(CAM_BLOCK), 0, h->stream, SC_ARGS);
    else        hipLaunchKernelGGL((k_schur_cam<false, 0>), dim3(h->Nc + 1), dim3(CAM_BLOCK), 0, h->stream, SC_ARGS);
  } else {
    if (robust) hipLaunchKernelGGL((k_schur_cam<true, 1>), dim3(h->Nc + 1), dim3(CAM_BLOCK), 0, h->stream, SC_ARGS);
    else        hipLaunchKernelGGL((k_schur_cam<false, 1>), dim3(h->Nc + 1), dim3(CAM_BLOCK), 0, h->stream, SC_ARGS);
  }
#undef SC_ARGS
}
// point pass with camera vector vtil; MODE 0 = PCG (y, partA), MODE 1 = back substitution
static void launch_schur_pt(ba_handle* h, bool robust, int mode, int k, double tol2, int min_iters) {
  if (h->Np == 0) return;
  Scope sc(h, mode == 0 ? BA_K_SCHUR_PT : BA_K_BACKSUB);
  const int w = h->cur;
#define SP_ARGS h->cs[w].p, h->pts[w].p, h->pt_off.p, h->p_cam.p, h->p_w.p, h->vtil.p, h->Hppinv.p, h->K4[0], h->K4[1], \
                h->Np, h->fixed, h->y.p, h->partA.p, k, h->st.p, h->partV.p, h->nblkV, tol2, min_iters, h->y0.p,       \
                h->Hpp.p, h->bp.p, h->pts[1 - w].p, h->partB.p
  if (mode == 0) {
    if (robust) hipLaunchKernelGGL((k_schur_pt<true, 0>), dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, SP_ARGS);
    else        hipLaunchKernelGGL((k_schur_pt<false, 0>), dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, SP_ARGS);
  } else {
    if (robust) hipLaunchKernelGGL((k_schur_pt<true, 1>), dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, SP_ARGS);
    else        hipLaunchKernelGGL((k_schur_pt<false, 1>), dim3(h->nblkA), dim3(PT_BLOCK), 0, h->stream, SP_ARGS);
  }
#undef SP_ARGS
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
  launch_reduce(h, h->partR.p, h->Nc, 2, h->scal.p);
  if (int rc = allreduce(h, h->scal.p, 2)) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (rdev) HIPCHECK(hipMemcpyAsync(r, rdev, 2 * (size_t)h->Nobs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
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
  launch_linearize(h, loss == BA_LOSS_HUBER, f_scale);
  if (int rc = allreduce(h, h->Hcc.p, 27 * (size_t)h->Nc)) return rc;
  h->linearized = true;
  h->lin_robust = (loss == BA_LOSS_HUBER);
  h->lin_fscale = f_scale;
  if (Hcc) HIPCHECK(hipMemcpyAsync(Hcc, h->Hcc.p, 21 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (bc) HIPCHECK(hipMemcpyAsync(bc, bc_ptr(h), 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (Hpp && h->Np) HIPCHECK(hipMemcpyAsync(Hpp, h->Hpp.p, 6 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (bp && h->Np) HIPCHECK(hipMemcpyAsync(bp, h->bp.p, 3 * (size_t)h->Np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return BA_OK;
}

// --------------------------------------------------------------------- K4 test hooks
extern "C" int ba_schur_rhs(ba_handle* h, double lambda, double* g) {
  if (!h || !g) return fail(BA_ERR_INVALID, "null argument");
  if (!h->linearized) return fail(BA_ERR_STATE, "ba_linearize first");
  if (set_device(h)) return BA_ERR_HIP;
  launch_damp(h, lambda);
  launch_schur_cam(h, h->lin_robust, 1, h->y0.p, 0, 0.0, 0);
  if (int rc = allreduce(h, h->comm.p, 6 * (size_t)h->Nc + 1)) return rc;
  {
    Scope sc(h, BA_K_PRECOND);
    hipLaunchKernelGGL(k_precond_invert, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->Hccd.p,
                       (const double*)nullptr, h->Nc, h->Minv.p);
  }
  {
    Scope sc(h, BA_K_MISC);
    hipLaunchKernelGGL(k_pcg_init, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, bc_ptr(h), h->comm.p, h->Hccd.p,
                       h->Minv.p, h->cs[h->cur].p, h->Nc, h->fixed, h->gvec.p, h->x.p, h->r.p, h->p.p, h->s.p, h->z.p,
                       h->vtil.p, h->partV.p, h->st.p);
  }
  HIPCHECK(hipMemcpyAsync(g, h->gvec.p, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return BA_OK;
}

extern "C" int ba_schur_apply(ba_handle* h, double lambda, const double* v, double* out) {
  if (!h || !v || !out) return fail(BA_ERR_INVALID, "null argument");
  if (!h->linearized) return fail(BA_ERR_STATE, "ba_linearize first");
  if (set_device(h)) return BA_ERR_HIP;
  launch_damp(h, lambda);
  HIPCHECK(hipMemcpyAsync(h->x.p, v, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    Scope sc(h, BA_K_MISC);
    hipLaunchKernelGGL(k_vtil, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->x.p, h->cs[h->cur].p, h->Nc,
                       h->fixed, h->vtil.p);
    // a PCG state that is not "done" so MODE 0 of the point pass runs
    hipLaunchKernelGGL(k_pcg_reset, dim3(1), dim3(64), 0, h->stream, h->st.p, h->partV.p, h->nblkV);
  }
  // point pass writes y = Hppinv W^T v (tol2 < 0 and a huge min_iters keep pcg_finished false)
  launch_schur_pt(h, h->lin_robust, 0, 0, -1.0, 1 << 30);
  launch_schur_cam(h, h->lin_robust, 1, h->y.p, 0, 0.0, 0);
  if (int rc = allreduce(h, h->comm.p, 6 * (size_t)h->Nc + 1)) return rc;
  {
    Scope sc(h, BA_K_MISC);
    hipLaunchKernelGGL(k_schur_combine, dim3((h->Nc + 63) / 64), dim3(64), 0, h->stream, h->Hccd.p, h->x.p, h->comm.p,
                       h->Nc, h->fixed, h->z.p);
  }
  HIPCHECK(hipMemcpyAsync(out, h->z.p, 6 * (size_t)h->Nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
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
  o->pcg_check_every = 4;
  o->profile = 0;
  o->verbose = 0;
  return BA_OK;
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// scalars block layout (device `scal`, pinned mirror `h_scal`)
//  [0] sse  [1] rho-sum  [2] pt g.d  [3] pt dDd  [4] pt |d|^2  [5] pt |x|^2      <- all-reduced (sum)
//  [8] cam g.d  [9] cam dDd  [10] dc.r  [11] cam |d|^2  [12] cam |x|^2          <- replicated
//  [16] max|bc|  [17] max|bp|
static int eval_cost(ba_handle* h, int which, bool robust, double fscale, double* sse, double* cost) {
  launch_residual(h, which, robust, fscale, nullptr);
  launch_reduce(h, h->partR.p, h->Nc, 2, h->scal.p);
  if (int rc = allreduce(h, h->scal.p, 2)) return rc;
  HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  *sse = h->h_scal[0];
  *cost = 0.5 * h->h_scal[1];
  return BA_OK;
}

extern "C" int ba_solve(ba_handle* h, const ba_options* opts, ba_summary* sum) {
  if (!h || !opts || !sum) return fail(BA_ERR_INVALID, "null argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "ba_set_problem / ba_set_params first");
  if (opts->loss != BA_LOSS_LINEAR && opts->loss != BA_LOSS_HUBER) return fail(BA_ERR_INVALID, "unknown loss");
  if (!(opts->f_scale > 0) || opts->max_iters < 0 || opts->pcg_max_iters < 1 || !(opts->initial_lambda > 0))
    return fail(BA_ERR_INVALID, "bad options");
  if (opts->jacobian_precision != 0) return fail(BA_ERR_INVALID, "jacobian_precision %d not built", opts->jacobian_precision);
  if (set_device(h)) return BA_ERR_HIP;
  memset(sum, 0, sizeof *sum);
  const bool robust = opts->loss == BA_LOSS_HUBER;
  const double fs = opts->f_scale;
  const int Nc = h->Nc;
  h->profile = opts->profile != 0;
  const int check_every = std::max(1, opts->pcg_check_every);
  const double tol2 = opts->pcg_tol * opts->pcg_tol;

  HIPCHECK(hipStreamSynchronize(h->stream));
  const double t_begin = now_s();
  double sse = 0, cost = 0;
  if (int rc = eval_cost(h, h->cur, robust, fs, &sse, &cost)) return rc;
  if (!std::isfinite(cost)) return fail(BA_ERR_NUMERIC, "non-finite cost at the initial parameters");
  sum->initial_sse = sse;
  sum->initial_cost = cost;
  double lambda = opts->initial_lambda, nu = 2.0;
  int it = 0, status = 0;
  bool need_linearize = true;

  while (it < opts->max_iters) {
    double t0 = now_s();
    if (need_linearize) {
      launch_linearize(h, robust, fs);
      if (int rc = allreduce(h, h->Hcc.p, 27 * (size_t)Nc)) return rc;
      h->linearized = true; h->lin_robust = robust; h->lin_fscale = fs;
      need_linearize = false;
      if (opts->gtol > 0) {
        {
          Scope sc(h, BA_K_MISC);
          hipLaunchKernelGGL(k_absmax, dim3(1), dim3(256), 0, h->stream, bc_ptr(h), 6 * (size_t)Nc, h->scal.p + 16);
          hipLaunchKernelGGL(k_absmax, dim3(1), dim3(256), 0, h->stream, h->bp.p, 3 * (size_t)h->Np, h->scal.p + 17);
        }
        HIPCHECK(hipMemcpyAsync(h->h_scal + 16, h->scal.p + 16, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        double gmax = std::max(h->h_scal[16], h->h_scal[17]);
        if (h->world > 1) {   // max over shards of the point gradient: tiny host-visible all-reduce of a sum is not a max;
          // every rank sees the same bc (all-reduced); bp is shard-local, so fold it through RCCL max
          HIPCHECK(hipMemcpyAsync(h->scal.p + 18, &gmax, sizeof(double), hipMemcpyHostToDevice, h->stream));
          ncclResult_t r = g_rccl.AllReduce(h->scal.p + 18, h->scal.p + 18, 1, ncclDouble, ncclMax, h->nccl, h->stream);
          if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclAllReduce(max) failed");
          HIPCHECK(hipMemcpyAsync(h->h_scal + 18, h->scal.p + 18, sizeof(double), hipMemcpyDeviceToHost, h->stream));
          HIPCHECK(hipStreamSynchronize(h->stream));
          gmax = h->h_scal[18];
        }
        if (gmax <= opts->gtol) { status = 3; break; }
      }
    }
    // ---- damped system, right-hand side, preconditioner
    launch_damp(h, lambda);
    launch_schur_cam(h, robust, 1, h->y0.p, 0, 0.0, 0);
    if (int rc = allreduce(h, h->comm.p, 6 * (size_t)Nc + 1)) return rc;
    if (opts->preconditioner == BA_PRECOND_SCHUR_JACOBI) {
      {
        Scope sc(h, BA_K_PRECOND);
        auto kern = robust ? k_schur_diag<true> : k_schur_diag<false>;
        hipLaunchKernelGGL(kern, dim3(Nc), dim3(CAM_BLOCK), 0, h->stream, h->cs[h->cur].p, h->pts[h->cur].p,
                           h->cam_off.p, h->c_pt.p, h->c_w.p, h->Hppinv.p, h->K4[0], h->K4[1], h->fixed, h->E.p);
      }
      if (int rc = allreduce(h, h->E.p, 21 * (size_t)Nc)) return rc;
    }
    {
      Scope sc(h, BA_K_PRECOND);
      hipLaunchKernelGGL(k_precond_invert, dim3((Nc + 63) / 64), dim3(64), 0, h->stream, h->Hccd.p,
                         opts->preconditioner == BA_PRECOND_SCHUR_JACOBI ? (const double*)h->E.p : (const double*)nullptr,
                         Nc, h->Minv.p);
    }
    {
      Scope sc(h, BA_K_PCG_UPDATE);
      hipLaunchKernelGGL(k_pcg_init, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, bc_ptr(h), h->comm.p, h->Hccd.p,
                         h->Minv.p, h->cs[h->cur].p, Nc, h->fixed, h->gvec.p, h->x.p, h->r.p, h->p.p, h->s.p, h->z.p,
                         h->vtil.p, h->partV.p, h->st.p);
    }
    double t1 = now_s();
    sum->seconds_linearize += t1 - t0;
    // ---- PCG
    int k = 0, pcg_done_iters = -1;
    while (k < opts->pcg_max_iters) {
      const int kend = std::min(opts->pcg_max_iters, k + check_every);
      for (; k < kend; ++k) {
        launch_schur_pt(h, robust, 0, k, tol2, opts->pcg_min_iters);
        launch_schur_cam(h, robust, 0, h->y.p, k, tol2, opts->pcg_min_iters);
        if (int rc = allreduce(h, h->comm.p, 6 * (size_t)Nc + 1)) return rc;
        Scope sc(h, BA_K_PCG_UPDATE);
        hipLaunchKernelGGL(k_pcg_step, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, k, h->comm.p, h->Hccd.p,
                           h->Minv.p, h->cs[h->cur].p, Nc, h->fixed, tol2, opts->pcg_min_iters, h->x.p, h->r.p, h->p.p,
                           h->s.p, h->z.p, h->vtil.p, h->partV.p, h->nblkV, h->st.p);
      }
      HIPCHECK(hipMemcpyAsync(h->h_st, h->st.p, 2 * sizeof(PcgState), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      const PcgState& s = h->h_st[k & 1];
      if (s.done) { pcg_done_iters = s.iters; break; }
    }
    if (pcg_done_iters < 0) pcg_done_iters = k;
    sum->pcg_iterations += pcg_done_iters;
    double t2 = now_s();
    sum->seconds_pcg += t2 - t1;
    // ---- step, trial point, gain ratio
    {
      Scope sc(h, BA_K_MISC);
      hipLaunchKernelGGL(k_cam_update, dim3(h->nblkV), dim3(VEC_BLOCK), 0, h->stream, h->cams[h->cur].p, h->x.p, h->r.p,
                         h->Hcc.p, bc_ptr(h), h->cs[h->cur].p, Nc, h->fixed, h->cams[1 - h->cur].p, h->vtil.p, h->partC.p);
    }
    launch_schur_pt(h, robust, 1, 0, 0.0, 0);
    launch_cam_prepare(h, 1 - h->cur);
    launch_residual(h, 1 - h->cur, robust, fs, nullptr);
    launch_reduce(h, h->partR.p, Nc, 2, h->scal.p);
    if (h->Np > 0) launch_reduce(h, h->partB.p, h->nblkA, 4, h->scal.p + 2);
    launch_reduce(h, h->partC.p, h->nblkV, 5, h->scal.p + 8);
    if (int rc = allreduce(h, h->scal.p, 6)) return rc;
    HIPCHECK(hipMemcpyAsync(h->h_scal, h->scal.p, 16 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    const double* S = h->h_scal;
    const double sse_new = S[0], cost_new = 0.5 * S[1];
    const double gTd = S[2] + S[8], dDd = S[3] + S[9], dcr = S[10];
    const double step2 = S[4] + S[11], x2 = S[5] + S[12];
    const double model = 0.5 * (lambda * dDd - gTd + dcr);
    const double rho = (model > 0 && std::isfinite(cost_new)) ? (cost - cost_new) / model : -1.0;
    ++it;
    if (opts->verbose)
      fprintf(stderr, "[ba] it %3d cost %.9e -> %.9e lambda %.3e rho %+.3f pcg %d |step| %.3e\n", it, cost, cost_new,
              lambda, rho, pcg_done_iters, std::sqrt(step2));
    bool stop = false;
    if (rho > 0 && std::isfinite(cost_new)) {
      const double dcost = cost - cost_new;
      h->cur = 1 - h->cur;
      cost = cost_new;
      sse = sse_new;
      sum->accepted++;
      lambda = std::max(lambda * std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)), 1e-12);
      nu = 2.0;
      need_linearize = true;
      h->linearized = false;
      if (dcost <= opts->ftol * cost) { status = 1; stop = true; }
    } else {
      lambda = std::min(lambda * nu, 1e12);
      nu *= 2.0;
    }
    if (!stop && std::sqrt(step2) <= opts->xtol * (opts->xtol + std::sqrt(x2))) { status = 2; stop = true; }
    sum->seconds_update += now_s() - t2;
    if (stop) break;
  }
  HIPCHECK(hipStreamSynchronize(h->stream));
  sum->seconds_total = now_s() - t_begin;
  sum->iterations = it;
  sum->status = status;
  sum->final_sse = sse;
  sum->final_cost = cost;
  sum->final_lambda = lambda;
  if (h->profile) flush_profile(h);
  h->profile = false;
  return BA_OK;
}

// ------------------------------------------------------------------------ bench hook
extern "C" int ba_time_kernel(ba_handle* h, int slot, int reps, double* mean_us) {
  if (!h || !mean_us || reps < 1) return fail(BA_ERR_INVALID, "bad argument");
  if (!h->have_params) return fail(BA_ERR_STATE, "no parameters set");
  if (set_device(h)) return BA_ERR_HIP;
  const bool robust = h->lin_robust;
  if ((slot == BA_K_SCHUR_PT || slot == BA_K_SCHUR_CAM || slot == BA_K_BACKSUB) && !h->linearized) {
    launch_linearize(h, robust, h->lin_fscale);
    launch_damp(h, 1e-4);
    h->linearized = true;
  }
  const bool saved = h->profile;
  h->profile = false;
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_pcg_reset, dim3(1), dim3(64), 0, h->stream, h->st.p, h->partV.p, h->nblkV);
  auto once = [&]() {
    switch (slot) {
      case BA_K_RESIDUAL: launch_residual(h, h->cur, robust, h->lin_fscale, nullptr); break;
      case BA_K_LINEARIZE_CAM:
      case BA_K_LINEARIZE_PT: launch_linearize(h, robust, h->lin_fscale); break;
      case BA_K_SCHUR_PT: launch_schur_pt(h, robust, 0, 0, -1.0, 1 << 30); break;
      case BA_K_SCHUR_CAM: launch_schur_cam(h, robust, 1, h->y.p, 0, 0.0, 0); break;
      case BA_K_POINT_INVERT: launch_damp(h, 1e-4); break;
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

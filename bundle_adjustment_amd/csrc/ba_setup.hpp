// ba_set_problem on the device: both observation orderings, their offsets, the per-workgroup camera windows and the layout
// statistics derived on the GPU from ONE upload of the caller's (cam_idx, pt_idx, uv) -- what replaces, for large
// problems, the host counting sorts that took 13-30 ms at the headline size for a 1.4-5 ms solve (the reference walks
// the same lists in Python: _gather_local_data / _prepare_sparsity_matrix, src/bundle_adjuster.py:74-120, 195-218).
//
// The result is BIT-EQUAL to the host build (tests/test_gpu_setup.py compares every array): the host sorts are stable
// counting sorts, i.e. they define for every observation a position that depends on the data only, and the kernels
// below reproduce those positions without relying on the order in which atomics are served:
//   point order    histogram of pt_idx (integer atomics: a count does not depend on the order of its increments),
//                  exclusive scan -> pt_off, scatter of the observation numbers into their point's segment in WHATEVER
//                  order the atomics hand out, then every segment sorted ascending by observation number (unique keys:
//                  the sorted segment is the stable order);
//   bank-aware visiting order (2-lane point passes): the host's greedy rule per group of eight points, one thread per group;
//   camera order   the same with the POSITIONS of the point-ordered list as keys: histogram of p_cam, scan -> cam_off,
//                  unordered scatter, every camera's segment sorted ascending by position (= ascending point, ties in
//                  visiting order: what the host's second counting sort produces);
//   long tracks    flags, exclusive scan, scatter (ascending point index);
//   windows, band / XCD statistics: per-workgroup reductions.
// Segment sorts are rank sorts (every key's rank = number of smaller keys; keys are unique): O(n^2 / lanes) per segment,
// which is nothing for tracks of ~10 observations and ~15 us for a camera with a thousand.
#pragma once
#include "ba_kernels.hpp"

namespace ba {

constexpr int SETUP_SCAN_ITEMS = 4;                      // items per thread in the scan kernels
constexpr int SETUP_SCAN_BLOCK = 1024 * SETUP_SCAN_ITEMS;
constexpr int SETUP_RANK_LDS = 12288;                    // keys of a segment staged in LDS by the workgroup rank sort (48 KB)

// ---- validation + histogram of the point index -----------------------------------------------------------------
// bad[0] = smallest observation number with an index out of range (INT_MAX: none)
__global__ void __launch_bounds__(256)
k_setup_hist(const int* __restrict__ cam_idx, const int* __restrict__ pt_idx, int n_obs, int n_cams, int n_pts,
             int* __restrict__ cnt, int* __restrict__ bad, const double2* __restrict__ uv, int* __restrict__ wide) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_obs) return;
  const int c = cam_idx[i], p = pt_idx[i];
  if (uv) {                                   // a pixel that is not a float32 value: the streams stay double2 (UvArr)
    const double2 v = uv[i];
    if (!(pixel_is_f32(v.x) && pixel_is_f32(v.y))) __hip_atomic_store(wide, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (c < 0 || c >= n_cams || p < 0 || p >= n_pts) { atomicMin(bad, i); return; }
  atomicAdd(cnt + p, 1);
}

// ---- exclusive scan of n ints (n <= 1024 * SETUP_SCAN_BLOCK): block sums, scan of the block sums, final pass -------
__global__ void __launch_bounds__(1024)
k_scan_block_sums(const int* __restrict__ in, int n, int* __restrict__ bsum) {
  __shared__ int sm[16];
  const int base = blockIdx.x * SETUP_SCAN_BLOCK + threadIdx.x * SETUP_SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int q = 0; q < SETUP_SCAN_ITEMS; ++q) s += (base + q < n) ? in[base + q] : 0;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += sm[w]; bsum[blockIdx.x] = t; }
}
// one workgroup: exclusive scan of the (<= 1024) block sums in place; total -> bsum[nb]
__global__ void __launch_bounds__(1024)
k_scan_top(int* __restrict__ bsum, int nb) {
  __shared__ int sm[1024];
  const int t = threadIdx.x;
  const int own = t < nb ? bsum[t] : 0;
  sm[t] = own;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? sm[t - o] : 0;
    __syncthreads();
    sm[t] += v;
    __syncthreads();
  }
  if (t < nb) bsum[t] = sm[t] - own;                      // inclusive - own = exclusive
  if (t == 1023) bsum[nb] = sm[1023];                     // (entries past nb are 0: the last inclusive value is the total)
}
// out[i] = sum of in[0 .. i); out[n] = total.  in and out may be the same array.
__global__ void __launch_bounds__(1024)
k_scan_final(const int* __restrict__ in, int n, const int* __restrict__ bsum, int* __restrict__ out) {
  __shared__ int sm[1024];
  const int t = threadIdx.x;
  const int base = blockIdx.x * SETUP_SCAN_BLOCK + t * SETUP_SCAN_ITEMS;
  int v[SETUP_SCAN_ITEMS], s = 0;
#pragma unroll
  for (int q = 0; q < SETUP_SCAN_ITEMS; ++q) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
  sm[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int u = t >= o ? sm[t - o] : 0;
    __syncthreads();
    sm[t] += u;
    __syncthreads();
  }
  int run = bsum[blockIdx.x] + sm[t] - s;                 // exclusive prefix of this thread's first item
#pragma unroll
  for (int q = 0; q < SETUP_SCAN_ITEMS; ++q) {
    if (base + q < n) out[base + q] = run;
    run += v[q];
  }
  if (blockIdx.x == gridDim.x - 1 && t == 1023) out[n] = bsum[gridDim.x];
}

// ---- unordered scatter of the observation numbers into their point's segment --------------------------------------
__global__ void __launch_bounds__(256)
k_setup_scatter_pt(const int* __restrict__ pt_idx, int n_obs, const int* __restrict__ pt_off, int* __restrict__ fill,
                   int* __restrict__ seg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_obs) return;
  const int p = pt_idx[i];
  seg[pt_off[p] + atomicAdd(fill + p, 1)] = i;
}

// ---- every point's segment sorted ascending by observation number; p_cam / p_pt of the sorted positions; per-point
//      track length histogram (-> median), camera span (band statistic).  Thread per point for tracks of at most 64
//      observations; longer ones are appended to `big` (any order) for k_setup_sort_big.
__global__ void __launch_bounds__(256)
k_setup_sort_pt(const int* __restrict__ pt_off, int n_pts, const int* __restrict__ seg, const int* __restrict__ cam_idx,
                int* __restrict__ p_src, int* __restrict__ p_cam, int* __restrict__ p_pt, int* __restrict__ len_hist, int hist_bins,
                unsigned long long* __restrict__ span_sum, int* __restrict__ n_tracks, int* __restrict__ big, int* __restrict__ n_big) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  long long span = 0;
  int tracks = 0;
  int b = 0, n = 0;
  if (p < n_pts) { b = pt_off[p]; n = pt_off[p + 1] - b; }
  {   // track-length histogram, one atomic per DISTINCT length in the wave (synthetic data: every track has the same length --
      // a hundred thousand atomics on one word took most of this kernel's millisecond)
    const int bin = p < n_pts ? min(n, hist_bins - 1) : -1;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(bin >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lb = __shfl(bin, leader, 64);
      const unsigned long long same = __ballot(bin == lb);
      if (lane == leader) atomicAdd(len_hist + lb, (int)__popcll(same));
      todo &= ~same;
    }
  }
  if (p < n_pts) {
    if (n > 64) {
      big[atomicAdd(n_big, 1)] = p;
    } else if (n > 0) {
      int lo = 0x7fffffff, hi = -1;
      if (n <= 16) {
        // the usual track: the segment once into registers (the quadratic rank loop below reads it n times from memory, one
        // cache line per lane and load), ranks by compare-and-count, pads (INT_MAX) never count
        int e[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) e[a] = a < n ? seg[b + a] : 0x7fffffff;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          if (a < n) {
            int r = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) r += e[q] < e[a];
            const int c = cam_idx[e[a]];
            p_src[b + r] = e[a]; p_cam[b + r] = c; p_pt[b + r] = p;
            lo = min(lo, c); hi = max(hi, c);
          }
        }
      } else {
        for (int a = 0; a < n; ++a) {
          const int e = seg[b + a];
          int r = 0;
          for (int q = 0; q < n; ++q) r += seg[b + q] < e;
          const int c = cam_idx[e];
          p_src[b + r] = e; p_cam[b + r] = c; p_pt[b + r] = p;
          lo = min(lo, c); hi = max(hi, c);
        }
      }
      span = hi - lo; tracks = 1;
    }
  }
  // one pair of atomics per wave
  for (int o = 32; o > 0; o >>= 1) { span += __shfl_down(span, o, 64); tracks += __shfl_down(tracks, o, 64); }
  if ((threadIdx.x & 63) == 0 && tracks) { atomicAdd(span_sum, (unsigned long long)span); atomicAdd(n_tracks, tracks); }
}
// long tracks: one 256-thread workgroup per entry of `big`
__global__ void __launch_bounds__(256)
k_setup_sort_big(const int* __restrict__ big, const int* __restrict__ n_big, const int* __restrict__ pt_off, const int* __restrict__ seg,
                 const int* __restrict__ cam_idx, int* __restrict__ p_src, int* __restrict__ p_cam, int* __restrict__ p_pt,
                 unsigned long long* __restrict__ span_sum, int* __restrict__ n_tracks) {
  __shared__ int s_lo, s_hi;
  for (int w = blockIdx.x; w < n_big[0]; w += gridDim.x) {
    const int p = big[w], b = pt_off[p], n = pt_off[p + 1] - b;
    if (threadIdx.x == 0) { s_lo = 0x7fffffff; s_hi = -1; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (int a = threadIdx.x; a < n; a += 256) {
      const int e = seg[b + a];
      int r = 0;
      for (int q = 0; q < n; ++q) r += seg[b + q] < e;
      const int c = cam_idx[e];
      p_src[b + r] = e; p_cam[b + r] = c; p_pt[b + r] = p;
      lo = min(lo, c); hi = max(hi, c);
    }
    atomicMin(&s_lo, lo); atomicMax(&s_hi, hi);
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(span_sum, (unsigned long long)(s_hi - s_lo)); atomicAdd(n_tracks, 1); }
    __syncthreads();
  }
}

// ---- long-track flags (scanned by the host code into the ascending list) -------------------------------------------
__global__ void __launch_bounds__(256)
k_setup_long_flags(const int* __restrict__ pt_off, int n_pts, int thr, int* __restrict__ flag) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n_pts) flag[p] = (pt_off[p + 1] - pt_off[p] > thr) ? 1 : 0;
}
__global__ void __launch_bounds__(256)
k_setup_long_list(const int* __restrict__ pt_off, int n_pts, int thr, const int* __restrict__ pos, int* __restrict__ long_pts) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n_pts && pt_off[p + 1] - pt_off[p] > thr) long_pts[pos[p]] = p;
}

// ---- bank-aware visiting order: the host's greedy rule (ba_set_problem, bank_aware_order), one thread per group of
//      eight points of a 16-point chunk of a point-pass range -------------------------------------------------------
__global__ void __launch_bounds__(256)
k_setup_bank_order(const int* __restrict__ pt_off, int n_pts, int nblk, int ppb, int* __restrict__ p_cam, int* __restrict__ p_src) {
  // thread -> (range b, chunk within the range, group g)
  const int chunks_per_range = (ppb + 15) / 16;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)nblk * chunks_per_range * 2) return;
  const int g = (int)(t & 1);
  const int chunk = (int)((t >> 1) % chunks_per_range), b = (int)((t >> 1) / chunks_per_range);
  const int p0 = min(n_pts, b * ppb), p1 = min(n_pts, (b + 1) * ppb);
  const int c0 = p0 + 16 * chunk;
  if (c0 >= p1) return;
  const int group_of_pair[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};
  int pts[8], npts = 0;
  for (int q = 0; q < 16 && c0 + q < p1; ++q)
    if (group_of_pair[q] == g) pts[npts++] = c0 + q;
  int maxlen = 0;
  for (int i = 0; i < npts; ++i) maxlen = max(maxlen, pt_off[pts[i] + 1] - pt_off[pts[i]]);
  if (maxlen > 64) return;
  int cur[8], end[8];
  for (int i = 0; i < npts; ++i) { cur[i] = pt_off[pts[i]]; end[i] = pt_off[pts[i] + 1]; }
  for (int step = 0; 2 * step < maxlen; ++step) {
    unsigned used = 0;
    for (int i = 0; i < npts; ++i) {
      for (int sub = 0; sub < 2 && cur[i] < end[i]; ++sub) {
        int pick = cur[i];
        for (int j = cur[i]; j < end[i]; ++j)
          if (!((used >> (p_cam[j] & 15)) & 1u)) { pick = j; break; }
        used |= 1u << (p_cam[pick] & 15);
        const int tc = p_cam[pick], ts = p_src[pick];
        p_cam[pick] = p_cam[cur[i]]; p_src[pick] = p_src[cur[i]];
        p_cam[cur[i]] = tc; p_src[cur[i]] = ts;
        ++cur[i];
      }
    }
  }
}

// ---- camera order ---------------------------------------------------------------------------------------------------
// A million atomics on a thousand counters are served one at a time per counter: both kernels first count in LDS over a
// tile of SETUP_CAM_TILE positions (one camera counter each in dynamic LDS, n_cams ints -- twice that for the scatter) and
// then touch every global counter once per tile.
constexpr int SETUP_CAM_ITEMS = 16;
constexpr int SETUP_CAM_TILE = 256 * SETUP_CAM_ITEMS;
__global__ void __launch_bounds__(256)
k_setup_hist_cam(const int* __restrict__ p_cam, int n_obs, int n_cams, int* __restrict__ cnt) {
  extern __shared__ int l_cnt[];
  for (int c = threadIdx.x; c < n_cams; c += 256) l_cnt[c] = 0;
  __syncthreads();
  const int j0 = blockIdx.x * SETUP_CAM_TILE + threadIdx.x;
#pragma unroll
  for (int q = 0; q < SETUP_CAM_ITEMS; ++q) {
    const int j = j0 + 256 * q;
    if (j < n_obs) atomicAdd(&l_cnt[p_cam[j]], 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_cams; c += 256) { const int v = l_cnt[c]; if (v) atomicAdd(cnt + c, v); }
}
// seg: every camera's segment holds its positions in ANY order (k_setup_sort_cam sorts them)
__global__ void __launch_bounds__(256)
k_setup_scatter_cam(const int* __restrict__ p_cam, int n_obs, int n_cams, const int* __restrict__ cam_off, int* __restrict__ fill,
                    int* __restrict__ seg) {
  extern __shared__ int l_cnt[];                 // [n_cams] counts of the tile, then [n_cams] the tile's first slot per camera
  int* l_base = l_cnt + n_cams;
  for (int c = threadIdx.x; c < n_cams; c += 256) l_cnt[c] = 0;
  __syncthreads();
  const int j0 = blockIdx.x * SETUP_CAM_TILE + threadIdx.x;
  int cam[SETUP_CAM_ITEMS], loc[SETUP_CAM_ITEMS];
#pragma unroll
  for (int q = 0; q < SETUP_CAM_ITEMS; ++q) {
    const int j = j0 + 256 * q;
    cam[q] = j < n_obs ? p_cam[j] : -1;
    loc[q] = cam[q] >= 0 ? atomicAdd(&l_cnt[cam[q]], 1) : 0;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_cams; c += 256) {
    const int v = l_cnt[c];
    l_base[c] = v ? cam_off[c] + atomicAdd(fill + c, v) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SETUP_CAM_ITEMS; ++q)
    if (cam[q] >= 0) seg[l_base[cam[q]] + loc[q]] = j0 + 256 * q;
}
// one workgroup per camera: its segment of positions sorted ascending (rank sort, keys unique), then
// c_pt = point of the position, c_orig = caller's row of the position
__global__ void __launch_bounds__(256)
k_setup_sort_cam(const int* __restrict__ cam_off, const int* __restrict__ seg, const int* __restrict__ p_pt, const int* __restrict__ p_src,
                 int* __restrict__ c_pt, int* __restrict__ c_orig) {
  __shared__ int keys[SETUP_RANK_LDS];
  const int c = blockIdx.x, b = cam_off[c], n = cam_off[c + 1] - b;
  const bool in_lds = n <= SETUP_RANK_LDS;
  if (in_lds) {
    for (int a = threadIdx.x; a < n; a += 256) keys[a] = seg[b + a];
    __syncthreads();
  }
  for (int a = threadIdx.x; a < n; a += 256) {
    const int e = in_lds ? keys[a] : seg[b + a];
    int r = 0;
    if (in_lds) { for (int q = 0; q < n; ++q) r += keys[q] < e; }
    else        { for (int q = 0; q < n; ++q) r += seg[b + q] < e; }
    c_pt[b + r] = p_pt[e];
    c_orig[b + r] = p_src[e];
  }
}
// offk[c][k] = cam_off[c] + (n k) / NPART
__global__ void __launch_bounds__(256)
k_setup_offk(const int* __restrict__ cam_off, int n_cams, int* __restrict__ offk) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_cams * (NPART + 1)) return;
  const int c = t / (NPART + 1), k = t % (NPART + 1);
  const long long n = cam_off[c + 1] - cam_off[c];
  offk[t] = cam_off[c] + (int)((n * k) / NPART);
}
// the two counts behind cam_band (group_of_block): observations whose point lies in the table slice of their partition /
// of their camera's range.  stat[0] += in_partition, stat[1] += in_band.
__global__ void __launch_bounds__(256)
k_setup_xcd_stat(const int* __restrict__ offk, const int* __restrict__ c_pt, int n_cams, int n_pts, unsigned long long* __restrict__ stat) {
  // one wave per (camera, partition) segment, grid-stride; one pair of atomics per WORKGROUP (two words for everybody)
  __shared__ long long s_a[4], s_b[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long long a = 0, bnd = 0;
  for (int seg = blockIdx.x * 4 + wv; seg < n_cams * NPART; seg += gridDim.x * 4) {
    const int c = seg / NPART, k = seg % NPART;
    const int cam_slice = (int)(((long long)c * NPART) / n_cams);
    for (int i = offk[c * (NPART + 1) + k] + lane; i < offk[c * (NPART + 1) + k + 1]; i += 64) {
      // slice of point q: the k with ceil(k Np / 8) <= q < ceil((k + 1) Np / 8)  <=>  k = floor(8 q / Np)  (checked below)
      const int q = c_pt[i];
      int s = (int)(((long long)q * NPART) / n_pts);
      // exact form of the host's table: slice k covers [ceil(k Np / NPART), ceil((k + 1) Np / NPART))
      while (s > 0 && q < (int)(((long long)s * n_pts + NPART - 1) / NPART)) --s;
      while (s < NPART - 1 && q >= (int)(((long long)(s + 1) * n_pts + NPART - 1) / NPART)) ++s;
      a += s == k; bnd += s == cam_slice;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); bnd += __shfl_down(bnd, o, 64); }
  if (lane == 0) { s_a[wv] = a; s_b[wv] = bnd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long ta = s_a[0] + s_a[1] + s_a[2] + s_a[3], tb = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    if (ta | tb) { atomicAdd(stat, (unsigned long long)ta); atomicAdd(stat + 1, (unsigned long long)tb); }
  }
}
// camera window [lo, hi] of every point-pass workgroup: ranges of ppb points, then long-track workgroups of long_spb entries
__global__ void __launch_bounds__(256)
k_setup_windows(const int* __restrict__ pt_off, const int* __restrict__ p_cam, int n_pts, int n_cams, int nblkP, int ppb,
                const int* __restrict__ long_pts, int n_long, int long_spb, int2* __restrict__ win) {
  __shared__ int s_lo, s_hi;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) { s_lo = n_cams; s_hi = -1; }
  __syncthreads();
  int lo = n_cams, hi = -1;
  if (b < nblkP) {
    const int p0 = min(n_pts, b * ppb), p1 = min(n_pts, (b + 1) * ppb);
    for (int j = pt_off[p0] + threadIdx.x; j < pt_off[p1]; j += 256) { const int c = p_cam[j]; lo = min(lo, c); hi = max(hi, c); }
  } else {
    const int w = b - nblkP;
    for (int q = w * long_spb; q < min(n_long, (w + 1) * long_spb); ++q) {
      const int p = long_pts[q];
      for (int j = pt_off[p] + threadIdx.x; j < pt_off[p + 1]; j += 256) { const int c = p_cam[j]; lo = min(lo, c); hi = max(hi, c); }
    }
  }
  atomicMin(&s_lo, lo); atomicMax(&s_hi, hi);
  __syncthreads();
  if (threadIdx.x == 0) {
    int l = s_lo, h2 = s_hi;
    if (h2 < l) { l = 0; h2 = -1; }
    win[b] = make_int2(l, h2 - l + 1);
  }
}
// ---- window-sized problems (host build): ONE upload, ONE kernel --------------------------------------------------------
// The host build of a sliding window (the reference's own use, src/pipeline.py:99: a few thousand observations) ships a
// dozen small arrays; as separate copies each is a DMA packet of its own and the three kernels behind them three more
// launches -- more device time than the window's solve.  They travel as ONE pinned arena and ONE copy instead, and this
// kernel deals the sections to their arrays and does what k_gather_uv (x 2) and k_init_flagged did, reading the arena.
constexpr int UNPACK_MAX_SECTIONS = 12;
struct UnpackArgs {
  const char* arena;                       // device copy of the pinned arena
  int n_sections;
  size_t off[UNPACK_MAX_SECTIONS];         // byte offset of a section in the arena (64-byte aligned)
  int* dst[UNPACK_MAX_SECTIONS];           // its array
  int words[UNPACK_MAX_SECTIONS];          // its length in 4-byte words
  // pixels and flagged index copies (n_obs > 0): offsets of uv (double2[n_obs], caller's order), p_src, c_orig, c_pt, p_cam
  int n_obs;
  size_t off_uv, off_psrc, off_corig, off_cpt, off_pcam;
  double2 *p_uv, *c_uv;
  int uv_f32;                              // the pixel streams are stored as float2 (UvArr, ba_kernels.hpp)
  int *c_ptf0, *c_ptf1, *p_camf0, *p_camf1;
};
__global__ void __launch_bounds__(256)
k_unpack_problem(UnpackArgs a) {
  const int stride = gridDim.x * 256;
  const int t0 = blockIdx.x * 256 + threadIdx.x;
  for (int sct = 0; sct < a.n_sections; ++sct) {
    const int* src = (const int*)(a.arena + a.off[sct]);
    for (int i = t0; i < a.words[sct]; i += stride) a.dst[sct][i] = src[i];
  }
  if (a.n_obs > 0) {
    const double2* uv = (const double2*)(a.arena + a.off_uv);
    const int* psrc = (const int*)(a.arena + a.off_psrc);
    const int* corig = (const int*)(a.arena + a.off_corig);
    const int* cpt = (const int*)(a.arena + a.off_cpt);
    const int* pcam = (const int*)(a.arena + a.off_pcam);
    for (int j = t0; j < a.n_obs; j += stride) {
      const double2 up = uv[psrc[j]], uc = uv[corig[j]];
      if (a.uv_f32) {
        ((float2*)a.p_uv)[j] = make_float2((float)up.x, (float)up.y);
        ((float2*)a.c_uv)[j] = make_float2((float)uc.x, (float)uc.y);
      } else {
        a.p_uv[j] = up;
        a.c_uv[j] = uc;
      }
      const int x = cpt[j], y = pcam[j];
      a.c_ptf0[j] = x; a.c_ptf1[j] = x; a.p_camf0[j] = y; a.p_camf1[j] = y;
    }
  }
}

// slot[p] = p (the caller's numbering is kept when the whole camera table fits in LDS)
__global__ void __launch_bounds__(256)
k_setup_iota(int* __restrict__ a, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = i;
}

}  // namespace ba

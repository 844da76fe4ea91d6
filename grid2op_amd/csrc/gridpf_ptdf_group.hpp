// gridpf_ptdf_group.hpp -- the integer half of gpf_ptdf_build_batch ON THE DEVICE: which lanes hold the same topology, and the class
// descriptor of every distinct topology (live buses, compact numbering, connectivity verdict, line-end tables, rows of the reduced B').
//
// Round 5 did this on the host: the lanes' topology rows crossed PCIe (4.6 MB for 2 048 lanes of a 118-substation grid), were hashed and
// compared on one core, and every unseen topology cost a table walk -- 0.8 ... 1.7 ms around a 98 us kernel.  The rows already live in the
// engine's HBM; four small kernels replace the host pass (and the descriptor upload):
//   ptdfg_hash_kernel    one wavefront per lane: 64-bit hash of its (topology row, shunt buses)
//   ptdfg_group_kernel   ONE workgroup: lanes sorted by (hash, lane) in LDS (bitonic), equal hashes = one class; class ids in hash order,
//                        the class's first lane, the slot order of the flows kernels (lanes grouped by class, groups padded to 16)
//   ptdfg_verify_kernel  one wavefront per lane: its row against its class representative's (a 64-bit collision would merge two
//                        topologies: flagged, and the host path takes over)
//   ptdfg_desc_kernel    one workgroup per class: the descriptor, exactly what the host's build_class writes (gridpf_capi.hip)
// The host reads back six integers (classes, slots, largest padded dimension / active-bus count, the two flags) to size the launch of
// the factorisation kernel; lane -> class map, descriptors and compact -> bus maps stay on the device until somebody asks for them.
//
// Reference stage: pp.rundcpp rebuilds all of this from the pandapower net on every call (grid2op/Backend/pandaPowerBackend.py:1090).
#pragma once
#include "gridpf_ptdf_batch.hpp"

namespace gpf {

constexpr int PTDFG_MAX_LANES = 4096;      // one-workgroup sort: 48 KB of LDS for keys + lane ids
constexpr int PTDFG_SORT_THREADS = 1024;
constexpr int PTDFG_MAX_BUS = 512;         // buses (n_sub * n_busbar) the descriptor kernel's LDS tables hold
constexpr int PTDFG_DESC_THREADS = 256;

struct PtdfGroupDev {
  // inputs
  const int* topo;            // [n_lanes][dim_topo] (engine rows)
  const int* shunt_bus;       // [n_lanes][n_shunt]
  int lane0, n, dim_topo, n_shunt, n_sub, n_busbar, n_line, n_gen, n_load, n_sto, n_inj;
  int inj_gen_p, inj_load_p, inj_sto_p, inj_sh_p;      // OutOff columns of the injection row
  const int *line_or_pos, *line_ex_pos, *line_or_sub, *line_ex_sub, *gen_pos, *gen_sub, *load_pos, *load_sub, *sto_pos, *sto_sub, *shunt_sub;
  const unsigned char* gen_slack;
  int desc_stride;
  // outputs
  unsigned long long* hash;   // [n]
  int* lane_class;            // [n]
  int* first_lane;            // [n] representative lane (index in the range) of each class
  int* order;                 // [16 n] slot -> lane (absolute), -1 padding
  int* blk_class;             // [n] class of every block of 16 slots
  int* desc;                  // [n][desc_stride]
  int* c2b;                   // [n][nb_tot] compact bus -> bus id
  int* info;                  // [8]: 0 classes, 1 slots, 2 largest n_pad, 3 largest n_act, 4 row mismatch inside a class, 5 descriptor error (1 bus id, 2 capacity)
};

__device__ __forceinline__ unsigned long long ptdfg_mix(unsigned long long x) {      // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}

// one wavefront per lane: sum over the positions of mix(position, value) -- order-free, so the lanes of the wavefront add up in any order
__global__ __launch_bounds__(64) void ptdfg_hash_kernel(PtdfGroupDev D) {
  const int k = blockIdx.x, l = threadIdx.x;
  if (k == 0 && l < 8) D.info[l] = 0;                     // (the kernels behind this one on the stream fill it)
  if (k >= D.n) return;
  const int* tp = D.topo + (size_t)(D.lane0 + k) * D.dim_topo;
  unsigned long long h = 0;
  for (int i = l; i < D.dim_topo; i += 64) h += ptdfg_mix(((unsigned long long)(unsigned)i << 32) ^ (unsigned)tp[i] ^ 0x9E3779B97F4A7C15ull);
  if (D.n_shunt) {
    const int* sp = D.shunt_bus + (size_t)(D.lane0 + k) * D.n_shunt;
    for (int i = l; i < D.n_shunt; i += 64) h += ptdfg_mix(((unsigned long long)(unsigned)(i + 0x40000000) << 32) ^ (unsigned)sp[i] ^ 0xC2B2AE3D27D4EB4Full);
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) h += __shfl_xor(h, off);
  if (l == 0) D.hash[k] = ptdfg_mix(h);
}

// exclusive scan of v[0 .. n) in LDS (in place), n <= cap, by all threads of the block; tmp: a second array of the same size
__device__ inline void ptdfg_scan_excl(int* v, int* tmp, int n, int tid, int nt) {
  // Hillis-Steele inclusive scan, then shift
  int* a = v;
  int* b = tmp;
  for (int d = 1; d < n; d <<= 1) {
    for (int i = tid; i < n; i += nt) b[i] = a[i] + (i >= d ? a[i - d] : 0);
    __syncthreads();
    int* t = a; a = b; b = t;
  }
  // a holds the inclusive scan
  for (int i = tid; i < n; i += nt) b[i] = i ? a[i - 1] : 0;
  __syncthreads();
  if (b != v) { for (int i = tid; i < n; i += nt) v[i] = b[i]; __syncthreads(); }
}

// Bitonic sort of NV = PTDFG_SORT_THREADS * E (key, index) pairs, ascending, by one workgroup: thread t holds the E consecutive elements
// t E .. t E + E - 1 in registers.  An exchange at distance j < E stays inside the thread, j < 64 E goes through wavefront shuffles, the rest
// (10 of the 66 steps of 2 048 elements) through LDS -- the all-LDS version with a barrier per step took 48 us of the 0.25 ms of a
// gpf_ptdf_build_batch call.  Leaves the sorted pairs in s_key / s_idx.
template <int E>
__device__ inline void ptdfg_sort_regs(const unsigned long long* __restrict__ hash, int n, unsigned long long* s_key, int* s_idx, int tid) {
  constexpr int NV = PTDFG_SORT_THREADS * E;
  unsigned long long key[E];
  int idx[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { const int i = tid * E + e; key[e] = i < n ? hash[i] : ~0ull; idx[e] = i < n ? i : 0x40000000 + i; }
  auto less = [](unsigned long long ka, int ia, unsigned long long kb, int ib) { return ka < kb || (ka == kb && ia < ib); };
  for (int k = 2; k <= NV; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {                                             // partner in this thread (register indices are literals after unrolling)
        const bool up = ((tid * E) & k) == 0;                  // (k > j: the same for the thread's E elements when k >= E; k < E: per pair below)
#pragma unroll
        for (int jj = 1; jj < E; jj <<= 1) {
          if (j != jj) continue;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            if ((e & jj) || (e | jj) >= E) continue;
            const int f = e | jj;
            const bool up_e = k >= E ? up : (e & k) == 0;
            if (less(key[f], idx[f], key[e], idx[e]) == up_e) { const unsigned long long tk = key[e]; key[e] = key[f]; key[f] = tk; const int ti = idx[e]; idx[e] = idx[f]; idx[f] = ti; }
          }
        }
      } else if (j < 64 * E) {                                 // partner in this wavefront
        const int m = j / E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = tid * E + e;
          const unsigned long long kb = __shfl_xor(key[e], m);
          const int ib = __shfl_xor(idx[e], m);
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          if (less(kb, ib, key[e], idx[e]) == keep_min) { key[e] = kb; idx[e] = ib; }
        }
      } else {                                                 // through LDS
#pragma unroll
        for (int e = 0; e < E; ++e) { s_key[tid * E + e] = key[e]; s_idx[tid * E + e] = idx[e]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = tid * E + e, p = i ^ j;
          const unsigned long long kb = s_key[p];
          const int ib = s_idx[p];
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          if (less(kb, ib, key[e], idx[e]) == keep_min) { key[e] = kb; idx[e] = ib; }
        }
        __syncthreads();
      }
    }
#pragma unroll
  for (int e = 0; e < E; ++e) { s_key[tid * E + e] = key[e]; s_idx[tid * E + e] = idx[e]; }
  __syncthreads();
}

__global__ __launch_bounds__(PTDFG_SORT_THREADS) void ptdfg_group_kernel(PtdfGroupDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ptdfg_smem[];        // 20 bytes per element of the padded range (host: lds_sort)
  __shared__ int s_nc;
  const int tid = threadIdx.x, nt = blockDim.x, n = D.n;
  int np2 = PTDFG_SORT_THREADS;                           // (the sort works on a multiple of the workgroup: host sizes the LDS alike)
  while (np2 < n) np2 <<= 1;
  unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(ptdfg_smem);
  int* const s_idx = reinterpret_cast<int*>(s_key + np2);
  int* const s_a = s_idx + np2;
  int* const s_b = s_a + np2;
  // sorted by (key, lane) ascending: equal hashes are adjacent, lanes ascending inside a run
  if (np2 == PTDFG_SORT_THREADS) ptdfg_sort_regs<1>(D.hash, n, s_key, s_idx, tid);
  else if (np2 == 2 * PTDFG_SORT_THREADS) ptdfg_sort_regs<2>(D.hash, n, s_key, s_idx, tid);
  else ptdfg_sort_regs<4>(D.hash, n, s_key, s_idx, tid);
  // heads of the runs of equal keys -> class index of every sorted position
  for (int i = tid; i < n; i += nt) s_a[i] = (i == 0 || s_key[i] != s_key[i - 1]) ? 1 : 0;
  __syncthreads();
  if (tid == 0) s_nc = 0;
  ptdfg_scan_excl(s_a, s_b, n, tid, nt);                 // s_a[i] = heads before position i
  // class of position i = s_a[i] + head(i) - 1
  for (int i = tid; i < n; i += nt) {
    const bool head = (i == 0 || s_key[i] != s_key[i - 1]);
    const int c = s_a[i] + (head ? 1 : 0) - 1;
    s_b[i] = c;                                          // class of sorted position i
    if (head) D.first_lane[c] = s_idx[i];                // (sorted by lane inside a class: the smallest lane)
    if (i == n - 1) s_nc = c + 1;
  }
  __syncthreads();
  const int nc = s_nc;
  for (int i = tid; i < n; i += nt) D.lane_class[s_idx[i]] = s_b[i];
  // sizes of the classes: s_a[c] = position of the head of class c (then sizes by difference)
  __syncthreads();
  for (int i = tid; i < n; i += nt) if (i == 0 || s_key[i] != s_key[i - 1]) s_a[s_b[i]] = i;
  __syncthreads();
  // padded sizes -> slot offsets
  int* s_pad = reinterpret_cast<int*>(s_key);            // (the keys are no longer needed: heads are in s_a / s_b)  [np2] ints
  int* s_tmp = s_pad + np2;                              // second half of the key array
  // keep head positions in registers-free form: copy to s_tmp2 = s_idx is still needed; use s_pad for padded sizes
  for (int c = tid; c < nc; c += nt) { const int sz = (c + 1 < nc ? s_a[c + 1] : n) - s_a[c]; s_pad[c] = (sz + 15) & ~15; }
  __syncthreads();
  ptdfg_scan_excl(s_pad, s_tmp, nc, tid, nt);            // s_pad[c] = first slot of class c
  // total slots
  if (tid == 0) {
    const int last_sz = n - s_a[nc - 1];
    D.info[0] = nc;
    D.info[1] = s_pad[nc - 1] + ((last_sz + 15) & ~15);
  }
  // slot order: position i of class c, rank r = i - head(c)
  for (int i = tid; i < n; i += nt) { const int c = s_b[i]; D.order[s_pad[c] + (i - s_a[c])] = D.lane0 + s_idx[i]; }
  for (int c = tid; c < nc; c += nt) {
    const int sz = (c + 1 < nc ? s_a[c + 1] : n) - s_a[c], psz = (sz + 15) & ~15;
    for (int r = sz; r < psz; ++r) D.order[s_pad[c] + r] = -1;
    for (int q = 0; q < psz / 16; ++q) D.blk_class[s_pad[c] / 16 + q] = c;
  }
}

// one wavefront per lane: the lane's row against its class representative's
__global__ __launch_bounds__(64) void ptdfg_verify_kernel(PtdfGroupDev D) {
  const int k = blockIdx.x, l = threadIdx.x;
  if (k >= D.n) return;
  const int rep = D.first_lane[D.lane_class[k]];
  if (rep == k) return;
  const int* a = D.topo + (size_t)(D.lane0 + k) * D.dim_topo;
  const int* b = D.topo + (size_t)(D.lane0 + rep) * D.dim_topo;
  bool diff = false;
  for (int i = l; i < D.dim_topo; i += 64) diff |= a[i] != b[i];
  if (D.n_shunt) {
    const int* sa = D.shunt_bus + (size_t)(D.lane0 + k) * D.n_shunt;
    const int* sb = D.shunt_bus + (size_t)(D.lane0 + rep) * D.n_shunt;
    for (int i = l; i < D.n_shunt; i += 64) diff |= sa[i] != sb[i];
  }
  if (__any(diff) && l == 0) atomicOr(&D.info[4], 1);
}

// block-wide exclusive scan of one flag per thread (blockDim = PTDFG_DESC_THREADS = 4 wavefronts); returns the thread's offset, *total = sum
__device__ inline int ptdfg_flag_scan(bool f, int tid, int* s_w, int* total) {
  const unsigned long long b = __ballot(f);
  const int l = tid & 63, w = tid >> 6;
  const int before = __popcll(b & ((1ull << l) - 1ull));
  if (l == 0) s_w[w] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < PTDFG_DESC_THREADS / 64; ++q) { if (q < w) off += s_w[q]; tot += s_w[q]; }
  __syncthreads();
  *total = tot;
  return off + before;
}

// one workgroup per class: the descriptor of the class representative's topology (the device twin of gpf_ptdf_build_batch's build_class)
__global__ __launch_bounds__(PTDFG_DESC_THREADS) void ptdfg_desc_kernel(PtdfGroupDev D) {
  __shared__ int s_act[PTDFG_MAX_BUS], s_ref[PTDFG_MAX_BUS], s_nl[PTDFG_MAX_BUS], s_no[PTDFG_MAX_BUS], s_cmp[PTDFG_MAX_BUS], s_lab[PTDFG_MAX_BUS];
  __shared__ short s_bf[256], s_bt[256];
  __shared__ __attribute__((aligned(16))) int s_ab[256];           // compact ends of line l: from | to << 16 (0xffff: out of service), read four at a time
  __shared__ int s_w[PTDFG_DESC_THREADS / 64], s_flag;
  const int c = blockIdx.x, tid = threadIdx.x, nt = PTDFG_DESC_THREADS;
  if (c >= D.info[0]) return;
  const int nbt = D.n_sub * D.n_busbar, nl = D.n_line;
  const int lane = D.lane0 + D.first_lane[c];
  const int* tp = D.topo + (size_t)lane * D.dim_topo;
  const int* sbp = D.shunt_bus + (size_t)lane * (D.n_shunt > 0 ? D.n_shunt : 0);
  int* d = D.desc + (size_t)c * D.desc_stride;
  int* lf = d + PTDFB_HDR;
  int* lt = lf + nl;
  int* ib = lt + nl;
  int* lflag = ib + D.n_inj;
  int* cptr = lflag + nl;
  int* cent = cptr + PTDFB_MAX_N + 1;
  int* c2b = D.c2b + (size_t)c * nbt;
  auto bus_of = [&](int sub, int local) -> int { return (local >= 1 && local <= D.n_busbar) ? sub + (local - 1) * D.n_sub : -1; };
  for (int b = tid; b < nbt; b += nt) { s_act[b] = 0; s_ref[b] = 0; s_nl[b] = 0; s_no[b] = 0; s_cmp[b] = -1; }
  for (int i = tid; i < D.n_inj; i += nt) ib[i] = -1;                 // (bus ids first, compacted at the end)
  if (tid == 0) s_flag = 0;
  __syncthreads();
  bool bad = false;
  for (int l = tid; l < nl; l += nt) {
    const int bo = tp[D.line_or_pos[l]], be = tp[D.line_ex_pos[l]];
    int f = -1, t = -1;
    if (bo >= 1 && be >= 1) {
      f = bus_of(D.line_or_sub[l], bo); t = bus_of(D.line_ex_sub[l], be);
      if (f < 0 || t < 0) { bad = true; f = t = -1; }
      else { s_act[f] = 1; s_act[t] = 1; atomicAdd(&s_nl[f], 1); atomicAdd(&s_nl[t], 1); }
    }
    s_bf[l] = (short)f; s_bt[l] = (short)t;
  }
  for (int i = tid; i < D.n_gen; i += nt) {
    const int b = bus_of(D.gen_sub[i], tp[D.gen_pos[i]]);
    if (b < 0) continue;
    s_act[b] = 1; atomicAdd(&s_no[b], 1);
    if (D.gen_slack[i]) s_ref[b] = 1; else ib[D.inj_gen_p + i] = b;
  }
  for (int i = tid; i < D.n_load; i += nt) { const int b = bus_of(D.load_sub[i], tp[D.load_pos[i]]); if (b >= 0) { s_act[b] = 1; atomicAdd(&s_no[b], 1); ib[D.inj_load_p + i] = b; } }
  for (int i = tid; i < D.n_sto; i += nt) { const int b = bus_of(D.sto_sub[i], tp[D.sto_pos[i]]); if (b >= 0) { s_act[b] = 1; atomicAdd(&s_no[b], 1); ib[D.inj_sto_p + i] = b; } }
  for (int i = tid; i < D.n_shunt; i += nt) { const int b = bus_of(D.shunt_sub[i], sbp[i]); if (b >= 0) { s_act[b] = 1; atomicAdd(&s_no[b], 1); ib[D.inj_sh_p + i] = b; } }
  if (bad) s_flag = 1;
  __syncthreads();
  if (s_flag) { if (tid == 0) atomicMax(&D.info[5], 1); return; }
  // compact numbering: active non-reference buses first (bus order), then the active reference buses
  int nr = 0, n_act = 0;
  for (int b0 = 0; b0 < nbt; b0 += nt) {
    const int b = b0 + tid;
    const bool f = b < nbt && s_act[b] && !s_ref[b];
    int tot;
    const int o = ptdfg_flag_scan(f, tid, s_w, &tot);
    if (f) { s_cmp[b] = nr + o; c2b[nr + o] = b; }
    nr += tot;
  }
  n_act = nr;
  for (int b0 = 0; b0 < nbt; b0 += nt) {
    const int b = b0 + tid;
    const bool f = b < nbt && s_act[b] && s_ref[b];
    int tot;
    const int o = ptdfg_flag_scan(f, tid, s_w, &tot);
    if (f) { s_cmp[b] = n_act + o; c2b[n_act + o] = b; }
    n_act += tot;
  }
  for (int q = n_act + tid; q < nbt; q += nt) c2b[q] = -1;
  const bool any_ref = n_act > nr;
  // connectivity: every active bus must reach a reference bus (label propagation over the in-service lines)
  int status = any_ref ? 0 : 3;
  if (any_ref) {
    for (int b = tid; b < nbt; b += nt) s_lab[b] = (s_act[b] && s_ref[b]) ? 1 : 0;
    __syncthreads();
    for (int sweep = 0; sweep < nbt; ++sweep) {
      int changed = 0;
      for (int l = tid; l < nl; l += nt) {
        const int f = s_bf[l], t = s_bt[l];
        if (f >= 0 && f != t && s_lab[f] != s_lab[t]) { s_lab[f] = 1; s_lab[t] = 1; changed = 1; }
      }
      if (!__syncthreads_or(changed)) break;
    }
    int isl = 0;
    for (int b = tid; b < nbt; b += nt) isl |= (s_act[b] && !s_lab[b]);
    if (__syncthreads_or(isl)) status = 2;
  }
  const int n_pad = nr + 15 < 16 ? 16 : ((nr + 15) & ~15);
  if (n_pad > PTDFB_MAX_N) { if (tid == 0) atomicMax(&D.info[5], 2); return; }
  if (tid == 0) { d[0] = nr; d[1] = n_act; d[2] = n_pad; d[3] = status; atomicMax(&D.info[2], n_pad); atomicMax(&D.info[3], n_act); }
  for (int l = tid; l < nl; l += nt) {
    const int f = s_bf[l], t = s_bt[l];
    const bool on = f >= 0 && f != t;
    lf[l] = on ? s_cmp[f] : -1; lt[l] = on ? s_cmp[t] : -1;
    lflag[l] = (on && ((s_nl[f] == 1 && s_no[f] == 0) || (s_nl[t] == 1 && s_no[t] == 0))) ? 1 : 0;
  }
  for (int i = tid; i < D.n_inj; i += nt) { const int b = ib[i]; ib[i] = b >= 0 ? s_cmp[b] : -1; }
  // (the compact line ends once more in LDS -- s_bf / s_bt are dead --: the row walks below read them nl times per thread)
  __syncthreads();
  for (int l = tid; l < nl; l += nt) {
    const int f = s_bf[l], t = s_bt[l];
    const bool on = f >= 0 && f != t;
    const short cf = on ? (short)s_cmp[f] : (short)-1, ct = on ? (short)s_cmp[t] : (short)-1;
    s_bf[l] = cf; s_bt[l] = ct;
    s_ab[l] = (int)((unsigned)(unsigned short)cf | ((unsigned)(unsigned short)ct << 16));
  }
  for (int l = nl + tid; l < ((nl + 3) & ~3); l += nt) s_ab[l] = -1;
  __syncthreads();
  // rows of the reduced B': for every non-reference bus r its lines in ascending order, as line | other end << 16.  Row lengths first
  // (thread r walks all lines: no atomics, ascending by construction), then an exclusive scan for the row pointers
  int* s_cnt = s_lab;                                                // (labels are dead)
  for (int r = tid; r < PTDFB_MAX_N + 1; r += nt) s_cnt[r] = 0;
  __syncthreads();
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  for (int r = tid; r < nr; r += nt) {                               // (r < n_pad <= 128 never equals the 0xffff of a line out of service)
    int cnt = 0;
    for (int l0 = 0; l0 < nl; l0 += 4) {
      const v4i_ v = *reinterpret_cast<const v4i_*>(&s_ab[l0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const unsigned w = (unsigned)v[q]; cnt += ((int)(w & 0xffffu) == r || (int)(w >> 16) == r) ? 1 : 0; }
    }
    s_cnt[r] = cnt;
  }
  __syncthreads();
  {
    int* tmp = s_nl;                                                 // (line-end counts are dead after lflag)
    ptdfg_scan_excl(s_cnt, tmp, PTDFB_MAX_N + 1 <= PTDFG_MAX_BUS ? PTDFB_MAX_N + 1 : PTDFG_MAX_BUS, tid, nt);
  }
  // s_cnt[r] = first entry of row r for r < nr; rows >= nr are empty: their pointer = total = s_cnt[nr]
  const int total = s_cnt[nr];
  for (int r = tid; r <= PTDFB_MAX_N; r += nt) cptr[r] = r < nr ? s_cnt[r] : total;
  for (int i = tid; i < 2 * nl; i += nt) cent[i] = 0;
  __syncthreads();
  for (int r = tid; r < nr; r += nt) {
    int q = s_cnt[r];
    for (int l0 = 0; l0 < nl; l0 += 4) {
      const v4i_ v = *reinterpret_cast<const v4i_*>(&s_ab[l0]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned w = (unsigned)v[k];
        const int a = (int)(w & 0xffffu), b = (int)(w >> 16);
        if (a == r) cent[q++] = (l0 + k) | (b << 16);
        else if (b == r) cent[q++] = (l0 + k) | (a << 16);
      }
    }
  }
}

}  // namespace gpf

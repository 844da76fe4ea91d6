// gridpf_sparse.hpp -- kernel S: block-sparse Newton-Raphson power flow + batched environment step, every grid size.
//
// One wavefront per grid instance, everything in LDS, but the linear algebra is a right-looking BLOCK-SPARSE LU on the
// substation graph (host-side symbolic analysis: gridpf_symbolic.hpp).  A block couples the NB busbars x {theta, |V|}
// of two substations (BS = 2*NB); rows / columns of inactive buses, reference buses and PV voltage magnitudes are
// identity, so ONE static pattern serves every topology (bus splits, outages) of the grid.  NB = 1 is used when no
// substation of the batch has more than one live busbar (the common case), NB = n_busbar otherwise.
//
// Work per factorisation is O(sum_k deg_k^2 * BS^3) (~10^4 FMA for 118 substations) instead of 2/3 n^3 = 4.6e6 for
// the dense Jacobian, and the 0.3 MB dense matrix (which cannot live in the 160 KB LDS) shrinks to a ~25 KB block
// array.  Assembly uses native LDS f64 atomics from branch / element lanes; the pipeline (K9, K1..K7) and its reference
// counterparts are listed in gridpf_common.hpp.
#pragma once
#include <type_traits>
#include "gridpf_common.hpp"

namespace gpf {

// Static (per grid) tables of kernel S live in ONE blob (doubles, then ints) so that a block can stage all of them in LDS
// with one coalesced copy: every per-element loop of the kernel then starts from LDS reads instead of a chain of dependent
// L2 / HBM loads.  Offsets are element offsets into the double / int section; the host builds the blob (gridpf_capi.hip).
struct StatOff {
  int n_dbl, n_int;
  int n_int_hot;   // the int section starts with the tables of the Newton loop (prog, pair_rc): staging tier 1 copies only these
  int br_y, br_bdc, sub_vn_kv, shunt_fact, gen_min_q, gen_max_q;
  int line_vn, load_vn, gen_vn, sto_vn, shunt_vn;   // nominal kV of the substation of every element (line: [n_line][2] = or, ex)
  int line_ka;     // [n_line][2] 1000 / (sqrt(3) vn) of the two ends: i [A] = |S| [MVA] / |V| [pu] * line_ka
  int gen_qmin_tot, gen_qmax_tot;   // [n_gen] sum of min_q / max_q over the generators of the generator's substation (every generator
                                    // connected, single busbar): what pfsoln's reactive split needs, per generator
  int gen_cnt;     // (int section) [n_gen] generators | slack generators << 16 of the generator's substation, same assumption
  int dc_inv;      // >= 0: [n_sub][n_sub] COLUMN-major inverse of the DC matrix B' of the reference topology (every line in service,
                   // every slack generator connected; reference / fixed rows are identity) inside the blob (small grids: staged in LDS
                   // with it); -2: the same table in global memory, SymDev::dc_inv_g (larger grids: read through L2); -1: none
  int line_or_pos, line_ex_pos, line_or_sub, line_ex_sub, br_slot, gen_pos, gen_sub, gen_slack, load_pos, load_sub, sto_pos,
      sto_sub, shunt_sub, pair_rc, up, prog;
  int pos_line;    // (int section) [dim_topo] line whose end sits at that topo_vect position, -1: the position of a generator / load / storage unit
};
#define GPF_STATOFF_INTS(X) X(n_dbl) X(n_int) X(n_int_hot) X(br_y) X(br_bdc) X(sub_vn_kv) X(shunt_fact) X(gen_min_q) X(gen_max_q) X(line_vn) X(load_vn) \
  X(gen_vn) X(sto_vn) X(shunt_vn) X(line_ka) X(gen_qmin_tot) X(gen_qmax_tot) X(gen_cnt) X(dc_inv) X(line_or_pos) X(line_ex_pos) X(line_or_sub) \
  X(line_ex_sub) X(br_slot) X(gen_pos) X(gen_sub) X(gen_slack) X(load_pos) X(load_sub) X(sto_pos) X(sto_sub) X(shunt_sub) X(pair_rc) X(up) X(prog) X(pos_line)
static_assert(sizeof(StatOff) == sizeof(int) * (0 GPF_STATOFF_INTS(GPF_COUNT_FIELD)), "GPF_STATOFF_INTS must list every field of StatOff");
// Pointer to a static table that is either staged in LDS or read in place: in place it is re-typed as a GLOBAL pointer
// (gptr, gridpf_common.hpp) so that global_load is emitted instead of flat_load.
template <class T, bool IN_LDS>
struct SP {
  const T* p;
  __device__ __forceinline__ T operator[](int i) const { return IN_LDS ? p[i] : (T)gptr(p)[i]; }
  __device__ __forceinline__ double4 ld4(size_t off) const {       // 4 consecutive doubles (32-byte aligned)
    if (IN_LDS) return *reinterpret_cast<const double4*>(p + off);
    typedef double v4d_ __attribute__((ext_vector_type(4)));
    const v4d_ v = *(GPF_GLOBAL const v4d_*)(gptr(p) + off);
    return make_double4(v.x, v.y, v.z, v.w);
  }
};
template <int STAGE>
struct StatView {
  static constexpr bool ALL = STAGE == 2, HOT = STAGE >= 1;
  SP<double, ALL> br_y, br_bdc, sub_vn_kv, shunt_fact, gen_min_q, gen_max_q, line_vn, line_ka, load_vn, gen_vn, sto_vn, shunt_vn, dc_inv, gen_qmin_tot, gen_qmax_tot;
  SP<int, ALL> line_or_pos, line_ex_pos, line_or_sub, line_ex_sub, br_slot, gen_pos, gen_sub, gen_slack, load_pos, load_sub,
      sto_pos, sto_sub, shunt_sub, gen_cnt, pos_line;
  SP<int, HOT> pair_rc;   // [nslot_y] slot_row | slot_col << 16 of the original-pattern blocks
  SP<int, HOT> up;        // [n_up][2] undirected off-diagonal pairs (gridpf_symbolic.hpp: build_upairs), single-busbar Newton loop
  SP<int, HOT> prog;      // level-scheduled program (layout: gridpf_symbolic.hpp)
  SP<int, false> node_of; // topology-class launches only (TopoClassDev::node_of)
  SP<double, false> dc_inv_g;   // StatOff::dc_inv == -2
};
template <int STAGE>
__device__ inline void stat_view(StatView<STAGE>& v, const StatOff& o, const double* d, const int* i) {
  v.br_y.p = d + o.br_y; v.br_bdc.p = d + o.br_bdc; v.sub_vn_kv.p = d + o.sub_vn_kv; v.shunt_fact.p = d + o.shunt_fact;
  v.gen_min_q.p = d + o.gen_min_q; v.gen_max_q.p = d + o.gen_max_q;
  v.dc_inv.p = d + (o.dc_inv >= 0 ? o.dc_inv : 0);
  v.gen_qmin_tot.p = d + o.gen_qmin_tot; v.gen_qmax_tot.p = d + o.gen_qmax_tot; v.gen_cnt.p = i + o.gen_cnt;
  v.line_vn.p = d + o.line_vn; v.line_ka.p = d + o.line_ka; v.load_vn.p = d + o.load_vn; v.gen_vn.p = d + o.gen_vn; v.sto_vn.p = d + o.sto_vn; v.shunt_vn.p = d + o.shunt_vn;
  v.line_or_pos.p = i + o.line_or_pos; v.line_ex_pos.p = i + o.line_ex_pos; v.line_or_sub.p = i + o.line_or_sub;
  v.line_ex_sub.p = i + o.line_ex_sub; v.br_slot.p = i + o.br_slot; v.gen_pos.p = i + o.gen_pos; v.gen_sub.p = i + o.gen_sub;
  v.gen_slack.p = i + o.gen_slack; v.load_pos.p = i + o.load_pos; v.load_sub.p = i + o.load_sub; v.sto_pos.p = i + o.sto_pos;
  v.sto_sub.p = i + o.sto_sub; v.shunt_sub.p = i + o.shunt_sub; v.pos_line.p = i + o.pos_line; v.pair_rc.p = i + o.pair_rc; v.up.p = i + o.up; v.prog.p = i + o.prog;
  v.node_of.p = nullptr;
}
// LDS bytes of the staged part of the static data.  Single-busbar kernels (nb1; the old level-header program at the head of the
// int section is not theirs): tier 1 = pair_rc + the flat program of the kernel's group width (n_flat ints), tier 2 = doubles +
// every int table from pair_rc on + the flat program.  NB > 1 kernels: tier 1 = the hot prefix of the int section
// (level-header program + pair_rc).  tier 0: nothing.
__host__ __device__ inline size_t stat_bytes(const StatOff& o, int tier, bool nb1, int n_flat) {
  if (tier == 0) return 0;
  if (!nb1) return ((size_t)o.n_int_hot * 4 + 15) & ~(size_t)15;
  const size_t ints = (size_t)((tier == 2 ? o.n_int : o.n_int_hot) - o.pair_rc) + (size_t)n_flat;
  return ((tier == 2 ? (size_t)o.n_dbl * 8 : 0) + ints * 4 + 15) & ~(size_t)15;
}

// Device view of a FlatProg (gridpf_symbolic.hpp): pass counts and section offsets (ints) of ONE group-width variant.
struct FlatDev {
  int n_fwd, n_scale, n_scale_rhs, n_back, scale_off, back_off, rhs_field0, n_words;
  int wave_closed;   // (group width 128) every destination of a pass is accumulated by ONE wavefront (FlatProg::wave_closed)
  int solo_fwd, solo_back;   // (group width 128) bit k: forward / back pass k only has items in wavefront 0 (FlatProg::solo_fwd)
};
#define GPF_FLATDEV_INTS(X) X(n_fwd) X(n_scale) X(n_scale_rhs) X(n_back) X(scale_off) X(back_off) X(rhs_field0) X(n_words) X(wave_closed) X(solo_fwd) X(solo_back)
static_assert(sizeof(FlatDev) == sizeof(int) * (0 GPF_FLATDEV_INTS(GPF_COUNT_FIELD)), "GPF_FLATDEV_INTS must list every field of FlatDev");
// group width (threads per instance) -> index of its flat-program variant: 16, 32, 64, 128 -> 0 .. 3
__host__ __device__ constexpr int gw_index(int gw) { return gw >= 128 ? 3 : gw >= 64 ? 2 : gw >= 32 ? 1 : 0; }

struct SymDev {
  int n, nslot, nslot_y, n_levels, back_off, n_prog, scale_off, n_scale;
  int nslot_lu;             // blocks of the plain LU: what the level-header program of the NB > 1 kernels touches (Symbolic::nslot_lu)
  int n_up;                 // undirected off-diagonal pairs of the original pattern ((nslot_y - n) / 2)
  int rslot0;               // first right-hand-side pseudo-slot of the single-busbar block array (Symbolic::rslot0)
  const int* flat[4];       // flat programs of the single-busbar kernels, one per group width (gw_index), global memory
  FlatDev fl[4];
  int back_first;           // highest level that has U entries (back substitution starts there)
  int static_connected;     // the substation graph with every line in service is connected (host check at gpf_create)
  const int* prog;          // level-scheduled program in global memory (tools/lu_bench; the kernels use StatView::prog)
  const double* dc_inv_g;   // StatOff::dc_inv == -2: the static DC inverse in global memory
  const double* stat_dbl;   // the static blob: [so.n_dbl] doubles ...
  const int* stat_int;      // ... and [so.n_int] ints
  StatOff so;
};

#define GPF_SYMDEV_INTS(X) X(n) X(nslot) X(nslot_y) X(n_levels) X(back_off) X(n_prog) X(scale_off) X(n_scale) X(nslot_lu) X(n_up) X(rslot0) X(back_first) \
  X(static_connected)

// Topology class (gridpf_capi.hip: build_topo_class): lanes whose substations are SPLIT do not fall back to NB = n_busbar
// blocks -- their bus-level graph (one node per live busbar: node = substation for busbar 1, extra nodes behind) gets its
// own symbolic analysis, cached per distinct line-end / busbar assignment, and the single-busbar kernel runs on it with
// 2x2 blocks.  A class carries what depends on the graph: program, pair table, branch slots and the (substation, busbar)
// -> node table; everything else is the grid's static blob.
struct TopoClassDev {
  SymDev sym;             // n = number of nodes; prog -> this class's program; stat_* / so: the grid's
  const int* pair_rc;     // [nslot_y]
  const int* up;          // [sym.n_up][2]
  const int* br_slot;     // [n_line][4]
  const int* node_of;     // [n_sub][n_busbar] node of a (substation, local busbar - 1), -1: that busbar has no element
};

struct DevParamsS {
  GridDev g;
  Bufs b;
  OutOff oo;
  SymDev sym;
  const TopoClassDev* classes;   // device array (topology-class launches), else nullptr
  int tc_rows;                   // LDS sizing of a topology-class launch: max number of nodes ...
  int tc_nslot, tc_nslot_y;      // ... blocks incl. fill / original-pattern blocks over the classes of the launch
  int dcf;                       // the LDS layout has room for the factored DC matrix (CarveP::Adc): the step kernel keeps it across steps
  EnvDyn env;                    // injection dynamics of the environment (EnvDyn::on == 0: off)
};

// -DGPF_TIMING developer build: cycle-counter stamps are kept in REGISTERS (a global store per stamp would be waited for
// at the next barrier and distort the phases) and written to b.work[inst][32] once at the end of the kernel.
#ifdef GPF_TIMING
constexpr int GPF_NSTAMP = 40;
constexpr int GPF_NPASS_T = 48;     // + per-pass stamps of the first factorisation of a solve (block_lu_flat), kept in LDS
constexpr int GPF_WORK_ROW = GPF_NSTAMP + GPF_NPASS_T;
static __shared__ long long gpf_pass_t[GPF_NPASS_T];
struct Stamps { long long v[GPF_NSTAMP]; };
#define GPF_STAMPS(k) do { stamps.v[(k)] = (long long)__builtin_readcyclecounter(); } while (0)
#define GPF_STAMPS_PARAM , Stamps& stamps
#define GPF_STAMPS_ARG , stamps
#define GPF_STAMPS_DECL Stamps stamps; for (int k_ = 0; k_ < GPF_NSTAMP; ++k_) stamps.v[k_] = 0
#define GPF_STAMPS_FLUSH(inst_) do { if (tid == 0) { for (int k_ = 0; k_ < GPF_NSTAMP; ++k_) P->b.work[(size_t)(inst_) * GPF_WORK_ROW + k_] = (double)stamps.v[k_]; \
    for (int k_ = 0; k_ < GPF_NPASS_T; ++k_) P->b.work[(size_t)(inst_) * GPF_WORK_ROW + GPF_NSTAMP + k_] = (double)gpf_pass_t[k_]; } } while (0)
#else
#define GPF_STAMPS(k) do {} while (0)
#define GPF_STAMPS_PARAM
#define GPF_STAMPS_ARG
#define GPF_STAMPS_DECL do {} while (0)
#define GPF_STAMPS_FLUSH(inst_) do {} while (0)
#endif

// Phase boundaries.  A block is exactly ONE wavefront and the LDS executes the operations of a wavefront in issue order,
// so a boundary would only have to order LDS accesses in the COMPILER (wavefront-scope fences, no instruction; unlike
// __syncthreads() that does not wait for the global stores of the results / injections still in flight)
// (measured: no difference on MI355X, so __syncthreads() stays the default; -DGPF_WAVE_SYNC selects the fences).
#ifndef GPF_WAVE_SYNC
#define GPF_SYNC() __syncthreads()
#else
#define GPF_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

// LIGHT phase boundary for phases that only exchange data through LDS.  The LDS executes the DS instructions of ONE
// wavefront in issue order (reads, writes and atomics alike), so inside a single-wavefront instance a read issued after
// another lane's write / atomic of an earlier phase sees it WITHOUT any s_waitcnt in between: only the COMPILER must be kept
// from moving LDS accesses across the boundary.  __syncthreads() / wavefront-scope fences cost an "s_waitcnt lgkmcnt(0)" per
// phase -- a full LDS round trip (~100+ cycles) on each of the ~150 phases of a step.  Instances served by several wavefronts
// (GW > WAVE) keep the real workgroup barrier.  -DGPF_HARD_SYNC restores __syncthreads() everywhere.
#ifndef GPF_HARD_SYNC
// (several wavefronts per instance: the boundary waits for the wavefront's LDS operations and the workgroup barrier -- NOT, as __syncthreads()
//  would, for its global loads / stores in flight: the flat-program sweeps fetch their item words from L2 two passes ahead, GPF_PF2_ON)
#define GPF_PF2_ON true
#define GPF_LSYNC() do { if (GW > WAVE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } } while (0)
#else
#define GPF_PF2_ON false
#define GPF_LSYNC() GPF_SYNC()
#endif

// Boundary after which lanes read what OTHER lanes of the block wrote to GLOBAL memory (`cond`, block-uniform): a workgroup
// barrier that also drains the vector-memory counter.  A single-wavefront block whose lanes only read back their own global
// stores needs neither (same-lane ordering is the hardware's): the stores of the result row then drain behind the next phases
// instead of being waited for -- a full store round trip, three or four times per environment step.
#define GPF_SYNC_IF(cond) do { if (GW > WAVE || (cond)) GPF_SYNC(); else GPF_LSYNC(); } while (0)

typedef short i16;

// Uniform launch constants (grid sizes, offsets of the symbolic program) reach the kernel through a parameter block in memory.
// Under SGPR pressure the compiler does not keep them in registers: it RE-LOADS them (s_load_dword + s_waitcnt lgkmcnt(0)) right
// where they are used -- inside the phases of the LU, where that wait also drains every LDS operation in flight.  Passing a
// value through pin_sgpr makes it opaque (no longer "a load the compiler may repeat"): it then lives in an SGPR or is spilled to
// a VGPR lane (v_readlane, no memory wait).
#ifndef GPF_JIT
__device__ __forceinline__ void pin_sgpr(int& x) { x = __builtin_amdgcn_readfirstlane(x); asm volatile("" : "+s"(x)); }
#define GPF_SPEC_G(v) do {} while (0)
#define GPF_SPEC_OO(v) do {} while (0)
#define GPF_SPEC_SYM(v) do {} while (0)
#define GPF_SPEC_SO(v) do {} while (0)
#else
// GRID-SPECIALISED BUILD (compiled at run time for ONE grid, gridpf_capi.hip: jit_*; -include of the generated header that defines the
// GPF_JIT_SET_* macros).  Every size of the grid, every offset into its static blob / result row / symbolic program header is a
// literal: the loads of the parameter block, the SGPRs (and SGPR spills to VGPR lanes: v_readlane / v_writelane are VALU instructions)
// that carried them, the address arithmetic on them and the branches on them fold away.  Same source, same arithmetic in the same
// order: results are bit-identical to the ahead-of-time kernels (tests/test_gpu_jit.py).
__device__ __forceinline__ void pin_sgpr(int&) {}
#define GPF_SPEC_G(v) GPF_JIT_SET_G(v)
#define GPF_SPEC_OO(v) GPF_JIT_SET_OO(v)
#define GPF_SPEC_SYM(v) GPF_JIT_SET_SYM(v)
#define GPF_SPEC_SO(v) GPF_JIT_SET_SO(v)
#endif

template <int NB>
struct CarveP {
  static constexpr int BS = 2 * NB;
  double* A;      // [nslot][BS*BS] row-major blocks
  double* Yb;     // [nslot_y][NB*NB][2]
  double* rhs;    // NB > 1: [n_sub][BS].  NB == 1: the right-hand side is pseudo-slot rslot0 + p of A (rows (b0, Re S), (b1, Im S))
  double *vm, *va, *e, *f, *Psp, *Qsp;   // [nbus]
  double* ivm;    // [nbus] 1 / |V| of the current iterate (NB == 1: written with e / f, so that the pair and mismatch phases of the
                  // Newton loop read it instead of running a reciprocal chain per pair / bus)
  double *Sre, *Sim;                     // NB > 1: [nbus] bus injections S; NB == 1: the second column of the pseudo-slots
  double* Gs;     // [nbus] aliases e (shunt conductance: only needed before the Newton loop and by the DC results)
  double* inj;    // [n_inj] staged injection row (only when STAGE; otherwise the lane's row in HBM/L2 is read directly)
  double* Adc;    // [nslot] factored scalar DC matrix kept across the solves of a launch (only when the plan says so, NB == 1)
  int* btype;     // [nbus]
  int* vidx;      // [nbus] last in-service generator of a bus (voltage set-point), -1: none; kept across the solves of a launch
  int* lab;       // [nbus] aliases f (connectivity labels: dead before the Newton loop)
  int* topo;      // alias of A during K1
  i16 *lor_b, *lex_b, *gen_b, *load_b, *sto_b, *sh_b;
  i8* sub_bb;     // [n_sub] live busbar (local id) of each substation (NB == 1)
  float* out_l;   // != nullptr: the float32 results row of the last solve as it was staged in LDS (row-1 half of the block array, dead by
                  // then) before it went to HBM in whole 128-byte lines (solve_instance_sparse: ROWLDS); set by every solve
};

// LDS bytes of ONE instance (without the program copy, which is shared by the IPW instances of a block).
template <int NB>
__host__ __device__ inline size_t lds_bytes_instance(const GridDev& g, int nslot, int nslot_y, bool stage_inj, int n_rows = -1, bool dcf = false) {
  constexpr int BS = 2 * NB;
  const size_t rows = n_rows > 0 ? (size_t)n_rows : (size_t)g.n_sub;      // block rows: substations, or nodes of a topology class
  const size_t nbus = rows * NB;
  // NB == 1: the block array also holds the right-hand side and the bus injections S as `rows` pseudo-slots behind slot
  // rslot0 = max(nslot, ceil(1.5 rows)) (Symbolic::rslot0; the K6 scratch at the head of the array stays below them)
  const size_t rs0 = (size_t)nslot > (3 * rows + 1) / 2 ? (size_t)nslot : (3 * rows + 1) / 2;
  size_t a_d = NB == 1 ? (rs0 + rows) * 4 : (size_t)nslot * BS * BS;
  const size_t topo_d = (((size_t)g.dim_topo + 1) / 2 + 2) & ~(size_t)1;
  if (a_d < topo_d) a_d = topo_d;
  const size_t nd = a_d + (size_t)nslot_y * NB * NB * 2 + (NB == 1 ? 7 * nbus : rows * BS + 8 * nbus) + (stage_inj ? (size_t)g.n_inj : 0) +
                    (dcf ? (size_t)nslot : 0);
  const size_t ni = 2 * nbus;
  const size_t n16 = 2 * (size_t)g.n_line + g.n_gen + g.n_load + g.n_sto + g.n_shunt;
  return (nd * 8 + ni * 4 + n16 * 2 + g.n_sub + 15) & ~(size_t)15;
}
// dynamic LDS of a block: IPW instances + (when staged) one copy of the static blob (stat_bytes, 0 when not staged)
template <int NB>
__host__ __device__ inline size_t lds_bytes_sparse(const GridDev& g, int nslot, int nslot_y, size_t static_bytes, bool stage_inj, int ipw = 1,
                                                   int n_rows = -1, bool dcf = false) {
  return (size_t)ipw * lds_bytes_instance<NB>(g, nslot, nslot_y, stage_inj, n_rows, dcf) + static_bytes;
}

// Byte layout of a lane's blob of kept topology-derived state (KeepArgs, gridpf_common.hpp) for the single-busbar kernels of a grid: the
// host sizes the buffer with it, the kernels read the offsets from StepArgs::keep.
__host__ __device__ inline void keep_layout(KeepArgs& k, const GridDev& g, int nslot, int nslot_y, int n_up) {
  const size_t nbus = (size_t)g.n_sub;
  const size_t n16 = 2 * (size_t)g.n_line + g.n_gen + g.n_load + g.n_sto + g.n_shunt;
  size_t o = (size_t)KEEP_HDR_INTS * 4 + ((size_t)g.dim_topo + g.n_shunt) * 4;
  o = (o + 7) & ~(size_t)7; k.off_kd = (int)o; o += 2 * (size_t)g.n_shunt * 8;
  o = (o + 15) & ~(size_t)15; k.off_m = (int)o;
  k.n_m = (int)((2 * nbus * 4 + n16 * 2 + g.n_sub + 3) / 4); o += (((size_t)k.n_m + 3) & ~(size_t)3) * 4;     // (whole 16-byte chunks)
  o = (o + 15) & ~(size_t)15; k.off_y = (int)o;
  const size_t ny = (size_t)nslot_y > 2 * (size_t)n_up + nbus ? (size_t)nslot_y : 2 * (size_t)n_up + nbus;
  o += ny * 16; k.off_d = (int)o; o += (size_t)nslot * 8;
  k.stride = (long long)((o + 15) & ~(size_t)15);
}

// Stage the static data in LDS (STAGE, see stat_bytes) or view it in place; visible to the block after the first barrier.
// NB1: single-busbar kernel -- its program is the flat program `flat` (n_flat ints, global memory) of the kernel's group width.
// global -> LDS copy of n16 16-byte chunks (both 16-byte aligned) with four loads in flight per lane: a plain element loop waits
// for one L2 round trip per iteration, which made the static staging the longest part of a one-step launch's prologue
__device__ __forceinline__ void stage_copy16(void* dst_lds, const void* src_global, int n16) {
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  v4i_* d = reinterpret_cast<v4i_*>(dst_lds);
  const auto s = (GPF_GLOBAL const v4i_*)src_global;
  const int st = blockDim.x;
  int i = threadIdx.x;
  for (; i + 3 * st < n16; i += 4 * st) {
    const v4i_ t0 = s[i], t1 = s[i + st], t2 = s[i + 2 * st], t3 = s[i + 3 * st];
    d[i] = t0; d[i + st] = t1; d[i + 2 * st] = t2; d[i + 3 * st] = t3;
  }
  for (; i < n16; i += st) d[i] = s[i];
}

template <int STAGE, bool NB1>
__device__ inline void make_stat_view(StatView<STAGE>& sv, const SymDev& S, unsigned char* lds_static, const int* flat, int n_flat) {
  const auto gd = gptr(S.stat_dbl);
  const auto gi = gptr(S.stat_int);
  stat_view(sv, S.so, S.stat_dbl, S.stat_int);
  sv.dc_inv_g.p = S.dc_inv_g;
  if (NB1) sv.prog.p = flat;
  if (STAGE == 0) return;
  if (!NB1) {                                                     // tier 1 of the NB > 1 kernels: [level-header program][pair_rc]
    int* si = reinterpret_cast<int*>(lds_static);
    for (int i = threadIdx.x; i < S.so.n_int_hot; i += blockDim.x) si[i] = gi[i];
    sv.prog.p = si + S.so.prog;
    sv.pair_rc.p = si + S.so.pair_rc;
    sv.up.p = si + S.so.up;
    return;
  }
  double* sd = reinterpret_cast<double*>(lds_static);
  int* si = reinterpret_cast<int*>(STAGE == 2 ? sd + S.so.n_dbl : sd);
  const int i0 = S.so.pair_rc, i1 = STAGE == 2 ? S.so.n_int : S.so.n_int_hot;       // int tables [pair_rc .. )
  // (every table of the blob is padded to 16 bytes: doubles to an even count, ints to a multiple of 4; the flat programs too)
  if (STAGE == 2) stage_copy16(sd, S.stat_dbl, S.so.n_dbl / 2);
  stage_copy16(si, S.stat_int + i0, (i1 - i0) / 4);
  int* sp = si + (i1 - i0);
  stage_copy16(sp, flat, n_flat / 4);
  if (STAGE == 2) stat_view(sv, S.so, sd, si - i0);
  sv.pair_rc.p = si;
  sv.up.p = si + (S.so.up - i0);
  sv.prog.p = sp;
}

template <int NB>
__device__ inline void carve_sparse(CarveP<NB>& c, unsigned char* base, const GridDev& g, int nslot, int nslot_y, bool stage_inj,
                                    int n_rows = -1, bool dcf = false) {
  constexpr int BS = 2 * NB;
  const size_t rows = n_rows > 0 ? (size_t)n_rows : (size_t)g.n_sub;
  const size_t nbus = rows * NB;
  double* d = reinterpret_cast<double*>(base);
  const size_t rs0 = (size_t)nslot > (3 * rows + 1) / 2 ? (size_t)nslot : (3 * rows + 1) / 2;
  size_t a_d = NB == 1 ? (rs0 + rows) * 4 : (size_t)nslot * BS * BS;
  const size_t topo_d = (((size_t)g.dim_topo + 1) / 2 + 2) & ~(size_t)1;
  if (a_d < topo_d) a_d = topo_d;
  c.A = d; c.topo = reinterpret_cast<int*>(d); d += a_d;
  c.Yb = d; d += (size_t)nslot_y * NB * NB * 2;
  c.rhs = d; if (NB > 1) d += rows * BS;
  c.vm = d; d += nbus; c.va = d; d += nbus; c.e = d; c.Gs = d; d += nbus; c.f = d; c.lab = reinterpret_cast<int*>(d); d += nbus;
  c.Psp = d; d += nbus; c.Qsp = d; d += nbus;
  c.ivm = d; if (NB == 1) d += nbus;
  c.Sre = d; if (NB > 1) d += nbus;
  c.Sim = d; if (NB > 1) d += nbus;
  c.inj = d; if (stage_inj) d += g.n_inj;
  c.Adc = d; if (dcf) d += nslot;
  int* i = reinterpret_cast<int*>(d);
  c.btype = i; i += nbus;
  c.vidx = i; i += nbus;
  i16* q = reinterpret_cast<i16*>(i);
  c.lor_b = q; q += g.n_line; c.lex_b = q; q += g.n_line;
  c.gen_b = q; q += g.n_gen; c.load_b = q; q += g.n_load; c.sto_b = q; q += g.n_sto; c.sh_b = q; q += g.n_shunt;
  c.sub_bb = reinterpret_cast<i8*>(q);
}

// Instance groups: a wavefront serves IPW instances, GW = 64 / IPW lanes each (small grids do not have 64-wide work), or
// -- large grids, whose phases loop over hundreds of items -- WPI wavefronts serve ONE instance (GW = 64 * WPI lanes, the
// phase boundaries become real workgroup barriers).  All lanes execute the same instruction stream (same grid, same
// symbolic program); collectives are scoped to the group.
// v + (v moved across lanes by a DPP control): cross-lane adds in the VALU instead of ds_bpermute round trips (~100 cycles each)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over aligned groups of GWS = 16 / 32 / 64 lanes of a wavefront, returned to every lane of the group
template <int GWS>
__device__ __forceinline__ double dpp_group_sum(double v) {
  v = dpp_add_f64<0xB1, 0xF>(v);                 // quad_perm [1,0,3,2]
  v = dpp_add_f64<0x4E, 0xF>(v);                 // quad_perm [2,3,0,1]
  v = dpp_add_f64<0x141, 0xF>(v);                // row_half_mirror
  v = dpp_add_f64<0x140, 0xF>(v);                // row_mirror: every lane of a 16-lane row holds the row sum
  if (GWS == 16) return v;
  v = dpp_add_f64<0x142, 0xA>(v);                // row_bcast15 into rows 1 and 3: they hold the sums of rows 0+1 / 2+3
  if (GWS == 32) {
    const double s0 = readlane_f64(v, 31), s1 = readlane_f64(v, 63);
    return (threadIdx.x & 32) ? s1 : s0;
  }
  v = dpp_add_f64<0x143, 0xC>(v);                // row_bcast31 into rows 2 and 3: lane 63 holds the wavefront sum
  return readlane_f64(v, 63);
}

// workgroup barrier for exchanges that only go through LDS (the collectives below): waits for the wavefront's LDS operations, not -- as
// __syncthreads() would -- for its global loads / stores in flight
#ifndef GPF_HARD_SYNC
#define GPF_WG_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define GPF_WG_LDS_BARRIER() __syncthreads()
#endif
template <int IPW, int WPI = 1>
struct Grp {
  static_assert(IPW == 1 || WPI == 1, "either several instances per wavefront or several wavefronts per instance");
  static constexpr int GW = WPI > 1 ? WAVE * WPI : WAVE / IPW;
  static constexpr int BLOCK = WAVE * WPI;
  static __device__ __forceinline__ unsigned long long mask() {
    return IPW == 1 ? ~0ull : (((1ull << (GW & 63)) - 1ull) << ((threadIdx.x / GW) * GW));
  }
  // over the lanes of the caller's instance
  static __device__ __forceinline__ bool any(bool x) {
    if (WPI > 1) return __syncthreads_or(x) != 0;
    return IPW == 1 ? (bool)__any(x) : ((__ballot(x) & mask()) != 0ull);
  }
  static __device__ __forceinline__ int count(bool x) {
    if (WPI > 1) return __syncthreads_count(x);
    return __popcll(__ballot(x) & mask());
  }
  // index inside the caller's instance of the first lane with x (single-wavefront instances only), -1: none
  static __device__ __forceinline__ int first(bool x) {
    const unsigned long long b = __ballot(x) & mask();
    return b ? (int)__ffsll((long long)b) - 1 - (IPW == 1 ? 0 : (int)(threadIdx.x / GW) * GW) : -1;
  }
  static __device__ __forceinline__ double sum(double v) {
    if (WPI > 1) {
      __shared__ double red_[WPI > 1 ? WPI : 1];
#pragma unroll
      for (int off = WAVE / 2; off; off >>= 1) v += __shfl_xor(v, off);
      GPF_WG_LDS_BARRIER();
      if ((threadIdx.x & (WAVE - 1)) == 0) red_[threadIdx.x / WAVE] = v;
      GPF_WG_LDS_BARRIER();
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < WPI; ++k) t += red_[k];
      return t;
    }
    return dpp_group_sum<(GW < WAVE ? GW : WAVE)>(v);
  }
  // two flags at once (bit 0 = any(a), bit 1 = any(b)).  Several wavefronts per instance: ONE exchange through LDS and ONE
  // barrier instead of two __syncthreads_or (two barriers each); BUF selects the exchange buffer -- two calls with the same BUF
  // must be separated by at least one other workgroup barrier (the slower wavefront may still be reading).
  template <int BUF, int N>
  static __device__ __forceinline__ unsigned any_bits(unsigned lane_bits) {      // bit k of the result = any lane has bit k set
    unsigned w = 0;
    if (WPI > 1 || IPW == 1) {
#pragma unroll
      for (int k = 0; k < N; ++k) w |= (__ballot((lane_bits >> k) & 1u) != 0ull ? 1u : 0u) << k;
    } else {
      const unsigned long long m = mask();
#pragma unroll
      for (int k = 0; k < N; ++k) w |= ((__ballot((lane_bits >> k) & 1u) & m) != 0ull ? 1u : 0u) << k;
    }
    if (WPI > 1) {
      __shared__ unsigned redb_[4][WPI > 1 ? WPI : 1];
      if ((threadIdx.x & (WAVE - 1)) == 0) redb_[BUF][threadIdx.x / WAVE] = w;
      GPF_WG_LDS_BARRIER();
      unsigned r = 0;
#pragma unroll
      for (int k = 0; k < WPI; ++k) r |= redb_[BUF][k];
      return r;
    }
    return w;
  }
  template <int BUF>
  static __device__ __forceinline__ unsigned any2(bool a, bool b) { return any_bits<BUF, 2>((a ? 1u : 0u) | (b ? 2u : 0u)); }
  // two sums at once (several wavefronts per instance: one exchange, one barrier; same BUF rule as any2)
  template <int BUF>
  static __device__ __forceinline__ void sum2(double& a, double& b) {
    if (WPI > 1) {
      __shared__ double reds_[2][2 * (WPI > 1 ? WPI : 1)];
#pragma unroll
      for (int off = WAVE / 2; off; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if ((threadIdx.x & (WAVE - 1)) == 0) { reds_[BUF][2 * (threadIdx.x / WAVE)] = a; reds_[BUF][2 * (threadIdx.x / WAVE) + 1] = b; }
      GPF_WG_LDS_BARRIER();
      double ta = 0.0, tb = 0.0;
#pragma unroll
      for (int k = 0; k < WPI; ++k) { ta += reds_[BUF][2 * k]; tb += reds_[BUF][2 * k + 1]; }
      a = ta; b = tb;
      return;
    }
    a = dpp_group_sum<(GW < WAVE ? GW : WAVE)>(a);
    b = dpp_group_sum<(GW < WAVE ? GW : WAVE)>(b);
  }
  // over ALL lanes of the block (every instance of the wavefront / every wavefront of the instance)
  static __device__ __forceinline__ bool block_any(bool x) { return WPI > 1 ? (__syncthreads_or(x) != 0) : (bool)__any(x); }
  static __device__ __forceinline__ bool block_all(bool x) { return WPI > 1 ? (__syncthreads_and(x) != 0) : (bool)__all(x); }
  // the same for a value that is already uniform over the caller's instance (the result of a group collective, a loop counter):
  // with one instance per block there is nothing to reduce
  static __device__ __forceinline__ bool block_any_u(bool x) { return IPW == 1 ? x : (bool)__any(x); }
  static __device__ __forceinline__ bool block_all_u(bool x) { return IPW == 1 ? x : (bool)__all(x); }
};

constexpr int BT_OFF = -1;   // inactive bus

// inverse of a BS x BS block held in registers (BS = 2: closed form; BS > 2: Gauss-Jordan with partial pivoting).
template <int BS>
__device__ __forceinline__ bool block_inverse(const double (&D)[BS * BS], double (&Di)[BS * BS]) {
  if (BS == 2) {
    const double det = D[0] * D[3] - D[1] * D[2];
    const double r = 1.0 / det;
    Di[0] = D[3] * r; Di[1] = -D[1] * r; Di[2] = -D[2] * r; Di[3] = D[0] * r;
    return fabs(det) > 1e-300 && fabs(det) < 1e300;
  }
  double M[BS][2 * BS];
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int q = 0; q < BS; ++q) { M[r][q] = D[r * BS + q]; M[r][BS + q] = (r == q) ? 1.0 : 0.0; }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < BS; ++k) {
    // partial pivoting by conditional row swaps (branch-free, fully unrolled)
#pragma unroll
    for (int r = k + 1; r < BS; ++r) {
      const bool sw = fabs(M[r][k]) > fabs(M[k][k]);
#pragma unroll
      for (int q = 0; q < 2 * BS; ++q) {
        const double a = M[k][q], b = M[r][q];
        M[k][q] = sw ? b : a;
        M[r][q] = sw ? a : b;
      }
    }
    const double pv = M[k][k];
    if (!(fabs(pv) > 1e-300) || !(fabs(pv) < 1e300)) ok = false;
    const double rp = 1.0 / pv;
#pragma unroll
    for (int q = 0; q < 2 * BS; ++q) M[k][q] *= rp;
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      if (r == k) continue;
      const double m = M[r][k];
#pragma unroll
      for (int q = 0; q < 2 * BS; ++q) M[r][q] = fma(-m, M[k][q], M[r][q]);
    }
  }
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int q = 0; q < BS; ++q) Di[r * BS + q] = M[r][BS + q];
  return ok;
}

// Level-scheduled block-sparse LU + solve, in place in LDS, for the NB > 1 kernels (blocks of 4x4 / 6x6; the single-busbar
// kernels run block_lu_flat below).  A: [nslot][BS*BS] blocks; rhs: [n][BS] right-hand side -> solution.  All pivots of a
// level are eliminated concurrently; trailing updates that hit the same block are combined with LDS f64 atomics.
template <int BS, int GW = WAVE, class PP = const int*>
__device__ inline bool block_lu_solve(const SymDev& S, PP prog, double* __restrict__ A,
                                      double* __restrict__ rhs, int tid, long long* dbg = nullptr) {
#ifdef GPF_TIMING
  const long long t_lu0 = __builtin_readcyclecounter();
#endif
  constexpr int B2 = BS * BS;
  constexpr int CHB = (GW / B2) * B2;      // U-block items per chunk: whole blocks only
  constexpr int CHR = (GW / BS) * BS;
  bool ok = true;
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const int piv_off = prog[8 * lv], n_piv = prog[8 * lv + 1], b_off = prog[8 * lv + 2], n_b = prog[8 * lv + 3], c_off = prog[8 * lv + 4],
              n_c = prog[8 * lv + 5], r_off = prog[8 * lv + 6], n_r = prog[8 * lv + 7];
    // (a) invert the pivot blocks in place
    for (int q = tid; q < n_piv; q += GW) {
      double* Ad = A + (size_t)prog[piv_off + q] * B2;      // diag slot of substation p is slot p
      double D[B2], Di[B2];
#pragma unroll
      for (int m = 0; m < B2; ++m) D[m] = Ad[m];
      if (!block_inverse<BS>(D, Di)) ok = false;
#pragma unroll
      for (int m = 0; m < B2; ++m) Ad[m] = Di[m];
    }
    GPF_LSYNC();
    // (b) scale the pivot block rows and right-hand sides: U'_pj = Dinv_p * A_pj, b'_p = Dinv_p * b_p
    //     (items of one block read a whole block column: every pass reads first, then writes)
    if (n_b * B2 + n_piv * BS <= GW) {
      const int nu = n_b * B2;
      double acc = 0.0;
      int dst = -1;            // >= 0: A element index; <= -2: rhs element index -(dst+2)
      if (tid < nu) {
        const unsigned w = (unsigned)prog[b_off + tid / B2];
        const int us = (int)(w & 0xffffu);
        const int r = (tid % B2) / BS, q = tid % BS;
        const double* Di = A + (size_t)(w >> 16) * B2 + r * BS;
        const double* Au = A + (size_t)us * B2 + q;
#pragma unroll
        for (int m = 0; m < BS; ++m) acc = fma(Di[m], Au[m * BS], acc);
        dst = us * B2 + (tid % B2);
      } else if (tid < nu + n_piv * BS) {
        const int it = tid - nu;
        const int p = prog[piv_off + it / BS];
        const double* Di = A + (size_t)p * B2 + (it % BS) * BS;
        const double* bp = rhs + (size_t)p * BS;
#pragma unroll
        for (int m = 0; m < BS; ++m) acc = fma(Di[m], bp[m], acc);
        dst = -(p * BS + (it % BS)) - 2;
      }
      GPF_LSYNC();
      if (dst >= 0) A[dst] = acc;
      else if (dst <= -2) rhs[-(dst + 2)] = acc;
      GPF_LSYNC();
    } else {
    for (int base = 0; base < n_b * B2; base += CHB) {
      const int it = base + tid;
      const bool on = tid < CHB && it < n_b * B2;
      double acc = 0.0;
      int us = 0;
      if (on) {
        const unsigned w = (unsigned)prog[b_off + it / B2];
        us = (int)(w & 0xffffu);
        const int r = (it % B2) / BS, q = it % BS;
        const double* Di = A + (size_t)(w >> 16) * B2 + r * BS;
        const double* Au = A + (size_t)us * B2 + q;
#pragma unroll
        for (int m = 0; m < BS; ++m) acc = fma(Di[m], Au[m * BS], acc);
      }
      GPF_LSYNC();
      if (on) A[(size_t)us * B2 + (it % B2)] = acc;
      GPF_LSYNC();
    }
    for (int base = 0; base < n_piv * BS; base += CHR) {
      const int it = base + tid;
      const bool on = tid < CHR && it < n_piv * BS;
      double acc = 0.0;
      int p = 0;
      if (on) {
        p = prog[piv_off + it / BS];
        const double* Di = A + (size_t)p * B2 + (it % BS) * BS;
        const double* bp = rhs + (size_t)p * BS;
#pragma unroll
        for (int m = 0; m < BS; ++m) acc = fma(Di[m], bp[m], acc);
      }
      GPF_LSYNC();
      if (on) rhs[(size_t)p * BS + (it % BS)] = acc;
      GPF_LSYNC();
    }
    }
    // (c) trailing updates A[dst] -= A[l] * U'[u] and rhs[row] -= A[l] * b'[p] (LDS atomics: blocks / rows may collide)
    for (int it = tid; it < n_c * B2; it += GW) {
      const int o = it / B2, r = (it % B2) / BS, q = it % BS;
      const unsigned w0 = (unsigned)prog[c_off + 2 * o];
      const int u = prog[c_off + 2 * o + 1] & 0xffff;
      const double* Al = A + (size_t)(w0 >> 16) * B2 + r * BS;
      const double* Au = A + (size_t)u * B2 + q;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < BS; ++m) acc = fma(Al[m], Au[m * BS], acc);
      atomicAdd(&A[(size_t)(w0 & 0xffffu) * B2 + r * BS + q], -acc);
    }
    for (int it = tid; it < n_r * BS; it += GW) {
      const int o = it / BS, r = it % BS;
      const unsigned w0 = (unsigned)prog[r_off + 2 * o];
      const int p = prog[r_off + 2 * o + 1];
      const double* Al = A + (size_t)(w0 & 0xffffu) * B2 + r * BS;
      const double* bp = rhs + (size_t)p * BS;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < BS; ++m) acc = fma(Al[m], bp[m], acc);
      atomicAdd(&rhs[(size_t)(w0 >> 16) * BS + r], -acc);
    }
    GPF_LSYNC();
  }
#ifdef GPF_TIMING
  const long long t_lu1 = __builtin_readcyclecounter();
#endif
  // back substitution, levels in reverse: x_p = b'_p - sum_j U'_pj x_j.  Pivots of a level are independent and only
  // depend on later levels, so every (pivot, U entry, row) item of a level runs in parallel and accumulates with
  // ds_add_f64.
  for (int lv = S.n_levels - 1; lv >= 0; --lv) {
    const int ent_off = prog[S.back_off + 2 * lv], n_ent = prog[S.back_off + 2 * lv + 1];
    for (int it = tid; it < n_ent * BS; it += GW) {
      const unsigned w = (unsigned)prog[ent_off + 2 * (it / BS)];      // u_slot | (u_col << 16)
      const int p = prog[ent_off + 2 * (it / BS) + 1], r = it % BS;
      const double* Au = A + (size_t)(w & 0xffffu) * B2 + r * BS;
      const double* xj = rhs + (size_t)(w >> 16) * BS;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < BS; ++m) acc = fma(Au[m], xj[m], acc);
      atomicAdd(&rhs[(size_t)p * BS + r], -acc);
    }
    GPF_LSYNC();
  }
#ifdef GPF_TIMING
  if (dbg) { dbg[0] = t_lu1 - t_lu0; dbg[1] = (long long)__builtin_readcyclecounter() - t_lu1; }
#endif
  return ok;
}

// ---- flat-program sweeps (gridpf_symbolic.hpp: FlatProg) -------------------------------------------------------------------
// Block LU + solve with 2x2 blocks on the flat program.  A: row 0 of every (pseudo-)slot at A + slot * 2, row 1 at
// A + HS + slot * 2 (HS = (rslot0 + n) * 2 doubles); the right-hand side lives in the pseudo-slots and holds s = D x on return
// (the solution is x_p = inv(D_p) s_p with the factored diagonal block D_p left in slot p).  The pivot inverse
// is recomputed by every item from the never-overwritten diagonal block; U and the right-hand side are NEVER scaled (round 3: the
// deferred scaling pass is gone): the back substitution accumulates s_p -= A_pj inv(D_j) s_j with ds_add_f64 and the caller forms
// x_p = inv(D_p) s_p where it consumes the solution (flat_solution).  What a phase costs is its instruction count: no level headers,
// no bounds / clamps, no "trailing update or right-hand side?" selects, byte offsets instead of slot indices.
template <int GW, class PP = const int*>
__device__ inline bool block_lu_flat(const FlatDev& F, PP prog, double* __restrict__ A, size_t HS, int tid, long long* dbg = nullptr) {
#ifdef GPF_TIMING
  const long long t_lu0 = __builtin_readcyclecounter();
#endif
  bool ok = true;
  // item words fetched TWO passes ahead: several wavefronts per instance (see GPF_LSYNC), and -- GPF_PF2_IG -- the instance-group kernels, whose
  // passes (~500 cycles) are no longer than an L2 round trip under load
#ifdef GPF_PF2_IG
  constexpr bool PF2 = GPF_PF2_ON && (GW > WAVE || (GW <= 32 && !std::is_same<PP, const int*>::value));
#else
  constexpr bool PF2 = GPF_PF2_ON && GW > WAVE;
#endif
  char* const a0 = reinterpret_cast<char*>(A);
  char* const a1 = a0 + HS * 8;
#define FL_LD2(base, f) (*reinterpret_cast<const double2*>((base) + (f)))
#define FL_D(base, f) (reinterpret_cast<double*>((base) + (f)))
#ifdef GPF_TIMING
#define GPF_PASS_T(i_) do { if (dbg && tid == 0 && (i_) < GPF_NPASS_T) gpf_pass_t[(i_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GPF_PASS_T(i_) do {} while (0)
#endif
  GPF_PASS_T(0);
  // the lane's item of a forward pass: A[dst] -= A[l] inv(D_p) A[u]
  auto fwd_item = [&](const unsigned w0, const unsigned w1) {
    if (w0 != 0xffffffffu) {
      const unsigned fd = w0 & 0xffffu, fl = w0 >> 16, fu = w1 & 0xffffu, fp = w1 >> 16;
      const double2 dA = FL_LD2(a0, fp), dB = FL_LD2(a1, fp), lA = FL_LD2(a0, fl), lB = FL_LD2(a1, fl);
      const double2 uA = FL_LD2(a0, fu), uB = FL_LD2(a1, fu);
      const double rd = -fast_rcp(fma(dA.x, dB.y, -dA.y * dB.x));
      // T = A_l * adj(D)
      const double t00 = fma(lA.x, dB.y, -lA.y * dB.x), t01 = fma(lA.y, dA.x, -lA.x * dA.y);
      const double t10 = fma(lB.x, dB.y, -lB.y * dB.x), t11 = fma(lB.y, dA.x, -lB.x * dA.y);
      double* d0_ = FL_D(a0, fd);
      double* d1_ = FL_D(a1, fd);
      atomicAdd(&d0_[0], fma(t00, uA.x, t01 * uB.x) * rd);
      atomicAdd(&d1_[0], fma(t10, uA.x, t11 * uB.x) * rd);
      if (fd < (unsigned)F.rhs_field0) {     // (a right-hand-side pseudo-slot: its second column is padding, nothing to accumulate)
        atomicAdd(&d0_[1], fma(t00, uA.y, t01 * uB.y) * rd);
        atomicAdd(&d1_[1], fma(t10, uA.y, t11 * uB.y) * rd);
      }
    }
  };
  // Walk of a sweep's passes: lane t executes item t of every pass, then the phase boundary.
  //  * The words of pass k + 1 are fetched before pass k computes; PF2 (several wavefronts per instance, whose phase boundary does not
  //    wait for global loads): those of pass k + 2, through three register pairs in rotation -- a plain copy "next = the one after"
  //    would wait for the load it copies.
  //  * SOLO passes (several wavefronts per instance; FlatDev::solo_fwd / solo_back): every item of the pass sits in wavefront 0, which
  //    runs it alone, and between two consecutive solo passes there is NO workgroup barrier and no wait for the atomics -- the LDS
  //    executes one wavefront's operations in issue order, the other wavefront sleeps at the barrier that ends the run.  A pass is
  //    latency (LDS round trip -> arithmetic -> LDS round trip, tools/lds_pass_bench.hip: 450 cycles for one wavefront alone, 630 -- 800
  //    for two wavefronts with the barrier and 4 instances per CU): the narrow tail levels and the back substitution of the
  //    118-substation grids (12 of 19 passes per factorisation) pay the single-wavefront price.
#define GPF_WALK(n_, at0_, MASK_, NEXT_SOLO_, ITEM_, TB_)                                                                                   \
  {                                                                                                                                \
    const unsigned solo_ = GW > WAVE ? (unsigned)(MASK_) : 0u;                                                                     \
    const int n_pass_ = (n_);                                                                                                      \
    auto step_ = [&](const unsigned a_, const unsigned b_, const int k) {                                                          \
      const bool so_ = GW > WAVE && k < 32 && ((solo_ >> (k & 31)) & 1u);                                                          \
      if (!so_ || tid < WAVE) ITEM_(a_, b_);                                                                                       \
      const bool sn_ = so_ && (k + 1 < n_pass_ ? (k + 1 < 32 && ((solo_ >> ((k + 1) & 31)) & 1u)) : (GW > WAVE && (NEXT_SOLO_)));  \
      if (sn_) { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }                 \
      else GPF_LSYNC();                                                                                                            \
      GPF_PASS_T((TB_) + k);                                                                                                       \
    };                                                                                                                             \
    int at = (at0_) + 2 * tid;                                                                                                     \
    unsigned p0 = (unsigned)prog[at], p1 = (unsigned)prog[at + 1];                                                                 \
    if (PF2) {                                                                                                                     \
      at += 2 * GW;                                                                                                                \
      unsigned q0 = (unsigned)prog[at], q1 = (unsigned)prog[at + 1], r0, r1;                                                       \
      for (int k = 0; k < n_pass_; k += 3) {                                                                                       \
        at += 2 * GW; r0 = (unsigned)prog[at]; r1 = (unsigned)prog[at + 1];                                                        \
        step_(p0, p1, k);                                                                                                          \
        if (k + 1 >= n_pass_) break;                                                                                               \
        at += 2 * GW; p0 = (unsigned)prog[at]; p1 = (unsigned)prog[at + 1];                                                        \
        step_(q0, q1, k + 1);                                                                                                      \
        if (k + 2 >= n_pass_) break;                                                                                               \
        at += 2 * GW; q0 = (unsigned)prog[at]; q1 = (unsigned)prog[at + 1];                                                        \
        step_(r0, r1, k + 2);                                                                                                      \
      }                                                                                                                            \
    } else {                                                                                                                       \
      for (int k = 0; k < n_pass_; ++k) {                                                                                          \
        at += 2 * GW;                                                                                                              \
        const unsigned n0 = (unsigned)prog[at], n1 = (unsigned)prog[at + 1];     /* words of the next pass */                      \
        step_(p0, p1, k);                                                                                                          \
        p0 = n0; p1 = n1;                                                                                                          \
      }                                                                                                                            \
    }                                                                                                                              \
  }
  GPF_WALK(F.n_fwd, 0, F.solo_fwd, F.n_back > 0 && (F.solo_back & 1), fwd_item, 1)   // (the first back pass continues a solo run)
#ifdef GPF_TIMING
  const long long t_lu1 = __builtin_readcyclecounter();
#endif
  // back substitution, levels in reverse, every entry of a level concurrently: s_p -= A_pj x_j with x_j = inv(D_j) s_j formed by
  // the item itself from the column's accumulated right-hand side (complete: j was eliminated after p, its own entries ran in an
  // earlier pass) -- U and the right-hand side are never scaled, so there is no scaling pass between the two sweeps
  auto back_item = [&](const unsigned w0, const unsigned w1) {
    if (w0 != 0xffffffffu) {
      const unsigned fu = w0 & 0xffffu, fj = w0 >> 16, fdj = fj - (unsigned)F.rhs_field0;
      const double2 uA = FL_LD2(a0, fu), uB = FL_LD2(a1, fu), dA = FL_LD2(a0, fdj), dB = FL_LD2(a1, fdj);
      const double s0 = *FL_D(a0, fj), s1 = *FL_D(a1, fj);
      const double rd = fast_rcp(fma(dA.x, dB.y, -dA.y * dB.x));
      const double x0 = fma(dB.y, s0, -dA.y * s1) * rd, x1 = fma(dA.x, s1, -dB.x * s0) * rd;
      atomicAdd(FL_D(a0, w1), -fma(uA.x, x0, uA.y * x1));
      atomicAdd(FL_D(a1, w1), -fma(uB.x, x0, uB.y * x1));
    }
  };
  GPF_WALK(F.n_back, F.back_off, F.solo_back, false, back_item, 1 + F.n_fwd)
#ifdef GPF_TIMING
  if (dbg) { dbg[0] = t_lu1 - t_lu0; dbg[1] = (long long)__builtin_readcyclecounter() - t_lu1; }
#endif
  return ok;
}

// Scalar variant for the DC system of the single-busbar layout: B' theta = P only couples element [0][0] of every block and the
// first entry of every right-hand-side pseudo-slot (the |V| rows are identity), so the same flat program runs on scalars.
// FACTOR = false: `fac` holds the FACTORED matrix of an earlier solve with the same topology (L, D and the unscaled U as left by
// the forward sweep), COMPACT: as one double per slot (CarveP::Adc) instead of element [0][0] of the row-0 half; only the
// right-hand-side items of the forward sweep and the back substitution run.  On return the first entry of pseudo-slot p holds
// s_p = d_p * theta_p (the caller divides by the pivot, like the 2x2 sweep's callers apply inv(D_p)).
template <int GW, bool FACTOR, bool COMPACT, class PP = const int*>
__device__ inline bool scalar_lu_flat(const FlatDev& F, PP prog, double* __restrict__ A, double* __restrict__ fac, int tid,
                                      long long* dbg = nullptr) {
#ifdef GPF_TIMING
  const long long t_lu0 = __builtin_readcyclecounter();
#endif
  bool ok = true;
#ifdef GPF_PF2_IG
  constexpr bool PF2 = GPF_PF2_ON && (GW > WAVE || (GW <= 32 && !std::is_same<PP, const int*>::value));
#else
  constexpr bool PF2 = GPF_PF2_ON && GW > WAVE;
#endif
  char* const a0 = reinterpret_cast<char*>(A);
  char* const fb = reinterpret_cast<char*>(fac);
  auto facp = [&](unsigned f) -> double* { return reinterpret_cast<double*>(fb + (COMPACT ? (f >> 1) : f)); };
  auto fwd_item = [&](const unsigned w0, const unsigned w1) {
    const unsigned fd = w0 & 0xffffu, fl = w0 >> 16, fu = w1 & 0xffffu, fp = w1 >> 16;
    const bool is_r = (int)fd >= F.rhs_field0;
    if (w0 != 0xffffffffu && (FACTOR || is_r)) {
      const double d = *facp(fp), al = *facp(fl);
      const double x = is_r ? *FL_D(a0, fu) : *facp(fu);
      double* dst = is_r ? FL_D(a0, fd) : facp(fd);
      atomicAdd(dst, -(al * x) * fast_rcp(d));
    }
  };
#undef GPF_PASS_T
#define GPF_PASS_T(i_) do {} while (0)
  GPF_WALK(F.n_fwd, 0, F.solo_fwd, F.n_back > 0 && (F.solo_back & 1), fwd_item, 0)
#ifdef GPF_TIMING
  const long long t_lu1 = __builtin_readcyclecounter();
#endif
  auto back_item = [&](const unsigned w0, const unsigned w1) {
    if (w0 != 0xffffffffu) {
      const unsigned fj = w0 >> 16;
      atomicAdd(FL_D(a0, w1), -(*facp(w0 & 0xffffu) * *FL_D(a0, fj)) * fast_rcp(*facp(fj - (unsigned)F.rhs_field0)));
    }
  };
  GPF_WALK(F.n_back, F.back_off, F.solo_back, false, back_item, 0)
#ifdef GPF_TIMING
  if (dbg) { dbg[0] = t_lu1 - t_lu0; dbg[1] = (long long)__builtin_readcyclecounter() - t_lu1; }
#endif
  return ok;
}
#undef GPF_PASS_T
#undef GPF_WALK
#undef FL_LD2
#undef FL_D

// What a solve may take over from the previous solve of the same block (multi-step launches, cascade rounds).
struct SolveCtl {
  bool inj_staged;    // the injection row is already in CarveP::inj
  bool topo_staged;   // the topology row is already in CarveP::topo
  bool reuse;         // BLOCK-UNIFORM: same topology as the previous solve of this block -> the element->bus maps, bus types,
                      // connectivity verdict and Ybus blocks in LDS are valid (and the DC factors when dcf)
  bool dcf;           // CarveP::Adc holds / receives the factored DC matrix (NB == 1)
  bool write_bus;     // write the float64 bus voltages (parity checks, the facade's stale-bus angles)
  bool sums_done;     // (with reuse) the caller already accumulated the bus injections Psp / Qsp / Gs of this solve (step kernel: K9
                      // scatters every element's new set-point as it computes it)
  bool warm;          // OPT-IN, not the reference's algorithm: with `reuse`, Newton starts from the previous solve's voltages
                      // (CarveP::va / vm still hold them) instead of the DC initialisation pandapower does on every call
  bool otraj;         // the results go to row `orow` of the per-step observation trajectory (Bufs::traj_out ...) instead of the
  int orow;           // lane's own row `orow` = lane of out / topo_out / shunt_bus_out / line_status (see write_nan_results)
  bool write_topo;    // write the topology-only outputs (topo_vect, line status, shunt buses) even when `reuse` says they stand:
                      // every row of the observation trajectory is complete
  // (several wavefronts per instance, tables in global memory, step kernel) the chronics-driven injections of this step were NOT written to the
  // lane's injection row: K9's owner lanes -- wavefront 0, load tid / tid + 64, generator tid -- hand them over in registers (a step of a
  // multi-step launch whose topology stands; the row gets the last step's values).  2.8 KB less HBM traffic per lane and step on 118 substations.
  bool inj_regs;
  float r_lp0, r_lq0, r_lp1, r_lq1, r_pp, r_vm;
  bool tc_rebuild;    // (with reuse + write_topo) TopoState::tc is not this launch's: the first topology positions are derived again (state loaded from a KeepArgs blob)
};
// per-group results of the topology phases, kept by the caller across solves
struct TopoState {
  int status;         // 0, GPF_ST_NOSLACK or GPF_ST_ISLANDED
  int nb;             // number of active buses
  bool dc_base;       // the DC start of this topology comes from the static inverse of the reference DC matrix (StatOff::dc_inv): every
                      // line in service, or (dc_out >= 0) exactly ONE line out, whose effect is a rank-1 correction
  int dc_out;         // the line that is out (Sherman-Morrison correction of the static inverse), -1: none
  bool gen_base;      // every generator is connected (single-busbar layout): the static per-generator bus totals apply
  int tc[2];          // topo_vect values of positions tid, tid + GW as the last solve with a NEW topology wrote them
};

// ---------------------------------------------------------------------------------------------------
// One complete power flow of the IPW instances of a wavefront (tid = lane within the instance group).  Returns the GPF_ST_*
// status of the caller's group.  Groups share the instruction stream: a group that has failed or finished keeps executing
// (its state is frozen / its results are overwritten by the caller), so barriers stay wave-uniform.
// YR (large single-busbar grids, 2 wavefronts per instance, tables in global memory): the Ybus blocks and pair-table words of the
// (at most YR_PASSES * GW) pairs stay in REGISTERS of the lane that owns the pair (yreg / rcreg, owned by the kernel so that they
// survive from one solve to the next like the LDS copy does); the 7.6 KB of LDS this frees on a 118-substation grid hold the
// factored DC matrix (CarveP::Adc) instead, which no longer has to be rebuilt and refactored by every step of a launch.
constexpr int YR_PASSES = 4;
template <int NB, int STAGE, int IPW, int WPI, bool TC, bool YR = false>
__device__ inline int solve_instance_sparse(const DevParamsS* __restrict__ P, const SymDev& S, const FlatDev& FL, const StatView<STAGE>& sv, CarveP<NB>& c, double2* yreg, unsigned* rcreg, int inst, int is_dc, int max_iter,
                                            double tol_pu, int tid, const SolveCtl& ctl, TopoState& ts, int& n_iter_out, int& nb_out, float& a_or_first GPF_STAMPS_PARAM) {
  typedef Grp<IPW, WPI> G;
  constexpr int GW = G::GW;
  constexpr int BS = 2 * NB;
  constexpr int B2 = BS * BS;
  GridDev g_loc = P->g;                                  // sizes pinned in registers (see pin_sgpr)
  GPF_SPEC_G(g_loc);
  pin_sgpr(g_loc.n_sub); pin_sgpr(g_loc.n_line); pin_sgpr(g_loc.n_gen); pin_sgpr(g_loc.n_load); pin_sgpr(g_loc.n_sto);
  pin_sgpr(g_loc.n_shunt); pin_sgpr(g_loc.dim_topo); pin_sgpr(g_loc.nb_tot);
  const GridDev& g = g_loc;
  const Bufs& b = P->b;
#ifdef GPF_JIT
  OutOff oo_loc = P->oo;
  GPF_SPEC_OO(oo_loc);
  const OutOff& oo = oo_loc;
#else
  const OutOff& oo = P->oo;
#endif
  const int nsub = g.n_sub;
  const int nbus = TC ? S.n : nsub * NB;               // block rows x NB: substations, or the nodes of the topology class
  // BITWISE REPRODUCIBILITY with several wavefronts per instance.  The LDS applies the f64 atomics of ONE wavefront in issue order,
  // those of two wavefronts in a timing-dependent order -- and floating-point addition is not associative.  So (1) the loops in
  // which element / line lanes accumulate into bus or block sums run on wavefront 0 alone (GWA lanes: they are short and, but for
  // the injection sums, only run when the topology changed), (2) the flat LU programs keep all items of a destination inside one
  // wavefront per pass (gridpf_symbolic.hpp: build_flat), (3) the Newton loop's S_i = sum_j T_ij is accumulated per wavefront --
  // wavefront 0 in the S column of the pseudo-slots, wavefront 1 in their (then unused) right-hand-side column -- and the two
  // partial sums are added in a fixed order by the mismatch phase.  Single-wavefront instances: GWA = GW, nothing changes.
  constexpr int GWA = WPI > 1 ? WAVE : GW;
  const bool acc_lane = WPI == 1 || tid < WAVE;
  const bool wave1 = WPI > 1 && tid >= WAVE;
  // element (r, col) of block `slot`: 2x2 blocks are stored split by row (see block_lu_solve), larger blocks contiguously
  const size_t HS = ((size_t)S.rslot0 + S.n) * 2;         // doubles per row half (single-busbar layout: slots + right-hand-side pseudo-slots)
  auto bel = [&](int slot, int r, int col) -> double* {
    return BS == 2 ? c.A + (size_t)r * HS + (size_t)slot * 2 + col : c.A + (size_t)slot * B2 + r * BS + col;
  };
  // right-hand side / solution (theta, |V| entries) and bus injection S of bus i.  Single-busbar layout: pseudo-slot rslot0 + i of
  // the block array, rows (b_theta, Re S) and (b_V, Im S); NB > 1: the separate arrays
  auto rhsT = [&](int i) -> double* { return BS == 2 ? c.A + ((size_t)S.rslot0 + i) * 2 : c.rhs + (size_t)(i / NB) * BS + 2 * (i % NB); };
  auto rhsV = [&](int i) -> double* { return BS == 2 ? c.A + HS + ((size_t)S.rslot0 + i) * 2 : c.rhs + (size_t)(i / NB) * BS + 2 * (i % NB) + 1; };
  auto SreP = [&](int i) -> double* { return BS == 2 ? c.A + ((size_t)S.rslot0 + i) * 2 + 1 : c.Sre + i; };
  auto SimP = [&](int i) -> double* { return BS == 2 ? c.A + HS + ((size_t)S.rslot0 + i) * 2 + 1 : c.Sim + i; };
  // V_i = e + j f of the Newton iterate.  Single-busbar layout: INTERLEAVED (e_i, f_i) pairs in the 2 nbus doubles that CarveP::e and
  // CarveP::f span (one ds_read_b128 per bus instead of two 8-byte reads in the pair / results phases; Gs and lab, which alias the
  // region, are dead by the time the Newton loop writes it); NB > 1: the two separate arrays.
  auto EF = [&](int i) -> double2 { return NB == 1 ? reinterpret_cast<const double2*>(c.e)[i] : make_double2(c.e[i], c.f[i]); };
  auto setEF = [&](int i, double e_, double f_) { if (NB == 1) reinterpret_cast<double2*>(c.e)[i] = make_double2(e_, f_); else { c.e[i] = e_; c.f[i] = f_; } };
  const auto topo_g = gptr(b.topo) + (size_t)inst * g.dim_topo;            // lane rows in HBM: explicit global address space
  const auto shb = gptr(b.shunt_bus) + (size_t)inst * g.n_shunt;
  n_iter_out = 0;
  nb_out = 0;
  c.out_l = nullptr;                                     // (set by the results phase when it stages the row in LDS)
  GPF_STAMPS(0);
  const auto inj_g = gptr(b.inj) + (size_t)inst * g.n_inj;
  if (STAGE && !ctl.inj_staged) {
    for (int i = tid; i < g.n_inj; i += GW) c.inj[i] = inj_g[i];
  }
  const bool reuse = ctl.reuse;
  const bool dc_kept = reuse && ctl.dcf && NB == 1;        // the factored DC matrix of the previous solve is still valid
  const bool warm = reuse && ctl.warm && !is_dc;            // block-uniform: skip the DC initialisation, keep va / |V| of PQ buses
  // Newton working set in registers (instance-group kernels, see the Newton loop below); block-uniform
#ifdef GPF_NO_NWR                                               /* developer A/B build (tools/build_worktree_variant.sh): general path only */
  constexpr bool NWR = false;
#else
  constexpr bool NWR = NB == 1 && WPI == 1 && !YR;             // every single-wavefront single-busbar kernel
#endif
  const bool nwr = NWR && !is_dc && nbus <= GW && S.n_up <= GW;
  // FUSED START of a step whose topology stands (reuse, K9 already accumulated the bus sums) on the reference topology of a small
  // grid (static DC inverse): the phases "initial |V|", "DC right-hand side", "theta = inv(B') P" and "Newton initialisation" only
  // exchange per-bus values that the bus lane itself produces and consumes -- they run as ONE phase inside the Newton
  // initialisation (the matrix-vector product reads Psp - Gs directly: the rows / columns of the reference buses of the static
  // inverse are exact unit vectors, so their right-hand-side entries do not matter).  Three phase boundaries and their LDS round
  // trips less per step.
  const bool fast_pre = nwr && reuse && ctl.sums_done && !warm && !TC && S.so.dc_inv != -1 && G::block_all_u(ts.dc_base && ts.dc_out < 0);
#define GPF_INJ(i_) (STAGE ? c.inj[(i_)] : (double)inj_g[(i_)])      /* staged row in LDS, else the lane's row in HBM / L2 */
  const double sn = g.sn_mva, inv_sn = g.inv_sn_mva;

  // ---- K1: element -> bus, bus activity / types / injections with LDS atomics from the element lanes ---------------------
  // (reuse: the maps and types of the previous solve stand, only the injection sums are rebuilt)
  if (!reuse && !ctl.topo_staged) for (int i = tid; i < g.dim_topo; i += GW) c.topo[i] = topo_g[i];
  const bool sums_done = reuse && ctl.sums_done;
  if (!sums_done) {
  for (int i = tid; i < nbus; i += GW) {
    if (!reuse) { c.btype[i] = BT_OFF; c.vidx[i] = -1; }
    c.Psp[i] = 0.0; c.Qsp[i] = 0.0; c.Gs[i] = 0.0;
  }
  if (NB == 1 && !TC && !reuse) for (int i = tid; i < nsub; i += GW) c.sub_bb[i] = 1;
  GPF_LSYNC();
  }
  GPF_STAMPS(27);
  const int* topo = c.topo;
  auto bus_of = [&](int sub, int local) -> int {
    if (TC) return sv.node_of[sub * g.n_busbar + (local - 1)];
    return (NB == 1) ? sub : sub * NB + (local - 1);
  };
  bool line_off = false, slack_off = false, gen_off = false;
  int n_line_off = 0, l_first_off = -1;                  // (single-wavefront instances: open lines of this group, the first of them)
  if (!sums_done) {
  if (!reuse)
  for (int l0 = 0; l0 < g.n_line; l0 += GW) {
    const int l = l0 + tid;
    const bool have = l < g.n_line;
    const bool off_l = have && !((topo[sv.line_or_pos[l]] >= 1) && (topo[sv.line_ex_pos[l]] >= 1));
    if (WPI == 1 && NB == 1 && !TC) {
      const int cnt = G::count(off_l);
      if (cnt) { if (l_first_off < 0) l_first_off = l0 + G::first(off_l); n_line_off += cnt; }
    }
    if (!have) continue;
    const int bo = topo[sv.line_or_pos[l]], be = topo[sv.line_ex_pos[l]];
    const bool on = (bo >= 1) && (be >= 1);
    line_off |= !on;
    const int so = sv.line_or_sub[l], se = sv.line_ex_sub[l];
    const int fo = on ? bus_of(so, bo) : -1, fe = on ? bus_of(se, be) : -1;
    c.lor_b[l] = (i16)fo;
    c.lex_b[l] = (i16)fe;
    if (on) {
      atomicMax(&c.btype[fo], BT_PQ);
      atomicMax(&c.btype[fe], BT_PQ);
      if (NB == 1 && !TC) { c.sub_bb[so] = (i8)bo; c.sub_bb[se] = (i8)be; }
    }
  }
  // (loads before generators: the order in which K9 adds the set-points of a step whose topology stands -- SolveCtl::sums_done --, so that a bus
  //  sum is the same floating-point number whichever phase accumulated it)
  if (acc_lane)
  for (int i = tid; i < g.n_load; i += GWA) {
    int bu;
    if (!reuse) {
      const int lb = topo[sv.load_pos[i]];
      const int sb = sv.load_sub[i];
      bu = lb >= 1 ? bus_of(sb, lb) : -1;
      c.load_b[i] = (i16)bu;
      if (bu >= 0) {
        atomicMax(&c.btype[bu], BT_PQ);
        if (NB == 1 && !TC) c.sub_bb[sb] = (i8)lb;
      }
    } else bu = c.load_b[i];
    if (bu >= 0) {
      atomicAdd(&c.Psp[bu], -GPF_INJ(oo.inj_load_p + i) * inv_sn);
      atomicAdd(&c.Qsp[bu], -GPF_INJ(oo.inj_load_q + i) * inv_sn);
    }
  }
  if (acc_lane)
  for (int i = tid; i < g.n_gen; i += GWA) {
    int bu;
    const bool sl = sv.gen_slack[i] != 0;
    if (!reuse) {
      const int lb = topo[sv.gen_pos[i]];
      const int sb = sv.gen_sub[i];
      bu = lb >= 1 ? bus_of(sb, lb) : -1;
      c.gen_b[i] = (i16)bu;
      slack_off |= sl && bu < 0;
      gen_off |= bu < 0;
      if (bu >= 0) {
        atomicMax(&c.btype[bu], sl ? BT_REF : BT_PV);
        atomicMax(&c.vidx[bu], i);
        if (NB == 1 && !TC) c.sub_bb[sb] = (i8)lb;
      }
    } else bu = c.gen_b[i];
    if (bu >= 0 && !sl) atomicAdd(&c.Psp[bu], GPF_INJ(oo.inj_gen_p + i) * inv_sn);
  }
  if (acc_lane)
  for (int i = tid; i < g.n_sto; i += GWA) {
    int bu;
    if (!reuse) {
      const int lb = topo[sv.sto_pos[i]];
      const int sb = sv.sto_sub[i];
      bu = lb >= 1 ? bus_of(sb, lb) : -1;
      c.sto_b[i] = (i16)bu;
      if (bu >= 0) {
        atomicMax(&c.btype[bu], BT_PQ);
        if (NB == 1 && !TC) c.sub_bb[sb] = (i8)lb;
      }
    } else bu = c.sto_b[i];
    if (bu >= 0) {
      atomicAdd(&c.Psp[bu], -GPF_INJ(oo.inj_sto_p + i) * inv_sn);
      atomicAdd(&c.Qsp[bu], -GPF_INJ(oo.inj_sto_q + i) * inv_sn);
    }
  }
  if (acc_lane)
  for (int i = tid; i < g.n_shunt; i += GWA) {
    int bu;
    if (!reuse) {
      const int lb = shb[i];
      const int sb = sv.shunt_sub[i];
      bu = lb >= 1 ? bus_of(sb, lb) : -1;
      c.sh_b[i] = (i16)bu;
      if (bu >= 0) {
        atomicMax(&c.btype[bu], BT_PQ);
        if (NB == 1 && !TC) c.sub_bb[sb] = (i8)lb;
      }
    } else bu = c.sh_b[i];
    if (bu >= 0) atomicAdd(&c.Gs[bu], GPF_INJ(oo.inj_sh_p + i) * sv.shunt_fact[i] * inv_sn);
  }
  }
  GPF_LSYNC();
  GPF_STAMPS(28);
  int nb = 0, nref = 0;
  if (!fast_pre)
  for (int i0 = 0; i0 < nbus; i0 += GW) {
    const int i = i0 + tid;
    const int bt = i < nbus ? c.btype[i] : BT_OFF;
    if (i < nbus) {
      const int vi = c.vidx[i];
      // initial |V|: set-point of the last in-service generator on PV / reference buses, 1 pu elsewhere
      const double vm_pq = warm ? c.vm[i] : 1.0;
      const bool has_sp = vi >= 0 && (bt == BT_PV || bt == BT_REF);
      if (!(WPI > 1 && ctl.inj_regs)) c.vm[i] = has_sp ? GPF_INJ(oo.inj_gen_vm + vi) : vm_pq;
      else if (!has_sp) c.vm[i] = vm_pq;                    // (the set-point buses are written by their generator's lane, below)
      if (!reuse) c.lab[i] = (bt == BT_REF) ? 1 : 0;
    }
    if (!reuse) {
      nb += G::count(bt != BT_OFF);
      nref += G::count(bt == BT_REF);
    }
  }
  if (WPI > 1 && ctl.inj_regs && tid < g.n_gen) {           // voltage set-point of generator tid from K9's register (SolveCtl::inj_regs)
    const int bu = c.gen_b[tid];
    if (bu >= 0 && c.vidx[bu] == tid) { const int bt = c.btype[bu]; if (bt == BT_PV || bt == BT_REF) c.vm[bu] = (double)ctl.r_vm; }
  }
  if (reuse) nb = ts.nb;
  nb_out = nb;
  if (!fast_pre) GPF_LSYNC();
  int status = reuse ? ts.status : ((nref == 0) ? 3 : 0);           // first failure of this group (0 = alive)
  if (!reuse) { ts.status = status; ts.nb = nb; }
  if (G::block_all_u(status != 0)) return status;
  GPF_STAMPS(1);

  // ---- connectivity ----------------------------------------------------------------------------------------------------
  // With one live busbar per substation and every line in service the bus graph IS the static substation graph, whose
  // connectivity the host checked at gpf_create: nothing to propagate (the DoNothing case).  Otherwise label propagation
  // from the reference buses.
  if (!reuse) {
  const unsigned off_bits = G::template any_bits<2, 3>((line_off ? 1u : 0u) | (slack_off ? 2u : 0u) | (gen_off ? 4u : 0u));
  ts.gen_base = NB == 1 && !TC && !(off_bits & 4u);
  // the DC matrix only depends on which lines are in service and where the reference buses are
  {
    const bool inv_ok = NB == 1 && !TC && S.so.dc_inv != -1 && !(off_bits & 2u);
    // exactly one line out, every substation still active, a real branch (two different substations): rank-1 correction
    bool one_out = false;
    if (WPI == 1 && inv_ok && n_line_off == 1 && nb == nsub) one_out = sv.line_or_sub[l_first_off] != sv.line_ex_sub[l_first_off];
    ts.dc_base = inv_ok && (WPI == 1 ? (n_line_off == 0 || one_out) : !(off_bits & 1u));
    ts.dc_out = (ts.dc_base && one_out) ? l_first_off : -1;
  }
  const bool conn_known = (NB == 1) && S.static_connected && !(off_bits & 1u);
  if (!G::block_all(conn_known))
  for (int sweep = 0; sweep < nbus; ++sweep) {
    int changed = 0;
    for (int l = tid; l < g.n_line; l += GW) {
      const int f = c.lor_b[l], t = c.lex_b[l];
      if (f >= 0) {
        const int lf = c.lab[f], lt = c.lab[t];
        if (lf != lt) { c.lab[f] = 1; c.lab[t] = 1; changed = 1; }
      }
    }
    GPF_LSYNC();
    if (!G::block_any(changed)) break;      // extra sweeps of a settled group are idempotent
  }
  {
    int bad = 0;
    if (!conn_known) for (int i = tid; i < nbus; i += GW) bad |= (c.btype[i] != BT_OFF && c.lab[i] == 0);
    if (status == 0 && G::any(bad)) status = 2;
    ts.status = status;
    if (G::block_all_u(status != 0)) return status;
  }
  }
  GPF_STAMPS(2);

  // ---- K2: block Ybus (original pattern) + K3: DC matrix in the block array, both with LDS atomics ----------------------------
  // (reuse: the Ybus blocks stand; the DC matrix is rebuilt unless its factors were kept)
  const bool do_y = !reuse && !is_dc;
  // reference topology on a small grid: theta = inv(B') P with the static inverse (one matrix-vector phase instead of assembling,
  // factoring and sweeping the DC system); takes precedence over kept factors.  Uniform over the block: a wavefront whose
  // groups differ takes the LU path for all of them.
  const bool dc_inv = NB == 1 && !TC && G::block_all_u(ts.dc_base);
  const bool dc_skip = dc_kept || dc_inv;              // no DC matrix to assemble
  // the static-inverse DC start runs inside the Newton initialisation of the register-resident path (see fast_pre): no right-hand
  // side phase, no separate matrix-vector phase -- also on a step that rebuilt its topology tables (one launch per step)
  const bool fuse_dc = nwr && !warm && dc_inv && G::block_all_u(ts.dc_out < 0);
  auto dcinv = [&](int idx) -> double { return S.so.dc_inv >= 0 ? sv.dc_inv[idx] : sv.dc_inv_g[idx]; };
  // operand pairs of the static-inverse matrix-vector product in flight per trip (the one-instance kernels live on 3 waves per SIMD
  // at <= 168 VGPRs: 4 pairs there)
#ifndef GPF_TUNE_MVC
#define GPF_TUNE_MVC 4        /* operand pairs in flight in the static-DC-inverse product of the one-instance kernels (-D: experiments) */
#endif
  constexpr int MVC = IPW > 1 ? 8 : GPF_TUNE_MVC;
  auto lidx = [&](int bus) -> int { return (NB == 1) ? 0 : bus % NB; };
  if (!warm && !fast_pre) {
  if (!dc_skip || do_y) {
  // YR: the Ybus blocks are assembled in the (still unused) row-1 half of the block array and then moved to registers
  double* const ydst = YR ? c.A + HS : c.Yb;
  if (!dc_skip) for (int i = tid; i < (NB == 1 ? S.nslot : S.nslot_lu) * B2; i += GW) c.A[i] = 0.0;
  if (do_y) for (int i = tid; i < S.nslot_y * NB * NB * 2; i += GW) ydst[i] = 0.0;
  GPF_LSYNC();
  if (acc_lane)
  for (int l = tid; l < g.n_line; l += GWA) {
    const int f = c.lor_b[l], t = c.lex_b[l];
    if (f < 0) continue;
    const int bi = lidx(f), bj = lidx(t);
    const int sff = sv.br_slot[4 * l + 0], sft = sv.br_slot[4 * l + 1], stf = sv.br_slot[4 * l + 2], stt = sv.br_slot[4 * l + 3];
    if (do_y) {
      const double4 ya = sv.br_y.ld4((size_t)8 * l), yb = sv.br_y.ld4((size_t)8 * l + 4);
      double* y;
      y = ydst + ((size_t)sff * NB * NB + bi * NB + bi) * 2; atomicAdd(&y[0], ya.x); atomicAdd(&y[1], ya.y);
      y = ydst + ((size_t)sft * NB * NB + bi * NB + bj) * 2; atomicAdd(&y[0], ya.z); atomicAdd(&y[1], ya.w);
      y = ydst + ((size_t)stf * NB * NB + bj * NB + bi) * 2; atomicAdd(&y[0], yb.x); atomicAdd(&y[1], yb.y);
      y = ydst + ((size_t)stt * NB * NB + bj * NB + bj) * 2; atomicAdd(&y[0], yb.z); atomicAdd(&y[1], yb.w);
    }
    if (f != t && !dc_skip) {
      const double bb = sv.br_bdc[l];
      const bool ff_ = c.btype[f] != BT_REF, tf_ = c.btype[t] != BT_REF;     // theta row / column live?
      const int rf = 2 * bi, rt = 2 * bj;
      if (ff_) atomicAdd(bel(sff, rf, rf), bb);
      if (tf_) atomicAdd(bel(stt, rt, rt), bb);
      if (ff_ && tf_) {
        atomicAdd(bel(sft, rf, rt), -bb);
        atomicAdd(bel(stf, rt, rf), -bb);
      }
    }
  }
  if (do_y && acc_lane) {
    for (int s = tid; s < g.n_shunt; s += GWA) {
      const int bu = c.sh_b[s];
      if (bu >= 0) {
        const int bi = lidx(bu);
        const int sub = (NB == 1) ? bu : bu / NB;
        const double fct = sv.shunt_fact[s] * inv_sn;
        double* y = ydst + ((size_t)sub * NB * NB + bi * NB + bi) * 2;       // diag slot of a substation == its index
        atomicAdd(&y[0], GPF_INJ(oo.inj_sh_p + s) * fct);
        atomicAdd(&y[1], -GPF_INJ(oo.inj_sh_q + s) * fct);
      }
    }
  }
  GPF_LSYNC();
  if (YR && do_y) {
#pragma unroll
    for (int k = 0; k < YR_PASSES; ++k) {
      const int pr = tid + k * GW;
      const unsigned w0 = pr < S.n_up ? (unsigned)sv.up[2 * pr] : 0u, w1 = pr < S.n_up ? (unsigned)sv.up[2 * pr + 1] : 0u;
      rcreg[2 * k] = w0; rcreg[2 * k + 1] = w1;
      yreg[2 * k] = pr < S.n_up ? *reinterpret_cast<const double2*>(ydst + (size_t)(w1 & 0xffffu) * 2) : make_double2(0.0, 0.0);
      yreg[2 * k + 1] = pr < S.n_up ? *reinterpret_cast<const double2*>(ydst + (size_t)(w1 >> 16) * 2) : make_double2(0.0, 0.0);
    }
    yreg[2 * YR_PASSES] = tid < nbus ? *reinterpret_cast<const double2*>(ydst + (size_t)tid * 2) : make_double2(0.0, 0.0);   // diagonal block of bus tid
    GPF_LSYNC();                                   // the identity rows below overwrite the scratch
  }
  }
  // identity rows (fixed variables) + DC right-hand side
  if (!fuse_dc)
  for (int i = tid; i < nbus; i += GW) {
    const int sub = (NB == 1) ? i : i / NB, bi = lidx(i);
    const int bt = c.btype[i];
    const bool th_live = (bt == BT_PQ || bt == BT_PV);
    if (!dc_skip) {
      if (!th_live) *bel(sub, 2 * bi, 2 * bi) = 1.0;
      *bel(sub, 2 * bi + 1, 2 * bi + 1) = 1.0;                   // |V| rows are identity in the DC system
    }
    *rhsT(i) = th_live ? (c.Psp[i] - c.Gs[i]) : 0.0;
    *rhsV(i) = 0.0;
  }
  if (!fuse_dc) GPF_LSYNC();
  }
  GPF_STAMPS(3);
  // the program is in LDS (tier >= 1) or read in place through a global-address-space pointer (tier 0)
  // Instance-group kernels (IPW > 1: the small grids) stream the flat program from global memory (L1 / L2 resident, a few KB shared
  // by every block), one pass ahead, although their other static tables are staged in LDS: item words that come back through the LDS
  // share its in-order counter with the pass's atomics, and the wait for them at the end of a pass (s_waitcnt lgkmcnt(0): the two
  // paths of the "valid item?" branch issue different numbers of LDS operations, so the compiler cannot count) drains the atomics of
  // every pass; global loads wait on vmcnt and leave the LDS queue alone (+1.2 % on 14 substations, and 2.8 KB of LDS per block).
  // One instance per wavefront (36+ substations): the passes are too short to hide an L2 round trip per pass -- measured -13 % at
  // 4 096 lanes -- so a staged program (tier >= 1) is read from LDS there.
  const int* const flat_g = STAGE == 0 ? sv.prog.p : S.flat[gw_index(GW)];     // (tier 0: the view already points at the global copy)
  constexpr bool PROG_LDS = STAGE >= 1 && IPW == 1;
  auto lu_ac = [&](long long* dbg) -> bool {
    if (BS == 2) {
      if (PROG_LDS) return block_lu_flat<GW>(FL, sv.prog.p, c.A, HS, tid, dbg);
      return block_lu_flat<GW>(FL, gptr(flat_g), c.A, HS, tid, dbg);
    }
    if (STAGE >= 1) return block_lu_solve<BS, GW>(S, sv.prog.p, c.A, c.rhs, tid, dbg);
    return block_lu_solve<BS, GW>(S, gptr(sv.prog.p), c.A, c.rhs, tid, dbg);
  };
  auto lu_dc = [&](long long* dbg) -> bool {       // single-busbar layout only
    if (dc_kept) {
      if (PROG_LDS) return scalar_lu_flat<GW, false, true>(FL, sv.prog.p, c.A, c.Adc, tid, dbg);
      return scalar_lu_flat<GW, false, true>(FL, gptr(flat_g), c.A, c.Adc, tid, dbg);
    }
    if (PROG_LDS) return scalar_lu_flat<GW, true, false>(FL, sv.prog.p, c.A, c.A, tid, dbg);
    return scalar_lu_flat<GW, true, false>(FL, gptr(flat_g), c.A, c.A, tid, dbg);
  };
  if (!warm && !fuse_dc) {
#ifdef GPF_TIMING
    bool ok = dc_inv ? true : (NB == 1) ? lu_dc(&stamps.v[20])
                                        : lu_ac(&stamps.v[20]);
#else
    bool ok = dc_inv ? true : (NB == 1) ? lu_dc(nullptr) : lu_ac(nullptr);
#endif
    for (int i = tid; i < nbus; i += GW) {
      double th;
      if (dc_inv) {                                                 // row i of inv(B') (column-major table) times the right-hand side
        // (the table is in the LDS-staged blob or in global memory: ONE branch around the whole product, not one per element -- a
        //  select between two address spaces per operand made every load wait for itself: 3 000 instead of 500 cycles on 14 substations)
        auto row_times_rhs = [&](const auto tab) -> double {
          double t0 = 0.0, t1 = 0.0;                                // two chains (even / odd k), MVC operand pairs in flight per trip
          for (int k0 = 0; k0 < nbus; k0 += MVC) {
            double a_[MVC], r_[MVC];
#pragma unroll
            for (int q = 0; q < MVC; ++q) {
              const int k = k0 + q < nbus ? k0 + q : nbus - 1;
              a_[q] = tab[k * nbus + i];
              r_[q] = *rhsT(k);
            }
#pragma unroll
            for (int q = 0; q < MVC; q += 2) {
              if (k0 + q < nbus) t0 = fma(a_[q], r_[q], t0);
              if (k0 + q + 1 < nbus) t1 = fma(a_[q + 1], r_[q + 1], t1);
            }
          }
          return t0 + t1;
        };
        th = S.so.dc_inv >= 0 ? row_times_rhs(sv.dc_inv) : row_times_rhs(sv.dc_inv_g);
      } else if (NB == 1) {                                         // flat sweeps leave s_i = d_i theta_i (scalar_lu_flat)
        const double d = dc_kept ? c.Adc[i] : c.A[(size_t)i * 2];
        if (!(fabs(d) > 1e-300) || !(fabs(d) < 1e300)) ok = false;
        th = *rhsT(i) * fast_rcp(d);
      } else th = *rhsT(i);
      const int bt = c.btype[i];
      c.va[i] = (bt == BT_PQ || bt == BT_PV) ? th : 0.0;
      if (bt != BT_OFF && !(fabs(th) < 1e300)) ok = false;
    }
    if (NB == 1 && ctl.dcf && !dc_skip)                            // keep the factors for the next solves of this launch
      for (int q = tid; q < S.nslot; q += GW) c.Adc[q] = c.A[(size_t)q * 2];
    GPF_LSYNC();
    if (WPI == 1 && dc_inv && G::block_any_u(ts.dc_out >= 0)) {
      // ONE line l = (f, t) out: B'_c = B' - b_l a a^T (a = e_f - e_t over the non-reference buses), so by Sherman-Morrison
      //   theta_c = theta_0 + X a * b_l (a^T theta_0) / (1 - b_l a^T X a),   X = inv(B') = the static table
      // -- two more reads of the table per bus instead of assembling, factoring and sweeping the DC system of the contingency
      // (the N-1 fan-out: every lane but the intact one of each environment).  c.va holds theta_0 (0 at reference buses).
      const int lo_ = ts.dc_out >= 0 ? ts.dc_out : 0;
      const int f_ = sv.line_or_sub[lo_], t_ = sv.line_ex_sub[lo_];
      const double bl = ts.dc_out >= 0 ? sv.br_bdc[lo_] : 0.0;
      const bool f_ref = c.btype[f_] == BT_REF, t_ref = c.btype[t_] == BT_REF;
      const double dth = c.va[f_] - c.va[t_];
      const double xff = f_ref ? 0.0 : dcinv(f_ * nbus + f_), xtt = t_ref ? 0.0 : dcinv(t_ * nbus + t_);
      const double xft = (f_ref || t_ref) ? 0.0 : dcinv(t_ * nbus + f_);
      const double den = 1.0 - bl * (xff - 2.0 * xft + xtt);
      if (!(fabs(den) > 1e-12)) ok = false;
      const double kk = bl * dth / den;
      for (int i = tid; i < nbus; i += GW) {
        const int bt = c.btype[i];
        const double u_ = (f_ref ? 0.0 : dcinv(f_ * nbus + i)) - (t_ref ? 0.0 : dcinv(t_ * nbus + i));
        if (bt == BT_PQ || bt == BT_PV) c.va[i] += u_ * kk;        // (every lane read c.va[f], c.va[t] above: one wavefront, in order)
      }
      GPF_LSYNC();
    }
    if (status == 0 && G::any(!ok)) status = 4;
    if (G::block_all_u(status != 0)) return status;
  }
  GPF_STAMPS(4);

  int it = 0;
  if (!is_dc) {
    bool converged = false;
    bool done = status != 0;                  // this group takes no further Newton steps (state frozen)
    const int n_pairs = S.nslot_y * NB * NB;
    // T = V_i conj(Y V_j) and the masked block [dP/dth dP/dV; dQ/dth dQ/dV] = [Im T, Re T/|Vj|; -Re T, Im T/|Vj|] of (i, j)
    auto t_of = [&](const double2 y, double ei, double fi, double ej, double fj, double& tr_, double& ti_) {
      const double aa = y.x * ej - y.y * fj, bb = y.x * fj + y.y * ej;
      tr_ = ei * aa + fi * bb;
      ti_ = fi * aa - ei * bb;
    };
    // NEWTON WORKING SET IN REGISTERS (single-wavefront kernels, round 4).  When every bus and every undirected pair of the instance has
    // its own lane (nbus <= GW, n_up <= GW: the 5-, 14- and 36-substation grids, and the topology classes of theirs that fit), what a lane needs in
    // all iterations of the loop is loaded ONCE per solve: the pair lane's table words, its two Ybus blocks and the bus-type masks of
    // its two ends; the bus lane's type, diagonal Ybus block, specified injections and -- carried from phase to phase instead of
    // going through LDS -- va, |V|, e, f, 1/|V| and S_i.  A pair phase is then ONE LDS round trip (V of the two ends) instead of two
    // dependent ones (table words -> operands), the mismatch phase reads only S_i, and ~35 % of the VALU instructions of the three
    // single-phase steps (address arithmetic, type compares, loop / exec-mask handling) are gone.  Same arithmetic in the same order
    // as the general path below: results are bit-identical.
    if (nwr) {
      const bool p_on = tid < S.n_up, b_on = tid < nbus;
      const int ib = b_on ? tid : 0;
      const unsigned w0 = p_on ? (unsigned)sv.up[2 * tid] : 0u, w1 = p_on ? (unsigned)sv.up[2 * tid + 1] : 0u;
      const int u = (int)(w0 & 0xffffu), v = (int)(w0 >> 16), suv = (int)(w1 & 0xffffu), svu = (int)(w1 >> 16);
      // (one instance per wavefront -- the 36-substation kernels, 3 waves per SIMD at <= 168 VGPRs -- has no registers for the Ybus
      //  blocks and the specified injections: they are re-read from LDS where they are used)
      constexpr bool KEEP_Y = IPW > 1;
      const double2* const p_yuv = reinterpret_cast<const double2*>(c.Yb + (size_t)suv * 2);
      const double2* const p_yvu = reinterpret_cast<const double2*>(c.Yb + (size_t)svu * 2);
      const double2* const p_yd = reinterpret_cast<const double2*>(c.Yb + (size_t)ib * 2);
      const double2 yuv0 = *p_yuv, yvu0 = *p_yvu, ydiag0 = *p_yd;
      const int btu = c.btype[u], btv = c.btype[v];
      const int bt = c.btype[ib];
      const double psp0 = c.Psp[ib], qsp0 = c.Qsp[ib];
#define NWR_YUV (KEEP_Y ? yuv0 : *p_yuv)
#define NWR_YVU (KEEP_Y ? yvu0 : *p_yvu)
#define NWR_YD (KEEP_Y ? ydiag0 : *p_yd)
#define NWR_PSP (KEEP_Y ? psp0 : c.Psp[ib])
#define NWR_QSP (KEEP_Y ? qsp0 : c.Qsp[ib])
      double va, vm;
#ifdef GPF_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      GPF_STAMPS(20);                 // (developer timing build: the solve's constants are in registers)
#endif
      if (fuse_dc) {
        // initial |V| (K1; a step that rebuilt its tables already has it in LDS), DC right-hand side (K3) and theta = inv(B') (P - G)
        // in the bus lane itself; row ib of the column-major static inverse, all operands in flight before the two FMA chains
        // (even / odd k, the order of the general path)
        const int vi = fast_pre ? c.vidx[ib] : -1;
        const double vm_lds = fast_pre ? 1.0 : c.vm[ib];
        // r_k = Psp_k - Gs_k of every bus goes through ONE contiguous array (c.ivm: dead until the sincos step below rewrites it), so that the
        // product reads one operand per term instead of two; the table is SYMMETRIC (gpf_create symmetrises it), so a lane may walk its
        // row k-contiguously when the table is staged in LDS, and column by column (coalesced across the lanes) when it is read through L2
        if (b_on) c.ivm[ib] = psp0 - c.Gs[ib];
        GPF_LSYNC();
        auto row_times_rhs = [&](const auto tab, const bool by_row) -> double {
          double t0 = 0.0, t1 = 0.0;                                // two chains (even / odd k), MVC operand pairs in flight per trip
          for (int k0 = 0; k0 < nbus; k0 += MVC) {
            double a_[MVC], r_[MVC];
#pragma unroll
            for (int q = 0; q < MVC; ++q) {
              const int k = k0 + q < nbus ? k0 + q : nbus - 1;
              a_[q] = by_row ? tab[ib * nbus + k] : tab[k * nbus + ib];
              r_[q] = c.ivm[k];
            }
#pragma unroll
            for (int q = 0; q < MVC; q += 2) {
              if (k0 + q < nbus) t0 = fma(a_[q], r_[q], t0);
              if (k0 + q + 1 < nbus) t1 = fma(a_[q + 1], r_[q + 1], t1);
            }
          }
          return t0 + t1;
        };
        const double th = S.so.dc_inv >= 0 ? row_times_rhs(sv.dc_inv, STAGE == 2) : row_times_rhs(sv.dc_inv_g, false);
        const bool live = (bt == BT_PQ || bt == BT_PV);
        va = live ? th : 0.0;
        vm = (vi >= 0 && (bt == BT_PV || bt == BT_REF)) ? GPF_INJ(oo.inj_gen_vm + vi) : vm_lds;
        const bool th_bad = b_on && bt != BT_OFF && !(fabs(th) < 1e300);
        if (status == 0 && G::any(th_bad)) { status = 4; done = true; }
      } else { va = c.va[ib]; vm = c.vm[ib]; }
#ifdef GPF_TIMING
      GPF_STAMPS(21);                 // (developer timing build: DC start done)
#endif
      const bool act = p_on && (btu != BT_OFF) && (btv != BT_OFF);
      const bool uP = (btu == BT_PQ || btu == BT_PV), uQ = (btu == BT_PQ), vP = (btv == BT_PQ || btv == BT_PV), vQ = (btv == BT_PQ);
      const bool acc_u = act && (yuv0.x != 0.0 || yuv0.y != 0.0), acc_v = act && (yvu0.x != 0.0 || yvu0.y != 0.0);
      const bool rowP = (bt == BT_PQ || bt == BT_PV), rowQ = (bt == BT_PQ);
      const bool diag_on = bt != BT_OFF && (ydiag0.x != 0.0 || ydiag0.y != 0.0);
      double* const Ad0 = bel(ib, 0, 0);
      double* const Ad1 = bel(ib, 1, 0);
      double* const b_uv0 = bel(suv, 0, 0); double* const b_uv1 = bel(suv, 1, 0);
      double* const b_vu0 = bel(svu, 0, 0); double* const b_vu1 = bel(svu, 1, 0);
      double e, f, ivm, Sr = 0.0, Si = 0.0;
      {
        double sn_, co;
        fast_sincos(va, sn_, co);
        e = vm * co; f = vm * sn_; ivm = fast_rcp(vm);
        if (b_on) { setEF(ib, e, f); c.ivm[ib] = ivm; *SreP(ib) = 0.0; *SimP(ib) = 0.0; }
      }
      for (int i = S.nslot_y * 2 + tid; i < S.nslot * 2; i += GW) { c.A[i] = 0.0; c.A[HS + i] = 0.0; }      // fill blocks start at zero
      GPF_LSYNC();
      GPF_STAMPS(10);
      while (true) {
        // ---- pair phase: T_uv, T_vu -> the two off-diagonal Jacobian blocks, S_u += T_uv, S_v += T_vu
        if (p_on) {
          const double2 efu = EF(u), efv = EF(v);
          const double ivmu = c.ivm[u], ivmv = c.ivm[v];
          double tr_, ti_, sr_, si_;
          const double2 yuv = NWR_YUV, yvu = NWR_YVU;
          t_of(yuv, efu.x, efu.y, efv.x, efv.y, tr_, ti_);
          t_of(yvu, efv.x, efv.y, efu.x, efu.y, sr_, si_);
          *reinterpret_cast<double2*>(b_uv0) = make_double2((uP && vP) ? ti_ : 0.0, (uP && vQ) ? tr_ * ivmv : 0.0);
          *reinterpret_cast<double2*>(b_uv1) = make_double2((uQ && vP) ? -tr_ : 0.0, (uQ && vQ) ? ti_ * ivmv : 0.0);
          *reinterpret_cast<double2*>(b_vu0) = make_double2((vP && uP) ? si_ : 0.0, (vP && uQ) ? sr_ * ivmu : 0.0);
          *reinterpret_cast<double2*>(b_vu1) = make_double2((vQ && uP) ? -sr_ : 0.0, (vQ && uQ) ? si_ * ivmu : 0.0);
          if (acc_u) { atomicAdd(SreP(u), tr_); atomicAdd(SimP(u), ti_); }
          if (acc_v) { atomicAdd(SreP(v), sr_); atomicAdd(SimP(v), si_); }
        }
        GPF_LSYNC();
        if (it == 0) GPF_STAMPS(11);
        // ---- mismatch phase: diagonal block (i, i), right-hand side, convergence test
        double fabs_mis = 0.0;
        bool bad = false;
        if (b_on) {
          Sr = *SreP(ib); Si = *SimP(ib);
          const double2 ydiag = NWR_YD;
          const double psp = NWR_PSP, qsp = NWR_QSP;
          double tr_, ti_;
          t_of(ydiag, e, f, e, f, tr_, ti_);
          if (diag_on) { Sr += tr_; Si += ti_; }
          const double2 r0 = make_double2(rowP ? ti_ : 0.0, (rowP && rowQ) ? tr_ * ivm : 0.0);
          const double2 r1 = make_double2((rowQ && rowP) ? -tr_ : 0.0, rowQ ? ti_ * ivm : 0.0);
          *reinterpret_cast<double2*>(Ad0) = make_double2(rowP ? r0.x - Si : 1.0, rowQ ? fma(Sr, ivm, r0.y) : r0.y);
          *reinterpret_cast<double2*>(Ad1) = make_double2(rowQ ? r1.x + Sr : r1.x, rowQ ? fma(Si, ivm, r1.y) : 1.0);
          const double mp = rowP ? (Sr - psp) : 0.0;
          const double mq = rowQ ? (Si - qsp) : 0.0;
          *rhsT(ib) = -mp; *rhsV(ib) = -mq;
          const double am = fmax(fabs(mp), fabs(mq));
          if (!(am <= 1e300)) bad = true;
          fabs_mis = am;
        }
        if (!done) {
          const unsigned fl = G::template any2<0>(!(fabs_mis < tol_pu), bad);
          if (fl & 2u) { status = 1; done = true; }
          else if (!(fl & 1u)) { converged = true; done = true; }
          else if (it >= max_iter) done = true;
          else ++it;
        }
        if (G::block_all_u(done)) break;
        GPF_LSYNC();
        if (it == 1) GPF_STAMPS(12);
        const bool ok = lu_ac(nullptr);
        if (it == 1) GPF_STAMPS(13);
        // ---- update (groups that are done keep their state) + preparation of the next pair phase
        bool fin = true, piv_ok = true;
        if (b_on) {
          double2 dx = make_double2(*rhsT(ib), *rhsV(ib));
          const double2 dA = *reinterpret_cast<const double2*>(Ad0), dB = *reinterpret_cast<const double2*>(Ad1);
          const double det = fma(dA.x, dB.y, -dA.y * dB.x);
          if (!(fabs(det) > 1e-300) || !(fabs(det) < 1e300)) piv_ok = false;
          const double rd = fast_rcp(det);
          dx = make_double2(fma(dB.y, dx.x, -dA.y * dx.y) * rd, fma(dA.x, dx.y, -dB.x * dx.x) * rd);
          if (!done && bt != BT_OFF) {
            if (!(fabs(dx.x) < 1e300) || !(fabs(dx.y) < 1e300)) fin = false;
            if (bt == BT_PQ || bt == BT_PV) va += dx.x;
            if (bt == BT_PQ) vm += dx.y;
            if (vm < 0.0) { vm = -vm; va += 3.14159265358979323846; }
            if (fabs(va) > 3.14159265358979323846) va = remainder(va, 6.28318530717958647692);
          }
          double sn_, co;
          fast_sincos(va, sn_, co);
          e = vm * co; f = vm * sn_; ivm = fast_rcp(vm);
          setEF(ib, e, f);
          c.ivm[ib] = ivm;
          *SreP(ib) = 0.0;
          *SimP(ib) = 0.0;
        }
        for (int i = S.nslot_y * 2 + tid; i < S.nslot * 2; i += GW) { c.A[i] = 0.0; c.A[HS + i] = 0.0; }
        GPF_LSYNC();
        if (!done && G::template any2<1>(!ok || !piv_ok, !fin) != 0u) { status = 4; done = true; }
        if (it == 1) GPF_STAMPS(14);
      }
      // what the results phase reads from LDS: the bus injections of the final state, va, |V|
      if (b_on) { *SreP(ib) = Sr; *SimP(ib) = Si; c.va[ib] = va; c.vm[ib] = vm; }
#undef NWR_YUV
#undef NWR_YVU
#undef NWR_YD
#undef NWR_PSP
#undef NWR_QSP
    } else {
      // Every phase of the loop is "issue all LDS reads -> compute -> write": the latency of a phase is a chain of dependent
      // LDS round trips, so read-modify-write sequences inside branches are avoided.  V = e + jf, S = 0 and zeroed fill
      // blocks are prepared by the phase BEFORE the pair phase (here for the first iteration, then by the update phase).
      for (int i = tid; i < nbus; i += GW) {
        const double va = c.va[i], vmi = c.vm[i];
        double sn_, co;
        fast_sincos(va, sn_, co);
        setEF(i, vmi * co, vmi * sn_);
        if (NB == 1) c.ivm[i] = fast_rcp(vmi);
        *SreP(i) = 0.0;
        *SimP(i) = 0.0;
        if (WPI > 1) { *rhsT(i) = 0.0; *rhsV(i) = 0.0; }      // wavefront 1's partial sums of S (see acc_lane above)
      }
      if (BS == 2) { for (int i = S.nslot_y * 2 + tid; i < S.nslot * 2; i += GW) { c.A[i] = 0.0; c.A[HS + i] = 0.0; } }
      else for (int i = S.nslot_y * B2 + tid; i < S.nslot_lu * B2; i += GW) c.A[i] = 0.0;     // fill blocks start at zero
      GPF_LSYNC();
      GPF_STAMPS(10);
      // Single-busbar layout: ONE lane per undirected pair (u, v) of connected substations computes both Jacobian blocks (u, v) and
      // (v, u) -- they share every operand but the Ybus block --, and the diagonal block of bus i is built by bus i's lane in the
      // mismatch phase (which needs S_i anyway): half the passes of one lane per block of the original pattern.
      // Tier 0 (tables in global memory): the words of the first pass stay in registers for the whole Newton loop and those of pass
      // k + 1 are fetched before pass k computes -- an L2 round trip per pass is otherwise the longest link of the phase.
      unsigned rc_first = 0, rc_first1 = 0;
      if (STAGE == 0 && !YR) {
        if (NB == 1) { if (tid < S.n_up) { rc_first = (unsigned)sv.up[2 * tid]; rc_first1 = (unsigned)sv.up[2 * tid + 1]; } }
        else if (tid < n_pairs) rc_first = (unsigned)sv.pair_rc[tid / (NB * NB)];
      }
      while (true) {
        // Jacobian blocks from the Ybus blocks: T_ij = V_i conj(Y_ij V_j); S_i += T_ij (LDS atomics)
        if (NB == 1) {
          auto upair_item = [&](const unsigned w0, const unsigned w1, const double2 yuv, const double2 yvu) {
            const int u = (int)(w0 & 0xffffu), v = (int)(w0 >> 16), suv = (int)(w1 & 0xffffu), svu = (int)(w1 >> 16);
            const int btu = c.btype[u], btv = c.btype[v];
            const double2 efu = EF(u), efv = EF(v);
            const double eu = efu.x, fu = efu.y, ivmu = c.ivm[u], ev = efv.x, fv = efv.y, ivmv = c.ivm[v];
            double tr_, ti_, sr_, si_;
            t_of(yuv, eu, fu, ev, fv, tr_, ti_);
            t_of(yvu, ev, fv, eu, fu, sr_, si_);
            const bool act = (btu != BT_OFF) && (btv != BT_OFF);
            const bool uP = (btu == BT_PQ || btu == BT_PV), uQ = (btu == BT_PQ), vP = (btv == BT_PQ || btv == BT_PV), vQ = (btv == BT_PQ);
            *reinterpret_cast<double2*>(bel(suv, 0, 0)) = make_double2((uP && vP) ? ti_ : 0.0, (uP && vQ) ? tr_ * ivmv : 0.0);
            *reinterpret_cast<double2*>(bel(suv, 1, 0)) = make_double2((uQ && vP) ? -tr_ : 0.0, (uQ && vQ) ? ti_ * ivmv : 0.0);
            *reinterpret_cast<double2*>(bel(svu, 0, 0)) = make_double2((vP && uP) ? si_ : 0.0, (vP && uQ) ? sr_ * ivmu : 0.0);
            *reinterpret_cast<double2*>(bel(svu, 1, 0)) = make_double2((vQ && uP) ? -sr_ : 0.0, (vQ && uQ) ? si_ * ivmu : 0.0);
            if (act && (yuv.x != 0.0 || yuv.y != 0.0)) { atomicAdd(wave1 ? rhsT(u) : SreP(u), tr_); atomicAdd(wave1 ? rhsV(u) : SimP(u), ti_); }
            if (act && (yvu.x != 0.0 || yvu.y != 0.0)) { atomicAdd(wave1 ? rhsT(v) : SreP(v), sr_); atomicAdd(wave1 ? rhsV(v) : SimP(v), si_); }
          };
          if (YR && S.n_up <= 2 * GW) {
            // at most two pairs per lane (the 118-substation grids: 179 pairs on 128 lanes): the bus values of BOTH pairs are requested
            // before the first pair computes -- one LDS round trip for the phase instead of two (the second trip used to queue behind
            // the first pair's block stores and atomics); same arithmetic, same order of the atomics
            struct PairOps { int btu, btv; double2 efu, efv; double ivmu, ivmv; };
            auto pair_load = [&](const unsigned w0) -> PairOps {
              const int u = (int)(w0 & 0xffffu), v = (int)(w0 >> 16);
              PairOps o;
              o.btu = c.btype[u]; o.btv = c.btype[v]; o.efu = EF(u); o.efv = EF(v); o.ivmu = c.ivm[u]; o.ivmv = c.ivm[v];
              return o;
            };
            auto pair_fin = [&](const PairOps& o, const unsigned w0, const unsigned w1, const double2 yuv, const double2 yvu) {
              const int u = (int)(w0 & 0xffffu), v = (int)(w0 >> 16), suv = (int)(w1 & 0xffffu), svu = (int)(w1 >> 16);
              const int btu = o.btu, btv = o.btv;
              const double eu = o.efu.x, fu = o.efu.y, ivmu = o.ivmu, ev = o.efv.x, fv = o.efv.y, ivmv = o.ivmv;
              double tr_, ti_, sr_, si_;
              t_of(yuv, eu, fu, ev, fv, tr_, ti_);
              t_of(yvu, ev, fv, eu, fu, sr_, si_);
              const bool act = (btu != BT_OFF) && (btv != BT_OFF);
              const bool uP = (btu == BT_PQ || btu == BT_PV), uQ = (btu == BT_PQ), vP = (btv == BT_PQ || btv == BT_PV), vQ = (btv == BT_PQ);
              *reinterpret_cast<double2*>(bel(suv, 0, 0)) = make_double2((uP && vP) ? ti_ : 0.0, (uP && vQ) ? tr_ * ivmv : 0.0);
              *reinterpret_cast<double2*>(bel(suv, 1, 0)) = make_double2((uQ && vP) ? -tr_ : 0.0, (uQ && vQ) ? ti_ * ivmv : 0.0);
              *reinterpret_cast<double2*>(bel(svu, 0, 0)) = make_double2((vP && uP) ? si_ : 0.0, (vP && uQ) ? sr_ * ivmu : 0.0);
              *reinterpret_cast<double2*>(bel(svu, 1, 0)) = make_double2((vQ && uP) ? -sr_ : 0.0, (vQ && uQ) ? si_ * ivmu : 0.0);
              if (act && (yuv.x != 0.0 || yuv.y != 0.0)) { atomicAdd(wave1 ? rhsT(u) : SreP(u), tr_); atomicAdd(wave1 ? rhsV(u) : SimP(u), ti_); }
              if (act && (yvu.x != 0.0 || yvu.y != 0.0)) { atomicAdd(wave1 ? rhsT(v) : SreP(v), sr_); atomicAdd(wave1 ? rhsV(v) : SimP(v), si_); }
            };
            const bool on0 = tid < S.n_up, on1 = tid + GW < S.n_up;
            const PairOps o0 = pair_load(rcreg[0]), o1 = pair_load(rcreg[2]);         // (lanes without a pair: words 0 -> bus 0, harmless reads)
            if (on0) pair_fin(o0, rcreg[0], rcreg[1], yreg[0], yreg[1]);
            if (on1) pair_fin(o1, rcreg[2], rcreg[3], yreg[2], yreg[3]);
          } else if (YR) {
  #pragma unroll
            for (int k = 0; k < YR_PASSES; ++k) if (tid + k * GW < S.n_up) upair_item(rcreg[2 * k], rcreg[2 * k + 1], yreg[2 * k], yreg[2 * k + 1]);
          } else {
            unsigned p0 = rc_first, p1 = rc_first1;
            for (int k = tid; k < S.n_up; k += GW) {
              unsigned w0, w1;
              if (STAGE == 0) { w0 = p0; w1 = p1; if (k + GW < S.n_up) { p0 = (unsigned)sv.up[2 * (k + GW)]; p1 = (unsigned)sv.up[2 * (k + GW) + 1]; } }
              else { w0 = (unsigned)sv.up[2 * k]; w1 = (unsigned)sv.up[2 * k + 1]; }
              upair_item(w0, w1, *reinterpret_cast<const double2*>(c.Yb + (size_t)(w1 & 0xffffu) * 2),
                         *reinterpret_cast<const double2*>(c.Yb + (size_t)(w1 >> 16) * 2));
            }
          }
        } else {
          auto pair_item = [&](int pr, const double2 y, const unsigned rc) {
            const int slot = pr / (NB * NB), bi = (pr / NB) % NB, bj = pr % NB;
            const int si = (int)(rc & 0xffffu), sj = (int)(rc >> 16);
            const int i = si * NB + bi, j = sj * NB + bj;
            const int bti = c.btype[i], btj = c.btype[j];
            const double2 efi = EF(i), efj = EF(j);
            const double ei = efi.x, fi = efi.y, ej = efj.x, fj = efj.y, vmj = c.vm[j];
            double tr_, ti_;
            t_of(y, ei, fi, ej, fj, tr_, ti_);
            const bool act = (bti != BT_OFF) && (btj != BT_OFF);
            const bool rowP = (bti == BT_PQ || bti == BT_PV), rowQ = (bti == BT_PQ);
            const bool colT = (btj == BT_PQ || btj == BT_PV), colV = (btj == BT_PQ);
            const double ivmj = fast_rcp(vmj);
            *reinterpret_cast<double2*>(bel(slot, 2 * bi, 2 * bj)) = make_double2((rowP && colT) ? ti_ : 0.0, (rowP && colV) ? tr_ * ivmj : 0.0);
            *reinterpret_cast<double2*>(bel(slot, 2 * bi + 1, 2 * bj)) = make_double2((rowQ && colT) ? -tr_ : 0.0, (rowQ && colV) ? ti_ * ivmj : 0.0);
            if (act && (y.x != 0.0 || y.y != 0.0)) { atomicAdd(wave1 ? rhsT(i) : SreP(i), tr_); atomicAdd(wave1 ? rhsV(i) : SimP(i), ti_); }
          };
          unsigned rc_pf = rc_first;
          for (int pr = tid; pr < n_pairs; pr += GW) {
            unsigned rc;
            if (STAGE == 0) { rc = rc_pf; if (pr + GW < n_pairs) rc_pf = (unsigned)sv.pair_rc[(pr + GW) / (NB * NB)]; }
            else rc = (unsigned)sv.pair_rc[pr / (NB * NB)];
            pair_item(pr, *reinterpret_cast<const double2*>(c.Yb + (size_t)pr * 2), rc);
          }
        }
        GPF_LSYNC();
        if (it == 0) GPF_STAMPS(11);
        double fabs_mis = 0.0;
        bool bad = false;
        for (int i = tid; i < nbus; i += GW) {
          const int sub = (NB == 1) ? i : i / NB, bi = lidx(i);
          double* Ad0 = bel(sub, 2 * bi, 2 * bi);
          double* Ad1 = bel(sub, 2 * bi + 1, 2 * bi);
          const int bt = c.btype[i];
          double Sr = *SreP(i), Si = *SimP(i);
          if (WPI > 1) { Sr += *rhsT(i); Si += *rhsV(i); }       // + wavefront 1's partial sums, always in this order
          const double vmi = c.vm[i], psp = c.Psp[i], qsp = c.Qsp[i];
          const bool rowP = (bt == BT_PQ || bt == BT_PV), rowQ = (bt == BT_PQ);
          const double ivmi = NB == 1 ? c.ivm[i] : fast_rcp(vmi);
          double2 r0, r1;
          if (NB == 1) {                           // the diagonal block (i, i) is built here: T_ii joins S_i last
            const double2 y = YR ? yreg[2 * YR_PASSES] : *reinterpret_cast<const double2*>(c.Yb + (size_t)i * 2);
            const double2 efi = EF(i);
            const double ei = efi.x, fi = efi.y;
            double tr_, ti_;
            t_of(y, ei, fi, ei, fi, tr_, ti_);
            if (bt != BT_OFF && (y.x != 0.0 || y.y != 0.0)) { Sr += tr_; Si += ti_; }
            *SreP(i) = Sr; *SimP(i) = Si;          // K6 reads the bus injections of the converged state
            r0 = make_double2(rowP ? ti_ : 0.0, (rowP && rowQ) ? tr_ * ivmi : 0.0);
            r1 = make_double2((rowQ && rowP) ? -tr_ : 0.0, rowQ ? ti_ * ivmi : 0.0);
          } else { r0 = *reinterpret_cast<const double2*>(Ad0); r1 = *reinterpret_cast<const double2*>(Ad1); }
          // dS/dVa_ii += j S_i ; dS/dVm_ii += S_i / |V_i| ; identity on the fixed variables
          *reinterpret_cast<double2*>(Ad0) = make_double2(rowP ? r0.x - Si : 1.0, rowQ ? fma(Sr, ivmi, r0.y) : r0.y);
          *reinterpret_cast<double2*>(Ad1) = make_double2(rowQ ? r1.x + Sr : r1.x, rowQ ? fma(Si, ivmi, r1.y) : 1.0);
          const double mp = rowP ? (Sr - psp) : 0.0;
          const double mq = rowQ ? (Si - qsp) : 0.0;
          if (BS == 2) { *rhsT(i) = -mp; *rhsV(i) = -mq; }
          else *reinterpret_cast<double2*>(c.rhs + (size_t)sub * BS + 2 * bi) = make_double2(-mp, -mq);
          const double am = fmax(fabs(mp), fabs(mq));
          if (!(am <= 1e300)) bad = true;
          fabs_mis = fmax(fabs_mis, am);
        }
        if (!done) {
          const unsigned fl = G::template any2<0>(!(fabs_mis < tol_pu), bad);
          if (fl & 2u) { status = 1; done = true; }
          else if (!(fl & 1u)) { converged = true; done = true; }
          else if (it >= max_iter) done = true;
          else ++it;
        }
        if (G::block_all_u(done)) break;
        // (several wavefronts per instance: the collective above -- executed by every lane, `done` is block-uniform -- ended with the
        //  workgroup barrier that orders the mismatch phase's LDS writes before the factorisation: no second one)
        if (!(WPI > 1 && IPW == 1)) GPF_LSYNC();
        if (it == 1) GPF_STAMPS(12);
#ifdef GPF_TIMING
        const bool ok = lu_ac(it == 1 ? &stamps.v[32] : nullptr);
#else
        const bool ok = lu_ac(nullptr);
#endif
        if (it == 1) GPF_STAMPS(13);
        // update (groups that are done keep their state) + preparation of the next pair phase (every group)
        bool fin = true, piv_ok = true;
        for (int i = tid; i < nbus; i += GW) {
          const int sub = (NB == 1) ? i : i / NB, bi = lidx(i);
          const int bt = c.btype[i];
          double va = c.va[i], vm = c.vm[i];
          double2 dx = make_double2(*rhsT(i), *rhsV(i));
          if (BS == 2) {                         // flat sweeps leave s_i = D_i x_i with the factored diagonal block in slot i
            const double2 dA = *reinterpret_cast<const double2*>(bel(sub, 0, 0)), dB = *reinterpret_cast<const double2*>(bel(sub, 1, 0));
            const double det = fma(dA.x, dB.y, -dA.y * dB.x);
            if (!(fabs(det) > 1e-300) || !(fabs(det) < 1e300)) piv_ok = false;
            const double rd = fast_rcp(det);
            dx = make_double2(fma(dB.y, dx.x, -dA.y * dx.y) * rd, fma(dA.x, dx.y, -dB.x * dx.x) * rd);
          }
          if (!done && bt != BT_OFF) {
            if (!(fabs(dx.x) < 1e300) || !(fabs(dx.y) < 1e300)) fin = false;
            if (bt == BT_PQ || bt == BT_PV) va += dx.x;
            if (bt == BT_PQ) vm += dx.y;
            if (vm < 0.0) { vm = -vm; va += 3.14159265358979323846; }
            if (fabs(va) > 3.14159265358979323846) va = remainder(va, 6.28318530717958647692);
            c.va[i] = va;
            c.vm[i] = vm;
          }
          double sn_, co;
          fast_sincos(va, sn_, co);
          setEF(i, vm * co, vm * sn_);
          if (NB == 1) c.ivm[i] = fast_rcp(vm);
          *SreP(i) = 0.0;
          *SimP(i) = 0.0;
          if (WPI > 1) { *rhsT(i) = 0.0; *rhsV(i) = 0.0; }
        }
        if (BS == 2) { for (int i = S.nslot_y * 2 + tid; i < S.nslot * 2; i += GW) { c.A[i] = 0.0; c.A[HS + i] = 0.0; } }
      else for (int i = S.nslot_y * B2 + tid; i < S.nslot_lu * B2; i += GW) c.A[i] = 0.0;
        if (!(WPI > 1 && IPW == 1)) GPF_LSYNC();      // (several wavefronts per instance: the barrier of the collective below is the boundary)
        if (!done && G::template any2<1>(!ok || !piv_ok, !fin) != 0u) { status = 4; done = true; }
        if (it == 1) GPF_STAMPS(14);
      }
    }
    if (status == 0 && !converged) status = 1;
  }
  n_iter_out = it;
  if (G::block_all_u(status != 0)) return status;
  GPF_STAMPS(5);

  // ---- K6: results ---------------------------------------------------------------------------------------------------------
  const auto out = gptr(ctl.otraj ? b.traj_out : b.out) + (size_t)ctl.orow * g.n_out;
  const auto lstat = gptr(ctl.otraj ? b.traj_lstat : b.line_status) + (size_t)ctl.orow * g.n_line;
  const bool wtopo = !reuse || ctl.write_topo;
  // WHOLE-LINE RESULT STORES (several wavefronts per instance: the 118-substation grids).  The row is 25 field segments of n_line / n_gen /
  // ... floats at offsets that are no multiples of a 128-byte line: stored field by field, every segment leaves a partial line at both
  // ends (PMC, round 5: 1.33 x the row's bytes written, 1.5 x the algorithmic traffic).  The lanes therefore put their values into a
  // copy of the row in LDS -- the row-1 half of the block array, whose blocks are dead after the last Newton iteration and are rebuilt
  // by the next solve -- placed so that LDS and HBM addresses agree modulo 128, and the block then stores it as 16 bytes per lane, 2 KB
  // per instruction, every instruction whole lines (dword stores only before the first and behind the last whole line of the row).
#ifdef GPF_NO_ROWLDS
  constexpr bool ROWLDS = false;
#else
  constexpr bool ROWLDS = NB == 1 && WPI > 1;
#endif
  const unsigned row_mis = (unsigned)((((size_t)ctl.orow * (size_t)g.n_out) * 4u) & 127u);       // (the arrays themselves are 256-byte aligned)
  const bool rowlds = ROWLDS && (size_t)S.rslot0 * 16 >= (size_t)g.n_out * 4 + 128;               // block-uniform: the row fits the slot part of the half
  float* const out_l = reinterpret_cast<float*>(c.A + HS) + row_mis / 4u;
  c.out_l = rowlds ? out_l : nullptr;
  auto put = [&](int idx, float v) { if (ROWLDS && rowlds) out_l[idx] = v; else out[idx] = v; };
  const double RAD2DEG = 57.295779513082320877;
  const double SQRT3 = 1.7320508075688772935;
  GPF_LSYNC();
  if (is_dc) {
    for (int i = tid; i < nbus; i += GW) { *SreP(i) = c.Gs[i]; *SimP(i) = 0.0; }
    GPF_LSYNC();
    if (acc_lane)
    for (int l = tid; l < g.n_line; l += GWA) {
      const int f = c.lor_b[l], t = c.lex_b[l];
      if (f < 0) continue;
      const double fl = (c.va[f] - c.va[t]) * sv.br_bdc[l];
      atomicAdd(SreP(f), fl);
      atomicAdd(SreP(t), -fl);
    }
    GPF_LSYNC();
  }
  // Small grids, AC, every generator connected: every element kind fits one pass of the group, so the five loops below -- each a
  // chain "element -> bus (LDS), bus state (LDS), arithmetic, stores" that waits for its own reads -- run as ONE straight-line block:
  // all element -> bus reads first, then all state / table reads, then the arithmetic (removing the load + generator loops
  // altogether bounds what they cost at 4.3 % of a 14-substation step).
  const bool gen_static_k6 = NB == 1 && !TC && G::block_all_u(ts.gen_base);
  constexpr bool K6F = IPW > 1;                   // (instance-group kernels only: the block costs ~40 VGPRs, which the one- and
                                                  //  two-wavefront kernels -- 3 waves/SIMD at <= 168, Ybus in registers at 256 -- do not have)
  const bool k6_fast = K6F && !is_dc && gen_static_k6 && g.n_line <= GW && g.n_load <= GW && g.n_gen <= GW && g.n_sto <= GW && g.n_shunt <= GW &&
                       g.n_load > 0;
  if (k6_fast) {
    const int l = tid;
    const bool hl = l < g.n_line, hd = l < g.n_load, hg = l < g.n_gen, hs = l < g.n_sto, hh = l < g.n_shunt;
    const int il = hl ? l : 0, id = hd ? l : 0, ig = hg ? l : 0, is_ = hs ? l : 0, ih = hh ? l : 0;
    // element -> bus
    const int f = c.lor_b[il], t = c.lex_b[il], bd = c.load_b[id], bg = c.gen_b[ig];
    const int bs = g.n_sto ? (int)c.sto_b[is_] : -1, bh = g.n_shunt ? (int)c.sh_b[ih] : -1;
    const int fc = f >= 0 ? f : 0, tc = t >= 0 ? t : 0, bdc = bd >= 0 ? bd : 0, bgc = bg >= 0 ? bg : 0, bsc = bs >= 0 ? bs : 0, bhc = bh >= 0 ? bh : 0;
    // bus state + tables
    const double2 ef_f = EF(fc), ef_t = EF(tc);
    const double vmf = c.vm[fc], vmt = c.vm[tc], ef = ef_f.x, ff = ef_f.y, et = ef_t.x, ft = ef_t.y, vaf = c.va[fc], vat = c.va[tc];
    const double4 ya = sv.br_y.ld4((size_t)8 * il), yb = sv.br_y.ld4((size_t)8 * il + 4);
    const double vnf = sv.line_vn[2 * il], vnt = sv.line_vn[2 * il + 1], kaf = sv.line_ka[2 * il], kat = sv.line_ka[2 * il + 1];
    const double ivmf = c.ivm[fc], ivmt = c.ivm[tc];       // 1 / |V| of the final iterate (Newton update phase)
    const double vmd = c.vm[bdc], vad = c.va[bdc], lpd = GPF_INJ(oo.inj_load_p + id), lqd = GPF_INJ(oo.inj_load_q + id), vnd = sv.load_vn[id];
    const double simg = *SimP(bgc), qspg = c.Qsp[bgc], sreg = *SreP(bgc), pspg = c.Psp[bgc], vmg = c.vm[bgc], vag = c.va[bgc];
    const int gw_ = sv.gen_cnt[ig], gsl = sv.gen_slack[ig];
    const double gqmn = sv.gen_qmin_tot[ig], gqmx = sv.gen_qmax_tot[ig], gmn = sv.gen_min_q[ig], gmx = sv.gen_max_q[ig], gvn = sv.gen_vn[ig];
    const double gpi = GPF_INJ(oo.inj_gen_p + ig);
    double vms = 0.0, vas = 0.0, sps = 0.0, sqs = 0.0, vns = 0.0, vmh = 0.0, hp = 0.0, hq = 0.0, hfac = 0.0, hvn = 0.0;
    int shb_h = -1;
    if (g.n_sto) { vms = c.vm[bsc]; vas = c.va[bsc]; sps = GPF_INJ(oo.inj_sto_p + is_); sqs = GPF_INJ(oo.inj_sto_q + is_); vns = sv.sto_vn[is_]; }
    if (g.n_shunt) { vmh = c.vm[bhc]; hp = GPF_INJ(oo.inj_sh_p + ih); hq = GPF_INJ(oo.inj_sh_q + ih); hfac = sv.shunt_fact[ih]; hvn = sv.shunt_vn[ih]; if (wtopo) shb_h = shb[ih]; }
    // lines
    {
      const bool on = f >= 0;
      const double ifr = ya.x * ef - ya.y * ff + ya.z * et - ya.w * ft;
      const double ifi = ya.x * ff + ya.y * ef + ya.z * ft + ya.w * et;
      const double itr = yb.x * ef - yb.y * ff + yb.z * et - yb.w * ft;
      const double iti = yb.x * ff + yb.y * ef + yb.z * ft + yb.w * et;
      const double pf = (ef * ifr + ff * ifi) * sn, qf = (ff * ifr - ef * ifi) * sn;
      const double pt = (et * itr + ft * iti) * sn, qt = (ft * itr - et * iti) * sn;
      // i = |S| / (sqrt(3) |V| vn): reciprocals instead of two float64 divisions (~25 VALU instructions each)
      const float a_or = on ? (float)(sqrt(pf * pf + qf * qf) * ivmf * kaf) : 0.f;
      const float a_ex = on ? (float)(sqrt(pt * pt + qt * qt) * ivmt * kat) : 0.f;
      if (hl) {
        if (wtopo) lstat[l] = on ? 1 : 0;
        out[oo.p_or + l] = on ? (float)pf : 0.f; out[oo.q_or + l] = on ? (float)qf : 0.f; out[oo.v_or + l] = on ? (float)(vmf * vnf) : 0.f;
        out[oo.a_or + l] = a_or; out[oo.th_or + l] = on ? (float)(vaf * RAD2DEG) : 0.f;
        out[oo.p_ex + l] = on ? (float)pt : 0.f; out[oo.q_ex + l] = on ? (float)qt : 0.f; out[oo.v_ex + l] = on ? (float)(vmt * vnt) : 0.f;
        out[oo.a_ex + l] = a_ex; out[oo.th_ex + l] = on ? (float)(vat * RAD2DEG) : 0.f;
      }
      a_or_first = hl ? a_or : a_or_first;
    }
    GPF_STAMPS(22);
    if (hd) {
      const bool on = bd >= 0;
      out[oo.load_p + l] = on ? (float)lpd : 0.f;
      out[oo.load_q + l] = on ? (float)lqd : 0.f;
      out[oo.load_v + l] = on ? (float)(vmd * vnd) : 0.f;
      out[oo.load_th + l] = on ? (float)(vad * RAD2DEG) : 0.f;
    }
    if (hs) {
      const bool on = bs >= 0;
      out[oo.sto_p + l] = on ? (float)sps : 0.f;
      out[oo.sto_q + l] = on ? (float)sqs : 0.f;
      out[oo.sto_v + l] = on ? (float)(vms * vns) : 0.f;
      out[oo.sto_th + l] = on ? (float)(vas * RAD2DEG) : 0.f;
    }
    if (hh) {
      const bool on = bh >= 0;
      const double v = on ? vmh : 0.0;
      const auto sbo = gptr(ctl.otraj ? b.traj_shb : b.shunt_bus_out) + (size_t)ctl.orow * g.n_shunt;
      out[oo.sh_p + l] = on ? (float)(hp * hfac * v * v) : 0.f;
      out[oo.sh_q + l] = on ? (float)(hq * hfac * v * v) : 0.f;
      out[oo.sh_v + l] = on ? (float)(v * hvn) : 0.f;
      if (wtopo) sbo[l] = on ? shb_h : -1;
    }
    GPF_STAMPS(23);
    GPF_STAMPS(24);
    if (hg) {
      float gp = 0.f, gq = 0.f, gv = 0.f, gth = 0.f;
      if (bg >= 0) {
        const double qtot = (simg - qspg) * sn;
        const int cn = gw_ & 0xffff, ns = gw_ >> 16;
        double q;
        if (cn == 1) q = qtot;
        else if (gqmn == gqmx) q = qtot * fast_rcp((double)cn);
        else q = gmn + (qtot - gqmn) * fast_rcp(gqmx - gqmn + 2.220446049250313e-16) * (gmx - gmn);
        double p = gpi;
        if (gsl) p = (sreg - pspg) * sn * fast_rcp((double)ns);
        gp = (float)p; gq = (float)q;
        gv = (float)(vmg * gvn);
        gth = (float)(vag * RAD2DEG);
      }
      out[oo.gen_p + l] = gp; out[oo.gen_q + l] = gq; out[oo.gen_v + l] = gv; out[oo.gen_th + l] = gth;
    }
  } else {
  for (int l = tid; l < g.n_line; l += GW) {
    const int f = c.lor_b[l], t = c.lex_b[l];
    if (wtopo) lstat[l] = f >= 0 ? 1 : 0;
    float p_or = 0.f, q_or = 0.f, v_or = 0.f, a_or = 0.f, th_or = 0.f;
    float p_ex = 0.f, q_ex = 0.f, v_ex = 0.f, a_ex = 0.f, th_ex = 0.f;
    if (f >= 0) {
      const double vnf = sv.line_vn[2 * l], vnt = sv.line_vn[2 * l + 1];
      const double vmf = c.vm[f], vmt = c.vm[t];
      double pf, qf, pt, qt;
      if (is_dc) {
        pf = (c.va[f] - c.va[t]) * sv.br_bdc[l] * sn;
        pt = -pf; qf = 0.0; qt = 0.0;
      } else {
        const double4 ya = sv.br_y.ld4((size_t)8 * l), yb = sv.br_y.ld4((size_t)8 * l + 4);
        const double2 ef_f = EF(f), ef_t = EF(t);
        const double ef = ef_f.x, ff = ef_f.y, et = ef_t.x, ft = ef_t.y;
        const double ifr = ya.x * ef - ya.y * ff + ya.z * et - ya.w * ft;
        const double ifi = ya.x * ff + ya.y * ef + ya.z * ft + ya.w * et;
        const double itr = yb.x * ef - yb.y * ff + yb.z * et - yb.w * ft;
        const double iti = yb.x * ff + yb.y * ef + yb.z * ft + yb.w * et;
        pf = (ef * ifr + ff * ifi) * sn;  qf = (ff * ifr - ef * ifi) * sn;
        pt = (et * itr + ft * iti) * sn;  qt = (ft * itr - et * iti) * sn;
      }
      p_or = (float)pf; q_or = (float)qf; p_ex = (float)pt; q_ex = (float)qt;
      a_or = (float)(sqrt(pf * pf + qf * qf) * fast_rcp(vmf) * sv.line_ka[2 * l]);
      a_ex = (float)(sqrt(pt * pt + qt * qt) * fast_rcp(vmt) * sv.line_ka[2 * l + 1]);
      v_or = (float)(vmf * vnf); v_ex = (float)(vmt * vnt);
      th_or = (float)(c.va[f] * RAD2DEG); th_ex = (float)(c.va[t] * RAD2DEG);
    }
    put(oo.p_or + l, p_or); put(oo.q_or + l, q_or); put(oo.v_or + l, v_or); put(oo.a_or + l, a_or); put(oo.th_or + l, th_or);
    if (l == tid) a_or_first = a_or;                 // the step kernel derives rho / the protection counters of this line from it
    put(oo.p_ex + l, p_ex); put(oo.q_ex + l, q_ex); put(oo.v_ex + l, v_ex); put(oo.a_ex + l, a_ex); put(oo.th_ex + l, th_ex);
  }
  GPF_STAMPS(22);
  const bool inj_regs = WPI > 1 && ctl.inj_regs;
  if (inj_regs && tid < WAVE) {                              // load p / q from the registers of K9's owner lanes (wavefront 0: loads tid, tid + 64)
    if (tid < g.n_load) { const bool on = c.load_b[tid] >= 0; put(oo.load_p + tid, on ? ctl.r_lp0 : 0.f); put(oo.load_q + tid, (on && !is_dc) ? ctl.r_lq0 : 0.f); }
    if (tid + WAVE < g.n_load) { const bool on = c.load_b[tid + WAVE] >= 0; put(oo.load_p + tid + WAVE, on ? ctl.r_lp1 : 0.f); put(oo.load_q + tid + WAVE, (on && !is_dc) ? ctl.r_lq1 : 0.f); }
  }
  for (int i = tid; i < g.n_load; i += GW) {
    const int bu = c.load_b[i];
    const bool on = bu >= 0;
    if (!inj_regs) {
      put(oo.load_p + i, on ? (float)GPF_INJ(oo.inj_load_p + i) : 0.f);
      put(oo.load_q + i, (on && !is_dc) ? (float)GPF_INJ(oo.inj_load_q + i) : 0.f);
    }
    put(oo.load_v + i, on ? (float)(c.vm[bu] * sv.load_vn[i]) : 0.f);
    put(oo.load_th + i, on ? (float)(c.va[bu] * RAD2DEG) : 0.f);
  }
  for (int i = tid; i < g.n_sto; i += GW) {
    const int bu = c.sto_b[i];
    const bool on = bu >= 0;
    put(oo.sto_p + i, on ? (float)GPF_INJ(oo.inj_sto_p + i) : 0.f);
    put(oo.sto_q + i, (on && !is_dc) ? (float)GPF_INJ(oo.inj_sto_q + i) : 0.f);
    put(oo.sto_v + i, on ? (float)(c.vm[bu] * sv.sto_vn[i]) : 0.f);
    put(oo.sto_th + i, on ? (float)(c.va[bu] * RAD2DEG) : 0.f);
  }
  const auto sbo = gptr(ctl.otraj ? b.traj_shb : b.shunt_bus_out) + (size_t)ctl.orow * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += GW) {
    const int bu = c.sh_b[i];
    const bool on = bu >= 0;
    const double v = on ? c.vm[bu] : 0.0;
    put(oo.sh_p + i, on ? (float)(GPF_INJ(oo.inj_sh_p + i) * sv.shunt_fact[i] * v * v) : 0.f);
    put(oo.sh_q + i, (on && !is_dc) ? (float)(GPF_INJ(oo.inj_sh_q + i) * sv.shunt_fact[i] * v * v) : 0.f);
    put(oo.sh_v + i, on ? (float)(v * sv.shunt_vn[i]) : 0.f);
    if (wtopo) sbo[i] = on ? shb[i] : -1;
  }
  // generators (pypower pfsoln): per-bus totals accumulated in LDS with atomics (the block array is dead by now and
  // serves as scratch), then a per-generator pass.  Bus balances: total generation at a bus = S_inj - (P,Q)_spec,
  // i.e. slack P = (Re S - Psp) * sn (Psp already holds +other generators -loads) and Q_gen,total = (Im S - Qsp) * sn.
  GPF_STAMPS(23);
  {
    double* qmin_t = c.A;
    double* qmax_t = c.A + nbus;
    int* cnt = reinterpret_cast<int*>(c.A + 2 * (size_t)nbus);
    int* nsl = cnt + nbus;
    // every generator connected (single-busbar layout): the per-bus totals are the static per-generator tables of the grid
    const bool gen_static = gen_static_k6;
    if (!gen_static) {
      GPF_LSYNC();
      for (int i = tid; i < nbus; i += GW) { qmin_t[i] = 0.0; qmax_t[i] = 0.0; cnt[i] = 0; nsl[i] = 0; }
      GPF_LSYNC();
      if (acc_lane)
      for (int i = tid; i < g.n_gen; i += GWA) {
        const int bu = c.gen_b[i];
        if (bu < 0) continue;
        atomicAdd(&cnt[bu], 1);
        atomicAdd(&qmin_t[bu], sv.gen_min_q[i]);
        atomicAdd(&qmax_t[bu], sv.gen_max_q[i]);
        if (sv.gen_slack[i]) atomicAdd(&nsl[bu], 1);
      }
      GPF_LSYNC();
    }
    GPF_STAMPS(24);
    for (int i = tid; i < g.n_gen; i += GW) {
      const int bu = c.gen_b[i];
      float gp = 0.f, gq = 0.f, gv = 0.f, gth = 0.f;
      if (bu >= 0) {
        const double qtot = (*SimP(bu) - c.Qsp[bu]) * sn;
        int cn, ns;
        double qmn, qmx;
        if (gen_static) { const int w = sv.gen_cnt[i]; cn = w & 0xffff; ns = w >> 16; qmn = sv.gen_qmin_tot[i]; qmx = sv.gen_qmax_tot[i]; }
        else { cn = cnt[bu]; ns = nsl[bu]; qmn = qmin_t[bu]; qmx = qmax_t[bu]; }
        const double mn = sv.gen_min_q[i], mx = sv.gen_max_q[i];
        double q;
        if (is_dc) q = 0.0;
        else if (cn == 1) q = qtot;
        else if (qmn == qmx) q = qtot * fast_rcp((double)cn);
        else q = mn + (qtot - qmn) * fast_rcp(qmx - qmn + 2.220446049250313e-16) * (mx - mn);
        double p = (inj_regs && i == tid) ? (double)ctl.r_pp : GPF_INJ(oo.inj_gen_p + i);       // (n_gen <= 64: generator i is lane i's in K9 too)
        if (sv.gen_slack[i]) p = (*SreP(bu) - c.Psp[bu]) * sn * fast_rcp((double)ns);
        gp = (float)p; gq = (float)q;
        gv = (float)(c.vm[bu] * sv.gen_vn[i]);
        gth = (float)(c.va[bu] * RAD2DEG);
      }
      put(oo.gen_p + i, gp); put(oo.gen_q + i, gq); put(oo.gen_v + i, gv); put(oo.gen_th + i, gth);
    }
  }
  }
  if (ROWLDS && rowlds) {
    GPF_LSYNC();                                                   // the row is complete in LDS
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const int n_out = g.n_out;
    int a = (int)(((128u - row_mis) & 127u) / 4u);                 // floats before the first whole line of the row
    a = a < n_out ? a : n_out;
    if (tid < a) out[tid] = out_l[tid];
    const int n4 = (n_out - a) / 4;
    const v4f_* const sl = reinterpret_cast<const v4f_*>(out_l + a);          // 16-byte aligned: out_l = row (mod 128)
    const auto dg = (GPF_GLOBAL v4f_*)(out + a);
    for (int i = tid; i < n4; i += GW) dg[i] = sl[i];
    const int t0 = a + 4 * n4;
    if (tid < n_out - t0) out[t0 + tid] = out_l[t0 + tid];
  }
  GPF_STAMPS(25);
  if (wtopo) {                           // topo_vect only depends on the topology: it stands when the topology does
    // one pass, every position written once by one lane: an element's bus, -1 for both ends of a line that is out of service
    // (StatOff::pos_line names the line of a position)
    const auto to = gptr(ctl.otraj ? b.traj_topo : b.topo_out) + (size_t)ctl.orow * g.dim_topo;
    // (observation trajectory: the row is written at every step although the topology stands -- the values of the first
    //  positions of every lane then come from registers instead of a global round trip to the lane's topology row)
    constexpr int TCN = IPW > 1 ? 2 : 0;          // (instance-group kernels: <= 2 positions per lane; -0.9 % on the 36-substation N-1 kernel otherwise)
#pragma unroll
    for (int k = 0; k < TCN; ++k) {
      const int i = tid + k * GW;
      if (i < g.dim_topo) {
        int val;
        if (reuse && !ctl.tc_rebuild) val = ts.tc[k];
        else {
          const int v = topo_g[i], pl = sv.pos_line[i];
          const bool line_out = pl >= 0 && c.lor_b[pl] < 0;
          val = (v >= 1 && !line_out) ? v : -1;
          ts.tc[k] = val;
        }
        to[i] = val;
      }
    }
    for (int i = tid + TCN * GW; i < g.dim_topo; i += GW) {
      const int v = topo_g[i], pl = sv.pos_line[i];
      const bool line_out = pl >= 0 && c.lor_b[pl] < 0;
      to[i] = (v >= 1 && !line_out) ? v : -1;
    }
  }
  GPF_STAMPS(26);
  if (ctl.write_bus) {
  const auto bvm = gptr(b.bus_vm) + (size_t)inst * g.nb_tot;
  const auto bva = gptr(b.bus_va) + (size_t)inst * g.nb_tot;
  const double nand = __builtin_nan("");
  for (int i = tid; i < g.nb_tot; i += GW) {
    const int sub = i % nsub, lb = i / nsub + 1;            // global bus = sub + (local-1)*n_sub
    int bu;
    if (TC) bu = lb <= g.n_busbar ? sv.node_of[sub * g.n_busbar + (lb - 1)] : -1;
    else if (NB == 1) bu = (c.sub_bb[sub] == lb) ? sub : -1;
    else bu = (lb <= NB) ? sub * NB + (lb - 1) : -1;
    const bool on = bu >= 0 && c.btype[bu] != BT_OFF;
    bvm[i] = on ? c.vm[bu] : nand;
    bva[i] = on ? c.va[bu] * RAD2DEG : nand;
  }
  }
  GPF_STAMPS(6);
  return status;
#undef GPF_INJ
}

// ---------------------------------------------------------------------------------------------------
// LDS carve + static view + symbolic header of a block.  Topology-class launches (TC): the header and the graph-dependent
// tables come from the class of the block's lanes (the host packs lanes of ONE class into a block), LDS is sized for the
// largest class of the launch.
#ifdef GPF_JIT
#define GPF_GRID_SYM Sg_loc
#define GPF_GRID_SYM_DECL SymDev Sg_loc = P->sym; GPF_SPEC_SYM(Sg_loc); GPF_SPEC_SO(Sg_loc.so);
#else
#define GPF_GRID_SYM P->sym
#define GPF_GRID_SYM_DECL
#endif
#define GPF_CARVE_AND_VIEW(G_)                                                                                                   \
  GPF_GRID_SYM_DECL                                                                                                              \
  SymDev S_loc = GPF_GRID_SYM;                                                                                                   \
  if (TC) { S_loc = P->classes[gptr(lane_class)[blockIdx.x * IPW]].sym; GPF_SPEC_SO(S_loc.so); }                                 \
  pin_sgpr(S_loc.n); pin_sgpr(S_loc.nslot); pin_sgpr(S_loc.nslot_y); pin_sgpr(S_loc.back_off);                                  \
  pin_sgpr(S_loc.scale_off); pin_sgpr(S_loc.n_scale); pin_sgpr(S_loc.back_first); pin_sgpr(S_loc.static_connected);              \
  pin_sgpr(S_loc.rslot0); pin_sgpr(S_loc.n_up);                                                                                 \
  const SymDev& S = S_loc;                                                                                                       \
  const int lds_rows = TC ? P->tc_rows : -1, lds_nslot = TC ? P->tc_nslot : (NB == 1 ? GPF_GRID_SYM.nslot : GPF_GRID_SYM.nslot_lu),                                        \
            lds_nslot_y = YR ? 0 : TC ? P->tc_nslot_y : GPF_GRID_SYM.nslot_y;      /* YR: the Ybus blocks live in registers */          \
  const bool lds_dcf = P->dcf != 0;                                                                                              \
  const size_t per_inst = lds_bytes_instance<NB>(G_, lds_nslot, lds_nslot_y, STAGE != 0, lds_rows, lds_dcf);                     \
  carve_sparse<NB>(c, smem + (size_t)grp * per_inst, G_, lds_nslot, lds_nslot_y, STAGE != 0, lds_rows, lds_dcf);                 \
  constexpr int GWI = gw_index(Grp<IPW, WPI>::GW);                                                                               \
  FlatDev F_loc = S_loc.fl[GWI];                                                                                                 \
  pin_sgpr(F_loc.n_fwd); pin_sgpr(F_loc.n_scale); pin_sgpr(F_loc.n_scale_rhs); pin_sgpr(F_loc.n_back); pin_sgpr(F_loc.scale_off); \
  pin_sgpr(F_loc.back_off); pin_sgpr(F_loc.rhs_field0);                                                                          \
  const FlatDev& FL = F_loc;                                                                                                     \
  StatView<STAGE> sv;                                                                                                            \
  make_stat_view<STAGE, NB == 1>(sv, GPF_GRID_SYM, smem + (size_t)IPW * per_inst, S_loc.flat[GWI], IPW > 1 ? 0 : F_loc.n_words);        \
  if (TC) {                                                                                                                      \
    const TopoClassDev& tc_ = P->classes[gptr(lane_class)[blockIdx.x * IPW]];                                              \
    sv.pair_rc.p = tc_.pair_rc; sv.up.p = tc_.up; sv.br_slot.p = tc_.br_slot; sv.node_of.p = tc_.node_of;                        \
  }

// developer (tools/unroll_bisect.py): -DGPF_FORCE_MINW=<waves per SIMD> makes the compiler fit the kernels below into 512 / waves
// registers whatever their MINW argument says, i.e. forces spills to scratch memory
#ifdef GPF_FORCE_MINW
#define GPF_MINW(m_) GPF_FORCE_MINW
#else
#define GPF_MINW(m_) m_
#endif

template <int NB, int STAGE, int IPW, int MINW, int WPI, bool TC = false, bool YR = false>
__global__ __launch_bounds__(WAVE * WPI, GPF_MINW(MINW)) void runpf_sparse_kernel(const DevParamsS* __restrict__ P, int lane0, const int* __restrict__ lane_list,
                                                            const int* __restrict__ lane_class, int is_dc, int max_iter,
                                                            double tol_pu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int GW = Grp<IPW, WPI>::GW;
  const int grp = threadIdx.x / GW, tid = threadIdx.x % GW;
  // contiguous range, or (mixed batches) a device list of lanes; both padded by the host with ghost lanes to a multiple of IPW
  const int inst = lane_list ? gptr(lane_list)[blockIdx.x * IPW + grp] : lane0 + blockIdx.x * IPW + grp;
  CarveP<NB> c;
#ifdef GPF_JIT
  GridDev g_k = P->g;
  GPF_SPEC_G(g_k);
  const GridDev& g = g_k;
#else
  const GridDev& g = P->g;
#endif
  GPF_CARVE_AND_VIEW(g);
  double2 yreg[2 * YR_PASSES + 1];
  unsigned rcreg[2 * YR_PASSES];
  int n_iter, nb;
  float a_first = 0.f;
  GPF_STAMPS_DECL;
  SolveCtl ctl;
  ctl.inj_staged = false; ctl.topo_staged = false; ctl.reuse = false; ctl.dcf = false; ctl.write_bus = true; ctl.warm = false; ctl.sums_done = false;
  ctl.otraj = false; ctl.orow = inst; ctl.write_topo = true; ctl.inj_regs = false; ctl.tc_rebuild = false;
  TopoState ts;
  ts.status = 0; ts.nb = 0; ts.dc_base = false; ts.dc_out = -1; ts.gen_base = false; ts.tc[0] = ts.tc[1] = -1;
  const int st = solve_instance_sparse<NB, STAGE, IPW, WPI, TC, YR>(P, S, FL, sv, c, yreg, rcreg, inst, is_dc, max_iter, tol_pu, tid, ctl, ts, n_iter, nb, a_first GPF_STAMPS_ARG);
  GPF_SYNC();
  if (st != 0) write_nan_results<GW>(g, P->b, inst, tid, inst, false);
  if (tid == 0) {
    const auto s = gptr(P->b.status) + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = 0;
  }
}

// ---- environment injection dynamics (EnvDyn) ----------------------------------------------------------------------------------
// One lane of an aligned group of LW lanes (a power of two <= 64, inside one wavefront) per generator / storage unit; reductions
// over the group with DPP moves (lanes of other groups of the wavefront run their own instance).
// (DPP cross-lane moves in the VALU: a ds_bpermute butterfly costs an LDS round trip per step, and the projection below runs ~100
//  reductions per env step.)  KIND 0: sum, 1: min, 2: max.  Lanes that a DPP control leaves untouched keep `old`: the identity of the
// operation (0 for the sum, the lane's own value for min / max).
template <int KIND, int CTRL, int ROW_MASK>
__device__ __forceinline__ double env_dpp_step(double v) {
  const int lo_ = __double2loint(v), hi_ = __double2hiint(v);
  const int lo = __builtin_amdgcn_update_dpp(KIND == 0 ? 0 : lo_, lo_, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(KIND == 0 ? 0 : hi_, hi_, CTRL, ROW_MASK, 0xF, false);
  const double o = __hiloint2double(hi, lo);
  return KIND == 0 ? v + o : KIND == 1 ? fmin(v, o) : fmax(v, o);
}
template <int LW, int KIND>
__device__ __forceinline__ double env_greduce(double v) {
  v = env_dpp_step<KIND, 0xB1, 0xF>(v);                // quad_perm [1,0,3,2]
  v = env_dpp_step<KIND, 0x4E, 0xF>(v);                // quad_perm [2,3,0,1]
  v = env_dpp_step<KIND, 0x141, 0xF>(v);               // row_half_mirror
  v = env_dpp_step<KIND, 0x140, 0xF>(v);               // row_mirror: every lane of a 16-lane row holds the row's result
  if (LW == 16) return v;
  v = env_dpp_step<KIND, 0x142, 0xA>(v);               // row_bcast15 into rows 1 and 3: lanes of rows 1 / 3 hold rows 0+1 / 2+3
  if (LW == 32) {
    const double s0 = readlane_f64(v, 31), s1 = readlane_f64(v, 63);
    return (threadIdx.x & 32) ? s1 : s0;
  }
  v = env_dpp_step<KIND, 0x143, 0xC>(v);               // row_bcast31 into rows 2 and 3: lane 63 holds the wavefront's result
  return readlane_f64(v, 63);
}
template <int LW> __device__ __forceinline__ double env_gsum(double v) { return env_greduce<LW, 0>(v); }
template <int LW> __device__ __forceinline__ double env_gmin(double v) { return env_greduce<LW, 1>(v); }
template <int LW> __device__ __forceinline__ double env_gmax(double v) { return env_greduce<LW, 2>(v); }
constexpr int ENV_BISECT = 52;    // halvings of the multiplier bracket: 2^-52 of a bracket of a few hundred MW is below the float32 state
constexpr int ENV_BISECT_COARSE = 14;   // halvings before the closed-form attempt of env_root
// Root s in [a_, b_] of  f(s) = sum over the group's active lanes of clip(c + s * g, lo, hi) = target  (f monotone: decreasing if
// `dec`, else increasing; one generator per lane).  f is piecewise linear with at most 2 breakpoints per lane: a few bisection steps
// isolate a bracket that (almost always) holds none, where the root is the solution of ONE linear equation -- accepted when every
// lane is in the same piece (at its lower bound / inside / at its upper bound) at the candidate as at the bracket's midpoint;
// otherwise the bisection runs on to ENV_BISECT halvings (what oracle/redispatch_oracle.py solve_exact does throughout).
template <int LW>
__device__ inline double env_root(bool act, double c, double gi, double lo, double hi, double target, double a_, double b_, bool dec) {
  auto val = [&](double s_) -> double { return fmin(fmax(fma(s_, gi, c), lo), hi); };
  auto piece = [&](double s_) -> int { const double v = fma(s_, gi, c); return v <= lo ? -1 : (v >= hi ? 1 : 0); };
  int it = 0;
  for (; it < ENV_BISECT_COARSE; ++it) {
    const double mid = 0.5 * (a_ + b_);
    const double fm = env_gsum<LW>(act ? val(mid) : 0.0);
    if (dec ? fm > target : fm < target) a_ = mid; else b_ = mid;
  }
  {
    const double mid = 0.5 * (a_ + b_);
    const int pc = act ? piece(mid) : 2;
    const double s_clip = env_gsum<LW>(pc == -1 ? lo : (pc == 1 ? hi : 0.0));
    const double s_g = env_gsum<LW>(pc == 0 ? gi : 0.0), s_c = env_gsum<LW>(pc == 0 ? c : 0.0);
    if (s_g == 0.0) return mid;                                   // every lane at a bound: f is constant over the bracket
    const double cand = (target - s_clip - s_c) / s_g;
    const bool same = env_gmax<LW>((act && piece(cand) != pc) ? 1.0 : 0.0) == 0.0;
    if (same && cand >= a_ && cand <= b_) return cand;
  }
  for (; it < ENV_BISECT; ++it) {
    const double mid = 0.5 * (a_ + b_);
    const double fm = env_gsum<LW>(act ? val(mid) : 0.0);
    if (dec ? fm > target : fm < target) a_ = mid; else b_ = mid;
  }
  return 0.5 * (a_ + b_);
}
// per-lane registers of the dynamics (lane k = generator k and storage unit k of the instance)
struct EnvRegs {
  float target, actual, prev_p, charge, amount_prev, limit, curt_prev;
  bool already, fresh;
  int illegal;        // cancelled (illegal) actions since the reset: instance-uniform
};
// One env step of the dynamics for the caller's instance.  new_p: chronics set-point of generator `k` (after the environment's own
// modifications); act_r / act_s: the agent's redispatch / storage action of this step for generator / unit k (0 = none).
// Returns the storage power of unit k (MW, load convention) in `sto_power`, false when the reference would end the episode
// (ImpossibleRedispatching, baseEnv.py:3227-3247).  Follows oracle/env_oracle.py InjectionDynamics.step line by line.
template <int LW>
__device__ inline bool env_dynamics_step(const EnvDyn& E, int k, int n_gen, int n_sto, float& new_p_f, float act_r, float act_s, float act_c,
                                         EnvRegs& R, float& sto_power) {
  const bool is_gen = k < n_gen, is_sto = k < n_sto;
  // ---- _compute_storage (:2829-2905) + _withdraw_storage_losses (:2777-2790)
  double amount = 0.0;
  sto_power = 0.f;
  const float charge_before = R.charge;                     // _storage_previous_charge (:2830)
  if (n_sto > 0) {
    float pw = 0.f;
    const bool any_act = is_sto && fabsf(act_s) >= 1e-7f;
    if (any_act) {
      double eff = 1.0;
      if (E.loss_on) eff = act_s > 0.f ? E.eff_c[k] : 1.0 / E.eff_d[k];
      R.charge += (float)((double)act_s * E.coeff * eff);
      pw = act_s;
    }
    // as soon as ANY unit of the environment acts (`modif`, :2861), EVERY unit is clamped to [Emin, Emax] and the power that the
    // clamp takes back joins the storage amount -- also an idle unit whose charge the losses pulled below Emin > 0
    const bool some = env_gmax<LW>(any_act ? 1.0 : 0.0) > 0.0;
    if (some && is_sto) {
      const double emax = E.Emax[k], emin = E.Emin[k];
      if ((double)R.charge > emax) {
        double t_ = (1.0 / E.coeff) * ((double)R.charge - emax);
        if (E.loss_on) t_ /= E.eff_c[k];
        pw -= (float)t_;
        R.charge = (float)emax;
      }
      if ((double)R.charge < emin) {
        double t_ = (1.0 / E.coeff) * ((double)R.charge - emin);
        if (E.loss_on) t_ *= E.eff_d[k];
        pw -= (float)t_;
        R.charge = (float)emin;
      }
      R.charge = fmaxf(R.charge, (float)emin);
    }
    amount = some ? (double)(float)env_gsum<LW>((double)pw) : 0.0;       // (the reference sums a float32 array)
    const double tmp = amount;
    amount -= (double)R.amount_prev;
    R.amount_prev = (float)tmp;
    if (E.loss_on && is_sto) R.charge = fmaxf(R.charge - (float)(E.loss[k] * E.coeff), 0.f);
    sto_power = pw;
  }
  // ---- _aux_handle_curtailment_without_limit (:2956-2982): renewable generators capped at limit * pmax; the change of the curtailed
  //      total against the previous step joins the right-hand side of the projection.  new_p_f comes back curtailed.
  double sum_curt = 0.0;
  if (E.renewable) {
    const bool ren = is_gen && E.renewable[k] != 0;
    const bool has_act = env_gmax<LW>((ren && act_c != -1.0f) ? 1.0 : 0.0) > 0.0;
    const bool any_lim = env_gmax<LW>((is_gen && fabsf(R.limit - 1.0f) >= 1e-7f) ? 1.0 : 0.0) > 0.0;
    if (has_act || any_lim) {
      if (ren && act_c != -1.0f) R.limit = act_c;
      const bool gc = is_gen && fabsf(R.limit - 1.0f) >= 1e-7f;
      const float before = new_p_f;
      if (gc) new_p_f = fminf((float)(E.pmax[k] * (double)R.limit), new_p_f);
      const double s_new = env_gsum<LW>(gc ? (double)new_p_f : 0.0), s_old = env_gsum<LW>(gc ? (double)before : 0.0);
      const float tmp = (float)((double)(float)s_new - (double)(float)s_old);
      sum_curt = (double)tmp - (double)R.curt_prev;
      R.curt_prev = tmp;
    } else {
      sum_curt = -(double)R.curt_prev;
      R.curt_prev = 0.f;
    }
  }
  // ---- _get_already_modified_gen (:2101-2115)
  if (is_gen && fabsf(act_r) > 1e-7f) {
    R.target = R.already ? R.target + act_r : R.actual + act_r;
    R.already = true;
  }
  // ---- _prepare_redisp (:2117-2186): a target dispatch beyond pmax - pmin (below pmin - pmax) can never be met -> the action is
  //      ILLEGAL: it is taken back out of the target and BaseEnv.step replaces the whole action by do-nothing (:3189-3212): the
  //      state of charge goes back to the previous step's, the storage amount is withdrawn, the losses are applied again; the
  //      storage power already handed to the backend stays (:3829-3831).  float32 comparisons like the reference's dt_float arrays.
  {
    const bool busy = env_gmax<LW>((is_gen && (fabsf(act_r) > 1e-7f || fabsf(R.target) > 1e-7f || fabsf(R.actual) > 1e-7f)) ? 1.0 : 0.0) > 0.0;
    const float span = is_gen ? (float)E.pmax[k] - (float)E.pmin[k] : 0.f;
    const bool illegal = busy && env_gmax<LW>((is_gen && (R.target > span || R.target < -span)) ? 1.0 : 0.0) > 0.0;
    if (illegal) {
      ++R.illegal;
      if (is_gen) R.target -= act_r;
      if (n_sto > 0) {
        if (is_sto) R.charge = charge_before;
        amount -= (double)R.amount_prev;
        if (E.loss_on && is_sto) R.charge = fmaxf(R.charge - (float)(E.loss[k] * E.coeff), 0.f);
      }
    }
  }
  // ---- _make_redisp gate (:2198-2209)
  const double tol = E.tol_poly;
  const double s_act = env_gsum<LW>(is_gen ? (double)R.actual : 0.0);
  const double m_mis = env_gmax<LW>(is_gen ? fabs((double)R.actual - (double)R.target) : 0.0);
  bool ok = true;
  if (fabs((double)(float)s_act) >= tol || m_mis >= tol || fabs(amount) >= tol || fabs(sum_curt) >= tol) {
    // ---- _compute_dispatch_vect (:2211-2470): the separable QP of gridpf_redispatch.hpp, one generator per lane
    const double np_ = (double)new_p_f, a = (double)R.actual, t = (double)R.target;
    const double pv = R.fresh ? np_ : (double)R.prev_p;
    bool part = false, mod = false;
    double lo = 0.0, hi = 0.0, w = 0.0, tv = 0.0, x = 0.0;
    double s_incr = 0.0, s_up = 0.0, s_down = 0.0, s_coef = 0.0;
    const double added = 0.5 * E.eps_poly;
    if (is_gen) {
      const double pmin = E.pmin[k], pmax = E.pmax[k], ru = E.ramp_up[k], rd = E.ramp_down[k];
      part = ((np_ > 0.0) || (fabs(a) >= 1e-7) || (t != a)) && E.redispatchable[k];
      const double incr = np_ - (pv - a);
      if (part) {
        s_incr = incr; s_down = fmax(pmin - pv, -rd); s_up = fmin(pmax - pv, ru);
        const double pth = np_ + a;
        lo = fmax(pmin - pth, -rd - incr) - added;
        hi = fmin(pmax - pth, ru - incr) + added;
        w = 1.0 / (ru + rd + E.eps_poly);
        s_coef = w;
        tv = t - a;
        mod = R.already;
      }
    }
    s_incr = env_gsum<LW>(s_incr); s_up = env_gsum<LW>(s_up); s_down = env_gsum<LW>(s_down); s_coef = env_gsum<LW>(s_coef);
    const int n_mod = (int)env_gsum<LW>(part && mod ? 1.0 : 0.0);
    const double rhs = amount - sum_curt;                      // storage - curtailment (+ detached: not modelled) (:2335-2340)
    const double sum_move = s_incr + rhs;
    if (sum_move > s_up || sum_move < s_down) ok = false;
    if (part) { w /= s_coef; if (n_mod == 0) mod = true; }
    const double s_lo = env_gsum<LW>(part ? lo : 0.0), s_hi = env_gsum<LW>(part ? hi : 0.0);
    if (rhs < s_lo || rhs > s_hi) ok = false;
    if (ok) {
      const bool pm = part && mod, pf = part && !mod;
      const double f_lo = env_gsum<LW>(pf ? lo : 0.0), f_hi = env_gsum<LW>(pf ? hi : 0.0);
      const double lam_lo = env_gmin<LW>(pm ? 2.0 * w * (tv - hi) : 1e300), lam_hi = env_gmax<LW>(pm ? 2.0 * w * (tv - lo) : -1e300);
      const double g_mod = pm ? -0.5 / w : 0.0, g_free = pf ? 1.0 / w : 0.0;        // x(lambda) = tv - lambda / (2 w);  x(alpha) = alpha / w
      const double s0 = env_gsum<LW>(pm ? fmin(fmax(tv, lo), hi) : 0.0);
      double lam = 0.0;
      int free_at = 0;
      if (rhs - s0 > f_hi) { free_at = 1; lam = env_root<LW>(pm, tv, g_mod, lo, hi, rhs - f_hi, lam_lo, 0.0, true); }
      else if (rhs - s0 < f_lo) { free_at = -1; lam = env_root<LW>(pm, tv, g_mod, lo, hi, rhs - f_lo, 0.0, lam_hi, true); }
      if (pm) x = fmin(fmax(fma(lam, g_mod, tv), lo), hi);
      const double got = env_gsum<LW>(pm ? x : 0.0);
      if (free_at != 0) {
        if (pf) x = free_at > 0 ? hi : lo;
        const double rest = rhs - got - (free_at > 0 ? f_hi : f_lo);
        const bool inside = pm && x > lo && x < hi;
        const int n_in = (int)env_gsum<LW>(inside ? 1.0 : 0.0);
        if (n_in > 0 && inside) x += rest / n_in;
      } else {
        const double r = rhs - got;
        double a_ = env_gmin<LW>(pf ? fmin(lo * w, hi * w) : 1e300), b_ = env_gmax<LW>(pf ? fmax(lo * w, hi * w) : -1e300);
        if (a_ <= b_) {
          const double alpha = env_root<LW>(pf, 0.0, g_free, lo, hi, r, a_, b_, false);
          if (pf) x = fmin(fmax(alpha * g_free, lo), hi);
        }
      }
      if (part) R.actual = (float)(a + x);
    }
  }
  if (ok && is_gen) { R.prev_p = new_p_f + R.actual; }
  if (ok) R.fresh = false;
  return ok;
}

// Batched environment steps: n_steps consecutive DoNothing env.step of every lane in ONE launch.  Per step: chronics row
// (+ jitter, rebalancing, redispatch delta) -> injections -> power flow -> results row -> overflow counters / cascade (K7) ->
// rho, status, episode bookkeeping, all written to HBM.  What does NOT change from one step to the next stays in LDS / registers:
// the static tables, the lane's chronics cursor and -- as long as no line tripped and no lane failed -- everything that only
// depends on the topology (element -> bus maps, bus types, connectivity verdict, Ybus blocks, the factored DC matrix).
// ENV: the kernel also evaluates the environment's injection dynamics (EnvDyn) at every step -- a separate instantiation, so that
// the plain DoNothing kernels do not carry its registers and code (measured: -2.5 % on the 14-substation headline otherwise).
template <int NB, int STAGE, int IPW, int MINW, int WPI, bool TC = false, bool YR = false, bool ENV = false>
__global__ __launch_bounds__(WAVE * WPI, GPF_MINW(MINW)) void step_sparse_kernel(const DevParamsS* __restrict__ P, const int* __restrict__ lane_list,
                                                           const int* __restrict__ lane_class, int max_iter, double tol_pu,
                                                           StepArgs sa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Grp<IPW, WPI> G;
  constexpr int GW = G::GW;
#ifdef GPF_JIT
  GridDev g_k = P->g;
  GPF_SPEC_G(g_k);
  const GridDev& g = g_k;
  OutOff oo_k = P->oo;
  GPF_SPEC_OO(oo_k);
  const OutOff& oo = oo_k;
#else
  const GridDev& g = P->g;
  const OutOff& oo = P->oo;
#endif
  const Bufs& b = P->b;
  const int grp0 = threadIdx.x / GW, tid0 = threadIdx.x % GW;
  const int inst0 = lane_list ? gptr(lane_list)[blockIdx.x * IPW + grp0] : sa.lane0 + blockIdx.x * IPW + grp0;   // ghost-padded by the host
  int grp = grp0, tid = tid0, inst = inst0;
  const bool ghost = inst >= (int)b.n_real_lanes;               // padding lane of an instance group: computes, never mutates state
  CarveP<NB> c;
  GPF_CARVE_AND_VIEW(g);
  double2 yreg[2 * YR_PASSES + 1];
  unsigned rcreg[2 * YR_PASSES];
  if (STAGE) GPF_SYNC();                                       // the static tables are read from here on
  GPF_STAMPS_DECL;
  GPF_STAMPS(8);
  // ---- kept state of the reference topology (KeepArgs), first half: EVERY load of the blob and of the lane's key rows is issued here, ahead of
  //      the lane constants, so that they share the launch's first (cold) round trip; they are used at the end of the prologue.  (Waiting for
  //      the key comparison before fetching the state cost three round trips: 14 k cycles on 118 substations, more than K1's rebuild.) -------
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  typedef int v2i_ __attribute__((ext_vector_type(2)));
  typedef double v2d_ __attribute__((ext_vector_type(2)));
  constexpr bool KEEP = NB == 1 && !TC;
  const bool keep_on = KEEP && sa.keep.p != nullptr;            // kernel-uniform
  const bool keep_dcf = P->dcf != 0;
  constexpr int KT = 2, KC = 2, KY = 4, KA = 8;                 // unrolled trips of the key / image / Ybus / DC-factor loads (longer rows: loops in the second half)
  int k_h0 = 0, k_h1 = 0, k_h2 = 0;                             // header words (KEEP_HDR_INTS)
  int kk[KT] = {}, tn[KT] = {};
  v4i_ im[KC] = {};
  double2 yy[KY] = {};
  double aa[KA] = {};
  bool k_same = true;
  if (keep_on) {
    const auto kb = (GPF_GLOBAL const unsigned char*)sa.keep.p + (YR ? (size_t)sa.keep.stride : 0);
    const auto hdr = (GPF_GLOBAL const int*)kb;
    { int z_ = 0; asm volatile("" : "+v"(z_)); k_h0 = hdr[z_]; k_h1 = hdr[z_ + 1]; k_h2 = hdr[z_ + 2]; }   // (vector loads: three more live SGPRs spilled this kernel to scratch)
    const auto kt = hdr + KEEP_HDR_INTS;
    const auto topo_now = gptr(b.topo) + (size_t)inst * g.dim_topo;
#pragma unroll
    for (int u = 0; u < KT; ++u) { const int i = tid + u * GW; const bool on_ = i < g.dim_topo; kk[u] = on_ ? kt[i] : 0; tn[u] = on_ ? topo_now[i] : 0; }
    if (tid < g.n_shunt) k_same = kt[g.dim_topo + tid] == gptr(b.shunt_bus)[(size_t)inst * g.n_shunt + tid];
    {                                                           // shunt p | q set-points: part of Ybus
      const auto kd = (GPF_GLOBAL const double*)(kb + sa.keep.off_kd);
      const auto inj_now = gptr(b.inj) + (size_t)inst * g.n_inj + oo.inj_sh_p;
      const double a0_ = tid < 2 * g.n_shunt ? kd[tid] : 0.0, b0_ = tid < 2 * g.n_shunt ? inj_now[tid] : 0.0;
      const double a1_ = tid + GW < 2 * g.n_shunt ? kd[tid + GW] : 0.0, b1_ = tid + GW < 2 * g.n_shunt ? inj_now[tid + GW] : 0.0;
      k_same &= a0_ == b0_ && a1_ == b1_;
    }
    const auto km16 = (GPF_GLOBAL const v4i_*)(kb + sa.keep.off_m);
    const int n_ch = (sa.keep.n_m + 3) >> 2;                    // 16-byte chunks of the LDS image (the blob is padded to whole chunks)
#pragma unroll
    for (int u = 0; u < KC; ++u) { const int ch = tid + u * GW; im[u] = ch < n_ch ? km16[ch] : v4i_{0, 0, 0, 0}; }
    const auto ky = (GPF_GLOBAL const v2d_*)(kb + sa.keep.off_y);
    auto ld2 = [&](int i_) { const v2d_ t_ = ky[i_]; return make_double2(t_.x, t_.y); };
    if (YR) {
#pragma unroll
      for (int k = 0; k < YR_PASSES; ++k) {
        const int pr = tid + k * GW;
        const bool on_ = pr < S.n_up;
        rcreg[2 * k] = on_ ? (unsigned)sv.up[2 * pr] : 0u; rcreg[2 * k + 1] = on_ ? (unsigned)sv.up[2 * pr + 1] : 0u;
        yreg[2 * k] = on_ ? ld2(2 * pr) : make_double2(0.0, 0.0);
        yreg[2 * k + 1] = on_ ? ld2(2 * pr + 1) : make_double2(0.0, 0.0);
      }
      yreg[2 * YR_PASSES] = tid < g.n_sub ? ld2(2 * S.n_up + tid) : make_double2(0.0, 0.0);
    } else {
#pragma unroll
      for (int u = 0; u < KY; ++u) { const int i = tid + u * GW; yy[u] = i < S.nslot_y ? ld2(i) : make_double2(0.0, 0.0); }
    }
    const auto ka = (GPF_GLOBAL const double*)(kb + sa.keep.off_d);
#pragma unroll
    for (int u = 0; u < KA; ++u) { const int q = tid + u * GW; aa[u] = (keep_dcf && q < S.nslot) ? ka[q] : 0.0; }
  }
  // ---- per-lane constants of the launch --------------------------------------------------------------------------------------
  const int tab = b.lane_table ? gptr(b.lane_table)[inst] : 0;
  const int off = b.lane_offset ? gptr(b.lane_offset)[inst] : 0;
  int row = (sa.t + off) % sa.T;
  if (row < 0) row += sa.T;
  const bool has_sc = b.lane_scale != nullptr;
  const bool has_delta = b.lane_gen_delta != nullptr;
  // jitter factors and redispatch delta of the elements this thread handles first stay in registers for the whole launch
  float sc_p0 = 1.f, sc_q0 = 1.f, gd0 = 0.f;
  if (has_sc && tid < g.n_load) { const auto sc = gptr(b.lane_scale) + (size_t)inst * 2 * g.n_load; sc_p0 = sc[tid]; sc_q0 = sc[g.n_load + tid]; }
  if (has_delta && tid < g.n_gen) gd0 = gptr(b.lane_gen_delta)[(size_t)inst * g.n_gen + tid];
  if (STAGE) { const auto inj_g = gptr(b.inj) + (size_t)inst * g.n_inj; for (int i = oo.inj_sto_p + tid; i < g.n_inj; i += GW) c.inj[i] = inj_g[i]; }   // storage / shunt set-points
  TopoState ts;
  ts.status = 0; ts.nb = 0; ts.dc_base = false; ts.dc_out = -1; ts.gen_base = false; ts.tc[0] = ts.tc[1] = -1;
  bool reuse = false;                                         // block-uniform
  int n_iter = 0, nb = 0, st = 0, rounds = 0;
  // state of the lane's OWN line (line `tid`: the line loops all map line l to lane l % GW) kept in registers for the launch:
  // thermal limit, protection counter (written back by every step) and the "already counted in this call" flag of K7
  float lim_first = 1e30f, lim_second = 1e30f;
  int ovc_first = 0, inc_first = 0, ovc_second = 0;
  if (tid < g.n_line) { lim_first = gptr(b.thermal_limit)[tid]; ovc_first = gptr(b.overflow_count)[(size_t)inst * g.n_line + tid]; }
  // (at most two lines per lane: the protection counters of both stay in registers for the whole launch and go to HBM with the last step --
  //  every step used to rewrite the lane's counter row, and its rho row although the trajectory carries rho: 1.5 KB per lane and step on 118 substations)
  const bool regs2 = g.n_line <= 2 * GW;
  if (regs2 && tid + GW < g.n_line) { lim_second = gptr(b.thermal_limit)[tid + GW]; ovc_second = gptr(b.overflow_count)[(size_t)inst * g.n_line + tid + GW]; }
  int ep_steps = 0, ep_resets = 0;
  if (tid == 0 && !ghost) { ep_steps = gptr(b.episode)[2 * (size_t)inst]; ep_resets = gptr(b.episode)[2 * (size_t)inst + 1]; }
  // environment injection dynamics (opt-in): lane k of the instance's first LWE lanes carries generator k and storage unit k
  constexpr int LWE = GW < WAVE ? GW : WAVE;
  constexpr bool env_on = ENV && NB == 1;                       // single-busbar kernels (incl. topology classes) only
  EnvRegs er;
  er.target = er.actual = er.prev_p = er.charge = er.amount_prev = er.curt_prev = 0.f; er.limit = 1.f; er.already = false; er.fresh = true;
  er.illegal = 0;
  float env_ar0 = 0.f, env_as0 = 0.f, env_ac0 = -1.f;
  if (env_on && tid < LWE) {
    const EnvDyn& E = P->env;
    if (tid < g.n_gen) {
      const size_t q = (size_t)inst * g.n_gen + tid;
      er.target = gptr(E.target)[q]; er.actual = gptr(E.actual)[q]; er.prev_p = gptr(E.prev_p)[q]; er.already = gptr(E.already)[q] != 0;
      if (E.act_redisp) env_ar0 = gptr(E.act_redisp)[q];
      er.limit = gptr(E.limit)[q];
      if (E.act_curtail) env_ac0 = gptr(E.act_curtail)[q];
    }
    if (tid < g.n_sto) {
      er.charge = gptr(E.charge)[(size_t)inst * g.n_sto + tid];
      if (E.act_storage) env_as0 = gptr(E.act_storage)[(size_t)inst * g.n_sto + tid];
    }
    er.amount_prev = gptr(E.amount_prev)[inst];
    er.curt_prev = gptr(E.curt_prev)[inst];
    er.fresh = gptr(E.fresh)[inst] != 0;
    er.illegal = gptr(E.illegal)[inst];
  }
  bool env_fail = false;                                       // this group's dynamics ended the episode in the current step
  // the chronics values this thread handles first (load tid, generator tid) of the NEXT step are fetched while the current step
  // computes: K9 then starts from registers instead of an L2 / HBM round trip
  // (single-wavefront instances only: the 2-wavefront kernels have no registers to spare -- measured -1.7 % on 118 substations)
  constexpr bool PF9 = WPI == 1;
  float pf_lp = 0.f, pf_lq = 0.f, pf_pp = 0.f, pf_pv = 1.f;
  if (PF9) {
    const auto ch0 = gptr(b.chron) + ((size_t)tab * sa.T + row) * g.n_chron;
    if (tid < g.n_load) { pf_lp = ch0[tid]; pf_lq = ch0[g.n_load + tid]; }
    if (tid < g.n_gen) { pf_pp = ch0[2 * g.n_load + tid]; pf_pv = ch0[2 * g.n_load + g.n_gen + tid]; }
  }
  // ---- kept state of the reference topology, second half: key comparison, state -> LDS, verdict of the block ----------------------------------
  bool keep_hit = false, keep_dirty = false;                    // block-uniform
  bool keep_claim = false;                                      // this instance is on the key and the blob has no state yet
  GPF_STAMPS(34);
  if (keep_on) {
    const auto kb = (GPF_GLOBAL const unsigned char*)sa.keep.p + (YR ? (size_t)sa.keep.stride : 0);
    const auto kt = (GPF_GLOBAL const int*)kb + KEEP_HDR_INTS;
    bool same = k_same;
#pragma unroll
    for (int u = 0; u < KT; ++u) same &= kk[u] == tn[u];
    {                                                           // (rows longer than the unrolled trips: grids beyond the shipped ones)
      const auto topo_now = gptr(b.topo) + (size_t)inst * g.dim_topo;
      const auto shb_now = gptr(b.shunt_bus) + (size_t)inst * g.n_shunt;
      const auto kd = (GPF_GLOBAL const double*)(kb + sa.keep.off_kd);
      const auto inj_now = gptr(b.inj) + (size_t)inst * g.n_inj + oo.inj_sh_p;
      for (int i = tid + KT * GW; i < g.dim_topo; i += GW) same &= kt[i] == topo_now[i];
      for (int i = tid + GW; i < g.n_shunt; i += GW) same &= kt[g.dim_topo + i] == shb_now[i];
      for (int i = tid + 2 * GW; i < 2 * g.n_shunt; i += GW) same &= kd[i] == inj_now[i];
    }
    GPF_STAMPS(36);
    // a state is trusted when an EARLIER launch wrote it (every store of that kernel is visible; the writer of this launch may be half way)
    const bool valid = k_h0 > KEEP_VALID && k_h0 - KEEP_VALID < sa.keep.launch;
    // ---- the state goes to LDS whatever the comparison says: a lane that misses rebuilds every one of these arrays from scratch ---------------
    {
      const auto km16 = (GPF_GLOBAL const v4i_*)(kb + sa.keep.off_m);
      const auto ky = (GPF_GLOBAL const v2d_*)(kb + sa.keep.off_y);
      const auto ka = (GPF_GLOBAL const double*)(kb + sa.keep.off_d);
      int* const img = c.btype;                                  // btype | vidx | element -> bus maps | sub_bb: one contiguous range of the carve (8-byte aligned)
      v2i_* const img2 = reinterpret_cast<v2i_*>(img);
      const int n_m = sa.keep.n_m;                               // ints of the range (the blob is padded to whole 16-byte chunks, the carve is not)
      auto put4 = [&](int ch, const v4i_ t_) {
        if (4 * ch + 3 < n_m) { img2[2 * ch] = v2i_{t_.x, t_.y}; img2[2 * ch + 1] = v2i_{t_.z, t_.w}; }
        else { if (4 * ch < n_m) img[4 * ch] = t_.x; if (4 * ch + 1 < n_m) img[4 * ch + 1] = t_.y; if (4 * ch + 2 < n_m) img[4 * ch + 2] = t_.z; }
      };
      const int n_ch = (n_m + 3) >> 2;
#pragma unroll
      for (int u = 0; u < KC; ++u) { const int ch = tid + u * GW; if (ch < n_ch) put4(ch, im[u]); }
      for (int ch = tid + KC * GW; ch < n_ch; ch += GW) put4(ch, km16[ch]);
      if (!YR) {
        double2* const yl = reinterpret_cast<double2*>(c.Yb);
#pragma unroll
        for (int u = 0; u < KY; ++u) { const int i = tid + u * GW; if (i < S.nslot_y) yl[i] = yy[u]; }
        for (int i = tid + KY * GW; i < S.nslot_y; i += GW) { const v2d_ t_ = ky[i]; yl[i] = make_double2(t_.x, t_.y); }
      }
      if (keep_dcf && (k_h2 & 4)) {                              // (a blob written from the static-inverse DC start holds no factors)
#pragma unroll
        for (int u = 0; u < KA; ++u) { const int q = tid + u * GW; if (q < S.nslot) c.Adc[q] = aa[u]; }
        for (int q = tid + KA * GW; q < S.nslot; q += GW) c.Adc[q] = ka[q];
      }
    }
    GPF_STAMPS(37);
    // every instance of the block must hit; the DC start of the block is the static inverse for all (no factors needed) or the kept factors
    const unsigned hb = G::template any_bits<3, 4>(((same && valid) ? 0u : 1u) | ((k_h2 & 1) ? 0u : 2u) | ((k_h2 & 4) ? 0u : 4u) | (same ? 0u : 8u));
    const unsigned hbb = IPW == 1 ? hb : ((__any(hb & 1u) ? 1u : 0u) | (__any(hb & 2u) ? 2u : 0u) | (__any(hb & 4u) ? 4u : 0u));
    keep_hit = !(hbb & 1u) && (!keep_dcf || !(hbb & 2u) || !(hbb & 4u));
    keep_claim = !(hb & 8u) && k_h0 == KEEP_KEYED && !ghost;
    if (keep_hit) {
      ts.status = k_h1 & 0xff; ts.nb = k_h1 >> 8; ts.dc_base = (k_h2 & 1) != 0; ts.gen_base = (k_h2 & 2) != 0; ts.dc_out = (k_h2 >> 3) - 1;
      reuse = true;
    }
    GPF_STAMPS(38);
    GPF_LSYNC();                                   // (a miss: K1 initialises these arrays with another lane mapping)
  }
  GPF_STAMPS(39);
  // Every step (and every cascade round) runs the same code on the same addresses, so the compiler would hoist each per-thread
  // pointer, offset and table entry it finds out of the loops (loop-invariant code motion) and keep them live for the whole
  // launch: > 250 VGPRs plus scratch spills.  GPF_REDERIVE makes the thread's coordinates opaque and re-derives the LDS carve
  // from them, so that what is live at a loop head is only what the loop really carries.
#define GPF_REDERIVE()                                                                                              \
  do {                                                                                                              \
    grp = grp0; tid = tid0; inst = inst0;                                                                           \
    asm volatile("" : "+v"(grp), "+v"(tid), "+v"(inst));                                                            \
    carve_sparse<NB>(c, smem + (size_t)grp * per_inst, g, lds_nslot, lds_nslot_y, STAGE != 0, lds_rows, lds_dcf);   \
  } while (0)
  // per-step observation trajectory (block-uniform): every step's rows go to [step][lane] of Bufs::traj_*, else to the lane's rows
  const bool tobs = b.traj_out != nullptr;
  for (int step = 0; step < sa.n_steps; ++step) {
    GPF_REDERIVE();
    const bool last = step + 1 == sa.n_steps;
    const bool otraj = tobs && step < b.traj_cap;
    int orow = otraj ? step * (int)b.lane_stride + inst : inst;
    GPF_STAMPS(19);
    // ---- K9: chronics row -> injections -----------------------------------------------------------------------------------
    bool sums_in_k9 = false, skip_inj = false;
    float k9_lp0 = 0.f, k9_lq0 = 0.f, k9_lp1 = 0.f, k9_lq1 = 0.f, k9_pp = 0.f, k9_vm = 1.f;
    {
      const auto ch = gptr(b.chron) + ((size_t)tab * sa.T + row) * g.n_chron;
      const auto sc = gptr(b.lane_scale) + (size_t)inst * 2 * g.n_load;
      const auto gdelta = gptr(b.lane_gen_delta) + (size_t)inst * g.n_gen;
      const auto inj_g = gptr(b.inj) + (size_t)inst * g.n_inj;
      // all global loads of the phase are issued up front (one round trip): the lane's topology row (when the topology phases
      // run again) and the first pass of the chronics row
      // scheduled maintenance (Chronics/gridStateFromFile.py maintenance.csv -> the environment's "maintenance" modification,
      // Action/baseAction.py: the line's status is set to -1 at every step of the outage; nothing reconnects it afterwards)
      bool any_maint = false;
      const auto mrow = gptr(b.maint) + ((size_t)tab * sa.T + row) * g.n_line;
      if (b.maint) {
        const auto topo = gptr(b.topo) + (size_t)inst * g.dim_topo;
        bool hit = false;
        if (!ghost)
        for (int l = tid; l < g.n_line; l += GW)
          if (mrow[l] && topo[sv.line_or_pos[l]] != -1) { topo[sv.line_or_pos[l]] = -1; topo[sv.line_ex_pos[l]] = -1; hit = true; }
        any_maint = G::block_any(hit);
        if (any_maint) reuse = false;
      }
      if (!reuse) { const auto topo = gptr(b.topo) + (size_t)inst * g.dim_topo; for (int i = tid; i < g.dim_topo; i += GW) c.topo[i] = topo[i]; }
      if (any_maint) {                       // the LDS copy may have been read before the stores above landed: same fix-up there
        GPF_SYNC();
        for (int l = tid; l < g.n_line; l += GW) if (mrow[l]) { c.topo[sv.line_or_pos[l]] = -1; c.topo[sv.line_ex_pos[l]] = -1; }
      }
      float pp_pre = pf_pp, pv_pre = pf_pv, lp_pre = pf_lp, lq_pre = pf_lq;
      // several wavefronts per instance (the load / generator loops run on wavefront 0 alone, two trips for up to 128 loads): ALL chronics
      // values of the lane -- both trips of the load loop, the generator's -- are requested here, one HBM round trip instead of three
      constexpr bool PRE2 = WPI > 1;
      constexpr int gw9 = WPI > 1 ? WAVE : GW;
      float lp2 = 0.f, lq2 = 0.f, sc2p = 1.f, sc2q = 1.f;
      if (!PF9) {
        pp_pre = tid < g.n_gen ? ch[2 * g.n_load + tid] : 0.f;
        pv_pre = tid < g.n_gen ? ch[2 * g.n_load + g.n_gen + tid] : 1.f;
        if (PRE2 && tid < gw9) {
          if (tid < g.n_load) { lp_pre = ch[tid]; lq_pre = ch[g.n_load + tid]; }
          if (tid + gw9 < g.n_load) {
            lp2 = ch[tid + gw9]; lq2 = ch[g.n_load + tid + gw9];
            if (has_sc) { sc2p = sc[tid + gw9]; sc2q = sc[g.n_load + tid + gw9]; }
          }
        }
      }
      if (PF9 && !last) {                           // next step's first values: in flight during this step's power flow
        const int row_n = row + 1 >= sa.T ? 0 : row + 1;
        const auto chn = gptr(b.chron) + ((size_t)tab * sa.T + row_n) * g.n_chron;
        if (tid < g.n_load) { pf_lp = chn[tid]; pf_lq = chn[g.n_load + tid]; }
        if (tid < g.n_gen) { pf_pp = chn[2 * g.n_load + tid]; pf_pv = chn[2 * g.n_load + g.n_gen + tid]; }
      }
      double sum_load = 0.0, sum_prod = 0.0;
      // The element -> bus maps stand (reuse): every element adds its new set-point to the bus sums Psp / Qsp / Gs right here
      // (the same LDS atomics K1 would issue from four more loops over the injection row, SolveCtl::sums_done)
      // (a first step that runs on the kept state of the reference topology -- KeepArgs -- leaves the sums to K1: its accumulation order is the
      //  one of a launch that rebuilds, so the launch's results do not depend on whether the blob was there: bit-identical either way)
      sums_in_k9 = reuse;
      // the chronics-driven injections stay in the owner lanes' registers instead of going to the lane's injection row and coming back (SolveCtl::inj_regs):
      // a step whose topology stands (nobody reads the row: K9 has the bus sums), not the last of the launch (the row holds the last step's values
      // for the API), no cascade (a re-solve after a trip rebuilds the sums FROM the row), no injection dynamics
      skip_inj = WPI > 1 && !STAGE && sums_in_k9 && !last && !env_on && sa.cascade == 0 && g.n_load <= 2 * WAVE && g.n_gen <= WAVE;
      const double inv_sn9 = g.inv_sn_mva;
      if (sums_in_k9) {
        const int nbus9 = TC ? S.n : g.n_sub * NB;
        for (int i = tid; i < nbus9; i += GW) { c.Psp[i] = 0.0; c.Qsp[i] = 0.0; c.Gs[i] = 0.0; }
        GPF_LSYNC();
      }
      GPF_STAMPS(16);
      // several wavefronts per instance: the loops below (which also accumulate the bus sums when the maps stand) run on wavefront 0
      // alone -- fixed order of the LDS atomics and of the load / generation totals: bitwise reproducibility, see solve_instance_sparse
      const bool lane9 = tid < gw9;
      if (lane9)
      for (int i = tid; i < g.n_load; i += gw9) {
        const bool i0_ = (PF9 || PRE2) && i == tid, i1_ = PRE2 && i == tid + gw9;
        float lp = i0_ ? lp_pre : i1_ ? lp2 : ch[i], lq = i0_ ? lq_pre : i1_ ? lq2 : ch[g.n_load + i];
        if (has_sc) { lp *= (i == tid) ? sc_p0 : i1_ ? sc2p : sc[i]; lq *= (i == tid) ? sc_q0 : i1_ ? sc2q : sc[g.n_load + i]; }
        if (STAGE) { c.inj[oo.inj_load_p + i] = (double)lp; c.inj[oo.inj_load_q + i] = (double)lq; }   // HBM copy: end of kernel
        else if (!skip_inj) { inj_g[oo.inj_load_p + i] = (double)lp; inj_g[oo.inj_load_q + i] = (double)lq; }
        if (i == tid) { k9_lp0 = lp; k9_lq0 = lq; } else if (i1_) { k9_lp1 = lp; k9_lq1 = lq; }
        sum_load += (double)lp;
        if (sums_in_k9) {
          const int bu = c.load_b[i];
          if (bu >= 0) { atomicAdd(&c.Psp[bu], -(double)lp * inv_sn9); atomicAdd(&c.Qsp[bu], -(double)lq * inv_sn9); }
        }
      }
      if (lane9)
      for (int i = tid; i < g.n_gen; i += gw9)
        if (!sv.gen_slack[i]) sum_prod += (double)(i == tid ? pp_pre : ch[2 * g.n_load + i]);
      float scale_p = 1.0f;
      GPF_STAMPS(17);
      if (sa.rebalance_on) {
        G::template sum2<0>(sum_load, sum_prod);
        scale_p = (sum_prod > 0.0) ? (float)(sa.rebalance * sum_load / sum_prod) : 1.0f;
      }
      GPF_STAMPS(18);
      env_fail = false;
      float env_np = 0.f;
      if (env_on && (WPI == 1 || tid < WAVE)) {                  // (wavefront-uniform: the whole wavefront takes part in the reductions)
        float np_k = tid < g.n_gen ? pp_pre : 0.f;
        if (tid < g.n_gen && !sv.gen_slack[tid]) np_k *= scale_p;
        float sto_pw = 0.f;
        const bool ok_e = env_dynamics_step<LWE>(P->env, tid, g.n_gen, g.n_sto, np_k, step == 0 ? env_ar0 : 0.f,
                                                 (step == 0 || P->env.hold_storage) ? env_as0 : 0.f, step == 0 ? env_ac0 : -1.f, er, sto_pw);
        env_fail = !ok_e;
        env_np = np_k;                                            // the (curtailed) chronics set-point of generator tid
        if (tid < g.n_sto) {                                      // the storage power the backend gets (set_storage, baseEnv.py:3831)
          if (STAGE) c.inj[oo.inj_sto_p + tid] = (double)sto_pw;
          else inj_g[oo.inj_sto_p + tid] = (double)sto_pw;
        }
      }
      if (lane9)
      for (int i = tid; i < g.n_gen; i += gw9) {
        float pp = (i == tid) ? pp_pre : ch[2 * g.n_load + i];
        if (!sv.gen_slack[i]) pp *= scale_p;
        if (env_on) pp = env_np + er.actual;                              // set_redispatch (:3830): chronics + actual dispatch, float32 (n_gen <= lanes: i == tid)
        if (has_delta) pp += (i == tid) ? gd0 : gdelta[i];
        const float pv_kv = (i == tid) ? pv_pre : ch[2 * g.n_load + g.n_gen + i];
        const float vn = (float)sv.gen_vn[i];
        const double vm_pu = (double)(pv_kv / vn);
        if (STAGE) { c.inj[oo.inj_gen_p + i] = (double)pp; c.inj[oo.inj_gen_vm + i] = vm_pu; }
        else if (!skip_inj) { inj_g[oo.inj_gen_p + i] = (double)pp; inj_g[oo.inj_gen_vm + i] = vm_pu; }
        if (i == tid) { k9_pp = pp; k9_vm = pv_kv / vn; }
        if (sums_in_k9) {
          const int bu = c.gen_b[i];
          if (bu >= 0 && !sv.gen_slack[i]) atomicAdd(&c.Psp[bu], (double)pp * inv_sn9);
        }
      }
      GPF_STAMPS(29);
      if (sums_in_k9 && lane9) {                     // storage and shunt set-points do not change during a launch
        for (int i = tid; i < g.n_sto; i += gw9) {
          const int bu = c.sto_b[i];
          if (bu >= 0) {
            const double sp = STAGE ? c.inj[oo.inj_sto_p + i] : (double)inj_g[oo.inj_sto_p + i];
            const double sq = STAGE ? c.inj[oo.inj_sto_q + i] : (double)inj_g[oo.inj_sto_q + i];
            atomicAdd(&c.Psp[bu], -sp * inv_sn9);
            atomicAdd(&c.Qsp[bu], -sq * inv_sn9);
          }
        }
        for (int i = tid; i < g.n_shunt; i += gw9) {
          const int bu = c.sh_b[i];
          if (bu >= 0) {
            const double hp = STAGE ? c.inj[oo.inj_sh_p + i] : (double)inj_g[oo.inj_sh_p + i];
            atomicAdd(&c.Gs[bu], hp * sv.shunt_fact[i] * inv_sn9);
          }
        }
      }
      GPF_SYNC_IF(!STAGE && !skip_inj);            // tier 0: the injection row went to global memory and is read back from there
      GPF_STAMPS(30);
    }
    // ---- power flow + K7 (Backend.next_grid_state) --------------------------------------------------------------------------
    n_iter = 0; nb = 0; st = 0; rounds = 0;
    float a_first = 0.f;                                        // a_or of line `tid` from the last solve of this step
    {
      const auto dround = gptr(b.disc_round) + (size_t)inst * g.n_line;
      for (int l = tid; l < g.n_line; l += GW) dround[l] = -1;
      // Backend.next_grid_state keeps a LOCAL copy of the protection counters that is advanced at most once per line
      // and per call (backend.py:1476-1520): local value = ovc + (line already counted this call ? 1 : 0); the "already
      // counted" flag lives in the rho buffer, reused as int scratch until the end of the step.
      const auto inc_flag = (GPF_GLOBAL int*)(gptr(b.rho) + (size_t)inst * g.n_line);
      if (sa.cascade) for (int l = tid + GW; l < g.n_line; l += GW) inc_flag[l] = 0;      // (line tid: inc_first)
      inc_first = 0;
    }
    if (!reuse) keep_dirty = true;                              // this step rebuilds the topology-derived state
    bool more = true;                                           // this group still cascades
    bool first = true;
    bool tripped = false;                                       // this group tripped a line during this step
    while (true) {
      GPF_REDERIVE();
      // a group whose cascade has ended re-solves its unchanged state along with the others (same results)
      int it_k = 0, nb_k = 0;
      SolveCtl ctl;
      ctl.inj_staged = true; ctl.topo_staged = first; ctl.reuse = first && reuse; ctl.dcf = P->dcf != 0 && (sa.n_steps > 1 || keep_on); ctl.write_bus = last; ctl.warm = sa.warm_start != 0 && !(keep_hit && step == 0); ctl.sums_done = first && sums_in_k9;   // (kept state holds no voltages to start from)
      orow = otraj ? step * (int)b.lane_stride + inst : inst;          // (re-derived: see GPF_REDERIVE)
      ctl.otraj = otraj; ctl.orow = orow; ctl.write_topo = otraj || (keep_hit && step == 0);   // (kept state: the lane's topology outputs may be another launch's)
      ctl.tc_rebuild = keep_hit && step == 0;
      ctl.inj_regs = skip_inj; ctl.r_lp0 = k9_lp0; ctl.r_lq0 = k9_lq0; ctl.r_lp1 = k9_lp1; ctl.r_lq1 = k9_lq1; ctl.r_pp = k9_pp; ctl.r_vm = k9_vm;
      GPF_STAMPS(31);
      const int st_k = solve_instance_sparse<NB, STAGE, IPW, WPI, TC, YR>(P, S, FL, sv, c, yreg, rcreg, inst, sa.is_dc, max_iter, tol_pu, tid, ctl, ts, it_k, nb_k, a_first GPF_STAMPS_ARG);
      first = false;
      GPF_SYNC_IF(sa.cascade != 0 && g.n_line > GW);   // (every line loop maps line l to lane l % GW: lanes read their own rows)
      if (more) { st = st_k; n_iter = it_k; nb = nb_k; }
      if (st != 0 || !sa.cascade || rounds >= sa.max_rounds) more = false;   // at most max_rounds re-solves
      int any_disc = 0;
      if (more && !ghost) {
        const auto out = gptr(otraj ? b.traj_out : b.out) + (size_t)orow * g.n_out;
        const auto ovc = gptr(b.overflow_count) + (size_t)inst * g.n_line;
        const auto dround = gptr(b.disc_round) + (size_t)inst * g.n_line;
        const auto inc_flag = (GPF_GLOBAL int*)(gptr(b.rho) + (size_t)inst * g.n_line);
        const auto topo = gptr(b.topo) + (size_t)inst * g.dim_topo;
        const auto thermal_limit = gptr(b.thermal_limit);
        for (int l = tid; l < g.n_line; l += GW) {
          const bool own = l == tid, own2 = regs2 && l == tid + GW;
          const float a = own ? a_first : (c.out_l ? c.out_l[oo.a_or + l] : (float)out[oo.a_or + l]);     // (staged row: the lane's value is in LDS)
          const float lim = own ? lim_first : own2 ? lim_second : (float)thermal_limit[l];
          const bool on = c.lor_b[l] >= 0;
          bool disc = on && (a > sa.hard_overflow * lim);
          int inc = own ? inc_first : (int)inc_flag[l];
          if (on && (a > sa.soft_overflow * lim) && !inc) { inc = 1; if (own) inc_first = 1; else inc_flag[l] = 1; }
          if (on && ((own ? ovc_first : own2 ? ovc_second : (int)ovc[l]) + inc) > sa.nb_ts_allowed) disc = true;
          if (disc) {
            topo[sv.line_or_pos[l]] = -1;
            topo[sv.line_ex_pos[l]] = -1;
            dround[l] = rounds;
            any_disc = 1;
          }
        }
      }
      const bool anyd = more && G::any(any_disc);
      GPF_SYNC_IF(G::block_any_u(anyd));           // a tripped line rewrote the topology row the next round stages with another lane mapping
      if (more && !anyd) more = false;
      if (more) tripped = true;
      if (!G::block_any_u(more)) break;
      if (more) ++rounds;
    }
    GPF_STAMPS(9);
    GPF_REDERIVE();
    if (env_on) {                                              // ImpossibleRedispatching ends the episode (baseEnv.py:3227-3247)
      const bool ef = WPI > 1 ? (__syncthreads_or(env_fail) != 0) : env_fail;
      if (ef && st == 0) st = 6;
    }
    orow = otraj ? step * (int)b.lane_stride + inst : inst;
    // ---- per-step outputs -----------------------------------------------------------------------------------------------------
    if (st != 0) { write_nan_results<GW>(g, b, inst, tid, orow, otraj); a_first = __builtin_nanf(""); }
    GPF_SYNC_IF(G::block_any_u(st != 0));          // NaN rows are written with another lane mapping than the line loop's
    {
      const auto out = gptr(otraj ? b.traj_out : b.out) + (size_t)orow * g.n_out;
      const auto ovc = gptr(b.overflow_count) + (size_t)inst * g.n_line;
      const auto rho = gptr(b.rho) + (size_t)inst * g.n_line;
      const auto thermal_limit = gptr(b.thermal_limit);
      GPF_GLOBAL float* traj = nullptr;
      if (b.traj_rho && step < b.traj_cap) traj = gptr(b.traj_rho) + ((size_t)step * b.lane_stride + inst) * g.n_line;
      for (int l = tid; l < g.n_line; l += GW) {
        const bool own = l == tid, own2 = regs2 && l == tid + GW;
        const float lim = own ? lim_first : own2 ? lim_second : (float)thermal_limit[l];
        const float a = own ? a_first : ((st == 0 && c.out_l) ? c.out_l[oo.a_or + l] : (float)out[oo.a_or + l]);   // (this lane wrote out[a_or + l] itself -- to the staged row in LDS or to HBM)
        const float r_ = a / lim;
        if (!traj || last || sa.cascade) rho[l] = r_;        // (the lane's own row = the last step; with the cascade on it doubles as the rounds' flag row)
        if (traj) traj[l] = r_;
        if (!ghost) {
          const int prev = own ? ovc_first : own2 ? ovc_second : (int)ovc[l];
          const int now = (a > sa.soft_overflow * lim) ? prev + 1 : 0;
          if (own) ovc_first = now; else if (own2) ovc_second = now;
          if (!regs2 || last) ovc[l] = now;
        }
      }
      // line cooldowns (obs.time_before_cooldown_line): one step passed; a line the protections tripped in this step starts its
      // reconnection cooldown; a maintenance / hazard under way holds the counter at its remaining duration (baseEnv.py:3352-3358,
      // 2590-2597: only on a converged step -- a failed one ends the episode)
      if (sa.nb_ts_reco >= 0 && st == 0 && !ghost) {
        const auto cool = gptr(b.cooldown) + (size_t)inst * g.n_line;
        const auto dround = gptr(b.disc_round) + (size_t)inst * g.n_line;
        const auto mdur = gptr(b.maint_dur) + ((size_t)tab * sa.T + row) * g.n_line;
        GPF_GLOBAL short* tcool = nullptr;
        if (b.traj_cool && step < b.traj_cap) tcool = gptr(b.traj_cool) + ((size_t)step * b.lane_stride + inst) * g.n_line;
        for (int l = tid; l < g.n_line; l += GW) {
          int cd = cool[l];
          cd = cd > 0 ? cd - 1 : 0;
          if (dround[l] >= 0) cd = sa.nb_ts_reco;
          if (b.maint_dur) { const int md = (int)mdur[l]; cd = md > cd ? md : cd; }
          cool[l] = cd;
          if (tcool) tcool[l] = (short)(cd > 32767 ? 32767 : cd);
        }
      }
    }
    const bool failed = st != 0;
    if (tid == 0) {
      const auto s = gptr(b.status) + (size_t)inst * 4;
      s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = rounds;
      if (!ghost) {
        gptr(b.done)[inst] = failed ? 1 : 0;
        if (b.traj_status && step < b.traj_cap) gptr(b.traj_status)[(size_t)step * b.lane_stride + inst] = (signed char)st;
        if (failed) { if (sa.auto_reset) { ep_steps = 0; ++ep_resets; } } else ++ep_steps;
      }
    }
    // game over + auto-reset: the lane restarts from the topology the host sent last, protection counters cleared (the
    // chronics cursor keeps running: the next step is the first of a new episode)
    if (failed && sa.auto_reset && !ghost) {
      const auto t0 = gptr(b.topo0) + (size_t)inst * g.dim_topo;
      const auto topo = gptr(b.topo) + (size_t)inst * g.dim_topo;
      const auto ovc = gptr(b.overflow_count) + (size_t)inst * g.n_line;
      for (int i = tid; i < g.dim_topo; i += GW) topo[i] = t0[i];
      for (int l = tid; l < g.n_line; l += GW) ovc[l] = 0;
      if (sa.nb_ts_reco >= 0) { const auto cool = gptr(b.cooldown) + (size_t)inst * g.n_line; for (int l = tid; l < g.n_line; l += GW) cool[l] = 0; }   // env.reset(): baseEnv.py:3979
      ovc_first = 0; ovc_second = 0;
      if (env_on) {                                              // env.reset(): dispatch cleared, storage back to its initial charge
        er.target = er.actual = er.prev_p = er.amount_prev = er.curt_prev = 0.f; er.limit = 1.f; er.already = false; er.fresh = true;
        er.illegal = 0;
        er.charge = (tid < g.n_sto && P->env.charge0) ? gptr(P->env.charge0)[tid] : 0.f;
      }
    }
    // the topology-derived state stands for the next step only if NO group of the block changed or lost its topology
    reuse = !G::block_any_u(failed || tripped);
    GPF_STAMPS(7);
    if (++row >= sa.T) row = 0;
    if (!last) GPF_SYNC_IF(G::block_any_u(failed && sa.auto_reset != 0));   // auto-reset rewrote topology rows with another lane mapping
  }
#undef GPF_REDERIVE
  grp = grp0; tid = tid0; inst = inst0;
  if (keep_on) {
    // The blob has a key but no state yet, this instance is on the key, rebuilt the state in this launch and nothing tripped or failed since
    // (`reuse`: LDS / registers hold the state of the lane's rows, which still equal the key): the first instance to claim the blob writes it.
    const auto kb = (GPF_GLOBAL unsigned char*)sa.keep.p + (YR ? (size_t)sa.keep.stride : 0);
    const auto hdr = (GPF_GLOBAL int*)kb;
    bool won = false;
    if (keep_claim && reuse && keep_dirty && tid == 0) won = atomicCAS((int*)hdr, KEEP_KEYED, KEEP_CLAIMED) == KEEP_KEYED;
    const bool all_dcb = G::block_all_u(ts.dc_base);     // (what solve_instance_sparse decided on: dc_inv)
    if (G::any(won)) {
      typedef double v2d_ __attribute__((ext_vector_type(2)));
      const auto km = (GPF_GLOBAL int*)(kb + sa.keep.off_m);
      const int* const img = c.btype;
      for (int i = tid; i < sa.keep.n_m; i += GW) km[i] = img[i];
      const auto ky = (GPF_GLOBAL v2d_*)(kb + sa.keep.off_y);
      auto st2 = [&](int i_, const double2 v_) { v2d_ t_; t_.x = v_.x; t_.y = v_.y; ky[i_] = t_; };
      if (YR) {
#pragma unroll
        for (int k = 0; k < YR_PASSES; ++k) {
          const int pr = tid + k * GW;
          if (pr < S.n_up) { st2(2 * pr, yreg[2 * k]); st2(2 * pr + 1, yreg[2 * k + 1]); }
        }
        if (tid < g.n_sub) st2(2 * S.n_up + tid, yreg[2 * YR_PASSES]);
      } else {
        const double2* const yl = reinterpret_cast<const double2*>(c.Yb);
        for (int i = tid; i < S.nslot_y; i += GW) st2(i, yl[i]);
      }
      // the factors are in LDS unless every instance of the block started from the static DC inverse (solve_instance_sparse: dc_inv)
      const bool adc = keep_dcf && !all_dcb;
      if (adc) { const auto ka = (GPF_GLOBAL double*)(kb + sa.keep.off_d); for (int q = tid; q < S.nslot; q += GW) ka[q] = c.Adc[q]; }
      if (tid == 0) {
        hdr[1] = (ts.status & 0xff) | (ts.nb << 8);
        hdr[2] = (ts.dc_base ? 1 : 0) | (ts.gen_base ? 2 : 0) | (adc ? 4 : 0) | ((ts.dc_out + 1) << 3);
        hdr[0] = KEEP_VALID + sa.keep.launch;     // (trusted by later launches only: no ordering needed against the stores above)
      }
    }
  }
  if (tobs) {
    // the getters / device views of the lane's own rows return the LAST step: copy its trajectory rows there
    const int ls = sa.n_steps <= b.traj_cap ? sa.n_steps - 1 : b.traj_cap - 1;
    const size_t r = (size_t)ls * b.lane_stride + inst;
    GPF_SYNC();                                    // rows written with the element / line lane mappings, read back coalesced
    { const auto s_ = gptr(b.traj_out) + r * g.n_out; const auto d_ = gptr(b.out) + (size_t)inst * g.n_out; for (int i = tid; i < g.n_out; i += GW) d_[i] = s_[i]; }
    { const auto s_ = gptr(b.traj_topo) + r * g.dim_topo; const auto d_ = gptr(b.topo_out) + (size_t)inst * g.dim_topo; for (int i = tid; i < g.dim_topo; i += GW) d_[i] = s_[i]; }
    { const auto s_ = gptr(b.traj_shb) + r * g.n_shunt; const auto d_ = gptr(b.shunt_bus_out) + (size_t)inst * g.n_shunt; for (int i = tid; i < g.n_shunt; i += GW) d_[i] = s_[i]; }
    { const auto s_ = gptr(b.traj_lstat) + r * g.n_line; const auto d_ = gptr(b.line_status) + (size_t)inst * g.n_line; for (int i = tid; i < g.n_line; i += GW) d_[i] = s_[i]; }
  }
  if (tid == 0 && !ghost) { gptr(b.episode)[2 * (size_t)inst] = ep_steps; gptr(b.episode)[2 * (size_t)inst + 1] = ep_resets; }
  if (STAGE && !ghost) {                                      // the last step's injection row -> HBM (gpf_get_injections, next launches)
    const auto inj_g = gptr(b.inj) + (size_t)inst * g.n_inj;
    const int n_wb = env_on ? oo.inj_sto_q : oo.inj_sto_p;    // (the dynamics also move the storage power)
    for (int i = tid; i < n_wb; i += GW) inj_g[i] = c.inj[i];
  }
  if (env_on && tid < LWE && !ghost) {
    const EnvDyn& E = P->env;
    if (tid < g.n_gen) {
      const size_t q = (size_t)inst * g.n_gen + tid;
      gptr(E.target)[q] = er.target; gptr(E.actual)[q] = er.actual; gptr(E.prev_p)[q] = er.prev_p; gptr(E.already)[q] = er.already ? 1 : 0;
      gptr(E.limit)[q] = er.limit;
    }
    if (tid < g.n_sto) gptr(E.charge)[(size_t)inst * g.n_sto + tid] = er.charge;
    if (tid == 0) {
      gptr(E.amount_prev)[inst] = er.amount_prev; gptr(E.curt_prev)[inst] = er.curt_prev; gptr(E.fresh)[inst] = er.fresh ? 1 : 0;
      gptr(E.illegal)[inst] = er.illegal;
    }
  }
  GPF_STAMPS(15);
  GPF_STAMPS_FLUSH(inst);
}

}  // namespace gpf

// gridpf_symbolic.hpp -- host-side symbolic analysis for the block-sparse kernels (gridpf_sparse.hpp).
//
// The Newton Jacobian and the DC matrix of a grid have the sparsity of its SUBSTATION graph when they are stored
// as dense blocks per (substation, substation) pair (block = all busbars x {theta, |V|} of the two substations):
// bus splits, line outages, PV/PQ/reference changes only change VALUES inside blocks (rows / columns of inactive
// or fixed variables become identity).  The symbolic factorisation is therefore computed ONCE per grid at
// gpf_create and shared by every lane and every topology:
//   * LEVEL-SCHEDULED minimum-degree ordering: each level is an independent set of low-degree substations that are
//     eliminated concurrently (their trailing updates may hit the same block -> the device uses LDS f64 atomics);
//   * fill pattern, block slots;
//   * a flat int32 "program" (level headers + packed item lists) that the device streams with coalesced loads.
//
// (The reference's solver is sparse too: pandapower calls scipy.sparse.linalg.spsolve / KLU per Newton iteration,
// grid2op/Backend/pandaPowerBackend.py:1081-1083.)
#pragma once
#include <algorithm>
#include <cstdint>
#include <set>
#include <vector>

namespace gpf {

struct Symbolic {
  int n = 0;                       // number of substations (block rows)
  int nslot = 0;                   // blocks of L+U including fill; slots [0, nslot_y) = original pattern (diag first)
  int nslot_y = 0;
  int n_levels = 0;
  std::vector<int> slot_row, slot_col;   // [nslot]
  std::vector<int> br_slot;        // [n_line][4] slots of the (ff, ft, tf, tt) blocks of every branch
  // flat program (all offsets are indices into `prog`):
  //   prog[0 .. 8*n_levels)            level headers {piv_off, n_piv, b_off, n_b, c_off, n_c, r_off, n_r}
  //   pivots      : substation ids
  //   b-items     : one int per U block          (pivot_sub << 16) | u_slot        -> U' = Dinv * A
  //   c-items     : two ints per trailing update  dst | (l_slot << 16), u_slot | (pivot_sub << 16)
  //                                                 -> A[dst] -= A[l] * Dinv[p] * A[u]
  //   r-items     : two ints per L block          l_slot | (l_row << 16), pivot_sub -> rhs[l_row] -= A[l] * b'[p]
  //                 (the r-items of a level directly follow its c-items: r_off == c_off + 2 * n_c, the 2x2 device path
  //                  walks both as one list)
  //   back section: level table [n_levels]{ent_off, n_ent}; entries two ints  u_slot | (u_col << 16), pivot_sub
  //                 -> x[pivot] -= U'[u_slot] * x[u_col]   (all entries of a level run concurrently, LDS atomics)
  std::vector<int> prog;
  int scale_off = 0, n_scale = 0;  // all U blocks of all levels: (pivot_sub << 16) | u_slot  (deferred U' = Dinv * A pass)
  int back_first = -1;             // highest level with U entries (-1: none)
  int back_off = 0;                // offset of the back-substitution level table
  int max_level_piv = 0;
  int rslot0 = 0;                  // first right-hand-side pseudo-slot of the flat program (FlatProg): max(nslot, ceil(1.5 n))
  // ---- Gauss-Jordan tail of the FLAT programs (build_flat; the level-header program above stays plain LU) --------------------------
  // The last levels of the elimination hold a handful of pivots each, yet every level costs the 2x2 sweeps one forward AND one back
  // phase.  From level gj_lv0 on, a pivot's column is therefore also eliminated from the tail rows ABOVE it (same item:
  // A_ij -= A_ik inv(D_k) A_kj, s_i -= A_ik inv(D_k) s_k for tail rows i of earlier levels with a block (i, k)): after the forward
  // passes every tail row reads D_k x_k = s_k and the tail needs NO back substitution -- one phase less per tail level and solve.
  // The rows above take fill the LU does not have (slots [nslot_lu, nslot)); gj_lv0 minimises the passes of the grid's usual group
  // width under a fill budget of nslot_lu / 10 blocks.
  int nslot_lu = 0;                // blocks of the plain LU (what the level-header program touches); nslot - nslot_lu = GJ fill
  int gj_lv0 = 0;                  // first tail level (n_levels: no tail)
  std::vector<int> gj_off;         // [n_levels + 1] offsets into gj_c / gj_r per level (pairs of ints)
  std::vector<int> gj_c, gj_r;     // extra items of the tail levels, encoded like the c-items / r-items of the level headers
  std::vector<int> gj_roff;
  std::vector<int> level_of;       // [n] elimination level of every substation
};

// FLAT program of the 2x2 / scalar sweeps for one group width GW (threads per instance).  The block array of an instance holds
// the right-hand side as n extra PSEUDO-SLOTS (slot rslot0 + p: rows (b_p[0], pad) and (b_p[1], pad)), so that a right-hand-side
// update is the same item as a trailing update (A[dst] -= A[l] * inv(D_p) * A[u] with dst / u pseudo-slots: the pad column only
// ever feeds pad columns) and b' = inv(D) b is one more scaling item.  Every sweep is a list of PASSES of exactly GW items --
// lane t of the instance executes item t of every pass, a level that has more than GW items takes several passes, short levels
// are padded with 0xFFFFFFFF words -- so the device walks `words + pass * stride + t` without level headers, bounds or
// item-kind selects.  Slot fields are BYTE offsets (slot * 16) into one row half of the block array.
//   forward : 2 words per item  dst | (l << 16), u | (pivot << 16)            (a barrier-delimited phase per pass)
//   back    : 2 words per item  u | (x_col pseudo-slot << 16), dst pseudo-slot (a phase per pass, levels in reverse):
//             s_dst -= A_u * inv(D_col) * s_col  -- U and the right-hand side are never scaled: the item applies the inverse of the
//             column's pivot block (slot `col`: field - rhs_field0) to the column's ACCUMULATED right-hand side s_col itself, as
//             every forward item recomputes its pivot inverse; the solution x_p = inv(D_p) s_p is formed by whoever consumes it
//             (the Newton update / the DC angle extraction of gridpf_sparse.hpp).  No scaling pass: two phases less per solve.
// The forward and back sections end with one extra all-invalid pass (the device prefetches the words of the next pass).
struct FlatProg {
  int gw = 0;
  bool wave_closed = true;                                     // gw > 64: every destination of a pass is written by ONE wavefront (below)
  // gw > 64: bit k set = every item of forward / back pass k sits in lanes [0, 64) -- a SOLO pass: wavefront 0 runs it alone, and a run of
  // consecutive solo passes needs no workgroup barrier in between (the LDS executes the operations of one wavefront in issue order).
  // Passes beyond the 32nd count as shared.
  unsigned solo_fwd = 0, solo_back = 0;
  int n_fwd = 0, n_scale = 0, n_scale_rhs = 0, n_back = 0;     // passes
  int scale_off = 0, back_off = 0;                             // int offsets of the sections (forward starts at 0)
  int rhs_field0 = 0;                                          // rslot0 * 16: fields >= this are right-hand-side pseudo-slots
  std::vector<int> words;
};

// gj_budget: most fill blocks the Gauss-Jordan tail may add (< 0: nslot_lu / 10; the engine lowers it when the blocks would cost a
// resident workgroup per CU, gridpf_capi.hip build_symbolic_resident)
struct FlatProg;
inline FlatProg build_flat(const Symbolic& S, int gw, int lane_opt = 0);
inline int flat_pass_count(const Symbolic& S, int gw);
// resched: 1 = re-schedule the levels on the elimination DAG (below), 0 = keep the greedy levels, -1 = whichever gives fewer passes of
// the grid's usual group width (ties: greedy)
inline Symbolic build_symbolic(int n_sub, int n_line, const int* line_or_sub, const int* line_ex_sub, int degree_slack = 1, int gj_budget = -1,
                               int resched = -1) {
  if (resched < 0) {
    const int gw_nat = n_sub <= 8 ? 16 : n_sub <= 24 ? 32 : n_sub < 64 ? 64 : 128;
    Symbolic A = build_symbolic(n_sub, n_line, line_or_sub, line_ex_sub, degree_slack, gj_budget, 0);
    Symbolic B = build_symbolic(n_sub, n_line, line_or_sub, line_ex_sub, degree_slack, gj_budget, 1);
    return flat_pass_count(B, gw_nat) < flat_pass_count(A, gw_nat) ? B : A;
  }
  Symbolic S;
  S.n = n_sub;
  std::vector<std::set<int>> adj(n_sub);
  for (int l = 0; l < n_line; ++l) {
    const int a = line_or_sub[l], b = line_ex_sub[l];
    if (a != b) { adj[a].insert(b); adj[b].insert(a); }
  }
  std::vector<std::vector<std::pair<int, int>>> row_slots(n_sub);   // row -> (col, slot)
  auto add_slot = [&](int r, int c) -> int {
    for (auto& pr : row_slots[r]) if (pr.first == c) return pr.second;
    const int s = (int)S.slot_row.size();
    S.slot_row.push_back(r);
    S.slot_col.push_back(c);
    row_slots[r].push_back({c, s});
    return s;
  };
  for (int s = 0; s < n_sub; ++s) add_slot(s, s);                    // diagonal first: slot s == (s, s)
  for (int s = 0; s < n_sub; ++s)
    for (int t : adj[s]) add_slot(s, t);
  S.nslot_y = (int)S.slot_row.size();
  S.br_slot.resize((size_t)4 * n_line);
  for (int l = 0; l < n_line; ++l) {
    const int a = line_or_sub[l], b = line_ex_sub[l];
    S.br_slot[4 * l + 0] = add_slot(a, a);
    S.br_slot[4 * l + 1] = add_slot(a, b);
    S.br_slot[4 * l + 2] = add_slot(b, a);
    S.br_slot[4 * l + 3] = add_slot(b, b);
  }
  // ---- level-scheduled minimum-degree elimination -------------------------------------------------------------------
  struct Level {
    std::vector<int> piv, b_items, c_items, r_items;
    std::vector<std::vector<int>> u_entries;   // per pivot: (u_slot << 16) | u_col
  };
  std::vector<Level> levels;
  std::vector<std::set<int>> g = adj;
  std::vector<char> done(n_sub, 0);
  std::vector<int> asap(n_sub, 0);                               // earliest level of every substation (depth in the elimination DAG)
  int remaining = n_sub;
  while (remaining > 0) {
    size_t mind = (size_t)-1;
    for (int s = 0; s < n_sub; ++s) if (!done[s]) mind = std::min(mind, g[s].size());
    // independent set of nodes of degree <= mind + degree_slack, lowest degree first
    std::vector<int> cand;
    for (int s = 0; s < n_sub; ++s) if (!done[s] && g[s].size() <= mind + (size_t)degree_slack) cand.push_back(s);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return g[a].size() < g[b].size(); });
    std::vector<char> blocked(n_sub, 0);
    Level L;
    for (int s : cand) {
      if (blocked[s]) continue;
      L.piv.push_back(s);
      blocked[s] = 1;
      for (int t : g[s]) blocked[t] = 1;
    }
    // symbolic elimination of the whole level (pivots are pairwise non-adjacent, so the order inside is irrelevant)
    for (int p : L.piv) {
      std::vector<int> nb(g[p].begin(), g[p].end());
      std::vector<int> ue;
      for (int j : nb) {
        const int us = add_slot(p, j);
        L.b_items.push_back((p << 16) | us);
        ue.push_back((us << 16) | j);
        const int ls = add_slot(j, p);
        L.r_items.push_back(ls | (j << 16));
        L.r_items.push_back(p);
      }
      L.u_entries.push_back(ue);
      for (int i : nb)
        for (int j : nb) {
          const int dst = add_slot(i, j), ls = add_slot(i, p), us = add_slot(p, j);
          L.c_items.push_back(dst | (ls << 16));
          L.c_items.push_back(us | (p << 16));
        }
    }
    for (int p : L.piv) {
      std::vector<int> nb(g[p].begin(), g[p].end());
      for (int i : nb) {
        g[i].erase(p);
        for (int j : nb) if (i != j) g[i].insert(j);
        asap[i] = std::max(asap[i], asap[p] + 1);            // i is updated by p: it can be eliminated one level after p at the earliest
      }
      g[p].clear();
      done[p] = 1;
      --remaining;
    }
    levels.push_back(std::move(L));
  }
  if (resched == 1)
  // ---- levels re-scheduled on the elimination DAG ------------------------------------------------------------------------------------
  // The greedy pass above fixes the elimination ORDER (hence the fill and every pivot's items) but holds pivots back until their
  // degree is minimal.  Any schedule that keeps a pivot after the pivots that update it is valid with the same items (two pivots
  // that are ready at the same time are never adjacent: one would update the other).  List scheduling on that DAG: a level takes
  // every ready pivot that has no slack left (it sits on a longest chain: the number of levels stays the DAG's height), then fills
  // the passes it has opened anyway -- capacity = its forward items rounded up to the grid's usual group width -- with further ready
  // pivots, largest first; the rest wait.  Fewer, fuller passes: 118 substations 14 -> 13 levels, 36 substations one forward
  // pass less.
  {
    const int gw_nat = n_sub <= 8 ? 16 : n_sub <= 24 ? 32 : n_sub < 64 ? 64 : 128;
    struct Rec { int p; std::vector<int> b, c, r, ue, succ; };
    std::vector<Rec> rec(n_sub);
    std::vector<int> n_pred(n_sub, 0);
    for (const Level& L : levels) {
      size_t cq = 0, rq = 0, bq = 0;
      for (size_t q = 0; q < L.piv.size(); ++q) {
        const int p = L.piv[q];
        Rec& R = rec[p];
        const size_t deg = L.u_entries[q].size();             // items of this pivot: deg b-items, deg r-items, deg^2 c-items (in order)
        R.p = p;
        R.ue = L.u_entries[q];
        R.b.assign(L.b_items.begin() + bq, L.b_items.begin() + bq + deg);
        R.r.assign(L.r_items.begin() + rq, L.r_items.begin() + rq + 2 * deg);
        R.c.assign(L.c_items.begin() + cq, L.c_items.begin() + cq + 2 * deg * deg);
        bq += deg; rq += 2 * deg; cq += 2 * deg * deg;
        for (int ue : R.ue) { R.succ.push_back(ue & 0xffff); ++n_pred[ue & 0xffff]; }      // p updates its neighbours at elimination time
      }
    }
    std::vector<int> tail(n_sub, 0);                           // longest chain below a pivot
    for (int lv = (int)levels.size() - 1; lv >= 0; --lv)
      for (int p : levels[lv].piv) for (int sc : rec[p].succ) tail[p] = std::max(tail[p], tail[sc] + 1);
    int height = 0;
    for (int sidx = 0; sidx < n_sub; ++sidx) height = std::max(height, asap[sidx] + tail[sidx] + 1);
    std::vector<Level> re;
    std::vector<char> placed(n_sub, 0);
    std::vector<int> ready;
    for (int sidx = 0; sidx < n_sub; ++sidx) if (n_pred[sidx] == 0) ready.push_back(sidx);
    int n_placed = 0;
    auto load = [&](int p) { const size_t d = rec[p].ue.size(); return d * d + d; };
    while (n_placed < n_sub) {
      const int lv = (int)re.size();
      std::vector<int> take, wait;
      size_t items = 0;
      for (int p : ready) if (lv + tail[p] + 1 >= height) { take.push_back(p); items += load(p); } else wait.push_back(p);
      if (take.empty() && !wait.empty()) { take.push_back(wait.back()); items += load(wait.back()); wait.pop_back(); }   // (cannot happen)
      const size_t cap = ((items + gw_nat - 1) / gw_nat) * gw_nat;
      std::stable_sort(wait.begin(), wait.end(), [&](int x, int y) { return load(x) > load(y); });
      std::vector<int> still;
      for (int p : wait) { if (items + load(p) <= cap) { take.push_back(p); items += load(p); } else still.push_back(p); }
      std::sort(take.begin(), take.end());
      Level D;
      for (int p : take) {
        const Rec& R = rec[p];
        D.piv.push_back(p);
        D.u_entries.push_back(R.ue);
        D.b_items.insert(D.b_items.end(), R.b.begin(), R.b.end());
        D.r_items.insert(D.r_items.end(), R.r.begin(), R.r.end());
        D.c_items.insert(D.c_items.end(), R.c.begin(), R.c.end());
        placed[p] = 1;
        ++n_placed;
      }
      ready = still;
      for (int p : take) for (int sc : rec[p].succ) if (--n_pred[sc] == 0) ready.push_back(sc);
      re.push_back(std::move(D));
    }
    levels = std::move(re);
  }
  for (const Level& L : levels) S.max_level_piv = std::max<int>(S.max_level_piv, (int)L.piv.size());
  S.nslot_lu = (int)S.slot_row.size();
  S.n_levels = (int)levels.size();
  // ---- Gauss-Jordan tail (see Symbolic::gj_lv0) ------------------------------------------------------------------------------------
  S.level_of.assign(n_sub, -1);
  for (int lv = 0; lv < S.n_levels; ++lv) for (int p : levels[lv].piv) S.level_of[p] = lv;
  {
    const int gw_nat = n_sub <= 8 ? 16 : n_sub <= 24 ? 32 : n_sub < 64 ? 64 : 128;   // group width the launch planner gives this grid
    std::vector<std::set<int>> pat(n_sub);                     // block pattern of L + U
    for (size_t q = 0; q < S.slot_row.size(); ++q) pat[S.slot_row[q]].insert(S.slot_col[q]);
    auto passes = [&](size_t items) { return (int)((items + gw_nat - 1) / gw_nat); };
    // simulate the tail from level l0 on: fill of the rows above + items per level -> passes of the forward + back sweeps
    auto simulate = [&](int l0, std::vector<std::set<int>>* out_pat, std::vector<size_t>* out_extra) -> std::pair<int, int> {
      std::vector<std::set<int>> R = pat;
      std::vector<size_t> extra(S.n_levels, 0);
      int fill = 0;
      for (int lv = l0; lv < S.n_levels; ++lv)
        for (int pk : levels[lv].piv) {
          std::vector<int> cols;
          for (int c : R[pk]) if (S.level_of[c] > lv) cols.push_back(c);
          for (int i = 0; i < n_sub; ++i)
            if (S.level_of[i] >= l0 && S.level_of[i] < lv && R[i].count(pk)) {
              for (int c : cols) if (R[i].insert(c).second) ++fill;
              extra[lv] += cols.size() + 1;
            }
        }
      int n_pass = 0;
      for (int lv = 0; lv < S.n_levels; ++lv) {
        const size_t fw = levels[lv].c_items.size() / 2 + levels[lv].r_items.size() / 2 + extra[lv];
        size_t bk = 0;
        if (lv < l0) for (const auto& ue : levels[lv].u_entries) bk += ue.size();
        n_pass += passes(fw) + passes(bk);
      }
      if (out_pat) *out_pat = R;
      if (out_extra) *out_extra = extra;
      return {n_pass, fill};
    };
    int best_l0 = S.n_levels, best_pass = simulate(S.n_levels, nullptr, nullptr).first;
    const int budget = gj_budget >= 0 ? gj_budget : std::max(4, S.nslot_lu / 10);
    for (int l0 = S.n_levels - 2; l0 >= 0; --l0) {
      const auto pf = simulate(l0, nullptr, nullptr);
      if (pf.second > budget) break;
      if (pf.first < best_pass) { best_pass = pf.first; best_l0 = l0; }
    }
    S.gj_lv0 = best_l0;
    S.gj_off.assign(S.n_levels + 1, 0);
    S.gj_roff.assign(S.n_levels + 1, 0);
    for (int lv = 0; lv < S.n_levels; ++lv) {
      if (lv >= best_l0)
        for (int pk : levels[lv].piv) {
          std::vector<int> cols;
          for (const auto& pr : row_slots[pk]) if (S.level_of[pr.first] > lv) cols.push_back(pr.first);
          std::sort(cols.begin(), cols.end());
          for (int i = 0; i < n_sub; ++i) {
            if (!(S.level_of[i] >= best_l0 && S.level_of[i] < lv)) continue;
            int ls = -1;
            for (const auto& pr : row_slots[i]) if (pr.first == pk) ls = pr.second;
            if (ls < 0) continue;
            for (int c : cols) {
              const int dst = add_slot(i, c), us = add_slot(pk, c);
              S.gj_c.push_back(dst | (ls << 16));
              S.gj_c.push_back(us | (pk << 16));
            }
            S.gj_r.push_back(ls | (i << 16));
            S.gj_r.push_back(pk);
          }
        }
      S.gj_off[lv + 1] = (int)S.gj_c.size();
      S.gj_roff[lv + 1] = (int)S.gj_r.size();
    }
  }
  S.nslot = (int)S.slot_row.size();
  S.rslot0 = std::max(S.nslot, (3 * n_sub + 1) / 2);
  // ---- flatten -------------------------------------------------------------------------------------------------------
  std::vector<int>& P = S.prog;
  P.assign((size_t)8 * S.n_levels, 0);
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const Level& L = levels[lv];
    int* h = nullptr;
    const int piv_off = (int)P.size();
    P.insert(P.end(), L.piv.begin(), L.piv.end());
    const int b_off = (int)P.size();
    P.insert(P.end(), L.b_items.begin(), L.b_items.end());
    const int c_off = (int)P.size();
    P.insert(P.end(), L.c_items.begin(), L.c_items.end());
    const int r_off = (int)P.size();
    P.insert(P.end(), L.r_items.begin(), L.r_items.end());
    h = P.data() + (size_t)8 * lv;
    h[0] = piv_off; h[1] = (int)L.piv.size(); h[2] = b_off; h[3] = (int)L.b_items.size();
    h[4] = c_off; h[5] = (int)L.c_items.size() / 2; h[6] = r_off; h[7] = (int)L.r_items.size() / 2;
  }
  S.scale_off = (int)P.size();
  for (const Level& L : levels) P.insert(P.end(), L.b_items.begin(), L.b_items.end());
  S.n_scale = (int)P.size() - S.scale_off;
  S.back_off = (int)P.size();
  P.resize(P.size() + (size_t)2 * S.n_levels, 0);
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const Level& L = levels[lv];
    const int ent_off = (int)P.size();
    int n_ent = 0;
    for (size_t q = 0; q < L.piv.size(); ++q)
      for (int ue : L.u_entries[q]) {
        const int us = ue >> 16, col = ue & 0xffff;
        P.push_back(us | (col << 16));
        P.push_back(L.piv[q]);
        ++n_ent;
      }
    P[S.back_off + 2 * lv + 0] = ent_off;
    P[S.back_off + 2 * lv + 1] = n_ent;
    if (n_ent > 0) S.back_first = lv;
  }
  return S;
}

// Undirected off-diagonal pairs of the original pattern (single-busbar Newton loop: ONE lane computes the Jacobian blocks (u, v)
// and (v, u) of a pair of connected substations): [n_up][2] ints = { u | v << 16, slot(u, v) | slot(v, u) << 16 }, u < v.
inline std::vector<int> build_upairs(const Symbolic& S) {
  std::vector<int> up;
  for (int s = S.n; s < S.nslot_y; ++s) {
    const int r = S.slot_row[s], c = S.slot_col[s];
    if (r >= c) continue;
    int back = -1;
    for (int t = S.n; t < S.nslot_y; ++t) if (S.slot_row[t] == c && S.slot_col[t] == r) { back = t; break; }
    if (back < 0) continue;                               // (cannot happen: the pattern is symmetric)
    up.push_back(r | (c << 16));
    up.push_back(s | (back << 16));
  }
  return up;
}

// false: the grid has too many blocks for 16-bit byte-offset fields (it would not fit the LDS either)
inline bool flat_fits(const Symbolic& S) { return (size_t)(S.rslot0 + S.n) * 16 <= 65536; }

// ---- LDS bank model of one pass (MI355X: 64 banks x 4 B) -------------------------------------------------------------------------
// ds_read_b128 is served in 4 groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 --, one LDS cycle per
// group when the 16-byte chunks fall on distinct bank quads ((addr / 16) mod 16), identical addresses broadcast; every further
// distinct address on a busy quad costs a cycle.  The 8-byte accesses (ds_add_f64, ds_read_b64 of the accumulated right-hand side) go
// in groups of 16 contiguous lanes over bank pairs ((addr / 8) mod 16); atomics on one address serialise.  The two row halves of the
// block array are a constant apart, so row 1 repeats the pattern of row 0.  `flat_pass_cost` = modelled LDS-array cycles of one pass
// for a given item -> lane assignment; `flat_assign_lanes` lowers it by swapping items between lanes (deterministic local search:
// the assignment only decides which lane executes an item and in which order a wavefront's atomics land -- still a fixed order).
struct FlatAcc { unsigned rd[3]; int n_rd; unsigned at[2]; int n_at; unsigned r8; bool has_r8; bool valid; };
inline int flat_b128_group(int lane) { const int l = lane & 31; return ((l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1) + 2 * ((lane >> 5) & 1); }
inline int flat_pass_cost(const std::vector<FlatAcc>& it, int gw) {
  int cost = 0;
  unsigned seen[16][8];
  int cnt[16];
  for (int base = 0; base < gw; base += 64) {
    for (int f = 0; f < 3; ++f)
      for (int g = 0; g < 4; ++g) {
        for (int q = 0; q < 16; ++q) cnt[q] = 0;
        int mx = 0;
        for (int l = 0; l < 64 && base + l < gw; ++l) {
          const FlatAcc& a = it[base + l];
          if (!a.valid || f >= a.n_rd || flat_b128_group(l) != g) continue;
          const unsigned ad = a.rd[f];
          const int q = (ad / 16) % 16;
          bool dup = false;
          for (int k = 0; k < cnt[q] && k < 8; ++k) dup |= seen[q][k] == ad;
          if (!dup) { if (cnt[q] < 8) seen[q][cnt[q]] = ad; ++cnt[q]; mx = std::max(mx, cnt[q]); }
        }
        cost += 2 * mx;                                    // rows 0 and 1
      }
    for (int g = 0; g < 4; ++g) {                          // 8-byte accesses: 16 contiguous lanes
      for (int f = 0; f < 3; ++f) {
        for (int q = 0; q < 16; ++q) cnt[q] = 0;
        int mx = 0;
        for (int l = 16 * g; l < 16 * g + 16 && base + l < gw; ++l) {
          const FlatAcc& a = it[base + l];
          if (!a.valid) continue;
          unsigned ad;
          if (f < 2) { if (f >= a.n_at) continue; ad = a.at[f]; }
          else { if (!a.has_r8) continue; ad = a.r8; }
          const int q = (ad / 8) % 16;
          if (f < 2) { ++cnt[q]; mx = std::max(mx, cnt[q]); }       // atomics: every access counts
          else {
            bool dup = false;
            for (int k = 0; k < cnt[q] && k < 8; ++k) dup |= seen[q][k] == ad;
            if (!dup) { if (cnt[q] < 8) seen[q][cnt[q]] = ad; ++cnt[q]; mx = std::max(mx, cnt[q]); }
          }
        }
        cost += 2 * mx;
      }
    }
  }
  return cost;
}
// items of ONE pass (it.size() == gw, invalid = padding); perm_unit: lanes may only be exchanged inside blocks of this many lanes
// (64 when several wavefronts share the pass: an item must stay in its wavefront, see the wave-closed packing below)
inline void flat_assign_lanes(std::vector<FlatAcc>& it, std::vector<std::pair<unsigned, unsigned>>& words, int gw, int perm_unit, int iters) {
  int cost = flat_pass_cost(it, gw);
  unsigned long long rng = 0x9E3779B97F4A7C15ull;                   // fixed seed: the program is a pure function of the grid
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 32); };
  for (int k = 0; k < iters && cost > 0; ++k) {
    const int x = (int)(next() % (unsigned)gw);
    const int y = (x / perm_unit) * perm_unit + (int)(next() % (unsigned)perm_unit);
    if (x == y || (!it[x].valid && !it[y].valid)) continue;
    std::swap(it[x], it[y]);
    const int c1 = flat_pass_cost(it, gw);
    if (c1 <= cost) { cost = c1; std::swap(words[x], words[y]); }
    else std::swap(it[x], it[y]);
  }
}

// lane_opt: iterations of the bank-conflict local search per pass (0: keep the sequential assignment -- topology classes are built
// at run time, inside a step, and skip it)
inline FlatProg build_flat(const Symbolic& S, int gw, int lane_opt) {
  FlatProg F;
  F.gw = gw;
  F.rhs_field0 = S.rslot0 * 16;
  const unsigned INV = 0xffffffffu;
  auto fld = [](int slot) -> unsigned { return (unsigned)slot * 16u; };
  auto rfld = [&](int row) -> unsigned { return (unsigned)(S.rslot0 + row) * 16u; };
  std::vector<int>& W = F.words;
  // Several wavefronts per instance (gw > 64): the items of a pass that accumulate into the SAME destination (ds_add_f64) must all
  // run in ONE wavefront -- the LDS applies the atomics of a wavefront in a fixed order, those of two wavefronts in a timing-dependent
  // one, and floating-point addition is not associative.  The items of a level are therefore grouped by destination and a group
  // never straddles the boundary between the two 64-lane halves of a pass (it may continue in the NEXT pass: a barrier separates
  // them); invalid items pad the gap.  Returns the number of passes the items took.  gw <= 64: plain sequential packing.
  const unsigned rhs0 = (unsigned)F.rhs_field0;
  auto emit_items = [&](std::vector<std::pair<unsigned, std::pair<unsigned, unsigned>>>& items, bool back) -> int {   // (destination key, (w0, w1))
    if (items.empty()) return 0;
    std::vector<std::pair<unsigned, unsigned>> seq;            // the level's items in lane order, padding included
    auto put = [&](unsigned a, unsigned b) { seq.push_back({a, b}); };
    if (gw > 64) {
      std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      for (size_t i = 0; i < items.size();) {
        size_t j = i;
        while (j < items.size() && items[j].first == items[i].first) ++j;
        const size_t g = j - i, in_half = seq.size() % 64;
        const bool first_half = (seq.size() / 64) % 2 == 0 || gw != 128;        // (gw = 128: halves alternate wavefront 0 / 1)
        if (g > 64) F.wave_closed = false;
        else if (in_half + g > 64 && first_half) while (seq.size() % 64) put(INV, INV);
        for (; i < j; ++i) put(items[i].second.first, items[i].second.second);
      }
    } else {
      // a level that needs several passes is split EVENLY between them (the bank model says why: the cost of a pass grows faster than
      // its item count, and a pass costs its latency whatever it holds)
      const size_t n_pass = (items.size() + gw - 1) / gw, per = (items.size() + n_pass - 1) / n_pass;
      size_t i = 0;
      for (size_t pz = 0; pz < n_pass; ++pz) {
        for (size_t k = 0; k < per && i < items.size(); ++k, ++i) put(items[i].second.first, items[i].second.second);
        while (seq.size() % gw) put(INV, INV);
      }
    }
    while (seq.size() % gw) put(INV, INV);
    const int n_pass = (int)(seq.size() / gw);
    if (lane_opt > 0)
      for (int pz = 0; pz < n_pass; ++pz) {
        std::vector<std::pair<unsigned, unsigned>> words(seq.begin() + (size_t)pz * gw, seq.begin() + (size_t)(pz + 1) * gw);
        std::vector<FlatAcc> acc(gw);
        for (int t = 0; t < gw; ++t) {
          FlatAcc& a = acc[t];
          const unsigned w0 = words[t].first, w1 = words[t].second;
          a = FlatAcc{};
          a.valid = w0 != INV;
          if (!a.valid) continue;
          if (!back) {                                         // reads D_p, A_l, A_u (2 rows each); 4 atomics on the rows of dst
            a.rd[0] = w1 >> 16; a.rd[1] = w0 >> 16; a.rd[2] = w1 & 0xffffu; a.n_rd = 3;
            a.at[0] = w0 & 0xffffu; a.at[1] = (w0 & 0xffffu) + 8; a.n_at = 2;
          } else {                                             // reads A_u, D_col (2 rows), s_col (8 bytes per row); 2 atomics on s_dst
            a.rd[0] = w0 & 0xffffu; a.rd[1] = (w0 >> 16) - rhs0; a.n_rd = 2;
            a.r8 = w0 >> 16; a.has_r8 = true;
            a.at[0] = w1; a.n_at = 1;
          }
        }
        flat_assign_lanes(acc, words, gw, gw > 64 ? 64 : gw, lane_opt);
        std::copy(words.begin(), words.end(), seq.begin() + (size_t)pz * gw);
      }
    if (gw > 64)
      for (int pz = 0; pz < n_pass; ++pz) {
        bool solo = true;
        for (int t = 64; t < gw; ++t) solo &= seq[(size_t)pz * gw + t].first == INV;
        const int k = (back ? F.n_back : F.n_fwd) + pz;
        if (solo && k < 32) (back ? F.solo_back : F.solo_fwd) |= 1u << k;
      }
    for (auto& w : seq) { W.push_back((int)w.first); W.push_back((int)w.second); }
    return n_pass;
  };
  // forward
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const int* h = S.prog.data() + (size_t)8 * lv;
    const int c_off = h[4], n_c = h[5], r_off = h[6], n_r = h[7];
    const int g0 = S.gj_off.empty() ? 0 : S.gj_off[lv], g1 = S.gj_off.empty() ? 0 : S.gj_off[lv + 1];
    const int q0 = S.gj_roff.empty() ? 0 : S.gj_roff[lv], q1 = S.gj_roff.empty() ? 0 : S.gj_roff[lv + 1];
    if (n_c + n_r + (g1 - g0) + (q1 - q0) == 0) continue;
    std::vector<std::pair<unsigned, std::pair<unsigned, unsigned>>> items;
    for (int o = g0; o < g1; o += 2) {                          // Gauss-Jordan tail: the pivots' columns in the tail rows above them
      const unsigned w0 = (unsigned)S.gj_c[o], w1 = (unsigned)S.gj_c[o + 1];
      items.push_back({fld(w0 & 0xffffu), {fld(w0 & 0xffffu) | (fld(w0 >> 16) << 16), fld(w1 & 0xffffu) | (fld(w1 >> 16) << 16)}});
    }
    for (int o = q0; o < q1; o += 2) {
      const unsigned w0 = (unsigned)S.gj_r[o];
      const int p = S.gj_r[o + 1];
      items.push_back({rfld((int)(w0 >> 16)), {rfld((int)(w0 >> 16)) | (fld(w0 & 0xffffu) << 16), rfld(p) | (fld(p) << 16)}});
    }
    for (int o = 0; o < n_c; ++o) {
      const unsigned w0 = (unsigned)S.prog[c_off + 2 * o], w1 = (unsigned)S.prog[c_off + 2 * o + 1];
      items.push_back({fld(w0 & 0xffffu), {fld(w0 & 0xffffu) | (fld(w0 >> 16) << 16), fld(w1 & 0xffffu) | (fld(w1 >> 16) << 16)}});
    }
    for (int o = 0; o < n_r; ++o) {
      const unsigned w0 = (unsigned)S.prog[r_off + 2 * o];
      const int p = S.prog[r_off + 2 * o + 1];
      items.push_back({rfld((int)(w0 >> 16)), {rfld((int)(w0 >> 16)) | (fld(w0 & 0xffffu) << 16), rfld(p) | (fld(p) << 16)}});
    }
    F.n_fwd += emit_items(items, false);
  }
  for (int k = 0; k < gw; ++k) { W.push_back((int)INV); W.push_back((int)INV); }
  // (no scaling pass: the U blocks and the right-hand side stay UNSCALED -- the back substitution is
  //  x_p = inv(D_p) (s_p - sum_j A_pj x_j) and every consumer of x_j applies inv(D_j) to the accumulated s_j itself)
  F.scale_off = (int)W.size();
  F.n_scale = 0;
  F.n_scale_rhs = 0;
  if (W.size() & 1) W.push_back((int)INV);                     // 8-byte alignment of the back section
  // back substitution, levels in reverse
  F.back_off = (int)W.size();
  for (int lv = S.back_first; lv >= 0; --lv) {
    const int ent_off = S.prog[S.back_off + 2 * lv], n_ent = S.prog[S.back_off + 2 * lv + 1];
    if (n_ent == 0 || lv >= S.gj_lv0) continue;                // (tail rows: reduced to D_p x_p = s_p by the forward passes)
    std::vector<std::pair<unsigned, std::pair<unsigned, unsigned>>> items;
    for (int o = 0; o < n_ent; ++o) {
      const unsigned w = (unsigned)S.prog[ent_off + 2 * o];
      const int p = S.prog[ent_off + 2 * o + 1];
      items.push_back({rfld(p), {fld(w & 0xffffu) | (rfld((int)(w >> 16)) << 16), rfld(p)}});
    }
    F.n_back += emit_items(items, true);
  }
  // (two all-invalid passes behind the last one where the sweeps fetch the words TWO passes ahead: several wavefronts per instance, and the
  //  instance-group kernels, which stream the program from global memory -- those programs are never staged in LDS, the padding costs nothing there)
  for (int k = 0; k < ((gw > 64 || gw <= 32) ? 2 : 1) * gw; ++k) { W.push_back((int)INV); W.push_back((int)INV); }
  while (W.size() & 3) W.push_back((int)INV);
  return F;
}

inline int flat_pass_count(const Symbolic& S, int gw) { const FlatProg F = build_flat(S, gw, 0); return F.n_fwd + F.n_back; }

// ---- slot layout search ---------------------------------------------------------------------------------------------------------
// Which LDS banks an item of a pass touches is decided by the SLOT numbers of its blocks (16 bytes per slot and row half: slot mod
// 16 is the bank class of a ds_read_b128).  The numbering of build_symbolic is arbitrary inside three ranges -- off-diagonal blocks of
// the original pattern [n, nslot_y), LU fill [nslot_y, nslot_lu), Gauss-Jordan fill [nslot_lu, nslot) (the diagonal slots are the
// substation ids) -- so a local search renumbers the blocks inside each range to lower the bank-model cost of the passes of the
// grid's usual group width (sequential lane assignment; the per-pass lane search of build_flat then starts from a better layout).
// relabel_slots: new number of slot s is perm[s].
inline Symbolic relabel_slots(const Symbolic& S0, const std::vector<int>& perm) {
  Symbolic S = S0;
  for (int q = 0; q < S0.nslot; ++q) { S.slot_row[perm[q]] = S0.slot_row[q]; S.slot_col[perm[q]] = S0.slot_col[q]; }
  for (auto& v : S.br_slot) v = perm[v];
  auto lo16 = [&](int w) { return (int)(((unsigned)w & 0xffff0000u) | (unsigned)perm[(unsigned)w & 0xffffu]); };
  auto hi16 = [&](int w) { return (int)(((unsigned)w & 0xffffu) | ((unsigned)perm[(unsigned)w >> 16] << 16)); };
  for (int lv = 0; lv < S.n_levels; ++lv) {
    int* h = S.prog.data() + (size_t)8 * lv;
    for (int k = 0; k < h[3]; ++k) S.prog[h[2] + k] = lo16(S.prog[h[2] + k]);                                    // (p << 16) | u
    for (int k = 0; k < h[5]; ++k) { int& w0 = S.prog[h[4] + 2 * k]; int& w1 = S.prog[h[4] + 2 * k + 1]; w0 = hi16(lo16(w0)); w1 = lo16(w1); }
    for (int k = 0; k < h[7]; ++k) { int& w0 = S.prog[h[6] + 2 * k]; w0 = lo16(w0); }                             // l | (row << 16), p
  }
  for (int k = 0; k < S.n_scale; ++k) S.prog[S.scale_off + k] = lo16(S.prog[S.scale_off + k]);
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const int off = S.prog[S.back_off + 2 * lv], n_ent = S.prog[S.back_off + 2 * lv + 1];
    for (int k = 0; k < n_ent; ++k) S.prog[off + 2 * k] = lo16(S.prog[off + 2 * k]);                              // u | (col << 16), p
  }
  for (size_t k = 0; k + 1 < S.gj_c.size(); k += 2) { S.gj_c[k] = hi16(lo16(S.gj_c[k])); S.gj_c[k + 1] = lo16(S.gj_c[k + 1]); }
  for (size_t k = 0; k + 1 < S.gj_r.size(); k += 2) S.gj_r[k] = lo16(S.gj_r[k]);
  return S;
}
// bank-model cost of every pass of the flat program of group width gw (sequential lanes) + the pair phase's block writes
inline long flat_layout_cost(const Symbolic& S, int gw) {
  const FlatProg F = build_flat(S, gw, 0);
  const unsigned INV = 0xffffffffu, rhs0 = (unsigned)F.rhs_field0;
  long tot = 0;
  std::vector<FlatAcc> acc(gw);
  auto pass = [&](int off, int k, bool back) {
    for (int t = 0; t < gw; ++t) {
      FlatAcc& a = acc[t];
      a = FlatAcc{};
      const unsigned w0 = (unsigned)F.words[off + 2 * (k * gw + t)], w1 = (unsigned)F.words[off + 2 * (k * gw + t) + 1];
      a.valid = w0 != INV;
      if (!a.valid) continue;
      if (!back) { a.rd[0] = w1 >> 16; a.rd[1] = w0 >> 16; a.rd[2] = w1 & 0xffffu; a.n_rd = 3; a.at[0] = w0 & 0xffffu; a.at[1] = (w0 & 0xffffu) + 8; a.n_at = 2; }
      else { a.rd[0] = w0 & 0xffffu; a.rd[1] = (w0 >> 16) - rhs0; a.n_rd = 2; a.r8 = w0 >> 16; a.has_r8 = true; a.at[0] = w1; a.n_at = 1; }
    }
    tot += flat_pass_cost(acc, gw);
  };
  for (int k = 0; k < F.n_fwd; ++k) pass(0, k, false);
  for (int k = 0; k < F.n_back; ++k) pass(F.back_off, k, true);
  const std::vector<int> up = build_upairs(S);                 // pair phase: lane k % gw writes both row halves of blocks (u, v) and (v, u)
  const int n_up = (int)up.size() / 2;
  for (int k0 = 0; k0 < n_up; k0 += gw) {
    for (int t = 0; t < gw; ++t) {
      FlatAcc& a = acc[t];
      a = FlatAcc{};
      a.valid = k0 + t < n_up;
      if (!a.valid) continue;
      const unsigned w1 = (unsigned)up[2 * (k0 + t) + 1];
      a.rd[0] = (w1 & 0xffffu) * 16u; a.rd[1] = (w1 >> 16) * 16u; a.n_rd = 2;
    }
    tot += flat_pass_cost(acc, gw);
  }
  return tot;
}
inline Symbolic optimize_slot_layout(const Symbolic& S0, int iters) {
  if (iters <= 0 || S0.nslot - S0.n < 2 || !flat_fits(S0)) return S0;
  const int gw = S0.n <= 8 ? 16 : S0.n <= 24 ? 32 : S0.n < 64 ? 64 : 128;
  Symbolic cur = S0;
  long cost = flat_layout_cost(cur, gw);
  unsigned long long rng = 0xD1B54A32D192ED03ull;               // fixed seed: the layout is a pure function of the grid
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 32); };
  std::vector<int> perm(S0.nslot);
  for (int k = 0; k < iters; ++k) {
    const int r = (int)(next() % 3u);
    const int a0 = r == 0 ? S0.n : r == 1 ? S0.nslot_y : S0.nslot_lu, a1 = r == 0 ? S0.nslot_y : r == 1 ? S0.nslot_lu : S0.nslot;
    if (a1 - a0 < 2) continue;
    const int x = a0 + (int)(next() % (unsigned)(a1 - a0)), y = a0 + (int)(next() % (unsigned)(a1 - a0));
    if (x == y || (x % 16) == (y % 16)) continue;               // same bank class: nothing changes
    for (int q = 0; q < S0.nslot; ++q) perm[q] = q;
    perm[x] = y; perm[y] = x;
    Symbolic nx = relabel_slots(cur, perm);
    const long c = flat_layout_cost(nx, gw);
    if (c <= cost) { cost = c; cur = std::move(nx); }
  }
  return cur;
}

}  // namespace gpf

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
};

inline Symbolic build_symbolic(int n_sub, int n_line, const int* line_or_sub, const int* line_ex_sub, int degree_slack = 1) {
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
      }
      g[p].clear();
      done[p] = 1;
      --remaining;
    }
    S.max_level_piv = std::max<int>(S.max_level_piv, (int)L.piv.size());
    levels.push_back(std::move(L));
  }
  S.nslot = (int)S.slot_row.size();
  S.n_levels = (int)levels.size();
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

}  // namespace gpf

// gridpf_symbolic.hpp -- host-side symbolic analysis for the block-sparse kernels (gridpf_sparse.hpp).
//
// The Newton Jacobian and the DC matrix of a grid have the sparsity of its SUBSTATION graph when they are stored
// as dense blocks per (substation, substation) pair (block = all busbars x {theta, |V|} of the two substations):
// bus splits, line outages, PV/PQ/reference changes only change VALUES inside blocks (rows / columns of inactive
// or fixed variables become identity).  The symbolic factorisation -- elimination order (minimum degree), fill
// pattern, and the list of block operations of a right-looking block LU without inter-block pivoting -- is
// therefore computed ONCE per grid at gpf_create and shared by every lane and every topology.
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
  std::vector<int> perm;           // perm[k] = substation eliminated at step k
  std::vector<int> slot_row, slot_col;   // [nslot]
  std::vector<int> diag_slot;      // [n] slot of (s, s)  (== s by construction)
  // per elimination step k (pivot p = perm[k]):
  std::vector<int> l_begin;        // [n+1] range into l_slot / l_row: blocks (i, p), i not yet eliminated
  std::vector<int> l_slot, l_row;
  std::vector<int> u_begin;        // [n+1] range into u_slot / u_col: blocks (p, j), j not yet eliminated
  std::vector<int> u_slot, u_col;
  std::vector<int> op_begin;       // [n+1] range into op_dst / op_l / op_u:  A[dst] -= A[l] * A[u]
  std::vector<int> op_dst, op_l, op_u;
  // branch -> slots of its four blocks (ff, ft, tf, tt) for the atomics-based assembly
  std::vector<int> br_slot;        // [n_line][4]
  int max_l = 0, max_ops = 0;
};

inline Symbolic build_symbolic(int n_sub, int n_line, const int* line_or_sub, const int* line_ex_sub) {
  Symbolic S;
  S.n = n_sub;
  std::vector<std::set<int>> adj(n_sub);
  for (int l = 0; l < n_line; ++l) {
    const int a = line_or_sub[l], b = line_ex_sub[l];
    if (a != b) { adj[a].insert(b); adj[b].insert(a); }
  }
  // slots of the original pattern: diagonal first (slot s = (s, s)), then the off-diagonal pairs
  auto key = [n_sub](int r, int c) { return (int64_t)r * n_sub + c; };
  std::vector<std::pair<int64_t, int>> slot_of;   // sorted lookup built at the end; use a map while building
  std::vector<std::vector<std::pair<int, int>>> row_slots(n_sub);   // row -> (col, slot)
  auto add_slot = [&](int r, int c) -> int {
    for (auto& pr : row_slots[r]) if (pr.first == c) return pr.second;
    const int s = (int)S.slot_row.size();
    S.slot_row.push_back(r);
    S.slot_col.push_back(c);
    row_slots[r].push_back({c, s});
    return s;
  };
  S.diag_slot.resize(n_sub);
  for (int s = 0; s < n_sub; ++s) S.diag_slot[s] = add_slot(s, s);
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
  // minimum-degree ordering with symbolic elimination on the (symmetric) substation graph
  std::vector<std::set<int>> g = adj;
  std::vector<char> done(n_sub, 0);
  S.perm.reserve(n_sub);
  S.l_begin.push_back(0);
  S.u_begin.push_back(0);
  S.op_begin.push_back(0);
  for (int k = 0; k < n_sub; ++k) {
    int best = -1;
    size_t bd = (size_t)-1;
    for (int s = 0; s < n_sub; ++s)
      if (!done[s] && g[s].size() < bd) { bd = g[s].size(); best = s; }
    const int p = best;
    done[p] = 1;
    S.perm.push_back(p);
    std::vector<int> nb(g[p].begin(), g[p].end());     // remaining neighbours (sorted)
    for (int i : nb) {
      S.l_slot.push_back(add_slot(i, p));
      S.l_row.push_back(i);
      S.u_slot.push_back(add_slot(p, i));
      S.u_col.push_back(i);
    }
    for (int i : nb)
      for (int j : nb) {
        S.op_dst.push_back(add_slot(i, j));             // creates fill when (i, j) is new
        S.op_l.push_back(add_slot(i, p));
        S.op_u.push_back(add_slot(p, j));
      }
    // graph update: clique among the neighbours, remove p
    for (int i : nb) {
      g[i].erase(p);
      for (int j : nb) if (i != j) g[i].insert(j);
    }
    g[p].clear();
    S.l_begin.push_back((int)S.l_slot.size());
    S.u_begin.push_back((int)S.u_slot.size());
    S.op_begin.push_back((int)S.op_dst.size());
    S.max_l = std::max<int>(S.max_l, (int)nb.size());
    S.max_ops = std::max<int>(S.max_ops, (int)(nb.size() * nb.size()));
  }
  S.nslot = (int)S.slot_row.size();
  return S;
}

}  // namespace gpf

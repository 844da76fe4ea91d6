// Developer tool (host only): how much the pass count of the flat LU program depends on the elimination order -- the greedy
// level-scheduled minimum-degree ordering of build_symbolic run on randomly relabelled copies of the substation graph (the labels
// only break ties) and with different degree slacks.  stdin: "n_sub n_line" then "or ex" per line.
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#include "../grid2op_amd/csrc/gridpf_symbolic.hpp"
int main(int argc, char** argv) {
  int n_sub, n_line; if (scanf("%d %d", &n_sub, &n_line) != 2) return 1;
  std::vector<int> a(n_line), b(n_line);
  for (int i = 0; i < n_line; ++i) if (scanf("%d %d", &a[i], &b[i]) != 2) return 1;
  const int trials = argc > 1 ? atoi(argv[1]) : 200;
  const int gw = n_sub < 64 ? 64 : 128;
  std::mt19937 rng(12345);
  int best_cost = 1 << 30;
  for (int t = 0; t < trials; ++t) {
    std::vector<int> perm(n_sub);
    std::iota(perm.begin(), perm.end(), 0);
    if (t) std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> pa(n_line), pb(n_line);
    for (int l = 0; l < n_line; ++l) { pa[l] = perm[a[l]]; pb[l] = perm[b[l]]; }
    for (int slack = 0; slack <= 3; ++slack) {
      gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, pa.data(), pb.data(), slack);
      gpf::FlatProg F = gpf::build_flat(S, gw);
      int solo = 0;
      for (int k = 0; k < F.n_fwd; ++k) solo += (F.solo_fwd >> k) & 1;
      for (int k = 0; k < F.n_back; ++k) solo += (F.solo_back >> k) & 1;
      const int duo = F.n_fwd + F.n_back - solo;
      const int cost = gw > 64 ? duo * 750 + solo * 480 : (F.n_fwd + F.n_back) * 550;
      if (cost < best_cost || t == 0) {
        if (cost < best_cost) best_cost = cost;
        printf("trial %4d slack %d: levels %2d, fwd %2d + back %2d passes (%d solo), nslot %d (lu %d), est. cycles %d\n", t, slack, S.n_levels, F.n_fwd, F.n_back, solo, S.nslot,
               S.nslot_lu, cost);
      }
    }
  }
}

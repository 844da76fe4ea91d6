// Developer tool (host only): level / pass statistics of the symbolic program of a grid (line_or_sub / line_ex_sub on stdin).
#include <cstdio>
#include <vector>
#include "../grid2op_amd/csrc/gridpf_symbolic.hpp"
int main(int argc, char** argv) {
  int n_sub, n_line; if (scanf("%d %d", &n_sub, &n_line) != 2) return 1;
  std::vector<int> a(n_line), b(n_line);
  for (int i = 0; i < n_line; ++i) if (scanf("%d %d", &a[i], &b[i]) != 2) return 1;
  int slack = argc > 1 ? atoi(argv[1]) : 1;
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, a.data(), b.data(), slack);
  printf("n=%d nslot_y=%d nslot=%d levels=%d rslot0=%d upairs=%zu\n", S.n, S.nslot_y, S.nslot, S.n_levels, S.rslot0, gpf::build_upairs(S).size() / 2);
  for (int lv = 0; lv < S.n_levels; ++lv) { const int* h = S.prog.data() + 8 * lv; printf("  level %2d: piv=%3d U=%3d upd=%4d rhs=%3d back=%3d\n", lv, h[1], h[3], h[5], h[7], S.prog[S.back_off + 2 * lv + 1]); }
  for (int gw : {16, 32, 64, 128}) { gpf::FlatProg F = gpf::build_flat(S, gw); printf("  GW=%3d: fwd=%d scale=%d(rhs %d) back=%d passes, %zu words, solo fwd=0x%x back=0x%x\n", gw, F.n_fwd, F.n_scale, F.n_scale_rhs, F.n_back, F.words.size(), F.solo_fwd, F.solo_back); }
}

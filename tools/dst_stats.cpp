// Developer tool (host only): how many items of each pass of the flat program are the ONLY writer of their destination in that pass.
#include <cstdio>
#include <map>
#include <vector>
#include "../grid2op_amd/csrc/gridpf_symbolic.hpp"
int main(int argc, char** argv) {
  int n_sub, n_line; if (scanf("%d %d", &n_sub, &n_line) != 2) return 1;
  std::vector<int> a(n_line), b(n_line);
  for (int i = 0; i < n_line; ++i) if (scanf("%d %d", &a[i], &b[i]) != 2) return 1;
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, a.data(), b.data());
  const int gw = argc > 1 ? atoi(argv[1]) : (n_sub < 64 ? 64 : 128);
  gpf::FlatProg F = gpf::build_flat(S, gw);
  long tot = 0, uniq = 0;
  auto pass = [&](int off, int k, bool back) {
    std::map<unsigned, int> cnt;
    int n = 0;
    for (int t = 0; t < gw; ++t) {
      const unsigned w0 = (unsigned)F.words[off + 2 * (k * gw + t)], w1 = (unsigned)F.words[off + 2 * (k * gw + t) + 1];
      if (w0 == 0xffffffffu) continue;
      ++n; ++cnt[back ? w1 : (w0 & 0xffffu)];
    }
    int u = 0; for (auto& kv : cnt) if (kv.second == 1) ++u;
    int mx = 0; for (auto& kv : cnt) mx = std::max(mx, kv.second);
    printf("  %s pass %2d: %3d items, %3d destinations, %3d items alone on theirs, worst %d\n", back ? "back" : "fwd ", k, n, (int)cnt.size(), u, mx);
    tot += n; uniq += u;
  };
  for (int k = 0; k < F.n_fwd; ++k) pass(0, k, false);
  for (int k = 0; k < F.n_back; ++k) pass(F.back_off, k, true);
  printf("GW=%d: %ld items, %ld (%.0f %%) alone on their destination in their pass\n", gw, tot, uniq, 100.0 * uniq / tot);
}

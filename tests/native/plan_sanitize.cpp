// Host-only harness for the plan compiler (fruits_amd/csrc/plan.cpp), built with
// -fsanitize=address,undefined by tests/test_host.py::test_plan_compiler_sanitized.
// Reads word lists from stdin:  W  then per word: L Dw depth  followed by L*Dw exponents
// and L alphas; then weighting, flags, and a list of group counts to lay out.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../fruits_amd/csrc/plan.h"

int main() {
  int cases = 0;
  if (scanf("%d", &cases) != 1) return 2;
  long checksum = 0;
  for (int c = 0; c < cases; ++c) {
    int W, weighting, flags, kind;
    if (scanf("%d %d %d %d", &kind, &W, &weighting, &flags) != 4) return 2;
    std::vector<int32_t> exps, L(W), Dw(W), depth(W);
    std::vector<float> alpha;
    for (int i = 0; i < W; ++i) {
      if (scanf("%d %d %d", &L[i], &Dw[i], &depth[i]) != 3) return 2;
      for (int j = 0; j < L[i] * Dw[i]; ++j) {
        int e;
        if (scanf("%d", &e) != 1) return 2;
        exps.push_back(e);
      }
      for (int j = 0; j < L[i]; ++j) {
        float a;
        if (scanf("%f", &a) != 1) return 2;
        alpha.push_back(a);
      }
    }
    std::string err;
    fr::Plan *p;
    if (kind == 1) {
      const float freqs[3] = {0.05f, 0.25f, 0.5f};
      p = fr::build_coswiss_plan(W, exps.data(), L.data(), Dw.data(), 3, freqs, 2, flags, err);
    } else {
      p = fr::build_plan(W, exps.data(), L.data(), Dw.data(), weighting ? alpha.data() : nullptr,
                         depth.data(), weighting, flags, err);
    }
    if (!p) {
      printf("case %d: rejected: %s\n", c, err.c_str());
      continue;
    }
    if (!p->cos) {
      for (int G : {1, 2, 3, 4, 7, 64}) {
        fr::GroupedProgram &gp = fr::grouped(*p, G);
        checksum += (long)gp.recs.size() + gp.group_begin.back() + gp.groups;
        for (const fr::NodeRec &r : gp.recs) checksum += r.w[0] + r.w[6];
      }
      // the static scheduler (frames, staged rows, output-row continuation entries)
      for (int G : {1, 2, 3}) {
        const fr::StaticSchedule sc = fr::static_schedule(*p, G);
        checksum += sc.ok ? (long)sc.entries.size() + sc.frames + sc.rows : -1;
        for (const fr::NodeRec &r : sc.entries) checksum += r.w[0] & 0xff;
      }
      // the plan in pieces (several piece sizes, units of one item and of many)
      for (int piece : {2, 7, 16, 64, 128}) {
        fr::PiecedProgram &pp = fr::pieced(*p, piece, piece == 7 ? 1 : 0);
        checksum += pp.ok ? (long)pp.types.size() + pp.chain_nodes : -1;
        for (const fr::PieceType &t : pp.types) {
          checksum += (long)t.recs.size() + t.units() + t.max_unit_nodes + t.body_rows;
          for (int32_t v : t.items) checksum += v & 0xff;
        }
        for (int32_t r : pp.row_of_walk) checksum += r;
      }
    } else {
      checksum += p->cos->factors.size() + p->cos->letter_begin.back();
      delete p->cos;
    }
    checksum += p->K + p->levels + (long)p->nodes.size();
    printf("case %d: K=%d nodes=%zu levels=%d units=%d\n", c, p->K, p->nodes.size(), p->levels,
           p->units());
    delete p;
  }
  printf("checksum %ld\n", checksum);
  return 0;
}

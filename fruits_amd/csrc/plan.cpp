// Plan compiler: word list -> prefix trie -> DFS program (see plan.h).
#include "plan.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <numeric>

namespace fr {
namespace {

struct TrieNode {
  std::vector<int32_t> exps;  // exponents, trailing zeros trimmed
  uint32_t alpha_bits = 0;    // bit pattern of this letter's alpha (0 when unweighted)
  int parent = -1;
  int depth = 0;              // letters from the root (1 = first letter)
  std::vector<int> children;
  std::vector<int32_t> emit;  // output rows
};

struct Key {
  std::vector<int32_t> exps;
  uint32_t alpha_bits;
  bool operator<(const Key &o) const {
    if (alpha_bits != o.alpha_bits) return alpha_bits < o.alpha_bits;
    return exps < o.exps;
  }
};

uint32_t fbits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

struct Builder {
  Plan *p;
  std::vector<TrieNode> trie;                 // trie[0] = root
  std::vector<std::map<Key, int>> child_map;  // only used when sharing
  std::map<uint32_t, int> alpha_index;        // alpha bits -> index in p->alphas
  std::map<int, int> dim_row;                 // X dimension -> LDS row

  int alpha_id(uint32_t bits) {
    auto it = alpha_index.find(bits);
    if (it != alpha_index.end()) return it->second;
    float f;
    std::memcpy(&f, &bits, 4);
    int id = (int)p->alphas.size();
    p->alphas.push_back(f);
    alpha_index[bits] = id;
    return id;
  }
  int row_plus(uint32_t bits) { return p->dims_used + 2 * alpha_id(bits); }
  int row_minus(uint32_t bits) { return p->dims_used + 2 * alpha_id(bits) + 1; }
  int row_lin(uint32_t bits) { return p->dims_used + alpha_id(bits); }  // Arctic: g*alpha

  void emit_node(int t, int level, int chain, int unit) {
    const TrieNode &tn = trie[t];
    NodeDesc nd{};
    nd.level = level;
    nd.flags = (chain ? F_CHAIN : 0) | (tn.children.empty() ? 0 : F_CHILDREN);
    nd.fac_begin = (int32_t)p->factors.size();
    nd.emit_mul = -1;
    nd.z_mul = -1;
    const bool weighted = p->weighting != 0;
    const bool has_parent = tn.depth > 1;
    if (p->semiring == kSemiArctic) {
      // (max, +), fruits/iss/semiring.py:282-338: tmp += el * Z[dim] per dimension;
      // non-total: then tmp -= g*alpha_{k-1}, emit cummax(tmp), children scan tmp + g*alpha_k;
      // total: tmp -= g*alpha_{k-1} (end of the previous letter), letters, tmp += g*alpha_k,
      //        cummax, emit tmp - g*alpha_k
      if (weighted && p->weighting == 2 && has_parent)
        p->factors.push_back(fac_arctic(row_lin(trie[tn.parent].alpha_bits), -1));
      for (size_t d = 0; d < tn.exps.size(); ++d)
        if (tn.exps[d] != 0) p->factors.push_back(fac_arctic(dim_row[(int)d], tn.exps[d]));
      if (weighted && p->weighting == 1) {
        if (has_parent)
          p->factors.push_back(fac_arctic(row_lin(trie[tn.parent].alpha_bits), -1) |
                               (p->letter_sum ? FAC_FOLD : 0));
        if (!tn.children.empty()) nd.z_mul = row_lin(tn.alpha_bits);
      } else if (weighted) {
        p->factors.push_back(fac_arctic(row_lin(tn.alpha_bits), 1));
        nd.emit_mul = row_lin(tn.alpha_bits);
      }
    } else {
    // total weighting: the factor exp(-g*alpha_{k-1}) is applied right after the
    // shift, before the letters (fruits/iss/semiring.py:154-157, then :143-149)
    if (weighted && p->weighting == 2 && has_parent)
      p->factors.push_back(row_minus(trie[tn.parent].alpha_bits));
    for (size_t d = 0; d < tn.exps.size(); ++d) {
      int occ = tn.exps[d];
      for (int r = 0; r < std::abs(occ); ++r)
        p->factors.push_back(dim_row[(int)d] | (occ < 0 ? FAC_DIV : 0));
    }
    if (weighted && p->weighting == 1) {
      // non-total: letters, then exp(-g*alpha_{k-1}) (semiring.py:111-119); the
      // child scan runs over s*exp(+g*alpha_k) (:122-124)
      if (has_parent) p->factors.push_back(row_minus(trie[tn.parent].alpha_bits));
      if (!tn.children.empty()) nd.z_mul = row_plus(tn.alpha_bits);
    } else if (weighted) {
      // total: letters, then exp(+g*alpha_k), scan, emit * exp(-g*alpha_k)
      // (semiring.py:150-153)
      p->factors.push_back(row_plus(tn.alpha_bits));
      nd.emit_mul = row_minus(tn.alpha_bits);
    }
    }
    nd.fac_count = (int32_t)p->factors.size() - nd.fac_begin;
    nd.emit_begin = (int32_t)p->emit_rows.size();
    for (int32_t r : tn.emit) p->emit_rows.push_back(r);
    nd.emit_count = (int32_t)tn.emit.size();
    p->nodes.push_back(nd);
    p->unit_of.push_back(unit);
    p->levels = std::max(p->levels, level + 1);
    if (tn.children.size() == 1) {
      emit_node(tn.children[0], level, 1, unit);
    } else {
      for (int c : tn.children) emit_node(c, level + 1, 0, unit);
    }
  }
};

double node_cost(const NodeDesc &nd) {
  const bool has_children = nd.flags & F_CHILDREN;
  const bool need2 = has_children && nd.z_mul >= 0;
  const bool need1 = nd.emit_count > 0 || (has_children && !need2);
  return (need1 ? 1.0 : 0.0) + (need2 ? 1.0 : 0.0) + 1.0 * nd.emit_count +
         0.1 * nd.fac_count;
}

}  // namespace

Plan *build_coswiss_plan(int W, const int32_t *exps, const int32_t *L, const int32_t *Dw, int F,
                         const float *freqs, int exponent, int total, std::string &err) {
  if (W < 0 || F < 0 || (W > 0 && (!exps || !L || !Dw)) || (F > 0 && !freqs)) {
    err = "fr_plan_create_coswiss: null argument";
    return nullptr;
  }
  if (exponent < 1) {
    err = "fr_plan_create_coswiss: exponent must be >= 1";
    return nullptr;
  }
  Plan *p = new Plan();
  CosProgram *c = new CosProgram();
  p->cos = c;
  p->W = W;
  p->K = W * F;
  c->W = W;
  c->F = F;
  c->exponent = exponent;
  c->total = total != 0;
  c->freqs.assign(freqs, freqs + F);
  c->letter_begin.push_back(0);
  c->fac_begin.push_back(0);
  const int32_t *e = exps;
  for (int i = 0; i < W; ++i) {
    if (L[i] < 1 || Dw[i] < 1) {
      err = "fr_plan_create_coswiss: word " + std::to_string(i) + " has invalid L/Dw";
      delete c;
      delete p;
      return nullptr;
    }
    for (int k = 0; k < L[i]; ++k) {
      for (int d = 0; d < Dw[i]; ++d) {
        const int32_t el = e[k * Dw[i] + d];
        if (el == 0) continue;
        if (d > FAC_ROW_MASK) {
          err = "fr_plan_create_coswiss: dimension beyond " + std::to_string(FAC_ROW_MASK + 1);
          delete c;
          delete p;
          return nullptr;
        }
        for (int r = 0; r < (el < 0 ? -el : el); ++r)
          c->factors.push_back(d | (el < 0 ? FAC_DIV : 0));
        p->max_dim = std::max(p->max_dim, d + 1);
      }
      c->fac_begin.push_back((int32_t)c->factors.size());
    }
    p->levels = std::max(p->levels, (int)L[i]);
    c->letter_begin.push_back((int32_t)c->fac_begin.size() - 1);
    e += (size_t)L[i] * Dw[i];
  }
  return p;
}

Plan *build_plan(int W, const int32_t *exps, const int32_t *L, const int32_t *Dw,
                 const float *alpha, const int32_t *depth, int weighting, int flags,
                 std::string &err) {
  if (W < 0 || (W > 0 && (!exps || !L || !Dw || !depth))) {
    err = "fr_plan_create: null argument";
    return nullptr;
  }
  if (weighting < 0 || weighting > 2) {
    err = "fr_plan_create: weighting must be 0, 1 or 2";
    return nullptr;
  }
  if (weighting != 0 && !alpha) {
    err = "fr_plan_create: weighted plan needs alpha";
    return nullptr;
  }
  for (int i = 0; i < W; ++i) {
    if (L[i] < 1 || Dw[i] < 1 || depth[i] < 0 || depth[i] > L[i]) {
      err = "fr_plan_create: word " + std::to_string(i) + " has invalid L/Dw/depth";
      return nullptr;
    }
  }

  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool share = (flags & 1) && attempt == 0;
    Plan *p = new Plan();
    p->W = W;
    p->weighting = weighting;
    p->semiring = (flags & 2) ? kSemiArctic : ((flags & 4) ? kSemiBayesian : kSemiReals);
    p->letter_sum = (flags & 8) != 0 && (flags & 2) != 0;
    p->shared = share;
    Builder b;
    b.p = p;
    b.trie.emplace_back();
    b.child_map.emplace_back();

    // dimensions used -> LDS rows, ascending
    {
      std::vector<char> used;
      const int32_t *e = exps;
      for (int i = 0; i < W; ++i) {
        if ((int)used.size() < Dw[i]) used.resize(Dw[i], 0);
        for (int k = 0; k < L[i]; ++k)
          for (int d = 0; d < Dw[i]; ++d)
            if (e[k * Dw[i] + d] != 0) used[d] = 1;
        e += (size_t)L[i] * Dw[i];
      }
      for (size_t d = 0; d < used.size(); ++d)
        if (used[d]) {
          b.dim_row[(int)d] = (int)p->row_src.size();
          p->row_src.push_back((int32_t)d);
          p->max_dim = (int)d + 1;
        }
      p->dims_used = (int)p->row_src.size();
    }

    // trie
    const int32_t *e = exps;
    const float *a = alpha;
    int row = 0;
    for (int i = 0; i < W; ++i) {
      int cur = 0;
      // depth 0: every prefix of the word was already output by an earlier word
      // (CachePlan can say so); the word contributes neither rows nor work
      for (int k = 0; k < (depth[i] > 0 ? L[i] : 0); ++k) {
        Key key;
        key.exps.assign(e + (size_t)k * Dw[i], e + (size_t)(k + 1) * Dw[i]);
        while (!key.exps.empty() && key.exps.back() == 0) key.exps.pop_back();
        key.alpha_bits = weighting ? fbits(a[k]) : 0u;
        int nxt = -1;
        if (share) {
          auto it = b.child_map[cur].find(key);
          if (it != b.child_map[cur].end()) nxt = it->second;
        }
        if (nxt < 0) {
          nxt = (int)b.trie.size();
          TrieNode tn;
          tn.exps = key.exps;
          tn.alpha_bits = key.alpha_bits;
          tn.parent = cur;
          tn.depth = k + 1;
          b.trie.push_back(tn);
          b.child_map.emplace_back();
          b.trie[cur].children.push_back(nxt);
          if (share) b.child_map[cur][key] = nxt;
        }
        cur = nxt;
        // prefix of length k+1 is output iff L-k <= depth (semiring.py:120,152)
        if (L[i] - k <= depth[i]) b.trie[cur].emit.push_back(row + depth[i] - (L[i] - k));
      }
      row += depth[i];
      e += (size_t)L[i] * Dw[i];
      if (a) a += L[i];
    }
    p->K = row;

    // register all alphas up front so LDS row numbers are final before emission
    if (weighting)
      for (size_t t = 1; t < b.trie.size(); ++t) b.alpha_id(b.trie[t].alpha_bits);
    for (int j = 0; j < p->aux_tables(); ++j) p->row_src.push_back(-(int32_t)(1 + j));

    // DFS program, one unit per root child
    p->unit_begin.push_back(0);
    int unit = 0;
    for (int c : b.trie[0].children) {
      b.emit_node(c, 0, 0, unit++);
      p->unit_begin.push_back((int32_t)p->nodes.size());
    }
    for (int u = 0; u < p->units(); ++u) {
      double cst = 0;
      for (int i = p->unit_begin[u]; i < p->unit_begin[u + 1]; ++i) cst += node_cost(p->nodes[i]);
      p->unit_cost.push_back(cst);
    }
    if (p->row_src.size() > (size_t)FAC_ROW_MASK) {
      err = "fr_plan_create: too many staged rows";
      delete p;
      return nullptr;
    }
    if (p->levels <= kMaxLevels) return p;
    // too deep for the register-frame walk: fall back to one chain per word
    delete p;
    if (!share) break;
  }
  err = "fr_plan_create: plan exceeds the frame limit";
  return nullptr;
}

// The device record of plan node i (see NodeRec).
static NodeRec node_record(const Plan &p, int i) {
  const NodeDesc &nd = p.nodes[i];
  NodeRec r{};
  // (a letter whose exponents cancel, [2-2], has no factor at all: the factor-table path
  // multiplies by none, the inline one always by its first)
  bool slow = nd.fac_count > kRecInlineFactors || nd.fac_count == 0 || p.letter_sum;
  for (int j = 0; j < nd.fac_count; ++j)
    if (p.multiplicative() && (p.factors[nd.fac_begin + j] & FAC_DIV)) slow = true;
  const bool kids = (nd.flags & F_CHILDREN) != 0;
  const bool need2 = kids && nd.z_mul >= 0;
  const bool need1 = nd.emit_count > 0 || (kids && !need2);
  const int fl = nd.flags | (slow ? F_SLOW : 0) | (need1 ? F_NEED1 : 0) | (need2 ? F_NEED2 : 0) |
                 (nd.emit_count > 0 ? F_EMIT : 0);
  r.w[0] = (nd.level & 0xff) | ((fl & 0xff) << 8);
  // (the first half of a record serves a multiply-only node alone: walk_fused.h)
  r.w[1] = (nd.fac_count & 0xffff) | ((nd.z_mul + 1) << 16) | ((nd.emit_mul + 1) << 24);
  for (int j = 0; j < kRecInlineFactors && j < nd.fac_count; ++j)
    r.w[2 + j] = p.multiplicative() ? (p.factors[nd.fac_begin + j] & FAC_ROW_MASK)
                                    : p.factors[nd.fac_begin + j];
  r.w[6] = nd.emit_count;
  for (int j = 0; j < kRecInlineEmits && j < nd.emit_count; ++j)
    r.w[7 + j] = p.emit_rows[nd.emit_begin + j];
  r.w[9] = i;
  r.w[10] = nd.emit_mul;
  r.w[11] = nd.z_mul;
  r.w[12] = nd.fac_begin;
  r.w[13] = nd.emit_begin;
  return r;
}

GroupedProgram &grouped(Plan &p, int G) {
  G = std::max(1, std::min(G, std::max(1, p.units())));
  auto it = p.programs.find(G);
  if (it != p.programs.end()) return it->second;
  GroupedProgram gp;
  gp.groups = G;
  const int U = p.units();
  // LPT: heaviest unit first onto the lightest group (ties: lowest index), then
  // each group keeps its units in plan order
  std::vector<int> order(U);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int x, int y) { return p.unit_cost[x] > p.unit_cost[y]; });
  std::vector<double> load(G, 0.0);
  std::vector<std::vector<int>> members(G);
  for (int u : order) {
    int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    load[g] += p.unit_cost[u];
    members[g].push_back(u);
  }
  gp.group_begin.push_back(0);
  for (int g = 0; g < G; ++g) {
    std::sort(members[g].begin(), members[g].end());
    gp.group_row_begin.push_back((int32_t)gp.slot_rows.size());
    for (int u : members[g])
      for (int i = p.unit_begin[u]; i < p.unit_begin[u + 1]; ++i) {
        const NodeDesc &nd = p.nodes[i];
        const NodeRec r = node_record(p, i);
        gp.recs.push_back(r);
        for (int j = 0; j < nd.emit_count; ++j) gp.slot_rows.push_back(p.emit_rows[nd.emit_begin + j]);
      }
    NodeRec end{};
    end.w[0] = kRecSentinelLevel;
    gp.recs.push_back(end);
    gp.group_begin.push_back((int32_t)gp.recs.size());
  }
  // shapes: level | flags << 8 | output rows (3: more than two, counted at run time) << 20 (the
  // count of a letter's inline factors stays a run-time test: it would triple the shapes - a
  // plan like of_weight(6,2) has 13 of them this way, the six most frequent cover 83 % of its
  // nodes - and the bodies are found by a chain of tests)
  {
    std::map<int32_t, int> count;
    std::vector<int32_t> of(gp.recs.size(), -1);
    for (size_t i = 0; i < gp.recs.size(); ++i) {
      const NodeRec &r = gp.recs[i];
      if ((r.w[0] & 0xff) == kRecSentinelLevel) continue;
      const int fl = (r.w[0] >> 8) & 0xff;
      of[i] = (r.w[0] & 0xff) | fl << 8 | std::min(r.w[6], 3) << 20;
      ++count[of[i]];
    }
    std::vector<std::pair<int, int32_t>> order;
    for (const auto &kv : count) order.push_back({-kv.second, kv.first});
    std::sort(order.begin(), order.end());
    std::map<int32_t, int> id;
    for (const auto &o : order) {
      id[o.second] = (int)gp.shapes.size();
      gp.shapes.push_back(o.second);
    }
    gp.shape_ids.resize(gp.recs.size());
    for (size_t i = 0; i < gp.recs.size(); ++i) gp.shape_ids[i] = of[i] < 0 ? -1 : id[of[i]];
  }
  return p.programs.emplace(G, std::move(gp)).first->second;
}

// The plan in pieces (plan.h, PiecedProgram).
PiecedProgram &pieced(Plan &p, int max_piece, int unit_nodes_limit) {
  max_piece = std::max(2, std::min(max_piece, 4096));
  auto hit = p.pieced.find(max_piece);
  if (hit != p.pieced.end()) return hit->second;
  PiecedProgram pp;
  pp.max_piece = max_piece;
  const int n = (int)p.nodes.size();
  if (p.cos || p.letter_sum || n == 0) return p.pieced.emplace(max_piece, std::move(pp)).first->second;
  // the trie behind the DFS order: an only child continues its parent's frame (F_CHAIN), the
  // other nodes hang below the last node of the level above
  std::vector<int> parent(n, -1), below(n, 0);
  std::vector<std::vector<int>> kids(n + 1);   // kids[i + 1]; kids[0]: the root's
  for (int u = 0; u < p.units(); ++u) {
    int last_at[kMaxLevels + 1];
    for (int &v : last_at) v = -1;
    for (int i = p.unit_begin[u]; i < p.unit_begin[u + 1]; ++i) {
      const int lv = p.nodes[i].level;
      parent[i] = (p.nodes[i].flags & F_CHAIN) ? last_at[lv] : (lv > 0 ? last_at[lv - 1] : -1);
      last_at[lv] = i;
      kids[parent[i] + 1].push_back(i);
    }
  }
  for (int i = n - 1; i >= 0; --i)
    if (parent[i] >= 0) below[parent[i]] += 1 + below[i];
  // Equal sub-tries get equal numbers - what a node computes (flags, weight rows, output rows,
  // the factors themselves) and the numbers of its children, SORTED: the order of the children
  // is the plan's word order, which differs from parent to parent, and since the walk order of
  // the output rows is ours to choose (row_of_walk) the children are walked in sorted order so
  // that equal sub-tries are equal record for record.
  std::vector<int> canon(n, 0);
  {
    std::map<std::vector<int32_t>, int> intern;
    for (int i = n - 1; i >= 0; --i) {
      const NodeDesc &nd = p.nodes[i];
      const NodeRec r = node_record(p, i);
      std::vector<int> &ks = kids[i + 1];
      std::stable_sort(ks.begin(), ks.end(), [&](int x, int y) { return canon[x] < canon[y]; });
      std::vector<int32_t> key{r.w[0] >> 8, r.w[1], nd.emit_count, nd.fac_count};
      for (int j = 0; j < nd.fac_count; ++j) key.push_back(p.factors[nd.fac_begin + j]);
      for (int c : ks) key.push_back(canon[c]);
      canon[i] = intern.emplace(key, (int)intern.size()).first->second;
    }
    std::stable_sort(kids[0].begin(), kids[0].end(), [&](int x, int y) { return canon[x] < canon[y]; });
  }
  // the nodes of the sub-trie below (and with) C in the sorted walk order
  auto subtrie = [&](int C, std::vector<int> &out) {
    std::vector<int> stack{C};
    while (!stack.empty()) {
      const int x = stack.back();
      stack.pop_back();
      out.push_back(x);
      const std::vector<int> &ks = kids[x + 1];
      for (auto it = ks.rbegin(); it != ks.rend(); ++it) stack.push_back(*it);
    }
  };
  // sub-tries of at least this many nodes below a node get an item of their own (the node goes
  // into the chain): such bodies repeat; smaller siblings share a body
  const int smin = std::max(1, std::min(24, max_piece / 2));
  struct Item {
    std::vector<int> chain, roots;
    int type = -1;
    std::vector<char> emits;   // per chain node: this item outputs its rows
  };
  std::vector<Item> items;
  auto chain_of = [&](int P) {
    std::vector<int> c;
    for (int x = P; x >= 0; x = parent[x]) c.push_back(x);
    std::reverse(c.begin(), c.end());
    return c;
  };
  auto add_item = [&](int P, const std::vector<int> &roots) {
    Item it;
    it.chain = chain_of(P);
    it.roots = roots;
    items.push_back(std::move(it));
  };
  std::vector<int> todo{-1};
  while (!todo.empty()) {
    const int P = todo.back();
    todo.pop_back();
    const std::vector<int> &ks = kids[P + 1];
    if ((P < 0 ? n : below[P]) <= max_piece) {
      add_item(P, ks);
      continue;
    }
    std::vector<int> small, deeper;
    int small_nodes = 0;
    for (int C : ks) {
      if (below[C] > max_piece) {
        deeper.push_back(C);
      } else if (below[C] >= smin) {
        add_item(C, kids[C + 1]);
      } else {
        if (small_nodes + 1 + below[C] > max_piece) {
          add_item(P, small);
          small.clear();
          small_nodes = 0;
        }
        small.push_back(C);
        small_nodes += 1 + below[C];
      }
    }
    if (!small.empty()) add_item(P, small);
    for (auto it = deeper.rbegin(); it != deeper.rend(); ++it) todo.push_back(*it);
  }
  // types: bodies with equal records (levels counted from the chain's end, rows from the body's
  // first) are one
  std::map<std::vector<int32_t>, int> type_of;
  std::vector<std::vector<int>> members;   // items of a type, in order
  for (size_t ii = 0; ii < items.size(); ++ii) {
    Item &it = items[ii];
    const int P = it.chain.empty() ? -1 : it.chain.back();
    const int shift = P < 0 ? 1 : -p.nodes[P].level;
    std::vector<int32_t> key, w;
    std::vector<int32_t> rel_emits;
    int rows = 0, nodes = 0, levels = 1, widest = 0;
    bool ok = true;
    std::vector<int> order;
    for (int C : it.roots) subtrie(C, order);
    for (int i : order) {
        NodeRec r = node_record(p, i);
        const NodeDesc &nd = p.nodes[i];
        const int lv = nd.level + shift;
        if (lv < 0 || lv >= kMaxLevels) ok = false;
        levels = std::max(levels, lv + 1);
        r.w[0] = (lv & 0xff) | (r.w[0] & ~0xff);
        r.w[7] = r.w[8] = 0;
        if (nd.emit_count > 0) r.w[7] = rows;
        if (nd.emit_count > 1) r.w[8] = rows + 1;
        r.w[9] = nodes;
        r.w[13] = (int32_t)rel_emits.size();
        for (int j = 0; j < nd.emit_count; ++j) rel_emits.push_back(rows + j);
        rows += nd.emit_count;
        widest = std::max(widest, nd.emit_count);
        ++nodes;
        key.push_back(r.w[0]);
        key.push_back(r.w[1]);
        key.push_back(nd.emit_count);
        for (int j = 0; j < nd.fac_count; ++j) key.push_back(p.factors[nd.fac_begin + j]);
        w.insert(w.end(), r.w, r.w + 16);
      }
    if (!ok) return p.pieced.emplace(max_piece, std::move(pp)).first->second;
    auto f = type_of.find(key);
    if (f == type_of.end()) {
      PieceType t;
      NodeRec end{};
      end.w[0] = kRecSentinelLevel;
      w.insert(w.end(), end.w, end.w + 16);
      t.body_w = w;
      t.body_nodes = nodes;
      t.body_rows = rows;
      t.levels = levels;
      t.widest_node = widest;
      t.emit_rows = rel_emits;
      t.recs.resize(w.size() / 16);
      std::memcpy(t.recs.data(), w.data(), w.size() * 4);
      f = type_of.emplace(key, (int)pp.types.size()).first;
      pp.types.push_back(std::move(t));
      members.emplace_back();
    }
    it.type = f->second;
    members[it.type].push_back((int)ii);
  }
  // every chain node is output by the first item whose chain holds it
  {
    std::vector<char> out(n, 0);
    int covered = 0;
    for (Item &it : items) {
      for (int x : it.chain) {
        it.emits.push_back(out[x] ? 0 : 1);
        if (!out[x]) ++covered;
        out[x] = 1;
      }
      for (int C : it.roots) covered += 1 + below[C];
    }
    if (covered != n) return p.pieced.emplace(max_piece, std::move(pp)).first->second;
  }
  // units (a few items of one type on one series) and the walk order of the output rows
  pp.row_of_walk.assign(p.K, -1);
  int q = 0;
  for (size_t ti = 0; ti < pp.types.size(); ++ti) {
    PieceType &t = pp.types[ti];
    // (a unit = several items behind one staging of the series' rows: kPieceUnitNodes)
    const int limit = unit_nodes_limit > 0 ? unit_nodes_limit : std::max(kPieceUnitNodes, max_piece + max_piece / 2);
    int unit_nodes = 0;
    t.unit_begin.push_back(0);
    auto close_unit = [&](int row_end) {
      t.unit_begin.push_back((int32_t)t.items.size() / 4);
      t.max_unit_nodes = std::max(t.max_unit_nodes, unit_nodes);
      t.max_unit_rows = std::max(t.max_unit_rows, row_end - t.unit_row0.back());
      unit_nodes = 0;
    };
    for (int ii : members[ti]) {
      const Item &it = items[ii];
      const int size = (int)it.chain.size() + t.body_nodes;
      if (unit_nodes > 0 && unit_nodes + size > limit) close_unit(q);
      if (unit_nodes == 0) t.unit_row0.push_back(q);
      const int32_t chain_off = (int32_t)t.recs.size() * 64;
      for (size_t c = 0; c < it.chain.size(); ++c) {
        const int x = it.chain[c];
        const NodeDesc &nd = p.nodes[x];
        NodeRec r = node_record(p, x);
        const int ne = it.emits[c] ? nd.emit_count : 0;
        const bool need2 = nd.z_mul >= 0, need1 = ne > 0 || !need2;
        const int fl = ((r.w[0] >> 8) & F_SLOW) | F_CHAIN | F_CHILDREN | (need1 ? F_NEED1 : 0) |
                       (need2 ? F_NEED2 : 0) | (ne > 0 ? F_EMIT : 0);
        r.w[0] = fl << 8;   // level 0
        r.w[6] = ne;
        r.w[7] = r.w[8] = 0;
        if (ne > 0) r.w[7] = q;
        if (ne > 1) r.w[8] = q + 1;
        r.w[13] = (int32_t)t.emit_rows.size();
        for (int j = 0; j < ne; ++j) {
          t.emit_rows.push_back(q + j);
          pp.row_of_walk[q + j] = p.emit_rows[nd.emit_begin + j];
        }
        q += ne;
        t.widest_node = std::max(t.widest_node, ne);
        t.recs.push_back(r);
      }
      NodeRec end{};
      end.w[0] = kRecSentinelLevel;
      t.recs.push_back(end);
      t.items.push_back(chain_off);
      t.items.push_back(q);
      t.items.push_back(unit_nodes);
      t.items.push_back(0);
      std::vector<int> order;
      for (int C : it.roots) subtrie(C, order);
      for (int i : order)
        for (int j = 0; j < p.nodes[i].emit_count; ++j)
          pp.row_of_walk[q++] = p.emit_rows[p.nodes[i].emit_begin + j];
      unit_nodes += size;
      pp.chain_nodes += (int)it.chain.size();
    }
    if (unit_nodes > 0) close_unit(q);
  }
  pp.ok = q == p.K;
  for (int32_t r : pp.row_of_walk)
    if (r < 0) pp.ok = false;
  return p.pieced.emplace(max_piece, std::move(pp)).first->second;
}

// Static schedule: see plan.h / walk.h.  Greedy list scheduling per group - among the nodes
// whose parent is done and whose rows are all staged, take children of open frames first
// (frames close early), then plan order; when nothing is ready, complete the row that
// readies the most nodes.  So the first nodes of a unit run on the first row alone while the
// loads of the other rows are still in flight.
static bool schedule_group(const NodeRec *recs, int n, int R, bool prefetch,
                           const std::vector<int32_t> &emit_rows, std::vector<NodeRec> &out,
                           int &frames_hi, int &need_mask) {
  std::vector<int> parent(n, -1), pending(n, 0), need(n, 0), last_at(kMaxLevels + 1, -1);
  int need_all = 0;
  for (int i = 0; i < n; ++i) {
    const NodeRec &r = recs[i];
    const int lv = r.w[0] & 0xff, fl = r.w[0] >> 8;
    if ((fl & F_SLOW) || (r.w[1] & 0xffff) > kRecInlineFactors || lv >= kMaxLevels) return false;
    parent[i] = (fl & F_CHAIN) ? last_at[lv] : (lv > 0 ? last_at[lv - 1] : -1);
    last_at[lv] = i;
    if (parent[i] >= 0) ++pending[parent[i]];
    for (int j = 0; j < (r.w[1] & 0xffff); ++j) need[i] |= 1 << (r.w[2 + j] & FAC_ROW_MASK);
    need_all |= need[i];
  }
  need_mask = need_all;
  std::vector<int> frame(n, -1);
  std::vector<char> done(n, 0);
  bool frame_used[kStaticMaxFrames] = {false, false, false, false};
  int staged = 0, left = n;
  while (left > 0) {
    int pick = -1, pick_rank = -1;
    int free_frames = 0;
    for (int f = 0; f < kStaticMaxFrames; ++f) free_frames += frame_used[f] ? 0 : 1;
    for (int i = 0; i < n; ++i) {
      if (done[i] || (need[i] & ~staged) != 0) continue;
      if (parent[i] >= 0 && !done[parent[i]]) continue;
      // a node with children opens a frame: its parent's frame is free again when this is
      // the parent's last child
      const bool frees_parent = parent[i] >= 0 && pending[parent[i]] == 1;
      if (pending[i] > 0 && free_frames == 0 && !frees_parent) continue;
      const int rank = parent[i] >= 0 ? 2 : 1;   // children of open frames first
      if (rank > pick_rank) {
        pick = i;
        pick_rank = rank;
      }
    }
    if (pick < 0) {
      int best = -1, best_gain = -1;
      for (int r = 0; r < R; ++r) {
        if ((staged & (1 << r)) || !(need_all & (1 << r))) continue;
        int gain = 0;
        for (int i = 0; i < n; ++i)
          if (!done[i] && (need[i] & ~(staged | (1 << r))) == 0) ++gain;
        if (gain > best_gain) {
          best = r;
          best_gain = gain;
        }
      }
      if (best < 0) return false;   // every row staged and still nothing ready: out of frames
      NodeRec e{};
      e.w[0] = kSchedStage;
      e.w[1] = best;
      out.push_back(e);
      staged |= 1 << best;
      if (prefetch && staged == need_all) {   // the row registers are free from here on
        NodeRec pf{};
        pf.w[0] = kSchedPrefetch;
        out.push_back(pf);
      }
      continue;
    }
    NodeRec e = recs[pick];
    const int par = parent[pick];
    e.w[14] = par >= 0 ? frame[par] : -1;
    if (par >= 0 && --pending[par] == 0) frame_used[frame[par]] = false;
    e.w[15] = -1;
    if (pending[pick] > 0) {
      int f = 0;
      while (f < kStaticMaxFrames && frame_used[f]) ++f;
      if (f == kStaticMaxFrames) return false;
      frame_used[f] = true;
      frame[pick] = f;
      e.w[15] = f;
      frames_hi = std::max(frames_hi, f + 1);
    }
    // output rows beyond the two a record holds become immediates of kSchedEmits entries
    const int ne = e.w[6];
    if (ne > kRecInlineEmits) e.w[6] = kRecInlineEmits;
    out.push_back(e);
    for (int j = kRecInlineEmits; j < ne;) {
      NodeRec m{};
      m.w[0] = kSchedEmits;
      int k = 0;
      for (; k < kSchedEmitsPerEntry && j < ne; ++k, ++j) m.w[2 + k] = emit_rows[e.w[13] + j];
      m.w[1] = k;
      out.push_back(m);
    }
    done[pick] = 1;
    --left;
  }
  NodeRec end{};
  end.w[0] = kRecSentinelLevel;
  out.push_back(end);
  return true;
}

StaticSchedule static_schedule(Plan &p, int G) {
  StaticSchedule sc;
  if (p.cos || p.aux_tables() != 0 || p.letter_sum || p.nodes.empty() ||
      (int)p.nodes.size() > kStaticMaxNodes || p.rows_staged() > kStaticMaxRows)
    return sc;
  const GroupedProgram &gp = grouped(p, G);
  const int R = p.rows_staged();
  int frames_hi = 1;
  for (int g = 0; g < gp.groups; ++g) {
    const int b = gp.group_begin[g], n = gp.group_begin[g + 1] - 1 - b;   // without the sentinel
    int mask = 0;
    sc.group_begin.push_back((int32_t)sc.entries.size());
    // the next unit of a workgroup is another group when G > 1: no prefetch entry then
    if (!schedule_group(gp.recs.data() + b, n, R, gp.groups == 1, p.emit_rows, sc.entries, frames_hi,
                        mask))
      return sc;
    sc.group_rows.push_back(mask);
  }
  sc.groups = gp.groups;
  sc.rows = R;
  sc.frames = frames_hi;
  sc.row_src = p.row_src;
  sc.ok = true;
  return sc;
}

}  // namespace fr

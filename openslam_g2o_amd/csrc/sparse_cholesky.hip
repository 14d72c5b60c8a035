// Multifrontal sparse block Cholesky: host symbolic analysis + gfx950 kernels.
// See sparse_cholesky.h for the reference functions this replaces.
#include "sparse_cholesky.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <type_traits>

namespace g2ohip {

constexpr int kFactorThreads = 256;
constexpr int kFwdChildren = 4;   // fused forward sweep: children per front handled by the factor kernel
constexpr int kGatherChildren = 4;   // scratch-slab fronts: children whose update matrices the merged level launch gathers at load time
constexpr int kGatherInts = 1024;                      // ... gather table of a front (header + children x blocks) staged in LDS
constexpr int kGatherHeader = 16;                      // ... its header: children, offsets of their update matrices
constexpr int kGatherLdsOff = 3 * 64 * 65 + 64 + 128;   // ... behind the level launch's own regions (doubles)
constexpr int kFactorThreadsGlobal = 512;
constexpr int kChainU = 6;  // doubles per thread that carry an update matrix from one chain front to the next
// register-resident wave kernel (wave_front.inc): limits of a front
constexpr int kPanelSolveWgs = 512;      // big_panel_solve_kernel (pivot blocks + panel rows of a level in one launch) on launches of at most this many workgroups
constexpr int kEgThreads = 256;           // big_extend_gather_kernel: threads per workgroup = scalar rows per chunk (128 / 64: no different)
constexpr int kFillChunk = 4096;         // doubles zeroed by one workgroup of big_fill_kernel
constexpr int kWvNPV = 24;             // pivot columns (scalars)
constexpr int kWvNTL = 3;              // 16-row tiles of boundary rows (48 rows)
constexpr int kWvT = (kWvNPV + 16 * kWvNTL + 1 + 15) / 16;   // 16 x 16 tiles per side of the front incl. the right-hand side row
constexpr int kWvTiles = kWvT * (kWvT + 1) / 2;

// =====================================================================================
// Host: nested dissection on the block graph (George-Liu automatic nested dissection:
// BFS level structure from a pseudo-peripheral node, separator = the part of the middle
// level that touches the next level).
// =====================================================================================
namespace {

struct NdWork {
  const std::vector<int>& xadj;
  const std::vector<int>& adj;
  std::vector<int> region, level, queue;
  NdWork(int n, const std::vector<int>& xa, const std::vector<int>& a) : xadj(xa), adj(a), region(n, 0), level(n, -1), queue() {
    queue.reserve(n);
  }
  // BFS inside region rid from s; fills queue (visit order) and level[]; returns #levels
  int bfs(int s, int rid) {
    queue.clear();
    queue.push_back(s);
    level[s] = 0;
    size_t head = 0;
    int maxl = 0;
    while (head < queue.size()) {
      int v = queue[head++];
      for (int q = xadj[v]; q < xadj[v + 1]; ++q) {
        int u = adj[q];
        if (region[u] != rid || level[u] >= 0) continue;
        level[u] = level[v] + 1;
        maxl = level[u];
        queue.push_back(u);
      }
    }
    return maxl + 1;
  }
  void clear_levels() {
    for (int v : queue) level[v] = -1;
  }
};

}  // namespace

void nested_dissection(int n, const std::vector<int>& xadj, const std::vector<int>& adj, int leaf, std::vector<int>& perm) {
  perm.assign(n, -1);
  if (n == 0) return;
  NdWork W(n, xadj, adj);
  struct Item {
    std::vector<int> nodes;
    int base, rid;
  };
  std::vector<Item> stack;
  {
    Item it;
    it.nodes.resize(n);
    std::iota(it.nodes.begin(), it.nodes.end(), 0);
    it.base = 0;
    it.rid = 0;
    stack.push_back(std::move(it));
  }
  int next_rid = 1;
  if (leaf < 1) leaf = 1;
  while (!stack.empty()) {
    Item it = std::move(stack.back());
    stack.pop_back();
    const int sz = (int)it.nodes.size();
    if (sz == 0) continue;
    // connected component of the first node
    int nlev = W.bfs(it.nodes[0], it.rid);
    if ((int)W.queue.size() < sz) {
      // split off this component; the rest is handled as another item (independent subtrees)
      Item comp, rest;
      comp.rid = next_rid++;
      rest.rid = next_rid++;
      comp.base = it.base;
      comp.nodes = W.queue;
      for (int v : comp.nodes) W.region[v] = comp.rid;
      W.clear_levels();
      rest.base = it.base + (int)comp.nodes.size();
      rest.nodes.reserve(sz - comp.nodes.size());
      for (int v : it.nodes)
        if (W.region[v] == it.rid) {
          W.region[v] = rest.rid;
          rest.nodes.push_back(v);
        }
      stack.push_back(std::move(rest));
      stack.push_back(std::move(comp));
      continue;
    }
    // connected: pseudo-peripheral start (two more sweeps)
    for (int sweep = 0; sweep < 2; ++sweep) {
      int far = W.queue.back();
      W.clear_levels();
      int nl2 = W.bfs(far, it.rid);
      if (nl2 <= nlev && sweep > 0) {
        nlev = nl2;
        break;
      }
      nlev = nl2;
    }
    if (sz <= leaf || nlev < 3) {
      // leaf: Cuthill-McKee order.  Start = the node of minimum degree (inside the region) in the last level of the
      // level structure (George & Liu's choice of a pseudo-peripheral node); the unnumbered neighbours of a node
      // are numbered by increasing degree.  On a band (a stretch of a camera trajectory) this is the natural order
      // from one end, in either direction: every column then reaches at most `half-width` blocks down, which is
      // what the sliding-window kernel (band_chain.inc) relies on; the plain queue order above numbers the
      // neighbours of the start node farthest first.
      auto deg = [&](int v) {
        int d = 0;
        for (int q = xadj[v]; q < xadj[v + 1]; ++q)
          if (W.region[adj[q]] == it.rid) ++d;
        return d;
      };
      int start = W.queue.back();
      {
        const int last_level = W.level[start];
        int best = deg(start);
        for (int k = sz - 1; k >= 0 && W.level[W.queue[k]] == last_level; --k) {
          const int d = deg(W.queue[k]);
          if (d < best || (d == best && W.queue[k] < start)) {
            best = d;
            start = W.queue[k];
          }
        }
      }
      W.clear_levels();
      std::vector<int> order;
      order.reserve(sz);
      order.push_back(start);
      W.level[start] = 0;
      std::vector<std::pair<int, int>> nbr;
      for (size_t head = 0; head < order.size(); ++head) {
        const int v = order[head];
        nbr.clear();
        for (int q = xadj[v]; q < xadj[v + 1]; ++q) {
          const int u = adj[q];
          if (W.region[u] == it.rid && W.level[u] < 0) {
            W.level[u] = W.level[v] + 1;
            nbr.emplace_back(deg(u), u);
          }
        }
        std::sort(nbr.begin(), nbr.end());
        for (const auto& pr : nbr) order.push_back(pr.second);
      }
      for (int k = 0; k < sz; ++k) perm[it.base + k] = order[k];
      for (int v : order) W.level[v] = -1;
      continue;
    }
    // level sizes
    std::vector<int> lsize(nlev, 0);
    for (int v : W.queue) lsize[W.level[v]]++;
    int best = -1;
    long best_cost = -1;
    int before = 0;
    int fallback = 1;
    long fallback_bal = -1;
    for (int m = 0; m < nlev; ++m) {
      if (m >= 1 && m <= nlev - 2) {
        int after = sz - before - lsize[m];
        int bal = std::min(before, after);
        if (bal > fallback_bal) {
          fallback_bal = bal;
          fallback = m;
        }
        if (bal * 4 >= sz) {  // each side at least 25 %
          long cost = (long)lsize[m] * 1000000L - bal;  // smallest separator, then best balance
          if (best < 0 || cost < best_cost) {
            best = m;
            best_cost = cost;
          }
        }
      }
      before += lsize[m];
    }
    const int m = best >= 0 ? best : fallback;
    Item A, B;
    A.rid = next_rid++;
    B.rid = next_rid++;
    std::vector<int> sep;
    for (int v : W.queue) {
      int lv = W.level[v];
      if (lv < m)
        A.nodes.push_back(v);
      else if (lv > m)
        B.nodes.push_back(v);
      else {
        bool touches = false;
        for (int q = xadj[v]; q < xadj[v + 1] && !touches; ++q) {
          int u = adj[q];
          if (W.region[u] == it.rid && W.level[u] == m + 1) touches = true;
        }
        if (touches)
          sep.push_back(v);
        else
          A.nodes.push_back(v);
      }
    }
    W.clear_levels();
    for (int v : A.nodes) W.region[v] = A.rid;
    for (int v : B.nodes) W.region[v] = B.rid;
    for (int v : sep) W.region[v] = -1;  // ordered
    A.base = it.base;
    B.base = it.base + (int)A.nodes.size();
    int sbase = B.base + (int)B.nodes.size();
    for (size_t k = 0; k < sep.size(); ++k) perm[sbase + k] = sep[k];
    stack.push_back(std::move(A));
    stack.push_back(std::move(B));
  }
}

// =====================================================================================
// Host: symbolic analysis
// =====================================================================================
namespace {

// lower pattern (CSC: for column j rows i>j) and its transpose (for row i the columns j<i)
// of the permuted matrix
void permuted_lower(int nb, const int* colptr, const int* rowidx, const std::vector<int>& iperm, std::vector<int>& cp,
                    std::vector<int>& ci, std::vector<int>& rp, std::vector<int>& ri) {
  cp.assign(nb + 1, 0);
  rp.assign(nb + 1, 0);
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      int r = rowidx[q];
      if (r == c) continue;
      int a = iperm[r], b = iperm[c];
      int i = std::max(a, b), j = std::min(a, b);
      cp[j + 1]++;
      rp[i + 1]++;
    }
  for (int k = 0; k < nb; ++k) {
    cp[k + 1] += cp[k];
    rp[k + 1] += rp[k];
  }
  ci.assign(cp[nb], 0);
  ri.assign(rp[nb], 0);
  std::vector<int> wc(cp.begin(), cp.end() - 1), wr(rp.begin(), rp.end() - 1);
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      int r = rowidx[q];
      if (r == c) continue;
      int a = iperm[r], b = iperm[c];
      int i = std::max(a, b), j = std::min(a, b);
      ci[wc[j]++] = i;
      ri[wr[i]++] = j;
    }
  for (int j = 0; j < nb; ++j) std::sort(ci.begin() + cp[j], ci.begin() + cp[j + 1]);
}

void etree(int nb, const std::vector<int>& rp, const std::vector<int>& ri, std::vector<int>& parent) {
  parent.assign(nb, -1);
  std::vector<int> anc(nb, -1);
  for (int k = 0; k < nb; ++k)
    for (int q = rp[k]; q < rp[k + 1]; ++q) {
      int i = ri[q];
      while (i != -1 && i < k) {
        int nx = anc[i];
        anc[i] = k;
        if (nx == -1) parent[i] = k;
        i = nx;
      }
    }
}

void postorder(int nb, const std::vector<int>& parent, std::vector<int>& post) {
  std::vector<int> head(nb, -1), next(nb, -1);
  for (int j = nb - 1; j >= 0; --j)
    if (parent[j] >= 0) {
      next[j] = head[parent[j]];
      head[parent[j]] = j;
    }
  post.clear();
  post.reserve(nb);
  std::vector<int> stk;
  for (int r = 0; r < nb; ++r) {
    if (parent[r] >= 0) continue;
    stk.push_back(r);
    while (!stk.empty()) {
      int v = stk.back();
      int c = head[v];
      if (c == -1) {
        post.push_back(v);
        stk.pop_back();
      } else {
        head[v] = next[c];
        stk.push_back(c);
      }
    }
  }
}

}  // namespace

void SparseCholesky::analyze(int nb, const int* colptr, const int* rowidx, hipStream_t st, bool host_only) {
  struct HostOnlyScope {
    bool prev;
    explicit HostOnlyScope(bool on) : prev(host_only_flag()) { host_only_flag() = on; }
    ~HostOnlyScope() { host_only_flag() = prev; }
  } host_scope(host_only);
  auto t0 = std::chrono::steady_clock::now();
  const int bs = bs_;
  CholSymbolic& S = sym_;
  S = CholSymbolic();
  S.nb = nb;
  S.bs = bs;
  // --- block graph
  std::vector<int> xadj(nb + 1, 0), adj;
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      int r = rowidx[q];
      if (r > c) throw ArgFailure("analyze: pattern must be upper triangular (row block <= column block)");
      if (r != c) {
        xadj[r + 1]++;
        xadj[c + 1]++;
      }
    }
  for (int k = 0; k < nb; ++k) xadj[k + 1] += xadj[k];
  adj.resize(xadj[nb]);
  {
    std::vector<int> w(xadj.begin(), xadj.end() - 1);
    for (int c = 0; c < nb; ++c)
      for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
        int r = rowidx[q];
        if (r != c) {
          adj[w[r]++] = c;
          adj[w[c]++] = r;
        }
      }
    // duplicates cannot occur: the pattern has unique (r,c)
  }
  std::vector<int> perm0;
  // leaf size: large systems get about as many leaves as the band kernel has chain slots (256 CUs x 8 one-wave chains, one
  // round instead of two and one tree level less: 0.53 -> 0.48 ms at the metric configuration), within what a chain may hold
  // (round 6) nd_leaf = 0 (default): by the shape of the graph.  A camera trajectory's reduced system is a BAND -- the level structure
  // from a pseudo-peripheral node has a handful of blocks per level --: its leaves are the band kernel's chains, long ones (32 blocks
  // and more).  Anything wider (pose graphs, loop closures) gets leaves of 4 blocks: the separators then do the ordering and the tree
  // is a third shallower (sphere 2 200: 57 -> 37 levels, 2.07 -> 1.44 ms per solve; manhattan: fill 2.45 -> 1.89 x the reference's
  // block-AMD; profiles/r6_nd_leaf.txt).
  int nd_leaf = opt.nd_leaf;
  bool band_like = true;   // (an explicit leaf size keeps the band kernel's supernode width)
  if (nd_leaf <= 0) {
    band_like = false;
    if (nb > 0) {
      NdWork W(nb, xadj, adj);
      int nlev = W.bfs(0, 0);
      for (int sweep = 0; sweep < 2; ++sweep) {
        const int far = W.queue.back();
        W.clear_levels();
        nlev = W.bfs(far, 0);
      }
      std::vector<int> lsize(nlev, 0);
      for (int v : W.queue) lsize[W.level[v]]++;
      const int widest = *std::max_element(lsize.begin(), lsize.end());
      band_like = widest <= 12 && (long long)W.queue.size() <= 6LL * nlev;   // (the component of block 0 stands for the graph)
    }
    nd_leaf = band_like ? 32 : 4;
  }
  if (opt.band_kernel && bs == 6 && nd_leaf >= 32) nd_leaf = std::min(std::max(nd_leaf, (nb / std::max(opt.world, 1) + 2047) / 2048), 128);   // (per rank: its share of the chains)
  nested_dissection(nb, xadj, adj, nd_leaf, perm0);
  // --- etree + postorder, compose
  std::vector<int> iperm(nb), cp, ci, rp, ri, parent, post;
  for (int k = 0; k < nb; ++k) iperm[perm0[k]] = k;
  permuted_lower(nb, colptr, rowidx, iperm, cp, ci, rp, ri);
  etree(nb, rp, ri, parent);
  postorder(nb, parent, post);
  S.perm.resize(nb);
  S.iperm.resize(nb);
  for (int k = 0; k < nb; ++k) S.perm[k] = perm0[post[k]];
  for (int k = 0; k < nb; ++k) S.iperm[S.perm[k]] = k;
  permuted_lower(nb, colptr, rowidx, S.iperm, cp, ci, rp, ri);
  etree(nb, rp, ri, S.parent);
  // --- column structures struct(j) = rows > j of L(:,j)
  std::vector<std::vector<int>> st_(nb);
  {
    std::vector<int> mark(nb, -1);
    std::vector<std::vector<int>> kids(nb);
    for (int j = 0; j < nb; ++j)
      if (S.parent[j] >= 0) kids[S.parent[j]].push_back(j);
    for (int j = 0; j < nb; ++j) {
      std::vector<int>& s = st_[j];
      mark[j] = j;
      for (int q = cp[j]; q < cp[j + 1]; ++q) {
        int i = ci[q];
        if (mark[i] != j) {
          mark[i] = j;
          s.push_back(i);
        }
      }
      for (int c : kids[j])
        for (int i : st_[c])
          if (i != j && mark[i] != j) {
            mark[i] = j;
            s.push_back(i);
          }
      std::sort(s.begin(), s.end());
    }
  }
  // --- supernodes (maximal chains with nested structure), capped in width
  // (a pivot panel wider than 64 scalars has no whole-GPU pass: its level would fall back to one workgroup per front -- 430 ms instead of
  // 18 on the 10 000-camera grid graph, profiles/r6_grid_sweep.txt -- so the cap is itself capped)
  const int max_sn_blocks = std::max(1, std::min(opt.max_sn_scalars, 64) / bs);
  // supernodes of the LDS-resident fronts: 24 scalars = the register fronts of the band graphs (wave_front_kernel); graphs that are not a
  // band have few such fronts and gain from twice the width (manhattan 0.845 -> 0.831, sphere 1.442 -> 1.414 ms: fewer levels)
  const int sn_lds = (opt.max_sn_scalars_lds <= 0) ? (band_like ? 24 : 48) : opt.max_sn_scalars_lds;
  // scratch-slab fronts of a thousand rows and more: panels of the full 64 scalars (every panel is a level of whole-GPU passes: fewer of them)
  // (10 000-camera grid graph: 169 -> 143 levels, 21.6 -> 20.6 ms; 49 729 cameras 100.7 -> 97.6; narrower thresholds cost the pose graphs
  // levels of their own: profiles/r6_grid_sweep.txt)
  constexpr int wide_rows = 1024;
  const int wide_blocks = opt.max_sn_scalars >= 48 ? std::max(max_sn_blocks, 64 / bs) : max_sn_blocks;
  S.sn_start.clear();
  {
    // Exact merges (identical structure) always; relaxed merges along a parent chain while the
    // explicit zero blocks stay below opt.relax_zeros of the dense panel and the front still fits LDS.
    long true_blocks = 0;  // structural blocks of the current supernode's columns
    for (int j = 0; j < nb; ++j) {
      bool merge = false;
      if (j > 0 && S.parent[j - 1] == j) {
        const int c0 = S.sn_start.back();
        const long w = j - c0 + 1, nbn = (long)st_[j].size();
        const long total = w * (w + 1) / 2 + w * nbn;
        const long tb = true_blocks + 1 + nbn;
        const bool exact = st_[j - 1].size() == st_[j].size() + 1 && total == tb;
        const size_t m = (size_t)(w + nbn) * bs;
        const bool fits = m * m * 8 <= std::min(opt.lds_front_bytes, opt.relax_front_bytes);
        // narrower panels for the fronts that live in LDS (shorter pivot loops per front, smaller solve panels), wide
        // ones for the scratch-slab fronts (each panel is a whole-GPU pass there)
        const bool lds_class = m * m * 8 <= opt.lds_front_bytes;
        const long cap = lds_class ? std::max(1, std::min(opt.max_sn_scalars, sn_lds) / bs) : (m >= (size_t)wide_rows ? wide_blocks : max_sn_blocks);
        if (w <= cap && (exact || (fits && (double)(total - tb) <= opt.relax_zeros * (double)total))) merge = true;
      }
      if (!merge) {
        S.sn_start.push_back(j);
        true_blocks = 0;
      }
      true_blocks += 1 + (long)st_[j].size();
    }
  }
  const int nf = (int)S.sn_start.size();
  S.sn_start.push_back(nb);
  std::vector<int> sn_of(nb);
  for (int f = 0; f < nf; ++f)
    for (int j = S.sn_start[f]; j < S.sn_start[f + 1]; ++j) sn_of[j] = f;
  S.f_ns.resize(nf);
  S.f_nb.resize(nf);
  S.f_parent.assign(nf, -1);
  S.f_level.assign(nf, 0);
  S.rows_off.assign(nf + 1, 0);
  for (int f = 0; f < nf; ++f) {
    int last = S.sn_start[f + 1] - 1;
    S.f_ns[f] = S.sn_start[f + 1] - S.sn_start[f];
    S.f_nb[f] = (int)st_[last].size();
    S.rows_off[f + 1] = S.rows_off[f] + S.f_nb[f];
    if (!st_[last].empty()) S.f_parent[f] = sn_of[st_[last][0]];
  }
  S.rows.resize(S.rows_off[nf]);
  for (int f = 0; f < nf; ++f) {
    int last = S.sn_start[f + 1] - 1;
    std::copy(st_[last].begin(), st_[last].end(), S.rows.begin() + S.rows_off[f]);
  }
  // local position of a permuted block row i in front f (pivots first, then boundary rows)
  auto local_pos = [&](int f, int i) -> int {
    int c0 = S.sn_start[f], c1 = S.sn_start[f + 1];
    if (i >= c0 && i < c1) return i - c0;
    const int* b = S.rows.data() + S.rows_off[f];
    const int* e = b + S.f_nb[f];
    const int* it = std::lower_bound(b, e, i);
    if (it == e || *it != i) return -1;
    return S.f_ns[f] + (int)(it - b);
  };
  // --- child -> parent relative indices, children lists, levels
  S.rel_off = S.rows_off;
  S.rel.assign(S.rows.size(), -1);
  S.child_off.assign(nf + 1, 0);
  for (int f = 0; f < nf; ++f)
    if (S.f_parent[f] >= 0) S.child_off[S.f_parent[f] + 1]++;
  for (int f = 0; f < nf; ++f) S.child_off[f + 1] += S.child_off[f];
  S.children.resize(S.child_off[nf]);
  {
    std::vector<int> w(S.child_off.begin(), S.child_off.end() - 1);
    for (int f = 0; f < nf; ++f) {
      int p = S.f_parent[f];
      if (p < 0) continue;
      S.children[w[p]++] = f;
      for (int k = 0; k < S.f_nb[f]; ++k) {
        int lp = local_pos(p, S.rows[S.rows_off[f] + k]);
        if (lp < 0) throw StateFailure("symbolic: child row missing in parent front");
        S.rel[S.rel_off[f] + k] = lp;
      }
      S.f_level[p] = std::max(S.f_level[p], S.f_level[f] + 1);  // fronts are postordered: children first
    }
  }
  // --- assembly lists of original blocks
  S.asm_off.assign(nf + 1, 0);
  const int nnzb = colptr[nb];
  std::vector<int> ent_front(nnzb), ent_pos(nnzb);
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      int r = rowidx[q];
      int a = S.iperm[r], b = S.iperm[c];
      int i = std::max(a, b), j = std::min(a, b);
      int tr = (a < b) ? 1 : 0;  // stored block is A(r,c); front holds F(i,j) = A(perm i, perm j)
      if (r == c) tr = 0;
      int f = sn_of[j];
      int lr = local_pos(f, i), lc = j - S.sn_start[f];
      if (lr < 0 || lr >= (1 << 15)) throw StateFailure("symbolic: assembly position out of range");
      ent_front[q] = f;
      ent_pos[q] = lr | (lc << 15) | (tr << 30);
      S.asm_off[f + 1]++;
    }
  for (int f = 0; f < nf; ++f) S.asm_off[f + 1] += S.asm_off[f];
  S.asm_q.resize(nnzb);
  S.asm_pos.resize(nnzb);
  {
    std::vector<int> w(S.asm_off.begin(), S.asm_off.end() - 1);
    for (int q = 0; q < nnzb; ++q) {
      int d = w[ent_front[q]]++;
      S.asm_q[d] = q;
      S.asm_pos[d] = ent_pos[q];
    }
  }
  // --- storage
  S.L_off.resize(nf);
  S.U_off.resize(nf);
  S.w_off.resize(nf);
  S.L_total = S.U_total = S.w_total = 0;
  stats_ = CholStats();
  int nlev = 0;
  for (int f = 0; f < nf; ++f) {
    long long m = (long long)(S.f_ns[f] + S.f_nb[f]) * bs, np = (long long)S.f_ns[f] * bs, nbs = (long long)S.f_nb[f] * bs;
    S.L_off[f] = S.L_total;
    S.L_total += m * np + np;   // panel + reciprocals of its diagonal (used by the triangular sweeps)
    S.U_off[f] = S.U_total;
    S.U_total += (long long)S.f_nb[f] * (S.f_nb[f] + 1) / 2 * bs * bs;   // lower-triangular blocks, packed by block column
    S.w_off[f] = S.w_total;
    S.w_total += nbs;
    stats_.nnzL += (size_t)(np * m - np * (np - 1) / 2);
    stats_.max_front_dim = std::max(stats_.max_front_dim, (size_t)m);
    for (long long k = 0; k < np; ++k) stats_.flops += (double)(m - k) * (double)(m - k);
    nlev = std::max(nlev, S.f_level[f] + 1);
  }
  stats_.n_fronts = nf;
  stats_.n_levels = nlev;
  stats_.bytes_L = (size_t)S.L_total * 8;
  stats_.bytes_U = (size_t)S.U_total * 8;
  // --- tasks: a front whose parent has no other child is fused with it (chain); the workgroup that
  // factorises the child carries the update matrix to the parent in registers.  Both fronts must be
  // LDS-resident and the carried matrix must fit kChainU doubles per thread.
  auto front_dim = [&](int f) { return (size_t)(S.f_ns[f] + S.f_nb[f]) * bs; };
  // LDS-resident iff the dense size is within the class limit AND everything the factor kernel keeps in LDS for this
  // front (packed blocks, rhs vectors, mailboxes, index tables) fits the per-workgroup budget
  long long lds_ints = 0;   // (index part of the last lds_need)
  auto lds_need = [&](int f) {
    const long long nbt = S.f_ns[f] + S.f_nb[f], mm = nbt * bs;
    const long long T_ = (bs % 3 == 0) ? 3 : bs;
    const long long nt0 = (nbt - 1) * bs / T_;
    long long ints = (2 + kVirtInts) * (long long)(S.asm_off[f + 1] - S.asm_off[f]) + std::max(nt0 * (nt0 + 1) / 2, (long long)S.f_nb[f] * (S.f_nb[f] + 1) / 2) + 4;
    for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
      const long long nbc = S.f_nb[S.children[ch]];
      ints += nbc * (nbc + 1) / 2 + nbc;
    }
    lds_ints = 4 * ints;
    return 8 * (nbt * (nbt + 1) / 2 * bs * bs + 2 * mm + 2 * (bs * bs + bs)) + 4 * ints;
  };
  // A launch sizes its LDS by the largest block part and the largest index part among ITS fronts separately (the kernel
  // places the index tables behind a launch-uniform block region), so the two parts are capped separately: their sum of
  // maxima then fits whatever fronts share a launch.  (A front with few rows but a child with hundreds of boundary
  // blocks -- a landmark seen by hundreds of poses -- used to push a launch beyond 160 KB.)
  auto is_lds = [&](int f) {
    if (front_dim(f) * front_dim(f) * 8 > opt.lds_front_bytes) return false;
    const long long total = lds_need(f);
    const long long cap = (long long)opt.lds_budget_bytes;
    return total <= cap && lds_ints <= 24 * 1024 && total - lds_ints <= 134 * 1024;   // (158 KB of the CU's 160 KB)
  };
  std::vector<int> chain_next(nf, -1), has_prev(nf, 0);
  if (opt.fuse_chains)
    for (int f = 0; f < nf; ++f) {
      const int p = S.f_parent[f];
      if (p < 0 || S.child_off[p + 1] - S.child_off[p] != 1 || !is_lds(f) || !is_lds(p)) continue;
      if ((size_t)S.f_nb[f] * (S.f_nb[f] + 1) / 2 * bs * bs > (size_t)kChainU * kFactorThreads) continue;
      if (S.f_nb[f] * bs > 256) continue;   // solve kernels carry the boundary vector in one round
      chain_next[f] = p;
      has_prev[p] = 1;
    }
  if (opt.max_chain_fronts > 0) {
    // Long chains are cut into segments of about equal length (a finer work granularity than whole leaf subtrees for
    // the dependency-driven launch; the update matrix then crosses the cut through HBM instead of registers).
    for (int f = 0; f < nf; ++f) {
      if (has_prev[f] || chain_next[f] < 0) continue;   // chain heads only
      int len = 0;
      for (int g = f; g >= 0; g = chain_next[g]) ++len;
      const int nseg = (len + opt.max_chain_fronts - 1) / opt.max_chain_fronts;
      if (nseg <= 1) continue;
      const int seg = (len + nseg - 1) / nseg;
      int k = 0;
      for (int g = f; g >= 0;) {
        const int nx = chain_next[g];
        if (++k % seg == 0 && nx >= 0) {
          chain_next[g] = -1;
          has_prev[nx] = 0;
        }
        g = nx;
      }
    }
  }
  S.task_ptr.assign(1, 0);
  S.task_fronts.clear();
  std::vector<int> task_of(nf, -1), task_level;
  for (int f = 0; f < nf; ++f) {
    if (has_prev[f]) continue;
    const int t = (int)S.task_ptr.size() - 1;
    int lvl = 0;
    for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) lvl = std::max(lvl, task_level[task_of[S.children[ch]]] + 1);
    for (int g = f; g >= 0; g = chain_next[g]) {
      S.task_fronts.push_back(g);
      task_of[g] = t;
    }
    S.task_ptr.push_back((int)S.task_fronts.size());
    task_level.push_back(lvl);
  }
  const int ntask = (int)task_level.size();
  nlev = 0;
  for (int t = 0; t < ntask; ++t) nlev = std::max(nlev, task_level[t] + 1);
  stats_.n_levels = nlev;
  stats_.n_tasks = ntask;
  // --- multi-GPU partition of the task tree: the top of the tree is split until there are >= world
  // subtrees and the longest-processing-time deal of them to the ranks is balanced within 10%; every task
  // above them is "shared" and executed redundantly by all ranks
  S.task_owner.assign(ntask, opt.world > 1 ? -2 : opt.rank);
  S.xroots.clear();
  if (opt.world > 1) {
    std::vector<std::vector<int>> tkids(ntask);
    std::vector<int> tparent(ntask, -1);
    for (int t = 0; t < ntask; ++t) {
      const int last = S.task_fronts[S.task_ptr[t + 1] - 1];
      const int pf = S.f_parent[last];
      if (pf >= 0) {
        tparent[t] = task_of[pf];
        tkids[task_of[pf]].push_back(t);
      }
    }
    std::vector<double> work(ntask, 0.0);
    for (int t = 0; t < ntask; ++t) {   // tasks are created in postorder of their head fronts: children first
      for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
        const double m = (double)front_dim(S.task_fronts[k]), np = (double)S.f_ns[S.task_fronts[k]] * bs;
        work[t] += np * m * m;
      }
      for (int c : tkids[t]) work[t] += work[c];
    }
    std::vector<int> cut;
    for (int t = 0; t < ntask; ++t)
      if (tparent[t] < 0) cut.push_back(t);
    // longest-processing-time deal of the current cut; returns max load / mean load
    std::vector<int> cut_rank;
    auto deal = [&]() {
      std::vector<int> order(cut.size());
      for (size_t k = 0; k < cut.size(); ++k) order[k] = (int)k;
      std::sort(order.begin(), order.end(), [&](int x, int y) { return work[cut[x]] != work[cut[y]] ? work[cut[x]] > work[cut[y]] : cut[x] < cut[y]; });
      std::vector<double> load(opt.world, 0.0);
      cut_rank.assign(cut.size(), 0);
      double total = 0.0;
      for (int k : order) {
        int r = 0;
        for (int q = 1; q < opt.world; ++q)
          if (load[q] < load[r]) r = q;
        load[r] += work[cut[k]];
        cut_rank[k] = r;
        total += work[cut[k]];
      }
      return *std::max_element(load.begin(), load.end()) / std::max(total / opt.world, 1e-300);
    };
    for (;;) {
      const bool enough = (int)cut.size() >= opt.world;
      if (enough && ((int)cut.size() >= 8 * opt.world || deal() <= 1.10)) break;
      int best = -1;
      for (size_t k = 0; k < cut.size(); ++k)
        if (!tkids[cut[k]].empty() && (best < 0 || work[cut[k]] > work[cut[best]])) best = (int)k;
      if (best < 0) break;
      const int t = cut[best];
      S.task_owner[t] = -1;   // shared
      cut.erase(cut.begin() + best);
      for (int c : tkids[t]) cut.push_back(c);
    }
    deal();
    for (size_t k = 0; k < cut.size(); ++k) {
      const int t = cut[k];
      std::vector<int> stk(1, t);
      while (!stk.empty()) {
        int u = stk.back();
        stk.pop_back();
        S.task_owner[u] = cut_rank[k];
        for (int c : tkids[u]) stk.push_back(c);
      }
      if (tparent[t] >= 0) S.xroots.push_back(t);   // its update matrix / vector feeds a shared task
    }
    std::sort(S.xroots.begin(), S.xroots.end());
    for (int t = 0; t < ntask; ++t)
      if (S.task_owner[t] == -2) throw StateFailure("partition: unassigned task");
  }
  S.pose_owner.assign(nb, opt.rank);
  for (int t = 0; t < ntask; ++t)
    for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
      const int f = S.task_fronts[k];
      for (int j = S.sn_start[f]; j < S.sn_start[f + 1]; ++j) S.pose_owner[S.perm[j]] = S.task_owner[t];
    }
  S.block_consumer.resize(nnzb);
  for (int q = 0; q < nnzb; ++q) S.block_consumer[q] = S.task_owner[task_of[ent_front[q]]];
  // --- launch lists (task ids) per phase: [0] this rank's tasks, [1] shared tasks; inside a level the
  // LDS-class tasks come first, then the scratch-slab (single large front) tasks
  S.level_fronts.clear();
  std::vector<long long> scratch_off;
  long long scratch_max = 0;
  for (int ph = 0; ph < 2; ++ph) {
    launches_[ph].assign(nlev, LevelLaunch());
    std::vector<std::vector<int>> lds(nlev), glb(nlev);
    for (int t = 0; t < ntask; ++t) {
      const bool in_phase = ph == 0 ? (S.task_owner[t] == opt.rank) : (S.task_owner[t] == -1);
      if (!in_phase) continue;
      (is_lds(S.task_fronts[S.task_ptr[t]]) ? lds : glb)[task_level[t]].push_back(t);
    }
    for (int l = 0; l < nlev; ++l) {
      LevelLaunch& LL = launches_[ph][l];
      LL.lds_begin = (int)S.level_fronts.size();
      LL.lds_count = (int)lds[l].size();
      // levels whose fronts all fit the register-resident wave kernel (wave_front.inc): <= kWvNPV pivot columns,
      // <= 16 kWvNTL boundary rows, <= kFwdChildren children
      LL.wv = opt.wave_kernel != 0 && glb[l].empty() && !lds[l].empty();
      for (int t : lds[l]) {
        if (!LL.wv) break;
        for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1] && LL.wv; ++k) {
          const int f = S.task_fronts[k];
          const int npv = S.f_ns[f] * bs, nbr = S.f_nb[f] * bs;
          if (npv > kWvNPV || nbr > 16 * kWvNTL || S.child_off[f + 1] - S.child_off[f] > kFwdChildren) LL.wv = false;
          if (S.asm_off[f + 1] - S.asm_off[f] > 64) LL.wv = false;   // (one table entry per lane)
          for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
            const int nbc = S.f_nb[S.children[ch]];
            if (nbc * (nbc + 1) / 2 > 64) LL.wv = false;
          }
        }
      }
      // wide launches run two waves per front (see front_factor_kernel); *_max_m = packed doubles of the largest front
      if (!LL.wv && opt.wave_front_tasks > 0 && LL.lds_count >= opt.wave_front_tasks) LL.sm_count = LL.lds_count;
      for (int i = 0; i < (int)lds[l].size(); ++i) {
        const int t = lds[l][i];
        S.level_fronts.push_back(t);
        scratch_off.push_back(0);
        int& mx = i < LL.sm_count ? LL.sm_max_m : LL.lds_max_m;
        for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
          const int nbt = S.f_ns[S.task_fronts[k]] + S.f_nb[S.task_fronts[k]];
          mx = std::max(mx, nbt * (nbt + 1) / 2 * bs * bs);
        }
      }
      LL.glb_begin = (int)S.level_fronts.size();
      LL.glb_count = (int)glb[l].size();
      long long so = 0;
      for (int t : glb[l]) {
        scratch_off.push_back(so);
        const long long m = (long long)front_dim(S.task_fronts[S.task_ptr[t]]);
        so += m * m;
        S.level_fronts.push_back(t);
        LL.glb_max_m = std::max(LL.glb_max_m, (int)m);
      }
      scratch_max = std::max(scratch_max, so);
      LL.glb_scratch = so;
      for (int q = LL.lds_begin; q < LL.glb_begin + LL.glb_count; ++q) {
        const int t = S.level_fronts[q];
        for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
          const int f = S.task_fronts[k];
          const int m = (int)front_dim(f);
          LL.max_m = std::max(LL.max_m, m);
          LL.max_panel = std::max(LL.max_panel, m * S.f_ns[f] * bs + S.f_ns[f] * bs);
        }
      }
    }
  }
  if (getenv("G2OHIP_PLAN_DUMP")) {   // per level: tasks, fronts per task, histogram of the largest front (blocks) per task
    for (int l = 0; l < nlev; ++l) {
      const LevelLaunch& LL = launches_[0][l];
      std::vector<int> hist(32, 0);
      long long nfr = 0, npiv = 0, sumtri = 0;
      for (int q = LL.lds_begin; q < LL.lds_begin + LL.lds_count; ++q) {
        const int t = S.level_fronts[q];
        int mx = 0;
        for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
          const int f = S.task_fronts[k], nbt = S.f_ns[f] + S.f_nb[f];
          mx = std::max(mx, nbt);
          ++nfr;
          npiv += S.f_ns[f];
          sumtri += nbt * (nbt + 1) / 2;
        }
        ++hist[std::min(mx, 31)];
      }
      fprintf(stderr, "level %d: tasks %d fronts %lld pivots %lld blocks %lld | max-front hist:", l, LL.lds_count, nfr, npiv, sumtri);
      for (int i = 0; i < 32; ++i)
        if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]);
      fprintf(stderr, "\n");
    }
  }
  // --- scratch-slab fronts: where each one lives, and CHAINS factorised in place.
  // A large supernode is cut into panels of <= max_sn_scalars pivot columns: a chain of fronts, each the only child of
  // the next, the parent's rows being exactly the child's boundary rows.  Such a parent is factorised IN PLACE in the
  // trailing part of its child's frontal matrix: the child's rank-npiv update writes there instead of a packed update
  // matrix, the parent adds its original blocks, and neither the zero fill, nor the extend-add, nor the O(m^2) update
  // matrix per panel exist (a dense m-row supernode used to cost m^3 / (6 * 48) doubles of update matrices).
  // Every front that is not such a parent owns a region of the slab for good (no reuse across levels: a chain keeps
  // its region over several levels).
  std::vector<int> scratch_ld(scratch_off.size(), 0), inpl_prev(nf, -1), inpl_next(nf, -1);
  {
    std::vector<int> slot_of(nf, -1), lvl_of(nf, -1), ph_of(nf, -1);
    auto level_big = [&](const LevelLaunch& LL) {
      if (LL.glb_count <= 0 || !opt.big_front_passes || LL.glb_max_m < opt.big_front_min_dim) return false;
      for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
        const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
        if (S.f_ns[f] * bs > 64 || S.child_off[f + 1] - S.child_off[f] > 16) return false;
      }
      return true;
    };
    std::vector<std::vector<char>> bigl(2, std::vector<char>(nlev, 0));
    for (int ph = 0; ph < 2; ++ph)
      for (int l = 0; l < nlev; ++l) {
        const LevelLaunch& LL = launches_[ph][l];
        bigl[ph][l] = level_big(LL) ? 1 : 0;
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          slot_of[f] = q;
          lvl_of[f] = l;
          ph_of[f] = ph;
        }
      }
    if (opt.inplace_chains && opt.world == 1)
      for (int f = 0; f < nf; ++f) {
        if (slot_of[f] < 0 || S.child_off[f + 1] - S.child_off[f] != 1) continue;
        const int c = S.children[S.child_off[f]];
        if (slot_of[c] < 0 || ph_of[c] != ph_of[f] || lvl_of[c] + 1 != lvl_of[f]) continue;
        if (!bigl[ph_of[f]][lvl_of[f]] || !bigl[ph_of[c]][lvl_of[c]]) continue;
        if (S.f_ns[f] + S.f_nb[f] != S.f_nb[c]) continue;
        bool ident = true;
        for (int k = 0; k < S.f_nb[c] && ident; ++k) ident = S.rel[S.rel_off[c] + k] == k;
        if (!ident) continue;
        inpl_prev[f] = c;
        inpl_next[c] = f;
      }
    long long total = 0;
    for (int ph = 0; ph < 2; ++ph)
      for (int l = 0; l < nlev; ++l) {   // (levels ascending: a child's region is placed before its in-place parent)
        LevelLaunch& LL = launches_[ph][l];
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          const long long m = (long long)front_dim(f);
          if (inpl_prev[f] >= 0) {
            const int c = inpl_prev[f], qc = slot_of[c];
            scratch_ld[q] = scratch_ld[qc];
            scratch_off[q] = scratch_off[qc] + (long long)S.f_ns[c] * bs * (scratch_ld[qc] + 1);
          } else {
            scratch_ld[q] = (int)m;
            scratch_off[q] = total;
            total += m * m;
          }
        }
      }
    scratch_max = std::max<long long>(total, 1);
    if (getenv("G2OHIP_PLAN_DUMP")) {
      int nglb = 0, nmem = 0, single = 0, rows_ne = 0, not_ident = 0;
      for (int f = 0; f < nf; ++f) {
        if (slot_of[f] < 0) continue;
        ++nglb;
        if (inpl_prev[f] >= 0) { ++nmem; continue; }
        if (S.child_off[f + 1] - S.child_off[f] != 1) continue;
        ++single;
        const int c = S.children[S.child_off[f]];
        if (S.f_ns[f] + S.f_nb[f] != S.f_nb[c]) ++rows_ne; else ++not_ident;
      }
      fprintf(stderr, "scratch-slab fronts %d, continued in place %d; single-child but not in place %d (rows differ %d, other %d); slab %.1f MB\n", nglb, nmem,
              single, rows_ne, not_ident, total * 8e-6);
    }
    // update matrices: none for a front whose parent continues in place
    bool any = false;
    for (int f = 0; f < nf; ++f) any = any || inpl_next[f] >= 0;
    if (any) {
      S.U_total = 0;
      for (int f = 0; f < nf; ++f) {
        S.U_off[f] = S.U_total;
        if (inpl_next[f] < 0) S.U_total += (long long)S.f_nb[f] * (S.f_nb[f] + 1) / 2 * bs * bs;
      }
      stats_.bytes_U = (size_t)S.U_total * 8;
    }
  }
  if (S.level_fronts.empty()) {
    S.level_fronts.push_back(0);
    scratch_off.push_back(0);
    scratch_ld.push_back(0);
  }
  // --- packed per-front records and per-parent extend-add descriptors
  std::vector<FrontRec> recs(nf);
  std::vector<ChildDesc> cdesc(S.children.size());
  std::vector<int> crel, cmap;
  crel.reserve(S.rel.size());
  const int T_ = (bs % 3 == 0) ? 3 : bs;
  int tri_max = 1;
  for (int f = 0; f < nf; ++f) {
    FrontRec& R = recs[f];
    std::memset(&R, 0, sizeof(R));
    R.ns = S.f_ns[f];
    R.nb = S.f_nb[f];
    R.c0 = S.sn_start[f];
    R.asm_off = S.asm_off[f];
    R.asm_cnt = S.asm_off[f + 1] - S.asm_off[f];
    R.child_off = S.child_off[f];
    R.child_cnt = S.child_off[f + 1] - S.child_off[f];
    R.crel_off = (int)crel.size();
    R.cmap_off = (int)cmap.size();
    R.L_off = S.L_off[f];
    R.pad[0] = 0;   // (set below once the children are known) third child fits the three-children fast path
    R.rows_off = S.rows_off[f];
    if (S.w_off[f] > 0x7fffffffLL) throw StateFailure("symbolic: solve workspace exceeds 2^31 doubles");
    R.w_off = (int)S.w_off[f];
    R.U_off = S.U_off[f];
    for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
      int c = S.children[ch];
      cdesc[ch].U_off = S.U_off[c];
      cdesc[ch].nbc = S.f_nb[c];
      cdesc[ch].crel_start = (int)crel.size() - R.crel_off;
      cdesc[ch].cmap_start = (int)cmap.size() - R.cmap_off;
      cdesc[ch].w_off = (int)S.w_off[c];
      const int* rl = S.rel.data() + S.rel_off[c];
      crel.insert(crel.end(), rl, rl + S.f_nb[c]);
      for (int ib = 0; ib < S.f_nb[c]; ++ib)
        for (int jb = 0; jb <= ib; ++jb) {
          if (rl[ib] >= (1 << 16) || rl[jb] >= (1 << 15)) throw StateFailure("symbolic: front too large for the packed child map");
          cmap.push_back(rl[ib] | (rl[jb] << 16));   // packed block (ib,jb), row-major lower order
        }
      if (ch - S.child_off[f] < 2) R.ch[ch - S.child_off[f]] = cdesc[ch];
    }
    if (R.child_cnt == 3) {   // third child within the three-children fast path of the factor kernel (8 doubles x 256 threads)?
      const int nbc2 = cdesc[R.child_off + 2].nbc;
      R.pad[0] = (nbc2 * (nbc2 + 1) / 2 * bs * bs <= 8 * kFactorThreads) ? 1 : 0;
    }
    R.crel_cnt = (int)crel.size() - R.crel_off;
    R.cmap_cnt = (int)cmap.size() - R.cmap_off;
    const int nt0 = (R.nb + R.ns - 1) * bs / T_;          // trailing tiles per side after the first pivot block
    R.tri_cnt = std::max(nt0 * (nt0 + 1) / 2, R.nb * (R.nb + 1) / 2);
    tri_max = std::max(tri_max, R.tri_cnt);
  }
  std::vector<int> tri(tri_max);
  {
    int k = 0;
    for (int i = 0; k < tri_max; ++i)
      for (int j = 0; j <= i && k < tri_max; ++j) tri[k++] = i | (j << 16);
  }
  for (int ph = 0; ph < 2; ++ph)
  for (LevelLaunch& LL : launches_[ph]) {
    LL.lds_idx_ints = LL.glb_idx_ints = LL.sm_idx_ints = 0;
    LL.lds_vec_m = 0;
    LL.lds_max_panel = 0;
    for (int q = LL.lds_begin; q < LL.glb_begin + LL.glb_count; ++q) {
      const int t = S.level_fronts[q];
      for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
        const FrontRec& R = recs[S.task_fronts[k]];
        if (q < LL.glb_begin) LL.lds_vec_m = std::max(LL.lds_vec_m, (R.ns + R.nb) * bs);
        if (q < LL.glb_begin) LL.lds_max_panel = std::max(LL.lds_max_panel, (R.ns + R.nb) * bs * R.ns * bs + R.ns * bs);
        if (q < LL.glb_begin) {   // may the factor kernel carry the forward sweep of this launch?
          const int nthr = LL.sm_count > 0 ? 128 : kFactorThreads;
          // (any number of children, any boundary size: the fifth and later children and those with more boundary rows
          // than threads are added by a loop; one right-hand side value per thread in the pivot part)
          const bool ok = R.ns * bs <= nthr && (opt.fuse_fwd_any || [&] {
            bool o = R.child_cnt <= kFwdChildren;
            for (int c = 0; c < R.child_cnt && o; ++c) o = cdesc[R.child_off + c].nbc * bs <= nthr;
            return o;
          }());
          if (!ok) LL.fuse_fwd = false;
        }
        if (q < LL.glb_begin) LL.wv_idx_ints = std::max(LL.wv_idx_ints, (2 + kVirtInts) * R.asm_cnt + R.cmap_cnt + R.crel_cnt);
        if (q < LL.lds_begin + LL.sm_count) LL.sm_idx_ints = std::max(LL.sm_idx_ints, (2 + kVirtInts) * R.asm_cnt + R.cmap_cnt + R.tri_cnt + R.crel_cnt);
        else if (q < LL.glb_begin) LL.lds_idx_ints = std::max(LL.lds_idx_ints, (2 + kVirtInts) * R.asm_cnt + R.cmap_cnt + R.tri_cnt + R.crel_cnt);
        else LL.glb_idx_ints = std::max(LL.glb_idx_ints, (2 + kVirtInts) * R.asm_cnt);
      }
    }
  }
  // --- grouped in-place chains.  The panels of one large supernode (a chain continued in place) each made a pass over the
  // whole trailing matrix: (m - k)^2 / 2 doubles read and written per 48 columns -- 460 GB for a 20 000-row front, which is
  // what its 250 ms were.  Panels are grouped by big_group: a panel inside a group updates only the columns of the group's
  // remaining panels (what their pivot blocks and panel rows need), the group's LAST panel updates everything behind the group
  // with all the group's pivot columns at once (they are adjacent columns of the same frontal matrix).
  std::vector<int> grp_prev(nf, 0), grp_rem(nf, 0), gtab, gtab_off(nf, -1);
  if (opt.big_group > 1)
    for (int f0 = 0; f0 < nf; ++f0) {
      if (inpl_prev[f0] >= 0 || inpl_next[f0] < 0) continue;   // (chain heads only)
      if (front_dim(f0) < opt.big_group_min_rows) continue;
      std::vector<int> chain;
      for (int f = f0; f >= 0; f = inpl_next[f]) chain.push_back(f);
      for (size_t g0 = 0; g0 < chain.size(); g0 += (size_t)opt.big_group) {
        const size_t g1 = std::min(chain.size(), g0 + (size_t)opt.big_group);
        int total = 0;
        for (size_t i = g0; i < g1; ++i) total += S.f_ns[chain[i]] * bs;
        if (total >= (1 << 15)) continue;
        int before = 0;
        for (size_t i = g0; i < g1; ++i) {
          const int f = chain[i], np_ = S.f_ns[f] * bs;
          if (i + 1 == g1) grp_prev[f] = before;                 // the group's last panel: all of the group's columns
          else grp_rem[f] = total - before - np_;                // inside the group: up to the group's end
          before += np_;
        }
        // the group's last panel reads the earlier panels' solved rows from THEIR L panels (not from the frontal matrix: the fused
        // solve + update kernel leaves the raw rows there): per earlier panel (L offset low / high, rows of its front, pivot columns,
        // row of its trailing part that is row 0 of the last panel's trailing part)
        if (g1 - g0 > 1) {
          const int fl = chain[g1 - 1];
          gtab_off[fl] = (int)gtab.size();
          gtab.push_back((int)(g1 - g0 - 1));
          for (size_t i = g0; i + 1 < g1; ++i) {
            const int f = chain[i];
            int rowoff = 0;
            for (size_t k = i + 1; k < g1; ++k) rowoff += S.f_ns[chain[k]] * bs;
            gtab.push_back((int)(unsigned int)(S.L_off[f] & 0xffffffffLL));
            gtab.push_back((int)(S.L_off[f] >> 32));
            gtab.push_back((int)front_dim(f));
            gtab.push_back(S.f_ns[f] * bs);
            gtab.push_back(rowoff);
          }
        }
      }
    }
  // --- trailing-update tiles of the scratch-slab fronts (big_front_update_kernel), per level launch
  std::vector<int> cinv;
  std::vector<int2> cinv_slot(S.level_fronts.size(), make_int2(-1, 0));   // per launch slot: (offset into cinv, ints) of the front's gather table
  {
    std::vector<int4> bt;
    int sw_max = 0;
    for (int ph = 0; ph < 2; ++ph)
      for (LevelLaunch& LL : launches_[ph]) {
        LL.bt_begin = (int)bt.size();
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          const int nt64 = (S.f_nb[f] * bs + 63) / 64;
          // w: bit 0 = update in place (the parent continues in this frontal matrix); grouped chains (grp_prev / grp_rem, below):
          // bits 1..15 = pivot columns of the group's earlier panels that ride along (the group's last panel), bits 16..31 = the
          // update stops at this column of the trailing matrix (a panel inside a group: the columns of the group's remaining panels)
          const int w = (inpl_next[f] >= 0 ? 1 : 0) | (grp_prev[f] << 1) | (grp_rem[f] << 16);
          if (grp_prev[f] > 0) LL.grouped = true;
          if (grp_rem[f] > 0) LL.group_in = true;
          const int ntc = grp_rem[f] > 0 ? std::min(nt64, (grp_rem[f] + 63) / 64) : nt64;
          for (int ti = 0; ti < nt64; ++ti)
            for (int tj = 0; tj <= std::min(ti, ntc - 1); ++tj) bt.push_back(make_int4(q, ti, tj, w));
        }
        LL.bt_count = (int)bt.size() - LL.bt_begin;
      }
    // --- fronts whose region is WRITTEN by the extend-add (big_extend_gather_kernel<.., true>: the children's entries or zero for every
    // lower block) instead of zero-filled, read and written: the heads of levels that run the one-launch extend-add as a whole-GPU
    // pass of its own.  Their original blocks -- and those of the fronts that continue in place behind them -- are added behind that
    // launch (LevelLaunch::la_*) instead of with everybody else's at the start of the phase.
    std::vector<char> write_head(nf, 0), late(nf, 0);
    std::vector<int> slot_of_f(nf, -1);
    for (int ph = 0; ph < 2; ++ph)
      for (LevelLaunch& LL : launches_[ph]) {
        int max_children = 0;
        bool ok = LL.glb_count > 0 && opt.big_front_passes && LL.glb_max_m >= opt.big_front_min_dim && opt.world == 1;
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          slot_of_f[f] = q;
          if (S.f_ns[f] * bs > 64) ok = false;
          max_children = std::max(max_children, S.child_off[f + 1] - S.child_off[f]);
        }
        LL.eg_write = ok && max_children >= 1 && max_children <= 7 && LL.bt_count > merge_tiles_of(LL);
        if (!LL.eg_write) continue;
        int heads = 0;
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          if (S.child_off[f + 1] == S.child_off[f] || inpl_prev[f] >= 0) continue;
          write_head[f] = 1;
          ++heads;
          for (int g = f; g >= 0; g = inpl_next[g]) late[g] = 1;
        }
        if (heads == 0) LL.eg_write = false;   // (a level of fronts continued in place: nothing to extend-add)
      }
    for (int ph = 0; ph < 2; ++ph)
      for (LevelLaunch& LL : launches_[ph]) {
        // zero-fill chunks of the fronts that start a region at this level
        LL.fz_begin = (int)bt.size();
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          if (inpl_prev[f] >= 0 || write_head[f]) continue;
          // (the lower block triangle only -- nothing uses what lies above a diagonal block: a chunk = a few columns from the first row of
          // their diagonal block down, ~kFillChunk doubles; the full squares were 5.4 GB = 1 ms per iteration of the 10 000-camera grid graph)
          const int m = (int)front_dim(f);
          for (int c0 = 0; c0 < m;) {
            const int row0 = (c0 / bs) * bs, h = m - row0;
            const int nc = std::max(1, std::min(m - c0, kFillChunk / h));
            bt.push_back(make_int4(q, c0, nc, row0));
            c0 += nc;
          }
        }
        LL.fz_count = (int)bt.size() - LL.fz_begin;
        // assembly chunks (32 original blocks each)
        LL.ba_begin = (int)bt.size();
        LL.big_ok = LL.glb_count > 0;
        int max_children = 0;
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          if (S.f_ns[f] * bs > 64) LL.big_ok = false;
          const int na = S.asm_off[f + 1] - S.asm_off[f];
          for (int e = 0; e < na && !late[f]; e += 32) bt.push_back(make_int4(q, e, std::min(32, na - e), inpl_prev[f] >= 0 ? 1 : 0));   // w: add to what is there
          max_children = std::max(max_children, S.child_off[f + 1] - S.child_off[f]);
        }
        LL.ba_count = (int)bt.size() - LL.ba_begin;
        if (max_children > 16) LL.big_ok = false;   // (one launch per child ordinal)
        // inverse block maps for the gather at load time (big_level_kernel): front block -> child's boundary block
        LL.gather = LL.big_ok && max_children <= kGatherChildren;
        // ... and for the extend-add of a level in ONE launch (big_extend_gather_kernel: a workgroup owns blocks of the parent and adds
        // its children's entries in child order -- one read-modify-write of the frontal matrix instead of one per child ordinal); the
        // header has room for seven children
        LL.eg_ok = LL.big_ok && (max_children >= 2 || LL.eg_write) && max_children <= 7;   // (one child per front: a pass per ordinal is one launch too -- unless it writes)
        LL.eg_begin = (int)bt.size();
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count && (LL.gather || LL.eg_ok); ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          const int nch = S.child_off[f + 1] - S.child_off[f], mb = S.f_ns[f] + S.f_nb[f];
          if (nch == 0 || inpl_prev[f] >= 0) continue;
          if (kGatherHeader + nch * mb > kGatherInts) LL.gather = false;   // (the merged level launch stages the table of a front in LDS)
          if (!LL.gather && !LL.eg_ok) break;
          // table: [0] children, [1 + 2 c], [2 + 2 c] offset of child c's update matrix (low, high word), [kGatherHeader + c mb + b] the maps
          const size_t t0 = cinv.size();
          cinv.resize(t0 + kGatherHeader + (size_t)nch * mb, -1);
          cinv[t0] = nch;
          for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
            const int c = S.children[ch], k0 = ch - S.child_off[f];
            cinv[t0 + 1 + 2 * k0] = (int)(unsigned int)(S.U_off[c] & 0xffffffffLL);
            cinv[t0 + 2 + 2 * k0] = (int)(S.U_off[c] >> 32);
            const int* rl = S.rel.data() + S.rel_off[c];
            for (int k = 0; k < S.f_nb[c]; ++k) cinv[t0 + kGatherHeader + (size_t)k0 * mb + rl[k]] = k;
          }
          cinv_slot[q] = make_int2((int)t0, kGatherHeader + nch * mb);
          if (LL.eg_ok) {   // chunks of lower blocks of ONE block column of the parent (consecutive rows: the stores of a chunk are contiguous per column),
                            // 256 scalar rows each: one row per thread (more per workgroup was measured slower: what the kernel lives on is requests in flight)
            const int cb = std::max(1, std::min(64, kEgThreads / bs));
            for (int jb = 0; jb < mb; ++jb)
              for (int ib = jb; ib < mb; ib += cb) bt.push_back(make_int4(q, jb, ib, std::min(cb, mb - ib)));
          }
        }
        LL.eg_count = LL.eg_ok ? (int)bt.size() - LL.eg_begin : 0;
        LL.eg_maxc = max_children;
        // original blocks added behind the writing extend-add: the heads of this level and the fronts continued in place behind them
        LL.la_begin = (int)bt.size();
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count && LL.eg_write; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          if (!write_head[f]) continue;
          for (int g = f; g >= 0; g = inpl_next[g]) {
            const int na = S.asm_off[g + 1] - S.asm_off[g];
            for (int e = 0; e < na; e += 32) bt.push_back(make_int4(slot_of_f[g], e, std::min(32, na - e), 1));
          }
        }
        LL.la_count = (int)bt.size() - LL.la_begin;
        if (LL.eg_write && (!LL.eg_ok || LL.eg_count == 0)) throw StateFailure("symbolic: a level marked for the writing extend-add has no gather tables");
        // extend-add passes: pass c handles child c of every front (the children of one front may hit the same blocks)
        LL.be_pass.clear();
        for (int c = 0; c < max_children && LL.big_ok; ++c) {
          const int b0 = (int)bt.size();
          for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
            const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
            if (S.child_off[f] + c >= S.child_off[f + 1] || inpl_prev[f] >= 0) continue;   // (in place: the child's update is there)
            const int nbc = S.f_nb[S.children[S.child_off[f] + c]];
            const int nblk = nbc * (nbc + 1) / 2;
            for (int b = 0; b < nblk; b += 64) bt.push_back(make_int4(q, c, b, std::min(64, nblk - b)));
          }
          LL.be_pass.emplace_back(b0, (int)bt.size() - b0);
        }
        // panel row chunks (256 rows below the pivot block each)
        LL.tr_begin = (int)bt.size();
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          const int rows = S.f_nb[f] * bs;
          for (int r = 0; r < rows; r += 256) bt.push_back(make_int4(q, r, 0, 0));
        }
        LL.tr_count = (int)bt.size() - LL.tr_begin;
        LL.tr_all = LL.glb_count > 0;   // every front has panel rows (big_panel_solve_kernel: a front without would have no workgroup)
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count; ++q)
          if (S.f_nb[S.task_fronts[S.task_ptr[S.level_fronts[q]]]] == 0) LL.tr_all = false;
        // row chunks for the multi-workgroup sweeps (big_forward_kernel / big_backward_kernel): the chunks of a front are
        // contiguous, w = ordinal | count << 16
        LL.sw_begin = (int)bt.size();
        bool sw_ok = LL.glb_count > 0;
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count && sw_ok; ++q) {
          const int t = S.level_fronts[q];
          if (S.task_ptr[t + 1] - S.task_ptr[t] != 1 || S.f_ns[S.task_fronts[S.task_ptr[t]]] * bs > 64) sw_ok = false;
        }
        for (int q = LL.glb_begin; q < LL.glb_begin + LL.glb_count && sw_ok; ++q) {
          const int f = S.task_fronts[S.task_ptr[S.level_fronts[q]]];
          const int rows = S.f_nb[f] * bs, G = std::max(1, (rows + 255) / 256);
          for (int g = 0; g < G; ++g) bt.push_back(make_int4(f, g * 256, std::max(0, std::min(256, rows - g * 256)), g | (G << 16)));
        }
        LL.sw_count = (int)bt.size() - LL.sw_begin;
        sw_max = std::max(sw_max, LL.sw_count);
      }
    // merged backward launches: maximal runs of consecutive levels with scratch-slab fronts (their LDS-class fronts, chains
    // included, ride along front by front: the kernel works per front, from the L panel in memory), top level first, a
    // front's chunks behind those of its parent.  A waiting workgroup never blocks the one it waits for: workgroups are
    // dispatched in order and the parent's come first.  Chunk word
    // w = ordinal | count << 8 | (wait for the parent's flag) << 16 | (raise the own flag) << 17.
    {
      constexpr int kMaxMerged = 1024;
      std::vector<int> fpar(nf, -1), fgroup(nf, -1);
      for (int f = 0; f < nf; ++f)
        for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) fpar[S.children[ch]] = f;
      int merged_max = 0;
      for (int ph = 0; ph < 2; ++ph) {
        bw_groups_[ph].clear();
        bw_of_level_[ph].assign(nlev, -1);
        int cur = -1;
        for (int l = nlev - 1; l >= 0; --l) {
          const LevelLaunch& LL = launches_[ph][l];
          std::vector<int4> lev;
          bool ok = LL.glb_count > 0 && opt.big_front_passes != 0;   // (levels of LDS fronts only: the dependency-driven launches)
          for (int q = LL.lds_begin; q < LL.glb_begin + LL.glb_count && ok; ++q) {
            const int t = S.level_fronts[q];
            for (int k = S.task_ptr[t + 1] - 1; k >= S.task_ptr[t] && ok; --k) {   // chain top first
              const int f = S.task_fronts[k];
              const int rows = S.f_nb[f] * bs, Gc = std::max(1, (rows + 255) / 256);
              if (S.f_ns[f] * bs > 64 || Gc > 255) ok = false;
              for (int g = 0; g < Gc && ok; ++g) lev.push_back(make_int4(f, g * 256, std::max(0, std::min(256, rows - g * 256)), g | (Gc << 8)));
            }
          }
          if (!ok || (int)lev.size() > kMaxMerged) { cur = -1; continue; }
          if (cur < 0 || bw_groups_[ph][cur].count + (int)lev.size() > kMaxMerged || bw_groups_[ph][cur].bottom_level != l + 1) {
            bw_groups_[ph].push_back(BwGroup{l, l, (int)bt.size(), 0});
            cur = (int)bw_groups_[ph].size() - 1;
          }
          BwGroup& G = bw_groups_[ph][cur];
          G.bottom_level = l;
          bw_of_level_[ph][l] = cur;
          for (int4 c : lev) {
            const int f = c.x;
            fgroup[f] = ph * 65536 + cur;
            const int par = fpar[f];
            const bool wait = par >= 0 && fgroup[par] == ph * 65536 + cur;
            c.w |= (wait ? 1 << 16 : 0) | (1 << 17);
            bt.push_back(c);
          }
          G.count = (int)bt.size() - G.begin;
          merged_max = std::max(merged_max, G.count);
        }
        // (a group of one level gains nothing: leave it to the per-level launch)
        for (size_t gi = 0; gi < bw_groups_[ph].size(); ++gi)
          if (bw_groups_[ph][gi].top_level == bw_groups_[ph][gi].bottom_level) bw_of_level_[ph][bw_groups_[ph][gi].top_level] = -1;
      }
      sw_max = std::max(sw_max, merged_max);
      d_sw_flag.alloc(((size_t)nf + 256) & ~(size_t)255);   // (a multiple of 1 KB: one fill kernel per zeroing)
    }
    // phase-wide copies of the fill and assembly chunks: the regions of the slab are never reused and the original
    // blocks do not depend on any child, so both passes can run once per phase instead of once per level
    for (int ph = 0; ph < 2; ++ph) {
      auto hoistable = [&](const LevelLaunch& LL) {
        return LL.big_ok && opt.big_front_passes && LL.glb_count > 0 && LL.glb_max_m >= opt.big_front_min_dim;
      };
      hz_begin_[ph] = (int)bt.size();
      for (LevelLaunch& LL : launches_[ph])
        if (hoistable(LL))
          for (int i = 0; i < LL.fz_count; ++i) { const int4 c = bt[LL.fz_begin + i]; bt.push_back(c); }
      hz_count_[ph] = (int)bt.size() - hz_begin_[ph];
      ha_begin_[ph] = (int)bt.size();
      for (LevelLaunch& LL : launches_[ph]) {
        LL.hoisted = hoistable(LL);
        if (LL.hoisted)
          for (int i = 0; i < LL.ba_count; ++i) { const int4 c = bt[LL.ba_begin + i]; bt.push_back(c); }
      }
      ha_count_[ph] = (int)bt.size() - ha_begin_[ph];
    }
    if (bt.empty()) bt.push_back(make_int4(0, 0, 0, 0));
    d_big_tiles.upload(bt, st);
    d_sw_part.alloc((size_t)std::max(sw_max, 1) * 64);
    d_sw_cnt.alloc((size_t)nf + 1);
    d_sw_cnt.zero(st);
  }
  // --- levels split over two streams: where the streams have to wait for each other (LevelLaunch::fork / join).  The side stream
  // runs the LDS fronts of the split levels in level order, the main stream everything else: a wait is needed only where a front
  // has a child on the OTHER stream that the last wait does not cover yet (a cross-stream wait costs ~10 us on the chain of
  // whole-GPU passes even when its event fired long ago: the queue drains at the barrier packet).
  for (int ph = 0; ph < 2; ++ph) {
    auto in_phase = [&](int t) { return ph == 0 ? (S.task_owner[t] == opt.rank) : (S.task_owner[t] == -1); };
    std::vector<char> on_side(ntask, 0);
    int fork_cover = -1, join_cover = 0, nsplit = 0, nfork = 0, njoin = 0;
    for (int l = 0; l < nlev; ++l) {
      LevelLaunch& LL = launches_[ph][l];
      LL.split_ok = LL.glb_count > 0 && LL.lds_count > 0 && LL.big_ok && opt.big_front_passes && LL.glb_max_m >= opt.big_front_min_dim;
      auto child_on = [&](int q0, int q1, bool side, int since) {
        for (int q = q0; q < q1; ++q) {
          const int t = S.level_fronts[q];
          for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) {
            const int f = S.task_fronts[k];
            for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
              const int ct = task_of[S.children[ch]];
              if (ct != t && in_phase(ct) && (on_side[ct] != 0) == side && task_level[ct] >= since) return true;
            }
          }
        }
        return false;
      };
      LL.join = child_on(LL.split_ok ? LL.glb_begin : LL.lds_begin, LL.glb_begin + LL.glb_count, true, join_cover);
      if (LL.join) join_cover = l;
      LL.fork = false;
      if (LL.split_ok) {
        LL.fork = fork_cover < 0 || child_on(LL.lds_begin, LL.lds_begin + LL.lds_count, false, fork_cover);
        if (LL.fork) fork_cover = l;
        for (int q = LL.lds_begin; q < LL.lds_begin + LL.lds_count; ++q) on_side[S.level_fronts[q]] = 1;
        ++nsplit;
      }
      nfork += LL.fork;
      njoin += LL.join;
    }
    if (getenv("G2OHIP_PLAN_DUMP") && nsplit > 0) fprintf(stderr, "phase %d: %d split levels, %d forks, %d joins\n", ph, nsplit, nfork, njoin);
  }
  // --- factorisation launch groups.  Consecutive levels with the same kernel variant (and nothing the fused
  // kernel cannot carry) may share one launch: workgroups are dispatched in blockIdx order and the slots are in
  // level order, so every child of a waiting parent is already running or done (no deadlock); the parent
  // prefetches its tables, then spins on a device-scope counter its children bump after a release fence.
  for (int f = 0; f < nf; ++f) recs[f].pad[1] = 0x00ffffff;
  for (int f = 0; f < nf; ++f)
    for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) recs[S.children[ch]].pad[1] = f & 0x00ffffff;
  std::vector<int> factor_order(S.level_fronts);   // launch slot -> task for the FACTOR launches (solve sweeps keep level order)
  for (int ph = 0; ph < 2; ++ph) {
    groups_[ph].clear();
    auto in_phase = [&](int t) { return ph == 0 ? (S.task_owner[t] == opt.rank) : (S.task_owner[t] == -1); };
    for (int l = 0; l < nlev;) {
      FactorGroup G{launches_[ph][l], l, l, false};
      int l1 = l + 1;
      const bool sm = G.LL.sm_count > 0, wv = G.LL.wv;
      auto plain = [](const LevelLaunch& X) { return X.glb_count == 0 && X.lds_count > 0 && X.fuse_fwd; };
      if (opt.dep_levels > 1 && nf < (1 << 24) && plain(G.LL)) {
        int end = G.LL.lds_begin + G.LL.lds_count;
        while (l1 < nlev && l1 - l < opt.dep_levels) {
          const LevelLaunch& N = launches_[ph][l1];
          if (!plain(N) || (N.sm_count > 0) != sm || N.wv != wv || N.lds_begin != end) break;
          end += N.lds_count;
          ++l1;
        }
        // children each task of the levels (l, l1) has to wait for (those inside the group)
        std::vector<std::pair<int, int>> waits;   // (front, count)
        std::vector<int> signals;
        bool ok = l1 - l > 1;
        for (int lev = l + 1; lev < l1 && ok; ++lev) {
          const LevelLaunch& N = launches_[ph][lev];
          for (int q = N.lds_begin; q < N.lds_begin + N.lds_count && ok; ++q) {
            const int t = S.level_fronts[q], f = S.task_fronts[S.task_ptr[t]];
            int cnt = 0;
            for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
              const int c = S.children[ch], ct = task_of[c];
              if (ct != t && in_phase(ct) && task_level[ct] >= l && task_level[ct] < l1) {
                ++cnt;
                signals.push_back(c);
              }
            }
            if (cnt > 127) ok = false;
            if (cnt > 0) waits.emplace_back(f, cnt);
          }
        }
        if (ok) {
          for (int lev = l + 1; lev < l1; ++lev) {
            const LevelLaunch& N = launches_[ph][lev];
            G.LL.lds_count += N.lds_count;
            if (sm) {
              G.LL.sm_count += N.sm_count;
              G.LL.sm_max_m = std::max(G.LL.sm_max_m, N.sm_max_m);
              G.LL.sm_idx_ints = std::max(G.LL.sm_idx_ints, N.sm_idx_ints);
            } else {
              G.LL.lds_max_m = std::max(G.LL.lds_max_m, N.lds_max_m);
              G.LL.lds_idx_ints = std::max(G.LL.lds_idx_ints, N.lds_idx_ints);
            }
            G.LL.max_m = std::max(G.LL.max_m, N.max_m);
            G.LL.join = G.LL.join || N.join;
            G.LL.lds_vec_m = std::max(G.LL.lds_vec_m, N.lds_vec_m);
            G.LL.max_panel = std::max(G.LL.max_panel, N.max_panel);
            G.LL.wv_pn = std::max(G.LL.wv_pn, N.wv_pn);
            G.LL.wv_idx_ints = std::max(G.LL.wv_idx_ints, N.wv_idx_ints);
          }
          G.LL.glb_begin = G.LL.lds_begin + G.LL.lds_count;
          for (auto& w : waits) recs[w.first].pad[1] |= w.second << 24;
          for (int c : signals) recs[c].pad[1] |= (int)0x80000000u;
          // the backward sweep runs the same groups top-down: the child task (its top front c) waits for the parent front
          for (auto& w : waits) recs[w.first].pad[0] |= w.second << 8;
          for (int c : signals) recs[c].pad[0] |= 2;
          G.last_level = l1 - 1;
          G.dep = true;
          // Launch order inside a wide group: level order would dispatch a parent only after EVERY task of the
          // level below (the last parents then wait a whole round for their children); a parent is placed
          // dep_delay slots (about the number of resident workgroups) behind its last child instead -- still
          // behind all its children, which is what the no-deadlock argument needs.
          if (sm && opt.dep_delay > 0) {
            const int b0 = G.LL.lds_begin, cnt = G.LL.lds_count;
            std::vector<long long> key(cnt);
            std::vector<int> pos_of_task(ntask, -1);
            for (int i = 0; i < cnt; ++i) pos_of_task[factor_order[b0 + i]] = i;
            for (int i = 0; i < cnt; ++i) {   // (level order: children before parents)
              const int t = factor_order[b0 + i], f = S.task_fronts[S.task_ptr[t]];
              long long k = -1;
              for (int ch = S.child_off[f]; ch < S.child_off[f + 1]; ++ch) {
                const int ct = task_of[S.children[ch]];
                if (ct != t && pos_of_task[ct] >= 0) k = std::max(k, key[pos_of_task[ct]]);
              }
              key[i] = k < 0 ? (long long)i : k + opt.dep_delay;
            }
            std::vector<int> ord(cnt);
            for (int i = 0; i < cnt; ++i) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
            std::vector<int> tmp(cnt);
            for (int i = 0; i < cnt; ++i) tmp[i] = factor_order[b0 + ord[i]];
            std::copy(tmp.begin(), tmp.end(), factor_order.begin() + b0);
          }
        } else {
          l1 = l + 1;
        }
      }
      if (getenv("G2OHIP_PLAN_DUMP"))
        fprintf(stderr, "phase %d group: levels %d..%d tasks %d dep %d sm %d wv %d (pn %d idx %d)\n", ph, G.first_level, G.last_level, G.LL.lds_count + G.LL.glb_count,
                (int)G.dep, G.LL.sm_count, (int)G.LL.wv, G.LL.wv_pn, G.LL.wv_idx_ints);
      groups_[ph].push_back(G);
      l = l1;
    }
  }
  // --- band chains (band_chain.inc): leaf tasks whose rows are a band of half-width <= 4 blocks plus one border of <= 4
  // blocks.  They go to the head of level 0 of their launch group and are factorised by the sliding-window kernel in a
  // launch of their own in front of the group's (the parents' dependency counters are bumped the same way).
  {
    std::vector<BandChainRec> brecs;
    std::vector<int> btab;
    band_ent_h_.clear();
    band_ent_asm_.clear();
    constexpr int kMaxFr = 48, kMaxBlk = 160, kMaxEnt = 4096;
    struct BandInfo {
      BandChainRec rec;
      std::vector<int> tab, ent_asm;
      std::vector<int4> ent;
    };
    int why_cnt[32] = {0};
    auto why = [&](int c) { ++why_cnt[c & 31]; return false; };
    auto try_band = [&](int t, BandInfo& out) -> bool {
      if (bs != 6 || !opt.band_kernel) return why(1);
      const int k0 = S.task_ptr[t], k1 = S.task_ptr[t + 1];
      const int nfr = k1 - k0;
      if (nfr < 1 || nfr > kMaxFr) return why(2);
      const int f0 = S.task_fronts[k0], fl = S.task_fronts[k1 - 1];
      if (S.child_off[f0 + 1] != S.child_off[f0]) return why(3);   // a leaf of the tree
      for (int k = k0 + 1; k < k1; ++k)
        if (S.task_fronts[k] != S.task_fronts[k - 1] + 1) return why(4);
      const int base = S.sn_start[f0], nblk = S.sn_start[fl + 1] - base;
      if (nblk < 1 || nblk > kMaxBlk) return why(5);
      // boundary rows of the last front: border = those the first front has already, the rest continues the band
      const int* Bl = S.rows.data() + S.rows_off[fl];
      const int nbl = S.f_nb[fl];
      const int* B0 = S.rows.data() + S.rows_off[f0];
      const int nb0 = S.f_nb[f0];
      std::vector<int> Sset, Rset;
      for (int k = 0; k < nbl; ++k) (std::binary_search(B0, B0 + nb0, Bl[k]) && nfr > 1 ? Sset : Rset).push_back(Bl[k]);
      if (nfr == 1) {   // one front: every boundary row may be band or border; the nearest four continue the band
        Sset.clear();
        Rset.assign(Bl, Bl + nbl);
      }
      // the staying band rows in the order they are met (first column that couples to them)
      {
        std::vector<std::pair<int, int>> key;
        for (int r : Rset) {
          int first = nblk;
          for (int j = base; j < base + nblk && first == nblk; ++j)
            if (std::binary_search(st_[j].begin(), st_[j].end(), r)) first = j - base;
          key.emplace_back(first, r);
        }
        std::sort(key.begin(), key.end());
        for (size_t k = 0; k < key.size(); ++k) Rset[k] = key[k].second;
        if (nfr == 1 && Rset.size() > 4) {   // the ones met first stay band rows, the others are the border
          Sset.assign(Rset.begin() + 4, Rset.end());
          Rset.resize(4);
          std::sort(Sset.begin(), Sset.end());
        }
      }
      if (Sset.size() > 4 || Rset.size() > 4) return why(6);
      auto cls = [&](int r, int& bandblk, int& borderk) -> bool {
        bandblk = borderk = -1;
        if (r >= base && r < base + nblk) {
          bandblk = r - base;
          return true;
        }
        for (size_t k = 0; k < Rset.size(); ++k)
          if (Rset[k] == r) {
            bandblk = nblk + (int)k;
            return true;
          }
        for (size_t k = 0; k < Sset.size(); ++k)
          if (Sset[k] == r) {
            borderk = (int)k;
            return true;
          }
        return false;
      };
      for (int j = base; j < base + nblk; ++j)
        for (int i : st_[j]) {
          int bb, bk;
          if (!cls(i, bb, bk)) return why(8);
          if (bb >= 0 && bb - (j - base) > 4) {
            static int shown = 0;
            if (getenv("G2OHIP_BAND_DEBUG") && shown++ < 3) {
              fprintf(stderr, "band reject: chain of %d blocks, col %d row class %d; pivots (original ids):", nblk, j - base, bb);
              for (int q = 0; q < nblk; ++q) fprintf(stderr, " %d", S.perm[base + q]);
              fprintf(stderr, " | R:");
              for (int r : Rset) fprintf(stderr, " %d", S.perm[r]);
              fprintf(stderr, " | S:");
              for (int r : Sset) fprintf(stderr, " %d", S.perm[r]);
              fprintf(stderr, "\n");
            }
            return why(9);   // beyond the window of its column
          }
        }
      BandChainRec& R = out.rec;
      std::memset(&R, 0, sizeof(R));
      R.f_first = f0;
      R.nfronts = nfr;
      R.nblk = nblk;
      R.nS = (int)Sset.size();
      R.nR = (int)Rset.size();
      R.c0 = base;
      out.tab.clear();
      out.ent.clear();
      out.ent_asm.clear();
      std::vector<int> colfront(nblk, 0);
      for (int fi = 0; fi < nfr; ++fi) {
        const int f = f0 + fi, pb = S.sn_start[f] - base, ns = S.f_ns[f];
        const long long Loff = S.L_off[f];
        int fr[kBandFrontInts];
        fr[0] = pb;
        fr[1] = ns * bs;
        fr[2] = (int)(unsigned int)(Loff & 0xffffffffLL);
        fr[3] = (int)(Loff >> 32);
        fr[4] = (ns + S.f_nb[f]) * bs;
        for (int d = 0; d < 8; ++d) {
          const int bb = pb + d;
          int prow = -1;
          if (bb < nblk) prow = base + bb;
          else if (bb - nblk < (int)Rset.size()) prow = Rset[bb - nblk];
          fr[5 + d] = prow >= 0 ? local_pos(f, prow) : -1;
          // the kernel addresses a row that is a pivot of the chain as (band position - first pivot of the front): the band rows
          // of a front have to be contiguous (no holes in the band)
          if (bb < nblk && fr[5 + d] >= 0 && fr[5 + d] != d) return why(17);
        }
        for (int k = 0; k < 4; ++k) {
          fr[13 + k] = k < (int)Sset.size() ? local_pos(f, Sset[k]) : -1;
          if (k < (int)Sset.size() && fr[13 + k] < 0) return why(10);   // (the border is carried by every front)
        }
        out.tab.insert(out.tab.end(), fr, fr + kBandFrontInts);
        for (int c = 0; c < ns; ++c) colfront[pb + c] = fi;
        for (int e = S.asm_off[f]; e < S.asm_off[f + 1]; ++e) {
          const int pos = S.asm_pos[e], lr = pos & 0x7fff, lc = (pos >> 15) & 0x7fff;
          int bb = -1, bk = -1;
          if (lr < ns) bb = pb + lr;
          else if (!cls(S.rows[S.rows_off[f] + lr - ns], bb, bk)) return why(11);
          const int C0 = (pb + lc) * bs;
          const int R0 = bb >= 0 ? bb * bs : (0x10000 | (bk * bs));
          if (bb >= 0 && (bb < pb + lc || bb - (pb + lc) > 4)) return why(12);
          out.ent.push_back(make_int4(S.asm_q[e], pos, 0, 0));
          out.ent.push_back(make_int4(0, 0, 0, R0));
          out.ent.push_back(make_int4(C0, 0, 0, 0));
          out.ent_asm.push_back(e);
        }
      }
      const int nent = (int)out.ent_asm.size();
      if (nent > kMaxEnt) return why(13);
      R.ntiles = ((nblk + (int)Rset.size()) * bs + 15) / 16;
      // table: front records | (16-byte aligned) one record per pivot block | tile -> first record
      while (out.tab.size() % 4) out.tab.push_back(0);
      R.pad[1] = (int)out.tab.size();
      for (int cb = 0; cb < nblk; ++cb) {
        const int fi = colfront[cb];
        int fr[5];
        for (int q = 0; q < 5; ++q) fr[q] = out.tab[(size_t)kBandFrontInts * fi + q];   // (copied: the vector grows below)
        // band rows of the front that are pivots of the chain: its own pivots and the contiguous run behind them
        int nbp = 0;
        while (nbp < 8 && fr[0] + nbp < nblk && out.tab[(size_t)kBandFrontInts * fi + 5 + nbp] == nbp) ++nbp;
        out.tab.push_back(fr[2]);
        out.tab.push_back(fr[3]);
        out.tab.push_back(fr[4]);
        out.tab.push_back(fr[0] | ((fr[1]) << 8) | (fi << 16) | (nbp << 24));
      }
      // per band tile the records of the blocks with rows (border blocks: columns) in it: (source offset, -, -, -), (flags, -,
      // first row relative to the tile | border row, first column relative to the window | to the tile)
      std::vector<std::vector<int>> lists(R.ntiles);
      for (int i = 0; i < nent; ++i) {
        const int R0 = out.ent[3 * i + 1].w, C0 = out.ent[3 * i + 2].x;
        const int a = (R0 & 0x10000) ? C0 : R0;   // border entries enter with their columns, band entries with their rows
        for (int j = a / 16; j <= (a + bs - 1) / 16; ++j) {
          if (j >= R.ntiles) return why(14);
          lists[j].push_back(i);
        }
      }
      R.pad[0] = (int)out.tab.size();
      int run = 0;
      std::vector<int4> recs2;
      std::vector<int> recs_asm;
      for (int j = 0; j < R.ntiles; ++j) {
        out.tab.push_back(run);
        if ((int)lists[j].size() > 32) return why(15);   // (kBandListCap)
        run += (int)lists[j].size();
        const int lo = 16 * j;
        for (int i : lists[j]) {
          const int4 e0 = out.ent[3 * i];
          const int R0 = out.ent[3 * i + 1].w, C0 = out.ent[3 * i + 2].x;
          const bool border = (R0 & 0x10000) != 0;
          const int flags = ((e0.y >> 30) & 1) | (border ? 4 : 0) | ((!border && R0 == C0) ? 8 : 0);
          if ((long long)e0.x * bs * bs > 0x7fffff00LL) return why(16);
          recs2.push_back(make_int4(e0.x * bs * bs, 0, 0, 0));
          recs2.push_back(make_int4(flags, 0, border ? (R0 & 0xffff) : R0 - lo, border ? C0 - lo : C0 - (lo - 32)));
          recs_asm.push_back(out.ent_asm[i]);
        }
      }
      out.tab.push_back(run);
      out.ent.swap(recs2);
      out.ent_asm.swap(recs_asm);
      R.nent = run;
      R.tab_n = (int)out.tab.size();
      const int nsl = S.f_ns[fl];
      for (size_t k = 0; k < Rset.size(); ++k) R.ublk |= (local_pos(fl, Rset[k]) - nsl) << (4 * (int)k);
      for (size_t k = 0; k < Sset.size(); ++k) R.ublk |= (local_pos(fl, Sset[k]) - nsl) << (4 * (4 + (int)k));
      return true;
    };
    int n_band = 0, n_leaf = 0;
    for (int ph = 0; ph < 2; ++ph)
      for (FactorGroup& G : groups_[ph]) {
        if (G.first_level != 0 || !G.LL.wv) continue;
        const LevelLaunch& L0 = launches_[ph][0];
        const int b0 = L0.lds_begin, cnt = L0.lds_count;
        std::vector<int> head, tail;
        std::vector<BandInfo> infos;
        for (int i = 0; i < cnt; ++i) {
          const int t = factor_order[b0 + i];
          BandInfo bi;
          ++n_leaf;
          if (try_band(t, bi)) {
            head.push_back(t);
            infos.push_back(std::move(bi));
          } else {
            tail.push_back(t);
          }
        }
        if (head.empty()) continue;
        std::copy(head.begin(), head.end(), factor_order.begin() + b0);
        std::copy(tail.begin(), tail.end(), factor_order.begin() + b0 + head.size());
        G.band_count = (int)head.size();
        G.band_rec0 = (int)brecs.size();
        for (BandInfo& bi : infos) {
          bi.rec.tab_off = (int)btab.size();
          bi.rec.ent0 = (int)band_ent_asm_.size();
          btab.insert(btab.end(), bi.tab.begin(), bi.tab.end());
          band_ent_h_.insert(band_ent_h_.end(), bi.ent.begin(), bi.ent.end());
          band_ent_asm_.insert(band_ent_asm_.end(), bi.ent_asm.begin(), bi.ent_asm.end());
          G.band_ent_cap = std::max(G.band_ent_cap, bi.rec.nent);
          G.band_tab_cap = std::max(G.band_tab_cap, bi.rec.tab_n);
          brecs.push_back(bi.rec);
        }
        n_band += G.band_count;
      }
    stats_.n_band = (size_t)n_band;
    stats_.nnzL_band = stats_.piv_band = 0;
    for (const BandChainRec& r : brecs)
      for (int f = r.f_first; f < r.f_first + r.nfronts; ++f) {
        const size_t m = (size_t)(S.f_ns[f] + S.f_nb[f]) * bs, np = (size_t)S.f_ns[f] * bs;
        stats_.nnzL_band += np * m - np * (np - 1) / 2;
        stats_.piv_band += np;
      }
    if (getenv("G2OHIP_PLAN_DUMP")) {
      fprintf(stderr, "band chains rejected by rule:");
      for (int c = 0; c < 32; ++c)
        if (why_cnt[c]) fprintf(stderr, " %d:%d", c, why_cnt[c]);
      fprintf(stderr, "\n");
    }
    if (getenv("G2OHIP_PLAN_DUMP")) fprintf(stderr, "band chains: %d of %d leaf tasks (entries %zu, table ints %zu)\n", n_band, n_leaf, band_ent_asm_.size(), btab.size());
    if (brecs.empty()) {
      BandChainRec z;
      std::memset(&z, 0, sizeof(z));
      brecs.push_back(z);
    }
    if (btab.empty()) btab.push_back(0);
    if (band_ent_h_.empty()) band_ent_h_.push_back(make_int4(0, 0, 0, 0));
    d_band_rec.upload(brecs, st);
    d_band_tab.upload(btab, st);
    d_band_ent.upload(band_ent_h_, st);
    plan_.band_rec = d_band_rec.p;
    plan_.band_tab = d_band_tab.p;
    plan_.band_ent = d_band_ent.p;
    plan_.band_entv = nullptr;
  }
  d_ready.alloc((size_t)std::max(nf, 1));
  d_ready.zero(st);
  // --- backward sweep of the tree levels by groups of fronts (tree_backward_kernel).  In a dependency-driven group the levels
  // above the leaf level whose fronts are all small (kTreePiv pivot columns, kTreeBnd boundary rows) are cut into groups top
  // down: a root and whole levels of descendants while they fit sixteen waves (the first group of a tree is made shallower so
  // that the groups below it are full: four levels of a binary tree); the fronts of the next level start groups of their own.
  // Groups are listed parents first (the launch order the no-deadlock argument needs).
  {
    std::vector<int4> grec;
    std::vector<int2> gfr;
    std::vector<int> grows(S.rows);
    if (grows.empty()) grows.push_back(0);
    for (int ph = 0; ph < 2; ++ph)
      for (FactorGroup& G : groups_[ph]) {
        G.tb_grp0 = (int)grec.size();
        G.tb_ngrp = G.tb_low = 0;
        if (!opt.tree_backward || !G.dep || !opt.dep_backward || G.LL.glb_count > 0) continue;
        auto small = [&](int f) { return S.f_ns[f] * bs <= kTreePiv && S.f_nb[f] * bs <= kTreeBnd && S.f_ns[f] > 0; };
        int lc = G.last_level + 1;
        for (int l = G.last_level; l > G.first_level; --l) {
          const LevelLaunch& N = launches_[ph][l];
          bool ok = true;
          for (int q = N.lds_begin; q < N.lds_begin + N.lds_count && ok; ++q) {
            const int t = S.level_fronts[q];
            for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) ok = ok && small(S.task_fronts[k]);
          }
          if (!ok) break;
          lc = l;
        }
        if (G.last_level + 1 - lc < 2) continue;
        std::vector<int> in_tree(nf, 0);
        for (int l = lc; l <= G.last_level; ++l) {
          const LevelLaunch& N = launches_[ph][l];
          for (int q = N.lds_begin; q < N.lds_begin + N.lds_count; ++q) {
            const int t = S.level_fronts[q];
            for (int k = S.task_ptr[t]; k < S.task_ptr[t + 1]; ++k) in_tree[S.task_fronts[k]] = 1;
          }
        }
        std::vector<std::vector<int>> kids(nf);
        std::vector<int> height(nf, 0), roots;
        for (int f2 = 0; f2 < nf; ++f2) {   // (fronts are numbered children first)
          if (!in_tree[f2]) continue;
          height[f2] = std::max(height[f2], 1);
          const int pf = S.f_parent[f2];
          if (pf >= 0 && in_tree[pf]) {
            kids[pf].push_back(f2);
            height[pf] = std::max(height[pf], height[f2] + 1);
          } else {
            roots.push_back(f2);
          }
        }
        // depth of the deepest full group: the largest d with 2^d - 1 <= sixteen fronts
        constexpr int kDepth = 4;
        std::vector<int> grp_of(nf, -1), pos_of(nf, 0);
        std::vector<int> queue(roots);
        for (size_t qi = 0; qi < queue.size(); ++qi) {
          const int r = queue[qi];
          const int gid = (int)grec.size();
          const int depth_cap = (height[r] - 1) % kDepth + 1;
          std::vector<int> level(1, r), next;
          const int first = (int)gfr.size();
          int cnt = 0, nl = 0;
          while (!level.empty()) {
            for (int f2 : level) {
              grp_of[f2] = gid;
              pos_of[f2] = cnt++;
              gfr.push_back(make_int2(f2, nl));
            }
            ++nl;
            next.clear();
            for (int f2 : level)
              for (int c : kids[f2]) next.push_back(c);
            if (next.empty()) break;
            if (nl >= depth_cap || cnt + (int)next.size() > kTreeWaves) {
              for (int c : next) queue.push_back(c);   // groups of their own
              break;
            }
            level.swap(next);
          }
          const int pf = S.f_parent[r];
          grec.push_back(make_int4(first, cnt, nl, (pf >= 0 && in_tree[pf]) ? pf : -1));
        }
        G.tb_ngrp = (int)grec.size() - G.tb_grp0;
        for (int l = G.first_level; l < lc; ++l) G.tb_low += launches_[ph][l].lds_count;
        // what a front releases: the groups rooted below it and the tasks of the per-task launch that wait for it; where the
        // boundary values of a front come from
        std::vector<int> rel(nf, 0);
        for (int f2 = 0; f2 < nf; ++f2) {
          const int pf = S.f_parent[f2];
          if (pf < 0 || !in_tree[pf]) continue;
          if (in_tree[f2]) {
            if (grp_of[f2] != grp_of[pf]) ++rel[pf];
          } else {
            const int t = task_of[f2];
            const bool top = S.task_fronts[S.task_ptr[t + 1] - 1] == f2;
            if (top && task_level[t] >= G.first_level && task_level[t] < lc && (recs[f2].pad[0] & 2)) ++rel[pf];
          }
        }
        for (int e = grec[G.tb_grp0].x; e < (int)gfr.size(); ++e) {
          const int f2 = gfr[e].x;
          if (rel[f2] > 0x7fffff) throw StateFailure("tree_backward: release count out of range");
          gfr[e].y |= rel[f2] << 8;
          for (int j = 0; j < S.f_nb[f2]; ++j) {
            const int r = S.rows[S.rows_off[f2] + j], o = sn_of[r];
            if (grp_of[o] == grp_of[f2]) grows[S.rows_off[f2] + j] = -1 - (pos_of[o] * kTreePiv + (r - S.sn_start[o]) * bs);
          }
        }
        if (getenv("G2OHIP_PLAN_DUMP")) {
          fprintf(stderr, "phase %d tree backward: levels %d..%d in %d groups (%d fronts), %d slots left to the per-task kernel\n", ph, lc, G.last_level,
                  G.tb_ngrp, (int)gfr.size() - grec[G.tb_grp0].x, G.tb_low);
          int mx = 0;
          long long tot = 0, cnt = 0;
          std::vector<int> hist(8, 0);
          for (int l = G.first_level; l < lc; ++l)
            for (int q = launches_[ph][l].lds_begin; q < launches_[ph][l].lds_begin + launches_[ph][l].lds_count; ++q) {
              const int t = S.level_fronts[q], n = S.task_ptr[t + 1] - S.task_ptr[t];
              mx = std::max(mx, n);
              tot += n;
              ++cnt;
              ++hist[std::min(7, n / 8)];
            }
          fprintf(stderr, "  tasks below: %lld, fronts per task avg %.1f max %d; by length /8:", cnt, cnt ? (double)tot / cnt : 0.0, mx);
          for (int v : hist) fprintf(stderr, " %d", v);
          fprintf(stderr, "\n");
        }
      }
    stats_.n_tree_groups = grec.size();
    if (grec.empty()) grec.push_back(make_int4(0, 0, 0, -1));
    if (gfr.empty()) gfr.push_back(make_int2(0, 0));
    d_tb_grec.upload(grec, st);
    d_tb_front.upload(gfr, st);
    d_tb_rows.upload(grows, st);
  }
  d_task_ptr.upload(S.task_ptr, st);
  d_task_fronts.upload(S.task_fronts, st);
  d_rec.upload(recs, st);
  d_cdesc.upload(cdesc, st);
  d_crel.upload(crel, st);
  if (cinv.empty()) cinv.push_back(-1);
  if (cinv_slot.empty()) cinv_slot.push_back(make_int2(-1, 0));
  if (gtab.empty()) gtab.push_back(0);
  d_gtab.upload(gtab, st);
  d_gtab_off.upload(gtab_off, st);
  d_cinv.upload(cinv, st);
  d_cinv_slot.upload(cinv_slot, st);
  d_cmap.upload(cmap, st);
  d_tri.upload(tri, st);
  // --- multi-GPU exchange plan: update matrices / vectors of the subtree roots, solution mask
  {
    std::vector<SegCopy> segs;
    long long xoff = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int t : S.xroots) {
        const int f = S.task_fronts[S.task_ptr[t + 1] - 1];
        SegCopy sc;
        sc.flags = (S.task_owner[t] == opt.rank ? 1 : 0) | (pass ? 2 : 0);
        sc.a = pass ? S.w_off[f] : S.U_off[f];
        sc.n = pass ? S.f_nb[f] * bs : S.f_nb[f] * (S.f_nb[f] + 1) / 2 * bs * bs;
        sc.b = xoff;
        xoff += sc.n;
        segs.push_back(sc);
      }
    n_xseg_ = (int)segs.size();
    xbuf_count_ = (size_t)xoff;
    if (n_xseg_ > 0) {
      d_xseg.upload(segs, st);
      d_xbuf.alloc(xbuf_count_);
    }
    std::vector<double> mask((size_t)nb * bs, 1.0);
    if (opt.world > 1)
      for (int j = 0; j < nb; ++j) {
        const int o = S.task_owner[task_of[sn_of[j]]];
        const double v = (o == opt.rank || (o == -1 && opt.rank == 0)) ? 1.0 : 0.0;
        for (int r = 0; r < bs; ++r) mask[(size_t)j * bs + r] = v;
      }
    d_xmask.upload(mask, st);
  }
  // --- upload
  std::vector<int> c0(S.sn_start.begin(), S.sn_start.end() - 1);
  d_f_ns.upload(S.f_ns, st);
  d_f_nb.upload(S.f_nb, st);
  d_f_c0.upload(c0, st);
  d_rows_off.upload(S.rows_off, st);
  d_rows.upload(S.rows, st);
  d_rel_off.upload(S.rel_off, st);
  d_rel.upload(S.rel, st);
  d_asm_off.upload(S.asm_off, st);
  d_asm_q.upload(S.asm_q, st);
  d_asm_pos.upload(S.asm_pos, st);
  d_child_off.upload(S.child_off, st);
  d_children.upload(S.children, st);
  d_level_fronts.upload(S.level_fronts, st);
  {
    std::vector<int2> slots(S.level_fronts.size());
    for (size_t q = 0; q < slots.size(); ++q) {
      const int t = S.level_fronts[q];
      const int a = S.task_ptr[t], b = S.task_ptr[t + 1];
      for (int k = a + 1; k < b; ++k)
        if (S.task_fronts[k] != S.task_fronts[k - 1] + 1) throw StateFailure("symbolic: chain fronts are not consecutive");
      slots[q] = make_int2(a < b ? S.task_fronts[a] : 0, b - a);
    }
    d_slots.upload(slots, st);
    for (size_t q = 0; q < slots.size(); ++q) {
      const int t = factor_order[q];
      const int a = S.task_ptr[t], b = S.task_ptr[t + 1];
      slots[q] = make_int2(a < b ? S.task_fronts[a] : 0, b - a);
    }
    d_fslots.upload(slots, st);
    // backward sweep of a dependency-driven group: the same slots in reverse (parents before children)
    std::vector<int2> rslots(slots.size());
    for (size_t q = 0; q < slots.size(); ++q) {
      const int t = S.level_fronts[slots.size() - 1 - q];
      const int a = S.task_ptr[t], b = S.task_ptr[t + 1];
      rslots[q] = make_int2(a < b ? S.task_fronts[a] : 0, b - a);
    }
    d_bslots.upload(rslots, st);
    n_slots_ = (int)slots.size();
  }
  d_perm.upload(S.perm, st);
  {
    std::vector<int> ip(S.perm.size(), 0);
    for (size_t k = 0; k < S.perm.size(); ++k) ip[S.perm[k]] = (int)k;
    if (ip.empty()) ip.push_back(0);
    d_iperm.upload(ip, st);
  }
  d_L_off.upload(S.L_off, st);
  d_U_off.upload(S.U_off, st);
  d_w_off.upload(S.w_off, st);
  d_scratch_off.upload(scratch_off, st);
  d_scratch_ld.upload(scratch_ld, st);
  d_L.alloc((size_t)S.L_total);
  d_L.zero(st);   // (band_chain.inc never writes the structural zeros of a panel)
  d_U.alloc((size_t)S.U_total);
  d_w.alloc((size_t)S.w_total);
  d_y.alloc((size_t)nb * bs);
  d_xp.alloc((size_t)nb * bs);
  d_scratch.alloc((size_t)scratch_max);
  d_status.alloc(1);
  d_status.zero(st);
  if (!host_only) G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
  plan_.task_ptr = d_task_ptr.p;
  plan_.task_fronts = d_task_fronts.p;
  plan_.rec = d_rec.p;
  plan_.cdesc = d_cdesc.p;
  plan_.crel = d_crel.p;
  plan_.gtab = d_gtab.p;
  plan_.gtab_off = d_gtab_off.p;
  plan_.cinv = d_cinv.p;
  plan_.cinv_slot = d_cinv_slot.p;
  plan_.cmap = d_cmap.p;
  plan_.tri = d_tri.p;
  plan_.f_ns = d_f_ns.p;
  plan_.f_nb = d_f_nb.p;
  plan_.f_c0 = d_f_c0.p;
  plan_.rows_off = d_rows_off.p;
  plan_.rows = d_rows.p;
  plan_.rel_off = d_rel_off.p;
  plan_.rel = d_rel.p;
  plan_.L_off = d_L_off.p;
  plan_.U_off = d_U_off.p;
  plan_.w_off = d_w_off.p;
  plan_.asm_off = d_asm_off.p;
  plan_.asm_q = d_asm_q.p;
  plan_.asm_pos = d_asm_pos.p;
  plan_.child_off = d_child_off.p;
  plan_.children = d_children.p;
  plan_.L = d_L.p;
  plan_.U = d_U.p;
  plan_.w = d_w.p;
  plan_.status = d_status.p;
  plan_.ready = d_ready.p;
  plan_.dep_spin_limit = opt.dep_spin_limit;
  plan_.lds_mfma = opt.lds_mfma;
  plan_.dbg = nullptr;
  plan_.tl = nullptr;
  plan_.slots = d_slots.p;
  spinv_planned_ = false;   // (the inverse fronts follow the new tree)
  analyzed_ = !host_only;
  if (!host_only) prepare_kernels();
  stats_.t_symbolic = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// =====================================================================================
// Device kernels
// =====================================================================================
namespace {


// sqrt(d) and 1/sqrt(d) together: v_rsq_f64 seed + two Goldschmidt steps + one Newton correction
// (the same recipe LLVM uses for f64 sqrt, without the fdiv a separate rsqrt would need).  ~1-2 ulp.
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double& s, double& r) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double dd = fma(-g, g, d);
  g = fma(dd, h, g);
  s = g;
  r = h + h;
}

// Front record through the constant address space: wave-uniform and never written by a kernel => scalar
// (SMEM) loads instead of a vector load per lane.
// Update matrices / vectors travel between workgroups that may run on different XCDs INSIDE one launch
// (dependency-driven launches): they are written and read with agent-scope accesses (sc1: coherent at the
// device level) instead of write-back/invalidate fences over the whole L2.
// Ordering argument of the hand-off (no release / acquire fence is used, deliberately -- a fence writes back and
// invalidates the whole L2 of the XCD per task):
//   producer: its st_coh stores are agent-scope (sc1) accesses: they complete in the device-coherent level of the memory
//             hierarchy; s_waitcnt vmcnt(0) returns when every one of them has been acknowledged there; then the
//             workgroup barrier; only then ONE lane increments the consumer's counter (agent-scope atomic): the increment
//             is issued after the acknowledgements in program order, so it cannot become visible before the data;
//   consumer: one lane polls the counter with agent-scope atomic loads; the workgroup barrier after the poll orders
//             every wave's ld_coh loads behind it (they are issued after the barrier, a control dependence the hardware
//             does not speculate across), and ld_coh reads the same device-coherent level, never a stale L1 / L2 line.
// This relies on gfx9's in-order issue and completion accounting of a wave's vector memory operations (vmcnt) and on sc1
// accesses being coherent at agent scope; it does not rely on workgroup dispatch order beyond "a child of a task is
// running or finished when the task starts" (deadlock freedom).
// Dependency counters of the grouped factor launches: relaxed agent-scope atomics.  The data a counter guards is written with sc1
// stores that have been acknowledged (s_waitcnt vmcnt(0)) before the increment and read with sc1 loads after the poll, so no cache
// has to be written back or invalidated.  (The textbook release increment / acquire poll -- an L2 write-back and an invalidate per
// task -- was kept as an A/B option through round 5: the same bits, slower; removed with its option in round 6.)
__device__ __forceinline__ void dep_add(int* c, int v) { (void)__hip_atomic_fetch_add(c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int dep_poll(const int* c) { return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_coh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ FrontRec load_front_rec(const FrontRec* p) {
  FrontRec rec;
  typedef int __attribute__((may_alias)) alias_int;
  const __attribute__((address_space(4))) int* rp = (const __attribute__((address_space(4))) int*)reinterpret_cast<uintptr_t>(p);
  alias_int* ri = reinterpret_cast<alias_int*>(&rec);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(FrontRec) / sizeof(int)); ++i) ri[i] = rp[i];
  return rec;
}

__device__ __forceinline__ ChildDesc load_child_desc(const ChildDesc* p) {
  ChildDesc cd;
  typedef int __attribute__((may_alias)) alias_int;
  const __attribute__((address_space(4))) int* rp = (const __attribute__((address_space(4))) int*)reinterpret_cast<uintptr_t>(p);
  alias_int* ri = reinterpret_cast<alias_int*>(&cd);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(ChildDesc) / sizeof(int)); ++i) ri[i] = rp[i];
  return cd;
}

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// One workgroup factorises one TASK = a chain of frontal matrices f1 -> f2 -> ... in which every
// front is the only child of the next one; the update matrix travels from front to front in
// registers, only the last one of the chain is written to HBM.  A single-front task is the plain
// multifrontal step.
//   F: LDS-resident fronts are stored as packed lower-triangular BS x BS blocks (block (bi,bj), bi >= bj,
//   at ((bi(bi+1)/2 + bj) * BS*BS, column-major inside): half the LDS of a dense square, so twice the
//   fronts per CU.  Large fronts live in an HBM scratch slab as a dense m x m column-major matrix.
// Per front: stage index tables, zero, assemble original blocks, extend-add the children's update
// matrices, partial Cholesky of the ns pivot block columns (blocked by BS, look-ahead on the
// diagonal factor), write the L panel (+ reciprocal diagonal) and hand over / store the update matrix.
// Lower-triangular block/tile sets are enumerated ROW-major (idx = i(i+1)/2 + j): the enumeration
// of an n x n triangle is a prefix of that of any larger one, so one table (P.tri) serves every size.
// LDS: F | 2 diagonal-factor mailboxes | s_q[na] s_pos[na] s_cmap[cmap_cnt] s_tri[tri_cnt]
// NTC = workgroup size: 256, or 128 for the wide bottom levels (wave 0: look-ahead diagonal factor, wave 1:
// trailing update -- the factorisation of a small front is a latency chain, so 6 two-wave workgroups per
// CU beat 3 four-wave ones).
template <int BS, bool USE_LDS, int NTC = (USE_LDS ? kFactorThreads : kFactorThreadsGlobal), bool VIRT = false>
#ifndef G2OHIP_OCC256
#define G2OHIP_OCC256 2
#endif
#ifndef G2OHIP_OCC128
#define G2OHIP_OCC128 3
#endif
__global__ void __launch_bounds__(NTC, (USE_LDS && NTC <= 256) ? (NTC == 128 ? G2OHIP_OCC128 : G2OHIP_OCC256) : 1) front_factor_kernel(
    CholPlanDev P, int slot0, const double* __restrict__ A, double* __restrict__ scratch,
    const long long* __restrict__ scratch_off, int idx_off_doubles, int wcap, const double* __restrict__ bperm,
    double* __restrict__ yout, int dep) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = (BS % 3 == 0) ? 3 : BS;  // register tile edge of the trailing update
  // Fused forward sweep (LDS fronts, bperm != nullptr): the right-hand side rides along as ONE EXTRA ROW of the
  // front, kept in a small LDS vector tv[m].  The panel row solve turns its pivot part into y = L11^-1 b (one
  // more row for the thread that owns it), one lane per trailing column leaves  w = b2 - L21 y  in its boundary
  // part -- exactly cs_lsolve restricted to this front -- so the forward sweep costs no launch, no second
  // read of L and almost no extra work.
  // LDS: F | tv[wcap] | wprev[wcap] (update vector handed down a chain) | 2 diagonal-factor mailboxes | index tables
  const bool fwd = USE_LDS && bperm != nullptr;
  constexpr int BB = BS * BS;
  constexpr int UNR = 6;                     // independent global loads in flight per thread
  constexpr int LA = BS / T;                 // look-ahead span in tiles
  constexpr int NLA = LA * (LA + 1) / 2;
  // launch slot -> (first front, chain length): the fronts of a chain have consecutive ids (postorder)
  const int2 slot = P.slots[slot0 + blockIdx.x];
  const int f_first = __builtin_amdgcn_readfirstlane(slot.x), t1 = __builtin_amdgcn_readfirstlane(slot.y), t0 = 0;
  double* F = USE_LDS ? smem : (scratch + scratch_off[blockIdx.x]);
  double* sd = smem + idx_off_doubles - 2 * (BB + BS);   // two mailboxes: [L_kk (BB) | 1/diag (BS)]
  double* wprev = sd - wcap;
  double* tv = wprev - wcap;
  int* s_q = reinterpret_cast<int*>(smem + idx_off_doubles);
  const int tid = threadIdx.x;
  constexpr int NT = NTC;
  constexpr int kCarry = kChainU * kFactorThreads / NTC;   // same carry capacity for every workgroup size
  double ucarry[kCarry];    // update matrix of the previous chain front: packed element tid + u * NT
  int ncarry = 0;
#ifdef G2OHIP_CHOL_STAMPS
  int nstamp = 0;
#define STAMP() do { if (P.dbg && blockIdx.x == 0 && tid == 0 && nstamp < 60) P.dbg[4 + nstamp++] = wall_clock64(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();

  for (int ti = t0; ti < t1; ++ti) {
    const int f = f_first + ti;
    // wave-uniform and never written by a kernel: constant address space => scalar (SMEM) loads
    const FrontRec rec = load_front_rec(P.rec + f);
    const int ns = rec.ns, nbd = rec.nb;
    const int nbt = ns + nbd;
    const int m = nbt * BS, npiv = ns * BS, ld = m;
    const int cs = USE_LDS ? BS : ld;   // column stride inside a block
    auto blk_off = [&](int bi, int bj) { return USE_LDS ? (bi * (bi + 1) / 2 + bj) * BB : bi * BS + ld * (bj * BS); };
    const int nF = USE_LDS ? nbt * (nbt + 1) / 2 * BB : m * m;
    const int m_ext = fwd ? m + 1 : m;                  // rows taking part in the row solve (row m = tv)
    const int na = rec.asm_cnt;
    int* s_pos = s_q + na;
    // LDS-resident fronts stage the child maps and the triangle table; scratch-slab (large) fronts
    // read them from global memory (they can exceed the LDS)
    int* s_cmap_l = s_pos + na;
    int* s_tri_l = s_cmap_l + rec.cmap_cnt;
    const int* s_cmap = USE_LDS ? s_cmap_l : (P.cmap + rec.cmap_off);
    const int* s_tri = USE_LDS ? s_tri_l : P.tri;
    const int* s_crel = USE_LDS ? s_tri_l + rec.tri_cnt : (P.crel + rec.crel_off);
    const bool carried = ncarry > 0;            // the only child arrived through registers
    // extend-add of up to N register-held elements per thread, branch-free: every LDS gather is issued
    // before the first add (elements past the end go to a sink slot next to the mailboxes)
    const int sink = idx_off_doubles - 2 * (BB + BS);   // LDS fronts: L part of mailbox 0 is unused
    auto scatter_add = [&](const double* vals, auto N_, int count, int cmap_start) {
      constexpr int N = decltype(N_)::value;
      constexpr int CH = (N % 6 == 0) ? 6 : N;   // chunked: bounds the registers held at once
#pragma unroll
      for (int h = 0; h < N; h += CH) {
        int dd[CH];
        double cur[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int t = tid + (h + u) * NT;
          const bool ok = t < count;
          const int tt = ok ? t : 0;
          const int blk = tt / BB, e = tt - blk * BB;
          const int d = s_cmap[cmap_start + blk];
          dd[u] = ok ? blk_off(d & 0xffff, d >> 16) + e % BS + cs * (e / BS) : sink;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) cur[u] = F[dd[u]];
#pragma unroll
        for (int u = 0; u < CH; ++u) F[dd[u]] = cur[u] + vals[h + u];
      }
    };

    // ---- issue the children's update-matrix loads first (fast path: <= 2 children that fit one round)
    const int nch = rec.child_cnt;
    const int nU0 = nch > 0 ? rec.ch[0].nbc * (rec.ch[0].nbc + 1) / 2 * BB : 0;
    const int nU1 = nch > 1 ? rec.ch[1].nbc * (rec.ch[1].nbc + 1) / 2 * BB : 0;
#ifndef G2OHIP_UC
#define G2OHIP_UC 12
#endif
    constexpr int UC = USE_LDS ? G2OHIP_UC : UNR;      // child elements in flight per thread and child
    const bool fast_children = !carried && nch <= 2 && nU0 <= UC * NT && nU1 <= UC * NT;
    // (third child's size is checked on the host side of this condition: every LDS front has nbc <= 15 blocks)
    const bool three_fit = nU0 <= 8 * NT && nU1 <= 8 * NT && (rec.pad[0] & 1) != 0;
#ifdef G2OHIP_CHOL_STAMPS
    if (P.dbg && blockIdx.x == 0 && tid == 0) { P.dbg[1] = nch; P.dbg[2] = nU0 * 10000LL + nU1; P.dbg[3] = fast_children; }
#endif
    // ---- stage the index tables in LDS, zero the front meanwhile.  The four tables are contiguous in LDS
    // (q | pos | cmap | tri); all their loads are issued before the first wait: one memory round trip.
    {
      const int n1 = na, n2 = 2 * na, n3 = USE_LDS ? n2 + rec.cmap_cnt : n2, n5 = USE_LDS ? n3 + rec.tri_cnt : n2;
      const int n4 = USE_LDS ? n5 + rec.crel_cnt : n2;   // (q | pos | cmap | tri | crel | v)
      const int n6 = VIRT ? n4 + kVirtInts * na : n4;    // virtual source: (count, 3 partial slots, first list index) per entry
      const int* g_q = (VIRT ? P.asm_vq : P.asm_q) + rec.asm_off;
      const int* g_pos = (VIRT ? P.asm_vpos : P.asm_pos) + rec.asm_off - n1;
      const int* g_cmap = P.cmap + rec.cmap_off - n2;
      const int* g_tri = P.tri - n3;
      const int* g_crel = P.crel + rec.crel_off - n5;
      const int* g_v = P.asm_v + (size_t)kVirtInts * rec.asm_off - n4;
      constexpr int SU = 4;
      for (int base = tid; base < n6; base += SU * NT) {
        int v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = base + u * NT;
          const int* src = i < n1 ? g_q : (i < n2 ? g_pos : (i < n3 ? g_cmap : (i < n5 ? g_tri : (i < n4 ? g_crel : g_v))));
          v[u] = (i < n6) ? src[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = base + u * NT;
          if (i < n6) s_q[i] = v[u];
        }
      }
    }
    for (int i = tid; i < nF; i += NT) F[i] = 0.0;
    // dependency-driven launch: the children of this front run in THIS launch.  Everything above did not depend
    // on them; now one lane waits for their count (bounded: a broken plan flags status 2 instead of hanging).
    const int dep_wait = dep ? ((rec.pad[1] >> 24) & 0x7f) : 0;
    if (dep_wait > 0 && tid == 0) {
      int spins = 0;
      while (dep_poll(P.ready + f) < dep_wait) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > P.dep_spin_limit || ((spins & 255) == 0 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
          __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __hip_atomic_store(P.ready + f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next factorisation
    }
    __syncthreads();
    STAMP();
    // the children's update matrices share the memory round trip of the original entries (branch-free
    // clamped loads so that all of them are in flight together)
    double u0[UC];
    if (fast_children) {
      const double* U0 = P.U + rec.ch[0].U_off;
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const int t = tid + u * NT;
        u0[u] = ld_coh(U0 + (t < nU0 ? t : 0));
      }
    }
    // right-hand side and the children's update vectors (same round trip).  The host only fuses the forward
    // sweep into launches whose fronts have <= kFwdChildren children with boundaries of <= NT scalars and
    // <= NT pivot columns, so one value per thread and child suffices.
    constexpr bool EARLY_W = NTC != 128;   // two-wave variant: registers are scarce, the launch is wide -> load at use
    double bval = 0.0, wv[kFwdChildren];
    int wn[kFwdChildren], wcrel[kFwdChildren], woff[kFwdChildren];
    if (fwd) {
      if (tid < npiv) bval = bperm[(size_t)rec.c0 * BS + tid];
#pragma unroll
      for (int c = 0; c < kFwdChildren; ++c) {
        wn[c] = 0;
        wcrel[c] = 0;
        woff[c] = 0;
        wv[c] = 0.0;
        if (c < nch) {
          int nbc_c, crel_c, woff_c;
          if (c < 2) {
            nbc_c = rec.ch[c].nbc; crel_c = rec.ch[c].crel_start; woff_c = rec.ch[c].w_off;
          } else {
            const ChildDesc cd = load_child_desc(P.cdesc + rec.child_off + c);
            nbc_c = cd.nbc; crel_c = cd.crel_start; woff_c = cd.w_off;
          }
          wn[c] = nbc_c * BS;
          wcrel[c] = crel_c;
          woff[c] = woff_c;
          if (wn[c] > NT && !(carried && c == 0)) wn[c] = -wn[c];   // more boundary rows than threads: added by the loop in add_child_vec
          if (EARLY_W && tid < wn[c] && !(carried && c == 0)) wv[c] = ld_coh(P.w + woff_c + tid);
        }
      }
    }
    // ---- original entries (each block lands on a distinct tile)
    if constexpr (VIRT) {
      // The matrix is not materialised: block q = base[q] (+ lambda on the diagonal of diagonal blocks) minus its
      // partial blocks, subtracted in list order -- the Schur complement's reduction pass folded into this load
      // (same operations in the same order as the stand-alone reduction: identical bits).
      const int nA = na * BB;
      const int* s_v = s_q + (USE_LDS ? 2 * na + rec.cmap_cnt + rec.tri_cnt + rec.crel_cnt : 2 * na);
      const double lam0 = P.vlam[0];
      constexpr int UA = 4, HB = BS / 2;
      const bool vsplit = P.vsplit != 0;
      for (int base = tid; base < nA; base += UA * NT) {
        // a block has one or two partials almost always: the base block and two partials per element are requested
        // together (one round trip for UA elements); the rare third and later ones follow
        double a[UA], p0[UA], p1[UA];
        int dst[UA], nn[UA], sv3[UA], pe[UA];
        bool dl[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
          const int t = base + u * NT;
          dst[u] = -1;
          a[u] = p0[u] = p1[u] = 0.0;
          nn[u] = 0;
          sv3[u] = 0;
          pe[u] = 0;
          dl[u] = false;
          if (t < nA) {
            const int e = t / BB, rc = t - e * BB;
            const int r = rc % BS, c = rc / BS;
            const int q = s_q[e], pos = s_pos[e];
            const int lr = pos & 0x7fff, lc = (pos >> 15) & 0x7fff, tr = (pos >> 30) & 1;
            const int rr = tr ? c : r, cc = tr ? r : c;               // element of the stored (upper) block
            const int rp = (vsplit && rr >= HB) ? 1 : 0, nrp = vsplit ? HB : BS;
            pe[u] = rp * (nrp * BS) + (rr - rp * nrp) + nrp * cc;     // partial layout [row part][column][row in part]
            const int* sv = s_v + kVirtInts * e;
            nn[u] = sv[0];
            sv3[u] = kVirtInts * e;
            if (q >= 0) a[u] = P.vbase[(size_t)q * BB + rr + BS * cc];
            p0[u] = P.vparts[(size_t)sv[1] * BB + pe[u]];
            p1[u] = P.vparts[(size_t)sv[2] * BB + pe[u]];
            dl[u] = (pos < 0) && rr == cc;                            // (bit 31: diagonal block)
            dst[u] = blk_off(lr, lc) + r + cs * c;
          }
        }
#pragma unroll
        for (int u = 0; u < UA; ++u)
          if (dst[u] >= 0) {
            double v = a[u];
            if (dl[u]) v += lam0;
            if (nn[u] > 0) v -= p0[u];
            if (nn[u] > 1) v -= p1[u];
            if (nn[u] > 2) {
              const int* sv = s_v + sv3[u];
              v -= P.vparts[(size_t)sv[3] * BB + pe[u]];
              for (int k = sv[4] + 3, kend = sv[4] + nn[u]; k < kend; k += 8) {   // (long lists: eight partials per round trip, list order kept)
                int sl[8];
                double pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sl[j] = P.vslots[min(k + j, kend - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = P.vparts[(size_t)sl[j] * BB + pe[u]];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                  if (k + j < kend) v -= pv[j];
              }
            }
            F[dst[u]] = v;
          }
      }
    } else {
      {
        const int nA = na * BB;
        for (int base = tid; base < nA; base += UNR * NT) {
          double v[UNR];
          int dst[UNR];
  #pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int t = base + u * NT;
            dst[u] = -1;
            v[u] = 0.0;
            if (t < nA) {
              const int e = t / BB, rc = t - e * BB;
              const int r = rc % BS, c = rc / BS;
              const int q = s_q[e], pos = s_pos[e];
              const int lr = pos & 0x7fff, lc = (pos >> 15) & 0x7fff, tr = (pos >> 30) & 1;
              v[u] = tr ? A[(size_t)q * BB + c + BS * r] : A[(size_t)q * BB + r + BS * c];
              dst[u] = blk_off(lr, lc) + r + cs * c;
            }
          }
  #pragma unroll
          for (int u = 0; u < UNR; ++u)
            if (dst[u] >= 0) F[dst[u]] = v[u];
        }
      }
    }
    if (fwd) {
      for (int i = tid; i < m; i += NT) tv[i] = (i == tid) ? bval : 0.0;   // (npiv <= NT; bval = 0 beyond the pivot part)
    }
    __syncthreads();
    STAMP();
    // the children's update vectors are added in the same phases as their update matrices (one child at a
    // time: rows may coincide); returns true when the fast path handled them
    auto add_child_vec = [&](int ch) {
      if (!fwd) return;
      bool generic = ch >= kFwdChildren;
#pragma unroll
      for (int c = 0; c < kFwdChildren; ++c)   // (compile-time register index)
        if (c == ch) {
          if (carried && c == 0) {   // (the chain predecessor's vector, in LDS)
            for (int i = tid; i < wn[c]; i += NT) tv[s_crel[wcrel[c] + i / BS] * BS + i % BS] += wprev[i];
          } else if (wn[c] < 0) {
            generic = true;
          } else if (tid < wn[c]) {
            tv[s_crel[wcrel[c] + tid / BS] * BS + tid % BS] += EARLY_W ? wv[c] : ld_coh(P.w + woff[c] + tid);
          }
        }
      if (generic) {   // a fifth or later child, or one with more boundary rows than threads: straight from memory
        const ChildDesc cd = load_child_desc(P.cdesc + rec.child_off + ch);
        const int n = cd.nbc * BS;
        for (int i = tid; i < n; i += NT) tv[s_crel[cd.crel_start + i / BS] * BS + i % BS] += ld_coh(P.w + cd.w_off + i);
      }
    };
    // ---- extend-add of the children (sequential over children: destinations may overlap).  Update
    // matrices are packed lower-triangular blocks (row-major block order); cmap gives, per packed
    // block, the destination block (row | col << 16) in this front.
    if (carried) {
      add_child_vec(0);
      if (USE_LDS) {
        scatter_add(ucarry, std::integral_constant<int, kCarry>(), ncarry, rec.ch[0].cmap_start);
      } else {
#pragma unroll
        for (int u = 0; u < kCarry; ++u) {
          const int t = tid + u * NT;
          if (t < ncarry) {
            const int blk = t / BB, e = t - blk * BB;
            const int d = s_cmap[rec.ch[0].cmap_start + blk];
            F[blk_off(d & 0xffff, d >> 16) + e % BS + cs * (e / BS)] += ucarry[u];
          }
        }
      }
      __syncthreads();
    } else if (fast_children) {
      // the second child's loads are issued before the first child is added (registers: both are live
      // only here); the two children may hit the same destination blocks, hence the barrier between them
      // (two-wave variant: registers are scarce and the launch is wide -- one child at a time, reusing u0)
      constexpr bool DUAL = NTC != 128;
      double u1[DUAL ? UC : 1];
      if (DUAL && nch > 1) {
        const double* U1 = P.U + rec.ch[1].U_off;
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const int t = tid + u * NT;
          u1[DUAL ? u : 0] = ld_coh(U1 + (t < nU1 ? t : 0));
        }
      }
      if (nch > 0) add_child_vec(0);
      if (USE_LDS) {
        scatter_add(u0, std::integral_constant<int, UC>(), nU0, rec.ch[0].cmap_start);
        if (nch > 1) {
          if (!DUAL) {
            const double* U1 = P.U + rec.ch[1].U_off;
#pragma unroll
            for (int u = 0; u < UC; ++u) {
              const int t = tid + u * NT;
              u0[u] = ld_coh(U1 + (t < nU1 ? t : 0));
            }
          }
          __syncthreads();
          add_child_vec(1);
          scatter_add(DUAL ? u1 : u0, std::integral_constant<int, UC>(), nU1, rec.ch[1].cmap_start);
        }
      } else {
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const int t = tid + u * NT;
          if (t < nU0) {
            const int blk = t / BB, e = t - blk * BB;
            const int d = s_cmap[rec.ch[0].cmap_start + blk];
            F[blk_off(d & 0xffff, d >> 16) + e % BS + cs * (e / BS)] += u0[u];
          }
        }
        if (nch > 1) {
          __syncthreads();
#pragma unroll
          for (int u = 0; u < UC; ++u) {
            const int t = tid + u * NT;
            if (t < nU1) {
              const int blk = t / BB, e = t - blk * BB;
              const int d = s_cmap[rec.ch[1].cmap_start + blk];
              F[blk_off(d & 0xffff, d >> 16) + e % BS + cs * (e / BS)] += u1[u];
            }
          }
        }
      }
      __syncthreads();
    } else if (USE_LDS && NTC == 256 && !carried && nch == 3 && three_fit) {
      // three children (a separator front merged with one of its child separators): all their update matrices are
      // requested together -- one memory round trip instead of two per child on the generic path below
      constexpr int U3 = 8;
      const ChildDesc cd2 = load_child_desc(P.cdesc + rec.child_off + 2);
      const int nU2 = cd2.nbc * (cd2.nbc + 1) / 2 * BB;
      double v0[U3], v1[U3], v2[U3];
      {
        const double* U0 = P.U + rec.ch[0].U_off;
        const double* U1 = P.U + rec.ch[1].U_off;
        const double* U2 = P.U + cd2.U_off;
#pragma unroll
        for (int u = 0; u < U3; ++u) {
          const int t = tid + u * NT;
          v0[u] = ld_coh(U0 + (t < nU0 ? t : 0));
          v1[u] = ld_coh(U1 + (t < nU1 ? t : 0));
          v2[u] = ld_coh(U2 + (t < nU2 ? t : 0));
        }
      }
      add_child_vec(0);
      scatter_add(v0, std::integral_constant<int, U3>(), nU0, rec.ch[0].cmap_start);
      __syncthreads();
      add_child_vec(1);
      scatter_add(v1, std::integral_constant<int, U3>(), nU1, rec.ch[1].cmap_start);
      __syncthreads();
      add_child_vec(2);
      scatter_add(v2, std::integral_constant<int, U3>(), nU2, cd2.cmap_start);
      __syncthreads();
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        const ChildDesc cd = P.cdesc[rec.child_off + ch];  // wave-uniform
        add_child_vec(ch);
        const int* cmap = s_cmap + cd.cmap_start;
        const double* Uc = P.U + cd.U_off;
        const int nU = cd.nbc * (cd.nbc + 1) / 2 * BB;
        for (int base = tid; base < nU; base += UNR * NT) {
          double v[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int t = base + u * NT;
            v[u] = (t < nU) ? ld_coh(Uc + t) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int t = base + u * NT;
            if (t < nU) {
              const int blk = t / BB, e = t - blk * BB;
              const int d = cmap[blk];
              F[blk_off(d & 0xffff, d >> 16) + e % BS + cs * (e / BS)] += v[u];
            }
          }
        }
        __syncthreads();
      }
    }
    // ---- partial Cholesky, one pivot block (BS columns) per step.
    // The BS x BS diagonal factor is a long dependent chain (rsqrt, scale, update x BS); it is kept off
    // the critical path by look-ahead: while waves 1.. apply the trailing update of step kb, wave 0
    // updates only the NEXT diagonal block and factorises it into a small LDS mailbox (sd).
    STAMP();
    // upd: the block still lacks the update of pivot step kb_ - 1; it is applied here in registers from the
    // (already row-solved) block (kb_, kb_ - 1) -- look-ahead without a trip of the diagonal block through LDS
    auto diag_factor = [&](int kb_, double* box, auto upd_) {             // executed by every lane of wave 0
      constexpr bool upd = decltype(upd_)::value;
      double* Fd = F + blk_off(kb_, kb_);
      double Lk[BS][BS], inv[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c)
#pragma unroll
        for (int r = 0; r < BS; ++r) Lk[r][c] = (r >= c) ? Fd[r + cs * c] : 0.0;
      if constexpr (upd) {
        const double* Xp = F + blk_off(kb_, kb_ - 1);
        double X[BS][BS];
#pragma unroll
        for (int q = 0; q < BS; ++q)
#pragma unroll
          for (int r = 0; r < BS; ++r) X[r][q] = Xp[r + cs * q];
#pragma unroll
        for (int c = 0; c < BS; ++c)
#pragma unroll
          for (int r = c; r < BS; ++r) {
            double v = Lk[r][c];
#pragma unroll
            for (int q = 0; q < BS; ++q) v -= X[r][q] * X[c][q];
            Lk[r][c] = v;
          }
      }
      bool bad = false;
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        double d = Lk[c][c];
        if (!(d > 0.0)) {
          bad = true;
          d = 1.0;
        }
        double sq, r;
        sqrt_and_rsqrt(d, sq, r);
        inv[c] = r;
        Lk[c][c] = sq;
#pragma unroll
        for (int i = c + 1; i < BS; ++i) Lk[i][c] *= r;
#pragma unroll
        for (int j = c + 1; j < BS; ++j)
#pragma unroll
          for (int i = j; i < BS; ++i) Lk[i][j] -= Lk[i][c] * Lk[j][c];
      }
      // publish: the packed LDS layout keeps the diagonal block contiguous, so it is its own mailbox
      // (only 1/diag goes to the box); the dense HBM layout needs the LDS copy
      if (tid == 0) {
        if (bad) atomicMax(P.status, 1);   // (2 = a dependency wait gave up: must survive)
#pragma unroll
        for (int c = 0; c < BS; ++c) {
#pragma unroll
          for (int r = c; r < BS; ++r) {
            if (!USE_LDS) box[r + BS * c] = Lk[r][c];
            Fd[r + cs * c] = Lk[r][c];   // final L_kk (the part above the diagonal stays zero)
          }
          box[BB + c] = inv[c];
        }
      }
    };
    // LDS fronts with a boundary worth it: the pivot steps update the remaining PANEL columns only (as for the scratch-slab
    // fronts) and the trailing matrix gets ONE rank-npiv update on the matrix cores afterwards -- the trailing matrix is
    // read and written once per front instead of once per pivot block
    const bool mfma_tail = USE_LDS && P.lds_mfma != 0 && NT >= 64 && nbd * BS >= (P.lds_mfma & 0xffff) && ns >= (P.lds_mfma >> 16);
    if (ns > 0) {
      if (tid < 64) diag_factor(0, sd, std::false_type());
      __syncthreads();
    }
    STAMP();
#if defined(G2OHIP_ABL) && G2OHIP_ABL == 2
    for (int kb = 0; kb < 0; ++kb) {
#else
    for (int kb = 0; kb < ns; ++kb) {
#endif
      const int k0 = kb * BS;
      const double* box = sd + (kb & 1) * (BB + BS);
      const double* Lb = USE_LDS ? F + blk_off(kb, kb) : box;   // column stride BS either way
      double Lk[BS][BS], inv[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        inv[c] = box[BB + c];
#pragma unroll
        for (int r = 0; r < BS; ++r) Lk[r][c] = (r > c) ? Lb[r + BS * c] : 0.0;
      }
      // rows below the diagonal block: x * Lkk' = row
      for (int i = k0 + BS + tid; i < m_ext; i += NT) {   // (row m = the right-hand side, kept in tv)
        double x[BS];
        const bool rhs_row = i == m;
        double* Fr = rhs_row ? tv + k0 : F + blk_off(i / BS, kb) + i % BS;
        const int rs = rhs_row ? 1 : cs;
#pragma unroll
        for (int c = 0; c < BS; ++c) x[c] = Fr[rs * c];
#pragma unroll
        for (int c = 0; c < BS; ++c) {
          double v = x[c];
#pragma unroll
          for (int q = 0; q < c; ++q) v -= x[q] * Lk[c][q];
          x[c] = v * inv[c];
        }
#pragma unroll
        for (int c = 0; c < BS; ++c) Fr[rs * c] = x[c];
      }
      if (kb < 2) STAMP();
      __syncthreads();
      if (kb < 2) STAMP();
      // trailing update with T x T register tiles over the lower triangle (tile coordinates relative
      // to the first trailing row; the first LA(LA+1)/2 table entries are the next diagonal block)
      const int r0 = k0 + BS;
      const int nt = (m - r0) / T;
      const int ntiles = nt * (nt + 1) / 2;
      const bool lookahead = (kb + 1 < ns) && NT > 64;
      auto update_tile = [&](int packed) {
        const int ti = packed & 0xffff, tj = packed >> 16;   // tile coordinates relative to the first trailing row
        const int bi = kb + 1 + ti / LA, bj = kb + 1 + tj / LA, ia = (ti % LA) * T, ja = (tj % LA) * T;
        double* Fc = F + blk_off(bi, bj) + ia + cs * ja;
        const double* Fa = F + blk_off(bi, kb) + ia;
        const double* Fb = F + blk_off(bj, kb) + ja;
        double cv[T][T], av[BS][T], bv[BS][T];   // all operands first: one LDS latency per tile, not one per q
#pragma unroll
        for (int q = 0; q < BS; ++q) {
#pragma unroll
          for (int a = 0; a < T; ++a) av[q][a] = Fa[a + cs * q];
#pragma unroll
          for (int b = 0; b < T; ++b) bv[q][b] = Fb[b + cs * q];
        }
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
          for (int a = 0; a < T; ++a) cv[a][b] = Fc[a + cs * b];
#pragma unroll
        for (int q = 0; q < BS; ++q)
#pragma unroll
          for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) cv[a][b] -= av[q][a] * bv[q][b];
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
          for (int a = 0; a < T; ++a) Fc[a + cs * b] = cv[a][b];
      };
      if (lookahead && tid < 64) {
        if constexpr (NTC == 128) {   // (two-wave variant: no registers to spare for the in-register update)
          if (tid < NLA) update_tile(s_tri[tid]);
          __threadfence_block();   // the wave's own LDS writes are complete before it re-reads the block
          if (kb < 2) STAMP();
          diag_factor(kb + 1, sd + ((kb + 1) & 1) * (BB + BS), std::false_type());
        } else {
          if (kb < 2) STAMP();
          diag_factor(kb + 1, sd + ((kb + 1) & 1) * (BB + BS), std::true_type());
        }
        if (kb < 2) STAMP();
      } else {
        const int first = lookahead ? NLA + tid - 64 : tid, stride = lookahead ? NT - 64 : NT;
        if (fwd) {   // right-hand side: tv[j] -= L[j, k0..k0+BS) . y_kb, one lane per trailing column
          for (int j = r0 + (lookahead ? tid - 64 : tid); j < m; j += stride) {
            const double* Lj = F + blk_off(j / BS, kb) + j % BS;
            double v = tv[j];
#pragma unroll
            for (int q = 0; q < BS; ++q) v -= Lj[cs * q] * tv[k0 + q];
            tv[j] = v;
          }
        }
        if (USE_LDS && !mfma_tail) {
          for (int idx = first; idx < ntiles; idx += stride) update_tile(s_tri[idx]);
        } else {
          // scratch-slab (large) fronts: only the remaining PANEL columns are updated here; the rank-npiv update
          // of the trailing matrix is one pass of big_front_update_kernel over the whole GPU afterwards
          const int ntc = (npiv - r0) / T;
          const int total = nt * ntc;
          (void)ntiles;
          for (int idx = lookahead ? tid - 64 : tid; idx < total; idx += stride) {
            const int ti_ = idx / ntc, tj_ = idx - ti_ * ntc;
            if (tj_ > ti_ || (lookahead && ti_ < LA)) continue;   // (upper triangle; the next diagonal block: wave 0)
            update_tile(ti_ | (tj_ << 16));
          }
        }
      }
      __syncthreads();
      STAMP();
    }
    if constexpr (USE_LDS) {
      if (mfma_tail) {
        // F22 -= L21 L21' as 16 x 16 tiles of v_mfma_f64_16x16x4_f64 (K = npiv), operands read from the packed blocks in
        // LDS.  Result element D[lane/16 + 4v][lane%16]: rows from the first operand, columns from the second.
        const int mt = nbd * BS, nt16 = (mt + 15) >> 4, ks = (npiv + 3) >> 2;
        const int wave = tid >> 6, nw = NT >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
        // per lane and k-step: offset of panel column k inside a block row (blocks (bi, 0..ns-1) are contiguous), and
        // whether the column exists -- the same for every tile: computed once per front (first six k-steps in registers)
        int koff[6];
        bool kon[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int k = 4 * u + lk, kk = min(k, npiv - 1), kb_ = kk / BS;
          koff[u] = kb_ * BB + BS * (kk - kb_ * BS);
          kon[u] = k < npiv;
        }
        // result columns of this lane inside a tile column: c = 16 tj + lk + 4 v
        // a wave owns whole tile ROWS (ti = wave, wave + nw, ...: the row-side operand and its block row are per-row work)
        for (int ti = nt16 - 1 - wave; ti >= 0; ti -= nw) {   // (longest rows first)
          const int rr = min(ti * 16 + lr, mt - 1), rrb = rr / BS, rri = rr - rrb * BS;
          const double* Ar = F + blk_off(ns + rrb, 0) + rri;
          const bool row_ok = ti * 16 + lr < mt;
          double av[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) av[u] = Ar[koff[u]];
#pragma unroll
          for (int u = 0; u < 6; ++u) av[u] = kon[u] ? av[u] : 0.0;
          const int rowbase = blk_off(ns + rrb, ns) + rri;   // block (rrb, jb) of the trailing part: + jb * BB + BS * (c - jb BS)
          for (int tj = 0; tj <= ti; ++tj) {
            const int rc = min(tj * 16 + lr, mt - 1), rcb = rc / BS;
            const double* Ac = F + blk_off(ns + rcb, 0) + (rc - rcb * BS);
            double bv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) bv[u] = Ac[koff[u]];
            // destinations requested with the operands
            int dst[4];
            double cur[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int c0_ = tj * 16 + lk + 4 * v, c = min(c0_, mt - 1), jb = c / BS;
              const bool ok = row_ok && c0_ < mt && rrb >= jb;
              dst[v] = ok ? rowbase + jb * BB + BS * (c - jb * BS) : sink;
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) cur[v] = F[dst[v]];
            mfma_d4 acc = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kon[u] ? bv[u] : 0.0, av[u], acc, 0, 0, 0);
            for (int s0 = 6; s0 < ks; ++s0) {   // (supernodes of more than 24 pivot columns)
              const int k = 4 * s0 + lk, kk = min(k, npiv - 1), kb_ = kk / BS;
              const int off = kb_ * BB + BS * (kk - kb_ * BS);
              const double a_ = Ar[off], b_ = Ac[off];
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < npiv ? b_ : 0.0, k < npiv ? a_ : 0.0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) F[dst[v]] = cur[v] - acc[v];
          }
        }
        __syncthreads();
      }
    }
    // ---- forward-sweep results: y (pivot part) to HBM, w (boundary part) to the next chain front or to HBM
    if (fwd) {
      for (int i = tid; i < npiv; i += NT) yout[(size_t)rec.c0 * BS + i] = tv[i];
      if (ti + 1 < t1) {
        for (int i = tid; i < nbd * BS; i += NT) wprev[i] = tv[npiv + i];
      } else {
        double* wf = P.w + rec.w_off;
        for (int i = tid; i < nbd * BS; i += NT) st_coh(wf + i, tv[npiv + i]);
      }
    }
    // ---- write the L panel (m x npiv) and the reciprocals of its diagonal.  A task whose parent waits in this
    // launch hands over its update matrix FIRST and writes the panel afterwards (the parent does not read L).
    double* Lg = P.L + rec.L_off;
    auto write_panel = [&]() {
      if (USE_LDS) {   // m x npiv column-major panel out of the packed blocks (blocks above the diagonal are zero)
        const float invm = 1.0f / (float)m;
        for (int t = tid; t < m * npiv; t += NT) {
          const int c = (int)(((float)t + 0.5f) * invm), r = t - c * m;
          const int bi = r / BS, bj = c / BS;
          Lg[t] = (bi >= bj) ? F[blk_off(bi, bj) + r % BS + BS * (c % BS)] : 0.0;
        }
      } else {
        for (int t = tid; t < m * npiv; t += NT) Lg[t] = F[t];  // ld == m: identical layout
      }
      for (int k = tid; k < npiv; k += NT) Lg[(size_t)m * npiv + k] = 1.0 / F[blk_off(k / BS, k / BS) + (k % BS) * (1 + cs)];
    };
    const bool dep_signal = dep && rec.pad[1] < 0;   // (bit 31) the parent front waits in this launch
#if !defined(G2OHIP_ABL) || G2OHIP_ABL != 1   // (ablation builds, make EXTRA=-DG2OHIP_ABL=n: 1 = no L panel, 2 = no pivot loop)
    if (!dep_signal) write_panel();
#endif
    // ---- update matrix (packed lower-triangular blocks, row-major): to the next chain front through
    // registers, or to HBM for a parent in a later launch
    STAMP();
    const int nU = nbd * (nbd + 1) / 2 * BB;
    if (ti + 1 < t1) {
      ncarry = nU;
#pragma unroll
      for (int u = 0; u < kCarry; ++u) {
        const int t = tid + u * NT;
        if (t < nU) {
          const int blk = t / BB, e = t - blk * BB;
          const int d = s_tri[blk];
          ucarry[u] = F[blk_off(ns + (d & 0xffff), ns + (d >> 16)) + e % BS + cs * (e / BS)];
        }
      }
    } else {
      ncarry = 0;
#pragma unroll
      for (int u = 0; u < kCarry; ++u) ucarry[u] = 0.0;   // ends the live range: no registers held across the next front
      if constexpr (USE_LDS) {   // (scratch-slab fronts: big_front_update_kernel writes the update matrix)
        double* Ug = P.U + rec.U_off;
        for (int t = tid; t < nU; t += NT) {
          const int blk = t / BB, e = t - blk * BB;
          const int d = s_tri[blk];
          st_coh(Ug + t, F[blk_off(ns + (d & 0xffff), ns + (d >> 16)) + e % BS + cs * (e / BS)]);
        }
      }
    }
    if (dep_signal) {
      __builtin_amdgcn_s_waitcnt(0);   // every wave: its own (coherent) U / w stores have been acknowledged
      __syncthreads();
      if (tid == 0) dep_add(P.ready + (rec.pad[1] & 0x00ffffff), 1);
      write_panel();
    }
    __syncthreads();   // F and the LDS tables are reused by the next front of the chain
    STAMP();
#ifdef G2OHIP_CHOL_STAMPS
    if (P.dbg && blockIdx.x == 0 && tid == 0 && nstamp < 58) {
      P.dbg[4 + nstamp++] = -(long long)(ns * 1000 + m);   // front marker
    }
#endif
  }
#ifdef G2OHIP_CHOL_STAMPS
  if (P.dbg && blockIdx.x == 0 && tid == 0) P.dbg[0] = nstamp;
#endif
}

// ---- Scratch-slab (large) fronts as whole-GPU passes.  One workgroup per front cannot feed a front of several
// hundred rows (zeroing, extend-add and the trailing update are megabytes each); per level the work is cut into
//   [extend-add] -> pivot block (npiv <= 64 columns, one wave) -> panel rows -> rank-npiv update of the trailing matrix (MFMA)
// in one of three forms: ONE launch for pivot blocks + panel tiles that solve their rows themselves and gather the children's
// entries where they load (big_level_kernel; levels of at most 256 tiles: the latency chains of the pose graphs); pivot block + panel
// rows in one launch (big_panel_solve_kernel, the rows as chained MFMAs) + big_front_update_kernel (levels of one or a few large
// fronts); big_diag_mfma + big_trsm + big_front_update (levels of many fronts).  The extend-add of the last two: one launch
// (big_extend_gather_kernel) that WRITES the parent's region -- its children's entries in child order, or zero -- followed by the
// original blocks of the chain (big_assemble_kernel); regions nobody writes that way are zero-filled (big_fill_kernel) and get
// their original blocks at the start of the phase.  The dense front F (m x m, column-major, lower block triangle) lives in the
// scratch slab; a supernode of more than 64 columns is a chain of fronts factorised IN PLACE in the first one's region, its
// trailing updates grouped (big_group).  profiles/r6_grid_timeline.txt shows the launches of one iteration.
template <int BS, bool VIRT>
__global__ void __launch_bounds__(256) big_assemble_kernel(CholPlanDev P, const int4* __restrict__ chunks, const double* __restrict__ A,
                                                          double* __restrict__ scratch, const long long* __restrict__ scratch_off,
                                                          const int* __restrict__ scratch_ld) {
  constexpr int BB = BS * BS, HB = BS / 2;
  const int4 ck = chunks[blockIdx.x];   // x: launch slot, y: first original block of the chunk, z: blocks, w: add (front continued in place)
  const int f = P.slots[ck.x].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int m = scratch_ld[ck.x];   // leading dimension of the front in the slab
  double* F = scratch + scratch_off[ck.x];
  const double lam0 = VIRT ? P.vlam[0] : 0.0;
  const bool vsplit = VIRT && P.vsplit != 0;
  const int nel = ck.z * BB;
  for (int base = threadIdx.x; base < nel; base += 4 * 256) {   // four elements per thread: index loads, then value loads, together
    int q[4], pos[4], ee[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(base + u * 256, nel - 1);
      ee[u] = rec.asm_off + ck.y + t / BB;
      q[u] = (VIRT ? P.asm_vq : P.asm_q)[ee[u]];
      pos[u] = (VIRT ? P.asm_vpos : P.asm_pos)[ee[u]];
    }
    double v[4];
    size_t dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(base + u * 256, nel - 1);
      const int rc = t % BB, r = rc % BS, c = rc / BS;
      const int lr = pos[u] & 0x7fff, lc = (pos[u] >> 15) & 0x7fff, tr = (pos[u] >> 30) & 1;
      const int rr = tr ? c : r, cc = tr ? r : c;   // element of the stored (upper) block
      if constexpr (VIRT) {   // base (+ lambda) - partial blocks in list order (see front_factor_kernel)
        const int rp = (vsplit && rr >= HB) ? 1 : 0, nrp = vsplit ? HB : BS;
        const int pe = rp * (nrp * BS) + (rr - rp * nrp) + nrp * cc;
        const int* sv = P.asm_v + (size_t)kVirtInts * ee[u];
        const int n = sv[0];
        double x = q[u] >= 0 ? P.vbase[(size_t)q[u] * BB + rr + BS * cc] : 0.0;
        if (pos[u] < 0 && rr == cc) x += lam0;
        for (int k = 0; k < n && k < 3; ++k) x -= P.vparts[(size_t)sv[1 + k] * BB + pe];
        // long lists (a pose pair sharing landmarks of many tiles: graphs with loop closures): slot ids, then values, eight
        // at a time -- one dependent round trip per eight partials instead of two per partial; subtracted in list order
        for (int k = sv[4] + 3, kend = sv[4] + n; k < kend; k += 8) {
          int sl[8];
          double pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) sl[j] = P.vslots[min(k + j, kend - 1)];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = P.vparts[(size_t)sl[j] * BB + pe];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k + j < kend) x -= pv[j];
        }
        v[u] = x;
      } else {
        v[u] = A[(size_t)q[u] * BB + rr + BS * cc];
      }
      dst[u] = (size_t)(lr * BS + r) + (size_t)m * (lc * BS + c);
    }
    if (ck.w) {   // in place behind the child's update: add
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * 256 < nel) F[dst[u]] += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * 256 < nel) F[dst[u]] = v[u];
    }
  }
}

// zero the lower block triangle of the fronts that start a region at this level (x: launch slot, y: first column, z: columns, w: first row --
// the first row of the diagonal block of column y; a region's leading dimension is its front's size)
__global__ void __launch_bounds__(256) big_fill_kernel(const int4* __restrict__ chunks, double* __restrict__ scratch,
                                                      const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  const int4 ck = chunks[blockIdx.x];
  const int m = scratch_ld[ck.x], h = m - ck.w;
  double* p = scratch + scratch_off[ck.x] + (size_t)ck.y * m + ck.w;
  const int rx = threadIdx.x & 63;   // a wave per column, its lanes down the rows
  for (int c = threadIdx.x >> 6; c < ck.z; c += 4) {
    double* pc = p + (size_t)c * m;
#pragma unroll 4
    for (int r = rx; r < h; r += 64) pc[r] = 0.0;
  }
}

template <int BS>
__global__ void __launch_bounds__(256) big_extend_add_kernel(CholPlanDev P, const int4* __restrict__ chunks, double* __restrict__ scratch,
                                                            const long long* __restrict__ scratch_off) {
  constexpr int BB = BS * BS;
  const int4 ck = chunks[blockIdx.x];   // x: launch slot, y: child ordinal, z: first packed block, w: blocks
  const int f = P.slots[ck.x].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int m = (rec.ns + rec.nb) * BS;
  double* F = scratch + scratch_off[ck.x];
  const ChildDesc cd = P.cdesc[rec.child_off + ck.y];
  const double* Uc = P.U + cd.U_off;
  const int* cmap = P.cmap + rec.cmap_off + cd.cmap_start;
  const int n = ck.w * BB;
  for (int base = threadIdx.x; base < n; base += 3 * 256) {   // three elements per thread in flight
    double v[3], cur[3];
    size_t dst[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = min(base + u * 256, n - 1);
      const int blk = ck.z + t / BB, e = t % BB;
      const int d = cmap[blk];
      dst[u] = (size_t)((d & 0xffff) * BS + e % BS) + (size_t)m * ((d >> 16) * BS + e / BS);
      v[u] = Uc[(size_t)blk * BB + e];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) cur[u] = F[dst[u]];
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (base + u * 256 < n) F[dst[u]] = cur[u] + v[u];
  }
}

// The extend-add of a level in ONE launch: a workgroup owns 64 lower blocks of a parent front and adds the entries its children have
// there, child by child in child order (the sums of the passes above, bit for bit) -- the frontal matrix is read and written once
// instead of once per child ordinal, and a level with two or three children per front is one launch instead of two or three.
// chunks: up to 64 consecutive block rows of one block column (the stores of a chunk are contiguous column by column).  Maps: CholPlanDev::cinv.
// WRITE: the region has NOT been zero-filled -- every lower block of the parent is written, the sum of the children's entries or zero
// (fronts without a map table, i.e. without children, are not in the chunk list: their regions were filled); the original blocks follow.
template <int BS, bool WRITE, int MAXC>
__global__ void __launch_bounds__(256) big_extend_gather_kernel(CholPlanDev P, const int4* __restrict__ chunks, double* __restrict__ scratch,
                                                               const long long* __restrict__ scratch_off) {
  // MAXC: children per front of the level at most (2: the binary tree of a nested dissection -- the loops over the children are unrolled)
  constexpr int BB = BS * BS;
  __shared__ long long soff[MAXC][64];   // per child and block of the chunk: offset of the child's block in P.U, or -1
  const int4 ck = chunks[blockIdx.x];   // x: launch slot, y: block column, z: first block row, w: block rows (<= 64)
  const int f = P.slots[ck.x].x;
  const int2 cs = P.cinv_slot[ck.x];
  const int* tab = P.cinv + cs.x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int mb = rec.ns + rec.nb, m = mb * BS;
  double* F = scratch + scratch_off[ck.x];
  const int nch = min(tab[0], MAXC);
  const int jb = ck.y, cnt = ck.w;
  if (threadIdx.x < 64) {
    const int ib = ck.z + min((int)threadIdx.x, cnt - 1);
#pragma unroll
    for (int ch = 0; ch < MAXC; ++ch) {
      long long o = -1LL;
      if (ch < nch) {
        const int ci = tab[kGatherHeader + ch * mb + ib], cj = tab[kGatherHeader + ch * mb + jb];
        const long long base = ((long long)tab[2 + 2 * ch] << 32) | (long long)(unsigned int)tab[1 + 2 * ch];
        o = (ci | cj) >= 0 ? base + (long long)(ci * (ci + 1) / 2 + cj) * BB : -1LL;
      }
      soff[ch][threadIdx.x] = o;
    }
  }
  __syncthreads();
  // scalar column c of the block column, rows down the chunk: consecutive threads, consecutive addresses of F; the elements of one
  // thread (rows tid, tid + 256 of every column) are requested three at a time
  const int rows = cnt * BS;
  double* Fc = F + (size_t)ck.z * BS + (size_t)m * (jb * BS);
  for (int r0 = threadIdx.x; r0 < rows; r0 += (int)blockDim.x) {
    const int blk = r0 / BS, eb = r0 - blk * BS;
    long long o[MAXC];
#pragma unroll
    for (int ch = 0; ch < MAXC; ++ch) o[ch] = soff[ch][blk];
    constexpr int CU = (MAXC <= 2 && BS <= 6) ? BS : 3;   // columns in flight
#pragma unroll
    for (int c0 = 0; c0 < BS; c0 += CU) {
      double v[CU], u[CU][MAXC];
      bool any = false;
#pragma unroll
      for (int q = 0; q < CU; ++q) {
        const int c = c0 + q < BS ? c0 + q : BS - 1;
#pragma unroll
        for (int ch = 0; ch < MAXC; ++ch) u[q][ch] = o[ch] >= 0 ? P.U[o[ch] + eb + BS * c] : 0.0;
        v[q] = WRITE ? 0.0 : Fc[(size_t)r0 + (size_t)m * c];
      }
#pragma unroll
      for (int ch = 0; ch < MAXC; ++ch) any = any || o[ch] >= 0;
#pragma unroll
      for (int q = 0; q < CU; ++q) {
        if (c0 + q < BS && (WRITE || any)) {
          double x = v[q];
#pragma unroll
          for (int ch = 0; ch < MAXC; ++ch)
            if (o[ch] >= 0) x += u[q][ch];   // (child order; a child without an entry here adds nothing -- not even + 0.0)
          Fc[(size_t)r0 + (size_t)m * (c0 + q)] = x;
        }
      }
    }
  }
}

// pivot block (npiv <= 64): right-looking Cholesky, one wave per front, lane i owns row i IN REGISTERS; column j of L
// reaches the other lanes through v_readlane (the loops are fully unrolled: lane and register indices are
// compile-time constants) -- no LDS, no barriers, ~n^2/2 FMAs with scalar operands
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
template <int BS, bool COH>
__device__ __forceinline__ void big_diag_body(const CholPlanDev& P, int slot, int lane, double* __restrict__ scratch,
                                              const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  // COH: the panel block and the reciprocal diagonal go to L with agent-scope stores (read by other workgroups of the
  // SAME launch: big_panel_kernel)
  constexpr int MAXB = 64 / BS, N = MAXB * BS;
  const int f = P.slots[slot].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int ns = rec.ns, m = (rec.ns + rec.nb) * BS, n = ns * BS;
  const int ld = scratch_ld[slot];   // leading dimension in the slab (>= m: a front continued in place sits inside its chain's first front)
  double* F = scratch + scratch_off[slot];
  double* Lg = P.L + rec.L_off;
  double r[N];
#pragma unroll
  for (int c = 0; c < N; ++c) r[c] = (c < n && lane < n && lane >= c) ? F[(size_t)lane + (size_t)ld * c] : 0.0;
  bool bad = false;
#pragma unroll
  for (int jb = 0; jb < MAXB; ++jb) {
    if (jb < ns) {   // (wave-uniform)
#pragma unroll
      for (int jj = 0; jj < BS; ++jj) {
        const int j = jb * BS + jj;
        double d = readlane_f64(r[j], j);
        if (!(d > 0.0)) {
          bad = true;
          d = 1.0;
        }
        double sq, rs;
        sqrt_and_rsqrt(d, sq, rs);
        const double l = lane > j ? r[j] * rs : (lane == j ? sq : 0.0);   // column j of L (zero above the diagonal)
        r[j] = l;
        if (lane == j) {
          if (COH) st_coh(Lg + (size_t)m * n + j, rs);
          else Lg[(size_t)m * n + j] = rs;
        }
#pragma unroll
        for (int c = j + 1; c < (jb + 1) * BS; ++c) r[c] -= l * readlane_f64(l, c);
#pragma unroll
        for (int cb = jb + 1; cb < MAXB; ++cb)
          if (cb < ns) {
#pragma unroll
            for (int cc = 0; cc < BS; ++cc) r[cb * BS + cc] -= l * readlane_f64(l, cb * BS + cc);
          }
      }
    }
  }
  if (bad && lane == 0) atomicMax(P.status, 1);
  if (lane < n) {
#pragma unroll
    for (int c = 0; c < N; ++c)
      if (c < n) {
        F[(size_t)lane + (size_t)ld * c] = r[c];
        if (COH) st_coh(Lg + (size_t)lane + (size_t)m * c, r[c]);
        else Lg[(size_t)lane + (size_t)m * c] = r[c];
      }
  }
}
template <int BS>
__global__ void __launch_bounds__(64) big_diag_kernel(CholPlanDev P, int slot0, double* __restrict__ scratch,
                                                     const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  big_diag_body<BS, false>(P, slot0 + blockIdx.x, threadIdx.x, scratch, scratch_off, scratch_ld);
}

// panel rows below the pivot block: x L11' = row, one thread per row, the row in registers
template <int BS>
__global__ void __launch_bounds__(256) big_trsm_kernel(CholPlanDev P, const int4* __restrict__ chunks, double* __restrict__ scratch,
                                                      const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  __shared__ double S[64 * 65];
  __shared__ double inv[64];
  constexpr int MAXB = 64 / BS;
  const int4 ck = chunks[blockIdx.x];   // x: launch slot, y: first row (relative to the pivot block's end)
  const int f = P.slots[ck.x].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int ns = rec.ns, m = (rec.ns + rec.nb) * BS, n = ns * BS;
  const int ld = scratch_ld[ck.x];
  double* F = scratch + scratch_off[ck.x];
  {   // thread = (row, column group of 4): all 16 columns of a thread in one round trip
    const int si = threadIdx.x & 63, sg = threadIdx.x >> 6;
    double t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = sg + 4 * u;
      t[u] = (si < n && j < n) ? F[(size_t)si + (size_t)ld * j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (sg + 4 * u < n) S[si + 65 * (sg + 4 * u)] = t[u];
  }
  __syncthreads();
  if (threadIdx.x < n) inv[threadIdx.x] = 1.0 / S[threadIdx.x * 66];
  __syncthreads();
  const int i = n + ck.y + threadIdx.x;
  if (i >= m) return;
  double* Lg = P.L + rec.L_off;
  double x[MAXB * BS];
#pragma unroll
  for (int cb = 0; cb < MAXB; ++cb)   // the whole row first: one round trip
#pragma unroll
    for (int c = 0; c < BS; ++c) x[cb * BS + c] = (cb < ns) ? F[(size_t)i + (size_t)ld * (cb * BS + c)] : 0.0;
#pragma unroll
  for (int cb = 0; cb < MAXB; ++cb) {
    if (cb < ns) {
#pragma unroll
      for (int qb = 0; qb < cb; ++qb)
#pragma unroll
        for (int c = 0; c < BS; ++c)
#pragma unroll
          for (int q = 0; q < BS; ++q) x[cb * BS + c] -= x[qb * BS + q] * S[(cb * BS + c) + 65 * (qb * BS + q)];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        double v = x[cb * BS + c];
#pragma unroll
        for (int q = 0; q < c; ++q) v -= x[cb * BS + q] * S[(cb * BS + c) + 65 * (cb * BS + q)];
        v *= inv[cb * BS + c];
        x[cb * BS + c] = v;
        F[(size_t)i + (size_t)ld * (cb * BS + c)] = v;
        Lg[(size_t)i + (size_t)m * (cb * BS + c)] = v;
      }
    }
  }
}

// Rank-npiv update of the trailing part of the scratch-slab (large) fronts of one level, over the whole GPU:
//   U(ib, jb) = F(ns+ib, ns+jb) - L(ns+ib, 0..npiv) L(ns+jb, 0..npiv)'        (packed lower blocks of P.U)
// one workgroup per 64 x 64 tile of the trailing matrix, one 32 x 32 quadrant per wave as 2 x 2 MFMA tiles
// (v_mfma_f64_16x16x4_f64: a true GEMM contraction, K = npiv).  Operands straight from the panel in the slab
// (L2-resident, 128-byte coalesced: 16 consecutive rows per k).  Layout of the instruction (probed on gfx950,
// tools/probe/mfma_f64_layout.hip): A[i][k] from lane i + 16k, B[j][k] from lane j + 16k, D[lane/16 + 4v][lane%16]
// in register v.  The ROW of the trailing matrix rides on j (lanes 0..15: consecutive addresses), the column on i.
template <int BS, int KS = 15>
__global__ void __launch_bounds__(256) big_front_update_kernel(CholPlanDev P, const int4* __restrict__ tiles,
                                                              double* __restrict__ scratch,
                                                              const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  constexpr int BB = BS * BS;
  const int4 td = tiles[blockIdx.x];   // x: launch slot of the front, y / z: tile row / column, w: bit 0 the parent continues in place,
                                       // bits 1..15 pivot columns of the group's earlier panels, bits 16..31 column limit (grouped chains)
                                       // (consecutive tiles on ONE XCD -- the same row operands in one L2 -- measured: 16.5 -> 17.2 / 80.6 -> 90.0 ms, slower)
  const int f = P.slots[td.x].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int ns = rec.ns, nbd = rec.nb;
  const int npiv = ns * BS, mt = nbd * BS;
  const int kprev = (td.w >> 1) & 0x7fff, climit = ((unsigned)td.w >> 16) ? (int)((unsigned)td.w >> 16) : mt;
  const bool inplace = td.w & 1;
  const int m = scratch_ld[td.x];   // leading dimension of the front in the slab
  double* F = scratch + scratch_off[td.x];
  double* U = P.U + rec.U_off;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
  const int r0 = td.y * 64 + (wave & 1) * 32, c0 = td.z * 64 + (wave >> 1) * 32;
  if (r0 >= mt || c0 >= climit || c0 > r0 + 31) return;   // outside, or entirely above the diagonal
  mfma_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  // this wave's part of the trailing tile (final before this launch), requested with the operands and consumed after the update
  // (49 729-camera grid graph 91.4 -> 90.4 ms)
  double fpre[16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = r0 + 16 * a + lr, c = c0 + 16 * b + lk + 4 * v;
        fpre[(a * 2 + b) * 4 + v] = (r < mt && c < climit && r / BS >= c / BS) ? F[(size_t)(npiv + r) + (size_t)m * (npiv + c)] : 0.0;
      }
  // Operands straight from the L panels (L2-resident, 128-byte coalesced: 16 consecutive rows per k), KS k-steps requested together (a
  // step per round trip would make the kernel a chain of L2 latencies).  kprev > 0 (the last panel of a grouped in-place chain): the
  // group's earlier panels first -- their solved rows for these rows sit in THEIR L panels (CholPlanDev::gtab) --, then this front's:
  // one rank-(kprev + npiv) update.
  // KS: k-steps requested together (15: a panel of 60 columns in ONE round trip: 93.1 -> 90.9 ms on the 49 729-camera grid graph against 6; 16 loses it again)
  const int rr[2] = {min(r0 + lr, mt - 1), min(r0 + 16 + lr, mt - 1)}, cc[2] = {min(c0 + lr, mt - 1), min(c0 + 16 + lr, mt - 1)};
  auto rank_update = [&](const double* __restrict__ Lp, long long mm, int nn) {   // Lp: row 0 of the trailing part, column 0 of the panel
    for (int k00 = 0; k00 < nn; k00 += 4 * KS) {
      double rv[KS][2], cv[KS][2];
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const int k0 = k00 + 4 * s_;
        const int k = min(k0 + lk, nn - 1);
        const double keep = (k0 + lk < nn) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          rv[s_][q] = Lp[rr[q] + mm * k] * keep;
          cv[s_][q] = Lp[cc[q] + mm * k];
        }
      }
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[s_][b], rv[s_][a], acc[a][b], 0, 0, 0);
    }
  };
  if (kprev > 0) {
    const int* gt = P.gtab + P.gtab_off[f];
    const int np_ = gt[0];
    for (int j = 0; j < np_; ++j) {
      const int* e = gt + 1 + 5 * j;
      const long long lo = ((long long)e[1] << 32) | (long long)(unsigned int)e[0];
      rank_update(P.L + lo + e[3] + e[4], (long long)e[2], e[3]);
    }
  }
  rank_update(P.L + rec.L_off + npiv, (long long)(npiv + mt), npiv);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = r0 + 16 * a + lr, c = c0 + 16 * b + lk + 4 * v;
        if (r < mt && c < climit) {
          const int ib = r / BS, jb = c / BS;
          if (ib >= jb) {
            const double x = fpre[(a * 2 + b) * 4 + v] - acc[a][b][v];
            if (inplace) F[(size_t)(npiv + r) + (size_t)m * (npiv + c)] = x;   // the parent is factorised here, in place
            else U[(size_t)(ib * (ib + 1) / 2 + jb) * BB + (r - ib * BS) + BS * (c - jb * BS)] = x;
          }
        }
      }
}

// Children's update vectors added to the rows [lo, lo + cnt) of a front's right-hand side kept in LDS (dst[0 .. cnt)),
// one child at a time and in child order (rows of two children may coincide; same order as front_forward_kernel).  The
// loads of the first two children (embedded in the front record, up to 1 024 boundary rows each) are issued by
// fwd_children_prefetch before the caller's own work and consumed here.  All 256 threads of the workgroup call both.
struct FwdChildPre {
  double w[2][4];
  int d[2][4];
  bool fast;
};
template <int BS>
__device__ __forceinline__ void fwd_children_prefetch(const CholPlanDev& P, const FrontRec& rec, int tid, FwdChildPre& pre) {
  const int nch = rec.child_cnt;
  const int n0 = nch > 0 ? rec.ch[0].nbc * BS : 0, n1 = nch > 1 ? rec.ch[1].nbc * BS : 0;
  pre.fast = nch <= 2 && n0 <= 1024 && n1 <= 1024;
  if (!pre.fast) return;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int nc = c == 0 ? n0 : n1;
    const double* wc = P.w + rec.ch[c].w_off;
    const int* rel = P.crel + rec.crel_off + rec.ch[c].crel_start;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      const bool on = i < nc;
      pre.w[c][u] = on ? wc[i] : 0.0;
      pre.d[c][u] = on ? rel[i / BS] * BS + i % BS : -1;
    }
  }
}
template <int BS>
__device__ __forceinline__ void fwd_children_apply(const CholPlanDev& P, const FrontRec& rec, int tid, const FwdChildPre& pre, double* dst,
                                                   int lo, int cnt, double* dst2 = nullptr, int lo2 = 0, int cnt2 = 0) {
  // (two disjoint row ranges of the front: dst <- [lo, lo + cnt), dst2 <- [lo2, lo2 + cnt2))
  const int nch = rec.child_cnt;
  auto put = [&](int d, double v) {
    const int r = d - lo, r2 = d - lo2;
    if (r >= 0 && r < cnt) dst[r] += v;
    else if (r2 >= 0 && r2 < cnt2) dst2[r2] += v;
  };
  if (pre.fast) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (pre.d[c][u] >= 0) put(pre.d[c][u], pre.w[c][u]);
        __syncthreads();
      }
    }
  } else {
    for (int ch = 0; ch < nch; ++ch) {
      const ChildDesc cd = P.cdesc[rec.child_off + ch];
      const int nbc = cd.nbc * BS;
      const double* wc = P.w + cd.w_off;
      const int* rel = P.crel + rec.crel_off + cd.crel_start;
      for (int i = tid; i < nbc; i += 256) put(rel[i / BS] * BS + i % BS, wc[i]);
      __syncthreads();
    }
  }
}

// Children's update matrices gathered where the frontal matrix is loaded (big_level_kernel<.., GATH>) instead of by one extend-add
// pass per child ordinal in front of the level's launch (two launches of ~6 us on every level of a pose graph's chain): the value at
// front position (r, c), block of r >= block of c, plus the children's entries there in child order -- the very sums the passes
// form, bit for bit.  inv: per child ordinal, front block -> the child's boundary block or -1 (CholPlanDev::cinv).
struct GatherCtx {
  const int* tab;   // the front's table in LDS (gather_stage): [0] children, [1 + 2 c], [2 + 2 c] offset of child c's update matrix, maps from kGatherHeader
  const double* U;
  int on, mb;
  __device__ __forceinline__ int children() const { return on ? tab[0] : 0; }   // (after the barrier behind gather_stage)
  __device__ __forceinline__ const double* child_U(int ch) const {
    return U + (((long long)tab[2 + 2 * ch] << 32) | (long long)(unsigned int)tab[1 + 2 * ch]);
  }
  __device__ __forceinline__ int inv(int ch, int b) const { return tab[kGatherHeader + ch * mb + b]; }
};
// whole workgroup; the table is addressed by the LAUNCH SLOT (requested next to the front record, not behind it); the caller puts a
// barrier between this and the first use
__device__ __forceinline__ GatherCtx gather_stage(const CholPlanDev& P, int slot, int mb, int* sinv) {
  const int2 cs = P.cinv_slot[slot];
  const int* src = P.cinv + max(cs.x, 0);
  for (int i = threadIdx.x; i < cs.y; i += blockDim.x) sinv[i] = src[i];
  return GatherCtx{sinv, P.U, cs.x >= 0 ? 1 : 0, mb};
}
// branch-free (the loads of a child's entries for all the caller's values are in flight together): the entry or, where the child
// has none, a dummy load of its first double and the value unchanged
template <int BS>
__device__ __forceinline__ double gather_child(const GatherCtx& g, int ch, const double* __restrict__ Uc, double v, int rb, int cb, int e) {
  const int ci = g.inv(ch, rb), cj = g.inv(ch, cb);
  const bool ok = (ci | cj) >= 0;
  const double u = Uc[ok ? (size_t)(ci * (ci + 1) / 2 + cj) * (BS * BS) + e : (size_t)0];
  return ok ? v + u : v;
}

// Panel solve and trailing update of the scratch-slab fronts of one level in ONE launch (two otherwise: on the critical
// path of a pose graph every launch is ~5 us of start-up next to a few us of work).  A workgroup owns a 64 x 64 tile of
// a trailing matrix: it solves ITS 2 x 64 panel rows against L11' itself (the rows of a tile are solved again by the
// other tiles that need them -- a few thousand redundant FMAs instead of a launch and a round trip through memory),
// keeps them in LDS as the operands of the MFMA update and -- the tiles of the first tile column -- writes them to L.
// (A variant that also carried the pivot block, with flags between workgroups, needed the registers of both roles at
// once: 674 spilled registers, slower than three launches.)
// FWD: the forward step of the front rides along, no forward launch of its own for the level (12 us on the chain).  The
// tiles of the first tile column hold L11 and their 64 solved panel rows in LDS: while waves 0-1 solve the panel rows,
// wave 2 solves the pivot part y = L11^-1 (b + children's w) (one wave, column by column, no barriers; the operation
// order per row is that of front_forward_kernel: bit-identical), then the rows of the update vector follow:
// w = (children's w) - L21 y.  Tile (0, 0) stores y.  (Fronts without boundary rows have no tiles: big_diag_mfma_kernel.)
// DEP (big_level_kernel: pivot blocks and panel tiles of a level in ONE launch): the tile waits for its front's flag -- with
// its panel rows, the children's vectors and the records already requested -- and reads L11 with device-coherent loads.
template <int BS, bool FWD, bool DEP, bool GATH = false>
__device__ __forceinline__ void big_panel_body(const CholPlanDev& P, const int4 td, double* __restrict__ scratch,
                                               const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld,
                                               const double* __restrict__ bperm, double* __restrict__ yout, double* psm, const int* flag) {
  constexpr int BB = BS * BS, MAXB = 64 / BS, NX = MAXB * BS;
  const int tid = threadIdx.x;
  // td: x: launch slot, y / z: tile row / column, w: bit 0 the parent continues in place (the panels of a GROUPED in-place chain never
  // come here: the group's last panel reads the earlier panels' solved rows from the frontal matrix, where big_trsm_kernel leaves them
  // and this kernel does not -- its tiles read the unsolved rows of the same launch; measured: it would buy 2 ms of 28)
  const int slot = td.x;
  const int f = P.slots[slot].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int ns = rec.ns, nbd = rec.nb;
  const int n = ns * BS, mt = nbd * BS, m = n + mt;
  const int climit = ((unsigned)td.w >> 16) ? (int)((unsigned)td.w >> 16) : mt;
  const bool inplace = td.w & 1;
  const int ld = scratch_ld[slot];
  double* F = scratch + scratch_off[slot];
  double* Lg = P.L + rec.L_off;
  double* S = psm;                 // L11, [64][65]
  double* inv = S + 64 * 65;       // 1 / diagonal
  double* XR = inv + 64;           // solved rows of the tile's row range, [64][65]: XR[row + 65 k]
  double* XC = XR + 64 * 65;       // ... of its column range
  double* tB = XC + 64 * 65;       // FWD: right-hand side rows of the tile row range, [64]
  double* ysl = tB + 64;           // FWD: pivot part of the right-hand side, then of the solution, [64]
  FwdChildPre pre;
  const bool fwd_rows = FWD && td.z == 0;
  double bJ = 0.0;
  if (fwd_rows) {
    fwd_children_prefetch<BS>(P, rec, tid, pre);
    if (tid < n) bJ = bperm[(size_t)rec.c0 * BS + tid];
  }
  // this thread's panel row (threads 0..63: row range, 64..127: column range), requested before the wait
  const bool rowthr = tid < 128;
  const int rloc = tid & 63;
  const int rglb = (tid < 64 ? td.y : td.z) * 64 + rloc;      // row of the trailing part
  const bool rok = rowthr && rglb < mt;
  double x[NX];
#pragma unroll
  for (int c = 0; c < NX; ++c) x[c] = (rok && c < n) ? F[(size_t)(n + rglb) + (size_t)ld * c] : 0.0;
  GatherCtx gc{nullptr, nullptr, 0, 0};
  int gnch = 0;
  if (GATH) {   // the children's entries of this panel row (the extend-add passes' sums: own value first, then child by child)
    gc = gather_stage(P, slot, ns + nbd, reinterpret_cast<int*>(psm + kGatherLdsOff));
    __syncthreads();
    gnch = gc.children();
    const int rb = ns + rglb / BS, re = rglb % BS;
    for (int ch = 0; ch < gnch; ++ch) {
      const int ci = rok ? gc.inv(ch, rb) : -1;
      const double* Uc = gc.child_U(ch);
      // (all the child's entries of the row requested before the first one is added: one round trip per child, not one per block)
      double u[NX];
      unsigned okm = 0;
#pragma unroll
      for (int cb = 0; cb < MAXB; ++cb) {
        const int cj = cb < ns ? gc.inv(ch, cb) : -1;
        const bool ok = (ci | cj) >= 0;
        okm |= ok ? 1u << cb : 0u;
        const double* ub = Uc + (ok ? (size_t)(ci * (ci + 1) / 2 + cj) * BB + re : (size_t)0);
#pragma unroll
        for (int e = 0; e < BS; ++e) u[cb * BS + e] = ub[ok ? BS * e : 0];
      }
#pragma unroll
      for (int cb = 0; cb < MAXB; ++cb)
#pragma unroll
        for (int e = 0; e < BS; ++e) x[cb * BS + e] = ((okm >> cb) & 1u) ? x[cb * BS + e] + u[cb * BS + e] : x[cb * BS + e];
    }
  }
  // this wave's part of the trailing tile (final before this launch: earlier levels and the extend-add passes wrote it):
  // requested now, consumed after the update -- one exposed round trip less per level (not with 7 x 7 blocks: no registers left)
  constexpr bool kPreTile = BS != 7;
  double fpre[kPreTile ? 16 : 1];
  if (kPreTile) {
    const int l_ = tid & 63, lr_ = l_ & 15, lk_ = l_ >> 4, wave_ = tid >> 6;
    const int R0_ = td.y * 64 + (wave_ & 1) * 32, C0_ = td.z * 64 + (wave_ >> 1) * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = R0_ + 16 * a + lr_, c = C0_ + 16 * b + lk_ + 4 * v;
          fpre[(a * 2 + b) * 4 + v] = (r < mt && c < climit && r / BS >= c / BS) ? F[(size_t)(n + r) + (size_t)ld * (n + c)] : 0.0;
        }
    if (GATH)
      for (int ch = 0; ch < gnch; ++ch) {   // (requested together, added afterwards: one round trip per child)
        const double* Uc = gc.child_U(ch);
        double u[16];
        unsigned okm = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int r = min(R0_ + 16 * a + lr_, mt - 1), c = min(C0_ + 16 * b + lk_ + 4 * v, mt - 1);
              const bool in = R0_ + 16 * a + lr_ < mt && C0_ + 16 * b + lk_ + 4 * v < climit && r / BS >= c / BS;
              const int ci = gc.inv(ch, ns + r / BS), cj = gc.inv(ch, ns + c / BS);
              const bool ok = in && (ci | cj) >= 0;
              okm |= ok ? 1u << ((a * 2 + b) * 4 + v) : 0u;
              u[(a * 2 + b) * 4 + v] = Uc[ok ? (size_t)(ci * (ci + 1) / 2 + cj) * BB + (r % BS) + BS * (c % BS) : (size_t)0];
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) fpre[i] = ((okm >> i) & 1u) ? fpre[i] + u[i] : fpre[i];
      }
  }
#ifdef G2OHIP_CHOL_STAMPS
  int nst_ = 0;
#define BSTAMP() do { if (P.dbg && blockIdx.x == gridDim.x - 1 && tid == 0 && nst_ < 40) P.dbg[4 + nst_++] = wall_clock64(); } while (0)
#else
#define BSTAMP() do {} while (0)
#endif
  BSTAMP();
  if (DEP) {   // the pivot block of this front is factorised by a workgroup of this launch
    if (tid == 0) {
      const int* fl = flag + f;
      int spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P.dep_spin_limit || ((spins & 255) == 0 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
          __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  BSTAMP();
  {   // L11 and the reciprocal diagonal (written by the pivot-block kernel: the launch before, or -- DEP -- a workgroup of this one)
    constexpr int UL = 16;   // 64 * 64 / 256: element (tid & 63, tid / 64 + 4 u) of a 64 x 64 grid (no integer divisions on this path)
    double t[UL];
    const int sr = tid & 63, sc = tid >> 6;
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int c = sc + 4 * u;
      const double* src = Lg + min(sr, n - 1) + (size_t)m * min(c, n - 1);
      t[u] = DEP ? ld_coh(src) : *src;
    }
    const double iv = DEP ? ld_coh(Lg + (size_t)m * n + min(tid, n - 1)) : Lg[(size_t)m * n + min(tid, n - 1)];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int c = sc + 4 * u;
      if (sr < n && c < n) S[sr + 65 * c] = t[u];
    }
    if (tid < n) inv[tid] = iv;
    if (fwd_rows && tid < 64) {
      tB[tid] = 0.0;
      ysl[tid] = bJ;
    }
  }
  __syncthreads();
  BSTAMP();
  if (fwd_rows) fwd_children_apply<BS>(P, rec, tid, pre, tB, n + td.y * 64, 64, ysl, 0, n);
  if (fwd_rows && (tid >> 6) == 2) {   // wave 2: y = L11^-1 t, next to the row solves of waves 0-1
    const int l = tid & 63;
    double tj = l < n ? ysl[l] : 0.0, yk_mine = 0.0;
    for (int k0 = 0; k0 < n; k0 += 8) {
      double sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) sv[u] = (k0 + u < n && l < n) ? S[l + 65 * (k0 + u)] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        if (k < n) {   // (uniform)
          const double yk = __shfl(tj, k) * inv[k];
          if (l == k) yk_mine = yk;
          if (l > k) tj -= sv[u] * yk;
        }
      }
    }
    if (l < n) {
      ysl[l] = yk_mine;
      if (td.y == 0) yout[(size_t)rec.c0 * BS + l] = yk_mine;
    }
  }
  if (rowthr) {   // x L11' = row (big_trsm_kernel)
    double* X = tid < 64 ? XR : XC;
#pragma unroll
    for (int cb = 0; cb < MAXB; ++cb) {
      if (cb < ns) {
#pragma unroll
        for (int qb = 0; qb < cb; ++qb)
#pragma unroll
          for (int c = 0; c < BS; ++c)
#pragma unroll
            for (int q = 0; q < BS; ++q) x[cb * BS + c] -= x[qb * BS + q] * S[(cb * BS + c) + 65 * (qb * BS + q)];
#pragma unroll
        for (int c = 0; c < BS; ++c) {
          double v = x[cb * BS + c];
#pragma unroll
          for (int q = 0; q < c; ++q) v -= x[cb * BS + q] * S[(cb * BS + c) + 65 * (cb * BS + q)];
          v *= inv[cb * BS + c];
          x[cb * BS + c] = v;
          X[rloc + 65 * (cb * BS + c)] = v;
        }
      }
    }
  }
  __syncthreads();
  BSTAMP();
  if (td.z == 0)   // the tiles of the first tile column write their rows of the panel to L (coalesced, from LDS)
    for (int i = tid; i < 64 * n; i += 256) {
      const int r = i & 63, k = i >> 6;
      if (td.y * 64 + r < mt) Lg[(size_t)(n + td.y * 64 + r) + (size_t)m * k] = XR[r + 65 * k];
    }
  if (fwd_rows && tid < 64 && td.y * 64 + tid < mt) {
    double v = tB[tid];
    for (int k = 0; k < n; ++k) v -= XR[tid + 65 * k] * ysl[k];
    P.w[rec.w_off + td.y * 64 + tid] = v;
  }
  {   // rank-n update of this tile (big_front_update_kernel, operands from LDS)
    double* U = P.U + rec.U_off;
    const int wave = tid >> 6, l = tid & 63, lr = l & 15, lk = l >> 4;
    const int r0 = (wave & 1) * 32, c0 = (wave >> 1) * 32;           // inside the tile
    const int R0 = td.y * 64 + r0, C0 = td.z * 64 + c0;               // in the trailing part
    if (R0 < mt && C0 < climit && C0 <= R0 + 31) {
      mfma_d4 acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
      for (int k0 = 0; k0 < n; k0 += 4) {
        const int k = min(k0 + lk, n - 1);
        const double keep = (k0 + lk < n) ? 1.0 : 0.0;
        double rv[2], cv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          rv[q] = XR[(r0 + 16 * q + lr) + 65 * k] * keep;
          cv[q] = XC[(c0 + 16 * q + lr) + 65 * k];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[b], rv[a], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = R0 + 16 * a + lr, c = C0 + 16 * b + lk + 4 * v;
            if (r < mt && c < climit) {
              const int ib = r / BS, jb = c / BS;
              if (ib >= jb) {
                double y0 = kPreTile ? fpre[(a * 2 + b) * 4 + v] : F[(size_t)(n + r) + (size_t)ld * (n + c)];
                if (GATH && !kPreTile)
                  for (int ch = 0; ch < gnch; ++ch) y0 = gather_child<BS>(gc, ch, gc.child_U(ch), y0, ns + ib, ns + jb, (r - ib * BS) + BS * (c - jb * BS));
                const double y = y0 - acc[a][b][v];
                if (inplace) F[(size_t)(n + r) + (size_t)ld * (n + c)] = y;
                else U[(size_t)(ib * (ib + 1) / 2 + jb) * BB + (r - ib * BS) + BS * (c - jb * BS)] = y;
              }
            }
          }
    }
  }
  __syncthreads();
  BSTAMP();
#ifdef G2OHIP_CHOL_STAMPS
  if (P.dbg && blockIdx.x == gridDim.x - 1 && tid == 0) { P.dbg[0] = nst_; P.dbg[1] = td.y; P.dbg[2] = td.z; P.dbg[3] = n * 1000 + mt; }
#endif
}

template <int BS, bool FWD>
__global__ void __launch_bounds__(256) big_panel_kernel(CholPlanDev P, const int4* __restrict__ tiles, double* __restrict__ scratch,
                                                       const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld,
                                                       const double* __restrict__ bperm, double* __restrict__ yout) {
  extern __shared__ __attribute__((aligned(16))) double psm[];
  big_panel_body<BS, FWD, false>(P, tiles[blockIdx.x], scratch, scratch_off, scratch_ld, bperm, yout, psm, nullptr);
}

// b_perm[new*bs + r] = b[old*bs + r]
__global__ void permute_in_kernel(int nb, int bs, const int* __restrict__ perm, const double* __restrict__ b, double* __restrict__ bp) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * bs) return;
  int k = t / bs, r = t - k * bs;
  bp[t] = b[(size_t)perm[k] * bs + r];
}
__global__ void permute_out_kernel(int nb, int bs, const int* __restrict__ perm, const double* __restrict__ xp, double* __restrict__ x) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * bs) return;
  int k = t / bs, r = t - k * bs;
  x[(size_t)perm[k] * bs + r] = xp[t];
}

// Forward sweep over one task (chain of fronts, leaves first):
//   y1 = L11 \ (b1 + children), w = (children on boundary) - L21 y1
// Blocked by BS: every thread redundantly solves the BS x BS triangular diagonal system (broadcast LDS
// reads, reciprocal diagonal), then the rows below are updated in parallel -> one barrier per pivot
// block.  Inside a chain the update vector w stays in LDS.
// LDS: [panel (optional)] [t: mcap] [ys: mcap] [wprev: mcap]
template <int BS, bool PANEL_LDS, bool GVEC = false>
__global__ void __launch_bounds__(256) front_forward_kernel(CholPlanDev P, int slot0,
                                                           const double* __restrict__ bperm, double* __restrict__ y, int panel_cap,
                                                           int mcap, double* gvec) {
  // GVEC: the three vectors of a front live in HBM (gvec) (3 * mcap doubles per workgroup) instead of LDS -- frontal
  // matrices of more than ~6 500 rows (dense reduced systems); the barriers order the accesses as they order LDS.
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int2 slot = P.slots[slot0 + blockIdx.x];
  const int f_first = __builtin_amdgcn_readfirstlane(slot.x), t0 = 0, t1 = __builtin_amdgcn_readfirstlane(slot.y);
  const int tid = threadIdx.x, NT = blockDim.x;
  double* Lp = smem;
  // (a compile-time choice: a run-time one would turn every access of the LDS variant into a flat access)
  double* t = GVEC ? gvec + (size_t)blockIdx.x * 3 * mcap : smem + (PANEL_LDS ? panel_cap : 0);
  double* ys = t + mcap;
  double* wprev = ys + mcap;
  int nprev = 0;
  for (int ti = t0; ti < t1; ++ti) {
    const int f = f_first + ti;
    const FrontRec rec = load_front_rec(P.rec + f);
    const int ns = rec.ns, nbd = rec.nb;
    const int m = (ns + nbd) * BS, npiv = ns * BS, c0 = rec.c0;
    const double* Lg = P.L + rec.L_off;
    const double* Lx = PANEL_LDS ? Lp : Lg;
    const double* Linv = Lx + (size_t)m * npiv;
    // Everything this front reads from memory is requested before the first wait (one round trip): the
    // right-hand side, the children's update vectors (fast path: <= 2 children whose boundary fits one
    // round; applied one child at a time, rows may coincide) and the L panel.
    const double bfirst = (tid < npiv) ? bperm[(size_t)c0 * BS + tid] : 0.0;
    const bool carried = nprev > 0;
    const int nch = rec.child_cnt;
    const int n0 = nch > 0 ? rec.ch[0].nbc * BS : 0, n1 = nch > 1 ? rec.ch[1].nbc * BS : 0;
    const bool fast_children = nch <= 2 && n0 <= NT && n1 <= NT;
    double w0 = 0.0, w1 = 0.0;
    int d0 = 0, d1 = 0;
    if (fast_children) {
      if (tid < n0) {
        w0 = carried ? 0.0 : P.w[rec.ch[0].w_off + tid];
        d0 = P.crel[rec.crel_off + rec.ch[0].crel_start + tid / BS] * BS + tid % BS;
      }
      if (tid < n1) {
        w1 = P.w[rec.ch[1].w_off + tid];
        d1 = P.crel[rec.crel_off + rec.ch[1].crel_start + tid / BS] * BS + tid % BS;
      }
    }
    if (PANEL_LDS) stage_copy<16>(Lp, Lg, m * npiv + npiv, tid, NT);
    if (tid < m) t[tid] = bfirst;
    for (int i = tid + NT; i < m; i += NT) t[i] = (i < npiv) ? bperm[(size_t)c0 * BS + i] : 0.0;
    __syncthreads();
    if (fast_children) {
      if (tid < n0) t[d0] += carried ? wprev[tid] : w0;
      if (nch > 1) {
        __syncthreads();
        if (tid < n1) t[d1] += w1;
      }
      if (nch > 0) __syncthreads();
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        const ChildDesc cd = P.cdesc[rec.child_off + ch];
        const int nbc = cd.nbc * BS;
        const double* wc = P.w + cd.w_off;
        const int* rel = P.crel + rec.crel_off + cd.crel_start;
        for (int i = tid; i < nbc; i += NT) t[rel[i / BS] * BS + (i % BS)] += (carried ? wprev[i] : wc[i]);
        __syncthreads();
      }
    }
    for (int kb = 0; kb < ns; ++kb) {
      const int k0 = kb * BS;
      double yv[BS];
      // panel in HBM (large fronts): the L values of this thread's first rows do not depend on y -- they are requested
      // together with the diagonal block (one round trip per pivot block instead of two)
      constexpr int RU = 4;
      double lv[RU][BS];
      if (!PANEL_LDS) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = min(k0 + BS + tid + u * NT, m - 1);
#pragma unroll
          for (int q = 0; q < BS; ++q) lv[u][q] = Lx[i + (size_t)m * (k0 + q)];
        }
      }
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        double v = t[k0 + c];
#pragma unroll
        for (int q = 0; q < c; ++q) v -= Lx[(k0 + c) + (size_t)m * (k0 + q)] * yv[q];
        yv[c] = v * Linv[k0 + c];
      }
      if (tid == 0) {
#pragma unroll
        for (int c = 0; c < BS; ++c) ys[k0 + c] = yv[c];
      }
      if (!PANEL_LDS) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = k0 + BS + tid + u * NT;
          if (i < m) {
            double v = t[i];
#pragma unroll
            for (int q = 0; q < BS; ++q) v -= lv[u][q] * yv[q];
            t[i] = v;
          }
        }
      }
      for (int i = k0 + BS + tid + (PANEL_LDS ? 0 : RU * NT); i < m; i += NT) {
        double v = t[i];
#pragma unroll
        for (int q = 0; q < BS; ++q) v -= Lx[i + (size_t)m * (k0 + q)] * yv[q];
        t[i] = v;
      }
      __syncthreads();
    }
    for (int i = tid; i < npiv; i += NT) y[(size_t)c0 * BS + i] = ys[i];
    if (ti + 1 < t1) {
      nprev = nbd * BS;
      for (int i = tid; i < nprev; i += NT) wprev[i] = t[npiv + i];
    } else {
      nprev = 0;
      double* wf = P.w + rec.w_off;
      for (int i = tid; i < nbd * BS; i += NT) wf[i] = t[npiv + i];
    }
    __syncthreads();
  }
}

// Row parts of the boundary product t[k] -= sum_i L[i,k] t[i] of the backward sweep: NT / npiv lanes per pivot column; small
// fronts (kTreePiv pivot columns, kTreeBnd boundary rows at most) use the largest power of two <= min(that, 4) so that
// tree_backward_kernel (two lanes per column, rows by parity) adds the same partial sums in the same order.
__host__ __device__ __forceinline__ int bw_parts(int NT, int npiv, int m) {
  int parts = NT / npiv > 0 ? NT / npiv : 1;
  if (npiv <= kTreePiv && m - npiv <= kTreeBnd) parts = parts >= 4 ? 4 : (parts >= 2 ? 2 : 1);
  return parts;
}

// Backward sweep over one task (chain top first): x1 = L11' \ (y1 - L21' x_boundary); same blocking.
// Inside a chain the child's boundary values are taken from the parent's vector kept in LDS.
// LDS: [panel (optional)] [t: mcap] [xs: mcap] [sp: NT] [fullprev: mcap]
template <int BS, bool PANEL_LDS, bool GVEC = false>
__global__ void __launch_bounds__(256) front_backward_kernel(CholPlanDev P, int slot0,
                                                            const double* __restrict__ y, double* __restrict__ xp, int panel_cap,
                                                            int mcap, int dep, double* gvec) {
  // dep != 0: dependency-driven launch over several levels, parents in front of their children; the top front of
  // a task waits for the parent front's counter, the bottom front releases its children (see the factor kernel).
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int2 slot = P.slots[slot0 + blockIdx.x];
  const int f_first = __builtin_amdgcn_readfirstlane(slot.x), t0 = 0, t1 = __builtin_amdgcn_readfirstlane(slot.y);
  const int tid = threadIdx.x, NT = blockDim.x;
  double* Lp = smem;
  double* t = GVEC ? gvec + (size_t)blockIdx.x * 3 * mcap : smem + (PANEL_LDS ? panel_cap : 0);   // (GVEC: see front_forward_kernel)
  double* xs = t + mcap;
  double* sp = GVEC ? smem + (PANEL_LDS ? panel_cap : 0) : xs + mcap;
  double* fullprev = GVEC ? xs + mcap : sp + NT;
  const int* prel = nullptr;   // this front's relative indices inside its chain parent
  for (int ti = t1 - 1; ti >= t0; --ti) {
    const int f = f_first + ti;
    const FrontRec rec = load_front_rec(P.rec + f);
    const int ns = rec.ns, nbd = rec.nb;
    const int m = (ns + nbd) * BS, npiv = ns * BS, c0 = rec.c0;
    const double* Lg = P.L + rec.L_off;
    const double* Lx = PANEL_LDS ? Lp : Lg;
    const double* Linv = Lx + (size_t)m * npiv;
    // request the pivot part of y and the boundary row indices (then the boundary values, one dependent
    // round trip) before the panel, so that the panel's round trip covers them
    {
      const int* relp = prel ? prel : P.rows + rec.rows_off;
      const bool bnd = tid >= npiv && tid < m;
      const double yfirst = (tid < npiv) ? y[(size_t)c0 * BS + tid] : 0.0;
      const int rfirst = bnd ? relp[(tid - npiv) / BS] * BS + (tid - npiv) % BS : 0;
      const bool dep_wait = dep && !prel && (rec.pad[0] & 2);
      if (dep_wait) {   // the panel does not depend on the parent: stage it, then wait for the boundary values
        if (PANEL_LDS) stage_copy<16>(Lp, Lg, m * npiv + npiv, tid, NT);
        if (tid == 0) {
          int* flag = P.ready + (rec.pad[1] & 0x00ffffff);
          int spins = 0;
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > P.dep_spin_limit || ((spins & 255) == 0 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
              __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
          __hip_atomic_fetch_add(flag, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last child leaves it at zero
        }
        __syncthreads();
      }
      double xfirst = 0.0;
      if (bnd && !prel) xfirst = ld_coh(xp + rfirst);
      if (PANEL_LDS && !dep_wait) stage_copy<16>(Lp, Lg, m * npiv + npiv, tid, NT);
      if (tid < m) t[tid] = (tid < npiv) ? yfirst : (prel ? fullprev[rfirst] : xfirst);
      for (int i = tid + NT; i < m; i += NT) {
        const int r = relp[(i - npiv) / BS] * BS + (i - npiv) % BS;   // (i >= NT >= npiv here unless npiv > NT)
        t[i] = (i < npiv) ? y[(size_t)c0 * BS + i] : (prel ? fullprev[r] : ld_coh(xp + r));
      }
    }
    __syncthreads();
    // boundary contribution t[k] -= sum_{i>=npiv} L[i,k] t[i]: (k, part) decomposition + LDS reduction
    {
      // (small fronts -- the ones tree_backward_kernel also handles: at most 24 pivot columns and 48 boundary rows -- split the
      // rows into 4, 2 or 1 parts, whatever the workgroup size allows more of: the two kernels form the same sums in the same order)
      const int parts = bw_parts(NT, npiv, m);
      const int k = tid % npiv, part = tid / npiv;
      double s = 0.0;
      if (part < parts && NT >= npiv)
      {
        if (PANEL_LDS) {
          for (int i = npiv + part; i < m; i += parts) s += Lx[i + (size_t)m * k] * t[i];
        } else {
          // panel in HBM (large fronts): eight loads in flight per thread instead of one latency per term
          const double* Lk = Lx + (size_t)m * k;
          double s0 = 0.0, s1 = 0.0;
          for (int i0 = npiv + part; i0 < m; i0 += 8 * parts) {
            double lv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) lv[u] = Lk[min(i0 + u * parts, m - 1)];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
              s0 += lv[u] * (i0 + u * parts < m ? t[i0 + u * parts] : 0.0);
              s1 += lv[u + 1] * (i0 + (u + 1) * parts < m ? t[i0 + (u + 1) * parts] : 0.0);
            }
          }
          s = s0 + s1;
        }
      }
      if (NT < npiv) {
        // fewer threads than pivot columns: loop over the columns serially
        for (int kk = tid; kk < npiv; kk += NT) {
          double s2 = 0.0;
          for (int i = npiv; i < m; ++i) s2 += Lx[i + (size_t)m * kk] * t[i];
          xs[kk] = s2;
        }
        __syncthreads();
        for (int kk = tid; kk < npiv; kk += NT) t[kk] -= xs[kk];
      } else {
        sp[tid] = (part < parts) ? s : 0.0;
        __syncthreads();
        if (tid < npiv) {
          double a = 0.0;
          for (int pp = 0; pp < parts; ++pp) a += sp[pp * npiv + tid];
          t[tid] -= a;
        }
      }
    }
    __syncthreads();
    for (int kb = ns - 1; kb >= 0; --kb) {
      const int k0 = kb * BS;
      double xv[BS];
#pragma unroll
      for (int c = BS - 1; c >= 0; --c) {
        double v = t[k0 + c];
#pragma unroll
        for (int q = c + 1; q < BS; ++q) v -= Lx[(k0 + q) + (size_t)m * (k0 + c)] * xv[q];
        xv[c] = v * Linv[k0 + c];
      }
      if (tid == 0) {
#pragma unroll
        for (int c = 0; c < BS; ++c) xs[k0 + c] = xv[c];
      }
      for (int j = tid; j < k0; j += NT) {
        double v = t[j];
#pragma unroll
        for (int q = 0; q < BS; ++q) v -= Lx[(k0 + q) + (size_t)m * j] * xv[q];
        t[j] = v;
      }
      __syncthreads();
    }
    for (int i = tid; i < npiv; i += NT) st_coh(xp + (size_t)c0 * BS + i, xs[i]);
    if (ti > t0) {
      // the next front down the chain is this one's only child: keep the whole local solution for it
      for (int i = tid; i < m; i += NT) fullprev[i] = (i < npiv) ? xs[i] : t[i];
      prel = P.crel + rec.crel_off + rec.ch[0].crel_start;
    }
    const int dep_release = (dep && ti == t0) ? (rec.pad[0] >> 8) : 0;   // child tasks waiting in this launch
    if (dep_release > 0) __builtin_amdgcn_s_waitcnt(0);   // every wave: its x stores have been acknowledged
    __syncthreads();
    if (dep_release > 0 && tid == 0) __hip_atomic_fetch_add(P.ready + f, dep_release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Backward sweep of the TREE levels by groups of fronts (option tree_backward).  The per-task kernel above pays a hand-off
// between workgroups per level of the elimination tree (the parent's x stored with agent-scope stores and acknowledged, a counter,
// the child's coherent loads: ~4-6 us of the ~11 us a level costs, profiles/r4_tree_handoff.txt) for a 24 x 24 triangular solve
// and a 24 x 48 product.  Here ONE workgroup of sixteen waves owns a GROUP: a front and its descendants down to a depth at
// which at most sixteen fronts are reached (four levels of a binary tree), one wave per front.  Before the group waits for the
// front above its root, every wave has requested its whole L panel into REGISTERS (lane (k, h): column k, rows of parity h --
// 24 + 24 doubles) and its index table; after the wait one round of coherent loads fetches the boundary values that come from
// fronts above the group; inside the group the pivot solutions travel through LDS and the levels are separated by workgroup
// barriers only.  A front releases the tasks below the group through the same `ready` counters as the per-task kernel (which
// sweeps the levels below: the long leaf chains).  Sums and their order are those of front_backward_kernel (bw_parts), bit for bit.
struct TreeGroupRec {
  int first, count, nlev, wait_front;   // entries of the group in the front list; levels; front above the root to wait for (-1: none)
};
// A small front's panel in one wave's registers: lane (k, h) = (lane >> 1, lane & 1) holds of column k the rows of parity h below
// the pivot block (L21[j]: row npiv + h + 2 j) and the whole pivot-block column (L11[q]: row q), the reciprocal diagonal, y.
template <int BS>
struct TreePanel {
  static constexpr int JH = kTreeBnd / 2, NSMAX = kTreePiv / BS;
  double L21[JH], L11[kTreePiv], linv, yk;
  int ns, npiv, nbs, m, c0;
};
template <int BS>
__device__ __forceinline__ void tree_panel_load(const CholPlanDev& P, const FrontRec& rec, const double* __restrict__ y, int lane, TreePanel<BS>& T) {
  T.ns = rec.ns;
  T.npiv = rec.ns * BS;
  T.nbs = rec.nb * BS;
  T.m = T.npiv + T.nbs;
  T.c0 = rec.c0;
  const int k = lane >> 1, h = lane & 1, m = T.m, npiv = T.npiv;
  const bool col = k < npiv;
  const double* Lg = P.L + rec.L_off;
  const double* Lk = Lg + (size_t)m * (col ? k : 0);
#pragma unroll
  for (int j = 0; j < TreePanel<BS>::JH; ++j) {
    const int i = npiv + h + 2 * j;
    const double v = Lk[min(i, m - 1)];
    T.L21[j] = (col && i < m) ? v : 0.0;
  }
#pragma unroll
  for (int q = 0; q < kTreePiv; ++q) {
    const double v = Lk[min(q, m - 1)];
    T.L11[q] = (col && q < npiv) ? v : 0.0;
  }
  T.linv = col ? Lg[(size_t)m * npiv + k] : 0.0;
  T.yk = col ? y[(size_t)rec.c0 * BS + k] : 0.0;
}
// x1 = L11' \ (y1 - L21' t) with the boundary values t[0 .. 48) in LDS (written by this wave): the pivot solution of column k in
// the lanes (k, *).  The sums and their order are front_backward_kernel's (parts = bw_parts of its workgroup size).
template <int BS>
__device__ __forceinline__ double tree_front_solve(const TreePanel<BS>& T, const double* tw, int lane, int parts) {
  constexpr int JH = TreePanel<BS>::JH, NSMAX = TreePanel<BS>::NSMAX;
  const int k = lane >> 1, h = lane & 1;
  double sa = 0.0, sb = 0.0;
  if (parts == 4) {   // part p of front_backward_kernel = rows p, p + 4, ...: the even / odd registers of lane (k, p & 1)
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const double tv = tw[h + 2 * j];
      if (j & 1) sb = fma(T.L21[j], tv, sb);
      else sa = fma(T.L21[j], tv, sa);
    }
  } else {
#pragma unroll
    for (int j = 0; j < JH; ++j) sa = fma(T.L21[j], tw[h + 2 * j], sa);
  }
  const double oa = __shfl_xor(sa, 1), ob = __shfl_xor(sb, 1);
  double a = 0.0;
  a += sa;
  a += oa;
  if (parts == 4) {
    a += sb;
    a += ob;
  }
  double z = T.yk - a, xout = 0.0;
#pragma unroll
  for (int kb = NSMAX - 1; kb >= 0; --kb) {
    if (kb < T.ns) {
      const int k0 = kb * BS;
      double xs[BS];
#pragma unroll
      for (int c = BS - 1; c >= 0; --c) {
        double v = z;
#pragma unroll
        for (int q = c + 1; q < BS; ++q) v = fma(-T.L11[k0 + q], xs[q], v);
        const double xc = v * T.linv;
        xs[c] = readlane_f64(xc, 2 * (k0 + c));
        if (k == k0 + c) xout = xc;
      }
      double zz = z;
#pragma unroll
      for (int q = 0; q < BS; ++q) zz = fma(-T.L11[k0 + q], xs[q], zz);
      z = k < k0 ? zz : z;
    }
  }
  return xout;
}
// (the body below is tree_panel_load + tree_front_solve written out: as calls the compiler spills 65 registers at the 128 this
// sixteen-wave workgroup may use per lane)
template <int BS>
__global__ void __launch_bounds__(kTreeWaves * 64) tree_backward_kernel(CholPlanDev P, const TreeGroupRec* __restrict__ groups,
                                                                       const int2* __restrict__ gfront, const int* __restrict__ grows,
                                                                       const double* __restrict__ y, double* __restrict__ xp, int nt_ref) {
  constexpr int NSMAX = kTreePiv / BS, JH = kTreeBnd / 2;
  __shared__ double xg[kTreeWaves * kTreePiv];
  __shared__ double tw[kTreeWaves][kTreeBnd];
  const TreeGroupRec g = groups[blockIdx.x];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool active = wave < g.count;
  const int2 gf = gfront[g.first + (active ? wave : 0)];   // (front, level inside the group | tasks to release << 8)
  const int f = __builtin_amdgcn_readfirstlane(gf.x), lev = __builtin_amdgcn_readfirstlane(gf.y) & 0xff,
            release = __builtin_amdgcn_readfirstlane(gf.y) >> 8;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int ns = rec.ns, npiv = ns * BS, nbs = rec.nb * BS, m = npiv + nbs, c0 = rec.c0;
  const int k = lane >> 1, h = lane & 1;
  const bool col = k < npiv;
  const double* Lg = P.L + rec.L_off;
  const double* Lk = Lg + (size_t)m * (col ? k : 0);
  // the panel: nothing of it depends on the fronts above
  double L21[JH], L11[kTreePiv];
#pragma unroll
  for (int j = 0; j < JH; ++j) {
    const int i = npiv + h + 2 * j;
    const double v = Lk[min(i, m - 1)];
    L21[j] = (col && i < m) ? v : 0.0;
  }
#pragma unroll
  for (int q = 0; q < kTreePiv; ++q) {
    const double v = Lk[min(q, m - 1)];
    L11[q] = (col && q < npiv) ? v : 0.0;
  }
  const double linv = col ? Lg[(size_t)m * npiv + k] : 0.0;
  const double yk = col ? y[(size_t)c0 * BS + k] : 0.0;
  const int src = lane < nbs ? grows[rec.rows_off + lane / BS] : 0;   // >= 0: block row in memory, < 0: -1 - (offset in xg)
  const int off = lane % BS;
  if (g.wait_front >= 0) {
    if (tid == 0) {
      int* flag = P.ready + g.wait_front;
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P.dep_spin_limit || ((spins & 255) == 0 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
          __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __hip_atomic_fetch_add(flag, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last child leaves it at zero
    }
    __syncthreads();
  }
  const double xb = (active && lane < nbs && src >= 0) ? ld_coh(xp + (size_t)src * BS + off) : 0.0;
  const int parts = bw_parts(nt_ref, max(npiv, 1), m);
  for (int l = 0; l < g.nlev; ++l) {
    if (active && lev == l) {
      if (lane < kTreeBnd) tw[wave][lane] = lane < nbs ? (src < 0 ? xg[-1 - src + off] : xb) : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // (same wave writes and reads: LDS operations of a wave complete in order)
      double sa = 0.0, sb = 0.0;
      if (parts == 4) {   // part p of front_backward_kernel = rows p, p + 4, ...: the even / odd registers of lane (k, p & 1)
#pragma unroll
        for (int j = 0; j < JH; ++j) {
          const double tv = tw[wave][h + 2 * j];
          if (j & 1) sb = fma(L21[j], tv, sb);
          else sa = fma(L21[j], tv, sa);
        }
      } else {
#pragma unroll
        for (int j = 0; j < JH; ++j) sa = fma(L21[j], tw[wave][h + 2 * j], sa);
      }
      const double oa = __shfl_xor(sa, 1), ob = __shfl_xor(sb, 1);
      double a = 0.0;
      a += sa;
      a += oa;
      if (parts == 4) {
        a += sb;
        a += ob;
      }
      double z = yk - a, xout = 0.0;
#pragma unroll
      for (int kb = NSMAX - 1; kb >= 0; --kb) {
        if (kb < ns) {
          const int k0 = kb * BS;
          double xs[BS];
#pragma unroll
          for (int c = BS - 1; c >= 0; --c) {
            double v = z;
#pragma unroll
            for (int q = c + 1; q < BS; ++q) v = fma(-L11[k0 + q], xs[q], v);
            const double xc = v * linv;
            xs[c] = readlane_f64(xc, 2 * (k0 + c));
            if (k == k0 + c) xout = xc;
          }
          double zz = z;
#pragma unroll
          for (int q = 0; q < BS; ++q) zz = fma(-L11[k0 + q], xs[q], zz);
          z = k < k0 ? zz : z;
        }
      }
      if (col && h == 0) {
        xg[wave * kTreePiv + k] = xout;
        st_coh(xp + (size_t)c0 * BS + k, xout);
      }
      if (release > 0) {   // tasks below the group (other groups, the per-task launch) wait for this front
        __builtin_amdgcn_s_waitcnt(0);   // the x stores have been acknowledged
        if (lane == 0) __hip_atomic_fetch_add(P.ready + f, release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
}

// Forward / backward step of the scratch-slab fronts of a level by SEVERAL workgroups per front.  One workgroup per front
// (front_forward_kernel / front_backward_kernel) pulls the whole m x npiv panel through one CU: ~55 GB/s, 14-16 us for
// the 2 000-row fronts at the top of a pose graph's tree, on the critical path of every level.  Here a workgroup owns 256
// boundary rows (chunk (front, first row, rows, ordinal | count << 16)):
//   forward : every chunk solves the pivot part redundantly (L11 in LDS, same operation order as the one-workgroup
//             kernel: results are bit-identical to it), then updates its own rows of the update vector;
//   backward: every chunk reduces its rows of L21' x_boundary to 64 partial sums (lanes along the rows: coalesced), the
//             chunk that delivers last adds the partials in chunk order and solves with L11' (hand-off as described above
//             ld_coh: agent-scope stores, vmcnt(0), barrier, one agent-scope increment).
template <int BS>
__global__ void __launch_bounds__(256) big_forward_kernel(CholPlanDev P, const int4* __restrict__ chunks, const double* __restrict__ bperm,
                                                         double* __restrict__ y) {
  __shared__ double L11[64 * 65], li[64], tJ[64], ys[64], tB[256];
  const int4 ck = chunks[blockIdx.x];
  const int f = ck.x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int npiv = rec.ns * BS, m = (rec.ns + rec.nb) * BS, c0 = rec.c0;
  const int r0 = ck.y, nrows = ck.z;
  const int tid = threadIdx.x;
  const double* Lg = P.L + rec.L_off;
  // Everything that depends on the front record only is requested before the first wait (one round trip): this
  // thread's row of the panel, the pivot block, the right-hand side and the update vectors of the first two children
  // (embedded in the record; up to 1 024 boundary rows each).
  double lv[64];
  {
    const double* Lr = Lg + npiv + r0 + min(tid, max(nrows - 1, 0));
#pragma unroll
    for (int k = 0; k < 64; ++k) lv[k] = (nrows > 0 && k < npiv) ? Lr[(size_t)m * k] : 0.0;
  }
  const int nch = rec.child_cnt;
  const int n0 = nch > 0 ? rec.ch[0].nbc * BS : 0, n1 = nch > 1 ? rec.ch[1].nbc * BS : 0;
  const bool fast = nch <= 2 && n0 <= 1024 && n1 <= 1024;
  double cw[2][4];
  int cd_[2][4];
  if (fast) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int nc = c == 0 ? n0 : n1;
      const double* wc = P.w + rec.ch[c].w_off;
      const int* rel = P.crel + rec.crel_off + rec.ch[c].crel_start;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u;
        const bool on = i < nc;
        cw[c][u] = on ? wc[i] : 0.0;
        cd_[c][u] = on ? rel[i / BS] * BS + i % BS : -1;
      }
    }
  }
  for (int c = tid >> 6; c < npiv; c += 4) {   // (column c by wave, rows by lane: no integer divisions)
    const int r = tid & 63;
    if (r < npiv) L11[r + 65 * c] = Lg[r + (size_t)m * c];
  }
  if (tid < npiv) {
    li[tid] = Lg[(size_t)m * npiv + tid];
    tJ[tid] = bperm[(size_t)c0 * BS + tid];
  }
  tB[tid] = 0.0;
  __syncthreads();
  // children one at a time (rows of two children may coincide), in child order like the one-workgroup kernel
  if (fast) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = cd_[c][u];
          if (d >= 0) {
            if (d < npiv) {
              tJ[d] += cw[c][u];
            } else {
              const int r = d - npiv - r0;
              if (r >= 0 && r < nrows) tB[r] += cw[c][u];
            }
          }
        }
        __syncthreads();
      }
    }
  } else {
    for (int ch = 0; ch < nch; ++ch) {
      const ChildDesc cd = P.cdesc[rec.child_off + ch];
      const int nbc = cd.nbc * BS;
      const double* wc = P.w + cd.w_off;
      const int* rel = P.crel + rec.crel_off + cd.crel_start;
      for (int i = tid; i < nbc; i += 256) {
        const int d = rel[i / BS] * BS + i % BS;
        if (d < npiv) {
          tJ[d] += wc[i];
        } else {
          const int r = d - npiv - r0;
          if (r >= 0 && r < nrows) tB[r] += wc[i];
        }
      }
      __syncthreads();
    }
  }
  for (int kb = 0; kb < rec.ns; ++kb) {
    const int k0 = kb * BS;
    double yv[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      double v = tJ[k0 + c];
#pragma unroll
      for (int q = 0; q < c; ++q) v -= L11[(k0 + c) + 65 * (k0 + q)] * yv[q];
      yv[c] = v * li[k0 + c];
    }
    if (tid == 0) {
#pragma unroll
      for (int c = 0; c < BS; ++c) ys[k0 + c] = yv[c];
    }
    const int j = k0 + BS + tid;
    if (j < npiv) {
      double v = tJ[j];
#pragma unroll
      for (int q = 0; q < BS; ++q) v -= L11[j + 65 * (k0 + q)] * yv[q];
      tJ[j] = v;
    }
    __syncthreads();
  }
  if (tid < nrows) {
    double v = tB[tid];
#pragma unroll
    for (int k = 0; k < 64; ++k)
      if (k < npiv) v -= lv[k] * ys[k];
    P.w[rec.w_off + r0 + tid] = v;
  }
  if ((ck.w & 0xffff) == 0 && tid < npiv) y[(size_t)c0 * BS + tid] = ys[tid];
}

// DEP: ONE launch for several consecutive levels (top level first; workgroups are dispatched in order, so the one a
// waiting workgroup needs is running or done -- no deadlock whatever the number resident): a chunk whose
// front's parent is in the same launch waits for the parent's flag -- with its panel, pivot block and records already
// loaded -- instead of for a launch of its own (per level ~15 us of start-up and dependent round trips -> the hand-off);
// the chunk that finishes a front raises that front's flag once its x stores have been acknowledged.
template <int BS, bool DEP>
__global__ void __launch_bounds__(256) big_backward_kernel(CholPlanDev P, const int4* __restrict__ chunks, const double* __restrict__ y,
                                                          double* __restrict__ xp, double* __restrict__ part, int* __restrict__ cnt,
                                                          int* __restrict__ flag) {
  __shared__ double L11[64 * 65], li[64], t[64], xs[64], xB[256];
  __shared__ int last_s;
  const int4 ck = chunks[blockIdx.x];
  const int f = ck.x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int npiv = rec.ns * BS, m = (rec.ns + rec.nb) * BS, c0 = rec.c0;
  const int r0 = ck.y, nrows = ck.z;
  const int g = DEP ? (ck.w & 0xff) : (ck.w & 0xffff), G = DEP ? ((ck.w >> 8) & 0xff) : (ck.w >> 16);
  const bool wait_parent = DEP && ((ck.w >> 16) & 1), raise_flag = DEP && ((ck.w >> 17) & 1);
  const double yk = (threadIdx.x < npiv) ? y[(size_t)c0 * BS + threadIdx.x] : 0.0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double* Lg = P.L + rec.L_off;
  // wave wv: columns wv, wv + 4, ...; lanes along the rows of the chunk (four rows per lane); all loads in flight first
  double lv[16][4];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = wv + 4 * kk;
    const double* Lk = Lg + (size_t)m * min(k, npiv - 1) + npiv + r0;
#pragma unroll
    for (int u = 0; u < 4; ++u) lv[kk][u] = (k < npiv && lane + 64 * u < nrows) ? Lk[lane + 64 * u] : 0.0;
  }
  int xrow = 0;
  {
    const int* rows = P.rows + rec.rows_off;
    const int r = r0 + tid;
    if (tid < nrows) xrow = rows[r / BS] * BS + r % BS;
  }
  for (int c = tid >> 6; c < npiv; c += 4) {   // (column c by wave, rows by lane: no integer divisions)
    const int r = tid & 63;
    if (r < npiv) L11[r + 65 * c] = Lg[r + (size_t)m * c];
  }
  if (tid < npiv) li[tid] = Lg[(size_t)m * npiv + tid];
  if (wait_parent) {   // the boundary values come from fronts of this launch: the parent's flag covers all ancestors
    if (tid == 0) {
      const int* fl = flag + (rec.pad[1] & 0x00ffffff);
      int spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P.dep_spin_limit || ((spins & 255) == 0 && __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
          __hip_atomic_store(P.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  xB[tid] = (tid < nrows) ? ld_coh(xp + (size_t)xrow) : 0.0;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = wv + 4 * kk;
    double s = (lv[kk][0] * xB[lane] + lv[kk][1] * xB[lane + 64]) + (lv[kk][2] * xB[lane + 128] + lv[kk][3] * xB[lane + 192]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0 && k < npiv) {
      if (G > 1) st_coh(part + (size_t)blockIdx.x * 64 + k, s);
      else xs[k] = s;   // (a front of one chunk: no hand-off)
    }
  }
  if (G > 1) {
    __builtin_amdgcn_s_waitcnt(0);   // the partial sums have been acknowledged at the device-coherent level
    __syncthreads();
    if (tid == 0) {
      const int prev = __hip_atomic_fetch_add(cnt + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_s = prev == G - 1;
      if (prev == G - 1) __hip_atomic_store(cnt + f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next solve)
    }
    __syncthreads();
    if (!last_s) return;
    if (tid < npiv) {
      double a = 0.0;
      const double* p0 = part + (size_t)(blockIdx.x - g) * 64 + tid;
      for (int gg = 0; gg < G; ++gg) a += ld_coh(p0 + (size_t)gg * 64);
      t[tid] = yk - a;
    }
  } else {
    __syncthreads();
    if (tid < npiv) t[tid] = yk - xs[tid];
  }
  __syncthreads();
  for (int kb = rec.ns - 1; kb >= 0; --kb) {
    const int k0 = kb * BS;
    double xv[BS];
#pragma unroll
    for (int c = BS - 1; c >= 0; --c) {
      double v = t[k0 + c];
#pragma unroll
      for (int q = c + 1; q < BS; ++q) v -= L11[(k0 + q) + 65 * (k0 + c)] * xv[q];
      xv[c] = v * li[k0 + c];
    }
    if (tid == 0) {
#pragma unroll
      for (int c = 0; c < BS; ++c) xs[k0 + c] = xv[c];
    }
    if (tid < k0) {
      double v = t[tid];
#pragma unroll
      for (int q = 0; q < BS; ++q) v -= L11[(k0 + q) + 65 * tid] * xv[q];
      t[tid] = v;
    }
    __syncthreads();
  }
  if (tid < npiv) st_coh(xp + (size_t)c0 * BS + tid, xs[tid]);
  if (raise_flag) {
    __builtin_amdgcn_s_waitcnt(0);   // the x stores have been acknowledged at the device-coherent level
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// exchange segments: dir 0 = pack (own subtree roots -> buffer), dir 1 = unpack (foreign roots <- buffer)
struct SegCopyDev { long long a, b; int n, flags; };
__global__ void seg_copy_kernel(const SegCopyDev* __restrict__ segs, double* __restrict__ U, double* __restrict__ w,
                                double* __restrict__ buf, int dir) {
  const SegCopyDev sc = segs[blockIdx.x];
  const bool mine = sc.flags & 1;
  double* arr = (sc.flags & 2) ? w : U;
  if (dir == 0) {
    if (!mine) return;
    for (int i = threadIdx.x; i < sc.n; i += blockDim.x) buf[sc.b + i] = arr[sc.a + i];
  } else {
    if (mine) return;
    for (int i = threadIdx.x; i < sc.n; i += blockDim.x) arr[sc.a + i] = buf[sc.b + i];
  }
}
__global__ void mask_kernel(size_t n, const double* __restrict__ mask, double* __restrict__ x) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) x[i] *= mask[i];
}

#include "wave_front.inc"
#include "band_chain.inc"

// Pivot block of a scratch-slab front (n <= 64 columns) on the matrix cores: the blocked right-looking Cholesky of
// wave_front_kernel restricted to the pivot block -- the symmetric n x n block as ten upper 16 x 16 tiles in accumulator
// layout dealt to four waves, four columns per step (row panel by symmetry, 4 x 4 pivot block factorised by every lane,
// scaled panel and rank-4 update one MFMA each).  The one-wave kernel (big_diag_kernel: a row per lane, columns broadcast
// by v_readlane) needs n^2 / 2 readlane + FMA pairs in ONE instruction stream: 23 us for 48 columns, on the critical path
// of every level of a pose graph; this one takes about half of that.
// FWD: the forward step of fronts WITHOUT boundary rows (roots: no panel tiles) rides along; the others: big_panel_kernel.
// Right-hand side and the children's update vectors are requested before the factorisation, y = L11^-1 t is solved from an
// LDS copy of the finished pivot block with the operation order of front_forward_kernel.
// COH (big_level_kernel): L11 and the reciprocal diagonal go to memory with device-coherent stores -- the panel tiles of the
// same launch read them.  lds: 512 doubles (FWD: 4 800).
// WINV (big_panel_solve_kernel: every workgroup of the panel solve factorises the pivot block itself): L11 stays in LDS (L11s, lis), the
// inverses of its diagonal 16 x 16 tiles follow (Ws), and only the workgroup with `writer` stores L11 and the reciprocal diagonal to L --
// the frontal matrix keeps the RAW pivot block, the other workgroups of the launch are still reading it.
template <int BS, bool FWD, bool COH, bool GATH = false, bool WINV = false>
__device__ __forceinline__ void big_diag_mfma_body(const CholPlanDev& P, const int slot, double* __restrict__ scratch,
                                                   const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld,
                                                   const double* __restrict__ bperm, double* __restrict__ yout, double* lds, bool writer = true) {
  double* Rb = lds;
  double* Lb = lds + 256;
  double* L11s = lds + 512;   // (FWD only)
  double* lis = L11s + 64 * 65;
  double* tJ = lis + 64;
  constexpr int T = 4, NT = 10;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = P.slots[slot].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int m = (rec.ns + rec.nb) * BS, n = rec.ns * BS;
  const int ld = scratch_ld[slot];
  double* F = scratch + scratch_off[slot];
  double* Lg = P.L + rec.L_off;
  const int lr = lane & 15, lk = lane >> 4;
  FwdChildPre pre;
  double bJ = 0.0;
  const bool fwd_here = FWD && rec.nb == 0;   // (fronts with boundary rows: big_panel_kernel)
  if (fwd_here) {
    fwd_children_prefetch<BS>(P, rec, tid, pre);
    if (tid < n) bJ = bperm[(size_t)rec.c0 * BS + tid];
  }
  // ONE wave factorises the pivot block (n <= 64: the ten upper 16 x 16 tiles in its registers), the others go on to the
  // barrier: no LDS, no barrier inside a 4-column step -- the 4 x 4 pivot block comes through v_readlane, the scaled panel
  // rows are register 0 of an MFMA and at once the operand of the update (the k-block of band_chain.inc; four waves with
  // two barriers per step took 1.25 us per step, this one 0.5).  Tile (ti, tj), ti <= tj, is S[tj (tj + 1) / 2 + ti].
  bool bad = false;
  if (GATH) {   // the pivot block with the children's entries added, by the whole workgroup, into LDS (the later L11 copy's place)
    const GatherCtx gc = gather_stage(P, slot, rec.ns + rec.nb, reinterpret_cast<int*>(lds + kGatherLdsOff));
    __syncthreads();
    const int gnch = gc.children();
    constexpr int UG = 16;   // 64 * 64 / 256
    double t[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int r = tid & 63, c = (tid >> 6) + 4 * u;
      t[u] = (r < n && c <= r) ? F[(size_t)r + (size_t)ld * c] : 0.0;
    }
    for (int ch = 0; ch < gnch; ++ch) {
      const double* Uc = gc.child_U(ch);
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const int r = min(tid & 63, n - 1), c = min((tid >> 6) + 4 * u, n - 1);
        const bool in = (tid & 63) < n && (tid >> 6) + 4 * u <= (tid & 63);
        const double g_ = gather_child<BS>(gc, ch, Uc, t[u], r / BS, in ? c / BS : r / BS, (r % BS) + BS * (c % BS));
        t[u] = in ? g_ : t[u];
      }
    }
#pragma unroll
    for (int u = 0; u < UG; ++u) L11s[(tid & 63) + 65 * ((tid >> 6) + 4 * u)] = t[u];
    __syncthreads();
  }
  if (w == 0) {
    wv_d4 S[NT];
#pragma unroll
    for (int tj = 0; tj < T; ++tj)
#pragma unroll
      for (int ti = 0; ti <= tj; ++ti)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * ti + lk + 4 * v, c = 16 * tj + lr;
          const int a = min(r, c), b = max(r, c);                       // F holds the lower triangle
          S[tj * (tj + 1) / 2 + ti][v] = (b < n) ? (GATH ? L11s[b + 65 * a] : F[(size_t)b + (size_t)ld * a]) : 0.0;
        }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const int k0 = 4 * kb;
      if (k0 < n) {   // (uniform)
        const int tk = k0 >> 4, vk = (k0 & 15) >> 2, c16 = k0 & 15;
        const int qd = tk * (tk + 1) / 2 + tk;
        // 4 x 4 pivot block (uniform values): D = Ld Ld', W = Ld^-1
        double D[4][4], Ld[4][4], rs[4], sq[4], W[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = a; b < 4; ++b) D[a][b] = readlane_f64(S[qd][vk], (c16 + b) + 16 * a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k0 + j < n) {
            double d = D[j][j];
            if (!(d > 0.0)) {
              bad = true;
              d = 1.0;
            }
            sqrt_and_rsqrt(d, sq[j], rs[j]);
#pragma unroll
            for (int i = j + 1; i < 4; ++i) Ld[i][j] = D[j][i] * rs[j];
#pragma unroll
            for (int c = j + 1; c < 4; ++c)
#pragma unroll
              for (int i = c; i < 4; ++i) D[c][i] -= Ld[i][j] * Ld[c][j];
          } else {
            sq[j] = 0.0;
            rs[j] = 0.0;
#pragma unroll
            for (int i = j + 1; i < 4; ++i) Ld[i][j] = 0.0;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          W[j][j] = rs[j];
#pragma unroll
          for (int i = j + 1; i < 4; ++i) {
            double s_ = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s_ += Ld[i][k] * W[k][j];
            W[i][j] = -rs[i] * s_;
          }
        }
        double aop = 0.0, sqsel = 0.0, rssel = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k <= i; ++k)
            if (lr == i && lk == k) aop = W[i][k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lk == k) {
            sqsel = sq[k];
            rssel = rs[k];
          }
        const int col = k0 + lk;
        if (lr == 0 && col < n) {
          if (COH) st_coh(Lg + (size_t)m * n + col, rssel);
          else if (!WINV || writer) Lg[(size_t)m * n + col] = rssel;
          if (WINV) lis[col] = rssel;
        }
        double Lp[T];   // scaled panel rows of the tile columns: operand layout of the update
#pragma unroll
        for (int t = 0; t < T; ++t) {
          Lp[t] = 0.0;
          if (16 * t < n) {
            const int row = 16 * t + lr;
            double v = 0.0;                                                     // (above the diagonal: zeros)
            if (t >= tk) {
              const wv_d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, S[t * (t + 1) / 2 + tk][vk], wv_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
              Lp[t] = r[0];
              v = r[0];
              if (t == tk) v = row > col ? v : (row == col ? sqsel : 0.0);
            }
            if (col < n && row < n) {
              if (COH) st_coh(Lg + (size_t)row + (size_t)m * col, v);
              else if (!WINV || writer) Lg[(size_t)row + (size_t)m * col] = v;
              if (!WINV) F[(size_t)row + (size_t)ld * col] = v;
              if (WINV) L11s[row + 65 * col] = v;
            }
          }
        }
#pragma unroll
        for (int tj = 0; tj < T; ++tj)
#pragma unroll
          for (int ti = 0; ti <= tj; ++ti)
            if (ti >= tk && 16 * tj < n)
              S[tj * (tj + 1) / 2 + ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv_neg(Lp[ti]), Lp[tj], S[tj * (tj + 1) / 2 + ti], 0, 0, 0);
      }
    }
  }
  (void)Rb;
  (void)Lb;
  if (bad && lane == 0) atomicMax(P.status, 1);
  if (WINV) {
    // The panel solve runs on the matrix cores, 16 columns at a time: it needs the INVERSES of the diagonal 16 x 16 tiles of L11
    // (triangular; the conditioning of a 16 x 16 tile only).  Wave t inverts tile t from the LDS copy -- lane j the column j, rows in
    // sequence, the multipliers broadcast -- into Ws[t][row][column] (zeros above the diagonal and beyond n).
    double* Ws = tJ + 64;
    __syncthreads();
    const int base = 16 * w;
    if (lane < 16) {
      const int j = lane;
      // (branch-free: every load unconditional -- rows beyond n read what the LDS holds and are dropped by the selects)
      double wv[16], ri[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double r_ = lis[base + i];
        ri[i] = base + i < n ? r_ : 0.0;
        wv[i] = i == j ? ri[i] : 0.0;
      }
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        double lrow[15];
#pragma unroll
        for (int k = 0; k < i; ++k) lrow[k] = L11s[(base + i) + 65 * (base + k)];
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
          if (k & 1) s1 = fma(lrow[k], wv[k], s1);
          else s0 = fma(lrow[k], wv[k], s0);
        }
        const double nv = -ri[i] * (s0 + s1);
        wv[i] = (i > j && base + i < n) ? nv : wv[i];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Ws[w * 256 + i * 16 + j] = wv[i];
    }
  }
  if (fwd_here) {
    if (tid < 64) tJ[tid] = bJ;
    __syncthreads();   // (also: the pivot block written above is visible to the whole workgroup)
    fwd_children_apply<BS>(P, rec, tid, pre, tJ, 0, n);
    for (int i = tid; i < n * n; i += 256) {
      const int r = i % n, c = i / n;
      L11s[r + 65 * c] = Lg[(size_t)r + (size_t)m * c];
    }
    if (tid < n) lis[tid] = Lg[(size_t)m * n + tid];
    __syncthreads();
    for (int kb = 0; kb < rec.ns; ++kb) {
      const int k0 = kb * BS;
      double yv[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        double v = tJ[k0 + c];
#pragma unroll
        for (int q = 0; q < c; ++q) v -= L11s[(k0 + c) + 65 * (k0 + q)] * yv[q];
        yv[c] = v * lis[k0 + c];
      }
      if (tid == 0) {
#pragma unroll
        for (int c = 0; c < BS; ++c) yout[(size_t)rec.c0 * BS + k0 + c] = yv[c];
      }
      const int j = k0 + BS + tid;
      if (j < n) {
        double v = tJ[j];
#pragma unroll
        for (int q = 0; q < BS; ++q) v -= L11s[j + 65 * (k0 + q)] * yv[q];
        tJ[j] = v;
      }
      __syncthreads();
    }
  }
}

template <int BS, bool FWD>
__global__ void __launch_bounds__(256) big_diag_mfma_kernel(CholPlanDev P, int slot0, double* __restrict__ scratch,
                                                           const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld,
                                                           const double* __restrict__ bperm, double* __restrict__ yout) {
  __shared__ double lds[FWD ? 4800 : 512];
  big_diag_mfma_body<BS, FWD, false>(P, slot0 + blockIdx.x, scratch, scratch_off, scratch_ld, bperm, yout, lds);
}

// Pivot block AND panel rows of the scratch-slab fronts of a level in one launch, the rows on the matrix cores (big_diag_mfma_kernel
// + big_trsm_kernel: two launches of 16 and 23 us on every level of a large separator's chain -- one wave's factorisation, then one
// thread per row with n^2 / 2 dependent multiply-adds; both mostly start-up and latency).  A workgroup owns 64 panel rows (a quarter
// of a chunk of big_trsm_kernel's list) and factorises the pivot block ITSELF (a few dozen workgroups repeat one wave's 8 us next to
// each other instead of queueing behind it); the first workgroup of a front writes L11 to L.  Then x L11' = row for 16 rows per wave:
// the rows are loaded TRANSPOSED into the accumulator layout D[k][row] (lane = row: coalesced) and block forward substitution over the
// 16-column tiles t of the panel runs as MFMAs whose B operand is the accumulator of an earlier tile -- the layouts chain (D[4 v + q][j]
// in register v of lane j + 16 q is B[j][k = q] of k-step v):
//     -Q_t = -R_t + sum_{u < t} L11(t, u) X_u        X_t = (-W_t) (-Q_t),   W_t = L11(t, t)^-1.
// The solved rows go to L only (big_front_update_kernel reads them there); the frontal matrix keeps the raw panel.
// Needs every front of the launch to have boundary rows (LevelLaunch::tr_all): a front without has no workgroup here.
template <int BS>
__global__ void __launch_bounds__(256) big_panel_solve_kernel(CholPlanDev P, const int4* __restrict__ chunks, double* __restrict__ scratch,
                                                             const long long* __restrict__ scratch_off, const int* __restrict__ scratch_ld) {
  __shared__ double lds[4800 + 4 * 256];
  const int4 ck = chunks[blockIdx.x >> 2];   // x: launch slot, y: first row (relative to the pivot block's end)
  const int sub = blockIdx.x & 3;
  const int f = P.slots[ck.x].x;
  const FrontRec rec = load_front_rec(P.rec + f);
  const int n = rec.ns * BS, mt = rec.nb * BS, m = n + mt;
  if (ck.y + 64 * sub >= mt) return;   // (whole workgroup)
  const int ld = scratch_ld[ck.x];
  const double* F = scratch + scratch_off[ck.x];
  double* Lg = P.L + rec.L_off;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
  const int row = ck.y + 64 * sub + 16 * wave + lr;
  // this wave's 16 rows, requested before the pivot block
  mfma_d4 X[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = 16 * t + 4 * v + lk;
      const double x = F[(size_t)(n + min(row, mt - 1)) + (size_t)ld * min(k, n - 1)];
      X[t][v] = (row < mt && k < n) ? -x : 0.0;
    }
  big_diag_mfma_body<BS, false, false, false, true>(P, ck.x, scratch, scratch_off, scratch_ld, nullptr, nullptr, lds, ck.y == 0 && sub == 0);
  __syncthreads();
  const double* L11s = lds + 512;
  const double* Ws = lds + 4800;
  const int ntl = (n + 15) >> 4;   // 16-column tiles of the panel (1..4)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntl) {   // (uniform)
      const int i = 16 * t + lr;
      double Lop[3][4], Wop[4];
#pragma unroll
      for (int u = 0; u < t; ++u)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int k = 16 * u + 4 * s_ + lk;
          const double x = L11s[i + 65 * k];
          Lop[u][s_] = i < n ? x : 0.0;   // (k < 16 t <= n - 1)
        }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) Wop[s_] = -Ws[t * 256 + lr * 16 + 4 * s_ + lk];
#pragma unroll
      for (int u = 0; u < t; ++u)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) X[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[u][s_], X[u][s_], X[t], 0, 0, 0);
      mfma_d4 y0 = mfma_d4{0.0, 0.0, 0.0, 0.0}, y1 = mfma_d4{0.0, 0.0, 0.0, 0.0};
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wop[0], X[t][0], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wop[1], X[t][1], y1, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wop[2], X[t][2], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wop[3], X[t][3], y1, 0, 0, 0);
      X[t] = y0 + y1;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = 16 * t + 4 * v + lk;
        if (row < mt && k < n) Lg[(size_t)(n + row) + (size_t)m * k] = X[t][v];
      }
    }
  }
}

// Pivot blocks and panel tiles of the scratch-slab fronts of a level in ONE launch: workgroups [0, ndiag) factorise the pivot
// blocks and raise their front's flag, the others are the tiles of big_panel_kernel and wait for it -- their start-up (tile
// and front records, panel rows, children's vectors: three to four dependent round trips) runs next to the pivot block
// instead of behind it.  The pivot-block workgroups come first in dispatch order: no deadlock.
template <int BS, bool FWD, bool GATH>
__global__ void __launch_bounds__(256) big_level_kernel(CholPlanDev P, int slot0, int ndiag, const int4* __restrict__ tiles,
                                                       double* __restrict__ scratch, const long long* __restrict__ scratch_off,
                                                       const int* __restrict__ scratch_ld, const double* __restrict__ bperm,
                                                       double* __restrict__ yout, int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) double psm[];
  if ((int)blockIdx.x < ndiag) {
    const int slot = slot0 + blockIdx.x;
    big_diag_mfma_body<BS, FWD, true, GATH>(P, slot, scratch, scratch_off, scratch_ld, bperm, yout, psm);
    __builtin_amdgcn_s_waitcnt(0);   // L11 and the reciprocal diagonal have been acknowledged at the device-coherent level
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag + P.slots[slot].x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    big_panel_body<BS, FWD, true, GATH>(P, tiles[blockIdx.x - ndiag], scratch, scratch_off, scratch_ld, bperm, yout, psm, flag);
  }
}

struct BigLaunch {   // whole-GPU passes over the scratch-slab fronts of one level (LevelLaunch::ba_* / be_pass / tr_*)
  bool ok;
  const int4* chunks;
  int ba_begin, ba_count, tr_begin, tr_count;
  const std::vector<std::pair<int, int>>* be_pass;
  int fz_begin, fz_count;
  bool hoisted = false;   // fill + assembly already done by the phase-wide passes
  bool fwd = false;       // the forward step rides along in the pivot-block and panel kernels (big_forward_carried)
  int* flag = nullptr;    // non-null: pivot blocks and panel tiles of the level in one launch (big_level_kernel), per-front flags
  const int* ld;   // leading dimension per launch slot
  bool fuse_panel;   // big_panel_kernel instead of big_trsm_kernel + big_front_update_kernel
  bool mfma_diag;    // big_diag_mfma_kernel instead of big_diag_kernel
  int merge_tiles = 256;   // the fused panel kernel on levels of at most this many tiles (CholOptions::big_merge_tiles)
  bool gather = false;     // the merged level launch gathers the children's update matrices itself: no extend-add passes (LevelLaunch::gather)
  int eg_begin = 0, eg_count = 0;   // the level's extend-add in one launch (big_extend_gather_kernel); 0: the passes per child ordinal
  bool eg_write = false;            // ... into regions that were not zero-filled (LevelLaunch::eg_write), the original blocks la_* behind it
  int eg_maxc = 7;                  // ... children per front of the level at most
  int la_begin = 0, la_count = 0;
  bool panel_solve = false;   // pivot blocks + panel rows of the level in one launch (big_panel_solve_kernel): LevelLaunch::tr_all and not a merged / fused level
};

__global__ void __launch_bounds__(256) fill_zero_kernel(double* __restrict__ p, size_t n) {
  const size_t n2 = n / 2, stride = (size_t)gridDim.x * blockDim.x;
  double2* p2 = reinterpret_cast<double2*>(p);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += stride) p2[i] = make_double2(0.0, 0.0);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}

// a launch the runtime refuses (grid / LDS beyond the limits) is reported with the kernel's name
#define G2OHIP_LAUNCH_CHECK(name_)                                                                              \
  do {                                                                                                            \
    const hipError_t e_ = hipGetLastError();                                                                      \
    if (e_ != hipSuccess) throw StateFailure(std::string("launch of ") + name_ + " refused: " + hipGetErrorString(e_)); \
  } while (0)
template <int BS, bool VIRT>
void launch_factor_level(const CholPlanDev& P, const int* d_tasks, const long long* d_scratch_off, double* d_scratch,
                         const double* dA, int lds_begin, int lds_count, int lds_max_m, int glb_begin, int glb_count,
                         int lds_idx_ints, int glb_idx_ints, int sm_count, int sm_max_m, int sm_idx_ints, int wcap,
                         const double* bperm, double* yout, int dep, const int4* big_tiles, int bt_count, const BigLaunch& big,
                         int wide_doubles, bool wv, int wv_pn, int wv_idx_ints, hipStream_t st, int parts = 3) {
  // parts: bit 0 = the fronts held in LDS / registers, bit 1 = the scratch-slab fronts (the two halves of a level are
  // independent: factor_phase may put them on different streams)
  if (!(parts & 1)) {
    lds_begin += sm_count;
    lds_count = 0;
    sm_count = 0;
  } else if (wv) {   // every front of the launch fits the register-resident wave kernel: one wavefront per task
    const size_t sh = ((size_t)kWvTiles * 256 + 64 + 2 * kWvT * 64) * sizeof(double) + (size_t)(8 * 64 + 2 * 16 * kWvT + 2 * 64) * sizeof(int);
    hipLaunchKernelGGL((wave_front_kernel<BS, VIRT>), dim3(lds_count), dim3(64 * kWvWaves), sh, st, wv_plan(P), lds_begin, dA, bperm, yout, dep);
    G2OHIP_LAUNCH_CHECK("wave_front_kernel");
    return;
  }
  if (sm_count > 0) {   // wide launch: two waves per front
    const int idx_off = sm_max_m + 2 * wcap + 2 * (BS * BS + BS);
    size_t sh = (size_t)idx_off * sizeof(double) + (size_t)(sm_idx_ints + 4) * sizeof(int);
    hipLaunchKernelGGL((front_factor_kernel<BS, true, 128, VIRT>), dim3(sm_count), dim3(128), sh, st, P, lds_begin, dA, d_scratch,
                       d_scratch_off + lds_begin, idx_off, wcap, bperm, yout, dep);
    G2OHIP_LAUNCH_CHECK("front_factor_kernel");
    lds_begin += sm_count;
    lds_count -= sm_count;
  }
  if (lds_count > 0) {
    const int idx_off = lds_max_m + 2 * wcap + 2 * (BS * BS + BS);   // F (packed doubles) | tv | wprev | mailboxes | index lists
    size_t sh = (size_t)idx_off * sizeof(double) + (size_t)(lds_idx_ints + 4) * sizeof(int);
    // fronts of a hundred rows and more (pose graphs; they keep a CU to themselves anyway): eight waves per front
    if (lds_max_m >= wide_doubles)
      hipLaunchKernelGGL((front_factor_kernel<BS, true, 512, VIRT>), dim3(lds_count), dim3(512), sh, st, P, lds_begin, dA, d_scratch,
                         d_scratch_off + lds_begin, idx_off, wcap, bperm, yout, dep);
    else
      hipLaunchKernelGGL((front_factor_kernel<BS, true, kFactorThreads, VIRT>), dim3(lds_count), dim3(kFactorThreads), sh, st, P, lds_begin, dA, d_scratch,
                         d_scratch_off + lds_begin, idx_off, wcap, bperm, yout, dep);
    if (hipPeekAtLastError() != hipSuccess)
      fprintf(stderr, "g2ohip: LDS front launch: %zu bytes of LDS (blocks %d doubles, vectors %d, index tables %d ints)\n", sh, lds_max_m, wcap,
              lds_idx_ints);
    G2OHIP_LAUNCH_CHECK("front_factor_kernel (LDS fronts)");
  }
  if (!(parts & 2)) return;
  if (glb_count > 0 && big.ok) {   // large fronts as whole-GPU passes
    if (big.fz_count > 0 && !big.hoisted) {   // zero the regions that start at this level (a kernel: hipMemsetAsync reaches ~1 TB/s only)
      hipLaunchKernelGGL(big_fill_kernel, dim3(big.fz_count), dim3(256), 0, st, big.chunks + big.fz_begin, d_scratch, d_scratch_off, big.ld);
      G2OHIP_LAUNCH_CHECK("big_fill_kernel");
    }
    if (big.ba_count > 0 && !big.hoisted)
      hipLaunchKernelGGL((big_assemble_kernel<BS, VIRT>), dim3(big.ba_count), dim3(256), 0, st, P, big.chunks + big.ba_begin, dA, d_scratch,
                         d_scratch_off, big.ld);
    G2OHIP_LAUNCH_CHECK("big_assemble_kernel");
    const bool level_launch = big.flag && big.mfma_diag && big.fuse_panel && bt_count > 0 && bt_count <= big.merge_tiles;
    bool any_pass = false;
    for (const auto& pass : *big.be_pass) any_pass = any_pass || pass.second > 0;
    const bool gather = level_launch && big.gather && any_pass;   // (the level's launch adds the children's update matrices where it loads the fronts)
    if (big.eg_write) {   // every child ordinal in one launch, into regions that were not zero-filled; then the original blocks
      if (big.eg_maxc <= 2)
        hipLaunchKernelGGL((big_extend_gather_kernel<BS, true, 2>), dim3(big.eg_count), dim3(kEgThreads), 0, st, P, big.chunks + big.eg_begin, d_scratch, d_scratch_off);
      else
        hipLaunchKernelGGL((big_extend_gather_kernel<BS, true, 7>), dim3(big.eg_count), dim3(kEgThreads), 0, st, P, big.chunks + big.eg_begin, d_scratch, d_scratch_off);
      G2OHIP_LAUNCH_CHECK("big_extend_gather_kernel");
      if (big.la_count > 0)
        hipLaunchKernelGGL((big_assemble_kernel<BS, VIRT>), dim3(big.la_count), dim3(256), 0, st, P, big.chunks + big.la_begin, dA, d_scratch,
                           d_scratch_off, big.ld);
      G2OHIP_LAUNCH_CHECK("big_assemble_kernel");
    } else if (!gather && big.eg_count > 0) {   // every child ordinal in one launch
      if (big.eg_maxc <= 2)
        hipLaunchKernelGGL((big_extend_gather_kernel<BS, false, 2>), dim3(big.eg_count), dim3(kEgThreads), 0, st, P, big.chunks + big.eg_begin, d_scratch, d_scratch_off);
      else
        hipLaunchKernelGGL((big_extend_gather_kernel<BS, false, 7>), dim3(big.eg_count), dim3(kEgThreads), 0, st, P, big.chunks + big.eg_begin, d_scratch, d_scratch_off);
      G2OHIP_LAUNCH_CHECK("big_extend_gather_kernel");
    } else if (!gather)
      for (const auto& pass : *big.be_pass)
        if (pass.second > 0)
          hipLaunchKernelGGL((big_extend_add_kernel<BS>), dim3(pass.second), dim3(256), 0, st, P, big.chunks + pass.first, d_scratch,
                             d_scratch_off);
    G2OHIP_LAUNCH_CHECK("big_extend_add_kernel");
    if (level_launch) {
      const size_t shp = (size_t)(3 * 64 * 65 + 64 + 128) * sizeof(double) + (gather ? kGatherInts * sizeof(int) : 0);   // (the pivot-block role needs 4 800 doubles of it)
#define G2OHIP_BIG_LEVEL(FW_, GA_, B_, Y_)                                                                                                     \
  hipLaunchKernelGGL((big_level_kernel<BS, FW_, GA_>), dim3(glb_count + bt_count), dim3(256), shp, st, P, glb_begin, glb_count, big_tiles, d_scratch, \
                     d_scratch_off, big.ld, B_, Y_, big.flag)
      if (big.fwd && gather) G2OHIP_BIG_LEVEL(true, true, bperm, yout);
      else if (big.fwd) G2OHIP_BIG_LEVEL(true, false, bperm, yout);
      else if (gather) G2OHIP_BIG_LEVEL(false, true, (const double*)nullptr, (double*)nullptr);
      else G2OHIP_BIG_LEVEL(false, false, (const double*)nullptr, (double*)nullptr);
#undef G2OHIP_BIG_LEVEL
      G2OHIP_LAUNCH_CHECK("big_level_kernel");
      return;
    }
    // (at most kPanelSolveWgs workgroups: each repeats the pivot block on one wave and holds a CU's registers meanwhile -- a level of many
    // fronts is faster with the two separate launches; profiles/r6_grid_sweep.txt)
    constexpr int ps_max = kPanelSolveWgs;
    if (big.mfma_diag && !big.fwd && big.panel_solve && big.tr_count > 0 && 4 * big.tr_count <= ps_max && !(big.fuse_panel && bt_count <= big.merge_tiles)) {
      // pivot blocks + panel rows in one launch (every front of the level has boundary rows: BigLaunch::panel_solve)
      hipLaunchKernelGGL((big_panel_solve_kernel<BS>), dim3(4 * big.tr_count), dim3(256), 0, st, P, big.chunks + big.tr_begin, d_scratch,
                         d_scratch_off, big.ld);
      G2OHIP_LAUNCH_CHECK("big_panel_solve_kernel");
    } else {
      if (big.mfma_diag && big.fwd)
        hipLaunchKernelGGL((big_diag_mfma_kernel<BS, true>), dim3(glb_count), dim3(256), 0, st, P, glb_begin, d_scratch, d_scratch_off, big.ld, bperm, yout);
      else if (big.mfma_diag)
        hipLaunchKernelGGL((big_diag_mfma_kernel<BS, false>), dim3(glb_count), dim3(256), 0, st, P, glb_begin, d_scratch, d_scratch_off, big.ld,
                           (const double*)nullptr, (double*)nullptr);
      else
        hipLaunchKernelGGL((big_diag_kernel<BS>), dim3(glb_count), dim3(64), 0, st, P, glb_begin, d_scratch, d_scratch_off, big.ld);
      G2OHIP_LAUNCH_CHECK("big_diag_kernel");
      if (big.fuse_panel && bt_count <= big.merge_tiles) {   // panel solve + update in one launch -- while the level is a latency chain (at most one
                                                 // workgroup per CU); a level that fills the GPU pays for the rows solved more than once
        if (bt_count > 0) {
          const size_t shp = (size_t)(3 * 64 * 65 + 64 + 128) * sizeof(double);
          if (big.fwd)
            hipLaunchKernelGGL((big_panel_kernel<BS, true>), dim3(bt_count), dim3(256), shp, st, P, big_tiles, d_scratch, d_scratch_off, big.ld, bperm, yout);
          else
            hipLaunchKernelGGL((big_panel_kernel<BS, false>), dim3(bt_count), dim3(256), shp, st, P, big_tiles, d_scratch, d_scratch_off, big.ld,
                               (const double*)nullptr, (double*)nullptr);
          G2OHIP_LAUNCH_CHECK("big_panel_kernel");
        }
        return;
      }
      if (big.tr_count > 0)
        hipLaunchKernelGGL((big_trsm_kernel<BS>), dim3(big.tr_count), dim3(256), 0, st, P, big.chunks + big.tr_begin, d_scratch,
                           d_scratch_off, big.ld);
      G2OHIP_LAUNCH_CHECK("big_trsm_kernel");
    }
    if (bt_count > 0)
    {
      // (a launch that fills the GPU several times over is served better by four waves per SIMD than by one round trip per panel:
      // 116 against 180 registers; 16.5 -> 16.3 / 80.8 -> 79.4 ms on the grid graphs, profiles/r6_grid_sweep.txt)
      constexpr int ks_split = 4096;
      if (bt_count > ks_split) hipLaunchKernelGGL((big_front_update_kernel<BS, 6>), dim3(bt_count), dim3(256), 0, st, P, big_tiles, d_scratch, d_scratch_off, big.ld);
      else hipLaunchKernelGGL((big_front_update_kernel<BS, 15>), dim3(bt_count), dim3(256), 0, st, P, big_tiles, d_scratch, d_scratch_off, big.ld);
    }
    G2OHIP_LAUNCH_CHECK("big_front_update_kernel");
  } else if (glb_count > 0) {
    const int idx_off = 2 * (BS * BS + BS);
    size_t sh = (size_t)idx_off * sizeof(double) + (size_t)(glb_idx_ints + 4) * sizeof(int);
    hipLaunchKernelGGL((front_factor_kernel<BS, false, kFactorThreadsGlobal, VIRT>), dim3(glb_count), dim3(kFactorThreadsGlobal), sh, st, P, glb_begin, dA,
                       d_scratch, d_scratch_off + glb_begin, idx_off, 0, (const double*)nullptr, (double*)nullptr, 0);
    G2OHIP_LAUNCH_CHECK("front_factor_kernel");
    if (bt_count > 0)   // their trailing matrices: one MFMA pass over all of them
      hipLaunchKernelGGL((big_front_update_kernel<BS>), dim3(bt_count), dim3(256), 0, st, P, big_tiles, d_scratch, d_scratch_off, big.ld);
    G2OHIP_LAUNCH_CHECK("big_front_update_kernel");
  }
}

}  // namespace

void SparseCholesky::set_virtual_blocks(const VirtualBlocks& vb, hipStream_t st) {
  if (!analyzed_) throw StateFailure("SparseCholesky::set_virtual_blocks before analyze");
  const size_t n = sym_.asm_q.size();
  std::vector<int> vq(n), vpos(n), v(n * kVirtInts);
  for (size_t e = 0; e < n; ++e) {
    const int q = sym_.asm_q[e];
    vq[e] = vb.base_idx[q];
    vpos[e] = sym_.asm_pos[e] | (vb.is_diag[q] >= 0 ? (int)0x80000000u : 0);
    const int k0 = vb.part_ptr[q], cnt = vb.part_ptr[q + 1] - k0;
    int* r = &v[e * kVirtInts];
    r[0] = cnt;
    for (int j = 0; j < 3; ++j) r[1 + j] = j < cnt ? vb.part_slot[k0 + j] : vb.zero_slot;
    r[4] = k0;
  }
  if (n == 0) {
    vq.push_back(-1);
    vpos.push_back(0);
    v.resize(kVirtInts, 0);
  }
  {
    std::vector<int> r8(vq.size() * 8, 0);
    for (size_t e = 0; e < vq.size(); ++e) {
      int* r = &r8[e * 8];
      r[0] = vq[e];
      r[1] = vpos[e];
      for (int j = 0; j < kVirtInts; ++j) r[2 + j] = v[e * kVirtInts + j];
    }
    d_asm_r8.upload(r8, st);
    plan_.asm_r8 = d_asm_r8.p;
  }
  if (!band_ent_asm_.empty()) {   // the band chains' records of their original blocks (band_chain.inc), virtual source
    std::vector<int4> ev(band_ent_h_);
    const long long bb = (long long)bs_ * bs_;
    bool fits = true;
    for (size_t i = 0; i < band_ent_asm_.size(); ++i) {
      const int e = band_ent_asm_[i];
      const int* r = &v[(size_t)e * kVirtInts];   // count, three partial slots, first list index
      for (int k = 1; k <= 3; ++k) fits = fits && (long long)r[k] * bb < 0xfffff000LL;
      fits = fits && (long long)vq[e] * bb < 0x7ffff000LL;
      ev[2 * i] = make_int4(vq[e] >= 0 ? (int)(vq[e] * bb) : -1, (int)(unsigned int)(r[1] * bb), (int)(unsigned int)(r[2] * bb), (int)(unsigned int)(r[3] * bb));
      const int4 h1 = band_ent_h_[2 * i + 1];
      ev[2 * i + 1] = make_int4((h1.x & 0xff) | (vpos[e] < 0 ? 2 : 0) | (r[0] << 8), r[4], h1.z, h1.w);
    }
    if (fits) {
      d_band_entv.upload(ev, st);
      plan_.band_entv = d_band_entv.p;
    } else {
      plan_.band_entv = nullptr;   // (offsets beyond 32 bits: the general kernel takes these chains)
    }
  }
  d_asm_vq.upload(vq, st);
  d_asm_vpos.upload(vpos, st);
  d_asm_v.upload(v, st);
  plan_.asm_vq = d_asm_vq.p;
  plan_.asm_vpos = d_asm_vpos.p;
  plan_.asm_v = d_asm_v.p;
  plan_.vslots = vb.d_part_slot;
  plan_.vbase = vb.base;
  plan_.vparts = vb.parts;
  plan_.vlam = vb.lam;
  plan_.vsplit = vb.split ? 1 : 0;
}

bool SparseCholesky::big_forward_carried(const LevelLaunch& LL) const {
  // (the conditions of the pivot-block kernel on the matrix cores, of the fused panel kernel and of single-front tasks with
  // at most 64 pivot columns)
  return opt.fuse_big_forward && opt.mfma_diag && opt.fuse_panel && opt.big_front_passes && LL.big_ok && LL.glb_count > 0 &&
         LL.glb_max_m >= opt.big_front_min_dim && LL.bt_count <= merge_tiles_of(LL) && LL.sw_count > 0;
}

void SparseCholesky::launch_factor(const LevelLaunch& LL, const double* dA, bool fwd, hipStream_t st, bool dep, int parts) {
#ifdef G2OHIP_CHOL_STAMPS
  if (d_dbg.p) {
    plan_.dbg = d_dbg.p + 64 * (dbg_launch_++ % 64);
  }
#endif
  CholPlanDev fplan = plan_;
  fplan.slots = d_fslots.p;
  const BigLaunch big{LL.big_ok && opt.big_front_passes && LL.glb_max_m >= opt.big_front_min_dim, d_big_tiles.p, LL.ba_begin, LL.ba_count, LL.tr_begin, LL.tr_count, &LL.be_pass,
                      LL.fz_begin, LL.fz_count, LL.hoisted && opt.hoist_big_assembly != 0, fwd && big_forward_carried(LL),
                      (opt.merge_diag_panel && !dep_off_) ? d_sw_flag.p : (int*)nullptr, d_scratch_ld.p, opt.fuse_panel != 0, opt.mfma_diag != 0, merge_tiles_of(LL), LL.gather && opt.big_gather != 0,
                      LL.eg_begin, LL.eg_count, LL.eg_write, LL.eg_maxc, LL.la_begin, LL.la_count, LL.tr_all};
  const bool virt = dA == nullptr;   // assemble from the virtual source (set_virtual_blocks)
  if (virt && !has_virtual_blocks()) throw StateFailure("SparseCholesky::factor: no matrix and no virtual source");
#define G2OHIP_FACTOR_LEVEL(BS_, V_)                                                                                          \
  launch_factor_level<BS_, V_>(fplan, d_level_fronts.p, d_scratch_off.p, d_scratch.p, dA, LL.lds_begin, LL.lds_count, LL.lds_max_m, \
                               LL.glb_begin, LL.glb_count, LL.lds_idx_ints, LL.glb_idx_ints, LL.sm_count, LL.sm_max_m,            \
                               LL.sm_idx_ints, LL.lds_vec_m, fwd ? d_xp.p : (const double*)nullptr, fwd ? d_y.p : (double*)nullptr,   \
                               dep ? 1 : 0, d_big_tiles.p + LL.bt_begin, LL.bt_count, big, opt.wide_front_doubles, LL.wv, LL.wv_pn,   \
                               LL.wv_idx_ints, st, parts)
  switch (bs_) {
    case 3:
      if (virt) G2OHIP_FACTOR_LEVEL(3, true); else G2OHIP_FACTOR_LEVEL(3, false);
      break;
    case 6:
      if (virt) G2OHIP_FACTOR_LEVEL(6, true); else G2OHIP_FACTOR_LEVEL(6, false);
      break;
    case 7:
      if (virt) G2OHIP_FACTOR_LEVEL(7, true); else G2OHIP_FACTOR_LEVEL(7, false);
      break;
    default:
      throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
  }
#undef G2OHIP_FACTOR_LEVEL
}

bool SparseCholesky::band_usable(const FactorGroup& G, const double* dA) const {
  // (a virtual source whose offsets do not fit the 32-bit records leaves the chains to the general kernel)
  return G.band_count > 0 && opt.band_kernel && bs_ == 6 && (dA != nullptr || plan_.band_entv != nullptr);
}

void SparseCholesky::launch_band(const FactorGroup& G, const double* dA, bool fused, hipStream_t st, bool dep) {
  const bool virt = dA == nullptr;
  if (virt && !has_virtual_blocks()) throw StateFailure("SparseCholesky::factor: no matrix and no virtual source");
  const BandPlanArgs B{plan_.band_rec, plan_.band_tab, virt ? plan_.band_entv : plan_.band_ent, 0, G.band_tab_cap,
                       getenv("G2OHIP_BAND_ABL") ? atoi(getenv("G2OHIP_BAND_ABL")) : 0};
  const size_t sh = (size_t)(5 * 256) * sizeof(double) + (size_t)(4 * kBandListCap) * sizeof(int4) + (size_t)(64 + 96 + G.band_tab_cap + 4) * sizeof(int);
  const double* bp = fused ? d_xp.p : (const double*)nullptr;
  double* yo = fused ? d_y.p : (double*)nullptr;
  const WvPlan wp = wv_plan(plan_);
  const int dep_i = dep ? 1 : 0;
  if (band_hook) band_hook(0);
  if (virt) hipLaunchKernelGGL((band_wave_kernel<6, true>), dim3(G.band_count), dim3(64), sh, st, wp, B, G.band_rec0, dA, bp, yo, dep_i);
  else hipLaunchKernelGGL((band_wave_kernel<6, false>), dim3(G.band_count), dim3(64), sh, st, wp, B, G.band_rec0, dA, bp, yo, dep_i);
  G2OHIP_LAUNCH_CHECK("band_wave_kernel");
  if (band_hook) band_hook(1);
}

bool SparseCholesky::has_band_chains(int phase) const {
  for (const FactorGroup& G : groups_[phase])
    if (G.band_count > 0 && opt.band_kernel && bs_ == 6) return true;
  return false;
}

// One-time per process: dynamic-LDS limits of the factor kernels (hipFuncSetAttribute also makes the runtime load this file's code
// object).  Called at the end of analyze() so that the first factorisation of a process does not pay for it (4 ms of the first
// solve of a Levenberg-Marquardt run were these calls), and again from factor_phase as a no-op.
void SparseCholesky::prepare_kernels() {
  static bool attr_done = false;
  if (!attr_done) {
    // allow > 64 KiB dynamic LDS for the LDS-resident front kernels
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true, kFactorThreads, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true, kFactorThreads, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true, kFactorThreads, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_panel_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<3, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<6, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<6, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<7, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<7, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<3, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<6, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<6, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<7, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)big_level_kernel<7, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
}

void SparseCholesky::factor_phase(const double* dA, int phase, hipStream_t st, bool fwd, int parts) {
  if (!analyzed_) throw StateFailure("SparseCholesky::factor before analyze");
  prepare_kernels();
  if (phase == 0 && (parts & 1)) {
    if (!skip_status_clear) G2OHIP_HIP_CHECK(hipMemsetAsync(d_status.p, 0, sizeof(int), st));
    skip_status_clear = false;
  }
#ifdef G2OHIP_CHOL_STAMPS
  if (phase == 0) {
    if (!d_dbg.p) d_dbg.alloc(64 * 64 + 6 * (size_t)n_slots_ + 6);
    G2OHIP_HIP_CHECK(hipMemsetAsync(d_dbg.p, 0, 64 * 64 * sizeof(long long), st));
    plan_.tl = d_dbg.p + 64 * 64;
    dbg_launch_ = 0;
  }
#endif
  if ((parts & 2) && opt.merge_diag_panel && !dep_off_ && d_sw_flag.p && ha_count_[phase] > 0) d_sw_flag.zero(st);   // per-front flags of big_level_kernel (phases with scratch-slab levels only)
  if ((parts & 2) && opt.hoist_big_assembly && (hz_count_[phase] > 0 || ha_count_[phase] > 0)) {
    const bool virt = dA == nullptr;
    if (virt && !has_virtual_blocks()) throw StateFailure("SparseCholesky::factor: no matrix and no virtual source");
    if (hz_count_[phase] > 0)
      hipLaunchKernelGGL(big_fill_kernel, dim3(hz_count_[phase]), dim3(256), 0, st, d_big_tiles.p + hz_begin_[phase], d_scratch.p, d_scratch_off.p, d_scratch_ld.p);
    G2OHIP_LAUNCH_CHECK("big_fill_kernel");
    if (ha_count_[phase] > 0) {
      const int4* ch = d_big_tiles.p + ha_begin_[phase];
      CholPlanDev fplan = plan_;
      fplan.slots = d_fslots.p;   // (as launch_factor)
#define G2OHIP_HOIST_ASM(BS_)                                                                                                    \
  if (virt) hipLaunchKernelGGL((big_assemble_kernel<BS_, true>), dim3(ha_count_[phase]), dim3(256), 0, st, fplan, ch, dA, d_scratch.p, d_scratch_off.p, d_scratch_ld.p); \
  else hipLaunchKernelGGL((big_assemble_kernel<BS_, false>), dim3(ha_count_[phase]), dim3(256), 0, st, fplan, ch, dA, d_scratch.p, d_scratch_off.p, d_scratch_ld.p)
      switch (bs_) {
        case 3: G2OHIP_HOIST_ASM(3); break;
        case 6: G2OHIP_HOIST_ASM(6); break;
        case 7: G2OHIP_HOIST_ASM(7); break;
        default: throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
      }
#undef G2OHIP_HOIST_ASM
      G2OHIP_LAUNCH_CHECK("big_assemble_kernel");
    }
  }
  bool fwd_pending = false;   // a forward step of large fronts is in flight on side_[1] (event ev_[3])
  bool side_dirty = false, side_used = false, side_unsure = false;   // side_[0]: work the main stream has not waited for (ev_[1]) / used in this call / plan flags void
  for (const FactorGroup& G : groups_[phase]) {
    if (G.dep && dep_off_) {   // (groups only hold levels the fused kernel carries completely)
      if (side_dirty) {
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[1], 0));
        side_dirty = false;
      }
      for (int l = G.first_level; l <= G.last_level; ++l) {
        LevelLaunch one = launches_[phase][l];
        if (l == G.first_level && band_usable(G, dA)) {   // (the same kernels as the grouped launch: results stay bit-identical)
          if (parts & 1) launch_band(G, dA, fwd, st, false);
          one.lds_begin += G.band_count;
          one.lds_count -= G.band_count;
          if (one.lds_count <= 0) continue;
        }
        if (parts & 2) launch_factor(one, dA, fwd, st, false);
      }
      continue;
    }
    const LevelLaunch& LL = G.LL;
    const bool fused = fwd && LL.fuse_fwd;
    if (!(parts & 2) && !band_usable(G, dA)) continue;   // (a band-only call: nothing else of this group)
    const bool big_passes = LL.big_ok && opt.big_front_passes && LL.glb_max_m >= opt.big_front_min_dim;
    if (opt.overlap_level_halves && st != nullptr && big_passes && LL.glb_count > 0 && LL.lds_count > 0 && !G.dep && (fused || !fwd)) {
      // (levels with large fronts only stay on one stream: a forward step moved to a side stream was measured to cost more
      // in cross-stream dependencies than the 14 us it hides)
      // The LDS fronts and the scratch-slab fronts of a level do not depend on each other: the one launch of the former
      // runs on a side stream next to the chain of whole-GPU passes of the latter; the forward step of the large fronts
      // runs on a second side stream next to the NEXT level's passes (only that level's LDS launch and forward step
      // read its update vectors).  Inside a stream capture the side streams become parallel branches of the graph.
      if (!side_[0]) {
        for (int i = 0; i < 2; ++i) G2OHIP_HIP_CHECK(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming));
      }
      // (lazy: the streams wait for each other only where the tree asks for it -- LevelLaunch::fork / join; the first use of the side
      // stream in a call always forks: inside a stream capture that is what makes it part of the graph)
      const bool lazy = opt.lazy_level_joins && LL.split_ok && !side_unsure;
      if (side_dirty && (!lazy || LL.join)) {
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[1], 0));
        side_dirty = false;
      }
      if (!lazy || LL.fork || !side_used) {
        G2OHIP_HIP_CHECK(hipEventRecord(ev_[0], st));
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(side_[0], ev_[0], 0));
        side_unsure = false;
      }
      if (fwd_pending) G2OHIP_HIP_CHECK(hipStreamWaitEvent(side_[0], ev_[3], 0));
      launch_factor(LL, dA, fused, side_[0], false, 1);
      G2OHIP_HIP_CHECK(hipEventRecord(ev_[1], side_[0]));
      side_dirty = side_used = true;
      launch_factor(LL, dA, fused, st, false, 2);
      if (fwd && !big_forward_carried(LL)) {
        G2OHIP_HIP_CHECK(hipEventRecord(ev_[2], st));
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(side_[1], ev_[2], 0));
        launch_solve(LL, true, side_[1], /*glb_only=*/true);
        G2OHIP_HIP_CHECK(hipEventRecord(ev_[3], side_[1]));
        fwd_pending = true;
      }
      if (!lazy) {
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[1], 0));
        side_dirty = false;
      }
      continue;
    }
    if (side_dirty && (!opt.lazy_level_joins || LL.join || LL.split_ok || side_unsure)) {   // (a front of this level has a child on the side stream)
      G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[1], 0));
      side_dirty = false;
    }
    if (LL.split_ok) side_unsure = true;   // (a level the plan put on the side stream ran here: its fork / join flags no longer describe the streams)
    if (opt.group_forward_side && st != nullptr && big_passes && (LL.grouped || LL.group_in) && LL.lds_count == 0 && LL.glb_count > 0 && !G.dep && fwd &&
        !big_forward_carried(LL)) {
      // Panels of a grouped in-place chain: the forward step of a panel (20 us of a 65 us level) needs the panel's solved rows and the
      // previous panel's update vector, nothing of the next panel's factorisation -- it runs on the side stream next to it (inside a
      // stream capture: a parallel branch of the graph); the first level that is not such a panel waits for it (fwd_pending below).
      if (!side_[0]) {
        for (int i = 0; i < 2; ++i) G2OHIP_HIP_CHECK(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming));
      }
      launch_factor(LL, dA, false, st, false, 2);
      G2OHIP_HIP_CHECK(hipEventRecord(ev_[2], st));
      G2OHIP_HIP_CHECK(hipStreamWaitEvent(side_[1], ev_[2], 0));
      launch_solve(LL, true, side_[1], /*glb_only=*/true);
      G2OHIP_HIP_CHECK(hipEventRecord(ev_[3], side_[1]));
      fwd_pending = true;
      continue;
    }
    if (fwd_pending) {   // (a level that is not split reads the update vectors on the main stream)
      G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[3], 0));
      fwd_pending = false;
    }
    if (band_usable(G, dA)) {
      // the band chains of level 0 first, in a launch of their own (they bump their parents' counters like the others)
      if (parts & 1) launch_band(G, dA, fused, st, G.dep);
      if (!(parts & 2)) continue;
      LevelLaunch rest = LL;
      rest.lds_begin += G.band_count;
      rest.lds_count -= G.band_count;
      if (rest.lds_count > 0) launch_factor(rest, dA, fused, st, G.dep);
    } else {
      launch_factor(LL, dA, fused, st, G.dep);
    }
    // what the factor kernel did not carry (fronts too large for LDS, launches outside the fused kernel's
    // limits) gets its forward step inside the same level
    const bool carried = fwd && big_forward_carried(LL);   // (the scratch-slab fronts' forward step went with their factorisation)
    if (fwd && !fused) launch_solve(LL, true, st, false, false, /*skip_glb=*/carried);
    else if (fwd && LL.glb_count > 0 && !carried) launch_solve(LL, true, st, /*glb_only=*/true);
  }
  if (fwd_pending) G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[3], 0));
  if (side_dirty) G2OHIP_HIP_CHECK(hipStreamWaitEvent(st, ev_[1], 0));
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// Sparse inverse (entries of A^-1 on the pattern of L): Z = A^-1 satisfies, for a frontal matrix with pivot columns J,
// boundary rows B, panel [L11; L21] and Y = L21 L11^-1:
//     Z_BB  from the parent's inverse front (the boundary rows of a front are rows of its parent),
//     Z_JB = -Y' Z_BB,        Z_JJ = (L11 L11')^-1 - Z_JB Y.
// (Takahashi, Fagan & Chin 1973; Erisman & Tinney 1975 -- what CHOLMOD's / CSparse's users get from "sparseinv"; the
// reference's computeMarginals solves for columns instead, linear_solver.h:62-66 -> MarginalCovarianceCholesky.)  Top
// down over the front levels, three kernels per level; every front keeps a dense m x m inverse front in d_Z.
__global__ void __launch_bounds__(256) spinv_gather_y_kernel(CholPlanDev P, int bs, const long long* __restrict__ zoff,
                                                           const int* __restrict__ fparent, double* __restrict__ Z,
                                                           const int4* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int4 wk = work[blockIdx.x];
  const int f = wk.x, a0 = wk.y;
  const int ns = P.f_ns[f], nb = P.f_nb[f];
  const int npiv = ns * bs, nbr = nb * bs, m = npiv + nbr;
  const int p = fparent[f];
  const int mp = (P.f_ns[p] + P.f_nb[p]) * bs;
  double* Zf = Z + zoff[f];
  const double* Zp = Z + zoff[p];
  const int* rel = P.rel + P.rel_off[f];
  const double* Lg = P.L + P.L_off[f];
  double* L11 = sm;                 // npiv x npiv (column-major, ld npiv)
  double* ys = sm + npiv * npiv;    // 64 rows x (npiv + 1)
  const int tid = threadIdx.x;
  for (int i = tid; i < npiv * npiv; i += 256) L11[i] = Lg[(i % npiv) + (size_t)m * (i / npiv)];
  __syncthreads();
  const int a = a0 + (tid & 63), part = tid >> 6;
  if (a < nbr) {
    const int pa = rel[a / bs] * bs + a % bs;
    for (int b = part; b < nbr; b += 4) {
      const int pb = rel[b / bs] * bs + b % bs;
      Zf[(size_t)(npiv + a) + (size_t)m * (npiv + b)] = Zp[(size_t)pa + (size_t)mp * pb];
    }
    if (part == 0) {   // y L11 = l (row a of L21), from the last column to the first
      double* y = ys + (tid & 63) * (npiv + 1);
      for (int k = npiv - 1; k >= 0; --k) {
        double v = Lg[(size_t)(npiv + a) + (size_t)m * k];
        for (int j = k + 1; j < npiv; ++j) v -= y[j] * L11[j + npiv * k];
        y[k] = v * Lg[(size_t)m * npiv + k];
      }
      for (int k = 0; k < npiv; ++k) Zf[(size_t)(npiv + a) + (size_t)m * k] = y[k];
    }
  }
}

__global__ void __launch_bounds__(256) spinv_zjb_kernel(CholPlanDev P, int bs, const long long* __restrict__ zoff, double* __restrict__ Z,
                                                      const int4* __restrict__ work) {
  const int4 wk = work[blockIdx.x];
  const int f = wk.x, b0 = wk.y;
  const int ns = P.f_ns[f], nb = P.f_nb[f];
  const int npiv = ns * bs, nbr = nb * bs, m = npiv + nbr;
  double* Zf = Z + zoff[f];
  const int b = b0 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  if (b >= nbr) return;
  // Z_JB(k, b) = -sum_a Y(a, k) Z_BB(a, b), this thread's k = part, part + 4, ... (Z_BB symmetric: read along rows)
  for (int k0 = part; k0 < npiv; k0 += 32) {
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    for (int a = 0; a < nbr; ++a) {
      const double z = Zf[(size_t)(npiv + b) + (size_t)m * (npiv + a)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 4 * u;
        if (k < npiv) acc[u] += Zf[(size_t)(npiv + a) + (size_t)m * k] * z;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 4 * u;
      if (k < npiv) Zf[(size_t)k + (size_t)m * (npiv + b)] = -acc[u];
    }
  }
}

__global__ void __launch_bounds__(256) spinv_zjj_kernel(CholPlanDev P, int bs, const long long* __restrict__ zoff, double* __restrict__ Z,
                                                      const int4* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int f = work[blockIdx.x].x;
  const int ns = P.f_ns[f], nb = P.f_nb[f];
  const int npiv = ns * bs, nbr = nb * bs, m = npiv + nbr;
  double* Zf = Z + zoff[f];
  const double* Lg = P.L + P.L_off[f];
  double* L11 = sm;                  // npiv x npiv
  double* W = sm + npiv * npiv;      // L11^-1, npiv x npiv (lower triangular)
  const int tid = threadIdx.x;
  for (int i = tid; i < npiv * npiv; i += 256) L11[i] = Lg[(i % npiv) + (size_t)m * (i / npiv)];
  __syncthreads();
  if (tid < npiv) {   // column tid of W: L11 w = e
    const int j = tid;
    for (int i = 0; i < npiv; ++i) {
      double v = (i == j) ? 1.0 : 0.0;
      if (i < j) {
        W[i + npiv * j] = 0.0;
        continue;
      }
      for (int q = j; q < i; ++q) v -= L11[i + npiv * q] * W[q + npiv * j];
      W[i + npiv * j] = v * Lg[(size_t)m * npiv + i];
    }
  }
  __syncthreads();
  for (int e = tid; e < npiv * npiv; e += 256) {
    const int k = e % npiv, l = e / npiv;
    double s_ = 0.0;
    for (int i = (k > l ? k : l); i < npiv; ++i) s_ += W[i + npiv * k] * W[i + npiv * l];
    for (int a = 0; a < nbr; ++a) s_ -= Zf[(size_t)k + (size_t)m * (npiv + a)] * Zf[(size_t)(npiv + a) + (size_t)m * l];
    L11[e] = s_;   // (L11 is no longer needed: Z_JJ is collected here, the Y block it reads must stay until all are done)
  }
  __syncthreads();
  for (int e = tid; e < npiv * npiv; e += 256) Zf[(size_t)(e % npiv) + (size_t)m * (e / npiv)] = L11[e];
  for (int e = tid; e < npiv * nbr; e += 256) {   // Z_BJ = Z_JB' in place of Y
    const int k = e % npiv, a = e / npiv;
    Zf[(size_t)(npiv + a) + (size_t)m * k] = Zf[(size_t)k + (size_t)m * (npiv + a)];
  }
}

void SparseCholesky::sparse_inverse(hipStream_t st) {
  if (!analyzed_) throw StateFailure("SparseCholesky::sparse_inverse before analyze");
  if (opt.world > 1) throw StateFailure("SparseCholesky::sparse_inverse: one GPU only");
  const CholSymbolic& S = sym_;
  const int nf = (int)S.f_ns.size();
  if (!spinv_planned_) {
    zoff_h_.assign(nf + 1, 0);
    int max_level = 0, max_npiv = 0;
    for (int f = 0; f < nf; ++f) {
      const long long m = (long long)(S.f_ns[f] + S.f_nb[f]) * bs_;
      zoff_h_[f + 1] = zoff_h_[f] + m * m;
      max_level = std::max(max_level, S.f_level[f]);
      max_npiv = std::max(max_npiv, S.f_ns[f] * bs_);
    }
    if ((size_t)(2 * max_npiv * max_npiv + 64 * (max_npiv + 1)) * sizeof(double) > 150 * 1024)
      throw StateFailure("SparseCholesky::sparse_inverse: pivot panel too wide for the LDS stage");
    std::vector<int4> work;
    spinv_levels_.clear();
    for (int lev = max_level; lev >= 0; --lev) {   // parents first
      SpInvLevel L;
      L.f_begin = (int)work.size();
      for (int f = 0; f < nf; ++f)
        if (S.f_level[f] == lev) work.push_back(make_int4(f, 0, 0, 0));
      L.f_count = (int)work.size() - L.f_begin;
      L.rc_begin = (int)work.size();
      for (int i = 0; i < L.f_count; ++i) {
        const int f = work[L.f_begin + i].x;
        for (int a0 = 0; a0 < S.f_nb[f] * bs_; a0 += 64) work.push_back(make_int4(f, a0, 0, 0));
      }
      L.rc_count = (int)work.size() - L.rc_begin;
      spinv_levels_.push_back(L);
    }
    d_spinv_work.upload(work, st);
    d_zoff.upload(zoff_h_, st);
    std::vector<int> par(S.f_parent.begin(), S.f_parent.end());
    for (int& v : par)
      if (v < 0) v = 0;   // (roots have no boundary rows: never read)
    d_fparent.upload(par, st);
    d_Z.alloc((size_t)std::max<long long>(zoff_h_[nf], 1));
    (void)hipFuncSetAttribute((const void*)spinv_gather_y_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)spinv_zjj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    spinv_npiv_max_ = max_npiv;
    spinv_planned_ = true;
  }
  const size_t sh1 = (size_t)(spinv_npiv_max_ * spinv_npiv_max_ + 64 * (spinv_npiv_max_ + 1)) * sizeof(double);
  const size_t sh3 = (size_t)(2 * spinv_npiv_max_ * spinv_npiv_max_) * sizeof(double);
  for (const SpInvLevel& L : spinv_levels_) {
    if (L.rc_count > 0) {
      hipLaunchKernelGGL(spinv_gather_y_kernel, dim3(L.rc_count), dim3(256), sh1, st, plan_, bs_, d_zoff.p, d_fparent.p, d_Z.p,
                         d_spinv_work.p + L.rc_begin);
      hipLaunchKernelGGL(spinv_zjb_kernel, dim3(L.rc_count), dim3(256), 0, st, plan_, bs_, d_zoff.p, d_Z.p, d_spinv_work.p + L.rc_begin);
    }
    if (L.f_count > 0)
      hipLaunchKernelGGL(spinv_zjj_kernel, dim3(L.f_count), dim3(256), sh3, st, plan_, bs_, d_zoff.p, d_Z.p, d_spinv_work.p + L.f_begin);
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

bool SparseCholesky::inverse_block(int r, int c, long long* offset, int* ld, bool* transposed) const {
  const CholSymbolic& S = sym_;
  if (!spinv_planned_ || r < 0 || c < 0 || r >= S.nb || c >= S.nb) return false;
  int pr = S.iperm[r], pc = S.iperm[c];
  const bool tr = pr < pc;   // stored: row >= column (in elimination order)
  if (tr) std::swap(pr, pc);
  const int f = (int)(std::upper_bound(S.sn_start.begin(), S.sn_start.end(), pc) - S.sn_start.begin()) - 1;
  const int c0 = S.sn_start[f], ns = S.f_ns[f];
  int rowpos;
  if (pr < c0 + ns) {
    rowpos = pr - c0;
  } else {
    const int* b = S.rows.data() + S.rows_off[f];
    const int* e = b + S.f_nb[f];
    const int* it = std::lower_bound(b, e, pr);
    if (it == e || *it != pr) return false;
    rowpos = ns + (int)(it - b);
  }
  const long long m = (long long)(ns + S.f_nb[f]) * bs_;
  *offset = zoff_h_[f] + (long long)rowpos * bs_ + m * ((long long)(pc - c0) * bs_);
  *ld = (int)m;
  *transposed = tr;
  return true;
}

void SparseCholesky::factor(const double* dA, hipStream_t st) {
  factor_phase(dA, 0, st);
  factor_phase(dA, 1, st);
}

// Factorisation with the forward sweep fused in (what every reference solve() call amounts to: it refactorises
// each time, linear_solver_csparse.h:106-142), then the backward sweep.
void SparseCholesky::factor_solve(const double* dA, const double* d_b, double* d_x, hipStream_t st) {
  solve_begin(d_b, st);
  factor_phase(dA, 0, st, true);
  factor_phase(dA, 1, st, true);
  solve_backward_phase(1, st);
  solve_backward_phase(0, st);
  solve_end(d_x, st);
}

void SparseCholesky::launch_solve(const LevelLaunch& LL, bool fwd, hipStream_t st, bool glb_only, bool dep, bool skip_glb, int dep_tail) {
  const size_t panel_limit = 48 * 1024;  // bytes of L panel staged in LDS
  int count = glb_only ? LL.glb_count : LL.lds_count + (skip_glb ? 0 : LL.glb_count);
  // dep: backward sweep of a dependency-driven group (no scratch-slab tasks) over the reversed slot list; dep_tail > 0: only its
  // last dep_tail slots (the lowest levels: the ones above them were swept by tree_backward_kernel)
  int slot0 = dep ? n_slots_ - (LL.lds_begin + LL.lds_count) : (glb_only ? LL.glb_begin : LL.lds_begin);
  if (dep && dep_tail > 0) {
    slot0 += count - dep_tail;
    count = dep_tail;
  }
  if (count == 0) return;
  CholPlanDev bplan = plan_;
  if (dep) bplan.slots = d_bslots.p;
  const int depi = dep ? 1 : 0;
  int max_m = LL.max_m, max_panel = LL.max_panel;
  if (skip_glb) {
    max_m = LL.lds_vec_m;
    max_panel = LL.lds_max_panel;
  }
  if (!dep && !skip_glb && opt.split_sweeps && LL.sw_count > 0 && LL.glb_max_m >= opt.split_sweeps_min_dim) {
    // the scratch-slab fronts of the level: several workgroups per front; the LDS / register fronts (independent of
    // them: same level) follow in their own launch, sized for themselves
    const int4* ch = d_big_tiles.p + LL.sw_begin;
#define G2OHIP_SPLIT_SWEEP(BS_)                                                                                                 \
  if (fwd) hipLaunchKernelGGL((big_forward_kernel<BS_>), dim3(LL.sw_count), dim3(256), 0, st, plan_, ch, d_xp.p, d_y.p);          \
  else hipLaunchKernelGGL((big_backward_kernel<BS_, false>), dim3(LL.sw_count), dim3(256), 0, st, plan_, ch, d_y.p, d_xp.p, d_sw_part.p, d_sw_cnt.p, (int*)nullptr)
    switch (bs_) {
      case 3: G2OHIP_SPLIT_SWEEP(3); break;
      case 6: G2OHIP_SPLIT_SWEEP(6); break;
      case 7: G2OHIP_SPLIT_SWEEP(7); break;
      default: throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
    }
#undef G2OHIP_SPLIT_SWEEP
    G2OHIP_LAUNCH_CHECK("big_forward_kernel / big_backward_kernel");
    if (glb_only || LL.lds_count == 0) return;
    count = LL.lds_count;
    max_m = LL.lds_vec_m;
    max_panel = LL.lds_max_panel;
  }
  bool panel = (size_t)max_panel * 8 <= panel_limit;
  int cap = panel ? max_panel : 0;
  int nthreads = max_m <= 64 ? 64 : (max_m <= 128 ? 128 : 256);
  size_t sh = ((size_t)cap + 3 * (size_t)max_m + nthreads + 8) * sizeof(double);
  double* gvec = nullptr;
  if (sh > 160 * 1024) {   // vectors of the fronts in HBM (dense reduced systems: fronts beyond ~6 500 rows)
    const size_t need = (size_t)count * 3 * (size_t)max_m;
    if (d_sweep_vec.n < need) d_sweep_vec.alloc(need);
    gvec = d_sweep_vec.p;
    cap = 0;
    sh = ((size_t)nthreads + 8) * sizeof(double);
  }
  if (sh > 64 * 1024) {   // (fronts of several thousand rows: dense couplings, e.g. a landmark seen by hundreds of poses)
    static bool attr_done = false;
    if (!attr_done) {
#define G2OHIP_SWEEP_ATTR(BS_)                                                                                                 \
  (void)hipFuncSetAttribute((const void*)front_forward_kernel<BS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
  (void)hipFuncSetAttribute((const void*)front_forward_kernel<BS_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void)hipFuncSetAttribute((const void*)front_backward_kernel<BS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void)hipFuncSetAttribute((const void*)front_backward_kernel<BS_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      G2OHIP_SWEEP_ATTR(3);
      G2OHIP_SWEEP_ATTR(6);
      G2OHIP_SWEEP_ATTR(7);
#undef G2OHIP_SWEEP_ATTR
      (void)hipGetLastError();
      attr_done = true;
    }
  }
#define G2OHIP_SOLVE_LAUNCH(BS_)                                                                                              \
  if (gvec) {   /* (fronts beyond the LDS limit: never with a staged panel) */                                                \
    if (fwd)                                                                                                                  \
      hipLaunchKernelGGL((front_forward_kernel<BS_, false, true>), dim3(count), dim3(nthreads), sh, st, plan_, slot0, d_xp.p, d_y.p, 0, max_m, gvec); \
    else                                                                                                                      \
      hipLaunchKernelGGL((front_backward_kernel<BS_, false, true>), dim3(count), dim3(nthreads), sh, st, bplan, slot0, d_y.p, d_xp.p, 0, max_m, depi, gvec); \
  } else if (fwd) {                                                                                                           \
    if (panel)                                                                                                                \
      hipLaunchKernelGGL((front_forward_kernel<BS_, true>), dim3(count), dim3(nthreads), sh, st, plan_, slot0, d_xp.p, d_y.p, cap, max_m, gvec);  \
    else                                                                                                                      \
      hipLaunchKernelGGL((front_forward_kernel<BS_, false>), dim3(count), dim3(nthreads), sh, st, plan_, slot0, d_xp.p, d_y.p, cap, max_m, gvec); \
  } else {                                                                                                                    \
    if (panel)                                                                                                                \
      hipLaunchKernelGGL((front_backward_kernel<BS_, true>), dim3(count), dim3(nthreads), sh, st, bplan, slot0, d_y.p, d_xp.p, cap, max_m, depi, gvec); \
    else                                                                                                                      \
      hipLaunchKernelGGL((front_backward_kernel<BS_, false>), dim3(count), dim3(nthreads), sh, st, bplan, slot0, d_y.p, d_xp.p, cap, max_m, depi, gvec); \
  }
  switch (bs_) {
    case 3: G2OHIP_SOLVE_LAUNCH(3) break;
    case 6: G2OHIP_SOLVE_LAUNCH(6) break;
    case 7: G2OHIP_SOLVE_LAUNCH(7) break;
    default: throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
  }
#undef G2OHIP_SOLVE_LAUNCH
}

void SparseCholesky::solve_begin(const double* d_b, hipStream_t st) {
  if (!analyzed_) throw StateFailure("SparseCholesky::solve before analyze");
  const int n = sym_.nb * bs_;
  if (n == 0) return;
  hipLaunchKernelGGL(permute_in_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sym_.nb, bs_, d_perm.p, d_b, d_xp.p);
}
void SparseCholesky::solve_forward_phase(int phase, hipStream_t st) {
  // d_xp holds the permuted rhs, y receives the pivot solutions
  for (size_t l = 0; l < launches_[phase].size(); ++l) launch_solve(launches_[phase][l], true, st);
}
void SparseCholesky::solve_backward_phase(int phase, hipStream_t st) {
  // overwrite d_xp with the solution, highest level first
  const bool merge = opt.merge_backward_levels && opt.split_sweeps && !dep_off_ && !bw_groups_[phase].empty();
  bool flags_zeroed = false;
  for (size_t g = groups_[phase].size(); g-- > 0;) {
    const FactorGroup& G = groups_[phase][g];
    if (G.dep && opt.dep_backward && !dep_off_) {
      if (G.tb_ngrp > 0) {   // the tree levels by groups of fronts, then the levels below them task by task
        const int nt_ref = G.LL.max_m <= 64 ? 64 : (G.LL.max_m <= 128 ? 128 : 256);   // (launch_solve's workgroup size: bw_parts)
        const TreeGroupRec* gr = reinterpret_cast<const TreeGroupRec*>(d_tb_grec.p) + G.tb_grp0;
#define G2OHIP_TREE_BACKWARD(BS_) \
  hipLaunchKernelGGL((tree_backward_kernel<BS_>), dim3(G.tb_ngrp), dim3(kTreeWaves * 64), 0, st, plan_, gr, d_tb_front.p, d_tb_rows.p, d_y.p, d_xp.p, nt_ref)
        switch (bs_) {
          case 3: G2OHIP_TREE_BACKWARD(3); break;
          case 6: G2OHIP_TREE_BACKWARD(6); break;
          case 7: G2OHIP_TREE_BACKWARD(7); break;
          default: throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
        }
#undef G2OHIP_TREE_BACKWARD
        G2OHIP_LAUNCH_CHECK("tree_backward_kernel");
        if (G.tb_low > 0) {
          launch_solve(G.LL, false, st, false, true, false, G.tb_low);
        }
      } else {
        launch_solve(G.LL, false, st, false, true);
      }
    } else {
      for (int l = G.last_level; l >= G.first_level; --l) {
        const int bg = merge ? bw_of_level_[phase][l] : -1;
        if (bg >= 0) {   // a run of levels of scratch-slab fronts: one launch at its top level, nothing at the others
          const BwGroup& B = bw_groups_[phase][bg];
          if (l != B.top_level) continue;
          if (!flags_zeroed) {
            d_sw_flag.zero(st);
            flags_zeroed = true;
          }
          const int4* ch = d_big_tiles.p + B.begin;
#define G2OHIP_MERGED_BACKWARD(BS_) \
  hipLaunchKernelGGL((big_backward_kernel<BS_, true>), dim3(B.count), dim3(256), 0, st, plan_, ch, d_y.p, d_xp.p, d_sw_part.p, d_sw_cnt.p, d_sw_flag.p)
          switch (bs_) {
            case 3: G2OHIP_MERGED_BACKWARD(3); break;
            case 6: G2OHIP_MERGED_BACKWARD(6); break;
            case 7: G2OHIP_MERGED_BACKWARD(7); break;
            default: throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
          }
#undef G2OHIP_MERGED_BACKWARD
          G2OHIP_LAUNCH_CHECK("big_backward_kernel (merged levels)");
          continue;
        }
        launch_solve(launches_[phase][l], false, st);
      }
    }
  }
}
void SparseCholesky::solve_end(double* d_x, hipStream_t st) {
  const int n = sym_.nb * bs_;
  if (n == 0) return;
  hipLaunchKernelGGL(permute_out_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sym_.nb, bs_, d_perm.p, d_xp.p, d_x);
  G2OHIP_HIP_CHECK(hipGetLastError());
}
void SparseCholesky::solve(const double* d_b, double* d_x, hipStream_t st) {
  solve_begin(d_b, st);
  solve_forward_phase(0, st);
  solve_forward_phase(1, st);
  solve_backward_phase(1, st);
  solve_backward_phase(0, st);
  solve_end(d_x, st);
}
double* SparseCholesky::reserve_exchange_tail(size_t n) {
  if (d_xbuf.n < xbuf_count_ + n || !d_xbuf.p) {
    DevBuf<double> bigger;
    bigger.alloc(xbuf_count_ + n);
    G2OHIP_HIP_CHECK(hipMemset(bigger.p, 0, (xbuf_count_ + n) * sizeof(double)));
    d_xbuf = std::move(bigger);
  }
  return d_xbuf.p + xbuf_count_;
}
void SparseCholesky::pack_exchange(hipStream_t st) {
  if (n_xseg_ == 0) return;
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_xbuf.p, 0, xbuf_count_ * sizeof(double), st));
  hipLaunchKernelGGL(seg_copy_kernel, dim3(n_xseg_), dim3(256), 0, st, reinterpret_cast<const SegCopyDev*>(d_xseg.p), d_U.p, d_w.p,
                     d_xbuf.p, 0);
}
void SparseCholesky::unpack_exchange(hipStream_t st) {
  if (n_xseg_ == 0) return;
  hipLaunchKernelGGL(seg_copy_kernel, dim3(n_xseg_), dim3(256), 0, st, reinterpret_cast<const SegCopyDev*>(d_xseg.p), d_U.p, d_w.p,
                     d_xbuf.p, 1);
}
void SparseCholesky::mask_solution(hipStream_t st) {
  const size_t n = (size_t)sym_.nb * bs_;
  if (opt.world <= 1 || n == 0) return;
  hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, d_xmask.p, d_xp.p);
}

bool SparseCholesky::failed(hipStream_t st) {
#ifdef G2OHIP_CHOL_STAMPS
  if (d_dbg.p && getenv("G2OHIP_CHOL_TIMELINE")) {   // (start, end) of every workgroup of the last wave-kernel launch, 10 ns ticks
    std::vector<long long> h(64 * 64 + 6 * (size_t)n_slots_);
    d_dbg.download(h.data(), h.size(), st);
    if (FILE* fp = fopen(getenv("G2OHIP_CHOL_TIMELINE"), "w")) {
      for (int q = 0; q < n_slots_; ++q) fprintf(fp, "%d %lld %lld %lld %lld %lld\n", q, h[64 * 64 + 6 * q], h[64 * 64 + 6 * q + 1], h[64 * 64 + 6 * q + 2], h[64 * 64 + 6 * q + 3], h[64 * 64 + 6 * q + 4]);   // slot, start, end, children there (last front), last front, its parent front
      fclose(fp);
    }
  }
  if (d_dbg.p && getenv("G2OHIP_CHOL_STAMPS_PRINT")) {
    std::vector<long long> h(64 * 64);
    d_dbg.download(h.data(), h.size(), st);
    for (int l = 0; l < dbg_launch_ && l < 64; ++l) {
      const long long* s = h.data() + 64 * l;
      fprintf(stderr, "launch %2d: (nch %lld nU %lld fast %lld)", l, s[1], s[2], s[3]);
      long long prev = s[4];
      for (int k = 1; k < (int)s[0]; ++k) {
        if (s[4 + k] < 0) { fprintf(stderr, " [ns=%lld m=%lld]", -s[4 + k] / 1000, -s[4 + k] % 1000); continue; }
        fprintf(stderr, " %.2f", (s[4 + k] - prev) * 0.01);
        prev = s[4 + k];
      }
      fprintf(stderr, "\n");
    }
  }
#endif
  int h = 0;
  d_status.download(&h, 1, st);
  return failed_with(h, st);
}

bool SparseCholesky::failed_with(int h, hipStream_t st) {
  if (h != 0 && d_ready.p) d_ready.zero(st);   // an aborted dependency-driven launch may leave counters behind
  if (h == 2) {   // a waiting workgroup gave up (should not happen: see analyze): per-level launches from now on
    if (!dep_off_) fprintf(stderr, "g2ohip: a dependency-driven launch gave up waiting; using one launch per level from now on\n");
    dep_off_ = true;
    dep_stalled_ = true;
  }
  return h != 0;
}

}  // namespace g2ohip

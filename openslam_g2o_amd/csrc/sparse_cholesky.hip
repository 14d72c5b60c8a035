// Multifrontal sparse block Cholesky: host symbolic analysis + gfx950 kernels.
// See sparse_cholesky.h for the reference functions this replaces.
#include "sparse_cholesky.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

namespace g2ohip {

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
      // leaf: BFS (Cuthill-McKee like) order from the peripheral node
      for (int k = 0; k < sz; ++k) perm[it.base + k] = W.queue[k];
      W.clear_levels();
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

void SparseCholesky::analyze(int nb, const int* colptr, const int* rowidx, hipStream_t st) {
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
  nested_dissection(nb, xadj, adj, opt.nd_leaf, perm0);
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
  const int max_sn_blocks = std::max(1, opt.max_sn_scalars / bs);
  S.sn_start.clear();
  for (int j = 0; j < nb; ++j) {
    bool merge = false;
    if (j > 0 && S.parent[j - 1] == j && st_[j - 1].size() == st_[j].size() + 1) {
      int width = j - S.sn_start.back();
      if (width < max_sn_blocks) merge = true;
    }
    if (!merge) S.sn_start.push_back(j);
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
    S.L_total += m * np;
    S.U_off[f] = S.U_total;
    S.U_total += nbs * nbs;
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
  // --- level lists: LDS-class fronts first, then global-class
  S.level_ptr.assign(nlev + 1, 0);
  for (int f = 0; f < nf; ++f) S.level_ptr[S.f_level[f] + 1]++;
  for (int l = 0; l < nlev; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
  S.level_fronts.resize(nf);
  launches_.assign(nlev, LevelLaunch());
  std::vector<long long> scratch_off(nf, 0);
  long long scratch_max = 0;
  {
    std::vector<std::vector<int>> lds(nlev), glb(nlev);
    for (int f = 0; f < nf; ++f) {
      size_t m = (size_t)(S.f_ns[f] + S.f_nb[f]) * bs;
      if (m * m * 8 <= opt.lds_front_bytes)
        lds[S.f_level[f]].push_back(f);
      else
        glb[S.f_level[f]].push_back(f);
    }
    for (int l = 0; l < nlev; ++l) {
      LevelLaunch& LL = launches_[l];
      int pos = S.level_ptr[l];
      LL.lds_begin = pos;
      LL.lds_count = (int)lds[l].size();
      for (int f : lds[l]) {
        S.level_fronts[pos++] = f;
        LL.lds_max_m = std::max(LL.lds_max_m, (S.f_ns[f] + S.f_nb[f]) * bs);
      }
      LL.glb_begin = pos;
      LL.glb_count = (int)glb[l].size();
      long long so = 0;
      for (int f : glb[l]) {
        scratch_off[pos] = so;
        long long m = (long long)(S.f_ns[f] + S.f_nb[f]) * bs;
        so += m * m;
        S.level_fronts[pos++] = f;
        LL.glb_max_m = std::max(LL.glb_max_m, (int)m);
      }
      scratch_max = std::max(scratch_max, so);
      for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
        int f = S.level_fronts[k];
        int m = (S.f_ns[f] + S.f_nb[f]) * bs;
        LL.max_m = std::max(LL.max_m, m);
        LL.max_panel = std::max(LL.max_panel, m * S.f_ns[f] * bs);
      }
    }
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
  d_perm.upload(S.perm, st);
  d_L_off.upload(S.L_off, st);
  d_U_off.upload(S.U_off, st);
  d_w_off.upload(S.w_off, st);
  d_scratch_off.upload(scratch_off, st);
  d_L.alloc((size_t)S.L_total);
  d_U.alloc((size_t)S.U_total);
  d_w.alloc((size_t)S.w_total);
  d_y.alloc((size_t)nb * bs);
  d_xp.alloc((size_t)nb * bs);
  d_scratch.alloc((size_t)scratch_max);
  d_status.alloc(1);
  d_status.zero(st);
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
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
  analyzed_ = true;
  stats_.t_symbolic = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// =====================================================================================
// Device kernels
// =====================================================================================
namespace {

constexpr int kFactorThreads = 256;
constexpr int kFactorThreadsGlobal = 512;

// column-major enumeration of the lower-triangular tiles of an nt x nt tile grid
__device__ __forceinline__ void tri_decode(int idx, int nt, int& ti, int& tj) {
  // tiles before column j: j*nt - j(j-1)/2
  float fn = 2.f * nt + 1.f;
  int j = (int)((fn - sqrtf(fn * fn - 8.f * (float)idx)) * 0.5f);
  if (j < 0) j = 0;
  if (j > nt - 1) j = nt - 1;
  while (j > 0 && j * nt - (j * (j - 1)) / 2 > idx) --j;
  while ((j + 1) * nt - ((j + 1) * j) / 2 <= idx) ++j;
  tj = j;
  ti = j + (idx - (j * nt - (j * (j - 1)) / 2));
}

// One workgroup factorises one frontal matrix.
//   F: m x m column-major (ld = m), lower triangle used.  LDS or an HBM scratch slab.
// Steps: zero, assemble original blocks, extend-add the children's update matrices,
// partial Cholesky of the ns pivot block columns (blocked by BS), write L panel and the
// update matrix U.
template <int BS, bool USE_LDS>
__global__ void __launch_bounds__(USE_LDS ? kFactorThreads : kFactorThreadsGlobal) front_factor_kernel(CholPlanDev P, const int* __restrict__ fronts, const double* __restrict__ A,
                                    double* __restrict__ scratch, const long long* __restrict__ scratch_off) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = (BS % 3 == 0) ? 3 : BS;  // register tile edge of the trailing update
  const int f = fronts[blockIdx.x];
  const int ns = P.f_ns[f], nbd = P.f_nb[f];
  const int m = (ns + nbd) * BS, npiv = ns * BS, ld = m;
  double* F = USE_LDS ? smem : (scratch + scratch_off[blockIdx.x]);
  const int tid = threadIdx.x, NT = blockDim.x;

  for (int i = tid; i < m * m; i += NT) F[i] = 0.0;
  __syncthreads();
  // ---- original entries (each block lands on a distinct tile)
  {
    const int a0 = P.asm_off[f], na = P.asm_off[f + 1] - a0;
    for (int t = tid; t < na * BS * BS; t += NT) {
      const int e = t / (BS * BS), rc = t - e * (BS * BS);
      const int r = rc % BS, c = rc / BS;
      const int q = P.asm_q[a0 + e], pos = P.asm_pos[a0 + e];
      const int lr = pos & 0x7fff, lc = (pos >> 15) & 0x7fff, tr = (pos >> 30) & 1;
      const double v = tr ? A[(size_t)q * BS * BS + c + BS * r] : A[(size_t)q * BS * BS + r + BS * c];
      F[(lr * BS + r) + (size_t)ld * (lc * BS + c)] = v;
    }
  }
  __syncthreads();
  // ---- extend-add of the children (sequential over children: tiles may overlap)
  for (int ch = P.child_off[f]; ch < P.child_off[f + 1]; ++ch) {
    const int c = P.children[ch];
    const int nbc = P.f_nb[c] * BS;
    const double* Uc = P.U + P.U_off[c];
    const int* rel = P.rel + P.rel_off[c];
    for (int t = tid; t < nbc * nbc; t += NT) {
      const int i = t % nbc, j = t / nbc;
      if (i < j) continue;
      const int li = rel[i / BS] * BS + (i % BS), lj = rel[j / BS] * BS + (j % BS);
      F[li + (size_t)ld * lj] += Uc[t];
    }
    __syncthreads();
  }
  // ---- partial Cholesky, one pivot block (BS columns) per step
  for (int kb = 0; kb < ns; ++kb) {
    const int k0 = kb * BS;
    double Lk[BS][BS];
    double inv[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c)
#pragma unroll
      for (int r = 0; r < BS; ++r) Lk[r][c] = (r >= c) ? F[(k0 + r) + (size_t)ld * (k0 + c)] : 0.0;
    __syncthreads();  // everyone holds the diagonal block before thread 0 overwrites it
    bool bad = false;
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      double d = Lk[c][c];
      if (!(d > 0.0)) {
        bad = true;
        d = 1.0;
      }
      const double r = rsqrt(d);
      inv[c] = r;
      Lk[c][c] = d * r;
#pragma unroll
      for (int i = c + 1; i < BS; ++i) Lk[i][c] *= r;
#pragma unroll
      for (int j = c + 1; j < BS; ++j)
#pragma unroll
        for (int i = j; i < BS; ++i) Lk[i][j] -= Lk[i][c] * Lk[j][c];
    }
    if (tid == 0) {
      if (bad) *P.status = 1;
#pragma unroll
      for (int c = 0; c < BS; ++c)
#pragma unroll
        for (int r = 0; r < BS; ++r) F[(k0 + r) + (size_t)ld * (k0 + c)] = (r >= c) ? Lk[r][c] : 0.0;
    }
    // rows below the diagonal block: x * Lkk' = row
    for (int i = k0 + BS + tid; i < m; i += NT) {
      double x[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) x[c] = F[i + (size_t)ld * (k0 + c)];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        double v = x[c];
#pragma unroll
        for (int q = 0; q < c; ++q) v -= x[q] * Lk[c][q];
        x[c] = v * inv[c];
      }
#pragma unroll
      for (int c = 0; c < BS; ++c) F[i + (size_t)ld * (k0 + c)] = x[c];
    }
    __syncthreads();
    // trailing update with T x T register tiles over the lower triangle
    const int r0 = k0 + BS;
    const int nt = (m - r0) / T;
    const int ntiles = nt * (nt + 1) / 2;
    for (int idx = tid; idx < ntiles; idx += NT) {
      int ti, tj;
      tri_decode(idx, nt, ti, tj);
      const int i0 = r0 + ti * T, j0 = r0 + tj * T;
      double acc[T][T];
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = 0.0;
#pragma unroll
      for (int q = 0; q < BS; ++q) {
        double av[T], bv[T];
#pragma unroll
        for (int a = 0; a < T; ++a) av[a] = F[(i0 + a) + (size_t)ld * (k0 + q)];
#pragma unroll
        for (int b = 0; b < T; ++b) bv[b] = F[(j0 + b) + (size_t)ld * (k0 + q)];
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int b = 0; b < T; ++b) acc[a][b] += av[a] * bv[b];
      }
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int a = 0; a < T; ++a) F[(i0 + a) + (size_t)ld * (j0 + b)] -= acc[a][b];
    }
    __syncthreads();
  }
  // ---- write L panel (m x npiv) and update matrix (lower part of the trailing block)
  double* Lg = P.L + P.L_off[f];
  for (int t = tid; t < m * npiv; t += NT) Lg[t] = F[t];  // ld == m: identical layout
  const int nbs = nbd * BS;
  double* Ug = P.U + P.U_off[f];
  for (int t = tid; t < nbs * nbs; t += NT) {
    const int i = t % nbs, j = t / nbs;
    Ug[t] = F[(npiv + i) + (size_t)ld * (npiv + j)];
  }
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

// Forward sweep for one front: y1 = L11 \ (b1 + children), w = (children on boundary) - L21 y1.
// LDS: panel (optional) + t[m] + ysol[npiv]
template <bool PANEL_LDS>
__global__ void __launch_bounds__(256) front_forward_kernel(CholPlanDev P, const int* __restrict__ fronts, int bs, const double* __restrict__ bperm,
                                     double* __restrict__ y, int panel_cap) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int f = fronts[blockIdx.x];
  const int ns = P.f_ns[f], nbd = P.f_nb[f];
  const int m = (ns + nbd) * bs, npiv = ns * bs, c0 = P.f_c0[f];
  const int tid = threadIdx.x, NT = blockDim.x;
  double* Lp = smem;
  double* t = smem + (PANEL_LDS ? panel_cap : 0);
  const double* Lg = P.L + P.L_off[f];
  if (PANEL_LDS)
    for (int i = tid; i < m * npiv; i += NT) Lp[i] = Lg[i];
  const double* Lx = PANEL_LDS ? Lp : Lg;
  for (int i = tid; i < m; i += NT) t[i] = (i < npiv) ? bperm[(size_t)c0 * bs + i] : 0.0;
  __syncthreads();
  for (int ch = P.child_off[f]; ch < P.child_off[f + 1]; ++ch) {
    const int c = P.children[ch];
    const int nbc = P.f_nb[c] * bs;
    const double* wc = P.w + P.w_off[c];
    const int* rel = P.rel + P.rel_off[c];
    for (int i = tid; i < nbc; i += NT) t[rel[i / bs] * bs + (i % bs)] += wc[i];
    __syncthreads();
  }
  double* ys = t + m;
  for (int k = 0; k < npiv; ++k) {
    const double yk = t[k] / Lx[k + (size_t)m * k];   // t[k] is final: only rows > k are touched below
    if (tid == 0) ys[k] = yk;
    for (int i = k + 1 + tid; i < m; i += NT) t[i] -= Lx[i + (size_t)m * k] * yk;
    __syncthreads();
  }
  for (int i = tid; i < npiv; i += NT) y[(size_t)c0 * bs + i] = ys[i];
  double* wf = P.w + P.w_off[f];
  for (int i = tid; i < nbd * bs; i += NT) wf[i] = t[npiv + i];
}

// Backward sweep for one front: x1 = L11' \ (y1 - L21' x_boundary)
template <bool PANEL_LDS>
__global__ void __launch_bounds__(256) front_backward_kernel(CholPlanDev P, const int* __restrict__ fronts, int bs, const double* __restrict__ y,
                                      double* __restrict__ xp, int panel_cap) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int f = fronts[blockIdx.x];
  const int ns = P.f_ns[f], nbd = P.f_nb[f];
  const int m = (ns + nbd) * bs, npiv = ns * bs, c0 = P.f_c0[f];
  const int tid = threadIdx.x, NT = blockDim.x;
  double* Lp = smem;
  double* t = smem + (PANEL_LDS ? panel_cap : 0);
  const double* Lg = P.L + P.L_off[f];
  if (PANEL_LDS)
    for (int i = tid; i < m * npiv; i += NT) Lp[i] = Lg[i];
  const double* Lx = PANEL_LDS ? Lp : Lg;
  const int* rows = P.rows + P.rows_off[f];
  for (int i = tid; i < m; i += NT)
    t[i] = (i < npiv) ? y[(size_t)c0 * bs + i] : xp[(size_t)rows[(i - npiv) / bs] * bs + ((i - npiv) % bs)];
  __syncthreads();
  // boundary contribution: t[k] -= sum_{i>=npiv} L[i,k] t[i]
  for (int k = tid; k < npiv; k += NT) {
    double s = 0.0;
    for (int i = npiv; i < m; ++i) s += Lx[i + (size_t)m * k] * t[i];
    t[k] -= s;
  }
  __syncthreads();
  double* xs = t + m;
  for (int k = npiv - 1; k >= 0; --k) {
    const double xk = t[k] / Lx[k + (size_t)m * k];   // t[k] is final: only entries < k are touched below
    if (tid == 0) xs[k] = xk;
    for (int j = tid; j < k; j += NT) t[j] -= Lx[k + (size_t)m * j] * xk;
    __syncthreads();
  }
  for (int i = tid; i < npiv; i += NT) xp[(size_t)c0 * bs + i] = xs[i];
}

template <int BS>
void launch_factor_level(const CholPlanDev& P, const int* d_fronts, const long long* d_scratch_off, double* d_scratch,
                         const double* dA, int lds_begin, int lds_count, int lds_max_m, int glb_begin, int glb_count,
                         hipStream_t st) {
  if (lds_count > 0) {
    size_t sh = (size_t)lds_max_m * lds_max_m * sizeof(double);
    hipLaunchKernelGGL((front_factor_kernel<BS, true>), dim3(lds_count), dim3(kFactorThreads), sh, st, P, d_fronts + lds_begin,
                       dA, d_scratch, d_scratch_off + lds_begin);
  }
  if (glb_count > 0) {
    hipLaunchKernelGGL((front_factor_kernel<BS, false>), dim3(glb_count), dim3(kFactorThreadsGlobal), 0, st, P,
                       d_fronts + glb_begin, dA, d_scratch, d_scratch_off + glb_begin);
  }
}

}  // namespace

void SparseCholesky::factor(const double* dA, hipStream_t st) {
  if (!analyzed_) throw StateFailure("SparseCholesky::factor before analyze");
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_status.p, 0, sizeof(int), st));
  static bool attr_done = false;
  if (!attr_done) {
    // allow > 64 KiB dynamic LDS for the LDS-resident front kernels
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)front_factor_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  for (const LevelLaunch& LL : launches_) {
    switch (bs_) {
      case 3:
        launch_factor_level<3>(plan_, d_level_fronts.p, d_scratch_off.p, d_scratch.p, dA, LL.lds_begin, LL.lds_count, LL.lds_max_m,
                               LL.glb_begin, LL.glb_count, st);
        break;
      case 6:
        launch_factor_level<6>(plan_, d_level_fronts.p, d_scratch_off.p, d_scratch.p, dA, LL.lds_begin, LL.lds_count, LL.lds_max_m,
                               LL.glb_begin, LL.glb_count, st);
        break;
      case 7:
        launch_factor_level<7>(plan_, d_level_fronts.p, d_scratch_off.p, d_scratch.p, dA, LL.lds_begin, LL.lds_count, LL.lds_max_m,
                               LL.glb_begin, LL.glb_count, st);
        break;
      default:
        throw ArgFailure("SparseCholesky: unsupported block size (3, 6, 7)");
    }
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void SparseCholesky::solve(const double* d_b, double* d_x, hipStream_t st) {
  if (!analyzed_) throw StateFailure("SparseCholesky::solve before analyze");
  const int n = sym_.nb * bs_;
  if (n == 0) return;
  const int thr = 256;
  hipLaunchKernelGGL(permute_in_kernel, dim3((n + thr - 1) / thr), dim3(thr), 0, st, sym_.nb, bs_, d_perm.p, d_b, d_xp.p);
  const size_t panel_limit = 48 * 1024;  // bytes of L panel staged in LDS
  auto run = [&](const LevelLaunch& LL, bool fwd) {
    int count = LL.lds_count + LL.glb_count;
    if (count == 0) return;
    const int* fl = d_level_fronts.p + LL.lds_begin;
    bool panel = (size_t)LL.max_panel * 8 <= panel_limit;
    int cap = panel ? LL.max_panel : 0;
    size_t sh = ((size_t)cap + 2 * (size_t)LL.max_m + 8) * sizeof(double);
    int nthreads = LL.max_m <= 64 ? 64 : (LL.max_m <= 128 ? 128 : 256);
    if (fwd) {
      if (panel)
        hipLaunchKernelGGL((front_forward_kernel<true>), dim3(count), dim3(nthreads), sh, st, plan_, fl, bs_, d_xp.p, d_y.p, cap);
      else
        hipLaunchKernelGGL((front_forward_kernel<false>), dim3(count), dim3(nthreads), sh, st, plan_, fl, bs_, d_xp.p, d_y.p, cap);
    } else {
      if (panel)
        hipLaunchKernelGGL((front_backward_kernel<true>), dim3(count), dim3(nthreads), sh, st, plan_, fl, bs_, d_y.p, d_xp.p, cap);
      else
        hipLaunchKernelGGL((front_backward_kernel<false>), dim3(count), dim3(nthreads), sh, st, plan_, fl, bs_, d_y.p, d_xp.p, cap);
    }
  };
  // forward: d_xp holds the permuted rhs, y receives the pivot solutions
  for (size_t l = 0; l < launches_.size(); ++l) run(launches_[l], true);
  // backward: overwrite d_xp with the solution, root level first
  for (size_t l = launches_.size(); l-- > 0;) run(launches_[l], false);
  hipLaunchKernelGGL(permute_out_kernel, dim3((n + thr - 1) / thr), dim3(thr), 0, st, sym_.nb, bs_, d_perm.p, d_xp.p, d_x);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

bool SparseCholesky::failed(hipStream_t st) {
  int h = 0;
  d_status.download(&h, 1, st);
  return h != 0;
}

}  // namespace g2ohip

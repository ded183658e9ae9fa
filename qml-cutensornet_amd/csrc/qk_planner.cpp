// qk_planner.cpp -- the host-side planner of a Gram share: work model, contraction order, tiles, ranks, queues.
//
// What it replaces (reference G = gpu_backend/kernel_state_ansatz.py): the chunk / round-robin bookkeeping of G:154, 184,
// 331-334, 384-385 and -- north star -- "contraction order chosen greedily on the host" (G:380 delegates it to cuTensorNet).
// Plain C++ (no HIP): part of libqkgram.so, and built host-only under the sanitizers by tests/host_san.
//
// The reference times its tiling phase INCLUDING the set-up (G:322, 432-434), so the planner sits on the cold path of every Gram:
// it is written to cost O(pairs x sites) SIMD multiply-adds spread over the host's cores, not a scalar pass per pair:
//   * every quantity of a pair but the algorithmic flop count (a min per site) is BILINEAR in per-state site vectors -- padded
//     work 16 sum_k (A_k (B_k B_k+1) + (A_k A_k+1) B_k+1), matrix instructions 6 sum_k (A_k/16 (B_k+1/16 ceil(b_k/4)) + ...) -- which
//     are laid out once per state (struct SiteTab) so that a pair costs a handful of fused multiply-adds per site, vectorised;
//   * pass 1 (all pairs of the Gram, needed by every rank to deal the tiles): matrix instructions in both orders (which state
//     plays Y) and the padded work of the cheaper one; pass 2 (this rank's pairs only): algorithmic flops, bytes, LDS-fit shares;
//   * both passes run on std::thread workers over disjoint ranges; every sum is a sum of integers below 2^53 held in doubles,
//     so the result does not depend on the thread count or on the vector width (tests/test_host_logic.py pins it).
#include "qk_plan.h"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

static inline int pad16(int x) { return qk_pad16(x); }

// ----------------------------------------------------------------------------------------
// worker threads
// ----------------------------------------------------------------------------------------
static int plan_threads() {
  if (const char* e = std::getenv("QK_PLAN_THREADS")) return std::max(1, std::min(64, std::atoi(e)));
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);  // the cores this process may run on (a rank's share of the host)
  return std::max(1, std::min(16, n));
}

// f(begin, end) over [0, n) in contiguous ranges, one per thread
template <typename F>
static void par_ranges(const int threads, const int64_t n, const F& f) {
  const int t = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
  if (t == 1) {
    f((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve((size_t)t - 1);
  for (int w = 1; w < t; ++w) pool.emplace_back([&, w] { f(n * w / t, n * (w + 1) / t); });
  f((int64_t)0, n / t);
  for (std::thread& th : pool) th.join();
}

// ----------------------------------------------------------------------------------------
// work model.  Algorithmic flops of one overlap: 8 real flops per complex multiply-add, cheaper
// association per site.  Padded: every bond rounded up to the 16-wide tile.  Matrix instructions of the site-fused sweep for the
// pair (x = a, y = b): per site (a^/16)(b'^/16) tiles of T, each ceil(b/4) k-steps in phase 1 and (a'^/16) column blocks x
// ceil(a/4)-bounded k-steps in phase 2 (x 3 for the 3M product, x 2 for p).
// ----------------------------------------------------------------------------------------
struct SiteTab {  // per state and site k (left bond t0 = chi_k, right bond t1 = chi_k+1; P = padded to 16), as doubles: [state][k]
  int n = 0;
  std::vector<double> t0, t1, t01, P0, P1, P01, T0, T1, TK;  // T0 = P0 / 16, T1 = P1 / 16, TK = T1 ceil(t0 / 4)
  std::vector<double> bytes, weight;                        // per state: 32 sum t0 t1;  sum P0 P1 (P0 + P1)
  std::vector<int> max_pad;                                 // per state: largest padded bond
  void build(const int n_states, const int n_sites, const int32_t* dims) {
    n = n_sites;
    const size_t tot = (size_t)n_states * n_sites;
    for (std::vector<double>* v : {&t0, &t1, &t01, &P0, &P1, &P01, &T0, &T1, &TK}) v->assign(tot, 0.0);
    bytes.assign((size_t)n_states, 0.0), weight.assign((size_t)n_states, 0.0), max_pad.assign((size_t)n_states, 0);
    for (int s = 0; s < n_states; ++s) {
      const int32_t* d = dims + (int64_t)s * (n_sites + 1);
      double by = 0, w = 0;
      int mp = 0;
      for (int k = 0; k < n_sites; ++k) {
        const size_t e = (size_t)s * n_sites + k;
        const int a0 = d[k], a1 = d[k + 1], A0 = pad16(a0), A1 = pad16(a1);
        t0[e] = a0, t1[e] = a1, t01[e] = (double)a0 * a1;
        P0[e] = A0, P1[e] = A1, P01[e] = (double)A0 * A1;
        T0[e] = A0 / 16, T1[e] = A1 / 16, TK[e] = (double)(A1 / 16) * ((a0 + 3) / 4);
        by += 16.0 * 2 * ((double)a0 * a1);
        w += (double)A0 * A1 * ((double)A0 + A1);
        mp = std::max(mp, std::max(A0, A1));
      }
      bytes[(size_t)s] = by, weight[(size_t)s] = w, max_pad[(size_t)s] = mp;
    }
  }
};

// pass 1 of a pair: matrix instructions with x = state i of X, y = state j of Y (c_xy), in the other order (c_yx: only meaningful
// for a symmetric plan, where both states come from one table), and the padded work in both orders
struct Pass1 {
  double c_xy, c_yx, fp_xy, fp_yx;
};
#if defined(__x86_64__)
#define QK_SIMD_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define QK_SIMD_CLONES
#endif
QK_SIMD_CLONES static Pass1 pair_pass1(const int n, const double* __restrict__ xT0, const double* __restrict__ xT1, const double* __restrict__ xTK, const double* __restrict__ xP0,
                                       const double* __restrict__ xP1, const double* __restrict__ xP01, const double* __restrict__ yT0, const double* __restrict__ yT1,
                                       const double* __restrict__ yTK, const double* __restrict__ yP0, const double* __restrict__ yP1, const double* __restrict__ yP01) {
  double cxy = 0, cyx = 0, fxy = 0, fyx = 0;
#pragma omp simd reduction(+ : cxy, cyx, fxy, fyx)
  for (int k = 0; k < n; ++k) {
    cxy += xT0[k] * yTK[k] + yT1[k] * xTK[k];
    cyx += yT0[k] * xTK[k] + xT1[k] * yTK[k];
    fxy += xP0[k] * yP01[k] + xP01[k] * yP1[k];
    fyx += yP0[k] * xP01[k] + yP01[k] * xP1[k];
  }
  return Pass1{6 * cxy, 6 * cyx, 16 * fxy, 16 * fyx};
}

// pass 2 of a pair (x = a, y = b): algorithmic flops, and the padded work that sits in sites whose X and X' fit `cap_two` / `cap_fit`
// elements
struct Pass2 {
  double f, ft, fn;
};
QK_SIMD_CLONES static Pass2 pair_pass2(const int n, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ a01, const double* __restrict__ A0,
                                       const double* __restrict__ A1, const double* __restrict__ A01, const double* __restrict__ b0, const double* __restrict__ b1,
                                       const double* __restrict__ b01, const double* __restrict__ B0, const double* __restrict__ B1, const double* __restrict__ B01, const double cap_two,
                                       const double cap_fit) {
  double f = 0, ft = 0, fn = 0;
#pragma omp simd reduction(+ : f, ft, fn)
  for (int k = 0; k < n; ++k) {
    const double f1 = a0[k] * b01[k] + a01[k] * b1[k];  // (a b 2 b' + 2 a a' b') / 2
    const double f2 = b0[k] * a01[k] + b01[k] * a1[k];  // (a b 2 a' + 2 b a' b') / 2
    f += f1 < f2 ? f1 : f2;
    const double w = A0[k] * B01[k] + A01[k] * B1[k];
    const double x0 = A0[k] * B0[k], x1 = A1[k] * B1[k];
    ft += (x0 <= cap_two && x1 <= cap_two) ? w : 0.0;
    fn += (x0 <= cap_fit && x1 <= cap_fit) ? w : 0.0;
  }
  return Pass2{16 * f, 16 * ft, 16 * fn};
}

// the scalar forms (the flat list and the QK_PLAN_QUADS plans, which are not on the default path)
static void pair_work(int n, const int32_t* a, const int32_t* b, const double plan_fit, double* flops, double* padded, double* bytes, double* fit_two = nullptr, double* fit_narrow = nullptr) {
  double f = 0, fp = 0, by = 0, ft = 0, fn = 0;
  for (int k = 0; k < n; ++k) {
    const double a0 = a[k], a1 = a[k + 1], b0 = b[k], b1 = b[k + 1];
    const double f1 = a0 * b0 * 2 * b1 + 2 * a0 * a1 * b1;
    const double f2 = a0 * b0 * 2 * a1 + 2 * b0 * a1 * b1;
    f += 8 * std::min(f1, f2);
    const double A0 = pad16(a[k]), A1 = pad16(a[k + 1]), B0 = pad16(b[k]), B1 = pad16(b[k + 1]);
    fp += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);
    if (A0 * B0 <= QKF_XCAP_TWO && A1 * B1 <= QKF_XCAP_TWO) ft += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);  // X and X' of this site fit the smaller buffer
    if (A0 * B0 <= plan_fit && A1 * B1 <= plan_fit) fn += 8 * (A0 * B0 * 2 * B1 + 2 * A0 * A1 * B1);          // ... with room to spare (QK_PLAN_FIT)
    by += 16.0 * 2 * (a0 * a1 + b0 * b1);
  }
  if (fit_two) *fit_two = ft;
  if (fit_narrow) *fit_narrow = fn;
  *flops = f;
  *padded = fp;
  *bytes = by + 8;
}
static double fused_cost(int n, const int32_t* a, const int32_t* b) {
  double c = 0;
  for (int k = 0; k < n; ++k) {
    const double A0 = pad16(a[k]) / 16, A1 = pad16(a[k + 1]) / 16, B1 = pad16(b[k + 1]) / 16;
    const double kb = (b[k] + 3) / 4, ka = (a[k] + 3) / 4;
    c += 6 * A0 * B1 * kb + 6 * B1 * A1 * ka;
  }
  return c;
}

// How many sites at either end of the chain the site-fused sweep should take from edge blocks (qk_fused.h: qkf_edge_prefix /
// qkf_edge_suffix) instead of walking them: the k that minimises, over a sample of this rank's pairs, the matrix instructions of
// the sweep plus a fixed cost per site walked (two barriers, set-up, load latencies: 375 instructions' worth = 2.5 us of a 12-wave
// workgroup, measured on uniform small chains, tools/site_overhead.py).  A block product costs tiles(a_k) tiles(b_k) 2^k / 4 x 3
// instructions x 2 (its tiles stream both blocks from L2).  Calibrated on the two headline sets with merged steps in the middle of
// the chain: 60 qubits x 6 layers, k = 6 / 7 / 8 / 9 measured 403.8 / 393.8 / 396.2 / 401.8 ms (model, relative to k = 8: 1.008 /
// 0.997 / 1 / 1.036); 40 qubits x 4 layers, k = 5 .. 9: 13.23 / 13.11 / 12.80 / 12.85 / 13.83 ms (model 1.054 / 1.021 / 0.995 / 1 /
// 1.087).  QK_EDGE=0 disables, QK_EDGE=k forces.  This is a contraction order chosen on the host (north star; reference call site
// G:380): while the bonds still grow like 2^k, the ends of the two states are cheaper to contract across their physical legs than
// along the chain.
static int choose_edge_k(const int n_sites, const int32_t* x_dims, const int32_t* y_dims, const std::vector<int32_t>& pairs) {
  constexpr int KMIN = 4, KMAX = 9;  // K = 2^k >= 16 (four k-steps in flight); 2^9 rows per block at most
  if (const char* e = std::getenv("QK_EDGE")) {
    const int v = std::atoi(e);
    if (v <= 0) return 0;
    return (v >= KMIN && v <= KMAX && n_sites >= 2 * v + 2) ? v : 0;
  }
  const int64_t np = (int64_t)pairs.size() / 2;
  if (np == 0 || n_sites < 2 * KMIN + 2) return 0;
  const int stride = n_sites + 1;
  const int64_t step = std::max<int64_t>(1, np / 512);
  const double over = 375.0;
  auto t16 = [](const int v) { return (double)((v + 15) / 16); };
  std::vector<double> total((size_t)KMAX + 1, 0.0);
  std::vector<double> site((size_t)n_sites);
  for (int64_t t = 0; t < np; t += step) {
    const int32_t* a = x_dims + (int64_t)pairs[2 * t] * stride;
    const int32_t* b = y_dims + (int64_t)pairs[2 * t + 1] * stride;
    for (int k = 0; k < n_sites; ++k)
      site[(size_t)k] = 6 * t16(a[k]) * t16(b[k + 1]) * ((b[k] + 3) / 4) + 6 * t16(b[k + 1]) * t16(a[k + 1]) * ((a[k] + 3) / 4) + over;
    double chain = 0;
    for (double v : site) chain += v;
    total[0] += chain;
    double ends = 0;  // cost of the sites the edges replace
    for (int k = 1; k <= KMAX && n_sites >= 2 * k + 2; ++k) {
      ends += site[(size_t)k - 1] + site[(size_t)(n_sites - k)];
      if (k < KMIN) continue;
      const double blocks = 2.0 * 3.0 * ((1 << k) / 4) * (t16(a[k]) * t16(b[k]) + t16(a[n_sites - k]) * t16(b[n_sites - k])) + 2 * over;
      total[(size_t)k] += chain - ends + blocks;
    }
  }
  int best = 0;
  for (int k = KMIN; k <= KMAX && n_sites >= 2 * k + 2; ++k)
    if (total[(size_t)k] > 0 && total[(size_t)k] < total[(size_t)best] * 0.995) best = k;
  return best;
}

// ----------------------------------------------------------------------------------------
// The tiled plan: XCD-aware work queues (default; QK_PLAN_XCD=0 or an explicit locality `block` selects the flat list).
// An MI355X has 8 XCDs with a private 4 MiB L2 each, and blocks are dealt to them round-robin.  With one cost-ordered list
// the 32 (or 64) workgroups that share an L2 stream 64 unrelated states through it (measured hit rate 39 %, 2.5-3 x the
// algorithmic bytes at the L2 <-> fabric boundary).  Here the states are sorted by weight, the Gram is cut into tiles of T x T
// pairs in that order -- pairs of a tile share their T x states and T y states and cost about the same, so the workgroups
// sweeping a tile walk the chain at a similar pace --, the tiles are dealt (heaviest first, to the least loaded) to the
// ranks and, per class of pairs, to 8 queues; a workgroup drains the queue of its own XCD and then steals (qk_pull).
// ----------------------------------------------------------------------------------------
// The part of the tiled plan that every rank needs and that is the same for all of them (pass 1 and the deal): made once per
// qk_plan_create -- or once for ALL ranks of a one-process communicator (qk_plan_create_all).
struct TiledItem {
  int32_t i, j;
  double fp;
};
struct TiledTile {
  int64_t start, count;
  double cost;
};
struct TiledGlobal {
  SiteTab xt, yt_own;
  bool sym = false;
  int n_sites = 0, nx = 0, ny = 0, world = 1;
  const int32_t *x_dims = nullptr, *y_dims = nullptr;
  std::vector<TiledItem> items;
  std::vector<TiledTile> tiles;
  std::vector<int> parent;               // the Gram tile a (possibly cut) tile came from
  std::vector<std::vector<int>> mine;    // per rank: its tiles, heaviest first
  std::vector<int64_t> per_rank;
  int64_t n_items = 0;
  const SiteTab& yt() const { return sym ? xt : yt_own; }
};

static void tiled_global(TiledGlobal& G, const int n_sites, const int nx, const int32_t* x_dims, const int ny, const int32_t* y_dims, const bool sym, const bool orient, const int world,
                         const int T, const int threads) {
  G.sym = sym, G.n_sites = n_sites, G.nx = nx, G.ny = ny, G.world = world, G.x_dims = x_dims, G.y_dims = y_dims;
  SiteTab& xt = G.xt;
  xt.build(nx, n_sites, x_dims);
  if (!sym) G.yt_own.build(ny, n_sites, y_dims);
  const SiteTab& yt = G.yt();
  typedef TiledItem Item;
  typedef TiledTile Tile;
  std::vector<Item>& items = G.items;
  std::vector<Tile>& tiles = G.tiles;
  std::vector<int>& parent = G.parent;
  auto order_of = [&](const int n, const SiteTab& tab) {
    std::vector<int> o((size_t)n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](const int u, const int v) { return tab.weight[(size_t)u] > tab.weight[(size_t)v]; });
    return o;
  };
  const std::vector<int> ox = order_of(nx, xt), oy = sym ? ox : order_of(ny, yt);
  // the tiles in (bj, bi) order with their item ranges (counts are known before any work is priced)
  // (tiles are Tx x-positions by Ty y-positions: square by default; QK_PLAN_TILE_X / QK_PLAN_TILE_Y cut them oblong, e.g. 32 x 2 --
  //  32 pairs that share each of two y states, whose tensors are the operand every wave of a workgroup re-reads in phase 1)
  int Tx = T, Ty = T;
  if (const char* e = std::getenv("QK_PLAN_TILE_X")) Tx = std::max(1, std::min(128, std::atoi(e)));
  if (const char* e = std::getenv("QK_PLAN_TILE_Y")) Ty = std::max(1, std::min(128, std::atoi(e)));
  // the order of contraction of a symmetric plan's pairs: per pair (default), or one choice per tile (QK_PLAN_ORIENT_TILE=1: the pairs of
  // a tile then really share their y states)
  const bool orient_tile = orient && std::getenv("QK_PLAN_ORIENT_TILE") && std::atoi(std::getenv("QK_PLAN_ORIENT_TILE")) != 0;
  const int nbx = (nx + Tx - 1) / Tx, nby = (ny + Ty - 1) / Ty;
  std::vector<int32_t> tile_b;  // (bi, bj) of each tile
  int64_t n_items = 0;
  for (int bj = 0; bj < nby; ++bj)
    for (int bi = 0; bi < nbx; ++bi) {
      const int u0 = bi * Tx, u1 = std::min(nx, (bi + 1) * Tx), v0 = bj * Ty, v1 = std::min(ny, (bj + 1) * Ty);
      int64_t cnt = 0;
      for (int v = v0; v < v1; ++v) cnt += sym ? std::max(0, std::min(u1, v + 1) - u0) : (u1 - u0);  // symmetric: positions u <= v
      if (cnt == 0) continue;
      tiles.push_back(Tile{n_items, cnt, 0.0});
      tile_b.push_back(bi), tile_b.push_back(bj);
      n_items += cnt;
    }
  items.assign((size_t)n_items, Item{0, 0, 0.0});
  G.n_items = n_items;
  // ---- pass 1, all pairs: the cheaper order of contraction and its padded work
  par_ranges(threads, (int64_t)tiles.size(), [&](const int64_t t_lo, const int64_t t_hi) {
    for (int64_t t = t_lo; t < t_hi; ++t) {
      const int bi = tile_b[2 * (size_t)t], bj = tile_b[2 * (size_t)t + 1];
      const int64_t q0 = tiles[(size_t)t].start;
      int64_t q = q0;
      double cost = 0, cost_sw = 0, c_keep = 0, c_swap = 0;
      for (int v = bj * Ty; v < std::min(ny, (bj + 1) * Ty); ++v)
        for (int u = bi * Tx; u < std::min(nx, (bi + 1) * Tx); ++u) {
          if (sym && u > v) continue;  // positions in the weight order: every unordered pair once
          int xi = ox[(size_t)u], yj = oy[(size_t)v];
          if (sym && !orient && xi > yj) std::swap(xi, yj);  // the plain symmetric list names a pair as i <= j
          const size_t ex = (size_t)xi * n_sites, ey = (size_t)yj * n_sites;
          const Pass1 r = pair_pass1(n_sites, &xt.T0[ex], &xt.T1[ex], &xt.TK[ex], &xt.P0[ex], &xt.P1[ex], &xt.P01[ex], &yt.T0[ey], &yt.T1[ey], &yt.TK[ey], &yt.P0[ey], &yt.P1[ey], &yt.P01[ey]);
          double fp = r.fp_xy;
          if (orient_tile) c_keep += r.c_xy, c_swap += r.c_yx, cost_sw += r.fp_yx;
          else if (orient && xi != yj && r.c_yx < r.c_xy) std::swap(xi, yj), fp = r.fp_yx;
          items[(size_t)q++] = Item{xi, yj, fp};
          cost += fp;
        }
      if (orient_tile && c_swap < c_keep) {  // the whole tile in the other order
        for (int64_t e = q0; e < q; ++e) {
          Item& it = items[(size_t)e];
          const size_t ex = (size_t)it.j * n_sites, ey = (size_t)it.i * n_sites;
          const Pass1 r = pair_pass1(n_sites, &xt.T0[ex], &xt.T1[ex], &xt.TK[ex], &xt.P0[ex], &xt.P1[ex], &xt.P01[ex], &yt.T0[ey], &yt.T1[ey], &yt.TK[ey], &yt.P0[ey], &yt.P1[ey], &yt.P01[ey]);
          std::swap(it.i, it.j), it.fp = r.fp_xy;
        }
        cost = cost_sw;
      }
      tiles[(size_t)t].cost = cost;
    }
  });
  std::vector<int> by_cost(tiles.size());
  std::iota(by_cost.begin(), by_cost.end(), 0);
  std::stable_sort(by_cost.begin(), by_cost.end(), [&](const int u, const int v) { return tiles[(size_t)u].cost > tiles[(size_t)v].cost; });
  parent.assign(tiles.size(), 0);
  std::iota(parent.begin(), parent.end(), 0);
  if (world > 1) {
    // several ranks: the lightest tiles (the last 3 % of the work) are dealt pair by pair, so that the shares end level to a
    // pair's cost instead of a tile's (cut into one-pair tiles here; they keep their place behind the whole tiles)
    double total = 0, acc = 0;
    for (const Tile& t : tiles) total += t.cost;
    std::vector<Tile> cut;
    std::vector<int> order, par;
    for (const int t : by_cost) {
      acc += tiles[(size_t)t].cost;
      if (acc <= 0.97 * total || tiles[(size_t)t].count == 1) {
        order.push_back((int)cut.size());
        cut.push_back(tiles[(size_t)t]), par.push_back(t);
      } else {
        std::vector<int64_t> q((size_t)tiles[(size_t)t].count);
        std::iota(q.begin(), q.end(), tiles[(size_t)t].start);
        std::stable_sort(q.begin(), q.end(), [&](const int64_t u, const int64_t v) { return items[(size_t)u].fp > items[(size_t)v].fp; });
        for (const int64_t e : q) {
          order.push_back((int)cut.size());
          cut.push_back(Tile{e, 1, items[(size_t)e].fp}), par.push_back(t);
        }
      }
    }
    tiles.swap(cut), by_cost.swap(order), parent.swap(par);
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](const int u, const int v) { return tiles[(size_t)u].cost > tiles[(size_t)v].cost; });
  }
  // tiles to ranks: heaviest first, each to the least loaded rank
  std::vector<double> load((size_t)world, 0.0);
  G.per_rank.assign((size_t)world, 0);
  G.mine.assign((size_t)world, std::vector<int>());
  for (const int t : by_cost) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    load[(size_t)r] += tiles[(size_t)t].cost, G.per_rank[(size_t)r] += tiles[(size_t)t].count;
    G.mine[(size_t)r].push_back(t);
  }
}

// One rank's plan from the shared part: pass 2 over its pairs, their classes, the 8 queues per class
static void tiled_rank(qk_plan* p, const TiledGlobal& G, const int rank, const double plan_fit, const int threads) {
  typedef TiledItem Item;
  typedef TiledTile Tile;
  const int n_sites = G.n_sites, nx = G.nx, ny = G.ny;
  const bool sym = G.sym;
  const SiteTab &xt = G.xt, &yt = G.yt();
  const std::vector<Item>& items = G.items;
  const std::vector<Tile>& tiles = G.tiles;
  const std::vector<int>& parent = G.parent;
  const std::vector<int>& mine = G.mine[(size_t)rank];
  const std::vector<int64_t>& per_rank = G.per_rank;
  const int64_t n_items = G.n_items;
  // ---- pass 2, this rank's pairs: algorithmic flops, bytes and the LDS-fit shares of the padded work
  struct Mine {
    double f, by, ft, fn;
  };
  std::vector<int64_t> mine_start((size_t)mine.size() + 1, 0);  // this rank's items, tile by tile
  for (size_t m = 0; m < mine.size(); ++m) mine_start[m + 1] = mine_start[m] + tiles[(size_t)mine[m]].count;
  std::vector<Mine> mw((size_t)mine_start.back());
  par_ranges(threads, (int64_t)mine.size(), [&](const int64_t m_lo, const int64_t m_hi) {
    for (int64_t m = m_lo; m < m_hi; ++m) {
      const Tile& tl = tiles[(size_t)mine[(size_t)m]];
      for (int64_t e = 0; e < tl.count; ++e) {
        const Item& it = items[(size_t)(tl.start + e)];
        const size_t ex = (size_t)it.i * n_sites, ey = (size_t)it.j * n_sites;
        const Pass2 r = pair_pass2(n_sites, &xt.t0[ex], &xt.t1[ex], &xt.t01[ex], &xt.P0[ex], &xt.P1[ex], &xt.P01[ex], &yt.t0[ey], &yt.t1[ey], &yt.t01[ey], &yt.P0[ey], &yt.P1[ey], &yt.P01[ey],
                                   (double)QKF_XCAP_TWO, plan_fit);
        mw[(size_t)(mine_start[(size_t)m] + e)] = Mine{r.f, xt.bytes[(size_t)it.i] + yt.bytes[(size_t)it.j] + 8, r.ft, r.fn};
      }
    }
  });
  // classes of this rank's pairs (see qk_plan_create): class 1 = nearly all of the work fits the fused sweep's smaller LDS buffer
  double split = 0.75;
  if (const char* e = std::getenv("QK_PLAN_SPLIT")) split = std::atof(e);
  double flops = 0, padded = 0, bytes = 0, fit_two = 0, fit_narrow = 0, small_work = 0;
  for (size_t m = 0; m < mine.size(); ++m) {
    const Tile& tl = tiles[(size_t)mine[m]];
    for (int64_t e = 0; e < tl.count; ++e) {
      const Item& it = items[(size_t)(tl.start + e)];
      const Mine& w = mw[(size_t)(mine_start[m] + e)];
      flops += w.f, padded += it.fp, bytes += w.by, fit_two += w.ft, fit_narrow += w.fn;
      if (it.fp > 0 && w.ft >= split * it.fp) small_work += it.fp;
    }
  }
  // the tile-reuse lower bound on this share's bytes: every state read once per Gram tile of the plan it takes part in
  // (a symmetric set is ONE image: a state that plays x in one pair of the tile and y in another is still read once)
  {
    std::vector<std::pair<int, int>> key;  // (parent tile, position in `mine`)
    key.reserve(mine.size());
    for (size_t m = 0; m < mine.size(); ++m) key.push_back({parent[(size_t)mine[m]], (int)m});
    std::sort(key.begin(), key.end());
    std::vector<int> seen_x((size_t)nx, -1), seen_y_own(sym ? 0 : (size_t)ny, -1);
    std::vector<int>& seen_y = sym ? seen_x : seen_y_own;
    double reuse = 0;
    for (const auto& kv : key) {
      const Tile& tl = tiles[(size_t)mine[(size_t)kv.second]];
      for (int64_t e = 0; e < tl.count; ++e) {
        const Item& it = items[(size_t)(tl.start + e)];
        if (seen_x[(size_t)it.i] != kv.first) seen_x[(size_t)it.i] = kv.first, reuse += xt.bytes[(size_t)it.i];
        if (seen_y[(size_t)it.j] != kv.first) seen_y[(size_t)it.j] = kv.first, reuse += yt.bytes[(size_t)it.j];
      }
      reuse += 8.0 * (double)tl.count;
    }
    p->tile_reuse_bytes = reuse;
  }
  // A MIXED set -- some states with every bond <= 32 next to larger ones -- keeps its small-small pairs on the one-wave sweep
  // (2 x 2 register tiles, 2-3 x faster per such pair than the multi-wave kernels): they form the second run instead, swept by
  // qk_sweep_wave2_kernel right behind the fused launch.  (A set whose bonds are all <= 32 runs that kernel anyway.)
  const std::vector<int>&mpx = xt.max_pad, &mpy = yt.max_pad;
  const bool any_large = *std::max_element(mpx.begin(), mpx.end()) > 32 || *std::max_element(mpy.begin(), mpy.end()) > 32;
  auto small_pair = [&](const Item& it) { return mpx[(size_t)it.i] <= 32 && mpy[(size_t)it.j] <= 32; };
  int64_t n_small_pairs = 0, n_mine = 0;
  for (const int t : mine)
    for (int64_t q = tiles[(size_t)t].start; q < tiles[(size_t)t].start + tiles[(size_t)t].count; ++q) n_small_pairs += small_pair(items[(size_t)q]) ? 1 : 0, ++n_mine;
  const bool mixed = any_large && n_small_pairs >= std::max<int64_t>(64, n_mine / 50) && n_small_pairs < n_mine && !std::getenv("QK_PLAN_NO_MIXED");
  p->second_wave2 = mixed;
  const bool two_classes = mixed || !(small_work < 0.05 * padded || small_work > 0.95 * padded);
  // with a quarter or more of the work in large pairs the second class is cut at the narrow site size (QK_PLAN_FIT)
  const bool narrow = !mixed && two_classes && small_work < 0.75 * padded;
  std::vector<uint8_t> cls((size_t)mine_start.back(), 0);
  for (size_t m = 0; m < mine.size(); ++m) {
    const Tile& tl = tiles[(size_t)mine[m]];
    for (int64_t e = 0; e < tl.count; ++e) {
      const Item& it = items[(size_t)(tl.start + e)];
      const Mine& w = mw[(size_t)(mine_start[m] + e)];
      cls[(size_t)(mine_start[m] + e)] = mixed ? (small_pair(it) ? 1 : 0) : ((two_classes && it.fp > 0 && (narrow ? w.fn : w.ft) >= split * it.fp) ? 1 : 0);
    }
  }
  p->pairs.clear(), p->groups.clear();
  p->pairs.reserve(2 * (size_t)mine_start.back()), p->groups.reserve(2 * (size_t)mine_start.back());
  p->second = qk_stats{};
  p->nq = 16;
  for (int c = 0; c < 2; ++c) {
    // this class's share of each tile, tiles to the 8 queues heaviest first / least loaded
    std::vector<std::pair<double, int>> part;  // (class-c cost of the tile, position in `mine`)
    for (size_t m = 0; m < mine.size(); ++m) {
      const Tile& tl = tiles[(size_t)mine[m]];
      double cc = 0;
      for (int64_t e = 0; e < tl.count; ++e)
        if (cls[(size_t)(mine_start[m] + e)] == c) cc += items[(size_t)(tl.start + e)].fp;
      if (cc > 0) part.push_back({cc, (int)m});
    }
    std::stable_sort(part.begin(), part.end(), [](const std::pair<double, int>& u, const std::pair<double, int>& v) { return u.first > v.first; });
    std::vector<double> ql(8, 0.0);
    std::vector<std::vector<int>> queue(8);
    for (const auto& pt : part) {
      const int qd = (int)(std::min_element(ql.begin(), ql.end()) - ql.begin());
      ql[(size_t)qd] += pt.first;
      queue[(size_t)qd].push_back(pt.second);
    }
    for (int qd = 0; qd < 8; ++qd) {
      p->qstart[8 * c + qd] = (int64_t)p->pairs.size() / 2;
      for (const int m : queue[(size_t)qd]) {
        const Tile& tl = tiles[(size_t)mine[(size_t)m]];
        for (int64_t e = 0; e < tl.count; ++e) {
          if (cls[(size_t)(mine_start[(size_t)m] + e)] != c) continue;
          const Item& it = items[(size_t)(tl.start + e)];
          const Mine& w = mw[(size_t)(mine_start[(size_t)m] + e)];
          p->groups.push_back((int32_t)(p->pairs.size() / 2));
          p->groups.push_back(1);
          p->pairs.push_back(it.i);
          p->pairs.push_back(it.j);
          if (c == 1) p->second.pairs += 1, p->second.flops += w.f, p->second.padded_flops += it.fp, p->second.bytes += w.by;
        }
      }
    }
  }
  const int64_t np = (int64_t)p->pairs.size() / 2;
  p->qstart[16] = np;
  p->group = 1;
  p->n_first = p->qstart[8];  // == np when there is one class
  p->total_pairs = n_items;
  p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
  p->stats.pairs = np;
  p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
  p->fit_two = padded > 0 ? fit_two / padded : 1.0;
  p->fit_narrow = padded > 0 ? fit_narrow / padded : 1.0;
  p->edge_k = choose_edge_k(n_sites, G.x_dims, G.y_dims, p->pairs);
}

static int plan_create(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims, uint32_t flags, int32_t world_size, int32_t rank, int32_t block,
                       qk_plan** out) {
  if (!out || !x_dims || n_sites <= 0 || nx <= 0) return qk_fail(QK_EINVAL, "qk_plan_create: bad argument");
  const bool sym = (flags & QK_PLAN_SYMMETRIC) != 0;
  const bool orient = sym && (flags & QK_PLAN_ORIENT) != 0 && !(flags & QK_PLAN_QUADS);
  if (sym) {
    y_dims = x_dims;
    ny = nx;
  } else if (!y_dims || ny <= 0)
    return qk_fail(QK_EINVAL, "qk_plan_create: y_dims required unless symmetric");
  if (world_size <= 0 || rank < 0 || rank >= world_size) return qk_fail(QK_EINVAL, "qk_plan_create: bad rank %d/%d", rank, world_size);
  // The narrow site size of the pair classes (QK_PLAN_FIT, elements of X): when a set holds a substantial share of LARGE pairs, only
  // pairs whose work sits in sites of at most this many elements go to the two-workgroup shape -- the 12-wave dual shape is the
  // better one from about 4 x 4 tiles per site on (uniform chains: bond 48 39.8 against 42.4 ms for the two-workgroup shape, bond 64
  // 99.8 against 86.6 ms).  60 qubits x 6 layers, whole sweep: 4608 (every site that fits the smaller buffer) 377.0 ms, 3584 365.9,
  // 3072 365.6, 2560 and below (one launch of the dual shape) 368.1.  A set without large pairs (40 qubits x 4 layers) stays on the
  // two-workgroup shape as a whole: 12.65 ms against 13.1-13.2 when split at the narrow size.
  double plan_fit = 3072.0;
  if (const char* e = std::getenv("QK_PLAN_FIT")) plan_fit = std::atof(e);
  const int block_arg = block;
  if (block <= 0) block = std::max(nx, ny);  // flat list (QK_PLAN_XCD=0): the whole pair list in cost order
  qk_plan* p = new (std::nothrow) qk_plan;
  if (!p) return qk_fail(QK_ENOMEM, "qk_plan_create: out of memory");
  p->n_sites = n_sites, p->nx = nx, p->ny = ny, p->symmetric = sym, p->world = world_size, p->rank = rank;

  struct Item {
    int32_t i, j;
    float cost;
    int32_t tile = 0;
    int32_t cls = 0;  // 1: nearly all of the pair's work sits in sites that fit the fused sweep's smaller LDS buffer
  };
  if (flags & QK_PLAN_QUADS) {
    // 2x2 blocks of pairs {i1, i2} x {j1, j2} (duos of consecutive states; the last duo of an odd set names its state
    // twice).  A symmetric Gram takes the duo pairs u <= v; its diagonal blocks then hold one mirrored pair (i > j) that
    // is computed redundantly.  Blocks are ordered by decreasing cost and dealt in serpentine order like pairs.
    p->quad = true;
    const int stride = n_sites + 1;
    const int nxd = (nx + 1) / 2, nyd = (ny + 1) / 2;
    struct Quad {
      int32_t i1, i2, j1, j2;
      double cost;
    };
    std::vector<Quad> quads;
    auto work = [&](int i, int j, double* f, double* fp, double* by) { pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, plan_fit, f, fp, by); };
    for (int v = 0; v < nyd; ++v)
      for (int u = 0; u < nxd; ++u) {
        if (sym && u > v) continue;
        Quad q{2 * u, std::min(2 * u + 1, nx - 1), 2 * v, std::min(2 * v + 1, ny - 1), 0.0};
        const int32_t is[2] = {q.i1, q.i2}, js[2] = {q.j1, q.j2};
        for (int b = 0; b < 2; ++b)
          for (int a = 0; a < 2; ++a) {
            double f, fp, by;
            work(is[a], js[b], &f, &fp, &by);
            q.cost += fp;
          }
        quads.push_back(q);
      }
    std::stable_sort(quads.begin(), quads.end(), [](const Quad& a, const Quad& b) { return a.cost > b.cost; });
    std::vector<int64_t> per_rank(world_size, 0);
    double flops = 0, padded = 0, bytes = 0;
    int64_t t = 0;
    for (const Quad& q : quads) {
      const int64_t u = t % (2 * (int64_t)world_size);
      const int r = (int)(u < world_size ? u : 2 * (int64_t)world_size - 1 - u);
      per_rank[r] += 4;
      if (r == rank) {
        const int32_t is[2] = {q.i1, q.i2}, js[2] = {q.j1, q.j2};
        for (int b = 0; b < 2; ++b)
          for (int a = 0; a < 2; ++a) {
            p->pairs.push_back(is[a]);
            p->pairs.push_back(js[b]);
            const bool redundant = (a == 1 && q.i2 == q.i1) || (b == 1 && q.j2 == q.j1) || (sym && is[a] > js[b]);
            if (!redundant) {
              double f, fp, by;
              work(is[a], js[b], &f, &fp, &by);
              flops += f, padded += fp, bytes += by;
            }
          }
      }
      ++t;
    }
    p->groups = {0, 0};
    p->n_first = (int64_t)p->pairs.size() / 2;
    p->total_pairs = 4 * t;
    p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
    p->stats.pairs = (int64_t)p->pairs.size() / 2;
    p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
    p->tile_reuse_bytes = bytes;
    *out = p;
    return QK_OK;
  }
  {
    const char* e = std::getenv("QK_PLAN_XCD");
    if (block_arg <= 0 && !(e && std::atoi(e) == 0)) {
      int T = 8;
      if (const char* te = std::getenv("QK_PLAN_TILE")) T = std::max(1, std::min(64, std::atoi(te)));
      p->plan_threads = plan_threads();
      TiledGlobal G;
      tiled_global(G, n_sites, nx, x_dims, ny, y_dims, sym, orient, world_size, T, p->plan_threads);
      tiled_rank(p, G, rank, plan_fit, p->plan_threads);
      *out = p;
      return QK_OK;
    }
  }
  std::vector<Item> tile;
  const int stride = n_sites + 1;
  int64_t t = 0;  // running index in the global order
  std::vector<int64_t> per_rank(world_size, 0);
  std::vector<int32_t> tile_of;  // locality tile of each pair of this rank
  double flops = 0, padded = 0, bytes = 0, fit_two = 0, fit_narrow = 0;
  const int nbx = (nx + block - 1) / block, nby = (ny + block - 1) / block;
  for (int bj = 0; bj < nby; ++bj)
    for (int bi = 0; bi < nbx; ++bi) {
      if (sym && bi > bj) continue;
      tile.clear();
      for (int j = bj * block; j < std::min(ny, (bj + 1) * block); ++j)
        for (int i = bi * block; i < std::min(nx, (bi + 1) * block); ++i) {
          if (sym && i > j) continue;
          double f, fp, by;
          int xi = i, yj = j;  // the cheaper order of contraction: which state plays Y (QK_PLAN_ORIENT)
          if (orient && i != j && fused_cost(n_sites, x_dims + (int64_t)j * stride, y_dims + (int64_t)i * stride) < fused_cost(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride))
            xi = j, yj = i;
          pair_work(n_sites, x_dims + (int64_t)xi * stride, y_dims + (int64_t)yj * stride, plan_fit, &f, &fp, &by);
          tile.push_back({xi, yj, (float)fp});
        }
      std::stable_sort(tile.begin(), tile.end(), [](const Item& u, const Item& v) { return u.cost > v.cost; });
      for (const Item& it : tile) {
        // serpentine deal (0..W-1, W-1..0, ...): in a cost-sorted run plain round-robin would hand rank 0 the
        // heaviest pair of every W (13 % more flops than rank W-1 on cfg4 at W = 8)
        const int64_t u = t % (2 * (int64_t)world_size);
        const int r = (int)(u < world_size ? u : 2 * (int64_t)world_size - 1 - u);
        ++per_rank[r];
        if (r == rank) {
          p->pairs.push_back(it.i);
          p->pairs.push_back(it.j);
          tile_of.push_back(bj * nbx + bi);
          double f, fp, by, ft, fn;
          pair_work(n_sites, x_dims + (int64_t)it.i * stride, y_dims + (int64_t)it.j * stride, plan_fit, &f, &fp, &by, &ft, &fn);
          flops += f, padded += fp, bytes += by, fit_two += ft, fit_narrow += fn;
        }
        ++t;
      }
    }
  // Regroup this rank's share: pairs that share the x state are made contiguous and cut into
  // groups of at most QK_GROUP (default 4) pairs -- one workgroup sweeps a group in lockstep so
  // that A_i is read once per group and the per-phase latencies are shared.  Groups are then
  // ordered by decreasing cost (longest first for the device-side queue) -- inside their locality tile when the plan has
  // tiles (`block`): the queue then walks the Gram tile by tile.
  {
    int G = 4;
    if (const char* e = std::getenv("QK_GROUP")) G = std::max(1, std::min(4, std::atoi(e)));
    p->group = G;
    const int64_t np = (int64_t)p->pairs.size() / 2;
    std::vector<Item> mine((size_t)np);
    // Two classes of pairs: a set of states of very different entanglement (the 60-qubit x 6-layer set: largest bond 40 ... 248,
    // median 78) holds pairs that are best swept by one 12-wave workgroup per CU next to pairs whose sites all fit the
    // smaller LDS buffer and are best swept two workgroups per CU.  Class-1 pairs (>= QK_PLAN_SPLIT, default 0.75, of their
    // padded work fits the smaller buffer) are listed behind the others; qk_gram_values may sweep the two runs with the two
    // shapes of the site-fused kernel.
    double split = 0.75;
    if (const char* e = std::getenv("QK_PLAN_SPLIT")) split = std::atof(e);
    double small_work = 0;
    for (int64_t q = 0; q < np; ++q) {
      double f, fp, by, ft;
      const int i = p->pairs[2 * q], j = p->pairs[2 * q + 1];
      pair_work(n_sites, x_dims + (int64_t)i * stride, y_dims + (int64_t)j * stride, plan_fit, &f, &fp, &by, &ft);
      const int cls = (fp > 0 && ft >= split * fp) ? 1 : 0;
      if (cls) small_work += fp, p->second.pairs += 1, p->second.flops += f, p->second.padded_flops += fp, p->second.bytes += by;
      mine[(size_t)q] = {i, j, (float)fp, tile_of[(size_t)q], cls};
    }
    if (small_work < 0.05 * padded || small_work > 0.95 * padded) {  // (nearly) one class: no split
      for (Item& it : mine) it.cls = 0;
      p->second = qk_stats{};
    }
    std::stable_sort(mine.begin(), mine.end(), [](const Item& u, const Item& v) {
      return u.cls != v.cls ? u.cls < v.cls : u.tile != v.tile ? u.tile < v.tile : u.i != v.i ? u.i < v.i : u.cost > v.cost;
    });
    struct Grp {
      int64_t start;
      int count;
      double cost;
      int32_t tile, cls;
    };
    std::vector<Grp> grp;
    for (int64_t q = 0; q < np;) {
      int c = 1;
      double cost = mine[(size_t)q].cost;
      while (c < G && q + c < np && mine[(size_t)(q + c)].i == mine[(size_t)q].i && mine[(size_t)(q + c)].tile == mine[(size_t)q].tile && mine[(size_t)(q + c)].cls == mine[(size_t)q].cls)
        cost += mine[(size_t)(q + c)].cost, ++c;
      grp.push_back({q, c, cost, mine[(size_t)q].tile, mine[(size_t)q].cls});
      q += c;
    }
    std::stable_sort(grp.begin(), grp.end(), [](const Grp& u, const Grp& v) { return u.cls != v.cls ? u.cls < v.cls : u.tile != v.tile ? u.tile < v.tile : u.cost > v.cost; });
    p->pairs.clear();
    p->n_first = np;
    for (const Grp& gq : grp) {
      if (gq.cls == 1 && p->n_first == np) p->n_first = (int64_t)p->pairs.size() / 2;  // where the class-1 run starts
      p->groups.push_back((int32_t)(p->pairs.size() / 2));
      p->groups.push_back(gq.count);
      for (int c = 0; c < gq.count; ++c) {
        p->pairs.push_back(mine[(size_t)(gq.start + c)].i);
        p->pairs.push_back(mine[(size_t)(gq.start + c)].j);
      }
    }
  }
  p->total_pairs = t;
  p->max_per_rank = *std::max_element(per_rank.begin(), per_rank.end());
  p->stats.pairs = (int64_t)p->pairs.size() / 2;
  p->stats.flops = flops, p->stats.padded_flops = padded, p->stats.bytes = bytes;
  p->tile_reuse_bytes = bytes;
  p->fit_two = padded > 0 ? fit_two / padded : 1.0;
  p->fit_narrow = padded > 0 ? fit_narrow / padded : 1.0;
  p->nq = 1;  // the flat list: one queue per launch
  p->edge_k = choose_edge_k(n_sites, x_dims, y_dims, p->pairs);
  *out = p;
  return QK_OK;
}

extern "C" int qk_plan_create(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims, uint32_t flags, int32_t world_size, int32_t rank, int32_t block,
                              qk_plan** out) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = plan_create(n_sites, nx, x_dims, ny, y_dims, flags, world_size, rank, block, out);
  if (rc == QK_OK) (*out)->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

// The plans of ALL ranks of a one-process communicator from ONE cost pass (qk_comm.hip: qk_gram_sharded): pass 1 and the deal of
// the tiles are shared, every rank then prices only its own pairs.  Same plans as world_size calls of qk_plan_create.
extern "C" int qk_plan_create_all(int32_t n_sites, int32_t nx, const int32_t* x_dims, int32_t ny, const int32_t* y_dims, uint32_t flags, int32_t world_size, qk_plan** out) {
  if (!out || world_size <= 0) return qk_fail(QK_EINVAL, "qk_plan_create_all: bad argument");
  for (int r = 0; r < world_size; ++r) out[r] = nullptr;
  const char* e = std::getenv("QK_PLAN_XCD");
  const bool tiled = !(flags & QK_PLAN_QUADS) && !(e && std::atoi(e) == 0) && x_dims && n_sites > 0 && nx > 0 && ((flags & QK_PLAN_SYMMETRIC) || (y_dims && ny > 0));
  int rc = QK_OK;
  if (!tiled) {  // the flat list and the quad plans are made rank by rank
    for (int r = 0; r < world_size && rc == QK_OK; ++r) rc = qk_plan_create(n_sites, nx, x_dims, ny, y_dims, flags, world_size, r, 0, &out[r]);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    const bool sym = (flags & QK_PLAN_SYMMETRIC) != 0, orient = sym && (flags & QK_PLAN_ORIENT) != 0;
    if (sym) y_dims = x_dims, ny = nx;
    double plan_fit = 3072.0;
    if (const char* f = std::getenv("QK_PLAN_FIT")) plan_fit = std::atof(f);
    int T = 8;
    if (const char* te = std::getenv("QK_PLAN_TILE")) T = std::max(1, std::min(64, std::atoi(te)));
    const int threads = plan_threads();
    TiledGlobal G;
    tiled_global(G, n_sites, nx, x_dims, ny, y_dims, sym, orient, world_size, T, threads);
    const double shared_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int r = 0; r < world_size && rc == QK_OK; ++r) {
      const auto t1 = std::chrono::steady_clock::now();
      qk_plan* p = new (std::nothrow) qk_plan;
      if (!p) {
        rc = qk_fail(QK_ENOMEM, "qk_plan_create_all: out of memory");
        break;
      }
      p->n_sites = n_sites, p->nx = nx, p->ny = ny, p->symmetric = sym, p->world = world_size, p->rank = r, p->plan_threads = threads;
      tiled_rank(p, G, r, plan_fit, threads);
      p->plan_ms = shared_ms / world_size + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      out[r] = p;
    }
  }
  if (rc != QK_OK)
    for (int r = 0; r < world_size; ++r) qk_plan_destroy(out[r]), out[r] = nullptr;
  return rc;
}

extern "C" int64_t qk_plan_num_pairs(const qk_plan* p) { return p ? (int64_t)p->pairs.size() / 2 : 0; }
extern "C" int64_t qk_plan_total_pairs(const qk_plan* p) { return p ? p->total_pairs : 0; }
extern "C" int64_t qk_plan_max_pairs_per_rank(const qk_plan* p) { return p ? p->max_per_rank : 0; }
extern "C" const int32_t* qk_plan_pairs(const qk_plan* p) { return p ? p->pairs.data() : nullptr; }
extern "C" int64_t qk_plan_first_run(const qk_plan* p) { return p ? (p->n_first > 0 ? p->n_first : (int64_t)p->pairs.size() / 2) : 0; }
extern "C" int32_t qk_plan_edge_sites(const qk_plan* p) { return p ? p->edge_k : 0; }
extern "C" int qk_plan_queues(const qk_plan* p, int64_t* qstart) {
  if (!p) return 0;
  if (qstart)
    for (int s = 0; s <= 16; ++s) qstart[s] = p->nq > 1 ? p->qstart[s] : (s == 0 ? 0 : (int64_t)p->pairs.size() / 2);
  return p->nq;
}
extern "C" int qk_plan_stats(const qk_plan* p, qk_stats* out) {
  if (!p || !out) return qk_fail(QK_EINVAL, "qk_plan_stats: null argument");
  *out = p->stats;
  return QK_OK;
}
extern "C" int qk_plan_cost(const qk_plan* p, double* plan_ms, int32_t* threads, double* tile_reuse_bytes) {
  if (!p) return qk_fail(QK_EINVAL, "qk_plan_cost: null plan");
  if (plan_ms) *plan_ms = p->plan_ms;
  if (threads) *threads = p->plan_threads;
  if (tile_reuse_bytes) *tile_reuse_bytes = p->tile_reuse_bytes;
  return QK_OK;
}

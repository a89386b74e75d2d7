// host_graph.cpp -- host-side crystal-graph builder behind include/chgnet_graph.h.
//
// Native replacement for the reference's structure -> CrystalGraph step
// (chgnet/graph/converter.py:102-190): periodic neighbour list, directed/undirected
// bond bookkeeping (create_graph.c:135-203 semantics) and the bond-graph ("line
// graph") enumeration (graph.py:283-327), emitting flat arrays instead of Python
// Node/Edge objects.  Pure C++17, no third-party containers.

#include "chgnet_graph.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <climits>
#include <cstdint>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

namespace {

struct NeighborRows {
  std::vector<int64_t> center, neighbor, image;  // image is [E,3]
  std::vector<double> dist;
};

inline void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// All-pairs periodic neighbour search, exact for any cell shape: along lattice axis k only the
// images with |df_k + image_k| <= r / h_k (h_k = V / |a_i x a_j|, the plane spacing) are visited.
int neighbor_list(int32_t n, const double* frac, const double* L, double r, double tol,
                  NeighborRows& rows) {
  const double *a = L, *b = L + 3, *c = L + 6;
  double bc[3], ca[3], ab[3];
  cross3(b, c, bc);
  cross3(c, a, ca);
  cross3(a, b, ab);
  const double vol = a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2];
  if (!(std::fabs(vol) > 1e-12)) return CHG_GRAPH_EINVAL;
  const double h[3] = {
      std::fabs(vol) / std::sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
      std::fabs(vol) / std::sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
      std::fabs(vol) / std::sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
  const double r2 = r * r;
  const double reach[3] = {r / h[0], r / h[1], r / h[2]};   // in fractional units along each axis

  // cartesian coordinates (row-vector convention: x = frac @ L)
  std::vector<double> cart(3 * static_cast<size_t>(n));
  for (int32_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k)
      cart[3 * i + k] = frac[3 * i] * a[k] + frac[3 * i + 1] * b[k] + frac[3 * i + 2] * c[k];

  for (int32_t i = 0; i < n; ++i) {
    for (int32_t j = 0; j < n; ++j) {
      // |(df + image)_k| * h_k is the distance to the lattice plane family k, hence <= the pair
      // distance: only images with |df_k + image_k| <= r / h_k can be inside the cutoff (exact bound,
      // valid for unwrapped fractional coordinates too)
      int lo[3], hi[3];
      for (int k = 0; k < 3; ++k) {
        const double df = frac[3 * j + k] - frac[3 * i + k];
        lo[k] = static_cast<int>(std::ceil(-df - reach[k] - 1e-9));
        hi[k] = static_cast<int>(std::floor(-df + reach[k] + 1e-9));
      }
      for (int ia = lo[0]; ia <= hi[0]; ++ia)
        for (int ib = lo[1]; ib <= hi[1]; ++ib)
          for (int ic = lo[2]; ic <= hi[2]; ++ic) {
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {
              const double dx = cart[3 * j + k] + ia * a[k] + ib * b[k] + ic * c[k] - cart[3 * i + k];
              d2 += dx * dx;
            }
            if (d2 < r2) {
              const double d = std::sqrt(d2);
              if (d > tol) {
                rows.center.push_back(i);
                rows.neighbor.push_back(j);
                rows.image.push_back(ia);
                rows.image.push_back(ib);
                rows.image.push_back(ic);
                rows.dist.push_back(d);
              }
            }
          }
    }
  }
  return CHG_GRAPH_OK;
}

// Cell-list neighbour search for large cells (1024+ atoms): O(n) instead of O(n^2), and THE SAME rows in THE SAME order as
// neighbor_list above (centre-major, inside a centre sorted by (neighbour, ia, ib, ic)) with bit-identical
// distances: candidates are found by binning the wrapped fractional coordinates into slabs no thinner than r
// along every lattice-plane direction, then every candidate goes through the identical distance expression
// (unwrapped coordinates, image relative to the coordinates as given) and the identical tests, and each
// centre's rows are sorted.  pymatgen's find_points_in_spheres (the reference's neighbour list,
// chgnet/graph/converter.py:132-134) is a cell list as well.
int neighbor_list_cells(int32_t n, const double* frac, const double* L, double r, double tol, NeighborRows& rows) {
  const double *a = L, *b = L + 3, *c = L + 6;
  double bc[3], ca[3], ab[3];
  cross3(b, c, bc);
  cross3(c, a, ca);
  cross3(a, b, ab);
  const double vol = a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2];
  if (!(std::fabs(vol) > 1e-12)) return CHG_GRAPH_EINVAL;
  const double h[3] = {
      std::fabs(vol) / std::sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
      std::fabs(vol) / std::sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
      std::fabs(vol) / std::sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
  const double r2 = r * r;
  int nb[3], reach_bins[3];
  for (int k = 0; k < 3; ++k) {
    nb[k] = std::max(1, std::min(1024, static_cast<int>(std::floor(h[k] / r))));
    // |x_j + I nb - x_i| <= r nb / h in bin units, so the bin offset is at most floor(r nb / h) + 1
    reach_bins[k] = static_cast<int>(std::floor(r * nb[k] / h[k] + 1e-9)) + 1;
  }
  std::vector<double> cart(3 * static_cast<size_t>(n));
  std::vector<int32_t> shift(3 * static_cast<size_t>(n)), bin3(3 * static_cast<size_t>(n)), bin_of(n);
  const size_t n_bins = static_cast<size_t>(nb[0]) * nb[1] * nb[2];
  std::vector<int32_t> bin_start(n_bins + 1, 0), bin_atoms(n);
  for (int32_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      cart[3 * i + k] = frac[3 * i] * a[k] + frac[3 * i + 1] * b[k] + frac[3 * i + 2] * c[k];
      double fl = std::floor(frac[3 * i + k]);
      double w = frac[3 * i + k] - fl;
      if (w >= 1.0) { w -= 1.0; fl += 1.0; }          // -1e-17 - floor(-1e-17) rounds to 1.0
      shift[3 * i + k] = static_cast<int32_t>(fl);
      bin3[3 * i + k] = std::min(nb[k] - 1, static_cast<int>(w * nb[k]));
    }
    bin_of[i] = (bin3[3 * i] * nb[1] + bin3[3 * i + 1]) * nb[2] + bin3[3 * i + 2];
    ++bin_start[bin_of[i] + 1];
  }
  for (size_t q = 0; q < n_bins; ++q) bin_start[q + 1] += bin_start[q];
  {
    std::vector<int32_t> fill(bin_start.begin(), bin_start.end() - 1);
    for (int32_t i = 0; i < n; ++i) bin_atoms[fill[bin_of[i]]++] = i;
  }
  struct Row { int32_t j, ia, ib, ic; double d; };
  std::vector<Row> found;
  for (int32_t i = 0; i < n; ++i) {
    found.clear();
    for (int oa = -reach_bins[0]; oa <= reach_bins[0]; ++oa)
      for (int ob = -reach_bins[1]; ob <= reach_bins[1]; ++ob)
        for (int oc = -reach_bins[2]; oc <= reach_bins[2]; ++oc) {
          const int o[3] = {oa, ob, oc};
          int t[3], wrap[3];        // target bin and the image (relative to WRAPPED coordinates) it is reached through
          for (int k = 0; k < 3; ++k) {
            const int x = bin3[3 * i + k] + o[k];
            wrap[k] = x >= 0 ? x / nb[k] : -((-x + nb[k] - 1) / nb[k]);
            t[k] = x - wrap[k] * nb[k];
          }
          const int32_t q = (t[0] * nb[1] + t[1]) * nb[2] + t[2];
          for (int32_t s = bin_start[q]; s < bin_start[q + 1]; ++s) {
            const int32_t j = bin_atoms[s];
            // image relative to the coordinates as given:  (f_j + I) - f_i = (w_j + wrap) - w_i  with w = f - shift
            const int ia = wrap[0] - shift[3 * j] + shift[3 * i], ib = wrap[1] - shift[3 * j + 1] + shift[3 * i + 1],
                      ic = wrap[2] - shift[3 * j + 2] + shift[3 * i + 2];
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {   // the expression of neighbor_list, term for term
              const double dx = cart[3 * j + k] + ia * a[k] + ib * b[k] + ic * c[k] - cart[3 * i + k];
              d2 += dx * dx;
            }
            if (d2 < r2) {
              const double d = std::sqrt(d2);
              if (d > tol) found.push_back(Row{j, ia, ib, ic, d});
            }
          }
        }
    std::sort(found.begin(), found.end(), [](const Row& x, const Row& y) {
      if (x.j != y.j) return x.j < y.j;
      if (x.ia != y.ia) return x.ia < y.ia;
      if (x.ib != y.ib) return x.ib < y.ib;
      return x.ic < y.ic;
    });
    for (const Row& w : found) {
      rows.center.push_back(i);
      rows.neighbor.push_back(w.j);
      rows.image.push_back(w.ia);
      rows.image.push_back(w.ib);
      rows.image.push_back(w.ic);
      rows.dist.push_back(w.d);
    }
  }
  return CHG_GRAPH_OK;
}

// all-pairs below this size (the window loop is cheaper than binning + sorting for small cells); the choice never
// changes the result.  method: 0 = by size, 1 = all pairs, 2 = cell list (chg_graph_build_with, tests).
int neighbor_search(int32_t n, const double* frac, const double* L, double r, double tol, int method, NeighborRows& rows) {
  const bool cells = method == CHG_GRAPH_SEARCH_CELLS || (method == CHG_GRAPH_SEARCH_AUTO && n >= 1024);
  return cells ? neighbor_list_cells(n, frac, L, r, tol, rows) : neighbor_list(n, frac, L, r, tol, rows);
}

template <class T>
T* dup_array(const std::vector<T>& v) {
  T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

int build_from_rows(int32_t n_atoms, int64_t E, const int64_t* center, const int64_t* neighbor,
                    const int64_t* image, const double* dist, double r_bond, chg_graph** out) {
  if (n_atoms < 0 || E < 0 || !out) return CHG_GRAPH_EINVAL;
  for (int64_t e = 0; e < E; ++e)
    if (center[e] < 0 || center[e] >= n_atoms || neighbor[e] < 0 || neighbor[e] >= n_atoms)
      return CHG_GRAPH_EINVAL;

  std::vector<int32_t> atom_graph(2 * E), img32(3 * E), d2u(E);
  std::vector<int32_t> u2d;          // first directed edge of each undirected bond
  std::vector<int32_t> u_second;     // second directed edge (-1 until paired)
  std::vector<int32_t> u_count;      // number of directed edges attached
  u2d.reserve(E / 2 + 1);
  u_second.reserve(E / 2 + 1);
  u_count.reserve(E / 2 + 1);

  // unordered atom pair -> undirected bonds created under it, in creation order
  // (create_graph.c:152-189; legacy graph.py:166-214)
  std::unordered_map<uint64_t, std::vector<int32_t>> by_pair;
  by_pair.reserve(static_cast<size_t>(E));

  for (int64_t e = 0; e < E; ++e) {
    const int64_t ci = center[e], ni = neighbor[e];
    atom_graph[2 * e] = static_cast<int32_t>(ci);
    atom_graph[2 * e + 1] = static_cast<int32_t>(ni);
    for (int k = 0; k < 3; ++k) img32[3 * e + k] = static_cast<int32_t>(image[3 * e + k]);
    const uint64_t lo = static_cast<uint64_t>(std::min(ci, ni)), hi = static_cast<uint64_t>(std::max(ci, ni));
    auto& group = by_pair[(hi << 32) | lo];
    int32_t joined = -1;
    for (int32_t u : group) {
      const int32_t f = u2d[u];  // its first directed edge must be the exact reverse
      if (center[f] == ni && neighbor[f] == ci && image[3 * f] == -image[3 * e] &&
          image[3 * f + 1] == -image[3 * e + 1] && image[3 * f + 2] == -image[3 * e + 2]) {
        joined = u;
        break;
      }
    }
    if (joined < 0) {
      joined = static_cast<int32_t>(u2d.size());
      u2d.push_back(static_cast<int32_t>(e));
      u_second.push_back(-1);
      u_count.push_back(1);
      group.push_back(joined);
    } else {
      if (u_count[joined] == 1) u_second[joined] = static_cast<int32_t>(e);
      u_count[joined] += 1;
    }
    d2u[e] = joined;
  }
  const int64_t Eu = static_cast<int64_t>(u2d.size());
  if (E != 2 * Eu) return CHG_GRAPH_EUNPAIRED;  // graph.py:273-278

  // per-centre neighbour groups in dict-insertion order (graph.py:23-33):
  // group order = first appearance of the neighbour id, edges inside a group in row order
  std::vector<std::vector<int32_t>> per_center(n_atoms);
  for (int64_t e = 0; e < E; ++e) per_center[center[e]].push_back(static_cast<int32_t>(e));
  std::vector<int32_t> first_seen(n_atoms, -1);
  for (int32_t i = 0; i < n_atoms; ++i) {
    auto& lst = per_center[i];
    int32_t rank = 0;
    std::vector<int32_t> touched;
    for (int32_t e : lst)
      if (first_seen[neighbor[e]] < 0) {
        first_seen[neighbor[e]] = rank++;
        touched.push_back(static_cast<int32_t>(neighbor[e]));
      }
    std::stable_sort(lst.begin(), lst.end(), [&](int32_t x, int32_t y) {
      return first_seen[neighbor[x]] < first_seen[neighbor[y]];
    });
    for (int32_t t : touched) first_seen[t] = -1;
  }

  // line graph (graph.py:283-327)
  std::vector<int32_t> bond_graph;
  for (int32_t u = 0; u < Eu; ++u) {
    const int32_t f = u2d[u];
    if (dist[f] > r_bond) continue;            // note '>' (graph.py:289)
    if (u_count[u] != 2) return CHG_GRAPH_EUNPAIRED;
    const int32_t ends[2] = {atom_graph[2 * f], atom_graph[2 * f + 1]};
    const int32_t des[2] = {f, u_second[u]};
    for (int s = 0; s < 2; ++s) {
      const int32_t ctr = ends[s], de = des[s];
      for (int32_t other : per_center[ctr]) {
        if (other == de) continue;
        if (dist[other] < r_bond) {            // note '<' (graph.py:313)
          bond_graph.push_back(ctr);
          bond_graph.push_back(u);
          bond_graph.push_back(de);
          bond_graph.push_back(d2u[other]);
          bond_graph.push_back(other);
        }
      }
    }
  }

  std::vector<char> is_center(n_atoms, 0);
  for (int64_t e = 0; e < E; ++e) is_center[center[e]] = 1;
  int32_t n_iso = 0;
  for (int32_t i = 0; i < n_atoms; ++i) n_iso += is_center[i] ? 0 : 1;

  chg_graph* g = static_cast<chg_graph*>(std::calloc(1, sizeof(chg_graph)));
  if (!g) return CHG_GRAPH_ENOMEM;
  g->n_atoms = n_atoms;
  g->n_directed = static_cast<int32_t>(E);
  g->n_undirected = static_cast<int32_t>(Eu);
  g->n_angles = static_cast<int32_t>(bond_graph.size() / 5);
  g->n_isolated = n_iso;
  g->atom_graph = dup_array(atom_graph);
  g->image = dup_array(img32);
  g->distance = dup_array(std::vector<double>(dist, dist + E));
  g->directed2undirected = dup_array(d2u);
  g->undirected2directed = dup_array(u2d);
  g->bond_graph = dup_array(bond_graph);
  if (!g->atom_graph || !g->image || !g->distance || !g->directed2undirected ||
      !g->undirected2directed || !g->bond_graph) {
    chg_graph_free(g);
    return CHG_GRAPH_ENOMEM;
  }
  *out = g;
  return CHG_GRAPH_OK;
}

}  // namespace

extern "C" {

int chg_graph_build(int32_t n_atoms, const double* frac, const double* lattice, double r_atom,
                    double r_bond, double numerical_tol, chg_graph** out) {
  return chg_graph_build_with(n_atoms, frac, lattice, r_atom, r_bond, numerical_tol, CHG_GRAPH_SEARCH_AUTO, out);
}

int chg_graph_build_with(int32_t n_atoms, const double* frac, const double* lattice, double r_atom,
                         double r_bond, double numerical_tol, int search, chg_graph** out) {
  if (n_atoms < 0 || !frac || !lattice || !out || !(r_atom > 0) || search < 0 || search > 2) return CHG_GRAPH_EINVAL;
  try {
    NeighborRows rows;
    const int st = neighbor_search(n_atoms, frac, lattice, r_atom, numerical_tol, search, rows);
    if (st != CHG_GRAPH_OK) return st;
    return build_from_rows(n_atoms, static_cast<int64_t>(rows.center.size()), rows.center.data(),
                           rows.neighbor.data(), rows.image.data(), rows.dist.data(), r_bond, out);
  } catch (const std::bad_alloc&) {
    return CHG_GRAPH_ENOMEM;
  }
}

int chg_graph_from_neighbors(int32_t n_atoms, int64_t n_edges, const int64_t* center,
                             const int64_t* neighbor, const int64_t* image, const double* distance,
                             double r_bond, chg_graph** out) {
  if (n_edges > 0 && (!center || !neighbor || !image || !distance)) return CHG_GRAPH_EINVAL;
  try {
    return build_from_rows(n_atoms, n_edges, center, neighbor, image, distance, r_bond, out);
  } catch (const std::bad_alloc&) {
    return CHG_GRAPH_ENOMEM;
  }
}

int chg_pack_batch(int32_t B, const chg_graph_view* v, const chg_packed_out* o, int32_t* n_bnodes, int32_t* bad_graph) {
  if (B < 0 || (B > 0 && !v) || !o || !n_bnodes) return CHG_GRAPH_EINVAL;
  // offsets first (serial, trivial), then every graph writes its own disjoint output ranges: graphs are spread over threads
  int64_t a0 = 0, e0 = 0, u0 = 0, g0 = 0;
  o->atom_off[0] = o->edge_off[0] = o->und_off[0] = o->ang_off[0] = 0;
  for (int32_t b = 0; b < B; ++b) {
    const chg_graph_view& g = v[b];
    if (g.n_atoms < 0 || g.n_directed < 0 || g.n_undirected < 0 || g.n_angles < 0 || g.n_directed != 2 * g.n_undirected) {
      if (bad_graph) *bad_graph = b;
      return CHG_GRAPH_EPAIRING;
    }
    a0 += g.n_atoms; e0 += g.n_directed; u0 += g.n_undirected; g0 += g.n_angles;
    if (a0 >= INT32_MAX || e0 >= INT32_MAX || g0 >= INT32_MAX) {
      if (bad_graph) *bad_graph = b;
      return CHG_GRAPH_EINVAL;
    }
    o->atom_off[b + 1] = static_cast<int32_t>(a0);
    o->edge_off[b + 1] = static_cast<int32_t>(e0);
    o->und_off[b + 1] = static_cast<int32_t>(u0);
    o->ang_off[b + 1] = static_cast<int32_t>(g0);
  }
  auto one_graph = [&](int32_t b) -> int {
    const chg_graph_view& g = v[b];
    const int32_t n = g.n_atoms, ed = g.n_directed, eu = g.n_undirected, na = g.n_angles;
    const int64_t a0 = o->atom_off[b], e0 = o->edge_off[b], u0 = o->und_off[b], g0 = o->ang_off[b];
    for (int32_t i = 0; i < n; ++i) {
      const int32_t z = g.atomic_number[i];
      if (z < 1 || z > 94) return CHG_GRAPH_ERANGE;
      o->z[a0 + i] = z;
      o->atom_owner[a0 + i] = b;
    }
    std::memcpy(o->frac + 3 * a0, g.frac, sizeof(float) * 3 * static_cast<size_t>(n));
    std::memcpy(o->lattice + 9 * static_cast<size_t>(b), g.lattice, sizeof(float) * 9);
    std::memcpy(o->e_image + 3 * e0, g.image, sizeof(float) * 3 * static_cast<size_t>(ed));
    for (int32_t k = 0; k < eu; ++k) {
      const int32_t f = g.undirected2directed[k];
      if (f < 0 || f >= ed) return CHG_GRAPH_ERANGE;
      o->u_u2d[u0 + k] = static_cast<int32_t>(e0 + f);
    }
    // bond-pair order and reverse edges: bond k = (first = u2d[k], second = its other directed edge)
    for (int32_t e = 0; e < ed; ++e) o->p_center[e0 + e] = -1;   // scratch: second edge of bond k, indexed by k
    for (int32_t e = 0; e < ed; ++e) {
      const int32_t c = g.atom_graph[2 * e], nb = g.atom_graph[2 * e + 1], k = g.directed2undirected[e];
      if (c < 0 || c >= n || nb < 0 || nb >= n || k < 0 || k >= eu) return CHG_GRAPH_ERANGE;
      o->e_center[e0 + e] = static_cast<int32_t>(a0 + c);
      o->e_nbr[e0 + e] = static_cast<int32_t>(a0 + nb);
      o->e_d2u[e0 + e] = static_cast<int32_t>(u0 + k);
      o->e_owner[e0 + e] = b;
      if (g.undirected2directed[k] != e) {
        if (o->p_center[e0 + k] != -1) return CHG_GRAPH_EPAIRING;   // a third edge on bond k
        o->p_center[e0 + k] = e;
      }
    }
    for (int32_t k = 0; k < eu; ++k) {
      const int32_t f = g.undirected2directed[k], s2 = o->p_center[e0 + k];
      if (s2 < 0 || g.directed2undirected[f] != k) return CHG_GRAPH_EPAIRING;
      o->e_rev[e0 + f] = static_cast<int32_t>(e0 + s2);
      o->e_rev[e0 + s2] = static_cast<int32_t>(e0 + f);
    }
    for (int32_t k = eu - 1; k >= 0; --k) {   // descending: slots 2k, 2k+1 >= k, the scratch entries still to be read sit below
      const int32_t f = g.undirected2directed[k], s2 = o->p_center[e0 + k];
      o->p_nbr[e0 + 2 * k] = static_cast<int32_t>(a0 + g.atom_graph[2 * f + 1]);
      o->p_nbr[e0 + 2 * k + 1] = static_cast<int32_t>(a0 + g.atom_graph[2 * s2 + 1]);
      const int32_t c1 = static_cast<int32_t>(a0 + g.atom_graph[2 * f]), c2 = static_cast<int32_t>(a0 + g.atom_graph[2 * s2]);
      o->p_center[e0 + 2 * k] = c1;
      o->p_center[e0 + 2 * k + 1] = c2;
    }
    for (int32_t a = 0; a < na; ++a) {
      const int32_t* r = g.bond_graph + 5 * static_cast<size_t>(a);
      if (r[0] < 0 || r[0] >= n || r[1] < 0 || r[1] >= eu || r[2] < 0 || r[2] >= ed || r[3] < 0 || r[3] >= eu || r[4] < 0 || r[4] >= ed)
        return CHG_GRAPH_ERANGE;
      o->a_ctr[g0 + a] = static_cast<int32_t>(a0 + r[0]);
      o->a_b1[g0 + a] = static_cast<int32_t>(u0 + r[1]);
      o->a_d1[g0 + a] = static_cast<int32_t>(e0 + r[2]);
      o->a_b2[g0 + a] = static_cast<int32_t>(u0 + r[3]);
      o->a_d2[g0 + a] = static_cast<int32_t>(e0 + r[4]);
    }
    return CHG_GRAPH_OK;
  };
  const int n_threads = B >= 64 ? static_cast<int>(std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()))) : 1;
  std::atomic<int> first_bad{INT32_MAX}, status{CHG_GRAPH_OK};
  auto worker = [&](int t) {
    for (int32_t b = t; b < B; b += n_threads) {
      const int st = one_graph(b);
      if (st != CHG_GRAPH_OK) {
        int cur = first_bad.load();
        while (b < cur && !first_bad.compare_exchange_weak(cur, b)) {}
        if (first_bad.load() == b) status.store(st);
        return;
      }
    }
  };
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
  }
  if (first_bad.load() != INT32_MAX) {
    if (bad_graph) *bad_graph = first_bad.load();
    return status.load();
  }
  // compact numbering of the bond-graph nodes (monotone in the undirected index)
  for (int64_t k = 0; k < u0; ++k) o->u_bnode[k] = -1;
  for (int64_t a = 0; a < g0; ++a) {
    o->u_bnode[o->a_b1[a]] = 0;
    o->u_bnode[o->a_b2[a]] = 0;
  }
  int32_t nn = 0;
  for (int64_t k = 0; k < u0; ++k)
    if (o->u_bnode[k] == 0) {
      o->u_bnode[k] = nn;
      o->bn_und[nn++] = static_cast<int32_t>(k);
    }
  for (int64_t a = 0; a < g0; ++a) {
    o->a_b1c[a] = o->u_bnode[o->a_b1[a]];
    o->a_b2c[a] = o->u_bnode[o->a_b2[a]];
  }
  *n_bnodes = nn;
  return CHG_GRAPH_OK;
}

void chg_graph_free(chg_graph* g) {
  if (!g) return;
  std::free(g->atom_graph);
  std::free(g->image);
  std::free(g->distance);
  std::free(g->directed2undirected);
  std::free(g->undirected2directed);
  std::free(g->bond_graph);
  std::free(g);
}

const char* chg_graph_strerror(int status) {
  switch (status) {
    case CHG_GRAPH_OK: return "ok";
    case CHG_GRAPH_EINVAL: return "invalid argument (null pointer, negative size, index out of range or singular lattice)";
    case CHG_GRAPH_ENOMEM: return "out of host memory";
    case CHG_GRAPH_EUNPAIRED: return "number of directed edges != 2 * number of undirected edges (directed edges are not complete)";
    case CHG_GRAPH_ERANGE: return "an index (or an atomic number outside 1..94) lies outside its structure";
    case CHG_GRAPH_EPAIRING: return "directed2undirected must map exactly two directed edges onto every undirected edge, one of them undirected2directed[k]";
    default: return "unknown status";
  }
}

}  // extern "C"

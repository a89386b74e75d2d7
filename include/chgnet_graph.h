/* chgnet_graph.h -- C-ABI of the host-side crystal-graph builder (libchgnet_graph.so).
 *
 * Replaces, for the path structure -> CrystalGraph record:
 *   - pymatgen Structure.get_neighbor_list, sole call site
 *     chgnet/graph/converter.py:132-134  (third-party, un-vendored; restated here)
 *   - the reference's only native entry point
 *     ReturnElems2* create_graph(int64_t*, int64_t, int64_t*, int64_t*, double*, int64_t)
 *     chgnet/graph/fast_converter_libraries/create_graph.c:100-107, wrapped by
 *     make_graph(...) chgnet/graph/cygraph.pyx:69-76
 *   - Graph.adjacency_list / Graph.line_graph_adjacency_list
 *     chgnet/graph/graph.py:226-247, 249-328
 * It emits the CrystalGraph tensors (chgnet/graph/crystalgraph.py:18-100) directly as
 * flat int32 / float64 arrays instead of Python Node/Edge objects.
 *
 * Plain C types only; all output arrays are owned by the library and released with
 * chg_graph_free().  Functions return 0 on success, a negative chg_graph_status
 * otherwise; chg_graph_strerror() maps a status to text.  Thread-safe (no globals).
 */
#ifndef CHGNET_GRAPH_H
#define CHGNET_GRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  CHG_GRAPH_OK = 0,
  CHG_GRAPH_EINVAL = -1,        /* bad argument (null pointer, n<0, singular lattice) */
  CHG_GRAPH_ENOMEM = -2,
  CHG_GRAPH_EUNPAIRED = -3      /* Ed != 2*Eu: graph.py:273-278 raises ValueError */
} chg_graph_status;

typedef struct chg_graph {
  int32_t n_atoms;
  int32_t n_directed;      /* Ed */
  int32_t n_undirected;    /* Eu, Ed == 2*Eu */
  int32_t n_angles;        /* A  */
  int32_t n_isolated;      /* atoms that never appear as a centre (converter.py:161) */
  int32_t* atom_graph;     /* [Ed,2] (centre, neighbour)                graph.py:240 */
  int32_t* image;          /* [Ed,3] periodic image of the neighbour    converter.py:180 */
  double*  distance;       /* [Ed]   neighbour-list distance (float64)               */
  int32_t* directed2undirected; /* [Ed]                                 graph.py:241-245 */
  int32_t* undirected2directed; /* [Eu] first directed edge of each bond graph.py:285 */
  int32_t* bond_graph;     /* [A,5] (atom, und_i, dir_i, und_j, dir_j)  graph.py:316-324 */
} chg_graph;

/* Periodic neighbour list + graph in one call.
 * frac [n,3] fractional coordinates, lattice [3,3] rows = a,b,c (Angstrom), both float64.
 * Pairs with numerical_tol < d < r_atom are listed centre-major (neighbour index, then
 * image, ascending within a centre).  Angles use bonds with d <= r_bond as the owning
 * bond and d < r_bond as the second bond, exactly as graph.py:289,313. */
int chg_graph_build(int32_t n_atoms, const double* frac, const double* lattice,
                    double r_atom, double r_bond, double numerical_tol, chg_graph** out);

/* The neighbour search behind chg_graph_build: all pairs with an exact image window (small cells) or a cell list
 * (slabs no thinner than r_atom along every lattice-plane direction; what pymatgen's find_points_in_spheres, the
 * reference's neighbour list, does too).  Both give the same rows in the same order with bit-identical distances;
 * AUTO switches at 1024 atoms (measured crossover on one host core: 256 atoms 5 ms vs 11 ms, 2048 atoms 100 ms vs 74 ms).  chg_graph_build_with exists so that tests can force either. */
enum { CHG_GRAPH_SEARCH_AUTO = 0, CHG_GRAPH_SEARCH_PAIRS = 1, CHG_GRAPH_SEARCH_CELLS = 2 };
int chg_graph_build_with(int32_t n_atoms, const double* frac, const double* lattice, double r_atom, double r_bond,
                         double numerical_tol, int search, chg_graph** out);

/* Same graph construction from a caller-supplied neighbour list (the arrays pymatgen's
 * get_neighbor_list returns); rows are taken in the given order, which fixes the
 * directed / undirected numbering exactly as create_graph.c:135-203 does. */
int chg_graph_from_neighbors(int32_t n_atoms, int64_t n_edges, const int64_t* center,
                             const int64_t* neighbor, const int64_t* image /*[E,3]*/,
                             const double* distance, double r_bond, chg_graph** out);

/* ---- batching: list of CrystalGraphs -> one packed batch in global numbering (what chg_batch_upload takes) ----
 * Replaces the per-graph Python loop of BatchedGraph.from_graphs (chgnet/model/model.py:820-899: index offsetting
 * 856-857, 873-877, atom_owners 879) plus the extra index arrays of chgnet_amd/pack.py (compact bond-graph nodes,
 * bond-pair order, reverse edges).  Every index is range-checked against its own structure first
 * (CHG_GRAPH_ERANGE; the reference raises IndexError from index_select / nn.Embedding for the same inputs).
 * All output arrays are caller-allocated with the sizes given by the counts (sum over the views); bn_und needs
 * room for n_undirected entries, the number actually used is returned in *n_bnodes. */
typedef struct chg_graph_view {
  int32_t n_atoms, n_directed, n_undirected, n_angles;
  const int32_t* atomic_number;        /* [n]     */
  const float* frac;                   /* [n,3]   */
  const float* lattice;                /* [3,3]   */
  const int32_t* atom_graph;           /* [ed,2]  */
  const float* image;                  /* [ed,3]  */
  const int32_t* directed2undirected;  /* [ed]    */
  const int32_t* undirected2directed;  /* [eu]    */
  const int32_t* bond_graph;           /* [a,5]   */
} chg_graph_view;

typedef struct chg_packed_out {
  int32_t *z, *atom_owner, *atom_off, *edge_off, *und_off, *ang_off;       /* [N] [N] [B+1] x4 */
  float *frac, *lattice, *e_image;                                          /* [N,3] [B,9] [Ed,3] */
  int32_t *e_center, *e_nbr, *e_d2u, *e_owner, *e_rev, *p_center, *p_nbr;   /* [Ed] */
  int32_t *u_u2d, *u_bnode, *bn_und;                                        /* [Eu] [Eu] [<=Eu] */
  int32_t *a_ctr, *a_b1, *a_d1, *a_b2, *a_d2, *a_b1c, *a_b2c;               /* [A] */
} chg_packed_out;

#define CHG_GRAPH_ERANGE (-4)   /* an index (or an atomic number outside 1..94) does not belong to its structure */
#define CHG_GRAPH_EPAIRING (-5) /* directed2undirected / undirected2directed do not pair every bond with two directed edges */
int chg_pack_batch(int32_t n_graphs, const chg_graph_view* views, const chg_packed_out* out, int32_t* n_bnodes,
                   int32_t* bad_graph /* index of the offending graph on error, may be null */);

void chg_graph_free(chg_graph* g);
const char* chg_graph_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif /* CHGNET_GRAPH_H */

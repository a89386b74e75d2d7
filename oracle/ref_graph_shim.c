/* ref_graph_shim.c -- TEST INFRASTRUCTURE: flat-array view of the reference's own C graph builder.
 *
 * Compiles the reference's native code where it lies (no copy):
 *     #include "fast_converter_libraries/create_graph.c"     (-I/root/reference/chgnet/graph)
 * and walks the structures create_graph() returns (chgnet/graph/fast_converter_libraries/create_graph.c:
 * 100-219) the way cygraph.pyx:99-175 and Graph.adjacency_list / line_graph_adjacency_list
 * (chgnet/graph/graph.py:226-247, 249-328) do, so that chgnet_amd's builder can be compared with the
 * reference's compiled "fast" algorithm element for element.  Built by oracle/Makefile into oracle/_ref/
 * (git-ignored); only tests/ load it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fast_converter_libraries/create_graph.c"

/* Returns 0 on success, -3 if the directed edges are not complete (graph.py:273-278) or an undirected bond
 * inside the bond cutoff does not have exactly two directed edges (graph.py:296-305).
 * Output buffers are caller-allocated: atom_graph[2E], d2u[E], u2d[E] (first n_und entries valid),
 * bond_graph[5 * cap_angles]; counts[0] = n_undirected, counts[1] = n_angles (may exceed cap_angles: then
 * only the count is valid). */
int ref_graph_flat(int64_t n_atoms, int64_t n_edges, int64_t* center, int64_t* neighbor, int64_t* image, double* distance,
                   double r_bond, int32_t* atom_graph, int32_t* d2u, int32_t* u2d, int32_t* bond_graph, int64_t cap_angles,
                   int64_t* counts) {
    ReturnElems2* g = create_graph(center, n_edges, neighbor, image, distance, n_atoms);
    int status = 0;
    for (int64_t e = 0; e < g->num_directed_edges; e++) {             /* Graph.adjacency_list */
        DirectedEdge* de = g->directed_edges_list[e];
        atom_graph[2 * e] = (int32_t)de->nodes.center;
        atom_graph[2 * e + 1] = (int32_t)de->nodes.neighbor;
        d2u[e] = (int32_t)de->undirected_edge_index;
    }
    counts[0] = g->num_undirected_edges;
    if (g->num_directed_edges != 2 * g->num_undirected_edges) status = -3;
    int64_t n_ang = 0;
    for (int64_t k = 0; k < g->num_undirected_edges && status == 0; k++) {   /* line_graph_adjacency_list */
        UndirectedEdge* u = g->undirected_edges_list[k];
        u2d[k] = (int32_t)u->directed_edge_indices[0];
        if (u->distance > r_bond) continue;
        if (u->num_directed_edges != 2) { status = -3; break; }
        const int64_t ends[2] = {u->nodes.center, u->nodes.neighbor};
        for (int s = 0; s < 2; s++) {
            const int64_t ctr = ends[s], de = u->directed_edge_indices[s];
            LongToDirectedEdgeList *grp, *tmp;
            HASH_ITER(hh, g->nodes[ctr].neighbors, grp, tmp) {       /* insertion order, like a Python dict */
                for (int i = 0; i < grp->num_directed_edges_in_group; i++) {
                    DirectedEdge* other = grp->directed_edges_list[i];
                    if (other->index == de) continue;
                    if (other->distance < r_bond) {
                        if (n_ang < cap_angles) {
                            int32_t* row = bond_graph + 5 * n_ang;
                            row[0] = (int32_t)ctr; row[1] = (int32_t)u->index; row[2] = (int32_t)de;
                            row[3] = (int32_t)other->undirected_edge_index; row[4] = (int32_t)other->index;
                        }
                        n_ang++;
                    }
                }
            }
        }
    }
    counts[1] = n_ang;
    /* release what create_graph allocated (cygraph.pyx:157-173 does the same on the Cython side) */
    free_LongToDirectedEdgeList_in_nodes(g->nodes, g->num_nodes);
    for (int64_t e = 0; e < g->num_directed_edges; e++) free(g->directed_edges_list[e]);
    for (int64_t k = 0; k < g->num_undirected_edges; k++) {
        free(g->undirected_edges_list[k]->directed_edge_indices);
        free(g->undirected_edges_list[k]);
    }
    free(g->directed_edges_list);
    free(g->undirected_edges_list);
    free(g->nodes);
    free(g);
    return status;
}

// liboct_minpath.so -- host fast path of the min-path boundary delineation (stays on the host, north_star).
// Same algorithm, edge weights and tie-breaking as the reference's pure-Python Dijkstra
// (/root/reference/oct_image_segmentation_models/min_path_processing/graph_search.py:5-105, 360-428):
// heap entries ordered by (distance, neighbour priority, insertion count); neighbour order of
// create_graph_structure (:108-225).  Results are pinned by tests/golden/min_path_golden.npz.
#include <cstddef>
#include <cstdint>
#include <queue>
#include <vector>

namespace {

struct Entry {
    double d; int prio; long cnt; int n; int v;
    bool operator>(const Entry& o) const {
        if (d != o.d) return d > o.d;
        if (prio != o.prio) return prio > o.prio;
        return cnt > o.cnt;
    }
};

// ordered neighbour list of node (row i, col j) in a gw x gh grid (graph_search.py:139-223)
inline int neighbours(int i, int j, int gw, int gh, int max_grad, int* out) {
    int k = 0;
    const int right = (j + 1) + i * gw, down = j + (i + 1) * gw;
    auto up = [&]() { for (int g = 1; g <= max_grad; ++g) if (i - g >= 0) out[k++] = (j + 1) + (i - g) * gw; };
    auto dn = [&]() { for (int g = 1; g <= max_grad; ++g) if (i + g <= gh - 1) out[k++] = (j + 1) + (i + g) * gw; };
    if (i == gh - 1) {
        if (j != gw - 1) { out[k++] = right; up(); }
    } else if (i == 0) {
        if (j == gw - 1) out[k++] = down;
        else if (j == 0) { out[k++] = right; out[k++] = down; dn(); }
        else { out[k++] = right; dn(); }
    } else {
        if (j == gw - 1) out[k++] = down;
        else if (j == 0) { out[k++] = right; out[k++] = down; up(); dn(); }
        else { out[k++] = right; up(); dn(); }
    }
    return k;
}

}  // namespace

extern "C" {

// prob: (gw, gh) row-major doubles [col][row], already with the two appended columns of ones.
// delin: gw-2 doubles.  Returns 0, or -1 if the end vertex is unreachable / arguments are bad.
int oct_minpath_delineate(const double* prob, int gw, int gh, int max_grad, double* delin) {
    if (!prob || !delin || gw < 3 || gh < 1 || max_grad < 1 || max_grad > 16) return -1;
    const int nv = gw * gh, max_ind = nv - 1;
    std::vector<int> prev(nv, -1);
    std::vector<char> done(nv, 0);
    std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> q;
    q.push({0.0, 0, 0, 0, 0});
    long add_count = 1;
    int nb[2 + 2 * 16];
    while (!q.empty()) {
        const Entry e = q.top(); q.pop();
        const int v = e.n;
        if (done[v]) continue;
        done[v] = 1; prev[v] = e.v;
        if (v == max_ind) break;
        const int col = v % gw, row = v / gw;
        const double pv = prob[(std::size_t)col * gh + row];
        const int k = neighbours(row, col, gw, gh, max_grad, nb);
        for (int i = 0; i < k; ++i) {
            const int n = nb[i];
            if (done[n]) continue;
            const int ncol = n % gw, nrow = n / gw;
            const double edge = 2.0 - (pv + prob[(std::size_t)ncol * gh + nrow]);
            const int prio = (ncol == col && nrow == row + 1) ? 0 : i + 1;
            q.push({e.d + edge, prio, add_count, n, v});
            ++add_count;
        }
    }
    if (!done[max_ind]) return -1;
    for (int c = 0; c < gw - 2; ++c) delin[c] = 0.0;
    // walk back from the bottom-right corner; earlier (closer to the end) visits are overwritten by later ones,
    // as the reference's reverse-ordered coordinate list does
    std::vector<int> order;
    int node = max_ind;
    while (node != 0) { order.push_back(node); node = prev[node]; }
    for (int node_i : order) {
        const int c = node_i % gw, r = node_i / gw;
        if (c != 0 && c != gw - 1) delin[c - 1] = (double)r;
    }
    return 0;
}

}  // extern "C"

// partition_host.hpp -- greedy partition of the nearest-neighbour graph into consensus-centre neighbourhoods on integer ids
// (isocon_partition_ids).  What the reference does on networkx graphs keyed by 2.5 kb strings with a pure-Python
// DFS (/root/reference/modules/partitions.py:301-413 get_partitions_no_copy, :416-593 partition_strings;
// end_invariant_functions.py:405-533 for the variant without the neighbour tie-break), here over the edge list the NN search returns:
// repeatedly take the node whose reachable set in the TRANSPOSED graph (everybody whose chain of nearest neighbours leads to it) has
// the largest total multiplicity, make the best-supported member its centre, remove the set.
// Deterministic statement of the reference's rule (the reference's own visiting order depends on PYTHONHASHSEED): candidates
// are ranked by (weight of the reachable set, direct in-neighbours of its representative, rank of the representative's sequence), every
// node of the strongly connected top of a set may represent it.  Same sweeps, same keys, same results as isocon_amd/partitions.py's
// partition_ids_py (kept there as the checker of this routine); tests/golden/g7_partitions.json holds outputs of the reference itself.
// Plain C++ (no HIP): tests/emul/partition_host.cpp compiles it with g++ -fsanitize=address,undefined.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/isocon_hip.h"

namespace {

struct PartitionGraph {
    uint32_t n = 0;
    std::vector<uint64_t> t_ptr, g_ptr;      // CSR of the transpose (b -> a: who points at me) and of G* (a -> b)
    std::vector<uint32_t> t_adj, g_adj;
    void build(uint32_t n_, uint64_t m, const uint32_t *ea, const uint32_t *eb)
    {
        n = n_;
        t_ptr.assign((size_t)n + 1, 0);
        g_ptr.assign((size_t)n + 1, 0);
        for (uint64_t e = 0; e < m; ++e) { ++t_ptr[(size_t)eb[e] + 1]; ++g_ptr[(size_t)ea[e] + 1]; }
        for (uint32_t i = 0; i < n; ++i) { t_ptr[i + 1] += t_ptr[i]; g_ptr[i + 1] += g_ptr[i]; }
        t_adj.resize(m);
        g_adj.resize(m);
        std::vector<uint64_t> tf(t_ptr.begin(), t_ptr.end() - 1), gf(g_ptr.begin(), g_ptr.end() - 1);
        for (uint64_t e = 0; e < m; ++e) { t_adj[tf[eb[e]]++] = ea[e]; g_adj[gf[ea[e]]++] = eb[e]; }
    }
};

// nodes reachable from `start` along `ptr / adj` among the alive ones, start included; `seen` holds stamp for them afterwards
inline void partition_reach(uint32_t start, const std::vector<uint64_t> &ptr, const std::vector<uint32_t> &adj, const std::vector<uint8_t> &alive,
                            std::vector<uint32_t> &seen, uint32_t stamp, std::vector<uint32_t> &out)
{
    out.clear();
    out.push_back(start);
    seen[start] = stamp;
    for (size_t head = 0; head < out.size(); ++head) {
        const uint32_t v = out[head];
        for (uint64_t e = ptr[v]; e < ptr[(size_t)v + 1]; ++e) {
            const uint32_t w = adj[e];
            if (alive[w] && seen[w] != stamp) { seen[w] = stamp; out.push_back(w); }
        }
    }
}

inline int partition_ids_impl(uint32_t n, const int32_t *degree, uint64_t n_edges, const uint32_t *edge_a, const uint32_t *edge_b, const uint32_t *rank,
                              int32_t nbr_tiebreak, uint32_t *out_centre, int64_t *out_weight, uint64_t *out_member_ptr, uint32_t *out_members,
                              uint32_t *n_parts)
{
    for (uint64_t e = 0; e < n_edges; ++e)
        if (edge_a[e] >= n || edge_b[e] >= n) return ISOCON_E_ARG;
    PartitionGraph G;
    G.build(n, n_edges, edge_a, edge_b);
    // weakly connected components in first-node order, then by size (stable), partitions.py:306-307
    std::vector<int32_t> comp_of(n, -1);
    std::vector<std::vector<uint32_t>> comps;
    {
        std::vector<uint32_t> stack;
        for (uint32_t s = 0; s < n; ++s) {
            if (comp_of[s] >= 0) continue;
            const int32_t cid = (int32_t)comps.size();
            comps.emplace_back();
            std::vector<uint32_t> &members = comps.back();
            comp_of[s] = cid;
            members.push_back(s);
            stack.assign(1, s);
            while (!stack.empty()) {
                const uint32_t v = stack.back();
                stack.pop_back();
                for (int side = 0; side < 2; ++side) {
                    const std::vector<uint64_t> &ptr = side ? G.g_ptr : G.t_ptr;
                    const std::vector<uint32_t> &adj = side ? G.g_adj : G.t_adj;
                    for (uint64_t e = ptr[v]; e < ptr[(size_t)v + 1]; ++e) {
                        const uint32_t w = adj[e];
                        if (comp_of[w] < 0) { comp_of[w] = cid; members.push_back(w); stack.push_back(w); }
                    }
                }
            }
        }
        std::stable_sort(comps.begin(), comps.end(), [](const std::vector<uint32_t> &x, const std::vector<uint32_t> &y) { return x.size() > y.size(); });
    }
    std::vector<int64_t> live_in(n);          // direct in-neighbours of G* still in the graph
    for (uint32_t v = 0; v < n; ++v) live_in[v] = (int64_t)(G.t_ptr[(size_t)v + 1] - G.t_ptr[v]);
    std::vector<uint8_t> alive(n, 0);
    std::vector<uint32_t> seen(n, 0), seen_g(n, 0), processed(n, 0), removed(n, 0);
    uint32_t stamp = 0, stamp_g = 0, sweep = 0;
    struct Cand { int64_t weight; uint32_t m; std::vector<uint32_t> reach; int64_t k1; uint32_t k2; };
    uint32_t parts = 0;
    uint64_t mem_at = 0;
    out_member_ptr[0] = 0;
    std::vector<uint32_t> order, top;
    for (const std::vector<uint32_t> &members : comps) {
        for (uint32_t v : members) alive[v] = 1;
        size_t n_alive = members.size();
        order = members;
        while (n_alive) {
            // One sweep: reachable set and weight of every start node that is not inside an earlier start's set (hubs first)
            order.erase(std::remove_if(order.begin(), order.end(), [&](uint32_t v) { return !alive[v]; }), order.end());
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return live_in[x] != live_in[y] ? live_in[x] > live_in[y] : rank[x] < rank[y]; });
            ++sweep;
            std::vector<Cand> cands;
            for (uint32_t m : order) {
                if (processed[m] == sweep) continue;
                Cand c;
                c.m = m;
                if (live_in[m] == 0) c.reach.assign(1, m);
                else {
                    partition_reach(m, G.t_ptr, G.t_adj, alive, seen, ++stamp, c.reach);
                    for (uint32_t v : c.reach) processed[v] = sweep;
                }
                c.weight = 0;
                if (c.reach.size() == 1) c.weight = degree[m];
                else for (uint32_t v : c.reach) c.weight += degree[v];
                c.k1 = 0; c.k2 = 0;
                cands.push_back(std::move(c));
            }
            std::stable_sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) { return x.weight > y.weight; });
            // Extract in key order for as long as the next set is untouched by what was removed in this sweep
            bool any_removed = false;
            size_t k = 0;
            bool stop = false;
            while (k < cands.size() && !stop) {
                size_t e = k;
                while (e < cands.size() && cands[e].weight == cands[k].weight) ++e;
                if (e - k > 1) {
                    for (size_t i = k; i < e; ++i) {
                        Cand &c = cands[i];
                        if (c.reach.size() == 1) { c.k1 = 0; c.k2 = rank[c.m]; continue; }
                        // the strongly connected top of the set: members of reach that m reaches along G*
                        ++stamp;
                        for (uint32_t v : c.reach) seen[v] = stamp;
                        partition_reach(c.m, G.g_ptr, G.g_adj, alive, seen_g, ++stamp_g, top);
                        uint32_t rep = 0xffffffffu;
                        for (uint32_t v : top) {
                            if (seen[v] != stamp) continue;
                            if (rep == 0xffffffffu) { rep = v; continue; }
                            if (nbr_tiebreak) { if (live_in[v] > live_in[rep] || (live_in[v] == live_in[rep] && rank[v] < rank[rep])) rep = v; }
                            else if (rank[v] < rank[rep]) rep = v;
                        }
                        c.k1 = nbr_tiebreak ? -live_in[rep] : 0;
                        c.k2 = rank[rep];
                    }
                    std::stable_sort(cands.begin() + (long)k, cands.begin() + (long)e,
                                     [](const Cand &x, const Cand &y) { return x.k1 != y.k1 ? x.k1 < y.k1 : x.k2 < y.k2; });
                }
                for (size_t i = k; i < e; ++i) {
                    Cand &c = cands[i];
                    if (any_removed) {
                        bool touched = false;
                        for (uint32_t v : c.reach) if (removed[v] == sweep) { touched = true; break; }
                        if (touched) { stop = true; break; }
                    }
                    // the centre: largest direct weight (own multiplicity + direct in-neighbours), then smallest sequence
                    uint32_t centre = c.m;
                    if (c.reach.size() > 1) {
                        centre = c.reach[0];
                        for (uint32_t v : c.reach) {
                            const int64_t dv = (int64_t)degree[v] + live_in[v], dc = (int64_t)degree[centre] + live_in[centre];
                            if (dv > dc || (dv == dc && rank[v] < rank[centre])) centre = v;
                        }
                    }
                    out_centre[parts] = centre;
                    out_weight[parts] = c.weight;
                    for (uint32_t v : c.reach) if (v != centre) out_members[mem_at++] = v;
                    out_member_ptr[++parts] = mem_at;
                    for (uint32_t v : c.reach) { removed[v] = sweep; alive[v] = 0; }
                    any_removed = true;
                    n_alive -= c.reach.size();
                    for (uint32_t v : c.reach)                      // their nearest neighbours lose an in-neighbour
                        for (uint64_t g = G.g_ptr[v]; g < G.g_ptr[(size_t)v + 1]; ++g) live_in[G.g_adj[g]] -= 1;
                }
                if (!stop) k = e;
            }
        }
    }
    *n_parts = parts;
    return ISOCON_OK;
}

}  // namespace

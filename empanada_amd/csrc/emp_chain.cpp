// Host-side label propagation over component tables (plain C++17, no GPU code; linked into libemp_hip.so).
//
// The whole-stack path keeps every O(#pixels) step on the GPU and hands the host one table of connected
// components per slice plus the component-to-component overlaps of consecutive slices.  What remains is the
// reference's slice-to-slice matcher: forward pass (RLEMatcher.__call__ per slice, matcher.py:262-323, driven by
// forward_matching, patterns.py:60-99), then the backward pass (backward_matching, patterns.py:101-121).  It is
// serial in z by definition; this file is that loop in native code (the numpy version costs ~0.12 ms per slice,
// which at 8 ranks x 256 slices is as long as a rank's forward pass).
//
// Exactness contract (mirrors inference/patterns.py::_ClassChain, which mirrors the reference):
//   iou  = (double)inter / (double)(area_t + area_m - inter)            matcher.py:193-195 (int64 / int64 -> fp64)
//   ioa  = (float)((double)inter / (double)area_m)                      matcher.py:194-196 (stored as fp32)
//   Hungarian step: scipy.optimize.linear_sum_assignment(iou, maximize=True).  When every row and column of the
//   overlap matrix holds at most one non-zero the optimum is forced (it must contain every positive entry), so no
//   solver runs; otherwise the caller-supplied callback runs scipy itself on the dense matrix, so ties break
//   exactly as in the reference.
//   Unmatched instances merge into the target with the largest IoA when ioa_max >= (float)ioa_thr (first maximum),
//   else get a fresh label (forward) / keep their own (backward); instances that end up with equal labels are merged
//   in order of first appearance.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <unordered_map>
#include <utility>
#include <unordered_set>
#include <vector>

extern "C" {
typedef int64_t (*emp_lsap_fn)(const double *iou, int64_t n_rows, int64_t n_cols, int64_t *rows_out, int64_t *cols_out);
}

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Rectangular linear sum assignment, restated from the published algorithm scipy.optimize.linear_sum_assignment
// implements (scipy 1.15.3, the routine the reference calls at matcher.py:213; D. F. Crouse, "On implementing 2D
// rectangular assignment algorithms", IEEE TAES 52(4), 2016: shortest augmenting paths with dual variables u, v).
// The assignment it returns among equally good ones depends on details that are therefore reproduced exactly:
//   * more rows than columns -> the transposed problem is solved; maximisation negates the costs;
//   * rows are augmented in order 0 .. nr-1; the list of unvisited columns starts as nc-1, nc-2, .., 0 and a visited
//     column is replaced by the LAST element of the list;
//   * among the unvisited columns the one with the lowest reduced path cost is taken next; on equal cost a column that
//     is still unassigned (a new sink) replaces the current candidate, otherwise the first one found stays.
// tests/test_sharded_gloo.py::test_native_lsap_equals_scipy compares it with scipy itself on random dense, sparse,
// tie-heavy and rectangular matrices; chain_from_tables(lsap='scipy') keeps the library call available.
// Returns the number of assigned pairs (min(nr, nc)), rows ascending; -1 if infeasible (cannot happen for finite costs).
int64_t lsap_maximize(const double *cost_in, int64_t nr, int64_t nc, int64_t *rows_out, int64_t *cols_out)
{
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;
    std::vector<double> cost((size_t)nr * nc);
    if (transpose) {
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = -cost_in[i * nc + j];
        std::swap(nr, nc);
    } else {
        for (int64_t q = 0; q < nr * nc; ++q) cost[q] = -cost_in[q];
    }
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int64_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int64_t cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int64_t num_remaining = nc;
        for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), inf);
        int64_t sink = -1, i = cur;
        while (sink == -1) {
            int64_t index = -1;
            double lowest = inf;
            SR[i] = 1;
            for (int64_t it = 0; it < num_remaining; ++it) {
                const int64_t j = remaining[it];
                const double r = min_val + cost[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) {
                    path[j] = i;
                    spc[j] = r;
                }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                    lowest = spc[j];
                    index = it;
                }
            }
            min_val = lowest;
            if (min_val == inf) return -1;
            const int64_t j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int64_t r2 = 0; r2 < nr; ++r2)
            if (SR[r2] && r2 != cur) u[r2] += min_val - spc[col4row[r2]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        while (true) {
            const int64_t r2 = path[j];
            row4col[j] = r2;
            std::swap(col4row[r2], j);
            if (r2 == cur) break;
        }
    }
    if (transpose) {                        // rows of the original problem are the columns here: sort by them
        std::vector<std::pair<int64_t, int64_t>> pairs(nr);
        for (int64_t r2 = 0; r2 < nr; ++r2) pairs[r2] = {col4row[r2], r2};
        std::sort(pairs.begin(), pairs.end());
        for (int64_t q = 0; q < nr; ++q) {
            rows_out[q] = pairs[q].first;
            cols_out[q] = pairs[q].second;
        }
    } else {
        for (int64_t r2 = 0; r2 < nr; ++r2) {
            rows_out[r2] = r2;
            cols_out[r2] = col4row[r2];
        }
    }
    return nr;
}

struct Inst {                       // ordered instances of one slice: labels, flat component positions, areas
    std::vector<int64_t> labels, comps, seg, areas;
    size_t size() const { return labels.size(); }
    int64_t seg_end(size_t i) const { return i + 1 < seg.size() ? seg[i + 1] : (int64_t)comps.size(); }
};

struct Chain {
    double iou_thr;
    float ioa_thr;
    int64_t next_label;
    emp_lsap_fn lsap;
    int error = 0;
    std::vector<int> row_nnz, col_nnz;             // scratch, reused over the slices
    std::vector<double> iou;
    std::vector<int64_t> rows, cols, arg;
    std::vector<float> best;

    // target / match instances + their non-zero intersections as sparse entries (row i of target, column j of match,
    // pixels; every (i, j) at most once, sorted by (i, j)) -> relabelled (possibly merged) match instances.
    // The dense nt x nm matrices of the reference exist only where scipy needs one (the Hungarian step).
    struct Entry { int64_t i, j, v; };
    Inst match(const Inst &target, const Inst &m, const std::vector<Entry> &ent, bool assign_new)
    {
        const int64_t nt = (int64_t)target.size(), nm = (int64_t)m.size();
        if (nm == 0) return m;
        std::vector<int64_t> new_labels(nm, -1);
        std::vector<int64_t> rest;
        if (nt > 0) {
            row_nnz.assign(nt, 0);
            col_nnz.assign(nm, 0);
            bool forced = true;
            for (const Entry &e : ent)
                if (++row_nnz[e.i] > 1 || ++col_nnz[e.j] > 1) forced = false;
            auto iou_of = [&](const Entry &e) {
                return (double)e.v / (double)(target.areas[e.i] + m.areas[e.j] - e.v);
            };
            if (!ent.empty() && !forced) {
                iou.assign((size_t)nt * nm, 0.0);
                for (const Entry &e : ent) iou[e.i * nm + e.j] = iou_of(e);
                rows.resize(std::min(nt, nm));
                cols.resize(std::min(nt, nm));
                const int64_t k = lsap ? lsap(iou.data(), nt, nm, rows.data(), cols.data())
                                       : lsap_maximize(iou.data(), nt, nm, rows.data(), cols.data());
                if (k < 0) {
                    error = 2;
                    return m;
                }
                for (int64_t q = 0; q < k; ++q)
                    if (iou[rows[q] * nm + cols[q]] >= iou_thr) new_labels[cols[q]] = target.labels[rows[q]];
            } else {
                for (const Entry &e : ent)
                    if (iou_of(e) >= iou_thr) new_labels[e.j] = target.labels[e.i];
            }
            // unmatched columns: first maximum of the fp32 IoA column (rows ascending; absent entries are 0)
            best.assign(nm, 0.f);
            arg.assign(nm, 0);
            for (const Entry &e : ent) {                      // entries are sorted by row: strict > keeps the first
                const float a = (float)((double)e.v / (double)m.areas[e.j]);
                if (a > best[e.j]) {
                    best[e.j] = a;
                    arg[e.j] = e.i;
                }
            }
            for (int64_t j = 0; j < nm; ++j) {
                if (new_labels[j] >= 0) continue;
                if (best[j] >= ioa_thr) new_labels[j] = target.labels[arg[j]];
                else rest.push_back(j);
            }
        } else {
            if (0.f >= ioa_thr) {              // the reference fails on the argmax of an empty sequence here
                error = 1;
                return m;
            }
            for (int64_t j = 0; j < nm; ++j) rest.push_back(j);
        }
        for (int64_t j : rest) {
            if (assign_new) new_labels[j] = next_label++;
            else new_labels[j] = m.labels[j];
        }
        // merge instances that received the same label, groups in order of first appearance
        std::unordered_map<int64_t, int64_t> group_of;
        std::vector<int64_t> grp(nm);
        std::vector<int64_t> glabels;
        for (int64_t j = 0; j < nm; ++j) {
            auto it = group_of.find(new_labels[j]);
            if (it == group_of.end()) {
                grp[j] = (int64_t)glabels.size();
                group_of.emplace(new_labels[j], grp[j]);
                glabels.push_back(new_labels[j]);
            } else {
                grp[j] = it->second;
            }
        }
        Inst out;
        if ((int64_t)glabels.size() == nm) {
            out = m;
            out.labels = new_labels;
            return out;
        }
        const int64_t ng = (int64_t)glabels.size();
        out.labels = glabels;
        out.areas.assign(ng, 0);
        out.seg.assign(ng, 0);
        std::vector<std::vector<int64_t>> members(ng);
        for (int64_t j = 0; j < nm; ++j) {
            members[grp[j]].push_back(j);
            out.areas[grp[j]] += m.areas[j];
        }
        out.comps.reserve(m.comps.size());
        for (int64_t gi = 0; gi < ng; ++gi) {
            out.seg[gi] = (int64_t)out.comps.size();
            for (int64_t j : members[gi])
                for (int64_t c = m.seg[j]; c < m.seg_end(j); ++c) out.comps.push_back(m.comps[c]);
        }
        return out;
    }
};

// instance index of every component position of a slice (-1 where the slice has no such component)
void comp_to_inst(const Inst &inst, int64_t n_comp, std::vector<int64_t> &map)
{
    map.assign(n_comp, -1);
    for (size_t i = 0; i < inst.size(); ++i)
        for (int64_t c = inst.seg[i]; c < inst.seg_end(i); ++c) map[inst.comps[c]] = (int64_t)i;
}

}  // namespace

extern "C" {

// scipy.optimize.linear_sum_assignment(cost, maximize=True) on a dense row-major (n_rows, n_cols) fp64 matrix: the
// native restatement used by emp_chain_class when no callback is given.  rows_out / cols_out hold min(n_rows, n_cols)
// entries; returns their number or -1.
int64_t emp_lsap_maximize(const double *cost, int64_t n_rows, int64_t n_cols, int64_t *rows_out, int64_t *cols_out)
{
    return lsap_maximize(cost, n_rows, n_cols, rows_out, cols_out);
}

// One class.  Components are given sorted by (slice, cc label): slice t owns [bounds[t], bounds[t+1]); a component's
// "position" is its index inside its slice.  Overlap triplets of the slice pair (t, t+1) are [tb_bounds[t],
// tb_bounds[t+1]) with pa = position in slice t, pb = position in slice t+1, tv = overlap in pixels.
// Outputs: comp_final[n] (final label of every component, in the sorted order) and the labels in order of first
// update when the slices are visited last to first (tracker dict order), n_seen of them.
// Returns 0, 1 (ioa_thr <= 0 with an empty target: the reference raises ValueError) or 2 (callback failed).
int emp_chain_class(int64_t D, const int64_t *bounds, const int64_t *comp_label, const int64_t *comp_area,
                    int is_thing, const int64_t *tb_bounds, const int64_t *pa, const int64_t *pb, const int64_t *tv,
                    int64_t class_id, int64_t label_divisor, double iou_thr, double ioa_thr, emp_lsap_fn lsap,
                    int64_t *comp_final, int64_t *seen_labels, int64_t *n_seen)
{
    std::vector<Inst> slices(D);
    for (int64_t t = 0; t < D; ++t) {
        const int64_t n = bounds[t + 1] - bounds[t];
        Inst &s = slices[t];
        s.labels.assign(comp_label + bounds[t], comp_label + bounds[t + 1]);
        s.areas.assign(comp_area + bounds[t], comp_area + bounds[t + 1]);
        s.comps.resize(n);
        s.seg.resize(n);
        for (int64_t i = 0; i < n; ++i) s.comps[i] = s.seg[i] = i;
    }
    std::vector<Inst> result;
    if (is_thing && D > 0) {
        Chain ch;
        ch.iou_thr = iou_thr;
        ch.ioa_thr = (float)ioa_thr;
        ch.next_label = class_id * label_divisor + 1;
        ch.lsap = lsap;
        std::vector<int64_t> rmap, cmap;
        std::vector<Chain::Entry> ent;
        auto inter_of = [&](int64_t t, const Inst &rows, const Inst &cols, bool transposed) {
            // rows live in slice t (+1 if transposed), cols in the other slice of the pair (t, t+1); component
            // overlaps are summed per (row instance, column instance)
            const int64_t n0 = bounds[t + 1] - bounds[t], n1 = bounds[t + 2] - bounds[t + 1];
            comp_to_inst(rows, transposed ? n1 : n0, rmap);
            comp_to_inst(cols, transposed ? n0 : n1, cmap);
            ent.clear();
            for (int64_t q = tb_bounds[t]; q < tb_bounds[t + 1]; ++q) {
                const int64_t a = transposed ? pb[q] : pa[q], b = transposed ? pa[q] : pb[q];
                const int64_t ri = rmap[a], ci = cmap[b];
                if (ri >= 0 && ci >= 0 && tv[q] != 0) ent.push_back({ri, ci, tv[q]});
            }
            std::sort(ent.begin(), ent.end(), [](const Chain::Entry &x, const Chain::Entry &y) {
                return x.i != y.i ? x.i < y.i : x.j < y.j;
            });
            size_t w = 0;
            for (size_t q = 0; q < ent.size(); ++q) {
                if (w && ent[w - 1].i == ent[q].i && ent[w - 1].j == ent[q].j) ent[w - 1].v += ent[q].v;
                else ent[w++] = ent[q];
            }
            ent.resize(w);
        };
        std::vector<Inst> fwd(D);
        fwd[0] = slices[0];
        if (fwd[0].size()) ch.next_label = *std::max_element(fwd[0].labels.begin(), fwd[0].labels.end()) + 1;
        for (int64_t t = 1; t < D; ++t) {
            if (fwd[t - 1].size() && slices[t].size()) inter_of(t - 1, fwd[t - 1], slices[t], false);
            else ent.clear();
            fwd[t] = ch.match(fwd[t - 1], slices[t], ent, true);
            if (ch.error) return ch.error;
        }
        result.resize(D);
        result[D - 1] = fwd[D - 1];
        if (result[D - 1].size())
            ch.next_label = *std::max_element(result[D - 1].labels.begin(), result[D - 1].labels.end()) + 1;
        for (int64_t t = D - 2; t >= 0; --t) {
            if (result[t + 1].size() && fwd[t].size()) inter_of(t, result[t + 1], fwd[t], true);
            else ent.clear();
            result[t] = ch.match(result[t + 1], fwd[t], ent, false);
            if (ch.error) return ch.error;
        }
    } else {
        result.swap(slices);
    }
    std::unordered_set<int64_t> seen;
    int64_t ns = 0;
    for (int64_t t = D - 1; t >= 0; --t) {
        const Inst &inst = result[t];
        for (size_t i = 0; i < inst.size(); ++i) {
            for (int64_t c = inst.seg[i]; c < inst.seg_end(i); ++c) comp_final[bounds[t] + inst.comps[c]] = inst.labels[i];
            if (seen.insert(inst.labels[i]).second) seen_labels[ns++] = inst.labels[i];
        }
    }
    *n_seen = ns;
    return 0;
}

}  // extern "C"

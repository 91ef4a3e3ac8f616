// Host-side mesh preprocessing: vertex graph -> sliced block-ELL pattern, per-cell block
// slots, greedy cell colouring.  Replaces what DOLFIN builds when it creates the
// FunctionSpace / sparsity pattern for `assemble` (fedm/functions.py:192,200).
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "fedm_internal.hpp"

namespace fedm {

// Micro-colouring of one patch's cell list (what `north_star` calls conflict-free writes via
// colouring, at the scale where it pays: the LDS accumulators of one workgroup).  Lane t of the
// workgroup adds cell t's entries with ds_add_f64; for a fixed (a, b, component) the lanes of one
// LDS lane group (`group` consecutive lanes) hit the accumulator column of their cell's a-th
// vertex, whose bank is the owned vertex id modulo `mod`.  Cells are dealt to the lane groups so
// that within a group, for each local index a, those bank classes are distinct wherever
// possible (greedy, first group with the fewest clashes; the order within a group is kept).
static void order_patch_cells(PatchCell *cells, int n, int group, int mod, bool allow_rotation) {
    if (n <= group) return;
    const int n_groups = (n + group - 1) / group;
    std::vector<std::vector<int>> members(n_groups);
    std::vector<std::array<uint64_t, 3>> used(n_groups, std::array<uint64_t, 3>{0, 0, 0});
    // second criterion: the gathers of the staged vertex data (coordinates, unknowns, history) are
    // ds_read_b64 over lane groups of 32 -- two adjacent groups of 16 -- whose bank is the local
    // vertex id (owned or halo) modulo 32 for the interleaved arrays' widest stride
    static const int read_weight = [] {
        const char *e = std::getenv("FEDM_PATCH_ORDER_READS");
        return e ? std::atoi(e) : 1;
    }();
    const int per_half = 32 / group > 0 ? 32 / group : 1;
    std::vector<std::array<uint32_t, 3>> read_used((n_groups + per_half - 1) / per_half, std::array<uint32_t, 3>{0, 0, 0});
    // A cell may also be turned (its three local vertices rotated cyclically: the orientation, and with it
    // every element tensor, is unchanged) -- three times the freedom to find a clash-free place.
    static const bool rotate = [] {
        const char *e = std::getenv("FEDM_PATCH_ROTATE");
        return !(e && e[0] == '0');
    }();
    auto turned = [](const PatchCell &pc, int k) {
        PatchCell r = pc;
        for (int a = 0; a < 3; ++a) {
            const int src = (a + k) % 3;
            r.lv[a] = pc.lv[src];
            r.tag[a] = pc.tag[src];
            for (int b = 0; b < 3; ++b) r.j[a * 3 + b] = pc.j[src * 3 + (b + k) % 3];
        }
        return r;
    };
    // Third criterion (the assembly kernels skip the emission of local row a for a whole wave when none of its
    // cells owns its a-th vertex): the patch's LAST wave -- the one that is not full: 34 of 64 lanes on the
    // tensor-product bench mesh -- takes the cells that own the fewest vertices, turned so that their owned
    // vertices come first: rows 1 and 2 (or row 2) of that wave are skipped, 7 instead of 9 emission blocks a
    // patch.  The other cells keep every freedom of the bank criteria above in the full waves.
    // (measured, tools/kernel_ab.py: 7 instead of 9 emission blocks a patch, but the chosen cells lose the freedom of
    // the bank criteria -- clashing pairs 1.8 -> 8.9 % on the tensor-product mesh, 6.7 -> 9.9 % on the refined one --
    // and the F + J kernel takes the same time within the noise, 74-75 us / 63-65 us: opt-in, FEDM_PATCH_CLASSES=1)
    static const bool by_class = [] {
        const char *e = std::getenv("FEDM_PATCH_CLASSES");
        return e && e[0] == '1';
    }();
    const bool classes = by_class && rotate && allow_rotation && n > SLICE;
    auto n_owned = [](const PatchCell &pc) { return (pc.lv[0] < SLICE) + (pc.lv[1] < SLICE) + (pc.lv[2] < SLICE); };
    std::vector<int> seq(n);
    for (int c = 0; c < n; ++c) seq[c] = c;
    std::vector<char> in_last(n, 0);
    const int last_first_group = ((n - 1) / SLICE) * (SLICE / group);   // first lane group of the last wave
    if (classes && group > 0 && SLICE % group == 0) {
        const int room = n - ((n - 1) / SLICE) * SLICE;               // cells of the last wave
        std::stable_sort(seq.begin(), seq.end(), [&](int x, int y) { return n_owned(cells[x]) < n_owned(cells[y]); });
        for (int k = 0; k < room; ++k) {
            const int c = seq[k];
            const int own = n_owned(cells[c]);
            if (own == 3) break;                                      // (nothing to skip for a cell that owns all three)
            in_last[c] = 1;
            int turn = 0;
            if (own == 2) turn = cells[c].lv[0] >= SLICE ? 1 : cells[c].lv[1] >= SLICE ? 2 : 0;
            else if (own == 1) turn = cells[c].lv[0] < SLICE ? 0 : cells[c].lv[1] < SLICE ? 1 : 2;
            if (turn) cells[c] = turned(cells[c], turn);
        }
        // the full waves first (so that they are full before anything spills over), the chosen cells last
        std::stable_sort(seq.begin(), seq.end(), [&](int x, int y) { return in_last[x] < in_last[y]; });
    }
    int first_group[2] = {0, 0}, end_group[2] = {n_groups, n_groups};
    if (classes) {
        end_group[0] = last_first_group;
        first_group[1] = last_first_group;
    }
    for (int ci = 0; ci < n; ++ci) {
        const int c = seq[ci];
        const int cls = in_last[c] ? 1 : 0;
        int best = -1, best_cost = 1 << 30, best_turn = 0;
        const int n_turns = (rotate && allow_rotation && !in_last[c]) ? 3 : 1;
        for (int pass = 0; pass < 2 && best < 0; ++pass) {   // second pass: any group with room (rounding at class borders)
            const int g_lo = (classes && pass == 0) ? first_group[cls] : 0, g_hi = (classes && pass == 0) ? end_group[cls] : n_groups;
            for (int k = 0; k < n_turns && best_cost > 0; ++k) {
                const PatchCell cand = k ? turned(cells[c], k) : cells[c];
                for (int g = g_lo; g < g_hi; ++g) {
                    if ((int)members[g].size() >= group) continue;
                    int cost = 0;
                    for (int a = 0; a < 3; ++a) {
                        if (cand.lv[a] < SLICE && ((used[g][a] >> (cand.lv[a] % mod)) & 1ULL)) cost += 4;
                        if (read_weight && ((read_used[g / per_half][a] >> (cand.lv[a] & 31)) & 1u)) cost += read_weight;
                    }
                    if (cost < best_cost) {
                        best_cost = cost;
                        best = g;
                        best_turn = k;
                    }
                    if (cost == 0) break;
                }
            }
        }
        if (best_turn) cells[c] = turned(cells[c], best_turn);
        members[best].push_back(c);
        for (int a = 0; a < 3; ++a) {
            if (cells[c].lv[a] < SLICE) used[best][a] |= 1ULL << (cells[c].lv[a] % mod);
            read_used[best / per_half][a] |= 1u << (cells[c].lv[a] & 31);
        }
    }
    std::vector<PatchCell> out;
    out.reserve(n);
    // groups keep their place (pairs of groups form the 32-lane halves); a group that is not full
    // is padded at the end only when it is the last one: move short groups behind the full ones
    std::stable_sort(members.begin(), members.end(),
                     [](const std::vector<int> &x, const std::vector<int> &y) { return x.size() > y.size(); });
    for (const auto &m : members)
        for (int c : m) out.push_back(cells[c]);
    std::copy(out.begin(), out.end(), cells);
}

void build_pattern(const fedm_mesh_desc &mesh, Pattern &pat, bool allow_rotation) {
    const int nv = mesh.n_vertices, nc = mesh.n_cells;
    const int32_t *cells = mesh.cells;
    pat.nv = nv;
    pat.nc = nc;
    pat.n_slices = (nv + SLICE - 1) / SLICE;
    pat.nvp = pat.n_slices * SLICE;

    // --- vertex adjacency (including self), sorted ----------------------------------
    std::vector<int64_t> cnt(nv + 1, 0);
    for (int c = 0; c < nc; ++c)
        for (int a = 0; a < 3; ++a) cnt[cells[3 * c + a] + 1] += 3;
    for (int v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
    std::vector<int> cand(cnt[nv]);
    {
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        for (int c = 0; c < nc; ++c)
            for (int a = 0; a < 3; ++a) {
                int v = cells[3 * c + a];
                for (int b = 0; b < 3; ++b) cand[pos[v]++] = cells[3 * c + b];
            }
    }
    std::vector<int64_t> rowptr(pat.nvp + 1, 0);
    std::vector<int> adj;
    adj.reserve(cnt[nv] / 2);
    pat.row_len.assign(pat.nvp, 1);
    for (int v = 0; v < nv; ++v) {
        auto b = cand.begin() + cnt[v], e = cand.begin() + cnt[v + 1];
        if (b == e) {  // isolated vertex: diagonal only
            adj.push_back(v);
        } else {
            std::sort(b, e);
            e = std::unique(b, e);
            adj.insert(adj.end(), b, e);
        }
        rowptr[v + 1] = (int64_t)adj.size();
        pat.row_len[v] = (int)(rowptr[v + 1] - rowptr[v]);
    }
    for (int v = nv; v < pat.nvp; ++v) {  // padding vertices: diagonal only
        adj.push_back(v);
        rowptr[v + 1] = (int64_t)adj.size();
    }
    pat.nnz_blocks = rowptr[nv];

    // --- slices ---------------------------------------------------------------------
    pat.slice_boff.assign(pat.n_slices + 1, 0);
    for (int s = 0; s < pat.n_slices; ++s) {
        int w = 0;
        for (int l = 0; l < SLICE; ++l) w = std::max(w, pat.row_len[s * SLICE + l]);
        pat.slice_boff[s + 1] = pat.slice_boff[s] + w;
    }
    pat.total_bc = pat.slice_boff[pat.n_slices];
    pat.colidx.resize((size_t)pat.total_bc * SLICE);
    pat.diag_slot.resize(pat.nvp);
    for (int s = 0; s < pat.n_slices; ++s) {
        int w = pat.slice_boff[s + 1] - pat.slice_boff[s];
        for (int l = 0; l < SLICE; ++l) {
            int v = s * SLICE + l;
            int len = pat.row_len[v];
            const int *row = adj.data() + rowptr[v];
            for (int j = 0; j < w; ++j) {
                size_t slot = (size_t)(pat.slice_boff[s] + j) * SLICE + l;
                pat.colidx[slot] = j < len ? row[j] : v;
                if (j < len && row[j] == v) pat.diag_slot[v] = (uint32_t)slot;
            }
        }
    }

    // --- slot of block (a, b) for every cell ------------------------------------------
    pat.cell_slots.resize((size_t)nc * 9);
    for (int c = 0; c < nc; ++c)
        for (int a = 0; a < 3; ++a) {
            int v = cells[3 * c + a];
            int s = v / SLICE, l = v % SLICE;
            const int *row = adj.data() + rowptr[v];
            int len = pat.row_len[v];
            for (int b = 0; b < 3; ++b) {
                int w = cells[3 * c + b];
                int j = (int)(std::lower_bound(row, row + len, w) - row);
                pat.cell_slots[(size_t)c * 9 + a * 3 + b] =
                    (uint32_t)((size_t)(pat.slice_boff[s] + j) * SLICE + l);
            }
        }

    // --- greedy colouring: cells of one colour share no vertex -------------------------
    std::vector<uint64_t> used(nv, 0);
    std::vector<int> colour(nc);
    int n_colours = 0;
    for (int c = 0; c < nc; ++c) {
        uint64_t m = used[cells[3 * c]] | used[cells[3 * c + 1]] | used[cells[3 * c + 2]];
        int k = 0;
        while (k < 63 && ((m >> k) & 1ULL)) ++k;
        colour[c] = k;
        n_colours = std::max(n_colours, k + 1);
        for (int a = 0; a < 3; ++a) used[cells[3 * c + a]] |= (1ULL << k);
    }
    pat.colour_ptr.assign(n_colours + 1, 0);
    for (int c = 0; c < nc; ++c) pat.colour_ptr[colour[c] + 1]++;
    for (int k = 0; k < n_colours; ++k) pat.colour_ptr[k + 1] += pat.colour_ptr[k];
    pat.colour_cells.resize(nc);
    {
        std::vector<int> pos(pat.colour_ptr.begin(), pat.colour_ptr.end() - 1);
        for (int c = 0; c < nc; ++c) pat.colour_cells[pos[colour[c]]++] = c;
    }

    // --- patches: every slice with the cells that touch one of its vertices ---------------
    {
        // thread t of a patch's workgroup evaluates cell t: FEDM_PATCH_ORDER="group,mod" (0 = by
        // cell number) chooses the micro-colouring of order_patch_cells
        int order_group = 16, order_mod = 16;
        if (const char *e = std::getenv("FEDM_PATCH_ORDER")) {
            order_group = std::atoi(e);
            const char *comma = std::strchr(e, ',');
            order_mod = comma ? std::atoi(comma + 1) : order_group;
            if (order_group < 0 || order_group > 64 || order_mod < 1 || order_mod > 64) order_group = 0;
        }
        std::vector<int64_t> vc_ptr(nv + 1, 0);
        for (int c = 0; c < nc; ++c)
            for (int a = 0; a < 3; ++a) vc_ptr[cells[3 * c + a] + 1]++;
        for (int v = 0; v < nv; ++v) vc_ptr[v + 1] += vc_ptr[v];
        std::vector<int> vc(vc_ptr[nv]);
        std::vector<int64_t> pos(vc_ptr.begin(), vc_ptr.end() - 1);
        for (int c = 0; c < nc; ++c)
            for (int a = 0; a < 3; ++a) vc[pos[cells[3 * c + a]]++] = c;
        pat.patch_cell_ptr.assign(pat.n_slices + 1, 0);
        pat.patch_halo_ptr.assign(pat.n_slices + 1, 0);
        pat.patch_cells.clear();
        pat.patch_halo.clear();
        pat.patch_ok = true;
        std::vector<int> pc, halo;
        for (int s = 0; s < pat.n_slices; ++s) {
            const int v0 = s * SLICE, v1 = std::min(nv, v0 + SLICE);
            pc.clear();
            for (int v = v0; v < v1; ++v) pc.insert(pc.end(), vc.begin() + vc_ptr[v], vc.begin() + vc_ptr[v + 1]);
            std::sort(pc.begin(), pc.end());
            pc.erase(std::unique(pc.begin(), pc.end()), pc.end());
            halo.clear();
            for (int c : pc)
                for (int a = 0; a < 3; ++a) {
                    const int v = cells[3 * c + a];
                    if (v < v0 || v >= v1) halo.push_back(v);
                }
            std::sort(halo.begin(), halo.end());
            halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
            const int width = pat.slice_boff[s + 1] - pat.slice_boff[s];
            if (halo.size() + SLICE > 255 || width > 254) pat.patch_ok = false;
            size_t first_cell = pat.patch_cells.size();
            for (int c : pc) {
                PatchCell e{};
                e.cell = c;
                for (int a = 0; a < 3; ++a) {
                    const int v = cells[3 * c + a];
                    int lv;
                    if (v >= v0 && v < v1) lv = v - v0;
                    else lv = SLICE + (int)(std::lower_bound(halo.begin(), halo.end(), v) - halo.begin());
                    e.lv[a] = (uint8_t)std::min(lv, 255);
                    e.tag[a] = mesh.facet_tags ? mesh.facet_tags[3 * c + a] : 0;
                }
                for (int a = 0; a < 3; ++a) {
                    const int v = cells[3 * c + a];
                    for (int b = 0; b < 3; ++b) {
                        if (v < v0 || v >= v1) {
                            e.j[a * 3 + b] = 0xFF;
                            continue;
                        }
                        const int *row = adj.data() + rowptr[v];
                        const int j = (int)(std::lower_bound(row, row + pat.row_len[v], cells[3 * c + b]) - row);
                        e.j[a * 3 + b] = (uint8_t)std::min(j, 254);
                    }
                }
                pat.patch_cells.push_back(e);
            }
            if (order_group > 0)
                order_patch_cells(pat.patch_cells.data() + first_cell, (int)(pat.patch_cells.size() - first_cell),
                                  order_group, order_mod, allow_rotation);
            pat.patch_halo.insert(pat.patch_halo.end(), halo.begin(), halo.end());
            pat.patch_cell_ptr[s + 1] = (int)pat.patch_cells.size();
            pat.patch_halo_ptr[s + 1] = (int)pat.patch_halo.size();
            pat.max_patch_width = std::max(pat.max_patch_width, width);
            pat.max_patch_verts = std::max(pat.max_patch_verts, SLICE + (int)halo.size());
            pat.max_patch_cells = std::max(pat.max_patch_cells, (int)pc.size());
        }
    }
}

}  // namespace fedm

// ---------------------------------------------------------------------------------------------
// Greedy aggregation (Vanek, Mandel, Brezina 1996) on the strength graph of a scalar CSR
// matrix; set-up step of the algebraic multigrid used for the constant Poisson block.
// Host code, cold path (once per mesh).
// ---------------------------------------------------------------------------------------------
extern "C" int fedm_amg_aggregate(int32_t n, const int64_t *indptr, const int32_t *indices,
                                  const uint8_t *strong, int32_t *agg, int32_t *n_agg_out) {
    if (n < 0 || !indptr || !indices || !strong || !agg || !n_agg_out) return -2;
    std::vector<int32_t> a(n, -1);
    int32_t na = 0;
    // pass 1: roots whose strong neighbourhood is still free
    for (int32_t i = 0; i < n; ++i) {
        if (a[i] >= 0) continue;
        bool free_nbhd = true;
        int cnt = 0;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j == i || !strong[k]) continue;
            ++cnt;
            if (a[j] >= 0) {
                free_nbhd = false;
                break;
            }
        }
        if (!free_nbhd || cnt == 0) continue;
        a[i] = na;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)
            if (strong[k] && indices[k] != i) a[indices[k]] = na;
        ++na;
    }
    // pass 2: attach leftovers to a neighbouring pass-1 aggregate
    std::vector<int32_t> b(a);
    for (int32_t i = 0; i < n; ++i) {
        if (a[i] >= 0) continue;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j != i && strong[k] && a[j] >= 0) {
                b[i] = a[j];
                break;
            }
        }
    }
    // pass 3: whatever is left forms aggregates with its free strong neighbours
    for (int32_t i = 0; i < n; ++i) {
        if (b[i] >= 0) continue;
        b[i] = na;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            const int32_t j = indices[k];
            if (j != i && strong[k] && b[j] < 0) b[j] = na;
        }
        ++na;
    }
    for (int32_t i = 0; i < n; ++i) agg[i] = b[i];
    *n_agg_out = na;
    return 0;
}

// Species sweeps of the field split, several per launch (one GPU).
//
// A sweep  z <- zs z + w (g - zs S z)  of fs_species_sweep_kernel (amg.hip) is a 7 us kernel that streams 25 MB:
// launch-to-drain latency, not bandwidth, and a Chebyshev(6) application is five of them.  Here a workgroup owns
// a TILE of consecutive slices (8 x 64 vertices: compact in the locality order of the vertices) and keeps the
// iterate of the tile AND of the n vertex layers around it in LDS.  Sweep k is then evaluated on the tile and the
// layers up to n - k (the rows of the layer vertices are computed redundantly by every tile that needs them, the
// overlapping-Schwarz way the deep halos of the multi-GPU path work), so that the last sweep
// has exact neighbours for the tile itself: the result is the one of n separate sweeps, bit for bit (same
// operands, same order of the sums).  Every row's entries -- half-precision planes of S = Duu^-1 J_uu and the
// tile-local column numbers the host prepared -- are loaded ONCE into registers and reused by all sweeps.
//
// Host side: FsTiles::build walks the block pattern breadth-first from every tile (layers 1..depth) and stores per
// tile the vertex list [tile | layer 1 | ... | layer depth] and, for the rows of the layers < depth, 16-bit local
// column numbers ([entry][row]: coalesced).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "amg.hpp"
#include "fedm_internal.hpp"

namespace fedm {

struct FsTiles {
    int tile_slices = 8, depth = 0, n_tiles = 0, width = 0, max_vertices = 0, max_rows = 0, record = 0;
    int threads = 512;            // per tile: a thread carries max_rows / threads rows in registers
    int *d_tile = nullptr;        // per tile: [vertex offset, column offset, row stride, cnt[0..depth]] (cnt[L]: vertices of layers <= L)
    int *d_vertex = nullptr;      // vertex numbers, tile by tile
    uint32_t *d_rowinfo = nullptr; // beside them: first entry of the vertex's matrix row (block column * 64 + lane) | entries << 26
    uint32_t *d_cols = nullptr;   // [column offset + (entry / 2) * row stride + row]: two 16-bit local columns a word
    long long total_rows = 0, total_vertices = 0;
    size_t bytes = 0;
    size_t lds_limit = 64 * 1024;   // dynamic LDS a workgroup may ask for on this device (fs_tiles_get)
    bool usable = false;
    void release() {
        for (void *p : {(void *)d_tile, (void *)d_vertex, (void *)d_cols, (void *)d_rowinfo})
            if (p) hipFree(p);
        d_tile = d_vertex = nullptr;
        d_rowinfo = nullptr;
        d_cols = nullptr;
        usable = false;
    }
};

static constexpr int FS_TILE_MAX_SWEEPS = 8;
struct FsTileWeights {
    int n;                              // sweeps of this launch
    double zs;                          // weight of the incoming iterate in its first sweep
    double w[FS_TILE_MAX_SWEEPS];
};

__device__ __forceinline__ int tile_of_block(int b, int n, int xcd) {
    if (!xcd) return b;
    const int per = n >> 3, full = per << 3;
    return b < full ? (b & 7) * per + (b >> 3) : b;
}

// NS species, at most W entries per row, SLOTS rows per thread.  Registers decide how many tiles are resident (all
// of them at once is the point: 651 tiles of the bench mesh on 768 workgroup slots): a row's planes stay in
// registers (two halves a word, as they lie in memory), its local column numbers go to LDS, the arithmetic is single
// precision with the half-precision entry as an operand of the fused multiply-add (v_fma_mix_f32: no conversions).
template <int NS, int W, int SLOTS>
__global__ __launch_bounds__(512, (NS > 2 ? 3 : SLOTS <= 3 ? 6 : 4)) void fs_tile_sweeps_kernel(
    const int *__restrict__ tiles, int record, int lds_vertices, int lds_rows, int width, const int *__restrict__ vertex,
    const uint32_t *__restrict__ rowinfo, const uint32_t *__restrict__ cols, const int *__restrict__ boff, const _Float16 *__restrict__ s16, unsigned zmask,
    const float *__restrict__ g32, const float *__restrict__ zin, float *__restrict__ zout32,
    double *__restrict__ zout, FsTileWeights wt, int last, const double *__restrict__ x0,
    const float *__restrict__ cpl32, double *__restrict__ b0, int xcd) {
    constexpr int NEQ = NS + 1, PL = NS * NS, PW = (PL + 1) / 2, W2 = (W + 1) / 2;
    extern __shared__ float fs_tile_lds[];
    float *za = fs_tile_lds, *zb = fs_tile_lds + (size_t)lds_vertices * NS;
    uint32_t *lcol = reinterpret_cast<uint32_t *>(fs_tile_lds + (size_t)2 * lds_vertices * NS);   // [W2][lds_rows]
    const int T = blockDim.x, tid = threadIdx.x;
    // (neighbouring tiles share layer vertices and their rows: a contiguous range of tiles per XCD, so that they meet in
    // one L2 -- workgroups go to the XCDs round robin)
    const int *tl = tiles + (size_t)tile_of_block(blockIdx.x, gridDim.x, xcd) * record;
    const int voff = tl[0], coff = tl[1], rstride = tl[2];
    const int *cnt = tl + 3;
    const int n = wt.n;
    const int n_vertices = cnt[n], n_rows = cnt[n - 1], n_own = cnt[0];

    // Branch-free loads: every load has a valid address -- row 0 / entry 0 stand in where there is nothing to load --
    // and a mask discards what does not belong (behind divergent branches every load of a thread got a basic block
    // and a full wait of its own: 20 us per launch)
    uint32_t a[SLOTS][W][PW];
    float gv[SLOTS][NS];
    int vglob[SLOTS];
    uint32_t pmask[PW];   // structurally zero planes (wave-uniform)
#pragma unroll
    for (int q = 0; q < PW; ++q)
        pmask[q] = (((zmask >> (2 * q)) & 1u) ? 0u : 0xffffu) | ((2 * q + 1 < PL && !((zmask >> (2 * q + 1)) & 1u)) ? 0xffff0000u : 0u);
    uint32_t rinfo[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {   // (the address chains of all rows first, side by side)
        const int r = tid + s * T;
        vglob[s] = vertex[voff + (r < n_rows ? r : 0)];
        rinfo[s] = rowinfo[voff + (r < n_rows ? r : 0)];
    }
    const int width2 = (width + 1) >> 1;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = tid + s * T;
        const int rr = r < n_rows ? r : 0;
        const int v = vglob[s];
        const int wrow = (int)(rinfo[s] >> 26) & (r < n_rows ? -1 : 0);
        const _Float16 *base = s16 + (size_t)(rinfo[s] & 0x3ffffffu) * PL;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const uint32_t have = k < wrow ? 0xffffffffu : 0u;
            const _Float16 *vp = base + (k < wrow ? (size_t)k * SLICE * PL : (size_t)0);
            if constexpr (PL % 2 == 0) {
                uint32_t wd[PW];
                __builtin_memcpy(wd, __builtin_assume_aligned(vp, 4 * PW >= 8 ? 8 : 4), 4 * PW);
#pragma unroll
                for (int q = 0; q < PW; ++q) a[s][k][q] = wd[q] & pmask[q] & have;
            } else {
                const uint16_t *hp = reinterpret_cast<const uint16_t *>(vp);
#pragma unroll
                for (int q = 0; q < PW; ++q)
                    a[s][k][q] = ((uint32_t)hp[2 * q] | (2 * q + 1 < PL ? (uint32_t)hp[2 * q + 1] << 16 : 0u)) & pmask[q] & have;
            }
        }
        // local columns, two to a word: from the tile's table into LDS (the tables hold ceil(width / 2) <= W2 words
        // per row; entries beyond a row's own name the row itself, whose plane entries are zero)
        const uint32_t *cbase = cols + (size_t)coff + rr;
        const uint32_t self2 = (uint32_t)rr | ((uint32_t)rr << 16);
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const uint32_t in_table = j < width2 ? 0xffffffffu : 0u;
            const uint32_t cw = (cbase[(size_t)(j < width2 ? j : 0) * rstride] & in_table) | (self2 & ~in_table);
            if (r < n_rows) lcol[j * lds_rows + r] = cw;
        }
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) gv[s][cidx] = g32[(size_t)v * NS + cidx];
    }
    // the incoming iterate on the tile and its n layers
    for (int i = tid; i < n_vertices; i += T) {
        const int v = vertex[voff + i];
#pragma unroll
        for (int cidx = 0; cidx < NS; ++cidx) za[i * NS + cidx] = zin[(size_t)v * NS + cidx];
    }
    __syncthreads();
    for (int k = 1; k <= n; ++k) {
        const int active = cnt[n - k];
        const float zs = (float)(k == 1 ? wt.zs : 1.0), om = (float)wt.w[k - 1];
        const bool fin = k == n;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int r = tid + s * T;
            if (r >= active) continue;
            float acc[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) acc[q] = 0.f;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                // (the packed words pass through an empty asm in every sweep: otherwise the compiler unpacks all
                // planes of all rows once, outside the sweep loop, and needs 200+ registers for them)
#pragma unroll
                for (int q = 0; q < PW; ++q) asm volatile("" : "+v"(a[s][e][q]));
                const uint32_t cw = lcol[(e >> 1) * lds_rows + r];
                const int col = (cw >> ((e & 1) * 16)) & 0xffffu;
                float zj[NS];
#pragma unroll
                for (int cidx = 0; cidx < NS; ++cidx) zj[cidx] = za[col * NS + cidx];
#pragma unroll
                for (int q = 0; q < NS; ++q)
#pragma unroll
                    for (int cidx = 0; cidx < NS; ++cidx) {
                        const int p = q * NS + cidx;
                        const uint16_t hb = (uint16_t)(a[s][e][p >> 1] >> ((p & 1) * 16));
                        _Float16 hv;
                        __builtin_memcpy(&hv, &hb, 2);
                        acc[q] = __builtin_fmaf((float)hv, zj[cidx], acc[q]);
                    }
            }
            float zn[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q)
                zn[q] = __builtin_fmaf(om, __builtin_fmaf(-zs, acc[q], gv[s][q]), zs * za[r * NS + q]);
            if (!fin) {
#pragma unroll
                for (int q = 0; q < NS; ++q) zb[r * NS + q] = zn[q];
            } else if (r < n_own) {
                const int v = vglob[s];
                if (last) {
#pragma unroll
                    for (int q = 0; q < NS; ++q) zout[(size_t)v * NEQ + q] = (double)zn[q];
                    zout[(size_t)v * NEQ + NS] = x0 ? x0[v] : 0.0;
                } else {
#pragma unroll
                    for (int q = 0; q < NS; ++q) zout32[(size_t)v * NS + q] = zn[q];
                }
            }
        }
        if (!fin) {
            __syncthreads();
            float *t = za;
            za = zb;
            zb = t;
        }
    }
    // the lagged coupling product of the lower-triangular split, b_phi -= J_phi,u (zs z) with the iterate BEFORE the
    // last sweep (still in LDS), as fs_species_sweep_kernel forms it; a loop of its own, not unrolled: inside the
    // sweep it cost 27 registers
    if (last && cpl32 != nullptr) {
        const double zs = n == 1 ? wt.zs : 1.0;
        for (int r = tid; r < n_own; r += T) {
            const int v = vertex[voff + r], lane = v & 63, bb0 = boff[v >> 6], w = boff[(v >> 6) + 1] - bb0;
            double accp = 0.0;
            for (int e = 0; e < w; ++e) {
                const int col = (lcol[(e >> 1) * lds_rows + r] >> ((e & 1) * 16)) & 0xffffu;
                const float *cp = cpl32 + (size_t)(bb0 + e) * NS * SLICE + lane;
#pragma unroll
                for (int cidx = 0; cidx < NS; ++cidx)
                    accp = __builtin_fma((double)cp[(size_t)cidx * SLICE], (double)za[col * NS + cidx], accp);
            }
            b0[v] = __builtin_fma(-zs, accp, b0[v]);
        }
    }
}

// ---- the same tiles for the finest level of the potential block's multigrid cycle ------------------------------------
// Behind the [S | P] product of the polynomial-smoother cycle (the one the hard regime of a streamer run uses) its k
// sweeps  x <- x + w_s Dinv (b - A x)  each stream the finest operator once.  Here they are ONE launch: a workgroup takes
// a tile with its k vertex layers, reads the incoming iterate on all of them and runs the sweeps in LDS, on the tile and
// the layers up to k - s.  A's rows are the potential-potential plane of the Jacobian itself (double precision,
// block-ELL order: the tile tables of the species sweeps apply; the hierarchy's own copy of A is its single-precision
// rounding), Dinv and b the level's vectors.  16 us of two kernels -> 11 us: +2.6 % steps/s in the late window.
// (The V(1,1) cycle's pair -- x = w Dinv b + P x_c, then one sweep -- was built the same way, with P's rows gathered
// for the layer vertices too, and measured 6 us SLOWER than its two kernels: not kept.)
struct MgTileWeights {
    int n;
    double w[4];
};

template <int W, int SLOTS>
__global__ __launch_bounds__(512, (SLOTS <= 3 ? 6 : 4)) void mg_tile_sweeps_kernel(
    const int *__restrict__ tiles, int record, int lds_vertices, int lds_rows, int width, const int *__restrict__ vertex,
    const uint32_t *__restrict__ rowinfo, const uint32_t *__restrict__ cols, const int *__restrict__ boff, const double *__restrict__ val, int neq2, int plane,
    const double *__restrict__ dinv, const double *__restrict__ b, const double *__restrict__ xin, MgTileWeights wt,
    double *__restrict__ out, int ostride, int ooff, int xcd) {
    constexpr int W2 = (W + 1) / 2;
    extern __shared__ float fs_tile_lds[];
    double *xa = reinterpret_cast<double *>(fs_tile_lds), *xb = xa + lds_vertices;
    uint32_t *lcol = reinterpret_cast<uint32_t *>(xb + lds_vertices);   // [W2][lds_rows]
    const int T = blockDim.x, tid = threadIdx.x;
    const int *tl = tiles + (size_t)tile_of_block(blockIdx.x, gridDim.x, xcd) * record;
    const int voff = tl[0], coff = tl[1], rstride = tl[2];
    const int *cnt = tl + 3;
    const int n = wt.n;
    const int n_vertices = cnt[n], n_rows = cnt[n - 1], n_own = cnt[0];
    double a[SLOTS][W], dv[SLOTS], bv[SLOTS];
    int vglob[SLOTS];
    uint32_t rinfo[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = tid + s * T;
        vglob[s] = vertex[voff + (r < n_rows ? r : 0)];
        rinfo[s] = rowinfo[voff + (r < n_rows ? r : 0)];
    }
    const int width2 = (width + 1) >> 1;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int r = tid + s * T;
        const int rr = r < n_rows ? r : 0;
        const int v = vglob[s];
        const uint32_t first = rinfo[s] & 0x3ffffffu;      // (block column * 64 + lane of the row's first entry)
        const int wrow = (int)(rinfo[s] >> 26) & (r < n_rows ? -1 : 0);
        const double *base = val + ((size_t)(first >> 6) * neq2 + plane) * SLICE + (first & 63u);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            // (branch-free, as in fs_tile_sweeps_kernel: a valid address always, a select on the value)
            const double x = base[(k < wrow ? (size_t)k : (size_t)0) * neq2 * SLICE];
            a[s][k] = k < wrow ? x : 0.0;
        }
        const uint32_t *cbase = cols + (size_t)coff + rr;
        const uint32_t self2 = (uint32_t)rr | ((uint32_t)rr << 16);
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const uint32_t in_table = j < width2 ? 0xffffffffu : 0u;
            const uint32_t cw = (cbase[(size_t)(j < width2 ? j : 0) * rstride] & in_table) | (self2 & ~in_table);
            if (r < n_rows) lcol[j * lds_rows + r] = cw;
        }
        dv[s] = dinv[v];
        bv[s] = b[v];
    }
    // the incoming iterate on the tile and its n layers
    for (int i = tid; i < n_vertices; i += T) xa[i] = xin[vertex[voff + i]];
    __syncthreads();
    for (int k = 1; k <= n; ++k) {
        const int active = cnt[n - k];
        const double w = wt.w[k - 1];
        const bool fin = k == n;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int r = tid + s * T;
            if (r >= active) continue;
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(a[s][e]));   // (keeps the row in its 2 W registers: no hoisted products)
                const uint32_t cw = lcol[(e >> 1) * lds_rows + r];
                const int col = (cw >> ((e & 1) * 16)) & 0xffffu;
                acc = __builtin_fma(a[s][e], xa[col], acc);
            }
            const double xn = xa[r] + w * dv[s] * (bv[s] - acc);
            if (!fin) xb[r] = xn;
            else if (r < n_own) out[(size_t)vglob[s] * ostride + ooff] = xn;
        }
        if (!fin) {
            __syncthreads();
            double *t = xa;
            xa = xb;
            xb = t;
        }
    }
}

// Host side of the tiles: vertex lists [tile | layer 1 | ... | layer depth] by a breadth-first walk over the block
// pattern, and the local column numbers (two to a word) of the rows of the layers < depth.  No device call in here:
// fedm_fieldsplit_tiles_stats runs it without a GPU.
struct FsTileTables {
    int tile_slices = 0, depth = 0, n_tiles = 0, width = 0, record = 0, max_vertices = 0, max_rows = 0;
    long long total_rows = 0, total_vertices = 0;
    std::vector<int> tiles, vertices;
    std::vector<uint32_t> cols, rowinfo;
};

static bool fs_tile_tables(const Pattern &pat, int tile_slices, int depth, FsTileTables &tt) {
    tt = FsTileTables();
    tt.tile_slices = tile_slices;
    tt.depth = depth;
    const int T = tile_slices * SLICE;
    tt.n_tiles = (pat.n_slices + tile_slices - 1) / tile_slices;
    tt.record = 3 + depth + 1;
    for (int s = 0; s < pat.n_slices; ++s) tt.width = std::max(tt.width, pat.slice_boff[s + 1] - pat.slice_boff[s]);
    const int width2 = (tt.width + 1) / 2;
    tt.tiles.assign((size_t)tt.n_tiles * tt.record, 0);
    std::vector<int> stamp(pat.nvp, -1), loc(pat.nvp, 0), list, frontier, next;
    std::vector<size_t> coff(tt.n_tiles);
    size_t n_cols = 0;
    // pass 1: vertex lists and counts
    for (int t = 0; t < tt.n_tiles; ++t) {
        int *tl = &tt.tiles[(size_t)t * tt.record];
        const int v0 = t * T, v1 = std::min(v0 + T, pat.nvp);
        list.clear();
        for (int v = v0; v < v1; ++v) {
            stamp[v] = t;
            list.push_back(v);
        }
        frontier = list;
        tl[3] = (int)list.size();
        for (int L = 1; L <= depth; ++L) {
            next.clear();
            for (int v : frontier) {
                const int slice = v >> 6, lane = v & 63;
                for (int bc = pat.slice_boff[slice]; bc < pat.slice_boff[slice + 1]; ++bc) {
                    const int col = pat.colidx[(size_t)bc * SLICE + lane];
                    if (stamp[col] != t) {
                        stamp[col] = t;
                        list.push_back(col);
                        next.push_back(col);
                    }
                }
            }
            tl[3 + L] = (int)list.size();
            frontier.swap(next);
        }
        const int rows = tl[3 + depth - 1];
        if ((long long)tt.vertices.size() + (long long)list.size() > 0x7fffffffLL ||
            n_cols + (size_t)rows * width2 > 0x7fffffffULL)
            return false;
        tl[0] = (int)tt.vertices.size();
        tl[1] = (int)n_cols;
        tl[2] = rows;
        coff[t] = n_cols;
        n_cols += (size_t)rows * width2;
        tt.vertices.insert(tt.vertices.end(), list.begin(), list.end());
        tt.max_vertices = std::max(tt.max_vertices, (int)list.size());
        tt.max_rows = std::max(tt.max_rows, rows);
        tt.total_rows += rows;
        tt.total_vertices += (long long)list.size();
    }
    if (tt.max_vertices > 65535) return false;
    // the matrix row of every listed vertex at a glance: the kernels ask for it together with the vertex number, so
    // the row's planes do not wait for a trip through the slice offsets
    tt.rowinfo.resize(tt.vertices.size());
    for (size_t i = 0; i < tt.vertices.size(); ++i) {
        const int v = tt.vertices[i], slice = v >> 6;
        const long long first = (long long)pat.slice_boff[slice] * SLICE + (v & 63);
        const int w = pat.slice_boff[slice + 1] - pat.slice_boff[slice];
        if (first >= (1LL << 26) || w > 63) return false;
        tt.rowinfo[i] = (uint32_t)first | ((uint32_t)w << 26);
    }
    // pass 2: local column numbers of the rows of the layers < depth
    tt.cols.assign(n_cols, 0u);
    for (int t = 0; t < tt.n_tiles; ++t) {
        const int *tl = &tt.tiles[(size_t)t * tt.record];
        const int nvt = tl[3 + depth], rows = tl[2];
        const int *vl = &tt.vertices[tl[0]];
        for (int i = 0; i < nvt; ++i) loc[vl[i]] = i;   // (every neighbour of a row below is in the list)
        for (int r = 0; r < rows; ++r) {
            const int v = vl[r], slice = v >> 6, lane = v & 63;
            const int b0 = pat.slice_boff[slice], w = pat.slice_boff[slice + 1] - b0;
            for (int k = 0; k < 2 * width2; ++k) {
                const uint32_t lc = (uint32_t)(k < w ? loc[pat.colidx[(size_t)(b0 + k) * SLICE + lane]] : r);
                tt.cols[coff[t] + (size_t)(k >> 1) * rows + r] |= lc << ((k & 1) * 16);
            }
        }
    }
    return true;
}

// the tables of a mesh without a GPU, and a self-check of them: out = {tiles, longest row, most vertices of a tile with
// its layers, most rows, rows of all tiles, vertices of all tiles, bytes, violations found}.  Checked: every vertex is
// the own vertex of exactly one tile; the layer counts grow; every local column of every row is a valid index and names
// the vertex the pattern names; entries beyond a row's own name the row itself.
// Dynamic LDS of one workgroup of the tile kernels: the iterate of the tile and its layers twice (ping, pong)
// and the packed 16-bit local column numbers of its rows, in the kernel instantiation for that row width.
static int fs_tiles_width_class(int width) { return width <= 7 ? 7 : width <= 9 ? 9 : 12; }
size_t fs_tiles_lds_bytes(int width, int max_vertices, int max_rows, int ns) {
    const int W = fs_tiles_width_class(width);
    return sizeof(float) * 2 * (size_t)max_vertices * ns + sizeof(uint32_t) * (size_t)((W + 1) / 2) * max_rows;
}
size_t mg_tiles_lds_bytes(int width, int max_vertices, int max_rows) {
    const int W = fs_tiles_width_class(width);
    return sizeof(double) * 2 * (size_t)max_vertices + sizeof(uint32_t) * (size_t)((W + 1) / 2) * max_rows;
}
// What a workgroup may ask for: the device's limit (160 KiB on gfx950), queried once per device; a mesh within the
// 16-bit index caps of the tables can need more -- scattered ghost layers of a partition make tiles of thousands
// of vertices -- and a launch beyond the limit fails WITHOUT a report inside a captured graph, leaving a stale
// preconditioner output: such tiles are refused when they are built, the sweeps then run one launch each.
static size_t fs_tiles_lds_limit(int device) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || v <= 0) {
        hipGetLastError();
        v = 64 * 1024;
    }
    return (size_t)v;
}

int fs_tiles_host_stats(const fedm_mesh_desc &mesh, int tile_slices, int depth, long long *out) {
    Pattern pat;
    build_pattern(mesh, pat);
    FsTileTables tt;
    if (tile_slices < 1 || tile_slices > 8 || depth < 1 || depth > FS_TILE_MAX_SWEEPS || !fs_tile_tables(pat, tile_slices, depth, tt))
        return -2;
    long long bad = 0;
    std::vector<int> owner(pat.nvp, 0);
    const int width2 = (tt.width + 1) / 2;
    for (int t = 0; t < tt.n_tiles; ++t) {
        const int *tl = &tt.tiles[(size_t)t * tt.record];
        const int *vl = &tt.vertices[tl[0]];
        for (int L = 0; L < depth; ++L) bad += tl[3 + L + 1] < tl[3 + L];
        bad += tl[2] != tl[3 + depth - 1];
        for (int i = 0; i < tl[3]; ++i) ++owner[vl[i]];
        const int nvt = tl[3 + depth], rows = tl[2];
        for (int r = 0; r < rows; ++r) {
            const int v = vl[r], slice = v >> 6, lane = v & 63;
            const int b0 = pat.slice_boff[slice], w = pat.slice_boff[slice + 1] - b0;
            for (int k = 0; k < 2 * width2; ++k) {
                const int lc = (int)((tt.cols[(size_t)tl[1] + (size_t)(k >> 1) * rows + r] >> ((k & 1) * 16)) & 0xffffu);
                if (lc >= nvt) ++bad;
                else if (k < w) bad += vl[lc] != pat.colidx[(size_t)(b0 + k) * SLICE + lane];
                else bad += lc != r;
            }
        }
    }
    for (int v = 0; v < pat.nvp; ++v) bad += owner[v] != 1;
    out[0] = tt.n_tiles;
    out[1] = tt.width;
    out[2] = tt.max_vertices;
    out[3] = tt.max_rows;
    out[4] = tt.total_rows;
    out[5] = tt.total_vertices;
    out[6] = (long long)(sizeof(int) * (tt.tiles.size() + tt.vertices.size()) + sizeof(uint32_t) * tt.cols.size());
    out[7] = bad;
    out[8] = (long long)fs_tiles_lds_bytes(tt.width, tt.max_vertices, tt.max_rows, 2);
    out[9] = (long long)mg_tiles_lds_bytes(tt.width, tt.max_vertices, tt.max_rows);
    return 0;
}

bool FsTiles_build(FsTiles &ft, const Pattern &pat, int tile_slices, int depth) {
    ft.release();
    FsTileTables tt;
    if (!fs_tile_tables(pat, tile_slices, depth, tt)) return false;
    ft.tile_slices = tile_slices;
    ft.depth = depth;
    ft.n_tiles = tt.n_tiles;
    ft.record = tt.record;
    ft.width = tt.width;
    ft.max_vertices = tt.max_vertices;
    ft.max_rows = tt.max_rows;
    ft.total_rows = tt.total_rows;
    ft.total_vertices = tt.total_vertices;
    ft.bytes = sizeof(int) * (tt.tiles.size() + tt.vertices.size()) + sizeof(uint32_t) * (tt.cols.size() + tt.rowinfo.size());
    if (hipMalloc((void **)&ft.d_tile, sizeof(int) * tt.tiles.size()) != hipSuccess ||
        hipMalloc((void **)&ft.d_vertex, sizeof(int) * tt.vertices.size()) != hipSuccess ||
        hipMalloc((void **)&ft.d_cols, sizeof(uint32_t) * std::max<size_t>(tt.cols.size(), 1)) != hipSuccess ||
        hipMalloc((void **)&ft.d_rowinfo, sizeof(uint32_t) * std::max<size_t>(tt.rowinfo.size(), 1)) != hipSuccess) {
        hipGetLastError();
        ft.release();
        return false;
    }
    hipMemcpy(ft.d_tile, tt.tiles.data(), sizeof(int) * tt.tiles.size(), hipMemcpyHostToDevice);
    hipMemcpy(ft.d_vertex, tt.vertices.data(), sizeof(int) * tt.vertices.size(), hipMemcpyHostToDevice);
    hipMemcpy(ft.d_cols, tt.cols.data(), sizeof(uint32_t) * tt.cols.size(), hipMemcpyHostToDevice);
    hipMemcpy(ft.d_rowinfo, tt.rowinfo.data(), sizeof(uint32_t) * tt.rowinfo.size(), hipMemcpyHostToDevice);
    ft.usable = true;
    return true;
}

void fs_tiles_release(Ctx &c) {
    if (!c.fs_tiles) return;
    c.fs_tiles->release();
    delete c.fs_tiles;
    c.fs_tiles = nullptr;
}

// The tiles of this context (built at first use); nullptr where the fused sweeps do not apply: several GPUs with
// an exchange before every sweep (one-layer halos), other than two or four species (the instantiations), rows too long or
// layers too wide for the kernel's instantiations, or FEDM_FS_TILES=0.
static FsTiles *fs_tiles_get(Ctx &c) {
    if (c.fs_tiles_state < 0) return nullptr;
    if (c.fs_tiles) return c.fs_tiles->usable ? c.fs_tiles : nullptr;
    if (c.capturing) return nullptr;   // (allocations and copies do not belong into a stream capture: fs_tiles_prepare)
    if (c.comm && !deep_halo_active(c)) return nullptr;   // (not for good: the global hierarchy may not be installed yet)
    c.fs_tiles_state = -1;
    const char *e = std::getenv("FEDM_FS_TILES");
    // (instantiated for two species; several GPUs: only with deep halos, where nothing is exchanged between the sweeps
    // and the ghost layers are rows of the local matrix like any other)
    if ((e && e[0] == '0') || (c.comm && !deep_halo_active(c)) || (c.ns != 2 && c.ns != 4)) return nullptr;
    // 8 slices (512 vertices) and 3 layers a tile, 512 threads: on the 576 x 576 bench mesh 651 workgroups of up to
    // 973 rows, two rows a thread, all resident at once (three workgroups of eight waves per CU); Chebyshev(6) is two
    // launches, 3 + 2 sweeps: 32 us instead of 44 us for five launches.  Measured alternatives (tools/fs_tiles_probe.py):
    // 5 layers in one launch 40 us (1462 rows, three a thread at 80 registers: spills, and 1.9 x the rows), 2 layers
    // in three launches 39 us, 256 threads a tile 38 us, tiles of 4 slices 52 us
    int tile_slices = 8, depth = 3, threads = 512;
    // four unknowns a vertex (the LMEA species block): a row's 16 half-precision planes are eight registers an entry,
    // so a thread keeps ONE row and a tile is three slices (192 vertices + three layers: 430 rows on the 141 x 141 crossed mesh)
    if (c.ns == 4) tile_slices = 3;
    if (const char *ts = std::getenv("FEDM_FS_TILE_SLICES")) tile_slices = std::atoi(ts);
    if (const char *td = std::getenv("FEDM_FS_TILE_DEPTH")) depth = std::atoi(td);
    if (const char *tt = std::getenv("FEDM_FS_TILE_THREADS")) threads = std::atoi(tt);
    if (c.fs_tiles_want_slices > 0) tile_slices = c.fs_tiles_want_slices;   // (fs_tiles_configure)
    if (c.fs_tiles_want_depth > 0) depth = c.fs_tiles_want_depth;
    if (c.fs_tiles_want_threads > 0) threads = c.fs_tiles_want_threads;
    tile_slices = std::max(1, std::min(8, tile_slices));
    depth = std::max(1, std::min(FS_TILE_MAX_SWEEPS, depth));
    threads = std::max(64, std::min(512, (threads + 63) / 64 * 64));
    FsTiles *ft = new FsTiles();
    ft->threads = threads;
    bool ok = FsTiles_build(*ft, c.pat, tile_slices, depth) && ft->width <= 12 && ft->max_rows <= (c.ns == 4 ? 1 : 8) * threads;
    if (ok) {
        // the species kernel must fit the workgroup's LDS budget (the multigrid kernel is checked at its launch:
        // it is an extra on the same tiles)
        const size_t need = fs_tiles_lds_bytes(ft->width, ft->max_vertices, ft->max_rows, c.ns);
        ft->lds_limit = fs_tiles_lds_limit(c.device);
        if (need > ft->lds_limit) {
            fprintf(stderr, "[fedm_amd] species-sweep tiles need %zu B of LDS per workgroup (%d vertices, %d rows a tile), "
                            "the device grants %zu: sweeps one launch each\n", need, ft->max_vertices, ft->max_rows,
                    ft->lds_limit);
            ok = false;
        }
    }
    if (!ok) {
        ft->release();
        delete ft;
        return nullptr;
    }
    c.fs_tiles = ft;
    c.fs_tiles_state = 1;
    return ft;
}

// mode 0: sweeps one by one from now on; 1: tiles (rebuilt with these parameters; 0 = the defaults / environment)
void fs_tiles_configure(Ctx &c, int mode, int tile_slices, int depth, int threads) {
    fs_tiles_release(c);
    c.fs_tiles_want_slices = tile_slices;
    c.fs_tiles_want_depth = depth;
    c.fs_tiles_want_threads = threads;
    c.fs_tiles_state = mode ? 0 : -1;
}

// builds the tiles where they apply (called with the preconditioner's set-up, outside any stream capture)
void fs_tiles_prepare(Ctx &c) {
    const bool had = c.fs_tiles != nullptr;
    FsTiles *ft = fs_tiles_get(c);
    if (ft && !had) {
        // the cycles' own graphs were recorded before the tiles existed: once more, with mg_tile_sweeps_kernel
        for (Amg *a : {c.amg, c.amg_alt})
            if (a && a->graph_exec) a->capture(c);
    }
}

int fs_tiles_info(Ctx &c, long long *out) {
    FsTiles *ft = fs_tiles_get(c);
    if (!ft) return 0;
    out[0] = ft->n_tiles;
    out[1] = ft->tile_slices;
    out[2] = ft->depth;
    out[3] = ft->width;
    out[4] = ft->max_vertices;
    out[5] = ft->max_rows;
    out[6] = (long long)ft->bytes;
    out[7] = ft->threads;
    out[8] = ft->total_rows;
    out[9] = ft->total_vertices;
    return 1;
}

template <int NS, int W>
static void fs_tiles_launch(Ctx &c, FsTiles &ft, unsigned zmask, const float *g32, const float *in, float *out32,
                            double *z, const FsTileWeights &wt, bool last, const double *x0, const float *cpl32,
                            double *b0) {
    const int T = ft.threads;
    const dim3 g(ft.n_tiles), b(T);
    const size_t lds = sizeof(float) * 2 * (size_t)ft.max_vertices * NS + sizeof(uint32_t) * (size_t)((W + 1) / 2) * ft.max_rows;
    const int slots = (ft.max_rows + T - 1) / T;
    static size_t lds_granted[5] = {0, 0, 0, 0, 0};   // per instantiation <NS, W, SL>: beyond the 64 KiB default it is opt-in
    static const int tile_xcd = [] {
        const char *e = std::getenv("FEDM_FS_TILE_XCD");
        return (e && e[0] == '0') ? 0 : 1;
    }();
#define FEDM_TILE_SWEEPS(SL, IDX)                                                                                \
    do {                                                                                                         \
        if (lds > 64 * 1024 && lds > lds_granted[IDX]) {                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&fs_tile_sweeps_kernel<NS, W, SL>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            lds_granted[IDX] = lds;                                                                              \
        }                                                                                                        \
        hipLaunchKernelGGL((fs_tile_sweeps_kernel<NS, W, SL>), g, b, lds, c.stream, ft.d_tile, ft.record, ft.max_vertices, \
                           ft.max_rows, ft.width, ft.d_vertex, ft.d_rowinfo, ft.d_cols, c.d_slice_boff, c.d_s16, zmask, g32, in, out32, z, wt, last ? 1 : 0, \
                           x0, cpl32, b0, tile_xcd);                                                             \
    } while (0)
    if constexpr (NS > 2) {
        FEDM_TILE_SWEEPS(1, 0);          // (fs_tiles_get: a row a thread)
    } else if (slots <= 2) FEDM_TILE_SWEEPS(2, 0);
    else if (slots <= 3) FEDM_TILE_SWEEPS(3, 1);
    else if (slots <= 4) FEDM_TILE_SWEEPS(4, 2);
    else if (slots <= 6) FEDM_TILE_SWEEPS(6, 3);
    else FEDM_TILE_SWEEPS(8, 4);
#undef FEDM_TILE_SWEEPS
}

// The n_sweeps sweeps after the first stage (g32: Duu^-1 alpha t_u; zs0, w: the weights of fs_finish_t), at most
// `depth` of them per launch; the last one writes z (and the lagged coupling product into b0 when cpl32 is set).
// false: not applicable here, the caller runs the sweeps one by one.
bool fs_tiles_sweeps(Ctx &c, int n_sweeps, unsigned zmask, const float *g32, float *ping0, float *ping1, double *z,
                     const double *x0, const float *cpl32, double *b0) {
    if (n_sweeps < 1) return false;
    FsTiles *ft = fs_tiles_get(c);
    if (!ft) return false;
    const float *in = g32;
    int done = 0, launch = 0;
    // even shares: 7 sweeps at depth 5 are 4 + 3, not 5 + 2 (the layers cost more than linearly)
    const int n_launches = (n_sweeps + ft->depth - 1) / ft->depth;
    while (done < n_sweeps) {
        const int left = n_sweeps - done, launches_left = n_launches - launch;
        const int n = (left + launches_left - 1) / launches_left;
        FsTileWeights wt;
        wt.n = n;
        wt.zs = done == 0 ? c.fs_w[0] : 1.0;
        for (int k = 0; k < FS_TILE_MAX_SWEEPS; ++k) wt.w[k] = k < n ? c.fs_w[done + 1 + k] : 0.0;
        const bool last = done + n == n_sweeps;
        float *out = last ? nullptr : (launch & 1 ? ping1 : ping0);
#define FEDM_TILE_W(NS_)                                                                                        \
    do {                                                                                                        \
        if (ft->width <= 7) fs_tiles_launch<NS_, 7>(c, *ft, zmask, g32, in, out, z, wt, last, x0, cpl32, b0);   \
        else if (ft->width <= 9) fs_tiles_launch<NS_, 9>(c, *ft, zmask, g32, in, out, z, wt, last, x0, cpl32, b0); \
        else fs_tiles_launch<NS_, 12>(c, *ft, zmask, g32, in, out, z, wt, last, x0, cpl32, b0);                 \
    } while (0)
        if (c.ns == 4) FEDM_TILE_W(4);
        else FEDM_TILE_W(2);
#undef FEDM_TILE_W
        if (hipPeekAtLastError() != hipSuccess) {
            // a refused launch (not seen on the meshes tried): no tiles from now on; outside a capture the caller
            // repeats the sweeps one by one, inside one the capture fails and the solver falls back to plain launches
            set_error(std::string("fs_tile_sweeps_kernel launch: ") + hipGetErrorString(hipGetLastError()));
            ft->usable = false;
            c.fs_tiles_state = -1;
            return false;
        }
        in = out;
        done += n;
        ++launch;
    }
    return true;
}

// The n sweeps x <- x + w[s] Dinv (b - A x) of the finest level as one launch (see mg_tile_sweeps_kernel), from xin; the
// result goes to out[row * ostride + ooff].  false: not applicable here (nothing launched).
bool mg_tiles_sweeps(Ctx &c, const EllMat &A, const double *b, const double *xin, int n, const double *w, double *out,
                     int ostride, int ooff) {
    static const bool off = [] {
        const char *e = std::getenv("FEDM_MG_TILES");
        return e && e[0] == '0';
    }();
    if (off || c.mg_tiles_off || c.comm || n < 2 || n > 4 || !A.dinv || A.n_rows != c.nv) return false;
    // (only tiles that exist: this runs inside stream captures -- the cycle's own graph, the Krylov steps'; they are
    // built with the field split's set-up, fs_tiles_prepare)
    FsTiles *ft = (c.fs_tiles_state > 0 && c.fs_tiles && c.fs_tiles->usable) ? c.fs_tiles : nullptr;
    if (!ft || n > ft->depth) return false;
    MgTileWeights wt;
    wt.n = n;
    for (int k = 0; k < 4; ++k) wt.w[k] = k < n ? w[k] : 0.0;
    const int T = ft->threads;
    const dim3 g(ft->n_tiles), bl(T);
    const int slots = (ft->max_rows + T - 1) / T;
    const int neq2 = c.neq * c.neq;
    const size_t mg_lds = mg_tiles_lds_bytes(ft->width, ft->max_vertices, ft->max_rows);
    if (mg_lds > ft->lds_limit) return false;     // (the sweeps as kernels of their own)
    static size_t mg_granted[6] = {0, 0, 0, 0, 0, 0};
#define FEDM_MG_TILE(WW, SL, IDX)                                                                                 \
    do {                                                                                                          \
        if (mg_lds > 64 * 1024 && mg_lds > mg_granted[IDX]) {                                                     \
            hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_tile_sweeps_kernel<WW, SL>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)mg_lds);                         \
            mg_granted[IDX] = mg_lds;                                                                             \
        }                                                                                                         \
        hipLaunchKernelGGL((mg_tile_sweeps_kernel<WW, SL>), g, bl, mg_lds,                                        \
                           c.stream, ft->d_tile, ft->record, ft->max_vertices, ft->max_rows, ft->width, ft->d_vertex, \
                           ft->d_rowinfo, ft->d_cols, c.d_slice_boff, c.d_val, neq2, neq2 - 1, A.dinv, b, xin, wt, out, ostride, ooff, 1); \
    } while (0)
#define FEDM_MG_TILE_S(WW, BASE)                                                                                  \
    do {                                                                                                          \
        if (slots <= 2) FEDM_MG_TILE(WW, 2, BASE);                                                                \
        else if (slots <= 3) FEDM_MG_TILE(WW, 3, BASE + 1);                                                       \
        else return false;                                                                                        \
    } while (0)
    if (ft->width <= 7) FEDM_MG_TILE_S(7, 0);
    else if (ft->width <= 9) FEDM_MG_TILE_S(9, 2);
    else FEDM_MG_TILE_S(12, 4);
#undef FEDM_MG_TILE_S
#undef FEDM_MG_TILE
    if (hipPeekAtLastError() != hipSuccess) {
        set_error(std::string("mg_tile_sweeps_kernel launch: ") + hipGetErrorString(hipGetLastError()));
        c.mg_tiles_off = true;
        return false;
    }
    return true;
}

}  // namespace fedm

// pseg_post.hip -- integer pre/post-process kernels: 4-connected component labelling
// (lock-free union-find, roots = minimum linear index, hence deterministic), majority vote,
// bounding-box painting, colour/overlay masks, Otsu histogram + glyph-height statistics.
//
// All of it is HBM-bound byte/integer work: one thread per pixel (or per 4 pixels), coalesced
// row-major access, wave-aggregated atomics where many lanes hit one counter.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "pseg_common.h"

namespace pseg {

// ---------------------------------------------------------------------------------------------
// connected components (4-connectivity)
// ---------------------------------------------------------------------------------------------
// MODE 0: foreground = bin != 0, neighbours connect when both are foreground
//         (cv2.connectedComponentsWithStats(binary, connectivity=4), lib/postprocess.py:10).
// MODE 1: every pixel is foreground, neighbours connect when their class is equal
//         (the per-class labelling of lib/postprocess.py:33 done for all classes at once).
// MODE 2: as MODE 0 with the two upper diagonals as neighbours too (connectivity=8, lib/evaluation.py:73,84-85).
template <int MODE>
__device__ __forceinline__ bool is_fg(const uint8_t* bin, const int64_t* cls, int p) {
    return MODE != 1 ? bin[p] != 0 : true;
}
template <int MODE>
__device__ __forceinline__ bool connects(const uint8_t* bin, const int64_t* cls, int p, int q) {
    return MODE != 1 ? (bin[q] != 0) : (cls[p] == cls[q]);
}

__device__ __forceinline__ int uf_find(const int* L, int x) {
    int r = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (r != x) {
        x = r;
        r = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

// find with path halving: every other node of the walked chain is pointed at its grandparent.  Links only ever point at a member of
// the same component with a smaller index, a halving store replaces a non-root's link by one further up its own chain, and a
// root (the only kind of node a union's atomicMin must find unchanged) is never stored to -- racing walkers may undo each other's
// shortcut, never the structure.  Without it a component that percolates through a figure keeps chains of one hop per tile it
// crosses (tens of dependent loads for every later find).
__device__ __forceinline__ int uf_find_halve(int* L, int x) {
    while (true) {
        const int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const int g = __hip_atomic_load(&L[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g == p) return p;
        __hip_atomic_store(&L[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = g;
    }
}

// Labels only ever decrease and always point at a member of the same component, so a stale
// read costs extra iterations, never correctness.
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    while (true) {
        a = uf_find_halve(L, a);
        b = uf_find_halve(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// Row pass without atomics: a wave covers 64 consecutive pixels; a ballot of the lanes that do NOT
// continue their left neighbour's run gives every lane the start of its run inside the wave, which
// becomes its label directly (roots stay the minimum linear index).  Only a run that continues
// across the wave's first lane needs a union, done by the column kernel after all labels exist.
template <int MODE>
__global__ __launch_bounds__(256) void ccl_rows_kernel(const uint8_t* bin, const int64_t* cls, int* L, int H, int W) {
    const int n = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool fg = false, link = false;
    if (p < n) {
        fg = is_fg<MODE>(bin, cls, p);
        const int x = p % W;
        link = fg && x > 0 && is_fg<MODE>(bin, cls, p - 1) && connects<MODE>(bin, cls, p, p - 1);
    }
    const unsigned long long brk = __ballot(!link);                 // lanes that start a run (or are background)
    const unsigned long long upto = brk & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int s = upto ? 63 - __clzll((long long)upto) : 0;          // run start lane (0: the run entered the wave)
    if (p < n) L[p] = fg ? p - (lane - s) : -1;
}

// Column pass + the row unions across wave boundaries.  A vertical union is only issued where an
// overlap between the run above and this run begins: if the left neighbours are linked along both
// rows and vertically, their union already joins the same two runs.
template <int MODE>
__global__ __launch_bounds__(256) void ccl_cols_kernel(const uint8_t* bin, const int64_t* cls, int* L, int H, int W) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    if (!is_fg<MODE>(bin, cls, p)) return;
    const int y = p / W, x = p - y * W;
    const bool link = x > 0 && is_fg<MODE>(bin, cls, p - 1) && connects<MODE>(bin, cls, p, p - 1);
    if (link && (threadIdx.x & 63) == 0) uf_union(L, p, p - 1);    // run continues from the previous wave
    if (y == 0) return;
    const int q = p - W;
    if (!(is_fg<MODE>(bin, cls, q) && connects<MODE>(bin, cls, p, q))) {
        if (MODE == 2) {       // the pixel above is paper: the diagonals are separate runs (above ink, they share its run)
            if (x > 0 && !link && bin[q - 1] != 0) uf_union(L, p, q - 1);   // with a left link, p-1 joins q-1 vertically
            if (x + 1 < W && bin[q + 1] != 0) uf_union(L, p, q + 1);
        }
        return;
    }
    if (link) {
        const bool link_up = is_fg<MODE>(bin, cls, q - 1) && connects<MODE>(bin, cls, q, q - 1);
        const bool up_left = is_fg<MODE>(bin, cls, q - 1) && connects<MODE>(bin, cls, p - 1, q - 1);
        if (link_up && up_left) return;
    }
    uf_union(L, p, q);
}

// path compression; with `hist`, the vote's per-root counters (ncls per root) are cleared on the way
__global__ void ccl_compress_kernel(int* L, int n, int* hist = nullptr, int ncls = 0) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || L[p] < 0) return;
    const int r = uf_find(L, p);
    L[p] = r;
    if (hist && r == p)
        for (int c = 0; c < ncls; ++c) hist[(size_t)p * ncls + c] = 0;
}

// ---- tile-local labelling (default) -------------------------------------------------------------------------
// A workgroup labels a 16 x 64 pixel tile entirely in LDS -- the same run-based row pass (one wave = one 64-pixel tile
// row), column unions and path compression as above, on LDS atomics -- and writes each pixel the GLOBAL index of its
// tile-local root (the raster-first pixel of the component's part inside the tile).  Only pixels on tile borders then
// need global unions (8 % of the page), followed by the global compress pass.  Roots are still the component's minimum
// linear index: the raster-first pixel of a component is the raster-first pixel of its tile part, hence a local root.
constexpr int CT_H = 32, CT_W = 64;

__device__ __forceinline__ int lds_find(const int* lab, int x) {
    int r = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (r != x) { x = r; r = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    return x;
}
__device__ __forceinline__ void lds_union(int* lab, int a, int b) {
    while (true) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void ccl_tile_kernel(const uint8_t* bin, const int64_t* cls, int* L, int H, int W, int* hist = nullptr,
                                                       int ncls = 0) {
    __shared__ int lab[CT_H * CT_W];
    const int tiles_x = (W + CT_W - 1) / CT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tx * CT_W + lane;
    // rows wave, wave + 4, ...: row pass (ballot of run starts)
    bool fgr[CT_H / 4], linkr[CT_H / 4];
#pragma unroll
    for (int j = 0; j < CT_H / 4; ++j) {
        const int r = wave + 4 * j, y = ty * CT_H + r;
        bool fg = false, link = false;
        if (y < H && x < W) {
            const int p = y * W + x;
            fg = is_fg<MODE>(bin, cls, p);
            link = fg && lane > 0 && is_fg<MODE>(bin, cls, p - 1) && connects<MODE>(bin, cls, p, p - 1);
        }
        const unsigned long long brk = __ballot(!link);
        const unsigned long long upto = brk & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int s = 63 - __clzll((long long)upto);             // lane 0 never links: upto != 0
        lab[r * CT_W + lane] = fg ? r * CT_W + s : -1;
        fgr[j] = fg; linkr[j] = link;
    }
    __syncthreads();
    // column pass inside the tile (unions only where an overlap of two runs begins)
#pragma unroll
    for (int j = 0; j < CT_H / 4; ++j) {
        const int r = wave + 4 * j, y = ty * CT_H + r;
        if (!fgr[j] || r == 0) continue;
        const int p = y * W + x, q = p - W;
        const int li = r * CT_W + lane;
        if (!(is_fg<MODE>(bin, cls, q) && connects<MODE>(bin, cls, p, q))) {
            if (MODE == 2) {
                if (lane > 0 && !linkr[j] && bin[q - 1] != 0) lds_union(lab, li, li - CT_W - 1);
                if (lane + 1 < CT_W && x + 1 < W && bin[q + 1] != 0) lds_union(lab, li, li - CT_W + 1);
            }
            continue;
        }
        if (linkr[j]) {
            const bool link_up = is_fg<MODE>(bin, cls, q - 1) && connects<MODE>(bin, cls, q, q - 1);
            const bool up_left = is_fg<MODE>(bin, cls, q - 1) && connects<MODE>(bin, cls, p - 1, q - 1);
            if (link_up && up_left) continue;
        }
        lds_union(lab, li, li - CT_W);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CT_H / 4; ++j) {
        const int r = wave + 4 * j, y = ty * CT_H + r;
        if (y >= H || x >= W) continue;
        int g = -1;
        if (fgr[j]) {
            const int root = lds_find(lab, r * CT_W + lane);
            g = (ty * CT_H + root / CT_W) * W + tx * CT_W + (root % CT_W);
            // the vote's counters live in the root's row of a page-sized array: every tile-local root clears its row here (a
            // component's final root is one of them: border unions only ever redirect a root to another tile's root)
            if (hist && root == r * CT_W + lane)
                for (int c = 0; c < ncls; ++c) hist[(size_t)g * ncls + c] = 0;
        }
        L[y * W + x] = g;
    }
}

// unions across tile borders.  Threads exist only for border pixels: the first row of every tile row band but the
// first (nby * W pixels, joined with the row above) and the first column of every tile column but the first (nbx * H
// pixels, joined with the pixel to the left; MODE 2 also the two diagonals that cross that vertical border).
template <int MODE>
__global__ __launch_bounds__(256) void ccl_border_kernel(const uint8_t* bin, const int64_t* cls, int* L, int H, int W, int nby, int nbx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nby * W) {                                    // horizontal borders
        const int y = (t / W + 1) * CT_H, x = t % W;
        const int p = y * W + x, q = p - W;
        if (!is_fg<MODE>(bin, cls, p)) return;
        if (is_fg<MODE>(bin, cls, q) && connects<MODE>(bin, cls, p, q)) {
            // as inside a tile: skip where the left neighbours already join the same two runs -- but not at a tile corner: there
            // the left links are border unions themselves, and the one of p's row is skipped for THIS pair's sake
            if (MODE == 0 && (x % CT_W) != 0 && bin[p - 1] != 0 && bin[q - 1] != 0) return;
            uf_union(L, p, q);
        } else if (MODE == 2) {
            if (x > 0 && bin[q - 1] != 0) uf_union(L, p, q - 1);
            if (x + 1 < W && bin[q + 1] != 0) uf_union(L, p, q + 1);
        }
        return;
    }
    const int u = t - nby * W;
    if (u >= nbx * H) return;                             // vertical borders
    const int x = (u / H + 1) * CT_W, y = u % H;
    const int p = y * W + x;
    if (!is_fg<MODE>(bin, cls, p)) return;
    if (is_fg<MODE>(bin, cls, p - 1) && connects<MODE>(bin, cls, p, p - 1)) {
        // (the pair above joins the same two column runs, unless a horizontal border runs between the two pairs)
        if (!(MODE == 0 && (y % CT_H) != 0 && bin[p - W] != 0 && bin[p - W - 1] != 0)) uf_union(L, p, p - 1);
    }
    if (MODE == 2) {
        // up-left diagonal of p (needed when the pixel above p is paper) and, seen from the other side, the up-right
        // diagonal of the pixel below-left of p (needed when the pixel above THAT one, i.e. left of p, is paper);
        // rows on a horizontal border are covered by the branch above for the first, here for the second
        if (y > 0 && (y % CT_H) != 0 && bin[p - W] == 0 && bin[p - W - 1] != 0) uf_union(L, p, p - W - 1);
        if (y + 1 < H && ((y + 1) % CT_H) != 0 && bin[p - 1] == 0 && bin[p + W - 1] != 0) uf_union(L, p, p + W - 1);
    }
}

template <int MODE>
static int ccl_run(const uint8_t* d_bin, const int64_t* d_cls, int* d_L, int H, int W,
                   hipStream_t st, int* d_hist = nullptr, int ncls = 0) {
    const int n = H * W;
    const int grid = cdiv(n, 256);
    if (PSEG_KNOB("PSEG_CCL_GLOBAL")) {
        ccl_rows_kernel<MODE><<<grid, 256, 0, st>>>(d_bin, d_cls, d_L, H, W);
        ccl_cols_kernel<MODE><<<grid, 256, 0, st>>>(d_bin, d_cls, d_L, H, W);
    } else {
        ccl_tile_kernel<MODE><<<cdiv(W, CT_W) * cdiv(H, CT_H), 256, 0, st>>>(d_bin, d_cls, d_L, H, W, d_hist, ncls);
        const int nby = (H - 1) / CT_H, nbx = (W - 1) / CT_W;
        if (nby * W + nbx * H > 0)
            ccl_border_kernel<MODE><<<cdiv(nby * W + nbx * H, 256), 256, 0, st>>>(d_bin, d_cls, d_L, H, W, nby, nbx);
    }
    // the vote (d_hist given, tile path) resolves and compresses the labels inside its counting pass: one pass over L less
    if (!d_hist || PSEG_KNOB("PSEG_CCL_GLOBAL")) ccl_compress_kernel<<<grid, 256, 0, st>>>(d_L, n, d_hist, ncls);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// roots (minimum linear index of the component, -1 on paper) of the ink components of `d_bin`
int ccl_roots(const uint8_t* d_bin, int* d_L, int H, int W, int connectivity, hipStream_t st) {
    return connectivity == 8 ? ccl_run<2>(d_bin, nullptr, d_L, H, W, st) : ccl_run<0>(d_bin, nullptr, d_L, H, W, st);
}

// ---------------------------------------------------------------------------------------------
// majority vote (lib/postprocess.py:9-26)
// ---------------------------------------------------------------------------------------------
// ---- page-global path (PSEG_CCL_GLOBAL, or more than V_NCLS_MAX classes): ccl_run + the two kernels below ----
// hist[root * ncls + class] += 1.  The histogram is as large as the page (one row of ncls counters per possible
// root), so every counter update that reaches memory is a scattered read-modify-write: the kernel's cost is the
// NUMBER of global atomics.  A workgroup therefore owns a 32 x 32 pixel tile (a glyph spans one to four tiles
// instead of thirty row segments), merges its pixels' (root, class) keys in an LDS hash table and flushes one atomic
// per distinct key.  Only the rows of roots are ever touched: the labelling's compress pass clears exactly those
// instead of a page-sized memset.
constexpr int VT = 32, VSLOTS = 2048;
template <typename LT>
__global__ __launch_bounds__(256) void vote_count_kernel(int* L, const LT* pred, int* hist, int H, int W, int ncls) {
    __shared__ int keys[VSLOTS];
    __shared__ int vals[VSLOTS];
    for (int i = threadIdx.x; i < VSLOTS; i += 256) { keys[i] = -1; vals[i] = 0; }
    __syncthreads();
    const int tiles_x = (W + VT - 1) / VT;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y = ty * VT + (threadIdx.x >> 3), x0 = tx * VT + (threadIdx.x & 7) * 4;
    int key = -1, cnt = 0;                        // run of equal keys inside this thread's four pixels
    auto put = [&](int k, int c) {
        unsigned slot = ((unsigned)k * 0x9E3779B1u) >> (32 - 11);
        while (true) {
            const int old = atomicCAS(&keys[slot], -1, k);
            if (old == -1 || old == k) { atomicAdd(&vals[slot], c); return; }
            slot = (slot + 1) & (VSLOTS - 1);
        }
    };
    if (y < H)
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            int k = -1;
            if (x < W) {
                const size_t p = (size_t)y * W + x;
                int r = L[p];
                if (r >= 0) {                                        // resolve to the component's root and leave it in L for the apply pass
                    const int r0 = r;
                    r = uf_find(L, r);
                    if (r != r0) L[p] = r;
                }
                const int64_t c = (int64_t)pred[p];
                if (r >= 0 && c >= 0 && c < ncls) k = r * ncls + (int)c;
            }
            if (k == key) { ++cnt; continue; }
            if (key >= 0) put(key, cnt);
            key = k; cnt = 1;
        }
    if (key >= 0) put(key, cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < VSLOTS; i += 256)
        if (keys[i] >= 0) atomicAdd(&hist[keys[i]], vals[i]);
}

// Every ink pixel reads its component's counters (a few cache lines per component) and takes
// np.argmax(bins) = the lowest class among the most frequent (lib/postprocess.py:22-23).
template <typename LT>
__global__ void vote_apply_kernel(const int* L, const int* hist, LT* pred, int n, int ncls) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int r = L[p];
    if (r < 0) return;
    const int* h = hist + (size_t)r * ncls;
    int best = 0, bv = h[0];
    for (int c = 1; c < ncls; ++c) {
        const int v = h[c];
        if (v > bv) { bv = v; best = c; }
    }
    pred[p] = (LT)best;
}

// ---- the vote's own tile pass (default) ------------------------------------------------------------------------
// One workgroup labels a 32 x 64 tile in LDS AND counts the classes of every tile-local component there, so that the
// page-wide passes over a 4 B/px label image (write it, resolve + count it, read it again to apply) disappear:
//   * the binarisation and the class map are read ONCE per pixel and become bit masks: a tile row's ink is one 64-bit ballot,
//     its pixels of class c another.  (ccl_tile_kernel re-reads bytes for each test: ~5 byte loads per pixel, and a CU's
//     texture addresser takes a 64-lane byte load at the pace of a 64-lane dword load.)
//   * everything after that is dense over RUNS (maximal stretches of ink in a row; see vote_tile_kernel), not pixels.
//   * a component with no ink neighbour across the tile's edge is CLOSED: its counts are final, the winner is written to those
//     of its pixels that have another class, and nothing else of it ever reaches memory.  An OPEN component gets the next of
//     the tile's V_OPEN_MAX root slots (id = tile * V_OPEN_MAX + slot): its counts go to row `id` of a compact histogram
//     (plain stores: no clearing pass, no atomics), its slot to the tile's rim table at every pixel of it that has an ink
//     neighbour across the edge (all that the border unions look at), one record per run to the tile's run list.
// No page-sized array is left: the union-find across tiles runs over the root slot ids (tiles * 192 ints: L2-resident), the
// rim table is 192 bytes per tile.
// vote_border_kernel then joins the open components across tile edges, vote_merge_kernel adds the counts of every root slot
// that is no longer a root to its final root's row, vote_apply_runs_kernel resolves each listed run and writes its pixels.
// Measured on the way (configs[4]'s page, same box as the page-global path at 0.158 ms): per-pixel labels + per-run counting
// in LDS with a page-sized label image 0.116; the same with eight waves per tile and compact slots 0.125; persistent
// workgroups that prefetch the next tile 0.151 (imbalance; 103 registers -> half the occupancy); run-based, eight waves
// 0.131; run-based, four waves 0.112.  SQ counters of the run-based kernel: 905 instructions per wave, waves waiting
// (SQ_WAIT_ANY) 70 % of their cycles -- the kernel is bound by latency chains at full occupancy, not by issue or bytes.
// (Round 3's tile-local attempt kept per-pixel LDS atomics for labelling AND counting and a page-sized label image: 0.262.)
constexpr int V_OPEN_MAX = 2 * (CT_H + CT_W);   // an open root owns at least one pixel of the tile's rim
constexpr int V_RUN_MAX = CT_H * CT_W / 2;      // runs of a tile: at most every other pixel starts one
constexpr int V_NCLS_MAX = 12;                  // LDS: 8 KiB of labels + 4 KiB of counters per class, under the 64 KiB a launch gets unasked (more classes: the page-global path)
// a run of an open component in its tile's run list: tile row | first lane << 5 | last lane << 11 | its root's slot << 17
typedef unsigned VRun;
// rim table of a tile: slot of the root of the pixel at [0, 64) top row, [64, 128) bottom row, [128, 160) left column, [160, 192) right column
constexpr int V_RIM = 2 * (CT_H + CT_W);
constexpr int V_TASK_MAX = 64;                 // border unions a tile lists: <= 32 overlap starts along its top row, <= 16 pairs across its left edge
// The kernel works on RUNS (maximal stretches of ink in a tile row), not pixels: after the one pass that turns the two maps
// into bit masks -- ink per row, one mask per class and row -- a tile is its run list (row, first, last lane; ordered by
// row, then column, so that the run holding bit b of row r is rowbase[r] + popcount(first-lane mask of r up to b) - 1), and
// every later phase is dense over runs: unions with the overlapping runs of the row above, the root, the class counts
// (popcount(class mask & run span)), open / closed, the winner.  A text tile has 100-300 runs for 2 048 pixels.
// counters: 16 bits per (class, root run) -- a tile has 2 048 pixels -- two runs to a word, bumped with 32-bit LDS adds
// (no carry can cross: a half never exceeds 2 048)
__device__ __forceinline__ unsigned long long bits_le(int b) { return b >= 63 ? ~0ull : ((2ull << b) - 1ull); }   // bits 0 .. b
template <typename LT, int NW>
__global__ __launch_bounds__(NW * 64) void vote_tile_kernel(const uint8_t* __restrict__ bin, LT* pred, int* __restrict__ P, int* __restrict__ hist,
                                                        uint8_t* __restrict__ rimtab, int* __restrict__ rootn, VRun* __restrict__ runs,
                                                        int* __restrict__ runn, unsigned short* __restrict__ tasks, int* __restrict__ taskn,
                                                        int H, int W, int ncls) {
    constexpr int NTH = NW * 64, RPW = CT_H / NW, HALF = V_RUN_MAX / 2;
    static_assert(NW >= 4 && RPW % 4 == 0, "four rows of a wave are one 64-lane dword load; waves 0-3 fetch the tile's surroundings");
    extern __shared__ __attribute__((aligned(16))) unsigned vsm[];
    unsigned* const cnt = vsm;                                                            // [ncls][V_RUN_MAX / 2]
    unsigned long long* const cmask = (unsigned long long*)(vsm + ncls * (V_RUN_MAX / 2));   // [class][row]: pixels of that class
    __shared__ unsigned long long m64[CT_H + 2];          // ink of tile rows -1 .. CT_H
    __shared__ unsigned lr[2];                            // ink left / right of the tile, bit = tile row
    __shared__ int rowcnt[CT_H], rowbase[CT_H + 1];       // runs of a row, runs before it
    __shared__ unsigned rl[V_RUN_MAX];                    // the runs: row | first lane << 5 | last lane << 11 (| an open root's slot << 17)
    __shared__ int parent[V_RUN_MAX];                     // union-find over run indices (staging area of the maps before that)
    __shared__ unsigned openbits[V_RUN_MAX / 32];
    __shared__ int nboth, ntask;                          // open runs | open roots << 16; border unions of this tile
    const int tiles_x = (W + CT_W - 1) / CT_W;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y0 = ty * CT_H, x0 = tx * CT_W, x = x0 + lane;
    const int rw0 = wave * RPW;                           // this wave's rows: rw0 .. rw0 + RPW - 1
    for (int i = threadIdx.x; i < ncls * HALF / 4; i += NTH) ((uint4*)cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < V_RUN_MAX / 32) openbits[threadIdx.x] = 0;
    if (threadIdx.x == 0) { nboth = 0; ntask = 0; }
    // ---- the two maps -> masks.  A wave's RPW rows of 64 bytes are ONE dword load per map (lane -> row lane / 16, dword
    // lane % 16) where the tile lies inside the page and rows are dword-aligned, handed to the lanes of the columns through
    // LDS; byte loads per row otherwise.  Waves 0, 1 also fetch the rows above / below the tile, waves 2, 3 the columns
    // left / right of it.
    uint8_t bb[RPW];
    long long cc[RPW];
    const bool wide = (W & 3) == 0 && x0 + CT_W <= W;
    if (wide) {
        uint8_t* const stg = (uint8_t*)(parent + wave * (V_RUN_MAX / NW));   // this wave's own: RPW x 64 B ink, RPW x 64 B classes
        static_assert(V_RUN_MAX * 4 / NW == 2 * RPW * 64, "the staging area is the parent array");
        unsigned vb[RPW / 4], vc[RPW / 4];
#pragma unroll
        for (int h = 0; h < RPW / 4; ++h) {
            const int yl = y0 + rw0 + 4 * h + (lane >> 4);
            const size_t o = (size_t)yl * W + x0 + (lane & 15) * 4;
            vb[h] = yl < H ? *(const unsigned*)(bin + o) : 0u;
            vc[h] = (sizeof(LT) == 1 && yl < H) ? *(const unsigned*)((const uint8_t*)pred + o) : 0u;
        }
#pragma unroll
        for (int h = 0; h < RPW / 4; ++h) {
            ((unsigned*)stg)[h * 64 + lane] = vb[h];
            if (sizeof(LT) == 1) ((unsigned*)stg)[RPW * 16 + h * 64 + lane] = vc[h];
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            bb[j] = stg[j * 64 + lane];
            if (sizeof(LT) == 1) cc[j] = stg[RPW * 64 + j * 64 + lane];
            else cc[j] = y0 + rw0 + j < H ? (long long)pred[(size_t)(y0 + rw0 + j) * W + x] : -1ll;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int y = y0 + rw0 + j;
            const bool in = y < H && x < W;
            bb[j] = in ? bin[(size_t)y * W + x] : (uint8_t)0;
            cc[j] = in ? (long long)pred[(size_t)y * W + x] : -1ll;   // (paper included: one round trip for both maps)
        }
    }
    bool hv = false;
    if (wave < 2) {
        const int y = wave == 0 ? y0 - 1 : y0 + CT_H;
        hv = y >= 0 && y < H && x < W && bin[(size_t)y * W + x] != 0;
    } else if (wave < 4) {
        const int y = y0 + (lane & 31), xx = wave == 2 ? x0 - 1 : x0 + CT_W;
        hv = lane < 32 && y < H && xx >= 0 && xx < W && bin[(size_t)y * W + xx] != 0;
    }
    unsigned long long mr[RPW];
    int nruns_w = 0;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = rw0 + j;
        const bool fg = bb[j] != 0;
        const unsigned long long m = __ballot(fg);
        mr[j] = m;
        const int nr = __popcll(m & ~(m << 1));
        nruns_w += nr;
        if (lane == 0) { m64[r + 1] = m; rowcnt[r] = nr; }
        if (m) {                                          // (uniform)
            const int cl = (fg && cc[j] >= 0 && cc[j] < ncls) ? (int)cc[j] : -1;
            for (int c = 0; c < ncls; ++c) {
                const unsigned long long cm = __ballot(cl == c);
                if (lane == 0) cmask[c * CT_H + r] = cm;
            }
        }
    }
    if (wave < 4) {
        const unsigned long long b = __ballot(hv);
        if (lane == 0) {
            if (wave == 0) m64[0] = b;
            else if (wave == 1) m64[CT_H + 1] = b;
            else lr[wave - 2] = (unsigned)b;
        }
    }
    if (!__syncthreads_or(nruns_w)) {                     // a tile of paper
        if (threadIdx.x == 0) { rootn[tile] = 0; runn[tile] = 0; taskn[tile] = 0; }
        return;
    }
    // ---- the run list (every wave scans the 32 row counts itself), every run its own parent ----
    int excl, n_all;
    {
        const int v = lane < CT_H ? rowcnt[lane] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < CT_H; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        excl = inc - v;
        n_all = __shfl(inc, CT_H - 1);
        if (wave == 0 && lane <= CT_H) rowbase[lane] = lane < CT_H ? excl : n_all;
    }
    const unsigned long long le = bits_le(lane);          // lanes <= this one
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = rw0 + j;
        const unsigned long long m = mr[j];
        const int base = __shfl(excl, r);
        if (m == 0) continue;
        const unsigned long long fm = m & ~(m << 1);
        if ((fm >> lane) & 1) {
            const unsigned long long brk = ~(m & (m << 1));   // lanes that do not continue the lane to their left
            const unsigned long long above = brk & ~le;
            const int e = above ? __builtin_ctzll(above) - 1 : 63;
            const int idx = base + __popcll(fm & (le >> 1));
            rl[idx] = (unsigned)(r | lane << 5 | e << 11);
            parent[idx] = idx;
        }
    }
    __syncthreads();
    // ---- unions with the runs of the row above that overlap ----
    for (int t = threadIdx.x; t < n_all; t += NTH) {
        const unsigned rec = rl[t];
        const int r = rec & 31, s0 = (rec >> 5) & 63, e = (rec >> 11) & 63;
        if (r == 0) continue;
        const unsigned long long mp = m64[r];             // (row r - 1)
        unsigned long long ov = bits_le(e) & ~(bits_le(s0) >> 1) & mp;
        const unsigned long long fmp = mp & ~(mp << 1);
        const int base = rowbase[r - 1];
        while (ov) {
            const int b = __builtin_ctzll(ov);
            const int u = base + __popcll(fmp & bits_le(b)) - 1;
            lds_union(parent, t, u);
            ov &= ~bits_le((rl[u] >> 11) & 63);
        }
    }
    __syncthreads();
    // ---- roots, class counts, open components ----
    const unsigned long long m_up = m64[0], m_dn = m64[CT_H + 1];
    const unsigned lr0 = lr[0], lr1 = lr[1];
    for (int t = threadIdx.x; t < n_all; t += NTH) {
        const unsigned rec = rl[t];
        const int r = rec & 31, s0 = (rec >> 5) & 63, e = (rec >> 11) & 63;
        const unsigned long long S = bits_le(e) & ~(bits_le(s0) >> 1);
        const int rt = lds_find(parent, t);
        if (rt != t) __hip_atomic_store(&parent[t], rt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (the unions are over: later phases read the root here)
        for (int c = 0; c < ncls; ++c) {
            const unsigned n = (unsigned)__popcll(cmask[c * CT_H + r] & S);
            if (n) __hip_atomic_fetch_add(&cnt[c * HALF + (rt >> 1)], n << ((rt & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool rim = (r == 0 && (S & m_up)) || (r == CT_H - 1 && (S & m_dn)) || (s0 == 0 && ((lr0 >> r) & 1)) || (e == CT_W - 1 && ((lr1 >> r) & 1));
        if (rim) atomicOr(&openbits[rt >> 5], 1u << (rt & 31));
    }
    __syncthreads();
    // ---- closed components: the winner to the pixels of another class; open roots: a slot, its counts, a root ----
    for (int t0 = 0; t0 < n_all; t0 += NTH) {
        const int t = t0 + threadIdx.x;
        bool open_root = false;
        int rt = 0;
        if (t < n_all) {
            rt = parent[t];
            const bool open = (openbits[rt >> 5] >> (rt & 31)) & 1;
            open_root = open && rt == t;
            if (!open) {
                const unsigned rec = rl[t];
                const int r = rec & 31, s0 = (rec >> 5) & 63, e = (rec >> 11) & 63;
                int best = 0, bv = -1;
                const unsigned* cp = cnt + (rt >> 1);
                const int sh = (rt & 1) * 16;
                for (int c = 0; c < ncls; ++c) {
                    const int v = (int)((cp[c * HALF] >> sh) & 0xffffu);
                    if (v > bv) { bv = v; best = c; }
                }
                unsigned long long wr = bits_le(e) & ~(bits_le(s0) >> 1) & ~cmask[best * CT_H + r];
                LT* const row = pred + (size_t)(y0 + r) * W + x0;
                while (wr) { const int b = __builtin_ctzll(wr); row[b] = (LT)best; wr &= wr - 1; }
            }
        }
        const unsigned long long orm = __ballot(open_root);
        if (orm) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&nboth, __popcll(orm) << 16);
            base = __shfl(base, 0);
            if (open_root) {
                const int k = (base >> 16) + __popcll(orm & (le >> 1)), id = tile * V_OPEN_MAX + k;
                rl[t] |= (unsigned)k << 17;
                P[id] = id;
                const unsigned* cp = cnt + (rt >> 1);
                const int sh = (rt & 1) * 16;
                for (int c = 0; c < ncls; ++c) hist[(size_t)id * ncls + c] = (int)((cp[c * HALF] >> sh) & 0xffffu);
            }
        }
    }
    __syncthreads();
    // ---- open runs: their record (with the root's slot), the slot where the border unions look ----
    uint8_t* const rim = rimtab + (size_t)tile * V_RIM;
    for (int t0 = 0; t0 < n_all; t0 += NTH) {
        const int t = t0 + threadIdx.x;
        bool open = false;
        unsigned rec = 0;
        int k = 0;
        if (t < n_all) {
            const int rt = parent[t];
            open = (openbits[rt >> 5] >> (rt & 31)) & 1;
            if (open) {
                rec = rl[t] & 0x1ffffu;
                k = (int)(rl[rt] >> 17);
                const int r = rec & 31, s0 = (rec >> 5) & 63, e = (rec >> 11) & 63;
                const unsigned long long S = bits_le(e) & ~(bits_le(s0) >> 1);
                if (r == 0) { unsigned long long w = S & m_up; while (w) { rim[__builtin_ctzll(w)] = (uint8_t)k; w &= w - 1; } }
                if (r == CT_H - 1) { unsigned long long w = S & m_dn; while (w) { rim[CT_W + __builtin_ctzll(w)] = (uint8_t)k; w &= w - 1; } }
                if (s0 == 0 && ((lr0 >> r) & 1)) rim[2 * CT_W + r] = (uint8_t)k;
                if (e == CT_W - 1 && ((lr1 >> r) & 1)) rim[2 * CT_W + CT_H + r] = (uint8_t)k;
                // the unions across this tile's top and left edge (the tiles below / right list theirs): one where an overlap with
                // a run of the row above begins; one per pixel pair across the left edge unless the pair above joins the same two
                // column runs.  slot | place in the neighbour's rim table << 8
                if (r == 0) {
                    const unsigned long long w = S & m_up;
                    unsigned long long st = w & ~(w << 1);
                    while (st) {
                        tasks[(size_t)tile * V_TASK_MAX + atomicAdd(&ntask, 1)] = (unsigned short)(k | (CT_W + __builtin_ctzll(st)) << 8);
                        st &= st - 1;
                    }
                }
                if (s0 == 0 && ((lr0 >> r) & 1) && !(r > 0 && (m64[r] & 1) && ((lr0 >> (r - 1)) & 1)))
                    tasks[(size_t)tile * V_TASK_MAX + atomicAdd(&ntask, 1)] = (unsigned short)(k | (2 * CT_W + CT_H + r) << 8);
            }
        }
        const unsigned long long om = __ballot(open);
        if (om) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&nboth, __popcll(om));
            base = __shfl(base, 0);
            if (open) runs[(size_t)tile * V_RUN_MAX + (base & 0xffff) + __popcll(om & (le >> 1))] = rec | (unsigned)k << 17;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { rootn[tile] = nboth >> 16; runn[tile] = nboth & 0xffff; taskn[tile] = ntask; }
}

// unions of the open components across tile edges, on the root slot ids: a wave per tile takes the tile's task list (written by
// vote_tile_kernel from its masks -- the binarisation is not read again), looks the neighbour's slot up in ITS rim table (places
// >= 2 CT_W: the left neighbour's right column, else the upper neighbour's bottom row) and joins.
// (a wave per tile, four tiles per workgroup: 12 288 one-wave workgroups of which most find an empty list were 17 us of dispatch)
__global__ __launch_bounds__(256) void vote_border_kernel(const uint8_t* __restrict__ rimtab, const unsigned short* __restrict__ tasks,
                                                          const int* __restrict__ taskn, int* P, int tiles_x, int tiles) {
    const int tile = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tiles) return;
    const int n = taskn[tile];
    if (lane >= n) return;
    const unsigned t = tasks[(size_t)tile * V_TASK_MAX + lane];
    const int place = (int)(t >> 8), nb = place >= 2 * CT_W ? tile - 1 : tile - tiles_x;
    uf_union(P, tile * V_OPEN_MAX + (int)(t & 255u), nb * V_OPEN_MAX + rimtab[(size_t)nb * V_RIM + place]);
}

// counts of the root slots that a border union redirected: added to the final root's row (and the slot pointed straight at
// it: the unions are over, the runs' finds end after one hop).  One wave per tile; the lanes of a wave that share their final
// root (a figure's percolating component: thousands of slots on the page, one root) are summed first, one lane adds.
__global__ __launch_bounds__(256) void vote_merge_kernel(int* P, int* hist, const int* rootn, int ncls, int tiles) {
    const int tile = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tiles) return;
    const int n = rootn[tile];
    for (int base = 0; base < n; base += 64) {
        int id = 0, r = -1;
        if (base + lane < n) {
            id = tile * V_OPEN_MAX + base + lane;
            r = uf_find_halve(P, id);
            if (r == id) r = -1;
            else __hip_atomic_store(&P[id], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lanes that share their final root are summed before one of them adds: group after group of the distinct roots among
        // the lanes (a text tile: a few, each a lane or two; a figure tile: one or two of dozens of lanes)
        int v[V_NCLS_MAX];
#pragma unroll
        for (int c = 0; c < V_NCLS_MAX; ++c) v[c] = (r >= 0 && c < ncls) ? hist[(size_t)id * ncls + c] : 0;
        while (true) {
            const unsigned long long act = __ballot(r >= 0);
            if (act == 0) break;
            const int r0 = __builtin_amdgcn_readlane(r, __builtin_ctzll(act));
            const bool mine = r == r0;
            const bool alone = __popcll(__ballot(mine)) == 1;
#pragma unroll
            for (int c = 0; c < V_NCLS_MAX; ++c) {
                if (c < ncls) {                               // (uniform)
                    int sum = mine ? v[c] : 0;
                    if (!alone) {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    }
                    if ((alone ? mine : lane == 0) && sum) atomicAdd(&hist[(size_t)r0 * ncls + c], sum);
                }
            }
            if (mine) r = -1;
        }
    }
}

// the runs of the open components: final root, np.argmax of its counters (the lowest class among the most frequent,
// lib/postprocess.py:22-23), written to the run's pixels
template <typename LT>
__global__ __launch_bounds__(256) void vote_apply_runs_kernel(const int* __restrict__ P, const int* __restrict__ hist, const VRun* __restrict__ runs,
                                                              const int* __restrict__ runn, LT* pred, int W, int ncls) {
    const int tile = blockIdx.x;
    const int n = runn[tile];
    if (n == 0) return;
    const int tiles_x = (W + CT_W - 1) / CT_W;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    for (int i = threadIdx.x; i < n; i += 256) {
        const VRun rec = runs[(size_t)tile * V_RUN_MAX + i];
        const int* h = hist + (size_t)uf_find(P, tile * V_OPEN_MAX + (int)(rec >> 17)) * ncls;
        int best = 0, bv = h[0];
        for (int c = 1; c < ncls; ++c) {
            const int v = h[c];
            if (v > bv) { bv = v; best = c; }
        }
        const int a = (rec >> 5) & 63, e = (rec >> 11) & 63;
        LT* o = pred + (size_t)(ty * CT_H + (int)(rec & 31)) * W + tx * CT_W;
        for (int k = a; k <= e; ++k) o[k] = (LT)best;
    }
}

// Workspace of the vote: grow-only, one per device, shared by every caller.  The tile path needs per tile 192 root slots
// (parent + ncls counters), a 4 KiB run list, a 2 KiB root-to-slot table and a 192-byte rim table: 65 MB for a 4096x3072
// 6-class page (the page-global path: a label image and a page-sized histogram, 50 + 302 MB); allocating and freeing it per
// call cost more than the kernels.  Calls are ordered by an event: a call on another stream than the previous user's first
// waits for that user's kernels, so two streams (or two threads) never run on the buffers at once.
// pseg_release_workspace() frees it.
struct VoteWs {
    void* p[3] = {nullptr, nullptr, nullptr};
    size_t bytes[3] = {0, 0, 0};
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
};
static std::mutex g_ws_mu;
static VoteWs g_ws[64];

int release_workspace(int dev) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    VoteWs& w = g_ws[dev & 63];
    if (w.used && w.last) (void)hipEventSynchronize(w.last);
    for (int i = 0; i < 3; ++i) { if (w.p[i]) (void)hipFree(w.p[i]); w.p[i] = nullptr; w.bytes[i] = 0; }
    if (w.last) (void)hipEventDestroy(w.last);
    w.last = nullptr; w.used = false; w.last_stream = nullptr;
    return PSEG_OK;
}

template <typename LT>
static int cc_vote_device(LT* d_pred, const uint8_t* d_bin, int H, int W, int ncls,
                          hipStream_t st) {
    if (H <= 0 || W <= 0) return PSEG_OK;
    if ((int64_t)H * W * std::max(ncls, 1) > 0x7fffffffLL)
        return fail(PSEG_EUNSUPPORTED, "page too large for 32-bit component indices");
    const int n = H * W;
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ws_mu);
    VoteWs& w = g_ws[dev & 63];
    if (!w.last) PSEG_HIP(hipEventCreateWithFlags(&w.last, hipEventDisableTiming));
    auto need = [&](int slot, size_t bytes) -> void* {
        if (w.bytes[slot] < bytes) {
            if (w.used) (void)hipEventSynchronize(w.last);     // the previous user may still be running on the old block
            if (w.p[slot]) (void)hipFree(w.p[slot]);
            w.p[slot] = nullptr; w.bytes[slot] = 0;
            if (hipMalloc(&w.p[slot], bytes) != hipSuccess) { w.p[slot] = nullptr; return nullptr; }
            w.bytes[slot] = bytes;
        }
        return w.p[slot];
    };
    const int tiles = cdiv(W, CT_W) * cdiv(H, CT_H);
    const bool global_path = PSEG_KNOB("PSEG_CCL_GLOBAL") || ncls > V_NCLS_MAX;
    if (w.used && w.last_stream != st) PSEG_HIP(hipStreamWaitEvent(st, w.last, 0));
    int rc = PSEG_OK;
    if (global_path) {
        // page-global union-find over pixel indices: a label image and one row of counters per possible root
        int* d_L = (int*)need(0, (size_t)n * 4);
        int* d_hist = (int*)need(1, (size_t)n * ncls * 4);
        if (!d_L || !d_hist) return fail(PSEG_ENOMEM, "hipMalloc(vote workspace) failed");
        rc = ccl_run<0>(d_bin, nullptr, d_L, H, W, st, d_hist, ncls);     // the compress pass clears the roots' counters
        if (rc == PSEG_OK) {
            vote_count_kernel<LT><<<cdiv(W, VT) * cdiv(H, VT), 256, 0, st>>>(d_L, d_pred, d_hist, H, W, ncls);
            vote_apply_kernel<LT><<<cdiv(n, 256), 256, 0, st>>>(d_L, d_hist, d_pred, n, ncls);
        }
    } else {
        // per tile: V_OPEN_MAX root slots (parent + a row of counters each), the rim table, the run list, two list lengths
        const size_t slots = (size_t)tiles * V_OPEN_MAX;
        const size_t runs_b = (size_t)tiles * V_RUN_MAX * sizeof(VRun), par_b = slots * 4, hist_b = slots * ncls * 4, rim_b = round_up((size_t)tiles * V_RIM, (size_t)16);
        const size_t task_b = (size_t)tiles * V_TASK_MAX * 2;
        char* d_aux = (char*)need(2, runs_b + par_b + hist_b + rim_b + task_b + 3 * (size_t)tiles * 4);
        if (!d_aux) return fail(PSEG_ENOMEM, "hipMalloc(vote workspace) failed");
        VRun* d_runs = (VRun*)d_aux;
        int* d_par = (int*)(d_aux + runs_b);
        int* d_hist = (int*)(d_aux + runs_b + par_b);
        uint8_t* d_rim = (uint8_t*)(d_aux + runs_b + par_b + hist_b);
        unsigned short* d_tasks = (unsigned short*)(d_aux + runs_b + par_b + hist_b + rim_b);
        int* d_rootn = (int*)(d_aux + runs_b + par_b + hist_b + rim_b + task_b);
        int* d_runn = d_rootn + tiles;
        int* d_taskn = d_runn + tiles;
        const size_t lds = (size_t)ncls * V_RUN_MAX * 2 + (size_t)ncls * CT_H * 8;
        // four waves per tile: the kernel waits (loads, LDS round trips, barriers) for 70 % of its wave cycles, and a CU's 32 wave slots
        // hold seven tiles of four waves (LDS) but four of eight -- same box, configs[4]'s page: 0.112 against 0.131 ms
        vote_tile_kernel<LT, 4><<<tiles, 256, lds, st>>>(d_bin, d_pred, d_par, d_hist, d_rim, d_rootn, d_runs, d_runn, d_tasks, d_taskn, H, W, ncls);
        const int nby = (H - 1) / CT_H, nbx = (W - 1) / CT_W;
        if (nby * W + nbx * H > 0) {                      // (a one-tile page has no open component)
            vote_border_kernel<<<cdiv(tiles, 4), 256, 0, st>>>(d_rim, d_tasks, d_taskn, d_par, cdiv(W, CT_W), tiles);
            vote_merge_kernel<<<cdiv(tiles, 4), 256, 0, st>>>(d_par, d_hist, d_rootn, ncls, tiles);
            vote_apply_runs_kernel<LT><<<tiles, 256, 0, st>>>(d_par, d_hist, d_runs, d_runn, d_pred, W, ncls);
        }
    }
    if (rc == PSEG_OK && hipGetLastError() != hipSuccess) rc = fail(PSEG_EHIP, "vote kernel launch failed");
    PSEG_HIP(hipEventRecord(w.last, st));
    w.used = true;
    w.last_stream = st;
    return rc;   // asynchronous on `st`
}

// ---------------------------------------------------------------------------------------------
// bounding boxes of components (shared by bbox fill and char-height statistics)
// ---------------------------------------------------------------------------------------------
// box[r] = {min x, min y, max x, max y} for root r; initialised to {INT_MAX, INT_MAX, -1, -1}.
__global__ void bbox_init_kernel(int4* box, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) box[p] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);
}

// One wave covers 64 consecutive pixels; lanes that share the leader's root first reduce their
// extent inside the wave so that a large component costs four atomics per wave, not per pixel.
__global__ void bbox_reduce_kernel(const int* L, const int64_t* cls, int skip_class0, int4* box,
                                   int H, int W) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = H * W;
    int r = -1, x = 0, y = 0;
    if (p < n) {
        r = L[p];
        if (skip_class0 && cls && cls[p] == 0) r = -1;
        y = p / W;
        x = p - y * W;
    }
    unsigned long long todo = __ballot(r >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lr = __shfl(r, leader);
        const bool mine = (r == lr);
        int x0 = mine ? x : 0x7fffffff, y0 = mine ? y : 0x7fffffff;
        int x1 = mine ? x : -1, y1 = mine ? y : -1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            x0 = min(x0, __shfl_xor(x0, off));
            y0 = min(y0, __shfl_xor(y0, off));
            x1 = max(x1, __shfl_xor(x1, off));
            y1 = max(y1, __shfl_xor(y1, off));
        }
        if (lane == leader) {
            int* b = (int*)&box[lr];
            atomicMin(b + 0, x0);
            atomicMin(b + 1, y0);
            atomicMax(b + 2, x1);
            atomicMax(b + 3, y1);
        }
        todo &= ~__ballot(mine);
    }
}

struct Comp {
    int x0, y0, x1, y1, cls;
};

__global__ void comp_compact_kernel(const int* L, const int64_t* cls, const int4* box, int n,
                                    Comp* comps, int* count, int cap) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || L[p] != p) return;
    const int4 b = box[p];
    if (b.z < 0) return;  // skipped (class 0) component
    const int i = atomicAdd(count, 1);
    if (i < cap) comps[i] = Comp{b.x, b.y, b.z, b.w, cls ? (int)cls[p] : 1};
}

// one workgroup per component paints its box: atomicMax == "later (higher) classes overwrite"
__global__ void bbox_paint_kernel(const Comp* comps, int* out, int W) {
    const Comp c = comps[blockIdx.x];
    const int bw = c.x1 - c.x0 + 1, bh = c.y1 - c.y0 + 1;
    for (int t = threadIdx.x; t < bw * bh; t += blockDim.x) {
        const int yy = c.y0 + t / bw, xx = c.x0 + t % bw;
        atomicMax(&out[(size_t)yy * W + xx], c.cls);
    }
}

__global__ void widen_kernel(const int* in, int64_t* out, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[p];
}

static int bbox_fill_device(const int64_t* d_pred, int64_t* d_out, int H, int W, hipStream_t st) {
    if (H <= 0 || W <= 0) return PSEG_OK;
    const int n = H * W;
    const int grid = cdiv(n, 256);
    int *d_L = nullptr, *d_tmp = nullptr, *d_count = nullptr;
    int4* d_box = nullptr;
    Comp* d_comps = nullptr;
    int rc = PSEG_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_L); (void)hipFree(d_tmp); (void)hipFree(d_count); (void)hipFree(d_box);
        (void)hipFree(d_comps);
    };
    if (hipMalloc((void**)&d_L, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void**)&d_tmp, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void**)&d_count, 4) != hipSuccess ||
        hipMalloc((void**)&d_box, (size_t)n * sizeof(int4)) != hipSuccess ||
        hipMalloc((void**)&d_comps, (size_t)n * sizeof(Comp)) != hipSuccess) {
        cleanup();
        return fail(PSEG_ENOMEM, "hipMalloc failed in bbox_fill");
    }
    rc = ccl_run<1>(nullptr, d_pred, d_L, H, W, st);
    if (rc != PSEG_OK) { cleanup(); return rc; }
    (void)hipMemsetAsync(d_tmp, 0, (size_t)n * 4, st);
    (void)hipMemsetAsync(d_count, 0, 4, st);
    bbox_init_kernel<<<grid, 256, 0, st>>>(d_box, n);
    // class 0 paints zeros into a zero image: skip it (and with it the page-sized background)
    bbox_reduce_kernel<<<grid, 256, 0, st>>>(d_L, d_pred, 1, d_box, H, W);
    comp_compact_kernel<<<grid, 256, 0, st>>>(d_L, d_pred, d_box, n, d_comps, d_count, n);
    int count = 0;
    if (hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        cleanup();
        return fail(PSEG_EHIP, "bbox_fill: count read-back failed");
    }
    if (count > 0) bbox_paint_kernel<<<count, 256, 0, st>>>(d_comps, d_tmp, W);
    widen_kernel<<<grid, 256, 0, st>>>(d_tmp, d_out, n);
    if (hipGetLastError() != hipSuccess) rc = fail(PSEG_EHIP, "bbox kernel launch failed");
    cleanup();
    return rc;
}

// ---------------------------------------------------------------------------------------------
// colour / overlay masks (lib/output.py:44-60).  HBM-bound: 1 (uint8 labels; 8 for the reference's int64) + 1 bytes
// in, 12 bytes out per pixel and mask set.  A workgroup owns 1024 consecutive pixels; a thread converts four of them
// (one dword of labels, one of the binarisation) and the 3072 bytes of each mask are exchanged through LDS so that
// every store instruction writes 16 bytes per lane of one contiguous run (whole 128-byte lines).
// ---------------------------------------------------------------------------------------------
template <typename LT>
__global__ __launch_bounds__(256) void masks_kernel(const LT* pred, const uint8_t* bin, const uint8_t* lut, int n_lut,
                                                    int n, uint8_t* color, uint8_t* overlay, uint8_t* inverted,
                                                    uint8_t* fgc, int wide) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[4][768];
    __shared__ uint32_t slut[256];                     // 0x00BBGGRR per label
    {
        const int l = threadIdx.x;
        slut[l] = l < n_lut ? (uint32_t)lut[l * 3] | ((uint32_t)lut[l * 3 + 1] << 8) | ((uint32_t)lut[l * 3 + 2] << 16) : 0u;
    }
    __syncthreads();
    const int pb = blockIdx.x * 1024;                  // first pixel of this workgroup
    const int p0 = pb + threadIdx.x * 4;
    const int cnt = min(4, n - p0);                    // <= 0: nothing for this thread
    uint32_t rgb[4] = {0, 0, 0, 0};
    uint32_t wb = 0;                                   // the four binarisation bytes
    if (cnt == 4) {
        if constexpr (sizeof(LT) == 1) {
            const uint32_t w = *(const uint32_t*)(pred + p0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t l = (w >> (8 * i)) & 0xff; rgb[i] = l < (uint32_t)n_lut ? slut[l] : 0u; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int64_t l = (int64_t)pred[p0 + i]; rgb[i] = (l >= 0 && l < n_lut) ? slut[l] : 0u; }
        }
        wb = *(const uint32_t*)(bin + p0);
    } else {
        for (int i = 0; i < cnt; ++i) {
            const int64_t l = (int64_t)pred[p0 + i];
            rgb[i] = (l >= 0 && l < n_lut) ? slut[l] : 0u;
            wb |= (uint32_t)bin[p0 + i] << (8 * i);
        }
    }
    // lib/output.py:44-60 with numpy's uint8 arithmetic: fg = 1 - binary (wraps); overlay is black where fg == 0 (binary ==
    // 1), inverted_overlay where binary == 0, fg_color_mask where fg != 0 (binary != 1)
    uint32_t m[4][4];                                  // [mask][pixel]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (wb >> (8 * i)) & 0xff;
        m[0][i] = rgb[i];
        m[1][i] = b == 1 ? 0u : rgb[i];
        m[2][i] = b == 0 ? 0u : rgb[i];
        m[3][i] = b != 1 ? 0u : rgb[i];
    }
    uint8_t* const dsts[4] = {color, overlay, inverted, fgc};
    if (wide) {
        // every output is 16-byte aligned: exchange through LDS, then 16-byte stores of contiguous runs
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!dsts[k]) continue;
            stage[k][3 * threadIdx.x] = m[k][0] | (m[k][1] << 24);
            stage[k][3 * threadIdx.x + 1] = (m[k][1] >> 8) | (m[k][2] << 16);
            stage[k][3 * threadIdx.x + 2] = (m[k][2] >> 16) | (m[k][3] << 8);
        }
        __syncthreads();
        const int nbytes = min(1024, n - pb) * 3;      // bytes of this workgroup's run
        if (threadIdx.x < 192) {
            const int o = threadIdx.x * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!dsts[k]) continue;
                uint8_t* d = dsts[k] + (size_t)pb * 3 + o;
                if (o + 16 <= nbytes) *(uint4*)d = *(const uint4*)((const uint8_t*)stage[k] + o);
                else for (int i = 0; o + i < nbytes; ++i) d[i] = ((const uint8_t*)stage[k])[o + i];
            }
        }
        return;
    }
    if (cnt <= 0) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint8_t* dst = dsts[k];
        if (!dst) continue;
        for (int i = 0; i < cnt; ++i) {
            uint8_t* d = dst + (size_t)(p0 + i) * 3;
            d[0] = (uint8_t)m[k][i]; d[1] = (uint8_t)(m[k][i] >> 8); d[2] = (uint8_t)(m[k][i] >> 16);
        }
    }
}

template <typename LT>
static int masks_device(const LT* d_pred, const uint8_t* d_binary, const uint8_t* d_lut, int n_lut, int H, int W,
                        uint8_t* d_color, uint8_t* d_overlay, uint8_t* d_inverted, uint8_t* d_fg_color, hipStream_t st) {
    if (n_lut < 1 || n_lut > 256) return fail(PSEG_EINVAL, "n_lut %d out of range (1..256)", n_lut);
    const int n = H * W;
    const uintptr_t al = (uintptr_t)d_color | (uintptr_t)d_overlay | (uintptr_t)d_inverted | (uintptr_t)d_fg_color;
    const int wide = (al & 15) == 0 && (((uintptr_t)d_pred | (uintptr_t)d_binary) & 3) == 0;
    if ((((uintptr_t)d_pred | (uintptr_t)d_binary) & 3) != 0 && sizeof(LT) == 1)
        return fail(PSEG_EINVAL, "uint8 label / binary maps must be 4-byte aligned");
    masks_kernel<LT><<<cdiv(n, 1024), 256, 0, st>>>(d_pred, d_binary, d_lut, n_lut, n, d_color, d_overlay, d_inverted,
                                                   d_fg_color, wide);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// ---------------------------------------------------------------------------------------------
// Otsu histogram + binarise (lib/image_ops.py:62-65)
// ---------------------------------------------------------------------------------------------
__global__ void hist256_kernel(const uint8_t* img, int n, unsigned* hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
        atomicAdd(&h[img[p]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// fg = (pixel > t) != !inverse   (threshold -> 255, then 255 - img unless inverse)
__global__ void binarize_kernel(const uint8_t* img, int n, int t, int inverse, uint8_t* out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        const bool above = img[p] > t;
        out[p] = (above == (inverse != 0)) ? 1 : 0;
    }
}

// cv2 getThreshVal_Otsu_8u restated (OpenCV 4.5.5 modules/imgproc/src/thresh.cpp, published
// algorithm): first maximum of q1*q2*(mu1-mu2)^2, double precision.
static int otsu_from_hist(const unsigned* h, int64_t n) {
    const double scale = 1.0 / (double)n;
    double mu = 0;
    for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0;
    int max_val = 0;
    const double eps = 1.1920928955078125e-07;  // FLT_EPSILON
    for (int i = 0; i < 256; ++i) {
        const double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (std::min(q1, q2) < eps || std::max(q1, q2) > 1.0 - eps) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return max_val;
}

}  // namespace pseg

using namespace pseg;

static int set_dev(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(PSEG_EHIP, "no HIP device visible: libpseg has no CPU fallback");
    if (device < 0 || device >= n) return fail(PSEG_EINVAL, "device %d of %d", device, n);
    PSEG_HIP(hipSetDevice(device));
    return PSEG_OK;
}

extern "C" {

int pseg_cc_vote_device(int device, int64_t* d_pred, const uint8_t* d_binary, int H, int W,
                        int n_classes, void* stream) {
    if (!d_pred || !d_binary) return fail(PSEG_EINVAL, "NULL argument");
    if (n_classes < 1) return fail(PSEG_EINVAL, "n_classes must be >= 1");
    PSEG_TRY(set_dev(device));
    return cc_vote_device<int64_t>(d_pred, d_binary, H, W, n_classes, (hipStream_t)stream);
}

int pseg_cc_vote_device_u8(int device, uint8_t* d_pred, const uint8_t* d_binary, int H, int W,
                           int n_classes, void* stream) {
    if (!d_pred || !d_binary) return fail(PSEG_EINVAL, "NULL argument");
    if (n_classes < 1 || n_classes > 256) return fail(PSEG_EINVAL, "n_classes must be in 1..256");
    PSEG_TRY(set_dev(device));
    return cc_vote_device<uint8_t>(d_pred, d_binary, H, W, n_classes, (hipStream_t)stream);
}

int pseg_release_workspace(int device) {
    PSEG_TRY(set_dev(device));
    return release_workspace(device);
}

int pseg_cc_vote(int device, int64_t* pred, const uint8_t* binary, int H, int W, int n_classes) {
    if (!pred || !binary) return fail(PSEG_EINVAL, "NULL argument");
    if (H < 0 || W < 0) return fail(PSEG_EINVAL, "negative shape");
    if (H == 0 || W == 0) return PSEG_OK;
    const size_t n = (size_t)H * W;
    if (n_classes < 1) {  // derive from the data, as np.bincount does
        int64_t m = 0;
        for (size_t i = 0; i < n; ++i) m = std::max(m, pred[i]);
        n_classes = (int)m + 1;
    }
    PSEG_TRY(set_dev(device));
    int64_t* d_pred = nullptr;
    uint8_t* d_bin = nullptr;
    PSEG_HIP(hipMalloc((void**)&d_pred, n * 8));
    if (hipMalloc((void**)&d_bin, n) != hipSuccess) { (void)hipFree(d_pred); return fail(PSEG_ENOMEM, "hipMalloc failed"); }
    int rc = PSEG_OK;
    if (hipMemcpy(d_pred, pred, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_bin, binary, n, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PSEG_EHIP, "H2D copy failed");
    if (rc == PSEG_OK) rc = cc_vote_device<int64_t>(d_pred, d_bin, H, W, n_classes, nullptr);
    if (rc == PSEG_OK && hipMemcpy(pred, d_pred, n * 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PSEG_EHIP, "D2H copy failed");
    (void)hipFree(d_pred);
    (void)hipFree(d_bin);
    return rc;
}

int pseg_bbox_fill(int device, const int64_t* pred, int64_t* out, int H, int W, int n_classes) {
    (void)n_classes;
    if (!pred || !out) return fail(PSEG_EINVAL, "NULL argument");
    if (H < 0 || W < 0) return fail(PSEG_EINVAL, "negative shape");
    if (H == 0 || W == 0) return PSEG_OK;
    if ((int64_t)H * W > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "page too large");
    const size_t n = (size_t)H * W;
    PSEG_TRY(set_dev(device));
    int64_t *d_pred = nullptr, *d_out = nullptr;
    PSEG_HIP(hipMalloc((void**)&d_pred, n * 8));
    if (hipMalloc((void**)&d_out, n * 8) != hipSuccess) { (void)hipFree(d_pred); return fail(PSEG_ENOMEM, "hipMalloc failed"); }
    int rc = PSEG_OK;
    if (hipMemcpy(d_pred, pred, n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(PSEG_EHIP, "H2D copy failed");
    if (rc == PSEG_OK) rc = bbox_fill_device(d_pred, d_out, H, W, nullptr);
    if (rc == PSEG_OK && hipMemcpy(out, d_out, n * 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PSEG_EHIP, "D2H copy failed");
    (void)hipFree(d_pred);
    (void)hipFree(d_out);
    return rc;
}

__global__ void widen_u8_i64_kernel(const uint8_t* in, int64_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void narrow_i64_u8_kernel(const int64_t* in, uint8_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint8_t)in[i];
}

// add_bounding_boxes on the compact uint8 label map, device-resident (the Predictor chain): widened to the int64 form the
// labelling kernels read, painted, narrowed back.  Synchronises `stream` (the component count is read by the host).
int pseg_bbox_fill_device_u8(int device, const uint8_t* d_pred, uint8_t* d_out, int H, int W, int n_classes, void* stream) {
    (void)n_classes;
    if (!d_pred || !d_out) return fail(PSEG_EINVAL, "NULL argument");
    if (H <= 0 || W <= 0) return PSEG_OK;
    if ((int64_t)H * W > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "page too large");
    PSEG_TRY(set_dev(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)H * W;
    // the two int64 planes live in the device's bounding-box scratch (kept between calls: 200 MB at 4096x3072 were allocated and
    // freed per call); the call ends synchronised and holds the scratch's lock to its end, so one call at a time uses it
    static std::mutex mu[64];
    static int64_t* scratch[64] = {nullptr};
    static size_t scratch_bytes[64] = {0};
    std::lock_guard<std::mutex> lk(mu[device & 63]);
    if (scratch_bytes[device & 63] < n * 16) {
        if (scratch[device & 63]) (void)hipFree(scratch[device & 63]);
        scratch[device & 63] = nullptr; scratch_bytes[device & 63] = 0;
        PSEG_HIP(hipMalloc((void**)&scratch[device & 63], n * 16));
        scratch_bytes[device & 63] = n * 16;
    }
    int64_t* const d_tmp = scratch[device & 63];
    const int g = (int)std::min<size_t>((n + 255) / 256, 8192);
    widen_u8_i64_kernel<<<g, 256, 0, st>>>(d_pred, d_tmp, n);
    int rc = bbox_fill_device(d_tmp, d_tmp + n, H, W, st);
    if (rc == PSEG_OK) {
        narrow_i64_u8_kernel<<<g, 256, 0, st>>>(d_tmp + n, d_out, n);
        if (hipGetLastError() != hipSuccess) rc = fail(PSEG_EHIP, "bbox narrow launch failed");
    }
    (void)hipStreamSynchronize(st);
    return rc;
}

int pseg_masks_device(int device, const int64_t* d_pred, const uint8_t* d_binary,
                      const uint8_t* d_lut, int n_lut, int H, int W, uint8_t* d_color,
                      uint8_t* d_overlay, uint8_t* d_inverted, uint8_t* d_fg_color, void* stream) {
    if (!d_pred || !d_binary || !d_lut) return fail(PSEG_EINVAL, "NULL argument");
    if (H <= 0 || W <= 0) return PSEG_OK;
    PSEG_TRY(set_dev(device));
    return masks_device<int64_t>(d_pred, d_binary, d_lut, n_lut, H, W, d_color, d_overlay, d_inverted, d_fg_color, (hipStream_t)stream);
}

int pseg_masks_device_u8(int device, const uint8_t* d_pred, const uint8_t* d_binary,
                         const uint8_t* d_lut, int n_lut, int H, int W, uint8_t* d_color,
                         uint8_t* d_overlay, uint8_t* d_inverted, uint8_t* d_fg_color, void* stream) {
    if (!d_pred || !d_binary || !d_lut) return fail(PSEG_EINVAL, "NULL argument");
    if (H <= 0 || W <= 0) return PSEG_OK;
    PSEG_TRY(set_dev(device));
    return masks_device<uint8_t>(d_pred, d_binary, d_lut, n_lut, H, W, d_color, d_overlay, d_inverted, d_fg_color, (hipStream_t)stream);
}

int pseg_masks(int device, const int64_t* pred, const uint8_t* binary, const uint8_t* lut,
               int n_lut, int H, int W, uint8_t* color, uint8_t* overlay, uint8_t* inverted,
               uint8_t* fg_color) {
    if (!pred || !binary || !lut || n_lut < 1) return fail(PSEG_EINVAL, "bad argument");
    if (H < 0 || W < 0) return fail(PSEG_EINVAL, "negative shape");
    if (H == 0 || W == 0) return PSEG_OK;
    PSEG_TRY(set_dev(device));
    const size_t n = (size_t)H * W;
    // one slab: pred | 4 outputs (16-byte aligned each: the kernel stores dwords) | binary | lut
    uint8_t* d = nullptr;
    const size_t ostride = (n * 3 + 15) & ~(size_t)15;
    const size_t off_out = n * 8, off_bin = off_out + 4 * ostride;
    const size_t off_lut_al = (off_bin + n + 15) & ~(size_t)15;
    PSEG_HIP(hipMalloc((void**)&d, off_lut_al + (size_t)n_lut * 3));
    int rc = PSEG_OK;
    if (hipMemcpy(d, pred, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + off_bin, binary, n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + off_lut_al, lut, (size_t)n_lut * 3, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PSEG_EHIP, "H2D copy failed");
    uint8_t* o[4] = {color, overlay, inverted, fg_color};
    uint8_t* dv[4];
    for (int i = 0; i < 4; ++i) dv[i] = o[i] ? d + off_out + (size_t)i * ostride : nullptr;
    if (rc == PSEG_OK)
        rc = pseg_masks_device(device, (const int64_t*)d, d + off_bin, d + off_lut_al, n_lut, H, W,
                               dv[0], dv[1], dv[2], dv[3], nullptr);
    for (int i = 0; i < 4 && rc == PSEG_OK; ++i)
        if (o[i] && hipMemcpy(o[i], dv[i], n * 3, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PSEG_EHIP, "D2H copy failed");
    (void)hipFree(d);
    return rc;
}

int pseg_otsu_char_height(int device, const uint8_t* gray, int H, int W, int inverse, int* height,
                          int* otsu) {
    if (!gray || !height) return fail(PSEG_EINVAL, "NULL argument");
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty image");
    if ((int64_t)H * W > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "image too large");
    PSEG_TRY(set_dev(device));
    const int n = H * W;
    const int grid = cdiv(n, 256);
    uint8_t *d_img = nullptr, *d_bin = nullptr;
    unsigned* d_hist = nullptr;
    int *d_L = nullptr, *d_count = nullptr;
    int4* d_box = nullptr;
    Comp* d_comps = nullptr;
    int rc = PSEG_OK;
    auto cleanup = [&]() {
        (void)hipDeviceSynchronize();
        (void)hipFree(d_img); (void)hipFree(d_bin); (void)hipFree(d_hist); (void)hipFree(d_L);
        (void)hipFree(d_count); (void)hipFree(d_box); (void)hipFree(d_comps);
    };
    if (hipMalloc((void**)&d_img, n) != hipSuccess || hipMalloc((void**)&d_bin, n) != hipSuccess ||
        hipMalloc((void**)&d_hist, 256 * 4) != hipSuccess || hipMalloc((void**)&d_L, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void**)&d_count, 4) != hipSuccess || hipMalloc((void**)&d_box, (size_t)n * sizeof(int4)) != hipSuccess ||
        hipMalloc((void**)&d_comps, (size_t)n * sizeof(Comp)) != hipSuccess) {
        cleanup();
        return fail(PSEG_ENOMEM, "hipMalloc failed in otsu_char_height");
    }
    unsigned hist[256];
    if (hipMemcpy(d_img, gray, n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_hist, 0, 256 * 4) != hipSuccess) {
        cleanup();
        return fail(PSEG_EHIP, "H2D copy failed");
    }
    hist256_kernel<<<std::min(grid, 2048), 256>>>(d_img, n, d_hist);
    if (hipMemcpy(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost) != hipSuccess) {
        cleanup();
        return fail(PSEG_EHIP, "histogram read-back failed");
    }
    const int t = otsu_from_hist(hist, n);
    if (otsu) *otsu = t;
    binarize_kernel<<<grid, 256>>>(d_img, n, t, inverse, d_bin);
    rc = ccl_run<0>(d_bin, nullptr, d_L, H, W, nullptr);
    if (rc != PSEG_OK) { cleanup(); return rc; }
    (void)hipMemset(d_count, 0, 4);
    bbox_init_kernel<<<grid, 256>>>(d_box, n);
    bbox_reduce_kernel<<<grid, 256>>>(d_L, nullptr, 0, d_box, H, W);
    comp_compact_kernel<<<grid, 256>>>(d_L, nullptr, d_box, n, d_comps, d_count, n);
    int count = 0;
    if (hipMemcpy(&count, d_count, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        cleanup();
        return fail(PSEG_EHIP, "count read-back failed");
    }
    std::vector<Comp> comps((size_t)count);
    if (count > 0 && hipMemcpy(comps.data(), d_comps, (size_t)count * sizeof(Comp), hipMemcpyDeviceToHost) != hipSuccess) {
        cleanup();
        return fail(PSEG_EHIP, "component read-back failed");
    }
    cleanup();
    // lib/image_ops.py:70-79: glyph-shaped components, upper median of heights
    std::vector<int> hs;
    for (auto& c : comps) {
        const int w = c.x1 - c.x0 + 1, h = c.y1 - c.y0 + 1;
        const double ar = (double)w / (double)h;
        if (0.5 < ar && ar < 2 && 10 < h && h < 60 && 5 < w && w < 50) hs.push_back(h);
    }
    if (hs.empty()) { *height = -1; return PSEG_OK; }
    std::sort(hs.begin(), hs.end());
    *height = hs[hs.size() / 2];
    return PSEG_OK;
}

}  // extern "C"

// pseg_eval.hip -- evaluation reductions over label maps: the joint (ink, mask class, predicted class)
// histogram behind fgpa / fgoverlap_per_class (lib/image_ops.py:8-55) and count_matches / total_accuracy
// (lib/evaluation.py:8-32), and the component tables of ConnectedComponentEval (lib/evaluation.py:73-117):
// labels in cv2.connectedComponentsWithStats' numbering, its stats and centroids, per-component class
// histograms, and the pixel order that turns `bbox(image)[component]` into one contiguous slice.
//
// Integer, HBM-bound work: one pass over 10 (int64 + u8 + u8) ... 17 B/px per table; counters are kept per
// workgroup in LDS or merged per wave before they reach a global atomic.
#include <cmath>
#include <cstring>
#include <vector>

#include "pseg_common.h"

#include <rocprim/rocprim.hpp>

namespace pseg {

template <typename T>
__device__ __forceinline__ int load_label(const void* a, size_t p, int ncls) {
    const long long v = (long long)((const T*)a)[p];
    return (v < 0 || v >= ncls) ? ncls : (int)v;            // one extra slot for labels outside [0, ncls)
}
__device__ __forceinline__ int load_any(const void* a, int bytes, size_t p, int ncls) {
    return bytes == 1 ? load_label<uint8_t>(a, p, ncls) : bytes == 4 ? load_label<int32_t>(a, p, ncls)
                                                                      : load_label<int64_t>(a, p, ncls);
}

// counts[b][m][p], b = (binary != 0), (ncls+1)^2 slots per plane.  LDS histogram per workgroup when it fits.
__global__ __launch_bounds__(256) void confusion_kernel(const void* pred, int pb, const void* mask, int mb,
                                                        const uint8_t* bin, size_t n, int ncls, int use_lds,
                                                        unsigned long long* counts) {
    extern __shared__ unsigned h[];
    const int K = ncls + 1, slots = 2 * K * K;
    if (use_lds) {
        for (int i = threadIdx.x; i < slots; i += blockDim.x) h[i] = 0;
        __syncthreads();
    }
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int b = bin ? (bin[t] != 0) : 1;
        const int key = (b * K + load_any(mask, mb, t, ncls)) * K + load_any(pred, pb, t, ncls);
        if (use_lds) atomicAdd(&h[key], 1u);
        else atomicAdd(&counts[key], 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < slots; i += blockDim.x)
            if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
    }
}

// ---- component numbering ----------------------------------------------------------------------------------
// cv2 numbers components in the order its scan creates their first provisional label: raster order of the
// first pixel for connectivity 4 (SAUF / Spaghetti4C), raster order of the first 2x2 block for connectivity 8
// (BBDT / Spaghetti work on 2x2 blocks).  key[root] = that position; roots sorted by key get 1, 2, ...
__global__ void comp_key_kernel(const int* L, int* key, int H, int W, int conn8) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int r = L[p];
    if (r < 0) return;
    if (!conn8) {
        if (r == p) key[p] = p;
        return;
    }
    const int y = p / W, x = p - y * W;
    atomicMin(&key[r], (y >> 1) * ((W + 1) >> 1) + (x >> 1));
}
__global__ void root_flag_kernel(const int* L, int* flag, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) flag[p] = (L[p] == p);
}
__global__ void root_compact_kernel(const int* L, const int* pos, const int* key, int* roots, int* keys, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && L[p] == p) { roots[pos[p]] = p; keys[pos[p]] = key[p]; }
}
__global__ void root_number_kernel(const int* roots_sorted, int* number, int nroots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nroots) number[roots_sorted[i]] = i + 1;
}
__global__ void relabel_kernel(const int* L, const int* number, int* out, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = L[p] >= 0 ? number[L[p]] : 0;
}
__global__ void iota_kernel(int* a, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) a[p] = p;
}

// ---- per-component tables ---------------------------------------------------------------------------------
// A wave holds 64 consecutive pixels, nearly always of one or two components: the lanes of one component are
// reduced inside the wave (butterfly over masked values) and its leader issues the atomics.
struct CompAcc { int minx, miny, maxx, maxy; unsigned long long area, sx, sy; };

__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void comp_stats_kernel(const int* lab, int H, int W, int nlab, CompAcc* acc) {
    const int n = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int l = -1, x = 0, y = 0;
    if (p < n) {
        l = lab[p];
        y = p / W;
        x = p - y * W;
        if (l < 0 || l >= nlab) l = -1;
    }
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ll = __shfl(l, leader);
        const bool in = l == ll;
        const unsigned long long same = __ballot(in);
        const int mnx = wave_min(in ? x : 0x7fffffff), mny = wave_min(in ? y : 0x7fffffff);
        const int mxx = wave_max(in ? x : -1), mxy = wave_max(in ? y : -1);
        const int sx = wave_sum(in ? x : 0), sy = wave_sum(in ? y : 0);      // 64 * 2^24 fits an int
        if (lane == leader) {
            CompAcc* a = acc + ll;
            atomicMin(&a->minx, mnx); atomicMin(&a->miny, mny);
            atomicMax(&a->maxx, mxx); atomicMax(&a->maxy, mxy);
            atomicAdd(&a->area, (unsigned long long)__popcll(same));
            atomicAdd(&a->sx, (unsigned long long)sx);
            atomicAdd(&a->sy, (unsigned long long)sy);
        }
        todo &= ~same;
    }
}
__global__ void comp_acc_init_kernel(CompAcc* acc, int nlab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nlab) acc[i] = CompAcc{0x7fffffff, 0x7fffffff, -1, -1, 0ull, 0ull, 0ull};
}

// eq[l] += (pred == mask); hp[l][pred]++, hm[l][mask]++   (K = ncls + 1 slots per component)
__global__ __launch_bounds__(256) void comp_hist_kernel(const int* lab, const void* pred, int pb, const void* mask,
                                                        int mb, int n, int nlab, int ncls, unsigned long long* eq,
                                                        unsigned long long* hp, unsigned long long* hm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int K = ncls + 1;
    int l = -1, pc = 0, mc = 0;
    if (p < n) {
        l = lab[p];
        if (l < 0 || l >= nlab) l = -1;
        pc = load_any(pred, pb, p, ncls);
        mc = load_any(mask, mb, p, ncls);
    }
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ll = __shfl(l, leader);
        const bool in = l == ll;
        const unsigned long long same = __ballot(in);
        // eq compares the raw class slots (out-of-range labels share slot ncls: callers range-check first)
        const unsigned long long e = __ballot(in && pc == mc);
        if (lane == leader && e) atomicAdd(&eq[ll], (unsigned long long)__popcll(e));
        unsigned long long tp = same;
        while (tp) {                                            // predicted classes inside this component
            const int ld = __ffsll((long long)tp) - 1;
            const int c = __shfl(pc, ld);
            const unsigned long long g = __ballot(in && pc == c);
            if (lane == ld) atomicAdd(&hp[(size_t)ll * K + c], (unsigned long long)__popcll(g));
            tp &= ~g;
        }
        unsigned long long tm = same;
        while (tm) {
            const int ld = __ffsll((long long)tm) - 1;
            const int c = __shfl(mc, ld);
            const unsigned long long g = __ballot(in && mc == c);
            if (lane == ld) atomicAdd(&hm[(size_t)ll * K + c], (unsigned long long)__popcll(g));
            tm &= ~g;
        }
        todo &= ~same;
    }
}

static int set_dev_eval(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(PSEG_EHIP, "no HIP device visible: libpseg has no CPU fallback");
    if (device < 0 || device >= n) return fail(PSEG_EINVAL, "device %d of %d", device, n);
    PSEG_HIP(hipSetDevice(device));
    return PSEG_OK;
}

struct DevBuf {                    // scope-bound device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) { p = nullptr; return fail(PSEG_ENOMEM, "hipMalloc(%zu) failed", bytes); }
        return PSEG_OK;
    }
    template <typename T> T* as() { return (T*)p; }
};

static bool ok_bytes(int b) { return b == 1 || b == 4 || b == 8; }

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_eval_confusion(int device, const void* pred, int pred_bytes, const void* mask, int mask_bytes,
                        const uint8_t* binary, int64_t n, int n_classes, int64_t* counts) {
    if (!pred || !mask || !counts) return fail(PSEG_EINVAL, "NULL argument");
    if (!ok_bytes(pred_bytes) || !ok_bytes(mask_bytes)) return fail(PSEG_EINVAL, "label element size must be 1, 4 or 8 bytes");
    if (n_classes < 1 || n_classes > 255) return fail(PSEG_EINVAL, "n_classes %d outside 1..255", n_classes);
    if (n < 0) return fail(PSEG_EINVAL, "negative size");
    const int K = n_classes + 1;
    const size_t slots = (size_t)2 * K * K;
    std::memset(counts, 0, slots * 8);
    if (n == 0) return PSEG_OK;
    PSEG_TRY(set_dev_eval(device));
    DevBuf dp, dm, db, dc;
    PSEG_TRY(dp.alloc((size_t)n * pred_bytes));
    PSEG_TRY(dm.alloc((size_t)n * mask_bytes));
    if (binary) PSEG_TRY(db.alloc((size_t)n));
    PSEG_TRY(dc.alloc(slots * 8));
    PSEG_HIP(hipMemcpy(dp.p, pred, (size_t)n * pred_bytes, hipMemcpyHostToDevice));
    PSEG_HIP(hipMemcpy(dm.p, mask, (size_t)n * mask_bytes, hipMemcpyHostToDevice));
    if (binary) PSEG_HIP(hipMemcpy(db.p, binary, (size_t)n, hipMemcpyHostToDevice));
    PSEG_HIP(hipMemset(dc.p, 0, slots * 8));
    const int use_lds = slots * 4 <= 48 * 1024;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    confusion_kernel<<<grid, 256, use_lds ? slots * 4 : 0, 0>>>(dp.p, pred_bytes, dm.p, mask_bytes, binary ? db.as<uint8_t>() : nullptr,
                                                                (size_t)n, n_classes, use_lds, dc.as<unsigned long long>());
    PSEG_HIP(hipGetLastError());
    PSEG_HIP(hipMemcpy(counts, dc.p, slots * 8, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

int pseg_cc_label(int device, const uint8_t* binary, int H, int W, int connectivity, int32_t* labels, int32_t* num_labels) {
    if (!binary || !labels || !num_labels) return fail(PSEG_EINVAL, "NULL argument");
    if (connectivity != 4 && connectivity != 8) return fail(PSEG_EINVAL, "connectivity %d (4 or 8)", connectivity);
    if (H < 0 || W < 0) return fail(PSEG_EINVAL, "negative size");
    *num_labels = 1;
    if (H == 0 || W == 0) return PSEG_OK;
    if ((int64_t)H * W > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "page too large for 32-bit component indices");
    PSEG_TRY(set_dev_eval(device));
    const int n = H * W, grid = cdiv(n, 256);
    DevBuf dbin, dL, dkey, dflag, dpos, dtmp;
    PSEG_TRY(dbin.alloc(n));
    PSEG_TRY(dL.alloc((size_t)n * 4));
    PSEG_TRY(dkey.alloc((size_t)n * 4));
    PSEG_TRY(dflag.alloc((size_t)n * 4));
    PSEG_TRY(dpos.alloc((size_t)n * 4));
    PSEG_HIP(hipMemcpy(dbin.p, binary, n, hipMemcpyHostToDevice));
    hipStream_t st = 0;
    PSEG_TRY(ccl_roots(dbin.as<uint8_t>(), dL.as<int>(), H, W, connectivity, st));
    PSEG_HIP(hipMemsetAsync(dkey.p, 0x7f, (size_t)n * 4, st));
    comp_key_kernel<<<grid, 256, 0, st>>>(dL.as<int>(), dkey.as<int>(), H, W, connectivity == 8);
    root_flag_kernel<<<grid, 256, 0, st>>>(dL.as<int>(), dflag.as<int>(), n);
    size_t tb = 0;
    PSEG_HIP(rocprim::exclusive_scan(nullptr, tb, dflag.as<int>(), dpos.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
    PSEG_TRY(dtmp.alloc(tb));
    PSEG_HIP(rocprim::exclusive_scan(dtmp.p, tb, dflag.as<int>(), dpos.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
    int last_pos = 0, last_flag = 0;
    PSEG_HIP(hipMemcpy(&last_pos, dpos.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost));
    PSEG_HIP(hipMemcpy(&last_flag, dflag.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost));
    const int nroots = last_pos + last_flag;
    *num_labels = nroots + 1;
    // flag / key buffers are free again: reuse them for the compacted and sorted root lists (4 * nroots <= n ints each)
    DevBuf droots, dkeys, droots2, dkeys2, dtmp2;
    if (nroots > 0) {
        PSEG_TRY(droots.alloc((size_t)nroots * 4));
        PSEG_TRY(dkeys.alloc((size_t)nroots * 4));
        PSEG_TRY(droots2.alloc((size_t)nroots * 4));
        PSEG_TRY(dkeys2.alloc((size_t)nroots * 4));
        root_compact_kernel<<<grid, 256, 0, st>>>(dL.as<int>(), dpos.as<int>(), dkey.as<int>(), droots.as<int>(), dkeys.as<int>(), n);
        size_t tb2 = 0;
        PSEG_HIP(rocprim::radix_sort_pairs(nullptr, tb2, dkeys.as<int>(), dkeys2.as<int>(), droots.as<int>(), droots2.as<int>(),
                                           (size_t)nroots, 0, 32, st));
        PSEG_TRY(dtmp2.alloc(tb2));
        PSEG_HIP(rocprim::radix_sort_pairs(dtmp2.p, tb2, dkeys.as<int>(), dkeys2.as<int>(), droots.as<int>(), droots2.as<int>(),
                                           (size_t)nroots, 0, 32, st));
        root_number_kernel<<<cdiv(nroots, 256), 256, 0, st>>>(droots2.as<int>(), dpos.as<int>(), nroots);   // dpos becomes number[root]
    }
    relabel_kernel<<<grid, 256, 0, st>>>(dL.as<int>(), dpos.as<int>(), dflag.as<int>(), n);
    PSEG_HIP(hipGetLastError());
    PSEG_HIP(hipMemcpy(labels, dflag.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

int pseg_cc_tables(int device, const int32_t* labels, int H, int W, int num_labels, const void* pred, int pred_bytes,
                   const void* mask, int mask_bytes, int n_classes, int32_t* stats, double* centroids, int64_t* eq,
                   int64_t* hist_pred, int64_t* hist_mask, int32_t* order) {
    if (!labels) return fail(PSEG_EINVAL, "NULL labels");
    if (H < 0 || W < 0 || num_labels < 1) return fail(PSEG_EINVAL, "bad size");
    if ((int64_t)H * W > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "page too large for 32-bit indices");
    const bool want_hist = eq || hist_pred || hist_mask;
    if (want_hist) {
        if (!pred || !mask) return fail(PSEG_EINVAL, "histograms need pred and mask");
        if (!ok_bytes(pred_bytes) || !ok_bytes(mask_bytes)) return fail(PSEG_EINVAL, "label element size must be 1, 4 or 8 bytes");
        if (n_classes < 1 || n_classes > 255) return fail(PSEG_EINVAL, "n_classes %d outside 1..255", n_classes);
    }
    const int n = H * W, K = n_classes + 1;
    if (stats) std::memset(stats, 0, (size_t)num_labels * 5 * 4);
    if (centroids) for (int i = 0; i < 2 * num_labels; ++i) centroids[i] = std::nan("");
    if (eq) std::memset(eq, 0, (size_t)num_labels * 8);
    if (hist_pred) std::memset(hist_pred, 0, (size_t)num_labels * K * 8);
    if (hist_mask) std::memset(hist_mask, 0, (size_t)num_labels * K * 8);
    if (n == 0) return PSEG_OK;
    PSEG_TRY(set_dev_eval(device));
    const int grid = cdiv(n, 256);
    hipStream_t st = 0;
    DevBuf dlab;
    PSEG_TRY(dlab.alloc((size_t)n * 4));
    PSEG_HIP(hipMemcpy(dlab.p, labels, (size_t)n * 4, hipMemcpyHostToDevice));
    if (stats || centroids) {
        DevBuf dacc;
        PSEG_TRY(dacc.alloc((size_t)num_labels * sizeof(CompAcc)));
        comp_acc_init_kernel<<<cdiv(num_labels, 256), 256, 0, st>>>(dacc.as<CompAcc>(), num_labels);
        comp_stats_kernel<<<grid, 256, 0, st>>>(dlab.as<int>(), H, W, num_labels, dacc.as<CompAcc>());
        PSEG_HIP(hipGetLastError());
        std::vector<CompAcc> acc(num_labels);
        PSEG_HIP(hipMemcpy(acc.data(), dacc.p, (size_t)num_labels * sizeof(CompAcc), hipMemcpyDeviceToHost));
        for (int i = 0; i < num_labels; ++i) {
            const CompAcc& a = acc[i];
            if (stats && a.area) {            // cv2: CC_STAT_LEFT, TOP, WIDTH, HEIGHT, AREA
                int32_t* s = stats + (size_t)i * 5;
                s[0] = a.minx; s[1] = a.miny; s[2] = a.maxx - a.minx + 1; s[3] = a.maxy - a.miny + 1; s[4] = (int32_t)a.area;
            }
            if (centroids && a.area) {
                centroids[2 * i] = (double)a.sx / (double)a.area;
                centroids[2 * i + 1] = (double)a.sy / (double)a.area;
            }
        }
    }
    if (want_hist) {
        DevBuf dp, dm, de, dhp, dhm;
        PSEG_TRY(dp.alloc((size_t)n * pred_bytes));
        PSEG_TRY(dm.alloc((size_t)n * mask_bytes));
        PSEG_TRY(de.alloc((size_t)num_labels * 8));
        PSEG_TRY(dhp.alloc((size_t)num_labels * K * 8));
        PSEG_TRY(dhm.alloc((size_t)num_labels * K * 8));
        PSEG_HIP(hipMemcpy(dp.p, pred, (size_t)n * pred_bytes, hipMemcpyHostToDevice));
        PSEG_HIP(hipMemcpy(dm.p, mask, (size_t)n * mask_bytes, hipMemcpyHostToDevice));
        PSEG_HIP(hipMemset(de.p, 0, (size_t)num_labels * 8));
        PSEG_HIP(hipMemset(dhp.p, 0, (size_t)num_labels * K * 8));
        PSEG_HIP(hipMemset(dhm.p, 0, (size_t)num_labels * K * 8));
        comp_hist_kernel<<<grid, 256, 0, st>>>(dlab.as<int>(), dp.p, pred_bytes, dm.p, mask_bytes, n, num_labels, n_classes,
                                               de.as<unsigned long long>(), dhp.as<unsigned long long>(), dhm.as<unsigned long long>());
        PSEG_HIP(hipGetLastError());
        if (eq) PSEG_HIP(hipMemcpy(eq, de.p, (size_t)num_labels * 8, hipMemcpyDeviceToHost));
        if (hist_pred) PSEG_HIP(hipMemcpy(hist_pred, dhp.p, (size_t)num_labels * K * 8, hipMemcpyDeviceToHost));
        if (hist_mask) PSEG_HIP(hipMemcpy(hist_mask, dhm.p, (size_t)num_labels * K * 8, hipMemcpyDeviceToHost));
    }
    if (order) {                         // pixel indices sorted by (label, raster position): radix sort is stable
        DevBuf didx, dk2, dv2, dtmp;
        PSEG_TRY(didx.alloc((size_t)n * 4));
        PSEG_TRY(dk2.alloc((size_t)n * 4));
        PSEG_TRY(dv2.alloc((size_t)n * 4));
        iota_kernel<<<grid, 256, 0, st>>>(didx.as<int>(), n);
        int bits = 1;
        while (bits < 31 && (1 << bits) < num_labels) ++bits;
        size_t tb = 0;
        PSEG_HIP(rocprim::radix_sort_pairs(nullptr, tb, dlab.as<int>(), dk2.as<int>(), didx.as<int>(), dv2.as<int>(), (size_t)n, 0, bits, st));
        PSEG_TRY(dtmp.alloc(tb));
        PSEG_HIP(rocprim::radix_sort_pairs(dtmp.p, tb, dlab.as<int>(), dk2.as<int>(), didx.as<int>(), dv2.as<int>(), (size_t)n, 0, bits, st));
        PSEG_HIP(hipMemcpy(order, dv2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return PSEG_OK;
}

}  // extern "C"

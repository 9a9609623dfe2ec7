// Range-level kernels: stable radix sort of run tables (rocPRIM through hipCUB), coverage voting /
// joining of ranges (C3), exact pairwise RLE intersection sweep (M2/C1) and volume fill (R4/Z1).
#include "emp_common.h"

#include <hipcub/hipcub.hpp>

extern "C" int emp_exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *tmp, void *stream);
extern "C" int64_t emp_scan_tmp_elems(int64_t n);

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------
extern "C" int64_t emp_sort_work_bytes(int64_t n)
{
    size_t bytes = 0;
    if (n < 1) n = 1;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                           (const int32_t *)nullptr, (int32_t *)nullptr, (int)n, 0, 64,
                                           (hipStream_t)0) != hipSuccess)
        return -1;
    return (int64_t)align_up((int64_t)bytes, 256) + 256;
}

extern "C" int emp_sort_u64_i32(const uint64_t *keys_in, uint64_t *keys_out, const int32_t *vals_in,
                                int32_t *vals_out, int64_t n, int begin_bit, int end_bit, void *work,
                                int64_t work_bytes, void *stream)
{
    EMP_REQUIRE(n >= 0 && n < (1LL << 31), "sort: bad n");
    if (n == 0) return EMP_OK;
    EMP_REQUIRE(keys_in && keys_out && vals_in && vals_out && work, "sort: null pointer");
    EMP_REQUIRE(begin_bit >= 0 && end_bit <= 64 && begin_bit < end_bit, "sort: bad bit range");
    size_t need = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (int)n, begin_bit,
                                       end_bit, emp_stream(stream));
    EMP_REQUIRE((int64_t)need <= work_bytes, "sort: workspace too small (%lld < %zu)", (long long)work_bytes, need);
    size_t wb = (size_t)work_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(work, wb, keys_in, keys_out, vals_in, vals_out, (int)n,
                                                      begin_bit, end_bit, emp_stream(stream));
    if (e != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "sort: %s", hipGetErrorString(e));
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// C3 voting by coverage count.  Events: (+1 at start, -1 at end), key = group<<41 | pos<<1 | type.
struct VoteWork {
    int64_t keys_in, keys_out, vals_in, vals_out, cov, flag_s, flag_e, scan_s, scan_e, scantmp, cub, cub_bytes, total;
};
static VoteWork vote_layout(int64_t n)
{
    VoteWork L;
    int64_t m = 2 * (n > 0 ? n : 1);
    int64_t o = 0;
    L.keys_in = o; o += align_up(m * 8, 256);
    L.keys_out = o; o += align_up(m * 8, 256);
    L.vals_in = o; o += align_up(m * 4, 256);
    L.vals_out = o; o += align_up(m * 4, 256);
    L.cov = o; o += align_up((m + 1) * 4, 256);
    L.flag_s = o; o += align_up(m * 4, 256);
    L.flag_e = o; o += align_up(m * 4, 256);
    L.scan_s = o; o += align_up((m + 1) * 4, 256);
    L.scan_e = o; o += align_up((m + 1) * 4, 256);
    L.scantmp = o; o += align_up(emp_scan_tmp_elems(m) * 4, 256);
    L.cub = o;
    L.cub_bytes = emp_sort_work_bytes(m);
    o += L.cub_bytes;
    L.total = o;
    return L;
}
extern "C" int64_t emp_vote_work_bytes(int64_t n) { return vote_layout(n).total; }

#define VOTE_POS_BITS 40
#define VOTE_GRP_SHIFT (VOTE_POS_BITS + 1)

__global__ void vote_events_kernel(const int64_t *__restrict__ starts, const int64_t *__restrict__ ends,
                                   const int32_t *__restrict__ grp, int64_t n, uint64_t *__restrict__ keys,
                                   int32_t *__restrict__ vals)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t g = (uint64_t)grp[i] << VOTE_GRP_SHIFT;
        keys[2 * i] = g | ((uint64_t)starts[i] << 1);
        vals[2 * i] = 1;
        keys[2 * i + 1] = g | ((uint64_t)ends[i] << 1) | 1ULL;
        vals[2 * i + 1] = -1;
    }
}

__global__ void vote_flags_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ cov, int64_t m,
                                  int thr, int32_t *__restrict__ flag_s, int32_t *__restrict__ flag_e)
{
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < m; f += (int64_t)gridDim.x * blockDim.x) {
        uint64_t pos = keys[f] >> 1;
        int fs = 0, fe = 0;
        if (f == 0 || (keys[f - 1] >> 1) != pos) {  // first event at this (group, position)
            int64_t i = f;
            while (i + 1 < m && (keys[i + 1] >> 1) == pos) ++i;
            int before = cov[f], after = cov[i + 1];
            fs = (before < thr && after >= thr);
            fe = (before >= thr && after < thr);
        }
        flag_s[f] = fs;
        flag_e[f] = fe;
    }
}

__global__ void vote_emit_kernel(const uint64_t *__restrict__ keys, int64_t m, const int32_t *__restrict__ flag_s,
                                 const int32_t *__restrict__ flag_e, const int32_t *__restrict__ scan_s,
                                 const int32_t *__restrict__ scan_e, int64_t *__restrict__ out_ranges)
{
    const uint64_t pos_mask = (1ULL << VOTE_POS_BITS) - 1ULL;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < m; f += (int64_t)gridDim.x * blockDim.x) {
        int64_t pos = (int64_t)((keys[f] >> 1) & pos_mask);
        if (flag_s[f]) out_ranges[2 * (int64_t)scan_s[f] + 0] = pos;
        if (flag_e[f]) out_ranges[2 * (int64_t)scan_e[f] + 1] = pos;
    }
}

__global__ void vote_offsets_kernel(const uint64_t *__restrict__ keys, int64_t m, const int32_t *__restrict__ scan_s,
                                    int n_groups, int32_t *__restrict__ out_off)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    uint64_t key = (uint64_t)g << VOTE_GRP_SHIFT;
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    out_off[g] = scan_s[lo];
}

extern "C" int emp_vote_ranges(const int64_t *starts, const int64_t *ends, const int32_t *grp, int64_t n,
                               int n_groups, int vote_thr, void *work, int64_t work_bytes, int64_t *out_ranges,
                               int32_t *out_off, void *stream)
{
    EMP_REQUIRE(out_off, "vote: null out_off");
    EMP_REQUIRE(n >= 0 && n < (1LL << 29), "vote: bad n");
    EMP_REQUIRE(n_groups >= 0 && n_groups < (1 << 22), "vote: too many groups");
    EMP_REQUIRE(vote_thr >= 1, "vote: vote_thr must be >= 1");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(out_off, 0, sizeof(int32_t) * (n_groups + 1), st) != hipSuccess)
            EMP_FAIL(EMP_ELAUNCH, "vote: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(starts && ends && grp && work && out_ranges, "vote: null pointer");
    VoteWork L = vote_layout(n);
    EMP_REQUIRE(work_bytes >= L.total, "vote: workspace too small");
    char *w = reinterpret_cast<char *>(work);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(w + L.keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(w + L.keys_out);
    int32_t *vals_in = reinterpret_cast<int32_t *>(w + L.vals_in);
    int32_t *vals_out = reinterpret_cast<int32_t *>(w + L.vals_out);
    int32_t *cov = reinterpret_cast<int32_t *>(w + L.cov);
    int32_t *flag_s = reinterpret_cast<int32_t *>(w + L.flag_s);
    int32_t *flag_e = reinterpret_cast<int32_t *>(w + L.flag_e);
    int32_t *scan_s = reinterpret_cast<int32_t *>(w + L.scan_s);
    int32_t *scan_e = reinterpret_cast<int32_t *>(w + L.scan_e);
    int32_t *scantmp = reinterpret_cast<int32_t *>(w + L.scantmp);
    int64_t m = 2 * n;
    int grid = emp_grid(m, 256, 4096);
    hipLaunchKernelGGL(vote_events_kernel, dim3(grid), dim3(256), 0, st, starts, ends, grp, n, keys_in, vals_in);
    EMP_CHECK_LAUNCH("emp_vote_ranges(events)");
    int rc = emp_sort_u64_i32(keys_in, keys_out, vals_in, vals_out, m, 0, 64, w + L.cub, L.cub_bytes, stream);
    if (rc != EMP_OK) return rc;
    rc = emp_exclusive_scan_i32(vals_out, m, cov, scantmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(vote_flags_kernel, dim3(grid), dim3(256), 0, st, keys_out, cov, m, vote_thr, flag_s, flag_e);
    EMP_CHECK_LAUNCH("emp_vote_ranges(flags)");
    rc = emp_exclusive_scan_i32(flag_s, m, scan_s, scantmp, stream);
    if (rc != EMP_OK) return rc;
    rc = emp_exclusive_scan_i32(flag_e, m, scan_e, scantmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(vote_emit_kernel, dim3(grid), dim3(256), 0, st, keys_out, m, flag_s, flag_e, scan_s, scan_e,
                       out_ranges);
    hipLaunchKernelGGL(vote_offsets_kernel, dim3((unsigned)emp_cdiv(n_groups + 1, 256)), dim3(256), 0, st, keys_out,
                       m, scan_s, n_groups, out_off);
    EMP_CHECK_LAUNCH("emp_vote_ranges");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// exact pairwise sweep (array_utils.py:340-403): stable merge by start (A before B on ties); the
// run that precedes the latest source change is the "check run"; every following run that starts
// at or before its end contributes min(ends) - max(starts).
__global__ void pair_intersections_kernel(const int64_t *__restrict__ starts, const int64_t *__restrict__ lens,
                                          const int64_t *__restrict__ inst_off, const int32_t *__restrict__ pairs,
                                          int64_t n_pairs, int64_t *__restrict__ out)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs;
         p += (int64_t)gridDim.x * blockDim.x) {
        int a = pairs[2 * p], b = pairs[2 * p + 1];
        int64_t ia = inst_off[a], ea = inst_off[a + 1];
        int64_t ib = inst_off[b], eb = inst_off[b + 1];
        int64_t total = 0;
        bool have_prev = false, have_chk = false;
        int prev_src = 0;
        int64_t prev_s = 0, prev_e = 0, chk_s = 0, chk_e = 0;
        while (ia < ea || ib < eb) {
            int src;
            if (ib >= eb) src = 0;
            else if (ia >= ea) src = 1;
            else src = (starts[ia] <= starts[ib]) ? 0 : 1;
            int64_t s, e;
            if (src == 0) { s = starts[ia]; e = s + lens[ia]; ++ia; }
            else { s = starts[ib]; e = s + lens[ib]; ++ib; }
            if (have_prev) {
                if (src != prev_src) { chk_s = prev_s; chk_e = prev_e; have_chk = true; }
                if (have_chk && !(chk_e < s)) {
                    int64_t hi = chk_e < e ? chk_e : e;
                    int64_t lo = chk_s > s ? chk_s : s;
                    total += hi - lo;
                }
            }
            prev_s = s; prev_e = e; prev_src = src; have_prev = true;
        }
        out[p] = total;
    }
}

extern "C" int emp_rle_pair_intersections(const int64_t *starts, const int64_t *lens, const int64_t *inst_off,
                                          const int32_t *pairs, int64_t n_pairs, int64_t *out_inter, void *stream)
{
    EMP_REQUIRE(n_pairs >= 0, "pair_intersections: bad n_pairs");
    if (n_pairs == 0) return EMP_OK;
    EMP_REQUIRE(starts && lens && inst_off && pairs && out_inter, "pair_intersections: null pointer");
    int grid = emp_grid(n_pairs, 64, 8192);
    hipLaunchKernelGGL(pair_intersections_kernel, dim3(grid), dim3(64), 0, emp_stream(stream), starts, lens, inst_off,
                       pairs, n_pairs, out_inter);
    EMP_CHECK_LAUNCH("emp_rle_pair_intersections");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// fill: one wave per run, lanes stride along the run (coalesced 256-B stores).
// pass 1 tags voxels with the highest instance order covering them, pass 2 turns tags into ids.
#define FILL_TAG 0x80000000u

template <int PASS>
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *__restrict__ vol, int64_t n_vox,
                                                       const int64_t *__restrict__ starts,
                                                       const int64_t *__restrict__ lens,
                                                       const int32_t *__restrict__ order, int64_t n_runs,
                                                       const uint32_t *__restrict__ ids)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += n_waves) {
        int64_t s = starts[r], e = s + lens[r];
        if (s < 0) s = 0;
        if (e > n_vox) e = n_vox;
        if (ids[order[r]] == 0) continue;          // id 0 = instance removed by a filter: it paints nothing and
                                                   // does not shadow the instances below it
        uint32_t tag = FILL_TAG | (uint32_t)order[r];
        for (int64_t v = s + lane; v < e; v += 64) {
            if (PASS == 0) {
                atomicMax(&vol[v], tag);
            } else {
                uint32_t cur = vol[v];
                if (cur & FILL_TAG) vol[v] = ids[cur & ~FILL_TAG];
            }
        }
    }
}

extern "C" int emp_fill_runs_u32(uint32_t *vol, int64_t n_vox, const int64_t *starts, const int64_t *lens,
                                 const int32_t *order, int64_t n_runs, const uint32_t *ids, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && n_vox >= 0, "fill: bad sizes");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(vol && starts && lens && order && ids, "fill: null pointer");
    int grid = emp_grid(n_runs * 64, 256, 8192);
    hipStream_t st = emp_stream(stream);
    hipLaunchKernelGGL(fill_u32_kernel<0>, dim3(grid), dim3(256), 0, st, vol, n_vox, starts, lens, order, n_runs, ids);
    hipLaunchKernelGGL(fill_u32_kernel<1>, dim3(grid), dim3(256), 0, st, vol, n_vox, starts, lens, order, n_runs, ids);
    EMP_CHECK_LAUNCH("emp_fill_runs_u32");
    return EMP_OK;
}

__global__ __launch_bounds__(256) void fill_u8_kernel(uint8_t *__restrict__ vol, int64_t n_vox,
                                                      const int64_t *__restrict__ starts,
                                                      const int64_t *__restrict__ lens, int64_t n_runs, uint8_t value)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += n_waves) {
        int64_t s = starts[r], e = s + lens[r];
        if (s < 0) s = 0;
        if (e > n_vox) e = n_vox;
        for (int64_t v = s + lane; v < e; v += 64) vol[v] = value;
    }
}

extern "C" int emp_fill_runs_u8(uint8_t *vol, int64_t n_vox, const int64_t *starts, const int64_t *lens,
                                int64_t n_runs, uint8_t value, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && n_vox >= 0, "fill_u8: bad sizes");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(vol && starts && lens, "fill_u8: null pointer");
    int grid = emp_grid(n_runs * 64, 256, 8192);
    hipLaunchKernelGGL(fill_u8_kernel, dim3(grid), dim3(256), 0, emp_stream(stream), vol, n_vox, starts, lens, n_runs,
                       value);
    EMP_CHECK_LAUNCH("emp_fill_runs_u8");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// M1: all pairs of boxes with strictly positive intersection (array_utils.py:144-172).
// One thread per (a, b) pair; boxes are tiny and stay in L1/L2.  Output order is arbitrary.
__global__ __launch_bounds__(256) void box_pairs_kernel(const int32_t *__restrict__ ba, int64_t na,
                                                        const int32_t *__restrict__ bb, int64_t nb, int nd,
                                                        const int32_t *__restrict__ src_a,
                                                        const int32_t *__restrict__ src_b, int upper_only,
                                                        int32_t *__restrict__ out, int64_t cap,
                                                        int32_t *__restrict__ n_out)
{
    const int64_t nbx = (nb + blockDim.x - 1) / blockDim.x;
    const int64_t i = blockIdx.x / nbx;
    const int64_t j = (int64_t)(blockIdx.x % nbx) * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    if (upper_only && j <= i) return;
    if (src_a && src_b && src_a[i] == src_b[j]) return;
    const int32_t *a = ba + i * 2 * nd, *b = bb + j * 2 * nd;
    bool hit = true;
    for (int k = 0; k < nd; ++k) {
        int lo = max(a[k], b[k]), hi = min(a[k + nd], b[k + nd]);
        hit = hit && (hi - lo > 0);
    }
    if (hit) {
        int slot = atomicAdd(n_out, 1);
        if (slot < cap) { out[2 * (int64_t)slot] = (int32_t)i; out[2 * (int64_t)slot + 1] = (int32_t)j; }
    }
}

extern "C" int emp_box_pairs(const int32_t *boxes_a, int64_t na, const int32_t *boxes_b, int64_t nb, int ndim,
                             const int32_t *src_a, const int32_t *src_b, int upper_only, int32_t *out_pairs,
                             int64_t cap, int32_t *n_out, void *stream)
{
    EMP_REQUIRE(n_out, "box_pairs: null n_out");
    EMP_REQUIRE(ndim == 2 || ndim == 3, "box_pairs: ndim must be 2 or 3");
    EMP_REQUIRE(na >= 0 && nb >= 0 && cap >= 0, "box_pairs: bad sizes");
    hipStream_t st = emp_stream(stream);
    if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "box_pairs: memset");
    if (na == 0 || nb == 0) return EMP_OK;
    EMP_REQUIRE(boxes_a && boxes_b && (out_pairs || cap == 0), "box_pairs: null pointer");
    int64_t blocks = na * emp_cdiv(nb, 256);
    EMP_REQUIRE(blocks < (1LL << 31), "box_pairs: too many boxes (%lld x %lld)", (long long)na, (long long)nb);
    hipLaunchKernelGGL(box_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, boxes_a, na, boxes_b, nb, ndim,
                       src_a, src_b, upper_only, out_pairs, cap, n_out);
    EMP_CHECK_LAUNCH("emp_box_pairs");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// fill straight from the run table of one stack: runs of a stack are disjoint, so plain stores suffice.
// value[comp] == 0 drops the component (filtered instance).  One wave per run.
__global__ __launch_bounds__(256) void fill_table_kernel(uint32_t *__restrict__ vol, int64_t HW, int slice0,
                                                         const int32_t *__restrict__ r_start,
                                                         const int32_t *__restrict__ r_len,
                                                         const int32_t *__restrict__ r_comp,
                                                         const int32_t *__restrict__ c_slice,
                                                         const uint32_t *__restrict__ value, int64_t n_runs,
                                                         int n_slices)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += n_waves) {
        int comp = r_comp[r];
        uint32_t v = value[comp];
        int sl = c_slice[comp] - slice0;
        if (v == 0 || sl < 0 || sl >= n_slices) continue;
        uint32_t *dst = vol + (int64_t)sl * HW + r_start[r];
        for (int i = lane; i < r_len[r]; i += 64) dst[i] = v;
    }
}

extern "C" int emp_fill_table_u32(uint32_t *vol, int64_t HW, int n_slices, int slice0, const int32_t *r_start,
                                  const int32_t *r_len, const int32_t *r_comp, const int32_t *c_slice,
                                  const uint32_t *value, int64_t n_runs, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && HW > 0 && n_slices >= 0, "fill_table: bad sizes");
    if (n_runs == 0 || n_slices == 0) return EMP_OK;
    EMP_REQUIRE(vol && r_start && r_len && r_comp && c_slice && value, "fill_table: null pointer");
    int grid = emp_grid(n_runs * 64, 256, 8192);
    hipLaunchKernelGGL(fill_table_kernel, dim3(grid), dim3(256), 0, emp_stream(stream), vol, HW, slice0, r_start, r_len,
                       r_comp, c_slice, value, n_runs, n_slices);
    EMP_CHECK_LAUNCH("emp_fill_table_u32");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// T1 (yz plane): scatter the runs of a yz stack (slices along x; each slice is a (Z, Y) image) into a dense
// (Z, Y, X) label volume: voxel (z, y, t) of run (slice t, pixels p..p+len) gets value[comp].  The 3D RLE
// the reference builds by decoding every pixel, sorting and re-encoding (tracker.py:83-88,110-113) is then
// read off the volume with the row-run kernels (runs along x).
__global__ __launch_bounds__(256) void scatter_yz_kernel(uint32_t *__restrict__ vol, int Y, int X,
                                                         const int32_t *__restrict__ r_start,
                                                         const int32_t *__restrict__ r_len,
                                                         const int32_t *__restrict__ r_comp,
                                                         const int32_t *__restrict__ c_slice,
                                                         const uint32_t *__restrict__ value, int64_t n_runs)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += n_waves) {
        int comp = r_comp[r];
        uint32_t v = value[comp];
        if (v == 0) continue;
        const int64_t t = c_slice[comp];
        const int64_t p0 = r_start[r];
        for (int i = lane; i < r_len[r]; i += 64) vol[(p0 + i) * X + t] = v;   // (z*Y + y) * X + t
    }
}

extern "C" int emp_scatter_yz_u32(uint32_t *vol, int Z, int Y, int X, const int32_t *r_start, const int32_t *r_len,
                                  const int32_t *r_comp, const int32_t *c_slice, const uint32_t *value,
                                  int64_t n_runs, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && Z > 0 && Y > 0 && X > 0, "scatter_yz: bad sizes");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(vol && r_start && r_len && r_comp && c_slice && value, "scatter_yz: null pointer");
    int grid = emp_grid(n_runs * 64, 256, 8192);
    hipLaunchKernelGGL(scatter_yz_kernel, dim3(grid), dim3(256), 0, emp_stream(stream), vol, Y, X, r_start, r_len,
                       r_comp, c_slice, value, n_runs);
    EMP_CHECK_LAUNCH("emp_scatter_yz_u32");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// R3: RLE <-> index lists (array_utils.py:209-252).
// decode: run i writes starts[i] .. starts[i]+runs[i]-1 at out[off[i] ..]; off = exclusive scan of runs (caller).
// encode: a sorted index list is cut where idx[i] != idx[i-1] + 1; flags -> scan -> compaction (caller scans).
__global__ __launch_bounds__(256) void rle_decode_kernel(const int64_t *__restrict__ starts,
                                                         const int64_t *__restrict__ runs,
                                                         const int64_t *__restrict__ off, int64_t n_runs,
                                                         int64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += n_waves) {
        const int64_t s = starts[r], o = off[r], len = runs[r];
        for (int64_t i = lane; i < len; i += 64) out[o + i] = s + i;
    }
}

extern "C" int emp_rle_decode(const int64_t *starts, const int64_t *runs, const int64_t *offsets, int64_t n_runs,
                              int64_t *out_indices, void *stream)
{
    EMP_REQUIRE(n_runs >= 0, "rle_decode: bad size");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(starts && runs && offsets && out_indices, "rle_decode: null pointer");
    hipLaunchKernelGGL(rle_decode_kernel, dim3(emp_grid(n_runs * 64, 256, 8192)), dim3(256), 0, emp_stream(stream),
                       starts, runs, offsets, n_runs, out_indices);
    EMP_CHECK_LAUNCH("emp_rle_decode");
    return EMP_OK;
}

__global__ void rle_encode_flags_kernel(const int64_t *__restrict__ idx, int64_t n, int32_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flags[i] = (i == 0 || idx[i] != idx[i - 1] + 1) ? 1 : 0;
}

__global__ void rle_encode_emit_kernel(const int64_t *__restrict__ idx, int64_t n, const int32_t *__restrict__ flags,
                                       const int32_t *__restrict__ scan, int64_t *__restrict__ starts,
                                       int64_t *__restrict__ runs)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (flags[i]) {
            int64_t j = i + 1;                       // run ends at the next flag; runs are short on average
            while (j < n && !flags[j]) ++j;
            starts[scan[i]] = idx[i];
            runs[scan[i]] = j - i;
        }
    }
}

extern "C" int emp_rle_encode(const int64_t *indices, int64_t n, int32_t *work, int64_t *out_starts,
                              int64_t *out_runs, int32_t *n_runs_out, void *stream)
{
    EMP_REQUIRE(n >= 0 && n < (1LL << 31) && n_runs_out, "rle_encode: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_runs_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "rle_encode: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(indices && work && out_starts && out_runs, "rle_encode: null pointer");
    int32_t *flags = work, *scan = work + n, *tmp = work + 2 * n + 1;
    int grid = emp_grid(n, 256, 4096);
    hipLaunchKernelGGL(rle_encode_flags_kernel, dim3(grid), dim3(256), 0, st, indices, n, flags);
    int rc = emp_exclusive_scan_i32(flags, n, scan, tmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(rle_encode_emit_kernel, dim3(grid), dim3(256), 0, st, indices, n, flags, scan, out_starts,
                       out_runs);
    if (hipMemcpyAsync(n_runs_out, scan + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "rle_encode: copy");
    EMP_CHECK_LAUNCH("emp_rle_encode");
    return EMP_OK;
}

extern "C" int64_t emp_rle_encode_work_elems(int64_t n) { return 2 * n + 1 + emp_scan_tmp_elems(n > 0 ? n : 1); }

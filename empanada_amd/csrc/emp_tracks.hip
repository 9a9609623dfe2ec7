// T1 on the device: run tables of a plane's slices -> per-instance 3D run lists, sorted by (instance, start),
// without leaving HBM.  replaces InstanceTracker.update / finish (empanada/inference/tracker.py:61-123) for the
// whole-stack path; the consensus (C1/C3/C4) and the fill (Z1) consume these arrays directly.
//
// A 3D run is (key, len): key = instance << 40 | flat (z, y, x) start, len in voxels.  40 bits of start cover
// 2^40 voxels (cfg 5 has 2^32), 24 bits of instance 16.7 M instances per plane set.
#include "emp_common.h"

extern "C" int emp_exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *tmp, void *stream);
extern "C" int64_t emp_scan_tmp_elems(int64_t n);
extern "C" int64_t emp_sort_work_bytes(int64_t n);
extern "C" int emp_sort_u64_i32(const uint64_t *keys_in, uint64_t *keys_out, const int32_t *vals_in,
                                int32_t *vals_out, int64_t n, int begin_bit, int end_bit, void *work,
                                int64_t work_bytes, void *stream);

#define TRK_POS_BITS 40
#define TRK_POS_MASK ((1ULL << TRK_POS_BITS) - 1ULL)

static inline int64_t trk_align(int64_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------ xy / xz lift
// Run i continues run i-1 (same instance, same slice, flat 2D indices contiguous) exactly when rle_encode over the
// instance's pixels of that slice would not start a new run there (array_utils.py:209-235): the table is in raster
// order, so the only run that can end at start-1 is the previous one.
__device__ __forceinline__ bool trk_continues(const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                                              const int32_t *c_slice, const int32_t *comp_inst, int64_t i)
{
    if (i == 0) return false;
    int ca = r_comp[i - 1], cb = r_comp[i];
    int ia = comp_inst[ca], ib = comp_inst[cb];
    return ib >= 0 && ia == ib && c_slice[ca] == c_slice[cb] && r_start[i - 1] + r_len[i - 1] == r_start[i];
}

__global__ void trk_heads_kernel(const int32_t *__restrict__ r_start, const int32_t *__restrict__ r_len,
                                 const int32_t *__restrict__ r_comp, const int32_t *__restrict__ c_slice,
                                 const int32_t *__restrict__ comp_inst, int64_t n, int32_t *__restrict__ head)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        head[i] = (comp_inst[r_comp[i]] >= 0 && !trk_continues(r_start, r_len, r_comp, c_slice, comp_inst, i)) ? 1 : 0;
}

__global__ void trk_emit_kernel(int axis, const int32_t *__restrict__ r_start, const int32_t *__restrict__ r_len,
                                const int32_t *__restrict__ r_comp, const int32_t *__restrict__ c_slice,
                                const int32_t *__restrict__ comp_inst, int64_t n, int W, int64_t YX, int X,
                                int slice0, int64_t inst_base, int64_t origin, const int32_t *__restrict__ head,
                                const int32_t *__restrict__ pos, uint64_t *__restrict__ out_key,
                                int64_t *__restrict__ out_len)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        int comp = r_comp[i];
        int64_t len = r_len[i];
        for (int64_t j = i + 1; j < n && trk_continues(r_start, r_len, r_comp, c_slice, comp_inst, j); ++j)
            len += r_len[j];
        int64_t st = r_start[i];
        int64_t sl = (int64_t)c_slice[comp] + slice0;
        int64_t st3;
        if (axis == 0) st3 = st + sl * YX;                                   // plane (Y, X), slices along z
        else if (axis == 1) st3 = (st / W) * YX + sl * (int64_t)X + (st % W);   // plane (Z, X), slices along y: only the
                                                                             // START is mapped (tracker.py:78-82)
        else st3 = (st / W) * (int64_t)X + (st % W) + origin;                // tile (., W) inside an image (., X): 2D
                                                                             // start, length kept (tile.py:155-166)
        int64_t o = pos[i];
        out_key[o] = ((uint64_t)(inst_base + comp_inst[comp]) << TRK_POS_BITS) | (uint64_t)st3;
        out_len[o] = len;
    }
}

extern "C" int64_t emp_track_work_elems(int64_t n_runs)
{
    int64_t n = n_runs > 0 ? n_runs : 1;
    return 2 * n + 2 + emp_scan_tmp_elems(n);
}

extern "C" int emp_track_lift(int axis, const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                              const int32_t *c_slice, const int32_t *comp_inst, int64_t n_runs, int H, int W, int Y,
                              int X, int slice0, int64_t inst_base, int32_t *work, uint64_t *out_key,
                              int64_t *out_len, int32_t *n_out, void *stream)
{
    EMP_REQUIRE(axis == 0 || axis == 1, "track_lift: axis must be 0 (xy) or 1 (xz); yz goes through emp_track_lift_yz");
    EMP_REQUIRE(n_runs >= 0 && n_runs < (1LL << 31) && H > 0 && W > 0 && Y > 0 && X > 0 && slice0 >= 0 && n_out,
                "track_lift: bad sizes");
    EMP_REQUIRE(axis == 0 ? (H == Y && W == X) : (W == X), "track_lift: plane shape does not match the volume");
    hipStream_t st = emp_stream(stream);
    if (n_runs == 0) {
        if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "track_lift: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(r_start && r_len && r_comp && c_slice && comp_inst && work && out_key && out_len,
                "track_lift: null pointer");
    int32_t *head = work, *pos = work + n_runs, *tmp = work + 2 * n_runs + 2;
    int grid = emp_grid(n_runs, 256, 4096);
    hipLaunchKernelGGL(trk_heads_kernel, dim3(grid), dim3(256), 0, st, r_start, r_len, r_comp, c_slice, comp_inst,
                       n_runs, head);
    EMP_CHECK_LAUNCH("emp_track_lift(heads)");
    int rc = emp_exclusive_scan_i32(head, n_runs, pos, tmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trk_emit_kernel, dim3(grid), dim3(256), 0, st, axis, r_start, r_len, r_comp, c_slice,
                       comp_inst, n_runs, W, (int64_t)Y * X, X, slice0, inst_base, (int64_t)0, head, pos, out_key,
                       out_len);
    EMP_CHECK_LAUNCH("emp_track_lift(emit)");
    if (hipMemcpyAsync(n_out, pos + n_runs, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "track_lift: count copy");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ tile lift (C5)
// Tiler.translate_rle_seg (empanada/inference/tile.py:122-168) for a whole stack of one tile's slices: the runs of
// every object (component of the tile's run table) merged where rle_encode of its flat TILE indices would merge them
// (a run ending at the tile's last column continues at column 0 of the next row, array_utils.py:209-235), the START
// mapped into the image frame -- (start / tw + y0) * X + start % tw + x0 -- and the length kept, so a run that wrapped
// inside the tile runs on past the tile's right edge in the image, exactly as in the reference.  Positions are 2D
// (the slice is carried by the object): key = (inst_base + comp_inst[comp]) << 40 | start2d.
extern "C" int emp_tile_lift(const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                             const int32_t *c_slice, const int32_t *comp_inst, int64_t n_runs, int tw, int X, int y0,
                             int x0, int64_t inst_base, int32_t *work, uint64_t *out_key, int64_t *out_len,
                             int32_t *n_out, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && n_runs < (1LL << 31) && tw > 0 && X >= tw && y0 >= 0 && x0 >= 0 && x0 + tw <= X && n_out,
                "tile_lift: bad sizes");
    hipStream_t st = emp_stream(stream);
    if (n_runs == 0) {
        if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "tile_lift: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(r_start && r_len && r_comp && c_slice && comp_inst && work && out_key && out_len,
                "tile_lift: null pointer");
    int32_t *head = work, *pos = work + n_runs, *tmp = work + 2 * n_runs + 2;
    int grid = emp_grid(n_runs, 256, 4096);
    hipLaunchKernelGGL(trk_heads_kernel, dim3(grid), dim3(256), 0, st, r_start, r_len, r_comp, c_slice, comp_inst,
                       n_runs, head);
    EMP_CHECK_LAUNCH("emp_tile_lift(heads)");
    int rc = emp_exclusive_scan_i32(head, n_runs, pos, tmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trk_emit_kernel, dim3(grid), dim3(256), 0, st, 2, r_start, r_len, r_comp, c_slice, comp_inst,
                       n_runs, tw, (int64_t)0, X, 0, inst_base, (int64_t)y0 * X + x0, head, pos, out_key, out_len);
    EMP_CHECK_LAUNCH("emp_tile_lift(emit)");
    if (hipMemcpyAsync(n_out, pos + n_runs, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "tile_lift: count copy");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ yz lift
// The yz tracker is the run-length encoding along x of the dense labelling (tracker.py:83-88,110-113): the runs of
// the scattered (Z, Y, Xl) volume (emp_scatter_yz_u32 with value = instance + 1, then emp_runs_count / _extract)
// are already that encoding; this kernel only turns (row, x) into flat starts of the full (Z, Y, X) frame.
__global__ void trk_yz_kernel(const int32_t *__restrict__ row_offsets, const int32_t *__restrict__ r_start,
                              const int32_t *__restrict__ r_len, const uint32_t *__restrict__ r_val, int64_t n_rows,
                              int64_t n, int Xl, int X, int x0, int64_t inst_base, uint64_t *__restrict__ out_key,
                              int64_t *__restrict__ out_len)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_rows;                         // last row whose offset is <= i
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t)row_offsets[mid] <= i) lo = mid; else hi = mid;
        }
        int64_t st3 = lo * (int64_t)X + x0 + (r_start[i] % Xl);
        out_key[i] = ((uint64_t)(inst_base + (int64_t)r_val[i] - 1) << TRK_POS_BITS) | (uint64_t)st3;
        out_len[i] = r_len[i];
    }
}

extern "C" int emp_track_lift_yz(const int32_t *row_offsets, const int32_t *r_start, const int32_t *r_len,
                                 const uint32_t *r_val, int64_t n_rows, int64_t n_runs, int Xl, int X, int x0,
                                 int64_t inst_base, uint64_t *out_key, int64_t *out_len, void *stream)
{
    EMP_REQUIRE(n_runs >= 0 && n_rows > 0 && Xl > 0 && X >= Xl && x0 >= 0 && x0 + Xl <= X, "track_lift_yz: bad sizes");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(row_offsets && r_start && r_len && r_val && out_key && out_len, "track_lift_yz: null pointer");
    hipLaunchKernelGGL(trk_yz_kernel, dim3(emp_grid(n_runs, 256, 4096)), dim3(256), 0, emp_stream(stream),
                       row_offsets, r_start, r_len, r_val, n_rows, n_runs, Xl, X, x0, inst_base, out_key, out_len);
    EMP_CHECK_LAUNCH("emp_track_lift_yz");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ sort (+ merge)
struct TrkSortWork {
    int64_t keys, idx_in, idx_out, len, flag, pos, scantmp, cub, cub_bytes, total;
};
static TrkSortWork trk_sort_layout(int64_t n)
{
    TrkSortWork L;
    if (n < 1) n = 1;
    int64_t o = 0;
    L.keys = o; o += trk_align(n * 8);
    L.idx_in = o; o += trk_align(n * 4);
    L.idx_out = o; o += trk_align(n * 4);
    L.len = o; o += trk_align(n * 8);
    L.flag = o; o += trk_align(n * 4);
    L.pos = o; o += trk_align((n + 1) * 4);
    L.scantmp = o; o += trk_align(emp_scan_tmp_elems(n) * 4);
    L.cub = o;
    L.cub_bytes = emp_sort_work_bytes(n);
    o += L.cub_bytes;
    L.total = o;
    return L;
}
extern "C" int64_t emp_track_sort_work_bytes(int64_t n) { return trk_sort_layout(n).total; }

__global__ void trk_set_kernel(int32_t *p, int32_t v) { *p = v; }

__global__ void trk_iota_kernel(int32_t *__restrict__ v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] = (int32_t)i;
}

__global__ void trk_gather_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ idx,
                                  const int64_t *__restrict__ len_in, int64_t n, uint64_t *__restrict__ out_key,
                                  int64_t *__restrict__ out_st, int64_t *__restrict__ out_len)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t k = keys[i];
        if (out_key) out_key[i] = k;
        if (out_st) out_st[i] = (int64_t)(k & TRK_POS_MASK);
        out_len[i] = len_in[idx[i]];
    }
}

// runs of one instance that touch (previous end == start) become one run: np.sort + rle_encode of tracker.finish
__global__ void trk_touch_heads_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ len, int64_t n,
                                       int32_t *__restrict__ head)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i - 1] + (uint64_t)len[i - 1] != keys[i]) ? 1 : 0;   // a carry into the instance
                                                                                       // bits cannot equal keys[i]
}

__global__ void trk_touch_emit_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ len, int64_t n,
                                      const int32_t *__restrict__ head, const int32_t *__restrict__ pos,
                                      uint64_t *__restrict__ out_key, int64_t *__restrict__ out_st,
                                      int64_t *__restrict__ out_len)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        int64_t total = len[i];
        for (int64_t j = i + 1; j < n && !head[j]; ++j) total += len[j];
        int64_t o = pos[i];
        if (out_key) out_key[o] = keys[i];
        if (out_st) out_st[o] = (int64_t)(keys[i] & TRK_POS_MASK);
        out_len[o] = total;
    }
}

extern "C" int emp_track_sort(const uint64_t *key_in, const int64_t *len_in, int64_t n, int merge_touching,
                              void *work, int64_t work_bytes, uint64_t *out_key, int64_t *out_st, int64_t *out_len,
                              int32_t *n_out, void *stream)
{
    EMP_REQUIRE(n >= 0 && n < (1LL << 31) && n_out, "track_sort: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "track_sort: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(key_in && len_in && work && out_len, "track_sort: null pointer");
    TrkSortWork L = trk_sort_layout(n);
    EMP_REQUIRE(work_bytes >= L.total, "track_sort: workspace too small");
    char *w = reinterpret_cast<char *>(work);
    uint64_t *keys = reinterpret_cast<uint64_t *>(w + L.keys);
    int32_t *idx_in = reinterpret_cast<int32_t *>(w + L.idx_in);
    int32_t *idx_out = reinterpret_cast<int32_t *>(w + L.idx_out);
    int64_t *len = reinterpret_cast<int64_t *>(w + L.len);
    int32_t *flag = reinterpret_cast<int32_t *>(w + L.flag);
    int32_t *pos = reinterpret_cast<int32_t *>(w + L.pos);
    int32_t *scantmp = reinterpret_cast<int32_t *>(w + L.scantmp);
    int grid = emp_grid(n, 256, 4096);
    hipLaunchKernelGGL(trk_iota_kernel, dim3(grid), dim3(256), 0, st, idx_in, n);
    int rc = emp_sort_u64_i32(key_in, keys, idx_in, idx_out, n, 0, 64, w + L.cub, L.cub_bytes, stream);
    if (rc != EMP_OK) return rc;
    if (!merge_touching) {
        hipLaunchKernelGGL(trk_gather_kernel, dim3(grid), dim3(256), 0, st, keys, idx_out, len_in, n, out_key, out_st,
                           out_len);
        EMP_CHECK_LAUNCH("emp_track_sort(gather)");
        hipLaunchKernelGGL(trk_set_kernel, dim3(1), dim3(1), 0, st, n_out, (int32_t)n);
        EMP_CHECK_LAUNCH("emp_track_sort(count)");
        return EMP_OK;
    }
    hipLaunchKernelGGL(trk_gather_kernel, dim3(grid), dim3(256), 0, st, keys, idx_out, len_in, n,
                       (uint64_t *)nullptr, (int64_t *)nullptr, len);
    hipLaunchKernelGGL(trk_touch_heads_kernel, dim3(grid), dim3(256), 0, st, keys, len, n, flag);
    EMP_CHECK_LAUNCH("emp_track_sort(heads)");
    rc = emp_exclusive_scan_i32(flag, n, pos, scantmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trk_touch_emit_kernel, dim3(grid), dim3(256), 0, st, keys, len, n, flag, pos, out_key, out_st,
                       out_len);
    EMP_CHECK_LAUNCH("emp_track_sort(emit)");
    if (hipMemcpyAsync(n_out, pos + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "track_sort: count copy");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ small table kernels
// out_off[k] = first sorted run whose instance is >= k, k = 0 .. n_inst (CSR offsets of the instances)
__global__ void trk_offsets_kernel(const uint64_t *__restrict__ keys, int64_t n, int64_t n_inst,
                                   int64_t *__restrict__ out_off)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_inst) return;
    uint64_t key = (uint64_t)k << TRK_POS_BITS;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    out_off[k] = lo;
}

extern "C" int emp_track_offsets(const uint64_t *keys_sorted, int64_t n, int64_t n_inst, int64_t *out_off,
                                 void *stream)
{
    EMP_REQUIRE(n >= 0 && n_inst >= 0 && n_inst < (1LL << 24) && out_off, "track_offsets: bad arguments");
    EMP_REQUIRE(n == 0 || keys_sorted, "track_offsets: null keys");
    hipLaunchKernelGGL(trk_offsets_kernel, dim3((unsigned)emp_cdiv(n_inst + 1, 256)), dim3(256), 0,
                       emp_stream(stream), keys_sorted, n, n_inst, out_off);
    EMP_CHECK_LAUNCH("emp_track_offsets");
    return EMP_OK;
}

// out[i] = obj_val[instance of run i] for runs stored instance by instance (off = CSR offsets over n_obj instances)
__global__ void trk_expand_kernel(const int64_t *__restrict__ off, const int32_t *__restrict__ obj_val, int64_t n_obj,
                                  int64_t n, int32_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_obj;                          // last instance whose offset is <= i
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= i) lo = mid; else hi = mid;
        }
        out[i] = obj_val[lo];
    }
}

extern "C" int emp_track_expand(const int64_t *off, const int32_t *obj_val, int64_t n_obj, int64_t n_runs,
                                int32_t *out, void *stream)
{
    EMP_REQUIRE(n_obj > 0 && n_runs >= 0, "track_expand: bad sizes");
    if (n_runs == 0) return EMP_OK;
    EMP_REQUIRE(off && obj_val && out, "track_expand: null pointer");
    hipLaunchKernelGGL(trk_expand_kernel, dim3(emp_grid(n_runs, 256, 4096)), dim3(256), 0, emp_stream(stream), off,
                       obj_val, n_obj, n_runs, out);
    EMP_CHECK_LAUNCH("emp_track_expand");
    return EMP_OK;
}

// keep the part of every run that lies in the flat voxel interval [lo, hi) (a rank's z-slab of the output volume;
// zarr_utils.py:11-47 splits runs at chunk borders the same way)
__global__ void trk_clip_flags_kernel(const uint64_t *__restrict__ key, const int64_t *__restrict__ len, int64_t n,
                                      int64_t lo, int64_t hi, int32_t *__restrict__ keep)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = (int64_t)(key[i] & TRK_POS_MASK), e = s + len[i];
        keep[i] = ((s > lo ? s : lo) < (e < hi ? e : hi)) ? 1 : 0;
    }
}

__global__ void trk_clip_emit_kernel(const uint64_t *__restrict__ key, const int64_t *__restrict__ len, int64_t n,
                                     int64_t lo, int64_t hi, const int32_t *__restrict__ keep,
                                     const int32_t *__restrict__ pos, uint64_t *__restrict__ out_key,
                                     int64_t *__restrict__ out_len)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        int64_t s = (int64_t)(key[i] & TRK_POS_MASK), e = s + len[i];
        int64_t s2 = s > lo ? s : lo, e2 = e < hi ? e : hi;
        int64_t o = pos[i];
        out_key[o] = (key[i] & ~TRK_POS_MASK) | (uint64_t)s2;
        out_len[o] = e2 - s2;
    }
}

extern "C" int emp_track_clip(const uint64_t *key, const int64_t *len, int64_t n, int64_t lo, int64_t hi,
                              int32_t *work, uint64_t *out_key, int64_t *out_len, int32_t *n_out, void *stream)
{
    EMP_REQUIRE(n >= 0 && n < (1LL << 31) && lo >= 0 && hi >= lo && n_out, "track_clip: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "track_clip: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(key && len && work && out_key && out_len, "track_clip: null pointer");
    int32_t *keep = work, *pos = work + n, *tmp = work + 2 * n + 2;
    int grid = emp_grid(n, 256, 4096);
    hipLaunchKernelGGL(trk_clip_flags_kernel, dim3(grid), dim3(256), 0, st, key, len, n, lo, hi, keep);
    EMP_CHECK_LAUNCH("emp_track_clip(flags)");
    int rc = emp_exclusive_scan_i32(keep, n, pos, tmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trk_clip_emit_kernel, dim3(grid), dim3(256), 0, st, key, len, n, lo, hi, keep, pos, out_key,
                       out_len);
    EMP_CHECK_LAUNCH("emp_track_clip(emit)");
    if (hipMemcpyAsync(n_out, pos + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "track_clip: count copy");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------ overlap triplets
// emp_runs_overlap_next emits one (comp_a, comp_b, pixels) triplet per overlapping RUN pair (~20 per component pair at
// 1024^2); the chain wants one per COMPONENT pair.  Sort by (a, b) and sum the equal keys on the device, so that the
// host receives and walks O(#component pairs) entries instead of O(#run pairs).
__global__ void trip_keys_kernel(const int32_t *__restrict__ trip, int64_t n, uint64_t *__restrict__ keys,
                                 int32_t *__restrict__ vals)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = ((uint64_t)(uint32_t)trip[3 * i] << 32) | (uint64_t)(uint32_t)trip[3 * i + 1];
        vals[i] = trip[3 * i + 2];
    }
}

__global__ void trip_heads_kernel(const uint64_t *__restrict__ keys, int64_t n, int32_t *__restrict__ head)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i - 1] != keys[i]) ? 1 : 0;
}

__global__ void trip_emit_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, int64_t n,
                                 const int32_t *__restrict__ head, const int32_t *__restrict__ pos,
                                 int32_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        int32_t total = vals[i];
        for (int64_t j = i + 1; j < n && !head[j]; ++j) total += vals[j];
        const int64_t o = pos[i];
        out[3 * o] = (int32_t)(keys[i] >> 32);
        out[3 * o + 1] = (int32_t)(keys[i] & 0xffffffffULL);
        out[3 * o + 2] = total;
    }
}

struct TripWork {
    int64_t keys_in, keys_out, vals_in, vals_out, head, pos, scantmp, cub, cub_bytes, total;
};
static TripWork trip_layout(int64_t n)
{
    TripWork L;
    if (n < 1) n = 1;
    int64_t o = 0;
    L.keys_in = o; o += trk_align(n * 8);
    L.keys_out = o; o += trk_align(n * 8);
    L.vals_in = o; o += trk_align(n * 4);
    L.vals_out = o; o += trk_align(n * 4);
    L.head = o; o += trk_align(n * 4);
    L.pos = o; o += trk_align((n + 1) * 4);
    L.scantmp = o; o += trk_align(emp_scan_tmp_elems(n) * 4);
    L.cub = o;
    L.cub_bytes = emp_sort_work_bytes(n);
    o += L.cub_bytes;
    L.total = o;
    return L;
}
extern "C" int64_t emp_triplets_reduce_work_bytes(int64_t n) { return trip_layout(n).total; }

extern "C" int emp_triplets_reduce(const int32_t *trip, int64_t n, void *work, int64_t work_bytes, int32_t *out,
                                   int32_t *n_out, void *stream)
{
    EMP_REQUIRE(n >= 0 && n < (1LL << 31) && n_out, "triplets_reduce: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "triplets_reduce: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(trip && work && out, "triplets_reduce: null pointer");
    TripWork L = trip_layout(n);
    EMP_REQUIRE(work_bytes >= L.total, "triplets_reduce: workspace too small");
    char *w = reinterpret_cast<char *>(work);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(w + L.keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(w + L.keys_out);
    int32_t *vals_in = reinterpret_cast<int32_t *>(w + L.vals_in);
    int32_t *vals_out = reinterpret_cast<int32_t *>(w + L.vals_out);
    int32_t *head = reinterpret_cast<int32_t *>(w + L.head);
    int32_t *pos = reinterpret_cast<int32_t *>(w + L.pos);
    int32_t *scantmp = reinterpret_cast<int32_t *>(w + L.scantmp);
    int grid = emp_grid(n, 256, 4096);
    hipLaunchKernelGGL(trip_keys_kernel, dim3(grid), dim3(256), 0, st, trip, n, keys_in, vals_in);
    EMP_CHECK_LAUNCH("emp_triplets_reduce(keys)");
    int rc = emp_sort_u64_i32(keys_in, keys_out, vals_in, vals_out, n, 0, 64, w + L.cub, L.cub_bytes, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trip_heads_kernel, dim3(grid), dim3(256), 0, st, keys_out, n, head);
    rc = emp_exclusive_scan_i32(head, n, pos, scantmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(trip_emit_kernel, dim3(grid), dim3(256), 0, st, keys_out, vals_out, n, head, pos, out);
    EMP_CHECK_LAUNCH("emp_triplets_reduce(emit)");
    if (hipMemcpyAsync(n_out, pos + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        EMP_FAIL(EMP_ELAUNCH, "triplets_reduce: count copy");
    return EMP_OK;
}

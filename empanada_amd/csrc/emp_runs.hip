// Run-length side of the post-processing: row-run extraction from a label stack (R2/R3),
// 8-connected components by union-find over runs (R1), component statistics, overlaps between
// consecutive slices (M2 for the matcher), and the generic int32 exclusive scan they share.
// Everything after emp_runs_extract is O(#runs), not O(#pixels).
#include "emp_common.h"

// ------------------------------------------------------------------------------------------
// exclusive scan of int32, n+1 outputs.  3 phases: per-block sums, scan of the sums by one block,
// per-block rescan with offset.  Block = 256 threads x 8 items.
#define SC_T 256
#define SC_I 8
#define SC_B (SC_T * SC_I)

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *lds, int *total)
{
    // wave scan + cross-wave fixup (4 waves)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) lds[wv] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += lds[i];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(SC_T) void scan_block_sums(const int32_t *__restrict__ in, int64_t n,
                                                        int32_t *__restrict__ sums)
{
    __shared__ int lds[4];
    int64_t base = (int64_t)blockIdx.x * SC_B + (int64_t)threadIdx.x * SC_I;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    block_exclusive_scan_256(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SC_T) void scan_sums(int32_t *__restrict__ sums, int64_t nb,
                                                  int32_t *__restrict__ total_out)
{
    __shared__ int lds[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += SC_T) {
        int64_t i = b0 + threadIdx.x;
        int v = (i < nb) ? sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan_256(v, lds, &tot);
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SC_T) void scan_apply(const int32_t *__restrict__ in, int64_t n,
                                                   const int32_t *__restrict__ sums,
                                                   int32_t *__restrict__ out)
{
    __shared__ int lds[4];
    int64_t base = (int64_t)blockIdx.x * SC_B + (int64_t)threadIdx.x * SC_I;
    int v[SC_I];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int ex = block_exclusive_scan_256(s, lds, nullptr) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

extern "C" int64_t emp_scan_tmp_elems(int64_t n) { return emp_cdiv(n > 0 ? n : 1, SC_B) + 1; }

extern "C" int emp_exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int32_t *tmp, void *stream)
{
    EMP_REQUIRE(out && tmp && (in || n == 0), "scan: null pointer");
    EMP_REQUIRE(n >= 0 && n < (1LL << 40), "scan: bad n");
    hipStream_t st = emp_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "scan: memset");
        return EMP_OK;
    }
    int64_t nb = emp_cdiv(n, SC_B);
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(SC_T), 0, st, in, n, tmp);
    hipLaunchKernelGGL(scan_sums, dim3(1), dim3(SC_T), 0, st, tmp, nb, out + n);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nb), dim3(SC_T), 0, st, in, n, tmp, out);
    EMP_CHECK_LAUNCH("emp_exclusive_scan_i32");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// Row runs.  One wave per image row; each lane owns VEC consecutive pixels (VEC = 4: one 16-byte
// load per lane, 1 KiB per wave instruction).  A run starts where a non-zero value differs from
// its left neighbour and ends where it differs from its right neighbour; the i-th start and the
// i-th end of a row belong to the same run, so starts and ends are compacted independently.
// Algorithmic traffic: 4 B/pixel read (count) + 4 B/pixel read (extract) + 12 B/run written.
template <int VEC, bool EXTRACT>
__global__ __launch_bounds__(256) void row_runs_kernel(const uint32_t *__restrict__ pan, int64_t n_rows, int W,
                                                       int32_t *__restrict__ row_counts,
                                                       const int32_t *__restrict__ row_offsets, int H,
                                                       int32_t *__restrict__ r_start,
                                                       int32_t *__restrict__ r_len,
                                                       uint32_t *__restrict__ r_val)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;            // lanes before this one
    auto fetch = [&](const uint32_t *src, int x0, uint32_t (&v)[VEC]) {
        const int x = x0 + lane * VEC;
        if (VEC == 4) {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (x < W) q = *reinterpret_cast<const uint4 *>(src + x);   // W % 4 == 0 on this path
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            v[0] = (x < W) ? src[x] : 0u;
        }
    };
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        const uint32_t *src = pan + row * W;
        const int y = (int)(row % H);
        int n_s = 0, n_e = 0;  // wave-uniform running counts of starts / ends in this row
        int32_t obase = EXTRACT ? row_offsets[row] : 0;
        // the chunk after the current one is requested before the current one is scanned (two loads in flight per
        // lane), and the two values a wave cannot get from a neighbouring lane come from the chunks in registers:
        // left of lane 0 = the previous chunk's last pixel, right of lane 63 = the next chunk's first (no dependent
        // loads inside the loop)
        uint32_t cur[VEC], nxt[VEC];
        fetch(src, 0, cur);
        uint32_t carry = 0u;                                            // pixel left of the chunk (0 at the row start)
        for (int x0 = 0; x0 < W; x0 += 64 * VEC) {
            const int x = x0 + lane * VEC;
            const bool more = x0 + 64 * VEC < W;
            if (more) fetch(src, x0 + 64 * VEC, nxt);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) nxt[j] = 0u;
            }
            uint32_t v[VEC + 2];  // v[0] = left neighbour, v[VEC+1] = right neighbour
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j + 1] = cur[j];
            uint32_t left = __shfl_up(v[VEC], 1);
            uint32_t right = __shfl_down(v[1], 1);
            const uint32_t next_first = __shfl(nxt[0], 0);
            if (lane == 0) left = carry;
            if (lane == 63) right = next_first;                         // 0 past the row's end
            v[0] = left;
            v[VEC + 1] = right;
            bool st[VEC], en[VEC];
            unsigned long long bs[VEC], be[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                bool in = (x + j) < W;
                uint32_t c = v[j + 1];
                uint32_t nb_r = ((x + j + 1) < W) ? v[j + 2] : 0u;
                st[j] = in && c != 0 && c != v[j];
                en[j] = in && c != 0 && c != nb_r;
                bs[j] = __ballot(st[j]);
                be[j] = __ballot(en[j]);
            }
            // positions by ballot + population count: starts (ends) of the lanes before this one, all sub-positions;
            // the lane's own follow in sub-position order -- raster order, since pixel = lane * VEC + j
            int tot_s = 0, tot_e = 0, ps = 0, pe = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                tot_s += __popcll(bs[j]);
                tot_e += __popcll(be[j]);
                if (EXTRACT) {
                    ps += __popcll(bs[j] & below);
                    pe += __popcll(be[j] & below);
                }
            }
            if (EXTRACT) {
                int is = obase + n_s + ps, ie = obase + n_e + pe;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (st[j]) { r_start[is] = y * W + x + j; r_val[is] = v[j + 1]; ++is; }
                    if (en[j]) { r_len[ie] = y * W + x + j + 1; ++ie; }  // end (exclusive), fixed up below
                }
            }
            n_s += tot_s;
            n_e += tot_e;
            carry = __shfl(cur[VEC - 1], 63);
#pragma unroll
            for (int j = 0; j < VEC; ++j) cur[j] = nxt[j];
        }
        if (!EXTRACT && lane == 0) row_counts[row] = n_s;
    }
}

__global__ void runs_fix_len(const int32_t *__restrict__ r_start, int32_t *__restrict__ r_len,
                             const int32_t *__restrict__ n_runs_dev)
{
    int64_t n = *n_runs_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        r_len[i] -= r_start[i];
}

static bool vec4_ok(const uint32_t *pan, int W) { return (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(pan) & 15) == 0); }

extern "C" int emp_runs_count(const uint32_t *pan, int D, int H, int W, int32_t *row_counts, void *stream)
{
    EMP_REQUIRE(pan && row_counts, "runs_count: null pointer");
    EMP_REQUIRE(D >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1LL << 31), "runs_count: bad shape");
    int64_t n_rows = (int64_t)D * H;
    if (n_rows == 0) return EMP_OK;
    int grid = emp_grid(n_rows * 64, 256, 16384);
    hipStream_t st = emp_stream(stream);
    if (vec4_ok(pan, W))
        hipLaunchKernelGGL((row_runs_kernel<4, false>), dim3(grid), dim3(256), 0, st, pan, n_rows, W, row_counts,
                           (const int32_t *)nullptr, H, (int32_t *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL((row_runs_kernel<1, false>), dim3(grid), dim3(256), 0, st, pan, n_rows, W, row_counts,
                           (const int32_t *)nullptr, H, (int32_t *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr);
    EMP_CHECK_LAUNCH("emp_runs_count");
    return EMP_OK;
}

extern "C" int emp_runs_extract(const uint32_t *pan, int D, int H, int W, const int32_t *row_offsets,
                                int32_t *r_start, int32_t *r_len, uint32_t *r_val, void *stream)
{
    EMP_REQUIRE(pan && row_offsets && r_start && r_len && r_val, "runs_extract: null pointer");
    EMP_REQUIRE(D >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1LL << 31), "runs_extract: bad shape");
    int64_t n_rows = (int64_t)D * H;
    if (n_rows == 0) return EMP_OK;
    int grid = emp_grid(n_rows * 64, 256, 16384);
    hipStream_t st = emp_stream(stream);
    if (vec4_ok(pan, W))
        hipLaunchKernelGGL((row_runs_kernel<4, true>), dim3(grid), dim3(256), 0, st, pan, n_rows, W,
                           (int32_t *)nullptr, row_offsets, H, r_start, r_len, r_val);
    else
        hipLaunchKernelGGL((row_runs_kernel<1, true>), dim3(grid), dim3(256), 0, st, pan, n_rows, W,
                           (int32_t *)nullptr, row_offsets, H, r_start, r_len, r_val);
    EMP_CHECK_LAUNCH("emp_runs_extract");
    hipLaunchKernelGGL(runs_fix_len, dim3(1024), dim3(256), 0, st, r_start, r_len, row_offsets + n_rows);
    EMP_CHECK_LAUNCH("emp_runs_extract(fix)");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// Union-find over runs.
struct LabelWork {
    int64_t parent, row, flag, scan, scantmp, hkeys, hvals, total, hsize;
};
static inline LabelWork label_layout(int64_t n)
{
    LabelWork L;
    int64_t T = 64;
    while (T < 2 * n) T <<= 1;
    L.hsize = T;
    L.parent = 0;
    L.row = L.parent + n;
    L.flag = L.row + n;
    L.scan = L.flag + n;
    L.scantmp = L.scan + n + 1;
    L.hkeys = L.scantmp + emp_scan_tmp_elems(n);
    L.hkeys += (L.hkeys & 1);  // 8-byte alignment for the 64-bit keys
    L.hvals = L.hkeys + 2 * T;
    L.total = L.hvals + T;
    return L;
}
extern "C" int64_t emp_runs_label_work_elems(int64_t n_runs) { return label_layout(n_runs > 0 ? n_runs : 1).total; }

__device__ __forceinline__ int uf_find(const int32_t *parent, int a)
{
    int p = parent[a];
    while (p != a) { a = p; p = parent[a]; }
    return a;
}
__device__ __forceinline__ void uf_unite(int32_t *parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&parent[a], b);  // hang the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

__device__ __forceinline__ bool class_is_cc(uint32_t val, int64_t div, uint32_t cc_mask)
{
    int64_t cls = (int64_t)val / div;
    return cls < 32 && ((cc_mask >> cls) & 1u);
}

__global__ void label_init_kernel(int64_t n, int64_t n_rows, const int32_t *__restrict__ row_offsets,
                                  int32_t *__restrict__ parent, int32_t *__restrict__ row)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        parent[i] = (int32_t)i;
        // row r with row_offsets[r] <= i < row_offsets[r+1]
        int64_t lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (row_offsets[mid] <= i) lo = mid; else hi = mid;
        }
        row[i] = (int32_t)lo;
    }
}

__device__ __forceinline__ uint64_t hash64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// non-CC classes: all runs of one value in one slice form one instance; find the first such run
__global__ void label_hash_kernel(int64_t n, int H, int64_t div, uint32_t cc_mask,
                                  const uint32_t *__restrict__ r_val, const int32_t *__restrict__ row,
                                  unsigned long long *__restrict__ hkeys, int32_t *__restrict__ hvals,
                                  int64_t T, int pass, int32_t *__restrict__ parent)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = r_val[i];
        if (class_is_cc(v, div, cc_mask)) continue;
        unsigned long long key = (((unsigned long long)(row[i] / H)) << 32 | v) + 1ULL;
        int64_t slot = (int64_t)(hash64(key) & (uint64_t)(T - 1));
        while (true) {
            unsigned long long cur = hkeys[slot];
            if (cur == key) break;
            if (cur == 0ULL) {
                unsigned long long old = atomicCAS(&hkeys[slot], 0ULL, key);
                if (old == 0ULL || old == key) break;
            }
            slot = (slot + 1) & (T - 1);
        }
        if (pass == 0) atomicMin(&hvals[slot], (int32_t)i);
        else parent[i] = hvals[slot];
    }
}

__global__ void label_union_kernel(int64_t n, int H, int W, int64_t div, uint32_t cc_mask,
                                   const int32_t *__restrict__ r_start, const int32_t *__restrict__ r_len,
                                   const uint32_t *__restrict__ r_val, const int32_t *__restrict__ row,
                                   const int32_t *__restrict__ row_offsets, int32_t *__restrict__ parent)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = r_val[i];
        if (!class_is_cc(v, div, cc_mask)) continue;
        int r = row[i];
        if (r % H == 0) continue;  // first row of a slice has no upper neighbour
        int a0 = r_start[i] % W, a1 = a0 + r_len[i];
        int lo = row_offsets[r - 1], hi = row_offsets[r];
        // first run of the row above whose end reaches a0 - 1 (8-connectivity): b1 >= a0
        int l = lo, h = hi;
        while (l < h) {
            int mid = (l + h) >> 1;
            int b1 = r_start[mid] % W + r_len[mid];
            if (b1 >= a0) h = mid; else l = mid + 1;
        }
        for (int j = l; j < hi; ++j) {
            int b0 = r_start[j] % W;
            if (b0 > a1) break;
            if (r_val[j] == v) uf_unite(parent, (int)i, j);
        }
    }
}

__global__ void label_flatten_kernel(int64_t n, int32_t *__restrict__ parent, int32_t *__restrict__ flag)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int r = uf_find(parent, (int)i);
        flag[i] = (r == (int)i) ? 1 : 0;
        // roots never change here, so writing the root index is race-free with concurrent finds
        if (r != (int)i) parent[i] = r;
    }
}

__global__ void label_comp_kernel(int64_t n, int H, const int32_t *__restrict__ parent,
                                  const int32_t *__restrict__ flag, const int32_t *__restrict__ scan,
                                  const int32_t *__restrict__ row, const uint32_t *__restrict__ r_val,
                                  int32_t *__restrict__ r_comp, int32_t *__restrict__ c_slice,
                                  int64_t *__restrict__ c_label, int64_t *__restrict__ c_area,
                                  int32_t *__restrict__ c_box, int32_t *__restrict__ c_first,
                                  int32_t *__restrict__ n_comp_out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int root = parent[i];
        int comp = scan[root];
        r_comp[i] = comp;
        if (flag[i]) {
            c_first[comp] = (int32_t)i;
            c_slice[comp] = row[i] / H;
            c_label[comp] = (int64_t)r_val[i];  // final label assigned per slice below
            c_area[comp] = 0;
            c_box[4 * comp + 0] = 0x7fffffff;
            c_box[4 * comp + 1] = 0x7fffffff;
            c_box[4 * comp + 2] = 0;
            c_box[4 * comp + 3] = 0;
        }
        if (i == 0) *n_comp_out = scan[n];
    }
}

// per slice: k-th component (raster order of first pixel) of a CC class gets class*div + k
__global__ void label_rank_kernel(int D, int64_t div, uint32_t cc_mask, const int32_t *__restrict__ n_comp_dev,
                                  const int32_t *__restrict__ c_slice, int64_t *__restrict__ c_label)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    int nc = *n_comp_dev;
    int lo = 0, hi = nc;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (c_slice[mid] < d) lo = mid + 1; else hi = mid; }
    int counters[32];
    for (int c = 0; c < 32; ++c) counters[c] = 0;
    for (int c = lo; c < nc && c_slice[c] == d; ++c) {
        int64_t cls = c_label[c] / div;
        if (cls < 32 && ((cc_mask >> cls) & 1u)) c_label[c] = cls * div + (++counters[cls]);
    }
}

__global__ void label_stats_kernel(int64_t n, int H, int W, const int32_t *__restrict__ r_start,
                                   const int32_t *__restrict__ r_len, const int32_t *__restrict__ r_comp,
                                   int64_t *__restrict__ c_area, int32_t *__restrict__ c_box)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int comp = r_comp[i];
        int y = r_start[i] / W, x0 = r_start[i] % W, x1 = x0 + r_len[i];
        atomicAdd(reinterpret_cast<unsigned long long *>(&c_area[comp]), (unsigned long long)r_len[i]);
        atomicMin(&c_box[4 * comp + 0], y);
        atomicMin(&c_box[4 * comp + 1], x0);
        atomicMax(&c_box[4 * comp + 2], y + 1);
        atomicMax(&c_box[4 * comp + 3], x1);
    }
}

extern "C" int emp_runs_label(const int32_t *r_start, const int32_t *r_len, const uint32_t *r_val,
                              const int32_t *row_offsets, int64_t n_runs, int D, int H, int W,
                              int64_t label_divisor, uint32_t cc_mask, int32_t *work, int32_t *r_comp,
                              int32_t *c_slice, int64_t *c_label, int64_t *c_area, int32_t *c_box,
                              int32_t *c_first, int32_t *n_comp_out, void *stream)
{
    EMP_REQUIRE(row_offsets && work && n_comp_out, "runs_label: null pointer");
    EMP_REQUIRE(n_runs >= 0 && n_runs < (1LL << 31), "runs_label: bad n_runs");
    EMP_REQUIRE(label_divisor > 0 && D >= 0 && H > 0 && W > 0, "runs_label: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (n_runs == 0) {
        if (hipMemsetAsync(n_comp_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "runs_label: memset");
        return EMP_OK;
    }
    EMP_REQUIRE(r_start && r_len && r_val && r_comp && c_slice && c_label && c_area && c_box && c_first,
                "runs_label: null pointer");
    LabelWork L = label_layout(n_runs);
    int32_t *parent = work + L.parent, *row = work + L.row, *flag = work + L.flag, *scan = work + L.scan;
    int32_t *scantmp = work + L.scantmp, *hvals = work + L.hvals;
    unsigned long long *hkeys = reinterpret_cast<unsigned long long *>(work + L.hkeys);
    int64_t n_rows = (int64_t)D * H;
    int grid = emp_grid(n_runs, 256, 4096);
    hipLaunchKernelGGL(label_init_kernel, dim3(grid), dim3(256), 0, st, n_runs, n_rows, row_offsets, parent, row);
    bool any_plain = false;
    // classes beyond bit 31 and classes whose bit is clear are "plain" (grouped by value)
    any_plain = (cc_mask != 0xffffffffu);
    if (any_plain) {
        if (hipMemsetAsync(hkeys, 0, sizeof(unsigned long long) * L.hsize, st) != hipSuccess ||
            hipMemsetAsync(hvals, 0x7f, sizeof(int32_t) * L.hsize, st) != hipSuccess)
            EMP_FAIL(EMP_ELAUNCH, "runs_label: memset");
        hipLaunchKernelGGL(label_hash_kernel, dim3(grid), dim3(256), 0, st, n_runs, H, label_divisor, cc_mask,
                           r_val, row, hkeys, hvals, L.hsize, 0, parent);
        hipLaunchKernelGGL(label_hash_kernel, dim3(grid), dim3(256), 0, st, n_runs, H, label_divisor, cc_mask,
                           r_val, row, hkeys, hvals, L.hsize, 1, parent);
    }
    hipLaunchKernelGGL(label_union_kernel, dim3(grid), dim3(256), 0, st, n_runs, H, W, label_divisor, cc_mask,
                       r_start, r_len, r_val, row, row_offsets, parent);
    hipLaunchKernelGGL(label_flatten_kernel, dim3(grid), dim3(256), 0, st, n_runs, parent, flag);
    EMP_CHECK_LAUNCH("emp_runs_label(union)");
    int rc = emp_exclusive_scan_i32(flag, n_runs, scan, scantmp, stream);
    if (rc != EMP_OK) return rc;
    hipLaunchKernelGGL(label_comp_kernel, dim3(grid), dim3(256), 0, st, n_runs, H, parent, flag, scan, row, r_val,
                       r_comp, c_slice, c_label, c_area, c_box, c_first, n_comp_out);
    hipLaunchKernelGGL(label_rank_kernel, dim3((unsigned)emp_cdiv(D, 64)), dim3(64), 0, st, D, label_divisor,
                       cc_mask, n_comp_out, c_slice, c_label);
    hipLaunchKernelGGL(label_stats_kernel, dim3(grid), dim3(256), 0, st, n_runs, H, W, r_start, r_len, r_comp,
                       c_area, c_box);
    EMP_CHECK_LAUNCH("emp_runs_label");
    return EMP_OK;
}

// ------------------------------------------------------------------------------------------
// overlaps between the runs of slice d and slice d+1 (same row, same class)
__global__ void overlap_next_kernel(int64_t n, int D, int H, int W, int64_t div,
                                    const int32_t *__restrict__ r_start, const int32_t *__restrict__ r_len,
                                    const int32_t *__restrict__ r_comp, const uint32_t *__restrict__ r_val,
                                    const int32_t *__restrict__ row_offsets, int32_t *__restrict__ out,
                                    int64_t cap, int32_t *__restrict__ n_out)
{
    const int64_t n_rows = (int64_t)D * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // row of run i
        int64_t lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (row_offsets[mid] <= i) lo = mid; else hi = mid;
        }
        int64_t r = lo;
        if (r / H >= D - 1) continue;
        int64_t rn = r + H;  // same y, next slice
        int a0 = r_start[i] % W, a1 = a0 + r_len[i];
        int ca = r_comp[i];
        int64_t cls = (int64_t)r_val[i] / div;  // class of the ORIGINAL value (labels may overflow div)
        int l = row_offsets[rn], h = row_offsets[rn + 1];
        int jl = l, jh = h;
        while (jl < jh) {  // first run with b1 > a0
            int mid = (jl + jh) >> 1;
            int b1 = r_start[mid] % W + r_len[mid];
            if (b1 > a0) jh = mid; else jl = mid + 1;
        }
        for (int j = jl; j < h; ++j) {
            int b0 = r_start[j] % W;
            if (b0 >= a1) break;
            int b1 = b0 + r_len[j];
            int cb = r_comp[j];
            if ((int64_t)r_val[j] / div != cls) continue;
            int ov = min(a1, b1) - max(a0, b0);
            int slot = atomicAdd(n_out, 1);
            if (slot < cap) {
                out[3 * (int64_t)slot + 0] = ca;
                out[3 * (int64_t)slot + 1] = cb;
                out[3 * (int64_t)slot + 2] = ov;
            }
        }
    }
}

extern "C" int emp_runs_overlap_next(const int32_t *r_start, const int32_t *r_len, const int32_t *r_comp,
                                     const uint32_t *r_val, const int32_t *row_offsets, int64_t n_runs, int D,
                                     int H, int W, int64_t label_divisor, int32_t *out_triplets,
                                     int64_t cap_triplets, int32_t *n_out, void *stream)
{
    EMP_REQUIRE(row_offsets && n_out, "overlap_next: null pointer");
    EMP_REQUIRE(label_divisor > 0 && D >= 0 && H > 0 && W > 0 && cap_triplets >= 0 && n_runs >= 0,
                "overlap_next: bad arguments");
    hipStream_t st = emp_stream(stream);
    if (hipMemsetAsync(n_out, 0, sizeof(int32_t), st) != hipSuccess) EMP_FAIL(EMP_ELAUNCH, "overlap_next: memset");
    if (n_runs == 0 || D < 2) return EMP_OK;
    EMP_REQUIRE(r_start && r_len && r_comp && r_val && (out_triplets || cap_triplets == 0),
                "overlap_next: null pointer");
    int grid = emp_grid(n_runs, 256, 4096);
    hipLaunchKernelGGL(overlap_next_kernel, dim3(grid), dim3(256), 0, st, n_runs, D, H, W, label_divisor, r_start,
                       r_len, r_comp, r_val, row_offsets, out_triplets, cap_triplets, n_out);
    EMP_CHECK_LAUNCH("emp_runs_overlap_next");
    return EMP_OK;
}

"""ctypes binding of libemp_hip.so (the C ABI in include/emp_hip.h).

The HIP library is the product; there is NO CPU fallback here.  If the shared library is
missing (not built) or a kernel is asked to run without a GPU, this module raises.
torch is used only to own device memory and to name the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libemp_hip.so')

MAX_KS = 11
MAX_CLASSES = 16
MAX_CENTERS = 4096


class HipError(RuntimeError):
    pass


_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_U32 = _c.c_uint32

# name -> (restype, argtypes); mirrors include/emp_hip.h one to one
SIGNATURES = {
    'emp_version': (_I, []),
    'emp_last_error': (_c.c_char_p, []),
    'emp_device_count': (_I, []),
    'emp_bn_act_nhwc': (_I, [_P, _P, _P, _P, _I, _L, _I, _P, _L, _P]),
    'emp_dwconv_nhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'emp_upsample_bilinear': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    'emp_conv_bn_act_proj_nhwc': (_I, [_P, _P, _P, _P, _I] + [_I] * 10 + [_P, _I, _P, _P, _L, _P]),
    'emp_conv_k_slab': (_I, [_L, _I, _I, _I]),
    'emp_conv_k_slab_cin': (_I, [_L, _I, _I, _I, _I]),
    'emp_conv_k_slab_geom': (_I, [_L, _I, _I, _I, _I, _I, _I, _I, _I]),
    'emp_conv1x1_ws_eligible': (_I, [_L, _I, _I, _I, _I, _I, _I, _I]),
    'emp_gconv_chunk': (_I, [_I]),
    'emp_gconv3x3_bn_act_nhwc': (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    'emp_conv_bn_act_nhwc': (_I, [_P, _P, _P, _P, _P, _L, _I] + [_I] * 10 + [_P, _L, _P]),
    'emp_conv_splitk_plan': (_I, [_L, _I, _I, _I, _I]),
    'emp_conv_splitk_bn_act_nhwc': (_I, [_P, _P, _P, _P, _P, _L, _I] + [_I] * 11 + [_P, _P, _L, _P]),
    'emp_wino_input_transform': (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    'emp_gemm_nt_batched': (_I, [_P, _P, _I, _L, _I, _I, _P, _P]),
    'emp_wino4_input_transform': (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    'emp_wino4_output_transform': (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _I, _P, _L, _P]),
    'emp_wino3_input_transform': (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    'emp_wino3_output_transform': (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _I, _P, _L, _P]),
    'emp_wino_gemm_fused': (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _P, _I, _P, _P]),
    'emp_wino_output_transform': (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _I, _P, _L, _P]),
    'emp_chain_class': (_I, [_L, _P, _P, _P, _I, _P, _P, _P, _P, _L, _L, _c.c_double, _c.c_double, _P, _P, _P, _P]),
    'emp_lsap_maximize': (_L, [_P, _L, _L, _P, _P]),
    'emp_slices_to_input': (_I, [_P, _L, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    'emp_pointwise_out_nhwc': (_I, [_P, _P, _P, _L, _L, _I, _I, _P, _P]),
    'emp_bn_relu_maxpool_nhwc': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'emp_stem_conv7_bn_relu_maxpool': (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    'emp_logits_to_prob': (_I, [_P, _I, _I, _L, _P, _P]),
    'emp_pr_upsample2x': (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    'emp_pr_topk_work_bytes': (_L, [_I, _L]),
    'emp_pr_topk': (_I, [_P, _I, _L, _I, _P, _L, _P, _P]),
    'emp_pr_point_sample': (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P]),
    'emp_pr_scatter': (_I, [_P, _I, _P, _I, _I, _I, _L, _P, _P]),
    'emp_median_harden_stack': (_I, [_P, _I, _I, _L, _I, _F, _P, _P, _P]),
    'emp_median_step': (_I, [_c.POINTER(_P), _I, _L, _P, _P]),
    'emp_harden': (_I, [_P, _I, _I, _L, _F, _P, _P]),
    'emp_find_centers': (_I, [_P, _I, _I, _I, _F, _I, _I, _P, _P, _P]),
    'emp_group_work_elems': (_L, [_I, _I]),
    'emp_group_pixels': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _U32, _P, _P, _P]),
    'emp_fuse_work_elems': (_L, [_I, _I, _I]),
    'emp_fuse_panoptic': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _U32, _L, _L, _L, _P, _P, _P, _P]),
    'emp_fuse_lut': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _U32, _L, _L, _P, _P]),
    'emp_fuse_apply': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _U32, _L, _L, _P, _P, _P, _P]),
    'emp_runs_count': (_I, [_P, _I, _I, _I, _P, _P]),
    'emp_scan_tmp_elems': (_L, [_L]),
    'emp_exclusive_scan_i32': (_I, [_P, _L, _P, _P, _P]),
    'emp_runs_extract': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'emp_runs_label_work_elems': (_L, [_L]),
    'emp_runs_label': (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _L, _U32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'emp_runs_overlap_next': (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _L, _P, _L, _P, _P]),
    'emp_rle_decode': (_I, [_P, _P, _P, _L, _P, _P]),
    'emp_rle_encode_work_elems': (_L, [_L]),
    'emp_rle_encode': (_I, [_P, _L, _P, _P, _P, _P, _P]),
    'emp_box_pairs': (_I, [_P, _L, _P, _L, _I, _P, _P, _I, _P, _L, _P, _P]),
    'emp_rle_pair_intersections': (_I, [_P, _P, _P, _P, _L, _P, _P]),
    'emp_sort_work_bytes': (_L, [_L]),
    'emp_sort_u64_i32': (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
    'emp_vote_work_bytes': (_L, [_L]),
    'emp_vote_ranges': (_I, [_P, _P, _P, _L, _I, _I, _P, _L, _P, _P, _P]),
    'emp_fill_runs_u32': (_I, [_P, _L, _P, _P, _P, _L, _P, _P]),
    'emp_fill_table_u32': (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    'emp_scatter_yz_u32': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    'emp_fill_runs_u8': (_I, [_P, _L, _P, _P, _L, _c.c_uint8, _P]),
    'emp_triplets_reduce_work_bytes': (_L, [_L]),
    'emp_triplets_reduce': (_I, [_P, _L, _P, _L, _P, _P, _P]),
    'emp_track_work_elems': (_L, [_L]),
    'emp_track_lift': (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P]),
    'emp_track_lift_yz': (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _I, _L, _P, _P, _P]),
    'emp_tile_lift': (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P]),
    'emp_track_sort_work_bytes': (_L, [_L]),
    'emp_track_sort': (_I, [_P, _P, _L, _I, _P, _L, _P, _P, _P, _P, _P]),
    'emp_track_offsets': (_I, [_P, _L, _L, _P, _P]),
    'emp_track_expand': (_I, [_P, _P, _L, _L, _P, _P]),
    'emp_track_clip': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """dlopen libemp_hip.so (no GPU needed for this) and bind every symbol of the ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} not found: build it with `python -m empanada_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def require_gpu():
    if torch.cuda._is_in_bad_fork():
        raise HipError("this process was forked from one that had initialised the GPU: HIP cannot be used here.  Start "
                       "the process with the 'spawn' method (multiprocessing.set_start_method('spawn')), or call "
                       "patterns.forward_matching / forward_multigpu, which re-start themselves that way")
    if not torch.cuda.is_available():
        raise HipError("empanada_amd needs an MI355X (HIP device) for this operation; no CPU fallback exists")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return t.data_ptr()


def as_u32(t):
    """reinterpret / convert an integer cuda tensor as uint32 (torch has few native uint32 ops)."""
    if t.dtype == torch.uint32:
        return t
    if t.dtype != torch.int32:
        t = t.to(torch.int32)          # keeps the low 32 bits
    return t.contiguous().view(torch.uint32)


def np_to_dev_u32(a):
    import numpy as np
    a = np.ascontiguousarray(a).astype(np.uint32)
    return torch.from_numpy(a.view(np.int32)).cuda().view(torch.uint32)


def stream():
    return torch.cuda.current_stream().cuda_stream


# optional per-call HIP-event timing (bench.py): name -> list of (start_event, end_event, algorithmic bytes or
# None, algorithmic flops or None).  Events are recorded on the stream the kernels are launched on (torch's current stream) and only read
# after the final sync.  Names in PROFILE_SKIP are not timed (bench.py samples the ~1600 dense-path calls of a
# pass in one pass only, so that event packets do not perturb the others).
PROFILE = None
PROFILE_SKIP = set()
# EMP_TRACE_CALLS=<file>: append one line per ABI call (name + integer arguments) BEFORE it is enqueued, flushed at
# once -- the tail of the file names the launches that were in flight when a queue aborts (tools/pmc_abort_probe.sh).
_TRACE = None
if os.environ.get('EMP_TRACE_CALLS'):
    _TRACE = open(os.environ['EMP_TRACE_CALLS'], 'a', buffering=1)


def trace(msg):
    """marker line in the EMP_TRACE_CALLS log (no-op without it)"""
    if _TRACE is not None:
        _TRACE.write('# ' + msg + '\n')


def call(name, *args, alg_bytes=None, alg_flops=None):
    """Call an int-returning ABI function; raise HipError with emp_last_error() on failure."""
    lib = load()
    if _TRACE is not None:
        _TRACE.write(name + ' ' + ' '.join(str(a) for a in args if isinstance(a, (int, float)) and abs(a) < (1 << 40))
                     + '\n')
    if PROFILE is not None and name not in PROFILE_SKIP:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1, alg_bytes, alg_flops))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HipError(f"{name} failed ({rc}): {lib.emp_last_error().decode()}")


def query(name, *args):
    return getattr(load(), name)(*args)


# ----------------------------------------------------------------------------- thin typed wrappers
def median_harden_stack(prob, ks, thr, want_prob=False):
    """prob (D,C,H,W) fp32 cuda -> sem (D,H,W) u8 [, filtered prob (D,C,H,W)]."""
    require_gpu()
    D, C, H, W = prob.shape
    prob = prob.contiguous()
    sem = torch.empty((D, H, W), dtype=torch.uint8, device=prob.device)
    outp = torch.empty_like(prob) if want_prob else None
    call('emp_median_harden_stack', _ptr(prob), D, C, H * W, int(ks), float(thr), _ptr(sem), _ptr(outp), stream())
    return (sem, outp) if want_prob else sem


def _expect(what, t, dtype=None, shape=None, numel=None):
    """operand check before a launch: the kernels index by the sizes they are TOLD -- a tensor of another dtype,
    shape or device would be read out of bounds on the GPU, which can take the whole node down"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipError(f"{what}: expected a tensor on the GPU")
    if dtype is not None and t.dtype not in (dtype if isinstance(dtype, tuple) else (dtype,)):
        raise HipError(f"{what}: dtype {t.dtype}, expected {dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise HipError(f"{what}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    if numel is not None and t.numel() != numel:
        raise HipError(f"{what}: {t.numel()} elements, expected {numel}")
    return t


def median_step(slices, out=None):
    """median over a list of ks same-shape fp32 cuda tensors (engines.py:59-66)."""
    require_gpu()
    ks = len(slices)
    if not 1 <= ks <= MAX_KS:
        raise HipError(f"median over {ks} slices (1..{MAX_KS} supported)")
    for i, s in enumerate(slices):
        _expect(f"median_step: slice {i}", s, torch.float32, slices[0].shape)
    slices = [s.contiguous() for s in slices]
    n = slices[0].numel()
    if out is None:
        out = torch.empty_like(slices[0])
    _expect("median_step: out", out, torch.float32, numel=n)
    arr = (_P * ks)(*[s.data_ptr() for s in slices])
    call('emp_median_step', arr, ks, n, _ptr(out), stream())
    return out


def find_centers(hmp, thr, k, cap=1024):
    """hmp (D,h,w) fp32 -> idx (D,cap) int32 raster-sorted, count (D) int32."""
    require_gpu()
    _expect("find_centers: heat map", hmp, torch.float32)
    D, h, w = hmp.shape
    hmp = hmp.contiguous()
    idx = torch.empty((D, cap), dtype=torch.int32, device=hmp.device)
    cnt = torch.empty((D,), dtype=torch.int32, device=hmp.device)
    call('emp_find_centers', _ptr(hmp), D, h, w, float(thr), int(k), int(cap), _ptr(idx), _ptr(cnt), stream())
    return idx, cnt


def group_pixels(idx, cnt, offsets, step, sem=None, thing_list=()):
    """offsets (D,2,h,w) fp32 -> ids (D,h,w) uint16.  sem (D,h,w) u8 restricts the vote to thing pixels."""
    require_gpu()
    _expect("group_pixels: offsets", offsets, torch.float32)
    D, two, h, w = offsets.shape
    if two != 2:
        raise HipError(f"group_pixels: offsets must be (D, 2, h, w), got {tuple(offsets.shape)}")
    _expect("group_pixels: centre indices", idx, torch.int32)
    if idx.dim() != 2 or idx.shape[0] != D:
        raise HipError(f"group_pixels: centre indices must be ({D}, cap), got {tuple(idx.shape)}")
    _expect("group_pixels: centre counts", cnt, torch.int32, (D,))
    idx, cnt = idx.contiguous(), cnt.contiguous()
    offsets = offsets.contiguous()
    ids = torch.empty((D, h, w), dtype=torch.uint16, device=offsets.device)
    mask = 0
    for t in thing_list:
        mask |= 1 << int(t)
    if sem is not None:
        _expect("group_pixels: class map", sem, torch.uint8, (D, h, w))
    work = torch.empty((query('emp_group_work_elems', D, idx.shape[1]),), dtype=torch.float32, device=offsets.device)
    call('emp_group_pixels', _ptr(idx), _ptr(cnt), idx.shape[1], _ptr(offsets), D, h, w, int(step),
         _ptr(sem.contiguous()) if sem is not None else None, mask, _ptr(work), _ptr(ids), stream())
    return ids


def fuse_panoptic(sem, ids, cap, n_classes, thing_list, label_divisor, stuff_area, void_label, up=1,
                  out_dtype=torch.uint32):
    """sem (D,H,W) u8, ids (D,H/up,W/up) u16 -> pan (D,H,W) uint32 or int64."""
    require_gpu()
    _expect("fuse_panoptic: class map", sem, torch.uint8)
    D, H, W = sem.shape
    up = int(up)
    if up < 1 or H % up or W % up:
        raise HipError(f"fuse_panoptic: a {H} x {W} class map is not {up} x the id map")
    _expect("fuse_panoptic: ids", ids, (torch.uint16, torch.int16), (D, H // up, W // up))
    mask = 0
    for t in thing_list:
        mask |= 1 << int(t)
    work = torch.empty((query('emp_fuse_work_elems', D, int(cap), int(n_classes)),), dtype=torch.int32,
                       device=sem.device)
    pan = torch.empty((D, H, W), dtype=out_dtype, device=sem.device)
    p32 = _ptr(pan) if out_dtype == torch.uint32 else None
    p64 = _ptr(pan) if out_dtype == torch.int64 else None
    assert (p32 is None) != (p64 is None), "out_dtype must be torch.uint32 or torch.int64"
    sem, ids = sem.contiguous(), ids.contiguous()
    call('emp_fuse_lut', _ptr(sem), _ptr(ids), D, H, W, int(up), int(cap), int(n_classes), mask, int(label_divisor),
         int(stuff_area), _ptr(work), stream())
    call('emp_fuse_apply', _ptr(sem), _ptr(ids), D, H, W, int(up), int(cap), int(n_classes), mask, int(label_divisor),
         int(void_label), _ptr(work), p32, p64, stream())
    return pan


def exclusive_scan_i32(x):
    require_gpu()
    n = x.numel()
    out = torch.empty((n + 1,), dtype=torch.int32, device=x.device)
    tmp = torch.empty((query('emp_scan_tmp_elems', n),), dtype=torch.int32, device=x.device)
    call('emp_exclusive_scan_i32', _ptr(x), n, _ptr(out), _ptr(tmp), stream())
    return out


class RunTable:
    """Result of extract_runs(): SoA run table + components of a (D,H,W) uint32 label stack."""
    __slots__ = ('D', 'H', 'W', 'n_runs', 'n_comp', 'row_offsets', 'r_start', 'r_len', 'r_val', 'r_comp',
                 'c_slice', 'c_label', 'c_area', 'c_box', 'c_first')


def extract_runs(pan, label_divisor, cc_classes):
    """pan (D,H,W) uint32 cuda -> RunTable (device tensors; two small host syncs for the counts)."""
    require_gpu()
    D, H, W = pan.shape
    pan = pan.contiguous()
    dev = pan.device
    rows = torch.empty((D * H,), dtype=torch.int32, device=dev)
    call('emp_runs_count', _ptr(pan), D, H, W, _ptr(rows), stream())
    offs = exclusive_scan_i32(rows)
    n_runs = int(offs[-1].item())
    t = RunTable()
    t.D, t.H, t.W, t.n_runs, t.row_offsets = D, H, W, n_runs, offs
    t.r_start = torch.empty((max(n_runs, 1),), dtype=torch.int32, device=dev)
    t.r_len = torch.empty_like(t.r_start)
    t.r_val = torch.empty((max(n_runs, 1),), dtype=torch.uint32, device=dev)
    call('emp_runs_extract', _ptr(pan), D, H, W, _ptr(offs), _ptr(t.r_start), _ptr(t.r_len), _ptr(t.r_val), stream())
    mask = 0
    for c in cc_classes:
        mask |= 1 << int(c)
    work = torch.empty((query('emp_runs_label_work_elems', n_runs),), dtype=torch.int32, device=dev)
    m = max(n_runs, 1)
    t.r_comp = torch.empty((m,), dtype=torch.int32, device=dev)
    t.c_slice = torch.empty((m,), dtype=torch.int32, device=dev)
    t.c_label = torch.empty((m,), dtype=torch.int64, device=dev)
    t.c_area = torch.empty((m,), dtype=torch.int64, device=dev)
    t.c_box = torch.empty((m, 4), dtype=torch.int32, device=dev)
    t.c_first = torch.empty((m,), dtype=torch.int32, device=dev)
    ncomp = torch.zeros((1,), dtype=torch.int32, device=dev)
    call('emp_runs_label', _ptr(t.r_start), _ptr(t.r_len), _ptr(t.r_val), _ptr(offs), n_runs, D, H, W,
         int(label_divisor), mask, _ptr(work), _ptr(t.r_comp), _ptr(t.c_slice), _ptr(t.c_label), _ptr(t.c_area),
         _ptr(t.c_box), _ptr(t.c_first), _ptr(ncomp), stream())
    t.n_comp = int(ncomp.item())
    for name in ('r_start', 'r_len', 'r_val', 'r_comp'):
        setattr(t, name, getattr(t, name)[:n_runs])
    for name in ('c_slice', 'c_label', 'c_area', 'c_box', 'c_first'):
        setattr(t, name, getattr(t, name)[:t.n_comp])
    return t


def overlap_next(t, label_divisor):
    """(comp_a, comp_b, overlap) int32 triplets between consecutive slices of a RunTable (device)."""
    require_gpu()
    dev = t.r_start.device
    cap = max(4 * t.n_runs, 1024)
    while True:
        out = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        n = torch.zeros((1,), dtype=torch.int32, device=dev)
        call('emp_runs_overlap_next', _ptr(t.r_start), _ptr(t.r_len), _ptr(t.r_comp), _ptr(t.r_val),
             _ptr(t.row_offsets), t.n_runs, t.D, t.H, t.W, int(label_divisor), _ptr(out), cap, _ptr(n), stream())
        cnt = int(n.item())
        if cnt <= cap:
            return reduce_triplets(out[:cnt])
        cap = cnt


def reduce_triplets(trip):
    """(n, 3) int32 (a, b, pixels) per run pair -> one row per (a, b) with the pixels summed, ascending (a, b)"""
    n = int(trip.shape[0])
    if n == 0:
        return trip
    dev = trip.device
    wb = query('emp_triplets_reduce_work_bytes', n)
    work = torch.empty((wb,), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    call('emp_triplets_reduce', _ptr(trip.contiguous()), n, _ptr(work), wb, _ptr(out), _ptr(cnt), stream())
    return out[:int(cnt.item())]


def sort_u64_i32(keys, vals, begin_bit=0, end_bit=64):
    require_gpu()
    n = keys.numel()
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    wb = query('emp_sort_work_bytes', n)
    work = torch.empty((wb,), dtype=torch.uint8, device=keys.device)
    call('emp_sort_u64_i32', _ptr(keys), _ptr(ko), _ptr(vals), _ptr(vo), n, begin_bit, end_bit, _ptr(work), wb, stream())
    return ko, vo


def vote_ranges(starts, ends, grp, n_groups, vote_thr):
    """ranges (int64 cuda) tagged with int32 group ids -> (out_ranges (m,2) int64, out_off (n_groups+1) int32)."""
    require_gpu()
    n = _expect("vote_ranges: starts", starts, torch.int64).numel()
    _expect("vote_ranges: ends", ends, torch.int64, numel=n)
    _expect("vote_ranges: groups", grp, torch.int32, numel=n)
    dev = starts.device
    wb = query('emp_vote_work_bytes', n)
    work = torch.empty((wb,), dtype=torch.uint8, device=dev)
    out = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
    off = torch.empty((n_groups + 1,), dtype=torch.int32, device=dev)
    call('emp_vote_ranges', _ptr(starts.contiguous()), _ptr(ends.contiguous()), _ptr(grp.contiguous()), n,
         int(n_groups), int(vote_thr), _ptr(work), wb, _ptr(out), _ptr(off), stream())
    return out, off


def rle_pair_intersections(starts, lens, inst_off, pairs):
    require_gpu()
    n = _expect("rle_pair_intersections: starts", starts, torch.int64).numel()
    _expect("rle_pair_intersections: lengths", lens, torch.int64, numel=n)
    _expect("rle_pair_intersections: instance offsets", inst_off, torch.int64)
    _expect("rle_pair_intersections: pairs", pairs, torch.int32)
    if pairs.dim() != 2 or pairs.shape[1] != 2:
        raise HipError(f"rle_pair_intersections: pairs must be (n, 2), got {tuple(pairs.shape)}")
    n_pairs = pairs.shape[0]
    out = torch.empty((n_pairs,), dtype=torch.int64, device=starts.device)
    call('emp_rle_pair_intersections', _ptr(starts), _ptr(lens), _ptr(inst_off), _ptr(pairs.contiguous()), n_pairs,
         _ptr(out), stream())
    return out


def fill_runs_u32(vol, starts, lens, order, ids):
    require_gpu()
    _expect("fill_runs_u32: volume", vol, (torch.uint32, torch.int32))
    n = _expect("fill_runs_u32: starts", starts, torch.int64).numel()
    _expect("fill_runs_u32: lengths", lens, torch.int64, numel=n)
    _expect("fill_runs_u32: order", order, torch.int32, numel=n)
    _expect("fill_runs_u32: ids", ids, (torch.uint32, torch.int32))
    if not (vol.is_contiguous() and starts.is_contiguous() and lens.is_contiguous() and order.is_contiguous()):
        raise HipError("fill_runs_u32: operands must be contiguous")
    call('emp_fill_runs_u32', _ptr(vol), vol.numel(), _ptr(starts), _ptr(lens), _ptr(order), starts.numel(),
         _ptr(ids), stream())
    return vol


def fill_runs_u8(vol, starts, lens, value):
    require_gpu()
    _expect("fill_runs_u8: volume", vol, torch.uint8)
    n = _expect("fill_runs_u8: starts", starts, torch.int64).numel()
    _expect("fill_runs_u8: lengths", lens, torch.int64, numel=n)
    call('emp_fill_runs_u8', _ptr(vol), vol.numel(), _ptr(starts), _ptr(lens), starts.numel(), int(value), stream())
    return vol


def box_pairs(boxes_a, boxes_b=None, src_a=None, src_b=None, upper_only=False):
    """boxes (n, 2*nd) int32 cuda -> (k, 2) int32 pairs with positive intersection (arbitrary order)."""
    require_gpu()
    self_pairs = boxes_b is None
    if self_pairs:
        boxes_b, src_b = boxes_a, src_a
    _expect("box_pairs: boxes", boxes_a, torch.int32)
    _expect("box_pairs: boxes", boxes_b, torch.int32)
    if boxes_a.dim() != 2 or boxes_b.dim() != 2 or boxes_a.shape[1] != boxes_b.shape[1] or boxes_a.shape[1] not in (4, 6):
        raise HipError(f"box_pairs: boxes must be (n, 4) or (n, 6), got {tuple(boxes_a.shape)} and {tuple(boxes_b.shape)}")
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    for what, src, n_ in (("src_a", src_a, na), ("src_b", src_b, nb)):
        if src is not None:
            _expect(f"box_pairs: {what}", src, torch.int32, numel=n_)
    nd = boxes_a.shape[1] // 2
    dev = boxes_a.device
    cap = max(16 * (na + nb), 4096)
    while True:
        out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        n = torch.zeros((1,), dtype=torch.int32, device=dev)
        call('emp_box_pairs', _ptr(boxes_a.contiguous()), na, _ptr(boxes_b.contiguous()), nb, nd, _ptr(src_a),
             _ptr(src_b), int(bool(upper_only)), _ptr(out), cap, _ptr(n), stream())
        cnt = int(n.item())
        if cnt <= cap:
            return out[:cnt]
        cap = cnt


def fill_table_u32(vol, table, value_u32, slice0=0):
    """vol (n_slices, H, W) uint32 device slab <- runs of `table` painted with value_u32[comp] (0 = skip)."""
    require_gpu()
    n_slices, H, W = vol.shape
    call('emp_fill_table_u32', _ptr(vol), H * W, n_slices, int(slice0), _ptr(table.r_start), _ptr(table.r_len),
         _ptr(table.r_comp), _ptr(table.c_slice), _ptr(value_u32), table.n_runs, stream())
    return vol


def bn_act_nhwc_(x, scale, shift, residual=None, relu=True, out=None):
    """relu?(x*scale[c] + shift[c] (+ residual)) on x (N,C,H,W fp32 in channels_last memory); in place on x, or
    written to `out`: an (N,C,H,W) channel slice of a wider channels_last buffer."""
    N, C, H, W = x.shape
    assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.float32
    _expect("bn_act: scale", scale, torch.float32, (C,))
    _expect("bn_act: shift", shift, torch.float32, (C,))
    if residual is not None:
        assert residual.shape == x.shape and residual.is_contiguous(memory_format=torch.channels_last)
        _expect("bn_act: residual", residual, torch.float32)
    ostride = 0
    dst = x
    if out is not None:
        assert out.shape == x.shape and out.dtype == torch.float32 and out.stride(1) == 1
        ostride = out.stride(3)
        assert out.stride(2) == W * ostride and out.stride(0) == H * W * ostride, "out must be an NHWC channel slice"
        dst = out
    call('emp_bn_act_nhwc', x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
         residual.data_ptr() if residual is not None else None, int(bool(relu)), N * H * W, C, dst.data_ptr(),
         ostride, stream(), alg_bytes=4 * x.numel() * (3 if residual is not None else 2))
    return dst


def yz_runs_along_x(table, value_u32, shape3d, slice0=0):
    """yz stack run table + per-component value -> 3D runs along x of the dense (Z,Y,X) labelling:
    (start3d int64, len int64, value int64) numpy arrays in raster order (emp_scatter_yz_u32 + row-run kernels).
    The table may hold only the slices [slice0, slice0 + table.D) of the x axis (slice-sharded runs): the scatter
    volume is then (Z, Y, table.D) wide and the starts are mapped into the full (Z, Y, X) frame."""
    require_gpu()
    import numpy as np
    Z, Y, X = shape3d
    Xl = table.D
    dev = table.r_start.device
    vol = torch.zeros((Z, Y, Xl), dtype=torch.int32, device=dev).view(torch.uint32)
    call('emp_scatter_yz_u32', _ptr(vol), Z, Y, Xl, _ptr(table.r_start), _ptr(table.r_len), _ptr(table.r_comp),
         _ptr(table.c_slice), _ptr(value_u32), table.n_runs, stream())
    rows = torch.empty((Z * Y,), dtype=torch.int32, device=dev)
    call('emp_runs_count', _ptr(vol), Z, Y, Xl, _ptr(rows), stream())
    offs = exclusive_scan_i32(rows)
    n = int(offs[-1].item())
    r_start = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    r_len = torch.empty_like(r_start)
    r_val = torch.empty((max(n, 1),), dtype=torch.uint32, device=dev)
    call('emp_runs_extract', _ptr(vol), Z, Y, Xl, _ptr(offs), _ptr(r_start), _ptr(r_len), _ptr(r_val), stream())
    offs_h = offs.cpu().numpy().astype(np.int64)
    st = r_start[:n].cpu().numpy().astype(np.int64)          # y * Xl + x inside plane z
    ln = r_len[:n].cpu().numpy().astype(np.int64)
    val = r_val[:n].cpu().numpy().astype(np.int64)
    row = np.searchsorted(offs_h, np.arange(n), side='right') - 1          # row = z * Y + y
    x = st % Xl
    return row * X + slice0 + x, ln, val


def rle_decode(starts, runs):
    """device int64 (starts, runs) -> int64 indices (emp_rle_decode); offsets via torch.cumsum (plumbing)."""
    require_gpu()
    n = _expect("rle_decode: starts", starts, torch.int64).numel()
    _expect("rle_decode: runs", runs, torch.int64, numel=n)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=starts.device)
    csum = torch.cumsum(runs, 0)
    off = (csum - runs).contiguous()
    out = torch.empty((int(csum[-1].item()),), dtype=torch.int64, device=starts.device)
    call('emp_rle_decode', _ptr(starts.contiguous()), _ptr(runs.contiguous()), _ptr(off), n, _ptr(out), stream())
    return out


def rle_encode(indices):
    """device int64 ascending indices -> (starts, runs) int64 (emp_rle_encode)."""
    require_gpu()
    n = _expect("rle_encode: indices", indices, torch.int64).numel()
    dev = indices.device
    work = torch.empty((query('emp_rle_encode_work_elems', n),), dtype=torch.int32, device=dev)
    st = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    rn = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    call('emp_rle_encode', _ptr(indices.contiguous()), n, _ptr(work), _ptr(st), _ptr(rn), _ptr(cnt), stream())
    k = int(cnt.item())
    return st[:k], rn[:k]


def dwconv_nhwc(x, w_kkc, bias, k):
    """x: (N, C, H, W) fp32 tensor in channels_last memory; w_kkc: (k*k, C); returns a new channels_last tensor
    (emp_dwconv_nhwc)."""
    require_gpu()
    N, C, H, W = x.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    call('emp_dwconv_nhwc', x.data_ptr(), _ptr(w_kkc), _ptr(bias), N, H, W, C, k, y.data_ptr(), stream(),
         alg_bytes=8 * x.numel())
    return y


def upsample_bilinear(x, size, out=None):
    """bilinear, align_corners=True: x (N,C,h,w) fp32 cuda, any strides -> (N,C,H,W).  `out` may be any
    (N,C,H,W) view (e.g. a channel slice of a channels_last buffer); default: same memory format as x
    (emp_upsample_bilinear)."""
    require_gpu()
    import ctypes
    N, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    assert x.is_cuda and x.dtype == torch.float32
    if out is None:
        cl = C % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last)   # few channels: planar output
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last if cl else torch.contiguous_format)
    assert out.shape == (N, C, H, W) and out.dtype == torch.float32 and out.is_cuda
    xs = (ctypes.c_int64 * 4)(*x.stride())
    ys = (ctypes.c_int64 * 4)(*out.stride())
    call('emp_upsample_bilinear', x.data_ptr(), N, C, h, w, xs, out.data_ptr(), H, W, ys, stream(),
         alg_bytes=4 * (x.numel() + N * C * H * W))
    return out


def conv_bn_act_nhwc(x, w_okkc, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, dil=1, out=None):
    """Fused convolution + per-channel affine + residual + ReLU on the fp32 matrix cores (emp_conv_bn_act_nhwc).
    x: (N,Cin,H,W) fp32 channels_last; w_okkc: (Cout,KH,KW,Cin) contiguous; residual: (N,Cout,OH,OW) channels_last
    (or a channel slice); out: optional (N,Cout,OH,OW) channel slice of a channels_last buffer.
    relu='gate': the squeeze-excite epilogue, out = residual * sigmoid(acc * scale + shift)."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout, KH, KW, _ = w_okkc.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    _expect("conv: weights", w_okkc, torch.float32)
    if w_okkc.shape[3] != Cin or not w_okkc.is_contiguous():
        raise HipError(f"conv: weights must be a contiguous (Cout, KH, KW, Cin = {Cin}) tensor, got {tuple(w_okkc.shape)}")
    for what, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _expect(f"conv: {what}", t, torch.float32, (Cout,))
    if residual is not None:
        _expect("conv: residual", residual, torch.float32)
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1

    def pixel_stride(t):
        assert t.shape == (N, Cout, OH, OW) and t.dtype == torch.float32 and t.stride(1) == 1
        ps = t.stride(3)
        assert t.stride(2) == OW * ps and t.stride(0) == OH * OW * ps, "NHWC channel slice required"
        return ps

    if out is None:
        out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    ops = pixel_stride(out)
    rps = pixel_stride(residual) if residual is not None else 0
    call('emp_conv_bn_act_nhwc', x.data_ptr(), _ptr(w_okkc), _ptr(scale), _ptr(shift),
         residual.data_ptr() if residual is not None else None, rps, 2 if relu == 'gate' else int(bool(relu)),
         N, H, W, Cin, Cout, KH, KW,
         stride, pad, dil, out.data_ptr(), ops, stream(),
         alg_bytes=4 * (x.numel() + w_okkc.numel() + N * Cout * OH * OW * (2 if residual is not None else 1)),
         alg_flops=2 * N * OH * OW * Cout * Cin * KH * KW)
    return out


def conv_splitk_plan(M, Cout, Cin, KH, KW):
    """number of K ranges emp_conv_splitk_bn_act_nhwc should use for this geometry (1: not worth splitting)"""
    return int(load().emp_conv_splitk_plan(int(M), int(Cout), int(Cin), int(KH), int(KW)))


def conv_splitk_bn_act_nhwc(x, w_okkc, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, dil=1, out=None,
                            k_splits=None):
    """conv_bn_act_nhwc for small launches: the reduction cut into k_splits ranges (emp_conv_splitk_bn_act_nhwc), partial
    sums through a workspace, epilogue in the second pass.  k_splits=None asks emp_conv_splitk_plan."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout, KH, KW, _ = w_okkc.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    _expect("conv: weights", w_okkc, torch.float32)
    if w_okkc.shape[3] != Cin or not w_okkc.is_contiguous():
        raise HipError(f"conv: weights must be a contiguous (Cout, KH, KW, Cin = {Cin}) tensor, got {tuple(w_okkc.shape)}")
    for what, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _expect(f"conv: {what}", t, torch.float32, (Cout,))
    if residual is not None:
        _expect("conv: residual", residual, torch.float32)
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    M = N * OH * OW
    if k_splits is None:
        k_splits = conv_splitk_plan(M, Cout, Cin, KH, KW)

    def pixel_stride(t):
        assert t.shape == (N, Cout, OH, OW) and t.dtype == torch.float32 and t.stride(1) == 1
        ps = t.stride(3)
        assert t.stride(2) == OW * ps and t.stride(0) == OH * OW * ps, "NHWC channel slice required"
        return ps

    if out is None:
        out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    ops = pixel_stride(out)
    rps = pixel_stride(residual) if residual is not None else 0
    work = torch.empty((int(k_splits) * M * Cout,), dtype=torch.float32, device=x.device)
    call('emp_conv_splitk_bn_act_nhwc', x.data_ptr(), _ptr(w_okkc), _ptr(scale), _ptr(shift),
         residual.data_ptr() if residual is not None else None, rps, int(bool(relu)), N, H, W, Cin, Cout, KH, KW,
         stride, pad, dil, int(k_splits), work.data_ptr(), out.data_ptr(), ops, stream(),
         alg_bytes=4 * (x.numel() + w_okkc.numel() + N * Cout * OH * OW * (2 if residual is not None else 1)),
         alg_flops=2 * N * OH * OW * Cout * Cin * KH * KW)
    return out


def gconv3x3_bn_act_nhwc(x, w_okkc, groups, scale=None, shift=None, relu=False, stride=1, out=None):
    """Grouped 3x3 convolution (padding 1, stride 1 / 2) + per-channel affine + ReLU on the fp32 matrix cores
    (emp_gconv3x3_bn_act_nhwc).  x: (N,C,H,W) fp32 channels_last; w_okkc: (C,3,3,C/groups) contiguous."""
    require_gpu()
    N, C, H, W = x.shape
    GW = C // groups
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    assert tuple(w_okkc.shape) == (C, 3, 3, GW) and w_okkc.is_contiguous() and GW * groups == C
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    assert out.shape == (N, C, OH, OW) and out.stride(1) == 1
    if N == 0:
        return out
    ops = out.stride(3)
    assert out.stride(2) == OW * ops and out.stride(0) == OH * OW * ops, "NHWC channel slice required"
    call('emp_gconv3x3_bn_act_nhwc', x.data_ptr(), C, _ptr(w_okkc), _ptr(scale), _ptr(shift), int(bool(relu)),
         N, H, W, groups, GW, stride, out.data_ptr(), ops, stream(),
         alg_bytes=4 * (x.numel() + w_okkc.numel() + N * C * OH * OW),
         alg_flops=2 * N * OH * OW * C * GW * 9)
    return out


def wino_tiles(N, H, W, dil, m=2):
    """(T, 3) int32 numpy table of Winograd F(m x m, 3x3) tiles (m = 2, 3 or 4) for a 3x3 convolution with dilation
    dil and padding dil: (n, y, x) of each tile's (m+2)x(m+2) patch origin, ordered (n, sub-grid row, sub-grid col,
    tile row, col)."""
    import numpy as np
    rows = []
    for ry in range(min(dil, H)):
        hs = -(-(H - ry) // dil)
        for ty in range(-(-hs // m)):
            rows.append(ry + dil * (m * ty - 1))
    cols = []
    for rx in range(min(dil, W)):
        ws = -(-(W - rx) // dil)
        for tx in range(-(-ws // m)):
            cols.append(rx + dil * (m * tx - 1))
    # group by sub-grid: rows are already grouped by ry, cols by rx
    ys, xs = np.meshgrid(np.array(rows, dtype=np.int32), np.array(cols, dtype=np.int32), indexing='ij')
    per = np.stack([ys.ravel(), xs.ravel()], axis=1)
    out = np.empty((N, len(per), 3), dtype=np.int32)
    out[:, :, 0] = np.arange(N, dtype=np.int32)[:, None]
    out[:, :, 1:] = per[None]
    return out.reshape(-1, 3)


def wino_filter_transform(w_oihw):
    """(Cout, Cin, 3, 3) -> U (16, Cout, Cin) = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], every step one
    fp32 rounding: rows first (r1 = 0.5 * ((g0 + g1) + g2), r2 = 0.5 * ((g0 - g1) + g2)), then columns likewise."""
    g = w_oihw.detach().float()

    def comb(a, b, c):
        return [a, 0.5 * ((a + b) + c), 0.5 * ((a - b) + c), c]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])              # 4 x (Cout, Cin, 3)
    U = []
    for rrow in rows:
        U.extend(comb(rrow[:, :, 0], rrow[:, :, 1], rrow[:, :, 2]))       # 4 x (Cout, Cin)
    return torch.stack(U, dim=0).contiguous()


def wino_conv_bn_act(x, U, tiles_dev, dil, scale=None, shift=None, relu=False, out=None, fused=True):
    """3x3 stride-1 convolution with padding == dilation through Winograd F(2x2,3x3) (emp_wino_input_transform,
    emp_gemm_nt_batched, emp_wino_output_transform).  x (N,Cin,H,W) fp32 channels_last; U from
    wino_filter_transform; tiles_dev = torch.from_numpy(wino_tiles(N,H,W,dil)).cuda()."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout = U.shape[1]
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    T = tiles_dev.shape[0]
    fused = fused and x.numel() < 2 ** 31
    Mw = torch.empty((16, T, Cout), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    assert out.shape == (N, Cout, H, W) and out.stride(1) == 1
    ops = out.stride(3)
    assert out.stride(2) == W * ops and out.stride(0) == H * W * ops, "NHWC channel slice required"
    st = stream()
    if fused:       # input transform inside the GEMM's loader: V never exists in memory
        call('emp_wino_gemm_fused', x.data_ptr(), N, H, W, Cin, dil, _ptr(tiles_dev), T, _ptr(U), Cout, _ptr(Mw), st,
             alg_bytes=4 * (x.numel() + U.numel() + Mw.numel()), alg_flops=2 * 16 * T * Cout * Cin)
    else:
        V = torch.empty((16, T, Cin), dtype=torch.float32, device=x.device)
        call('emp_wino_input_transform', x.data_ptr(), N, H, W, Cin, dil, _ptr(tiles_dev), T, _ptr(V), st,
             alg_bytes=4 * (x.numel() + V.numel()))
        call('emp_gemm_nt_batched', _ptr(V), _ptr(U), 16, T, Cout, Cin, _ptr(Mw), st,
             alg_bytes=4 * (V.numel() + U.numel() + Mw.numel()), alg_flops=2 * 16 * T * Cout * Cin)
    call('emp_wino_output_transform', _ptr(Mw), _ptr(tiles_dev), T, N, H, W, Cout, dil, _ptr(scale), _ptr(shift),
         int(bool(relu)), out.data_ptr(), ops, st, alg_bytes=4 * (Mw.numel() + N * Cout * H * W))
    return out


def pointwise_out_nhwc(x, w, bias=None):
    """1x1 convolution to 1..4 channels: x (N,C,H,W) fp32 channels_last, w (Cout, C) -> (N,Cout,H,W) contiguous
    (planar) fp32 (emp_pointwise_out_nhwc)."""
    require_gpu()
    N, C, H, W = x.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    Cout = w.shape[0]
    out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
    call('emp_pointwise_out_nhwc', x.data_ptr(), _ptr(w), _ptr(bias), N * H * W, H * W, C, Cout, _ptr(out), stream(),
         alg_bytes=4 * (x.numel() + out.numel()))
    return out


def bn_relu_maxpool_nhwc(x, scale, shift):
    """max_pool2d(relu(x*scale + shift), 3, 2, 1) on x (N,C,H,W) fp32 channels_last (emp_bn_relu_maxpool_nhwc)."""
    require_gpu()
    N, C, H, W = x.shape
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    call('emp_bn_relu_maxpool_nhwc', x.data_ptr(), _ptr(scale), _ptr(shift), N, H, W, C, y.data_ptr(), stream(),
         alg_bytes=4 * (x.numel() + y.numel()))
    return y


def logits_to_prob(logits, out=None):
    """sigmoid (C == 1) / softmax over dim 1 (C > 1) of (N,C,H,W) fp32 CUDA logits (emp_logits_to_prob).  `out`: optional
    contiguous (N,C,H,W) fp32 destination (may be `logits` itself)."""
    require_gpu()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4
    x = logits if logits.is_contiguous() else logits.contiguous()
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.is_cuda and out.dtype == torch.float32 and out.shape == x.shape and out.is_contiguous()
    if x.numel():
        call('emp_logits_to_prob', x.data_ptr(), N, C, H * W, out.data_ptr(), stream(), alg_bytes=8 * x.numel())
    return out


# ----------------------------------------------------------------------------- D10: PointRend subdivision step
def pr_upsample2x(logits):
    """(N,C,h,w) fp32 planar -> (upsampled (N,C,2h,2w), uncertainty (N, 4hw)): F.interpolate(x2, bilinear,
    align_corners=False) + calculate_uncertainty (point_rend.py:62-79, 244-248) in one pass"""
    require_gpu()
    N, C, h, w = logits.shape
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
    up = torch.empty((N, C, 2 * h, 2 * w), dtype=torch.float32, device=logits.device)
    unc = torch.empty((N, 4 * h * w), dtype=torch.float32, device=logits.device)
    call('emp_pr_upsample2x', logits.data_ptr(), N, C, h, w, up.data_ptr(), unc.data_ptr(), stream(),
         alg_bytes=4 * (logits.numel() + up.numel() + unc.numel()))
    return up, unc


def pr_topk(unc, k):
    """(N, HW) fp32 -> (N, k) int32 pixel indices of the k largest per row (exact; ties at the k-th value to the lowest
    indices; strictly larger ones first, each group in pixel order) -- torch.topk of point_rend.py:117"""
    require_gpu()
    N, HW = unc.shape
    assert unc.is_cuda and unc.dtype == torch.float32 and unc.is_contiguous() and 1 <= k <= HW
    wb = query('emp_pr_topk_work_bytes', N, HW)
    work = torch.empty((wb,), dtype=torch.uint8, device=unc.device)
    idx = torch.empty((N, k), dtype=torch.int32, device=unc.device)
    call('emp_pr_topk', unc.data_ptr(), N, HW, int(k), _ptr(work), wb, _ptr(idx), stream(), alg_bytes=4 * 6 * unc.numel())
    return idx


def pr_point_sample(features, coarse, idx, H, W, ld):
    """features (N,CF,Hf,Wf) channels_last, coarse (N,C,Hf,Wf) contiguous, idx (N,k) int32 on the (H,W) grid ->
    (X0, X1) two (N*k, ld) matrices: X0 = [sampled features | sampled coarse | 0], X1 = [uninitialised | sampled
    coarse | 0] (point_sample x 2 + the torch.cat of StandardPointHead.forward, point_rend.py:35-60,182-190)"""
    require_gpu()
    N, CF, Hf, Wf = features.shape
    C = coarse.shape[1]
    assert features.is_cuda and features.dtype == torch.float32 and features.stride(1) == 1
    fps = features.stride(3)
    assert features.stride(2) == Wf * fps and features.stride(0) == Hf * Wf * fps, "NHWC features required"
    assert coarse.is_contiguous() and coarse.shape == (N, C, Hf, Wf) and idx.shape[0] == N and idx.dtype == torch.int32
    k = idx.shape[1]
    X0 = torch.empty((N * k, ld), dtype=torch.float32, device=features.device)
    X1 = torch.empty((N * k, ld), dtype=torch.float32, device=features.device)
    call('emp_pr_point_sample', features.data_ptr(), fps, coarse.data_ptr(), N, Hf, Wf, CF, C, _ptr(idx), k, int(H), int(W),
         X0.data_ptr(), X1.data_ptr(), int(ld), stream(), alg_bytes=4 * N * k * (4 * CF + 2 * ld))
    return X0, X1


def pr_scatter(points, idx, logits):
    """logits (N,C,H,W)[n, c, idx[n, j]] = points[n * k + j, c]   (scatter_ of point_rend.py:258-265)"""
    require_gpu()
    N, C, H, W = logits.shape
    k = idx.shape[1]
    assert points.shape[0] == N * k and points.stride(1) == 1 and logits.is_contiguous()
    call('emp_pr_scatter', points.data_ptr(), points.stride(0), _ptr(idx), N, C, k, H * W, logits.data_ptr(), stream(),
         alg_bytes=8 * N * k * C)
    return logits


def as_pixels(mat, cols=None):
    """(P, ld) row-major matrix -> the (1, cols, P, 1) channels_last view emp_conv_bn_act_nhwc takes (a 1x1 convolution
    over P 'pixels'); cols < ld gives the channel slice [0, cols)"""
    P, ld = mat.shape
    v = mat.view(1, P, 1, ld).permute(0, 3, 1, 2)
    return v if cols is None else v[:, :cols]


def stem_conv7_bn_relu_maxpool(x, w_tc, scale, shift):
    """max_pool2d(relu(bn(conv2d(x, w, stride 2, padding 3))), 3, 2, 1) of the ResNet stem in one kernel
    (emp_stem_conv7_bn_relu_maxpool).  x: (N,1,H,W) fp32 contiguous; w_tc: (49, 64); returns (N,64,PH,PW) channels_last."""
    require_gpu()
    N, C, H, W = x.shape
    assert C == 1 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    assert tuple(w_tc.shape) == (49, 64) and w_tc.is_contiguous()
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    y = torch.empty((N, 64, PH, PW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    call('emp_stem_conv7_bn_relu_maxpool', x.data_ptr(), _ptr(w_tc), _ptr(scale), _ptr(shift), N, H, W, y.data_ptr(),
         stream(), alg_bytes=4 * (x.numel() + y.numel()), alg_flops=2 * N * OH * OW * 64 * 49)
    return y


def wino4_filter_transform(w_oihw):
    """(Cout, Cin, 3, 3) -> U (36, Cout, Cin) = fp32(G g G^T) for F(4x4,3x3), evaluated elementwise in fp64 (rows
    first, then columns): G rows = [1/4 0 0], [-1/6 -1/6 -1/6], [-1/6 1/6 -1/6], [1/24 1/12 1/6],
    [1/24 -1/12 1/6], [0 0 1]."""
    g = w_oihw.detach().double().cpu()

    def comb(a, b, c):
        return [a / 4.0, -((a + b) + c) / 6.0, ((b - a) - c) / 6.0, (a / 24.0 + b / 12.0) + c / 6.0,
                (a / 24.0 - b / 12.0) + c / 6.0, c]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])
    U = []
    for r in rows:
        U.extend(comb(r[:, :, 0], r[:, :, 1], r[:, :, 2]))
    return torch.stack(U, dim=0).float().contiguous()


def wino4_conv_bn_act(x, U, tiles_dev, dil, scale=None, shift=None, relu=False, out=None):
    """3x3 stride-1 convolution with padding == dilation through Winograd F(4x4,3x3) (emp_wino4_input_transform,
    emp_gemm_nt_batched x 36, emp_wino4_output_transform); tiles from wino_tiles(..., m=4)."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout = U.shape[1]
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    T = tiles_dev.shape[0]
    V = torch.empty((36, T, Cin), dtype=torch.float32, device=x.device)
    Mw = torch.empty((36, T, Cout), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    assert out.shape == (N, Cout, H, W) and out.stride(1) == 1
    ops = out.stride(3)
    assert out.stride(2) == W * ops and out.stride(0) == H * W * ops, "NHWC channel slice required"
    st = stream()
    call('emp_wino4_input_transform', x.data_ptr(), N, H, W, Cin, dil, _ptr(tiles_dev), T, _ptr(V), st,
         alg_bytes=4 * (x.numel() + V.numel()))
    call('emp_gemm_nt_batched', _ptr(V), _ptr(U), 36, T, Cout, Cin, _ptr(Mw), st,
         alg_bytes=4 * (V.numel() + U.numel() + Mw.numel()), alg_flops=2 * 36 * T * Cout * Cin)
    call('emp_wino4_output_transform', _ptr(Mw), _ptr(tiles_dev), T, N, H, W, Cout, dil, _ptr(scale), _ptr(shift),
         int(bool(relu)), out.data_ptr(), ops, st, alg_bytes=4 * (Mw.numel() + N * Cout * H * W))
    return out


def conv_k_slab(M, Cout, batch=1, has_residual=False, Cin=None, geom=None, relu=False):
    """K-slab (16, 32 or 64) emp_conv_bn_act_nhwc / emp_gemm_nt_batched use for an (M x Cout) output, `batch` GEMMs per
    launch: fixes the summation order the oracle mirrors.  geom = (KH, KW, stride, pad) of a convolution: short-K
    pointwise layers go to the weight-stationary kernel (emp_conv1x1.hip), which sums over 64-channel slabs."""
    if geom is not None:
        KH, KW, stride, pad = geom
        return int(load().emp_conv_k_slab_geom(int(M), int(Cout), int(bool(has_residual)), int(Cin), int(KH), int(KW),
                                               int(stride), int(pad), 2 if relu == 'gate' else int(bool(relu))))
    if Cin is not None:
        return int(load().emp_conv_k_slab_cin(int(M), int(Cout), int(batch), int(bool(has_residual)), int(Cin)))
    return int(load().emp_conv_k_slab(int(M), int(Cout), int(batch), int(bool(has_residual))))


_WINO3_G = [[0.5, 0.0, 0.0], [-0.5, -0.5, -0.5], [-1.0 / 6, 1.0 / 6, -1.0 / 6], [1.0 / 6, 1.0 / 3, 2.0 / 3], [0.0, 0.0, 1.0]]


def wino3_filter_transform(w_oihw):
    """(Cout, Cin, 3, 3) -> U (25, Cout, Cin) = fp32(G g G^T) for F(3x3,3x3), elementwise fp64: rows first, each
    row combination ((G[u][0]*g0 + G[u][1]*g1) + G[u][2]*g2), then columns likewise."""
    g = w_oihw.detach().double().cpu()

    def comb(a, b, c):
        return [(G[0] * a + G[1] * b) + G[2] * c for G in _WINO3_G]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])
    U = []
    for r in rows:
        U.extend(comb(r[:, :, 0], r[:, :, 1], r[:, :, 2]))
    return torch.stack(U, dim=0).float().contiguous()


def wino3_conv_bn_act(x, U, tiles_dev, dil, scale=None, shift=None, relu=False, out=None):
    """3x3 stride-1 convolution with padding == dilation through Winograd F(3x3,3x3) (emp_wino3_input_transform,
    emp_gemm_nt_batched x 25, emp_wino3_output_transform); tiles from wino_tiles(..., m=3)."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout = U.shape[1]
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    T = tiles_dev.shape[0]
    V = torch.empty((25, T, Cin), dtype=torch.float32, device=x.device)
    Mw = torch.empty((25, T, Cout), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    assert out.shape == (N, Cout, H, W) and out.stride(1) == 1
    ops = out.stride(3)
    assert out.stride(2) == W * ops and out.stride(0) == H * W * ops, "NHWC channel slice required"
    st = stream()
    call('emp_wino3_input_transform', x.data_ptr(), N, H, W, Cin, dil, _ptr(tiles_dev), T, _ptr(V), st,
         alg_bytes=4 * (x.numel() + V.numel()))
    call('emp_gemm_nt_batched', _ptr(V), _ptr(U), 25, T, Cout, Cin, _ptr(Mw), st,
         alg_bytes=4 * (V.numel() + U.numel() + Mw.numel()), alg_flops=2 * 25 * T * Cout * Cin)
    call('emp_wino3_output_transform', _ptr(Mw), _ptr(tiles_dev), T, N, H, W, Cout, dil, _ptr(scale), _ptr(shift),
         int(bool(relu)), out.data_ptr(), ops, st, alg_bytes=4 * (Mw.numel() + N * Cout * H * W))
    return out


def conv_bn_act_proj_nhwc(x, w_okkc, scale, shift, relu, proj_w, proj_b=None, stride=1, pad=0, dil=1, keep=False):
    """Convolution + affine + ReLU whose epilogue also applies a following 1x1 convolution to <= 4 channels
    (emp_conv_bn_act_proj_nhwc).  x (N,Cin,H,W) channels_last, w_okkc (Cout in {128, 256}, KH, KW, Cin), proj_w
    (n, Cout), proj_b (n) or None.  Returns the planar (N, n, OH, OW) result (and the activation if keep)."""
    require_gpu()
    N, Cin, H, W = x.shape
    Cout, KH, KW, _ = w_okkc.shape
    n = proj_w.shape[0]
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    acc = torch.zeros((N, n, OH, OW), dtype=torch.float32, device=x.device)
    act = (torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
           if keep else None)
    call('emp_conv_bn_act_proj_nhwc', x.data_ptr(), _ptr(w_okkc), _ptr(scale), _ptr(shift), int(bool(relu)), N, H, W,
         Cin, Cout, KH, KW, stride, pad, dil, _ptr(proj_w), n, _ptr(acc), act.data_ptr() if keep else None, 0,
         stream(), alg_bytes=4 * (x.numel() + w_okkc.numel() + acc.numel() + (act.numel() if keep else 0)),
         alg_flops=2 * N * OH * OW * Cout * (Cin * KH * KW + n))
    if proj_b is not None:
        acc += proj_b.view(1, -1, 1, 1)
    return (acc, act) if keep else acc

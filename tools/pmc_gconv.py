"""A few launches of the grouped 3x3 convolution kernel on the RegNetY-6.4GF stage shapes (128 slices of a 512^2 tile per
call) for rocprofv3 --pmc passes, and the summary of such a pass.
collect:   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
           SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_gconv.py
           (a second pass with SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS into <dir2>)
summarise: python tools/pmc_gconv.py --summary <dir> [<dir2>] > profiles/<name>.md"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B = 128
#        name                       C     G   hw  stride
CASES = [('stage1 144 (2 x 72) @128', 144, 2, 128, 1), ('stage2 288 (4 x 72) @64', 288, 4, 64, 1),
         ('stage3 576 (8 x 72) @32', 576, 8, 32, 1), ('stage3.0 576 @64 stride 2', 576, 8, 64, 2),
         ('stage4 1296 (18 x 72) @16', 1296, 18, 16, 1)]


def run():
    import torch
    from empanada_amd import _hip
    for name, C, G, hw, stride in CASES:
        x = torch.randn(B, C, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
        w = (torch.randn(C, C // G, 3, 3, device='cuda') * 0.04).permute(0, 2, 3, 1).contiguous()
        sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
        for _ in range(3):
            _hip.gconv3x3_bn_act_nhwc(x, w, G, sc, sh, True, stride)
        torch.cuda.synchronize()
        print(name, flush=True)


def _load(d):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if 'gconv3x3' in r['Kernel_Name']]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r['Dispatch_Id'], {'k': r['Kernel_Name'].split('(')[0].replace('void ', '')})[r['Counter_Name']] = \
            float(r['Counter_Value'])
    return list(by.values())


def summary(d, d2=None):
    a = _load(d)
    b = _load(d2) if d2 else [{}] * len(a)
    names = [c[0] for c in CASES for _ in range(3)]
    print("# SQ counters of gconv3x3_f32_kernel<24, 5> on the RegNetY-6.4GF stage shapes, 128 slices per call (rocprofv3 --pmc)\n")
    print("WAVE_CYCLES = WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: matrix pipe / dependency; "
          "WAIT_INST_LDS is its LDS-issue part) + ACTIVE_INST_ANY.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x "
          "SQ_BUSY_CU_CYCLES).  Second pass: LDS bank-conflict cycles per LDS-active cycle.\n")
    print("| layer | parked % | issue-stall % (LDS part) | active % | MFMA busy / (4 x CU busy) | LDS conflict / active |\n|---|---|---|---|---|---|")
    for i, c in enumerate(a):
        if i % 3 != 2:
            continue                                   # third launch of every case
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        e = b[i] if i < len(b) else {}
        lds = (f"{e.get('SQ_LDS_BANK_CONFLICT', 0) / max(e.get('SQ_LDS_IDX_ACTIVE', 0), 1):.3f}"
               if 'SQ_LDS_BANK_CONFLICT' in e else 'n/a')
        print(f"| {names[i] if i < len(names) else '?'} | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.1f} | "
              f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} ({100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.1f}) | "
              f"{100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} | "
              f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(4 * c.get('SQ_BUSY_CU_CYCLES', 1), 1):.3f} | {lds} |")


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--summary':
        summary(*sys.argv[2:4])
    else:
        run()

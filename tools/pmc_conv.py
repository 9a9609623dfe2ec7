"""A few launches of the MFMA kernel on the bench's layer shapes (32 slices of 1024^2 per model call) for rocprofv3
--pmc passes, and the summary of such a pass.
collect:   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
           SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv
           -d <dir> -- python3 tools/pmc_conv.py
summarise: python tools/pmc_conv.py --summary <dir> > profiles/<name>.md"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B = 32
#        name                          cin   cout  hw  k pad dil res
CASES = [('head.pw 256->256 @256', 256, 256, 256, 1, 0, 1, False), ('l4.down 1024->2048 @64', 1024, 2048, 64, 1, 0, 1, False),
         ('l4.conv1 2048->512 @64', 2048, 512, 64, 1, 0, 1, False), ('l4.conv3 512->2048 @64 +res', 512, 2048, 64, 1, 0, 1, True),
         ('l1.conv3 64->256 @256 +res', 64, 256, 256, 1, 0, 1, True), ('l2.conv1 512->128 @128', 512, 128, 128, 1, 0, 1, False),
         ('l3.conv2 256->256 3x3 s2 @128', 256, 256, 128, 3, 1, 1, False)]


def run():
    import torch
    from empanada_amd import _hip
    for name, cin, cout, hw, k, pad, dil, res in CASES:
        x = torch.randn(B, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device='cuda') * 0.02).permute(0, 2, 3, 1).contiguous()
        sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
        stride = 2 if 's2' in name else 1
        r = None
        if res:
            r = torch.randn(B, cout, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            _hip.conv_bn_act_nhwc(x, w, sc, sh, r, True, stride, pad, dil)
        torch.cuda.synchronize()
        print(name, flush=True)
    # batched GEMM of layer4's F(4,3): 36 x [T x 512] x [512 x 512], T = 32 images x 16 x 16 tiles
    V = torch.randn(36, 8192, 512, device='cuda')
    U = torch.randn(36, 512, 512, device='cuda')
    M = torch.empty(36, 8192, 512, device='cuda')
    for _ in range(3):
        _hip.call('emp_gemm_nt_batched', V.data_ptr(), U.data_ptr(), 36, 8192, 512, 512, M.data_ptr(), _hip.stream())
    torch.cuda.synchronize()
    print('wino4 gemm 36 x 8192 x 512 x 512', flush=True)


def summary(d):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if 'conv_igemm' in r['Kernel_Name']]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r['Dispatch_Id'], {'k': r['Kernel_Name'].split('(')[0].replace('void ', ''),
                                          'grid': r.get('Grid_Size', '')})[r['Counter_Name']] = float(r['Counter_Value'])
    names = [c[0] for c in CASES for _ in range(3)] + ['wino4 gemm 36x8192x512x512'] * 3
    print("# SQ counters of the MFMA kernel on the bench's layer shapes (rocprofv3 --pmc, one pass)\n")
    print("WAVE_CYCLES = WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA pipe / dependency; "
          "WAIT_INST_LDS is its LDS-issue part) + ACTIVE_INST_ANY, per the MI355X guide.  MFMA busy = "
          "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4 SIMDs... reported as the ratio to CU-busy cycles x 4.\n")
    print("| layer | kernel | parked % | issue-stall % (LDS part) | active % | MFMA busy / (4 x CU busy) |\n|---|---|---|---|---|---|")
    for i, (disp, c) in enumerate(by.items()):
        if i % 3 != 2:
            continue                                   # third launch of every case
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        print(f"| {names[i] if i < len(names) else '?'} | `{c['k']}` | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.1f} | "
              f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} ({100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.1f}) | "
              f"{100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} | "
              f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(4 * c.get('SQ_BUSY_CU_CYCLES', 1), 1):.3f} |")


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--summary':
        summary(sys.argv[2])
    else:
        run()

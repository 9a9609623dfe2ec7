"""Where the time of ONE batch-1 forward goes (the per-slice protocol's forward: 1 x 1 x S x S, tuned, replayed as a HIP
graph).  Run under `rocprofv3 --kernel-trace`; `summary` mode then prints the dispatches of the last replay in order:
duration, gap to the previous dispatch, grid / block, kernel.
usage: rocprofv3 --kernel-trace -d gpurun_out/prof_b1 -o b1 --output-format csv -- python3 tools/prof_batch1.py run [S] [model]
       python tools/prof_batch1.py summary gpurun_out/prof_b1/.../b1_kernel_trace.csv [n_replays]"""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch

    import bench
    from empanada_amd.models import GraphedForward, prepare_for_inference, tune_fused_convs
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    name = sys.argv[3] if len(sys.argv) > 3 else 'pdl_r50'
    dev = torch.device('cuda')
    net = prepare_for_inference(bench.build_model(name), dev)
    x = torch.rand((1, 1, S, S), device=dev).contiguous(memory_format=torch.channels_last)
    rep = tune_fused_convs(net, x)
    counts = {}
    for best, _ in rep.values():
        counts[best] = counts.get(best, 0) + 1
    print('conv sites at batch 1:', counts)
    g = GraphedForward(net)
    with torch.no_grad():
        for _ in range(3):
            g(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 20
        for _ in range(N):
            g(x)
        torch.cuda.synchronize()
        print(f'graph replay: {(time.perf_counter() - t0) / N * 1e3:.3f} ms / forward')


def summary():
    rows = list(csv.DictReader(open(sys.argv[2])))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the last n replays are identical sequences: find the period from the tail
    names = [r['Kernel_Name'] for r in rows]
    per = None
    for p in range(50, len(rows) // 2):
        if names[-p:] == names[-2 * p:-p]:
            per = p
            break
    assert per, 'no periodic tail found'
    sel = rows[-per:]
    prev_end = int(rows[-per - 1]['End_Timestamp'])
    tot = busy = 0
    print(f'# one replayed batch-1 forward: {per} dispatches\n')
    print('| # | us | gap us | grid | block | kernel |\n|---|---|---|---|---|---|')
    agg = {}
    for i, r in enumerate(sel):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gx = [int(r.get(f'Grid_Size_{a}', r.get('Grid_Size', 0)) or 0) for a in 'XYZ'] if 'Grid_Size_X' in r else [int(r['Grid_Size'])]
        bx = [int(r.get(f'Workgroup_Size_{a}', 0) or 0) for a in 'XYZ'] if 'Workgroup_Size_X' in r else [int(r['Workgroup_Size'])]
        k = r['Kernel_Name']
        k = k if len(k) < 90 else k[:87] + '...'
        print(f'| {i} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} | {gx} | {bx} | `{k}` |')
        busy += e - s
        a = agg.setdefault(k, [0, 0])
        a[0] += e - s
        a[1] += 1
        prev_end = e
    tot = int(sel[-1]['End_Timestamp']) - int(rows[-per - 1]['End_Timestamp'])
    print(f'\nwindow {tot / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({100 * busy / tot:.0f} %)\n')
    print('| kernel | total us | calls |\n|---|---|---|')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f'| `{k}` | {v[0] / 1e3:.1f} | {v[1]} |')


if __name__ == '__main__':
    (run if sys.argv[1] == 'run' else summary)()

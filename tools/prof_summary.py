"""Summarise a rocprofv3 --kernel-trace CSV over the bench's TIMED region only.

The full-run --stats table is dominated by MIOpen's one-off find/verification kernels of the warm-up;
this script keeps the dispatches after the end of the warm-up step (marked by the first fill kernel)
and prints per-kernel totals, call counts and average durations.
usage: python tools/prof_summary.py <kernel_trace.csv> [n_steps] > profiles/<name>.md
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [r for r in rows if 'fill_u32_kernel<1>' in r['Kernel_Name'] or 'fill_table_kernel' in r['Kernel_Name']]
    t0 = int(marks[0]['End_Timestamp']) if len(marks) > steps else int(rows[0]['Start_Timestamp'])
    sel = [r for r in rows if int(r['Start_Timestamp']) > t0]
    t1 = int(sel[-1]['End_Timestamp'])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r['Kernel_Name']]
        a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a[1] += 1
    busy = sum(v[0] for v in agg.values())
    print(f"# rocprofv3 kernel trace, timed region only ({steps} steps)\n")
    print(f"window {(t1 - t0) / 1e6:.1f} ms, GPU busy {busy / 1e6:.1f} ms ({100 * busy / (t1 - t0):.0f}%), "
          f"{len(sel)} dispatches\n")
    mine = ('median_harden', 'median_step', 'find_centers', 'sort_centers', 'group_pixels', 'fuse_', 'row_runs',
            'runs_fix', 'label_', 'overlap_next', 'fill_', 'scan_', 'vote_', 'pair_inter', 'box_pairs', 'cells_', 'dwconv_', 'bn_act_', 'rle_', 'conv_igemm', 'wino_', 'upsample_nhwc', 'upsample_planar',
            'group_centers', 'conv1x1_ws', 'pr_hist', 'pr_pick', 'pr_count', 'pr_emit', 'pr_init', 'pr_sample', 'pr_scatter', 'pr_upsample', 'wino3_', 'wino4_', 'pointwise_out', 'bn_relu_maxpool', 'slices_to_input', 'gconv3x3', 'trk_', 'triplet', 'stem7')
    print("| kernel | total ms | calls | avg us | % busy | hand-written |\n|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        hw = any(m in k for m in mine) and 'at::' not in k
        if v[0] / busy < 0.002 and not hw:
            continue
        name = k if len(k) < 100 else k[:97] + '...'
        print(f"| `{name}` | {v[0] / 1e6:.3f} | {v[1]} | {v[0] / v[1] / 1e3:.1f} | {100 * v[0] / busy:.2f} | {'yes' if hw else ''} |")


if __name__ == '__main__':
    main()

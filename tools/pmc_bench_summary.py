"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py over its TIMED region:
per hand-written kernel, launches and mean HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes;
the factor 2 is the gfx950 correction calibrated in profiles/r1_pmc_postproc_256x512x512.md).
usage: python tools/pmc_bench_summary.py <fetch_dir> <write_dir> > profiles/<name>.md"""
import collections
import csv
import glob
import sys


def load(d, counter):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    k = glob.glob(d + '/*/*_kernel_trace.csv')[0]
    gy = {}
    for r in csv.DictReader(open(k)):
        gy[r['Dispatch_Id']] = (int(r['Grid_Size_Y']), int(r['Start_Timestamp']))
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    marks = [i for i, r in enumerate(rows) if 'fill_table_kernel' in r['Kernel_Name'] or 'fill_u32_kernel<1>' in r['Kernel_Name']]
    rows = rows[marks[0] + 1:] if len(marks) > 1 else rows          # after the warm-up pass
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'conv_igemm' in name:
            g = gy.get(r['Dispatch_Id'], (1, 0))[0]
            name += f' [batched x{g}]' if g > 1 else ''
        a = agg[name]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    return agg


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    mine = ('conv_igemm', 'conv1x1_ws', 'stem7', 'pr_', 'logits_to_prob', 'gconv3x3', 'wino_', 'wino3_', 'wino4_', 'pointwise_out', 'bn_relu_maxpool', 'slices_to_input', 'dwconv', 'bn_act', 'upsample_', 'median', 'find_centers', 'group_pixels', 'fuse_',
            'row_runs', 'label_', 'overlap_next', 'fill_table', 'fill_u32', 'trk_', 'trip_', 'vote_', 'pair_inter', 'box_pairs')
    print("# rocprofv3 PMC passes over bench.py (timed pass only): HBM traffic per launch\n")
    print("traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB), separate passes\n")
    print("| kernel | launches | FETCH_SIZE KiB / launch | WRITE_SIZE KiB / launch | traffic MB / launch |\n|---|---|---|---|---|")
    for name in sorted(fetch, key=lambda n: -(2 * fetch[n][0] + write.get(n, [0, 1])[0])):
        if not any(m in name for m in mine) or 'at::' in name:
            continue
        f, n = fetch[name]
        w = write.get(name, [0.0, max(n, 1)])
        tr = (2 * f / n + w[0] / max(w[1], 1)) * 1024 / 1e6
        print(f"| `{name[:70]}` | {n} | {f / n:.0f} | {w[0] / max(w[1], 1):.0f} | {tr:.1f} |")


main()

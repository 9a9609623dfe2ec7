"""Isolated run of the orthoplane post-processing (no forward running beside it): per ABI call, HIP-event duration and
achieved HBM GB/s against the algorithmic bytes of DESIGN.md section 4 -- the roofline evidence for the post-processing
kernels (inside bench.py they share the GPU with the next plane's forward and read 1.3-2x longer).
usage: python tools/postproc_isolated.py [size] > profiles/<name>.md ; also meant to be run under
`rocprofv3 --kernel-trace --stats` (tools/prof_summary.py does not apply: there is no warm-up marker; use the stats csv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from empanada_amd import _hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda', 0)
stacks, heads, n_obj, _ = bench.build_inputs_ortho(S, dev)
del stacks
bench.postprocess_planes(heads, (S, S, S), None, {})             # warm-up
torch.cuda.synchronize()
_hip.PROFILE = {}
REPS = 3
for _ in range(REPS):
    n_found, vols, _ = bench.postprocess_planes(heads, (S, S, S), None, {})
torch.cuda.synchronize()
prof, _hip.PROFILE = _hip.PROFILE, None
vox = float(S) ** 3
f = float((heads['xy']['sem'] >= bench.ENGINE['confidence_thr']).float().mean().item())
print(f"# post-processing kernels in isolation, orthoplane {S}^3 (3 planes + consensus + fill), {n_found} instances\n")
print(f"HIP events around every ABI call on its launch stream, {REPS} passes; algorithmic bytes per voxel as in DESIGN.md "
      f"section 4 (thing fraction {f:.3f}); peak = 8000 GB/s (spec), ~6300 achievable\n")
print("| ABI call | calls / pass | avg ms | ms / pass | alg. B / voxel | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|---|")
rows = []
for name, evs in prof.items():
    ms = [e[0].elapsed_time(e[1]) for e in evs]
    rows.append((name, len(ms) / REPS, float(np.mean(ms)), float(np.sum(ms)) / REPS))
for name, calls, avg, per in sorted(rows, key=lambda r: -r[3]):
    if name in bench.ALG_BYTES:
        b = bench.ALG_BYTES[name](f)
        gbs = b * vox / (avg * 1e-3) / 1e9
        print(f"| `{name}` | {calls:.0f} | {avg:.3f} | {per:.2f} | {b:.2f} | {gbs:.0f} | {gbs / 8000:.2f} |")
    else:
        print(f"| `{name}` | {calls:.0f} | {avg:.3f} | {per:.2f} | O(#runs) | — | — |")
print(f"\ntotal {sum(r[3] for r in rows):.1f} ms of GPU time per pass over {3 * vox / 1e6:.0f} M plane-voxels")

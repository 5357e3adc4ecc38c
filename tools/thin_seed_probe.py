"""GPU side of the second thin fixture's seed choice (tests/golden/make_golden.py stage_thin_search / thin_cands):
for every candidate frame in tests/golden/_cand/ count the integers (z symbols, CDF indexes, y symbols) on which the
product disagrees with the reference's run.  Prints one line per seed."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd.vaeformer import VAEformer  # noqa: E402

dev = torch.device("cuda:0")
net = VAEformer(0, **synth.thin_model_kwargs())
synth.load_synthetic(net, seed=7)
net = net.to(dev)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "_cand", "thin_cand_*.npz"))):
    g = np.load(f)
    seed = int(os.path.basename(f)[len("thin_cand_"):-4])
    x = synth.synth_frame(8, seed=seed).unsqueeze(0).to(dev)
    y = net.encode_latent(x, type='float')[0]
    s = net._latent_side_frame(y[0])
    torch.cuda.synchronize()
    z_mis = int((s["z_sym"].cpu().reshape(-1).numpy() != g["z_sym"].reshape(-1)).sum())
    i_mis = int((s["idx"].cpu().reshape(-1).numpy() != g["idx_full"].astype(np.int32)).sum())
    y_mis = int((s["y_sym"].cpu().reshape(-1).numpy() != g["sym_full"].astype(np.int32)).sum())
    print(f"seed {seed}: z flips {z_mis}, idx flips {i_mis}, y-symbol flips {y_mis}; reference margins z "
          f"{g['margin_z'][0]:.2e} y {g['margin_y'][0]:.2e} scale {g['margin_scale'][0]:.2e}", flush=True)
    rows.append(dict(seed=seed, z_flips=z_mis, idx_flips=i_mis, sym_flips=y_mis))
# cross-implementation agreement rate: a frame's streams are byte-identical to the reference-python-written ones, and the
# reference's .bin decodes on this build, exactly when no integer differs
ok = [r for r in rows if r["z_flips"] == r["idx_flips"] == r["sym_flips"] == 0]
print(f"{len(ok)} of {len(rows)} thin frames agree with the reference run on EVERY integer (z symbols, CDF indexes, y symbols)")
import json  # noqa: E402
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(dict(frames=len(rows), all_integers_equal=len(ok), per_frame=rows),
          open(os.path.join(root, "gpurun_out", "thin_cross_decode_rate.json"), "w"), indent=1)

"""BUILD-CONTAINER-ONLY check (skipped wherever /root/reference is absent, e.g. the GPU box):
the reference's OWN Python `compress()` / `decompress()` run with the PRODUCT's coder
(cra5_amd.ans, C ABI) plugged in as `compressai.ans` must reproduce the committed golden byte
streams (which were produced with the oracle coder) and round-trip."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from cra5_amd import ans as product_ans
from oracle import ref_shim
ref_shim.install(ans_module=product_ans)
sys.path.insert(0, %(root)r + "/tests/golden")
import importlib.util
spec = importlib.util.spec_from_file_location("mk", %(root)r + "/tests/golden/make_golden.py")
# reuse the thin-model builder without re-running the shim install
src = open(%(root)r + "/tests/golden/make_golden.py").read().replace("ref_shim.install()", "pass")
ns = {"__name__": "mk", "__file__": %(root)r + "/tests/golden/make_golden.py"}
exec(compile(src, "make_golden.py", "exec"), ns)
net = ns["build_thin"](); ns["load_synth"](net, seed=7)
g = np.load(%(root)r + "/tests/golden/thin_e2e.npz")
x = ns["synth"].synth_frame(8, seed=int(g["x_seed"][0])).unsqueeze(0)
with torch.no_grad():
    out = net.compress(x)
    rec = net.decompress(out["strings"], out["z_shape"], return_format="latent")
assert out["strings"][0][0] == g["y_string"].tobytes(), "y stream differs"
assert out["strings"][1][0] == g["z_string"].tobytes(), "z stream differs"
assert np.allclose(rec.reshape(-1)[::37].numpy(), g["y_hat_sub"], atol=1e-6)
print("REFERENCE+PRODUCT-CODER OK", len(out["strings"][0][0]), len(out["strings"][1][0]))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/cra5"), reason="reference tree only exists in the build container")
def test_reference_python_with_product_coder():
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "REFERENCE+PRODUCT-CODER OK" in r.stdout

"""the CNN zoo codecs (bmshj2018-factorized / -hyperprior, mbt2018-mean) on the HIP
kernels against golden vectors produced by the REFERENCE's own classes
(cra5/models/compressai/models/google.py:64-508; tests/golden/make_golden.py --stage cnn)."""
import json

import numpy as np
import pytest
import torch

from cra5_amd import cnn, synth

gpu = pytest.mark.gpu
N, M = 32, 48


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float(torch.sqrt(torch.mean((a - b) ** 2)) / torch.sqrt(torch.mean(b ** 2)))


def sub(t, step):
    return t.detach().reshape(-1)[::step].cpu()


@gpu
@pytest.mark.parametrize("name,cls", [("factorized", cnn.FactorizedPrior), ("hyperprior", cnn.ScaleHyperprior),
                                      ("meanscale", cnn.MeanScaleHyperprior)])
def test_cnn_zoo_vs_reference_golden(dev, golden_dir, name, cls, ledger):
    g = np.load(f"{golden_dir}/cnn_zoo.npz")
    keys = json.load(open(f"{golden_dir}/state_keys.json"))["cnn"][name]
    net = cls(N, M)
    synth.load_synthetic(net, seed=11)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == keys        # same module tree / buffers / tables
    net = net.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    y = net.g_a(x[0])
    e_y = rel(y, g[f"{name}_y"])
    fw = net(x)
    e_fw = rel(sub(fw["x_hat"], 5), g[f"{name}_xhat_fw"])
    gy = torch.Generator().manual_seed(5)
    y_hat = torch.round(3.0 * torch.randn((1,) + tuple(y.shape), generator=gy))[0].to(dev)
    e_dec = rel(sub(net.g_s(y_hat), 5), g[f"{name}_xhat_synth"])
    print(f"{name}: y rel {e_y:.2e}, forward x_hat rel {e_fw:.2e}, decoder (given y_hat) rel {e_dec:.2e}")
    assert e_y <= 1e-5 and e_dec <= 1e-5
    assert e_fw <= 1e-3          # forward() rounds y: a .5-boundary flip moves x_hat locally, not globally
    for k, v in fw["likelihoods"].items():
        bits = float((-torch.log2(v.double())).sum())
        assert abs(bits - g[f"{name}_bits_{k}"][0]) <= 2e-3 * g[f"{name}_bits_{k}"][0], k
    if name != "factorized":
        z = net.h_a(net._h_a_in(y))
        assert rel(z, g[f"{name}_z"]) <= 1e-5
    # entropy coding: streams byte-identical to the reference python's when the integer side agrees,
    # decode(encode(x)) reproduces the forward pass' quantised reconstruction either way
    out = net.compress(x)
    assert list(out["shape"]) == list(g[f"{name}_shape"])
    same = [out["strings"][i][0] == g[f"{name}_string{i}"].tobytes() for i in range(len(out["strings"]))]
    lens = [(len(out["strings"][i][0]), len(g[f"{name}_string{i}"])) for i in range(len(out["strings"]))]
    print(f"{name}: streams identical to the reference python's: {same} (bytes {lens})")
    for (a, b) in lens:
        assert abs(a - b) <= 16
    rec = net.decompress(out["strings"], out["shape"])["x_hat"]
    assert rec.shape == x.shape
    assert rel(sub(rec, 5), g[f"{name}_xhat_rt"]) <= 1e-3
    if all(same):
        assert rel(sub(rec, 5), g[f"{name}_xhat_rt"]) <= 1e-5
        ledger.ran(f"cnn {name}: streams == reference-python-written streams, round trip <= 1e-5", f"bytes {lens}")
    else:
        ledger.not_applicable(f"cnn {name}: streams == reference-python-written streams", f"identical per stream: {same}")
    # ... and decoding the REFERENCE's streams gives the reference's reconstruction
    ref_strings = [[g[f"{name}_string{i}"].tobytes()] for i in range(len(out["strings"]))]
    if all(same) or name == "factorized":
        rec2 = net.decompress(ref_strings, out["shape"])["x_hat"]
        assert rel(sub(rec2, 5), g[f"{name}_xhat_rt"]) <= 1e-5
        ledger.ran(f"cnn {name}: reference-python-written streams decode to the reference's reconstruction")
    else:
        ledger.not_applicable(f"cnn {name}: reference-python-written streams decode to the reference's reconstruction",
                              "needs bit-identical h_s indexes; streams differ")


@gpu
def test_factorized_relu_vs_reference_golden(dev, golden_dir):
    """`bmshj2018-factorized-relu` (google.py:166-199) against the reference class (cnn_relu.npz, make_golden.py
    --stage cnn_relu)."""
    g = np.load(f"{golden_dir}/cnn_relu.npz")
    keys = json.load(open(f"{golden_dir}/state_keys.json"))["cnn"]["factorized_relu"]
    net = cnn.FactorizedPriorReLU(N, M)
    synth.load_synthetic(net, seed=11)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == keys
    net = net.to(dev)
    x = torch.from_numpy(np.load(f"{golden_dir}/cnn_zoo.npz")["x"]).to(dev)
    y = net.g_a(x[0])
    assert rel(y, g["y"]) <= 1e-5
    fw = net(x)
    assert rel(sub(fw["x_hat"], 5), g["xhat_fw"]) <= 1e-3
    bits = float((-torch.log2(fw["likelihoods"]["y"].double())).sum())
    assert abs(bits - g["bits_y"][0]) <= 2e-3 * g["bits_y"][0]
    gy = torch.Generator().manual_seed(5)
    y_hat = torch.round(3.0 * torch.randn((1,) + tuple(y.shape), generator=gy))[0].to(dev)
    assert rel(sub(net.g_s(y_hat), 5), g["xhat_synth"]) <= 1e-5
    out = net.compress(x)
    assert list(out["shape"]) == list(g["shape"])
    same = out["strings"][0][0] == g["string0"].tobytes()
    print("factorized-relu: stream identical to the reference python's:", same, len(out["strings"][0][0]), len(g["string0"]))
    assert abs(len(out["strings"][0][0]) - len(g["string0"])) <= 8
    rec = net.decompress(out["strings"], out["shape"])
    assert rel(sub(rec["x_hat"], 5), g["xhat_rt"]) <= (1e-5 if same else 1e-3)


def test_cnn_zoo_entry_errors():
    with pytest.raises(ValueError, match="architecture"):
        cnn.cnn_model("nope", 1)
    with pytest.raises(ValueError, match="quality"):
        cnn.cnn_model("bmshj2018-hyperprior", 9)
    with pytest.raises(RuntimeError, match="Pre-trained"):
        cnn.cnn_model("mbt2018-mean", 3, pretrained=True)
    m = cnn.cnn_model("bmshj2018-factorized", 6)
    assert (m.N, m.M) == (192, 320)
    from cra5_amd import zoo
    h = zoo.bmshj2018_hyperprior(2)
    assert isinstance(h, cnn.ScaleHyperprior) and (h.N, h.M) == (128, 192)
    with pytest.raises(ValueError, match="between"):
        zoo.mbt2018_mean(0)
    with pytest.raises(RuntimeError, match="not yet available"):
        zoo.bmshj2018_factorized(1, pretrained=True)
    r = zoo.bmshj2018_factorized_relu(7)
    assert isinstance(r, cnn.FactorizedPriorReLU) and (r.N, r.M) == (192, 320)

"""Host-side logic of the drop-in surface (no GPU): zoo entry point, state-dict layout,
block pattern, CDF tables, `.bin` container, API plumbing, loud failure without a GPU."""
import io
import json
import struct

import numpy as np
import os

import pytest
import torch

from cra5_amd import binfmt, synth
from cra5_amd.entropy import EntropyBottleneck, GaussianConditional, get_scale_table
from cra5_amd.vaeformer import VAEformer, block_windows, config_for
from cra5_amd.zoo import load_pretrained, rename_key, vaeformer_pretrained


def test_zoo_errors_match_reference():
    """zoo/image.py:315-319, 281-282, 290 of the reference."""
    with pytest.raises(ValueError, match="Invalid metric"):
        vaeformer_pretrained(268, metric="psnr")
    with pytest.raises(ValueError, match="Invalid quality"):
        vaeformer_pretrained(0)
    with pytest.raises(ValueError, match="Invalid quality value"):
        vaeformer_pretrained(3)
    with pytest.raises(RuntimeError, match="Pre-trained model not yet available"):
        vaeformer_pretrained(268, pretrained=True)


def test_rename_key():
    assert rename_key("module.entropy_bottleneck._biases.2") == "entropy_bottleneck._bias2"
    assert rename_key("entropy_bottleneck._matrices.0") == "entropy_bottleneck._matrix0"
    assert rename_key("entropy_bottleneck._factors.3") == "entropy_bottleneck._factor3"
    assert rename_key("g_a.blocks.0.norm1.weight") == "g_a.blocks.0.norm1.weight"
    assert load_pretrained({"module.x": 1}) == {"x": 1}


def test_state_dict_layout_matches_reference(golden_dir):
    keys = json.load(open(f"{golden_dir}/state_keys.json"))
    thin = VAEformer(0, **synth.thin_model_kwargs())
    assert {k: list(v.shape) for k, v in thin.state_dict().items()} == keys["thin"]
    assert list(thin.state_dict()) == list(keys["thin"])  # same ORDER too
    cfg = config_for(268)
    assert (cfg["embed_dim"], cfg["depth"], cfg["num_heads"], cfg["latent_dim"]) == (1024, 24, 16, 256)
    assert (cfg["h_embed_dim"], cfg["h_depth"], cfg["h_num_heads"], cfg["z_dim"]) == (360, 8, 5, 256)


def test_block_pattern():
    """SURVEY.md appendix A1 (probe-verified in the reference)."""
    enc = block_windows(0, 12, 4, [(24, 24), (12, 48), (48, 12)])
    assert enc == [(24, 24), (12, 48), (48, 12), None] * 3
    dec = block_windows(12, 24, 4, [(24, 24), (12, 48), (48, 12)])
    assert dec == [(24, 24), (12, 48), (48, 12), None] * 3
    thin = VAEformer(0, **synth.thin_model_kwargs())
    assert [b.window for b in thin.g_a.blocks] == [(24, 24), (12, 48), (48, 12), None, None]
    assert [b.window for b in thin.g_s.blocks] == [(24, 24), (12, 48), (48, 12), None]
    assert all(b.window is None for b in thin.h_a.blocks) and len(thin.h_s.blocks) == 2


def test_from_state_dict_and_buffer_resize():
    """models/base.py:69-89: `load_state_dict` resizes the empty CDF buffers to the checkpoint's (thin model).
    The `backbone.` prefix / `kl_loss.logvar` / rename_key handling of the full `pretrained=True` route
    (vaeformer.py:168-185, zoo/pretrained.py:36-64) is tested in tests/test_checkpoint_route.py."""
    thin = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(thin, seed=1)
    sd = thin.state_dict()
    other = VAEformer(0, **synth.thin_model_kwargs())
    assert other.gaussian_conditional._quantized_cdf.numel() == 0
    with pytest.raises(ValueError, match="Uninitialized CDFs"):
        other.gaussian_conditional._check()
    other.load_state_dict(sd)
    assert torch.equal(other.gaussian_conditional._quantized_cdf, thin.gaussian_conditional._quantized_cdf)
    assert torch.equal(other.entropy_bottleneck._offset, thin.entropy_bottleneck._offset)


def test_tables_match_reference(golden_dir):
    g = np.load(f"{golden_dir}/tables_default.npz")
    gc = GaussianConditional(None)
    assert gc.update_scale_table(get_scale_table(), force=True)
    assert not gc.update_scale_table(get_scale_table())  # already initialised, no force
    assert np.array_equal(gc._quantized_cdf.numpy(), g["gc_cdf"])
    assert np.array_equal(gc._cdf_length.numpy(), g["gc_len"]) and np.array_equal(gc._offset.numpy(), g["gc_off"])
    thin = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(thin, seed=7)
    eb = thin.entropy_bottleneck
    assert np.array_equal(eb._quantized_cdf.numpy(), g["eb_cdf"])
    assert np.array_equal(eb._cdf_length.numpy(), g["eb_len"]) and np.array_equal(eb._offset.numpy(), g["eb_off"])
    with pytest.raises(ValueError):
        GaussianConditional([3.0, 1.0])
    with pytest.raises(ValueError):
        GaussianConditional("x")


def test_bin_container_byte_layout():
    """cra5_api.py:108-117 / api/utils.py:10-34: big-endian uint32 zH, zW, n, then len+bytes."""
    blob = binfmt.pack_bin([[b"\x01\x02\x03\x04YYYY"], [b"ZZ"]], (18, 36))
    expect = struct.pack(">III", 18, 36, 2) + struct.pack(">I", 8) + b"\x01\x02\x03\x04YYYY" + struct.pack(">I", 2) + b"ZZ"
    assert blob == expect
    strings, shape = binfmt.unpack_bin(blob)
    assert strings == [[b"\x01\x02\x03\x04YYYY"], [b"ZZ"]] and shape == (18, 36)
    g = io.BytesIO()
    binfmt.write_bin(g, [[b"\x01\x02\x03\x04YYYY"], [b"ZZ"]], (18, 36))        # the copy-free writer: the same bytes
    assert g.getvalue() == expect
    f = io.BytesIO()
    assert binfmt.write_uints(f, (1, 2)) == 8 and binfmt.write_bytes(f, b"abc") == 3
    f.seek(0)
    assert binfmt.read_uints(f, 2) == (1, 2) and binfmt.read_bytes(f, 3) == b"abc"


def test_compute_fails_loudly_without_gpu():
    thin = VAEformer(0, **synth.thin_model_kwargs())
    x = torch.zeros(1, 8, 721, 1440)
    for call in (lambda: thin.compress(x), lambda: thin.encode_latent(x), lambda: thin(x),
                 lambda: thin.decode_latent(torch.zeros(1, 16, 72, 144)),
                 lambda: thin.decompress([[b""], [b""]], (18, 36))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    from cra5_amd import ops
    with pytest.raises(RuntimeError, match="non-GPU tensor"):
        ops.gemm_nt(torch.zeros(4, 4), torch.zeros(4, 4))


def test_api_host_plumbing(tmp_path):
    """Channel order, mean/std lookup, normalise round trip (cra5_api.py:228-271)."""
    from cra5_amd.api import cra5_api
    thin = VAEformer(0, **synth.thin_model_kwargs())
    api = cra5_api(local_root=str(tmp_path), device="cpu", weights=thin)
    assert len(api.channels_to_vname) == 268 and api.mean.shape == (268, 1, 1)
    assert api.channels_to_vname[0] == "z_1000" and api.channels_to_vname[36] == "z_1"
    assert api.channels_to_vname[37] == "q_1000" and api.channels_to_vname[259] == "v10"
    assert api.channels_to_vname[266] == "tp" and api.channels_to_vname[267] == "msl"
    assert api.vname_to_channels["t_850"] == 4 * 37 + 6
    x = api.mean + api.std * torch.randn(268, 4, 5)   # physical-units data
    n = api.normalization(x)
    assert abs(float(n.mean())) < 0.2 and 0.8 < float(n.std()) < 1.2
    back = api.de_normalization(n.clone())
    assert torch.allclose(back, x, rtol=1e-5, atol=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        api.encode_to_latent("2024-06-01T00:00:00", data=torch.zeros(268, 721, 1440))


def test_api_stats_equal_reference_dump(tmp_path, golden_dir):
    """mean / std vectors and channel names == what the reference's own get_mean_std /
    channel_vname_mapping return (tests/golden/make_golden.py --stage stats)."""
    import numpy as np
    from cra5_amd.api import cra5_api
    g = np.load(f"{golden_dir}/era5_stats_ref.npz")
    api = cra5_api(local_root=str(tmp_path), device="cpu", weights=VAEformer(0, **synth.thin_model_kwargs()))
    mean, std = api.get_mean_std()
    assert mean.dtype == np.float32 and np.array_equal(mean, g["mean"]) and np.array_equal(std, g["std"])
    assert [api.channels_to_vname[i] for i in range(268)] == list(g["vnames"])
    assert api.level_mapping == list(g["level_mapping"])
    assert all(api.vname_to_channels[n] == i for i, n in enumerate(g["vnames"]))


def test_netcdf_ingest_channel_order(tmp_path):
    """read_data_from_nc (cra5_api.py:195-226) on tiny synthetic NetCDF-3 files: pressure variables x
    levels in cfg.pressure_level order whatever the file's level order, packed (scale/offset) variables
    unpacked, then the singles with tp x 1000."""
    import numpy as np
    from scipy.io import netcdf_file
    from cra5_amd.api import cra5_api
    api = cra5_api(local_root=str(tmp_path), device="cpu", weights=VAEformer(0, **synth.thin_model_kwargs()))
    ts = "2024-06-01T00:00:00"
    d = tmp_path / "ERA5" / "2024"
    d.mkdir(parents=True)
    H, W = 3, 4
    rng = np.random.default_rng(5)
    levels = np.array(api.total_levels, dtype=np.float32)[::-1].copy()   # file order: 1 ... 1000 hPa
    truth = {}
    f = netcdf_file(str(d / f"{ts}_pressure.nc"), "w")
    for name, n in (("time", 1), ("level", 37), ("latitude", H), ("longitude", W)):
        f.createDimension(name, n)
    lv = f.createVariable("level", "f4", ("level",))
    lv[:] = levels
    for k, v in enumerate(api.vnames["pressure"]):
        a = rng.standard_normal((1, 37, H, W)).astype(np.float32) * (k + 1)
        var = f.createVariable(v, "i2" if v == "q" else "f4", ("time", "level", "latitude", "longitude"))
        if v == "q":   # packed like CDS legacy files
            sf, ao = np.float32(1e-3), np.float32(0.5)
            packed = np.rint((a - ao) / sf).astype(np.int16)
            var[:] = packed
            var.scale_factor, var.add_offset = sf, ao
            a = packed.astype(np.float64) * sf + ao
        else:
            var[:] = a
        truth[v] = np.asarray(a, dtype=np.float32)
    f.close()
    f = netcdf_file(str(d / f"{ts}_single.nc"), "w")
    for name, n in (("time", 1), ("latitude", H), ("longitude", W)):
        f.createDimension(name, n)
    for v in api.vnames["single"]:
        a = rng.standard_normal((1, H, W)).astype(np.float32)
        f.createVariable(v, "f4", ("time", "latitude", "longitude"))[:] = a
        truth[v] = a
    f.close()
    x = api.read_data_from_nc(ts)
    assert x.shape == (268, H, W) and x.dtype == np.float32
    for ch in range(268):
        name = api.channels_to_vname[ch]
        if "_" in name and name.split("_")[0] in api.vnames["pressure"]:
            v, lev = name.split("_")
            want = truth[v][0, list(levels).index(float(lev))]
        else:
            want = truth[name][0] * (1000 if name == "tp" else 1)
        assert np.allclose(x[ch], want, rtol=1e-6, atol=1e-7), name
    # a NetCDF-4 (HDF5) file is refused with a clear message when xarray is absent
    try:
        import xarray  # noqa: F401
    except ImportError:
        (d / "2024-06-01T01:00:00_pressure.nc").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
        (d / "2024-06-01T01:00:00_single.nc").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
        with pytest.raises(RuntimeError, match="NetCDF-4"):
            api.read_data_from_nc("2024-06-01T01:00:00")


def test_api_visualisation_helpers(tmp_path):
    """show_image / show_latent (cra5_api.py:273-341): figure files at the reference's paths."""
    import numpy as np
    from cra5_amd.api import cra5_api
    api = cra5_api(local_root=str(tmp_path), device="cpu", weights=VAEformer(0, **synth.thin_model_kwargs()))
    rng = np.random.default_rng(0)
    ori = rng.standard_normal((268, 20, 40)).astype(np.float32)
    rec = ori + 0.01 * rng.standard_normal(ori.shape).astype(np.float32)
    ts = "2024-06-01T00:00:00"
    p = api.show_image(torch.from_numpy(rec).unsqueeze(0), ts, show_variables=['z_500', 't_850'], data=ori)
    assert p == f"{tmp_path}/CRA5_vis/2024/{ts}_reconstruction.png" and os.path.getsize(p) > 1000
    q = api.show_latent(rng.standard_normal((1, 256, 72, 144)).astype(np.float32), ts, save_path=str(tmp_path / "v"))
    assert q.endswith(f"v/{ts}_latent.png") and os.path.getsize(q) > 1000


def test_install_dropin():
    import cra5_amd
    cra5_amd.install_dropin()
    from cra5.api import cra5_api  # noqa: F401
    from cra5.models.compressai.zoo import vaeformer_pretrained as vp
    assert vp is vaeformer_pretrained


def test_synth_is_deterministic():
    a = synth.synth_tensor("g_a.blocks.0.attn.qkv.weight", (6, 4), seed=7)
    b = synth.synth_tensor("g_a.blocks.0.attn.qkv.weight", (6, 4), seed=7)
    c = synth.synth_tensor("g_a.blocks.1.attn.qkv.weight", (6, 4), seed=7)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert synth.synth_tensor("gaussian_conditional._offset", (0,), 7) is None
    assert torch.equal(synth.synth_frame(3, 5, 4, 6), synth.synth_frame(3, 5, 4, 6))


def test_gpu_phase_slot_gate_orders_waiters_by_priority():
    """vaeformer._SlotGate: at most n holders; among the waiters the smaller priority number goes first, FIFO among
    equals (encode-side GPU phases ahead of decode-side ones in the frame pipeline)."""
    import threading
    import time
    from cra5_amd.vaeformer import _SlotGate
    g = _SlotGate(2)
    g.acquire(1)
    g.acquire(1)
    order, lock = [], threading.Lock()

    def worker(prio, name):
        g.acquire(prio)
        with lock:
            order.append(name)
        time.sleep(0.005)
        g.release()
    ts = [threading.Thread(target=worker, args=a) for a in ((2, "d1"), (0, "e1"), (2, "d2"), (1, "m"), (0, "e2"))]
    for t in ts:
        t.start()
        time.sleep(0.02)          # arrival order = list order
    g.release()
    g.release()
    for t in ts:
        t.join(timeout=10)
    assert order[:2] == ["e1", "e2"] and order[2] == "m" and order[3:] == ["d1", "d2"]
    # the gate is reusable and never lets more than n in
    inside, peak = [0], [0]

    def burst():
        g.acquire(0)
        with lock:
            inside[0] += 1
            peak[0] = max(peak[0], inside[0])
        time.sleep(0.002)
        with lock:
            inside[0] -= 1
        g.release()
    ts = [threading.Thread(target=burst) for _ in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=10)
    assert peak[0] == 2


def test_profiles_and_tools_readmes_match_the_directories():
    """Every file under profiles/ is described in profiles/README.md (by name, by a `*` pattern, or as the `.txt` /
    `.json` twin of a named file), every rNN_ file the README names exists, and tools/README.md names every script
    under tools/ and no script that is gone."""
    import fnmatch
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "profiles", "README.md")).read()
    files = sorted(f for f in os.listdir(os.path.join(root, "profiles")) if f != "README.md")
    pats = re.findall(r"`([^`\s]*\*[^`\s]*)`", txt)
    stems = {f.rsplit(".", 1)[0] for f in files if f in txt}
    missing = [f for f in files if f not in txt and not any(fnmatch.fnmatch(f, p) for p in pats)
               and f.rsplit(".", 1)[0] not in stems]
    assert not missing, f"profiles/ files the README does not describe: {missing}"
    named = set(re.findall(r"`((?:r\d\d_)[A-Za-z0-9_.\-]+\.(?:json|csv|txt))`", txt))
    assert not [m for m in named if m not in files], [m for m in named if m not in files]
    t = open(os.path.join(root, "tools", "README.md")).read()
    have = {f for f in os.listdir(os.path.join(root, "tools")) if os.path.isfile(os.path.join(root, "tools", f))}
    have |= {"probes/" + f for f in os.listdir(os.path.join(root, "tools", "probes"))}
    scripts = {f for f in have if f.endswith((".py", ".sh", ".hip")) and not f.startswith("_")}
    assert not [f for f in scripts if f not in t], [f for f in scripts if f not in t]
    table = t.split("| script |", 1)[1]
    listed = set(re.findall(r"`([A-Za-z0-9_/]+\.(?:py|sh|hip))`", table)) - {"bench.py"}
    assert not [f for f in listed if f not in have], [f for f in listed if f not in have]


def test_netcdf4_branch_through_a_stub_xarray(tmp_path, monkeypatch):
    """The xarray branch of cra5_api._open_nc (what the reference itself does: cra5_api.py:203-208,
    `xr.open_dataset(path, engine='netcdf4')`, `ds[name].data`, `ds.close()`) has never run in this image (no xarray, no
    netCDF4).  A stub module with exactly the three members the branch uses makes it EXECUTE: the call signature, the
    channel assembly on what it returns (levels in the file's own order, `tp` x 1000) and the close() calls are pinned;
    the HDF5 decoding itself is xarray's job and stays untested offline."""
    import sys
    import types
    import numpy as np
    from cra5_amd.api import cra5_api
    api = cra5_api(local_root=str(tmp_path), device="cpu", weights=VAEformer(0, **synth.thin_model_kwargs()))
    H, W = 2, 3
    rng = np.random.default_rng(9)
    levels = np.array(api.total_levels, dtype=np.float64)[::-1].copy()
    store, opened, closed = {}, [], []
    for k, v in enumerate(api.vnames["pressure"]):
        store[v] = rng.standard_normal((1, 37, H, W)).astype(np.float32) * (k + 1)
    for v in api.vnames["single"]:
        store[v] = rng.standard_normal((1, H, W)).astype(np.float32)
    store["level"] = levels

    class _Var:
        def __init__(self, a):
            self.data = a

    class _DS:
        def __init__(self, path):
            self.path = path

        def __getitem__(self, name):
            return _Var(store[name])

        def close(self):
            closed.append(self.path)

    def open_dataset(path, engine=None, **kw):
        assert engine == "netcdf4" and not kw          # the reference's call, cra5_api.py:203-204
        opened.append(path)
        return _DS(path)
    monkeypatch.setitem(sys.modules, "xarray", types.SimpleNamespace(open_dataset=open_dataset))
    ts = "2024-06-01T02:00:00"
    x = api.read_data_from_nc(ts)                      # (no file on disk: the stub never touches the path)
    assert opened == [f"{tmp_path}/ERA5/2024/{ts}_pressure.nc", f"{tmp_path}/ERA5/2024/{ts}_single.nc"]
    assert sorted(closed) == sorted(opened)
    assert x.shape == (268, H, W) and x.dtype == np.float32
    for ch in range(268):
        name = api.channels_to_vname[ch]
        if "_" in name and name.split("_")[0] in api.vnames["pressure"]:
            v, lev = name.split("_")
            want = store[v][0, list(levels).index(float(lev))]
        else:
            want = store[name][0] * (1000 if name == "tp" else 1)
        assert np.array_equal(x[ch], want.astype(np.float32)), name


def test_runtime_config_is_the_one_place_settings_come_from():
    """cra5_amd/config.py: defaults < CRA5_* environment < explicit arguments; invalid values are refused at construction;
    a model built with `runtime=` takes every setting from the object and none from the environment."""
    from cra5_amd.config import RuntimeConfig
    d = RuntimeConfig()
    assert (d.precision, d.gemm_engine, d.attn_engine, d.range_guard, d.gpu_slots, d.inflight, d.link_serial) == \
        ("fp32", "split", "split", True, 3, 12, True)
    env = {"CRA5_PRECISION": "f16", "CRA5_GPU_SLOTS": "2", "CRA5_RANGE_GUARD": "0", "CRA5_LINK_SERIAL": "off",
           "CRA5_WEIGHTS": "/x/y.pth", "CRA5_SWITCH_INTERVAL": "0.001", "CRA5_UNKNOWN_SWITCH": "1"}
    e = RuntimeConfig.from_env(env)
    assert (e.precision, e.gpu_slots, e.range_guard, e.link_serial, e.weights, e.switch_interval_s) == \
        ("f16", 2, False, False, "/x/y.pth", 0.001)
    assert RuntimeConfig.from_env(env, gpu_slots=5).gpu_slots == 5
    assert e.replace(precision="fp32").precision == "fp32" and e.precision == "f16"
    for bad in (dict(precision="bf16"), dict(gemm_engine="blas"), dict(attn_engine="x"), dict(inflight=0), dict(copy_threads=65)):
        with pytest.raises(ValueError):
            RuntimeConfig(**bad)
    with pytest.raises(ValueError):
        RuntimeConfig.from_env({"CRA5_GEMM": "cublas"})
    assert set(RuntimeConfig.ENV) == {f for f in RuntimeConfig.__dataclass_fields__}
    assert "set_by_environment" in d.describe() and d.describe()["precision"] == "fp32"
    net = VAEformer(0, runtime=e.replace(gemm_engine="f32"), **synth.thin_model_kwargs())
    assert (net.precision, net.gemm_mode, net.gpu_slots, net.range_guard) == ("f16", "f32", 2, False)
    assert net.runtime.gemm_engine == "f32"

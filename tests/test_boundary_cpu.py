"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly
what include/satmvs.h declares (no compute calls without a GPU), argument errors are reported the
documented way, and the host-side helpers behave like the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from satmvs_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "satmvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smvs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from satmvs_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), "include/satmvs.h declares %s but the library does not export it" % name
    assert declared == _lib.EXPORTED_SYMBOLS, "ctypes table and header disagree"


def test_version_and_error_strings(lib):
    assert lib.smvs_version().decode().startswith("satmvs-hip")
    assert lib.smvs_last_error().decode() == "" or isinstance(lib.smvs_last_error().decode(), str)


def test_argument_errors_do_not_touch_the_gpu(lib):
    """Null pointers / bad ranges are rejected before any HIP call, with a message."""
    from satmvs_amd import _lib
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_rpc_costvol_fwd", None, None, 2, None, None, 1, None, 1, 8, 4, 8, 8, 0, 4, 4, 0, None)
    dummy = C.c_void_p(16)
    arr = (C.c_void_p * 2)(16, 16)
    with pytest.raises(_lib.SatMVSNativeError, match="n_src"):
        _lib.call("smvs_rpc_costvol_fwd", dummy, arr, 9, dummy, dummy, 1, dummy, 1, 8, 4, 8, 8, 0, 4, 4, 0, None)
    with pytest.raises(_lib.SatMVSNativeError, match="plane range"):
        _lib.call("smvs_rpc_costvol_fwd", dummy, arr, 2, dummy, dummy, 1, dummy, 1, 8, 4, 8, 8, 3, 9, 4, 0, None)
    with pytest.raises(_lib.SatMVSNativeError, match="does not fit"):
        _lib.call("smvs_rpc_costvol_fwd", dummy, arr, 2, dummy, dummy, 1, dummy, 1, 8, 4, 8, 8, 0, 4, 2, 0, None)
    with pytest.raises(_lib.SatMVSNativeError, match="dir must be"):
        _lib.call("smvs_rpc_project", dummy, dummy, dummy, dummy, dummy, dummy, 4, 7, None)


def test_regulariser_entry_points_validate_arguments(lib):
    """The widened entry points (RED / CostRegNet / FeatureNet) reject bad shapes and null pointers before any
    HIP call, and their size queries return 0 for shapes they do not serve."""
    from satmvs_amd import _lib
    dummy = C.c_void_p(16)
    arr = (C.c_void_p * 2)(16, 16)
    assert lib.smvs_red_workspace_bytes(1, 8, 30, 40) == 0                 # H not a multiple of 8
    assert lib.smvs_red_workspace_bytes(1, 8, 32, 40) > 0
    assert lib.smvs_red_pred_workspace_bytes(1, 8, 32, 40) > lib.smvs_red_workspace_bytes(1, 8, 32, 40)
    assert lib.smvs_costreg_workspace_bytes(1, 8, 8, 16, 20) == 0          # W not a multiple of 8
    assert lib.smvs_costreg_workspace_bytes(1, 8, 8, 16, 24) > 0
    assert lib.smvs_featnet_workspace_bytes(2, 40, 54, 8, 0) == 0          # W not a multiple of 4
    assert lib.smvs_featnet_workspace_bytes(2, 40, 56, 8, 0) > 0
    assert lib.smvs_featnet_workspace_bytes(2, 40, 56, 8, 1) > lib.smvs_featnet_workspace_bytes(2, 40, 56, 8, 0)
    assert lib.smvs_featnet_workspace_bytes(2, 40, 56, 8, 2) == 0          # unknown arch
    assert lib.smvs_featnet_packed_floats(8, 0) > 0 and lib.smvs_featnet_packed_floats(8, 1) > 0
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_red_step_fwd", None, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 1 << 30, 1, 8, 32, 40, None)
    with pytest.raises(_lib.SatMVSNativeError, match="multiple of 8"):
        _lib.call("smvs_red_step_fwd", dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 1 << 30, 1, 8, 30, 40, None)
    with pytest.raises(_lib.SatMVSNativeError, match="workspace too small"):
        _lib.call("smvs_red_step_fwd", dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 16, 1, 8, 32, 40, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_red_volume_planes", 0, dummy, arr, 2, dummy, dummy, 1, dummy, dummy, dummy, dummy, dummy, None,
                  dummy, 1 << 30, 1, 8, 4, 32, 40, 0, 4, None)
    with pytest.raises(_lib.SatMVSNativeError, match="bad plane range"):
        _lib.call("smvs_red_pred_planes", 0, dummy, arr, 2, dummy, dummy, 1, dummy, dummy, dummy, dummy, dummy, dummy,
                  dummy, 1 << 30, 1, 8, 4, 32, 40, 2, 6, None)
    with pytest.raises(_lib.SatMVSNativeError, match="geo_kind"):
        _lib.call("smvs_red_pred_planes", 3, dummy, arr, 2, dummy, dummy, 1, dummy, dummy, dummy, dummy, dummy, dummy,
                  dummy, 1 << 30, 1, 8, 4, 32, 40, 0, 4, None)
    with pytest.raises(_lib.SatMVSNativeError, match="multiple of 8"):
        _lib.call("smvs_costreg_fwd", dummy, dummy, dummy, dummy, 1 << 30, 1, 8, 8, 16, 20, None)
    with pytest.raises(_lib.SatMVSNativeError, match="arch must be"):
        _lib.call("smvs_featnet_fwd", dummy, dummy, dummy, dummy, dummy, dummy, 1 << 30, 2, 40, 56, 8, 5, None)
    with pytest.raises(_lib.SatMVSNativeError, match="multiple of 4"):
        _lib.call("smvs_featnet_fwd", dummy, dummy, dummy, dummy, dummy, dummy, 1 << 30, 2, 40, 54, 8, 0, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null parameter pointer"):
        _lib.call("smvs_featnet_pack_weights", (C.c_void_p * 63)(), 8, 0, dummy, None)


def test_product_path_has_no_cpu_fallback():
    from satmvs_amd import _lib
    from satmvs_amd.modules import warping
    with pytest.raises(_lib.SatMVSNativeError, match="no CPU fallback"):
        warping.variance_cost_volume([torch.zeros(1, 2, 4, 4)] * 2, torch.zeros(1, 2, 170, dtype=torch.float64),
                                     torch.zeros(1, 2), "rpc")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under satmvs_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "satmvs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


# ---- host helpers -------------------------------------------------------------------------------------
def test_iccv_known_answers(golden):
    """tools/iccv_solver.py:42-64 Test1/Test2 (the reference's only known-answer tests)."""
    from satmvs_amd import rpc_synth
    g = golden("iccv")
    x1, it1 = rpc_synth.iccv_solve(g["A1"], g["L1"])
    np.testing.assert_allclose(x1, [-0.1030, 2.3208, -1.2069, -0.5348], atol=5e-5)
    np.testing.assert_allclose(x1, g["x1"], rtol=1e-12)
    assert it1 == int(g["it1"])
    x2, it2 = rpc_synth.iccv_solve(g["A2"], g["L2"])
    np.testing.assert_allclose(x2, [-1.5, 1.5, -0.5, 0.5], atol=1e-8)
    assert it2 == int(g["it2"])


def test_inverse_fit_matches_reference(golden):
    from satmvs_amd import rpc_synth
    g = golden("iccv")
    rng = np.random.default_rng(0)
    x, y, h = rng.uniform(0, 256, 200), rng.uniform(0, 128, 200), rng.uniform(0, 400, 200)
    for d, full in zip(g["direct"], g["full"]):
        mine, its = rpc_synth.fit_inverse_rpc(d)
        assert its == 1000                                   # the fit always hits the cap
        np.testing.assert_array_equal(mine[:90], full[:90])
        la, lo = rpc_synth.photo2obj(mine, x, y, h)
        la2, lo2 = rpc_synth.photo2obj(full, x, y, h)
        np.testing.assert_allclose(la, la2, rtol=0, atol=1e-11)
        np.testing.assert_allclose(lo, lo2, rtol=0, atol=1e-11)
        assert rpc_synth.roundtrip_error(mine, 256, 128).max() < 5e-4


def test_qc_tensor_roundtrip(golden):
    from satmvs_amd import rpc_synth
    from satmvs_amd.modules.warping import qc_dict_to_rpc
    g = golden("rpc_warp_qc")
    np.testing.assert_array_equal(rpc_synth.coeffs_to_qc_tensor(g["qc_c20"]), g["qc_tensor"])
    rpc = g["rpc"][:, 0]
    keys = ["line_off", "samp_off", "lat_off", "lon_off", "height_off", "line_scale", "samp_scale", "lat_scale",
            "lon_scale", "height_scale"]
    d = {k: torch.from_numpy(np.ascontiguousarray(rpc[:, i])) for i, k in enumerate(keys)}
    for j, nm in enumerate(["line_num", "line_den", "samp_num", "samp_den", "lat_num", "lat_den", "lon_num", "lon_den"]):
        d[nm + "_tensor"] = torch.from_numpy(np.stack([rpc_synth.coeffs_to_qc_tensor(r[10 + 20 * j:30 + 20 * j]) for r in rpc]))
    np.testing.assert_allclose(qc_dict_to_rpc(d).numpy(), rpc, rtol=3e-16, atol=0)


def test_depth_range_golden(golden):
    from satmvs_amd.modules import depth_range
    g = golden("depth_range")
    B, H, W = g["cur"].shape
    s1 = depth_range.get_depth_range_samples(torch.from_numpy(g["dv"]), 8, 10.0, "cpu", torch.float32, [B, H, W])
    s2 = depth_range.get_depth_range_samples(torch.from_numpy(g["cur"]), 6, 5.0, "cpu", torch.float32, [B, H, W])
    np.testing.assert_array_equal(s1.numpy(), g["s1"])
    np.testing.assert_array_equal(s2.numpy(), g["s2"])


def test_plane_range_partition():
    from satmvs_amd.shard import plane_range
    for D in (1, 7, 8, 48, 64):
        for G in (1, 2, 3, 8):
            spans = [plane_range(D, g, G) for g in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == D
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_dropin_packages_resolve_reference_import_lines(tmp_path):
    """dropin/ on PYTHONPATH makes the reference's own import lines (train.py:9-11, networks/casred.py:4-6) resolve to
    the native engine with the reference's names; a stand-in `networks/loss.py` later on the path is still reachable."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake_ref = tmp_path / "ref"
    (fake_ref / "networks").mkdir(parents=True)
    (fake_ref / "networks" / "loss.py").write_text("def cas_mvsnet_loss():\n    return 'reference loss'\n")
    code = (
        "from networks.casred import CascadeREDNet, Infer_CascadeREDNet\n"
        "from networks.casmvs import CascadeMVSNet\n"
        "from networks.ucs import UCSNet\n"
        "from networks.loss import *\n"
        "from modules.warping import *\n"
        "from modules.module import *\n"
        "from modules.depth_range import *\n"
        "import satmvs_amd.networks.casred as n, satmvs_amd.modules.warping as w\n"
        "assert CascadeREDNet is n.CascadeREDNet and rpc_warping is w.rpc_warping and homo_warping is w.homo_warping\n"
        "assert RPC_Photo2Obj is w.RPC_Photo2Obj and callable(get_depth_range_samples) and callable(depth_regression)\n"
        "assert cas_mvsnet_loss() == 'reference loss'\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "dropin"), root, str(fake_ref)]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_data_io_matches_reference_readers(golden, tmp_path):
    """PFM / .rpc readers against what the reference's own load_pfm / load_rpc_as_array returned for the committed files
    (tests/golden/gen_golden.py::gen_io), and writer -> reader round trips incl. colour and big-endian PFMs."""
    from satmvs_amd import data_io
    g = golden("io")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io")
    assert np.array_equal(data_io.load_pfm(os.path.join(here, "height.pfm")), g["pfm"])
    rpc, hmax, hmin = data_io.load_rpc_as_array(os.path.join(here, "view.rpc"))
    assert np.array_equal(rpc, g["rpc"]) and hmax == float(g["h_max"]) and hmin == float(g["h_min"])
    rng = np.random.default_rng(0)
    col = rng.standard_normal((5, 7, 3)).astype(np.float32)
    data_io.save_pfm(str(tmp_path / "c.pfm"), col)
    assert np.array_equal(data_io.load_pfm(str(tmp_path / "c.pfm")), col)
    big = np.flipud(g["pfm"]).astype(">f4")
    with open(tmp_path / "b.pfm", "wb") as f:
        f.write(b"Pf\n%d %d\n1.000000\n" % (big.shape[1], big.shape[0]))
        f.write(big.tobytes())
    assert np.array_equal(data_io.load_pfm(str(tmp_path / "b.pfm")), g["pfm"])
    data_io.save_rpc(str(tmp_path / "r.rpc"), g["rpc"])
    assert np.array_equal(data_io.load_rpc_as_array(str(tmp_path / "r.rpc"))[0], g["rpc"])
    with pytest.raises(Exception):
        data_io.save_pfm(str(tmp_path / "x.pfm"), col.astype(np.float64))


def test_round2_entry_points_validate_arguments(lib):
    """The entry points added in round 2 (generated heights, window regression, consistency filter) reject bad arguments
    before any HIP call; the size queries report the kernels' limits as 0 (callers then take the composite)."""
    from satmvs_amd import _lib
    from satmvs_amd.modules.depth_range import _HeightGenStruct
    dummy = C.c_void_p(16)
    arr = (C.c_void_p * 2)(16, 16)

    def gen(prev=16, hp=8, wp=16, ih=32, iw=64, nd=6, interval=5.0):
        return _HeightGenStruct(prev, hp, wp, ih, iw, nd, interval)
    g = gen()
    with pytest.raises(_lib.SatMVSNativeError, match="ndepth differs"):
        _lib.call("smvs_rpc_costvol_fwd_gen", dummy, arr, 2, dummy, C.addressof(g), dummy, 1, 8, 7, 16, 32, 0, 7, 7, 0, None)
    g4 = gen(ih=64, iw=128)                                   # image / stage = 4: only stage 1, which passes planes
    with pytest.raises(_lib.SatMVSNativeError, match="scale 1 or 2"):
        _lib.call("smvs_height_hypotheses", C.addressof(g4), dummy, 1, 16, 32, None)
    gbad = gen(ih=33)
    with pytest.raises(_lib.SatMVSNativeError, match="integer multiple"):
        _lib.call("smvs_softmax_regress_fwd_gen", dummy, C.addressof(gbad), dummy, dummy, 1, 6, 16, 32, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null"):
        _lib.call("smvs_window_regress_fwd_gen", dummy, None, dummy, dummy, None, 0.0, 1, 6, 16, 32, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_window_regress_fwd", dummy, None, 1, dummy, dummy, None, 0.0, 1, 6, 16, 32, None)
    with pytest.raises(_lib.SatMVSNativeError, match="go together"):
        _lib.call("smvs_rpc_geo_consistency", dummy, dummy, dummy, dummy, 8, 8, 8, 8, 1.0, 2.5, dummy, dummy, dummy, dummy,
                  dummy, None, None)
    assert lib.smvs_red_workspace_bytes(1, 32, 4096, 4096) == 0            # (C+8)*H*W*4 >= 2^31: beyond 32-bit offsets
    assert lib.smvs_costreg_workspace_bytes(1, 8, 64, 8192, 64) == 0        # D*H/4 exceeds one launch grid
    assert lib.smvs_featnet_workspace_bytes(1, 8192, 8192, 8, 0) == 0       # feature map >= 2 GiB


def test_generated_heights_python_composite_matches_reference(golden):
    """GeneratedHeights.materialize() on the CPU is the reference's own op sequence (bilinear resize, samples, trilinear
    resize): identical to the reference-generated tensors; stage_hypotheses picks planes / composite as documented."""
    from satmvs_amd.modules.depth_range import GeneratedHeights, stage1_planes, stage_hypotheses
    g = golden("depth_range")
    _, H, W = g["cur"].shape
    a = GeneratedHeights(torch.from_numpy(g["prev_a"]), 6, 5.0, (H, W), (H // 2, W // 2))
    assert tuple(a.shape) == (1, 6, H // 2, W // 2) and a.dim() == 4
    assert np.array_equal(a.materialize().numpy(), g["r_a"])
    b = GeneratedHeights(torch.from_numpy(g["prev_b"]), 8, 2.5, (H, W), (H, W))
    assert np.array_equal(b.materialize().numpy(), g["r_b"])
    assert GeneratedHeights.supported((H, W), (H // 2, W // 2)) and not GeneratedHeights.supported((H, W), (H // 4, W // 4))
    planes = stage_hypotheses(None, torch.from_numpy(g["dv"]), 8, 10.0, (H, W), (H // 4, W // 4), torch.float32, "cpu", 1)
    assert planes.shape == (1, 8) and np.array_equal(np.broadcast_to(planes.numpy()[:, :, None, None], g["r1"].shape), g["r1"])
    assert torch.equal(planes, stage1_planes(torch.from_numpy(g["dv"]), 8))
    t = stage_hypotheses(torch.from_numpy(g["prev_a"]), None, 6, 5.0, (H, W), (H // 2, W // 2), torch.float32, "cpu", 1)
    assert isinstance(t, torch.Tensor) and np.array_equal(t.numpy(), g["r_a"])          # CPU: the composite, not a generator


def test_round3_training_entry_points_validate_arguments(lib):
    """smvs_groupnorm1_* / smvs_gru_* (training path) and smvs_costvol_bwd reject null pointers, bad activations, strides,
    alignments and sizes before any HIP call."""
    from satmvs_amd import _lib
    d = C.c_void_p(4096)
    odd = C.c_void_p(4100)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_groupnorm1_fwd", None, 64, d, d, 1e-5, 0, d, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="act must be"):
        _lib.call("smvs_groupnorm1_fwd", d, 64, d, d, 1e-5, 3, d, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="batch stride"):
        _lib.call("smvs_groupnorm1_fwd", d, 63, d, d, 1e-5, 1, d, d, d, 2, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="grid limit"):
        _lib.call("smvs_groupnorm1_fwd", d, 64 * 8, d, d, 1e-5, 1, d, d, d, 70000, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="needs the forward output"):
        _lib.call("smvs_groupnorm1_bwd", d, d, 64, None, d, d, 2, d, 64, d, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="batch stride"):
        _lib.call("smvs_groupnorm1_bwd", d, d, 64, d, d, d, 2, d, 8, d, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_groupnorm1_pair_fwd", d, d, d, None, d, 1e-5, 1, d, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_groupnorm1_pair_bwd", d, d, d, d, d, d, 1, d, d, d, None, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_groupnorm1_pair_fwd_mul", d, d, d, d, d, 1e-5, 1, d, d, d, d, None, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_groupnorm1_fwd_blend", d, 64, d, d, 1e-5, 2, d, d, d, None, 64, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="batch stride"):
        _lib.call("smvs_groupnorm1_fwd_blend", d, 64, d, d, 1e-5, 2, d, d, d, d, 63, d, d, 1, 8, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="16-byte aligned"):
        _lib.call("smvs_gru_blend_fwd", d, odd, d, d, 64, None)
    with pytest.raises(_lib.SatMVSNativeError, match="non-positive"):
        _lib.call("smvs_gru_blend_bwd", d, d, d, d, d, d, d, 0, None)
    with pytest.raises(_lib.SatMVSNativeError, match="null pointer"):
        _lib.call("smvs_gru_mul_cat_fwd", d, None, d, d, 1, 8, 8, 64, None)
    with pytest.raises(_lib.SatMVSNativeError, match="bad dimension"):
        _lib.call("smvs_gru_mul_cat_bwd", d, d, d, d, d, 1, 8, 0, 64, None)
    arr = (C.c_void_p * 2)(4096, 4096)
    with pytest.raises(_lib.SatMVSNativeError, match="border-tap encoding"):
        _lib.call("smvs_costvol_bwd", 0, d, d, arr, 2, d, d, 1, d, arr, 1, 1, 1, 40000, 8, None)
    with pytest.raises(_lib.SatMVSNativeError, match="n_src"):
        _lib.call("smvs_costvol_bwd", 0, d, d, arr, 8, d, d, 1, d, arr, 1, 1, 1, 8, 8, None)


def test_dataset_assembler_matches_reference(golden):
    """satmvs_amd.dataset (PNG views, sample lists, get_sample / get_pred_sample: SURVEY 8f-4) against the reference's own
    MVSDataset run on the same scene folder (tests/golden/scene/, gen_golden.py::gen_dataset): images, three-scale RPCs,
    height range, height maps and masks at the three scales -- identical values and dtypes, for ref view 2 (the reference's
    default, including its height-range quirk), ref view 0 and the all-views "pred" list."""
    from satmvs_amd.dataset import MVSDataset
    g = golden("dataset")
    scene = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")
    for mode, ref_view in (("test", 2), ("test", 0), ("pred", 2)):
        ds = MVSDataset(scene, mode, 3, ref_view=ref_view)
        assert len(ds) == int(g["%s%d.len" % (mode, ref_view)])
        for i in range(len(ds)):
            smp = ds[i]
            key = "%s%d.%s.%s" % (mode, ref_view, smp["out_view"], smp["out_name"])
            want = g[key + ".imgs"]
            assert smp["imgs"].dtype == want.dtype == np.float32 and smp["imgs"].shape == want.shape == (3, 3, 32, 64)
            assert np.array_equal(smp["imgs"], want), key
            assert smp["depth_values"].dtype == np.float32 and np.array_equal(smp["depth_values"], g[key + ".depth_values"])
            for st in ("stage1", "stage2", "stage3"):
                assert smp["cam_para"][st].dtype == np.float64 and np.array_equal(smp["cam_para"][st], g[key + ".cam." + st]), (key, st)
                if mode != "pred":
                    assert smp["depth"][st].dtype == np.float32 and np.array_equal(smp["depth"][st], g[key + ".depth." + st]), (key, st)
                    assert smp["mask"][st].dtype == np.float32 and np.array_equal(smp["mask"][st], g[key + ".mask." + st]), (key, st)
            if mode != "pred":
                assert 0.0 < smp["mask"]["stage3"].mean() < 1.0          # the scene's heights leave the range somewhere: the mask is exercised
    # a sample feeds the networks as it is: (V,3,H,W) images, per-stage (V,170) RPCs
    smp = MVSDataset(scene, "pred", 3)[0]
    assert abs(float(smp["imgs"][0, 0].mean())) < 1e-5 and abs(float(smp["imgs"][0, 0].std()) - 1.0) < 1e-3


def test_dataset_qc_samples_match_reference(golden):
    """MVSDataset(use_qc=True) -- get_sample_qc / get_pred_sample_qc over data_io.load_rpc_as_qc_tensor / to_tensor -- against the
    reference's own MVSDataset(use_qc=True) on the same scene folder (tests/golden/dataset_qc.npz, gen_golden.py::gen_dataset_qc):
    per stage a LIST of V dictionaries (ten float64 scalars, eight symmetric (4,4,4) tensors), identical values; the QC tensors fold
    back to the 170-vector the non-QC sample carries (modules.warping.qc_dict_to_rpc), so both forms drive the same kernels."""
    import torch
    from satmvs_amd.dataset import MVSDataset
    from satmvs_amd import data_io
    from satmvs_amd.modules.warping import qc_dict_to_rpc
    g = golden("dataset_qc")
    scene = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")
    scal, tens = data_io._QC_SCALARS, data_io._QC_TENSORS
    for mode, ref_view in (("test", 2), ("test", 0), ("pred", 2)):
        ds = MVSDataset(scene, mode, 3, ref_view=ref_view, use_qc=True)
        plain = MVSDataset(scene, mode, 3, ref_view=ref_view)
        assert len(ds) == int(g["%s%d.len" % (mode, ref_view)])
        for i in range(len(ds)):
            smp = ds[i]
            key = "%s%d.%s.%s" % (mode, ref_view, smp["out_view"], smp["out_name"])
            assert np.array_equal(smp["imgs"], g[key + ".imgs"]) and smp["imgs"].dtype == np.float32
            assert smp["depth_values"].dtype == np.float32 and np.array_equal(smp["depth_values"], g[key + ".depth_values"]), key
            for st in ("stage1", "stage2", "stage3"):
                cams = smp["cam_para"][st]
                assert isinstance(cams, list) and len(cams) == 3 and all(isinstance(c, dict) for c in cams)
                got_s = np.array([[c[k] for k in scal] for c in cams], np.float64)
                got_t = np.stack([np.stack([c[k + "_tensor"] for k in tens]) for c in cams])
                assert np.array_equal(got_s, g[key + ".cam." + st + ".scalars"]), (key, st)
                assert got_t.dtype == np.float64 and np.array_equal(got_t, g[key + ".cam." + st + ".tensors"]), (key, st)
                if mode != "pred":
                    assert np.array_equal(smp["depth"][st], g[key + ".depth." + st]) and np.array_equal(smp["mask"][st], g[key + ".mask." + st])
                # the dictionaries, batched as a DataLoader would, fold back to the 170-vectors of the plain sample
                for v, c in enumerate(cams):
                    batched = {k: torch.as_tensor(np.asarray(val))[None] for k, val in c.items()}
                    back = qc_dict_to_rpc(batched)[0].numpy()
                    want = plain[i]["cam_para"][st][v]
                    assert np.allclose(back, want, rtol=1e-15, atol=0.0), (key, st, v)
    # the stages do not alias each other (the reference deep-copies before dividing the image-side normalisation)
    smp = MVSDataset(scene, "pred", 3, use_qc=True)[0]
    assert smp["cam_para"]["stage1"][0]["line_off"] * 4 == smp["cam_para"]["stage3"][0]["line_off"]
    assert smp["cam_para"]["stage1"][0]["lat_num_tensor"] is not smp["cam_para"]["stage3"][0]["lat_num_tensor"]


def test_train_mode_augments_like_the_reference():
    """mode="train" applies the reference's random_color (preprocess.py:163-178) by default: four enhancement factors drawn with
    np.random.randint in the reference's order, so the same seed gives the same factors; augment=False switches it off."""
    from PIL import ImageEnhance
    from satmvs_amd import dataset as D
    scene = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")
    img = D.read_img(os.path.join(scene, "image", "0", "tile_a.png"))
    np.random.seed(7)
    got = np.asarray(D.image_augment(img))
    np.random.seed(7)
    f = [np.random.randint(1, 301) / 100., np.random.randint(10, 201) / 100., np.random.randint(10, 201) / 100., np.random.randint(0, 301) / 100.]
    want = ImageEnhance.Sharpness(ImageEnhance.Contrast(ImageEnhance.Brightness(ImageEnhance.Color(img).enhance(f[0])).enhance(f[1])).enhance(f[2])).enhance(f[3])
    assert np.array_equal(got, np.asarray(want))
    off = D.MVSDataset(scene, "train", 3, augment=False)
    assert np.array_equal(off[0]["imgs"], D.MVSDataset(scene, "test", 3)[0]["imgs"])
    np.random.seed(3)
    on = D.MVSDataset(scene, "train", 3)[0]["imgs"]
    assert on.shape == off[0]["imgs"].shape and not np.array_equal(on, off[0]["imgs"])
    # seed=: a reproducible augmentation stream of the dataset's own, independent of the global np.random state
    a = D.MVSDataset(scene, "train", 3, seed=11)[0]["imgs"]
    np.random.seed(99)
    b = D.MVSDataset(scene, "train", 3, seed=11)[0]["imgs"]
    c = D.MVSDataset(scene, "train", 3, seed=12)[0]["imgs"]
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, off[0]["imgs"])


def test_arith_scope_is_thread_local_and_matches_the_header():
    """satmvs_amd._lib.arith_scope (the per-call arithmetic of include/satmvs.h, SMVS_CALL_ARITH_*): nesting, restoration, no leak into
    other threads, and the bit values the Python layer ORs into depth_is_4d are the header's."""
    import re
    import threading
    from satmvs_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "satmvs.h")).read()
    vals = {k: int(v, 16) for k, v in re.findall(r"SMVS_CALL_ARITH_(EXACT|FUSED|MASK) = (0x[0-9a-f]+)", hdr)}
    assert vals == {"EXACT": _lib.CALL_ARITH_BITS["exact"], "FUSED": _lib.CALL_ARITH_BITS["fused"], "MASK": 0x300}
    assert re.search(r"int arith;", hdr)                                   # smvs_height_gen.arith
    from satmvs_amd.modules.depth_range import _HeightGenStruct
    assert _HeightGenStruct._fields_[-1][0] == "arith"
    assert _lib.call_arith_bits() == 0
    seen = {}
    with _lib.arith_scope("fused"):
        assert _lib.call_arith_bits() == 0x200
        with _lib.arith_scope("exact"):
            assert _lib.call_arith_bits() == 0x100
            t = threading.Thread(target=lambda: seen.setdefault("other", _lib.call_arith_bits()))
            t.start(); t.join()
        assert _lib.call_arith_bits() == 0x200
    assert _lib.call_arith_bits() == 0 and seen["other"] == 0
    with pytest.raises(ValueError):
        _lib.arith_scope("fast")

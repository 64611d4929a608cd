"""CPU-side checks of the boundary: the C-ABI library loads and exports every declared symbol; the module
mirrors keep the reference's state-dict layout and initialisation; host-side schedule scalars match golden."""
import os
import re

import pytest
import torch

from tests.conftest import ROOT, load_golden
from tests.golden import golden_util as gu


def _built():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _built()
    from pafuse_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pafuse_hip.h")).read()
    declared = set(re.findall(r"\b(pafuse_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pafuse_version()


def test_shipped_library_holds_no_packed_fp32_instructions():
    """The build promise behind the side streams (pafuse_amd/build_flags.py): no v_pk_{add,mul,fma}_f32 anywhere in the
    device code of the library that ships - checked on the binary, not on the flags."""
    import subprocess
    import tempfile
    _built()
    from pafuse_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin/"
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([llvm + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
        subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev])
        isa = subprocess.run([llvm + "llvm-objdump", "-d", dev], capture_output=True, text=True, check=True).stdout
    assert len(re.findall(r"\bv_mfma_f32_32x32x16_bf16\b", isa)) > 100           # the right code object was read
    packed = re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", isa)
    assert not packed, f"{len(packed)} packed-fp32 instructions in libpafuse_hip.so"
    assert b"no packed-fp32 VALU" in _lib.load().pafuse_version()


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors must have the C layout: check against sizes computed by the C compiler."""
    import ctypes, subprocess, tempfile
    from pafuse_amd import _lib
    src = '#include "pafuse_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu\\n",' \
          'sizeof(pafuse_block_weights),sizeof(pafuse_mixste2_weights),sizeof(pafuse_d3dp_config),' \
          'sizeof(pafuse_ddim_step));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [ctypes.sizeof(_lib.BlockWeights), ctypes.sizeof(_lib.MixSTE2Weights),
                     ctypes.sizeof(_lib.D3DPConfig), ctypes.sizeof(_lib.DDIMStep)]


def test_state_dict_layout_matches_reference():
    from __graft_entry__ import make_model
    from tests.golden.state_template import d3dp_template
    model, sd = make_model(2, 2, device="cpu")
    tmpl = d3dp_template()
    got = model.state_dict()
    assert len(got) == 636 and set(got) == set(tmpl)
    for k, v in tmpl.items():
        assert tuple(got[k].shape) == tuple(v.shape) and got[k].dtype == v.dtype, k
    assert gu.sha256_of(sd) == load_golden("g5_d3dp.npz")["sha"].numpy().tobytes()
    # DataParallel checkpoints: module.-prefixed keys load after stripping
    model.load_state_dict({k[7:]: v for k, v in {"module." + k: v for k, v in sd.items()}.items()})


def test_checkpoints_are_interchangeable_with_the_reference(tmp_path):
    """save_state writes the reference's file: 'model_pos' keys = the 636 template keys with DataParallel's
    ``module.`` prefix (main_h3wb.py:699-705,1029 save the wrapper; :252 loads into a wrapper, strictly), next to a
    pickled numpy RandomState; read_checkpoint reads it back under torch >= 2.6 and load_checkpoint accepts the path,
    the dict, and un-prefixed dicts alike."""
    import numpy as np
    from __graft_entry__ import make_model
    from pafuse_amd import h3wb, harness
    from tests.golden.state_template import d3dp_template
    model, sd = make_model(2, 2, device="cpu")
    opt = torch.optim.AdamW(model.parameters(), lr=6e-5)
    for wrapped in (False, True):
        m = torch.nn.DataParallel(model) if wrapped else model
        fname = h3wb.save_state(m, opt, 7, 6e-5, str(tmp_path), random_state=np.random.RandomState(3), tag=f"w{wrapped}")
        with pytest.raises(Exception):                      # the RandomState pickle: refused by the safe loader
            torch.load(fname, map_location="cpu", weights_only=True)
        ckpt = harness.read_checkpoint(fname)
        assert set(ckpt) == {"optimizer", "epoch", "lr", "model_pos", "random_state"} and ckpt["epoch"] == 7
        assert set(ckpt["model_pos"]) == {"module." + k for k in d3dp_template()}
        for src in (fname, ckpt, ckpt["model_pos"], {k[7:]: v for k, v in ckpt["model_pos"].items()}):
            fresh, _ = make_model(2, 2, seed=99, device="cpu")
            harness.load_checkpoint(fresh, src)
            assert gu.sha256_of(fresh.state_dict()) == gu.sha256_of(sd)
    bare = harness.read_checkpoint(h3wb.save_state(model, opt, 1, 6e-5, str(tmp_path), reference_layout=False))
    assert set(bare["model_pos"]) == set(d3dp_template())


def test_schedule_buffers_and_step_scalars_match_golden():
    from __graft_entry__ import make_model
    z = load_golden("g2_schedule.npz")
    for T in (1, 5, 10):
        model, _ = make_model(1, T, device="cpu")
        for k in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_mean_coef2", "posterior_log_variance_clipped"):
            # bit-exact on the host that made the fixture; other hosts' libm may differ in the last place
            assert torch.allclose(getattr(model, k), z["buf." + k], rtol=1e-13, atol=0), k
        assert model.time_pairs() == [tuple(p) for p in z[f"pairs.{T}"].tolist()]
        steps = model.ddim_steps()
        coefs = [[s.sqrt_alpha_next, s.c, s.sigma] for s in steps if not s.last]
        assert torch.allclose(torch.tensor(coefs, dtype=torch.float64).reshape(-1, 3), z[f"coefs.{T}"], rtol=1e-12, atol=0)
        assert steps[len(steps) - 1].last == 1 and steps[0].time == 999


def test_default_init_matches_reference_rng_order():
    """torch.manual_seed(s); MixSTE2(...) must draw the reference's initial weights (same creation order)."""
    import pafuse_amd
    z = load_golden("g8_init.npz")
    torch.manual_seed(123)
    m = pafuse_amd.MixSTE2(num_frame=27, num_joints=42, in_chans=5, embed_dim_ratio=256, depth=2, num_heads=8,
                           drop_path_rate=0.0, is_train=False)
    assert gu.sha256_of(m.state_dict()) == z["sha"].numpy().tobytes()


def test_cpu_call_raises_not_falls_back():
    import pafuse_amd
    from pafuse_amd import _lib
    m = pafuse_amd.MixSTE2(27, 24, 5, 384, 1, 8, drop_path_rate=0.0, is_train=False)
    with pytest.raises(_lib.PafuseError):
        m(torch.zeros(1, 27, 24, 2), torch.zeros(1, 1, 27, 24, 3), torch.zeros(1, dtype=torch.long))


def test_harness_host_logic_matches_reference():
    """clip cutting (G10), flip copy and part centring (G6) of pafuse_amd.harness against reference outputs."""
    from types import SimpleNamespace
    from pafuse_amd import harness
    z = load_golden("g10_clips.npz")
    for n in (10, 27, 54, 60):
        assert torch.equal(harness.cut_clips(z[f"x2.{n}"]), z[f"c2.{n}"]), n
        assert torch.equal(harness.cut_clips(z[f"x3.{n}"][0]), z[f"c3.{n}"]), n
    x2d, x2f = gu.synthetic_inputs_2d(B=2)
    assert torch.equal(harness.flip_2d(x2d, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT), x2f)
    g6 = load_golden("g6_index_ops.npz")
    ds = SimpleNamespace(parts_joint_indices=gu.DATASET_PART_JOINTS, root_indices=gu.ROOT_INDICES)
    assert torch.equal(harness.center_pose_parts(g6["pose"], ds), g6["centred"])
    ds.parts_connection_indices = dict(gu.CONNECTION_INDICES)
    pose = g6["pose"].clone()
    assert torch.equal(harness.wb_pose_from_parts(pose, ds), g6["wb_out"]) and torch.equal(pose, g6["pose"])
    assert len(harness.ACCUMULATORS) == 14


def test_config_tree_and_overrides():
    from pafuse_amd import config
    args = config.load(overrides=["ft2d.num_proposals=20", "ft2d.sampling_timesteps=10", "model.test_time_augmentation=False",
                                  "general.evaluate=pafuse_model.bin", "ft2d.scale=0.5"])
    assert (args.ft2d.num_proposals, args.ft2d.sampling_timesteps, args.ft2d.scale) == (20, 10, 0.5)
    assert args.model.test_time_augmentation is False and args.general.evaluate == "pafuse_model.bin"
    assert args.model.number_of_frames == 27 and args.data.num_kps == 134 and args.data.merge_hands is True
    with pytest.raises(ValueError):
        config.load(overrides=["nonsense"])
    ref_yaml = "/root/reference/config/config.yaml"
    if os.path.exists(ref_yaml):                                 # build container only: defaults = the reference's file
        ref = config.load(ref_yaml)
        for section, values in config.DEFAULTS.items():
            for k, v in values.items():
                assert getattr(getattr(ref, section), k) == v, (section, k)
        # the module mirror constructs from the tree exactly like from the reference's args
        from __graft_entry__ import make_model
        assert make_model(1, 1, device="cpu")[0].frames == ref.model.number_of_frames


def test_torch_custom_ops_are_registered_for_the_hip_device_only():
    """pafuse::* ops exist with fake-tensor shape functions; a CPU tensor has no kernel (no fallback)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from pafuse_amd import torch_ops as to
    for name in ("linear", "layer_norm", "attention", "block", "mixste_eval", "ddim_loop"):
        assert hasattr(torch.ops.pafuse, name)
    assert len(to.mixste_param_names(27, 24, 384, 8, 8)) == 208
    with FakeTensorMode():
        x = torch.empty(10, 384, device="cuda")
        w, b = torch.empty(1152, 384, device="cuda"), torch.empty(1152, device="cuda")
        assert torch.ops.pafuse.linear(x, w, b, True).shape == (10, 1152)
        assert torch.ops.pafuse.attention(torch.empty(54, 1152, device="cuda"), 8, 27, 2).shape == (54, 384)
        noise = torch.empty(2, 1, 3, 27, 134, 3, device="cuda")
        out = torch.ops.pafuse.ddim_loop(torch.empty(1, 27, 134, 2, device="cuda"), torch.empty(1, 27, 134, 2, device="cuda"),
                                         noise, [], [], torch.empty(134, device="cuda"), 8, 8, [999, 499], [0.0] * 10,
                                         True, 1.0)
        assert out.shape == (1, 2, 3, 27, 134, 3)
    with pytest.raises(NotImplementedError):
        torch.ops.pafuse.linear(torch.zeros(2, 32), torch.zeros(32, 32), torch.zeros(32))


def test_h3wb_loader_matches_reference_on_synthetic_files():
    """pafuse_amd.h3wb (npz loader, root-joint insertion, camera normalisation, mm->m, screen normalisation, fetch)
    against the reference's loader on the synthetic H3WB files of tests/golden/h3wb_synth (golden G14), bit for bit."""
    import numpy as np
    from pafuse_amd import h3wb
    z = load_golden("g14_h3wb_loader.npz")
    ds = h3wb.Human3WBDataset(os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    assert ds.num_kps == int(z["num_kps"]) == 134
    assert ds.skeleton().joints_left() == z["joints_left"].tolist()
    assert ds.skeleton().joints_right() == z["joints_right"].tolist()
    assert ds.keypoints_metadata["keypoints_symmetry"][0] == z["kps_left"].tolist()
    assert list(ds.skeleton().parents()) == z["parents"].tolist()
    assert ds.root_indices == gu.ROOT_INDICES and ds.parts_connection_indices == gu.CONNECTION_INDICES
    for part, idx in ds.parts_joint_indices.items():
        assert idx == z["part." + part].tolist() == list(gu.DATASET_PART_JOINTS[part])
    for subject in ("S1", "S8"):
        for i, cam in enumerate(ds.cameras()[subject]):
            assert np.array_equal(cam["intrinsic"], z[f"cam.{subject}.{i}.intrinsic"].numpy())
            assert np.array_equal(cam["translation"], z[f"cam.{subject}.{i}.translation"].numpy())
    assert np.array_equal(ds["S1"]["Walking 1"]["positions"], z["positions.S1.Walking 1"].numpy())
    keypoints = h3wb.prepare_keypoints(ds)
    for tag, (subjects, stride, filt) in {"test": (["S8"], 1, None), "train2": (["S1", "S5"], 2, ["Dir"])}.items():
        cams, p3, p2 = h3wb.fetch(subjects, keypoints, ds, stride, filt)
        assert len(p2) == int(z[f"fetch.{tag}.n"]) == len(cams) == len(p3)
        for i in range(len(p2)):
            assert np.array_equal(cams[i], z[f"fetch.{tag}.{i}.cam"].numpy())
            assert np.array_equal(p3[i], z[f"fetch.{tag}.{i}.p3"].numpy())
            assert np.array_equal(p2[i], z[f"fetch.{tag}.{i}.p2"].numpy())
    seqs = list(h3wb.iter_sequences(cams, p3, p2))
    assert len(seqs) == len(p2) and seqs[0][0].shape == (1, 9) and seqs[0][2].shape == (1,) + p2[0].shape


def test_in_the_wild_host_helpers(tmp_path):
    """OpenPifPaf JSON reader, clip stitching and the camera-to-world rotation of the in-the-wild caller
    (in_the_wild/h3wb_diffusion.py:57-70,118-140)."""
    import json
    import numpy as np
    from pafuse_amd import harness
    rng = np.random.default_rng(3)
    flat = rng.uniform(0, 1000, (5, 133 * 3)).astype(np.float32)
    path = tmp_path / "clip.openpifpaf.json"
    with open(path, "w") as f:
        for row in flat:
            f.write(json.dumps({"predictions": [{"keypoints": row.tolist()}, {"keypoints": [0.0] * 399}]}) + "\n")
    kps = harness.load_pifpaf_keypoints(str(path))
    assert kps.shape == (5, 134, 2) and kps.dtype == np.float32
    assert np.array_equal(kps[:, 1:, 0], flat[:, ::3]) and np.array_equal(kps[:, 1:, 1], flat[:, 1::3])
    assert np.array_equal(kps[:, 0], (kps[:, 12] + kps[:, 13]) / 2.)
    # stitching inverts cut_clips for every length class
    for n in (10, 27, 54, 60):
        seq = torch.arange(n, dtype=torch.float32).reshape(n, 1, 1).expand(n, 4, 3)
        clips = harness.cut_clips(seq)[:, None, None]                       # [clips, T=1, P=1, 27, J, 3]
        assert torch.equal(harness.stitch_clips(clips, n)[0, 0], seq), n
    z = load_golden("g15_camera_to_world.npz")
    out = harness.camera_to_world(z["X"], z["rot"])
    assert torch.allclose(out, z["out"], rtol=0, atol=1e-6)


def test_training_clip_generator_matches_reference():
    """pafuse_amd.h3wb.ChunkedClips against the reference's ChunkedGenerator_Seq (golden G16): item count, the
    RandomState(1234) shuffles of two consecutive epochs, edge padding, flip augmentation of 2-D, 3-D and camera."""
    import numpy as np
    from pafuse_amd import h3wb
    z = load_golden("g16_chunked_generator.npz")
    ds = h3wb.Human3WBDataset(os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    keypoints = h3wb.prepare_keypoints(ds)
    kl, kr = ds.keypoints_metadata["keypoints_symmetry"]
    cams, p3, p2 = h3wb.fetch(["S1", "S5"], keypoints, ds)
    gen = h3wb.ChunkedClips(3, cams, p3, p2, 27, shuffle=True, augment=True, kps_left=kl, kps_right=kr,
                            joints_left=list(ds.skeleton().joints_left()), joints_right=list(ds.skeleton().joints_right()))
    assert gen.batch_num() == int(z["num_batches"]) and len(gen.pairs) == int(z["num_pairs"])
    for epoch in range(2):
        for b, (cam, b3, b2) in enumerate(gen.next_epoch()):
            if b < 2:
                assert np.array_equal(cam, z[f"e{epoch}.b{b}.cam"].numpy())
                assert np.array_equal(b3, z[f"e{epoch}.b{b}.p3"].numpy())
                assert np.array_equal(b2, z[f"e{epoch}.b{b}.p2"].numpy())


def test_rank_sharded_clip_generator_covers_the_reference_batches():
    """one process per GPU: every rank serves its slice of the reference's GLOBAL batches (same seed, same shuffle), so
    an epoch stays one pass over the data and the ranks' clips together are exactly the unsharded batch (golden G16
    pins that one to the reference).  World size 5 > the 3-clip batch also exercises the empty-share case."""
    import numpy as np
    from pafuse_amd import h3wb
    ds = h3wb.Human3WBDataset(os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    keypoints = h3wb.prepare_keypoints(ds)
    kl, kr = ds.keypoints_metadata["keypoints_symmetry"]
    cams, p3, p2 = h3wb.fetch(["S1", "S5"], keypoints, ds)
    kw = dict(shuffle=True, augment=True, kps_left=kl, kps_right=kr, joints_left=list(ds.skeleton().joints_left()),
              joints_right=list(ds.skeleton().joints_right()))
    for world in (2, 5):
        whole = h3wb.ChunkedClips(3, cams, p3, p2, 27, **kw)
        ranks = [h3wb.ChunkedClips(3, cams, p3, p2, 27, shard=(r, world), **kw) for r in range(world)]
        assert all(g.batch_num() == whole.batch_num() for g in ranks)
        its = [g.next_epoch() for g in ranks]
        for cam, b3, b2 in whole.next_epoch():
            parts = [next(it) for it in its]
            shares = [g.last_share for g in ranks]
            assert all(s[1] == len(b2) and s[2] == world for s in shares) and sum(s[0] for s in shares) == len(b2)
            for r, ((c_r, b3_r, b2_r), (n_r, _, _)) in enumerate(zip(parts, shares)):
                want = slice(r, None, world)
                if n_r == 0:                    # padded with the batch's first clip, weight 0 in train_epoch
                    assert len(b2_r) == 1 and np.array_equal(b2_r[0], b2[0])
                else:
                    assert np.array_equal(b2_r, b2[want]) and np.array_equal(b3_r, b3[want]) and np.array_equal(c_r, cam[want])


def test_training_loss_variants_match_the_reference_formulas():
    """model.mse_loss / model.weighted_loss (common/loss.py:9-27, main_h3wb.py:724-727)."""
    from pafuse_amd import h3wb
    g = torch.Generator().manual_seed(0)
    p, t = torch.randn(2, 27, 134, 3, generator=g), torch.randn(2, 27, 134, 3, generator=g)
    w = torch.tensor(list(h3wb.WEIGHTED_LOSS_HEAD) + [1.0] * 116)
    d = (p - t).pow(2).sum(-1).sqrt()
    assert w.shape == (134,) and float(w.sum()) == 116 + 6 + 3 + 16 + 2 + 10
    assert torch.allclose(h3wb.mpjpe_loss(p, t), d.mean())
    assert torch.allclose(h3wb.mpjpe_loss(p, t, mse_loss=True), d.pow(2).mean())
    assert torch.allclose(h3wb.mpjpe_loss(p, t, w), (w * d).mean())
    assert torch.allclose(h3wb.mpjpe_loss(p, t, w, True), (w * d).pow(2).mean())


def test_evaluation_log_lines_are_the_reference_s():
    """harness.format_report against lines produced by the reference's own print / write statements
    (main_h3wb.py:406-509, restated here as data: the format strings and their order are the contract)."""
    from pafuse_amd import harness
    rep = {k: [float(i + 1) + 0.125 * j for j in range(2)] for i, k in enumerate(harness.ACCUMULATORS)}
    printed, written = harness.format_report(rep, True, action="Walking")
    assert printed[0] == written[0] == "----Walking----" and printed[1] == "Test time augmentation: True"
    assert printed[2] == "step 0 : Protocol #1 Error (MPJPE) J_Best: 1.000000 mm"
    assert printed[3] == "step 0 : Protocol #1 Error (MPJPE) P_Best: 2.000000 mm" and printed[3] not in written
    assert printed[4] == "step 0 : Protocol #1 Error (MPJPE) P_Agg: 3.000000 mm"
    assert printed[5] == "step 0 : Protocol #1 Error (MPJPE) J_Agg: 4.000000 mm"
    assert printed[6] == "-----------------> Part-Based Evaluation <-----------------" and written.count(printed[6]) == 4
    assert "step 0 : Protocol #1 Error (MPJPE) P_Best Part-Based HANDS: 8.500000 mm" in written      # (9 + 8) / 2
    assert "step 1 : Protocol #1 Error (MPJPE) P_Agg Part-Based RIGHT HAND: 14.125000 mm" in written
    assert printed[-1] == written[-1] == "----------" and len(printed) == 2 + 2 * 18 + 1
    p2, w2 = harness.format_report(rep, False)
    assert p2[0] == "----------" and w2[0].startswith("step 0") and p2[1] == "Test time augmentation: False"


def test_bench_parity_object_on_known_differences():
    """bench.py's `parity` object (the HIP default against the oracle on the CPU baseline's sample) computed on
    tensors with a KNOWN difference: identical runs give zeros and all pairs inside 1e-4 mm; a uniform shift of one hypothesis by
    1e-6 m moves P-Agg by at most 1e-3 mm / P and shows in max_abs_m; the object names its yardstick and its sizes."""
    import bench
    from oracle import d3dp_oracle as orc
    from pafuse_amd import synthetic as gu
    g = torch.Generator().manual_seed(3)
    ref = orc.center_pose_parts(torch.randn(1, 2, 3, 27, 134, 3, generator=g) * 0.25)
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    same = bench.parity_object(orc, ref.clone(), ref, gu, x2d, 3, 2)
    assert same["max_abs_m"] == 0.0 and same["pairs_within_1e-4_mm"] == "8 of 8" and same["j_agg_different_picks"] == 0.0
    assert all(v == 0.0 for v in same["dMPJPE_mm"].values()) and same["P"] == 3 and same["T"] == 2 and "oracle" in same["vs"]
    out = ref.clone()
    out[:, :, 1] += 1e-6
    moved = bench.parity_object(orc, out, ref, gu, x2d, 3, 2)
    assert abs(moved["max_abs_m"] - 1e-6) < 2e-7
    assert 0.0 < moved["dMPJPE_mm"]["P-Agg"] <= 1e-3 / 3 + 1e-9 and set(moved["dMPJPE_mm"]) == {"J-Best", "P-Best", "P-Agg", "J-Agg"}

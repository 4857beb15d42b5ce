"""CPU: host-side logic of the product (no kernels run): the C-ABI library loads and exports every
symbol include/msau_hip.h declares, parameter naming/shapes mirror the reference, plans for every
BASELINE configuration fit the LDS / register geometry, SAME-padding arithmetic."""
import os
import re

import numpy as np
import pytest
import torch

from msau_amd import _lib as L
from msau_amd.model import MSAUWrapper, param_shapes
from msau_amd.plan import ConvOp, Plan, same_pads
from oracle import msau_oracle as O
from tests.golden_util import NET_CASES, load_net_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "msau_hip.h")).read()
    declared = set(re.findall(r"\b(msau_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/msau_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert lib.msau_version() >= 1


def test_library_carries_the_hash_of_the_kernel_sources_it_was_built_from(tmp_path, monkeypatch):
    """The .so ships with the tree but is not in its history: the only tie between the two is the stamp build() links in
    (msau_source_hash) -- load() refuses a library of other sources, build() rebuilds one whatever the file times say."""
    from msau_amd import build as B
    lib = L.load()
    assert lib.msau_version() >= 9
    assert lib.msau_source_hash().decode() == B.source_hash() == B.stamped_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", B.source_hash())
    # a library whose stamp names other sources is refused (the check of load(), on a doctored copy of the bytes)
    blob = open(B.LIB, "rb").read()
    i = blob.find(B.STAMP_MARK) + len(B.STAMP_MARK)
    bad = tmp_path / "libmsau_hip.so"
    bad.write_bytes(blob[:i] + b"0123456789abcdef" + blob[i + 16:])
    assert B.stamped_hash(str(bad)) == "0123456789abcdef"
    code = ("import msau_amd._lib as L\nL.LIB_PATH = %r\n"
            "try:\n    L.load()\nexcept L.MsauHipError as e:\n    print('refused:', e)\n" % str(bad))
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k != "MSAU_HIP_LIB"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert "refused:" in out.stdout and "0123456789abcdef" in out.stdout, (out.stdout, out.stderr[-500:])


def test_error_reporting_across_the_abi():
    g = L.ConvPackGeom()
    import ctypes as C
    rc = L.load().msau_conv_pack_geometry(L.F32, 7, 0, 8, 3, 3, 1, 1, 1, C.byref(g))
    assert rc == -1 and b"multiples of 8" in L.load().msau_last_error() or b"channels" in L.load().msau_last_error()
    with pytest.raises(L.MsauHipError):
        L.call("msau_conv_pack_geometry", 5, 8, 0, 8, 3, 3, 1, 1, 1, C.byref(g))


def test_same_pads_matches_reference_pad_2d():
    assert same_pads(9, 4) == (1, 2) and same_pads(336, 4) == (1, 2)
    assert same_pads(42, 3, 1, 8) == (8, 8)
    assert same_pads(7, 2, 2) == (0, 1) and same_pads(8, 2, 2) == (0, 0)
    for n in range(1, 40):
        for k, d in ((3, 1), (3, 2), (3, 8), (4, 1), (1, 1)):
            assert same_pads(n, k, 1, d) == O.same_pads(n, k, 1, d)


@pytest.mark.parametrize("cfg", [dict(O.DEFAULT_CFG), dict(O.DEFAULT_CFG, channels=768, num_blocks=2),
                                 dict(O.DEFAULT_CFG, channels=32, num_blocks=1),
                                 dict(O.DEFAULT_CFG, channels=13, featRoot=4, res_depth=3, scale_space_num=3)])
def test_param_shapes_match_oracle_and_reference_order(cfg):
    assert list(param_shapes(cfg).items()) == list(O.param_shapes(cfg).items())


def test_state_dict_keys_are_the_references():
    g, cfg, sd, x, label = load_net_case("net_f8_c13_33x26")
    m = MSAUWrapper(13, 5, dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax"))
    keys = list(m.state_dict().keys())
    assert keys == [str(s) for s in g["param_names"]]          # names + order recorded from the reference
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())
    m.load_state_dict(sd)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    # parameters are views of one flat fp32 buffer
    assert all(p.data_ptr() >= m.flat_parameters.data_ptr() for p in m.parameters())
    n636 = MSAUWrapper(64, 5, dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax"))
    assert sum(p.numel() for p in n636.parameters()) == 636167 and len(n636.state_dict()) == 196


def test_save_load_roundtrip(tmp_path):
    m = MSAUWrapper(13, 5, dict(scale_space_num=3, res_depth=2, featRoot=8, final_act="softmax", seed=3))
    m.save(str(tmp_path / "w.pth"))
    m2 = MSAUWrapper(13, 5, dict(scale_space_num=3, res_depth=2, featRoot=8, final_act="softmax", seed=4))
    assert not torch.equal(m.flat_parameters, m2.flat_parameters)
    m2.load_weights(str(tmp_path / "w.pth"))
    assert torch.equal(m.flat_parameters, m2.flat_parameters)


def test_init_statistics_follow_reference_formulae():
    m = MSAUWrapper(64, 5, dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax", seed=0))
    sd = m.state_dict()
    w = sd["msau_net.blocks.0.downsamplingblock.conv_res_list.3.conv_res_list.0.custom_conv.weight"]
    assert abs(float(w.std()) - np.sqrt(2.0 / (9 * 64 + 64))) < 0.1 * np.sqrt(2.0 / (9 * 64 + 64))
    b = sd["msau_net.blocks.0.downsamplingblock.conv1s.0.conv.bias"]
    assert abs(float(b.mean()) - 0.1) < 1e-3


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
@pytest.mark.parametrize("case", [
    dict(channels=32, num_blocks=1, B=2, H=128, W=128),          # BASELINE cfg 1
    dict(channels=64, num_blocks=3, B=2, H=336, W=256),          # cfg 2 / 3 geometry
    dict(channels=768, num_blocks=2, B=1, H=336, W=256),         # cfg 4
    dict(channels=64, num_blocks=3, B=1, H=512, W=384),          # cfg 5 geometry (plain blocks)
    dict(channels=13, num_blocks=3, B=1, H=33, W=26),
])
def test_plans_build_for_baseline_configs(case, dtype):
    """every conv / wgrad launch of every BASELINE configuration has a tile that fits LDS and registers"""
    cfg = dict(O.DEFAULT_CFG, channels=case["channels"], num_blocks=case["num_blocks"])
    shapes = param_shapes(cfg)
    poff, off = {}, 0
    for k, s in shapes.items():
        poff[k] = off
        off += -(-int(np.prod(s)) // 4) * 4
    plan = Plan(cfg, case["B"], case["H"], case["W"], dtype, torch.device("cpu"), poff, dict(shapes), training=True)
    convs = [op for op in plan.ops if isinstance(op, ConvOp)]
    n_conv = sum(1 for op in convs if op.kind == "conv")
    n_deconv = sum(1 for op in convs if op.kind == "deconv")
    if case["num_blocks"] == 3:
        # 54 + 9 + 23 + 3 convs and 9 deconvs, minus the dead last-stage attention's 3 projections
        assert n_conv == 54 + 9 + 23 + 3 - 3 and n_deconv == 9
    for op in convs:
        assert op.fdesc is not None and (op.wdesc is not None or not plan.training)
    # the two stage outputs that carry the loss
    assert plan.logits.C == 5 and (plan.aux is not None) == (case["num_blocks"] >= 2)


def test_dead_parameters_are_the_last_stage_attention():
    m = MSAUWrapper(64, 5, dict(scale_space_num=4, res_depth=2, featRoot=8, final_act="softmax"))
    assert len(m._dead) == 6 and sum(int(np.prod(m._pshape[k])) for k in m._dead) == 5200


def test_activation_name_relu_and_elu_construct_anything_else_is_refused():
    """model/model.py:412-416 knows "relu" and "elu" (any other name leaves `self.activation` unset and the reference's own constructor
    dies two lines later): both construct here with the reference's state_dict keys, anything else is a ValueError; the box variant's
    kernels are ReLU only and say so."""
    from msau_amd.model_box import BMSAUWrapper
    kw = dict(scale_space_num=3, res_depth=2, featRoot=8, final_act="softmax")
    keys = list(MSAUWrapper(13, 5, dict(kw)).state_dict())
    assert list(MSAUWrapper(13, 5, dict(kw, activation_name="elu")).state_dict()) == keys
    with pytest.raises(ValueError, match="activation_name"):
        MSAUWrapper(13, 5, dict(kw, activation_name="tanh"))
    with pytest.raises(NotImplementedError, match="relu"):
        BMSAUWrapper(13, 5, dict(kw, activation_name="elu"))


def test_compute_without_gpu_raises_not_falls_back():
    m = MSAUWrapper(13, 5, dict(scale_space_num=3, res_depth=2, featRoot=8, final_act="softmax"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 13, 16, 16))


def test_roofline_tool_known_answers():
    """tools/roofline.py re-derives SURVEY.md 8(d)'s per-tile work from the architecture; these are its known answers."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("roofline", os.path.join(os.path.dirname(__file__), "..", "tools", "roofline.py"))
    RF = importlib.util.module_from_spec(spec)
    import sys
    sys.modules["roofline"] = RF
    spec.loader.exec_module(RF)
    w = RF.work(RF.CONFIGS["cfg2"])
    assert (w["convs"], w["deconvs"], w["params"]) == (98, 9, 636167)
    assert round(w["conv_fwd_flops"] / 1e9, 3) == 8.434
    assert round(w["attn_live_flops"] / 1e9, 3) == 0.520
    assert round(w["first_conv_flops"] / 1e9, 3) == 0.793
    assert round(w["train_flops"] / 1e9, 2) == 26.07
    assert round(w["act_elems"] / 1e6, 2) == 76.75 and round(w["bytes_train"] / 1e6) == 460
    assert w["bound"] == "hbm"
    w1, w4 = RF.work(RF.CONFIGS["cfg1"]), RF.work(RF.CONFIGS["cfg4"])
    assert round(w1["train_flops"] / 1e9, 2) == 1.52 and round(w1["act_elems"] / 1e6, 2) == 4.04
    assert round(w4["train_flops"] / 1e9, 2) == 34.87 and round(w4["act_elems"] / 1e6, 1) == 110.9
    # the parameter count of the walk agrees with the model's own table
    cfg = dict(channels=64, n_class=5, featRoot=8, scale_space_num=4, res_depth=2, filter_size=3, num_blocks=3)
    assert sum(int(np.prod(s)) for s in param_shapes(cfg).values()) == w["params"]


def test_constructor_defaults_are_the_references_and_construct():
    """the reference's constructor defaults (model/model.py:406-408: 6 scales -> 256 channels, res_depth 3, featRoot 8) are
    inside the kernels' envelope since round 2; beyond 256 channels it fails loud and early"""
    m = MSAUWrapper(4, 3)
    assert (m.scale_space_num, m.res_depth, m.featRoot) == (6, 3, 8)
    assert tuple(m.state_dict()["msau_net.blocks.0.downsamplingblock.conv1s.5.conv.weight"].shape) == (256, 128, 3, 3)
    with pytest.raises(NotImplementedError, match="support up to 256"):
        MSAUWrapper(4, 3, dict(scale_space_num=7))


def test_abi_struct_mirrors_have_the_librarys_sizes():
    """every struct that crosses the ABI by pointer: ctypes mirror size == sizeof() in the compiled library"""
    lib = L.load()
    import ctypes
    for which, st in enumerate(L.ABI_STRUCTS):
        assert lib.msau_sizeof(which) == ctypes.sizeof(st), st.__name__
    assert lib.msau_sizeof(len(L.ABI_STRUCTS)) == -1


def test_integration_md_stub_matches_the_header():
    """The ctypes stub printed in INTEGRATION.md is executed: its msau_conv_desc must have the library's size and the
    field order of msau_amd._lib.ConvDesc (round 1 shipped a stub 16 bytes short of the struct the library copies)."""
    import ctypes, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"class msau_conv_desc\(C\.Structure\):.*?\n\n", text, re.S)
    assert m, "INTEGRATION.md no longer contains the msau_conv_desc stub"
    ns = {"C": ctypes}
    exec(m.group(0), ns)
    stub = ns["msau_conv_desc"]
    assert ctypes.sizeof(stub) == L.load().msau_sizeof(0) == ctypes.sizeof(L.ConvDesc)
    assert [f[0] for f in stub._fields_] == [f[0] for f in L.ConvDesc._fields_]
    assert [ctypes.sizeof(f[1]) for f in stub._fields_] == [ctypes.sizeof(f[1]) for f in L.ConvDesc._fields_]


def test_reference_written_checkpoint_loads(tmp_path):
    """A dict checkpoint written by the reference's own utils.io_utils.save_checkpoint (tests/golden/train, generator:
    oracle/gen_goldens.py::train_goldens) loads into MSAUWrapper: keys epoch / model_type / optimizer / model_state /
    optimizer_state / cg, the pickled Adam object included (io_utils.py:83-105)."""
    import ast
    from msau_amd.training import create_filename, load_checkpoint
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train")
    meta = np.load(os.path.join(gdir, "ref_checkpoint_meta.npz"), allow_pickle=True)
    cfg = ast.literal_eval(str(meta["cfg"]))
    m = MSAUWrapper(cfg["channels"], cfg["n_class"], dict(scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"],
                                                         featRoot=cfg["featRoot"], final_act="softmax"))
    # the reference pickles its optimizer OBJECT: refused unless the caller vouches for the file
    with pytest.raises(RuntimeError, match="trusted=True"):
        load_checkpoint(os.path.join(gdir, "ref_checkpoint.pth.tar"), model=m, map_location="cpu")
    ckpt = load_checkpoint(os.path.join(gdir, "ref_checkpoint.pth.tar"), model=m, map_location="cpu", trusted=True)
    assert set(ckpt) == {"epoch", "model_type", "optimizer", "model_state", "optimizer_state", "cg"}
    assert ckpt["epoch"] == 7 and ckpt["model_type"] == "msau" and ckpt["cg"] is None
    assert isinstance(ckpt["optimizer"], torch.optim.Adam)
    cs = float(sum(float(v.double().abs().sum()) for v in m.state_dict().values()))
    assert abs(cs - float(meta["state_checksum"])) <= 1e-9 * cs
    assert list(m.state_dict()) == list(ckpt["model_state"])
    # the file name rule of io_utils.py:65-80
    import types
    args = types.SimpleNamespace(ckptdir=str(tmp_path), bmname=None, dataset="funsd", method="msau", hidden_dim=20, output_dim=20)
    assert os.path.relpath(create_filename(args.ckptdir, args, False, num_epochs=7), args.ckptdir) == str(meta["rel_path"])
    assert create_filename(args.ckptdir, args, False, num_epochs=0).endswith("funsd_msau_h20_o20.pth.tar")
    assert create_filename(args.ckptdir, args, True).endswith(os.path.join("funsd_msau_h20_o20", "best.pth.tar"))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference (build container only)")
def test_golden_recipe_reproduces_committed_fixtures(tmp_path):
    """oracle/gen_goldens.py re-run against the reference reproduces ops.npz, the FUNSD fixtures (hash seed pinned in a
    child interpreter) and the training-side fixtures bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub in ("train", "test"):
        os.makedirs(tmp_path / "funsd" / sub)
    env = dict(os.environ, MSAU_GOLDEN_OUT=str(tmp_path), MSAU_GOLDEN_NETS="0", MSAU_GOLDEN_KV="0", PYTHONDONTWRITEBYTECODE="1",
               PYTHONHASHSEED="random")
    subprocess.run([sys.executable, os.path.join(root, "oracle", "gen_goldens.py")], env=env, check=True, cwd=str(tmp_path),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    gold = os.path.join(root, "tests", "golden")
    for rel in ("ops.npz", os.path.join("funsd", "chargrid.npz"), os.path.join("funsd", "bertgrid.npz"),
                os.path.join("train", "unet_loss.npz"), os.path.join("train", "unet_loss_weighted.npz")):
        a = np.load(os.path.join(gold, rel), allow_pickle=True)
        b = np.load(os.path.join(str(tmp_path), rel), allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), rel
        for k in a.files:
            if a[k].dtype == object or a[k].dtype.kind in "US":
                assert str(a[k]) == str(b[k]), (rel, k)
            else:
                assert np.array_equal(a[k], b[k], equal_nan=True), (rel, k)


def test_isa_of_the_built_kernels_has_no_cross_half_packed_fp32_adds(tmp_path):
    """Round 2 traced run-to-run deviations of the level-0 LRN backward (beside the side stream) to its chains of
    v_pk_add_f32 with op_sel -- packed fp32 instructions that take an operand from the OTHER half of a register pair -- and
    builds elementwise.hip without packed-fp32 instructions (msau_amd/build.py; the listing of the
    deviating build: profiles/r03_lrn_bwd_c8_packed_isa.s).  This test reads the ISA of the library as built and keeps the
    pattern from coming back unnoticed:
      * elementwise.o has NO packed-fp32 instruction at all (the flag is in force);
      * the translation units of the bf16 train path that run on the main stream beside the weight gradients (conv, conv_lean,
        conv_pair, conv_rows, wgrad_lean, conv_wgrad, attention_mfma, pack, raster, ownerconv, boxconv) have NO packed-fp32
        instruction with op_sel / op_sel_hi;
      * attention.hip (the fp32-storage VALU attention) does use them -- known, counted, outside the bf16 path."""
    import shutil
    import subprocess
    from msau_amd import build as B
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    B.build(verbose=False)
    pk = re.compile(r"v_pk_(add|mul|fma)_f32")
    counts = {}
    for src in B.SOURCES:
        name = src.replace(".hip", "")
        obj = tmp_path / (name + ".o")
        shutil.copy(os.path.join(B.CSRC, name + ".o"), obj)
        subprocess.run([objdump, "--offloading", str(obj)], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp_path) if f.startswith(name + ".o.") and "gfx950" in f]
        if not dev:                                  # a host-only translation unit (sequence.hip, comm.hip)
            counts[name] = (0, 0)
            continue
        dis = subprocess.run([objdump, "-d", str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
        lines = [ln for ln in dis.splitlines() if pk.search(ln)]
        counts[name] = (len(lines), sum(1 for ln in lines if "op_sel" in ln))
    assert counts["elementwise"] == (0, 0), counts["elementwise"]
    clean = ("conv", "conv_lean", "conv_pair", "conv_rows", "wgrad_lean", "conv_wgrad", "attention_mfma", "pointwise", "pack", "raster", "ownerconv", "boxconv")
    assert all(counts[n][1] == 0 for n in clean), {n: counts[n] for n in clean}
    assert counts["conv_lean"][0] > 0                 # the check does see packed instructions where they are
    assert counts["boxconv"] == (0, 0), counts["boxconv"]       # (built without the instructions altogether, like elementwise)


def test_lds_strides_are_conflict_free_under_the_real_lane_groups():
    """The tile kernels' fragment reads are ds_read_b128, one per lane, lane = (lr = lane & 15: pixel / weight row, lg = lane >> 4:
    8-channel k-group).  MI355X_MICROARCH.md (LDS table): a ds_read_b128 is served in four groups of 16 lanes --
    {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over 64 banks of 4 bytes; a group is conflict-free when its lanes
    hit 16 different 16-byte slots of the 256-byte bank row.  The strides msau_common.h picks (round 4) must be, for every bf16
    configuration the kernels instantiate with >= 16 channels per chunk; the odd-slot padding of rounds 1-3 was not."""
    lib = L.load()
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]

    def worst(stride, lg_step):
        """max lanes of one group on one 16-byte slot, lane address = lr * stride + lg * lg_step"""
        w = 0
        for g in groups:
            slots = [((l & 15) * stride + (l >> 4) * lg_step) // 16 % 16 for l in g]
            w = max(w, max(slots.count(s) for s in set(slots)))
        return w

    for c8 in (2, 4, 8, 16):                                   # 8-channel groups per pixel: 16 ... 128 channels
        ps = lib.msau_lds_pixel_stride(c8 * 16, 2, c8, 1)
        assert ps >= c8 * 16 and ps % 16 == 0 and ps - c8 * 16 <= 48
        assert worst(ps, 16) == 1, (c8, ps)                    # neighbouring lane groups read neighbouring channel groups
        old = c8 * 16 + 16 if c8 % 2 == 0 else c8 * 16
        assert worst(old, 16) == 2                             # what rounds 1-3 used: every read a 2-way conflict
        ps2 = lib.msau_lds_pixel_stride(c8 * 16, 2, c8, 2)     # stride-2 reads (the transposed conv's data gradient): lanes 2 pixels apart
        assert worst(2 * ps2, 16) == 1, (c8, ps2)
    for nks in (2, 3, 5, 9, 18, 36):                           # weight rows: lr = output channel, lg = 8 consecutive k
        ws = lib.msau_lds_wrow_stride(nks, 2)
        assert ws >= nks * 64 and worst(ws, 16) == 1, (nks, ws)
    # not reachable by padding (documented): 8-channel pixels keep the old stride
    assert lib.msau_lds_pixel_stride(16, 2, 1, 1) == 16

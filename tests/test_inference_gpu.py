"""GPU: the forward-only inference path through the C ABI -- device one-hot painter, liveness-reused
activation buffers, softmax + argmax head (fused into the end conv and stand-alone) -- against the reference's NHWC
prediction (tests/golden/kv/kv.npz, made by the reference network on the reference's own painted input) and against
the training-style forward of the same library."""
import json
import os

import numpy as np
import pytest
import torch

from msau_amd import _lib as L
from msau_amd.inference import KVModel
from msau_amd.inference.generic_util import to_categorical
from msau_amd.model import MSAUWrapper
from oracle import msau_oracle as O
from tests.golden_util import GOLDEN, NET_CASES, load_net_case, rel_err

pytestmark = pytest.mark.gpu
KV = os.path.join(GOLDEN, "kv")


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C", [5, 8, 58, 64])
def test_onehot_ids_equals_to_categorical(dtype, C):
    rng = np.random.RandomState(C)
    ids = rng.randint(0, C, size=(2, 19, 23)).astype(np.int32)
    ids[0, 0, 0], ids[1, 3, 4] = -1, C + 3                       # out-of-range ids paint nothing
    Cs = -(-C // 8) * 8
    t = torch.full((2, 19, 23, Cs), 7.0, dtype=torch.float32 if dtype == "fp32" else torch.bfloat16, device="cuda")
    L.call("msau_onehot_ids", _stream(), L.F32 if dtype == "fp32" else L.BF16, torch.from_numpy(ids).cuda().data_ptr(),
           t.data_ptr(), ids.size, C, Cs)
    want = np.zeros((2, 19, 23, Cs), np.float32)
    ok = (ids >= 0) & (ids < C)
    want[..., :C][ok] = to_categorical(ids[ok], C)
    assert np.array_equal(t.float().cpu().numpy(), want)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C", [2, 5, 16, 17, 40])
def test_softmax_argmax_head_kernel(dtype, C):
    torch.manual_seed(C)
    Cs = -(-C // 8) * 8
    td = torch.float32 if dtype == "fp32" else torch.bfloat16
    x = (torch.randn(3001, Cs) * 3).to(td).cuda()
    x[5, :C] = 1.25                                                # a full tie: first index wins
    x[6, 0], x[6, C - 1] = 9.0, 9.0
    probs = torch.empty(3001, C, device="cuda")
    amax = torch.empty(3001, dtype=torch.uint8, device="cuda")
    L.call("msau_softmax_argmax_nhwc", _stream(), L.F32 if dtype == "fp32" else L.BF16, x.data_ptr(), probs.data_ptr(),
           amax.data_ptr(), 3001, C, Cs)
    ref = torch.softmax(x[:, :C].float(), dim=1)
    assert float((probs - ref).abs().max()) < 2e-6
    assert torch.equal(amax.long().cpu(), torch.from_numpy(np.argmax(probs.cpu().numpy(), -1)))   # np.argmax of ITS probs
    assert int(amax[5]) == 0 and int(amax[6]) == 0


def _model(cfg, sd, dtype, **extra):
    kw = dict(scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"], featRoot=cfg["featRoot"],
              filter_size=cfg["filter_size"], pool_size=cfg["pool_size"], final_act="softmax",
              num_blocks=cfg["num_blocks"], dtype=dtype, activation_name=cfg.get("activation", "relu"), **extra)
    m = MSAUWrapper(cfg["channels"], cfg["n_class"], kw)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["net_f8_c13_33x26", "net_f4_c13_b2_64x48", "net_cfg2_336x256x64"])
def test_predict_nhwc_equals_forward_and_reference(name, dtype):
    """forward-only plan (reused buffers, head) == the API forward of the same weights, bit for bit; and both match
    the reference's prediction.  cfg2 is large enough for the lean end conv, i.e. the fused MSAU_CONV_HEAD epilogue."""
    g, cfg, sd, x, _ = load_net_case(name)
    m = _model(cfg, sd, dtype)
    xg = x.cuda()
    with torch.no_grad():
        pred, logits, _ = m(xg)
    probs, amax = m.predict_nhwc(inp=xg)
    plan = m._plan_for(xg, False)
    assert plan.reuse and plan.head_fused == (name == "net_cfg2_336x256x64")
    assert torch.equal(probs.permute(0, 3, 1, 2), pred), "head differs from softmax(forward logits)"
    assert torch.equal(amax.long(), torch.from_numpy(np.argmax(probs.cpu().numpy(), -1)).cuda())
    tol = 2e-4 if dtype == "fp32" else 6e-2
    if "pred" in g.files:
        assert rel_err(probs.permute(0, 3, 1, 2).cpu(), g["pred"]) < tol
    else:
        assert rel_err(probs.permute(0, 3, 1, 2)[:, :, ::7, ::5].cpu(), g["pred_sub"]) < tol
    # the stand-alone head on the stored logits gives the same bits as the fused one
    lg = plan.logits
    p2 = torch.empty_like(probs)
    a2 = torch.empty_like(amax)
    L.call("msau_softmax_argmax_nhwc", _stream(), plan.dtype, lg.data.data_ptr(), p2.data_ptr(), a2.data_ptr(), lg.npix, lg.C, lg.Cs)
    assert torch.equal(p2, probs) and torch.equal(a2, amax)
    # no reuse, same result
    m2 = _model(cfg, sd, dtype, reuse_activations=False)
    q, a = m2.predict_nhwc(inp=xg)
    assert not m2._plan_for(xg, False).reuse
    assert torch.equal(q, probs) and torch.equal(a, amax)
    # repeated calls on the reused buffers are stable
    r, _ = m.predict_nhwc(inp=xg)
    assert torch.equal(r, q)


def test_predict_from_ids_equals_dense_input():
    g, cfg, sd, x, _ = load_net_case("net_f4_c13_b2_64x48")
    m = _model(cfg, sd, "fp32")
    ids = torch.randint(0, cfg["channels"], (2, 64, 48), dtype=torch.int32)
    dense = torch.from_numpy(to_categorical(ids.numpy(), cfg["channels"])).permute(0, 3, 1, 2).float()
    pa, aa = m.predict_nhwc(ids=ids.cuda())
    pa, aa = pa.clone(), aa.clone()
    pb, ab = m.predict_nhwc(inp=dense.cuda())
    assert torch.equal(pa, pb) and torch.equal(aa, ab)
    # HIP-graph replay of the same sweep: same bits, also after the input changes and after a host sync
    pg, ag = m.predict_nhwc(ids=ids.cuda(), graph=True)
    assert torch.equal(pg, pa) and torch.equal(ag, aa)
    ids2 = torch.randint(0, cfg["channels"], (2, 64, 48), dtype=torch.int32).cuda()
    want, _ = m.predict_nhwc(ids=ids2)
    want = want.clone()
    torch.cuda.synchronize()
    got, _ = m.predict_nhwc(ids=ids2, graph=True)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        m.predict_nhwc()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict_nhwc(ids=None, inp=dense)


@pytest.fixture(scope="module")
def kv_gold():
    return np.load(os.path.join(KV, "kv.npz")), json.load(open(os.path.join(KV, "kv.json")))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_kvmodel_network_matches_reference_prediction(kv_gold, dtype, tmp_path):
    """KVModel on the device vs the reference network fed the reference's to_categorical input (17 classes, 60 tokens:
    the stand-alone head and the generic kernels, at the 3-px-per-line scale KVModel really runs at)."""
    g, meta = kv_gold
    cfg, seed = meta["net"]["cfg"], meta["net"]["seed"]
    sd = O.init_params(cfg, seed)
    cs = float(sum(float(v.double().abs().sum()) for v in sd.values()))
    assert abs(cs - meta["net"]["weights_checksum"]) <= 1e-9 * cs
    wpath = str(tmp_path / "kv_weights.pt")
    torch.save(sd, wpath)
    km = KVModel()
    km.load(model_weight=wpath, charset=os.path.join(KV, "charset.txt"), n_class=meta["n_class"], dtype=dtype,
            model_kwargs=dict(featRoot=cfg["featRoot"], scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"],
                              filter_size=cfg["filter_size"], pool_size=cfg["pool_size"], final_act="softmax"))
    assert km.n_token == meta["n_token"] and km.net.channels == meta["n_token"]
    a_pred, a_cls = km._run_net(g["d0.input_mask"])
    want = g["net.pred_nhwc"]
    assert a_pred.shape == want.shape and a_pred.dtype == np.float32
    tol = 2e-4 if dtype == "fp32" else 6e-2
    assert rel_err(a_pred, want) < tol
    assert np.array_equal(a_cls, np.argmax(a_pred, -1))
    if dtype == "fp32":
        # class decisions agree with the reference's wherever its top-2 margin exceeds the tolerance
        top2 = np.sort(want, -1)[..., -2:]
        sure = (top2[..., 1] - top2[..., 0]) > 1e-3
        assert sure.mean() > 0.5 and np.array_equal(a_cls[sure], np.argmax(want, -1)[sure])
    # end to end: JSON in, fields out; identical to post-processing the same prediction by hand
    kv, img = km.predict((os.path.join(KV, "layout0.json"), None))
    assert img is None
    import copy
    values, _ = KVModel._extract_value(g["d0.line_mask"], g["d0.char_mask"], copy.deepcopy(meta["d0"]["lines"]), a_pred, meta["n_class"])
    from msau_amd.inference import post_process_kv
    assert kv == post_process_kv(values)
    # evaluation bookkeeping runs (ground truth = the layout itself)
    res = km.run_test([os.path.join(KV, "layout0.json")], str(tmp_path), label_dir=None)
    assert res == [kv]

#!/usr/bin/env python3
"""Golden-vector generator  --  TEST INFRASTRUCTURE, build-container only.

Imports the reference's own ``model.model`` / ``model.layers`` (read-only mount at
/root/reference, CPU) and writes small ``.npz`` fixtures under ``tests/golden/``.
The reference never travels: only the numeric inputs/outputs land in the repo.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_goldens.py

Weights for the net-level goldens are drawn by ``oracle.msau_oracle.init_params(cfg, seed)``
(seeded torch.Generator, reproducible) and loaded into the reference module with
``load_state_dict`` -- fixtures then only need the seeds, a checksum of what the seeds
produced, and the reference's outputs.  Op-level goldens store full tensors.
"""
import io
import os
import sys
import contextlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MSAU_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

from oracle import msau_oracle as O

with contextlib.redirect_stdout(io.StringIO()):
    from model.model import MSAUWrapper, MultiConvResidualBlock            # reference
    from model.layers import layers as RL                                  # reference
    from model.layers import attention as RA                               # reference

OUT = os.environ.get("MSAU_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def summarize(t: torch.Tensor, n: int = 64):
    """norm, sum, strided sample (up to n elements) of a tensor."""
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // n)
    return np.array([float(f.norm()), float(f.sum())]), f[::stride][:n].float().numpy()


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def build_ref(cfg):
    kw = dict(scale_space_num=cfg["scale_space_num"], res_depth=cfg["res_depth"],
              featRoot=cfg["featRoot"], filter_size=cfg["filter_size"],
              pool_size=cfg["pool_size"], final_act="softmax", activation_name=cfg.get("activation", "relu"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = MSAUWrapper(cfg["channels"], cfg["n_class"], kw)
    net.msau_net.num_blocks = cfg["num_blocks"]          # blocks are independent modules
    return net


def net_golden(name, cfg, B, H, W, seed, dense=False, with_step=True):
    sd = O.init_params(dict(cfg, num_blocks=3), seed)    # the reference always owns 3 blocks
    net = build_ref(cfg)
    net.load_state_dict(sd)
    x, label = O.synthetic_batch(B, cfg["channels"], H, W, cfg["n_class"], seed + 1, dense=dense)
    pred, logits, aux = net(x)
    out = dict(cfg=np.array(repr(cfg)), B=B, H=H, W=W, seed=seed, dense=dense,
               weights_checksum=checksum(sd), input_checksum=float(x.double().abs().sum()),
               label_checksum=float(label.sum()))
    big = H * W > 20000                      # keep fixtures small: strided sample + norms
    def put(nm, t):
        if big:
            out[nm + "_sub"] = t.detach()[:, :, ::7, ::5].numpy()
            out[nm + "_summary"] = summarize(t)[0]
        else:
            out[nm] = t.detach().numpy()
    put("pred", pred); put("logits", logits)
    if aux is not None:
        put("aux", aux)
    if with_step:
        # reference loss is batch-1 only (model.py:452-453): evaluate per sample, average
        losses = []
        for i in range(B):
            l_i = net.loss(logits[i:i + 1], aux[i:i + 1], label[i:i + 1])
            losses.append(l_i)
        loss = sum(losses) / B
        net.zero_grad()
        loss.backward()
        out["loss"] = float(loss)
        names, gsum, gsamp, dead = [], [], [], []
        for k, p in net.named_parameters():
            if int(k.split(".")[2]) >= cfg["num_blocks"]:
                continue
            names.append(k)
            if p.grad is None:
                dead.append(k)
                gsum.append(np.zeros(2)); gsamp.append(np.zeros(1, np.float32))
            else:
                s, smp = summarize(p.grad)
                gsum.append(s); gsamp.append(smp)
        out["param_names"] = np.array(names)
        out["dead_params"] = np.array(dead)
        out["grad_summary"] = np.stack(gsum)
        out["grad_samples"] = np.array(gsamp, dtype=object)
        # one optimiser step exactly as train_chargrid_funsd_msau.py:24-26,57-59
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, net.parameters()), lr=1e-4)
        before = {k: p.detach().clone() for k, p in net.named_parameters()}
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        out["grad_norm"] = float(gn)
        dsum, dsamp = [], []
        for k, p in net.named_parameters():
            if k in names:
                s, smp = summarize(p.detach() - before[k])
                dsum.append(s); dsamp.append(smp)
        out["delta_summary"] = np.stack(dsum)
        out["delta_samples"] = np.array(dsamp, dtype=object)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "loss" in out and out["loss"])


def op_goldens():
    torch.manual_seed(1234)
    out = {}

    def fwd_bwd(tag, module, x, **kw):
        x = x.clone().requires_grad_(True)
        y = module(x, **kw)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[f"{tag}.x"] = x.detach().numpy(); out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.gy"] = gy.numpy(); out[f"{tag}.gx"] = x.grad.numpy()
        for k, p in module.named_parameters():
            out[f"{tag}.p.{k}"] = p.detach().numpy()
            out[f"{tag}.g.{k}"] = p.grad.numpy()

    # A2: 3x3 / 1x1 / 4x4 SAME convs, odd sizes, with and without ReLU
    fwd_bwd("conv3", RL.Conv2dBnLrnDrop([3, 3, 8, 16], activation=torch.nn.ReLU), torch.randn(2, 8, 11, 9))
    fwd_bwd("conv3lin", RL.Conv2dBnLrnDrop([3, 3, 16, 8], activation=None), torch.randn(1, 16, 7, 13))
    fwd_bwd("conv1", RL.Conv2dBnLrnDrop([1, 1, 32, 16], activation=torch.nn.ReLU), torch.randn(2, 32, 5, 6))
    fwd_bwd("conv4", RL.Conv2dBnLrnDrop([4, 4, 8, 5], activation=None), torch.randn(2, 8, 9, 10))
    fwd_bwd("conv3c13", RL.Conv2dBnLrnDrop([3, 3, 13, 8], activation=None), torch.randn(1, 13, 6, 7))
    # A3: dilated conv + LRN, all dilations used by S=4
    for d, (ci, co) in zip((1, 2, 4, 8), ((5, 8), (8, 16), (16, 32), (32, 64))):
        fwd_bwd(f"dil{d}", RL.DilConv2dBnLrnDrop([3, 3, ci, co], rate=d, activation=None),
                3.0 * torch.randn(1, ci, 19, 17))
    # LRN alone with large activations so the normaliser matters
    for C in (8, 16, 64):
        lrn_mod = torch.nn.LocalResponseNorm(C)
        fwd_bwd(f"lrn{C}", lrn_mod, 20.0 * torch.randn(2, C, 5, 7))
    # A4: transposed conv, output_size odd and even
    fwd_bwd("deconv_even", RL.Deconv2DBnLrnDrop([3, 3, 8, 16], activation=None), torch.randn(2, 16, 5, 4),
            output_size=[10, 8])
    fwd_bwd("deconv_odd", RL.Deconv2DBnLrnDrop([3, 3, 8, 16], activation=None), torch.randn(1, 16, 5, 4),
            output_size=[9, 7])
    # A5: residual block
    fwd_bwd("res", MultiConvResidualBlock(2, 3, 8, False, torch.nn.ReLU), torch.randn(2, 8, 9, 7))
    # A6: self attention (C=64 -> d=8) and (C=32 -> d=4)
    fwd_bwd("attn64", RA.SAWrapperBlock(64), torch.randn(2, 64, 5, 6))
    fwd_bwd("attn32", RA.SAWrapperBlock(32), 2.0 * torch.randn(1, 32, 7, 3))
    # A7: zero SAME pad + 2x2 max pool on odd sizes (post-ReLU input, with ties at 0)
    x = torch.relu(torch.randn(2, 8, 7, 9)).requires_grad_(True)
    y = torch.nn.functional.max_pool2d(RL.pad_2d(x, 'SAME', 'pool2d', 2, 2, 2, 2), 2, 2)
    gy = torch.randn_like(y); y.backward(gy)
    out["pool.x"], out["pool.y"], out["pool.gy"], out["pool.gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    # A11: masked CE loss, batch 1
    with contextlib.redirect_stdout(io.StringIO()):
        w = MSAUWrapper(4, 5, dict(scale_space_num=2, res_depth=1, featRoot=4, final_act="softmax"))
    lg = torch.randn(1, 5, 6, 7, requires_grad=True); ax = torch.randn(1, 5, 6, 7, requires_grad=True)
    lab = torch.randint(0, 5, (1, 6, 7))
    loss = w.loss(lg, ax, lab); loss.backward()
    out["ce.logits"], out["ce.aux"], out["ce.label"] = lg.detach().numpy(), ax.detach().numpy(), lab.numpy()
    out["ce.loss"], out["ce.glogits"], out["ce.gaux"] = float(loss), lg.grad.numpy(), ax.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("wrote ops", len(out), "arrays")


if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_NET", "1") == "1":
    op_goldens()
if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_NET", "1") == "1" and os.environ.get("MSAU_GOLDEN_NETS", "1") == "1":
    base = dict(n_class=5, scale_space_num=4, res_depth=2, featRoot=8, filter_size=3, pool_size=2, num_blocks=3)
    # G2: the hyper-parameters the reference instantiates (train_chargrid_funsd_msau.py:211-214), odd size
    net_golden("net_f8_c13_33x26", dict(base, channels=13), 1, 33, 26, seed=11)
    # batch 2, even size, small width (exercises 8-channel padding: featRoot 4, d = 32//8 = 4)
    net_golden("net_f4_c13_b2_64x48", dict(base, channels=13, featRoot=4), 2, 64, 48, seed=12)
    # cfg 1 miniature: 1 stage (aux is None -> no loss golden)
    net_golden("net_1stage_c32_b2_40x40", dict(base, channels=32, num_blocks=1), 2, 40, 40, seed=13, with_step=False)
    # cfg 4 miniature: 2 stages, dense (BERT-like) input
    net_golden("net_2stage_c24_dense_24x40", dict(base, channels=24, num_blocks=2), 1, 24, 40, seed=14, dense=True)
    # res_depth 3 / scale_space_num 3 (wrapper defaults differ from the train script)
    net_golden("net_r3_s3_c8_21x35", dict(base, channels=8, res_depth=3, scale_space_num=3), 1, 21, 35, seed=15)
    # the reference's CONSTRUCTOR DEFAULTS (model/model.py:406-408): 6 scales -> 256 channels, dilation up to 32, res_depth 3
    net_golden("net_defaults_s6_r3_c8_70x96", dict(base, channels=8, scale_space_num=6, res_depth=3), 1, 70, 96, seed=17)
    # activation_name="elu" (model/model.py:412-416): the only other activation the wrapper knows; odd size, all levels
    net_golden("net_elu_f8_c13_33x26", dict(base, channels=13, activation="elu"), 1, 33, 26, seed=18)
    # cfg 2 geometry checksum: 336x256x64, 3 stages, forward + loss + grads summaries
    net_golden("net_cfg2_336x256x64", dict(base, channels=64), 1, 336, 256, seed=16)


# ---------------------------------------------------------------------------------------------
# G3: chargrid pipeline goldens.  Synthetic FUNSD-format documents -> the reference's own
# preprocessing + loader (modules it imports but never uses on this path are stubbed).
# ---------------------------------------------------------------------------------------------
def funsd_goldens():
    import json, pickle, random, tempfile, types, shutil
    for name in ("cv2", "skimage", "skimage.morphology", "tensorboardX", "sentence_transformers"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "skimage.morphology":
                m.skeletonize = lambda *a, **k: None
            if name == "sentence_transformers":
                m.SentenceTransformer = lambda *a, **k: None
            sys.modules[name] = m
    with contextlib.redirect_stdout(io.StringIO()):
        import funsd_preprocessing_word_level as RP          # reference
        import data_generator_funsd_bert as RD               # reference
    rng = random.Random(2024)
    words_pool = ["Invoice", "No.", "12345", "Date:", "2019-03-07", "TOTAL", "$1,204.50", "Name", "J.", "Doe", "",
                  "Qty", "7", "Address:", "42", "Elm", "St.", "fax", "(555)", "010-9999"]
    labels_pool = ["question", "answer", "header", "other"]
    out_dir = os.path.join(OUT, "funsd")
    os.makedirs(os.path.join(out_dir, "train"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "test"), exist_ok=True)

    def make_doc(n_lines, seed):
        r = random.Random(seed)
        form, y = [], 30
        for lid in range(n_lines):
            x = r.randint(20, 200)
            words, x0 = [], x
            h = r.randint(14, 26)
            for _ in range(r.randint(1, 4)):
                t = r.choice(words_pool)
                w = max(6, 9 * max(len(t), 1) + r.randint(-3, 6))
                words.append({"box": [x0, y, x0 + w, y + h], "text": t})
                x0 += w + r.randint(5, 14)
            form.append({"box": [x, y, x0 - 5, y + h], "text": " ".join(w["text"] for w in words),
                         "label": labels_pool[lid % 4] if lid < 4 else r.choice(labels_pool),
                         "words": words, "linking": [], "id": lid})
            y += h + r.randint(6, 30)
        return {"form": form}

    for split, seeds in (("train", (1, 2)), ("test", (3,))):
        for s in seeds:
            with open(os.path.join(out_dir, split, f"doc{s}.json"), "w") as fh:
                json.dump(make_doc(6 + s, 100 + s), fh)
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            train, inv = RP.get_preprocessed_list_word_msau(os.path.join(out_dir, "train"))
            test, _ = RP.get_preprocessed_list_word_msau(os.path.join(out_dir, "test"), inv_dict_charset=inv)
        res = {"charset": np.array(sorted(inv.keys()))}
        for split, docs in (("train", train), ("test", test)):
            docs.sort(key=lambda d: d["file_path"])
            pickle.dump(docs, open(f"{split}.pkl", "wb"))
        tr = RD.FUNSDCharGridDataLoaderBoxMaskBoxLabel("train.pkl")
        te = RD.FUNSDCharGridDataLoaderBoxMaskBoxLabel("test.pkl", tr.labels)
        res["labels_json"] = np.array(json.dumps(tr.labels))
        for split, ds in (("train", tr), ("test", te)):
            for i in range(len(ds)):
                it = ds[i]
                res[f"{split}{i}.mask"] = it["mask"].numpy().astype(np.uint8)
                res[f"{split}{i}.label"] = it["label"].numpy().astype(np.uint8)
                res[f"{split}{i}.nwords"] = len(it["ocr_values"])
                res[f"{split}{i}.file"] = np.array(os.path.basename(ds.inp_list[i]["file_path"]))
                res[f"{split}{i}.word_to_textline"] = np.array(ds.inp_list[i]["word_to_textline"])
                res[f"{split}{i}.feat_sums"] = np.array([f.sum() for f in ds.inp_list[i]["charset_feature"]])
        np.savez_compressed(os.path.join(out_dir, "chargrid.npz"), **res)
        print("wrote funsd goldens", {k: v.shape for k, v in res.items() if k.endswith("mask")})
        # BERT-embedding variant (BASELINE configs[3]): get_box_mask_box_label paints one `transformer_feature` vector per
        # text LINE box (data_generator_funsd_bert.py:64-93,240).  No script in the tree produces that pickle field
        # (SURVEY appendix B), so seeded N(0,1) vectors stand in for the sentence embeddings (24-d keeps the fixture small;
        # the painter is width-agnostic).
        bres = {}
        for split in ("train", "test"):
            docs = pickle.load(open(f"{split}.pkl", "rb"))
            for di, d in enumerate(docs):
                r = np.random.RandomState(500 + di + (0 if split == "train" else 50))
                d["transformer_feature"] = r.randn(len(d["cells"]), 24).astype(np.float32)
                bres[f"{split}{di}.feats"] = d["transformer_feature"]
                bres[f"{split}{di}.cells"] = np.array([[c.x, c.y, c.w, c.h] for c in d["cells"]], np.int64)
                bres[f"{split}{di}.labels"] = np.array(d["labels"])
            pickle.dump(docs, open(f"{split}_bert.pkl", "wb"))
        btr = RD.FUNSDBertDataLoaderBoxMaskBoxLabel("train_bert.pkl")
        bte = RD.FUNSDBertDataLoaderBoxMaskBoxLabel("test_bert.pkl", btr.labels)
        bres["labels_json"] = np.array(json.dumps(btr.labels))
        for split, ds in (("train", btr), ("test", bte)):
            for i in range(len(ds)):
                it = ds[i]
                bres[f"{split}{i}.mask"] = it["mask"].numpy().astype(np.float32)
                bres[f"{split}{i}.label"] = it["label"].numpy().astype(np.uint8)
        np.savez_compressed(os.path.join(out_dir, "bertgrid.npz"), **bres)
        print("wrote bert goldens", {k: v.shape for k, v in bres.items() if k.endswith("mask")})
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_FUNSD", "1") == "1":
    # label ids follow Python set() order (data_generator_funsd_bert.py:196-201): pin the hash seed by running
    # this part in a child interpreter started with PYTHONHASHSEED=0 (CPU-only script, build container only)
    if os.environ.get("PYTHONHASHSEED") == "0":
        funsd_goldens()
    else:
        import subprocess
        env = dict(os.environ, PYTHONHASHSEED="0", MSAU_GOLDEN_NET="0", MSAU_GOLDEN_KV="0", MSAU_GOLDEN_FUNSD="1",
                   MSAU_GOLDEN_TRAIN="0", PYTHONDONTWRITEBYTECODE="1")
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)


# ---------------------------------------------------------------------------------------------
# G5: inference-side goldens.  Synthetic layout+OCR JSON -> the reference's own
# KVModel._generate_masks_from_label / _extract_value / post_process_kv / read_json_gt /
# sort_box_reading_order / morph helpers, plus the reference network's NHWC prediction for the
# painted one-hot input (weights from seeds, as in the net goldens).  cv2 / skimage are only
# used by the reference's drawing code and are stubbed.
# ---------------------------------------------------------------------------------------------
def kv_goldens():
    import copy, json, random, types, warnings
    warnings.simplefilter("ignore")
    for name in ("cv2", "skimage", "skimage.morphology"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "skimage.morphology":
                m.skeletonize = lambda *a, **k: None
            sys.modules[name] = m
    with contextlib.redirect_stdout(io.StringIO()):
        from inference import kv_model as RK                 # reference
        from inference import generic_util as RG             # reference
        from inference import morph_util as RM               # reference
    out_dir = os.path.join(OUT, "kv")
    os.makedirs(out_dir, exist_ok=True)
    charset = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0:-./#"
    with open(os.path.join(out_dir, "charset.txt"), "w") as fh:
        fh.write(charset)
    n_class = 17
    words = ["Bank", "of", "Elm", "Branch:", "North-7", "Acct", "No.", "00123-4567", "Type", "Savings", "Name", "J. Doe",
             "KANA", "ジョン", "Inst.", "First Union", "", "Ref#", "2019/03/07", "x"]

    def make_doc(seed, n_lines):
        r = random.Random(seed)
        lines, y = [], r.randint(40, 120)
        for i in range(n_lines):
            h = r.randint(18, 34)
            x = r.randint(30, 400)
            for _ in range(r.randint(1, 3)):                 # several cells on the same row
                text = " ".join(r.choice(words) for _ in range(r.randint(1, 3))).strip()
                w = max(10, int(h * 0.55 * max(len(text), 1)) + r.randint(-4, 8))
                value = r.choice([0, 0, 1, 2, 3, 4, 4, 5, 6, 9, 10, 10, 11, 12])
                lines.append({"box": [x, y, x + w, y + h], "text": text, "type": r.choice([0, 1, 2]), "value": value})
                x += w + r.randint(12, 60)
            y += h + r.randint(4, 26)
        return {"img_shape": [y + 60, 1200], "lines": lines}

    res, meta = {}, {}
    rk = RK.KVModel()
    rk.charset = " " + "$" + charset
    rk.tok_to_id = {t: i for i, t in enumerate(rk.charset)}
    rk.id_to_tok = {i: t for t, i in rk.tok_to_id.items()}
    rk.n_token = len(rk.tok_to_id)
    rk.n_class = n_class
    meta["n_token"], meta["n_class"] = rk.n_token, n_class

    def jsonable(o):
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (list, tuple)):
            return [jsonable(v) for v in o]
        if isinstance(o, dict):
            return {str(k): jsonable(v) for k, v in o.items()}
        return o

    for di, (seed, n_lines) in enumerate(((11, 7), (12, 12), (13, 4))):
        doc = make_doc(seed, n_lines)
        path = os.path.join(out_dir, f"layout{di}.json")
        with open(path, "w") as fh:
            json.dump(doc, fh, ensure_ascii=False)
        inp, line_mask, char_mask, lines, scale, bg_pad, bbox = rk._generate_masks_from_label(path)
        res[f"d{di}.input_mask"], res[f"d{di}.line_mask"], res[f"d{di}.char_mask"] = inp, line_mask, char_mask
        m = {"scale": float(scale), "bg_pad": int(bg_pad), "bbox": [int(v) for v in bbox], "lines": jsonable(lines)}
        # a plausible class map: every line votes for class value+1 over (part of) its box; some lines are split
        # between two classes, some classes get several blobs, everything else is background; then noise
        r = np.random.RandomState(seed)
        H, W = inp.shape
        score = r.rand(H, W, n_class).astype(np.float32) * 0.6
        score[:, :, 0] += 1.0
        for li, l in enumerate(lines):
            x1, y1, x2, y2 = l["box"]
            c = doc["lines"][li]["value"] + 1 if doc["lines"][li]["value"] > 0 else 0
            if c < 2:
                continue
            if r.rand() < 0.35 and x2 - x1 > 8:              # shared line: second half goes to the next class
                xm = (x1 + x2) // 2
                score[y1:y2, x1:xm, c] += 2.0
                score[y1:y2, xm:x2, min(c + 1, n_class - 1)] += 2.0
            else:
                score[y1:y2, x1:x2, c] += 2.0
            if r.rand() < 0.3:                                 # a stray blob of the same class elsewhere
                yy, xx = r.randint(0, max(1, H - 3)), r.randint(0, max(1, W - 6))
                score[yy:yy + 3, xx:xx + 6, c] += 2.5
        e = np.exp(score - score.max(-1, keepdims=True))
        pred = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        res[f"d{di}.pred"] = pred.astype(np.float16)          # the test feeds exactly these (fp16-rounded) values
        pred_in = res[f"d{di}.pred"].astype(np.float32)
        values, new_mask = rk._extract_value(line_mask, char_mask, copy.deepcopy(lines), pred_in, n_class)
        m["values"] = jsonable(values)
        res[f"d{di}.kept"] = new_mask[:, :, 1:].astype(np.uint8)
        res[f"d{di}.kept0"] = new_mask[:, :, 0].astype(np.float32)
        m["kv"] = jsonable(RK.post_process_kv(values))
        gt = RG.read_json_gt(path, scale=scale, offset=(bbox[0] - bg_pad, bbox[1] - bg_pad))
        m["gt"] = jsonable(gt)
        meta[f"d{di}"] = m

    # reading order / morphology on random inputs
    r = random.Random(5)
    ro = []
    for _ in range(6):
        cells = []
        for i in range(r.randint(1, 9)):
            x, y, w, h = r.randint(0, 300), r.randint(0, 200), r.randint(5, 120), r.randint(5, 40)
            cells.append({"box": [x, y, x + w, y + h], "tag": i})
        order = [c["tag"] for c in RG.sort_box_reading_order(copy.deepcopy(cells))]
        ro.append({"cells": cells, "order": order})
    meta["reading_order"] = ro
    rs = np.random.RandomState(9)
    for i, (shape, p) in enumerate((((23, 31), 0.35), ((8, 50), 0.6), ((40, 7), 0.2))):
        mk = rs.rand(*shape) < p
        res[f"morph{i}.in"] = mk
        res[f"morph{i}.closing13"] = RM.r_closing(mk, (1, 3))
        res[f"morph{i}.opening22"] = RM.r_opening(mk, (2, 2))
        lab, objs = RM.connected_components(mk)
        res[f"morph{i}.labels"] = lab.astype(np.int32)
        res[f"morph{i}.objects"] = np.array([[o[0].start, o[0].stop, o[1].start, o[1].stop] for o in objs], np.int32)
    boxes = [[rs.randint(0, 50), rs.randint(0, 50)] for _ in range(12)]
    boxes = [[x, y, x + rs.randint(1, 40), y + rs.randint(1, 40)] for x, y in boxes]
    meta["boxes"] = boxes
    meta["filter_overlap"] = jsonable(RM.filter_overlap_boxes(copy.deepcopy(boxes), return_indices=True))
    meta["filter_overlap_bigger"] = jsonable(RM.filter_overlap_boxes_bigger(copy.deepcopy(boxes), intersect_thres=0.5, return_indices=True))
    meta["iou"] = [[float(RM.IoU(a, b)) for b in boxes[:4]] for a in boxes[:4]]
    meta["intersect_area"] = [[float(RM.intersect_area(a, b)) for b in boxes[:4]] for a in boxes[:4]]

    # the reference network on the painted input of doc 0 (kv_model.py:274-279,305-309), weights from seeds
    cfg = dict(channels=rk.n_token, n_class=n_class, featRoot=8, scale_space_num=4, res_depth=2, filter_size=3,
               pool_size=2, num_blocks=3)
    sd = O.init_params(cfg, 77)
    net = build_ref(cfg)
    net.load_state_dict(sd)
    net.eval()
    inp = res["d0.input_mask"]
    batch_x = torch.from_numpy(np.expand_dims(RG.to_categorical(inp, rk.n_token), 0)).transpose(1, -1).transpose(2, 3).float()
    with torch.set_grad_enabled(False):
        a_pred, _, _ = net(batch_x)
        a_pred = torch.transpose(a_pred, 1, -1).transpose(1, 2).numpy()
    res["net.pred_nhwc"] = a_pred[0]
    meta["net"] = {"cfg": cfg, "seed": 77, "weights_checksum": checksum(sd)}
    np.savez_compressed(os.path.join(out_dir, "kv.npz"), **res)
    with open(os.path.join(out_dir, "kv.json"), "w") as fh:
        json.dump(meta, fh, ensure_ascii=False, indent=1)
    print("wrote kv goldens", {k: v.shape for k, v in res.items() if k.endswith("input_mask")}, a_pred.shape)


if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_KV", "1") == "1":
    kv_goldens()


# ---------------------------------------------------------------------------------------------
# G6: training-side goldens.  The reference's own `UNetLoss` (model/training/cost.py:35-65) on
# random logits / one-hot targets, and a dict checkpoint written by the reference's
# `utils.io_utils.save_checkpoint` (io_utils.py:83-105) for a miniature net after one Adam step.
# ---------------------------------------------------------------------------------------------
def train_goldens():
    import types, tempfile, shutil
    for name in ("tensorboardX", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "tensorboardX":
                m.SummaryWriter = lambda *a, **k: None
            sys.modules[name] = m
    with contextlib.redirect_stdout(io.StringIO()):
        from model.training.cost import UNetLoss                    # reference
        from utils import io_utils as RIO                           # reference
    out_dir = os.path.join(OUT, "train")
    os.makedirs(out_dir, exist_ok=True)
    res = {}
    g = torch.Generator().manual_seed(4242)
    crit = UNetLoss({})
    for tag, (B, C, H, W, with_aux) in (("a", (2, 5, 9, 7, True)), ("b", (1, 17, 6, 11, True)), ("c", (3, 4, 5, 5, False))):
        logits = (2.0 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
        aux = (2.0 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True) if with_aux else None
        lab = torch.randint(0, C, (B, H, W), generator=g)
        tgt = torch.nn.functional.one_hot(lab, C).permute(0, 3, 1, 2).float()
        kw = {"aux_logits": aux, "aux_tgt": tgt} if with_aux else {}
        acc, loss, final = crit(logits, tgt, kw)
        loss.backward()
        res[f"{tag}.logits"], res[f"{tag}.label"] = logits.detach().numpy(), lab.numpy()
        res[f"{tag}.acc"], res[f"{tag}.loss"] = float(acc), float(loss)
        res[f"{tag}.final"] = float(final) if final is not None else np.nan
        res[f"{tag}.glogits"] = logits.grad.numpy()
        if with_aux:
            res[f"{tag}.aux"], res[f"{tag}.gaux"] = aux.detach().numpy(), aux.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "unet_loss.npz"), **res)
    # reference-written checkpoint: miniature net (featRoot 4, 2 scales), one clip + Adam step, then save_checkpoint
    cfg = dict(channels=6, n_class=3, scale_space_num=2, res_depth=1, featRoot=4, filter_size=3, pool_size=2, num_blocks=3)
    sd = O.init_params(cfg, 91)
    net = build_ref(cfg)
    net.load_state_dict(sd)
    x, label = O.synthetic_batch(1, cfg["channels"], 12, 10, cfg["n_class"], 92)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, net.parameters()), lr=1e-4)
    pred, logits, aux = net(x)
    loss = net.loss(logits, aux, label)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    args = types.SimpleNamespace(ckptdir=tempfile.mkdtemp(), bmname=None, dataset="funsd", method="msau", hidden_dim=20, output_dim=20)
    try:
        RIO.save_checkpoint(net, opt, args, num_epochs=7)
        src = RIO.create_filename(args.ckptdir, args, False, num_epochs=7)
        shutil.copy(src, os.path.join(out_dir, "ref_checkpoint.pth.tar"))
        rel = os.path.relpath(src, args.ckptdir)
    finally:
        shutil.rmtree(args.ckptdir)
    with torch.no_grad():
        pred2, logits2, aux2 = net(x)
    np.savez_compressed(os.path.join(out_dir, "ref_checkpoint_meta.npz"), cfg=np.array(repr(cfg)), x=x.numpy(),
                        logits_after=logits2.numpy(), aux_after=aux2.numpy(), rel_path=np.array(rel),
                        state_checksum=checksum(net.state_dict()), loss=float(loss))
    print("wrote train goldens", sorted(res)[:4], rel)


if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_TRAIN", "1") == "1":
    train_goldens()


# ---------------------------------------------------------------------------------------------
# G7: `UNetLoss(class_weights=...)` -- the reference's weighted cross entropy (model/training/cost.py:24-31:
# torch.nn.CrossEntropyLoss(weight)) on random logits / one-hot targets, with and without auxiliary logits; a 17-class case
# (the key-value head's class count) and a class whose weight is zero.  Own fixture, own switch: the other training fixtures
# (a pickled checkpoint among them) are not rewritten.
# ---------------------------------------------------------------------------------------------
def weighted_loss_goldens():
    import types
    for name in ("tensorboardX", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    with contextlib.redirect_stdout(io.StringIO()):
        from model.training.cost import UNetLoss                    # reference
    out_dir = os.path.join(OUT, "train")
    os.makedirs(out_dir, exist_ok=True)
    res = {}
    g = torch.Generator().manual_seed(777)
    for tag, (B, C, H, W, with_aux) in (("a", (2, 5, 9, 7, True)), ("b", (1, 17, 6, 11, True)), ("c", (3, 4, 5, 5, False))):
        cw = (0.25 + 2.0 * torch.rand(C, generator=g)).numpy().astype(np.float32)
        if tag == "c":
            cw[2] = 0.0
        crit = UNetLoss({"class_weights": [float(v) for v in cw]})
        logits = (2.0 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
        aux = (2.0 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True) if with_aux else None
        lab = torch.randint(0, C, (B, H, W), generator=g)
        tgt = torch.nn.functional.one_hot(lab, C).permute(0, 3, 1, 2).float()
        acc, loss, final = crit(logits, tgt, {"aux_logits": aux, "aux_tgt": tgt} if with_aux else {})
        loss.backward()
        res[f"{tag}.class_weights"] = cw
        res[f"{tag}.logits"], res[f"{tag}.label"] = logits.detach().numpy(), lab.numpy()
        res[f"{tag}.acc"], res[f"{tag}.loss"] = float(acc), float(loss)
        res[f"{tag}.final"] = float(final) if final is not None else np.nan
        res[f"{tag}.glogits"] = logits.grad.numpy()
        if with_aux:
            res[f"{tag}.aux"], res[f"{tag}.gaux"] = aux.detach().numpy(), aux.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "unet_loss_weighted.npz"), **res)
    print("wrote weighted-loss goldens", {t: res[f"{t}.loss"] for t in "abc"})


if __name__ == "__main__" and os.environ.get("MSAU_GOLDEN_WLOSS", "1") == "1":
    weighted_loss_goldens()

"""CPU: the chargrid input pipeline counterpart (msau_amd/data/funsd.py) against golden arrays the
reference's own funsd_preprocessing_word_level.py + data_generator_funsd_bert.py produced for the
committed synthetic FUNSD-format documents (tests/golden/funsd; generator: oracle/gen_goldens.py)."""
import json
import os
import pickle

import numpy as np
import torch

from msau_amd.data import funsd as F

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "funsd")


def _build(tmp_path):
    g = np.load(os.path.join(G, "chargrid.npz"), allow_pickle=True)
    train, inv = F.get_preprocessed_list_word_msau(os.path.join(G, "train"))
    test, inv2 = F.get_preprocessed_list_word_msau(os.path.join(G, "test"), inv_dict_charset=inv)
    for name, docs in (("train", train), ("test", test)):
        docs.sort(key=lambda d: d["file_path"])
        with open(tmp_path / f"{name}.pkl", "wb") as fh:
            pickle.dump(docs, fh)
    labels = json.loads(str(g["labels_json"]))
    tr = F.FUNSDCharGridDataLoaderBoxMaskBoxLabel(str(tmp_path / "train.pkl"), labels)
    te = F.FUNSDCharGridDataLoaderBoxMaskBoxLabel(str(tmp_path / "test.pkl"), labels)
    return g, inv, tr, te


def test_preprocess_and_chargrid_match_reference(tmp_path):
    g, inv, tr, te = _build(tmp_path)
    assert sorted(inv.keys()) == [str(c) for c in g["charset"]]
    for split, ds in (("train", tr), ("test", te)):
        assert len(ds) == sum(1 for k in g.files if k.startswith(split) and k.endswith(".mask"))
        for i in range(len(ds)):
            it = ds[i]
            doc = ds.inp_list[i]
            assert os.path.basename(doc["file_path"]) == str(g[f"{split}{i}.file"])
            assert set(doc.keys()) == {"file_path", "word_to_textline", "cells_word", "cells", "labels", "ids",
                                       "link", "charset_feature"}
            assert list(doc["word_to_textline"]) == list(g[f"{split}{i}.word_to_textline"])
            assert np.array_equal(np.array([f.sum() for f in doc["charset_feature"]]), g[f"{split}{i}.feat_sums"])
            assert it["mask"].dtype == torch.float32 and it["mask"].dim() == 4 and it["label"].dim() == 3
            assert np.array_equal(it["mask"].numpy().astype(np.uint8), g[f"{split}{i}.mask"]), (split, i)   # bit exact
            assert np.array_equal(it["label"].numpy().astype(np.uint8), g[f"{split}{i}.label"]), (split, i)
            assert len(it["ocr_values"]) == int(g[f"{split}{i}.nwords"])


def test_default_label_map_and_labels_file(tmp_path, monkeypatch):
    g, inv, tr, te = _build(tmp_path)
    monkeypatch.chdir(tmp_path)
    ds = F.FUNSDCharGridDataLoaderBoxMaskBoxLabel(str(tmp_path / "train.pkl"))
    assert sorted(ds.labels.values()) == list(range(len(ds.labels)))          # a bijection onto 0..n-1
    assert json.load(open(tmp_path / "labels")) == ds.labels                  # the reference also writes ./labels
    many = ds[[0, 1]]
    assert isinstance(many, list) and len(many) == 2


def test_charset_helpers():
    cs, inv = F.get_charset("ab  c\nA b")
    assert cs == ["A", "a", "b", "c"]
    m = F.transform_from_charset("a?c", inv)
    assert m.shape == (3, 4) and m[1].sum() == 0 and m[0, 1] == 1 and m[2, 3] == 1


def test_bert_dense_painter_matches_reference(tmp_path):
    """`get_box_mask_box_label` / `FUNSDBertDataLoaderBoxMaskBoxLabel` (data_generator_funsd_bert.py:64-93,240: the loader
    BASELINE.json configs[3] names) against arrays the reference's own loader painted from the same documents and the
    same per-line feature vectors (tests/golden/funsd/bertgrid.npz)."""
    g = np.load(os.path.join(G, "bertgrid.npz"), allow_pickle=True)
    train, inv = F.get_preprocessed_list_word_msau(os.path.join(G, "train"))
    test, _ = F.get_preprocessed_list_word_msau(os.path.join(G, "test"), inv_dict_charset=inv)
    labels = json.loads(str(g["labels_json"]))
    for name, docs in (("train", train), ("test", test)):
        docs.sort(key=lambda d: d["file_path"])
        for di, d in enumerate(docs):
            d["transformer_feature"] = g[f"{name}{di}.feats"]
            assert np.array_equal(np.array([[c.x, c.y, c.w, c.h] for c in d["cells"]]), g[f"{name}{di}.cells"])
        with open(tmp_path / f"{name}.pkl", "wb") as fh:
            pickle.dump(docs, fh)
        ds = F.FUNSDBertDataLoaderBoxMaskBoxLabel(str(tmp_path / f"{name}.pkl"), labels, write_labels_file=False)
        for i in range(len(ds)):
            it = ds[i]
            assert it["mask"].dtype == torch.float32 and tuple(it["mask"].shape) == tuple(g[f"{name}{i}.mask"].shape)
            assert np.array_equal(it["mask"].numpy(), g[f"{name}{i}.mask"]), (name, i)             # bit exact
            assert np.array_equal(it["label"].numpy().astype(np.uint8), g[f"{name}{i}.label"]), (name, i)
            assert it["ocr_values"] == [c.ocr_value for c in ds.inp_list[i]["cells"]]

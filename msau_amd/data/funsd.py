"""FUNSD word-level chargrid pipeline: JSON annotations -> cells + one-hot character features ->
per-document chargrid `mask [1,C,H,W]` and `label [1,H,W]`.

API-compatible counterpart of the reference's CPU pipeline; the behaviours restated here (including
the odd ones) are listed in SURVEY.md Appendix B and pinned by tests/golden/funsd/* which were produced
by running the reference itself on synthetic FUNSD-format documents (oracle/gen_goldens.py).

reference: funsd_preprocessing_word_level.py:44-114, data_generator_funsd_bert.py:49-62,149-230,
utils/graph_building_utils.py:208-236,410-417.
"""
from __future__ import annotations

import glob
import json
import os
import pickle
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class CellNode:
    """A text box: position, size and OCR string (graph_building_utils.py:208-236; only the fields the
    chargrid path reads)."""
    current_created_cells = 0

    def __init__(self, x, y, w, h, ocr_value="", is_sub=False):
        self.x, self.y, self.w, self.h = x, y, w, h
        self.ocr_value = ocr_value
        self.is_sub = is_sub
        self.name = "node_" + str(CellNode.current_created_cells)
        CellNode.current_created_cells += 1

    def __repr__(self):
        return f"CellNode({self.x},{self.y},{self.w},{self.h},{self.ocr_value!r})"


def get_list_cells(list_bboxs, ocr_values) -> List[CellNode]:
    return [CellNode(b[0], b[1], b[2], b[3], ocr_values[i]) for i, b in enumerate(list_bboxs)]


# ---- character set (funsd_preprocessing_word_level.py:33-57) --------------------------------------
def get_inv_dict_charset(charset) -> Dict[str, np.ndarray]:
    eye = np.eye(len(charset))
    return {c: eye[i] for i, c in enumerate(charset)}


def get_charset(corpus: str):
    """sorted set of all non-whitespace characters -> (charset, char -> one-hot row)"""
    charset = sorted(set("".join(corpus.split())))
    return charset, get_inv_dict_charset(charset)


def transform_from_charset(text: str, inv_dict_charset) -> np.ndarray:
    """[len(text), |charset|] one-hot matrix; unknown characters stay all-zero rows"""
    mat = np.zeros((len(text), len(inv_dict_charset)))
    for i, c in enumerate(text):
        row = inv_dict_charset.get(c)
        if row is not None:
            mat[i, :] = row
    return mat


def _xywh(box):
    return [box[0], box[1], box[2] - box[0] + 1, box[3] - box[1] + 1]


def get_preprocessed_list_word_msau(dirpath: str, inv_dict_charset=None):
    """Every `*.json` under dirpath -> dict(file_path, word_to_textline, cells_word, cells, labels, ids,
    link, charset_feature).  The charset comes from the LINE texts of this directory unless given
    (funsd_preprocessing_word_level.py:60-114)."""
    docs, corpus = [], []
    for path in glob.glob(os.path.join(dirpath, "*.json")):
        with open(path) as fh:
            form = json.load(fh)["form"]
        line_boxes, line_text, labels, links, ids = [], [], [], [], []
        word_boxes, word_text, word_to_line = [], [], []
        for line in form:
            line_boxes.append(_xywh(line["box"]))
            line_text.append(line["text"])
            for word in line["words"]:
                word_boxes.append(_xywh(word["box"]))
                word_text.append(word["text"])
                word_to_line.append(len(line_boxes) - 1)
            labels.append(line["label"])
            links.append(line["linking"])
            ids.append(line["id"])
        corpus.extend(line_text)
        docs.append({"file_path": path, "word_to_textline": word_to_line,
                     "cells_word": get_list_cells(word_boxes, word_text),
                     "cells": get_list_cells(line_boxes, line_text),
                     "labels": labels, "ids": ids, "link": links})
    if inv_dict_charset is None:
        charset, inv_dict_charset = get_charset(" ".join(corpus))
        print("Charset len: ", len(charset))
    for doc in docs:
        doc["charset_feature"] = [transform_from_charset(c.ocr_value, inv_dict_charset) for c in doc["cells_word"]]
    return docs, inv_dict_charset


# ---- chargrid painting (data_generator_funsd_bert.py:49-62,149-186) --------------------------------
def get_min_max_x_y_w_h(cells) -> Tuple[float, float, float, float, float, float]:
    return (min(c.x for c in cells), min(c.y for c in cells), max(c.x + c.w for c in cells),
            max(c.y + c.h for c in cells), min(c.w for c in cells), min(c.h for c in cells))


def chargrid_geometry(words: List[CellNode]):
    """grid size and the two x scales of the reference: words use min(w / len(text)) (zero-length words
    replaced by the mean ratio), label boxes use the minimum word width"""
    min_x, min_y, max_x, max_y, min_w, min_h = get_min_max_x_y_w_h(words)
    W = int((max_x - min_x) / min_w) + 1
    H = int((max_y - min_y) / min_h) + 1
    ratios = [c.w / len(c.ocr_value) if len(c.ocr_value) != 0 else 0 for c in words]
    mean = sum(ratios) / len(ratios)
    min_scale = min(r if r != 0 else mean for r in ratios)
    return dict(min_x=min_x, min_y=min_y, min_w=min_w, min_h=min_h, min_scale=min_scale, H=H, W=W)


def word_char_boxes(words: List[CellNode], geo) -> List[Tuple[int, int, int, int, int, int]]:
    """(word index, char index, y0, y1, x0, x1) of every painted character, in painting order (later
    boxes overwrite earlier ones); coordinates may exceed the grid and are clipped by the painter"""
    out = []
    for wi, c in enumerate(words):
        x = int((c.x - geo["min_x"]) / geo["min_scale"])
        y = int((c.y - geo["min_y"]) / geo["min_h"])
        w = max(int(c.w / geo["min_scale"]), 1)
        h = max(int(c.h / geo["min_h"]), 1)
        n = len(c.ocr_value) if len(c.ocr_value) != 0 else w
        cw = max(int(w / n), 1)
        for j in range(len(c.ocr_value)):
            out.append((wi, j, y, y + h, x + cw * j, x + cw * (j + 1)))
    return out


def get_box_mask_box_label_word(dataset_instance, idx):
    doc = dataset_instance.inp_list[idx]
    words, lines = doc["cells_word"], doc["cells"]
    geo = chargrid_geometry(words)
    H, W = geo["H"], geo["W"]
    label = np.zeros((H, W)).astype("uint8")
    grid = np.zeros((doc["charset_feature"][0].shape[-1], H, W))
    for wi, j, y0, y1, x0, x1 in word_char_boxes(words, geo):
        grid[:, y0:y1, x0:x1] = doc["charset_feature"][wi][j][:, None, None]
    for li, c in enumerate(lines):
        x = int((c.x - geo["min_x"]) / geo["min_w"])
        y = int((c.y - geo["min_y"]) / geo["min_h"])
        w = max(int(c.w / geo["min_w"]), 1)
        h = max(int(c.h / geo["min_h"]), 1)
        label[y:y + h, x:x + w] = doc["labels"][li] + 1           # 0 stays "unlabelled"
    return {"ocr_values": [c.ocr_value for c in words], "mask": grid, "label": label}


def getitem_box_bert(dataset_instance, idx):
    """data_generator_funsd_bert.py:22-29"""
    doc = dataset_instance.inp_list[idx]
    return {"ocr_values": [c.ocr_value for c in doc["cells"]], "feats": doc["transformer_feature"], "label": doc["labels"]}


def getitem_box_chargrid(dataset_instance, idx):
    """data_generator_funsd_bert.py:30-37"""
    doc = dataset_instance.inp_list[idx]
    return {"ocr_values": [c.ocr_value for c in doc["cells"]], "feats": np.array(doc["charset_feature"]), "label": doc["labels"]}


def line_boxes(lines: List[CellNode]):
    """grid size and (line index, y0, y1, x0, x1) of every text-line box of the dense painter, in painting order
    (data_generator_funsd_bert.py:71-83: both scales are the minimum LINE width / height)"""
    min_x, min_y, max_x, max_y, min_w, min_h = get_min_max_x_y_w_h(lines)
    W = int((max_x - min_x) / min_w) + 1
    H = int((max_y - min_y) / min_h) + 1
    out = []
    for li, c in enumerate(lines):
        x = int((c.x - min_x) / min_w)
        y = int((c.y - min_y) / min_h)
        w = max(int(c.w / min_w), 1)
        h = max(int(c.h / min_h), 1)
        out.append((li, y, y + h, x, x + w))
    return H, W, out


def get_box_mask_box_label(dataset_instance, idx):
    """BERT-embedding chargrid (data_generator_funsd_bert.py:64-93): one feature vector per text LINE painted over its box,
    labels over the same boxes (+1, 0 = unlabelled)"""
    lines = dataset_instance.inp_list[idx]["cells"]
    box = dataset_instance.getitem_box(dataset_instance, idx)
    H, W, boxes = line_boxes(lines)
    label = np.zeros((H, W)).astype("uint8")
    grid = np.zeros((box["feats"].shape[-1], H, W))
    for li, y0, y1, x0, x1 in boxes:
        grid[:, y0:y1, x0:x1] = np.asarray(box["feats"][li])[:, None, None]
        label[y0:y1, x0:x1] = box["label"][li] + 1
    return {"ocr_values": box["ocr_values"], "mask": grid, "label": label}


class FUNSDMaskDataLoader(Dataset):
    """pickle of preprocessed documents -> items {"ocr_values", "mask": [1,C,H,W], "label": [1,H,W]}.
    The label -> id map comes from the FIRST document's label set unless given; like the reference it is
    also written to ./labels (data_generator_funsd_bert.py:188-230)."""

    def __init__(self, funsd_pickle_path, labels_dict=None, getitem_box=None,
                 getitem_mask=get_box_mask_box_label_word, write_labels_file=True):
        with open(funsd_pickle_path, "rb") as fh:
            self.inp_list = pickle.load(fh)
        if labels_dict is None:
            self.labels = {label: i for i, label in enumerate(list(set(self.inp_list[0]["labels"])))}
            if write_labels_file:
                with open("labels", "w") as fh:
                    json.dump(self.labels, fh)
        else:
            self.labels = labels_dict
        for doc in self.inp_list:
            doc["labels"] = np.array([self.labels[lab] for lab in doc["labels"]])
        self.getitem_box = getitem_box
        self.getitem_mask = getitem_mask

    def __len__(self):
        return len(self.inp_list)

    def getitem(self, idx):
        item = self.getitem_mask(self, idx)
        return {"ocr_values": item["ocr_values"],
                "mask": torch.Tensor(item["mask"]).unsqueeze(0),
                "label": torch.Tensor(item["label"]).unsqueeze(0)}

    def __getitem__(self, idx):
        if type(idx) != int:
            return [self.getitem(i) for i in idx[:]]
        return self.getitem(idx)


class FUNSDCharGridDataLoaderBoxMaskBoxLabel(FUNSDMaskDataLoader):
    def __init__(self, funsd_pickle_path, labels_dict=None, **kw):
        super().__init__(funsd_pickle_path, labels_dict=labels_dict, getitem_box=getitem_box_chargrid,
                         getitem_mask=get_box_mask_box_label_word, **kw)


class FUNSDBertDataLoaderBoxMaskBoxLabel(FUNSDMaskDataLoader):
    """data_generator_funsd_bert.py:240-245: the loader BASELINE.json configs[3] names (768-d line embeddings)"""

    def __init__(self, funsd_pickle_path, labels_dict=None, **kw):
        super().__init__(funsd_pickle_path, labels_dict=labels_dict, getitem_box=getitem_box_bert,
                         getitem_mask=get_box_mask_box_label, **kw)

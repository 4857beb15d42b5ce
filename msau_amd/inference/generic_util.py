"""File lists, reading order, one-hot coding, ground-truth reading and CSV reports of the inference tools
(reference: inference/generic_util.py).  The drawing helpers of the reference need OpenCV + a TrueType font and only
produce debug images; they are outside the hot path and are not rebuilt (KVModel.predict returns `None` for the image).
"""
from __future__ import annotations

import csv
import json
import os

import numpy as np

from .morph_util import union_boxes


def read_image_list(pathToList, prefix=None):
    """one path per line; '#' lines are comments, and a '#' line shorter than 3 characters ends the list
    (generic_util.py:14-36)"""
    names = []
    with open(pathToList, "r") as fh:
        for line in fh:
            if line[0] == "#":
                if len(line) < 3:
                    break
                continue
            name = line[:-1] if line[-1] == "\n" else line
            names.append(name if prefix is None else prefix + name)
    return names


def glob_folder(path, extension):
    """{basename up to the first '.': full path} for every file under `path` ending in `extension`; first one wins"""
    found = {}
    for dirpath, _, filenames in os.walk(path):
        for fn in filenames:
            if not fn.endswith("{}".format(extension)):
                continue
            key = os.path.basename(fn).split(".")[0]
            if key in found:
                print("Duplicated file name: {}, existing file: {}".format(os.path.join(dirpath, fn), found[key]))
            else:
                found[key] = os.path.join(dirpath, fn)
    return found


def _comes_before(box, lead):
    """does `box` displace the current reading-order leader `lead`?  (generic_util.py:66-84)"""
    lx1, ly1, lx2, ly2 = lead["box"]
    x1, y1, x2, y2 = box["box"]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    if cy <= (ly1 + ly2) / 2 - (y2 - y1) / 2:
        return True                                    # clearly above: its centre is half its own height higher
    return cx < lx2 and cy < ly2                       # or starts inside the leader's lower-right quadrant


def sort_box_reading_order(boxes):
    """Selection sort into reading order: repeatedly sweep the remaining cells once, letting each cell that
    `_comes_before` the current leader take over, and emit the final leader.  The input list is consumed, as in the
    reference (generic_util.py:52-95)."""
    if len(boxes) == 0:
        return boxes
    ordered = []
    while len(boxes) > 1:
        lead = boxes[0]
        for cand in boxes[1:]:
            if _comes_before(cand, lead):
                lead = cand
        ordered.append(lead)
        boxes.remove(lead)
    ordered.append(boxes[0])
    return ordered


def to_categorical(target_vector, n_labels):
    """integer mask -> uint8 one-hot with a trailing class axis (generic_util.py:97-98)"""
    return np.eye(n_labels, dtype="B")[target_vector]


def read_json_gt(json_path, scale=1.0, offset=(0, 0)):
    """Ground truth of a layout JSON -> {class id: ([union box, box...], joined text)}; the lines' boxes are shifted
    by `offset`, scaled and truncated in place; only lines with value > 0 and type > 0 count, their class is
    value + 1 (generic_util.py:214-251)."""
    with open(json_path, "r") as fh:
        doc = json.load(fh)
    _ = [int(d * scale) for d in doc["img_shape"]]       # the reference requires the key (generic_util.py:218)
    ox, oy = offset
    groups = {}
    for line in doc["lines"]:
        x1, y1, x2, y2 = line["box"]
        line["box"] = [int((x1 - ox) * scale), int((y1 - oy) * scale), int((x2 - ox) * scale), int((y2 - oy) * scale)]
        if line["value"] > 0 and line["type"] > 0:
            groups.setdefault(line["value"] + 1, []).append(line)
    answers = {}
    for cls, members in groups.items():
        members = sort_box_reading_order(members)
        rects = [m["box"] for m in members]
        if cls != 1:
            answers[cls] = ([union_boxes(rects)] + rects, "".join(m["text"] for m in members))
    return answers


def write_csv_report_by_row(file_list, kv_results, output_path, ca_map=None):
    """one CSV row per (file, field)"""
    with open(output_path, "w") as fh:
        out = csv.writer(fh, delimiter=",")
        out.writerow(["file_name", "field_name", "predict", "correct_answer", "T/F"])
        fields = kv_results[0].keys()
        for path, item in zip(file_list, kv_results):
            stem = os.path.basename(path).split(".")[0]
            for k in fields:
                if ca_map is not None:
                    out.writerow([stem, k, item[k], ca_map[stem][k][1], ca_map[stem][k][0]])
                else:
                    out.writerow([stem, k, item[k], "(none)", "(none)"])


def write_csv_report(file_list, kv_results, output_path, ca_map=None):
    """one CSV row per file, one column per field (plus its correct answer when `ca_map` is given)"""
    with open(output_path, "w") as fh:
        out = csv.writer(fh, delimiter=",")
        fields = list(kv_results[0].keys())
        header = ["filename"]
        for k in fields:
            header.append(k)
            if ca_map is not None:
                header.append("correct_answer")
        out.writerow(header)
        for path, item in zip(file_list, kv_results):
            stem = os.path.basename(path).split(".")[0]
            row = [stem]
            for k in fields:
                row.append(item[k])
                if ca_map is not None:
                    row.append(ca_map[stem][k][1])
            out.writerow(row)

"""Key-value inference model (reference: inference/kv_model.py `KVModel`).

Host side as in the reference -- character-id / line-id / character-position masks from the layout+OCR JSON, the
connected-component post-processing that turns the class map into field texts -- with the network between them run
by the forward-only HIP plan:

    reference (kv_model.py:274-279, 305-309)              here
    to_categorical(mask) -> float [1,C,H,W] -> .cuda()    int32 id mask [1,H,W] -> device, one-hot painted by a kernel
    net(batch_x) (autograd-free forward)                  MSAUWrapper.predict_nhwc: buffers reused, nothing saved
    softmax, transpose to NHWC, .cpu()                    softmax + argmax in the end conv's epilogue, NHWC output

There is no CPU network path: without the HIP library / a GPU `predict` raises (msau_amd.model).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .generic_util import read_json_gt, sort_box_reading_order, to_categorical  # noqa: F401  (API parity)
from .morph_util import IoU, area, connected_components, intersect_boxes, r_closing, union_boxes, ycenter
from .postprocess import CLASS_NAMES, post_process_kv  # noqa: F401


def _obj_box(o):
    return [o[1].start, o[0].start, o[1].stop, o[0].stop]


class KVModel:
    """Inference key-value model; attribute and method names follow kv_model.py:15-387."""

    default_config = {"scale": 3.0, "charset": "", "model_kv": "", "n_class": 0}

    # fields whose value may span several text lines (kv_model.py:155)
    multiple_lines_fields = (5, 11)

    def __init__(self, net=None):
        self.net = net                      # the reference expects the caller to set this before load()
        self.scale = None
        self.tok_to_id, self.id_to_tok = None, None
        self.blank_idx = 1
        self.n_token = 1
        self.charset = ""
        self.n_class = 1

    # ---- configuration (kv_model.py:36-59) ---------------------------------------------------------------------
    def load(self, **config):
        """config: model_weight (state_dict file), charset (text file or None), n_class; optional `model_kwargs`
        and `dtype` build the network here when the caller did not supply one."""
        self.set_charset(config["charset"])
        self.n_class = config["n_class"]
        self.scale = self.default_config["scale"]
        if self.net is None:
            from ..model import MSAUWrapper
            kw = dict(config.get("model_kwargs", {}))
            if "dtype" in config:
                kw["dtype"] = config["dtype"]
            self.net = MSAUWrapper(channels=self.n_token, n_class=self.n_class, model_kwargs=kw)
        if torch.cuda.is_available():
            self.net = self.net.cuda()
        self.net.load_weights(config["model_weight"])
        self.net.eval()

    def set_charset(self, path_charset):
        """token 0 is ' ' (also the background id), token 1 is '$' (unknown / blank), then the file's characters"""
        if path_charset is None:
            self.charset = None
            return
        with open(path_charset, "r") as fh:
            self.charset = " " + "$" + fh.read()
        self.blank_idx = 1
        self.tok_to_id = {tok: i for i, tok in enumerate(self.charset)}      # later duplicates win, as in the reference
        self.id_to_tok = {i: tok for tok, i in self.tok_to_id.items()}
        self.n_token = len(self.tok_to_id)

    @staticmethod
    def _read_json_layout_ocr(json_path):
        with open(json_path, "r") as fh:
            return json.load(fh)

    # ---- input masks (kv_model.py:83-148) ------------------------------------------------------------------------
    def _generate_masks_from_label(self, label_path):
        """-> (char-id mask, line-id mask, char-position mask  [uint16, H x W], lines (boxes rewritten to grid
        coordinates), scale, background pad, (min_x, min_y, max_x, max_y) of the text in page coordinates).
        The grid is the text bounding box grown by 3 median line heights, scaled so that a median line is 3 px high.
        Digits are folded to '0'; character k of a line occupies [x1 + k*cw, x1 + k*cw + 0.9*cw) (at least 1 px,
        at most 1.2 line heights); later lines / characters overwrite earlier ones."""
        doc = self._read_json_layout_ocr(label_path)
        lines = doc["lines"]
        left, top = min(l["box"][0] for l in lines), min(l["box"][1] for l in lines)
        right, bottom = max(l["box"][2] for l in lines), max(l["box"][3] for l in lines)
        text_bbox = (left, top, right, bottom)
        median_h = np.median([l["box"][3] - l["box"][1] for l in lines])
        bg_pad = int(median_h * 3)
        left, top, right, bottom = left - bg_pad, top - bg_pad, right + bg_pad, bottom + bg_pad
        scale = 3.0 / median_h
        shape = [int((bottom - top) * scale), int((right - left) * scale)]
        char_ids = np.zeros(shape, dtype="uint16")
        line_ids = np.zeros(shape, dtype="uint16")
        char_pos = np.zeros(shape, dtype="uint16")
        for li, line in enumerate(lines):
            _type, _value = line["type"], line["value"]                 # required keys (kv_model.py:115)
            bx1, by1, bx2, by2 = line["box"]
            x1, y1 = int((bx1 - left) * scale), int((by1 - top) * scale)
            x2, y2 = int((bx2 - left) * scale), int((by2 - top) * scale)
            line["box"] = [x1, y1, x2, y2]
            text = "".join("0" if ch.isdigit() else ch for ch in line["text"])
            if not text:
                continue
            pitch = max(1.0 * (x2 - x1) / len(text), 1.0)
            glyph_w = min(max(0.9 * pitch, 1.0), int((y2 - y1) * 1.2))
            line_ids[y1:y2, x1:x2] = li + 1
            for k, ch in enumerate(text):
                xs = x1 + k * pitch
                a, b = int(xs), int(xs + glyph_w)
                char_ids[y1:y2, a:b] = self.tok_to_id.get(ch, self.blank_idx)
                line_ids[y1:y2, a:b] = li + 1
                char_pos[y1:y2, a:b] = k + 1
        return char_ids, line_ids, char_pos, lines, scale, bg_pad, text_bbox

    # ---- post-processing (kv_model.py:150-261) ------------------------------------------------------------------
    @staticmethod
    def _extract_value(line_mask, char_mask, label_lines, pred_mask, num_classes, pred_class=None):
        """Class map -> per-class (text, [region box], intersection box, union box) and the cleaned class mask.

        For every class c >= 2: close the argmax region with a 1x3 element, take its connected components and keep
        the one with the largest bounding box (multi-line fields: the top-most one as the main region plus every
        other component with a box area > 5); a main box of area < 5 drops the class.  The text lines under the kept
        regions belong to the field.  A line claimed by one field contributes its whole text; a line shared between
        fields contributes the character span its kept pixels cover (start widened by one character, end snapped to
        the line end when within 3 characters of it).  `pred_class` may carry a precomputed argmax (the device head's)."""
        n_class = pred_mask.shape[2]
        multi = KVModel.multiple_lines_fields
        values = [("", None, None, None)] * n_class
        if pred_class is None:
            pred_class = np.argmax(pred_mask, axis=-1)
        kept = np.zeros(pred_mask.shape)
        kept[:, :, 0] = pred_mask[:, :, 0]
        claims = [0] * (len(label_lines) + 1)
        field_lines = [[] for _ in range(num_classes + 1)]
        field_boxes = [[] for _ in range(num_classes + 1)]
        for i, line in enumerate(label_lines):
            line["id"] = i + 1

        def lines_under(labels, comp):
            return [v for v in np.unique(line_mask[labels == comp + 1]) if v > 0]

        for c in range(2, n_class):
            labels, objects = connected_components(r_closing(pred_class == c, (1, 3)))
            if len(objects) == 0:
                continue
            if c in multi:
                order = np.argsort([-ycenter(o) for o in objects])       # last = top-most
            else:
                order = np.argsort([area(o) for o in objects])           # last = largest box
            main = order[-1]
            if area(objects[main]) < 5:
                continue
            extra = []
            if c in multi:
                for comp in order[:-1]:
                    if area(objects[comp]) > 5:
                        extra.append(comp)
                        field_boxes[c].append(_obj_box(objects[comp]))
            field_boxes[c].append(_obj_box(objects[main]))
            ids = lines_under(labels, main)
            for comp in extra:
                ids += lines_under(labels, comp)
                kept[:, :, c][labels == comp + 1] = 1
            field_lines[c] = list(set(ids))
            for v in ids:
                claims[v] += 1
            kept[:, :, c][labels == main + 1] = 1

        for c in range(2, n_class):
            if len(field_lines[c]) == 0:
                continue
            ordered = sort_box_reading_order([label_lines[i - 1] for i in field_lines[c] if i > 0])
            text, rects = "", []
            for line in ordered:
                rects.append(line["box"])
                if claims[line["id"]] <= 1:
                    text += line["text"]
                else:
                    x1, y1, x2, y2 = line["box"]
                    covered = set(np.unique(char_mask[y1:y2, x1:x2][kept[:, :, c][y1:y2, x1:x2] > 0]))
                    covered.discard(0)
                    if len(covered) == 0:
                        continue                                        # (also skips the line break below)
                    first, last = min(covered), max(covered)
                    if last > len(line["text"]) - 3:
                        last = len(line["text"]) + 1
                    text += line["text"][first - 2 if first >= 2 else 0: last - 1]
                if c in multi:
                    text += "\n"
            if len(text) > 0 and text[-1] == "\n":
                text = text[:-1]
            merged = union_boxes(rects)
            values[c] = (text, [field_boxes[c][-1]], intersect_boxes(field_boxes[c] + [merged]),
                         union_boxes(field_boxes[c] + [merged]))
        return values, kept

    # ---- evaluation bookkeeping the reference does inside its drawing routine (generic_util.py:166-189) ----------
    @staticmethod
    def _count_predictions(values, n_class, eval_results, correct_answers):
        for value_id in range(1, n_class):
            if values[value_id][1] is None:
                continue
            for box in values[value_id][1]:
                eval_results[value_id]["num_pred"] += 1
                if correct_answers is not None:
                    gt = correct_answers[value_id][0][:1] if value_id in correct_answers else []
                    if any(IoU(box, g) > 0.7 for g in gt):
                        eval_results[value_id]["num_correct"] += 1

    # ---- the network (kv_model.py:274-309) ------------------------------------------------------------------------
    def _run_net(self, input_mask):
        """char-id mask [H,W] -> (pred fp32 [H,W,n_class] numpy, argmax uint8 [H,W] numpy) through the HIP plan"""
        if not torch.cuda.is_available():
            raise RuntimeError("KVModel.predict runs the network through libmsau_hip.so on an MI355X; no GPU is "
                               "visible and there is no CPU fallback")
        ids = torch.from_numpy(input_mask.astype(np.int32))[None]
        pred, amax = self.net.predict_nhwc(ids=ids.cuda())
        return pred[0].cpu().numpy(), amax[0].cpu().numpy()

    def predict(self, data, debug_info=None, label_path=None, eval_results=None):
        """data = (layout JSON path, page image or None) -> ({field: text}, debug image).
        The debug rendering of the reference (OpenCV + PIL drawing) is not part of this build: the second result is
        always None; everything that feeds `kv_results` and `eval_results` is computed as in kv_model.py:264-347."""
        json_path, _debug_im = data
        input_im, line_mask, char_mask, label_lines, scale, bg_pad, (min_x, min_y, _max_x, _max_y) = \
            self._generate_masks_from_label(json_path)

        correct_answers = None
        if label_path is not None:
            try:
                correct_answers = read_json_gt(label_path, scale=scale, offset=(min_x - bg_pad, min_y - bg_pad))
            except IOError as e:
                print("Error reading CA", e)
        if correct_answers is not None:
            for value_id in correct_answers:
                eval_results[value_id]["num_label"] += 1

        a_pred, a_cls = self._run_net(input_im)
        values, _pred_mask = self._extract_value(line_mask, char_mask, label_lines, a_pred, self.n_class,
                                                 pred_class=a_cls.astype(np.int64))
        kv_results = post_process_kv(values)
        if eval_results is not None:
            self._count_predictions(values, self.n_class, eval_results, correct_answers)
        return kv_results, None

    def run_test(self, list_inf, out_dir, label_dir=None, img_dir=None):
        """predict every layout JSON of `list_inf`; with `label_dir`, print per-class counts and precision / recall /
        F1 over region boxes (kv_model.py:350-387).  Unlike the reference a missing page image does not skip the
        document, because no debug image is drawn."""
        eval_results = [{"num_pred": 0, "num_correct": 0, "num_label": 0} for _ in range(self.n_class)]
        kv_results = []
        for file_path in list_inf:
            basename = os.path.basename(file_path).split(".")[0]
            label_path = os.path.join(label_dir, basename + ".json") if label_dir is not None else None
            result, _ = self.predict((file_path, None), debug_info=("", None), label_path=label_path,
                                     eval_results=eval_results)
            print(basename)
            print(result)
            kv_results.append(result)
        if label_dir is not None:
            for c, count in enumerate(eval_results):
                if count["num_pred"] > 0 or count["num_label"] > 0:
                    print(c, count)
            n_correct = np.sum([c["num_correct"] for c in eval_results])
            recall = 1.0 * n_correct / np.sum([c["num_label"] for c in eval_results])
            precision = 1.0 * n_correct / np.sum([c["num_pred"] for c in eval_results])
            f1 = 2 * recall * precision / (recall + precision)
            print("Precision : {}   Recall : {}    F1-score : {}".format(precision, recall, f1))
        return kv_results

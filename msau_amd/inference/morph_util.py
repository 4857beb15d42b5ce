"""Mask / box helpers of the post-processing (reference: inference/morph_util.py).  CPU, scipy.ndimage.
Objects are the slice pairs `scipy.ndimage.find_objects` returns: (rows, cols)."""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


# ---- connected components and object geometry (morph_util.py:13-63) ------------------------------------------------
def connected_components(image, thres=0):
    """4-connected labelling -> (label image, list of bounding slices); `thres` > 0 binarises first."""
    binary = image > thres if thres > 0 else image
    labels, _ = ndi.label(binary)
    return labels, ndi.find_objects(labels)


def width(s):
    return s[1].stop - s[1].start


def height(s):
    return s[0].stop - s[0].start


def area(s):
    return width(s) * height(s)


def min_dim(b):
    return min(width(b), height(b))


def max_dim(b):
    return max(width(b), height(b))


def xcenter(s):
    return np.mean([s[1].stop, s[1].start])


def ycenter(s):
    return np.mean([s[0].stop, s[0].start])


def aspect_normalized(s):
    a = height(s) * 1.0 / width(s)
    return 1.0 / a if a < 1 else a


# ---- rectangular morphology; outside the image counts as 0 (morph_util.py:66-86) -----------------------------------
def r_dilation(image, size, origin=0):
    return ndi.maximum_filter(image, size, origin=origin, mode="constant")


def r_erosion(image, size, origin=0):
    return ndi.minimum_filter(image, size, origin=origin, mode="constant")


def r_opening(image, size, origin=0):
    return r_dilation(r_erosion(image, size, origin=origin), size, origin=origin)


def r_closing(image, size, origin=0):
    """dilate then erode; the reference ignores `origin` here (morph_util.py:82-86) and so does this."""
    return r_erosion(r_dilation(image, size), size)


# ---- boxes [x1, y1, x2, y2] (morph_util.py:88-205) ---------------------------------------------------------------
def intersect_boxes(boxes):
    if not boxes:
        return None
    return [max(b[0] for b in boxes), max(b[1] for b in boxes), min(b[2] for b in boxes), min(b[3] for b in boxes)]


def union_boxes(boxes):
    if not boxes:
        return None
    return [min(b[0] for b in boxes), min(b[1] for b in boxes), max(b[2] for b in boxes), max(b[3] for b in boxes)]


def rect_area(rect):
    return (rect[2] - rect[0]) * (rect[3] - rect[1])


def intersect_area(box_a, box_b, min_thresh=2):
    """inclusive-pixel overlap area, 0 unless the overlap is at least `min_thresh` wide and high"""
    left, right = max(box_a[0], box_b[0]), min(box_a[2], box_b[2])
    top, bottom = max(box_a[1], box_b[1]), min(box_a[3], box_b[3])
    if left <= right - min_thresh and top <= bottom - min_thresh:
        return 1.0 * (right - left + 1) * (bottom - top + 1)
    return 0.0


def _rect(entry, with_meta):
    return entry[0] if with_meta else entry


def filter_overlap_boxes(boxes, with_meta=False, return_indices=False):
    """drop boxes fully inside a not-narrower box that has not itself been dropped (scan order matters)"""
    if len(boxes) < 2:
        return boxes
    dropped = [False] * len(boxes)
    for i in range(len(boxes)):
        x1, y1, x2, y2 = _rect(boxes[i], with_meta)
        for j in range(len(boxes)):
            if i == j or dropped[j]:
                continue
            x3, y3, x4, y4 = _rect(boxes[j], with_meta)
            if abs(x1 - x2) <= abs(x3 - x4) and x1 >= x3 and x2 <= x4 and y1 >= y3 and y2 <= y4:
                dropped[i] = True
                break
    if return_indices:
        return dropped
    return [b for b, d in zip(boxes, dropped) if not d]


def filter_overlap_boxes_bigger(boxes, with_meta=False, intersect_thres=0.9, min_area=0, return_indices=False):
    """drop the smaller of two boxes when the overlap covers more than `intersect_thres` of it"""
    if len(boxes) < 2:
        return boxes
    dropped = [False] * len(boxes)
    for i in range(len(boxes)):
        ra = _rect(boxes[i], with_meta)
        ai = rect_area(ra)
        for j in range(len(boxes)):
            if i == j:
                continue
            rb = _rect(boxes[j], with_meta)
            aj = rect_area(rb)
            small = min(ai, aj)
            if ai <= aj and intersect_area(ra, rb, min_thresh=0) > intersect_thres * small and small > min_area:
                dropped[i] = True
                break
    if return_indices:
        return dropped
    return [b for b, d in zip(boxes, dropped) if not d]


def check_intersect_boxes(boxes, scale):
    return [any(i != j and intersect_area(boxes[i], boxes[j]) > scale * 15 for j in range(len(boxes)))
            for i in range(len(boxes))]


def is_overlap(big_box, small_box, pad=2):
    """`small_box` lies inside `big_box` grown by `pad`"""
    x1, y1, x2, y2 = small_box
    x3, y3, x4, y4 = big_box
    return x1 >= x3 - pad and x2 <= x4 + pad and y1 >= y3 - pad and y2 <= y4 + pad


def IoU(rect_a, rect_b):
    """overlap relative to rect_a's area -- not a symmetric IoU (morph_util.py:197-201)"""
    return 1.0 * intersect_area(rect_a, rect_b, min_thresh=0) / rect_area(rect_a)


def scale_rect(rect, scale_factor):
    return [int(v * scale_factor) for v in rect]


def scale_pts(pts, scale_factor):
    return [[int(v * scale_factor) for v in pt] for pt in pts]

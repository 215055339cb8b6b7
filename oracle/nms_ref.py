"""CPU oracle of the rotated BEV NMS behind CenterHead's post-processing (TEST INFRASTRUCTURE ONLY).

numpy (float32) restatement of pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:
  * ``box_overlap``  :107-207  -- intersection polygon of two rotated rectangles: edge crossings (``intersection``
    :63-92, with the bounding-box pre-test ``check_rect_cross`` :42-48), corners of one box inside the other with a
    1e-2 margin (``check_in_box2d`` :50-61), bubble sort by ``atan2`` around the mean point (:98-100,:178-187), fan area;
  * ``iou_bev``      :209-217  -- overlap / max(sa + sb - overlap, 1e-8);
  * ``nms_kernel`` + the host loop of ``nms_gpu`` (:236-278, iou3d_nms.cpp:90-135): boxes in descending score order,
    a box suppresses every LATER box whose IoU exceeds the threshold (strict >), greedy keep.
Parity status: "parity unpinned" against real CUDA output (no nvcc / NVIDIA GPU here, the reference ships no vectors);
pinned to hand-derived cases (identical boxes, disjoint boxes, axis-aligned overlaps with known areas) in tests/.
"""
import math

import numpy as np

F = np.float32
EPS = F(1e-8)   # iou3d_nms_kernel.cu:15
MARGIN = F(1e-2)  # :52


def _cross3(p1, p2, p0):  # :38-40
    return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])


def _intersection(p1, p0, q1, q0):  # :63-92
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1, s2, s3, s4 = _cross3(q0, p1, p0), _cross3(p1, q1, p0), _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(s5 - s1) > EPS:
        return (F((s5 * q0[0] - s1 * q1[0]) / (s5 - s1)), F((s5 * q0[1] - s1 * q1[1]) / (s5 - s1)))
    a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
    a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
    D = a0 * b1 - a1 * b0
    return (F((b0 * c1 - b1 * c0) / D), F((a1 * c0 - a0 * c1) / D))


def _in_box(box, p):  # :50-61
    c, s = F(math.cos(-float(box[6]))), F(math.sin(-float(box[6])))
    rx = (p[0] - box[0]) * c + (p[1] - box[1]) * (-s)
    ry = (p[0] - box[0]) * s + (p[1] - box[1]) * c
    return abs(rx) < box[3] / F(2) + MARGIN and abs(ry) < box[4] / F(2) + MARGIN


def _corners(box):  # :114-150
    hx, hy = box[3] / F(2), box[4] / F(2)
    c, s = F(math.cos(float(box[6]))), F(math.sin(float(box[6])))
    # :109-111 axis-aligned corners first, :94-98 rotate_around_center subtracts the centre again (fp32 each step)
    x1, x2, y1, y2 = F(box[0] - hx), F(box[0] + hx), F(box[1] - hy), F(box[1] + hy)
    out = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        out.append((F(F(F(px - box[0]) * c) + F(F(py - box[1]) * (-s))) + box[0],
                    F(F(F(px - box[0]) * s) + F(F(py - box[1]) * c)) + box[1]))
    out = [(F(a), F(b)) for a, b in out]
    return out + [out[0]]


def box_overlap(a, b):
    a, b = np.asarray(a, F), np.asarray(b, F)
    ca, cb = _corners(a), _corners(b)
    pts = []
    for i in range(4):
        for j in range(4):
            p = _intersection(ca[i + 1], ca[i], cb[j + 1], cb[j])
            if p is not None:
                pts.append(p)
    for k in range(4):
        if _in_box(a, cb[k]):
            pts.append(cb[k])
        if _in_box(b, ca[k]):
            pts.append(ca[k])
    n = len(pts)
    if n == 0:
        return F(0.0)
    cx, cy = F(sum(p[0] for p in pts) / F(n)), F(sum(p[1] for p in pts) / F(n))
    ang = lambda p: F(math.atan2(float(p[1] - cy), float(p[0] - cx)))  # noqa: E731
    for j in range(n - 1):  # :178-187 bubble sort, descending angle moves right
        for i in range(n - j - 1):
            if ang(pts[i]) > ang(pts[i + 1]):
                pts[i], pts[i + 1] = pts[i + 1], pts[i]
    area = F(0.0)
    for k in range(n - 1):
        ux, uy = pts[k][0] - pts[0][0], pts[k][1] - pts[0][1]
        vx, vy = pts[k + 1][0] - pts[0][0], pts[k + 1][1] - pts[0][1]
        area = F(area + (ux * vy - uy * vx))
    return F(abs(area) / F(2.0))


def iou_bev(a, b):  # :209-217
    sa, sb = F(a[3]) * F(a[4]), F(b[3]) * F(b[4])
    ov = box_overlap(a, b)
    return F(ov / max(sa + sb - ov, EPS))


def nms(boxes, scores, thresh, pre_maxsize=None):
    """iou3d_nms_utils.nms_gpu (:83-98): returns the kept indices into `boxes`, best score first."""
    boxes = np.asarray(boxes, F)
    order = np.argsort(-np.asarray(scores, np.float64), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order]
    n = b.shape[0]
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        for j in range(i + 1, n):
            if not removed[j] and iou_bev(b[i], b[j]) > F(thresh):
                removed[j] = True
    return order[np.array(keep, dtype=np.int64)]

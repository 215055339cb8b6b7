// nms_bev.hip -- rotated bird's-eye-view NMS behind CenterHead's post-processing.
//
// Replaces iou3d_nms_cuda.nms_gpu (ref: pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:90-135 + iou3d_nms_kernel.cu:236-278):
// the reference computes a suppression bit matrix on the GPU (one THREAD per row and 64-column block, a serial loop over
// the 64 columns), copies it to the host and walks it on the CPU.  Here
//   k_nms_mask : one WAVEFRONT per (64-row block, 64-column block); lane = column, the row loop is wave-uniform, so one
//                `__ballot` IS the 64-bit mask word of a row -- no per-thread bit assembly;
//   k_nms_scan : the greedy walk on the device (one wavefront, the removed-bits words live in its lanes, 16 mask rows
//                in flight at a time); only the kept indices and their count are read back.
// The overlap of two rotated rectangles follows the reference's procedure so that the same boxes are kept: edge
// crossings (bounding-box pre-test, strict sign test; ref :42-48,:63-92), corners of one box inside the other with a
// 1e-2 margin (:50-61), points ordered by atan2 around their mean (:98-100,:178-187), fan area (:199-206);
// IoU = overlap / max(sa + sb - overlap, 1e-8) (:209-217); a box suppresses LATER boxes with IoU > thresh (strict).
#include "common.hip.h"

#define NMS_EPS 1e-8f
#define NMS_MARGIN 1e-2f

struct P2 {
    float x, y;
};

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool seg_cross(P2 p1, P2 p0, P2 q1, P2 q0, P2 &out) {
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > NMS_EPS) {
        out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        out.x = (b0 * c1 - b1 * c0) / D;
        out.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

struct Box7 {
    float x, y, z, dx, dy, dz, r;
};

__device__ __forceinline__ bool inside(const Box7 &b, P2 p) {
    const float c = cosf(-b.r), s = sinf(-b.r);
    const float rx = (p.x - b.x) * c + (p.y - b.y) * (-s), ry = (p.x - b.x) * s + (p.y - b.y) * c;
    return fabsf(rx) < b.dx / 2 + NMS_MARGIN && fabsf(ry) < b.dy / 2 + NMS_MARGIN;
}

__device__ __forceinline__ void corners_of(const Box7 &b, P2 (&c)[5]) {
    // the reference's arithmetic, rounding for rounding (iou3d_nms_kernel.cu:109-111,125-128: axis-aligned corners
    // x -+ dx/2 FIRST; :94-98 rotate_around_center subtracts the centre again) -- for large |x| the corners differ in the
    // last bits from (-+ dx/2) rotated directly, enough to flip a pair whose IoU sits at the threshold
    const float hx = b.dx / 2, hy = b.dy / 2, cs = cosf(b.r), sn = sinf(b.r);
    const float x1 = b.x - hx, x2 = b.x + hx, y1 = b.y - hy, y2 = b.y + hy;
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k].x = (px[k] - b.x) * cs + (py[k] - b.y) * (-sn) + b.x;
        c[k].y = (px[k] - b.x) * sn + (py[k] - b.y) * cs + b.y;
    }
    c[4] = c[0];
}

__device__ float rect_overlap(const Box7 &a, const Box7 &b) {
    P2 ca[5], cb[5], pts[24];
    corners_of(a, ca);
    corners_of(b, cb);
    int n = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 p;
            if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], p)) {
                pts[n++] = p;
                sx += p.x;
                sy += p.y;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (inside(a, cb[k])) {
            pts[n++] = cb[k];
            sx += cb[k].x;
            sy += cb[k].y;
        }
        if (inside(b, ca[k])) {
            pts[n++] = ca[k];
            sx += ca[k].x;
            sy += ca[k].y;
        }
    }
    if (n == 0) return 0.f;
    const float mx = sx / n, my = sy / n;
    float ang[24];
    for (int k = 0; k < n; ++k) ang[k] = atan2f(pts[k].y - my, pts[k].x - mx);
    for (int j = 0; j < n - 1; ++j)  // the reference's bubble sort (stable for equal angles)
        for (int i = 0; i < n - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const float t = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = t;
                const P2 q = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = q;
            }
    float area = 0.f;
    for (int k = 0; k < n - 1; ++k)
        area += (pts[k].x - pts[0].x) * (pts[k + 1].y - pts[0].y) - (pts[k].y - pts[0].y) * (pts[k + 1].x - pts[0].x);
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const Box7 &a, const Box7 &b) {
    const float sa = a.dx * a.dy, sb = b.dx * b.dy, ov = rect_overlap(a, b);
    return ov / fmaxf(sa + sb - ov, NMS_EPS);
}

__device__ __forceinline__ Box7 load_box(const float *boxes, int i) {
    const float *p = boxes + (size_t)i * 7;
    return Box7{p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
}

// grid (column blocks, row blocks), one wavefront each; mask (n, col_blocks) words
__global__ void __launch_bounds__(MSSVT_WAVE) k_nms_mask(int n, float thresh, const float *boxes, unsigned long long *mask) {
    const int cb = blockIdx.x, rb = blockIdx.y, lane = lane_id();
    const int col_blocks = gridDim.x;
    const int j = cb * MSSVT_WAVE + lane;
    const bool col_ok = j < n;
    const Box7 bj = load_box(boxes, col_ok ? j : 0);
    const int r_end = min(n, (rb + 1) * MSSVT_WAVE);
    for (int i = rb * MSSVT_WAVE; i < r_end; ++i) {
        unsigned long long word = 0ull;
        if (cb >= rb) {  // only later boxes can be suppressed (blocks left of the diagonal stay empty)
            const Box7 bi = load_box(boxes, i);
            const bool hit = col_ok && j > i && iou_bev(bi, bj) > thresh;
            word = __ballot(hit);
        }
        if (lane == 0) mask[(size_t)i * col_blocks + cb] = word;
    }
}

// one wavefront; lane l owns the removed-bits words l, l + 64, ... (NMS_WPL per lane)
#define NMS_WPL 4
__global__ void __launch_bounds__(MSSVT_WAVE) k_nms_scan(int n, int col_blocks, const unsigned long long *mask, int *keep,
                                                         int *num_keep) {
    const int lane = lane_id();
    unsigned long long remv[NMS_WPL];
#pragma unroll
    for (int k = 0; k < NMS_WPL; ++k) remv[k] = 0ull;
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 16) {
        // the mask rows of the next 16 boxes travel together (they do not depend on the decisions)
        unsigned long long rows[16][NMS_WPL];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < NMS_WPL; ++k) {
                const int w = lane + MSSVT_WAVE * k;
                rows[r][k] = (i0 + r < n && w < col_blocks) ? mask[(size_t)(i0 + r) * col_blocks + w] : 0ull;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + r;
            if (i >= n) continue;
            const int w = i >> 6;  // word of box i: lane w % 64, register w / 64
            unsigned long long mine = 0ull;
#pragma unroll
            for (int k = 0; k < NMS_WPL; ++k) mine = (w / MSSVT_WAVE) == k ? remv[k] : mine;
            const unsigned lo = __shfl((unsigned)(mine & 0xFFFFFFFFull), w % MSSVT_WAVE);
            const unsigned hi = __shfl((unsigned)(mine >> 32), w % MSSVT_WAVE);
            const unsigned long long word = ((unsigned long long)hi << 32) | lo;
            if (!((word >> (i & 63)) & 1ull)) {  // wave-uniform
                if (lane == 0) keep[cnt] = i;
                ++cnt;
#pragma unroll
                for (int k = 0; k < NMS_WPL; ++k) remv[k] |= rows[r][k];
            }
        }
    }
    if (lane == 0) *num_keep = cnt;
}

extern "C" long long mssvt_nms_workspace_bytes(int num_boxes) {
    const long long cbk = (num_boxes + MSSVT_WAVE - 1) / MSSVT_WAVE;
    return (long long)num_boxes * cbk * 8;
}

extern "C" int mssvt_nms_bev(int num_boxes, const float *boxes_sorted, float thresh, void *workspace, int *keep,
                             int *num_keep_dev, void *stream) {
    if (num_boxes < 0 || !keep || !num_keep_dev) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (num_boxes == 0) return (int)hipMemsetAsync(num_keep_dev, 0, sizeof(int), st);
    if (!boxes_sorted || !workspace) return MSSVT_E_BADARG;
    const int cbk = divup(num_boxes, MSSVT_WAVE);
    if (cbk > MSSVT_WAVE * NMS_WPL) return MSSVT_E_TOOLARGE;  // 16384 boxes
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(workspace);
    k_nms_mask<<<dim3(cbk, cbk), MSSVT_WAVE, 0, st>>>(num_boxes, thresh, boxes_sorted, mask);
    k_nms_scan<<<1, MSSVT_WAVE, 0, st>>>(num_boxes, cbk, mask, keep, num_keep_dev);
    return mssvt_launch_status();
}

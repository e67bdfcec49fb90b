// The node's image pre-processing (vision_semantic_segmentation_node.py:83-98) for ONE output pixel: BGR->RGB, cv2.undistort
// (plumb-bob remap, bilinear, zero border), INTER_AREA downscale by an integer factor.  Undistortion happens at full resolution
// (as in the reference) by sampling the f x f box of undistorted pixels on the fly.  Shared by k_preprocess (the stand-alone
// kernel) and by the stem's loader (seg_stem_mfma.hip), which is why the two give the same bytes.
#pragma once
#include "seg_types.h"

namespace avl {

// camera part: lives in DEVICE memory when the stem reads it (one captured graph serves both cameras)
struct PreCamera {
    float fx, fy, cx, cy, k1, k2, p1, p2, k3;
    int undistort;
};

struct PreParams {
    PreCamera cam;
    int factor;
};

__device__ __forceinline__ void undistorted_rgb(const unsigned char* __restrict__ bgr, int H, int W, const PreCamera& q, int u, int v,
                                                int (&rgb)[3]) {
    if (!q.undistort) {
        const unsigned char* p = bgr + 3ll * ((long long)v * W + u);
        rgb[0] = p[2]; rgb[1] = p[1]; rgb[2] = p[0];
        return;
    }
    const double x = ((double)u - q.cx) / q.fx, y = ((double)v - q.cy) / q.fy;
    const double r2 = x * x + y * y;
    const double radial = 1.0 + q.k1 * r2 + q.k2 * r2 * r2 + q.k3 * r2 * r2 * r2;
    const double xd = x * radial + 2.0 * q.p1 * x * y + q.p2 * (r2 + 2.0 * x * x);
    const double yd = y * radial + q.p1 * (r2 + 2.0 * y * y) + 2.0 * q.p2 * x * y;
    const float sx = (float)(q.fx * xd + q.cx), sy = (float)(q.fy * yd + q.cy);
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const float ax = sx - (float)x0, ay = sy - (float)y0;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = x0 + dx, yy = y0 + dy;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const float wgt = (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay);
            const unsigned char* p = bgr + 3ll * ((long long)yy * W + xx);
            acc[0] += wgt * (float)p[2]; acc[1] += wgt * (float)p[1]; acc[2] += wgt * (float)p[0];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = min(max(__float2int_rn(acc[c]), 0), 255);
}

// pixel (ox, oy) of the network's RGB input: the INTER_AREA mean of its f x f box of undistorted source pixels
__device__ __forceinline__ void preprocessed_rgb(const unsigned char* __restrict__ bgr, int H, int W, const PreCamera& q, int f, int ox, int oy,
                                                 int (&out)[3]) {
    int sum[3] = {0, 0, 0};
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            int rgb[3];
            undistorted_rgb(bgr, H, W, q, ox * f + dx, oy * f + dy, rgb);
            sum[0] += rgb[0]; sum[1] += rgb[1]; sum[2] += rgb[2];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (f == 1) out[c] = sum[c];
        else if (f == 2) out[c] = (sum[c] + 2) >> 2;                                   // OpenCV's integer 2x2 path
        else out[c] = min(max(__float2int_rn((float)sum[c] * (1.0f / (float)(f * f))), 0), 255);
    }
}

// host side: K (row-major 3x3) and dist (k1,k2,p1,p2,k3), both NULL = no undistortion
inline PreCamera make_pre_camera(const double* K, const double* dist) {
    PreCamera q;
    memset(&q, 0, sizeof(q));
    q.undistort = (K && dist) ? 1 : 0;
    if (q.undistort) {
        q.fx = (float)K[0]; q.cx = (float)K[2]; q.fy = (float)K[4]; q.cy = (float)K[5];
        q.k1 = (float)dist[0]; q.k2 = (float)dist[1]; q.p1 = (float)dist[2]; q.p2 = (float)dist[3]; q.k3 = (float)dist[4];
    }
    return q;
}

}  // namespace avl

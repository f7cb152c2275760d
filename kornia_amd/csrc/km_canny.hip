// kornia_amd - the back half of the Canny detector for gfx950: gradient magnitude, direction binning, non-maximum
// suppression and the two thresholds in one pass, and the hysteresis as block-local fixed points in LDS.
//
// Reference: kornia/filters/canny.py:119-159.  After the Gaussian blur and the Sobel derivatives (km_filter2d_sep_fwd,
// km_spatial_gradient_fwd) the reference runs ~25 elementwise / convolution launches over (B,8,H,W) stacks and a host-synchronised
// `while` loop of 6 more per hysteresis iteration; what they compute per pixel is
//   m    = sqrt(gx*gx + gy*gy + eps)
//   d    = round(atan2(gy, gx) * (4 / pi))                          (-4 .. 4, multiples of 45 degrees, ties to even)
//   keep = min(m - m[neighbour d mod 8], m - m[neighbour (d + 4) mod 8]) > 0      (zero-padded magnitude outside the image)
//   mag  = m * keep ;  edges = 0.5 (mag > low) + 0.5 (mag > high)
// with the neighbour order east, south-east, south, ... (kornia/filters/kernels.py:943-976), and then: a weak pixel (0.5) that
// touches a strong one (1) in its 8-neighbourhood becomes strong, repeated until nothing changes; the weak pixels left are dropped.
// The promotion is monotone, so its fixed point does not depend on the order in which pixels are visited: every workgroup iterates
// its 64 x 64 tile (+ a 1-pixel frame) to a LOCAL fixed point in LDS and the host relaunches until no workgroup changed anything
// (a handful of launches instead of one host round trip per pixel of edge length).
#include <stdlib.h>

#include "km_common.h"

struct KmCannyArgs {
    const float* grads;  // (B,2,H,W): gx plane, gy plane
    float* mag;          // (B,H,W)
    float* edges;        // (B,H,W): 0 / 0.5 / 1
    int B, H, W;
    float low, high, eps;
    uint32_t tiles_x, tiles_y, nblocks;
};

// magnitude of pixel (i, j) of image `g` (gx plane; gy plane at + plane), 0 outside the image: F.pad of the magnitude with zeros
__device__ __forceinline__ float kmc_mag_at(const float* g, size_t plane, int i, int j, int H, int W, float eps) {
    if (i < 0 || i >= H || j < 0 || j >= W) return 0.0f;
    const float gx = g[(size_t)i * W + j], gy = g[plane + (size_t)i * W + j];
    return sqrtf((gx * gx + gy * gy) + eps);
}

__global__ __launch_bounds__(256) void km_canny_nms_kernel(const KmCannyArgs a) {
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int j = (int)tx * 64 + (threadIdx.x & 63);
    const int i0 = (int)ty * 16 + (threadIdx.x >> 6) * 4;
    if (j >= a.W) return;
    const size_t plane = (size_t)a.H * a.W;
    const float* g = a.grads + (size_t)b * 2 * plane;
    const float k4pi = 1.2732395447351628f;  // (float)(4 / pi): the Python scalar, rounded to the tensor's dtype
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        if (i >= a.H) break;
        const float gx = g[(size_t)i * a.W + j], gy = g[plane + (size_t)i * a.W + j];
        const float m = sqrtf((gx * gx + gy * gy) + a.eps);
        const float d = rintf(atan2f(gy, gx) * k4pi);
        int k = (d == d) ? (int)d : 0;  // -4 .. 4 (a NaN direction has no neighbour to prefer)
        k &= 7;                         // d mod 8 with the sign of the divisor: -1 -> 7, -4 -> 4
        // neighbour k: (dy, dx) = east, south-east, south, south-west, west, north-west, north, north-east
        const int dy = (k >= 1 && k <= 3) ? 1 : ((k >= 5) ? -1 : 0);
        const int dx = (k == 0 || k == 1 || k == 7) ? 1 : ((k >= 3 && k <= 5) ? -1 : 0);
        const float ahead = m - kmc_mag_at(g, plane, i + dy, j + dx, a.H, a.W, a.eps);
        const float behind = m - kmc_mag_at(g, plane, i - dy, j - dx, a.H, a.W, a.eps);
        const float keep = (fminf(ahead, behind) > 0.0f) ? 1.0f : 0.0f;
        const float mo = m * keep;
        const size_t o = (size_t)b * plane + (size_t)i * a.W + j;
        km_st(a.mag + o, mo);
        km_st(a.edges + o, (mo > a.low ? 0.5f : 0.0f) + (mo > a.high ? 0.5f : 0.0f));
    }
}

// ---- hysteresis ---------------------------------------------------------------------------------------------------------
#define KMH_T 64                 // tile edge
#define KMH_P (KMH_T + 2)        // with the 1-pixel frame
struct KmHystArgs {
    float* state;   // (B,H,W) in place: 0 / 0.5 / 1 (weak pixels connected to strong ones become 1)
    float* out;     // (B,H,W): 1 where state is 1, else 0 - rewritten by every launch, final after the launch that changes nothing
    int* changed;   // device flag, set to 1 by a workgroup that promoted a pixel
    int B, H, W;
    uint32_t tiles_x, tiles_y, nblocks;
};

__global__ __launch_bounds__(256) void km_canny_hysteresis_kernel(const KmHystArgs a) {
    __shared__ uint8_t s[KMH_P * KMH_P];  // 0 none / 1 weak / 2 strong
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int X0 = (int)tx * KMH_T - 1, Y0 = (int)ty * KMH_T - 1;  // image position of s[0]
    const size_t plane = (size_t)a.H * a.W;
    float* st = a.state + (size_t)b * plane;
    float* out = a.out + (size_t)b * plane;
    for (int e = threadIdx.x; e < KMH_P * KMH_P; e += 256) {
        const int li = e / KMH_P, lj = e - li * KMH_P;
        const int i = Y0 + li, j = X0 + lj;
        uint8_t c = 0;
        if (i >= 0 && i < a.H && j >= 0 && j < a.W) {
            const float v = st[(size_t)i * a.W + j];
            c = v == 1.0f ? 2 : (v == 0.5f ? 1 : 0);
        }
        s[e] = c;
    }
    __syncthreads();
    // local fixed point over the tile's interior (the frame is read-only: its pixels belong to the neighbouring workgroups)
    bool any_change = false;
    for (;;) {
        bool ch = false;
        for (int e = threadIdx.x; e < KMH_T * KMH_T; e += 256) {
            const int li = 1 + e / KMH_T, lj = 1 + (e & (KMH_T - 1));
            const int c = li * KMH_P + lj;
            if (s[c] == 1) {
                const bool touch = s[c - KMH_P - 1] == 2 || s[c - KMH_P] == 2 || s[c - KMH_P + 1] == 2 || s[c - 1] == 2 || s[c + 1] == 2 ||
                                   s[c + KMH_P - 1] == 2 || s[c + KMH_P] == 2 || s[c + KMH_P + 1] == 2;
                if (touch) { s[c] = 2; ch = true; }  // (a neighbour may see this promotion in the same sweep: the fixed point is the same)
            }
        }
        any_change = any_change || ch;
        if (!__syncthreads_or((int)ch)) break;
    }
    for (int e = threadIdx.x; e < KMH_T * KMH_T; e += 256) {
        const int li = 1 + e / KMH_T, lj = 1 + (e & (KMH_T - 1));
        const int i = Y0 + li, j = X0 + lj;
        if (i < a.H && j < a.W) {
            const uint8_t c = s[li * KMH_P + lj];
            const size_t o = (size_t)i * a.W + j;
            if (c == 2) st[o] = 1.0f;  // (only promotions are written back: neighbours read this workgroup's frame pixels concurrently)
            out[o] = c == 2 ? 1.0f : 0.0f;
        }
    }
    if (__syncthreads_or((int)any_change) && threadIdx.x == 0) atomicOr(a.changed, 1);
}

extern "C" {

// grads (B,2,H,W) fp32 = spatial_gradient's (B,1,2,H,W) output; mag, edges (B,H,W) fp32, written.
int km_canny_nms_fwd(const void* grads, void* mag, void* edges, int B, int H, int W, double low, double high, double eps, void* stream) {
    if (B == 0 || H == 0 || W == 0) return 0;
    KM_REQUIRE(grads && mag && edges, "km_canny_nms_fwd: null pointer");
    KM_REQUIRE(B > 0 && H > 0 && W > 0, "km_canny_nms_fwd: bad shape");
    KmCannyArgs a;
    a.grads = (const float*)grads; a.mag = (float*)mag; a.edges = (float*)edges;
    a.B = B; a.H = H; a.W = W;
    a.low = (float)low; a.high = (float)high; a.eps = (float)eps;
    a.tiles_x = (uint32_t)((W + 63) / 64);
    a.tiles_y = (uint32_t)((H + 15) / 16);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_canny_nms_fwd: grid too large");
    a.nblocks = (uint32_t)nb;
    hipLaunchKernelGGL(km_canny_nms_kernel, dim3(a.nblocks), dim3(256), 0, (hipStream_t)stream, a);
    return km_check_launch("km_canny_nms_fwd");
}

// One sweep of the hysteresis: every 64 x 64 tile is iterated to its local fixed point.  state (B,H,W) fp32 in place, out (B,H,W)
// fp32 written, changed: one int on the device, OR-ed with 1 when a pixel was promoted (the caller zeroes it, launches, reads it
// back and repeats while it is set).
int km_canny_hysteresis_sweep(void* state, void* out, int* changed, int B, int H, int W, void* stream) {
    if (B == 0 || H == 0 || W == 0) return 0;
    KM_REQUIRE(state && out && changed, "km_canny_hysteresis_sweep: null pointer");
    KM_REQUIRE(B > 0 && H > 0 && W > 0, "km_canny_hysteresis_sweep: bad shape");
    KmHystArgs a;
    a.state = (float*)state; a.out = (float*)out; a.changed = changed;
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = (uint32_t)((W + KMH_T - 1) / KMH_T);
    a.tiles_y = (uint32_t)((H + KMH_T - 1) / KMH_T);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_canny_hysteresis_sweep: grid too large");
    a.nblocks = (uint32_t)nb;
    hipLaunchKernelGGL(km_canny_hysteresis_kernel, dim3(a.nblocks), dim3(256), 0, (hipStream_t)stream, a);
    return km_check_launch("km_canny_hysteresis_sweep");
}

}  // extern "C"

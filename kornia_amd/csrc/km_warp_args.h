// kornia_amd - argument block of the 2-D warp kernels, shared by km_warp.hip and km_warp_cubic.hip.
#pragma once

#include "km_sampler.h"

template <typename T>
struct KmWarpArgs {
    typedef typename KmTraits<T>::R R;
    const T* src;     // (B,C,H,W)
    const R* mat;     // (B_M,9) row-major, compute dtype
    T* dst;           // fwd: (B,C,h,w)
    const T* gout;    // bwd: (B,C,h,w)
    R* gsrc;          // bwd: (B,C,H,W) accumulators in compute dtype, pre-zeroed (nullable)
    double* gmat;     // bwd: (B_M,9) fp64 accumulators, pre-zeroed (nullable)
    const R* fill;    // (C) compute dtype, pad == fill only
    const T* grid;    // KM_COORD_GRID: (B_M,h,w,2) normalised sampling grid in the image dtype
    R* ggrid;         // KM_COORD_GRID bwd: (B,h,w,2) gradient wrt the grid, written (nullable)
    const uint8_t* apply;  // fwd, nullable: (B) per-sample switch of the augmentation layer - a sample whose entry is 0 is copied (h == H, w == W)
    KmWarpGeom<R> g;
    uint32_t tiles_x, tiles_y, nblocks;
    uint32_t reverse;  // lean forward only: the XCDs walk their block ranges backwards (km_traversal_next)
    uint32_t stream_out;  // lean forward only: streaming stores (km_stream_stores)
};

// The per-sample probability blend of the augmentation layer (kornia/augmentation/base.py:348-393) folded into the forward: a
// sample that is NOT transformed is copied by the workgroups that would have warped it (same mapping: this thread's column j, rows
// i_base + r * row_step), so `torch.where(to_apply, transformed, input)` - a third full pass - disappears.
template <typename T>
__device__ __forceinline__ void km_fwd_copy_rows(const KmWarpArgs<T>& a, uint32_t b, int j, int i_base, int row_step, int n_rows) {
    const auto& g = a.g;
    if (j >= g.w) return;
    const size_t plane = (size_t)g.h * g.w;  // == H * W (checked on the host)
    for (int c = 0; c < g.C; ++c) {
        const T* __restrict__ sp = a.src + ((size_t)b * g.C + c) * plane;
        T* __restrict__ dp = a.dst + ((size_t)b * g.C + c) * plane;
        for (int r = 0; r < n_rows; ++r) {
            const int i = i_base + r * row_step;
            if (i < g.h) dp[(size_t)i * g.w + j] = sp[(size_t)i * g.w + j];
        }
    }
}


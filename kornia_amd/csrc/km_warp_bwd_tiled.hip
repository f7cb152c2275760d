// kornia_amd - "owner-computes" gradient of the bilinear warps (zeros / fill padding) with respect to the
// IMAGE, for gfx950.  (The gradient with respect to the matrix is km_warp_gm.hip.)
//
// Two ways of producing both gradients from ONE read of grad_out were built and measured on MI355X (256x3x512^2) and are not in the
// tree: the matrix-gradient terms inside the tile-owner loop (source taps gathered with each pixel's grad_out: 1.32 ms under the
// scatter's 80-register bound, 0.92 ms with 128 registers and 2 workgroups per CU), and the two kernels' workgroups interleaved in one
// launch (tile-owner and matrix-gradient workgroups of the same image as neighbours on a CU, grad_out shared through L2: 0.845 ms) -
// against 0.77 - 0.79 ms for the two launches.  Both kernels are limited by the latency their own resident waves can hide, so taking
// registers, LDS or workgroup slots from one to host the other costs more than the second read of grad_out.
//
// The generic backward (km_warp.hip) scatters 4*C fp32 atomics per output pixel into HBM - what ATen's
// grid_sampler_2d_backward does - and needs grad_src zeroed first: ~5e bytes of HBM traffic per element
// against 2e algorithmic, serialised by the L2 atomic units (4.2 ms at 256x3x512^2).  Here:
//
//   * a workgroup OWNS one KMT_TW x KMT_TH tile of grad_src (all channels) and accumulates it in LDS;
//   * the output pixels q that can touch the tile are found by pushing the tile rectangle (grown by the
//     1-pixel bilinear footprint) through the inverse map G (source pixel -> output index): a projective map
//     sends the rectangle to a convex quad, so the bounding box of the four mapped corners plus an explicit
//     rounding bound contains them all (kmt_tile_box);
//   * every q of the box recomputes its sampling position with EXACTLY the forward's instruction sequence
//     and adds its four contributions w * grad_out[q, c] to the taps that fall inside the tile.  Taps in
//     other tiles are added by those tiles' owners, so every contribution is added exactly once;
//   * gfx950 LDS float atomics are ~40x slower than integer ones (profiles/r01_lds_atomics_microbench.txt),
//     so the accumulators are int32 fixed point (see the kernel comment): deterministic, order-independent;
//   * the tile is converted and written once with coalesced stores: no memset, no global atomics.
//
// If the tile straddles the vanishing line of G (corner denominators of mixed sign or ~0) the pre-image is
// not a bounded convex quad: that tile scans the whole output image (correct, slower; only the tiles crossed
// by the line pay).
//
// HBM traffic: read grad_out ~1.1x (box overlap, mostly served by L2) + write grad_src once = 2e bytes/element.
// The kernel is VALU-issue bound (~150 instructions per visited pixel for the exact coordinate pipeline and
// the 12 quantise+atomic pairs), not HBM bound: see DESIGN.md for the counters.
#include <stdlib.h>

#include "km_warp_tile.h"

#ifndef KMT_NT
#define KMT_NT 512          // threads per workgroup
#endif
#define KMT_NW (KMT_NT / 64)
#ifndef KMT_UNROLL
#define KMT_UNROLL 2       // pixels in flight per thread in the scatter loop
#endif
#define KMT_CC 3            // channels per pass
#define KMT_LDS_BYTES ((KMT_BAND_W + KMT_TAB) * 16 + KMT_CC * KMT_PLANE * 4)

template <typename T>
struct KmWarpTiledArgs {
    const T* gout;       // (B,C,h,w)
    const float* mat;    // (B_M,9)
    float* gsrc;         // (B,C,H,W) fp32, written completely (no pre-zeroing needed)
    KmWarpGeom<float> g;
    uint32_t tiles_x, tiles_y, nblocks;
    uint32_t reverse;    // the XCDs walk their block ranges backwards (km_traversal_next)
    uint32_t stream_out; // streaming stores of the tile flush (km_stream_stores)
};

// The four contributions w * grad_out[q, c] of a visited pixel to the taps that fall inside the tile.  FIXED: int32 fixed-point LDS
// accumulators; otherwise IEEE float LDS atomics (slow; non-finite gradients, vanishing-line tiles, extreme magnification).
template <int CC, bool FIXED>
__device__ __forceinline__ void kmt_pix_scatter(const KmtPix& q, const float (&go)[CC], int* s_acc, float scale, uint32_t TWc, uint32_t THc, uint32_t& seen_bits) {
    // the fixed-point scale was chosen for |grad_out| <= bound: remember the largest magnitude seen, as an integer
    // (sign cleared, IEEE bit patterns order like unsigned integers and NaN / inf sort above every finite value)
    if (FIXED) {
        uint32_t mb = __float_as_uint(go[0]) & 0x7fffffffu;
#pragma unroll
        for (int c = 1; c < CC; ++c) mb = max(mb, __float_as_uint(go[c]) & 0x7fffffffu);
        seen_bits = max(seen_bits, mb);
    }
    const KmlTaps& t = q.t;
    const uint32_t ux = q.ux, uy = q.uy;
    const bool in_x0 = ux < TWc, in_x1 = (ux + 1u) < TWc;
    const bool in_y0 = uy < THc, in_y1 = (uy + 1u) < THc;
    bool t00 = in_x0 && in_y0, t01 = in_x1 && in_y0, t10 = in_x0 && in_y1, t11 = in_x1 && in_y1;  // (lane masks: s_and_b64)
    const int l00 = (int)(uy * (uint32_t)KMT_TW + ux);
    if (FIXED) {
        // scale = 2^k: (wx * wy) * scale == wx * (wy * scale) bit for bit.  Tap-outer order: one exec-mask region per tap.
        const float wy0s = t.wy0 * scale, wy1s = t.wy1 * scale;
        const float w00 = t.wx1 * wy1s, w01 = t.wx0 * wy1s, w10 = t.wx1 * wy0s, w11 = t.wx0 * wy0s;
        int* accp = s_acc + l00;
        if (t00) {
#pragma unroll
            for (int c = 0; c < CC; ++c) atomicAdd(accp + c * KMT_PLANE, kmt_quant(w00 * go[c]));
        }
        if (t01) {
#pragma unroll
            for (int c = 0; c < CC; ++c) atomicAdd(accp + c * KMT_PLANE + 1, kmt_quant(w01 * go[c]));
        }
        if (t10) {
#pragma unroll
            for (int c = 0; c < CC; ++c) atomicAdd(accp + c * KMT_PLANE + KMT_TW, kmt_quant(w10 * go[c]));
        }
        if (t11) {
#pragma unroll
            for (int c = 0; c < CC; ++c) atomicAdd(accp + c * KMT_PLANE + KMT_TW + 1, kmt_quant(w11 * go[c]));
        }
    } else {
        const bool num = (q.x == q.x) & (q.y == q.y);  // a NaN position touches nothing (ATen: the converted index is out of bounds)
        t00 = t00 && num; t01 = t01 && num; t10 = t10 && num; t11 = t11 && num;
        const float w00 = t.wx1 * t.wy1, w01 = t.wx0 * t.wy1, w10 = t.wx1 * t.wy0, w11 = t.wx0 * t.wy0;
        float* accp = (float*)s_acc + l00;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            if (t00) atomicAdd(accp + c * KMT_PLANE, w00 * go[c]);
            if (t01) atomicAdd(accp + c * KMT_PLANE + 1, w01 * go[c]);
            if (t10) atomicAdd(accp + c * KMT_PLANE + KMT_TW, w10 * go[c]);
            if (t11) atomicAdd(accp + c * KMT_PLANE + KMT_TW + 1, w11 * go[c]);
        }
    }
}

struct KmtBand {
    int jb, bwb;   // first output column of the band, its width (<= KMT_BAND_W)
    int ib, nrows; // first output row of the band, its height (<= KMT_TAB)
};

// One band of the box, walked as a linear list of pixels: element e = base + tid, (row, column) = (e / bwb, e % bwb).  Lane
// utilisation is bwb * nrows / (a multiple of KMT_NT) whatever the shape of the box, consecutive lanes read consecutive
// grad_out pixels.  Two pixels in flight per thread (FIXED: the IEEE float path is not worth unrolling).
template <typename T, int CM, int ALIGN, int CC, bool FAST, bool FIXED>
__device__ __forceinline__ void kmt_scatter_band(const KmWarpGeom<float>& g, const float (&m)[9], const KmtBand& bd, const T* const (&gout_c)[CC],
                                                 const float4* s_u4, const float4* s_v4, int* s_acc, float scale, uint32_t X0, uint32_t TWc,
                                                 uint32_t Y0, uint32_t THc, uint32_t& seen_bits) {
    const int tid = threadIdx.x;
    const float Wm1 = (float)(g.W - 1), Hm1 = (float)(g.H - 1), hW = (float)g.W / 2, hH = (float)g.H / 2;
    const int bwb = bd.bwb;
    const int di = kmt_uniform(KMT_NT / bwb), dj = kmt_uniform(KMT_NT % bwb);  // element e + KMT_NT is di rows, dj columns on
    const int nq = bwb * bd.nrows;
    // element e = base + tid  (tid < 2^24: the float quotient is off by at most one)
    int qi = (int)(((float)tid + 0.5f) / (float)bwb), qj = tid - qi * bwb;
    if (qj < 0) { qi -= 1; qj += bwb; }
    if (qj >= bwb) { qi += 1; qj -= bwb; }
    const uint32_t row0 = (uint32_t)bd.ib * (uint32_t)g.w + (uint32_t)bd.jb;  // the host guarantees 4 * h * w < 2^32
    int base = 0;
    if (FIXED) {
        // KMT_UNROLL pixels per thread and iteration, all their grad_out loads issued first: a tile's time is (iterations) x (memory
        // latency + the pixels' arithmetic), so fewer, fatter iterations shorten it.  (A software-pipelined form - the loads of
        // iteration k + 1 issued before the pixels of iteration k - measured slower: 0.48 vs 0.45 ms on the same box.)
        for (; base + KMT_UNROLL * KMT_NT <= nq; base += KMT_UNROLL * KMT_NT) {
            int pqi[KMT_UNROLL], pqj[KMT_UNROLL];
            float go[KMT_UNROLL][CC];
#pragma unroll
            for (int s4 = 0; s4 < KMT_UNROLL; ++s4) {
                pqi[s4] = qi; pqj[s4] = qj;
                kmt_load_go<T, CC>(gout_c, row0 + (uint32_t)qi * (uint32_t)g.w + (uint32_t)qj, go[s4]);
                kmt_advance(qi, qj, di, dj, bwb);
            }
#pragma unroll
            for (int s4 = 0; s4 < KMT_UNROLL; ++s4) {
                const float4 c0 = s_u4[pqj[s4]], r0 = s_v4[pqi[s4]];
                KmtPix q;
                kmt_pix_position<CM, ALIGN, FAST>(m, kmt_half(c0), kmt_half(r0), true, Wm1, hW, Hm1, hH, X0, Y0, q);
                kmt_pix_scatter<CC, FIXED>(q, go[s4], s_acc, scale, TWc, THc, seen_bits);
            }
        }
    }
    for (; base < nq; base += KMT_NT) {
        const bool valid = base + tid < nq;
        const int vqi = valid ? qi : 0, vqj = valid ? qj : 0;
        float go[CC];
        kmt_load_go<T, CC>(gout_c, row0 + (uint32_t)vqi * (uint32_t)g.w + (uint32_t)vqj, go);
        const float4 c0 = s_u4[vqj], r0 = s_v4[vqi];
        KmtPix q;
        kmt_pix_position<CM, ALIGN, FAST>(m, kmt_half(c0), kmt_half(r0), valid, Wm1, hW, Hm1, hH, X0, Y0, q);
        kmt_pix_scatter<CC, FIXED>(q, go, s_acc, scale, TWc, THc, seen_bits);
        kmt_advance(qi, qj, di, dj, bwb);
    }
}

// One channel chunk (CC channels) of one tile: zero, choose the fixed-point scale, scatter, convert and write.
//
// gfx950 measurements that shape this (scratch micro-benchmark, 2048 blocks x 256 threads):
//   ds_add_f32 / ds_add_rtn_f32    193 cycles per wave-instruction per CU   (float LDS atomics are ~40x slower
//   ds_add_u32 / ds_add_rtn_u32      5 cycles per wave-instruction per CU    than integer ones)
// so the per-tap contributions w * grad_out are accumulated as int32 fixed point: scale = 2^k with
// k = 30 - hb - ceil(log2 M), M >= max |grad_out| over the visited pixels (hb = head-room bits for the number of taps that
// can land on one source pixel, from the Jacobian bound), adding floor(w * g * scale + 0.5) with ds_add_u32.  Per-term error
// <= 2^-(k+1), i.e. <= M * 2^-(27-hb): the same order as one fp32 ulp of M.  Integer addition is associative, so the result
// is independent of the order in which waves run: bit-reproducible, unlike float atomics.
//
// The kernel is bound by the latency of its serial phases, not by instruction issue (profiles/r02_*: 62 % of the wave
// time is spent waiting; halving the VALU count per pixel changed nothing), so the phases are arranged to overlap:
//   * M is SPECULATED: 8 x (max over a KMT_NT-pixel sample) - reading the whole box twice would cost ~0.3 ms at
//     256x3x512^2.  The sample is taken around the tile's own location in the output (the pre-image of a tile is there for
//     the near-identity warps of the hot cases, and a gradient field has the same order of magnitude everywhere), so its
//     loads are issued BEFORE the box is known and fly while wave 0 computes the box and the others zero the accumulators.
//     The scatter pass tracks the largest magnitude it loads; only a tile that sees a larger one (or NaN / inf) is redone with
//     the exact maximum over its box (attempt 1).  The factor 8 costs 3 bits of the fixed-point resolution;
//   * three block barriers per tile (accumulators zeroed + box + sample | coordinate tables | scatter done) when the box
//     fits one band of the tables - any warp that does not shrink the image by more than 2x.
template <typename T, int CM, int ALIGN, int CC>
__device__ __forceinline__ void kmt_tile_chunk(const KmWarpTiledArgs<T>& a, const float (&m)[9], uint32_t b, int cbase, int X0, int Y0, int TWc, int THc,
                                               bool box_pending, const int* s_box, float4* s_u4, float4* s_v4, int* s_acc, float* red_max) {
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t src_plane = (size_t)g.H * g.W, dst_plane = (size_t)g.h * g.w;
    float* gsrc_b = a.gsrc + (size_t)b * g.C * src_plane;
    const T* gout_c[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) gout_c[c] = a.gout + ((size_t)b * g.C + (size_t)(cbase + c)) * dst_plane;
    // ---- speculative bound: one pixel per thread on a KMT_NT-point lattice over the tile's own location in the output ----
    float vsample = 0.f;
    if (g.h > 0 && g.w > 0) {
        constexpr int LX = 32, LY = KMT_NT / LX;  // lattice
        const float fx = (float)g.w / (float)g.W, fy = (float)g.h / (float)g.H;
        const int sx = X0 + ((tid % LX) * KMT_TW + KMT_TW / 2) / LX, sy = Y0 + ((tid / LX) * KMT_TH + KMT_TH / 2) / LY;
        const int jq = min(g.w - 1, max(0, (int)((float)sx * fx))), iq = min(g.h - 1, max(0, (int)((float)sy * fy)));
        const uint32_t off = (uint32_t)iq * (uint32_t)g.w + (uint32_t)jq;
#pragma unroll
        for (int c = 0; c < CC; ++c) vsample = fmaxf(vsample, km_fabs((float)km_ld(km_at(gout_c[c], off))));
    }
    // ---- zero the accumulators (while the first chunk's box is still being computed by wave 0, the other waves do it) ----
    if (box_pending) {
        if (wave != 0)
            for (int e = tid - 64; e < CC * KMT_PLANE / 4; e += KMT_NT - 64) ((int4*)s_acc)[e] = make_int4(0, 0, 0, 0);
    } else {
        for (int e = tid; e < CC * KMT_PLANE / 4; e += KMT_NT) ((int4*)s_acc)[e] = make_int4(0, 0, 0, 0);
    }
    {
        vsample = vsample * 8.0f;
        const bool bad = !(vsample <= 3.0e38f);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vsample = fmaxf(vsample, __shfl_down(vsample, off, 64));
        const unsigned long long badmask = __ballot(bad);
        if (lane == 0) red_max[wave] = badmask ? __int_as_float(0x7f800000) : vsample;
    }
    __syncthreads();  // accumulators zeroed, box published, sample maxima published
    const int j0 = kmt_uniform(s_box[0]), j1 = kmt_uniform(s_box[1]), i0 = kmt_uniform(s_box[2]), i1 = kmt_uniform(s_box[3]);
    const int hb = kmt_uniform(s_box[4]);
    const bool fixed_ok = kmt_uniform(s_box[5]) != 0;  // fixed-point accumulation is accurate enough (bounded multiplicity)
    const int bw = j1 - j0 + 1, bh = i1 - i0 + 1;
    const bool empty = (bw <= 0 || bh <= 0);
    const float inv_bw = kmt_uniform(bw > 0 ? 1.0f / (float)bw : 0.f);

    float scale = 1.f, inv_scale = 1.f;
    bool finite = false;
    for (int attempt = (fixed_ok ? 0 : 1); attempt < 2; ++attempt) {
        float M;
        if (attempt == 0) {
            M = red_max[0];
#pragma unroll
            for (int k = 1; k < KMT_NW; ++k) M = fmaxf(M, red_max[k]);
        } else {
            // the exact maximum over the box
            float vmax = 0.f;
            bool bad = false;
            if (!empty) {
                const int nq = bw * bh;
                for (int base = 0; base < nq; base += 4 * KMT_NT) {
                    float vv[4][CC];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int e = min(base + s4 * KMT_NT + tid, nq - 1);  // clamped: duplicates do not change a max
                        int qi = (int)(((float)e + 0.5f) * inv_bw);
                        int qj = e - qi * bw;
                        if (qj < 0) { qi -= 1; qj += bw; }
                        if (qj >= bw) { qi += 1; qj -= bw; }
                        const uint32_t off = (uint32_t)(i0 + qi) * (uint32_t)g.w + (uint32_t)(j0 + qj);
#pragma unroll
                        for (int c = 0; c < CC; ++c) vv[s4][c] = km_fabs((float)km_ld(km_at(gout_c[c], off)));
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            bad = bad || !(vv[s4][c] <= 3.0e38f);
                            vmax = fmaxf(vmax, vv[s4][c]);
                        }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
            const unsigned long long badmask = __ballot(bad);
            __syncthreads();  // red_max is still being read by attempt 0's readers
            if (lane == 0) red_max[wave] = badmask ? __int_as_float(0x7f800000) : vmax;
            __syncthreads();
            M = red_max[0];
#pragma unroll
            for (int k = 1; k < KMT_NW; ++k) M = fmaxf(M, red_max[k]);
        }
        // scale = 2^k with |w * g * scale| * (taps per pixel) < 2^30
        finite = (M <= 3.0e38f) && fixed_ok;  // else: IEEE float accumulation (slow ds_add_f32)
        int kexp = 0;
        if (finite && M > 0.f) {
            int ex2;
            (void)frexpf(M, &ex2);  // M = f * 2^ex2, f in [0.5, 1)  =>  M < 2^ex2
            kexp = 30 - hb - ex2;
            kexp = max(-126, min(126, kexp));
        }
        scale = kmt_uniform(ldexpf(1.0f, kexp));
        inv_scale = kmt_uniform(ldexpf(1.0f, -kexp));
        const float bound = finite ? M : __int_as_float(0x7f800000);  // the float path accepts anything

        // ---- scatter pass ----
        uint32_t seen_bits = 0;  // largest |grad_out| bit pattern this thread has loaded
        if (!empty) {
            // the box is walked in bands of at most KMT_BAND_W columns x KMT_TAB rows (one band for any warp that does not
            // shrink the image by more than 2x), whose coordinate halves sit in the LDS tables
            bool first_band = true;
            for (int jb = j0; jb <= j1; jb += KMT_BAND_W) {
                KmtBand bd;
                bd.jb = jb;
                bd.bwb = min(KMT_BAND_W, j1 - jb + 1);
                for (int ib = i0; ib <= i1; ib += KMT_TAB) {
                    bd.ib = ib;
                    bd.nrows = min(i1, ib + KMT_TAB - 1) - ib + 1;
                    if (!first_band) __syncthreads();  // the previous band's readers are done with the tables
                    first_band = false;
                    if (tid < bd.bwb && (ib == i0)) {
                        const float u = km_base_x<float, CM>(g, jb + tid);
                        const KmlHalf h = kml_col_half<CM>(m, u);
                        s_u4[tid] = make_float4(h.a, h.b, h.c, u);
                    }
                    bool okr = true;
                    const int rt = tid - (KMT_NT - KMT_TAB);  // the row table is filled by the LAST threads: the first ones fill the columns
                    if (rt >= 0 && rt < bd.nrows) {
                        const float v = km_base_y<float, CM>(g, ib + rt);
                        const KmlHalf h = kml_row_half<CM>(m, v);
                        s_v4[rt] = make_float4(h.a, h.b, h.c, v);
                        okr = kml_row_guard<CM>(g, m, v);
                    }
                    const bool fast = __syncthreads_and((int)okr) != 0;  // every row of the band has safe division operands
                    const uint32_t uX0 = (uint32_t)X0, uY0 = (uint32_t)Y0, uTW = (uint32_t)TWc, uTH = (uint32_t)THc;
                    if (!finite) kmt_scatter_band<T, CM, ALIGN, CC, false, false>(g, m, bd, gout_c, s_u4, s_v4, s_acc, scale, uX0, uTW, uY0, uTH, seen_bits);
                    else if (fast) kmt_scatter_band<T, CM, ALIGN, CC, true, true>(g, m, bd, gout_c, s_u4, s_v4, s_acc, scale, uX0, uTW, uY0, uTH, seen_bits);
                    else kmt_scatter_band<T, CM, ALIGN, CC, false, true>(g, m, bd, gout_c, s_u4, s_v4, s_acc, scale, uX0, uTW, uY0, uTH, seen_bits);
                }
            }
        }
        // the fixed-point scale was chosen for |grad_out| <= bound; anything larger (or NaN) voids the attempt
        const bool exceeded = seen_bits > __float_as_uint(bound);
        const int redo = __syncthreads_or((int)exceeded);
        if (attempt == 0 && redo) {
            for (int e = tid; e < CC * KMT_PLANE / 4; e += KMT_NT) ((int4*)s_acc)[e] = make_int4(0, 0, 0, 0);  // discard the attempt
            continue;  // the barriers at the top of attempt 1 order these stores before the next atomics
        }
        break;
    }

    // ---- convert and write the tile ----
    if (TWc == KMT_TW && (g.W & 3) == 0 && ((uintptr_t)a.gsrc & 15) == 0) {
        // full-width tile, 16-byte aligned rows: 16 lanes x 16 bytes cover a tile row, a wave writes 4 rows per store
        const int col4 = (tid & 15) * 4, row0 = tid >> 4;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float* outp = gsrc_b + (size_t)(cbase + c) * src_plane + (size_t)(Y0 + row0) * g.W + (X0 + col4);
            const int* accp = s_acc + c * KMT_PLANE + row0 * KMT_TW + col4;
            const size_t ostep = (size_t)(KMT_NT / 16) * g.W;
#pragma unroll 2
            for (int r = row0; r < THc; r += KMT_NT / 16) {
                KM_CHECK_ALIGNED(accp, 16);
                KM_CHECK_ALIGNED(outp, 16);
                const int4 q = *reinterpret_cast<const int4*>(accp);
                float4 v;
                if (finite) {
                    v = make_float4((float)q.x * inv_scale, (float)q.y * inv_scale, (float)q.z * inv_scale, (float)q.w * inv_scale);
                } else {
                    v = make_float4(__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w));
                }
                if (a.stream_out) {
                    typedef float km_f4v __attribute__((ext_vector_type(4)));
                    km_f4v vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
                    __builtin_nontemporal_store(vv, reinterpret_cast<km_f4v*>(outp));
                } else {
                    *reinterpret_cast<float4*>(outp) = v;
                }
                outp += ostep;
                accp += (KMT_NT / 16) * KMT_TW;
            }
        }
    } else {
        // ragged right edge / unaligned rows: lane -> column, waves -> rows, one float per store
        const int col = tid & (KMT_TW - 1), row0 = tid >> 6;
        if (col < TWc) {
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                float* outp = gsrc_b + (size_t)(cbase + c) * src_plane + (size_t)(Y0 + row0) * g.W + (X0 + col);
                const int* accp = s_acc + c * KMT_PLANE + row0 * KMT_TW + col;
                const size_t ostep = (size_t)KMT_NW * g.W;
#pragma unroll 4
                for (int r = row0; r < THc; r += KMT_NW) {
                    *outp = finite ? (float)(*accp) * inv_scale : *(const float*)accp;
                    outp += ostep;
                    accp += KMT_NW * KMT_TW;
                }
            }
        }
    }
}

// Tile-owner SCATTER with fixed-point LDS accumulators (see the file header and kmt_tile_chunk).
#ifndef KMT_MIN_WAVES
#define KMT_MIN_WAVES 6
#endif

// one tile (tx, ty) of image b, all channels: the body of a workgroup
template <typename T, int CM, int ALIGN>
__device__ __forceinline__ void kmt_tile_block(const KmWarpTiledArgs<T>& a, uint32_t tx, uint32_t ty, uint32_t b, char* smem_raw, float* red_max, int* s_box) {
    const KmWarpGeom<float>& g = a.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int X0 = (int)tx * KMT_TW, Y0 = (int)ty * KMT_TH;
    const int X1 = min(X0 + KMT_TW, g.W), Y1 = min(Y0 + KMT_TH, g.H);  // tile = [X0,X1) x [Y0,Y1)
    const int TWc = X1 - X0, THc = Y1 - Y0;

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }

    // ---- box of output pixels that can touch the tile: computed by wave 0 only (~400 block-uniform VALU instructions),
    //      published through LDS by the first barrier of the first channel chunk
    if (wave == 0) {
        const KmtBox bx = kmt_tile_box<CM>(g, m, X0, X1, Y0, Y1);
        if (lane == 0) {
            s_box[0] = bx.j0; s_box[1] = bx.j1; s_box[2] = bx.i0; s_box[3] = bx.i1;
            s_box[4] = (int)ceilf(log2f(fmaxf(bx.mult, 1.f))) + 1;  // head-room bits
            s_box[5] = bx.fixed_ok ? 1 : 0;
        }
    }

    // LDS carve: coordinate tables, then the int32 accumulators [cc][TH][TW]
    float4* s_u4 = (float4*)smem_raw;   // [KMT_BAND_W] per column of the band
    float4* s_v4 = s_u4 + KMT_BAND_W;   // [KMT_TAB] per row of the band
    int* s_acc = (int*)(s_v4 + KMT_TAB);

    // channels in chunks of 3 (RGB: one pass); a remainder of 1 or 2 channels goes one channel at a time
    int cbase = 0;
    for (; cbase + KMT_CC <= g.C; cbase += KMT_CC) {
        if (cbase) __syncthreads();  // the previous chunk's flush is done with the accumulators
        kmt_tile_chunk<T, CM, ALIGN, KMT_CC>(a, m, b, cbase, X0, Y0, TWc, THc, cbase == 0, s_box, s_u4, s_v4, s_acc, red_max);
    }
    for (; cbase < g.C; ++cbase) {
        if (cbase) __syncthreads();
        kmt_tile_chunk<T, CM, ALIGN, 1>(a, m, b, cbase, X0, Y0, TWc, THc, cbase == 0, s_box, s_u4, s_v4, s_acc, red_max);
    }
}

template <typename T, int CM, int ALIGN>
__global__ __launch_bounds__(KMT_NT, KMT_MIN_WAVES) void km_warp_bwd_tiled_kernel(const KmWarpTiledArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float red_max[KMT_NW];
    __shared__ int s_box[8];
    static_assert(KMT_BAND_W + KMT_TAB <= KMT_NT || KMT_NT >= 2 * KMT_TAB, "the column and row tables are filled by disjoint threads");
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    kmt_tile_block<T, CM, ALIGN>(a, tx, ty, b, smem_raw, red_max, s_box);
}

template <typename T, int CM>
static int kmt_launch(const KmWarpTiledArgs<T>& a, hipStream_t s) {
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_bwd_tiled_kernel<T, CM, 1>), dim3(a.nblocks), dim3(KMT_NT), (size_t)KMT_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL((km_warp_bwd_tiled_kernel<T, CM, 0>), dim3(a.nblocks), dim3(KMT_NT), (size_t)KMT_LDS_BYTES, s, a);
    return km_check_launch("km_warp2d_bwd(tiled)");
}

template <typename T>
static int kmt_run(const void* gout, const void* mat, void* gsrc, int B, int C, int H, int W, int h, int w, int B_M, int coord_mode, int norm_coords, int pad,
                   int align, hipStream_t s) {
    KmWarpTiledArgs<T> a;
    a.gout = (const T*)gout; a.mat = (const float*)mat; a.gsrc = (float*)gsrc;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, pad, align);
    a.tiles_x = (uint32_t)((W + KMT_TW - 1) / KMT_TW);
    a.tiles_y = (uint32_t)((H + KMT_TH - 1) / KMT_TH);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d_bwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    a.reverse = km_traversal_next(s);
    a.stream_out = km_stream_stores((uint64_t)B * C * H * W * sizeof(float));
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmt_launch<T, KM_COORD_PERSPECTIVE>(a, s);
        case KM_COORD_AFFINE: return kmt_launch<T, KM_COORD_AFFINE>(a, s);
        default: return kmt_launch<T, KM_COORD_HOMOGRAPHY>(a, s);
    }
}

// 1 if the tile-owner kernel computes grad_src for these modes (bilinear, zeros/fill padding, fp32 compute)
int km_warp_bwd_tiled_supported(int interp, int pad, int dtype, const void* gsrc) {
    if (km_config().warp_bwd_generic) return 0;  // KM_WARP_BWD_ALGO=generic forces the atomic scatter kernel
    return (interp == KM_INTERP_BILINEAR && (pad == KM_PAD_ZEROS || pad == KM_PAD_FILL) && dtype != KM_F64 && gsrc != nullptr) ? 1 : 0;
}

// 32-bit byte offsets inside a grad_out plane (the caller falls back to the generic kernel otherwise)
int km_warp_bwd_tiled_dims_ok(int h, int w) { return ((uint64_t)h * (uint64_t)w * 4 < (1ull << 32)) ? 1 : 0; }

// grad_src (pad zeros / fill does not change it)
int km_warp_bwd_tiled_run(const void* gout, const void* mat, void* gsrc, int B, int C, int H, int W, int h, int w, int B_M, int coord_mode, int norm_coords,
                          int pad, int align, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmt_run<float>(gout, mat, gsrc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, s);
#ifndef KMT_DEV_F32_ONLY  // (development builds: one storage type compiles in a third of the time)
        case KM_BF16: return kmt_run<km_bf16>(gout, mat, gsrc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, s);
        default: return kmt_run<km_f16>(gout, mat, gsrc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, s);
#else
        default: return -1;
#endif
    }
}

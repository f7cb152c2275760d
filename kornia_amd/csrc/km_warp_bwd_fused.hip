// kornia_amd - BOTH gradients of the bilinear warps (zeros / fill padding) from ONE read of grad_out, for gfx950:
// grad wrt the image (the tile-owner scatter of km_warp_bwd_tiled.hip) and grad wrt the (B,3,3) matrix (km_warp_gm.hip) in
// one persistent launch.  Replaces the autograd backward of kornia/geometry/transform/imgwarp.py:143-174 (warp_perspective),
// :246-290 (warp_affine) and :1539-1546 (homography_warp) when both the image and the matrix require a gradient.
//
// Why it is one kernel now.  The two launches read grad_out twice (4e bytes per element against the op's 3e, SURVEY.md 8(d):
// 3.31 GB moved for 2.42 GB at 256x3x512^2).  Three earlier one-read forms lost to them because the matrix
// gradient's source taps were GATHERED from global memory inside the owner loop.  Here the owner of a source tile already has
// that tile on chip:
//
//   * the matrix gradient is linear in the tap values: dL/dM = sum_q J_q^T sum_taps dw_tap/d(x,y) * src[tap] * grad_out[q].
//     Every tap belongs to exactly one tile, so each owner adds the terms of the taps IT owns - the same ownership rule that
//     makes the scatter add every contribution to grad_src exactly once;
//   * the owner keeps its 64 x 64 x C source tile in LDS next to the 64 x 64 x C fixed-point accumulators (96 KB of the CU's
//     160 KB): all tap reads are ds_read, no gathers, no texture-address traffic beyond the two coalesced streams;
//   * one workgroup per CU, PERSISTENT, walking runs of horizontally adjacent tiles.  Everything a tile needs from HBM - the
//     box of grad_out pixels that can touch it (KMO_SLOTS pixels per thread) and its source tile - is requested one tile
//     AHEAD into registers: the source tile when the previous tile's scatter starts, the grad_out pixels slot by slot as the
//     previous tile's slots are consumed (one register set holds both tiles).  The loads of tile k + 1 fly while tile k is
//     scattered and the flush of tile k drains while tile k + 1 starts: the memory pipe of the CU never waits for a phase;
//   * because the whole box is in registers before the first contribution is added, the fixed-point scale comes from the
//     EXACT maximum of |grad_out| over the box: no speculation, no redo pass, 3 more bits than km_warp_bwd_tiled.hip;
//   * barriers inside the tile loop wait for LDS only (KM_LDS_BARRIER): a __syncthreads() would drain the loads in flight.
//
// Three launches, so that the persistent loop carries nothing it rarely needs (inlined, the general path and the box computation cost
// the loop 250 spilled registers; as function calls, more):
//   1. km_warp_bwd_boxes_kernel   one THREAD per tile: its box (kmt_tile_box, ~400 dependent instructions), head-room bits and
//                                 class -> an 80-byte record of the caller's workspace;
//   2. km_warp_bwd_fused_kernel   the persistent loop over the REGULAR tiles (box fits the registers, fixed point accurate, division
//                                 operands in range).  A regular tile that meets a non-finite gradient marks itself and leaves;
//   3. km_warp_bwd_general_kernel the remaining tiles (minification beyond ~1.2x, the vanishing line, > 7x magnification, NaN / inf in
//                                 grad_out): bands of the box, one pixel per thread and pass, IEEE float LDS atomics where fixed point
//                                 is not accurate enough - the same arithmetic, slower.  Exits at once when there are none.
//
// HBM traffic: grad_out ~1.07x (box overlap, served by L2) + src once + grad_src once = 3e bytes / element.
//
// Built with -fno-slp-vectorize (kornia_amd/build.py): a packed v_pk_fma_f32 names a register PAIR, and when the unused half of the pair
// is a slot with a load in flight the compiler waits for that load - in the middle of the scatter.  What the tile loop looks like is
// shaped by where the compiler's waits land (it cannot count loads issued under a branch, and it orders every write of a register
// after the loads in flight): see the comments at kmo_stage, kmo_issue_src and kmo_process.
#include <stdlib.h>

#include <atomic>

#include "km_warp_gm_rows.h"
#include "km_warp_tile.h"

// KMO_PROFILE (variant libraries only, profiles/time_bwd_phases.py): every wave sums the shader cycles (s_memtime) it spends in each phase
// of the tile loop; km_debug_fused_profile() copies the table out.  Costs ~10 % of the kernel; the default build has none of it.
#ifdef KMO_PROFILE
#define KMO_PROF_PHASES 16  // 0-7: the phases of the tile loop; 8-15: parts of the stage (8-10) and of the description (11-14)
__device__ unsigned long long kmo_prof_out[512 * 16 * KMO_PROF_PHASES];
__device__ unsigned long long kmo_prof_rt[512 * 4];  // per worker: start, end on the constant-rate counter (s_memrealtime, 100 MHz), XCC id, -
#define KMO_T(k)                                                              \
    {                                                                         \
        KM_SCHED_FENCE();                                                     \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();           \
        prof[k] += t_ - tlast;                                                \
        tlast = t_;                                                           \
        KM_SCHED_FENCE();                                                     \
    }
#define KMO_PROF_PARAMS , unsigned long long (&prof)[KMO_PROF_PHASES], unsigned long long& tlast
#define KMO_PROF_ARGS , prof, tlast
#else
#define KMO_T(k)
#define KMO_PROF_PARAMS
#define KMO_PROF_ARGS
#endif

#ifndef KMO_ABL
#define KMO_ABL 0  // timing experiments only (wrong results): 1 no matrix-gradient work, 2 no LDS atomics, 4 no per-pixel work; round 6 - what the phases OUTSIDE the
                   // scatter cost when they are simply gone (the upper bound of overlapping them): 8 no flush, 16 no image-end sums, 32 no barrier B1, 64 no barrier B2,
                   // 128 no maximum of |grad_out| in the stage (profiles/r06/run6_*)
#endif
#ifndef KMO_NT
#define KMO_NT 1024        // threads per workgroup (one workgroup per CU: 16 waves)
#endif
#define KMO_NW (KMO_NT / 64)
#ifndef KMO_SLOTS
#define KMO_SLOTS 6        // grad_out pixels per thread held in registers: boxes up to KMO_NT * KMO_SLOTS pixels are "regular"
#endif
#ifndef KMO_TH
#define KMO_TH 64          // tile height (the tile width is KMT_TW = 64: the flush maps 16 lanes to a 256-byte row)
#endif
#define KMO_PLANE (KMO_TH * KMT_TW)
#ifndef KMO_WG_PER_CU
#define KMO_WG_PER_CU 1    // persistent workgroups per CU (what the LDS of a workgroup allows)
#endif
#ifndef KMO_REC_FIELDS
#define KMO_REC_FIELDS 1   // tile coordinates and walk steps of a tile from its workspace record instead of integer divisions in the tile loop
#endif
#ifndef KMO_GM_LDS
#define KMO_GM_LDS 1       // image-end sums of the matrix gradient through the (dead) source tile in LDS instead of nine wave reductions per wave
#endif
#ifndef KMO_RING_DESC
#define KMO_RING_DESC 1    // the LDS ring holds DESCRIBED tiles (wave 0 derives every block-uniform field once per 64 tiles, lane = tile) instead of workspace records
#endif
#ifndef KMO_RED_ATOMIC
#define KMO_RED_ATOMIC 1   // the maximum of |grad_out| over a pass through one LDS atomic word (two, alternating) instead of 16 wave partials read back by every wave
#endif
#ifndef KMO_FLUSH_STRAIGHT
#define KMO_FLUSH_STRAIGHT 1  // full tiles: the flush's three accumulator reads in flight together (one LDS round trip instead of three)
#endif
#ifndef KMO_SRC_AT
#define KMO_SRC_AT 0       // slot of the scatter after which the next tile's source tile is requested
#endif
#ifndef KMO_SRC_DMA
#define KMO_SRC_DMA 1      // fp32 storage: the next tile's source tile by LDS-DMA (global_load_lds) into a SECOND source buffer instead of through registers
#endif
#ifndef KMO_TAB_AHEAD
#define KMO_TAB_AHEAD 0    // the coordinate-table entries of a slot read one slot ahead of its arithmetic
#endif
#ifndef KMO_WALK_COLS
#define KMO_WALK_COLS 1    // the box is walked with a FIXED column per thread (thread = (row tid / bw, column tid % bw) of a slab of KMO_NT / bw rows; slot s is the pixel s slabs below): see KmoWalk
#endif
#ifndef KMO_READS_FIRST
#define KMO_READS_FIRST 1  // 1: tap by tap, a tap's three source reads in front of its three atomics (step 1.4395 -> 1.4364 ms, op 0.554 -> 0.545: profiles/r06/run18_*); 0: interleaved atomic, read, atomic, read, ...; 2: ALL twelve reads of a pixel in front of all its atomics, unconditional - measured 9 % SLOWER (op 0.593 against 0.545, run19: every pixel then issues the reads of taps it does not own)
#endif
#ifndef KMO_FENCE_EVERY
#define KMO_FENCE_EVERY 1  // slots of the scatter between two scheduling fences (1: one pixel at a time)
#endif
#define KMO_RUN (KMO_SRC_DMA ? 16 : 64)  // records wave 0 fetches from the workspace at once (lane = tile), into alternating halves of a 2 x KMO_RUN ring (the second source buffer takes 48 KB of the ring's CU)
#define KMO_RING_INTS (KMO_RING_DESC ? 32 : 20)  // ints per tile in the LDS ring
#define KMO_BOX_INTS 20    // workspace record of a tile: j0, j1, i0, i1, head-room bits, flags, first plane, matrix row, the 9 matrix entries, tile (tx, ty), (image, group), walk steps
enum { KMO_F_FIXED = 1, KMO_F_REGULAR = 4, KMO_F_NONFINITE = 8 };

template <typename T>
struct KmWarpFusedArgs {
    const T* gout;       // (B,C,h,w)
    const T* src;        // (B,C,H,W)
    const float* mat;    // (B_M,9)
    float* gsrc;         // (B,C,H,W) fp32, written completely
    double* gmat;        // (B_M,9) fp64 accumulators, pre-zeroed
    const float* fill;   // (C), pad == fill only
    int* ws;             // (ntiles, KMO_BOX_INTS) workspace: the tiles' boxes and classes
    KmWarpGeom<float> g;
    uint32_t tiles_x, tiles_y, ntiles, nruns, run_len;  // a run = run_len horizontally adjacent tiles (run_len == tiles_x or 1)
    uint32_t nworkers;   // == gridDim.x of the persistent launch
    uint32_t general_tiles;  // tiles per group of the general launch (1 .. 64: one ballot)
    uint32_t general_grid;   // workgroups of the general launch (each walks its share of the groups)
    uint32_t reverse;    // the launch walks the batch backwards (km_traversal_next)
    uint32_t stream_out; // streaming stores of the tile flush (km_stream_stores)
    // Channel groups.  A launch covers channels c0 .. c0 + ngrp * cc - 1 of every image in groups of cc (= the kernel's CC: 3 or 1): an "image"
    // of the tile sequence is an (image, group) pair, so any channel count runs through the same LDS tiles as RGB / grey - C = 3 a + r takes one
    // launch sequence with CC = 3 (a groups) and one with CC = 1 (r groups).  All groups of an image add to the same matrix gradient.
    uint32_t c0, ngrp, cc;
    uint32_t first;      // the first launch sequence of the call: it zeroes gmat (boxes) and scans for unvisited non-finite gradients (general)
    uint32_t scan;       // ... unless the launch policy says not to (KM_WARP_BWD_SCAN=0 / km_config_set("warp_bwd_no_scan", 1): see kmo_scan_unvisited)
};

__host__ __device__ constexpr int kmo_lds_bytes(int cc) {
    return (KMT_BAND_W + KMT_TAB) * 16 + (KMO_SRC_DMA ? 3 : 2) * cc * KMO_PLANE * 4 + 2 * KMO_RUN * KMO_RING_INTS * 4 + KMO_NW * 8 + KMO_NW * 9 * 8;
}

// One tile (everything block-uniform)
struct KmoTile {
    int t;             // linear tile index (b, ty, tx), < 0: the sequence has ended
    int b, X0, Y0, TWc, THc;   // b: (image, channel group) index of the sequence
    int plane0, bm;    // plane index of the group's first channel in src / gsrc / grad_out; row of mat / gmat
    int j0, j1, i0, i1, hb;
    bool fixed_ok, regular, svec, fvec;
    int bw, nq;
    int di, dj;        // row / column step of the linear walk over the box (KMO_NT / bw, KMO_NT % bw)
    int p, npass;      // the box is walked in npass passes of KMO_CAP pixels; this work item is pass p
};
#define KMO_CAP (KMO_NT * KMO_SLOTS)  // pixels of a box the registers of a workgroup hold

template <typename T>
__device__ __forceinline__ void kmo_tile_coords(const KmWarpFusedArgs<T>& a, uint32_t t, int& b, int& tx, int& ty) {
    const uint32_t row = t / a.tiles_x;
    tx = (int)(t - row * a.tiles_x);
    b = (int)(row / a.tiles_y);
    ty = (int)(row - (uint32_t)b * a.tiles_y);
}

// q-th tile of this worker: runs are dealt round-robin to the workers (the 32 workers of an XCD take consecutive runs, i.e.
// neighbouring tile rows of the same images, so the rows of grad_out that neighbouring boxes share stay in one L2)
template <typename T>
__device__ __forceinline__ int kmo_tile_of(const KmWarpFusedArgs<T>& a, uint32_t q) {
    const uint32_t R = a.run_len, NWK = a.nworkers, w = blockIdx.x;
    const uint32_t lw = (NWK % 8u == 0u) ? (w % 8u) * (NWK / 8u) + w / 8u : w;
    const uint32_t r = q / R, k = q - r * R;
    uint32_t rho = r * NWK + lw;
    if (rho >= a.nruns) return -1;
    if (a.reverse) rho = a.nruns - 1u - rho;
    return (int)(rho * R + k);  // runs tile the (image, tile row, tile column) order: run_len divides tiles_x
}

// ---- launch 1: boxes and classes of all tiles, one thread per tile ---------------------------------------------------------------
template <typename T, int CM>
__global__ __launch_bounds__(256) void km_warp_bwd_boxes_kernel(const KmWarpFusedArgs<T> a) {
    const KmWarpGeom<float>& g = a.g;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    // the fp64 accumulators of the matrix gradient start at zero: this launch precedes every atomic on them (the caller need not zero)
    if (a.first && a.gmat)
        for (uint32_t k = t; k < (uint32_t)g.B_M * 9u; k += gridDim.x * 256u) a.gmat[k] = 0.0;
    if (t >= a.ntiles) return;
    int bg, tx, ty;
    kmo_tile_coords(a, t, bg, tx, ty);
    const int b = bg / (int)a.ngrp, grp = bg - b * (int)a.ngrp;  // (image, channel group)
    float m[9];
    const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = mp[k];
    const int X0 = tx * KMT_TW, Y0 = ty * KMO_TH;
    const int X1 = min(X0 + KMT_TW, g.W), Y1 = min(Y0 + KMO_TH, g.H);
    const KmtBox bx = kmt_tile_box_padded<CM>(g, m, X0, X1, Y0, Y1);
    const int bw = bx.j1 - bx.j0 + 1, bh = bx.i1 - bx.i0 + 1;
    const bool empty = bw <= 0 || bh <= 0;
    // one band of the coordinate tables; larger boxes than the registers hold are walked in several passes (rotations, magnification)
    const bool fits = empty || (bw <= KMT_BAND_W && bh <= KMT_TAB);
    // every pixel of the box has division operands inside the range of the shared-reciprocal division (km_lean.h)
    const bool fast = empty ? true : kml_div_guard<CM>(g, m, bx.j0, bx.j1, bx.i0, bx.i1);
    int* o = a.ws + (size_t)t * KMO_BOX_INTS;
    o[0] = bx.j0; o[1] = bx.j1; o[2] = bx.i0; o[3] = bx.i1;
    o[4] = (int)ceilf(log2f(fmaxf(bx.mult, 1.f))) + 1;  // head-room bits
    o[5] = (bx.fixed_ok ? KMO_F_FIXED : 0) | ((fits && fast && bx.fixed_ok) ? KMO_F_REGULAR : 0);
    o[6] = b * g.C + (int)a.c0 + grp * (int)a.cc;  // plane index of the group's first channel (< 2^31: checked by the host)
    o[7] = g.B_M == 1 ? 0 : b;                      // row of mat / gmat
#pragma unroll
    for (int k = 0; k < 9; ++k) o[8 + k] = __float_as_int(m[k]);  // (the tile loop reads its matrix from LDS, not through a vector load)
    // what the persistent loop would otherwise derive with integer divisions, executed by all 16 waves of a workgroup for every tile
    const int bwc = max(bw, 1);
    o[17] = tx | (ty << 16);                               // tile column / row (< 2^16 each: checked by the host)
    o[18] = bg;                                            // (image, channel group) index of the sequence
    o[19] = (KMO_NT / bwc) | ((KMO_NT % bwc) << 16);       // row / column step of the linear walk over the box, KMO_NT pixels apart
}

// tile t from its (block-uniform) workspace record
// the matrix of a tile from its record (block-uniform values in vector registers: what the caller does not use costs nothing)
__device__ __forceinline__ void kmo_matrix(const int* rec, float (&m)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = __int_as_float(rec[8 + k]);
}
// record of the q-th tile of this worker in the LDS ring
__device__ __forceinline__ const int* kmo_ring(const int* s_box, uint32_t q) { return s_box + (q % (2u * KMO_RUN)) * KMO_RING_INTS; }

// U: the record is block-uniform (read through LDS / scalar registers); !U: one record per lane (kmo_fetch_boxes)
template <typename T, bool U = true>
__device__ __forceinline__ void kmo_describe(const KmWarpFusedArgs<T>& a, int t, const int* rec, KmoTile& d) {
    const KmWarpGeom<float>& g = a.g;
    auto un = [](int v) { return U ? kmt_uniform(v) : v; };
    d = KmoTile{};
    d.t = t;
    if (t < 0) return;
    int tx, ty;
#if KMO_REC_FIELDS
    {
        const int txy = un(rec[17]);
        tx = txy & 0xffff; ty = (int)((uint32_t)txy >> 16);
        d.b = un(rec[18]);
        const int st = un(rec[19]);
        d.di = st & 0xffff; d.dj = (int)((uint32_t)st >> 16);
    }
#else
    kmo_tile_coords(a, (uint32_t)t, d.b, tx, ty);
#endif
    d.X0 = tx * KMT_TW; d.Y0 = ty * KMO_TH;
    d.TWc = min(d.X0 + KMT_TW, g.W) - d.X0; d.THc = min(d.Y0 + KMO_TH, g.H) - d.Y0;
    d.j0 = un(rec[0]); d.j1 = un(rec[1]); d.i0 = un(rec[2]); d.i1 = un(rec[3]);
    d.hb = un(rec[4]);
    d.plane0 = un(rec[6]); d.bm = un(rec[7]);
    const int fl = un(rec[5]);
    d.fixed_ok = (fl & KMO_F_FIXED) != 0; d.regular = (fl & KMO_F_REGULAR) != 0;
    const int bw = d.j1 - d.j0 + 1, bh = d.i1 - d.i0 + 1;
    d.bw = bw;
#if !KMO_REC_FIELDS
    d.di = un(KMO_NT / max(bw, 1)); d.dj = un(KMO_NT % max(bw, 1));
#endif
    d.nq = (bw > 0 && bh > 0) ? bw * bh : 0;
    d.p = 0;
#if KMO_WALK_COLS
    d.npass = d.regular ? max(1, (bh + KMO_SLOTS * max(d.di, 1) - 1) / (KMO_SLOTS * max(d.di, 1))) : 1;  // (a pass = KMO_SLOTS slabs of di rows)
#else
    d.npass = d.regular ? max(1, (d.nq + KMO_CAP - 1) / KMO_CAP) : 1;  // (a tile of the general launch is ONE item of the persistent loop, whatever its box)
#endif
    // 16-byte rows of the source tile / of the tile flush
    d.svec = (sizeof(T) == 4) && d.TWc == KMT_TW && (g.W & 3) == 0 && ((uintptr_t)a.src & 15) == 0;
    d.fvec = d.TWc == KMT_TW && (g.W & 3) == 0 && ((uintptr_t)a.gsrc & 15) == 0;
}

// records of tiles q0 .. q0 + 63 of this worker, workspace -> LDS, one lane per tile (wave 0)
template <typename T>
__device__ __forceinline__ void kmo_fetch_boxes(const KmWarpFusedArgs<T>& a, uint32_t q0, int lane, int* s_box) {
    if (lane >= KMO_RUN) return;  // (lane = tile of the run)
    const int t = kmo_tile_of(a, q0 + (uint32_t)lane);
    static_assert(KMO_BOX_INTS % 4 == 0 && KMO_RING_INTS % 4 == 0 && KMO_RUN <= 64, "16-byte pieces; one lane per tile of a run");
    int4* dst = reinterpret_cast<int4*>(s_box) + (KMO_RING_INTS / 4) * (((q0 / KMO_RUN) & 1u) * KMO_RUN + (uint32_t)lane);
#if KMO_RING_DESC
    // The description of a tile is ~150 scalar instructions (and three LDS round trips) that all 16 waves of the workgroup executed for
    // every tile, between the barrier and the scatter: here wave 0 derives it once, 64 tiles at a time in its vector lanes, and the
    // tile loop reads the finished fields (kmo_load_desc).  Entries 8 .. 16 are the matrix, where kmo_matrix expects it.
    KmoTile d;
    const int* rec = a.ws + (size_t)max(t, 0) * KMO_BOX_INTS;
    kmo_describe<T, false>(a, t, rec, d);
    int4 mq[3] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
    if (t >= 0) { mq[0] = *reinterpret_cast<const int4*>(rec + 8); mq[1] = *reinterpret_cast<const int4*>(rec + 12); mq[2].x = rec[16]; }
    const int flags = (d.fixed_ok ? 1 : 0) | (d.regular ? 2 : 0) | (d.svec ? 4 : 0) | (d.fvec ? 8 : 0);
    dst[0] = make_int4(d.t, d.j0, d.i0, d.i1);
    dst[1] = make_int4(d.hb, flags, d.plane0, d.bm);
    dst[2] = mq[0];
    dst[3] = mq[1];
    dst[4] = make_int4(mq[2].x, d.X0, d.Y0, d.TWc);
    dst[5] = make_int4(d.THc, d.bw, d.nq, d.di);
    dst[6] = make_int4(d.dj, d.npass, d.b, d.j1);
#else
    if (t >= 0) {
        const int4* rec = reinterpret_cast<const int4*>(a.ws + (size_t)t * KMO_BOX_INTS);
#pragma unroll 1
        for (int k = 0; k < KMO_BOX_INTS / 4; ++k) dst[k] = rec[k];
    } else {
        dst[0] = make_int4(0, -1, 0, -1);
        dst[1] = make_int4(0, 0, 0, 0);
    }
#endif
}

#if KMO_RING_DESC
// the described tile at r (kmo_fetch_boxes) -> block-uniform registers: seven 16-byte LDS reads in flight together, no arithmetic
__device__ __forceinline__ void kmo_load_desc(const int* r, KmoTile& d) {
    const int4* p = reinterpret_cast<const int4*>(r);
    const int4 a0 = p[0], a1 = p[1], a4 = p[4], a5 = p[5], a6 = p[6];
    d.t = kmt_uniform(a0.x); d.j0 = kmt_uniform(a0.y); d.i0 = kmt_uniform(a0.z); d.i1 = kmt_uniform(a0.w);
    d.hb = kmt_uniform(a1.x);
    const int fl = kmt_uniform(a1.y);
    d.fixed_ok = (fl & 1) != 0; d.regular = (fl & 2) != 0; d.svec = (fl & 4) != 0; d.fvec = (fl & 8) != 0;
    d.plane0 = kmt_uniform(a1.z); d.bm = kmt_uniform(a1.w);
    d.X0 = kmt_uniform(a4.y); d.Y0 = kmt_uniform(a4.z); d.TWc = kmt_uniform(a4.w);
    d.THc = kmt_uniform(a5.x); d.bw = kmt_uniform(a5.y); d.nq = kmt_uniform(a5.z); d.di = kmt_uniform(a5.w);
    d.dj = kmt_uniform(a6.x); d.npass = kmt_uniform(a6.y); d.b = kmt_uniform(a6.z); d.j1 = kmt_uniform(a6.w);
    d.p = 0;
}
#endif

// first pixel of this thread in a box of width bw walked as a linear list (element e = tid; tid < 2^24: the float quotient is off by at most one)
__device__ __forceinline__ void kmo_first(int tid, int bw, int& qi, int& qj) {  // tid: element index (< 2^24)
    // (a reciprocal, not a division: 15 instructions fewer per thread and tile; the two corrections below absorb its error)
    qi = (int)(((float)tid + 0.5f) * kmt_uniform(__builtin_amdgcn_rcpf((float)bw)));
    qj = tid - qi * bw;
    if (qj < 0) { qi -= 1; qj += bw; }
    if (qj >= bw) { qi += 1; qj -= bw; }
}

// ---- requests of one tile: its box of grad_out (regular tiles) and its source tile -----------------------------------------------
// Every load below is UNCONDITIONAL (rows / columns beyond a ragged tile are clamped to its last one and masked when the registers are
// stored to LDS): a zero-initialised register + conditional load is a write the compiler orders after every load in flight.
template <typename T, int CC>
__device__ __forceinline__ void kmo_issue_src(const KmWarpFusedArgs<T>& a, const KmoTile& d, float (&S)[CC * KMO_PLANE / KMO_NT]) {
    constexpr int SREG = CC * KMO_PLANE / KMO_NT;
    static_assert(CC * KMO_PLANE % (4 * KMO_NT) == 0, "whole 16-byte pieces per thread");
    const KmWarpGeom<float>& g = a.g;
    const size_t src_plane = (size_t)g.H * g.W;
    const T* src_b = a.src + (size_t)d.plane0 * src_plane;
    const int tid = threadIdx.x;
    if (d.svec) {
#pragma unroll
        for (int k = 0; k < SREG / 4; ++k) {
            const int idx4 = k * KMO_NT + tid;
            const int c = idx4 / (KMO_PLANE / 4), rem = idx4 % (KMO_PLANE / 4);
            const int r = min(rem / (KMT_TW / 4), d.THc - 1), x4 = (rem % (KMT_TW / 4)) * 4;
            const float* p = reinterpret_cast<const float*>(src_b) + (size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (size_t)(d.X0 + x4);
            KM_CHECK_ALIGNED(p, 16);
            const float4 v = *reinterpret_cast<const float4*>(p);
            S[4 * k + 0] = v.x; S[4 * k + 1] = v.y; S[4 * k + 2] = v.z; S[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SREG; ++k) {
            const int idx = k * KMO_NT + tid;
            const int c = idx / KMO_PLANE, rem = idx % KMO_PLANE;
            const int r = min(rem / KMT_TW, d.THc - 1), x = min(rem % KMT_TW, d.TWc - 1);
            S[k] = (float)km_ld(src_b + (size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (size_t)(d.X0 + x));
        }
    }
}

// the source tile into LDS, as (v - fill) for pad == fill (the rounding sequence of the oracle: (v - fill) first).  Cells beyond a ragged
// tile hold copies of its last row / column: no tap ever lands there (the in-tile test of kmo_pix uses the tile's real size).
template <int CC>
__device__ __forceinline__ void kmo_store_src(const KmoTile& d, const float (&S)[CC * KMO_PLANE / KMO_NT], float* s_src, const float (&fill)[CC], bool is_fill) {
    constexpr int SREG = CC * KMO_PLANE / KMO_NT;
    const int tid = threadIdx.x;
    if (d.svec) {
#pragma unroll
        for (int k = 0; k < SREG / 4; ++k) {
            const int idx4 = k * KMO_NT + tid;
            float f = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) f = (idx4 / (KMO_PLANE / 4) == c) ? fill[c] : f;
            float4 v = make_float4(S[4 * k + 0], S[4 * k + 1], S[4 * k + 2], S[4 * k + 3]);
            if (is_fill) { v.x -= f; v.y -= f; v.z -= f; v.w -= f; }
            reinterpret_cast<float4*>(s_src)[idx4] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SREG; ++k) {
            const int idx = k * KMO_NT + tid;
            float f = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) f = (idx / KMO_PLANE == c) ? fill[c] : f;
            s_src[idx] = is_fill ? S[k] - f : S[k];
        }
    }
}

// fp32 storage (KMO_SRC_DMA): the same pieces of the source tile, every lane's straight into LDS (KM_GLDS16: the wave's 64 pieces land at
// consecutive 16-byte cells from a wave-uniform base - exactly kmo_store_src's layout, idx4 = k * KMO_NT + tid).  No register holds the tile
// on its way, no ds_write: what kmo_issue_src + kmo_store_src do in two phases a tile apart is ONE request here, and the 12 registers of S
// are free during the scatter.  Ragged / unaligned tiles: one float per lane (a tile row per wave instruction), the same clamped addresses.
// pad == fill stages the raw values; kmo_fill_in_lds subtracts the fill once they have landed.
template <typename T, int CC>
__device__ __forceinline__ void kmo_dma_src(const KmWarpFusedArgs<T>& a, const KmoTile& d, float* s_dst) {
    static_assert(sizeof(T) == 4, "LDS-DMA moves bytes: fp32 storage only");
    constexpr int SREG = CC * KMO_PLANE / KMO_NT;
    const KmWarpGeom<float>& g = a.g;
    const size_t src_plane = (size_t)g.H * g.W;
    const float* src_b = reinterpret_cast<const float*>(a.src) + (size_t)d.plane0 * src_plane;
    const int tid = threadIdx.x;
    const int wbase = kmt_uniform(tid & ~63);  // first thread of the wave
    if (d.svec) {
#pragma unroll
        for (int k = 0; k < SREG / 4; ++k) {
            const int idx4 = k * KMO_NT + tid;
            const int c = idx4 / (KMO_PLANE / 4), rem = idx4 % (KMO_PLANE / 4);
            const int r = min(rem / (KMT_TW / 4), d.THc - 1), x4 = (rem % (KMT_TW / 4)) * 4;
            const float* p = src_b + (size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (size_t)(d.X0 + x4);
            KM_CHECK_ALIGNED(p, 16);
            KM_GLDS16(p, s_dst + 4 * (k * KMO_NT + wbase));
        }
    } else {
#pragma unroll
        for (int k = 0; k < SREG; ++k) {
            const int idx = k * KMO_NT + tid;
            const int c = idx / KMO_PLANE, rem = idx % KMO_PLANE;
            const int r = min(rem / KMT_TW, d.THc - 1), x = min(rem % KMT_TW, d.TWc - 1);
            KM_GLDS4(src_b + (size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (size_t)(d.X0 + x), s_dst + (k * KMO_NT + wbase));
        }
    }
}
// pad == fill: (v - fill) in place, every thread on the cells its own requests filled (they have landed: the caller waited for them)
template <int CC>
__device__ __forceinline__ void kmo_fill_in_lds(const KmoTile& d, float* s_src, const float (&fill)[CC]) {
    constexpr int SREG = CC * KMO_PLANE / KMO_NT;
    const int tid = threadIdx.x;
    if (d.svec) {
#pragma unroll
        for (int k = 0; k < SREG / 4; ++k) {
            const int idx4 = k * KMO_NT + tid;
            float f = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) f = (idx4 / (KMO_PLANE / 4) == c) ? fill[c] : f;
            float4 v = reinterpret_cast<float4*>(s_src)[idx4];
            v.x -= f; v.y -= f; v.z -= f; v.w -= f;
            reinterpret_cast<float4*>(s_src)[idx4] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SREG; ++k) {
            const int idx = k * KMO_NT + tid;
            float f = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) f = (idx / KMO_PLANE == c) ? fill[c] : f;
            s_src[idx] = s_src[idx] - f;
        }
    }
}

// ---- one visited output pixel: contributions to grad_src (taps inside the tile) and this tile's share of its matrix-gradient terms ----
// FIXED: int32 fixed-point LDS accumulators; otherwise IEEE float LDS atomics (non-finite gradients, vanishing-line tiles, extreme
// magnification).  The image gradient at the sample is  gix = sum_c g_c ((ne - nw) wy1 + (se - sw) wy0),  giy = sum_c g_c ((sw - nw) wx1 +
// (se - ne) wx0)  (km_warp_gm.hip): tap by tap,  nw: (-wy1, -wx1),  ne: (+wy1, -wx0),  sw: (-wy0, +wx1),  se: (+wy0, +wx0)  times
// dot = sum_c g_c src[tap, c].  A tap outside the tile is another owner's (or outside the image: it does not exist in the reference's sum).
template <int CM, int CC, bool FAST, bool FIXED>
__device__ __forceinline__ void kmo_pix(const KmtPix& q, const float (&go)[CC], int* s_acc, const float* s_src, float scale, uint32_t TWc, uint32_t THc,
                                        float u, float v, float mx, float my, float (&A)[9]) {
    const KmlTaps& t = q.t;
    const uint32_t ux = q.ux, uy = q.uy;
    const bool in_x0 = ux < TWc, in_x1 = (ux + 1u) < TWc;
    const bool in_y0 = uy < THc, in_y1 = (uy + 1u) < THc;
    bool t00 = in_x0 && in_y0, t01 = in_x1 && in_y0, t10 = in_x0 && in_y1, t11 = in_x1 && in_y1;
    if (!FIXED) {
        const bool num = (q.x == q.x) & (q.y == q.y);  // a NaN position touches nothing (ATen: the converted index is out of bounds)
        t00 = t00 && num; t01 = t01 && num; t10 = t10 && num; t11 = t11 && num;
    }
    // scale = 2^k: (wx * wy) * scale == wx * (wy * scale) bit for bit.  Tap-outer order: one exec-mask region per tap (the four taps
    // without exec-mask regions - zero weights, non-owned taps parked on a cell of the lane's own - and a wave-uniform skip of pixels that
    // own no tap measured the same time: profiles/r04/bwd_fused_per_pixel_variants.txt).
    const float wy0s = FIXED ? t.wy0 * scale : t.wy0, wy1s = FIXED ? t.wy1 * scale : t.wy1;
    const float w00 = t.wx1 * wy1s, w01 = t.wx0 * wy1s, w10 = t.wx1 * wy0s, w11 = t.wx0 * wy0s;
    float gix = 0.f, giy = 0.f;
    const int l00 = (int)(uy * (uint32_t)KMT_TW + ux);
#if KMO_READS_FIRST == 2
    if (FIXED) {
        // ALL twelve source reads of the pixel in front of ALL its atomics, unconditional (a tap another tile owns reads the plane's first cell and
        // its product is dropped): LDS operations of a wave complete in order, so the pixel's critical path - position -> source values -> terms -
        // waits for ONE round trip, not for four that each queue behind the previous tap's atomics.  The sums are the same fma chains in the same
        // order (a dropped tap adds +-w * 0 = 0 to a finite sum: FIXED tiles have finite positions).
        const int b00 = t00 ? l00 : 0, b01 = t01 ? l00 : -1, b10 = t10 ? l00 : -KMT_TW, b11 = t11 ? l00 : -(KMT_TW + 1);
        float sv_[4][CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            sv_[0][c] = (KMO_ABL & 1) ? 0.f : s_src[b00 + c * KMO_PLANE];
            sv_[1][c] = (KMO_ABL & 1) ? 0.f : s_src[b01 + c * KMO_PLANE + 1];
            sv_[2][c] = (KMO_ABL & 1) ? 0.f : s_src[b10 + c * KMO_PLANE + KMT_TW];
            sv_[3][c] = (KMO_ABL & 1) ? 0.f : s_src[b11 + c * KMO_PLANE + KMT_TW + 1];
        }
#define KMO_TAP_ADD(pred, OFF, W)                                                                                     \
        if ((pred) && !(KMO_ABL & 2)) {                                                                               \
            _Pragma("unroll") for (int c = 0; c < CC; ++c) atomicAdd(s_acc + l00 + c * KMO_PLANE + (OFF), kmt_quant((W) * go[c])); \
        }
        KMO_TAP_ADD(t00, 0, w00)
        KMO_TAP_ADD(t01, 1, w01)
        KMO_TAP_ADD(t10, KMT_TW, w10)
        KMO_TAP_ADD(t11, KMT_TW + 1, w11)
#undef KMO_TAP_ADD
        float dot_[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) d = (c == 0) ? go[c] * sv_[k][c] : km_fma(go[c], sv_[k][c], d);
            dot_[k] = d;
        }
        dot_[0] = t00 ? dot_[0] : 0.f; dot_[1] = t01 ? dot_[1] : 0.f; dot_[2] = t10 ? dot_[2] : 0.f; dot_[3] = t11 ? dot_[3] : 0.f;
        gix = km_fma(-t.wy1, dot_[0], gix); giy = km_fma(-t.wx1, dot_[0], giy);
        gix = km_fma(+t.wy1, dot_[1], gix); giy = km_fma(-t.wx0, dot_[1], giy);
        gix = km_fma(-t.wy0, dot_[2], gix); giy = km_fma(+t.wx1, dot_[2], giy);
        gix = km_fma(+t.wy0, dot_[3], gix); giy = km_fma(+t.wx0, dot_[3], giy);
    } else {
#endif
#if KMO_READS_FIRST
    // the tap's source values are REQUESTED IN FRONT of its atomics: LDS operations of a wave complete in order, so a read queued behind the
    // atomics waits for them as well - and the wait for the reads is on the wave's critical path (position -> taps -> terms), the atomics are not
#define KMO_TAP(pred, OFF, W, SX, WX, SY, WY)                                                        \
    if (pred) {                                                                                       \
        float sv_[CC];                                                                                \
        _Pragma("unroll") for (int c = 0; c < CC; ++c) sv_[c] = (KMO_ABL & 1) ? 0.f : s_src[l00 + c * KMO_PLANE + (OFF)]; \
        _Pragma("unroll") for (int c = 0; c < CC; ++c) {                                              \
            if (!(KMO_ABL & 2)) {                                                                     \
                if (FIXED) atomicAdd(s_acc + l00 + c * KMO_PLANE + (OFF), kmt_quant((W) * go[c]));    \
                else atomicAdd((float*)s_acc + l00 + c * KMO_PLANE + (OFF), (W) * go[c]);             \
            }                                                                                         \
        }                                                                                             \
        float dot = 0.f;                                                                              \
        _Pragma("unroll") for (int c = 0; c < CC; ++c) dot = (c == 0) ? go[c] * sv_[c] : km_fma(go[c], sv_[c], dot); \
        gix = km_fma(SX (WX), dot, gix);                                                              \
        giy = km_fma(SY (WY), dot, giy);                                                              \
    }
#else
#define KMO_TAP(pred, OFF, W, SX, WX, SY, WY)                                                        \
    if (pred) {                                                                                       \
        float dot = 0.f;                                                                              \
        _Pragma("unroll") for (int c = 0; c < CC; ++c) {                                              \
            if (!(KMO_ABL & 2)) {                                                                     \
                if (FIXED) atomicAdd(s_acc + l00 + c * KMO_PLANE + (OFF), kmt_quant((W) * go[c]));    \
                else atomicAdd((float*)s_acc + l00 + c * KMO_PLANE + (OFF), (W) * go[c]);             \
            }                                                                                         \
            if (!(KMO_ABL & 1)) {                                                                     \
                const float sv = s_src[l00 + c * KMO_PLANE + (OFF)];                                  \
                dot = (c == 0) ? go[c] * sv : km_fma(go[c], sv, dot);                                 \
            }                                                                                         \
        }                                                                                             \
        gix = km_fma(SX (WX), dot, gix);                                                              \
        giy = km_fma(SY (WY), dot, giy);                                                              \
    }
#endif
    KMO_TAP(t00, 0, w00, -, t.wy1, -, t.wx1)
    KMO_TAP(t01, 1, w01, +, t.wy1, -, t.wx0)
    KMO_TAP(t10, KMT_TW, w10, -, t.wy0, +, t.wx1)
    KMO_TAP(t11, KMT_TW + 1, w11, +, t.wy0, +, t.wx0)
#undef KMO_TAP
#if KMO_READS_FIRST == 2
    }
#endif
    // this tile's share of the pixel's terms (zero when it owns none of the taps: no 0 * inf from a degenerate position)
    const bool any = t00 | t01 | t10 | t11;
    float ax, ay, az;
    kmg_terms<CM, FAST>(q.p, gix * mx, giy * my, ax, ay, az);
    ax = any ? ax : 0.f; ay = any ? ay : 0.f; az = any ? az : 0.f;
    A[0] = km_fma(ax, u, A[0]); A[1] = km_fma(ax, v, A[1]); A[2] += ax;
    A[3] = km_fma(ay, u, A[3]); A[4] = km_fma(ay, v, A[4]); A[5] += ay;
    A[6] = km_fma(az, u, A[6]); A[7] = km_fma(az, v, A[7]); A[8] += az;
}

struct KmoConsts {
    float Wm1, Hm1, hW, hH, mx, my;
};

// Walk of a box.
// KMO_WALK_COLS (round 6): thread tid sits at (row tid / bw, column tid % bw) of a SLAB of P = KMO_NT / bw whole rows of the box (threads beyond
// P * bw idle: 44 of 1 024 on a 70-column box) and slot s is the pixel s slabs further down, in the SAME column.  Nothing of the walk is per-lane
// arithmetic any more: the column-table entry is read once per tile, the row index and the element offset advance by block-uniform steps (P,
// P * w), a slot is valid while s P + row < rows of the pass.  The linear list it replaces (element e = s * KMO_NT + tid, both coordinates
// advanced with carry for the current AND the next tile's box, a 32-bit multiply per request) was 29 of the 151 vector issue slots of a
// visited pixel; a 70 x 70 box is 5 slabs of 14 rows = 80 wave-slots where the list had 77.  A pass of a box larger than the registers hold
// is KMO_SLOTS slabs.
// Otherwise: a linear list of pixels, KMO_NT apart: element e = s * KMO_NT + tid sits at (qi, qj) of the box
struct KmoWalk {
#if KMO_WALK_COLS
    int trow;       // this thread's row in a slab (0x10000 for the idle threads beyond the slab)
    int nrows;      // rows of this pass (0: nothing to request)
    int P;          // rows of a slab
    uint32_t off;   // element offset of this thread's slot-0 pixel in a grad_out plane
    uint32_t step;  // ... from one slot to the next: P * w
#else
    int qi, qj, di, dj, bw, nq;
    uint32_t row0;  // element offset of the box's first pixel in a grad_out plane
#endif
};
#if KMO_WALK_COLS
// rows of pass d.p of a regular tile's box
__device__ __forceinline__ int kmo_pass_rows(const KmoTile& d) {
    const int P = max(d.di, 1);
    return (d.regular && d.nq > 0) ? min((d.i1 - d.i0 + 1) - d.p * KMO_SLOTS * P, KMO_SLOTS * P) : 0;
}
#endif
template <typename T>
__device__ __forceinline__ void kmo_walk_init(const KmWarpFusedArgs<T>& a, const KmoTile& d, KmoWalk& wk) {
#if KMO_WALK_COLS
    const int P = max(d.di, 1);
    int trow, tcol;
    kmo_first((int)threadIdx.x, max(d.bw, 1), trow, tcol);
    wk.P = P;
    wk.nrows = kmo_pass_rows(d);
    wk.off = (uint32_t)(d.i0 + d.p * KMO_SLOTS * P + trow) * (uint32_t)a.g.w + (uint32_t)(d.j0 + tcol);
    wk.trow = trow < P ? trow : 0x10000;
    wk.step = (uint32_t)P * (uint32_t)a.g.w;
#else
    wk.bw = max(d.bw, 1);
    wk.nq = d.regular ? min(d.nq - d.p * KMO_CAP, KMO_CAP) : 0;  // pixels of this pass (a tile of the general path loads its pixels itself)
    wk.di = d.di;
    wk.dj = d.dj;
    kmo_first(d.p * KMO_CAP + (int)threadIdx.x, wk.bw, wk.qi, wk.qj);
    wk.row0 = (uint32_t)d.i0 * (uint32_t)a.g.w + (uint32_t)d.j0;
#endif
}
__device__ __forceinline__ void kmo_walk_none(KmoWalk& wk) {  // nothing to request (the end of the sequence, a tile of the general launch)
#if KMO_WALK_COLS
    wk.nrows = 0;
#else
    wk.nq = 0;
#endif
}
// request slot s of the walk's tile (zeros beyond the end of the box) and advance
template <typename T, int CC>
__device__ __forceinline__ void kmo_request_slot(const T* const (&gout_c)[CC], int w, int s, KmoWalk& wk, float (&Gs)[CC]) {
#if KMO_WALK_COLS
    (void)w;
    const bool valid = wk.trow < wk.nrows - s * wk.P;
#pragma unroll
    for (int c = 0; c < CC; ++c) Gs[c] = 0.f;
    if (valid) kmt_load_go<T, CC>(gout_c, wk.off + (uint32_t)s * wk.step, Gs);
#else
    const bool valid = s * KMO_NT + (int)threadIdx.x < wk.nq;
#pragma unroll
    for (int c = 0; c < CC; ++c) Gs[c] = 0.f;
    if (valid) kmt_load_go<T, CC>(gout_c, wk.row0 + (uint32_t)wk.qi * (uint32_t)w + (uint32_t)wk.qj, Gs);
    kmt_advance(wk.qi, wk.qj, wk.di, wk.dj, wk.bw);
#endif
}

// The pixels a thread holds in registers (regular tiles): element e = s * KMO_NT + tid of the box walked as a linear list.  As soon
// as slot s has been consumed its registers are REFILLED with slot s of the next tile (wn, gout_n): one register set holds both tiles,
// and the next tile's requests are spread over the whole scatter instead of issued in one burst.
template <typename T, int CM, int ALIGN, int CC, bool FAST, bool FIXED, int PADX>
__device__ __forceinline__ void kmo_process(const float (&m)[9], const KmoTile& d, const KmoConsts& k, float (&G)[KMO_SLOTS][CC], const float4* s_u4,
                                            const float4* s_v4, int* s_acc, const float* s_src, float scale, float (&A)[9], int w, KmoWalk& wn,
                                            const T* const (&gout_n)[CC], bool mine, const KmWarpFusedArgs<T>& a, const KmoTile& nxt,
                                            float (&S)[CC * KMO_PLANE / KMO_NT], float* s_src_next) {
    const int tid = threadIdx.x;
    const int bw = max(d.bw, 1);
#if KMO_WALK_COLS
    static_assert(!KMO_TAB_AHEAD, "KMO_TAB_AHEAD belongs to the linear walk");
    const int P = max(d.di, 1);
    const int nrows = kmo_pass_rows(d);  // rows of this pass
    int trow, tcol;
    kmo_first(tid, bw, trow, tcol);
    const int rbase = d.p * KMO_SLOTS * P + trow;  // row-table index of this thread's slot-0 pixel
    trow = trow < P ? trow : 0x10000;
    const float4 c0 = s_u4[min(tcol, KMT_BAND_W - 1)];  // (the same column for every slot; a tile of the general launch may be wider than the table: nothing of it is used)
#pragma unroll
    for (int s = 0; s < KMO_SLOTS; ++s) {
        if (mine && s * P < nrows && !(KMO_ABL & 4)) {  // block-uniform
            const bool valid = trow < nrows - s * P;
            const float4 r0 = s_v4[valid ? rbase + s * P : 0];
            KmtPix q;
            float gdx, gdy;
            kmt_pix_position_pad<CM, ALIGN, FAST, PADX>(m, kmt_half(c0), kmt_half(r0), valid, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, a.g.W, a.g.H, q, gdx, gdy);
            kmo_pix<CM, CC, FAST, FIXED>(q, G[s], s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
        }
#else
    const int di = d.di, dj = d.dj;
    const int nqp = min(d.nq - d.p * KMO_CAP, KMO_CAP);  // pixels of this pass
    int qi, qj;
    kmo_first(d.p * KMO_CAP + tid, bw, qi, qj);
#if KMO_TAB_AHEAD
    // the table entries of a slot are read ONE SLOT AHEAD - in front of the previous pixel's twelve LDS atomics and tap reads, which an LDS
    // read issued behind them would queue behind (LDS operations of a wave complete in order)
    float4 cN = make_float4(0.f, 0.f, 0.f, 0.f), rN = cN;
    bool vN = false;
    if (mine && 0 < nqp && !(KMO_ABL & 4)) {
        vN = tid < nqp;
        cN = s_u4[vN ? qj : 0]; rN = s_v4[vN ? qi : 0];
        kmt_advance(qi, qj, di, dj, bw);
    }
#endif
#pragma unroll
    for (int s = 0; s < KMO_SLOTS; ++s) {
        // (mine == false: a tile this launch leaves to the general one - only the refill below happens.  One code path for both, so
        // that the compiler sees ONE set of registers for the slots: two paths meant copies, and a wait for every load in flight)
        if (mine && s * KMO_NT < nqp && !(KMO_ABL & 4)) {  // block-uniform
#if KMO_TAB_AHEAD
            const bool valid = vN;
            const float4 c0 = cN, r0 = rN;
            if ((s + 1) * KMO_NT < nqp) {
                vN = (s + 1) * KMO_NT + tid < nqp;
                cN = s_u4[vN ? qj : 0]; rN = s_v4[vN ? qi : 0];
                kmt_advance(qi, qj, di, dj, bw);
            }
#else
            const bool valid = s * KMO_NT + tid < nqp;
            const int vqi = valid ? qi : 0, vqj = valid ? qj : 0;
            const float4 c0 = s_u4[vqj], r0 = s_v4[vqi];
#endif
            KmtPix q;
            float gdx, gdy;
            kmt_pix_position_pad<CM, ALIGN, FAST, PADX>(m, kmt_half(c0), kmt_half(r0), valid, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, a.g.W, a.g.H, q, gdx, gdy);
            kmo_pix<CM, CC, FAST, FIXED>(q, G[s], s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
#if !KMO_TAB_AHEAD
            kmt_advance(qi, qj, di, dj, bw);
#endif
        }
#endif
        kmo_request_slot<T, CC>(gout_n, w, s, wn, G[s]);
        // the next tile's source tile is requested here, not before the loop: registers that are live across the whole loop get
        // moved by the register allocator at its entry, and a move of a register with a load in flight is a wait for that load
        if (s == KMO_SRC_AT && nxt.t >= 0 && nxt.regular && nxt.p == 0) {
            if constexpr (KMO_SRC_DMA && sizeof(T) == 4) kmo_dma_src<T, CC>(a, nxt, s_src_next);
            else kmo_issue_src<T, CC>(a, nxt, S);
        }
        if ((s + 1) % KMO_FENCE_EVERY == 0) KM_SCHED_FENCE();  // one pixel at a time: interleaving the slots costs more registers than it hides latency
    }
}

// column / row tables of one band of a box: (m0 u, m3 u, m6 u, u) per column, (m1 v, m4 v, m7 v, v) per row
template <int CM>
__device__ __forceinline__ bool kmo_fill_tables(const KmWarpGeom<float>& g, const float (&m)[9], int jb, int bwb, int ib, int nrows, float4* s_u4, float4* s_v4) {
    const int tid = threadIdx.x;
    static_assert(KMO_NT >= KMT_BAND_W + KMT_TAB, "the column and row tables are filled by disjoint threads");
    if (tid < bwb) {
        const float u = km_base_x<float, CM>(g, jb + tid);
        const KmlHalf h = kml_col_half<CM>(m, u);
        s_u4[tid] = make_float4(h.a, h.b, h.c, u);
    }
    bool okr = true;
    const int rt = tid - (KMO_NT - KMT_TAB);  // the row table is filled by the LAST threads: the first ones fill the columns
    if (rt >= 0 && rt < nrows) {
        const float v = km_base_y<float, CM>(g, ib + rt);
        const KmlHalf h = kml_row_half<CM>(m, v);
        s_v4[rt] = make_float4(h.a, h.b, h.c, v);
        okr = kml_row_guard<CM>(g, m, v);
    }
    return okr;
}

// k with |w g 2^k| (taps per pixel) < 2^30 for |g| <= M (M finite)
__device__ __forceinline__ int kmo_scale_exp(float M, int hb) {
    int kexp = 0;
    if (M > 0.f) {
        int ex2;
        (void)frexpf(M, &ex2);  // M = f * 2^ex2, f in [0.5, 1)  =>  M < 2^ex2
        kexp = max(-126, min(126, 30 - hb - ex2));
    }
    return kmt_uniform(kexp);
}
__device__ __forceinline__ void kmo_scale(float M, int hb, float& scale, float& inv_scale) {
    int kexp = 0;
    if (M > 0.f) {
        int ex2;
        (void)frexpf(M, &ex2);  // M = f * 2^ex2, f in [0.5, 1)  =>  M < 2^ex2
        kexp = 30 - hb - ex2;
        kexp = max(-126, min(126, kexp));
    }
    scale = kmt_uniform(ldexpf(1.0f, kexp));
    inv_scale = kmt_uniform(ldexpf(1.0f, -kexp));
}

// The general path of one tile (launch 3): exact maximum over the box, bands of the box through the tables, one pixel per thread and
// pass; IEEE float accumulation when fixed point is not accurate enough or the gradients are not finite.
template <typename T, int CM, int ALIGN, int CC, int PADX>
__device__ __forceinline__ void kmo_general_tile(const KmWarpFusedArgs<T>& a, const float (&m)[9], const KmoTile& d, const KmoConsts& k, float4* s_u4, float4* s_v4,
                                                 int* s_acc, const float* s_src, uint32_t* s_red, float (&A)[9], float& inv_scale, bool& finite) {
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t dst_plane = (size_t)g.h * g.w;
    const T* gout_c[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) gout_c[c] = a.gout + ((size_t)d.plane0 + (size_t)c) * dst_plane;
    const int bw = d.j1 - d.j0 + 1, bh = d.i1 - d.i0 + 1;
    const bool empty = (bw <= 0 || bh <= 0);
    finite = false;
    float scale = 1.f;
    inv_scale = 1.f;
    if (empty) { finite = true; return; }
    if (d.fixed_ok) {
        // the exact maximum of |grad_out| over the box (as integers: sign cleared, NaN / inf sort above every finite value)
        uint32_t mb = 0;
        const long long nq = (long long)bw * bh;
        for (long long base = 0; base < nq; base += KMO_NT) {
            const long long e = min(base + tid, nq - 1);  // clamped: duplicates do not change a max
            const int qi = (int)(e / bw), qj = (int)(e - (long long)qi * bw);
            const uint32_t off = (uint32_t)(d.i0 + qi) * (uint32_t)g.w + (uint32_t)(d.j0 + qj);
#pragma unroll
            for (int c = 0; c < CC; ++c) mb = max(mb, __float_as_uint((float)km_ld(km_at(gout_c[c], off))) & 0x7fffffffu);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mb = max(mb, (uint32_t)__shfl_down((int)mb, off, 64));
        if (lane == 0) s_red[wave] = mb;
        __syncthreads();
        uint32_t Mb = s_red[0];
#pragma unroll
        for (int w = 1; w < KMO_NW; ++w) Mb = max(Mb, s_red[w]);
        Mb = (uint32_t)kmt_uniform((int)Mb);
        finite = Mb < 0x7f800000u;
        if (finite) kmo_scale(__uint_as_float(Mb), d.hb, scale, inv_scale);
    }
    bool first_band = true;
    for (int jb = d.j0; jb <= d.j1; jb += KMT_BAND_W) {
        const int bwb = min(KMT_BAND_W, d.j1 - jb + 1);
        for (int ib = d.i0; ib <= d.i1; ib += KMT_TAB) {
            const int nrows = min(d.i1, ib + KMT_TAB - 1) - ib + 1;
            if (!first_band) __syncthreads();  // the previous band's readers are done with the tables
            first_band = false;
            const bool okr = kmo_fill_tables<CM>(g, m, jb, bwb, ib, nrows, s_u4, s_v4);
            const bool fast = __syncthreads_and((int)okr) != 0;  // every row of the band has safe division operands
            const int nq = bwb * nrows;
            const int di = kmt_uniform(KMO_NT / bwb), dj = kmt_uniform(KMO_NT % bwb);
            int qi, qj;
            kmo_first(tid, bwb, qi, qj);
            const uint32_t row0 = (uint32_t)ib * (uint32_t)g.w + (uint32_t)jb;
            int base = 0;
            if (finite && fast) {
                // the common case of this launch (a box that did not fit the persistent loop's registers: rotation, minification): 4 pixels
                // per thread and pass, their loads issued together - a pass costs one memory latency, not four
                constexpr int GS = 4;
                for (; base + GS * KMO_NT <= nq; base += GS * KMO_NT) {
                    float go4[GS][CC];
                    int pqi[GS], pqj[GS];
#pragma unroll
                    for (int s = 0; s < GS; ++s) {
                        pqi[s] = qi; pqj[s] = qj;
                        kmt_load_go<T, CC>(gout_c, row0 + (uint32_t)qi * (uint32_t)g.w + (uint32_t)qj, go4[s]);
                        kmt_advance(qi, qj, di, dj, bwb);
                    }
#pragma unroll
                    for (int s = 0; s < GS; ++s) {
                        const float4 c0 = s_u4[pqj[s]], r0 = s_v4[pqi[s]];
                        KmtPix q;
                        float gdx, gdy;
                        kmt_pix_position_pad<CM, ALIGN, true, PADX>(m, kmt_half(c0), kmt_half(r0), true, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, g.W, g.H, q, gdx, gdy);
                        kmo_pix<CM, CC, true, true>(q, go4[s], s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
                    }
                }
            }
            for (; base < nq; base += KMO_NT) {
                const bool valid = base + tid < nq;
                const int vqi = valid ? qi : 0, vqj = valid ? qj : 0;
                float go[CC];
                kmt_load_go<T, CC>(gout_c, row0 + (uint32_t)vqi * (uint32_t)g.w + (uint32_t)vqj, go);
                const float4 c0 = s_u4[vqj], r0 = s_v4[vqi];
                KmtPix q;
                if (!finite) {
                    float gdx, gdy;
                    kmt_pix_position_pad<CM, ALIGN, false, PADX>(m, kmt_half(c0), kmt_half(r0), valid, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, g.W, g.H, q, gdx, gdy);
                    kmo_pix<CM, CC, false, false>(q, go, s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
                } else if (fast) {
                    float gdx, gdy;
                    kmt_pix_position_pad<CM, ALIGN, true, PADX>(m, kmt_half(c0), kmt_half(r0), valid, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, g.W, g.H, q, gdx, gdy);
                    kmo_pix<CM, CC, true, true>(q, go, s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
                } else {
                    float gdx, gdy;
                    kmt_pix_position_pad<CM, ALIGN, false, PADX>(m, kmt_half(c0), kmt_half(r0), valid, k.Wm1, k.hW, k.Hm1, k.hH, (uint32_t)d.X0, (uint32_t)d.Y0, g.W, g.H, q, gdx, gdy);
                    kmo_pix<CM, CC, false, true>(q, go, s_acc, s_src, scale, (uint32_t)d.TWc, (uint32_t)d.THc, c0.w, r0.w, PADX ? k.mx * gdx : k.mx, PADX ? k.my * gdy : k.my, A);
                }
                kmt_advance(qi, qj, di, dj, bwb);
            }
        }
    }
}

// convert and write the tile, leaving the accumulators zeroed for the next one
template <typename T, int CC>
__device__ __forceinline__ void kmo_flush(const KmWarpFusedArgs<T>& a, const KmoTile& d, int* s_acc, bool finite, float inv_scale, bool discard = false) {
    if (KMO_ABL & 8) return;
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x;
    const size_t src_plane = (size_t)g.H * g.W;
    float* gsrc_b = a.gsrc + (size_t)d.plane0 * src_plane;
#if KMO_FLUSH_STRAIGHT
    if (d.fvec && d.THc == KMO_TH && finite && KMO_TH * KMT_TW / 4 == KMO_NT) {
        // a full tile: one 16-byte piece per thread and channel, the reads of all channels in flight together (- 1 % on the kernel:
        // profiles/r04/bwd_fused_latency_variants.txt, runs 34 and 38)
        int4 q[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) q[c] = reinterpret_cast<const int4*>(s_acc + c * KMO_PLANE)[tid];
#pragma unroll
        for (int c = 0; c < CC; ++c) reinterpret_cast<int4*>(s_acc + c * KMO_PLANE)[tid] = make_int4(0, 0, 0, 0);
        if (discard) return;
        float* outp = gsrc_b + (size_t)(d.Y0 + (tid >> 4)) * g.W + (d.X0 + (tid & 15) * 4);
#pragma unroll
        for (int c = 0; c < CC; ++c, outp += src_plane) {
            KM_CHECK_ALIGNED(outp, 16);
            const float4 v = make_float4((float)q[c].x * inv_scale, (float)q[c].y * inv_scale, (float)q[c].z * inv_scale, (float)q[c].w * inv_scale);
            if (a.stream_out) {
                typedef float km_f4v __attribute__((ext_vector_type(4)));
                km_f4v vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
                __builtin_nontemporal_store(vv, reinterpret_cast<km_f4v*>(outp));
            } else {
                *reinterpret_cast<float4*>(outp) = v;
            }
        }
        return;
    }
#endif
    if (d.fvec) {
        // full-width tile, 16-byte aligned rows: 16 lanes x 16 bytes cover a tile row, a wave writes 4 rows per store
        const int col4 = (tid & 15) * 4, row0 = tid >> 4;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            for (int r = row0; r < d.THc; r += KMO_NT / 16) {
                float* outp = gsrc_b + (size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (d.X0 + col4);
                int* accp = s_acc + c * KMO_PLANE + r * KMT_TW + col4;
                KM_CHECK_ALIGNED(accp, 16);
                KM_CHECK_ALIGNED(outp, 16);
                const int4 q = *reinterpret_cast<const int4*>(accp);
                *reinterpret_cast<int4*>(accp) = make_int4(0, 0, 0, 0);
                float4 v;
                if (finite) v = make_float4((float)q.x * inv_scale, (float)q.y * inv_scale, (float)q.z * inv_scale, (float)q.w * inv_scale);
                else v = make_float4(__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w));
                if (discard) {
                    // (a tile that met a non-finite gradient after its first pass: the accumulators are zeroed, the general launch writes it)
                } else if (a.stream_out) {
                    typedef float km_f4v __attribute__((ext_vector_type(4)));
                    km_f4v vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
                    __builtin_nontemporal_store(vv, reinterpret_cast<km_f4v*>(outp));
                } else {
                    *reinterpret_cast<float4*>(outp) = v;
                }
            }
        }
    } else {
        // ragged right edge / unaligned rows: lane -> column, waves -> rows, one float per store
        const int col = tid & (KMT_TW - 1), row0 = tid >> 6;
        if (col < d.TWc) {
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                for (int r = row0; r < d.THc; r += KMO_NW) {
                    int* accp = s_acc + c * KMO_PLANE + r * KMT_TW + col;
                    const int q = *accp;
                    *accp = 0;
                    if (!discard) gsrc_b[(size_t)c * src_plane + (size_t)(d.Y0 + r) * g.W + (d.X0 + col)] = finite ? (float)q * inv_scale : __int_as_float(q);
                }
            }
        }
    }
}

struct KmoLds {
    float4 *s_u4, *s_v4;
    int* s_acc;
    float* s_src;
    float* s_src2;   // KMO_SRC_DMA: the second source buffer (tile q uses buffer q & 1)
    int* s_box;
    uint32_t* s_red;
    double* s_gm;
};
template <int CC>
__device__ __forceinline__ KmoLds kmo_carve(char* smem_raw) {
    KmoLds l;
    l.s_u4 = (float4*)smem_raw;                             // [KMT_BAND_W] per column of the box
    l.s_v4 = l.s_u4 + KMT_BAND_W;                           // [KMT_TAB] per row of the box
    l.s_acc = (int*)(l.s_v4 + KMT_TAB);                     // [CC][TH][TW] fixed-point accumulators
    l.s_src = (float*)(l.s_acc + CC * KMO_PLANE);           // [CC][TH][TW] the source tile (minus fill)
    l.s_src2 = l.s_src + CC * KMO_PLANE;                    // [CC][TH][TW] (KMO_SRC_DMA only)
    l.s_box = (int*)(l.s_src + (KMO_SRC_DMA ? 2 : 1) * CC * KMO_PLANE);  // [2 KMO_RUN][KMO_BOX_INTS] ring of records of this worker's tiles
    l.s_red = (uint32_t*)(l.s_box + 2 * KMO_RUN * KMO_RING_INTS);  // [KMO_NW] (8-byte slots: keeps s_gm aligned)
    l.s_gm = (double*)(l.s_red + 2 * KMO_NW);               // [KMO_NW][9] matrix-gradient partials of a finished image
    return l;
}
template <int ALIGN>
__device__ __forceinline__ KmoConsts kmo_consts(const KmWarpGeom<float>& g) {
    KmoConsts kc;
    kc.Wm1 = (float)(g.W - 1); kc.Hm1 = (float)(g.H - 1); kc.hW = (float)g.W / 2; kc.hH = (float)g.H / 2;
    kc.mx = ALIGN ? kc.Wm1 / 2 : kc.hW; kc.my = ALIGN ? kc.Hm1 / 2 : kc.hH;  // d (pixel) / d (normalised): km_unnormalize's multiplier
    return kc;
}
// the wave partials of a finished image in s_gm -> 9 fp64 atomics
template <int CM, bool TOTALS = false>
__device__ __forceinline__ void kmo_gm_commit(const double* s_gm, double* gmat_b, int tid) {
    if (tid < 9 && gmat_b) {  // (gmat_b == nullptr: the caller wants the image gradient only)
        double s = 0.0;
        if (TOTALS) {
            s = s_gm[tid];  // (the persistent loop's image-end reduction leaves the nine totals)
        } else {
#pragma unroll
            for (int w = 0; w < KMO_NW; ++w) s += s_gm[w * 9 + tid];
        }
        if (s != 0.0 && !(CM == KM_COORD_AFFINE && tid >= 6)) km_atomic_add(gmat_b + tid, s);
    }
}

// A tile whose requests have arrived: source tile -> LDS, exact maximum of |grad_out| over its box, coordinate tables
template <typename T, int CM, int CC>
__device__ __forceinline__ void kmo_stage(const KmWarpFusedArgs<T>& a, const KmoTile& d, const int* rec, const float (&G)[KMO_SLOTS][CC],
                                          const float (&S)[CC * KMO_PLANE / KMO_NT], const KmoLds& l, float* s_src_cur, const float (&fillv)[CC], bool is_fill, int lane, int wave, uint32_t par KMO_PROF_PARAMS) {
    // (no early exit for a tile this launch does not own: every path through here must CONSUME the slots and the source registers,
    // or the compiler - which cannot know that nothing was requested for such a tile - waits for them later, inside the scatter, with
    // a wait that also covers the next tile's requests)
    if constexpr (KMO_SRC_DMA && sizeof(T) == 4) {
        // this wave's LDS-DMA requests (issued during the previous item's scatter, in front of most of this item's grad_out requests) have
        // landed: the compiler does not know them - its own waits cover them only when a later load of its own is waited for
        KM_VMCNT0();
        if (d.p == 0 && is_fill && d.t >= 0 && d.regular) kmo_fill_in_lds<CC>(d, s_src_cur, fillv);
    } else {
        if (d.p == 0) kmo_store_src<CC>(d, S, s_src_cur, fillv, is_fill);  // (later passes of a box: the tile is in LDS already, no request was made)
    }
    KMO_T(8)  // stage: source tile -> LDS
    uint32_t mb = 0;
#pragma unroll
    for (int s = 0; s < KMO_SLOTS; ++s)
#pragma unroll
        for (int c = 0; c < CC; ++c) mb = max(mb, __float_as_uint(G[s][c]) & 0x7fffffffu);
#if KMO_ABL & 128
    { int keep = (int)mb; KM_OPAQUE(keep); }  // (the slots are still consumed here: the compiler's waits for them stay where they are)
    if (lane == 63 && wave == 0) l.s_red[par] = 0x3f800000u;
#elif KMO_RED_ATOMIC
    // one word per work item (two, alternating): sixteen partials that every wave reads back become one atomic per wave here and one read
    // after the barrier; the wave's maximum through DPP instead of six LDS round trips.  (All 64 lanes ATOMICALLY on the one word: + 12 %
    // on the whole kernel - same-address lanes serialise; profiles/r04/bwd_fused_per_pixel_variants.txt, run 28.)
    (void)wave;
    mb = km_wave_umax_last(mb);
    if (lane == 63) atomicMax(l.s_red + par, mb);
#else
    (void)par;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mb = max(mb, (uint32_t)__shfl_down((int)mb, off, 64));
    if (lane == 0) l.s_red[wave] = mb;
#endif
    KMO_T(9)  // stage: maximum of |grad_out|
    if (d.t >= 0 && d.regular && d.nq > 0 && d.p == 0) {
        float m[9];
        kmo_matrix(rec, m);
        (void)kmo_fill_tables<CM>(a.g, m, d.j0, d.bw, d.i0, d.i1 - d.i0 + 1, l.s_u4, l.s_v4);
    }
    KMO_T(10)  // stage: coordinate tables
}

// ---- launch 2: the persistent loop over the regular tiles -------------------------------------------------------------------------
template <typename T, int CM, int ALIGN, int CC, int PADX>
__global__ __launch_bounds__(KMO_NT, KMO_WG_PER_CU * KMO_NT / 256) void km_warp_bwd_fused_kernel(const KmWarpFusedArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KmoLds l = kmo_carve<CC>(smem_raw);
    const KmoConsts kc = kmo_consts<ALIGN>(g);
    const bool is_fill = (g.pad == KM_PAD_FILL);
    float fillv[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) fillv[c] = is_fill ? a.fill[c] : 0.f;
    const size_t dst_plane = (size_t)g.h * g.w;
    constexpr bool DMA = KMO_SRC_DMA && sizeof(T) == 4;  // the source tiles by LDS-DMA, double-buffered (kmo_dma_src)

    // ---- prologue: records of the first tiles, accumulators zeroed, the first tile requested and staged ----
    if (wave == 0) kmo_fetch_boxes(a, 0u, lane, l.s_box);
    for (int e = tid; e < CC * KMO_PLANE / 4; e += KMO_NT) ((int4*)l.s_acc)[e] = make_int4(0, 0, 0, 0);
    if (KMO_RED_ATOMIC && tid < 2) l.s_red[tid] = 0u;
    __syncthreads();
    KmoTile cur;
#if KMO_RING_DESC
    kmo_load_desc(kmo_ring(l.s_box, 0u), cur);
#else
    kmo_describe(a, kmo_tile_of(a, 0u), kmo_ring(l.s_box, 0u), cur);
#endif
    float G[KMO_SLOTS][CC];             // grad_out of this thread's pixels: the current tile's, refilled slot by slot with the next tile's
    float S[CC * KMO_PLANE / KMO_NT];   // this thread's part of the source tile on its way to LDS
#pragma unroll
    for (int k = 0; k < CC * KMO_PLANE / KMO_NT; ++k) S[k] = 0.f;
    {
        const T* gout_c[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) gout_c[c] = a.gout;
        KmoWalk w0;
        kmo_walk_init(a, cur, w0);
        if (cur.t >= 0 && cur.regular) {
#pragma unroll
            for (int c = 0; c < CC; ++c) gout_c[c] = a.gout + ((size_t)cur.plane0 + (size_t)c) * dst_plane;
            if constexpr (DMA) kmo_dma_src<T, CC>(a, cur, l.s_src);  // (tile q uses source buffer q & 1)
            else kmo_issue_src<T, CC>(a, cur, S);
        } else {
            kmo_walk_none(w0);
        }
#pragma unroll
        for (int s = 0; s < KMO_SLOTS; ++s) kmo_request_slot<T, CC>(gout_c, g.w, s, w0, G[s]);
    }
    float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int pending_b = -1;      // row of gmat whose matrix-gradient partials sit in s_gm
    KmoTile prev = KmoTile{};  // the tile whose accumulators wait to be flushed (prev_discard: zeroed without being written)
    prev.t = -1;
    float prev_inv_scale = 1.f;
    bool prev_discard = false;
    // state of the tile across the passes of its box
    int kexp = 0;              // the accumulators hold contributions times 2^kexp
    uint32_t bound_bits = 0;   // ... chosen for |grad_out| <= this (bit pattern)
    bool tile_ok = true;       // no non-finite gradient met so far

#ifdef KMO_PROFILE
    unsigned long long prof[KMO_PROF_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // A work item is (tile, pass): a box larger than the registers of the workgroup (KMO_CAP pixels: rotations beyond ~10 degrees,
    // magnification) is walked in several passes, the next pass requested slot by slot during the current one like a next tile.
    for (uint32_t q = 0, item = 0; cur.t >= 0; ++item) {
        // ---- this item's requests have arrived: (first pass: source tile -> LDS, coordinate tables;) exact maximum of |grad_out| over the
        //      pass - consumed BEFORE the flush of the previous tile issues its stores (a wait for loads that has stores behind it in the
        //      queue waits for those too)
        KMO_T(7)  // (loop tail of the previous item)
#ifdef KMO_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KMO_T(0)  // wait for this item's requests
        float* const s_src_cur = (DMA && (q & 1u)) ? l.s_src2 : l.s_src;
        float* const s_src_nxt = (DMA && !(q & 1u)) ? l.s_src2 : l.s_src;  // (of tile q + 1)
        kmo_stage<T, CM, CC>(a, cur, kmo_ring(l.s_box, q), G, S, l, s_src_cur, fillv, is_fill, lane, wave, item & 1u KMO_PROF_ARGS);
        KMO_T(1)  // stage: source tile -> LDS, maxima, tables
        // ---- flush of the previous tile (zeroes the accumulators) ----
        if (prev.t >= 0) kmo_flush<T, CC>(a, prev, l.s_acc, true, prev_inv_scale, prev_discard);
        prev.t = -1;
        KMO_T(2)  // flush
        const bool last_pass = cur.p + 1 >= cur.npass;
        if (wave == 0 && last_pass && (q + 1u) % KMO_RUN == 0u) kmo_fetch_boxes(a, q + 1u, lane, l.s_box);  // (into the half of the ring tile q is not in)
        if (!(KMO_ABL & 32)) KM_LDS_BARRIER();  // B1: source tile, maxima, tables in LDS, accumulators zero (first pass)
        KMO_T(3)  // barrier B1

        // ---- the NEXT item: the next pass of this box, or the first pass of the next tile (whose source tile is requested after the
        //      first slot of the scatter); its grad_out slot by slot during the scatter ----
        KmoTile nxt;
        if (last_pass) {
#if KMO_RING_DESC
            kmo_load_desc(kmo_ring(l.s_box, q + 1u), nxt);
#else
            kmo_describe(a, kmo_tile_of(a, q + 1u), kmo_ring(l.s_box, q + 1u), nxt);
#endif
        } else {
            nxt = cur;
            nxt.p = cur.p + 1;
        }
        KMO_T(11)  // description: the next tile's fields
        const T* gout_n[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) gout_n[c] = a.gout;
        KmoWalk wn;
        kmo_walk_init(a, nxt, wn);
        if (nxt.t >= 0 && nxt.regular) {
#pragma unroll
            for (int c = 0; c < CC; ++c) gout_n[c] = a.gout + ((size_t)nxt.plane0 + (size_t)c) * dst_plane;
        } else {
            kmo_walk_none(wn);  // (the end of the sequence, or a tile of the general launch: nothing to request)
        }
        KMO_T(12)  // description: walk of the next box, plane pointers
        // matrix-gradient partials of the image finished before this tile
        if (pending_b >= 0) kmo_gm_commit<CM, (KMO_GM_LDS && CC * KMO_PLANE >= 9 * KMO_NT)>(l.s_gm, a.gmat ? a.gmat + (size_t)pending_b * 9 : nullptr, tid);
        pending_b = -1;
        KMO_T(13)  // description: matrix-gradient commit

        // ---- the fixed-point scale: from the exact maximum of the first pass; a later pass with a larger one rescales the accumulators ----
        if (cur.p == 0) tile_ok = cur.regular;
        if (cur.regular && tile_ok) {
#if KMO_RED_ATOMIC
            uint32_t Mb = l.s_red[item & 1u];
#else
            uint32_t Mb = l.s_red[0];
#pragma unroll
            for (int w = 1; w < KMO_NW; ++w) Mb = max(Mb, l.s_red[w]);
#endif
            Mb = (uint32_t)kmt_uniform((int)Mb);
            if (Mb >= 0x7f800000u) {
                // NaN / inf in the box: IEEE float accumulation is the general launch's; mark the tile and leave it alone (what earlier
                // passes added is discarded with the accumulators; their share of the matrix gradient stays - that image's matrix
                // gradient is not finite either way)
                tile_ok = false;
                if (tid == 0) a.ws[(size_t)cur.t * KMO_BOX_INTS + 5] = (cur.fixed_ok ? KMO_F_FIXED : 0) | KMO_F_REGULAR | KMO_F_NONFINITE;
            } else if (cur.p == 0 || bound_bits == 0u) {
                // (as long as every gradient met was zero the accumulators are zero and the scale is still free: a first pass over zeros
                // must not pin it at 2^0 for the small gradients of a later pass)
                bound_bits = Mb;
                kexp = kmo_scale_exp(__uint_as_float(Mb), cur.hb);
            } else if (Mb > bound_bits) {
                const int knew = kmo_scale_exp(__uint_as_float(Mb), cur.hb);
                bound_bits = Mb;
                if (knew < kexp) {  // (block-uniform) a coarser scale: shift what has been accumulated, rounding to nearest
                    const int sh = min(kexp - knew, 31);
                    for (int e = tid; e < CC * KMO_PLANE; e += KMO_NT) {
                        const int v = l.s_acc[e];
                        l.s_acc[e] = sh >= 31 ? 0 : (int)(((long long)v + (1ll << (sh - 1))) >> sh);
                    }
                    kexp = knew;
                    KM_LDS_BARRIER();
                }
            }
        }
        const bool mine = cur.regular && tile_ok;  // this launch scatters this item
        const float scale = kmt_uniform(ldexpf(1.0f, kexp)), inv_scale = kmt_uniform(ldexpf(1.0f, -kexp));
        KMO_T(4)  // next item's description, matrix-gradient commit, scale
        {
            float m[9];
            kmo_matrix(kmo_ring(l.s_box, q), m);
            kmo_process<T, CM, ALIGN, CC, true, true, PADX>(m, cur, kc, G, l.s_u4, l.s_v4, l.s_acc, s_src_cur, scale, A, g.w, wn, gout_n, mine, a, nxt, S, s_src_nxt);
        }
        KMO_T(5)  // scatter
        if (!(KMO_ABL & 64)) KM_LDS_BARRIER();  // B2: every contribution of the pass is in the accumulators; the tables, s_red and (last pass) the source tile are free
        KMO_T(6)  // barrier B2
        // (this item's maximum has been read by every wave; the word's next writers - the item after the next - are behind the next B1)
        if (KMO_RED_ATOMIC && tid == 0) l.s_red[item & 1u] = 0u;

        if (last_pass) {
            // ---- an image that ends here publishes its matrix-gradient partials ----
            if ((KMO_ABL & 16) && (nxt.t < 0 || nxt.b != cur.b)) {  // (ablation: the partials are consumed - their arithmetic stays - and dropped)
#pragma unroll
                for (int k = 0; k < 9; ++k) { KM_OPAQUE(A[k]); A[k] = 0.f; }
            }
            if (!(KMO_ABL & 16) && (nxt.t < 0 || nxt.b != cur.b)) {  // (per (image, group): the groups of an image meet in gmat's atomics)
                if (KMO_GM_LDS && CC * KMO_PLANE >= 9 * KMO_NT) {
                    // Nine 64-lane fp64 reductions per wave are 108 ds_bpermute + 54 v_add_f64 for each of the 16 waves - 2 000 cycles per tile on
                    // average (profiles/r03_bwd_phases.txt: "loop tail").  The source tile is dead after the last pass of its tile: the 1024
                    // threads park their nine fp32 partials there, and wave k sums row k - sixteen conflict-free reads per lane, fp64 adds, ONE
                    // wave reduction.  The same fp32 partials summed in fp64, in another order; two more LDS barriers per image.
                    float* s_part = s_src_cur;  // (KMO_SRC_DMA: the other buffer is receiving the next tile)
#pragma unroll
                    for (int k = 0; k < 9; ++k) { s_part[k * KMO_NT + tid] = A[k]; A[k] = 0.f; }
                    KM_LDS_BARRIER();
                    if (wave < 9) {
                        double s = 0.0;
#pragma unroll
                        for (int jj = 0; jj < KMO_NT / 64; ++jj) s += (double)s_part[wave * KMO_NT + jj * 64 + lane];
                        s = km_wave_sum(s);
                        if (lane == 0) l.s_gm[wave] = s;
                    }
                    KM_LDS_BARRIER();  // (the next tile's stage rewrites the source tile)
                } else {
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const double s = km_wave_sum((double)A[k]);
                        if (lane == 0) l.s_gm[wave * 9 + k] = s;
                        A[k] = 0.f;
                    }
                }
                pending_b = cur.bm;
            }
            if (cur.regular) {  // flushed (or, after a non-finite gradient, only zeroed) after the next item's stage
                prev = cur;
                prev_inv_scale = inv_scale;
                prev_discard = !tile_ok;
            }
            ++q;
        }
        cur = nxt;
    }
    if (prev.t >= 0) kmo_flush<T, CC>(a, prev, l.s_acc, true, prev_inv_scale, prev_discard);
#ifdef KMO_PROFILE
    if (lane == 0 && blockIdx.x < 512u) {
#pragma unroll
        for (int k = 0; k < KMO_PROF_PHASES; ++k) kmo_prof_out[((size_t)blockIdx.x * 16 + (wave & 15)) * KMO_PROF_PHASES + k] = prof[k];
        if (wave == 0) {
            unsigned xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            kmo_prof_rt[blockIdx.x * 4 + 0] = rt0;
            kmo_prof_rt[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            kmo_prof_rt[blockIdx.x * 4 + 2] = xcc;
        }
    }
#endif
    // ---- epilogue: the last image's partials ----
    __syncthreads();
    if (pending_b >= 0) kmo_gm_commit<CM, (KMO_GM_LDS && CC * KMO_PLANE >= 9 * KMO_NT)>(l.s_gm, a.gmat ? a.gmat + (size_t)pending_b * 9 : nullptr, tid);
}

// ---- launch 3, first part: non-finite gradients at output pixels NO tile visits ---------------------------------------------------
// The tiles' boxes cover every output pixel whose footprint touches the source image; a pixel that samples entirely outside it is read by
// nobody.  In the reference such a pixel still multiplies grad_out into the matrix gradient - 0 * inf, 0 * NaN (ATen's backward takes
// the out-of-bounds taps as zeros and multiplies) - so an image whose grad_out holds an inf or NaN ANYWHERE has a non-finite matrix
// gradient (every pixel's term is a finite factor times its grad_out).  Visited pixels produce theirs through the IEEE path of this
// launch; this scan finds the rest: 16 x 16 blocks of the output are classified by their corners (a projective map with a denominator of
// one sign sends the block to a convex quad), blocks that may hold an unvisited pixel are tested pixel by pixel with the forward's own
// position and a margin (testing a visited pixel as well changes nothing: its gradient already made the result non-finite), and an image
// with a hit gets NaN added to its accumulators.
// NON-FINITE POSITIONS (round 6).  A pixel whose sampling position is NaN / +-inf (a singular or NaN-carrying matrix, a projective
// denominator of exactly zero) is in no tile's box either, and in the reference its weights are NaN: the zeros gathered for its taps times
// those weights put NaN into the grid gradient whatever grad_out holds there (tests/golden/nonfinite_coords.npz).  The scan evaluates the
// reference's own terms for such a pixel - gix is NaN iff the y weights are, giy iff the x weights are - through km_gm_terms, and adds NaN to
// exactly the rows of the matrix gradient those terms reach (homography mode with a NaN in H[0, :]: rows 1 and 2, not row 0).
// WHICH REFERENCE.  This is ATen's CPU kernel (the oracle, tests/golden/nonfinite_outside.npz).  ATen's CUDA / HIP grid_sampler backward - what the
// reference runs on an accelerator - skips out-of-bounds taps instead, and such a gradient leaves the matrix gradient finite.  The launch policy
// `warp_bwd_no_scan` (KM_WARP_BWD_SCAN=0) selects that behaviour and saves this scan (~20 us of a 0.55 ms backward at config 2: it reads one
// 128-byte line per row, side and channel along the image's borders); the default is the CPU reference's, which is what parity is pinned to.
struct KmoScanMap {
    float au, bu, av, bv;  // base coordinates: u = au * j + bu, v = av * i + bv
    float sx, ox, sy, oy;  // source pixel = s * normalised + o
};
template <int CM, int ALIGN>
__device__ __forceinline__ KmoScanMap kmo_scan_map(const KmWarpGeom<float>& g) {
    KmoScanMap k;
    if (CM == KM_COORD_AFFINE) { k.au = g.lin_step_x; k.bu = g.lin_lo_x; k.av = g.lin_step_y; k.bv = g.lin_lo_y; }
    else if (CM == KM_COORD_HOMOGRAPHY && !g.norm_coords) { k.au = 1.f; k.bu = 0.f; k.av = 1.f; k.bv = 0.f; }
    else { k.au = g.w > 1 ? 2.f / (float)(g.w - 1) : 0.f; k.bu = -1.f; k.av = g.h > 1 ? 2.f / (float)(g.h - 1) : 0.f; k.bv = -1.f; }
    k.sx = ALIGN ? 0.5f * (float)(g.W - 1) : 0.5f * (float)g.W; k.ox = ALIGN ? k.sx : k.sx - 0.5f;
    k.sy = ALIGN ? 0.5f * (float)(g.H - 1) : 0.5f * (float)g.H; k.oy = ALIGN ? k.sy : k.sy - 0.5f;
    return k;
}
// approximate source position of output pixel (j, i) (a few ulps and one v_rcp away from the forward's) and the sign of its denominator
template <int CM>
__device__ __forceinline__ void kmo_scan_pos(const KmoScanMap& k, const float (&m)[9], float j, float i, float& x, float& y, float& den) {
    const float u = km_fma(k.au, j, k.bu), v = km_fma(k.av, i, k.bv);
    const float nx = km_fma(m[0], u, km_fma(m[1], v, m[2])), ny = km_fma(m[3], u, km_fma(m[4], v, m[5]));
    den = (CM == KM_COORD_AFFINE) ? 1.f : km_fma(m[6], u, km_fma(m[7], v, m[8]));
    const float r = (CM == KM_COORD_AFFINE) ? 1.f : __builtin_amdgcn_rcpf(den);
    x = km_fma(nx * r, k.sx, k.ox);
    y = km_fma(ny * r, k.sy, k.oy);
}
// the forward's own position of output pixel (j, i): km_gen_coord + km_unnormalize, operation for operation (zeros / fill padding)
template <int CM, int ALIGN>
__device__ __forceinline__ void kmo_scan_exact(const KmWarpGeom<float>& g, const float (&m)[9], int j, int i, KmCoord<float>& cd, float& x, float& y) {
    km_gen_coord<float, CM>(m, km_base_x<float, CM>(g, j), km_base_y<float, CM>(g, i), cd);
    float mx, my;
    x = km_unnormalize(cd.gx, g.W, ALIGN, mx);
    y = km_unnormalize(cd.gy, g.H, ALIGN, my);
}
// true unless the pixel is certain to have a tap inside the image (NaN positions: true)
__device__ __forceinline__ bool kmo_scan_maybe_unvisited(float x, float y, float W, float H) {
    const float e = 0.03125f;  // far above the error of the approximate position, far below a pixel
    return !((x > -1.f + e) && (x < W - e) && (y > -1.f + e) && (y < H - e));
}
#define KMO_SCAN_BLK 16
#ifndef KMO_SCAN_FORK
#define KMO_SCAN_FORK 1   // 0: the scan's launch stays on the launch stream, between the boxes and the persistent loop (A/B)
#endif
#ifndef KMO_SCAN_NB
#define KMO_SCAN_NB 2   // candidate blocks a wave scans at a time
#endif
#ifndef KMO_SCAN_CH
#define KMO_SCAN_CH 3   // channels whose loads fly together
#endif
// One workgroup classifies KMO_NT blocks at a time (thread = block: its four corners), compacts the candidates into an LDS list and deals
// them to its 16 waves - the candidates are the blocks along the image's borders, i.e. whole rows of blocks: left with the wave that
// classified them, a few waves walked 30 - 60 blocks each while the rest had none (180 us for the launch; profiles/r04/bwd_general_launch_grid.txt).
template <typename T, int CM, int ALIGN, int NT, int NBLK = KMO_SCAN_NB>  // NT: threads of the workgroup (the candidate list holds NT blocks); NBLK: candidates of a wave in flight
__device__ __forceinline__ void kmo_scan_unvisited(const KmWarpFusedArgs<T>& a, uint32_t* s_list, uint32_t* s_count) {
    constexpr int NWV = NT / 64;
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KmoScanMap k = kmo_scan_map<CM, ALIGN>(g);
    const uint32_t bx = (uint32_t)(g.w + KMO_SCAN_BLK - 1) / KMO_SCAN_BLK, by = (uint32_t)(g.h + KMO_SCAN_BLK - 1) / KMO_SCAN_BLK;
    const uint32_t per_image = bx * by;
    const uint64_t nblk = (uint64_t)per_image * (uint64_t)g.B;
    const float Wf = (float)g.W, Hf = (float)g.H;
    const size_t dst_plane = (size_t)g.h * g.w;
    for (uint64_t base = (uint64_t)blockIdx.x * NT; base < nblk; base += (uint64_t)gridDim.x * NT) {
        if (tid == 0) *s_count = 0u;
        __syncthreads();
        // ---- thread = block: its four corners ----
        const uint64_t e = base + (uint64_t)tid;
        bool todo = false;
        if (e < nblk) {
            const uint32_t b = (uint32_t)(e / per_image);
            const uint32_t r = (uint32_t)(e - (uint64_t)b * per_image);
            const uint32_t ty = r / bx, tx = r - ty * bx;
            float m[9];
            const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0u : b) * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q) m[q] = mp[q];
            const float j0 = (float)(tx * KMO_SCAN_BLK), j1 = (float)min((int)(tx * KMO_SCAN_BLK) + KMO_SCAN_BLK - 1, g.w - 1);
            const float i0 = (float)(ty * KMO_SCAN_BLK), i1 = (float)min((int)(ty * KMO_SCAN_BLK) + KMO_SCAN_BLK - 1, g.h - 1);
            float dmin = 3.0e38f, dmax = -3.0e38f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x, y, d;
                kmo_scan_pos<CM>(k, m, (c & 1) ? j1 : j0, (c & 2) ? i1 : i0, x, y, d);
                todo = todo || kmo_scan_maybe_unvisited(x, y, Wf, Hf);
                dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
            }
            // the corners speak for the block only where the denominator keeps its sign well away from zero
            todo = todo || !((dmin > 0.f && dmin > 1e-3f * dmax) || (dmax < 0.f && dmax < 1e-3f * dmin));
        }
        // ---- compaction: one LDS atomic per wave ----
        const unsigned long long vote = __ballot(todo);
        uint32_t wbase = 0u;
        if (lane == 0 && vote) wbase = atomicAdd(s_count, (uint32_t)__popcll(vote));
        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
        if (todo) s_list[wbase + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull))] = (uint32_t)tid;
        __syncthreads();
        const uint32_t n = *s_count;
        // ---- wave w takes candidates w, w + 16, ... TWO at a time (their matrix loads, then their grad_out loads, fly together):
        //      64 lanes = 16 columns x 4 rows of a block, four steps ----  (KMO_SCAN_NB at a time)
        for (uint32_t it = (uint32_t)wave; it < n; it += NBLK * NWV) {
            constexpr int NB = NBLK, NS = KMO_SCAN_BLK / 4;
            uint32_t bb[NB], tty[NB], ttx[NB];
            float m[NB][9];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t iu = min(it + (uint32_t)u * NWV, n - 1u);  // (the last pair of an odd list repeats its candidate: a max is idempotent)
                const uint64_t eb = base + (uint64_t)s_list[iu];
                bb[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(eb / per_image));
                const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(eb - (uint64_t)bb[u] * per_image));
                tty[u] = rb / bx; ttx[u] = rb - tty[u] * bx;
                const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0u : bb[u]) * 9;  // (wave-uniform address)
#pragma unroll
                for (int q = 0; q < 9; ++q) m[u][q] = mp[q];
            }
            // (per pixel the EXACT forward position - km_gen_coord's arithmetic, not the classification's approximation: whether a position
            // is finite is a property of the forward's own roundings)
            bool need[NB][NS], posbad[NB][NS];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int j = (int)(ttx[u] * KMO_SCAN_BLK) + (lane & 15);
#pragma unroll
                for (int rr = 0; rr < NS; ++rr) {
                    const int i = (int)(tty[u] * KMO_SCAN_BLK) + rr * 4 + (lane >> 4);
                    const bool in_out = (j < g.w) && (i < g.h);
                    KmCoord<float> cd;
                    float x, y;
                    kmo_scan_exact<CM, ALIGN>(g, m[u], in_out ? j : 0, in_out ? i : 0, cd, x, y);
                    posbad[u][rr] = in_out && !(km_finite(x) && km_finite(y));
                    need[u][rr] = in_out && kmo_scan_maybe_unvisited(x, y, Wf, Hf);  // (true for a NaN / inf position)
                }
            }
            uint32_t worst[NB][NS];  // largest exponent field seen at the pixel
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int rr = 0; rr < NS; ++rr) worst[u][rr] = 0u;
            // (the loads of up to KMO_SCAN_CH channels of all NB blocks fly TOGETHER: channel by channel the launch was a chain of memory
            // round trips - matrices, then each channel of each pair - 26 us for 57 MB)
            for (int c0 = 0; c0 < g.C; c0 += KMO_SCAN_CH) {
                T raw[KMO_SCAN_CH][NB][NS];
#pragma unroll
                for (int cc = 0; cc < KMO_SCAN_CH; ++cc) {
                    const int c = min(c0 + cc, g.C - 1);  // (a channel beyond the last repeats it: a max is idempotent)
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const T* gp = a.gout + ((size_t)bb[u] * g.C + (size_t)c) * dst_plane + (size_t)((int)(ttx[u] * KMO_SCAN_BLK) + (lane & 15));
#pragma unroll
                        for (int rr = 0; rr < NS; ++rr) {
                            const int i = (int)(tty[u] * KMO_SCAN_BLK) + rr * 4 + (lane >> 4);
                            raw[cc][u][rr] = *(need[u][rr] ? gp + (size_t)i * g.w : a.gout);
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < KMO_SCAN_CH; ++cc)
#pragma unroll
                    for (int u = 0; u < NB; ++u)
#pragma unroll
                        for (int rr = 0; rr < NS; ++rr) {
                            const uint32_t bits = __float_as_uint((float)km_ld(&raw[cc][u][rr])) & 0x7f800000u;
                            worst[u][rr] = max(worst[u][rr], need[u][rr] ? bits : 0u);
                        }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                // rows of the matrix gradient that become NaN: bit 0 -> entries 0-2 (the terms in ax), bit 1 -> 3-5 (ay), bit 2 -> 6-8 (az)
                bool all_rows = false, any_bad = false;
#pragma unroll
                for (int rr = 0; rr < NS; ++rr) {
                    all_rows = all_rows || (worst[u][rr] == 0x7f800000u && !posbad[u][rr]);  // a finite position, a non-finite gradient: 0 * inf in every term
                    any_bad = any_bad || posbad[u][rr];
                }
                uint32_t rows = __ballot(all_rows) != 0ull ? 7u : 0u;
                if (__ballot(any_bad) != 0ull) {  // (wave-uniform, rare) non-finite POSITIONS: the reference's own terms, taps gathered as zeros
                    bool f0 = false, f1 = false, f2 = false;
                    const int j = (int)(ttx[u] * KMO_SCAN_BLK) + (lane & 15);
#pragma unroll 1
                    for (int rr = 0; rr < NS; ++rr) {
                        if (!posbad[u][rr]) continue;
                        const int i = (int)(tty[u] * KMO_SCAN_BLK) + rr * 4 + (lane >> 4);
                        KmCoord<float> cd;
                        float x, y;
                        kmo_scan_exact<CM, ALIGN>(g, m[u], j, i, cd, x, y);
                        // gix = sum_c go_c ((ne - nw) wy1 + (se - sw) wy0) with every tap a zero: 0 unless the y weights (or go) are not finite
                        const bool gobad = worst[u][rr] == 0x7f800000u;
                        const float qnan = __int_as_float(0x7fc00000);
                        const float gix = (km_finite(y) && !gobad) ? 0.f : qnan, giy = (km_finite(x) && !gobad) ? 0.f : qnan;
                        float ax, ay, az;
                        km_gm_terms<CM>(cd, gix, giy, ax, ay, az);
                        f0 = f0 || !km_finite(ax); f1 = f1 || !km_finite(ay); f2 = f2 || !km_finite(az);
                    }
                    rows |= (__ballot(f0) != 0ull ? 1u : 0u) | (__ballot(f1) != 0ull ? 2u : 0u) | (__ballot(f2) != 0ull ? 4u : 0u);
                }
                if (lane < (CM == KM_COORD_AFFINE ? 6 : 9) && ((rows >> (lane / 3)) & 1u))
                    km_atomic_add(a.gmat + (size_t)(g.B_M == 1 ? 0u : bb[u]) * 9 + lane, (double)__int_as_float(0x7fc00000));
            }
        }
        __syncthreads();  // (the list is rewritten by the next chunk)
    }
}

// The scan as a launch of its own (round 6).  Inside the general launch it ran in that launch's 1 024-thread workgroups, each holding a CU's
// LDS, strictly BEHIND the persistent loop: 24 us of a 0.55 ms backward (profiles/r05/run43_*).  It needs nothing the persistent loop produces -
// the matrices, grad_out, and gmat zeroed by the boxes launch - so it is forked onto a side stream behind the boxes launch and runs BESIDE the
// persistent loop: 256-thread workgroups of at most 64 registers (the persistent workgroup's 16 waves of 108 leave one wave per SIMD and 3 KB
// of LDS on every CU), joined in front of the general launch.  While the stream is being captured into a graph (or there is no side stream) the
// same kernel runs on the launch stream between the boxes and the persistent loop.
#define KMO_SCAN_NT 256
template <typename T, int CM, int ALIGN>
__global__ __launch_bounds__(KMO_SCAN_NT, 8) void km_warp_bwd_scan_kernel(const KmWarpFusedArgs<T> a) {  // (8 waves per SIMD: at most 64 registers)
    __shared__ uint32_t s_list[KMO_SCAN_NT];
    __shared__ uint32_t s_count;
    kmo_scan_unvisited<T, CM, ALIGN, KMO_SCAN_NT, 1>(a, s_list, &s_count);  // (one candidate per wave at a time: 64 registers)
}

// ---- launch 3: the tiles the persistent loop left (class "general", or a non-finite gradient met at run time) ---------------------
// Workgroup w looks at the records of tiles general_tiles * w ... (lane = tile, one ballot; up to 64 of them, fewer for small problems so
// that a batch of a few images whose every tile is marked still spreads over the chip) and walks the marked ones.
template <typename T, int CM, int ALIGN, int CC, int PADX>
__global__ __launch_bounds__(KMO_NT) void km_warp_bwd_general_kernel(const KmWarpFusedArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ unsigned long long s_todo;
    const KmWarpGeom<float>& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KmoLds l = kmo_carve<CC>(smem_raw);
    const KmoConsts kc = kmo_consts<ALIGN>(g);
    const bool is_fill = (g.pad == KM_PAD_FILL);
    float fillv[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) fillv[c] = is_fill ? a.fill[c] : 0.f;
    // (the scan for non-finite gradients / positions at unvisited pixels is a launch of its own since round 6: km_warp_bwd_scan_kernel)
    // A workgroup of this launch holds a CU's LDS: the grid is one workgroup per persistent worker, each walking its share of the tile
    // groups (a grid of one workgroup per group - 1024 at config 2, four rounds of dispatch with 114 KB of LDS each - cost 45 us per
    // round on some boxes, whatever the workgroups then did: profiles/r04/bwd_general_launch_grid.txt)
    bool zeroed = false;
    const uint32_t ngroups = (a.ntiles + a.general_tiles - 1u) / a.general_tiles;
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const uint32_t t0 = grp * a.general_tiles;
    __syncthreads();  // (s_todo of the previous group has been read by everyone)
    if (wave == 0) {
        const uint32_t t = t0 + (uint32_t)lane;
        int fl = KMO_F_REGULAR;
        if ((uint32_t)lane < a.general_tiles && t < a.ntiles) fl = a.ws[(size_t)t * KMO_BOX_INTS + 5];
        const unsigned long long todo = __ballot(!(fl & KMO_F_REGULAR) || (fl & KMO_F_NONFINITE));
        if (lane == 0) s_todo = todo;
    }
    __syncthreads();
    unsigned long long todo = s_todo;
    if (todo == 0ull) continue;  // (block-uniform)
    if (!zeroed) {
        for (int e = tid; e < CC * KMO_PLANE / 4; e += KMO_NT) ((int4*)l.s_acc)[e] = make_int4(0, 0, 0, 0);
        zeroed = true;  // (every flush leaves the accumulators zeroed)
    }
    for (int k = 0; k < 64; ++k) {
        if (!((todo >> k) & 1ull)) continue;
        const int t = (int)t0 + k;
        KmoTile d;
        float m[9];
        kmo_describe(a, t, a.ws + (size_t)t * KMO_BOX_INTS, d);
        kmo_matrix(a.ws + (size_t)t * KMO_BOX_INTS, m);
        __syncthreads();  // the previous tile's flush is done with the accumulators, its readers with the source tile
        {
            float S[CC * KMO_PLANE / KMO_NT];
            kmo_issue_src<T, CC>(a, d, S);
            kmo_store_src<CC>(d, S, l.s_src, fillv, is_fill);
        }
        __syncthreads();
        float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float inv_scale = 1.f;
        bool finite = true;
        kmo_general_tile<T, CM, ALIGN, CC, PADX>(a, m, d, kc, l.s_u4, l.s_v4, l.s_acc, l.s_src, l.s_red, A, inv_scale, finite);
        __syncthreads();
        kmo_flush<T, CC>(a, d, l.s_acc, finite, inv_scale);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double s = km_wave_sum((double)A[i]);
            if (lane == 0) l.s_gm[wave * 9 + i] = s;
        }
        __syncthreads();
        kmo_gm_commit<CM>(l.s_gm, a.gmat ? a.gmat + (size_t)d.bm * 9 : nullptr, tid);
    }
    }
}

template <typename T, int CM, int ALIGN, int CC, int PADX>
static int kmo_launch_k(const KmWarpFusedArgs<T>& a, hipStream_t s) {
    constexpr int lds = kmo_lds_bytes(CC);
    // (the attribute is kept per DEVICE by the runtime: one flag per device ordinal; idempotent - a race sets the same values twice)
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<bool>* attr_set_p = (dev >= 0 && dev < 64) ? &attr_done[dev] : nullptr;
    if (!attr_set_p || !attr_set_p->load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)km_warp_bwd_fused_kernel<T, CM, ALIGN, CC, PADX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)km_warp_bwd_general_kernel<T, CM, ALIGN, CC, PADX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { km_set_error("km_warp2d_bwd(fused): hipFuncSetAttribute(%d bytes of LDS) failed: %s", lds, hipGetErrorString(e)); return (int)e; }
        if (attr_set_p) attr_set_p->store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((km_warp_bwd_boxes_kernel<T, CM>), dim3((a.ntiles + 255u) / 256u), dim3(256), 0, s, a);
    bool forked = false;
    if (a.first && a.scan && PADX == 0 && a.gmat) {  // (border / reflection: every output pixel is some tile's)
        const uint64_t nblk = (uint64_t)((a.g.w + KMO_SCAN_BLK - 1) / KMO_SCAN_BLK) * (uint64_t)((a.g.h + KMO_SCAN_BLK - 1) / KMO_SCAN_BLK) * (uint64_t)a.g.B;
        const uint64_t want = (nblk + KMO_SCAN_NT - 1) / KMO_SCAN_NT;
        const int cus = km_device_cus();
        const uint64_t cap = (uint64_t)(cus > 0 ? cus : 2) * 4u;  // (one wave per SIMD beside the persistent workgroup: 4 waves = 1 workgroup per CU at a time; a few rounds)
        hipStream_t side = KMO_SCAN_FORK ? km_side_fork(s) : nullptr;
        hipLaunchKernelGGL((km_warp_bwd_scan_kernel<T, CM, ALIGN>), dim3((unsigned)(want < cap ? (want ? want : 1) : cap)), dim3(KMO_SCAN_NT), 0, side ? side : s, a);
        forked = side != nullptr;
    }
    hipLaunchKernelGGL((km_warp_bwd_fused_kernel<T, CM, ALIGN, CC, PADX>), dim3(a.nworkers), dim3(KMO_NT), (size_t)lds, s, a);
    if (forked) {
        const int rc = km_side_join(s);
        if (rc != 0) return rc;
    }
#ifndef KMO_NO_GENERAL  // (variant libraries only: what the launch itself costs)
    {
        const uint32_t ngroups = (a.ntiles + a.general_tiles - 1u) / a.general_tiles;
#ifdef KMO_GENERAL_FULL_GRID
        const uint32_t ggrid = ngroups;
#else
        const uint32_t ggrid = ngroups < a.general_grid ? ngroups : a.general_grid;
#endif
        hipLaunchKernelGGL((km_warp_bwd_general_kernel<T, CM, ALIGN, CC, PADX>), dim3(ggrid), dim3(KMO_NT), (size_t)lds, s, a);
    }
#endif
    return km_check_launch("km_warp2d_bwd(fused)");
}
template <typename T, int CM, int PADX>
static int kmo_launch_p(const KmWarpFusedArgs<T>& a, hipStream_t s) {
    if (a.cc == 3) return a.g.align ? kmo_launch_k<T, CM, 1, 3, PADX>(a, s) : kmo_launch_k<T, CM, 0, 3, PADX>(a, s);
    return a.g.align ? kmo_launch_k<T, CM, 1, 1, PADX>(a, s) : kmo_launch_k<T, CM, 0, 1, PADX>(a, s);
}
template <typename T, int CM>
static int kmo_launch(const KmWarpFusedArgs<T>& a, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {  // (border / reflection: fp32 storage only - km_warp_bwd_fused_supported)
        if (a.g.pad == KM_PAD_BORDER) return kmo_launch_p<T, CM, KM_PAD_BORDER>(a, s);
        if (a.g.pad == KM_PAD_REFLECTION) return kmo_launch_p<T, CM, KM_PAD_REFLECTION>(a, s);
    }
    return kmo_launch_p<T, CM, 0>(a, s);
}

// one launch sequence (boxes, persistent loop, general) over channels c0 .. c0 + ngrp * cc - 1 of every image
template <typename T>
static int kmo_run_groups(KmWarpFusedArgs<T>& a, int B, uint32_t c0, uint32_t ngrp, uint32_t cc, bool first, int coord_mode, hipStream_t s) {
    a.c0 = c0; a.ngrp = ngrp; a.cc = cc; a.first = first ? 1u : 0u;
    a.scan = km_config().warp_bwd_no_scan ? 0u : 1u;
    const uint64_t ntiles = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * ngrp;
    a.ntiles = (uint32_t)ntiles;
    const int cus = km_device_cus();
#ifdef KMO_WORKERS_OVERRIDE  // (variant libraries only: fewer persistent workgroups than CUs - what the clock does when part of the chip idles)
    const uint32_t workers_max = KMO_WORKERS_OVERRIDE;
#else
    const uint32_t workers_max = cus > 0 ? (uint32_t)cus * KMO_WG_PER_CU : 3u;  // (the host build of the kernels reports 0 CUs: a few workers, several tiles each)
#endif
    // a run = a row of tiles of one image when that still gives every CU several runs; single tiles otherwise
    const uint64_t rows = (uint64_t)a.tiles_y * (uint64_t)B * ngrp;
    a.run_len = (rows >= 4ull * workers_max) ? a.tiles_x : 1u;
    a.nruns = (uint32_t)(ntiles / a.run_len);
    a.nworkers = (uint32_t)((uint64_t)a.nruns < (uint64_t)workers_max ? a.nruns : workers_max);
    {   // the general launch: groups of up to 64 tiles (one ballot), one workgroup per persistent worker walking its share of them
        // (one group per workgroup when the persistent workers' count of them covers the tiles: one ballot, one pair of barriers)
        const uint64_t per = (ntiles + (uint64_t)workers_max - 1) / (uint64_t)workers_max;
        a.general_tiles = (uint32_t)(per < 1 ? 1 : (per > 64 ? 64 : per));
        a.general_grid = workers_max;
    }
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmo_launch<T, KM_COORD_PERSPECTIVE>(a, s);
        case KM_COORD_AFFINE: return kmo_launch<T, KM_COORD_AFFINE>(a, s);
        default: return kmo_launch<T, KM_COORD_HOMOGRAPHY>(a, s);
    }
}

template <typename T>
static int kmo_run(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, void* ws, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, hipStream_t s) {
    KmWarpFusedArgs<T> a;
    a.gout = (const T*)gout; a.src = (const T*)src; a.mat = (const float*)mat; a.gsrc = (float*)gsrc; a.gmat = gmat; a.fill = (const float*)fill;
    a.ws = (int*)ws;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, pad, align);
    a.tiles_x = (uint32_t)((W + KMT_TW - 1) / KMT_TW);
    a.tiles_y = (uint32_t)((H + KMO_TH - 1) / KMO_TH);
    // C = 3 a3 + r1: the groups of three through the RGB instantiation, the rest one channel at a time through the grey one
    const uint32_t a3 = (uint32_t)C / 3u, r1 = (uint32_t)C % 3u;
    const uint64_t per_image = (uint64_t)a.tiles_x * a.tiles_y;
    KM_REQUIRE(per_image * (uint64_t)B * (a3 > r1 ? a3 : r1) < (1ull << 26) && (uint64_t)B * (uint64_t)C < (1ull << 31) && a.tiles_x < 65536u && a.tiles_y < 65536u,
               "km_warp2d_bwd: grid too large");
    if (per_image * (uint64_t)B == 0 || C == 0)  // (nothing to add: the accumulators are still this path's to zero; gmat == nullptr: image gradient only)
        return (gmat && B_M > 0) ? (int)hipMemsetAsync(gmat, 0, (size_t)B_M * 9 * sizeof(double), s) : 0;
    a.reverse = km_traversal_next(s);
    a.stream_out = km_stream_stores((uint64_t)B * C * H * W * sizeof(float));
    int rc = 0;
    if (a3) rc = kmo_run_groups(a, B, 0u, a3, 3u, true, coord_mode, s);
    if (rc == 0 && r1) rc = kmo_run_groups(a, B, 3u * a3, r1, 1u, a3 == 0, coord_mode, s);
    return rc;
}

// 1 if the one-read kernel computes both gradients for these modes (bilinear, every padding mode - border / reflection with fp32 storage -,
// fp32 compute; any channel count - fill
// values are per channel of an RGB / grey image, so `fill` keeps C in {1, 3})
int km_warp_bwd_fused_supported(int interp, int pad, int dtype, int C, int H, int W, int h, int w) {
    if (!km_config().warp_bwd_fused) return 0;
    if (!(interp == KM_INTERP_BILINEAR && dtype != KM_F64)) return 0;
    if ((pad == KM_PAD_BORDER || pad == KM_PAD_REFLECTION) && dtype != KM_F32) return 0;  // (these two: fp32 storage)
    if (C < 1 || (pad == KM_PAD_FILL && !(C == 1 || C == 3))) return 0;
    // 32-bit byte offsets inside a plane
    return ((uint64_t)H * W * 4 < (1ull << 32) && (uint64_t)h * w * 4 < (1ull << 32)) ? 1 : 0;
}

// bytes of workspace the one-read backward needs for these sizes: one KMO_BOX_INTS * 4 = 80-byte record per 64 x 64 tile of the source and
// channel group of the larger of its (at most two) launch sequences
size_t km_warp_bwd_fused_workspace(int B, int C, int H, int W) {
    const uint64_t a3 = (uint64_t)C / 3u, r1 = (uint64_t)C % 3u;
    const uint64_t ntiles = (uint64_t)((W + KMT_TW - 1) / KMT_TW) * (uint64_t)((H + KMO_TH - 1) / KMO_TH) * (uint64_t)B * (a3 > r1 ? a3 : r1);
    return (size_t)(ntiles * KMO_BOX_INTS * sizeof(int));
}

int km_warp_bwd_fused_run(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, void* ws, int B, int C, int H, int W, int h, int w,
                          int B_M, int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmo_run<float>(gout, src, mat, gsrc, gmat, ws, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
#ifndef KMO_DEV_F32_ONLY  // (development builds: one storage type compiles in a third of the time)
        case KM_BF16: return kmo_run<km_bf16>(gout, src, mat, gsrc, gmat, ws, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        default: return kmo_run<km_f16>(gout, src, mat, gsrc, gmat, ws, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
#else
        default: return -1;
#endif
    }
}

#ifdef KMO_PROFILE
// (variant libraries only) the per-wave phase table of the last persistent launch: [worker][wave][phase] shader cycles
extern "C" int km_debug_fused_profile_rt(unsigned long long* host_out, int n_entries) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmo_prof_rt), (size_t)n_entries * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int km_debug_fused_profile(unsigned long long* host_out, int n_entries) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(kmo_prof_out), (size_t)n_entries * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

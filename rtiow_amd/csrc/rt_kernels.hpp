// rt_kernels.hpp -- the render megakernel and the resolve kernels (gfx950).
//
// One launch renders a shard of the image: main.rs:122-139 without to_rgba.
//
// Execution shape:
//   * persistent grid: as many 256-thread workgroups as stay resident; every
//     LANE owns one path at a time and never waits for its neighbours -- when
//     its path ends it immediately starts its next pixel-sample ("path
//     regeneration"), so the sphere scan always runs with full waves;
//   * work = single pixel-samples in PIXEL-MAJOR order (item w = pixel * spp + sample); a wave
//     reserves a block of kItemBlock consecutive items with ONE returning atomic on a device-wide
//     counter, STARTS them 64 at a time (item -> pixel, Philox, lens sample: every lane busy) into a
//     per-wave LDS queue and deals the started samples to the lanes whose paths end (__ballot +
//     popcount + v_mbcnt rank); a block is at most 8 neighbouring pixels, so its 64-lane wave traces
//     coherent camera rays and the launch ends on single samples;
//   * the sphere scan (HittableList::hit, mod.rs:54-70) decides nothing: it is a
//     conservative FILTER (rt_device.hpp) -- it may send a sphere to the exact
//     test needlessly, never drop one the reference would hit.  The shipped form
//     (MODE 5, the tube filter) evaluates it on the bf16 matrix pipe, 16 rays x 2
//     directions x 32 spheres per instruction, and only over the tiles of spheres the wave's
//     rays can reach (the table is tiled by position: a grid over x and z, rt_device.hpp
//     grid_cells); MODE 1 (VALU + scalar loads) and MODE 0 (no filter at all) are the
//     validation modes of the product build, MODEs 2-4 (earlier matrix-pipe forms) exist
//     only in the cross-check build (-DRTIOW_CROSSCHECK_MODES);
//   * spheres the filter keeps are marked in per-ray LDS bitmaps, pooled over the
//     wave and put through the reference's exact f64 test (sphere.rs:16-34), so every
//     hit decision and every shading value is the reference's own f64 arithmetic;
//   * radiance is exact u64 fixed point (contract C5): a finished sample is added to its block's
//     per-pixel sums in the wave's LDS, and a block goes to the frame buffer ONCE, when its last
//     sample has finished (one 64-bit atomic per pixel and channel per block: the frame buffer sees
//     ~3 x 64-byte memory-side requests per 256 samples).
#pragma once
#include "rt_device.hpp"
#include "rt_diag.hpp"
#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_params.hpp"
#endif
#include <type_traits>

#ifndef RT_GROUP_SKIP
#define RT_GROUP_SKIP 1     // MODE 5: a 16-ray group skips the tiles none of its rays can reach (0: every group scans the wave's whole list)
#endif

namespace rt {

struct KCamera {
    double origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius;
};

struct KParams {
    KCamera cam;
    double t_min;
    int32_t width, height;
    int32_t spp, sample_begin, max_depth;
    uint32_t k0, k1;
    int32_t tile_rows, shard_index, shard_count;
    int32_t rows;              // compact rows of this shard
    int32_t n_spheres;
    int32_t use_ring;          // 1: per-wave LDS block sums (whenever a block's pixels fit the ring's pixel slots: rt_api.hip picks block_items for that); 0: every sample goes to the frame buffer directly
    uint32_t npix;             // rows * width
    uint32_t n_blocks;         // work blocks of block_items consecutive pixel-samples, pixel-major: item w = pixel * spp + sample
    unsigned long long total_items;   // npix * spp
    double inv_spp;            // 1.0 / spp (block -> first pixel)
    double inv_width;          // 1.0 / width (first pixel -> row, column)
    // per-lane divisions of small numerators by launch constants, as multiply-high (udiv_small below):
    // floor(2^32 / d) + 1 for d = spp, width, tile_rows (unused where d == 1 or d >= 2^16)
    uint32_t magic_spp, magic_width, magic_tile;
    uint32_t block_items;      // pixel-samples per work block: ITEMS, or -- launches of few samples per pixel -- the multiple of 64 below it whose pixels
                               // still fit the ring's pixel slots (rt_api.hip)
    const float *filt;         // [n][4]  f32 filter record (cx, cy, cz, K')
#ifdef RTIOW_CROSSCHECK_MODES
    KXcheckTables x;           // the B operands of scan modes 2-4 (xcheck/rt_xcheck_params.hpp)
#endif
    const uint4 *btube;        // [tiles/2 + 1][64] MODE 5 B operands: 32 spheres x 16 K-slots, each column scaled by 2 / its bound (rt_device.hpp)
    float tube_rho;            // MODE 5 radius floor
    float scene_scale;         // MODE 5: sum over axes of the largest |coordinate| a scanned sphere reaches (error margins of grid_cells)
    // MODE 5: the table's columns are ordered by position (rt_api.hip): tiles [0, n_global) hold the spheres every
    // ray scans, tile n_global + iz * grid_dim + ix those whose centre lies in cell (ix, iz) of a square xz grid.
    const double *geo_slot;    // [columns][4] exact (cx, cy, cz, r*r) in column order
    const uint32_t *slot_orig; // [columns] place in the caller's list of the sphere in each column
    float grid[8];             // x0, z0, 1/cell, x1, z1, y lo, y hi, largest radius of a sphere that lives in a cell
    int32_t grid_dim;          // cells per side; 0: no grid (every tile is scanned, columns in list order)
    unsigned long long grid_rows;      // bit k * grid_dim for every k with (k + 1) * grid_dim <= 64 (grids of <= 64 cells)
    int32_t n_global;
    int32_t n_tiles;
    int32_t n_always;          // spheres that skip the filter and are always tested exactly
    int32_t always_idx[8];
    const double *geo;         // [n][4]  exact (cx, cy, cz, r*r)
    const double *mat;         // [n][10] exact (1/r, param, albedo rgb, kind, 1/param, r0(1/ir), r0(ir), -)
    unsigned long long *fix;   // [rows][width][3] exact sums
    unsigned int *queue;       // work counter
    unsigned long long *stats; // [0] rays [1] samples [2] candidates [3] exact roots [4] samples sent to the frame buffer one by one [8..15] diagnostic builds [16..79] rays per bounce index (DIAG)
};

constexpr int RT_KIND_LAMBERTIAN = 0, RT_KIND_METAL = 1, RT_KIND_DIALECTRIC = 2;
constexpr int kBlock = 256;
constexpr int kCandCap = 24;        // MODE 1: per-lane candidate slots
constexpr int kScanUnroll = 8;      // MODE 1: spheres per scalar-load batch / overflow check
constexpr int kItemBlock = 256;     // pixel-samples a wave reserves per atomic on the work counter
// ... and 1 024 for launches of at least 2 x 10^8 pixel-samples at >= 147 samples per pixel (ceil(1023 / 147) + 1 = 8 pixels: still kRingSlots; the
// small-grid kernel: >= 69 samples per pixel, ceil(1023 / 69) + 1 = 16 pixels, its large blocks' ring has 2 x 16 pixel slots, render_kernel below):
// reserving a block is a returning atomic the whole wave waits for, and a block's sums are one frame-buffer request per pixel and channel:
// a quarter of both (1200x675x500: 52.7 -> 51.3 ms; 10k spheres 1920x1080x256: 93.7 -> 91.5 ms).  Smaller launches keep 256: their last
// blocks are the end-of-launch tail (1200x675x147 is 5 % slower with 1 024).
constexpr int kItemBlockLarge = 1024;
static_assert(kItemBlockLarge + 32768 < 65536, "udiv_small: numerators x < d + kItemBlockLarge with d < 2^15 keep x * d < 2^32");
constexpr int kLargeMinSpp = 147;
constexpr int kLargeMinSppSmallGrid = 69;
constexpr unsigned long long kLargeMinItems = 200000000ull;   // (1200x675: blocks of 1 024 against 256 at 150 / 200 / 250 / 300 / 350 spp: +3.5 / -0.5 / -1.9 / -2.2 / -2.5 %)
constexpr int kRingSlots = 8;       // pixels a block may touch when its sums are kept in LDS: ceil((block_items - 1) / spp) + 1 <= 8 -- blocks of 256 from
                                    //   37 spp per launch on, of 192 / 128 / 64 down to 9 spp (rt_api.hip); below that samples go to the frame buffer one by one
constexpr int kRingDepth = 4;       // blocks of one wave that may be unfinished at the same time (older ones: see `orphan`)
                                    //   (the shipped scan mode's kernels: 2 blocks x 16 pixels, render_kernel below: blocks of 256 from 17 spp on, sums down to 5 spp)
constexpr int kMatStride = 10;      // doubles per material record
constexpr int kListCap = 126;       // MODE 5: longest list of tiles a wave scans by list; beyond, it scans the whole table
constexpr int kSegTilesTube = 28;   // MODE 5: 14 bitmap words of 32 columns per segment (a wave's list is 5-13 tiles long)
constexpr int kSegTiles = 36;       // matrix filter: tiles (of 16 spheres) per candidate-bitmap segment

__device__ __forceinline__ D3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

// x / d for a launch constant d >= 1 and a numerator with x < d + 65536 and x < 2^31, without a division (a `/`
// would have the compiler keep one reciprocal per divisor in a VGPR for the whole bounce loop):
//   d == 1: x;  d >= 2^15: the quotient is 0 or 1;  else floor(x * M / 2^32), M = floor(2^32/d) + 1,
//   exact because x * (M d - 2^32) <= x d < 2^32: the numerators here are < d + kItemBlockLarge (spp, width) or < 2^16
//   (rows / tile_rows), so x < 2^15 + 2^10 and x d < 2^32 for every d < 2^15.
__device__ __forceinline__ uint32_t udiv_small(uint32_t x, uint32_t d, uint32_t magic)
{
    const uint32_t big = x >= d ? 1u : 0u;
    const uint32_t q = __umulhi(x, magic);
    return d == 1u ? x : (d >= 32768u ? big : q);
}



// MODE 0: every sphere goes through the exact test (validation mode, RT_FLAG_NO_FILTER):
//         same results by construction of the filter.
// MODE 1: f32 filter on the VALU with scalar-loaded sphere records + deferred exact tests.
// MODEs 2-4 are earlier matrix-pipe forms of the filter, kept as cross-checks and compiled only with
// -DRTIOW_CROSSCHECK_MODES (the product library carries modes 0, 1 and 5):
// MODE 2: the same filter on the f32 MATRIX pipe: the filter is two K = 4 products,
//         HB = R1 x S and Q = R2 x S, of per-ray rows R1 = (-g, o.g), R2 = (-2o, |o|^2(1-kappa))
//         with per-sphere columns S = (c, 1); v_mfma_f32_16x16x4_f32 evaluates them for
//         16 rays x 16 spheres at a time as the very fma chains of filter_keeps().
// MODE 3: the same two products on the bf16 matrix pipe (16x the f32 rate): every f32 operand
//         is the exact sum of three bf16 pieces, and one v_mfma_f32_16x16x32_bf16 adds up the
//         8 significant piece products of each of the 4 components (K = 32).
// MODE 4: the filter as ONE contraction of 11 per-ray terms with 11 per-sphere terms (rt_device.hpp,
//         "lifted" form) on the bf16 matrix pipe, two chained K = 32 MFMAs per 16 rays x 16 spheres:
//         the VALU no longer squares and subtracts, it only looks at the sign of the result.
// MODE 5: the tube filter: |u_k.(c-o)| <= r for two directions u_1, u_2 perpendicular to the ray, linear
//         in c, so two bf16 pieces per factor suffice: ONE v_mfma_f32_32x32x16_bf16 tests 16 rays
//         (both directions) against 32 spheres -- a third of the matrix-pipe time of modes 3/4.
constexpr int kRowPad = 80;         // floats per row of the ray-operand transpose buffer

typedef float f32x4 __attribute__((ext_vector_type(4)));

// DIAG: also count filter candidates, exact roots and rays per bounce index (rt_stats.candidates /
// exact_roots / live_per_bounce).  Off in the shipped path: the counters cost spilled registers and
// ~15 % of the frame time.
// SMALLGRID (MODE 5 only): the scene's tile grid has at most 64 cells (with the global tiles: a list of <= 64 tiles) --
// the host picks this instantiation then; it carries neither the large-grid list code nor the scan-every-tile fallback.
// U53 (RT_FLAG_UNIFORM53): every uniform takes two consecutive Philox words (53 random bits, as rand's gen::<f64>()) instead
// of one word's 32 bits: same draw order, same runs; the rejection tests run in f64 as the reference writes them (the
// integer form needs the 2^-31 lattice of single words).  An optional mode: ~2x the Philox work of the retry loops.
// ITEMS: pixel-samples per work block (kItemBlock, or kItemBlockLarge for launches with enough samples: see rt_api.hip).
template <int MODE, bool DIAG, bool SMALLGRID = false, bool U53 = false, int ITEMS = 256>
// second launch bound = waves per SIMD the register allocator must leave room for: the bounce loop
// is latency-bound, and the 4th wave is worth more than the few cold values it spills.  Only the shipped kernel
// (MODE 5 without the diagnostic counters) fits four workgroups' LDS on a CU (40 000 of 40 960 bytes each); the
// diagnostic variant and the cross-check modes 2-4 carry 1-14 KB more and run three.
__global__ __launch_bounds__(kBlock, (MODE == 5 && !DIAG) ? 4 : (MODE >= 2) ? 3 : 5) void render_kernel(const KParams P)
{
    // The block sums' ring (s_ring below) has the same 768 bytes per wave in both shapes: 4 blocks x 8 pixels, or -- the shipped scan mode's
    // kernels, all but the large-grid kernel's instantiation for blocks of 1 024 -- 2 blocks x 16 pixels.  Two blocks in flight are enough: a block of 256 lasts ~11 passes, one of
    // 1 024 ~43, and the samples of the block before the previous one that are still open when a block begins (paths of more than
    // ~15 / ~50 bounces: ~1 in 10^3 / none) go the orphans' way; two counters instead of four, a shorter cascade when samples finish.  And 16
    // pixel slots let launches from 17 samples per pixel on keep the sums of blocks of 256 in LDS (ceil(255 / 17) + 1 = 16; smaller blocks: from 5) and
    // launches from kLargeMinSppSmallGrid = 69 on take large blocks (ceil(1023 / 69) + 1 = 16).  Measured, interleaved: 1200x675x500 48.87 ->
    // 48.74 ms, 1200x675x100 10.53 -> 10.49 ms, 10k spheres 1920x1080 x20 7.78 -> 7.50 ms, x100 34.98 -> 34.88 ms; the large-grid kernel ON LARGE
    // BLOCKS is 0.7 % slower with the same ring (10k spheres 1920x1080x256: 85.99 -> 86.58 ms, fewer instructions, another schedule) and
    // keeps 4 x 8 and kLargeMinSpp = 147.  The cross-check scan modes and the diagnostic-counter kernels keep 4 x 8 too.
    constexpr bool kWideRing = SMALLGRID || (MODE == 5 && !DIAG && ITEMS == kItemBlock);
    constexpr int kRingDepth = kWideRing ? 2 : 4;
    constexpr int kRingSlots = kWideRing ? 16 : 8;
    static_assert(kRingDepth * kRingSlots == rt::kRingDepth * rt::kRingSlots, "same LDS either way");
    __shared__ uint16_t cand[MODE == 1 ? kCandCap : 1][MODE == 1 ? kBlock : 1];     // MODE 1: per-lane candidate lists
    constexpr bool MATRIX = (MODE >= 2);
    constexpr bool LIFTED = (MODE == 4);
    constexpr bool TUBE = (MODE == 5);
    constexpr bool RAYOP = (MODE == 2 || MODE == 3);
    __shared__ float s_rayop[RAYOP ? kBlock / 64 : 1][RAYOP ? 8 : 1][RAYOP ? kRowPad : 1];
    // per ray, one bit per sphere of the current segment (kSeg tiles of 16 columns): set by whichever lane
    // holds the passing result (ds_or, nothing returned, nothing waited for), read by the owner.  The same LDS
    // first carries the per-ray operand dwords to the MFMA layout (4 KB) and, for large grids, the wave's cell
    // bitmap (512 B).
    constexpr int kSeg = TUBE ? kSegTilesTube : kSegTiles;
    static_assert(TUBE || (kSeg / 2) * 64 * sizeof(unsigned int) >= 32 * kStageStride * sizeof(uint4), "operand staging fits the bitmap");
    // MODE 5: bitmap words (14 x 256 B) and the list of tiles this pass scans (128 x 4 B) share the 4 KB the staging
    // needs before either exists
    static_assert(!TUBE || ((kSeg / 2) * 64 + kListCap + 2) * sizeof(unsigned int) == 64 * 4 * sizeof(uint4), "tube: staging = bitmap + tile list");
    constexpr int kBitWords = TUBE ? (kSeg / 2) * 64 + kListCap + 2 : (kSeg / 2) * 64;
    __shared__ __attribute__((aligned(16))) unsigned int s_bits[MATRIX ? kBlock / 64 : 1][MATRIX ? kBitWords : 1];
    __shared__ unsigned int s_sum[(MATRIX && !TUBE) ? kBlock : 1];     // per ray: which of its bitmap words are non-zero
    // The wave's queue of started samples (below, (a)): the image-plane coordinates u, v (f64) and, as 32-bit columns,
    // the two accepted lens words, pixel, sample and (events drawn << 8 | pixel slot of the block): 36 bytes per entry.
    __shared__ double s_quv[kBlock / 64][2][64];
    __shared__ unsigned int s_qid[kBlock / 64][5][64];
    // pooled exact tests: ring of waiting (column << 6 | ray) pairs per wave (columns < 2^26); per-ray minimum root and its sphere
    __shared__ unsigned int s_pool[MATRIX ? kBlock / 64 : 1][MATRIX ? 128 : 1];
    __shared__ unsigned long long s_best[MATRIX ? kBlock : 1];
    __shared__ unsigned int s_bidx[MATRIX ? kBlock : 1];
    // Per wave, the exact sums (u64 fixed point) of up to kRingDepth unfinished blocks, [pixel slot][channel]; a block is
    // written to the frame buffer once, by whichever pass finishes its last sample (how many samples each block
    // still waits for is wave-uniform state in SGPRs).  s_rpix: first pixel (compact index) of the entry's block.
    __shared__ unsigned long long s_ring[kBlock / 64][kRingDepth][kRingSlots * 3];
    __shared__ unsigned int s_rpix[kBlock / 64][kRingDepth];
    __shared__ unsigned int s_live[DIAG ? kBlock / 64 : 1][DIAG ? 64 : 1];    // DIAG: rays per bounce index, per wave
    // so does the path throughput (contract C3): read and written once per bounce
    __shared__ double s_thr[3][kBlock];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    {   // this wave's block sums and their bookkeeping (own wave only: no barrier needed)
        unsigned long long *z = &s_ring[tid >> 6][0][0];
        for (int k = lane; k < kRingDepth * kRingSlots * 3; k += 64) z[k] = 0ull;
        if (lane < kRingDepth) s_rpix[tid >> 6][lane] = 0u;
    }
    if (DIAG) s_live[tid >> 6][tid & 63] = 0u;                          // this wave's row only
    // number of set bits of a wave mask below this lane (v_mbcnt: no per-lane 64-bit mask to keep)
    auto rank_below = [](unsigned long long m) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };

    // The filter table is read-only for the whole launch and indexed by a
    // wave-uniform counter: reading it through the constant address space makes
    // the scan's loads scalar (s_load_dwordx4 into SGPRs).
    const float __attribute__((address_space(4))) *filt =
        (const float __attribute__((address_space(4))) *)(uintptr_t)P.filt;
    const double *__restrict__ geo = P.geo;
    const double *__restrict__ mat = P.mat;
    const int n = P.n_spheres;
    const double t_min = P.t_min;
    const double wm1 = (double)(P.width - 1);
    const double hm1 = (double)(P.height - 1);

    bool dead = false, alive = false;
    uint32_t pix_local = 0, pix_global = 0;
    int s = 0;
    uint32_t my_blk = 0;                           // sequence number (within this wave) of the block of this lane's sample << 4 | its pixel slot
    D3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    int depth = 0;
    uint32_t ev = 0;
    uint32_t n_rays = 0, n_samples = 0, n_direct = 0;   // wave totals (uniform)
    unsigned long long tot_cand = 0, tot_roots = 0;     // wave totals (uniform)
    // this wave's current block of work items (all wave-uniform): items [blk_next, blk_end) of it are still to be dealt;
    // its first item is sample blk_s0 of compact pixel blk_pix0 = row blk_rr0, column blk_i0 of the shard
    uint32_t blk_next = 0, blk_end = 0, blk_seq = 0xFFFFFFFFu, blk_pix0 = 0, blk_s0 = 0, blk_rr0 = 0, blk_i0 = 0;
    bool queue_empty = false;
    // the wave's queue of camera rays: entries [q_head, q_head + q_count) of s_qray / s_qid (wave-uniform)
    uint32_t q_head = 0, q_count = 0;
    double *quv_w = &s_quv[tid >> 6][0][0];
    unsigned int *qid_w = &s_qid[tid >> 6][0][0];
    unsigned long long *ring_w = &s_ring[tid >> 6][0][0];
    unsigned int *rpix_w = &s_rpix[tid >> 6][0];
    // samples the wave's four youngest blocks still wait for (wave-uniform; index = age: 0 is the current block
    // blk_seq, k is block blk_seq - k, whose sums are ring entry (blk_seq - k) % kRingDepth); 0 = complete or none
    uint32_t rem0 = 0, rem1 = 0, rem2 = 0, rem3 = 0;           // (rem2, rem3: rings of 4 only)
    static_assert(kRingDepth == 4 || kRingDepth == 2, "the block counters are written out for a ring of 4 or 2");
    // one block's sums -> frame buffer (wave-uniform call; `e` uniform): lane l < 24 holds (pixel slot l/3, channel l%3),
    // which are 24 consecutive u64 of the frame buffer; zero sums (unused slots, black pixels) are not sent
    auto flush_ring = [&](uint32_t e) {
        if (lane < kRingSlots * 3) {
            const unsigned long long v = ring_w[e * (kRingSlots * 3) + lane];
            if (v != 0ull) {
                atomicAdd(P.fix + (size_t)rpix_w[e] * 3u + (size_t)lane, v);
                ring_w[e * (kRingSlots * 3) + lane] = 0ull;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    RT_DIAG_DECLARE();              // (rt_diag.hpp: nothing in the shipped library)
    // One pass of the bounce loop = five phases, marked "==== PHASE n" below; the helpers they call are the lambdas defined next to their
    // data (flush_ring, exact_test_g, pool_round, enumerate, finish_pool, look_tube, do_tile, build_list).  The phases themselves stay
    // INLINE: wrapped into named lambdas (round 5 tried it: take_samples / start_camera_rays / shade / accumulate_finished) the same code
    // compiles to a different register allocation at this kernel's zero-headroom budget (128 VGPRs) -- one wrapper alone puts 20 scratch
    // instructions into the small-grid kernel, all four cost +0.3 % frame time (profiles/r05_experiments.txt 8) -- so the structure is in
    // the comments and tools/isa_fingerprint.py guards refactorings (the machine code of this file's kernels before = after).
    for (;;) {
        // Wave priorities through a pass (s_setprio): 1 from here -- taking samples, refill, camera rays, then the filter rows, the ground
        // test and the tile list --, 3 in the tile loop (the matrix pipe is fed sooner and the other waves' vector work fills the time
        // the MFMAs take), 2 while the candidates are enumerated, 1 in the pooled exact rounds, 0 while the hit is shaded (the redraw
        // loop runs at a quarter of the lanes), 1 again for the finished samples' sums.  Measured against round 2's choice (1 in the
        // tile loop, 0 elsewhere): 1200x675x500 50.30 -> 49.73 ms (-1.15 %), configs[1] 10.79 -> 10.69, 10k spheres -1.0 %; no
        // priorities at all +1.1 %; a dozen other assignments within 0.3 % of this one or slower (profiles/r04_experiments.txt 30).
        __builtin_amdgcn_s_setprio(1);
        RT_COUNT(0);
        // ==== PHASE 1: take_samples ============================================================================================
        // ---- (a) lanes without a path take the next camera rays of the wave's queue -------------------------
        // Starting a sample (item -> pixel, Philox, lens rejection, the f64 camera arithmetic: ~340 vector
        // instructions) used to run in every pass with only the ~38 % of lanes whose path had just ended.  Now the
        // wave starts its next 64 items at once, with every lane busy -- everything up to the image-plane coordinates
        // and the accepted lens sample -- parks them in LDS (36 bytes each) and hands them out over the next ~2.7
        // passes: a lane that needs work takes entry head + its rank among the takers and builds its ray (camera.rs:47-54).  The queue is refilled only when it is empty, so its entries always belong to the
        // wave's current block (blk_seq, blk_pix0).
        bool fresh = false;                                         // this lane starts a sample in this pass
        double cam_u = 0.0, cam_v = 0.0;
        uint32_t lens_wx = 0u, lens_wy = 0u;
        for (;;) {
            const bool want = !alive && !dead && !fresh;
            const unsigned long long m = __ballot(want);
            if (m == 0ull) break;
            if (q_count == 0u) {
                RT_STAMP(0);
                // ---- (b) refill: main.rs:131-134 + camera.rs:47-54 for the next (up to) 64 items of the block ----
                // A wave reserves kItemBlock consecutive items with ONE returning atomic on the device-wide counter:
                // one word sustains only ~88 dequeues/us chip-wide (MI355X_MICROARCH.md, row "dequeue").
                if (blk_next == blk_end) {                          // (wave-uniform) the block is used up: reserve the next one
                    uint32_t nb = 0xFFFFFFFFu;
                    if (!queue_empty) {
                        const int leader = (int)__builtin_ctzll(m);
                        if (lane == leader) nb = atomicAdd(P.queue, 1u);
                        nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(nb, leader));
                    }
                    if (nb >= P.n_blocks) {                         // no work left anywhere: these lanes are done
                        queue_empty = true;
                        if (want) dead = true;
                        break;
                    }
                    // first item of the block -> (pixel, sample): W0 / spp in f64 (W0 < 2^39: exact), one correction step
                    const unsigned long long W0 = (unsigned long long)nb * (unsigned long long)P.block_items;     // (block_items <= ITEMS)
                    const unsigned long long left = P.total_items - W0;
                    const uint32_t n_items = left < (unsigned long long)P.block_items ? (uint32_t)left : P.block_items;
                    uint32_t p0 = (uint32_t)((double)W0 * P.inv_spp);
                    long long rem = (long long)(W0 - (unsigned long long)p0 * (unsigned long long)(uint32_t)P.spp);
                    if (rem < 0) { p0 -= 1u; rem += (long long)P.spp; }
                    else if (rem >= (long long)P.spp) { p0 += 1u; rem -= (long long)P.spp; }
                    blk_pix0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);
                    blk_s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rem);
                    uint32_t rr0 = (uint32_t)((double)blk_pix0 * P.inv_width);     // pixel / width the same way
                    int i0 = (int)(blk_pix0 - rr0 * (uint32_t)P.width);
                    if (i0 < 0) { rr0 -= 1u; i0 += P.width; } else if (i0 >= P.width) { rr0 += 1u; i0 -= P.width; }
                    blk_rr0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rr0);                  // (keeps them in SGPRs)
                    blk_i0 = (uint32_t)__builtin_amdgcn_readfirstlane(i0);
                    blk_next = 0u; blk_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_items);
                    blk_seq += 1u;
                    if (P.use_ring) {
                        // The oldest of the four blocks leaves the ring.  If a path of > ~30 bounces still holds it open,
                        // what it has collected goes out now and its remaining samples will go to the frame buffer
                        // directly when they finish ("orphans": their age is then >= kRingDepth).
                        if constexpr (kRingDepth == 4) {
                            if (rem3 != 0u) flush_ring(blk_seq & (uint32_t)(kRingDepth - 1));
                            rem3 = rem2; rem2 = rem1; rem1 = rem0; rem0 = blk_end;
                        } else {
                            if (rem1 != 0u) flush_ring(blk_seq & (uint32_t)(kRingDepth - 1));
                            rem1 = rem0; rem0 = blk_end;
                        }
                        if (lane == 0) rpix_w[blk_seq & (uint32_t)(kRingDepth - 1)] = blk_pix0;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                const uint32_t n_gen = min(64u, blk_end - blk_next);
                RT_COUNT_MAIN(1);
                if ((uint32_t)lane < n_gen) {
                    // item blk_next + lane of the block: sample blk_s0 + that of pixel blk_pix0, carried over into the next pixels
                    const uint32_t s_rel = blk_s0 + blk_next + (uint32_t)lane;
                    const uint32_t dq = udiv_small(s_rel, (uint32_t)P.spp, P.magic_spp);   // pixel slot within the block (< kRingSlots when use_ring)
                    const uint32_t col = blk_i0 + dq;
                    const uint32_t rq = udiv_small(col, (uint32_t)P.width, P.magic_width);
                    const uint32_t i = col - rq * (uint32_t)P.width;
                    const uint32_t rr = blk_rr0 + rq;
                    const uint32_t lt = udiv_small(rr, (uint32_t)P.tile_rows, P.magic_tile);   // local tile
                    const uint32_t j = (lt * (uint32_t)P.shard_count + (uint32_t)P.shard_index) * (uint32_t)P.tile_rows
                                       + (rr - lt * (uint32_t)P.tile_rows);
                    const uint32_t g_pix = j * (uint32_t)P.width + i;
                    const uint32_t g_s = (uint32_t)P.sample_begin + (s_rel - dq * (uint32_t)P.spp);
                    U4 w = philox4x32_10(g_pix, g_s, 0u, 0u, P.k0, P.k1);
                    uint32_t g_ev = 1u;
                    const double u = ((double)i + (U53 ? u01_53(w.x, w.y) : u01(w.x))) / wm1;     // main.rs:131
                    const double v = ((double)j + (U53 ? u01_53(w.z, w.w) : u01(w.y))) / hm1;     // main.rs:132
                    // vec3.rs:59-68: redraw until x*x + y*y < 1.  x = u11(w) = m * 2^-31 with the integer m = w - 2^31: the loop runs
                    // on the integers (rt_device.hpp, unit_disk_accepts: the reference's f64 comparison, bit for bit), the accepted
                    // pair is converted once.
                    // Block 0 = (u jitter, v jitter, lens x, lens y); every further block holds TWO tries.
                    uint32_t wx = w.z, wy = w.w;
                    if constexpr (U53) {
                        // B_0 = (u jitter, v jitter), then ONE block per unit-disk try (two 53-bit draws); vec3.rs:59-68 in f64 as
                        // written.  The queue keeps no lens words: the lane that takes the sample recomputes block g_ev - 1.
                        double lx, ly;
                        do {
                            w = philox4x32_10(g_pix, g_s, g_ev, 0u, P.k0, P.k1);
                            g_ev++;
                            lx = u11_53(w.x, w.y); ly = u11_53(w.z, w.w);
                        } while (!(lx * lx + ly * ly < 1.0));
                    } else
                    while (!unit_disk_accepts(wx, wy)) {
                        w = philox4x32_10(g_pix, g_s, g_ev, 0u, P.k0, P.k1);
                        g_ev++;
                        wx = w.x; wy = w.y;
                        if (!unit_disk_accepts(wx, wy)) { wx = w.z; wy = w.w; }
                    }
                    quv_w[0 * 64 + lane] = u;
                    quv_w[1 * 64 + lane] = v;
                    qid_w[0 * 64 + lane] = wx;
                    qid_w[1 * 64 + lane] = wy;
                    qid_w[2 * 64 + lane] = g_pix;
                    qid_w[3 * 64 + lane] = g_s;
                    qid_w[4 * 64 + lane] = (g_ev << 8) | dq;        // (the event counter has 24 bits here: 8 million lens retries)
                }
                blk_next += n_gen;
                q_head = 0u; q_count = n_gen;
                __builtin_amdgcn_wave_barrier();
                RT_STAMP(8);
            }
            const uint32_t r = rank_below(m);
            RT_LDS_QUEUE_READS(want && r < q_count, q_head + r);
            if (want && r < q_count) {
                const uint32_t e = q_head + r;
                cam_u = quv_w[0 * 64 + e]; cam_v = quv_w[1 * 64 + e];
                lens_wx = qid_w[0 * 64 + e]; lens_wy = qid_w[1 * 64 + e];
                pix_global = qid_w[2 * 64 + e];
                s = (int)qid_w[3 * 64 + e];
                const uint32_t meta = qid_w[4 * 64 + e];
                ev = meta >> 8;
                pix_local = blk_pix0 + (meta & 255u);
                my_blk = (blk_seq << 4) | (meta & 15u);
                fresh = true;
            }
            const uint32_t cnt = (uint32_t)__popcll(m);
            const uint32_t took = cnt < q_count ? cnt : q_count;
            q_head += took; q_count -= took;
        }
        RT_STAMP(0);
        // ==== PHASE 2: start_camera_rays =======================================================================================
        if (fresh) {                                                // camera.rs:47-54
            double lx = u11(lens_wx), ly = u11(lens_wy);
            if constexpr (U53) {                                    // the accepted unit-disk try is the last block the start drew
                const U4 lb = philox4x32_10(pix_global, (uint32_t)s, ev - 1u, 0u, P.k0, P.k1);
                lx = u11_53(lb.x, lb.y); ly = u11_53(lb.z, lb.w);
            }
            const D3 cam_origin = ld3(P.cam.origin);
            const D3 rd = mk(lx, ly, 0.0) * P.cam.lens_radius;
            const D3 offset = ld3(P.cam.u) * rd.x + ld3(P.cam.v) * rd.y;
            o = cam_origin + offset;
            d = (((ld3(P.cam.llc) + ld3(P.cam.horizontal) * cam_u) + ld3(P.cam.vertical) * cam_v) - cam_origin) - offset;
            s_thr[0][tid] = 1.0; s_thr[1][tid] = 1.0; s_thr[2][tid] = 1.0;
            depth = P.max_depth;
            alive = true;
        }

        RT_STAMP(9);
        // ---- (c) every lane of the wave is out of work: done ------------------
        const unsigned long long alive_mask = __ballot(alive);
        if (alive_mask == 0ull) break;
        n_rays += (uint32_t)__popcll(alive_mask);
        // DIAG: rays traced per bounce index (0 = camera ray), per wave in LDS, flushed at exit
        if (DIAG && alive) atomicAdd(&s_live[tid >> 6][min(P.max_depth - depth, 63)], 1u);

        // ==== PHASE 3: scan (filter rows, tile list, tile loop, enumeration, pooled exact tests) ===============================
        // ---- (d) HittableList::hit, mod.rs:54-70 -------------------------------
        double closest = __builtin_inf();
        int hit = -1;
        uint32_t n_cand = 0, n_roots = 0;                           // this bounce, this lane
        const double a = length_squared(d);                         // sphere.rs:20
        // sphere.rs:16-34 for sphere idx, exactly as the reference computes it.
        // ORDERED (std::true_type): the caller visits the spheres in LIST order and this is HittableList::hit's loop body
        // as written (mod.rs:61-67 with sphere.rs:29-33's `root < t_min || t_max < root`), degenerate values included:
        // a NaN root fails neither comparison and is ACCEPTED, after which closest_so_far is NaN and every later sphere
        // with a root >= t_min is accepted too -- what the reference does with a zero-length, NaN or infinite direction.
        // Used wherever the visit IS in list order: the no-filter and VALU-filter modes, and the rays outside the
        // filter's analysed range (which is where every such direction ends up).
        auto exact_test_g = [&](int idx, const double4 g, auto ordered) {
            const D3 oc = o - mk(g.x, g.y, g.z);
            const double half_b = dot(oc, d);
            const double c = length_squared(oc) - g.w;              // g.w = radius*radius
            const double disc = half_b * half_b - a * c;
            if (disc < 0.0) return;                                 // sphere.rs:25
            if constexpr (decltype(ordered)::value) {
                if (DIAG && !(half_b > 0.0 && c > 0.0)) n_roots++;  // (counted like the other form: where that one takes a root)
                const double sq = __builtin_sqrt(disc);
                double r = (-half_b - sq) / a;                      // sphere.rs:28-34
                if (r < t_min || closest < r) {
                    r = (-half_b + sq) / a;
                    if (r < t_min || closest < r) return;
                }
                closest = r;                                        // mod.rs:63-64
                hit = idx;
                return;
            }
            // Both roots are <= 0 < t_min when the origin is outside (c > 0) and the
            // sphere lies behind the ray (half_b > 0): sqrt(disc) <= half_b, so the
            // reference's two range tests (sphere.rs:29-33) both fail.  Skip the sqrt.
            // (Not for a == 0: there the far root is 0/0, which the reference accepts -- such rays take the ORDERED form.)
            if (half_b > 0.0 && c > 0.0) return;
            if (DIAG) n_roots++;
            const double sqrtd = __builtin_sqrt(disc);
            // sphere.rs:28-34 + mod.rs:61-67, written so that the ORDER in which a ray's
            // candidates are visited does not matter: the reference keeps sphere idx iff
            // its root r* (the near root if >= t_min, else the far root) satisfies
            // t_min <= r* <= closest-so-far, so the scan ends with the smallest r*, and among
            // equal r* with the LAST sphere of the list.
            double root = (-half_b - sqrtd) / a;
            if (root < t_min) {
                root = (-half_b + sqrtd) / a;
                if (root < t_min) return;
            }
            if (root < closest || (root == closest && idx > hit)) {
                closest = root;                                     // mod.rs:63-64
                hit = idx;
            }
        };
        auto exact_test = [&](int idx, auto ordered) {
            exact_test_g(idx, *reinterpret_cast<const double4 *>(geo + 4 * (size_t)idx), ordered);
        };
        // exact tests over a per-lane list of `cnt` sphere indices in cand[][tid]
        auto test_list = [&](int cnt) {
            // trip count = longest list among the active lanes (exec-masked vote)
            for (int k = 0; __any(k < cnt); ++k) {
                if (k < cnt) { if (DIAG) n_cand++; exact_test((int)cand[k][tid], std::true_type{}); }
            }
        };

        if (MATRIX) {
            // ---- the filter on the matrix pipe: the whole wave takes part ------------
            const int wave = tid >> 6;
            const int col = lane & 15, quad = lane >> 4;
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            unsigned int *bits_w = &s_bits[wave][0];                // [word][ray]
            unsigned int *sum_w = &s_sum[TUBE ? 0 : wave * 64];         // (MODE 5 has no summary words)
#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_looks.inc"      // check_half, check_sign_half: the looks of modes 2-4
#endif
            // ---- candidates -> exact tests, pooled over the wave -------------------------
            // A ray has 1.1 candidates on average but the longest list in a wave has 5-6, and the
            // exact test is ~80 f64 instructions: testing list entry k of every lane together would
            // run the test 5-6 times per bounce at ~20 % lane use.  Instead the owners only ENUMERATE
            // their bitmaps into a per-wave ring of (ray, sphere) pairs, and whenever 64 pairs are
            // waiting each lane takes ONE pair: it fetches that ray's (o, d) from the owner lane
            // (ds_bpermute), computes the root with the reference's own f64 operations, and the
            // per-ray minimum is taken in LDS on an order-preserving u64 image of the f64 root
            // (ds_min_u64); equal roots resolve to the larger sphere index (ds_max_u32), the same
            // rule as exact_test().  The ring persists across bitmap segments.
            // The u64 image of a root for ds_min_u64.  Every value that enters is >= t_min > 0, +inf, or a NaN (a root below t_min is not a
            // root, sphere.rs:29-33; rt_render_device refuses t_min <= 0): for those the f64's own bit pattern IS order-preserving, and a NaN
            // (0x7FF8..., or 0xFFF8... with the sign set) sorts above +inf, i.e. never wins -- the identity replaces round 1-4's general
            // total-order transform (a compare, two selects and three bit operations per pooled round and again when the minimum is taken).
            auto f64_key = [](double x) -> unsigned long long { return (unsigned long long)__double_as_longlong(x); };
            auto key_f64 = [](unsigned long long k) -> double { return __longlong_as_double((long long)k); };
            auto from_lane_f64 = [](int byte_addr, double x) -> double {
                const long long b = __double_as_longlong(x);
                const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b & 0xFFFFFFFFll));
                const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
                return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
            };
            unsigned int *pool_w = &s_pool[wave][0];
            unsigned long long *best_w = &s_best[wave * 64];
            unsigned int *bidx_w = &s_bidx[wave * 64];
            // The large-grid kernel seeds each ray's minimum with what the always-exact list (or the in-order scan of a ray
            // outside the filter's range) has found, below: `closest` and `hit` then need no registers across the tile loop
            // (1 spilled VGPR instead of 7, their reloads sat in the candidate-recording path; 10k-sphere scene -0.8 %.  The
            // small-grid kernel has no spilled VGPR and measured 0.7 % SLOWER with the seed.)
            constexpr bool SEED = TUBE && !SMALLGRID;
            if constexpr (!SEED) {
                best_w[lane] = 0x7FF0000000000000ull;                   // f64_key(+inf): closest = infinity
                bidx_w[lane] = 0u;                                      // sphere index + 1; 0 = none
            }
            uint32_t pool_n = 0, pool_done = 0;                         // wave-uniform
            auto pool_round = [&]() {
                RT_STAMP(10);
                RT_COUNT(5);
                const uint32_t e_i = pool_done + (uint32_t)lane;
                const bool act = e_i < pool_n;
                const uint32_t e = act ? pool_w[e_i & 127u] : (uint32_t)lane;
                const int r = (int)(e & 63u), slot = (int)(e >> 6);         // (26 bits of column: 64 M columns)
                // MODE 5 numbers candidates by their column in the table (spheres are tiled by position); everything
                // the reference decides by a sphere's place in the list uses idx, the place in the caller's list
                const int idx = TUBE ? (int)P.slot_orig[slot] : slot;
                const double *__restrict__ pgeo = TUBE ? P.geo_slot : geo;
                const int src = r << 2;
                const D3 ro = mk(from_lane_f64(src, o.x), from_lane_f64(src, o.y), from_lane_f64(src, o.z));
                const D3 rd = mk(from_lane_f64(src, d.x), from_lane_f64(src, d.y), from_lane_f64(src, d.z));
                bool has_root = false;
                unsigned long long key = 0ull;
                if (act) {
                    if (DIAG) n_cand++;
                    // sphere.rs:16-34, exactly as exact_test() computes it
                    const double4 g = *reinterpret_cast<const double4 *>(pgeo + 4 * (size_t)slot);
                    const double ra = length_squared(rd);
                    const D3 oc = ro - mk(g.x, g.y, g.z);
                    const double half_b = dot(oc, rd);
                    const double c = length_squared(oc) - g.w;
                    const double disc = half_b * half_b - ra * c;
                    if (!(disc < 0.0) && !(half_b > 0.0 && c > 0.0)) {
                        if (DIAG) n_roots++;
                        const double sqrtd = __builtin_sqrt(disc);
                        double root = (-half_b - sqrtd) / ra;
                        if (root < t_min) root = (-half_b + sqrtd) / ra;
                        if (!(root < t_min)) {
                            // (a NaN root maps above +inf and never wins, as in exact_test(); root >= t_min > 0 otherwise)
                            key = f64_key(root);
                            has_root = true;
                        }
                    }
                }
                // Three phases, each by all lanes before the next begins (LDS operations of one wave execute in program
                // order; the wave barriers keep the compiler from moving one phase's LDS operations into another's):
                // (1) every root lowers its ray's minimum; (2) a lane that lowered it forgets the sphere recorded for the
                // old minimum; (3) the roots EQUAL to the final minimum record the largest sphere index among them --
                // the order-independent form of mod.rs:61-67 (smallest root, ties -> the later sphere of the list).
                unsigned long long old = 0ull;
                RT_LDS_N(4, 12, 4, false, false, src, true);                            // the twelve ds_bpermute of (o, d)
                RT_LDS(2, 8, true, false, &best_w[r], has_root);
                if (has_root) old = atomicMin(&best_w[r], key);
                __builtin_amdgcn_wave_barrier();
                RT_LDS(3, 4, false, false, &bidx_w[r], has_root && old > key);
                if (has_root && old > key) bidx_w[r] = 0u;
                __builtin_amdgcn_wave_barrier();
                RT_LDS(7, 8, false, true, &best_w[r], has_root);
                RT_LDS_POOL_TIE(has_root, r, key);
                if (has_root && best_w[r] == key) atomicMax(&bidx_w[r], (unsigned)idx + 1u);
                __builtin_amdgcn_wave_barrier();
                pool_done = min(pool_done + 64u, pool_n);
                RT_STAMP(2);
            };
            // owners push the candidates of segment seg0 (ascending sphere order); rounds run as the ring fills
            // MODE 5: the tiles this pass scans -- every tile of the table in turn, or the list behind the bitmap words
            bool list_all = true;
            auto enumerate = [&](int seg0, int nwords_tube = 0) {
                unsigned summary = 0u, word = 0u;
                if constexpr (TUBE) {
                    // which of the segment's words hold a candidate of this ray (the recording side sets bits only)
                    // (min(word, 1) << w | summary: a v_min_u32 and a v_lshl_or_b32 per word where `word != 0 ? 1 << w : 0` is a compare, a select and an or)
                    for (int w = 0; w < nwords_tube; ++w) summary |= min(bits_w[w * 64 + lane], 1u) << w;
                    if (!alive) summary = 0u;
                } else {
                    summary = alive ? s_sum[tid] : 0u;
                }
                int wbase = 0;
                for (;;) {
                    const bool has = (summary | word) != 0u;
                    const unsigned long long m = __ballot(has);
                    if (m == 0ull) break;
                    RT_COUNT_ENUM_TRIP(m);
                    RT_LDS_ENUM_READS(has, word, summary, seg0);
                    if (has) {
                        if (word == 0u) {
                            const int w = __builtin_ctz(summary);
                            summary &= summary - 1u;
                            word = bits_w[w * 64 + lane];
                            if constexpr (TUBE)         // seg0: position in the tile list of the segment's first tile
                                wbase = 32 * (!SMALLGRID && list_all ? seg0 + w : (int)(bits_w[(kSeg / 2) * 64 + seg0 + w] & 0x0FFFFFFFu));
                            else
                                wbase = 16 * seg0 + 32 * w;
                        }
                        const int bpos = __builtin_ctz(word);
                        word &= word - 1u;
                        const uint32_t pos = pool_n + rank_below(m);
                        pool_w[pos & 127u] = ((uint32_t)(wbase + bpos) << 6) | (uint32_t)lane;
                    }
                    pool_n += (uint32_t)__popcll(m);
                    if (pool_n - pool_done >= 64u) pool_round();
                }
            };
            // after the last segment: drain the ring, then every owner takes its minimum
            auto finish_pool = [&]() {
                RT_STAMP(10);
                while (pool_done < pool_n) pool_round();
                const unsigned int hb = bidx_w[lane];
                if constexpr (SEED) {
                    // (the seeded minimum IS the closest hit: equal roots took the maximum of the list indices, seed included --
                    //  mod.rs:61-67's later-sphere-wins; a ray outside the filter's range has no candidates, its seed, NaN roots
                    //  included, comes back bit for bit)
                    if (alive) { hit = (int)hb - 1; closest = key_f64(best_w[lane]); }
                } else
                if (alive && hb != 0u) {
                    const double root = key_f64(best_w[lane]);
                    const int idx = (int)hb - 1;
                    if (root < closest || (root == closest && idx > hit)) { closest = root; hit = idx; }
                }
            };
            const int nt = P.n_tiles;                   // even; the tables hold nt + 2 tiles

            if constexpr (TUBE) {
                // The record of the first always-exact sphere (the ground) is asked for HERE and used after the filter rows have been
                // built: ~130 vector instructions that need nothing from memory run under its latency (-0.5 % frame time).
                const double4 g_first = *reinterpret_cast<const double4 *>(geo + 4 * (size_t)(P.n_always > 0 ? P.always_idx[0] : 0));
                const bool scan = alive;
                const unsigned long long scan_mask = __ballot(scan);
                if (scan_mask != 0ull) {
                float ray_of[3], ray_df[3], ray_o1;                 // (the ray in f32, shared by the filter rows and the footprint)
                ray_f32(o, d, ray_of, ray_df, ray_o1);
                // (rows of every lane as if it had a ray inside the analysed range; the lanes without one -- none in the steady state: a lane
                //  whose path ends starts its next sample at once -- and the rays outside it get rows that keep nothing behind ONE wave-level
                //  branch: 16 selects per pass less)
                TubeRay T = make_tube<false>(ray_of, ray_df, ray_o1, P.tube_rho);
                if (!scan) T.sane = true;
                if (__builtin_expect(__ballot(!scan || !T.sane) != 0ull, 0)) {
                    if (!scan || !T.sane) tube_rows_keep_nothing(T);
                }
                bf16x8 A[4];
                {
                    uint32_t w[2][8];
                    tube_a_words(T, w);
                    tube_stage_operands(reinterpret_cast<uint4 *>(bits_w), lane, w, A);
                }
                // the spheres that skip the filter (the ground): tested exactly by every ray
                if (alive)
                    for (int e = 0; e < P.n_always; ++e) {
                        if (DIAG) n_cand++;
                        if (e == 0) exact_test_g(P.always_idx[0], g_first, std::false_type{});
                        else exact_test(P.always_idx[e], std::false_type{});
                    }
                // outside the analysed range (zero, NaN and infinite directions are): HittableList::hit as written, over
                // the whole list in list order (what the always-exact list and the pool find for this ray is a subset of it)
                if (scan && !T.sane) {
                    closest = __builtin_inf(); hit = -1;
                    for (int i = 0; i < n; ++i) { if (DIAG) n_cand++; exact_test(i, std::true_type{}); }
                }
                if constexpr (SEED) {
                    best_w[lane] = f64_key(closest);                    // (+inf without a hit: 0x7FF0000000000000)
                    bidx_w[lane] = (unsigned)(hit + 1);                 // sphere index + 1; 0 = none
                }
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                const int ntt = nt >> 1;                    // tiles of 32 spheres; the tables hold ntt + 1
                const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint4 *>(P.btube), 0, (ntt + 1) * 1024, 0x00020000);
                const int voff = lane * 16;
                auto load_b = [&](int t32) -> bf16x8 {
                    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, voff, t32 * 1024, 0));
                };
                const int col32 = lane & 31, hh = lane >> 5;
                const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                // results of one MFMA: acc[8bb + j] / acc[8bb + 4 + j] are H_1 / H_2 of ray 16G + 8bb + 4hh + j against
                // sphere 32 t + col32, in units of half the sphere's bound (the host scales every column): the pair is
                // kept iff |H_1| < 2 and |H_2| < 2, i.e. iff bit 30 of BOTH f32 patterns is clear (biased exponent
                // < 128; infinities and NaNs have it set).  So the look is bit logic: X_bb = AND over the lane's four rays
                // 8 bb + j of (H_1 | H_2) -- one v_or_b32 and three v_bitop3_b32 (a & (b | c)) per half -- has bit 30
                // clear iff one of them keeps this sphere, and one compare per half, |X_bb| < 2.0, reads that bit: the OR of the two
                // wave masks feeds the wave-level branch, each mask its half of the recording path.  10 vector instructions of
                // 2.7-4.9 SIMD cycles where max/min-trees took 14 of 4.3-5.4 (tools/valu_cost_table.hip: v_max/v_min*
                // cost 4.3-5.4 cycles at four waves per SIMD, v_or/v_bitop3 2.7-3.1).
                auto keeps = [](uint32_t x) -> bool { return __builtin_fabsf(__uint_as_float(x)) < kTubeKeepBelow; };
                auto look_tube = [&](int G, const f32x16 &acc, int wrel) {
                    // X_bb: the AND over the four rays 8 bb + j of this lane; the wave-level branch tests X_0 & X_1
                    uint32_t X[2];
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        X[bb] = __float_as_uint(acc[8 * bb]) | __float_as_uint(acc[8 * bb + 4]);
#pragma unroll
                        for (int j = 1; j < 4; ++j)
                            X[bb] = __builtin_amdgcn_bitop3_b32(X[bb], __float_as_uint(acc[8 * bb + j]), __float_as_uint(acc[8 * bb + 4 + j]), 0xE0);
                    }
                    // one wave-level test per HALF, made once: the branch below is taken when either half keeps something, and the halves'
                    // masks are what the recording path branches on (round 4 tested X_0 & X_1 here -- one compare -- and each half again behind
                    // the branch, which two thirds of the looks take: -0.3 % on the book scene, -2 % on the 10k-sphere scene)
                    const unsigned long long kh[2] = {__ballot(keeps(X[0])), __ballot(keeps(X[1]))};
                    if (__builtin_expect((kh[0] | kh[1]) != 0ull, 0)) {
                        RT_COUNT(3);
                        int colv = col32;
                        asm volatile("" : "+v"(colv));              // keep the address arithmetic on this side of the branch
                        const unsigned bit = 1u << colv;
                        int rbase = wrel * 64 + 4 * hh;             // (4 hh folded into the row's base, not into each word's index: -0.3 %)
                        unsigned int *row = bits_w + rbase;
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            if (kh[bb] != 0ull) {
                                RT_COUNT(4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    RT_LDS(0, 4, true, false, &row[16 * G + 8 * bb + j],
                                           keeps(__float_as_uint(acc[8 * bb + j]) | __float_as_uint(acc[8 * bb + 4 + j])));
                                    if (keeps(__float_as_uint(acc[8 * bb + j]) | __float_as_uint(acc[8 * bb + 4 + j])))
                                        atomicOr(&row[16 * G + 8 * bb + j], bit);
                                }
                            }
                        }
                    }
                };
                // ---- which tiles: the global ones and the grid cells some ray of the wave can reach ---------------
                // Every ray finds the cells of its footprint (rt_device.hpp, grid_cells: columns ix0.., rows iz0..); the
                // union over the wave becomes a list of tiles in LDS (tl[]: the global tiles, then the cells' tiles in
                // ascending order, then two spare entries for the loop's look-ahead).  If a footprint cannot be
                // computed, or the list is longer than kListCap, the wave scans every tile of the table instead.
                int n_list = ntt;
                unsigned int *tl = bits_w + (kSeg / 2) * 64;                // (behind the bitmap words)
                auto wave_or = [](int v) -> int {               // the OR over the wave (DPP; lane 63 collects it)
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 -> rows 1, 3
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2, 3
                    return __builtin_amdgcn_readlane(v, 63);
                };
                auto rows_or = [](int v) -> int {               // the OR over each row of 16 lanes (lane 16 g + 15 collects group g's)
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
                    return v;
                };
                auto wave_scan_add = [](int v) -> int {         // inclusive prefix sum over the lanes (DPP)
                    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
                    return v;
                };
                if (SMALLGRID || P.grid_dim > 0) {
                    int ix0 = 0, nx = 0, iz0 = 0, nz = 0, cnt = 0;
                    GridSeg seg;
                    RT_STAMP(5);
                    if (scan && T.sane) cnt = grid_cells(ray_of, ray_df, ray_o1, P.grid, P.grid_dim, P.scene_scale, ix0, nx, iz0, nz, SMALLGRID ? nullptr : &seg);
                    RT_STAMP(7);
                    const int gcells = P.grid_dim * P.grid_dim;
                    if (SMALLGRID || P.n_global + gcells <= 64) {
                        // a small grid: the cells fit ONE 64-bit mask per ray (bit iz * grid_dim + ix); the wave's set
                        // of cells is the OR over its lanes, taken in registers, and lane c finds the place of cell c's
                        // tile in the list by counting the set cells below it
                        unsigned long long m = cnt < 0 ? ~0ull : 0ull;          // (cannot tell: every cell)
                        if (cnt > 0) {
                            // the footprint's columns, repeated in each of its rows: the rows sit grid_dim bits apart and
                            // the run is narrower than that, so the product has no carries
                            const unsigned long long run = (unsigned long long)(((1u << nx) - 1u) << ix0);
                            const unsigned long long rows = P.grid_rows & (~0ull >> (64 - nz * P.grid_dim));
                            m = (run * rows) << (iz0 * P.grid_dim);
                        }
#if RT_GROUP_SKIP
                        // ... and per 16-ray GROUP: a DPP row is 16 lanes, so after the four row steps lane 16 g + 15 holds group g's
                        // cells; the wave's set is the OR of the four.  Each list entry carries, in its top four bits, which groups
                        // can reach the tile: the tile loop skips the matrix instruction and the look of the others.
                        // (skipping the high dword for grids of at most 32 cells -- the book scene's 4 x 4 -- measured SLOWER: 11.18 vs 11.10 ms)
                        const int vlo = rows_or((int)(unsigned)m), vhi = rows_or((int)(unsigned)(m >> 32));
                        unsigned mlo = 0u, mhi = 0u, gm = 0u;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(vlo, 16 * g + 15), hi = (unsigned)__builtin_amdgcn_readlane(vhi, 16 * g + 15);
                            mlo |= lo; mhi |= hi;
                            if (__builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo)) gm |= 1u << g;    // (lane c: is cell c in group g's set)
                        }
                        const unsigned kAll = 0xF0000000u;
#else
                        const unsigned mlo = (unsigned)wave_or((int)(unsigned)m), mhi = (unsigned)wave_or((int)(unsigned)(m >> 32));
                        const unsigned gm = 0u, kAll = 0u;
#endif
                        const unsigned long long cells = (((unsigned long long)mhi << 32) | mlo) & (~0ull >> (64 - gcells));
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                        tl[lane] = (unsigned)(lane < P.n_global ? lane : ntt) | kAll;      // (past the list: the spare tile)
                        if (lane < 2) tl[64 + lane] = (unsigned)ntt | kAll;
                        __builtin_amdgcn_wave_barrier();
                        if ((cells >> lane) & 1ull) tl[P.n_global + rank] = (unsigned)(P.n_global + lane) | (gm << 28);
                        __builtin_amdgcn_wave_barrier();
                        list_all = false;
                        n_list = P.n_global + __builtin_popcountll(cells);
                    } else if (!SMALLGRID && __ballot(cnt < 0) == 0ull) {
                        // a large grid: one word per grid ROW in LDS (32 bits for grids of at most 32 cells per side -- the 10k-sphere
                        // scene's 19 x 19 --, else 64: shifts, counts and bit scans of a 64-bit word are two or three instructions each),
                        // ORed by the rays; lane l then owns row l, a prefix sum over the rows' cell counts gives each row its place in
                        // the list.  One set of row words per 16-ray group (at most 2 KB of the bitmap area, free until the tile loop):
                        // each list entry carries in its top four bits which groups can reach the tile.
                        if (cnt == 0) nz = 0;
                        auto build_list = [&](auto zero) {
                            typedef decltype(zero) W;
                            constexpr int kGroups = RT_GROUP_SKIP ? 4 : 1;
                            W *tm = reinterpret_cast<W *>(bits_w);                          // [kGroups][64]
#pragma unroll
                            for (int g = 0; g < kGroups; ++g) tm[g * 64 + lane] = (W)0;
                            int lg_ = lane;
                            asm volatile("" : "+v"(lg_));           // (computed here, per pass: hoisted out of the bounce loop this address was spilled to scratch memory -- a vector-memory
                                                                    //  round trip and an s_waitcnt vmcnt(0) per pass; 10k-sphere scene -0.65 %)
                            W *tm_g = tm + (RT_GROUP_SKIP ? (lg_ >> 4) * 64 : 0);
                            __builtin_amdgcn_wave_barrier();
                            // row by row: the columns of the part of the clipped piece that lies in the row's band (grid_row_run)
                            for (int k = 0; __any(k < nz); ++k) {
                                RT_COUNT_ROWS_TRIP(1);
                                if (k < nz) {
                                    int rx0, rnx;                                           // 1 <= rnx, rnx + rx0 <= grid_dim <= 32 or 63
                                    grid_row_run(seg, ix0, ix0 + nx - 1, (float)(iz0 + k), rx0, rnx);
                                    atomicOr(&tm_g[iz0 + k], (W)((W)(~(W)0 >> (8 * (int)sizeof(W) - rnx)) << rx0));
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            W wg[kGroups];
                            W mw = (W)0;
#pragma unroll
                            for (int g = 0; g < kGroups; ++g) { wg[g] = tm[g * 64 + lane]; mw |= wg[g]; }
                            const int mine = sizeof(W) == 8 ? __builtin_popcountll(mw) : __builtin_popcount((unsigned)mw);
                            const int upto = wave_scan_add(mine);
                            const int tn = P.n_global + __builtin_amdgcn_readlane(upto, 63);
                            if (tn <= kListCap) {
                                const unsigned kAll = RT_GROUP_SKIP ? 0xF0000000u : 0u;
                                if (lane < P.n_global) tl[lane] = (unsigned)lane | kAll;
                                int pos = P.n_global + upto - mine;
                                const int row0 = P.n_global + lane * P.grid_dim;
                                while (__any(mw != (W)0)) {
                                    RT_COUNT_ROWS_TRIP(6);
                                    if (mw != (W)0) {
                                        const int bpos = sizeof(W) == 8 ? __builtin_ctzll(mw) : __builtin_ctz((unsigned)mw);
                                        unsigned gm = 0u;
#pragma unroll
                                        for (int g = 0; g < (RT_GROUP_SKIP ? 4 : 0); ++g) gm |= (unsigned)((wg[g] >> bpos) & (W)1) << g;
                                        tl[pos++] = (unsigned)(row0 + bpos) | (gm << 28);
                                        mw &= mw - (W)1;
                                    }
                                }
                                int ntt_ = ntt;
                                asm volatile("" : "+s"(ntt_));      // (likewise: the broadcast of this scalar sat in a spilled register; -0.6 %)
                                if (lane < 2) tl[tn + lane] = (unsigned)ntt_ | kAll;
                                __builtin_amdgcn_wave_barrier();
                                list_all = false;
                                n_list = tn;
                            }
                        };
                        if (P.grid_dim <= 32) build_list(0u); else build_list(0ull);
                    }
                }
                RT_STAMP(3);
                RT_STAMP(5);
                for (int t0 = 0; t0 < n_list; t0 += kSeg / 2) {
                    __builtin_amdgcn_s_setprio(3);
                    const int nwords = min(kSeg / 2, n_list - t0);          // one bitmap word per 32-sphere tile
                    for (int w = 0; w < nwords; ++w) bits_w[w * 64 + lane] = 0u;
                    // the segment's tiles (and two more for the look-ahead), one per lane
                    const int listv = !SMALLGRID && list_all ? (min(t0 + lane, ntt) | (RT_GROUP_SKIP ? (int)0xF0000000u : 0))
                                                             : (int)tl[min(t0 + lane, kListCap + 1)];
                    // j < 20, wave-uniform: the tile, and (top four bits) the 16-ray groups that can reach it
                    auto tile_at = [&](int j) -> int { return __builtin_amdgcn_readlane(listv, j) & 0x0FFFFFFF; };
                    auto groups_at = [&](int j) -> unsigned { return (unsigned)__builtin_amdgcn_readlane(listv, j) >> 28; };
                    __builtin_amdgcn_wave_barrier();
                    // one 32-sphere tile: four independent MFMAs (one per 16-ray group); two results in
                    // flight so the matrix pipe works on the next group while the VALU looks at this one
                    auto do_tile = [&](int w, const bf16x8 &b, unsigned groups) {
                        RT_COUNT(7);
                        RT_COUNT_N(2, __builtin_popcount(groups));      // (counter 2: matrix instructions + looks executed)
#if RT_GROUP_SKIP
                        if (groups != 0xFu) {
                            // some groups cannot reach this tile (none of their rays' footprints holds its cell): only the others
                            // go through the matrix pipe and the look
#pragma unroll
                            for (int G = 0; G < 4; ++G)
                                if ((groups >> G) & 1u) {
                                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[G], b, zero16, 0, 0, 0);
                                    look_tube(G, acc, w);
                                }
                            return;
                        }
#endif
                        // (the empty asm pins the issue order: the scheduler would otherwise sink each MFMA
                        //  below the previous look to share registers, and the wave would sit out the full
                        //  matrix-pipe latency four times per tile)
                        f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], b, zero16, 0, 0, 0);
                        f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], b, zero16, 0, 0, 0);
                        asm volatile("" : "+v"(acc1));
                        look_tube(0, acc0, w);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], b, zero16, 0, 0, 0);
                        asm volatile("" : "+v"(acc0));
                        look_tube(1, acc1, w);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[3], b, zero16, 0, 0, 0);
                        asm volatile("" : "+v"(acc1));
                        look_tube(2, acc0, w);
                        look_tube(3, acc1, w);
                    };
                    {
                        // B operands ping-pong between two register sets, each fetched a tile ahead.
                        bf16x8 bp = load_b(tile_at(0)), bq;
                        int w = 0;
                        for (; w + 1 < nwords; w += 2) {
                            bq = load_b(tile_at(w + 1));
                            do_tile(w, bp, groups_at(w));
                            bp = load_b(tile_at(w + 2));
                            do_tile(w + 1, bq, groups_at(w + 1));
                        }
                        if (w < nwords) do_tile(w, bp, groups_at(w));
                    }
                    __builtin_amdgcn_wave_barrier();
                    RT_STAMP(6);
                    __builtin_amdgcn_s_setprio(2);
                    enumerate(t0, nwords);
                }
                __builtin_amdgcn_s_setprio(1);
                finish_pool();
                __builtin_amdgcn_s_setprio(0);              // (the shading below)
                }   // scan_mask != 0
            }
#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_scan.inc"       // else if constexpr (LIFTED) { mode 4 } else { modes 2, 3 }
#endif
        }
        RT_STAMP(2);
        if (alive && !MATRIX) {
            if (MODE == 0) {
                for (int i = 0; i < n; ++i) { if (DIAG) n_cand++; exact_test(i, std::true_type{}); }
            } else {
                const RayFilter f = make_filter(o, d);
                int cnt = 0;
                auto drain = [&]() { test_list(cnt); cnt = 0; };
                // One filter test.  Candidates are rare per sphere (about one wave-test in ten
                // has any), so the push sits behind a wave-level branch that is normally not taken.
                auto test = [&](float cx, float cy, float cz, float kp, int i) {
                    const bool keep = filter_keeps(f, cx, cy, cz, kp);
                    if (__builtin_expect(__ballot(keep) != 0ull, 0)) {
                        if (keep) {
                            cand[cnt][tid] = (uint16_t)i;
                            cnt++;
                        }
                    }
                };
                // The filter records stream through SGPRs in batches of kScanUnroll spheres
                // (32 dwords, two s_load_dwordx16): one scalar-memory wait per ~100 VALU ops,
                // hidden by the other waves of the SIMD.
                constexpr int BW = 4 * kScanUnroll;
                const int nb = n / kScanUnroll;
                for (int bidx = 0; bidx < nb; ++bidx) {
                    const float __attribute__((address_space(4))) *q = filt + (size_t)BW * (size_t)bidx;
                    float rec[BW];
#pragma unroll
                    for (int k = 0; k < BW; ++k) rec[k] = q[k];
#pragma unroll
                    for (int k = 0; k < kScanUnroll; ++k)
                        test(rec[4 * k + 0], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3], bidx * kScanUnroll + k);
                    // drain before any lane could overflow its list
                    if (__any(cnt > kCandCap - kScanUnroll)) drain();
                }
                for (int i = nb * kScanUnroll; i < n; ++i) {
                    const float __attribute__((address_space(4))) *q = filt + 4 * (size_t)i;
                    test(q[0], q[1], q[2], q[3], i);
                }
                drain();
            }
        }

        if (DIAG) {   // fold this bounce's counts into the wave totals (all lanes are here)
            unsigned long long pk = (unsigned long long)n_cand | ((unsigned long long)n_roots << 32);
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) pk += __shfl_xor(pk, sh);      // both halves < 2^32: no carry across
            tot_cand += pk & 0xFFFFFFFFull; tot_roots += pk >> 32;
        }
        RT_STAMP(3);
        // ==== PHASE 4: shade ===================================================================================================
        // ---- (e) shade: main.rs:44-56 + materials.rs ----------------------------
        bool finished = false;                                          // this lane's sample ended in this pass
        D3 radiance = mk(0.0, 0.0, 0.0);
        if (alive) {
            bool done = false;
            D3 L = mk(0.0, 0.0, 0.0);
            const bool is_hit = hit >= 0;
            int kind = -1;
            bool front = false;
            D3 p = o, nrm = mk(0.0, 0.0, 0.0), sp = mk(0.0, 0.0, 0.0), albedo = mk(1.0, 1.0, 1.0);
            double param = 0.0, inv_param = 0.0, r0_front = 0.0, r0_back = 0.0;
            uint32_t w_first = 0u, w_second = 0u;
            if (is_hit) {
                const double *mrec = mat + kMatStride * (size_t)hit;
                const double4 g = *reinterpret_cast<const double4 *>(geo + 4 * (size_t)hit);
                const double2 mA = *reinterpret_cast<const double2 *>(mrec);          // 1/r, param
                const double2 mB = *reinterpret_cast<const double2 *>(mrec + 2);      // albedo r, g
                const double2 mC = *reinterpret_cast<const double2 *>(mrec + 4);      // albedo b, kind
                // (the Dialectric's constants with the rest of the record -- one round trip, not two when the sphere turns out to be glass: -0.25 %)
                const double2 mD_ = *reinterpret_cast<const double2 *>(mrec + 6);     // 1/ir, r0(1/ir)
                const double2 mE_ = *reinterpret_cast<const double2 *>(mrec + 8);     // r0(ir), -
                // One Philox block for every lane with a hit: the first unit-sphere try of a Lambertian/Metal lane, and -- computed
                // ahead, consumed (ev++) only if the draw is really made -- the Dialectric's reflectance draw.  One wave-level call,
                // not two; and it runs HERE, between asking for the sphere's records and using them: it needs nothing from memory,
                // so its ~160 issue cycles pass under the loads' latency (-0.45 % frame time; the empty asm pins the order).
                U4 w = philox4x32_10(pix_global, (uint32_t)s, ev, 0u, P.k0, P.k1);
                asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
                kind = (int)mC.y;
                param = mA.y;
                albedo = mk(mB.x, mB.y, mC.x);
                // sphere.rs:36-37 + mod.rs:20-30
                p = o + d * closest;                                             // ray.rs:15-17
                const D3 outward = (p - mk(g.x, g.y, g.z)) * mA.x;               // / radius = * (1/radius)
                front = dot(d, outward) < 0.0;
                nrm = front ? outward : (mk(0.0, 0.0, 0.0) - outward);
                w_first = w.x;
                if constexpr (U53) w_second = w.y;
                RT_STAMP(11);
                if constexpr (U53) {
                  if (kind != RT_KIND_DIALECTRIC) {
                    // vec3.rs:37-45 with 53-bit draws: a try is SIX consecutive words -- two tries per three blocks -- and the
                    // test is the reference's own f64 expression x*x + y*y + z*z < 1.0 (vec3.rs:87-89)
                    U4 b1 = philox4x32_10(pix_global, (uint32_t)s, ev + 1u, 0u, P.k0, P.k1);
                    uint32_t nblk = 2u, c0 = b1.z, c1 = b1.w;
                    double sx = u11_53(w.x, w.y), sy = u11_53(w.z, w.w), sz = u11_53(b1.x, b1.y);
                    bool ok = sx * sx + sy * sy + sz * sz < 1.0;
                    while (!ok) {
                        RT_COUNT_MAIN(6);
                        U4 b = philox4x32_10(pix_global, (uint32_t)s, ev + nblk, 0u, P.k0, P.k1);
                        nblk++;
                        sx = u11_53(c0, c1); sy = u11_53(b.x, b.y); sz = u11_53(b.z, b.w);       // try 2m+1: ends its block
                        ok = sx * sx + sy * sy + sz * sz < 1.0;
                        if (!ok) {
                            b = philox4x32_10(pix_global, (uint32_t)s, ev + nblk, 0u, P.k0, P.k1);
                            b1 = philox4x32_10(pix_global, (uint32_t)s, ev + nblk + 1u, 0u, P.k0, P.k1);
                            nblk += 2u;
                            sx = u11_53(b.x, b.y); sy = u11_53(b.z, b.w); sz = u11_53(b1.x, b1.y); // try 2m+2 = try 0 of the next three blocks
                            c0 = b1.z; c1 = b1.w;
                            ok = sx * sx + sy * sy + sz * sz < 1.0;
                        }
                    }
                    ev += nblk;
                    sp = mk(sx, sy, sz);
                  } else {
                    inv_param = mD_.x; r0_front = mD_.y; r0_back = mE_.x;
                  }
                } else
                if (kind != RT_KIND_DIALECTRIC) {
                    // vec3.rs:37-45: redraw until |p|^2 < 1 -- on the integers behind the three uniforms, where the
                    // reference's f64 comparison is exact (see unit_sphere_accepts); converted once, after the loop.
                    // The tries of one scatter take consecutive words of consecutive blocks:
                    // four tries per three blocks.
                    uint32_t tx = w.x, ty = w.y, tz = w.z, nblk = 1u;
                    bool ok = unit_sphere_accepts(tx, ty, tz);
                    RT_COUNT_REDRAW_LANES(ok);                                   // (diagnostic builds: lanes that draw a unit-sphere sample / that fail try 0)
                    if (!ok) {
                        uint32_t c0 = w.w;                                       // the word left over from the block before
                        do {
                            RT_COUNT_MAIN(6);
                            U4 b = philox4x32_10(pix_global, (uint32_t)s, ev + nblk, 0u, P.k0, P.k1);
                            nblk++;
                            tx = c0; ty = b.x; tz = b.y;                         // try 4m+1
                            ok = unit_sphere_accepts(tx, ty, tz);
                            if (!ok) {
                                const uint32_t c1 = b.z, c2 = b.w;
                                b = philox4x32_10(pix_global, (uint32_t)s, ev + nblk, 0u, P.k0, P.k1);
                                nblk++;
                                tx = c1; ty = c2; tz = b.x;                      // try 4m+2
                                ok = unit_sphere_accepts(tx, ty, tz);
                                if (!ok) {
                                    tx = b.y; ty = b.z; tz = b.w;                // try 4m+3: no new block
                                    ok = unit_sphere_accepts(tx, ty, tz);
                                    if (!ok) {
                                        b = philox4x32_10(pix_global, (uint32_t)s, ev + nblk, 0u, P.k0, P.k1);
                                        nblk++;
                                        tx = b.x; ty = b.y; tz = b.z; c0 = b.w;  // try 4m+4 = try 0 of the next three blocks
                                        ok = unit_sphere_accepts(tx, ty, tz);
                                    }
                                }
                            }
                        } while (!ok);
                    }
                    ev += nblk;
                    sp = mk(u11(tx), u11(ty), u11(tz));
                } else {
                    inv_param = mD_.x; r0_front = mD_.y; r0_back = mE_.x;
                }
            }
            // Every branch normalises exactly one vector (vec3.rs:107-109: v * (1/sqrt(v.v))): the
            // sky and Dialectric take unit(d), Lambertian unit(sample), Metal unit(reflect(d,n)).
            // One shared f64 sqrt + divide for the whole wave instead of one per branch.
            RT_STAMP(12);
            D3 V = d;
            if (kind == RT_KIND_LAMBERTIAN) V = sp;
            if (kind == RT_KIND_METAL) V = reflect(d, nrm);
            const D3 uV = unit_vector(V);
            if (!is_hit) {
                const double t = 0.5 * (uV.y + 1.0);                             // main.rs:54-55
                const D3 sky = mk(1.0, 1.0, 1.0) * (1.0 - t) + mk(0.5, 0.7, 1.0) * t;
                L = mk(s_thr[0][tid], s_thr[1][tid], s_thr[2][tid]) * sky;       // contract C3
                done = true;
            } else {
                D3 ndir;
                if (kind == RT_KIND_LAMBERTIAN) {                                // materials.rs:21-31
                    ndir = nrm + uV;
                    const double eps = 1e-8;
                    // (vec3.rs:111-114 is_near_zero: three compares, and the six selects only behind a wave-level branch that is all but never taken)
                    const bool near_zero = __builtin_fabs(ndir.x) < eps && __builtin_fabs(ndir.y) < eps && __builtin_fabs(ndir.z) < eps;
                    if (__builtin_expect(__ballot(near_zero) != 0ull, 0)) {
                        if (near_zero) ndir = nrm;
                    }
                } else if (kind == RT_KIND_METAL) {                              // materials.rs:48-62
                    ndir = uV + sp * param;
                    if (dot(ndir, nrm) <= 0.0) done = true;                      // absorbed: L = 0
                } else {                                                         // materials.rs:76-105
                    // 1.0/ir and ((1-ri)/(1+ri))^2 for both ratios are per-sphere constants: the same
                    // f64 operations, done once on the host (rt_api.hip)
                    const double ratio = front ? inv_param : param;
                    const double cos_theta = min_1(-dot(uV, nrm));
                    const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
                    bool do_refract = false;
                    if (ratio * sin_theta <= 1.0) {                              // && short-circuit
                        const double r0 = front ? r0_front : r0_back;            // materials.rs:79
                        const double x = 1.0 - cos_theta;
                        const double x2 = x * x;
                        const double refl = r0 + (1.0 - r0) * ((x2 * x2) * x);   // materials.rs:80
                        ev++;                                                    // the block drawn above
                        do_refract = refl <= (U53 ? u01_53(w_first, w_second) : u01(w_first));
                    }
                    ndir = do_refract ? refract(uV, nrm, ratio) : reflect(uV, nrm);
                }
                if (kind != RT_KIND_DIALECTRIC) {                                // Dialectric: (1,1,1), x * 1.0 == x
                    s_thr[0][tid] = s_thr[0][tid] * albedo.x;
                    s_thr[1][tid] = s_thr[1][tid] * albedo.y;
                    s_thr[2][tid] = s_thr[2][tid] * albedo.z;
                }
                o = p;
                d = ndir;
                depth -= 1;
                if (depth <= 0) done = true;                                    // main.rs:40-42: L = 0
            }
            if (done) {
                alive = false;
                finished = true;
                radiance = L;
            }
        }
        RT_STAMP(13);
        __builtin_amdgcn_s_setprio(1);
        // ==== PHASE 5: accumulate_finished =====================================================================================
        // ---- (f) finished samples -> their block's sums (LDS) or, without a ring entry, the frame buffer ----
        {
            // age of this lane's block among the wave's blocks (0 = the current one); blocks of age >= kRingDepth have
            // lost their ring entry (sequence numbers compare modulo 2^28: the ring is 4 deep)
            const uint32_t age = (blk_seq - (my_blk >> 4)) & 0x0FFFFFFFu;
            const bool direct = finished && !(P.use_ring && age < (uint32_t)kRingDepth);
            const bool ringed = finished && !direct;
            RT_LDS_RING_ADDS(ringed, my_blk);
            if (finished) {
                const unsigned long long q0 = quantize(radiance.x), q1 = quantize(radiance.y), q2 = quantize(radiance.z);
                if (ringed) {
                    unsigned long long *acc = ring_w + ((my_blk >> 4) & (uint32_t)(kRingDepth - 1)) * (kRingSlots * 3) + (my_blk & 15u) * 3u;
                    atomicAdd(acc + 0, q0); atomicAdd(acc + 1, q1); atomicAdd(acc + 2, q2);
                } else {                                            // too few samples per pixel for block sums, or an orphan of a long-gone block
                    unsigned long long *px = P.fix + (size_t)pix_local * 3u;
                    atomicAdd(px + 0, q0); atomicAdd(px + 1, q1); atomicAdd(px + 2, q2);
                }
            }
            n_direct += (uint32_t)__popcll(__ballot(direct));
            unsigned long long fm = __ballot(ringed);
            if (fm != 0ull) {
                // count the finished samples per block (nearly always the current and the previous one), and write
                // out the blocks that have just received their last sample (LDS operations of a wave execute in
                // program order: the adds above are in the sums flush_ring reads)
                __builtin_amdgcn_wave_barrier();
                const unsigned long long m0 = __ballot(ringed && age == 0u);
                rem0 -= (uint32_t)__popcll(m0);
                if (m0 != 0ull && rem0 == 0u) flush_ring(blk_seq & (uint32_t)(kRingDepth - 1));
                fm &= ~m0;
                if constexpr (kRingDepth == 2) {
                    if (fm != 0ull) {                               // (the rest finished in the block before)
                        rem1 -= (uint32_t)__popcll(fm);
                        if (rem1 == 0u) flush_ring((blk_seq - 1u) & 1u);
                    }
                } else
                if (fm != 0ull) {
                    const unsigned long long m1 = __ballot(ringed && age == 1u);
                    rem1 -= (uint32_t)__popcll(m1);
                    if (m1 != 0ull && rem1 == 0u) flush_ring((blk_seq - 1u) & 3u);
                    fm &= ~m1;
                    if (fm != 0ull) {
                        const unsigned long long m2 = __ballot(ringed && age == 2u);
                        rem2 -= (uint32_t)__popcll(m2);
                        if (m2 != 0ull && rem2 == 0u) flush_ring((blk_seq - 2u) & 3u);
                        fm &= ~m2;
                        if (fm != 0ull) {
                            rem3 -= (uint32_t)__popcll(fm);
                            if (rem3 == 0u) flush_ring((blk_seq - 3u) & 3u);
                        }
                    }
                }
            }
        }
        n_samples += (uint32_t)__popcll(__ballot(finished));
        RT_STAMP(4);
    }
    // every block of this wave has finished its last sample and has been written out; a ring entry that
    // still holds something here would be a bookkeeping error -- its sums go out rather than get lost
    if (rem0 != 0u) flush_ring(blk_seq & (uint32_t)(kRingDepth - 1));
    if (rem1 != 0u) flush_ring((blk_seq - 1u) & (uint32_t)(kRingDepth - 1));
    if constexpr (kRingDepth == 4) {
        if (rem2 != 0u) flush_ring((blk_seq - 2u) & 3u);
        if (rem3 != 0u) flush_ring((blk_seq - 3u) & 3u);
    }

        // (end of the bounce loop body is stamped at the top of the next iteration as phase 4)
    // wave totals -> device counters
    RT_DIAG_FLUSH();
    {
        const unsigned long long nc = tot_cand, nr = tot_roots;
        if (DIAG) {
            const unsigned int nl = s_live[tid >> 6][lane];
            if (nl) atomicAdd(P.stats + 16 + lane, (unsigned long long)nl);
        }
        if (lane == 0) {
            atomicAdd(P.stats + 0, (unsigned long long)n_rays);
            atomicAdd(P.stats + 1, (unsigned long long)n_samples);
            if (n_direct) atomicAdd(P.stats + 4, (unsigned long long)n_direct);
            atomicAdd(P.stats + 2, nc);
            atomicAdd(P.stats + 3, nr);
        }
    }
}

// exact sum -> f64 value (one rounding above 2^53)
__device__ __forceinline__ double fix_to_f64(unsigned long long q)
{
    return ((double)(uint32_t)(q >> 32) * 4294967296.0 + (double)(uint32_t)q) * (1.0 / 4294967296.0);
}

__global__ void fix_to_f32_kernel(const unsigned long long *__restrict__ fix, float *__restrict__ out, long long count)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; k < count; k += stride) out[k] = (float)fix_to_f64(fix[k]);
}

// Color::to_rgba, vec3.rs:403-421, in f64, + the row flip of main.rs:141-145.
__device__ __forceinline__ uint8_t as_u8(double x)
{   // Rust `as u8`: saturating, NaN -> 0
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}
__device__ __forceinline__ double clamp_r(double x, double lo, double hi)
{   // f64::clamp: NaN stays NaN
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
__global__ void resolve_rgba8_kernel(const unsigned long long *__restrict__ fix, uint8_t *__restrict__ out,
                                     int width, int rows, double scale, int flip)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npix = (long long)width * rows;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; k < npix; k += stride) {
        const int r = (int)(k / width);
        const int i = (int)(k - (long long)r * width);
        const int dst = flip ? rows - 1 - r : r;
        const unsigned long long *q = fix + k * 3;
        uchar4 px;
        px.x = as_u8(256.0 * clamp_r(__builtin_sqrt(scale * fix_to_f64(q[0])), 0.0, 0.999));
        px.y = as_u8(256.0 * clamp_r(__builtin_sqrt(scale * fix_to_f64(q[1])), 0.0, 0.999));
        px.z = as_u8(256.0 * clamp_r(__builtin_sqrt(scale * fix_to_f64(q[2])), 0.0, 0.999));
        px.w = 255;
        reinterpret_cast<uchar4 *>(out)[(long long)dst * width + i] = px;
    }
}

__global__ void philox_kat_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t k0, uint32_t k1, uint32_t *out)
{
    const U4 r = philox4x32_10(c0, c1, c2, c3, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}


// One tile of the MODE 5 (tube) filter exactly as the render kernel evaluates it: 64 rays against
// the 32 columns of `tile`.  h_out[ray][column][k] = lambda u_k.(c - o) as the matrix pipe returns it;
// rows_out[ray] = (lambda u_1, lambda u_2, t_1, t_2, sane).
__global__ __launch_bounds__(64) void tube_products_kernel(const double *o, const double *d, const uint4 *tile, float rho,
                                                         float *h_out, float *rows_out)
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ uint4 stage[64 * 4];
    const int lane = threadIdx.x, col = lane & 31, hh = lane >> 5;
    const TubeRay T = make_tube(mk(o[3 * lane], o[3 * lane + 1], o[3 * lane + 2]),
                                mk(d[3 * lane], d[3 * lane + 1], d[3 * lane + 2]), rho);
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 3; ++i) rows_out[lane * 9 + 3 * k + i] = T.u[k][i];
    rows_out[lane * 9 + 6] = T.t[0]; rows_out[lane * 9 + 7] = T.t[1]; rows_out[lane * 9 + 8] = T.sane ? 1.0f : 0.0f;
    bf16x8 A[4];
    uint32_t w[2][8];
    tube_a_words(T, w);
    tube_stage_operands(stage, lane, w, A);
    const bf16x8 b = __builtin_bit_cast(bf16x8, tile[lane]);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int G = 0; G < 4; ++G) {
        const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[G], b, zero16, 0, 0, 0);
        for (int bb = 0; bb < 2; ++bb)
            for (int j = 0; j < 4; ++j) {
                const int ray = 16 * G + 8 * bb + 4 * hh + j;
                h_out[(ray * 32 + col) * 2 + 0] = acc[8 * bb + j];
                h_out[(ray * 32 + col) * 2 + 1] = acc[8 * bb + 4 + j];
            }
    }
}

__global__ void f64_div_sqrt_kernel(const double *a, const double *b, int n, double *q, double *r)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { q[k] = a[k] / b[k]; r[k] = __builtin_sqrt(a[k]); }
}

// known-answer hook: the kernel's own quantize() (contract C5) on n radiance values
__global__ void quantize_kernel(const double *x, int n, unsigned long long *q)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) q[k] = quantize(x[k]);
}

// known-answer hook: the kernel's own rejection tests and word -> uniform conversions on n triples of Philox words.
// acc[k] = unit_sphere_accepts(w) | unit_disk_accepts(w.x, w.y) << 1; uni[k] = (u01(w.x), u11(w.x), u11(w.y), u11(w.z))
__global__ void unit_accept_kernel(const uint32_t *w, int n, uint32_t *acc, double *uni)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const uint32_t x = w[3 * k], y = w[3 * k + 1], z = w[3 * k + 2];
        acc[k] = (unit_sphere_accepts(x, y, z) ? 1u : 0u) | (unit_disk_accepts(x, y) ? 2u : 0u);
        uni[4 * k] = u01(x); uni[4 * k + 1] = u11(x); uni[4 * k + 2] = u11(y); uni[4 * k + 3] = u11(z);
    }
}

} // namespace rt

#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_kernels.hpp"    // known-answer kernels of modes 2-4
#endif

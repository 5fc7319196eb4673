// rt_diag.hpp -- the diagnostic layer of the render megakernel.  NOTHING here is part of the shipped library: every macro below is
// empty unless ONE of the diagnostic builds is selected (tools/build_diag_libs.sh, tools/lds_conflicts.sh):
//
//   -DRT_PHASE_STAMPS    wave-time per phase of the bounce loop (s_memtime between phases), summed into stats[8..15] and
//                        stats[80..87]; the stamps serialise the phases: read the SHARES (tools/phase_shares.py)
//   -DRT_BLOCK_COUNTS    wave-level execution counts of the main blocks into stats[8..15] (tools/block_counts.py); with
//                        -DRT_COUNT_ROWS counters 1 and 6 count the large grid's footprint-row and list-emission trips, with
//                        -DRT_COUNT_ENUM the enumeration's trips and the candidates it pushes, with -DRT_COUNT_REDRAW the lanes that draw a unit-sphere
//                        sample and those that fail try 0, instead of refills and redraw trips
//   -DRT_EXIT_TIMES      when the first / last / average wave leaves the kernel (100 MHz real-time clock; tools/exit_times.py)
//   -DRT_LDS_CONFLICTS   a software model of the LDS bank serialisation at every LDS site of the kernel (tools/lds_conflicts.py)
//
// The kernel source carries the call sites (RT_STAMP(k), RT_COUNT(k), RT_LDS(...), ...) and two hooks, RT_DIAG_DECLARE() at the top of
// render_kernel and RT_DIAG_FLUSH() at its end; the names the macros use (tid, lane, P, ring_w, quv_w, qid_w, bits_w, bidx_w, best_w,
// ...) are the kernel's own locals at those sites.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if (defined(RT_PHASE_STAMPS) + defined(RT_BLOCK_COUNTS) + defined(RT_EXIT_TIMES) + defined(RT_LDS_CONFLICTS)) > 1
#error "one diagnostic build at a time: RT_PHASE_STAMPS, RT_BLOCK_COUNTS, RT_EXIT_TIMES or RT_LDS_CONFLICTS"
#endif

#define RT_DIAG_NOTHING do { } while (0)

// ---- defaults: the shipped library ------------------------------------------------------------------------------------------------
#define RT_DIAG_DECLARE() RT_DIAG_NOTHING
#define RT_DIAG_FLUSH() RT_DIAG_NOTHING
#define RT_STAMP(k) RT_DIAG_NOTHING
#define RT_COUNT(k) RT_DIAG_NOTHING
#define RT_COUNT_N(k, n) RT_DIAG_NOTHING
#define RT_COUNT_MAIN(k) RT_DIAG_NOTHING                /* counters 1 / 6 in their default meaning (refills, redraw trips) */
#define RT_COUNT_ROWS_TRIP(k) RT_DIAG_NOTHING           /* ... as the large grid's footprint-row / list-emission trips (-DRT_COUNT_ROWS) */
#define RT_COUNT_ENUM_TRIP(m) RT_DIAG_NOTHING           /* ... as the enumeration's trips and pushed candidates (-DRT_COUNT_ENUM) */
#define RT_COUNT_REDRAW_LANES(ok) RT_DIAG_NOTHING       /* ... as the lanes that draw a unit-sphere sample and those that fail try 0 (-DRT_COUNT_REDRAW) */
#define RT_LDS(site, BYTES, ATOMIC, LOAD64, ptr, active) RT_DIAG_NOTHING
#define RT_LDS_N(site, n, BYTES, ATOMIC, LOAD64, byte_addr, active) RT_DIAG_NOTHING
#define RT_LDS_QUEUE_READS(take, entry) RT_DIAG_NOTHING
#define RT_LDS_POOL_TIE(has_root, r, key) RT_DIAG_NOTHING
#define RT_LDS_ENUM_READS(has, word, summary, seg0) RT_DIAG_NOTHING
#define RT_LDS_RING_ADDS(ringed, my_blk) RT_DIAG_NOTHING

// ---- RT_PHASE_STAMPS --------------------------------------------------------------------------------------------------------------
#if defined(RT_PHASE_STAMPS)
#undef RT_DIAG_DECLARE
#undef RT_DIAG_FLUSH
#undef RT_STAMP
#define RT_DIAG_DECLARE() unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime()
#define RT_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
                         __builtin_amdgcn_s_waitcnt(0); ph[k] += tn_ - tprev; tprev = tn_; } while (0)
#define RT_DIAG_FLUSH() do { if (lane == 0) { \
        for (int k_ = 0; k_ < 8; ++k_) atomicAdd(P.stats + 8 + k_, ph[k_]); \
        for (int k_ = 8; k_ < 16; ++k_) atomicAdd(P.stats + 80 + (k_ - 8), ph[k_]); } } while (0)     /* finer split: tools/phase_shares.py */
#endif

// ---- RT_EXIT_TIMES ----------------------------------------------------------------------------------------------------------------
#if defined(RT_EXIT_TIMES)
#undef RT_DIAG_DECLARE
#undef RT_DIAG_FLUSH
#define RT_DIAG_DECLARE() const unsigned long long t_wave_start = __builtin_amdgcn_s_memrealtime()
#define RT_DIAG_FLUSH() do { if (lane == 0) { \
        const unsigned long long te_ = __builtin_amdgcn_s_memrealtime(); \
        atomicMax(P.stats + 8, te_); atomicMax(P.stats + 9, ~te_); atomicAdd(P.stats + 10, te_); \
        atomicMax(P.stats + 11, ~t_wave_start); atomicAdd(P.stats + 12, 1ull); } } while (0)
#endif

// ---- RT_BLOCK_COUNTS --------------------------------------------------------------------------------------------------------------
#if defined(RT_BLOCK_COUNTS)
#undef RT_DIAG_DECLARE
#undef RT_DIAG_FLUSH
#undef RT_COUNT
#undef RT_COUNT_N
#define RT_DIAG_DECLARE() __shared__ unsigned int s_cnt[kBlock / 64][8]; if ((tid & 63) < 8) s_cnt[tid >> 6][tid & 7] = 0u
#define RT_COUNT(k) do { if ((int)(tid & 63) == (int)__builtin_ctzll(__ballot(true))) s_cnt[tid >> 6][k] += 1u; } while (0)
#define RT_COUNT_N(k, n) do { if ((int)(tid & 63) == (int)__builtin_ctzll(__ballot(true))) s_cnt[tid >> 6][k] += (unsigned)(n); } while (0)
#define RT_DIAG_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(P.stats + 8 + k_, (unsigned long long)s_cnt[tid >> 6][k_]); } while (0)
#if defined(RT_COUNT_ROWS)
#undef RT_COUNT_ROWS_TRIP
#define RT_COUNT_ROWS_TRIP(k) RT_COUNT(k)
#elif defined(RT_COUNT_ENUM)
#undef RT_COUNT_ENUM_TRIP
#define RT_COUNT_ENUM_TRIP(m) do { RT_COUNT(1); RT_COUNT_N(6, __popcll(m)); } while (0)
#elif defined(RT_COUNT_REDRAW)
#undef RT_COUNT_REDRAW_LANES
#define RT_COUNT_REDRAW_LANES(ok) do { const unsigned n1_ = (unsigned)__popcll(__ballot(true)), n6_ = (unsigned)__popcll(__ballot(!(ok))); \
                                       RT_COUNT_N(1, n1_); RT_COUNT_N(6, n6_); } while (0)      /* (the ballots BEFORE RT_COUNT_N narrows to its leader lane) */
#else
#undef RT_COUNT_MAIN
#define RT_COUNT_MAIN(k) RT_COUNT(k)
#endif
#endif

// ---- RT_LDS_CONFLICTS -------------------------------------------------------------------------------------------------------------
#if defined(RT_LDS_CONFLICTS)
namespace rt {
// Diagnostic build only (tools/lds_conflicts.py): a software model of the LDS bank serialisation of ONE wave-level LDS
// instruction, per MI355X_MICROARCH.md section LDS: the lanes are served in fixed groups (4-byte accesses: 2 x 32 lanes, 32
// banks; 8-byte stores and atomics: 4 x 16 lanes, 16 bank pairs; 8-byte loads: 2 x 32 lanes, 32 bank pairs), identical
// addresses broadcast for a load, and every further distinct address on a busy bank costs one more LDS cycle; an atomic
// serialises same-address lanes too.  Returns the EXTRA cycles of the instruction (what SQ_LDS_BANK_CONFLICT counts), wave-uniform.
template <int BYTES, bool ATOMIC, bool LOAD64 = false>
__device__ __noinline__ uint32_t lds_extra_cycles(uint32_t byte_addr, bool active)
{
    constexpr int G = (BYTES == 8 && !LOAD64) ? 16 : 32;            // lanes per group
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(active);
    const uint32_t unit = byte_addr / (uint32_t)BYTES;              // address in access units
    const uint32_t bank = unit % (uint32_t)G;                       // (32 dword banks = 16 or 32 units of this size: one unit per lane of a group)
    bool first = active;                                            // the first lane of its group with this address
    for (int k = 0; k < 64; ++k) {
        const uint32_t uk = (uint32_t)__builtin_amdgcn_readlane((int)unit, k);
        if (((m >> k) & 1ull) && k < lane && k / G == lane / G && uk == unit) first = false;
    }
    const unsigned long long counted = ATOMIC ? m : __ballot(first);
    uint32_t cnt = 0;                                               // accesses the bank of this lane has to serve one after the other
    for (int k = 0; k < 64; ++k) {
        const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)bank, k);
        if (((counted >> k) & 1ull) && k / G == lane / G && bk == bank) cnt++;
    }
    if (!active) cnt = 0;
    uint32_t extra = 0;
    for (int g = 0; g < 64 / G; ++g) {
        uint32_t mx = 0;
        for (int k = g * G; k < (g + 1) * G; ++k) mx = max(mx, (uint32_t)__builtin_amdgcn_readlane((int)cnt, k));
        extra += mx > 1u ? mx - 1u : 0u;
    }
    return extra;
}
} // namespace rt
#undef RT_DIAG_DECLARE
#undef RT_DIAG_FLUSH
#undef RT_LDS
#undef RT_LDS_N
#undef RT_LDS_QUEUE_READS
#undef RT_LDS_POOL_TIE
#undef RT_LDS_ENUM_READS
#undef RT_LDS_RING_ADDS
// modelled extra LDS cycles per site, into stats[8..15]: 0 recording ds_or, 1 block-sum ds_add_u64, 2 pool ds_min_u64, 3 pool ds_max_u32 +
// reset, 4 ds_bpermute of the pool, 5 bitmap / tile-list reads of the enumeration, 6 sample-queue reads, 7 pool ring + per-ray result reads
#define RT_DIAG_DECLARE() unsigned long long lds_x[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RT_DIAG_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(P.stats + 8 + k_, lds_x[k_]); } while (0)
#define RT_LDS(site, BYTES, ATOMIC, LOAD64, ptr, active) do { lds_x[site] += rt::lds_extra_cycles<BYTES, ATOMIC, LOAD64>( \
        (uint32_t)(uintptr_t)(const void __attribute__((address_space(3))) *)(ptr), (active)); } while (0)
#define RT_LDS_N(site, n, BYTES, ATOMIC, LOAD64, byte_addr, active) do { lds_x[site] += (unsigned long long)(n) * \
        rt::lds_extra_cycles<BYTES, ATOMIC, LOAD64>((uint32_t)(byte_addr), (active)); } while (0)
#define RT_LDS_QUEUE_READS(take, entry) do { const bool take_ = (take); const uint32_t e_ = take_ ? (entry) : 0u; \
        RT_LDS(6, 8, false, true, &quv_w[e_], take_); RT_LDS(6, 8, false, true, &quv_w[64 + e_], take_); \
        for (int c_ = 0; c_ < 5; ++c_) RT_LDS(6, 4, false, false, &qid_w[c_ * 64 + e_], take_); } while (0)
#define RT_LDS_POOL_TIE(has_root, r, key) do { const bool eq_ = (has_root) && best_w[r] == (key); RT_LDS(3, 4, true, false, &bidx_w[r], eq_); } while (0)
#define RT_LDS_ENUM_READS(has, word, summary, seg0) do { if constexpr (TUBE) { \
        const bool need_ = (has) && (word) == 0u; \
        const int w_ = need_ ? __builtin_ctz(summary) : 0; \
        RT_LDS(5, 4, false, false, &bits_w[w_ * 64 + lane], need_); \
        if (SMALLGRID || !list_all) RT_LDS(5, 4, false, false, &bits_w[(kSeg / 2) * 64 + (seg0) + w_], need_); } } while (0)
#define RT_LDS_RING_ADDS(ringed, my_blk) do { if (__ballot(ringed) != 0ull) { \
        unsigned long long *acc_d = ring_w + (((my_blk) >> 4) & (uint32_t)(kRingDepth - 1)) * (kRingSlots * 3) + ((my_blk) & 15u) * 3u; \
        RT_LDS(1, 8, true, false, acc_d + 0, ringed); RT_LDS(1, 8, true, false, acc_d + 1, ringed); RT_LDS(1, 8, true, false, acc_d + 2, ringed); } } while (0)
#endif

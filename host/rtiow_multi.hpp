// rtiow_multi.hpp -- the reference's row parallelism (rayon over rows, src/main.rs:122-123, and the ordered
// collect() at :139) across GPUs, natively: one rt_context per device, each driven by its own host thread
// through the device-buffer form of the C ABI, rows dealt round-robin in tiles (rt_params.shard_index /
// shard_count), ONE RCCL ncclGather of the exact u64 sums to the first device over xGMI, rows put back in
// image order with ONE strided 2D device copy per shard, resolve (Color::to_rgba + flip) on the first device.
//
// Because the Philox counter is keyed by the global pixel and sample index and the sums are exact integers,
// the assembled frame equals the single-GPU frame bit for bit (tests/test_host_cpp.py).
#pragma once
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "rtiow_hip.h"

namespace rtiow {

struct ShardJob {
    int device = 0;
    rt_context *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint64_t *d_fix = nullptr;          // [pad_rows][W][3], this shard's rows in ascending j
    int rows = 0;
    rt_stats stats{};
    std::string err;
};

// Where shard k's compact rows go in the frame, as strided copies: `pieces` runs of `rows` rows each, run i from compact
// row src_row + i * src_pitch_rows to image row dst_row + i * dst_pitch_rows.  Shard k of n owns tiles k, k+n, ... of T
// rows; its local tile lt is tile k + lt n of the image, so its FULL tiles are equally strided in the frame: one entry
// (one 2D device copy) places them all, and only the image's ragged last tile (H not a multiple of T) is an entry of
// its own.  At most two entries per shard: n (+1) copy calls per frame instead of one per tile -- 8 instead of 4 320 for
// 7680x4320 at the default tile of one row on 8 GPUs.  (The inverse of rt_shard_row_index; tests/test_host_cpp.py.)
struct RowCopy { int dst_row, src_row, rows, pieces, dst_pitch_rows, src_pitch_rows; };
inline std::vector<RowCopy> reassembly_plan(int H, int T, int n, int k)
{
    std::vector<RowCopy> plan;
    const int ntiles = (H + T - 1) / T, full_tiles = H / T;             // tiles [0, full_tiles) hold T rows each
    const int mine = k < full_tiles ? (full_tiles - 1 - k) / n + 1 : 0; // full tiles k, k+n, ... < full_tiles
    if (mine > 0) plan.push_back({k * T, 0, T, mine, n * T, T});
    if (full_tiles < ntiles && full_tiles % n == k)                     // the ragged last tile is this shard's
        plan.push_back({full_tiles * T, (full_tiles / n) * T, H - full_tiles * T, 1, n * T, T});
    return plan;
}

// rgba_out: [height][width][4], top row first (flip applied, as main.rs:141-145 leaves it).
// Returns 0 on success; *err says what failed otherwise.
inline int render_sharded(const std::vector<int> &devices, bool force_rccl, const std::vector<rt_sphere> &flat,
                          const rt_camera &cam, const rt_params &base, uint8_t *rgba_out, rt_stats *stats_out,
                          std::string *err, int *copy_calls_out = nullptr)
{
    const int n = (int)devices.size();
    if (n == 0) { *err = "empty device list"; return 1; }
    const std::set<int> uniq(devices.begin(), devices.end());
    const bool distinct = (int)uniq.size() == n;
    const bool use_rccl = (distinct && n > 1) || (force_rccl && distinct);
    if (!distinct && uniq.size() != 1) { *err = "a device list with repeats must name ONE device (RCCL cannot mix)"; return 1; }
    const int W = base.width, H = base.height, T = base.tile_rows;
    const int ntiles = (H + T - 1) / T;
    const int pad_rows = ((ntiles + n - 1) / n) * T;                    // most rows any shard can own
    const size_t row_words = (size_t)W * 3, pad_words = (size_t)pad_rows * row_words;

    std::vector<ShardJob> jobs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    uint64_t *d_staging = nullptr;                                      // root: the gathered parts [n][pad_rows][W][3]
    uint64_t *d_full = nullptr;
    uint8_t *d_rgba = nullptr;
    // ONE exit path: whatever was created is released, on success and on every error return
    auto cleanup = [&]() {
        for (int k = 0; k < n; ++k) {
            ShardJob &J = jobs[k];
            if (J.ctx || J.d_fix || J.stream) (void)hipSetDevice(J.device);
            if (J.stream) { (void)hipStreamSynchronize(J.stream); }
            if (J.d_fix) (void)hipFree(J.d_fix);
            if (J.stream) (void)hipStreamDestroy(J.stream);
            if (J.ctx) rt_destroy(J.ctx);
            if (comms[k]) ncclCommDestroy(comms[k]);
            J.d_fix = nullptr; J.stream = nullptr; J.ctx = nullptr; comms[k] = nullptr;
        }
        if (d_full || d_rgba || d_staging) (void)hipSetDevice(devices[0]);
        (void)hipFree(d_full); (void)hipFree(d_rgba); (void)hipFree(d_staging);
        d_full = nullptr; d_rgba = nullptr; d_staging = nullptr;
    };
    auto bail = [&](const std::string &msg) { *err = msg; cleanup(); return 1; };
    auto hip_ok = [](hipError_t e, const char *what, std::string *msg) {
        if (e == hipSuccess) return true;
        *msg = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    std::string msg;
    if (use_rccl) {
        const ncclResult_t r = ncclCommInitAll(comms.data(), n, devices.data());
        if (r != ncclSuccess) return bail(std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    }
    if (!hip_ok(hipSetDevice(devices[0]), "hipSetDevice", &msg)) return bail(msg);
    if (use_rccl && !hip_ok(hipMalloc((void **)&d_staging, (size_t)n * pad_words * sizeof(uint64_t)), "hipMalloc staging", &msg)) return bail(msg);

    // ---- phase 1: every shard, on its own host thread: context, scene, buffers, the render launch (main.rs:122-136).
    // No collective in this phase: a shard that fails here (bad device, out of memory, a rejected scene) simply
    // reports, and no other shard is left waiting inside a gather for it.
    auto work = [&](int k) {
        ShardJob &J = jobs[k];
        J.device = devices[k];
        rt_params p = base;
        p.shard_index = k; p.shard_count = n;
        int rc = 0;
        auto fail = [&](const char *what) { J.err = std::string(what) + " (shard " + std::to_string(k) + "): " + rt_last_error(); };
        if ((rc = rt_create(J.device, &J.ctx))) return fail("rt_create");
        if ((rc = rt_upload_scene(J.ctx, flat.data(), (int32_t)flat.size()))) return fail("rt_upload_scene");
        int32_t rows = 0;
        if ((rc = rt_shard_rows(&p, &rows))) return fail("rt_shard_rows");
        J.rows = rows;
        if (!hip_ok(hipSetDevice(J.device), "hipSetDevice", &J.err)) return;
        if (!hip_ok(hipStreamCreateWithFlags(&J.stream, hipStreamNonBlocking), "hipStreamCreate", &J.err)) return;
        if (!hip_ok(hipMalloc((void **)&J.d_fix, pad_words * sizeof(uint64_t)), "hipMalloc", &J.err)) return;
        if (!hip_ok(hipMemsetAsync(J.d_fix, 0, pad_words * sizeof(uint64_t), J.stream), "hipMemsetAsync", &J.err)) return;
        if ((rc = rt_render_device(J.ctx, &cam, &p, J.d_fix, J.stream))) return fail("rt_render_device");
    };
    {
        std::vector<std::thread> threads;
        for (int k = 0; k < n; ++k) threads.emplace_back(work, k);
        for (auto &t : threads) t.join();
    }
    for (const ShardJob &J : jobs) if (!J.err.empty()) return bail(J.err);

    // ---- phase 2 (only when EVERY shard has launched): THE collective of the path -- every rank sends its padded
    // rows behind its render on its own stream, rank 0 receives all of them.  One thread drives all communicators
    // of this process, so the calls form one group (ncclGroupStart/End).
    if (use_rccl) {
        ncclResult_t r = ncclGroupStart();
        for (int k = 0; k < n && r == ncclSuccess; ++k)
            r = ncclGather(jobs[k].d_fix, d_staging, pad_words, ncclUint64, 0, comms[k], jobs[k].stream);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return bail(std::string("ncclGather: ") + ncclGetErrorString(r));
    }
    for (int k = 0; k < n; ++k) {
        ShardJob &J = jobs[k];
        if (!hip_ok(hipSetDevice(J.device), "hipSetDevice", &msg)) return bail(msg);
        if (!hip_ok(hipStreamSynchronize(J.stream), "hipStreamSynchronize", &msg)) return bail(msg);
        if (rt_last_stats(J.ctx, &J.stats)) return bail(std::string("rt_last_stats (shard ") + std::to_string(k) + "): " + rt_last_error());
    }

    // rank 0: rows back into image order (the ordered collect() of main.rs:139): reassembly_plan() above, one 2D
    // device copy per shard (+ one for a ragged last tile)
    if (!hip_ok(hipSetDevice(devices[0]), "hipSetDevice", &msg)) return bail(msg);
    if (!hip_ok(hipMalloc((void **)&d_full, (size_t)H * row_words * sizeof(uint64_t)), "hipMalloc frame", &msg)) return bail(msg);
    if (!hip_ok(hipMalloc((void **)&d_rgba, (size_t)H * W * 4), "hipMalloc rgba", &msg)) return bail(msg);
    hipStream_t s0 = jobs[0].stream;
    const size_t row_bytes = row_words * sizeof(uint64_t);
    int copy_calls = 0;
    for (int k = 0; k < n; ++k) {
        const uint64_t *src = use_rccl ? d_staging + (size_t)k * pad_words : jobs[k].d_fix;
        for (const RowCopy &c : reassembly_plan(H, T, n, k)) {
            // (pieces == 1: one contiguous run; hipMemcpy2DAsync with height 1 is that)
            if (!hip_ok(hipMemcpy2DAsync(d_full + (size_t)c.dst_row * row_words, (size_t)c.dst_pitch_rows * row_bytes,
                                         src + (size_t)c.src_row * row_words, (size_t)c.src_pitch_rows * row_bytes,
                                         (size_t)c.rows * row_bytes, (size_t)c.pieces, hipMemcpyDeviceToDevice, s0), "hipMemcpy2DAsync rows", &msg)) return bail(msg);
            ++copy_calls;
        }
    }
    if (copy_calls_out) *copy_calls_out = copy_calls;
    long long spp_total = base.spp;
    if (rt_resolve_rgba8_device(jobs[0].ctx, d_full, W, H, spp_total, 1, d_rgba, s0)) return bail(rt_last_error());
    if (!hip_ok(hipMemcpyAsync(rgba_out, d_rgba, (size_t)H * W * 4, hipMemcpyDeviceToHost, s0), "hipMemcpyAsync rgba", &msg)) return bail(msg);
    if (!hip_ok(hipStreamSynchronize(s0), "hipStreamSynchronize", &msg)) return bail(msg);

    if (stats_out) {
        *stats_out = jobs[0].stats;
        for (int k = 1; k < n; ++k) {
            stats_out->samples += jobs[k].stats.samples; stats_out->rays_traced += jobs[k].stats.rays_traced;
            stats_out->sphere_tests += jobs[k].stats.sphere_tests;
            stats_out->kernel_ms = std::max(stats_out->kernel_ms, jobs[k].stats.kernel_ms);
        }
    }
    cleanup();
    return 0;
}

} // namespace rtiow

// rtiow_render -- the reference's main() (src/main.rs:104-177) on the GPU path: build the scene,
// build the camera, render through the C ABI, flip + to_rgba, save the image (main.rs:177 `image_buffer.save("image.png")`:
// an RGBA8 PNG when --out ends in .png, else the same bytes without alpha as a P6 PPM; no preview window).
//
//   rtiow_render [--width W] [--height H] [--spp N] [--depth D] [--seed S] [--scene-seed S]
//                [--grid LO HI] [--device K] [--out image.png|image.ppm] [--dump-scene scene.bin] [--scene scene.bin]
//                [--devices 0,1,..  [--tile-rows T] [--force-rccl]] [--uniform53] [--two-calls] [--passes N]
//   rtiow_render --reassembly-plan H T N     (no GPU: the strided copies that put N shards' rows back in image order)
//   rtiow_render --test-png W H out.png      (no GPU: a fixed pattern through the PNG writer -- r = 7x + 13y, g = x ^ y, b = x y, mod 256, alpha 255)
//
// Single device: ONE call, rt_render_rgba8 (the sums stay on the device); --two-calls takes them through host memory
// instead (rt_render, then rt_resolve_rgba8): the same bytes.
// --passes N: main.rs:130-137's sample loop as N additive launches (sample_begin, RT_FLAG_ACCUMULATE | RT_FLAG_OVERLAPPED) issued alternately on TWO streams of one
// context, so that pass k + 1 fills the end-of-launch tail of pass k (a context holds two launches' state); the sums are exact integers, so the
// image is the one the single call gives, byte for byte.
// --devices: the frame's rows are dealt round-robin to one rt_context per listed device, each driven by
// its own host thread, and gathered with ONE RCCL ncclGather to the first device (host/rtiow_multi.hpp).
// A device may be listed more than once (two contexts on one GPU from two threads: the threading rule of
// include/rtiow_hip.h); RCCL does not allow that within a communicator, so such a list gathers with plain
// device copies.  --force-rccl runs the RCCL path even for a single device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rtiow_host.hpp"
#include "rtiow_multi.hpp"

static int die(const char *what, int rc)
{
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    int width = 400, height = 225, spp = 10, depth = 50, device = 0, lo = -11, hi = 11;
    unsigned long long seed = 1, scene_seed = 1;
    std::string out = "image.ppm", dump, scene_file;
    std::vector<int> devices;                    // --devices 0,1,...: one context + host thread per entry
    bool force_rccl = false, uniform53 = false, two_calls = false;
    int passes = 1;
    int tile_rows = 1;
    if (argc == 5 && !std::strcmp(argv[1], "--reassembly-plan")) {
        const int H = std::atoi(argv[2]), T = std::atoi(argv[3]), n = std::atoi(argv[4]);
        if (H < 1 || T < 1 || n < 1) { std::fprintf(stderr, "--reassembly-plan H T N: all >= 1\n"); return 2; }
        for (int k = 0; k < n; ++k)
            for (const rtiow::RowCopy &c : rtiow::reassembly_plan(H, T, n, k))
                std::printf("%d %d %d %d %d %d %d\n", k, c.dst_row, c.src_row, c.rows, c.pieces, c.dst_pitch_rows, c.src_pitch_rows);
        return 0;
    }
    if (argc == 5 && !std::strcmp(argv[1], "--test-png")) {
        const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
        if (W < 1 || H < 1) { std::fprintf(stderr, "--test-png W H out.png: W, H >= 1\n"); return 2; }
        std::vector<unsigned char> px((size_t)W * H * 4);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                unsigned char *q = &px[((size_t)y * W + x) * 4];
                q[0] = (unsigned char)(7 * x + 13 * y); q[1] = (unsigned char)(x ^ y); q[2] = (unsigned char)(x * y); q[3] = 255;
            }
        if (!rtiow::write_png(argv[4], px.data(), W, H)) { std::perror(argv[4]); return 1; }
        return 0;
    }
    for (int i = 1; i < argc; ++i) {
        auto arg = [&](const char *n) { return !std::strcmp(argv[i], n) && i + 1 < argc; };
        if (arg("--width")) width = std::atoi(argv[++i]);
        else if (arg("--height")) height = std::atoi(argv[++i]);
        else if (arg("--spp")) spp = std::atoi(argv[++i]);
        else if (arg("--depth")) depth = std::atoi(argv[++i]);
        else if (arg("--seed")) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (arg("--scene-seed")) scene_seed = std::strtoull(argv[++i], nullptr, 0);
        else if (arg("--device")) device = std::atoi(argv[++i]);
        else if (arg("--out")) out = argv[++i];
        else if (arg("--devices")) { for (char *t = std::strtok(argv[++i], ","); t; t = std::strtok(nullptr, ",")) devices.push_back(std::atoi(t)); }
        else if (arg("--tile-rows")) tile_rows = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--force-rccl")) force_rccl = true;
        else if (!std::strcmp(argv[i], "--uniform53")) uniform53 = true;
        else if (!std::strcmp(argv[i], "--two-calls")) two_calls = true;
        else if (arg("--passes")) passes = std::atoi(argv[++i]);
        else if (arg("--dump-scene")) dump = argv[++i];
        else if (arg("--scene")) scene_file = argv[++i];
        else if (!std::strcmp(argv[i], "--grid") && i + 2 < argc) { lo = std::atoi(argv[++i]); hi = std::atoi(argv[++i]); }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    const rtiow::HittableList world = rtiow::random_scene(scene_seed, lo, hi);      // main.rs:106
    std::vector<rt_sphere> flat = world.flatten();
    if (!scene_file.empty()) {                   // a flat scene file instead of random_scene()
        FILE *f = std::fopen(scene_file.c_str(), "rb");
        if (!f) { std::perror(scene_file.c_str()); return 1; }
        std::fseek(f, 0, SEEK_END);
        const long bytes = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (bytes < 0 || bytes % (long)sizeof(rt_sphere)) { std::fprintf(stderr, "%s: not a whole number of 72-byte records\n", scene_file.c_str()); return 1; }
        flat.resize((size_t)bytes / sizeof(rt_sphere));
        if (std::fread(flat.data(), sizeof(rt_sphere), flat.size(), f) != flat.size()) { std::perror(scene_file.c_str()); return 1; }
        std::fclose(f);
    }
    if (!dump.empty()) {                         // the flat scene file: count + 72-byte records
        FILE *f = std::fopen(dump.c_str(), "wb");
        if (!f) { std::perror(dump.c_str()); return 1; }
        std::fwrite(flat.data(), sizeof(rt_sphere), flat.size(), f);
        std::fclose(f);
        std::printf("%zu spheres -> %s\n", flat.size(), dump.c_str());
        return 0;
    }
    const rtiow::Camera cam(rtiow::Point3(13, 2, 3), rtiow::Point3(0, 0, 0), rtiow::Vec3(0, 1, 0), 20.0,
                            (double)width / (double)height, 0.1, 10.0);             // main.rs:108-118
    rt_params p{};
    p.width = width; p.height = height; p.spp = spp; p.sample_begin = 0; p.max_depth = depth;
    p.t_min = 0.0001; p.seed = seed; p.tile_rows = 8; p.shard_index = 0; p.shard_count = 1; p.flags = uniform53 ? RT_FLAG_UNIFORM53 : 0u;
    const size_t npix = (size_t)width * height;
    const rt_camera rc_cam = cam.flat();
    std::vector<uint8_t> rgba(npix * 4);
    rt_stats st{};
    if (!devices.empty()) {
        // one rt_context per listed device, each driven by its own host thread; rows dealt round-robin;
        // one RCCL gather of the exact sums to the first device (main.rs:122-123 + the ordered collect() :139)
        p.tile_rows = tile_rows;
        std::string err;
        int copy_calls = 0;
        if (rtiow::render_sharded(devices, force_rccl, flat, rc_cam, p, rgba.data(), &st, &err, &copy_calls)) {
            std::fprintf(stderr, "render_sharded failed: %s\n", err.c_str());
            return 1;
        }
        std::printf("%zu shards, tiles of %d rows: %d device copies put the rows back in image order\n", devices.size(), tile_rows, copy_calls);
    } else {
        rt_context *ctx = nullptr;
        int rc = rt_create(device, &ctx);
        if (rc) return die("rt_create", rc);
        rc = rt_upload_scene(ctx, flat.data(), (int32_t)flat.size());
        if (rc) return die("rt_upload_scene", rc);
        if (passes > 1) {
            // progressive passes, overlapped: device buffers, two streams, every pass adds its samples to the same exact sums
            if (passes > spp) { std::fprintf(stderr, "--passes %d: more passes than samples per pixel\n", passes); return 2; }
            auto hip_die = [](hipError_t e, const char *what) { std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; };
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) return hip_die(e, "hipSetDevice");
            hipStream_t st2[2] = {nullptr, nullptr};
            void *d_fix = nullptr, *d_rgba = nullptr;
            for (int k = 0; k < 2; ++k) if ((e = hipStreamCreateWithFlags(&st2[k], hipStreamNonBlocking)) != hipSuccess) return hip_die(e, "hipStreamCreate");
            if ((e = hipMalloc(&d_fix, npix * 3 * sizeof(uint64_t))) != hipSuccess) return hip_die(e, "hipMalloc");
            if ((e = hipMalloc(&d_rgba, npix * 4)) != hipSuccess) return hip_die(e, "hipMalloc");
            if ((e = hipMemset(d_fix, 0, npix * 3 * sizeof(uint64_t))) != hipSuccess) return hip_die(e, "hipMemset");      // zero BEFORE any stream adds to it
            unsigned long long rays = 0;
            float kernel_ms = 0.0f;
            int begin = 0;
            for (int k = 0; k < passes; ++k) {
                rt_params q = p;
                q.spp = spp / passes + (k < spp % passes ? 1 : 0);      // the samples dealt as evenly as they go
                q.sample_begin = begin; q.flags |= RT_FLAG_ACCUMULATE | RT_FLAG_OVERLAPPED;
                begin += q.spp;
                rc = rt_render_device(ctx, &rc_cam, &q, d_fix, st2[k & 1]);
                if (rc) return die("rt_render_device", rc);
            }
            for (int k = 0; k < 2; ++k) if ((e = hipStreamSynchronize(st2[k])) != hipSuccess) return hip_die(e, "hipStreamSynchronize");
            rc = rt_last_stats(ctx, &st);                                // (the latest launch only; the ray count below is the frame's)
            if (rc) return die("rt_last_stats", rc);
            rays = st.rays_traced * (unsigned long long)passes; kernel_ms = st.kernel_ms * passes;   // (an estimate for the summary line)
            st.rays_traced = rays; st.kernel_ms = kernel_ms;
            rc = rt_resolve_rgba8_device(ctx, d_fix, width, height, spp, 1, d_rgba, st2[0]);
            if (rc) return die("rt_resolve_rgba8_device", rc);
            if ((e = hipMemcpyAsync(rgba.data(), d_rgba, npix * 4, hipMemcpyDeviceToHost, st2[0])) != hipSuccess) return hip_die(e, "hipMemcpyAsync");
            if ((e = hipStreamSynchronize(st2[0])) != hipSuccess) return hip_die(e, "hipStreamSynchronize");
            (void)hipFree(d_fix); (void)hipFree(d_rgba);
            for (int k = 0; k < 2; ++k) (void)hipStreamDestroy(st2[k]);
        } else if (two_calls) {                  // the sums through host memory: rt_render, then rt_resolve_rgba8 (same bytes)
            std::vector<uint64_t> fix(npix * 3);
            rc = rt_render(ctx, &rc_cam, &p, nullptr, fix.data(), &st);                  // main.rs:122-136
            if (rc) return die("rt_render", rc);
            rc = rt_resolve_rgba8(ctx, fix.data(), width, height, spp, 1, rgba.data());  // main.rs:137,141-145
            if (rc) return die("rt_resolve_rgba8", rc);
        } else {
            // main.rs:122-145 in one call: the sums stay on the device, the flipped RGBA8 bytes come back
            rc = rt_render_rgba8(ctx, &rc_cam, &p, 1, rgba.data(), &st);
            if (rc) return die("rt_render_rgba8", rc);
        }
        rt_destroy(ctx);
    }
    if (out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0) {             // main.rs:177
        if (!rtiow::write_png(out.c_str(), rgba.data(), width, height)) { std::perror(out.c_str()); return 1; }
    } else {
        FILE *f = std::fopen(out.c_str(), "wb");
        if (!f) { std::perror(out.c_str()); return 1; }
        std::fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (size_t k = 0; k < npix; ++k) std::fwrite(&rgba[4 * k], 1, 3, f);
        std::fclose(f);
    }
    std::printf("%dx%d spp %d: %llu rays, kernel %.3f ms (%.1f Msamples/s) -> %s\n", width, height, spp,
                (unsigned long long)st.rays_traced, st.kernel_ms, npix * (double)spp / st.kernel_ms / 1e3, out.c_str());
    return 0;
}

// rt_api.hip -- host runtime behind include/rtiow_hip.h (C ABI of librtiow_hip.so).
//
// Owns the device-side scene, the work counter, the statistics words and the
// HIP events of one context; validates parameters; launches the persistent
// render kernel.  No CPU fallback: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rtiow_hip.h"
#include "rt_kernels.hpp"
#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_host_ctx.hpp"
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define RT_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,        \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

} // namespace

struct rt_context {
    int device = 0;
    int cu_count = 0;
    float *d_filt = nullptr;       // [n][4] f32 filter records (MODE 1)
    double *d_geo = nullptr;       // [n][4] exact geometry
    double *d_mat = nullptr;       // [n][kMatStride] exact materials
    uint4 *d_btube = nullptr;      // [tiles/2 + 1][64] MODE 5 (tube filter) B operands
    float tube_rho = 1.0f;         // MODE 5 radius floor
    double *d_geo_slot = nullptr;  // MODE 5: [slots][4] exact geometry in table (slot) order
    uint32_t *d_slot_orig = nullptr;   // MODE 5: [slots] list index of the sphere in each column of the table
    int n_global = 0;              // MODE 5: tiles [0, n_global) are scanned for every ray; the rest are grid cells
    int grid_dim = 0;              // MODE 5: cells per side of the xz grid (0: no grid, every tile is scanned)
    float grid[8] = {};            // x0, z0, 1/cell, x1, z1, y lo, y hi, pad (rt_device.hpp, grid_cells)
    float scene_scale = 0.0f;      // MODE 5: KParams::scene_scale
#ifdef RTIOW_CROSSCHECK_MODES
    XcheckScene x;                 // device tables of scan modes 2-4 (xcheck/rt_xcheck_host_ctx.hpp)
#endif
    int n_tiles = 0;
    int n_always = 0;
    int always_idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int scan_mode = 5;             // filter: 5 tube bf16x2 MFMA (default, shipped), 1 VALU + scalar loads (cross-check);
                                   // with -DRTIOW_CROSSCHECK_MODES also 2 f32 MFMA, 3 bf16x3 MFMA, 4 lifted bf16x3 MFMA
    int n_spheres = -1;
    // Per-launch state -- the work counter, the statistics words and the two events -- exists kSlots times, used in turn: a context
    // may have kSlots renders in flight (on different streams: the second fills the first one's end-of-launch tail); the fields
    // below name the slot of the LATEST launch, which is what rt_last_stats reports on.
    static constexpr int kSlots = 2;
    unsigned int *q_slots[kSlots] = {nullptr, nullptr};
    unsigned long long *s_slots[kSlots] = {nullptr, nullptr};
    hipEvent_t e0_slots[kSlots] = {nullptr, nullptr}, e1_slots[kSlots] = {nullptr, nullptr};
    bool slot_used[kSlots] = {false, false};
    int cur = 0;
    unsigned int *d_queue = nullptr;
    unsigned long long *d_stats = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t own_stream = nullptr;
    bool launched = false;
    unsigned long long zero_depth_samples = 0;
    rt_stats last{};
    // staging for the host-buffer entry points
    void *d_stage_fix = nullptr; size_t stage_fix_bytes = 0;
    void *d_stage_sum = nullptr; size_t stage_sum_bytes = 0;
    void *d_stage_rgba = nullptr; size_t stage_rgba_bytes = 0;
    int blocks_per_cu = 0;     // 0 = occupancy query
    int ring_min_spp = 0;                 // RTIOW_RING_MIN_SPP (diagnostic): spp per launch from which block sums are kept in LDS (0: the kernel's own minimum)
};

namespace {

int validate_params(const rt_params *p)
{
    if (!p) return fail(RT_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->width < 2 || p->height < 2)
        return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be >= 2 (u,v divide by W-1,H-1; main.rs:131-132)");
    if (p->width > 65535 || p->height > 65535)
        return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be <= 65535");
    if ((long long)p->width * p->height > 0x7fffffffLL)
        return fail(RT_ERR_INVALID_ARGUMENT, "width*height does not fit the 32-bit Philox pixel counter");
    if (p->spp < 0 || p->sample_begin < 0 || (long long)p->spp + p->sample_begin > 0x7fffffffLL)
        return fail(RT_ERR_INVALID_ARGUMENT, "spp/sample_begin out of range");
    if (p->max_depth < 0) return fail(RT_ERR_INVALID_ARGUMENT, "max_depth must be >= 0");
    if (!(p->t_min > 0.0)) return fail(RT_ERR_INVALID_ARGUMENT, "t_min must be > 0 (reference: 1e-4, main.rs:44)");
    if (p->tile_rows < 1) return fail(RT_ERR_INVALID_ARGUMENT, "tile_rows must be >= 1");
    if (p->shard_count < 1 || p->shard_index < 0 || p->shard_index >= p->shard_count)
        return fail(RT_ERR_INVALID_ARGUMENT, "need 0 <= shard_index < shard_count");
    // (a bit this library does not know would be a request it silently ignores: a caller built against a newer header finds out here)
    if (p->flags & ~RT_FLAG_KNOWN)
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown flags 0x%x (this library knows 0x%x)", p->flags & ~RT_FLAG_KNOWN, RT_FLAG_KNOWN);
    return RT_OK;
}

// rows owned by a shard: tiles t = shard_index, shard_index+shard_count, ...
int shard_rows(const rt_params *p)
{
    const long long T = p->tile_rows, H = p->height;
    const long long ntiles = (H + T - 1) / T;
    long long rows = 0;
    for (long long t = p->shard_index; t < ntiles; t += p->shard_count) {
        const long long lo = t * T;
        rows += (lo + T <= H) ? T : (H - lo);
    }
    return (int)rows;
}

int ensure(void **ptr, size_t *have, size_t need)
{
    if (*have >= need && *ptr) return RT_OK;
    if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; *have = 0; }
    RT_HIP(hipMalloc(ptr, need ? need : 16));
    *have = need;
    return RT_OK;
}

// K' of the scan filter for one sphere (rt_device.hpp):
// |c|^2 (1-kappa) - r^2 (1+2 kappa) in f64, rounded DOWN to f32 (a smaller K' keeps more);
// spheres outside f32's comfortable range get -inf: always kept.
float filter_kprime(const rt_sphere &s, double KU)
{
    const double kappa = KU / (1.0 - KU);
    const double r2 = s.radius * s.radius;
    const double c2 = s.center[0] * s.center[0] + s.center[1] * s.center[1] + s.center[2] * s.center[2];
    if (!(r2 > 1e-30) || !(c2 + r2 < 1e30)) return -INFINITY;
    const double exact = c2 * (1.0 - kappa) - r2 * (1.0 + 2.0 * kappa);
    float kp = (float)(exact - std::fabs(exact) * 1e-12);
    if ((double)kp > exact) kp = std::nextafterf(kp, -INFINITY);
    return kp;
}

uint32_t host_bf16_rne(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
void host_split_bf16x3(float x, uint32_t p[3])
{
    auto back = [](uint32_t b) { uint32_t u = b << 16; float f; memcpy(&f, &u, 4); return f; };
    p[0] = host_bf16_rne(x);
    const float r1 = x - back(p[0]);
    p[1] = host_bf16_rne(r1);
    const float r2 = r1 - back(p[1]);
    p[2] = host_bf16_rne(r2);
}

// ---- MODE 5 (tube filter, rt_device.hpp): per-sphere columns and bounds ----
// Radius floor rho: the rows of a ray are scaled by rho / (rho + e_ray), which keeps the test sound for
// every sphere whose bound is >= rho; smaller spheres are tested with the bound rho.  The lower quartile
// of the radii leaves three quarters of the scene untouched and keeps the scaling close to 1.
float tube_radius_floor(const rt_sphere *spheres, int n, const char *skip)
{
    std::vector<double> r;
    for (int i = 0; i < n; ++i)
        if (!(skip && skip[i]) && std::fabs(spheres[i].radius) > 1e-15 && std::fabs(spheres[i].radius) < 1e15)
            r.push_back(std::fabs(spheres[i].radius));
    if (r.empty()) return 1.0f;
    std::nth_element(r.begin(), r.begin() + r.size() / 4, r.end());
    return (float)r[r.size() / 4];
}
// bound of one sphere: max(R, rho), R = r (1+64u) + 640u |c| rounded up; +inf outside the analysed range
float tube_bound(const rt_sphere &s, float rho)
{
    const double r = std::fabs(s.radius);
    const double c2 = s.center[0] * s.center[0] + s.center[1] * s.center[1] + s.center[2] * s.center[2];
    if (!(r * r > 1e-30) || !(c2 + r * r < 1e30)) return INFINITY;
    const double R = r * (1.0 + (double)rt::kTubeBasisErr) + (double)rt::kTubeCenterErr * std::sqrt(c2);
    float f = (float)(R * (1.0 + 1e-12));
    if ((double)f < R) f = std::nextafterf(f, INFINITY);
    return f > rho ? f : rho;
}
// sigma of one sphere: 2 (1 - 2^-6) / bound, rounded DOWN to a bf16 (rt_device.hpp: "kept" <=> |H| < 2);
// 0 for a sphere outside the analysed range (bound = +inf): H = 0, always kept.  Returns the bf16 bit pattern.
uint32_t tube_sigma_bits(float bound)
{
    if (!(bound < INFINITY)) return 0u;
    const float f = (float)(2.0 * (1.0 - 1.0 / 64.0) / (double)bound);
    uint32_t u; memcpy(&u, &f, 4);
    return u >> 16;                                    // truncation = rounding down (sigma > 0)
}
// one tile of 32 columns: B operands [64] (lane l = column l&31, K-slots 8(l>>5)..+7) and, for the tests, the
// bounds [32] the columns were scaled with.  `s[c] == nullptr`: a column no ray keeps (padding, always-exact list):
// all zero but for K-slot 15, where 4 meets the 1 every ray carries there.
void tube_tile(const rt_sphere *const s[32], float rho, uint4 out_b[64], float out_r[32])
{
    for (int c = 0; c < 32; ++c) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        out_r[c] = -1.0f;
        if (s[c]) {
            out_r[c] = tube_bound(*s[c], rho);
            const uint32_t sg = tube_sigma_bits(out_r[c]);
            uint32_t sgu = sg << 16; float sigma; memcpy(&sigma, &sgu, 4);
            if (sg != 0u) {                            // (always-kept columns keep sigma = 0, c = 0: H = 0)
                for (int i = 0; i < 3; ++i) {
                    // two bf16 pieces of sigma * (the f64 centre): |sigma c - (y1 + y2)| <= 2^-16 |sigma c|
                    const double ci = s[c]->center[i] * (double)sigma;
                    const uint32_t y1 = host_bf16_rne((float)ci);
                    uint32_t u1 = y1 << 16; float f1; memcpy(&f1, &u1, 4);
                    const uint32_t y2 = host_bf16_rne((float)(ci - (double)f1));
                    w[2 * i + 0] = y1 | (y2 << 16);
                    w[2 * i + 1] = y1 | (y2 << 16);
                }
            }
            w[6] = sg | (sg << 16);                    // sigma against the three exact pieces of t
            w[7] = sg;
        } else {
            w[7] = 0x4080u << 16;                      // K-slot 15: 4.0 -> H = 4 for every ray: never kept
        }
        out_b[c] = make_uint4(w[0], w[1], w[2], w[3]);
        out_b[32 + c] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// releases every device table of the scene and marks the context as having none
void free_scene(rt_context *ctx)
{
    (void)hipFree(ctx->d_filt); (void)hipFree(ctx->d_geo); (void)hipFree(ctx->d_mat);
    (void)hipFree(ctx->d_btube); (void)hipFree(ctx->d_geo_slot); (void)hipFree(ctx->d_slot_orig);
    ctx->d_filt = nullptr; ctx->d_geo = ctx->d_mat = nullptr; ctx->d_btube = nullptr; ctx->d_geo_slot = nullptr; ctx->d_slot_orig = nullptr;
#ifdef RTIOW_CROSSCHECK_MODES
    ctx->x.release();
#endif
    ctx->n_spheres = -1;
}

// one table: allocate + copy; on failure the caller frees the whole scene
template <typename T>
int upload_table(T **dst, const T *src, size_t count)
{
    RT_HIP(hipMalloc((void **)dst, (count ? count : 1) * sizeof(T)));
    if (count) RT_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

template <int MODE, bool DIAG, bool SMALLGRID = false, bool U53 = false, int ITEMS = rt::kItemBlock>
int launch_render(rt_context *ctx, const rt::KParams &kp, hipStream_t stream, int *grid_out)
{
    int per_cu = ctx->blocks_per_cu;
    if (per_cu <= 0) {
        int occ = 0;
        RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rt::render_kernel<MODE, DIAG, SMALLGRID, U53, ITEMS>, rt::kBlock, 0));
        per_cu = occ < 1 ? 1 : (occ > 8 ? 8 : occ);
    }
    // persistent grid, but never more lanes than there are work items
    long long grid = (long long)ctx->cu_count * per_cu;
    const long long need = (long long)((kp.total_items + rt::kBlock - 1) / rt::kBlock);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    *grid_out = (int)grid;
    RT_HIP(hipEventRecord(ctx->ev0, stream));
    hipLaunchKernelGGL((rt::render_kernel<MODE, DIAG, SMALLGRID, U53, ITEMS>), dim3((unsigned)grid), dim3(rt::kBlock), 0, stream, kp);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(ctx->ev1, stream));
    return RT_OK;
}

// MODE 5 numbers candidates by table column in 26 bits (the pool word of rt_kernels.hpp is column << 6 | ray), and the
// large-grid kernel keeps one 64-bit word per grid row in LDS with one lane per row (a run of columns is (1 << n) - 1 << x0).
constexpr size_t kMaxColumns = (size_t)1 << 26;
constexpr int kMaxGridDim = 63;

// Where MODE 5 puts each sphere in its table of columns (tiles of 32), and the grid the kernel finds tiles with.
struct TileLayout {
    int grid_dim = 0, n_global = 0;
    float grid[8] = {};             // rt_kernels.hpp, KParams::grid
    float scale = 0.0f;             // sum over axes of the largest |coordinate| of the grid's box
    std::vector<int> slot_of;       // column -> place in the caller's list, -1 = padding; a multiple of 32 long
};

// The spheres that skip the filter and are always tested exactly: much larger than the rest of the scene (the
// ground), the filter would keep them for nearly every ray.  The choice only moves work, never results.
std::vector<int> always_exact_list(const rt_sphere *spheres, int n)
{
    std::vector<int> out;
    if (n <= 0) return out;
    std::vector<double> radii(n);
    for (int i = 0; i < n; ++i) radii[i] = std::fabs(spheres[i].radius);
    std::vector<double> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double big = 8.0 * sorted[n / 2];
    std::vector<int> order;
    for (int i = 0; i < n; ++i) if (radii[i] > big) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return radii[x] > radii[y]; });
    for (size_t k = 0; k < order.size() && k < 8; ++k) out.push_back(order[k]);
    return out;
}

TileLayout tile_layout(const rt_sphere *spheres, int n, const char *never)
{
    TileLayout L;
    // ---- which column of the table holds which sphere --------------------------------------------------
    // Spheres are put into tiles of 32 columns by WHERE they are, so that a wave only scans the tiles its rays
    // can reach (rt_device.hpp, grid_cells): a square grid over the xz extent of the small spheres, one tile
    // per cell (what does not fit a cell's 32 columns overflows), preceded by "global" tiles that every ray scans:
    // spheres too large for a cell and the overflow.  The order of the columns decides nothing: ties are resolved
    // on the spheres' positions in the caller's list (slot_orig).  Diagnostic knobs, read at upload:
    // RTIOW_NO_GRID=1 (columns in list order, every tile scanned), RTIOW_GRID_DIM=G (cells per side).
    std::vector<int> filtered;
    for (int i = 0; i < n; ++i) if (!never[i]) filtered.push_back(i);
    std::vector<int> &slot_of = L.slot_of;
    L.grid_dim = 0; L.n_global = 0;
    if (filtered.size() > 64 && !env_int("RTIOW_NO_GRID", 0)) {
        std::vector<double> rr;
        for (int i : filtered) rr.push_back(std::fabs(spheres[i].radius));
        std::nth_element(rr.begin(), rr.begin() + rr.size() / 2, rr.end());
        const double med = rr[rr.size() / 2];
        double x0 = INFINITY, x1 = -INFINITY, z0 = INFINITY, z1 = -INFINITY;
        for (int i : filtered) {
            if (std::fabs(spheres[i].radius) > 3.0 * med) continue;
            x0 = std::min(x0, spheres[i].center[0]); x1 = std::max(x1, spheres[i].center[0]);
            z0 = std::min(z0, spheres[i].center[2]); z1 = std::max(z1, spheres[i].center[2]);
        }
        const double extent = std::max(x1 - x0, z1 - z0);
        // The kernel finds cells with f32 arithmetic on (coordinate - x0) * (1 / cell): the grid exists only while that
        // is meaningful -- a positive extent below 1e15 (coordinates up to 1e15 are legal, so extents up to 2e15 occur)
        // whose cell size has a finite, normal f32 reciprocal.  extent == 0 (every small sphere above the same point):
        // ONE cell of size 1.  Anything else: no grid, every tile scanned, columns in list order.
        const bool one_cell = extent == 0.0;
        const bool grid_ok = one_cell || (extent > 0.0 && extent < 1e15);
        if (grid_ok) {
        // cells(G): the spheres of each cell of a G x G grid, and what does not go into a cell
        std::vector<std::vector<int>> cells;
        std::vector<int> global;
        double cell = 1.0;
        auto assign = [&](int G, bool keep) -> int {       // -> number of global tiles
            cell = one_cell ? 1.0 : extent / G;
            std::vector<int> count((size_t)G * G, 0);
            if (keep) { cells.assign((size_t)G * G, {}); global.clear(); }
            size_t n_glob = 0;
            for (int i : filtered) {
                const double r = std::fabs(spheres[i].radius);
                bool to_cell = !(r > 3.0 * med || r > 0.25 * cell);
                size_t c = 0;
                if (to_cell) {
                    int ix = (int)std::floor((spheres[i].center[0] - x0) / cell), iz = (int)std::floor((spheres[i].center[2] - z0) / cell);
                    ix = std::max(0, std::min(ix, G - 1)); iz = std::max(0, std::min(iz, G - 1));
                    c = (size_t)iz * G + ix;
                    to_cell = count[c] < 32;                        // the cell's tile is full: overflow
                }
                if (to_cell) { ++count[c]; if (keep) cells[c].push_back(i); }
                else { ++n_glob; if (keep) global.push_back(i); }
            }
            return (int)((n_glob + 31) / 32);
        };
        // The grid's resolution: a wave scans the global tiles plus the cells its 64 rays touch, and rays are lines --
        // the cells touched grow like G (measured on the book scenes: about 1.4 G - 1.6 of G x G), while coarse cells
        // overflow into global tiles.  Take the G with the smallest  global tiles + 1.4 G.
        int G = one_cell ? 1 : env_int("RTIOW_GRID_DIM", 0);
        if (G <= 0) {
            double best = INFINITY;
            for (int g = 1; g <= kMaxGridDim; ++g) {
                if ((double)g * g > (double)filtered.size()) break;
                const int ng = assign(g, false);
                if (ng > 48) continue;                              // (the kernel's list holds 126 tiles)
                const double cost = ng + 1.4 * g;
                if (cost < best) { best = cost; G = g; }
            }
        }
        G = std::max(1, std::min(G, kMaxGridDim));          // (no G qualified: G = 1 will not either, and the grid stays off)
        (void)assign(G, true);
        double ylo = INFINITY, yhi = -INFINITY, pad = 0.0;
        for (const std::vector<int> &c : cells)
            for (int i : c) {
                const double r = std::fabs(spheres[i].radius);
                ylo = std::min(ylo, spheres[i].center[1] - r); yhi = std::max(yhi, spheres[i].center[1] + r);
                pad = std::max(pad, r);
            }
        const int n_global = (int)((global.size() + 31) / 32);
        const float inv_cell = (float)(1.0 / cell);
        if (std::isnormal(inv_cell) && (size_t)(n_global + G * G) * 32 <= (size_t)kMaxColumns && n_global <= 48 && ylo <= yhi) {
            L.grid_dim = G; L.n_global = n_global;
            slot_of.assign((size_t)(n_global + G * G) * 32, -1);
            for (size_t k = 0; k < global.size(); ++k) slot_of[k] = global[k];
            for (size_t c = 0; c < cells.size(); ++c)
                for (size_t k = 0; k < cells[c].size(); ++k) slot_of[((size_t)n_global + c) * 32 + k] = cells[c][k];
            auto down = [](double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; };
            auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; };
            L.grid[0] = down(x0); L.grid[1] = down(z0);
            // the kernel turns a coordinate into a cell with THESE f32 values; rounding 1/cell either way only
            // shifts cell borders by ~1e-7 cells, which the kernel's own margin (1e-3 cells) covers
            L.grid[2] = inv_cell;
            L.grid[3] = up(x0 + G * cell); L.grid[4] = up(z0 + G * cell);
            L.grid[5] = down(ylo); L.grid[6] = up(yhi); L.grid[7] = up(pad);
            // the kernel's error margins are relative to the size of what a ray can reach inside the grid's box
            const double gs = std::max(std::fabs(x0), std::fabs(x0 + G * cell)) + pad + std::max(std::fabs(ylo), std::fabs(yhi)) +
                              std::max(std::fabs(z0), std::fabs(z0 + G * cell)) + pad;
            L.scale = (float)gs * 1.0001f;
        }
        }   // grid_ok
    }
    if (L.grid_dim == 0) {                              // no grid: the columns in list order, every tile scanned
        slot_of.assign((size_t)((filtered.empty() ? 0 : filtered.back() + 1) + 31) / 32 * 32, -1);
        for (int i : filtered) slot_of[i] = i;
    }
    return L;
}

#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_host.inc"       // B operands of scan modes 2-4
#endif

} // namespace

extern "C" {

const char *rt_last_error(void) { return g_err; }
const char *rt_backend_name(void) { return "hip-gfx950"; }
int32_t rt_abi_version(void) { return RTIOW_HIP_ABI_VERSION; }
#ifndef RT_SOURCE_SHA
#define RT_SOURCE_SHA "unknown"
#endif
const char *rt_build_source_sha(void) { return RT_SOURCE_SHA; }

int rt_create(int32_t device_id, rt_context **out)
{
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device visible (%s); librtiow_hip has no CPU fallback",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return fail(RT_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, count);
    RT_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only",
                    device_id, prop.gcnArchName);
    rt_context *ctx = new (std::nothrow) rt_context();
    if (!ctx) return fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount;
    ctx->blocks_per_cu = env_int("RTIOW_BLOCKS_PER_CU", 0);
    ctx->ring_min_spp = env_int("RTIOW_RING_MIN_SPP", 0);                             // (raised to the kernel's own minimum per launch)
    ctx->scan_mode = env_int("RTIOW_SCAN_MODE", 5);
#ifdef RTIOW_CROSSCHECK_MODES
    if (ctx->scan_mode < 1 || ctx->scan_mode > 5) ctx->scan_mode = 5;
#else
    if (ctx->scan_mode != 1 && ctx->scan_mode != 5) {
        const int asked = ctx->scan_mode;
        rt_destroy(ctx);
        return fail(RT_ERR_INVALID_ARGUMENT, "RTIOW_SCAN_MODE=%d: this build carries scan modes 1 and 5 (modes 2-4 need "
                    "a -DRTIOW_CROSSCHECK_MODES build)", asked);
    }
#endif
    hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess, e4 = hipSuccess;
    for (int k = 0; k < rt_context::kSlots; ++k) {
        hipError_t a = hipMalloc((void **)&ctx->q_slots[k], 64), b = hipMalloc((void **)&ctx->s_slots[k], 1024);
        hipError_t c = hipEventCreate(&ctx->e0_slots[k]), d = hipEventCreate(&ctx->e1_slots[k]);
        if (a != hipSuccess) e1 = a;
        if (b != hipSuccess) e2 = b;
        if (c != hipSuccess) e3 = c;
        if (d != hipSuccess) e4 = d;
    }
    ctx->d_queue = ctx->q_slots[0]; ctx->d_stats = ctx->s_slots[0]; ctx->ev0 = ctx->e0_slots[0]; ctx->ev1 = ctx->e1_slots[0];
    hipError_t e5 = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        rt_destroy(ctx);
        return fail(RT_ERR_HIP, "context setup failed on device %d", device_id);
    }
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_context *ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    free_scene(ctx);
    for (int k = 0; k < rt_context::kSlots; ++k) {
        (void)hipFree(ctx->q_slots[k]); (void)hipFree(ctx->s_slots[k]);
        if (ctx->e0_slots[k]) (void)hipEventDestroy(ctx->e0_slots[k]);
        if (ctx->e1_slots[k]) (void)hipEventDestroy(ctx->e1_slots[k]);
    }
    (void)hipFree(ctx->d_stage_fix); (void)hipFree(ctx->d_stage_sum); (void)hipFree(ctx->d_stage_rgba);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

int rt_upload_scene(rt_context *ctx, const rt_sphere *spheres, int32_t n)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n < 0 || (n > 0 && !spheres)) return fail(RT_ERR_INVALID_ARGUMENT, "bad sphere list");
    if (n > RT_MAX_SPHERES) return fail(RT_ERR_INVALID_ARGUMENT, "at most %d spheres (RT_MAX_SPHERES)", RT_MAX_SPHERES);
    // (scan mode 1, the VALU cross-check filter, keeps 16-bit candidate lists in LDS)
    if (ctx->scan_mode == 1 && n > 65535) return fail(RT_ERR_INVALID_ARGUMENT, "RTIOW_SCAN_MODE=1 (the cross-check filter) takes at most 65535 spheres");
    for (int i = 0; i < n; ++i) {
        const rt_sphere &s = spheres[i];
        if (s.kind < RT_LAMBERTIAN || s.kind > RT_DIALECTRIC)
            return fail(RT_ERR_INVALID_ARGUMENT, "sphere %d: unknown material kind %d", i, s.kind);
        const double mags[4] = { s.center[0], s.center[1], s.center[2], s.radius };
        for (double v : mags)
            if (!(std::fabs(v) < 1e15))
                return fail(RT_ERR_INVALID_ARGUMENT, "sphere %d: coordinates/radius must be finite and below 1e15", i);
        // sphere.rs:37 divides by the radius: a zero radius makes every normal inf/NaN in the reference too
        if (s.radius == 0.0) return fail(RT_ERR_INVALID_ARGUMENT, "sphere %d: radius must not be zero", i);
    }
    RT_HIP(hipSetDevice(ctx->device));
    // the previous scene may still be in use by a launch on any stream
    RT_HIP(hipDeviceSynchronize());
    free_scene(ctx);
    const size_t cnt = (size_t)(n > 0 ? n : 1);
    std::vector<double> geo(cnt * 4, 0.0), mat(cnt * rt::kMatStride, 0.0);
    for (int i = 0; i < n; ++i) {
        const rt_sphere &s = spheres[i];
        // exact records: radius*radius (sphere.rs:22) and 1.0/radius (vec3.rs:371-375 applied
        // at sphere.rs:37) are per-sphere constants, each one f64 rounding, as in the reference
        const double r2 = s.radius * s.radius;
        geo[4 * i + 0] = s.center[0]; geo[4 * i + 1] = s.center[1]; geo[4 * i + 2] = s.center[2];
        geo[4 * i + 3] = r2;
        double *m = &mat[(size_t)rt::kMatStride * i];
        m[0] = 1.0 / s.radius;
        m[1] = s.param;
        m[2] = s.albedo[0]; m[3] = s.albedo[1]; m[4] = s.albedo[2];
        m[5] = (double)s.kind;
        if (s.kind == RT_DIALECTRIC) {
            // materials.rs:84-87 `1.0/self.ir` and :79 `((1-ri)/(1+ri)).powi(2)` for the two ratios a
            // Dialectric can see (front: 1/ir, back: ir): the reference's own f64 operations, hoisted
            m[6] = 1.0 / s.param;
            double r0 = (1.0 - m[6]) / (1.0 + m[6]); m[7] = r0 * r0;
            r0 = (1.0 - s.param) / (1.0 + s.param); m[8] = r0 * r0;
            m[2] = 1.0; m[3] = 1.0; m[4] = 1.0;                                  // attenuation (1,1,1), :103
        }
    }
    // the spheres that skip the filter (always_exact_list above)
    ctx->n_always = 0;
    for (int i : always_exact_list(spheres, n)) ctx->always_idx[ctx->n_always++] = i;
    // tile count (tiles of 16 columns) rounded up to even, plus two spare tiles so the pipelined loops
    // never branch on a table bound (padding columns are never kept)
    const int n_tiles = 2 * ((n + 31) / 32);
    ctx->n_tiles = n_tiles;
    int rc = RT_OK;
    // Only the table of the scan mode this context runs is built (RTIOW_SCAN_MODE, read at rt_create).
    if (ctx->scan_mode == 5) {      // the tube filter (shipped)
        std::vector<char> never(n > 0 ? n : 1, 0);
        for (int e = 0; e < ctx->n_always; ++e) never[ctx->always_idx[e]] = 1;
        ctx->tube_rho = tube_radius_floor(spheres, n, never.data());
        // which column of the table holds which sphere, and the grid the kernel finds tiles with
        const TileLayout L = tile_layout(spheres, n, never.data());
        const std::vector<int> &slot_of = L.slot_of;
        ctx->grid_dim = L.grid_dim; ctx->n_global = L.n_global;
        for (int k = 0; k < 8; ++k) ctx->grid[k] = L.grid[k];
        ctx->scene_scale = L.scale;
        const int n_tiles32 = (int)(slot_of.size() / 32);
        ctx->n_tiles = 2 * n_tiles32;                          // (counted in 16-column units, as the other scan modes do)
        const size_t ttc = (size_t)n_tiles32 + 1;              // one spare tile: the pipelined loop never branches on a table bound
        std::vector<uint4> btube(ttc * 64);
        std::vector<float> rtube(ttc * 32);
        std::vector<double> geo_slot(ttc * 32 * 4, 0.0);
        std::vector<uint32_t> slot_orig(ttc * 32, 0xFFFFFFFFu);
        for (size_t t = 0; t < ttc; ++t) {
            const rt_sphere *col[32];
            for (int c = 0; c < 32; ++c) {
                const size_t slot = 32 * t + c;
                const int i = slot < slot_of.size() ? slot_of[slot] : -1;
                col[c] = i >= 0 ? &spheres[i] : nullptr;
                if (i >= 0) {
                    slot_orig[slot] = (uint32_t)i;
                    for (int k = 0; k < 4; ++k) geo_slot[4 * slot + k] = geo[4 * (size_t)i + k];
                }
            }
            tube_tile(col, ctx->tube_rho, &btube[t * 64], &rtube[t * 32]);
        }
        if (!rc) rc = upload_table(&ctx->d_btube, btube.data(), btube.size());
        if (!rc) rc = upload_table(&ctx->d_geo_slot, geo_slot.data(), geo_slot.size());
        if (!rc) rc = upload_table(&ctx->d_slot_orig, slot_orig.data(), slot_orig.size());
    }
    // filter records of the f32 evaluation schemes (mode 1, and the sources of the mode 2/3 tables):
    // centre rounded to f32 + K'
    std::vector<float> filt(cnt * 4, 0.0f);
    if (ctx->scan_mode >= 1 && ctx->scan_mode <= 3) {
        for (int i = 0; i < n; ++i) {
            const rt_sphere &s = spheres[i];
            filt[4 * i + 0] = (float)s.center[0]; filt[4 * i + 1] = (float)s.center[1];
            filt[4 * i + 2] = (float)s.center[2]; filt[4 * i + 3] = filter_kprime(s, (double)rt::kFilterKU);
        }
        if (ctx->scan_mode == 1 && !rc) rc = upload_table(&ctx->d_filt, filt.data(), filt.size());
    }
#ifdef RTIOW_CROSSCHECK_MODES
    if (!rc) rc = xcheck_upload_tables(ctx, spheres, n, n_tiles, filt);
#endif
    if (!rc) rc = upload_table(&ctx->d_geo, geo.data(), geo.size());
    if (!rc) rc = upload_table(&ctx->d_mat, mat.data(), mat.size());
    if (rc) {                       // a failed upload leaves NO scene behind (message of the failing call kept)
        free_scene(ctx);
        return rc;
    }
    ctx->n_spheres = n;
    return RT_OK;
}

int rt_shard_rows(const rt_params *p, int32_t *out_rows)
{
    if (!out_rows) return fail(RT_ERR_INVALID_ARGUMENT, "out_rows is NULL");
    int rc = validate_params(p);
    if (rc) return rc;
    *out_rows = shard_rows(p);
    return RT_OK;
}

int rt_shard_row_index(const rt_params *p, int32_t compact_row, int32_t *out_j)
{
    if (!out_j) return fail(RT_ERR_INVALID_ARGUMENT, "out_j is NULL");
    int rc = validate_params(p);
    if (rc) return rc;
    if (compact_row < 0 || compact_row >= shard_rows(p))
        return fail(RT_ERR_INVALID_ARGUMENT, "compact_row out of range");
    const int lt = compact_row / p->tile_rows;
    *out_j = (lt * p->shard_count + p->shard_index) * p->tile_rows + (compact_row - lt * p->tile_rows);
    return RT_OK;
}

int rt_render_device(rt_context *ctx, const rt_camera *cam, const rt_params *p, void *d_fix, void *stream_v)
{
    if (!ctx || !cam) return fail(RT_ERR_INVALID_ARGUMENT, "ctx/cam is NULL");
    int rc = validate_params(p);
    if (rc) return rc;
    if (ctx->n_spheres < 0) return fail(RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    const int rows = shard_rows(p);
    if (rows > 0 && !d_fix) return fail(RT_ERR_INVALID_ARGUMENT, "d_fix is NULL");
    RT_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;

    const long long npix = (long long)rows * p->width;
    // Work items are single pixel-samples in pixel-major order (item w = pixel * spp + sample), handed out in
    // blocks of kItemBlock consecutive items; the work counter counts blocks.
    const unsigned long long total_items = (unsigned long long)npix * (unsigned long long)p->spp;
    // Blocks of kItemBlockLarge for launches that are long enough for their last blocks not to matter (rt_kernels.hpp): the shipped
    // scan mode without the diagnostic counters, block sums in LDS, >= 147 samples per pixel (69 for scenes on the small-grid kernel), >= 2 x 10^8 pixel-samples.  RTIOW_LARGE_BLOCK_MIN_ITEMS
    // moves the last threshold (tests: 0 = every launch that qualifies otherwise; a huge value = never).
    const int mode_now = (p->flags & RT_FLAG_NO_FILTER) ? 0 : ctx->scan_mode;
    const char *lb_env = getenv("RTIOW_LARGE_BLOCK_MIN_ITEMS");
    const unsigned long long lb_min = (lb_env && *lb_env) ? strtoull(lb_env, nullptr, 0)
                                    : (p->flags & RT_FLAG_OVERLAPPED) ? 0ull : rt::kLargeMinItems;   // (overlapped passes: the next pass fills the tail)
    const bool small_grid_scene = ctx->grid_dim > 0 && ctx->n_global + ctx->grid_dim * ctx->grid_dim <= 64;
    // Block sums in LDS: a block's consecutive samples must touch no more pixels than its sums have slots -- ceil((items - 1) / spp) + 1 <= 8,
    // or 16 on the shipped scan mode's kernels --: blocks of 256 from 37 (17) samples per pixel on, and below that the largest multiple of 64 (a block
    // is started 64 samples at a time) that fits: 192, 128 or 64 pixel-samples, down to 9 (5) samples per pixel.  Fewer: every sample is added
    // to the frame buffer with three 64-bit atomics of its own -- a quarter of the frame time at 20-32 samples per pixel (1200x675x32: 5.46 ms
    // that way, 4.04 ms with block sums).  RTIOW_RING_MIN_SPP=n (tests): no block sums below n samples per pixel.
    // (the large-grid kernel's instantiation for blocks of 1 024 keeps 4 x 8: a launch that will take large blocks is sized for 8 slots)
    const bool shipped_kernel = mode_now == 5 && !(p->flags & RT_FLAG_DIAG_STATS);
    const bool large_candidate = shipped_kernel && !small_grid_scene && p->spp >= rt::kLargeMinSpp && total_items >= lb_min && p->spp >= ctx->ring_min_spp;
    const bool wide_ring = shipped_kernel && !large_candidate;
    const unsigned ring_slots = wide_ring ? 2u * (unsigned)rt::kRingSlots : (unsigned)rt::kRingSlots;
    unsigned small_block = 0;
    for (unsigned items = rt::kItemBlock; items >= 64u && p->spp >= 1; items -= 64u)
        if ((items - 1u + (unsigned)p->spp - 1u) / (unsigned)p->spp + 1u <= ring_slots) { small_block = items; break; }
    const bool use_ring = small_block != 0u && p->spp >= ctx->ring_min_spp;
    const bool large_blocks = mode_now == 5 && !(p->flags & RT_FLAG_DIAG_STATS) && use_ring && small_block == (unsigned)rt::kItemBlock &&
                              p->spp >= (small_grid_scene ? rt::kLargeMinSppSmallGrid : rt::kLargeMinSpp) && total_items >= lb_min;
    const unsigned item_block = large_blocks ? rt::kItemBlockLarge : use_ring ? small_block : rt::kItemBlock;
    const unsigned long long n_blocks = (total_items + item_block - 1) / item_block;
    if (n_blocks > 0x7fffffffULL)
        return fail(RT_ERR_INVALID_ARGUMENT, "rows*width*spp = %llu pixel-samples in one launch: at most 2^31 blocks of %d "
                    "(split the samples over several launches with sample_begin and RT_FLAG_ACCUMULATE)", total_items, (int)item_block);
    rt::KParams kp;
    memset(&kp, 0, sizeof(kp));
    static_assert(sizeof(rt::KCamera) == sizeof(rt_camera), "camera layouts must match");
    memcpy(&kp.cam, cam, sizeof(rt_camera));
    kp.width = p->width; kp.height = p->height;
    kp.spp = p->spp; kp.sample_begin = p->sample_begin; kp.max_depth = p->max_depth;
    kp.t_min = p->t_min;
    kp.k0 = (uint32_t)p->seed; kp.k1 = (uint32_t)(p->seed >> 32);
    kp.tile_rows = p->tile_rows; kp.shard_index = p->shard_index; kp.shard_count = p->shard_count;
    kp.rows = rows; kp.n_spheres = ctx->n_spheres;
    kp.npix = (uint32_t)npix; kp.total_items = total_items; kp.n_blocks = (uint32_t)n_blocks;
    kp.inv_spp = p->spp > 0 ? 1.0 / (double)p->spp : 0.0;
    kp.inv_width = 1.0 / (double)p->width;
    // udiv_small (rt_kernels.hpp): numerators are < d + kItemBlockLarge (spp, width) or < 65536 (rows / tile_rows), so for
    // d < 2^15 the product x * d stays below 2^32 and floor(x * M / 2^32) is the exact quotient
    auto magic_for = [](long long d) -> uint32_t {
        return (d <= 1 || d >= 32768) ? 0u : (uint32_t)(0x100000000ULL / (unsigned long long)d + 1ULL);
    };
    kp.magic_spp = magic_for(p->spp); kp.magic_width = magic_for(p->width); kp.magic_tile = magic_for(p->tile_rows);
    kp.use_ring = use_ring ? 1 : 0;
    kp.block_items = item_block;
    kp.filt = ctx->d_filt; kp.geo = ctx->d_geo; kp.mat = ctx->d_mat;
    kp.n_tiles = ctx->n_tiles;
#ifdef RTIOW_CROSSCHECK_MODES
    kp.x.bmat = ctx->x.d_bmat; kp.x.kpt = ctx->x.d_kpt;
    kp.x.bmat16 = ctx->x.d_bmat16; kp.x.kpt16 = ctx->x.d_kpt16; kp.x.bmatL = ctx->x.d_bmatL;
#endif
    kp.btube = ctx->d_btube; kp.tube_rho = ctx->tube_rho;
    kp.geo_slot = ctx->d_geo_slot; kp.slot_orig = ctx->d_slot_orig;
    kp.n_global = ctx->n_global; kp.grid_dim = ctx->grid_dim;
    kp.grid_rows = 0ull;
    for (int k = 0; ctx->grid_dim > 0 && (k + 1) * ctx->grid_dim <= 64; ++k) kp.grid_rows |= 1ull << (k * ctx->grid_dim);
    for (int k = 0; k < 8; ++k) kp.grid[k] = ctx->grid[k];
    kp.scene_scale = ctx->scene_scale;
    kp.n_always = ctx->n_always;
    for (int e = 0; e < 8; ++e) kp.always_idx[e] = ctx->always_idx[e];
    kp.fix = (unsigned long long *)d_fix;
    // the other slot of per-launch state; if the launch that used it last is still running (on another stream), THIS stream waits for
    // it -- the host does not -- before the counters are cleared
    ctx->cur = (ctx->cur + 1) % rt_context::kSlots;
    ctx->d_queue = ctx->q_slots[ctx->cur]; ctx->d_stats = ctx->s_slots[ctx->cur];
    ctx->ev0 = ctx->e0_slots[ctx->cur]; ctx->ev1 = ctx->e1_slots[ctx->cur];
    if (ctx->slot_used[ctx->cur]) RT_HIP(hipStreamWaitEvent(stream, ctx->ev1, 0));
    ctx->slot_used[ctx->cur] = true;
    kp.queue = ctx->d_queue; kp.stats = ctx->d_stats;

    if (!(p->flags & RT_FLAG_ACCUMULATE) && npix > 0)
        RT_HIP(hipMemsetAsync(d_fix, 0, (size_t)npix * 3 * sizeof(unsigned long long), stream));
    RT_HIP(hipMemsetAsync(ctx->d_queue, 0, 64, stream));
    RT_HIP(hipMemsetAsync(ctx->d_stats, 0, 1024, stream));

    memset(&ctx->last, 0, sizeof(ctx->last));
    ctx->last.n_spheres = ctx->n_spheres;
    ctx->last.block_threads = rt::kBlock;
    ctx->zero_depth_samples = 0;
    if (p->max_depth == 0 || kp.total_items == 0) {
        // ray_color(depth <= 0) returns black without tracing (main.rs:40-42): the sums stay
        // as they are and no ray is cast; nothing to launch.
        RT_HIP(hipEventRecord(ctx->ev0, stream));
        RT_HIP(hipEventRecord(ctx->ev1, stream));
        ctx->zero_depth_samples = (unsigned long long)npix * (unsigned long long)p->spp;
        ctx->launched = true;
        return RT_OK;
    }

    int grid = 0;
    const bool diag = (p->flags & RT_FLAG_DIAG_STATS) != 0;
    const int mode = (p->flags & RT_FLAG_NO_FILTER) ? 0 : ctx->scan_mode;
    ctx->last.scan_mode = mode;
    ctx->last.kernel_variant = 0;
    const bool small_grid = small_grid_scene;
    if (p->flags & RT_FLAG_UNIFORM53) {
        // 53-bit uniforms: instantiated for the shipped scan mode (both grid variants) and for RT_FLAG_NO_FILTER
        if (diag || (mode != 0 && mode != 5))
            return fail(RT_ERR_INVALID_ARGUMENT, "RT_FLAG_UNIFORM53 runs with scan mode 5 (the default) or RT_FLAG_NO_FILTER, without RT_FLAG_DIAG_STATS");
        if (mode == 0) rc = launch_render<0, false, false, true>(ctx, kp, stream, &grid);
        else if (small_grid) rc = large_blocks ? launch_render<5, false, true, true, rt::kItemBlockLarge>(ctx, kp, stream, &grid)
                                              : launch_render<5, false, true, true>(ctx, kp, stream, &grid);
        else rc = large_blocks ? launch_render<5, false, false, true, rt::kItemBlockLarge>(ctx, kp, stream, &grid)
                               : launch_render<5, false, false, true>(ctx, kp, stream, &grid);
        if (rc) return rc;
        ctx->last.kernel_variant = 2 | ((mode == 5 && small_grid) ? 1 : 0) | (mode == 5 && large_blocks ? 4 : 0);
        ctx->launched = true;
        ctx->last.grid_blocks = grid;
        return RT_OK;
    }
    switch (mode * 2 + (diag ? 1 : 0)) {
    case 0: rc = launch_render<0, false>(ctx, kp, stream, &grid); break;
    case 1: rc = launch_render<0, true>(ctx, kp, stream, &grid); break;
    case 2: rc = launch_render<1, false>(ctx, kp, stream, &grid); break;
    case 3: rc = launch_render<1, true>(ctx, kp, stream, &grid); break;
#ifdef RTIOW_CROSSCHECK_MODES
    case 4: rc = launch_render<2, false>(ctx, kp, stream, &grid); break;
    case 5: rc = launch_render<2, true>(ctx, kp, stream, &grid); break;
    case 6: rc = launch_render<3, false>(ctx, kp, stream, &grid); break;
    case 7: rc = launch_render<3, true>(ctx, kp, stream, &grid); break;
    case 8: rc = launch_render<4, false>(ctx, kp, stream, &grid); break;
    case 9: rc = launch_render<4, true>(ctx, kp, stream, &grid); break;
#endif
    case 10:        // (the shipped kernel has a leaner instantiation for scenes whose tile grid has <= 64 cells)
        if (small_grid) {
            rc = large_blocks ? launch_render<5, false, true, false, rt::kItemBlockLarge>(ctx, kp, stream, &grid)
                              : launch_render<5, false, true>(ctx, kp, stream, &grid);
            ctx->last.kernel_variant = 1;
        } else rc = large_blocks ? launch_render<5, false, false, false, rt::kItemBlockLarge>(ctx, kp, stream, &grid)
                                 : launch_render<5, false>(ctx, kp, stream, &grid);
        if (large_blocks) ctx->last.kernel_variant |= 4;
        break;
    default: rc = launch_render<5, true>(ctx, kp, stream, &grid); break;
    }
    if (rc) return rc;
    ctx->launched = true;
    ctx->last.grid_blocks = grid;
    return RT_OK;
}

int rt_last_stats(rt_context *ctx, rt_stats *stats)
{
    if (!ctx || !stats) return fail(RT_ERR_INVALID_ARGUMENT, "ctx/stats is NULL");
    if (!ctx->launched) return fail(RT_ERR_INVALID_ARGUMENT, "no launch to report on");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.0f;
    RT_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    unsigned long long h[80];
    RT_HIP(hipMemcpy(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 64; ++k) ctx->last.live_per_bounce[k] = h[16 + k];
    ctx->last.rays_traced = h[0];
    ctx->last.samples = h[1] + ctx->zero_depth_samples;
    ctx->last.candidates = h[2];
    ctx->last.exact_roots = h[3];
    ctx->last.direct_samples = h[4];
    ctx->last.sphere_tests = h[0] * (unsigned long long)(ctx->last.n_spheres > 0 ? ctx->last.n_spheres : 0);
    ctx->last.kernel_ms = ms;
    *stats = ctx->last;
    return RT_OK;
}

#if defined(RT_PHASE_STAMPS) || defined(RT_BLOCK_COUNTS) || defined(RT_EXIT_TIMES) || defined(RT_LDS_CONFLICTS)
extern "C" int rt_debug_phase_cycles(rt_context *ctx, unsigned long long out[8])
{
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipEventSynchronize(ctx->ev1));
    RT_HIP(hipMemcpy(out, ctx->d_stats + 8, 64, hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_debug_phase_cycles16(rt_context *ctx, unsigned long long out[16])      // the 8 above + the finer split at stats[80..87]
{
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipEventSynchronize(ctx->ev1));
    RT_HIP(hipMemcpy(out, ctx->d_stats + 8, 64, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(out + 8, ctx->d_stats + 80, 64, hipMemcpyDeviceToHost));
    return RT_OK;
}
#endif

int rt_fix_to_f32_device(rt_context *ctx, const void *d_fix, int64_t count, void *d_out_f32, void *stream_v)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (count < 0 || (count > 0 && (!d_fix || !d_out_f32))) return fail(RT_ERR_INVALID_ARGUMENT, "bad buffers");
    if (count == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    long long blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rt::fix_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_v,
                       (const unsigned long long *)d_fix, (float *)d_out_f32, (long long)count);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_resolve_rgba8_device(rt_context *ctx, const void *d_fix, int32_t width, int32_t rows,
                            int64_t spp, int32_t flip, void *d_rgba, void *stream_v)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (width < 1 || rows < 0 || spp < 1) return fail(RT_ERR_INVALID_ARGUMENT, "bad width/rows/spp");
    if (rows == 0) return RT_OK;
    if (!d_fix || !d_rgba) return fail(RT_ERR_INVALID_ARGUMENT, "bad buffers");
    RT_HIP(hipSetDevice(ctx->device));
    const long long npix = (long long)width * rows;
    long long blocks = (npix + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const double scale = 1.0 / (double)spp;            // vec3.rs:409
    hipLaunchKernelGGL(rt::resolve_rgba8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_v,
                       (const unsigned long long *)d_fix, (uint8_t *)d_rgba, (int)width, (int)rows, scale, (int)flip);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_render(rt_context *ctx, const rt_camera *cam, const rt_params *p,
              float *out_sum, uint64_t *out_fix, rt_stats *stats)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    int rc = validate_params(p);
    if (rc) return rc;
    const int rows = shard_rows(p);
    const size_t count = (size_t)rows * p->width * 3;
    if (count > 0 && !out_sum && !out_fix) return fail(RT_ERR_INVALID_ARGUMENT, "no output buffer");
    RT_HIP(hipSetDevice(ctx->device));
    rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, count * sizeof(uint64_t));
    if (rc) return rc;
    rc = ensure(&ctx->d_stage_sum, &ctx->stage_sum_bytes, count * sizeof(float));
    if (rc) return rc;
    rt_params q = *p;
    q.flags &= ~RT_FLAG_ACCUMULATE;                     // host form always starts from zero
    rc = rt_render_device(ctx, cam, &q, ctx->d_stage_fix, ctx->own_stream);
    if (rc) return rc;
    if (out_sum) {
        rc = rt_fix_to_f32_device(ctx, ctx->d_stage_fix, (int64_t)count, ctx->d_stage_sum, ctx->own_stream);
        if (rc) return rc;
        RT_HIP(hipMemcpyAsync(out_sum, ctx->d_stage_sum, count * sizeof(float), hipMemcpyDeviceToHost, ctx->own_stream));
    }
    if (out_fix)
        RT_HIP(hipMemcpyAsync(out_fix, ctx->d_stage_fix, count * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    if (stats) return rt_last_stats(ctx, stats);
    return RT_OK;
}

int rt_resolve_rgba8(rt_context *ctx, const uint64_t *fix, int32_t width, int32_t rows,
                     int64_t spp, int32_t flip, uint8_t *out_rgba)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (width < 1 || rows < 0 || spp < 1) return fail(RT_ERR_INVALID_ARGUMENT, "bad width/rows/spp");
    if (rows == 0) return RT_OK;
    if (!fix || !out_rgba) return fail(RT_ERR_INVALID_ARGUMENT, "bad buffers");
    RT_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)width * rows;
    int rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, npix * 3 * sizeof(uint64_t));
    if (rc) return rc;
    rc = ensure(&ctx->d_stage_rgba, &ctx->stage_rgba_bytes, npix * 4);
    if (rc) return rc;
    RT_HIP(hipMemcpyAsync(ctx->d_stage_fix, fix, npix * 3 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->own_stream));
    rc = rt_resolve_rgba8_device(ctx, ctx->d_stage_fix, width, rows, spp, flip, ctx->d_stage_rgba, ctx->own_stream);
    if (rc) return rc;
    RT_HIP(hipMemcpyAsync(out_rgba, ctx->d_stage_rgba, npix * 4, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

// main.rs:122-145 in one call: the exact sums never leave the device; 4 bytes per pixel come back, straight into the caller's
// (pageable) buffer.  Measured at 1200x675 (profiles/r05_end_to_end.txt): wall - kernel = 0.16 ms with this plain copy, 0.27 ms
// through a pinned landing buffer + memcpy, 0.88 ms for rt_render + rt_resolve_rgba8 (the sums out and in again).
int rt_render_rgba8(rt_context *ctx, const rt_camera *cam, const rt_params *p, int32_t flip,
                    uint8_t *out_rgba, rt_stats *stats)
{
    if (!ctx) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    int rc = validate_params(p);
    if (rc) return rc;
    if (p->spp < 1) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_rgba8 needs spp >= 1 (to_rgba divides by the sample count, vec3.rs:409)");
    const int rows = shard_rows(p);
    const size_t npix = (size_t)rows * p->width;
    if (npix > 0 && !out_rgba) return fail(RT_ERR_INVALID_ARGUMENT, "out_rgba is NULL");
    RT_HIP(hipSetDevice(ctx->device));
    rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, npix * 3 * sizeof(uint64_t));
    if (rc) return rc;
    rc = ensure(&ctx->d_stage_rgba, &ctx->stage_rgba_bytes, npix * 4);
    if (rc) return rc;
    rt_params q = *p;
    q.flags &= ~RT_FLAG_ACCUMULATE;                     // host form always starts from zero
    rc = rt_render_device(ctx, cam, &q, ctx->d_stage_fix, ctx->own_stream);
    if (rc) return rc;
    if (npix > 0) {
        rc = rt_resolve_rgba8_device(ctx, ctx->d_stage_fix, p->width, rows, (int64_t)p->spp, flip, ctx->d_stage_rgba, ctx->own_stream);
        if (rc) return rc;
        RT_HIP(hipMemcpyAsync(out_rgba, ctx->d_stage_rgba, npix * 4, hipMemcpyDeviceToHost, ctx->own_stream));
        RT_HIP(hipStreamSynchronize(ctx->own_stream));
    } else {
        RT_HIP(hipStreamSynchronize(ctx->own_stream));
    }
    if (stats) return rt_last_stats(ctx, stats);
    return RT_OK;
}

int rt_f64_div_sqrt_device(rt_context *ctx, const double *a, const double *b, int32_t n,
                           double *out_div, double *out_sqrt)
{
    if (!ctx || !a || !b || !out_div || !out_sqrt || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * sizeof(double);
    int rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, 4 * bytes);
    if (rc) return rc;
    double *da = (double *)ctx->d_stage_fix, *db = da + n, *dq = db + n, *dr = dq + n;
    RT_HIP(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, ctx->own_stream));
    hipLaunchKernelGGL(rt::f64_div_sqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->own_stream,
                       (const double *)da, (const double *)db, (int)n, dq, dr);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(out_div, dq, bytes, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(out_sqrt, dr, bytes, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

int rt_quantize_device(rt_context *ctx, const double *x, int32_t n, uint64_t *out)
{
    if (!ctx || !x || !out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * sizeof(double);
    int rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, 2 * bytes);
    if (rc) return rc;
    double *dx = (double *)ctx->d_stage_fix;
    unsigned long long *dq = (unsigned long long *)(dx + n);
    RT_HIP(hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, ctx->own_stream));
    hipLaunchKernelGGL(rt::quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->own_stream, (const double *)dx, (int)n, dq);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(out, dq, bytes, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

int rt_unit_accept_device(rt_context *ctx, const uint32_t *words, int32_t n, uint32_t *out_accept, double *out_uniforms)
{
    if (!ctx || !words || !out_accept || !out_uniforms || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t wb = (size_t)n * 3 * sizeof(uint32_t), ab = (size_t)n * sizeof(uint32_t), ub = (size_t)n * 4 * sizeof(double);
    int rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, ub + wb + ab);
    if (rc) return rc;
    double *du = (double *)ctx->d_stage_fix;
    uint32_t *dw = (uint32_t *)(du + 4 * (size_t)n), *da = dw + 3 * (size_t)n;
    RT_HIP(hipMemcpyAsync(dw, words, wb, hipMemcpyHostToDevice, ctx->own_stream));
    hipLaunchKernelGGL(rt::unit_accept_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->own_stream, (const uint32_t *)dw, (int)n, da, du);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(out_accept, da, ab, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(out_uniforms, du, ub, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

#ifdef RTIOW_CROSSCHECK_MODES
#include "xcheck/rt_xcheck_hooks.inc"      // rt_filter_products_device, rt_filter_lifted_device
#endif

int rt_tube_tile_host(const rt_sphere *spheres32, uint32_t *out_words, float *out_bound, float *out_rho)
{
    if (!spheres32 || !out_words || !out_bound || !out_rho) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    const float rho = tube_radius_floor(spheres32, 32, nullptr);
    *out_rho = rho;
    const rt_sphere *col[32];
    for (int c = 0; c < 32; ++c) col[c] = &spheres32[c];
    uint4 tile[64];
    tube_tile(col, rho, tile, out_bound);
    memcpy(out_words, tile, sizeof(tile));
    return RT_OK;
}

int rt_filter_tube_device(rt_context *ctx, const double *o, const double *d, const rt_sphere *spheres32,
                          float *out_h, float *out_rows, float *out_bound, float *out_rho)
{
    if (!ctx || !o || !d || !spheres32 || !out_h || !out_rows || !out_bound || !out_rho)
        return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    RT_HIP(hipSetDevice(ctx->device));
    const float rho = tube_radius_floor(spheres32, 32, nullptr);
    *out_rho = rho;
    const rt_sphere *col[32];
    for (int c = 0; c < 32; ++c) col[c] = &spheres32[c];
    uint4 tile[64];
    tube_tile(col, rho, tile, out_bound);
    const size_t in_b = 3072 + sizeof(tile), out_b = (64 * 32 * 2 + 64 * 9) * 4;
    int rc = ensure(&ctx->d_stage_fix, &ctx->stage_fix_bytes, in_b + out_b);
    if (rc) return rc;
    char *base = (char *)ctx->d_stage_fix;
    double *d_o = (double *)base, *d_d = d_o + 192;
    uint4 *d_tile = (uint4 *)(base + 3072);
    float *d_h = (float *)(base + in_b), *d_rows = d_h + 64 * 32 * 2;
    RT_HIP(hipMemcpyAsync(d_o, o, 1536, hipMemcpyHostToDevice, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(d_d, d, 1536, hipMemcpyHostToDevice, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(d_tile, tile, sizeof(tile), hipMemcpyHostToDevice, ctx->own_stream));
    hipLaunchKernelGGL(rt::tube_products_kernel, dim3(1), dim3(64), 0, ctx->own_stream,
                       (const double *)d_o, (const double *)d_d, (const uint4 *)d_tile, rho, d_h, d_rows);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(out_h, d_h, 64 * 32 * 2 * 4, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipMemcpyAsync(out_rows, d_rows, 64 * 9 * 4, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

int rt_tile_layout_host(const rt_sphere *spheres, int32_t n, int32_t out_dims[2], float out_grid[8], int32_t *out_slot_of, int32_t cap)
{
    if (!spheres || n < 0 || n > RT_MAX_SPHERES || !out_dims || !out_grid || (!out_slot_of && cap > 0)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tile_layout_host: bad argument");
    std::vector<char> never(n > 0 ? n : 1, 0);
    for (int i : always_exact_list(spheres, n)) never[i] = 1;
    const TileLayout L = tile_layout(spheres, n, never.data());
    out_dims[0] = L.grid_dim; out_dims[1] = L.n_global;
    for (int k = 0; k < 8; ++k) out_grid[k] = L.grid[k];
    if ((size_t)cap < L.slot_of.size()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tile_layout_host: out_slot_of too small");
    for (size_t k = 0; k < L.slot_of.size(); ++k) out_slot_of[k] = L.slot_of[k];
    return (int)L.slot_of.size();
}

int rt_philox_device(rt_context *ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!ctx || !ctr || !key || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    RT_HIP(hipSetDevice(ctx->device));
    int rc = ensure(&ctx->d_stage_rgba, &ctx->stage_rgba_bytes, 64);
    if (rc) return rc;
    hipLaunchKernelGGL(rt::philox_kat_kernel, dim3(1), dim3(1), 0, ctx->own_stream,
                       ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], (uint32_t *)ctx->d_stage_rgba);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(out, ctx->d_stage_rgba, 16, hipMemcpyDeviceToHost, ctx->own_stream));
    RT_HIP(hipStreamSynchronize(ctx->own_stream));
    return RT_OK;
}

} // extern "C"

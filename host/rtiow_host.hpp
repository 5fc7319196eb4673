// rtiow_host.hpp -- compiled-language host side above the C ABI (include/rtiow_hip.h).
//
// The reference is a Rust binary; with no Rust toolchain in the image the host mirror is C++.
// It keeps the reference's interface for this path -- Vec3/Point3/Color, Camera::new,
// Sphere, the Scatter materials (Lambertian, Metal, Dialectric -- reference spelling),
// HittableList with push(), random_scene() -- and adds the one thing the reference lacks:
// every object can flatten itself into the rt_sphere record the GPU path consumes.
// Nothing here computes radiance.  (rtiow_amd/scene.py is the same mirror in Python; the two
// build bit-identical scenes, tests/test_host_cpp.py.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

#include "rtiow_hip.h"

namespace rtiow {

struct Vec3 {                                   // src/vec3.rs:4-9
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double length_squared() const { return x * x + y * y + z * z; }                    // :87-89
    double length() const { return std::sqrt(length_squared()); }                      // :83-85
    Vec3 cross(const Vec3 &r) const { return {y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x}; }   // :99-105
    Vec3 unit_vector() const { const double s = 1.0 / length(); return {x * s, y * s, z * s}; }  // :107-109, 371-375
};
using Point3 = Vec3;
using Color = Vec3;
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return a * (1.0 / s); }                       // :371-375

struct Camera {                                 // src/camera.rs:4-45
    Point3 origin, lower_left_corner;
    Vec3 horizontal, vertical, u, v, w;
    double lens_radius = 0;
    Camera(Point3 look_from, Point3 look_at, Vec3 v_up, double v_fov, double aspect_ratio,
           double aperture, double focus_dist)
    {
        const double theta = v_fov * (3.14159265358979323846 / 180.0);   // f64::to_radians
        const double viewport_height = 2.0 * std::tan(theta / 2.0);
        const double viewport_width = aspect_ratio * viewport_height;
        w = (look_from - look_at).unit_vector();
        u = v_up.cross(w).unit_vector();
        v = w.cross(u);
        origin = look_from;
        horizontal = (focus_dist * viewport_width) * u;
        vertical = (focus_dist * viewport_height) * v;
        lower_left_corner = ((look_from - horizontal / 2.0) - vertical / 2.0) - focus_dist * w;
        lens_radius = aperture / 2.0;
    }
    rt_camera flat() const
    {
        rt_camera c{};
        auto put = [](double *d, const Vec3 &s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
        put(c.origin, origin); put(c.lower_left_corner, lower_left_corner);
        put(c.horizontal, horizontal); put(c.vertical, vertical); put(c.u, u); put(c.v, v);
        c.lens_radius = lens_radius;
        return c;
    }
};

struct Scatter {                                // src/materials.rs:5-7 (parameters only: scatter() runs on the GPU)
    virtual ~Scatter() = default;
    virtual void flat(rt_sphere &out) const = 0;
};
struct Lambertian : Scatter {                   // :9-19
    Color albedo;
    explicit Lambertian(Color a) : albedo(a) {}
    void flat(rt_sphere &o) const override { o.kind = RT_LAMBERTIAN; o.albedo[0] = albedo.x; o.albedo[1] = albedo.y; o.albedo[2] = albedo.z; o.param = 0.0; }
};
struct Metal : Scatter {                        // :34-46
    Color albedo; double fuzz;
    Metal(Color a, double f) : albedo(a), fuzz(f) {}
    void flat(rt_sphere &o) const override { o.kind = RT_METAL; o.albedo[0] = albedo.x; o.albedo[1] = albedo.y; o.albedo[2] = albedo.z; o.param = fuzz; }
};
struct Dialectric : Scatter {                   // :64-74
    double ir;
    explicit Dialectric(double index_of_refraction) : ir(index_of_refraction) {}
    void flat(rt_sphere &o) const override { o.kind = RT_DIALECTRIC; o.albedo[0] = o.albedo[1] = o.albedo[2] = 0.0; o.param = ir; }
};

struct Sphere {                                 // src/shapes/sphere.rs:9-13,44-52
    Point3 center; double radius; std::shared_ptr<Scatter> mat;
    Sphere(Point3 cen, double r, std::shared_ptr<Scatter> m) : center(cen), radius(r), mat(std::move(m)) {}
};

struct HittableList {                           // src/shapes/mod.rs:52; push keeps order (ties: later wins)
    std::vector<Sphere> objects;
    void push(Sphere s) { objects.push_back(std::move(s)); }
    std::vector<rt_sphere> flatten() const
    {
        std::vector<rt_sphere> out(objects.size());
        for (size_t i = 0; i < objects.size(); ++i) {
            rt_sphere r{};
            r.center[0] = objects[i].center.x; r.center[1] = objects[i].center.y; r.center[2] = objects[i].center.z;
            r.radius = objects[i].radius;
            objects[i].mat->flat(r);
            out[i] = r;
        }
        return out;
    }
};

// Seeded uniform stream for the scene builder: Philox4x32-10, counter (block, 0, 0, 0x5CE9E),
// u = (word >> 8) * 2^-24 -- identical to rtiow_amd/philox.py.
class UniformStream {
public:
    explicit UniformStream(uint64_t seed) : k0_((uint32_t)seed), k1_((uint32_t)(seed >> 32)) {}
    double next()
    {
        if (pos_ >= 4) { refill(); pos_ = 0; }
        return (double)(w_[pos_++] >> 8) * (1.0 / 16777216.0);
    }
private:
    void refill()
    {
        uint32_t c0 = block_++, c1 = 0, c2 = 0, c3 = 0x5CE9E, k0 = k0_, k1 = k1_;
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        w_[0] = c0; w_[1] = c1; w_[2] = c2; w_[3] = c3;
    }
    uint32_t k0_, k1_, block_ = 0, w_[4] = {0, 0, 0, 0};
    int pos_ = 4;
};

// src/main.rs:59-102 with a seeded stream; grid (-11,11) is the reference's 23x23 lattice.
inline HittableList random_scene(uint64_t seed = 1, int lo = -11, int hi = 11)
{
    UniformStream rng(seed);
    HittableList world;
    world.push(Sphere(Point3(0, -1000, 0), 1000, std::make_shared<Lambertian>(Color(0.5, 0.5, 0.5))));
    for (int a = lo; a <= hi; ++a)
        for (int b = lo; b <= hi; ++b) {
            const double a_prime = (double)a + (0.9 * rng.next());
            const double b_prime = (double)b + (0.9 * rng.next());
            const Point3 center(a_prime, 0.2, b_prime);
            if ((center - Point3(4, 0.2, 0)).length() > 0.9) {
                const double x = rng.next();
                std::shared_ptr<Scatter> m;
                if (x >= 0.0 && x <= 0.8) {
                    // Vec3 {x: rng.gen(), y: rng.gen(), z: rng.gen()} draws in field order
                    // (vec3.rs:21-24); C++ leaves argument evaluation order open, so draw first.
                    const double x1 = rng.next(), y1 = rng.next(), z1 = rng.next();
                    const double x2 = rng.next(), y2 = rng.next(), z2 = rng.next();
                    m = std::make_shared<Lambertian>(Color(x1, y1, z1) * Color(x2, y2, z2));
                } else if (x >= 0.8 && x <= 0.95) {
                    const double ux = rng.next(), uy = rng.next(), uz = rng.next();
                    const Color albedo(0.5 + 0.5 * ux, 0.5 + 0.5 * uy, 0.5 + 0.5 * uz);
                    const double fuzz = 0.5 * rng.next();
                    m = std::make_shared<Metal>(albedo, fuzz);
                } else {
                    m = std::make_shared<Dialectric>(1.5);
                }
                world.push(Sphere(center, 0.2, m));
            }
        }
    world.push(Sphere(Point3(0, 1, 0), 1.0, std::make_shared<Dialectric>(1.5)));
    world.push(Sphere(Point3(-4, 1, 0), 1.0, std::make_shared<Lambertian>(Color(0.4, 0.2, 0.1))));
    world.push(Sphere(Point3(4, 1, 0), 1.0, std::make_shared<Metal>(Color(0.7, 0.6, 0.5), 0.0)));
    return world;
}

// ---- output stage: main.rs:147,177 `image_buffer.save("image.png")` ------------------------------------------------
// An 8-bit RGBA PNG, rows top first (the order main.rs:141-145 produces), alpha as to_rgba() set it (255): the colour
// type, bit depth and pixels of the file the reference saves through the `image` crate.  No zlib in the build: the
// IDAT stream is zlib-framed DEFLATE in STORED blocks (RFC 1950/1951: legal for every decoder, 5 bytes of overhead per
// 65 535 bytes), the CRC-32 of each chunk and the Adler-32 of the stream are computed here.
inline uint32_t png_crc32(uint32_t crc, const unsigned char *p, size_t n)
{
    struct Table { uint32_t t[256]; };
    static const Table tab = [] {                                       // (function-local static: initialised once, thread-safe)
        Table x{};
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            x.t[i] = c;
        }
        return x;
    }();
    const uint32_t *table = tab.t;
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

inline bool write_png(const char *path, const unsigned char *rgba_top_first, int width, int height)
{
    if (width < 1 || height < 1) return false;
    auto be32 = [](std::vector<unsigned char> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); };
    // the filtered image: a filter-type byte (0 = None) before each row
    const size_t row = 4 * (size_t)width;
    std::vector<unsigned char> raw;
    raw.reserve((row + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba_top_first + (size_t)y * row, rgba_top_first + ((size_t)y + 1) * row);
    }
    // zlib stream: CMF/FLG, stored blocks, Adler-32 (big-endian)
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);                     // BFINAL, BTYPE = 00
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = pos; i < pos + n; ++i) { a += raw[i]; if (a >= 65521u) a -= 65521u; b += a; if (b >= 65521u) b -= 65521u; }
        pos += n;
    }
    be32(z, (b << 16) | a);
    if (z.size() > 0x7fffffffu) return false;                           // a PNG chunk's length field holds at most 2^31 - 1 (one IDAT chunk here)
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    auto chunk = [&](const char tag[4], const std::vector<unsigned char> &body) {
        std::vector<unsigned char> c;
        be32(c, (uint32_t)body.size());
        c.insert(c.end(), tag, tag + 4);
        c.insert(c.end(), body.begin(), body.end());
        const uint32_t crc = png_crc32(0u, c.data() + 4, c.size() - 4);
        be32(c, crc);
        return std::fwrite(c.data(), 1, c.size(), f) == c.size();
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // depth 8, RGBA, deflate, adaptive, no interlace
    ok = ok && chunk("IHDR", ihdr) && chunk("IDAT", z) && chunk("IEND", {});
    return (std::fclose(f) == 0) && ok;
}

} // namespace rtiow

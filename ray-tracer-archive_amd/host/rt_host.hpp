// rt_host.hpp — host side above the C ABI: a C++ mirror of the reference's scene-construction API.
//
// The reference's host code is Rust (raytracer/src); no Rust toolchain exists in this image, so the
// host layer a Rust user would keep — Camera, HittableList/BVHNode, the primitive / material /
// texture constructors, the scene functions of main.rs and write_color — is mirrored here with the
// same names, argument order and meaning.  Objects are shared_ptr graphs exactly like the
// reference's Arc<dyn Hittable>; flatten() serialises a graph into the RtSceneDesc the C ABI takes
// (include/rt_hip.h).  Nothing in this file touches the GPU.
#pragma once

#include "../../include/rt_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace rt {

constexpr double PI = 3.14159265358979323846264338327950288;   // rt_weekend.rs:2
inline double degrees_to_radians(double d) { return d * PI / 180.0; }   // rt_weekend.rs:4-6

// Host-side RNG for scene construction (replaces rand::random in the scene functions and in
// Perlin::new): SplitMix64, sequential, seeded per scene.
struct SceneRng {
    uint64_t state;
    explicit SceneRng(uint64_t seed) : state(seed) {}
    static uint64_t fin(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t next64() { state += 0x9E3779B97F4A7C15ull; return fin(state); }
    double random_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }        // rt_weekend.rs:8
    double random_double_range(double lo, double hi) { return lo + (hi - lo) * random_double(); }     // rt_weekend.rs:13
    uint32_t random_int(uint32_t lo, uint32_t hi) { return lo + (uint32_t)std::floor(random_double() * (double)(hi - lo + 1)); }   // rt_weekend.rs:16
};

// vec3.rs
struct Vec3 {
    double e[3];
    Vec3() : e{0, 0, 0} {}
    Vec3(double a, double b, double c) : e{a, b, c} {}
    double x() const { return e[0]; } double y() const { return e[1]; } double z() const { return e[2]; }
    double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    double length() const { return std::sqrt(length_squared()); }
    Vec3 unit() const { double l = length(); return Vec3(e[0] / l, e[1] / l, e[2] / l); }
    Vec3 operator+(const Vec3& o) const { return Vec3(e[0] + o.e[0], e[1] + o.e[1], e[2] + o.e[2]); }
    Vec3 operator-(const Vec3& o) const { return Vec3(e[0] - o.e[0], e[1] - o.e[1], e[2] - o.e[2]); }
    Vec3 operator*(const Vec3& o) const { return Vec3(e[0] * o.e[0], e[1] * o.e[1], e[2] * o.e[2]); }
    Vec3 operator*(double t) const { return Vec3(e[0] * t, e[1] * t, e[2] * t); }
    Vec3 operator/(double t) const { return Vec3(e[0] / t, e[1] / t, e[2] / t); }
    static Vec3 random(SceneRng& g) { double a = g.random_double(), b = g.random_double(), c = g.random_double(); return Vec3(a, b, c); }   // vec3.rs:48
    static Vec3 random_range(SceneRng& g, double lo, double hi) {                                                                        // vec3.rs:53
        double a = g.random_double_range(lo, hi), b = g.random_double_range(lo, hi), c = g.random_double_range(lo, hi);
        return Vec3(a, b, c);
    }
    RtVec3 abi() const { return RtVec3{e[0], e[1], e[2]}; }
};
inline Vec3 operator*(double t, const Vec3& v) { return v * t; }
inline Vec3 cross(const Vec3& u, const Vec3& v) {
    return Vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], -(u.e[0] * v.e[2] - u.e[2] * v.e[0]), u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
using Point3 = Vec3;
using Color3 = Vec3;

// camera.rs
struct Camera {
    Point3 origin, lower_left_corner; Vec3 horizontal, vertical, u, v, w; double lens_radius = 0, time0 = 0, time1 = 0;
    // Camera::new(lookfrom, lookat, vup, &[vfov, aspect_ratio, aperture, focus_dist], time0, time1)  camera.rs:21-59
    static Camera construct(const Point3& lookfrom, const Point3& lookat, const Vec3& vup, const double scope[4], double time0, double time1) {
        const double vfov = scope[0], aspect_ratio = scope[1], aperture = scope[2], focus_dist = scope[3];
        const double theta = degrees_to_radians(vfov);
        const double h = std::tan(theta / 2.0);
        const double viewport_height = 2.0 * h;
        const double viewport_width = aspect_ratio * viewport_height;
        Camera c;
        c.w = (lookfrom - lookat).unit();
        c.u = cross(vup, c.w).unit();
        c.v = cross(c.w, c.u);
        c.origin = lookfrom;
        c.horizontal = focus_dist * viewport_width * c.u;
        c.vertical = focus_dist * viewport_height * c.v;
        c.lower_left_corner = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focus_dist * c.w;
        c.lens_radius = aperture / 2.0;
        c.time0 = time0; c.time1 = time1;
        return c;
    }
    RtCamera abi() const {
        RtCamera o;
        o.origin = origin.abi(); o.lower_left_corner = lower_left_corner.abi(); o.horizontal = horizontal.abi(); o.vertical = vertical.abi();
        o.u = u.abi(); o.v = v.abi(); o.w = w.abi(); o.lens_radius = lens_radius; o.time0 = time0; o.time1 = time1;
        return o;
    }
};

// perlin.rs:14-25, 53-66 — table construction (the lookups run on the device)
struct Perlin {
    RtPerlin t;
    static std::shared_ptr<Perlin> construct(SceneRng& g) {
        auto p = std::make_shared<Perlin>();
        for (int i = 0; i < 256; ++i) { Vec3 v = Vec3::random_range(g, -1.0, 1.0).unit(); p->t.ranvec[i][0] = v.e[0]; p->t.ranvec[i][1] = v.e[1]; p->t.ranvec[i][2] = v.e[2]; }
        generate_perm(g, p->t.perm_x); generate_perm(g, p->t.perm_y); generate_perm(g, p->t.perm_z);
        return p;
    }
    static void generate_perm(SceneRng& g, uint32_t* p) {
        for (uint32_t i = 0; i < 256; ++i) p[i] = i;
        for (uint32_t i = 255; i >= 1; --i) { uint32_t target = g.random_int(0, i); uint32_t tmp = p[i]; p[i] = p[target]; p[target] = tmp; }
    }
};

// texture.rs
struct Texture { RtTexture rec{}; std::shared_ptr<Texture> even, odd; std::shared_ptr<Perlin> noise; std::shared_ptr<std::vector<uint8_t>> data; uint32_t width = 0, height = 0; virtual ~Texture() {} };
struct SolidColor { static std::shared_ptr<Texture> construct(const Color3& c) { auto t = std::make_shared<Texture>(); t->rec.kind = RT_TEX_SOLID; t->rec.color = c.abi(); return t; } };   // texture.rs:22
struct CheckerTexture { static std::shared_ptr<Texture> construct_color(const Color3& c1, const Color3& c2) {                                                                              // texture.rs:52
    auto t = std::make_shared<Texture>(); t->rec.kind = RT_TEX_CHECKER; t->even = SolidColor::construct(c1); t->odd = SolidColor::construct(c2); return t; } };
struct NoiseTexture { static std::shared_ptr<Texture> construct(double scale, SceneRng& g) {                                                                                              // texture.rs:83
    auto t = std::make_shared<Texture>(); t->rec.kind = RT_TEX_NOISE; t->rec.scale = scale; t->noise = Perlin::construct(g); return t; } };
struct ImageTexture { static std::shared_ptr<Texture> construct(const uint8_t* data, uint32_t width, uint32_t height) {                                                                    // texture.rs:108
    auto t = std::make_shared<Texture>(); t->rec.kind = RT_TEX_IMAGE; t->width = width; t->height = height;
    t->data = std::make_shared<std::vector<uint8_t>>(data ? data : nullptr, data ? data + (size_t)width * height * 3 : nullptr); return t; } };

// material.rs
struct Material { RtMaterial rec{}; std::shared_ptr<Texture> tex; };
struct Lambertian {
    static std::shared_ptr<Material> construct(const Color3& a) { return construct_texture(SolidColor::construct(a)); }                       // material.rs:35
    static std::shared_ptr<Material> construct_texture(std::shared_ptr<Texture> a) { auto m = std::make_shared<Material>(); m->rec.kind = RT_MAT_LAMBERTIAN; m->tex = a; return m; }   // material.rs:40
};
struct Metal { static std::shared_ptr<Material> construct(const Color3& albedo, double fuzz) {                                                 // material.rs:87
    auto m = std::make_shared<Material>(); m->rec.kind = RT_MAT_METAL; m->rec.albedo = albedo.abi(); m->rec.fuzz = fuzz < 1.0 ? fuzz : 1.0; return m; } };
struct Dielectric { static std::shared_ptr<Material> construct(double ir) { auto m = std::make_shared<Material>(); m->rec.kind = RT_MAT_DIELECTRIC; m->rec.ir = ir; return m; } };   // material.rs:119
struct DiffuseLight { static std::shared_ptr<Material> construct_color(const Color3& c) {                                                      // material.rs:168
    auto m = std::make_shared<Material>(); m->rec.kind = RT_MAT_DIFFUSE_LIGHT; m->tex = SolidColor::construct(c); return m; } };
struct Isotropic { static std::shared_ptr<Material> construct_color(const Color3& c) {                                                         // material.rs:202 (commented)
    auto m = std::make_shared<Material>(); m->rec.kind = RT_MAT_ISOTROPIC; m->tex = SolidColor::construct(c); return m; } };

// hittable.rs:51 — one node of the Arc<dyn Hittable> graph
struct Hittable {
    int32_t kind = 0; double p[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::shared_ptr<Material> mat; std::vector<std::shared_ptr<Hittable>> children;
};
using HittablePtr = std::shared_ptr<Hittable>;
inline HittablePtr make_hittable(int32_t kind) { auto h = std::make_shared<Hittable>(); h->kind = kind; return h; }

struct Sphere { static HittablePtr construct(const Point3& c, double r, std::shared_ptr<Material> m) {                                       // sphere.rs:25
    auto h = make_hittable(RT_HIT_SPHERE); h->p[0] = c.e[0]; h->p[1] = c.e[1]; h->p[2] = c.e[2]; h->p[3] = r; h->mat = m; return h; } };
struct MovingSphere { static HittablePtr construct(const Point3& c0, const Point3& c1, double t0, double t1, double r, std::shared_ptr<Material> m) {   // moving_sphere.rs:19
    auto h = make_hittable(RT_HIT_MOVING_SPHERE); for (int i = 0; i < 3; ++i) { h->p[i] = c0.e[i]; h->p[3 + i] = c1.e[i]; } h->p[6] = t0; h->p[7] = t1; h->p[8] = r; h->mat = m; return h; } };
inline HittablePtr make_rect(int32_t kind, double a0, double a1, double b0, double b1, double k, std::shared_ptr<Material> m) {
    auto h = make_hittable(kind); h->p[0] = a0; h->p[1] = a1; h->p[2] = b0; h->p[3] = b1; h->p[4] = k; h->mat = m; return h;
}
struct XyRect { static HittablePtr construct(double x0, double x1, double y0, double y1, double k, std::shared_ptr<Material> m) { return make_rect(RT_HIT_XY_RECT, x0, x1, y0, y1, k, m); } };   // aarect.rs:19
struct XzRect { static HittablePtr construct(double x0, double x1, double z0, double z1, double k, std::shared_ptr<Material> m) { return make_rect(RT_HIT_XZ_RECT, x0, x1, z0, z1, k, m); } };   // aarect.rs:69
struct YzRect { static HittablePtr construct(double y0, double y1, double z0, double z1, double k, std::shared_ptr<Material> m) { return make_rect(RT_HIT_YZ_RECT, y0, y1, z0, z1, k, m); } };   // aarect.rs:138
struct Triangle { static HittablePtr construct(const Point3& a, const Point3& b, const Point3& c, std::shared_ptr<Material> m) {            // not in the reference
    auto h = make_hittable(RT_HIT_TRIANGLE); for (int i = 0; i < 3; ++i) { h->p[i] = a.e[i]; h->p[3 + i] = b.e[i]; h->p[6 + i] = c.e[i]; } h->mat = m; return h; } };
struct Box { static HittablePtr construct(const Point3& p0, const Point3& p1, std::shared_ptr<Material> m) {                                // boxes.rs:17
    auto h = make_hittable(RT_HIT_BOX); for (int i = 0; i < 3; ++i) { h->p[i] = p0.e[i]; h->p[3 + i] = p1.e[i]; } h->mat = m; return h; } };
struct Translate { static HittablePtr construct(HittablePtr p, const Vec3& displacement) {                                                   // hittable.rs:68
    auto h = make_hittable(RT_HIT_TRANSLATE); h->children.push_back(p); h->p[0] = displacement.e[0]; h->p[1] = displacement.e[1]; h->p[2] = displacement.e[2]; return h; } };
struct RotateY { static HittablePtr construct(HittablePtr p, double angle) { auto h = make_hittable(RT_HIT_ROTATE_Y); h->children.push_back(p); h->p[0] = angle; return h; } };   // hittable.rs:107
struct FlipFace { static HittablePtr construct(HittablePtr p) { auto h = make_hittable(RT_HIT_FLIP_FACE); h->children.push_back(p); return h; } };                                   // hittable.rs:188
struct ConstantMedium { static HittablePtr construct_color(HittablePtr b, double d, const Color3& c) {                                       // constant_medium.rs:22 (commented)
    auto h = make_hittable(RT_HIT_CONSTANT_MEDIUM); h->children.push_back(b); h->p[0] = d; h->mat = Isotropic::construct_color(c); return h; } };

// hittable_list.rs:17-35
struct HittableList {
    HittablePtr node;
    HittableList() : node(make_hittable(RT_HIT_LIST)) {}
    static HittableList construct(HittablePtr object) { HittableList l; l.add(object); return l; }
    void add(HittablePtr object) { node->children.push_back(object); }
    size_t len() const { return node->children.size(); }
    operator HittablePtr() const { return node; }
};
// bvh.rs:74 BVHNode::construct2(list, time0, time1). The tree itself is built where the graph is
// compiled (the device library, csrc/scene_compile.cpp — intended semantics).
struct BVHNode { static HittablePtr construct2(const HittableList& list, double time0, double time1) {
    auto h = make_hittable(RT_HIT_BVH); h->children = list.node->children; h->p[0] = time0; h->p[1] = time1; return h; } };

// main.rs:141-169
inline void write_color(const double pixel_color[3], uint32_t samples_per_pixel, uint8_t out[3]) {
    for (int i = 0; i < 3; ++i) {
        double c = pixel_color[i];
        if (c != c) c = 0.0;                                   // Replace NaN (on the SUM)
        const double scale = 1.0 / (double)samples_per_pixel;
        c = std::sqrt(scale * c);
        c = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);           // rt_weekend::clamp
        const double q = 256.0 * c;
        out[i] = q != q ? 0 : (uint8_t)q;                        // Rust `as u8`: saturating, NaN -> 0
    }
}

// ------------------------------------------------------------------------------------------------
// flatten: Arc graph -> RtSceneDesc (owning storage)
// ------------------------------------------------------------------------------------------------
struct FlatScene {
    std::vector<RtHittable> hittables; std::vector<int32_t> children; std::vector<RtMaterial> materials; std::vector<RtTexture> textures;
    std::vector<RtPerlin> perlins; std::vector<RtImage> images; std::vector<std::shared_ptr<std::vector<uint8_t>>> image_data;
    RtSceneDesc desc{};
    std::map<const Hittable*, int32_t> hid; std::map<const Material*, int32_t> mid; std::map<const Texture*, int32_t> tid; std::map<const Perlin*, int32_t> pid;

    int32_t add_texture(const std::shared_ptr<Texture>& t) {
        auto it = tid.find(t.get()); if (it != tid.end()) return it->second;
        RtTexture r = t->rec; r.a = -1; r.b = -1;
        if (r.kind == RT_TEX_CHECKER) { r.a = add_texture(t->even); r.b = add_texture(t->odd); }
        else if (r.kind == RT_TEX_NOISE) {
            auto pi = pid.find(t->noise.get());
            if (pi == pid.end()) { perlins.push_back(t->noise->t); pid[t->noise.get()] = (int32_t)perlins.size() - 1; r.a = (int32_t)perlins.size() - 1; } else r.a = pi->second;
        } else if (r.kind == RT_TEX_IMAGE) {
            if (t->data && !t->data->empty()) { image_data.push_back(t->data); images.push_back(RtImage{t->data->data(), t->width, t->height}); r.a = (int32_t)images.size() - 1; }
        }
        textures.push_back(r); tid[t.get()] = (int32_t)textures.size() - 1;
        return (int32_t)textures.size() - 1;
    }
    int32_t add_material(const std::shared_ptr<Material>& m) {
        if (!m) return -1;
        auto it = mid.find(m.get()); if (it != mid.end()) return it->second;
        RtMaterial r = m->rec; r.texture = m->tex ? add_texture(m->tex) : -1;
        materials.push_back(r); mid[m.get()] = (int32_t)materials.size() - 1;
        return (int32_t)materials.size() - 1;
    }
    int32_t add_hittable(const HittablePtr& h) {
        auto it = hid.find(h.get()); if (it != hid.end()) return it->second;   // a shared Arc keeps one id
        std::vector<int32_t> kids; for (auto& c : h->children) kids.push_back(add_hittable(c));
        RtHittable r{}; r.kind = h->kind; r.material = add_material(h->mat); for (int i = 0; i < 10; ++i) r.p[i] = h->p[i];
        if (h->kind == RT_HIT_LIST || h->kind == RT_HIT_BVH) { r.first_child = (int32_t)children.size(); r.n_children = (int32_t)kids.size(); children.insert(children.end(), kids.begin(), kids.end()); }
        else if (!kids.empty()) { r.first_child = kids[0]; r.n_children = 1; }
        else { r.first_child = -1; r.n_children = 0; }
        hittables.push_back(r); hid[h.get()] = (int32_t)hittables.size() - 1;
        return (int32_t)hittables.size() - 1;
    }
    void finish(const HittablePtr& world, const HittablePtr& lights, int background_mode, const Color3& background, uint64_t bvh_seed,
                int bvh_builder = RT_BVH_REFERENCE) {
        desc.abi_version = RT_ABI_VERSION;
        desc.world = add_hittable(world);
        desc.lights = lights ? add_hittable(lights) : -1;
        desc.hittables = hittables.data(); desc.n_hittables = hittables.size();
        desc.children = children.data(); desc.n_children = children.size();
        desc.materials = materials.data(); desc.n_materials = materials.size();
        desc.textures = textures.data(); desc.n_textures = textures.size();
        desc.perlins = perlins.data(); desc.n_perlins = perlins.size();
        desc.images = images.data(); desc.n_images = images.size();
        desc.background_mode = background_mode; desc.background = background.abi(); desc.bvh_seed = bvh_seed; desc.bvh_builder = bvh_builder;
    }
};

// ------------------------------------------------------------------------------------------------
// Scene functions of main.rs (most are commented out at the reference's HEAD; restated here)
// ------------------------------------------------------------------------------------------------
struct SceneRecipe {
    HittablePtr world, lights; int background_mode = RT_BG_CONSTANT; Color3 background;
    Point3 lookfrom, lookat; Vec3 vup{0, 1, 0}; double vfov = 40, aperture = 0, focus_dist = 10, time0 = 0, time1 = 1;
    Camera camera(double aspect_ratio) const { const double scope[4] = {vfov, aspect_ratio, aperture, focus_dist}; return Camera::construct(lookfrom, lookat, vup, scope, time0, time1); }
};

// Book-1 final scene (canonical book definition; reference remnants main.rs:171-242, camera :706-709).
// variant 0: solid ground, static Lambertian/Metal/Dielectric spheres, sky gradient (BASELINE C1/C2).
// variant 1: the reference's remnant — checker ground and diffuse-only MovingSpheres (main.rs:180-218).
inline SceneRecipe random_scene(uint64_t scene_seed, int variant, bool use_bvh) {
    SceneRng g(scene_seed);
    HittableList world;
    if (variant == 1) {
        auto checker = CheckerTexture::construct_color(Color3(0.2, 0.3, 0.1), Color3(0.9, 0.9, 0.9));
        world.add(Sphere::construct(Point3(0, -1000, 0), 1000, Lambertian::construct_texture(checker)));
    } else {
        world.add(Sphere::construct(Point3(0, -1000, 0), 1000, Lambertian::construct(Color3(0.5, 0.5, 0.5))));
    }
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            const double choose_mat = g.random_double();
            const double cx = (double)a + 0.9 * g.random_double();
            const double cz = (double)b + 0.9 * g.random_double();
            const Point3 center(cx, 0.2, cz);
            if ((center - Point3(4, 0.2, 0)).length() > 0.9) {
                if (choose_mat < 0.8) {
                    const Color3 c1 = Color3::random(g), c2 = Color3::random(g);   // main.rs:203, left operand first
                    const Color3 albedo = c1 * c2;
                    auto m = Lambertian::construct(albedo);
                    if (variant == 1) {
                        const Point3 center2 = center + Vec3(0, g.random_double_range(0, 0.5), 0);
                        world.add(MovingSphere::construct(center, center2, 0.0, 1.0, 0.2, m));
                    } else world.add(Sphere::construct(center, 0.2, m));
                } else if (variant == 0) {
                    if (choose_mat < 0.95) {
                        const Color3 albedo = Color3::random_range(g, 0.5, 1);
                        const double fuzz = g.random_double_range(0, 0.5);
                        world.add(Sphere::construct(center, 0.2, Metal::construct(albedo, fuzz)));
                    } else world.add(Sphere::construct(center, 0.2, Dielectric::construct(1.5)));
                }
            }
        }
    world.add(Sphere::construct(Point3(0, 1, 0), 1.0, Dielectric::construct(1.5)));
    world.add(Sphere::construct(Point3(-4, 1, 0), 1.0, Lambertian::construct(Color3(0.4, 0.2, 0.1))));
    world.add(Sphere::construct(Point3(4, 1, 0), 1.0, Metal::construct(Color3(0.7, 0.6, 0.5), 0.0)));
    SceneRecipe r;
    r.world = use_bvh ? BVHNode::construct2(world, 0.0, variant == 1 ? 1.0 : 0.0) : (HittablePtr)world;
    r.background_mode = RT_BG_SKY_GRADIENT; r.background = Color3(0.5, 0.7, 1.0);
    r.lookfrom = Point3(13, 2, 3); r.lookat = Point3(0, 0, 0); r.vfov = 20; r.aperture = 0.1; r.focus_dist = 10;
    r.time0 = 0; r.time1 = variant == 1 ? 1.0 : 0.0;
    return r;
}

// main.rs:337-433 cornell_box + lights main.rs:669-684 + camera main.rs:688-718
inline SceneRecipe cornell_box() {
    HittableList objects;
    auto red = Lambertian::construct(Color3(0.65, 0.05, 0.05));
    auto white = Lambertian::construct(Color3(0.73, 0.73, 0.73));
    auto green = Lambertian::construct(Color3(0.12, 0.45, 0.15));
    auto light = DiffuseLight::construct_color(Color3(15, 15, 15));
    objects.add(YzRect::construct(0, 555, 0, 555, 555, green));
    objects.add(YzRect::construct(0, 555, 0, 555, 0, red));
    objects.add(FlipFace::construct(XzRect::construct(213, 343, 227, 332, 554, light)));
    objects.add(XzRect::construct(0, 555, 0, 555, 0, white));
    objects.add(XzRect::construct(0, 555, 0, 555, 555, white));
    objects.add(XyRect::construct(0, 555, 0, 555, 555, white));
    HittablePtr box1 = Box::construct(Point3(0, 0, 0), Point3(165, 330, 165), white);
    box1 = RotateY::construct(box1, 15.0);
    box1 = Translate::construct(box1, Vec3(265, 0, 295));
    objects.add(box1);
    objects.add(Sphere::construct(Point3(190, 90, 190), 90, Dielectric::construct(1.5)));
    HittableList lights;
    lights.add(XzRect::construct(213, 343, 227, 332, 554, DiffuseLight::construct_color(Color3(15, 15, 15))));
    lights.add(Sphere::construct(Point3(190, 90, 190), 90, Dielectric::construct(1.5)));
    SceneRecipe r;
    r.world = objects; r.lights = lights;
    r.background_mode = RT_BG_CONSTANT; r.background = Color3(0, 0, 0);
    r.lookfrom = Point3(278, 278, -800); r.lookat = Point3(278, 278, 0); r.vfov = 40; r.aperture = 0; r.focus_dist = 10; r.time0 = 0; r.time1 = 1;
    return r;
}

// main.rs:435-519 cornell_smoke (commented in the reference). `lit`: the light wrapped in FlipFace as cornell_box does (main.rs:359-361) —
// under HEAD's DiffuseLight::emitted (material.rs:184-190, front face only) the literal scene's light faces the ceiling and the room
// stays dark; the lit twin is the parity frame that carries signal. Nothing else differs.
inline SceneRecipe cornell_smoke(bool lit = false) {
    HittableList objects;
    auto red = Lambertian::construct(Color3(0.65, 0.05, 0.05));
    auto white = Lambertian::construct(Color3(0.73, 0.73, 0.73));
    auto green = Lambertian::construct(Color3(0.12, 0.45, 0.15));
    auto light = DiffuseLight::construct_color(Color3(7, 7, 7));
    objects.add(YzRect::construct(0, 555, 0, 555, 555, green));
    objects.add(YzRect::construct(0, 555, 0, 555, 0, red));
    if (lit) objects.add(FlipFace::construct(XzRect::construct(113, 443, 127, 432, 554, light)));
    else objects.add(XzRect::construct(113, 443, 127, 432, 554, light));
    objects.add(XzRect::construct(0, 555, 0, 555, 555, white));
    objects.add(XzRect::construct(0, 555, 0, 555, 0, white));
    objects.add(XyRect::construct(0, 555, 0, 555, 555, white));
    HittablePtr box1 = Box::construct(Point3(0, 0, 0), Point3(165, 330, 165), white);
    box1 = Translate::construct(RotateY::construct(box1, 15.0), Vec3(265, 0, 295));
    HittablePtr box2 = Box::construct(Point3(0, 0, 0), Point3(165, 165, 165), white);
    box2 = Translate::construct(RotateY::construct(box2, -18.0), Vec3(130, 0, 65));
    objects.add(ConstantMedium::construct_color(box1, 0.01, Color3(0, 0, 0)));
    objects.add(ConstantMedium::construct_color(box2, 0.01, Color3(1, 1, 1)));
    SceneRecipe r;
    r.world = objects;
    r.background_mode = RT_BG_CONSTANT; r.background = Color3(0, 0, 0);
    r.lookfrom = Point3(278, 278, -800); r.lookat = Point3(278, 278, 0); r.vfov = 40; r.aperture = 0; r.focus_dist = 10; r.time0 = 0; r.time1 = 1;
    return r;
}

// main.rs:521-649 final_scene (book 2; commented in the reference). `earth` = decoded earthmap.jpg
// (main.rs:601-612) or nullptr (ImageTexture then returns cyan, texture.rs:118-120). `lit`: the light wrapped in FlipFace (as
// main.rs:359-361 does for Cornell) so that it shines DOWN under HEAD's front-face-only DiffuseLight (material.rs:184-190); the literal
// scene (main.rs:548-553) is nearly black. Nothing else differs, the scene RNG draws are the same.
inline SceneRecipe final_scene(uint64_t scene_seed, const uint8_t* earth, uint32_t earth_w, uint32_t earth_h, bool lit = false) {
    SceneRng g(scene_seed);
    HittableList boxes1;
    auto ground = Lambertian::construct(Color3(0.48, 0.83, 0.53));
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            const double w = 100.0;
            const double x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            const double x1 = x0 + w, y1 = g.random_double_range(1.0, 101.0), z1 = z0 + w;
            boxes1.add(Box::construct(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
        }
    HittableList objects;
    objects.add(BVHNode::construct2(boxes1, 0.0, 1.0));
    if (lit) objects.add(FlipFace::construct(XzRect::construct(123, 423, 147, 412, 554, DiffuseLight::construct_color(Color3(7, 7, 7)))));
    else objects.add(XzRect::construct(123, 423, 147, 412, 554, DiffuseLight::construct_color(Color3(7, 7, 7))));
    const Point3 center1(400, 400, 200), center2 = center1 + Vec3(30, 0, 0);
    objects.add(MovingSphere::construct(center1, center2, 0.0, 1.0, 50.0, Lambertian::construct(Color3(0.7, 0.3, 0.1))));
    objects.add(Sphere::construct(Point3(260, 150, 45), 50.0, Dielectric::construct(1.5)));
    objects.add(Sphere::construct(Point3(0, 150, 145), 50.0, Metal::construct(Color3(0.8, 0.8, 0.9), 1.0)));
    auto boundary = Sphere::construct(Point3(360, 150, 145), 70.0, Dielectric::construct(1.5));
    objects.add(boundary);
    objects.add(ConstantMedium::construct_color(boundary, 0.2, Color3(0.2, 0.4, 0.9)));
    auto boundary2 = Sphere::construct(Point3(0, 0, 0), 5000.0, Dielectric::construct(1.5));
    objects.add(ConstantMedium::construct_color(boundary2, 0.0001, Color3(1, 1, 1)));
    objects.add(Sphere::construct(Point3(400, 200, 400), 100.0, Lambertian::construct_texture(ImageTexture::construct(earth, earth_w, earth_h))));
    objects.add(Sphere::construct(Point3(220, 280, 300), 80.0, Lambertian::construct_texture(NoiseTexture::construct(0.1, g))));
    HittableList boxes2;
    auto white = Lambertian::construct(Color3(0.73, 0.73, 0.73));
    for (int j = 0; j < 1000; ++j) boxes2.add(Sphere::construct(Point3::random_range(g, 0.0, 165.0), 10.0, white));
    objects.add(Translate::construct(RotateY::construct(BVHNode::construct2(boxes2, 0.0, 1.0), 15.0), Vec3(-100, 270, 395)));
    SceneRecipe r;
    r.world = objects;
    r.background_mode = RT_BG_CONSTANT; r.background = Color3(0, 0, 0);
    r.lookfrom = Point3(478, 278, -600); r.lookat = Point3(278, 278, 0); r.vfov = 40; r.aperture = 0; r.focus_dist = 10; r.time0 = 0; r.time1 = 1;
    return r;
}

// Wavefront OBJ import (README.md:151-153 lists it as an optional task the reference never did; BASELINE
// config 5 asks for an imported mesh). Supports `v x y z` and `f a b c ...` (1-based or negative indices,
// `i/j/k` forms, polygons fanned into triangles). Vertices are mapped p -> p*scale + offset.
// Returns the number of triangles added, or -1 if the file cannot be read.
inline long load_obj(const std::string& path, std::shared_ptr<Material> m, double scale, const Vec3& offset, HittableList& into) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    std::vector<Point3> v;
    long n_tri = 0;
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
            double x, y, z;
            if (std::sscanf(line + 2, "%lf %lf %lf", &x, &y, &z) == 3) v.push_back(Point3(x * scale, y * scale, z * scale) + offset);
        } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            std::vector<long> idx;
            char* p = line + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == 0 || *p == '\n' || *p == '\r') break;
                char* end = nullptr;
                long i = std::strtol(p, &end, 10);
                if (end == p) break;
                idx.push_back(i > 0 ? i - 1 : (long)v.size() + i);
                p = end;
                while (*p && *p != ' ' && *p != '\t' && *p != '\n') ++p;   // skip /vt/vn
            }
            for (size_t k = 2; k < idx.size(); ++k) {
                const long a = idx[0], b = idx[k - 1], c = idx[k];
                if (a < 0 || b < 0 || c < 0 || a >= (long)v.size() || b >= (long)v.size() || c >= (long)v.size()) continue;
                into.add(Triangle::construct(v[a], v[b], v[c], m));
                ++n_tri;
            }
        }
    }
    std::fclose(f);
    return n_tri;
}

// BASELINE config 5: n random spheres resting on a flat ground + a procedurally generated triangle
// mesh (no mesh file exists in the reference). Not a reference scene.
inline SceneRecipe big_scene(uint64_t scene_seed, uint32_t n_spheres, uint32_t mesh_subdiv, const char* obj_path = nullptr) {
    SceneRng g(scene_seed);
    HittableList world;
    world.add(XzRect::construct(-600, 600, -600, 600, 0, Lambertian::construct(Color3(0.5, 0.5, 0.5))));
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const double choose_mat = g.random_double();
        const double r = g.random_double_range(0.1, 0.5);
        const double x = g.random_double_range(-500, 500), z = g.random_double_range(-500, 500);
        const Point3 c(x, r, z);
        if (choose_mat < 0.8) { const Color3 c1 = Color3::random(g), c2 = Color3::random(g); world.add(Sphere::construct(c, r, Lambertian::construct(c1 * c2))); }
        else if (choose_mat < 0.95) { const Color3 al = Color3::random_range(g, 0.5, 1); const double fz = g.random_double_range(0, 0.5); world.add(Sphere::construct(c, r, Metal::construct(al, fz))); }
        else world.add(Sphere::construct(c, r, Dielectric::construct(1.5)));
    }
    if (obj_path && obj_path[0]) load_obj(obj_path, Metal::construct(Color3(0.8, 0.7, 0.3), 0.05), 1.0, Vec3(0, 0, 0), world);
    // torus mesh: major radius 30, minor 10, centred at (0, 12, 0); 2*nu*nv triangles
    else if (mesh_subdiv > 0) {
        const uint32_t nu = mesh_subdiv, nv = mesh_subdiv / 2 > 3 ? mesh_subdiv / 2 : 3;
        auto mat = Metal::construct(Color3(0.8, 0.7, 0.3), 0.05);
        auto P = [&](uint32_t iu, uint32_t iv) {
            const double a = 2 * PI * (double)(iu % nu) / nu, b = 2 * PI * (double)(iv % nv) / nv;
            const double R0 = 30, r0 = 10;
            return Point3((R0 + r0 * std::cos(b)) * std::cos(a), 12 + r0 * std::sin(b), (R0 + r0 * std::cos(b)) * std::sin(a));
        };
        for (uint32_t iu = 0; iu < nu; ++iu)
            for (uint32_t iv = 0; iv < nv; ++iv) {
                world.add(Triangle::construct(P(iu, iv), P(iu + 1, iv), P(iu + 1, iv + 1), mat));
                world.add(Triangle::construct(P(iu, iv), P(iu + 1, iv + 1), P(iu, iv + 1), mat));
            }
    }
    SceneRecipe r;
    r.world = BVHNode::construct2(world, 0.0, 0.0);
    r.background_mode = RT_BG_SKY_GRADIENT; r.background = Color3(0.5, 0.7, 1.0);
    r.lookfrom = Point3(130, 40, 60); r.lookat = Point3(0, 5, 0); r.vfov = 30; r.aperture = 0.0; r.focus_dist = 10; r.time0 = 0; r.time1 = 0;
    return r;
}

}  // namespace rt

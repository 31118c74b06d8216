// host_capi.cpp — C exports of the host mirror (include/rt_host.h). No GPU code.
#include "../../include/rt_host.h"
#include "rt_host.hpp"
#include "jpeg_writer.hpp"

#include <sys/stat.h>

#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct RtHostScene {
    rt::SceneRecipe recipe;
    rt::FlatScene flat;
};

static bool file_readable(const char* path) { FILE* f = std::fopen(path, "r"); if (!f) return false; std::fclose(f); return true; }

extern "C" {

int rt_host_scene_create(const char* name, uint64_t scene_seed, uint64_t arg0, uint64_t arg1, const uint8_t* image, uint32_t image_w, uint32_t image_h,
                         RtHostScene** out) {
    if (!name || !out) return RT_ERR_INVALID;
    *out = nullptr;
    const std::string n(name);
    auto* s = new RtHostScene();
    bool sah_by_prefix = false;
    if (n == "book1") s->recipe = rt::random_scene(scene_seed, 0, true);
    else if (n == "book1_list") s->recipe = rt::random_scene(scene_seed, 0, false);
    else if (n == "book1_ref") s->recipe = rt::random_scene(scene_seed, 1, true);
    else if (n == "cornell") s->recipe = rt::cornell_box();
    else if (n == "cornell_smoke") s->recipe = rt::cornell_smoke();
    else if (n == "cornell_smoke_lit") s->recipe = rt::cornell_smoke(true);
    else if (n == "final") s->recipe = rt::final_scene(scene_seed, image, image_w, image_h);
    else if (n == "final_lit") s->recipe = rt::final_scene(scene_seed, image, image_w, image_h, true);
    else if (n == "big" || n == "big_sah") s->recipe = rt::big_scene(scene_seed, (uint32_t)arg0, (uint32_t)arg1);
    else if (n == "book1_sah") s->recipe = rt::random_scene(scene_seed, 0, true);
    // BASELINE config 5 with an IMPORTED mesh: "big_obj:<path>" / "big_obj_sah:<path>" = arg0 random spheres + the OBJ file's triangles
    else if ((n.rfind("big_obj:", 0) == 0 || n.rfind("big_obj_sah:", 0) == 0) && !file_readable(n.c_str() + n.find(':') + 1)) { delete s; return RT_ERR_INVALID; }
    else if (n.rfind("big_obj:", 0) == 0) s->recipe = rt::big_scene(scene_seed, (uint32_t)arg0, 0, n.c_str() + 8);
    else if (n.rfind("big_obj_sah:", 0) == 0) { s->recipe = rt::big_scene(scene_seed, (uint32_t)arg0, 0, n.c_str() + 12); sah_by_prefix = true; }
    else if (n.rfind("obj:", 0) == 0) {
        // "obj:<path>": the mesh alone on a ground rect (Lambertian 0.5), camera framing the unit-ish model
        rt::HittableList world;
        const long nt = rt::load_obj(n.substr(4), rt::Lambertian::construct(rt::Color3(0.7, 0.3, 0.3)), arg0 ? (double)arg0 : 1.0, rt::Vec3(0, 0, 0), world);
        if (nt < 0) { delete s; return RT_ERR_INVALID; }
        world.add(rt::XzRect::construct(-50, 50, -50, 50, -1.0, rt::Lambertian::construct(rt::Color3(0.5, 0.5, 0.5))));
        s->recipe.world = rt::BVHNode::construct2(world, 0.0, 0.0);
        s->recipe.background_mode = RT_BG_SKY_GRADIENT; s->recipe.background = rt::Color3(0.5, 0.7, 1.0);
        s->recipe.lookfrom = rt::Point3(3, 2.5, 5); s->recipe.lookat = rt::Point3(0, 0, 0); s->recipe.vfov = 35; s->recipe.aperture = 0; s->recipe.focus_dist = 10;
        s->recipe.time0 = 0; s->recipe.time1 = 0;
    }
    else { delete s; return RT_ERR_INVALID; }
    const bool sah = sah_by_prefix || (n.find(':') == std::string::npos && n.size() > 4 && n.compare(n.size() - 4, 4, "_sah") == 0);
    s->flat.finish(s->recipe.world, s->recipe.lights, s->recipe.background_mode, s->recipe.background, rt::SceneRng::fin(scene_seed ^ 0xB5AD4ECEDA1CE2A9ull),
                   sah ? RT_BVH_SAH : RT_BVH_REFERENCE);
    *out = s;
    return RT_OK;
}
const RtSceneDesc* rt_host_scene_desc(const RtHostScene* s) { return s ? &s->flat.desc : nullptr; }
int rt_host_scene_camera(const RtHostScene* s, double aspect_ratio, RtCamera* out) {
    if (!s || !out) return RT_ERR_INVALID;
    *out = s->recipe.camera(aspect_ratio).abi();
    return RT_OK;
}
void rt_host_scene_destroy(RtHostScene* s) { delete s; }

void rt_host_camera_new(const double* lf, const double* la, const double* up, const double* scope4, double time0, double time1, RtCamera* out) {
    *out = rt::Camera::construct(rt::Point3(lf[0], lf[1], lf[2]), rt::Point3(la[0], la[1], la[2]), rt::Vec3(up[0], up[1], up[2]), scope4, time0, time1).abi();
}
void rt_host_write_color(const double* c, uint32_t spp, uint8_t* out3) { rt::write_color(c, spp, out3); }
int rt_host_tonemap(const float* rgb_sum, uint32_t w, uint32_t h, uint32_t spp, uint8_t* rgb8) {
    if (!rgb_sum || !rgb8 || spp == 0) return RT_ERR_INVALID;
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) { const double c[3] = {rgb_sum[3 * i], rgb_sum[3 * i + 1], rgb_sum[3 * i + 2]}; rt::write_color(c, spp, rgb8 + 3 * i); }
    return RT_OK;
}

static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void chunk(std::vector<uint8_t>& png, const char* type, const std::vector<uint8_t>& data) {
    put32(png, (uint32_t)data.size());
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put32(png, (uint32_t)crc32(0L, png.data() + at, (uInt)(png.size() - at)));
}
int rt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) {
    if (!path || !rgb8 || !w || !h) return -1;
    std::vector<uint8_t> raw; raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb8 + (size_t)y * w * 3, rgb8 + (size_t)(y + 1) * w * 3); }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -1;
    z.resize(zlen);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr; put32(ihdr, w); put32(ihdr, h); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr); chunk(png, "IDAT", z); chunk(png, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    const size_t wr = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    return wr == png.size() ? 0 : -1;
}

int rt_host_write_jpeg(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h, int quality) {
    if (!path || !rgb8 || !w || !h || w > 65535u || h > 65535u) return -1;
    const std::vector<uint8_t> jpg = rtjpeg::encode(rgb8, w, h, quality);
    FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    const size_t wr = std::fwrite(jpg.data(), 1, jpg.size(), f);
    std::fclose(f);
    return wr == jpg.size() ? 0 : -1;
}

int rt_host_write_image(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h, int quality) {
    if (!path) return -1;
    // main.rs:653-656: `create_dir_all(path.parent())` before anything is rendered into "output/book3/image12.jpg"
    std::string p(path);
    for (size_t i = 1; i < p.size(); ++i) if (p[i] == '/') { const std::string dir = p.substr(0, i); (void)mkdir(dir.c_str(), 0777); }
    const size_t dot = p.rfind('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot + 1);
    for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    if (ext == "jpg" || ext == "jpeg") return rt_host_write_jpeg(path, rgb8, w, h, quality);
    if (ext == "png") return rt_host_write_png(path, rgb8, w, h);
    return -1;
}

}  // extern "C"

/*
 * rt_host.h — C exports of the host-side mirror (ray-tracer-archive_amd/host/rt_host.hpp).
 *
 * These are NOT part of the GPU hot path: they are the host code the reference keeps on its side of
 * the boundary — scene functions (main.rs:171-649), Camera::new (camera.rs:21-59), write_color
 * (main.rs:141-169), image encode (main.rs:791-796: JPEG at quality 100 into output/book3/imageNN.jpg; PNG as well) — exported with a C
 * ABI so that non-C++ callers (the Python tests and bench, a Rust binding) can use them.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include "rt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtHostScene RtHostScene;   /* owns the storage behind an RtSceneDesc */

/* name: "book1" (random_scene, main.rs:171-242, book definition, wrapped in BVHNode::construct2),
 *       "book1_list" (the same as a plain HittableList, as main.rs:666 uses it),
 *       "book1_ref" (the reference's remnant: checker ground + MovingSpheres, main.rs:180-218),
 *       "cornell" (main.rs:337-433 + lights main.rs:669-684), "cornell_smoke" (main.rs:435-519),
 *       "final" (main.rs:521-649; `image` = decoded earthmap RGB8 or NULL),
 *       "big" (BASELINE config 5: arg0 spheres + a torus mesh of subdivision arg1);
 *       "obj:<path>" (a Wavefront OBJ mesh, scaled by arg0 if non-zero, on a ground rect);
 *       a "_sah" suffix ("big_sah", "book1_sah") selects RT_BVH_SAH for the scene's BVH objects.
 * scene_seed seeds the scene's random draws; desc.bvh_seed is derived from it. */
int rt_host_scene_create(const char* name, uint64_t scene_seed, uint64_t arg0, uint64_t arg1,
                         const uint8_t* image, uint32_t image_w, uint32_t image_h, RtHostScene** out);
const RtSceneDesc* rt_host_scene_desc(const RtHostScene* s);
/* the camera main.rs sets up for that scene (lookfrom/lookat/vfov/aperture/focus/time) */
int rt_host_scene_camera(const RtHostScene* s, double aspect_ratio, RtCamera* out);
void rt_host_scene_destroy(RtHostScene* s);

/* Camera::new(lookfrom, lookat, vup, [vfov, aspect_ratio, aperture, focus_dist], time0, time1) */
void rt_host_camera_new(const double* lookfrom3, const double* lookat3, const double* vup3, const double* scope4,
                        double time0, double time1, RtCamera* out);
/* write_color(pixel_color, samples_per_pixel) -> [u8;3] */
void rt_host_write_color(const double* pixel_color3, uint32_t samples_per_pixel, uint8_t* out3);
/* write_color over a whole rgb_sum frame (f32, as rt_render returns it) */
int rt_host_tonemap(const float* rgb_sum, uint32_t width, uint32_t height, uint32_t samples_per_pixel, uint8_t* rgb8);
/* RGB8 -> PNG file (zlib). Returns 0, or -1 if the file cannot be written (the reference prints
 * and continues on an encode failure, main.rs:793-796). */
int rt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);
/* RGB8 -> baseline JPEG (JFIF, YCbCr 4:4:4, the standard's Huffman tables; host/jpeg_writer.hpp). quality 1..100: the reference
 * writes its frame with `ImageOutputFormat::Jpeg(quality)`, quality = 100 (main.rs:721,793). Returns 0 / -1 like rt_host_write_png. */
int rt_host_write_jpeg(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, int quality);
/* What main.rs:653-656 + 791-796 do with `path`: create its parent directories ("output/book3/"), then encode by extension
 * (.jpg / .jpeg / .png). */
int rt_host_write_image(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, int quality);

#ifdef __cplusplus
}
#endif
#endif

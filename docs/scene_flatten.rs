//! scene_flatten.rs — the reference's `Arc<dyn Hittable>` graph -> the flat `RtSceneDesc` of include/rt_hip.h.
//!
//! Where it goes: `raytracer/src/scene_flatten.rs` (docs/main_rs.patch adds `mod scene_flatten;`). It adds ONE method to each of the
//! three traits — `Hittable` (hittable.rs:51-60), `Material` (material.rs:11-21), `Texture` (texture.rs:7-9):
//!
//!     fn flatten(&self, b: &mut SceneBuilder) -> i32;      // the id of this object's record
//!
//! and implements it next to each `hit` / `scatter` / `value` (the impl blocks below are written as they would appear in this file with
//! `impl Flatten for T`; a maintainer may equally move each body into the trait impl of its own module).
//!
//! UNVERIFIED TEXT: this image has no Rust toolchain; nothing compiles this file. tests/test_abi.py checks that every
//! RT_HIT_* / RT_MAT_* / RT_TEX_* kind of the header is produced here. Field names are the reference's (sphere.rs:12-16,
//! moving_sphere.rs:9-16, aarect.rs:10-17/60-67/129-136, boxes.rs:11-15, hittable.rs:63-66/99-105/184-186, hittable_list.rs:12-14,
//! material.rs:24-27/75-78/111-113/159-161, texture.rs:12-14/41-44/72-75/99-104, perlin.rs:6-11). Private in the
//! reference, to be made `pub(crate)`: `Box::{box_min, box_max}`, `FlipFace::ptr`, `Metal::{albedo, fuzz}`, `Dielectric::ir`,
//! `DiffuseLight::emit`.
#![allow(dead_code)]
use crate::gpu_ffi::*;
use std::collections::HashMap;
use std::sync::Arc;

/// Owns the flat arrays an `RtSceneDesc` points into. Shared `Arc`s are emitted once: the id of an object is keyed by its address
/// (a boundary sphere that is both in the world and inside a ConstantMedium, the one material of 1000 spheres).
#[derive(Default)]
pub struct SceneBuilder {
    pub hittables: Vec<RtHittable>,
    pub children: Vec<i32>,
    pub materials: Vec<RtMaterial>,
    pub textures: Vec<RtTexture>,
    pub perlins: Vec<RtPerlin>,
    pub images: Vec<RtImage>,
    seen: HashMap<usize, i32>,          // address of the Arc's payload -> record id (per table: ids of different tables never mix,
    seen_mat: HashMap<usize, i32>,      //  so three maps)
    seen_tex: HashMap<usize, i32>,
}

fn v(p: crate::vec3::Vec3) -> RtVec3 { RtVec3 { x: p.x(), y: p.y(), z: p.z() } }
impl From<crate::vec3::Vec3> for RtVec3 { fn from(p: crate::vec3::Vec3) -> Self { v(p) } }

impl SceneBuilder {
    pub fn new() -> Self { Self::default() }

    /// A leaf record: `material` = material id or -1, `child` = child id or -1, `p` = the kind's parameters (rt_hip.h RtHittableKind).
    pub fn hittable(&mut self, kind: i32, material: i32, child: i32, n_children: i32, p: &[f64]) -> i32 {
        let mut q = [0.0f64; 10];
        q[..p.len()].copy_from_slice(p);
        self.hittables.push(RtHittable { kind, material, first_child: child, n_children, p: q });
        (self.hittables.len() - 1) as i32
    }
    /// RT_HIT_LIST / RT_HIT_BVH: the ids of the members go to `children`, contiguously.
    pub fn group(&mut self, kind: i32, ids: &[i32], p: &[f64]) -> i32 {
        let first = self.children.len() as i32;
        self.children.extend_from_slice(ids);
        self.hittable(kind, -1, first, ids.len() as i32, p)
    }
    pub fn hittable_of(&mut self, o: &Arc<dyn Flatten>) -> i32 {
        let key = Arc::as_ptr(o) as *const () as usize;
        if let Some(&id) = self.seen.get(&key) { return id; }
        let id = o.flatten(self);
        self.seen.insert(key, id);
        id
    }
    pub fn material_of(&mut self, m: &Arc<dyn FlattenMaterial>) -> i32 {
        let key = Arc::as_ptr(m) as *const () as usize;
        if let Some(&id) = self.seen_mat.get(&key) { return id; }
        let id = m.flatten(self);
        self.seen_mat.insert(key, id);
        id
    }
    pub fn texture_of(&mut self, t: &Arc<dyn FlattenTexture>) -> i32 {
        let key = Arc::as_ptr(t) as *const () as usize;
        if let Some(&id) = self.seen_tex.get(&key) { return id; }
        let id = t.flatten(self);
        self.seen_tex.insert(key, id);
        id
    }
    fn texture(&mut self, kind: i32, a: i32, b: i32, color: RtVec3, scale: f64) -> i32 {
        self.textures.push(RtTexture { kind, a, b, _pad: 0, color, scale });
        (self.textures.len() - 1) as i32
    }
    fn material(&mut self, kind: i32, texture: i32, albedo: RtVec3, fuzz: f64, ir: f64) -> i32 {
        self.materials.push(RtMaterial { kind, texture, albedo, fuzz, ir });
        (self.materials.len() - 1) as i32
    }

    /// The descriptor. `self` must outlive every use of it (it points into the vectors above). `lights` = id of the lights list
    /// (main.rs:669-684) or -1 for the book-1/2 integrator; `background` = main.rs:692.
    pub fn desc(&self, world: i32, lights: i32, background_mode: i32, background: crate::vec3::Vec3, bvh_seed: u64) -> RtSceneDesc {
        RtSceneDesc {
            abi_version: RT_ABI_VERSION, _pad0: 0,
            hittables: self.hittables.as_ptr(), n_hittables: self.hittables.len() as u64,
            children: self.children.as_ptr(), n_children: self.children.len() as u64,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as u64,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as u64,
            perlins: self.perlins.as_ptr(), n_perlins: self.perlins.len() as u64,
            images: self.images.as_ptr(), n_images: self.images.len() as u64,
            world, lights, background_mode, bvh_builder: RT_BVH_REFERENCE,
            background: v(background), bvh_seed,
        }
    }
}

/// What `Hittable`, `Material`, `Texture` gain (written as separate traits here so that this file stands alone; in the crate the method
/// is added to the existing traits and `Arc<dyn Hittable>` is used where `Arc<dyn Flatten>` appears).
pub trait Flatten { fn flatten(&self, b: &mut SceneBuilder) -> i32; }
pub trait FlattenMaterial { fn flatten(&self, b: &mut SceneBuilder) -> i32; }
pub trait FlattenTexture { fn flatten(&self, b: &mut SceneBuilder) -> i32; }

// ---------------------------------------------------------------------------------------------------------------------------------
// textures (texture.rs)
// ---------------------------------------------------------------------------------------------------------------------------------
impl FlattenTexture for crate::texture::SolidColor {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { b.texture(RT_TEX_SOLID, -1, -1, v(self.color_value), 0.0) }
}
impl FlattenTexture for crate::texture::CheckerTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let (even, odd) = (b.texture_of(&self.even), b.texture_of(&self.odd));     // a = even, b = odd (texture.rs:60-69)
        b.texture(RT_TEX_CHECKER, even, odd, RtVec3::default(), 0.0)
    }
}
impl FlattenTexture for crate::texture::NoiseTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let n = &self.noise;                                                         // perlin.rs:6-11: 256 vectors, three permutations
        let mut p = RtPerlin { ranvec: [[0.0; 3]; 256], perm_x: [0; 256], perm_y: [0; 256], perm_z: [0; 256] };
        for i in 0..256 {
            p.ranvec[i] = [n.ranvec[i].x(), n.ranvec[i].y(), n.ranvec[i].z()];
            p.perm_x[i] = n.perm_x[i]; p.perm_y[i] = n.perm_y[i]; p.perm_z[i] = n.perm_z[i];
        }
        b.perlins.push(p);
        b.texture(RT_TEX_NOISE, (b.perlins.len() - 1) as i32, -1, RtVec3::default(), self.scale)
    }
}
impl FlattenTexture for crate::texture::ImageTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        // RGB8, row-major, bytes_per_scanline = 3 * width (texture.rs:108-115); an empty image renders cyan (texture.rs:118-120): data = null
        let data = if self.data.is_empty() { std::ptr::null() } else { self.data.as_ptr() };
        b.images.push(RtImage { data, width: self.width, height: self.height });
        b.texture(RT_TEX_IMAGE, (b.images.len() - 1) as i32, -1, RtVec3::default(), 0.0)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// materials (material.rs)
// ---------------------------------------------------------------------------------------------------------------------------------
impl FlattenMaterial for crate::material::Lambertian {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let t = b.texture_of(&self.albedo); b.material(RT_MAT_LAMBERTIAN, t, RtVec3::default(), 0.0, 0.0) }
}
impl FlattenMaterial for crate::material::Metal {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { b.material(RT_MAT_METAL, -1, v(self.albedo), self.fuzz, 0.0) }     // fuzz already clamped (material.rs:90)
}
impl FlattenMaterial for crate::material::Dielectric {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { b.material(RT_MAT_DIELECTRIC, -1, RtVec3::default(), 0.0, self.ir) }
}
impl FlattenMaterial for crate::material::DiffuseLight {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let t = b.texture_of(&self.emit); b.material(RT_MAT_DIFFUSE_LIGHT, t, RtVec3::default(), 0.0, 0.0) }
}
/// `Isotropic` is commented out in the reference (material.rs:193-220); once restored it has one field, `albedo: Arc<dyn Texture>`.
pub fn flatten_isotropic(b: &mut SceneBuilder, albedo: &Arc<dyn FlattenTexture>) -> i32 {
    let t = b.texture_of(albedo);
    b.material(RT_MAT_ISOTROPIC, t, RtVec3::default(), 0.0, 0.0)
}

// ---------------------------------------------------------------------------------------------------------------------------------
// hittables
// ---------------------------------------------------------------------------------------------------------------------------------
impl Flatten for crate::sphere::Sphere {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let m = b.material_of(&self.mat_ptr);
        b.hittable(RT_HIT_SPHERE, m, -1, 0, &[self.center.x(), self.center.y(), self.center.z(), self.radius])
    }
}
impl Flatten for crate::moving_sphere::MovingSphere {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let m = b.material_of(&self.mat_ptr);
        b.hittable(RT_HIT_MOVING_SPHERE, m, -1, 0, &[self.center0.x(), self.center0.y(), self.center0.z(),
                                                      self.center1.x(), self.center1.y(), self.center1.z(), self.time0, self.time1, self.radius])
    }
}
impl Flatten for crate::aarect::XyRect {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let m = b.material_of(&self.mp); b.hittable(RT_HIT_XY_RECT, m, -1, 0, &[self.x0, self.x1, self.y0, self.y1, self.k]) }
}
impl Flatten for crate::aarect::XzRect {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let m = b.material_of(&self.mp); b.hittable(RT_HIT_XZ_RECT, m, -1, 0, &[self.x0, self.x1, self.z0, self.z1, self.k]) }
}
impl Flatten for crate::aarect::YzRect {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let m = b.material_of(&self.mp); b.hittable(RT_HIT_YZ_RECT, m, -1, 0, &[self.y0, self.y1, self.z0, self.z1, self.k]) }
}
/// `Box` keeps its six rects in `sides` (boxes.rs:17-74); the library rebuilds them from the two corners in the same order, so only the
/// corners and the material (the one all six sides share: `sides.objects[0]`'s) travel.
pub fn flatten_box(b: &mut SceneBuilder, box_min: crate::vec3::Vec3, box_max: crate::vec3::Vec3, mat: &Arc<dyn FlattenMaterial>) -> i32 {
    let m = b.material_of(mat);
    b.hittable(RT_HIT_BOX, m, -1, 0, &[box_min.x(), box_min.y(), box_min.z(), box_max.x(), box_max.y(), box_max.z()])
}
/// Not in the reference (README.md:151-153 asks for it): a triangle is three points and a material.
pub fn flatten_triangle(b: &mut SceneBuilder, v0: [f64; 3], v1: [f64; 3], v2: [f64; 3], mat: &Arc<dyn FlattenMaterial>) -> i32 {
    let m = b.material_of(mat);
    b.hittable(RT_HIT_TRIANGLE, m, -1, 0, &[v0[0], v0[1], v0[2], v1[0], v1[1], v1[2], v2[0], v2[1], v2[2]])
}
impl Flatten for crate::hittable::Translate {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let c = b.hittable_of(&self.ptr);
        b.hittable(RT_HIT_TRANSLATE, -1, c, 1, &[self.offset.x(), self.offset.y(), self.offset.z()])
    }
}
impl Flatten for crate::hittable::RotateY {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let c = b.hittable_of(&self.ptr);
        // the reference keeps sin and cos of the angle (hittable.rs:101-102); the ABI takes degrees and recomputes them the same way
        b.hittable(RT_HIT_ROTATE_Y, -1, c, 1, &[self.sin_theta.atan2(self.cos_theta).to_degrees()])
    }
}
impl Flatten for crate::hittable::FlipFace {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 { let c = b.hittable_of(&self.ptr); b.hittable(RT_HIT_FLIP_FACE, -1, c, 1, &[]) }
}
impl Flatten for crate::hittable_list::HittableList {
    fn flatten(&self, b: &mut SceneBuilder) -> i32 {
        let ids: Vec<i32> = self.objects.iter().map(|o| b.hittable_of(o)).collect();
        b.group(RT_HIT_LIST, &ids, &[])
    }
}
/// `BVHNode` (bvh.rs:10-14) holds only `left`, `right` and a box, and its `construct` loses objects (it sorts the whole
/// array at every level). What crosses the boundary is therefore the LIST that was handed to `BVHNode::construct2(list, time0, time1)`
/// (bvh.rs:74) — keep it next to the node (`objects: Vec<Arc<dyn Hittable>>, time0, time1`, three new fields set in `construct2`) —
/// and the library builds the tree the reference intended (random axis per node from `bvh_seed`, stable sort of the node's own
/// sub-range, median split) or, with `bvh_builder = RT_BVH_SAH`, a binned-SAH tree over the same objects.
pub fn flatten_bvh(b: &mut SceneBuilder, objects: &[Arc<dyn Flatten>], time0: f64, time1: f64) -> i32 {
    let ids: Vec<i32> = objects.iter().map(|o| b.hittable_of(o)).collect();
    b.group(RT_HIT_BVH, &ids, &[time0, time1])
}
/// `ConstantMedium` is commented out in the reference (constant_medium.rs:9-29); once restored: `boundary: Arc<dyn Hittable>`,
/// `phase_function: Arc<dyn Material>` (an Isotropic), `neg_inv_density: f64` (= -1 / density, constant_medium.rs:25).
pub fn flatten_constant_medium(b: &mut SceneBuilder, boundary: &Arc<dyn Flatten>, phase_function: &Arc<dyn FlattenMaterial>, neg_inv_density: f64) -> i32 {
    let (c, m) = (b.hittable_of(boundary), b.material_of(phase_function));
    b.hittable(RT_HIT_CONSTANT_MEDIUM, m, c, 1, &[-1.0 / neg_inv_density])
}

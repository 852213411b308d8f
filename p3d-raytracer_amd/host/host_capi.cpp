// host_capi.cpp — the p3d_host_* half of include/p3d.h: .p3f -> Scene -> accel builds ->
// flat p3d_scene_desc.  Pure host code (no HIP).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "accel_build.hpp"
#include "p3d.h"
#include "p3d_error.hpp"
#include "scene_model.hpp"

struct p3d_host_scene {
  p3d::Scene scene;
  std::unique_ptr<p3d::BVH> bvh;
  std::unique_ptr<p3d::Grid> grid;
  std::vector<p3d_prim> prims;
  std::vector<p3d_material> materials;
  std::vector<p3d_light> lights;
  p3d_scene_desc desc{};
  bool flat_valid = false;
};

namespace {

void flatten(p3d_host_scene& hs) {
  using namespace p3d;
  const Scene& S = hs.scene;
  hs.prims.assign(S.getNumObjects(), p3d_prim{});
  for (int i = 0; i < S.getNumObjects(); ++i) {
    const Object* o = S.getObject(i);
    p3d_prim& p = hs.prims[i];
    o->pack(p.v, p.n);
    p.type = static_cast<uint32_t>(o->kind());
    p.material = static_cast<uint32_t>(S.materialIndex(o->GetMaterial()));
    const AABB b = o->GetBoundingBox();
    p.bmin[0] = b.min.x; p.bmin[1] = b.min.y; p.bmin[2] = b.min.z;
    p.bmax[0] = b.max.x; p.bmax[1] = b.max.y; p.bmax[2] = b.max.z;
  }
  hs.materials.clear();
  for (const auto& m : S.allMaterials()) {
    p3d_material f{};
    const Color cd = m->GetDiffColor(), cs = m->GetSpecColor(), em = m->GetEmission();
    f.diff_color[0] = cd.r(); f.diff_color[1] = cd.g(); f.diff_color[2] = cd.b();
    f.spec_color[0] = cs.r(); f.spec_color[1] = cs.g(); f.spec_color[2] = cs.b();
    f.emission[0] = em.r(); f.emission[1] = em.g(); f.emission[2] = em.b();
    f.diffuse = m->GetDiffuse(); f.specular = m->GetSpecular(); f.shine = m->GetShine();
    f.transmittance = m->GetTransmittance(); f.refr_index = m->GetRefrIndex();
    f.reflection = m->GetReflection();
    hs.materials.push_back(f);
  }
  hs.lights.clear();
  for (int i = 0; i < S.getNumLights(); ++i) {
    const Light* l = S.getLight(i);
    p3d_light f{};
    f.position[0] = l->position.x; f.position[1] = l->position.y; f.position[2] = l->position.z;
    f.color[0] = l->color.r(); f.color[1] = l->color.g(); f.color[2] = l->color.b();
    hs.lights.push_back(f);
  }
  p3d_scene_desc& d = hs.desc;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = P3D_ABI_VERSION;
  d.n_prims = static_cast<uint32_t>(hs.prims.size());
  d.n_materials = static_cast<uint32_t>(hs.materials.size());
  d.n_lights = static_cast<uint32_t>(hs.lights.size());
  d.prims = hs.prims.data();
  d.materials = hs.materials.data();
  d.lights = hs.lights.data();
  if (const Camera* c = S.GetCamera()) {
    p3d_camera& k = d.camera;
    k.eye[0] = c->eye.x; k.eye[1] = c->eye.y; k.eye[2] = c->eye.z;
    k.u[0] = c->u.x; k.u[1] = c->u.y; k.u[2] = c->u.z;
    k.v[0] = c->v.x; k.v[1] = c->v.y; k.v[2] = c->v.z;
    k.n[0] = c->n.x; k.n[1] = c->n.y; k.n[2] = c->n.z;
    k.plane_dist = c->plane_dist; k.w = c->w; k.h = c->h; k.focal_ratio = c->focal_ratio;
    k.aperture = c->aperture; k.res_x = c->res_x; k.res_y = c->res_y;
  }
  const Color bg = S.GetBackgroundColor();
  d.background[0] = bg.r(); d.background[1] = bg.g(); d.background[2] = bg.b();
  if (hs.bvh) {
    d.n_bvh_nodes = static_cast<uint32_t>(hs.bvh->flatNodes().size());
    d.bvh_nodes = hs.bvh->flatNodes().data();
    d.bvh_prim_index = hs.bvh->primOrder().data();
    d.n_bvh_prim_index = static_cast<uint32_t>(hs.bvh->primOrder().size());
    d.bvh_max_depth = hs.bvh->maxDepth();
  }
  if (hs.grid) {
    d.has_grid = 1;
    d.grid = hs.grid->describe();
  }
  hs.flat_valid = true;
}

std::vector<p3d::Object*> object_list(const p3d::Scene& S) {
  std::vector<p3d::Object*> v;
  for (int i = 0; i < S.getNumObjects(); ++i) v.push_back(S.getObject(i));
  return v;
}

}  // namespace

extern "C" {

int p3d_host_scene_load(const char* p3f_path, uint32_t flags, p3d_host_scene** out) {
  if (!p3f_path || !out) return p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_load: null argument");
  auto hs = std::make_unique<p3d_host_scene>();
  if (!hs->scene.load_p3f(p3f_path, (flags & P3D_LOAD_LEGACY_F11) != 0))
    return p3d::fail(P3D_ERR_IO, std::string("cannot open ") + p3f_path);
  *out = hs.release();
  return P3D_OK;
}

void p3d_host_scene_destroy(p3d_host_scene* hs) { delete hs; }

int p3d_host_scene_set_resolution(p3d_host_scene* hs, int32_t rx, int32_t ry) {
  if (!hs || !hs->scene.view.present) return p3d::fail(P3D_ERR_INVALID, "scene has no camera (`v` block)");
  if (rx > 0) hs->scene.view.xres = rx;
  if (ry > 0) hs->scene.view.yres = ry;
  hs->scene.rebuildCamera();
  hs->flat_valid = false;
  return P3D_OK;
}

int p3d_host_scene_set_lens(p3d_host_scene* hs, float aperture_ratio, float focal_ratio) {
  if (!hs || !hs->scene.view.present) return p3d::fail(P3D_ERR_INVALID, "scene has no camera (`v` block)");
  hs->scene.view.aperture = aperture_ratio;
  hs->scene.view.focal = focal_ratio;
  hs->scene.rebuildCamera();
  hs->flat_valid = false;
  return P3D_OK;
}

// main.cpp:725-745: SPP x SPP copies of every light on a LIGHT_SIDE square, colour / SPP^2
int p3d_host_scene_replicate_lights(p3d_host_scene* hs, uint32_t spp_sqrt, float light_side) {
  if (!hs || spp_sqrt == 0) return p3d::fail(P3D_ERR_INVALID, "replicate_lights: bad argument");
  using namespace p3d;
  const int SPP = static_cast<int>(spp_sqrt);
  const float step = light_side / SPP;
  const float start = -light_side / 2 + step / 2;
  const float end = light_side / 2;
  std::vector<std::unique_ptr<Light>> grown;
  const int limit = hs->scene.getNumLights();
  for (int k = 0; k < limit; ++k) {
    const Light* light = hs->scene.getLight(k);
    const Color avg = light->color / static_cast<float>(SPP * SPP);
    for (float i = start; i < end; i += step)
      for (float j = start; j < end; j += step)
        grown.emplace_back(new Light(Vector(light->position.x + i, light->position.y + j, light->position.z), avg));
  }
  hs->scene.setLights(std::move(grown));
  hs->flat_valid = false;
  return P3D_OK;
}

int p3d_host_scene_desc(p3d_host_scene* hs, int build_bvh, int build_grid, const p3d_scene_desc** out) {
  if (!hs || !out) return p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_desc: null argument");
  for (int i = 0; i < hs->scene.getNumObjects(); ++i)
    if (!hs->scene.getObject(i)->GetMaterial())
      return p3d::fail(P3D_ERR_INVALID, "object declared before the first `f` line has no material");
  if (build_bvh && !hs->bvh) {  // built once: BVH::objs is append-only in the reference (bvh.cpp:101)
    hs->bvh = std::make_unique<p3d::BVH>();
    hs->bvh->build(object_list(hs->scene));
    hs->flat_valid = false;
  }
  if (build_grid && !hs->grid) {
    hs->grid = std::make_unique<p3d::Grid>();
    for (p3d::Object* o : object_list(hs->scene)) hs->grid->addObject(o);
    hs->grid->Build();
    hs->flat_valid = false;
  }
  if (!hs->flat_valid) flatten(*hs);
  *out = &hs->desc;
  return P3D_OK;
}

int p3d_host_scene_has_skybox(p3d_host_scene* hs) { return hs && hs->scene.SkyboxLoaded() ? 1 : 0; }

int p3d_host_scene_load_skybox(p3d_host_scene* hs, const char* sky_dir) {
  if (!hs || !sky_dir) return p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_load_skybox: null argument");
  return hs->scene.LoadSkybox(sky_dir) ? P3D_OK : P3D_ERR_IO;  // (the message names the face that could not be read)
}

int p3d_host_scene_skybox_face(p3d_host_scene* hs, int face, const uint8_t** img, uint32_t* res_x, uint32_t* res_y) {
  if (!hs || !img) return p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_skybox_face: null argument");
  *img = hs->scene.SkyboxFace(face, res_x, res_y);
  return *img ? P3D_OK : p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_skybox_face: no cubemap loaded, or face not in 0..5");
}

int p3d_host_scene_bind_device(p3d_host_scene* hs, p3d_scene* scene) {
  if (!hs) return p3d::fail(P3D_ERR_INVALID, "p3d_host_scene_bind_device: null argument");
  if (hs->bvh) hs->bvh->bindDevice(scene);
  if (hs->grid) hs->grid->bindDevice(scene);
  return hs->scene.bindDevice(scene) ? P3D_OK : P3D_ERR_NO_DEVICE;
}

}  // extern "C"

namespace p3d {
HostClasses host_classes(::p3d_host_scene* hs) { return HostClasses{&hs->scene, hs->bvh.get(), hs->grid.get()}; }
}  // namespace p3d

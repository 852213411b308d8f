// host_api_check.cpp — exercises the reference-named query methods of the host classes (scene_model.hpp,
// accel_build.hpp) against a bound device scene and prints what they return, one record per line, floats as
// hex literals.  tests/test_gpu_parity.py compiles it, runs it on the GPU box and compares every line with
// the oracle.  usage: host_api_check scene.p3f [skybox_ppm_dir]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <memory>
#include <vector>

#include "accel_build.hpp"
#include "p3d.h"
#include "scene_model.hpp"

using namespace p3d;

static void die(const char* what) {
  std::fprintf(stderr, "%s: %s\n", what, p3d_last_error());
  std::exit(1);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  p3d_host_scene* hs = nullptr;
  if (p3d_host_scene_load(argv[1], 0, &hs) != P3D_OK) die("load");
  const p3d_scene_desc* desc = nullptr;
  if (p3d_host_scene_desc(hs, 1, 1, &desc) != P3D_OK) die("desc");
  HostClasses hc = host_classes(hs);
  // unbound: every query fails loudly instead of computing on the host
  {
    Ray r(Vector(0, 0, 5), Vector(0, 0, -1));
    float t = 0;
    Object* o = nullptr;
    Vector hp;
    const bool a = hc.scene->getObject(0)->intercepts(r, t), b = hc.bvh->intersect_bvh(r, &o, hp), c = hc.grid->Traverse(r);
    std::printf("UNBOUND %d %d %d\n", (int)a, (int)b, (int)c);
  }
  p3d_scene* dev = nullptr;
  if (p3d_scene_create(desc, 0, &dev) != P3D_OK) die("create");
  if (p3d_host_scene_bind_device(hs, dev) != P3D_OK) die("bind");
  Camera* cam = hc.scene->GetCamera();
  // a fixed pseudo-random sequence (LCG) of pixels, lens samples, rays and points
  uint32_t seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f; };
  for (int k = 0; k < 8; ++k) {
    const Vector px(rnd() * cam->GetResX(), rnd() * cam->GetResY(), 0), ls(rnd() * 2 - 1, rnd() * 2 - 1, 0);
    Ray a = cam->PrimaryRay(px), b = cam->PrimaryRay(ls, px);
    std::printf("PRIMARY %a %a %a %a %a %a %a %a %a %a %a %a %a %a %a %a\n", px.x, px.y, ls.x, ls.y, a.origin.x, a.origin.y, a.origin.z,
                a.direction.x, a.direction.y, a.direction.z, b.origin.x, b.origin.y, b.origin.z, b.direction.x, b.direction.y, b.direction.z);
  }
  const int n_obj = hc.scene->getNumObjects();
  std::vector<Ray> batch;
  for (int k = 0; k < 64; ++k) {
    Ray r = cam->PrimaryRay(Vector(rnd() * cam->GetResX(), rnd() * cam->GetResY(), 0));
    if (k % 3 == 0) r.direction = r.direction * (0.5f + 3 * rnd());  // unnormalised directions (Q8)
    const Ray r0 = r;
    batch.push_back(r0);
    const int obj = k % n_obj;
    float t = -1.0f;
    const bool hit = hc.scene->getObject(obj)->intercepts(r, t);
    std::printf("OBJ %d %a %a %a %a %a %a %d %a %a %a %a\n", obj, r0.origin.x, r0.origin.y, r0.origin.z, r0.direction.x, r0.direction.y,
                r0.direction.z, (int)hit, t, r.direction.x, r.direction.y, r.direction.z);
    const Vector p(rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2), nn = hc.scene->getObject(obj)->getNormal(p);
    std::printf("NORMAL %d %a %a %a %a %a %a\n", obj, p.x, p.y, p.z, nn.x, nn.y, nn.z);
    Object* ho = nullptr;
    Vector hp(0, 0, 0);
    Ray q = r0;
    const bool bh = hc.bvh->intersect_bvh(q, &ho, hp);
    int id = -1;
    for (int i = 0; bh && i < n_obj; ++i) if (hc.scene->getObject(i) == ho) id = i;
    std::printf("BVH %a %a %a %a %a %a %d %a %a %a %d\n", r0.origin.x, r0.origin.y, r0.origin.z, r0.direction.x, r0.direction.y, r0.direction.z,
                id, hp.x, hp.y, hp.z, (int)hc.bvh->bool_intersect_bvh(q));
    ho = nullptr; hp = Vector(0, 0, 0);
    Ray g = r0;
    const bool gh = hc.grid->Traverse(g, &ho, hp);
    id = -1;
    for (int i = 0; gh && i < n_obj; ++i) if (hc.scene->getObject(i) == ho) id = i;
    Ray g2 = r0;
    std::printf("GRID %a %a %a %a %a %a %d %a %a %a %d\n", r0.origin.x, r0.origin.y, r0.origin.z, r0.direction.x, r0.direction.y, r0.direction.z,
                id, hp.x, hp.y, hp.z, (int)hc.grid->Traverse(g2));
  }
  {  // the batched forms of the two BVH queries: the same rays in ONE launch each, printed like the single-ray lines
    const size_t n = batch.size();
    std::vector<Object*> objs(n, nullptr);
    std::vector<Vector> hps(n, Vector(0, 0, 0));
    std::unique_ptr<bool[]> hit(new bool[n]), occ(new bool[n]);
    if (!hc.bvh->intersect_bvh(batch.data(), n, hit.get(), objs.data(), hps.data())) die("batched intersect_bvh");
    if (!hc.bvh->bool_intersect_bvh(batch.data(), n, occ.get())) die("batched bool_intersect_bvh");
    for (size_t k = 0; k < n; ++k) {
      int id = -1;
      for (int i = 0; hit[k] && i < n_obj; ++i) if (hc.scene->getObject(i) == objs[k]) id = i;
      const Ray& r0 = batch[k];
      std::printf("BVH %a %a %a %a %a %a %d %a %a %a %d\n", r0.origin.x, r0.origin.y, r0.origin.z, r0.direction.x, r0.direction.y, r0.direction.z,
                  id, hps[k].x, hps[k].y, hps[k].z, (int)occ[k]);
    }
  }
  if (argc > 2) {
    if (!hc.scene->LoadSkybox(argv[2])) die("LoadSkybox");
    for (int k = 0; k < 64; ++k) {
      Ray r(Vector(0, 0, 0), Vector(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1));
      const Color c = hc.scene->GetSkyboxColor(r);
      std::printf("SKY %a %a %a %a %a %a\n", r.direction.x, r.direction.y, r.direction.z, c.r(), c.g(), c.b());
    }
  }
  p3d_scene_destroy(dev);
  p3d_host_scene_destroy(hs);
  return 0;
}

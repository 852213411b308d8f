// scene_model.cpp — shape constructors, camera frame and the .p3f loader of the host side.
#include "scene_model.hpp"
#include "jpeg_decode.hpp"

#include "p3d.h"
#include "p3d_error.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace p3d {

namespace {
constexpr float kEpsilon = 0.0001f;                 // scene.h:31 EPSILON
constexpr float kPi = 3.141592653589793238462f;     // camera.h:13 (float literal)

inline float lo3(float a, float b, float c) { return std::min(std::min(a, b), c); }
inline float hi3(float a, float b, float c) { return std::max(std::max(a, b), c); }
}  // namespace

// ---- shapes -----------------------------------------------------------------
void Sphere::pack(float v[9], float n[3]) const {
  const float g[9] = {center.x, center.y, center.z, radius, 0, 0, 0, 0, 0};
  std::memcpy(v, g, sizeof(g));
  n[0] = n[1] = n[2] = 0.f;
}

// scene.cpp:12-35: unit normal of (P1-P0)x(P2-P0); bbox grown by EPSILON on every side
Triangle::Triangle(const Vector& P0, const Vector& P1, const Vector& P2) {
  points[0] = P0; points[1] = P1; points[2] = P2;
  normal = (P1 - P0) % (P2 - P0);
  normal.normalize();
  Min = Vector(lo3(P0.x, P1.x, P2.x) - kEpsilon, lo3(P0.y, P1.y, P2.y) - kEpsilon,
               lo3(P0.z, P1.z, P2.z) - kEpsilon);
  Max = Vector(hi3(P0.x, P1.x, P2.x) + kEpsilon, hi3(P0.y, P1.y, P2.y) + kEpsilon,
               hi3(P0.z, P1.z, P2.z) + kEpsilon);
}
void Triangle::pack(float v[9], float n[3]) const {
  for (int k = 0; k < 3; ++k) {
    v[3 * k] = points[k].x; v[3 * k + 1] = points[k].y; v[3 * k + 2] = points[k].z;
  }
  n[0] = normal.x; n[1] = normal.y; n[2] = normal.z;
}

void aaBox::pack(float v[9], float n[3]) const {
  const float g[9] = {min.x, min.y, min.z, max.x, max.y, max.z, 0, 0, 0};
  std::memcpy(v, g, sizeof(g));
  n[0] = n[1] = n[2] = 0.f;
}

// scene.cpp:102-113: PN = unit (P2-P1)x(P0-P1), anchored at P0
Plane::Plane(const Vector& P0, const Vector& P1, const Vector& P2) {
  PN = (P2 - P1) % (P0 - P1);
  PN.normalize();
  A = P0;
}
void Plane::pack(float v[9], float n[3]) const {
  const float g[9] = {PN.x, PN.y, PN.z, A.x, A.y, A.z, 0, 0, 0};
  std::memcpy(v, g, sizeof(g));
  n[0] = PN.x; n[1] = PN.y; n[2] = PN.z;
}

// ---- camera.h:34-63 -----------------------------------------------------------
Camera::Camera(const Vector& from, const Vector& At, const Vector& Up, float angle, float hither,
               float yon, int ResX, int ResY, float Aperture_ratio, float Focal_ratio)
    : eye(from), at(At), up(Up), fovy(angle), vnear(hither), vfar(yon), focal_ratio(Focal_ratio),
      aperture_ratio(Aperture_ratio), res_x(ResX), res_y(ResY) {
  n = eye - at;
  plane_dist = n.length();
  n = n / plane_dist;
  u = up % n;
  u = u / u.length();
  v = n % u;
  h = 2 * plane_dist * std::tan((kPi * angle / 180) / 2.0f);
  w = (static_cast<float>(res_x) / res_y) * h;
  aperture = Aperture_ratio * (w / res_x);  // lens aperture = ratio * pixel size
}

// camera.h:65-82
Ray Camera::PrimaryRay(const Vector& pixel_sample) {
  Vector ps;
  ps.x = w * (pixel_sample.x / res_x - 0.5f);
  ps.y = h * (pixel_sample.y / res_y - 0.5f);
  ps.z = -plane_dist;
  const Vector vX = u * ps.x, vY = v * ps.y, vZ = n * ps.z;
  Vector ray_dir = vX + vY + vZ;
  ray_dir.normalize();
  return Ray(eye, ray_dir);
}
// camera.h:84-115
Ray Camera::PrimaryRay(const Vector& lens_sample, const Vector& pixel_sample) {
  Vector ps;
  ps.x = w * (pixel_sample.x / res_x - 0.5f);
  ps.y = h * (pixel_sample.y / res_y - 0.5f);
  ps.z = -plane_dist;
  Vector ls;
  ls.x = lens_sample.x * aperture;
  ls.y = lens_sample.y * aperture;
  ls.z = 0;
  Vector p;
  p.x = ps.x * focal_ratio;
  p.y = ps.y * focal_ratio;
  const Vector vX = u * (p.x - ls.x), vY = v * (p.y - ls.y), vZ = n * -(focal_ratio * plane_dist);
  Vector ray_dir = vX + vY + vZ;
  ray_dir.normalize();
  const Vector eye_offset = eye + (u * ls.x) + (v * ls.y);
  return Ray(eye_offset, ray_dir);
}

// ---- Object: the two virtuals of scene.h:88-89, answered by the device ----------
bool Object::intercepts(Ray& r, float& t) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "Object::intercepts: the scene is not bound to a device scene (Scene::bindDevice)"); return false; }
  const float o[3] = {r.origin.x, r.origin.y, r.origin.z};
  float d[3] = {r.direction.x, r.direction.y, r.direction.z};
  uint8_t hit = 0;
  float tt = t;
  if (p3d_object_intercepts(dev_, index_, 1, o, d, &hit, &tt) != P3D_OK) return false;
  r.direction = Vector(d[0], d[1], d[2]);  // Sphere::intercepts normalises the caller's ray (scene.cpp:156)
  if (hit) t = tt;
  return hit != 0;
}
Vector Object::getNormal(Vector point) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "Object::getNormal: the scene is not bound to a device scene (Scene::bindDevice)"); return Vector(); }
  const float p[3] = {point.x, point.y, point.z};
  float nrm[3] = {0, 0, 0};
  if (p3d_object_normal(dev_, index_, 1, p, nrm) != P3D_OK) return Vector();
  return Vector(nrm[0], nrm[1], nrm[2]);
}

// ---- Scene --------------------------------------------------------------------
bool Scene::bindDevice(::p3d_scene* dev) {
  dev_ = dev;
  for (size_t i = 0; i < objects.size(); ++i) objects[i]->bind(dev, static_cast<uint32_t>(i));
  if (dev && skybox_loaded) {
    p3d_skybox_desc sky{};
    for (int f = 0; f < 6; ++f) {
      sky.face[f].img = skybox_img[f].img.data();
      sky.face[f].res_x = skybox_img[f].resX; sky.face[f].res_y = skybox_img[f].resY; sky.face[f].bpp = skybox_img[f].BPP;
    }
    return p3d_scene_set_skybox(dev, &sky) == P3D_OK;
  }
  return true;
}

// scene.cpp:329-377: <dir>/{right,left,top,bottom,front,back}.jpg, decoded here (host/jpeg_decode.cpp: baseline JPEG, the
// six shipped faces byte for byte what PIL's libjpeg-turbo gives) where the reference asks DevIL; a face that is not there as
// .jpg is taken from a binary PPM of the same name (what round 3 needed scenes/skybox_to_ppm.py for).
bool Scene::LoadSkybox(const char* sky_dir) {
  static const char* maps[6] = {"/right", "/left", "/top", "/bottom", "/front", "/back"};  // scene.cpp:333
  for (int f = 0; f < 6; ++f) {
    std::vector<uint8_t> top_down;
    uint32_t w = 0, h = 0;
    const std::string jpg = std::string(sky_dir) + maps[f] + ".jpg";
    std::ifstream jfile(jpg, std::ios::binary);
    if (jfile) {
      const std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(jfile)), std::istreambuf_iterator<char>());
      std::string why;
      if (!jpeg_decode_rgb(bytes.data(), bytes.size(), top_down, w, h, why)) {
        fail(P3D_ERR_IO, "Scene::LoadSkybox: " + jpg + ": " + why);
        return false;
      }
    } else {
      const std::string path = std::string(sky_dir) + maps[f] + ".ppm";
      std::ifstream file(path, std::ios::binary);
      std::string magic;
      int maxv = 0;
      if (file) file >> magic >> w >> h >> maxv;
      if (!file || magic != "P6" || maxv != 255 || w == 0 || h == 0) {
        fail(P3D_ERR_IO, "Scene::LoadSkybox: cannot read " + jpg + " nor " + path + " (baseline JPEG or binary PPM)");
        return false;
      }
      file.get();
      top_down.resize((size_t)w * h * 3);
      file.read(reinterpret_cast<char*>(top_down.data()), (std::streamsize)top_down.size());
      if (!file) { fail(P3D_ERR_IO, "Scene::LoadSkybox: short file " + path); return false; }
    }
    Face& F = skybox_img[f];
    F.resX = w; F.resY = h; F.BPP = 3;
    F.img.resize(top_down.size());
    for (uint32_t y = 0; y < h; ++y)  // IL_ORIGIN_LOWER_LEFT (scene.cpp:344-345): bottom row first
      std::memcpy(&F.img[(size_t)y * w * 3], &top_down[(size_t)(h - 1 - y) * w * 3], (size_t)w * 3);
    std::printf("Skybox face %d: Image sucessfully loaded.\n", f);  // scene.cpp:352
  }
  skybox_loaded = true;
  skyboxDir = sky_dir;
  return dev_ ? bindDevice(dev_) : true;
}

Color Scene::GetSkyboxColor(Ray& r) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "Scene::GetSkyboxColor: the scene is not bound to a device scene (Scene::bindDevice)"); return Color(); }
  const float d[3] = {r.direction.x, r.direction.y, r.direction.z};  // the RAW direction (scene.cpp:381)
  float rgb[3] = {0, 0, 0};
  if (p3d_skybox_color(dev_, 1, d, rgb) != P3D_OK) return Color();
  return Color(rgb[0], rgb[1], rgb[2]);
}

int Scene::materialIndex(const Material* m) const {
  for (size_t i = 0; i < materials.size(); ++i)
    if (materials[i].get() == m) return static_cast<int>(i);
  return -1;
}

void Scene::rebuildCamera() {
  if (!view.present) return;
  camera.reset(new Camera(view.from, view.at, view.up, view.angle, view.hither,
                          static_cast<float>(100.0 * view.hither), view.xres, view.yres, view.aperture,
                          view.focal));
}

namespace {
std::istream& operator>>(std::istream& s, Vector& v) { return s >> v.x >> v.y >> v.z; }
std::istream& operator>>(std::istream& s, Color& c) {
  float r = 0, g = 0, b = 0;
  s >> r >> g >> b;
  c = Color(r, g, b);
  return s;
}
}  // namespace

// Grammar: SURVEY.md Appendix B / scene.cpp:472-628.  iostream extraction is used on
// purpose: numeric parsing and the "a failed read ends the parse" behaviour of the
// shipped loader come from it.
bool Scene::load_p3f(const char* name, bool legacy_f11) {
  std::ifstream in(name, std::ios::in);
  if (!in.is_open()) return false;

  Material* current = nullptr;
  bool stop = false, env_failed = false;

  auto keyword = [&](const char* expect) {  // scene.cpp:465-470: mismatch only warns
    std::string tok;
    in >> tok;
    if (tok != expect) std::fprintf(stderr, "'%s' expected.\n", expect);
  };
  auto attach = [&](Object* o) {
    if (current) o->SetMaterial(current);
    addObject(o);
  };

  const std::map<std::string, std::function<void()>> handlers = {
      {"f", [&] {
         // Kd, Ks, Shine, T, ior are extracted as double and narrowed (scene.cpp:489-494)
         Color cd, cs, em;
         double Kd = 0, Ks = 0, Shine = 0, T = 0, ior = 0;
         in >> cd >> Kd >> cs >> Ks >> Shine >> T >> ior;
         bool has_emission = true;
         if (legacy_f11) {  // extension: 11-number form has no emission triple
           const std::streampos mark = in.tellg();
           std::string peek;
           has_emission = false;
           if (in >> peek) {
             char* end = nullptr;
             std::strtod(peek.c_str(), &end);
             has_emission = end != peek.c_str() && *end == '\0';
           }
           in.clear();
           in.seekg(mark);
         }
         if (has_emission) in >> em;
         materials.emplace_back(new Material(cd, static_cast<float>(Kd), cs, static_cast<float>(Ks),
                                             static_cast<float>(Shine), static_cast<float>(T),
                                             static_cast<float>(ior), em));
         current = materials.back().get();
       }},
      {"s", [&] {
         Vector c;
         float r = 0;
         in >> c >> r;
         attach(new Sphere(c, r));
       }},
      {"box", [&] {
         Vector lo, hi;
         in >> lo >> hi;
         attach(new aaBox(lo, hi));
       }},
      {"p", [&] {
         unsigned nverts = 0;
         in >> nverts;
         if (nverts != 3) {
           std::fprintf(stderr, "Unsupported number of vertices.\n");
           stop = true;
           return;
         }
         Vector a, b, c;
         in >> a >> b >> c;
         attach(new Triangle(a, b, c));
       }},
      {"pl", [&] {
         Vector a, b, c;
         in >> a >> b >> c;
         attach(new Plane(a, b, c));
       }},
      {"l", [&] {
         Vector pos;
         Color col;
         in >> pos >> col;
         addLight(new Light(pos, col));
       }},
      {"v", [&] {
         keyword("from"); in >> view.from;
         keyword("at"); in >> view.at;
         keyword("up"); in >> view.up;
         keyword("angle"); in >> view.angle;
         keyword("hither"); in >> view.hither;
         keyword("resolution"); in >> view.xres >> view.yres;
         keyword("aperture"); in >> view.aperture;
         keyword("focal"); in >> view.focal;
         view.present = true;
         rebuildCamera();  // yon = 100 * hither (scene.cpp:594)
       }},
      {"bclr", [&] {
         Color c;
         in >> c;
         SetBackgroundColor(c);
       }},
      {"env", [&] {  // scene.cpp:605-610: LoadSkybox(<dir>) + SetSkyBoxFlg(true)
         // The reference decodes <dir>/*.jpg through DevIL and exits if a face is missing.  Here the folder is looked up relative
         // to the working directory as the reference does and, failing that, next to the scene file; its faces are decoded by
         // LoadSkybox (baseline JPEG, or binary PPM).  With no such folder the directory is only recorded (GetSkyboxDir) and a
         // later LoadSkybox / p3d_scene_set_skybox supplies the faces.
         std::string dir;
         in >> dir;
         skyboxDir = dir;
         SetSkyBoxFlg(true);
         std::string beside = name;
         const size_t slash = beside.find_last_of('/');
         beside = slash == std::string::npos ? dir : beside.substr(0, slash + 1) + dir;
         for (const std::string& cand : {dir, beside}) {
           if (!std::ifstream(cand + "/right.jpg", std::ios::binary).is_open() && !std::ifstream(cand + "/right.ppm", std::ios::binary).is_open()) continue;
           if (!LoadSkybox(cand.c_str())) stop = env_failed = true;  // faces present but unreadable: the reference would exit(0) here
           break;
         }
       }},
  };

  std::string cmd;
  while (!stop && (in >> cmd)) {
    auto it = handlers.find(cmd);
    if (it != handlers.end()) {
      it->second();
    } else if (cmd[0] == '#') {
      in.ignore(1024, '\n');
    } else {
      std::fprintf(stderr, "unknown command '%s'.\n", cmd.c_str());
      break;
    }
  }
  return !env_failed;
}

}  // namespace p3d

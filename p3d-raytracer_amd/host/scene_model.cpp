// scene_model.cpp — shape constructors, camera frame and the .p3f loader of the host side.
#include "scene_model.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>

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

// ---- Scene --------------------------------------------------------------------
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
  bool stop = false;

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
      {"env", [&] {  // the six cubemap JPEGs are not loaded: miss shading uses bclr (SKYBOX false)
         in >> skyboxDir;
         SetSkyBoxFlg(true);
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
  return true;
}

}  // namespace p3d

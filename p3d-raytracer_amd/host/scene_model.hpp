// scene_model.hpp — host-side scene model of the MI355X renderer.
//
// Keeps the reference's API surface (class and method names of Raytracing/scene.h:34-225,
// camera.h:15-116, vector.h, color.h, boundingBox.h) so that code written against the
// reference's Scene / Camera / Object / Material / Light getters keeps compiling.  The classes
// DESCRIBE the scene; every ray query — Object::intercepts / getNormal, BVH::intersect_bvh /
// bool_intersect_bvh, Grid::Traverse x2, Scene::GetSkyboxColor, and of course rayTracing /
// Radiance — is answered by the HIP kernels behind include/p3d.h: the methods of those names
// below forward ONE query to the device scene the host scene was bound to (Scene::bindDevice)
// and fail (false / zero vector, p3d_last_error set) when there is none.  There is no host
// implementation of any intersection test in this library.  Only Camera::PrimaryRay (camera.h:65-115,
// a dozen float operations, as header-only in the reference) computes on the host.  The host
// builds the acceleration structures (as the reference does, main.cpp:701-720) and flattens
// everything into a p3d_scene_desc.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct p3d_scene;  // include/p3d.h: the device-resident scene

namespace p3d {

// ---- vector.h / vector.cpp -------------------------------------------------
struct Vector {
  float x = 0.f, y = 0.f, z = 0.f;
  Vector() = default;
  Vector(float ax, float ay, float az) : x(ax), y(ay), z(az) {}
  float sqrdLength() const { return x * x + y * y + z * z; }
  float length() const { return std::sqrt(sqrdLength()); }
  float getIndex(int op) const { return op == 0 ? x : (op == 1 ? y : z); }
  // vector.cpp:65-70: reciprocal taken in double, narrowed to float, applied in place
  Vector& normalize() {
    const float inv = static_cast<float>(1.0 / static_cast<double>(length()));
    x *= inv; y *= inv; z *= inv;
    return *this;
  }
  Vector operator+(const Vector& o) const { return {x + o.x, y + o.y, z + o.z}; }
  Vector operator-(const Vector& o) const { return {x - o.x, y - o.y, z - o.z}; }
  Vector operator*(float f) const { return {x * f, y * f, z * f}; }
  Vector operator/(float f) const { return {x / f, y / f, z / f}; }
  float operator*(const Vector& o) const { return x * o.x + y * o.y + z * o.z; }  // inner product
  Vector operator%(const Vector& o) const {                                       // cross product
    return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x};
  }
};

// ---- ray.h ------------------------------------------------------------------
class Ray {
 public:
  Ray(const Vector& o, const Vector& dir, int ix = 0, int jx = 0) : origin(o), direction(dir), i(ix), j(jx) {}
  Vector origin;
  Vector direction;
  int i, j;
  uint64_t id = 0;
  Vector getDirection() { return direction.normalize(); }  // ray.h:16-18: normalises IN PLACE, every call
};

// ---- color.h ----------------------------------------------------------------
class Color {
 public:
  Color() = default;
  Color(float r, float g, float b) : R(r), G(g), B(b) {}
  float r() const { return R; }
  float g() const { return G; }
  float b() const { return B; }
  float r(float v) { return R = v; }
  float g(float v) { return G = v; }
  float b(float v) { return B = v; }
  float sum() const { return R + G + B; }
  Color operator/(float c) const { return {R / c, G / c, B / c}; }

 private:
  float R = 0.f, G = 0.f, B = 0.f;
};

// ---- boundingBox.h ----------------------------------------------------------
struct AABB {
  Vector min{-1.f, -1.f, -1.f}, max{1.f, 1.f, 1.f};  // boundingBox.cpp:6-10: default box is [-1,1]^3
  AABB() = default;
  AABB(const Vector& lo, const Vector& hi) : min(lo), max(hi) {}
  static AABB empty() { return {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; }
  Vector centroid() const { return (min + max) / 2; }
  void extend(const AABB& o) {  // boundingBox.cpp:104-112
    if (min.x > o.min.x) min.x = o.min.x;
    if (min.y > o.min.y) min.y = o.min.y;
    if (min.z > o.min.z) min.z = o.min.z;
    if (max.x < o.max.x) max.x = o.max.x;
    if (max.y < o.max.y) max.y = o.max.y;
    if (max.z < o.max.z) max.z = o.max.z;
  }
};

// ---- scene.h:34-71 ------------------------------------------------------------
class Material {
 public:
  Material() = default;
  Material(const Color& c, float Kd, const Color& cs, float Ks, float Shine, float T, float ior,
           const Color& E)
      : m_diffColor(c), m_specColor(cs), m_E(E), m_Refl(Ks), m_T(T), m_Diff(Kd), m_Shine(Shine),
        m_Spec(Ks), m_RIndex(ior) {}  // m_Refl = Ks (scene.h:42)
  Color GetDiffColor() const { return m_diffColor; }
  Color GetSpecColor() const { return m_specColor; }
  Color GetEmission() const { return m_E; }
  float GetDiffuse() const { return m_Diff; }
  float GetSpecular() const { return m_Spec; }
  float GetShine() const { return m_Shine; }
  float GetReflection() const { return m_Refl; }
  float GetTransmittance() const { return m_T; }
  float GetRefrIndex() const { return m_RIndex; }
  void SetDiffColor(const Color& c) { m_diffColor = c; }
  void SetSpecColor(const Color& c) { m_specColor = c; }
  void SetEmission(const Color& c) { m_E = c; }
  void SetDiffuse(float v) { m_Diff = v; }
  void SetSpecular(float v) { m_Spec = v; }
  void SetShine(float v) { m_Shine = v; }
  void SetReflection(float v) { m_Refl = v; }
  void SetTransmittance(float v) { m_T = v; }
  void SetRefrIndex(float v) { m_RIndex = v; }

 private:
  Color m_diffColor{0.2f, 0.2f, 0.2f}, m_specColor{1.f, 1.f, 1.f}, m_E{0.f, 0.f, 0.f};
  float m_Refl = 1.f, m_T = 0.f, m_Diff = 0.2f, m_Shine = 20.f, m_Spec = 0.8f, m_RIndex = 1.f;
};

struct Light {  // scene.h:73-81
  Light(const Vector& pos, const Color& col) : position(pos), color(col) {}
  Vector position;
  Color color;
};

// ---- scene.h:83-179 -------------------------------------------------------------
enum class Kind : uint32_t { Sphere = 0, Triangle = 1, Box = 2, Plane = 3 };

class Object {
 public:
  virtual ~Object() = default;
  Material* GetMaterial() const { return m_Material; }
  void SetMaterial(Material* m) { m_Material = m; }
  virtual Kind kind() const = 0;
  virtual AABB GetBoundingBox() const = 0;
  virtual Vector getCentroid() const = 0;
  // nine geometry floats + shading normal in the layout of p3d_prim (include/p3d.h)
  virtual void pack(float v[9], float n[3]) const = 0;
  // scene.h:88-89.  One query on the device (p3d_object_intercepts / p3d_object_normal); the sphere test leaves
  // r.direction normalised, as the reference's does (scene.cpp:156).
  bool intercepts(Ray& r, float& t);
  Vector getNormal(Vector point);
  void bind(::p3d_scene* dev, uint32_t index) { dev_ = dev; index_ = index; }

 protected:
  Material* m_Material = nullptr;
  ::p3d_scene* dev_ = nullptr;  // set by Scene::bindDevice
  uint32_t index_ = 0;          // position in Scene::objects = hit ID
};

class Sphere final : public Object {
 public:
  Sphere(const Vector& c, float r) : center(c), radius(r) {}
  Kind kind() const override { return Kind::Sphere; }
  AABB GetBoundingBox() const override {  // scene.cpp:194-198
    const Vector rr(radius, radius, radius);
    return {center - rr, center + rr};
  }
  Vector getCentroid() const override { return center; }
  Vector GetCenter() const { return center; }
  float GetRadius() const { return radius; }
  void pack(float v[9], float n[3]) const override;

 private:
  Vector center;
  float radius;
};

class Triangle final : public Object {
 public:
  Triangle(const Vector& P0, const Vector& P1, const Vector& P2);  // scene.cpp:12-35
  Kind kind() const override { return Kind::Triangle; }
  AABB GetBoundingBox() const override { return {Min, Max}; }
  Vector getCentroid() const override { return GetBoundingBox().centroid(); }  // scene.h:123-125
  void pack(float v[9], float n[3]) const override;

 private:
  Vector points[3], normal, Min, Max;
};

class aaBox final : public Object {
 public:
  aaBox(const Vector& lo, const Vector& hi) : min(lo), max(hi) {}
  Kind kind() const override { return Kind::Box; }
  AABB GetBoundingBox() const override { return {min, max}; }
  Vector getCentroid() const override { return (max + min) / 2; }  // scene.cpp:269-271
  void pack(float v[9], float n[3]) const override;

 private:
  Vector min, max;
};

class Plane final : public Object {
 public:
  Plane(const Vector& P0, const Vector& P1, const Vector& P2);  // scene.cpp:102-113
  Kind kind() const override { return Kind::Plane; }
  AABB GetBoundingBox() const override { return AABB(); }       // scene.h:114 (Q12)
  Vector getCentroid() const override { return Vector(); }      // scene.h:112
  void pack(float v[9], float n[3]) const override;

 private:
  Vector PN, A;
};

// ---- camera.h -----------------------------------------------------------------
class Camera {
 public:
  Camera(const Vector& from, const Vector& At, const Vector& Up, float angle, float hither, float yon,
         int ResX, int ResY, float Aperture_ratio, float Focal_ratio);
  int GetResX() const { return res_x; }
  int GetResY() const { return res_y; }
  float GetFov() const { return fovy; }
  float GetPlaneDist() const { return plane_dist; }
  float GetFar() const { return vfar; }
  float GetAperture() const { return aperture; }
  Ray PrimaryRay(const Vector& pixel_sample);                             // camera.h:65-82
  Ray PrimaryRay(const Vector& lens_sample, const Vector& pixel_sample);  // camera.h:84-115 (thin lens)
  // raw state for the device descriptor
  Vector eye, at, up, u, v, n;
  float fovy, vnear, vfar, plane_dist, focal_ratio, aperture, w, h, aperture_ratio;
  int res_x, res_y;
};

// ---- scene.h:182-225 ------------------------------------------------------------
class Scene {
 public:
  Camera* GetCamera() const { return camera.get(); }
  void SetCamera(Camera* c) { camera.reset(c); }
  Color GetBackgroundColor() const { return bgColor; }
  void SetBackgroundColor(const Color& c) { bgColor = c; }
  bool GetSkyBoxFlg() const { return SkyBoxFlg; }
  void SetSkyBoxFlg(bool f) { SkyBoxFlg = f; }
  const std::string& GetSkyboxDir() const { return skyboxDir; }
  // scene.cpp:329-377: the six faces <dir>/{right,left,top,bottom,front,back}.jpg.  The reference decodes them through
  // DevIL; here host/jpeg_decode.cpp does (baseline JPEG; a face missing as .jpg is read from a binary .ppm of the same
  // name).  Kept bottom row first (IL_ORIGIN_LOWER_LEFT).  Uploaded to the bound device scene, now or at bindDevice.
  bool LoadSkybox(const char* sky_dir);
  // skybox_img[face] of scene.h:218-223: decoded RGB bytes (bottom row first) and size; nullptr before LoadSkybox
  const uint8_t* SkyboxFace(int face, uint32_t* res_x, uint32_t* res_y) const {
    if (!skybox_loaded || face < 0 || face > 5) return nullptr;
    if (res_x) *res_x = skybox_img[face].resX;
    if (res_y) *res_y = skybox_img[face].resY;
    return skybox_img[face].img.data();
  }
  bool SkyboxLoaded() const { return skybox_loaded; }
  Color GetSkyboxColor(Ray& r);  // scene.cpp:379-457, one lookup on the device (p3d_skybox_color)
  // The device scene that answers the ray queries of this scene's objects (p3d_scene_create of this scene's
  // descriptor).  Not owned.  nullptr unbinds.
  bool bindDevice(::p3d_scene* dev);
  ::p3d_scene* device() const { return dev_; }

  int getNumObjects() const { return static_cast<int>(objects.size()); }
  void addObject(Object* o) { o->bind(dev_, static_cast<uint32_t>(objects.size())); objects.emplace_back(o); }
  Object* getObject(unsigned i) const { return i < objects.size() ? objects[i].get() : nullptr; }
  int getNumLights() const { return static_cast<int>(lights.size()); }
  void addLight(Light* l) { lights.emplace_back(l); }
  Light* getLight(unsigned i) const { return i < lights.size() ? lights[i].get() : nullptr; }
  void setLights(std::vector<std::unique_ptr<Light>> ls) { lights = std::move(ls); }

  // scene.cpp:472-628.  legacy_f11: also accept the 11-number `f` line (extension).
  // Returns false only when the file cannot be opened (the reference returns true always).
  bool load_p3f(const char* name, bool legacy_f11 = false);

  int materialIndex(const Material* m) const;
  const std::vector<std::unique_ptr<Material>>& allMaterials() const { return materials; }

  // the `v` block as parsed, so the camera can be re-created (resolution / lens overrides)
  struct ViewBlock {
    Vector from, at, up;
    float angle = 0, hither = 0, aperture = 0, focal = 0;
    int xres = 0, yres = 0;
    bool present = false;
  } view;
  void rebuildCamera();

 private:
  std::vector<std::unique_ptr<Object>> objects;
  std::vector<std::unique_ptr<Light>> lights;
  std::vector<std::unique_ptr<Material>> materials;
  std::unique_ptr<Camera> camera;
  Color bgColor;
  bool SkyBoxFlg = false;
  std::string skyboxDir;
  ::p3d_scene* dev_ = nullptr;
  struct Face { std::vector<uint8_t> img; uint32_t resX = 0, resY = 0, BPP = 3; } skybox_img[6];  // scene.h:218-223
  bool skybox_loaded = false;
};

}  // namespace p3d

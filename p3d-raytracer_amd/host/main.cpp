// p3d_render — the console front end.  Does what the reference's main() does when its GL
// window is off (main.cpp:1004-1019): ask for a scene, load it, render a frame, print the
// time, save the image — with the frame loop (main.cpp:747-820) running on the MI355X
// through include/p3d.h.  Options that are #defines in constants.h are flags here.
//
//   p3d_render [scene.p3f] [--whitted|--pathtrace] [--accel none|grid|bvh] [--depth N]
//              [--spp N(sqrt)] [--aa 0|1] [--dof 0|1] [--soft 0|1] [--tent] [--gamma G]
//              [--res W H] [--seed S] [--legacy-f11] [--device D]
//              [--out image.png|image.ppm]   default RT_Output.png, as saveImgFile writes it (main.cpp:674-689,851);
//                               PNG through zlib (8-bit RGB, no DevIL), a name ending in .ppm gives a binary PPM
//              [--stack literal|per_pixel]   p3d_config.stack_mode (default literal: the reference's one hit_stack)
//              [--device-bvh]   build a linear BVH on the GPU instead of the reference's tree on the host
//                               (p3d_scene_create_device_bvh: same closest hits, shadow feelers may differ)
//              [--skybox DIR]   DIR/{right,left,top,bottom,front,back}.ppm (binary P6; convert the
//                               reference's JPEGs once with scenes/skybox_to_ppm.py) -> SKYBOX true
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <zlib.h>

#include "p3d.h"

namespace {

int die(const char* what) {
  std::fprintf(stderr, "%s: %s\n", what, p3d_last_error());
  return 1;
}

// img_Data is stored bottom row first (main.cpp:818-820, y = 0 at the bottom); image files
// start with the top row, so rows are written in reverse.
bool save_ppm(const std::string& path, const std::vector<uint8_t>& rgb8, int w, int h) {
  std::ofstream f(path, std::ios::binary);
  if (!f) return false;
  f << "P6\n" << w << " " << h << "\n255\n";
  for (int y = h - 1; y >= 0; --y) f.write(reinterpret_cast<const char*>(&rgb8[(size_t)y * w * 3]), (std::streamsize)w * 3);
  return (bool)f;
}

// saveImgFile (main.cpp:674-689) without DevIL: 8-bit RGB PNG, one IDAT chunk, filter type 0 on every scanline.
bool save_png(const std::string& path, const std::vector<uint8_t>& rgb8, int w, int h) {
  std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
  for (int y = 0; y < h; ++y) {  // file rows are top-down, img_Data is bottom-up
    uint8_t* row = &raw[(size_t)y * (1 + (size_t)w * 3)];
    row[0] = 0;
    std::memcpy(row + 1, &rgb8[(size_t)(h - 1 - y) * w * 3], (size_t)w * 3);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
  std::ofstream f(path, std::ios::binary);
  if (!f) return false;
  auto be32 = [](uint32_t v, uint8_t* p) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
  auto chunk = [&](const char type[4], const uint8_t* data, uint32_t n) {
    uint8_t len[4], crcb[4];
    be32(n, len);
    f.write(reinterpret_cast<const char*>(len), 4);
    f.write(type, 4);
    if (n) f.write(reinterpret_cast<const char*>(data), n);
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
    if (n) crc = crc32(crc, data, n);
    be32((uint32_t)crc, crcb);
    f.write(reinterpret_cast<const char*>(crcb), 4);
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write(reinterpret_cast<const char*>(sig), 8);
  uint8_t ihdr[13];
  be32((uint32_t)w, ihdr); be32((uint32_t)h, ihdr + 4);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8 bits, colour type 2 (RGB)
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), (uint32_t)zlen);
  chunk("IEND", nullptr, 0);
  return (bool)f;
}

// One cubemap face from a binary PPM, stored bottom row first like DevIL's lower-left origin
// (scene.cpp:344-345).  JPEG decoding is deliberately not part of this program.
bool load_ppm_face(const std::string& path, std::vector<uint8_t>& bytes, uint32_t& w, uint32_t& h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string magic;
  int maxv = 0;
  f >> magic >> w >> h >> maxv;
  if (magic != "P6" || maxv != 255 || w == 0 || h == 0) return false;
  f.get();
  std::vector<uint8_t> top_down((size_t)w * h * 3);
  f.read(reinterpret_cast<char*>(top_down.data()), (std::streamsize)top_down.size());
  if (!f) return false;
  bytes.resize(top_down.size());
  for (uint32_t y = 0; y < h; ++y) std::memcpy(&bytes[(size_t)y * w * 3], &top_down[(size_t)(h - 1 - y) * w * 3], (size_t)w * 3);
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  p3d_config cfg;
  p3d_config_default(&cfg);
  std::string scene_path, skybox_dir, out = "RT_Output.png";  // main.cpp:851
  int res_w = 0, res_h = 0, device = 0;
  bool device_bvh = false;
  uint32_t load_flags = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "--whitted") cfg.integrator = P3D_WHITTED;
    else if (a == "--pathtrace") cfg.integrator = P3D_PATHTRACE;
    else if (a == "--accel") {
      const std::string v = next("--accel");
      cfg.accel = v == "none" ? P3D_ACCEL_NONE : (v == "grid" ? P3D_ACCEL_GRID : P3D_ACCEL_BVH);
    } else if (a == "--depth") cfg.max_depth = std::atoi(next("--depth"));
    else if (a == "--spp") cfg.spp_sqrt = (uint32_t)std::atoi(next("--spp"));
    else if (a == "--aa") cfg.antialiasing = (uint32_t)std::atoi(next("--aa"));
    else if (a == "--dof") cfg.depth_of_field = (uint32_t)std::atoi(next("--dof"));
    else if (a == "--soft") cfg.soft_shadows = (uint32_t)std::atoi(next("--soft"));
    else if (a == "--tent") cfg.sample_mode = P3D_SAMPLE_TENT;
    else if (a == "--gamma") cfg.gamma = (float)std::atof(next("--gamma"));
    else if (a == "--seed") cfg.seed = std::strtoull(next("--seed"), nullptr, 0);
    else if (a == "--res") { res_w = std::atoi(next("--res")); res_h = std::atoi(next("--res")); }
    else if (a == "--legacy-f11") load_flags |= P3D_LOAD_LEGACY_F11;
    else if (a == "--out") out = next("--out");
    else if (a == "--skybox") skybox_dir = next("--skybox");
    else if (a == "--stack") cfg.stack_mode = std::string(next("--stack")) == "per_pixel" ? P3D_STACK_PER_PIXEL : P3D_STACK_LITERAL;
    else if (a == "--device") device = std::atoi(next("--device"));
    else if (a == "--device-bvh") device_bvh = true;
    else if (a[0] != '-') scene_path = a;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (scene_path.empty()) {  // main.cpp:968-980: prompt for a name under P3D_Scenes/
    std::string name;
    std::cout << "Input the Scene Name: ";
    if (!(std::cin >> name)) return 2;
    scene_path = "P3D_Scenes/" + name;
  }

  p3d_host_scene* hs = nullptr;
  if (p3d_host_scene_load(scene_path.c_str(), load_flags, &hs) != P3D_OK) {
    std::printf("\nError opening P3F file.\n");
    return die("load");
  }
  if (res_w > 0 && p3d_host_scene_set_resolution(hs, res_w, res_h) != P3D_OK) return die("resolution");
  if (cfg.soft_shadows && !cfg.antialiasing &&  // main.cpp:725-745
      p3d_host_scene_replicate_lights(hs, cfg.spp_sqrt, cfg.light_side) != P3D_OK)
    return die("lights");

  const auto t_build0 = std::chrono::high_resolution_clock::now();
  const p3d_scene_desc* desc = nullptr;
  if (p3d_host_scene_desc(hs, cfg.accel == P3D_ACCEL_BVH && !device_bvh, cfg.accel == P3D_ACCEL_GRID, &desc) != P3D_OK) return die("flatten");
  const int W = desc->camera.res_x, H = desc->camera.res_y;
  std::printf("\nResolutionX = %d  ResolutionY= %d.\n", W, H);  // main.cpp:986
  p3d_scene* scene = nullptr;
  if (device_bvh && cfg.accel == P3D_ACCEL_BVH) {
    float build_ms = 0;
    if (p3d_scene_create_device_bvh(desc, device, &scene, &build_ms) != P3D_OK) return die("scene_create_device_bvh");
    std::printf("BVH built on the GPU in %.2f ms\n", build_ms);
  } else if (p3d_scene_create(desc, device, &scene) != P3D_OK) {
    return die("scene_create");
  }
  if (!skybox_dir.empty()) {  // Scene::LoadSkybox (scene.cpp:329-377) + SKYBOX true (constants.h:30)
    static const char* names[6] = {"right", "left", "top", "bottom", "front", "back"};
    std::vector<uint8_t> bytes[6];
    p3d_skybox_desc sky{};
    for (int i = 0; i < 6; ++i) {
      uint32_t w = 0, h = 0;
      if (!load_ppm_face(skybox_dir + "/" + names[i] + ".ppm", bytes[i], w, h)) {
        std::fprintf(stderr, "cannot read %s/%s.ppm\n", skybox_dir.c_str(), names[i]);
        return 1;
      }
      std::printf("Skybox face %d: Image sucessfully loaded.\n", i);  // scene.cpp:352
      sky.face[i].img = bytes[i].data(); sky.face[i].res_x = w; sky.face[i].res_y = h; sky.face[i].bpp = 3;
    }
    if (p3d_scene_set_skybox(scene, &sky) != P3D_OK) return die("skybox");
    cfg.skybox = 1;
  }
  const auto t_build1 = std::chrono::high_resolution_clock::now();

  std::vector<uint8_t> img((size_t)3 * W * H);
  p3d_tile tile{0, 0, W, H, 0, 1};
  p3d_stats st{};
  cfg.collect_stats = 1;
  const auto t0 = std::chrono::high_resolution_clock::now();
  if (p3d_render_tile(scene, &cfg, &tile, nullptr, nullptr, img.data(), &st) != P3D_OK) return die("render");
  const auto t1 = std::chrono::high_resolution_clock::now();
  std::printf("Drawing finished!\n");
  const double secs = std::chrono::duration<double>(t1 - t0).count();
  const uint64_t rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract + st.rays_bounce + st.rays_light;
  std::printf("\nDone: %.2f (sec)\n", secs);  // main.cpp:1012
  std::printf("accel build + upload %.3f s; kernel %.3f ms; %llu rays; %.1f Mrays/s (kernel)\n",
              std::chrono::duration<double>(t_build1 - t_build0).count(), st.kernel_ms, (unsigned long long)rays,
              st.kernel_ms > 0 ? rays / (st.kernel_ms * 1e3) : 0.0);
  const bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
  if (!(ppm ? save_ppm(out, img, W, H) : save_png(out, img, W, H))) {
    std::printf("Error saving Image file\n");
    return 1;
  }
  std::printf("Image file created\n");
  p3d_scene_destroy(scene);
  p3d_host_scene_destroy(hs);
  return 0;
}

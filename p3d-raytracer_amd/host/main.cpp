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
//              [--skybox DIR]   DIR/{right,left,top,bottom,front,back}.jpg (baseline JPEG, decoded by the library as
//                               Scene::LoadSkybox asks DevIL to; .ppm accepted too) -> SKYBOX true.  A scene's own
//                               `env <dir>` line does the same when the folder is found.
//              [--gpus N]       the frame's rows dealt to N GPUs of this node in 8-row stripes (one device scene each, this
//                               one process driving them), every GPU's part of the image brought to GPU 0 by ONE
//                               ncclGather over xGMI (RCCL, loaded at run time), de-interleaved on the host.  Same image
//                               as one GPU renders, bit for bit (P3D_STACK_LITERAL included: include/p3d.h, p3d_tile).
//              [--verify]       with --gpus: one more frame whose float RGB + hit IDs (16 B/px, what bench.py gathers) go
//                               through a second ncclGather; GPU 0 then renders the whole frame alone and the gathered
//                               frame is compared with it bit for bit - colours, hit IDs and the u8 image.  Prints the
//                               verdict; a mismatch is exit code 3.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
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

// ---- --gpus N: one process, N devices, one RCCL gather per frame (ncclGather, rccl.h:745) ----
// RCCL is loaded with dlopen: a run without --gpus neither needs nor loads it.  Only the five entry points used here
// are declared (types as in rccl.h: ncclComm_t is an opaque pointer, ncclResult_t / ncclDataType_t are enums; ncclUint8 = 1).
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Gather)(const void* send, void* recv, size_t count, int datatype, int root, void* comm, hipStream_t stream) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { std::fprintf(stderr, "--gpus: cannot load librccl.so: %s\n", dlerror()); return false; }
    auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) std::fprintf(stderr, "--gpus: librccl.so has no %s\n", n); return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Gather = (decltype(Gather))sym("ncclGather");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString;
  }
};
constexpr int kNcclUint8 = 1;  // rccl.h: ncclUint8
constexpr int kStripeRows = 8;

// Renders the frame on n GPUs into `img` (bottom row first, 3 bytes per pixel).  Returns 0, or 1 after printing why not.
int render_multi_gpu(const p3d_scene_desc* desc, bool device_bvh, p3d_config cfg, int n, bool verify, std::vector<uint8_t>& img, double* secs) {
  const int W = desc->camera.res_x, H = desc->camera.res_y;
  int ndev = p3d_device_count();
  if (n > ndev) { std::fprintf(stderr, "--gpus %d: this node shows %d HIP device(s)\n", n, ndev); return 1; }
  if (H % (kStripeRows * n) != 0) { std::fprintf(stderr, "--gpus %d: the image height %d is not a multiple of %d rows\n", n, H, kStripeRows * n); return 1; }
  Rccl rccl;
  if (!rccl.load()) return 1;
  auto hip_ok = [](hipError_t e, const char* what) { if (e != hipSuccess) std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return e == hipSuccess; };
  auto nccl_ok = [&](int r, const char* what) { if (r != 0) std::fprintf(stderr, "%s: %s\n", what, rccl.GetErrorString(r)); return r == 0; };
  std::vector<int> devs(n);
  for (int d = 0; d < n; ++d) devs[d] = d;
  std::vector<void*> comms(n, nullptr);
  if (!nccl_ok(rccl.CommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll")) return 1;
  const size_t part = (size_t)3 * W * (H / n);  // bytes of one GPU's stripes
  std::vector<p3d_scene*> scenes(n, nullptr);
  std::vector<hipStream_t> streams(n, nullptr);
  std::vector<uint8_t*> d_part(n, nullptr);
  uint8_t* d_all = nullptr;
  cfg.collect_stats = 0;
  for (int d = 0; d < n; ++d) {
    const int rc = device_bvh && cfg.accel == P3D_ACCEL_BVH ? p3d_scene_create_device_bvh(desc, d, &scenes[d], nullptr) : p3d_scene_create(desc, d, &scenes[d]);
    if (rc != P3D_OK) return die("scene_create");
    if (!hip_ok(hipSetDevice(d), "hipSetDevice") || !hip_ok(hipStreamCreate(&streams[d]), "hipStreamCreate") ||
        !hip_ok(hipMalloc((void**)&d_part[d], part), "hipMalloc"))
      return 1;
    if (d == 0 && !hip_ok(hipMalloc((void**)&d_all, part * n), "hipMalloc")) return 1;
  }
  auto frame = [&]() {
    for (int d = 0; d < n; ++d) {  // every GPU renders its stripes: rows d*8 .. d*8+7, then every n-th stripe
      p3d_tile t{0, d * kStripeRows, W, H / n, kStripeRows, n};
      if (p3d_render_tile_device(scenes[d], &cfg, &t, nullptr, nullptr, d_part[d], streams[d], nullptr) != P3D_OK) return false;
    }
    if (!nccl_ok(rccl.GroupStart(), "ncclGroupStart")) return false;
    for (int d = 0; d < n; ++d)
      if (!nccl_ok(rccl.Gather(d_part[d], d == 0 ? d_all : nullptr, part, kNcclUint8, 0, comms[d], streams[d]), "ncclGather")) return false;
    if (!nccl_ok(rccl.GroupEnd(), "ncclGroupEnd")) return false;
    for (int d = 0; d < n; ++d)
      if (!hip_ok(hipSetDevice(d), "hipSetDevice") || !hip_ok(hipStreamSynchronize(streams[d]), "hipStreamSynchronize")) return false;
    return true;
  };
  if (!frame()) return 1;  // first frame: code objects loaded, tile schedules and halo chains recorded
  const auto t0 = std::chrono::high_resolution_clock::now();
  if (!frame()) return 1;
  *secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  for (int d = 0; d < n; ++d)
    if (p3d_scene_status(scenes[d]) != P3D_OK) return die("render");
  std::vector<uint8_t> all(part * n);
  if (!hip_ok(hipSetDevice(0), "hipSetDevice") || !hip_ok(hipMemcpy(all.data(), d_all, all.size(), hipMemcpyDeviceToHost), "hipMemcpy")) return 1;
  // de-interleave: stripe k of GPU d holds frame rows (k * n + d) * 8 .. + 7
  const size_t row_bytes = (size_t)3 * W;
  for (int d = 0; d < n; ++d)
    for (int r = 0; r < H / n; ++r) {
      const int y = ((r / kStripeRows) * n + d) * kStripeRows + r % kStripeRows;
      std::memcpy(&img[(size_t)y * row_bytes], &all[(size_t)d * part + (size_t)r * row_bytes], row_bytes);
    }
  int verdict = 0;
  if (verify) {
    // One more frame, all outputs: every GPU renders float RGB + hit IDs (packed: 12 B/px of colour, then 4 B/px of IDs,
    // for its own rows) next to the u8 image; the packed buffers go to GPU 0 through a second gather.  Then GPU 0 renders
    // the whole frame alone and the de-interleaved gathered frame must equal it bit for bit (SURVEY.md 8(e) invariant).
    const size_t px_part = (size_t)W * (H / n), packed = px_part * 16;
    std::vector<uint8_t*> d_packed(n, nullptr);
    uint8_t* d_packed_all = nullptr;
    bool ok = true;
    for (int d = 0; d < n && ok; ++d) {
      ok = hip_ok(hipSetDevice(d), "hipSetDevice") && hip_ok(hipMalloc((void**)&d_packed[d], packed), "hipMalloc");
      if (ok && d == 0) ok = hip_ok(hipMalloc((void**)&d_packed_all, packed * n), "hipMalloc");
    }
    for (int d = 0; d < n && ok; ++d) {
      p3d_tile t{0, d * kStripeRows, W, H / n, kStripeRows, n};
      ok = p3d_render_tile_device(scenes[d], &cfg, &t, (float*)d_packed[d], (int32_t*)(d_packed[d] + px_part * 12), d_part[d], streams[d], nullptr) == P3D_OK;
      if (!ok) std::fprintf(stderr, "--verify render: %s\n", p3d_last_error());
    }
    if (ok) ok = nccl_ok(rccl.GroupStart(), "ncclGroupStart");
    for (int d = 0; d < n && ok; ++d)
      ok = nccl_ok(rccl.Gather(d_part[d], d == 0 ? d_all : nullptr, part, kNcclUint8, 0, comms[d], streams[d]), "ncclGather") &&
           nccl_ok(rccl.Gather(d_packed[d], d == 0 ? d_packed_all : nullptr, packed, kNcclUint8, 0, comms[d], streams[d]), "ncclGather");
    if (ok) ok = nccl_ok(rccl.GroupEnd(), "ncclGroupEnd");
    for (int d = 0; d < n && ok; ++d) ok = hip_ok(hipSetDevice(d), "hipSetDevice") && hip_ok(hipStreamSynchronize(streams[d]), "hipStreamSynchronize") && p3d_scene_status(scenes[d]) == P3D_OK;
    const size_t px = (size_t)W * H;
    std::vector<float> g_rgb(px * 3), one_rgb(px * 3);
    std::vector<int32_t> g_hit(px), one_hit(px);
    std::vector<uint8_t> g_u8(px * 3), one_u8(px * 3), raw(packed * n), raw8(part * n);
    if (ok) ok = hip_ok(hipSetDevice(0), "hipSetDevice") && hip_ok(hipMemcpy(raw.data(), d_packed_all, raw.size(), hipMemcpyDeviceToHost), "hipMemcpy") &&
                 hip_ok(hipMemcpy(raw8.data(), d_all, raw8.size(), hipMemcpyDeviceToHost), "hipMemcpy");
    for (int d = 0; d < n && ok; ++d)
      for (int r = 0; r < H / n; ++r) {
        const int y = ((r / kStripeRows) * n + d) * kStripeRows + r % kStripeRows;
        const uint8_t* base = &raw[(size_t)d * packed];
        std::memcpy(&g_rgb[(size_t)y * W * 3], base + (size_t)r * W * 12, (size_t)W * 12);
        std::memcpy(&g_hit[(size_t)y * W], base + px_part * 12 + (size_t)r * W * 4, (size_t)W * 4);
        std::memcpy(&g_u8[(size_t)y * row_bytes], &raw8[(size_t)d * part + (size_t)r * row_bytes], row_bytes);
      }
    if (ok) {  // the whole frame on GPU 0 alone, through the host-buffer call
      p3d_tile full{0, 0, W, H, 0, 1};
      ok = p3d_render_tile(scenes[0], &cfg, &full, one_rgb.data(), one_hit.data(), one_u8.data(), nullptr) == P3D_OK;
      if (!ok) std::fprintf(stderr, "--verify single-GPU frame: %s\n", p3d_last_error());
    }
    for (int d = 0; d < n; ++d) { (void)hipSetDevice(d); if (d_packed[d]) (void)hipFree(d_packed[d]); }
    (void)hipSetDevice(0);
    if (d_packed_all) (void)hipFree(d_packed_all);
    if (!ok) return 1;
    size_t bad_rgb = 0, bad_hit = 0, bad_u8 = 0;
    for (size_t i = 0; i < px * 3; ++i) {
      uint32_t a, b;
      std::memcpy(&a, &g_rgb[i], 4); std::memcpy(&b, &one_rgb[i], 4);
      bad_rgb += a != b;
      bad_u8 += g_u8[i] != one_u8[i];
    }
    for (size_t i = 0; i < px; ++i) bad_hit += g_hit[i] != one_hit[i];
    const bool same = bad_rgb == 0 && bad_hit == 0 && bad_u8 == 0 && g_u8 == img;
    std::printf("verify: frame gathered from %d GPU(s) vs the frame GPU 0 renders alone (float RGB, hit IDs, u8 image): %s", n, same ? "bit-identical\n" : "MISMATCH");
    if (!same) std::printf(" (%zu colour words, %zu hit IDs, %zu bytes differ%s)\n", bad_rgb, bad_hit, bad_u8, g_u8 == img ? "" : "; the timed frame's u8 image differs from the verify frame's");
    verdict = same ? 0 : 3;
  }
  for (int d = 0; d < n; ++d) {
    (void)hipSetDevice(d);
    (void)hipFree(d_part[d]);
    (void)hipStreamDestroy(streams[d]);
    p3d_scene_destroy(scenes[d]);
    (void)rccl.CommDestroy(comms[d]);
  }
  (void)hipSetDevice(0);
  (void)hipFree(d_all);
  return verdict;
}

}  // namespace

int main(int argc, char** argv) {
  p3d_config cfg;
  p3d_config_default(&cfg);
  std::string scene_path, skybox_dir, out = "RT_Output.png";  // main.cpp:851
  int res_w = 0, res_h = 0, device = 0, gpus = 0;
  bool device_bvh = false, verify = false;
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
    else if (a == "--gpus") gpus = std::atoi(next("--gpus"));
    else if (a == "--verify") verify = true;
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
  if (gpus > 0) {  // several GPUs of this node (or one, through the same path): stripes + one RCCL gather
    if (!skybox_dir.empty()) { std::fprintf(stderr, "--gpus with --skybox is not wired up in this front end\n"); return 2; }
    std::vector<uint8_t> img((size_t)3 * W * H);
    double secs = 0;
    if (const int rc = render_multi_gpu(desc, device_bvh, cfg, gpus, verify, img, &secs)) return rc;
    std::printf("Drawing finished!\n\nDone: %.2f (sec)\n", secs);
    std::printf("%d GPU(s): stripes of %d rows, one ncclGather of %zu bytes per GPU; second frame %.3f ms (render + gather)\n", gpus, kStripeRows,
                (size_t)3 * W * (H / gpus), secs * 1e3);
    const bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
    if (!(ppm ? save_ppm(out, img, W, H) : save_png(out, img, W, H))) { std::printf("Error saving Image file\n"); return 1; }
    std::printf("Image file created\n");
    p3d_host_scene_destroy(hs);
    return 0;
  }
  p3d_scene* scene = nullptr;
  if (device_bvh && cfg.accel == P3D_ACCEL_BVH) {
    float build_ms = 0;
    if (p3d_scene_create_device_bvh(desc, device, &scene, &build_ms) != P3D_OK) return die("scene_create_device_bvh");
    std::printf("BVH built on the GPU in %.2f ms\n", build_ms);
  } else if (p3d_scene_create(desc, device, &scene) != P3D_OK) {
    return die("scene_create");
  }
  if (skybox_dir.empty() && p3d_host_scene_has_skybox(hs)) {  // `env <dir>` in the scene file, faces found as PPMs: SKYBOX true (constants.h:30)
    if (p3d_host_scene_bind_device(hs, scene) != P3D_OK) return die("skybox");
    cfg.skybox = 1;
  }
  if (!skybox_dir.empty()) {  // Scene::LoadSkybox (scene.cpp:329-377) + SKYBOX true (constants.h:30)
    if (p3d_host_scene_load_skybox(hs, skybox_dir.c_str()) != P3D_OK) return die("skybox");
    if (p3d_host_scene_bind_device(hs, scene) != P3D_OK) return die("skybox");
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

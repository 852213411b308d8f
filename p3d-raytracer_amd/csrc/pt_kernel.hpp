// pt_kernel.hpp — the path tracer (Radiance, main.cpp:313-516) as a persistent per-lane
// bounce loop.
//
// The reference recurses:  R = E + e + f * R(child).  Here every lane owns one pixel (or, from
// 16 samples per pixel, a quarter of one: SUB = 4 below) and runs ONE flat loop whose body is
// a single bounce; a lane whose path ended starts its next
// sample (or pops its pending dielectric branch) in the same iteration instead of waiting
// for the slowest lane of the wave, so the wave only idles lanes in the very last
// iterations of a tile (the "persistent threads" bounce loop BASELINE.json asks for).  Radiance is carried as
//     L += T * (E + e);   T *= f
// which is the same sum evaluated outermost-first (the recursion evaluates innermost-
// first): results agree with the recursive oracle to float rounding (~1e-6 relative),
// far inside the 1e-4 tolerance; hit decisions and the RNG stream are identical.
//
// sin/cos: the reference calls libm (cosf/sinf main.cpp:400, cos/sin main.cpp:436).  libm
// differs between platforms by an ulp, which flips rare hit decisions and would make CPU
// and GPU paths diverge visibly at 256 spp.  Both this kernel and the oracle therefore use
// the same explicit double-precision routine (det_sincos) built from + - * floor only.
#pragma once

#include "kernels.hpp"

#ifndef P3D_PT_WAVES
#define P3D_PT_WAVES 4  // minimum waves per SIMD asked of the register allocator: 128 VGPRs, 12-15 spilled dwords for the BVH
                        // instantiations.  cfg3: 3 waves 190 ms, 4 waves 160 ms, 5 waves (96 VGPRs, 51 spilled) 177 ms.  (Before the
                        // deferred dielectric branches left LDS, 4 waves did not fit and 3 was the optimum.)
#endif

namespace p3d {

__device__ __forceinline__ void det_sincos(double x, double& s_out, double& c_out) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
  const double kd = floor(x * two_over_pi + 0.5);
  const double r = (x - kd * pio2_hi) - kd * pio2_lo;
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double sp = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double s = r + (z * r) * (S1 + z * sp);
  const double cp = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double c = 1.0 - (0.5 * z - z * cp);
  const long long k = (long long)kd;
  switch (k & 3) {
    case 0: s_out = s; c_out = c; break;
    case 1: s_out = c; c_out = -s; break;
    case 2: s_out = -s; c_out = -c; break;
    default: s_out = -c; c_out = s; break;
  }
}

constexpr float kPIf = 3.141592653589793238462f;  // camera.h:13

// pending dielectric branch (main.cpp:512-513): 3 float4 per entry, 2 entries per lane, in the
// global scratch the Whitted kernel uses for its level records (rarely touched: only the first
// two bounces on glass fork; 6 KB of LDS per wave would cost the path tracer a wave per SIMD)
struct Pending {
  float4* base;     // &scratch[thread]; float4 q of entry e at base[(e * 3 + q) * stride]
  uint32_t stride;  // threads of the launch
  int n;
};

template <int ACCEL, bool LDS, bool STATS, int SUB = 1>
__global__ void __launch_bounds__(kBlock, P3D_PT_WAVES) pt_kernel(const RenderParams P) {
  extern __shared__ float4 smem[];
  uint32_t tx, ty;
  if (!tile_of_block(P, tx, ty)) return;
  P3D_TL_BEGIN()
  const unsigned long long t_begin = P.tile_cost ? wall_clock64() : 0;
  DevScene sc = P.sc;
  stage_scene<LDS, true>(sc, P, smem);

  constexpr int PT_STACK = LDS ? kStackLds8 : kStackWindow;  // (device_core.hpp: registers set this kernel's occupancy)
  const uint32_t lane = threadIdx.x;
  constexpr int TP = SUB == 4 ? 4 : 8;                       // tile edge in pixels
  const uint32_t px = SUB == 4 ? lane >> 2 : lane;           // pixel of the tile this lane works for
  const uint32_t sub = SUB == 4 ? lane & 3u : 0u;
  const int c = (int)(tx * TP + (px % TP)), r = (int)(ty * TP + (px / TP));
  Counters<STATS> ct;
  if (STATS) reinterpret_cast<Counters<true>&>(ct).clear();
  Stack st;
  stack_bind(st, smem, P.lds_scene_f4, lane, P.stack_cap, P.spill, P.level_stride, blockIdx.x * kBlock + lane);
  Pending pend;
  pend.base = P.levels + (blockIdx.x * kBlock + lane);
  pend.stride = P.level_stride;
  pend.n = 0;
  // after the node stack (only allocated for SUB == 4); explicit LDS address space: a generic
  // pointer would compile to flat_load/flat_store, which are not ordered with the ds_* traffic
  LdsPtPixelShared& shared = *(LdsPtPixelShared*)(smem + P.lds_scene_f4 + stack_lds_f4(PT_STACK, P.stack_cap));
  if (SUB == 4 && sub == 0) {
    shared.next_start[px] = 0;
    shared.next_add[px] = 0;
    shared.first_hit[px] = -1;
    shared.colour[0][px] = 0.0f; shared.colour[1][px] = 0.0f; shared.colour[2][px] = 0.0f;
    for (int k = 0; k < kPtRing; ++k) shared.tag[k][px] = 0;
  }

  const bool active = c < P.w && r < P.h;
  if (active) {
    const int sh = P.stripe_h > 0 ? P.stripe_h : 1, ss = P.stripe_h > 0 ? P.stripe_stride : 1;
    const int x = P.x0 + c;
    const int y = P.y0 + (r / sh) * sh * ss + (r % sh);
    const int SPP = (int)P.spp_sqrt;
    const int n_samples = SPP * SPP;
    const int MAXD = P.max_depth;
    if (sub == 0) ct.add(kPixels);

    F3 color = f3(0, 0, 0);  // pixel accumulator (main.cpp:792)
    int first_hit = -1;
    int s = 0;               // SUB == 1: next sample to start; SUB == 4: the sample this lane is tracing
    int si = 0, sj = 0;
    bool alive = false, in_sample = false, first_ray = false;
    Rng rng;
    RayS ray;
    F3 T = f3(1, 1, 1), L = f3(0, 0, 0);
    int depth = 0;

#ifdef P3D_PT_PROFILE
    RegionProf prof; prof.init();
#endif
    // SUB == 4 lets lanes of one wave wait for each other (a full ring, lane 0 waiting for the last
    // samples of its pixel).  A waiting lane must never spin on its own: the loop is therefore
    // wave-uniform — its exit test is a ballot every lane of the wave takes part in, once per trip —
    // and a waiting lane simply sits out the rest of the trip.  (With a per-lane `continue` as the
    // only way round, LLVM split the wait cycle off as an inner loop for the brute-force and grid
    // instantiations and the waiting lanes starved the working ones.)  The trip bound is a backstop:
    // no lane can need more trips than the pixel's whole sample set traced by one lane.
    const unsigned long long trips_max = P.debug_trip_bound ? (unsigned long long)P.debug_trip_bound
                                                            : (unsigned long long)n_samples * (unsigned)(MAXD + 2) * 4ull + 1024ull;
    uint32_t trips_left = trips_max > 0xffffffffull ? 0xffffffffu : (uint32_t)trips_max;
    bool done = false, holding = false;
    const uint32_t spp_magic = (uint32_t)((0x100000000ull + (unsigned)SPP - 1) / (unsigned)SPP);
    while (true) {
      if (SUB == 4) {
        if (trips_left-- == 0) {  // the pixel would be written with samples missing: the call fails (P3D_ERR_CAPACITY)
          if (!done) atomicOr(P.status, kHoErrTrips);
          done = true;
        }
        if (__ballot(!done) == 0) break;
        if (done) continue;
      }
      PT_REGION(0)
      // lane 0 of a pixel adds its finished samples to the pixel colour, strictly in sample order:
      // whenever it is between two of its own samples, and every fourth trip while it traces one
      if (SUB == 4 && sub == 0 && (!alive || (trips_left & 3) == 0)) {
        uint32_t na = shared.next_add[px];
        if (na < (uint32_t)n_samples && shared.tag[na % kPtRing][px] == na + 1) {
          F3 sum = f3(shared.colour[0][px], shared.colour[1][px], shared.colour[2][px]);
          do {
            const int k = (int)(na % kPtRing);
            sum = sum + f3(shared.radiance[k][0][px], shared.radiance[k][1][px], shared.radiance[k][2][px]);
            ++na;
          } while (na < (uint32_t)n_samples && shared.tag[na % kPtRing][px] == na + 1);
          shared.colour[0][px] = sum.x; shared.colour[1][px] = sum.y; shared.colour[2][px] = sum.z;
          shared.next_add[px] = na;
        }
      }
      if (!alive) {
        PT_REGION(1)
        if (pend.n > 0) {  // resume the deferred reflection branch of a dielectric hit
          --pend.n;
          const float4 q0 = pend.base[(size_t)(pend.n * 3 + 0) * pend.stride], q1 = pend.base[(size_t)(pend.n * 3 + 1) * pend.stride],
                       q2 = pend.base[(size_t)(pend.n * 3 + 2) * pend.stride];
          ray_set(ray, f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y));
          T = f3(q1.z, q1.w, q2.x);
          depth = __float_as_int(q2.y);
          alive = true;
        } else {
          if (in_sample) {
            if (SUB == 4) {  // post the finished sample; its ring slot is free (guaranteed when it was handed out)
              const int k = s % kPtRing;
              shared.radiance[k][0][px] = L.x; shared.radiance[k][1][px] = L.y; shared.radiance[k][2][px] = L.z;
              shared.tag[k][px] = (uint32_t)s + 1;
            } else {
              color = color + L;
            }
            in_sample = false;
          }
          if (SUB == 4) {
            // Take the pixel's next sample (one LDS atomic hands simultaneous takers distinct
            // tickets), then hold it until the ring has room for its radiance: the oldest sample
            // still being traced blocks the adder, and with it the slot kPtRing samples ahead.
            if (!holding) {
              s = (int)__hip_atomic_fetch_add(&shared.next_start[px], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              holding = true;
            }
            if (s >= n_samples) {  // nothing left to start: finished, except lane 0 while samples remain to be added
              done = sub != 0 || shared.next_add[px] >= (uint32_t)n_samples;
              continue;
            }
            if ((uint32_t)s >= shared.next_add[px] + kPtRing) continue;  // wait for room
            holding = false;
            si = (int)__umulhi((uint32_t)s, spp_magic);  // s / SPP (exact: s * SPP < 2^32)
            sj = s - si * SPP;
          } else if (s == n_samples) {
            break;
          }
          rng.seed_stream(P.seed, (uint32_t)(y * sc.cam.res_x + x), (uint32_t)s);
          stack_clear(st);
          F3 o, d;
          make_primary(P, sc.cam, x, y, si, sj, rng, o, d);
          ray_set(ray, o, d);
          ct.add(kRaysPrimary);
          T = f3(1, 1, 1);
          L = f3(0, 0, 0);
          depth = MAXD;
          first_ray = (s == 0);
          if (SUB == 1) {
            ++s;
            if (++sj == SPP) { sj = 0; ++si; }
          }
          alive = true;
          in_sample = true;
        }
      }
      // ---- one bounce: the body of Radiance ----
      PT_REGION(2)
      F3 Pn;
      Geom g;
      const int obj = closest_hit<ACCEL, PT_STACK, !LDS, true>(sc, st, ray, Pn, g, ct);
      PT_REGION(3)
      if (first_ray) {
        first_hit = obj;
        if (SUB == 4) shared.first_hit[px] = obj;
        first_ray = false;
      }
      if (obj < 0 || depth == 0) {  // main.cpp:350-355: the background acts as an environment light
        L = L + T * miss_color(P.sc, P.skybox != 0, ray.d);
        alive = false;
        continue;
      }
      if (P.debug_view == P3D_DEBUG_TEST_INTERSECT) {  // main.cpp:359
        L = L + T * f3(1, 0, 0);
        alive = false;
        continue;
      }
      ct.add(kShadedHits);
      const uint32_t m = geom_material(g);
      const float4 m0 = sc.mats[4 * m], m1 = sc.mats[4 * m + 1], m2 = sc.mats[4 * m + 2], m3 = sc.mats[4 * m + 3];
      const F3 E = xyz(m3);
      const F3 norm = get_normal(g, sc.normals, Pn);                         // main.cpp:366
      const F3 norml = (dot(norm, ray.d) < 0) ? norm : norm * -1.0f;         // main.cpp:368
      const F3 intercept_out = offset_intersection(Pn, norm);
      const F3 intercept_in = offset_intersection(Pn, norm * -1.0f);
      F3 f = xyz(m0);
      const float p = max3_ref(f.x, f.y, f.z);
      if (--depth <= MAXD - 5) {  // Russian roulette, main.cpp:382-388
        if (rng.rand_float() < p) {
          f = f * (1 / p);
        } else {
          L = L + T * E;
          alive = false;
          continue;
        }
      }
      if (m0.w == 1.0f) {  // ideal diffuse, main.cpp:391-480
        PT_REGION(4)
        const float r1 = 2 * kPIf * rng.rand_float();
        const float r2 = rng.rand_float();
        const float r2s = sqrtf(r2);
        const F3 w = norml;
        const F3 u = normalized(cross(gt_0p1(fabsf(w.x)) ? f3(0, 1, 0) : f3(1, 0, 0), w));
        const F3 v = cross(w, u);
        double s1, c1;
        det_sincos((double)r1, s1, c1);
        const F3 d = normalized((u * (float)c1 * r2s + v * (float)s1 * r2s) + w * sqrtf(1 - r2));
        F3 e = f3(0, 0, 0);
        for (uint32_t k = 0; k < sc.n_emitters; ++k) {  // explicit light sampling, main.cpp:407-477
          const uint32_t lobj = sc.emitters[k];
          const Geom lg = load_geom(sc.ogeom, lobj);
          const F3 center = f3(lg.a.x, lg.a.y, lg.a.z);
          const float rad = lg.a.w;
          const F3 sw = center - intercept_out;
          const F3 su = normalized(cross(gt_0p1(fabsf(sw.x)) ? f3(0, 1, 0) : f3(1, 0, 0), sw));
          const F3 sv = cross(sw, su);
          const F3 ic = intercept_out - center;
          const double cos_a_max = sqrt(1 - ((double)rad * (double)rad) / (double)dot(ic, ic));
          const double eps1 = rng.erand48();
          const double eps2 = rng.erand48();
          const double cos_a = 1 - eps1 + eps1 * cos_a_max;
          const double sin_a = sqrt(1 - cos_a * cos_a);
          const double phi = (double)(2 * kPIf) * eps2;
          double sphi, cphi;
          det_sincos(phi, sphi, cphi);
          const F3 l = normalized((su * (float)cphi * (float)sin_a + sv * (float)sphi * (float)sin_a) + sw * (float)cos_a);
          RayS feeler;
          ray_set(feeler, intercept_out, l);
          ct.add(kRaysLight);
          // what main.cpp:472-475 multiplies when the sample is visible, worked out before the traversal: two floats live
          // through it instead of l, norml and a double (same operands, same operations: same bits)
          const float omega_f = (float)((double)(2 * kPIf) * (1 - cos_a_max));
          const float l_dot_n = dot(l, norml);
          F3 hp2;
          Geom g2;
          PT_REGION(5)
          const int hit2 = closest_hit<ACCEL, PT_STACK, !LDS, true>(sc, st, feeler, hp2, g2, ct);
          PT_REGION(6)
          if (hit2 >= 0 && hit2 == (int)lobj) {  // main.cpp:472-475
            const F3 emi = xyz(sc.mats[4 * geom_material(load_geom(sc.ogeom, lobj)) + 3]);
            e = e + f * (emi * l_dot_n * omega_f) * (1 / kPIf);
          }
        }
        L = L + T * (E + e);
        T = T * f;
        ray_set(ray, intercept_out, d);
        ct.add(kRaysBounce);
        continue;
      }
      if (m1.w == 1.0f) {  // mirror, main.cpp:481-484
        PT_REGION(7)
        L = L + T * E;
        T = T * f;
        ray_set(ray, intercept_out, ray.d - norm * (2 * dot(norm, ray.d)));
        ct.add(kRaysBounce);
        continue;
      }
      // dielectric, main.cpp:486-515
      PT_REGION(8)
      const F3 refl_d = ray.d - norm * 2 * dot(norm, ray.d);
      const bool into = dot(norm, norml) > 0;
      const double nc = 1.0, nt = (double)m2.z;
      const double nnt = into ? nc / nt : nt / nc;
      const double ddn = (double)dot(ray.d, norml);
      const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
      L = L + T * E;
      T = T * f;
      if (cos2t < 0) {  // total internal reflection
        ray_set(ray, intercept_out, refl_d);
        ct.add(kRaysBounce);
        continue;
      }
      const F3 tdir = normalized(ray.d * (float)nnt - norm * (float)((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
      const double a = nt - nc, b = nt + nc;
      const double R0 = (a * a) / (b * b);
      const double cc = 1 - (into ? -ddn : (double)dot(tdir, norm));
      const double Re = R0 + (1 - R0) * cc * cc * cc * cc * cc;
      const double Tr = 1 - Re;
      const double Pp = 0.25 + 0.5 * Re;
      const double RP = Re / Pp, TP = Tr / (1 - Pp);
      if (depth <= MAXD - 2) {  // main.cpp:509-511: choose one
        if (rng.erand48() < Pp) {
          T = T * (float)RP;
          ray_set(ray, intercept_out, refl_d);
        } else {
          T = T * (float)TP;
          ray_set(ray, intercept_out, tdir);
        }
        ct.add(kRaysBounce);
      } else {  // first two bounces trace both; g++ evaluates the transmission operand first
        {  // at most two levels fork (depth > MAX_DEPTH-2), so two pending entries suffice
          const F3 Tr_ = T * (float)Re;
          pend.base[(size_t)(pend.n * 3 + 0) * pend.stride] = make_float4(intercept_out.x, intercept_out.y, intercept_out.z, refl_d.x);
          pend.base[(size_t)(pend.n * 3 + 1) * pend.stride] = make_float4(refl_d.y, refl_d.z, Tr_.x, Tr_.y);
          pend.base[(size_t)(pend.n * 3 + 2) * pend.stride] = make_float4(Tr_.z, __int_as_float(depth), 0, 0);
          ++pend.n;
        }
        T = T * (float)Tr;
        ray_set(ray, intercept_in, tdir);
        ct.add(kRaysBounce, 2);
      }
    }
    if (SUB == 4 && sub == 0) color = f3(shared.colour[0][px], shared.colour[1][px], shared.colour[2][px]);
    if (P.antialiasing) color = color / (float)(SPP * SPP);  // main.cpp:800

#ifdef P3D_PT_PROFILE
    PT_REGION(9)
    prof.flush();
#endif
    if (sub == 0) {  // SUB == 4: lane 0 of the pixel holds its colour
      if (SUB == 4) first_hit = shared.first_hit[px];
      const size_t k = (size_t)r * P.w + c;
      if (P.rgb) {
        P.rgb[3 * k] = color.x; P.rgb[3 * k + 1] = color.y; P.rgb[3 * k + 2] = color.z;
      }
      if (P.hit_id) P.hit_id[k] = first_hit;
      if (P.rgb8) {
        F3 gc = color;
        if (P.gamma != 1.0f) {
          const double ig = (double)(1 / P.gamma);
          gc = f3((float)pow_spec((double)color.x, ig), (float)pow_spec((double)color.y, ig), (float)pow_spec((double)color.z, ig));
        }
        P.rgb8[3 * k] = u8fromfloat(gc.x); P.rgb8[3 * k + 1] = u8fromfloat(gc.y); P.rgb8[3 * k + 2] = u8fromfloat(gc.z);
      }
    }
  }
  if (STATS) flush_stats<STATS>(ct, P.stats);
  record_tile_cost(P, tx, ty, t_begin);
  P3D_TL_END()
}

}  // namespace p3d

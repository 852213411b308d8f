#!/usr/bin/env python3
"""build/mkexp.py NAME edit[,edit...] -> build/exp_NAME (patched copy of csrc/) ; edits are named below"""
import sys, os, shutil, re
name, edits = sys.argv[1], sys.argv[2].split(',')
ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
src = ROOT + '/p3d-raytracer_amd/csrc'; dst = ROOT + '/build/exp_' + name
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst, ignore=shutil.ignore_patterns('*.o'))
def sub(fn, old, new, count=1):
    p = os.path.join(dst, fn); s = open(p).read()
    assert s.count(old) >= 1, (fn, old[:60])
    s = s.replace(old, new) if count == 0 else s.replace(old, new, count)
    open(p, 'w').write(s)
for e in edits:
    if e == 'early':
        sub('kernels.hpp', 'constexpr bool EARLY_HIT = !AA;', 'constexpr bool EARLY_HIT = !AA && !LDS;')
    elif e == 'inside':
        sub('device_core.hpp', '''      if (fin) {\\
        if (l_t0 < 0 && l_t1 > 0) l_t = 0;\\
        if (r_t0 < 0 && r_t1 > 0) r_t = 0;\\
      } else {\\
        if (is_inside(xyz(l.lo), xyz(l.hi), ray.o)) l_t = 0;\\
        if (is_inside(xyz(r.lo), xyz(r.hi), ray.o)) r_t = 0;\\
      }\\
''', '''      if (is_inside(xyz(l.lo), xyz(l.hi), ray.o)) l_t = 0;\\
      if (is_inside(xyz(r.lo), xyz(r.hi), ray.o)) r_t = 0;\\
''')
    elif e == 'odd':
        sub('device_core.hpp', 'return !(ax < INFINITY && ax >= 0.75f) || !(ay < INFINITY && ay >= 0.75f) || !(az < INFINITY && az >= 0.75f);',
            'return !(ax < INFINITY) || !(ay < INFINITY) || !(az < INFINITY);')
        sub('device_core.hpp', '(!sc.odd_boxes && !__any(ray.odd_inv))', '(!__any(ray.odd_inv))', 0)
    elif e == 'nopad':
        sub('p3d_capi.hip', '''  blob.push_back(make_float4(0, 0, 0, 0));
  blob.push_back(make_float4(0, 0, 0, 0));
  s->off_nodes''', '  s->off_nodes')
    elif e == 'oddinv':
        sub('device_core.hpp', 'return !(ax < INFINITY && ax >= 0.75f) || !(ay < INFINITY && ay >= 0.75f) || !(az < INFINITY && az >= 0.75f);',
            'return !(ax < INFINITY) || !(ay < INFINITY) || !(az < INFINITY);')
    elif e == 'stack8':
        sub('device_core.hpp', '''    lds_u16* d = s.dbase + s.sp * kBlock;
    *d = (uint16_t)pack_desc16(node);
    *stack_t_of(s, d) = __float_as_uint(t);
''', '''    *(s.base + s.sp * kBlock) = (unsigned long long)node | ((unsigned long long)__float_as_uint(t) << 32);
''')
        sub('device_core.hpp', '''    const lds_u16* d = s.dbase + i * kBlock;
    e = make_uint2(unpack_desc16(*d), *stack_t_of(s, d));
''', '''    e = unpack_entry(*(s.base + i * kBlock));
''')
        sub('device_core.hpp', 'return cap * kBlock * (spill ? 8u : 6u) / 16u;', 'return cap * kBlock * 8u / 16u;')
    elif e == 'pad8k':
        sub('p3d_capi.hip', '(sub4 ? sizeof(PtPixelShared) : 0) + (cold_lds ? (size_t)kColdDwords * kBlock * sizeof(float) : 0);', '(sub4 ? sizeof(PtPixelShared) : 0) + (cold_lds ? (size_t)kColdDwords * kBlock * sizeof(float) : 0) + (lds_scene ? 576 : 0);')
    elif e.startswith('wide'):
        sub('p3d_capi.hip', 'std::max<uint32_t>(64, H.n_units / kBlock)));', 'std::max<uint32_t>(64, std::min<uint32_t>(H.n_units / kBlock, %su))));' % e[4:])
    elif e.startswith('nv'):
        sub('kernels.hpp', '__global__ void __launch_bounds__(kBlock, LIT == 2 ? P3D_LIST_WAVES : ((LDS || AA) ? P3D_WHITTED_WAVES : P3D_WHITTED_GLOBAL_WAVES)) whitted_kernel(',
            '__global__ void __launch_bounds__(kBlock, LIT == 2 ? P3D_LIST_WAVES : ((LDS || AA) ? P3D_WHITTED_WAVES : P3D_WHITTED_GLOBAL_WAVES)) __attribute__((amdgpu_num_vgpr((LDS && LIT == 1 && !AA && !STATS) ? %s : 0))) whitted_kernel(' % e[2:])
    elif e == 'lit1w5':
        sub('kernels.hpp', '__launch_bounds__(kBlock, LIT == 2 ? P3D_LIST_WAVES : ((LDS || AA) ? P3D_WHITTED_WAVES : P3D_WHITTED_GLOBAL_WAVES)) whitted_kernel(',
            '__launch_bounds__(kBlock, LIT == 2 ? P3D_LIST_WAVES : ((LDS && LIT == 1 && !AA) ? 5 : ((LDS || AA) ? P3D_WHITTED_WAVES : P3D_WHITTED_GLOBAL_WAVES))) whitted_kernel(')
    elif e == 'fastall':
        sub('device_core.hpp', 'constexpr bool FASTIN = VOTE;', 'constexpr bool FASTIN = true;', 0)
    elif e == 'ptnofast':
        sub('pt_kernel.hpp', 'closest_hit<ACCEL, PT_STACK, !LDS, true>', 'closest_hit<ACCEL, PT_STACK, !LDS, !LDS>', 0)
    elif e == 'pt6':
        sub('pt_kernel.hpp', 'constexpr int PT_STACK = LDS ? kStackLds8 : kStackWindow;', 'constexpr int PT_STACK = LDS ? kStackLds6 : kStackWindow;')
        sub('p3d_capi.hip', '(pt ? kStackLds8 : kStackLds6)', 'kStackLds6')
    elif e == 'ldirlit':
        sub('whitted_level.inc', 'if (COLD) {  // the light direction', 'if (COLD || (LIT == 1)) {  // the light direction')
    elif e == 'ldirall':
        sub('whitted_level.inc', 'if (COLD) {  // the light direction', 'if (true) {  // the light direction')
    elif e == 'cap12':
        sub('p3d_capi.hip', 'const uint32_t cap = cfg->accel == P3D_ACCEL_BVH ? (spilling ? window : bound) : 1;', 'const uint32_t cap = cfg->accel == P3D_ACCEL_BVH ? (spilling ? window : std::min<uint32_t>(bound, 12)) : 1;')
    else:
        raise SystemExit('unknown edit ' + e)
print(dst)

// one instantiation of the path tracer only: fast register-pressure experiments (one_pt_regs.sh)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "p3d.h"
#include "kernels.hpp"
#include "pt_kernel.hpp"
#ifndef ONE_SUB
#define ONE_SUB 4
#endif
namespace p3d {
template __global__ void pt_kernel<P3D_ACCEL_BVH, true, false, ONE_SUB>(const RenderParams);
}

// one instantiation only: fast register-pressure experiments
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "p3d.h"
#include "kernels.hpp"
#ifndef ONE_GHOSTS
#define ONE_GHOSTS true
#endif
namespace p3d {
template __global__ void whitted_kernel<P3D_ACCEL_BVH, true, false, false, false, 1, ONE_LIT, ONE_GHOSTS>(const RenderParams);
}

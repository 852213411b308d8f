// Accuracy check of pow_spec (device_core.hpp) against the host's libm pow, the function the
// reference calls in main.cpp:224.  Built and run by tests/test_gpu_parity.py on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "device_core.hpp"

__global__ void k(const float* x, const float* y, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = p3d::pow_spec((double)x[i], (double)y[i]);
}

int main() {
  const int n = 1 << 21;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> ux(0.0f, 1.0000002f), uy(0.25f, 400.0f);
  const float shines[] = {10.f, 30.0827f, 100.f, 300.f, 101.148f, 20.f, 1.f, 2.f, 0.5f, 60.f, 40.f, 80.f};
  std::vector<float> hx(n), hy(n);
  for (int i = 0; i < n; ++i) {
    hx[i] = (i % 997 == 0) ? 0.0f : ((i % 991 == 0) ? 1.0f : ux(rng));
    hy[i] = (i & 1) ? shines[(i >> 1) % 12] : uy(rng);
  }
  float *dx, *dy;
  double* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, n * 8);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, dout, n);
  std::vector<double> out(n);
  if (hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("hip error\n"); return 2; }
  double worst = 0;
  long mism = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = std::pow((double)hx[i], (double)hy[i]);
    if (ref > 1e-300) worst = std::fmax(worst, std::fabs(out[i] - ref) / ref);
    // below 1e-300 only the float rounding (0) is observable; counted by the mismatch test
    if ((float)out[i] != (float)ref) ++mism;
  }
  std::printf("n=%d worst_rel=%.3e float_mismatches=%ld\n", n, worst, mism);
  return 0;
}

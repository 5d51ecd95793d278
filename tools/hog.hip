// dev: a co-tenant for tools/corun_probe.py -- `nwg` workgroups that hold a CU each (the whole LDS) for `iters` steps of
// (a) a dependent VALU chain, (b) the same with a workgroup barrier + LDS exchange per step (the shape of an FPS round),
// (c) (b) + one 64-byte global load per wave and step.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hog.hip -o tools/_build/libhog.so
#include <hip/hip_runtime.h>

template <int MODE>
__global__ __launch_bounds__(1024) void hog_kernel(float *out, const float *src, int iters) {
  extern __shared__ float lds[];
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) v = v * 1.0001f + 0.5f;
    if (MODE >= 1) {
      if ((threadIdx.x & 63) == 0) lds[(i & 1) * 16 + (threadIdx.x >> 6)] = v;
      __syncthreads();
      v += lds[(i & 1) * 16 + (threadIdx.x & 15)];
    }
    if (MODE >= 2) v += src[((i * 1024 + threadIdx.x) & 0xfffff)];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

extern "C" int hog_launch(int mode, int nwg, int threads, int iters, int lds_bytes, float *out, const float *src, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto k = mode == 0 ? hog_kernel<0> : (mode == 1 ? hog_kernel<1> : hog_kernel<2>);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), lds_bytes, s, out, src, iters);
  return static_cast<int>(hipGetLastError());
}

// where the workgroups of a grid run: out[wg] = XCC_ID | HW_ID << 8 (s_getreg), each workgroup busy for ~`iters` steps so that
// the grid's workgroups coexist like a real kernel's
__global__ __launch_bounds__(256) void where_kernel(unsigned *out, float *sink, int iters) {
  extern __shared__ float lds[];
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) v = v * 1.0001f + 0.5f;
  }
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) | (hw << 8);
  if (v == 12345.f) sink[0] = v + lds[0];
}
extern "C" int where_launch(int nwg, int iters, int lds_bytes, unsigned *out, float *sink, void *stream) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(where_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), out, sink, iters);
  return static_cast<int>(hipGetLastError());
}

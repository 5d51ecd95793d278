// Dev probe: shader clock actually delivered to a lightly loaded GPU (1 / 8 / 256 blocks), and
// the latency of the building blocks of an FPS round.  hipcc --offload-arch=gfx950 -O3 tools/clk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void spin(float *out, long long *clk, int iters) {
  float v = threadIdx.x * 1e-3f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;  // dependent chain: 1 VALU op (fma) per iter
  const long long t1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

__global__ void round_parts(float *out, long long *clk, int iters) {
  __shared__ unsigned long long slot[3];
  if (threadIdx.x < 3) slot[threadIdx.x] = 0;
  __syncthreads();
  float v = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) __syncthreads();
  long long t1 = clock64();
  for (int i = 0; i < iters; ++i) {
    if ((threadIdx.x & 63) == 0) atomicMax(&slot[i % 3], (unsigned long long)(i + threadIdx.x));
    __syncthreads();
    v += (float)(slot[i % 3] & 1);
  }
  long long t2 = clock64();
  for (int i = 0; i < iters; ++i) {
    float m = v;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    v += m * 1e-9f;
  }
  long long t3 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; clk[2] = t3 - t2; }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void kmax_step(unsigned &hi, unsigned &lo) {
  const unsigned ohi = __builtin_amdgcn_update_dpp(0u, hi, CTRL, ROW_MASK, 0xf, false);
  const unsigned olo = __builtin_amdgcn_update_dpp(0u, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned long long mine = ((unsigned long long)hi << 32) | lo, other = ((unsigned long long)ohi << 32) | olo;
  if (other > mine) { hi = ohi; lo = olo; }
}
__device__ __forceinline__ float wave_max_f32(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__global__ void round_parts2(float *out, long long *clk, int iters) {
  __shared__ uint2 cand[2][8];
  __shared__ unsigned long long slot[3];
  if (threadIdx.x < 3) slot[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float v = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {  // plain write + barrier + read + 3 DPP steps
    if (lane == 0) cand[i & 1][w] = make_uint2(__float_as_uint(v) + i, (unsigned)threadIdx.x);
    __syncthreads();
    uint2 kk = cand[i & 1][lane & 7];
    unsigned hi = kk.x, lo = kk.y;
    kmax_step<0x111, 0xf>(hi, lo);
    kmax_step<0x112, 0xf>(hi, lo);
    kmax_step<0x114, 0xf>(hi, lo);
    hi = __builtin_amdgcn_readlane(hi, 7);
    lo = __builtin_amdgcn_readlane(lo, 7);
    v += (float)((hi ^ lo) & 1);
  }
  long long t1 = clock64();
  for (int i = 0; i < iters; ++i) {  // DPP wave max asm
    v += wave_max_f32(v) * 1e-9f;
  }
  long long t2 = clock64();
  for (int i = 0; i < iters; ++i) {  // no-return atomic (result unused), 32-bit
    if (lane == 0) atomicMax((unsigned *)&slot[i % 3], (unsigned)(i + threadIdx.x));
    __syncthreads();
    v += (float)(((unsigned *)&slot[i % 3])[0] & 1);
  }
  long long t3 = clock64();
  for (int i = 0; i < iters; ++i) {  // scalar global load of 3 dwords at a data-dependent uniform address
    const int a = __builtin_amdgcn_readfirstlane(((int)v) & 1023);
    v += out[a * 3] + out[a * 3 + 1] + out[a * 3 + 2];
  }
  long long t4 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x + 4096] = v;
  if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; clk[2] = t3 - t2; clk[3] = t4 - t3; }
}

int main() {
  float *out; long long *clk;
  hipMalloc(&out, 2048 * 512 * 4); hipMalloc(&clk, 4096 * 8);
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, max shader clock %d kHz\n", wall_khz, clk_khz);
  const int iters = 2000000;
  for (int blocks : {1, 8, 256, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, 0, out, clk, iters);
      hipDeviceSynchronize();
    }
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double secs = (double)h[1] / (wall_khz * 1e3);
    printf("blocks %4d: clock64 delta %lld, wall %.3f ms -> counter rate %.1f MHz; %.2f ns per dependent fma -> %.2f counter ticks/op\n",
           blocks, h[0], secs * 1e3, h[0] / secs / 1e6, secs * 1e9 / iters, (double)h[0] / iters);
  }
  hipLaunchKernelGGL(round_parts, dim3(8), dim3(512), 0, 0, out, clk, 20000);
  hipDeviceSynchronize();
  long long h[3]; hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
  printf("per iteration (clock64 ticks): barrier %.1f, atomicMax+barrier+read %.1f, shfl-max(6 steps) %.1f\n", h[0] / 2e4, h[1] / 2e4, h[2] / 2e4);
  hipLaunchKernelGGL(round_parts2, dim3(8), dim3(512), 0, 0, out, clk, 20000);
  hipDeviceSynchronize();
  long long g[4]; hipMemcpy(g, clk, 32, hipMemcpyDeviceToHost);
  printf("per iteration (ticks): write+barrier+read+3 DPP(64-bit) %.1f, DPP wave max %.1f, atomicMax u32+barrier+read %.1f, dependent scalar/vector global load x3 %.1f\n",
         g[0] / 2e4, g[1] / 2e4, g[2] / 2e4, g[3] / 2e4);
  return 0;
}

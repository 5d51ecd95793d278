// Dev probe: do VALU instructions execute under a chain of fp32 MFMAs on one SIMD?  One workgroup of 8 waves per
// CU (2 waves per SIMD), kernel durations from HIP events:
//   (a) every wave: chain of dependent v_mfma_f32_32x32x2_f32      (b) every wave: 16 independent VALU FMAs per step
//   (c) every wave: one MFMA + 16 FMAs per step, interleaved        (d) waves 0-3 do (a), waves 4-7 do (b)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o tools/_build/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, const float *in, int steps) {
  const int w = threadIdx.x >> 6;
  f32x16 acc = {}, acc1 = {}, acc2 = {}, acc3 = {};
  const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && w < 4) || MODE >= 4;
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && w >= 4) || MODE == 4 || MODE == 6;
  constexpr bool kFour = MODE >= 4 && MODE <= 5;  // four independent accumulators
  constexpr bool kTwo = MODE >= 6;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (do_mfma) {
        if (kFour) {
          if (it % 4 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          if (it % 4 == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
          if (it % 4 == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
          if (it % 4 == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        } else if (kTwo) {
          if (it % 2 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          if (it % 2 == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
      if (do_valu) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * b + a;
      }
      if (MODE == 2 || MODE == 4 || MODE == 6) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc1[i] + acc2[i] + acc3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&in, 128 * sizeof(float));
  (void)hipMemset(in, 0, 128 * sizeof(float));
  const int steps = 4000;  // x 16 MFMAs / 256 FMAs per wave
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const char *names[7] = {"(a) MFMA chain in every wave", "(b) 16 FMAs per step in every wave", "(c) MFMA + 16 FMAs interleaved in every wave",
                          "(d) MFMA waves paired with VALU waves on each SIMD", "(e) as (c), four independent accumulators",
                          "(f) as (a), four independent accumulators", "(g) as (c), two independent accumulators"};
  for (int mode = 0; mode < 7; ++mode) {
    float ms = 0.f;
    for (int r = 0; r < 2; ++r) {
      (void)hipEventRecord(e0, 0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        case 5: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
        default: hipLaunchKernelGGL(probe<6>, dim3(256), dim3(512), 0, 0, out, in, steps); break;
      }
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 2 waves x steps x 16 MFMAs (64 cycles each) resp. 2 x steps x 256 FMAs (4 cycles each)
    printf("%-52s %8.3f ms  = %6.1f ns per step and wave pair\n", names[mode], ms, ms * 1e6 / steps);
  }
  printf("one step = 16 MFMAs (1024 cycles) and/or 256 FMAs (1024 cycles) per wave; 2 waves per SIMD\n");
  return 0;
}

// dev probe: issue rate of v_mfma_f32_32x32x2_f32 with operand fragments arriving from LDS in different ways.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_probe.hip -o tools/_build/mfma_lds_probe && tools/_build/mfma_lds_probe
// Every variant: 256 workgroups x 256 threads (one wave per SIMD), ITER groups of 8 MFMAs per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
constexpr int ITER = 512;

template <int V>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc) {
  __shared__ float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 16384; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a0 = 1.0f + lane, a1 = 2.0f, b0 = 0.5f, b1 = 0.25f, b2 = 0.125f, b3 = 0.0625f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (V == 0) {  // registers only
    for (int it = 0; it < ITER; ++it) {
      acc[0] = MFMA(a0, b0, acc[0]); acc[1] = MFMA(a0, b1, acc[1]); acc[2] = MFMA(a0, b2, acc[2]); acc[3] = MFMA(a0, b3, acc[3]);
      acc[4] = MFMA(a1, b0, acc[4]); acc[5] = MFMA(a1, b1, acc[5]); acc[6] = MFMA(a1, b2, acc[6]); acc[7] = MFMA(a1, b3, acc[7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (V == 1 || V == 2) {  // six b32 fragments per group, double buffered (V == 2: natural order, no buffering)
    float af[2][2], bf[2][4];
    auto frag = [&](int kk, int s) {
      const float *pa = lds + ((2 * kk + h) & 63) * 256 + l31;
      const float *pb = lds + 8192 + ((2 * kk + h) & 63) * 128 + l31;
      af[s][0] = pa[0]; af[s][1] = pa[32];
      bf[s][0] = pb[0]; bf[s][1] = pb[32]; bf[s][2] = pb[64]; bf[s][3] = pb[96];
    };
    frag(0, 0);
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
      const int s = it & 1;
      if (V == 1) frag(it + 1, s ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = MFMA(af[s][0], bf[s][0], acc[0]); acc[1] = MFMA(af[s][0], bf[s][1], acc[1]);
      acc[2] = MFMA(af[s][0], bf[s][2], acc[2]); acc[3] = MFMA(af[s][0], bf[s][3], acc[3]);
      acc[4] = MFMA(af[s][1], bf[s][0], acc[4]); acc[5] = MFMA(af[s][1], bf[s][1], acc[5]);
      acc[6] = MFMA(af[s][1], bf[s][2], acc[6]); acc[7] = MFMA(af[s][1], bf[s][3], acc[7]);
      __builtin_amdgcn_sched_barrier(0);
      if (V == 2) frag(it + 1, s ^ 1);
    }
  } else if (V == 3) {  // one b64 + one b128 per group (permuted layouts), double buffered
    f32x2 af[2];
    f32x4 bf[2];
    auto frag = [&](int kk, int s) {
      af[s] = *reinterpret_cast<const f32x2 *>(lds + ((2 * kk + h) & 63) * 128 + 2 * l31);
      bf[s] = *reinterpret_cast<const f32x4 *>(lds + 8192 + ((2 * kk + h) & 63) * 128 + 4 * l31);
    };
    frag(0, 0);
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
      const int s = it & 1;
      frag(it + 1, s ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = MFMA(af[s][0], bf[s][0], acc[0]); acc[1] = MFMA(af[s][0], bf[s][1], acc[1]);
      acc[2] = MFMA(af[s][0], bf[s][2], acc[2]); acc[3] = MFMA(af[s][0], bf[s][3], acc[3]);
      acc[4] = MFMA(af[s][1], bf[s][0], acc[4]); acc[5] = MFMA(af[s][1], bf[s][1], acc[5]);
      acc[6] = MFMA(af[s][1], bf[s][2], acc[6]); acc[7] = MFMA(af[s][1], bf[s][3], acc[7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (V == 4 || V == 5) {  // dx-like: two accumulators (V == 5: four), A from two b128 per 8 MFMAs, B in registers
    f32x4 av[2][2];
    auto frag = [&](int kq, int s) {
      av[s][0] = *reinterpret_cast<const f32x4 *>(lds + l31 * 132 + h * 64 + 4 * (kq & 15));
      av[s][1] = *reinterpret_cast<const f32x4 *>(lds + 4224 + l31 * 132 + h * 64 + 4 * (kq & 15));
    };
    frag(0, 0);
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
      const int s = it & 1;
      frag(it + 1, s ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (V == 4) {
        acc[0] = MFMA(av[s][0][0], b0, acc[0]); acc[1] = MFMA(av[s][1][0], b0, acc[1]);
        acc[0] = MFMA(av[s][0][1], b1, acc[0]); acc[1] = MFMA(av[s][1][1], b1, acc[1]);
        acc[0] = MFMA(av[s][0][2], b2, acc[0]); acc[1] = MFMA(av[s][1][2], b2, acc[1]);
        acc[0] = MFMA(av[s][0][3], b3, acc[0]); acc[1] = MFMA(av[s][1][3], b3, acc[1]);
      } else {
        acc[0] = MFMA(av[s][0][0], b0, acc[0]); acc[1] = MFMA(av[s][1][0], b0, acc[1]);
        acc[2] = MFMA(av[s][0][1], b1, acc[2]); acc[3] = MFMA(av[s][1][1], b1, acc[3]);
        acc[0] = MFMA(av[s][0][2], b2, acc[0]); acc[1] = MFMA(av[s][1][2], b2, acc[1]);
        acc[2] = MFMA(av[s][0][3], b3, acc[2]); acc[3] = MFMA(av[s][1][3], b3, acc[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int V>
void run(const char *name) {
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, out, cyc);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, out, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += v;
  const double per = sum / 1024 / (ITER * 8.0);
  const double us = ms * 1e3 / 10;
  printf("%-58s %6.1f clocks/MFMA  kernel %7.1f us  -> %6.1f ns/MFMA/SIMD = %5.1f TF/s, implied clock %.2f GHz\n", name, per, us,
         us * 1e3 / (ITER * 8.0), 256.0 * 4 * ITER * 8 * 4096 / (us * 1e-6) / 1e12, per / (us * 1e3 / (ITER * 8.0)));
}

int main() {
  run<0>("registers only, 8 accumulators");
  run<1>("6 x ds_read_b32 per 8 MFMAs, double buffered");
  run<2>("6 x ds_read_b32 per 8 MFMAs, fetched after use");
  run<3>("b64 + b128 per 8 MFMAs, double buffered");
  run<4>("2 x b128 per 8 MFMAs, 2 accumulators");
  run<5>("2 x b128 per 8 MFMAs, 4 accumulators");
  return 0;
}

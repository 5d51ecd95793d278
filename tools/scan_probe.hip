// dev probe: cost of the per-channel running max + arg-max chain of the pooled set-abstraction layer (64 rows, one wave per SIMD)
//   hipcc --offload-arch=gfx950 -O3 tools/scan_probe.hip -o tools/_build/scan_probe && tools/_build/scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int V>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, int *outi, unsigned long long *cyc, int base_in, float sgin,
                                             unsigned long long chg) {
  __shared__ float lds[64 * 260];
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 260; i += 256) lds[i] = in[i % 4096] + i;
  __syncthreads();
  float vall[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) vall[r] = lds[r * 260 + tid];
  float best = -INFINITY;
  int arg = 0;
  const float sg = sgin;
  int base = __builtin_amdgcn_readfirstlane(base_in);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  if (V == 0) {  // as in the kernel: scalar row-in-group, multiply by the sign
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const float v = vall[r] * sg;
      if (v > best) { best = v; arg = base + r; }
    }
  } else if (V == 1) {  // tile-row constants
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const float v = vall[r] * sg;
      if (v > best) { best = v; arg = r; }
    }
    arg += base;
  } else if (V == 2) {  // no multiply, constants
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const float v = vall[r];
      if (v > best) { best = v; arg = r; }
    }
    arg += base;
  } else if (V == 3) {  // quads behind a scalar test (all taken: no group start)
#pragma unroll
    for (int rq = 0; rq < 64; rq += 4) {
      if (__builtin_expect(((chg >> rq) & 0xfull) == 0ull, 1)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float v = vall[rq + u] * sg;
          if (v > best) { best = v; arg = base + rq + u; }
        }
      } else {
        best = -INFINITY; arg = 0; base = 7 - rq;
      }
    }
  } else if (V == 4) {  // max only, arg recovered afterwards by equality (two passes)
#pragma unroll
    for (int r = 0; r < 64; ++r) best = fmaxf(best, vall[r] * sg);
#pragma unroll
    for (int r = 63; r >= 0; --r) arg = (vall[r] * sg == best) ? r : arg;
    arg += base;
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + tid] = best;
  outi[blockIdx.x * 256 + tid] = arg;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int V>
void run(const char *name, const float *in, float *out, int *outi, unsigned long long *cyc) {
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, in, out, outi, cyc, 5, 1.0f, 0ull);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += v;
  printf("%-62s %7.0f clocks per 64 rows = %5.1f per row\n", name, s / 1024, s / 1024 / 64);
}

int main() {
  float *in, *out;
  int *outi;
  unsigned long long *cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 65536 * 4); hipMalloc(&outi, 65536 * 4); hipMalloc(&cyc, 1024 * 8);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = sinf(i * 0.37f);
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0>("v * sg; if (v > best) { best = v; arg = base + r; }", in, out, outi, cyc);
  run<1>("... arg = r (tile-row constants), base added once", in, out, outi, cyc);
  run<2>("... and no multiply", in, out, outi, cyc);
  run<3>("quads behind a scalar test, as in the kernel", in, out, outi, cyc);
  run<4>("v_max chain, then arg by equality (two passes)", in, out, outi, cyc);
  return 0;
}

// Micro-benchmark: the practical ceiling of the walk kernel's TRAFFIC MIX on MI355X - a kernel
// that moves exactly the headline's bytes in the walk's shape and does nothing else: per unit
// (series n, group g of 3) read the series' 3 rows (24 KB) and write 6 output rows of 8 KB to
// out[k][n][:] (k = 6 g .. 6 g + 5), one workgroup per unit or a persistent grid.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_mix.hip -o /tmp/sm && /tmp/sm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double vd2 __attribute__((ext_vector_type(2)));
constexpr int T = 1024, D = 3, K = 18;

template <int G, int READ>   // READ 0: no input, 1: plain loads, 2: non-temporal loads
__global__ __launch_bounds__(256) void mix_kernel(const double *X, double *out, int N, int spread) {
  const int tid = threadIdx.x;
  const int units = N * G;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int q = u >> 3, r = u & 7;
    const int n = (q / G) * 8 + r, g = q % G;
    vd2 v[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int k = 0; k < 2; ++k)
      {
        const vd2 *src = reinterpret_cast<const vd2 *>(X + ((size_t)n * D + d) * T + 2 * (k * 256 + tid));
        if constexpr (READ == 0) v[d][k] = vd2{(double)n, (double)tid};
        else if constexpr (READ == 1) v[d][k] = *src;
        else v[d][k] = __builtin_nontemporal_load(src);
      }
#pragma unroll
    for (int j = 0; j < K / G; ++j) {
      const int k = g * (K / G) + j;
      double *dst = out + ((size_t)k * N + n) * T;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        vd2 w = v[j % D][h];
        w.x += (double)j;
        *reinterpret_cast<vd2 *>(dst + 2 * (h * 256 + tid)) = w;
      }
      for (int s = 0; s < spread; ++s) __builtin_amdgcn_s_sleep(16);   // time between two rows
    }
  }
}

template <int G, int READ>
static void run(const double *X, double *out, int N, int grid, const char *what) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const double bytes = 8.0 * N * T * ((READ ? D : 0) + K);
  const int g = grid ? grid : N * G;
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(a);
    for (int rep = 0; rep < 10; ++rep)
      hipLaunchKernelGGL((mix_kernel<G, READ>), dim3(g), dim3(256), 0, 0, X, out, N, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms / 10);
  }
  std::sort(ts.begin(), ts.end());
  printf("N %5d  G %d  %-18s %-16s: %7.1f us  %.2f TB/s of %4.0f MB\n", N, G, what,
         grid ? "persistent 1536" : "one wg per unit", ts[3] * 1e3, bytes / (ts[3] * 1e-3) / 1e12, bytes / 1e6);
}

int main() {
  for (int N : {2048, 8192}) {
    double *X, *out;
    hipMalloc(&X, (size_t)N * D * T * 8);
    hipMalloc(&out, (size_t)K * N * T * 8);
    hipMemset(X, 0, (size_t)N * D * T * 8);
    for (int grid : {0, 1536}) {
      run<3, 0>(X, out, N, grid, "writes only");
      run<3, 1>(X, out, N, grid, "reads + writes");
      run<3, 2>(X, out, N, grid, "nt reads + writes");
      run<1, 0>(X, out, N, grid, "writes only");
      run<1, 1>(X, out, N, grid, "reads + writes");
      run<1, 2>(X, out, N, grid, "nt reads + writes");
    }
    hipFree(X); hipFree(out);
  }
  return 0;
}

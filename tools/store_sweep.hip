// Micro-benchmark: what limits a streaming WRITE on MI355X - workgroups in flight, bytes per
// visit, persistent or one workgroup per chunk, plain or non-temporal stores.  All variants
// write every byte of the buffer once, consecutive chunks by consecutive workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/store_sweep.hip -o /tmp/ss && /tmp/ss
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double vd2 __attribute__((ext_vector_type(2)));

// CH bytes per workgroup visit (256 threads, 16-byte stores, each wave CH/4 contiguous bytes)
template <int CH, bool NT>
__global__ __launch_bounds__(256) void sweep_kernel(double *out, long long chunks) {
  constexpr int EL = CH / 8, PER = EL / 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long c = blockIdx.x; c < chunks; c += gridDim.x) {
    double *dst = out + c * EL + wave * (EL / 4);
#pragma unroll
    for (int p = 0; p < PER / 2; ++p) {
      vd2 v = {(double)c, (double)p};
      vd2 *q = reinterpret_cast<vd2 *>(dst + p * 128 + lane * 2);
      if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v;
    }
  }
}

template <int CH, bool NT>
static void run(double *out, size_t bytes, int grid) {
  const long long chunks = (long long)(bytes / CH);
  const int g = grid > 0 ? grid : (int)chunks;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(a);
    for (int rep = 0; rep < 5; ++rep)
      hipLaunchKernelGGL((sweep_kernel<CH, NT>), dim3(g), dim3(256), 0, 0, out, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms / 5);
  }
  std::sort(ts.begin(), ts.end());
  printf("%5.0f MB  chunk %5d B  grid %7d%s  %s: %7.1f us  %.2f TB/s\n", bytes / 1e6, CH, g,
         grid > 0 ? " (persistent)" : " (one per chunk)", NT ? "nontemporal" : "plain      ",
         ts[3] * 1e3, bytes / (ts[3] * 1e-3) / 1e12);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  for (size_t bytes : {(size_t)301989888, (size_t)1207959552}) {
    double *out;
    hipMalloc(&out, bytes);
    hipMemset(out, 0, bytes);
    for (int grid : {1024, 1536, 2048, 3072, 4096, 0}) {
      run<4096, false>(out, bytes, grid);
      run<8192, false>(out, bytes, grid);
      run<16384, false>(out, bytes, grid);
      run<8192, true>(out, bytes, grid);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int rep = 0; rep < 5; ++rep) hipMemsetAsync(out, 0, bytes, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%5.0f MB  hipMemsetAsync: %7.1f us  %.2f TB/s\n", bytes / 1e6, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
    hipFree(out);
  }
  return 0;
}

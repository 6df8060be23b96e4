// Micro-benchmark: what the memory system sustains for the walk kernel's STORE PATTERN.
// grid = resident workgroups (256 threads); every workgroup writes `iters` chunks of CH bytes
// (each wave CH/4 contiguous bytes with 16-byte stores), the chunk address given by a pattern:
//   0  fill-like      chunk index = i * grid + b                     (all workgroups sweep together)
//   1  planes (K,N,T) plane k = i % K, series n = b + (i / K) * grid  -> [k][n]
//   2  private (N,K,T) chunk index = b * iters + i                    (one sequential stream per workgroup)
//   3  planes, series of one XCD contiguous: b' = (b % 8) * (grid / 8) + b / 8
//   4  planes, two series per visit: chunk pairs [k][2n], [k][2n+1] written back to back
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double vd2 __attribute__((ext_vector_type(2)));

template <int CH>
__global__ __launch_bounds__(256) void pattern_kernel(double *out, int pattern, int K, int iters,
                                                       long long chunks, int gap) {
  const int b = blockIdx.x, grid = gridDim.x, tid = threadIdx.x;
  constexpr int EL = CH / 8;           // doubles per chunk
  constexpr int PER = EL / 256;        // doubles per thread per chunk (2 -> one 16-byte store)
  const int wave = tid >> 6, lane = tid & 63;
  for (int i = 0; i < iters; ++i) {
    long long c;
    if (pattern == 0) c = (long long)i * grid + b;
    else if (pattern == 1) c = (long long)(i % K) * (chunks / K) + b + (long long)(i / K) * grid;
    else if (pattern == 2) c = (long long)b * iters + i;
    else if (pattern == 3) {
      const int bb = (b % 8) * (grid / 8) + b / 8;
      c = (long long)(i % K) * (chunks / K) + bb + (long long)(i / K) * grid;
    } else {
      const int j = i >> 1;
      c = (long long)(j % K) * (chunks / K) + 2 * (b + (long long)(j / K) * grid) + (i & 1);
    }
    if (c >= chunks) continue;
    double *dst = out + c * EL + wave * (EL / 4);
#pragma unroll
    for (int p = 0; p < PER / 2; ++p) {
      vd2 v = {(double)i, (double)b};
      *reinterpret_cast<vd2 *>(dst + p * 128 + lane * 2) = v;
    }
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
  }
}

int main(int argc, char **argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1536;
  const int K = 18;
  double *out;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (long long series : {2048LL, 8192LL}) {
    const long long chunks = series * K;          // 8 KB chunks
    const size_t bytes = (size_t)chunks * 8192;
    hipMalloc(&out, bytes);
    const int iters = (int)((chunks + grid - 1) / grid);
    for (int gap : {0, 4}) {
      for (int pattern = 0; pattern < 5; ++pattern) {
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) {
          hipEventRecord(a);
          for (int rep = 0; rep < 5; ++rep)
            hipLaunchKernelGGL(pattern_kernel<8192>, dim3(grid), dim3(256), 0, 0, out, pattern, K,
                               pattern == 4 ? 2 * ((iters + 1) / 2) : iters, chunks, gap);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms / 5);
        }
        std::sort(ts.begin(), ts.end());
        printf("series %lld (%.0f MB) grid %d gap %d pattern %d: %.1f us  %.2f TB/s\n", series,
               bytes / 1e6, grid, gap, pattern, ts[3] * 1e3, bytes / (ts[3] * 1e-3) / 1e12);
      }
    }
    hipFree(out);
  }
  return 0;
}

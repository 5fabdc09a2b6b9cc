// Gather-rate microbenchmark for the hash-grid encoder: how many
// divergent 4-/8-byte loads per clock one CU sustains from an L2-resident 2 MB table.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// MODE 0: 8 random dword loads          1: 4 random dwordx2 loads
//      2: 4 random + 4 partner (idx^1)  3: 4 random dwordx2 + 4 dword on odd lanes only
//      4: 4 random dword loads          5: 8 dword loads, 4 lines (2 per line, +16 entries apart)
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const unsigned* __restrict__ table, unsigned mask, int iters,
                                                     unsigned* __restrict__ out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned h = mix(tid * 977u + it * 0x9e3779b9u);
    unsigned idx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = mix(h + j * 0x85ebca6bu) & mask;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc += table[idx[j]];
        acc += table[mix(idx[j] + 77u) & mask];
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 v = *reinterpret_cast<const uint2*>(table + (idx[j] & ~1u));
        acc += v.x + 3u * v.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc += table[idx[j]];
        acc += 3u * table[idx[j] ^ 1u];
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 v = *reinterpret_cast<const uint2*>(table + (idx[j] & ~1u));
        acc += v.x + 3u * v.y;
      }
      if (threadIdx.x & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += 5u * table[mix(idx[j] + 77u) & mask];
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += table[idx[j]];
    } else if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc += table[idx[j] & ~31u];
        acc += 3u * table[(idx[j] & ~31u) + 16u];
      }
    }
  }
  out[tid] = acc;
}

template <int MODE>
float run(const unsigned* table, unsigned mask, int iters, unsigned* out, int blocks) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather_kernel<MODE><<<blocks, 256>>>(table, mask, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) gather_kernel<MODE><<<blocks, 256>>>(table, mask, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / 5.f;
}

int main() {
  const int blocks = 256 * 16, iters = 64;
  unsigned *table, *out;
  const char* names[6] = {"8 x dword random", "4 x dwordx2 random", "4 x (dword + partner idx^1)",
                          "4 x dwordx2 + 4 x dword on odd lanes", "4 x dword random", "4 x (2 dwords in one 128-B line)"};
  for (int log2n : {19, 23}) {
    const unsigned n = 1u << log2n;
    CK(hipMalloc(&table, n * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = i * 2654435761u;
    CK(hipMemcpy(table, h.data(), n * 4, hipMemcpyHostToDevice));
    float ms[6];
    ms[0] = run<0>(table, n - 1, iters, out, blocks);
    ms[1] = run<1>(table, n - 1, iters, out, blocks);
    ms[2] = run<2>(table, n - 1, iters, out, blocks);
    ms[3] = run<3>(table, n - 1, iters, out, blocks);
    ms[4] = run<4>(table, n - 1, iters, out, blocks);
    ms[5] = run<5>(table, n - 1, iters, out, blocks);
    const double groups = (double)blocks * 256 * iters;  // lane-iterations ("sample-levels")
    printf("table %u entries (%.1f MB)\n", n, n * 4 / 1048576.0);
    for (int m = 0; m < 6; ++m)
      printf("  %-40s %8.3f ms  %7.2f ps per lane-iter  %6.2f CU-cycles@2.4GHz per lane-iter\n", names[m], ms[m],
             ms[m] * 1e9 / groups, ms[m] * 1e-3 / groups * 256 * 2.4e9);
    CK(hipFree(table)); CK(hipFree(out));
  }
  return 0;
}

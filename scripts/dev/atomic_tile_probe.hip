// Dev probe: how fast can 252 workgroups add their 256x256 fp32 partial tiles into a shared [N1, N2] output with
// device-scope fp32 atomics (the alternative to gemm_tn's slab write + tn_reduce), compared with plain stores of the
// same bytes into private slabs + a reduce pass?
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/atomic_tile_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// tile t of split s: 256 x 256 floats at out + tile offset (row-major [N1][N2] with 256-wide tiles)
__global__ __launch_bounds__(512) void atomic_tiles(float* out, int N2, int tiles_n, int tiles) {
  const int tile = blockIdx.x % tiles, tm = tile / tiles_n, tn = tile % tiles_n;
  float* base = out + (size_t)tm * 256 * N2 + tn * 256;
  const float v = 1.0f + blockIdx.x * 1e-6f;
  for (int i = threadIdx.x; i < 256 * 256; i += 512) {
    const int r = i >> 8, c = i & 255;
    __hip_atomic_fetch_add(base + (size_t)r * N2 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(512) void store_slabs(float* slabs, int N2, int tiles_n, int tiles, size_t slab_elems) {
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles, tm = tile / tiles_n, tn = tile % tiles_n;
  float* base = slabs + (size_t)split * slab_elems + (size_t)tm * 256 * N2 + tn * 256;
  const float v = 1.0f + blockIdx.x * 1e-6f;
  for (int i = threadIdx.x * 4; i < 256 * 256; i += 512 * 4) {
    const int r = i >> 8, c = i & 255;
    *reinterpret_cast<float4*>(base + (size_t)r * N2 + c) = float4{v, v, v, v};
  }
}
__global__ void reduce_slabs(const float* slabs, float* out, int n4, int splits, size_t stride4) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    float4 v = {0, 0, 0, 0};
    for (int s = 0; s < splits; ++s) { const float4 w = reinterpret_cast<const float4*>(slabs)[(size_t)s * stride4 + i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
int main() {
  const int shapes[3][2] = {{768, 768}, {2304, 768}, {3072, 768}};
  for (auto& sh : shapes) {
    const int N1 = sh[0], N2 = sh[1], tiles_n = N2 / 256, tiles = (N1 / 256) * tiles_n, splits = 256 / tiles, grid = tiles * splits;
    const size_t elems = (size_t)N1 * N2;
    float *out, *slabs;
    CK(hipMalloc(&out, elems * 4)); CK(hipMalloc(&slabs, elems * 4 * splits));
    CK(hipMemset(out, 0, elems * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a));
      for (int i = 0; i < 20; ++i) atomic_tiles<<<grid, 512>>>(out, N2, tiles_n, tiles);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const float t_atomic = ms / 20 * 1e3f;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a));
      for (int i = 0; i < 20; ++i) {
        store_slabs<<<grid, 512>>>(slabs, N2, tiles_n, tiles, elems);
        reduce_slabs<<<2048, 256>>>(slabs, out, (int)(elems / 4), splits, elems / 4);
      }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const float t_slab = ms / 20 * 1e3f;
    printf("N1=%d N2=%d: %d tiles x %d splits: atomics %.1f us   slab stores + reduce %.1f us\n", N1, N2, tiles, splits, t_atomic, t_slab);
    CK(hipFree(out)); CK(hipFree(slabs));
  }
  return 0;
}

// What does hipMemMap accept on this device, what does a mapping cost, and does a kernel stream from mapped memory as fast as
// from hipMalloc'ed memory?  (tph_vm_set in csrc/ctx.hip is built on the answers.)  Prints one JSON object.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_probe tools/ubench_vmm_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <vector>

__global__ void __launch_bounds__(256) k_read(const double2* __restrict__ p, size_t n2, double* sink) {
  double acc = 0.0;
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256 * 4) {
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < hi ? p[i + k * 256] : make_double2(0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

static double time_read(const void* p, size_t bytes, double* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = -2; r < 10; ++r) {
    if (r == 0) hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, 0, (const double2*)p, bytes / 16, sink);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)bytes * 10 / (ms * 1e-3) / 1e12;      // TB/s
}

int main() {
  const size_t MB = 1 << 20;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
  hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
  printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu, \"map\": [", gmin, grec);
  void* va = nullptr;
  const size_t VA = 4096 * MB;
  hipError_t e = hipMemAddressReserve(&va, VA, 0, nullptr, 0);
  if (e != hipSuccess) { printf("], \"error\": \"reserve: %s\"}\n", hipGetErrorString(e)); return 1; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  struct T { size_t off_mb, size_mb; };
  const T tests[] = {{0, 2}, {2, 2}, {4, 4}, {10, 8}, {18, 6}, {24, 8}, {32, 32}, {66, 64}, {130, 2}, {132, 4}, {136, 120}, {256, 256}, {512, 1}, {513, 3}};
  bool first = true;
  for (const T& t : tests) {
    hipMemGenericAllocationHandle_t h;
    hipError_t c = hipMemCreate(&h, t.size_mb * MB, &prop, 0), m = hipErrorUnknown, s = hipErrorUnknown;
    if (c == hipSuccess) {
      m = hipMemMap((char*)va + t.off_mb * MB, t.size_mb * MB, 0, h, 0);
      if (m == hipSuccess) s = hipMemSetAccess((char*)va + t.off_mb * MB, t.size_mb * MB, &acc, 1);
    }
    fflush(stdout);
    printf("%s{\"off_mb\": %zu, \"size_mb\": %zu, \"create\": \"%s\", \"map\": \"%s\", \"access\": \"%s\"}", first ? "" : ", ", t.off_mb, t.size_mb,
           hipGetErrorName(c), hipGetErrorName(m), hipGetErrorName(s));
    first = false;
    (void)hipGetLastError();
  }
  printf("], ");
  // cost of a piece (create + map + set access), every piece naturally aligned: 1024 pieces of 2 MiB, 256 of 8 MiB, 32 of 64 MiB,
  // each size in its own address range
  double* sink; hipMalloc(&sink, 64);
  void* plain; hipMalloc(&plain, 2048 * MB); hipMemset(plain, 0, 2048 * MB);
  hipDeviceSynchronize();
  const double r_plain = time_read(plain, 2048 * MB, sink);
  printf("\"read_tb_s_hipMalloc_2g\": %.3f, \"pieces\": [", r_plain);
  const size_t sizes[3] = {2 * MB, 8 * MB, 64 * MB};
  for (int k = 0; k < 3; ++k) {
    const size_t sz = sizes[k];
    const int n = (int)(2048 * MB / sz);
    void* vb = nullptr;
    if (hipMemAddressReserve(&vb, 2048 * MB, sz, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); continue; }
    auto t0 = std::chrono::steady_clock::now();
    int ok = 0;
    for (int i = 0; i < n; ++i) {
      hipMemGenericAllocationHandle_t h;
      char* at = (char*)vb + (size_t)i * sz;
      if (hipMemCreate(&h, sz, &prop, 0) == hipSuccess && hipMemMap(at, sz, 0, h, 0) == hipSuccess && hipMemSetAccess(at, sz, &acc, 1) == hipSuccess) ++ok;
      else break;
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (ok ? ok : 1);
    (void)hipGetLastError();
    double rd = -1.0;
    if (ok == n) {                                   // never touch an address that did not get its mapping
      hipMemset(vb, 0, 2048 * MB);
      hipDeviceSynchronize();
      rd = time_read(vb, 2048 * MB, sink);
    }
    printf("%s{\"piece_mb\": %zu, \"ok\": %d, \"of\": %d, \"us_per_piece\": %.1f, \"va_aligned_to_piece\": %d, \"read_tb_s_2g\": %.3f}", k ? ", " : "", sz / MB, ok, n, us,
           (int)(((uintptr_t)vb % sz) == 0), rd);
    fflush(stdout);
  }
  printf("]}\n");
  return 0;
}

// wstream — how fast can a pure 16-byte-per-lane write stream go on this part, as a function of how it is launched?
// (torch's fill reaches 6.9 TB/s; the persistent store pattern of the step kernels measured 4.3-5.9 TB/s.)
//   make -C tools wstream && tools/wstream <MiB>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

// persistent: every thread strides over the whole buffer
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_persist(v4* dst, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  const v4 val = {1, 2, 3, (uint32_t)threadIdx.x};
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < nvec; i += stride) dst[i] = val;
}
// persistent, but each wave writes CH consecutive KiB before striding on (like one 64-env obs chunk = 4.7 KiB)
template <int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void k_persist_chunk(v4* dst, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const v4 val = {1, 2, 3, (uint32_t)threadIdx.x};
  for (int64_t c = wave; c * (CH * 64) < nvec; c += nwaves)
#pragma unroll
    for (int k = 0; k < CH; ++k) { const int64_t i = c * (CH * 64) + k * 64 + lane; if (i < nvec) dst[i] = val; }
}
// persistent waves that take their next CH-KiB chunk from a global ticket counter: chunks are handed out in address order,
// so the chip's write front stays one dense range however the waves drift
template <int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void k_ticket(v4* dst, int64_t nvec, unsigned long long* counter, unsigned long long base) {
  const int lane = threadIdx.x & 63;
  const v4 val = {1, 2, 3, (uint32_t)threadIdx.x};
  const int64_t nchunks = (nvec + CH * 64 - 1) / (CH * 64);
  for (;;) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(counter, 1ull);
    t = __shfl(t, 0) - base;
    if ((int64_t)t >= nchunks) break;
#pragma unroll
    for (int k = 0; k < CH; ++k) { const int64_t i = (int64_t)t * (CH * 64) + k * 64 + lane; if (i < nvec) dst[i] = val; }
  }
}

// one tile per block, VT vectors per thread (the shape of torch's vectorized elementwise kernels)
template <int BLOCK, int VT>
__global__ __launch_bounds__(BLOCK) void k_tile(v4* dst, int64_t nvec) {
  const int64_t base = (int64_t)blockIdx.x * (BLOCK * VT);
  const v4 val = {1, 2, 3, (uint32_t)threadIdx.x};
#pragma unroll
  for (int k = 0; k < VT; ++k) { const int64_t i = base + k * BLOCK + threadIdx.x; if (i < nvec) dst[i] = val; }
}

template <typename F>
static double time_us(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
  const int64_t mib = argc > 1 ? atoll(argv[1]) : 1024;
  const int64_t bytes = mib << 20, nvec = bytes / 16;
  v4* d = nullptr;
  if (hipMalloc((void**)&d, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  auto report = [&](const char* name, double us) { printf("%-44s %8.1f us  %7.0f GB/s\n", name, us, bytes / us / 1e3); };
  const int it = 30;
  report("persistent 256 x 1024", time_us([&] { hipLaunchKernelGGL(k_persist<1024>, dim3(256), dim3(1024), 0, 0, d, nvec); }, it));
  report("persistent 512 x 1024", time_us([&] { hipLaunchKernelGGL(k_persist<1024>, dim3(512), dim3(1024), 0, 0, d, nvec); }, it));
  report("persistent 2048 x 256", time_us([&] { hipLaunchKernelGGL(k_persist<256>, dim3(2048), dim3(256), 0, 0, d, nvec); }, it));
  report("persistent 256 x 1024, 5 KiB per wave visit", time_us([&] { hipLaunchKernelGGL((k_persist_chunk<1024, 5>), dim3(256), dim3(1024), 0, 0, d, nvec); }, it));
  report("persistent 512 x 1024, 5 KiB per wave visit", time_us([&] { hipLaunchKernelGGL((k_persist_chunk<1024, 5>), dim3(512), dim3(1024), 0, 0, d, nvec); }, it));
  unsigned long long* ctr = nullptr;
  (void)hipMalloc((void**)&ctr, 8); (void)hipMemset(ctr, 0, 8);
  unsigned long long launches = 0;
  const int64_t nchunks5 = (nvec + 5 * 64 - 1) / (5 * 64);
  auto ticket = [&](int blocks) {
    // every wave takes exactly one ticket past the end: the counter advances by nchunks + waves per launch
    const unsigned long long per = (unsigned long long)nchunks5 + (unsigned long long)blocks * 16ull;
    hipLaunchKernelGGL((k_ticket<1024, 5>), dim3(blocks), dim3(1024), 0, 0, d, nvec, ctr, launches * per);
    ++launches;
  };
  report("ticketed 256 x 1024, 5 KiB per ticket", time_us([&] { ticket(256); }, it));
  (void)hipMemset(ctr, 0, 8); launches = 0;
  report("ticketed 512 x 1024, 5 KiB per ticket", time_us([&] { ticket(512); }, it));
  report("tile 256 threads x 4 vectors", time_us([&] { hipLaunchKernelGGL((k_tile<256, 4>), dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, 0, d, nvec); }, it));
  report("tile 256 threads x 1 vector", time_us([&] { hipLaunchKernelGGL((k_tile<256, 1>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, d, nvec); }, it));
  report("tile 1024 threads x 4 vectors", time_us([&] { hipLaunchKernelGGL((k_tile<1024, 4>), dim3((unsigned)((nvec + 4095) / 4096)), dim3(1024), 0, 0, d, nvec); }, it));
  (void)hipFree(d);
  return 0;
}

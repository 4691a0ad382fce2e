// wbench — store-pattern microbenchmark: what HBM rate does the step kernel's ACCESS PATTERN reach
// with no compute at all?  (a) one contiguous stream, (b) the kernel's 11 interleaved streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(1024) void k_stream(uint4* dst, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride)
    dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

// mode bit0: obs, bit1: reward, bit2: state rw, bit3: dir/done bytes, bit4: action reads
__global__ __launch_bounds__(1024) void k_pattern(uint8_t* obs, double* rew, uint4* state, uint8_t* dir, uint8_t* done,
                                                  const uint8_t* act, int64_t stride_envs, int64_t n_chunks, int mode) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t acc = 0;
  for (int64_t c = (int64_t)blockIdx.x * nw + wv; c < n_chunks; c += (int64_t)gridDim.x * nw) {
    const int64_t n = c * 64 + lane;
    uint4 st = make_uint4(lane, 2, 3, 4);
    if (mode & 4) st = state[n];
    if (mode & 16) for (int a = 0; a < 3; ++a) acc += act[a * stride_envs + n];
    st.x += acc;
    const bool nt = mode & 32;
    if (mode & 2) for (int a = 0; a < 3; ++a) { if (nt) __builtin_nontemporal_store((double)st.x, &rew[a * stride_envs + n]); else rew[a * stride_envs + n] = (double)st.x; }
    if (mode & 8) { for (int a = 0; a < 3; ++a) { if (nt) __builtin_nontemporal_store((uint8_t)st.y, &dir[a * stride_envs + n]); else dir[a * stride_envs + n] = (uint8_t)st.y; }
                    if (nt) __builtin_nontemporal_store((uint8_t)1, &done[n]); else done[n] = 1; }
    if (mode & 4) state[n] = st;
    if (mode & 1)
      for (int a = 0; a < 3; ++a) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4* dst = reinterpret_cast<v4*>(obs + (a * stride_envs + c * 64) * 75);
        const v4 val = {st.x, st.y, st.z, st.w};
        for (int i = 0; i < 5; ++i) { const int idx = i * 64 + lane; if (idx < 300) { if (nt) __builtin_nontemporal_store(val, &dst[idx]); else dst[idx] = val; } }
      }
  }
}

// the v4 kernel's exact streams, inputs prefetched one iteration ahead (asm loads, counted vmcnt):
// state uint4 RMW, actions dword read, dir_done dword, 3 f64 rewards, 3 obs chunks
template <int MODE>
__global__ __launch_bounds__(1024) void k_pattern4(uint8_t* obs, double* rew, uint4* state, uint32_t* dd,
                                                   const uint32_t* act, int64_t stride_envs, int64_t n_chunks) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t cs = (int64_t)gridDim.x * nw;
  int64_t c = (int64_t)blockIdx.x * nw + wv;
  v4 cur, nxt; uint32_t ca, na;
  if (c < n_chunks) {
    const void* p0 = state + c * 64 + lane; const void* p1 = act + c * 64 + lane;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cur) : "v"(p0) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=&v"(ca) : "v"(p1) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur), "+v"(ca) :: "memory");
  }
  for (; c < n_chunks; c += cs) {
    const int64_t n = c * 64 + lane;
    const bool more = c + cs < n_chunks;
    if (more) {
      const void* p0 = state + (c + cs) * 64 + lane; const void* p1 = act + (c + cs) * 64 + lane;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(nxt) : "v"(p0) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=&v"(na) : "v"(p1) : "memory");
    }
    v4 st = cur; st.x += ca;
    if (MODE == 0) {
      for (int a = 0; a < 3; ++a) rew[a * stride_envs + n] = (double)st.x;
      dd[n] = st.y;
    } else if (MODE == 1) {   // [S][32 B] record: 3 f64 rewards + dir_done
      v4* r = reinterpret_cast<v4*>(rew) + n * 2;
      r[0] = st; r[1] = st;
    }
    *reinterpret_cast<v4*>(state + n) = st;
    for (int a = 0; a < 3; ++a) {
      v4* dst = reinterpret_cast<v4*>(obs + (a * stride_envs + c * 64) * 75);
      for (int i = 0; i < 5; ++i) { const int idx = i * 64 + lane; if (idx < 300) dst[idx] = st; }
    }
    if (more) { asm volatile("s_waitcnt vmcnt(16)" : "+v"(nxt), "+v"(na) :: "memory"); cur = nxt; ca = na; }
  }
}

// inputs fetched in groups of G chunks per wave (one read burst), then G chunks of stores
template <int G>
__global__ __launch_bounds__(1024) void k_pattern_grouped(uint8_t* obs, double* rew, uint4* state, uint32_t* dd,
                                                          const uint32_t* act, int64_t stride_envs, int64_t n_chunks) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t cs = (int64_t)gridDim.x * nw;
  for (int64_t c0 = (int64_t)blockIdx.x * nw + wv; c0 < n_chunks; c0 += cs * G) {
    v4 st[G]; uint32_t ca[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t c = c0 + g * cs;
      if (c < n_chunks) { st[g] = *reinterpret_cast<const v4*>(state + c * 64 + lane); ca[g] = act[c * 64 + lane]; }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t c = c0 + g * cs;
      if (c >= n_chunks) break;
      const int64_t n = c * 64 + lane;
      v4 s = st[g]; s.x += ca[g];
      for (int a = 0; a < 3; ++a) rew[a * stride_envs + n] = (double)s.x;
      dd[n] = s.y;
      *reinterpret_cast<v4*>(state + n) = s;
      for (int a = 0; a < 3; ++a) {
        v4* dst = reinterpret_cast<v4*>(obs + (a * stride_envs + c * 64) * 75);
        for (int i = 0; i < 5; ++i) { const int idx = i * 64 + lane; if (idx < 300) dst[idx] = s; }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4194304;
  const int nw = argc > 2 ? atoi(argv[2]) : 16;
  const int grid = argc > 3 ? atoi(argv[3]) : 256;
  uint8_t *obs, *dir, *done, *act; double* rew; uint4* state;
  hipMalloc((void**)&obs, n * 225); hipMalloc((void**)&rew, n * 24); hipMalloc((void**)&state, n * 16);
  hipMalloc((void**)&dir, n * 3); hipMalloc((void**)&done, n); hipMalloc((void**)&act, n * 3);
  hipMemset(act, 1, n * 3); hipMemset(state, 0, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double bytes, auto fn) {
    std::vector<float> v;
    for (int r = 0; r < 7; ++r) {
      hipEventRecord(e0, 0);
      for (int k = 0; k < 20; ++k) fn();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); v.push_back(ms * 1e3f / 20);
    }
    std::sort(v.begin(), v.end());
    printf("%-44s %9.2f us  %7.0f GB/s\n", name, v[3], bytes / v[3] / 1e3);
  };
  timeit("stream: contiguous dwordx4 (obs bytes)", n * 225.0, [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (uint4*)obs, n * 225 / 16); });
  struct { const char* name; int mode; double bytes; } pats[] = {
      {"pattern: obs only", 1, 225.0}, {"pattern: obs+reward", 3, 249.0}, {"pattern: obs+reward+state", 7, 281.0},
      {"pattern: all writes + state read", 15, 285.0}, {"pattern: all (kernel's streams)", 31, 288.0},
      {"pattern: everything but obs", 30, 63.0}, {"pattern: all, nt on obs/reward/dir/done", 63, 288.0},
      {"pattern: obs only nt", 33, 225.0}};
  for (auto& pt : pats)
    timeit(pt.name, n * pt.bytes, [&] { hipLaunchKernelGGL(k_pattern, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dir, done, act, n, n / 64, pt.mode); });
  uint32_t* dd; hipMalloc((void**)&dd, n * 4);
  uint32_t* act4; hipMalloc((void**)&act4, n * 4); hipMemset(act4, 1, n * 4);
  hipFree(rew); hipMalloc((void**)&rew, n * 32);
  timeit("pattern4: v4 streams, prefetched inputs", n * 289.0, [&] { hipLaunchKernelGGL(k_pattern4<0>, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dd, act4, n, n / 64); });
  timeit("pattern4: reward+dir_done as one 32-B record", n * 293.0, [&] { hipLaunchKernelGGL(k_pattern4<1>, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dd, act4, n, n / 64); });
  timeit("pattern4: obs + state RMW + act only", n * 261.0, [&] { hipLaunchKernelGGL(k_pattern4<3>, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dd, act4, n, n / 64); });
  timeit("grouped reads G=4", n * 289.0, [&] { hipLaunchKernelGGL(k_pattern_grouped<4>, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dd, act4, n, n / 64); });
  timeit("grouped reads G=8", n * 289.0, [&] { hipLaunchKernelGGL(k_pattern_grouped<8>, dim3(grid), dim3(nw * 64), 0, 0, obs, rew, state, dd, act4, n, n / 64); });
  return 0;
}

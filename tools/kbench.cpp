// kbench — torch-free micro-benchmark of the step kernel for A/B work.
// usage: kbench <map.bin> <n_envs> <steps> <rounds> lib1.so [lib2.so ...]
// env MG_WORKLOAD=doorkey<size>x<agents> (e.g. doorkey8x3) times the mutable-grid kernel on DoorKey instead of
// the Cluttered map (the map argument is still read but unused).
// Each library is dlopen'ed separately (its own copy of the kernels); rounds are interleaved
// across libraries in one process (cdna_hip_programming.md §5.4 rule 24).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#include "mgrid.h"

struct Api {
  void* dl;
  decltype(&mg_create) create;
  decltype(&mg_set_static_map) set_map;
  decltype(&mg_reset) reset;
  decltype(&mg_step) step;
  decltype(&mg_get_buffers) get_buffers;
  decltype(&mg_get_stream) get_stream;
  decltype(&mg_sync) sync;
  decltype(&mg_check_errors) check;
  decltype(&mg_last_error) last_error;
  mg_handle* h;
  uint8_t* ring;
  hipStream_t stream;
  std::vector<float> us;
};

#define SYM(a, f, n) a.f = (decltype(a.f))dlsym(a.dl, n); if (!a.f) { fprintf(stderr, "missing %s\n", n); return 1; }

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: kbench map.bin n_envs steps rounds lib.so...\n"); return 2; }
  const int64_t n = atoll(argv[2]);
  const int steps = atoi(argv[3]), rounds = atoi(argv[4]);
  std::vector<uint8_t> cells(15 * 15 * 3);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(cells.data(), 1, cells.size(), f) != cells.size()) { fprintf(stderr, "bad map\n"); return 2; }
  fclose(f);
  const int R = 16;
  // workload: "" = Cluttered-Fixed-15x15 (shared map); doorkey<S>x<A>, meetup<S>x<A> (3 goals, no clutter),
  // cluttered<S>x<A> (random goal + 25 walls per env and reset) = mutable-grid kernel
  int dk_size = 0, dk_agents = 0, kind = MG_KIND_STATIC;
  if (const char* w = getenv("MG_WORKLOAD")) {
    if (sscanf(w, "doorkey%dx%d", &dk_size, &dk_agents) == 2) kind = MG_KIND_DOORKEY;
    else if (sscanf(w, "meetup%dx%d", &dk_size, &dk_agents) == 2) kind = MG_KIND_MEETUP;
    else if (sscanf(w, "cluttered%dx%d", &dk_size, &dk_agents) == 2) kind = MG_KIND_CLUTTERED;
  }
  const int A = dk_size ? dk_agents : 3, V = kind == MG_KIND_DOORKEY ? 7 : 5;
  // bytes one env-step has to move: obs + dir_done + rewards + actions + env_state r/w (+ carry, aux and grid r/w)
  const double obs_bytes = kind == MG_KIND_MEETUP ? dk_size * dk_size * 3 : A * V * V * 3;
  const double env_bytes = !dk_size ? 308.0 /* SURVEY §8(d) */ : obs_bytes + (A < 4 ? 4 : 8) + 8.0 * A + 4 + 8.0 * (A + 1) +
                           (dk_size ? 8 + 2.0 * ((dk_size * dk_size + 3) / 4 * 4) : 0) +
                           ((kind == MG_KIND_MEETUP || kind == MG_KIND_CLUTTERED) ? 32 : 0);
  std::vector<Api> apis;
  for (int i = 5; i < argc; ++i) {
    Api a{};
    a.dl = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
    if (!a.dl) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
    SYM(a, create, "mg_create") SYM(a, set_map, "mg_set_static_map") SYM(a, reset, "mg_reset") SYM(a, step, "mg_step")
    SYM(a, get_buffers, "mg_get_buffers") SYM(a, get_stream, "mg_get_stream") SYM(a, sync, "mg_sync")
    SYM(a, check, "mg_check_errors") SYM(a, last_error, "mg_last_error")
    mg_config c{};
    c.width = c.height = 15; c.n_agents = 3; c.view_size = 5; c.max_steps = 100; c.shaping_kind = MG_SHAPING_V8;
    c.auto_reset = 1; c.n_envs = n; c.seed = 0xC0FFEE;
    c.kernel_path = getenv("MG_PATH") ? atoi(getenv("MG_PATH")) : 0;
    if (dk_size) {
      c.width = c.height = dk_size; c.n_agents = dk_agents; c.view_size = V; c.env_kind = kind;
      c.max_steps = kind == MG_KIND_DOORKEY ? 10 * dk_size * dk_size : 250;
      if (kind == MG_KIND_MEETUP) { c.gen_n_goals = 3; c.gen_n_clutter = 0; c.view_size = 7; }
      if (kind == MG_KIND_CLUTTERED) { c.gen_n_goals = 1; c.gen_n_clutter = 25; }
    }
    if (a.create(&c, 0, &a.h) != MG_OK) { fprintf(stderr, "create: %s\n", a.last_error()); return 1; }
    if (!dk_size && a.set_map(a.h, cells.data()) != MG_OK) { fprintf(stderr, "map: %s\n", a.last_error()); return 1; }
    mg_buffers b;
    a.get_buffers(a.h, &b);
    const size_t sz = (size_t)4 * b.env_stride;
    std::vector<uint8_t> host(sz * R);
    uint64_t s = 88172645463325252ull + i;
    for (auto& v : host) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint8_t)((s >> 33) % 7); }
    if (hipMalloc((void**)&a.ring, sz * R) != hipSuccess) return 1;
    hipMemcpy(a.ring, host.data(), sz * R, hipMemcpyHostToDevice);
    void* st; a.get_stream(a.h, &st); a.stream = (hipStream_t)st;
    a.reset(a.h, nullptr);
    for (int k = 0; k < 50; ++k) a.step(a.h, a.ring + (size_t)(k % R) * sz);
    // stop at the first library whose warm-up does not complete cleanly: nothing later is launched on a faulted device
    if (a.sync(a.h) != MG_OK || hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "%s: warm-up failed: %s\n", argv[i], a.last_error()); return 3; }
    apis.push_back(a);
  }
  typedef int (*phase_fn)(unsigned long long*, int);
  for (auto& a : apis)        // timing-only builds with -DMG_PHASE_TIMERS: clear what the warm-up accumulated
    if (phase_fn ph = (phase_fn)dlsym(a.dl, "mg_debug_phase_counters")) { unsigned long long tmp[16]; ph(tmp, 1); }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < rounds; ++r)
    for (auto& a : apis) {
      mg_buffers b; a.get_buffers(a.h, &b);
      const size_t sz = (size_t)4 * b.env_stride;
      hipEventRecord(e0, a.stream);
      for (int k = 0; k < steps; ++k) a.step(a.h, a.ring + (size_t)(k % R) * sz);
      hipEventRecord(e1, a.stream);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        fprintf(stderr, "timing failed (device error)\n"); return 3;
      }
      a.us.push_back(ms * 1e3f / steps);
    }
  for (size_t i = 0; i < apis.size(); ++i) {
    auto v = apis[i].us; std::sort(v.begin(), v.end());
    const float med = v[v.size() / 2], mn = v.front();
    uint32_t err = 0; apis[i].check(apis[i].h, &err);
    printf("%-40s n=%lld  median %.2f us  min %.2f us  alg %.0f GB/s (%.1f%% of 8 TB/s, %.0f B/env-step)  %.3e agent-steps/s  err=%u\n",
           argv[5 + i], (long long)n, med, mn, env_bytes * n / med / 1e3, env_bytes * n / med / 1e3 / 80.0, env_bytes,
           (double)A * n / (med * 1e-6), err);
    if (phase_fn ph = (phase_fn)dlsym(apis[i].dl, "mg_debug_phase_counters")) {
      unsigned long long c[16];
      if (ph(c, 0) == MG_OK) {
        unsigned long long tot = 0;
        for (int k = 0; k < 11; ++k) tot += c[k];
        const double chunk_steps = (double)c[11];          // chunks the waves counted themselves
        static const char* names[11] = {"grid rows -> LDS", "unpack, actions, order", "step_one_agent x A", "rewards, shaper",
                                        "reset / auto-reset", "state + grid stores", "views", "staging + obs stores", "tail",
                                        "prefetch issue", "counted wait for prefetch"};
        printf("    phase timers (shader-clock cycles per chunk-step of one wave, stalls included; %% of the chunk; %.0f chunk-steps):\n", chunk_steps);
        for (int k = 0; k < 11; ++k) printf("      %-26s %9.0f  %5.1f%%\n", names[k], c[k] / chunk_steps, 100.0 * c[k] / (double)tot);
        printf("      %-26s %9.0f\n", "sum", tot / chunk_steps);
      }
    }
    if (getenv("MG_KBENCH_GEOMETRY")) {      // launch geometry of the handle (workgroup size, grid, LDS per workgroup)
      auto info = (decltype(&mg_get_launch_info))dlsym(apis[i].dl, "mg_get_launch_info");
      int32_t path = 0, bt = 0, gb = 0, lds = 0;
      if (info && info(apis[i].h, &path, &bt, &gb, &lds) == MG_OK)
        printf("    geometry: %d threads x %d workgroups, %d B LDS per workgroup (%d waves in flight)\n", bt, gb, lds, bt / 64 * gb);
    }
  }
  return 0;
}

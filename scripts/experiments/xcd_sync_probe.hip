// xcd_sync_probe.hip - EXPERIMENT (profiles/r06_experiments.md 11): what ONE phase boundary of a fused routing kernel costs when all workgroups
// of a segment share an XCD (its L2 as the coherence point) against the agent-scope form round 5 measured (r05_experiments.md 3).
// A "round" = every workgroup stores 1 KiB, meets its team at a barrier, loads the 1 KiB of another team member and checks it.
//   mode 0: team = the whole grid;   barrier = agent-scope atomic add + agent-scope relaxed poll; data = sc1 stores / sc1 loads (round 5's form)
//   mode 1: team = the workgroups with the same (blockIdx % 8) = one XCD (observed placement, checked with HW_REG_XCC_ID);
//           barrier = WORKGROUP-scope atomic add (executes in the XCD's L2) + agent-scope relaxed poll (sc1: bypasses L1, L2-served);
//           data = plain stores (kept in the L2) / sc1 loads
//   mode 2: mode 1's data path with mode 0's agent-scope barrier atomics (separates the two effects)
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/experiments/xcd_sync_probe.hip -o scripts/experiments/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* data, int* ctr, int rounds, int* bad, int* xcc_mismatch) {
  const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  const int team = MODE == 0 ? 0 : (b & 7), team_size = MODE == 0 ? G : G / 8;
  int* c = ctr + team * 64;                                   // (a 256-byte line per team)
  if (MODE != 0 && tid == 0) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 15) != (b & 7)) atomicAdd(xcc_mismatch, 1);
  }
  const int partner = MODE == 0 ? (b + 1) % G : (b + 8) % G;   // another member of the team
  int errs = 0;
  for (int r = 1; r <= rounds; ++r) {
    uint32_t* mine = data + ((size_t)(r & 1) * G + b) * 256;
    const uint32_t v = (uint32_t)r * 65536u + (uint32_t)b;
    if (MODE == 0) st_sc1(mine + tid, v + tid); else mine[tid] = v + tid;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 1) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = r * team_size;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const uint32_t* theirs = data + ((size_t)(r & 1) * G + partner) * 256;
    const uint32_t got = ld_sc1(theirs + tid);
    if (got != (uint32_t)r * 65536u + (uint32_t)partner + tid) ++errs;
  }
  if (errs) atomicAdd(bad, errs);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t* data; int *ctr, *bad, *mm;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {64, 128, cus, 2 * cus}) {
    CK(hipMalloc(&data, (size_t)2 * G * 1024)); CK(hipMalloc(&ctr, 8 * 256)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&mm, 4));
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f; int hbad = 0, hmm = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 8 * 256)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(mm, 0, 4)); CK(hipMemset(data, 0, (size_t)2 * G * 1024));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, data, ctr, rounds, bad, mm);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, data, ctr, rounds, bad, mm);
        else hipLaunchKernelGGL(probe<2>, dim3(G), dim3(256), 0, 0, data, ctr, rounds, bad, mm);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        int hb, hm; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mm, 4, hipMemcpyDeviceToHost));
        hbad += hb; hmm += hm;
      }
      printf("grid %4d  mode %d  %7.3f us per round  stale words %d  workgroups off their XCD %d\n", G, mode, best * 1e3f / rounds, hbad, hmm);
    }
    CK(hipFree(data)); CK(hipFree(ctr)); CK(hipFree(bad)); CK(hipFree(mm));
  }
  return 0;
}

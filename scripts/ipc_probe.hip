// Probe (run on the GPU box): can two processes on ONE device share an uncached / fine-grained allocation through hipIpc
// handles and signal each other from running kernels?  (parent = rank 0, forked child = rank 1; handles travel through a pipe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e)); exit(2); } } while (0)
static int g_rank = 0;
__global__ void pingpong(unsigned long long* mine, unsigned long long* theirs, int rank, int rounds, int* err) {
  // rank 0 writes k into theirs[0], waits for mine[0] == k (rank 1 echoes)
  for (int k = 1; k <= rounds; ++k) {
    if (rank == 0) __hip_atomic_store(theirs, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)k) {
      if (++spins > (1u << 24)) { *err = k; return; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (rank == 1) __hip_atomic_store(theirs, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main(int argc, char** argv) {
  const unsigned flags = argc > 1 ? (unsigned)atoi(argv[1]) : 3;   // 0: hipMalloc, 1: finegrained, 3: uncached
  int p01[2], p10[2];
  if (pipe(p01) || pipe(p10)) return 1;
  pid_t pid = fork();            // fork BEFORE any HIP call
  g_rank = pid == 0 ? 1 : 0;
  CK(hipSetDevice(0));
  void* mine = nullptr;
  if (flags == 0) CK(hipMalloc(&mine, 4096));
  else CK(hipExtMallocWithFlags(&mine, 4096, flags));
  CK(hipMemset(mine, 0, 4096));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h, ho;
  CK(hipIpcGetMemHandle(&h, mine));
  int wr = g_rank == 0 ? p01[1] : p10[1], rd = g_rank == 0 ? p10[0] : p01[0];
  if (write(wr, &h, sizeof(h)) != sizeof(h)) return 1;
  if (read(rd, &ho, sizeof(ho)) != sizeof(ho)) return 1;
  void* theirs = nullptr;
  CK(hipIpcOpenMemHandle(&theirs, ho, hipIpcMemLazyEnablePeerAccess));
  int* err; CK(hipMalloc((void**)&err, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  CK(hipEventRecord(e0, 0));
  pingpong<<<1, 1>>>((unsigned long long*)mine, (unsigned long long*)theirs, g_rank, rounds, err);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("rank %d flags %u: err=%d  %.2f us per round trip (both kernels of two processes resident)\n", g_rank, flags, herr, 1000.f * ms / rounds);
  CK(hipIpcCloseMemHandle(theirs));
  if (g_rank == 0) { int st; waitpid(pid, &st, 0); return herr != 0 || st != 0; }
  return herr != 0;
}

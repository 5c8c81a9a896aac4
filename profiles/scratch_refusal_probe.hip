// What does the HIP runtime do when a kernel's scratch (private segment) reservation cannot be met?  (VERDICT r03 "what's weak" 6: an abort
// inside rt_render_tiles_device under a build whose kernels asked for 1104 - 5984 B of scratch per lane.)  A stand-alone experiment, run
// ONCE on the GPU box as a child process with core dumps off (profiles/r04_run1.sh):
//   probe <bytes to leave free> <which: 0 = 4 KB per lane, 1 = 32 KB per lane>
// It fills device memory with hipMalloc until about <bytes to leave free> are left, launches a kernel whose private array cannot live in
// registers, synchronises, and prints every return code.  If the runtime aborts the process instead of returning an error, the parent
// sees the signal.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int WORDS>
__global__ void hog(unsigned *out, unsigned n) {
  unsigned a[WORDS];
  for (int i = 0; i < WORDS; i++) a[i] = i * 2654435761u + threadIdx.x;
  unsigned s = 0;
  for (unsigned k = 0; k < n; k++) s += a[(s + k * 7u) % WORDS];      // dynamic indexing: the array stays in scratch
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define SAY(call) do { hipError_t e_ = (call); printf("%-44s -> %d %s\n", #call, (int)e_, hipGetErrorString(e_)); fflush(stdout); } while (0)

int main(int argc, char **argv) {
  const size_t leave = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)2 << 30;
  const int which = argc > 2 ? atoi(argv[2]) : 0;
  hipDeviceProp_t p;
  SAY(hipGetDeviceProperties(&p, 0));
  const size_t slots = (size_t)p.multiProcessorCount * (p.maxThreadsPerMultiProcessor / 64);
  hipFuncAttributes fa;
  SAY(hipFuncGetAttributes(&fa, which ? (const void *)hog<8192> : (const void *)hog<1024>));
  printf("CUs %d, threads per CU %d -> %zu wave slots; kernel scratch %zu B per lane -> reservation %zu B for every slot\n", p.multiProcessorCount,
         p.maxThreadsPerMultiProcessor, slots, (size_t)fa.localSizeBytes, (size_t)fa.localSizeBytes * 64u * slots);
  size_t fr = 0, tot = 0;
  SAY(hipMemGetInfo(&fr, &tot));
  printf("free %zu of %zu\n", fr, tot);
  std::vector<void *> held;
  while (fr > leave + ((size_t)1 << 30)) {
    size_t want = fr - leave;
    if (want > ((size_t)16 << 30)) want = (size_t)16 << 30;
    void *q = nullptr;
    if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); break; }
    held.push_back(q);
    (void)hipMemGetInfo(&fr, &tot);
  }
  printf("holding %zu allocations, free now %zu\n", held.size(), fr);
  fflush(stdout);
  unsigned *out = nullptr;
  SAY(hipMalloc((void **)&out, (size_t)65536 * 256 * 4));
  if (which) hipLaunchKernelGGL(hog<8192>, dim3(65536), dim3(256), 0, 0, out, 64u);
  else hipLaunchKernelGGL(hog<1024>, dim3(65536), dim3(256), 0, 0, out, 64u);
  SAY(hipGetLastError());
  SAY(hipDeviceSynchronize());
  SAY(hipMemGetInfo(&fr, &tot));
  printf("free after the launch %zu\n", fr);
  for (void *q : held) (void)hipFree(q);
  printf("done\n");
  return 0;
}

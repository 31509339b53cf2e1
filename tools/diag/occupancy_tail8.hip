// How many workgroups of the 512-thread tail kernels share a CU (hipOccupancyMaxActiveBlocksPerMultiprocessor),
// and the 1024-thread ones for comparison.  Build + run on the MI355X box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I st-dadk_amd/csrc -I include tools/diag/occupancy_tail8.hip -o /tmp/occ && /tmp/occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tail.h"
#include "tail_body.h"
namespace stdadk {
void set_error(const char *, ...) {}
template <int NW, int MT, bool BF>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void k(TailFwdArgs f, TailBwdArgs b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NW];
  Tail<NW>::template tail_fwd_body<MT, false, BF>(f, smem, red, blockIdx.x);
  __syncthreads();
  Tail<NW>::template tail_bwd_body<MT, BF>(b, smem, blockIdx.x);
}
}
using namespace stdadk;
template <int NW, int MT, bool BF>
void q(const char *name) {
  size_t lds = Tail<NW>::template tail_bwd_lds_floats<MT, BF>() * 4;
  size_t fwd = BF ? (size_t)(16 * MT * ACT_LD) * 4 + (size_t)16 * MT * ABF_LD * 2 : (size_t)2 * 16 * MT * ACT_LD * 4;
  if (fwd > lds) lds = fwd;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<NW, MT, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<NW, MT, BF>, 64 * NW, lds);
  hipFuncAttributes at;
  hipFuncGetAttributes(&at, reinterpret_cast<const void *>(k<NW, MT, BF>));
  printf("%-28s threads %4d dynamic LDS %6zu B static %zu B regs %d -> %d workgroups per CU (%s)\n", name, 64 * NW, lds,
         at.sharedSizeBytes, at.numRegs, nb, hipGetErrorString(e));
}
int main() {
  q<8, 2, false>("Tail<8> 32 rows fp32");
  q<8, 2, true>("Tail<8> 32 rows bf16");
  q<8, 1, false>("Tail<8> 16 rows fp32");
  q<16, 4, false>("Tail<16> 64 rows fp32");
  q<16, 2, false>("Tail<16> 32 rows fp32");
  q<16, 1, false>("Tail<16> 16 rows fp32");
  return 0;
}

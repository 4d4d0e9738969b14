// L2 -> LDS fill rate of global_load_lds_dwordx4 by the shape of the piece one wave instruction fetches
// (1 KiB = R rows x (1024 / R) contiguous bytes), every workgroup re-reading the same L2-resident panel.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/fill_lab.hip -o tools/experiments/fill_lab
// run:   tools/experiments/fill_lab [panel_KiB] [row_stride_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// each wave issues `per_step` DMA instructions per step into a ring of `stages` x 8 waves x per_step KiB
template <int PER_STEP, int STAGES>
__global__ __launch_bounds__(512) void fill_kernel(const char* panel, long panel_bytes, int rows_per_instr, long row_stride,
                                                   int steps, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const int seg = 1024 / rows_per_instr;                 // contiguous bytes per row and instruction
  const int lanes_per_row = seg / 16;
  const int r = lane / lanes_per_row, c = lane % lanes_per_row;
  // instruction i of wave w at step s reads rows (unit * rows_per_instr + r) of the panel, k offset advancing per step
  const long rows_in_panel = panel_bytes / row_stride;
  long koff = 0;
  int buf = 0;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < PER_STEP; ++i) {
      const long row = ((long)(wave * PER_STEP + i) * rows_per_instr + r + (long)blockIdx.x * 0) % rows_in_panel;
      const char* src = panel + row * row_stride + (koff % (row_stride - seg + 1)) / 16 * 16 + c * 16;
      glds16(src, lds_base + (buf * 8 * PER_STEP + wave * PER_STEP + i) * 1024);
    }
    koff += seg;
    buf = buf + 1 == STAGES ? 0 : buf + 1;
    if (s >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP * (STAGES - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[17] == 123) sink[0] = 1;
}

int main(int argc, char** argv) {
  const long panel_kib = argc > 1 ? atol(argv[1]) : 2048;
  const long row_stride = argc > 2 ? atol(argv[2]) : 4096;
  const long panel_bytes = panel_kib * 1024;
  char* panel;
  int* sink;
  hipMalloc(&panel, panel_bytes + 4096);
  hipMalloc(&sink, 4);
  hipMemset(panel, 1, panel_bytes + 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  constexpr int PER = 5, ST = 3;
  const int smem = ST * 8 * PER * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<PER, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int steps = 2000;
  for (int grid : {32, 256}) {
    for (int rows : {1, 2, 4, 8, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<PER, ST>), dim3(grid), dim3(512), smem, 0, panel, panel_bytes, rows, row_stride, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          const double bytes = (double)grid * steps * 8 * PER * 1024;
          printf("grid %3d  piece %2d rows x %4d B  (row stride %ld, panel %ld KiB): %7.1f GB/s per CU  %6.2f TB/s chip\n", grid, rows,
                 1024 / rows, row_stride, panel_kib, bytes / ms / 1e6 / grid, bytes / ms / 1e9);
        }
      }
    }
  }
  return 0;
}

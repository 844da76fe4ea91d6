// Developer tool: what the pieces of ONE forward pass of the flat block LU cost on gfx950 (cycles per wave, WPB wavefronts per block,
// blocks sized so that 4 blocks share a CU): 6 x ds_read_b128, the 2x2 arithmetic chain, 4 x ds_add_f64 vs plain read-modify-write.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_pass_bench.hip -o tools/_build/lds_pass_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ double frcp(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); return fma(fma(-x, r, 1.0), r, r); }
template <int MODE>
__global__ void k(long long* out, int iters, int lanes_on, int stride, int half2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* A = reinterpret_cast<double*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int nd = half2 * 2;
  for (int i = tid; i < nd; i += blockDim.x) A[i] = 1.0 + 1e-3 * i;
  __syncthreads();
  const bool on = lane < lanes_on;
  // "slots": 16-byte chunks; row 1 at +2048 doubles.  lane l works on slots (l*stride) % 1024 etc.
  const unsigned smask = (unsigned)(half2 / 2 - 1);      // slots per row half (power of two)
  unsigned fp = ((lane * stride) & smask) * 16, fl = ((lane * stride + 7) & smask) * 16, fu = ((lane * stride + 13) & smask) * 16, fd = ((lane * stride + 29) & smask) * 16;
  char* a0 = reinterpret_cast<char*>(A); char* a1 = a0 + (size_t)half2 * 8;
  double acc = 0.0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (on) {
      if (MODE == 0 || MODE >= 4) {             // reads only / full pass
        const double2 dA = *(const double2*)(a0 + fp), dB = *(const double2*)(a1 + fp), lA = *(const double2*)(a0 + fl), lB = *(const double2*)(a1 + fl);
        const double2 uA = *(const double2*)(a0 + fu), uB = *(const double2*)(a1 + fu);
        if (MODE == 0) { acc += dA.x + dB.y + lA.x + lB.y + uA.x + uB.y; }
        else {
          const double rd = -frcp(fma(dA.x, dB.y, -dA.y * dB.x));
          const double t00 = fma(lA.x, dB.y, -lA.y * dB.x), t01 = fma(lA.y, dA.x, -lA.x * dA.y);
          const double t10 = fma(lB.x, dB.y, -lB.y * dB.x), t11 = fma(lB.y, dA.x, -lB.x * dA.y);
          const double r00 = fma(t00, uA.x, t01 * uB.x) * rd * 1e-9, r10 = fma(t10, uA.x, t11 * uB.x) * rd * 1e-9;
          const double r01 = fma(t00, uA.y, t01 * uB.y) * rd * 1e-9, r11 = fma(t10, uA.y, t11 * uB.y) * rd * 1e-9;
          if (MODE == 4) { atomicAdd((double*)(a0 + fd), r00); atomicAdd((double*)(a1 + fd), r10); atomicAdd((double*)(a0 + fd) + 1, r01); atomicAdd((double*)(a1 + fd) + 1, r11); }
          if (MODE == 5) {                      // destination read with the operands, plain stores
            const double2 x0 = *(const double2*)(a0 + fd), x1 = *(const double2*)(a1 + fd);
            *(double2*)(a0 + fd) = make_double2(x0.x + r00, x0.y + r01); *(double2*)(a1 + fd) = make_double2(x1.x + r10, x1.y + r11);
          }
          if (MODE == 6) acc += r00 + r10 + r01 + r11;     // arithmetic only (with the reads)
        }
      }
      if (MODE == 1) { atomicAdd((double*)(a0 + fd), 1e-9); atomicAdd((double*)(a1 + fd), 1e-9); atomicAdd((double*)(a0 + fd) + 1, 1e-9); atomicAdd((double*)(a1 + fd) + 1, 1e-9); }
      if (MODE == 2) { *(double2*)(a0 + fd) = make_double2(acc, 1.0); *(double2*)(a1 + fd) = make_double2(1.0, acc); }
      if (MODE == 3) { const double2 x0 = *(const double2*)(a0 + fd), x1 = *(const double2*)(a1 + fd);
                       *(double2*)(a0 + fd) = make_double2(x0.x + 1e-9, x0.y); *(double2*)(a1 + fd) = make_double2(x1.x, x1.y + 1e-9); }
    }
    if (blockDim.x > 64) __syncthreads(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    fd = (fd + 16 * 64) & (smask * 16 + 15) & ~15u; fp = (fp + 16 * 3) & (smask * 16 + 15) & ~15u;
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
  if (acc == 12345.678) out[0] = 0;
}
int main(int argc, char** argv) {
  long long* d; CK(hipMalloc(&d, 8 * 8192));
  const int iters = 400;
  const char* names[] = {"6 x ds_read_b128", "4 x ds_add_f64", "2 x ds_write_b128", "2 x read_b128 + 2 x write_b128 (rmw)", "full item, atomics", "full item, dst read + plain stores", "reads + arithmetic only"};
  const bool thr = argc > 1;            // any argument: the throughput regime of the 36-substation N-1 kernel (12 one-wavefront blocks of 12.7 KB per CU)
  const int lds_bytes = thr ? 12800 : 40000;
  if (thr) printf("throughput regime: 12 blocks of %d B LDS per CU, one wavefront each; cycles per pass of ONE wavefront (x 12 per CU in flight)\n", lds_bytes);
  for (int wpb : {1, 2}) for (int blocks : {256, 1024, 3072}) for (int lanes_on : {64, 48, 16}) for (int stride : {1, 5}) {
    if (thr != (blocks == 3072) || (thr && (wpb != 1 || stride != 5))) continue;
    if (!thr && lanes_on == 48) continue;
    printf("--- %d wave(s)/block, %d blocks (%d per CU), %d lanes active, slot stride %d\n", wpb, blocks, blocks / 256, lanes_on, stride);
    for (int mode = 0; mode < 7; ++mode) {
      auto launch = [&](auto kern) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), lds_bytes, 0, d, iters, lanes_on, stride, thr ? 512 : 2048); };
      switch (mode) { case 0: launch(k<0>); break; case 1: launch(k<1>); break; case 2: launch(k<2>); break; case 3: launch(k<3>); break; case 4: launch(k<4>); break; case 5: launch(k<5>); break; default: launch(k<6>); }
      CK(hipDeviceSynchronize());
      static long long h[8192]; CK(hipMemcpy(h, d, 8 * blocks * wpb, hipMemcpyDeviceToHost));
      double s = 0; for (int i = 0; i < blocks * wpb; ++i) s += h[i];
      printf("  %-44s %8.1f cycles per pass\n", names[mode], s / (blocks * wpb) / iters);
    }
  }
  return 0;
}

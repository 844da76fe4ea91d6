// Developer tool: LDS / VALU latency calibration on gfx950 (one wavefront per block).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void lat(long long* out, int iters) {
  __shared__ int chain[256];
  __shared__ double acc[256];
  const int tid = threadIdx.x;
  for (int i = tid; i < 256; i += 64) { chain[i] = (i + 17) & 255; acc[i] = 1.0; }
  __syncthreads();
  int p = tid;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) p = chain[p];                      // dependent ds_read_b32
  long long t1 = __builtin_readcyclecounter();
  double v = 1.0 + p;
  for (int i = 0; i < iters; ++i) { v = acc[(p + i) & 255] + v * 1e-30; p = (int)v & 255; }   // ds_read_b64 + cvt chain
  long long t2 = __builtin_readcyclecounter();
  double a = v;
  for (int i = 0; i < iters; ++i) a = fma(a, 1.0000001, 1e-9);       // dependent f64 FMA chain
  long long t3 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) { atomicAdd(&acc[(tid + i) & 255], 1e-9); a += acc[(tid + i + 1) & 255]; }   // atomic + dependent-ish read
  long long t4 = __builtin_readcyclecounter();
  double r = a;
  for (int i = 0; i < iters; ++i) r = __builtin_amdgcn_rcp(r + 1.5);  // dependent v_rcp_f64 + add
  long long t5 = __builtin_readcyclecounter();
  if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = (long long)(r + p); }
}
int main() {
  long long* d; hipMalloc(&d, 64);
  const int iters = 2000;
  for (int blocks : {1, 256 * 8, 256 * 16}) {
    hipLaunchKernelGGL(lat, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipDeviceSynchronize();
    long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    printf("blocks %5d: ds_read_b32 chain %.1f cyc, ds_read_b64+cvt chain %.1f, f64 fma chain %.1f, ds_add_f64 + read %.1f, rcp+add chain %.1f\n", blocks,
           (double)h[0] / iters, (double)h[1] / iters, (double)h[2] / iters, (double)h[3] / iters, (double)h[4] / iters);
  }
  return 0;
}

// Developer tool: determine the operand / result lane layout of v_mfma_f64_16x16x4_f64 on gfx950 empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {   // A [16][4], B [4][16] row-major
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];      // hypothesis: lane holds A[i = l%16][k = l/16]
  const double b = B[(l / 16) * 16 + (l % 16)];     //             lane holds B[k = l/16][j = l%16]
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = sin(i * 1.3) + 0.1 * i; hB[i] = cos(i * 0.7) - 0.05 * i; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, e3 = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const double d = hD[l * 4 + v];
    e1 = fmax(e1, fabs(d - ref[(4 * (l / 16) + v) * 16 + (l % 16)]));   // i = 4*(l/16)+v, j = l%16
    e2 = fmax(e2, fabs(d - ref[(4 * v + (l / 16)) * 16 + (l % 16)]));   // i = 4*v + l/16,  j = l%16
    e3 = fmax(e3, fabs(d - ref[(l % 16) * 16 + 4 * (l / 16) + v]));     // transposed
  }
  printf("layout errors: i=4*(l/16)+v: %.3e   i=4*v+l/16: %.3e   transposed: %.3e\n", e1, e2, e3);
  return 0;
}

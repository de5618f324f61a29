// Register layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction) by one-hot probing: lane la holds
// a = 1, lane lb holds b = 1, everything else 0 -- which lanes see a non-zero result?  (The 16x16x4 form's layout is in the guide
// and in use in csrc/qp_ipm.h; the small form is not documented in this image.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_layout.hip -o tools/micro/mfma_f64_layout && tools/micro/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_probe(unsigned long long *hit) {  // hit[la * 64 + lb] = mask of lanes with d != 0
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0) hit[la * 64 + lb] = m;
    }
}

int main() {
  unsigned long long *d_hit, h[4096];
  CK(hipMalloc(&d_hit, sizeof(h)));
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_hit);
  CK(hipMemcpy(h, d_hit, sizeof(h), hipMemcpyDeviceToHost));
  // For every A lane: the B lanes it meets and the output lane of each meeting
  for (int la = 0; la < 64; ++la) {
    printf("a@%2d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      const unsigned long long m = h[la * 64 + lb];
      if (!m) continue;
      printf("  b@%d->", lb);
      for (int o = 0; o < 64; ++o)
        if (m >> o & 1) printf("%d,", o);
    }
    printf("\n");
  }
  // hypothesis: A[i = l & 3][k = (l >> 2) & 3] of block l >> 4; B[k = (l >> 2) & 3][j = l & 3] of block l >> 4; D[i][j] at ?
  int ok_std = 1;
  for (int la = 0; la < 64 && ok_std; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const int ba = la >> 4, ka = (la >> 2) & 3, ia = la & 3, bb = lb >> 4, kb = (lb >> 2) & 3, jb = lb & 3;
      const bool meet = ba == bb && ka == kb;
      const unsigned long long want = meet ? 1ull << (16 * ba + 4 * ia + jb) : 0ull;
      if (h[la * 64 + lb] != want) { ok_std = 0; break; }
    }
  printf("hypothesis A[i=l&3][k=(l>>2)&3], B[k=(l>>2)&3][j=l&3], D[i=(l>>2)&3][j=l&3], block l>>4: %s\n", ok_std ? "HOLDS" : "does not hold");
  return 0;
}

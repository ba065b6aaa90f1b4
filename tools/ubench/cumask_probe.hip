// Which compute units does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X (8 XCDs x 32 CUs)?
// A grid of one-wave workgroups that spin ~50 us each records (XCC_ID, HW_ID) per workgroup; per mask pattern the histogram of
// XCDs and the number of distinct (xcc, se, cu) triples is printed.   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int G = 2048;
  unsigned* d;
  hipMalloc(&d, G * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, s, d, 5000);   // 50 us at 100 MHz
  hipStreamSynchronize(s);
  std::vector<unsigned> h(G * 2);
  hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  int hist[16] = {0};
  std::set<unsigned long long> cus;
  for (int i = 0; i < G; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    hist[xcc]++;
    cus.insert(((unsigned long long)xcc << 32) | (hw & 0x00001f00u) | ((unsigned long long)((hw >> 13) & 0x7) << 16));   // cu_id bits 8..11, sh 12, se 13..15
  }
  printf("%-28s distinct CUs %3zu  XCD histogram:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %4d", hist[x]);
  printf("\n");
  if (cus.size() <= 40) {   // the physical ids on XCD 0: (se, sh, cu)
    printf("    XCD 0:");
    for (unsigned long long k : cus) if ((k >> 32) == 0) printf(" (se %llu sh %llu cu %llu)", (k >> 16) & 7, (k >> 12) & 1, (k >> 8) & 15);
    printf("\n");
  }
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  std::vector<uint32_t> all(8, 0xffffffffu);
  run("all 256", all);
  for (int w = 0; w < 8; ++w) { std::vector<uint32_t> m(8, 0); m[w] = 0xffffffffu; char nm[64]; snprintf(nm, 64, "word %d (bits %d..%d)", w, 32 * w, 32 * w + 31); run(nm, m); }
  { std::vector<uint32_t> m(8, 0x01010101u); run("every 8th bit (0, 8, ..)", m); }
  { std::vector<uint32_t> m(8, 0x000000ffu); run("bits 0-7 of every word", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xff; run("bits 0-7", m); }
  { std::vector<uint32_t> m(8, 0xfffffffeu); run("all but bit 0 of each word", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xffff; run("bits 0-15", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xff00; run("bits 8-15", m); }
  { std::vector<uint32_t> m(8, 0); m[1] = 0xff; run("bits 32-39", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0x01010101u; run("bits 0, 8, 16, 24", m); }
  return 0;
}

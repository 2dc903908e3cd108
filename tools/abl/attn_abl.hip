// Ablation / timing harness for the attention kernels (no torch): includes attention.hip compiled with -DIA_ABL=n.
// usage: attn_abl B L nh [mode: 0 fwd, 1 bwd] ; prints average time of 20 launches.
#ifndef IA_ATTN_SRC
#define IA_ATTN_SRC "../../item_alignment_amd/csrc/attention.hip"
#endif
#include IA_ATTN_SRC
// the library's partial-sum fold (layernorm.hip), referenced by ia_attn_bwd_bias: not exercised by this harness
int ia_sum_rows_f32(const float*, int, int, float*, int, hipStream_t) { return 0; }
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
  int B = atoi(argv[1]), L = atoi(argv[2]), nh = atoi(argv[3]); int mode = argc > 4 ? atoi(argv[4]) : 0;
  float drop = argc > 5 ? atof(argv[5]) : 0.f;
  int H = nh * 64; size_t T = (size_t)B * L;
  std::vector<uint16_t> h(T * 3 * H);
  uint32_t s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; uint32_t u; std::memcpy(&u, &f, 4); x = u >> 16; }
  void *qkv, *out, *dout, *dqkv; float *lse, *delta; uint8_t* mask;
  hipMalloc(&qkv, T * 3 * H * 2); hipMalloc(&out, T * H * 2); hipMalloc(&dout, T * H * 2); hipMalloc(&dqkv, T * 3 * H * 2);
  hipMalloc(&lse, (size_t)B * nh * L * 4); hipMalloc(&delta, (size_t)B * nh * L * 4); hipMalloc(&mask, T);
  hipMemcpy(qkv, h.data(), T * 3 * H * 2, hipMemcpyHostToDevice);
  hipMemcpy(dout, h.data(), T * H * 2, hipMemcpyHostToDevice);
  hipMemset(mask, 1, T);
  char* p = (char*)qkv; char* g = (char*)dqkv;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() {
    if (mode == 0) return ia_attn_fwd(p, p + 2 * H, p + 4 * H, 3 * H, mask, out, H, lse, B, nh, L, 0.125f, drop, 1, 0);
    return ia_attn_bwd(p, p + 2 * H, p + 4 * H, 3 * H, mask, out, dout, H, lse, delta, g, g + 2 * H, g + 4 * H, 3 * H, B, nh, L, 0.125f, drop, 1, 0);
  };
  ia_attn_fwd(p, p + 2 * H, p + 4 * H, 3 * H, mask, out, H, lse, B, nh, L, 0.125f, drop, 1, 0);
  for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) run();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1000 / 20; double fl = (mode == 0 ? 4.0 : 10.0) * B * nh * (double)L * L * 64;
  printf("abl=%d mode=%d B=%d L=%d nh=%d: %.1f us  %.1f TF/s\n", 0, mode, B, L, nh, us, fl / us * 1e-6);
  return 0;
}

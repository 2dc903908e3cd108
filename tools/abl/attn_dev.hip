// Development harness for the attention kernels (no torch): includes attention.hip, checks the forward (and backward) against an fp32
// host reference on a few (sequence, head) pairs and times the launches.  Kernel variants are picked with IA_ATTN_FWD / IA_ATTN_BWD.
// usage: attn_dev B L nh [mode: 0 fwd, 1 bwd] [drop] [amp: input std * 4] [masked: 0 none, 1 right padding + a hole]
//        [check: 0 time only, 1 compare sampled (sequence, head) pairs with the host reference, N >= 2: also N launches whose whole
//         output is scanned for non-finite / absurd entries -- rare timing-dependent faults do not show up in the sampled check]
#ifndef IA_ATTN_SRC
#define IA_ATTN_SRC "../../item_alignment_amd/csrc/attention.hip"
#endif
#include IA_ATTN_SRC
int ia_sum_rows_f32(const float*, int, int, float*, int, hipStream_t) { return 0; }
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

static float bf2f_h(uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; std::memcpy(&f, &x, 4); return f; }
static uint16_t f2bf_h(float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
  const int B = atoi(argv[1]), L = atoi(argv[2]), nh = atoi(argv[3]);
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const float drop = argc > 5 ? atof(argv[5]) : 0.f;
  const float amp = argc > 6 ? atof(argv[6]) : 1.f;
  const int masked = argc > 7 ? atoi(argv[7]) : 0;
  const int check = argc > 8 ? atoi(argv[8]) : 1;
  const int H = nh * 64; const size_t T = (size_t)B * L;
  std::vector<uint16_t> h(T * 3 * H), hdo(T * H);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (auto& x : h) x = f2bf_h(rnd() * 4.f * amp);           // uniform(-2, 2) * amp: std 1.15 * amp
  for (auto& x : hdo) x = f2bf_h(rnd() * 2.f);
  std::vector<uint8_t> hm(T, 1);
  std::vector<int> lens(B, L);
  if (masked) {
    for (int b = 0; b < B; ++b) {
      lens[b] = std::max(1, L - (int)((b * 37) % (L / 2 + 1)));
      for (int j = lens[b]; j < L; ++j) hm[(size_t)b * L + j] = 0;
      if (lens[b] > 20) hm[(size_t)b * L + 17] = 0;
    }
  }
  void *qkv, *out, *dout, *dqkv; float *lse, *delta; uint8_t* mask;
  hipMalloc(&qkv, T * 3 * H * 2 + (4u << 20)); hipMemset(qkv, 0, T * 3 * H * 2 + (4u << 20)); hipMalloc(&out, T * H * 2); hipMalloc(&dout, T * H * 2); hipMalloc(&dqkv, T * 3 * H * 2);
  hipMalloc(&lse, (size_t)B * nh * L * 4); hipMalloc(&delta, (size_t)B * nh * L * 4); hipMalloc(&mask, T);
  hipMemcpy(qkv, h.data(), T * 3 * H * 2, hipMemcpyHostToDevice);
  hipMemcpy(dout, hdo.data(), T * H * 2, hipMemcpyHostToDevice);
  hipMemcpy(mask, hm.data(), T, hipMemcpyHostToDevice);
  hipMemset(out, 0xFF, T * H * 2); hipMemset(dqkv, 0xFF, T * 3 * H * 2);
  char* p = (char*)qkv; char* g = (char*)dqkv;
  const uint8_t* mk = masked ? mask : nullptr;
  auto fwd = [&]() { return ia_attn_fwd(p, p + 2 * H, p + 4 * H, 3 * H, mk, out, H, lse, B, nh, L, 0.125f, drop, 1, 0); };
  auto bwd = [&]() { return ia_attn_bwd(p, p + 2 * H, p + 4 * H, 3 * H, mk, out, dout, H, lse, delta, g, g + 2 * H, g + 4 * H, 3 * H, B, nh, L, 0.125f, drop, 1, 0); };
  if (fwd()) { printf("fwd launch failed\n"); return 1; }
  if (mode == 1 && bwd()) { printf("bwd launch failed\n"); return 1; }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  const char* fv = getenv("IA_ATTN_FWD"); const char* bv = getenv("IA_ATTN_BWD");
  char tag[64]; snprintf(tag, sizeof tag, "fwd=%s bwd=%s", fv ? fv : "-", bv ? bv : "-");
  if (check && drop == 0.f) {
    std::vector<uint16_t> ho(T * H), hg(T * 3 * H); std::vector<float> hl((size_t)B * nh * L);
    hipMemcpy(ho.data(), out, T * H * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost);
    if (mode == 1) hipMemcpy(hg.data(), dqkv, T * 3 * H * 2, hipMemcpyDeviceToHost);
    double eo = 0, mo = 0, el = 0, eg[3] = {0, 0, 0}, mg[3] = {0, 0, 0};
    const int bs[2] = {0, B - 1}, hs[2] = {0, nh - 1};
    std::vector<float> P((size_t)L * L), dP((size_t)L * L);
    for (int bi = 0; bi < (B > 1 ? 2 : 1); ++bi) for (int hi = 0; hi < (nh > 1 ? 2 : 1); ++hi) {
      const int b = bs[bi], hd = hs[hi];
      auto Q = [&](int i, int d) { return bf2f_h(h[((size_t)b * L + i) * 3 * H + hd * 64 + d]); };
      auto K = [&](int i, int d) { return bf2f_h(h[((size_t)b * L + i) * 3 * H + H + hd * 64 + d]); };
      auto V = [&](int i, int d) { return bf2f_h(h[((size_t)b * L + i) * 3 * H + 2 * H + hd * 64 + d]); };
      auto DO = [&](int i, int d) { return bf2f_h(hdo[((size_t)b * L + i) * H + hd * 64 + d]); };
      std::vector<float> O((size_t)L * 64, 0.f), dlt(L, 0.f);
      for (int i = 0; i < L; ++i) {
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) {
          float a = 0; for (int d = 0; d < 64; ++d) a += Q(i, d) * K(j, d);
          a = hm[(size_t)b * L + j] ? a * 0.125f : -INFINITY;
          P[(size_t)i * L + j] = a; m = std::max(m, a);
        }
        double l = 0;
        for (int j = 0; j < L; ++j) { float e = std::exp(P[(size_t)i * L + j] - m); P[(size_t)i * L + j] = e; l += e; }
        for (int j = 0; j < L; ++j) P[(size_t)i * L + j] /= (float)l;
        for (int d = 0; d < 64; ++d) { float a = 0; for (int j = 0; j < L; ++j) a += P[(size_t)i * L + j] * V(j, d); O[(size_t)i * 64 + d] = a; }
        const double lse_ref = (m + std::log(l)) * 1.4426950408889634;
        el = std::max(el, std::fabs(lse_ref - hl[((size_t)b * nh + hd) * L + i]));
        for (int d = 0; d < 64; ++d) {
          const float got = bf2f_h(ho[((size_t)b * L + i) * H + hd * 64 + d]);
          eo = std::max(eo, (double)std::fabs(got - O[(size_t)i * 64 + d])); mo = std::max(mo, (double)std::fabs(O[(size_t)i * 64 + d]));
        }
      }
      if (mode == 1) {
        for (int i = 0; i < L; ++i) { float a = 0; for (int d = 0; d < 64; ++d) a += DO(i, d) * O[(size_t)i * 64 + d]; dlt[i] = a; }
        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) {
          float a = 0; for (int d = 0; d < 64; ++d) a += DO(i, d) * V(j, d);
          dP[(size_t)i * L + j] = P[(size_t)i * L + j] * (a - dlt[i]);       // dS
        }
        for (int i = 0; i < L; ++i) for (int d = 0; d < 64; ++d) {
          float dq = 0, dk = 0, dv = 0;
          for (int j = 0; j < L; ++j) { dq += dP[(size_t)i * L + j] * K(j, d); dk += dP[(size_t)j * L + i] * Q(j, d); dv += P[(size_t)j * L + i] * DO(j, d); }
          const float want[3] = {dq * 0.125f, dk * 0.125f, dv};
          for (int c = 0; c < 3; ++c) {
            const float got = bf2f_h(hg[((size_t)b * L + i) * 3 * H + c * H + hd * 64 + d]);
            eg[c] = std::max(eg[c], (double)std::fabs(got - want[c])); mg[c] = std::max(mg[c], (double)std::fabs(want[c]));
          }
        }
      }
    }
    printf("[%s] B=%d L=%d nh=%d amp=%.1f masked=%d: O rel err %.2e  lse abs err %.2e", tag, B, L, nh, amp, masked, eo / mo, el);
    if (mode == 1) printf("  dq %.2e dk %.2e dv %.2e", eg[0] / mg[0], eg[1] / mg[1], eg[2] / mg[2]);
    printf("\n");
  }
  for (int i = 0; i < 3; ++i) { if (mode == 0) fwd(); else bwd(); }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, tot = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) { if (mode == 0) fwd(); else bwd(); }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms); tot += ms;
  }
  const double us = tot * 1000 / 60, usb = best * 1000 / 20; const double fl = (mode == 0 ? 4.0 : 10.0) * B * nh * (double)L * L * 64;
  for (int rep = 0; rep < (check >= 3 ? check : check >= 2 ? 1 : 0); ++rep) {   // whole-tensor scan of a launch: non-finite or absurd entries
    std::vector<uint16_t> hg(mode == 1 ? T * 3 * H : T * H);
    if (mode == 0) fwd(); else bwd();
    hipDeviceSynchronize();
    hipMemcpy(hg.data(), mode == 1 ? dqkv : out, hg.size() * 2, hipMemcpyDeviceToHost);
    const size_t ld = mode == 1 ? 3 * H : H; size_t nbad = 0;
    for (size_t i = 0; i < hg.size(); ++i) {
      const float v = bf2f_h(hg[i]);
      if (!(std::fabs(v) < 1e3f)) {
        if (nbad < 2) printf("   bad %g at row %zu (b %zu, i %zu) col %zu (part %zu head %zu d %zu)\n", v, i / ld, i / ld / L, i / ld % L, i % ld, i % ld / H, i % H / 64, i % 64);
        ++nbad;
      }
    }
    printf("[%s] scan: %zu bad of %zu\n", tag, nbad, hg.size());
  }
  printf("[%s] mode=%d B=%d L=%d nh=%d drop=%.2f masked=%d: %.1f us avg (%.1f best)  %.1f TF/s\n", tag, mode, B, L, nh, drop, masked, us, usb, fl / us * 1e-6);
  return 0;
}

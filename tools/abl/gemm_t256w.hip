// PARKED EXPERIMENT (round 2, versions 1-3) -- not compiled into the library.  SUPERSEDED: the t256w namespace is back in csrc/gemm.hip
// with a fourth main loop (whole k-tile of fragments in registers, buffer free after half an iteration, DMA pieces spread behind the
// MFMAs of the second half, counted vmcnt(8)) that beats T256 by 5-10 % and serves every GEMM without a VALU-heavy epilogue.
//  This is the `t256w` namespace that sat in csrc/gemm.hip behind
// IA_GEMM_WIDE=1 (it also needs `raw_rsrc_at` / `i32x4` from below): the 256 x 256 x 64 GEMM with ONE wave per SIMD owning a
// 128 x 128 part (256 accumulators in AGPRs), fragments of k-step s+1 requested before the MFMAs of step s, one barrier per
// k-tile.  k-tiles 0 and 1 of an output tile arrive by LDS-DMA (issued before the previous tile's epilogue); inside the loop
// k-tiles travel global -> VGPR -> LDS through two register sets of 16 x 16 bytes per lane (asm buffer loads, explicit counted
// vmcnt waits: 31 younger loads in the steady state), i.e. two more k-tiles in flight than the LDS double buffer holds -- the
// "prefetch global read 2" scheme of Tensile's 256x256x64 kernels.  Steady state and the last four k-tiles are straight-line
// code: any branch around the accumulator updates makes hipcc copy all 256 of them.
// Results (tools/abl/gemm_wide.py, gemm_wide_abl.py; all epilogues and operand forms, parity green in all three versions):
//   v1  DMA inside the loop (pieces of k-tile u+2 between the MFMAs of the last k-step of tile u): 8192^3 1072 TFLOP/s where T256 gets
//       1086 on the same box; with the fetches sent out of range (IA_GEMM_DBG=2) 1466-1549 against T256's 1337 -- the 128 x 128
//       wave tile does remove T256's LDS-port ceiling (192 + 64 KiB of LDS traffic per k-tile -> 128 + 64 KiB);
//   v2  register staging with builtin loads: hipcc waits vmcnt(0) in front of many LDS writes -> 924-1041;
//   v3  (this file) asm loads + counted waits: 1043 (T256 1098 same box; K = 4096 shapes 1037 vs 1161), out-of-range fetches 1381.
//   So a k-tile's worth of extra lead (v3 has ~2.8 us) does not buy anything: it is not fetch latency.  With the MFMAs compiled out
//   (-DIA_GEMM_NOMATH, v1) the fetch stream alone takes 1.27 us per 64 KiB k-tile per CU (0.79 us when every k-tile re-reads
//   k-tile 0), the no-fetch MFMA loop 1.4 us, both together 2.0 us: real data arriving (VGPR / LDS write-back, L2 and fabric
//   activity under the power cap) slows the MFMA stream itself.  hipBLASLt's hand-written 256x256x64 kernel (16x16x32 MFMAs) gets
//   1343-1541 on the same shapes and boxes, so the ceiling is not physical; what it does differently was not found.
// To revive: paste the namespace back after `}  // namespace t256` in csrc/gemm.hip and dispatch to t256w::gemm_kernel (256 threads,
// t256::LDS_BYTES of dynamic LDS, k loops of at least 5 k-tiles) in launch().

typedef __attribute__((ext_vector_type(4))) int i32x4;
// a buffer window (see rsrc_at in gemm.hip) as four raw descriptor words in SGPRs, for asm buffer loads
IA_DEV i32x4 raw_rsrc_at(const bf16* base, uint64_t total_bytes, uint64_t origin) {
  const uint64_t ob = origin * 2, rem = total_bytes > ob ? total_bytes - ob : 0;
  const uint64_t addr = (uint64_t)(uintptr_t)(base + origin);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((addr >> 32) & 0xFFFFu));
  r[2] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(rem < 0x7FFFFFF0ull ? rem : 0x7FFFFFF0ull));
  r[3] = 0x00020000;
  return r;
}

// ============================================================================== T256W (4 waves, 128 x 128 per wave, 32x32x16)
// Same 256 x 256 x 64 block tile, LDS layout, DMA pattern, tile order and epilogue as T256, but ONE wave per SIMD owning a
// 128 x 128 part (256 accumulator registers).  Why: T256's 128 x 64 wave tiles read 24 fragments per 32 MFMAs, i.e. 192 KiB of
// LDS reads + 64 KiB of DMA writes per k-tile against 2048 MFMA cycles per SIMD -- at 128 B/clk the LDS port is as busy as the
// matrix pipe, and every conflict or bubble shows (MFMA-only loop 1630 TFLOP/s, + fragment reads 1528, + DMA 1282).  A 128 x 128
// wave tile reads 8 fragments per 16 MFMAs: 128 + 64 KiB per k-tile, 75 % of the MFMA time.  With one wave per SIMD nothing else
// hides latency, so the wave software-pipelines itself: the fragments of k-step s+1 are requested (asm LDS reads into the other
// register set) before the 16 MFMAs of step s are issued, the MFMAs of a k-tile's last step run after the barrier that frees its
// buffer (under the next tile's first fragment reads), and the 16 DMA pieces of k-tile u+2 are slipped between those MFMAs.
// One workgroup barrier per k-tile.
namespace t256w {
#ifdef IA_GEMM_NOMATH
constexpr bool NOMATH = true;
#else
constexpr bool NOMATH = false;
#endif
using t256::BM;
using t256::BN;
using t256::TILE_BYTES;
using t256::STAGE_BYTES;
using t256::LDS_BYTES;

// One operand's four fragments of a k-step.  A k-strided operand's fragment arrives as two transpose reads: the halves are kept
// apart until the wait (the wait asm ties the RAW read destinations; assembling the 128-bit value earlier could be scheduled as
// register copies in front of the wait).
template <bool KS> struct Op;
template <> struct Op<false> { bf16x8 v[4]; };
template <> struct Op<true> { s16x4 lo[4], hi[4]; };
IA_DEV bf16x8 frag_of(const Op<false>& o, int j) { return o.v[j]; }
IA_DEV bf16x8 frag_of(const Op<true>& o, int j) {
  s16x8 r = {o.lo[j][0], o.lo[j][1], o.lo[j][2], o.lo[j][3], o.hi[j][0], o.hi[j][1], o.hi[j][2], o.hi[j][3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int IMM>
IA_DEV bf16x8 rd128(uint32_t addr) {
  bf16x8 d;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
  return d;
}
template <int IMM>
IA_DEV s16x4 rd_tr(uint32_t addr) {
  s16x4 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
  return d;
}

// Per-lane LDS byte offsets of one operand's fragments inside a k-tile buffer.  k-contiguous tile ([256 rows][64 k], chunk XOR
// (row>>1)&7): fragment (j, ks) of rows x0 + j*32 + li sits at base[ks] + j*4096 -- the XOR only depends on the lane and ks.
// k-strided tile ([64 k][256 x], 32-byte slot XOR (k&3)<<2): fragment (j, ks) of columns x0 + j*32.. sits at base[j] + ks*8192 --
// the XOR lands on the bits j occupies, so it is folded per j.  (x0 = 0 or 128: the wave's half of the tile.)
template <bool KS>
IA_DEV void frag_bases(uint32_t (&base)[4], uint32_t tile_addr, int x0, int lane, int nperm_row) {
  if (!KS) {
    const int hh = lane >> 5;
    const int row = x0 + nperm_row;                       // li, or the permuted row of a k-contiguous B operand
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) base[ks] = tile_addr + row * 128 + ((((ks * 2 + hh) ^ ((row >> 1) & 7))) << 4);
  } else {
    const int p = lane & 15, G = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * (G >> 1) + (p >> 2);            // + ks*16 through the immediate
      const int col = x0 + j * 32 + 16 * (G & 1) + (p & 3) * 4;
      base[j] = tile_addr + row * 512 + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2;
    }
  }
}

template <int S>
IA_DEV void read_operand(Op<false>& f, const uint32_t (&base)[4], uint32_t bufoff) {
  const uint32_t a = base[S] + bufoff;
  f.v[0] = rd128<0>(a); f.v[1] = rd128<4096>(a); f.v[2] = rd128<8192>(a); f.v[3] = rd128<12288>(a);
}
template <int S>
IA_DEV void read_operand(Op<true>& f, const uint32_t (&base)[4], uint32_t bufoff) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {             // two transpose reads = the 8 k values of one 32x32x16 operand fragment
    f.lo[j] = rd_tr<S * 8192>(base[j] + bufoff);
    f.hi[j] = rd_tr<S * 8192 + 4 * 512>(base[j] + bufoff);
  }
}

// wait until at most N LDS operations are outstanding and tie the read destinations to the wait (the MFMAs that consume them
// cannot be scheduled above it)
template <int N> IA_DEV void tie(Op<false>& o) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]) : "n"(N));
}
template <int N> IA_DEV void tie(Op<true>& o) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(o.lo[0]), "+v"(o.lo[1]), "+v"(o.lo[2]), "+v"(o.lo[3]), "+v"(o.hi[0]), "+v"(o.hi[1]), "+v"(o.hi[2]), "+v"(o.hi[3]) : "n"(N));
}

template <bool AKS, bool BKS, int PEND>
IA_DEV void main_loop(const GemmArgs& p, char* smem, f32x16 (&acc)[4][4], __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, i32x4 rawA, i32x4 rawB, int xa, int xb,
                      int kt0, int ktaA0, int ktaB0, int n_tiles, int nk_all, int wm, int wn, int wave, int lane, bool prologue_only,
                      bool stores_in_flight) {
  const int li = lane & 31;
  const int gt = wave * 64 + lane;                  // thread index inside the workgroup (0..255)
  // ---- DMA: 8 pieces per operand and k-tile, one lane offset per operand (see t256::main_loop)
  uint32_t voffA, stepA, voffB, stepB;
  if (!AKS) { const int row = gt >> 3; voffA = (uint32_t)(((xa + row) * p.lda + (((gt & 7) ^ ((row >> 1) & 7)) * 8)) * 2); stepA = (uint32_t)(32 * p.lda * 2); }
  else { const int row = gt >> 5; voffA = (uint32_t)((row * p.lda + xa + (((gt & 31) ^ ((row & 3) << 2)) * 8)) * 2); stepA = (uint32_t)(8 * p.lda * 2); }
  if (!BKS) { const int row = gt >> 3; voffB = (uint32_t)(((xb + row) * p.ldb + (((gt & 7) ^ ((row >> 1) & 7)) * 8)) * 2); stepB = (uint32_t)(32 * p.ldb * 2); }
  else { const int row = gt >> 5; voffB = (uint32_t)((row * p.ldb + xb + (((gt & 31) ^ ((row & 3) << 2)) * 8)) * 2); stepB = (uint32_t)(8 * p.ldb * 2); }
  const uint32_t kstepA = AKS ? (uint32_t)(BK * p.lda * 2) : (uint32_t)(BK * 2), kstepB = BKS ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);
  char* const my_part = smem + wave * 1024;
  const bool dma_on = !(p.dbg & 2);

  // piece i (0..7: A, 8..15: B) of k-tile u -> buffer u & 1.  Branch-free: lanes past K (the ragged last k-tile) and, with
  // valid == false, the whole piece are sent out of range -- they write zeros into a buffer nobody reads any more -- so the MFMA
  // stream of the loop has no control flow around it (a branch around accumulator updates costs a copy of all 256 of them).
  auto dma_piece = [&](int u, int i, bool valid) {
    const bool isB = i >= 8;
    const int j = i & 7;
    char* dst = my_part + (isB ? TILE_BYTES : 0) + (u & 1) * 2 * TILE_BYTES + j * 4096;
    const int kt = kt0 + u, kta = (p.dbg & 4) ? 0 : (isB ? ktaB0 : ktaA0) + u;      // dbg 4: every k-tile re-fetches k-tile 0 (cache-resident)
    const uint32_t voff = isB ? voffB : voffA, kstep = isB ? kstepB : kstepA, pstep = isB ? stepB : stepA;
    const bool ks = isB ? BKS : AKS;
    const int k = ks ? kt * BK + j * 8 + (gt >> 5) : kt * BK + ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;
    const uint32_t off = (valid && k < p.K) ? voff + (uint32_t)(kta * kstep + j * pstep) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rsB : rsA, IA_LDS(dst), 16, off, 0, 0, 0);
  };
  auto dma_tile = [&](int u, bool valid) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dma_piece(u, i, valid);
  };
  // The same piece fetched into registers instead: inside the loop k-tiles travel global -> VGPR -> LDS.  The LDS holds two k-tiles
  // (128 KiB) and a DMA can only start once its target buffer has been read out, i.e. at most ONE k-tile period (~1.4 us) before
  // its data is needed -- less than the L2 / fabric round trip under load, and the fetch path (~13 TB/s over the chip, measured
  // with the MFMAs compiled out) is nearly as busy as the matrix pipe, so every such stall is lost for good.  Two register sets
  // of 16 x 16 bytes per lane keep two MORE k-tiles in flight (the wave has 256 VGPRs beside its 256 accumulators).
  // asm on purpose: hipcc cannot count loads that stay in flight across loop iterations and falls back to vmcnt(0) in front of the
  // LDS writes (measured: the builtin form of this loop was slower than the plain DMA loop); here every wait is an explicit counted one
  auto load_piece = [&](u32x4& dst, int u, int i) {           // u >= n_tiles (the zero pad tile of an odd count): out of range -> zeros
    const bool isB = i >= 8;
    const int j = i & 7;
    const int kt = kt0 + u, kta = (p.dbg & 4) ? 0 : (isB ? ktaB0 : ktaA0) + u;
    const uint32_t voff = isB ? voffB : voffA, kstep = isB ? kstepB : kstepA, pstep = isB ? stepB : stepA;
    const bool ks = isB ? BKS : AKS;
    const int k = ks ? kt * BK + j * 8 + (gt >> 5) : kt * BK + ((gt & 7) ^ (((gt >> 3) >> 1) & 7)) * 8;
    const uint32_t off = (u < n_tiles && dma_on && k < p.K) ? voff + (uint32_t)(kta * kstep + j * pstep) : OOB;
    if (isB) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(dst) : "v"(off), "s"(rawB));
    else     asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(dst) : "v"(off), "s"(rawA));
  };
  const uint32_t my_lds = ia_lds_addr(smem) + wave * 1024 + lane * 16;
  auto store_piece = [&](uint32_t bufoff, int i, u32x4 v) {        // where DMA piece i of the k-tile would have landed
    const uint32_t addr = my_lds + bufoff + (uint32_t)((i >= 8 ? TILE_BYTES : 0) + (i & 7) * 4096);
    asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory");
  };

  if (prologue_only) {       // called ahead of time (before the previous tile's epilogue): just start the first two k-tiles
    dma_tile(0, true);
    dma_tile(1, n_tiles > 1);      // (zero-filled when it does not exist: the loop below runs an even number of k-tiles)
    return;
  }
  // the prologue DMA of this tile.  After a full-tile epilogue exactly PEND store instructions were issued behind it and
  // may stay in flight (vmcnt retires in order: at most PEND outstanding <=> every older DMA piece has landed).
  if (stores_in_flight) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PEND < 60 ? PEND : 60) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- fragment addresses
  const int nperm = ((li >> 2) & 1) * 16 + (li >> 3) * 4 + (li & 3);      // B row <-> n so that a lane ends up with 16 consecutive columns
  uint32_t baseA[4], baseB[4];
  const uint32_t smem_addr = ia_lds_addr(smem);
  frag_bases<AKS>(baseA, smem_addr, wm * 128, lane, li);
  frag_bases<BKS>(baseB, smem_addr + TILE_BYTES, wn * 128, lane, nperm);
  constexpr int NR = ((AKS ? 8 : 4) + (BKS ? 8 : 4)) > 15 ? 15 : ((AKS ? 8 : 4) + (BKS ? 8 : 4));    // LDS ops of one fragment set

  Op<AKS> a0, a1;
  Op<BKS> b0, b1;
  read_operand<0>(a0, baseA, 0u);
  read_operand<0>(b0, baseB, 0u);

  // k-tiles 0 and 1 arrive by DMA (issued before the previous tile's epilogue), k-tiles 2 .. n_even-1 through the register sets
  // (n_even = n_tiles rounded up to even: the loop has no control flow around the accumulators, a pad tile is all zeros).
  const int n_even = (n_tiles + 1) & ~1;
  u32x4 R0[16], R1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { R0[i] = u32x4{0u, 0u, 0u, 0u}; R1[i] = u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
  for (int i = 0; i < 16; ++i) load_piece(R0[i], 2, i);
#pragma unroll
  for (int i = 0; i < 16; ++i) load_piece(R1[i], 3, i);

  // the 16 MFMAs of one k-step.  MODE > 0: slot i between them waits for piece i of the register set R (k-tile u2 = u + 2, requested
  // two k-tiles ago) and hands it to the LDS buffer that just became free; MODE 1 also re-arms the registers with the same piece of
  // k-tile u2 + 2.  VMEM retires in order, so "piece i has landed" = at most (15 - i) + 16 + i = 31 younger loads outstanding in the
  // steady state (the rest of this set, the whole other set, the i re-armed so far); MODE 2 / 3 = the last two register tiles of
  // the k loop (no re-arming: 31 - i, then 15 - i).  No control flow anywhere near the accumulators.
  auto mfmas = [&](const Op<AKS>& fa, const Op<BKS>& fb, auto MODE, u32x4 (&R)[16], uint32_t bufoff, int u2) {
    constexpr int mode = decltype(MODE)::value;
    bf16x8 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { va[j] = frag_of(fa, j); vb[j] = frag_of(fb, j); }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int i = mi * 4 + ni;
        if (mode == 1) asm volatile("s_waitcnt vmcnt(31)" : "+v"(R[i]));
        if (mode == 2) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R[i]) : "n"(31 - i));
        if (mode == 3) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R[i]) : "n"(15 - i));
        if (mode >= 1) store_piece(bufoff, i, R[i]);
        if (mode == 1) load_piece(R[i], u2 + 2, i);
        if (NOMATH) continue;       // ablation build (-DIA_GEMM_NOMATH): transfers + waits + barriers only
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[ni], va[mi], acc[mi][ni], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);      // the MFMAs stay in their step (the waits of the next step are volatile asm, MFMAs are not)
  };
  using M0 = std::integral_constant<int, 0>;

  // one k-tile: buffer u & 1.  R = the register set that holds k-tile u + 2.
  auto ktile = [&](int u, u32x4 (&R)[16], auto MODE) {
    const uint32_t bo = (uint32_t)(u & 1) * 2 * TILE_BYTES, bn = bo ^ (2 * TILE_BYTES);
    // step 0: request step 1, compute step 0
    read_operand<1>(a1, baseA, bo); read_operand<1>(b1, baseB, bo);
    tie<NR>(a0); tie<NR>(b0);
    mfmas(a0, b0, M0{}, R, 0u, 0);
    // step 1
    read_operand<2>(a0, baseA, bo); read_operand<2>(b0, baseB, bo);
    tie<NR>(a1); tie<NR>(b1);
    mfmas(a1, b1, M0{}, R, 0u, 0);
    // step 2
    read_operand<3>(a1, baseA, bo); read_operand<3>(b1, baseB, bo);
    tie<NR>(a0); tie<NR>(b0);
    mfmas(a0, b0, M0{}, R, 0u, 0);
    // step 3: every fragment of this k-tile is in registers (and this wave's LDS writes of k-tile u+1 have landed): once all waves
    // are here the buffer is free for k-tile u+2 and k-tile u+1 is complete
    tie<0>(a1); tie<0>(b1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(p.dbg & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_operand<0>(a0, baseA, bn); read_operand<0>(b0, baseB, bn);       // (past the last k-tile: unused)
    mfmas(a1, b1, MODE, R, bo, u + 2);
  };
  // n_even >= 6 (the launcher sends shorter k loops to the other kernel): steady state, then the last four k-tiles in straight-line code
  int u = 0;
  do {                        // n_even >= 6: at least one trip (no merge of "loop ran" / "loop skipped" accumulators)
    ktile(u, R0, std::integral_constant<int, 1>{});
    ktile(u + 1, R1, std::integral_constant<int, 1>{});
    u += 2;
  } while (u + 4 < n_even);
  ktile(u, R0, std::integral_constant<int, 2>{});
  ktile(u + 1, R1, std::integral_constant<int, 3>{});
  ktile(u + 2, R0, M0{});
  ktile(u + 3, R1, M0{});
}

// the T256 epilogue (see t256::gemm_kernel) for one 128 x 64 half (NH = 0 / 1) of the wave's 128 x 128 part
template <int EPI, bool OUTF32, bool BKS, int NH>
IA_DEV void drain_half(const GemmArgs& p, f32x16 (&acc)[4][4], int m0, int n0, char* stg, int lane_e, bool full) {
  const int hh = lane_e >> 5, li = lane_e & 31;
  const int wrow = li & 15, rrow = lane_e >> 3, c8 = lane_e & 7;
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ADD;
  constexpr bool HAS_AUX = EPI == EPI_ADD || EPI == EPI_BIAS_ADD || EPI == EPI_DGELU || EPI == EPI_DGELU_CS;
  constexpr int AHEAD = 4;
  auto drain = [&](auto PREFETCHED) {
    constexpr bool PRE = decltype(PREFETCHED)::value;
    f32x4 pb0, pb1;
    float cs[8];
    if (EPI == EPI_DGELU_CS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mov_b32 %0, 0" : "=v"(cs[j]));
    }
    bf16x8 ax[AHEAD + 1];
    auto aux_of = [&](int c) {
      const int row = m0 + (c >> 2) * 32 + ((c >> 1) & 1) * 16 + (c & 1) * 8 + rrow;
      return *reinterpret_cast<const bf16x8*>(p.aux + (size_t)row * p.ldaux + n0 + c8 * 8);
    };
    if (PRE && HAS_BIAS) { pb0 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8); pb1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c8 * 8 + 4); }
    if (PRE && HAS_AUX) {
#pragma unroll
      for (int c = 0; c < AHEAD; ++c) ax[c] = aux_of(c);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int h16 = 0; h16 < 2; ++h16) {
        if ((li >> 4) == h16) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int chunk = (ni * 32 + (BKS ? rg * 8 + hh * 4 : hh * 16 + rg * 4)) >> 2;
              const f32x16& a = acc[mi][NH * 2 + ni];
              const f32x4 v = {a[rg * 4], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3]};
              *reinterpret_cast<f32x4*>(stg + wrow * 256 + ((chunk ^ wrow) << 4)) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + rrow;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8) ^ row) << 4));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * c8 + 1) ^ row) << 4));
          const int m = m0 + mi * 32 + h16 * 16 + row, n = n0 + c8 * 8;
          const int c = mi * 4 + h16 * 2 + it;
          if (PRE) {
            if (HAS_AUX && c + AHEAD < 16) {
              ax[(c + AHEAD) % (AHEAD + 1)] = aux_of(c + AHEAD);
              asm volatile("" ::: "memory");
            }
            if (!(p.dbg & 64)) epi_store8<EPI, OUTF32, true>(p, m, n, lo, hi, pb0, pb1, ax[c % (AHEAD + 1)], cs);
          } else {
            if (m < p.M && n < p.N && !(p.dbg & 64)) epi_store8<EPI, OUTF32, false>(p, m, n, lo, hi, pb0, pb1, ax[0], cs);
          }
          if (p.dbg & 64) asm volatile("" : : "v"(lo), "v"(hi));
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (EPI == EPI_DGELU_CS) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = cs[r];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
        {
          const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
          v = __builtin_bit_cast(float, sw[0]) + __builtin_bit_cast(float, sw[1]);
        }
        {
          const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
          v = __builtin_bit_cast(float, sw[0]) + __builtin_bit_cast(float, sw[1]);
        }
        cs[r] = v;
      }
      if (rrow == 0 && n0 + c8 * 8 < p.N) {
        float* dst = p.csum_part + (size_t)(m0 >> 7) * p.N + n0 + c8 * 8;
        gstore16(dst, f32x4{cs[0], cs[1], cs[2], cs[3]});
        gstore16(dst + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
      }
    }
  };
  if (full && (HAS_BIAS || HAS_AUX) && !(p.dbg & 256)) drain(std::true_type{}); else drain(std::false_type{});
}

template <bool AKS, bool BKS, int EPI, bool OUTF32>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves, 128 x 128 each, one per SIMD
  const int nk_all = (p.K + BK - 1) / BK;
  const int total_tiles = p.tiles_m * p.tiles_n;
  int first_tile = blockIdx.x;
  bool ordered = false;
  p.split_id = 0;
  if (p.splits > 1) {                  // 1-D grid of tiles x splits, XCD-aware with the split outermost (see t256::gemm_kernel)
    const int w = xcd_chunk(blockIdx.x, gridDim.x);
    p.split_id = w / total_tiles;
    first_tile = w % total_tiles;
    ordered = true;
  }
  const int kt0 = p.split_id * p.nk_per_split;
  const int n_tiles = min(nk_all, kt0 + p.nk_per_split) - kt0;
  constexpr int PEND = 2 * (16 * epi_stores_per_call<EPI, OUTF32>() + epi_extra_stores<EPI>());   // store instructions of one full-tile epilogue, per wave
  auto coords = [&](int tile, int& bm, int& bn) {
    if (ordered) tile_of_order(p, tile, bm, bn); else tile_of_index(p, tile, total_tiles, bm, bn);
  };
  auto run = [&](int tile, bool prologue_only, f32x16 (&acc)[4][4], bool stores_in_flight) {
    int bm, bn;
    coords(tile, bm, bn);
    int lane = lane0;
    asm volatile("" : "+v"(lane));      // keep per-lane address arithmetic from being hoisted across the tile loop
    const uint64_t oa = (uint64_t)(AKS ? kt0 * BK : bm * BM) * p.lda, ob = (uint64_t)(BKS ? kt0 * BK : bn * BN) * p.ldb;
    main_loop<AKS, BKS, PEND>(p, smem, acc, rsrc_at(p.A, p.a_bytes, oa), rsrc_at(p.B, p.b_bytes, ob), raw_rsrc_at(p.A, p.a_bytes, oa),
                              raw_rsrc_at(p.B, p.b_bytes, ob), AKS ? bm * BM : 0, BKS ? bn * BN : 0, kt0,
                              AKS ? 0 : kt0, BKS ? 0 : kt0, n_tiles, nk_all, wm, wn, wave, lane, prologue_only, stores_in_flight);
  };

  f32x16 acc[4][4];
  int tile = first_tile;
  run(tile, true, acc, false);
  bool stores_in_flight = false;
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    run(tile, false, acc, stores_in_flight);
    const int next = ordered ? total_tiles : tile + gridDim.x;
    if (next < total_tiles) run(next, true, acc, false);      // the next tile's first two k-tiles travel under this epilogue

    int bm, bn;
    coords(tile, bm, bn);
    const int m0 = bm * BM + wm * 128, n0 = bn * BN + wn * 128;
    int lane_e = lane0;
    asm volatile("" : "+v"(lane_e));
    char* stg = smem + 2 * 2 * TILE_BYTES + wave * STAGE_BYTES;
    const bool full = m0 + 128 <= p.M && n0 + 128 <= p.N;      // both halves inside C: the store count of the tile is exact
    if (!(p.dbg & 32)) {
      drain_half<EPI, OUTF32, BKS, 0>(p, acc, m0, n0, stg, lane_e, full || (m0 + 128 <= p.M && n0 + 64 <= p.N));
      drain_half<EPI, OUTF32, BKS, 1>(p, acc, m0, n0 + 64, stg, lane_e, full);
    }
    if (next >= total_tiles) break;
    stores_in_flight = !(p.dbg & 96) && full;
    if (!stores_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile = next;
  }
}
}  // namespace t256w

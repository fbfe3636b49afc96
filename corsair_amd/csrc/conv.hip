// Sparse convolution forward (output-stationary gather -> f32 MFMA -> fused epilogue) and the
// small dense row ops of the ResUNet (affine/ReLU/residual, row L2 normalise, per-sample max).
//
// Replaces MinkowskiConvolution / MinkowskiConvolutionTranspose / MinkowskiBatchNorm(eval) /
// MEF.relu / SparseTensor.__iadd__ as used by model/resunet.py:207-280 and
// model/residual_block.py:60-73 of the reference.
//
// Kernel design (gfx950):
//   * one workgroup (4 waves) owns a tile of TM output rows x TN output channels and walks the
//     reduction dimension in the canonical order (k = 0..26 outer, ci ascending inner), so every
//     output element is ONE f32 fma chain in that order -- v_mfma_f32_32x32x2_f32 is exactly such
//     a chain, which makes the result bit-identical to the CPU oracle and run-to-run stable
//     (no atomics, no scatter).
//   * per 32-channel chunk the tile's input rows are gathered with 16-byte coalesced loads
//     (8 lanes cover one 128-B row segment) into LDS (row pitch 33 dwords: conflict-free column
//     reads for the MFMA A operand), the weight slab [32, TN] is staged next to it; the loads of
//     chunk c+1 are issued before the MFMAs of chunk c.
//   * the neighbour table of the tile ([TM,27] int32) is read once into LDS; offsets no row of
//     the tile uses are skipped (exact: they would add zeros).  Tiles are formed over the kernel
//     map's row list, which orders output rows by their 27-bit neighbour-presence mask, so the rows
//     of a tile share most of their absent offsets (for transposed maps: the parity class, <= 8
//     live offsets of 27) and the skip removes most of the zero work.
//   * epilogue (BN affine / bias, residual add, ReLU) is applied to the accumulators and written
//     with an arbitrary leading dimension so decoder outputs land directly in the concat buffer.
#include "common.h"
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>
#include <algorithm>

namespace cs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float epilogue(float v, int c, const float* __restrict__ scale,
                                          const float* __restrict__ shift, const float* res_row,
                                          int relu) {
  if (scale)
    v = __fmaf_rn(v, scale[c], shift[c]);
  else if (shift)
    v = v + shift[c];
  if (res_row) v = v + res_row[c];
  if (relu) v = fmaxf(v, 0.0f);
  return v;
}

template <int WM, int WN, int NT>
__global__ __launch_bounds__(256) void k_conv_mfma(
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowlist, int kvol, int64_t n_in,
    int64_t n_out, const float* __restrict__ in, int ld_in, int cin, const float* __restrict__ w,
    int cout,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, int ld_res, int relu, float* __restrict__ out,
    int ld_out) {
  constexpr int TM = 32 * WM;
  constexpr int TN = 32 * NT * WN;
  constexpr int KC = 32;
  constexpr int APITCH = KC + 1;
  constexpr int A_PER_THREAD = TM / 32;        // float4 gathers per thread per chunk
  constexpr int B_PER_THREAD = (KC * TN / 4) / 256;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(B_PER_THREAD >= 1, "tile too small");

  __shared__ float A_lds[TM * APITCH];
  __shared__ float B_lds[KC * TN];
  __shared__ int32_t nbr_lds[TM * 27];
  __shared__ int32_t orow_lds[TM];   // output row of tile slot r (rows are visited in rowlist order)
  __shared__ unsigned kmask_lds;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;

  if (tid == 0) kmask_lds = 0;
  __syncthreads();
  {
    unsigned local_mask = 0;
    for (int i = tid; i < TM * kvol; i += 256) {
      int r = i / kvol, k = i - r * kvol;
      int64_t o = row0 + r;
      int32_t v = -1;
      if (o < n_out) {
        if (rowlist) o = rowlist[o];
        v = nbr ? nbr[o * kvol + k] : (int32_t)o;
        if (k == 0) orow_lds[r] = (int32_t)o;
      } else if (k == 0) {
        orow_lds[r] = -1;
      }
      nbr_lds[r * 27 + k] = v;
      if (v >= 0) local_mask |= 1u << k;
    }
    if (local_mask) atomicOr(&kmask_lds, local_mask);
  }
  __syncthreads();
  const unsigned kmask = kmask_lds;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  const int cchunks = cin / KC;
  const int a_r = tid >> 3;   // 0..31
  const int a_c4 = tid & 7;   // float4 column within the 32-channel chunk

  float4 a_reg[A_PER_THREAD];
  float4 b_reg[B_PER_THREAD];

  auto load_chunk = [&](int k, int cc) {
    const int ci0 = cc * KC;
#pragma unroll
    for (int j = 0; j < A_PER_THREAD; ++j) {
      int r = a_r + 32 * j;
      int32_t src = nbr_lds[r * 27 + k];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src >= 0)
        v = *reinterpret_cast<const float4*>(in + (int64_t)src * ld_in + ci0 + a_c4 * 4);
      a_reg[j] = v;
    }
    const float* wk = w + ((int64_t)k * cin + ci0) * cout;
#pragma unroll
    for (int j = 0; j < B_PER_THREAD; ++j) {
      int idx = tid + 256 * j;
      int r = idx / (TN / 4);
      int c4 = idx - r * (TN / 4);
      int col = n0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < cout) v = *reinterpret_cast<const float4*>(wk + (int64_t)r * cout + col);
      b_reg[j] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < A_PER_THREAD; ++j) {
      int r = a_r + 32 * j;
      float* dst = A_lds + r * APITCH + a_c4 * 4;
      dst[0] = a_reg[j].x;
      dst[1] = a_reg[j].y;
      dst[2] = a_reg[j].z;
      dst[3] = a_reg[j].w;
    }
#pragma unroll
    for (int j = 0; j < B_PER_THREAD; ++j) {
      int idx = tid + 256 * j;
      *reinterpret_cast<float4*>(B_lds + idx * 4) = b_reg[j];
    }
  };

  // iterate over (k, cc) skipping unused offsets; software pipeline depth 1
  int k = 0;
  while (k < kvol && !((kmask >> k) & 1u)) ++k;
  int cc = 0;
  bool have = k < kvol;
  if (have) load_chunk(k, cc);
  while (have) {
    store_chunk();
    __syncthreads();
    // advance to the next chunk and issue its loads
    int nk = k, ncc = cc + 1;
    if (ncc == cchunks) {
      ncc = 0;
      ++nk;
      while (nk < kvol && !((kmask >> nk) & 1u)) ++nk;
    }
    const bool next = nk < kvol;
    if (next) load_chunk(nk, ncc);

    const float* a_base = A_lds + (wm * 32 + (lane & 31)) * APITCH + (lane >> 5);
    const float* b_base = B_lds + (lane >> 5) * TN + wn * 32 * NT + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      float a = a_base[2 * kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float b = b_base[2 * kk * TN + t * 32];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    k = nk;
    cc = ncc;
    have = next;
  }

  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + wn * 32 * NT + t * 32 + (lane & 31);
    if (col >= cout) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      const int64_t o = orow_lds[r];
      if (o < 0) continue;
      const float* res_row = residual ? residual + o * ld_res : nullptr;
      out[o * ld_out + col] = epilogue(acc[t][i], col, scale, shift, res_row, relu);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_conv_dma<RG, CG, NT>: the production kernel of the 3x3x3 / strided / transposed / 1x1 convolutions
// (round 3).  Same arithmetic as k_conv_mfma (one v_mfma_f32_32x32x2_f32 chain per output element,
// k = 0..26 outer, ci ascending inner: bit-identical to the oracle), different machinery around it.
//
// What bounds the kernel (measured, tools/ubench/mfma_f32_valu.hip): on gfx950 the f32 MFMA runs on the
// SIMD's vector ALU -- a VALU instruction of ANY wave of the SIMD does not overlap with it, it adds its
// 2.3 - 5 cycles to the 64 of an MFMA.  SIMD time = sum of MFMAs + sum of VALU instructions, so the design
// goal is "no VALU instruction that is not the arithmetic itself":
//
//   * staging by LDS-DMA through BUFFER descriptors (buffer_load_dwordx4 ... offen lds): gathered input
//     rows and the weight slab of a 32-channel chunk go L2 -> LDS without VGPR staging or ds_write.  The
//     address of a piece is descriptor base + per-lane 32-bit offset (row * pitch, computed when the
//     neighbour row is fetched) + SCALAR offset (channel chunk / weight slab): no 64-bit VALU
//     arithmetic per chunk.  An absent neighbour is an out-of-range offset: the hardware writes zeros
//     (tools/ubench/buffer_lds_oob.hip), no zero row, no select.  Two stage buffers; the DMA
//     instructions of chunk c+1 go out between the MFMAs of chunk c; one barrier per chunk.
//   * the gathered rows form a 128-B-pitch image (one DMA instruction writes 64 x 16 B contiguously:
//     8 rows); bank conflicts of the A-fragment reads are removed on the SOURCE side: slot p of row r
//     holds channels 4 (p ^ ((r >> 1) & 7)) .. +3, so a ds_read_b128 of one 4-channel piece by 32
//     lanes = 32 rows touches every bank group once.  One ds_read_b128 feeds two MFMA k-steps (channels
//     4j + h and 4j + 2 + h for lane half h: the one v_cndmask per MFMA that is left).
//   * a wave owns ONE 32-row group (x 32 NT columns) and skips every chunk of an offset none of ITS 32
//     rows has; the workgroup (RG row groups x CG column groups = 4 waves) walks the union of its
//     groups' offsets in lock step for the shared weight slab, but the matrix pipe only sees the
//     per-group work.  With rows ordered by the Gray rank of the 27-bit neighbour mask a 32-row group
//     executes 1.2 - 1.4x its useful MFMAs (64-row tiles of the 12-bit order of rounds 1-2: 1.7 - 2.3x).
//   * tiles: RG x CG x NT = 4 x 1 x {1, 2, 4} (128 rows x 32 / 64 / 128 columns: the more columns a wave
//     owns the more MFMAs every gathered byte and every per-chunk instruction feeds), 2 x 2 and 1 x 4
//     where a layer has too few rows to fill the chip with 128-row tiles.
// ------------------------------------------------------------------------------------------------
template <int RG, int CG, int NT>
struct ConvDmaCfg {
  static constexpr int TM = 32 * RG;
  static constexpr int TN = 32 * NT * CG;
  static constexpr int A_BYTES = TM * 128;           // 32 channels x 4 B per row
  static constexpr int B_BYTES = TN * 128;           // 32 slab rows x TN x 4 B
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // nothing else lives in LDS: 32 - 64 KB, 2 - 5 workgroups per CU
};

using i32x4 = __attribute__((ext_vector_type(4))) int;

// raw buffer descriptor over [base, base + bytes): stride 0, offsets at or beyond `bytes` read as zero
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uintptr_t)base);
  r.y = __builtin_amdgcn_readfirstlane((int)((uintptr_t)base >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = __builtin_amdgcn_readfirstlane(0x00020000);
  return r;
}
// LDS-DMA of 64 x 16 B: lane l's bytes at srd.base + voff + soff land at LDS byte lds + 16 l.  Inline asm
// for the reason given at lds_dma16 (common.h): the caller orders the data with s_waitcnt vmcnt + barrier.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void buf_dma16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  soff = __builtin_amdgcn_readfirstlane(soff);   // wave-uniform by contract; folds away when already scalar
  lds = __builtin_amdgcn_readfirstlane(lds);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd),
               "s"(soff), "s"(lds)
               : "m0");
#endif
}
#pragma clang diagnostic pop

template <int RG, int CG, int NT, bool GATHER, bool TRACE>
__global__ __launch_bounds__(256) void k_conv_dma(
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowlist, const uint32_t* __restrict__ gmask, int kvol,
    int64_t n_out, const float* __restrict__ in, int ld_in, unsigned in_bytes, int cin, const float* __restrict__ w, int cout,
    unsigned w_bytes, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, int ld_res, int relu, float* __restrict__ out, int ld_out,
    unsigned long long* __restrict__ trace, int rev_order) {
  using C = ConvDmaCfg<RG, CG, NT>;
  constexpr int TM = C::TM, TN = C::TN;
  static_assert(RG * CG == 4, "4 waves");
  constexpr int A_PIECES = 4 / CG;          // 1-KiB DMA instructions per wave for its row group's 32 rows
  constexpr int B_PIECES = TN / 32;         // ... and for the weight slab (TN / 8 pieces over 4 waves)
  static_assert(A_PIECES + B_PIECES <= 8, "one DMA per A-fragment step of the chunk");
  constexpr int K_END = 32;                 // sentinel offset: no chunk left
  // ONE LDS object (a second one beside a DMA-staged array can make hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(128))) char lds[C::LDS_BYTES];

  unsigned long long rt_begin = 0, rt_loop0 = 0, rt_loop1 = 0;
  if (TRACE) rt_begin = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS addresses of the DMA stay in SGPRs
  const int rg = wave / CG;
  const int cg = wave % CG;
  const int half = lane >> 5;
  const int rl = lane & 31;
  // workgroups take the tiles from the END of the tiling order first: the Gray rank puts the groups with the most
  // offsets last, and heaviest-first leaves the cheap tiles for the ragged end of the launch (CS_CONV_FWD_ORDER=1: front first)
  const int64_t row0 = (int64_t)(rev_order ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * TM;
  const int n0 = blockIdx.y * TN;
  const int cchunks = cin / 32;

  // ---- rows of this wave's group, offsets they use ----
  // `nbr` is the neighbour table in OUTPUT-ROW order (cs_kernelmap::d_nbr, absent = -1), `rowlist` the tiling order
  // (tile slot t holds output row rowlist[t]) and `gmask` the offsets present in every 32-row group of that order,
  // made when the map is built.  Round 5: the copy of the table in tiling order (216 B per map row written and read
  // once per forward, 0.24 ms per stress forward) is gone -- a tile's 32 rows are 32 separate 108-byte segments either
  // way, the row index costs one load per tile.  GATHER = false (1x1): row t is output row t, one "offset".
  // lane (rl, half): output row of group slot rl; both halves hold the same row
  int32_t my_o = -1;
  {
    const int64_t t = row0 + rg * 32 + rl;
    if (t < n_out) my_o = rowlist ? rowlist[t] : (int32_t)t;
  }
  unsigned mymask, kmask;
  if (GATHER) {
    const uint32_t* gm = gmask + row0 / 32;          // (padded to 8 groups: a tile never reads past it)
    mymask = gm[rg];
    kmask = gm[0];
#pragma unroll
    for (int g = 1; g < RG; ++g) kmask |= gm[g];
  } else {
    mymask = row0 + rg * 32 < n_out ? 1u : 0u;
    kmask = 1u;
  }
  mymask = __builtin_amdgcn_readfirstlane(mymask);
  kmask = __builtin_amdgcn_readfirstlane(kmask);

  // DMA geometry.  A: piece i of this wave covers rows 8 (cg A_PIECES + i) .. + 7 of group rg; lane ->
  // (row = lane >> 3, slot = lane & 7), source channels 4 (slot ^ ((row_in_group >> 1) & 7)) .. + 3.
  // B: piece p = wave + 4 j, float index p * 256 + 4 lane of the [32][TN] slab.
  unsigned a_nbr_off[A_PIECES];   // byte offset of nbr[t][0] of the tile row this lane fetches for piece i
  bool a_row_ok[A_PIECES];
  unsigned a_c4b[A_PIECES];       // byte offset of the lane's 4 channels inside the 128-B chunk of a row
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int r = (cg * A_PIECES + i) * 8 + (lane >> 3);
    const int64_t t = row0 + rg * 32 + r;
    a_row_ok[i] = t < n_out;
    // GATHER: byte offset of row rowlist[t] of the neighbour table (the table is in OUTPUT-ROW order, the tile in tiling
    // order: the row index comes from the lane of this wave that holds tile slot r, loaded above as my_o)
    const int32_t o_r = __shfl(my_o, r);
    a_nbr_off[i] = GATHER ? (unsigned)(o_r >= 0 ? o_r : 0) * (unsigned)kvol * 4u : (unsigned)(t < n_out ? t : 0);
    a_c4b[i] = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
  }
  unsigned b_voff[B_PIECES];      // byte offset of this lane's 16 bytes inside the slab of (k, cc)
#pragma unroll
  for (int j = 0; j < B_PIECES; ++j) {
    const int f = (wave + 4 * j) * 256 + 4 * lane;
    const int sr = f / TN;
    b_voff[j] = (unsigned)(sr * cout + (f - sr * TN)) * 4u;
  }
  const i32x4 srd_a = make_srd(in, in_bytes);
  const i32x4 srd_b = make_srd(w + n0, w_bytes - (unsigned)n0 * 4u);
  const unsigned ld_in_b = (unsigned)ld_in * 4u;
  const unsigned absent_row = in_bytes / ld_in_b;          // = n_in (in_bytes = n_in * ld_in * 4)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const unsigned a_lds = lds_base + rg * 4096 + cg * A_PIECES * 1024;
  const unsigned b_lds = lds_base + C::A_BYTES + wave * 1024;

  // chunk stepping without branches: next offset of the union above k, K_END when there is none
  auto step = [&](int& k, int& cc) {
    const int c1 = cc + 1;
    const bool wrap = c1 == cchunks;
    const unsigned rest = k < 31 ? kmask >> (k + 1) : 0u;
    const int knext = rest ? k + 1 + __builtin_ctz(rest) : K_END;
    cc = wrap ? 0 : c1;
    k = k >= K_END ? K_END : (wrap ? knext : k);
  };
  // neighbour rows of the A pieces for offset k: scalar base + 32-bit lane offset (one plain load each,
  // consumed one iteration later).  GATHER is a template flag, not a test of `nbr`: a load inside a run-time
  // branch is waited for where the branch ends -- and that vmcnt(0) takes the DMA issued before it along
  auto fetch_src = [&](int k, int32_t (&src)[A_PIECES]) {
    const char* nbr_k = reinterpret_cast<const char*>(nbr + (k < kvol ? k : 0));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i)
      src[i] = GATHER ? *reinterpret_cast<const int32_t*>(nbr_k + a_nbr_off[i]) : (int32_t)a_nbr_off[i];
  };
  // byte offsets of the gathered rows (an absent neighbour, a row past the end: out of range = zeros)
  auto a_offsets = [&](const int32_t (&src)[A_PIECES], unsigned (&vo)[A_PIECES]) {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i)
      // an absent neighbour is -1 in the table: as unsigned it clamps to row n_in, ONE PAST the input tensor, and the
      // buffer descriptor's range check returns zeros for that row (one v_min_u32 per piece and chunk)
      vo[i] = __umul24(GATHER ? min((unsigned)src[i], absent_row) : (unsigned)src[i], ld_in_b) + a_c4b[i];
  };
  auto b_soff = [&](int k, int cc) { return (unsigned)(((k < K_END ? k : 0) * cin + cc * 32) * cout) * 4u; };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  // residual rows are requested NOW (NT <= 2: 16 NT registers held through the loop): the loop hides the round
  // trip that would otherwise stand between the last MFMA and the stores.  Unconditional loads (rows past the
  // end read the tile's first row): a load under a per-row condition is waited for before the next is issued.
  constexpr bool RES_EARLY = NT <= 2;
  int32_t orow[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) orow[i] = __shfl(my_o, (i & 3) + 8 * (i >> 2) + 4 * half);
  const int32_t o_safe = rowlist ? rowlist[row0] : (int32_t)row0;   // the tile's first row always exists
  float res[NT][16];
  if (RES_EARLY && residual) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        res[t][i] = residual[(int64_t)(orow[i] >= 0 ? orow[i] : o_safe) * ld_res + n0 + cg * 32 * NT + t * 32 + (lane & 31)];
  }

  // chunk c = (k, cc), c + 1 = (nk, ncc), c + 2 = (k2, cc2)
  int k = kmask ? __builtin_ctz(kmask) : K_END, cc = 0;
  int nk = k, ncc = cc;
  step(nk, ncc);
  int k2 = nk, cc2 = ncc;
  step(k2, cc2);
  unsigned vo_n[A_PIECES];         // row offsets of the chunk staged next (c + 1)
  int32_t src2[A_PIECES];          // neighbour rows of chunk c + 2
  if (k < K_END) {
    int32_t s0[A_PIECES], s1[A_PIECES];
    fetch_src(k, s0);
    fetch_src(nk, s1);
    fetch_src(k2, src2);
    unsigned vo0[A_PIECES];
    a_offsets(s0, vo0);
    if ((mymask >> k) & 1u) {
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) buf_dma16(vo0[i], srd_a, 0u, a_lds + i * 1024);
    }
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) buf_dma16(b_voff[j], srd_b, b_soff(k, 0), b_lds + j * 4096);
    a_offsets(s1, vo_n);
  }
  // LDS read addresses: A piece j of this lane's row sits at slot j ^ swz -> base ^ (j << 4); the stage
  // buffer is toggled by adding +- STAGE_BYTES once per chunk
  unsigned a_rd = lds_base + rg * 4096 + rl * 128 + (((rl >> 1) & 7) << 4) + 4 * half;
  unsigned b_rd = lds_base + C::A_BYTES + (half * TN + cg * 32 * NT + rl) * 4;
  int buf = 0;
  unsigned long long t_start = 0, t_wait = 0, t_vm = 0, t_body = 0, n_chunk = 0, n_act = 0, t_pre = 0;
  if (TRACE) {
    t_start = __builtin_amdgcn_s_memtime();
    rt_loop0 = __builtin_amdgcn_s_memrealtime();
  }
  // One chunk, accumulators in `src` before and in `dst` after it.  hipcc selects the first MFMA behind `if (active)`
  // in its untied form (its SrcC has a second use on the skipping path) and, with ONE home for the accumulators, copies
  // all 16 NT of them every chunk to feed it.  With two homes that alternate per chunk the untied MFMA IS the move;
  // only a chunk the wave skips pays for a copy.
  auto chunk = [&](f32x16 (&src)[NT], f32x16 (&dst)[NT]) __attribute__((always_inline)) {
    // the DMA of this chunk has landed (every wave waits for its own pieces, then the barrier) and every
    // wave is done reading the other buffer (it read it before arriving here)
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (TRACE) t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TRACE) t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (TRACE) t2 = __builtin_amdgcn_s_memtime();
    const bool active = (mymask >> k) & 1u;
    // first fragment reads of this chunk: their latency runs under the address arithmetic below
    // (each lane half reads ITS two channels of the 16-B slot -- 4 j + half and 4 j + 2 + half, one ds_read2_b32 --
    // instead of the whole slot and a v_cndmask per MFMA: VALU instructions take the matrix pipe's cycles)
    float2 av[8];
    float bv[16][NT];
    auto rd = [&](int j) {
      const float* ap = reinterpret_cast<const float*>(lds + ((a_rd ^ (unsigned)(j << 4)) - lds_base));
      av[j].x = ap[0];
      av[j].y = ap[2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bv[2 * j][t] = *reinterpret_cast<const float*>(lds + (b_rd - lds_base) + ((4 * j) * TN + t * 32) * 4);
        bv[2 * j + 1][t] = *reinterpret_cast<const float*>(lds + (b_rd - lds_base) + ((4 * j + 2) * TN + t * 32) * 4);
      }
    };
    // (unconditional: a wave that skips this chunk reads bytes it never uses, a read inside a branch would be
    // waited for where the branch ends)
    rd(0);
    rd(1);
    // row offsets of chunk c + 2 from the neighbour rows fetched one iteration ago, BEFORE this iteration
    // puts anything on the memory queue: hipcc's own wait for those (long finished) loads is then a no-op;
    // behind the DMAs its counted vmcnt would wait for the DMAs instead
    unsigned vo_nn[A_PIECES];
    a_offsets(src2, vo_nn);
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) asm volatile("" ::"v"(vo_nn[i]));
    int k3 = k2, cc3 = cc2;
    step(k3, cc3);
    fetch_src(k3, src2);             // neighbour rows of chunk c + 3 (used one iteration from now)
    if (TRACE) t3 = __builtin_amdgcn_s_memtime();
    // (uniform values that hipcc would otherwise carry in VGPRs and read back with v_readfirstlane per DMA)
    nk = __builtin_amdgcn_readfirstlane(nk);
    ncc = __builtin_amdgcn_readfirstlane(ncc);
    const bool stage_b = nk < K_END;
    const bool stage_a = stage_b && ((mymask >> nk) & 1u);   // nobody reads the rows of a group without offset nk
    const unsigned a_dst = a_lds + (buf ^ 1) * C::STAGE_BYTES, b_dst = b_lds + (buf ^ 1) * C::STAGE_BYTES;
    const unsigned a_so = __builtin_amdgcn_readfirstlane((unsigned)ncc * 128u);
    const unsigned b_so = __builtin_amdgcn_readfirstlane(b_soff(nk, ncc));
    if (active) {
      // one chunk on the matrix pipe.  Hand-placed software pipeline (the volatile DMA statements pin the order):
      // fragment reads run two steps ahead of their MFMAs, one DMA instruction of the next stage goes out
      // behind the MFMAs of each of the first A_PIECES + B_PIECES steps: the CU's DMA traffic is spread over
      // the chunk instead of being one burst behind the barrier
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 2 < 8) rd(j + 2);
        const float a0 = av[j].x;
        const float a1 = av[j].y;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          dst[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[2 * j][t], j == 0 ? src[t] : dst[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          dst[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[2 * j + 1][t], dst[t], 0, 0, 0);
        if (j < A_PIECES) {
          if (stage_a) buf_dma16(vo_n[j], srd_a, a_so, a_dst + j * 1024);
        } else if (j < A_PIECES + B_PIECES) {
          if (stage_b) buf_dma16(b_voff[j - A_PIECES], srd_b, b_so, b_dst + (j - A_PIECES) * 4096);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) dst[t] = src[t];
      if (stage_a) {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) buf_dma16(vo_n[i], srd_a, a_so, a_dst + i * 1024);
      }
      if (stage_b) {
#pragma unroll
        for (int j = 0; j < B_PIECES; ++j) buf_dma16(b_voff[j], srd_b, b_so, b_dst + j * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) vo_n[i] = vo_nn[i];
    a_rd += buf ? -C::STAGE_BYTES : C::STAGE_BYTES;
    b_rd += buf ? -C::STAGE_BYTES : C::STAGE_BYTES;
    if (TRACE) {
      t_vm += t1 - t0;
      t_wait += t2 - t1;
      t_pre += t3 - t2;
      t_body += __builtin_amdgcn_s_memtime() - t3;
      ++n_chunk;
      n_act += active;
    }
    k = nk; cc = ncc;
    nk = k2; ncc = cc2;
    k2 = k3; cc2 = cc3;
    buf ^= 1;
  };
  {
    f32x16 acc_b[NT];
    while (k < K_END) {
      chunk(acc, acc_b);
      if (k >= K_END) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = acc_b[t];
        break;
      }
      chunk(acc_b, acc);
    }
  }
  if (TRACE) {
    rt_loop1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* tr = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
      tr[0] = __builtin_amdgcn_s_memtime() - t_start; tr[1] = t_vm; tr[2] = t_wait; tr[3] = t_body; tr[4] = n_chunk;
      tr[5] = n_act; tr[6] = t_pre;
    }
  }

  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + cg * 32 * NT + t * 32 + (lane & 31);
    if (!RES_EARLY && residual) {
#pragma unroll
      for (int i = 0; i < 16; ++i) res[t][i] = residual[(int64_t)(orow[i] >= 0 ? orow[i] : o_safe) * ld_res + col];
    }
    const float sc = scale ? scale[col] : 1.f;
    const float sh = shift ? shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[t][i];
      if (scale)
        v = __fmaf_rn(v, sc, sh);
      else if (shift)
        v = v + sh;
      if (residual) v = v + res[t][i];
      if (relu) v = fmaxf(v, 0.0f);
      if (orow[i] >= 0) out[(int64_t)orow[i] * ld_out + col] = v;
    }
  }
  if (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* tr = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
      // 100 MHz ticks: kernel entry (absolute), prologue, loop, epilogue packed 16 bits each above it
      tr[7] = (rt_begin & 0xffffffffULL) | ((rt_loop0 - rt_begin) & 0xffff) << 32 | ((rt_loop1 - rt_loop0) & 0xffff) << 48;
      tr[6] = t_pre | (rt_end - rt_loop1) << 40;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT, off by default (CS_CONV_SPLIT=3 or 2; SURVEY 8d: "bf16 only behind a parity-checked flag"):
// the same convolutions on the bf16 matrix cores.  The exact chain above runs on the SIMD's f32 vector ALU (157 TF peak);
// v_mfma_f32_32x32x16_bf16 is a true matrix-core instruction (16x the rate), so every f32 operand is cut into NS bf16
// pieces (truncation: a = hi + mid + lo EXACTLY for NS = 3, every piece 8 significant bits) and the products whose
// weight is above 2^-24 are accumulated in f32: NS = 3 keeps hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi (6 MFMAs per 16
// channels, dropped terms <= 2^-23 |a||b|), NS = 2 keeps hi.hi, hi.lo, lo.hi (3 MFMAs, ~2^-16 |a||b|).  NOT bit-identical
// to the oracle's fma chain (other summation order, other rounding): the default path and every parity test stay on
// k_conv_dma; tools/conv_split_report.py reports speed and the feature / ranking differences of this path.
//   * input rows: the same gathered f32 row image as k_conv_dma (LDS-DMA through a buffer descriptor, swizzled slots);
//     a lane reads ITS 8 channels of a 16-channel step (two ds_read_b128) and cuts them in registers: v_perm_b32 packs the
//     high halves of two values, v_and + v_sub give the exact remainder -- 5.5 VALU instructions per value for NS = 3.
//   * weights: cut once per layer (k_split_weights, cached by pointer) into fragment order
//     [offset][32-channel chunk][piece][step][lane half][cout][8 x bf16]: a B fragment is one ds_read_b128.
// ------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NS>
__global__ void k_split_weights(const float* __restrict__ w, int kvol, int cin, int cout, uint16_t* __restrict__ wq) {
  // one thread per (offset, channel, cout)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)kvol * cin * cout;
  if (i >= total) return;
  const int col = (int)(i % cout);
  const int ci = (int)((i / cout) % cin);
  const int k = (int)(i / ((int64_t)cout * cin));
  const int cc = ci >> 5, s = (ci >> 4) & 1, h = (ci >> 3) & 1, e = ci & 7;
  float r = w[i];
  const int64_t chunk = ((int64_t)k * (cin >> 5) + cc) * (NS * 4);
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    unsigned bits;
    if (pl == NS - 1 && NS == 2) {
      // last piece of the two-piece form: round to nearest even instead of cutting
      const unsigned u = __float_as_uint(r);
      bits = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    } else {
      bits = __float_as_uint(r) >> 16;
    }
    r = r - __uint_as_float(bits << 16);
    const int seg = (pl * 2 + s) * 2 + h;
    wq[((chunk + seg) * cout + col) * 8 + e] = (uint16_t)bits;
  }
}

template <int RG, int CG, int NT, int NS, bool GATHER>
__global__ __launch_bounds__(256) void k_conv_split(
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowlist, const uint32_t* __restrict__ gmask, int kvol,
    int64_t n_out, const float* __restrict__ in, int ld_in, unsigned in_bytes, int cin, const uint16_t* __restrict__ wq, int cout,
    unsigned wq_bytes, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, int ld_res, int relu, float* __restrict__ out, int ld_out, int rev_order, int dbg) {
  constexpr int TM = 32 * RG, TN = 32 * NT * CG;
  static_assert(RG * CG == 4, "4 waves");
  constexpr int A_PIECES = 4 / CG;
  constexpr int SEGS = NS * 4;
  constexpr int A_BYTES = TM * 128, B_BYTES = SEGS * TN * 16, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NPB = B_BYTES / 1024;      // 1-KiB DMA instructions per weight stage
  constexpr int B_PIECES = (NPB + 3) / 4;
  constexpr int K_END = 32;
  __shared__ __attribute__((aligned(128))) char lds[2 * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / CG;
  const int cg = wave % CG;
  const int half = lane >> 5;
  const int rl = lane & 31;
  const int64_t row0 = (int64_t)(rev_order ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * TM;
  const int n0 = blockIdx.y * TN;
  const int cchunks = cin / 32;

  int32_t my_o = -1;
  {
    const int64_t t = row0 + rg * 32 + rl;
    if (t < n_out) my_o = rowlist ? rowlist[t] : (int32_t)t;
  }
  unsigned mymask, kmask;
  if (GATHER) {
    const uint32_t* gm = gmask + row0 / 32;
    mymask = gm[rg];
    kmask = gm[0];
#pragma unroll
    for (int g = 1; g < RG; ++g) kmask |= gm[g];
  } else {
    mymask = row0 + rg * 32 < n_out ? 1u : 0u;
    kmask = 1u;
  }
  mymask = __builtin_amdgcn_readfirstlane(mymask);
  kmask = __builtin_amdgcn_readfirstlane(kmask);

  unsigned a_nbr_off[A_PIECES];
  unsigned a_c4b[A_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int r = (cg * A_PIECES + i) * 8 + (lane >> 3);
    const int64_t t = row0 + rg * 32 + r;
    // GATHER: byte offset of row rowlist[t] of the neighbour table (the table is in OUTPUT-ROW order, the tile in tiling
    // order: the row index comes from the lane of this wave that holds tile slot r, loaded above as my_o)
    const int32_t o_r = __shfl(my_o, r);
    a_nbr_off[i] = GATHER ? (unsigned)(o_r >= 0 ? o_r : 0) * (unsigned)kvol * 4u : (unsigned)(t < n_out ? t : 0);
    a_c4b[i] = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
  }
  // weight stage = [SEGS][TN][16 B], linear in 16-B units u = piece * 64 + lane: segment u / TN, column u % TN
  unsigned b_voff[B_PIECES];
#pragma unroll
  for (int j = 0; j < B_PIECES; ++j) {
    const int u = (wave + 4 * j) * 64 + lane;
    const int seg = u / TN, col = u - seg * TN;
    b_voff[j] = (unsigned)(seg * cout + col) * 16u;
  }
  const i32x4 srd_a = make_srd(in, in_bytes);
  const i32x4 srd_b = make_srd(wq + (size_t)n0 * 8, wq_bytes - (unsigned)n0 * 16u);
  const unsigned ld_in_b = (unsigned)ld_in * 4u;
  const unsigned absent_row = in_bytes / ld_in_b;          // = n_in (in_bytes = n_in * ld_in * 4)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const unsigned a_lds = lds_base + rg * 4096 + cg * A_PIECES * 1024;
  const unsigned b_lds = lds_base + A_BYTES + wave * 1024;

  auto step = [&](int& k, int& cc) {
    const int c1 = cc + 1;
    const bool wrap = c1 == cchunks;
    const unsigned rest = k < 31 ? kmask >> (k + 1) : 0u;
    const int knext = rest ? k + 1 + __builtin_ctz(rest) : K_END;
    cc = wrap ? 0 : c1;
    k = k >= K_END ? K_END : (wrap ? knext : k);
  };
  auto fetch_src = [&](int k, int32_t (&src)[A_PIECES]) {
    const char* nbr_k = reinterpret_cast<const char*>(nbr + (k < kvol ? k : 0));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i)
      src[i] = GATHER ? *reinterpret_cast<const int32_t*>(nbr_k + a_nbr_off[i]) : (int32_t)a_nbr_off[i];
  };
  auto stage = [&](int k, int cc, const int32_t (&src)[A_PIECES], int b) {
    const unsigned a_so = __builtin_amdgcn_readfirstlane((unsigned)cc * 128u);
    const unsigned b_so = __builtin_amdgcn_readfirstlane((unsigned)((k * cchunks + cc) * SEGS) * (unsigned)cout * 16u);
    if ((mymask >> k) & 1u) {
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i)
        buf_dma16(__umul24(GATHER ? min((unsigned)src[i], absent_row) : (unsigned)src[i], ld_in_b) + a_c4b[i], srd_a, a_so,
                  a_lds + b * STAGE_BYTES + i * 1024);
    }
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j)
      if (wave + 4 * j < NPB) buf_dma16(b_voff[j], srd_b, b_so, b_lds + b * STAGE_BYTES + j * 4096);
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  int32_t orow[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) orow[i] = __shfl(my_o, (i & 3) + 8 * (i >> 2) + 4 * half);

  int k = kmask ? __builtin_ctz(kmask) : K_END, cc = 0;
  int nk = k, ncc = cc;
  step(nk, ncc);
  int32_t src_n[A_PIECES];   // neighbour rows of the chunk staged next
  if (k < K_END) {
    int32_t s0[A_PIECES];
    fetch_src(k, s0);
    fetch_src(nk, src_n);
    stage(k, 0, s0, 0);
  }
  // slot of channel block q (4 channels) of row rl: q ^ ((rl >> 1) & 7); this lane reads blocks 4 s + 2 half + {0, 1}
  const unsigned a_rd0 = rg * 4096 + rl * 128 + ((((rl >> 1) & 7) ^ (2 * half)) << 4);
  const unsigned b_rd0 = A_BYTES + (half * TN + cg * 32 * NT + rl) * 16;
  int buf = 0;
  while (k < K_END) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool active = (mymask >> k) & 1u;
    if (nk < K_END && !(dbg & 1)) stage(nk, ncc, src_n, buf ^ 1);
    int k2 = nk, cc2 = ncc;
    step(k2, cc2);
    fetch_src(k2, src_n);   // waited for by the vmcnt(0) of the next iteration, used by its stage()
    if (active && !(dbg & 2)) {
      const char* sb = lds + buf * STAGE_BYTES;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(sb + (a_rd0 ^ (unsigned)((4 * s) << 4)));
        const float4 v1 = *reinterpret_cast<const float4*>(sb + (a_rd0 ^ (unsigned)((4 * s + 1) << 4)));
        float r[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        union {
          unsigned u[4];
          bf16x8 v;
        } ap[NS];
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const unsigned x = __float_as_uint(r[2 * e2]), y = __float_as_uint(r[2 * e2 + 1]);
            if (pl == NS - 1 && NS == 2) {
              unsigned pk;
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(r[2 * e2]), "v"(r[2 * e2 + 1]));
              ap[pl].u[e2] = pk;
            } else {
              ap[pl].u[e2] = __builtin_amdgcn_perm(y, x, 0x07060302u);
              if (pl + 1 < NS) {
                r[2 * e2] = r[2 * e2] - __uint_as_float(x & 0xffff0000u);
                r[2 * e2 + 1] = r[2 * e2 + 1] - __uint_as_float(y & 0xffff0000u);
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          bf16x8 bp[NS];
#pragma unroll
          for (int pl = 0; pl < NS; ++pl)
            bp[pl] = *reinterpret_cast<const bf16x8*>(sb + b_rd0 + ((pl * 2 + s) * 2 * TN + t * 32) * 16);
          // smallest products first
#pragma unroll
          for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa <= sum; ++pa) {
              const int pb = sum - pa;
              if (pa < NS && pb < NS) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa].v, bp[pb], acc[t], 0, 0, 0);
            }
        }
      }
    }
    k = nk;
    cc = ncc;
    nk = k2;
    ncc = cc2;
    buf ^= 1;
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + cg * 32 * NT + t * 32 + (lane & 31);
    const float sc = scale ? scale[col] : 1.f;
    const float sh = shift ? shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (orow[i] < 0) continue;
      float v = acc[t][i];
      if (scale)
        v = __fmaf_rn(v, sc, sh);
      else if (shift)
        v = v + sh;
      if (residual) v = v + residual[(int64_t)orow[i] * ld_res + col];
      if (relu) v = fmaxf(v, 0.0f);
      out[(int64_t)orow[i] * ld_out + col] = v;
    }
  }
}

// Cin = 1 (the 1 -> 32 stem, model/resunet.py:49-57): a 27-term fma chain per output.  A lane first
// gathers the <= 27 scalar inputs of ITS row (27 independent loads in flight), the wave shares them
// through LDS and then every lane owns one output channel (its 27 weights in registers) and walks the
// wave's 64 rows: stores are whole 128-B rows.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv_stem(const int32_t* __restrict__ nbr, int64_t n_out,
                                                   const float* __restrict__ in, int ld_in,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ residual, int ld_res, int relu,
                                                   float* __restrict__ out, int ld_out) {
  static_assert(COUT == 32, "one lane half per row parity");
  __shared__ __attribute__((aligned(16))) float xs[4][64][28];   // 112-B rows: seven 16-B reads, broadcast (conflict-free)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
  {
    // every load is unconditional (absent neighbours / rows past the end read row 0 and are zeroed afterwards):
    // a per-element "load or not" makes hipcc wait for each of the 27 gathers in turn
    const bool live = base + lane < n_out;
    const int64_t o = live ? base + lane : n_out - 1;
    int32_t src[27];
    float x[28];
#pragma unroll
    for (int k = 0; k < 27; ++k) src[k] = nbr[o * 27 + k];
#pragma unroll
    for (int k = 0; k < 27; ++k) x[k] = in[(int64_t)(src[k] >= 0 ? src[k] : 0) * ld_in];
    // an absent offset contributes fma(0, w, acc) = acc: the chain of the present offsets, in order, like the oracle
    // (the accumulator starts at +0 and a sum of products with +0 can never turn it into -0)
#pragma unroll
    for (int k = 0; k < 27; ++k) x[k] = (live && src[k] >= 0) ? x[k] : 0.f;
    x[27] = 0.f;
    float4* row = reinterpret_cast<float4*>(&xs[wave][lane][0]);
#pragma unroll
    for (int q = 0; q < 7; ++q) row[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
  }
  __syncthreads();
  const int co = lane & 31;
  float wr[28];
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = w[k * COUT + co];
  wr[27] = 0.f;
  // two rows per step (two independent fma chains in flight per lane)
  for (int r = lane >> 5; r < 64; r += 4) {
    const int64_t o0 = base + r, o1 = o0 + 2;
    if (o0 >= n_out) break;
    const float4* row0 = reinterpret_cast<const float4*>(&xs[wave][r][0]);
    const float4* row1 = reinterpret_cast<const float4*>(&xs[wave][r + 2][0]);
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const float4 v0 = row0[q], v1 = row1[q];
      acc0 = __fmaf_rn(v0.x, wr[4 * q], acc0);
      acc1 = __fmaf_rn(v1.x, wr[4 * q], acc1);
      acc0 = __fmaf_rn(v0.y, wr[4 * q + 1], acc0);
      acc1 = __fmaf_rn(v1.y, wr[4 * q + 1], acc1);
      acc0 = __fmaf_rn(v0.z, wr[4 * q + 2], acc0);
      acc1 = __fmaf_rn(v1.z, wr[4 * q + 2], acc1);
      if (q < 6) {
        acc0 = __fmaf_rn(v0.w, wr[4 * q + 3], acc0);
        acc1 = __fmaf_rn(v1.w, wr[4 * q + 3], acc1);
      }
    }
    out[o0 * ld_out + co] = epilogue(acc0, co, scale, shift, residual ? residual + o0 * ld_res : nullptr, relu);
    if (o1 < n_out) out[o1 * ld_out + co] = epilogue(acc1, co, scale, shift, residual ? residual + o1 * ld_res : nullptr, relu);
  }
}


// Generic VALU path (any cin / cout / alignment; used for cin = 1, the 1 -> 32 stem conv).
// One thread per (out row, out channel); same canonical fma order.
__global__ void k_conv_generic(const int32_t* __restrict__ nbr, int kvol, int64_t n_out,
                               const float* __restrict__ in, int ld_in, int cin,
                               const float* __restrict__ w, int cout,
                               const float* __restrict__ scale, const float* __restrict__ shift,
                               const float* __restrict__ residual, int ld_res, int relu,
                               float* __restrict__ out, int ld_out) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_out * cout) return;
  int64_t o = t / cout;
  int co = (int)(t - o * cout);
  float acc = 0.0f;
  for (int k = 0; k < kvol; ++k) {
    int32_t src = nbr ? nbr[o * kvol + k] : (int32_t)o;
    if (src < 0) continue;
    const float* x = in + (int64_t)src * ld_in;
    const float* wk = w + (int64_t)k * cin * cout + co;
    for (int ci = 0; ci < cin; ++ci) acc = __fmaf_rn(x[ci], wk[(int64_t)ci * cout], acc);
  }
  const float* res_row = residual ? residual + o * ld_res : nullptr;
  out[o * ld_out + co] = epilogue(acc, co, scale, shift, res_row, relu);
}

__global__ void k_affine_act(int64_t n, int c, const float* __restrict__ in, int ld_in,
                             const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ residual, int ld_res, int relu,
                             float* __restrict__ out, int ld_out) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < n * c; t += stride) {
    int64_t r = t / c;
    int col = (int)(t - r * c);
    const float* res_row = residual ? residual + r * ld_res : nullptr;
    out[r * ld_out + col] = epilogue(in[r * ld_in + col], col, scale, shift, res_row, relu);
  }
}

// one wave per row; sequential-per-lane partial sums then a fixed xor-tree -> deterministic
__global__ void k_row_l2norm(int64_t n, int c, const float* __restrict__ in, int ld_in, float eps,
                             float* __restrict__ out, int ld_out) {
  int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* x = in + row * ld_in;
  float s = 0.0f;
  for (int i = lane; i < c; i += 64) s = __fmaf_rn(x[i], x[i], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  float nrm = sqrtf(s);
  nrm = fmaxf(nrm, eps);
  for (int i = lane; i < c; i += 64) out[row * ld_out + i] = x[i] / nrm;
}

// c <= 16 (the 16-channel voxel features): 16 lanes per row, four rows per wave.  The same butterfly as above from
// offset 8 down -- the offsets 32 and 16 of the one-wave-per-row kernel only ever add the exact zeros of the lanes
// beyond c -- so the same sums, bit for bit, with a quarter of the waves and whole 64-B rows per load.
__global__ void k_row_l2norm16(int64_t n, int c, const float* __restrict__ in, int ld_in, float eps,
                               float* __restrict__ out, int ld_out) {
  const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = g >> 4;
  const int ch = (int)(g & 15);
  const bool live = row < n && ch < c;
  const float x = live ? in[row * ld_in + ch] : 0.0f;
  float s = __fmaf_rn(x, x, 0.0f);
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  float nrm = sqrtf(s);
  nrm = fmaxf(nrm, eps);
  if (live) out[row * ld_out + ch] = x / nrm;
}

__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ void k_segmax_init(unsigned* buf, int64_t n) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) buf[t] = f2ord(-INFINITY);
}
// One atomic per RUN of rows of one sample (not one per row): a thread owns one column of a block
// of SEGMAX_ROWS consecutive rows, keeps the running maximum while the batch index stays the same and flushes it when
// it changes (rows grouped by sample, the usual case: 32 x fewer atomics; any other order is still correct).
constexpr int SEGMAX_ROWS = 32;
__global__ void k_segmax_runs(int64_t n, int c, const float* __restrict__ in, int ld_in,
                              const int32_t* __restrict__ batch, int batch_ld, int n_batch, unsigned* obuf) {
  const int col = blockIdx.y * blockDim.x + threadIdx.x;
  if (col >= c) return;
  const int64_t r0 = (int64_t)blockIdx.x * SEGMAX_ROWS;
  const int64_t r1 = r0 + SEGMAX_ROWS < n ? r0 + SEGMAX_ROWS : n;
  int cur = -1;
  unsigned best = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const int b = batch[r * batch_ld];
    if (b != cur) {
      if (cur >= 0 && cur < n_batch) atomicMax(&obuf[(int64_t)cur * c + col], best);
      cur = b;
      best = 0;   // f2ord maps every float above 0: the first value of the run replaces it
    }
    const unsigned v = f2ord(in[r * ld_in + col]);
    best = v > best ? v : best;
  }
  if (cur >= 0 && cur < n_batch && r1 > r0) atomicMax(&obuf[(int64_t)cur * c + col], best);
}
__global__ void k_segmax_fin(unsigned* buf, int64_t n) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) reinterpret_cast<float*>(buf)[t] = ord2f(buf[t]);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// ------------------------------------------------------------------------------------------------
// Instance normalisation (MinkowskiInstanceNorm of the IN network variants, model/common.py:23-24):
// per sample and channel  out = (x - mean) / sqrt(var + eps) * weight + bias  with the biased variance.
// Fixed summation order (the oracle restates it): a sample's rows in chunks of INORM_CHUNK consecutive
// rows, f64 sequential sum inside a chunk, chunk sums added in chunk order.  mean and var are rounded to
// f32, 1/sqrt in f64 rounded to f32, the affine part in f32 without contraction.
// Rows must be grouped by sample (seg[b] .. seg[b+1], the collate order).
// ------------------------------------------------------------------------------------------------
constexpr int INORM_CHUNK = 256;
constexpr int INORM_SLICES = 64;

__device__ __forceinline__ int64_t inorm_slot(const int32_t* seg, int b) { return (int64_t)(seg[b] / INORM_CHUNK) + b; }

// grid (INORM_SLICES, n_batch, channel groups of 256); thread = channel.  PASS 0: sum of x; PASS 1: sum of
// (x - mean)^2 with the f32 mean of PASS 0.
template <int PASS>
__global__ __launch_bounds__(256) void k_inorm_partial(const float* __restrict__ x, int ld, int c,
                                                       const int32_t* __restrict__ seg,
                                                       const float* __restrict__ mean,
                                                       double* __restrict__ partial) {
  const int b = blockIdx.y;
  const int ch = blockIdx.z * 256 + threadIdx.x;
  if (ch >= c) return;
  const int r0 = seg[b], r1 = seg[b + 1];
  const int chunks = (r1 - r0 + INORM_CHUNK - 1) / INORM_CHUNK;
  const float m = PASS ? mean[(int64_t)b * c + ch] : 0.f;
  for (int k = blockIdx.x; k < chunks; k += gridDim.x) {
    const int a = r0 + k * INORM_CHUNK, e = min(r1, a + INORM_CHUNK);
    double acc = 0.0;
    for (int r = a; r < e; ++r) {
      const float v = x[(int64_t)r * ld + ch];
      if (PASS) {
        const float d = v - m;
        acc += (double)d * (double)d;
      } else {
        acc += (double)v;
      }
    }
    partial[(inorm_slot(seg, b) + k) * c + ch] = acc;
  }
}

// one thread per (sample, channel): chunk sums in order.  PASS 0 -> mean; PASS 1 -> 1 / sqrt(var + eps)
template <int PASS>
__global__ void k_inorm_stat(const double* __restrict__ partial, int c, int n_batch,
                             const int32_t* __restrict__ seg, float eps, float* __restrict__ stat) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_batch * c) return;
  const int b = (int)(t / c), ch = (int)(t - (int64_t)b * c);
  const int len = seg[b + 1] - seg[b];
  const int chunks = (len + INORM_CHUNK - 1) / INORM_CHUNK;
  double acc = 0.0;
  for (int k = 0; k < chunks; ++k) acc += partial[(inorm_slot(seg, b) + k) * c + ch];
  if (len == 0) {
    stat[t] = 0.f;
    return;
  }
  const float v = (float)(acc / (double)len);
  stat[t] = PASS ? (float)(1.0 / sqrt((double)v + (double)eps)) : v;
}

__global__ void k_inorm_apply(const float* __restrict__ x, int ld_in, int c, int n_batch,
                              const int32_t* __restrict__ seg, const float* __restrict__ mean,
                              const float* __restrict__ inv_std, const float* __restrict__ weight,
                              const float* __restrict__ bias, float* __restrict__ out, int ld_out) {
  const int b = blockIdx.y;
  const int r0 = seg[b], r1 = seg[b + 1];
  const int64_t total = (int64_t)(r1 - r0) * c;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int r = r0 + (int)(t / c), ch = (int)(t % c);
    const float d = x[(int64_t)r * ld_in + ch] - mean[(int64_t)b * c + ch];
    float v = d * inv_std[(int64_t)b * c + ch];
    if (weight) v = v * weight[ch];
    if (bias) v = v + bias[ch];
    out[(int64_t)r * ld_out + ch] = v;
  }
}

}  // namespace cs

using namespace cs;

namespace {
struct SplitKey {
  const float* w;
  int kvol, cin, cout, ns;
  bool operator<(const SplitKey& o) const {
    return std::tie(w, kvol, cin, cout, ns) < std::tie(o.w, o.kvol, o.cin, o.cout, o.ns);
  }
};
std::mutex g_split_mu;
std::map<SplitKey, uint16_t*> g_split_w;   // pieces of a layer's weights (experiment; never freed: a handful of MB)
}  // namespace

extern "C" {

int cs_conv_fwd(const cs_kernelmap* km, int64_t n_in, int64_t n_out, const float* d_in, int ld_in,
                int cin, const float* d_w, int cout, const float* d_scale, const float* d_shift,
                const float* d_residual, int ld_res, int relu, float* d_out, int ld_out,
                void* stream) {
  if (n_out == 0 && n_in == 0) return CS_OK;  // empty batch (empty torch tensors have NULL storage)
  CS_REQUIRE(d_in && d_w && d_out, CS_ERR_INVALID, "cs_conv_fwd: NULL tensor");
  CS_REQUIRE(cin >= 1 && cout >= 1 && ld_in >= cin && ld_out >= cout, CS_ERR_INVALID,
             "cs_conv_fwd: bad channel / leading dimension (cin %d ld_in %d cout %d ld_out %d)",
             cin, ld_in, cout, ld_out);
  CS_REQUIRE(!d_scale || d_shift, CS_ERR_INVALID, "cs_conv_fwd: scale without shift");
  CS_REQUIRE(!d_residual || ld_res >= cout, CS_ERR_INVALID, "cs_conv_fwd: bad residual ld");
  CS_REQUIRE(n_in < (1LL << 31) && n_out < (1LL << 31), CS_ERR_INVALID,
             "cs_conv_fwd: too many rows");
  int kvol = 1;
  const int32_t* nbr = nullptr;
  const int32_t* rowlist = nullptr;
  const int32_t* nbr_t = nullptr;     // neighbour table (row order) when the tiling order + group masks exist (k_conv_dma)
  const uint32_t* gmask = nullptr;
  if (km) {
    CS_REQUIRE(km->n_out == n_out && km->n_in == n_in, CS_ERR_INVALID,
               "cs_conv_fwd: kernel map is for %lld -> %lld rows, tensors have %lld -> %lld",
               (long long)km->n_in, (long long)km->n_out, (long long)n_in, (long long)n_out);
    kvol = km->kvol;
    nbr = km->d_nbr;
    rowlist = km->d_rowlist;
    nbr_t = (km->d_rowlist && km->d_gmask) ? km->d_nbr : nullptr;
    gmask = km->d_gmask;
  } else {
    CS_REQUIRE(n_in == n_out, CS_ERR_INVALID, "cs_conv_fwd: 1x1 conv needs n_in == n_out");
  }
  if (n_out == 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  // profiling: work units = 2 x pairs x Cin x Cout.  The pair count of a freshly built map may still be on its way to the host:
  // waiting for it here stalled the host behind every map build (measured: 4 - 9 % of the profiled pass), so the map
  // collects the factor and delivers the units when the count is known (kernelmap_pairs, cs_kernelmap_free)
  double flop = 0.0;
  if (prof_enabled()) {
    const double per_pair = 2.0 * (double)cin * (double)cout;
    if (!km)
      flop = per_pair * (double)n_out;
    else if (km->num_pairs >= 0 || !km->cnt_ready || hipEventQuery(km->cnt_ready) == hipSuccess)
      flop = per_pair * (double)kernelmap_pairs(km);
    else
      kernelmap_defer_prof(const_cast<cs_kernelmap*>(km), per_pair);
  }
  ProfScope prof("conv", s, flop);
  const bool mfma_ok = (cin % 32 == 0) && (cout % 4 == 0) && (ld_in % 4 == 0) &&
                       aligned16(d_in) && aligned16(d_w);
  // production path: LDS-DMA staged kernel with per-row-group offset skipping (k_conv_dma); CS_CONV_DMA=0 falls
  // back to the round-1/2 register-staged kernel (k_conv_mfma), CS_CONV_CFG=<RG><CG><NT> forces one tile shape
  const bool dma_on = !(getenv("CS_CONV_DMA") && getenv("CS_CONV_DMA")[0] == '0');
  const int dma_cfg = getenv("CS_CONV_CFG") ? atoi(getenv("CS_CONV_CFG")) : 0;
  // the DMA kernel addresses rows and weights with 32-bit byte offsets below DMA_OOB and multiplies row
  // indices as 24-bit integers
  const int64_t in_bytes64 = n_in * (int64_t)ld_in * 4, w_bytes64 = (int64_t)kvol * cin * cout * 4;
  const bool dma_ok = dma_on && mfma_ok && cout % 32 == 0 && kvol <= 27 && (!km || (nbr_t && gmask)) && in_bytes64 < (1LL << 31) &&
                      w_bytes64 < (1LL << 31) && n_in < (1LL << 24) && (int64_t)ld_in * 4 < (1LL << 24) &&
                      n_out * (int64_t)kvol * 4 < (1LL << 32);
  const unsigned in_bytes = (unsigned)in_bytes64, w_bytes = (unsigned)w_bytes64;
  // CS_CONV_SPLIT=3 / 2: the bf16-piece experiment (k_conv_split), read per call so that a report can switch it
  const int split_ns = getenv("CS_CONV_SPLIT") ? atoi(getenv("CS_CONV_SPLIT")) : 0;
  CS_REQUIRE(split_ns == 0 || split_ns == 2 || split_ns == 3, CS_ERR_INVALID, "cs_conv_fwd: CS_CONV_SPLIT must be 2 or 3");
  if (cin == 1 && cout == 32 && kvol == 27 && nbr && n_in >= 1 && dma_on) {
    hipLaunchKernelGGL((k_conv_stem<32>), dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, s, nbr, n_out, d_in,
                       ld_in, d_w, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out);
  } else if (dma_ok && split_ns && nbr_t) {   // (1x1 layers stay on the exact kernel: measured slower in pieces)
    // experiment (see k_conv_split): bf16 matrix cores, weights cut once per layer and kept by pointer
    const uint16_t* wq = nullptr;
    const size_t wq_bytes64 = (size_t)kvol * cin * cout * split_ns * 2;
    CS_REQUIRE(wq_bytes64 < (1ULL << 31), CS_ERR_INVALID, "cs_conv_fwd: split weights too large");
    const int64_t w_total = (int64_t)kvol * cin * cout;
    auto cut = [&](uint16_t* buf) {
      if (split_ns == 3)
        hipLaunchKernelGGL(k_split_weights<3>, dim3((unsigned)ceil_div(w_total, 256)), dim3(256), 0, s, d_w, kvol, cin, cout, buf);
      else
        hipLaunchKernelGGL(k_split_weights<2>, dim3((unsigned)ceil_div(w_total, 256)), dim3(256), 0, s, d_w, kvol, cin, cout, buf);
    };
    // default: the weights are cut on every call into stream-ordered scratch (~10 us).  CS_CONV_SPLIT_CACHE=1 keeps the
    // pieces per weight POINTER for the life of the process -- only valid while the caller keeps those weights alive
    // and unchanged (an engine's parameters; tools/conv_split_report.py) -- cs_conv_split_reset() drops them.
    PoolBuf<uint16_t> wq_scratch;
    if (getenv("CS_CONV_SPLIT_CACHE") && getenv("CS_CONV_SPLIT_CACHE")[0] == '1') {
      std::lock_guard<std::mutex> lock(g_split_mu);
      SplitKey key{d_w, kvol, cin, cout, split_ns};
      auto it = g_split_w.find(key);
      if (it == g_split_w.end()) {
        uint16_t* buf = nullptr;
        CS_HIP_CHECK(hipMalloc(&buf, wq_bytes64));
        cut(buf);
        CS_LAUNCH_CHECK();
        CS_HIP_CHECK(hipStreamSynchronize(s));   // other streams may use the cached pieces right away
        it = g_split_w.emplace(key, buf).first;
      }
      wq = it->second;
    } else {
      pool_use_stream(s);
      CS_REQUIRE(wq_scratch.alloc(wq_bytes64 / 2), CS_ERR_HIP, "cs_conv_fwd: out of memory for the split weights");
      cut(wq_scratch.p);
      CS_LAUNCH_CHECK();
      wq = wq_scratch.p;
    }
    const int rev_order = 1;
    const int sdbg = getenv("CS_CONV_SPLIT_DBG") ? atoi(getenv("CS_CONV_SPLIT_DBG")) : 0;   // timing probes only (wrong results)
    int cfg = getenv("CS_CONV_SPLIT_CFG") ? atoi(getenv("CS_CONV_SPLIT_CFG")) : 0;
    if ((cfg == 412 || cfg == 221) && cout % 64) cfg = 0;
    if ((cfg == 222 || cfg == 141) && cout % 128) cfg = 0;
    // the tile shapes of the exact kernel (CS_CONV_SPLIT_CFG sweep on the stress batch, three pieces, whole forward: these
    // 5.1 - 5.2 ms, 2x2x1 everywhere 4.9 - 5.1, 4x1x2 6.4, 4x1x1 6.6; exact chain 6.0: profiles/r4e_conv_split_cfg_sweep.txt)
    if (!cfg) cfg = cout % 128 == 0 ? 141 : (cout % 64 == 0 ? 221 : 411);
#define CS_SPLIT_LAUNCH(RG, CG, NT)                                                                            \
  do {                                                                                                          \
    const dim3 grid((unsigned)ceil_div(n_out, 32 * RG), (unsigned)(cout / (32 * NT * CG)));                     \
    if (split_ns == 3)                                                                                          \
      hipLaunchKernelGGL((k_conv_split<RG, CG, NT, 3, true>), grid, dim3(256), 0, s, nbr_t, rowlist, gmask, kvol, n_out, d_in, \
                         ld_in, in_bytes, cin, wq, cout, (unsigned)wq_bytes64, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out, rev_order, sdbg); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_conv_split<RG, CG, NT, 2, true>), grid, dim3(256), 0, s, nbr_t, rowlist, gmask, kvol, n_out, d_in, \
                         ld_in, in_bytes, cin, wq, cout, (unsigned)wq_bytes64, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out, rev_order, sdbg); \
  } while (0)
    switch (cfg) {
      case 412: CS_SPLIT_LAUNCH(4, 1, 2); break;
      case 221: CS_SPLIT_LAUNCH(2, 2, 1); break;
      case 222: CS_SPLIT_LAUNCH(2, 2, 2); break;
      case 141: CS_SPLIT_LAUNCH(1, 4, 1); break;
      default: CS_SPLIT_LAUNCH(4, 1, 1); break;
    }
#undef CS_SPLIT_LAUNCH
  } else if (dma_ok) {
#define CS_DMA_LAUNCH(RG, CG, NT)                                                                              \
  do {                                                                                                          \
    const dim3 grid((unsigned)ceil_div(n_out, 32 * RG), (unsigned)(cout / (32 * NT * CG)));                     \
    if (trace)                                                                                                  \
      hipLaunchKernelGGL((k_conv_dma<RG, CG, NT, true, true>), grid, dim3(256), 0, s, nbr_t, rowlist, gmask, kvol, n_out, d_in, \
                         ld_in, in_bytes, cin, d_w, cout, w_bytes, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out, trace, rev_order); \
    else if (nbr_t)                                                                                             \
      hipLaunchKernelGGL((k_conv_dma<RG, CG, NT, true, false>), grid, dim3(256), 0, s, nbr_t, rowlist, gmask, kvol, n_out, d_in, \
                         ld_in, in_bytes, cin, d_w, cout, w_bytes, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out, trace, rev_order); \
    else                                                                                                        \
      hipLaunchKernelGGL((k_conv_dma<RG, CG, NT, false, false>), grid, dim3(256), 0, s, nbr_t, rowlist, gmask, kvol, n_out, d_in, \
                         ld_in, in_bytes, cin, d_w, cout, w_bytes, d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out, trace, rev_order); \
  } while (0)
    // CS_CONV_TRACE=1: per-wave phase cycles of this launch, summed and printed (diagnostics; synchronises)
    unsigned long long* trace = nullptr;
    const size_t trace_n = (size_t)ceil_div(n_out, 32) * (size_t)(cout / 32) * 4 * 8;
    if (nbr_t && getenv("CS_CONV_TRACE") && getenv("CS_CONV_TRACE")[0] == '1') {
      if (hipMalloc(&trace, trace_n * 8) != hipSuccess) trace = nullptr;
      if (trace) (void)hipMemsetAsync(trace, 0, trace_n * 8, s);
    }
    // (a persistent-workgroup variant with next-tile prefetch, k_conv_dma_p, was measured neutral in round 3 -- the
    // registers the prefetch holds cost the resident workgroup per CU it was meant to make unnecessary -- and was
    // removed in round 4: HISTORY.md 7c)
    const int rev_order = !(getenv("CS_CONV_FWD_ORDER") && getenv("CS_CONV_FWD_ORDER")[0] == '1');
    int cfg = dma_cfg;
    const int64_t t128 = ceil_div(n_out, 128), t64 = ceil_div(n_out, 64);
    if (cfg == 412 && cout % 64) cfg = 0;
    if ((cfg == 221) && cout % 64) cfg = 0;
    if ((cfg == 222 || cfg == 141 || cfg == 414) && cout % 128) cfg = 0;
    if (!cfg) {
      // 128-row tiles (4 row groups share a weight slab) while they still give every CU several workgroups,
      // 64- and 32-row tiles for the coarse levels
      // measured per layer on the stress and chair batches (CS_CONV_CFG sweep, profiles/r3t_conv_cfg_sweep.txt):
      // the shapes with the smallest LDS footprint win -- 64 x 64 (32 KB, 5 workgroups per CU) for Cout = 64 n
      // (304 vs 347 us for 4 x 1 x 2 on the stride-2 64 -> 64 layers, 477 vs 525 at stride 1), 32 x 128 (40 KB) for
      // Cout = 128 n (374 - 394 vs 446 us for 2 x 2 x 2): the kernel is limited by how many workgroups hide each
      // other's barriers, prologues and epilogues, not by the MFMAs a DMA instruction feeds
      (void)t128;
      (void)t64;
      if (cout % 128 == 0)
        cfg = 141;
      else if (cout % 64 == 0)
        cfg = 221;
      else
        cfg = 411;
    }
    switch (cfg) {
      case 412: CS_DMA_LAUNCH(4, 1, 2); break;
      case 414: CS_DMA_LAUNCH(4, 1, 4); break;
      case 221: CS_DMA_LAUNCH(2, 2, 1); break;
      case 222: CS_DMA_LAUNCH(2, 2, 2); break;
      case 141: CS_DMA_LAUNCH(1, 4, 1); break;
      default: CS_DMA_LAUNCH(4, 1, 1); break;
    }
#undef CS_DMA_LAUNCH
    if (trace) {
      std::vector<unsigned long long> h(trace_n);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h.data(), trace, trace_n * 8, hipMemcpyDeviceToHost);
      (void)hipFree(trace);
      double tot = 0, vm = 0, wt = 0, cp = 0, nc = 0, na = 0, nw = 0, td = 0, pro = 0, lp = 0, epi = 0;
      unsigned long long rmin = ~0ULL, rmax = 0;
      for (size_t i = 0; i + 8 <= trace_n; i += 8)
        if (h[i]) {
          tot += h[i]; vm += h[i + 1]; wt += h[i + 2]; cp += h[i + 3]; nc += h[i + 4]; na += h[i + 5]; nw += 1;
          td += h[i + 6] & 0xffffffffffULL;
          const unsigned long long b = h[i + 7] & 0xffffffffULL, p0 = (h[i + 7] >> 32) & 0xffff, l0 = h[i + 7] >> 48,
                                   e0 = h[i + 6] >> 40;
          pro += p0; lp += l0; epi += e0;
          rmin = std::min(rmin, b); rmax = std::max(rmax, b + p0 + l0 + e0);
        }
      fprintf(stderr, "[conv trace] wave lifetime (us): prologue %.2f loop %.2f epilogue %.2f; kernel span %.1f us; mean resident "
              "workgroups per CU %.2f\n", pro / nw / 100, lp / nw / 100, epi / nw / 100, (rmax - rmin) / 100.0,
              (pro + lp + epi) / 4 / (double)(rmax - rmin) / 256);
      fprintf(stderr, "[conv trace] cfg %d n_out %lld %d->%d waves %.0f: per wave loop %.0f cyc = vmcnt %.0f + barrier %.0f + body %.0f; "
              "chunks %.1f active %.1f; per chunk: vmcnt %.0f barrier %.0f offsets+fetch %.0f body %.0f\n",
              cfg, (long long)n_out, cin, cout, nw, tot / nw, vm / nw, wt / nw, cp / nw, nc / nw, na / nw, vm / nc, wt / nc,
              td / nc, cp / nc);
    }
  } else if (mfma_ok) {
    // 64 x 128 tiles unless that leaves half of the 256 CUs without a workgroup (coarsest level)
    const bool short_tiles = !(getenv("CS_CONV_TILE") && getenv("CS_CONV_TILE")[0] == '0');
    if (cout % 128 == 0 && short_tiles && ceil_div(n_out, 64) * (cout / 128) <= 512) {
      // few rows (stride 4 / 8 levels of a 32-cloud batch): 32-row x 128-column tiles double the workgroups
      // and narrow the union of offsets a tile has to walk (stride-4 layers 182 -> 165 us, conv4_tr 133 -> 100;
      // CS_CONV_TILE=0: the 64-row tiles)
      dim3 grid((unsigned)ceil_div(n_out, 32), (unsigned)(cout / 128));
      hipLaunchKernelGGL((k_conv_mfma<1, 4, 1>), grid, dim3(256), 0, s, nbr, rowlist, kvol, n_in, n_out,
                         d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu,
                         d_out, ld_out);
    } else if (cout % 128 == 0 && ceil_div(n_out, 64) * (cout / 128) > 128) {
      dim3 grid((unsigned)ceil_div(n_out, 64), (unsigned)(cout / 128));
      hipLaunchKernelGGL((k_conv_mfma<2, 2, 2>), grid, dim3(256), 0, s, nbr, rowlist, kvol, n_in, n_out,
                         d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu,
                         d_out, ld_out);
    } else if (cout > 32) {
      dim3 grid((unsigned)ceil_div(n_out, 64), (unsigned)ceil_div(cout, 64));
      hipLaunchKernelGGL((k_conv_mfma<2, 2, 1>), grid, dim3(256), 0, s, nbr, rowlist, kvol, n_in, n_out,
                         d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu,
                         d_out, ld_out);
    } else {
      dim3 grid((unsigned)ceil_div(n_out, 128), 1);
      hipLaunchKernelGGL((k_conv_mfma<4, 1, 1>), grid, dim3(256), 0, s, nbr, rowlist, kvol, n_in, n_out,
                         d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu,
                         d_out, ld_out);
    }
  } else {
    const int64_t total = n_out * cout;
    hipLaunchKernelGGL(k_conv_generic, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, nbr,
                       kvol, n_out, d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual,
                       ld_res, relu, d_out, ld_out);
  }
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_conv_split_reset(void) {
  std::lock_guard<std::mutex> lock(g_split_mu);
  for (auto& kv : g_split_w) (void)hipFree(kv.second);
  g_split_w.clear();
  return CS_OK;
}

int cs_affine_act(int64_t n, int c, const float* d_in, int ld_in, const float* d_scale,
                  const float* d_shift, const float* d_residual, int ld_res, int relu,
                  float* d_out, int ld_out, void* stream) {
  CS_REQUIRE(d_in && d_out && c >= 1 && ld_in >= c && ld_out >= c, CS_ERR_INVALID,
             "cs_affine_act: bad argument");
  CS_REQUIRE(!d_scale || d_shift, CS_ERR_INVALID, "cs_affine_act: scale without shift");
  if (n == 0) return CS_OK;
  int64_t total = n * c;
  unsigned g = (unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
  hipLaunchKernelGGL(k_affine_act, dim3(g), dim3(256), 0, (hipStream_t)stream, n, c, d_in, ld_in,
                     d_scale, d_shift, d_residual, ld_res, relu, d_out, ld_out);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_row_l2_normalize(int64_t n, int c, const float* d_in, int ld_in, float eps, float* d_out,
                        int ld_out, void* stream) {
  CS_REQUIRE(d_in && d_out && c >= 1 && ld_in >= c && ld_out >= c, CS_ERR_INVALID,
             "cs_row_l2_normalize: bad argument");
  if (n == 0) return CS_OK;
  if (c <= 16)
    hipLaunchKernelGGL(k_row_l2norm16, dim3((unsigned)ceil_div(n * 16, 256)), dim3(256), 0, (hipStream_t)stream, n, c, d_in,
                       ld_in, eps, d_out, ld_out);
  else
    hipLaunchKernelGGL(k_row_l2norm, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0,
                       (hipStream_t)stream, n, c, d_in, ld_in, eps, d_out, ld_out);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_segmented_max(int64_t n, int c, const float* d_in, int ld_in, const int32_t* d_batch,
                     int batch_ld, int n_batch, float* d_out, void* stream) {
  CS_REQUIRE(d_in && d_batch && d_out && c >= 1 && n_batch >= 0 && batch_ld >= 1,
             CS_ERR_INVALID, "cs_segmented_max: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int64_t on = (int64_t)n_batch * c;
  if (on == 0) return CS_OK;
  unsigned* obuf = reinterpret_cast<unsigned*>(d_out);
  hipLaunchKernelGGL(k_segmax_init, dim3((unsigned)ceil_div(on, 256)), dim3(256), 0, s, obuf, on);
  if (n > 0)
    hipLaunchKernelGGL(k_segmax_runs, dim3((unsigned)ceil_div(n, SEGMAX_ROWS), (unsigned)ceil_div(c, 256)), dim3(256), 0, s,
                       n, c, d_in, ld_in, d_batch, batch_ld, n_batch, obuf);
  hipLaunchKernelGGL(k_segmax_fin, dim3((unsigned)ceil_div(on, 256)), dim3(256), 0, s, obuf, on);
  CS_LAUNCH_CHECK();
  return CS_OK;
}

int cs_instance_norm(int64_t n, int c, const float* d_in, int ld_in, const int32_t* d_seg, int n_batch,
                     const float* d_weight, const float* d_bias, float eps, float* d_out, int ld_out,
                     void* stream) {
  CS_REQUIRE(d_in && d_out && d_seg && c >= 1 && ld_in >= c && ld_out >= c && n_batch >= 0 && n >= 0 &&
                 n < (1LL << 31) && eps >= 0.f,
             CS_ERR_INVALID, "cs_instance_norm: bad argument");
  if (n == 0 || n_batch == 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  pool_use_stream(s);
  const int64_t slots = n / INORM_CHUNK + n_batch + 1;
  PoolBuf<double> partial((size_t)slots * c);
  PoolBuf<float> mean((size_t)n_batch * c), inv_std((size_t)n_batch * c);
  CS_REQUIRE(partial.p && mean.p && inv_std.p, CS_ERR_HIP, "cs_instance_norm: scratch allocation failed");
  const dim3 pg(INORM_SLICES, (unsigned)n_batch, (unsigned)ceil_div(c, 256));
  const unsigned sg = (unsigned)ceil_div((int64_t)n_batch * c, 256);
  hipLaunchKernelGGL(k_inorm_partial<0>, pg, dim3(256), 0, s, d_in, ld_in, c, d_seg, (const float*)nullptr,
                     partial.p);
  hipLaunchKernelGGL(k_inorm_stat<0>, dim3(sg), dim3(256), 0, s, partial.p, c, n_batch, d_seg, eps, mean.p);
  hipLaunchKernelGGL(k_inorm_partial<1>, pg, dim3(256), 0, s, d_in, ld_in, c, d_seg, mean.p, partial.p);
  hipLaunchKernelGGL(k_inorm_stat<1>, dim3(sg), dim3(256), 0, s, partial.p, c, n_batch, d_seg, eps, inv_std.p);
  hipLaunchKernelGGL(k_inorm_apply, dim3(64, (unsigned)n_batch), dim3(256), 0, s, d_in, ld_in, c, n_batch,
                     d_seg, mean.p, inv_std.p, d_weight, d_bias, d_out, ld_out);
  CS_LAUNCH_CHECK();
  return CS_OK;  // no synchronisation: the scratch returns to this thread's stream-ordered cache
}

}  // extern "C"

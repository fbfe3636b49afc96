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
// k_conv_lacc: sparse convolution with accumulators in LDS and per-offset ROW COMPACTION (round 2).
//
// k_conv_mfma keeps a tile's accumulators in registers, so every row of a tile pays the MFMA time of
// every offset the TILE uses: x1.6 - 2.2 of the useful work on real occupancy (a voxel has ~10 of its 27
// neighbours).  Here a WAVE owns 64 output rows x TN channels whose accumulators live in its private
// slice of LDS; for every offset k the rows that have neighbour k are compacted (ballot + mbcnt: lane =
// row) and only they go through the matrix pipe, 32 at a time:
//     C  <- LDS accumulator rows of the group      (v_mfma_f32_32x32x2_f32 accumulator INPUT)
//     C  <- the fma chain over the Cin channels    (same chain, same order: bit-identical results)
//     LDS <- C
// Executed / useful MFMA work on the bench clouds: 1.2 - 1.6 (mask-sorted rows), against 1.6 - 2.6.
// No workgroup barrier anywhere: the four waves of a workgroup are independent (LDS is wave-private,
// LDS operations of one wave execute in order), so a wave never waits for its neighbours' gathers.
// Operands come straight from global memory / L2, no LDS staging:
//   * A (gathered input rows): lane (half h, row j) loads channels [16 h, 16 h + 16) of its row with four
//     16-byte loads; MFMA i of the 32-channel chunk needs channel 2 i + h in lane (h, j), which is what ONE
//     v_permlane32_swap per register pair produces: swap(X[2m], X[2m+1]) leaves (ch 2m | ch 2m+1) in the
//     first register = operand of MFMA m, and (ch 16+2m | ch 17+2m) in the second = operand of MFMA 8+m.
//   * B (weights W[k][ci][co], reference layout): lane (h, c) reads W[k][ci0 + 2 i + h][n0 + 32 t + c]:
//     two coalesced 128-byte rows per MFMA, L2-resident (a layer's weights are 27 Cin Cout 4 B <= 7 MB).
// The next chunk's A and B registers are loaded while the current chunk's MFMAs run.
// ------------------------------------------------------------------------------------------------
#ifndef CONV_DBG
#define CONV_DBG 0   // timing experiments only (tools/conv_dbg.sh): 1 no A loads, 2 no B loads, 4 no MFMAs
#endif
template <int NT>
__global__ __launch_bounds__(256) void k_conv_lacc(
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ rowlist, int kvol, int64_t n_out,
    const float* __restrict__ in, int ld_in, int cin, const float* __restrict__ w, int cout,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, int ld_res, int relu, float* __restrict__ out, int ld_out) {
  constexpr int TN = 32 * NT;
  constexpr int R = 64;                    // output rows per wave (lane = row)
  constexpr int DUMMY = R;                 // accumulator row of the padded slots of a group
  // dynamic LDS: 4 waves x ((R + 1) x TN floats + 2 lists x R uint16) = 66 KiB for TN = 64
  extern __shared__ __attribute__((aligned(16))) char lacc_lds[];
  constexpr int ROWB = TN * 4;             // bytes per accumulator row
  constexpr int WAVE_BYTES = (R + 1) * ROWB + 2 * R * 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (row0 >= n_out) return;               // whole wave; no barriers below
  const int n0 = blockIdx.y * TN;
  char* ACCB = lacc_lds + wave * WAVE_BYTES;                              // accumulators, byte-addressed
  unsigned short* LIST = reinterpret_cast<unsigned short*>(ACCB + (R + 1) * ROWB);   // [2][R] byte offsets of rows

  // this lane's output row, its 27 neighbour rows (registers) and their presence mask
  int o = -1;
  if (row0 + lane < n_out) o = rowlist ? rowlist[row0 + lane] : (int)(row0 + lane);
  int nb[27];
  unsigned mask = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    nb[k] = (o >= 0 && k < kvol) ? nbr[(int64_t)o * kvol + k] : -1;
    mask |= (nb[k] >= 0 ? 1u : 0u) << k;
  }
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < TN / 4; ++q) *reinterpret_cast<float4*>(ACCB + lane * ROWB + 16 * q) = z;
    if (lane < TN / 4) *reinterpret_cast<float4*>(ACCB + DUMMY * ROWB + 16 * lane) = z;
  }
  const int cchunks = cin >> 5;

  // first offset >= from that some row of this wave has (wave-uniform), 27 = none
  auto next_k = [&](int from) {
    int kk = from;
    while (kk < kvol && __ballot((mask >> kk) & 1u) == 0ULL) ++kk;
    return kk < kvol ? kk : 27;
  };
  // compacted list of the rows that have offset kk, as byte offsets of their accumulator rows, padded with the
  // DUMMY row to whole groups of 32 -> LIST[buf][0 .. ceil32(count)); nbk = this lane's neighbour for kk
  auto build_list = [&](int kk, int buf, int& nbk) {
    const bool has = (mask >> kk) & 1u;
    const unsigned long long bal = __ballot(has);
    const int cnt = (int)__popcll(bal);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    if (has) LIST[buf * R + pos] = (unsigned short)(lane * ROWB);
    if (lane >= cnt && lane < ((cnt + 31) & ~31)) LIST[buf * R + lane] = (unsigned short)(DUMMY * ROWB);
    nbk = nb[0];
#pragma unroll
    for (int q = 1; q < 27; ++q) nbk = kk == q ? nb[q] : nbk;   // uniform selects, once per offset
    __builtin_amdgcn_wave_barrier();       // (compiler ordering only: same-wave LDS ops execute in order)
    return cnt;
  };
  // input row this lane gathers for group (g0 of the list in buf): compacted slot g0 + c (both halves: the
  // same row).  Padded slots read some valid row: their products land in the DUMMY accumulator row and are
  // never used, so the loads need no predicate (an exec-masked load puts a branch and a register merge into
  // the pipeline).
  auto prep = [&](int nbk, int g0, int buf) -> const float* {
    const int rho = (int)LIST[buf * R + g0 + c] / ROWB;   // owner lane of the row (DUMMY -> 64 -> lane 0)
    const int src = max(__shfl(nbk, rho), 0);
    return in + (int64_t)src * ld_in + 16 * h;
  };
  auto load_a = [&](const float* p4, float4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dst[q] = (CONV_DBG & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(p4 + 4 * q);
  };
  // B operand: uniform base (SGPRs) + one per-lane 32-bit offset shared by every load
  const int loff = h * cout + c;
  auto wslab = [&](int kk, int cc) { return w + ((int64_t)kk * cin + cc * 32) * cout + n0; };   // wave-uniform

  int k = next_k(0);
  if (k < 27) {
    int buf = 0, g0 = 0, nbk = 0, nnbk = 0;
    int mk = build_list(k, buf, nbk);
    const float* arow = prep(nbk, g0, buf);
    float4 x[4];
    float b[16][NT];
    load_a(arow, x);
    {
      const float* wu = wslab(k, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[i][t] = (CONV_DBG & 2) ? 1.0f : (wu + (int64_t)(2 * i) * cout + 32 * t)[loff];
    }
    bool more = true;
    while (more) {
      // LDS byte addresses of the accumulator rows of this lane's 16 result registers: slots
      // (i & 3) + 8 (i >> 2) + 4 h of the group, column c
      unsigned addr[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 pr = *reinterpret_cast<const uint2*>(&LIST[buf * R + g0 + 8 * q + 4 * h]);
        addr[4 * q + 0] = (pr.x & 0xffffu) + 4 * c;
        addr[4 * q + 1] = (pr.x >> 16) + 4 * c;
        addr[4 * q + 2] = (pr.y & 0xffffu) + 4 * c;
        addr[4 * q + 3] = (pr.y >> 16) + 4 * c;
      }
      f32x16 acc[NT];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][i] = *reinterpret_cast<const float*>(ACCB + addr[i] + 128 * t);
      // the group after this one: its first operands are requested under this group's last chunk
      int nk = k, ng0 = g0 + 32, nmk = mk, nbuf = buf;
      nnbk = nbk;
      if (ng0 >= mk) {
        nk = next_k(k + 1);
        ng0 = 0;
        nbuf = buf ^ 1;
        if (nk < 27) nmk = build_list(nk, nbuf, nnbk);
      }
      more = nk < 27;
      const int kn = more ? nk : k;         // (after the last group: harmless re-reads of valid addresses)
      const float* narow = more ? prep(nnbk, ng0, nbuf) : arow;
      for (int cc = 0; cc < cchunks; ++cc) {
        float a[16];
        {
          const float X[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w,
                               x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[2 * m]), __float_as_uint(X[2 * m + 1]),
                                                             false, false);
            a[m] = __uint_as_float(sw[0]);
            a[8 + m] = __uint_as_float(sw[1]);
          }
        }
        // operands of the chunk after this one (next chunk of the group, or first chunk of the next group):
        // one branch-free set of loads; every B register is reloaded right behind the MFMA that consumed it
        const bool last = cc + 1 == cchunks;
        const float* pa = last ? narow : arow + (cc + 1) * 32;
        const float* wu = last ? wslab(kn, 0) : wslab(k, cc + 1);
        load_a(pa, x);
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (CONV_DBG & 4)
              acc[t][i] += a[i] * b[i][t];
            else
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i][t], acc[t], 0, 0, 0);
            b[i][t] = (CONV_DBG & 2) ? 1.0f : (wu + (int64_t)(2 * i) * cout + 32 * t)[loff];
          }
        // keep that order: one MFMA, then the reload of the register it consumed (hipcc otherwise sinks all the
        // loads below the MFMA block, where the next chunk waits a full L2 latency for them)
#pragma unroll
        for (int j = 0; j < 16 * NT; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<float*>(ACCB + addr[i] + 128 * t) = acc[t][i];
      __builtin_amdgcn_wave_barrier();
      k = nk;
      g0 = ng0;
      mk = nmk;
      buf = nbuf;
      nbk = nnbk;
      arow = narow;
    }
  }

  // epilogue: row-major LDS -> coalesced global rows (64 / TN rows per pass)
  {
    constexpr int RPP = 64 / TN;            // rows per pass
    const int col = n0 + (lane & (TN - 1));
    const int sub = lane / TN;
    const float sc = scale ? scale[col] : 1.0f;
    const float sh = shift ? shift[col] : 0.0f;
    for (int r0 = 0; r0 < R; r0 += RPP) {
      const int r = r0 + sub;
      const int o_r = __shfl(o, r);
      if (o_r < 0) continue;
      float v = *reinterpret_cast<const float*>(ACCB + r * ROWB + 4 * (lane & (TN - 1)));
      if (scale)
        v = __fmaf_rn(v, sc, sh);
      else if (shift)
        v = v + sh;
      if (residual) v = v + residual[(int64_t)o_r * ld_res + col];
      if (relu) v = fmaxf(v, 0.0f);
      out[(int64_t)o_r * ld_out + col] = v;
    }
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
__global__ void k_segmax(int64_t n, int c, const float* __restrict__ in, int ld_in,
                         const int32_t* __restrict__ batch, int batch_ld, int n_batch,
                         unsigned* obuf) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  int64_t r = t / c;
  int col = (int)(t - r * c);
  int b = batch[r * batch_ld];
  if (b < 0 || b >= n_batch) return;
  atomicMax(&obuf[(int64_t)b * c + col], f2ord(in[r * ld_in + col]));
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
  if (km) {
    CS_REQUIRE(km->n_out == n_out && km->n_in == n_in, CS_ERR_INVALID,
               "cs_conv_fwd: kernel map is for %lld -> %lld rows, tensors have %lld -> %lld",
               (long long)km->n_in, (long long)km->n_out, (long long)n_in, (long long)n_out);
    kvol = km->kvol;
    nbr = km->d_nbr;
    rowlist = km->d_rowlist;
  } else {
    CS_REQUIRE(n_in == n_out, CS_ERR_INVALID, "cs_conv_fwd: 1x1 conv needs n_in == n_out");
  }
  if (n_out == 0) return CS_OK;
  hipStream_t s = (hipStream_t)stream;
  // (the pair count of a freshly built map is resolved here only when profiling is on)
  const double flop = prof_enabled() ? 2.0 * (double)(km ? kernelmap_pairs(km) : n_out) * (double)cin * (double)cout : 0.0;
  ProfScope prof("conv", s, flop);
  const bool mfma_ok = (cin % 32 == 0) && (cout % 4 == 0) && (ld_in % 4 == 0) &&
                       aligned16(d_in) && aligned16(d_w);
  // CS_CONV_LACC=1: the LDS-accumulator kernel with per-offset row compaction (k_conv_lacc).  Bit-identical
  // results, 1.2 - 1.6x fewer MFMAs, but measured SLOWER than the register-accumulator kernel below on the
  // bench shapes (DESIGN.md "what was tried on the sparse convolution"): operands fetched per 32-row group
  // straight from L2 cost more than the LDS-staged 64-row tiles save.  Off by default, kept under test.
  const bool lacc_on = getenv("CS_CONV_LACC") && getenv("CS_CONV_LACC")[0] == '1';
  if (mfma_ok && lacc_on && nbr && kvol > 1 && cout % 32 == 0) {
    const unsigned wgs = (unsigned)ceil_div(ceil_div(n_out, 64), 4);
    constexpr int LDS2 = 4 * (65 * 64 * 4 + 256), LDS1 = 4 * (65 * 32 * 4 + 256);
    static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_lacc<2>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    CS_REQUIRE(attr2 == hipSuccess, CS_ERR_HIP, "cs_conv_fwd: cannot reserve %d bytes of LDS", LDS2);
    CS_REQUIRE(kvol <= 27, CS_ERR_UNSUPPORTED, "cs_conv_fwd: kernel volume %d", kvol);
    // 64-column tiles halve the gather traffic, 32-column tiles double the number of waves: the wide form only
    // when it still gives every SIMD several waves (CS_CONV_NT=1/2 forces one, for experiments)
    const int force_nt = getenv("CS_CONV_NT") ? atoi(getenv("CS_CONV_NT")) : 0;
    const bool wide = force_nt ? force_nt == 2 : (int64_t)ceil_div(n_out, 64) * (cout / 64) >= 4096;
    if (cout % 64 == 0 && wide)
      hipLaunchKernelGGL((k_conv_lacc<2>), dim3(wgs, (unsigned)(cout / 64)), dim3(256), LDS2, s, nbr, rowlist, kvol,
                         n_out, d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu, d_out,
                         ld_out);
    else
      hipLaunchKernelGGL((k_conv_lacc<1>), dim3(wgs, (unsigned)(cout / 32)), dim3(256), LDS1, s, nbr, rowlist, kvol,
                         n_out, d_in, ld_in, cin, d_w, cout, d_scale, d_shift, d_residual, ld_res, relu, d_out,
                         ld_out);
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
    hipLaunchKernelGGL(k_segmax, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, s, n, c,
                       d_in, ld_in, d_batch, batch_ld, n_batch, obuf);
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

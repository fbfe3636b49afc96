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
  if (mfma_ok) {
    // 64 x 128 tiles unless that leaves half of the 256 CUs without a workgroup (coarsest level)
    if (cout % 128 == 0 && ceil_div(n_out, 64) * (cout / 128) > 128) {
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

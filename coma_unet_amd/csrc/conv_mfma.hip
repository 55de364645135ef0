// bf16 MFMA implicit-GEMM convolution kernels for gfx950 (CDNA4, wave64).
//
//   y[m][n] = sum_tap sum_c x[pos(m, tap)][c] * wk[tap][n][c]      (fp32 accumulate)
//
// GEMM view: M = output voxels (one sample per blockIdx.z), N = output channels, K = taps x C.
// Operands are staged through LDS as [row][32 channels] bf16 tiles (64-byte rows, 16-byte chunks
// XOR-swizzled by (row >> 2) & 3 so that the ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16
// are bank-conflict free), double-buffered with register prefetch of the next K-step.
//
// Replaces the cuDNN dispatches behind nn.Conv3d / nn.ConvTranspose3d forward and data-gradient
// in the reference model (attn_unet_data_parallel.py; MONAI Convolution / CondConv call sites).
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator fragment

// ---- element type of the activations / kernel-layout weights: bf16 (v_mfma_f32_32x32x16_bf16) or fp32
// (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, 1/16 of the bf16 rate -- the "fp32 mode" of the model,
// north_star's 1e-3 tolerance path).  Both use the SAME LDS images: a row is 64 bytes = 4 pieces of 16 bytes
// (32 bf16 or 16 fp32 channels), and the 16-byte fragment a lane reads feeds one bf16 MFMA (K = 16: lane half h holds
// k = 8h..8h+7) or four fp32 MFMAs (K = 2 each: MFMA m takes channel 4h + m of the piece from lane half h, so the
// K pair of MFMA m is {m, 4 + m} of the 8 channels the two halves read -- the same for both operands).
template <typename T> struct elem;
template <> struct elem<bf16_t> { static constexpr int EPB = 8; };    // elements per 16-byte piece
template <> struct elem<float> { static constexpr int EPB = 4; };

__device__ __forceinline__ f32x16_t mma_piece(const uint4& a, const uint4& b, f32x16_t acc, bf16_t) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&a), *reinterpret_cast<const bf16x8_t*>(&b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mma_piece(const uint4& a, const uint4& b, f32x16_t acc, float) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  return acc;
}

// fused norm statistics: a block ADDS its {sum, sumsq} of one (group, channel) to the caller's zeroed fp64 record
// rec[COMA_STAT_REPLICAS][G][N][2] (replica stride rounded up to a 64-byte line) with global_atomic_add_f64; every consumer
// sums the replicas and derives mean / rstd itself (norm.hip, NormStat).  Replicas: float atomics execute at the memory
// side at ~25 ns per request to one 64-byte line, so a few hundred blocks adding to ONE record would queue for
// microseconds at the end of the kernel; spread over 8 replicas by block index the queue per line is 8x shorter.
__device__ __forceinline__ void stat_add(double2* rec, int G, int N, int g, int n, double a, double c) {
  const long rs = ((long)G * N * 2 + 7) & ~7L;
  double* q = reinterpret_cast<double*>(rec) + (long)(blockIdx.x & (COMA_STAT_REPLICAS - 1)) * rs + ((long)g * N + n) * 2;
  unsafeAtomicAdd(q, a);
  unsafeAtomicAdd(q + 1, c);
}

struct GatherP {
  const void* x; int ldx; long sbx; int Di, Hi, Wi, C;
  void* y; int ldy; long sby; int Do, Ho, Wo, N;
  const void* w; long wsb;          // [b][tap][N][C]
  const float* bias; int bsb;
  int k, stride, flip;
  int Mz, My, Mx;                      // M-grid (MODE 0: output grid; MODE 1: coarse grid)
  int ksplit;                          // > 1: the K loop (taps x channel chunks) is cut into ksplit slices on blockIdx.y
  float* part; long part_sb;           //      whose fp32 partial tiles are merged atomically into part[b][voxel][N]
  int accum;                           // y += conv(x) instead of y = conv(x) (a data gradient added to one that is already there)
  double2* stats; int stats_inst;      // optional fused norm statistics of the stored outputs (not with ksplit > 1), as conv_mfma_halo2_k
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 2) | (chunk ^ ((row >> 2) & 3)); }  // 16-B slot index

// blockIdx -> work-item remap: the dispatcher deals consecutive blocks round-robin over the 8 XCDs
// (each with a private 4 MiB L2); give every XCD one CONTIGUOUS range of the work so that
// neighbouring tiles (which share halo voxels) hit in the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// tile id -> (tix, tiy, tiz): x slowest, then blocks of 8 z-tiles, then y, z-in-block fastest, so
// the ~64 tiles an XCD works on at once form an 8(y) x 8(z) patch (halo overlap 3.2x -> ~1.3x).
// ids run over ntx * ceil(ntz/8) * nty * 8; returns false for the padding ids (tiz >= ntz).
__device__ __forceinline__ bool tile_coords(int id, int ntx, int nty, int ntz, int& tix, int& tiy, int& tiz) {
  const int ntzb = (ntz + 7) >> 3;
  const int zin = id & 7;
  int t = id >> 3;
  tiy = t % nty; t /= nty;
  const int tzb = t % ntzb;
  tix = t / ntzb;
  tiz = tzb * 8 + zin;
  return tiz < ntz && tix < ntx;
}

// MODE 0: in = m*stride - pad + tap  (all taps; `flip` mirrors the weight tap index)
// MODE 1: stride-2 transposed gather, one output-parity class per blockIdx.y slice
template <int BN, int MODE, typename T = bf16_t>
__global__ __launch_bounds__(256, 2) void conv_mfma_gather_k(GatherP p) {
  constexpr int EPB = elem<T>::EPB, CKL = 4 * EPB, LCK = EPB == 8 ? 5 : 4;   // channels per K step (one 64-byte row) and its log2
  constexpr int BM = (BN == 128) ? 128 : 256;
  constexpr int WAVES_N = (BN == 128) ? 2 : 1;
  constexpr int NT = BN / 32 / WAVES_N;           // N-tiles per wave
  constexpr int MT = 2;                           // M-tiles per wave (64 rows)
  constexpr int A_PIECES = BM * 4 / 256;          // 16-B pieces per thread
  constexpr int B_PIECES = (BN * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ldsA = reinterpret_cast<uint4*>(smem);                       // [2][BM*4]
  uint4* ldsB = ldsA + 2 * BM * 4;                                    // [2][BN*4]
  int* rowoff = reinterpret_cast<int*>(ldsB + 2 * BN * 4);            // [BM] output element offset or -1

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int b = blockIdx.z;
  int nblk, cls = 0, ysl = blockIdx.y, ksi = 0;
  if (p.ksplit > 1) { ksi = ysl % p.ksplit; ysl /= p.ksplit; }
  if (MODE == 1) { cls = ysl & 7; nblk = ysl >> 3; } else { nblk = ysl; }
  const int n0 = nblk * BN;
  const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
  const long Mtot = (long)p.Mz * p.My * p.Mx;
  const long m0 = (long)blockIdx.x * BM;
  const T* xb = static_cast<const T*>(p.x) + (long)b * p.sbx;
  const T* wb = static_cast<const T*>(p.w) + (long)b * p.wsb;
  const int pad = (p.k - 1) >> 1;

  // ---- per-thread gather rows (fixed over the K loop) ----
  const int chunk = tid & 3;
  int rz[A_PIECES], ry[A_PIECES], rx[A_PIECES];   // base input coordinates (tap 0 / delta 0)
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int row = (tid >> 2) + 64 * i;
    const long m = m0 + row;
    if (m < Mtot) {
      const int mx = (int)(m % p.Mx), my = (int)((m / p.Mx) % p.My), mz = (int)(m / ((long)p.Mx * p.My));
      if (MODE == 0) { rz[i] = mz * p.stride - pad; ry[i] = my * p.stride - pad; rx[i] = mx * p.stride - pad; }
      else { rz[i] = mz; ry[i] = my; rx[i] = mx; }
      if (chunk == 0) {
        int off;
        if (MODE == 0) off = (int)(m * p.ldy);
        else {
          const int oz = 2 * mz + pz, oy = 2 * my + py, ox = 2 * mx + px;
          off = (oz < p.Do && oy < p.Ho && ox < p.Wo) ? (((oz * p.Ho + oy) * p.Wo + ox) * p.ldy) : -1;
        }
        rowoff[row] = off;
      }
    } else {
      rz[i] = ry[i] = rx[i] = -(1 << 20);
      if (chunk == 0) rowoff[row] = -1;
    }
  }

  // ---- K loop bookkeeping ----
  const int cchunks = p.C >> LCK;
  int ntaps;
  if (MODE == 0) ntaps = p.k * p.k * p.k; else ntaps = (1 + pz) * (1 + py) * (1 + px);
  const int nsteps_all = ntaps * cchunks;
  // split-K (deep layers: a few hundred voxels, thousands of K steps, fewer tiles than CUs): this block's slice
  const int s_begin = p.ksplit > 1 ? (int)((long)nsteps_all * ksi / p.ksplit) : 0;
  const int s_end = p.ksplit > 1 ? (int)((long)nsteps_all * (ksi + 1) / p.ksplit) : nsteps_all;
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;        // (block-uniform; the host keeps ksplit below every class's step count)

  uint4 ra[A_PIECES], rb[B_PIECES];
  // K-step counters advanced incrementally (no integer division in the loop)
  const int l_nx = MODE == 0 ? p.k : 1 + px, l_ny = MODE == 0 ? p.k : 1 + py;
  int l_t = s_begin / cchunks, l_cc = s_begin % cchunks;
  int l_jx = l_t % l_nx, l_jy = (l_t / l_nx) % l_ny, l_jz = l_t / (l_nx * l_ny);
  auto load_step = [&](int) {
    const int t = l_t, cc = l_cc;
    int dz, dy, dx, wtap;
    if (MODE == 0) {
      dz = l_jz; dy = l_jy; dx = l_jx;
      wtap = p.flip ? (ntaps - 1 - t) : t;
    } else {
      // per dim: parity 0 -> tap 1, delta 0 ; parity 1 -> j=0: tap 0, delta +1 ; j=1: tap 2, delta 0
      const int jx = l_jx, jy = l_jy, jz = l_jz;
      const int tx = px ? (jx ? 2 : 0) : 1, ty = py ? (jy ? 2 : 0) : 1, tz = pz ? (jz ? 2 : 0) : 1;
      dx = px ? (jx ? 0 : 1) : 0; dy = py ? (jy ? 0 : 1) : 0; dz = pz ? (jz ? 0 : 1) : 0;
      wtap = (tz * 3 + ty) * 3 + tx;
    }
    const int c0 = cc * CKL + chunk * EPB;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int iz = rz[i] + dz, iy = ry[i] + dy, ix = rx[i] + dx;
      const bool ok = (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (ok) ra[i] = *reinterpret_cast<const uint4*>(xb + (long)((iz * p.Hi + iy) * p.Wi + ix) * p.ldx + c0);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      const int piece = tid + 256 * i;
      const int n = piece >> 2;
      rb[i] = make_uint4(0, 0, 0, 0);
      if (n < BN) rb[i] = *reinterpret_cast<const uint4*>(wb + ((long)wtap * p.N + n0 + n) * p.C + c0);
    }
    // advance (cc fastest, then x, y, z of the tap)
    if (++l_cc == cchunks) {
      l_cc = 0; ++l_t;
      if (++l_jx == l_nx) { l_jx = 0; if (++l_jy == l_ny) { l_jy = 0; ++l_jz; } }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) ldsA[buf * BM * 4 + swz((tid >> 2) + 64 * i, chunk)] = ra[i];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      const int piece = tid + 256 * i;
      const int n = piece >> 2;
      if (n < BN) ldsB[buf * BN * 4 + swz(n, chunk)] = rb[i];
    }
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_step(0);
  store_step(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  int cur = 0;
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) load_step(step + 1);
    const uint4* A = ldsA + cur * BM * 4;
    const uint4* B = ldsB + cur * BN * 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = A[swz(wm * 64 + i * 32 + fr, ks * 2 + fh)];
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = B[swz(wn * (NT * 32) + j * 32 + fr, ks * 2 + fh)];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_piece(af[i], bfr[j], acc[i][j], T());
    }
    if (step + 1 < nsteps) store_step(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: + bias, cast, store (row = voxel, lane column = channel) ----
  T* yb = static_cast<T*>(p.y) + (long)b * p.sby;
  const bool do_stats = p.stats != nullptr;          // (host: never together with ksplit > 1)
  float st_s[NT], st_q[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (NT * 32) + j * 32 + fr;
    const float bv = p.bias ? p.bias[b * p.bsb + n] : 0.f;
    st_s[j] = 0.f; st_q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int off = rowoff[row];
        if (off >= 0) {
          if (p.ksplit > 1) atomicAdd(p.part + (long)b * p.part_sb + (long)(off / p.ldy) * p.N + n, acc[i][j][e]);
          else {
            const T o = static_cast<T>(acc[i][j][e] + bv + (p.accum ? static_cast<float>(yb[off + n]) : 0.f));
            yb[off + n] = o;
            if (do_stats) { const float r = static_cast<float>(o); st_s[j] += r; st_q[j] = fmaf(r, r, st_q[j]); }
          }
        }
      }
    }
  }
  // fused statistics: a lane holds ONE channel per N-tile; its two half-waves (fh) and the block's M-waves fold through LDS
  if (do_stats) {
    __syncthreads();                                  // the K loop's LDS images are dead
    float* red = reinterpret_cast<float*>(smem);      // [4 waves][NT][32][2]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = st_s[j], c = st_q[j];
      a += __shfl_xor(a, 32, 64); c += __shfl_xor(c, 32, 64);
      if (fh == 0) { red[((wid * NT + j) * 32 + fr) * 2] = a; red[((wid * NT + j) * 32 + fr) * 2 + 1] = c; }
    }
    __syncthreads();
    for (int t = tid; t < BN; t += 256) {             // channel t of the block's BN: wave column wn = t / (NT * 32)
      const int twn = t / (NT * 32), tj = (t / 32) % NT, tfr = t & 31;
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4 / WAVES_N; ++w) {
        const int wv_ = w * WAVES_N + twn;
        a += (double)red[((wv_ * NT + tj) * 32 + tfr) * 2]; c += (double)red[((wv_ * NT + tj) * 32 + tfr) * 2 + 1];
      }
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, p.stats_inst ? b : 0, n0 + t, a, c);
    }
  }
}

// y[v][n] = T(part[v][n] + bias[n])  (the merge of the split-K partials)
template <typename T>
__global__ __launch_bounds__(256) void gather_finalize_k(const float* __restrict__ part, long part_sb, T* __restrict__ y, int ldy,
                                                         long sby, int N, long V, const float* __restrict__ bias, int bsb, int accum) {
  const int b = blockIdx.y;
  const long total = V * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long v = e / N; const int n = (int)(e - v * N);
    T* dst = y + (long)b * sby + v * ldy + n;
    *dst = static_cast<T>(part[(long)b * part_sb + e] + (bias ? bias[b * bsb + n] : 0.f) + (accum ? static_cast<float>(*dst) : 0.f));
  }
}

// 8 channels starting at `ptr`, of which `nvalid` exist; `vec` = 16-byte access is legal here: the base is 16-byte
// aligned and the voxel pitch is a multiple of 8 channels, so the 8-channel piece lies inside the voxel's row even when
// fewer than 8 of its channels belong to this tensor (a channel slice of a wider, padded buffer) -- those are masked off.
__device__ __forceinline__ uint4 load8(const bf16_t* ptr, int nvalid, bool vec) {
  if (vec) {
    uint4 v = *reinterpret_cast<const uint4*>(ptr);
    if (nvalid < 8) {
      auto m = [&](int j) -> unsigned { const int k = nvalid - 2 * j; return k >= 2 ? 0xffffffffu : (k == 1 ? 0xffffu : 0u); };
      v.x &= m(0); v.y &= m(1); v.z &= m(2); v.w &= m(3);
    }
    return v;
  }
  unsigned short e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = j < nvalid ? reinterpret_cast<const unsigned short*>(ptr)[j] : (unsigned short)0;
  return make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                    e[6] | ((unsigned)e[7] << 16));
}

// The same piece WITHOUT the mask when the vector access is legal: a prefetch must not consume its data (the mask would
// put an s_waitcnt vmcnt(0) behind every load and serialise a tile's 13 loads); apply mask8() when the piece is stored.
__device__ __forceinline__ uint4 load8_raw(const bf16_t* ptr, int nvalid, bool vec) {
  if (vec) return *reinterpret_cast<const uint4*>(ptr);
  return load8(ptr, nvalid, false);
}
__device__ __forceinline__ uint4 mask8(int nvalid) {
  auto m = [&](int j) -> unsigned { const int k = nvalid - 2 * j; return k >= 2 ? 0xffffffffu : (k == 1 ? 0xffffu : 0u); };
  return make_uint4(m(0), m(1), m(2), m(3));
}

// =====================================================================================
// Stride-1 3x3x3 convolution (forward, and data-gradient via `flip`) with the input HALO
// staged once per channel chunk in LDS: a block owns 8 M-tiles (2 z-planes x 4 row groups of
// 32 voxels) = 256 output voxels x 32 output channels; each of the 27 taps re-reads the same
// LDS halo at a shifted voxel address, so the 27-fold input reuse never leaves the CU.  The
// weights of the current chunk stream through LDS one kz-plane (9 taps) at a time.
// CK = channels per chunk (32, or 16 for the thin full-resolution layers so that an MFMA K step
// is exactly one tap).  Rows are XOR-swizzled so every ds_read_b128 fragment read is
// bank-conflict free; ~70 KB LDS => 2 blocks (8 waves) per CU overlap staging with MFMA.
// =====================================================================================
struct HaloP {
  const void* x; int ldx; long sbx; int D, H, W, C;
  void* y; int ldy; long sby; int N;
  const void* w; long wsb;
  const float* bias; int bsb;
  int flip, vecx, vecw;
  int ntx, nty, ntz;
};

template <int CPR> __device__ __forceinline__ int hswz(int row, int chunk) {
  // 16-byte slot index of (row, chunk) in a [rows][CPR pieces] image
  if (CPR == 4) return (row << 2) | (chunk ^ ((row >> 2) & 3));
  return (row << 1) | (chunk ^ ((row >> 3) & 1));
}

// CK = channels per LDS row (bf16: 32, or 16 for the thin layers; fp32: 16)
template <int CK, int LX, int VEC, typename T = bf16_t>   // VEC=1: every piece is a legal, fully valid 16-byte load
__global__ __launch_bounds__(256, 2) void conv_mfma_halo_k(HaloP p) {
  constexpr int EPB = elem<T>::EPB;
  static_assert(EPB == 8 || VEC == 1, "fp32 staging is vector-only");
  constexpr int TX = 1 << LX, RY = 32 / TX;      // M-tile = RY rows of TX voxels
  constexpr int TY = 4 * RY, TZ = 2;
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ;
  constexpr int CPR = CK / EPB;                   // 16-byte chunks per row
  constexpr int KS = CPR / 2;                     // fragment reads (16-byte pieces per lane half) per tap
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Hl = reinterpret_cast<uint4*>(smem);                  // halo  [HV][CPR]
  uint4* Wl = Hl + HV * CPR;                                   // weights [9][32][CPR]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, n0 = blockIdx.y * 32;
  int tix, tiy, tiz;
  if (!tile_coords(xcd_remap(blockIdx.x, gridDim.x), p.ntx, p.nty, p.ntz, tix, tiy, tiz)) return;
  const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
  const T* xb = static_cast<const T*>(p.x) + (long)b * p.sbx;
  const T* wb = static_cast<const T*>(p.w) + (long)b * p.wsb;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // halo voxel (tap 0,0,0 corner) of this lane's row in each of the wave's two M-tiles
  int hbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wid * 2 + i;                    // M-tile: z = j >> 2, row group = j & 3
    const int zz = j >> 2, yy = (j & 3) * RY + (fr >> LX), xx = fr & (TX - 1);
    hbase[i] = (zz * HY + yy) * HX + xx;
  }

  // ---- staging with register prefetch: all loads of a phase are issued before any is consumed ----
  constexpr int HP = HV * CPR, HIT = (HP + 255) / 256;          // halo pieces, per-thread iterations
  constexpr int WP = 9 * 32 * CPR, WIT = (WP + 255) / 256;      // weight pieces of one kz-plane
  uint4 hreg[HIT], wreg[WIT];
  auto ld_piece = [&](const T* src, int nvalid, int vecok) -> uint4 {
    if constexpr (VEC) return *reinterpret_cast<const uint4*>(src);
    else return load8(reinterpret_cast<const bf16_t*>(src), nvalid, vecok);
  };
  auto load_halo = [&](int c0) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int piece = tid + 256 * it;
      const int row = piece / CPR, ch = piece % CPR;
      const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const int cbeg = c0 + ch * EPB;
      hreg[it] = make_uint4(0, 0, 0, 0);
      if (piece < HP && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W &&
          cbeg < p.C)
        hreg[it] = ld_piece(xb + (long)((gz * p.H + gy) * p.W + gx) * p.ldx + cbeg, p.C - cbeg, p.vecx);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int piece = tid + 256 * it;
      if (piece < HP) Hl[hswz<CPR>(piece / CPR, piece % CPR)] = hreg[it];
    }
  };
  auto load_w = [&](int c0, int g) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int piece = tid + 256 * it;
      const int ch = piece % CPR, n = (piece / CPR) & 31, t9 = piece / (CPR * 32);
      const int gt = g * 9 + t9;
      const int wt = p.flip ? 26 - gt : gt;
      const int cbeg = c0 + ch * EPB;
      wreg[it] = make_uint4(0, 0, 0, 0);
      if (piece < WP && n0 + n < p.N && cbeg < p.C)
        wreg[it] = ld_piece(wb + ((long)wt * p.N + n0 + n) * p.C + cbeg, p.C - cbeg, p.vecw);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int piece = tid + 256 * it;
      if (piece < WP) Wl[hswz<CPR>((piece / (CPR * 32)) * 32 + ((piece / CPR) & 31), piece % CPR)] = wreg[it];
    }
  };

  const int nchunks = (p.C + CK - 1) / CK;
  load_halo(0);
  load_w(0, 0);
  for (int cc = 0; cc < nchunks; ++cc) {
    const int c0 = cc * CK;
    __syncthreads();                 // every wave is done reading the previous chunk's LDS images
    store_halo();
#pragma unroll 1
    for (int g = 0; g < 3; ++g) {
      if (g > 0) __syncthreads();    // previous kz-plane's weights are no longer being read
      store_w();
      __syncthreads();
      if (g < 2) load_w(c0, g + 1);
      else if (cc + 1 < nchunks) { load_w(c0 + CK, 0); load_halo(c0 + CK); }
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int t9 = dy * 3 + dx;
          const int toff = (g * HY + dy) * HX + dx;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const uint4 bv = Wl[hswz<CPR>(t9 * 32 + fr, ks * 2 + fh)];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const uint4 av = Hl[hswz<CPR>(hbase[i] + toff, ks * 2 + fh)];
              acc[i] = mma_piece(av, bv, acc[i], T());
            }
          }
        }
      }
    }
  }
  // ---- epilogue ----
  const int n = n0 + fr;
  if (n < p.N) {
    const float bv = p.bias ? p.bias[b * p.bsb + n] : 0.f;
    T* yb = static_cast<T*>(p.y) + (long)b * p.sby;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = wid * 2 + i;
      const int gz = z0 + (j >> 2);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int gy = y0 + (j & 3) * RY + (row >> LX), gx = x0 + (row & (TX - 1));
        if (gz < p.D && gy < p.H && gx < p.W)
          yb[(long)((gz * p.H + gy) * p.W + gx) * p.ldy + n] = static_cast<T>(acc[i][e] + bv);
      }
    }
  }
}

// =====================================================================================
// conv_mfma_halo2_k -- the production stride-1 3x3x3 kernel for W >= 32, C % 32 == 0.
// Same tiling as conv_mfma_halo_k (2 x 4 x 32 output voxels x 32 output channels per tile) but
//  * PERSISTENT: a block walks a contiguous run of tiles of one (sample, 32-channel slice), so the
//    per-thread staging descriptors are computed once and, when C == 32, ALL 27 taps' weights stay
//    resident in LDS for the whole kernel (no weight traffic, two barriers per tile);
//  * LDS images use an 80-byte row pitch instead of an XOR swizzle: 16 rows x 80 B land on 16
//    distinct 16-B slots of the 256-B bank row, so ds_read_b128 fragment reads stay conflict-free
//    while every tap/K-step address is `lane base + compile-time immediate` (no VALU per read);
//  * the MFMA is issued as (weights x voxels): a lane then holds 4 x 4 consecutive output
//    channels of ONE voxel, and the epilogue is four 8-byte stores per M-tile at immediate offsets;
//  * the next tile's halo (13 x 16 B per thread) is in flight in registers while this tile computes.
// PMC on the first halo kernel showed 16 VALU + 4 SALU instructions per MFMA and 54 % of wave
// cycles waiting; this structure brings the instruction mix under 2 VALU per MFMA.
// =====================================================================================
// ---- diagnostic build only (-DCOMA_STAMPS; profiles/stamps_halo2.py): per-phase s_memtime sums of conv_mfma_halo2_k.
// The stamps leave the kernel through g_stamps alone (no output value depends on them); the shipped library has none.
#ifdef COMA_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(var) const unsigned long long var = stamp_now()
#define STAMP_ADD(slot, a, b) st_acc[slot] += (b) - (a)
extern "C" int coma_debug_read_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

struct Halo2P {
  const void* x; int ldx; long sbx; int D, H, W, C;
  void* y; int ldy; long sby; int N;
  const void* w; long wsb;
  const float* bias; int bsb;
  int flip, vecx, vecw;
  int ntx, nty, ntz, ids_total, ids_per_block;
  unsigned xbytes, wbytes;   // bytes of one sample of x / of one weight set (buffer descriptors: out-of-range pieces read as zero)
  int st8;             // output rows allow aligned 8-byte (4-channel) stores
  int st16;            // ... and aligned 16-byte (8-channel) stores
  double2* stats;      // optional: per-block {sum, sumsq} of the stored outputs, [chunk][G][N]
  int stats_inst;      // B: groups = the B samples (InstanceNorm), 0: one group (BatchNorm)
  const void* wfrag;   // conv_mfma_duo_k, C >= 64: the weights re-laid in fragment order (duo_relayout_k), else null
  long wfrag_sb;       // ... elements between two samples' weight sets
};

// CK = channels per LDS row (bf16: 32, or 16 for the thin full-resolution layers: one MFMA K step per tap; fp32: 16);
// VEC = every piece is a legal, fully valid 16-byte load; OCC = blocks per CU; T = element type (see `elem`).
template <int RESIDENT, int CK, int VEC, int OCC, typename T = bf16_t>
__global__ __launch_bounds__(256, OCC) void conv_mfma_halo2_k(Halo2P p) {
  constexpr int EPB = elem<T>::EPB;
  constexpr bool F32 = EPB == 4;
  static_assert(!F32 || (VEC == 1 && OCC == 1 && RESIDENT == 2), "fp32: vector staging, one block per CU, per-chunk weights");
  constexpr bool THIN = RESIDENT == 1 && OCC == 2;      // the few-channel variant (one chunk per tile, two blocks per CU)
#ifdef COMA_HALO2_NO_YREUSE
  constexpr bool YREUSE = false;
#else
  constexpr bool YREUSE = !(EPB == 4) && CK == 32 && OCC == 1 && RESIDENT != 0;   // the thick bf16 variants (see the tap loop)
#endif
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ;
  constexpr int CPR = CK / EPB, KS = CPR / 2;        // 16-byte pieces per row, fragment reads per tap and operand
  constexpr int P = CPR == 4 ? 80 : 48;               // LDS row pitch (bytes): 16 rows land on 16 distinct 16-B slots
  constexpr int HP = HV * CPR, HIT = (HP + 255) / 256;  // halo pieces, per-thread iterations
  constexpr int WT = RESIDENT ? 27 : 9;               // taps held in LDS at once
  constexpr int WP = WT * 32 * CPR, WIT = (WP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;                                    // [HV][80]
  char* Wl = smem + HV * P;                           // [WT*32][80]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, n0 = blockIdx.y * 32;
  const int fr = lane & 31, fh = lane >> 5;
  const T* xb = static_cast<const T*>(p.x) + (long)b * p.sbx;
  const T* wb = static_cast<const T*>(p.w) + (long)b * p.wsb;
  T* yb = static_cast<T*>(p.y) + (long)b * p.sby;
  const int nchunks = (p.C + CK - 1) / CK;

  // ---- staging descriptors (tile independent) ----
  int h_roff[HIT], h_lds[HIT], h_z[HIT], h_y[HIT], h_x[HIT], h_ch[HIT];
#pragma unroll
  for (int it = 0; it < HIT; ++it) {
    const int piece = tid + 256 * it;
    const int row = piece / CPR, ch = piece % CPR;
    const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
    h_z[it] = piece < HP ? hz : (1 << 20); h_y[it] = hy; h_x[it] = hx;
    h_ch[it] = ch * EPB;
    h_roff[it] = ((hz * p.H + hy) * p.W + hx) * p.ldx + ch * EPB;
    h_lds[it] = row * P + ch * 16;
  }
  int w_goff[WIT], w_lds[WIT], w_ch[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int piece = tid + 256 * it;
    const int ch = piece % CPR, n = (piece / CPR) & 31, t = piece / (CPR * 32);     // t: tap within the LDS image
    w_ch[it] = ch * EPB;
    w_goff[it] = (piece < WP && n0 + n < p.N) ? (n0 + n) * p.C + ch * EPB : -1;   // + wtap*N*C + c0 at load time
    w_lds[it] = (t * 32 + n) * P + ch * 16;
  }
  // VEC: staging goes through buffer loads -- one descriptor per operand, 32-bit byte offsets, the hardware range check
  // zero-fills an out-of-range piece.  A piece outside the volume gets an out-of-range offset by ONE select: no branch,
  // no exec masking, no zero-initialised destination, no 64-bit address arithmetic.  (Stamped build of the 64 -> 32
  // forward at 128^3: issuing a chunk's 27 conditional 64-bit loads took 2400 cycles, 22 % of the kernel, next to 3860 for
  // its 108 MFMAs.)  The weights' offsets are tile independent: tap, row and flip are folded in once per kernel.
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, VEC ? p.xbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wb), 0, VEC ? p.wbytes : 0, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;
  unsigned w_boff[WIT];
  if constexpr (VEC) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int piece = tid + 256 * it;
      const int t = piece / (CPR * 32);
      const int wt = p.flip ? 26 - t : t;
      w_boff[it] = (w_goff[it] >= 0 && t < 27) ? (unsigned)((wt * p.N * p.C + w_goff[it]) * (int)sizeof(T)) : OOB;
    }
  }
  // fragment read bases
  int a_base[2];   // voxel operand (MFMA B): this lane's voxel row in each of the wave's two M-tiles
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wid * 2 + i;
    a_base[i] = (((j >> 2) * HY + (j & 3)) * HX + fr) * P + fh * 16;
  }
  const int w_base = fr * P + fh * 16;

  uint4 hreg[HIT], wreg[WIT];
  auto load_halo = [&](int z0, int y0, int x0, int c0) {
    const int zb = z0 - 1, yb0 = y0 - 1, xb0 = x0 - 1;
    const long org = ((long)(zb * p.H + yb0) * p.W + xb0) * p.ldx + c0;
    if constexpr (VEC) {
      const unsigned org_b = (unsigned)(org * (long)sizeof(T));   // (mod 2^32; negative at the volume's faces: a valid piece's sum is its true offset)
#pragma unroll
      for (int it = 0; it < HIT; ++it) {
        const bool ok = (unsigned)(zb + h_z[it]) < (unsigned)p.D && (unsigned)(yb0 + h_y[it]) < (unsigned)p.H &&
                        (unsigned)(xb0 + h_x[it]) < (unsigned)p.W;
        const unsigned voff = ok ? org_b + (unsigned)h_roff[it] * (unsigned)sizeof(T) : OOB;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, 0, 0);
        hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const bool ok = (unsigned)(zb + h_z[it]) < (unsigned)p.D && (unsigned)(yb0 + h_y[it]) < (unsigned)p.H &&
                      (unsigned)(xb0 + h_x[it]) < (unsigned)p.W;
      hreg[it] = make_uint4(0, 0, 0, 0);
      if constexpr (VEC) { if (ok) hreg[it] = *reinterpret_cast<const uint4*>(xb + org + h_roff[it]); }
      else if (ok && c0 + h_ch[it] < p.C) hreg[it] = load8_raw(reinterpret_cast<const bf16_t*>(xb + org + h_roff[it]), p.C - c0 - h_ch[it], p.vecx);   // mask: at the LDS store
    }
  };
  auto store_halo = [&](int c0) {
#pragma unroll
    for (int it = 0; it < HIT; ++it)
      if (tid + 256 * it < HP) {
        uint4 v = hreg[it];
        if (!VEC) { const uint4 m = mask8(p.C - c0 - h_ch[it]); v.x &= m.x; v.y &= m.y; v.z &= m.z; v.w &= m.w; }
        *reinterpret_cast<uint4*>(Hl + h_lds[it]) = v;
      }
  };
  auto load_w = [&](int c0, int g) {     // RESIDENT: all 27 taps (g ignored); else kz-plane g
    if constexpr (VEC && RESIDENT != 0) {
      const int c0_b = c0 * (int)sizeof(T);               // (chunk offset: a scalar)
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_boff[it], c0_b, 0);
        wreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int piece = tid + 256 * it;
      const int t = piece / (CPR * 32) + (RESIDENT ? 0 : g * 9);
      const int wt = p.flip ? 26 - t : t;
      wreg[it] = make_uint4(0, 0, 0, 0);
      if constexpr (VEC) { if (w_goff[it] >= 0) wreg[it] = *reinterpret_cast<const uint4*>(wb + (long)wt * p.N * p.C + c0 + w_goff[it]); }
      else if (w_goff[it] >= 0 && c0 + w_ch[it] < p.C)
        wreg[it] = load8(reinterpret_cast<const bf16_t*>(wb + (long)wt * p.N * p.C + c0 + w_goff[it]), p.C - c0 - w_ch[it], p.vecw);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      if (tid + 256 * it < WP) *reinterpret_cast<uint4*>(Wl + w_lds[it]) = wreg[it];
  };

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;

  // first valid tile
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id >= id_end) {        // nothing to do for this block (padding ids): still publish a zero partial
    return;      // (nothing to add to the statistics record)
  }
  if (RESIDENT == 1) { load_w(0, 0); store_w(); }
  load_halo(tiz * TZ, tiy * TY, tix * TX, 0);
  if (RESIDENT != 1) load_w(0, 0);

#ifdef COMA_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_begin = stamp_now();
#endif
  const bool has_bias = p.bias != nullptr, do_stats = p.stats != nullptr;
  float bv[4][4];
  float st_s[4][4], st_q[4][4];     // fused norm statistics of this lane's 16 channels (stored values)
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * g4 + 4 * fh + q;
      bv[g4][q] = (p.bias && n < p.N) ? p.bias[b * p.bsb + n] : 0.f;
      st_s[g4][q] = 0.f; st_q[g4][q] = 0.f;
    }

  while (id < id_end) {
    const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
    // next valid tile (for the prefetch)
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;

    f32x16_t acc[2];
    if (!THIN) {   // (thin variant: one chunk, the first MFMA of the tile takes a literal-zero C operand)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    }

    for (int cc = 0; cc < nchunks; ++cc) {
      STAMP(t0);
      __syncthreads();                       // all waves finished reading the previous halo / weights
      STAMP(t1);
#ifdef COMA_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (diagnostic: how long the prefetched pieces still need)
#endif
      STAMP(t1b);
      store_halo(cc * CK);
      if (RESIDENT) {
        if (RESIDENT == 2) store_w();        // this chunk's 27 taps (fetched while the previous chunk / tile was computed)
        STAMP(t2);
        __syncthreads();
        STAMP(t3);
        // What to prefetch while this chunk computes: the next chunk of this tile, or chunk 0 of the next tile.
        // VEC: the loads are NOT issued here in one burst -- a CU's vector-memory front end takes a chunk's 110 KB of
        // requests at ~64 B/clk, i.e. ~2000 cycles during which an in-order wave issues no MFMA (stamped build: 22-25 % of
        // the kernel) -- but one per tap inside the MFMA loop below, through descriptors whose range is zero when there
        // is nothing to prefetch (every piece then reads as zero: no branch in the loop).
        const bool same_tile = RESIDENT == 2 && cc + 1 < nchunks;
        const bool pref = same_tile || has_next;
        const int pz = same_tile ? z0 : ntiz * TZ, py = same_tile ? y0 : ntiy * TY, px = same_tile ? x0 : ntix * TX;
        const int pc0 = same_tile ? cc * CK + CK : 0;
        if constexpr (!VEC) {
          if (RESIDENT == 2) { if (pref) { load_w(pc0, 0); load_halo(pz, py, px, pc0); } }
          else if (has_next) load_halo(pz, py, px, 0);
        }
        const __amdgpu_buffer_rsrc_t rs_xp = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, (VEC && pref) ? p.xbytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wb), 0, (VEC && pref && RESIDENT == 2) ? p.wbytes : 0, 0x00020000);
        const unsigned porg_b = (unsigned)((((long)((pz - 1) * p.H + (py - 1)) * p.W + (px - 1)) * p.ldx + pc0) * (long)sizeof(T));
        const int pc0_b = pc0 * (int)sizeof(T);
        auto pref_piece = [&](int t) __attribute__((always_inline)) {      // t: the tap whose MFMAs cover this piece's issue
          if constexpr (VEC) {
            if (RESIDENT == 2 && t < WIT) {
              const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_wp, w_boff[t], pc0_b, 0);
              wreg[t] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            const int it = RESIDENT == 2 ? t - WIT : t;
            if (it >= 0 && it < HIT) {
              const bool ok = (unsigned)(pz - 1 + h_z[it]) < (unsigned)p.D && (unsigned)(py - 1 + h_y[it]) < (unsigned)p.H &&
                              (unsigned)(px - 1 + h_x[it]) < (unsigned)p.W;
              const unsigned voff = ok ? porg_b + (unsigned)h_roff[it] * (unsigned)sizeof(T) : OOB;
              const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_xp, voff, 0, 0);
              hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
            }
          }
        };
        static_assert(!VEC || (RESIDENT == 2 ? WIT + HIT : HIT) <= 27, "one prefetch piece per tap");
        STAMP(t4);
        STAMP_ADD(0, t0, t1); STAMP_ADD(1, t1, t1b); STAMP_ADD(2, t1b, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4);
        // LDS fragments are read one tap ahead into a second register set (hipcc otherwise issues each ds_read right
        // before the MFMA that consumes it and waits lgkmcnt(0): the full LDS latency on every MFMA at 1-2 waves/SIMD)
        if constexpr (YREUSE) {
          // The wave's two M-tiles are y-neighbours (rows y, y + 1 of one z): for a fixed (kz, kx) their three ky taps read
          // halo rows y .. y + 3, four distinct fragments per K step instead of six -- 14 fragment reads per 12 MFMAs
          // instead of 18 (the LDS array is the unit all four waves share: 1.5 ds_read_b128 per MFMA keeps it 75 % busy).
          uint4 wg[2][3][KS], xr[2][4][KS];
          auto rdg = [&](int g, int bf) {
            const int kz = g / 3, kx = g % 3;
            const int toff = (kz * HY * HX + kx) * P;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
              for (int r = 0; r < 4; ++r) xr[bf][r][ks] = *reinterpret_cast<const uint4*>(Hl + a_base[0] + toff + r * HX * P + ks * 32);
#pragma unroll
              for (int ky = 0; ky < 3; ++ky)
                wg[bf][ky][ks] = *reinterpret_cast<const uint4*>(Wl + w_base + (kz * 9 + ky * 3 + kx) * 32 * P + ks * 32);
            }
          };
          rdg(0, 0);
#pragma unroll
          for (int g = 0; g < 9; ++g) {
            if (g + 1 < 9) rdg(g + 1, (g + 1) & 1);
            pref_piece(3 * g); pref_piece(3 * g + 1); pref_piece(3 * g + 2);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = mma_piece(wg[g & 1][ky][ks], xr[g & 1][i + ky][ks], acc[i], T());
#define COMA_SGB(mfma, ds, valu, vmem)                                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, mfma, 0);                            \
            if (ds) __builtin_amdgcn_sched_group_barrier(0x100, ds, 0);                      \
            if (valu) __builtin_amdgcn_sched_group_barrier(0x006, valu, 0);                  \
            if (vmem) __builtin_amdgcn_sched_group_barrier(0x020, vmem, 0);
            COMA_SGB(1, 2, 4, 0) COMA_SGB(1, 2, 4, 0) COMA_SGB(1, 2, 4, 1) COMA_SGB(1, 1, 3, 0)
            COMA_SGB(1, 1, 3, 0) COMA_SGB(1, 1, 3, 1) COMA_SGB(1, 1, 3, 0) COMA_SGB(1, 1, 3, 0)
            COMA_SGB(1, 1, 3, 1) COMA_SGB(1, 1, 3, 0) COMA_SGB(1, 1, 0, 0) COMA_SGB(1, 0, 0, 0)
#undef COMA_SGB
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
        uint4 wv[2][KS], xv[2][2][KS];
        auto rd = [&](int t, int bf) {
          const int toff = (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * P;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            wv[bf][ks] = *reinterpret_cast<const uint4*>(Wl + w_base + t * 32 * P + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) xv[bf][i][ks] = *reinterpret_cast<const uint4*>(Hl + a_base[i] + toff + ks * 32);
          }
        };
        rd(0, 0);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          if (t + 1 < 27) rd(t + 1, (t + 1) & 1);
          pref_piece(t);
          if constexpr (!(!F32 && KS == 2 && OCC == 1)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[i] = mma_piece(wv[t & 1][ks], xv[t & 1][i][ks], (THIN && t == 0 && ks == 0) ? zero : acc[i], T());
            }
          // one tap = 4 MFMAs (bf16, KS = 2): the next tap's 6 fragment reads and this tap's prefetch piece (address
          // arithmetic + one buffer load) go BETWEEN them, not in front: an in-order wave otherwise issues ~100 cycles of
          // LDS / VALU work while the matrix pipe has only the tail of the previous tap to run
          if constexpr (!F32 && KS == 2 && OCC == 1) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        }
        STAMP(t5);
        STAMP_ADD(5, t4, t5);
      } else {
#pragma unroll 1
        for (int g = 0; g < 3; ++g) {
          if (g > 0) __syncthreads();        // previous kz-plane's weights are no longer read
          store_w();
          __syncthreads();
          if (g < 2) load_w(cc * CK, g + 1);
          else if (cc + 1 < nchunks) { load_w(cc * CK + CK, 0); load_halo(z0, y0, x0, cc * CK + CK); }
          else if (has_next) { load_w(0, 0); load_halo(ntiz * TZ, ntiy * TY, ntix * TX, 0); }
          const int gofs = g * HY * HX * P;
          uint4 wv[2][KS], xv[2][2][KS];     // fragments one tap ahead (see the resident branch)
          auto rd = [&](int t, int bf) {
            const int toff = ((t / 3) * HX + t % 3) * P;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              wv[bf][ks] = *reinterpret_cast<const uint4*>(Wl + w_base + t * 32 * P + ks * 32);
#pragma unroll
              for (int i = 0; i < 2; ++i) xv[bf][i][ks] = *reinterpret_cast<const uint4*>(Hl + a_base[i] + gofs + toff + ks * 32);
            }
          };
          rd(0, 0);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) rd(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
              for (int i = 0; i < 2; ++i) acc[i] = mma_piece(wv[t & 1][ks], xv[t & 1][i][ks], acc[i], T());
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // ---- epilogue: lane = one voxel, 4 groups of 4 consecutive channels ----
    STAMP(t6);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = wid * 2 + i;
      const int gz = z0 + (j >> 2), gy = y0 + (j & 3), gx = x0 + fr;
      if constexpr (F32) {
        // fp32: the lane's 4 consecutive channels of a group are one 16-byte store as they stand
        const bool valid = gz < p.D && gy < p.H && gx < p.W;
        float* dst = reinterpret_cast<float*>(yb) + ((long)(gz * p.H + gy) * p.W + gx) * p.ldy + n0 + 4 * fh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (n0 + 8 * g4 >= p.N) continue;            // (wave-uniform)
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = acc[i][g4 * 4 + q] + bv[g4][q];
            if (do_stats) { const float r = valid ? o[q] : 0.f; st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]); }
          }
          if (valid) {
            if (p.st16 && n0 + 8 * g4 + 4 * fh + 3 < p.N) *reinterpret_cast<float4*>(dst + 8 * g4) = make_float4(o[0], o[1], o[2], o[3]);
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (n0 + 8 * g4 + 4 * fh + q < p.N) dst[8 * g4 + q] = o[q];
            }
          }
        }
      } else
      if (!THIN && p.st16 && n0 + 32 <= p.N) {
        // Full 32-channel tiles: the two half-waves exchange 4-channel groups (v_permlane32_swap), so that every lane
        // owns 8 CONSECUTIVE channels of its voxel and writes them with one 16-byte store (lanes fh = 0: channels
        // 16 gp + 0..7, fh = 1: 16 gp + 8..15) instead of four 8-byte stores that each cover a quarter of a 64-byte row.
        const bool valid = gz < p.D && gy < p.H && gx < p.W;
        unsigned pk[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = static_cast<bf16_t>(acc[i][g4 * 4 + q] + bv[g4][q]);
            const float r = valid ? static_cast<float>(o[q]) : 0.f;
            st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]);
          }
          const uint2 u = *reinterpret_cast<const uint2*>(o);
          pk[g4][0] = u.x; pk[g4][1] = u.y;
        }
        bf16_t* vox = yb + ((long)(gz * p.H + gy) * p.W + gx) * p.ldy + n0 + 8 * fh;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
          if (valid) *reinterpret_cast<uint4*>(vox + 16 * gp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      } else if (THIN && p.st16 && n0 + 16 <= p.N && ((p.N - n0) & 15) == 0) {
        // thin variant, 16 (or 32) output channels: the same exchange, one 16-byte store per lane and 16 channels
        const bool valid = gz < p.D && gy < p.H && gx < p.W;
        bf16_t* vox = yb + ((long)(gz * p.H + gy) * p.W + gx) * p.ldy + n0 + 8 * fh;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          if (n0 + 16 * gp >= p.N) continue;            // (wave-uniform)
          unsigned pk[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int g4 = 2 * gp + h;
            bf16_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float a = acc[i][g4 * 4 + q];
              if (has_bias) a += bv[g4][q];
              o[q] = static_cast<bf16_t>(a);
              if (do_stats) {
                const float r = valid ? static_cast<float>(o[q]) : 0.f;
                st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]);
              }
            }
            const uint2 u = *reinterpret_cast<const uint2*>(o);
            pk[h][0] = u.x; pk[h][1] = u.y;
          }
          const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
#ifdef COMA_ABLATE_STORE      // (diagnostic build only: the epilogue without its global stores; outputs are wrong)
          asm volatile("" :: "v"(r0[0]), "v"(r1[0]), "v"(r0[1]), "v"(r1[1]), "v"(vox));
#else
          if (valid) *reinterpret_cast<uint4*>(vox + 16 * gp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
#endif
        }
      } else if (gz < p.D && gy < p.H && gx < p.W) {
        bf16_t* dst = yb + ((long)(gz * p.H + gy) * p.W + gx) * p.ldy + n0 + 4 * fh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // wave-uniform skips: at one or two waves per SIMD this VALU work is not hidden behind anything, so channel
          // groups past N (zero padding of the thin layers), the bias add of a data-gradient call and the statistics
          // of a call that did not ask for them are not computed at all
          // (thin variant only: on the one-block-per-CU kernel the same branches measured 328-340 -> 402 us on the
          // 32 -> 32 forward, boxes differing: roughly +15-20 %)
          if (THIN && n0 + 8 * g4 >= p.N) continue;
          bf16_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = acc[i][g4 * 4 + q];
            if (!THIN || has_bias) a += bv[g4][q];
            o[q] = static_cast<bf16_t>(a);
            if (!THIN || do_stats) {
              const float r = static_cast<float>(o[q]);
              st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]);
            }
          }
          if (p.st8 && n0 + 8 * g4 + 4 * fh + 3 < p.N) *reinterpret_cast<uint2*>(dst + 8 * g4) = *reinterpret_cast<uint2*>(o);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n0 + 8 * g4 + 4 * fh + q < p.N) dst[8 * g4 + q] = o[q];
          }
        }
      }
    }
    STAMP(t7);
    STAMP_ADD(6, t6, t7);
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
#ifdef COMA_STAMPS
  {
    const unsigned long long st_end = stamp_now();
    if (lane == 0) {
      for (int k = 0; k < 7; ++k) atomicAdd(&g_stamps[k], st_acc[k]);
      atomicAdd(&g_stamps[7], st_end - st_begin);
      atomicAdd(&g_stamps[8], 1ull);
    }
  }
#endif
  // ---- fused statistics: lanes -> wave (butterfly over the 32 voxel lanes) -> block (LDS) -> partial[chunk] ----
  if (p.stats) {
    __syncthreads();                                  // LDS images are dead; reuse the front as a [4 waves][32 ch][2] table
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = st_s[g4][q], c = st_q[g4][q];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (fr == 0) { red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2] = a; red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2 + 1] = c; }
      }
    __syncthreads();
    if (tid < 32 && n0 + tid < p.N) {
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 32 + tid) * 2]; c += (double)red[(w * 32 + tid) * 2 + 1]; }
      const int g = p.stats_inst ? b : 0;
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, n0 + tid, a, c);
    }
  }
}

// =====================================================================================
// conv_mfma_duo_k -- the thick stride-1 3x3x3 kernel as TWO 4-wave groups per CU that alternate roles every phase.
// conv_mfma_halo2_k runs one 4-wave block per CU, and its phases do not overlap: stamped, the MFMA loop is 55 % of the
// kernel, LDS staging stores 18 %, the epilogue 13 %, barriers and waits 14 % -- the matrix pipe idles 45 % of the time.
// Here a workgroup has 8 waves = 2 groups (each SIMD holds one wave of either group).  In a phase one group runs the 54
// MFMAs of a (tile, 16-channel chunk) step out of ITS halo image while the other group does everything else for its own
// next step: writes the halo chunk it fetched two phases ago into its image, writes its half of the next chunk's weights,
// finishes a completed tile (bias, rounding, statistics, 16-byte stores) and issues the loads of the step after next.  One
// workgroup barrier per phase; the loads stay in flight across barriers (raw s_barrier behind an LDS-only wait).
//   LDS: 2 halo images of 816 rows x 48 B (16 channels + pad: 16 consecutive rows land on 16 distinct 16-byte slots)
//        + 2 weight images of 27 taps x 64 lanes x 16 B in FRAGMENT order (a wave's read is 1 KB contiguous) = 130.5 KB.
//   The two groups walk alternate tiles of the block's run and the same chunk sequence, so a weight image serves both
//   (phases 2s and 2s + 1) while the other one is refilled, half by either group.  C == 32: the two images hold the
//   layer's two chunks for the whole kernel.
//   A wave's two M-tiles are y-neighbours: for a fixed (kz, kx) their three ky taps read four distinct halo rows, so a
//   (kz, kx) group is 4 + 3 fragment reads for 6 MFMAs.
// =====================================================================================
// wk[b][tap][N][C] -> [b][n tile][16-channel chunk][tap][lane = (n & 31) + 32 * ((c >> 3) & 1)][8 channels]: the order in which
// conv_mfma_duo_k holds a weight image in LDS, so that a group's half image is ONE contiguous 13.8 KB run (from the plain
// layout a 16-channel piece is 32 bytes of a 2 C-byte row: at C = 64 the staging group pulled 4x the useful bytes through L1)
__global__ __launch_bounds__(256) void duo_relayout_k(const uint4* __restrict__ src, uint4* __restrict__ dst, int N, int C8, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // piece (b, tap, n, c8)
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  long t = i / C8;
  const int n = (int)(t % N); t /= N;
  const int tap = (int)(t % 27);
  const long b = t / 27;
  const int nt = N >> 5, nch = C8 >> 1;
  const long o = ((((b * nt + (n >> 5)) * nch + (c8 >> 1)) * 27 + tap) * 64) + (n & 31) + 32 * (c8 & 1);
  dst[o] = src[i];
}

__device__ __forceinline__ void duo_barrier() {
  // LDS traffic of this wave has landed; buffer loads issued for later phases stay in flight (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LX = 5: tiles of 2 x 4 x 32 voxels (W >= 32); LX = 4: 2 x 8 x 16 (the 16^3 level: an M-tile is two x-rows of 16 voxels)
template <int STATS, int LX = 5>
__global__ __launch_bounds__(512, 1) void conv_mfma_duo_k(Halo2P p) {
  constexpr int TX = 1 << LX, RY = 32 / TX, TY = 4 * RY, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ;
  static_assert(HV <= 816, "the halo image is sized for the 2 x 4 x 32 tile");
  constexpr int P = 48, HB = 816 * P, WB = 27 * 64 * 16;
  constexpr int HP = HV * 2, HIT = (HP + 255) / 256;        // halo pieces of a 16-channel chunk; per thread of a group
  constexpr int WH = 27 * 32, WIT = (WH + 255) / 256;       // weight pieces per half image; per thread of a group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the wave index through readfirstlane: everything derived from it (group, tile walk, descriptors, role branches) is
  // then provably wave-uniform -- otherwise the compiler wraps every buffer load in a waterfall loop
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2, wq = wid & 3, gt = tid & 255;
  char* const Hl = smem + grp * HB;
  char* const Wl = smem + 2 * HB;
  const int b = blockIdx.z, n0 = blockIdx.y * 32;
  const int fr = lane & 31, fh = lane >> 5;
  const bf16_t* xb = static_cast<const bf16_t*>(p.x) + (long)b * p.sbx;
  const bf16_t* wb = static_cast<const bf16_t*>(p.w) + (long)b * p.wsb;
  bf16_t* yb = static_cast<bf16_t*>(p.y) + (long)b * p.sby;
  const int nch = p.C >> 4;
  const bool w_static = nch <= 2;

  // ---- staging descriptors ----
  // Halo piece `it` of this thread is row (gt >> 1) + 128 it, half gt & 1.  Its halo coordinates and global offset are
  // rebuilt from the FIRST piece's coordinates at every fetch (row + 128 = 3 x-lines + 26: a few adds and compares per
  // piece on the staging wave) instead of living in 14 registers: at two waves per SIMD registers are the scarce thing.
  const int h_row0 = gt >> 1, h_half = gt & 1;
  unsigned h_zyx0 = ((unsigned)(h_row0 / (HX * HY)) << 20) | ((unsigned)((h_row0 / HX) % HY) << 10) | (unsigned)(h_row0 % HX);
  // in-volume test of a halo piece as two packed subtractions (fields of 9 bits under a guard bit each): halo coordinate h
  // of a tile at origin o is inside iff lo <= h <= hi with lo = (o == 0), hi = min(dim - o, 510) per axis
  constexpr unsigned GUARD = (1u << 29) | (1u << 19) | (1u << 9);
  const int h_lds0 = (gt >> 1) * P + (gt & 1) * 16;          // + it * 128 * P
  constexpr unsigned OOB = 0x7fff0000u;
  const int w_half = grp == 1 ? 0 : WH;                       // group 1 fills pieces [0, WH), group 0 [WH, 2 WH)
  // weight piece q = w_half + gt + 256 it = (tap, lane): tap = q >> 6 advances by 4 per `it`, the lane part is a per-thread
  // constant -- its global offset is rebuilt at every fetch (a handful of VALU on the staging wave; registers are the
  // scarce thing at two waves per SIMD)
  const int w_q0 = w_half + gt;
  const bool wf = p.wfrag != nullptr;                         // fragment-ordered weights: piece q of a chunk image sits at q * 16
  const unsigned w_lane_b = wf ? (unsigned)((w_q0 & 63) * 16)
                               : (unsigned)((((n0 + (w_q0 & 31)) * p.C) + ((w_q0 >> 5) & 1) * 8) * 2);
  const int w_lds0 = w_q0 * 16;                               // + it * 4096
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, p.xbytes, 0x00020000);
  const bf16_t* wfb = wf ? static_cast<const bf16_t*>(p.wfrag) + (long)b * p.wfrag_sb + (long)blockIdx.y * (nch * 27 * 512) : wb;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wfb), 0, wf ? (unsigned)(nch * 27 * 1024) : p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, 0, 0x00020000);   // reads as zero

  // fragment bases: this wave's M-tiles are rows y0, y0 + 1 of plane zq
  const int zq = wq >> 1, y0w = (wq & 1) * 2;
  // (LX = 4: M-tile j = 2 wq + i is rows 2 (j & 3), 2 (j & 3) + 1 of plane zq; the lane's voxel is (fr >> 4, fr & 15) in it)
  const int a_base = LX == 5 ? ((zq * HY + y0w) * HX + fr) * P + fh * 16
                             : ((zq * HY + y0w * RY + (fr >> LX)) * HX + (fr & (TX - 1))) * P + fh * 16;
  const int w_frag = lane * 16;

  // ---- this group's tile walk ----
  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  struct Cur { int valid, cc, tix, tiy, tiz, id; };
  auto first_tile = [&](Cur& c, int from) {        // first valid id >= from with the group's parity relative to id_begin
    c.cc = 0; c.valid = 0; c.tix = c.tiy = c.tiz = 0;
    int id = from;
    while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, c.tix, c.tiy, c.tiz)) id += 2;
    c.id = id;
    c.valid = id < id_end;
  };
  auto advance = [&](Cur& c) {
    if (!c.valid) return;
    if (c.cc + 1 < nch) { ++c.cc; return; }
    first_tile(c, c.id + 2);
  };
  // tiles of the two groups: count them (the phase count must be the same for every wave of the block)
  int nt0 = 0, nt1 = 0;
  {
    int a, bq, c;
    for (int id = id_begin; id < id_end; ++id)
      if (tile_coords(id, p.ntx, p.nty, p.ntz, a, bq, c)) { if ((id - id_begin) & 1) ++nt1; else ++nt0; }
  }
  const int nsteps = (nt0 > nt1 ? nt0 : nt1) * nch;
  if (nsteps == 0) return;
  const int nphases = 2 * nsteps + 4;                // (a multiple of 4: nch is even) + 1 round: the last tiles are finished in a staging phase

  // (native 4-dword vectors: as HIP uint4 structs the loop-carried pieces were split into scalars, the loads landed in
  // temporaries and the copies into the carried registers put an s_waitcnt vmcnt(0) in front of every barrier)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  // Halo pieces are fetched for TWO consecutive chunk steps at once (hreg[0]: the even chunk, hreg[1]: the odd one -- the
  // two 32-byte halves of a 64-byte run of every voxel row) and written to LDS one step apart: fetched one step at a time,
  // every 128-byte line went through L1 once per 16-channel chunk (4x the useful bytes at C = 64; staging alone 506 us
  // for 64 -> 32 at 128^3 next to 325 us of matrix phases).
  u32x4_t hreg[2][HIT], wreg[WIT];
  auto load_halo = [&](const Cur& c) {               // c: the even chunk step of a pair
    const int z0 = c.tiz * TZ, y0 = c.tiy * TY, x0 = c.tix * TX;
    const int zb = z0 - 1, yb0 = y0 - 1, xb0 = x0 - 1;
    const unsigned org_b = (unsigned)((((long)(zb * p.H + yb0) * p.W + xb0) * p.ldx + c.cc * 16) * 2);
    const __amdgpu_buffer_rsrc_t rs = c.valid ? rs_x : rs_0;
    const unsigned lo = ((unsigned)(z0 == 0) << 20) | ((unsigned)(y0 == 0) << 10) | (unsigned)(x0 == 0);
    const int hz_ = p.D - z0 < 510 ? p.D - z0 : 510, hy_ = p.H - y0 < 510 ? p.H - y0 : 510, hx_ = p.W - x0 < 510 ? p.W - x0 : 510;
    const unsigned hi = (((unsigned)hz_ << 20) | ((unsigned)hy_ << 10) | (unsigned)hx_) | GUARD;
    asm volatile("" : "+v"(h_zyx0));                 // (keeps the per-piece values out of loop-invariant registers)
    int hz = (int)(h_zyx0 >> 20), hy = (int)((h_zyx0 >> 10) & 1023), hx = (int)(h_zyx0 & 1023);
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const unsigned zyx = (gt + 256 * it < HP) ? ((unsigned)hz << 20) | ((unsigned)hy << 10) | (unsigned)hx : 0x1ff7fdffu;   // (invalid: above any limit)
      const bool ok = ((((zyx | GUARD) - lo) & (hi - zyx)) & GUARD) == GUARD;
      const unsigned roff = (unsigned)((((hz * p.H + hy) * p.W + hx) * p.ldx + h_half * 8) * 2);
      const unsigned voff = ok ? org_b + roff : OOB;
      hx += 128 % HX; hy += 128 / HX;                  // the next piece: 128 rows = 3 x-lines + 26 voxels on (LX = 4: 7 + 2)
      if (hx >= HX) { hx -= HX; ++hy; }
      if (hy >= HY) { hy -= HY; ++hz; }
      hreg[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
      hreg[1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 32u, 0, 0);  // (OOB + 32 is still out of range)
    }
  };
  auto store_halo = [&](auto odd_c) {
    constexpr int odd = decltype(odd_c)::value;
#pragma unroll
    for (int it = 0; it < HIT; ++it)
      if (gt + 256 * it < HP) *reinterpret_cast<u32x4_t*>(Hl + h_lds0 + it * 128 * P) = hreg[odd][it];
  };
  typedef std::integral_constant<int, 0> even_t;
  typedef std::integral_constant<int, 1> odd_t;
  auto load_w = [&](int ws, bool on) {               // this group's half of weight step ws (chunk ws % nch)
    const int c0_b = wf ? (ws % nch) * (27 * 1024) : (ws % nch) * 32;
    const __amdgpu_buffer_rsrc_t rs = on ? rs_w : rs_0;
    const unsigned tap_b = wf ? 1024u : (unsigned)(p.N * p.C * 2);
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int tap = (w_q0 >> 6) + 4 * it;
      const int wt = p.flip ? 26 - tap : tap;
      const unsigned voff = (gt + 256 * it < WH) ? (unsigned)wt * tap_b + w_lane_b : OOB;
      wreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, c0_b, 0);
    }
  };
  auto store_w = [&](int ws) {
    char* dst = Wl + (ws & 1) * WB + w_lds0;
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      if (gt + 256 * it < WH) *reinterpret_cast<u32x4_t*>(dst + it * 4096) = wreg[it];
  };
  auto w_needed = [&](int ws) { return ws < 2 || !w_static; };

  f32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float st_s[4][4], st_q[4][4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st_s[g4][q] = 0.f; st_q[g4][q] = 0.f; }
  float* const bias_l = reinterpret_cast<float*>(smem + 2 * HB + 2 * WB);      // this block's 32 bias values (read at every tile end)
  if (tid < 32) bias_l[tid] = p.bias ? p.bias[b * p.bsb + n0 + tid] : 0.f;

  // ---- prologue: weight step 0 complete, group 0's first halo in place, the next loads in flight ----
  Cur cC, cB, cA;            // computed last / to be stored next / the next PAIR of chunk steps to fetch (cA.cc even)
  cC.valid = 0; cC.cc = 0; cC.tix = cC.tiy = cC.tiz = 0; cC.id = 0;
  first_tile(cB, id_begin + grp);
  cA = cB;
  auto advance2 = [&](Cur& c) {                      // two chunk steps on (nch is even)
    if (!c.valid) return;
    if (c.cc + 2 < nch) { c.cc += 2; return; }
    first_tile(c, c.id + 2);
  };
  load_halo(cA); advance2(cA);                       // steps 0 and 1
  load_w(0, true);
  store_w(0);
  if (grp == 0) {
    store_halo(even_t{});
    cC = cB; advance(cB);
  }
  load_w(1, w_needed(1));
  duo_barrier();

  // ================= matrix phase: step ph >> 1 of this group =================
  auto matrix_phase = [&](int ph) __attribute__((always_inline)) {
#ifdef COMA_DUO_NO_MFMA          // (diagnostic builds only: profiles/ablate_duo.sh)
      return;
#endif
      if (__builtin_amdgcn_readfirstlane(cC.valid)) {
        // (readfirstlane: the walk's fields may live in VGPRs, and a vector compare turned this into 32 v_cndmask per phase)
        if (__builtin_amdgcn_readfirstlane(cC.cc) == 0) {          // a new tile (zeroed here, in place: zeroing in the epilogue made the compiler keep a second copy)
          asm volatile("" ::: "memory");       // (a real branch: if-converted, this is 32 v_cndmask in every matrix phase)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        }
        const char* Wc = Wl + ((ph >> 1) & 1) * WB + w_frag;
        // xr: LX = 5 the four halo rows y .. y + 3 the wave's two y-neighbour M-tiles share; LX = 4 rows ky + 2 i of the
        // two M-tiles (each two x-rows high: a tap shift by one row gives different fragments: six reads, rows 0 .. 4)
        constexpr int NXR = LX == 5 ? 4 : 5;
        uint4 wg[2][3], xr[2][NXR];
        auto rdg = [&](int g, int bf) {
          const int kz = g / 3, kx = g % 3;
          const int toff = (kz * HY * HX + kx) * P;
#pragma unroll
          for (int r = 0; r < NXR; ++r) xr[bf][r] = *reinterpret_cast<const uint4*>(Hl + a_base + toff + r * HX * P);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) wg[bf][ky] = *reinterpret_cast<const uint4*>(Wc + (kz * 9 + ky * 3 + kx) * 1024);
        };
        rdg(0, 0);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
          if (g + 1 < 9) rdg(g + 1, (g + 1) & 1);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = mma_piece(wg[g & 1][ky], xr[g & 1][RY * i + ky], acc[i], bf16_t());
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  // ================= everything else, for this group's NEXT matrix phase =================
  // (two forms, chosen at compile time: EVEN writes the even chunk step of a pair and finishes the tile the previous odd step
  // completed; ODD writes the odd step and fetches the next pair -- a run-time choice put the fetch behind a branch, and
  // loop-carried pieces behind a branch are copied through temporaries with a vmcnt(0) wait)
  auto other_phase = [&](auto odd_c, int ph) __attribute__((always_inline)) {
      constexpr int odd = decltype(odd_c)::value;
#ifdef COMA_DUO_NO_STAGE
      if (cC.valid && cC.cc == nch - 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(acc[i]));
      }
      cC = cB; advance(cB);
      return;
#endif
      store_halo(odd_c);                                // (waits for the pieces fetched two or more phases ago)
      const int ws = (ph >> 1) + 1;
      if (w_needed(ws)) store_w(ws);
      // every piece fetched two phases ago has been consumed (or, masked off, may be dropped): say so -- a piece stored under
      // a lane mask or a skipped weight refill stays "in flight" for the waitcnt pass, which then drains vmcnt in front of
      // the next fetch AND at the start of the next matrix phase
      __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): free at run time, the pieces are two phases old
      // finish the tile whose last chunk this group computed in the previous phase
      if constexpr (odd == 0) {
#ifdef COMA_DUO_NO_EPI
      if (cC.valid && cC.cc == nch - 1) {          // (diagnostic: the accumulators stay live, nothing is stored)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(acc[i]));
      }
#else
      if (cC.valid && cC.cc == nch - 1) {
        const int x0 = cC.tix * TX, y0 = cC.tiy * TY, z0 = cC.tiz * TZ;
        float4 bq[4];                                 // this lane's 16 bias values, four LDS reads in flight together
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) bq[g4] = *reinterpret_cast<const float4*>(bias_l + 8 * g4 + 4 * fh);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int j = wq * 2 + i;
          const int gz = z0 + (j >> 2), gy = y0 + (j & 3) * RY + (fr >> LX), gx = x0 + (fr & (TX - 1));
          const bool valid = gz < p.D && gy < p.H && gx < p.W;
          unsigned pk[4][2];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16_t o[4];
            const float bv[4] = {bq[g4].x, bq[g4].y, bq[g4].z, bq[g4].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              o[q] = static_cast<bf16_t>(acc[i][g4 * 4 + q] + bv[q]);
              if constexpr (STATS) {
                const float r = valid ? static_cast<float>(o[q]) : 0.f;
                st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]);
              }
            }
            const uint2 u = *reinterpret_cast<const uint2*>(o);
            pk[g4][0] = u.x; pk[g4][1] = u.y;
          }
          bf16_t* vox = yb + ((long)(gz * p.H + gy) * p.W + gx) * p.ldy + n0 + 8 * fh;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
            if (valid) *reinterpret_cast<uint4*>(vox + 16 * gp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#endif
      }
      // rotate the walk; behind the odd step of a pair, fetch the next pair
      cC = cB; advance(cB);
      if constexpr (odd == 1) { load_halo(cA); advance2(cA); }
      if (w_needed(ws + 1)) load_w(ws + 1, true);
  };
  // Two straight-line loops, one per role order (NOT one loop with a role branch: the pieces in flight are loop-carried
  // registers, and across a branch the compiler parked the loads in temporaries and copied them behind an s_waitcnt
  // vmcnt(0) -- in front of every barrier).  Both loops execute two barriers per round.
  // Enter the loops with nothing in flight: the waitcnt pass merges the prologue's pending loads (other registers than the
  // loop's) into the loop header state and then waits vmcnt(0) wherever the loop body reuses one of those registers --
  // at the start of every matrix phase.  (The builtin, not inline asm: the pass has to see this wait.)
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) only
  if (grp == 0) {        // (its first halo was written by the prologue: even step in place, the odd one follows)
#pragma unroll 1
    for (int ph = 0; ph < nphases; ph += 4) {
      matrix_phase(ph); duo_barrier(); other_phase(odd_t{}, ph + 1); duo_barrier();
      matrix_phase(ph + 2); duo_barrier(); other_phase(even_t{}, ph + 3); duo_barrier();
    }
  } else {
#pragma unroll 1
    for (int ph = 0; ph < nphases; ph += 4) {
      other_phase(even_t{}, ph); duo_barrier(); matrix_phase(ph + 1); duo_barrier();
      other_phase(odd_t{}, ph + 2); duo_barrier(); matrix_phase(ph + 3); duo_barrier();
    }
  }

  // ---- fused statistics: lanes -> wave -> block (LDS) -> the caller's record ----
  if constexpr (STATS) {
    if (p.stats) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);      // [8 waves][32 ch][2]
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = st_s[g4][q], c = st_q[g4][q];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
          if (fr == 0) { red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2] = a; red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2 + 1] = c; }
        }
      __syncthreads();
      if (tid < 32 && n0 + tid < p.N) {
        double a = 0.0, c = 0.0;
        for (int w = 0; w < 8; ++w) { a += (double)red[(w * 32 + tid) * 2]; c += (double)red[(w * 32 + tid) * 2 + 1]; }
        const int g = p.stats_inst ? b : 0;
        stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, n0 + tid, a, c);
      }
    }
  }
}

// =====================================================================================
// conv_thin16_k -- stride-1 3x3x3 convolution (forward, and data-gradient via `flip`) of the few-channel full-resolution
// layers (C <= 16 input channels, N <= 32 output channels: the prompt / UQ tail 3->16->16->1, 2->8->8->1, the 1->32 head
// convolution and their data-gradients), W >= 32.  These layers are bandwidth-bound (2 (C + N) bytes per voxel); on the
// 32x32x16 tiles of conv_mfma_halo2_k they were bound by MFMA time spent on zero padding instead (16 -> 16: half of every
// tile, 8 -> 8: 7/8).  Here:
//  * v_mfma_f32_16x16x32_bf16 with the TAPS packed along K: K = 32 = 2 taps x 16 channels (CP = 16) or 4 taps x 8
//    channels (CP = 8), so a 3x3x3 stencil is 14 or 7 MFMAs per 16 voxels x 16 output channels instead of 27 (x2 padding);
//  * the weights never touch LDS: NM x NB fragments (<= 56 VGPRs) are loaded once per kernel and stay in registers;
//  * the lane's B fragment is ONE 16-byte LDS read at `lane base of the MFMA + compile-time voxel-group offset`: the tap a
//    lane reads is a per-lane constant folded into that base;
//  * staging through buffer loads (hardware zero fill outside the volume), the next tile's pieces issued one by one
//    inside the MFMA loop; channels of a wider / padded buffer that are not this tensor's are masked at the LDS store;
//  * lane = (voxel, 4 consecutive output channels): 8-byte stores, BN / IN statistics out of the epilogue.
// Tile = 2 x 4 x 32 voxels = 16 groups of 16 voxels, 4 per wave; persistent blocks, 2 per CU.
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct Thin16P {
  const bf16_t* x; int ldx; long sbx; int D, H, W, C;
  bf16_t* y; int ldy; long sby; int N;
  const bf16_t* w; long wsb;
  const float* bias; int bsb;
  int flip;
  unsigned xbytes;
  int ntx, nty, ntz, ids_total, ids_per_block;
  int st8;             // output rows allow aligned 8-byte (4-channel) stores
  double2* stats; int stats_inst;
};

template <int CP, int NB>      // CP = padded input channels per tap (16 or 8), NB = 16-channel output blocks (1 or 2)
__global__ __launch_bounds__(256, 2) void conv_thin16_k(Thin16P p) {
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ;
  constexpr int TPM = 32 / CP, NM = (27 + TPM - 1) / TPM;        // taps per MFMA, MFMAs per voxel group
  constexpr int PCH = CP / 8;                                     // 16-byte pieces per halo row
  // LDS row pitch = the row itself (32 or 16 bytes): the four 16-lane groups of a ds_read_b128 then each read 16
  // voxels x 16 bytes out of consecutive rows, which is conflict-free on the 64 x 4-byte banks.  (The 48-byte pitch of the
  // 32x32x16 kernels made this read 2-way: 16 -> 16 at 128^3 145 us, of which ~80 us was the fragment reads.)
  constexpr int P = CP * 2;
  constexpr int HP = HV * PCH, HIT = (HP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
#ifdef COMA_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_begin = stamp_now();
  bool st_first = true;
#endif
  const int lv = lane & 15, lg = lane >> 4;                       // voxel within the group / k group (A, B) = row group (D)
  const bf16_t* xb = p.x + (long)b * p.sbx;
  const bf16_t* wb = p.w + (long)b * p.wsb;
  bf16_t* yb = p.y + (long)b * p.sby;

  // ---- weights -> registers: lane (n = lv, k group lg) holds w[tap][nb*16 + n][8 channels] of its tap in each MFMA ----
  const int tsub = CP == 16 ? lg >> 1 : lg, cofs = CP == 16 ? 8 * (lg & 1) : 0;
  // The weight tensor (27 N C bf16, <= 14 KB) is copied to LDS once, coalesced, and each lane picks its fragments there
  // (per-lane 2-byte global loads were 56..112 scattered requests per lane).
  {
    const int total = 27 * p.N * p.C;
    unsigned short* Wl = reinterpret_cast<unsigned short*>(smem);
    const unsigned short* wg = reinterpret_cast<const unsigned short*>(wb);
    if ((total & 7) == 0 && (((uintptr_t)wg) & 15) == 0) {
      for (int i = tid; i < (total >> 3); i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(wg)[i];
    } else {
      for (int i = tid; i < total; i += 256) Wl[i] = wg[i];
    }
  }
  __syncthreads();
  bf16x8_t wf[NM][NB];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int tap = m * TPM + tsub;
    const int wt = p.flip ? 26 - tap : tap;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 16 + lv;
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cofs + j;
        const bool ok = tap < 27 && n < p.N && c < p.C;
        const unsigned short v = reinterpret_cast<const unsigned short*>(smem)[ok ? (wt * p.N + n) * p.C + c : 0];
        e[j] = ok ? v : (unsigned short)0;
      }
      wf[m][nb] = (bf16x8_t){(short)e[0], (short)e[1], (short)e[2], (short)e[3], (short)e[4], (short)e[5], (short)e[6], (short)e[7]};
    }
  }
  // ---- B-fragment bases: this wave's first voxel group, this lane's voxel, its tap in MFMA m, its channel half ----
  const int gz = wid >> 1, gy0 = 2 * (wid & 1);
  int abase[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int tap = m * TPM + tsub < 27 ? m * TPM + tsub : 0;       // (padding taps have zero weights: read tap 0)
    const int kx = tap % 3, ky = (tap / 3) % 3, kz = tap / 9;
    abase[m] = (((gz + kz) * HY + gy0 + ky) * HX + lv + kx) * P + cofs * 2;
  }
  // ---- staging descriptors ----
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;
  int h_z[HIT], h_y[HIT], h_x[HIT], h_lds[HIT];
  unsigned h_boff[HIT];
#pragma unroll
  for (int it = 0; it < HIT; ++it) {
    const int piece = tid + 256 * it;
    const int row = piece / PCH, ch = piece % PCH;
    const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
    h_z[it] = piece < HP ? hz : (1 << 20); h_y[it] = hy; h_x[it] = hx;
#ifdef COMA_ABLATE_HALO      // (diagnostic: fetch the tile's own voxels only)
    if (hz < 1 || hz > TZ || hy < 1 || hy > TY || hx < 1 || hx > TX) h_z[it] = 1 << 20;
#endif
    h_boff[it] = (unsigned)((((hz * p.H + hy) * p.W + hx) * p.ldx + ch * 8) * 2);
    h_lds[it] = row * P + ch * 16;
  }
  // channels of the row's piece that belong to this tensor (the rest is padding or a neighbour's slice of a wider buffer)
  const uint4 hmask = mask8(p.C - 8 * (tid % PCH));       // (piece = tid + 256 it: the piece's channel half is a per-thread constant)
  uint4 hreg[HIT];
  auto issue = [&](int it, int z0, int y0, int x0) __attribute__((always_inline)) {
    const bool ok = (unsigned)(z0 - 1 + h_z[it]) < (unsigned)p.D && (unsigned)(y0 - 1 + h_y[it]) < (unsigned)p.H &&
                    (unsigned)(x0 - 1 + h_x[it]) < (unsigned)p.W;
    const unsigned org_b = (unsigned)((((long)((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * p.ldx) * 2);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? org_b + h_boff[it] : OOB, 0, 0);
    hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HIT; ++it)
      if (tid + 256 * it < HP) {
        const uint4 m = hmask;
        uint4 v = hreg[it];
        v.x &= m.x; v.y &= m.y; v.z &= m.z; v.w &= m.w;
        *reinterpret_cast<uint4*>(Hl + h_lds[it]) = v;
      }
  };

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  const bool do_stats = p.stats != nullptr;
  float st_s[NB][4], st_q[NB][4], bv[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb * 16 + 4 * lg + j;
      st_s[nb][j] = 0.f; st_q[nb][j] = 0.f;
      bv[nb][j] = (p.bias && n < p.N) ? p.bias[b * p.bsb + n] : 0.f;
    }
  if (id < id_end) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) issue(it, tiz * TZ, tiy * TY, tix * TX);
  }
  while (id < id_end) {
    const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    // nothing to prefetch after the last tile: a zero-range descriptor makes every piece read as zero (no branch in the loop)
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, has_next ? p.xbytes : 0, 0x00020000);
    STAMP(t0);
#ifdef COMA_STAMPS
    if (st_first) { st_acc[0] += t0 - st_begin; st_first = false; }
#endif
    __syncthreads();                       // the previous tile's fragment reads are done
    STAMP(t1);
#ifdef COMA_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(t2);
    store_halo();
    __syncthreads();
    STAMP(t3);
    f32x4_t acc[4][NB];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[q][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // B fragments run two MFMA steps (8 LDS reads) ahead of their use: left to itself the scheduler keeps ONE read in
    // flight and every 16-cycle MFMA then waits out an LDS round trip.
    uint4 fr[3][4];
    auto ldfrag = [&](int m, int q) __attribute__((always_inline)) {
      return *reinterpret_cast<const uint4*>(Hl + abase[m] + ((q >> 1) * HX + (q & 1) * 16) * P);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) { fr[0][q] = ldfrag(0, q); fr[1][q] = ldfrag(1, q); }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m + 2 < NM) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[(m + 2) % 3][q] = ldfrag(m + 2, q);
      }
      if (m < HIT) {                       // one staging piece of the next tile per MFMA step
        const int it = m;
        const bool ok = (unsigned)(ntiz * TZ - 1 + h_z[it]) < (unsigned)p.D && (unsigned)(ntiy * TY - 1 + h_y[it]) < (unsigned)p.H &&
                        (unsigned)(ntix * TX - 1 + h_x[it]) < (unsigned)p.W;
        const unsigned org_b = (unsigned)((((long)((ntiz * TZ - 1) * p.H + (ntiy * TY - 1)) * p.W + (ntix * TX - 1)) * p.ldx) * 2);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_n, ok ? org_b + h_boff[it] : OOB, 0, 0);
        hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef COMA_ABLATE_MFMA
      if (m > 0) continue;
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) {        // voxel group q of this wave: row gy0 + (q >> 1), x half q & 1
        const uint4 f = fr[m % 3][q];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[q][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m][nb], *reinterpret_cast<const bf16x8_t*>(&f), acc[q][nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(HIT <= NM, "one staging piece per MFMA step");
    STAMP(t4);
    // ---- epilogue: lane = voxel lv of group q, output channels nb*16 + 4 lg + 0..3 ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vz = z0 + gz, vy = y0 + gy0 + (q >> 1), vx = x0 + (q & 1) * 16 + lv;
      const bool valid = vz < p.D && vy < p.H && vx < p.W;
      bf16_t* vox = yb + ((long)(vz * p.H + vy) * p.W + vx) * p.ldy;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 16 + 4 * lg;
        if (n >= p.N) continue;
        bf16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = static_cast<bf16_t>(acc[q][nb][j] + bv[nb][j]);
          if (do_stats) { const float r = valid ? static_cast<float>(o[j]) : 0.f; st_s[nb][j] += r; st_q[nb][j] = fmaf(r, r, st_q[nb][j]); }
        }
#ifdef COMA_ABLATE_STORE
        if (valid && acc[q][nb][0] == 123.456f) {
#else
        if (valid) {
#endif
          if (p.st8 && n + 3 < p.N) *reinterpret_cast<uint2*>(vox + n) = *reinterpret_cast<const uint2*>(o);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < p.N) vox[n + j] = o[j];
          }
        }
      }
    }
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
    STAMP(t5);
    STAMP_ADD(1, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(3, t2, t3); STAMP_ADD(4, t3, t4); STAMP_ADD(5, t4, t5);
  }
#ifdef COMA_STAMPS
  {
    const unsigned long long st_end = stamp_now();
    if (lane == 0) {
      for (int k = 0; k < 7; ++k) atomicAdd(&g_stamps[k], st_acc[k]);
      atomicAdd(&g_stamps[7], st_end - st_begin);
      atomicAdd(&g_stamps[8], 1ull);
    }
  }
#endif
  // ---- fused statistics: the 16 voxel lanes of a row group -> wave -> block (LDS) -> partial[chunk] ----
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][32 ch][2]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = st_s[nb][j], c = st_q[nb][j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (lv == 0) { red[(wid * 32 + nb * 16 + 4 * lg + j) * 2] = a; red[(wid * 32 + nb * 16 + 4 * lg + j) * 2 + 1] = c; }
      }
    __syncthreads();
    if (tid < NB * 16 && tid < p.N) {
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 32 + tid) * 2]; c += (double)red[(w * 32 + tid) * 2 + 1]; }
      const int g = p.stats_inst ? b : 0;
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, tid, a, c);
    }
  }
}

// =====================================================================================
// conv_thin16f_k -- the fp32 twin of conv_thin16_k (exact fp32: v_mfma_f32_16x16x4_f32): stride-1 3x3x3 forward /
// data-gradient of the few-channel full-resolution layers on fp32 tensors (C <= 16, N <= 16, or C <= 8 and N <= 32).
// In fp32 mode these ran on the 32x32x2 halo kernel with half of every tile zero padding (16 -> 16: 1.05 ms at 128^3) or on
// the VALU kernels (16 -> 1: 0.94 ms, 3 -> 16 data-gradient: 0.94 ms).  K = 4 per MFMA stays inside one tap: the lane of
// k group lg holds CP/4 consecutive channels of its voxel (ONE 16- or 8-byte LDS read per tap and voxel group) and feeds
// element j of it to the tap's MFMA j; the weights (27 x 16 NB x CP floats, <= 27.6 KB) sit in LDS in the same order, one
// read per tap shared by the wave's four voxel groups.  Staging, tile walk, epilogue and statistics as in conv_thin16_k.
// =====================================================================================
struct Thin16FP {
  const float* x; int ldx; long sbx; int D, H, W, C;
  float* y; int ldy; long sby; int N;
  const float* w; long wsb;
  const float* bias; int bsb;
  int flip;
  unsigned xbytes;
  int ntx, nty, ntz, ids_total, ids_per_block;
  int st16;            // output rows allow aligned 16-byte (4-channel) stores
  double2* stats; int stats_inst;
};

template <int CP, int NB>      // CP = padded input channels (8 or 16), NB = 16-channel output blocks (1 or 2)
__global__ __launch_bounds__(256, 2) void conv_thin16f_k(Thin16FP p) {
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ;
  constexpr int CQ = CP / 4;                    // channels per lane and tap = MFMAs per tap
  constexpr int P = CP * 4;                     // LDS row pitch (bytes)
  constexpr int PCH = CP / 4;                   // 16-byte pieces per halo row
  constexpr int HP = HV * PCH, HIT = (HP + 255) / 256;
  constexpr int WROW = NB * 16;                 // weight rows (output channels) per tap
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;                              // [HV][CP] floats
  float* Wl = reinterpret_cast<float*>(smem + HV * P);     // [27][WROW][CP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int lv = lane & 15, lg = lane >> 4;
  const float* xb = p.x + (long)b * p.sbx;
  const float* wb = p.w + (long)b * p.wsb;
  float* yb = p.y + (long)b * p.sby;

  // ---- weights -> LDS (zero padded in n and c; `flip` mirrors the taps for the data-gradient) ----
  for (int i = tid; i < 27 * WROW * CP; i += 256) {
    const int c = i % CP, n = (i / CP) % WROW, t = i / (CP * WROW);
    const int wt = p.flip ? 26 - t : t;
    const bool ok = n < p.N && c < p.C;
    const float v = wb[ok ? ((long)wt * p.N + n) * p.C + c : 0];
    Wl[i] = ok ? v : 0.f;
  }
  // ---- fragment bases ----
  const int gz = wid >> 1, gy0 = 2 * (wid & 1);
  const int xbase = ((gz * HY + gy0) * HX + lv) * P + lg * CQ * 4;      // + tap offset + voxel-group offset
  const int wbase = (lv * CP + lg * CQ) * 4;                            // + (tap * WROW + nb * 16) * CP * 4
  // ---- staging descriptors ----
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;
  int h_z[HIT], h_y[HIT], h_x[HIT];
  unsigned h_boff[HIT];
#pragma unroll
  for (int it = 0; it < HIT; ++it) {
    const int piece = tid + 256 * it;
    const int row = piece / PCH, ch = piece % PCH;
    const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
    h_z[it] = piece < HP ? hz : (1 << 20); h_y[it] = hy; h_x[it] = hx;
    h_boff[it] = (unsigned)((((hz * p.H + hy) * p.W + hx) * p.ldx + ch * 4) * 4);
  }
  // channels of this thread's pieces that belong to the tensor (a piece = 4 channels; the quarter is a per-thread constant)
  const int nval = p.C - 4 * (tid % PCH);
  const uint4 hmask = make_uint4(nval > 0 ? ~0u : 0u, nval > 1 ? ~0u : 0u, nval > 2 ? ~0u : 0u, nval > 3 ? ~0u : 0u);
  uint4 hreg[HIT];
  auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < HIT; ++it)
      if (tid + 256 * it < HP) {
        uint4 v = hreg[it];
        v.x &= hmask.x; v.y &= hmask.y; v.z &= hmask.z; v.w &= hmask.w;
        *reinterpret_cast<uint4*>(Hl + (tid + 256 * it) * 16) = v;
      }
  };

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  const bool do_stats = p.stats != nullptr;
  float st_s[NB][4], st_q[NB][4], bv[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb * 16 + 4 * lg + j;
      st_s[nb][j] = 0.f; st_q[nb][j] = 0.f;
      bv[nb][j] = (p.bias && n < p.N) ? p.bias[b * p.bsb + n] : 0.f;
    }
  if (id < id_end) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * TX;
      const bool ok = (unsigned)(z0 - 1 + h_z[it]) < (unsigned)p.D && (unsigned)(y0 - 1 + h_y[it]) < (unsigned)p.H &&
                      (unsigned)(x0 - 1 + h_x[it]) < (unsigned)p.W;
      const unsigned org_b = (unsigned)((((long)((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * p.ldx) * 4);
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? org_b + h_boff[it] : OOB, 0, 0);
      hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
  while (id < id_end) {
    const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, has_next ? p.xbytes : 0, 0x00020000);
    __syncthreads();                       // the previous tile's fragment reads are done (first tile: the weight image is complete)
    store_halo();
    __syncthreads();
    f32x4_t acc[4][NB];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[q][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // fragments one tap ahead (the four voxel groups' rows + the weights of the tap)
    float xf[2][4][CQ], wfr[2][NB][CQ];
    auto rd = [&](int t, int bf) __attribute__((always_inline)) {
      const int toff = (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * P;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const char* src = Hl + xbase + toff + ((q >> 1) * HX + (q & 1) * 16) * P;
        if constexpr (CQ == 4) { const float4 v = *reinterpret_cast<const float4*>(src); xf[bf][q][0] = v.x; xf[bf][q][1] = v.y; xf[bf][q][2] = v.z; xf[bf][q][3] = v.w; }
        else if constexpr (CQ == 2) { const float2 v = *reinterpret_cast<const float2*>(src); xf[bf][q][0] = v.x; xf[bf][q][1] = v.y; }
        else xf[bf][q][0] = *reinterpret_cast<const float*>(src);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const char* src = reinterpret_cast<const char*>(Wl) + wbase + (t * WROW + nb * 16) * CP * 4;
        if constexpr (CQ == 4) { const float4 v = *reinterpret_cast<const float4*>(src); wfr[bf][nb][0] = v.x; wfr[bf][nb][1] = v.y; wfr[bf][nb][2] = v.z; wfr[bf][nb][3] = v.w; }
        else if constexpr (CQ == 2) { const float2 v = *reinterpret_cast<const float2*>(src); wfr[bf][nb][0] = v.x; wfr[bf][nb][1] = v.y; }
        else wfr[bf][nb][0] = *reinterpret_cast<const float*>(src);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (t + 1 < 27) rd(t + 1, (t + 1) & 1);
      if (t < HIT) {                       // one staging piece of the next tile per tap
        const int it = t;
        const bool ok = (unsigned)(ntiz * TZ - 1 + h_z[it]) < (unsigned)p.D && (unsigned)(ntiy * TY - 1 + h_y[it]) < (unsigned)p.H &&
                        (unsigned)(ntix * TX - 1 + h_x[it]) < (unsigned)p.W;
        const unsigned org_b = (unsigned)((((long)((ntiz * TZ - 1) * p.H + (ntiy * TY - 1)) * p.W + (ntix * TX - 1)) * p.ldx) * 4);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_n, ok ? org_b + h_boff[it] : OOB, 0, 0);
        hreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < CQ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[q][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[t & 1][nb][j], xf[t & 1][q][j], acc[q][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(HIT <= 27, "one staging piece per tap");
    // ---- epilogue: lane = voxel lv of group q, output channels nb*16 + 4 lg + 0..3 ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vz = z0 + gz, vy = y0 + gy0 + (q >> 1), vx = x0 + (q & 1) * 16 + lv;
      const bool valid = vz < p.D && vy < p.H && vx < p.W;
      float* vox = yb + ((long)(vz * p.H + vy) * p.W + vx) * p.ldy;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 16 + 4 * lg;
        if (n >= p.N) continue;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = acc[q][nb][j] + bv[nb][j];
          if (do_stats) { const float r = valid ? o[j] : 0.f; st_s[nb][j] += r; st_q[nb][j] = fmaf(r, r, st_q[nb][j]); }
        }
        if (valid) {
          if (p.st16 && n + 3 < p.N) *reinterpret_cast<float4*>(vox + n) = make_float4(o[0], o[1], o[2], o[3]);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < p.N) vox[n + j] = o[j];
          }
        }
      }
    }
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- fused statistics (as conv_thin16_k) ----
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][32 ch][2]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = st_s[nb][j], c = st_q[nb][j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (lv == 0) { red[(wid * 32 + nb * 16 + 4 * lg + j) * 2] = a; red[(wid * 32 + nb * 16 + 4 * lg + j) * 2 + 1] = c; }
      }
    __syncthreads();
    if (tid < NB * 16 && tid < p.N) {
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 32 + tid) * 2]; c += (double)red[(w * 32 + tid) * 2 + 1]; }
      const int g = p.stats_inst ? b : 0;
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, tid, a, c);
    }
  }
}

static bool aligned16(const void* p);
static bool thin16f_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  static const bool on = []{ const char* e = getenv("COMA_THIN16F"); return !(e && e[0] == '0'); }();
  return on && d->ksize == 3 && d->stride == 1 && x->dtype == COMA_F32 && y->dtype == COMA_F32 && x->W >= 32 && x->C <= 16 &&
         y->C <= (x->C > 8 ? 16 : 32) && x->ld % 4 == 0 && x->sb % 4 == 0 && (!x->data || aligned16(x->data)) &&
         (unsigned long long)t_vox(x) * x->ld * 4 < 0x7fff0000ull && (long)t_vox(y) * y->ld < (1L << 31);
}

static int conv_thin16f(const coma_conv_desc* d, const coma_tensor* x, const void* wk, const float* bias, const coma_tensor* y,
                        hipStream_t s, double2* stats, int stats_inst, int* stats_chunks) {
  Thin16FP q;
  q.x = (const float*)x->data; q.ldx = (int)x->ld; q.sbx = x->sb; q.D = x->D; q.H = x->H; q.W = x->W; q.C = x->C;
  q.y = (float*)y->data; q.ldy = (int)y->ld; q.sby = y->sb; q.N = y->C;
  q.w = (const float*)wk; q.wsb = d->per_sample_w ? 27L * y->C * x->C : 0;
  q.bias = bias; q.bsb = d->per_sample_w ? y->C : 0;
  q.flip = d->form == 1;
  q.xbytes = (unsigned)((unsigned long long)t_vox(x) * x->ld * 4);
  q.st16 = y->ld % 4 == 0 && y->sb % 4 == 0 && (((uintptr_t)y->data) & 15) == 0;
  q.ntx = (q.W + 31) / 32; q.nty = (q.H + 3) / 4; q.ntz = (q.D + 1) / 2;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  int gx = 512 / x->B;                                 // two blocks per CU, one round (a tile is 16x the bf16 MFMA time)
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  q.stats = nullptr; q.stats_inst = stats_inst;
  if (stats) { q.stats = stats; *stats_chunks = 1; }
  const dim3 grid((unsigned)gx, 1, (unsigned)x->B);
  const bool c16 = q.C > 8, n32 = q.N > 16;
  const int cp = c16 ? 16 : q.C > 4 ? 8 : 4, wrow = n32 ? 32 : 16;      // (C <= 4: one MFMA per tap)
  const size_t lds = (size_t)34 * 6 * 4 * cp * 4 + (size_t)27 * wrow * cp * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_thin16f_k<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_k<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_k<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_k<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_k<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  coma_set_kernel_tag("conv_thin16f_k<%d, %d>", cp, n32 ? 2 : 1);
  if (c16) hipLaunchKernelGGL((conv_thin16f_k<16, 1>), grid, dim3(256), lds, s, q);
  else if (cp == 8) { if (n32) hipLaunchKernelGGL((conv_thin16f_k<8, 2>), grid, dim3(256), lds, s, q); else hipLaunchKernelGGL((conv_thin16f_k<8, 1>), grid, dim3(256), lds, s, q); }
  else { if (n32) hipLaunchKernelGGL((conv_thin16f_k<4, 2>), grid, dim3(256), lds, s, q); else hipLaunchKernelGGL((conv_thin16f_k<4, 1>), grid, dim3(256), lds, s, q); }
  COMA_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================
// conv_mfma_pw_k -- 1x1x1 convolution for small channel counts (8 <= C <= 64, N <= 64): the attention
// gate's W_g / W_x (C -> C/2) and their data-gradients at full resolution.  HBM-bound: no LDS at all --
// the weight fragments live in registers for the whole kernel, each lane streams 16-byte channel chunks
// of its voxel straight into the MFMA B operand (weights x voxels orientation), and stores four
// channels per 8-byte write.  Replaces the VALU direct kernel (16 x 32 FMAs per voxel) on these layers.
// =====================================================================================
struct PwP {
  const bf16_t* x; int ldx; long sbx; long V; int C;
  bf16_t* y; int ldy; long sby; int N;
  const bf16_t* w; long wsb;
  const float* bias; int bsb;
  int st8;
  int st16;            // output rows allow aligned 16-byte (8-channel) accesses
  double2* stats;      // optional fused {sum, sumsq} partials of the stored outputs, [chunk][G][N] (as conv_mfma_halo2_k)
  int stats_inst;
  int accum;           // y += conv(x)
};

template <int KS, int NT>   // KS = ceil(C / 16) K steps, NT = ceil(N / 32) row tiles
__global__ __launch_bounds__(256) void conv_mfma_pw_k(PwP p) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int b = blockIdx.y;
  const bf16_t* xb = p.x + (long)b * p.sbx;
  const bf16_t* wb = p.w + (long)b * p.wsb;
  bf16_t* yb = p.y + (long)b * p.sby;
  // weight fragments: lane (row n = fr, half fh) holds w[n][16 ks + 8 fh .. +8]
  bf16x8_t wf[NT][KS];
  float bv[NT][4][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + fh * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < p.N && c < p.C) v = *reinterpret_cast<const uint4*>(wb + (long)n * p.C + c);
      wf[j][ks] = *reinterpret_cast<const bf16x8_t*>(&v);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = j * 32 + 8 * g4 + 4 * fh + q;
        bv[j][g4][q] = (p.bias && nn < p.N) ? p.bias[b * p.bsb + nn] : 0.f;
      }
  }
  const bool do_stats = p.stats != nullptr;
  float st_s[NT][4][4], st_q[NT][4][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int q = 0; q < 4; ++q) { st_s[j][g4][q] = 0.f; st_q[j][g4][q] = 0.f; }
  const long mtiles = (p.V + 31) / 32;
  for (long mt = (long)blockIdx.x * 4 + wid; mt < mtiles; mt += (long)gridDim.x * 4) {
    const long v = mt * 32 + fr;
    const bool live = v < p.V;
    bf16x8_t xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + fh * 8;
      uint4 t = make_uint4(0, 0, 0, 0);
      if (live && c < p.C) t = *reinterpret_cast<const uint4*>(xb + v * p.ldx + c);
      xf[ks] = *reinterpret_cast<const bf16x8_t*>(&t);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x16_t acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], xf[ks], acc, 0, 0, 0);
      const int nval = p.N - j * 32 < 32 ? p.N - j * 32 : 32;
      if (p.st16 && (nval & 15) == 0) {
        // 16 or 32 valid channels: the two half-waves exchange 4-channel groups (v_permlane32_swap) so that a lane owns 8
        // CONSECUTIVE channels per 16-channel half: one 16-byte store (and, accumulating, one 16-byte load) instead of two
        // 8-byte ones -- 8-byte accesses run at 0.5-0.7 of the 16-byte rate, and the accumulate mode doubles them
        bf16_t* vox = yb + v * p.ldy + j * 32 + 8 * fh;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          if (16 * gp >= nval) continue;                 // (wave-uniform)
          float of[2][4];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) of[h][q] = acc[(2 * gp + h) * 4 + q] + bv[j][2 * gp + h][q];
          if (p.accum) {                                  // (wave-uniform) y += : see conv_mfma_tconv_k
            uint4 u = make_uint4(0, 0, 0, 0);
            if (live) u = *reinterpret_cast<const uint4*>(vox + 16 * gp);
            const auto s0 = __builtin_amdgcn_permlane32_swap(u.x, u.z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(u.y, u.w, false, false);
            const unsigned w0[2] = {s0[0], s1[0]}, w1[2] = {s0[1], s1[1]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              of[0][2 * h] += __uint_as_float(w0[h] << 16); of[0][2 * h + 1] += __uint_as_float(w0[h] & 0xffff0000u);
              of[1][2 * h] += __uint_as_float(w1[h] << 16); of[1][2 * h + 1] += __uint_as_float(w1[h] & 0xffff0000u);
            }
          }
          unsigned pk[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              o[q] = static_cast<bf16_t>(of[h][q]);
              if (do_stats) {
                const float r = live ? static_cast<float>(o[q]) : 0.f;
                st_s[j][2 * gp + h][q] += r; st_q[j][2 * gp + h][q] = fmaf(r, r, st_q[j][2 * gp + h][q]);
              }
            }
            const uint2 u2 = *reinterpret_cast<const uint2*>(o);
            pk[h][0] = u2.x; pk[h][1] = u2.y;
          }
          const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
          if (live) *reinterpret_cast<uint4*>(vox + 16 * gp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      } else if (live) {
        bf16_t* dst = yb + v * p.ldy + j * 32 + 4 * fh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (j * 32 + 8 * g4 >= p.N) continue;          // (wave-uniform) zero-padded channel groups: nothing to convert or sum
          bf16_t o[4];
          float old[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.accum) {                                  // (wave-uniform) y += : read what is there
            if (p.st8 && j * 32 + 8 * g4 + 4 * fh + 3 < p.N) {
              const uint2 u = *reinterpret_cast<const uint2*>(dst + 8 * g4);
              old[0] = __uint_as_float(u.x << 16); old[1] = __uint_as_float(u.x & 0xffff0000u);
              old[2] = __uint_as_float(u.y << 16); old[3] = __uint_as_float(u.y & 0xffff0000u);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (j * 32 + 8 * g4 + 4 * fh + q < p.N) old[q] = static_cast<float>(dst[8 * g4 + q]);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = static_cast<bf16_t>(acc[g4 * 4 + q] + bv[j][g4][q] + old[q]);
            if (do_stats) {
              const float r = static_cast<float>(o[q]);
              st_s[j][g4][q] += r; st_q[j][g4][q] = fmaf(r, r, st_q[j][g4][q]);
            }
          }
          if (p.st8 && j * 32 + 8 * g4 + 4 * fh + 3 < p.N) *reinterpret_cast<uint2*>(dst + 8 * g4) = *reinterpret_cast<uint2*>(o);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (j * 32 + 8 * g4 + 4 * fh + q < p.N) dst[8 * g4 + q] = o[q];
          }
        }
      }
    }
  }
  // ---- fused statistics: lanes -> wave -> block -> partial[chunk] (same scheme as conv_mfma_halo2_k) ----
  if (p.stats) {
    __shared__ float red[4 * 64 * 2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = st_s[j][g4][q], c = st_q[j][g4][q];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
          const int n = j * 32 + 8 * g4 + 4 * fh + q;
          if (fr == 0) { red[(wid * 64 + n) * 2] = a; red[(wid * 64 + n) * 2 + 1] = c; }
        }
    __syncthreads();
    if (tid < NT * 32 && tid < p.N) {
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 64 + tid) * 2]; c += (double)red[(w * 64 + tid) * 2 + 1]; }
      const int g = p.stats_inst ? b : 0;
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, tid, a, c);
    }
  }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static bool halo_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  // stride-1 3x3x3 (either gather form); any channel counts; rows wide enough to fill 32-voxel M-tiles
  // 8^3 grids with >= 32 channels both sides run faster on the gather kernel with split-K (512 -> 512: 97 vs 144 us)
  if (x->W < 16 && x->C % 32 == 0 && y->C % 32 == 0) return false;
  return d->ksize == 3 && d->stride == 1 && x->W >= 8 && (long)x->H * x->W >= 32 &&
         (x->C >= 8 || y->C >= 8 || x->C * y->C >= 8);
}

static bool pw_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  return d->ksize == 1 && d->stride == 1 && x->C >= 8 && x->C <= 64 && x->C % 8 == 0 && y->C >= 4 && y->C <= 64 &&
         x->ld % 8 == 0 && x->sb % 8 == 0 && (!x->data || aligned16(x->data)) && t_vox(x) >= 4096;
}


// =====================================================================================
// conv_mfma_tconv_k -- stride-2 3x3x3 TRANSPOSED convolution (the up-convolutions, attn_unet_data_parallel.py:120-131,223)
// and the data-gradient of the stride-2 encoder convolutions (:318-325), coarse width >= 32, with the COARSE halo staged
// once per channel chunk in LDS (round 3; these layers ran on conv_mfma_gather_k: every K step re-gathered from global,
// 3.0x its algorithmic HBM bytes, 8 % of the bf16 peak).
//   out[2m + p] = sum over taps t with (2m + p + 1 - t) even of x[(2m + p + 1 - t) / 2] w[t], per dimension:
//     p = 0: tap 1 at x[m];   p = 1: tap 0 at x[m + 1] and tap 2 at x[m]
// so a coarse tile of 2 x 4 x 32 voxels plus a ONE-voxel halo on the high side of each dimension (3 x 5 x 33 = 495 rows)
// yields all 8 output-parity classes of its 4 x 8 x 64 fine voxels: 27 (class, tap) pairs = 27 MFMA K-steps per coarse
// voxel and 32-channel chunk, none of them a zero-stuffed tap.  The pairs are walked by halo offset delta in {0,1}^3: the
// voxel fragment of a delta is read once and feeds up to 8 pairs -- about one LDS fragment read per MFMA (the stride-1
// kernel needs 1.5).  The classes are processed in two balanced passes of 4 (13 + 14 pairs; see TC_* below): a wave keeps
// 4 classes x 2 M-tiles = 128 accumulator registers and each pass streams the channel chunks through one 75 KB LDS image
// (the pass's 14 weight slots x 32 outputs 35 KB + halo 39 KB), one block per CU.
// Everything else follows conv_mfma_halo2_k: persistent blocks over XCD-contiguous tile runs, buffer loads with hardware
// zero fill, the next step's 15 staging pieces issued inside the MFMA loop, 80-byte LDS rows, (weights x voxels)
// orientation with 16-byte stores, norm statistics out of the epilogue (STATS), COMA_ACCUMULATE.
// Measured (MI355X, 2 x 64^3 x 64 -> 2 x 128^3 x 32, 58 GFLOP, 335 MB): 139 us = 416 TFLOP/s against 298 us on the gather
// kernel; 128 -> 64 to 64^3: 51 us = 567 TFLOP/s against 94 us.  The layer is output-heavy (8 fine voxels per coarse one:
// 256 outputs per lane and tile next to 216 MFMAs) and memory-side bound: PMC of the first version 262 MB written (exact)
// but 130-260 MB fetched for 67 MB of input (the output stream evicts the halo between the two passes; TCC hit rate 61 %),
// 41 % of the wave cycles waiting.  Two blocks per CU (a 256-register variant, 12 spilled registers) measured 147 us /
// 47 us: no gain where it matters, so one block per CU stays.  Next: both channel chunks of a tile's halo resident for its
// two passes (C = 64: 115 KB), or LDS-DMA staging into a double-buffered image.
// =====================================================================================
struct TconvP {
  const void* x; int ldx; long sbx; int D, H, W, C;      // coarse input
  void* y; int ldy; long sby; int Do, Ho, Wo, N;         // fine output (<= 2 x coarse per dimension)
  const void* w; long wsb;                               // [b][27][N][C]
  const float* bias; int bsb;
  int ntx, nty, ntz, ids_total, ids_per_block;
  unsigned xbytes, wbytes;
  int accum;
  double2* stats; int stats_inst;
};

// The 27 (class, tap) pairs in two balanced groups of parity classes -- A = {0, 3, 5, 6} (13 pairs), B = {1, 2, 4, 7} (14
// pairs) -- each walked by halo offset delta.  A wave holds the accumulators of ONE group at a time (4 classes x 2 M-tiles =
// 128 registers; all 8 classes at once spilled 135 registers next to the 22 staging pieces in flight), so a tile is two
// passes over its channel chunks, each staging the halo and only the 13 / 14 taps it needs (75 KB of LDS).  Every tap
// belongs to exactly one pair: LDS weight slot k of a pass holds the tap of pair k.
__device__ constexpr int TC_NP[2] = {13, 14};
__device__ constexpr int TC_DELTA[2][14] = {{0, 0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6, 6}, {0, 0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6, 7}};
__device__ constexpr int TC_LCLS[2][14] = {{0, 1, 2, 3, 1, 2, 1, 3, 1, 2, 3, 2, 3, 3}, {0, 1, 2, 3, 0, 3, 1, 3, 3, 2, 3, 3, 3, 3}};
__device__ constexpr int TC_TAP[2][14] = {{13, 17, 23, 25, 15, 21, 11, 19, 9, 5, 7, 3, 1, 27}, {14, 16, 22, 26, 12, 24, 10, 20, 18, 4, 8, 6, 2, 0}};
__device__ constexpr int TC_CLS[2][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}};

// STATS: the fused norm statistics (forward of the up-convolutions); without them the kernel also takes COMA_ACCUMULATE
// (data gradients).  Two instantiations because the statistics' 32 accumulators are live across the whole tile loop: the
// variant without them needs no scratch.
template <typename T, bool STATS>
__global__ __launch_bounds__(256, 1) void conv_mfma_tconv_k(TconvP p) {
  constexpr int EPB = elem<T>::EPB;
  constexpr bool F32 = EPB == 4;
  constexpr int CK = 4 * EPB;                          // channels per 64-byte LDS row: 32 bf16 / 16 fp32
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 1, HY = TY + 1, HZ = TZ + 1, HV = HX * HY * HZ;   // 495 halo rows
  constexpr int P = 80;                                // LDS row pitch (bytes): conflict-free ds_read_b128 over 16 consecutive rows
  constexpr int HP = HV * 4, HIT = (HP + 255) / 256;   // 1980 halo pieces of 16 bytes -> 8 per thread
  constexpr int WS = 14, WP = WS * 32 * 4, WIT = WP / 256;   // 14 weight slots = 1792 pieces -> 7 per thread
  static_assert(WP % 256 == 0 && HIT + WIT <= 2 * 13, "at most two prefetch pieces per (class, tap) pair");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;                                     // [14 * 32][80]  (first: its fragment offsets then fit the ds_read immediate)
  char* Hl = smem + WS * 32 * P;                       // [HV][80]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, n0 = blockIdx.y * 32;
  const int fr = lane & 31, fh = lane >> 5;
  const T* xb = static_cast<const T*>(p.x) + (long)b * p.sbx;
  const T* wb = static_cast<const T*>(p.w) + (long)b * p.wsb;
  T* yb = static_cast<T*>(p.y) + (long)b * p.sby;
  const int nchunks = p.C / CK;
  constexpr unsigned OOB = 0x7fff0000u;

  // ---- staging descriptors (tile independent).  The layer is output-heavy (8 fine voxels per coarse one): next to 216
  // MFMAs per tile every VALU instruction counts (first version: 13 VALU per MFMA, 26 % of the wave cycles).  Halo piece:
  // BYTE offset in x relative to the tile origin; its halo coordinates as three 9-bit fields hz << 20 | hy << 10 | hx, so
  // that the in-volume test of a piece is ONE subtraction against the packed per-tile limits (guard bits 9 / 19 / 29
  // survive iff every field is within its limit); its LDS byte offset ----
  unsigned h_boff[HIT], h_zyx[HIT];
#pragma unroll
  for (int it = 0; it < HIT; ++it) {
    const int piece = tid + 256 * it;
    const int row = piece >> 2, ch = piece & 3;
    const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
    h_boff[it] = (unsigned)((((hz * p.H + hy) * p.W + hx) * p.ldx + ch * EPB) * (int)sizeof(T));
    h_zyx[it] = piece < HP ? ((unsigned)hz << 20) | ((unsigned)hy << 10) | (unsigned)hx : 0x1ff00000u;      // (invalid: beyond any limit)
  }
  const int h_lds0 = (tid >> 2) * P + (tid & 3) * 16;      // piece `it` of this thread: + it * 64 rows (an immediate, not a register)
  constexpr unsigned GUARD = (1u << 29) | (1u << 19) | (1u << 9);
  // weight piece `it` of this thread: slot 2 it + (tid >> 7), output row n = (tid >> 2) & 31, 16-byte chunk tid & 3; the
  // slot's tap differs between the two passes: both byte offsets are kept (OOB = the empty 14th slot of pass A)
  const int w_s0 = tid >> 7, w_n = (tid >> 2) & 31, w_ch = tid & 3;
  const int w_lds0 = (w_s0 * 32 + w_n) * P + w_ch * 16;
  constexpr int W_LSTEP = 2 * 32 * P;
  // (the 14 byte offsets of a thread's weight pieces are rebuilt at every fetch -- one multiply-add from a compile-time tap
  // pair -- instead of living in 14 registers: with the statistics the kernel spilled 28)
  unsigned w_row = (unsigned)(((long)(n0 + w_n) * p.C + w_ch * EPB) * (long)sizeof(T));
  const unsigned w_tap = (unsigned)((long)p.N * p.C * (long)sizeof(T));         // bytes between two taps
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wb), 0, p.wbytes, 0x00020000);

  // fragment read bases: voxel operand = this lane's coarse voxel row in each of the wave's two M-tiles
  int a_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wid * 2 + i;
    a_base[i] = (((j >> 2) * HY + (j & 3)) * HX + fr) * P + fh * 16;
  }
  const int w_base = fr * P + fh * 16;

  typedef unsigned tc_u32x4_t __attribute__((ext_vector_type(4)));      // (native vectors: see conv_mfma_duo_k)
  tc_u32x4_t hreg[HIT], wreg[WIT];
  // lim: the tile's packed limits ((D - z0 - 1) << 20 | (H - y0 - 1) << 10 | (W - x0 - 1)) | GUARD, fields clamped to 511
  auto tile_lim = [&](int z0, int y0, int x0) -> unsigned {
    const int lz = p.D - z0 - 1 < 511 ? p.D - z0 - 1 : 511, ly = p.H - y0 - 1 < 511 ? p.H - y0 - 1 : 511, lx = p.W - x0 - 1 < 511 ? p.W - x0 - 1 : 511;
    return ((unsigned)lz << 20) | ((unsigned)ly << 10) | (unsigned)lx | GUARD;
  };
  auto halo_piece = [&](const __amdgpu_buffer_rsrc_t& rs, int it, unsigned lim, unsigned org_b) -> tc_u32x4_t {
    const bool ok = ((lim - h_zyx[it]) & GUARD) == GUARD;
    const unsigned voff = ok ? org_b + h_boff[it] : OOB;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
  };
  // slot 2 it + w_s0 of pass `grp` holds tap TC_TAP[grp][slot] (27 = the empty 14th slot of group A: reads as zero)
  auto w_piece = [&](const __amdgpu_buffer_rsrc_t& rs, int it, int grp, int c0_b) -> tc_u32x4_t {
    const int t0 = grp ? TC_TAP[1][2 * it] : TC_TAP[0][2 * it], t1 = grp ? TC_TAP[1][2 * it + 1] : TC_TAP[0][2 * it + 1];
    const int tap = w_s0 ? t1 : t0;
    asm volatile("" : "+v"(w_row));                  // (not hoisted back into loop-invariant registers)
    const unsigned voff = tap < 27 ? w_row + (unsigned)tap * w_tap : OOB;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, c0_b, 0);
  };
  auto tile_org = [&](int z0, int y0, int x0, int c0) -> unsigned {
    return (unsigned)((((long)(z0 * p.H + y0) * p.W + x0) * p.ldx + c0) * (long)sizeof(T));
  };

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id >= id_end) return;
  {
    const unsigned org = tile_org(tiz * TZ, tiy * TY, tix * TX, 0), lim = tile_lim(tiz * TZ, tiy * TY, tix * TX);
#pragma unroll
    for (int it = 0; it < HIT; ++it) hreg[it] = halo_piece(rs_x, it, lim, org);
#pragma unroll
    for (int it = 0; it < WIT; ++it) wreg[it] = w_piece(rs_w, it, 0, 0);
  }

  constexpr bool do_stats = STATS;
  const bool accum = !STATS && p.accum;
  float st_s[4][4], st_q[4][4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st_s[g4][q] = 0.f; st_q[g4][q] = 0.f; }

  while (id < id_end) {
    const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;

#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      f32x16_t acc[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[c][i][e] = 0.f;

      for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();                       // all waves finished reading the previous halo / weights
#pragma unroll
        for (int it = 0; it < HIT; ++it)
          if (tid + 256 * it < HP) *reinterpret_cast<tc_u32x4_t*>(Hl + h_lds0 + it * 64 * P) = hreg[it];
#pragma unroll
        for (int it = 0; it < WIT; ++it) *reinterpret_cast<tc_u32x4_t*>(Wl + w_lds0 + it * W_LSTEP) = wreg[it];
        __syncthreads();
        // what to prefetch while this step computes: the next chunk of this pass, chunk 0 of the tile's second pass, or
        // chunk 0 / pass A of the next tile; through descriptors whose range is zero when there is nothing left (no branch)
        const bool same_pass = cc + 1 < nchunks;
        const bool same_tile = same_pass || grp == 0;
        const bool pref = same_tile || has_next;
        const int pz = same_tile ? z0 : ntiz * TZ, py = same_tile ? y0 : ntiy * TY, px = same_tile ? x0 : ntix * TX;
        const int pc0 = same_pass ? (cc + 1) * CK : 0;
        const int pgrp = same_pass ? grp : 1 - grp;
        const __amdgpu_buffer_rsrc_t rs_xp = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, pref ? p.xbytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wb), 0, pref ? p.wbytes : 0, 0x00020000);
        const unsigned porg = tile_org(pz, py, px, pc0), plim = tile_lim(pz, py, px);
        const int pc0_b = pc0 * (int)sizeof(T);

        uint4 wv[2][2], xv[2][2][2];           // fragments one pair (weights) / one delta (voxels) ahead
        auto rd_w = [&](int k, int bf) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wv[bf][ks] = *reinterpret_cast<const uint4*>(Wl + w_base + k * 32 * P + ks * 32);
        };
        auto rd_x = [&](int di, int bf) {
          const int doff = ((((di >> 2) & 1) * HY + ((di >> 1) & 1)) * HX + (di & 1)) * P;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xv[bf][i][ks] = *reinterpret_cast<const uint4*>(Hl + a_base[i] + doff + ks * 32);
        };
        rd_x(0, 0);
        rd_w(0, 0);
        constexpr int NPMAX = 14;
#pragma unroll
        for (int k = 0; k < NPMAX; ++k) {
          if (k >= TC_NP[grp]) continue;         // (compile time: grp and k are unrolled)
          const int di = TC_DELTA[grp][k], lc = TC_LCLS[grp][k];
          if (k + 1 < TC_NP[grp]) {
            rd_w(k + 1, (k + 1) & 1);
            if (TC_DELTA[grp][k + 1] != di) rd_x(TC_DELTA[grp][k + 1], TC_DELTA[grp][k + 1] & 1);
          }
          // prefetch pieces: 15 per step over 13 / 14 pairs (one per pair, the last ones two)
          if (k < WIT) wreg[k] = w_piece(rs_wp, k, pgrp, pc0_b);
          else if (k - WIT < HIT) hreg[k - WIT] = halo_piece(rs_xp, k - WIT, plim, porg);
          if (k == TC_NP[grp] - 1) {
#pragma unroll
            for (int r = TC_NP[grp] - WIT; r < HIT; ++r) hreg[r] = halo_piece(rs_xp, r, plim, porg);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[lc][i] = mma_piece(wv[k & 1][ks], xv[di & 1][i][ks], acc[lc][i], T());
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- epilogue of this pass's 4 classes: lane = one coarse voxel of each M-tile, 4 groups of 4 consecutive channels ----
      const bool has_bias = p.bias != nullptr;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wid * 2 + i;
        const int cz = z0 + (j >> 2), cy = y0 + (j & 3), cx = x0 + fr;
        const bool cin = cz < p.D && cy < p.H && cx < p.W;
        const bool vz1 = 2 * cz + 1 < p.Do, vy1 = 2 * cy + 1 < p.Ho, vx1 = 2 * cx + 1 < p.Wo;      // (odd fine sizes: the last odd plane is absent)
        const long vbase = ((long)(2 * cz * p.Ho + 2 * cy) * p.Wo + 2 * cx) * p.ldy + n0;         // fine voxel (2cz, 2cy, 2cx)
#pragma unroll
        for (int lc = 0; lc < 4; ++lc) {
          __builtin_amdgcn_sched_barrier(0);     // one class at a time: interleaved, the 8 unrolled instances spill
          const int cls = TC_CLS[grp][lc];
          const bool valid = cin && (!(cls & 4) || vz1) && (!(cls & 2) || vy1) && (!(cls & 1) || vx1);
          const long voff = vbase + ((long)(((cls >> 2) & 1) * p.Ho + ((cls >> 1) & 1)) * p.Wo + (cls & 1)) * p.ldy;   // (scalar class offset)
          float of[4][4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int q = 0; q < 4; ++q) of[g4][q] = acc[lc][i][g4 * 4 + q];
          if (has_bias) {                        // (uniform)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
              for (int q = 0; q < 4; ++q) of[g4][q] += p.bias[b * p.bsb + n0 + 8 * g4 + 4 * fh + q];
          }
          if constexpr (F32) {
            float* dst = reinterpret_cast<float*>(yb) + voff + 4 * fh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
              if (accum && valid) old = *reinterpret_cast<const float4*>(dst + 8 * g4);
              const float o[4] = {of[g4][0] + old.x, of[g4][1] + old.y, of[g4][2] + old.z, of[g4][3] + old.w};
              if (do_stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float r = valid ? o[q] : 0.f; st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]); }
              }
              if (valid) *reinterpret_cast<float4*>(dst + 8 * g4) = make_float4(o[0], o[1], o[2], o[3]);
            }
          } else {
            // the two half-waves exchange 4-channel groups (v_permlane32_swap): every lane then owns 8 CONSECUTIVE channels
            // of its voxel per 16-channel half and writes them with one 16-byte store (as conv_mfma_halo2_k)
            bf16_t* vox = reinterpret_cast<bf16_t*>(yb) + voff + 8 * fh;
            if (accum) {
              // y += : what is already there, brought into the accumulators' lane layout -- a lane reads the 8 consecutive
              // channels it will store (fh = 0: 16 gp + 0..7, fh = 1: 16 gp + 8..15) and the halves swap back the 4-channel
              // groups that belong to the other one (the inverse of the exchange in front of the store)
#pragma unroll
              for (int gp = 0; gp < 2; ++gp) {
                uint4 u = make_uint4(0, 0, 0, 0);
                if (valid) u = *reinterpret_cast<const uint4*>(vox + 16 * gp);
                const auto s0 = __builtin_amdgcn_permlane32_swap(u.x, u.z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(u.y, u.w, false, false);
                const unsigned w0[2] = {s0[0], s1[0]}, w1[2] = {s0[1], s1[1]};       // channel group 2 gp / 2 gp + 1 of this lane
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  of[2 * gp][2 * h] += __uint_as_float(w0[h] << 16); of[2 * gp][2 * h + 1] += __uint_as_float(w0[h] & 0xffff0000u);
                  of[2 * gp + 1][2 * h] += __uint_as_float(w1[h] << 16); of[2 * gp + 1][2 * h + 1] += __uint_as_float(w1[h] & 0xffff0000u);
                }
              }
            }
            unsigned pk[4][2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              bf16_t o[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                o[q] = static_cast<bf16_t>(of[g4][q]);
                if (do_stats) { const float r = valid ? static_cast<float>(o[q]) : 0.f; st_s[g4][q] += r; st_q[g4][q] = fmaf(r, r, st_q[g4][q]); }
              }
              const uint2 u = *reinterpret_cast<const uint2*>(o);
              pk[g4][0] = u.x; pk[g4][1] = u.y;
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
              if (valid) *reinterpret_cast<uint4*>(vox + 16 * gp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
          }
        }
      }
    }
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- fused statistics (as conv_mfma_halo2_k) ----
  if (STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = st_s[g4][q], c = st_q[g4][q];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (fr == 0) { red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2] = a; red[(wid * 32 + 8 * g4 + 4 * fh + q) * 2 + 1] = c; }
      }
    __syncthreads();
    if (tid < 32 && n0 + tid < p.N) {
      double a = 0.0, c = 0.0;
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 32 + tid) * 2]; c += (double)red[(w * 32 + tid) * 2 + 1]; }
      const int g = p.stats_inst ? b : 0;
      stat_add(p.stats, p.stats_inst ? p.stats_inst : 1, p.N, g, n0 + tid, a, c);
    }
  }
}

static bool tconv_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (d->form != 1 || d->stride != 2 || d->ksize != 3 || x->dtype != y->dtype) return false;
  static const bool off = getenv("COMA_NO_TCONV") != nullptr;      // (A/B measurements against the gather kernel)
  if (off) return false;
  const int es = esize(x->dtype), ck = 64 / es;
  if (x->W < 32 || x->C % ck || y->C % 32) return false;
  if ((x->ld * es) % 16 || (x->sb * es) % 16 || (y->ld * es) % 16 || (y->sb * es) % 16) return false;
  if ((x->data && !aligned16(x->data)) || (y->data && !aligned16(y->data))) return false;
  if (y->D > 2 * x->D || y->H > 2 * x->H || y->W > 2 * x->W || y->D < 2 * x->D - 1 || y->H < 2 * x->H - 1 || y->W < 2 * x->W - 1) return false;
  return (unsigned long long)t_vox(x) * x->ld * es < 0x7fff0000ull && (long)t_vox(y) * y->ld < (1L << 31);
}

template <typename T>
static int conv_mfma_tconv(const coma_conv_desc* d, const coma_tensor* x, const void* wk, const float* bias, const coma_tensor* y,
                           hipStream_t s, double2* stats, int stats_inst, int* stats_chunks, int accum) {
  COMA_CHECK(aligned16(wk), "conv_mfma_tconv: weights must be 16-byte aligned");
  TconvP q;
  q.x = x->data; q.ldx = (int)x->ld; q.sbx = x->sb; q.D = x->D; q.H = x->H; q.W = x->W; q.C = x->C;
  q.y = y->data; q.ldy = (int)y->ld; q.sby = y->sb; q.Do = y->D; q.Ho = y->H; q.Wo = y->W; q.N = y->C;
  q.w = wk; q.wsb = d->per_sample_w ? 27L * y->C * x->C : 0;
  q.bias = bias; q.bsb = d->per_sample_w ? y->C : 0;
  q.xbytes = (unsigned)((unsigned long long)t_vox(x) * x->ld * sizeof(T));
  q.wbytes = (unsigned)(27ull * y->C * x->C * sizeof(T));
  q.accum = accum;
  q.ntx = (q.W + 31) / 32; q.nty = (q.H + 3) / 4; q.ntz = (q.D + 1) / 2;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  const int nblk_n = y->C / 32;
  int gx = 512 / (nblk_n * x->B);                      // one block per CU, about two rounds
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  q.stats = nullptr; q.stats_inst = stats_inst;
  static const bool fuse_stats = getenv("COMA_TCONV_NO_STATS") == nullptr;      // (A/B: statistics as a separate pass)
  if (stats && !accum && fuse_stats) { q.stats = stats; *stats_chunks = 1; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_tconv_k<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_tconv_k<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_tconv_k<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_tconv_k<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const size_t lds = (size_t)(33 * 5 * 3 + 14 * 32) * 80;
  coma_set_kernel_tag("conv_mfma_tconv_k<%s, %d>", sizeof(T) == 2 ? "__bf16" : "float", q.stats ? 1 : 0);
  const dim3 grid((unsigned)gx, (unsigned)nblk_n, (unsigned)x->B);
  if (q.stats) hipLaunchKernelGGL((conv_mfma_tconv_k<T, true>), grid, dim3(256), lds, s, q);
  else hipLaunchKernelGGL((conv_mfma_tconv_k<T, false>), grid, dim3(256), lds, s, q);
  COMA_LAUNCH_CHECK();
  return 0;
}

bool conv_mfma_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (x->dtype != COMA_BF16 || y->dtype != COMA_BF16) return false;
  if ((long)t_vox(x) * x->ld >= (1L << 31) || (long)t_vox(y) * y->ld >= (1L << 31)) return false;
  if (halo_ok(d, x, y)) return true;
  if (pw_ok(d, x, y)) return true;
  if (x->C % 32 || y->C % 32) return false;
  if (x->ld % 8 || y->ld % 8 || x->sb % 8 || y->sb % 8) return false;
  if (x->data && !aligned16(x->data)) return false;
  if ((long)t_vox(x) * x->ld >= (1L << 31) || (long)t_vox(y) * y->ld >= (1L << 31)) return false;
  if (d->form == 1 && d->stride == 2 && d->ksize != 3) return false;
  return true;
}

// ---- fp32 mode (v_mfma_f32_32x32x2_f32): which problems the MFMA kernels take; the rest stays on conv_direct ----
static bool f32_halo_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (x->W < 16 && x->C % 32 == 0 && y->C % 32 == 0) return false;       // 8^3 grids: gather + split-K (as bf16)
  return d->ksize == 3 && d->stride == 1 && x->W >= 8 && (long)x->H * x->W >= 32 && x->C % 16 == 0 && y->C >= 16 &&
         x->ld % 4 == 0 && x->sb % 4 == 0 && (!x->data || aligned16(x->data)) && (unsigned long long)t_vox(x) * x->ld * 4 < 0x7fff0000ull;
}
bool conv_f32mfma_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (x->dtype != COMA_F32 || y->dtype != COMA_F32) return false;
  if ((long)t_vox(x) * x->ld >= (1L << 31) || (long)t_vox(y) * y->ld >= (1L << 31)) return false;
  if (thin16f_ok(d, x, y)) return true;
  if (f32_halo_ok(d, x, y)) return true;
  if (x->C % 16 || y->C % 32) return false;
  if (x->ld % 4 || y->ld % 4 || x->sb % 4 || y->sb % 4) return false;
  if (x->data && !aligned16(x->data)) return false;
  if (d->form == 1 && d->stride == 2 && d->ksize != 3) return false;
  return true;
}

// split-K factor for a gather launch of `blocks` tiles and `nsteps` K steps (1 = no split)
static int gather_ksplit(long blocks, int nsteps, bool f32) {
  // fp32 MFMA: a K step is 16x the MFMA time of a bf16 one, so the global-load latency per step is covered with one
  // block per CU; split only to fill the chip
  if (f32) {
    if (blocks >= 192 || nsteps < 8) return 1;
    long ks = 256 / blocks;
    if (ks > 8) ks = 8;
    if (ks > nsteps / 4) ks = nsteps / 4;
    return ks < 2 ? 1 : (int)ks;
  }
  // a block's K loop exposes one global-load latency per 32-channel step (8 MFMAs per wave): with one block per CU a
  // deep layer is latency-bound, so the split aims at ~4 resident blocks per CU, not merely at filling the CUs
  // (measured: more slices than this lose to their own atomic traffic -- every slice adds a full fp32 tile)
  if (blocks >= 384 || nsteps < 16) return 1;
  long ks = 512 / blocks;
  if (ks > 8) ks = 8;
  if (ks > nsteps / 4) ks = nsteps / 4;
  return ks < 2 ? 1 : (int)ks;
}

template <int BN, typename T>
static int launch_gather(const GatherP& p0, int mode, int B, hipStream_t s, void* ws, size_t ws_bytes, int ws_zeroed,
                         double2* stats, int stats_inst, int* stats_chunks) {
  GatherP p = p0;
  constexpr int BM = (BN == 128) ? 128 : 256;
  constexpr int LCK = elem<T>::EPB == 8 ? 5 : 4;
  const long Mtot = (long)p.Mz * p.My * p.Mx;
  const size_t lds = (size_t)2 * (BM + BN) * 64 + BM * 4;
  const unsigned gx = (unsigned)((Mtot + BM - 1) / BM), gy = (unsigned)(p.N / BN) * (mode == 1 ? 8 : 1);
  const long Vout = (long)p.Do * p.Ho * p.Wo;
  const int ntaps_max = mode == 1 ? 8 : p.k * p.k * p.k;      // (mode 1: the 8-tap parity class bounds the split)
  p.ksplit = gather_ksplit((long)gx * gy * B, (mode == 1 ? 1 : ntaps_max) * (p.C >> LCK), LCK == 4);
  p.part = nullptr; p.part_sb = 0;
  if (p.ksplit > 1 && ws && ws_bytes >= sizeof(float) * (size_t)B * Vout * p.N) {
    p.part = (float*)ws; p.part_sb = Vout * p.N;
    if (!ws_zeroed && hipMemsetAsync(ws, 0, sizeof(float) * (size_t)B * Vout * p.N, s) != hipSuccess) { coma_set_error("conv split-K memset failed"); return 2; }
  } else p.ksplit = 1;
  p.stats = nullptr; p.stats_inst = stats_inst;
  if (stats && p.ksplit == 1 && !p.accum) { p.stats = stats; *stats_chunks = 1; }     // (split-K: the values are not final in this kernel)
  dim3 grid(gx, gy * p.ksplit, (unsigned)B);
  coma_set_kernel_tag("conv_mfma_gather_k<%d, %d, %s>", BN, mode, LCK == 4 ? "float" : "__bf16");
  if (mode == 0) hipLaunchKernelGGL((conv_mfma_gather_k<BN, 0, T>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((conv_mfma_gather_k<BN, 1, T>), grid, dim3(256), lds, s, p);
  COMA_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    long nb = (Vout * p.N + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(gather_finalize_k<T>, dim3((unsigned)nb, (unsigned)B), dim3(256), 0, s, (const float*)p.part, p.part_sb, (T*)p.y, p.ldy,
                       p.sby, p.N, Vout, p.bias, p.bsb, p.accum);
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

template <int CK, int LX, int VEC, typename T>
static int launch_halo(const HaloP& p0, int B, hipStream_t s) {
  HaloP p = p0;
  constexpr int TX = 1 << LX, RY = 32 / TX, TY = 4 * RY, TZ = 2;
  constexpr int HV = (TX + 2) * (TY + 2) * (TZ + 2);
  p.ntx = (p.W + TX - 1) / TX; p.nty = (p.H + TY - 1) / TY; p.ntz = (p.D + TZ - 1) / TZ;
  const size_t lds = (size_t)(HV + 9 * 32) * CK * sizeof(T);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)conv_mfma_halo_k<CK, LX, VEC, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
  dim3 grid((unsigned)(p.ntx * p.nty * ((p.ntz + 7) / 8) * 8), (unsigned)((p.N + 31) / 32), (unsigned)B);
  coma_set_kernel_tag("conv_mfma_halo_k<%d, %d, %d, %s>", CK, LX, VEC, sizeof(T) == 4 ? "float" : "__bf16");
  hipLaunchKernelGGL((conv_mfma_halo_k<CK, LX, VEC, T>), grid, dim3(256), lds, s, p);
  COMA_LAUNCH_CHECK();
  return 0;
}

// bytes of scratch conv_mfma_duo_k wants for a fragment-ordered copy of the weights (C >= 64 stride-1 3^3 bf16 layers), else 0
static bool aligned16(const void* p);
static size_t duo_frag_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  static const bool on = getenv("COMA_DUO_C32_ONLY") == nullptr && getenv("COMA_NO_DUO") == nullptr;
  if (!on || x->dtype != COMA_BF16 || d->ksize != 3 || d->stride != 1 || x->W < 16) return 0;
  if (x->C < 64 || x->C % 16 || y->C % 32) return 0;
  return (size_t)(d->per_sample_w ? x->B : 1) * 27 * y->C * x->C * 2;
}

// stats != NULL requests fused {sum, sumsq} partials; *stats_chunks receives the number of chunks written, or stays
// 0 when the selected kernel variant cannot fuse them (the caller then runs the stand-alone statistics pass).
template <typename T>
static int conv_mfma_halo(const coma_conv_desc* d, const coma_tensor* x, const void* wk, const float* bias,
                          const coma_tensor* y, hipStream_t s, double2* stats = nullptr, int stats_inst = 0,
                          int* stats_chunks = nullptr, void* ws = nullptr, size_t ws_bytes = 0) {
  constexpr int EPB = elem<T>::EPB;
  constexpr bool F32 = EPB == 4;
  HaloP p;
  p.x = x->data; p.ldx = (int)x->ld; p.sbx = x->sb; p.D = x->D; p.H = x->H; p.W = x->W; p.C = x->C;
  p.y = y->data; p.ldy = (int)y->ld; p.sby = y->sb; p.N = y->C;
  p.w = wk; p.wsb = d->per_sample_w ? 27L * y->C * x->C : 0;
  p.bias = bias; p.bsb = d->per_sample_w ? y->C : 0;
  p.flip = d->form == 1;
  p.vecx = x->ld % EPB == 0 && x->sb % EPB == 0 && aligned16(x->data);
  p.vecw = x->C % EPB == 0 && aligned16(wk);
  const bool thin = !F32 && x->C <= 16;
  const int lx = x->W >= 32 ? 5 : (x->W >= 16 ? 4 : 3);
  // (vector path: buffer loads with 32-bit byte offsets below the out-of-range marker 0x7fff0000)
  const bool vec = p.vecx && p.vecw && x->C % (4 * EPB) == 0 && (unsigned long long)t_vox(x) * x->ld * sizeof(T) < 0x7fff0000ull;
  if (F32) COMA_CHECK(vec, "conv_mfma(fp32): operands must allow 16-byte channel pieces (C %% 16 == 0, aligned)");
  if constexpr (!F32) {
    static const bool thin16_on = []{ const char* e = getenv("COMA_THIN16"); return !(e && e[0] == '0'); }();
    const unsigned long long xb_ = (unsigned long long)t_vox(x) * x->ld * sizeof(T);
    if (thin16_on && thin && lx == 5 && y->C <= (x->C > 8 ? 16 : 32) && p.vecx && xb_ < 0x7fff0000ull) {
      Thin16P q;
      q.x = (const bf16_t*)p.x; q.ldx = p.ldx; q.sbx = p.sbx; q.D = p.D; q.H = p.H; q.W = p.W; q.C = p.C;
      q.y = (bf16_t*)p.y; q.ldy = p.ldy; q.sby = p.sby; q.N = p.N; q.w = (const bf16_t*)p.w; q.wsb = p.wsb; q.bias = p.bias; q.bsb = p.bsb;
      q.flip = p.flip; q.xbytes = (unsigned)xb_;
      q.st8 = y->ld % 4 == 0 && y->sb % 4 == 0 && (((uintptr_t)y->data) & 7) == 0;
      q.ntx = (q.W + 31) / 32; q.nty = (q.H + 3) / 4; q.ntz = (q.D + 1) / 2;
      q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
      int gx = 1024 / x->B;                              // >= 2 blocks per CU
      if (gx < 1) gx = 1;
      if (gx > q.ids_total) gx = q.ids_total;
      q.ids_per_block = (q.ids_total + gx - 1) / gx;
      gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
      q.stats = nullptr; q.stats_inst = stats_inst;
      if (stats) { q.stats = stats; *stats_chunks = 1; }
      dim3 grid((unsigned)gx, 1, (unsigned)x->B);
      const bool c16 = q.C > 8, n32 = q.N > 16;
      size_t lds = (size_t)34 * 6 * 4 * (c16 ? 32 : 16);
      if (lds < (size_t)27 * q.N * q.C * 2) lds = (size_t)27 * q.N * q.C * 2;     // the weights are staged there first (<= 13.8 KB)
      coma_set_kernel_tag("conv_thin16_k<%d, %d>", c16 ? 16 : 8, n32 ? 2 : 1);
      if (c16) hipLaunchKernelGGL((conv_thin16_k<16, 1>), grid, dim3(256), lds, s, q);
      else     { if (n32) hipLaunchKernelGGL((conv_thin16_k<8, 2>), grid, dim3(256), lds, s, q);  else hipLaunchKernelGGL((conv_thin16_k<8, 1>), grid, dim3(256), lds, s, q); }
      COMA_LAUNCH_CHECK();
      return 0;
    }
  }
  if ((lx == 5 && (thin || vec)) || (lx == 4 && !F32 && !thin && vec)) {      // (lx == 4: the two-group kernel only)
    Halo2P q;
    q.x = p.x; q.ldx = p.ldx; q.sbx = p.sbx; q.D = p.D; q.H = p.H; q.W = p.W; q.C = p.C;
    q.y = p.y; q.ldy = p.ldy; q.sby = p.sby; q.N = p.N; q.w = p.w; q.wsb = p.wsb; q.bias = p.bias; q.bsb = p.bsb;
    q.flip = p.flip; q.vecx = p.vecx; q.vecw = p.vecw;
    {   // descriptor ranges (one sample / one weight set), capped below the out-of-range marker 0x7fff0000
      const unsigned long long xb_ = (unsigned long long)t_vox(x) * x->ld * sizeof(T), wb_ = 27ull * y->C * x->C * sizeof(T);
      q.xbytes = (unsigned)(xb_ < 0x7fff0000ull ? xb_ : 0x7fff0000ull);
      q.wbytes = (unsigned)(wb_ < 0x7fff0000ull ? wb_ : 0x7fff0000ull);
    }
    q.stats = nullptr; q.stats_inst = stats_inst;
    q.st8 = y->ld % 4 == 0 && y->sb % 4 == 0 && (((uintptr_t)y->data) & 7) == 0;
    q.st16 = y->ld % EPB == 0 && y->sb % EPB == 0 && (((uintptr_t)y->data) & 15) == 0;
    const int txv = lx == 5 ? 32 : 16, tyv = lx == 5 ? 4 : 8;      // tile: 2 x 4 x 32 voxels, or 2 x 8 x 16 on the 16-wide grids
    q.ntx = (q.W + txv - 1) / txv; q.nty = (q.H + tyv - 1) / tyv; q.ntz = (q.D + 1) / 2;
    q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
    const int nblk_n = (q.N + 31) / 32;
    int gx = (thin ? 1024 : 512) / (nblk_n * x->B);   // thin: 2 blocks per CU, thick: 1
    if (F32) gx = 256 / (nblk_n * x->B);               // fp32 tiles are 16x longer: one round of one block per CU balances best
    if (gx < 1) gx = 1;
    if (gx > q.ids_total) gx = q.ids_total;
    q.ids_per_block = (q.ids_total + gx - 1) / gx;
    gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
    dim3 grid((unsigned)gx, (unsigned)nblk_n, (unsigned)x->B);
    if (stats) { q.stats = stats; *stats_chunks = 1; }
    if constexpr (!F32) {
      // two 4-wave groups per CU alternating matrix and staging phases (conv_mfma_duo_k): full 32-channel output tiles with
      // 16-byte stores, 16-channel input chunks
      // Measured (128^3, batch 2; microbenchmarks, duo / halo2): 32 -> 32 forward 282-315 / 321-339 us.  C >= 64 from the plain
      // weight layout was SLOWER (64 -> 32 732-757 / 683-704 us: a 16-channel weight piece is 32 bytes of a 128-byte row and the
      // staging group pulled 4x the useful bytes through L1); with the weights re-laid in fragment order by a 4-us launch
      // (duo_relayout_k into the caller's scratch, coma_conv_fwd_ws_bytes): 64 -> 32 696 / 691 and 665 / 689 (data gradient),
      // 64 -> 64 at 64^3 129 / 144 and 134 / 150, 128 -> 64 270 / 273 and 277 / 297, 256 -> 128 at 32^3 117 / 125 and 126 / 134;
      // step 18.09 -> 17.99 ms.  COMA_NO_DUO=1: conv_mfma_halo2_k everywhere; COMA_DUO_C32_ONLY=1: the C == 32 layers only.
      static const bool no_duo = getenv("COMA_NO_DUO") != nullptr, duo_all = getenv("COMA_DUO_C32_ONLY") == nullptr;
      q.wfrag = nullptr; q.wfrag_sb = 0;
      const size_t frag_bytes = duo_frag_bytes(d, x, y);      // C >= 64: scratch for the fragment-ordered copy of the weights
      const bool wide_ok = duo_all && q.C > 32 && frag_bytes > 0 && ws && ws_bytes >= frag_bytes && aligned16(ws);
      if (!no_duo && !thin && vec && (q.C == 32 || wide_ok) && q.N % 32 == 0 && q.st16 &&
          q.D <= 510 && q.H <= 510 && q.W <= 510) {
        if (q.C > 32) {
          const long pieces = (long)(frag_bytes / 16);
          hipLaunchKernelGGL(duo_relayout_k, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, (const uint4*)wk, (uint4*)ws, q.N, q.C / 8, pieces);
          COMA_LAUNCH_CHECK();
          q.wfrag = ws; q.wfrag_sb = d->per_sample_w ? 27L * q.N * q.C : 0;
        }
        int g2 = 256 / (nblk_n * x->B);                   // one 512-thread block per CU, once
        if (g2 < 1) g2 = 1;
        if (g2 > q.ids_total) g2 = q.ids_total;
        q.ids_per_block = (q.ids_total + g2 - 1) / g2;
        g2 = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
        const dim3 grid2((unsigned)g2, (unsigned)nblk_n, (unsigned)x->B);
        if (stats) { q.stats = stats; *stats_chunks = 1; }
        const size_t lds2 = (size_t)2 * 34 * 6 * 4 * 48 + (size_t)2 * 27 * 64 * 16 + 128;
        static bool attr2 = false;
        if (!attr2) {
          (void)hipFuncSetAttribute((const void*)conv_mfma_duo_k<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute((const void*)conv_mfma_duo_k<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute((const void*)conv_mfma_duo_k<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute((const void*)conv_mfma_duo_k<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr2 = true;
        }
        if (lx == 5) {
          coma_set_kernel_tag("conv_mfma_duo_k<%d, 5>", stats ? 1 : 0);
          if (stats) hipLaunchKernelGGL((conv_mfma_duo_k<1, 5>), grid2, dim3(512), lds2, s, q);
          else hipLaunchKernelGGL((conv_mfma_duo_k<0, 5>), grid2, dim3(512), lds2, s, q);
        } else {
          coma_set_kernel_tag("conv_mfma_duo_k<%d, 4>", stats ? 1 : 0);
          if (stats) hipLaunchKernelGGL((conv_mfma_duo_k<1, 4>), grid2, dim3(512), lds2, s, q);
          else hipLaunchKernelGGL((conv_mfma_duo_k<0, 4>), grid2, dim3(512), lds2, s, q);
        }
        COMA_LAUNCH_CHECK();
        return 0;
      }
    }
    if (lx == 5) {
    constexpr int HV2 = 34 * 6 * 4;
    const bool resident = thin || (!F32 && q.C == 32);
    // C >= 64: all 27 taps of the current 32-channel chunk in LDS, refetched per chunk (RESIDENT = 2).  The 9-taps-per-
    // kz-plane streaming mode (RESIDENT = 0) it replaces left one plane of MFMAs (~0.5 us) to cover each weight fetch and
    // cost 14 barriers per tile: 64 -> 32 at 128^3 940 -> 832 us, 128 -> 64 at 64^3 384 -> 328 us, 256 -> 128 at 32^3 186 -> 157 us.
    const size_t lds = thin ? (size_t)(HV2 + 27 * 32) * 48 : (size_t)HV2 * 80 + (size_t)27 * 32 * 80;
    static bool attr = false;
    if (!attr) {
      if constexpr (F32) {
        (void)hipFuncSetAttribute((const void*)conv_mfma_halo2_k<2, 16, 1, 1, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      } else {
        (void)hipFuncSetAttribute((const void*)conv_mfma_halo2_k<1, 32, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_mfma_halo2_k<2, 32, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_mfma_halo2_k<1, 16, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      }
      attr = true;
    }
    coma_set_kernel_tag(F32 ? "conv_mfma_halo2_k<2, 16, 1, 1, float>" : thin ? "conv_mfma_halo2_k<1, 16, 0, 2, __bf16>"
                        : resident ? "conv_mfma_halo2_k<1, 32, 1, 1, __bf16>" : "conv_mfma_halo2_k<2, 32, 1, 1, __bf16>");
    if constexpr (F32) {
      hipLaunchKernelGGL((conv_mfma_halo2_k<2, 16, 1, 1, float>), grid, dim3(256), lds, s, q);
    } else {
      if (thin) hipLaunchKernelGGL((conv_mfma_halo2_k<1, 16, 0, 2>), grid, dim3(256), lds, s, q);
      else if (resident) hipLaunchKernelGGL((conv_mfma_halo2_k<1, 32, 1, 1>), grid, dim3(256), lds, s, q);
      else hipLaunchKernelGGL((conv_mfma_halo2_k<2, 32, 1, 1>), grid, dim3(256), lds, s, q);
    }
    COMA_LAUNCH_CHECK();
    return 0;
    }
    if (stats_chunks) *stats_chunks = 0;      // (lx == 4 and not taken by the two-group kernel: the first-generation kernel below, no fused statistics)
  }
  if constexpr (F32) {
    if (lx == 5) return launch_halo<16, 5, 1, float>(p, x->B, s);
    if (lx == 4) return launch_halo<16, 4, 1, float>(p, x->B, s);
    return launch_halo<16, 3, 1, float>(p, x->B, s);
  } else {
    if (thin) {
      if (lx == 5) return launch_halo<16, 5, 0, bf16_t>(p, x->B, s);
      if (lx == 4) return launch_halo<16, 4, 0, bf16_t>(p, x->B, s);
      return launch_halo<16, 3, 0, bf16_t>(p, x->B, s);
    }
    if (vec) {
      if (lx == 5) return launch_halo<32, 5, 1, bf16_t>(p, x->B, s);
      if (lx == 4) return launch_halo<32, 4, 1, bf16_t>(p, x->B, s);
      return launch_halo<32, 3, 1, bf16_t>(p, x->B, s);
    }
    if (lx == 5) return launch_halo<32, 5, 0, bf16_t>(p, x->B, s);
    if (lx == 4) return launch_halo<32, 4, 0, bf16_t>(p, x->B, s);
    return launch_halo<32, 3, 0, bf16_t>(p, x->B, s);
  }
}

// bytes of workspace with which the deep layers' K loop may be split over more blocks (0: never split)
// y += conv(x) is implemented by the gather and pointwise kernels (the data gradients that land on an activation other
// consumers also feed: strided / transposed / 1x1x1 layers), not by the halo-tiled stride-1 kernels
bool conv_mfma_accumulate_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (x->dtype == COMA_F32) return conv_f32mfma_supported(d, x, y) && !thin16f_ok(d, x, y) && !f32_halo_ok(d, x, y);
  return conv_mfma_supported(d, x, y) && !halo_ok(d, x, y);
}

static int gather_ksplit(long blocks, int nsteps, bool f32);
size_t conv_mfma_fwd_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* y) {
  if (x->dtype == COMA_F32) { if (thin16f_ok(d, x, y) || f32_halo_ok(d, x, y)) return 0; }
  else if (halo_ok(d, x, y)) return duo_frag_bytes(d, x, y);
  else if (pw_ok(d, x, y)) return 0;
  if (tconv_ok(d, x, y)) return 0;
  const size_t bytes = sizeof(float) * (size_t)y->B * t_vox(y) * y->C;
  if (bytes > ((size_t)64 << 20)) return 0;
  // only when launch_gather will really split the K loop (the same arithmetic as there): callers that hand out private
  // pre-zeroed scratch should not be asked for megabytes a launch never touches
  const bool f32 = x->dtype == COMA_F32;
  const int BN = y->C % 128 == 0 ? 128 : y->C % 64 == 0 ? 64 : 32, BM = BN == 128 ? 128 : 256, LCK = f32 ? 4 : 5;
  const int mode = d->form == 1 && d->stride == 2;
  const long Mtot = mode ? (long)((y->D + 1) / 2) * ((y->H + 1) / 2) * ((y->W + 1) / 2) : (long)t_vox(y);
  const long gx = (Mtot + BM - 1) / BM, gy = (long)(y->C / BN) * (mode ? 8 : 1);
  const int taps = d->ksize * d->ksize * d->ksize;
  return gather_ksplit(gx * gy * y->B, (mode ? 1 : taps) * (x->C >> LCK), f32) > 1 ? bytes : 0;
}

int conv_mfma_fwd(const coma_conv_desc* d, const coma_tensor* x, const void* wk, const float* bias,
                  const coma_tensor* y, hipStream_t s, double2* stats, int stats_inst, int* stats_chunks, void* ws,
                  size_t ws_bytes, int ws_zeroed, int accum) {
  const bool f32 = x->dtype == COMA_F32;
  COMA_CHECK(!accum || conv_mfma_accumulate_ok(d, x, y), "conv_mfma: this problem's kernel family cannot accumulate into y");
  if (f32 && thin16f_ok(d, x, y)) return conv_thin16f(d, x, wk, bias, y, s, stats, stats_inst, stats_chunks);
  if (f32) { if (f32_halo_ok(d, x, y)) return conv_mfma_halo<float>(d, x, wk, bias, y, s, stats, stats_inst, stats_chunks); }
  else if (halo_ok(d, x, y)) return conv_mfma_halo<bf16_t>(d, x, wk, bias, y, s, stats, stats_inst, stats_chunks, ws, ws_bytes);
  if (tconv_ok(d, x, y)) {
    if (f32) return conv_mfma_tconv<float>(d, x, wk, bias, y, s, stats, stats_inst, stats_chunks, accum);
    return conv_mfma_tconv<bf16_t>(d, x, wk, bias, y, s, stats, stats_inst, stats_chunks, accum);
  }
  if (!f32 && pw_ok(d, x, y) && !(x->C % 32 == 0 && y->C % 32 == 0 && x->C * y->C > 64 * 32)) {
    COMA_CHECK(aligned16(wk), "conv_mfma: weights must be 16-byte aligned");
    PwP q;
    q.x = (const bf16_t*)x->data; q.ldx = (int)x->ld; q.sbx = x->sb; q.V = t_vox(x); q.C = x->C;
    q.y = (bf16_t*)y->data; q.ldy = (int)y->ld; q.sby = y->sb; q.N = y->C;
    q.w = (const bf16_t*)wk; q.wsb = d->per_sample_w ? (long)y->C * x->C : 0;
    q.bias = bias; q.bsb = d->per_sample_w ? y->C : 0;
    q.st8 = y->ld % 4 == 0 && y->sb % 4 == 0 && (((uintptr_t)y->data) & 7) == 0;
    q.st16 = y->ld % 8 == 0 && y->sb % 8 == 0 && (((uintptr_t)y->data) & 15) == 0;
    const int ks = (x->C + 15) / 16, nt = (y->C + 31) / 32;
    long nb = ((q.V + 31) / 32 + 3) / 4;
    if (nb > 2048) nb = 2048;
    q.stats = nullptr; q.stats_inst = stats_inst; q.accum = accum;
    if (stats) { q.stats = stats; *stats_chunks = 1; }
    dim3 grid((unsigned)nb, (unsigned)x->B);
    coma_set_kernel_tag("conv_mfma_pw_k<%d, %d>", ks > 4 ? 4 : ks, nt);
#define PWL(K_, N_) hipLaunchKernelGGL((conv_mfma_pw_k<K_, N_>), grid, dim3(256), 0, s, q)
    if (nt == 1) { if (ks == 1) PWL(1, 1); else if (ks == 2) PWL(2, 1); else if (ks == 3) PWL(3, 1); else PWL(4, 1); }
    else { if (ks == 1) PWL(1, 2); else if (ks == 2) PWL(2, 2); else if (ks == 3) PWL(3, 2); else PWL(4, 2); }
#undef PWL
    COMA_LAUNCH_CHECK();
    return 0;
  }
  COMA_CHECK(aligned16(wk) && aligned16(x->data), "conv_mfma: operands must be 16-byte aligned");
  GatherP p;
  p.x = x->data; p.ldx = (int)x->ld; p.sbx = x->sb; p.Di = x->D; p.Hi = x->H; p.Wi = x->W; p.C = x->C;
  p.y = y->data; p.ldy = (int)y->ld; p.sby = y->sb; p.Do = y->D; p.Ho = y->H; p.Wo = y->W; p.N = y->C;
  const int taps = d->ksize * d->ksize * d->ksize;
  p.w = wk; p.wsb = d->per_sample_w ? (long)taps * y->C * x->C : 0;
  p.bias = bias; p.bsb = d->per_sample_w ? y->C : 0;
  p.k = d->ksize; p.stride = d->stride; p.flip = 0; p.accum = accum;
  int mode = 0;
  if (d->form == 1 && d->stride == 1) { p.flip = 1; }
  if (d->form == 1 && d->stride == 2) {
    mode = 1;
    p.Mz = (y->D + 1) / 2; p.My = (y->H + 1) / 2; p.Mx = (y->W + 1) / 2;
  } else {
    p.Mz = y->D; p.My = y->H; p.Mx = y->W;
  }
  if (f32) {
    if (y->C % 128 == 0) return launch_gather<128, float>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
    if (y->C % 64 == 0) return launch_gather<64, float>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
    return launch_gather<32, float>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
  }
  if (y->C % 128 == 0) return launch_gather<128, bf16_t>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
  if (y->C % 64 == 0) return launch_gather<64, bf16_t>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
  return launch_gather<32, bf16_t>(p, mode, x->B, s, ws, ws_bytes, ws_zeroed, stats, stats_inst, stats_chunks);
}

// =====================================================================================
// Weight gradient on MFMA:  dwk[b][tap][n][c] = sum_m dy[.][n] * x[.][c]  over the voxels m of
// the M-grid, one operand read densely and the other through a tap-shifted (strided) gather:
//   FORM 0 (nn.Conv3d):          dense = dy[m],  gathered = x[m*stride - pad + tap]
//   FORM 1 (nn.ConvTranspose3d): dense = x[m],   gathered = dy[m*2 - pad + tap]
// A block stages one tile of dense voxels and the matching halo of the gathered tensor in LDS
// as [voxel][channel] images, then every tap re-reads the SAME halo at its shifted address: the
// 27-fold operand reuse happens in LDS, not in L2/HBM.  The GEMM reduction index is the voxel,
// so both MFMA operands are read with ds_read_b64_tr_b16 (4 voxels x 16 channels per 16-lane
// group, delivered channel-per-lane).  The 27 taps are dealt to the 4 waves (7,7,7,6); each wave
// keeps its taps' 32x32 fp32 accumulators in registers across all tiles of the block
// (weight-gradient-stationary) and merges them into dwk with fp32 atomics at the end.
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

struct WgradP2 {
  const void* dyp; int ldn; long sbn;     // dy  (N channels)
  const void* xp;  int ldc; long sbc;     // x   (C channels)
  int N, C;
  int Mz, My, Mx;      // dense grid
  int Gz, Gy, Gx;      // gathered grid
  int k, stride, pad;
  int lx, ly, lz;      // log2 of the tile dims (tile = 2^lz x 2^ly x 2^lx dense voxels)
  int hz, hy, hx;      // halo dims
  int ntx, nty, ntz;   // tiles per dim
  int tiles_total, tiles_per_block;
  float* dwk; long wsb;
  int cblocks;         // ceil(C / (32*TC))
  int vec_n, vec_c;    // 16-byte loads legal on dy / x
  unsigned m_hx, m_hxy; // magic multipliers: n / hx == umulhi(n, m_hx), n / (hx*hy) == umulhi(n, m_hxy)
  int plain;           // every dwk element is produced by exactly one block: plain stores, no memset, no atomics
  int cp, pg;          // fp32 kernel: gathered channels per tap in an MFMA tile (power of two <= 32), gathered LDS row pitch (bytes)
  int nrep; long rep_stride;   // fp32 kernel, small outputs: blocks merge into one of nrep replicas (summed afterwards)
};

template <int TN, int TC, int FORM, int VEC>
__global__ __launch_bounds__(256, 1) void conv_mfma_wgrad_k(WgradP2 p) {
  constexpr int CDB = 32 * (FORM == 0 ? TN : TC);   // dense-side channels per block
  constexpr int CGB = 32 * (FORM == 0 ? TC : TN);   // gathered-side channels per block
  constexpr int PD = CDB * 2, PG = CGB * 2;         // LDS row pitches (bytes)
  constexpr int MAXT = 7;                           // taps per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int TM = 1 << (p.lx + p.ly + p.lz);
  char* Dt = smem;
  char* Gt = smem + TM * PD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tap lists live in SGPRs, tap tests are scalar branches
  const int b = blockIdx.z;
  const int nb = blockIdx.y / p.cblocks, cb = blockIdx.y % p.cblocks;
  const int n0 = nb * 32 * TN, c0 = cb * 32 * TC;
  const bf16_t* dyp16 = static_cast<const bf16_t*>(p.dyp);
  const bf16_t* xp16 = static_cast<const bf16_t*>(p.xp);
  const bf16_t* dense = (FORM == 0 ? dyp16 + (long)b * p.sbn + n0 : xp16 + (long)b * p.sbc + c0);
  const bf16_t* gath = (FORM == 0 ? xp16 + (long)b * p.sbc + c0 : dyp16 + (long)b * p.sbn + n0);
  const int ldd = FORM == 0 ? p.ldn : p.ldc, ldg = FORM == 0 ? p.ldc : p.ldn;
  const int chd = (FORM == 0 ? p.N - n0 : p.C - c0), chg = (FORM == 0 ? p.C - c0 : p.N - n0);   // channels left
  const bool vecd = FORM == 0 ? p.vec_n : p.vec_c, vecg = FORM == 0 ? p.vec_c : p.vec_n;
  const int ntaps = p.k * p.k * p.k;
  const int HV = p.hz * p.hy * p.hx;
  const int tx = 1 << p.lx, ty = 1 << p.ly;

  f32x16_t acc[MAXT][TN][TC];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

  // per-wave tap list (wave-uniform; hoisted out of every loop: no integer division inside)
  int tap_w[MAXT], toff_w[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tap = p.k == 1 ? (t == 0 ? 0 : ntaps) : wid + 4 * t;
    tap_w[t] = tap;
    const int kx = tap % p.k, ky = (tap / p.k) % p.k, kz = tap / (p.k * p.k);
    toff_w[t] = __builtin_amdgcn_readfirstlane(((kz * p.hy + ky) * p.hx + kx) * PG);
  }
  // lane roles for the transposed reads
  const int g16 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int chan_b = ((g16 & 1) * 16 + 4 * pp) * 2;   // byte offset of this lane's 4 channels in a 32-channel block
  const int vrow = 8 * (g16 >> 1) + q;                // voxel (within a 16-voxel K step) whose row this lane addresses

  const int tile_begin = xcd_remap(blockIdx.x, gridDim.x) * p.tiles_per_block;
  int tile_end = tile_begin + p.tiles_per_block;
  if (tile_end > p.tiles_total) tile_end = p.tiles_total;

  // ---- software pipeline: the next tile's global loads are in flight (registers) while this tile computes ----
  constexpr int MAXP = 20;                      // 16-byte pieces per thread (host guarantees pieces <= 256 * MAXP)
  uint4 sv[MAXP];
  const int ndp = TM * (CDB / 8), ngp = HV * (CGB / 8);
  auto piece_dst = [&](int piece) -> int {
    if (piece < ndp) return (piece / (CDB / 8)) * PD + (piece % (CDB / 8)) * 16;
    const int pg = piece - ndp;
    return TM * PD + (pg / (CGB / 8)) * PG + (pg % (CGB / 8)) * 16;
  };
  auto load_tile = [&](int x0, int y0, int z0) {
    const int bz = z0 * p.stride - p.pad, by = y0 * p.stride - p.pad, bx = x0 * p.stride - p.pad;
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      const int piece = tid + 256 * u;
      sv[u] = make_uint4(0, 0, 0, 0);
      if (piece < ndp) {
        const int row = piece / (CDB / 8), ch = piece % (CDB / 8);
        const int vx = row & (tx - 1), vy = (row >> p.lx) & (ty - 1), vz = row >> (p.lx + p.ly);
        const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
        if (gz < p.Mz && gy < p.My && gx < p.Mx && ch * 8 < chd) {
          const bf16_t* src = dense + (unsigned)(((gz * p.My + gy) * p.Mx + gx) * ldd + ch * 8);
          sv[u] = VEC ? *reinterpret_cast<const uint4*>(src) : load8(src, chd - ch * 8, vecd);
        }
      } else if (piece < ndp + ngp) {
        const int pg = piece - ndp;
        const int row = pg / (CGB / 8), ch = pg % (CGB / 8);
        const int q1 = (int)__umulhi((unsigned)row, p.m_hx), hzi = (int)__umulhi((unsigned)row, p.m_hxy);
        const int hxi = row - q1 * p.hx, hyi = q1 - hzi * p.hy;
        const int gz = bz + hzi, gy = by + hyi, gx = bx + hxi;
        if ((unsigned)gz < (unsigned)p.Gz && (unsigned)gy < (unsigned)p.Gy && (unsigned)gx < (unsigned)p.Gx && ch * 8 < chg) {
          const bf16_t* src = gath + (unsigned)(((gz * p.Gy + gy) * p.Gx + gx) * ldg + ch * 8);
          sv[u] = VEC ? *reinterpret_cast<const uint4*>(src) : load8(src, chg - ch * 8, vecg);
        }
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      const int piece = tid + 256 * u;
      if (piece < ndp + ngp) *reinterpret_cast<uint4*>(smem + piece_dst(piece)) = sv[u];
    }
  };

  int tile = tile_begin, tix = 0, tiy = 0, tiz = 0;
  while (tile < tile_end && !tile_coords(tile, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++tile;
  if (tile < tile_end) load_tile(tix << p.lx, tiy << p.ly, tiz << p.lz);
  while (tile < tile_end) {
    int nt = tile + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nt < tile_end && !tile_coords(nt, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nt;
    __syncthreads();   // previous tile's reads are done
    store_tile();
    __syncthreads();
    if (nt < tile_end) load_tile(ntix << p.lx, ntiy << p.ly, ntiz << p.lz);
    // ---- MFMA over the tile's voxels, 16 per K step ----
    const int ksteps = TM >> 4;
    for (int ks = 0; ks < ksteps; ++ks) {
      if (p.k == 1 && (ks & 3) != wid) continue;     // 1x1x1: the 4 waves share the single tap by K step
      const int v1 = ks * 16 + vrow, v2 = v1 + 4;
      const int x1 = v1 & (tx - 1), y1 = (v1 >> p.lx) & (ty - 1), z1 = v1 >> (p.lx + p.ly);
      const int x2 = v2 & (tx - 1), y2 = (v2 >> p.lx) & (ty - 1), z2 = v2 >> (p.lx + p.ly);
      const int d1 = v1 * PD + chan_b, d2 = v2 * PD + chan_b;
      const int g1 = ((z1 * p.stride * p.hy + y1 * p.stride) * p.hx + x1 * p.stride) * PG + chan_b;
      const int g2 = ((z2 * p.stride * p.hy + y2 * p.stride) * p.hx + x2 * p.stride) * PG + chan_b;
      constexpr int TD = FORM == 0 ? TN : TC, TG = FORM == 0 ? TC : TN;
      bf16x8_t df[TD];
#pragma unroll
      for (int i = 0; i < TD; ++i) {
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(Dt + d1 + i * 64));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(Dt + d2 + i * 64));
        df[i] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (tap_w[t] < ntaps) {
          const int toff = toff_w[t];
          bf16x8_t gf[TG];
#pragma unroll
          for (int j = 0; j < TG; ++j) {
            const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(Gt + g1 + toff + j * 64));
            const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(Gt + g2 + toff + j * 64));
            gf[j] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) {
              if (FORM == 0) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[i], gf[j], acc[t][i][j], 0, 0, 0);
              else acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[i], df[j], acc[t][i][j], 0, 0, 0);
            }
        }
      }
    }
    tile = nt; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- merge into dwk[b][tap][n][c] ----
  float* wout = p.dwk + (long)b * p.wsb;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tap = tap_w[t];
    if (tap < ntaps) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int n = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            const int c = c0 + j * 32 + fr;
            if (n < p.N && c < p.C) {
              if (p.plain) wout[((long)tap * p.N + n) * p.C + c] = acc[t][i][j][e];
              else atomicAdd(wout + ((long)tap * p.N + n) * p.C + c, acc[t][i][j][e]);
            }
          }
    }
  }
}

// =====================================================================================
// conv_f32_wgrad_k -- the weight gradient in fp32 mode on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation).  Same block structure as conv_mfma_wgrad_k (one slab of dwk per block, dense tile + gathered halo staged
// once in LDS as [voxel][channel] fp32 rows, accumulators stationary over the block's tiles), but the MFMA reduces over
// TWO voxels per instruction and takes one fp32 per lane and operand: lane (index = lane & 31, voxel = lane >> 5) reads
// its value with a plain ds_read_b32 -- no transposed reads.  An fp32 MFMA occupies the SIMD for 64 cycles, so the LDS
// reads and the address arithmetic of a voxel pair sit in its shadow; the next pair's fragments are read one step ahead.
//
// The 32 indices of the GATHERED operand are (tap, channel) pairs: with >= 32 gathered channels an MFMA tile is one tap
// x 32 channels (27 tiles); with fewer channels (the 1..16-channel layers of the full-resolution tail) CP = next power
// of two >= C channels of 32 / CP taps share a tile (14, 7, 4, 2 or 1 tiles instead of 27) -- the tap offset is just a
// per-lane constant in the gathered read's address.  The tiles are dealt to min(4, tiles) wave groups; with fewer than
// four tiles the remaining waves split the voxel pairs.  NT = tiles per wave is a template parameter: every wave runs
// NT unconditional MFMAs per pair (a wave with fewer real tiles recomputes one into an accumulator that is never
// stored), so the pipeline has no wave-dependent control flow and hipcc keeps its counted lgkmcnt waits.
// =====================================================================================
template <int FORM, int NT>
__global__ __launch_bounds__(256, 1) void conv_f32_wgrad_k(WgradP2 p) {
  constexpr int PD = 128;                            // dense LDS row pitch (bytes): 32 fp32 channels, zero padded
  constexpr int MAXP = NT == 7 ? 21 : 24;            // 16-byte staging pieces per thread (host guarantees the fit; with 7 tiles
                                                     // per wave = 112 accumulator registers 21 is what fits without spilling)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int TM = 1 << (p.lx + p.ly + p.lz);
  const int PG = p.pg;                               // gathered LDS row pitch (bytes): max(CP, 4) channels
  char* Dt = smem;
  char* Gt = smem + TM * PD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int b = blockIdx.z;
  const int nb = blockIdx.y / p.cblocks, cb = blockIdx.y % p.cblocks;
  const int n0 = nb * 32, c0 = cb * 32;
  const float* dyp = reinterpret_cast<const float*>(p.dyp);
  const float* xp = reinterpret_cast<const float*>(p.xp);
  const float* dense = (FORM == 0 ? dyp + (long)b * p.sbn + n0 : xp + (long)b * p.sbc + c0);
  const float* gath = (FORM == 0 ? xp + (long)b * p.sbc + c0 : dyp + (long)b * p.sbn + n0);
  const int ldd = FORM == 0 ? p.ldn : p.ldc, ldg = FORM == 0 ? p.ldc : p.ldn;
  const int chd = (FORM == 0 ? p.N - n0 : p.C - c0), chg = (FORM == 0 ? p.C - c0 : p.N - n0);   // channels left
  const bool vecd = FORM == 0 ? p.vec_n : p.vec_c, vecg = FORM == 0 ? p.vec_c : p.vec_n;
  const int ntaps = p.k * p.k * p.k;
  const int HV = p.hz * p.hy * p.hx;
  const int tx = 1 << p.lx, ty = 1 << p.ly;
  const int CP = p.cp, lcp = 31 - __builtin_clz(CP), TPT = 32 >> lcp;       // channels per tap in a tile, taps per tile
  const int ntiles = (ntaps + TPT - 1) / TPT;
  const int wt = ntiles >= 4 ? 4 : ntiles, ws = 4 / wt;                       // wave groups over tiles x over voxel pairs
  const int tg = wid % wt, ps = wid / wt;

  f32x16_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // this lane's (tap, channel) in each of the wave's tiles -> byte offset of the gathered read; a lane / tile without a
  // real (tap, channel) reads tap 0 (its column or row of the accumulator is never stored)
  int goff[NT];
  const int gch = fr & (CP - 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = tg + wt * t;
    const int tap = i * TPT + (fr >> lcp);
    const bool ok = i < ntiles && tap < ntaps && gch < chg;
    const int tp = ok ? tap : 0;
    const int kx = tp % p.k, ky = (tp / p.k) % p.k, kz = tp / (p.k * p.k);
    goff[t] = ((kz * p.hy + ky) * p.hx + kx) * PG + (ok ? gch : 0) * 4;
    // one tap per tile (>= 32 gathered channels): the tap offset is wave-uniform -> a scalar register, the lane's channel
    // offset goes into the common lane term
    if (NT == 7) goff[t] = __builtin_amdgcn_readfirstlane(((kz * p.hy + ky) * p.hx + kx) * PG);
  }

  const int tile_begin = xcd_remap(blockIdx.x, gridDim.x) * p.tiles_per_block;
  int tile_end = tile_begin + p.tiles_per_block;
  if (tile_end > p.tiles_total) tile_end = p.tiles_total;

  // ---- staging: 16-byte pieces (4 channels), the next tile's loads in flight while this tile computes ----
  uint4 sv[MAXP];
  const int gpr = PG >> 4;                            // pieces per gathered row
  const int ndp = TM * 8, ngp = HV * gpr;
  // A staged piece = 4 channels of which 1..4 exist.  Every load of the tile is issued UNCONDITIONALLY from a clamped
  // address and nothing touches its result before store_tile: a conditional load (or a mask applied right away) makes
  // hipcc copy the value after an s_waitcnt vmcnt(0) -- 24 serialised L2 round trips per tile, half the kernel's time.
  // Validity (inside the volume / inside the tensor's channels) travels as one bit per piece and is applied at the LDS
  // store.  The vector / element-wise choice is made ONCE around the whole unrolled loop for the same reason.
  const int lgpr = 31 - __builtin_clz(gpr);           // (gpr = 1, 2, 4 or 8)
  unsigned long long okbits = 0;
#define F32WG_LOAD_LOOP(LD4)                                                                                            \
  _Pragma("unroll") for (int u = 0; u < MAXP; ++u) {                                                                    \
    const int piece = tid + 256 * u;                                                                                    \
    const bool isd = 256 * u < ndp;      /* block-uniform: ndp = 8 TM is a multiple of 256 -> scalar base pointers */     \
    const int pg = piece - ndp;                                                                                         \
    const int row = isd ? piece >> 3 : pg >> lgpr, ch = isd ? piece & 7 : pg & (gpr - 1);                               \
    const int vx = row & (tx - 1), vy = (row >> p.lx) & (ty - 1), vz = row >> (p.lx + p.ly);                            \
    const int q1 = (int)__umulhi((unsigned)row, p.m_hx), hzi = (int)__umulhi((unsigned)row, p.m_hxy);                  \
    const int gz = isd ? z0 + vz : bz + hzi, gy = isd ? y0 + vy : by + (q1 - hzi * p.hy), gx = isd ? x0 + vx : bx + (row - q1 * p.hx); \
    const int Lz = isd ? p.Mz : p.Gz, Ly = isd ? p.My : p.Gy, Lx = isd ? p.Mx : p.Gx;                                  \
    const bool ok = piece < ndp + ngp && (unsigned)gz < (unsigned)Lz && (unsigned)gy < (unsigned)Ly &&                  \
                    (unsigned)gx < (unsigned)Lx && ch * 4 < (isd ? chd : chg);                                          \
    const unsigned off = ok ? (unsigned)(((gz * Ly + gy) * Lx + gx) * (isd ? ldd : ldg) + ch * 4) : 0u;                 \
    const float* src = (isd ? dense : gath) + off;                                                                      \
    const int nvalid = ok ? (isd ? chd : chg) - ch * 4 : 1; (void)nvalid;                                               \
    okbits |= (unsigned long long)ok << u;                                                                              \
    sv[u] = LD4;                                                                                                        \
    if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0);   /* bound the live address temporaries: 6 loads per group */      \
  }
  // (vector loads only: the host sends operands without 16-byte aligned rows to the direct kernel.  A second, element-wise
  //  copy of this unrolled loop took the 7-tile variant to 78 KB -- past the 64 KB instruction cache two CUs share.)
  auto load_next = [&](int x0, int y0, int z0) __attribute__((always_inline)) {
    const int bz = z0 * p.stride - p.pad, by = y0 * p.stride - p.pad, bx = x0 * p.stride - p.pad;
    okbits = 0;
    F32WG_LOAD_LOOP(*reinterpret_cast<const uint4*>(src))
  };
#undef F32WG_LOAD_LOOP
  const bool partd = (chd & 3) != 0 && chd < 32, partg = (chg & 3) != 0 && chg < 32;     // a piece with 1..3 valid channels exists
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      const int piece = tid + 256 * u;
      if (piece < ndp + ngp) {
        uint4 v = sv[u];
        if (!((okbits >> u) & 1)) v = make_uint4(0, 0, 0, 0);
        if (partd || partg) {       // (block-uniform) channels past the tensor's own: padding or a neighbour's slice -> zero
          const bool isd = 256 * u < ndp;
          const int ch = isd ? (piece & 7) : ((piece - ndp) & (gpr - 1));
          const int nv = (isd ? chd : chg) - ch * 4;
          if (nv < 4) { v.w = 0; if (nv < 3) v.z = 0; if (nv < 2) v.y = 0; if (nv < 1) v.x = 0; }
        }
        reinterpret_cast<uint4*>(smem)[piece] = v;     // dense rows, then halo rows: contiguous
      }
    }
  };

  // one voxel pair (2 q, 2 q + 1: x neighbours of one row) -> the lane's dense value and its tiles' gathered values
  const int lane_d = fh * PD + fr * 4, lane_g = fh * p.stride * PG + (NT == 7 ? gch * 4 : 0);
  auto rd = [&](int q, float& d, float (&g)[NT]) {
    const int v = 2 * q;
    const int x = v & (tx - 1), y = (v >> p.lx) & (ty - 1), z = v >> (p.lx + p.ly);
    d = *reinterpret_cast<const float*>(Dt + v * PD + lane_d);
    const char* gp = Gt + ((z * p.stride * p.hy + y * p.stride) * p.hx + x * p.stride) * PG + lane_g;
#pragma unroll
    for (int t = 0; t < NT; ++t) g[t] = *reinterpret_cast<const float*>(gp + goff[t]);
  };
  auto mm = [&](float d, const float (&g)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (FORM == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, g[t], acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[t], d, acc[t], 0, 0, 0);
    }
  };

  const int npairs = TM >> 1;                               // (a multiple of 8)
  int tile = tile_begin, tix = 0, tiy = 0, tiz = 0;
  while (tile < tile_end && !tile_coords(tile, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++tile;
  if (tile < tile_end) load_next(tix << p.lx, tiy << p.ly, tiz << p.lz);
  while (tile < tile_end) {
    int nt = tile + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nt < tile_end && !tile_coords(nt, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nt;
    __syncthreads();   // previous tile's reads are done
    store_tile();
    __syncthreads();
    if (nt < tile_end) load_next(ntix << p.lx, ntiy << p.ly, ntiz << p.lz);
    float dA, dB, gA[NT], gB[NT];
    // one pair ahead: the next pair's reads go out in one group behind this pair's MFMAs.  Measured alternatives on the
    // 32 -> 32 layer at 128^3 (this order: 79-82 TFLOP/s): a row walk with one v_add per read instead of the (x, y, z)
    // decomposition per pair 62-65; reads strictly alternating with the MFMAs 57-59; hipcc's own order (every read sunk to
    // its MFMA behind lgkmcnt(0)) 70.
    rd(ps, dA, gA);
    for (int q = ps; q < npairs; q += 2 * ws) {
      rd(q + ws, dB, gB);
      __builtin_amdgcn_sched_barrier(0);
      mm(dA, gA);
      __builtin_amdgcn_sched_barrier(0);
      rd(q + 2 * ws < npairs ? q + 2 * ws : ps, dA, gA);        // (the last step re-reads pair `ps`: harmless, keeps the loop uniform)
      __builtin_amdgcn_sched_barrier(0);
      mm(dB, gB);
      __builtin_amdgcn_sched_barrier(0);
    }
    tile = nt; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- merge into dwk[b][tap][n][c] ----
  float* wout = p.dwk + (long)b * p.wsb + (long)(blockIdx.x % (unsigned)p.nrep) * p.rep_stride;
  if constexpr (NT == 7) {
    // one tap per tile (both forms): row r of the accumulator is output channel n0 + r, the lane's column is input channel
    // c0 + fr.  One pointer per tile, stepped through the 16 rows (kept compact: the 112 merges of the 7-tile variant
    // with per-element index arithmetic alone were 13 KB of code)
    const int c = c0 + fr;
    const bool full = n0 + 32 <= p.N;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tap = tg + wt * t;
      if (tap < ntaps && c < p.C) {
        float* pe = wout + ((long)tap * p.N + n0 + 4 * fh) * p.C + c;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (full || n0 + (e & 3) + 8 * (e >> 2) + 4 * fh < p.N) {
            if (p.plain) *pe = acc[t][e]; else atomicAdd(pe, acc[t][e]);
          }
          pe += ((e & 3) == 3 ? 5 : 1) * p.C;
        }
      }
    }
  } else {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * fh;        // accumulator row of this element; the lane's column is fr
      int tap, n, c;
      if (FORM == 0) {       // rows = dense n, columns = gathered (tap, c)
        const int i = tg + wt * t, tp = i * TPT + (fr >> lcp);
        tap = (i < ntiles && tp < ntaps && gch < chg) ? tp : -1; n = n0 + r; c = c0 + gch;
      } else {               // rows = gathered (tap, n), columns = dense c: the row's (tap, n) is lane-independent
        const int i = tg + wt * t, tp = i * TPT + (r >> lcp), nn = r & (CP - 1);
        tap = (i < ntiles && tp < ntaps && nn < chg) ? tp : -1; n = n0 + nn; c = c0 + fr;
      }
      if (tap >= 0 && n < p.N && c < p.C) {
        if (p.plain) wout[((long)tap * p.N + n) * p.C + c] = acc[t][e];
        else atomicAdd(wout + ((long)tap * p.N + n) * p.C + c, acc[t][e]);
      }
    }
  }
  }
}

// =====================================================================================
// conv_mfma_wgrad2_k -- stride-1 3x3x3 weight gradient for W >= 32 (any channel counts; one
// 32(n) x 32(c) slab of dwk per block, thin layers are zero-padded in LDS once).
// Same algorithm as conv_mfma_wgrad_k, restructured after its PMC profile (14.6 VALU + 7 SALU per
// MFMA, 64 % of wave cycles waiting): the tile is fixed at 2 x 4 x 32 voxels so that EVERY
// transposed LDS read is `lane base + compile-time immediate` (16 fully unrolled K steps x the
// wave's 7 taps, code specialised per wave), the per-thread staging descriptors are hoisted out
// of the tile loop, invalid (padding-channel) pieces are never staged, and all of a tile's global
// loads are issued before the first is consumed.
// =====================================================================================
struct Wgrad2P {
  const bf16_t* dyp; int ldn; long sbn;
  const bf16_t* xp; int ldc; long sbc;
  int N, C, D, H, W;
  int ntx, nty, ntz, tiles_total, tiles_per_block, cblocks;
  int vec_n, vec_c;
  int lgd, lgg;                 // log2 of the 8-channel chunks per staged dy / x row that are dealt to threads (1, 2 or 4)
  float* dwk; long wsb;
  int nrep; long rep_stride;    // small outputs: blocks merge into one of nrep replicas (summed afterwards)
};

// One tile's MFMAs for the calling wave.  toff[t] = LDS byte offset of the wave's t-th tap (wave-uniform
// scalars, so each gathered read costs one v_add with an SGPR operand; K-step offsets are scalars too and
// the +4-voxel second read is an immediate).
template <int HX, int HY, int NT>
__device__ __forceinline__ void wgrad2_tile(const char* Dt, const char* Gt, int lane_d, int lane_g, const int (&toff)[NT],
                                            int ntap, f32x16_t (&acc)[NT], int ks0, int ksd) {
  constexpr int PD = 64, PG = 64;
  // Fragments are read one step ahead into a second register set: hipcc otherwise issues every transposed read right
  // before its MFMA and waits lgkmcnt(0), exposing the LDS latency on each of them.
  auto rd_d = [&](int ks) -> bf16x8_t {
    const int z = ks >> 3, y = (ks >> 1) & 3, xh = ks & 1;
    const char* dp = Dt + lane_d + ((z * 4 + y) * 32 + xh * 16) * PD;
    const s4_t dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(dp));
    const s4_t dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(dp + 4 * PD));
    return (bf16x8_t){dlo[0], dlo[1], dlo[2], dlo[3], dhi[0], dhi[1], dhi[2], dhi[3]};
  };
  auto rd_g = [&](int ks, int t) -> bf16x8_t {
    const int z = ks >> 3, y = (ks >> 1) & 3, xh = ks & 1;
    const char* gp = Gt + lane_g + ((z * HY + y) * HX + xh * 16) * PG;
    const s4_t glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(gp + toff[t]));
    const s4_t ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(gp + toff[t] + 4 * PG));
    return (bf16x8_t){glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
  };
  // Every wave runs all NT column tiles (a wave with fewer valid ones recomputes tile 0 into an accumulator that is
  // never stored): no wave-dependent control flow inside the pipeline, and the slowest wave has NT tiles anyway.
  (void)ntap;
  bf16x8_t df = rd_d(ks0), gf = rd_g(ks0, 0);
#pragma unroll 2
  for (int ks = ks0; ks < 16; ks += ksd) {
    // K step ks = 16 consecutive x of row (z = ks >> 3, y = (ks >> 1) & 3), x half = ks & 1
    const int kn = ks + ksd < 16 ? ks + ksd : ks;      // (the last step re-reads itself: harmless, keeps the loop uniform)
    bf16x8_t dn = df;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bf16x8_t gn;
      if (t + 1 < NT) gn = rd_g(ks, t + 1);
      else { dn = rd_d(kn); gn = rd_g(kn, 0); }
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, gf, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      gf = gn;
    }
    df = dn;
  }
}

// VEC=1: every staged piece is a legal, fully valid 16-byte load; OCC = blocks per CU; K = kernel size (3 or 1:
// for 1x1x1 there is one tap and the four waves share it by K step instead of by tap).
// PK = taps packed side by side in the 32 MFMA columns when the layer has few input channels: 1 (C > 16: one tap,
// 32 channels), 2 (C <= 16: column j = tap kx0 + (j >> 4), channel j & 15), 4 (C <= 8: tap kx0 + (j >> 3), channel
// j & 7).  A packed tile is a (kz, ky) row of the kernel: {kx 0,1} {kx 2,-} for PK = 2 (18 MFMA tiles instead of 27),
// {kx 0,1,2,-} for PK = 4 (9 tiles); the "-" columns multiply the voxel one past kx = 2 and are never stored.  The
// x-shifted neighbour is a per-lane constant in the transposed read's address, so the inner loop is unchanged.
template <int VEC, int OCC, int K, int PK>
__global__ __launch_bounds__(256, OCC) void conv_mfma_wgrad2_k(Wgrad2P p) {
  constexpr int PADK = (K - 1) / 2;
  constexpr int NTILE = K == 1 ? 1 : (PK == 1 ? 27 : PK == 2 ? 18 : 9), NT = (NTILE + 3) / 4;
  constexpr int TX = 32, TY = 4, TZ = 2, TM = TX * TY * TZ, HX = TX + K - 1, HY = TY + K - 1, HZ = TZ + K - 1, HV = HX * HY * HZ;
  constexpr int NDP = TM * 4, NGP = HV * 4;                     // 16-byte pieces (dense, halo)
  constexpr int DIT = NDP / 256, GIT = (NGP + 255) / 256;        // 4 and 13 per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Dt = smem;                 // dense dy tile  [TM][32 ch]
  char* Gt = smem + TM * 64;       // gathered x halo [HV][32 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the per-wave code paths below are scalar branches
  const int b = blockIdx.z;
  const int nb = blockIdx.y / p.cblocks, cb = blockIdx.y % p.cblocks;
  const int n0 = nb * 32, c0 = cb * 32;
  const bf16_t* dyb = p.dyp + (long)b * p.sbn + n0;
  const bf16_t* xb = p.xp + (long)b * p.sbc + c0;
  const int chn = p.N - n0, chc = p.C - c0;      // valid channels of this block's slices
  // Staging: a 16-byte piece = 8 channels of one row.  Rows keep their 64-byte LDS slots, but only the 1, 2 or 4 chunks
  // a thin operand really has are dealt to the threads (p.lgd / p.lgg = log2 of that count): with the fixed 4-chunks-
  // per-row deal an 8-channel operand kept 3 of 4 lanes idle through 13 + 4 quarter-filled load instructions per tile.
  // Every piece of a thread still has the same 8-channel chunk (256 is a multiple of the chunk count).
  // (PK == 1, the >= 32-channel layers: the compile-time 4-chunk deal -- with run-time shifts the compiler stopped hoisting
  //  the per-piece row arithmetic out of the tile loop and the 64 -> 32 weight gradient went from 223 to 355 us)
  const int lgd = PK == 1 ? 2 : p.lgd, lgg = PK == 1 ? 2 : p.lgg;
  const int chd = tid & ((1 << lgd) - 1), chg = tid & ((1 << lgg) - 1);
  const bool dlive = chd * 8 < chn, glive = chg * 8 < chc;

  // zero the whole LDS image once: padding channels / never-staged pieces stay zero
  for (int i = tid; i < NDP + NGP; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  f32x16_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int g16 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int vrow = 8 * (g16 >> 1) + q;                          // voxel within a 16-voxel K step
  const int chan_b = ((g16 & 1) * 16 + 4 * pp) * 2;
  const int lane_d = vrow * 64 + chan_b;
  // gathered operand: which 4-channel piece of which x-neighbour this lane feeds into the 16-lane transpose
  const int chunk = (g16 & 1) * 4 + pp;
  const int lane_g = PK == 1 ? vrow * 64 + chan_b
                   : PK == 2 ? (vrow + (chunk >> 2)) * 64 + (chunk & 3) * 8
                             : (vrow + (chunk >> 1)) * 64 + (chunk & 1) * 8;

  int toff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = wid + 4 * t < NTILE ? wid + 4 * t : 0;   // MFMA tile index: tap (PK 1), (row, kx pair) (PK 2), row (PK 4)
    const int row = PK == 1 ? i / 3 : PK == 2 ? i >> 1 : i, kx0 = PK == 1 ? i % 3 : PK == 2 ? (i & 1) * 2 : 0;
    toff[t] = K == 1 ? 0 : __builtin_amdgcn_readfirstlane((((row / 3) * HY + row % 3) * HX + kx0) * 64);
  }
  const int ntap = K == 1 ? 1 : (NTILE - wid + 3) / 4;

  const int tile_begin = xcd_remap(blockIdx.x, gridDim.x) * p.tiles_per_block;
  int tile_end = tile_begin + p.tiles_per_block;
  if (tile_end > p.tiles_total) tile_end = p.tiles_total;

  // ---- software pipeline: tile t+1's global loads are in flight (registers) while tile t computes ----
  uint4 dv[DIT], gv[GIT];
  auto load_tile = [&](int x0, int y0, int z0) {
    if (dlive) {
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int row = (tid + 256 * it) >> lgd;
        const int vx = row & 31, vy = (row >> 5) & 3, vz = row >> 7;
        const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
        dv[it] = make_uint4(0, 0, 0, 0);
        if (row < TM && gz < p.D && gy < p.H && gx < p.W) {
          const bf16_t* src = dyb + (unsigned)(((gz * p.H + gy) * p.W + gx) * p.ldn + chd * 8);   // < 2^31 elements (checked)
          dv[it] = VEC ? *reinterpret_cast<const uint4*>(src) : load8_raw(src, chn - chd * 8, p.vec_n);
        }
      }
    }
    if (glive) {
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int row = (tid + 256 * it) >> lgg;
        const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
        const int gz = z0 - PADK + hz, gy = y0 - PADK + hy, gx = x0 - PADK + hx;
        gv[it] = make_uint4(0, 0, 0, 0);
        if (row < HV && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
          const bf16_t* src = xb + (unsigned)(((gz * p.H + gy) * p.W + gx) * p.ldc + chg * 8);
          gv[it] = VEC ? *reinterpret_cast<const uint4*>(src) : load8_raw(src, chc - chg * 8, p.vec_c);
        }
      }
    }
  };
  // (every piece of a thread has the same channel chunk: one mask per operand, applied when the piece goes to LDS)
  const uint4 dmask = mask8(VEC ? 8 : chn - chd * 8), gmask = mask8(VEC ? 8 : chc - chg * 8);
  auto and4 = [](uint4 v, uint4 m) { v.x &= m.x; v.y &= m.y; v.z &= m.z; v.w &= m.w; return v; };
  auto store_tile = [&]() {
    if (dlive) {
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int row = (tid + 256 * it) >> lgd;
        if (row < TM) reinterpret_cast<uint4*>(Dt)[row * 4 + chd] = VEC ? dv[it] : and4(dv[it], dmask);
      }
    }
    if (glive) {
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int row = (tid + 256 * it) >> lgg;
        if (row < HV) reinterpret_cast<uint4*>(Gt)[row * 4 + chg] = VEC ? gv[it] : and4(gv[it], gmask);
      }
    }
  };

  int tile = tile_begin, tix = 0, tiy = 0, tiz = 0;
  while (tile < tile_end && !tile_coords(tile, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++tile;
  if (tile < tile_end) load_tile(tix * TX, tiy * TY, tiz * TZ);
  while (tile < tile_end) {
    int nt = tile + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nt < tile_end && !tile_coords(nt, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nt;
    __syncthreads();     // previous tile's LDS reads are done
    store_tile();
    __syncthreads();
    constexpr bool EARLY = OCC == 1 || PK > 1 || K == 1;    // few accumulators: room to hold the next tile in registers
    if (EARLY) { if (nt < tile_end) load_tile(ntix * TX, ntiy * TY, ntiz * TZ); }   // prefetch under the MFMAs
    wgrad2_tile<HX, HY, NT>(Dt, Gt, lane_d, lane_g, toff, ntap, acc, K == 1 ? wid : 0, K == 1 ? 4 : 1);
    if (!EARLY) { if (nt < tile_end) load_tile(ntix * TX, ntiy * TY, ntiz * TZ); }  // the co-resident block covers this latency
    tile = nt;
  }
  // ---- merge into dwk[b][tap][n][c] ----
  float* wout = p.dwk + (long)b * p.wsb + (long)(blockIdx.x % (unsigned)p.nrep) * p.rep_stride;
  const int fr = lane & 31, fh = lane >> 5;
  const int sub = PK == 1 ? 0 : PK == 2 ? fr >> 4 : fr >> 3;           // x-neighbour of this lane's column
  const int cc = PK == 1 ? fr : PK == 2 ? (fr & 15) : (fr & 7);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int i = wid + 4 * t;
    const int row = PK == 1 ? i / 3 : PK == 2 ? i >> 1 : i, kx = (PK == 1 ? i % 3 : PK == 2 ? (i & 1) * 2 : 0) + sub;
    const int tap = K == 1 ? 0 : row * 3 + kx;
    if (t < ntap && kx < 3) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int c = c0 + cc;
        if (n < p.N && c < p.C) atomicAdd(wout + ((long)tap * p.N + n) * p.C + c, acc[t][e]);
      }
    }
  }
}

static bool wgrad2_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  return d->form == 0 && d->stride == 1 && (d->ksize == 3 || d->ksize == 1) && x->dtype == COMA_BF16 && dy->dtype == COMA_BF16 &&
         x->W >= 32 && (long)t_vox(x) * x->ld < (1L << 31) && (long)t_vox(dy) * dy->ld < (1L << 31);
}

// Small weight tensors (the 1..16-channel layers, 1x1x1 gates): a thousand blocks merging into a few cache lines
// serialise in the L2 atomic unit (1 -> 32 channels at 128^3: 520 us, of which ~400 us were atomics).  Their blocks
// merge into WGRAD_NREP replicas in the workspace instead, summed by one tiny kernel.
#define WGRAD_NREP 64
#define WGRAD_REP_MAX_ELEMS 16384
static long wgrad_out_elems(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  return (long)d->ksize * d->ksize * d->ksize * dy->C * x->C * (d->per_sample_w ? x->B : 1);
}
// out[i] = sum over the replicas.  A few thousand outputs x 64 replicas: the serial 64-load walk per thread this replaces was
// latency-bound (17.7 us per launch, 13 launches per step); here 8 lanes share an output (8 independent loads each) and
// fold through a wave shuffle.
__global__ __launch_bounds__(256) void wgrad_replica_sum_k(const float* __restrict__ rep, int nrep, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int part = threadIdx.x & 7;
  float a = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int r = part; r < nrep; r += 8) a += rep[(long)r * n + i];
  }
  a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 1, 64);
  if (part == 0 && i < n) out[i] = a;
}

// =====================================================================================
// conv_thin16f_wgrad_k -- fp32 weight gradient of the few-channel full-resolution layers (stride-1 3x3x3, C <= 16,
// N <= 16, or C <= 4 and N <= 32), exact fp32 on v_mfma_f32_16x16x4_f32 with the VOXELS along K:
//   dW[tap][n][c] += dy[v][n] * x[v + tap][c]:  rows = 16 output channels (A = dy), columns = 16 = TPM taps x CP channels
//   (B = x at the tap's offset: 27 / 14 / 7 column tiles for CP = 16 / 8 / 4), 4 voxels per MFMA.
// A wave keeps ALL column tiles' accumulators (<= 108 VGPRs) and walks its 64 voxels of every tile in 16 steps; both
// operands are one 4-byte LDS read per lane and MFMA (dense dy tile [256][16 NB], x halo [816][CP]).  The generic fp32
// weight-gradient kernel spent 32 x 32 tiles on these layers: 16 -> 16 at 128^3 1.8 ms, 16 -> 1 1.8 ms, 8 -> 8 1.2 ms.
// Blocks merge through LDS (ds_add) and then into one of `nrep` replicas with fp32 atomics.
// =====================================================================================
struct Thin16FWP {
  const float* x; int ldx; long sbx; int D, H, W, C;
  const float* dy; int ldn; long sbn; int N;
  unsigned xbytes, dbytes;
  int ntx, nty, ntz, ids_total, ids_per_block;
  float* dwk; long wsb;
  int nrep; long rep_stride;
};

template <int CP, int NB>      // (CP = 16: 27 column tiles = 108 accumulator registers + 68 of staged pieces: one block per CU)
__global__ __launch_bounds__(256, CP == 16 ? 1 : 2) void conv_thin16f_wgrad_k(Thin16FWP p) {
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ, TM = TX * TY * TZ;
  constexpr int TPM = 16 / CP, NM = (27 + TPM - 1) / TPM;
  constexpr int ND = NB * 16;                        // floats per dense dy row
  constexpr int PCH = CP / 4, DCH = ND / 4;          // 16-byte pieces per halo / dense row
  constexpr int HP = HV * PCH, DP = TM * DCH, HIT = (HP + 255) / 256, DIT = DP / 256, NIT = HIT + DIT;
  static_assert(NIT <= 32, "two staging pieces per K step at most");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Hl = reinterpret_cast<float*>(smem);                      // [HV][CP]
  float* Dl = Hl + HV * CP;                                        // [TM][ND]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int lv = lane & 15, lg = lane >> 4;
  const float* xb = p.x + (long)b * p.sbx;
  const float* db = p.dy + (long)b * p.sbn;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, p.dbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;

  // ---- staging descriptors: pieces 0..HIT-1 = x halo, HIT.. = dense dy ----
  int s_z[NIT], s_y[NIT], s_x[NIT];
  unsigned s_boff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it < HIT) {
      const int piece = tid + 256 * it;
      const int row = piece / PCH, ch = piece % PCH;
      const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
      s_z[it] = piece < HP ? hz - 1 : (1 << 20); s_y[it] = hy - 1; s_x[it] = hx - 1;
      s_boff[it] = (unsigned)(ch * 16);
    } else {
      const int piece = tid + 256 * (it - HIT);
      const int row = piece / DCH, ch = piece % DCH;
      s_z[it] = row >> 7; s_y[it] = (row >> 5) & 3; s_x[it] = row & 31;
      s_boff[it] = (unsigned)(ch * 16);
    }
  }
  const int xval = p.C - 4 * (tid % PCH), dval = p.N - 4 * (tid % DCH);
  const uint4 xmask = make_uint4(xval > 0 ? ~0u : 0u, xval > 1 ? ~0u : 0u, xval > 2 ? ~0u : 0u, xval > 3 ? ~0u : 0u);
  const uint4 dmask = make_uint4(dval > 0 ? ~0u : 0u, dval > 1 ? ~0u : 0u, dval > 2 ? ~0u : 0u, dval > 3 ? ~0u : 0u);
  uint4 sreg[NIT];
  auto issue = [&](int it, int z0, int y0, int x0, const __amdgpu_buffer_rsrc_t& rx, const __amdgpu_buffer_rsrc_t& rd) __attribute__((always_inline)) {
    const int gz = z0 + s_z[it], gy = y0 + s_y[it], gx = x0 + s_x[it];
    const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const int ld = it < HIT ? p.ldx : p.ldn;
    const unsigned off = (unsigned)(((gz * p.H + gy) * p.W + gx) * ld) * 4u + s_boff[it];
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(it < HIT ? rx : rd, ok ? off : OOB, 0, 0);
    sreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      uint4 v = sreg[it];
      if (it < HIT) {
        if (tid + 256 * it < HP) {
          v.x &= xmask.x; v.y &= xmask.y; v.z &= xmask.z; v.w &= xmask.w;
          reinterpret_cast<uint4*>(Hl)[tid + 256 * it] = v;
        }
      } else {
        v.x &= dmask.x; v.y &= dmask.y; v.z &= dmask.z; v.w &= dmask.w;
        reinterpret_cast<uint4*>(Dl)[tid + 256 * (it - HIT)] = v;
      }
    }
  };

  // ---- fragment addressing (floats): this wave's 64 voxels = rows (gz, gy0 + {0, 1}), K step s = voxels 4 s + lg ----
  const int gz = wid >> 1, gy0 = 2 * (wid & 1);
  const int dbase = ((gz * 4 + gy0) * 32 + lg) * ND + lv;                       // + step offset, + nb * 16
  const int sub = lv / CP, cc = lv % CP;
  const int xbase = ((gz * HY + gy0) * HX + lg) * CP + cc;                      // + step offset + tap offset
  int toff[NM];                                                                  // (CP = 16: compile-time constants)
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int t = TPM == 1 ? m : (m * TPM + sub < 27 ? m * TPM + sub : 0);
    toff[m] = (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * CP;
  }
  f32x4_t acc[NM][NB];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[m][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id < id_end) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue(it, tiz * TZ, tiy * TY, tix * TX, rs_x, rs_d);
  }
  while (id < id_end) {
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    const __amdgpu_buffer_rsrc_t rn_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, has_next ? p.xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, has_next ? p.dbytes : 0, 0x00020000);
    __syncthreads();
    store_tile();
    __syncthreads();
    float af[2][NB], bf[2][NM];
    auto rd = [&](int s_, int buf) __attribute__((always_inline)) {
      const int so_d = ((s_ >> 3) * 32 + 4 * (s_ & 7)) * ND, so_x = ((s_ >> 3) * HX + 4 * (s_ & 7)) * CP;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) af[buf][nb] = Dl[dbase + so_d + nb * 16];
#pragma unroll
      for (int m = 0; m < NM; ++m) bf[buf][m] = Hl[xbase + so_x + toff[m]];
    };
    rd(0, 0);
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      if (s_ + 1 < 16) rd(s_ + 1, (s_ + 1) & 1);
      if (s_ < NIT) issue(s_, ntiz * TZ, ntiy * TY, ntix * TX, rn_x, rn_d);
      if (s_ + 16 < NIT) issue(s_ + 16, ntiz * TZ, ntiy * TY, ntix * TX, rn_x, rn_d);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s_ & 1][nb], bf[s_ & 1][m], acc[m][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- merge: the four waves' tiles add up in LDS, then one fp32 atomic per weight into this block's replica ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                     // [NM][NB][16 n][16 col]
  for (int i = tid; i < NM * NB * 256; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&red[((m * NB + nb) * 16 + 4 * lg + j) * 16 + lv], acc[m][nb][j]);
  __syncthreads();
  float* wout = p.dwk + (long)b * p.wsb + (long)(blockIdx.x % (unsigned)p.nrep) * p.rep_stride;
  for (int i = tid; i < NM * NB * 256; i += 256) {
    const int col = i & 15, row = (i >> 4) & 15, nb = (i >> 8) % NB, m = i / (256 * NB);
    const int tap = m * TPM + col / CP, c = col % CP, n = nb * 16 + row;
    if (tap < 27 && n < p.N && c < p.C) atomicAdd(wout + ((long)tap * p.N + n) * p.C + c, red[i]);
  }
}

// =====================================================================================
// conv_thin16_wgrad_k -- bf16 weight gradient of the few-channel full-resolution layers (stride-1 3x3x3, C <= 16, N <= 16,
// or C <= 8 and N <= 32) on v_mfma_f32_16x16x32_bf16 with the VOXELS along K (the structure of conv_thin16f_wgrad_k):
// rows = 16 output channels, columns = 16 = TPM taps x CP channels (27 / 14 column tiles), 32 voxels per MFMA.  Both
// operands are voxel-major in LDS (dense dy tile [256][16 NB], x halo [816][CP]) and are read TRANSPOSED with
// ds_read_b64_tr_b16: a 16-lane group fetches 4 voxels x 16 columns, lane 4q+p supplying the address of voxel q, columns
// 4p..4p+3 -- so for CP = 8 the lanes p = 2, 3 simply point at the next tap's rows.  K index 8g+j <-> voxel x = 4g+j (j < 4),
// 16+4g+(j-4): the two 16-lane groups of a 32-lane half then read 8 consecutive 32-byte rows (conflict-free).
// The 32x32x16 kernel it replaces here packed 2 or 4 taps into 32 columns and padded N to 32: 16 -> 16 at 128^3 212 us.
// =====================================================================================
struct Thin16WP {
  const bf16_t* x; int ldx; long sbx; int D, H, W, C;
  const bf16_t* dy; int ldn; long sbn; int N;
  unsigned xbytes, dbytes;
  int ntx, nty, ntz, ids_total, ids_per_block;
  float* dwk; long wsb;
  int nrep; long rep_stride;
};

template <int CP, int NB>
__global__ __launch_bounds__(256, 2) void conv_thin16_wgrad_k(Thin16WP p) {
  constexpr int TX = 32, TY = 4, TZ = 2, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HV = HX * HY * HZ, TM = TX * TY * TZ;
  constexpr int TPM = 16 / CP, NM = (27 + TPM - 1) / TPM;
  constexpr int ND = NB * 16;                        // bf16 per dense dy row
  constexpr int PX = CP * 2, PD = ND * 2;            // row pitches (bytes)
  constexpr int PCH = CP / 8, DCH = ND / 8;          // 16-byte pieces per halo / dense row
  constexpr int HP = HV * PCH, DP = TM * DCH, HIT = (HP + 255) / 256, DIT = DP / 256, NIT = HIT + DIT;
  constexpr int RING = 8;                            // B fragments in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;                                   // [HV][CP]
  char* Dl = smem + (HV + 2) * PX;                   // [TM][ND]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int lv = lane & 15, lg = lane >> 4;
  const bf16_t* xb = p.x + (long)b * p.sbx;
  const bf16_t* db = p.dy + (long)b * p.sbn;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(db), 0, p.dbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;

  int s_z[NIT], s_y[NIT], s_x[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it < HIT) {
      const int piece = tid + 256 * it;
      const int row = piece / PCH;
      const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
      s_z[it] = piece < HP ? hz - 1 : (1 << 20); s_y[it] = hy - 1; s_x[it] = hx - 1;
    } else {
      const int row = (tid + 256 * (it - HIT)) / DCH;
      s_z[it] = row >> 7; s_y[it] = (row >> 5) & 3; s_x[it] = row & 31;
    }
  }
  const unsigned xoffb = (unsigned)((tid % PCH) * 16), doffb = (unsigned)((tid % DCH) * 16);
  const uint4 xmask = mask8(p.C - 8 * (tid % PCH)), dmask = mask8(p.N - 8 * (tid % DCH));
  uint4 sreg[NIT];
  auto issue = [&](int it, int z0, int y0, int x0, const __amdgpu_buffer_rsrc_t& rx, const __amdgpu_buffer_rsrc_t& rd) __attribute__((always_inline)) {
    const int gz = z0 + s_z[it], gy = y0 + s_y[it], gx = x0 + s_x[it];
    const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const unsigned off = (unsigned)(((gz * p.H + gy) * p.W + gx) * (it < HIT ? p.ldx : p.ldn)) * 2u + (it < HIT ? xoffb : doffb);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(it < HIT ? rx : rd, ok ? off : OOB, 0, 0);
    sreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      uint4 v = sreg[it];
      if (it < HIT) {
        if (tid + 256 * it < HP) {
          v.x &= xmask.x; v.y &= xmask.y; v.z &= xmask.z; v.w &= xmask.w;
          reinterpret_cast<uint4*>(Hl)[tid + 256 * it] = v;
        }
      } else {
        v.x &= dmask.x; v.y &= dmask.y; v.z &= dmask.z; v.w &= dmask.w;
        reinterpret_cast<uint4*>(Dl)[tid + 256 * (it - HIT)] = v;
      }
    }
  };

  // ---- transposed-read addressing: lane 4q+p of group lg supplies voxel x = 4 lg + q (+16 for the second read), columns 4p.. ----
  const int gz = wid >> 1, gy0 = 2 * (wid & 1);
  const int q4 = lv >> 2, p4 = lv & 3;
  const int vx = 4 * lg + q4;
  const int d_addr = ((gz * 4 + gy0) * 32 + vx) * PD + 8 * p4;                                   // + row step s * 32 * PD, + 16 * PD second read, + nb * 32 bytes
  const int sub = (4 * p4) / CP, cin = (4 * p4) % CP;
  const int x_addr = ((gz * HY + gy0) * HX + vx) * PX + cin * 2;                                 // + s * HX * PX, + 16 * PX second read, + tap offset
  int toff[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int t = TPM == 1 ? m : (m * TPM + sub < 27 ? m * TPM + sub : 0);    // (padding taps read tap 0: their columns are never stored)
    toff[m] = (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * PX;
  }
  f32x4_t acc[NM][NB];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[m][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id < id_end) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue(it, tiz * TZ, tiy * TY, tix * TX, rs_x, rs_d);
  }
  while (id < id_end) {
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    const __amdgpu_buffer_rsrc_t rn_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, has_next ? p.xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(db), 0, has_next ? p.dbytes : 0, 0x00020000);
    __syncthreads();
    store_tile();
    __syncthreads();
    // the sequence of (row step s, column tile m) pairs is unrolled; B fragments run RING pairs ahead
    constexpr int NPAIR = 2 * NM;
    bf16x8_t bring[RING];
    auto rdb = [&](int pi) __attribute__((always_inline)) -> bf16x8_t {
      const int s_ = pi / NM, m = pi % NM;
      const char* src = Hl + x_addr + s_ * HX * PX + toff[m];
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src + 16 * PX));
      return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rda = [&](int s_, int nb) __attribute__((always_inline)) -> bf16x8_t {
      const char* src = Dl + d_addr + s_ * 32 * PD + nb * 32;
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src + 16 * PD));
      return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    bf16x8_t afr[2][NB];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) afr[s_][nb] = rda(s_, nb);
#pragma unroll
    for (int pi = 0; pi < RING && pi < NPAIR; ++pi) bring[pi] = rdb(pi);
#pragma unroll
    for (int pi = 0; pi < NPAIR; ++pi) {
      const int s_ = pi / NM, m = pi % NM;
      const bf16x8_t bcur = bring[pi % RING];
      if (pi + RING < NPAIR) bring[pi % RING] = rdb(pi + RING);
      if (pi < NIT) issue(pi, ntiz * TZ, ntiy * TY, ntix * TX, rn_x, rn_d);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[s_][nb], bcur, acc[m][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(NIT <= NPAIR, "one staging piece per MFMA step");
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- merge (as conv_thin16f_wgrad_k) ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                     // [NM][NB][16 n][16 col]
  for (int i = tid; i < NM * NB * 256; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&red[((m * NB + nb) * 16 + 4 * lg + j) * 16 + lv], acc[m][nb][j]);
  __syncthreads();
  float* wout = p.dwk + (long)b * p.wsb + (long)(blockIdx.x % (unsigned)p.nrep) * p.rep_stride;
  for (int i = tid; i < NM * NB * 256; i += 256) {
    const int col = i & 15, row = (i >> 4) & 15, nb = (i >> 8) % NB, m = i / (256 * NB);
    const int tap = m * TPM + col / CP, c = col % CP, n = nb * 16 + row;
    if (tap < 27 && n < p.N && c < p.C) atomicAdd(wout + ((long)tap * p.N + n) * p.C + c, red[i]);
  }
}

static bool thin16_wgrad_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  static const bool on = []{ const char* e = getenv("COMA_THIN16W"); return !(e && e[0] == '0'); }();
  const int nmax = x->C <= 8 ? 32 : 16;
  return on && d->form == 0 && d->ksize == 3 && d->stride == 1 && x->dtype == COMA_BF16 && dy->dtype == COMA_BF16 && x->W >= 32 &&
         x->C <= 16 && dy->C <= nmax && x->ld % 8 == 0 && x->sb % 8 == 0 && dy->ld % 8 == 0 && dy->sb % 8 == 0 &&
         (!x->data || aligned16(x->data)) && (!dy->data || aligned16(dy->data)) &&
         (unsigned long long)t_vox(x) * x->ld * 2 < 0x7fff0000ull && (unsigned long long)t_vox(dy) * dy->ld * 2 < 0x7fff0000ull;
}

static int conv_thin16_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, void* ws,
                             size_t ws_bytes, hipStream_t s, int zeroed) {
  Thin16WP q;
  q.x = (const bf16_t*)x->data; q.ldx = (int)x->ld; q.sbx = x->sb; q.D = x->D; q.H = x->H; q.W = x->W; q.C = x->C;
  q.dy = (const bf16_t*)dy->data; q.ldn = (int)dy->ld; q.sbn = dy->sb; q.N = dy->C;
  q.xbytes = (unsigned)((unsigned long long)t_vox(x) * x->ld * 2);
  q.dbytes = (unsigned)((unsigned long long)t_vox(dy) * dy->ld * 2);
  q.ntx = (q.W + 31) / 32; q.nty = (q.H + 3) / 4; q.ntz = (q.D + 1) / 2;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  int gx = 512 / x->B;
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  const long wsz1 = 27L * q.N * q.C, wsz = wsz1 * (d->per_sample_w ? x->B : 1);
  q.wsb = d->per_sample_w ? wsz1 : 0;
  const bool replicas = wsz <= WGRAD_REP_MAX_ELEMS && ws && ws_bytes >= sizeof(float) * wsz * WGRAD_NREP;
  q.nrep = replicas ? WGRAD_NREP : 1;
  q.rep_stride = replicas ? wsz : 0;
  q.dwk = replicas ? (float*)ws : dwk;
  if (!(zeroed & (replicas ? COMA_ZEROED_WS : COMA_ZEROED_OUT)) && hipMemsetAsync(q.dwk, 0, sizeof(float) * wsz * q.nrep, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const int cp = q.C > 8 ? 16 : 8, nb = q.N > 16 ? 2 : 1;
  const int nm = cp == 16 ? 27 : 14;
  size_t lds = (size_t)(34 * 6 * 4 + 2) * cp * 2 + (size_t)256 * nb * 16 * 2;
  if (lds < (size_t)nm * nb * 256 * 4) lds = (size_t)nm * nb * 256 * 4;
  const dim3 grid((unsigned)gx, 1, (unsigned)x->B);
  coma_set_kernel_tag("conv_thin16_wgrad_k<%d, %d>", cp, nb);
  if (cp == 16) hipLaunchKernelGGL((conv_thin16_wgrad_k<16, 1>), grid, dim3(256), lds, s, q);
  else if (nb == 2) hipLaunchKernelGGL((conv_thin16_wgrad_k<8, 2>), grid, dim3(256), lds, s, q);
  else hipLaunchKernelGGL((conv_thin16_wgrad_k<8, 1>), grid, dim3(256), lds, s, q);
  COMA_LAUNCH_CHECK();
  if (replicas) {
    hipLaunchKernelGGL(wgrad_replica_sum_k, dim3((unsigned)((wsz + 31) / 32)), dim3(256), 0, s, (const float*)ws, WGRAD_NREP, wsz, dwk);
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

// =====================================================================================
// conv_f32_wgrad16_k -- fp32 weight gradient of the >= 32-channel stride-1 3x3x3 layers (W >= 32) in the voxels-along-K
// form of conv_thin16f_wgrad_k: a block owns a 32 (n) x 32 (c) weight tile pair and walks 2x4x32-voxel tiles; wave w keeps
// the 27 taps of ONE 16 x 16 sub-tile (n half = w & 1, c half = w >> 1: 108 accumulator registers) and runs all 256
// voxels of the tile through them, 4 voxels per v_mfma_f32_16x16x4_f32, both operands one 4-byte LDS read per lane
// (dense dy tile [256][32], x halo [816][32]: 136 KB, one block per CU).  The next tile's 34 staging pieces per thread
// are fetched into registers in one burst right after the LDS image is complete (3-4 % of a tile's 55 k MFMA cycles).
// =====================================================================================
struct F32W16P {
  const float* dn; int ldd; long sbd; int Mz, My, Mx;      // dense operand (FORM 0: dy, FORM 1: x) on the coarse grid
  const float* ga; int ldg; long sbg; int Gz, Gy, Gx;      // gathered operand (FORM 0: x, FORM 1: dy)
  int N, C;
  unsigned dbytes, gbytes;
  int ntx, nty, ntz, ids_total, ids_per_block, cblocks;
  float* dwk; long wsb;
};

// S = stride (1 or 2), FORM = 0 forward convolution (dense dy, gathered x) / 1 transposed stride-2 convolution (dense x,
// gathered dy).  S = 2: a 1 x 2 x 32 dense tile (64 voxels) with its 3 x 5 x 65 gathered region (125 KB of LDS).
template <int S, int FORM>
__global__ __launch_bounds__(256, 1) void conv_f32_wgrad16_k(F32W16P p) {
  constexpr int TX = 32, TY = S == 1 ? 4 : 2, TZ = S == 1 ? 2 : 1, TM = TX * TY * TZ;
  constexpr int HX = (TX - 1) * S + 3, HY = (TY - 1) * S + 3, HZ = (TZ - 1) * S + 3, HV = HX * HY * HZ;
  constexpr int CB = 32;                               // channels per block on either side
  constexpr int HIT = (HV * 8 + 255) / 256, DIT = TM * 8 / 256, NIT = HIT + DIT;      // 16-byte pieces per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Hl = reinterpret_cast<float*>(smem);          // [HV][32] gathered
  float* Dl = Hl + HV * CB;                            // [TM][32] dense
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int n0 = (blockIdx.y / p.cblocks) * CB, c0 = (blockIdx.y % p.cblocks) * CB;
  const int lv = lane & 15, lg = lane >> 4;
  const float* dnb = p.dn + (long)b * p.sbd + (FORM == 0 ? n0 : c0);
  const float* gab = p.ga + (long)b * p.sbg + (FORM == 0 ? c0 : n0);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gab), 0, p.gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dnb), 0, p.dbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;

  // staging: piece = tid + 256 it -> (row = piece >> 3, 4-channel chunk = tid & 7); positions relative to the tile origin + 1
  const unsigned choff = (unsigned)((tid & 7) * 16);
  int s_pos[NIT];                                      // packed z | y << 4 | x << 8
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it < HIT) {
      const int row = (tid + 256 * it) >> 3;
      const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
      s_pos[it] = row < HV ? (hz | (hy << 4) | (hx << 8)) : (15 | (15 << 4) | (1023 << 8));
    } else {
      const int row = (tid + 256 * (it - HIT)) >> 3;
      s_pos[it] = ((row / (TY * 32)) + 1) | ((((row >> 5) % TY) + 1) << 4) | (((row & 31) + 1) << 8);
    }
  }
  uint4 sreg[NIT];
  auto issue_all = [&](int z0, int y0, int x0, const __amdgpu_buffer_rsrc_t& rg, const __amdgpu_buffer_rsrc_t& rd) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int sc = it < HIT ? S : 1;
      const int gz = sc * z0 - 1 + (s_pos[it] & 15), gy = sc * y0 - 1 + ((s_pos[it] >> 4) & 15), gx = sc * x0 - 1 + (s_pos[it] >> 8);
      const bool ok = it < HIT ? ((unsigned)gz < (unsigned)p.Gz && (unsigned)gy < (unsigned)p.Gy && (unsigned)gx < (unsigned)p.Gx)
                               : ((unsigned)gz < (unsigned)p.Mz && (unsigned)gy < (unsigned)p.My && (unsigned)gx < (unsigned)p.Mx);
      const unsigned off = it < HIT ? (unsigned)(((gz * p.Gy + gy) * p.Gx + gx) * p.ldg) * 4u + choff
                                    : (unsigned)(((gz * p.My + gy) * p.Mx + gx) * p.ldd) * 4u + choff;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(it < HIT ? rg : rd, ok ? off : OOB, 0, 0);
      sreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it < HIT) { if (((tid + 256 * it) >> 3) < HV) reinterpret_cast<uint4*>(Hl)[tid + 256 * it] = sreg[it]; }
      else reinterpret_cast<uint4*>(Dl)[tid + 256 * (it - HIT)] = sreg[it];
    }
  };

  const int nh = wid & 1, ch = wid >> 1;               // this wave's 16 x 16 sub-tile (n half, c half)
  // rows of the MFMA = output channels n (operand A), columns = input channels c (operand B)
  const int dbase = lg * CB + 16 * (FORM == 0 ? nh : ch) + lv;            // dense:    + voxel row * CB
  const int gbase = S * lg * CB + 16 * (FORM == 0 ? ch : nh) + lv;        // gathered: + region row * CB + tap offset
  f32x4_t acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id < id_end) issue_all(tiz * TZ, tiy * TY, tix * TX, rs_g, rs_d);
  while (id < id_end) {
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    const __amdgpu_buffer_rsrc_t rn_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gab), 0, has_next ? p.gbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dnb), 0, has_next ? p.dbytes : 0, 0x00020000);
    __syncthreads();
    store_tile();
    __syncthreads();
    issue_all(ntiz * TZ, ntiy * TY, ntix * TX, rn_g, rn_d);
    // TZ * TY rows of 32 dense voxels = 8 K steps each; fragments one step ahead
#pragma unroll 1
    for (int r = 0; r < TZ * TY; ++r) {
      const int drow = r * 32 * CB, grow = ((((r / TY) * S) * HY + (r % TY) * S) * HX) * CB;
      float df[2], gf[2][27];
      auto rd = [&](int s_, int buf) __attribute__((always_inline)) {
        df[buf] = Dl[dbase + drow + 4 * s_ * CB];
#pragma unroll
        for (int t = 0; t < 27; ++t) gf[buf][t] = Hl[gbase + grow + S * 4 * s_ * CB + (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * CB];
      };
      rd(0, 0);
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        if (s_ + 1 < 8) rd(s_ + 1, (s_ + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 27; ++t)
          acc[t] = FORM == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(df[s_ & 1], gf[s_ & 1][t], acc[t], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s_ & 1][t], df[s_ & 1], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  // ---- merge: every wave owns its outputs within the block: fp32 atomics straight into dwk[tap][n][c] ----
  float* wout = p.dwk + (long)b * p.wsb;
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * nh + 4 * lg + j, c = c0 + 16 * ch + lv;
      atomicAdd(wout + ((long)t * p.N + n) * p.C + c, acc[t][j]);
    }
}

static bool f32_wgrad16_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  static const bool on = []{ const char* e = getenv("COMA_F32W16"); return !(e && e[0] == '0'); }();
  const coma_tensor* dn = d->form == 0 ? dy : x;        // dense (coarse) operand
  const bool shape = d->ksize == 3 && ((d->form == 0 && (d->stride == 1 || d->stride == 2)) || (d->form == 1 && d->stride == 2));
  return on && shape && x->dtype == COMA_F32 && dy->dtype == COMA_F32 && dn->W >= 32 &&
         x->C % 32 == 0 && dy->C % 32 == 0 && x->ld % 4 == 0 && x->sb % 4 == 0 && dy->ld % 4 == 0 && dy->sb % 4 == 0 &&
         (!x->data || aligned16(x->data)) && (!dy->data || aligned16(dy->data)) &&
         (unsigned long long)t_vox(x) * x->ld * 4 < 0x7fff0000ull && (unsigned long long)t_vox(dy) * dy->ld * 4 < 0x7fff0000ull;
}

static int conv_f32_wgrad16(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, hipStream_t s, int zeroed) {
  const coma_tensor* dn = d->form == 0 ? dy : x;
  const coma_tensor* ga = d->form == 0 ? x : dy;
  F32W16P q;
  q.dn = (const float*)dn->data; q.ldd = (int)dn->ld; q.sbd = dn->sb; q.Mz = dn->D; q.My = dn->H; q.Mx = dn->W;
  q.ga = (const float*)ga->data; q.ldg = (int)ga->ld; q.sbg = ga->sb; q.Gz = ga->D; q.Gy = ga->H; q.Gx = ga->W;
  q.N = dy->C; q.C = x->C;
  // (descriptor ranges are measured from the block's channel offset; the voxel bounds test keeps every piece inside)
  q.dbytes = (unsigned)((unsigned long long)t_vox(dn) * dn->ld * 4);
  q.gbytes = (unsigned)((unsigned long long)t_vox(ga) * ga->ld * 4);
  const int S = d->stride, tz = S == 1 ? 2 : 1, ty = S == 1 ? 4 : 2;
  q.ntx = (q.Mx + 31) / 32; q.nty = (q.My + ty - 1) / ty; q.ntz = (q.Mz + tz - 1) / tz;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  q.cblocks = q.C / 32;
  const int pairs = q.cblocks * (q.N / 32);
  int gx = 256 / (pairs * x->B);
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  const long wsz1 = 27L * q.N * q.C, wsz = wsz1 * (d->per_sample_w ? x->B : 1);
  q.wsb = d->per_sample_w ? wsz1 : 0;
  q.dwk = dwk;
  if (!(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dwk, 0, sizeof(float) * wsz, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const size_t lds = S == 1 ? (size_t)(34 * 6 * 4 + 256) * 32 * 4 : (size_t)(65 * 5 * 3 + 64) * 32 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_f32_wgrad16_k<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_f32_wgrad16_k<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_f32_wgrad16_k<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)gx, (unsigned)pairs, (unsigned)x->B);
  coma_set_kernel_tag("conv_f32_wgrad16_k<%d, %d>", S, d->form);
  if (S == 1) hipLaunchKernelGGL((conv_f32_wgrad16_k<1, 0>), grid, dim3(256), lds, s, q);
  else if (d->form == 0) hipLaunchKernelGGL((conv_f32_wgrad16_k<2, 0>), grid, dim3(256), lds, s, q);
  else hipLaunchKernelGGL((conv_f32_wgrad16_k<2, 1>), grid, dim3(256), lds, s, q);
  COMA_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================
// conv_bf16_wgrad16_k -- bf16 weight gradient of >= 32-channel 3x3x3 layers with a coarse width >= 32 (stride 2 and
// transposed: the layers the gather-style conv_mfma_wgrad_k ran at 150-200 TFLOP/s), voxels along K on
// v_mfma_f32_16x16x32_bf16 as in conv_thin16_wgrad_k / conv_f32_wgrad16_k: a block owns a 32 x 32 weight tile pair, wave w
// the 27 taps of one 16 x 16 sub-tile (108 accumulator registers), one dense row of 32 voxels per MFMA.  Both operands sit
// voxel-major in LDS as two 16-channel planes with 32-byte rows (so that the 8 rows a 32-lane half reads transposed are
// conflict-free) and are fetched with ds_read_b64_tr_b16; a stride-2 gathered operand just doubles the row step in the
// lanes' addresses.  The whole (row, tap) sequence of a tile is unrolled; gathered fragments run RING MFMAs ahead.
// =====================================================================================
struct B16W16P {
  const bf16_t* dn; int ldd; long sbd; int Mz, My, Mx;
  const bf16_t* ga; int ldg; long sbg; int Gz, Gy, Gx;
  int N, C;
  unsigned dbytes, gbytes;
  int ntx, nty, ntz, ids_total, ids_per_block, cblocks;
  float* dwk; long wsb;
};

template <int S, int FORM>      // (108 accumulators + 36 ring + 68 staged-piece registers: one block per CU)
__global__ __launch_bounds__(256, 1) void conv_bf16_wgrad16_k(B16W16P p) {
  constexpr int TX = 32, TY = S == 1 ? 4 : 2, TZ = S == 1 ? 2 : 1, TM = TX * TY * TZ, NR = TY * TZ;
  constexpr int HX = (TX - 1) * S + 3, HY = (TY - 1) * S + 3, HZ = (TZ - 1) * S + 3, HV = HX * HY * HZ;
  constexpr int HIT = (HV * 4 + 255) / 256, DIT = (TM * 4 + 255) / 256, NIT = HIT + DIT;      // 16-byte pieces per thread
  constexpr int GPL = HV * 32, DPL = TM * 32;           // bytes per 16-channel plane
  constexpr int RING = 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;                                      // gathered: [2 planes][HV][16 ch]
  char* Dl = smem + 2 * GPL;                            // dense:    [2 planes][TM][16 ch]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int n0 = (blockIdx.y / p.cblocks) * 32, c0 = (blockIdx.y % p.cblocks) * 32;
  const int lv = lane & 15, lg = lane >> 4;
  const bf16_t* dnb = p.dn + (long)b * p.sbd + (FORM == 0 ? n0 : c0);
  const bf16_t* gab = p.ga + (long)b * p.sbg + (FORM == 0 ? c0 : n0);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(gab), 0, p.gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dnb), 0, p.dbytes, 0x00020000);
  constexpr unsigned OOB = 0x7fff0000u;

  // staging: piece = tid + 256 it -> (row = piece >> 2, 8-channel chunk = tid & 3); LDS slot: plane chunk >> 1, half chunk & 1
  const unsigned choff = (unsigned)((tid & 3) * 16);
  const int lds_ch = ((tid & 3) >> 1), lds_hf = (tid & 1) * 16;
  int s_pos[NIT];                                      // packed z | y << 4 | x << 8 (relative to the tile origin - 1)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it < HIT) {
      const int row = (tid + 256 * it) >> 2;
      const int hx = row % HX, hy = (row / HX) % HY, hz = row / (HX * HY);
      s_pos[it] = row < HV ? (hz | (hy << 4) | (hx << 8)) : (15 | (15 << 4) | (1023 << 8));
    } else {
      const int row = (tid + 256 * (it - HIT)) >> 2;
      s_pos[it] = row < TM ? (((row / (TY * 32)) + 1) | ((((row >> 5) % TY) + 1) << 4) | (((row & 31) + 1) << 8)) : (15 | (15 << 4) | (1023 << 8));
    }
  }
  uint4 sreg[NIT];
  auto issue = [&](int it, int z0, int y0, int x0, const __amdgpu_buffer_rsrc_t& rg, const __amdgpu_buffer_rsrc_t& rd) __attribute__((always_inline)) {
    const int sc = it < HIT ? S : 1;
    const int gz = sc * z0 - 1 + (s_pos[it] & 15), gy = sc * y0 - 1 + ((s_pos[it] >> 4) & 15), gx = sc * x0 - 1 + (s_pos[it] >> 8);
    const bool ok = it < HIT ? ((unsigned)gz < (unsigned)p.Gz && (unsigned)gy < (unsigned)p.Gy && (unsigned)gx < (unsigned)p.Gx)
                             : ((unsigned)gz < (unsigned)p.Mz && (unsigned)gy < (unsigned)p.My && (unsigned)gx < (unsigned)p.Mx);
    const unsigned off = it < HIT ? (unsigned)(((gz * p.Gy + gy) * p.Gx + gx) * p.ldg) * 2u + choff
                                  : (unsigned)(((gz * p.My + gy) * p.Mx + gx) * p.ldd) * 2u + choff;
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(it < HIT ? rg : rd, ok ? off : OOB, 0, 0);
    sreg[it] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it < HIT) {
        const int row = (tid + 256 * it) >> 2;
        if (row < HV) *reinterpret_cast<uint4*>(Hl + lds_ch * GPL + row * 32 + lds_hf) = sreg[it];
      } else {
        const int row = (tid + 256 * (it - HIT)) >> 2;
        if (row < TM) *reinterpret_cast<uint4*>(Dl + lds_ch * DPL + row * 32 + lds_hf) = sreg[it];
      }
    }
  };

  const int nh = wid & 1, ch = wid >> 1;               // this wave's 16 x 16 sub-tile (n half, c half)
  // transposed-read addresses: lane 4q+p of group lg supplies voxel x = 4 lg + q (second read: + 16), columns 4p..4p+3
  const int q4 = lv >> 2, p4 = lv & 3;
  const int vx = 4 * lg + q4;
  const int d_addr = (FORM == 0 ? nh : ch) * DPL + vx * 32 + 8 * p4;              // + dense row r * 32 * 32
  const int g_addr = (FORM == 0 ? ch : nh) * GPL + S * vx * 32 + 8 * p4;          // + gathered row offset + tap offset
  f32x4_t acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int id_begin = xcd_remap(blockIdx.x, gridDim.x) * p.ids_per_block;
  int id_end = id_begin + p.ids_per_block;
  if (id_end > p.ids_total) id_end = p.ids_total;
  int id = id_begin, tix = 0, tiy = 0, tiz = 0;
  while (id < id_end && !tile_coords(id, p.ntx, p.nty, p.ntz, tix, tiy, tiz)) ++id;
  if (id < id_end) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue(it, tiz * TZ, tiy * TY, tix * TX, rs_g, rs_d);
  }
  while (id < id_end) {
    int nid = id + 1, ntix = 0, ntiy = 0, ntiz = 0;
    while (nid < id_end && !tile_coords(nid, p.ntx, p.nty, p.ntz, ntix, ntiy, ntiz)) ++nid;
    const bool has_next = nid < id_end;
    const __amdgpu_buffer_rsrc_t rn_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(gab), 0, has_next ? p.gbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dnb), 0, has_next ? p.dbytes : 0, 0x00020000);
    __syncthreads();
    store_tile();
    __syncthreads();
    // One dense row (32 voxels) = one K step = 27 MFMAs (unrolled); the rows are a run-time loop.  Gathered fragments run
    // RING MFMAs ahead, across the row boundary (RING divides 27: the ring slot of a tap is a compile-time constant); the
    // next tile's staging pieces are issued under the first row (peeled), one per MFMA.
    auto rdg = [&](int goff, int t) __attribute__((always_inline)) -> bf16x8_t {
      const char* src = Hl + g_addr + goff + (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * 32;
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src + S * 16 * 32));
      return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto rdd = [&](int r) __attribute__((always_inline)) -> bf16x8_t {
      const char* src = Dl + d_addr + r * 32 * 32;
      const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src));
      const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(src + 16 * 32));
      return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto growoff = [&](int r) __attribute__((always_inline)) { return ((((r / TY) * S) * HY + (r % TY) * S) * HX) * 32; };
    bf16x8_t dcur = rdd(0), gring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) gring[i] = rdg(0, i);
#define COMA_W16_ROW(R_, FIRST_)                                                                                          \
    {                                                                                                                     \
      const int r_ = (R_), rn_ = r_ + 1 < NR ? r_ + 1 : r_;                                                               \
      const int goff_r = growoff(r_), goff_n = growoff(rn_);                                                              \
      bf16x8_t dnext = dcur;                                                                                              \
      _Pragma("unroll") for (int t = 0; t < 27; ++t) {                                                                    \
        const bf16x8_t gcur = gring[t % RING];                                                                            \
        gring[t % RING] = t + RING < 27 ? rdg(goff_r, t + RING) : rdg(goff_n, t + RING - 27);                             \
        if (t == 0) dnext = rdd(rn_);                                                                                     \
        if (FIRST_ && t < NIT) issue(t, ntiz * TZ, ntiy * TY, ntix * TX, rn_g, rn_d);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        acc[t] = FORM == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(dcur, gcur, acc[t], 0, 0, 0)                         \
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(gcur, dcur, acc[t], 0, 0, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
      }                                                                                                                   \
      dcur = dnext;                                                                                                       \
    }
    static_assert(NIT <= 27 && 27 % RING == 0, "staging pieces under the first row; ring slots compile-time");
    COMA_W16_ROW(0, true)
#pragma unroll 1
    for (int r = 1; r < NR; ++r) COMA_W16_ROW(r, false)
#undef COMA_W16_ROW
    id = nid; tix = ntix; tiy = ntiy; tiz = ntiz;
  }
  float* wout = p.dwk + (long)b * p.wsb;
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * nh + 4 * lg + j, c = c0 + 16 * ch + lv;
      atomicAdd(wout + ((long)t * p.N + n) * p.C + c, acc[t][j]);
    }
}

static bool bf16_wgrad16_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  static const int mode = []{ const char* e = getenv("COMA_B16W16"); return e ? atoi(e) : 1; }();   // 0 off, 1 stride-2 / transposed, 2 also stride 1
  const coma_tensor* dn = d->form == 0 ? dy : x;
  const bool shape = d->ksize == 3 && ((d->stride == 2 && (d->form == 0 || d->form == 1)) || (mode >= 2 && d->form == 0 && d->stride == 1));
  return mode > 0 && shape && x->dtype == COMA_BF16 && dy->dtype == COMA_BF16 && dn->W >= 32 &&
         x->C % 32 == 0 && dy->C % 32 == 0 && x->ld % 8 == 0 && x->sb % 8 == 0 && dy->ld % 8 == 0 && dy->sb % 8 == 0 &&
         (!x->data || aligned16(x->data)) && (!dy->data || aligned16(dy->data)) &&
         (unsigned long long)t_vox(x) * x->ld * 2 < 0x7fff0000ull && (unsigned long long)t_vox(dy) * dy->ld * 2 < 0x7fff0000ull;
}

static int conv_bf16_wgrad16(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, hipStream_t s, int zeroed) {
  const coma_tensor* dn = d->form == 0 ? dy : x;
  const coma_tensor* ga = d->form == 0 ? x : dy;
  B16W16P q;
  q.dn = (const bf16_t*)dn->data; q.ldd = (int)dn->ld; q.sbd = dn->sb; q.Mz = dn->D; q.My = dn->H; q.Mx = dn->W;
  q.ga = (const bf16_t*)ga->data; q.ldg = (int)ga->ld; q.sbg = ga->sb; q.Gz = ga->D; q.Gy = ga->H; q.Gx = ga->W;
  q.N = dy->C; q.C = x->C;
  q.dbytes = (unsigned)((unsigned long long)t_vox(dn) * dn->ld * 2);
  q.gbytes = (unsigned)((unsigned long long)t_vox(ga) * ga->ld * 2);
  const int S = d->stride, tz = S == 1 ? 2 : 1, ty = S == 1 ? 4 : 2;
  q.ntx = (q.Mx + 31) / 32; q.nty = (q.My + ty - 1) / ty; q.ntz = (q.Mz + tz - 1) / tz;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  q.cblocks = q.C / 32;
  const int pairs = q.cblocks * (q.N / 32);
  int gx = 256 / (pairs * x->B);
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  const long wsz1 = 27L * q.N * q.C, wsz = wsz1 * (d->per_sample_w ? x->B : 1);
  q.wsb = d->per_sample_w ? wsz1 : 0;
  q.dwk = dwk;
  if (!(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dwk, 0, sizeof(float) * wsz, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const size_t lds = S == 1 ? (size_t)(34 * 6 * 4 + 256) * 64 : (size_t)(65 * 5 * 3 + 64) * 64;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_bf16_wgrad16_k<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_wgrad16_k<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bf16_wgrad16_k<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)gx, (unsigned)pairs, (unsigned)x->B);
  coma_set_kernel_tag("conv_bf16_wgrad16_k<%d, %d>", S, d->form);
  if (S == 1) hipLaunchKernelGGL((conv_bf16_wgrad16_k<1, 0>), grid, dim3(256), lds, s, q);
  else if (d->form == 0) hipLaunchKernelGGL((conv_bf16_wgrad16_k<2, 0>), grid, dim3(256), lds, s, q);
  else hipLaunchKernelGGL((conv_bf16_wgrad16_k<2, 1>), grid, dim3(256), lds, s, q);
  COMA_LAUNCH_CHECK();
  return 0;
}

static bool thin16f_wgrad_ok(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  static const bool on = []{ const char* e = getenv("COMA_THIN16F"); return !(e && e[0] == '0'); }();
  const int nmax = x->C <= 4 ? 32 : 16;
  return on && d->form == 0 && d->ksize == 3 && d->stride == 1 && x->dtype == COMA_F32 && dy->dtype == COMA_F32 && x->W >= 32 &&
         x->C <= 16 && dy->C <= nmax && x->ld % 4 == 0 && x->sb % 4 == 0 && dy->ld % 4 == 0 && dy->sb % 4 == 0 &&
         (!x->data || aligned16(x->data)) && (!dy->data || aligned16(dy->data)) &&
         (unsigned long long)t_vox(x) * x->ld * 4 < 0x7fff0000ull && (unsigned long long)t_vox(dy) * dy->ld * 4 < 0x7fff0000ull;
}

static int conv_thin16f_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, void* ws,
                              size_t ws_bytes, hipStream_t s, int zeroed) {
  Thin16FWP q;
  q.x = (const float*)x->data; q.ldx = (int)x->ld; q.sbx = x->sb; q.D = x->D; q.H = x->H; q.W = x->W; q.C = x->C;
  q.dy = (const float*)dy->data; q.ldn = (int)dy->ld; q.sbn = dy->sb; q.N = dy->C;
  q.xbytes = (unsigned)((unsigned long long)t_vox(x) * x->ld * 4);
  q.dbytes = (unsigned)((unsigned long long)t_vox(dy) * dy->ld * 4);
  q.ntx = (q.W + 31) / 32; q.nty = (q.H + 3) / 4; q.ntz = (q.D + 1) / 2;
  q.ids_total = q.ntx * q.nty * ((q.ntz + 7) / 8) * 8;
  int gx = (q.C > 8 ? 256 : 512) / x->B;
  if (gx < 1) gx = 1;
  if (gx > q.ids_total) gx = q.ids_total;
  q.ids_per_block = (q.ids_total + gx - 1) / gx;
  gx = (q.ids_total + q.ids_per_block - 1) / q.ids_per_block;
  const long wsz1 = 27L * q.N * q.C, wsz = wsz1 * (d->per_sample_w ? x->B : 1);
  q.wsb = d->per_sample_w ? wsz1 : 0;
  const bool replicas = wsz <= WGRAD_REP_MAX_ELEMS && ws && ws_bytes >= sizeof(float) * wsz * WGRAD_NREP;
  q.nrep = replicas ? WGRAD_NREP : 1;
  q.rep_stride = replicas ? wsz : 0;
  q.dwk = replicas ? (float*)ws : dwk;
  if (!(zeroed & (replicas ? COMA_ZEROED_WS : COMA_ZEROED_OUT)) && hipMemsetAsync(q.dwk, 0, sizeof(float) * wsz * q.nrep, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const int cp = q.C > 8 ? 16 : q.C > 4 ? 8 : 4, nb = q.N > 16 ? 2 : 1;
  const int nm = (27 + 16 / cp - 1) / (16 / cp);
  size_t lds = (size_t)34 * 6 * 4 * cp * 4 + (size_t)256 * nb * 16 * 4;
  if (lds < (size_t)nm * nb * 256 * 4) lds = (size_t)nm * nb * 256 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_thin16f_wgrad_k<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_wgrad_k<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_wgrad_k<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_thin16f_wgrad_k<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)gx, 1, (unsigned)x->B);
  coma_set_kernel_tag("conv_thin16f_wgrad_k<%d, %d>", cp, nb);
  if (cp == 16) hipLaunchKernelGGL((conv_thin16f_wgrad_k<16, 1>), grid, dim3(256), lds, s, q);
  else if (cp == 8) hipLaunchKernelGGL((conv_thin16f_wgrad_k<8, 1>), grid, dim3(256), lds, s, q);
  else if (nb == 2) hipLaunchKernelGGL((conv_thin16f_wgrad_k<4, 2>), grid, dim3(256), lds, s, q);
  else hipLaunchKernelGGL((conv_thin16f_wgrad_k<4, 1>), grid, dim3(256), lds, s, q);
  COMA_LAUNCH_CHECK();
  if (replicas) {
    hipLaunchKernelGGL(wgrad_replica_sum_k, dim3((unsigned)((wsz + 31) / 32)), dim3(256), 0, s, (const float*)ws, WGRAD_NREP, wsz, dwk);
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

static int conv_mfma_wgrad2(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, void* ws,
                            size_t ws_bytes, hipStream_t s, int zeroed) {
  Wgrad2P p;
  p.dyp = (const bf16_t*)dy->data; p.ldn = (int)dy->ld; p.sbn = dy->sb;
  p.xp = (const bf16_t*)x->data; p.ldc = (int)x->ld; p.sbc = x->sb;
  p.N = dy->C; p.C = x->C; p.D = x->D; p.H = x->H; p.W = x->W;
  p.ntx = (p.W + 31) / 32; p.nty = (p.H + 3) / 4; p.ntz = (p.D + 1) / 2;
  p.tiles_total = p.ntx * p.nty * ((p.ntz + 7) / 8) * 8;
  p.cblocks = (p.C + 31) / 32;
  {   // live 8-channel chunks per staged row, rounded up to 1 / 2 / 4 (blocks of a wide operand always see 4)
    const int nd = p.N > 32 ? 4 : (p.N + 7) / 8, ng = p.C > 32 ? 4 : (p.C + 7) / 8;
    p.lgd = nd <= 1 ? 0 : nd <= 2 ? 1 : 2; p.lgg = ng <= 1 ? 0 : ng <= 2 ? 1 : 2;
  }
  const int pairs = ((p.N + 31) / 32) * p.cblocks;
  p.vec_n = dy->ld % 8 == 0 && dy->sb % 8 == 0 && (((uintptr_t)dy->data) & 15) == 0;
  p.vec_c = x->ld % 8 == 0 && x->sb % 8 == 0 && (((uintptr_t)x->data) & 15) == 0;
  // 512 blocks = the 2 resident blocks per CU, once: every block ends with a 27 x 32 x 32 fp32 atomic merge, and a second
  // round of blocks doubled that traffic (64 -> 64 at 64^3: 215 -> 156 us, 128 -> 128 at 32^3: 136 -> 101 us)
  int chunks = 512 / (pairs * x->B);
  if (chunks < 1) chunks = 1;
  if (chunks > p.tiles_total) chunks = p.tiles_total;
  p.tiles_per_block = (p.tiles_total + chunks - 1) / chunks;
  chunks = (p.tiles_total + p.tiles_per_block - 1) / p.tiles_per_block;
  const long taps = (long)d->ksize * d->ksize * d->ksize;
  p.dwk = dwk; p.wsb = d->per_sample_w ? taps * p.N * p.C : 0;
  const long wsz = taps * p.N * p.C * (d->per_sample_w ? x->B : 1);
  const bool replicas = wsz <= WGRAD_REP_MAX_ELEMS && ws && ws_bytes >= sizeof(float) * wsz * WGRAD_NREP;
  p.nrep = replicas ? WGRAD_NREP : 1;
  p.rep_stride = replicas ? wsz : 0;
  if (replicas) p.dwk = (float*)ws;
  if (!(zeroed & (replicas ? COMA_ZEROED_WS : COMA_ZEROED_OUT)) && hipMemsetAsync(p.dwk, 0, sizeof(float) * wsz * p.nrep, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  const size_t lds = (size_t)(256 + (d->ksize == 3 ? 816 + 1 : 256)) * 64;   // +1 row: the unused packed column reads one voxel past the halo
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<0, 2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<1, 2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<0, 2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<1, 2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<0, 2, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad2_k<1, 2, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const bool vec = p.vec_n && p.vec_c && p.N % 8 == 0 && p.C % 8 == 0;
  const dim3 grid((unsigned)chunks, (unsigned)pairs, (unsigned)x->B);
  coma_set_kernel_tag("conv_mfma_wgrad2_k<%d, 2, %d, %d>", (int)vec, d->ksize, d->ksize == 1 ? 1 : p.C <= 8 ? 4 : p.C <= 16 ? 2 : 1);
#define WG2(K_, PK_) do { if (vec) hipLaunchKernelGGL((conv_mfma_wgrad2_k<1, 2, K_, PK_>), grid, dim3(256), lds, s, p); \
                          else hipLaunchKernelGGL((conv_mfma_wgrad2_k<0, 2, K_, PK_>), grid, dim3(256), lds, s, p); } while (0)
  if (d->ksize == 1) WG2(1, 1);
  else if (p.C <= 8) WG2(3, 4);
  else if (p.C <= 16) WG2(3, 2);
  else WG2(3, 1);
#undef WG2
  COMA_LAUNCH_CHECK();
  if (replicas) {
    hipLaunchKernelGGL(wgrad_replica_sum_k, dim3((unsigned)((wsz + 31) / 32)), dim3(256), 0, s, (const float*)ws, WGRAD_NREP, wsz, dwk);
    COMA_LAUNCH_CHECK();
  }
  return 0;
}

static int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct WgradPlan { WgradP2 p; int TM; size_t lds; int tn, tc; dim3 grid; bool ok; };

static WgradPlan wgrad_plan(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  WgradPlan pl{};
  pl.ok = false;
  const bool f32 = x->dtype == COMA_F32 && dy->dtype == COMA_F32;
  if (!f32 && (x->dtype != COMA_BF16 || dy->dtype != COMA_BF16)) return pl;
  if (d->ksize != 3 && d->ksize != 1) return pl;
  if ((long)t_vox(x) * x->ld >= (1L << 31) || (long)t_vox(dy) * dy->ld >= (1L << 31)) return pl;
  if (d->form == 1 && d->stride != 2) return pl;
  WgradP2& p = pl.p;
  p.dyp = dy->data; p.ldn = (int)dy->ld; p.sbn = dy->sb;
  p.xp = x->data; p.ldc = (int)x->ld; p.sbc = x->sb;
  p.N = dy->C; p.C = x->C;
  const coma_tensor* dn = d->form == 0 ? dy : x;     // dense
  const coma_tensor* ga = d->form == 0 ? x : dy;     // gathered
  p.Mz = dn->D; p.My = dn->H; p.Mx = dn->W; p.Gz = ga->D; p.Gy = ga->H; p.Gx = ga->W;
  p.k = d->ksize; p.stride = d->stride; p.pad = d->pad;
  const int vq = f32 ? 4 : 8;                        // elements per 16-byte piece
  p.vec_n = dy->ld % vq == 0 && dy->sb % vq == 0 && (!dy->data || aligned16(dy->data));
  p.vec_c = x->ld % vq == 0 && x->sb % vq == 0 && (!x->data || aligned16(x->data));
  pl.tn = (dy->C > 32 && !(x->C > 32 && x->C > dy->C)) ? 2 : 1;
  pl.tc = (pl.tn == 1 && x->C > 32) ? 2 : 1;
  if (f32) pl.tn = pl.tc = 1;
  const int gch = d->form == 0 ? x->C : dy->C;
  // bytes per element, channels per piece, 16-byte staging pieces per thread
  const int esz = f32 ? 4 : 2, ppc = f32 ? 4 : 8, maxp = f32 ? ((gch >= 32 && d->ksize == 3) ? 21 : 24) : 20;
  // fp32 kernel: the gathered operand's 32 MFMA indices are (tap, channel) pairs -- cp channels (a power of two) per tap
  p.cp = 32; p.pg = 128;
  if (f32 && gch < 32) { p.cp = 1; while (p.cp < gch) p.cp <<= 1; p.pg = (p.cp < 4 ? 4 : p.cp) * 4; }
  if (f32 && !(p.vec_n && p.vec_c)) return pl;        // fp32 kernel: 16-byte staging loads only
  // tile: up to 256 dense voxels at stride 1, 64 at stride 2 (the halo grows 8x); shrink until the LDS image
  // and the per-thread register staging budget (20 x 16-byte pieces) fit
  const int cdb = 32 * (d->form == 0 ? pl.tn : pl.tc), cgb = f32 ? p.pg / 4 : 32 * (d->form == 0 ? pl.tc : pl.tn);
  bool fits = false;
  for (int budget = d->stride == 1 ? 8 : 6; budget >= 4 && !fits; --budget) {
    int lx = ilog2_ceil(p.Mx); if (lx > 5) lx = 5;
    if (d->stride == 2 && lx > 4) lx = 4;
    if (lx > budget) lx = budget;
    int rem = budget - lx;
    int ly = ilog2_ceil(p.My); if (ly > (rem + 1) / 2) ly = (rem + 1) / 2;
    rem -= ly;
    int lz = ilog2_ceil(p.Mz); if (lz > rem) lz = rem;
    while (lx + ly + lz < 4) ++lx;     // at least one 16-voxel K step
    p.lx = lx; p.ly = ly; p.lz = lz;
    pl.TM = 1 << (lx + ly + lz);
    p.hz = ((1 << lz) - 1) * p.stride + p.k; p.hy = ((1 << ly) - 1) * p.stride + p.k; p.hx = ((1 << lx) - 1) * p.stride + p.k;
    pl.lds = (size_t)pl.TM * cdb * esz + (size_t)p.hz * p.hy * p.hx * cgb * esz + (f32 ? 1024 : 0);
    fits = pl.lds <= 160 * 1024 && pl.TM * (cdb / ppc) + p.hz * p.hy * p.hx * (cgb / ppc) <= 256 * maxp;
  }
  if (!fits) return pl;
  p.ntx = (p.Mx + (1 << p.lx) - 1) >> p.lx; p.nty = (p.My + (1 << p.ly) - 1) >> p.ly; p.ntz = (p.Mz + (1 << p.lz) - 1) >> p.lz;
  p.tiles_total = p.ntx * p.nty * ((p.ntz + 7) / 8) * 8;   // ids incl. z padding (tile_coords)
  p.m_hx = (unsigned)((1ull << 32) / (unsigned)p.hx) + 1u;
  p.m_hxy = (unsigned)((1ull << 32) / (unsigned)(p.hx * p.hy)) + 1u;
  p.cblocks = (p.C + 32 * pl.tc - 1) / (32 * pl.tc);
  const int pairs = ((p.N + 32 * pl.tn - 1) / (32 * pl.tn)) * p.cblocks;
  // one block per CU, ONE round (256 blocks): a second round repeats every block's 27 x n x c fp32 atomic merge
  // (32 -> 64 stride 2 at 128^3: 307 -> 254 us; 256 -> 256 at 16^3: 163 -> 107 us)
  int chunks = 256 / (pairs * x->B);
  // Deepest layers (>= 64 weight tiles, <= 1024 voxels): ONE block per tile and sample.  With per-sample weights (or one sample) every
  // dwk element then has a single producer: plain stores instead of memset + fp32 atomics (512 -> 512 at 8^3: the
  // 28 M atomics of a 2-chunk split cost ~90 of the kernel's 142 us).
  // (measured: at <= 1024 dense voxels a quarter of the CUs busy without atomics beats all of them with; at 4096
  // voxels the single block's MFMA work is the longer pole and the old split wins)
  if (pairs * x->B >= 64 && (long)p.Mz * p.My * p.Mx <= 1024) chunks = 1;
  if (chunks < 1) chunks = 1;
  if (chunks > p.tiles_total) chunks = p.tiles_total;
  p.tiles_per_block = (p.tiles_total + chunks - 1) / chunks;
  chunks = (p.tiles_total + p.tiles_per_block - 1) / p.tiles_per_block;
  p.plain = chunks == 1 && (d->per_sample_w || x->B == 1);
  if (f32 && p.cp < 8 && d->ksize == 3) p.plain = 0;     // fewer than 4 MFMA tiles: several waves of a block sum the same outputs
  if (f32 && d->ksize == 1) p.plain = 0;                  // 1x1x1: the four waves split the voxel pairs of the single tile
  pl.grid = dim3((unsigned)chunks, (unsigned)pairs, (unsigned)x->B);
  const long taps = (long)d->ksize * d->ksize * d->ksize;
  p.wsb = d->per_sample_w ? taps * p.N * p.C : 0;
  pl.ok = true;
  return pl;
}

bool conv_mfma_wgrad_supported(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  return wgrad2_ok(d, x, dy) || thin16f_wgrad_ok(d, x, dy) || wgrad_plan(d, x, dy).ok;      // (bf16 and fp32: the plan checks the dtype pair)
}
size_t conv_mfma_wgrad_ws_bytes(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy) {
  const long wsz = wgrad_out_elems(d, x, dy);
  return (wgrad2_ok(d, x, dy) || x->dtype == COMA_F32) && wsz <= WGRAD_REP_MAX_ELEMS ? sizeof(float) * wsz * WGRAD_NREP : 0;
}

// zeroed: COMA_ZEROED_OUT -- dwk is all zeros on entry; COMA_ZEROED_WS -- the first conv_mfma_wgrad_ws_bytes() bytes of ws are
// (the caller's pre-zeroed arena: one memset per training step instead of one per layer)
int conv_mfma_wgrad(const coma_conv_desc* d, const coma_tensor* x, const coma_tensor* dy, float* dwk, void* ws, size_t ws_bytes,
                    hipStream_t s, int zeroed) {
  if (thin16_wgrad_ok(d, x, dy)) return conv_thin16_wgrad(d, x, dy, dwk, ws, ws_bytes, s, zeroed);
  if (bf16_wgrad16_ok(d, x, dy)) return conv_bf16_wgrad16(d, x, dy, dwk, s, zeroed);
  if (wgrad2_ok(d, x, dy)) return conv_mfma_wgrad2(d, x, dy, dwk, ws, ws_bytes, s, zeroed);
  if (thin16f_wgrad_ok(d, x, dy)) return conv_thin16f_wgrad(d, x, dy, dwk, ws, ws_bytes, s, zeroed);
  if (f32_wgrad16_ok(d, x, dy)) return conv_f32_wgrad16(d, x, dy, dwk, s, zeroed);
  WgradPlan pl = wgrad_plan(d, x, dy);
  COMA_CHECK(pl.ok, "conv_mfma_wgrad: unsupported problem");
  pl.p.dwk = dwk;
  const long wsz = (long)d->ksize * d->ksize * d->ksize * dy->C * x->C * (d->per_sample_w ? x->B : 1);
  if (!pl.p.plain && !(zeroed & COMA_ZEROED_OUT) && hipMemsetAsync(dwk, 0, sizeof(float) * wsz, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
  if (x->dtype == COMA_F32) {
    // small outputs (the 1..16-channel layers): hundreds of blocks merging into a few cache lines serialise in the
    // atomic unit -- they merge into WGRAD_NREP replicas in the workspace, summed by one small kernel (as wgrad2 does)
    const bool replicas = !pl.p.plain && wsz <= WGRAD_REP_MAX_ELEMS && ws && ws_bytes >= sizeof(float) * wsz * WGRAD_NREP;
    pl.p.nrep = replicas ? WGRAD_NREP : 1;
    pl.p.rep_stride = replicas ? wsz : 0;
    if (replicas) {
      pl.p.dwk = (float*)ws;
      if (!(zeroed & COMA_ZEROED_WS) && hipMemsetAsync(ws, 0, sizeof(float) * wsz * WGRAD_NREP, s) != hipSuccess) { coma_set_error("wgrad memset failed"); return 2; }
    }
    const int taps = d->ksize * d->ksize * d->ksize, tpt = 32 / pl.p.cp, ntiles = (taps + tpt - 1) / tpt;
    const int nt = (ntiles + 3) / 4;                  // tiles per wave: 27 -> 7, 14 -> 4, 7 -> 2, <= 4 -> 1
    static bool attr = false;
    if (!attr) {
#define F32ATTR(F, N_) (void)hipFuncSetAttribute((const void*)conv_f32_wgrad_k<F, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      F32ATTR(0, 1); F32ATTR(0, 2); F32ATTR(0, 4); F32ATTR(0, 7); F32ATTR(1, 1); F32ATTR(1, 2); F32ATTR(1, 4); F32ATTR(1, 7);
#undef F32ATTR
      attr = true;
    }
    coma_set_kernel_tag("conv_f32_wgrad_k<%d, %d>", d->form, nt == 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 7);
#define F32WL(F) do { if (nt == 1) hipLaunchKernelGGL((conv_f32_wgrad_k<F, 1>), pl.grid, dim3(256), pl.lds, s, pl.p); \
                      else if (nt == 2) hipLaunchKernelGGL((conv_f32_wgrad_k<F, 2>), pl.grid, dim3(256), pl.lds, s, pl.p); \
                      else if (nt <= 4) hipLaunchKernelGGL((conv_f32_wgrad_k<F, 4>), pl.grid, dim3(256), pl.lds, s, pl.p); \
                      else hipLaunchKernelGGL((conv_f32_wgrad_k<F, 7>), pl.grid, dim3(256), pl.lds, s, pl.p); } while (0)
    if (d->form == 0) F32WL(0); else F32WL(1);
#undef F32WL
    COMA_LAUNCH_CHECK();
    if (replicas) {
      hipLaunchKernelGGL(wgrad_replica_sum_k, dim3((unsigned)((wsz + 31) / 32)), dim3(256), 0, s, (const float*)ws, WGRAD_NREP, wsz, dwk);
      COMA_LAUNCH_CHECK();
    }
    return 0;
  }
  const bool vecall = pl.p.vec_n && pl.p.vec_c && dy->C % 8 == 0 && x->C % 8 == 0;
#define WL(TNV, TCV, F)                                                                                          \
  do {                                                                                                           \
    if (vecall) {                                                                                                \
      (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad_k<TNV, TCV, F, 1>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      hipLaunchKernelGGL((conv_mfma_wgrad_k<TNV, TCV, F, 1>), pl.grid, dim3(256), pl.lds, s, pl.p);              \
    } else {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)conv_mfma_wgrad_k<TNV, TCV, F, 0>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      hipLaunchKernelGGL((conv_mfma_wgrad_k<TNV, TCV, F, 0>), pl.grid, dim3(256), pl.lds, s, pl.p);              \
    }                                                                                                            \
  } while (0)
  coma_set_kernel_tag("conv_mfma_wgrad_k<%d, %d, %d, %d>", pl.tn, pl.tc, d->form, (int)vecall);
  if (d->form == 0) {
    if (pl.tn == 2) WL(2, 1, 0); else if (pl.tc == 2) WL(1, 2, 0); else WL(1, 1, 0);
  } else {
    if (pl.tn == 2) WL(2, 1, 1); else if (pl.tc == 2) WL(1, 2, 1); else WL(1, 1, 1);
  }
#undef WL
  COMA_LAUNCH_CHECK();
  return 0;
}

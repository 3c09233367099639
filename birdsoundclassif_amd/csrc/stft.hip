// STFT magnitude in dB on the gfx950 fp64 matrix core (v_mfma_f64_16x16x4_f64).
//
// Replaces librosa.stft + np.abs + amp_to_db + the row crop of File_Processor.spectrogram
// (reference nbm_model/nbm_datasets/prepare_dataset.py:228-247).
//
// Why fp64: the image is 20 log10 |X| of bins that sit 80-100 dB under the loudest bin of the same frame; a 1324-term fp32
// accumulation (and an fp32 basis) leaves an ABSOLUTE error of ~1e-5 max|X|, i.e. a dB error of 0.1 exactly in those rows
// (measured round 1: image error up to 1e-3).  The waveform is exact in fp32 (PCM), the basis is built in float64, the
// products and sums below are fp64: the image error drops to the fp32 rounding of the final dB value (< 1e-6).
//
// Why it is still cheap: the frame is REAL and the periodic Hann window is symmetric (w[n] = w[N-n], w[0] = 0), so with
//   s[k] = x[k] + x[N-k],  d[k] = x[k] - x[N-k]        (k = 0 .. N/2; formed in fp64 from the fp32 samples: exact)
//   Re X[f] = sum_k s[k] * c[f][k],   c[f][k] = w[k] cos(2 pi f k / N)   (c[f][N/2] halved: s[N/2] = 2 x[N/2])
//   Im X[f] = sum_k d[k] * z[f][k],   z[f][k] = w[k] sin(2 pi f k / N)
// the two GEMMs run over K = N/2 + 1 = 663 (padded to 664) instead of 1324: 1.04 GFLOP per 3 s clip, 0.85 ms per 64
// clips at the fp64 MFMA rate (half the fp32 rate on this part).
//
// Data flow per workgroup (4 waves): 64 consecutive frames x 128 bins.  The 64 frames overlap (hop 132 of 1324 samples),
// so their samples are ONE contiguous 9 640-float span of the waveform: it is copied to LDS once (38.6 KB) and every
// K-step reads its operands from there (ds_read_b32 at t * hop + k and t * hop + N - k: 2-way bank conflicts, nothing next
// to a 64-cycle MFMA); no frame matrix exists anywhere.  The basis is stored in MFMA fragment order ([bin tile][k step]
// [lane] {cos, sin}), so a wave fetches both A operands of a bin tile with one coalesced 16-byte-per-lane load straight
// into registers (L2-resident: 4 MB shared by every workgroup).  Each wave owns BR x FC = 2 x 4 tiles of 16 bins x 16
// frames, real and imaginary part: 16 independent accumulators (128 VGPRs).
//
// Epilogue: |X| = sqrt(re^2 + im^2) in fp64 -> fp32 -> 20 log10f(max(floor, .)), stores along time, wave-reduced
// min / max -> two atomics per wave on the per-row order-preserving keys.
#include "nbm_common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct StftParams {
  const float* wave; long long wave_ld;
  const f64x2* basis;            // [bin tiles][n_ksteps][64] {cos, sin}
  float* db; long long db_bs; int db_ld;
  uint32_t* minmax;
  int n_frames, hop, n_fft, n_ksteps, n_bins, seg_n;
  float floor_amp;
};

template <int BR, int FC>
__global__ __launch_bounds__(256, 2) void stft_f64_kernel(const StftParams p) {
  extern __shared__ __attribute__((aligned(16))) float seg[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t0 = blockIdx.x * (16 * FC), b = blockIdx.z;

  // ---- the samples of frames t0 .. t0 + 16 FC - 1 (+ x[N] of the last one) -> LDS; zeros beyond the end of the row
  {
    const long long off0 = (long long)t0 * p.hop;
    const float* __restrict__ src = p.wave + (long long)b * p.wave_ld + off0;
    const long long avail = p.wave_ld - off0;
    for (int i = tid * 4; i < p.seg_n; i += 1024) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i + 3 < avail) {
        v = *reinterpret_cast<const f32x4*>(src + i);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i + e < avail) v[e] = src[i + e];
      }
      *reinterpret_cast<f32x4*>(seg + i) = v;
    }
  }
  __syncthreads();

  const int bt0 = (blockIdx.y * 4 + wv) * BR;           // first bin tile of this wave
  if (bt0 * 16 >= p.n_bins) return;                      // no barrier below
  const int tl = lane & 15, kq = lane >> 4;

  const float* fwd[FC];
  const float* mir[FC];
#pragma unroll
  for (int j = 0; j < FC; ++j) {
    fwd[j] = seg + (j * 16 + tl) * p.hop + kq;
    mir[j] = seg + (j * 16 + tl) * p.hop + p.n_fft - kq;
  }
  const f64x2* __restrict__ ap[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) ap[i] = p.basis + ((long long)(bt0 + i) * p.n_ksteps) * 64 + lane;

  f64x4 re[BR][FC], im[BR][FC];
#pragma unroll
  for (int i = 0; i < BR; ++i)
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      re[i][j] = f64x4{0., 0., 0., 0.};
      im[i][j] = f64x4{0., 0., 0., 0.};
    }

  // Software pipeline: the operands of step ks + 1 (basis fragments from L2, samples from LDS, fp64 sum / difference)
  // are fetched and formed while the 16 MFMAs of step ks run.
  f64x2 a_cur[BR], a_nxt[BR];
  double s[FC], d[FC];
#pragma unroll
  for (int i = 0; i < BR; ++i) a_cur[i] = ap[i][0];
#pragma unroll
  for (int j = 0; j < FC; ++j) {
    const double xa = (double)fwd[j][0], xb = (double)mir[j][0];
    s[j] = xa + xb;
    d[j] = xa - xb;
  }
  for (int ks = 0; ks < p.n_ksteps; ++ks) {
    const int kn = ks + 1 < p.n_ksteps ? ks + 1 : ks;
    float xa[FC], xb[FC];
#pragma unroll
    for (int i = 0; i < BR; ++i) a_nxt[i] = ap[i][(long long)kn * 64];
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      xa[j] = fwd[j][4 * kn];
      xb[j] = mir[j][-4 * kn];
    }
#pragma unroll
    for (int i = 0; i < BR; ++i)
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i][0], s[j], re[i][j], 0, 0, 0);
        im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i][1], d[j], im[i][j], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      const double da = (double)xa[j], db = (double)xb[j];
      s[j] = da + db;
      d[j] = da - db;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) a_cur[i] = a_nxt[i];
    // issue order inside the step: the wave is in-order, so anything placed behind the 16 back-to-back MFMAs would wait
    // for all of them -- spread the LDS reads over the first MFMAs and the conversions / adds over the remaining ones
    __builtin_amdgcn_sched_group_barrier(0x020, BR, 0);
#pragma unroll
    for (int z = 0; z < 2 * FC; ++z) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int z = 0; z < 2 * BR * FC - 2 * FC; ++z) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (4 * FC + 2 * BR * FC - 2 * FC - 1) / (2 * BR * FC - 2 * FC), 0);
    }
  }

  // ---- epilogue.  f64 16x16x4 C/D layout: col = lane & 15 (frame), row = (lane >> 4) + 4 reg (bin)
  float* __restrict__ yg = p.db + (long long)b * p.db_bs;
  float vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
  for (int i = 0; i < BR; ++i)
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      const int t = t0 + j * 16 + tl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bin = (bt0 + i) * 16 + kq + 4 * r;
        const double x = re[i][j][r], y = im[i][j][r];
        const float mag = (float)sqrt(x * x + y * y);
        const float v = 20.0f * log10f(fmaxf(p.floor_amp, mag));
        if (bin < p.n_bins && t < p.n_frames) {
          yg[(long long)bin * p.db_ld + t] = v;
          vmin = fminf(vmin, v);
          vmax = fmaxf(vmax, v);
        }
      }
    }
  vmin = nbm_wave_min(vmin);
  vmax = nbm_wave_max(vmax);
  if (lane == 0 && vmin <= vmax) {
    atomicMin(p.minmax + 2 * b, nbm_f2key(vmin));
    atomicMax(p.minmax + 2 * b + 1, nbm_f2key(vmax));
  }
}

}  // namespace

// db[b][f][t] = 20 log10(max(floor, |STFT|)) -- see nbm_hip.h.
extern "C" int nbm_stft_db(const float* wave, int64_t wave_ld, int batch, int n_frames, int hop, int n_fft,
                           const double* basis, int basis_bin_tiles, int basis_ksteps, int n_bins, float floor_amp,
                           float* db, int64_t db_bs, int db_ld, uint32_t* minmax, void* stream) {
  constexpr int BR = 2, FC = 4;
  if (!wave || !basis || !db || !minmax || batch <= 0 || n_frames <= 0 || hop <= 0 || n_fft <= 0 || n_bins <= 0)
    return NBM_EINVAL;
  if ((n_fft & 1) || basis_ksteps * 4 < n_fft / 2 + 1 || db_ld < n_frames) return NBM_EINVAL;
  if (basis_bin_tiles % (4 * BR) || basis_bin_tiles * 16 < n_bins) return NBM_EINVAL;
  if ((hop & 3) || (wave_ld & 3) || !nbm_aligned16(wave) || !nbm_aligned16(basis)) return NBM_EALIGN;
  // the last frame must stay inside its row (x[N] of it is read too, but weighted with w[0] = 0: zeros are fine there)
  if ((int64_t)(n_frames - 1) * hop + n_fft > wave_ld) return NBM_EINVAL;
  StftParams p{};
  p.wave = wave; p.wave_ld = wave_ld; p.basis = reinterpret_cast<const f64x2*>(basis);
  p.db = db; p.db_bs = db_bs; p.db_ld = db_ld; p.minmax = minmax;
  p.n_frames = n_frames; p.hop = hop; p.n_fft = n_fft; p.n_ksteps = basis_ksteps; p.n_bins = n_bins;
  p.floor_amp = floor_amp;
  // samples a workgroup touches: frames 0 .. 16 FC - 1, k up to 4 n_ksteps - 1 forwards, N - k >= ... backwards, and x[N]
  const int kmax = 4 * basis_ksteps - 1;
  int span = (16 * FC - 1) * hop + (kmax > n_fft ? kmax : n_fft) + 1;
  p.seg_n = (span + 3) & ~3;
  if (kmax > n_fft) return NBM_EINVAL;                    // mirrored index N - k must stay >= 0
  const size_t lds = (size_t)p.seg_n * sizeof(float);
  if (lds > 64 * 1024) return NBM_EUNSUPPORTED;
  dim3 grid((n_frames + 16 * FC - 1) / (16 * FC), (n_bins + 64 * BR - 1) / (64 * BR), batch);
  hipLaunchKernelGGL((stft_f64_kernel<BR, FC>), grid, dim3(256), lds, (hipStream_t)stream, p);
  return nbm_launch_status();
}

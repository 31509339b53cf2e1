// A2-A5: materialising feature builder  out[b,:] = [ X | phi(coords) | psi(t) ]   (HBM-write bound)
//
// Replaces SpatialBasisEmbedding.forward/_wendland/_gaussian/_triangular
// (stnf/models/st_interp.py:433-491), TemporalBasisEmbedding.forward (:583-596) and the feature
// concat of STInterpMLP.forward (:843-846).
//
// Layout: one workgroup = 256 threads owns a tile of TILE_C = 1024 output columns and walks
// ROWS_PER_WG observation rows.  A thread keeps its 4 knots (cx, cy, 1/(bw*cal)) in registers for
// the whole walk, the observation (x, y, t) of the current row is wave-uniform (scalar loads), and
// every row costs each wave ONE 1-KiB contiguous store (float4 per lane), i.e. a workgroup writes
// 4 KiB contiguous per row.  Algorithmic bytes: 12 B read + 4*(p+Ks+Kt) B written per observation.
#include "basis.h"

#include <stdlib.h>

namespace stdadk {

constexpr int TILE_C = 1024;
constexpr int RB_THREADS = 256;

// kinds of an output column
enum { COL_X = 0, COL_S = 1, COL_T = 2, COL_PAD = 3 };

template <int VEC, int BASIS>
__global__ __launch_bounds__(RB_THREADS) void rbf_build_kernel(
    const float *__restrict__ coords, const float *__restrict__ t, const float *__restrict__ X,
    int64_t B, int p, const float *__restrict__ s_centers, const float *__restrict__ s_bw,
    int64_t Ks, float cal, const float *__restrict__ t_centers, const float *__restrict__ t_bw,
    int64_t Kt, float *__restrict__ out, int64_t ld_out, int rows_per_wg) {
  const int tid = threadIdx.x;
  const int64_t tile_c0 = (int64_t)blockIdx.x * TILE_C;
  const int64_t row0 = (int64_t)blockIdx.y * rows_per_wg;
  const int64_t row1 = min(row0 + (int64_t)rows_per_wg, B);
  const int64_t D = (int64_t)p + Ks + Kt;
  // workgroup-uniform: does this column tile lie entirely inside the spatial block [p, p+Ks)?
  const bool all_spatial = tile_c0 >= p && tile_c0 + TILE_C <= p + Ks;

  if (all_spatial) {
    // this thread's 4 columns: VEC==4 -> 4 consecutive (one float4 store); VEC==1 -> strided by 256
    int64_t col[4];
    float c0[4], c1[4], sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      col[j] = (VEC == 4) ? tile_c0 + 4 * tid + j : tile_c0 + tid + (int64_t)RB_THREADS * j;
      const int64_t k = col[j] - p;
      c0[j] = s_centers[2 * k];
      c1[j] = s_centers[2 * k + 1];
      sc[j] = knot_scale(s_bw[k], cal);
    }
    // hot loop: 4 phi per lane, one 16-byte store per lane and row (1 KiB per wave, 4 KiB per WG)
    for (int64_t b = row0; b < row1; ++b) {
      const float x = coords[2 * b], y = coords[2 * b + 1];      // wave-uniform: scalar loads
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = phi_eval<BASIS>(x, y, c0[j], c1[j], sc[j]);
      float *orow = out + b * ld_out;
      if (VEC == 4) {
        *reinterpret_cast<float4 *>(orow + col[0]) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) orow[col[j]] = v[j];
      }
    }
    return;
  }
  // Edge tiles (covariate copy, temporal basis, zero padding, a few spatial columns): usually only
  // a fraction of the tile's 1024 columns exists, so the 4 waves split the ROWS (wave w takes rows
  // row0+w, +4, ...) and each wave sweeps the tile's columns 256 at a time (4 per lane).
  const int lane = tid & 63, wave = tid >> 6;
  for (int64_t cbase = tile_c0; cbase < min(tile_c0 + (int64_t)TILE_C, ld_out); cbase += 256) {
    int64_t col[4];
    float c0[4], c1[4], sc[4];
    int kind[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      col[j] = (VEC == 4) ? cbase + 4 * lane + j : cbase + lane + 64 * j;
      const int64_t c = col[j];
      c0[j] = 0.f; c1[j] = 0.f; sc[j] = 1.f;
      if (c >= p && c < p + Ks) {
        kind[j] = COL_S;
        const int64_t k = c - p;
        c0[j] = s_centers[2 * k];
        c1[j] = s_centers[2 * k + 1];
        sc[j] = knot_scale(s_bw[k], cal);
      } else if (c < p) {
        kind[j] = COL_X;
      } else if (c < D) {
        kind[j] = COL_T;
        const int64_t k = c - p - Ks;
        c0[j] = t_centers[k];
        sc[j] = t_bw[k];
      } else {
        kind[j] = COL_PAD;
      }
    }
    for (int64_t b = row0 + wave; b < row1; b += RB_THREADS / 64) {
      const float x = Ks > 0 ? coords[2 * b] : 0.f;
      const float y = Ks > 0 ? coords[2 * b + 1] : 0.f;
      const float tt = (Kt > 0) ? t[b] : 0.f;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kind[j] == COL_S) {
          v[j] = phi_eval<BASIS>(x, y, c0[j], c1[j], sc[j]);
        } else if (kind[j] == COL_T) {
          v[j] = psi_eval(tt, c0[j], sc[j]);
        } else if (kind[j] == COL_X) {
          v[j] = X[b * p + col[j]];
        } else {
          v[j] = 0.f;
        }
      }
      float *orow = out + b * ld_out;
      if (VEC == 4 && col[3] < ld_out) {
        *reinterpret_cast<float4 *>(orow + col[0]) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col[j] < ld_out) orow[col[j]] = v[j];
      }
    }
  }
}

template <int VEC, int BASIS>
static void launch_rbf(const float *coords, const float *t, const float *X, int64_t B, int p,
                       const float *s_centers, const float *s_bw, int64_t Ks, float cal,
                       const float *t_centers, const float *t_bw, int64_t Kt, float *out,
                       int64_t ld_out, hipStream_t st) {
  const int64_t n_tiles = ceil_div(ld_out, TILE_C);
  // ONE launch covers every column tile so the few edge tiles run beside the spatial ones.  Rows per workgroup:
  // 4.  The grid's x index (column tile) runs fastest, so the workgroups in flight at any time cover a dense
  // band of consecutive rows of the output: with 4 rows each that band is a few MB that the memory controller
  // sees as long sequential write bursts; with 64 rows per workgroup every workgroup is its own 4-KiB-per-42-KiB
  // strided stream (~200 of them) and the DRAM pages thrash once the output no longer fits the 256 MiB
  // Infinity Cache (MI355X, tools/bench_rbf_rows.py: C2 x 16 384 rows = 680 MB 4.94 -> 5.69 TB/s, C4 x 4 096
  // = 816 MB 4.55 -> 5.24, C4 x 16 384 = 3.3 GB 5.21 -> 6.14; 1-2 rows lose again to the per-workgroup
  // knot-table loads).  STDADK_RBF_ROWS overrides (measurement aid).
  int rows = 4;
  { const char *e = getenv("STDADK_RBF_ROWS"); if (e && atoi(e) > 0) rows = atoi(e); }
  while (ceil_div(B, rows) > 65535) rows <<= 1;      // gridDim.y limit
  dim3 grid((unsigned)n_tiles, (unsigned)ceil_div(B, rows));
  STDADK_LAUNCH((rbf_build_kernel<VEC, BASIS>), grid, dim3(RB_THREADS), 0, st, coords, t, X, B, p, s_centers,
                s_bw, Ks, cal, t_centers, t_bw, Kt, out, ld_out, rows);
}

}  // namespace stdadk

using namespace stdadk;

extern "C" int stdadk_rbf_build_f32(const float *coords, const float *t, const float *X, int64_t B,
                                    int32_t p, const float *s_centers, const float *s_bw,
                                    int64_t Ks, int32_t basis, const float *t_centers,
                                    const float *t_bw, int64_t Kt, float *out, int64_t ld_out,
                                    stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && p >= 0 && Ks >= 0 && Kt >= 0, STDADK_E_ARG, "rbf_build: negative size");
  STDADK_REQUIRE(basis >= 0 && basis <= 2, STDADK_E_ARG, "rbf_build: unknown basis %d", basis);
  STDADK_REQUIRE(ld_out >= p + Ks + Kt, STDADK_E_SHAPE, "rbf_build: ld_out %lld < p+Ks+Kt %lld",
                 (long long)ld_out, (long long)(p + Ks + Kt));
  if (B == 0 || ld_out == 0) return 0;
  STDADK_REQUIRE(out != nullptr, STDADK_E_ARG, "rbf_build: out is NULL");
  STDADK_REQUIRE(Ks == 0 || (coords && s_centers && s_bw), STDADK_E_ARG,
                 "rbf_build: spatial inputs NULL");
  STDADK_REQUIRE(Kt == 0 || (t && t_centers && t_bw), STDADK_E_ARG, "rbf_build: temporal inputs NULL");
  STDADK_REQUIRE(p == 0 || X, STDADK_E_ARG, "rbf_build: X is NULL with p=%d", p);
  static const float cals[3] = {1.000000f, 0.223477f, 0.654714f};  // st_interp.py:56-60
  const float cal = cals[basis];
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = aligned16(out) && (ld_out % 4 == 0);
#define GO(V, BS) launch_rbf<V, BS>(coords, t, X, B, p, s_centers, s_bw, Ks, cal, t_centers, t_bw, Kt, out, ld_out, st)
  if (vec4) {
    if (basis == 0) GO(4, 0); else if (basis == 1) GO(4, 1); else GO(4, 2);
  } else {
    if (basis == 0) GO(1, 0); else if (basis == 1) GO(1, 1); else GO(1, 2);
  }
#undef GO
  STDADK_CHECK_LAUNCH("rbf_build");
  return 0;
}

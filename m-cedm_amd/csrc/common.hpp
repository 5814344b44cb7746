// common.hpp -- shared declarations for libmcedm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mcedm_hip.h"

namespace mcedm {

void set_error(const char* fmt, ...);

#define MCEDM_HIP_TRY(expr)                                                                   \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      ::mcedm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return MCEDM_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

#define MCEDM_LAUNCH_CHECK(what)                                                               \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess) {                                                                  \
      ::mcedm::set_error("launch of %s failed: %s", what, hipGetErrorString(e__));            \
      return MCEDM_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

#define MCEDM_REQUIRE(cond, ...)                                                              \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      ::mcedm::set_error(__VA_ARGS__);                                                        \
      return MCEDM_ERR_INVALID;                                                               \
    }                                                                                         \
  } while (0)

// Kernel-variant choices.  Three levels, first hit wins: the plan whose entry point is executing on this thread
// (mcedm_*_plan_set_variant; a field of the plan, so two plans in one process -- on two threads or two streams -- cannot flip each
// other's kernels), the process-wide test hooks mcedm_op_set_* (kernel-level calls have no plan), the environment.
enum KernelVariant { KV_CONV_WINO = 0, KV_CONV_WINO1 = 1, KV_CONV_RESIDENT = 2, KV_CONV8 = 3, KV_ATTN_FUSED = 4, KV_WGRAD_WINO = 5,
                     KV_CONV1X1_REG = 6, KV_COUNT = 7 };
struct KernelVariants { int v[KV_COUNT] = {-1, -1, -1, -1, -1, -1, -1}; };
const KernelVariants* current_variants();                 // of the executing plan-level call on this thread, or null
struct VariantScope {                                      // first statement of every extern "C" function that takes a plan
  const KernelVariants* prev;
  explicit VariantScope(const KernelVariants* kv);
  ~VariantScope();
};
// value of switch `which`: the executing plan's if it set one, else `global` (an mcedm_op_set_* value) if >= 0, else env_default
static inline int variant_choice(int which, int global, int env_default) {
  const KernelVariants* kv = current_variants();
  if (kv && kv->v[which] >= 0) return kv->v[which];
  return global >= 0 ? global : env_default;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
#ifdef __HIPCC__
// 2-D grid (nx tiles of one (sample, head) along x, ny = B * heads along y): the nx workgroups that stream the SAME keys / values land on
// ONE XCD (workgroup ids go round-robin over the 8 XCDs in dispatch order, x fastest), so that after the first of them the others
// find the streamed tiles in that XCD's L2.  A bijection of the grid; ny % 8 != 0 (or map == 0): the identity.
__device__ __forceinline__ void xcd_group_map(int map, int& bx, int& by) {
  const int nx = gridDim.x, ny = gridDim.y;
  bx = blockIdx.x; by = blockIdx.y;
  if (!map || (ny & 7) != 0) return;
  const int lin = bx + by * nx;
  const int c = lin & 7, i = lin >> 3;
  by = (i / nx) * 8 + c;
  bx = i - (i / nx) * nx;
}
#endif

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------
// Per-(sample, channel) input transform applied while a conv / GEMM stages its input:
//   v' = act((v - mean) * scale + offset)
// GroupNorm, the FiLM modulation addcmul(shift, norm(x), scale+1) and the EDM c_in factor
// are all instances of it (adm_blocks.py:94-97,166; mcedm.py:208).
struct __attribute__((aligned(16))) Coef {
  float mean, scale, offset, pad;
};

// RS_S2: no resampling of the source but the conv itself has stride 2 over a source padded by one zero row / column at
// the bottom / right (the DDPM Downsample, models/ddim_blocks.py:97-101); forward conv only.
enum Resample { RS_NONE = 0, RS_UP = 1, RS_DOWN = 2, RS_S2 = 3 };

// Geometry of a fused GroupNorm statistics table: the pixel tiling of the conv that produced the tensor.  One record
// per (sample, tile, 4-channel block) = (sum, M2): the sum of the block's values inside the tile and their squared
// deviations from the TILE's own mean, so no large-mean cancellation ever happens in fp32; the consumer merges the
// records in fp64 (Chan et al.), with the element count of a tile recomputed from this geometry.
struct SumTiles {
  int tiles = 0;     // records per sample and channel block
  int tiles_x = 0;   // tiles along W
  int ph = 0, pw = 0;
  int rc = 4;        // channels per record: 4, or 2 where a consumer's GroupNorm has 2-channel groups (ConvArgs::gsum_rc)
};
__host__ __device__ static inline int sum_tile_count(const SumTiles& g, int t, int H, int W) {
  const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
  const int h = H - ty * g.ph < g.ph ? H - ty * g.ph : g.ph;
  const int w = W - tx * g.pw < g.pw ? W - tx * g.pw : g.pw;
  return g.rc * h * w;
}

// Arguments of the implicit-GEMM convolution (3x3 pad 1, or 1x1).
// GroupNorm (+FiLM) -> per-(sample, channel) transform rows (K1)
struct GnArgs {
  const float* xa; const float* xb; int Ca, Cb;
  int HW, B, groups;
  const float* gamma; const float* beta;     // [C]
  const float* film;  // row n: film[n*film_stride + (0..C) = scale | (C..2C) = shift], or null
  int film_batch;     // 1: one row per sample, 0: row 0 broadcast over the batch
  int film_stride;
  float eps;
  Coef* coef;        // out [B][C]
  float* stats;      // out [B][groups][2] (mean, rstd) or null (kept for backward)
  const float* suma; const float* sumb;     // launch_gn_coef_from_sums: per-tile (sum, M2) tables of xa / xb
  SumTiles ta, tb;                          // tiling of those tables
  int W;                                    // image width (H = HW / W): a tile's element count comes from its geometry
};
int launch_gn_coef_from_sums(const GnArgs& a, hipStream_t stream);
int launch_gn_coef(const GnArgs& a, hipStream_t stream);
// true when the (sum, sumsq) tables in `a` can replace a pass over the tensor (4-channel groups tile the GroupNorm groups)
bool gn_sums_usable(const GnArgs& a);

struct ConvArgs {
  const float* xa;   // first source of the virtual channel concat [B, Ca, Hs, Ws] (may be null => zeros)
  const float* xb;   // second source [B, Cb, Hs, Ws] (null iff Cb == 0)
  int Ca, Cb;
  const Coef* coef;  // [coef_batch ? B : 1][Ca + Cb], or null (identity, no activation)
  int coef_batch;    // 1: per-sample rows, 0: one row broadcast over the batch
  int act;           // 1: SiLU after the affine
  int resample;      // Resample applied to the (activated) source before the conv
  int Hs, Ws;        // source spatial size
  int H, W;          // conv input == output spatial size
  const float* wpk;  // packed weights (see pack_conv_weights)
  const float* wino; // optional: the same weights in Winograd F(2x2, 3x3) form (launch_pack_conv_wino), or null
  const float* bias; // [Cout] (packed order) or null
  const float* res;  // residual [B, Cout, Hr, Wr] or null
  int res_mode;      // Resample applied to the residual source
  float* out;        // [B, Cout, H, W]
  int Cout;
  int B;
  float* gsum;       // optional [B][tiles][ceil(Cout/4)][2]: per output tile and 4-channel block, (sum, M2 about the
                     // tile mean) of the OUTPUT written by the epilogue (no atomics); feeds the next GroupNorm without
                     // a stats pass.  Must hold B * conv_max_tiles(H, W) * ceil(Cout/4) * 2 floats.
  int gsum_rc;       // channels per statistics record: 0 / 4 = quads; 2 = pairs (table holds ceil(Cout/2) records per tile)
  SumTiles* gsum_tiles;   // host out: the tiling the launcher used (rows of gsum per sample, tile shape, record width)
  // gn_on: the kernel derives this sample's transform rows itself from the producers' per-tile (sum, sumsq) tables
  // (gn.suma / gn.sumb ...; gn.coef / gn.stats / gn.xa / gn.xb are not used), so no GroupNorm kernel runs at all
  // Optional second GEMM onto the same output tile: a 1x1 projection of another (un-transformed) tensor, folded into this
  // conv as extra K chunks at the centre tap.  Used for the decoder blocks' skip projection (adm_blocks.py:150-151, 171:
  // x = conv1(...) + skip(orig)): the projected tensor is never written or read back and one launch disappears.
  // 3x3, un-resampled convs only.  sk_wpk: packed 1x1 weights (rows [Cin padded to 16][CoutP]); sk_bias: [Cout] or null.
  const float* sk_xa; const float* sk_xb; int sk_Ca, sk_Cb;
  const float* sk_wpk; const float* sk_bias;
  GnArgs gn; int gn_on;
  int coef_rows;     // set by the launcher: 1 = coef holds Ca+Cb rows (per sample if coef_batch), 0 = a single identity row
  unsigned long long* dbg;   // diagnostics only (mcedm_op_set_conv_debug): 4 timestamps (10 ns) + CU id per workgroup
};

int launch_conv(const ConvArgs& a, int taps, hipStream_t stream);
// conv_wino.hip: Winograd F(2x2, 3x3) kernel for un-resampled 3x3 convs with Cout % 128 == 0 on (8, 16)-divisible images
size_t conv_wino_packed_floats(int Cout, int Cin);
int launch_pack_conv_wino(const float* w, float* dst, int Cout, int Cin, int transpose_flip, hipStream_t stream);   // w [Cout][Cin][3][3]; transpose_flip: w is [Cin][Cout][3][3], build the data-gradient weights
bool conv_wino_applicable(const ConvArgs& a, int taps);
bool conv_wino_shape_ok(int Cout, int Cin, int H, int W);
void set_conv_wino(int enable);                          // 1 / 0, -1: default (env MCEDM_WINOGRAD, else on)
bool conv_wino_preferred(const ConvArgs& a);              // env MCEDM_WINOGRAD (default on), MCEDM_WINO_MIN_HW (default 32 x 32)
int launch_conv_wino(const ConvArgs& a, hipStream_t stream);
static inline int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }     // channel padding of the packed tables
int conv_resolve_identity(ConvArgs& a);                  // points a missing transform table at the identity row
unsigned long long* conv_debug_buffer();
int try_launch_conv_resident(const ConvArgs& a, int taps, hipStream_t stream);   // conv_resident.hip; -1: not served there
int try_launch_conv1x1_reg(const ConvArgs& a, int taps, hipStream_t stream);     // conv1x1_reg.hip (un-transformed 1x1 convs); -1: not served
void set_conv1x1_reg(int enable);                        // 1 / 0, -1: default (env MCEDM_CONV1X1_REG, else on)
void set_conv_resident(int enable);                      // 1 / 0, -1: default (env MCEDM_CONV_RESIDENT, else on)
static inline int conv_max_tiles(int H, int W) { return ((H + 3) / 4) * ((W + 7) / 8); }   // smallest pixel tile is 4x8 (conv_resident.hip)
void set_conv_tile_override(int mt, int ph, int pw);
void set_conv8(int enable);   // 1 / 0, -1: default (env MCEDM_CONV8, else off)
void set_conv_debug(unsigned long long* buf);   // test hook; (0,0,0) restores the heuristic
// geometry the packer must use for a given (Cout, taps): tile height over Cout and K-chunk
int conv_mt_for(int Cout);
int conv_kc_for(int taps);
size_t conv_packed_floats(int Cout, int Cin, int taps);
// src [Cout][Cin][taps]; perm_qkv_heads > 0 permutes output rows from the reference's interleaved
// (head, c, {q,k,v}) order to (head, {q,k,v}, c).  transpose_flip: build the dgrad weights.
int launch_pack_conv(const float* w, float* dst, int Cout, int Cin, int taps, int qkv_heads, int transpose_flip,
                     hipStream_t stream);
int launch_pack_bias(const float* b, float* dst, int Cout, int qkv_heads, hipStream_t stream);

// GroupNorm statistics + coefficient table (K1).  x = virtual concat of xa[B,Ca,HW], xb[B,Cb,HW].


// sigma-embedding MLP + all FiLM affine rows (K6)
struct EmbArgs {
  const float* labels; int n;      // noise labels [n]
  int ch;                          // embedding width
  const float* freqs;              // [ch/2]
  const float* w0; const float* b0; const float* w1; const float* b1;  // map_layer0/1, [ch][ch] row-major (out,in)
  const float* waff; const float* baff; int rows;  // concatenated affine weights [rows][ch], bias [rows]
  float* emb;                      // out [n][ch]
  float* film;                     // out [n][rows]
};
int launch_embedding(const EmbArgs& a, hipStream_t stream);

// attention (K5): qkv [B][heads][3][64][T] -> a [B][heads*64][T]
int launch_attention(const float* qkv, float* out, int B, int heads, int T, hipStream_t stream);
// attn_fused.hip: z = proj(attention(qkv(group_norm(y)))) + y in one launch (8 x 8 tokens, 64 channels, one head; inference)
bool attn_block_fused_applicable(int C, int heads, int H, int W, int groups);
int launch_attn_block64(const float* y, float* z, const float* gamma, const float* beta, float eps, int groups, const float* wq,
                        const float* bq, const float* wp, const float* bp, float* gsum, SumTiles* gsum_tiles, int B,
                        hipStream_t stream);
void set_attn_fused(int enable);      // 1 / 0, -1: default (env MCEDM_ATTN_FUSED, else on)

}  // namespace mcedm

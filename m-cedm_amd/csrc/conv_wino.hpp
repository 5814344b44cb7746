// conv_wino.hpp -- constants and LDS layouts shared by the two Winograd F(2x2, 3x3) kernels (conv_wino.hip: eight waves per
// workgroup, two per SIMD; conv_wino1.hip: four waves, one per SIMD with all sixteen positions).  Device code only.
#pragma once
#include "conv_tile.hpp"

namespace mcedm {

constexpr int WPH = 8, WPW = 16;                 // output pixels per workgroup
constexpr int WTX = WPW / 2, WTY = WPH / 2;      // 2x2 patches: 8 x 4 = 32 = one MFMA N block
constexpr int WKC = 8;                           // input channels per chunk
static_assert(WTX * WTY == 32, "one MFMA N block of patches per workgroup");
// MB = 32-channel output blocks per workgroup: 4 (128 channels, 512 threads, one workgroup per CU, two chunks per stage) or
// 2 (64 channels -- the ch = 64 networks --, 256 threads, two workgroups per CU, one chunk per stage so that both fit the LDS)
template <int MB_>
struct WinoCfg {
  static constexpr int MB = MB_, NT = 128 * MB_, NW = 2 * MB_, MT = 32 * MB_;
  static constexpr int SC = MB_ == 4 ? 2 : 1;          // chunks per stage of the K loop (one barrier per stage)
  static constexpr int CPW = WKC / NW;                  // channels of a chunk that one wave stages: 1 or 2
  static constexpr int TI = 4 / MB_;                    // patch rows (of the four) per thread in the input transform
  static constexpr int VBUF = SC * 16 * 288, RBUF = SC * WKC * 196;      // floats per stage (VPOS, RPLANE below)
  static constexpr int LDS_ROWS_OFF = 2 * VBUF + 2 * RBUF;              // transform rows start here (floats; 16-byte aligned)
  static constexpr int XCH_FLOATS = NW * 16 * 64;                       // epilogue exchange: one round of 16 registers x 64 lanes per wave
  static constexpr int RED_FLOATS = 2 * (MT / 2) * 3 + MT;              // statistics records of the two halves (pairs at most) + the bias row
  static_assert(MB_ == 4 || MB_ == 2, "128 or 64 output channels per workgroup");
  static_assert(LDS_ROWS_OFF % 4 == 0, "LDS layout");
};
constexpr int RROWS = WPH + 2, RPITCH = WPW + 2; // raw tile with halo: 10 x 18
constexpr int RPLANE = 196;                      // floats between channels of the raw tile (180 used); = 4 mod 32: hipcc merges the transform's two
                                                 // adjacent 8-byte reads into ds_read2_b64, which banks at dword mod 32 over 16-lane groups = eight
                                                 // channels x two patch columns x two dwords: channels 4 banks apart cover the 32 banks once
constexpr int RSUB = (RROWS * RPITCH + 63) / 64; // raw elements per lane and channel: 3
constexpr int WINO_IL_K = 5;                     // side-work instructions the K loop's recipe admits behind each MFMA
// V tile of one (chunk, position): [k parity h][patch 32][k-step 4]; the h = 1 block starts at float 144 = 16 mod 32: the
// transform's dword writes bank at dword mod 32 per 32-lane group (four patch columns x eight channels: 4 ttx + (k >> 1) + 16 (k & 1)
// covers the 32 banks once); the 16-byte B-fragment reads (dword mod 64, 16-lane groups inside one h half) only need it 16-byte
// aligned.  (160 and RPLANE 208, the first layout, were 2-way on those writes and 4-way on the transform's reads:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.48, profiles/r3_s128_mfma_lds_counters.json.)
constexpr int VH1 = 144, VPOS = 288;
static_assert(RPLANE % 32 == 4 && VH1 % 32 == 16 && VH1 >= 128 && VPOS >= VH1 + 128 && RPLANE >= RROWS * RPITCH && RPLANE % 2 == 0 && VPOS % 4 == 0 && VH1 % 4 == 0, "LDS layout");



// conv_wino1.hip: the 128-channel shape with ONE wave per SIMD (see there); -1: not served (shape / switch), else the launch status
int try_launch_conv_wino1(const ConvArgs& a, hipStream_t stream);
void set_conv_wino1(int enable);                 // 1 / 0, -1: default (env MCEDM_WINO1, else off: measured 6-9 % slower)
// tiles per persistent workgroup (a divisor of the tiles per image; conv_wino.hip)
int wino_tiles_per_wg(long long total, int tiles_img, int slots);

}  // namespace mcedm

// plan.hpp -- host-side description of the network (block list, parameter table, packed-weight
// layout) and of the activation layout inside the caller's workspace.
#pragma once
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace mcedm {

constexpr size_t NONE = (size_t)-1;

// first-fit pool over the caller's workspace, simulated on the host (offsets only)
struct Pool {
  std::vector<std::pair<size_t, size_t>> free_;   // (offset, bytes), sorted by offset
  size_t top = 0, peak = 0;
  bool keep_all = false;
  size_t alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].second >= bytes) {
        const size_t off = free_[i].first;
        if (free_[i].second == bytes) free_.erase(free_.begin() + i);
        else { free_[i].first += bytes; free_[i].second -= bytes; }
        return off;
      }
    const size_t off = top;
    top += bytes;
    peak = std::max(peak, top);
    return off;
  }
  void release(size_t off, size_t bytes) {
    if (keep_all) return;
    bytes = align_up(bytes ? bytes : 1, 256);
    auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(off, (size_t)0));
    it = free_.insert(it, {off, bytes});
    if (it + 1 != free_.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; free_.erase(it + 1); }
    if (it != free_.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; it = free_.erase(it) - 1; }
    if (it->first + it->second == top) { top = it->first; free_.erase(it); }
  }
};

struct ParamInfo {
  std::string name;
  int ndim = 0;
  int64_t shape[4] = {1, 1, 1, 1};
  int64_t numel = 0;
};

struct ConvP {   // one Conv2d with weights
  int w = -1, b = -1;          // parameter indices
  int cin = 0, cout = 0, taps = 0;
  int qkv_heads = 0;           // > 0: output rows re-ordered to (head, {q,k,v}, c)
  size_t wpk = NONE, bias = NONE;      // float offsets into the packed buffer
  size_t wpk_dgrad = NONE;             // transposed + mirrored weights for the data gradient
  size_t wino = NONE, wino_dgrad = NONE;   // both again in Winograd F(2x2, 3x3) form (conv_wino.hip) where that kernel can serve
};

struct NormP {
  int w = -1, b = -1;
  int C = 0, groups = 0;
  size_t gamma = NONE, beta = NONE;    // float offsets into the packed buffer
};

struct BlockP {
  std::string key;
  int cin = 0, cout = 0;
  bool up = false, down = false, attn = false;
  int heads = 0;
  int skip_kernel = -1;        // -1 none (identity), 0 resample only, 1 1x1 conv
  NormP norm0, norm1, norm2;
  ConvP conv0, conv1, skip, qkv, proj;
  int aff_w = -1, aff_b = -1;
  int film_row0 = 0;           // first row of this block's (scale|shift) in the film table
};

// a tensor inside the workspace
struct TRef {
  size_t off = NONE;           // byte offset
  int C = 0, H = 0, W = 0;
  size_t bytes = 0;
  size_t sums = NONE;          // byte offset of the fused GroupNorm per-tile (sum, sumsq) table of this tensor, if any
  int sum_tiles = 0;           // tiles per sample the producing conv used (filled while the forward is scheduled)
  int ref = 0;                 // live references while the layout is being simulated
};

struct BlockLayout {
  int xa = -1, xb = -1;        // input tensor ids (xb = popped skip for concat blocks)
  int coef0 = -1, h = -1, coef1 = -1, sk = -1, y = -1, coef2 = -1, qkv = -1, a = -1, z = -1;
  int stats0 = -1, stats1 = -1, stats2 = -1;
  int xd = -1;                 // down blocks: the activated, 2x2-averaged input of conv0 (temporary)
  int out = -1;                // y or z
  int Hin = 0, Win = 0, H = 0, W = 0;
};

struct Layout {
  std::vector<TRef> t;         // all tensors
  std::vector<BlockLayout> blocks;   // encoder blocks then decoder blocks
  int film = -1, t0 = -1, coef_out = -1, stats_out = -1, last = -1;
  // dx_cond head (adm_blocks.py:334-362): cat(x, dx) staging buffer (MCEDM_DX_CAT); conv_in output, dx_enc.0 output, its
  // GELU, dx_enc.2 output in front of combine_enc, whose output is then t0 (MCEDM_DX_ENC)
  int xdx = -1, xf = -1, d1 = -1, g1 = -1, d2 = -1;
  size_t sums_base = 0, sums_bytes = 0;   // arena of all fused-statistics tables
  size_t total_bytes = 0;
};

}  // namespace mcedm

struct mcedm_plan {
  mcedm_unet_desc desc;
  std::vector<mcedm::ParamInfo> params;
  mcedm::ConvP conv_in, conv_out;
  mcedm::ConvP dx_enc0, dx_enc2, combine;      // MCEDM_DX_ENC only (adm_blocks.py:266-280)
  mcedm::NormP out_norm;
  std::vector<mcedm::BlockP> enc, dec;
  int map0_w = -1, map0_b = -1, map1_w = -1, map1_b = -1;
  // packed buffer (float offsets)
  size_t freqs = mcedm::NONE, w0 = mcedm::NONE, b0 = mcedm::NONE, w1 = mcedm::NONE, b1 = mcedm::NONE;
  size_t waff = mcedm::NONE, baff = mcedm::NONE;
  int film_rows = 0;
  size_t packed_floats = 0;
  int levels_div = 1;          // 2^(n_levels-1)
  mcedm::KernelVariants variants;   // per-plan kernel choices (mcedm_unet_plan_set_variant); -1 = process default
};

namespace mcedm {
int build_layout(const mcedm_plan& P, int B, int H, int W, int training, int n_noise, Layout* out);

// header in front of the U-Net activations: EDM coefficient rows, conv_in transform, F / F_uncond
struct Header {
  size_t coefs4, c_noise, coef_in, F, Fu, total;
};
Header header_for(const mcedm_plan& P, int B, int H, int W);

template <class T>
static inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }

// bytes the backward needs behind the training-mode activations (gradient buffers + scratch)
size_t backward_scratch_bytes(const mcedm_plan& P, const Layout& L, int B, int H, int W);
}

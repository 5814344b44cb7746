// conv1x1_reg.hip -- the un-transformed 1x1 convolution (the decoder blocks' skip projection, adm_blocks.py:150-151, 171) as a
// register-direct GEMM:  out[n][co][px] = bias[co] + sum_ci W[co][ci] * cat(xa, xb)[n][ci][px]  (+ res[n][co][px]).
//
// Why a kernel of its own: conv_mfma_kernel<128, 8, 32, 1, 4, 1, 16> runs these at 0.50-0.55 of the fp32 matrix rate (432 us for
// 256 -> 128 at 128^2, B = 32; a LONE workgroup needs 71 us for a tile whose MFMAs take 31 us): per 16-channel chunk it stages the
// pixel tile through registers, an (unconditional) activation and LDS behind two barriers.  A 1x1 conv needs none of that:
//   * the B operand of v_mfma_f32_32x32x2_f32 is one value per lane, k = lane >> 5, column = lane & 31 -- and WHICH pixel a column
//     stands for is free.  A lane loads 16 bytes = pixels 4 l .. 4 l + 3 of channel ci (h = lane >> 5 picks the channel parity): its
//     four registers are the B operands of FOUR 32-column blocks (block e <-> pixels 4 l + e), straight from global memory, 512
//     contiguous bytes per channel and half-wave.  The accumulators of the four blocks then hold four consecutive pixels per lane:
//     16-byte stores.  No LDS, no transposition, no barrier on the pixel side.
//   * the A operand (weights) is the same for every pixel tile: the whole [128 co][Cin <= 256] matrix sits in LDS for the life of
//     the (persistent) workgroup, in MFMA fragment order -- per channel row 16-channel groups of four 16-byte slots
//     (slot = 2 g + h, four k-steps each), slots XOR-swizzled by (co >> 2) & 3, rows Cin + 16 floats apart (= 16 mod 64): the
//     ds_read_b128 fragment reads are bank-conflict free.
// Workgroup = 512 threads = 8 waves (2 per SIMD): wave = (64-channel half, 128-pixel quarter) of a 128-channel x 512-pixel tile =
// 2 x 4 accumulator blocks; K loop in stages of 16 channels: 8 16-byte loads for the stage after next... one stage ahead, 4 LDS
// reads, 64 MFMAs, no barrier.  Sums run over ci in ascending order (pairs (2 s + h) inside an MFMA): deterministic, batch-size
// independent (a pixel's result does not depend on which tile or workgroup computes it).
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "conv_tile.hpp"
#include "prof.hpp"

namespace mcedm {

constexpr int C1_KS = 16;                   // channels per stage
// output channels per workgroup MT = 128 (tile of 512 pixels: 2 channel halves x 4 pixel quarters of 128) or 64 (the ch = 64
// network's 128 -> 64 skip projections: one channel half x 8 pixel parts = 1024 pixels, every input element fetched once)
constexpr int c1_px(int mt) { return mt == 128 ? 512 : 1024; }

struct Conv1Args {
  const float* xa; const float* xb; int Ca, Cb;
  const float* wpk; const float* bias; const float* res; float* out;
  int Cout, coutp, B;
  unsigned HW;
  int tiles_img, ntiles, per;               // 512-pixel tiles per image, in all, per workgroup
  int nwg;                                  // workgroups along x: workgroup b takes tiles b, b + nwg, b + 2 nwg, ...
};

// NI = 32-channel blocks per wave: 2 (64 channels x 128 pixels per wave, 16-byte loads; every input element is fetched by the two
// waves that share a pixel quarter) or 4 (all 128 channels x 64 pixels per wave, 8-byte loads: every element fetched ONCE per
// workgroup, twice the LDS fragment reads)
template <int NI, int MT = 128>
__global__ __launch_bounds__(512, 1) void conv1x1_reg_kernel(const Conv1Args p) {
  static_assert(MT == 128 || (MT == 64 && NI == 2), "conv1x1_reg: 64-channel workgroups exist in the NI = 2 form only");
  constexpr int C1_MT = MT, C1_PX = c1_px(MT);
  constexpr int NE = 8 / NI;                                      // pixels (= 32-column blocks) per lane
  typedef float bvec __attribute__((ext_vector_type(NE)));
  extern __shared__ float wl[];                                   // [128][Cin + 16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int coh = (NI == 2 && MT == 128) ? (wave & 1) : 0, pq = (NI == 2 && MT == 128) ? (wave >> 1) : wave, l31 = lane & 31, h = lane >> 5;
  const int Cin = p.Ca + p.Cb, nst = Cin / C1_KS;
  const int pitch = Cin + 16;
  const int m0 = blockIdx.y * C1_MT;

  // ---- the weight matrix of this workgroup's 128 output channels -> LDS, fragment order (once per workgroup)
  {
    const int nslots = C1_MT * (Cin / 4);                         // 16-byte slots
    for (int id = tid; id < nslots; id += 512) {
      const int co = id % C1_MT, q = id / C1_MT;                  // consecutive lanes: consecutive output channels (coalesced rows)
      const int c = q >> 2, sig = q & 3, g = sig >> 1, hh = sig & 1;
      f32x4 v;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ci = C1_KS * c + 8 * g + 2 * s + hh;
        v[s] = (m0 + co < p.coutp) ? p.wpk[(size_t)ci * p.coutp + m0 + co] : 0.f;
      }
      *reinterpret_cast<f32x4*>(wl + co * pitch + c * C1_KS + 4 * (sig ^ ((co >> 2) & 3))) = v;
    }
  }
  __syncthreads();

  const unsigned lvo = 4u * ((unsigned)h * p.HW + (unsigned)NE * l31);      // lane part of every B load: channel parity, first pixel
  const int afo = (coh * 64 + l31) * pitch + 4 * ((l31 >> 2) & 3) * 0;   // row of this lane's first A block (the swizzle is applied per slot)
  const int asw = (l31 >> 2) & 3;

  // strided tile assignment: at any time the chip's workgroups work on ~nwg CONSECUTIVE tiles, i.e. on whole channel planes of a
  // few samples, channel by channel -- neighbouring 2 KB pieces of a plane are requested at about the same time
  for (int t = blockIdx.x; t < p.ntiles; t += p.nwg) {
    const int n = t / p.tiles_img, tq = t - n * p.tiles_img;
    const unsigned px0 = (unsigned)tq * C1_PX + (unsigned)pq * (32u * NE);
    // B operand of stage c, slot (g, s): 16 bytes = channel 16 c + 8 g + 2 s + h, pixels px0 + 4 l31 ..  Issued from inline
    // assembly (scalar plane base + 32-bit lane offset): hipcc's wait-count insertion puts s_waitcnt vmcnt(0) at the head of
    // the K loop for loads it can see -- the whole memory latency exposed once per stage -- so these are invisible to it and
    // every consumer is ordered by the explicit counted wait below (wait_b).
    auto load_b1 = [&](int c, int g, int s) -> bvec {
      const int ci = C1_KS * c + 8 * g + 2 * s;
      const bool in_a = ci < p.Ca;
      const float* plane = in_a ? p.xa + ((size_t)n * p.Ca + ci) * p.HW + px0 : p.xb + ((size_t)n * p.Cb + (ci - p.Ca)) * p.HW + px0;
      bvec v;
      if (NE == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lvo), "s"(uniform_ptr(plane)) : "memory");
      else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(lvo), "s"(uniform_ptr(plane)) : "memory");
      return v;
    };
    // slot (g, s) was requested exactly eight loads ago (one per slot, in slot order): seven younger loads may still be in flight
    auto wait_b = [&](bvec& v) { asm volatile("s_waitcnt vmcnt(7)" : "+v"(v)); };
    f32x16 acc[NI][NE];
    {
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          asm volatile("" : "+v"(z));
          acc[i][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, z, zero16, 0, 0, 0);
        }
    }
    // ONE operand buffer: a slot's 16 bytes are requested again (for the next stage) right behind the eight MFMAs that consumed them,
    // so the eight loads of a stage are spread over its 64 MFMAs and each has 7/8 of a stage to arrive.  (All eight issued together
    // in front of the stage back the vector-memory path up and the wave -- in-order issue -- cannot reach its MFMAs: 384 us instead
    // of the 250 us of matrix time, with loads alone taking 212 us.)
    bvec bq[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[g][s] = load_b1(0, g, s);
    for (int c = 0; c < nst; ++c) {
      const int cn = c + 1 < nst ? c + 1 : c;                      // the last reload is dropped (unconditional: no phi copies)
      const float* ar = wl + afo + c * C1_KS;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 af[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) af[i] = *reinterpret_cast<const f32x4*>(ar + 32 * i * pitch + 4 * ((2 * g + h) ^ asw));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wait_b(bq[g][s]);
#ifndef MCEDM_C1_LOADS_ONLY
#pragma unroll
          for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bq[g][s][e], acc[i][e], 0, 0, 0);
#else                                                              // diagnostic build: consume the loads, skip the matrix work (wrong results)
          acc[0][0][0] += bq[g][s][0] + bq[g][s][NE - 1] + af[0][s] + af[NI - 1][s];
#endif
          bq[g][s] = load_b1(cn, g, s);
        }
      }
    }
    // the last stage's (dropped) reloads are still in flight: their registers must not be reused before they have landed
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]),
                 "+v"(bq[1][2]), "+v"(bq[1][3]));
    // ---- epilogue: + bias (+ residual), 16-byte stores: registers (e = 0 .. 3) of one accumulator row are four consecutive pixels
    const size_t obase = ((size_t)n * p.Cout) * p.HW + px0 + (unsigned)NE * l31;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + coh * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) {
          const float bv = p.bias ? p.bias[co] : 0.f;
          bvec v;
#pragma unroll
          for (int e = 0; e < NE; ++e) v[e] = acc[i][e][r] + bv;
          const size_t o = obase + (size_t)co * p.HW;
          if (p.res) v += *reinterpret_cast<const bvec*>(p.res + o);
          *reinterpret_cast<bvec*>(p.out + o) = v;
        }
      }
  }
}

static int g_c1 = -1;         // -1: default (env MCEDM_CONV1X1_REG, else on)
void set_conv1x1_reg(int enable) { g_c1 = enable; }
static int c1_env() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MCEDM_CONV1X1_REG"); env = e ? atoi(e) : 1; }
  return variant_choice(KV_CONV1X1_REG, g_c1, env);
}

// -1: not served here (shape, transform, switch); else the launch status
int try_launch_conv1x1_reg(const ConvArgs& a, int taps, hipStream_t stream) {
  const int Cin = a.Ca + a.Cb;
  const unsigned long long HW = (unsigned long long)a.H * a.W;
  if (!c1_env() || taps != 1 || a.resample != RS_NONE || a.coef || a.act || a.sk_wpk || a.gsum || a.gn_on) return -1;
  if (a.res && a.res_mode != RS_NONE) return -1;
  const int MT = a.Cout % 128 == 0 ? 128 : 64;
  const int C1_MT = MT, C1_PX = c1_px(MT);
  if (!a.xa || a.Cout % C1_MT != 0 || a.Ca % C1_KS != 0 || a.Cb % C1_KS != 0 || Cin < C1_KS || Cin > 256) return -1;
  if ((a.Cb > 0) != (a.xb != nullptr)) return -1;
  if (HW % C1_PX != 0 || HW < 1024) return -1;                        // whole 512-pixel tiles; <= 16 x 16 stays on the resident kernels
  if (4ull * a.B * a.Ca * HW >= (1ull << 32) || 4ull * a.B * (a.Cb ? a.Cb : 1) * HW >= (1ull << 32)) return -1;
  if (((reinterpret_cast<size_t>(a.xa) | reinterpret_cast<size_t>(a.xb) | reinterpret_cast<size_t>(a.out) | reinterpret_cast<size_t>(a.res)) & 15) != 0)
    return -1;
  Conv1Args p{};
  p.xa = a.xa; p.xb = a.xb; p.Ca = a.Ca; p.Cb = a.Cb;
  p.wpk = a.wpk; p.bias = a.bias; p.res = a.res; p.out = a.out;
  p.Cout = a.Cout; p.coutp = cout_padded(a.Cout); p.B = a.B; p.HW = (unsigned)HW;
  p.tiles_img = (int)(HW / C1_PX);
  p.ntiles = a.B * p.tiles_img;
  const int mblocks = a.Cout / C1_MT;
  // (a function of the image size only -- never of the batch: a sample's bits must not depend on the batch it is computed in)
  if (MT == 64 && HW < 4096) return -1;                               // one 1024-pixel tile per image: the resident kernels' job
  // persistent workgroups: one round on the chip's CUs (the weight image is loaded once per workgroup)
  int wgs = 256 / mblocks;
  if (wgs < 1) wgs = 1;
  if (wgs > p.ntiles) wgs = p.ntiles;
  p.per = ceil_div(p.ntiles, wgs);
  wgs = ceil_div(p.ntiles, p.per);
  p.nwg = wgs;
  const size_t lds = (size_t)C1_MT * (Cin + 16) * sizeof(float);
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  MCEDM_HIP_TRY(hipGetDevice(&dev));
  MCEDM_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv1x1_reg_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)conv1x1_reg_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MCEDM_HIP_TRY(hipFuncSetAttribute((const void*)(conv1x1_reg_kernel<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[dev].store(true, std::memory_order_release);
  }
  const double px = (double)a.B * (double)HW;
  ProfScope ps("conv1x1_reg_kernel", 2.0 * px * a.Cout * Cin, 4.0 * (px * (Cin + a.Cout * (a.res ? 2 : 1)) + (double)a.Cout * Cin), stream);
  static int ni_env = -1;                                  // MCEDM_C1_NI: 2 or 4 (A/B runs; the two differ in the last bits of nothing: same sums)
  if (ni_env < 0) { const char* e = getenv("MCEDM_C1_NI"); ni_env = e ? atoi(e) : 2; }      // NI = 4 measured 20 % slower (436 vs 364 us)
  if (MT == 64) hipLaunchKernelGGL((conv1x1_reg_kernel<2, 64>), dim3(wgs, mblocks), dim3(512), lds, stream, p);
  else if (ni_env == 2) hipLaunchKernelGGL(conv1x1_reg_kernel<2>, dim3(wgs, mblocks), dim3(512), lds, stream, p);
  else hipLaunchKernelGGL(conv1x1_reg_kernel<4>, dim3(wgs, mblocks), dim3(512), lds, stream, p);
  MCEDM_LAUNCH_CHECK("conv1x1_reg_kernel");
  return MCEDM_OK;
}

}  // namespace mcedm

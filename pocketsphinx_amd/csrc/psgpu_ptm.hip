// psgpu_ptm.hip -- phonetically-tied-mixture senone scoring on gfx950, batched entry.
//
// Replaces, bit-exactly, the per-frame work of ptm_mgau_frame_eval()
// (reference src/ptm_mgau.c:408-454, compallsen) for whole batches of utterances:
//
//   ptm_lane_kernel    eval_topn + eval_cb (ptm_mgau.c:87-226), frames on lanes:
//       one wavefront = one (codebook, stream) chain x 64 consecutive frames.
//       The chain's 128 Gaussians are wave-uniform and arrive through the
//       scalar cache as SGPR operands, the lane's 13 feature values sit in
//       VGPRs, the fp32 distances are computed with the reference's exact
//       sub/mul/mul/sub order (no FMA contraction: SURVEY F5) and each lane
//       keeps its five best selection keys in a max/min bubble -- no cross-lane
//       traffic.  By the closed form (psgpu_ptm_dev.h) the first four keys are
//       the reference's list unless scores tie; tied entries are flagged.
//
//   ptm_chain_kernel   the exact sequential procedure, codewords on lanes:
//       seed re-score, threshold scan in codeword order, insert-ahead-of-equals,
//       skip-if-present, emulated with wave ballots on wave-uniform list state.
//       Repairs the flagged entries after the lane kernel (persistent grid), and
//       scores whole batches when ds_ratio > 1 (frames that only re-score seeds).
//
//   ptm_senone_kernel_f3n4 / ptm_senone_kernel
//       ptm_mgau_codebook_norm + ptm_mgau_senone_eval (ptm_mgau.c:265-295,
//       :326-403).  One workgroup per frame: normalise the 126 top-N lists in
//       LDS, gather the uint8 mixture weights (slot layout: one aligned dword =
//       the four slots of a lane), log-add through the table in LDS, block
//       minimum, coalesced int16 row out.
//
// The model (3.7 MB) stays resident in L2 / Infinity Cache; HBM traffic is the
// feature rows in and the int16 score rows out (+ the chain-major top-N lists
// between the kernels).  See DESIGN.md for the roofline discussion.
#include "psgpu_ptm_dev.h"
#include <cstdlib>
#include <cstring>
#include <vector>

// ---------------------------------------------------------------------------
// ptm_chain_kernel: top-N chains, codewords on lanes (128 densities, 2 per
// lane; compile-time stream length).
//
// Work decomposition: one wavefront = one (codebook, stream) chain x one chunk
// of `chunk` consecutive global frames (utterances lie back to back).  Chunks
// are independent although the reference's top-N update is sequential in time:
// whenever a frame's closed form applies its list does not depend on the
// list carried in, so a chunk that does not start an utterance re-derives its
// incoming state by stepping back to the nearest earlier scan frame whose
// closed form holds (almost always the frame just before the chunk; at worst
// the utterance's first frame with its seed list) and replaying forward
// without publishing.  Results are therefore identical to a sequential march.
// ---------------------------------------------------------------------------

// One chunk [fbeg, fend) of one chain: recover the incoming state, march,
// publish.  Shared by the plain launch, the chunked fix-up and the per-entry
// fix-up of ptm_chain_kernel.
template <int LEN, int N>
__device__ __forceinline__
void chain_chunk(const PtmDev &p, const float *__restrict__ feats,
                 const int32_t *__restrict__ utt_off, int32_t n_utt,
                 int chain, int fbeg, int fend, int total_frames,
                 const uint8_t *__restrict__ seed_in, uint8_t *__restrict__ seed_out,
                 int32_t *__restrict__ topn_score, uint32_t *__restrict__ topn_cw, int lane)
{
    const int n_chain = p.n_chain;
    const int f = chain % p.n_feat;
    const int ds = p.ds_ratio;
    // utterance holding frame fbeg: largest u with utt_off[u] <= fbeg (empty
    // utterances share an offset; the largest such u is the non-empty one)
    int u;
    {
        int lo = 0, hi = n_utt;          // invariant: utt_off[lo] <= fbeg < utt_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (utt_off[mid] <= fbeg) lo = mid; else hi = mid;
        }
        u = lo;
    }
    int ubeg = utt_off[u], uend = utt_off[u + 1];

    // Gaussian parameters of codewords `lane` and `lane + 64`
    float m0[LEN], v0[LEN], m1[LEN], v1[LEN];
    {
        const float *mp = p.mean + ((size_t)chain * 128 + lane) * LEN;
        const float *vp = p.var + ((size_t)chain * 128 + lane) * LEN;
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            m0[j] = mp[j];
            v0[j] = vp[j];
            m1[j] = mp[64 * LEN + j];
            v1[j] = vp[64 * LEN + j];
        }
    }
    const float det0 = p.det[(size_t)chain * 128 + lane];
    const float det1 = p.det[(size_t)chain * 128 + lane + 64];
    const int32_t tag0 = 127 - lane;                     // 127 - codeword

    TopN<N> L;
    auto load_seed = [&](int utt) __attribute__((always_inline)) {
        // ptm_mgau.c:790-793 for a fresh decoder, else the carried codewords
#pragma unroll
        for (int i = 0; i < N; ++i) {
            L.cw[i] = seed_in ? (int32_t)seed_in[((size_t)utt * n_chain + chain) * N + i] : i;
            L.sc[i] = kMaxNegInt32;
        }
    };

    const float *xbase = feats + f * LEN;
    auto distances = [&](int t, float &d0, float &d1) __attribute__((always_inline)) {
        const float *x = xbase + (size_t)t * p.veclen;   // wave-uniform: scalar loads
        d0 = det0; d1 = det1;
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            const float xj = x[j];
            d0 = gau_step(d0, xj, m0[j], v0[j]);
            d1 = gau_step(d1, xj, m1[j], v1[j]);
        }
    };

    auto closed_form = [&](float d0, float d1) __attribute__((always_inline)) -> bool {
        return closed_form_top4(L, d0, d1, tag0);
    };

    // ---- incoming state
    int t = fbeg;
    load_seed(u);
    if (fbeg > ubeg) {
        int ws = fbeg - 1;
        ws -= (ws - ubeg) % ds;                 // latest scan frame before the chunk
        for (;;) {
            float d0, d1;
            distances(ws, d0, d1);
            if (__builtin_expect(closed_form(d0, d1), 1)) break;
            if (ws == ubeg) {                   // utterance start: exact step from the seed
                exact_frame_step<N>(L, d0, d1, lane, true);
                break;
            }
            ws -= ds;
        }
        t = ws + 1;                             // frames ws+1 .. fbeg-1 are replayed silently
    }

    for (; t < fend; ++t) {
        if (t == uend) {                        // next utterance starts here
            do { ++u; ubeg = uend; uend = utt_off[u + 1]; } while (uend == ubeg);
            load_seed(u);
        }
        float d0, d1;
        distances(t, d0, d1);
        const bool scan = (ds == 1) || (((t - ubeg) % ds) == 0);
        if (__builtin_expect(!(scan && closed_form(d0, d1)), 0))
            exact_frame_step<N>(L, d0, d1, lane, scan);
        if (t >= fbeg && lane == 0) {           // publish the raw list of this frame
            const size_t o = (size_t)chain * total_frames + t;      // chain-major: [chain][frame]
            *reinterpret_cast<int4 *>(topn_score + o * N) = make_int4(L.sc[0], L.sc[1], L.sc[2], L.sc[3]);
            topn_cw[o] = (uint32_t)L.cw[0] | ((uint32_t)L.cw[1] << 8) |
                         ((uint32_t)L.cw[2] << 16) | ((uint32_t)L.cw[3] << 24);
            if (seed_out && t == uend - 1) {    // carry-out of this utterance
#pragma unroll
                for (int i = 0; i < N; ++i)
                    seed_out[((size_t)u * n_chain + chain) * N + i] = (uint8_t)L.cw[i];
            }
        }
    }
}

template <int LEN, int N, int OCC>
__global__ __launch_bounds__(256, OCC)
void ptm_chain_kernel(PtmDev p, const float *__restrict__ feats,
                      const int32_t *__restrict__ utt_off, int32_t n_utt,
                      int32_t total_frames, int32_t chunk,
                      const uint8_t *__restrict__ seed_in, uint8_t *__restrict__ seed_out,
                      int32_t *__restrict__ topn_score, uint32_t *__restrict__ topn_cw,
                      const uint8_t *__restrict__ open_flags,
                      const int32_t *__restrict__ fix_count, const int32_t *__restrict__ fix_list,
                      int32_t fix_thr, int32_t fr0, int32_t fr_n)
{
    // (fr0, fr_n: the frames [fr0, fr0 + fr_n) this launch is responsible for -- a batch scored in ranges, psgpu_ptm_score_batch_dev)
    static_assert(N == 4, "codeword lists are published as one packed uint32");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(
        (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int n_chain = p.n_chain;

    // Fix-up after ptm_lane_kernel (persistent grid).  Usual case, a handful of
    // open entries: every open (frame, chain) is re-derived on its own as a
    // one-frame chunk -- one or two frames of work instead of a chunk's march.
    if (fix_list) {
        const int n = __builtin_amdgcn_readfirstlane(*fix_count);
        const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
        if (n <= fix_thr) {
            for (int e = wave; e < n; e += n_waves) {
                const int ent = __builtin_amdgcn_readfirstlane(fix_list[e]);
                const int t = ent / n_chain;
                chain_chunk<LEN, N>(p, feats, utt_off, n_utt, ent - t * n_chain, t, t + 1, total_frames,
                                    seed_in, seed_out, topn_score, topn_cw, lane);
            }
            return;
        }
        // Many open entries (e.g. a model with duplicated codewords): chunked
        // form -- open_flags[chain][frame] != 0 marks the frames whose list is
        // not the plain top-4; every chunk that holds one is marched again.
        const int n_chunks = (fr_n + chunk - 1) / chunk;
        for (int w = wave; w < n_chunks * n_chain; w += n_waves) {
            const int g = w / n_chain;
            const int chain = w - g * n_chain;
            const int fbeg = fr0 + g * chunk;
            const int fend = min(fr0 + fr_n, fbeg + chunk);
            bool any = false;
            for (int t0 = fbeg; t0 < fend; t0 += 64) {
                const int tt = t0 + lane;
                any |= __ballot(tt < fend && open_flags[(size_t)chain * total_frames + tt] != 0) != 0;
            }
            if (any)
                chain_chunk<LEN, N>(p, feats, utt_off, n_utt, chain, fbeg, fend, total_frames, seed_in, seed_out,
                                    topn_score, topn_cw, lane);
        }
        return;
    }

    const int n_chunks = (fr_n + chunk - 1) / chunk;
    if (wave >= n_chunks * n_chain)
        return;
    const int g = wave / n_chain;
    const int chain = wave - g * n_chain;
    const int fbeg = fr0 + g * chunk;
    const int fend = min(fr0 + fr_n, fbeg + chunk);

    chain_chunk<LEN, N>(p, feats, utt_off, n_utt, chain, fbeg, fend, total_frames, seed_in, seed_out,
                        topn_score, topn_cw, lane);
}

// ---------------------------------------------------------------------------
// kernel 1c: any shape (n_density <= 256, top-N 1..8, any stream lengths, any ds_ratio): the batched form of
// ptm_frame_topn_generic (psgpu_ptm_frame.hip).
//
// One wavefront = one (utterance, chain): the utterance's frames in order, the list state wave-uniform, carried from
// frame to frame as ptm_mgau_frame_eval carries it (ptm_mgau.c:435-441) -- the exact sequential procedure of eval_topn +
// eval_cb (generic_frame_step, psgpu_ptm_dev.h), no closed form, hence nothing to repair afterwards.  Codewords on lanes, four
// a lane (k * 64 + lane); the frame's feature values are wave-uniform (scalar loads); a codeword's parameters come from
// the L1 / L2-resident tables every frame (a chain's are at most 256 x len x 8 bytes).  Parallelism = utterances x chains
// (a persistent grid walks them): a batch has plenty, one utterance has n_mgau x n_feat wavefronts.  Lists leave chain-major,
// [chain][frame][N], the layout every consumer of the batched lists reads (N = 4: the same bytes as the packed word).
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256)
void ptm_batch_topn_generic(PtmDev p, const float *__restrict__ feats, const int32_t *__restrict__ utt_off, int32_t n_utt,
                            int32_t total_frames, const uint8_t *__restrict__ seed_in, uint8_t *__restrict__ seed_out,
                            int32_t *__restrict__ topn_score, uint8_t *__restrict__ topn_cw)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    const long long n_work = (long long)n_utt * p.n_chain;
    for (long long w = wave; w < n_work; w += n_waves) {
        const int u = (int)(w / p.n_chain), chain = (int)(w - (long long)u * p.n_chain);
        const int ubeg = utt_off[u], uend = utt_off[u + 1];
        if (uend <= ubeg) continue;                       // (an empty utterance: its seed stays as the caller left it)
        const int cb = chain / p.n_feat, f = chain - cb * p.n_feat;
        const int len = p.featlen[f];
        // packed [mgau][feat][density][featlen[f]] (ms_gauden.c:211-221)
        const size_t base = (size_t)cb * p.n_density * p.veclen + (size_t)p.n_density * p.featoff[f];
        const float *mp[kGenK], *vp[kGenK];
        float dt[kGenK];
#pragma unroll
        for (int k = 0; k < kGenK; ++k) {
            const int cw = min(k * 64 + lane, p.n_density - 1);
            mp[k] = p.mean + base + (size_t)cw * len; vp[k] = p.var + base + (size_t)cw * len;
            dt[k] = p.det[(size_t)chain * p.n_density + cw];
        }
        TopN<N> L;
#pragma unroll
        for (int i = 0; i < N; ++i) {       // ptm_mgau.c:790-793 for a fresh decoder, else the carried codewords
            L.cw[i] = seed_in ? (int32_t)seed_in[((size_t)u * p.n_chain + chain) * N + i] : i;
            L.sc[i] = kMaxNegInt32;
        }
        for (int t = ubeg; t < uend; ++t) {
            const float *x = feats + (size_t)t * p.veclen + p.featoff[f];      // wave-uniform: scalar loads
            float d[kGenK];
#pragma unroll
            for (int k = 0; k < kGenK; ++k) d[k] = dt[k];
            for (int j = 0; j < len; ++j) {
                const float xj = x[j];
#pragma unroll
                for (int k = 0; k < kGenK; ++k) d[k] = gau_step(d[k], xj, mp[k][j], vp[k][j]);
            }
            generic_frame_step<N, false>(L, d, d, lane, p.n_density, ((t - ubeg) % p.ds_ratio) == 0);
            if (lane == 0) {
                const size_t o = ((size_t)chain * total_frames + t) * N;
#pragma unroll
                for (int i = 0; i < N; ++i) { topn_score[o + i] = L.sc[i]; topn_cw[o + i] = (uint8_t)L.cw[i]; }
            }
        }
        if (seed_out && lane == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) seed_out[((size_t)u * p.n_chain + chain) * N + i] = (uint8_t)L.cw[i];
        }
    }
}

// ---------------------------------------------------------------------------
// kernel 1b: frames on lanes (ds_ratio == 1).
//
// One wavefront = one chain x 64 consecutive frames, one frame per lane.  The
// chain's Gaussian parameters are wave-uniform and arrive through the scalar
// cache as SGPR operands; the lane's 13 feature values sit in VGPRs; the 128
// codewords are visited in index order.  There is no cross-lane traffic at
// all: each lane keeps its five best selection keys (clamped truncated score
// << 7 | 127 - codeword) sorted with a max/min bubble (10 VALU ops per
// codeword).  By the closed form (psgpu_ptm_dev.h) the first four keys ARE the
// reference's list whenever the five best scores are pairwise distinct and in
// range; otherwise the (frame, chain) entry is flagged and ptm_chain_kernel,
// launched next in fix-up mode, re-derives those frames with the exact
// sequential procedure.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t med3_i32(int32_t a, int32_t b, int32_t c)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef float ptm_f2 __attribute__((ext_vector_type(2)));

template <int LEN, int FPL>                     // FPL = frames per lane
__global__ __launch_bounds__(256, FPL >= 4 ? 4 : 8)
void ptm_lane_kernel(PtmDev p, const float *__restrict__ feats, int32_t total_frames,
                     const int32_t *__restrict__ utt_off, int32_t n_utt,
                     uint8_t *__restrict__ seed_out,
                     int32_t *__restrict__ topn_score, uint32_t *__restrict__ topn_cw,
                     uint8_t *__restrict__ open_flags,
                     int32_t *__restrict__ fix_count, int32_t *__restrict__ fix_list, int32_t fix_cap, int32_t fr0, int32_t fr_n)
{
    // (this launch: the frames [fr0, fr0 + fr_n) of the batch; the lists' layout is the whole batch's)
    const int lane = threadIdx.x & 63;
    const int n_tiles = (fr_n + 64 * FPL - 1) / (64 * FPL);
    // A workgroup = four consecutive tiles of ONE chain (its wavefronts stream the same parameters through the scalar
    // cache).  Which (chain, tiles) a workgroup gets decides how often the feature vectors come from HBM: with the chains
    // outermost every chain swept all of them (PMC, r02: 30 GB read for 240 MB of features).  Now XCD x (workgroups go to
    // the XCDs round robin) owns the x-th eighth of the tiles and walks it in blocks of kLaneSuper workgroups' tiles -- 16 k
    // frames, 2.5 MB of features, which its 4 MB L2 keeps -- all chains of a block before the next block.
    constexpr int kLaneSuper = 64;
    const int n_units = (n_tiles + 3) >> 2, upx = (n_units + 7) >> 3;      // units of four tiles; units per XCD
    const int xcd = (int)(blockIdx.x & 7), i = (int)(blockIdx.x >> 3);
    const int sb = i / (kLaneSuper * p.n_chain), rem = i - sb * kLaneSuper * p.n_chain;
    const int s_cur = min(kLaneSuper, upx - sb * kLaneSuper);
    if (s_cur <= 0) return;
    const int chain = __builtin_amdgcn_readfirstlane(rem / s_cur);
    const int unit = xcd * upx + sb * kLaneSuper + (rem - chain * s_cur);
    const int tile = __builtin_amdgcn_readfirstlane(unit * 4 + (int)(threadIdx.x >> 6));
    if (chain >= p.n_chain || unit >= n_units || tile >= n_tiles)
        return;
    const int f = chain % p.n_feat;

    int t[FPL];
    bool valid[FPL];
    float x[FPL][LEN];
#pragma unroll
    for (int q = 0; q < FPL; ++q) {
        t[q] = fr0 + (tile * FPL + q) * 64 + lane;
        valid[q] = t[q] < fr0 + fr_n;
        const int tl = valid[q] ? t[q] : fr0 + fr_n - 1;
        const float *xp = feats + (size_t)tl * p.veclen + f * LEN;
#pragma unroll
        for (int j = 0; j < LEN; ++j) x[q][j] = xp[j];
    }
    // frames as pairs: the halves of one packed operand live in one aligned register pair from the start
    ptm_f2 xq[FPL / 2 > 0 ? FPL / 2 : 1][LEN];
    if (FPL % 2 == 0) {
#pragma unroll
        for (int q = 0; q < FPL; q += 2)
#pragma unroll
            for (int j = 0; j < LEN; ++j) xq[q / 2][j] = (ptm_f2){ x[q][j], x[q + 1][j] };
    }
    const float *mean = p.mean + (size_t)chain * 128 * LEN;     // wave-uniform: scalar loads
    const float *var = p.var + (size_t)chain * 128 * LEN;
    const float *det = p.det + (size_t)chain * 128;

    int32_t k0[FPL], k1[FPL], k2[FPL], k3[FPL], k4[FPL];
#pragma unroll
    for (int q = 0; q < FPL; ++q) k0[q] = k1[q] = k2[q] = k3[q] = k4[q] = kMaxNegInt32;
    auto insert = [&](int q, float d, int cw) {
        const float c = __builtin_amdgcn_fmed3f(d, (float)kKeyLo, (float)kKeyHi);
        const int32_t k = ((int32_t)c << 7) | (127 - cw);
        // insertion into the sorted five: new_i = med3(old_{i-1}, old_i, k), all five from the
        // OLD values -- one max and four v_med3_i32, independent of each other
        const int32_t n4 = med3_i32(k3[q], k4[q], k), n3 = med3_i32(k2[q], k3[q], k),
                      n2 = med3_i32(k1[q], k2[q], k), n1 = med3_i32(k0[q], k1[q], k);
        k0[q] = max(k0[q], k); k1[q] = n1; k2[q] = n2; k3[q] = n3; k4[q] = n4;
    };
#pragma unroll 2
    for (int cw = 0; cw < 128; ++cw) {
        const float *m = mean + cw * LEN, *v = var + cw * LEN;
        const float dt = det[cw];
        if (FPL % 2 == 0) {
            // two frames per lane as the halves of packed single-precision operations (v_pk_add_f32 / v_pk_mul_f32: both
            // halves rounded as the scalar operations are, the parameters broadcast from SGPRs): the distance is 4 VALU
            // operations per dimension, and this kernel runs at the VALU issue rate
#pragma unroll
            for (int q = 0; q < FPL; q += 2) {
                ptm_f2 d = { dt, dt };
#pragma unroll
                for (int j = 0; j < LEN; ++j) {
                    const ptm_f2 diff = xq[q / 2][j] - (ptm_f2){ m[j], m[j] };
                    const ptm_f2 sq = diff * diff;
                    const ptm_f2 c = sq * (ptm_f2){ v[j], v[j] };
                    d = d - c;
                }
                insert(q, d.x, cw); insert(q + 1, d.y, cw);
            }
        }
        else {
#pragma unroll
            for (int q = 0; q < FPL; ++q) {
                float d = dt;
#pragma unroll
                for (int j = 0; j < LEN; ++j)
                    d = gau_step(d, x[q][j], m[j], v[j]);
                insert(q, d, cw);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < FPL; ++q) {
        const int32_t s0 = k0[q] >> 7, s1 = k1[q] >> 7, s2 = k2[q] >> 7, s3 = k3[q] >> 7, s4 = k4[q] >> 7;
        const bool open = (s0 == s1) | (s1 == s2) | (s2 == s3) | (s3 == s4) | (s0 >= kKeyHi) | (s3 <= kKeyLo);
        if (valid[q]) {
            const int tt = t[q];
            const size_t o = (size_t)chain * total_frames + tt;         // chain-major: a wave stores 1 KB contiguous
            *reinterpret_cast<int4 *>(topn_score + o * 4) = make_int4(s0, s1, s2, s3);
            const uint32_t c0 = 127 - (k0[q] & 127), c1 = 127 - (k1[q] & 127), c2 = 127 - (k2[q] & 127),
                           c3 = 127 - (k3[q] & 127);
            topn_cw[o] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
            open_flags[(size_t)chain * total_frames + tt] = open ? 1 : 0;
            if (open) {
                const int32_t idx = atomicAdd(fix_count, 1);
                if (idx < fix_cap) fix_list[idx] = tt * p.n_chain + chain;
            }
            if (seed_out && !open) {
                // carry-out of an utterance = the list of its last frame (acmod never
                // resets it); flagged frames are written by the fix-up
                int lo = 0, hi = n_utt;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (utt_off[mid] <= tt) lo = mid; else hi = mid;
                }
                if (utt_off[lo + 1] - 1 == tt) {
                    uint8_t *so = seed_out + ((size_t)lo * p.n_chain + chain) * 4;
                    so[0] = (uint8_t)c0; so[1] = (uint8_t)c1; so[2] = (uint8_t)c2; so[3] = (uint8_t)c3;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// kernel 2: normalise + mixture-weight log-sum, one workgroup per frame
// ---------------------------------------------------------------------------
constexpr int kSenThreads = 256;

template <int N>
__global__ __launch_bounds__(kSenThreads)
void ptm_senone_kernel(PtmDev p, const int32_t *__restrict__ topn_score,
                       const uint8_t *__restrict__ topn_cw,
                       int16_t *__restrict__ senscr, int32_t *__restrict__ best_out,
                       uint32_t flags, int32_t total_frames)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [n_sen] int16 scores | [n_chain*N] cw | [n_chain*N] norm score |
    //         [256] log-add table | [16] int norm | [8] int red
    int16_t *s_out = reinterpret_cast<int16_t *>(smem);
    const int out_bytes = ((p.n_sen * 2 + 15) / 16) * 16;
    const int list_bytes = ((p.n_chain * N + 15) / 16) * 16;
    uint8_t *s_cw = smem + out_bytes;
    uint8_t *s_sc = s_cw + list_bytes;
    uint8_t *s_la = s_sc + list_bytes;
    int32_t *s_norm = reinterpret_cast<int32_t *>(s_la + 256);
    int32_t *s_red = s_norm + 16;

    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int n_ent = p.n_chain * N;
    // lists are chain-major: entry (chain c, rank k) of this frame
    auto ent = [&](int i) { return ((size_t)(i / N) * total_frames + frame) * N + (i % N); };

    if (tid < p.n_feat) s_norm[tid] = kWorstScore;
    if (tid < 8) s_red[tid] = 0x7fffffff;
    for (int i = tid; i < 256; i += kSenThreads)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    __syncthreads();

    // ptm_mgau_codebook_norm (:265-295): norm[f] = max over codebooks of
    // (best score >> 10)
    for (int i = tid; i < p.n_chain; i += kSenThreads) {
        const int f = i % p.n_feat;
        atomicMax(&s_norm[f], topn_score[ent(i * N)] >> kSenscrShift);
    }
    __syncthreads();
    for (int i = tid; i < n_ent; i += kSenThreads) {
        const int f = (i / N) % p.n_feat;
        int32_t v = topn_score[ent(i)] >> kSenscrShift;
        v = -(v - s_norm[f]);
        if (v > kMaxNegAscr) v = kMaxNegAscr;
        s_sc[i] = (uint8_t)v;
        s_cw[i] = topn_cw[ent(i)];
    }
    __syncthreads();

    // ptm_mgau_senone_eval (:326-403)
    int32_t mybest = 0x7fffffff;
    for (int s = tid; s < p.n_sen; s += kSenThreads) {
        const int cb = p.sen2cb[s];
        int32_t ascore = 0;
        for (int f = 0; f < p.n_feat; ++f) {
            const int li = (cb * p.n_feat + f) * N;
            const uint8_t *wrow = p.mixw + (size_t)f * p.n_density * p.n_sen + s;
            int32_t fden = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int32_t y = (int32_t)wrow[(size_t)s_cw[li + k] * p.n_sen] + s_sc[li + k];
                if (k == 0)
                    fden = y;
                else {
                    // fast_logmath_add (tied_mgau_common.h:106-125)
                    const int32_t lo_ = min(fden, y);
                    const int32_t d = max(fden, y) - lo_;
                    fden = lo_ - (d < 256 ? (int32_t)s_la[d] : 0);
                }
            }
            ascore += fden;
        }
        s_out[s] = (int16_t)ascore;
        mybest = min(mybest, ascore);
    }
    // block minimum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mybest = min(mybest, __shfl_xor(mybest, off));
    if ((tid & 63) == 0) atomicMin(&s_red[0], mybest);
    __syncthreads();
    const int32_t best = s_red[0];
    if (best_out && tid == 0) best_out[frame] = best;
    const int32_t sub = (flags & PSGPU_PTM_RAW_SCORES) ? 0 : best;
    int16_t *orow = senscr + (size_t)frame * p.n_sen;
    for (int s = tid; s < p.n_sen; s += kSenThreads)
        orow[s] = (int16_t)(s_out[s] - sub);    // int16 store as in :398-400
}

// ---------------------------------------------------------------------------
// kernel 2, fast variant (3 streams, top-4).
//
// The model is re-laid out once at creation into "slots": senones sorted by
// codebook, every codebook padded to a multiple of four slots (pad slots are
// never stored).  A lane owns groups of four consecutive slots; the group
// shares its 12 (stream, rank) mixture-weight rows, so the four weights of a
// row arrive as ONE aligned dword from the slot-ordered, row-padded copy of
// mixw, and there is no divergence between lanes.  All 12 loads of a group
// are in flight before its first log-add.  Scores are staged in LDS in
// senone order, the block minimum is subtracted, and the row leaves in
// coalesced dword stores.
// ---------------------------------------------------------------------------
constexpr int kSenMaxIters = 8;
constexpr int kLaSize = 512;                    // log-add table padded with zeros

// fast_logmath_add (tied_mgau_common.h:106-125): min(x,y) - T[|x-y|].  With
// uint8 weights and scores <= 96 the index stays below 512; the LDS copy of
// the table is zero beyond the reference's entries.
__device__ __forceinline__ int32_t logadd8_lds(const uint8_t *s_la, int32_t x, int32_t y)
{
    const int32_t lo = min(x, y);
    const int32_t d = max(x, y) - lo;
    return lo - (int32_t)s_la[d];
}

// SAD = true: the chain runs on values biased by kSadBias (the scores' bytes carry it), so that every intermediate is >= 0 and
// |x - y| is ONE unsigned instruction (v_sad_u32) instead of max - min; the host selects it when the model's log-add table
// keeps the running sums above -kSadBias (3 x its largest entry <= kSadBias; en-us: 3 x 7).  The staged row and the block
// minimum carry 3 x kSadBias, taken off with the normaliser in the row's last pass.
constexpr int32_t kSadBias = 64;
// DIRECT = true (un-normalised rows wanted, an even number of senones): a lane's scores go from its registers straight to the row in
// device memory -- the slot layout puts even senones on even slots (psgpu_ptm_model_create), so two consecutive senones of a lane
// are one aligned 32-bit store -- and the 10 KB of LDS that stage a frame's row for the normaliser's subtraction are not asked
// for: beside a resident search kernel that leaves a compute unit 34 KB of LDS, the staged form gets two workgroups there, this one
// as many as there are wave slots.
template <int ITERS, int kSenFr, bool SAD, bool DIRECT = false>      // kSenFr = frames per workgroup
__global__ __launch_bounds__(512)
void ptm_senone_kernel_f3n4(PtmDev p, const int32_t *__restrict__ topn_score,
                            const uint32_t *__restrict__ topn_cw,
                            int16_t *__restrict__ senscr, int32_t *__restrict__ best_out,
                            uint32_t flags, int32_t total_frames, int32_t *__restrict__ zero_word, int32_t fr0, int32_t fr_n)
{
    constexpr int N = 4, NF = 3, MAXC = 256;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0)
        *zero_word = 0;                         // open-entry counter of the top-N pass, for the next call
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int out_stride = ((p.n_sen * 2 + 15) / 16) * 8;         // int16 entries per staged row
    int16_t *s_out = reinterpret_cast<int16_t *>(smem);          // [kSenFr][out_stride]
    __shared__ uint32_t s_cw32[kSenFr][MAXC];        // packed codewords per chain (n_chain <= MAXC)
    __shared__ uint32_t s_sc32[kSenFr][MAXC];        // packed normalised scores per chain
    __shared__ uint8_t s_la[kLaSize];
    __shared__ int32_t s_norm[kSenFr][NF];
    __shared__ int32_t s_best[kSenFr];

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    // workgroups go to the eight XCDs round robin (MI355X_MICROARCH "Workgroup dispatch"); the lists are chain-major, so a
    // 128-byte line of them holds eight consecutive frames of one chain: with frame = workgroup index every XCD's L2 fetched
    // every line (PMC, r02: 38 GB read per 1.5 M frames for 4 GB of lists).  XCD x now takes the x-th eighth of the frames:
    // consecutive frames share an L2.  (The grid is a multiple of eight; workgroups past the end leave.)
    const int per_xcd = (int)gridDim.x >> 3;
    const int blk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int f0 = fr0 + blk * kSenFr;
    if (f0 >= fr0 + fr_n) return;
    const int nf = min(kSenFr, fr0 + fr_n - f0);

    if (tid < kSenFr * NF) s_norm[tid / NF][tid % NF] = kWorstScore;
    if (tid < kSenFr) s_best[tid] = 0x7fffffff;
    for (int i = tid; i < kLaSize; i += nthr)
        s_la[i] = (i < p.logadd8_size) ? p.logadd8[i] : 0;
    __syncthreads();

    // ptm_mgau_codebook_norm (:265-295).  Lists are chain-major, so the kSenFr
    // frames of one chain are 64 contiguous bytes of scores + 16 of codewords.
    int4 sc[kSenFr];
    uint32_t cw[kSenFr];
    const int i = tid;
    if (i < p.n_chain) {
        const size_t lo = (size_t)i * total_frames + f0;
#pragma unroll
        for (int fr = 0; fr < kSenFr; ++fr) {
            if (fr < nf) {
                sc[fr] = *reinterpret_cast<const int4 *>(topn_score + (lo + fr) * N);
                cw[fr] = topn_cw[lo + fr];
                atomicMax(&s_norm[fr][i % NF], sc[fr].x >> kSenscrShift);
            }
        }
    }
    __syncthreads();
    if (i < p.n_chain) {
#pragma unroll
        for (int fr = 0; fr < kSenFr; ++fr) {
            if (fr < nf) {
                const int32_t norm = s_norm[fr][i % NF];
                const int32_t a = min(kMaxNegAscr, -((sc[fr].x >> kSenscrShift) - norm));
                const int32_t b = min(kMaxNegAscr, -((sc[fr].y >> kSenscrShift) - norm));
                const int32_t c = min(kMaxNegAscr, -((sc[fr].z >> kSenscrShift) - norm));
                const int32_t d = min(kMaxNegAscr, -((sc[fr].w >> kSenscrShift) - norm));
                s_sc32[fr][i] = ((uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24))
                                + (SAD ? 0x01010101u * (uint32_t)kSadBias : 0u);       // (<= 96 + 64 a byte)
                s_cw32[fr][i] = cw[fr];
            }
        }
    }
    __syncthreads();

    // ptm_mgau_senone_eval (:326-403), slot order, kSenFr frames back to back
    int32_t mybest[kSenFr];
#pragma unroll
    for (int fr = 0; fr < kSenFr; ++fr) {
        mybest[fr] = 0x7fffffff;
        if (fr < nf) {
            int16_t *orow_s = s_out + fr * out_stride;
            int16_t *orow_d = senscr + (size_t)(f0 + fr) * p.n_sen;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int g = tid + it * nthr;
                if (g < p.n_groups) {
                    const uint32_t cb = p.group_cb[g];
                    const uint32_t slot = (uint32_t)g << 2;
                    uint32_t w[NF][N], nsc[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const uint32_t c4 = s_cw32[fr][cb * NF + f];
                        nsc[f] = s_sc32[fr][cb * NF + f];
                        // (offsets as 24-bit multiply-adds -- v_mad_u32_u24, full rate -- on a 32-bit offset from the table's base:
                        //  written as row * stride in 64 bits the compiler emits v_mad_u64_u32, a quarter-rate instruction, twelve
                        //  times per group; the host checked that the table is smaller than 2^24 bytes)
                        const uint32_t base = __umul24((uint32_t)f * (uint32_t)p.n_density, (uint32_t)p.slot_stride) + slot;
#pragma unroll
                        for (int k = 0; k < N; ++k) {
                            uint32_t off;
#if defined(__HIP_DEVICE_COMPILE__)
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(__builtin_amdgcn_ubfe(c4, 8 * k, 8)), "s"((uint32_t)p.slot_stride), "v"(base));
#else
                            off = ((c4 >> (8 * k)) & 0xff) * (uint32_t)p.slot_stride + base;
#endif
                            w[f][k] = *reinterpret_cast<const uint32_t *>(p.mixw_slot + off);
                        }
                    }
                    const uint2 sen2 = *reinterpret_cast<const uint2 *>(p.slot_sen + slot);   // 4 x uint16
                    // the 4 slots x 3 streams are 12 independent log-add chains: advance them in
                    // lock-step so that 12 table reads are in flight per wait (pad slots are
                    // computed too -- their weights are valid bytes -- and dropped at the store)
                    int32_t fden[4][NF];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int f = 0; f < NF; ++f)
                            fden[b][f] = (int32_t)((w[f][0] >> (8 * b)) & 0xff) + (int32_t)(nsc[f] & 0xff);
#pragma unroll
                    for (int k = 1; k < N; ++k) {
                        int32_t lo_[4][NF], dd[4][NF];
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int f = 0; f < NF; ++f) {
                                const int32_t y = (int32_t)((w[f][k] >> (8 * b)) & 0xff) +
                                                  (int32_t)((nsc[f] >> (8 * k)) & 0xff);
                                lo_[b][f] = min(fden[b][f], y);
                                // (signed: a running sum goes below zero when a small weight meets the best codeword --
                                //  min - T[d] with min < T[d]; an unsigned |a - b| would then index far outside the table.
                                //  The biased form keeps every value >= 0: one v_sad_u32)
#if defined(__HIP_DEVICE_COMPILE__)
                                if (SAD) asm("v_sad_u32 %0, %1, %2, 0" : "=v"(dd[b][f]) : "v"(fden[b][f]), "v"(y));
                                else
#endif
                                dd[b][f] = max(fden[b][f], y) - lo_[b][f];
                            }
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int f = 0; f < NF; ++f)
                                fden[b][f] = lo_[b][f] - (int32_t)s_la[dd[b][f]];   // fast_logmath_add (tied_mgau_common.h:106-125)
                    }
                    if (DIRECT) {
                        int32_t a4[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) a4[b] = fden[b][0] + fden[b][1] + fden[b][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t pr = h ? sen2.y : sen2.x, s0 = pr & 0xffff, s1 = pr >> 16;
                            const int32_t v0 = a4[2 * h] - (SAD ? 3 * kSadBias : 0), v1 = a4[2 * h + 1] - (SAD ? 3 * kSadBias : 0);
                            if (s0 != 0xffff) mybest[fr] = min(mybest[fr], a4[2 * h]);
                            if (s1 != 0xffff) mybest[fr] = min(mybest[fr], a4[2 * h + 1]);
                            if (s0 != 0xffff && s1 == s0 + 1 && !(s0 & 1))
                                __builtin_nontemporal_store(((uint32_t)v0 & 0xffffu) | ((uint32_t)v1 << 16), reinterpret_cast<uint32_t *>(orow_d) + (s0 >> 1));
                            else {
                                if (s0 != 0xffff) orow_d[s0] = (int16_t)v0;
                                if (s1 != 0xffff) orow_d[s1] = (int16_t)v1;
                            }
                        }
                    }
                    else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int32_t ascore = fden[b][0] + fden[b][1] + fden[b][2];
                        const uint32_t sen = ((b < 2 ? sen2.x : sen2.y) >> (16 * (b & 1))) & 0xffff;
                        if (sen != 0xffff) {            // pad slots are dropped
                            orow_s[sen] = (int16_t)ascore;
                            mybest[fr] = min(mybest[fr], ascore);
                        }
                    }
                    }
                }
            }
        }
    }
    // per-frame block minimum
#pragma unroll
    for (int fr = 0; fr < kSenFr; ++fr) {
        int32_t mb = mybest[fr];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            mb = min(mb, __shfl_xor(mb, off));
        if ((tid & 63) == 0 && fr < nf) atomicMin(&s_best[fr], mb);
    }
    __syncthreads();
    for (int fr = 0; fr < nf; ++fr) {
        const int frame = f0 + fr;
        const int32_t best = s_best[fr] - (SAD ? 3 * kSadBias : 0);
        if (best_out && tid == 0) best_out[frame] = best;
        if (DIRECT) continue;                       // (the rows are out already)
        const int32_t sub = ((flags & PSGPU_PTM_RAW_SCORES) ? 0 : best) + (SAD ? 3 * kSadBias : 0);
        int16_t *orow = senscr + (size_t)frame * p.n_sen;
        const int16_t *srow = s_out + fr * out_stride;
        if ((p.n_sen & 1) == 0) {                   // rows stay 4-byte aligned
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(srow);
            uint32_t *o32 = reinterpret_cast<uint32_t *>(orow);
            for (int j = tid; j < (p.n_sen >> 1); j += nthr) {
                const uint32_t v = s32[j];
                const uint32_t a = ((v & 0xffff) - (uint32_t)sub) & 0xffff;     // int16 wrap as in :398-400
                const uint32_t b = ((v >> 16) - (uint32_t)sub) & 0xffff;
                __builtin_nontemporal_store(a | (b << 16), &o32[j]);     // (streamed out once: must not push the weight table out of L2)
            }
        }
        else {
            for (int j = tid; j < p.n_sen; j += nthr)
                orow[j] = (int16_t)(srow[j] - sub);
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int upload(T **dst, const T *src, size_t n)
{
    PSGPU_HIP(hipMalloc((void **)dst, n * sizeof(T) ? n * sizeof(T) : 1));
    PSGPU_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PSGPU_OK;
}

// the weights once more, senone-major: [n_sen][n_feat][dens_stride], dens_stride = n_density rounded up to 64.  A kernel that
// scores single senones (psgpu_sen_dev.h) then finds the topn x n_feat weights of a senone in n_feat cache lines instead of
// topn x n_feat of them (in the reference's [stream][density][senone] order every weight of a senone is n_sen bytes from
// the next)
static int upload_by_senone(psgpu_ptm_model_t *m, const uint8_t *mixw)
{
    const size_t ds = ((size_t)m->n_density + 63) / 64 * 64, per = (size_t)m->n_feat * ds;
    std::vector<uint8_t> t((size_t)m->n_sen * per, 0);
    for (int f = 0; f < m->n_feat; ++f)
        for (int d = 0; d < m->n_density; ++d) {
            const uint8_t *row = mixw + ((size_t)f * m->n_density + d) * m->n_sen;
            for (int s = 0; s < m->n_sen; ++s) t[(size_t)s * per + (size_t)f * ds + d] = row[s];
        }
    return upload(&m->mixw_sen, t.data(), t.size());
}

extern "C" {

int psgpu_ptm_model_create(psgpu_ptm_model_t **out,
                           int32_t n_mgau, int32_t n_feat, int32_t n_density,
                           const int32_t *featlen, int32_t n_sen, int32_t topn,
                           int32_t ds_ratio,
                           const float *mean, const float *var, const float *det,
                           const uint8_t *mixw, const uint8_t *sen2cb,
                           const uint8_t *logadd8, int32_t logadd8_size)
{
    PSGPU_REQUIRE(out && featlen && mean && var && det && mixw && sen2cb && logadd8,
                  "psgpu_ptm_model_create: NULL argument");
    PSGPU_REQUIRE(n_mgau > 0 && n_mgau <= 256, "n_mgau %d outside 1..256 (ptm_mgau.c:838)", n_mgau);
    PSGPU_REQUIRE(n_feat > 0 && n_feat <= 16, "n_feat %d outside 1..16", n_feat);
    PSGPU_REQUIRE(n_density >= 1 && n_density <= 256, "n_density %d outside 1..256", n_density);
    PSGPU_REQUIRE(topn >= 1 && topn <= PSGPU_MAX_TOPN && topn <= n_density, "topn %d outside 1..%d", topn, PSGPU_MAX_TOPN);
    PSGPU_REQUIRE(ds_ratio >= 1, "ds_ratio %d < 1", ds_ratio);
    PSGPU_REQUIRE(n_sen > 0 && n_sen < 32768, "n_sen %d outside 1..32767", n_sen);
    PSGPU_REQUIRE(logadd8_size >= 256, "log-add table has %d < 256 entries (logmath.c:112)", logadd8_size);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;

    psgpu_ptm_model_t *m = new psgpu_ptm_model_t();
    m->n_mgau = n_mgau; m->n_feat = n_feat; m->n_density = n_density;
    m->n_sen = n_sen; m->topn = topn; m->ds_ratio = ds_ratio;
    m->n_chain = n_mgau * n_feat;
    m->veclen = 0; m->uniform_len = featlen[0];
    for (int f = 0; f < n_feat; ++f) {
        m->featlen[f] = featlen[f];
        m->featoff[f] = m->veclen;
        m->veclen += featlen[f];
        if (featlen[f] != featlen[0]) m->uniform_len = 0;
    }
    if (m->veclen > 64) {
        psgpu_set_error("feature vector of %d floats exceeds 64", m->veclen);
        delete m;
        return PSGPU_EINVAL;
    }
    // the batched kernels are specialised; every other shape is served by the per-call entry
    m->fast_shape = (n_density == 128 && topn == 4 && m->uniform_len == 13);
    hipGetDevice(&m->device);
    const size_t npar = (size_t)n_mgau * n_density * m->veclen;
    if ((rc = upload(&m->mean, mean, npar)) || (rc = upload(&m->var, var, npar)) ||
        (rc = upload(&m->det, det, (size_t)m->n_chain * n_density)) ||
        (rc = upload(&m->mixw, mixw, (size_t)n_feat * n_density * n_sen)) ||
        (rc = upload_by_senone(m, mixw)) ||
        (rc = upload(&m->sen2cb, sen2cb, (size_t)n_sen)) ||
        (rc = upload(&m->logadd8, logadd8, (size_t)logadd8_size))) {
        psgpu_ptm_model_free(m);
        return rc;
    }
    m->logadd8_size = logadd8_size;
    m->la_max = 0;
    for (int i = 0; i < logadd8_size; ++i) m->la_max = std::max<int32_t>(m->la_max, logadd8[i]);
    m->h_sen2cb = (uint8_t *)malloc((size_t)n_sen);
    memcpy(m->h_sen2cb, sen2cb, (size_t)n_sen);
    {
        // slot layout for the fast senone kernel: senones grouped by codebook,
        // each codebook padded to a multiple of 4 slots
        std::vector<std::vector<int>> by_cb(256);
        for (int i = 0; i < n_sen; ++i) by_cb[sen2cb[i]].push_back(i);
        std::vector<uint16_t> slot_sen;
        std::vector<uint8_t> group_cb;
        // (a senone sits on a slot of its own parity -- a pad where the parities part, i.e. where a codebook's senone ids jump: en-us's
        //  are its three CI senones and one range of CD senones -- so that the direct-store form of the fast senone kernel writes two
        //  consecutive senones of a lane as one aligned 32-bit word; a model whose ids would need pads on more than a quarter of
        //  the slots keeps the plain layout: the kernel checks each pair and falls back to 16-bit stores)
        for (int pass = 0; pass < 2; ++pass) {
            slot_sen.clear(); group_cb.clear();
            size_t pads = 0;
            for (int cb = 0; cb < 256; ++cb) {
                if (by_cb[cb].empty()) continue;
                for (int i : by_cb[cb]) {
                    if (pass == 0 && ((slot_sen.size() ^ (size_t)i) & 1)) { slot_sen.push_back(0xffff); ++pads; }
                    slot_sen.push_back((uint16_t)i);
                }
                while (slot_sen.size() % 4) slot_sen.push_back(0xffff);
                while (group_cb.size() < slot_sen.size() / 4) group_cb.push_back((uint8_t)cb);
            }
            if (pass == 0 && 4 * pads <= slot_sen.size()) break;
        }
        m->n_groups = (int32_t)group_cb.size();
        const size_t n_slots = slot_sen.size();
        m->slot_stride = (int32_t)(((n_slots + 63) / 64) * 64);
        const size_t rows = (size_t)n_feat * n_density;
        if (rows * (size_t)m->slot_stride >= (1u << 24)) m->fast_shape = 0;      // (the fast senone kernel's 24-bit table offsets)
        std::vector<uint8_t> ms(rows * m->slot_stride, 255);
        for (size_t r = 0; r < rows; ++r)
            for (size_t sl = 0; sl < n_slots; ++sl)
                if (slot_sen[sl] != 0xffff)
                    ms[r * m->slot_stride + sl] = mixw[r * (size_t)n_sen + slot_sen[sl]];
        if ((rc = upload(&m->mixw_slot, ms.data(), ms.size())) ||
            (rc = upload(&m->group_cb, group_cb.data(), group_cb.size())) ||
            (rc = upload(&m->slot_sen, slot_sen.data(), slot_sen.size()))) {
            psgpu_ptm_model_free(m);
            return rc;
        }
    }
    *out = m;
    return PSGPU_OK;
}

void psgpu_ptm_model_free(psgpu_ptm_model_t *m)
{
    if (!m) return;
    hipFree(m->mean); hipFree(m->var); hipFree(m->det);
    hipFree(m->mixw); hipFree(m->sen2cb); hipFree(m->logadd8);
    hipFree(m->mixw_slot); hipFree(m->group_cb); hipFree(m->slot_sen); hipFree(m->mixw_sen);
    free(m->h_sen2cb);
    for (PtmWorkspace &w : m->ws) {
        hipFree(w.open_flags); hipFree(w.fix_list);
        if (w.aux) { hipStreamSynchronize(w.aux); hipStreamDestroy(w.aux); for (hipEvent_t e : w.ev) if (e) hipEventDestroy(e); }
    }
    for (int i = 0; i < 4; ++i) if (m->ev[i]) hipEventDestroy(m->ev[i]);
    delete m;
}

// device pointers to the tables, for kernels outside this file that score senones themselves (psgpu_flat.hip)
int psgpu_ptm_model_view(const psgpu_ptm_model_t *m, psgpu_ptm_view_t *out)
{
    PSGPU_REQUIRE(m && out, "psgpu_ptm_model_view: NULL argument");
    PSGPU_REQUIRE(m->ds_ratio == 1, "psgpu_ptm_model_view: a model with -ds %d re-scores carried lists on most frames; not supported by the in-kernel scorer", m->ds_ratio);
    memset(out, 0, sizeof *out);
    out->mean = m->mean; out->var = m->var; out->det = m->det;
    out->mixw = m->mixw; out->sen2cb = m->sen2cb; out->logadd8 = m->logadd8; out->mixw_sen = m->mixw_sen;
    out->n_mgau = m->n_mgau; out->n_feat = m->n_feat; out->n_density = m->n_density; out->n_sen = m->n_sen;
    out->veclen = m->veclen; out->topn = m->topn; out->logadd8_size = m->logadd8_size;
    for (int f = 0; f < 16; ++f) { out->featlen[f] = m->featlen[f]; out->featoff[f] = m->featoff[f]; }
    return PSGPU_OK;
}

// the open-entry flags the last psgpu_ptm_score_batch_dev on `stream` left ([n_chain][total_frames], chain-major like the
// lists; 1 = the list is not the seed-independent top-N: ties among the best five, or out of range)
int psgpu_ptm_batch_open_flags(psgpu_ptm_model_t *m, void *stream, const uint8_t **flags_dev)
{
    PSGPU_REQUIRE(m && flags_dev, "psgpu_ptm_batch_open_flags: NULL argument");
    PtmWorkspace *ws = ptm_workspace(m, (hipStream_t)stream, false);
    PSGPU_REQUIRE(ws && ws->open_flags, "psgpu_ptm_batch_open_flags: no batch has been scored on this stream");
    *flags_dev = ws->open_flags;
    return PSGPU_OK;
}

int32_t psgpu_ptm_n_sen(const psgpu_ptm_model_t *m) { return m->n_sen; }
int32_t psgpu_ptm_n_chain(const psgpu_ptm_model_t *m) { return m->n_chain; }
int32_t psgpu_ptm_veclen(const psgpu_ptm_model_t *m) { return m->veclen; }
int32_t psgpu_ptm_topn(const psgpu_ptm_model_t *m) { return m->topn; }

constexpr int kPtmMaxRanges = 16;

// The top-N pass of the frames [fr0, fr0 + fr_n) of a batch on `st` (lists in the whole batch's layout): the frames-on-lanes kernel
// and the exact fix-up behind it, with range r's share of the workspace's open-entry list and its counter; or the chain kernel alone
// (-ds > 1, PSGPU_NO_LANE_KERNEL); any other shape: the exact sequential procedure per (utterance, chain), whole batch only.
static int ptm_topn_range(psgpu_ptm_model_t *m, const float *feats_dev, const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                          const uint8_t *seed_in_dev, uint8_t *seed_out_dev, int32_t *topn_score_dev, uint8_t *topn_cw_dev,
                          PtmWorkspace *ws, int r, int32_t fr0, int32_t fr_n, bool lane_path, bool mark, hipStream_t st)
{
    // (mark: the model's timing event between the main pass and the fix-up is recorded in this range)
    static const int forced = [] { const char *e = getenv("PSGPU_CHUNK"); return e ? atoi(e) : 0; }();
    int chunk = forced;
    if (chunk <= 0) {
        // chunk length: enough wavefronts to fill 256 CUs x 32 waves a few times over, but long enough to amortise the parameter
        // load and the one-frame warm-up of every chunk
        const long long target_waves = 4LL * 256 * 32;
        long long c = ((long long)fr_n * m->n_chain + target_waves - 1) / target_waves;
        chunk = (int)(c < 32 ? 32 : (c > 512 ? 512 : c));
    }
    if (lane_path) {
        // chunk length of the CHUNKED fix-up form (only used when many entries are open)
        static const int fix_chunk = [] { const char *e = getenv("PSGPU_FIX_CHUNK"); return e ? atoi(e) : 32; }();
        chunk = fix_chunk > 0 ? fix_chunk : 32;
    }
    const long long n_chunks = ((long long)fr_n + chunk - 1) / chunk;
    const long long waves = n_chunks * m->n_chain;
    const int blocks = (int)((waves + 3) / 4);
    static const int occ = [] { const char *e = getenv("PSGPU_CHAIN_OCC"); return e ? atoi(e) : 8; }();
    const PtmDev pv = dev_view(m);
    uint32_t *cw32 = reinterpret_cast<uint32_t *>(topn_cw_dev);
#define PSGPU_CHAIN(GRID, FLAGS, CNT, LST, THR)                                                          \
    do {                                                                                                 \
        if (occ >= 8)                                                                                    \
            hipLaunchKernelGGL((ptm_chain_kernel<13, 4, 8>), dim3(GRID), dim3(256), 0, st, pv, feats_dev, \
                               utt_off_dev, n_utt, total_frames, chunk, seed_in_dev, seed_out_dev,       \
                               topn_score_dev, cw32, FLAGS, CNT, LST, THR, fr0, fr_n);                   \
        else                                                                                             \
            hipLaunchKernelGGL((ptm_chain_kernel<13, 4, 7>), dim3(GRID), dim3(256), 0, st, pv, feats_dev, \
                               utt_off_dev, n_utt, total_frames, chunk, seed_in_dev, seed_out_dev,       \
                               topn_score_dev, cw32, FLAGS, CNT, LST, THR, fr0, fr_n);                   \
    } while (0)
    if (lane_path) {
        // frames-on-lanes main pass, then the two fix-up forms (exactly one of them does work)
        const size_t need = (size_t)fr_n * m->n_chain;
        int32_t *fix_count = ws->fix_list + ws->flags_cap + r;       // the counters lie behind the list
        int32_t *fix_list = ws->fix_list + (size_t)fr0 * m->n_chain;
        const int32_t fix_thr = (int32_t)(need / 64);
        static const int fix_grid = [] { const char *e = getenv("PSGPU_FIX_GRID"); return e ? atoi(e) : 512; }();
        // frames per lane.  2 and 4 use the packed single-precision form of the distance (v_pk_add_f32 / v_pk_mul_f32), half
        // the VALU instructions per codeword-frame -- measured on the 512 x 30 s batch (r02, profiles/): 41.2 / 42.2 / 42.1 ms
        // of scorer stage for 1 / 2 / 4, i.e. the packed operations issue at half rate here and buy nothing; 1 stays the default
        static const int fpl = [] { const char *e = getenv("PSGPU_LANE_FPL"); return e ? atoi(e) : 1; }();
        const long long n_tiles = ((long long)fr_n + 64 * fpl - 1) / (64 * fpl);
        // workgroups: eight XCD shares of ceil(units / 8) four-tile units, each for every chain (see the kernel)
        const long long lane_units = (n_tiles + 3) / 4, lane_upx = (lane_units + 7) / 8;
        const long long lw = 8 * lane_upx * m->n_chain * 4;
        if (fpl == 4)
            hipLaunchKernelGGL((ptm_lane_kernel<13, 4>), dim3((unsigned)((lw + 3) / 4)), dim3(256), 0, st,
                               pv, feats_dev, total_frames, utt_off_dev, n_utt, seed_out_dev,
                               topn_score_dev, cw32, ws->open_flags, fix_count, fix_list, (int32_t)need, fr0, fr_n);
        else if (fpl == 2)
            hipLaunchKernelGGL((ptm_lane_kernel<13, 2>), dim3((unsigned)((lw + 3) / 4)), dim3(256), 0, st,
                               pv, feats_dev, total_frames, utt_off_dev, n_utt, seed_out_dev,
                               topn_score_dev, cw32, ws->open_flags, fix_count, fix_list, (int32_t)need, fr0, fr_n);
        else
            hipLaunchKernelGGL((ptm_lane_kernel<13, 1>), dim3((unsigned)((lw + 3) / 4)), dim3(256), 0, st,
                               pv, feats_dev, total_frames, utt_off_dev, n_utt, seed_out_dev,
                               topn_score_dev, cw32, ws->open_flags, fix_count, fix_list, (int32_t)need, fr0, fr_n);
        PSGPU_HIP(hipGetLastError());
        if (m->timing && mark) hipEventRecord(m->ev[1], st);
        PSGPU_CHAIN(fix_grid, (const uint8_t *)ws->open_flags, (const int32_t *)fix_count, (const int32_t *)fix_list, fix_thr);
    }
    else {
        if (m->timing && mark) hipEventRecord(m->ev[1], st);
        PSGPU_CHAIN(blocks, (const uint8_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0);
    }
#undef PSGPU_CHAIN
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

// the workspace's open-entry flags, list and counters for a batch of `total_frames` (the lane path); counters zeroed if a call before
// left them dirty (normally the senone kernel of the previous call zeroed them)
static int ptm_ws_prepare(psgpu_ptm_model_t *m, PtmWorkspace *ws, int32_t total_frames, hipStream_t st)
{
    const size_t need = (size_t)total_frames * m->n_chain;
    if (need > ws->flags_cap) {                                      // (two host threads scoring on one model use two streams, hence two workspaces)
        PSGPU_HIP(hipStreamSynchronize(st));
        if (ws->aux) PSGPU_HIP(hipStreamSynchronize(ws->aux));
        hipFree(ws->open_flags); hipFree(ws->fix_list);
        ws->open_flags = nullptr; ws->fix_list = nullptr; ws->flags_cap = 0;
        PSGPU_HIP(hipMalloc((void **)&ws->open_flags, need));
        PSGPU_HIP(hipMalloc((void **)&ws->fix_list, (need + kPtmMaxRanges) * sizeof(int32_t)));
        ws->flags_cap = need;
        ws->count_dirty = 1;
    }
    if (ws->count_dirty) PSGPU_HIP(hipMemsetAsync(ws->fix_list + ws->flags_cap, 0, sizeof(int32_t) * kPtmMaxRanges, st));
    ws->count_dirty = 1;
    return PSGPU_OK;
}

static bool ptm_lane_path(const psgpu_ptm_model_t *m)
{
    static const int no_lane = [] { const char *e = getenv("PSGPU_NO_LANE_KERNEL"); return e ? atoi(e) : 0; }();
    return m->fast_shape && m->ds_ratio == 1 && !no_lane;
}

static int ptm_topn_generic(psgpu_ptm_model_t *m, const float *feats_dev, const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                            const uint8_t *seed_in_dev, uint8_t *seed_out_dev, int32_t *topn_score_dev, uint8_t *topn_cw_dev, hipStream_t st)
{
    // every other shape psgpu_ptm_frame_eval serves (topn 1..8 -- a user's knob, config_macro.h:384 --, up to 256 densities,
    // any stream lengths, any -ds): the exact sequential procedure, one wavefront per (utterance, chain)
    const PtmDev pv = dev_view(m);
    const long long waves = (long long)n_utt * m->n_chain;
    const unsigned grid = (unsigned)std::min<long long>((waves + 3) / 4, 256LL * 8);
    if (m->timing) { hipEventRecord(m->ev[0], st); }
#define PSGPU_GEN(NN) case NN: hipLaunchKernelGGL((ptm_batch_topn_generic<NN>), dim3(grid), dim3(256), 0, st, pv, feats_dev, utt_off_dev, n_utt, \
                                                  total_frames, seed_in_dev, seed_out_dev, topn_score_dev, topn_cw_dev); break;
    switch (m->topn) { PSGPU_GEN(1) PSGPU_GEN(2) PSGPU_GEN(3) PSGPU_GEN(4) PSGPU_GEN(5) PSGPU_GEN(6) PSGPU_GEN(7) default: PSGPU_GEN(8) }
#undef PSGPU_GEN
    if (m->timing) { hipEventRecord(m->ev[1], st); hipEventRecord(m->ev[2], st); }
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_ptm_topn_dev(psgpu_ptm_model_t *m, const float *feats_dev,
                       const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                       const uint8_t *seed_in_dev, uint8_t *seed_out_dev,
                       int32_t *topn_score_dev, uint8_t *topn_cw_dev, void *stream)
{
    PSGPU_REQUIRE(m && feats_dev && utt_off_dev && topn_score_dev && topn_cw_dev,
                  "psgpu_ptm_topn_dev: NULL argument");
    PSGPU_REQUIRE(n_utt >= 0 && total_frames >= 0, "negative sizes");
    PSGPU_REQUIRE(seed_in_dev == nullptr || seed_in_dev != seed_out_dev,
                  "seed_in and seed_out must not alias (chunks read seeds while others write carry-outs)");
    if (n_utt == 0 || total_frames == 0) return PSGPU_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!m->fast_shape)
        return ptm_topn_generic(m, feats_dev, utt_off_dev, n_utt, total_frames, seed_in_dev, seed_out_dev, topn_score_dev, topn_cw_dev, st);
    const bool lane_path = ptm_lane_path(m);
    PtmWorkspace *ws = nullptr;
    if (lane_path) {
        ws = ptm_workspace(m, st, true);
        int rc = ptm_ws_prepare(m, ws, total_frames, st);
        if (rc) return rc;
    }
    if (m->timing) hipEventRecord(m->ev[0], st);
    int rc = ptm_topn_range(m, feats_dev, utt_off_dev, n_utt, total_frames, seed_in_dev, seed_out_dev, topn_score_dev, topn_cw_dev, ws, 0, 0,
                            total_frames, lane_path, true, st);
    if (m->timing) hipEventRecord(m->ev[2], st);
    return rc;
}

// the senone pass of the frames [fr0, fr0 + fr_n); zw: the open-entry counter this launch zeroes for the next call (or NULL)
static int ptm_senone_range(psgpu_ptm_model_t *m, int32_t total_frames, const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                            int16_t *senscr_dev, int32_t *best_dev, uint32_t flags, int32_t *zw, int32_t fr0, int32_t fr_n, hipStream_t st, bool *used_fast)
{
    static const int force_generic = [] { const char *e = getenv("PSGPU_SENONE_GENERIC"); return e ? atoi(e) : 0; }();
    *used_fast = false;
    if (!force_generic && m->fast_shape && m->n_feat == 3 && m->topn == 4 && m->n_chain <= 256 && m->n_sen < 0xffff) {
        // block = 4..8 waves: pick the width that wastes the fewest lanes
        int best_w = 0, iters = 0;
        double best_eff = 0;
        static const int force_w = [] { const char *e = getenv("PSGPU_SEN_W"); return e ? atoi(e) : 0; }();
        static const int sen_fr = [] { const char *e = getenv("PSGPU_SEN_FR"); return e ? atoi(e) : 1; }();
        // measured on MI355X (tools/sensweep.sh): the kernel is latency-bound and runs best with the
        // most workgroups in flight per CU, i.e. the narrowest workgroup that covers a frame's
        // groups in <= kSenMaxIters passes (en-us: 4 waves x 6 passes, 8 workgroups per CU)
        for (int w = (force_w ? force_w : 4); w <= (force_w ? force_w : 8) && !best_w; ++w) {
            const int it = (m->n_groups + 64 * w - 1) / (64 * w);
            const double eff = (double)m->n_groups / ((double)it * 64 * w);
            if (it <= kSenMaxIters && eff > best_eff + 1e-9) { best_eff = eff; best_w = w; iters = it; }
        }
        if (best_w) {
            const int kSenFr = (sen_fr == 1 || sen_fr == 4) ? sen_fr : 2;
            const dim3 grid((((fr_n + kSenFr - 1) / kSenFr + 7) / 8) * 8), block(64 * best_w);      // (a multiple of eight: see the kernel)
            // (PSGPU_SENONE_DIRECT=1: measured on the 512 x 30 s batch, two batches in flight, profiles/round6_scorer_ab.txt -- the scorer
            //  stage beside the other batch's search 63.4 -> 57.5 ms, alone 37.8 -> 39.8 (narrower stores), but the search beside it
            //  71.3 -> 73.0 ms and a step 77.7 -> 79.5: the step IS the search's time beside the stages, and a senone kernel that gets
            //  three times the wavefronts onto a compute unit disturbs the search's trips to device memory more.  The staged form
            //  stays the default; the direct form is for a scorer that runs alone beside LDS-hungry work.)
            static const int want_direct = [] { const char *e = getenv("PSGPU_SENONE_DIRECT"); return e ? atoi(e) : 0; }();
            const bool direct = want_direct && (flags & PSGPU_PTM_RAW_SCORES) && (m->n_sen & 1) == 0 && ((uintptr_t)senscr_dev & 3) == 0 && kSenFr == 1;
            const size_t sm = direct ? 0 : (((size_t)m->n_sen * 2 + 15) / 16) * 16 * kSenFr;
            const PtmDev pv = dev_view(m);
            const uint32_t *cw32 = reinterpret_cast<const uint32_t *>(topn_cw_dev);
#define PSGPU_SEN_CASE_(I, SAD) \
                if (direct) hipLaunchKernelGGL((ptm_senone_kernel_f3n4<I, 1, SAD, true>), grid, block, sm, st, pv,   \
                                       topn_score_dev, cw32, senscr_dev, best_dev, flags, total_frames, zw, fr0, fr_n); \
                else if (kSenFr == 1) hipLaunchKernelGGL((ptm_senone_kernel_f3n4<I, 1, SAD>), grid, block, sm, st, pv,   \
                                       topn_score_dev, cw32, senscr_dev, best_dev, flags, total_frames, zw, fr0, fr_n); \
                else if (kSenFr == 4) hipLaunchKernelGGL((ptm_senone_kernel_f3n4<I, 4, SAD>), grid, block, sm, st, pv, \
                                       topn_score_dev, cw32, senscr_dev, best_dev, flags, total_frames, zw, fr0, fr_n); \
                else hipLaunchKernelGGL((ptm_senone_kernel_f3n4<I, 2, SAD>), grid, block, sm, st, pv,               \
                                       topn_score_dev, cw32, senscr_dev, best_dev, flags, total_frames, zw, fr0, fr_n);
#define PSGPU_SEN_CASE(I) case I: if (sad) { PSGPU_SEN_CASE_(I, true) } else { PSGPU_SEN_CASE_(I, false) } break;
            // the biased form of the log-add chain (v_sad_u32): when 3 x the table's largest entry <= the bias
            // (PSGPU_SENONE_SAD=0: the signed form, for A/B runs and the tests of that path)
            static const int use_sad = [] { const char *e = getenv("PSGPU_SENONE_SAD"); return e ? atoi(e) : 1; }();
            const bool sad = use_sad && 3 * m->la_max <= kSadBias;
            switch (iters) {
                PSGPU_SEN_CASE(1) PSGPU_SEN_CASE(2) PSGPU_SEN_CASE(3) PSGPU_SEN_CASE(4)
                PSGPU_SEN_CASE(5) PSGPU_SEN_CASE(6) PSGPU_SEN_CASE(7) default: PSGPU_SEN_CASE(8)
            }
#undef PSGPU_SEN_CASE
#undef PSGPU_SEN_CASE_
            PSGPU_HIP(hipGetLastError());
            *used_fast = true;
            return PSGPU_OK;
        }
    }
    PSGPU_REQUIRE(fr0 == 0 && fr_n == total_frames, "psgpu_ptm_senone_dev: the any-shape senone kernel takes whole batches");
    // any shape (and the fast shape on request): one workgroup per frame, lists [chain][frame][topn]
    const int out_bytes = ((m->n_sen * 2 + 15) / 16) * 16;
    const int list_bytes = ((m->n_chain * m->topn + 15) / 16) * 16;
    const size_t smem = (size_t)out_bytes + 2 * list_bytes + 256 + 16 * 4 + 8 * 4;
    PSGPU_REQUIRE(smem <= 160 * 1024, "psgpu_ptm_senone_dev: %zu bytes of LDS for a frame's score row and lists exceed a compute unit's", smem);
#define PSGPU_SEN_GEN(NN) case NN: {                                                                                                        \
        if (smem > 64 * 1024)                                                                                                              \
            PSGPU_HIP(hipFuncSetAttribute((const void *)ptm_senone_kernel<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));    \
        hipLaunchKernelGGL((ptm_senone_kernel<NN>), dim3(total_frames), dim3(kSenThreads), smem, st, dev_view(m),                           \
                           topn_score_dev, topn_cw_dev, senscr_dev, best_dev, flags, total_frames); } break;
    switch (m->topn) { PSGPU_SEN_GEN(1) PSGPU_SEN_GEN(2) PSGPU_SEN_GEN(3) PSGPU_SEN_GEN(4) PSGPU_SEN_GEN(5) PSGPU_SEN_GEN(6) PSGPU_SEN_GEN(7)
                       default: PSGPU_SEN_GEN(8) }
#undef PSGPU_SEN_GEN
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_ptm_senone_dev(psgpu_ptm_model_t *m, int32_t total_frames,
                         const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                         int16_t *senscr_dev, int32_t *best_dev, uint32_t flags,
                         void *stream)
{
    PSGPU_REQUIRE(m && topn_score_dev && topn_cw_dev && senscr_dev,
                  "psgpu_ptm_senone_dev: NULL argument");
    if (total_frames <= 0) return PSGPU_OK;
    hipStream_t st = (hipStream_t)stream;
    PtmWorkspace *ws = ptm_workspace(m, st, false);
    int32_t *zw = (ws && ws->fix_list) ? ws->fix_list + ws->flags_cap : nullptr;       // (the whole-batch top-N pass used counter 0)
    bool fast = false;
    const int rc = ptm_senone_range(m, total_frames, topn_score_dev, topn_cw_dev, senscr_dev, best_dev, flags, zw, 0, total_frames, st, &fast);
    if (rc == PSGPU_OK && fast && zw) ws->count_dirty = 0;
    return rc;
}

int psgpu_ptm_kernel_timing(psgpu_ptm_model_t *m, int32_t enable)
{
    PSGPU_REQUIRE(m != nullptr, "psgpu_ptm_kernel_timing: NULL model");
    if (enable && !m->ev[0])
        for (int i = 0; i < 4; ++i) PSGPU_HIP(hipEventCreate(&m->ev[i]));
    m->timing = enable ? 1 : 0;
    return PSGPU_OK;
}

int psgpu_ptm_last_kernel_ms(psgpu_ptm_model_t *m, float *ms3)
{
    PSGPU_REQUIRE(m && ms3 && m->timing, "psgpu_ptm_last_kernel_ms: timing is not enabled");
    // (a caller of psgpu_ptm_topn_dev alone -- the decode pipeline scoring from lists -- records no senone kernel: 0 then)
    PSGPU_HIP(hipEventSynchronize(m->ev[2]));
    PSGPU_HIP(hipEventElapsedTime(&ms3[0], m->ev[0], m->ev[1]));    // main top-N kernel
    PSGPU_HIP(hipEventElapsedTime(&ms3[1], m->ev[1], m->ev[2]));    // exact fix-up launch
    ms3[2] = 0.0f;
    if (m->sen_timed) {
        PSGPU_HIP(hipEventSynchronize(m->ev[3]));
        PSGPU_HIP(hipEventElapsedTime(&ms3[2], m->ev[2], m->ev[3]));    // senone kernel
    }
    return PSGPU_OK;
}

// The whole scorer for a batch.  Large batches of the specialised shape go in RANGES of frames: range r's top-N pass on the caller's
// stream while the senone pass of range r - 1 runs on a stream of the workspace's own -- the first is bound by the vector ALU and uses
// no LDS, the second by the latency of its table look-ups and (beside a resident search kernel) by the LDS its staged row takes:
// side by side they fill each other's gaps (bench.py stage_ms.scorer).  The caller's stream waits for the last senone
// pass before the call returns control of it; results and layouts are those of one pass over the whole batch.
int psgpu_ptm_score_batch_dev(psgpu_ptm_model_t *m,
                              const float *feats_dev, const int32_t *utt_off_dev,
                              int32_t n_utt, int32_t total_frames,
                              const uint8_t *seed_in_dev, uint8_t *seed_out_dev,
                              int32_t *topn_score_dev, uint8_t *topn_cw_dev,
                              int16_t *senscr_dev, int32_t *best_dev,
                              uint32_t flags, void *stream)
{
    static const int n_ranges_env = [] { const char *e = getenv("PSGPU_PTM_RANGES"); return e ? atoi(e) : 0; }();
    hipStream_t st = (hipStream_t)stream;
    int R = 1;
    if (m && senscr_dev && n_utt > 0 && ptm_lane_path(m) && m->n_feat == 3 && m->n_chain <= 256 && m->n_sen < 0xffff) {
        // (measured on the 512 x 30 s batch, profiles/round6_scorer_ranges.txt: 1 / 4 / 6 / 8 ranges give 63.8 / 65.4 / 65.7 / 65.3 ms of
        //  scorer stage beside the other batch's search and 38.0 / 37.8 / 38.2 / 38.0 ms alone -- the two passes do not fill each
        //  other's gaps, both want the vector ALU; one range stays the default, PSGPU_PTM_RANGES is the A/B knob)
        R = n_ranges_env > 0 ? n_ranges_env : 1;
        R = std::min(R, kPtmMaxRanges);
        while (R > 1 && total_frames / R < 4096) --R;
    }
    if (R <= 1) {
        int rc = psgpu_ptm_topn_dev(m, feats_dev, utt_off_dev, n_utt, total_frames, seed_in_dev,
                                    seed_out_dev, topn_score_dev, topn_cw_dev, stream);
        if (rc != PSGPU_OK || senscr_dev == nullptr) return rc;
        rc = psgpu_ptm_senone_dev(m, total_frames, topn_score_dev, topn_cw_dev, senscr_dev,
                                  best_dev, flags, stream);
        if (rc == PSGPU_OK && m->timing) { hipEventRecord(m->ev[3], st); m->sen_timed = true; }
        return rc;
    }
    PSGPU_REQUIRE(feats_dev && utt_off_dev && topn_score_dev && topn_cw_dev, "psgpu_ptm_score_batch_dev: NULL argument");
    PSGPU_REQUIRE(seed_in_dev == nullptr || seed_in_dev != seed_out_dev, "seed_in and seed_out must not alias");
    PtmWorkspace *ws = ptm_workspace(m, st, true);
    int rc = ptm_ws_prepare(m, ws, total_frames, st);
    if (rc) return rc;
    if (!ws->aux) {
        PSGPU_HIP(hipStreamCreateWithFlags(&ws->aux, hipStreamNonBlocking));
        for (int i = 0; i <= kPtmMaxRanges; ++i) PSGPU_HIP(hipEventCreateWithFlags(&ws->ev[i], hipEventDisableTiming));
    }
    // ranges of whole 256-frame units (a top-N workgroup's four tiles)
    const int32_t per = (int32_t)((((int64_t)total_frames + R - 1) / R + 255) / 256 * 256);
    if (m->timing) hipEventRecord(m->ev[0], st);
    int r = 0;
    bool all_fast = true;
    for (int32_t fr0 = 0; fr0 < total_frames; fr0 += per, ++r) {
        const int32_t fr_n = std::min(per, total_frames - fr0);
        if ((rc = ptm_topn_range(m, feats_dev, utt_off_dev, n_utt, total_frames, seed_in_dev, seed_out_dev, topn_score_dev, topn_cw_dev, ws, r,
                                 fr0, fr_n, true, fr0 + per >= total_frames, st)))
            return rc;
        PSGPU_HIP(hipEventRecord(ws->ev[r], st));
        PSGPU_HIP(hipStreamWaitEvent(ws->aux, ws->ev[r], 0));
        bool fast = false;
        if ((rc = ptm_senone_range(m, total_frames, topn_score_dev, topn_cw_dev, senscr_dev, best_dev, flags, ws->fix_list + ws->flags_cap + r, fr0, fr_n,
                                   ws->aux, &fast)))
            return rc;
        all_fast = all_fast && fast;
    }
    if (m->timing) hipEventRecord(m->ev[2], st);         // (the top-N passes' end; the senone passes go on beside)
    PSGPU_HIP(hipEventRecord(ws->ev[kPtmMaxRanges], ws->aux));
    PSGPU_HIP(hipStreamWaitEvent(st, ws->ev[kPtmMaxRanges], 0));
    if (all_fast) ws->count_dirty = 0;
    if (m->timing) { hipEventRecord(m->ev[3], st); m->sen_timed = true; }
    return PSGPU_OK;
}

int psgpu_ptm_score_batch(psgpu_ptm_model_t *m,
                          const float *feats, const int32_t *utt_off, int32_t n_utt,
                          uint8_t *seed_cw,
                          int32_t *topn_score, uint8_t *topn_cw,
                          int16_t *senscr, int32_t *best, uint32_t flags)
{
    PSGPU_REQUIRE(m && feats && utt_off && n_utt >= 0, "psgpu_ptm_score_batch: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    const int32_t T = utt_off[n_utt];
    PSGPU_REQUIRE(T >= 0 && utt_off[0] == 0, "utt_off must start at 0 and be non-decreasing");
    for (int u = 0; u < n_utt; ++u)
        PSGPU_REQUIRE(utt_off[u + 1] >= utt_off[u], "utt_off must be non-decreasing");
    if (T == 0) return PSGPU_OK;
    const size_t n_ent = (size_t)T * m->n_chain * m->topn;
    float *d_feat = nullptr; int32_t *d_off = nullptr, *d_sc = nullptr, *d_best = nullptr;
    uint8_t *d_seed = nullptr, *d_seed_out = nullptr, *d_cw = nullptr; int16_t *d_scr = nullptr;
    int rc = PSGPU_OK;
    auto cleanup = [&]() {
        hipFree(d_feat); hipFree(d_off); hipFree(d_sc); hipFree(d_best);
        hipFree(d_seed); hipFree(d_seed_out); hipFree(d_cw); hipFree(d_scr);
    };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) {                 \
        psgpu_set_error("%s -> %s", #call, hipGetErrorString(e_)); cleanup();          \
        return e_ == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP; } } while (0)
    TRY(hipMalloc((void **)&d_feat, (size_t)T * m->veclen * sizeof(float)));
    TRY(hipMalloc((void **)&d_off, (size_t)(n_utt + 1) * sizeof(int32_t)));
    TRY(hipMalloc((void **)&d_sc, n_ent * sizeof(int32_t)));
    TRY(hipMalloc((void **)&d_cw, n_ent));
    TRY(hipMemcpy(d_feat, feats, (size_t)T * m->veclen * sizeof(float), hipMemcpyHostToDevice));
    TRY(hipMemcpy(d_off, utt_off, (size_t)(n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (seed_cw) {
        TRY(hipMalloc((void **)&d_seed, (size_t)n_utt * m->n_chain * m->topn));
        TRY(hipMemcpy(d_seed, seed_cw, (size_t)n_utt * m->n_chain * m->topn, hipMemcpyHostToDevice));
        TRY(hipMalloc((void **)&d_seed_out, (size_t)n_utt * m->n_chain * m->topn));
        // empty utterances pass their seed through unchanged
        TRY(hipMemcpy(d_seed_out, seed_cw, (size_t)n_utt * m->n_chain * m->topn, hipMemcpyHostToDevice));
    }
    if (senscr) TRY(hipMalloc((void **)&d_scr, (size_t)T * m->n_sen * sizeof(int16_t)));
    if (best) TRY(hipMalloc((void **)&d_best, (size_t)T * sizeof(int32_t)));
    rc = psgpu_ptm_score_batch_dev(m, d_feat, d_off, n_utt, T, d_seed, d_seed_out, d_sc, d_cw,
                                   d_scr, d_best, flags, nullptr);
    if (rc == PSGPU_OK) {
        TRY(hipDeviceSynchronize());
        // device lists are chain-major ([chain][frame][topn]); the host interface keeps the
        // reference's frame-major view (s->f->topn[cb][feat][k] per frame)
        if (topn_score) {
            std::vector<int32_t> tmp(n_ent);
            TRY(hipMemcpy(tmp.data(), d_sc, n_ent * sizeof(int32_t), hipMemcpyDeviceToHost));
            const size_t N_ = (size_t)m->topn, C_ = (size_t)m->n_chain;
            for (size_t c = 0; c < C_; ++c)
                for (size_t t = 0; t < (size_t)T; ++t)
                    memcpy(topn_score + (t * C_ + c) * N_, tmp.data() + (c * (size_t)T + t) * N_, N_ * sizeof(int32_t));
        }
        if (topn_cw) {
            std::vector<uint8_t> tmp(n_ent);
            TRY(hipMemcpy(tmp.data(), d_cw, n_ent, hipMemcpyDeviceToHost));
            const size_t N_ = (size_t)m->topn, C_ = (size_t)m->n_chain;
            for (size_t c = 0; c < C_; ++c)
                for (size_t t = 0; t < (size_t)T; ++t)
                    memcpy(topn_cw + (t * C_ + c) * N_, tmp.data() + (c * (size_t)T + t) * N_, N_);
        }
        if (senscr) TRY(hipMemcpy(senscr, d_scr, (size_t)T * m->n_sen * sizeof(int16_t), hipMemcpyDeviceToHost));
        if (best) TRY(hipMemcpy(best, d_best, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (seed_cw) TRY(hipMemcpy(seed_cw, d_seed_out, (size_t)n_utt * m->n_chain * m->topn, hipMemcpyDeviceToHost));
    }
#undef TRY
    cleanup();
    return rc;
}

}  // extern "C"

// psgpu_hmm.hip -- the per-frame Viterbi step of hmm_vit_eval() (reference
// src/hmm.c:786-805) for batches of active HMMs on gfx950.
//
// Replaces hmm_vit_eval_3st_lr (:529-607), _3st_lr_mpx (:609-707), _5st_lr
// (:222-350) and _5st_lr_mpx (:355-525) bit-exactly (int32 path scores, the
// reference's arg-max tie rules, WORST_SCORE clamps, stale-t2 behaviour of the
// 3-state form, BAD_SSID handling and ssid propagation of the multiplex form).
//
// Layout: one 64-byte record per HMM (psgpu_hmm_rec_t: the fields of hmm_t,
// hmm.h:169-182, minus the context pointer and frame stamp), so that a lane
// working on an arbitrary member of the active list moves exactly one aligned
// 64-byte line in and one out -- the access pattern is a gather/scatter over
// the active list and nothing else is worth staging.  Transition matrices
// (uint8 tp[tmat][from][to], tmat.h:60-66) are tiny and sit in LDS; senone
// scores are gathered from the frame's int16 row of the HMM's utterance; the
// multiplex form goes through sseq[ssid][state] (bin_mdef.h:119) first.
// One lane = one HMM; the block max of the returned best scores is folded into
// best[utt] with one atomic per wavefront (evaluate_channels keeps the same
// maximum, ngram_search_fwdtree.c:701-715).
#include "psgpu_internal.h"
#include <cstring>
#include <cstdlib>
#include <climits>

static_assert(sizeof(psgpu_hmm_rec_t) == 64, "HMM record must be one 64-byte line");

struct psgpu_hmm_ctx_s {
    int32_t n_emit, n_tmat, n_sseq, n_sen;
    uint8_t *tp;             // device [n_tmat][n_emit][n_emit+1]
    uint16_t *sseq;          // device [n_sseq][n_emit]
    // staging for the host-buffer entry point
    hipStream_t stream;
    psgpu_hmm_rec_t *h_recs, *d_recs;
    int16_t *h_scr, *d_scr;
    int32_t *h_best, *d_best;
    int32_t cap;
    // zero-copy path of the host-buffer entry (small batches): host-mapped buffers + completion word
    psgpu_hmm_rec_t *z_recs, *zd_recs;     // [kZeroCopyMax]
    int16_t *z_scr, *zd_scr;
    uint32_t *z_word, *zd_word;            // [0] completion word, [1] best score
    uint32_t *d_count;
    uint32_t seq;
    // per-launch completion arguments (nullptr for the plain device entry)
    uint32_t *launch_done_count, *launch_done_word;
    uint32_t launch_seq;
    // phone-loop scratch
    int16_t *pl_css; int32_t pl_cap;        // [frames][64][4 or 8] normalised CI state scores
};

constexpr int32_t kZeroCopyMax = 2048;

#include "psgpu_hmm_dev.h"
#include "psgpu_sen_dev.h"

constexpr int kHmmThreads = 256;
constexpr int kTpLdsMax = 4096;      // (transition matrices of up to this many bytes sit in LDS: en-us 504, a 5-state model of 136 matrices)

// fold a wave's running maximum into best[utt] (one atomic per wave per flush)
__device__ __forceinline__ void flush_best(int32_t *best_out, int utt, int32_t m)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m != kMaxNegInt32) atomicMax(&best_out[utt], m);
}

typedef int hmm_v4i __attribute__((ext_vector_type(4)));
template <int NE, bool STREAM>      // STREAM: a dense list beyond the caches' reach, read and written around them
__global__ __launch_bounds__(kHmmThreads)
void hmm_vit_kernel(psgpu_hmm_rec_t *__restrict__ recs, const int32_t *__restrict__ active,
                    int32_t n_active, const uint16_t *__restrict__ utt_of_hmm,
                    const int16_t *__restrict__ senscr, int32_t senscr_stride,
                    const uint8_t *__restrict__ tp_g, int32_t tp_bytes,
                    const uint16_t *__restrict__ sseq, int32_t *__restrict__ best_out,
                    uint32_t *__restrict__ done_count, uint32_t *__restrict__ done_word, uint32_t seq)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tp[kTpLdsMax];
    __shared__ int32_t s_wbest[kHmmThreads / 64], s_wutt[kHmmThreads / 64];
    // dense lists (active == NULL: records base .. base + 63 of a wavefront lie side by side): the records travel through LDS, 1 KB
    // contiguous per wave-instruction (sixteen full lines) instead of sixteen bytes of each of 64 lines four times over -- the
    // kernel's time was the address processing of those accesses (bench.py extra.hmm_vit_kernel).  Quad q of record r sits at
    // quad slot 4 r + ((q + (r >> 1)) & 3): a record's work-item then reads its four quads without bank conflicts beyond the
    // eight-lane groups a 128-bit LDS read is served in.
    __shared__ int4 s_stage[kHmmThreads / 64][256];
    const bool tp_in_lds = tp_bytes <= kTpLdsMax;
    if (tp_in_lds) {
        for (int i = threadIdx.x * 4; i < tp_bytes; i += kHmmThreads * 4) {
            if (i + 4 <= tp_bytes) *reinterpret_cast<uint32_t *>(s_tp + i) = *reinterpret_cast<const uint32_t *>(tp_g + i);
            else for (int j = i; j < tp_bytes; ++j) s_tp[j] = tp_g[j];
        }
        __syncthreads();
    }
    // persistent blocks: the grid is sized to the machine, every lane walks
    // the active list with a grid stride.  The running best of a wave is kept
    // in a register together with the utterance it belongs to and flushed
    // when the utterance changes (active lists are grouped by utterance).
    int32_t acc = kMaxNegInt32;
    int acc_utt = -1;
    const int stride = gridDim.x * kHmmThreads;
    constexpr bool stream_list = STREAM;
    for (int base = blockIdx.x * kHmmThreads; base < n_active; base += stride) {
        const int i = base + threadIdx.x;
        const bool live = i < n_active;
        int32_t best = kMaxNegInt32;
        int utt = 0;
        const int lane = threadIdx.x & 63;
        const bool dense = !active && (base + (int)(threadIdx.x & ~63u) + 64 <= n_active);      // (wave-uniform: a full wavefront of consecutive records)
        int4 *const stg = s_stage[threadIdx.x >> 6];
        if (dense) {
            const int4 *gp = reinterpret_cast<const int4 *>(recs + (i - lane));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pq = k * 64 + lane, r = pq >> 2, q = pq & 3;
                // (a list beyond the caches' reach -- 2^20 records, 64 MB -- is streamed: neither its reads nor its writes are worth a line there)
                if (stream_list) *reinterpret_cast<hmm_v4i *>(&stg[4 * r + ((q + (r >> 1)) & 3)]) = __builtin_nontemporal_load(reinterpret_cast<const hmm_v4i *>(&gp[pq]));
                else stg[4 * r + ((q + (r >> 1)) & 3)] = gp[pq];
            }
        }
        if (live) {
            const int idx = active ? active[i] : i;
            // one 64-byte line in
            const int4 *rp = reinterpret_cast<const int4 *>(recs + idx);
            int4 q0, q1, q2, q3;
            if (dense) {
                const int sw = lane >> 1;
                q0 = stg[4 * lane + (sw & 3)]; q1 = stg[4 * lane + ((1 + sw) & 3)]; q2 = stg[4 * lane + ((2 + sw) & 3)]; q3 = stg[4 * lane + ((3 + sw) & 3)];
            }
            else { q0 = rp[0]; q1 = rp[1]; q2 = rp[2]; q3 = rp[3]; }
            utt = utt_of_hmm ? utt_of_hmm[idx] : 0;
            HmmRegs h;
            h.score[0] = q0.x; h.score[1] = q0.y; h.score[2] = q0.z; h.score[3] = q0.w;
            h.score[4] = q1.x; h.history[0] = q1.y; h.history[1] = q1.z; h.history[2] = q1.w;
            h.history[3] = q2.x; h.history[4] = q2.y; h.out_score = q2.z; h.out_history = q2.w;
            h.bestscore = q3.x;
            h.senid[0] = (uint16_t)(q3.y & 0xffff); h.senid[1] = (uint16_t)((uint32_t)q3.y >> 16);
            h.senid[2] = (uint16_t)(q3.z & 0xffff); h.senid[3] = (uint16_t)((uint32_t)q3.z >> 16);
            h.senid[4] = (uint16_t)(q3.w & 0xffff);
            const uint32_t tm = (uint32_t)q3.w >> 16;
            const bool mpx = (tm & PSGPU_HMM_MPX) != 0;
            const uint32_t tmatid = tm & 0x7fffu;
            const uint8_t *tp = (tp_in_lds ? s_tp : tp_g) + (size_t)tmatid * NE * (NE + 1);
            const int16_t *ss = senscr + (size_t)utt * senscr_stride;
            if (NE == 3)
                best = mpx ? vit3_mpx(h, tp, ss, sseq) : vit3(h, tp, ss);
            else if (NE == 5)
                best = mpx ? vit5_mpx(h, tp, ss, sseq) : vit5(h, tp, ss);
            else
                best = vit_any<NE>(h, tp, ss, sseq, mpx);        // 1, 2, 4 states: hmm_vit_eval_anytopo
            // one 64-byte line out
            q0 = make_int4(h.score[0], h.score[1], h.score[2], h.score[3]);
            q1 = make_int4(h.score[4], h.history[0], h.history[1], h.history[2]);
            q2 = make_int4(h.history[3], h.history[4], h.out_score, h.out_history);
            q3 = make_int4(h.bestscore, (int32_t)((uint32_t)h.senid[0] | ((uint32_t)h.senid[1] << 16)),
                           (int32_t)((uint32_t)h.senid[2] | ((uint32_t)h.senid[3] << 16)),
                           (int32_t)((uint32_t)h.senid[4] | (tm << 16)));
            if (dense) {
                const int sw = lane >> 1;
                stg[4 * lane + (sw & 3)] = q0; stg[4 * lane + ((1 + sw) & 3)] = q1; stg[4 * lane + ((2 + sw) & 3)] = q2; stg[4 * lane + ((3 + sw) & 3)] = q3;
            }
            else {
                int4 *wp = reinterpret_cast<int4 *>(recs + idx);
                wp[0] = q0; wp[1] = q1; wp[2] = q2; wp[3] = q3;
            }
        }
        if (dense) {
            int4 *gp = reinterpret_cast<int4 *>(recs + (i - lane));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int slot = k * 64 + lane, r = slot >> 2, q = ((slot & 3) - (r >> 1)) & 3;
                if (stream_list) __builtin_nontemporal_store(*reinterpret_cast<const hmm_v4i *>(&stg[slot]), reinterpret_cast<hmm_v4i *>(&gp[4 * r + q]));
                else gp[4 * r + q] = stg[slot];
            }
        }
        if (best_out) {
            const unsigned long long lv = __ballot(live);
            if (lv) {
                const int first = __ffsll((long long)lv) - 1;
                const int utt0 = __builtin_amdgcn_readlane(utt, first);
                if (__ballot(live && utt != utt0) == 0) {       // wave-uniform utterance (the usual case)
                    if (utt0 != acc_utt) {
                        if (acc_utt >= 0) flush_best(best_out, acc_utt, acc);
                        acc = kMaxNegInt32;
                        acc_utt = utt0;
                    }
                    acc = max(acc, best);
                }
                else if (live)
                    atomicMax(&best_out[utt], best);
            }
        }
    }
    if (best_out) {
        // combine the block's waves when they agree on the utterance
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc = max(acc, __shfl_xor(acc, off));
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_wbest[w] = acc; s_wutt[w] = acc_utt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int a = 0; a < kHmmThreads / 64; ++a) {
                if (s_wutt[a] < 0) continue;
                int32_t m = s_wbest[a];
                for (int b2 = a + 1; b2 < kHmmThreads / 64; ++b2)
                    if (s_wutt[b2] == s_wutt[a]) { m = max(m, s_wbest[b2]); s_wutt[b2] = -1; }
                if (m != kMaxNegInt32) atomicMax(&best_out[s_wutt[a]], m);
            }
        }
    }
    if (done_word) {
        // zero-copy host entry: records / best live in host-mapped memory; the last
        // workgroup to finish publishes the call's sequence number for the polling host
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t k = atomicAdd(done_count, 1u);
            if (k == gridDim.x - 1) {
                *done_count = 0;
                __threadfence_system();
                __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}


// ---------------------------------------------------------------------------
// The phone-loop search of a whole utterance in one launch:
// phone_loop_search_start (phone_loop_search.c:165-184) followed by
// phone_loop_search_step (:302-340) for every frame -- evaluate_hmms,
// store_scores, prune_hmms, phone_transition (:201-300) and the renormalisation
// test (:320-325) -- on un-normalised senone score rows that are already on the
// device.  One wavefront per utterance, lane = CI phone (n_phones <= 64): the
// search is a recurrence over frames, its per-frame work is 42 HMMs.  Histories
// are not tracked: nothing reads them (the search produces no back-pointers,
// only pls->penalties).  Per frame the kernel leaves the penalties, the ring
// value behind them and the HMM state, so that a host decoder can take any
// frame's vector as pls->penalties and resume stepping on the host from any
// frame.
// ---------------------------------------------------------------------------
struct PlDev {
    int32_t n_phones, window, beam, pbeam, pip, n_list, norm_mode;
    double weight;
    const uint16_t *ssid;      // [n_phones]
    const int16_t *tmatid;     // [n_phones]
    const uint16_t *ci_list;   // [n_list] senones the all-phones-active list holds (acmod_flags2list incl. bridges)
};

constexpr int kPlMaxWindow = 32;
constexpr int kPlCarryMagic = 0x5ea4c4ed;
constexpr int kPlCarryWords = 64 * 8 + kPlMaxWindow * 64 + 8;     // psgpu_phone_loop_run_carry_dev: an utterance's state between calls

// DPP wave maximum (same sequence as psgpu_ptm_dev.h): ~20 cycles instead of six
// dependent ds_bpermute round trips -- the search below is one wave marching through
// the frames, its step latency is the whole cost.
__device__ __forceinline__ int32_t pl_wave_max(int32_t v)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Pre-pass, parallel over frames (one wave per frame): the frame's normaliser -- what
// acmod_score subtracts for the all-phones-active list (ptm_mgau.c:393-400), or the
// all-senone minimum under -compallsen -- and every CI phone's normalised state scores
// packed as 4 x int16 per phone, so that the sequential kernel reads 8 contiguous bytes
// per lane per frame.
template <int NE>
__global__ __launch_bounds__(256)
void phone_loop_prep_kernel(PlDev p, const uint16_t *__restrict__ sseq, const int16_t *__restrict__ raw,
                            int64_t raw_stride, const int32_t *__restrict__ best_all, int32_t total,
                            int16_t *__restrict__ css)                 // [total][64][NE <= 3 ? 4 : 8]
{
    constexpr int W = NE <= 3 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= total) return;
    const int16_t *row = raw + (size_t)t * raw_stride;
    int32_t nb;
    if (p.norm_mode == 2) nb = best_all[t];
    else if (p.norm_mode == 0) nb = 0;                   // final scores (a scorer without a per-call normaliser: s2_semi_mgau)
    else {
        nb = 0x7fffffff;
        for (int i = lane; i < p.n_list; i += 64) nb = min(nb, (int32_t)row[p.ci_list[i]]);
        nb = -pl_wave_max(-nb);
    }
    int16_t v[W];
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = 0;
    if (lane < p.n_phones) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
        {   // (int16 difference: PTM's never leaves the range, the ms scorer clamps it, ms_mgau.c:269-277)
            const int32_t x = (int32_t)((uint32_t)(int32_t)row[sseq[(size_t)p.ssid[lane] * NE + i]] - (uint32_t)nb);
            v[i] = (int16_t)(x > 32767 ? 32767 : (x < -32768 ? -32768 : x));
        }
    }
    int16_t *o = css + ((size_t)t * 64 + lane) * W;
#pragma unroll
    for (int i = 0; i < W; ++i) o[i] = v[i];
}

// The same from the scorer's top-N lists instead of score rows (psgpu_phone_loop_run_lists_dev): the frame's lists of every
// (codebook, stream) chain are normalised per stream (ptm_mgau_codebook_norm, ptm_mgau.c:265-295), the senones of the
// all-phones-active list -- some 130 of 5126 -- are evaluated (ptm_mgau_senone_eval, :326-403; csrc/psgpu_sen_dev.h), the
// rest as above.  One wavefront per frame; chains <= 128, listed senones <= 256.
constexpr int kPlListMax = 256;
template <int NE>
__global__ __launch_bounds__(256)
void phone_loop_prep_lists_kernel(PlDev p, const uint16_t *__restrict__ sseq, SenModel sm, const uint8_t *__restrict__ la, int32_t la_size,
                                  const int32_t *__restrict__ tsc, const uint32_t *__restrict__ tcw, int32_t n_chain, int32_t total,
                                  int16_t *__restrict__ css)
{
    constexpr int W = NE <= 3 ? 4 : 8;
    __shared__ uint32_t s_cw[4][128], s_sc[4][128];
    __shared__ int32_t s_val[4][kPlListMax];
    __shared__ uint8_t s_la[512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wv;
    const bool live = t < total;
    for (int i = threadIdx.x; i < 512; i += 256) s_la[i] = i < la_size ? la[i] : 0;
    // the frame's lists: chains lane and lane + 64
    int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    uint32_t ca = 0, cb = 0;
    const int c1 = lane + 64;
    if (live && lane < n_chain) {
        const int4 q = *reinterpret_cast<const int4 *>(tsc + ((size_t)lane * total + t) * 4);
        a0 = q.x; a1 = q.y; a2 = q.z; a3 = q.w; ca = tcw[(size_t)lane * total + t];
    }
    if (live && c1 < n_chain) {
        const int4 q = *reinterpret_cast<const int4 *>(tsc + ((size_t)c1 * total + t) * 4);
        b0 = q.x; b1 = q.y; b2 = q.z; b3 = q.w; cb = tcw[(size_t)c1 * total + t];
    }
    int32_t norm[kSenStreams];
#pragma unroll
    for (int q = 0; q < kSenStreams; ++q) {
        int32_t v = kMaxNegInt32;
        if (lane < n_chain && lane % kSenStreams == q) v = a0 >> kSenShift;
        if (c1 < n_chain && c1 % kSenStreams == q) v = max(v, b0 >> kSenShift);
        norm[q] = pl_wave_max(v);
    }
    if (lane < n_chain) { s_sc[wv][lane] = sen_pack_scores(a0, a1, a2, a3, norm[lane % kSenStreams]); s_cw[wv][lane] = ca; }
    if (c1 < n_chain) { s_sc[wv][c1] = sen_pack_scores(b0, b1, b2, b3, norm[c1 % kSenStreams]); s_cw[wv][c1] = cb; }
    __syncthreads();
    int32_t nb = 0x7fffffff;
    if (live)
        for (int i = lane; i < p.n_list; i += 64) {
            const int32_t a = sen_eval_f3n4(sm, s_cw[wv], s_sc[wv], s_la, (int)p.ci_list[i]);
            s_val[wv][i] = a;
            nb = min(nb, (int32_t)(int16_t)a);
        }
    nb = -pl_wave_max(-nb);
    __syncthreads();
    if (!live) return;
    int16_t v[W];
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = 0;
    if (lane < p.n_phones) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int sen = sseq[(size_t)p.ssid[lane] * NE + i];
            int lo = 0, hi = p.n_list - 1;                       // the list is in ascending senone order
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)p.ci_list[mid] < sen) lo = mid + 1; else hi = mid; }
            v[i] = (int16_t)(uint16_t)((uint32_t)(int32_t)(int16_t)s_val[wv][lo] - (uint32_t)nb);
        }
    }
    int16_t *o = css + ((size_t)t * 64 + lane) * W;
#pragma unroll
    for (int i = 0; i < W; ++i) o[i] = v[i];
}

template <int NE>
__global__ __launch_bounds__(64)
void phone_loop_kernel(PlDev p, const uint8_t *__restrict__ tp_g, const int16_t *__restrict__ css,
                       const int32_t *__restrict__ utt_off, int32_t *__restrict__ penalties,
                       int32_t *__restrict__ pen_now, int32_t *__restrict__ state, int32_t *__restrict__ carry_all, int32_t resume)
{
    constexpr int W = NE <= 3 ? 4 : 8;
    constexpr int kAhead = 4;                                     // frames fetched ahead of the one being searched
    __shared__ int32_t s_ring[kPlMaxWindow][64];
    const int lane = threadIdx.x, u = blockIdx.x;
    const int t0 = utt_off[u], T = utt_off[u + 1] - t0;
    if (T <= 0) return;
    const bool on = lane < p.n_phones;
    const int ph = on ? lane : 0;
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) { h.score[i] = kW; h.history[i] = -1; h.senid[i] = (uint16_t)i; }
    h.out_score = kW; h.out_history = -1; h.bestscore = kW;
    // hmm_clear + hmm_enter(hmm, 0, -1, 0) (:170-175)
    h.score[0] = 0;
    int frame = 0;
    // this phone's transition matrix in registers
    uint8_t tpl[NE * (NE + 1)];
#pragma unroll
    for (int i = 0; i < NE * (NE + 1); ++i) tpl[i] = tp_g[(size_t)p.tmatid[ph] * NE * (NE + 1) + i];
    for (int w = 0; w < p.window; ++w) s_ring[w][lane] = 0;       // memset(pen_buf, 0) (:177-178)
    int ptr = 0;
    int32_t best_score = 0;                                       // pls->best_score (:180)
    // psgpu_phone_loop_run_carry_dev: the utterance goes on where the previous call's frames ended -- the phones' HMMs, the
    // penalty ring and its position, the best score; `frame` counts from the utterance's start, so this call's frames are
    // numbered from `base` = the frames of the calls before
    int32_t *const carry = carry_all ? carry_all + (size_t)u * kPlCarryWords : nullptr;
    int base = 0;
    // (its last word but four: the block holds a state -- zeroed by the caller for an utterance that starts afresh among resumed ones)
    if (carry && resume && carry[64 * 8 + kPlMaxWindow * 64 + 3] == kPlCarryMagic) {
        const int32_t *const cs = carry + lane * 8;
#pragma unroll
        for (int i = 0; i < NE; ++i) h.score[i] = cs[i];
        h.out_score = cs[5]; h.bestscore = cs[6]; frame = cs[7];
        for (int w = 0; w < p.window; ++w) s_ring[w][lane] = carry[64 * 8 + w * 64 + lane];
        ptr = carry[64 * 8 + kPlMaxWindow * 64]; best_score = carry[64 * 8 + kPlMaxWindow * 64 + 1]; base = carry[64 * 8 + kPlMaxWindow * 64 + 2];
    }
    // a small register queue of upcoming frames' scores: the march never waits for memory
    typedef int16_t vec_t __attribute__((ext_vector_type(W)));
    const vec_t *in = reinterpret_cast<const vec_t *>(css) + (size_t)t0 * 64 + lane;
    vec_t q[kAhead];
#pragma unroll
    for (int a = 0; a < kAhead; ++a) q[a] = in[(size_t)min(a, T - 1) * 64];
    for (int t = 0; t < T; ++t) {
        int16_t ss[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) ss[i] = q[0][i];
#pragma unroll
        for (int a = 0; a + 1 < kAhead; ++a) q[a] = q[a + 1];
        q[kAhead - 1] = in[(size_t)min(t + kAhead, T - 1) * 64];
        // renormalize_hmms (:186-199): every phone, whether active or not
        if (best_score + 2 * p.beam < kW) {
#pragma unroll
            for (int i = 0; i < NE; ++i) if (h.score[i] > kW) h.score[i] -= best_score;
            if (h.out_score > kW) h.out_score -= best_score;
        }
        // evaluate_hmms (:201-221)
        int32_t sc = kW;
        const int ta = base + t;                                  // the frame's number in its utterance
        const bool act = on && frame >= ta;
        if (act) sc = (NE == 3) ? vit3(h, tpl, ss) : vit5(h, tpl, ss);
        const int32_t bs = pl_wave_max(act ? max(sc, kW) : kW);
        best_score = bs;
        // store_scores (:223-245): (int32)((bestscore - best) * pl_weight), then the window maximum
        const int32_t pen = (int32_t)((double)(h.bestscore - bs) * p.weight);
        s_ring[ptr][lane] = pen;
        ptr = (ptr + 1 == p.window) ? 0 : ptr + 1;
        int32_t mx = kW;
        for (int w = 0; w < p.window; ++w) mx = max(mx, s_ring[w][lane]);
        if (on) {
            penalties[(size_t)(t0 + t) * p.n_phones + lane] = mx;
            if (pen_now) pen_now[(size_t)(t0 + t) * p.n_phones + lane] = pen;
        }
        // prune_hmms (:247-266)
        if (act) {
            if (h.bestscore > bs + p.beam) frame = ta + 1;
            else {                                                // hmm_clear_scores
#pragma unroll
                for (int i = 0; i < NE; ++i) h.score[i] = kW;
                h.out_score = kW; h.bestscore = kW;
            }
        }
        // phone_transition (:268-300): every phone is entered by the best exiting phone
        const int32_t np = (on && frame == ta + 1) ? h.out_score + p.pip : kMaxNegInt32;
        const bool exits = on && frame == ta + 1 && np > bs + p.pbeam;
        const int32_t m = pl_wave_max(exits ? np : kMaxNegInt32);
        if (m != kMaxNegInt32 && on) {
            if (frame < ta || m > h.score[0]) { h.score[0] = m; frame = ta + 1; }
        }
        if (on && state) {
            int32_t *st = state + ((size_t)(t0 + t) * p.n_phones + lane) * 8;
#pragma unroll
            for (int i = 0; i < NE; ++i) st[i] = h.score[i];
            st[5] = h.out_score; st[6] = h.bestscore; st[7] = frame;
        }
    }
    if (carry) {
        int32_t *const cs = carry + lane * 8;
#pragma unroll
        for (int i = 0; i < NE; ++i) cs[i] = h.score[i];
        cs[5] = h.out_score; cs[6] = h.bestscore; cs[7] = frame;
        for (int w = 0; w < p.window; ++w) carry[64 * 8 + w * 64 + lane] = s_ring[w][lane];
        if (lane == 0) { carry[64 * 8 + kPlMaxWindow * 64] = ptr; carry[64 * 8 + kPlMaxWindow * 64 + 1] = best_score; carry[64 * 8 + kPlMaxWindow * 64 + 2] = base + T; carry[64 * 8 + kPlMaxWindow * 64 + 3] = kPlCarryMagic; }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
extern "C" {

int psgpu_hmm_ctx_create(psgpu_hmm_ctx_t **out, int32_t n_emit_state, int32_t n_tmat,
                         const uint8_t *tp, int32_t n_sseq, const uint16_t *sseq, int32_t n_sen)
{
    PSGPU_REQUIRE(out && tp && sseq, "psgpu_hmm_ctx_create: NULL argument");
    PSGPU_REQUIRE(n_emit_state >= 1 && n_emit_state <= 5, "n_emit_state %d outside 1..5 (HMM_MAX_NSTATE, hmm.h)", n_emit_state);
    PSGPU_REQUIRE(n_tmat > 0 && n_tmat < 32768 && n_sseq > 0 && n_sseq < 65535 && n_sen > 0,
                  "bad table sizes (n_tmat %d, n_sseq %d, n_sen %d)", n_tmat, n_sseq, n_sen);
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    psgpu_hmm_ctx_t *c = new psgpu_hmm_ctx_t();
    memset(c, 0, sizeof *c);
    c->n_emit = n_emit_state; c->n_tmat = n_tmat; c->n_sseq = n_sseq; c->n_sen = n_sen;
    const size_t tpb = (size_t)n_tmat * n_emit_state * (n_emit_state + 1);
    hipError_t e = hipMalloc((void **)&c->tp, tpb);
    if (e == hipSuccess) e = hipMemcpy(c->tp, tp, tpb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&c->sseq, (size_t)n_sseq * n_emit_state * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(c->sseq, sseq, (size_t)n_sseq * n_emit_state * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_scr, (size_t)n_sen * sizeof(int16_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_scr, (size_t)n_sen * sizeof(int16_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_best, sizeof(int32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_best, sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->z_recs, (size_t)kZeroCopyMax * sizeof(psgpu_hmm_rec_t), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->z_scr, (size_t)n_sen * sizeof(int16_t), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->z_word, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->zd_recs, c->z_recs, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->zd_scr, c->z_scr, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->zd_word, c->z_word, 0);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_count, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->d_count, 0, sizeof(uint32_t));
    if (e != hipSuccess) {
        psgpu_set_error("psgpu_hmm_ctx_create: %s", hipGetErrorString(e));
        psgpu_hmm_ctx_free(c);
        return e == hipErrorOutOfMemory ? PSGPU_ENOMEM : PSGPU_EHIP;
    }
    *out = c;
    return PSGPU_OK;
}

void psgpu_hmm_ctx_free(psgpu_hmm_ctx_t *c)
{
    if (!c) return;
    if (c->stream) hipStreamDestroy(c->stream);
    hipFree(c->tp); hipFree(c->sseq); hipFree(c->d_recs); hipFree(c->d_scr); hipFree(c->d_best);
    if (c->h_recs) hipHostFree(c->h_recs);
    if (c->h_scr) hipHostFree(c->h_scr);
    if (c->h_best) hipHostFree(c->h_best);
    if (c->z_recs) hipHostFree(c->z_recs);
    if (c->z_scr) hipHostFree(c->z_scr);
    if (c->z_word) hipHostFree(c->z_word);
    hipFree(c->d_count); hipFree(c->pl_css);
    delete c;
}

int32_t psgpu_hmm_n_emit_state(const psgpu_hmm_ctx_t *c) { return c->n_emit; }

int psgpu_hmm_vit_eval_dev(psgpu_hmm_ctx_t *c, psgpu_hmm_rec_t *recs_dev,
                           const int32_t *active_idx_dev, int32_t n_active,
                           const uint16_t *utt_of_hmm_dev,
                           const int16_t *senscr_dev, int32_t senscr_stride,
                           int32_t *best_dev, void *stream)
{
    PSGPU_REQUIRE(c && recs_dev && senscr_dev, "psgpu_hmm_vit_eval_dev: NULL argument");
    PSGPU_REQUIRE(n_active >= 0, "negative n_active");
    if (n_active == 0) return PSGPU_OK;
    // persistent grid: at most 8 workgroups of 4 waves per CU on 256 CUs
    int blocks = (n_active + kHmmThreads - 1) / kHmmThreads;
    // (persistent workgroups: as many as stay resident together -- 20.5 KB of LDS each, seven on a compute unit of 256)
    static const int max_blocks = [] { const char *e = getenv("PSGPU_HMM_BLOCKS"); return e ? atoi(e) : 1792; }();
    if (blocks > max_blocks) blocks = max_blocks;
    const int32_t tpb = c->n_tmat * c->n_emit * (c->n_emit + 1);
#define HMM_LAUNCH_(NE, STREAM)                                                                                        \
    hipLaunchKernelGGL((hmm_vit_kernel<NE, STREAM>), dim3(blocks), dim3(kHmmThreads), 0, (hipStream_t)stream,           \
                       recs_dev, active_idx_dev, n_active, utt_of_hmm_dev, senscr_dev, senscr_stride,                   \
                       (const uint8_t *)c->tp, tpb, (const uint16_t *)c->sseq, best_dev,                                \
                       c->launch_done_count, c->launch_done_word, c->launch_seq)
    // (a dense list of more than 2^20 records -- 64 MB -- is streamed around the caches)
#define HMM_LAUNCH(NE) do { if (!active_idx_dev && n_active > (1 << 20)) HMM_LAUNCH_(NE, true); else HMM_LAUNCH_(NE, false); } while (0)
    switch (c->n_emit) {
    case 1: HMM_LAUNCH(1); break;
    case 2: HMM_LAUNCH(2); break;
    case 3: HMM_LAUNCH(3); break;
    case 4: HMM_LAUNCH(4); break;
    default: HMM_LAUNCH(5); break;
    }
#undef HMM_LAUNCH
#undef HMM_LAUNCH_
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

void *psgpu_hmm_ctx_stream(psgpu_hmm_ctx_t *c) { return c ? (void *)c->stream : nullptr; }

struct PlListsArg { const psgpu_ptm_view_t *v; const int32_t *tsc; const uint8_t *tcw; };
static int pl_run(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                  const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                  const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev, const PlListsArg *ls,
                  const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                  int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream, int32_t *carry_dev = nullptr, int32_t resume = 0);

int psgpu_phone_loop_run_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                             const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                             const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev,
                             const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                             int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream)
{
    PSGPU_REQUIRE(raw_dev || n_utt == 0, "psgpu_phone_loop_run_dev: NULL score rows");
    return pl_run(c, pp, ssid_dev, tmatid_dev, ci_list_dev, n_list, raw_dev, raw_stride, best_dev, nullptr, utt_off_dev, n_utt, total_frames,
                  penalties_dev, pen_now_dev, state_dev, stream);
}

int32_t psgpu_phone_loop_carry_words(void) { return kPlCarryWords; }

int psgpu_phone_loop_carry_restart(int32_t *carry_dev, int32_t u, void *stream)
{
    PSGPU_REQUIRE(carry_dev && u >= 0, "psgpu_phone_loop_carry_restart: bad argument");
    PSGPU_HIP(hipMemsetAsync(carry_dev + (size_t)u * kPlCarryWords + 64 * 8 + kPlMaxWindow * 64 + 3, 0, sizeof(int32_t), (hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_phone_loop_run_carry_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                                   const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                                   const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev,
                                   const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                                   int32_t *penalties_dev, int32_t *carry_dev, int32_t resume, void *stream)
{
    PSGPU_REQUIRE((raw_dev && carry_dev) || n_utt == 0, "psgpu_phone_loop_run_carry_dev: NULL score rows / carry buffer");
    return pl_run(c, pp, ssid_dev, tmatid_dev, ci_list_dev, n_list, raw_dev, raw_stride, best_dev, nullptr, utt_off_dev, n_utt, total_frames,
                  penalties_dev, nullptr, nullptr, stream, carry_dev, resume != 0);
}

int psgpu_phone_loop_run_lists_dev(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                                   const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                                   const psgpu_ptm_view_t *v, const int32_t *topn_score_dev, const uint8_t *topn_cw_dev,
                                   const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                                   int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream)
{
    PSGPU_REQUIRE(v && (n_utt == 0 || (topn_score_dev && topn_cw_dev)), "psgpu_phone_loop_run_lists_dev: NULL argument");
    PSGPU_REQUIRE(v->n_feat == kSenStreams && v->topn == kSenTopn && v->n_mgau * v->n_feat <= 128 && n_list >= 1 && n_list <= kPlListMax
                  && v->logadd8_size >= 256 && v->mixw_sen,
                  "psgpu_phone_loop_run_lists_dev: a 3-stream top-4 scorer of at most 128 chains and a list of at most %d senones", kPlListMax);
    const PlListsArg ls = { v, topn_score_dev, topn_cw_dev };
    return pl_run(c, pp, ssid_dev, tmatid_dev, ci_list_dev, n_list, nullptr, 0, nullptr, &ls, utt_off_dev, n_utt, total_frames,
                  penalties_dev, pen_now_dev, state_dev, stream);
}

static int pl_run(psgpu_hmm_ctx_t *c, const psgpu_phone_loop_params_t *pp, const uint16_t *ssid_dev,
                  const int16_t *tmatid_dev, const uint16_t *ci_list_dev, int32_t n_list,
                  const int16_t *raw_dev, int64_t raw_stride, const int32_t *best_dev, const PlListsArg *ls,
                  const int32_t *utt_off_dev, int32_t n_utt, int32_t total_frames,
                  int32_t *penalties_dev, int32_t *pen_now_dev, int32_t *state_dev, void *stream, int32_t *carry_dev, int32_t resume)
{
    PSGPU_REQUIRE(c && pp && n_utt >= 0, "psgpu_phone_loop_run_dev: bad argument");
    PSGPU_REQUIRE(c->n_emit == 3 || c->n_emit == 5, "phone loop: %d emitting states (3 or 5 are built)", c->n_emit);
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE(ssid_dev && tmatid_dev && (raw_dev || ls) && utt_off_dev && penalties_dev, "psgpu_phone_loop_run_dev: NULL device buffer");
    PSGPU_REQUIRE(pp->n_phones >= 1 && pp->n_phones <= 64, "n_phones %d outside 1..64", pp->n_phones);
    PSGPU_REQUIRE(pp->window >= 1 && pp->window <= kPlMaxWindow, "window %d outside 1..%d", pp->window, kPlMaxWindow);
    PSGPU_REQUIRE(!(best_dev != nullptr && ci_list_dev != nullptr && n_list > 0),
                  "at most one of best_dev (compallsen) and ci_list_dev (active-list normalisation); neither: the rows are final scores");
    PSGPU_REQUIRE(!ls || ci_list_dev, "scoring from lists needs the all-phones-active senone list");
    PlDev p;
    p.n_phones = pp->n_phones; p.window = pp->window; p.beam = pp->beam; p.pbeam = pp->pbeam; p.pip = pp->pip;
    p.n_list = n_list; p.norm_mode = best_dev ? 2 : ((ci_list_dev && n_list > 0) ? 1 : 0); p.weight = pp->penalty_weight;
    p.ssid = ssid_dev; p.tmatid = tmatid_dev; p.ci_list = ci_list_dev;
    // (NULL is the default stream, as for every other entry point.  It used to select the context's own non-blocking stream:
    //  a caller that ran the scorer before and the search after this call on the default stream then raced with it.)
    hipStream_t st = (hipStream_t)stream;
    PSGPU_REQUIRE(total_frames >= 0, "negative frame count");
    if (total_frames == 0) return PSGPU_OK;
    const int W = c->n_emit <= 3 ? 4 : 8;
    if (total_frames > c->pl_cap) {
        PSGPU_HIP(hipFree(c->pl_css)); c->pl_css = nullptr; c->pl_cap = 0;
        PSGPU_HIP(hipMalloc((void **)&c->pl_css, sizeof(int16_t) * (size_t)total_frames * 64 * W));
        c->pl_cap = total_frames;
    }
    const dim3 pg((total_frames + 3) / 4);
    SenModel sm = { nullptr, nullptr, 0, 0 };
    if (ls) { sm.mixw = ls->v->mixw_sen; sm.sen2cb = ls->v->sen2cb; sm.n_sen = ls->v->n_sen; sm.n_density = ls->v->n_density; }
    if (c->n_emit == 3) {
        if (ls)
            hipLaunchKernelGGL((phone_loop_prep_lists_kernel<3>), pg, dim3(256), 0, st, p, (const uint16_t *)c->sseq, sm, ls->v->logadd8,
                               ls->v->logadd8_size, ls->tsc, reinterpret_cast<const uint32_t *>(ls->tcw), ls->v->n_mgau * ls->v->n_feat,
                               total_frames, c->pl_css);
        else
        hipLaunchKernelGGL((phone_loop_prep_kernel<3>), pg, dim3(256), 0, st, p, (const uint16_t *)c->sseq, raw_dev, raw_stride,
                           best_dev, total_frames, c->pl_css);
        hipLaunchKernelGGL((phone_loop_kernel<3>), dim3(n_utt), dim3(64), 0, st, p, (const uint8_t *)c->tp,
                           (const int16_t *)c->pl_css, utt_off_dev, penalties_dev, pen_now_dev, state_dev, carry_dev, resume);
    }
    else {
        if (ls)
            hipLaunchKernelGGL((phone_loop_prep_lists_kernel<5>), pg, dim3(256), 0, st, p, (const uint16_t *)c->sseq, sm, ls->v->logadd8,
                               ls->v->logadd8_size, ls->tsc, reinterpret_cast<const uint32_t *>(ls->tcw), ls->v->n_mgau * ls->v->n_feat,
                               total_frames, c->pl_css);
        else
        hipLaunchKernelGGL((phone_loop_prep_kernel<5>), pg, dim3(256), 0, st, p, (const uint16_t *)c->sseq, raw_dev, raw_stride,
                           best_dev, total_frames, c->pl_css);
        hipLaunchKernelGGL((phone_loop_kernel<5>), dim3(n_utt), dim3(64), 0, st, p, (const uint8_t *)c->tp,
                           (const int16_t *)c->pl_css, utt_off_dev, penalties_dev, pen_now_dev, state_dev, carry_dev, resume);
    }
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

int psgpu_hmm_vit_eval(psgpu_hmm_ctx_t *c, psgpu_hmm_rec_t *recs, int32_t n,
                       const int16_t *senscr, int32_t *best)
{
    PSGPU_REQUIRE(c && senscr && (recs || n == 0), "psgpu_hmm_vit_eval: NULL argument");
    PSGPU_REQUIRE(n >= 0, "negative n");
    if (best) *best = kWorstScore;
    if (n == 0) return PSGPU_OK;
    static const int no_zero_copy = [] { const char *e = getenv("PSGPU_NO_POLL"); return e ? atoi(e) : 0; }();
    if (n <= kZeroCopyMax && !no_zero_copy) {
        // small batch (a single decoder's active list): the kernel works on host-mapped
        // records and scores over PCIe and publishes a completion word; no copies, no sync call
        memcpy(c->z_recs, recs, (size_t)n * sizeof(psgpu_hmm_rec_t));
        memcpy(c->z_scr, senscr, (size_t)c->n_sen * sizeof(int16_t));
        c->z_word[1] = (uint32_t)kWorstScore;
        const uint32_t seq = ++c->seq ? c->seq : ++c->seq;
        c->launch_done_count = c->d_count; c->launch_done_word = c->zd_word; c->launch_seq = seq;
        int rc = psgpu_hmm_vit_eval_dev(c, c->zd_recs, nullptr, n, nullptr, c->zd_scr, c->n_sen,
                                        reinterpret_cast<int32_t *>(c->zd_word + 1), c->stream);
        c->launch_done_count = nullptr; c->launch_done_word = nullptr;
        if (rc != PSGPU_OK) return rc;
        bool done = false;
        volatile uint32_t *w = c->z_word;
        for (long i = 0; i < 200000000L; ++i) {
            if (w[0] == seq) { done = true; break; }
            __builtin_ia32_pause();
        }
        if (!done) PSGPU_HIP(hipStreamSynchronize(c->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        memcpy(recs, c->z_recs, (size_t)n * sizeof(psgpu_hmm_rec_t));
        if (best) *best = (int32_t)w[1];
        return PSGPU_OK;
    }
    if (n > c->cap) {
        const int32_t cap = n < 1024 ? 1024 : n + n / 2;
        if (c->h_recs) hipHostFree(c->h_recs);
        hipFree(c->d_recs);
        c->h_recs = nullptr; c->d_recs = nullptr; c->cap = 0;
        PSGPU_HIP(hipHostMalloc((void **)&c->h_recs, (size_t)cap * sizeof(psgpu_hmm_rec_t), hipHostMallocDefault));
        PSGPU_HIP(hipMalloc((void **)&c->d_recs, (size_t)cap * sizeof(psgpu_hmm_rec_t)));
        c->cap = cap;
    }
    memcpy(c->h_recs, recs, (size_t)n * sizeof(psgpu_hmm_rec_t));
    memcpy(c->h_scr, senscr, (size_t)c->n_sen * sizeof(int16_t));
    *c->h_best = kWorstScore;       // hmm.h:84: the evaluate loops start from WORST_SCORE
    PSGPU_HIP(hipMemcpyAsync(c->d_recs, c->h_recs, (size_t)n * sizeof(psgpu_hmm_rec_t), hipMemcpyHostToDevice, c->stream));
    PSGPU_HIP(hipMemcpyAsync(c->d_scr, c->h_scr, (size_t)c->n_sen * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    PSGPU_HIP(hipMemcpyAsync(c->d_best, c->h_best, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    int rc = psgpu_hmm_vit_eval_dev(c, c->d_recs, nullptr, n, nullptr, c->d_scr, c->n_sen, c->d_best, c->stream);
    if (rc != PSGPU_OK) return rc;
    PSGPU_HIP(hipMemcpyAsync(c->h_recs, c->d_recs, (size_t)n * sizeof(psgpu_hmm_rec_t), hipMemcpyDeviceToHost, c->stream));
    PSGPU_HIP(hipMemcpyAsync(c->h_best, c->d_best, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    PSGPU_HIP(hipStreamSynchronize(c->stream));
    memcpy(recs, c->h_recs, (size_t)n * sizeof(psgpu_hmm_rec_t));
    if (best) *best = *c->h_best;
    return PSGPU_OK;
}

}  // extern "C"

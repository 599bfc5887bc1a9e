// psgpu_flat.hip -- the flat-lexicon second pass on gfx950: whole
// utterances, one workgroup per utterance, every frame inside the kernel.
//
// Replaces ngram_fwdflat_start + ngram_fwdflat_search x T + ngram_fwdflat_finish (reference
// src/ngram_search_fwdflat.c:223-414, 416-877, 925-960) and the back-pointer helpers they share with the first
// pass (src/ngram_search.c:301-498): the utterance's vocabulary from the first pass's back-pointer table, one HMM
// chain per word ([multiplex root][word-internal phones][right-context fan-out], contiguous), evaluation, beam
// pruning with phone transitions inside the chain, word exits into the back-pointer table, word transitions
// with the float-weighted language score (:700-706), silence / filler entry, the next active word list.
// Output: back-pointer table, score stack and frame marks in the reference's columns, as the first pass's kernel.
//
// Parallelism: utterances across workgroups (two a compute unit; the host orders the launch so that the largest share a compute unit
// with the smallest).  Inside a frame: the active channels gathered sixteen work-items a word; one work-item per
// active CHANNEL for the evaluation and for the pruning's decision (a channel's entry state is written by its one predecessor only);
// word exits queued in LDS, ranked by counting, a word's entry = the maximum of its exits' keys (the walk in queue order only where
// their histories name different real words); word transitions one work-item per (word of the frame's window, new entry) PAIR, the
// best pair of a word one 64-bit maximum ("the first best exit wins" = the earliest index among the highest scores); workgroup prefix
// sums for back-pointer positions and the next active word list (vocabulary order, then fillers: ngram_search_fwdflat.c:853-869).
// Barriers between phases that exchange through LDS only wait for the LDS counter (ff_sync_lds); where device memory crosses, a full
// one.  The utterance's vocabulary and chain layout are built on the host from the first pass's table (host threads) --
// build_fwdflat_wordlist's list surgery is sequential; an on-device build can replace it behind the same entry.
// Oracle: oracle/ps_oracle_flat.c (pinned to the reference); checked on the CPU through tests/hostsim.
#include "psgpu_hmm_dev.h"
#include "psgpu_lm_dev.h"
#include "psgpu_sen_dev.h"
#include "psgpu_wave_dev.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

// Builds for the workgroup simulator (tests/hostsim: g++, no __HIPCC__) keep, beside the lazily maintained top-N lists of the
// scoring mode, the lists as the reference maintains them -- every chain re-scored and re-ordered in every frame -- and stop
// where a replayed list differs from them.  The product has none of this.
#if !defined(__HIPCC__)
#define PSGPU_FF_CHECK_LAZY 1
#include <cstdio>
extern "C" { int psgpu_sim_ff_exit_cap = 256; }      // (tests shrink it to drive the frames through the other path)
#define FF_EXIT_CAP psgpu_sim_ff_exit_cap
extern "C" { int psgpu_sim_ff_el_cap = 384; }
#define FF_EL_CAP psgpu_sim_ff_el_cap
extern "C" { int psgpu_sim_ff_awl_regs = 512; }
#define FF_AWL_REGS psgpu_sim_ff_awl_regs
extern "C" { int psgpu_sim_ff_pair_rows = 96; }     // new entries of a frame the word transitions take as (word, entry) pairs
#define FF_PAIR_ROWS psgpu_sim_ff_pair_rows
extern "C" { int psgpu_sim_ff_slice_chunk = 256; }   // ... and words of the frame's window a chunk of those pairs takes (<= kFfThreads)
#define FF_SL_CHUNK psgpu_sim_ff_slice_chunk
extern "C" { int psgpu_sim_ff_force_walk = 0; }      // every exiting word's entry by the walk over its exits (the path of words whose exits' histories differ)
#define FF_FORCE_WALK psgpu_sim_ff_force_walk
#else
#define FF_EXIT_CAP kFfMaxExit
#define FF_EL_CAP kFfMaxEl
#define FF_AWL_REGS (kFfRegRows * kFfThreads)
#define FF_PAIR_ROWS kFfNewRows
#define FF_SL_CHUNK kFfThreads
#define FF_FORCE_WALK 0
#endif

constexpr int kFfThreads = 256;
constexpr int kFfMaxCi = 64;
constexpr int kFfMaxSen = 8192;        // senones (LDS bitmap of the frame's active list, scoring mode)
constexpr int kFfMaxEnt = 1024;        // (codebook, stream) chains x top-N entries held in LDS (en-us PTM: 126 x 4)
constexpr int kFfMaxCb = 256;          // codebooks
constexpr int kFfMaxTopn = 8;
constexpr int kFfMaxFan = 128;         // fan-outs into right-context channels queued per frame (more: done in place)
constexpr int kFfChanMask = (1 << 28) - 1, kFfClearBit = 1 << 29, kFfEnteredBit = 1 << 28;   // FfUtt::elist entries
constexpr int kFfMaxEl = 384;          // entries of the frame's active-channel list held in LDS (the rest in the slab)
constexpr int kFfAwlLds = 256;         // active words whose list entries the next frame's gather finds in LDS
constexpr int kFfRegRows = 2;           // vocabularies up to kFfRegRows x 256 words (+ fillers) keep their static records in registers
constexpr int kFfMaxTp = 2048;         // bytes of transition matrices held in LDS (more: read from device memory)
#ifndef PSGPU_FF_MAX_EXIT
#define PSGPU_FF_MAX_EXIT 256
#endif
constexpr int kFfMaxExit = PSGPU_FF_MAX_EXIT;
constexpr int kFfNewRows = 96;          // new entries (exiting WORDS) of a frame whose rows the word transitions find in LDS (30 s of speech, 115 words: at most 33)        // word exits of a frame queued in LDS (more: the frame's exits through the slab's flags)

// Scoring mode (psgpu_fwdflat_search_feats_dev): the kernel is handed the feature rows and the PTM model and produces
// each frame's senone scores itself, as ptm_mgau_frame_eval does when the second pass calls it (ptm_mgau.c:408-454 with
// compallsen off): the frame's active senone list from the active channels (compute_fwdflat_sen_active +
// acmod_flags2list, bridging entries included), the codebooks those senones touch (:297-321), every carried list
// re-scored (eval_topn :87-136), the touched codebooks scanned exactly as the reference scans them (eval_cb :151-226,
// one work-item per (codebook, stream), sequential in codeword order -- the acceptance rule depends on it), the
// per-stream normaliser over the touched codebooks only (:265-295), the listed senones (:326-403).  This is why pass-2
// scores are not a shift of pass-1 rows: in pass 1 the phone loop keeps every codebook touched, here nothing does.
struct alignas(16) FfQuad { int32_t x, y, z, w; };
struct FfRaw {
    psgpu_ptm_view_t pm;
    const float *feats;                  // [total][veclen]
    const int32_t *seed;                 // [n_utt][n_chain][topn] codewords of the history slot pass-2 frame 0 starts from
    // optional: the batch scorer's lists of the same frames (psgpu_ptm_score_batch_dev: chain-major scores / packed codewords /
    // open flags).  Where an entry is not open its list is the top-N of ALL the chain's densities whatever the seeds were
    // (the closed form): exactly what eval_topn + eval_cb of a touched codebook arrive at here, so the kernel
    // takes it instead of scanning 128 densities in one work-item; open entries (ties, out of range) are scanned as before.
    const int32_t *tsc;
    const uint32_t *tcw;
    const uint8_t *open;
    int32_t total;
};

struct FfDev {
    int32_t n_ci, n_emit, n_sen, n_w, n1;
    int32_t beam, pip, silpen, fillpen, fwdflatbeam, fwdflatwbeam, min_ef_width, max_sf_win;
    int32_t startwid, finishwid, silwid, filler_start, filler_end, sil_ci;
    float lwf;
    const int32_t *w1_wid, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx, *w1_of_word;
    const int32_t *d_pronlen, *d_first, *d_last, *d_last2, *d_base, *d_filler;
    const int32_t *rs_n, *rs_ssid, *rs_cimap, *ldiph, *ci_tmat, *lm;
    const int32_t *pron_off, *pron_ci, *pron_ssid, *ci_ssid;
    const uint8_t *tp;
    const uint16_t *sseq;
    int32_t tp_bytes;
    int32_t use_trie;
    int32_t fill_known;                  // a filler word is a word of the language model (it can be in an utterance's vocabulary)
    LmDev trie;
};

// per-utterance state; channels [0, n1) are the permanent single-phone words, then the chains of the vocabulary
struct FfUtt {
    int32_t nwd, n_chan, n_frame, awl_cap;
    const int32_t *wl_wid, *wl_chain, *wl_len, *wl_node_off, *node_sf;     // [nwd] (+1), [n nodes]: vocabulary, host-built
    const int32_t *fr_off, *fr_words;    // the same nodes by start frame: [n_frame + 2] offsets into [n nodes] vocabulary positions
    int32_t *wseen;                      // [nwd + 1] frame stamp: the word was taken as a successor in that frame already
    int32_t *wstat;                      // [nwd + 1][4] a vocabulary word's static side for the word transitions: word, first channel, first phone | second << 8, base word
    int32_t *score, *hist;               // [C][5]
    int32_t *out, *outh, *best, *frame;  // [C]
    int32_t *senid;                      // [C][5]
    int32_t *tmat, *mpx, *rcid, *xflag;  // [C]
    int32_t *elist;                      // [C] the frame's active channels (evaluation work list; bit 30: </s>'s root, bit 29: see the pruning)
    int32_t *einfo;                      // [C] per entry of elist: position of the channel's word in the active word list << 10 | position in its chain
    int32_t *wchain, *wlen, *wrcs;       // [n_w] chain offset (channel index) or -1, chain length, right-context channels (0: single phone)
    int32_t *word_active, *word_lat_idx; // [n_w] (word_active: frame stamp)
    int32_t *awl[2];                     // [awl_cap][3] the active words: word, first channel, channels | right contexts << 10 | single phone << 20
    int32_t *cnt_a, *cnt_b;              // [awl_cap + 1] scan scratch of the exits' slab path
    int32_t *bp, *bss, *bp_table_idx, *step, *result;
    const int32_t *w1_ssid_in;           // [n1][n_emit] or NULL
    int16_t *nrow; int32_t *nrow32;      // [n_sen] the frame's scores (scoring mode)
    int32_t bp_cap, bss_cap;
};

// What the kernel is actually handed per utterance: the same fields as offsets (in int32 units) from buffers that are
// kernel arguments.  Pointers loaded from memory are generic to the compiler (every access a flat_load / flat_store, both
// wait counters); pointers formed from a kernel argument are global.
#define FF_SLAB_FIELDS(X) X(wstat) X(score) X(hist) X(out) X(outh) X(best) X(frame) X(senid) X(tmat) X(mpx) X(rcid) X(xflag) X(elist) X(einfo) X(wchain) \
    X(wlen) X(wrcs) X(wseen) X(word_active) X(word_lat_idx) X(cnt_a) X(cnt_b)
#define FF_VOC_FIELDS(X) X(wl_wid) X(wl_chain) X(wl_len) X(wl_node_off) X(node_sf) X(fr_off) X(fr_words)
struct FfOff {
#define X(f) int64_t f;
    FF_SLAB_FIELDS(X) FF_VOC_FIELDS(X)
#undef X
    int64_t awl0, awl1, nrow32, nrow;    // nrow32 / nrow: -1 when the scores are given
    int32_t nwd, n_chan, n_frame, awl_cap;
    int32_t utt, pad_;                   // the utterance this workgroup searches (the launch order is the host's: see ff_search)
};
struct FfBufs {
    int32_t *slab; const int32_t *voc; int32_t *bp, *bss, *idx, *step, *res; const int32_t *w1_ssid;
    int32_t bp_cap, bss_cap, max_frames;
    long long *prof;                     // PSGPU_FT_PROFILE builds: [n_utt][16] cycles per phase (tools/build_prof_lib.py)
};
// per-phase cycle counts of work-item 0 (a profiling build only: -DPSGPU_FT_PROFILE; the product kernel has none of it)
#ifdef PSGPU_FT_PROFILE
#define FF_PROF(i) do { if (tid == 0) { const long long t_ = clock64(); s_prof[i] += t_ - s_last; s_last = t_; } } while (0)
#define FF_PROFS(i) do { if (tid == 0) s_prof[i] += clock64() - s_last; } while (0)      /* since the last FF_PROF, without moving it */
#else
#define FF_PROFS(i) do { } while (0)
#define FF_PROF(i) do { } while (0)
#endif

struct psgpu_fwdflat_s {
    FfDev d;
    std::vector<void *> allocs;
    std::vector<int32_t> h_pronlen, h_last, h_last2, h_rs_n, h_known;   // host copies for the vocabulary build
    // a search call's working buffers, kept from call to call and only ever grown (a batch call of 512 x 30 s: 50 MB of slab, 110 MB of
    // first-pass columns on the host; allocating, freeing and copying through pageable memory each call cost more than the copies):
    // device [0] slab, [1] vocabularies, [2] descriptors; pinned host [3] first-pass columns, [4] vocabularies, [5] descriptors
    void *work[6] = {};
    size_t work_bytes[6] = {};
};
// the buffer k of at least `bytes` bytes (its contents are not kept when it grows)
static int ff_work(psgpu_fwdflat_s *m, int k, size_t bytes, void **out)
{
    if (bytes > m->work_bytes[k]) {
        if (m->work[k]) { if (k < 3) hipFree(m->work[k]); else hipHostFree(m->work[k]); }
        m->work[k] = nullptr; m->work_bytes[k] = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        PSGPU_HIP(k < 3 ? hipMalloc(&m->work[k], want) : hipHostMalloc(&m->work[k], want, hipHostMallocDefault));
        m->work_bytes[k] = want;
    }
    *out = m->work[k];
    return PSGPU_OK;
}

#define FBP(u, col, i) ((u).bp[(size_t)(col) * (u).bp_cap + (i)])
enum { F_FRAME, F_VALID, F_WID, F_BP, F_SCORE, F_SIDX, F_REAL, F_PREAL, F_LAST, F_LAST2 };

__device__ __forceinline__ void ff_clear(const FfDev &p, FfUtt &u, int c)             // hmm_clear, hmm.c:181-198
{
    for (int i = 0; i < p.n_emit; ++i) { u.score[c * 5 + i] = kW; u.hist[c * 5 + i] = -1; }
    u.out[c] = kW; u.outh[c] = -1; u.best[c] = kW; u.frame[c] = -1;
}
__device__ __forceinline__ void ff_clear_scores(const FfDev &p, FfUtt &u, int c)      // hmm_clear_scores, hmm.c:167-179
{
    for (int i = 0; i < p.n_emit; ++i) u.score[c * 5 + i] = kW;
    u.out[c] = kW; u.best[c] = kW;
}
__device__ __forceinline__ void ff_init(const FfDev &p, FfUtt &u, int c, int mpx, int ssid, int tmatid, int rcid)
{
    u.mpx[c] = mpx; u.tmat[c] = tmatid; u.rcid[c] = rcid; u.xflag[c] = 0;
    if (mpx) {
        u.senid[c * 5] = ssid;
        for (int i = 1; i < p.n_emit; ++i) u.senid[c * 5 + i] = kBadSsid;
    }
    else
        for (int i = 0; i < p.n_emit; ++i) u.senid[c * 5 + i] = p.sseq[(size_t)ssid * p.n_emit + i];
    ff_clear(p, u, c);
}
__device__ __forceinline__ void ff_enter(FfUtt &u, int c, int32_t score, int32_t hist, int frame)
{
    u.score[c * 5] = score; u.hist[c * 5] = hist; u.frame[c] = frame;
}
__device__ __forceinline__ void ff_enter_if_better(FfUtt &u, int c, int32_t score, int32_t hist, int cf)
{
    if (u.frame[c] < cf || score > u.score[c * 5]) ff_enter(u, c, score, hist, cf + 1);
}
__device__ __forceinline__ void ff_normalize(const FfDev &p, FfUtt &u, int c, int32_t norm)
{
    for (int i = 0; i < p.n_emit; ++i) if (u.score[c * 5 + i] > kW) u.score[c * 5 + i] -= norm;
    if (u.out[c] > kW) u.out[c] -= norm;
}

template <int NE>
__device__ __forceinline__ int32_t ff_eval(const FfDev &p, FfUtt &u, int c, const int16_t *row, const uint8_t *tp_all, int32_t &out_score, int32_t &out_hist, int32_t &score0)
{
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        h.score[i] = i < NE ? u.score[c * 5 + i] : kW;
        h.history[i] = i < NE ? u.hist[c * 5 + i] : -1;
        h.senid[i] = i < NE ? (uint16_t)u.senid[c * 5 + i] : 0;
    }
    h.out_score = u.out[c]; h.out_history = u.outh[c]; h.bestscore = u.best[c];
    const uint8_t *tp = tp_all + (size_t)u.tmat[c] * NE * (NE + 1);
    int32_t b;
    if (NE == 3) b = u.mpx[c] ? vit3_mpx(h, tp, row, p.sseq) : vit3(h, tp, row);
    else         b = u.mpx[c] ? vit5_mpx(h, tp, row, p.sseq) : vit5(h, tp, row);
#pragma unroll
    for (int i = 0; i < NE; ++i) { u.score[c * 5 + i] = h.score[i]; u.hist[c * 5 + i] = h.history[i]; u.senid[c * 5 + i] = h.senid[i]; }
    u.out[c] = h.out_score; u.outh[c] = h.out_history; u.best[c] = h.bestscore;
    out_score = h.out_score; out_hist = h.out_history; score0 = h.score[0];
    return b;
}

__device__ __forceinline__ int32_t ff_lm(const FfDev &p, int w3, int w2, int w1)      // ngram_tg_score(...) >> SENSCR_SHIFT
{
    if (p.use_trie) { int nu = 0; return lm_tg_score_one(p.trie, w3, w2, w1, nu) >> 10; }     // (one model: psgpu_fwdflat_set_lm refuses an interpolated set)
    const size_t n1 = (size_t)p.n_w + 1;
    return p.lm[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}
// set_real_wid, ngram_search.c:341-372
__device__ __forceinline__ void ff_set_real_wid(const FfDev &p, FfUtt &u, int bp)
{
    const int prev = FBP(u, F_BP, bp), wid = FBP(u, F_WID, bp);
    if (p.d_filler[wid]) {
        if (prev != -1) { FBP(u, F_REAL, bp) = FBP(u, F_REAL, prev); FBP(u, F_PREAL, bp) = FBP(u, F_PREAL, prev); }
        else { FBP(u, F_REAL, bp) = p.d_base[wid]; FBP(u, F_PREAL, bp) = -1; }
    }
    else {
        FBP(u, F_REAL, bp) = p.d_base[wid];
        FBP(u, F_PREAL, bp) = prev != -1 ? FBP(u, F_REAL, prev) : -1;
    }
}
// ngram_search_save_bp, ngram_search.c:376-498, with the position of a new entry (bpidx, bss_head) given by the caller
__device__ __forceinline__ void ff_save_bp(const FfDev &p, FfUtt &u, int32_t bpidx, int32_t bss_head, int frame, int w, int32_t score,
                           int32_t path, int rc)
{
    const int bp = u.word_lat_idx[w];
    if (bp != -1) {
        if (FBP(u, F_SCORE, bp) < score) {
            const int ob = FBP(u, F_BP, bp);
            if (ob != path) {
                const int32_t b0 = ob == -1 ? -1 : FBP(u, F_PREAL, ob), b1 = ob == -1 ? -1 : FBP(u, F_REAL, ob);
                const int32_t n0 = path == -1 ? -1 : FBP(u, F_PREAL, path), n1 = path == -1 ? -1 : FBP(u, F_REAL, path);
                if (b0 != n0 || b1 != n1) ff_set_real_wid(p, u, bp);      // with the old bp still in place, as the reference
                FBP(u, F_BP, bp) = path;
            }
            FBP(u, F_SCORE, bp) = score;
        }
        if (FBP(u, F_SIDX, bp) != -1) u.bss[FBP(u, F_SIDX, bp) + rc] = score;
        return;
    }
    u.word_lat_idx[w] = bpidx;
    FBP(u, F_WID, bpidx) = w; FBP(u, F_FRAME, bpidx) = frame; FBP(u, F_BP, bpidx) = path; FBP(u, F_SCORE, bpidx) = score;
    FBP(u, F_SIDX, bpidx) = bss_head; FBP(u, F_VALID, bpidx) = 1;
    FBP(u, F_LAST, bpidx) = p.d_last[w];
    int rcsize = 0;
    if (p.d_pronlen[w] == 1) { FBP(u, F_LAST2, bpidx) = -1; FBP(u, F_SIDX, bpidx) = -1; }
    else {
        FBP(u, F_LAST2, bpidx) = p.d_last2[w];
        rcsize = p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]];
    }
    for (int i = 0; i < rcsize; ++i) u.bss[bss_head + i] = kW;
    if (rcsize) u.bss[bss_head + rc] = score;
    ff_set_real_wid(p, u, bpidx);
}

// ... when the word has no entry in this frame yet (the branch at ngram_search.c:438-497; set_real_wid :341-372), with everything
// it reads handed in by the caller (the pruning asked for it when it queued the exit): stores only.  (The word's block of the score
// stack is the caller's.)
__device__ __forceinline__ void ff_new_bp(FfUtt &u, int32_t bpidx, int32_t bss_head, int frame, int w, int32_t score, int32_t path,
                                          bool single, int32_t last, int32_t last2, int32_t base, bool filler,
                                          int32_t path_real, int32_t path_preal)
{
    // (word_lat_idx -- "the word's entry of this frame" -- is the slab path's (ff_save_bp): here a word's exits are found side by side in
    //  the sorted queue, and the array stays -1 throughout)
    FBP(u, F_WID, bpidx) = w; FBP(u, F_FRAME, bpidx) = frame; FBP(u, F_BP, bpidx) = path; FBP(u, F_SCORE, bpidx) = score;
    FBP(u, F_SIDX, bpidx) = single ? -1 : bss_head; FBP(u, F_VALID, bpidx) = 1;
    FBP(u, F_LAST, bpidx) = last; FBP(u, F_LAST2, bpidx) = last2;
    if (filler) { FBP(u, F_REAL, bpidx) = path != -1 ? path_real : base; FBP(u, F_PREAL, bpidx) = path != -1 ? path_preal : -1; }
    else { FBP(u, F_REAL, bpidx) = base; FBP(u, F_PREAL, bpidx) = path_real; }
}

// exclusive prefix sum of a[0..n) in place by the whole workgroup; returns the total.  Ends with a barrier.
__device__ __forceinline__ int32_t ff_block_scan(int32_t *a, int n, int32_t *tmp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (n + kFfThreads - 1) / kFfThreads;
    const int b = min(n, tid * per), e = min(n, b + per);
    int32_t sum = 0;
    for (int i = b; i < e; ++i) sum += a[i];
    const int32_t incl = ft_wave_incl<FtAdd>(sum);
    if (lane == 63) tmp[tid >> 6] = incl;
    __syncthreads();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kFfThreads / 64; ++w) { const int32_t t = tmp[w]; total += t; if (w < (tid >> 6)) base += t; }
    int32_t run = base + incl - sum;
    for (int i = b; i < e; ++i) { const int32_t k = a[i]; a[i] = run; run += k; }
    __syncthreads();
    return total;
}

// first channel and chain length of word w (a vocabulary word's chain, else its permanent single-phone channel)
__device__ __forceinline__ int ff_root(const FfDev &p, const FfUtt &u, int w, int &len)
{
    const int c = u.wchain[w];
    if (c >= 0) { len = u.wlen[w]; return c; }
    len = 1;
    return p.w1_of_word[w];
}

// float -> int32 as the reference does (ptm_mgau.c:129-132, :220-223)
__device__ __forceinline__ int32_t ff_dist_to_int(float d) { return (d < (float)kMaxNegInt32) ? kMaxNegInt32 : (int32_t)d; }
// the diagonal-Gaussian distance of ptm_mgau.c:102-128: fp32, one rounding per operation, dimensions in order
__device__ __forceinline__ float ff_density(const psgpu_ptm_view_t &pm, const float *x, size_t base, int chain, int cw, int len)
{
    float d = pm.det[(size_t)chain * pm.n_density + cw];
    const float *m = pm.mean + base + (size_t)cw * len, *v = pm.var + base + (size_t)cw * len;
    for (int j = 0; j < len; ++j) {
        const float diff = __fsub_rn(x[j], m[j]);
        d = __fsub_rn(d, __fmul_rn(__fmul_rn(diff, diff), v[j]));
    }
    return d;
}
// A workgroup barrier for phases that exchange data through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier, and
// the fence waits for EVERY outstanding access of the wavefront -- the stores to the slab and the tables that nothing in the next
// phase reads, loads asked for ahead of need: a trip to device memory per barrier, ~27 barriers a frame.  This one waits for the LDS
// counter alone.  Where work-items hand each other data through DEVICE memory (a channel's state after the evaluation, the pruning's
// and the transitions' entries, the active word list) the kernel keeps __syncthreads(): the comments at those barriers say what crosses.
__device__ __forceinline__ void ff_sync_lds()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
// out[t] = max of v over work-items 0 .. t - 1 (-1 for work-item 0).  Ends with a barrier.
__device__ __forceinline__ int32_t ff_block_excl_max(int32_t v, int32_t *tmp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    // (data-parallel-primitive moves, not shuffles -- six vector operations instead of six trips through the LDS crossbar: psgpu_wave_dev.h;
    //  the values are >= -1, FtMax's identity stands for "none")
    const int32_t incl = ft_wave_incl<FtMax>(v);
    if (lane == 63) tmp[tid >> 6] = incl;
    int32_t excl = ft_wave_excl<FtMax>(v);
    if (lane == 0 || excl < -1) excl = -1;
    ff_sync_lds();
    for (int w = 0; w < (tid >> 6); ++w) excl = max(excl, tmp[w]);
    ff_sync_lds();
    return excl;
}

// exclusive prefix sum of one value per work-item; total = the workgroup's sum.  One barrier inside (tmp must not be in use before
// the caller's previous barrier, nor be written again before its next).
__device__ __forceinline__ int32_t ff_block_excl_sum(int32_t v, int32_t *tmp, int32_t &total)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t incl = ft_wave_incl<FtAdd>(v);
    if (lane == 63) tmp[tid >> 6] = incl;
    ff_sync_lds();
    int32_t base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < kFfThreads / 64; ++w) { const int32_t t = tmp[w]; total += t; if (w < (tid >> 6)) base += t; }
    return base + incl - v;
}
// an active word list entry: word, first channel, channels | right-context count << 10 | single phone << 20
__device__ __forceinline__ void ff_awl_put(const FfDev &p, const FfUtt &u, int32_t *awl, int pos, int w)
{
    int len; const int c0 = ff_root(p, u, w, len);
    awl[pos * 3] = w; awl[pos * 3 + 1] = c0; awl[pos * 3 + 2] = len | (u.wrcs[w] << 10) | ((u.wchain[w] < 0 ? 1 : 0) << 20);
}

#if defined(__HIPCC__)
extern __shared__ __attribute__((aligned(16))) unsigned char ff_dyn_lds[];
#endif

template <int NE, bool RAW>
__global__ __launch_bounds__(kFfThreads, 2)            // (two workgroups a compute unit: 256 registers a work-item)
void fwdflat_kernel(FfDev p, const FfOff *__restrict__ offs, FfBufs bf, const int16_t *__restrict__ senscr, int64_t scr_stride,
                    const int32_t *__restrict__ utt_off, FfRaw rw)
{
    __shared__ uint32_t s_bits[RAW ? kFfMaxSen / 32 : 1];
    __shared__ uint8_t s_lcw[RAW ? kFfMaxEnt : 1];                // the scorer's lists: codeword,
    __shared__ int32_t s_lsc[RAW ? kFfMaxEnt : 1];                //   score
    __shared__ uint8_t s_s2cb[RAW ? kFfMaxSen : 1];               // the model's senone -> codebook map
    __shared__ uint8_t s_cbact[RAW ? kFfMaxCb : 1], s_la[RAW ? 512 : 1];  // (log-add table readable up to 511: zero beyond the reference's entries)
    // senones scored evenly over the work-items (the first pass's way, psgpu_sen_dev.h): the frame's lists packed four to a word, the
    // listed senones as a list (prefix sum over the bitmap words' populations)
    __shared__ uint32_t s_pcw[RAW ? kFfMaxEnt / 4 : 1], s_psc[RAW ? kFfMaxEnt / 4 : 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_slist[RAW ? kFfMaxSen : 1];   // the frame's listed senones in the order they were first marked
    // the word transitions' rows (a chunk of the window's words: first phone or -1, base word, best pair's key): the listed senones' place --
    // that list is dead once the frame's senones are scored -- or, when the scores are given, an array of their own
    __shared__ __attribute__((aligned(16))) int32_t s_wt_own[RAW ? 4 : 5 * kFfThreads];
    static_assert(!RAW || kFfMaxSen * 2 >= 20 * kFfThreads, "the rows fit the listed senones' array");
    int32_t *const s_wfirst = RAW ? reinterpret_cast<int32_t *>(s_slist) : s_wt_own, *const s_wbase = s_wfirst + kFfThreads;
    unsigned long long *const s_wkey = reinterpret_cast<unsigned long long *>(s_wfirst + 2 * kFfThreads);
    static_assert(kFfMaxExit <= kFfThreads, "a group of exits per work-item at most");
    // the exits' phase uses the same place: per exiting word (a group of the sorted queue) its best exit's key, the contexts exited into, and
    // whether its exits' histories name different real words
    unsigned long long *const s_gkey = reinterpret_cast<unsigned long long *>(s_wfirst);
    uint32_t *const s_ghave = reinterpret_cast<uint32_t *>(s_wfirst + 2 * kFfThreads);
    int32_t *const s_gdiff = s_wfirst + 4 * kFfThreads;
    __shared__ int32_t s_nl;
    __shared__ int32_t s_norm[16], s_nb;
    __shared__ int32_t s_scan[kFfThreads / 64], s_scan2[kFfThreads / 64];
#ifdef PSGPU_FF_CHECK_LAZY
    __shared__ int32_t s_shadow[RAW ? kFfMaxEnt : 1];
#endif
    __shared__ uint16_t s_openq[RAW ? kFfMaxEnt / 4 : 1];        // touched chains whose entry of this frame is open: scanned a wavefront each
    __shared__ int32_t s_nopen;
    __shared__ int32_t s_lk[RAW ? kFfMaxEnt / 4 : 1];    // per chain: the last frame after which s_lcw holds its list (-1: the seed)
    __shared__ int32_t s_el[kFfMaxEl][4];                        // the active-channel list: channel | flags, word's list position << 10 | chain
                                                                 //   position, word, channels after it | word's right-context count << 10 | single-phone << 20
    __shared__ int32_t s_ex[kFfMaxExit][11], s_nex, s_tot[2];     // the frame's word exits: (word's list position << 10 | chain position), channel,
    __shared__ int32_t s_nbp[kFfNewRows][10];                     // the frame's new back-pointers: word, last / last-but-one phone, score, sorted
                                                                 //   position of the word's first exit, real word ids (two), which right contexts it exited into (64 bits)
    __shared__ FfQuad s_srt[kFfMaxExit + 4];                      // the queue in sorted order, what a walk over a word's exits reads: word's list
                                                                 //   position, score, history, rc slot -- one 16-byte read an exit
    __shared__ uint16_t s_ord[kFfMaxExit];                       //   score, history, right-context count of the word, rc slot, ordinal, stack offset,
                                                                 //   the word's last / last-but-one phone, base word | filler << 30, the history's two real words
    __shared__ int32_t s_fan[kFfMaxFan][4], s_nfan;      // the pruning's queued fan-outs: first target, count, score, history
    __shared__ uint8_t s_tp[kFfMaxTp];                           // the model's transition matrices
#if defined(__HIPCC__)
    int16_t *const s_row = reinterpret_cast<int16_t *>(ff_dyn_lds);           // (scoring mode) the frame's senone scores [n_sen]: dynamic LDS
#else
    __shared__ int16_t s_row[RAW ? kFfMaxSen : 1];                               // (the workgroup simulator)
#endif
    __shared__ int32_t s_sc[8];          // best_score, bpidx, bss_head, status, n_frame done, -, -, length of the evaluation list
    __shared__ unsigned long long s_key;
    const int tid = threadIdx.x;
#ifdef PSGPU_FT_PROFILE
    __shared__ long long s_prof[48], s_last, s_t5;
    __shared__ int s_pscan;
    __shared__ int s_over;
    if (tid == 0) { for (int i = 0; i < 48; ++i) s_prof[i] = 0; s_last = clock64(); s_pscan = 0; }
#endif
    FfUtt u;
    int ub;                               // the utterance of this workgroup
    {
        const FfOff o = offs[blockIdx.x];
        ub = o.utt;
#define X(f) u.f = bf.slab + o.f;
        FF_SLAB_FIELDS(X)
#undef X
#define X(f) u.f = bf.voc + o.f;
        FF_VOC_FIELDS(X)
#undef X
        u.awl[0] = bf.slab + o.awl0; u.awl[1] = bf.slab + o.awl1;
        u.nrow32 = RAW ? bf.slab + o.nrow32 : nullptr;
        u.nrow = RAW ? reinterpret_cast<int16_t *>(bf.slab + o.nrow) : nullptr;
        u.nwd = o.nwd; u.n_chan = o.n_chan; u.n_frame = o.n_frame; u.awl_cap = o.awl_cap;
        u.bp = bf.bp + (size_t)ub * 10 * bf.bp_cap; u.bss = bf.bss + (size_t)ub * bf.bss_cap;
        u.bp_table_idx = bf.idx + (size_t)ub * (bf.max_frames + 2); u.step = bf.step + (size_t)ub * bf.max_frames * 4;
        u.result = bf.res + (size_t)ub * 8;
        u.w1_ssid_in = bf.w1_ssid ? bf.w1_ssid + (size_t)ub * p.n1 * p.n_emit : nullptr;
        u.bp_cap = bf.bp_cap; u.bss_cap = bf.bss_cap;
    }
    const int t0 = utt_off[ub], T = utt_off[ub + 1] - t0;
    int n_awl0 = 0, n_awl1 = 0;           // (two scalars, and the lists through selected pointers below: an array or a struct member indexed by
                                          //  a run-time value sends the whole FfUtt to scratch memory -- 344 bytes a lane, a trip to memory per use)

    // ---- build_fwdflat_chan (:305-368) on the host-made layout, ngram_fwdflat_start (:370-414)
    for (int i = tid; i < p.n1; i += kFfThreads) {
        ff_init(p, u, i, p.w1_mpx[i], p.w1_ssid[i], p.w1_tmat[i], -1);
        if (u.w1_ssid_in && p.w1_mpx[i])        // what the first pass left in the permanent channels (hmm_clear keeps the ssids)
            for (int k = 0; k < p.n_emit; ++k) u.senid[i * 5 + k] = u.w1_ssid_in[i * p.n_emit + k];
    }
    for (int k = tid; k <= u.nwd; k += kFfThreads) u.wseen[k] = -1;
    for (int w = tid; w < p.n_w; w += kFfThreads) { u.wchain[w] = -1; u.wlen[w] = 0; u.wrcs[w] = 0; u.word_active[w] = -1; u.word_lat_idx[w] = -1; }
    __syncthreads();
    for (int k = tid; k < u.nwd; k += kFfThreads) {
        const int w = u.wl_wid[k], c0 = u.wl_chain[k];
        {
            const int w1 = p.w1_of_word[w];
            const int ci2 = c0 >= 0 ? p.pron_ci[p.pron_off[w] + 1] : p.w1_ci2[w1];
            *reinterpret_cast<FfQuad *>(u.wstat + 4 * (size_t)k) = FfQuad{ w, c0 >= 0 ? c0 : w1, p.d_first[w] | (ci2 << 8), p.d_base[w] };
        }
        if (c0 < 0) continue;
        const int len = p.d_pronlen[w], last = p.d_last[w], last2 = p.d_last2[w], nrc = p.rs_n[last * p.n_ci + last2];
        int c = c0;
        u.wchain[w] = c0; u.wlen[w] = u.wl_len[k]; u.wrcs[w] = nrc;
        ff_init(p, u, c++, 1, p.ci_ssid[p.d_first[w]], p.ci_tmat[p.d_first[w]], -1);
        for (int q = 1; q < len - 1; ++q) {
            const int o = p.pron_off[w] + q;
            ff_init(p, u, c++, 0, p.pron_ssid[o], p.ci_tmat[p.pron_ci[o]], -1);
        }
        for (int r = 0; r < nrc; ++r)
            ff_init(p, u, c++, 0, p.rs_ssid[((size_t)last * p.n_ci + last2) * p.n_ci + r], p.ci_tmat[last], r);
    }
    const bool tp_lds = p.tp_bytes <= kFfMaxTp;
    if (tp_lds) for (int i = tid; i < p.tp_bytes; i += kFfThreads) s_tp[i] = p.tp[i];
    if (tid == 0) {
        s_sc[0] = 0; s_sc[1] = 0; s_sc[2] = 0; s_sc[3] = 0; s_sc[4] = 0;
        ff_enter(u, p.w1_of_word[p.startwid], 0, -1, 0);
        ff_awl_put(p, u, u.awl[0], 0, p.startwid);
    }
    n_awl0 = 1;
    const int n_chain = RAW ? rw.pm.n_mgau * rw.pm.n_feat : 0, topn = RAW ? rw.pm.topn : 0;
    if (RAW) {
        for (int i = tid; i < n_chain * topn; i += kFfThreads) { s_lcw[i] = rw.seed[(size_t)ub * n_chain * topn + i]; s_lsc[i] = 0; }
        for (int i = tid; i < n_chain && i < kFfMaxEnt / 4; i += kFfThreads) s_lk[i] = -1;
#ifdef PSGPU_FF_CHECK_LAZY
        for (int i = tid; i < n_chain * topn; i += kFfThreads) s_shadow[i] = s_lcw[i];
#endif
        for (int i = tid; i < 512; i += kFfThreads) s_la[i] = (i < rw.pm.logadd8_size && i < 256) ? rw.pm.logadd8[i] : 0;
        for (int i = tid; i < rw.pm.n_sen; i += kFfThreads) s_s2cb[i] = rw.pm.sen2cb[i];
    }
    __syncthreads();

    // the next active word list's candidates, candidate tid + 256 j in slot j (vocabularies up to kFfRegRows x 256 words): static
    const int n_tail = p.n_w - p.startwid, n_all = u.nwd + n_tail;
    int wq[kFfRegRows] = {}, c0q[kFfRegRows] = {}, axq[kFfRegRows] = {};
    if (n_all <= FF_AWL_REGS) {
#pragma unroll
        for (int j = 0; j < kFfRegRows; ++j) {
            const int i = tid + j * kFfThreads;
            const int w = i < u.nwd ? u.wl_wid[i] : (i < n_all ? p.startwid + (i - u.nwd) : p.startwid);
            const int32_t ch = u.wchain[w], ln = u.wlen[w], rcs = u.wrcs[w], w1 = p.w1_of_word[w];
            wq[j] = w; c0q[j] = ch >= 0 ? ch : w1;
            axq[j] = (ch >= 0 ? ln : 1) | (rcs << 10) | ((ch < 0 ? 1 : 0) << 20);
        }
    }
    // the active word list as the frame before wrote it, for the gather at the top of the frame: in the place of the new entries' rows, dead
    // between the word transitions and the next frame's exits
    static_assert(kFfNewRows * 10 >= 3 * kFfAwlLds, "the active word list's LDS copy fits the new entries' rows");
    int32_t *const s_awl = &s_nbp[0][0];
    bool awl_lds = false;
    FfQuad pre_q = { 0, 0, 0, 0 };
    uint32_t pre_c4 = 0;
    uint8_t pre_open = 0;
    const bool ahead = RAW && rw.tsc && topn == 4 && n_chain <= kFfThreads;
    int sl_b0 = 0, sl_b1 = 0;                 // the window's slice of the nodes-by-start-frame list, of the frame about to begin
    auto slice_bounds = [&](int fr) {
        int sf0 = fr - p.max_sf_win, ef0 = fr + p.max_sf_win;
        if (sf0 < 0) sf0 = 0;
        if (ef0 > u.n_frame) ef0 = u.n_frame;
        sl_b0 = 0; sl_b1 = 0;
        // (through the vector path -- the index is the same in every work-item, which the compiler cannot see: a scalar load would make
        //  the next LDS wait, which shares its counter, a trip to memory)
        if (ef0 > sf0) { sl_b0 = u.fr_off[sf0 + (tid >> 12)]; sl_b1 = u.fr_off[ef0 + (tid >> 12)]; }
    };
    slice_bounds(0);
    for (int f = 0; f < T; ++f) {
        const int cur = f & 1, nxt = cur ^ 1, nf = f + 1, na = cur ? n_awl1 : n_awl0;
        int32_t *const awl_c = cur ? u.awl[1] : u.awl[0], *const awl_n = cur ? u.awl[0] : u.awl[1];
        const int16_t *const row_dev = RAW ? nullptr : senscr + (size_t)(t0 + f) * scr_stride;
        // ---- the frame's active channels, gathered into one list first (one work-item per word walks its chain once: order
        //      irrelevant): the senone marking below and fwdflat_eval_chan both go over it one work-item per CHANNEL -- the marking
        //      used to walk every active word's chain a second time, one work-item per word (11 % of the frame,
        //      profiles/r03_fwdflat_phase_profile.txt).  Nothing between here and the evaluation changes a channel's frame stamp.
        FF_PROFS(40);
        if (tid == 0) { s_sc[7] = 0; s_nfan = 0; s_nex = 0; s_tot[0] = 0; s_tot[1] = 0; s_nl = 0; s_nb = 0x7fffffff; s_nopen = 0; }
        FF_PROFS(41);
        // the word transitions' successors (below): the window's slice of the nodes-by-start-frame list depends on f alone -- its bounds were
        // asked for a frame ahead, its words are after the next barrier, their static quads after the one after that: nothing waits for them
        const int sl_q0 = sl_b0, sl_n = sl_b1 - sl_b0;      // (this frame's bounds were asked for during the frame before)
#if defined(PSGPU_FT_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "s"(sl_n));
#endif
        FF_PROFS(42);
        if (RAW) {
            for (int i = tid; i < (rw.pm.n_sen + 31) >> 5; i += kFfThreads) s_bits[i] = 0u;
            for (int i = tid; i < rw.pm.n_mgau; i += kFfThreads) s_cbact[i] = 0;
            if (tid < 16) s_norm[tid] = kW;
        }
        FF_PROFS(43);
        ff_sync_lds();
        FF_PROFS(32);
        for (int i = tid >> 4; i < na; i += kFfThreads / 16) {          // sixteen work-items a word: its chain's stamps read side by side
            int w, c0, wx;
            if (awl_lds) { w = s_awl[i * 3]; c0 = s_awl[i * 3 + 1]; wx = s_awl[i * 3 + 2]; }
            else { w = awl_c[i * 3]; c0 = awl_c[i * 3 + 1]; wx = awl_c[i * 3 + 2]; }
            const int len = wx & 1023;
#ifdef PSGPU_FT_PROFILE
            if ((tid & 15) == 0) atomicAdd(&s_pscan, len);
            if (i == 0) FF_PROFS(33);
#endif
            for (int k = tid & 15; k < len; k += 16) {
                const int c = c0 + k;
                // (the stamp and what an active channel needs next asked for together: one trip to memory instead of two)
                const int32_t stamp = u.frame[c];
                int32_t sid[NE];
                bool mpx = false;
                if (RAW) {
                    mpx = u.mpx[c] != 0;
#pragma unroll
                    for (int q = 0; q < NE; ++q) sid[q] = u.senid[c * 5 + q];
                }
#ifdef PSGPU_FT_PROFILE
                if (i == 0 && k == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                    { int m_ = mpx, s0_ = RAW ? sid[0] : 0; asm volatile("" :: "v"(stamp), "v"(s0_), "v"(m_)); }
#endif
                    FF_PROFS(34);
                }
#endif
                if (stamp == f) {                             // bit 30: the root of </s>, which does not count towards the best score
                    const int pos = atomicAdd(&s_sc[7], 1);
                    const int cf = c | ((k == 0 && w == p.finishwid) ? (1 << 30) : 0);
                    if (pos < FF_EL_CAP) {
                        int32_t *x = s_el[pos];
                        x[0] = cf; x[1] = (i << 10) | k; x[2] = w; x[3] = (len - k - 1) | (wx & ~1023);
                    }
                    else { u.elist[pos] = cf; u.einfo[pos] = (i << 10) | k; }
                    if (RAW) {
                        // compute_fwdflat_sen_active (:416-442): the channel's senones into the frame's bitmap, and -- the first time
                        // a senone is marked -- into the list the evaluation below goes over (its order does not matter there)
                        if (mpx) {                            // (a multiplexed channel's states through their senone sequences: asked for together)
#pragma unroll
                            for (int q = 0; q < NE; ++q) sid[q] = sid[q] == kBadSsid ? -1 : (int32_t)p.sseq[(size_t)sid[q] * NE + q];
                        }
#pragma unroll
                        for (int q = 0; q < NE; ++q) {
                            const int sen = sid[q];
                            if (sen < 0) continue;
                            const uint32_t bit = 1u << (sen & 31);
                            if (!(atomicOr(&s_bits[sen >> 5], bit) & bit)) s_slist[atomicAdd(&s_nl, 1)] = (uint16_t)sen;
                        }
                    }
                }
            }
        }
        FF_PROFS(35);
        ff_sync_lds();
#ifdef PSGPU_FT_PROFILE
        if (tid == 0) { s_prof[27] += na; s_prof[28] += s_pscan; s_prof[29] += s_sc[7]; s_pscan = 0; }
#endif
        FF_PROF(8);
        const int sl_k = tid < sl_n && tid < FF_SL_CHUNK ? u.fr_words[sl_q0 + tid] : -1;
        slice_bounds(nf);
        // the batch scorer's entry of this frame for the work-item's chain (`ahead`): asked for here, behind the gather's last load (loads
        // come back in order: asked for at the top of the frame, these lines -- read once, no cache has them -- made the gather's first
        // wait a trip to HBM) and ahead of a phase that asks device memory nothing
        if (ahead && tid < n_chain) {
            const size_t o = (size_t)tid * rw.total + t0 + f;
            pre_q = *reinterpret_cast<const FfQuad *>(rw.tsc + o * 4); pre_c4 = rw.tcw[o]; pre_open = rw.open[o];
        }

        const int n_eval = s_sc[7];
        struct FfEnt { int32_t c, inf, w, aux; };
        auto ent = [&](int e) -> FfEnt {
            if (e < FF_EL_CAP) { const int32_t *x = s_el[e]; return { x[0], x[1], x[2], x[3] }; }
            const int32_t inf = u.einfo[e], w = awl_c[(inf >> 10) * 3], wx = awl_c[(inf >> 10) * 3 + 2];
            return { u.elist[e], inf, w, ((wx & 1023) - (inf & 1023) - 1) | (wx & ~1023) };
        };
        if (RAW) {
            const psgpu_ptm_view_t &pm = rw.pm;
            const float *x = rw.feats + (size_t)(t0 + f) * pm.veclen;
            const int nwords = (pm.n_sen + 31) >> 5;
            const int n_l0 = s_nl;
            {
                static_assert(kFfMaxSen / 32 <= kFfThreads, "one bitmap word per work-item");
                const uint32_t bw = tid < nwords ? s_bits[tid] : 0u;
                // s_prev[w] = the highest senone listed in the words before w: where acmod_flags2list's bridging entries go
                const int32_t pv = ff_block_excl_max(bw ? tid * 32 + 31 - __clz((int)bw) : -1, s_scan);
                // every listed senone (bridging entries included) touches its codebook: ptm_mgau_calc_cb_active (:297-321).  Only the
                // first senone of a bitmap word can be more than 255 past its predecessor; the entries in between are listed too
                if (bw) {
                    const int sen = tid * 32 + __ffs((int)bw) - 1;
                    for (int last = pv < 0 ? 0 : pv; sen - last > 255;) {
                        last += 255;
                        s_cbact[s_s2cb[last]] = 1;
                        s_slist[atomicAdd(&s_nl, 1)] = (uint16_t)last;
                    }
                }
            }
            for (int i = tid; i < n_l0; i += kFfThreads) s_cbact[s_s2cb[s_slist[i]]] = 1;
            ff_sync_lds();
            FF_PROF(0);
            // ---- eval_topn for every chain, eval_cb for the touched codebooks' chains; one work-item per chain.
            //      With the batch scorer's lists at hand (`lazy`) a chain costs nothing in most frames: a touched codebook's list is
            //      the batch scorer's where that entry is closed (whatever was carried in), and an UNTOUCHED codebook's eval_topn only
            //      re-scores and re-orders the carried codewords -- state that nothing reads until the chain is next scanned with
            //      an open entry (ties: rare).  Then the re-orderings since the chain's last known list (s_lk) are replayed: a frame
            //      whose four re-scored values are pairwise distinct leaves them in descending order whatever the order before, so
            //      the replay starts at the latest such frame (almost always the one before this).
            const bool lazy = rw.tsc && topn == 4;
            for (int ch = tid; ch < n_chain; ch += kFfThreads) {
                const int cb = ch / pm.n_feat, fs = ch % pm.n_feat, len = pm.featlen[fs];
                const size_t base = (size_t)cb * pm.n_density * pm.veclen + (size_t)pm.n_density * pm.featoff[fs];
                const float *xs = x + pm.featoff[fs];
                int32_t cw[kFfMaxTopn], sc[kFfMaxTopn];
#ifdef PSGPU_FF_CHECK_LAZY
                auto shadow_rescore = [&]() {                    // eval_topn's re-ordering of the reference's carried list
                    int32_t c4[kFfMaxTopn], s4[kFfMaxTopn];
                    for (int i = 0; i < topn; ++i) {
                        const int c = s_shadow[ch * topn + i];
                        const int32_t v = ff_dist_to_int(ff_density(pm, xs, base, ch, c, len));
                        int j = i;
                        for (; j > 0 && v > s4[j - 1]; --j) { s4[j] = s4[j - 1]; c4[j] = c4[j - 1]; }
                        s4[j] = v; c4[j] = c;
                    }
                    for (int i = 0; i < topn; ++i) s_shadow[ch * topn + i] = c4[i];
                };
                if (lazy && !s_cbact[cb]) shadow_rescore();
#endif
                // the batch scorer's entry of this chain and frame: asked for at the top of the frame (`ahead`: one chain per work-item,
                // its registers hold it), where the trip to memory -- these lines are read once, no cache has them -- runs beside the
                // active list's instead of after it
                FfQuad q = pre_q;
                uint32_t c4 = pre_c4;
                bool closed = ahead && !pre_open;
                if (lazy && !ahead) {
                    const size_t o = (size_t)ch * rw.total + t0 + f;
                    q = *reinterpret_cast<const FfQuad *>(rw.tsc + o * 4); c4 = rw.tcw[o]; closed = !rw.open[o];
                }
                if (lazy && !s_cbact[cb]) continue;
                if (closed) {
                    s_lsc[ch * 4] = q.x; s_lsc[ch * 4 + 1] = q.y; s_lsc[ch * 4 + 2] = q.z; s_lsc[ch * 4 + 3] = q.w;
                    s_lcw[ch * 4] = c4 & 0xff; s_lcw[ch * 4 + 1] = (c4 >> 8) & 0xff; s_lcw[ch * 4 + 2] = (c4 >> 16) & 0xff; s_lcw[ch * 4 + 3] = c4 >> 24;
                    s_lk[ch] = f;
#ifdef PSGPU_FF_CHECK_LAZY
                    for (int i = 0; i < 4; ++i) s_shadow[ch * 4 + i] = s_lcw[ch * 4 + i];
#endif
                    atomicMax(&s_norm[fs], q.x >> 10);       // ptm_mgau_codebook_norm (:272-279)
                    continue;
                }
                if (lazy && pm.n_density == 128) { s_openq[atomicAdd(&s_nopen, 1)] = (uint16_t)ch; continue; }     // (a wavefront's job, below)
                for (int i = 0; i < topn; ++i) cw[i] = s_lcw[ch * topn + i];
                auto rescore = [&](const float *xg) {            // re-score, stable descending insertion with strict '>' (:71-85)
                    for (int i = 0; i < topn; ++i) {
                        const int c = cw[i];
                        const int32_t v = ff_dist_to_int(ff_density(pm, xg, base, ch, c, len));
                        int j = i;
                        for (; j > 0 && v > sc[j - 1]; --j) { sc[j] = sc[j - 1]; cw[j] = cw[j - 1]; }
                        sc[j] = v; cw[j] = c;
                    }
                };
                if (lazy) {
                    int g = f - 1;
                    for (; g > s_lk[ch]; --g) {
                        rescore(rw.feats + (size_t)(t0 + g) * pm.veclen + pm.featoff[fs]);
                        if (sc[0] != sc[1] && sc[1] != sc[2] && sc[2] != sc[3]) break;     // (sorted: neighbours suffice)
                    }
                    if (g <= s_lk[ch]) {                         // no such frame: every re-ordering since the known list, in order
                        for (int i = 0; i < topn; ++i) cw[i] = s_lcw[ch * topn + i];
                        g = s_lk[ch];
                    }
                    for (++g; g < f; ++g) rescore(rw.feats + (size_t)(t0 + g) * pm.veclen + pm.featoff[fs]);
                    s_lk[ch] = f;
#ifdef PSGPU_FF_CHECK_LAZY
                    for (int i = 0; i < topn; ++i)
                        if (cw[i] != s_shadow[ch * topn + i]) {
                            printf("fwdflat: chain %d frame %d: replayed list entry %d is codeword %d, the reference carries %d\n", ch, f, i, cw[i], s_shadow[ch * topn + i]);
                            abort();
                        }
#endif
                }
                rescore(xs);
                if (s_cbact[cb])
                    for (int c = 0; c < pm.n_density; ++c) {     // codewords in index order against the moving threshold (:151-226)
                        const float th = (float)sc[topn - 1];
                        const float d = ff_density(pm, xs, base, ch, c, len);
                        if (d < th) continue;
                        bool in = false;
                        for (int i = 0; i < topn; ++i) in |= cw[i] == c;
                        if (in) continue;
                        const int32_t v = ff_dist_to_int(d);
                        int q = topn - 1;                        // ahead of equal scores, the old worst drops (:140-149)
                        for (; q > 0 && v >= sc[q - 1]; --q) { sc[q] = sc[q - 1]; cw[q] = cw[q - 1]; }
                        sc[q] = v; cw[q] = c;
                    }
                for (int i = 0; i < topn; ++i) { s_lcw[ch * topn + i] = cw[i]; s_lsc[ch * topn + i] = sc[i]; }
#ifdef PSGPU_FF_CHECK_LAZY
                for (int i = 0; i < topn; ++i) s_shadow[ch * topn + i] = cw[i];
#endif
                if (s_cbact[cb]) atomicMax(&s_norm[fs], sc[0] >> 10);        // ptm_mgau_codebook_norm (:272-279)
            }
            ff_sync_lds();
            if (s_nopen) {
                // An open entry of a touched codebook: the reference's own procedure on the list it would carry here -- the re-orderings
                // since the chain's last known list replayed (see above), then eval_topn + eval_cb of this frame (ptm_mgau.c:71-226) -- by
                // one WAVEFRONT: the chain's 128 densities two a work-item, the list wave-uniform, the scan in codeword order by ballots
                // (psgpu_ptm_dev.h exact_frame_step's way).  In one work-item this was 5 k cycles of the average frame.
                const int lane = tid & 63;
                for (int qi = tid >> 6; qi < s_nopen; qi += kFfThreads / 64) {
                    const int ch = s_openq[qi], cb = ch / pm.n_feat, fs = ch % pm.n_feat, len = pm.featlen[fs];
                    const size_t base = (size_t)cb * pm.n_density * pm.veclen + (size_t)pm.n_density * pm.featoff[fs];
                    int32_t cw[4], sc[4];
                    float d0 = 0.0f, d1 = 0.0f;
                    auto dens = [&](int g) {
                        const float *xg = rw.feats + (size_t)(t0 + g) * pm.veclen + pm.featoff[fs];
                        d0 = ff_density(pm, xg, base, ch, lane, len); d1 = ff_density(pm, xg, base, ch, lane + 64, len);
                    };
                    auto value = [&](int c) { return c < 64 ? __shfl(d0, c) : __shfl(d1, c - 64); };
                    auto rescore = [&]() {                       // (:71-85)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = cw[i];
                            const int32_t v = ff_dist_to_int(value(c));
                            sc[i] = v; cw[i] = c;
#pragma unroll
                            for (int j = i; j > 0; --j)
                                if (sc[j] > sc[j - 1]) {
                                    const int32_t ts = sc[j]; sc[j] = sc[j - 1]; sc[j - 1] = ts;
                                    const int32_t tc = cw[j]; cw[j] = cw[j - 1]; cw[j - 1] = tc;
                                }
                        }
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i) cw[i] = s_lcw[ch * 4 + i];
                    const int lk = s_lk[ch];
                    int g = f - 1;
                    for (; g > lk; --g) {
                        dens(g); rescore();
                        if (sc[0] != sc[1] && sc[1] != sc[2] && sc[2] != sc[3]) break;
                    }
                    if (g <= lk) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) cw[i] = s_lcw[ch * 4 + i];
                        g = lk;
                    }
                    for (++g; g < f; ++g) { dens(g); rescore(); }
#ifdef PSGPU_FF_CHECK_LAZY
                    for (int i = 0; i < 4; ++i)
                        if (cw[i] != s_shadow[ch * 4 + i]) {
                            fprintf(stderr, "fwdflat: chain %d frame %d: replayed list entry %d is codeword %d, the reference carries %d\n", ch, f, i, cw[i], s_shadow[ch * 4 + i]);
                            abort();
                        }
#endif
                    dens(f); rescore();
                    for (int pos = 0;;) {                        // (:151-226) codewords from `pos` on in index order against the moving threshold
                        const float th = (float)sc[3];
                        bool in0 = false, in1 = false;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { in0 |= cw[i] == lane; in1 |= cw[i] == lane + 64; }
                        unsigned long long b0 = __ballot(d0 >= th && !in0), b1 = __ballot(d1 >= th && !in1);
                        if (pos >= 64) { b0 = 0; b1 = pos >= 128 ? 0ull : (b1 & (~0ull << (pos - 64))); }
                        else b0 &= ~0ull << pos;
                        if ((b0 | b1) == 0) break;
                        const int c = b0 ? __ffsll(b0) - 1 : 64 + __ffsll(b1) - 1;
                        const int32_t v = ff_dist_to_int(value(c));
                        int q = 3;                               // ahead of equal scores, the old worst drops (:140-149)
#pragma unroll
                        for (int k = 3; k > 0; --k)
                            if (q == k && v >= sc[k - 1]) { sc[k] = sc[k - 1]; cw[k] = cw[k - 1]; q = k - 1; }
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (q == k) { sc[k] = v; cw[k] = c; }
                        pos = c + 1;
                    }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { s_lcw[ch * 4 + i] = cw[i]; s_lsc[ch * 4 + i] = sc[i]; }
#ifdef PSGPU_FF_CHECK_LAZY
                        for (int i = 0; i < 4; ++i) s_shadow[ch * 4 + i] = cw[i];
#endif
                        s_lk[ch] = f;
                        atomicMax(&s_norm[fs], sc[0] >> 10);
                    }
                }
                ff_sync_lds();
            }
            FF_PROF(9);
            // The scorer's usual shape (3 streams, top-4, senone-major weights at hand): each listed senone's twelve weights from three
            // cache lines (sen_eval_f3n4, the first pass's), the lists packed four to a word
            const bool fast = topn == 4 && pm.n_feat == kSenStreams && pm.mixw_sen != nullptr && n_chain <= kFfMaxEnt / 4;
            for (int ch = tid; ch < n_chain; ch += kFfThreads) {              // (:280-291), touched codebooks only
                if (!s_cbact[ch / pm.n_feat]) continue;
                const int32_t nm = s_norm[ch % pm.n_feat];
                uint32_t pc = 0, ps = 0;
                for (int k = 0; k < topn; ++k) {
                    int32_t v = nm - (s_lsc[ch * topn + k] >> 10);
                    v = v > kMaxNegAscr ? kMaxNegAscr : v;
                    s_lsc[ch * topn + k] = v;
                    if (k < 4) { pc |= (uint32_t)(s_lcw[ch * topn + k] & 0xff) << (8 * k); ps |= (uint32_t)(v & 0xff) << (8 * k); }
                }
                if (fast) { s_pcw[ch] = pc; s_psc[ch] = ps; }
            }
            ff_sync_lds();
            FF_PROF(1);
            // ---- ptm_mgau_senone_eval (:326-403) for the listed senones; the frame's scores are those minus their minimum
            const int n_l = s_nl;
            int32_t mn = 0x7fffffff;
            auto senone = [&](int sen) {
                const int cb = s_s2cb[sen];
                int32_t a = 0;
                for (int fs = 0; fs < pm.n_feat; ++fs) {
                    const int li = (cb * pm.n_feat + fs) * topn;
                    const uint8_t *mw = pm.mixw + (size_t)fs * pm.n_density * pm.n_sen + sen;
                    int32_t fden = (int32_t)mw[(size_t)s_lcw[li] * pm.n_sen] + s_lsc[li];
                    for (int k = 1; k < topn; ++k) {             // fast_logmath_add (tied_mgau_common.h:106-125)
                        const int32_t y = (int32_t)mw[(size_t)s_lcw[li + k] * pm.n_sen] + s_lsc[li + k];
                        const int32_t lo = min(fden, y);
                        const uint32_t dd = (uint32_t)(max(fden, y) - lo);
                        fden = lo - (dd < 256u ? (int32_t)s_la[dd] : 0);
                    }
                    a += fden;
                }
                return a;
            };
            auto wg_min = [&](int32_t v) {                       // the list's minimum to s_nb
                v = ft_wave_incl<FtMin>(v);
                if ((tid & 63) == 63) atomicMin(&s_nb, v);
            };
            if (n_l <= 4 * kFfThreads) {                         // (every frame in practice: a senone's value waits in a register)
                const SenModel smod = { pm.mixw_sen, pm.sen2cb, pm.n_sen, pm.n_density };
                int32_t av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kFfThreads;
                    av[j] = 0x7fffffff;
                    if (i < n_l) av[j] = fast ? sen_eval_f3n4_cb(smod, s_pcw, s_psc, s_la, (int)s_slist[i], s_s2cb[s_slist[i]]) : senone((int)s_slist[i]);
                    mn = min(mn, av[j]);
                }
                wg_min(mn);
                ff_sync_lds();
                const int32_t nb = s_nb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + j * kFfThreads;
                    if (i < n_l) s_row[s_slist[i]] = (int16_t)(uint16_t)((uint32_t)(int32_t)(int16_t)av[j] - (uint32_t)nb);
                }
            }
            else {
                const SenModel smod = { pm.mixw_sen, pm.sen2cb, pm.n_sen, pm.n_density };
                for (int i = tid; i < n_l; i += kFfThreads) {
                    const int sen = s_slist[i];
                    const int32_t a = fast ? sen_eval_f3n4_cb(smod, s_pcw, s_psc, s_la, sen, s_s2cb[sen]) : senone(sen);
                    u.nrow32[sen] = a;
                    mn = min(mn, a);
                }
                wg_min(mn);
                ff_sync_lds();
                const int32_t nb = s_nb;
                for (int i = tid; i < n_l; i += kFfThreads) {
                    const int sen = s_slist[i];
                    s_row[sen] = (int16_t)(uint16_t)((uint32_t)(int32_t)(int16_t)u.nrow32[sen] - (uint32_t)nb);
                }
            }
            ff_sync_lds();
        }
        FF_PROF(2);
        // ---- ngram_search_mark_bptable, failure test, renormalisation (:825-838)
        const int32_t bp_first = s_sc[1];                    // (the frame's first back-pointer: written for the host, kept for the word transitions)
        if (tid == 0) u.bp_table_idx[f] = bp_first;
        const int32_t best_in = s_sc[0];
        if (best_in == kW || best_in < kW) break;
        if (best_in + 2 * p.beam < kW)                       // fwdflat_renormalize_scores (:784-810)
            for (int i = tid; i < na; i += kFfThreads) {
                const int c0 = awl_c[i * 3 + 1], len = awl_c[i * 3 + 2] & 1023;
                for (int k = 0; k < len; ++k) if (u.frame[c0 + k] == f) ff_normalize(p, u, c0 + k, best_in);
            }
        // (a full barrier where work-items exchanged through device memory since the last one: renormalised channels, the part of the
        //  active-channel list that did not fit LDS)
        if (best_in + 2 * p.beam < kW || n_eval > FF_EL_CAP) __syncthreads(); else ff_sync_lds();
        FfQuad sl_w = { 0, 0, 0, 0 };
        bool sl_on = false;
        if (sl_k >= 0) {                                      // (a word with two nodes in the window is taken once)
            sl_on = atomicExch(&u.wseen[sl_k], nf) != nf;
            sl_w = *reinterpret_cast<const FfQuad *>(u.wstat + 4 * (size_t)sl_k);
        }
        if (tid == 0) { s_sc[0] = kW; s_sc[5] = kW; s_sc[6] = 0; s_key = 0ull; }
        ff_sync_lds();
        FF_PROF(3);
        int32_t k_best = kW, k_out = kW, k_outh = -1, k_rc = -1, k_rc1 = -1, k_nfr = 0, k_s0 = kW;     // the first entry a work-item evaluates
        // ---- fwdflat_eval_chan (:444-480), one work-item per channel of the list made at the top of the frame (a word near its end
        //      has its whole right-context fan-out, 20-40 channels, active at once)
        {
            int32_t b = kW;
            // (one instantiation per pairing of score row and transition matrices: LDS / device memory -- the address space is the
            // compiler's to infer from the argument)
            auto eval_all = [&](const int16_t *row, const uint8_t *tp_all) __attribute__((always_inline)) {
            for (int i = tid; i < n_eval; i += kFfThreads) {
                const int e = i < FF_EL_CAP ? s_el[i][0] : u.elist[i];
                const int c = e & kFfChanMask;
                if (i < kFfThreads && i < FF_EL_CAP) {       // (what the pruning of this entry reads besides: asked for with the state)
                    const int rem = s_el[i][3] & 1023;
                    k_rc = u.rcid[c]; k_rc1 = rem > 0 ? u.rcid[c + 1] : -1;
                    k_nfr = rem > 0 ? u.frame[c + 1] : 0;   // (not its score[0]: the successor's own evaluation is changing it)
                }
                int32_t o_s, o_h, o_0;
                const int32_t sc = ff_eval<NE>(p, u, c, row, tp_all, o_s, o_h, o_0);
                if (i < kFfThreads) { k_best = sc; k_out = o_s; k_outh = o_h; k_s0 = o_0; }
                if (!(e & (1 << 30))) b = max(b, sc);
            }
            };
            if (RAW) { if (tp_lds) eval_all(s_row, s_tp); else eval_all(s_row, p.tp); }
            else { if (tp_lds) eval_all(row_dev, s_tp); else eval_all(row_dev, p.tp); }
            if (b > kW) atomicMax(&s_sc[0], b);
        }
        __syncthreads();
        const int32_t best_score = s_sc[0];
        const int32_t thresh = best_score + p.fwdflatbeam, wordthresh = best_score + p.fwdflatwbeam;
        FF_PROF(4);
        // ---- fwdflat_prune_chan (:482-607), one work-item per ACTIVE CHANNEL (the list made at the top of the frame).  The reference
        //      walks a word's chain front to back; what a channel decides depends on its own evaluation alone (best / out score),
        //      and its entry state (score[0], history[0]) is written by its one predecessor in the chain only, so the channels of
        //      a chain can be decided side by side: (1) retain / exit / hand on to the successor(s) -- the fan-out of a word's
        //      last-but-one phone into its right-context channels is queued and (2) spread over a wavefront's work-items;
        //      (3) a channel that neither survived nor was entered is cleared (the walk's "else if frame != nf").  A channel that
        //      was not active is never looked at: it was cleared when it left the list, its best score is WORST_SCORE.
        for (int e = tid; e < n_eval; e += kFfThreads) {
            const FfEnt en = ent(e);
            const int c = en.c & kFfChanMask, k = en.inf & 1023, rem = en.aux & 1023, w = en.w;
            // (the successor's entry state with the rest: its stamp is f or f + 1 whoever writes it meanwhile, its score[0] is written by
            // this work-item alone; the entry a work-item evaluated first is still in its registers)
            const bool kept = e < kFfThreads && e < FF_EL_CAP;
            const int32_t best = kept ? k_best : u.best[c], out = kept ? k_out : u.out[c], hist = kept ? k_outh : u.outh[c];
            const int32_t rc = kept ? k_rc : u.rcid[c], rc1 = kept ? k_rc1 : (rem > 0 ? u.rcid[c + 1] : -1);
            const int32_t nfr = kept ? k_nfr : (rem > 0 ? u.frame[c + 1] : 0), nsc = rem > 0 ? u.score[(c + 1) * 5] : 0;
            if (best > thresh) {
                int32_t newscore = out;
                u.frame[c] = nf; u.word_active[w] = nf;
                if (k == 0 ? rem > 0 : rc < 0) {
                    newscore += p.pip;
                    if (newscore > thresh) {
                        if (rc1 >= 0 && rem > 1) {
                            const int q = atomicAdd(&s_nfan, 1);
                            if (q < kFfMaxFan) { s_fan[q][0] = c + 1; s_fan[q][1] = rem; s_fan[q][2] = newscore; s_fan[q][3] = hist; }
                            else for (int j = 1; j <= rem; ++j) ff_enter_if_better(u, c + j, newscore, hist, f);
                        }
                        else if (nfr < f || newscore > nsc) ff_enter(u, c + 1, newscore, hist, nf);
                    }
                }
                else if (newscore > wordthresh) {            // a word exit: queued, see below
                    const int q = atomicAdd(&s_nex, 1);
                    if (q < FF_EXIT_CAP) {
                        int32_t *x = s_ex[q];
                        x[0] = en.inf; x[1] = w | ((en.aux >> 20) << 30); x[2] = newscore; x[3] = hist;
                        x[4] = (en.aux >> 10) & 1023;
                        x[5] = k == 0 ? 0 : rc;
                        x[6] = p.d_last[w]; x[7] = (en.aux >> 20) ? -1 : p.d_last2[w]; x[8] = p.d_base[w] | (p.d_filler[w] ? (1 << 30) : 0);
                        x[9] = hist != -1 ? FBP(u, F_REAL, hist) : -1; x[10] = hist != -1 ? FBP(u, F_PREAL, hist) : -1;
                    }
                }
            }
            else if (k > 0) {
                // the walk's "else if (frame != nf) clear": did its predecessor in the chain enter it?  For the entry a work-item
                // evaluated itself (its state-0 score as the evaluation left it is in a register) the predecessor's decision is taken
                // here a second time from the same values -- no look at the stamp after a barrier; for the others the stamp decides
                int v = en.c | kFfClearBit | kFfEnteredBit;                     // (both: look at the stamp)
                if (kept) {
                    const int rcs = (en.aux >> 10) & 1023;
                    const int pred = rc < 0 ? c - 1 : c - k + (k + rem - rcs);
                    const int32_t pb = u.best[pred], po = u.out[pred] + p.pip;
                    const bool entered = pb > thresh && pb > kW && po > thresh && po > k_s0;
                    v = en.c | (entered ? kFfEnteredBit : kFfClearBit);
                }
                if (e < FF_EL_CAP) s_el[e][0] = v; else u.elist[e] = v;
            }
        }
        __syncthreads();
        FF_PROF(10);
        for (int q = tid >> 6, nq = min(s_nfan, kFfMaxFan); q < nq; q += kFfThreads / 64)
            for (int j = tid & 63; j < s_fan[q][1]; j += 64) ff_enter_if_better(u, s_fan[q][0] + j, s_fan[q][2], s_fan[q][3], f);
        // (the clears below look at a stamp -- which a fan-out or a predecessor may just have written -- only for entries beyond a
        //  work-item's first; the others decide from registers)
        if (n_eval > kFfThreads || n_eval > FF_EL_CAP) __syncthreads(); else ff_sync_lds();
        for (int e = tid; e < n_eval; e += kFfThreads) {
            const int v = e < FF_EL_CAP ? s_el[e][0] : u.elist[e];
            const int both = kFfClearBit | kFfEnteredBit;
            if ((v & both) == both) { if (u.frame[v & kFfChanMask] != nf) ff_clear_scores(p, u, v & kFfChanMask); continue; }
#ifdef PSGPU_FF_CHECK_LAZY
            if ((v & both) && ((v & kFfEnteredBit) != 0) != (u.frame[v & kFfChanMask] == nf)) {
                fprintf(stderr, "fwdflat: frame %d channel %d: the predecessor's decision taken twice differs (%x, stamp %d)\n", f, v & kFfChanMask, (unsigned)v, u.frame[v & kFfChanMask]);
                abort();
            }
#endif
            if (v & kFfClearBit) ff_clear_scores(p, u, v & kFfChanMask);
        }
        ff_sync_lds();
        FF_PROF(11);
        // ---- the exits' back-pointers (ngram_search_save_bp as fwdflat_prune_chan calls it, :528-540, :588-600): one entry per
        //      exiting WORD in active-list order, its right-context exits applied in chain order (the first creates the entry, a later
        //      one with a strictly better score takes it over, each leaves its score in its slot of the word's stack block).  The
        //      queue is sorted by (word's list position, chain position) by counting; a word's first exit learns how many words
        //      and stack entries precede it from the sorted queue and then walks its group -- everything but the table itself in LDS.
        const int n_exq = s_nex;
#ifdef PSGPU_FT_PROFILE
        if (tid == 0) { s_prof[13] += n_exq > FF_EXIT_CAP ? 1 : 0; s_prof[14] += n_exq; s_over = n_exq > FF_EXIT_CAP; s_t5 = clock64();
                        s_prof[23] += n_exq > 192; s_prof[24] += n_exq > 224; s_prof[25] += n_exq > 256; s_prof[26] += n_exq > 320; }
#endif
        if (s_nex == 0) { }                                  // (a frame without exits: nothing to write, no barrier to meet)
        else if (s_nex <= FF_EXIT_CAP) {
            static_assert(kFfMaxExit <= kFfThreads, "one queued exit per work-item");
            const int32_t bpidx = s_sc[1], bss_head = s_sc[2];
            const int n_ex = s_nex;
            {   // ranks by counting, over the keys laid side by side first (the sorted queue's place, not yet in use): four keys a read, eight in flight
                int32_t *const keys = reinterpret_cast<int32_t *>(s_srt);
                if (tid < n_ex) { keys[tid] = s_ex[tid][0]; s_gkey[tid] = 0ull; s_ghave[2 * tid] = 0u; s_ghave[2 * tid + 1] = 0u; s_gdiff[tid] = 0; }
                if (tid < 8) keys[n_ex + tid] = 0x7fffffff;
                ff_sync_lds();
                if (tid < n_ex) {
                    const int32_t key = keys[tid];
                    int r = 0;
                    for (int j = 0; j < n_ex; j += 8) {
                        const FfQuad a = *reinterpret_cast<const FfQuad *>(keys + j), b = *reinterpret_cast<const FfQuad *>(keys + j + 4);
                        r += (a.x < key) + (a.y < key) + (a.z < key) + (a.w < key) + (b.x < key) + (b.y < key) + (b.z < key) + (b.w < key);
                    }
                    s_ord[r] = (uint16_t)tid;
                }
            }
            FF_PROFS(36);
            ff_sync_lds();
            // sorted position r = tid: a word's first exit (`head`) counts one entry and the word's stack block; one prefix sum gives
            // every exit the number of entries and stack slots before its word
            const int32_t *x = s_ex[tid < n_ex ? s_ord[tid] : 0];
            const int i = x[0] >> 10;
            if (tid < n_ex) s_srt[tid] = FfQuad{ i, x[2], x[3], x[5] };
            if (tid < 4) s_srt[n_ex + tid] = FfQuad{ -1, 0, 0, 0 };            // (sentinels: a walk reads four exits at a time)
            const bool mine = tid < n_ex, head = mine && (tid == 0 || (s_ex[s_ord[tid - 1]][0] >> 10) != i);
            int32_t total;
            const int32_t before = ff_block_excl_sum(head ? ((x[4] << 10) | 1) : 0, s_scan, total);
            const int32_t n_exit = total & 1023, n_bss = total >> 10;
            FF_PROFS(37);
            const bool full = bpidx + n_exit >= u.bp_cap || bss_head + n_bss + p.n_ci >= u.bss_cap;
            if (tid == 0) { if (full) s_sc[3] = 1; else { s_sc[1] = bpidx + n_exit; s_sc[2] = bss_head + n_bss; } }   // (full: nothing is written)
            if (mine && !full) {
                const bool single = (x[1] >> 30) != 0;
                const int32_t bsh = bss_head + (before >> 10) - (head ? 0 : x[4]);
                if (!single && x[4]) {
                    u.bss[bsh + x[5]] = x[2];                               // its score in its slot of the word's stack block
                    // the contexts nothing exited into (ngram_search.c:468-475 fills the block before the first score goes in): a word's exits
                    // are queued in context order, each takes the slots between its predecessor's and its own, the last one the rest -- not
                    // one work-item a word over all of them
                    for (int q = head ? 0 : s_srt[tid - 1].w + 1; q < x[5]; ++q) u.bss[bsh + q] = kW;
                    if (s_srt[tid + 1].x != i) for (int q = x[5] + 1; q < x[4]; ++q) u.bss[bsh + q] = kW;
                }
                // the word's entry takes the best of its exits, the first among equals (the update branch of save_bp, ngram_search.c:405-437:
                // a later exit takes the entry over with a strictly better score): every exit offers its key to the word's maximum and its
                // context to the word's set -- a walk by the word's first exit over all of them is only needed where it has side effects,
                // i.e. where two of the exits' histories name different real words (set_real_wid with the old back-pointer in place)
                const int g = (before & 1023) - (head ? 0 : 1);
                atomicMax(&s_gkey[g], ((unsigned long long)((uint32_t)x[2] ^ 0x80000000u) << 32) | (uint32_t)(0xffffffffu - (uint32_t)tid));
                atomicOr(&s_ghave[2 * g + (x[5] >> 5)], 1u << (x[5] & 31));
                if (!head) { const int32_t *pe = s_ex[s_ord[tid - 1]]; if (pe[9] != x[9] || pe[10] != x[10]) s_gdiff[g] = 1; }
                if (head) {
                    const int w = x[1] & 0x3fffffff;
                    ff_new_bp(u, bpidx + g, bsh, f, w, x[2], x[3], single, x[6], x[7], x[8] & 0x3fffffff, (x[8] >> 30) != 0, x[9], x[10]);
                }
            }
            ff_sync_lds();
            if (mine && !full) {
                if (head) {
                    const int w = x[1] & 0x3fffffff, g = before & 1023;
                    const int32_t bpi = bpidx + g;
                    int32_t cs = x[2], cp = x[3];
                    int32_t cp_real = x[9], cp_preal = x[10];          // the real words of the entry's history (asked for when the exit was queued)
                    bool dirty = false, requirk = false;
                    unsigned long long have = 1ull << x[5];
                    if (!s_gdiff[g] && !FF_FORCE_WALK) {
                        const unsigned long long key = s_gkey[g];
                        const int wpos = (int)(0xffffffffu - (uint32_t)key);
                        have = (unsigned long long)s_ghave[2 * g] | ((unsigned long long)s_ghave[2 * g + 1] << 32);
                        if (wpos != tid) {
                            cs = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u); cp = s_srt[wpos].z; dirty = true;
                            if (cp != x[3]) FBP(u, F_BP, bpi) = cp;
                        }
                    }
                    else
                    for (int r2 = tid + 1; r2 < n_ex; r2 += 4) {                 // the walk, in the queue's order
                        const FfQuad y4[4] = { s_srt[r2], s_srt[r2 + 1], s_srt[r2 + 2], s_srt[r2 + 3] };    // four exits read at a time
                        bool more = true;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const FfQuad y = y4[t];
                            more = more && y.x == i;                             // (the sentinels after the queue's end belong to no word)
                            if (!more) continue;
                            have |= 1ull << y.w;
                            if (cs < y.y) {
                                if (cp != y.z) {
                                    // (the two histories' real words: every exit's came with it into the queue -- no trip to the table)
                                    const int32_t *xe = s_ex[s_ord[r2 + t]];
                                    const int32_t n1 = xe[9], n0 = xe[10];
                                    if (cp_preal != n0 || cp_real != n1) { ff_set_real_wid(p, u, bpi); requirk = true; }      // with the old bp still in place, as the reference
                                    FBP(u, F_BP, bpi) = y.z;
                                    cp = y.z; cp_real = n1; cp_preal = n0;
                                }
                                cs = y.y; dirty = true;
                            }
                        }
                        if (!more) break;
                    }
                    if (dirty) FBP(u, F_SCORE, bpi) = cs;
                    if ((before & 1023) < FF_PAIR_ROWS) {   // what the word transitions below read of this entry (a frame with more new entries than rows reads the table)
                        int32_t *nb = s_nbp[before & 1023];
                        const bool filler = (x[8] >> 30) != 0;
                        const int32_t base = x[8] & 0x3fffffff;
                        nb[0] = w; nb[1] = x[6]; nb[2] = x[7]; nb[3] = cs; nb[4] = tid;
                        if (requirk) { nb[5] = FBP(u, F_REAL, bpi); nb[6] = FBP(u, F_PREAL, bpi); }
                        else if (filler) { nb[5] = x[3] != -1 ? x[9] : base; nb[6] = x[3] != -1 ? x[10] : -1; }
                        else { nb[5] = base; nb[6] = x[9]; }
                        nb[7] = (int32_t)(uint32_t)have; nb[8] = (int32_t)(uint32_t)(have >> 32);
                    }
                }
            }
            FF_PROFS(38);
            ff_sync_lds();
            FF_PROFS(39);
        }
        else {
            // more exits than the queue holds: through flags in the slab, one work-item per exiting word walking its chain
            const int32_t bpidx = s_sc[1], bss_head = s_sc[2];
            for (int i = tid; i < na; i += kFfThreads) { u.cnt_a[i] = 0; u.cnt_b[i] = 0; }
            __syncthreads();
            for (int e = tid; e < n_eval; e += kFfThreads) {                      // the exit test again (what failed it was cleared above)
                const FfEnt en = ent(e);
                const int c = en.c & kFfChanMask, i = en.inf >> 10, k = en.inf & 1023, w = en.w;
                if (!(u.best[c] > thresh && u.out[c] > wordthresh)) continue;
                if (k == 0 ? (en.aux & 1023) > 0 : u.rcid[c] < 0) continue;
                u.xflag[c] = 1; u.cnt_a[i] = 1;
                u.cnt_b[i] = p.d_pronlen[w] > 1 ? p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]] : 0;
            }
            __syncthreads();
            const int32_t n_exit = ff_block_scan(u.cnt_a, na, s_scan);
            const int32_t n_bss = ff_block_scan(u.cnt_b, na, s_scan);
            const bool full = bpidx + n_exit >= u.bp_cap || bss_head + n_bss + p.n_ci >= u.bss_cap;
            for (int i = tid; i < na; i += kFfThreads) {
                if ((i + 1 < na ? u.cnt_a[i + 1] : n_exit) == u.cnt_a[i]) continue;
                const int w = awl_c[i * 3], c0 = awl_c[i * 3 + 1], len = awl_c[i * 3 + 2] & 1023;
                const int32_t bpi = bpidx + u.cnt_a[i], bsh = bss_head + u.cnt_b[i];
                for (int k = 0; k < len; ++k) {
                    const int c = c0 + k;
                    if (!u.xflag[c]) continue;
                    u.xflag[c] = 0;
                    if (!full) ff_save_bp(p, u, bpi, bsh, f, w, u.out[c], u.outh[c], k == 0 ? 0 : u.rcid[c]);
                }
            }
            __syncthreads();
            if (tid == 0) { if (full) s_sc[3] = 1; else { s_sc[1] = bpidx + n_exit; s_sc[2] = bss_head + n_bss; } }   // (full: nothing was written)
        }
        ff_sync_lds();
        if (s_sc[3]) break;

        FF_PROF(5);
        // ---- fwdflat_word_transition (:642-782).  The successors of a frame's exits are the vocabulary words with a node in the frame's
        //      window: a contiguous slice of the nodes-by-start-frame list (get_expand_wordlist :609-640), taken one work-item an entry and
        //      asked for at the top of the frame -- the slice, its words (a word with two nodes in the window is taken once: wseen) and
        //      their static quads are in registers by now.  The work is one work-item per (word, new entry) PAIR: its right-context slot and
        //      language score asked for together, the best pair of a word kept as one 64-bit maximum in LDS -- "the first best exit wins"
        //      (:700-720: a later exit replaces an earlier one only with a strictly better score) is the earliest index among the highest
        //      scores --, then the word's one work-item enters it.  <sil>'s best exit (:745-753) is one more row of pairs.  A frame's new
        //      entries are read from LDS as the exits' phase left them; a frame whose exits overflowed the queue loads their rows from the
        //      table first and finds exit scores in the score stack.
        const int bp0 = bp_first, bp1 = s_sc[1];
        const int n_new = bp1 - bp0;
        if (n_new > 0) {
            const bool lds_exits = n_exq <= FF_EXIT_CAP, by_pairs = n_new <= FF_PAIR_ROWS;
            auto exit_score = [&](const int32_t *r, int slot) {      // bscore_stack[s_idx + slot] of a new entry: the exit into that context
                if (!lds_exits) return u.bss[r[4] + slot];
                // (a word's exits are queued in chain order = right-context order: the slot's place among them is a population count)
                const unsigned long long have = (unsigned long long)(uint32_t)r[7] | ((unsigned long long)(uint32_t)r[8] << 32);
                if (!((have >> slot) & 1ull)) return kW;
                return s_srt[r[4] + __popcll(have & ((1ull << slot) - 1ull))].y;
            };
            if (by_pairs) {
                if (!lds_exits && tid < n_new) {              // the rows from the table: word, last / last-but-one phone, score, stack offset, real words
                    const int b = bp0 + tid;
                    int32_t *r = s_nbp[tid];
                    r[0] = FBP(u, F_WID, b); r[1] = FBP(u, F_LAST, b); r[2] = FBP(u, F_LAST2, b); r[3] = FBP(u, F_SCORE, b);
                    r[4] = FBP(u, F_SIDX, b); r[5] = FBP(u, F_REAL, b); r[6] = FBP(u, F_PREAL, b);
                }
                if (!lds_exits && tid < n_new) u.word_lat_idx[s_nbp[tid][0]] = -1;      // (the slab path's exits set it)
            }
            else {
                __syncthreads();                             // (the new entries are read from the table: what the exits' phase wrote must be there)
                for (int b = bp0 + tid; b < bp1; b += kFfThreads) {
                    const int wid = FBP(u, F_WID, b);
                    u.word_lat_idx[wid] = -1;
                    if (wid == p.finishwid) continue;
                    const int l2 = FBP(u, F_LAST2, b), l1 = FBP(u, F_LAST, b);
                    const int32_t sil = l2 == -1 ? FBP(u, F_SCORE, b)
                        : u.bss[FBP(u, F_SIDX, b) + p.rs_cimap[((size_t)l1 * p.n_ci + l2) * p.n_ci + p.sil_ci]];
                    // best exit into silence, the earliest on ties (:745-753): key = (score, -index)
                    if (sil > kW)
                        atomicMax(&s_key, ((unsigned long long)(uint32_t)(sil - kW) << 32) | (uint32_t)(0x7fffffff - b));
                }
            }
            for (int cb = 0; cb == 0 || cb < sl_n; cb += FF_SL_CHUNK) {
                // this chunk's slice entries, one a work-item (the first chunk's were asked for at the top of the frame)
                FfQuad wq4 = sl_w;
                bool on = sl_on;
                if (cb > 0) {
                    on = false;
                    if (cb + tid < sl_n && tid < FF_SL_CHUNK) {
                        const int k = u.fr_words[sl_q0 + cb + tid];
                        on = atomicExch(&u.wseen[k], nf) != nf;
                        wq4 = *reinterpret_cast<const FfQuad *>(u.wstat + 4 * (size_t)k);
                    }
                }
                const int n_act = min(sl_n - cb, FF_SL_CHUNK);
                const int c0 = wq4.y, first = wq4.z & 0xff, ci2 = wq4.z >> 8;
                int32_t cur_fr = 0, cur_sc = 0;
                if (on) { cur_fr = u.frame[c0]; cur_sc = u.score[c0 * 5]; }
                if (by_pairs) {
                    s_wfirst[tid] = on ? first : -1; s_wbase[tid] = wq4.w; s_wkey[tid] = 0ull;
                    ff_sync_lds();
                    const int n_row = n_act + (cb == 0 ? 1 : 0);                     // (the first chunk: one more row, <sil>'s)
                    for (int pr = tid, n_pair = n_row * n_new; pr < n_pair; pr += kFfThreads) {
                        const int t = pr / n_new, e = pr - t * n_new;
                        const int32_t *r = s_nbp[e];
                        const bool sil = t == n_act;
                        const int fi = sil ? p.sil_ci : s_wfirst[t];
                        if (r[0] == p.finishwid || fi < 0) continue;
                        int32_t slot = 0, lmv = 0;
                        if (r[2] != -1) slot = p.rs_cimap[((size_t)r[1] * p.n_ci + r[2]) * p.n_ci + fi];
                        if (!sil) lmv = ff_lm(p, s_wbase[t], r[5], r[6]);
                        int32_t newscore = r[2] == -1 ? r[3] : exit_score(r, slot);
                        if (sil) {       // best exit into silence, the earliest on ties (:745-753): key = (score, -index)
                            if (newscore > kW)
                                atomicMax(&s_key, ((unsigned long long)(uint32_t)(newscore - kW) << 32) | (uint32_t)(0x7fffffff - (bp0 + e)));
                            continue;
                        }
                        if (newscore == kW) continue;
                        // "newscore += lwf * (ngram_tg_score(...) >> SENSCR_SHIFT)": float product and sum, truncated (:700-706)
                        const float prod = __fmul_rn(p.lwf, (float)lmv);
                        newscore = (int32_t)__fadd_rn((float)newscore, prod);
                        newscore += p.pip;
                        if (newscore > thresh)
                            atomicMax(&s_wkey[t], ((unsigned long long)((uint32_t)newscore ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - e));
                    }
                    ff_sync_lds();
                    if (on) {
                        const unsigned long long key = s_wkey[tid];
                        const int32_t sc = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
                        const int win = 0x7fffffff - (int)(uint32_t)key;
                        if (key && (cur_fr < f || sc > cur_sc)) {
                            ff_enter(u, c0, sc, bp0 + win, nf);
                            u.senid[c0 * 5] = p.ldiph[((size_t)first * p.n_ci + ci2) * p.n_ci + s_nbp[win][1]];
                            u.word_active[wq4.x] = nf;
                        }
                    }
                    if (cb + FF_SL_CHUNK < sl_n) ff_sync_lds();                    // (the next chunk overwrites the rows)
                }
                else if (on) {
                    // more new entries than rows: the word's work-item goes through the table, exits in order
                    for (int b = bp0; b < bp1; ++b) {
                        if (FBP(u, F_WID, b) == p.finishwid) continue;
                        const int l2 = FBP(u, F_LAST2, b), l1 = FBP(u, F_LAST, b);
                        int32_t newscore = l2 == -1 ? FBP(u, F_SCORE, b)
                            : u.bss[FBP(u, F_SIDX, b) + p.rs_cimap[((size_t)l1 * p.n_ci + l2) * p.n_ci + first]];
                        if (newscore == kW) continue;
                        const float prod = __fmul_rn(p.lwf, (float)ff_lm(p, wq4.w, FBP(u, F_REAL, b), FBP(u, F_PREAL, b)));
                        newscore = (int32_t)__fadd_rn((float)newscore, prod);
                        newscore += p.pip;
                        if (newscore > thresh && (cur_fr < f || newscore > cur_sc)) {
                            cur_fr = nf; cur_sc = newscore;
                            ff_enter(u, c0, newscore, b, nf);
                            u.senid[c0 * 5] = p.ldiph[((size_t)first * p.n_ci + ci2) * p.n_ci + l1];
                            u.word_active[wq4.x] = nf;
                        }
                    }
                }
            }
            if (!by_pairs) ff_sync_lds();                  // (<sil>'s key is complete)
        }
#ifdef PSGPU_FT_PROFILE
        if (tid == 0) {      // the word transitions' shape: [16] frames with new entries, [17] their entries, [18] slice entries, [19] (word, entry) pairs,
            const long long t_ = clock64();   // [20] / [21] this phase's cycles in frames whose exits fit / overflow the queue, [22] the per-utterance maximum of entries
            if (n_new > 0) { s_prof[16] += 1; s_prof[17] += n_new; s_prof[18] += sl_n; s_prof[19] += (long long)n_new * sl_n; }
            s_prof[s_over ? 21 : 20] += t_ - s_last;
            if (n_new > s_prof[22]) s_prof[22] = n_new;
        }
#endif
        // (a filler word that is a word of the language model can have been entered as a successor: the fillers' test must see it)
        if (p.fill_known) __syncthreads();
        FF_PROF(12);
        if (bp1 > bp0) {
            // <sil> and the noise words (:755-769)
            const unsigned long long key = s_key;
            const int32_t silscore = key ? (int32_t)(uint32_t)(key >> 32) + kW : kW;
            const int32_t silbp = key ? 0x7fffffff - (int32_t)(uint32_t)(key & 0xffffffffu) : 0;
            for (int w = p.filler_start - 1 + tid; w <= p.filler_end; w += kFfThreads) {
                const bool is_sil = w == p.filler_start - 1;       // slot filler_start - 1 stands for <sil>
                if (!is_sil && w == p.silwid) continue;
                const int ww = is_sil ? p.silwid : w;
                const int c = p.w1_of_word[ww];
                if (c < 0) continue;                               // noise words that are not a single phone have no channel
                const int32_t ns = silscore + (is_sil ? p.silpen : p.fillpen) + p.pip;
                if (ns > thresh && ns > kW && (u.frame[c] < f || ns > u.score[c * 5])) {
                    ff_enter(u, c, ns, silbp, nf);
                    u.word_active[ww] = nf;
                }
            }
        }
        __syncthreads();
        // initial channels of words that stayed inactive (:771-781): a root still stamped with this frame was evaluated and neither kept
        // nor entered.  With the candidates' records in registers the pass below asks for the stamp beside the word's (a root stamped f
        // belongs to a word of this frame's list, and every such word is a candidate); otherwise over the list
        if (n_all > FF_AWL_REGS)
            for (int i = tid; i < na; i += kFfThreads) {
                const int c0 = awl_c[i * 3 + 1];
                if (u.frame[c0] == f) ff_clear_scores(p, u, c0);
            }
        FF_PROF(6);
        // ---- next active word list (:853-869): the vocabulary in its order (words below <s>), then <s> and above by id
        int32_t n_next;
        if (n_all <= FF_AWL_REGS) {                          // a row of 256 candidates at a time, their places by a prefix sum (one row, usually)
            n_next = 0;
#pragma unroll
            for (int j = 0; j < kFfRegRows; ++j) {                    // (a candidate's word, first channel and lengths never change: registers)
                if (j * kFfThreads >= n_all) break;
                const int i = tid + j * kFfThreads;
                const int32_t wa = u.word_active[wq[j]], rfr = i < n_all && c0q[j] >= 0 ? u.frame[c0q[j]] : -1;
                if (rfr == f) ff_clear_scores(p, u, c0q[j]);
                const bool on = i < n_all && wa == nf && (i < u.nwd ? wq[j] < p.startwid : true);
                int32_t row_total;
                const int pos = n_next + ff_block_excl_sum(on ? 1 : 0, (j & 1) ? s_scan2 : s_scan, row_total);
                if (on) {
                    int32_t *a = awl_n + 3 * pos; a[0] = wq[j]; a[1] = c0q[j]; a[2] = axq[j];
                    if (pos < kFfAwlLds) { s_awl[pos * 3] = wq[j]; s_awl[pos * 3 + 1] = c0q[j]; s_awl[pos * 3 + 2] = axq[j]; }
                }
                n_next += row_total;
            }
            awl_lds = n_next <= kFfAwlLds;
        }
        else {
            // larger vocabularies: eight consecutive candidates a work-item per round (their words, then their stamps, asked for
            // together), places by a prefix sum over the round's counts
            n_next = 0;
            for (int base = 0, round = 0; base < n_all; base += 8 * kFfThreads, ++round) {
                int wv[8], fl = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = base + tid * 8 + j;
                    wv[j] = i < u.nwd ? u.wl_wid[i] : p.startwid + (i < n_all ? i - u.nwd : 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = base + tid * 8 + j;
                    if (i < n_all && u.word_active[wv[j]] == nf && (i < u.nwd ? wv[j] < p.startwid : true)) fl |= 1 << j;
                }
                int32_t round_total;
                int pos = n_next + ff_block_excl_sum(__popc(fl), (round & 1) ? s_scan2 : s_scan, round_total);
#pragma unroll
                for (int j = 0; j < 8; ++j) if (fl & (1 << j)) ff_awl_put(p, u, awl_n, pos++, wv[j]);
                n_next += round_total;
            }
            awl_lds = false;
        }
        if (nxt) n_awl1 = n_next; else n_awl0 = n_next;
        if (tid == 0) {
            u.step[f * 4] = s_sc[0]; u.step[f * 4 + 1] = 0; u.step[f * 4 + 2] = s_sc[1]; u.step[f * 4 + 3] = n_next;
            ++s_sc[4];
        }
        __syncthreads();
        FF_PROF(7);
#ifdef PSGPU_FT_PROFILE
        if (tid == 0 && s_over) s_prof[15] += clock64() - s_t5;      // cycles from the pruning's decisions to the frame's end, frames whose exits overflow the queue
#endif
    }
#ifdef PSGPU_FT_PROFILE
    if (tid == 0 && bf.prof) for (int i = 0; i < 48; ++i) bf.prof[(size_t)ub * 48 + i] = s_prof[i];
#endif
    if (tid == 0) {
        u.bp_table_idx[s_sc[4]] = s_sc[1];                       // ngram_fwdflat_finish: mark one past the last frame
        u.result[0] = s_sc[1]; u.result[1] = s_sc[2]; u.result[2] = s_sc[4]; u.result[3] = s_sc[3]; u.result[4] = s_sc[0];
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static const T *ff_up(psgpu_fwdflat_s *m, const T *src, size_t n, int *rc)
{
    void *d = nullptr;
    if (*rc != PSGPU_OK) return nullptr;
    const size_t bytes = n * sizeof(T);
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess || hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        psgpu_set_error("fwdflat model upload failed");
        *rc = PSGPU_ENOMEM;
        hipFree(d);
        return nullptr;
    }
    m->allocs.push_back(d);
    return (const T *)d;
}

extern "C" {

int psgpu_fwdflat_create(psgpu_fwdflat_t **out, const psgpu_fwdflat_tables_t *t)
{
    PSGPU_REQUIRE(out && t && t->ft && t->ft->par && t->pron_off && t->pron_ci && t->pron_ssid && t->ci_ssid && t->lm_known,
                  "psgpu_fwdflat_create: NULL argument");
    *out = nullptr;
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    const psgpu_fwdtree_tables_t *ft = t->ft;
    const int32_t *q = ft->par;
    psgpu_fwdflat_s *m = new psgpu_fwdflat_s();
    FfDev &d = m->d;
    memset(&d, 0, sizeof d);
    d.n_ci = q[0]; d.n_emit = q[1]; d.n_sen = q[2]; d.n_w = q[3]; d.n1 = q[6];
    d.beam = q[8]; d.pip = q[13]; d.silpen = q[15]; d.fillpen = q[16];
    d.startwid = q[19]; d.finishwid = q[20]; d.silwid = q[21]; d.filler_start = q[22]; d.filler_end = q[23]; d.sil_ci = q[24];
    d.fwdflatbeam = t->fwdflatbeam; d.fwdflatwbeam = t->fwdflatwbeam; d.min_ef_width = t->min_ef_width; d.max_sf_win = t->max_sf_win;
    d.lwf = t->lwf;
    if (!(d.n_emit == 3 || d.n_emit == 5) || d.n_ci < 1 || d.n_ci > kFfMaxCi || d.n_w < 1 || d.n1 < 1 ||
        d.startwid < 0 || d.startwid >= d.n_w) {
        psgpu_set_error("fwdflat: unsupported shape (n_emit %d, n_ci %d, words %d, single-phone words %d)", d.n_emit, d.n_ci, d.n_w, d.n1);
        delete m;
        return PSGPU_EINVAL;
    }
    const size_t nci3 = (size_t)d.n_ci * d.n_ci * d.n_ci, n1 = (size_t)d.n_w + 1;
    std::vector<int32_t> w1_of(d.n_w, -1);
    for (int i = 0; i < d.n1; ++i) w1_of[ft->w1_wid[i]] = i;
    if (w1_of[d.startwid] < 0 || w1_of[d.silwid] < 0) {
        psgpu_set_error("fwdflat: <s> and <sil> must be single-phone words");
        delete m;
        return PSGPU_EINVAL;
    }
    const size_t n_pron = (size_t)t->pron_off[d.n_w];
    d.w1_wid = ff_up(m, ft->w1_wid, d.n1, &rc); d.w1_ci2 = ff_up(m, ft->w1_ci2, d.n1, &rc);
    d.w1_ssid = ff_up(m, ft->w1_ssid, d.n1, &rc); d.w1_tmat = ff_up(m, ft->w1_tmat, d.n1, &rc); d.w1_mpx = ff_up(m, ft->w1_mpx, d.n1, &rc);
    d.w1_of_word = ff_up(m, w1_of.data(), d.n_w, &rc);
    d.d_pronlen = ff_up(m, ft->dict_pronlen, d.n_w, &rc); d.d_first = ff_up(m, ft->dict_first, d.n_w, &rc);
    d.d_last = ff_up(m, ft->dict_last, d.n_w, &rc); d.d_last2 = ff_up(m, ft->dict_last2, d.n_w, &rc);
    d.d_base = ff_up(m, ft->dict_basewid, d.n_w, &rc); d.d_filler = ff_up(m, ft->dict_filler, d.n_w, &rc);
    d.rs_n = ff_up(m, ft->rssid_n, (size_t)d.n_ci * d.n_ci, &rc); d.rs_ssid = ff_up(m, ft->rssid_ssid, nci3, &rc);
    d.rs_cimap = ff_up(m, ft->rssid_cimap, nci3, &rc); d.ldiph = ff_up(m, ft->ldiph_lc, nci3, &rc);
    d.ci_tmat = ff_up(m, ft->ci_tmat, d.n_ci, &rc);
    d.lm = ft->lm ? ff_up(m, ft->lm, (size_t)d.n_w * n1 * n1, &rc) : nullptr;
    d.pron_off = ff_up(m, t->pron_off, (size_t)d.n_w + 1, &rc); d.pron_ci = ff_up(m, t->pron_ci, n_pron, &rc);
    d.pron_ssid = ff_up(m, t->pron_ssid, n_pron, &rc); d.ci_ssid = ff_up(m, t->ci_ssid, d.n_ci, &rc);
    d.tp = ff_up(m, ft->tp, (size_t)ft->n_tmat * d.n_emit * (d.n_emit + 1), &rc);
    d.tp_bytes = (int32_t)((size_t)ft->n_tmat * d.n_emit * (d.n_emit + 1));
    d.sseq = ff_up(m, ft->sseq, (size_t)ft->n_sseq * d.n_emit, &rc);
    m->h_pronlen.assign(ft->dict_pronlen, ft->dict_pronlen + d.n_w);
    m->h_last.assign(ft->dict_last, ft->dict_last + d.n_w); m->h_last2.assign(ft->dict_last2, ft->dict_last2 + d.n_w);
    m->h_rs_n.assign(ft->rssid_n, ft->rssid_n + (size_t)d.n_ci * d.n_ci);
    m->h_known.assign(t->lm_known, t->lm_known + d.n_w);
    d.fill_known = 0;                    // (a filler word the language model knows can be a successor AND be entered as a filler in one frame: order matters then)
    for (int w = 0; w < d.n_w; ++w)
        if (m->h_known[w] && (w == d.silwid || (w >= d.filler_start && w <= d.filler_end))) d.fill_known = 1;
    if (rc != PSGPU_OK) { psgpu_fwdflat_free(m); return rc; }
    *out = m;
    return PSGPU_OK;
}

const LmDev *psgpu_lm_dev(const psgpu_lm_t *lm);     // psgpu_lm.hip

int psgpu_fwdflat_set_lm(psgpu_fwdflat_t *m, const psgpu_lm_t *lm)
{
    PSGPU_REQUIRE(m && lm, "psgpu_fwdflat_set_lm: NULL argument");
    const LmDev *d = psgpu_lm_dev(lm);
    PSGPU_REQUIRE(d->n_words == m->d.n_w, "psgpu_fwdflat_set_lm: the model maps %d dictionary words, the search has %d", d->n_words, m->d.n_w);
    // (the first pass reaches its model through one out-of-line look-up, the second inlines it into a kernel that has no registers to
    //  spare for a call: an interpolated set -- ngram_model_set_interp, reachable through the API only -- stays with the first pass)
    PSGPU_REQUIRE(d->n_set == 0, "psgpu_fwdflat_set_lm: the second pass on the device takes one model (a set's current member), not an interpolated set");
    m->d.trie = *d;
    m->d.use_trie = 1;
    return PSGPU_OK;
}

void psgpu_fwdflat_free(psgpu_fwdflat_t *m)
{
    if (!m) return;
    for (void *p : m->allocs) hipFree(p);
    for (int k = 0; k < 6; ++k) if (m->work[k]) { if (k < 3) hipFree(m->work[k]); else hipHostFree(m->work[k]); }
    delete m;
}

}  // extern "C"

// build_fwdflat_wordlist (:223-300) for one utterance from the first pass's back-pointer columns (frame, wid, bp):
// one node per (start frame, word), new nodes at the head of their start frame's list, nodes with too few end points
// (and </s> not ending in the last frame) dropped, the vocabulary in order of first appearance walking the frames.
struct FfVocab { std::vector<int32_t> wid, chain, len, node_off, node_sf, fr_off, fr_words; int32_t n_chan = 0; };

static void ff_build_vocab(const psgpu_fwdflat_s *m, const int32_t *fr, const int32_t *wid, const int32_t *bpc, int nb,
                           int n_frame, int32_t chan_base, FfVocab &v)
{
    const FfDev &d = m->d;
    std::vector<int32_t> n_sf, n_wid, n_fef, n_lef, n_next, head(n_frame + 2, -1);
    for (int i = 0; i < nb; ++i) {
        const int sf = bpc[i] < 0 ? 0 : fr[bpc[i]] + 1, ef = fr[i], w = wid[i];
        if (w < 0 || w >= d.n_w || sf > n_frame || !m->h_known[w]) continue;
        int nd;
        for (nd = head[sf]; nd >= 0 && n_wid[nd] != w; nd = n_next[nd]);
        if (nd >= 0) n_lef[nd] = ef;
        else {
            n_sf.push_back(sf); n_wid.push_back(w); n_fef.push_back(ef); n_lef.push_back(ef);
            n_next.push_back(head[sf]); head[sf] = (int32_t)n_sf.size() - 1;
        }
    }
    for (int f = 0; f < n_frame; ++f) {
        int prev = -1, next;
        for (int nd = head[f]; nd >= 0; nd = next) {
            next = n_next[nd];
            if (n_lef[nd] - n_fef[nd] < d.min_ef_width || (n_wid[nd] == d.finishwid && n_lef[nd] < n_frame - 1)) {
                if (prev < 0) head[f] = next; else n_next[prev] = next;
            }
            else prev = nd;
        }
    }
    std::vector<int32_t> slot(d.n_w, -1);
    std::vector<std::vector<int32_t>> sfs;
    for (int f = 0; f < n_frame; ++f)
        for (int nd = head[f]; nd >= 0; nd = n_next[nd]) {
            const int w = n_wid[nd];
            if (slot[w] < 0) { slot[w] = (int32_t)v.wid.size(); v.wid.push_back(w); sfs.emplace_back(); }
            sfs[slot[w]].push_back(f);
        }
    int32_t c = chan_base;
    v.node_off.push_back(0);
    for (size_t k = 0; k < v.wid.size(); ++k) {
        const int w = v.wid[k];
        if (m->h_pronlen[w] == 1) { v.chain.push_back(-1); v.len.push_back(1); }
        else {
            const int len = 1 + (m->h_pronlen[w] - 2) + m->h_rs_n[m->h_last[w] * d.n_ci + m->h_last2[w]];
            v.chain.push_back(c); v.len.push_back(len); c += len;
        }
        v.node_sf.insert(v.node_sf.end(), sfs[k].begin(), sfs[k].end());
        v.node_off.push_back((int32_t)v.node_sf.size());
    }
    v.n_chan = c - chan_base;
    // the nodes by start frame (a counting sort): the successors of a frame's exits are the words with a node in its window
    v.fr_off.assign((size_t)n_frame + 2, 0);
    for (size_t k = 0; k < sfs.size(); ++k) for (int sf : sfs[k]) ++v.fr_off[(size_t)sf + 1];
    for (int f = 0; f <= n_frame; ++f) v.fr_off[(size_t)f + 1] += v.fr_off[f];
    v.fr_words.assign(v.node_sf.size(), 0);
    {
        std::vector<int32_t> at(v.fr_off.begin(), v.fr_off.end() - 1);
        for (size_t k = 0; k < sfs.size(); ++k) for (int sf : sfs[k]) v.fr_words[at[sf]++] = (int32_t)k;
    }
}

// raw == NULL: the frames' scores are given (senscr_dev); else the kernel scores its own senones from raw->feats
static int ff_search(psgpu_fwdflat_t *m, const int16_t *senscr_dev, int64_t scr_stride, const FfRaw *raw,
                     const int32_t *utt_off_dev, int32_t n_utt, int32_t max_frames,
                     int32_t bp1_cap, const int32_t *bp1_dev, const int32_t *result1_dev,
                     const int32_t *w1_ssid_dev, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev,
                     int32_t *bss_dev, int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, void *stream)
{
    PSGPU_REQUIRE(m && n_utt >= 0 && max_frames >= 0 && bp_cap > 0 && bss_cap > 0 && bp1_cap > 0, "psgpu_fwdflat_search: bad argument");
    PSGPU_REQUIRE(m->d.lm || m->d.use_trie, "psgpu_fwdflat_search: no language model (dense table or psgpu_fwdflat_set_lm)");
#ifdef PSGPU_FT_PROFILE
    double t_host[6]; int n_th = 0;
    auto stamp = [&]() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); if (n_th < 6) t_host[n_th++] = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    stamp();
#endif
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE((senscr_dev || raw) && utt_off_dev && bp1_dev && result1_dev && bp_dev && bss_dev && idx_dev && step_dev && result_dev,
                  "psgpu_fwdflat_search: NULL device buffer");
    const FfDev &d = m->d;
    hipStream_t st = (hipStream_t)stream;
    // ---- the first pass's tables to the host: counts, then the three columns the vocabulary needs
    std::vector<int32_t> res1((size_t)n_utt * 8);
    PSGPU_HIP(hipMemcpyAsync(res1.data(), result1_dev, sizeof(int32_t) * res1.size(), hipMemcpyDeviceToHost, st));
    PSGPU_HIP(hipStreamSynchronize(st));
    // three strided copies for the whole batch: column k of every utterance's table, cut to the longest table
    int max_nb = 1;
    for (int u = 0; u < n_utt; ++u) {
        const int nb = res1[(size_t)u * 8];
        PSGPU_REQUIRE(nb >= 0 && nb <= bp1_cap, "psgpu_fwdflat_search: utterance %d: %d first-pass back-pointers, capacity %d", u, nb, bp1_cap);
        max_nb = std::max(max_nb, nb);
    }
    int rcw;
    int32_t *cols = nullptr;
    if ((rcw = ff_work(m, 3, sizeof(int32_t) * (size_t)3 * n_utt * max_nb, (void **)&cols))) return rcw;
    {
        static const int kCol[3] = {F_FRAME, F_WID, F_BP};
        for (int k = 0; k < 3; ++k)
            PSGPU_HIP(hipMemcpy2DAsync(cols + (size_t)k * n_utt * max_nb, sizeof(int32_t) * max_nb,
                                       bp1_dev + (size_t)kCol[k] * bp1_cap, sizeof(int32_t) * 10 * bp1_cap,
                                       sizeof(int32_t) * max_nb, n_utt, hipMemcpyDeviceToHost, st));
    }
    PSGPU_HIP(hipStreamSynchronize(st));
#ifdef PSGPU_FT_PROFILE
    stamp();
#endif
    // ---- vocabulary + chain layout per utterance, slab sizes
    std::vector<FfVocab> voc(n_utt);
    std::vector<size_t> slab_off(n_utt + 1, 0), voc_off(n_utt + 1, 0);
    const int n_tail = d.n_w - d.startwid;
    {   // the utterances' vocabularies are independent of each other: host threads (30 s of audio on the small task: 18 k first-pass
        // entries and ~170 us an utterance; 512 of them one after another were a quarter of the call)
        const int n_thr = (int)std::max(1u, std::min({ std::thread::hardware_concurrency(), 32u, (unsigned)((n_utt + 15) / 16) }));
        std::atomic<int> failed{0};                      // (an exception must not leave a worker: std::terminate would take the host process)
        auto work = [&](int t) {
            try {
                for (int u = t; u < n_utt; u += n_thr) {
                    const int nb = res1[(size_t)u * 8], nfr = res1[(size_t)u * 8 + 2];
                    const int32_t *cu = cols + (size_t)u * max_nb;
                    ff_build_vocab(m, cu, cu + (size_t)n_utt * max_nb, cu + (size_t)2 * n_utt * max_nb, nb, nfr, d.n1, voc[u]);
                }
            } catch (...) { failed.store(1); }
        };
        if (n_thr <= 1) work(0);
        else {
            std::vector<std::thread> thr;
            for (int t = 0; t < n_thr; ++t) thr.emplace_back(work, t);
            for (auto &t : thr) t.join();
        }
        if (failed.load()) { psgpu_set_error("psgpu_fwdflat_search: out of host memory building the utterances' vocabularies"); return PSGPU_ENOMEM; }
    }
    for (int u = 0; u < n_utt; ++u) {
        const int nfr = res1[(size_t)u * 8 + 2];
        const size_t C = (size_t)d.n1 + voc[u].n_chan, nwd = voc[u].wid.size(), cap = nwd + n_tail + 1;
        for (size_t k = 0; k < nwd; ++k)
            PSGPU_REQUIRE(voc[u].len[k] < 1024, "psgpu_fwdflat_search: a word chain of %d channels (FfUtt::einfo holds 10 bits)", voc[u].len[k]);
        PSGPU_REQUIRE(cap < (1u << 21), "psgpu_fwdflat_search: %zu active words (FfUtt::einfo holds 21 bits)", cap);
        slab_off[u + 1] = slab_off[u] + 4 * (nwd + 1) + C * (5 + 5 + 4 + 5 + 6) + 5 * (size_t)d.n_w + (nwd + 1) + 6 * cap + 2 * (cap + 1) + 16
                        + (raw ? (size_t)d.n_sen + (size_t)d.n_sen / 2 + 2 : 0);
        slab_off[u + 1] = (slab_off[u + 1] + 3) & ~(size_t)3;        // (an utterance's slab begins with its quads: 16-byte aligned)
        voc_off[u + 1] = voc_off[u] + 3 * nwd + (nwd + 1) + 2 * voc[u].node_sf.size() + (size_t)nfr + 2 + 4;
    }
#ifdef PSGPU_FT_PROFILE
    stamp();
#endif
    int32_t *slab = nullptr, *vdev = nullptr, *vhost = nullptr;
    FfOff *d_utts = nullptr, *ho = nullptr;
    if ((rcw = ff_work(m, 0, sizeof(int32_t) * slab_off[n_utt], (void **)&slab)) || (rcw = ff_work(m, 1, sizeof(int32_t) * voc_off[n_utt], (void **)&vdev))
        || (rcw = ff_work(m, 2, sizeof(FfOff) * n_utt, (void **)&d_utts)) || (rcw = ff_work(m, 4, sizeof(int32_t) * voc_off[n_utt], (void **)&vhost))
        || (rcw = ff_work(m, 5, sizeof(FfOff) * n_utt, (void **)&ho)))
        return rcw;
    hipError_t e = hipSuccess;
    // The launch order.  The launch ends with its slowest utterance, and an utterance runs a quarter faster once the workgroup it shares
    // a compute unit with has finished (77 k -> 60 k cycles a frame): the utterances with the most first-pass entries -- the larger
    // vocabularies, the busier frames -- go first, one a compute unit while there are compute units, and the second round pairs them with
    // the smallest ones (workgroups are handed out in index order, a free compute unit before a second workgroup on a busy one).
    // PSGPU_FF_ORDER=0: the caller's order.
    std::vector<int> order(n_utt);
    for (int i = 0; i < n_utt; ++i) order[i] = i;
    {
        static const int want = [] { const char *e = getenv("PSGPU_FF_ORDER"); return e ? atoi(e) : 1; }();
        int n_cu = 0;
        if (want && n_utt > 2 && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0) == hipSuccess && n_cu > 0 && n_utt > n_cu) {
            std::vector<int> by(n_utt);
            for (int i = 0; i < n_utt; ++i) by[i] = i;
            std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return res1[(size_t)a * 8] > res1[(size_t)b * 8]; });   // most entries first
            // round 0: the n_cu largest, descending; the rest ascending, so that position n_cu + k (the second workgroup of the k-th compute
            // unit) is the k-th smallest
            for (int k = 0; k < n_cu; ++k) order[k] = by[k];
            for (int k = n_cu; k < n_utt; ++k) order[k] = by[n_utt - 1 - (k - n_cu)];
        }
    }
    for (int slot = 0; slot < n_utt; ++slot) {
        const int i = order[slot];
        FfUtt u;
        memset(&u, 0, sizeof u);
        const FfVocab &v = voc[i];
        const size_t nwd = v.wid.size(), C = (size_t)d.n1 + v.n_chan, cap = nwd + n_tail + 1;
        int32_t *vh = vhost + voc_off[i], *vd = vdev + voc_off[i];
        auto put = [&](const std::vector<int32_t> &a, size_t n) { const int32_t *r = vd; if (n) memcpy(vh, a.data(), sizeof(int32_t) * n); vh += n; vd += n; return r; };
        u.nwd = (int32_t)nwd; u.n_chan = v.n_chan; u.n_frame = res1[(size_t)i * 8 + 2]; u.awl_cap = (int32_t)cap;
        u.wl_wid = put(v.wid, nwd); u.wl_chain = put(v.chain, nwd); u.wl_len = put(v.len, nwd);
        u.wl_node_off = put(v.node_off, nwd + 1); u.node_sf = put(v.node_sf, v.node_sf.size());
        u.fr_off = put(v.fr_off, v.fr_off.size()); u.fr_words = put(v.fr_words, v.fr_words.size());
        int32_t *q = slab + slab_off[i];
        auto take = [&](size_t n) { int32_t *r = q; q += n; return r; };
        u.wstat = take(4 * (nwd + 1));
        u.score = take(C * 5); u.hist = take(C * 5); u.out = take(C); u.outh = take(C); u.best = take(C); u.frame = take(C);
        u.senid = take(C * 5); u.tmat = take(C); u.mpx = take(C); u.rcid = take(C); u.xflag = take(C); u.elist = take(C); u.einfo = take(C);
        u.wchain = take(d.n_w); u.wlen = take(d.n_w); u.wrcs = take(d.n_w); u.wseen = take(nwd + 1); u.word_active = take(d.n_w); u.word_lat_idx = take(d.n_w);
        u.awl[0] = take(3 * cap); u.awl[1] = take(3 * cap);
        u.cnt_a = take(cap + 1); u.cnt_b = take(cap + 1);
        u.nrow32 = raw ? take(d.n_sen) : nullptr;
        u.nrow = raw ? reinterpret_cast<int16_t *>(take((size_t)d.n_sen / 2 + 1)) : nullptr;
        FfOff &o = ho[slot];
        o.utt = i; o.pad_ = 0;
#define X(f) o.f = u.f - slab;
        FF_SLAB_FIELDS(X)
#undef X
#define X(f) o.f = u.f - vdev;
        FF_VOC_FIELDS(X)
#undef X
        o.awl0 = u.awl[0] - slab; o.awl1 = u.awl[1] - slab;
        o.nrow32 = raw ? u.nrow32 - slab : -1; o.nrow = raw ? reinterpret_cast<int32_t *>(u.nrow) - slab : -1;
        o.nwd = u.nwd; o.n_chan = u.n_chan; o.n_frame = u.n_frame; o.awl_cap = u.awl_cap;
    }
    PSGPU_HIP(hipMemcpyAsync(vdev, vhost, sizeof(int32_t) * voc_off[n_utt], hipMemcpyHostToDevice, st));
    PSGPU_HIP(hipMemcpyAsync(d_utts, ho, sizeof(FfOff) * n_utt, hipMemcpyHostToDevice, st));
#ifdef PSGPU_FT_PROFILE
    stamp();
#endif
    FfRaw rw;
    memset(&rw, 0, sizeof rw);
    if (raw) rw = *raw;
    FfBufs bf;
    bf.slab = slab; bf.voc = vdev; bf.bp = bp_dev; bf.bss = bss_dev; bf.idx = idx_dev; bf.step = step_dev; bf.res = result_dev;
    bf.w1_ssid = w1_ssid_dev; bf.bp_cap = bp_cap; bf.bss_cap = bss_cap; bf.max_frames = max_frames;
    bf.prof = nullptr;
#ifdef PSGPU_FT_PROFILE
    if (hipMalloc((void **)&bf.prof, sizeof(long long) * 48 * (size_t)n_utt) != hipSuccess) bf.prof = nullptr;
#endif
    // scoring mode: the frame's senone scores live in LDS ([n_sen] int16, dynamic) beside ~53 KB of static arrays
    const size_t dyn = raw ? (((size_t)d.n_sen * 2 + 15) & ~(size_t)15) : 0;
#if defined(__HIPCC__)
#define FF_DYN_LDS(NE)                                                                                                     \
    if (dyn + 56 * 1024 > 65536) {                                                                                         \
        static const hipError_t attr_rc = hipFuncSetAttribute((const void *)fwdflat_kernel<NE, true>,                      \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, kFfMaxSen * 2);  \
        PSGPU_HIP(attr_rc);                                                                                                \
    }
#else
#define FF_DYN_LDS(NE)
#endif
    if (d.n_emit == 3 && raw) {
        FF_DYN_LDS(3)
        hipLaunchKernelGGL((fwdflat_kernel<3, true>), dim3(n_utt), dim3(kFfThreads), dyn, st, d, d_utts, bf, senscr_dev, scr_stride, utt_off_dev, rw);
    }
    else if (d.n_emit == 3)
        hipLaunchKernelGGL((fwdflat_kernel<3, false>), dim3(n_utt), dim3(kFfThreads), 0, st, d, d_utts, bf, senscr_dev, scr_stride, utt_off_dev, rw);
    else if (raw) {
        FF_DYN_LDS(5)
        hipLaunchKernelGGL((fwdflat_kernel<5, true>), dim3(n_utt), dim3(kFfThreads), dyn, st, d, d_utts, bf, senscr_dev, scr_stride, utt_off_dev, rw);
    }
    else
        hipLaunchKernelGGL((fwdflat_kernel<5, false>), dim3(n_utt), dim3(kFfThreads), 0, st, d, d_utts, bf, senscr_dev, scr_stride, utt_off_dev, rw);
#undef FF_DYN_LDS
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);          // (this entry is synchronous: the pinned staging buffers are the next call's)
#ifdef PSGPU_FT_PROFILE
    stamp();
    fprintf(stderr, "fwdflat host: first-pass columns to the host %.2f ms, vocabularies %.2f ms, allocations + tables to the device %.2f ms, kernel %.2f ms\n",
            t_host[1] - t_host[0], t_host[2] - t_host[1], t_host[3] - t_host[2], t_host[4] - t_host[3]);
    if (e == hipSuccess && bf.prof) {    // a profiling build: per-phase cycle counts of work-item 0, averaged over the utterances, per frame
        static const char *const names[13] = { "rest of: senones of the active channels (bitmap, codebooks)", "rest of: top-N lists (normalise, pack)",
            "senone evaluation + normaliser", "mark, renormalise, reset", "evaluate (gather + hmm_vit_eval)", "rest of: prune + exits (the exits' back-pointers)",
            "rest of: word transitions (fillers, clear)", "next active word list", "active channels gathered + senones marked", "top-N lists taken / evaluated",
            "prune: decisions", "prune: fan-outs + clears", "word transitions: exits scanned, successors entered" };
        std::vector<long long> h((size_t)48 * n_utt);
        std::vector<int32_t> r((size_t)8 * n_utt);
        hipMemcpy(h.data(), bf.prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), result_dev, 4 * r.size(), hipMemcpyDeviceToHost);
        double frames = 0, tot = 0, acc[13] = {};
        for (int u = 0; u < n_utt; ++u) { frames += r[(size_t)u * 8 + 2]; for (int i = 0; i < 13; ++i) acc[i] += (double)h[(size_t)u * 48 + i]; }
        for (int i = 0; i < 13; ++i) tot += acc[i];
        fprintf(stderr, "fwdflat_kernel profile: %d utterances, %.0f frames, %.0f cycles per frame (work-item 0)\n", n_utt, frames, tot / (frames > 0 ? frames : 1));
        {
            double a[8] = {}, mx = 0, slow = 0, sum = 0;
            for (int u = 0; u < n_utt; ++u) {
                for (int i = 0; i < 7; ++i) a[i] += (double)h[(size_t)u * 48 + 16 + i];
                mx = std::max(mx, (double)h[(size_t)u * 48 + 22]);
                double t = 0; for (int i = 0; i < 13; ++i) t += (double)h[(size_t)u * 48 + i];
                slow = std::max(slow, t); sum += t;
            }
            fprintf(stderr, "  word transitions: %.1f %% of the frames have new entries: %.1f entries, %.1f words in the window, %.0f pairs each; most entries in a frame %.0f; "
                    "phase cycles per frame: %.0f in frames whose exits fit the queue, %.0f in those that overflow\n", 100.0 * a[0] / frames, a[1] / std::max(a[0], 1.0),
                    a[2] / std::max(a[0], 1.0), a[3] / std::max(a[0], 1.0), mx, a[4] / frames, a[5] / frames);
            fprintf(stderr, "  slowest utterance: %.0f cycles = %.2f x the mean\n", slow, slow / (sum / n_utt));
            double g3[3] = {}, ex[4] = {}, sub[4] = {};
            for (int u = 0; u < n_utt; ++u) {
                for (int i = 0; i < 3; ++i) g3[i] += (double)h[(size_t)u * 48 + 27 + i];
                for (int i = 0; i < 4; ++i) { ex[i] += (double)h[(size_t)u * 48 + 23 + i]; sub[i] += (double)h[(size_t)u * 48 + 32 + i]; }
            }
            fprintf(stderr, "  per frame: %.1f active words, %.1f channels in their chains, %.1f of them active\n", g3[0] / frames, g3[1] / frames, g3[2] / frames);
            fprintf(stderr, "  frames with more than 192 / 224 / 256 / 320 exits: %.2f / %.2f / %.2f / %.2f %%\n", 100 * ex[0] / frames, 100 * ex[1] / frames, 100 * ex[2] / frames, 100 * ex[3] / frames);
            {
                double z[4] = {};
                for (int u = 0; u < n_utt; ++u) for (int i = 0; i < 4; ++i) z[i] += (double)h[(size_t)u * 48 + 40 + i];
                fprintf(stderr, "  the frame's top, cycles since the last frame's end: loop head %.0f, counters reset %.0f, slice bounds there %.0f, before the barrier %.0f\n",
                        z[0] / frames, z[1] / frames, z[2] / frames, z[3] / frames);
            }
            {
                double z[4] = {};
                for (int u = 0; u < n_utt; ++u) for (int i = 0; i < 4; ++i) z[i] += (double)h[(size_t)u * 48 + 36 + i];
                fprintf(stderr, "  the exits' phase (LDS queue), cycles since the pruning's end, per frame: ranks %.0f, places %.0f, entries written (work-item 0) %.0f, barrier passed %.0f\n",
                        z[0] / frames, z[1] / frames, z[2] / frames, z[3] / frames);
            }
            fprintf(stderr, "  the gather, cycles since the frame's start (wavefront 0): top barrier passed %.0f, first list entry there %.0f, first stamp there %.0f, loop left %.0f\n",
                    sub[0] / frames, sub[1] / frames, sub[2] / frames, sub[3] / frames);
        }
        {
            double ov = 0, ne = 0, o256 = 0;
            for (int u = 0; u < n_utt; ++u) { ov += (double)h[(size_t)u * 48 + 13]; ne += (double)h[(size_t)u * 48 + 14]; o256 += (double)h[(size_t)u * 48 + 15]; }
            fprintf(stderr, "  exits queued per frame %.1f; frames whose exits exceed the LDS queue: %.1f %%, cycles from their decisions to their end: %.0f each\n", ne / (frames > 0 ? frames : 1), 100.0 * ov / (frames > 0 ? frames : 1), o256 / (ov > 0 ? ov : 1));
        }
        for (int i = 0; i < 13; ++i)
            fprintf(stderr, "  %2d %-60s %9.0f cycles/frame  %5.1f %%\n", i, names[i], acc[i] / (frames > 0 ? frames : 1), 100.0 * acc[i] / (tot > 0 ? tot : 1));
    }
    hipFree(bf.prof);
#endif
    PSGPU_HIP(e);
    return PSGPU_OK;
}

extern "C" int psgpu_fwdflat_search_dev(psgpu_fwdflat_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                                        const int32_t *utt_off_dev, int32_t n_utt, int32_t max_frames,
                                        int32_t bp1_cap, const int32_t *bp1_dev, const int32_t *result1_dev,
                                        const int32_t *w1_ssid_dev, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev,
                                        int32_t *bss_dev, int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, void *stream)
{
    PSGPU_REQUIRE(senscr_dev, "psgpu_fwdflat_search_dev: NULL score rows");
    return ff_search(m, senscr_dev, scr_stride, nullptr, utt_off_dev, n_utt, max_frames, bp1_cap, bp1_dev, result1_dev, w1_ssid_dev,
                     bp_cap, bss_cap, bp_dev, bss_dev, idx_dev, step_dev, result_dev, stream);
}

extern "C" int psgpu_fwdflat_search_feats_dev(psgpu_fwdflat_t *m, const psgpu_ptm_view_t *ptm, const float *feats_dev,
                                              const int32_t *topn_seed_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                              int32_t max_frames, int32_t bp1_cap, const int32_t *bp1_dev,
                                              const int32_t *result1_dev, const int32_t *w1_ssid_dev, int32_t bp_cap,
                                              int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev, int32_t *idx_dev,
                                              int32_t *step_dev, int32_t *result_dev, void *stream)
{
    PSGPU_REQUIRE(m && ptm && feats_dev && topn_seed_dev, "psgpu_fwdflat_search_feats_dev: NULL argument");
    PSGPU_REQUIRE(ptm->n_sen == m->d.n_sen && ptm->n_sen <= kFfMaxSen && ptm->n_mgau >= 1 && ptm->n_mgau <= kFfMaxCb &&
                  ptm->n_feat >= 1 && ptm->n_feat <= 16 && ptm->topn >= 1 && ptm->topn <= kFfMaxTopn &&
                  (int64_t)ptm->n_mgau * ptm->n_feat * ptm->topn <= kFfMaxEnt && ptm->logadd8_size <= 256 && ptm->n_density >= ptm->topn,
                  "psgpu_fwdflat_search_feats_dev: model shape outside this version (senones %d, codebooks %d, streams %d, top-N %d)",
                  ptm->n_sen, ptm->n_mgau, ptm->n_feat, ptm->topn);
    PSGPU_REQUIRE(ptm->mean && ptm->var && ptm->det && ptm->mixw && ptm->sen2cb && ptm->logadd8, "psgpu_fwdflat_search_feats_dev: NULL model table");
    FfRaw rw;
    memset(&rw, 0, sizeof rw);
    rw.pm = *ptm; rw.feats = feats_dev; rw.seed = topn_seed_dev;
    return ff_search(m, nullptr, 0, &rw, utt_off_dev, n_utt, max_frames, bp1_cap, bp1_dev, result1_dev, w1_ssid_dev,
                     bp_cap, bss_cap, bp_dev, bss_dev, idx_dev, step_dev, result_dev, stream);
}

extern "C" int psgpu_fwdflat_search_feats_lists_dev(psgpu_fwdflat_t *m, const psgpu_ptm_view_t *ptm, const float *feats_dev,
                                                    const int32_t *topn_seed_dev, const int32_t *topn_score_dev,
                                                    const uint8_t *topn_cw_dev, const uint8_t *open_flags_dev, int32_t total_frames,
                                                    const int32_t *utt_off_dev, int32_t n_utt,
                                                    int32_t max_frames, int32_t bp1_cap, const int32_t *bp1_dev,
                                                    const int32_t *result1_dev, const int32_t *w1_ssid_dev, int32_t bp_cap,
                                                    int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev, int32_t *idx_dev,
                                                    int32_t *step_dev, int32_t *result_dev, void *stream)
{
    PSGPU_REQUIRE(m && ptm && feats_dev && topn_seed_dev && topn_score_dev && topn_cw_dev && open_flags_dev,
                  "psgpu_fwdflat_search_feats_lists_dev: NULL argument");
    PSGPU_REQUIRE(ptm->n_sen == m->d.n_sen && ptm->n_sen <= kFfMaxSen && ptm->n_mgau >= 1 && ptm->n_mgau <= kFfMaxCb &&
                  ptm->n_feat >= 1 && ptm->n_feat <= 16 && ptm->topn == 4 &&
                  (int64_t)ptm->n_mgau * ptm->n_feat * ptm->topn <= kFfMaxEnt && ptm->logadd8_size <= 256 && ptm->n_density >= ptm->topn,
                  "psgpu_fwdflat_search_feats_lists_dev: model shape outside this version (top-4 lists)");
    PSGPU_REQUIRE(ptm->mean && ptm->var && ptm->det && ptm->mixw && ptm->sen2cb && ptm->logadd8, "psgpu_fwdflat_search_feats_lists_dev: NULL model table");
    PSGPU_REQUIRE(((uintptr_t)topn_score_dev & 15) == 0 && ((uintptr_t)topn_cw_dev & 3) == 0, "psgpu_fwdflat_search_feats_lists_dev: misaligned lists");
    FfRaw rw;
    memset(&rw, 0, sizeof rw);
    rw.pm = *ptm; rw.feats = feats_dev; rw.seed = topn_seed_dev;
    rw.tsc = topn_score_dev; rw.tcw = reinterpret_cast<const uint32_t *>(topn_cw_dev); rw.open = open_flags_dev; rw.total = total_frames;
    return ff_search(m, nullptr, 0, &rw, utt_off_dev, n_utt, max_frames, bp1_cap, bp1_dev, result1_dev, w1_ssid_dev,
                     bp_cap, bss_cap, bp_dev, bss_dev, idx_dev, step_dev, result_dev, stream);
}

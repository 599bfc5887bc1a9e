// psgpu_search.hip -- the lexicon-tree search on gfx950, first
// version: whole utterances, one workgroup per utterance, every frame inside the kernel.
//
// Replaces ngram_fwdtree_start + ngram_fwdtree_search x T + ngram_fwdtree_finish
// (reference src/ngram_search_fwdtree.c:469-520, 1452-1495, 1497-1533) and the
// back-pointer helpers they call (src/ngram_search.c:301-498, 583-674): evaluate_channels,
// histogram / beam pruning with phone and last-phone transitions, language-model scores at
// word entry, right-context channel allocation, save_bp, bptable_maxwpf, word_transition,
// deactivate_channels.  Output: the back-pointer table in the reference's own columns
// (bptbl_t, ngram_search.h:112-124), the right-context score stack and the per-frame marks.
//
// Inputs are what the other kernels leave on the device: per-frame senone scores
// (normalised rows here; the un-normalised rows + active-list normaliser come next) and
// the phone-loop penalties.  Static tables are the reference's own (tree, dictionary,
// dict2pid, beams) flattened to index arrays; the language model is a dense table over
// dictionary word ids, so this version is for small vocabularies (turtle, tidigits).
//
// Parallelism in this version: utterances across workgroups; inside a frame the HMM
// evaluation and the tree pruning run across the 256 threads (the pruning in the
// order-free per-node formulation that oracle/ps_oracle_search.c proves equivalent to the
// reference's sequential walk: decisions on a snapshot, prefix sums for list positions);
// the word-level bookkeeping (tens of items per frame) is still one thread.  Results are
// the reference's, bit for bit (tests/test_search_gpu.py against reference dumps).
//
// Two formulations of the per-frame passes (psgpu_fwdtree_set_mode):
//   PER_NODE    (default) the pruning decides every node of the tree (work ~ tree size), 256 work-items;
//   ACTIVE_LIST the pruning visits only the roots, the listed nodes and their children (work ~ active
//               channels, oracle prune_tree_list), the word-level positions come from workgroup prefix sums
//               instead of single-thread loops, and a large tree gets a 1024-work-item workgroup: the form
//               for large vocabularies.  Same tables, bit for bit.
#include "psgpu_hmm_dev.h"
#include "psgpu_lm_dev.h"
#include <algorithm>
#include <cstring>
#include <vector>

constexpr int kFtThreads = 256;        // work-items per utterance (PER_NODE, and ACTIVE_LIST on small trees)
constexpr int kFtThreadsBig = 1024;    // ACTIVE_LIST on trees beyond the LDS scratch
constexpr int kFtMaxN = 4096;          // tree nodes (LDS scratch of the pruning)
constexpr int kFtMaxCi = 64;
constexpr int kFtMaxSen = 8192;        // senones (LDS bitmap of the active list, raw-score mode)

struct FtDev {
    int32_t n_ci, n_emit, n_sen, n_w, R, M, N, n1, n1lm, TOT;
    int32_t beam, pbeam, lpbeam, lponlybeam, wbeam, pip, nwpen, silpen, fillpen, maxhmmpf, maxwpf;
    int32_t startwid, finishwid, silwid, filler_start, filler_end, sil_ci, has_pl;
    const int32_t *node_ci, *node_ci2, *node_ssid, *node_tmat, *node_child, *node_sib, *node_pw, *parent;
    const int32_t *homophone, *w1_wid, *w1_ci, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx, *w1_of_word;
    const int32_t *d_pronlen, *d_first, *d_last, *d_last2, *d_base, *d_filler;
    const int32_t *rs_n, *rs_ssid, *rs_cimap, *ldiph, *ci_tmat, *lm, *wc_off;
    const uint8_t *tp;
    const uint16_t *sseq;
    int32_t big;                         // tree or vocabulary beyond the LDS scratch: list / word scratch in the utterance's slab
    int32_t use_trie;                    // language scores from the trie (psgpu_fwdtree_set_lm) instead of the dense table
    int32_t list_mode;                   // PSGPU_FWDTREE_ACTIVE_LIST: per-frame work proportional to the active channels
    int32_t *w1_out;                     // optional [n_utt][n1][n_emit]: the single-phone channels' ssids when the pass ends
    LmDev trie;
};

// per-utterance state (one slab per utterance; all int32 unless noted)
struct FtUtt {
    // channels: [0, N) tree nodes, [N, N + n1) single-phone words, [N + n1, N + n1 + TOT) last-phone slots
    int32_t *score, *hist;               // [C][5]
    int32_t *out, *outh, *best, *frame;  // [C]
    int32_t *senid;                      // [C][5]  senone ids, or per-state ssids of multiplex HMMs
    int32_t *tmat, *mpx;                 // [C]
    int32_t *present;                    // [TOT]
    int32_t *acl[2], *awl[2];            // [N], [n_w]
    int32_t *word_active, *word_lat_idx; // [n_w]
    int32_t *cand_wid, *cand_score, *cand_bp, *cand_next;   // [n_w + 1]
    int32_t *lt_sf, *lt_dscr, *lt_bp;    // [n_w]
    int32_t *csf_ef, *csf_cand;          // [n_w + 1]
    int32_t *bp;                         // [10][bp_cap] columns: frame valid wid bp score s_idx real_wid prev_real_wid last last2
    int32_t *bss;                        // [bss_cap]
    int32_t *bp_table_idx;               // [T + 2]
    int32_t *o_frame, *o_s0, *o_best, *o_out, *o_outh, *pos, *flag;   // [N] pruning snapshot / decisions
    int32_t *step;                       // [T][4] best_score, last_phone_best_score, bpidx, n_active_chan (diagnostics)
    int32_t *result;                     // [8] bpidx, bss_head, n_frame, status
    int32_t *g_cnt, *g_w;                // [max(R + N, n_w) + 1], [4][n_w]: scratch for large trees / vocabularies (FtDev.big)
    int16_t *nrow;                       // [n_sen] the frame's normalised scores (raw-score mode)
    int32_t *cand_mark;                  // [n_w] frame in which the word was last a last-phone candidate (ACTIVE_LIST)
    int32_t *xlist, *xslot;              // [TOT] (candidate, slot) pairs of the entering loop (ACTIVE_LIST)
    int32_t *elist, *eword;              // [TOT] the frame's present last-phone channels and the index of their word in the
                                         //       active word list (ACTIVE_LIST: evaluation / pruning work list)
    int32_t bp_cap, bss_cap;
};

// ACTIVE_LIST is handed the per-utterance fields as offsets (int32 units) from buffers that are kernel arguments: a pointer
// loaded from memory is generic to the compiler (every access a flat_load / flat_store, both wait counters), a pointer formed
// from a kernel argument is global.  (The default formulation keeps reading FtUtt, as measured.)
#define FT_SLAB_FIELDS(X) X(score) X(hist) X(out) X(outh) X(best) X(frame) X(senid) X(tmat) X(mpx) X(present) X(word_active) \
    X(word_lat_idx) X(cand_wid) X(cand_score) X(cand_bp) X(cand_next) X(lt_sf) X(lt_dscr) X(lt_bp) X(csf_ef) X(csf_cand) \
    X(o_frame) X(o_s0) X(o_best) X(o_out) X(o_outh) X(pos) X(flag) X(cand_mark) X(elist) X(eword) X(xlist) X(xslot)
struct FtOff {
#define X(f) int64_t f;
    FT_SLAB_FIELDS(X)
#undef X
    int64_t acl0, acl1, awl0, awl1, g_cnt, g_w, nrow;
};
struct FtBufs {
    int32_t *slab, *bp, *bss, *idx, *step, *res;
    int32_t bp_cap, bss_cap, max_frames;
};

struct psgpu_fwdtree_s {
    FtDev d;
    std::vector<void *> allocs;
    int32_t C;
};

#define BPC(u, col, i) ((u).bp[(size_t)(col) * (u).bp_cap + (i)])
enum { B_FRAME, B_VALID, B_WID, B_BP, B_SCORE, B_SIDX, B_REAL, B_PREAL, B_LAST, B_LAST2 };

template <int CS, int C1>
__device__ __forceinline__ void ch_clear(const FtDev &p, FtUtt &u, int c)      // hmm_clear, hmm.c:181-196
{
    for (int i = 0; i < p.n_emit; ++i) { u.score[c * CS + i] = kW; u.hist[c * CS + i] = -1; }
    u.out[(c) * C1] = kW; u.outh[(c) * C1] = -1; u.best[(c) * C1] = kW; u.frame[(c) * C1] = -1;
}
template <int CS, int C1>
__device__ __forceinline__ void ch_init(const FtDev &p, FtUtt &u, int c, int mpx, int ssid, int tmatid)   // hmm_init :146-168
{
    u.mpx[(c) * C1] = mpx; u.tmat[(c) * C1] = tmatid;
    if (mpx) {
        u.senid[c * CS] = ssid;
        for (int i = 1; i < p.n_emit; ++i) u.senid[c * CS + i] = kBadSsid;
    }
    else
        for (int i = 0; i < p.n_emit; ++i) u.senid[c * CS + i] = p.sseq[(size_t)ssid * p.n_emit + i];
    ch_clear<CS, C1>(p, u, c);
}
template <int CS, int C1>
__device__ __forceinline__ void ch_enter(FtUtt &u, int c, int32_t score, int32_t hist, int frame)   // hmm_enter :198-204
{
    u.score[c * CS] = score; u.hist[c * CS] = hist; u.frame[(c) * C1] = frame;
}
template <int CS, int C1>
__device__ __forceinline__ void ch_normalize(const FtDev &p, FtUtt &u, int c, int32_t norm)      // hmm_normalize :206-217
{
    for (int i = 0; i < p.n_emit; ++i) if (u.score[c * CS + i] > kW) u.score[c * CS + i] -= norm;
    if (u.out[(c) * C1] > kW) u.out[(c) * C1] -= norm;
}

// hmm_vit_eval on channel c with the frame's score row
template <int NE, int CS, int C1>
__device__ __forceinline__ int32_t ch_eval(const FtDev &p, FtUtt &u, int c, const int16_t *row)
{
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        h.score[i] = i < NE ? u.score[c * CS + i] : kW;
        h.history[i] = i < NE ? u.hist[c * CS + i] : -1;
        h.senid[i] = i < NE ? (uint16_t)u.senid[c * CS + i] : 0;
    }
    h.out_score = u.out[(c) * C1]; h.out_history = u.outh[(c) * C1]; h.bestscore = u.best[(c) * C1];
    const uint8_t *tp = p.tp + (size_t)u.tmat[(c) * C1] * NE * (NE + 1);
    int32_t b;
    if (NE == 3) b = u.mpx[(c) * C1] ? vit3_mpx(h, tp, row, p.sseq) : vit3(h, tp, row);
    else         b = u.mpx[(c) * C1] ? vit5_mpx(h, tp, row, p.sseq) : vit5(h, tp, row);
#pragma unroll
    for (int i = 0; i < NE; ++i) { u.score[c * CS + i] = h.score[i]; u.hist[c * CS + i] = h.history[i]; u.senid[c * CS + i] = h.senid[i]; }
    u.out[(c) * C1] = h.out_score; u.outh[(c) * C1] = h.out_history; u.best[(c) * C1] = h.bestscore;
    return b;
}

__device__ __forceinline__ int32_t ft_pen(const FtDev &p, const int32_t *pp, int ci) { return p.has_pl ? pp[ci] : 0; }
__device__ __forceinline__ int32_t ft_lm(const FtDev &p, int w3, int w2, int w1)
{
    if (p.use_trie) {                    // ngram_tg_score(...) >> SENSCR_SHIFT, ngram_search_fwdtree.c:1118, :1342
        int nu;
        return lm_tg_score(p.trie, w3, w2, w1, nu) >> 10;
    }
    const size_t n1 = (size_t)p.n_w + 1;
    return p.lm[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}
// ngram_search_exit_score, ngram_search.c:653-674
__device__ __forceinline__ int32_t ft_exit_score(const FtDev &p, const FtUtt &u, int bp, int rcphone)
{
    const int l2 = BPC(u, B_LAST2, bp);
    if (l2 == -1) return BPC(u, B_SCORE, bp);
    const int l1 = BPC(u, B_LAST, bp);
    return u.bss[BPC(u, B_SIDX, bp) + p.rs_cimap[((size_t)l1 * p.n_ci + l2) * p.n_ci + rcphone]];
}
// set_real_wid, ngram_search.c:341-372
__device__ __forceinline__ void ft_set_real_wid(const FtDev &p, FtUtt &u, int bp)
{
    const int prev = BPC(u, B_BP, bp), wid = BPC(u, B_WID, bp);
    if (p.d_filler[wid]) {
        if (prev != -1) { BPC(u, B_REAL, bp) = BPC(u, B_REAL, prev); BPC(u, B_PREAL, bp) = BPC(u, B_PREAL, prev); }
        else { BPC(u, B_REAL, bp) = p.d_base[wid]; BPC(u, B_PREAL, bp) = -1; }
    }
    else {
        BPC(u, B_REAL, bp) = p.d_base[wid];
        BPC(u, B_PREAL, bp) = prev != -1 ? BPC(u, B_REAL, prev) : -1;
    }
}
// ngram_search_save_bp, ngram_search.c:376-498 (single thread).  Returns false when a table is full.
__device__ __forceinline__ bool ft_save_bp(const FtDev &p, FtUtt &u, int32_t &bpidx, int32_t &bss_head, int frame, int w, int32_t score,
                           int32_t path, int rc)
{
    const int bp = u.word_lat_idx[w];
    if (bp != -1) {
        if (BPC(u, B_SCORE, bp) < score) {
            const int ob = BPC(u, B_BP, bp);
            if (ob != path) {
                const int32_t b0 = ob == -1 ? -1 : BPC(u, B_PREAL, ob), b1 = ob == -1 ? -1 : BPC(u, B_REAL, ob);
                const int32_t n0 = path == -1 ? -1 : BPC(u, B_PREAL, path), n1 = path == -1 ? -1 : BPC(u, B_REAL, path);
                if (b0 != n0 || b1 != n1) ft_set_real_wid(p, u, bp);      // with the old bp still in place, as the reference
                BPC(u, B_BP, bp) = path;
            }
            BPC(u, B_SCORE, bp) = score;
        }
        if (BPC(u, B_SIDX, bp) != -1) u.bss[BPC(u, B_SIDX, bp) + rc] = score;
        return true;
    }
    if (bpidx >= u.bp_cap || bss_head + p.n_ci >= u.bss_cap) return false;
    u.word_lat_idx[w] = bpidx;
    BPC(u, B_WID, bpidx) = w; BPC(u, B_FRAME, bpidx) = frame; BPC(u, B_BP, bpidx) = path; BPC(u, B_SCORE, bpidx) = score;
    BPC(u, B_SIDX, bpidx) = bss_head; BPC(u, B_VALID, bpidx) = 1;
    BPC(u, B_LAST, bpidx) = p.d_last[w];
    int rcsize = 0;
    if (p.d_pronlen[w] == 1) { BPC(u, B_LAST2, bpidx) = -1; BPC(u, B_SIDX, bpidx) = -1; }
    else {
        BPC(u, B_LAST2, bpidx) = p.d_last2[w];
        rcsize = p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]];
    }
    for (int i = 0; i < rcsize; ++i) u.bss[bss_head + i] = kW;
    if (rcsize) u.bss[bss_head + rc] = score;
    ft_set_real_wid(p, u, bpidx);
    ++bpidx;
    bss_head += rcsize;
    return true;
}

// Exclusive prefix sum of a[0..n) in place by the whole workgroup (a in LDS, written before a barrier);
// returns the total to every thread.  tmp: NT / 64 words of LDS.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ int32_t ft_block_scan(int32_t *a, int n, int32_t *tmp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (n + NT - 1) / NT;
    const int b = min(n, tid * per), e = min(n, b + per);
    int32_t sum = 0;
    for (int i = b; i < e; ++i) sum += a[i];
    int32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) tmp[tid >> 6] = incl;
    __syncthreads();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int32_t t = tmp[w]; total += t; if (w < (tid >> 6)) base += t; }
    int32_t run = base + incl - sum;
    for (int i = b; i < e; ++i) { const int32_t k = a[i]; a[i] = run; run += k; }
    __syncthreads();
    return total;
}
// out[t] = max of v over threads 0 .. t - 1 (-1 for thread 0): one value per thread.  Ends with a barrier.
__device__ __forceinline__ int32_t ft_block_excl_max(int32_t v, int32_t *tmp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    int32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(incl, d); if (lane >= d) incl = max(incl, o); }
    if (lane == 63) tmp[tid >> 6] = incl;
    int32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = -1;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) excl = max(excl, tmp[w]);
    __syncthreads();
    return excl;
}

template <int NE, int NT, bool LIST>
__global__ __launch_bounds__(NT)
void fwdtree_kernel(FtDev p, const FtUtt *__restrict__ utts, const int16_t *__restrict__ senscr, int64_t scr_stride,
                    const int32_t *__restrict__ penalties, const int32_t *__restrict__ utt_off, int32_t raw_mode,
                    int32_t pl_window, const FtOff *__restrict__ offs, FtBufs bf)
{
    __shared__ uint32_t s_bits[kFtMaxSen / 32];
    __shared__ int32_t s_prev[kFtMaxSen / 32];
    __shared__ int32_t s_nb;
    __shared__ int32_t s_cnt[kFtMaxN + 1];
    __shared__ int32_t s_red[8];
    __shared__ int32_t s_scan[NT / 64];
    __shared__ int32_t s_bins[256];
    __shared__ int32_t s_sc[8];          // best_score, lpbest, dynamic_beam, bpidx, bss_head, n_cand, status, n_frame
    __shared__ unsigned long long s_evals;
    __shared__ int32_t s_nwc, s_nwc2;    // ACTIVE_LIST: lengths of the word level's evaluation list and of the entering list
    // channel state: PER_NODE keeps one array per field ([C][5] / [C]); ACTIVE_LIST one record of CS ints per channel (score,
    // hist, out, outh, best, frame, senid, tmat, mpx: 64 bytes for 3-state models) -- its passes gather by channel id, and a
    // record is one cache line where the arrays are nine
    constexpr int CS = LIST ? (NE == 3 ? 16 : 24) : 5, C1 = LIST ? CS : 1;
    const int tid = threadIdx.x;
    FtUtt u;
    if (LIST) {
        const FtOff o = offs[blockIdx.x];
#define X(f) u.f = bf.slab + o.f;
        FT_SLAB_FIELDS(X)
#undef X
        u.acl[0] = bf.slab + o.acl0; u.acl[1] = bf.slab + o.acl1; u.awl[0] = bf.slab + o.awl0; u.awl[1] = bf.slab + o.awl1;
        u.g_cnt = bf.slab + o.g_cnt; u.g_w = bf.slab + o.g_w;           // (offset 0 when unused: never dereferenced then)
        u.nrow = reinterpret_cast<int16_t *>(bf.slab + o.nrow);
        u.bp = bf.bp + (size_t)blockIdx.x * 10 * bf.bp_cap; u.bss = bf.bss + (size_t)blockIdx.x * bf.bss_cap;
        u.bp_table_idx = bf.idx + (size_t)blockIdx.x * (bf.max_frames + 2); u.step = bf.step + (size_t)blockIdx.x * bf.max_frames * 4;
        u.result = bf.res + (size_t)blockIdx.x * 8;
        u.bp_cap = bf.bp_cap; u.bss_cap = bf.bss_cap;
    }
    else
        u = utts[blockIdx.x];
    // list-position / candidate / word scratch: LDS when the tree and the vocabulary fit (kFtMaxN entries), else the slab
    int32_t *const cnt = p.big ? u.g_cnt : s_cnt;
    const int t0 = utt_off[blockIdx.x], T = utt_off[blockIdx.x + 1] - t0;
    const int N = p.N, R = p.R, W1 = N, WC = N + p.n1;
    int n_acl[2] = {0, 0}, n_awl[2] = {0, 0};           // uniform copies (every thread tracks them identically)

    // ---- hmm_init of every permanent channel, ngram_fwdtree_start (:469-520)
    for (int c = tid; c < N; c += NT) ch_init<CS, C1>(p, u, c, c < R, p.node_ssid[c], p.node_tmat[c]);
    for (int i = tid; i < p.n1; i += NT) ch_init<CS, C1>(p, u, W1 + i, p.w1_mpx[i], p.w1_ssid[i], p.w1_tmat[i]);
    for (int i = tid; i < p.TOT; i += NT) u.present[i] = 0;
    for (int w = tid; w < p.n_w; w += NT) { u.word_lat_idx[w] = -1; u.lt_sf[w] = -1; u.word_active[w] = 0; }
    if (LIST) {                                  // pos is kept at -1 between frames; no word has been a candidate yet
        for (int c = tid; c < N; c += NT) u.pos[c] = -1;
        for (int w = tid; w < p.n_w; w += NT) u.cand_mark[w] = -1;
    }
    if (tid == 0) {
        s_sc[0] = 0; s_sc[1] = 0; s_sc[2] = p.beam; s_sc[3] = 0; s_sc[4] = 0; s_sc[5] = 0; s_sc[6] = 0; s_sc[7] = 0;
        s_evals = 0ull;
    }
    __syncthreads();
    if (tid == 0) ch_enter<CS, C1>(u, W1 + p.w1_of_word[p.startwid], 0, -1, 0);
    __syncthreads();

    for (int f = 0; f < T; ++f) {
        const int cur = f & 1, nxt = cur ^ 1, nf = f + 1;
        const int16_t *row = senscr + (size_t)(t0 + f) * scr_stride;
        // raw mode: the phone loop runs pl_window frames ahead and stops at the last frame
        const int32_t *pp = penalties + (size_t)(t0 + (raw_mode ? min(f + pl_window, T - 1) : f)) * p.n_ci;
        if (raw_mode) {
            // ---- compute_sen_active (:526-564) + acmod_flags2list (acmod.c:1223-1275) + the scorer's
            //      active-list normalisation (ptm_mgau.c:393-400) on un-normalised rows: the frame's scores
            //      are raw - min over the listed senones, bridging entries included
            const int nwords = (p.n_sen + 31) >> 5;
            for (int i = tid; i < nwords; i += NT) s_bits[i] = 0u;
            if (tid == 0) s_nb = 0x7fffffff;
            __syncthreads();
            auto mark = [&](int c) {
                for (int k = 0; k < NE; ++k) {
                    int sen = u.senid[c * CS + k];
                    if (u.mpx[(c) * C1]) { if (sen == kBadSsid) continue; sen = p.sseq[(size_t)sen * NE + k]; }
                    atomicOr(&s_bits[sen >> 5], 1u << (sen & 31));
                }
            };
            for (int i = tid; i < R; i += NT) if (u.frame[(i) * C1] == f) mark(i);
            for (int i = tid; i < n_acl[cur]; i += NT) mark(u.acl[cur][i]);
            for (int i = tid; i < n_awl[cur]; i += NT) {
                const int w = u.awl[cur][i];
                for (int k = p.wc_off[w]; k < p.wc_off[w + 1]; ++k) if (u.present[k]) mark(WC + k);
            }
            for (int i = tid; i < p.n1; i += NT) if (u.frame[(W1 + i) * C1] == f) mark(W1 + i);
            __syncthreads();
            {   // s_prev[w] = the highest senone listed in the words before w (one bitmap word per thread)
                static_assert(kFtMaxSen / 32 <= NT, "one bitmap word per thread");
                const uint32_t bw = tid < nwords ? s_bits[tid] : 0u;
                const int32_t pv = ft_block_excl_max(bw ? tid * 32 + 31 - __clz((int)bw) : -1, s_scan);
                if (tid < nwords) s_prev[tid] = pv;
            }
            __syncthreads();
            int32_t mn = 0x7fffffff;
            for (int w = tid; w < nwords; w += NT) {
                uint32_t b = s_bits[w];
                int prev = s_prev[w];
                while (b) {
                    const int sen = w * 32 + __ffs((int)b) - 1;
                    b &= b - 1;
                    for (int last = prev < 0 ? 0 : prev; sen - last > 255;) { last += 255; mn = min(mn, (int32_t)row[last]); }
                    mn = min(mn, (int32_t)row[sen]);
                    prev = sen;
                }
            }
            atomicMin(&s_nb, mn);
            __syncthreads();
            const int32_t nb = s_nb;
            for (int w = tid; w < nwords; w += NT) {
                uint32_t b = s_bits[w];
                while (b) {
                    const int sen = w * 32 + __ffs((int)b) - 1;
                    b &= b - 1;
                    u.nrow[sen] = (int16_t)(uint16_t)((uint32_t)(int32_t)row[sen] - (uint32_t)nb);
                }
            }
            __syncthreads();
            row = u.nrow;
        }
        // ---- ngram_search_mark_bptable, failure test, renormalisation (:1467-1480)
        if (tid == 0) u.bp_table_idx[f] = s_sc[3];
        const int32_t best_in = s_sc[0];
        if (best_in == kW || best_in < kW) break;
        if (best_in + 2 * p.beam < kW) {                      // renormalize_scores (:566-603)
            for (int i = tid; i < R; i += NT) if (u.frame[(i) * C1] == f) ch_normalize<CS, C1>(p, u, i, best_in);
            for (int i = tid; i < n_acl[cur]; i += NT) ch_normalize<CS, C1>(p, u, u.acl[cur][i], best_in);
            for (int i = tid; i < n_awl[cur]; i += NT) {
                const int w = u.awl[cur][i];
                for (int k = p.wc_off[w]; k < p.wc_off[w + 1]; ++k) if (u.present[k]) ch_normalize<CS, C1>(p, u, WC + k, best_in);
            }
            for (int i = tid; i < p.n1; i += NT) if (u.frame[(W1 + i) * C1] == f) ch_normalize<CS, C1>(p, u, W1 + i, best_in);
        }
        if (tid < 8) s_red[tid] = kW;
        if (LIST && tid == 0) s_nwc = 0;
        __syncthreads();
        if (LIST) {
            // a word near its end has its whole right-context fan-out (20-40 channels) present at once: the word level's
            // channels are gathered into one list first (order irrelevant: independent evaluations, a maximum and a count)
            // and evaluated one work-item per channel below
            for (int i = tid; i < n_awl[cur]; i += NT) {
                const int w = u.awl[cur][i];
                u.word_active[w] = 0;
                for (int k = p.wc_off[w]; k < p.wc_off[w + 1]; ++k)
                    if (u.present[k]) { const int q = atomicAdd(&s_nwc, 1); u.elist[q] = WC + k; u.eword[q] = i; }
            }
            __syncthreads();
        }
        // ---- evaluate_channels (:605-715): s_red[0] roots, [1] tree, [2] word level; [3..5] counts
        {
            int32_t b0 = kW, b1 = kW, b2 = kW; int n0 = 0, n2 = 0;
            for (int i = tid; i < R; i += NT)
                if (u.frame[(i) * C1] == f) { b0 = max(b0, ch_eval<NE, CS, C1>(p, u, i, row)); ++n0; }
            for (int i = tid; i < n_acl[cur]; i += NT) b1 = max(b1, ch_eval<NE, CS, C1>(p, u, u.acl[cur][i], row));
            if (LIST)
                for (int i = tid; i < s_nwc; i += NT) { b2 = max(b2, ch_eval<NE, CS, C1>(p, u, u.elist[i], row)); ++n2; }
            else
            for (int i = tid; i < n_awl[cur]; i += NT) {
                const int w = u.awl[cur][i];
                u.word_active[w] = 0;
                for (int k = p.wc_off[w]; k < p.wc_off[w + 1]; ++k)
                    if (u.present[k]) { b2 = max(b2, ch_eval<NE, CS, C1>(p, u, WC + k, row)); ++n2; }
            }
            for (int i = tid; i < p.n1; i += NT) {
                if (u.frame[(W1 + i) * C1] < f) continue;
                const int32_t sc = ch_eval<NE, CS, C1>(p, u, W1 + i, row);
                if (p.w1_wid[i] != p.finishwid) b2 = max(b2, sc);
                ++n2;
            }
            atomicMax(&s_red[0], b0); atomicMax(&s_red[1], b1); atomicMax(&s_red[2], b2);
            if (n0) atomicAdd(&s_red[3], n0 - 0);            // (s_red[3..4] start at kW: corrected below)
            if (n2) atomicAdd(&s_red[4], n2);
        }
        __syncthreads();
        if (tid == 0) {
            const int32_t bs = max(max(s_red[0], s_red[1]), s_red[2]);
            s_sc[0] = bs; s_sc[1] = s_red[2];
            s_evals += (unsigned long long)((s_red[3] - kW) + n_acl[cur] + (s_red[4] - kW));
            s_sc[5] = 0;                                        // n_lastphn_cand
            // dynamic beam (:1133-1181)
            s_sc[2] = p.beam;
        }
        for (int i = tid; i < 256; i += NT) s_bins[i] = 0;
        __syncthreads();
        const int32_t best_score = s_sc[0];
        if (p.maxhmmpf != -1 && s_evals > (unsigned long long)p.maxhmmpf) {
            const int32_t bw = -p.beam / 256;
            for (int i = tid; i < R + n_acl[cur]; i += NT) {
                const int c = i < R ? i : u.acl[cur][i - R];
                int32_t b = (best_score - u.best[(c) * C1]) / bw;
                if (b >= 256) b = 255;
                atomicAdd(&s_bins[b], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int i, nh = 0;
                for (i = 0; i < 256; ++i) { nh += s_bins[i]; if (nh > p.maxhmmpf) break; }
                s_sc[2] = -(i * bw);
            }
            __syncthreads();
        }
        const int32_t thresh = best_score + s_sc[2];
        const int32_t npt = best_score + p.pbeam, lpt = best_score + p.lpbeam;

        // ---- prune_root_chan + prune_nonroot_chan (:722-877), order-free formulation
        if (LIST) {
            // work proportional to the active channels (oracle prune_tree_list): the items are the roots, the listed
            // nodes and their children.  Reads of another node's state go to the snapshot (o_out, o_outh, flag, pos) of
            // a root or listed node, writes to the item's own channel and decision word, so the items are independent.
            const int na = n_acl[cur];
            for (int q = tid; q < na; q += NT) u.pos[u.acl[cur][q]] = q;
            for (int i = tid; i < R + na; i += NT) {
                const int node = i < R ? i : u.acl[cur][i - R];
                const bool active = i < R ? u.frame[(node) * C1] >= f : true;
                u.o_out[node] = u.out[(node) * C1]; u.o_outh[node] = u.outh[(node) * C1];
                u.flag[node] = (active && u.best[(node) * C1] > thresh) ? 1 : 0;
            }
            __syncthreads();
            auto decide = [&](int c) {
                const int P = p.parent[c], pc = u.pos[c];
                const bool in_acl = pc >= 0, par_active = P < R || u.pos[P] >= 0;
                const int32_t news = (par_active ? u.o_out[P] : kW) + p.pip;
                const bool parent_can = par_active && (u.flag[P] & 1) && (p.has_pl || news > npt)
                                        && (news + ft_pen(p, pp, p.node_ci[c]) > npt);
                const bool parent_first = P < R || !in_acl || u.pos[P] < pc;
                const bool retc = in_acl && (u.flag[c] & 1);
                bool fire;
                if (!in_acl || parent_first) fire = parent_can && (u.frame[(c) * C1] < f || news > u.score[c * CS]);
                else if (retc)               fire = parent_can && news > u.score[c * CS];
                else                         fire = parent_can;
                const bool entered_first = fire && parent_first;
                const bool listed = fire && (P < R || !(in_acl && !parent_first && retc));
                if (in_acl && !retc && !entered_first) ch_clear<CS, C1>(p, u, c);
                if (retc) u.frame[(c) * C1] = nf;
                if (fire) ch_enter<CS, C1>(u, c, news, u.o_outh[P], nf);
                u.o_frame[c] = (fire ? (listed ? 2 : 4) : 0) | ((retc && !entered_first) ? 8 : 0);
            };
            for (int i = tid; i < R + na; i += NT) {
                const int node = i < R ? i : u.acl[cur][i - R];
                if (i >= R) decide(node);
                // a node that is not retained enters none of its children: their (stale) decision words are not
                // looked at below either, so they need no visit -- on a large tree most roots are idle most of the time
                if (!(u.flag[node] & 1)) continue;
                for (int c = p.node_child[node]; c >= 0; c = p.node_sib[c]) if (u.pos[c] < 0) decide(c);
            }
            __syncthreads();
            for (int q = tid; q < na; q += NT) u.pos[u.acl[cur][q]] = -1;        // (nothing below reads pos or a root's frame
            for (int i = tid; i < R; i += NT) if (u.flag[i] & 1) u.frame[(i) * C1] = nf;  //  before the next barrier)
        }
        else {
            for (int i = tid; i < N; i += NT) {
                u.pos[i] = -1; u.o_frame[i] = u.frame[(i) * C1]; u.o_s0[i] = u.score[i * CS]; u.o_best[i] = u.best[(i) * C1];
                u.o_out[i] = u.out[(i) * C1]; u.o_outh[i] = u.outh[(i) * C1];
            }
            __syncthreads();
            for (int q = tid; q < n_acl[cur]; q += NT) u.pos[u.acl[cur][q]] = q;
            __syncthreads();
            // flag bits: 1 retained, 2 fire (listed by parent), 4 fire (not listed), 8 self-append
            for (int c = tid; c < N; c += NT) {
                const bool active = c < R ? u.o_frame[c] >= f : u.pos[c] >= 0;
                u.flag[c] = (active && u.o_best[c] > thresh) ? 1 : 0;
            }
            __syncthreads();
            for (int c = R + tid; c < N; c += NT) {
                const int P = p.parent[c], pc = u.pos[c];
                const bool in_acl = pc >= 0, retc = u.flag[c] & 1;
                const int32_t news = u.o_out[P] + p.pip;
                const bool par_active = P < R ? true : u.pos[P] >= 0;
                const bool parent_can = par_active && (u.flag[P] & 1) && (p.has_pl || news > npt)
                                        && (news + ft_pen(p, pp, p.node_ci[c]) > npt);
                const bool parent_first = P < R || !in_acl || u.pos[P] < pc;
                bool fire;
                if (!in_acl || parent_first) fire = parent_can && (u.o_frame[c] < f || news > u.o_s0[c]);
                else if (retc)               fire = parent_can && news > u.o_s0[c];
                else                         fire = parent_can;
                const bool entered_first = fire && parent_first;
                const bool selfapp = in_acl && retc && !entered_first;
                const bool listed = fire && (P < R || !(in_acl && !parent_first && retc));
                const bool cleared = in_acl && !retc && !entered_first;
                if (cleared) ch_clear<CS, C1>(p, u, c);
                if (in_acl && retc) u.frame[(c) * C1] = nf;
                if (fire) ch_enter<CS, C1>(u, c, news, u.o_outh[P], nf);
                // decision word for the list phase; o_frame[c] is read by this thread only, so it can be reused
                u.o_frame[c] = (fire ? (listed ? 2 : 4) : 0) | (selfapp ? 8 : 0);
            }
            for (int i = tid; i < R; i += NT) if (u.flag[i] & 1) u.frame[(i) * C1] = nf;
            __syncthreads();
        }
        // list positions: root phase (segment per root), then one segment per list position
        for (int i = tid; i < R + n_acl[cur]; i += NT) {
            const int node = i < R ? i : u.acl[cur][i - R];
            int k = (i >= R && (u.o_frame[node] & 8)) ? 1 : 0;
            if (!LIST || (u.flag[node] & 1))
                for (int c = p.node_child[node]; c >= 0; c = p.node_sib[c]) k += (u.o_frame[c] & 2) ? 1 : 0;
            cnt[i] = k;
        }
        __syncthreads();
        const int32_t n_listed = ft_block_scan<NT>(cnt, R + n_acl[cur], s_scan);      // exclusive prefix sum
        for (int i = tid; i < R + n_acl[cur]; i += NT) {
            const int node = i < R ? i : u.acl[cur][i - R];
            int o = cnt[i];
            if (i >= R && (u.o_frame[node] & 8)) u.acl[nxt][o++] = node;
            if (!LIST || (u.flag[node] & 1))
                for (int c = p.node_child[node]; c >= 0; c = p.node_sib[c]) if (u.o_frame[c] & 2) u.acl[nxt][o++] = c;
        }
        n_acl[nxt] = n_listed;
        __syncthreads();
        // last-phone candidates: list order, homophone chain inside
        for (int i = tid; i < R + n_acl[cur]; i += NT) {
            const int node = i < R ? i : u.acl[cur][i - R];
            const int32_t news = u.o_out[node] + p.pip;
            int k = 0;
            if ((u.flag[node] & 1) && (p.has_pl || news > lpt))
                for (int w = p.node_pw[node]; w >= 0; w = p.homophone[w]) k += (news + ft_pen(p, pp, p.d_last[w]) > lpt) ? 1 : 0;
            cnt[i] = k;
        }
        __syncthreads();
        {
            const int32_t n_cand_all = ft_block_scan<NT>(cnt, R + n_acl[cur], s_scan);
            if (tid == 0) s_sc[5] = n_cand_all;
        }
        __syncthreads();
        for (int i = tid; i < R + n_acl[cur]; i += NT) {
            const int node = i < R ? i : u.acl[cur][i - R];
            const int32_t news = u.o_out[node] + p.pip;
            int o = cnt[i];
            if ((u.flag[node] & 1) && (p.has_pl || news > lpt))
                for (int w = p.node_pw[node]; w >= 0; w = p.homophone[w])
                    if (news + ft_pen(p, pp, p.d_last[w]) > lpt) {
                        u.cand_wid[o] = w; u.cand_score[o] = news - p.nwpen; u.cand_bp[o] = u.o_outh[node]; ++o;
                    }
        }
        __syncthreads();

        // ---- word level: last_phone_transition (:884-1035).  Candidates of one frame name distinct words (a word has
        //      one penultimate tree node, a node one place in the active list), so each candidate's best
        //      predecessor -- exit score + language score over the back-pointers of its start frame, the
        //      look-ups that dominate this step -- is found by its own thread; last_ltrans (lt_*) is the
        //      reference's per-word cache keyed by start frame.  Should two candidates ever share a word, the
        //      reference's loops are run as written by one thread.
        {
            const int n_cand = s_sc[5];
            if (tid == 0) s_red[7] = 0;
            __syncthreads();
            if (LIST) {                                  // O(1) per candidate: the frame stamp of the word
                for (int i = tid; i < n_cand; i += NT)
                    if (atomicExch(&u.cand_mark[u.cand_wid[i]], f) == f) s_red[7] = 1;
            }
            else
            for (int i = tid; i < n_cand; i += NT) {
                const int w = u.cand_wid[i];
                for (int j = 0; j < i; ++j) if (u.cand_wid[j] == w) s_red[7] = 1;
            }
            __syncthreads();
        }
        if (s_red[7] == 0) {
            const int n_cand = s_sc[5];
            int32_t bestscore = kW;
            for (int i = tid; i < n_cand; i += NT) {
                const int cb = u.cand_bp[i], w = u.cand_wid[i];
                int32_t score = u.cand_score[i];
                if (cb != -1) {
                    const int first = p.d_first[w];
                    score -= ft_exit_score(p, u, cb, first);
                    const int ef = BPC(u, B_FRAME, cb);
                    if (u.lt_sf[w] != ef + 1) {
                        int32_t best = kW, bestbp = u.lt_bp[w];
                        const int b1 = u.bp_table_idx[ef + 1], base = p.d_base[w];
                        for (int bp = u.bp_table_idx[ef]; bp < b1; ++bp) {
                            if (!BPC(u, B_VALID, bp)) continue;
                            int32_t dscr = ft_exit_score(p, u, bp, first);
                            if (dscr > kW) dscr += ft_lm(p, base, BPC(u, B_REAL, bp), BPC(u, B_PREAL, bp));
                            if (dscr > best) { best = dscr; bestbp = bp; }
                        }
                        u.lt_dscr[w] = best; u.lt_bp[w] = bestbp; u.lt_sf[w] = ef + 1;
                    }
                }
                score += u.lt_dscr[w];
                u.cand_score[i] = score;
                u.cand_bp[i] = u.lt_bp[w];
                bestscore = max(bestscore, score);
            }
            if (bestscore > kW) atomicMax(&s_sc[1], bestscore);
        }
        else
        if (tid == 0) {
            const int n_cand = s_sc[5];
            int n_csf = 0;
            for (int i = 0; i < n_cand; ++i) {
                const int cb = u.cand_bp[i], w = u.cand_wid[i];
                if (cb == -1) continue;
                u.cand_score[i] -= ft_exit_score(p, u, cb, p.d_first[w]);
                const int ef = BPC(u, B_FRAME, cb);
                if (u.lt_sf[w] != ef + 1) {
                    int j;
                    for (j = 0; j < n_csf; ++j) if (u.csf_ef[j] == ef) break;
                    if (j < n_csf) u.cand_next[i] = u.csf_cand[j];
                    else { j = n_csf++; u.cand_next[i] = -1; u.csf_ef[j] = ef; }
                    u.csf_cand[j] = i;
                    u.lt_dscr[w] = kW;
                    u.lt_sf[w] = ef + 1;
                }
            }
            for (int i = 0; i < n_csf; ++i) {
                const int b1 = u.bp_table_idx[u.csf_ef[i] + 1];
                for (int bp = u.bp_table_idx[u.csf_ef[i]]; bp < b1; ++bp) {
                    if (!BPC(u, B_VALID, bp)) continue;
                    for (int j = u.csf_cand[i]; j >= 0; j = u.cand_next[j]) {
                        const int w = u.cand_wid[j];
                        int32_t dscr = ft_exit_score(p, u, bp, p.d_first[w]);
                        if (dscr > kW) dscr += ft_lm(p, p.d_base[w], BPC(u, B_REAL, bp), BPC(u, B_PREAL, bp));
                        if (dscr > u.lt_dscr[w]) { u.lt_dscr[w] = dscr; u.lt_bp[w] = bp; }
                    }
                }
            }
            int32_t bestscore = s_sc[1];
            for (int i = 0; i < n_cand; ++i) {
                const int w = u.cand_wid[i];
                u.cand_score[i] += u.lt_dscr[w];
                u.cand_bp[i] = u.lt_bp[w];
                if (u.cand_score[i] > bestscore) bestscore = u.cand_score[i];
            }
            s_sc[1] = bestscore;
            s_sc[5] = n_cand;
        }
        __syncthreads();
        {
            // ---- last_phone_transition's entering loop (:1004-1030), one thread per candidate.  Candidates of
            //      one frame name distinct words (a word has one penultimate tree node) -- if that ever fails the
            //      loop is run by one thread in candidate order.
            const int n_cand = s_sc[5];
            const int32_t cthresh = s_sc[1] + p.lponlybeam;
            const bool dup = s_red[7] != 0;                     // (found above)
            if (LIST && !dup) {
                // one work-item per (entering candidate, right context): the word's slots are exactly its right contexts, so
                // "allocate the missing ones, then enter every present one" is, per slot, "create if missing, then enter"
                if (tid == 0) s_nwc2 = 0;
                for (int i = tid; i < n_cand; i += NT) cnt[i] = 0;
                __syncthreads();
                for (int i = tid; i < n_cand; i += NT) {
                    if (!(u.cand_score[i] > cthresh)) continue;
                    const int w = u.cand_wid[i], nrc = p.wc_off[w + 1] - p.wc_off[w];
                    const int q = atomicAdd(&s_nwc2, nrc);
                    for (int r = 0; r < nrc; ++r) { u.xlist[q + r] = i; u.xslot[q + r] = p.wc_off[w] + r; }
                }
                __syncthreads();
                for (int j = tid; j < s_nwc2; j += NT) {
                    const int i = u.xlist[j], slot = u.xslot[j], w = u.cand_wid[i], c = WC + slot;
                    if (!u.present[slot]) {                     // ngram_search_alloc_all_rc (ngram_search.c:583-633)
                        const int last = p.d_last[w], last2 = p.d_last2[w];
                        ch_init<CS, C1>(p, u, c, 0, p.rs_ssid[((size_t)last * p.n_ci + last2) * p.n_ci + (slot - p.wc_off[w])], p.ci_tmat[last]);
                        u.present[slot] = 1;
                    }
                    if (u.frame[(c) * C1] < f || u.cand_score[i] > u.score[c * CS]) {
                        ch_enter<CS, C1>(u, c, u.cand_score[i], u.cand_bp[i], nf);
                        cnt[i] = 1;
                    }
                }
            }
            else
            for (int i = (dup ? (tid == 0 ? 0 : n_cand) : tid); i < n_cand; i += (dup ? 1 : NT)) {
                int k = 0;
                if (u.cand_score[i] > cthresh) {
                    const int w = u.cand_wid[i];
                    // ngram_search_alloc_all_rc (ngram_search.c:583-633)
                    const int last = p.d_last[w], last2 = p.d_last2[w], nrc = p.rs_n[last * p.n_ci + last2];
                    for (int r = 0; r < nrc; ++r) {
                        const int slot = p.wc_off[w] + r;
                        if (!u.present[slot]) {
                            ch_init<CS, C1>(p, u, WC + slot, 0, p.rs_ssid[((size_t)last * p.n_ci + last2) * p.n_ci + r], p.ci_tmat[last]);
                            u.present[slot] = 1;
                        }
                    }
                    for (int slot = p.wc_off[w]; slot < p.wc_off[w + 1]; ++slot) {
                        if (!u.present[slot]) continue;
                        const int c = WC + slot;
                        if (u.frame[(c) * C1] < f || u.cand_score[i] > u.score[c * CS]) { ch_enter<CS, C1>(u, c, u.cand_score[i], u.cand_bp[i], nf); ++k; }
                    }
                }
                cnt[i] = k > 0;
            }
            __syncthreads();
            if (LIST) {                                  // stable compaction by a prefix sum
                const int32_t nawl = ft_block_scan<NT>(cnt, n_cand, s_scan);
                for (int i = tid; i < n_cand; i += NT)
                    if ((i + 1 < n_cand ? cnt[i + 1] : nawl) != cnt[i]) {
                        const int w = u.cand_wid[i];
                        u.awl[nxt][cnt[i]] = w; u.word_active[w] = 1;
                    }
                if (tid == 0) s_red[5] = nawl;
            }
            else
            if (tid == 0) {
                int nawl = 0;
                for (int i = 0; i < n_cand; ++i)
                    if (cnt[i]) { const int w = u.cand_wid[i]; u.awl[nxt][nawl++] = w; u.word_active[w] = 1; }
                s_red[5] = nawl;
            }
            __syncthreads();
            // ---- prune_word_chan (:1038-1128): pass A, one thread per active word -- keep / free the
            //      right-context channels, count the survivors, note whether the word exits
            const int32_t nwt = s_sc[1] + p.wbeam, lpth = s_sc[1] + p.lponlybeam;
            const int wst = p.big ? p.n_w : 1024;                // n_awl <= n_w
            int32_t *w_k = p.big ? u.g_w : cnt, *w_exit = w_k + wst, *w_bp = w_k + 2 * wst, *w_bss = w_k + 3 * wst;
            if (LIST) {
                // one work-item per channel of the evaluation list (the channels present when the frame was evaluated; the ones
                // this frame's candidates have just allocated were all entered for the next frame and have no score yet: the
                // word-at-a-time walk below would neither count nor free them), survivors counted per word by atomics
                for (int i = tid; i < n_awl[cur]; i += NT) { w_k[i] = 0; w_exit[i] = 0; }
                __syncthreads();
                for (int j = tid; j < s_nwc; j += NT) {
                    const int c = u.elist[j], i = u.eword[j];
                    if (u.best[(c) * C1] > lpth) {
                        u.frame[(c) * C1] = nf;
                        atomicAdd(&w_k[i], 1);
                        if (u.out[(c) * C1] > nwt) atomicOr(&w_exit[i], 1);
                    }
                    else if (u.frame[(c) * C1] != nf) u.present[c - WC] = 0;
                }
                __syncthreads();
            }
            for (int i = tid; i < n_awl[cur]; i += NT) {
                const int w = u.awl[cur][i];
                int k = LIST ? w_k[i] : 0, ex = LIST ? w_exit[i] : 0;
                if (!LIST)
                for (int slot = p.wc_off[w]; slot < p.wc_off[w + 1]; ++slot) {
                    if (!u.present[slot]) continue;
                    const int c = WC + slot;
                    if (u.best[(c) * C1] > lpth) { u.frame[(c) * C1] = nf; ++k; ex |= (u.out[(c) * C1] > nwt); }
                    else if (u.frame[(c) * C1] != nf) u.present[slot] = 0;
                }
                w_k[i] = k; w_exit[i] = ex;
                if (LIST) {                              // inputs of the three prefix sums below
                    w_k[i] = (k > 0 && !u.word_active[w]) ? 1 : 0;
                    w_bp[i] = ex ? 1 : 0;
                    w_bss[i] = ex ? p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]] : 0;
                }
            }
            __syncthreads();
            if (LIST) {                                  // positions by workgroup prefix sums
                const int na = n_awl[cur];
                const int32_t bpidx = s_sc[3], bss_head = s_sc[4], nawl = s_red[5];    // (rewritten below, after the scans' barriers)
                const int32_t n_exit = ft_block_scan<NT>(w_bp, na, s_scan);
                const int32_t n_bss = ft_block_scan<NT>(w_bss, na, s_scan);
                const int32_t n_app = ft_block_scan<NT>(w_k, na, s_scan);
                for (int i = tid; i < na; i += NT) {
                    w_bp[i] += bpidx; w_bss[i] += bss_head;
                    if ((i + 1 < na ? w_k[i + 1] : n_app) != w_k[i]) {
                        const int w = u.awl[cur][i];
                        u.awl[nxt][nawl + w_k[i]] = w; u.word_active[w] = 1;
                    }
                }
                if (tid == 0) {
                    if (bpidx + n_exit + p.n1 >= u.bp_cap || bss_head + n_bss + p.n_ci >= u.bss_cap) s_sc[6] = 1;
                    s_sc[3] = bpidx + n_exit; s_sc[4] = bss_head + n_bss; s_red[5] = nawl + n_app;
                }
            }
            else
            if (tid == 0) {                                     // positions: back-pointers, score stack, next active words
                int32_t bpidx = s_sc[3], bss_head = s_sc[4];
                int nawl = s_red[5];
                for (int i = 0; i < n_awl[cur]; ++i) {
                    const int w = u.awl[cur][i];
                    w_bp[i] = bpidx; w_bss[i] = bss_head;
                    if (w_exit[i]) { ++bpidx; bss_head += p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]]; }
                    if (w_k[i] > 0 && !u.word_active[w]) { u.awl[nxt][nawl++] = w; u.word_active[w] = 1; }
                }
                if (bpidx + p.n1 >= u.bp_cap || bss_head + p.n_ci >= u.bss_cap) s_sc[6] = 1;
                s_sc[3] = bpidx; s_sc[4] = bss_head; s_red[5] = nawl;
            }
            __syncthreads();
            if (!s_sc[6]) {
                // pass B: every exiting word writes its own back-pointer (first exit creates, the others update)
                for (int i = tid; i < n_awl[cur]; i += NT) {
                    if (!w_exit[i]) continue;
                    const int w = u.awl[cur][i];
                    int32_t bpi = w_bp[i], bsh = w_bss[i];
                    for (int slot = p.wc_off[w]; slot < p.wc_off[w + 1]; ++slot) {
                        if (!u.present[slot]) continue;
                        const int c = WC + slot;
                        if (u.frame[(c) * C1] == nf && u.best[(c) * C1] > lpth && u.out[(c) * C1] > nwt)
                            ft_save_bp(p, u, bpi, bsh, f, w, u.out[(c) * C1], u.outh[(c) * C1], slot - p.wc_off[w]);
                    }
                }
            }
            __syncthreads();
        }
        if (LIST && !s_sc[6]) {
            // single-phone words (:1100-1127), one work-item per word; back-pointer positions in list order by prefix sums
            const int32_t nwt = s_sc[1] + p.wbeam, lpth = s_sc[1] + p.lponlybeam;
            const int wst = p.big ? p.n_w : 1024;                // n1 <= n_w
            int32_t *f_ex = p.big ? u.g_w : cnt, *f_new = f_ex + wst, *f_rc = f_ex + 2 * wst;
            for (int i = tid; i < p.n1; i += NT) {
                const int c = W1 + i;
                int ex = 0, nw = 0, rcn = 0;
                if (u.frame[(c) * C1] >= f && u.best[(c) * C1] > lpth) {
                    u.frame[(c) * C1] = nf;
                    if (u.out[(c) * C1] > nwt) {
                        const int w = p.w1_wid[i];
                        ex = 1;
                        if (u.word_lat_idx[w] == -1) {
                            nw = 1;
                            rcn = p.d_pronlen[w] == 1 ? 0 : p.rs_n[p.d_last[w] * p.n_ci + p.d_last2[w]];
                        }
                    }
                }
                f_ex[i] = ex; f_new[i] = nw; f_rc[i] = rcn;
            }
            __syncthreads();
            const int32_t bpidx0 = s_sc[3], bss0 = s_sc[4];
            const int32_t n_new = ft_block_scan<NT>(f_new, p.n1, s_scan);
            const int32_t n_rc = ft_block_scan<NT>(f_rc, p.n1, s_scan);
            for (int i = tid; i < p.n1; i += NT)
                if (f_ex[i]) {
                    int32_t bpi = bpidx0 + f_new[i], bsh = bss0 + f_rc[i];
                    if (!ft_save_bp(p, u, bpi, bsh, f, p.w1_wid[i], u.out[(W1 + i) * C1], u.outh[(W1 + i) * C1], 0)) s_sc[6] = 1;
                }
            __syncthreads();
            if (tid == 0) { s_sc[3] = bpidx0 + n_new; s_sc[4] = bss0 + n_rc; }
        }
        if (tid == 0 && !s_sc[6]) {
            int32_t bpidx = s_sc[3], bss_head = s_sc[4];
            bool ok = true;
            const int32_t nwt = s_sc[1] + p.wbeam, lpth = s_sc[1] + p.lponlybeam;
            for (int i = 0; i < p.n1 && ok && !LIST; ++i) {
                const int c = W1 + i;
                if (u.frame[(c) * C1] < f) continue;
                if (u.best[(c) * C1] > lpth) {
                    u.frame[(c) * C1] = nf;
                    if (u.out[(c) * C1] > nwt) ok = ft_save_bp(p, u, bpidx, bss_head, f, p.w1_wid[i], u.out[(c) * C1], u.outh[(c) * C1], 0);
                }
            }
            // bptable_maxwpf (:1193-1241)
            if (!(p.maxwpf == -1 || p.maxwpf == p.n_w)) {
                const int b0 = u.bp_table_idx[f];
                int32_t bestscr = kMaxNegInt32; int bestbp = -1, n = 0;
                for (int bp = b0; bp < bpidx; ++bp)
                    if (p.d_filler[BPC(u, B_WID, bp)]) {
                        if (BPC(u, B_SCORE, bp) > bestscr) { bestscr = BPC(u, B_SCORE, bp); bestbp = bp; }
                        BPC(u, B_VALID, bp) = 0; ++n;
                    }
                if (bestbp >= 0) { BPC(u, B_VALID, bestbp) = 1; --n; }
                n = (bpidx - b0) - n;
                for (; n > p.maxwpf; --n) {
                    int32_t worst = 0x7fffffff; int wbp = -1;
                    for (int bp = b0; bp < bpidx; ++bp)
                        if (BPC(u, B_VALID, bp) && BPC(u, B_SCORE, bp) < worst) { worst = BPC(u, B_SCORE, bp); wbp = bp; }
                    if (wbp < 0) break;
                    BPC(u, B_VALID, wbp) = 0;
                }
            }
            s_sc[3] = bpidx; s_sc[4] = bss_head; if (!ok) s_sc[6] = 1;
        }
        __syncthreads();
        n_awl[nxt] = s_red[5];
        if (s_sc[6]) break;

        // ---- word_transition (:1243-1427)
        const int bp0 = u.bp_table_idx[f], bp1 = s_sc[3];
        int32_t *brc_score = s_bins, *brc_path = s_bins + kFtMaxCi, *brc_lc = s_bins + 2 * kFtMaxCi;
        if (tid == 0) s_red[6] = 0;
        __syncthreads();
        for (int bp = bp0 + tid; bp < bp1; bp += NT) {
            u.word_lat_idx[BPC(u, B_WID, bp)] = -1;
            if (BPC(u, B_WID, bp) != p.finishwid) atomicAdd(&s_red[6], 1);
        }
        for (int rc = tid; rc < p.n_ci; rc += NT) {     // best exit per right-context phone, earliest bp on ties
            int32_t bs = kW; int path = 0, lc = 0;
            for (int bp = bp0; bp < bp1; ++bp) {
                if (BPC(u, B_WID, bp) == p.finishwid) continue;
                const int l2 = BPC(u, B_LAST2, bp), l1 = BPC(u, B_LAST, bp);
                const int32_t sc = l2 == -1 ? BPC(u, B_SCORE, bp)
                    : u.bss[BPC(u, B_SIDX, bp) + p.rs_cimap[((size_t)l1 * p.n_ci + l2) * p.n_ci + rc]];
                if (sc > bs) { bs = sc; path = bp; lc = l1; }
            }
            brc_score[rc] = bs; brc_path[rc] = path; brc_lc[rc] = lc;
        }
        __syncthreads();
        if (s_red[6] > 0) {
            for (int i = tid; i < R; i += NT) {          // tree roots (:1306-1325)
                const int ci = p.node_ci[i];
                const int32_t ns = brc_score[ci] + p.nwpen + p.pip;
                if (ns + ft_pen(p, pp, ci) > thresh && (u.frame[(i) * C1] < f || ns > u.score[i * CS])) {
                    ch_enter<CS, C1>(u, i, ns, brc_path[ci], nf);
                    u.senid[i * CS] = p.ldiph[((size_t)ci * p.n_ci + p.node_ci2[i]) * p.n_ci + brc_lc[ci]];
                }
            }
            for (int i = tid; i < p.n1lm; i += NT) {     // in-LM single-phone words (:1331-1388)
                const int w = p.w1_wid[i];
                int32_t ds = kMaxNegInt32; int dbp = 0;
                for (int bp = bp0; bp < bp1; ++bp) {
                    if (!BPC(u, B_VALID, bp)) continue;
                    int32_t ns = ft_exit_score(p, u, bp, p.d_first[w]);
                    if (ns != kW) ns += ft_lm(p, p.d_base[w], BPC(u, B_REAL, bp), BPC(u, B_PREAL, bp));
                    if (ns > ds) { ds = ns; dbp = bp; }
                }
                u.lt_dscr[w] = ds; u.lt_bp[w] = dbp;
                if (w == p.startwid) continue;
                const int c = W1 + i;
                const int32_t ns = ds + p.pip;
                if (ns + ft_pen(p, pp, p.w1_ci[i]) > thresh && (u.frame[(c) * C1] < f || ns > u.score[c * CS])) {
                    ch_enter<CS, C1>(u, c, ns, dbp, nf);
                    u.senid[c * CS] = p.ldiph[((size_t)p.w1_ci[i] * p.n_ci + p.w1_ci2[i]) * p.n_ci + p.d_last[BPC(u, B_WID, dbp)]];
                }
            }
            for (int w = p.filler_start - 1 + tid; w <= p.filler_end; w += NT) {    // <sil> and noise words (:1390-1426)
                // slot filler_start - 1 stands for <sil>, which is handled whatever its place in the dictionary
                const bool is_sil = w == p.filler_start - 1;
                if (!is_sil && (w == p.startwid || w == p.silwid)) continue;
                const int i = p.w1_of_word[is_sil ? p.silwid : w];
                if (i < 0) continue;
                const int c = W1 + i;
                const int32_t ns = brc_score[p.sil_ci] + (is_sil ? p.silpen : p.fillpen) + p.pip;
                if (ns + ft_pen(p, pp, p.w1_ci[i]) > thresh && (u.frame[(c) * C1] < f || ns > u.score[c * CS]))
                    ch_enter<CS, C1>(u, c, ns, brc_path[p.sil_ci], nf);
            }
        }
        __syncthreads();
        // ---- deactivate_channels (:1429-1450)
        for (int i = tid; i < R; i += NT) if (u.frame[(i) * C1] == f) ch_clear<CS, C1>(p, u, i);
        for (int i = tid; i < p.n1; i += NT) if (u.frame[(W1 + i) * C1] == f) ch_clear<CS, C1>(p, u, W1 + i);
        if (tid == 0) {
            u.step[f * 4] = s_sc[0]; u.step[f * 4 + 1] = s_sc[1]; u.step[f * 4 + 2] = s_sc[3]; u.step[f * 4 + 3] = n_acl[nxt];
            ++s_sc[7];
        }
        __syncthreads();
    }
    if (tid == 0) {
        u.bp_table_idx[s_sc[7]] = s_sc[3];                       // ngram_fwdtree_finish: mark one past the last frame
        u.result[0] = s_sc[3]; u.result[1] = s_sc[4]; u.result[2] = s_sc[7]; u.result[3] = s_sc[6];
        u.result[4] = s_sc[0];                                   // ngs->best_score as the last frame left it
    }
    // what the second pass inherits besides the tables: the permanent single-phone channels keep their per-state ssids
    // through hmm_clear (ngram_fwdflat_start, ngram_search_fwdflat.c:385-392)
    if (p.w1_out)
        for (int i = tid; i < p.n1 * NE; i += NT)
            p.w1_out[((size_t)blockIdx.x * p.n1 + i / NE) * NE + i % NE] = u.senid[(W1 + i / NE) * CS + i % NE];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static const T *ft_up(psgpu_fwdtree_s *m, const T *src, size_t n, int *rc)
{
    void *d = nullptr;
    if (*rc != PSGPU_OK) return nullptr;
    if (hipMalloc(&d, n * sizeof(T) ? n * sizeof(T) : 1) != hipSuccess ||
        hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        psgpu_set_error("fwdtree model upload failed");
        *rc = PSGPU_ENOMEM;
        hipFree(d);
        return nullptr;
    }
    m->allocs.push_back(d);
    return (const T *)d;
}

extern "C" {

int psgpu_fwdtree_create(psgpu_fwdtree_t **out, const psgpu_fwdtree_tables_t *t)
{
    PSGPU_REQUIRE(out && t && t->par, "psgpu_fwdtree_create: NULL argument");
    *out = nullptr;
    int rc = psgpu_check_device();
    if (rc != PSGPU_OK) return rc;
    const int32_t *q = t->par;
    psgpu_fwdtree_s *m = new psgpu_fwdtree_s();
    FtDev &d = m->d;
    memset(&d, 0, sizeof d);
    d.n_ci = q[0]; d.n_emit = q[1]; d.n_sen = q[2]; d.n_w = q[3]; d.R = q[4]; d.M = q[5]; d.N = d.R + d.M; d.n1 = q[6];
    d.n1lm = q[7]; d.beam = q[8]; d.pbeam = q[9]; d.lpbeam = q[10]; d.lponlybeam = q[11]; d.wbeam = q[12]; d.pip = q[13];
    d.nwpen = q[14]; d.silpen = q[15]; d.fillpen = q[16]; d.maxhmmpf = q[17]; d.maxwpf = q[18]; d.startwid = q[19];
    d.finishwid = q[20]; d.silwid = q[21]; d.filler_start = q[22]; d.filler_end = q[23]; d.sil_ci = q[24]; d.has_pl = q[25];
    d.big = (d.N + d.R > kFtMaxN || d.n_w > 1024) ? 1 : 0;
    if (!(d.n_emit == 3 || d.n_emit == 5) || d.n_ci < 1 || d.n_ci > kFtMaxCi || d.N < 1 || d.n_w < 1) {
        psgpu_set_error("fwdtree: unsupported shape (n_emit %d, n_ci %d, tree nodes %d, words %d)", d.n_emit, d.n_ci, d.N, d.n_w);
        delete m;
        return PSGPU_EINVAL;
    }
    const size_t nci3 = (size_t)d.n_ci * d.n_ci * d.n_ci, n1 = (size_t)d.n_w + 1;
    std::vector<int32_t> parent(d.N, -1), w1_of(d.n_w, -1), wc_off(d.n_w + 1, 0);
    for (int i = 0; i < d.N; ++i)
        for (int c = t->node_child[i]; c >= 0; c = t->node_sib[c]) parent[c] = i;
    for (int i = 0; i < d.n1; ++i) w1_of[t->w1_wid[i]] = i;
    int tot = 0;
    for (int w = 0; w < d.n_w; ++w) {
        wc_off[w] = tot;
        if (t->dict_pronlen[w] > 1) tot += t->rssid_n[t->dict_last[w] * d.n_ci + t->dict_last2[w]];
    }
    wc_off[d.n_w] = tot;
    d.TOT = tot;
    m->C = d.N + d.n1 + tot;
    d.node_ci = ft_up(m, t->node_ci, d.N, &rc); d.node_ci2 = ft_up(m, t->node_ci2, d.N, &rc);
    d.node_ssid = ft_up(m, t->node_ssid, d.N, &rc); d.node_tmat = ft_up(m, t->node_tmat, d.N, &rc);
    d.node_child = ft_up(m, t->node_child, d.N, &rc); d.node_sib = ft_up(m, t->node_sib, d.N, &rc);
    d.node_pw = ft_up(m, t->node_penult_wid, d.N, &rc); d.parent = ft_up(m, parent.data(), d.N, &rc);
    d.homophone = ft_up(m, t->homophone_set, d.n_w, &rc);
    d.w1_wid = ft_up(m, t->w1_wid, d.n1, &rc); d.w1_ci = ft_up(m, t->w1_ci, d.n1, &rc); d.w1_ci2 = ft_up(m, t->w1_ci2, d.n1, &rc);
    d.w1_ssid = ft_up(m, t->w1_ssid, d.n1, &rc); d.w1_tmat = ft_up(m, t->w1_tmat, d.n1, &rc); d.w1_mpx = ft_up(m, t->w1_mpx, d.n1, &rc);
    d.w1_of_word = ft_up(m, w1_of.data(), d.n_w, &rc);
    d.d_pronlen = ft_up(m, t->dict_pronlen, d.n_w, &rc); d.d_first = ft_up(m, t->dict_first, d.n_w, &rc);
    d.d_last = ft_up(m, t->dict_last, d.n_w, &rc); d.d_last2 = ft_up(m, t->dict_last2, d.n_w, &rc);
    d.d_base = ft_up(m, t->dict_basewid, d.n_w, &rc); d.d_filler = ft_up(m, t->dict_filler, d.n_w, &rc);
    d.rs_n = ft_up(m, t->rssid_n, (size_t)d.n_ci * d.n_ci, &rc); d.rs_ssid = ft_up(m, t->rssid_ssid, nci3, &rc);
    d.rs_cimap = ft_up(m, t->rssid_cimap, nci3, &rc); d.ldiph = ft_up(m, t->ldiph_lc, nci3, &rc);
    d.ci_tmat = ft_up(m, t->ci_tmat, d.n_ci, &rc);
    d.lm = t->lm ? ft_up(m, t->lm, (size_t)d.n_w * n1 * n1, &rc) : nullptr;     // NULL: psgpu_fwdtree_set_lm supplies the trie
    d.wc_off = ft_up(m, wc_off.data(), (size_t)d.n_w + 1, &rc);
    d.tp = ft_up(m, t->tp, (size_t)t->n_tmat * d.n_emit * (d.n_emit + 1), &rc);
    d.sseq = ft_up(m, t->sseq, (size_t)t->n_sseq * d.n_emit, &rc);
    if (rc != PSGPU_OK) { psgpu_fwdtree_free(m); return rc; }
    *out = m;
    return PSGPU_OK;
}

const LmDev *psgpu_lm_dev(const psgpu_lm_t *lm);     // psgpu_lm.hip

int psgpu_fwdtree_set_lm(psgpu_fwdtree_t *m, const psgpu_lm_t *lm)
{
    PSGPU_REQUIRE(m && lm, "psgpu_fwdtree_set_lm: NULL argument");
    const LmDev *d = psgpu_lm_dev(lm);
    PSGPU_REQUIRE(d->n_words == m->d.n_w, "psgpu_fwdtree_set_lm: the model maps %d dictionary words, the search has %d", d->n_words, m->d.n_w);
    m->d.trie = *d;
    m->d.use_trie = 1;
    return PSGPU_OK;
}

int psgpu_fwdtree_set_mode(psgpu_fwdtree_t *m, int32_t mode)
{
    PSGPU_REQUIRE(m && (mode == PSGPU_FWDTREE_PER_NODE || mode == PSGPU_FWDTREE_ACTIVE_LIST), "psgpu_fwdtree_set_mode: bad argument");
    m->d.list_mode = mode;
    return PSGPU_OK;
}

int psgpu_fwdtree_set_w1_ssid_out(psgpu_fwdtree_t *m, int32_t *w1_ssid_dev)
{
    PSGPU_REQUIRE(m, "psgpu_fwdtree_set_w1_ssid_out: NULL argument");
    m->d.w1_out = w1_ssid_dev;
    return PSGPU_OK;
}

void psgpu_fwdtree_free(psgpu_fwdtree_t *m)
{
    if (!m) return;
    for (void *p : m->allocs) hipFree(p);
    delete m;
}

// Search n_utt utterances.  senscr_dev [total][scr_stride] int16: the scores acmod_score hands the search for each
// frame; penalties_dev [total][n_ci]; utt_off_dev [n_utt + 1].  Per utterance u the columns of the back-pointer
// table go to bp_dev + u * 10 * bp_cap, the score stack to bss_dev + u * bss_cap, the frame marks to
// idx_dev + u * (max_frames + 2), per-frame diagnostics to step_dev + u * max_frames * 4, and
// result_dev + u * 8 = {n back-pointers, score-stack length, frames searched, status (1 = a table was full)}.
int psgpu_fwdtree_search_dev(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                             const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                             int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                             int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                             int32_t pl_window, void *stream)
{
    PSGPU_REQUIRE(m && n_utt >= 0 && max_frames >= 0 && bp_cap > 0 && bss_cap > 0, "psgpu_fwdtree_search_dev: bad argument");
    PSGPU_REQUIRE(!raw_scores || (m->d.n_sen <= kFtMaxSen && pl_window >= 0), "raw-score mode: n_sen %d > %d or negative pl_window",
                  m->d.n_sen, kFtMaxSen);
    PSGPU_REQUIRE(m->d.lm || m->d.use_trie, "psgpu_fwdtree_search_dev: no language model (dense table or psgpu_fwdtree_set_lm)");
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE(senscr_dev && penalties_dev && utt_off_dev && bp_dev && bss_dev && idx_dev && step_dev && result_dev,
                  "psgpu_fwdtree_search_dev: NULL device buffer");
    const FtDev &d = m->d;
    hipStream_t st = (hipStream_t)stream;
    // per-utterance work slab
    const size_t C = m->C;
    const size_t per = C * (d.list_mode ? 24 : 5 + 5 + 4 + 5 + 2) + d.TOT + 2 * (size_t)d.N + 2 * (size_t)d.n_w + 2 * (size_t)d.n_w
                     + 4 * ((size_t)d.n_w + 1) + 3 * (size_t)d.n_w + 2 * ((size_t)d.n_w + 1) + 7 * (size_t)d.N + 64
                     + ((size_t)d.n_sen + 1) / 2 + 1 + (size_t)d.n_w + (d.list_mode ? 4 * ((size_t)d.TOT + 1) : 0)
                     + (d.big ? (size_t)std::max(d.N + d.R, d.n_w) + 1 + 4 * (size_t)d.n_w : 0);
    int32_t *slab = nullptr;
    FtUtt *d_utts = nullptr;
    PSGPU_HIP(hipMalloc((void **)&slab, sizeof(int32_t) * per * n_utt));
    std::vector<FtUtt> hu(n_utt);
    for (int i = 0; i < n_utt; ++i) {
        int32_t *q = slab + per * i;
        FtUtt &u = hu[i];
        auto take = [&](size_t n) { int32_t *r = q; q += n; return r; };
        if (d.list_mode) {                                      // one record per channel (see the kernel)
            const int ne = d.n_emit, rs = ne == 3 ? 16 : 24;
            int32_t *rec = take(C * rs);
            u.score = rec; u.hist = rec + ne; u.out = rec + 2 * ne; u.outh = u.out + 1; u.best = u.out + 2; u.frame = u.out + 3;
            u.senid = u.out + 4; u.tmat = u.senid + ne; u.mpx = u.tmat + 1;
        }
        else {
            u.score = take(C * 5); u.hist = take(C * 5); u.out = take(C); u.outh = take(C); u.best = take(C); u.frame = take(C);
            u.senid = take(C * 5); u.tmat = take(C); u.mpx = take(C);
        }
        u.present = take(d.TOT);
        u.acl[0] = take(d.N); u.acl[1] = take(d.N); u.awl[0] = take(d.n_w); u.awl[1] = take(d.n_w);
        u.word_active = take(d.n_w); u.word_lat_idx = take(d.n_w);
        u.cand_wid = take(d.n_w + 1); u.cand_score = take(d.n_w + 1); u.cand_bp = take(d.n_w + 1); u.cand_next = take(d.n_w + 1);
        u.lt_sf = take(d.n_w); u.lt_dscr = take(d.n_w); u.lt_bp = take(d.n_w);
        u.csf_ef = take(d.n_w + 1); u.csf_cand = take(d.n_w + 1);
        u.o_frame = take(d.N); u.o_s0 = take(d.N); u.o_best = take(d.N); u.o_out = take(d.N); u.o_outh = take(d.N);
        u.pos = take(d.N); u.flag = take(d.N);
        u.nrow = reinterpret_cast<int16_t *>(take(((size_t)d.n_sen + 1) / 2 + 1));
        u.cand_mark = take(d.n_w);
        u.elist = d.list_mode ? take((size_t)d.TOT + 1) : nullptr;
        u.eword = d.list_mode ? take((size_t)d.TOT + 1) : nullptr;
        u.xlist = d.list_mode ? take((size_t)d.TOT + 1) : nullptr; u.xslot = d.list_mode ? take((size_t)d.TOT + 1) : nullptr;
        u.g_cnt = d.big ? take((size_t)std::max(d.N + d.R, d.n_w) + 1) : nullptr;
        u.g_w = d.big ? take(4 * (size_t)d.n_w) : nullptr;
        u.bp = bp_dev + (size_t)i * 10 * bp_cap; u.bss = bss_dev + (size_t)i * bss_cap;
        u.bp_table_idx = idx_dev + (size_t)i * (max_frames + 2); u.step = step_dev + (size_t)i * max_frames * 4;
        u.result = result_dev + (size_t)i * 8;
        u.bp_cap = bp_cap; u.bss_cap = bss_cap;
    }
    std::vector<FtOff> ho(d.list_mode ? n_utt : 0);
    for (size_t i = 0; i < ho.size(); ++i) {
        const FtUtt &u = hu[i];
        FtOff &o = ho[i];
#define X(f) o.f = u.f - slab;
        FT_SLAB_FIELDS(X)
#undef X
        o.acl0 = u.acl[0] - slab; o.acl1 = u.acl[1] - slab; o.awl0 = u.awl[0] - slab; o.awl1 = u.awl[1] - slab;
        o.g_cnt = u.g_cnt ? u.g_cnt - slab : 0; o.g_w = u.g_w ? u.g_w - slab : 0;
        o.nrow = reinterpret_cast<int32_t *>(u.nrow) - slab;
    }
    FtOff *d_offs = nullptr;
    FtBufs bf;
    bf.slab = slab; bf.bp = bp_dev; bf.bss = bss_dev; bf.idx = idx_dev; bf.step = step_dev; bf.res = result_dev;
    bf.bp_cap = bp_cap; bf.bss_cap = bss_cap; bf.max_frames = max_frames;
    hipError_t e = hipMalloc((void **)&d_utts, sizeof(FtUtt) * n_utt);
    if (e == hipSuccess && d.list_mode) e = hipMalloc((void **)&d_offs, sizeof(FtOff) * n_utt);
    if (e == hipSuccess && d.list_mode) e = hipMemcpyAsync(d_offs, ho.data(), sizeof(FtOff) * n_utt, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_utts, hu.data(), sizeof(FtUtt) * n_utt, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);          // hu is about to go out of scope
    if (e != hipSuccess) { hipFree(slab); hipFree(d_utts); hipFree(d_offs); PSGPU_HIP(e); }
    // ACTIVE_LIST on a tree beyond the LDS scratch: ~10^4 active channels per frame, 16 waves per utterance
    const int nt = (d.list_mode && d.big) ? kFtThreadsBig : kFtThreads;
#define FT_LAUNCH(NE, NT, LIST)                                                                                          \
    hipLaunchKernelGGL((fwdtree_kernel<NE, NT, LIST>), dim3(n_utt), dim3(NT), 0, st, d, d_utts, senscr_dev, scr_stride, \
                       penalties_dev, utt_off_dev, raw_scores, pl_window, d_offs, bf)
    if (d.n_emit == 3) {
        if (!d.list_mode) FT_LAUNCH(3, kFtThreads, false);
        else if (nt == kFtThreads) FT_LAUNCH(3, kFtThreads, true);
        else FT_LAUNCH(3, kFtThreadsBig, true);
    }
    else {
        if (!d.list_mode) FT_LAUNCH(5, kFtThreads, false);
        else if (nt == kFtThreads) FT_LAUNCH(5, kFtThreads, true);
        else FT_LAUNCH(5, kFtThreadsBig, true);
    }
#undef FT_LAUNCH
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);          // the slab is freed below: this entry is synchronous
    hipFree(slab); hipFree(d_utts); hipFree(d_offs);
    PSGPU_HIP(e);
    return PSGPU_OK;
}

}  // extern "C"

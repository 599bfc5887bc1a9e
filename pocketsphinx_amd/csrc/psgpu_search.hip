// psgpu_search.hip -- the lexicon-tree search on gfx950: whole utterances, one workgroup per
// utterance, every frame inside the kernel.
//
// Replaces ngram_fwdtree_start + ngram_fwdtree_search x T + ngram_fwdtree_finish
// (reference src/ngram_search_fwdtree.c:469-520, 1452-1495, 1497-1533) and the
// back-pointer helpers they call (src/ngram_search.c:301-498, 583-674): evaluate_channels,
// histogram / beam pruning with phone and last-phone transitions, language-model scores at
// word entry, right-context channel allocation, save_bp, bptable_maxwpf, word_transition,
// deactivate_channels.  Output: the back-pointer table in the reference's own columns
// (bptbl_t, ngram_search.h:112-124), the right-context score stack and the per-frame marks.
//
// Inputs are what the other kernels leave on the device: per-frame senone scores (normalised rows, or the scorer's
// un-normalised rows, in which case the kernel builds each frame's active senone list and normaliser itself) and the
// phone-loop penalties.  Static tables are the reference's own (tree, dictionary, dict2pid, beams) flattened to index
// arrays; language scores come from a dense table over dictionary word ids or from the model's trie (psgpu_lm_dev.h).
//
// Formulation: per-frame work proportional to the active channels (oracle/ps_oracle_search.c `prune_tree_list`): the
// pruning visits the roots, the listed nodes and their children; decisions are taken on a snapshot and list positions
// come from workgroup prefix sums, which the oracle proves equivalent to the reference's sequential walk.  Results are
// the reference's, bit for bit (tests/test_search_gpu.py against reference dumps).
//
// Memory: a recurrence over frames is bound by the latency of its dependent accesses, so where the tree is small enough
// (FtDev.small: en-us + a few hundred words) everything the tree level touches per frame lives in LDS -- channel state
// (structure of arrays: consecutive channels in consecutive banks), the pruning snapshot, the active lists, the per-word
// tables, the children lists of the tree (CSR), the frame's score row (copied in one frame ahead) -- and only the
// right-context fan-out of the words' last phones, the back-pointer table and the language model stay in global memory.
// Larger trees ("slab layouts") keep the word level's arrays in a per-utterance slab in global memory (same source lines, other
// base pointers), the tree level's state as compact channels in list order behind an index in LDS (round 5), and run 1024
// work-items per utterance.
#include "psgpu_hmm_dev.h"
#include "psgpu_lm_dev.h"
#include "psgpu_wave_dev.h"
#include "psgpu_sen_dev.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#if !defined(__HIPCC__)
// the workgroup simulator's build (tests/hostsim) checks the two facts about the active list that the slab layouts' pruning
// relies on instead of reading them (see there), on every node of every frame of every golden it runs
#define PSGPU_FT_CHECK_LISTS 1
#include <cstdio>
#endif

#ifndef PSGPU_FT_THREADS
#define PSGPU_FT_THREADS 256
#endif
constexpr int kFtThreads = PSGPU_FT_THREADS;   // work-items per utterance (LDS layout and the medium slab layout)
#ifndef PSGPU_FT_THREADS_BIG
#define PSGPU_FT_THREADS_BIG 1024
#endif
#ifndef PSGPU_FT_PAIRS
#define PSGPU_FT_PAIRS 4
#endif
#ifndef PSGPU_FT_IPT
#define PSGPU_FT_IPT 2
#endif
#ifndef PSGPU_FT_CHUNK_START
#define PSGPU_FT_CHUNK_START (1 << 30)
#endif
#ifndef PSGPU_FT_CHUNK_ADAPT
#define PSGPU_FT_CHUNK_ADAPT 0
#endif
constexpr int kFtChunkStart = PSGPU_FT_CHUNK_START;    // slab layouts: items of a frame's first pruning chunk (at most what the LDS arrays hold)
constexpr bool kFtChunkAdapt = PSGPU_FT_CHUNK_ADAPT != 0;      // ... and the later chunks sized so that their pairs fill one round
#ifndef PSGPU_FT_TOP_UNROLL
#define PSGPU_FT_TOP_UNROLL 4
#endif
constexpr int kFtTopUnroll = PSGPU_FT_TOP_UNROLL;      // slab layouts: positions of the active list a work-item marks the senones of at a time
#ifndef PSGPU_FT_EVAL_UNROLL
#define PSGPU_FT_EVAL_UNROLL 2
#endif
constexpr int kFtEvalUnroll = PSGPU_FT_EVAL_UNROLL;    // slab layouts: positions of the active list a work-item evaluates at a time (their loads asked for together)
constexpr int kFtIpt = PSGPU_FT_IPT;           // slab layouts: consecutive items (roots / listed nodes) of a pruning chunk a work-item takes to LDS
#ifndef PSGPU_FT_POOL_WORDS_BIG
#define PSGPU_FT_POOL_WORDS_BIG 39808
#endif
constexpr int kFtPairs = PSGPU_FT_PAIRS;       // slab layouts: consecutive pairs of the pruning a work-item decides at a time
constexpr int kFtThreadsBig = PSGPU_FT_THREADS_BIG;    // ... on trees beyond kFtBigNodes
constexpr int kFtBigNodes = 4096;
constexpr int kFtMaxBitWords = 8192;   // slab layouts: a bitmap of the listed tree nodes in LDS for trees up to 32 x this many nodes
constexpr int kFtLdsWords = 16768;     // LDS layout: at most 63 KB of arrays (dynamic LDS, + 2.4 KB fixed) per workgroup when scoring from lists; a launch
                                       // that reads score rows leaves the last 3.6 KB out (FtLay::rows_total): two workgroups per CU then take ~126 of its
                                       // 160 KB and leave the rest to the kernels of other streams that run beside the search
constexpr int kFtListCap = 1024;       // listed senones per frame kept as a list (LDS layout, scoring from top-N lists); more: scored where found
constexpr int kFtMaxChains = 128;      // (codebook, stream) chains whose lists the LDS layout holds
constexpr int kFtMinEvl = 512;         // the frame's evaluation list holds at least this many entries (LDS layout)
constexpr int kFtMaxCi = 64;
constexpr int kFtMaxSen = 8192;        // senones (LDS bitmap of the active list, raw-score mode)
constexpr int kFtSlabSen = 16384;      // slab layouts: senones whose frame row the workgroup copies into LDS (at most 32 KB of the pool)
constexpr int kFtSlabTp = 4096;        // slab layouts: bytes of transition matrices kept in LDS (likewise)
constexpr int kFtSlabPoolWords = 39808;    // slab layouts: words of dynamic LDS a workgroup may ask for (155.5 of a compute unit's 160 KB; ~3.5 KB are static)
constexpr int kFtRcBlk = 64;           // slab layouts: right-context channels in a block of the pool (a word's fan-out: at most n_ci <= 64 of them)
constexpr int kFtMaxRootWords = 128;   // slab layouts: roots <= 32 x this (n_ci <= 64: at most 4,096 (first, second phone) pairs)
constexpr int kFtLbBlock = 1024;       // slab layouts: words of the listed-nodes bitmap per block of 16-bit prefix populations (32,768 nodes: a count fits)
constexpr int kFtLiveMagic = 0x5ea4c4ed;
constexpr int kFtLiveHdr = 32;         // FtBufs::live: words ahead of the pool's copy
constexpr int kFtWordCh = 0x40000000;  // evaluation-list entries that name a right-context channel

// word offsets of the per-utterance arrays the tree level works on ("fast" arrays: LDS in the small layout, the
// utterance's slab otherwise)
struct FtLay {
    int32_t rec;                         // [CH][..] channel records of the tree nodes and the single-phone words
    int32_t acl0, acl1, awl0, awl1;      // [N], [n_w] active lists (this frame / next frame)
    int32_t word_active, word_lat_idx, lt_sf, lt_dscr, lt_bp, cand_mark;    // [n_w]
    int32_t cand_wid, cand_score, cand_bp;                                  // [n_w + 1]
    int32_t o_out, o_outh, pos, flag, o_frame;                              // [N] pruning snapshot / decisions
    int32_t node_blk, word_blk;          // small layout: the interleaved per-node and per-word arrays (FtCol; the fields above that name their
                                         // columns are unused there)
    int32_t xfr, xfr_cap;                // small layout: the frame's exits [xfr_cap][n_ci + 3] = {word, real word id, previous real word id, score per
                                         // right context phone}, in the evaluation list's words (idle after prune_word_chan)
    int32_t cnt;                         // [cnt_words] scan scratch
    int32_t cnt2, cnt3, woff;            // [n_w + 2] per-candidate / per-active-word scratch; first slot index of each active word
    int32_t ckey;                        // [n_w + 2] 64-bit (score, back-pointer) keys of the pair searches
    int32_t present;                     // bytes: right-context channel allocated (ngram_search_alloc_all_rc / _free_all_rc); LDS layout: [TOT], a
                                         // word's slots side by side; slab layouts: [rc_blocks][kFtRcBlk], by pool block
    int32_t wblk, rcfree;                // slab layouts: [n_w] the pool block that holds the word's right-context channels, or -1; [rc_blocks] free blocks
    int32_t l_cw, l_sc, l_la, l_list, l_norm;    // small layout, scoring from top-N lists: the frame's lists (packed codewords / scores per
                                         // chain), log-add table (512 bytes), listed senones (uint16), per-wavefront stream maxima
    int32_t itb;                         // slab layouts: [R][4] per root: out, out history, best, 1 = evaluated this frame, as the evaluation left them
    // slab layouts: the listed tree nodes' channels, COMPACT and in list order (see the kernel's note "compact channels"):
    int32_t cq;                          // [2][ND + 2][ccap][4] two buffers taking turns by frame parity; per buffer ND arrays of quads with the channel's
                                         // scores, histories, out score, out history (ND = 2 for 3 states, 3 for 5) and two with what is static per node
    int32_t csum;                        // [ccap][4] per list position: out, out history, best, score[0] as the evaluation left them
    int32_t cxfer;                       // [2][ccap] per list position (two lists taking turns like the buffers): where the node's channel comes from -- (its
                                         // position in the frame before's list + 1, or 0: a new channel) | what the pruning did to it << 28
    int32_t cxpl;                        // [2][ccap][2] ... and, if the pruning entered it, the entering score and history
    int32_t cperm;                       // [ccap] rank among the listed nodes by node id -> list position, when the frame's list outgrows the LDS table
    int32_t ccap;                        // listed nodes a frame may hold: N - R, every node but the roots
    int32_t evl, evl_cap;                // [evl_cap] the frame's evaluation list (small layout: what the pool has left)
    int32_t wc_off;                      // small layout: copy of the words' first right-context slot
    int32_t row, pen;                    // small layout: the frame's score row (int16) and two penalty rows
    int32_t rows_total;                  // small layout: words of the pool a launch that reads score ROWS needs (row and l_* lie behind)
    int32_t kid_off, kids, parent, ci, pw;     // small layout: copies of the static tree tables
    int32_t dfirst, dbase, w1w;          // small layout: copies of the words' first phone and base word id [n_w], the single-phone words' ids [n1]
    int32_t dfill;                       // small layout: copy of the words' filler flags [n_w]
    int32_t dlast, homo, w1ci, w1ci2;    // small layout: copies of the words' last phone and homophone link [n_w], the single-phone words' phones [n1]
    int32_t tp;                          // small layout: copy of the transition matrices (bytes)
    int32_t total;
};

struct FtDev {
    int32_t n_ci, n_emit, n_sen, n_w, R, M, N, n1, n1lm, TOT, CH;
    int32_t beam, pbeam, lpbeam, lponlybeam, wbeam, pip, nwpen, silpen, fillpen, maxhmmpf, maxwpf;
    int32_t startwid, finishwid, silwid, filler_start, filler_end, sil_ci, has_pl;
    const int32_t *node_ci, *node_ci2, *node_ssid, *node_tmat, *node_pw, *parent, *kid_off, *kids;
    const int32_t *kids_ci;              // slab layouts: per node its children (child | ci << 24), then its penultimate-phone words (word | last phone << 24)
    // slab layouts: what the pruning and the senone marking ask about a node, as quads (one request each), shared by all
    // utterances (8 MB at 248 k nodes: L2 / MALL resident):
    const int32_t *node_q1;              // [N][4] parent | ci << 24, the node's first entry in kids_ci, children | penultimate-phone words << 16, that first
                                         // entry or -1
    const int32_t *node_st1;             // [N][4] the node's senone ids and transition matrix as a compact channel carries them: 16 bits each (3 states:
                                         // s0 | s1 << 16, s2 | tmat << 16, 0; 5 states: s0 | s1 << 16, s2 | s3 << 16, s4 | tmat << 16), then kids_ci[first child]
    const int32_t *node_q2;              // [N][4] penultimate-phone word, its last phone, its homophone link, 0
    const int32_t *node_sen;             // [N][4 or 8] the node's senone ids (sseq[node_ssid]): states 0..n_emit-1, then 0
    const int32_t *slot_sen;             // [TOT][4 or 8] a right-context channel's senone ids (what ngram_search_alloc_all_rc gives it: sseq of
                                         // the rssid of its word's last two phones), then 0, its transition matrix in the last word
    const int32_t *homophone, *w1_wid, *w1_ci, *w1_ci2, *w1_ssid, *w1_tmat, *w1_mpx, *w1_of_word;
    const int32_t *d_pronlen, *d_first, *d_last, *d_last2, *d_base, *d_filler;
    const int32_t *rs_n, *rs_ssid, *rs_cimap, *ldiph, *ci_tmat, *lm, *wc_off;
    const uint8_t *tp;
    const uint16_t *sseq;
    int32_t n_tmat;
    int32_t small;                       // the fast arrays fit the LDS pool
    int32_t lb_words;                    // slab layouts: words of the listed-nodes bitmap in LDS
    // slab layouts: the dynamic LDS pool behind the pruning's item arrays (word offsets; set per launch, ft_slab_pool): the bitmap, its
    // words' prefix populations (uint16), the frame's score row, the transition matrices, the rank -> list position table (uint16)
    int32_t lds_lb, lds_pre, lds_row, lds_tp, lds_perm, lds_perm_cap, lds_words;
    int32_t use_trie;                    // language scores from the trie (psgpu_fwdtree_set_lm) instead of the dense table
    int32_t cnt_words;
    // slab layouts, capacities that grow on demand (psgpu_fwdtree_grow): tree nodes a frame may list (status 4 when a frame wants
    // more), blocks of the right-context channels' pool (status 5 when a frame needs one and none is free)
    int32_t listed_cap, rc_blocks;
    int32_t wl_global;                   // slab layouts: the word level's scratch arrays in the slab (a frame's counts outgrew the LDS arrays: status 6)
    int32_t wl_cap;                      // ... the LDS arrays' capacity in words, if less than what the pool gives (PSGPU_FWDTREE_WL_CAP: a test's knob)
    FtLay lay;
    // always in the utterance's slab (int32 units from its start): last-phone channel records; `fast`: the FtLay arrays
    // when they are not in LDS
    int64_t g_wrec, g_fast, per;
    const LmDev *trie_dev;               // the trie's descriptor in device memory
};

struct FtBufs {
    int32_t *slab, *bp, *bss, *idx, *step, *res, *w1_out;
    int32_t *bpa;                        // [n_utt][bp_cap][kBpRow] the back-pointer tables while the search runs (FtTab); bp: the caller's columns
    int32_t *bssx;                       // LDS layout: [n_utt][bp_cap][n_ci] an exit's score for every right context phone (the score stack read through
                                         // the context map once, when the entry is written: see the kernel's exits step), or NULL
    // scoring from the scorer's top-N lists instead of score rows (psgpu_fwdtree_search_lists_dev): tsc == NULL = rows
    const int32_t *tsc;                  // [chain][total][4] raw scores, chain-major
    const uint32_t *tcw;                 // [chain][total] four codewords packed
    const uint8_t *mixw, *sen2cb, *la;   // the scorer's mixture weights [3][n_density][n_sen], senone -> codebook, 8-bit log-add table
    int32_t ls_total, ls_chains, ls_density, ls_la_size;
    const int32_t *mpx_in;               // session state: per utterance [(R + n1)][n_emit] per-state ssids of the multiplexed channels, or NULL
    int32_t *mpx_out;
    int32_t *hyp, *hyp_n;                // the hypotheses, written by the kernel's last step (NULL: not wanted; see psgpu_fwdtree_hyp_out)
    int32_t max_words;
    long long *prof;                     // PSGPU_FT_PROFILE builds: [n_utt][32] cycles per phase (tools/build_prof_lib.py)
    int32_t bp_cap, bss_cap, max_frames;
    int32_t lag;                         // search all but the last `lag` frames of every utterance (psgpu_fwdtree_search_lag); 0: all
    // a search that goes on where the handle's previous call stopped (psgpu_fwdtree_search_resume; LDS layout): per utterance
    // kFtLiveHdr words {frames searched, the frame loop's carried registers, the counters} + the LDS pool as the last frame left it.
    // The tables, the score stack, the frame marks, bpa / bssx and the last-phone channels' slab are the caller's / the handle's and
    // stay where they are between the calls
    int32_t *live;
    int32_t live_mode;                   // bit 0: save the state when the call stops; bit 1: start from the saved state
    // psgpu_fwdtree_search_streams: per utterance {frames scored so far, frame the search goes on to, whether an utterance that STARTS
    // in this call takes its multiplexed channels' ssids from mpx_in (a decoder's next utterance) or keeps a new decoder's} instead of
    // back-to-back offsets and one lag: utterances in progress that grow at their own pace; utt_off [u] alone places the utterance's rows (row of frame f
    // at (utt_off[u] + f) * stride: a caller that keeps only the frames not yet searched passes a start before its buffer)
    const int32_t *ext;
};

struct psgpu_fwdtree_s {
    FtDev d;
    std::vector<void *> allocs;
    int32_t *slab = nullptr;             // work slab, kept between calls (grown on demand)
    size_t slab_words = 0;
    int32_t *bssx = nullptr;             // LDS layout: the exits' scores by right context phone (FtBufs::bssx), likewise
    size_t bssx_words = 0;
    int32_t *bpa = nullptr;              // the tables as entries while a search runs (FtBufs::bpa), likewise
    size_t bpa_words = 0;
    int32_t *hyp_out = nullptr, *hyp_n_out = nullptr;    // psgpu_fwdtree_hyp_out: for the NEXT search call only
    int32_t hyp_max_words = 0;
    int32_t lag_next = 0;                // psgpu_fwdtree_search_lag: for the NEXT search call only
    // psgpu_fwdtree_search_resume
    const int32_t *ext_next = nullptr;   // psgpu_fwdtree_search_streams: the NEXT search call's FtBufs::ext
    int32_t live_next = 0;               // the NEXT search call's FtBufs::live_mode
    int32_t *live = nullptr;             // FtBufs::live, kept between calls
    size_t live_words = 0;
    bool live_valid = false;             // the latest search call saved its state ...
    int32_t live_small = 0, live_n_utt = 0, live_bp_cap = 0, live_bss_cap = 0, live_max_frames = 0, live_raw = 0, live_window = 0;    // ... for these
    int32_t *live_bp = nullptr, *live_bss = nullptr, *live_idx = nullptr, *live_step = nullptr;
#ifdef PSGPU_FT_PROFILE
    long long *prof_buf = nullptr; const int32_t *prof_res = nullptr; int32_t prof_n_utt = 0; hipEvent_t prof_ev = nullptr;
#endif
};

// ---- channel records ---------------------------------------------------------------------------------------------
// One record per HMM instance = the fields of hmm_t (hmm.h:169-182).  Addressed as base[c * cst + field * fst]: the
// last-phone channels (global memory) are arrays of records (cst = record size, fst = 1: a record is one cache line);
// the tree's channels in LDS are a structure of arrays (cst = 1, fst = number of channels: work-items on consecutive
// channels hit consecutive banks).
template <int NE> struct ChF {
    // records in device memory are read and written four words at a time: what an evaluation rewrites is [0, FRAME], what the
    // pruning and the exits look at is the quad [OUT, FRAME], the senone ids are the last quad(s)
    static constexpr int SCORE = 0, HIST = NE, TMAT = 2 * NE, MPX = 2 * NE + 1, OUT = 2 * NE + 2, OUTH = 2 * NE + 3,
                         BEST = 2 * NE + 4, FRAME = 2 * NE + 5, SENID = 2 * NE + 6, WORDS = 3 * NE + 6,
                         REC = NE == 3 ? 16 : 24;
    static_assert(OUT % 4 == 0 && SENID % 4 == 0 && REC % 4 == 0 && WORDS <= REC, "quads");
};
struct alignas(16) FtQuad { int32_t x, y, z, w; };
struct alignas(8) FtPair { int32_t x, y; };
// One column of a block of interleaved arrays (LDS layout: AOS, element i at word i * K of the column's base) or a plain array
// (slab layouts).  Why interleave: this kernel's speed follows the number of scalar values it keeps alive -- the compiler gives
// every array's base a scalar register, spills what does not fit into vector-register lanes and reads it back with v_readlane
// wherever it is used (1,600 such reads in the LDS layout's kernel; a build with the LDS offsets as compile-time constants ran
// 6 % faster).  The columns of a block share ONE base; a column's offset is an immediate of the LDS instruction.  K is odd: work-items
// on consecutive elements hit different banks.
template <bool AOS, int K, typename T = int32_t>
struct FtCol {
    T *b;
    __device__ __forceinline__ T &operator[](int i) const { return AOS ? b[i * K] : b[i]; }
};
// the blocks' columns: per tree node / list position (N + 1 rows) and per dictionary word (n_w + 2 rows)
enum { NC_ACL0, NC_ACL1, NC_POS, NC_OOUT, NC_OOUTH, NC_FLAG, NC_OFRAME, NC_KIDOFF, NC_PARENT, NC_CI, NC_PW, kFtNodeCols };
enum { WC_AWL0, WC_AWL1, WC_ACTIVE, WC_LATIDX, WC_LTSF, WC_LTDSCR, WC_LTBP, WC_CMARK, WC_CWID, WC_CSCORE, WC_CBP, WC_WCOFF, WC_DFIRST, WC_DBASE,
       WC_DLAST, WC_HOMO, WC_DFILL, kFtWordCols };
static_assert(kFtNodeCols % 2 == 1 && kFtWordCols % 2 == 1, "odd strides");
// Slab layouts (tree state in device memory).  What a workgroup pays there is the bytes and cache lines it touches per frame
// (at 256 utterances of the 134,865-word task the kernel is bound by the memory system): the roots and the single-phone words
// keep 64-byte records (ChF); every other tree node's channel exists while the node is listed, COMPACT, at the node's position
// in the frame's active list (see "compact channels" in the kernel), and the right-context channels come from a pool of blocks.
// What the pruning needs about a node is either carried by its compact channel or static and shared by all utterances
// (FtDev::node_q1 / node_st1 / kids_ci), and FtLay::itb holds, per root, {out, out history, best, evaluated} as the evaluation
// left them.  A decision only reads; its outcome is written for the place it gives in the next list (FtLay::cxfer).
struct ChView {
    int32_t *b;
    int cst, fst;
    __device__ __forceinline__ int32_t &at(int c, int field) const { return b[c * cst + field * fst]; }
};

template <int NE>
__device__ __forceinline__ void ch_clear(const ChView &v, int c)                 // hmm_clear, hmm.c:181-196
{
    using F = ChF<NE>;
#pragma unroll
    for (int i = 0; i < NE; ++i) { v.at(c, F::SCORE + i) = kW; v.at(c, F::HIST + i) = -1; }
    v.at(c, F::OUT) = kW; v.at(c, F::OUTH) = -1; v.at(c, F::BEST) = kW; v.at(c, F::FRAME) = -1;
}
template <int NE>
__device__ __forceinline__ void ch_init(const ChView &v, int c, int mpx, int ssid, int tmatid, const uint16_t *sseq)   // hmm_init :146-168
{
    using F = ChF<NE>;
    v.at(c, F::MPX) = mpx; v.at(c, F::TMAT) = tmatid;
    if (mpx) {
        v.at(c, F::SENID) = ssid;
#pragma unroll
        for (int i = 1; i < NE; ++i) v.at(c, F::SENID + i) = kBadSsid;
    }
    else {
#pragma unroll
        for (int i = 0; i < NE; ++i) v.at(c, F::SENID + i) = sseq[(size_t)ssid * NE + i];
    }
    ch_clear<NE>(v, c);
}
// hmm_init of a plain (not multiplexed) channel followed by hmm_enter, on a record in device memory: written as quads
template <int NE>
__device__ __forceinline__ void ch_init_enter_rec(int32_t *rec, int ssid, int tmatid, const uint16_t *sseq, int32_t score, int32_t hist, int frame)
{
    using F = ChF<NE>;
    constexpr int NQ = (F::WORDS + 3) / 4;
    int32_t w[4 * NQ];
#pragma unroll
    for (int i = 0; i < 4 * NQ; ++i) w[i] = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) { w[F::SCORE + i] = kW; w[F::HIST + i] = -1; w[F::SENID + i] = sseq[(size_t)ssid * NE + i]; }
    w[F::OUT] = kW; w[F::OUTH] = -1; w[F::BEST] = kW; w[F::MPX] = 0; w[F::TMAT] = tmatid;
    w[F::SCORE] = score; w[F::HIST] = hist; w[F::FRAME] = frame;
    FtQuad *dst = reinterpret_cast<FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NQ; ++k) dst[k] = FtQuad{ w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
}
// the same with the channel's senone ids and transition matrix given (FtDev::slot_sen: one static quad or two instead of the chain
// last two phones -> rssid -> sseq, three dependent trips to device memory)
template <int NE>
__device__ __forceinline__ void ch_init_enter_rec_s(int32_t *rec, const int32_t *slot_sen, int32_t score, int32_t hist, int frame)
{
    using F = ChF<NE>;
    constexpr int NQ = (F::WORDS + 3) / 4, NS = NE <= 3 ? 4 : 8;
    int32_t sq[NS];
    const FtQuad *src = reinterpret_cast<const FtQuad *>(slot_sen);
#pragma unroll
    for (int k = 0; k < NS / 4; ++k) { const FtQuad q = src[k]; sq[4 * k] = q.x; sq[4 * k + 1] = q.y; sq[4 * k + 2] = q.z; sq[4 * k + 3] = q.w; }
    int32_t w[4 * NQ];
#pragma unroll
    for (int i = 0; i < 4 * NQ; ++i) w[i] = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) { w[F::SCORE + i] = kW; w[F::HIST + i] = -1; w[F::SENID + i] = sq[i]; }
    w[F::OUT] = kW; w[F::OUTH] = -1; w[F::BEST] = kW; w[F::MPX] = 0; w[F::TMAT] = sq[NS - 1];
    w[F::SCORE] = score; w[F::HIST] = hist; w[F::FRAME] = frame;
    FtQuad *dst = reinterpret_cast<FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NQ; ++k) dst[k] = FtQuad{ w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
}
template <int NE>
__device__ __forceinline__ void ch_enter(const ChView &v, int c, int32_t score, int32_t hist, int frame)   // hmm_enter :198-204
{
    using F = ChF<NE>;
    v.at(c, F::SCORE) = score; v.at(c, F::HIST) = hist; v.at(c, F::FRAME) = frame;
}
template <int NE>
__device__ __forceinline__ void ch_normalize(const ChView &v, int c, int32_t norm)      // hmm_normalize :206-217
{
    using F = ChF<NE>;
#pragma unroll
    for (int i = 0; i < NE; ++i) if (v.at(c, F::SCORE + i) > kW) v.at(c, F::SCORE + i) -= norm;
    if (v.at(c, F::OUT) > kW) v.at(c, F::OUT) -= norm;
}
// A 3-state model's transition matrix (12 bytes, 4-byte aligned: the tables are LDS copies that start on a word) as three word
// reads instead of the eight byte reads the Viterbi step would make of it
template <int NE>
__device__ __forceinline__ const uint8_t *ft_tp_row(const uint8_t *tpall, int tmat, uint8_t (&buf)[NE * (NE + 1)])
{
    if (NE == 3) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(tpall + (size_t)tmat * 12);
        const uint32_t a = q[0], b = q[1], c = q[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { buf[k] = (uint8_t)(a >> (8 * k)); buf[4 + k] = (uint8_t)(b >> (8 * k)); buf[8 + k] = (uint8_t)(c >> (8 * k)); }
        return buf;
    }
    return tpall + (size_t)tmat * NE * (NE + 1);
}
// hmm_vit_eval on channel c with the frame's scores
template <int NE, typename S>
__device__ __forceinline__ int32_t ch_eval(const ChView &v, int c, const S &row, const uint8_t *tpall, const uint16_t *sseq)
{
    using F = ChF<NE>;
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        h.score[i] = i < NE ? v.at(c, F::SCORE + i) : kW;
        h.history[i] = i < NE ? v.at(c, F::HIST + i) : -1;
        h.senid[i] = i < NE ? (uint16_t)v.at(c, F::SENID + i) : 0;
    }
    h.out_score = v.at(c, F::OUT); h.out_history = v.at(c, F::OUTH); h.bestscore = v.at(c, F::BEST);
    uint8_t tpb[NE * (NE + 1)];
    const uint8_t *tp = ft_tp_row<NE>(tpall, v.at(c, F::TMAT), tpb);
    const int mpx = v.at(c, F::MPX);
    int32_t b;
    if (NE == 3) b = mpx ? vit3_mpx(h, tp, row, sseq) : vit3(h, tp, row);
    else         b = mpx ? vit5_mpx(h, tp, row, sseq) : vit5(h, tp, row);
#pragma unroll
    for (int i = 0; i < NE; ++i) { v.at(c, F::SCORE + i) = h.score[i]; v.at(c, F::HIST + i) = h.history[i]; }
    if (mpx) {
#pragma unroll
        for (int i = 0; i < NE; ++i) v.at(c, F::SENID + i) = h.senid[i];
    }
    v.at(c, F::OUT) = h.out_score; v.at(c, F::OUTH) = h.out_history; v.at(c, F::BEST) = h.bestscore;
    return b;
}
// the same on a record in device memory (16-byte aligned, ChF<NE>::REC words): the record comes in and goes out as quads
// -- a work-item's record is one or two cache lines, and 64 work-items asking for it word by word are 64 requests per word
template <int NE, typename S>
__device__ __forceinline__ int32_t ch_eval_rec(int32_t *rec, const S &row, const uint8_t *tpall, const uint16_t *sseq)
{
    using F = ChF<NE>;
    constexpr int NQ = (F::WORDS + 3) / 4, NW = F::SENID / 4;       // quads read; quads an evaluation rewrites ([0, FRAME])
    int32_t w[4 * NQ];
    const FtQuad *src = reinterpret_cast<const FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NQ; ++k) { const FtQuad q = src[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        h.score[i] = i < NE ? w[F::SCORE + i] : kW;
        h.history[i] = i < NE ? w[F::HIST + i] : -1;
        h.senid[i] = i < NE ? (uint16_t)w[F::SENID + i] : 0;
    }
    h.out_score = w[F::OUT]; h.out_history = w[F::OUTH]; h.bestscore = w[F::BEST];
    uint8_t tpb[NE * (NE + 1)];
    const uint8_t *tp = ft_tp_row<NE>(tpall, w[F::TMAT], tpb);
    const int mpx = w[F::MPX];
    int32_t b;
    if (NE == 3) b = mpx ? vit3_mpx(h, tp, row, sseq) : vit3(h, tp, row);
    else         b = mpx ? vit5_mpx(h, tp, row, sseq) : vit5(h, tp, row);
#pragma unroll
    for (int i = 0; i < NE; ++i) { w[F::SCORE + i] = h.score[i]; w[F::HIST + i] = h.history[i]; }
    w[F::OUT] = h.out_score; w[F::OUTH] = h.out_history; w[F::BEST] = h.bestscore;
    FtQuad *dst = reinterpret_cast<FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[k] = FtQuad{ w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
    if (mpx) {
#pragma unroll
        for (int i = 0; i < NE; ++i) rec[F::SENID + i] = h.senid[i];
    }
    return b;
}
// The same for a TREE channel of the slab layouts: a non-root node's FRAME word receives `posword` (its position + 1 in the
// active list: see the layouts' note above), and the four words the pruning's item phase wants come back in `item`
template <int NE, typename S>
__device__ __forceinline__ int32_t ch_eval_tree(int32_t *rec, const S &row, const uint8_t *tpall, const uint16_t *sseq, bool set_pos,
                                                int32_t posword, FtQuad &item)
{
    using F = ChF<NE>;
    constexpr int NQ = (F::WORDS + 3) / 4, NW = F::SENID / 4;
    int32_t w[4 * NQ];
    const FtQuad *src = reinterpret_cast<const FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NQ; ++k) { const FtQuad q = src[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    HmmRegs h;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        h.score[i] = i < NE ? w[F::SCORE + i] : kW;
        h.history[i] = i < NE ? w[F::HIST + i] : -1;
        h.senid[i] = i < NE ? (uint16_t)w[F::SENID + i] : 0;
    }
    h.out_score = w[F::OUT]; h.out_history = w[F::OUTH]; h.bestscore = w[F::BEST];
    uint8_t tpb[NE * (NE + 1)];
    const uint8_t *tp = ft_tp_row<NE>(tpall, w[F::TMAT], tpb);
    const int mpx = w[F::MPX];
    int32_t b;
    if (NE == 3) b = mpx ? vit3_mpx(h, tp, row, sseq) : vit3(h, tp, row);
    else         b = mpx ? vit5_mpx(h, tp, row, sseq) : vit5(h, tp, row);
#pragma unroll
    for (int i = 0; i < NE; ++i) { w[F::SCORE + i] = h.score[i]; w[F::HIST + i] = h.history[i]; }
    w[F::OUT] = h.out_score; w[F::OUTH] = h.out_history; w[F::BEST] = h.bestscore;
    if (set_pos) w[F::FRAME] = posword;
    FtQuad *dst = reinterpret_cast<FtQuad *>(rec);
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[k] = FtQuad{ w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
    if (mpx) {
#pragma unroll
        for (int i = 0; i < NE; ++i) rec[F::SENID + i] = h.senid[i];
    }
    item = FtQuad{ h.out_score, h.out_history, h.bestscore, h.score[0] };
    return b;
}
// [OUT, OUTH, BEST, FRAME] of a record in device memory
template <int NE>
__device__ __forceinline__ FtQuad ch_summary(const int32_t *rec) { return *reinterpret_cast<const FtQuad *>(rec + ChF<NE>::OUT); }

// ---- the back-pointer table ------------------------------------------------------------------------------------------
// While a search runs its table is an array of ENTRIES, sixteen words (one 64-byte line) each: everything a step asks about an
// entry -- frame, score, last phones; validity and real word ids -- comes in one line and through one base pointer, a column's offset
// being an immediate of the load (in the reference's column arrays, bptbl_t, ngram_search.h:112-124, an entry's fields are ten
// lines behind ten base pointers).  The kernel's last step copies the entries into the caller's columns (ft_table_out).
constexpr int kBpRow = 16;
struct FtTab {
    int32_t *bp, *bss, *idx;             // bp: [bp_cap][kBpRow]
    int32_t bp_cap, bss_cap;
};
#define BPC(t, col, i) ((t).bp[(size_t)(i) * kBpRow + (col)])
enum { B_FRAME, B_VALID, B_WID, B_BP, B_SCORE, B_SIDX, B_REAL, B_PREAL, B_LAST, B_LAST2, kBpCols,
       // words of an entry that are not columns of the caller's table: the first entry of its frame and of the next frame (what the
       // predecessor search wants of the frame index, bp_table_idx, without a trip of its own)
       B_F0 = kBpCols, B_F1 };

__device__ __forceinline__ int32_t ft_lm(const FtDev &p, const int32_t *lmtab, int w3, int w2, int w1)
{
    // ngram_tg_score(...) >> SENSCR_SHIFT, ngram_search_fwdtree.c:1118, :1342
    if (p.use_trie) return lm_tg_score_call(p.trie_dev, w3, w2, w1) >> 10;
    const size_t n1 = (size_t)p.n_w + 1;
    return lmtab[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}
// ngram_search_exit_score, ngram_search.c:653-674
__device__ __forceinline__ int32_t ft_exit_score(const FtTab &t, const int32_t *rs_cimap, int n_ci, int bp, int rcphone)
{
    const int l2 = BPC(t, B_LAST2, bp);
    if (l2 == -1) return BPC(t, B_SCORE, bp);
    const int l1 = BPC(t, B_LAST, bp);
    return t.bss[BPC(t, B_SIDX, bp) + rs_cimap[((size_t)l1 * n_ci + l2) * n_ci + rcphone]];
}
// The same without a branch between its loads: every address is valid whatever the entry holds (single-phone entries: the context
// map is read at context 0 and the score stack at the table's start, then dropped), so the entry's four columns come back in ONE
// trip to device memory, the context map in a second, the stacked score in a third -- as written above the compiler must wait for
// last2 before it may ask for anything else, and a caller's `if (!valid) continue` in front adds another trip: a frame's pair
// searches were chains of five to seven dependent trips.
__device__ __forceinline__ int32_t ft_exit_score_bf(const FtTab &t, const int32_t *rs_cimap, int n_ci, int bp, int rcphone)
{
    const int32_t score = BPC(t, B_SCORE, bp), l2 = BPC(t, B_LAST2, bp), l1 = BPC(t, B_LAST, bp), sidx = BPC(t, B_SIDX, bp);
    const int32_t cm = rs_cimap[((size_t)l1 * n_ci + max(l2, 0)) * n_ci + rcphone];
    const int32_t ss = t.bss[max(sidx, 0) + max(cm, 0)];
    return l2 == -1 ? score : ss;
}
// The same through the table of the exits' scores by right context phone (FtBufs::bssx, LDS layout): the entry's columns and its
// score for `rcphone` in ONE trip -- the context map was applied when the entry was written.  Single-phone entries (last2 == -1)
// have no row: what is read there is dropped.
__device__ __forceinline__ int32_t ft_exit_score_x(const FtTab &t, const int32_t *bssx, int n_ci, int bp, int rcphone)
{
    const int32_t score = BPC(t, B_SCORE, bp), l2 = BPC(t, B_LAST2, bp);
    const int32_t xs = bssx[(size_t)bp * n_ci + rcphone];
    return l2 == -1 ? score : xs;
}
// the dense language-model table's entry (p.use_trie == 0), an unconditional load
__device__ __forceinline__ int32_t ft_lm_dense(const FtDev &p, const int32_t *lmtab, int w3, int w2, int w1)
{
    const size_t n1 = (size_t)p.n_w + 1;
    return lmtab[((size_t)w3 * n1 + (size_t)(w2 + 1)) * n1 + (size_t)(w1 + 1)];
}
// set_real_wid, ngram_search.c:341-372
__device__ __forceinline__ void ft_set_real_wid(const FtTab &t, const int32_t *d_filler, const int32_t *d_base, int bp)
{
    const int prev = BPC(t, B_BP, bp), wid = BPC(t, B_WID, bp);
    if (d_filler[wid]) {
        if (prev != -1) { BPC(t, B_REAL, bp) = BPC(t, B_REAL, prev); BPC(t, B_PREAL, bp) = BPC(t, B_PREAL, prev); }
        else { BPC(t, B_REAL, bp) = d_base[wid]; BPC(t, B_PREAL, bp) = -1; }
    }
    else {
        BPC(t, B_REAL, bp) = d_base[wid];
        BPC(t, B_PREAL, bp) = prev != -1 ? BPC(t, B_REAL, prev) : -1;
    }
}
// the static per-word tables save_bp reads
struct FtDict { const int32_t *d_pronlen, *d_last, *d_last2, *d_base, *d_filler, *rs_n; int n_ci; };
// ngram_search_save_bp, ngram_search.c:376-498 (single thread).  Returns false when a table is full.
template <typename WL>
__device__ __forceinline__ bool ft_save_bp(const FtTab &t, const FtDict &d, const WL &word_lat_idx, int32_t &bpidx, int32_t &bss_head,
                                           int frame, int w, int32_t score, int32_t path, int rc)
{
    const int bp = word_lat_idx[w];
    if (bp != -1) {
        if (BPC(t, B_SCORE, bp) < score) {
            const int ob = BPC(t, B_BP, bp);
            if (ob != path) {
                const int32_t b0 = ob == -1 ? -1 : BPC(t, B_PREAL, ob), b1 = ob == -1 ? -1 : BPC(t, B_REAL, ob);
                const int32_t n0 = path == -1 ? -1 : BPC(t, B_PREAL, path), n1 = path == -1 ? -1 : BPC(t, B_REAL, path);
                if (b0 != n0 || b1 != n1) ft_set_real_wid(t, d.d_filler, d.d_base, bp);      // with the old bp still in place, as the reference
                BPC(t, B_BP, bp) = path;
            }
            BPC(t, B_SCORE, bp) = score;
        }
        if (BPC(t, B_SIDX, bp) != -1) t.bss[BPC(t, B_SIDX, bp) + rc] = score;
        return true;
    }
    if (bpidx >= t.bp_cap || bss_head + d.n_ci >= t.bss_cap) return false;
    word_lat_idx[w] = bpidx;
    BPC(t, B_WID, bpidx) = w; BPC(t, B_FRAME, bpidx) = frame; BPC(t, B_BP, bpidx) = path; BPC(t, B_SCORE, bpidx) = score;
    BPC(t, B_SIDX, bpidx) = bss_head; BPC(t, B_VALID, bpidx) = 1;
    BPC(t, B_LAST, bpidx) = d.d_last[w];
    int rcsize = 0;
    if (d.d_pronlen[w] == 1) { BPC(t, B_LAST2, bpidx) = -1; BPC(t, B_SIDX, bpidx) = -1; }
    else {
        BPC(t, B_LAST2, bpidx) = d.d_last2[w];
        rcsize = d.rs_n[d.d_last[w] * d.n_ci + d.d_last2[w]];
    }
    for (int i = 0; i < rcsize; ++i) t.bss[bss_head + i] = kW;
    if (rcsize) t.bss[bss_head + rc] = score;
    ft_set_real_wid(t, d.d_filler, d.d_base, bpidx);
    ++bpidx;
    bss_head += rcsize;
    return true;
}

// ngram_search_find_exit (ngram_search.c:500-544, frame_idx = -1) + the walk of ngram_search_bp_hyp / the segment iterator
// (:546-581, 903-1010) over ONE utterance's table, by one work-item.  hyp [max_words][4] = wid, start frame, end frame, path
// score at the word's end, in spoken order; hn [4] = number of words (may exceed max_words: then only the LAST max_words are
// stored), path score of the exit, exit back-pointer, 0.
// (TB: a table whose fields BPX(tb, col, i) reads -- the kernel's entries or the caller's columns)
struct FtTabCols { const int32_t *bp; int32_t bp_cap; };
__device__ __forceinline__ int32_t ft_tab_get(const FtTab &t, int col, int i) { return BPC(t, col, i); }
__device__ __forceinline__ int32_t ft_tab_get(const FtTabCols &t, int col, int i) { return t.bp[(size_t)col * t.bp_cap + i]; }
template <typename TB>
__device__ __forceinline__ void ft_backtrace_one(const TB &tb, const int32_t *idx, int n_frame, int finish_wid, int max_words,
                                                 int32_t *hyp, int32_t *hn)
{
#define BPX(t, col, i) ft_tab_get(t, col, i)
    hn[0] = 0; hn[1] = kW; hn[2] = -1; hn[3] = 0;
    if (n_frame == 0) return;
    int f = n_frame - 1;
    const int end = idx[f];
    while (f >= 0 && idx[f] == end) --f;
    if (f < 0) return;
    int best = -1; int32_t best_score = kW;
    for (int bp = idx[f]; bp < end; ++bp) {
        const int wid = BPX(tb, B_WID, bp);
        if (wid == finish_wid || BPX(tb, B_SCORE, bp) > best_score) { best_score = BPX(tb, B_SCORE, bp); best = bp; }
        if (wid == finish_wid) break;
    }
    int n = 0;
    for (int b = best; b != -1; b = BPX(tb, B_BP, b)) ++n;
    hn[0] = n; hn[1] = best_score; hn[2] = best;
    int k = n - 1;
    const int skip = n > max_words ? n - max_words : 0;
    for (int b = best; b != -1 && k >= skip; --k) {
        const int prev = BPX(tb, B_BP, b);
        int32_t *h = hyp + (size_t)(k - skip) * 4;
        h[0] = BPX(tb, B_WID, b); h[1] = prev == -1 ? 0 : BPX(tb, B_FRAME, prev) + 1; h[2] = BPX(tb, B_FRAME, b); h[3] = BPX(tb, B_SCORE, b);
        b = prev;
    }
#undef BPX
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS operations, not for its outstanding
// device-memory loads and stores (which __syncthreads() drains: the score-row prefetch would never survive a phase, and
// every store would be waited for ~50 times a frame).  Where work-items exchange data through DEVICE memory (the
// right-context channels' records, the back-pointer table) the kernel keeps __syncthreads().  LDS = false: everything
// is in device memory (slab layout), every barrier is a full one.
template <bool LDS>
__device__ __forceinline__ void ft_sync()
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
#else
    __syncthreads();
#endif
}

// Exclusive prefix sum of a[0..n) in place by the whole workgroup (a written before a barrier); returns the total to
// every thread.  tmp: NT / 64 words of LDS.  Ends with a barrier.
template <int NT, bool LDS>
__device__ __forceinline__ int32_t ft_block_scan(int32_t *a, int n, int32_t *tmp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (n <= 64) {                       // one wavefront, one barrier (most of the word level's lists are this short)
        if (tid < 64) {
            const int32_t v = lane < n ? a[lane] : 0;
            const int32_t incl = ft_wave_incl<FtAdd>(v);
            if (lane < n) a[lane] = incl - v;
            if (lane == 63) tmp[0] = incl;
        }
        ft_sync<LDS>();
        return tmp[0];
    }
    const int per = (n + NT - 1) / NT;
    const int b = min(n, tid * per), e = min(n, b + per);
    int32_t sum = 0;
    for (int i = b; i < e; ++i) sum += a[i];
    const int32_t incl = ft_wave_incl<FtAdd>(sum);
    if (lane == 63) tmp[tid >> 6] = incl;
    ft_sync<LDS>();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int32_t t = tmp[w]; total += t; if (w < (tid >> 6)) base += t; }
    int32_t run = base + incl - sum;
    for (int i = b; i < e; ++i) { const int32_t k = a[i]; a[i] = run; run += k; }
    ft_sync<LDS>();
    return total;
}

// Exclusive prefix sums over the workgroup's work-items of K values each (in tid order), results in registers, totals to every
// work-item: ONE barrier.  tmp: K * NT / 64 words of LDS that nobody touches until the NEXT barrier after the call.
template <int NT, int K, bool LDS>
__device__ __forceinline__ void ft_scan_tid(int32_t (&v)[K], int32_t *tmp, int32_t (&total)[K])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t incl[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { incl[k] = ft_wave_incl<FtAdd>(v[k]); if (lane == 63) tmp[k * (NT / 64) + wv] = incl[k]; }
    ft_sync<LDS>();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { const int32_t t = tmp[k * (NT / 64) + w]; tot += t; if (w < wv) base += t; }
        total[k] = tot;
        v[k] = base + incl[k] - v[k];
    }
}

// K exclusive prefix sums at once (same barriers as one)
template <int NT, int K, bool LDS>
__device__ __forceinline__ void ft_block_scan_k(int32_t *const (&a)[K], int n, int32_t *tmp, int32_t (&total)[K])
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (n <= 64) {
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int32_t v = lane < n ? a[k][lane] : 0;
                const int32_t incl = ft_wave_incl<FtAdd>(v);
                if (lane < n) a[k][lane] = incl - v;
                if (lane == 63) tmp[k] = incl;
            }
        }
        ft_sync<LDS>();
#pragma unroll
        for (int k = 0; k < K; ++k) total[k] = tmp[k];
        return;
    }
    const int per = (n + NT - 1) / NT;
    const int b = min(n, tid * per), e = min(n, b + per);
    int32_t sum[K], incl[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { sum[k] = 0; for (int i = b; i < e; ++i) sum[k] += a[k][i]; incl[k] = sum[k]; }
#pragma unroll
    for (int k = 0; k < K; ++k) incl[k] = ft_wave_incl<FtAdd>(incl[k]);
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < K; ++k) tmp[k * (NT / 64) + (tid >> 6)] = incl[k];
    }
    ft_sync<LDS>();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { const int32_t t = tmp[k * (NT / 64) + w]; tot += t; if (w < (tid >> 6)) base += t; }
        total[k] = tot;
        int32_t run = base + incl[k] - sum[k];
        for (int i = b; i < e; ++i) { const int32_t v = a[k][i]; a[k][i] = run; run += v; }
    }
    ft_sync<LDS>();
}
// the segment of item j in the exclusive prefix sums off[0..n] (off[0] = 0 <= j < off[n]): the largest i with off[i] <= j
// (empty segments are skipped)
__device__ __forceinline__ int ft_seg_find(const int32_t *off, int n, int j)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= j) lo = mid; else hi = mid; }
    return lo;
}
// "best score, earliest back-pointer among equals" as one 64-bit maximum: the searches over (item, back-pointer) pairs
// below reduce with atomicMax where the reference keeps `if (score > best) { best = score; bestbp = bp; }` in a loop
// over ascending back-pointers
__device__ __forceinline__ unsigned long long ft_key(int32_t score, int bp)
{
    return ((unsigned long long)((uint32_t)score ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - bp);
}
__device__ __forceinline__ unsigned long long ft_key_floor(int32_t floor)        // "nothing better than `floor` seen"
{
    return ((unsigned long long)((uint32_t)floor ^ 0x80000000u) << 32) | 0xffffffffull;
}
__device__ __forceinline__ bool ft_key_none(unsigned long long k) { return (uint32_t)k == 0xffffffffu; }
__device__ __forceinline__ int32_t ft_key_score(unsigned long long k) { return (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); }
__device__ __forceinline__ int ft_key_bp(unsigned long long k) { return 0x7fffffff - (int)(uint32_t)k; }

// per-phase cycle counts of work-item 0 (a profiling build only: -DPSGPU_FT_PROFILE; the product kernel has none of it)
#ifdef PSGPU_FT_PROFILE
#define FT_PROF(i) do { if (tid == 0) { const long long t_ = clock64(); s_prof[i] += t_ - s_last; s_last = t_; } } while (0)
// distribution of the evaluation's duration over the frames: [44] frames over 16k cycles, [45] their cycles, [46] the longest,
// [47] evaluations in the frames over 16k
#define FT_PROFD0() do { if (tid == 0) s_d0 = clock64(); } while (0)
#define FT_PROFD1(n) do { if (tid == 0) { const long long d_ = clock64() - s_d0; if (d_ > 16000) { s_prof[44] += 1; s_prof[45] += d_; s_prof[47] += (n); } if (d_ > s_prof[46]) s_prof[46] = d_; } } while (0)
// the other wavefronts' view of one phase: k = 0 marks its start, k = 1, 2 end intervals (slots 32 + 4 * wavefront + k)
#define FT_PROFW(k) do { if ((tid & 63) == 0 && tid > 0 && tid < 256) { const long long t_ = clock64(); if (k) s_prof[32 + 4 * (tid >> 6) + (k)] += t_ - s_lastw[tid >> 6]; s_lastw[tid >> 6] = t_; } } while (0)
#else
#define FT_PROF(i) do { } while (0)
#define FT_PROFW(k) do { } while (0)
#define FT_PROFD0() do { } while (0)
#define FT_PROFD1(n) do { } while (0)
#endif

// Registers.  The LDS layout's pool limits a compute unit to two workgroups, i.e. two waves per SIMD, and a compiler that
// sees that hands the kernel the whole register file -- 251 VGPRs (allocated: 256) a wave, 2 x 256 = all 512 of a SIMD:
// NO other wave can start on a compute unit that holds two of these workgroups, so with 512 utterances resident every kernel
// of every other stream waits for the first utterance to finish (measured: a 256-element fill launched beside the search
// completes 76 ms after the search started whenever it was launched, profiles/r03_overlap.txt).  The search is a
// latency-bound recurrence that leaves most issue slots idle; the throughput-bound stages of ANOTHER batch (front end,
// scorer) are the natural co-runners.  So the pool is dynamic LDS -- its size is not the compiler's business -- and the
// kernel asks for the occupancy of kFtWavesPerEu waves per SIMD, i.e. at most 168 VGPRs a wave: two resident workgroups then
// leave a third of every SIMD's registers (and six of its eight wave slots) to other kernels.
#ifndef PSGPU_FT_WAVES
#define PSGPU_FT_WAVES 3
#endif
// PSGPU_FT_ROWS_DEVICE = 1: a launch that reads score ROWS (LDS layout) reads them where the scorer left them, in device
// memory, instead of copying each frame's row into LDS one frame ahead: 10 KB less LDS per workgroup (en-us: 5126 x 2 bytes)
#ifndef PSGPU_FT_ROWS_DEVICE
#define PSGPU_FT_ROWS_DEVICE 0
#endif
constexpr bool kFtRowsDevice = PSGPU_FT_ROWS_DEVICE != 0;
// (s_setprio 3 for this kernel's waves -- ahead of the co-runners' at the SIMD's arbiter -- changed nothing: 112.4 vs 111.9 ms
//  per step; what a busy device costs the search is the latency of its device-memory accesses)
constexpr int kFtWavesPerEu = PSGPU_FT_WAVES;
#if defined(__HIPCC__)
extern __shared__ __attribute__((aligned(16))) int32_t ft_dyn_pool[];
#ifndef PSGPU_FT_WAVES_BIG
#define PSGPU_FT_WAVES_BIG 1
#endif
#define FT_KERNEL_ATTR(SMALL) __attribute__((amdgpu_waves_per_eu((SMALL) ? kFtWavesPerEu : (NT == kFtThreadsBig ? PSGPU_FT_WAVES_BIG : 1))))
#else
#define FT_KERNEL_ATTR(SMALL)
#endif

template <int NE, int NT, bool SMALL, bool LISTS>
__global__ __launch_bounds__(NT) FT_KERNEL_ATTR(SMALL)
void fwdtree_kernel(FtDev p, const int16_t *__restrict__ senscr_, int64_t scr_stride, const int32_t *__restrict__ penalties_,
                    const int32_t *__restrict__ utt_off_, int32_t raw_mode, int32_t pl_window, FtBufs bf)
{
    using F = ChF<NE>;
#if defined(__HIPCC__)
    int32_t *const s_pool = ft_dyn_pool;                 // SMALL: kFtLdsWords words of dynamic LDS (the launch says so)
#else
    __shared__ __attribute__((aligned(16))) int32_t s_pool[SMALL ? kFtLdsWords : kFtSlabPoolWords];     // (the workgroup simulator)
#endif
    __shared__ uint32_t s_bits[kFtMaxSen / 32];
    __shared__ int32_t s_nb;
    __shared__ int32_t s_red[8];
    __shared__ int32_t s_scan[6 * NT / 64];
    __shared__ int32_t s_bins[320];      // histogram of the maxhmmpf beam; word_transition: 64 keys (64-bit) + 3 x 64 decoded
    __shared__ int32_t s_sc[8];          // best_score, lpbest, dynamic_beam, bpidx, bss_head, n_cand, status, n_frame
    __shared__ unsigned long long s_evals;
    __shared__ int32_t s_nsen;           // listed senones, summed over the frames (raw-score mode)
    __shared__ int32_t s_nev;            // length of the frame's evaluation list
    // slab layouts: the pruning step works on chunks of kPrIC roots / listed nodes ("items") whose snapshot lies in LDS (the
    // dynamic pool, kPrArrays arrays of kPrIC words), so that the (item, child) pairs of a chunk find their item by a bisection
    // in LDS and read the parent's side of a decision -- and, for an item's own entry, the node's side -- from LDS
    constexpr int kPrIC = SMALL ? 1 : kFtIpt * NT;
    int32_t *const s_it_node = s_pool, *const s_it_out = s_pool + kPrIC, *const s_it_outh = s_pool + 2 * kPrIC,
            *const s_it_fp = s_pool + 3 * kPrIC, *const s_it_k0 = s_pool + 4 * kPrIC, *const s_it_par = s_pool + 5 * kPrIC,
            *const s_it_sc0 = s_pool + 6 * kPrIC, *const s_it_kid0 = s_pool + 7 * kPrIC,
            *const s_it_poff = s_pool + 8 * kPrIC;                                            // (poff: kPrIC + 1 entries)
    // ... and the index of the frame's active list (slab layouts; behind the item arrays, ft_slab_pool): a bitmap of the listed nodes,
    // the populations of the bitmap words before each word (so that a node's RANK among the listed nodes is two LDS reads and a
    // population count) and the table rank -> position in the list.  A pair whose other node is NOT listed knows that node's state
    // without asking (it was cleared when it left the list) -- two thirds of the pairs of the 134,865-word task; one whose other node
    // IS listed finds that node's compact record by its position.
    uint32_t *const s_lb = reinterpret_cast<uint32_t *>(s_pool + (SMALL ? 0 : p.lds_lb));
    uint16_t *const s_pre = reinterpret_cast<uint16_t *>(s_pool + (SMALL ? 0 : p.lds_pre));
    uint16_t *const s_perm = reinterpret_cast<uint16_t *>(s_pool + (SMALL ? 0 : p.lds_perm));
    __shared__ int32_t s_sup[kFtMaxBitWords / kFtLbBlock];       // listed nodes before each block of kFtLbBlock bitmap words (s_pre counts from the block's start)
    __shared__ int32_t s_nroot;          // slab layouts: roots evaluated in the frame
    // slab layouts: which roots are entered for this frame / already stamped for the next (two bitmaps taking turns): the frame's
    // passes over the roots ask these instead of every root's record (a 64-byte line each, several times a frame)
    __shared__ uint32_t s_rb[2][SMALL ? 1 : kFtMaxRootWords];
    auto rb_get = [&](int which, int i) { return (s_rb[which][i >> 5] >> (i & 31)) & 1u; };
    auto rb_set = [&](int which, int i) { atomicOr(&s_rb[which][i >> 5], 1u << (i & 31)); };
    __shared__ int32_t s_penb[SMALL ? 1 : kFtMaxCi];     // slab layouts: the frame's phone-loop penalties
    // slab layouts: the frame's score row and the transition matrices in LDS too (the pool) -- a channel's evaluation then asks device
    // memory for its record only (the eight transition bytes and three senone scores of a 3-state HMM were eleven requests of their own)
    int16_t *const s_rowb = reinterpret_cast<int16_t *>(s_pool + (SMALL ? 0 : p.lds_row));
    uint8_t *const s_tpb = reinterpret_cast<uint8_t *>(s_pool + (SMALL ? 0 : p.lds_tp));
    const int tid = threadIdx.x;
#ifdef PSGPU_FT_PROFILE
    __shared__ long long s_prof[48], s_last, s_lastw[4], s_d0;
    if (tid == 0) { for (int i = 0; i < 48; ++i) s_prof[i] = 0; s_last = clock64(); }
#endif
    const int N = p.N, R = p.R, n1 = p.n1, n_ci = p.n_ci;
#ifdef PSGPU_FT_FIXLAY                     /* (a measuring build: the LDS layout of ONE task as compile-time constants) */
    static constexpr FtLay kFixLay = PSGPU_FT_FIXLAY;
    const FtLay &L = SMALL ? kFixLay : p.lay;
#else
    const FtLay &L = p.lay;
#endif

    // ---- pointers.  Everything the host handed over is global memory (psgpu_as_global: see psgpu_internal.h).
    const int16_t *const senscr = psgpu_as_global(senscr_);
    const int32_t *const penalties = psgpu_as_global(penalties_);
    const int32_t *const utt_off = psgpu_as_global(utt_off_);
    int32_t *const gs = psgpu_as_global(bf.slab) + (size_t)blockIdx.x * p.per;
    int32_t *const fb = SMALL ? s_pool : gs + p.g_fast;
    constexpr int TREC = F::REC;
    // (LDS layout: records of F::WORDS = 3 NE + 6 words side by side -- an odd stride, so consecutive channels still hit different
    //  banks, and a field's offset is an immediate of the LDS instruction instead of field x channels held in a scalar register: this
    //  kernel's speed follows the number of scalar values it keeps alive)
    const ChView tv = { fb + L.rec, SMALL ? F::WORDS : TREC, 1 };                 // tree nodes [0, N), single-phone words [N, N + n1)
    const ChView wv = { gs + p.g_wrec, F::REC, 1 };                               // last-phone slots [0, TOT)
    // (LDS layout: columns of the two interleaved blocks, see FtCol; slab layouts: arrays of the utterance's slab)
    int32_t *const nodeA = fb + L.node_blk, *const wordA = fb + L.word_blk;
    using NCol = FtCol<SMALL, kFtNodeCols>;
    using WCol = FtCol<SMALL, kFtWordCols>;
    using NColC = FtCol<SMALL, kFtNodeCols, const int32_t>;
    using WColC = FtCol<SMALL, kFtWordCols, const int32_t>;
    const WCol word_active = { SMALL ? wordA + WC_ACTIVE : fb + L.word_active }, word_lat_idx = { SMALL ? wordA + WC_LATIDX : fb + L.word_lat_idx },
               lt_sf = { SMALL ? wordA + WC_LTSF : fb + L.lt_sf }, lt_dscr = { SMALL ? wordA + WC_LTDSCR : fb + L.lt_dscr },
               lt_bp = { SMALL ? wordA + WC_LTBP : fb + L.lt_bp }, cand_mark = { SMALL ? wordA + WC_CMARK : fb + L.cand_mark },
               cand_wid = { SMALL ? wordA + WC_CWID : fb + L.cand_wid }, cand_score = { SMALL ? wordA + WC_CSCORE : fb + L.cand_score },
               cand_bp = { SMALL ? wordA + WC_CBP : fb + L.cand_bp };
    const NCol o_out = { SMALL ? nodeA + NC_OOUT : fb + L.o_out }, o_outh = { SMALL ? nodeA + NC_OOUTH : fb + L.o_outh },
               pos = { SMALL ? nodeA + NC_POS : fb + L.pos }, flag = { SMALL ? nodeA + NC_FLAG : fb + L.flag },
               o_frame = { SMALL ? nodeA + NC_OFRAME : fb + L.o_frame };
    int32_t *const cnt = fb + L.cnt, *const cnt2 = fb + L.cnt2, *const cnt3 = fb + L.cnt3, *const woff = fb + L.woff;
    // the frame's evaluation list: 16-bit entries in the LDS layout (channel index < 2^15, or 0x8000 | active word << 6 | right
    // context: the host checked n_w <= 512, n_ci <= 64) so that a list of EVERY channel fits the pool -- it cannot overflow
    using EvT = typename std::conditional<SMALL, uint16_t, int32_t>::type;
    EvT *const evl = reinterpret_cast<EvT *>(fb + L.evl);
    auto evl_put = [&](int at, int code) {
        if (SMALL) evl[at] = (EvT)((code & kFtWordCh) ? (0x8000 | (((code >> 8) & 0x1ff) << 6) | (code & 63)) : code);
        else evl[at] = (EvT)code;
    };
    auto evl_get = [&](int e) -> int {
        const int v = (int)evl[e];
        if (SMALL) return (v & 0x8000) ? (kFtWordCh | (((v >> 6) & 0x1ff) << 8) | (v & 63)) : v;
        return v;
    };
    unsigned long long *const ckey = reinterpret_cast<unsigned long long *>(fb + L.ckey);
    // scoring from the scorer's top-N lists (LDS layout only; the host sees to that)
    constexpr bool lists = LISTS && SMALL;               // (a template parameter: the score row is LDS here and device memory otherwise,
                                                         //  and an access whose address space is a run-time matter becomes a flat_* access)
    // slab layouts: the same parameter says where the word level's scratch arrays live -- LDS (the pruning's item arrays, idle during the
    // word level) or the slab: a compile-time matter for the same reason.  A frame whose counts do not fit the LDS arrays ends the
    // utterance with status 6; psgpu_fwdtree_grow(m, 6) switches the handle to the slab's arrays for good
    constexpr bool WLDS = LISTS && !SMALL;
    uint32_t *const l_cw = reinterpret_cast<uint32_t *>(fb + L.l_cw), *const l_sc = reinterpret_cast<uint32_t *>(fb + L.l_sc);
    uint8_t *const l_la = reinterpret_cast<uint8_t *>(fb + L.l_la);
    uint16_t *const l_list = reinterpret_cast<uint16_t *>(fb + L.l_list);
    int32_t *const l_norm = fb + L.l_norm;
    const SenModel smod = { psgpu_as_global(bf.mixw), psgpu_as_global(bf.sen2cb), p.n_sen, bf.ls_density };
    const int32_t *const tsc = psgpu_as_global(bf.tsc);
    const uint32_t *const tcw = psgpu_as_global(bf.tcw);
    uint8_t *const present = reinterpret_cast<uint8_t *>(fb + L.present);
    // slab layouts: the right-context channels come from a pool of blocks of kFtRcBlk records (and `present` bytes) -- a word gets a block
    // when its first channel is allocated (ngram_search_alloc_all_rc, ngram_search.c:583-633) and gives it back when it leaves the active
    // word list with no channel left (ngram_search_free_all_rc, :635-652); channel r of word w is record wblk[w] * kFtRcBlk + r.  (The LDS
    // layout's vocabularies are small: every (word, right context) keeps its own slot, wc_off[w] + r.)
    int32_t *const wblk = fb + (SMALL ? 0 : L.wblk), *const rcfree = fb + (SMALL ? 0 : L.rcfree);
    __shared__ int32_t s_nfree;                          // free blocks (the first s_nfree entries of rcfree)
    FtTab tb;
    tb.bp = psgpu_as_global(bf.bpa) + (size_t)blockIdx.x * bf.bp_cap * kBpRow; tb.bss = psgpu_as_global(bf.bss) + (size_t)blockIdx.x * bf.bss_cap;
    tb.idx = psgpu_as_global(bf.idx) + (size_t)blockIdx.x * (bf.max_frames + 2);
    tb.bp_cap = bf.bp_cap; tb.bss_cap = bf.bss_cap;
    int32_t *const step = psgpu_as_global(bf.step) + (size_t)blockIdx.x * bf.max_frames * 4;
    int32_t *const result = psgpu_as_global(bf.res) + (size_t)blockIdx.x * 8;
    // static tables
    const int32_t *const node_ci2 = psgpu_as_global(p.node_ci2), *const node_ssid = psgpu_as_global(p.node_ssid),
                  *const node_tmat = psgpu_as_global(p.node_tmat), *const homophone = psgpu_as_global(p.homophone),
                  *const w1_wid = psgpu_as_global(p.w1_wid), *const w1_ci = psgpu_as_global(p.w1_ci), *const w1_ci2 = psgpu_as_global(p.w1_ci2),
                  *const w1_ssid = psgpu_as_global(p.w1_ssid), *const w1_tmat = psgpu_as_global(p.w1_tmat), *const w1_mpx = psgpu_as_global(p.w1_mpx),
                  *const w1_of_word = psgpu_as_global(p.w1_of_word), *const d_first = psgpu_as_global(p.d_first),
                  *const d_last = psgpu_as_global(p.d_last), *const d_last2 = psgpu_as_global(p.d_last2), *const d_base = psgpu_as_global(p.d_base),
                  *const d_filler = psgpu_as_global(p.d_filler), *const rs_n = psgpu_as_global(p.rs_n), *const rs_ssid = psgpu_as_global(p.rs_ssid),
                  *const rs_cimap = psgpu_as_global(p.rs_cimap), *const ldiph = psgpu_as_global(p.ldiph), *const ci_tmat = psgpu_as_global(p.ci_tmat),
                  *const lmtab = psgpu_as_global(p.lm);
    const uint8_t *const tpall = SMALL ? reinterpret_cast<const uint8_t *>(fb + L.tp) : s_tpb;
    if (!SMALL) {
        const uint8_t *const g_tp = psgpu_as_global(p.tp);
        for (int i = tid; i < p.n_tmat * NE * (NE + 1); i += NT) s_tpb[i] = g_tp[i];
    }
    const uint16_t *const sseq = psgpu_as_global(p.sseq);
    const int32_t *const kids_ci = psgpu_as_global(p.kids_ci);
    const FtQuad *const node_q1 = reinterpret_cast<const FtQuad *>(psgpu_as_global(p.node_q1)),
                 *const node_q2 = reinterpret_cast<const FtQuad *>(psgpu_as_global(p.node_q2));
    const int32_t *const slot_sen = psgpu_as_global(p.slot_sen);
    FtQuad *const itb = reinterpret_cast<FtQuad *>(fb + L.itb);        // slab layouts only: [R]
    // ---- compact channels (slab layouts).  A tree node's channel exists while the node is listed: its record lives at the node's
    //      POSITION in the frame's active list, in arrays of quads (one array per quad of the record: sixty-four work-items on
    //      consecutive positions read 1 KB in a row), two buffers taking turns.  The pruning moves no channel: it writes, per place
    //      of the next list, the node and where its channel comes from (cxfer: its position in this frame's list, or "new", and the
    //      entering score and history if it was entered); the next frame makes the channels at their new places as it goes -- its
    //      first pass fetches the static side (from the old place or, for a new channel, from the static tables: parent, children,
    //      senones: the evaluation and the pruning of a listed node ask the static tables nothing), its evaluation fetches the scores
    //      and histories from the old place (old places rise with the new ones: the requests of neighbouring work-items fall into
    //      the same lines), evaluates and writes.  Per buffer: ND quads {score[0..NE), history[0..NE), out score, out history}, then
    //      {first entry, parent | ci << 24, first entry's index, children | penultimate-phone words << 16} (the pruning's side) and
    //      {senones and transition matrix, 16 bits each, ...} (the evaluation's; the node itself is the list's entry).  What a frame costs in device memory is then mostly streams; the random accesses
    //      left are a decision's look at its OTHER node (by rank -> position, see s_lb) and the static rows of the nodes that ENTER
    //      the list.
    constexpr int ND = NE == 3 ? 2 : 3;
    const int ccap = SMALL ? 1 : L.ccap;
    FtQuad *const cq = reinterpret_cast<FtQuad *>(fb + (SMALL ? 0 : L.cq));
    FtQuad *const csum = reinterpret_cast<FtQuad *>(fb + (SMALL ? 0 : L.csum));
    int32_t *const cxf = fb + (SMALL ? 0 : L.cxfer);
    FtPair *const cxp = reinterpret_cast<FtPair *>(fb + (SMALL ? 0 : L.cxpl));
    int32_t *const g_perm = fb + (SMALL ? 0 : L.cperm);
    auto cbuf = [&](int b, int k) { return cq + (size_t)(b * (ND + 2) + k) * ccap; };      // array k of buffer b
    // (the evaluation's static side -- senones and transition matrix -- is 8 bytes a channel for 3-state models: its array holds pairs)
    using FtSen = typename std::conditional<NE == 3, FtPair, FtQuad>::type;
    auto csen = [&](int b) { return reinterpret_cast<FtSen *>(cq + (size_t)(b * (ND + 2) + ND + 1) * ccap); };
    auto sen_of = [&](const FtQuad &q) { FtSen r; r.x = q.x; r.y = q.y; if constexpr (NE != 3) { r.z = q.z; r.w = q.w; } return r; };
    auto sen_z = [&](const FtSen &q) { if constexpr (NE != 3) return q.z; else return 0; };
    const FtQuad *const node_st1 = reinterpret_cast<const FtQuad *>(psgpu_as_global(p.node_st1));
    bool perm_lds = true;                                              // the current list's rank -> position table is the LDS one (uniform)
    const FtDict dict = { psgpu_as_global(p.d_pronlen), d_last, d_last2, d_base, d_filler, rs_n, n_ci };
    // the tree's structure: LDS copies in the small layout
    const int32_t *const kids = SMALL ? fb + L.kids : psgpu_as_global(p.kids);
    const NColC kid_off = { SMALL ? nodeA + NC_KIDOFF : psgpu_as_global(p.kid_off) }, parent = { SMALL ? nodeA + NC_PARENT : psgpu_as_global(p.parent) },
                node_ci = { SMALL ? nodeA + NC_CI : psgpu_as_global(p.node_ci) }, node_pw = { SMALL ? nodeA + NC_PW : psgpu_as_global(p.node_pw) };
    const WColC wc_off = { SMALL ? wordA + WC_WCOFF : psgpu_as_global(p.wc_off) };
    auto pslot = [&](int w, int r) { return SMALL ? wc_off[w] + r : wblk[w] * kFtRcBlk + r; };
    // the words' first phones / base ids and the single-phone words' ids: LDS copies in the small layout (the pair searches'
    // first trip to device memory is then the back-pointer entries' alone)
    const WColC dfirst_f = { SMALL ? wordA + WC_DFIRST : d_first }, dbase_f = { SMALL ? wordA + WC_DBASE : d_base },
                dlast_f = { SMALL ? wordA + WC_DLAST : d_last }, homo_f = { SMALL ? wordA + WC_HOMO : homophone },
                dfill_f = { SMALL ? wordA + WC_DFILL : d_filler };
    const int32_t *const w1w_f = SMALL ? fb + L.w1w : w1_wid, *const w1ci_f = SMALL ? fb + L.w1ci : w1_ci, *const w1ci2_f = SMALL ? fb + L.w1ci2 : w1_ci2;
    // psgpu_fwdtree_search_resume: the utterance's saved state; f0 = the frames searched by the calls before
    // (slab layouts: everything the frames share is in the utterance's slab -- which stays where it is between the calls -- but the
    //  counters and the carried registers)
    int32_t *const live = bf.live ? psgpu_as_global(bf.live) + (size_t)blockIdx.x * (kFtLiveHdr + (SMALL ? L.rows_total : 0)) : nullptr;
    // (word 16: the block holds a saved search -- zeroed by the caller for an utterance that starts afresh among resumed ones)
    const bool resumed = live != nullptr && (bf.live_mode & 2) != 0 && live[16] == kFtLiveMagic;
    const int f0 = resumed ? live[0] : 0;
    if (SMALL && !resumed) {                             // (a resumed search loads the pool, these copies with it)
        for (int i = tid; i < p.n_w; i += NT) {
            int32_t *const r = wordA + i * kFtWordCols;
            r[WC_DFIRST] = d_first[i]; r[WC_DBASE] = d_base[i]; r[WC_DLAST] = d_last[i]; r[WC_HOMO] = homophone[i]; r[WC_DFILL] = d_filler[i];
        }
        for (int i = tid; i < n1; i += NT) { fb[L.w1w + i] = w1_wid[i]; fb[L.w1ci + i] = w1_ci[i]; fb[L.w1ci2 + i] = w1_ci2[i]; }
    }
    if (SMALL && !resumed) {
        const int32_t *const g_ko = psgpu_as_global(p.kid_off), *const g_k = psgpu_as_global(p.kids), *const g_p = psgpu_as_global(p.parent),
                      *const g_c = psgpu_as_global(p.node_ci), *const g_w = psgpu_as_global(p.node_pw);
        for (int i = tid; i <= N; i += NT) nodeA[i * kFtNodeCols + NC_KIDOFF] = g_ko[i];
        for (int i = tid; i < p.M; i += NT) fb[L.kids + i] = g_k[i];
        for (int i = tid; i < N; i += NT) { int32_t *const r = nodeA + i * kFtNodeCols; r[NC_PARENT] = g_p[i]; r[NC_CI] = g_c[i]; r[NC_PW] = g_w[i]; }
        const int32_t *const g_o = psgpu_as_global(p.wc_off);
        for (int i = tid; i <= p.n_w; i += NT) wordA[i * kFtWordCols + WC_WCOFF] = g_o[i];
        const uint8_t *const g_tp = psgpu_as_global(p.tp);
        uint8_t *const l_tp = reinterpret_cast<uint8_t *>(fb + L.tp);
        for (int i = tid; i < p.n_tmat * NE * (NE + 1); i += NT) l_tp[i] = g_tp[i];
    }
    // small layout: the frame's exits in LDS (in the evaluation list's words, idle after prune_word_chan) and every exit's score by
    // right context phone in device memory -- see the exits step
    constexpr int32_t kXSingle = 0x40000000;                             // xfr word 0: a single-phone word's entry (one score, word 3)
    int32_t *const xfr = fb + L.xfr;
    const int xst = n_ci + 3, xcap = SMALL ? L.xfr_cap : 0;
    int32_t *const bssx = (SMALL && bf.bssx) ? psgpu_as_global(bf.bssx) + (size_t)blockIdx.x * bf.bp_cap * n_ci : nullptr;
    __shared__ int32_t s_xbad;                                           // an entry of the frame is not in xfr as the word transitions expect it
    int16_t *const s_row = reinterpret_cast<int16_t *>(fb + L.row);       // small layout only
    int32_t *const s_pen = fb + L.pen;                                      // small layout only: [2][n_ci]

    // an utterance shorter than the look-ahead window is never searched by the reference: ps_end_utt steps the main search
    // over the last pl_window frames only `if (output_frame >= pl_window)` (pocketsphinx.c:1329-1333)
    // (bf.lag > 0: an utterance in progress -- the phone loop has seen T_in frames, the search steps through the first
    //  T_in - lag of them, as ps_search_forward leaves the two between calls, pocketsphinx.c:1173-1197)
    const int32_t *const ext = bf.ext ? psgpu_as_global(bf.ext) + 3 * (size_t)blockIdx.x : nullptr;
    const int t0 = utt_off[blockIdx.x], T_in = ext ? ext[0] : utt_off[blockIdx.x + 1] - t0,
              T = ext ? min(ext[1], T_in) : (bf.lag > 0 ? max(T_in - bf.lag, 0) : ((raw_mode && T_in < pl_window) ? 0 : T_in));
    const int W1 = SMALL ? N : R;                        // single-phone word i is channel W1 + i of tv (slab layouts: tv holds the roots and these)
    int n_acl_cur = 0, n_awl_cur = 0;                    // list lengths: uniform copies (every thread tracks them identically)
    uint32_t evals_run = 0u;                             // HMM evaluations so far (ngs->st.n_hmm_eval; saturating: compared with maxhmmpf), likewise
    int nwc_cur = 0;                                     // right-context channels of the active words (the last of woff's prefix sums), likewise

    // ---- hmm_init of every permanent channel, ngram_fwdtree_start (:469-520)
    if (!resumed) {
        for (int c = tid; c < W1; c += NT) ch_init<NE>(tv, c, c < R, node_ssid[c], node_tmat[c], sseq);
        for (int i = tid; i < n1; i += NT) ch_init<NE>(tv, W1 + i, w1_mpx[i], w1_ssid[i], w1_tmat[i], sseq);
        if (SMALL) { for (int i = tid; i < p.TOT; i += NT) present[i] = 0; }
        else {
            for (int i = tid; i < p.rc_blocks * (kFtRcBlk / 4); i += NT) reinterpret_cast<int32_t *>(present)[i] = 0;
            for (int i = tid; i < p.rc_blocks; i += NT) rcfree[i] = i;
            for (int w = tid; w < p.n_w; w += NT) wblk[w] = -1;
        }
        for (int w = tid; w < p.n_w; w += NT) { word_lat_idx[w] = -1; lt_sf[w] = -1; word_active[w] = 0; cand_mark[w] = -1; }
        if (SMALL) { for (int c = tid; c < N; c += NT) pos[c] = -1; }      // pos is kept at -1 between frames
    }
    if (tid == 0) {
        s_sc[0] = 0; s_sc[1] = 0; s_sc[2] = p.beam; s_sc[3] = 0; s_sc[4] = 0; s_sc[5] = 0; s_sc[6] = 0; s_sc[7] = 0;
        s_evals = 0ull; s_nb = 0x7fffffff; s_nsen = 0; s_nev = 0; s_nroot = 0; s_nfree = SMALL ? 0 : p.rc_blocks;
    }
    {
        const int nwords = (p.n_sen + 31) >> 5;
        for (int i = tid; i < nwords; i += NT) s_bits[i] = 0u;
        if (!SMALL) {
            for (int i = tid; i < p.lb_words; i += NT) s_lb[i] = 0u;
            for (int i = tid; i < 2 * kFtMaxRootWords; i += NT) s_rb[i / kFtMaxRootWords][i % kFtMaxRootWords] = 0u;
        }
    }
    // slab layouts: the index of an active list -- bitmap of its nodes, the populations of the bitmap words before each word, rank ->
    // position -- from the list itself (node ids in list order).  Called by every work-item with the bitmap all zero; ends without a
    // barrier (the table's first readers are a frame's pruning pairs, behind many)
    auto build_index = [&](const int32_t *list, int n) {
        for (int o = tid; o < n; o += NT) { const int c = list[o]; atomicOr(&s_lb[c >> 5], 1u << (c & 31)); }
        __syncthreads();
        // K consecutive words a work-item, K a power of two: a work-item's words lie in one block of kFtLbBlock
        const int nwl = p.lb_words;
        int K = 1;
        while (K * NT < nwl) K *= 2;
        const int b = min(nwl, tid * K), e = min(nwl, b + K);
        int32_t v[1] = { 0 }, tot[1];
        for (int w = b; w < e; ++w) v[0] += __popc(s_lb[w]);
        ft_scan_tid<NT, 1, true>(v, s_scan, tot);
        if (b < nwl && (b & (kFtLbBlock - 1)) == 0) s_sup[b / kFtLbBlock] = v[0];
        ft_sync<true>();
        if (b < nwl) {
            int32_t run = v[0] - s_sup[b / kFtLbBlock];
            for (int w = b; w < e; ++w) { s_pre[w] = (uint16_t)run; run += __popc(s_lb[w]); }
        }
        __syncthreads();
        perm_lds = n <= p.lds_perm_cap;                  // (a frame that lists more than the LDS table holds keeps the table in the slab)
        for (int o = tid; o < n; o += NT) {
            const int c = list[o];
            const int r = s_sup[c >> 15] + (int)s_pre[c >> 5] + __popc(s_lb[c >> 5] & ((1u << (c & 31)) - 1u));
            if (perm_lds) s_perm[r] = (uint16_t)o; else g_perm[r] = o;
        }
#ifdef PSGPU_FT_CHECK_LISTS
        __syncthreads();
        for (int o = tid; o < n; o += NT) {
            const int c = list[o];
            const int r = c < 0 ? -1 : s_sup[c >> 15] + (int)s_pre[c >> 5] + __popc(s_lb[c >> 5] & ((1u << (c & 31)) - 1u));
            const int ps = r < 0 ? -1 : (perm_lds ? (int)s_perm[r] : g_perm[r]);
            if (ps != o) { printf("index of %d listed nodes: position %d holds node %d, whose rank %d leads to position %d\n", n, o, c, r, ps); abort(); }
        }
        __syncthreads();
#endif
    };
    static_assert(kFtLbBlock == 1024, "a bitmap word's block is node >> 15");
    if (resumed) {
        // the pool as the previous call's last frame left it (the static tables' copies with it), the counters, the frame loop's
        // carried registers; the senone bitmap, the listed-nodes bitmap and the normaliser are reset at every frame's end and are as
        // the lines above left them
        __syncthreads();
        if (SMALL) {
            // (sixteen bytes a work-item, four requests in flight: a load a loop step waited for was 0.1 ms of a resumed launch)
            const FtQuad *const src = reinterpret_cast<const FtQuad *>(live + kFtLiveHdr);
            FtQuad *const dst = reinterpret_cast<FtQuad *>(s_pool);
            const int nq = L.rows_total >> 2;            // (the pool's arrays are padded to four words: ft_layout)
            for (int i = tid; i < nq; i += 4 * NT) {
                const int i1 = min(i + NT, nq - 1), i2 = min(i + 2 * NT, nq - 1), i3 = min(i + 3 * NT, nq - 1);
                const FtQuad a = src[i], b = src[i1], c = src[i2], e = src[i3];
                dst[i] = a; dst[i1] = b; dst[i2] = c; dst[i3] = e;
            }
        }
        if (tid < 8) s_sc[tid] = live[8 + tid];
        if (tid == 0) { s_evals = (unsigned long long)(uint32_t)live[5] | ((unsigned long long)(uint32_t)live[6] << 32); s_nsen = live[7]; s_nfree = live[17]; }
        n_acl_cur = live[1]; n_awl_cur = live[2]; evals_run = (uint32_t)live[3]; nwc_cur = live[4];
        __syncthreads();
        if (!SMALL) {                                    // (the compact channels are in the slab, where the last frame left them; their index is LDS)
            build_index(fb + ((f0 & 1) ? L.acl1 : L.acl0), n_acl_cur);
            for (int i = tid; i < R; i += NT) if (tv.at(i, F::FRAME) == f0) rb_set(f0 & 1, i);
            __syncthreads();
        }
    }
    // Scores: the frame's row is read from LDS (s_row).  From rows: the next frame's row travels from HBM into s_row while this
    // frame's word level runs -- issued after the evaluation, the row's last reader -- by LDS-DMA (global_load_lds_dword: no
    // registers in between; sixteen registers a work-item held across two thirds of the frame were most of what the kernel
    // spilled under its 168-register budget).  The loads are counted by vmcnt: waited for before the frame's last barrier.
    // From lists: the frame's scores are computed into s_row.  Penalties: two LDS rows taking turns.
    const int row_dw = (p.n_sen + 1) >> 1;              // dwords per score row (the host checked the alignment)
    auto row_fetch = [&](int fr) {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(senscr + (size_t)(t0 + fr) * scr_stride);
        uint32_t *d = reinterpret_cast<uint32_t *>(s_row);
#if defined(__HIP_DEVICE_COMPILE__)
        // sixteen bytes a lane (1 KB a wave-instruction: an LDS-DMA instruction costs the wave ~100-200 cycles of issue whatever it
        // moves), the row's last dwords singly.  q0: the wavefront's first quad; the LDS address is base + lane * 16 (aux 2:
        // non-temporal, a row is read once)
        const int row_q = row_dw >> 2;
        for (int q0 = tid & ~63; q0 < row_q; q0 += NT) {
            if (q0 + (tid & 63) < row_q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 4 * (q0 + (tid & 63))),
                                                 (__attribute__((address_space(3))) void *)(d + 4 * q0), 16, 0, 2);
        }
        if (tid < 64 && 4 * row_q + tid < row_dw)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 4 * row_q + tid),
                                             (__attribute__((address_space(3))) void *)(d + 4 * row_q), 4, 0, 2);
#else
        for (int i = tid; i < row_dw; i += NT) d[i] = g[i];
#endif
    };
    auto pen_frame = [&](int f) { return t0 + (raw_mode ? min(f + pl_window, T_in - 1) : f); };
    // scoring from lists: a frame's lists -- per (codebook, stream) chain four raw scores and four codewords -- travel one frame
    // ahead in registers like the score row does, 5 words a work-item instead of 16: loaded after the evaluation, the streams'
    // normalisers (ptm_mgau_codebook_norm, ptm_mgau.c:265-295: the maximum over the codebooks of best score >> 10) reduced per
    // wavefront at the frame's end, packed into LDS at the top of the frame they belong to
    int32_t ps0 = 0, ps1 = 0, ps2 = 0, ps3 = 0;
    uint32_t pcw = 0;
    auto lists_load = [&](int fr) {
        if (tid < bf.ls_chains) {
            const FtQuad q = *reinterpret_cast<const FtQuad *>(tsc + ((size_t)tid * bf.ls_total + t0 + fr) * 4);
            ps0 = q.x; ps1 = q.y; ps2 = q.z; ps3 = q.w;
            pcw = tcw[(size_t)tid * bf.ls_total + t0 + fr];
        }
    };
    auto lists_norm = [&]() {                            // (every work-item: wavefront scans)
        const int fs = tid % kSenStreams;
        const int32_t v = tid < bf.ls_chains ? (ps0 >> kSenShift) : FtMax::id;
#pragma unroll
        for (int q = 0; q < kSenStreams; ++q) {
            const int32_t m = ft_wave_incl<FtMax>(fs == q ? v : FtMax::id);
            if ((tid & 63) == 63) l_norm[(tid >> 6) * kSenStreams + q] = m;
        }
    };
    auto lists_pack = [&]() {
        if (tid < bf.ls_chains) {
            int32_t nm = FtMax::id;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) nm = max(nm, l_norm[w * kSenStreams + tid % kSenStreams]);
            l_sc[tid] = sen_pack_scores(ps0, ps1, ps2, ps3, nm);
            l_cw[tid] = pcw;
        }
    };
    // (a resumed search that has failed -- its best score at the floor -- stops at its first frame below and reads nothing)
    if (SMALL && f0 < T && !(resumed && live[8] <= kW)) {   // the first frame's scores and penalties (f0 & 1: the penalty rows take turns)
        if (lists) {
            const uint8_t *const la = psgpu_as_global(bf.la);
            for (int i = tid; i < 512; i += NT) l_la[i] = i < bf.ls_la_size ? la[i] : 0;
            lists_load(f0);
            lists_norm();
        }
        else if (!kFtRowsDevice) {
            const uint32_t *g = reinterpret_cast<const uint32_t *>(senscr + (size_t)(t0 + f0) * scr_stride);
            uint32_t *d = reinterpret_cast<uint32_t *>(s_row);
            for (int i = tid; i < row_dw; i += NT) d[i] = g[i];
        }
        if (p.has_pl) for (int i = tid; i < n_ci; i += NT) s_pen[(f0 & 1) * n_ci + i] = penalties[(size_t)pen_frame(f0) * n_ci + i];
    }
    __syncthreads();
    // a session's second and later utterances: the multiplexed permanent channels (roots, single-phone words) start with the
    // per-state ssids the previous utterance left -- hmm_clear (hmm.c:181-196) resets scores and histories only, and a
    // state's ssid decides which senone the search lists for it
    if (bf.mpx_in && !resumed && (!ext || ext[2])) {
        const int32_t *const mi = psgpu_as_global(bf.mpx_in) + (size_t)blockIdx.x * (R + n1) * NE;
        for (int i = tid; i < (R + n1) * NE; i += NT) {
            const int q = i / NE, c = q < R ? q : W1 + (q - R);
            if (q < R || w1_mpx[q - R]) tv.at(c, F::SENID + i % NE) = mi[i];
        }
    }
    if (tid == 0 && !resumed) ch_enter<NE>(tv, W1 + w1_of_word[p.startwid], 0, -1, 0);
    __syncthreads();

    for (int f = f0; f < T; ++f) {
        const int cur = f & 1, nxt = cur ^ 1, nf = f + 1;
        const NCol aclc = { SMALL ? nodeA + (cur ? NC_ACL1 : NC_ACL0) : fb + (cur ? L.acl1 : L.acl0) },
                   acln = { SMALL ? nodeA + (cur ? NC_ACL0 : NC_ACL1) : fb + (cur ? L.acl0 : L.acl1) };
        const WCol awlc = { SMALL ? wordA + (cur ? WC_AWL1 : WC_AWL0) : fb + (cur ? L.awl1 : L.awl0) },
                   awln = { SMALL ? wordA + (cur ? WC_AWL0 : WC_AWL1) : fb + (cur ? L.awl0 : L.awl1) };
        // raw mode: the phone loop runs pl_window frames ahead and stops at the last frame
        // (slab layouts: the frame's penalty row is copied to LDS here -- its last readers, the previous frame's word
        //  transitions, are behind a barrier; its first reader, the pruning, is behind the barriers below)
        if (!SMALL && p.has_pl && tid < n_ci && s_sc[0] > kW) s_penb[tid] = penalties[(size_t)pen_frame(f) * n_ci + tid];
        const int32_t *const pp = SMALL ? s_pen + cur * n_ci : s_penb;
        constexpr bool ROW_LDS = SMALL && (LISTS || !kFtRowsDevice);          // the frame's row is in LDS (slab layouts: s_rowb, copied below)
        constexpr bool kSlabRowLds = true;                   // (the slab layouts' row in LDS: +3 % on the 134,865-word task)
        const int16_t *const row = ROW_LDS ? s_row : ((SMALL || !kSlabRowLds) ? senscr + (size_t)(t0 + f) * scr_stride : s_rowb);
        if (!SMALL && kSlabRowLds && s_sc[0] > kW) {          // (a search that has failed stops below and reads no row)
            // (its last readers, the previous frame's evaluation, are behind barriers; its first readers -- the senone marks
            //  below read the row for the normaliser -- are behind the barrier that follows)
            const int16_t *const g = senscr + (size_t)(t0 + f) * scr_stride;
            if (((scr_stride & 1) == 0) && (((uintptr_t)senscr & 3) == 0)) {
                const uint32_t *g32 = reinterpret_cast<const uint32_t *>(g);
                uint32_t *d32 = reinterpret_cast<uint32_t *>(s_rowb);
                for (int i = tid; i < (p.n_sen + 1) >> 1; i += NT) d32[i] = g32[i];
            }
            else for (int i = tid; i < p.n_sen; i += NT) s_rowb[i] = g[i];
            ft_sync<true>();
        }
        auto ft_pen = [&](int ci) { return p.has_pl ? pp[ci] : 0; };
        if (lists) lists_pack();                             // this frame's lists (read after the next barrier)
        // ---- ngram_search_mark_bptable, failure test, renormalisation (:1467-1480)
        if (tid == 0) { tb.idx[f] = s_sc[3]; s_xbad = 0; }     // (s_nev was zeroed before the previous frame's last barrier)
        const int32_t best_in = s_sc[0];
        if (best_in == kW || best_in < kW) break;
        const int32_t bp0 = s_sc[3];                          // this frame's first back-pointer
        if (tid < 8) s_red[tid] = kW;
        // a word near its end has its whole right-context fan-out (20-40 channels) allocated at once: the word level is
        // worked on one work-item per channel.  The channels of the active words are the segments [woff[i], woff[i + 1])
        // of one index range; an item finds its word by bisection (ft_seg_find).
        const int naw = n_awl_cur, na = n_acl_cur;
        // (woff -- the prefix sums of the active words' right-context counts -- and their total were made in the previous frame,
        //  in the single-phone words' scan, as soon as this frame's word list was complete: no barrier, no scan here)
        for (int i = tid; i < naw; i += NT) word_active[awlc[i]] = 0;
        // (slab layouts: a block of the pool per active word, kFtRcBlk places each -- sixty-four consecutive work-items look at one word's
        //  block, no search for the word a channel belongs to)
        const int nwc = SMALL ? nwc_cur : naw * kFtRcBlk;
        // ---- the frame's evaluation list: every HMM instance evaluate_channels (:605-715) visits -- roots entered for this
        //      frame, the listed tree nodes, the allocated right-context channels of the active words, the single-phone
        //      words entered for this frame -- compacted into one list (order irrelevant: independent evaluations, maxima
        //      and counts), so that marking the senones, the evaluation and the renormalisation each are ONE pass of the
        //      workgroup instead of four loops each paying its own latency.  Tree channels: their channel index;
        //      right-context channels: kFtWordCh | index of the word in the active list << 8 | right context.
        {
            const int n_items = R + na + n1 + nwc, lane = tid & 63;
            // raw-score mode: the senones of the listed channels are marked (compute_sen_active, :526-564) and the minimum of
            // their raw scores taken in the same pass
            int32_t mn = 0x7fffffff;
            auto mark_sen = [&](int sen) { atomicOr(&s_bits[sen >> 5], 1u << (sen & 31)); if (!lists) mn = min(mn, (int32_t)row[sen]); };
            auto mark = [&](const ChView &v, int c) {
                const int mpx = v.at(c, F::MPX);
                int sen[NE];
                bool ok[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) { sen[k] = v.at(c, F::SENID + k); ok[k] = !(mpx && sen[k] == kBadSsid); }
                if (mpx) {                                   // (the states' senone ids in ONE trip to the sseq table: a state without an ssid reads entry 0)
#pragma unroll
                    for (int k = 0; k < NE; ++k) sen[k] = sseq[(size_t)(ok[k] ? sen[k] : 0) * NE + k];
                }
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (ok[k]) {
                        atomicOr(&s_bits[sen[k] >> 5], 1u << (sen[k] & 31));
                        if (!lists) mn = min(mn, (int32_t)row[sen[k]]);
                    }
            };
            // slab layouts: the tree's items (roots, listed nodes) are not copied into the evaluation list -- the evaluation
            // walks them in item order -- and a listed node's senones come from the static table (a root's depend on its
            // left context: its record)
            const int Rq = SMALL ? R : 0, naq = SMALL ? na : 0;
            if (!SMALL) {
                int n_act_root = 0;
                for (int i0 = 0; i0 < R; i0 += NT) {
                    const int i = i0 + tid;
                    const bool act_root = i < R && rb_get(cur, i) != 0u;
                    if (act_root && raw_mode) mark(tv, i);
                    n_act_root += __popcll(__ballot(act_root));
                }
                if (raw_mode) {
                    // the listed nodes' senones: at the old place's static side, or (a new channel) in the static tables -- kFtTopUnroll
                    // positions a work-item at a time: the transfer words first, then the gathers they name, then the marks
                    const int32_t *const xf = cxf + (size_t)cur * ccap;
                    for (int j0 = tid; j0 < na; j0 += kFtTopUnroll * NT) {
                        int32_t xs[kFtTopUnroll], nd_[kFtTopUnroll]; FtSen a[kFtTopUnroll];
#pragma unroll
                        for (int u = 0; u < kFtTopUnroll; ++u) { const int j = j0 + u * NT; xs[u] = j < na ? xf[j] : 1; nd_[u] = 0; }
#pragma unroll
                        for (int u = 0; u < kFtTopUnroll; ++u) { const int j = j0 + u * NT; if (j < na && (xs[u] & 0x0fffffff) == 0) nd_[u] = aclc[j]; }
#pragma unroll
                        for (int u = 0; u < kFtTopUnroll; ++u) {
                            const int src = (xs[u] & 0x0fffffff) - 1;
                            a[u] = sen_of(FtQuad{ 0, 0, 0, 0 });
                            if (j0 + u * NT < na) { if (src >= 0) a[u] = csen(nxt)[src]; else a[u] = sen_of(node_st1[nd_[u]]); }
                        }
#pragma unroll
                        for (int u = 0; u < kFtTopUnroll; ++u) {
                            if (j0 + u * NT >= na) continue;
                            mark_sen(a[u].x & 0xffff); mark_sen((int)((uint32_t)a[u].x >> 16)); mark_sen(a[u].y & 0xffff);
                            if (NE == 5) { mark_sen((int)((uint32_t)a[u].y >> 16)); mark_sen(sen_z(a[u]) & 0xffff); }
                        }
                    }
                }
                if (lane == 0 && n_act_root) atomicAdd(&s_nroot, n_act_root);
            }
            for (int k0 = 0; k0 < n_items - (R - Rq) - (na - naq); k0 += NT) {
                int k = k0 + tid, code = -1;
                if (k < Rq) { if (tv.at(k, F::FRAME) == f) code = k; }
                else if ((k -= Rq) < naq) code = aclc[k];
                else if ((k -= naq) < n1) { if (tv.at(W1 + k, F::FRAME) == f) code = W1 + k; }
                else if ((k -= n1) < nwc) {
                    int i, r, slot; bool there;
                    if (SMALL) { i = ft_seg_find(woff, naw, k); r = k - woff[i]; slot = pslot(awlc[i], r); there = present[slot] != 0; }
                    else {
                        i = k / kFtRcBlk; r = k % kFtRcBlk;
                        const int w = awlc[i];
                        slot = pslot(w, r);
                        there = r < wc_off[w + 1] - wc_off[w] && present[slot] != 0;
                    }
                    if (there) {
                        code = kFtWordCh | (i << 8) | r;
                        if (raw_mode) {                          // (never multiplexed: the senone ids are the record's last quad(s))
                            const FtQuad *q = reinterpret_cast<const FtQuad *>(wv.b + (size_t)slot * F::REC + F::SENID);
                            const FtQuad a = q[0];
                            mark_sen(a.x); mark_sen(a.y); mark_sen(a.z);
                            if (NE == 5) { mark_sen(a.w); mark_sen(q[1].x); }
                        }
                    }
                }
                if (raw_mode && code >= 0 && !(code & kFtWordCh)) mark(tv, code);
                const unsigned long long m = __ballot(code >= 0);
                int base = 0;
                if (lane == 0 && m) base = atomicAdd(&s_nev, __popcll(m));
                base = ft_lane(base, 0);
                const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                if (code >= 0 && at < L.evl_cap) evl_put(at, code);
            }
            if (raw_mode) {
                mn = ft_wave_incl<FtMin>(mn);                 // (lane 63: the wavefront's minimum)
                if (lane == 63 && mn != 0x7fffffff) atomicMin(&s_nb, mn);
            }
        }
        ft_sync<SMALL>();
        const int n_evl = s_nev;                             // entries of the evaluation list (slab layouts: the word level's)
        const int n_ev = n_evl + (SMALL ? 0 : s_nroot + na); // HMM instances evaluated this frame
        if (n_evl > L.evl_cap) { if (tid == 0) s_sc[6] = 2; ft_sync<SMALL>(); break; }      // status 2: the LDS layout's list is full
        FT_PROF(0);
        auto word_slot = [&](int code) { return pslot(awlc[(code >> 8) & 0x3fffff], code & 255); };
        if (best_in + 2 * p.beam < kW) {                      // renormalize_scores (:566-603)
            for (int e = tid; e < n_evl; e += NT) {
                const int c = evl_get(e);
                if (c & kFtWordCh) ch_normalize<NE>(wv, word_slot(c), best_in); else ch_normalize<NE>(tv, c, best_in);
            }
            if (!SMALL) {
                for (int i = tid; i < R; i += NT) if (rb_get(cur, i)) ch_normalize<NE>(tv, i, best_in);
                // (the listed nodes' channels are made by this frame's evaluation: it normalises them as it does)
            }
            __syncthreads();
        }
        int32_t nb = 0;
        if (raw_mode) {
            // ---- compute_sen_active (:526-564) + acmod_flags2list (acmod.c:1223-1275) + the scorer's
            //      active-list normalisation (ptm_mgau.c:393-400) on un-normalised rows: the frame's scores
            //      are raw - min over the listed senones, bridging entries included
            //      (acmod_flags2list walks the flags in senone order and bridges a gap of more than 255 with entries of its
            //      own: last + 255, last + 510, ...  Only the first senone of a bitmap word can be that far from its
            //      predecessor; the predecessor is the highest bit of the words before -- a running maximum across the lanes,
            //      continued by a backward walk by the first lane of a wavefront that holds a senone)
            const int nwords = (p.n_sen + 31) >> 5, lane = tid & 63;
            int32_t mn = 0x7fffffff, n_listed_sen = 0;
            if (lists) {
                // ---- the listed senones are scored here, from the frame's top-N lists (ptm_mgau_senone_eval, :326-403): the
                //      bitmap is turned into a list (a prefix sum over its words' populations) so that the ~400 evaluations of a
                //      frame -- twelve weight loads and nine table look-ups each -- spread evenly over the work-items; scores
                //      go to the LDS row the evaluation reads, un-normalised, their minimum is the frame's normaliser
                const int w = tid;                               // (one bitmap word per work-item: n_sen <= 32 NT)
                const uint32_t b = w < nwords ? s_bits[w] : 0u;
                const int hi = b ? w * 32 + 31 - __clz((int)b) : -1;
                int prev = ft_wave_excl<FtMax>(hi);
                auto score = [&](int sen) {
                    const int32_t a = sen_eval_f3n4(smod, l_cw, l_sc, l_la, sen);
                    s_row[sen] = (int16_t)a;                     // (int16 as the scorer's rows, ptm_mgau.c:398-400)
                    mn = min(mn, a);
                };
                if (b) {
                    n_listed_sen = __popc(b);
                    if (prev < 0)
                        for (int q = w - lane - 1; q >= 0; --q) { const uint32_t pb = s_bits[q]; if (pb) { prev = q * 32 + 31 - __clz((int)pb); break; } }
                    const int sen = w * 32 + __ffs((int)b) - 1;
                    for (int last = prev < 0 ? 0 : prev; sen - last > 255;) { last += 255; score(last); }   // bridging entries (rare)
                }
                if (w <= nwords) cnt[w] = w < nwords ? __popc(b) : 0;
                ft_sync<SMALL>();
                const int n_list = ft_block_scan<NT, SMALL>(cnt, nwords + 1, s_scan);
                if (b) {
                    int o = cnt[w];
                    for (uint32_t bb = b; bb; bb &= bb - 1, ++o) {
                        const int sen = w * 32 + __ffs((int)bb) - 1;
                        if (o < kFtListCap) l_list[o] = (uint16_t)sen; else score(sen);      // (a frame with more: scored where found)
                    }
                }
                ft_sync<SMALL>();
                for (int i = tid; i < min(n_list, kFtListCap); i += NT) score((int)l_list[i]);
            }
            else
            for (int w0 = 0; w0 < nwords; w0 += NT) {
                const int w = w0 + tid;
                const uint32_t b = w < nwords ? s_bits[w] : 0u;
                const int hi = b ? w * 32 + 31 - __clz((int)b) : -1;
                int prev = ft_wave_excl<FtMax>(hi);          // (none: the identity, negative)
                if (!b) continue;
                n_listed_sen += __popc(b);
                if (prev < 0)
                    for (int q = w - lane - 1; q >= 0; --q) { const uint32_t pb = s_bits[q]; if (pb) { prev = q * 32 + 31 - __clz((int)pb); break; } }
                const int sen = w * 32 + __ffs((int)b) - 1;
                for (int last = prev < 0 ? 0 : prev; sen - last > 255;) { last += 255; mn = min(mn, (int32_t)row[last]); }
            }
            n_listed_sen = ft_wave_incl<FtAdd>(n_listed_sen); mn = ft_wave_incl<FtMin>(mn);     // (lane 63: the wavefront's)
            if (lane == 63 && n_listed_sen) atomicAdd(&s_nsen, n_listed_sen);
            if (lane == 63 && mn != 0x7fffffff) atomicMin(&s_nb, mn);
            ft_sync<SMALL>();
            nb = (raw_mode & 2) ? 0 : s_nb;              // (bit 1: the rows are final scores -- a scorer that does not normalise over the list)
            FT_PROF(2);
        }
        // ---- evaluate_channels (:605-715): s_red[0] every channel, [2] word level (ngs->last_phone_best_score: the
        //      right-context channels and the single-phone words but </s>)
        {
            const SenRowNorm sr = { row, nb };
            int32_t b_all = kW, b_word = kW;
            FT_PROFW(0);
            FT_PROFD0();
            if (!SMALL) {
                // the tree's items in item order: the record comes in and goes out once; the evaluation leaves, for the pruning,
                // the node's list position in its record and {out, out history, best, score[0]} in the item's compact slot
                // (an idle root's slot says so in its last word; a root's says 1 there: no decision reads a root's score)
                for (int i = tid; i < R; i += NT) {
                    int32_t *const rec = tv.b + (size_t)i * TREC;
                    FtQuad it = FtQuad{ kW, -1, kW, 0 };
                    if (rb_get(cur, i)) {
                        const int32_t sc = ch_eval_tree<NE>(rec, sr, tpall, sseq, false, 0, it);
                        b_all = max(b_all, sc);
                        it.w = 1;
                    }
                    itb[i] = it;
                }
                // the listed nodes' compact channels, position by position: ND + 1 quads in, ND + 1 out, every one in a row with its
                // neighbours'; never multiplexed (the senones are the channel's own)
                // (two positions a work-item at a time: their loads are asked for together.)  The channel is made here: scores and
                // histories from its old place, with the entering score of the pruning that entered it -- or a cleared channel's, entered
                // (hmm_enter into a cleared channel) -- and, in a frame that renormalises (:566-603), less the normaliser
                const bool renorm = best_in + 2 * p.beam < kW;
                const int32_t *const xf = cxf + (size_t)cur * ccap;
                const FtPair *const xp = cxp + (size_t)cur * ccap;
                for (int j0 = tid; j0 < na; j0 += kFtEvalUnroll * NT) {
                    int32_t w[kFtEvalUnroll][4 * ND]; FtSen sq[kFtEvalUnroll]; FtQuad s0[kFtEvalUnroll]; int32_t x[kFtEvalUnroll], nd_[kFtEvalUnroll]; FtPair pl[kFtEvalUnroll];
#pragma unroll
                    for (int u = 0; u < kFtEvalUnroll; ++u) { const int j = min(j0 + u * NT, na - 1); x[u] = xf[j]; nd_[u] = aclc[j]; }
#pragma unroll
                    for (int u = 0; u < kFtEvalUnroll; ++u) {
                        const int src = (x[u] & 0x0fffffff) - 1, kind = (int)((uint32_t)x[u] >> 28);
                        const bool old_ = src >= 0 && kind != 4;
                        // the static side: from the old place, or (a new channel) from the static tables
                        if (src >= 0) { s0[u] = cbuf(nxt, ND)[src]; sq[u] = csen(nxt)[src]; }
                        else { const FtQuad e0 = node_q1[nd_[u]]; sq[u] = sen_of(node_st1[nd_[u]]); s0[u] = FtQuad{ e0.w, e0.x, e0.y, e0.z }; }
                        pl[u] = FtPair{ 0, 0 };
                        if (kind != 0) pl[u] = xp[min(j0 + u * NT, na - 1)];
#pragma unroll
                        for (int k = 0; k < ND; ++k) {
                            FtQuad q = FtQuad{ 0, 0, 0, 0 };
                            if (old_) q = cbuf(nxt, k)[src];
                            w[u][4 * k] = q.x; w[u][4 * k + 1] = q.y; w[u][4 * k + 2] = q.z; w[u][4 * k + 3] = q.w;
                        }
                        if (!old_) {
#pragma unroll
                            for (int k = 0; k < 4 * ND; ++k) w[u][k] = (k < NE || k == 2 * NE) ? kW : -1;
                        }
                        if (kind != 0) { w[u][0] = pl[u].x; w[u][NE] = pl[u].y; }          // hmm_enter
                        if (renorm) {
#pragma unroll
                            for (int k = 0; k < NE; ++k) if (w[u][k] > kW) w[u][k] -= best_in;
                            if (w[u][2 * NE] > kW) w[u][2 * NE] -= best_in;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kFtEvalUnroll; ++u) {
                    const int j = j0 + u * NT;
                    if (j >= na) continue;
                    HmmRegs h;
#pragma unroll
                    for (int k = 0; k < 5; ++k) { h.score[k] = k < NE ? w[u][k] : kW; h.history[k] = k < NE ? w[u][NE + k] : -1; h.senid[k] = 0; }
                    h.senid[0] = (uint16_t)(sq[u].x & 0xffff); h.senid[1] = (uint16_t)((uint32_t)sq[u].x >> 16); h.senid[2] = (uint16_t)(sq[u].y & 0xffff);
                    int tm = (int)((uint32_t)sq[u].y >> 16);
                    if (NE == 5) { h.senid[3] = (uint16_t)((uint32_t)sq[u].y >> 16); h.senid[4] = (uint16_t)(sen_z(sq[u]) & 0xffff); tm = (int)((uint32_t)sen_z(sq[u]) >> 16); }
                    h.out_score = w[u][2 * NE]; h.out_history = w[u][2 * NE + 1]; h.bestscore = kW;
                    uint8_t tpb[NE * (NE + 1)];
                    const uint8_t *tp = ft_tp_row<NE>(tpall, tm, tpb);
                    const int32_t sc = NE == 3 ? vit3(h, tp, sr) : vit5(h, tp, sr);
#pragma unroll
                    for (int k = 0; k < NE; ++k) { w[u][k] = h.score[k]; w[u][NE + k] = h.history[k]; }
                    w[u][2 * NE] = h.out_score; w[u][2 * NE + 1] = h.out_history;
#pragma unroll
                    for (int k = 0; k < ND; ++k) cbuf(cur, k)[j] = FtQuad{ w[u][4 * k], w[u][4 * k + 1], w[u][4 * k + 2], w[u][4 * k + 3] };
                    cbuf(cur, ND)[j] = s0[u]; csen(cur)[j] = sq[u];
                    csum[j] = FtQuad{ h.out_score, h.out_history, h.bestscore, h.score[0] };
                    b_all = max(b_all, sc);
                    }
                }
            }
            for (int e = tid; e < n_evl; e += NT) {
                const int c = evl_get(e);
                if (c & kFtWordCh) {
                    const int32_t sc = ch_eval_rec<NE>(wv.b + (size_t)word_slot(c) * F::REC, sr, tpall, sseq);
                    b_all = max(b_all, sc); b_word = max(b_word, sc);
                }
                else {
                    const int32_t sc = SMALL ? ch_eval<NE>(tv, c, sr, tpall, sseq) : ch_eval_rec<NE>(tv.b + (size_t)c * TREC, sr, tpall, sseq);
                    if (c < W1) b_all = max(b_all, sc);
                    else if (w1w_f[c - W1] != p.finishwid) { b_all = max(b_all, sc); b_word = max(b_word, sc); }   // (:688-694: </s> never sets the best score)
                }
            }
            FT_PROF(18);
            FT_PROFW(1);
            // (one atomic per wavefront: a per-lane atomicMax on one address is compiled to a serial loop over the lanes)
            b_all = ft_wave_incl<FtMax>(b_all); b_word = ft_wave_incl<FtMax>(b_word);
            if ((tid & 63) == 63) {
                if (b_all > kW) atomicMax(&s_red[0], b_all);
                if (b_word > kW) atomicMax(&s_red[2], b_word);
            }
            if (raw_mode) {                                  // the bitmap is free again: cleared for the next frame
                const int nwords = (p.n_sen + 31) >> 5;
                for (int i = tid; i < nwords; i += NT) s_bits[i] = 0u;
            }
        }
        // (LDS only.  The evaluation's stores to the right-context channels' records are still on their way -- the records of
        //  512 utterances do not fit the L2, a store is acknowledged microseconds later -- and nothing needs them yet: the
        //  next reader of a record is the work-item that wrote it (prune_word_chan walks the evaluation list as the evaluation
        //  did), the next writer from another work-item is last_phone_transition, behind the full barrier that ends the
        //  candidates' step.)
        ft_sync<SMALL>();
        FT_PROF(19);
        FT_PROFW(2);
        FT_PROFD1(n_ev);
        // small layout: the next frame's score row and penalties start their way from HBM now -- the barriers from here to
        // the language-model look-ups wait for LDS only, so the loads stay in flight across them; they are written to LDS
        // at the end of the frame (this frame's row has been read: evaluation is over)
        int32_t pre_pen = 0;
        if (SMALL && nf < T) {
            if (lists) lists_load(nf);
            else if (!kFtRowsDevice) row_fetch(nf);
            if (p.has_pl && tid < n_ci) pre_pen = penalties[(size_t)pen_frame(nf) * n_ci + tid];
        }
        // the frame's best scores are complete in s_red behind the barrier above and stay untouched until the next frame's top:
        // every work-item reads them there (no hand-over through work-item 0 and a second barrier); what work-item 0 records
        // below is next read behind later barriers
        const int32_t best_score = s_red[0];
        evals_run = evals_run + (uint32_t)n_ev < evals_run ? 0xffffffffu : evals_run + (uint32_t)n_ev;
        if (tid == 0) {
            s_sc[0] = best_score; s_sc[1] = s_red[2];
            s_evals += (unsigned long long)n_ev;
            s_sc[5] = 0;                                        // n_lastphn_cand
            s_nb = 0x7fffffff;
        }
        FT_PROF(3);
        // dynamic beam (ngram_search_fwdtree.c:1133-1181): the reference compares the utterance's CUMULATIVE evaluation count
        // with maxhmmpf, so from some frame on the histogram is consulted every frame.  It counts every root and every
        // listed node; when there are no more than maxhmmpf of them the running sum never passes it and the loop leaves
        // i == 256 -- known without building the histogram.  Otherwise: 256 bins, one prefix sum, the first bin whose
        // running sum passes maxhmmpf.
        int32_t dyn_beam = p.beam;
        if (p.maxhmmpf != -1 && evals_run > (uint32_t)p.maxhmmpf) {
            const int32_t bw = -p.beam / 256;
            if (R + na <= p.maxhmmpf) dyn_beam = -(256 * bw);
            else {
                for (int i = tid; i < 257; i += NT) s_bins[i] = 0;
                if (tid == 0) s_bins[257] = 256;
                ft_sync<true>();
                for (int i = tid; i < R + na; i += NT) {
                    const int c = i < R ? i : aclc[i - R];
                    int32_t b = (best_score - (SMALL ? tv.at(c, F::BEST) : (i < R ? itb[i].z : csum[i - R].z))) / bw;
                    if (b >= 256) b = 255;
                    atomicAdd(&s_bins[b], 1);
                }
                ft_sync<true>();
                ft_block_scan<NT, true>(s_bins, 257, s_scan);              // s_bins[i + 1] = bins[0] + .. + bins[i]
                {
                    int32_t first = (tid < 256 && s_bins[tid + 1] > p.maxhmmpf) ? tid : 256;
                    first = ft_wave_incl<FtMin>(first);
                    if ((tid & 63) == 63 && first < 256) atomicMin(&s_bins[257], first);
                }
                ft_sync<true>();
                dyn_beam = -(s_bins[257] * bw);
                ft_sync<true>();                                           // (s_bins is reused by the word transitions)
            }
        }
        const int32_t thresh = best_score + dyn_beam;
        const int32_t npt = best_score + p.pbeam, lpt = best_score + p.lpbeam;

        int32_t n_listed;
        if constexpr (!SMALL) {
        // ---- prune_root_chan + prune_nonroot_chan (:722-877) + the last-phone candidates (:824-870), slab layouts.
        //      The same order-free formulation as below (oracle prune_tree_list: items = the roots and the listed nodes; a
        //      decision is a function of the node's and its parent's state as the evaluation left them), arranged for
        //      state that lives in DEVICE memory, where a frame costs the cache lines it touches and the instructions that
        //      ask for them, not arithmetic:
        //        * an item's side comes from arrays indexed by its POSITION (the evaluation's summary and the static quads its
        //          compact channel carries), read in order; a root's from its compact slot and the static quads;
        //        * one work-item per (item, child) PAIR, four consecutive pairs at a time with their loads asked for
        //          together; a child that is itself listed is decided twice -- by its own item (whose outcome counts) and
        //          by its parent's pair (which needs it for the next list);
        //        * a pair's OTHER node, if listed, is found through the list's index (bitmap, rank, position: LDS) and read from
        //          the summary array; if not listed its state is known without asking;
        //        * decisions only READ channels: the pair that gives a node its place in the next list writes where the node's channel
        //          comes from and what happens to it (cxfer), and the next frame makes it there;
        //        * positions in the next active list = prefix sums over the pairs' outcomes in pair order (= list order);
        //        * the frame's penalty row and the chunk's items in LDS.
        //      Facts the decisions rely on instead of reading them:
        //        a node that is NOT in the active list has no channel (the reference's was cleared when it left the list, or never
        //        entered): its frame is below f and its scores are WORST_SCORE, so "frame < f || news > score" is true without
        //        looking; a node that IS in the list was entered or retained for this frame.
        {
            const int n_item = R + na;
            constexpr int KP = kFtPairs;
            int carry_l = 0, carry_c = 0;                        // next list's entries / candidates so far (uniform)
            const FtQuad *const cs0 = cbuf(cur, ND);
            int32_t *const xfer = cxf + (size_t)nxt * ccap;    // the next list's
            FtPair *const xfpl = cxp + (size_t)nxt * ccap;
            bool over = false;                                   // the next list outgrows the compact buffers (uniform)
            // (a chunk takes as many items as fill ONE round of pairs -- KP pairs a work-item -- judged from the chunk before: a second round
            //  with a fifth of its places taken costs what a full one does)
            int nci = kFtChunkStart < kPrIC ? kFtChunkStart : kPrIC;
            for (int c0 = 0, nci_used = 0; c0 < n_item; c0 += nci_used) {
                nci_used = nci;
                // -- the chunk's items, two consecutive ones a work-item -> LDS; their pairs counted: the item's own entry (listed nodes),
                //    its children (retained items), the words whose penultimate phone it is (retained items whose out score can reach the
                //    last-phone beam, :824-870)
                int32_t np[kFtIpt];
                {
                    int node[kFtIpt]; FtQuad it[kFtIpt], q1[kFtIpt];
#pragma unroll
                    for (int u = 0; u < kFtIpt; ++u) {
                        const int i = kFtIpt * tid + u < nci_used ? c0 + kFtIpt * tid + u : n_item;      // (beyond the chunk: no item)
                        node[u] = -1; it[u] = FtQuad{ kW, -1, kW, 0 }; q1[u] = FtQuad{ 0, 0, 0, -1 };
                        if (i < R) { node[u] = i; it[u] = itb[i]; q1[u] = node_q1[i]; }
                        else if (i < n_item) {
                            const FtQuad a = cs0[i - R];         // first entry, parent | ci << 24, first entry's index, children | words << 16
                            it[u] = csum[i - R];
                            node[u] = aclc[i - R]; q1[u] = FtQuad{ a.y, a.z, a.w, a.x };
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kFtIpt; ++u) {
                        const int li = kFtIpt * tid + u, i = li < nci_used ? c0 + li : n_item;
                        const bool active = i < R ? it[u].w != 0 : node[u] >= 0;
                        const bool fl = active && it[u].z > thresh;
                        const int nk = fl ? (q1[u].z & 0xffff) : 0, nw = (int)((uint32_t)q1[u].z >> 16);
                        const bool cand = fl && nw > 0 && (p.has_pl || it[u].x + p.pip > lpt);
                        np[u] = node[u] >= 0 ? (i >= R ? 1 : 0) + nk + (cand ? nw : 0) : 0;
                        s_it_node[li] = node[u]; s_it_out[li] = it[u].x; s_it_outh[li] = it[u].y; s_it_fp[li] = (fl ? 1 : 0) | (nk << 1);
                        s_it_k0[li] = q1[u].y; s_it_par[li] = q1[u].x; s_it_sc0[li] = it[u].w; s_it_kid0[li] = q1[u].w;
                        if (i < R && fl) { tv.at(node[u], F::FRAME) = nf; rb_set(nxt, i); }     // a retained root stays (no decision reads this stamp before it is >= f)
                    }
                }
                FT_PROF(28);
                // exclusive prefix sums over the chunk's items, in item order (two consecutive items a work-item)
                int32_t n_pair;
                {
                    const int lane = tid & 63, wv_ = tid >> 6;
                    int32_t sp = 0;
#pragma unroll
                    for (int u = 0; u < kFtIpt; ++u) sp += np[u];
                    const int32_t ip = ft_wave_incl<FtAdd>(sp);
                    if (lane == 63) s_scan[wv_] = ip;
                    ft_sync<true>();
                    int32_t bp_ = 0; n_pair = 0;
#pragma unroll
                    for (int w = 0; w < NT / 64; ++w) {
                        const int32_t a_ = s_scan[w];
                        n_pair += a_;
                        if (w < wv_) bp_ += a_;
                    }
                    const int32_t op = bp_ + ip - sp;
                    { int32_t o_ = op;
#pragma unroll
                      for (int u = 0; u < kFtIpt; ++u) { s_it_poff[kFtIpt * tid + u] = o_; o_ += np[u]; } }
                    if (tid == NT - 1) s_it_poff[kPrIC] = n_pair;
                    ft_sync<true>();                                 // (the chunk's LDS arrays are complete; s_scan is free again)
                }
                if (kFtChunkAdapt) {
                    const long long want = (long long)KP * NT * nci_used / (n_pair > 0 ? n_pair : 1);
                    nci = (int)(want > kPrIC ? kPrIC : (want < 4 * 64 ? 4 * 64 : want)) & ~1;
                }
                FT_PROF(5);
                // -- the chunk's pairs, four consecutive ones a work-item: the item's own entry first (listed nodes), then its children
                //    in sibling order, then its penultimate-phone words in chain order
                for (int p0 = 0; p0 < n_pair; p0 += KP * NT) {
                    int li[KP], q[KP], c[KP], cci[KP]; bool val[KP];
                    const int jb = p0 + KP * tid;
                    // (which item a pair belongs to: a bisection of the chunk's offsets in LDS)
                    {
                        int l0 = jb < n_pair ? ft_seg_find(s_it_poff, kPrIC, jb) : 0;
#pragma unroll
                        for (int v = 0; v < KP; ++v) {
                            const int j = jb + v;
                            val[v] = j < n_pair;
                            if (val[v] && s_it_poff[l0 + 1] <= j) { ++l0; if (s_it_poff[l0 + 1] <= j) l0 = ft_seg_find(s_it_poff, kPrIC, j); }
                            li[v] = l0;
                            const int self = (c0 + l0 >= R) ? 1 : 0;
                            q[v] = val[v] ? j - s_it_poff[l0] - self : 0;
                        }
                    }
                    FT_PROF(1);
                    // a child's id and phone (a word's id and last phone): the item's first entry came with the item; the others from the
                    // static table
#pragma unroll
                    for (int v = 0; v < KP; ++v) {
                        c[v] = -1; cci[v] = 0;
                        if (val[v]) {
                            if (q[v] < 0) { c[v] = s_it_node[li[v]]; cci[v] = (uint32_t)s_it_par[li[v]] >> 24; }
                            else {
                                // (the first child from LDS, the others from device memory: two loads -- the compiler would make it ONE
                                //  through a selected pointer, i.e. a generic access that waits on both memory counters; the empty
                                //  asm pins the LDS value in a register before the other load exists)
                                uint32_t kc_ = (uint32_t)s_it_kid0[li[v]];
#if defined(__HIP_DEVICE_COMPILE__)
                                asm volatile("" : "+v"(kc_));
#endif
                                if (q[v] > 0) kc_ = (uint32_t)kids_ci[s_it_k0[li[v]] + q[v]];
                                c[v] = (int)(kc_ & 0xffffffu); cci[v] = (int)(kc_ >> 24);
                            }
                        }
                    }
                    // the OTHER node: a child's {out, out history, best, position + 1} and score[0] -- for an item's own entry the
                    // PARENT's.  A root's come from its record (its last word is its frame stamp); a listed node's from the summary
                    // array at its position (bitmap -> rank -> position: LDS); a node that is not listed has nothing to say
                    FtQuad qx[KP]; int32_t csc[KP];
#pragma unroll
                    for (int v = 0; v < KP; ++v) {
                        qx[v] = FtQuad{ kW, -1, kW, -1 };
                        csc[v] = kW;
                        if (val[v] && q[v] < (s_it_fp[li[v]] >> 1)) {     // (a word's pair looks at nothing)
                            const int o_ = q[v] < 0 ? (s_it_par[li[v]] & 0xffffff) : c[v];
                            if (o_ < R) qx[v] = *reinterpret_cast<const FtQuad *>(tv.b + (size_t)o_ * TREC + F::OUT);
                            else {
                                const uint32_t wb = s_lb[o_ >> 5];
                                if ((wb >> (o_ & 31)) & 1u) {
                                    const int r = s_sup[o_ >> 15] + (int)s_pre[o_ >> 5] + __popc(wb & ((1u << (o_ & 31)) - 1u));
                                    const int ps = perm_lds ? (int)s_perm[r] : g_perm[r];
#ifdef PSGPU_FT_CHECK_LISTS
                                    if (ps >= na || aclc[ps] != o_) {
                                        int where = -1, nb = 0;
                                        for (int z = 0; z < na; ++z) if (aclc[z] == o_) where = z;
                                        for (int z = 0; z < p.lb_words; ++z) nb += __popc(s_lb[z]);
                                        printf("frame %d: node %d -> rank %d -> position %d holds node %d (%d listed; the node is at %d; %d bits set; pair q %d of item %d node %d, perm_lds %d)\n", f, o_, r, ps, ps < na ? aclc[ps] : -1, na, where, nb, q[v], c0 + li[v], s_it_node[li[v]], (int)perm_lds);
                                        abort();
                                    }
#endif
                                    const FtQuad sm = csum[ps];
                                    qx[v] = FtQuad{ sm.x, sm.y, sm.z, ps + 1 };
                                    csc[v] = sm.w;
                                }
                            }
                        }
                    }
                    int32_t bit[KP], cbit[KP], act[KP], a_news[KP], a_outh[KP], c_at[KP];
#pragma unroll
                    for (int v = 0; v < KP; ++v) {
                        bit[v] = 0; cbit[v] = 0; act[v] = 0; a_news[v] = 0; a_outh[v] = -1; c_at[v] = -1;
                        if (!val[v]) continue;
                        const bool self = q[v] < 0;
                        const int my_pos = c0 + li[v] - R;           // the item's position in the active list (a listed node's)
                        if (!self && q[v] >= (s_it_fp[li[v]] >> 1)) {
                            // a last-phone candidate (:824-870): {word, score without the word insertion penalty, history}
                            a_news[v] = s_it_out[li[v]] + p.pip; a_outh[v] = s_it_outh[li[v]];
                            cbit[v] = (a_news[v] + ft_pen(cci[v]) > lpt) ? 1 : 0;
                            continue;
                        }
                        const int P = self ? (s_it_par[li[v]] & 0xffffff) : s_it_node[li[v]];
                        // the parent's side: a root (active: its stamp is this frame's or the next's), or a node with a list position
                        const bool p_root = P < R;
                        const bool p_active = !self || (p_root ? qx[v].w >= f : qx[v].w > 0);
                        const int p_pos = self ? (p_root ? -1 : qx[v].w - 1) : my_pos;
                        const bool p_flag = self ? (p_active && qx[v].z > thresh) : (s_it_fp[li[v]] & 1);
                        const int32_t p_out = self ? qx[v].x : s_it_out[li[v]], p_outh = self ? qx[v].y : s_it_outh[li[v]];
                        // the node's side
                        const bool in_acl = self || qx[v].w > 0;
                        const int c_pos = self ? my_pos : qx[v].w - 1;
                        const bool retc = self ? (s_it_fp[li[v]] & 1) : (in_acl && qx[v].z > thresh);
                        const int32_t c_score = self ? s_it_sc0[li[v]] : csc[v];
                        const int32_t news = (p_active ? p_out : kW) + p.pip;
                        const bool parent_can = p_active && p_flag && (p.has_pl || news > npt) && (news + ft_pen(cci[v]) > npt);
                        const bool parent_first = p_root || !in_acl || p_pos < c_pos;
                        bool fire;                                   // (frame < f: exactly the nodes that are not listed, see above)
                        if (!in_acl) fire = parent_can;
                        else if (parent_first || retc) fire = parent_can && news > c_score;
                        else fire = parent_can;
                        const bool entered_first = fire && parent_first;
                        const bool listed = fire && (p_root || !(in_acl && !parent_first && retc));
                        // what the decision does to the channel of a node it lists: 0 = kept, 1 = a new channel, entered (an unlisted node),
                        // 2 = entered (a listed node: score[0] and history[0]), 4 = cleared, then entered; a node nobody lists was cleared
                        a_news[v] = news; a_outh[v] = p_outh;
                        // (a listed node is decided twice with the same inputs and the same outcome: by its own entry and by its
                        //  parent's pair; whichever of the two gives the node its place in the next list also says what becomes of its
                        //  channel)
                        const bool clr = in_acl && !retc && !entered_first;
                        if (self) {
                            act[v] = fire ? 2 : 0;
                            bit[v] = (retc && !entered_first) ? 1 : 0;
                            c_at[v] = my_pos;
                        }
                        else {
                            act[v] = !in_acl ? 1 : (clr ? 4 : 2);
                            bit[v] = listed ? 1 : 0;
                            c_at[v] = in_acl ? c_pos : -1;
                        }
                    }
                    FT_PROF(13);
                    // positions in the next active list and in the candidate list: pair order
                    {
                        const int lane = tid & 63, wv_ = tid >> 6;
                        int32_t sb = 0, sc_ = 0;
#pragma unroll
                        for (int v = 0; v < KP; ++v) { sb += bit[v]; sc_ += cbit[v]; }
                        const int32_t ib = ft_wave_incl<FtAdd>(sb), ic = ft_wave_incl<FtAdd>(sc_);
                        if (lane == 63) { s_scan[wv_] = ib; s_scan[NT / 64 + wv_] = ic; }
                        FT_PROF(22);
                        ft_sync<true>();
                        FT_PROF(31);
                        int32_t base = 0, tot = 0, base_c = 0, tot_c = 0;
#pragma unroll
                        for (int w = 0; w < NT / 64; ++w) {
                            const int32_t a_ = s_scan[w], b_ = s_scan[NT / 64 + w];
                            tot += a_; tot_c += b_;
                            if (w < wv_) { base += a_; base_c += b_; }
                        }
                        int o = carry_l + base + ib - sb, oc = carry_c + base_c + ic - sc_;
                        if (carry_l + tot > ccap) over = true;       // (uniform: every work-item sees the same totals)
                        if (!over) {
#pragma unroll
                            for (int v = 0; v < KP; ++v) {
                                if (!val[v]) continue;
                                if (cbit[v]) { cand_wid[oc] = c[v]; cand_score[oc] = a_news[v] - p.nwpen; cand_bp[oc] = a_outh[v]; ++oc; }
                                // a node of the next list: its place, and where its channel comes from -- its position in this frame's list
                                // (or -1: a new channel) and what the decision does to it; the next frame's evaluation makes the channel
                                if (bit[v]) {
                                    acln[o] = c[v]; xfer[o] = (c_at[v] + 1) | (act[v] << 28);
                                    if (act[v]) xfpl[o] = FtPair{ a_news[v], a_outh[v] };
                                    ++o;
                                }
                            }
                        }
                        carry_l += tot; carry_c += tot_c;
                        ft_sync<true>();                             // (s_scan; the chunk's LDS arrays before the next chunk overwrites them)
                    }
                }
                FT_PROF(6);
            }
            n_listed = carry_l;
            if (tid == 0) { s_sc[5] = carry_c; s_red[7] = 0; if (over) s_sc[6] = 4; }     // status 4: more than ccap tree channels listed in a frame
            __syncthreads();                                     // (device memory: every decision has been taken; the next list and its cxfer are complete)
            if (over) break;
            for (int i = tid; i < p.lb_words; i += NT) s_lb[i] = 0u;       // (this frame's index has been read for the last time)
            __syncthreads();                                     // (the bitmap is clear)
            build_index(acln.b, n_listed);
            FT_PROF(4);
        }
        } else {
            // ---- prune_root_chan + prune_nonroot_chan (:722-877), order-free formulation, one work-item per tree NODE.  The items
            //      are the roots and the listed nodes (oracle prune_tree_list); a decision is a function of the node's state and its
            //      parent's as the evaluation left them.  Reads of another node's state go to the snapshot (o_out, o_outh, flag, pos)
            //      of a root or listed node, writes to the node's own channel and decision word.  No work-item walks a node's
            //      children (those walks -- three of them, their lengths uneven -- were a quarter of a frame): the children array
            //      (CSR) is gone through position by position, a child's place among its listed siblings is a difference of one
            //      prefix sum over that array, an item's place in the next list a prefix sum over the items.
            int32_t *const Sarr = cnt, *const ibase = cnt + (R + N + 1);         // [M + 1], [R + na]
            for (int q = tid; q < na; q += NT) pos[aclc[q]] = q;
            for (int i = tid; i < R + na; i += NT) {
                const int node = i < R ? i : aclc[i - R];
                const bool active = i < R ? tv.at(node, F::FRAME) >= f : true;
                o_out[node] = tv.at(node, F::OUT); o_outh[node] = tv.at(node, F::OUTH);
                flag[node] = (active && tv.at(node, F::BEST) > thresh) ? 1 : 0;
            }
            ft_sync<SMALL>();
            FT_PROF(4);
            {
                // -- the decisions, child by child of the children array; listed children counted by a running prefix sum
                int32_t carry = 0;
                for (int j0 = 0, rnd = 0; j0 < p.M; j0 += NT, ++rnd) {
                    const int j = j0 + tid;
                    int32_t b2[1] = { 0 };
                    if (j < p.M) {
                        const int c = kids[j], P = parent[c], pc = pos[c];
                        const int pp = P < R ? 0 : pos[P];
                        const bool in_acl = pc >= 0, par_active = P < R || pp >= 0;
                        const int pflag = par_active ? (flag[P] & 1) : 0;
                        int32_t dec = 0;
                        if (in_acl || pflag) {               // (a node that is not listed under a parent that is not retained: nothing happens to it)
                            const int32_t news = (par_active ? o_out[P] : kW) + p.pip;
                            const bool parent_can = par_active && pflag && (p.has_pl || news > npt) && (news + ft_pen(node_ci[c]) > npt);
                            const bool parent_first = P < R || !in_acl || pp < pc;
                            const bool retc = in_acl && (flag[c] & 1);
                            bool fire;
                            if (!in_acl || parent_first) fire = parent_can && (tv.at(c, F::FRAME) < f || news > tv.at(c, F::SCORE));
                            else if (retc)               fire = parent_can && news > tv.at(c, F::SCORE);
                            else                         fire = parent_can;
                            const bool entered_first = fire && parent_first;
                            const bool listed = fire && (P < R || !(in_acl && !parent_first && retc));
                            if (in_acl && !retc && !entered_first) ch_clear<NE>(tv, c);
                            if (retc) tv.at(c, F::FRAME) = nf;
                            if (fire) ch_enter<NE>(tv, c, news, o_outh[P], nf);
                            dec = (fire ? (listed ? 2 : 4) : 0) | ((retc && !entered_first) ? 8 : 0);
                        }
                        o_frame[c] = dec;
                        b2[0] = (dec & 2) ? 1 : 0;
                    }
                    int32_t tot[1];
                    ft_scan_tid<NT, 1, SMALL>(b2, s_scan + (rnd & 1) * (NT / 64), tot);
                    if (j < p.M) Sarr[j] = carry + b2[0];
                    carry += tot[0];
                }
                if (tid == 0) Sarr[p.M] = carry;
            }
            FT_PROF(28);
            ft_sync<SMALL>();
            FT_PROF(5);
            {
                // -- the items in list order: place in the next list (itself when it stays, then its listed children) and the
                //    last-phone candidates (list order, homophone chain inside), two prefix sums over the items in one pass
                int32_t carry_l = 0, carry_c = 0;
                for (int i0 = 0, rnd = 0; i0 < R + na; i0 += NT, ++rnd) {
                    const int i = i0 + tid;
                    int32_t v[2] = { 0, 0 };
                    int node = 0; bool self = false, fl = false; int32_t news = 0;
                    if (i < R + na) {
                        node = i < R ? i : aclc[i - R];
                        fl = (flag[node] & 1) != 0;
                        self = i >= R && (o_frame[node] & 8);
                        v[0] = (self ? 1 : 0) + Sarr[kid_off[node + 1]] - Sarr[kid_off[node]];
                        news = o_out[node] + p.pip;
                        if (fl && (p.has_pl || news > lpt))
                            for (int w = node_pw[node]; w >= 0; w = homo_f[w]) v[1] += (news + ft_pen(dlast_f[w]) > lpt) ? 1 : 0;
                        if (i < R && fl) tv.at(i, F::FRAME) = nf;          // a retained root stays
                    }
                    int32_t tot[2];
                    ft_scan_tid<NT, 2, SMALL>(v, s_scan + 2 * (NT / 64) + (rnd & 1) * 2 * (NT / 64), tot);
                    if (i < R + na) {
                        const int32_t o = carry_l + v[0];
                        ibase[i] = o + (self ? 1 : 0);                    // where its listed children start
                        if (self) acln[o] = node;
                        int oc = carry_c + v[1];
                        if (fl && (p.has_pl || news > lpt))
                            for (int w = node_pw[node]; w >= 0; w = homo_f[w])
                                if (news + ft_pen(dlast_f[w]) > lpt) {
                                    cand_wid[oc] = w; cand_score[oc] = news - p.nwpen; cand_bp[oc] = o_outh[node]; ++oc;
                                }
                    }
                    carry_l += tot[0]; carry_c += tot[1];
                }
                n_listed = carry_l;
                if (tid == 0) { s_sc[5] = carry_c; s_red[7] = 0; }
            }
            ft_sync<SMALL>();
            FT_PROF(6);
            // -- the listed children to their places
            for (int j = tid; j < p.M; j += NT) {
                const int c = kids[j];
                if (o_frame[c] & 2) {
                    const int P = parent[c];
                    acln[ibase[P < R ? P : R + pos[P]] + Sarr[j] - Sarr[kid_off[P]]] = c;
                }
            }
        }
        __syncthreads();                                     // (device memory: the evaluation's records, tb.idx[f] -- see above)
        if (SMALL) { for (int q = tid; q < na; q += NT) pos[aclc[q]] = -1; }     // (pos is read by nobody until the next frame's pruning sets it)
        FT_PROF(7);

        // ---- word level: last_phone_transition (:884-1035).  Candidates of one frame name distinct words (a word has
        //      one penultimate tree node, a node one place in the active list).  Each candidate's best predecessor --
        //      exit score + language score over the back-pointers of its start frame, the look-ups that dominate this
        //      step -- is searched one work-item per (candidate, back-pointer) pair: a loop over the back-pointers in one
        //      work-item is a chain of dependent loads per iteration.  last_ltrans (lt_*) is the reference's per-word
        //      cache keyed by start frame.  Should two candidates ever share a word, the reference's loops are run as
        //      written by one thread.
        const int n_cand = s_sc[5];
        // slab layouts: the word level's per-candidate / per-active-word scratch arrays in LDS -- in the pruning's item arrays, idle until
        // the next frame's pruning -- while the frames' counts fit (WLDS; else the slab's, another instantiation): their prefix sums, the
        // searches in them and the survivor counts (atomics) are then LDS operations
        // (LDS: `cnt` -- the candidates' counts and their prefix sums, later four arrays over the active words / three over the
        //  single-phone words: the arrays that are searched and counted into; the two plain per-candidate arrays stay in the slab)
        constexpr bool wl_lds = WLDS;
        const int wstf = wl_lds ? max(naw, n1) + 2 : p.n_w + 1;
        const int wl_a = max(n_cand + 1, 4 * wstf);
        if (WLDS && wl_a > min(9 * kPrIC, p.wl_cap)) { if (tid == 0) s_sc[6] = 6; ft_sync<SMALL>(); break; }     // status 6
        int32_t *const cntf = wl_lds ? s_pool : cnt, *const cnt2f = cnt2, *const cnt3f = cnt3;
        {
            for (int i = tid; i < n_cand; i += NT) {
                const int cb = cand_bp[i], w = cand_wid[i];
                // O(1) per candidate: the frame stamp of the word.  (Two candidates naming one word would need two paths to that word
                // in the tree: create_search_channels builds one per dictionary entry.  The reference's loops would cope; here it ends
                // the utterance with status 3 -- looked at behind this step's barrier, nothing in between is kept.)
                if (atomicExch(&cand_mark[w], f) == f) s_red[7] = 1;
                int need = 0, b0 = 0, sf = -1;
                if (cb != -1) {
                    const int ef = BPC(tb, B_FRAME, cb);                 // (one trip with the exit score's columns)
                    cand_score[i] -= bssx ? ft_exit_score_x(tb, bssx, n_ci, cb, dfirst_f[w]) : ft_exit_score_bf(tb, rs_cimap, n_ci, cb, dfirst_f[w]);
                    // (the frame's entries [bp_table_idx[ef], bp_table_idx[ef + 1]): the entry carries both, same cache line)
                    const int32_t e0 = BPC(tb, B_F0, cb), e1 = BPC(tb, B_F1, cb);
                    if (lt_sf[w] != ef + 1) { b0 = e0; need = e1 - b0; sf = ef + 1; }
                }
                cntf[i] = need; cnt2f[i] = b0; cnt3f[i] = sf;
                ckey[i] = ft_key_floor(kW);
            }
            if (tid == 0) cntf[n_cand] = 0;
            ft_sync<SMALL>();
            if (s_red[7] != 0) { if (tid == 0) s_sc[6] = 3; ft_sync<SMALL>(); break; }
            const int n_pair = ft_block_scan<NT, SMALL>(cntf, n_cand + 1, s_scan);
            FT_PROF(29);
            for (int j = tid; j < n_pair; j += NT) {
                const int i = ft_seg_find(cntf, n_cand, j), bp = cnt2f[i] + (j - cntf[i]), w = cand_wid[i];
                // (every load of the pair before the first test: three trips to device memory -- the entry's columns; context map
                //  and language-model entry; stacked score -- where the tests in between made seven)
                const int32_t valid = BPC(tb, B_VALID, bp), real = BPC(tb, B_REAL, bp), preal = BPC(tb, B_PREAL, bp);
                int32_t dscr = bssx ? ft_exit_score_x(tb, bssx, n_ci, bp, dfirst_f[w]) : ft_exit_score_bf(tb, rs_cimap, n_ci, bp, dfirst_f[w]);
                const int32_t lmv = p.use_trie ? 0 : ft_lm_dense(p, lmtab, dbase_f[w], real, preal);
                if (!valid) continue;
                if (dscr > kW) dscr += p.use_trie ? ft_lm(p, lmtab, dbase_f[w], real, preal) : lmv;
                atomicMax(&ckey[i], ft_key(dscr, bp));
            }
            ft_sync<SMALL>();
            FT_PROF(30);
            int32_t bestscore = kW;
            for (int i = tid; i < n_cand; i += NT) {
                const int w = cand_wid[i];
                if (cnt3f[i] >= 0) {                          // searched: best = WORST_SCORE keeps the old back-pointer
                    const unsigned long long k = ckey[i];
                    if (ft_key_none(k)) lt_dscr[w] = kW;
                    else { lt_dscr[w] = ft_key_score(k); lt_bp[w] = ft_key_bp(k); }
                    lt_sf[w] = cnt3f[i];
                }
                const int32_t score = cand_score[i] + lt_dscr[w];
                cand_score[i] = score;
                cand_bp[i] = lt_bp[w];
                bestscore = max(bestscore, score);
            }
            bestscore = ft_wave_incl<FtMax>(bestscore);
            if ((tid & 63) == 63 && bestscore > kW) atomicMax(&s_sc[1], bestscore);
        }
        ft_sync<SMALL>();
        FT_PROF(8);
        {
            // ---- last_phone_transition's entering loop (:1004-1030).  One work-item per (entering candidate, right context):
            //      the word's slots are exactly its right contexts, so "allocate the missing ones, then enter every present
            //      one" is, per slot, "create if missing, then enter".
            const int32_t cthresh = s_sc[1] + p.lponlybeam;
            {
                for (int i = tid; i < n_cand; i += NT) {
                    const int w = cand_wid[i];
                    cntf[i] = cand_score[i] > cthresh ? wc_off[w + 1] - wc_off[w] : 0;
                    cnt2f[i] = 0;                                // "entered a channel"
                    if (!SMALL && cntf[i] > 0 && wblk[w] < 0) {
                        // the word's block of the pool (a frame's candidates name distinct words): one that no word holds -- which one
                        // is nobody's business.  None left: status 5, the frame is not finished
                        const int at = atomicAdd(&s_nfree, -1) - 1;
                        if (at < 0) s_red[7] = 2;
                        wblk[w] = at < 0 ? 0 : rcfree[at];
                    }
                }
                if (tid == 0) cntf[n_cand] = 0;
                ft_sync<SMALL>();
                const int n_ent = ft_block_scan<NT, SMALL>(cntf, n_cand + 1, s_scan);
                for (int j = tid; j < n_ent; j += NT) {
                    const int i = ft_seg_find(cntf, n_cand, j), w = cand_wid[i], r = j - cntf[i], slot = pslot(w, r);
                    int32_t *const rec = wv.b + (size_t)slot * F::REC;
                    if (s_red[7] == 2) continue;                 // (the pool ran out: see above)
                    if (!present[slot]) {
                        // ngram_search_alloc_all_rc (ngram_search.c:583-633), then hmm_enter into the cleared channel (its frame
                        // is -1: the test below always passes): the whole record is written at once
                        ch_init_enter_rec_s<NE>(rec, slot_sen + ((size_t)wc_off[w] + r) * (NE <= 3 ? 4 : 8), cand_score[i], cand_bp[i], nf);
                        present[slot] = 1;
                        cnt2f[i] = 1;
                    }
                    else if (rec[F::FRAME] < f || cand_score[i] > rec[F::SCORE]) {
                        ch_enter<NE>(wv, slot, cand_score[i], cand_bp[i], nf);
                        cnt2f[i] = 1;
                    }
                }
            }
            __syncthreads();                                     // (device memory is exchanged here)
            if (!SMALL && s_red[7] == 2) { if (tid == 0) s_sc[6] = 5; __syncthreads(); break; }     // status 5: the right-context channels' pool is full
            FT_PROF(9);
            {                                            // stable compaction by a prefix sum
                int32_t nawl;
                if (n_cand <= NT) {                          // (one candidate a work-item: its flag and place in registers, one barrier)
                    int32_t v[1] = { tid < n_cand ? cnt2f[tid] : 0 }, t1[1];
                    const int32_t mine = v[0];
                    ft_scan_tid<NT, 1, SMALL>(v, s_scan, t1);
                    nawl = t1[0];
                    if (mine) { const int w = cand_wid[tid]; awln[v[0]] = w; word_active[w] = 1; }
                }
                else {
                    nawl = ft_block_scan<NT, SMALL>(cnt2f, n_cand, s_scan);
                    for (int i = tid; i < n_cand; i += NT)
                        if ((i + 1 < n_cand ? cnt2f[i + 1] : nawl) != cnt2f[i]) {
                            const int w = cand_wid[i];
                            awln[cnt2f[i]] = w; word_active[w] = 1;
                        }
                }
                if (tid == 0) s_red[5] = nawl;
            }
            ft_sync<SMALL>();
            FT_PROF(10);
            // ---- prune_word_chan (:1038-1128): keep / free the right-context channels, count the survivors per word,
            //      note whether the word exits.  One work-item per right-context channel of the evaluation list (the channels
            //      this frame's candidates have just allocated were entered for the next frame and have no score yet: the
            //      reference's walk neither counts nor frees them), survivors counted per word by atomics
            const int32_t nwt = s_sc[1] + p.wbeam, lpth = s_sc[1] + p.lponlybeam;
            const int wst = wstf;                                 // naw <= n_w
            int32_t *const w_k = cntf, *const w_exit = cntf + wst, *const w_bp = cntf + 2 * wst, *const w_bss = cntf + 3 * wst;
            for (int i = tid; i < naw; i += NT) { w_k[i] = 0; w_exit[i] = 0; }
            ft_sync<SMALL>();
            for (int e = tid; e < n_evl; e += NT) {
                const int c = evl_get(e);
                if (!(c & kFtWordCh)) continue;
                const int i = (c >> 8) & 0x3fffff, slot = word_slot(c);
                const FtQuad q = ch_summary<NE>(wv.b + (size_t)slot * F::REC);      // out, out history, best, frame
                if (q.z > lpth) {
                    wv.at(slot, F::FRAME) = nf;
                    atomicAdd(&w_k[i], 1);
                    if (q.x > nwt) atomicOr(&w_exit[i], 1);
                }
                else if (q.w != nf) present[slot] = 0;
            }
            ft_sync<SMALL>();                                    // (its store -- the frame stamp -- is read by nobody this frame)
            FT_PROF(11);
            for (int i = tid; i <= naw; i += NT) {
                // inputs of the three prefix sums below: next active word list, back-pointers, score-stack entries
                const int w = i < naw ? awlc[i] : 0;
                const int k = i < naw ? w_k[i] : 0, ex = i < naw ? w_exit[i] : 0;
                w_k[i] = (k > 0 && !word_active[w]) ? 1 : 0;
                if (!SMALL && i < naw && k == 0 && !word_active[w]) {
                    // the word leaves the active list with no channel left (every `present` byte of its block is clear): the block
                    // goes back to the pool (pushes only in this step, pops only in the entering step: barriers between)
                    const int blk = wblk[w];
                    wblk[w] = -1;
                    rcfree[atomicAdd(&s_nfree, 1)] = blk;
                }
                w_bp[i] = ex ? 1 : 0;
                w_bss[i] = ex ? wc_off[w + 1] - wc_off[w] : 0;     // = rssid n_ssid of the word's last two phones
            }
            ft_sync<SMALL>();
            const int32_t bpidx = s_sc[3], bss_head = s_sc[4], nawl0 = s_red[5];     // (rewritten below, after the scans' barriers)
            int32_t tot[3];
            {
                int32_t *const arr[3] = { w_bp, w_bss, w_k };
                ft_block_scan_k<NT, 3, SMALL>(arr, naw + 1, s_scan, tot);
            }
            const int32_t n_exit = tot[0], n_bss = tot[1], n_app = tot[2];
            for (int i = tid; i < naw; i += NT)
                if (w_k[i + 1] != w_k[i]) {
                    const int w = awlc[i];
                    awln[nawl0 + w_k[i]] = w; word_active[w] = 1;
                }
            if (tid == 0) {
                // (a full table: status 1, the frame's exits are not written and the counts stay what the tables hold)
                if (bpidx + n_exit + n1 >= tb.bp_cap || bss_head + n_bss + n_ci >= tb.bss_cap) s_sc[6] = 1;
                else { s_sc[3] = bpidx + n_exit; s_sc[4] = bss_head + n_bss; }
                s_red[5] = nawl0 + n_app;
            }
            ft_sync<SMALL>();
            FT_PROF(12);
            if (!s_sc[6]) {
                // ---- the exits' back-pointers (ngram_search_save_bp, ngram_search.c:376-498).  An exiting word owns one
                //      new entry and wc_off-many score-stack slots; the reference merges its exiting channels into the entry
                //      in right-context order (the first creates it, a better one updates it).  One wavefront per exiting
                //      word, one lane per right context (n_ci <= 64): a lane reads its channel, writes its score-stack slot
                //      and fetches the language-model state of its history; the merge is a running maximum across lanes.
                //        entry score / path = those of the last lane that raised the running maximum (strictly);
                //        real word ids      = set_real_wid's, which the reference runs BEFORE it stores a new path
                //                             (:420-436): the ids of the path the entry held before the last update that
                //                             changed them -- or the creating path's when no update did.
                const int lane = tid & 63;
                for (int e = tid >> 6; e < n_exit; e += NT / 64) {
                    const int i = ft_seg_find(w_bp, naw, e), w = awlc[i], j0 = w_bss[i], nrc = w_bss[i + 1] - j0, bpi = bpidx + e;
                    const int32_t w_last2 = d_last2[w], w_filler = d_filler[w];      // (asked for with the channels' records, not after the merge)
                    // small layout: the row of the context map that ngram_search_exit_score (ngram_search.c:653-674) would read for this
                    // entry later, right context phone by right context phone -- read ONCE, now, beside the channels' records
                    int32_t cmrow = 0;
                    if (SMALL && lane < n_ci) cmrow = rs_cimap[((size_t)dlast_f[w] * n_ci + w_last2) * n_ci + lane];
                    int32_t it[4] = { kW, -1, -1, -1 };          // out score, history, its real / prev_real wid
                    if (lane < nrc) {
                        const int slot = pslot(w, lane);
                        const FtQuad q = ch_summary<NE>(wv.b + (size_t)slot * F::REC);      // out, out history, best, frame
                        if (present[slot] && q.z > lpth && q.x > nwt) {     // (best > lpth: prune_word_chan has stamped it)
                            const int32_t path = q.y;
                            it[0] = q.x; it[1] = path;
                            if (path != -1) { it[2] = BPC(tb, B_REAL, path); it[3] = BPC(tb, B_PREAL, path); }
                        }
                        tb.bss[bss_head + j0 + lane] = it[0];    // (no exit: WORST_SCORE, as the creation fills it)
                    }
                    // an exit has out > nwt > WORST_SCORE
                    const int32_t pm = ft_wave_excl<FtMax>(it[0]);     // maximum over the lanes before this one (none: below WORST_SCORE)
                    const bool rec = it[0] > kW && it[0] > pm;   // this lane creates or updates the entry
                    const unsigned long long m = __ballot(rec);
                    if (m == 0) continue;                        // (cannot happen: w_exit says one channel exits)
                    const int first = __ffsll(m) - 1, last = 63 - __clzll((long long)m);
                    const unsigned long long below = m & ((1ull << lane) - 1ull);
                    const int prev = below ? 63 - __clzll((long long)below) : first;
                    const int32_t pr = __shfl(it[2], prev), pp = __shfl(it[3], prev);      // (prev differs per lane)
                    const unsigned long long dm = __ballot(rec && below != 0 && (pr != it[2] || pp != it[3]));
                    int src = first;
                    if (dm) src = 63 - __clzll((long long)(m & ((1ull << (63 - __clzll((long long)dm))) - 1ull)));
                    const int32_t S = ft_lane(it[0], last), P = ft_lane(it[1], last);
                    const int32_t rw_real = ft_lane(it[2], src), rw_preal = ft_lane(it[3], src);
                    if (SMALL) {
                        // the entry's exit score for every right context phone (= its score-stack segment seen through the context map):
                        // to device memory for the predecessor searches of later frames, to LDS for this frame's word transitions
                        const int32_t xs = __shfl(it[0], cmrow);
                        if (lane < n_ci) {
                            if (bssx) bssx[(size_t)bpi * n_ci + lane] = xs;
                            if (bpi - bp0 < xcap) xfr[(bpi - bp0) * xst + 3 + lane] = xs;
                        }
                        if (lane == 0 && bpi - bp0 < xcap) {
                            int32_t *const x = xfr + (bpi - bp0) * xst;
                            x[0] = w;
                            x[1] = w_filler ? (rw_real != -1 ? rw_real : dbase_f[w]) : dbase_f[w];
                            x[2] = w_filler ? (rw_real != -1 ? rw_preal : -1) : rw_real;
                        }
                    }
                    if (lane == 0) {
                        word_lat_idx[w] = bpi;
                        BPC(tb, B_WID, bpi) = w; BPC(tb, B_FRAME, bpi) = f; BPC(tb, B_BP, bpi) = P; BPC(tb, B_SCORE, bpi) = S;
                        BPC(tb, B_SIDX, bpi) = bss_head + j0; BPC(tb, B_VALID, bpi) = 1;
                        BPC(tb, B_LAST, bpi) = dlast_f[w]; BPC(tb, B_LAST2, bpi) = w_last2;
                        // set_real_wid (:341-372) from the path it was last evaluated with (real ids are >= 0: -1 = no path)
                        if (w_filler) {
                            BPC(tb, B_REAL, bpi) = rw_real != -1 ? rw_real : dbase_f[w];
                            BPC(tb, B_PREAL, bpi) = rw_real != -1 ? rw_preal : -1;
                        }
                        else { BPC(tb, B_REAL, bpi) = dbase_f[w]; BPC(tb, B_PREAL, bpi) = rw_real; }
                    }
                }
            }
            ft_sync<SMALL>();                                    // (the new entries are first read behind word_transition's full barrier)
            FT_PROF(14);
        }
        if (!s_sc[6]) {
            // single-phone words (:1100-1127), one work-item per word; back-pointer positions in list order by prefix sums
            const int32_t nwt = s_sc[1] + p.wbeam, lpth = s_sc[1] + p.lponlybeam;
            const int wst = wstf;                                 // n1 <= n_w
            int32_t *const f_ex = cntf, *const f_new = cntf + wst, *const f_rc = cntf + 2 * wst;
            // (when the single-phone words and the next frame's active words are no more than the work-items -- nearly always -- each
            //  work-item keeps its word's flags and gets its prefix sums in registers, ft_scan_tid: no arrays, one barrier)
            const int naw_n0 = s_red[5];
            const bool one_each = (SMALL ? max(n1, naw_n0) : n1) + 1 <= NT;       // (slab layouts: no woff -- a block of the pool per active word)
            int my_ex = 0, my_new = 0, my_rc = 0;
            for (int i = tid; i <= n1; i += NT) {
                int ex = 0, nw = 0, rcn = 0;
                if (i < n1) {
                    const int c = W1 + i;
                    if (tv.at(c, F::FRAME) >= f && tv.at(c, F::BEST) > lpth) {
                        tv.at(c, F::FRAME) = nf;
                        if (tv.at(c, F::OUT) > nwt) {
                            const int w = w1w_f[i];
                            ex = 1;
                            if (word_lat_idx[w] == -1) nw = 1;   // (a single-phone word has no right-context fan-out: rcn stays 0,
                                                                  //  dict_is_single_phone = pronunciation length 1)
                        }
                    }
                }
                if (one_each) { my_ex = ex; my_new = nw; my_rc = rcn; }
                else { f_ex[i] = ex; f_new[i] = nw; f_rc[i] = rcn; }
            }
            // the NEXT frame's active words are complete since the positions step (s_red[5] of them in awln): their right-context
            // counts ride in the same scan, so that the next frame starts without a barrier and a scan of its own
            const int naw_n = naw_n0, n_sc = (SMALL ? max(n1, naw_n) : n1) + 1;
            const int32_t bpidx0 = s_sc[3], bss0 = s_sc[4];
            int32_t tot[3];
            if (one_each) {
                int32_t v[3] = { my_new, my_rc, 0 };
                if (SMALL && tid < naw_n) { const int w = awln[tid]; v[2] = wc_off[w + 1] - wc_off[w]; }
                ft_scan_tid<NT, 3, SMALL>(v, s_scan, tot);
                if (SMALL && tid < n_sc) woff[tid] = v[2];       // (read by the next frame's list building, behind barriers)
                my_new = v[0]; my_rc = v[1];
            }
            else {
                for (int i = n1 + 1 + tid; i < n_sc; i += NT) { f_new[i] = 0; f_rc[i] = 0; }
                if (SMALL)
                for (int i = tid; i < n_sc; i += NT) {
                    int k = 0;
                    if (i < naw_n) { const int w = awln[i]; k = wc_off[w + 1] - wc_off[w]; }
                    woff[i] = k;
                }
                ft_sync<SMALL>();
                if (SMALL) {
                    int32_t *const arr[3] = { f_new, f_rc, woff };
                    ft_block_scan_k<NT, 3, SMALL>(arr, n_sc, s_scan, tot);
                }
                else {                                       // (the slab layouts' evaluation lists need no woff: a block per active word)
                    int32_t *const arr[2] = { f_new, f_rc };
                    int32_t t2[2];
                    ft_block_scan_k<NT, 2, SMALL>(arr, n_sc, s_scan, t2);
                    tot[0] = t2[0]; tot[1] = t2[1]; tot[2] = 0;
                }
            }
            nwc_cur = tot[2];
            FT_PROF(20);
            for (int i = tid; i < n1; i += NT)
                if (one_each ? my_ex : f_ex[i]) {
                    int32_t bpi = bpidx0 + (one_each ? my_new : f_new[i]), bsh = bss0 + (one_each ? my_rc : f_rc[i]);
                    const int w = w1w_f[i];
                    const int32_t score = tv.at(W1 + i, F::OUT), path = tv.at(W1 + i, F::OUTH);
                    if (word_lat_idx[w] == -1 && bpi < tb.bp_cap && bsh + n_ci < tb.bss_cap) {
                        // the new entry of a single-phone word (ngram_search_save_bp's creating branch for pronunciation length 1,
                        // ngram_search.c:438-498): no score-stack segment, its real word ids by set_real_wid (:341-372) -- one trip
                        // to device memory (the path's ids) where the general routine makes four
                        const int32_t pr = path == -1 ? -1 : BPC(tb, B_REAL, path), pp = path == -1 ? -1 : BPC(tb, B_PREAL, path);
                        word_lat_idx[w] = bpi;
                        BPC(tb, B_WID, bpi) = w; BPC(tb, B_FRAME, bpi) = f; BPC(tb, B_BP, bpi) = path; BPC(tb, B_SCORE, bpi) = score;
                        BPC(tb, B_SIDX, bpi) = -1; BPC(tb, B_VALID, bpi) = 1; BPC(tb, B_LAST, bpi) = dlast_f[w]; BPC(tb, B_LAST2, bpi) = -1;
                        if (dfill_f[w]) {
                            BPC(tb, B_REAL, bpi) = path != -1 ? pr : dbase_f[w];
                            BPC(tb, B_PREAL, bpi) = path != -1 ? pp : -1;
                        }
                        else { BPC(tb, B_REAL, bpi) = dbase_f[w]; BPC(tb, B_PREAL, bpi) = path != -1 ? pr : -1; }
                        if (SMALL && bpi - bp0 < xcap) {
                            int32_t *const x = xfr + (bpi - bp0) * xst;
                            x[0] = w | kXSingle;
                            x[1] = dfill_f[w] ? (path != -1 ? pr : dbase_f[w]) : dbase_f[w];
                            x[2] = dfill_f[w] ? (path != -1 ? pp : -1) : (path != -1 ? pr : -1);
                            x[3] = score;
                        }
                    }
                    else {
                        if (SMALL) s_xbad = 1;                   // (the general routine: this frame's word transitions read the table itself)
                        if (!ft_save_bp(tb, dict, word_lat_idx, bpi, bsh, f, w, score, path, 0)) s_sc[6] = 1;
                    }
                }
            FT_PROF(21);
            // (bptable_maxwpf, when it applies, reads the frame's entries: a full barrier.  Otherwise nothing between here and
            //  word_transition's barrier reads what this step wrote: the counters work-item 0 advances below were read by every
            //  work-item before the scans' barriers)
            if (!(p.maxwpf == -1 || p.maxwpf == p.n_w)) __syncthreads();
            if (tid == 0) { s_sc[3] = bpidx0 + tot[0]; s_sc[4] = bss0 + tot[1]; }
        }
        if (tid == 0 && !s_sc[6]) {
            const int32_t bpidx = s_sc[3];
            // bptable_maxwpf (:1193-1241)
            if (!(p.maxwpf == -1 || p.maxwpf == p.n_w)) {
                int32_t bestscr = kMaxNegInt32; int bestbp = -1, n = 0;
                for (int bp = bp0; bp < bpidx; ++bp)
                    if (d_filler[BPC(tb, B_WID, bp)]) {
                        if (BPC(tb, B_SCORE, bp) > bestscr) { bestscr = BPC(tb, B_SCORE, bp); bestbp = bp; }
                        BPC(tb, B_VALID, bp) = 0; ++n;
                    }
                if (bestbp >= 0) { BPC(tb, B_VALID, bestbp) = 1; --n; }
                n = (bpidx - bp0) - n;
                for (; n > p.maxwpf; --n) {
                    int32_t worst = 0x7fffffff; int wbp = -1;
                    for (int bp = bp0; bp < bpidx; ++bp)
                        if (BPC(tb, B_VALID, bp) && BPC(tb, B_SCORE, bp) < worst) { worst = BPC(tb, B_SCORE, bp); wbp = bp; }
                    if (wbp < 0) break;
                    BPC(tb, B_VALID, wbp) = 0;
                }
            }
        }
        FT_PROF(15);
        // ---- word_transition (:1243-1427).  The best exit per right-context phone (earliest back-pointer among equals) and
        //      the single-phone words' best predecessors are maxima over (back-pointer, phone) / (word, back-pointer)
        //      pairs: one work-item per pair.
        unsigned long long *const brc_key = reinterpret_cast<unsigned long long *>(s_bins);      // [kFtMaxCi]
        int32_t *const brc_score = s_bins + 2 * kFtMaxCi, *const brc_path = s_bins + 3 * kFtMaxCi, *const brc_lc = s_bins + 4 * kFtMaxCi;
        for (int rc = tid; rc < n_ci; rc += NT) brc_key[rc] = ft_key_floor(kW);
        for (int i = tid; i < p.n1lm; i += NT) ckey[i] = ft_key_floor(kMaxNegInt32);
        if (tid == 0) s_red[6] = 0;
        __syncthreads();                                     // (device memory is exchanged here)
        const int n_awl_nxt = s_red[5];
        FT_PROF(23);
        if (s_sc[6]) break;
        const int bp1 = s_sc[3], nbp = bp1 - bp0;
        // small layout: the frame's entries are in LDS (xfr: word, real word ids, score per right context phone) when there are no
        // more than it holds, every one was written by the two fast paths above and bptable_maxwpf does not apply (it clears
        // `valid` flags): the pair searches below then go to device memory for the language model only
        const bool use_x = SMALL && nbp <= xcap && !s_xbad && (p.maxwpf == -1 || p.maxwpf == p.n_w);
        for (int j = tid; j < nbp * n_ci; j += NT) {
            // (the entry's columns in one trip, context map, stacked score: three, not five)
            const int bp = bp0 + j / n_ci, rc = j % n_ci;
            int wid; int32_t sc;
            if (use_x) {
                const int32_t *const x = xfr + (j / n_ci) * xst;
                const int32_t x0 = x[0];
                wid = x0 & 0xffffff; sc = (x0 & kXSingle) ? x[3] : x[3 + rc];
            }
            else { wid = BPC(tb, B_WID, bp); sc = ft_exit_score_bf(tb, rs_cimap, n_ci, bp, rc); }
            if (rc == 0) {
                word_lat_idx[wid] = -1; if (wid != p.finishwid) atomicAdd(&s_red[6], 1);
                BPC(tb, B_F0, bp) = bp0; BPC(tb, B_F1, bp) = bp1;      // (the frame is complete: see the predecessor search)
            }
            if (wid == p.finishwid) continue;
            if (sc > kW) atomicMax(&brc_key[rc], ft_key(sc, bp));
        }
        for (int j = tid; j < p.n1lm * nbp; j += NT) {             // in-LM single-phone words (:1331-1388): best predecessor
            const int i = j / nbp, bp = bp0 + j % nbp, w = w1w_f[i];
            int32_t valid, real, preal, ns;
            if (use_x) {
                const int32_t *const x = xfr + (j % nbp) * xst;
                const int32_t x0 = x[0];
                valid = 1; real = x[1]; preal = x[2]; ns = (x0 & kXSingle) ? x[3] : x[3 + dfirst_f[w]];
            }
            else {
                valid = BPC(tb, B_VALID, bp); real = BPC(tb, B_REAL, bp); preal = BPC(tb, B_PREAL, bp);
                ns = ft_exit_score_bf(tb, rs_cimap, n_ci, bp, dfirst_f[w]);
            }
            const int32_t lmv = p.use_trie ? 0 : ft_lm_dense(p, lmtab, dbase_f[w], real, preal);
            if (!valid) continue;
            if (ns != kW) ns += p.use_trie ? ft_lm(p, lmtab, dbase_f[w], real, preal) : lmv;
            atomicMax(&ckey[i], ft_key(ns, bp));
        }
        FT_PROF(24);
        ft_sync<SMALL>();
        FT_PROF(16);
        for (int rc = tid; rc < n_ci; rc += NT) {
            const unsigned long long k = brc_key[rc];
            const bool none = ft_key_none(k);
            const int path = none ? 0 : ft_key_bp(k);
            brc_score[rc] = none ? kW : ft_key_score(k); brc_path[rc] = path;
            // (an entry's last phone is its word's: ngram_search_save_bp, ngram_search.c:461)
            brc_lc[rc] = none ? 0 : (use_x ? dlast_f[xfr[(path - bp0) * xst] & 0xffffff] : BPC(tb, B_LAST, path));
        }
        ft_sync<SMALL>();
        FT_PROF(25);
        if (s_red[6] > 0) {
            // the tree roots on wavefronts 1.., the single-phone words on wavefront 0: their chains of dependent loads run side by
            // side (three loops one after the other cost the sum of their latencies).  The single-phone words keep the reference's
            // order within a work-item -- in-LM words, then <sil> and the noise words: a word entered by both loops must see the
            // first's result (measured: with the two on different wavefronts `numbers` diverged at random)
            if (tid >= 64 || NT == 64) {
                const int t0r = NT == 64 ? tid : tid - 64, str = NT == 64 ? NT : NT - 64;
                for (int i = t0r; i < R; i += str) {         // tree roots (:1306-1325)
                    const int ci = node_ci[i];
                    const int32_t ns = brc_score[ci] + p.nwpen + p.pip;
                    // (slab layouts: a root that is neither entered for this frame nor stamped for the next has a frame below f)
                    if (ns + ft_pen(ci) > thresh
                        && (SMALL ? (tv.at(i, F::FRAME) < f || ns > tv.at(i, F::SCORE)) : ((!rb_get(cur, i) && !rb_get(nxt, i)) || ns > tv.at(i, F::SCORE)))) {
                        if (!SMALL) rb_set(nxt, i);
                        ch_enter<NE>(tv, i, ns, brc_path[ci], nf);
                        tv.at(i, F::SENID) = ldiph[((size_t)ci * n_ci + node_ci2[i]) * n_ci + brc_lc[ci]];
                    }
                }
            }
            if (tid < 64) {
                for (int i = tid; i < p.n1lm; i += 64) {     // in-LM single-phone words (:1331-1388)
                    const int w = w1w_f[i];
                    const unsigned long long kk = ckey[i];
                    const int32_t ds = ft_key_none(kk) ? kMaxNegInt32 : ft_key_score(kk);
                    const int dbp = ft_key_none(kk) ? 0 : ft_key_bp(kk);
                    // (asked for before the tests: the frame has entries, entry 0 exists)
                    const int pw = (use_x && !ft_key_none(kk)) ? (xfr[(dbp - bp0) * xst] & 0xffffff) : BPC(tb, B_WID, dbp);
                    lt_dscr[w] = ds; lt_bp[w] = dbp;
                    if (w == p.startwid) continue;
                    const int c = W1 + i;
                    const int32_t ns = ds + p.pip;
                    if (ns + ft_pen(w1ci_f[i]) > thresh && (tv.at(c, F::FRAME) < f || ns > tv.at(c, F::SCORE))) {
                        ch_enter<NE>(tv, c, ns, dbp, nf);
                        tv.at(c, F::SENID) = ldiph[((size_t)w1ci_f[i] * n_ci + w1ci2_f[i]) * n_ci + dlast_f[pw]];
                    }
                }
                for (int w = p.filler_start - 1 + tid; w <= p.filler_end; w += 64) {    // <sil> and noise words (:1390-1426)
                    // slot filler_start - 1 stands for <sil>, which is handled whatever its place in the dictionary
                    const bool is_sil = w == p.filler_start - 1;
                    if (!is_sil && (w == p.startwid || w == p.silwid)) continue;
                    const int i = w1_of_word[is_sil ? p.silwid : w];
                    if (i < 0) continue;
                    const int c = W1 + i;
                    const int32_t ns = brc_score[p.sil_ci] + (is_sil ? p.silpen : p.fillpen) + p.pip;
                    if (ns + ft_pen(w1ci_f[i]) > thresh && (tv.at(c, F::FRAME) < f || ns > tv.at(c, F::SCORE)))
                        ch_enter<NE>(tv, c, ns, brc_path[p.sil_ci], nf);
                }
            }
        }
        ft_sync<SMALL>();
        FT_PROF(26);
        // ---- deactivate_channels (:1429-1450)
        if (SMALL) { for (int i = tid; i < R; i += NT) if (tv.at(i, F::FRAME) == f) ch_clear<NE>(tv, i); }
        else {
            for (int i = tid; i < R; i += NT) if (rb_get(cur, i) && !rb_get(nxt, i)) ch_clear<NE>(tv, i);
            __syncthreads();
            for (int i = tid; i < kFtMaxRootWords; i += NT) s_rb[cur][i] = 0u;     // (the frame after next's "stamped" bits)
        }
        for (int i = tid; i < n1; i += NT) if (tv.at(W1 + i, F::FRAME) == f) ch_clear<NE>(tv, W1 + i);
        if (tid == 0) {
            step[f * 4] = s_sc[0]; step[f * 4 + 1] = s_sc[1]; step[f * 4 + 2] = s_sc[3]; step[f * 4 + 3] = n_listed;
            ++s_sc[7];
            s_nev = 0; s_nroot = 0;                              // the next frame's evaluation list starts empty
        }
        FT_PROF(27);
        n_acl_cur = n_listed; n_awl_cur = n_awl_nxt;
        if (SMALL && nf < T) {                               // the next frame's score row and penalties take their place
            if (lists) lists_norm();
#if defined(__HIP_DEVICE_COMPILE__)
            else if (!kFtRowsDevice) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next row has landed in LDS (this wavefront's part)
#endif
            if (p.has_pl && tid < n_ci) s_pen[nxt * n_ci + tid] = pre_pen;
        }
        ft_sync<SMALL>();
        FT_PROF(17);
    }
#ifdef PSGPU_FT_PROFILE
    if (tid == 0 && bf.prof) for (int i = 0; i < 48; ++i) psgpu_as_global(bf.prof)[(size_t)blockIdx.x * 48 + i] = s_prof[i];
#endif
    __syncthreads();                                             // the table's last entries (device memory, other work-items') before the copy and the backtrace
    // (a resumed call: the entries the calls before have written are final and in the caller's columns already -- read here, from
    //  the block the lines below rewrite, rather than carried through the frames)
    const int bp_from = (live && (bf.live_mode & 2) && live[16] == kFtLiveMagic) ? live[8 + 3] : 0;
    __syncthreads();
    if (live && (bf.live_mode & 1)) {                            // psgpu_fwdtree_search_resume: what the next call starts from
        if (SMALL) for (int i = tid; i < L.rows_total; i += NT) live[kFtLiveHdr + i] = s_pool[i];
        if (tid < 8) live[8 + tid] = s_sc[tid];
        if (tid == 0) {
            live[0] = s_sc[7]; live[1] = n_acl_cur; live[2] = n_awl_cur; live[3] = (int32_t)evals_run; live[4] = nwc_cur;
            live[5] = (int32_t)(s_evals & 0xffffffffull); live[6] = (int32_t)(s_evals >> 32); live[7] = s_nsen;
            live[16] = kFtLiveMagic; live[17] = s_nfree;
        }
    }
    {   // the table in the caller's columns (bptbl_t, ngram_search.h:112-124): ft_table_out
        int32_t *const out = psgpu_as_global(bf.bp) + (size_t)blockIdx.x * kBpCols * bf.bp_cap;
        const int n_bp = s_sc[3];
        for (int j = bp_from * 4 + tid; j < n_bp * 4; j += NT) {       // (a quad of an entry per work-item; a resumed call: its own entries)
            const int i = j >> 2, q = j & 3;
            if (q == 3) continue;                                // (words 12-15: unused)
            const FtQuad v = *reinterpret_cast<const FtQuad *>(tb.bp + (size_t)i * kBpRow + 4 * q);
            out[(size_t)(4 * q) * bf.bp_cap + i] = v.x; out[(size_t)(4 * q + 1) * bf.bp_cap + i] = v.y;
            if (q < 2) { out[(size_t)(4 * q + 2) * bf.bp_cap + i] = v.z; out[(size_t)(4 * q + 3) * bf.bp_cap + i] = v.w; }
        }
    }
    // the hypothesis, while the table's tail is still in this compute unit's cache: a separate backtrace launch is one more
    // dispatch per batch, and a tiny dispatch issued beside ANOTHER stream's resident search kernel was measured stalling for
    // that kernel's whole duration (profiles/r03_overlap.txt).  (The table was written by this workgroup; the full barrier
    // above ordered those stores before this read.)  ONE walk down the path by work-item 0 -- a hop is one trip to device memory:
    // an entry's words lie in one line -- into LDS the frames no longer need, then every work-item writes its word in spoken order:
    // a live stream pays the walk at every step (as two walks of several loads a hop it was a fifth of a 10-frame step).
    int32_t *const bt = SMALL ? fb + L.evl : s_pool;              // {word, end frame, score} per word, last word first
    const int bt_cap = (SMALL ? (L.evl_cap + 1) / 2 : 9 * kPrIC) / 3;
    __syncthreads();                                              // (the pool has been saved: its words may be written over)
    if (tid == 0) {
        tb.idx[s_sc[7]] = s_sc[3];                               // ngram_fwdtree_finish: mark one past the last frame
        result[0] = s_sc[3]; result[1] = s_sc[4]; result[2] = s_sc[7]; result[3] = s_sc[6];
        result[4] = s_sc[0];                                     // ngs->best_score as the last frame left it
        result[5] = (int32_t)(s_evals & 0xffffffffull); result[6] = (int32_t)(s_evals >> 32); result[7] = s_nsen;
        if (bf.hyp) {
            // ngram_search_find_exit (ngram_search.c:500-544, frame_idx = -1): as ft_backtrace_one
            int32_t *const hn = psgpu_as_global(bf.hyp_n) + (size_t)blockIdx.x * 4;
            int n = 0, best = -1; int32_t best_score = kW;
            int f = s_sc[7] - 1;
            if (f >= 0) {
                const int end = tb.idx[f];
                while (f >= 0 && tb.idx[f] == end) --f;
                if (f >= 0)
                    for (int bp = tb.idx[f]; bp < end; ++bp) {
                        const int wid = BPC(tb, B_WID, bp);
                        if (wid == p.finishwid || BPC(tb, B_SCORE, bp) > best_score) { best_score = BPC(tb, B_SCORE, bp); best = bp; }
                        if (wid == p.finishwid) break;
                    }
            }
            for (int b = best; b != -1; ++n) {
                const FtQuad q = *reinterpret_cast<const FtQuad *>(tb.bp + (size_t)b * kBpRow);      // frame, valid, word, predecessor
                const int32_t sc = BPC(tb, B_SCORE, b);
                if (n < bt_cap) { bt[3 * n] = q.z; bt[3 * n + 1] = q.x; bt[3 * n + 2] = sc; }
                b = q.w;
            }
            hn[0] = n; hn[1] = best_score; hn[2] = best; hn[3] = 0;
            s_red[0] = n; s_red[1] = best;
        }
    }
    if (bf.hyp) {
        __syncthreads();
        const int n = s_red[0];
        int32_t *const hyp = psgpu_as_global(bf.hyp) + (size_t)blockIdx.x * bf.max_words * 4;
        if (n <= bt_cap) {
            // spoken position k = n - 1 - j of the walk's j-th word; the last max_words of them are kept; a word starts a frame after
            // its predecessor's end
            const int skip = n > bf.max_words ? n - bf.max_words : 0;
            for (int k = skip + tid; k < n; k += NT) {
                const int j = n - 1 - k;
                int32_t *const h = hyp + (size_t)(k - skip) * 4;
                h[0] = bt[3 * j]; h[1] = j + 1 < n ? bt[3 * (j + 1) + 1] + 1 : 0; h[2] = bt[3 * j + 1]; h[3] = bt[3 * j + 2];
            }
        }
        else if (tid == 0)                                       // (a path of more words than the LDS at hand holds: the two walks)
            ft_backtrace_one(tb, tb.idx, s_sc[7], p.finishwid, bf.max_words, hyp, psgpu_as_global(bf.hyp_n) + (size_t)blockIdx.x * 4);
    }
    // what the second pass inherits besides the tables: the permanent single-phone channels keep their per-state ssids
    // through hmm_clear (ngram_fwdflat_start, ngram_search_fwdflat.c:385-392)
    if (bf.mpx_out) {
        int32_t *const mo = psgpu_as_global(bf.mpx_out) + (size_t)blockIdx.x * (R + n1) * NE;
        for (int i = tid; i < (R + n1) * NE; i += NT) {
            const int q = i / NE;
            mo[i] = tv.at(q < R ? q : W1 + (q - R), F::SENID + i % NE);
        }
    }
    if (bf.w1_out) {
        int32_t *const w1o = psgpu_as_global(bf.w1_out);
        for (int i = tid; i < n1 * NE; i += NT)
            w1o[((size_t)blockIdx.x * n1 + i / NE) * NE + i % NE] = tv.at(W1 + i / NE, F::SENID + i % NE);
    }
}

// ngram_search_find_exit (ngram_search.c:500-544, frame_idx = -1) + the walk of ngram_search_bp_hyp / the segment iterator
// (:546-581, 903-1010) over an utterance's table: one work-item per utterance.  hyp [max_words][4] = wid, start frame, end
// frame, path score at the word's end, in spoken order; hyp_n [4] = number of words (may exceed max_words: then only the
// LAST max_words are stored), path score of the exit, exit back-pointer, 0.
__global__ void fwdtree_backtrace_kernel(const int32_t *__restrict__ bp_all, const int32_t *__restrict__ idx_all,
                                         const int32_t *__restrict__ res_all, int32_t n_utt, int32_t max_frames, int32_t bp_cap,
                                         int32_t finish_wid, int32_t max_words, int32_t *__restrict__ hyp_all, int32_t *__restrict__ hyp_n_all)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_utt) return;
    const FtTabCols tb = { bp_all + (size_t)u * kBpCols * bp_cap, bp_cap };
    ft_backtrace_one(tb, idx_all + (size_t)u * (max_frames + 2), res_all[(size_t)u * 8 + 2], finish_wid, max_words,
                     hyp_all + (size_t)u * max_words * 4, hyp_n_all + (size_t)u * 4);
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

// the per-utterance arrays: offsets of the fast arrays (LDS pool or slab) and of the slab.  Returns false when the LDS layout
// was asked for and does not fit.
static bool ft_layout(FtDev &d, bool small)
{
    const int ne = d.n_emit, rec = ne == 3 ? 16 : 24, words = 3 * ne + 6;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t r = o; o += (n + 3) & ~(int64_t)3; return (int32_t)r; };
    FtLay &L = d.lay;
    memset(&L, 0, sizeof L);
    // (slab layouts: records of the roots and the single-phone words only -- the other tree nodes' channels exist while they are listed, compact)
    L.rec = take(small ? (int64_t)d.CH * words : ((int64_t)d.R + d.n1) * rec);
    if (small) {                         // the interleaved blocks (FtCol): the per-node and per-word arrays as columns
        L.node_blk = take(((int64_t)d.N + 1) * kFtNodeCols); L.word_blk = take(((int64_t)d.n_w + 2) * kFtWordCols);
    }
    else {
        o = (o + 31) & ~(int64_t)31; L.acl0 = take(d.N); o = (o + 31) & ~(int64_t)31; L.acl1 = take(d.N); L.awl0 = take(d.n_w); L.awl1 = take(d.n_w);
        L.word_active = take(d.n_w); L.word_lat_idx = take(d.n_w); L.lt_sf = take(d.n_w); L.lt_dscr = take(d.n_w); L.lt_bp = take(d.n_w);
        L.cand_mark = take(d.n_w);
        L.cand_wid = take(d.n_w + 1); L.cand_score = take(d.n_w + 1); L.cand_bp = take(d.n_w + 1);
    }
    // (the pruning's snapshot: columns of the node block in the LDS layout; the slab layouts keep it in the nodes' records)
    L.cnt = take(d.cnt_words);
    L.cnt2 = take(d.n_w + 2); L.cnt3 = take(d.n_w + 2); L.woff = take(d.n_w + 2); L.ckey = take(2 * ((int64_t)d.n_w + 2));
    L.present = take(small ? ((int64_t)d.TOT + 3) / 4 : (int64_t)d.rc_blocks * kFtRcBlk / 4);
    if (!small) { L.wblk = take(d.n_w); L.rcfree = take(d.rc_blocks); }
    d.lb_words = 0;
    if (small) {
        L.pen = take(2 * (int64_t)d.n_ci);
        L.kids = take(d.M);
        L.tp = take(((int64_t)d.n_tmat * ne * (ne + 1) + 3) / 4);
        L.w1w = take(d.n1); L.w1ci = take(d.n1); L.w1ci2 = take(d.n1);
        // what only scoring from top-N lists (psgpu_fwdtree_search_lists_dev) needs lies at the pool's end -- the lists, the
        // log-add table, the listed senones: a launch that reads score rows asks for less LDS
        if (!kFtRowsDevice) L.row = take(((int64_t)d.n_sen + 1) / 2 + 4);
        const int64_t tail = 2 * (int64_t)kFtMaxChains + 512 / 4 + kFtListCap / 2 + kFtThreads / 64 * kSenStreams + 16
                             + (kFtRowsDevice ? ((int64_t)d.n_sen + 1) / 2 + 8 : 0);
        // the frame's evaluation list takes what is left
        // (16-bit entries; a list of every channel when that fits -- then it cannot overflow -- else what is left)
        const int64_t left = ((int64_t)kFtLdsWords - o - tail) & ~(int64_t)3, full = (int64_t)d.R + d.N + d.n1 + d.TOT + 4;
        if (2 * left < kFtMinEvl || d.n_sen > kFtMaxSen || d.n_w > 512 || d.n_ci > 64 || d.N + d.n1 >= 0x8000) return false;
        L.evl_cap = (int32_t)std::min<int64_t>(2 * left, full);
        if (const char *cap = getenv("PSGPU_FWDTREE_EVL_CAP"))   // (a test's knob: a list that small fills up, status 2)
            L.evl_cap = (int32_t)std::max<int64_t>(64, std::min<int64_t>(L.evl_cap, atoll(cap)));
        L.evl = take((L.evl_cap + 1) / 2);
        // the frame's exits for the word transitions lie in the evaluation list's words: the list's last reader of a frame is
        // prune_word_chan, the exits are written after it and read until the frame's end
        L.xfr = L.evl; L.xfr_cap = ((L.evl_cap + 1) / 2) / (d.n_ci + 3);
        if (const char *cap = getenv("PSGPU_FWDTREE_XFR_CAP"))   // (a test's knob: frames with more exits take the table's path)
            L.xfr_cap = (int32_t)std::max<int64_t>(0, std::min<int64_t>(L.xfr_cap, atoll(cap)));
        L.rows_total = (int32_t)o;
        if (kFtRowsDevice) L.row = take(((int64_t)d.n_sen + 1) / 2 + 4);     // (scoring from lists computes the frame's scores into it)
        L.l_cw = take(kFtMaxChains); L.l_sc = take(kFtMaxChains); L.l_la = take(512 / 4); L.l_list = take(kFtListCap / 2);
        L.l_norm = take(kFtThreads / 64 * kSenStreams);
    }
    else {
        L.itb = take(4 * (int64_t)d.R);
        const int nd = ne == 3 ? 2 : 3;
        L.ccap = (int32_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)d.N - d.R, d.listed_cap));
        // (the streamed arrays start on 128-byte lines -- a wavefront's 1 KB request then touches eight of them, not nine -- and so does each
        //  of the arrays inside: the capacity is a multiple of 32 places)
        L.ccap = (L.ccap + 31) & ~31;
        auto take_line = [&](int64_t n) { o = (o + 31) & ~(int64_t)31; return take(n); };
        L.cq = take_line(2 * (int64_t)(nd + 2) * L.ccap * 4); L.csum = take_line(4 * (int64_t)L.ccap); L.cxfer = take_line(2 * (int64_t)L.ccap);
        L.cxpl = take_line(2 * 2 * (int64_t)L.ccap); L.cperm = take_line(L.ccap);
        if (((int64_t)d.N + 31) / 32 > kFtMaxBitWords || d.R > 32 * kFtMaxRootWords) return false;     // (the listed-nodes bitmap and the roots' live in LDS)
        d.lb_words = (int32_t)(((int64_t)d.N + 31) / 32);
        L.evl_cap = (int32_t)std::min<int64_t>((int64_t)d.n1 + (int64_t)d.rc_blocks * kFtRcBlk + 64, 0x7ffffff0);      // (the word level's: single-phone words, pool channels)
        L.evl = take(L.evl_cap);
    }
    if (o > 0x7fffff00) return false;
    L.total = (int32_t)o;
    d.small = small ? 1 : 0;
    int64_t g = 0;
    auto gtake = [&](int64_t n) { const int64_t r = g; g += (n + 31) & ~(int64_t)31; return r; };       // 128-byte lines
    d.g_wrec = gtake(small ? (int64_t)d.TOT * rec : (int64_t)d.rc_blocks * kFtRcBlk * rec);
    d.g_fast = small ? 0 : gtake(o);
    d.per = g;
    return true;
}

// slab layouts: the dynamic LDS pool of a launch with `nt` work-items an utterance -- the pruning's item arrays (9 arrays of 2 nt words
// + 16), the listed-nodes bitmap, its words' prefix populations (uint16), the frame's score row, the transition matrices, and in what
// is left of kFtSlabPoolWords the rank -> list position table (uint16 entries; a frame that lists more keeps the table in the slab)
static void ft_slab_pool(FtDev &d, int nt)
{
    int32_t o = 9 * kFtIpt * nt + 16;
    auto take = [&](int32_t n) { const int32_t r = o; o += (n + 3) & ~3; return r; };
    d.lds_lb = take(d.lb_words);
    d.lds_pre = take(d.lb_words / 2 + 2);
    d.lds_row = take((d.n_sen + 1) / 2 + 4);
    d.lds_tp = take((d.n_tmat * d.n_emit * (d.n_emit + 1) + 3) / 4);
    d.lds_perm = o;
    // (PSGPU_FT_POOL_WORDS_BIG: an A/B build's knob -- a pool of half a compute unit's LDS lets two workgroups of a large tree share one)
    const int32_t budget = nt == kFtThreadsBig ? std::min(kFtSlabPoolWords, PSGPU_FT_POOL_WORDS_BIG) : kFtSlabPoolWords;
    const int32_t left = std::max(0, budget - o) & ~3;
    d.lds_perm_cap = (int32_t)std::min<int64_t>(std::min<int64_t>(2 * (int64_t)left, 65528), ((int64_t)d.lay.ccap + 7) & ~(int64_t)7);     // (16-bit positions)
    if (const char *cap = getenv("PSGPU_FWDTREE_PERM_CAP"))     // (a test's knob: frames that list more take the table in the slab)
        d.lds_perm_cap = (int32_t)std::max<int64_t>(0, std::min<int64_t>(d.lds_perm_cap, atoll(cap) & ~7ll));
    o += d.lds_perm_cap / 2;
    d.lds_words = (o + 3) & ~3;
}

#ifdef PSGPU_FT_PROFILE
// a profiling build: the per-phase cycle counts of the handle's PREVIOUS search, averaged over its utterances, per frame -- printed when
// the handle is used next (or freed), so that the launch itself does not wait and searches of other handles run beside it as they do in
// the product build
// (an interval ends at its marker: "x: to barrier" = work-item 0's own work, the next interval = its wait at the barrier + the rest)
static void ft_prof_report(psgpu_fwdtree_s *m)
{
    if (!m->prof_buf) return;
    hipEventSynchronize(m->prof_ev);
    const int n_utt = m->prof_n_utt;
    {
        static const char *const names[32] = { "top: lists, senone marks", "slab pairs: bisection", "normaliser", "evaluate: after barrier (prefetch issue, beam)", "prune: snapshot", "prune: decide, barrier | slab: chunk scans + candidates",
            "next active list | slab: pairs", "last-phone candidates", "predecessor search: decode + max", "entering", "active words", "prune_word_chan",
            "positions (scans)", "slab pairs: loads + decisions", "exits", "single-phone: barrier + counters", "word_transition: pairs, barrier", "frame end: row to LDS",
            "evaluate: loop", "evaluate: barrier", "single-phone: flags + scan", "single-phone: save", "slab pairs: compaction", "word_transition: init + barrier",
            "word_transition: pair loops", "word_transition: decode keys", "word_transition: enter", "deactivate + step", "prune: decide loop | slab: chunk items",
            "predecessor search: exit scores + scan", "predecessor search: pairs", "slab pairs: barrier" };
        std::vector<long long> h((size_t)48 * n_utt);
        std::vector<int32_t> r((size_t)8 * n_utt);
        hipMemcpy(h.data(), m->prof_buf, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), m->prof_res, 4 * r.size(), hipMemcpyDeviceToHost);
        hipFree(m->prof_buf); m->prof_buf = nullptr;
        double frames = 0, tot = 0, acc[32] = {};
        for (int u = 0; u < n_utt; ++u) { frames += r[(size_t)u * 8 + 2]; for (int i = 0; i < 32; ++i) acc[i] += (double)h[(size_t)u * 48 + i]; }
        for (int i = 0; i < 32; ++i) tot += acc[i];
        fprintf(stderr, "fwdtree_kernel profile: %d utterances, %.0f frames, %.0f cycles per frame (work-item 0)\n", n_utt, frames, tot / (frames > 0 ? frames : 1));
        static const int order[] = { 0, 2, 18, 19, 3, 4, 28, 5, 1, 13, 22, 31, 6, 7, 29, 30, 8, 9, 10, 11, 12, 14, 20, 21, 15, 23, 24, 16, 25, 26, 27, 17 };
        for (int w = 1; w < 4; ++w) {
            double a1 = 0, a2 = 0;
            for (int u = 0; u < n_utt; ++u) { a1 += (double)h[(size_t)u * 48 + 32 + 4 * w + 1]; a2 += (double)h[(size_t)u * 48 + 32 + 4 * w + 2]; }
            fprintf(stderr, "  wavefront %d: evaluate loop %9.0f, its barrier %9.0f cycles/frame\n", w, a1 / (frames > 0 ? frames : 1), a2 / (frames > 0 ? frames : 1));
        }
        {
            double n16 = 0, c16 = 0, mx = 0, e16 = 0;
            for (int u = 0; u < n_utt; ++u) { n16 += (double)h[(size_t)u * 48 + 44]; c16 += (double)h[(size_t)u * 48 + 45]; mx = std::max(mx, (double)h[(size_t)u * 48 + 46]); e16 += (double)h[(size_t)u * 48 + 47]; }
            fprintf(stderr, "  evaluation over 16k cycles: %.0f of %.0f frames, %.0f cycles and %.1f evaluations each on average; longest %.0f\n", n16, frames, c16 / (n16 > 0 ? n16 : 1), e16 / (n16 > 0 ? n16 : 1), mx);
        }
        for (int i : order) fprintf(stderr, "  %2d %-48s %9.0f cycles/frame  %5.1f %%\n", i, names[i], acc[i] / (frames > 0 ? frames : 1), 100.0 * acc[i] / (tot > 0 ? tot : 1));
    }
}
#endif

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
    if (!(d.n_emit == 3 || d.n_emit == 5) || d.n_ci < 1 || d.n_ci > kFtMaxCi || d.N < 1 || d.n_w < 1) {
        psgpu_set_error("fwdtree: unsupported shape (n_emit %d, n_ci %d, tree nodes %d, words %d)", d.n_emit, d.n_ci, d.N, d.n_w);
        delete m;
        return PSGPU_EINVAL;
    }
    const size_t nci3 = (size_t)d.n_ci * d.n_ci * d.n_ci, n1 = (size_t)d.n_w + 1;
    std::vector<int32_t> parent(d.N, -1), w1_of(d.n_w, -1), wc_off(d.n_w + 1, 0), kid_off(d.N + 1, 0), kids(d.M > 0 ? d.M : 1, 0);
    // the children of every node, in sibling order, as one array (CSR): a walk along the sibling pointers is a chain of
    // dependent loads, a range of an array is not
    {
        int k = 0;
        for (int i = 0; i < d.N; ++i) {
            kid_off[i] = k;
            for (int c = t->node_child[i]; c >= 0; c = t->node_sib[c]) {
                if (k >= d.M || c < d.R || c >= d.N || parent[c] != -1) {
                    psgpu_set_error("fwdtree: the tree tables are not a tree (node %d, child %d)", i, c);
                    delete m;
                    return PSGPU_EINVAL;
                }
                parent[c] = i; kids[k++] = c;
            }
        }
        kid_off[d.N] = k;
    }
    for (int i = 0; i < d.n1; ++i) w1_of[t->w1_wid[i]] = i;
    int64_t tot = 0;
    for (int w = 0; w < d.n_w; ++w) {
        wc_off[w] = (int32_t)tot;
        if (t->dict_pronlen[w] > 1) tot += t->rssid_n[t->dict_last[w] * d.n_ci + t->dict_last2[w]];
    }
    if (tot > 0x7ffffff0) { psgpu_set_error("fwdtree: %lld last-phone channels", (long long)tot); delete m; return PSGPU_EINVAL; }
    wc_off[d.n_w] = (int32_t)tot;
    d.TOT = (int32_t)tot;
    d.CH = d.N + d.n1;
    d.n_tmat = t->n_tmat;
    {   // (PSGPU_FWDTREE_LISTED_CAP / PSGPU_FWDTREE_RC_BLOCKS: tests' knobs -- capacities that small fill up, status 4 / 5)
        const char *lc = getenv("PSGPU_FWDTREE_LISTED_CAP"), *rb = getenv("PSGPU_FWDTREE_RC_BLOCKS");
        d.listed_cap = (int32_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)d.N - d.R, lc ? atoll(lc) : 65536));
        d.rc_blocks = (int32_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)d.n_w, rb ? atoll(rb) : 2048));
        const char *wc = getenv("PSGPU_FWDTREE_WL_CAP");
        d.wl_cap = wc ? (int32_t)std::max<int64_t>(0, atoll(wc)) : 0x7fffffff;
        d.wl_global = 0;
    }
    d.cnt_words = std::max(std::max(2 * (d.R + d.N + 1), 4 * d.n_w + 4), kFtMaxSen / 32 + 4);     // (two arrays over the roots and listed nodes; .. + 4: the senone bitmap's word populations)
    d.node_ci = ft_up(m, t->node_ci, d.N, &rc); d.node_ci2 = ft_up(m, t->node_ci2, d.N, &rc);
    d.node_ssid = ft_up(m, t->node_ssid, d.N, &rc); d.node_tmat = ft_up(m, t->node_tmat, d.N, &rc);
    d.kid_off = ft_up(m, kid_off.data(), (size_t)d.N + 1, &rc); d.kids = ft_up(m, kids.data(), kids.size(), &rc);
    {
        if (d.N >= (1 << 24) || d.n_ci > 128) { psgpu_set_error("fwdtree: %d tree nodes (at most 2^24 - 1)", d.N); psgpu_fwdtree_free(m); return PSGPU_EINVAL; }
        // per node, side by side: its children (child | ci << 24, sibling order), then the words whose penultimate phone it is (word |
        // last phone << 24, the homophone chain in order: the last-phone candidates of ngram_search_fwdtree.c:824-870) -- the pruning
        // works on (item, entry) pairs, one work-item each, and a walk along the chain would be a loop of dependent loads
        if (d.n_w >= (1 << 24)) { psgpu_set_error("fwdtree: %d dictionary words (at most 2^24 - 1)", d.n_w); psgpu_fwdtree_free(m); return PSGPU_EINVAL; }
        std::vector<int32_t> kc, xoff((size_t)d.N + 1, 0), xnw(d.N, 0);
        kc.reserve(kids.size() + (size_t)d.n_w);
        for (int c = 0; c < d.N; ++c) {
            xoff[c] = (int32_t)kc.size();
            for (int k = kid_off[c]; k < kid_off[c + 1]; ++k) kc.push_back((int32_t)((uint32_t)kids[k] | ((uint32_t)t->node_ci[kids[k]] << 24)));
            int guard = 0;
            for (int w = t->node_penult_wid[c]; w >= 0 && guard <= d.n_w; w = t->homophone_set[w], ++guard) {
                kc.push_back((int32_t)((uint32_t)w | ((uint32_t)t->dict_last[w] << 24)));
                ++xnw[c];
            }
            if (guard > d.n_w || xnw[c] > 0xffff) { psgpu_set_error("fwdtree: node %d: a chain of %d penultimate-phone words", c, xnw[c]); psgpu_fwdtree_free(m); return PSGPU_EINVAL; }
        }
        xoff[d.N] = (int32_t)kc.size();
        if (kc.empty()) kc.push_back(0);
        d.kids_ci = ft_up(m, kc.data(), kc.size(), &rc);
        const int nsq = d.n_emit <= 3 ? 4 : 8;
        std::vector<int32_t> q1((size_t)d.N * 4, 0), q2((size_t)d.N * 4, 0), qs((size_t)d.N * nsq, 0), st1((size_t)d.N * 4, 0);
        for (int c = 0; c < d.N; ++c)
            if (kid_off[c + 1] - kid_off[c] > 0xffff || t->node_tmat[c] > 0xffff || t->node_tmat[c] < 0) {
                psgpu_set_error("fwdtree: node %d has %d children / transition matrix %d (16-bit fields)", c, kid_off[c + 1] - kid_off[c], t->node_tmat[c]);
                psgpu_fwdtree_free(m);
                return PSGPU_EINVAL;
            }
        for (int c = 0; c < d.N; ++c) {
            const int k0 = kid_off[c], nk = kid_off[c + 1] - k0, pw = t->node_penult_wid[c];
            q1[(size_t)c * 4] = (int32_t)((uint32_t)(parent[c] < 0 ? 0 : parent[c]) | ((uint32_t)t->node_ci[c] << 24));
            const int x0 = xoff[c], nx = xoff[c + 1] - x0;
            q1[(size_t)c * 4 + 1] = x0; q1[(size_t)c * 4 + 2] = nk | (xnw[c] << 16); q1[(size_t)c * 4 + 3] = nx > 0 ? kc[x0] : -1;
            {
                uint32_t pk[3] = { 0, 0, 0 };
                for (int k = 0; k <= d.n_emit; ++k) {       // (the states' senones, then the transition matrix: 16 bits each)
                    const uint32_t v = k < d.n_emit ? (uint32_t)t->sseq[(size_t)t->node_ssid[c] * d.n_emit + k] : (uint32_t)t->node_tmat[c];
                    pk[k >> 1] |= (v & 0xffffu) << (16 * (k & 1));
                }
                st1[(size_t)c * 4] = (int32_t)pk[0]; st1[(size_t)c * 4 + 1] = (int32_t)pk[1]; st1[(size_t)c * 4 + 2] = (int32_t)pk[2];
                st1[(size_t)c * 4 + 3] = nx > 0 ? kc[x0] : -1;
            }
            q2[(size_t)c * 4] = pw; q2[(size_t)c * 4 + 1] = pw >= 0 ? t->dict_last[pw] : 0; q2[(size_t)c * 4 + 2] = pw >= 0 ? t->homophone_set[pw] : -1;
            for (int k = 0; k < d.n_emit; ++k) qs[(size_t)c * nsq + k] = t->sseq[(size_t)t->node_ssid[c] * d.n_emit + k];
        }
        d.node_q1 = ft_up(m, q1.data(), q1.size(), &rc); d.node_q2 = ft_up(m, q2.data(), q2.size(), &rc);
        d.node_sen = ft_up(m, qs.data(), qs.size(), &rc);
        d.node_st1 = ft_up(m, st1.data(), st1.size(), &rc);
        // ... and what ngram_search_alloc_all_rc (ngram_search.c:583-633) gives a word's right-context channels, per slot
        std::vector<int32_t> ss((size_t)std::max<int64_t>(tot, 1) * nsq, 0);
        for (int w = 0; w < d.n_w; ++w) {
            if (t->dict_pronlen[w] <= 1) continue;
            const int last = t->dict_last[w], last2 = t->dict_last2[w], n = t->rssid_n[last * d.n_ci + last2];
            for (int r = 0; r < n; ++r) {
                const int ssid = t->rssid_ssid[((size_t)last * d.n_ci + last2) * d.n_ci + r];
                int32_t *q = ss.data() + ((size_t)wc_off[w] + r) * nsq;
                for (int k = 0; k < d.n_emit; ++k) q[k] = t->sseq[(size_t)ssid * d.n_emit + k];
                q[nsq - 1] = t->ci_tmat[last];
            }
        }
        d.slot_sen = ft_up(m, ss.data(), ss.size(), &rc);
    }
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
    // layout: LDS when everything the tree level touches fits the pool (PSGPU_FWDTREE_LAYOUT=slab forces the other one: the
    // parity tests run both)
    const char *force = getenv("PSGPU_FWDTREE_LAYOUT");
    if ((force && !strcmp(force, "slab")) || !ft_layout(d, true)) {
        if (!ft_layout(d, false)) {
            psgpu_set_error("fwdtree: %d tree nodes (the slab layouts keep a bitmap of at most %d nodes in LDS), or per-utterance arrays beyond 8 GB",
                            d.N, 32 * kFtMaxBitWords);
            psgpu_fwdtree_free(m);
            return PSGPU_EINVAL;
        }
    }
    if (getenv("PSGPU_FT_DUMP_LAYOUT")) {                 // (a measuring aid: the layout as an initialiser list)
        const int32_t *w = reinterpret_cast<const int32_t *>(&d.lay);
        fprintf(stderr, "FtLay {");
        for (size_t i = 0; i < sizeof(FtLay) / 4; ++i) fprintf(stderr, "%s%d", i ? "," : "", w[i]);
        fprintf(stderr, "}\n");
    }
    *out = m;
    return PSGPU_OK;
}

const LmDev *psgpu_lm_dev(const psgpu_lm_t *lm);     // psgpu_lm.hip
const LmDev *psgpu_lm_dev_ptr(const psgpu_lm_t *lm);

int psgpu_fwdtree_set_lm(psgpu_fwdtree_t *m, const psgpu_lm_t *lm)
{
    PSGPU_REQUIRE(m && lm, "psgpu_fwdtree_set_lm: NULL argument");
    const LmDev *d = psgpu_lm_dev(lm);
    PSGPU_REQUIRE(d->n_words == m->d.n_w, "psgpu_fwdtree_set_lm: the model maps %d dictionary words, the search has %d", d->n_words, m->d.n_w);
    m->d.trie_dev = psgpu_lm_dev_ptr(lm);
    m->d.use_trie = 1;
    return PSGPU_OK;
}

int32_t psgpu_fwdtree_n_single_phone_words(const psgpu_fwdtree_t *m) { return m ? m->d.n1 : 0; }

int psgpu_fwdtree_use_slab_layout(psgpu_fwdtree_t *m)
{
    PSGPU_REQUIRE(m, "psgpu_fwdtree_use_slab_layout: NULL argument");
    if (!m->d.small) return PSGPU_OK;
    // (the layout is a table of offsets and the choice of kernel: the model's tables on the device stay as they are; the work slab
    //  is sized by the next search call)
    if (!ft_layout(m->d, false)) {
        const bool back = ft_layout(m->d, true);
        (void)back;
        psgpu_set_error("fwdtree: the search's per-utterance arrays exceed 8 GB in the slab layout");
        return PSGPU_EINVAL;
    }
    return PSGPU_OK;
}

int psgpu_fwdtree_grow(psgpu_fwdtree_t *m, int32_t status)
{
    PSGPU_REQUIRE(m && (status == 4 || status == 5 || status == 6),
                  "psgpu_fwdtree_grow: status 4 (listed tree nodes), 5 (right-context channel pool) or 6 (the word level's LDS arrays)");
    FtDev &d = m->d;
    PSGPU_REQUIRE(!d.small, "psgpu_fwdtree_grow: the LDS layout has none of these capacities");
    if (status == 6) {
        PSGPU_REQUIRE(!d.wl_global, "psgpu_fwdtree_grow: the word level's arrays are in the slab already");
        d.wl_global = 1; m->live_valid = false;
        return PSGPU_OK;
    }
    const int32_t old_l = d.listed_cap, old_r = d.rc_blocks;
    if (status == 4) {
        PSGPU_REQUIRE((int64_t)d.listed_cap < (int64_t)d.N - d.R, "psgpu_fwdtree_grow: the compact channels hold every tree node already");
        d.listed_cap = (int32_t)std::min<int64_t>((int64_t)d.N - d.R, 2 * (int64_t)d.listed_cap);
    }
    else {
        PSGPU_REQUIRE(d.rc_blocks < d.n_w, "psgpu_fwdtree_grow: the pool holds a block per word already");
        d.rc_blocks = (int32_t)std::min<int64_t>(d.n_w, 2 * (int64_t)d.rc_blocks);
    }
    if (!ft_layout(d, false)) {
        d.listed_cap = old_l; d.rc_blocks = old_r;
        const bool back = ft_layout(d, false);
        (void)back;
        psgpu_set_error("psgpu_fwdtree_grow: the search's per-utterance arrays would exceed 8 GB");
        return PSGPU_EINVAL;
    }
    m->live_valid = false;                               // (a saved search's arrays lay where the old layout put them)
    return PSGPU_OK;
}

int psgpu_fwdtree_full_capacity(psgpu_fwdtree_t *m)
{
    PSGPU_REQUIRE(m, "psgpu_fwdtree_full_capacity: NULL argument");
    FtDev &d = m->d;
    if (d.small) return PSGPU_OK;                        // (the LDS layout has none of these capacities)
    const int32_t old_l = d.listed_cap, old_r = d.rc_blocks, old_w = d.wl_global;
    d.listed_cap = (int32_t)std::max<int64_t>(1, (int64_t)d.N - d.R);
    d.rc_blocks = std::max<int32_t>(1, d.n_w);
    d.wl_global = 1;
    if (!ft_layout(d, false)) {
        d.listed_cap = old_l; d.rc_blocks = old_r; d.wl_global = old_w;
        const bool back = ft_layout(d, false);
        (void)back;
        psgpu_set_error("psgpu_fwdtree_full_capacity: the search's per-utterance arrays would exceed 8 GB");
        return PSGPU_EINVAL;
    }
    if (d.listed_cap != old_l || d.rc_blocks != old_r || d.wl_global != old_w) m->live_valid = false;
    return PSGPU_OK;
}

int psgpu_fwdtree_layout(const psgpu_fwdtree_t *m, int32_t *lds_layout, int64_t *slab_bytes_per_utt)
{
    PSGPU_REQUIRE(m, "psgpu_fwdtree_layout: NULL argument");
    if (lds_layout) *lds_layout = m->d.small;
    if (slab_bytes_per_utt) *slab_bytes_per_utt = 4 * m->d.per;
    return PSGPU_OK;
}

#ifdef PSGPU_FT_PROFILE
static void ft_prof_report(psgpu_fwdtree_s *m);
#endif
void psgpu_fwdtree_free(psgpu_fwdtree_t *m)
{
    if (!m) return;
#ifdef PSGPU_FT_PROFILE
    ft_prof_report(m);
#endif
    for (void *p : m->allocs) hipFree(p);
    hipFree(m->slab);
    hipFree(m->bssx);
    hipFree(m->bpa);
    hipFree(m->live);
    delete m;
}

// Search n_utt utterances.  senscr_dev [total][scr_stride] int16: the scores acmod_score hands the search for each
// frame; penalties_dev [total][n_ci]; utt_off_dev [n_utt + 1].  Per utterance u the columns of the back-pointer
// table go to bp_dev + u * 10 * bp_cap, the score stack to bss_dev + u * bss_cap, the frame marks to
// idx_dev + u * (max_frames + 2), per-frame diagnostics to step_dev + u * max_frames * 4, and
// result_dev + u * 8 = {n back-pointers, score-stack length, frames searched, status (1 = a table was full)}.
// Asynchronous on `stream` (the work slab belongs to the handle: one search at a time per handle).
int psgpu_fwdtree_search_dev(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                             const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                             int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                             int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                             int32_t pl_window, int32_t *w1_ssid_out_dev, void *stream)
{
    return psgpu_fwdtree_search_session_dev(m, senscr_dev, scr_stride, penalties_dev, utt_off_dev, n_utt, max_frames, bp_cap, bss_cap,
                                            bp_dev, bss_dev, idx_dev, step_dev, result_dev, raw_scores, pl_window, w1_ssid_out_dev,
                                            nullptr, nullptr, stream);
}

int32_t psgpu_fwdtree_n_mpx_channels(const psgpu_fwdtree_t *m) { return m ? m->d.R + m->d.n1 : 0; }

struct FtListsArg { const int32_t *tsc; const uint32_t *tcw; const uint8_t *mixw, *sen2cb, *la; int32_t total, chains, density, la_size; };

static int ft_search(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride, const FtListsArg *ls,
                     const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                     int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                     int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                     int32_t pl_window, int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev,
                     int32_t *mpx_ssid_out_dev, void *stream);

int psgpu_fwdtree_search_session_dev(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride,
                                     const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                     int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                                     int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                                     int32_t pl_window, int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev,
                                     int32_t *mpx_ssid_out_dev, void *stream)
{
    PSGPU_REQUIRE(senscr_dev || n_utt == 0, "psgpu_fwdtree_search_session_dev: NULL score rows");
    return ft_search(m, senscr_dev, scr_stride, nullptr, penalties_dev, utt_off_dev, n_utt, max_frames, bp_cap, bss_cap, bp_dev, bss_dev,
                     idx_dev, step_dev, result_dev, raw_scores, pl_window, w1_ssid_out_dev, mpx_ssid_in_dev, mpx_ssid_out_dev, stream);
}

int32_t psgpu_fwdtree_can_score_lists(const psgpu_fwdtree_t *m, const psgpu_ptm_view_t *v)
{
    return (m && v && m->d.small && v->n_feat == kSenStreams && v->topn == kSenTopn && v->n_mgau * v->n_feat <= kFtMaxChains
            && v->n_mgau * v->n_feat <= kFtThreads && v->n_sen == m->d.n_sen && v->n_sen <= kFtMaxSen && v->n_sen < 65536
            && v->logadd8_size >= 256 && v->mixw_sen != nullptr) ? 1 : 0;
}

int psgpu_fwdtree_search_lists_dev(psgpu_fwdtree_t *m, const psgpu_ptm_view_t *v, const int32_t *topn_score_dev,
                                   const uint8_t *topn_cw_dev, int32_t total_frames,
                                   const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                                   int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                                   int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t pl_window,
                                   int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev, int32_t *mpx_ssid_out_dev, void *stream)
{
    PSGPU_REQUIRE(m && v && (n_utt == 0 || (topn_score_dev && topn_cw_dev)), "psgpu_fwdtree_search_lists_dev: NULL argument");
    PSGPU_REQUIRE(psgpu_fwdtree_can_score_lists(m, v),
                  "psgpu_fwdtree_search_lists_dev: needs the LDS layout and a 3-stream top-4 scorer of at most %d chains (psgpu_fwdtree_can_score_lists)",
                  kFtMaxChains);
    PSGPU_REQUIRE(((uintptr_t)topn_score_dev & 15) == 0 && ((uintptr_t)topn_cw_dev & 3) == 0, "psgpu_fwdtree_search_lists_dev: misaligned lists");
    const FtListsArg ls = { topn_score_dev, reinterpret_cast<const uint32_t *>(topn_cw_dev), v->mixw_sen, v->sen2cb, v->logadd8, total_frames,
                            v->n_mgau * v->n_feat, v->n_density, v->logadd8_size };
    return ft_search(m, nullptr, 0, &ls, penalties_dev, utt_off_dev, n_utt, max_frames, bp_cap, bss_cap, bp_dev, bss_dev, idx_dev, step_dev,
                     result_dev, 1, pl_window, w1_ssid_out_dev, mpx_ssid_in_dev, mpx_ssid_out_dev, stream);
}

static int ft_search(psgpu_fwdtree_t *m, const int16_t *senscr_dev, int64_t scr_stride, const FtListsArg *ls,
                     const int32_t *penalties_dev, const int32_t *utt_off_dev, int32_t n_utt,
                     int32_t max_frames, int32_t bp_cap, int32_t bss_cap, int32_t *bp_dev, int32_t *bss_dev,
                     int32_t *idx_dev, int32_t *step_dev, int32_t *result_dev, int32_t raw_scores,
                     int32_t pl_window, int32_t *w1_ssid_out_dev, const int32_t *mpx_ssid_in_dev,
                     int32_t *mpx_ssid_out_dev, void *stream)
{
    PSGPU_REQUIRE(m && n_utt >= 0 && max_frames >= 0 && bp_cap > 0 && bss_cap > 0, "psgpu_fwdtree_search_dev: bad argument");
    PSGPU_REQUIRE(!raw_scores || (m->d.n_sen <= kFtMaxSen && pl_window >= 0), "raw-score mode: n_sen %d > %d or negative pl_window",
                  m->d.n_sen, kFtMaxSen);
    PSGPU_REQUIRE(m->d.lm || m->d.use_trie, "psgpu_fwdtree_search_dev: no language model (dense table or psgpu_fwdtree_set_lm)");
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE((senscr_dev || ls) && penalties_dev && utt_off_dev && bp_dev && bss_dev && idx_dev && step_dev && result_dev,
                  "psgpu_fwdtree_search_dev: NULL device buffer");
    FtDev d = m->d;
    // the LDS layout copies score rows as dwords: rows must start on 4-byte boundaries
    if (!ls && d.small && ((scr_stride & 1) || ((uintptr_t)senscr_dev & 3))) ft_layout(d, false);
    // the slab layouts keep a frame's score row and the transition matrices in fixed LDS arrays (the LDS layout sizes its pool from
    // the model: ft_layout refused it at create if it did not fit)
    PSGPU_REQUIRE(d.small || (d.n_sen <= kFtSlabSen && d.n_tmat * d.n_emit * (d.n_emit + 1) <= kFtSlabTp),
                  "fwdtree (slab layout): %d senones / %d transition matrices (the kernel keeps a frame's row of at most %d scores and %d "
                  "bytes of matrices in LDS)", d.n_sen, d.n_tmat, kFtSlabSen, kFtSlabTp);
    hipStream_t st = (hipStream_t)stream;
    const size_t need = (size_t)d.per * n_utt;
    if (need > m->slab_words) {
        if (m->slab) { PSGPU_HIP(hipStreamSynchronize(st)); hipFree(m->slab); m->slab = nullptr; m->slab_words = 0; }
        m->live_valid = false;                           // (a saved search's channels lay there)
        PSGPU_HIP(hipMalloc((void **)&m->slab, sizeof(int32_t) * need));
        m->slab_words = need;
    }
    // LDS layout: the exits' scores by right context phone, one row of n_ci per back-pointer
    const size_t need_x = d.small ? (size_t)n_utt * (size_t)bp_cap * (size_t)d.n_ci : 0;
    if (need_x > m->bssx_words) {
        if (m->bssx) { PSGPU_HIP(hipStreamSynchronize(st)); hipFree(m->bssx); m->bssx = nullptr; m->bssx_words = 0; }
        m->live_valid = false;
        PSGPU_HIP(hipMalloc((void **)&m->bssx, sizeof(int32_t) * need_x));
        m->bssx_words = need_x;
    }
    const size_t need_a = (size_t)n_utt * (size_t)bp_cap * kBpRow;
    if (need_a > m->bpa_words) {
        if (m->bpa) { PSGPU_HIP(hipStreamSynchronize(st)); hipFree(m->bpa); m->bpa = nullptr; m->bpa_words = 0; }
        m->live_valid = false;
        PSGPU_HIP(hipMalloc((void **)&m->bpa, sizeof(int32_t) * need_a));
        m->bpa_words = need_a;
    }
    FtBufs bf;
    bf.bpa = m->bpa;
    bf.bssx = d.small ? m->bssx : nullptr;
    bf.slab = m->slab; bf.bp = bp_dev; bf.bss = bss_dev; bf.idx = idx_dev; bf.step = step_dev; bf.res = result_dev;
    bf.w1_out = w1_ssid_out_dev;
    bf.mpx_in = mpx_ssid_in_dev; bf.mpx_out = mpx_ssid_out_dev;
    bf.tsc = nullptr; bf.tcw = nullptr; bf.mixw = bf.sen2cb = bf.la = nullptr; bf.ls_total = bf.ls_chains = bf.ls_density = bf.ls_la_size = 0;
    if (ls) {
        bf.tsc = ls->tsc; bf.tcw = ls->tcw; bf.mixw = ls->mixw; bf.sen2cb = ls->sen2cb; bf.la = ls->la;
        bf.ls_total = ls->total; bf.ls_chains = ls->chains; bf.ls_density = ls->density; bf.ls_la_size = ls->la_size;
    }
    bf.hyp = m->hyp_out; bf.hyp_n = m->hyp_n_out; bf.max_words = m->hyp_max_words;
    m->hyp_out = nullptr; m->hyp_n_out = nullptr; m->hyp_max_words = 0;      // (one call's worth)
    bf.lag = m->lag_next; m->lag_next = 0;
    // psgpu_fwdtree_search_resume (one call's worth, like the lag)
    bf.ext = m->ext_next; m->ext_next = nullptr;
    bf.live = nullptr; bf.live_mode = m->live_next; m->live_next = 0;
    if (bf.live_mode) {
        if (bf.live_mode & 2)
            PSGPU_REQUIRE(m->live_valid && m->live_n_utt == n_utt && m->live_bp_cap == bp_cap && m->live_bss_cap == bss_cap
                          && m->live_small == d.small && m->live_max_frames == max_frames && m->live_raw == raw_scores && m->live_window == pl_window && m->live_bp == bp_dev
                          && m->live_bss == bss_dev && m->live_idx == idx_dev && m->live_step == step_dev,
                          "psgpu_fwdtree_search_resume: nothing to resume -- the handle's previous search call must have kept its state "
                          "(mode bit 0) for the same utterances, table capacities and table buffers");
        const size_t need_l = (size_t)n_utt * (size_t)(kFtLiveHdr + (d.small ? d.lay.rows_total : 0));
        if (need_l > m->live_words) {
            PSGPU_REQUIRE(!(bf.live_mode & 2), "psgpu_fwdtree_search_resume: the saved state does not fit its buffer");
            if (m->live) { PSGPU_HIP(hipStreamSynchronize(st)); hipFree(m->live); m->live = nullptr; m->live_words = 0; }
            PSGPU_HIP(hipMalloc((void **)&m->live, sizeof(int32_t) * need_l));
            m->live_words = need_l;
        }
        bf.live = m->live;
    }
    m->live_valid = (bf.live_mode & 1) != 0;
    if (m->live_valid) {
        m->live_small = d.small; m->live_n_utt = n_utt; m->live_bp_cap = bp_cap; m->live_bss_cap = bss_cap; m->live_max_frames = max_frames; m->live_raw = raw_scores;
        m->live_window = pl_window; m->live_bp = bp_dev; m->live_bss = bss_dev; m->live_idx = idx_dev; m->live_step = step_dev;
    }
    bf.prof = nullptr;
#ifdef PSGPU_FT_PROFILE
    ft_prof_report(m);
    PSGPU_HIP(hipMalloc((void **)&bf.prof, sizeof(long long) * 48 * (size_t)n_utt));
#endif
    bf.bp_cap = bp_cap; bf.bss_cap = bss_cap; bf.max_frames = max_frames;
    // ~10^4 active channels per frame on a large tree: 16 waves per utterance
    const bool big = d.N + d.R > kFtBigNodes || d.n_w > 1024;
    if (!d.small) {
        ft_slab_pool(d, big ? kFtThreadsBig : kFtThreads);
        PSGPU_REQUIRE(d.lds_words <= kFtSlabPoolWords, "fwdtree (slab layout): the kernel's LDS arrays take %d words (at most %d)", d.lds_words, kFtSlabPoolWords);
    }
    const size_t pool_bytes = d.small ? sizeof(int32_t) * (size_t)(ls ? d.lay.total : d.lay.rows_total) : sizeof(int32_t) * (size_t)d.lds_words;
#if defined(__HIPCC__)                    /* a pool that takes the workgroup's LDS beyond the default 64 KB (scoring from lists): say so once */
#define FT_DYN_LDS(NE, NT, SMALL, LISTS)                                                                              \
        if (pool_bytes + 4096 > 65536) {                                                                              \
            static const hipError_t attr_rc = hipFuncSetAttribute((const void *)fwdtree_kernel<NE, NT, SMALL, LISTS>, \
                              hipFuncAttributeMaxDynamicSharedMemorySize,                                            \
                              (int)((SMALL) ? sizeof(int32_t) * kFtLdsWords : sizeof(int32_t) * kFtSlabPoolWords));          \
            PSGPU_HIP(attr_rc);                                                                                       \
        }
#else
#define FT_DYN_LDS(NE, NT, SMALL, LISTS)
#endif
#define FT_LAUNCH(NE, NT, SMALL, LISTS) do {                                                                          \
        FT_DYN_LDS(NE, NT, SMALL, LISTS)                                                                              \
        hipLaunchKernelGGL((fwdtree_kernel<NE, NT, SMALL, LISTS>), dim3(n_utt), dim3(NT), pool_bytes, st,                \
                           d, senscr_dev, scr_stride, penalties_dev, utt_off_dev, raw_scores, pl_window, bf);        \
    } while (0)
    if (d.n_emit == 3) {
        if (d.small && ls) FT_LAUNCH(3, kFtThreads, true, true);
        else if (d.small) FT_LAUNCH(3, kFtThreads, true, false);
        else if (!big && !d.wl_global) FT_LAUNCH(3, kFtThreads, false, true);
        else if (!big) FT_LAUNCH(3, kFtThreads, false, false);
        else if (!d.wl_global) FT_LAUNCH(3, kFtThreadsBig, false, true);
        else FT_LAUNCH(3, kFtThreadsBig, false, false);
    }
    else {
        if (d.small && ls) FT_LAUNCH(5, kFtThreads, true, true);
        else if (d.small) FT_LAUNCH(5, kFtThreads, true, false);
        else if (!big && !d.wl_global) FT_LAUNCH(5, kFtThreads, false, true);
        else if (!big) FT_LAUNCH(5, kFtThreads, false, false);
        else if (!d.wl_global) FT_LAUNCH(5, kFtThreadsBig, false, true);
        else FT_LAUNCH(5, kFtThreadsBig, false, false);
    }
#undef FT_LAUNCH
#undef FT_DYN_LDS
    PSGPU_HIP(hipGetLastError());
#ifdef PSGPU_FT_PROFILE
    m->prof_buf = bf.prof; m->prof_res = result_dev; m->prof_n_utt = n_utt;
    if (!m->prof_ev) hipEventCreateWithFlags(&m->prof_ev, hipEventDisableTiming);
    hipEventRecord(m->prof_ev, st);
#endif
    return PSGPU_OK;
}

int psgpu_fwdtree_search_lag(psgpu_fwdtree_t *m, int32_t lag)
{
    PSGPU_REQUIRE(m && lag >= 0, "psgpu_fwdtree_search_lag: bad argument");
    m->lag_next = lag;
    return PSGPU_OK;
}

int psgpu_fwdtree_search_streams(psgpu_fwdtree_t *m, const int32_t *ext_dev)
{
    PSGPU_REQUIRE(m, "psgpu_fwdtree_search_streams: NULL argument");
    m->ext_next = ext_dev;
    return PSGPU_OK;
}

int psgpu_fwdtree_search_restart(psgpu_fwdtree_t *m, int32_t u, void *stream)
{
    PSGPU_REQUIRE(m && m->live_valid && u >= 0 && u < m->live_n_utt, "psgpu_fwdtree_search_restart: no saved search for that utterance");
    const size_t per = (size_t)(kFtLiveHdr + (m->live_small ? m->d.lay.rows_total : 0));
    PSGPU_HIP(hipMemsetAsync(m->live + (size_t)u * per + 16, 0, sizeof(int32_t), (hipStream_t)stream));
    return PSGPU_OK;
}

int psgpu_fwdtree_search_resume(psgpu_fwdtree_t *m, int32_t mode)
{
    PSGPU_REQUIRE(m && mode >= 0 && mode <= 3, "psgpu_fwdtree_search_resume: bad argument");
    m->live_next = mode;
    return PSGPU_OK;
}

int psgpu_fwdtree_hyp_out(psgpu_fwdtree_t *m, int32_t *hyp_dev, int32_t *hyp_n_dev, int32_t max_words)
{
    PSGPU_REQUIRE(m && ((hyp_dev && hyp_n_dev && max_words > 0) || (!hyp_dev && !hyp_n_dev)), "psgpu_fwdtree_hyp_out: bad argument");
    m->hyp_out = hyp_dev; m->hyp_n_out = hyp_dev ? hyp_n_dev : nullptr; m->hyp_max_words = hyp_dev ? max_words : 0;
    return PSGPU_OK;
}

int psgpu_fwdtree_backtrace_dev(const psgpu_fwdtree_t *m, const int32_t *bp_dev, const int32_t *idx_dev, const int32_t *result_dev,
                                int32_t n_utt, int32_t max_frames, int32_t bp_cap, int32_t max_words, int32_t *hyp_dev,
                                int32_t *hyp_n_dev, void *stream)
{
    PSGPU_REQUIRE(m && n_utt >= 0 && max_frames >= 0 && bp_cap > 0 && max_words > 0, "psgpu_fwdtree_backtrace_dev: bad argument");
    if (n_utt == 0) return PSGPU_OK;
    PSGPU_REQUIRE(bp_dev && idx_dev && result_dev && hyp_dev && hyp_n_dev, "psgpu_fwdtree_backtrace_dev: NULL device buffer");
    hipLaunchKernelGGL(fwdtree_backtrace_kernel, dim3((n_utt + 63) / 64), dim3(64), 0, (hipStream_t)stream, bp_dev, idx_dev, result_dev,
                       n_utt, max_frames, bp_cap, m->d.finishwid, max_words, hyp_dev, hyp_n_dev);
    PSGPU_HIP(hipGetLastError());
    return PSGPU_OK;
}

}  // extern "C"
